"""dev soak: opv_decode_payloads against the CPU oracle's FrameDecoder restatement on N random payloads of many kinds
(encoded frames at several noise levels, pure noise, few-level inputs full of trellis ties, tiny and huge scales, sparse
zeros): metric, quantised taps, deinterleaved taps, Viterbi bits and bytes must be equal for every one.
usage: decoder_soak.py [N=20000] [seed=1]"""
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tests"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def make(n, seed):
    from oracle_lib import Oracle
    o = Oracle()
    rng = np.random.default_rng(seed)
    soft = np.empty((n, 2144))
    for k in range(n):
        kind = k % 8
        if kind in (0, 1, 2):            # a real coded frame: bit 1 -> negative soft, plus noise of three strengths
            bits = o.encode_frame(rng.integers(0, 256, 134, dtype=np.uint8)).astype(np.float64)
            s = (1.0 - 2.0 * bits) * 2.4e11
            soft[k] = s + rng.standard_normal(2144) * 2.4e11 * (0.3, 0.8, 1.6)[kind]
        elif kind == 3:
            soft[k] = rng.standard_normal(2144) * 3e10
        elif kind == 4:                  # few levels: exact ties in the trellis and on quantiser boundaries
            soft[k] = rng.integers(-3, 4, 2144) * 1e10
        elif kind == 5:
            soft[k] = rng.standard_normal(2144) * 10.0 ** rng.uniform(-12, 3)      # around the 1e-10 drop threshold
        elif kind == 6:
            soft[k] = rng.standard_normal(2144) * 1e200
        else:
            s = rng.standard_normal(2144) * 1e11
            s[rng.random(2144) < rng.uniform(0.1, 0.99)] = 0.0
            soft[k] = s
    return soft


def oracle_chunk(soft):
    from oracle_lib import Oracle
    o = Oracle()
    return [o.frame_decode(s) for s in soft]


def main():
    from amd_lib import load
    amd = load()
    soft = make(N, SEED)
    d = amd.Demod(1, max_samples=1 << 20)
    r = d.decode_payloads(soft, taps=True)
    d.close()
    W = 12
    with ProcessPoolExecutor(W) as ex:
        exp = [e for part in ex.map(oracle_chunk, np.array_split(soft, W * 8)) for e in part]
    bad = 0
    dropped = 0
    for k, e in enumerate(exp):
        ok = r["metrics"][k] == e["metric"]
        if e["metric"] < 0:
            dropped += 1
        else:
            ok = ok and np.array_equal(r["q"][k], e["q"]) and np.array_equal(r["deint"][k], e["deint"]) \
                and np.array_equal(r["bits"][k], e["bits"]) and np.array_equal(r["frames"][k], e["frame"])
        if not ok:
            bad += 1
            if bad <= 5:
                print("MISMATCH payload", k, "kind", k % 8, "metric", int(r["metrics"][k]), "vs", e["metric"])
    met = np.array([e["metric"] for e in exp])
    print(f"{N} payloads (seed {SEED}): {bad} mismatches; {dropped} dropped by the scale threshold, "
          f"{int((met == 0).sum())} perfect, metric range {met[met >= 0].min()}..{met.max()}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
