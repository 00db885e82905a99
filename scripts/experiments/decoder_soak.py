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


from soak_inputs import decoder_payloads as make, oracle_decode_chunk as oracle_chunk  # noqa: E402  (shared with tests/test_gpu_parity.py)


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
