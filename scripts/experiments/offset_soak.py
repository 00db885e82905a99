"""dev soak: the offset search (estimate_offset, reference src/opv-demod.cpp:131-202) against the CPU oracle on many random
openings of a capture - random start inside a BERT run, carrier offset in and beyond the +/-1530 Hz span, level, Eb/N0 from
0 dB to clean, lengths from 3000 to 45000 samples (the search uses min(N, 40000)), some pure noise and some digital silence
with a burst. The estimate must be EQUAL for every one; streams on which the near-tie guard fired are counted.
usage: offset_soak.py [rounds=4] [seed=1]   (512 streams per round, batch mode)"""
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tests"))
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
S = 512


def make(seed):
    from oracle_lib import Oracle, impair
    o = Oracle()
    rng = np.random.default_rng(seed)
    base = o.modulate(o.bert_frames(3, "K%d" % (seed % 1000), first=seed))
    caps = []
    for k in range(S):
        n = int(rng.choice([3000, 8000, 20000, 39999, 40000, 40001, 45000]))
        at = int(rng.integers(0, base.size // 2 - n - 1))
        x = base[2 * at: 2 * (at + n)]
        kind = k % 10
        if kind == 8:                                            # noise only
            x = np.zeros_like(x)
            x = impair(x + 1, amp=float(rng.uniform(50, 3000)), ebn0_db=-20.0, seed=seed * 1000 + k)
        elif kind == 9:                                          # digital silence with a burst somewhere
            y = np.zeros_like(x)
            a, b = sorted(int(v) for v in rng.integers(0, n, 2))
            y[2 * a: 2 * b] = x[2 * a: 2 * b]
            x = y
        else:
            ebn0 = None if kind == 0 else float(rng.uniform(0, 25))
            x = impair(x, amp=float(rng.uniform(100, 16000)), f0_hz=float(rng.uniform(-2500, 2500)), ebn0_db=ebn0,
                       seed=seed * 1000 + k)
        caps.append(np.ascontiguousarray(x))
    return caps


def oracle_chunk(caps):
    from oracle_lib import Oracle
    o = Oracle()
    return [o.estimate_offset(c) for c in caps]


def main():
    from amd_lib import load
    amd = load()
    bad = guarded = total = 0
    for r in range(ROUNDS):
        caps = make(SEED * 100 + r)
        d = amd.Demod(S, max_samples=46000, streaming=False)
        d.receive(caps)
        got = [d.state(k) for k in range(S)]
        d.close()
        with ProcessPoolExecutor(14) as ex:
            parts = [caps[i::14] for i in range(14)]
            res = list(ex.map(oracle_chunk, parts))
        exp = [None] * S
        for i, part in enumerate(res):
            exp[i::14] = part
        for k in range(S):
            total += 1
            guarded += got[k].offset_ties != 0
            if got[k].est_offset_hz != exp[k]:
                bad += 1
                if bad <= 8:
                    print("MISMATCH round", r, "stream", k, "n", caps[k].size // 2, got[k].est_offset_hz, "vs", exp[k], "ties", got[k].offset_ties)
        print(f"round {r}: {total} streams so far, {bad} mismatches, guard fired on {guarded}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
