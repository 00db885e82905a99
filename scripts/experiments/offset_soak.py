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


from soak_inputs import offset_openings, oracle_offset_chunk as oracle_chunk  # noqa: E402  (shared with tests/test_gpu_parity.py)


def make(seed):
    return offset_openings(seed, S)


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
