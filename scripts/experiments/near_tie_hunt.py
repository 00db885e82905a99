"""dev tool (CPU only): find openings on which estimate_offset (reference src/opv-demod.cpp:131-202) is decided by a NEAR tie - two
candidates whose energies differ by less than 1e-11 relative but are not equal - i.e. the inputs on which the offset search's
near-tie guard fires and the last places of sin / cos decide. One in several hundred ordinary noisy openings is one; the GPU test
tests/test_gpu_parity.py::test_offset_search_near_ties_on_ordinary_captures regenerates the ones found here from their (seed, k).
usage: near_tie_hunt.py <first_seed> <n_seeds> [openings per seed = 256]      prints one line per hit and a summary"""
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tests"))
from soak_inputs import near_tie_class, offset_opening  # noqa: E402


def work(args):
    seed, per = args
    from oracle_lib import Oracle
    o = Oracle()
    hits = []
    for k in range(per):
        x = offset_opening(seed, k)
        off, e = o.estimate_offset(x, energies=True)
        cls = near_tie_class(e)
        if cls:
            hits.append((seed, k, off, cls))
    return hits


if __name__ == "__main__":
    s0, ns = int(sys.argv[1]), int(sys.argv[2])
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    total = 0
    with ProcessPoolExecutor(8) as ex:
        for hits in ex.map(work, [(s, per) for s in range(s0, s0 + ns)]):
            for h in hits:
                print(h, flush=True)
            total += len(hits)
    print(f"{total} near-tie openings among {ns * per}")
