"""dev: the frame decoder alone (opv_decode_payloads: k_payload_scale + k_decode_payloads) on N payloads, for a kernel trace:
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_dec -- python3 scripts/experiments/decode_bench.py 65536
usage: decode_bench.py [N=65536] [reps=3]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tests"))
from amd_lib import load  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
amd = load()
rng = np.random.default_rng(3)
base = rng.standard_normal((256, 2144)) * 2.4e11 + np.where(rng.random((256, 2144)) < 0.5, 2.4e11, -2.4e11) * 2.0
soft = np.tile(base, (N // 256, 1))
d = amd.Demod(1, max_samples=1 << 20)
for r in range(REPS):
    t0 = time.perf_counter()
    out = d.decode_payloads(soft)
    dt = time.perf_counter() - t0
    print(f"rep {r}: {N} payloads in {dt * 1e3:.1f} ms end to end (host copies included), metric range {out['metrics'].min()}..{out['metrics'].max()}", flush=True)
d.close()
