"""dev: S independent streams x F frames, every one its own BERT capture through the device channel (workload.generate: what
bench.py's extras.many_streams_unique_captures runs), one opv_process on the automatic mapping. For PMC passes of the
sixteen-streams-per-wave front-end on data that cannot hit in L2:
  rocprofv3 --pmc FETCH_SIZE --kernel-include-regex k_msk_frontend_x16 --output-format csv -d OUT -- python3 many_unique.py 32768 8
usage: many_unique.py [S=32768] [F=8]"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_opv_amd, load_pkg_module  # noqa: E402

amd, workload = load_opv_amd(), load_pkg_module("workload")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
n = amd.lib().opv_tx_modulated_samples(F)
d = amd.Demod(S, max_samples=n + 64, streaming=True)
d_iq, tx, n = workload.generate(amd, d, torch, dev, range(S), F, 16.0)
d.enable_timing(True)
for k in range(S):
    d.attach(k, d_iq[k].data_ptr(), n, eof=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
d.process()
d.sync()
dt = time.perf_counter() - t0
kt = d.kernel_times()
print(f"S={S} F={F}: {d.frontend_kernel()}, whole process {S * n / dt / 1e6:.0f} Msamples/s ({dt * 1e3:.1f} ms), front-end alone "
      f"{S * n / kt['msk_frontend'] / 1e3:.0f} Msamples/s ({kt['msk_frontend']:.2f} ms), algorithmic bytes of the front-end {S * n * 4.0015 / 1e9:.2f} GB, kernels {kt}")
d.close()
