"""dev: run S streams x F frames (the same clean capture attached to every stream) on one mapping."""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from __graft_entry__ import load_opv_amd
amd = load_opv_amd()
S, F, spw = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iq = amd.modulate(amd.bert_frames(F))
n = iq.size // 2
d_iq = torch.from_numpy(iq).cuda()
d = amd.Demod(S, max_samples=n + 64, streaming=True)
d.set_frontend(spw)
d.enable_timing(True)
for rep in range(2):
    d.reset()
    for s in range(S):
        d.attach(s, d_iq.data_ptr(), n, eof=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d.process(); d.sync()
    dt = time.perf_counter() - t0
fr, _ = d.pop_frames(S - 1)
kt = d.kernel_times()
fe = S * n / (kt["msk_frontend"] * 1e-3) / 1e6
print(f"S={S} F={F} spw={spw}: front-end alone {fe:.1f} Msamples/s ({kt['msk_frontend']:.2f} ms), whole process {S * n / dt / 1e6:.1f} Msamples/s ({dt * 1e3:.2f} ms), kernels {kt}, frames {len(fr)}")
