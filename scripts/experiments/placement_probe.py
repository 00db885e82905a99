"""dev: S streams x F frames on the one-wave-per-stream front-end; prints, from the kernel's own wave-info tap,
the histogram of waves per CU / per SIMD and the shader clock every wave measured (s_memtime / s_memrealtime)."""
import collections
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from __graft_entry__ import load_opv_amd

amd = load_opv_amd()
S, F = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
iq = amd.modulate(amd.bert_frames(F))
n = iq.size // 2
d_iq = torch.from_numpy(iq).cuda()
d = amd.Demod(S, max_samples=n + 64, streaming=True)
d.set_frontend(int(sys.argv[4]) if len(sys.argv) > 4 else 1)
d.enable_timing(True)
for rep in range(reps):
    d.reset()
    for s in range(S):
        d.attach(s, d_iq.data_ptr(), n, eof=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d.process()
    d.sync()
    dt = time.perf_counter() - t0
kt = d.kernel_times()
info = np.array([d.wave_info(s) for s in range(S)], dtype=np.uint64)
nsym = d.state(0).total_symbols
hw, xcc, cyc, ticks = info[:, 0], info[:, 1] & 0xF, info[:, 2].astype(np.float64), info[:, 3].astype(np.float64)
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu
per_cu = collections.Counter(cukey.tolist())
per_simd = collections.Counter(((cukey << 2) | simd).tolist())
hc = collections.Counter(per_cu.values())
hs = collections.Counter(per_simd.values())
clk = cyc / ticks * 100.0  # MHz
fe = S * n / (kt["msk_frontend"] * 1e-3) / 1e6
print(f"S={S} F={F}: front-end {kt['msk_frontend']:.2f} ms = {fe:.0f} Msamples/s ({fe / S:.1f} per wave); "
      f"CUs used {len(per_cu)}, SIMDs used {len(per_simd)}; waves/CU {dict(sorted(hc.items()))}; "
      f"waves/SIMD {dict(sorted(hs.items()))}; in-kernel clock MHz min/med/max {clk.min():.0f}/{np.median(clk):.0f}/{clk.max():.0f}; "
      f"cycles/symbol min/med/max {cyc.min() / nsym:.0f}/{np.median(cyc) / nsym:.0f}/{cyc.max() / nsym:.0f}; "
      f"wave time ms min/med/max {ticks.min() / 1e5:.2f}/{np.median(ticks) / 1e5:.2f}/{ticks.max() / 1e5:.2f}")
