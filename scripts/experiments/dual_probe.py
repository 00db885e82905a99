"""dev: one front-end mapping (1, 4 or 16 streams per wave; default 1) on S streams x F clean frames: front-end time, cycles per symbol.
(Rounds 2 - 5 also ran the comparison mappings -1 / -2 through this script; those left the tree in round 6.)"""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from __graft_entry__ import load_opv_amd
amd = load_opv_amd()
S, F = int(sys.argv[1]), int(sys.argv[2])
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
iq = amd.modulate(amd.bert_frames(F))
n = iq.size // 2
d_iq = torch.from_numpy(iq).cuda()
d = amd.Demod(S, max_samples=n + 64, streaming=True)
d.set_frontend(mode)
d.enable_timing(True)
for rep in range(2):
    d.reset()
    for s in range(S):
        d.attach(s, d_iq.data_ptr(), n, eof=True)
    d.process(); d.sync()
kt = d.kernel_times()
nsym = d.state(0).total_symbols
info = np.array([d.wave_info(s) for s in range(S)], dtype=np.float64)
print(f"mode {mode} S={S} F={F}: front-end {kt['msk_frontend']:.2f} ms = {S * n / kt['msk_frontend'] / 1e3:.0f} Msamples/s; "
      f"cycles/symbol {np.median(info[:, 2]) / nsym:.0f}; frames {len(d.pop_frames(0)[0])}")
