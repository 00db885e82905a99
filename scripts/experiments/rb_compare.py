"""dev: one noisy, frequency-shifted stream through the one-wave front-end (product + swap reductions, mode -1) and the row-broadcast variant (mode 1):
first soft symbol / chunk-log entry where they differ, and by how much."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from __graft_entry__ import load_opv_amd
amd = load_opv_amd()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 6
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else -1200.0
sigma = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
iq = amd.modulate(amd.bert_frames(F))
n = iq.size // 2
d_clean = torch.from_numpy(iq).cuda()
d_iq = torch.empty_like(d_clean)
out = {}
for mode in (-1, 1):
    d = amd.Demod(1, max_samples=n + 64, streaming=True)
    d.set_frontend(mode)
    if mode == -1:
        d.channel(d_clean.data_ptr(), d_iq.data_ptr(), n, gain=1.0, f0_hz=f0, sigma=sigma, seed=5)
        d.sync()
    d.attach(0, d_iq.data_ptr(), n, eof=True)
    d.process(); d.sync()
    out[mode] = (np.array(d.soft(0)), np.array(d.chunks(0)), d.pop_frames(0)[0])
    d.close()
a, b = out[-1], out[1]
print("symbols", len(a[0]), len(b[0]), "chunks", a[1].shape, b[1].shape, "frames", len(a[2]), len(b[2]))
m = min(len(a[0]), len(b[0]))
rel = np.abs(a[0][:m] - b[0][:m]) / (np.abs(a[0][:m]).mean() + 1e-300)
bad = np.nonzero(rel > 1e-12)[0]
print("max rel soft diff", rel.max(), "first > 1e-12 at", bad[:5], "of", m)
for i in list(bad[:3]):
    print(i, a[0][i], b[0][i])
k = min(len(a[1]), len(b[1]))
print("chunk logs (fo, tf, mu, leftover, nsym), first 4:")
for i in range(min(k, 4)):
    print(" ", a[1][i], "\n ", b[1][i])
print("frames equal:", [bytes(x) == bytes(y) for x, y in zip(a[2], b[2])])
