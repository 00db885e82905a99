"""dev: contexts with very many streams (beyond the 8192 of the bench sweep): every stream gets the same clean F-frame capture
(attached, zero-copy), all must release F frames equal to the transmitted ones. usage: many_streams.py S F [S F ...]"""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from __graft_entry__ import load_opv_amd
amd = load_opv_amd()
a = [int(v) for v in sys.argv[1:]] or [20000, 3]
for S, F in zip(a[0::2], a[1::2]):
    tx = amd.bert_frames(F)
    iq = amd.modulate(tx)
    n = iq.size // 2
    d_iq = torch.from_numpy(iq).cuda()
    d = amd.Demod(S, max_samples=n + 64, streaming=True)
    for s in range(S):
        d.attach(s, d_iq.data_ptr(), n, eof=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d.process(); d.sync()
    dt = time.perf_counter() - t0
    bad = 0
    for s in range(S):
        fr, _ = d.pop_frames(s)
        bad += not (len(fr) == F and np.array_equal(fr, tx))
    print(f"S={S} F={F}: kernel {d.frontend_kernel()}, {S * n / dt / 1e6:.0f} Msamples/s ({dt * 1e3:.1f} ms), streams with wrong output: {bad}", flush=True)
    d.close()
