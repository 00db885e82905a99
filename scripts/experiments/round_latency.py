"""dev: latency of one live round (push one 86720-sample chunk per stream -> process -> pop) for S streams."""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from __graft_entry__ import load_opv_amd
amd = load_opv_amd()
S = int(sys.argv[1]); R = int(sys.argv[2]) if len(sys.argv) > 2 else 30
iq = amd.modulate(amd.bert_frames(R + 2))
pinned = torch.from_numpy(iq).pin_memory().numpy()
d = amd.Demod(S, max_samples=4 * 86720 + 65536, streaming=True)
CH = 86720
lat, t_push, t_proc, t_pop = [], [], [], []
nfr = 0
for r in range(R):
    blk = pinned[2 * r * CH: 2 * (r + 1) * CH]
    t0 = time.perf_counter()
    d.push_batch(range(S), [blk] * S)
    t1 = time.perf_counter()
    d.process(); d.sync()
    t2 = time.perf_counter()
    for s in range(S):
        fr, _ = d.pop_frames(s)
        nfr += len(fr)
    t3 = time.perf_counter()
    lat.append(t3 - t0); t_push.append(t1 - t0); t_proc.append(t2 - t1); t_pop.append(t3 - t2)
m = lambda a: 1e3 * float(np.median(a[3:]))
mean = lambda a: 1e3 * float(np.mean(a[3:]))
print(f"S={S}: mean round {mean(lat):.3f} ms (push {mean(t_push):.3f}, process {mean(t_proc):.3f}, pop {mean(t_pop):.3f}); max round {1e3*max(lat[3:]):.3f}")
print(f"S={S}: round {m(lat):.3f} ms = push {m(t_push):.3f} + process+sync {m(t_proc):.3f} + pop {m(t_pop):.3f}; frames {nfr}; chunk = 40 ms of signal per stream")
