"""dev: device-memory leak probe of the C ABI (contexts, transmit-chain scratch, decoder scratch, RCCL communicator + gather)."""
import sys, numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from __graft_entry__ import load_opv_amd
amd = load_opv_amd()
dev = torch.device('cuda', 0)
fr = amd.bert_frames(50)
n = amd.lib().opv_tx_modulated_samples(50)
out = torch.empty(2 * n, dtype=torch.int16, device=dev)
torch.cuda.synchronize()
def free(): return torch.cuda.mem_get_info()[0]
f0 = None
for it in range(60):
    d = amd.Demod(4, max_samples=n + 64, streaming=True)
    d.modulate_device(fr, out.data_ptr())
    for k in range(4):
        d.attach(k, out.data_ptr(), n, eof=True)
    d.process(); d.sync()
    d.decode_payloads(np.random.default_rng(it).normal(size=(3, 2144)))
    comm, uid = amd.comm_create(1, 0)
    fv = torch.empty((1, 4, d.device_frames()[3], 134), dtype=torch.uint8, device=dev); cv = torch.empty((1, 4), dtype=torch.int32, device=dev)
    d.gather_frames(comm, 0, fv.data_ptr(), cv.data_ptr()); d.sync()
    amd.comm_destroy(comm)
    assert int(cv.sum()) == 4 * 50, cv
    d.close()
    del fv, cv
    torch.cuda.synchronize()
    if it == 9: f0 = free()
print("free HBM after 10 cycles:", f0, "after 60:", free(), "diff MB:", (f0 - free()) / 1e6)
