"""dev: the device transmit chain on runs around and beyond the end of the build-time NCO checkpoint table (4096 frames)
against the host modulator. usage: tx_long_run.py N [N ...]"""
import sys, time, numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from __graft_entry__ import load_opv_amd
amd = load_opv_amd()
for N in [int(a) for a in sys.argv[1:]] or [4200]:
    fr = np.random.default_rng(3).integers(0, 256, (N, 134), dtype=np.uint8)
    t0 = time.time(); host = amd.modulate(fr); t1 = time.time()
    d = amd.Demod(1, max_samples=1 << 16)
    n = amd.lib().opv_tx_modulated_samples(N)
    out = torch.empty(2 * n, dtype=torch.int16, device="cuda")
    try:
        t2 = time.time(); patched = d.modulate_device(fr, out.data_ptr()); torch.cuda.synchronize(); t3 = time.time()
    except Exception as e:
        print("frames", N, "ERROR", e, flush=True)
        patched = None; t3 = t2 = 0
    dev = out.cpu().numpy()
    eq = bool(np.array_equal(dev, host))
    print("frames", N, "host s", round(t1 - t0, 2), "device s", round(t3 - t2, 3), "patched", patched, "equal", eq, flush=True)
    if not eq:
        bad = np.nonzero(dev != host)[0]
        print("  first diff at int16 index", int(bad[0]), "= frame", int(bad[0] // (2 * 2168 * 40)), "symbol", int(bad[0] // 80), "count", int(bad.size), flush=True)
    d.close(); del out
