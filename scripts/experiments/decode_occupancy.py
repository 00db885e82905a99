import sys, time
sys.path.insert(0, 'tests')
import numpy as np
from amd_lib import load
amd = load()
d = amd.Demod(1, max_samples=1 << 20)
print("occupancy [rb, rb_wg4, x16_wg4, x4_wg4, frame_decode, frame_scale]:", d.occupancy())
rng = np.random.default_rng(3)
base = rng.standard_normal((256, 2144)) * 2.4e11
import ctypes as C
L = amd.lib()
for N in (256, 1024, 2048, 4096, 8192, 16384, 32768):
    soft = np.tile(base, (N // 256, 1))
    d.decode_payloads(soft)
    t0 = time.perf_counter(); d.decode_payloads(soft); dt = time.perf_counter() - t0
    print(N, "payloads", round(dt * 1e3, 3), "ms end to end")
d.close()
