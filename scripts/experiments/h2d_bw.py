"""dev: pinned host -> device copy bandwidth for a few sizes (what opv_push_iq can reach)."""
import time, torch
for mb in (0.35, 3.5, 35, 350):
    n = int(mb * 1e6)
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    reps = max(3, int(2e9 / n))
    t0 = time.perf_counter()
    for _ in range(reps):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mb} MB x {reps}: {n * reps / dt / 1e9:.1f} GB/s")
