"""Multi-GPU host logic: independent IQ streams shard contiguously across ranks (one process
per GPU), there is no data-path collective; decoded frames return to rank 0 with ONE gather
(RCCL over xGMI on GPUs: backend "nccl"; the same code runs on CPU tensors with "gloo").

Layout at the root: [world][streams_per_rank][frame_capacity][134] uint8 + [world][streams_per_rank]
int32 counts, i.e. global stream order g = rank * streams_per_rank + local index
(SURVEY.md §8e)."""
import torch
import torch.distributed as dist


def stream_range(rank, world, total_streams):
    """Contiguous shard of the global stream list owned by `rank` (64/GPU for 512 on 8)."""
    if total_streams % world:
        raise ValueError("total_streams must be a multiple of world size")
    per = total_streams // world
    return range(rank * per, (rank + 1) * per)


def gather_frames(frames, counts, dst=0):
    """frames: [S, cap, 134] uint8, counts: [S] int32 (same shapes on every rank).
    Returns (frames_all [world, S, cap, 134], counts_all [world, S]) on dst, (None, None) elsewhere."""
    if not dist.is_initialized():
        return frames.unsqueeze(0), counts.unsqueeze(0)
    world = dist.get_world_size()      # (a one-rank group still goes through the collective: bench.py's OPV_BENCH_FORCE_DIST self-test)
    rank = dist.get_rank()
    if dist.get_backend() == "gloo" and frames.is_cuda:
        # gloo gathers host tensors only (the world-size-2-on-one-GPU test; RCCL refuses two ranks on one device)
        frames, counts = frames.cpu(), counts.cpu()
    fl = [torch.empty_like(frames) for _ in range(world)] if rank == dst else None
    cl = [torch.empty_like(counts) for _ in range(world)] if rank == dst else None
    dist.gather(frames, fl, dst=dst)
    dist.gather(counts, cl, dst=dst)
    if rank != dst:
        return None, None
    return torch.stack(fl), torch.stack(cl)


def flatten_global(frames_all, counts_all):
    """Root-side helper: list of per-global-stream byte arrays in global stream order."""
    out = []
    W, S = counts_all.shape
    for r in range(W):
        for s in range(S):
            n = int(counts_all[r, s])
            out.append(frames_all[r, s, :n].contiguous())
    return out
