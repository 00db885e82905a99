"""CPU-only, world_size 2, gloo: the N>1 host path of bench.py — contiguous stream sharding
and the single gather of decoded frames to rank 0 (RCCL on GPUs, gloo here)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))


def _worker(rank, world, port, total_streams, cap, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    spec = importlib.util.spec_from_file_location("sharding", ROOT / "opv-cxx-demod_amd" / "sharding.py")
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    from oracle_lib import Oracle
    o = Oracle()
    mine = sh.stream_range(rank, world, total_streams)
    S = len(mine)
    frames = torch.zeros((S, cap, 134), dtype=torch.uint8)
    counts = torch.zeros((S,), dtype=torch.int32)
    for i, g in enumerate(mine):                       # stream g "decodes" g%3+1 frames of callsign S<g>
        n = g % 3 + 1
        frames[i, :n] = torch.from_numpy(o.bert_frames(n, f"S{g}", 0xBBAADD, 10 * g))
        counts[i] = n
    fa, ca = sh.gather_frames(frames, counts, dst=0)
    if rank == 0:
        flat = sh.flatten_global(fa, ca)
        ok = len(flat) == total_streams
        for g, f in enumerate(flat):
            ok &= np.array_equal(f.numpy(), o.bert_frames(g % 3 + 1, f"S{g}", 0xBBAADD, 10 * g))
        q.put(bool(ok))
    else:
        assert fa is None and ca is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:                        # an OS-assigned free port (a pid-derived one collides between concurrent jobs)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 8, 4, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_stream_range_contiguous():
    import importlib.util
    spec = importlib.util.spec_from_file_location("sharding", ROOT / "opv-cxx-demod_amd" / "sharding.py")
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    assert [list(sh.stream_range(r, 8, 512))[0] for r in range(8)] == [64 * r for r in range(8)]
    assert len(sh.stream_range(3, 8, 512)) == 64
    with pytest.raises(ValueError):
        sh.stream_range(0, 3, 64)
