"""GPU tests of BASELINE configs[4]'s shape: the REAL pipeline under world_size > 1, and a 512-stream context.

RCCL refuses two ranks on one device, so on a one-GPU box the two ranks of the first test share cuda:0 and
rendezvous over gloo (sharding.gather_frames moves the library's device views through host memory for gloo);
on the 8-GPU node the same code runs under "nccl" (bench.py). Every GLOBAL stream's frames, Viterbi metrics and
sync positions are checked against the oracle run on that stream's own bytes."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

pytestmark = pytest.mark.gpu


def _free_port():
    """an OS-assigned free TCP port for a rendezvous on 127.0.0.1 (as conftest.run_rccl_selftest picks its own)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _n_gpus():
    """GPUs visible to this job, without initialising one (torch.cuda.device_count() does not, on this image)"""
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


NEEDS_2_GPUS = pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 GPUs: RCCL refuses two ranks on one device (armed for an N-GPU box; "
                                                        "the 1-GPU pool runs the same code over gloo / at world 1)")


def _rank_main(rank, world, port, per_rank, n_frames, ebn0, q, backend="gloo"):
    """one rank of the real pipeline. backend "gloo": every rank on cuda:0, host-side collectives (the one-GPU rehearsal);
    backend "nccl": rank r on cuda:r, RCCL collectives on device tensors - what bench.py does on an N-GPU node."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_opv_amd, load_pkg_module
    from oracle_lib import Oracle
    dev_index = rank if backend == "nccl" else 0
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = (lambda t: t.to(dev)) if backend == "nccl" else (lambda t: t)      # RCCL moves device tensors only
    amd, sharding, workload = load_opv_amd(), load_pkg_module("sharding"), load_pkg_module("workload")
    mine = sharding.stream_range(rank, world, world * per_rank)
    n = amd.lib().opv_tx_modulated_samples(n_frames)
    dm = amd.Demod(per_rank, max_samples=n + 64, streaming=True, device=dev_index)
    d_iq, tx, n = workload.generate(amd, dm, torch, dev, mine, n_frames, ebn0)
    for k in range(per_rank):
        dm.attach(k, d_iq[k].data_ptr(), n, eof=True)
    dm.process()
    dm.sync()
    frames_view, counts_view = workload.frame_views(dm, torch, dev)
    # the checker's side of the same shard: oracle on this rank's own bytes, gathered the same way
    o = Oracle()
    cap = frames_view.shape[1]
    exp_frames = torch.zeros((per_rank, cap, 134), dtype=torch.uint8)
    exp_counts = torch.zeros((per_rank,), dtype=torch.int32)
    exp_meta = torch.zeros((per_rank, cap, 2), dtype=torch.int64)
    got_meta = torch.zeros((per_rank, cap, 2), dtype=torch.int64)
    host_iq = d_iq.cpu().numpy()
    for k in range(per_rank):
        e = o.receive(host_iq[k], streaming=True, want_soft=False)
        nf = len(e["frames"])
        exp_counts[k] = nf
        exp_frames[k, :nf] = torch.from_numpy(e["frames"])
        exp_meta[k, :nf, 0] = torch.from_numpy(e["metrics"].astype(np.int64))
        exp_meta[k, :nf, 1] = torch.from_numpy(e["frame_sym"].astype(np.int64))
        fr, meta = dm.pop_frames(k)
        got_meta[k, :len(fr), 0] = torch.from_numpy(meta["viterbi_metric"].astype(np.int64))
        got_meta[k, :len(fr), 1] = torch.from_numpy(meta["release_symbol"].astype(np.int64))
    fa, ca = sharding.gather_frames(frames_view, counts_view, dst=0)          # the product's gather (device views)
    ea, eca = sharding.gather_frames(comm(exp_frames), comm(exp_counts), dst=0)
    got_meta, exp_meta = comm(got_meta), comm(exp_meta)
    ml = [torch.empty_like(got_meta) for _ in range(world)] if rank == 0 else None
    el = [torch.empty_like(exp_meta) for _ in range(world)] if rank == 0 else None
    dist.gather(got_meta, ml, dst=0)
    dist.gather(exp_meta, el, dst=0)
    if rank == 0:
        if backend == "nccl":
            assert fa.is_cuda and fa.device == dev              # the gather landed in rank 0's HBM, not in host memory
        got, exp = sharding.flatten_global(fa, ca), sharding.flatten_global(ea, eca)
        ok = len(got) == world * per_rank and bool((ca.cpu() == eca.cpu()).all())
        n_frames_total = 0
        for g in range(world * per_rank):
            ok &= bool(torch.equal(got[g].cpu(), exp[g].cpu()))
            n_frames_total += len(got[g])
        ok &= bool(torch.equal(torch.stack(ml).cpu(), torch.stack(el).cpu()))
        # and who sent what: the Base-40 callsign of every global stream's first frame is S<g> (workload.stream_params)
        ok &= [amd.callsign_of(got[g][0].cpu().numpy()) for g in range(world * per_rank)] == [f"S{g}" for g in range(world * per_rank)]
        q.put((bool(ok), n_frames_total))
    dm.close()
    dist.barrier()
    dist.destroy_process_group()


def test_world2_real_pipeline_every_global_stream_vs_oracle():
    """two processes, each opv_process on its contiguous shard of 8 streams (16 global streams x 12 frames,
    16 dB, f0 -2000..-1048 Hz), frames gathered to rank 0 by sharding.gather_frames"""
    ok, nfr = _run_ranks(2, "gloo")
    assert ok is True
    assert nfr >= 16 * 11, nfr


def _run_ranks(world, backend, per_rank=8, n_frames=12):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")                          # fresh interpreters: nothing of this process's GPU state is inherited
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, per_rank, n_frames, 16.0, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    return q.get(timeout=10)


def test_real_pipeline_rank_code_under_nccl_world_1():
    """the rank function of the N > 1 RCCL test below, run where it can run on one GPU: backend "nccl" with ONE rank
    (device_id init, device-tensor gathers of the oracle's side, the landing-in-HBM assertion) - so that the armed test is
    not the first execution of any of its lines except the second rank."""
    ok, nfr = _run_ranks(1, "nccl")
    assert ok is True and nfr >= 8 * 11, nfr


@NEEDS_2_GPUS
@pytest.mark.parametrize("world", [2, 4])          # (ranks + this process stay within the pool's six GPU processes per job)
def test_worldN_real_pipeline_over_rccl_every_global_stream_vs_oracle(world):
    """ARMED FOR AN N-GPU BOX (skipped on the 1-GPU pool): BASELINE configs[4] as north_star words it - one process per GPU,
    rank r on cuda:r, init_process_group("nccl") = RCCL over xGMI, each rank's opv_process on its contiguous shard of 8
    streams, ONE gather of the decoded frames to rank 0 - and every GLOBAL stream's frames, Viterbi metrics and sync
    positions against the oracle run on that stream's own bytes (streams are independent: ref src/opv-demod.cpp:999-1001)."""
    if _n_gpus() < world:
        pytest.skip(f"{_n_gpus()} GPUs visible, {world} needed")
    ok, nfr = _run_ranks(world, "nccl")
    assert ok is True and nfr >= world * 8 * 11, nfr


@NEEDS_2_GPUS
def test_c_side_gather_frames_all_over_two_devices():
    """ARMED FOR AN N-GPU BOX: the C ABI's one-process-many-GPUs form (include/opv_demod.h: opv_comm_init_all +
    opv_gather_frames_all, the N ranks' ncclGathers issued by one thread inside one RCCL group): a context per device, each on
    its own shard of 6 streams, frames and counts gathered into device 0's HBM; the gathered [N][S][cap][134] / [N][S] equal
    every context's own buffers and every global stream's frames the oracle's."""
    _gather_all_in_one_process(min(_n_gpus(), 4))


def test_c_side_gather_frames_all_world_1():
    """the body of the armed test above on the one device of the pool (opv_comm_init_all / opv_gather_frames_all with n = 1), so
    that on an N-GPU box only the second device is new to it"""
    _gather_all_in_one_process(1)


def _gather_all_in_one_process(N):
    import torch
    from __graft_entry__ import load_opv_amd, load_pkg_module
    from oracle_lib import Oracle
    amd, workload = load_opv_amd(), load_pkg_module("workload")
    S, F = 6, 5
    n = amd.lib().opv_tx_modulated_samples(F)
    dms, iqs, views = [], [], []
    for r in range(N):
        dev = torch.device("cuda", r)
        with torch.cuda.device(dev):
            dm = amd.Demod(S, max_samples=n + 64, streaming=True, device=r)
            d_iq, tx, n = workload.generate(amd, dm, torch, dev, range(r * S, (r + 1) * S), F, 16.0)
            for k in range(S):
                dm.attach(k, d_iq[k].data_ptr(), n, eof=True)
            dm.process()
            dms.append(dm)
            iqs.append(d_iq)
            views.append(workload.frame_views(dm, torch, dev))
    cap = views[0][0].shape[1]
    dev0 = torch.device("cuda", 0)
    fa = torch.full((N, S, cap, 134), 0xEE, dtype=torch.uint8, device=dev0)
    ca = torch.full((N, S), -7, dtype=torch.int32, device=dev0)
    torch.cuda.synchronize(dev0)
    comms = amd.comm_init_all(list(range(N)))
    amd.gather_frames_all(dms, comms, 0, fa.data_ptr(), ca.data_ptr())
    for dm in dms:
        dm.sync()
    o = Oracle()
    for r in range(N):
        fv, cv = views[r]
        assert bool(torch.equal(fa[r].cpu(), fv.cpu())) and bool(torch.equal(ca[r].cpu(), cv.cpu())), r
        host = iqs[r].cpu().numpy()
        for k in range(S):
            e = o.receive(host[k], streaming=True, want_soft=False)
            assert int(ca[r, k]) == len(e["frames"]) and np.array_equal(fa[r, k, :len(e["frames"])].cpu().numpy(), e["frames"]), (r, k)
            assert amd.callsign_of(e["frames"][0]) == f"S{r * S + k}"
    for c in comms:
        amd.comm_destroy(c)
    for dm in dms:
        dm.close()


def test_bench_rccl_leg_executes_at_world_1(request):
    """configs[4]'s collective leg on hardware: `OPV_BENCH_FORCE_DIST=1 bench.py --gpus 1` (a child process started by
    conftest.pytest_sessionstart before this process touched the GPU) runs bench.py's N > 1 path over RCCL with one rank:
    nccl process group bound to the device, gather of the [S, cap, 134] frame buffer + counts from the library's zero-copy
    views, MAX all-reduce of the step time. The gathered tensor must equal the local view (asserted inside bench.py) and
    every gathered frame is compared with what was sent. The log goes to gpurun_out/ (copied to profiles/ by hand)."""
    import json
    from conftest import run_rccl_selftest
    r = getattr(request.config, "_opv_rccl_selftest", None) or run_rccl_selftest()
    both = r["stdout"].splitlines() + r["stderr"].splitlines()          # (RCCL's NCCL_DEBUG lines go to stdout)
    nccl_lines = [ln for ln in both if ("NCCL" in ln or "RCCL" in ln or "HIP version" in ln or "ROCm version" in ln) and "alt_rsmi" not in ln and " Channel " not in ln]
    bench_lines = [ln for ln in r["stdout"].splitlines() if ln.startswith('{"metric"')]
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "rccl_selftest.txt").write_text(
        f"$ {r['cmd']}\nexit {r['rc']} after {r['seconds']:.1f} s\n--- the bench line\n" + "\n".join(bench_lines) + "\n"
        f"--- {len(nccl_lines)} NCCL/RCCL lines of the child (first 80)\n" + "\n".join(nccl_lines[:80]) + "\n"
        + ("--- stderr tail\n" + r["stderr"][-3000:] if r["rc"] else ""))
    assert r["rc"] == 0, r["stderr"][-3000:]
    assert len(bench_lines) == 1, r["stdout"][-2000:]
    line = json.loads(bench_lines[0])
    assert line["n_gpus"] == 1 and line["collective"]["backend"] == "nccl" and line["collective"]["world"] == 1
    assert line["collective"]["gathered_shape"] == [1, 8, line["collective"]["gathered_shape"][2], 134]
    chk = line["check"]
    assert chk["gathered_frames_total"] == 8 * 12
    assert chk["gathered_frames_exact"] >= 8 * 12 - 1          # 16 dB: at most a stray channel error
    assert chk["gathered_equals_local_view"] is True
    assert chk["edge_ties"] == 0 and chk["offset_ties"] == 0


def _nccl_world1_main(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_opv_amd, load_pkg_module
    from oracle_lib import Oracle
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    amd, sharding, workload = load_opv_amd(), load_pkg_module("sharding"), load_pkg_module("workload")
    S, F = 8, 6
    n = amd.lib().opv_tx_modulated_samples(F)
    dm = amd.Demod(S, max_samples=n + 64, streaming=True, device=0)
    d_iq, tx, n = workload.generate(amd, dm, torch, dev, range(S), F, 16.0)
    for k in range(S):
        dm.attach(k, d_iq[k].data_ptr(), n, eof=True)
    dm.process()
    dm.sync()
    fv, cv = workload.frame_views(dm, torch, dev)           # zero-copy views of library memory
    fa, ca = sharding.gather_frames(fv, cv, dst=0)          # RCCL gather, one rank
    torch.cuda.synchronize()
    ok = fa.is_cuda and tuple(fa.shape) == (1,) + tuple(fv.shape) and bool(torch.equal(fa[0], fv)) and bool(torch.equal(ca[0], cv))
    ok &= fa.data_ptr() != fv.data_ptr()                    # a gathered copy, not the view handed back
    o = Oracle()
    host = d_iq.cpu().numpy()
    flat = sharding.flatten_global(fa, ca)
    for k in range(S):
        e = o.receive(host[k], streaming=True, want_soft=False)
        ok &= bool(np.array_equal(flat[k].cpu().numpy(), e["frames"]))
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok &= float(t.item()) == 1.25
    q.put((bool(ok), dist.get_backend(), int(ca.sum())))
    dm.close()
    dist.barrier()
    dist.destroy_process_group()


def test_gather_frames_under_nccl_world_1():
    """sharding.gather_frames under init_process_group("nccl", world_size=1) on the library's zero-copy DevPtr views
    of a real 8-stream run: the gathered [1, 8, cap, 134] tensor equals the view, and every stream's frames equal the
    oracle's. (world > 1 on one device is refused by RCCL: that shape runs over gloo in the test above.)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_main, args=(_free_port(), q))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    ok, backend, nfr = q.get(timeout=10)
    assert ok is True and backend == "nccl" and nfr >= 8 * 5


def test_c_side_gather_frames_world_1():
    """include/opv_demod.h: opv_comm_unique_id + opv_comm_init + opv_gather_frames - the C ABI's own RCCL gather (ncclGather on
    the context's stream, RCCL bound by dlopen: here PyTorch's copy, already in the process) at world 1 on a real 6-stream
    run: the gathered [1][S][cap][134] frames and [1][S] counts equal the library's buffers, and every stream's frames the
    oracle's. (The same call from a stand-alone C++ process: test_rx_bridge_shards_over_contexts_and_gathers_in_cxx.)"""
    import torch
    from __graft_entry__ import load_opv_amd, load_pkg_module
    from oracle_lib import Oracle
    amd, workload = load_opv_amd(), load_pkg_module("workload")
    dev = torch.device("cuda", 0)
    S, F = 6, 5
    n = amd.lib().opv_tx_modulated_samples(F)
    dm = amd.Demod(S, max_samples=n + 64, streaming=True)
    d_iq, tx, n = workload.generate(amd, dm, torch, dev, range(100, 100 + S), F, 16.0)
    for k in range(S):
        dm.attach(k, d_iq[k].data_ptr(), n, eof=True)
    dm.process()
    fv, cv = workload.frame_views(dm, torch, dev)
    comm, uid = amd.comm_create(1, 0, device=0)
    assert len(uid) == 128
    fa = torch.full((1,) + tuple(fv.shape), 0xEE, dtype=torch.uint8, device=dev)
    ca = torch.full((1, S), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    dm.gather_frames(comm, 0, fa.data_ptr(), ca.data_ptr())        # asynchronous, behind the kernels on the context's stream
    dm.sync()
    assert bool(torch.equal(fa[0], fv)) and bool(torch.equal(ca[0], cv))
    o = Oracle()
    host = d_iq.cpu().numpy()
    for k in range(S):
        e = o.receive(host[k], streaming=True, want_soft=False)
        assert int(ca[0, k]) == len(e["frames"]) and np.array_equal(fa[0, k, :len(e["frames"])].cpu().numpy(), e["frames"]), k
    amd.comm_destroy(comm)
    dm.close()


def test_512_stream_context_vs_oracle():
    """configs[4]'s stream count in one context: 512 streams (8 shards of 64, global ids 0..511) x 3 frames,
    Eb/N0 16 dB, every stream against the oracle (frames, metrics, sync positions, tracker lines, offset estimate)."""
    import torch
    from __graft_entry__ import load_opv_amd, load_pkg_module
    from oracle_lib import Oracle
    from test_gpu_parity import events_match, no_ties
    amd, workload = load_opv_amd(), load_pkg_module("workload")
    dev = torch.device("cuda", 0)
    S, F = 512, 3
    n = amd.lib().opv_tx_modulated_samples(F)
    dm = amd.Demod(S, max_samples=n + 64, streaming=True)
    d_iq, tx, n = workload.generate(amd, dm, torch, dev, range(S), F, 16.0)
    for k in range(S):
        dm.attach(k, d_iq[k].data_ptr(), n, eof=True)
    dm.process()
    dm.sync()
    host = d_iq.cpu().numpy()
    from concurrent.futures import ProcessPoolExecutor
    from soak_inputs import host_workers, oracle_receive_job
    total = guarded = 0
    with ProcessPoolExecutor(host_workers()) as pool:
        exps = pool.map(oracle_receive_job, [host[k] for k in range(S)], chunksize=8)
        for k, e in enumerate(exps):
            fr, meta = dm.pop_frames(k)
            assert np.array_equal(fr, e["frames"]), k
            assert np.array_equal(meta["viterbi_metric"], e["metrics"]) and np.array_equal(meta["release_symbol"], e["frame_sym"]), k
            events_match(amd, dm.pop_events(k), e["events"])
            assert dm.state(k).est_offset_hz == e["est_offset"], k
            no_ties(dm.state(k), f"stream {k} of 512", offset_ties=None)
            guarded += dm.state(k).offset_ties > 0
            total += len(fr)
    assert total >= S * (F - 1)
    print(f"offset-search near-tie guard fired on {guarded} of {S} streams (estimates equal the oracle's on all)")
    assert guarded <= 4
    dm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("S,forced", [(522, 0), (2060, 0), (8200, 0), (16400, 0), (8200, 4), (2060, 16), (600, 4)])
def test_many_stream_contexts_on_the_automatic_mapping(S, forced):
    """Every launch shape opv_process has for many streams, each against the oracle - decisions on EVERY stream, all soft symbols
    (< 1e-9 of their mean) and the tracker's events on eight streams spread over the first, a middle and the last (partly filled)
    workgroup. forced != 0: opv_set_frontend(forced) where the automatic choice would be another kernel - four per wave beyond
    8192 streams (more 16-stream workgroups than the chip holds at once: the rest queue) and below 2049, sixteen per wave below 8193.
    16 400 streams: beyond 16 384 the sixteen-per-wave kernel runs EIGHT waves (128 streams, 8 rings + the table = 151 600 B of
    LDS) per workgroup (k_msk_frontend_x16_wg8: 128 full workgroups + one with 16 streams, i.e. one busy wave and seven that have
    nothing to do). 8200 streams: beyond 8192 the shim takes sixteen streams per wave by itself (k_msk_frontend_x16_wg4: 128 full workgroups
    of 64 streams + one with 8, i.e. a wave with eight idle quads and three waves that have nothing to do). 522 streams: one wave per stream, FOUR waves per workgroup (k_msk_frontend_rb_wg4, from 513 streams; 130 full
    workgroups + a partly filled one). 2060 streams: more than the one-wave-per-stream kernel holds in two rounds, the
    shim takes the four-per-wave mapping by itself (from 2049 streams; 128 full workgroups + a partly filled one, and
    the last wave has idle rows). 24 distinct captures (16 dB, different offsets / payloads) shared by the streams,
    every stream against the oracle's result for its capture."""
    import torch
    from __graft_entry__ import load_opv_amd, load_pkg_module
    from oracle_lib import Oracle
    amd, workload = load_opv_amd(), load_pkg_module("workload")
    dev = torch.device("cuda", 0)
    D, F = 24, 3
    n = amd.lib().opv_tx_modulated_samples(F)
    from test_gpu_parity import events_match, soft_err
    dm = amd.Demod(S, max_samples=n + 64, streaming=True)
    if forced:
        dm.set_frontend(forced)
    d_iq, tx, n = workload.generate(amd, dm, torch, dev, range(D), F, 16.0)
    for k in range(S):
        dm.attach(k, d_iq[k % D].data_ptr(), n, eof=True)
    dm.process()
    dm.sync()
    want = {(522, 0): "k_msk_frontend_rb_wg4", (2060, 0): "k_msk_frontend_x4_wg4", (8200, 0): "k_msk_frontend_x16_wg4", (16400, 0): "k_msk_frontend_x16_wg8",
            (8200, 4): "k_msk_frontend_x4_wg4", (2060, 16): "k_msk_frontend_x16_wg4", (600, 4): "k_msk_frontend_x4_wg4"}[(S, forced)]
    assert dm.frontend_kernel() == want or (not forced and os.environ.get("OPV_FRONTEND"))
    host = d_iq.cpu().numpy()
    o = Oracle()
    exp = [o.receive(host[j], streaming=True, want_soft=False) for j in range(D)]
    # first workgroup, a middle one, the last (partly filled) one: all soft symbols and the tracker's lines
    picks = [0, 1, 2, S // 2, S // 2 + 1, S - 3, S - 2, S - 1]
    worst = 0.0
    for k in picks:
        e = o.receive(host[k % D], streaming=True)
        a, _ = soft_err(dm.soft(k), e["soft"])
        worst = max(worst, a)
        assert a < 1e-9, (k, a)
        events_match(amd, dm.pop_events(k), e["events"])
    print(f"{want} at {S} streams: soft max|d|/mean|soft| over streams {picks} = {worst:.2e}")
    for k in range(S):
        e = exp[k % D]
        fr, meta = dm.pop_frames(k)
        assert np.array_equal(fr, e["frames"]), k
        assert np.array_equal(meta["viterbi_metric"], e["metrics"]) and np.array_equal(meta["release_symbol"], e["frame_sym"]), k
        st = dm.state(k)
        assert st.est_offset_hz == e["est_offset"] and st.total_symbols == e["n_soft"], k
        assert st.edge_ties == 0, k
    dm.close()
