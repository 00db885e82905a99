#!/usr/bin/env python3
"""bench.py — throughput of the MI355X-native OPV MSK receive chain (demod + Viterbi).

Contract (driver):  python bench.py --gpus N --steps K --warmup W. One rank per GPU: under
torch.distributed.run the ranks come from the launcher (RANK / LOCAL_RANK / WORLD_SIZE); from a bare shell
`python bench.py --gpus N` starts its N ranks itself (child processes, before anything touches a GPU) and
relays rank 0's line. Rank 0 prints ONE JSON line.

Workload. A "step" is one pass of the whole hot path (offset search -> MSK front-end -> sync
tracker -> frame decode) over one batch of synthetic captures that are ALREADY RESIDENT IN HBM.
Per GPU the batch is BASELINE.json configs[3]: 64 independent IQ streams x 1000 frames
(86 724 000 samples each, 2.168 MSPS), `-s` semantics; with N GPUs that is configs[4] shape
(64 streams per GPU, weak scaling), decoded frames gathered to rank 0 with one RCCL gather.
Stream k is its own BERT capture generated in HBM by the device modulator (bit-identical to what
`opv-mod -S S<k> -B 1000` would emit) and passed through the device channel tool: amplitude 2000, carrier offset
f0_k = -2000 + 4000 k/63 Hz (the edge streams sit on the AFC clamp, ref src/opv-demod.cpp:303, outside the +/-1530 Hz
span of the offset search, :135,169), AWGN at Eb/N0 = 16 dB - SURVEY.md §8d C4 as written. The decode is checked:
every stream releases its 1000 frames and >= 99 % of the 64 x 1000 equal the transmitted ones (a full-size
encode -> channel -> decode round trip). `value` is units / the bracketed time of the K steps (the driver's contract);
the median step ("median of >= 5") is reported next to it, and extras.all_clean_variant is C4's
"all-clean variant": the same 64 x 1000 captures straight from the modulator.

Also reported (same run, outside the timed steps): the reference `opv-demod -s -r -q` itself timed on the host
(cpu_baseline.kind = "reference") when the prebuilt oracle/_ref binary travelled with the snapshot, else the C oracle
("port") - always - and the EXTRAS: configs[1] (ONE clean 1000-frame stream), the all-clean variant, a many-short-streams
sweep that shows the throughput-bound regime, configs[4]'s whole workload on one GPU, 32 768 unique streams, the
PCIe-inclusive rates, the live-serving capacity. The extras share ONE wall-clock budget (--extras-budget, default 75 s): they
run in a fixed order of priority, each only if its estimated cost still fits, and extras.skipped_for_budget names the ones
that did not (profiles/collect.sh records the line with a budget that holds them all).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FRAME_SAMPLES = 86720
ALGO_BYTES_PER_SAMPLE = 4.0 + 134.0 / FRAME_SAMPLES  # 4 B in + 134 B / frame out
HBM_PEAK_GBS = 8000.0                                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def load_amd():
    from __graft_entry__ import load_opv_amd
    return load_opv_amd()


def load_pkg(name):
    from __graft_entry__ import load_pkg_module
    return load_pkg_module(name)


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as child processes (this parent never
    touches a GPU, and nothing is exec'ed from a process that has), watch them, exit with the worst status. A rank
    that fails takes the others with it: they would otherwise sit in the rendezvous until its timeout."""
    import socket
    with socket.socket() as sk:                        # a free port now; rank 0's store binds it a moment later
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    me = [sys.executable, str(Path(__file__).resolve())] + argv
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen(me, env=env))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            if p.poll() is None:
                continue
            live.remove(p)
            if p.returncode != 0 and rc == 0:
                rc = abs(p.returncode)
                for q in live:                         # fresh children this parent started itself: safe to end
                    q.terminate()
        time.sleep(0.05)
    sys.exit(rc)


def cpu_baseline(iq_bytes, n_samples):
    """Time the reference CPU opv-demod (or the oracle port) on this box's host cores."""
    ref = ROOT / "oracle" / "_ref" / "opv-demod"
    if ref.exists():
        t0 = time.perf_counter()
        p = subprocess.run([str(ref), "-s", "-r", "-q"], input=iq_bytes, stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        nf = len(p.stdout) // 134
        cpu_baseline.ref_stdout = p.stdout              # (kept for extras.cli_drop_in: our CLI on the same bytes)
        return {"value": n_samples / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "reference",
                "sample": f"oracle/_ref/opv-demod -s -r -q on the clean {n_samples // FRAME_SAMPLES}-frame "
                          f"capture via a pipe ({dt:.2f} s, {nf} frames out)"}
    sys.path.insert(0, str(ROOT / "tests"))       # the oracle binding is test infrastructure; only this leg uses it
    from oracle_lib import Oracle
    o = Oracle()
    iq = np.frombuffer(iq_bytes, np.int16)
    t0 = time.perf_counter()
    r = o.receive(iq, streaming=True, want_soft=False)
    dt = time.perf_counter() - t0
    return {"value": n_samples / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"oracle/opv_oracle.c oro_receive on the clean {n_samples // FRAME_SAMPLES}-frame capture "
                      f"in memory ({dt:.2f} s, {len(r['frames'])} frames out)"}


def cli_drop_in(iq_bytes, n_samples):
    """The drop-in measured the way the reference CPU binary is (cpu_baseline): opv-cxx-demod_amd/bin/opv-demod -s -r -q as
    a child process, the same capture through a pipe - process start, HIP initialisation, stdin reads, a round of kernels per
    chunk and the 134-byte writes included; its stdout against the reference child's."""
    exe = ROOT / "opv-cxx-demod_amd" / "bin" / "opv-demod"
    if not exe.exists():
        return None
    t0 = time.perf_counter()
    p = subprocess.run([str(exe), "-s", "-r", "-q"], input=iq_bytes, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    ref = getattr(cpu_baseline, "ref_stdout", None)
    return {"Msamples/s": round(n_samples / dt / 1e6, 2), "s": round(dt, 2), "exit": p.returncode, "frames_out": len(p.stdout) // 134,
            "stdout_identical_to_reference": (p.stdout == ref) if ref is not None else None,
            "what": "bin/opv-demod -s -r -q (one stream = one wavefront) on the capture the reference binary was timed on, via a pipe"}


class Budget:
    """One wall-clock budget for everything bench.py does beyond the contract's line: an extra runs only if its estimated cost
    still fits; what was skipped and what each one took is part of the line (extras.budget, extras.skipped_for_budget)."""

    def __init__(self, seconds):
        self.seconds = float(seconds)
        self.t_end = time.perf_counter() + self.seconds
        self.skipped, self.spent = [], {}

    def left(self):
        return self.t_end - time.perf_counter()

    def run(self, name, est_s, fn, always=False):
        """fn() if `est_s` still fits (or `always`: the contract's cpu_baseline is not optional, only accounted for);
        an extra that raises costs its own entry, never the bench line"""
        if not always and self.left() < est_s:
            self.skipped.append({"extra": name, "needs_s": est_s, "left_s": round(max(self.left(), 0.0), 1)})
            return None
        t0 = time.perf_counter()
        try:
            return fn()
        except AssertionError:
            raise
        except Exception as e:
            return {"error": repr(e)[:300]}
        finally:
            self.spent[name] = round(time.perf_counter() - t0, 1)

    def report(self):
        return {"seconds": self.seconds, "spent_s": self.spent, "spent_total_s": round(sum(self.spent.values()), 1),
                "order": "cpu_baseline (always), configs[1], all-clean variant, cpu on all cores, stream sweep, configs[4] on one GPU, "
                         "many unique streams, PCIe-inclusive + live round, CLI drop-in, live capacity"}


def live_capacity(dev_index, budget=None):
    """The reference's real caller shape (`opv-modem -R`, src/opv-modem.cpp:673-838: IQ arrives at 2.168 MSPS, 40 ms per frame), for N
    streams at once: how many live streams ONE context serves in real time. bin/opv-live-capacity (host/opv_live_capacity.cpp, a
    C++ caller of the C ABI like opv-rx-bridge) runs 120 serving rounds - one 86 720-sample chunk per stream pushed from pinned host
    memory over PCIe (every stream's chunk at its own host addresses: N x 347 KB of distinct memory per round), opv_process, every
    stream's frames popped and compared with what was sent - and reports the round-time distribution; the capacity is the largest N
    probed whose p99 round stays under the 40 ms of signal a round consumes, to 256 streams - for the serial loop (push, process,
    pop) and for the double-buffered one (opv_push_iq_batch_async: round r + 1 crosses PCIe while round r is processed and
    popped). A round is bound by the PCIe link, so its time is linear in N: every probe's p99 predicts the capacity and the next
    probe goes there (a secant search; three probes for the serial loop, two for the pipelined one, where the doubling + linear
    walk of round 5 took thirteen). Every probe is charged to the extras' budget - 60 rounds each while that is tight, 120 when
    it is not; when it runs out the search ends with the best bracket it has ("cut_short")."""
    exe = ROOT / "opv-cxx-demod_amd" / "bin" / "opv-live-capacity"
    if not exe.exists():
        return None
    probes = {}
    t_end = time.perf_counter() + (budget.left() if budget is not None else 150.0)   # the whole search is bounded by the extras' budget
    cut = []
    rounds = 120 if (budget is None or budget.left() > 150.0) else 60

    def probe(n, pipelined=False):
        key = (n, pipelined)
        if key not in probes:
            left = t_end - time.perf_counter()
            if left < (12.0 if rounds == 120 else 8.0):   # a probe: process start + the rounds of <= 40 ms + the check
                cut.append(n)
                return None
            try:
                p = subprocess.run([str(exe), str(n), str(rounds), "6", str(dev_index)] + (["--pipelined"] if pipelined else []),
                                   capture_output=True, text=True, timeout=min(60.0, left))
                line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                probes[key] = json.loads(line[-1]) if p.returncode == 0 and line else {"streams": n, "pipelined": pipelined, "error": (p.stderr or p.stdout)[-200:], "rc": p.returncode}
            except subprocess.TimeoutExpired:
                probes[key] = {"streams": n, "pipelined": pipelined, "error": "probe timed out", "rc": None}
        r = probes[key]
        return "error" not in r and r["round_ms_p99"] < 40.0 and r["frames_wrong"] == 0

    def secant(start, pipelined, grid=256, most=6):
        """(largest N probed that passes, smallest that fails) with the two `grid` apart, each probe placed where the last one's
        p99 says the 40 ms line is (round time is linear in N)"""
        lo, hi, n = 0, None, start
        for _ in range(most):
            ok = probe(n, pipelined)
            if ok is None:
                break
            if ok:
                lo = max(lo, n)
            else:
                hi = n if hi is None else min(hi, n)
            if hi is not None and hi - lo <= grid:
                break
            p99 = probes[(n, pipelined)].get("round_ms_p99")
            nxt = int(n * 40.0 / p99 * (0.985 if ok else 1.0)) // grid * grid if p99 else (n * 2 if ok else n // 2)
            nxt = max(nxt, lo + grid)
            if hi is not None:
                nxt = min(nxt, hi - grid)
            if nxt <= lo or nxt > 16384:
                break
            n = nxt
        return lo, hi
    lo, hi = secant(4096, False)
    plo, phi = (secant(lo + 768, True) if lo and not cut else (0, None))
    best, pbest = probes.get((lo, False)), probes.get((plo, True))
    return {"streams": lo, "round_ms_p99": best["round_ms_p99"] if best else None, "round_ms_p50": best["round_ms_p50"] if best else None,
            "Msamples/s_sustained": round(lo * 2.168, 1), "first_n_over_40ms": hi,
            "cut_short": {"probes_not_run": cut, "why": "the extras' wall-clock budget ran out; the figures are the best bracket reached"} if cut else None,
            "pipelined": {"streams": plo, "round_ms_p99": pbest["round_ms_p99"] if pbest else None, "round_ms_p50": pbest["round_ms_p50"] if pbest else None,
                          "Msamples/s_sustained": round(plo * 2.168, 1),
                          "what": "the same rounds with opv_push_iq_batch_async: the next round's chunks cross PCIe while this round is processed and popped"},
            "probes": [probes[k] for k in sorted(probes)],
            "rounds_per_probe": rounds,
            "what": "largest probed N with p99 round < 40 ms over the probe's rounds: one 86720-sample chunk per stream from pinned host memory "
                    "(opv_push_iq_batch), opv_process + opv_sync, opv_pop_frames of every stream (bin/opv-live-capacity)"}


def cpu_all_cores(iq_bytes, n_samples):
    """The reference is single-threaded and streams are independent: one reference process per host core this
    job may use, each on its own copy of a bounded sample (200 frames) of the capture."""
    ref = ROOT / "oracle" / "_ref" / "opv-demod"
    if not ref.exists():
        return None
    cores = min(len(os.sched_getaffinity(0)), 16)     # a one-GPU job's CPU share on the pool's boxes is 16 cores
    nfr = min(200, n_samples // FRAME_SAMPLES)
    part = iq_bytes[: nfr * FRAME_SAMPLES * 4]
    t0 = time.perf_counter()
    procs = [subprocess.Popen([str(ref), "-s", "-r", "-q"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL) for _ in range(cores)]
    import threading
    outs = [None] * cores

    def feed(i):
        outs[i] = procs[i].communicate(part)[0]
    th = [threading.Thread(target=feed, args=(i,)) for i in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    model = ""
    try:
        model = next(ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name"))
    except Exception:
        pass
    return {"value": round(cores * nfr * FRAME_SAMPLES / dt / 1e6, 2), "unit": "Msamples/s", "cores": cores, "kind": "reference",
            "cpu": model, "sample": f"{cores} concurrent oracle/_ref/opv-demod -s -r -q, {nfr} frames each ({dt:.2f} s, "
                                    f"{sum(len(o) for o in outs) // 134} frames out)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=64, help="streams per GPU")
    ap.add_argument("--frames", type=int, default=1000, help="frames per stream")
    ap.add_argument("--ebn0", type=float, default=16.0, help="dB; <=0 disables noise")
    ap.add_argument("--no-extras", action="store_true", help="skip configs[1], the sweep and the CPU baseline")
    ap.add_argument("--no-big", action="store_true", help="skip the 512-stream x F-frame single-GPU run of the extras (178 GB of HBM)")
    ap.add_argument("--extras-budget", type=float, default=75.0,
                    help="wall-clock seconds for everything beyond the contract's line (the extras, in a fixed order of priority; "
                         "what does not fit is listed in extras.skipped_for_budget)")
    args = ap.parse_args()

    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if args.steps < 1 or args.warmup < 0 or args.streams < 1 or args.frames < 1:
        print("bench.py: --steps, --streams and --frames must be >= 1, --warmup >= 0", file=sys.stderr)
        sys.exit(2)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus, sys.argv[1:])              # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    # Rehearsal switches for a box with ONE GPU (tests/test_bench_contract.py): OPV_BENCH_SHARE_DEVICE=1 puts every rank
    # on cuda:0 instead of cuda:LOCAL_RANK, and OPV_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one
    # device) for the rendezvous, the frame gather and the MAX of the step time. Everything else - rank spawning, the
    # shard arithmetic, the cross-rank expectation, the N > 1 cpu_baseline leg - is the code the 8-GPU node runs.
    backend = os.environ.get("OPV_BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        print(f"bench.py: OPV_BENCH_BACKEND={backend!r}: nccl (= RCCL) or gloo", file=sys.stderr)
        sys.exit(2)
    dev_index = 0 if os.environ.get("OPV_BENCH_SHARE_DEVICE") == "1" else local_rank
    if dev_index >= torch.cuda.device_count():
        print(f"bench.py: rank {rank}: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} GPU(s) visible "
              f"(one rank per GPU; OPV_BENCH_SHARE_DEVICE=1 + OPV_BENCH_BACKEND=gloo rehearses N > 1 on one)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # OPV_BENCH_FORCE_DIST=1 runs the N>1 code path (RCCL init, gather of the frame buffer, MAX over
    # ranks) even with one rank: a self-test of that path on boxes with a single GPU.
    use_dist = world > 1 or os.environ.get("OPV_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    amd = load_amd()
    amd.lib()
    sharding, workload = load_pkg("sharding"), load_pkg("workload")
    DevPtr = workload.DevPtr
    S, F = args.streams, args.frames

    # ---- synthetic input, generated IN HBM: device modulator (bit-identical to `opv-mod`, see
    # tests) -> device channel tool. Every stream carries its own BERT payload sequence. Rank r owns the
    # contiguous shard [r S, (r + 1) S) of the world * S global streams (sharding.stream_range).
    mine = sharding.stream_range(rank, world, world * S)
    n = amd.lib().opv_tx_modulated_samples(F)
    dm = amd.Demod(S, max_samples=n + 64, streaming=True, device=dev_index)
    gen_t = {}
    d_iq, tx_all, n = workload.generate(amd, dm, torch, dev, mine, F, args.ebn0, timing=gen_t)
    t_mod = gen_t["generate_s"]                          # BERT frames + device transmit chain + channel tool, all streams (allocation excluded)
    tx_frames = amd.bert_frames(F)                       # configs[1] / CPU-baseline capture (W5NYV)

    frames_view, counts_view = workload.frame_views(dm, torch, dev)
    fcap = frames_view.shape[1]
    gathered = {}
    expect = torch.from_numpy(tx_all).to(dev)

    stats = {}

    def step(check=True):
        dm.reset()
        for k in range(S):
            dm.attach(k, d_iq[k].data_ptr(), n, eof=True)
        dm.process()
        dm.sync()
        if use_dist:                                     # RCCL over xGMI: frames + counts back to rank 0
            gathered["frames"], gathered["counts"] = sharding.gather_frames(frames_view, counts_view, dst=0)
        if check:
            # full-size round trip: every stream must release exactly F frames, in order, and
            # (at 16 dB a handful of frames carry residual channel errors) >= 99% of them must
            # equal the transmitted bytes; clean runs (--ebn0 0) must be 100% exact.
            cnt = counts_view.cpu().numpy()
            neq = (frames_view[:, :F, :] != expect).any(dim=2)
            n_bad = int(neq.sum().item())
            stats["frames_total"] = int(S * F)
            stats["frames_exact"] = int(S * F - n_bad)
            limit = 0 if args.ebn0 <= 0 else 0.01 * S * F
            if not bool((cnt == F).all()) or n_bad > limit:
                bad = [(k, int(cnt[k])) for k in range(S) if cnt[k] != F][:4]
                where = [(int(k), int(f)) for k, f in zip(*np.nonzero(neq.cpu().numpy()))][:8]
                raise SystemExit(f"bench.py: rank {rank}: decode check failed: streams with a wrong frame count "
                                 f"{bad}; {n_bad} frames differ from the transmitted ones, first (stream, frame) {where}")

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    dm.enable_timing(True)
    kt = []
    barrier()
    t0 = time.perf_counter()
    marks = [t0]
    for _ in range(args.steps):
        step(check=False)
        kt.append(dm.kernel_times())
        marks.append(time.perf_counter())                 # (every step ends in opv_sync, and in the gather when N > 1)
    barrier()
    dt = time.perf_counter() - t0
    step_s = np.diff(np.array(marks))
    step(check=True)                                      # untimed: the timed configuration decodes correctly
    fe_kernel = dm.frontend_kernel()                      # what opv_process actually launched for this stream count
    # the two input classes the product reports instead of reproducing (include/opv_demod.h: edge_ties, offset_ties)
    states = [dm.state(k) for k in range(S)]
    stats["edge_ties"] = int(sum(st.edge_ties for st in states))
    stats["offset_ties"] = int(sum(st.offset_ties for st in states))
    # live issue view of the dominant kernel: every wave times itself (s_memtime / s_memrealtime, opv_tap_wave_info)
    wi = [dm.wave_info(k) for k in range(S)]
    cyc = np.array([w[2] for w in wi], np.float64)
    tick = np.array([w[3] for w in wi], np.float64)
    syms = np.array([st.total_symbols for st in states], np.float64)
    per_wave = 16.0 if "_x16" in fe_kernel else 4.0 if "_x4" in fe_kernel else 1.0   # streams sharing a wave (k_frontend_x4.hip, k_frontend_x16.hip)
    live_cps = float(np.median(cyc / np.maximum(syms * per_wave, 1.0))) if cyc.min() > 0 else None
    live_clock = float(np.median(cyc / np.maximum(tick, 1.0)) * 100e6) if tick.min() > 0 else None
    collective = None
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ts = torch.tensor(step_s, dtype=torch.float64, device=t.device)
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)          # step k of the job = its slowest rank's step k
        step_s = ts.cpu().numpy()
        # what every rank says it owned: [first global stream, one past the last, samples per stream, frames released]
        own = torch.tensor([mine.start, mine.stop, n, int(counts_view.sum().item())], dtype=torch.int64, device=t.device)
        owned = [torch.empty_like(own) for _ in range(world)] if rank == 0 else None
        dist.gather(own, owned, dst=0)
        if rank == 0:
            g = gathered["frames"]                       # [world, S, fcap, 134] in global stream order
            assert bool((g[0].to(dev) == frames_view).all().item()), "gathered frames of rank 0 differ from the local ones"
            stats["gathered_equals_local_view"] = True
            collective = {"backend": dist.get_backend(), "world": world, "gathered_shape": list(g.shape),
                          "bytes_per_rank": int(frames_view.numel() + 4 * counts_view.numel()),
                          "op": "one dist.gather of the [S, cap, 134] frame buffer + one of the [S] counts per step "
                                "(zero-copy views of library memory), MAX all-reduce of the step time"}
            assert bool((gathered["counts"] == F).all().item()), "a gathered stream released a wrong number of frames"
            exp_all = np.stack([np.stack([workload.tx_frames(amd, gk, F) for gk in sharding.stream_range(r, world, world * S)])
                                for r in range(world)])
            same = (g[:, :, :F, :].cpu().numpy() == exp_all).all(axis=3)
            stats["gathered_frames_total"] = int(same.size)
            stats["gathered_frames_exact"] = int(same.sum())
            # who sent what, read back from the gathered bytes themselves: the Base-40 callsign of every stream's
            # first frame is S<global id> (workload.stream_params), so a rank that decoded a second copy of
            # another rank's shard shows up here (and in exp_all above)
            collective["rank_shards"] = [[int(o[0]), int(o[1])] for o in owned]
            collective["rank_frames_released"] = [int(o[3]) for o in owned]
            collective["gathered_callsigns"] = [amd.callsign_of(g[r, k, 0].cpu().numpy()) for r in range(world) for k in range(S)]
            assert same.mean() > 0.99, "frames gathered from the other ranks do not match what they were sent"

    total_samples = float(world) * S * n * args.steps
    msps = total_samples / dt / 1e6
    med_s = float(np.median(step_s))
    fe_ms = float(np.mean([k["msk_frontend"] for k in kt]))
    launch_samples = float(S) * n
    achieved = launch_samples * ALGO_BYTES_PER_SAMPLE / (fe_ms * 1e-3) / 1e9

    # HBM bytes and instruction counts of the dominant kernel come from separate rocprofv3 --pmc passes (bench.py
    # cannot profile itself); they are attached only when the stored profile is of THIS kernel and configuration.
    # Cycles per symbol and the clock are measured live in every run (above); a stored profile whose cycles differ
    # from the live ones by more than 3 % is flagged as stale.
    n_sym = float(syms.sum())
    traffic = None
    traffic_note = "no PMC passes recorded for this kernel / configuration (see profiles/collect.sh)"
    issue = {"wave_cycles_per_symbol": round(live_cps, 1) if live_cps else None,
             "clock_GHz": round(live_clock / 1e9, 3) if live_clock else None,
             "source": "live: s_memtime / s_memrealtime of every wave (opv_tap_wave_info), median over the streams"}
    n_simd = 1024
    waves = int(np.ceil(S / per_wave))
    try:
        tj = json.loads(sorted((ROOT / "profiles").glob("r[0-9][0-9]_traffic.json"))[-1].read_text())
        w = tj["workload"]
        if tj.get("kernel") == fe_kernel and (w["streams_per_gpu"], w["frames_per_stream"], w["ebn0"], w.get("f0_edge_hz", 1500.0)) == \
                (S, F, args.ebn0, workload.F0_EDGE_HZ):
            traffic = round(tj["hbm_bytes_per_launch"] / (fe_ms * 1e-3) / 1e9, 3)
            traffic_note = tj["source"] + "; " + tj["correction"]
            ips = tj.get("instr_per_symbol")
            if ips:
                per_sym = float(sum(ips.values()))
                issue["instr_per_symbol"] = ips
                issue["instr_source"] = "rocprofv3 SQ_INSTS_* pass of the same kernel and workload (" + tj["source"].split(";")[0] + ")"
                stored = tj.get("wave_cycles_per_symbol")
                if stored and live_cps:
                    issue["profile_wave_cycles_per_symbol"] = stored
                    issue["profile_stale"] = bool(abs(stored - live_cps) > 0.03 * live_cps)
                if live_cps:
                    # a wave-instruction occupies its SIMD's issue port for 4 cycles (MI355X_MICROARCH.md, 'vector-instruction
                    # ISSUE cost', one wave per SIMD): the fraction of its own SIMD's slots a stream's wave fills ...
                    issue["wave_issue_frac"] = round(per_sym * per_wave * 4.0 / (live_cps * per_wave), 3)
                    # ... and the same over the whole chip for the duration of the kernel
                    slots = n_simd * (fe_ms * 1e-3) * (live_clock or 2.4e9) / 4.0
                    issue["chip_issue_frac"] = round(per_sym * n_sym / slots, 4)
    except Exception:
        pass
    issue["waves"] = waves
    issue["simds"] = n_simd
    issue["note"] = ("wave_issue_frac: issued wave-instructions x 4 cycles / the wave's own cycles; chip_issue_frac: issued "
                     "wave-instructions / (1024 SIMDs x kernel cycles / 4)")
    # fp64 vector view: the REFERENCE formulation's arithmetic, ~150 flop per sample (2 sincos + 3 complex
    # lerps + 6 complex MACs), at the rate this kernel demodulates, against the fp64 vector peak (AMD spec 78.6 TFLOP/s =
    # half the fp32 vector peak of MI355X_MICROARCH.md). The kernel itself issues far fewer (no sincos, 60 lerps for 120).
    FP64_PEAK_TF, ALGO_FLOP_PER_SAMPLE = 78.6, 150.0
    fp64_tf = launch_samples * ALGO_FLOP_PER_SAMPLE / (fe_ms * 1e-3) / 1e12
    fp64_view = {"bound": "fp64-valu", "algorithmic_flop_per_sample": ALGO_FLOP_PER_SAMPLE, "achieved": round(fp64_tf, 3),
                 "peak": FP64_PEAK_TF, "unit": "TFLOP/s", "frac": round(fp64_tf / FP64_PEAK_TF, 5)}
    # The bound that actually binds (the HBM figure above it is the contract's): with at most one wave per SIMD a stream's
    # wave owns an issue port that takes one wave-instruction per 4 cycles, and the per-symbol feedback recurrence keeps
    # every symbol's instructions on that one port - frac = issued instructions x 4 / the wave's cycles; once every SIMD
    # carries stream waves the same count is taken against all 1024 ports for the duration of the kernel.
    if waves <= n_simd:
        binding = {"bound": "wave-issue", "frac": issue.get("wave_issue_frac"),
                   "achieved": round(4.0 * sum(issue["instr_per_symbol"].values()), 1) if "instr_per_symbol" in issue else None,
                   "peak": issue["wave_cycles_per_symbol"], "unit": "cycles per symbol (issuing / elapsed, one wave on its SIMD)",
                   "waves": waves, "simds": n_simd}
    else:
        binding = {"bound": "chip-issue", "frac": issue.get("chip_issue_frac"), "achieved": None, "peak": None,
                   "unit": "issued wave-instructions / (1024 SIMDs x kernel cycles / 4)", "waves": waves, "simds": n_simd}
    binding["note"] = ("frac needs the SQ_INSTS_* pass of this kernel and workload (profiles/collect.sh); null when the stored profile "
                       "is of another kernel or configuration") if binding["frac"] is None else \
                      "instruction counts from the stored rocprofv3 pass (roofline.issue.instr_source), cycles measured live in this run"
    if waves <= n_simd:
        regime = (f"issue/latency-bound per-symbol feedback recurrence: {waves} waves on {n_simd} SIMDs "
                  f"({100.0 * waves / n_simd:.1f} % of the chip's issue ports can be used at all), not bandwidth (DESIGN.md §3.1)")
    else:
        regime = f"{waves} waves on {n_simd} SIMDs: every SIMD carries a stream wave, the kernel is at the chip's issue capacity"

    # the bound that binds FIRST in the note: a record that keeps only the note's head still says why 0.4 % of HBM is not the story
    if binding["frac"] is not None and binding["bound"] == "wave-issue":
        binding_head = f"binding=wave-issue {binding['frac']:.3f} ({binding['achieved'] / 4.0:.0f} instr x 4 / {binding['peak']:.1f} cyc); "
    elif binding["frac"] is not None:
        binding_head = f"binding=chip-issue {binding['frac']:.3f} (issued wave-instructions / (1024 SIMDs x kernel cycles / 4)); "
    else:
        binding_head = f"binding={binding['bound']} n/a (no SQ_INSTS pass stored for this kernel / workload); "

    out = {
        "metric": "IQ Msamples/s demod+Viterbi (×real-time @2.168MSPS); BER vs ref",
        "value": round(msps, 3), "unit": "Msamples/s", "x_realtime": round(msps / 2.168, 1),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "median": {"ms_per_step": round(med_s * 1e3, 3), "value": round(float(world) * S * n / med_s / 1e6, 3),
                   "step_ms": [round(float(x) * 1e3, 3) for x in step_s],
                   "note": "median over the timed steps (SURVEY.md §8d); value / ms_per_step above are the contract's total-time figures"},
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"configs[3]: " if (S, F) == (64, 1000) else f"configs[3] shape at --streams {S} --frames {F}: ") +
                               f"{S} concurrent IQ streams/GPU x {F} frames, -s semantics, "
                               f"{workload.RECIPE} across the 64 streams of a shard (SURVEY.md §8d C4: edge streams on the AFC clamp, "
                               f"outside the +/-1530 Hz search span), Eb/N0 {args.ebn0:g} dB" +
                               (f"; x{world} GPUs = configs[4] shape, RCCL gather of frames to rank 0" if world > 1 else ""),
                   "streams_per_gpu": S, "frames_per_stream": F, "samples_per_stream": n,
                   "parallelism": f"streams sharded {S}/GPU, no data-path collective"},
        "frames_checked": f"rank0: {stats.get('frames_exact')}/{stats.get('frames_total')} decoded frames equal the "
                          f"transmitted bytes (rest = channel errors at {args.ebn0:g} dB); GPU==reference parity is tests/",
        "roofline": {"bound": "hbm", "kernel": fe_kernel, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                     "traffic_note": traffic_note,
                     "binding": binding,
                     "issue": issue,
                     "fp64_valu": fp64_view,
                     "kernel_ms": round(fe_ms, 3),
                     "note": binding_head + regime + "; extras.stream_sweep shows the front-end with the chip filled"},
        "kernel_ms": {k: round(float(np.mean([x[k] for x in kt])), 3) for k in kt[0]},
        "check": stats,
    }
    if collective:
        out["collective"] = collective
    # what a timed step spends outside the four kernels: opv_reset_stream (two stream syncs, a 90 KB upload, a
    # metrics fill), 64 opv_attach calls, the launches, and the frame gather when N > 1
    out["non_kernel_ms_per_step"] = round(out["ms_per_step"] - sum(out["kernel_ms"].values()), 3)

    if rank == 0 and not args.no_extras and world == 1:
        extras = {}
        budget = Budget(args.extras_budget)

        def put(key, est_s, fn):
            r = budget.run(key, est_s, fn)
            if r is not None:
                extras[key] = r
        # the clean 1000-frame capture of configs[1] (W5NYV), made on the device: the CPU baseline's input as well
        one = amd.Demod(1, max_samples=n + 64, streaming=True, device=dev_index)
        d_base = torch.empty(2 * n, dtype=torch.int16, device=dev)
        t0 = time.perf_counter()
        one.modulate_device(tx_frames, d_base.data_ptr())
        one.sync()
        t_dev_mod = time.perf_counter() - t0
        raw = d_base.cpu().numpy().tobytes()
        # the contract's cpu_baseline: not optional, only accounted for
        t0 = time.perf_counter()
        out["cpu_baseline"] = cpu_baseline(raw, n)
        budget.spent["cpu_baseline"] = round(time.perf_counter() - t0, 1)

        def x_configs1():                                # configs[1]: one clean 1000-frame stream
            for rep in range(2):
                one.reset()
                one.attach(0, d_base.data_ptr(), n, eof=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                one.process()
                one.sync()
                t1 = time.perf_counter() - t0
            fr, meta = one.pop_frames(0)
            assert np.array_equal(fr, tx_frames), "configs[1] frames differ"
            return {"Msamples/s": round(n / t1 / 1e6, 3), "ms": round(t1 * 1e3, 2),
                    "frames": int(len(fr)), "all_metric_0": bool((meta["viterbi_metric"] == 0).all())}
        put("configs1_single_clean_stream", 4.0, x_configs1)
        one.close()

        def x_all_clean():
            # SURVEY.md §8d C4's "all-clean variant for peak throughput": the same S per-stream BERT captures exactly as the
            # modulator emits them (full scale, no offset, no noise), same context, same step (reset, S attaches, opv_process,
            # sync); every frame must equal the transmitted one with Viterbi metric 0
            try:
                d_cl, tx_cl, _n = workload.generate(amd, dm, torch, dev, mine, F, None, clean=True)
                cl_s, cl_fe = [], []
                for rep in range(1 + 5):
                    dm.reset()
                    for k in range(S):
                        dm.attach(k, d_cl[k].data_ptr(), n, eof=True)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    dm.process()
                    dm.sync()
                    if rep:                                  # (the first one is the warm-up run)
                        cl_s.append(time.perf_counter() - t0)
                        cl_fe.append(dm.kernel_times()["msk_frontend"])
                exp_c = torch.from_numpy(tx_cl).to(dev)
                cl_bad = int((frames_view[:, :F, :] != exp_c).any(dim=2).sum().item())
                cl_cnt = bool((counts_view == F).all().item())
                cl_states = [dm.state(k) for k in range(S)]
                cl_med = float(np.median(cl_s))
                res_clean = {
                    "workload": f"{S} streams x {F} frames straight from the device modulator (amplitude 16383, no offset, no noise)",
                    "Msamples/s": round(S * n / cl_med / 1e6, 3), "x_realtime": round(S * n / cl_med / 2.168e6, 1),
                    "ms_per_step_median": round(cl_med * 1e3, 3), "steps": len(cl_s), "frontend_ms_median": round(float(np.median(cl_fe)), 3),
                    "every_stream_released_all_frames": cl_cnt, "frames_exact": S * F - cl_bad, "frames_total": S * F,
                    "frames_perfect": int(sum(st.frames_perfect for st in cl_states)),
                    "edge_ties": int(sum(st.edge_ties for st in cl_states)), "offset_ties": int(sum(st.offset_ties for st in cl_states))}
            finally:
                d_cl = exp_c = None
                torch.cuda.empty_cache()
                step(check=False)                        # dm's frame buffer holds the contract workload's frames again (compared below)
            assert cl_cnt and cl_bad == 0, "all-clean variant: a decoded frame differs from the transmitted one"
            return res_clean
        put("all_clean_variant", 8.0, x_all_clean)
        put("cpu_baseline_all_cores", 5.0, lambda: cpu_all_cores(raw, n))

        def x_sweep():                                   # throughput-bound regime: many short streams carved out of the resident captures
            sweep = {}
            for ns, nfr in ((128, 480), (192, 320), (256, 240), (512, 120), (1024, 60), (2048, 30), (4096, 15), (8192, 7), (16384, 3)):
                if nfr > F:
                    continue
                per = F // nfr
                if S * per < ns:
                    continue
                sub_n = nfr * FRAME_SAMPLES
                m = amd.Demod(ns, max_samples=sub_n + 64, streaming=True, device=dev_index)
                m.enable_timing(True)
                ent = {}
                for spw in (1, 4, 16):                      # streams per wavefront (opv_set_frontend)
                    if spw == 4 and ns < 4096:         # (the four-per-wave mapping is the automatic choice from 2049 streams on)
                        continue
                    if spw == 16 and ns < 4096:        # (sixteen per wave: 1024 waves = one per SIMD need 16 384 streams)
                        continue
                    if spw == 1 and ns > 8192:
                        continue
                    m.set_frontend(spw)
                    for rep in range(2):
                        m.reset()
                        for j in range(ns):
                            k, seg = j % S, j // S
                            m.attach(j, d_iq[k].data_ptr() + 4 * seg * sub_n, sub_n, eof=True)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        m.process()
                        m.sync()
                        t1 = time.perf_counter() - t0
                    f_, m_, c_, cap_ = m.device_frames()
                    cnt = torch.as_tensor(DevPtr(c_, (ns,), "<i4"), device=dev).cpu().numpy()
                    fe_ms = m.kernel_times()["msk_frontend"]
                    ent[f"{spw}_per_wave"] = {"Msamples/s": round(ns * sub_n / t1 / 1e6, 1), "ms": round(t1 * 1e3, 2),
                                              "frontend_alone_Msamples/s": round(ns * sub_n / fe_ms / 1e3, 1),
                                              "frames_released": int(cnt.sum())}
                sweep[f"{ns}x{nfr}"] = ent
                m.close()
            return sweep
        put("stream_sweep", 5.0, x_sweep)
        if "stream_sweep" in extras and "error" not in extras["stream_sweep"]:
            sweep = extras["stream_sweep"]
            tgt = [int(k.split("x")[0]) for k, v in sweep.items() if max(e["Msamples/s"] for e in v.values()) >= 21680.0]
            extras["streams_for_target"] = {"target_Msamples/s": 21680.0, "smallest_swept_stream_count_meeting_it": min(tgt) if tgt else None,
                                            "swept": sorted(int(k.split("x")[0]) for k in sweep)}

        def x_configs4():
            # BASELINE configs[4]'s whole workload (512 streams x F frames, 64 per GPU on eight of them) on THIS one GPU:
            # the same generator, global stream ids 0..511, everything resident in HBM (178 GB at F = 1000), one launch.
            free_b, _tot = torch.cuda.mem_get_info()
            need_b = 512 * n * 4 * 1.10 + (4 << 30)
            if free_b < need_b:
                return {"skipped": f"{free_b / 1e9:.0f} GB of HBM free, {need_b / 1e9:.0f} GB needed"}
            else:
                big = amd.Demod(512, max_samples=n + 64, streaming=True, device=dev_index)
                d_big, tx_big, _n = workload.generate(amd, big, torch, dev, range(512), F, args.ebn0)
                big.enable_timing(True)
                for rep in range(2):
                    big.reset()
                    for k in range(512):
                        big.attach(k, d_big[k].data_ptr(), n, eof=True)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    big.process()
                    big.sync()
                    t1 = time.perf_counter() - t0
                fv, cv = workload.frame_views(big, torch, dev)
                ok_counts = bool((cv == F).all().item())
                exp_b = torch.from_numpy(tx_big).to(dev)
                n_bad = int((fv[:, :F, :] != exp_b).any(dim=2).sum().item())
                res_big = {
                    "streams": 512, "frames_per_stream": F, "iq_GB_in_hbm": round(512 * n * 4 / 1e9, 1),
                    "Msamples/s": round(512 * n / t1 / 1e6, 1), "x_realtime": round(512 * n / t1 / 2.168e6, 0), "ms": round(t1 * 1e3, 2),
                    "kernel_ms": {k: round(v, 3) for k, v in big.kernel_times().items()},
                    "every_stream_released_all_frames": ok_counts, "frames_exact": 512 * F - n_bad, "frames_total": 512 * F}
                big.close()
                del d_big, fv, cv, exp_b
                torch.cuda.empty_cache()
            return res_big
        if not args.no_big and S == 64:
            put("configs4_workload_on_one_gpu", 9.0, x_configs4)

        def x_many():
            # The many-stream regime on data of its own (not carved out of the 64 captures): 32 768 independent streams x 8 frames,
            # every one its own BERT capture through the device channel (91 GB of IQ in HBM), one opv_process on the automatic
            # mapping - sixteen streams per wavefront, two waves per SIMD (k_frontend_x16.hip). The only place where this path's
            # HBM fraction is not negligible: reported with its own roofline figures.
            NS, NF = 32768, 8
            n8 = amd.lib().opv_tx_modulated_samples(NF)
            free_b, _tot = torch.cuda.mem_get_info()
            need_b = NS * n8 * 4 * 1.05 + (12 << 30)
            if free_b < need_b:
                return {"skipped": f"{free_b / 1e9:.0f} GB of HBM free, {need_b / 1e9:.0f} GB needed"}
            else:
                ms = amd.Demod(NS, max_samples=n8 + 64, streaming=True, device=dev_index)
                gt = {}
                d_ms, tx_ms, _n8 = workload.generate(amd, ms, torch, dev, range(NS), NF, args.ebn0, timing=gt)
                ms.enable_timing(True)
                for rep in range(2):
                    ms.reset()
                    for k in range(NS):
                        ms.attach(k, d_ms[k].data_ptr(), n8, eof=True)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    ms.process()
                    ms.sync()
                    t1 = time.perf_counter() - t0
                fv, cv = workload.frame_views(ms, torch, dev)
                exp_m = torch.from_numpy(tx_ms).to(dev)
                n_bad = int((fv[:, :NF - 1, :] != exp_m[:, :NF - 1, :]).any(dim=2).sum().item())
                ktm = ms.kernel_times()
                fe_gbs = NS * n8 * ALGO_BYTES_PER_SAMPLE / (ktm["msk_frontend"] * 1e-3) / 1e9
                res_many = {
                    "streams": NS, "frames_per_stream": NF, "iq_GB_in_hbm": round(NS * n8 * 4 / 1e9, 1), "generate_s": round(gt["generate_s"], 1),
                    "frontend_kernel": ms.frontend_kernel(),
                    "Msamples/s": round(NS * n8 / t1 / 1e6, 1), "x_realtime": round(NS * n8 / t1 / 2.168e6, 0), "ms": round(t1 * 1e3, 2),
                    "frontend_alone_Msamples/s": round(NS * n8 / ktm["msk_frontend"] / 1e3, 1),
                    "kernel_ms": {k: round(v, 3) for k, v in ktm.items()},
                    "roofline_frontend": {"bound": "hbm", "achieved": round(fe_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": round(fe_gbs / HBM_PEAK_GBS, 4)},
                    "frames_released": int(cv.sum().item()), "frames_exact_of_first_7": NS * (NF - 1) - n_bad, "frames_compared": NS * (NF - 1)}
                ms.close()
                del d_ms, fv, cv, exp_m
                torch.cuda.empty_cache()
            return res_many
        if not args.no_big and S == 64:
            put("many_streams_unique_captures", 14.0, x_many)

        def x_pcie():
            # PCIe-inclusive: the boundary's host-buffer entry points instead of HBM-resident captures. Pinned host copies of the
            # first FH frames of every stream are pushed in rounds of RH frames: "host_pushed" with one opv_push_iq per stream (a copy
            # each, on the library's copy stream), "host_pushed_batched" with one opv_push_iq_batch_async per round (ONE gather kernel
            # for the 64 blocks; opv_process queues behind it on the device) - either way round r + 1 crosses PCIe while the kernels
            # of round r run. Same streams attached in HBM are timed beside it. Then a live serving round.
            both = {}
            FH, RH = min(F, 100), 10
            if FH >= 2 * RH:
                sub_n = FH * FRAME_SAMPLES
                host = [d_iq[k][: 2 * sub_n].cpu().pin_memory() for k in range(S)]
                host_np = [h.numpy() for h in host]
                hp = amd.Demod(S, max_samples=sub_n + 64, streaming=True, device=dev_index)
                res = {}
                for mode in ("hbm_attached", "host_pushed", "host_pushed", "host_pushed_batched", "host_pushed_batched"):
                    hp.reset()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    if mode == "hbm_attached":
                        for k in range(S):
                            hp.attach(k, d_iq[k].data_ptr(), sub_n, eof=True)
                        hp.process()
                    else:
                        per = RH * FRAME_SAMPLES
                        for r in range(0, sub_n, per):
                            m = min(per, sub_n - r)
                            if mode == "host_pushed_batched":
                                hp.push_batch(range(S), [host_np[k][2 * r: 2 * (r + m)] for k in range(S)], wait=False)
                            else:
                                for k in range(S):
                                    hp.push(k, host_np[k][2 * r: 2 * (r + m)])
                            hp.process()
                        if mode == "host_pushed_batched":
                            hp.push_wait()
                        for k in range(S):
                            hp.flush(k)
                        hp.process()
                    hp.sync()
                    t1 = time.perf_counter() - t0
                    f_, m_, c_, cap_ = hp.device_frames()
                    cnt = torch.as_tensor(DevPtr(c_, (S,), "<i4"), device=dev).cpu().numpy()
                    res[mode] = {"Msamples/s": round(S * sub_n / t1 / 1e6, 1), "ms": round(t1 * 1e3, 2),
                                 "GB/s_over_pcie": None if mode == "hbm_attached" else round(S * sub_n * 4 / t1 / 1e9, 2),
                                 "frames_released": int(cnt.sum())}
                    if mode != "hbm_attached":     # the pushed streams are fresh (reset drops the attachment)
                        fr_h = torch.as_tensor(DevPtr(f_, (S, cap_, 134), "|u1"), device=dev)[:, : FH - 1, :]
                        assert bool((fr_h == frames_view[:, : FH - 1, :]).all().item()), "pushed-path frames differ from the attached run"
                res["config"] = f"{S} streams x {FH} frames, rounds of {RH} frames, pinned host buffers"
                both["pcie_inclusive"] = res
                hp.close()
                # live serving round: one 86720-sample chunk (40 ms of signal) per stream pushed from host memory,
                # processed, frames popped - what a multi-stream receiver does every 40 ms
                lv = amd.Demod(S, max_samples=4 * FRAME_SAMPLES + 65536, streaming=True, device=dev_index)
                rounds = []
                for r in range(min(14, FH - 1)):
                    blks = [host_np[k][2 * r * FRAME_SAMPLES: 2 * (r + 1) * FRAME_SAMPLES] for k in range(S)]
                    t0 = time.perf_counter()
                    lv.push_batch(range(S), blks)
                    t1 = time.perf_counter()
                    lv.process()
                    lv.sync()
                    t2 = time.perf_counter()
                    got = sum(len(lv.pop_frames(k)[0]) for k in range(S))
                    t3 = time.perf_counter()
                    rounds.append((t3 - t0, t1 - t0, t2 - t1, t3 - t2, got))
                med = lambda i: round(1e3 * float(np.median([x[i] for x in rounds[3:]])), 3)
                both["live_round"] = {"streams": S, "signal_ms_per_round": 40.0, "round_ms": med(0), "push_ms": med(1),
                                        "process_ms": med(2), "pop_ms": med(3), "frames_per_round": int(rounds[-1][4])}
                lv.close()
                del host, host_np
            return both
        both = budget.run("pcie_inclusive+live_round", 5.0, x_pcie)
        if both is not None:
            extras.update(both if "error" not in both else {"pcie_inclusive": both})
        put("cli_drop_in", 3.0, lambda: cli_drop_in(raw, n))
        if not args.no_big:
            put("live_capacity", 18.0, lambda: live_capacity(dev_index, budget))
        extras["budget"] = budget.report()
        extras["skipped_for_budget"] = budget.skipped
        out["extras"] = extras
        out["setup"] = {"device_modulate_s_all_streams": round(t_mod, 2),
                        "device_modulate_one_stream": {"s": round(t_dev_mod, 3), "Msamples/s": round(n / t_dev_mod / 1e6, 1),
                                                        "note": "opv_tx_modulate_device end to end on a fresh context: 134 B/frame H2D, k_tx_encode, k_tx_scan_frames, "
                                                                "k_tx_expand_phases (first call of a context), k_tx_modulate, count D2H + sync"}}
    elif rank == 0 and args.no_extras:
        out["cpu_baseline"] = None
    elif rank == 0:
        # N > 1: the same bounded CPU sample as the N = 1 line, timed on rank 0's host cores while
        # the other ranks wait at the end of the job (outside the timed region)
        raw = np.ascontiguousarray(amd.modulate(tx_frames)).tobytes()
        out["cpu_baseline"] = cpu_baseline(raw, n)

    dm.close()
    if rank == 0:
        print(json.dumps(out, ensure_ascii=False))
    if use_dist:
        dist.barrier()                 # rank 0 may still have been timing the CPU baseline: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
