"""dev: what the host-libm decision of offset-search ties costs a many-stream round: 8192 streams x 7 frames carved out of the bench
captures, sixteen per wave, three timed rounds each with the host decision on and off (OPV_OFFSET_DISTRUST_LIBM). Round 5: the caller
waited for the search kernel and decided on its own thread; round 6: a host function in stream order (the opv_process CALL is timed
separately from the round). -> profiles/r05_tie_cost_probe.txt, profiles/r06_tie_cost_probe.txt"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from __graft_entry__ import load_opv_amd, load_pkg_module
amd, workload = load_opv_amd(), load_pkg_module("workload")
dev = torch.device("cuda", 0)
S0, F = 64, 1000
n = amd.lib().opv_tx_modulated_samples(F)
gen = amd.Demod(1, max_samples=n + 64, streaming=True)
d_iq, tx, n = workload.generate(amd, gen, torch, dev, range(S0), F, 16.0)
gen.close()
FS = 86720
for distrust in (False, True, False, True):
    os.environ.pop("OPV_OFFSET_DISTRUST_LIBM", None)
    if distrust: os.environ["OPV_OFFSET_DISTRUST_LIBM"] = "1"
    ns, nfr = 8192, 7
    sub_n = nfr * FS
    m = amd.Demod(ns, max_samples=sub_n + 64, streaming=True)
    m.enable_timing(True)
    m.set_frontend(16)
    ts, calls = [], []
    for rep in range(3):
        m.reset()
        for j in range(ns):
            k, seg = j % S0, j // S0
            m.attach(j, d_iq[k].data_ptr() + 4 * seg * sub_n, sub_n, eof=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); m.process(); calls.append(time.perf_counter() - t0); m.sync(); ts.append(time.perf_counter() - t0)
    ties = [m.state(j).offset_ties for j in range(ns)]
    print("host ties" if m.offset_ties_on_host() else "device only", "round ms", [round(t * 1e3, 2) for t in ts], "of which the opv_process call", [round(t * 1e3, 2) for t in calls],
          "decided on the host", m.offset_ties_decided_on_host(), "kernel ms", {k: round(v, 2) for k, v in m.kernel_times().items()},
          "guarded streams", int(np.sum(np.array(ties) > 0)), "sum ties", int(np.sum(ties)))
    m.close()
