"""EXPERIMENT (round 5): the one-wave front-end with SIX streams per workgroup (two waves on a SIMD, 256 VGPRs:
`make -C opv-cxx-demod_amd wg6`, OPV_LIB=.../build/wg6/libopv_demod_hip.so) beside the shipped four-per-workgroup launch, at 1024 ... 2048
streams: front-end time, cycles per symbol of the waves, and whether every stream's frames are the same.
(needs the tree of commit 55ab88e, which carries the kernel and `make wg6`)
usage (GPU box): OPV_LIB=$PWD/opv-cxx-demod_amd/build/wg6/libopv_demod_hip.so python scripts/experiments/wg6_probe.py"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_opv_amd, load_pkg_module  # noqa: E402

amd, workload = load_opv_amd(), load_pkg_module("workload")
dev = torch.device("cuda", 0)
D, F = 64, 30
n = amd.lib().opv_tx_modulated_samples(F)
gen = amd.Demod(1, max_samples=n + 64, streaming=True)
d_iq, tx, n = workload.generate(amd, gen, torch, dev, range(D), F, 16.0)
gen.close()
for S in (1024, 1536, 2048):
    res = {}
    for wg6 in (False, True):
        os.environ.pop("OPV_WG6", None)
        if wg6:
            os.environ["OPV_WG6"] = "1"
        dm = amd.Demod(S, max_samples=n + 64, streaming=True)
        dm.set_frontend(1)
        dm.enable_timing(True)
        for rep in range(2):
            dm.reset()
            for k in range(S):
                dm.attach(k, d_iq[k % D].data_ptr(), n, eof=True)
            dm.process()
            dm.sync()
        fe = dm.kernel_times()["msk_frontend"]
        fv, cv = workload.frame_views(dm, torch, dev)
        wi = [dm.wave_info(k) for k in range(0, S, 7)]
        st = [dm.state(k) for k in range(0, S, 7)]
        cps = np.array([w[2] / max(s.total_symbols, 1) for w, s in zip(wi, st)])
        res[wg6] = (fv[:, :F].clone(), cv.clone())
        print(f"S={S} F={F} {dm.frontend_kernel():24s} front-end {fe:7.2f} ms = {S * n / fe / 1e3:8.1f} Msamples/s; cycles per symbol of a wave "
              f"min/med/max {cps.min():.0f}/{np.median(cps):.0f}/{cps.max():.0f}; frames released {int(cv.sum())}")
        dm.close()
    same = bool(torch.equal(res[False][0], res[True][0])) and bool(torch.equal(res[False][1], res[True][1]))
    print(f"S={S}: frames and counts identical between the two launches: {same}")
