"""one-off soak: EVERY stream of bench.py's workload (64 streams x 1000 frames, 16 dB, f0 -2000..+2000 Hz: BASELINE configs[3])
through the HIP path in one context, and through the CPU oracle (a worker process per host core, one stream each at a time in host memory;
`512 1000` is BASELINE configs[4]'s whole workload, global streams 0..511, on one GPU):
frames, Viterbi metrics, sync positions, tracker events (kind / count / symbol), symbol count, offset estimate of every stream.
tests/test_gpu_parity.py::test_config3_full_size_all_streams does the same in every suite run (round 4); this prints a line per stream."""
import sys
import time
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


from soak_inputs import host_workers, oracle_receive_job as oracle_job  # noqa: E402  (shared with tests/test_gpu_parity.py)


def main():
    import torch
    from __graft_entry__ import load_opv_amd, load_pkg_module
    amd, workload = load_opv_amd(), load_pkg_module("workload")
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    ebn0 = float(sys.argv[3]) if len(sys.argv) > 3 else 16.0        # (12 / 6 dB: the noise-dominated parity of SURVEY.md §7-5 on all streams)
    dev = torch.device("cuda", 0)
    n = amd.lib().opv_tx_modulated_samples(F)
    dm = amd.Demod(S, max_samples=n + 64, streaming=True)
    d_iq, tx, n = workload.generate(amd, dm, torch, dev, range(S), F, ebn0)
    for k in range(S):
        dm.attach(k, d_iq[k].data_ptr(), n, eof=True)
    t0 = time.time()
    dm.process()
    dm.sync()
    print(f"HIP path: {S} streams x {F} frames at Eb/N0 {ebn0:g} dB in {time.time() - t0:.2f} s ({dm.frontend_kernel()})", flush=True)
    bad = 0
    W = host_workers()
    with ProcessPoolExecutor(W) as pool:
        pending = {}
        for k in range(S):
            pending[k] = pool.submit(oracle_job, d_iq[k].cpu().numpy())
            if len(pending) >= W or k == S - 1:
                for j, fut in sorted(pending.items()):
                    e = fut.result()
                    fr, meta = dm.pop_frames(j)
                    ev = dm.pop_events(j)
                    st = dm.state(j)
                    ok = (np.array_equal(fr, e["frames"]) and np.array_equal(meta["viterbi_metric"], e["metrics"])
                          and np.array_equal(meta["release_symbol"], e["frame_sym"]) and len(ev) == len(e["events"])
                          and all(np.array_equal(ev[f], e["events"][f]) for f in ("kind", "count", "sym_idx"))
                          and st.total_symbols == e["n_soft"] and st.est_offset_hz == e["est_offset"]
                          and abs(st.freq_offset_hz - e["final_freq_offset"]) < 1e-6 and st.edge_ties == 0)
                    bad += not ok
                    print(f"stream {j:3d}: {len(fr)} frames, {int((fr[:F] == tx[j][:len(fr)]).all(axis=1).sum()) if len(fr) <= F else -1} equal to the transmitted ones, "
                          f"{len(ev)} tracker events, est {st.est_offset_hz:+.0f} Hz, final AFC {st.freq_offset_hz:+.3f} Hz (oracle {e['final_freq_offset']:+.3f}), "
                          f"offset_ties {st.offset_ties}: {'== oracle' if ok else 'DIFFERS'}", flush=True)
                pending = {}
    print(f"{S - bad} of {S} streams equal the oracle in every frame, metric, sync position and tracker event")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
