"""dev soak: one stream of ~2.0e9 samples (8 GB of IQ, just under the 2^31-sample limit of an attached
capture) in -s and batch mode: exercises 32-bit sample indices / >4 GB byte offsets in the kernels."""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from __graft_entry__ import load_opv_amd
amd = load_opv_amd()
F, REP = 1000, 24
tx = amd.bert_frames(F)
n1 = amd.lib().opv_tx_modulated_samples(F)
tmp = amd.Demod(1, max_samples=1 << 20)
d_one = torch.empty(2 * n1, dtype=torch.int16, device="cuda")
tmp.modulate_device(tx, d_one.data_ptr()); tmp.sync(); tmp.close()
n = n1 * REP
assert n < 2 ** 31, n
d_all = d_one.repeat(REP)
print("samples", n, "bytes", d_all.numel() * 2)
for streaming in (True, False):
    d = amd.Demod(1, max_samples=n + 64, streaming=streaming)
    d.attach(0, d_all.data_ptr(), n, eof=True)
    t0 = time.perf_counter(); d.process(); d.sync(); dt = time.perf_counter() - t0
    fr, meta = d.pop_frames(0)
    st = d.state(0)
    good = sum(int(np.array_equal(fr[i], tx[i % F])) for i in range(len(fr))) if len(fr) == F * REP else -1
    per_rep = [(fr[k * F:(k + 1) * F] == tx).all(axis=1).sum() if len(fr) >= (k + 1) * F else None for k in (0, REP // 2, REP - 1)]
    print(f"streaming={streaming}: {dt:.1f} s, frames {len(fr)} (expected {F * REP}), exact in order {good}, first/mid/last repeat exact {per_rep}, "
          f"symbols {st.total_symbols}, chunks {st.n_chunks}, origin {st.chunk_origin}, metrics zero {(meta['viterbi_metric'] == 0).mean():.4f}")
    d.close()

# junction behaviour against the oracle on three repeats (260 M samples; ~16 s of CPU per mode)
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from oracle_lib import Oracle
O = Oracle()
x = d_one.repeat(3).cpu().numpy()
for streaming in (True, False):
    d = amd.Demod(1, max_samples=x.size // 2 + 64, streaming=streaming)
    g = d.receive([x])[0]
    d.close()
    e = O.receive(x, streaming=streaming, want_soft=True)
    ok = np.array_equal(g["frames"], e["frames"]) and np.array_equal(g["meta"]["viterbi_metric"], e["metrics"]) and \
        np.array_equal(g["meta"]["release_symbol"], e["frame_sym"])
    err = np.max(np.abs(g["soft"] - e["soft"])) / np.mean(np.abs(e["soft"]))
    print(f"3 repeats vs oracle, streaming={streaming}: frames/metrics/sync identical {ok} ({len(e['frames'])} frames), soft max err {err:.2e}")
