"""dev: cost of an idle channel - exact digital silence (every symbol goes through the out-of-line silence routine), silence
with single +/-1 LSB samples (one-tap windows: edge_ties), and weak noise - against a normal signal, one stream, F frames' worth."""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from __graft_entry__ import load_opv_amd
amd = load_opv_amd()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = F * 86720
rng = np.random.default_rng(1)
caps = {"signal": amd.modulate(amd.bert_frames(F))[: 2 * n], "zeros": np.zeros(2 * n, np.int16)}
z = np.zeros(2 * n, np.int16); z[rng.integers(0, 2 * n, n // 500)] = 1; caps["zeros + sparse LSBs"] = z
caps["weak noise"] = rng.integers(-2, 3, 2 * n).astype(np.int16)
for name, x in caps.items():
    d = amd.Demod(1, max_samples=n + 64, streaming=True)
    d.enable_timing(True)
    dx = torch.from_numpy(x).cuda()
    d.attach(0, dx.data_ptr(), n, eof=True)
    d.process(); d.sync()
    st = d.state(0); kt = d.kernel_times()
    print(f"{name:22s}: front-end {kt['msk_frontend']:.1f} ms for {st.total_symbols} symbols = {kt['msk_frontend'] * 1e6 / max(st.total_symbols, 1):.0f} ns per symbol "
          f"({n / kt['msk_frontend'] / 1e3 / 2.168:.0f}x real-time), edge_ties {st.edge_ties}, frames {st.frames_released}", flush=True)
    d.close()
