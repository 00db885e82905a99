#!/usr/bin/env python3
"""BER / FER versus Eb/N0 of the receive chain on MI355X (the curve the reference
never published). Device modulator -> device channel (amp 2000, f0 = +700 Hz, AWGN) -> the HIP hot
path; decoded frames are compared with the transmitted ones by position in the stream.
Run on the GPU box: python scripts/ber_curve.py > gpurun_out/ber_curve.json"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
from amd_lib import load  # noqa: E402

amd = load()
S, F = 32, 250
dev = torch.device("cuda", 0)
n = amd.lib().opv_tx_modulated_samples(F)
clean = torch.empty(2 * n, dtype=torch.int16, device=dev)
iq = torch.empty((S, 2 * n), dtype=torch.int16, device=dev)
dm = amd.Demod(S, max_samples=n + 64, streaming=True)
tx = np.stack([amd.bert_frames(F, callsign=f"B{k}", first=37 * k) for k in range(S)])
rows = []
for ebn0 in [6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18]:
    sigma = float(np.sqrt(80.0 * 2000.0 ** 2 / 10.0 ** (ebn0 / 10.0) / 2.0))
    dm.reset()
    for k in range(S):
        dm.modulate_device(tx[k], clean.data_ptr())
        dm.channel(clean.data_ptr(), iq[k].data_ptr(), n, gain=2000.0 / 16383.0, f0_hz=700.0, sigma=sigma,
                   seed=ebn0 * 1000 + k)
    dm.sync()
    for k in range(S):
        dm.attach(k, iq[k].data_ptr(), n, eof=True)
    dm.process()
    sent = released = exact = bit_err = bits = 0
    for k in range(S):
        fr, meta = dm.pop_frames(k)
        sent += F
        released += len(fr)
        # a frame released at symbol r carries payload of transmitted frame (r - 2167) / 2168 when locked to the grid
        idx = (meta["release_symbol"].astype(np.int64) - 2167) // 2168
        on_grid = ((meta["release_symbol"].astype(np.int64) - 2167) % 2168 == 0) & (idx >= 0) & (idx < F)
        for f, i, ok in zip(fr, idx, on_grid):
            if not ok:
                continue
            d = np.unpackbits(f ^ tx[k][i])
            bit_err += int(d.sum())
            bits += d.size
            exact += int(d.sum() == 0)
    rows.append({"ebn0_db": ebn0, "frames_sent": sent, "frames_released": released, "frames_exact": exact,
                 "fer": round(1 - exact / sent, 5), "ber_over_on_grid_frames": (bit_err / bits) if bits else None})
    print(rows[-1], file=sys.stderr)
print(json.dumps({"config": f"{S} streams x {F} frames, amp 2000, f0 +700 Hz, device AWGN, -s semantics", "rows": rows}, indent=1))
