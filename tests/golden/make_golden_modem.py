#!/usr/bin/env python3
"""Fixture for the Level-0 integration test: what the REFERENCE's own pair `opv-modem -R -d opv-demod` delivers.

Run in the build container only (needs oracle/_ref/opv-modem and oracle/_ref/opv-demod, built from /root/reference by
`make -C oracle ref`). For each case the input capture is made by the recipe below (oracle modulator + the seeded numpy
channel of tests/oracle_lib.py::impair: reproducible anywhere), written to the parent's stdin in 16 KB pieces, and the
UDP datagrams the parent sends are recorded. What is stored is DATA: the input's sha256 + recipe, the datagram bytes.

  python tests/golden/make_golden_modem.py
"""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
from oracle_lib import Oracle, impair, ref_binary, run_under_reference_modem  # noqa: E402

CASES = {   # name: (frames, callsign, channel)
    "clean12": (12, "W5NYV", None),
    "noisy100_14dB": (100, "KB5MU", dict(amp=2000.0, f0_hz=-640.0, ebn0_db=14.0, seed=31)),
}


def capture(o, name):
    n, cs, ch = CASES[name]
    iq = o.modulate(o.bert_frames(n, cs))
    return impair(iq, **ch) if ch else iq


def main():
    assert ref_binary("opv-modem") and ref_binary("opv-demod"), "run `make -C oracle ref` first"
    o = Oracle()
    meta, arrays = {}, {}
    for name in CASES:
        iq = capture(o, name)
        grams, rc, err = run_under_reference_modem(ref_binary("opv-demod"), iq)
        assert rc == 0 and all(len(g) == 134 for g in grams), (rc, err[-500:])
        fr = np.frombuffer(b"".join(grams), np.uint8).reshape(-1, 134)
        exp = o.receive(iq, streaming=True, want_soft=False)["frames"]
        # the parent adds nothing and drops nothing: its datagrams ARE the child's stdout records
        assert np.array_equal(fr, exp), f"{name}: datagrams differ from the oracle's frames ({len(fr)} vs {len(exp)})"
        arrays[name] = fr
        meta[name] = {"frames_sent": CASES[name][0], "callsign": CASES[name][1], "channel": CASES[name][2],
                      "iq_sha256": hashlib.sha256(iq.tobytes()).hexdigest(), "iq_samples": int(iq.size // 2),
                      "datagrams": int(len(fr)), "parent_exit": rc}
        print(name, meta[name])
    np.savez_compressed(HERE / "modem_parent.npz", **arrays)
    (HERE / "modem_parent.json").write_text(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
