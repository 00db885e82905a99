#!/usr/bin/env python3
"""Golden fixture for `opv-demod -c` (coherent / Costas-loop mode, reference src/opv-demod.cpp:365-572,
:1144-1161), made FROM THE COMPILED REFERENCE (oracle/_ref). Data only: the 10-frame loopback
capture is regenerated from its recipe; stored are the reference's soft symbols, decoded bytes and
tracker lines.

  python tests/golden/make_golden_coherent.py
"""
import hashlib
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
from oracle_lib import Oracle, Reference, impair, ref_binary  # noqa: E402


def main():
    assert Reference.available() and ref_binary("opv-demod"), "run `make -C oracle ref` first"
    ref, orc = Reference(), Oracle()
    clean = orc.modulate(orc.bert_frames(10))
    meta, arrays = {}, {}
    for tag, cap, pll in (("clean", clean, 50.0), ("p700_16dB_pll20", impair(clean, 2000.0, 700.0, 16.0, seed=11), 20.0)):
        args = ["-c", "-r"] + ([] if pll == 50.0 else ["-p", str(pll)])
        p = subprocess.run([str(ref_binary("opv-demod"))] + args, input=cap.tobytes(), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE)
        r = ref.receive(cap, streaming=False, coherent=True, pll_bw=pll)
        assert r["frames"].tobytes() == p.stdout, "harness and binary disagree"
        lines = [ln for ln in p.stderr.decode("utf-8").split("\n")
                 if ln.startswith("[") or ln.startswith("Estimated") or ln.startswith("PLL") or ln.startswith("Demodulated")
                 or ln.startswith("Summary") or ln.startswith("Final state")]
        meta[tag] = {"iq_sha256": hashlib.sha256(cap.tobytes()).hexdigest(), "pll_bw": pll,
                     "stdout_sha256": hashlib.sha256(p.stdout).hexdigest(), "n_frames": int(len(r["frames"])),
                     "exit": p.returncode, "est_offset": r["est_offset"], "final_freq_offset": r["final_freq_offset"],
                     "stderr_lines": lines}
        arrays[tag + "_soft"] = r["soft"]
        arrays[tag + "_frames"] = r["frames"]
        arrays[tag + "_metrics"] = r["metrics"]
        arrays[tag + "_frame_sym"] = r["frame_sym"]
    (HERE / "coherent.json").write_text(json.dumps(meta, indent=1, ensure_ascii=False))
    np.savez_compressed(HERE / "coherent.npz", **arrays)
    print({k: (v["n_frames"], v["final_freq_offset"]) for k, v in meta.items()})


if __name__ == "__main__":
    main()
