#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE COMPILED REFERENCE.

Run in the build container only (needs /root/reference, built by `make -C oracle ref` into
oracle/_ref/). What is written here is DATA: inputs (or their sha256 + the recipe that makes
them) and the reference's outputs. No reference source text is stored.

  python tests/golden/make_golden.py          # ~2 min (hashes a 1000-frame run too)

Sources of truth used:
  * oracle/_ref/opv-mod, oracle/_ref/opv-demod  — the reference binaries themselves
    (stdout bytes, stderr event lines).
  * oracle/_ref/libopv_ref.so — the reference classes driven exactly like the reference's
    main() does (tests/oracle_lib.py:Reference.receive), for soft symbols and carry state.
    Its frames/logs are cross-checked against the binaries before anything is written.
"""
import hashlib
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
from oracle_lib import (CODED_BITS, FRAME_BYTES, Oracle, Reference, impair, ref_binary)  # noqa: E402


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def run_mod(args, stdin=b""):
    return subprocess.run([str(ref_binary("opv-mod"))] + args, input=stdin, stdout=subprocess.PIPE,
                          stderr=subprocess.DEVNULL, check=True).stdout


def run_demod(iq_bytes, args):
    p = subprocess.run([str(ref_binary("opv-demod"))] + args, input=iq_bytes, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    return p.stdout, p.stderr.decode("utf-8"), p.returncode


def event_lines(stderr_text):
    return [ln for ln in stderr_text.split("\n") if ln.startswith("[") and "]" in ln and "s]" not in ln[:12]]


def main():
    assert Reference.available() and ref_binary("opv-mod") and ref_binary("opv-demod"), "run `make -C oracle ref` first"
    ref = Reference()
    meta = {}
    arrays = {}

    # ---- transmit chain pins ---------------------------------------------------------------
    tx = {}
    for n in (10, 100, 1000):
        iq = run_mod(["-S", "W5NYV", "-B", str(n)])
        tx[str(n)] = {"bytes": len(iq), "sha256": sha(iq)}
        if n == 10:
            iq10 = iq
        if n == 100:
            iq100 = iq
        if n == 1000:
            for mode, args in (("stream", ["-s", "-r", "-q"]),):
                out, err, rc = run_demod(iq, args)
                tx["1000_frames_" + mode] = {"bytes": len(out), "sha256": sha(out), "rc": rc,
                                             "events_sha256": sha("\n".join(event_lines(err)).encode())}
        del iq
    meta["opv_mod_bert_W5NYV"] = tx

    # raw-mode KAT of the reference Makefile:28-33 (3 hand-built frames through -R)
    raw_frames = bytes(sum(([0, 0, 3, 0x74, 0x26, 0x97, 0xBB, 0xAA, 0xDD, 0, 0, 0] +
                            [(i + j) & 0xFF for j in range(122)] for i in range(3)), []))
    raw_iq = run_mod(["-R"], stdin=raw_frames)
    out, err, rc = run_demod(raw_iq, ["-s", "-r"])
    assert out == raw_frames
    meta["raw_kat"] = {"iq_sha256": sha(raw_iq), "iq_bytes": len(raw_iq), "frames_sha256": sha(raw_frames)}
    arrays["raw_kat_frames"] = np.frombuffer(raw_frames, np.uint8).reshape(3, FRAME_BYTES)

    # ---- config 1: W5NYV x10, both modes ---------------------------------------------------
    iq = np.frombuffer(iq10, np.int16)
    for mode, streaming, args in (("stream", True, ["-s", "-r"]), ("batch", False, ["-r"])):
        out, err, rc = run_demod(iq10, args)
        r = ref.receive(iq, streaming=streaming)
        assert out == r["frames"].tobytes(), "harness driver disagrees with the reference binary"
        assert event_lines(err) == r["log"].strip().split("\n")
        arrays[f"c1_{mode}_frames"] = r["frames"]
        arrays[f"c1_{mode}_metrics"] = r["metrics"]
        arrays[f"c1_{mode}_quality"] = r["quality"]
        arrays[f"c1_{mode}_frame_sym"] = r["frame_sym"]
        arrays[f"c1_{mode}_soft"] = r["soft"]
        arrays[f"c1_{mode}_chunks"] = r["chunks"]  # freq_offset, timing_freq, (nan), leftover, nsoft
        meta[f"c1_{mode}"] = {"rc": rc, "est_offset": r["est_offset"], "final_freq_offset": r["final_freq_offset"],
                              "final_timing_freq": r["final_timing_freq"], "final_state": r["final_state"],
                              "events": r["log"].strip().split("\n"), "frames_sha256": sha(out),
                              "stderr_sha256": sha(err.encode())}
        if mode == "stream":
            (HERE / "c1_stream_stderr.txt").write_text(err)
    # the -h text of the reference binary (src/opv-demod.cpp:962-971), argv[0] normalised
    _, err, rc = run_demod(b"", ["-h"])
    assert rc == 0
    lines = err.split("\n")
    lines[0] = "Usage: opv-demod [options] < input.iq"
    (HERE / "usage_stderr.txt").write_text("\n".join(lines))
    out, err, rc = run_demod(iq10, ["-s", "-r", "-q", "-o", "1000"])
    r = ref.receive(iq, streaming=True, init_offset=1000.0)
    assert out == r["frames"].tobytes()
    arrays["c1_stream_o1000_soft"] = r["soft"]
    arrays["c1_stream_o1000_frames"] = r["frames"]
    meta["c1_stream_o1000"] = {"events": r["log"].strip().split("\n"), "final_freq_offset": r["final_freq_offset"]}

    # ---- 100 frames clean ------------------------------------------------------------------
    out, err, rc = run_demod(iq100, ["-s", "-r", "-q"])
    meta["c100_stream"] = {"frames_sha256": sha(out), "bytes": len(out), "events_sha256": sha("\n".join(event_lines(err)).encode())}
    arrays["c100_stream_frames"] = np.frombuffer(out, np.uint8).reshape(-1, FRAME_BYTES)

    # ---- frame-decoder taps for the first 3 payloads of config 1 --------------------------
    ora = Oracle()  # only for re-deriving q/deint the way SURVEY §8c describes; cross-checked below
    r = ref.receive(iq, streaming=True)
    # recover the payloads: softs between sync+1 .. sync+2144
    pay = []
    for fs in r["frame_sym"][:3]:
        pay.append(r["soft"][int(fs) - CODED_BITS + 1: int(fs) + 1])
    pay = np.array(pay)
    taps_bits, taps_metric, taps_frames, taps_deint = [], [], [], []
    perm = ref.deinterleave_perm()
    for p in pay:
        m, fr = ref.frame_decode(p)
        scale = 0.0
        for v in p:
            scale += abs(v)
        scale /= CODED_BITS
        q = np.clip(((-p / scale) * 3.5 + 3.5 + 0.5).astype(np.int64), 0, 7).astype(np.int32)
        de = q[perm]
        m2, bits = ref.viterbi(de)
        assert m2 == m
        taps_bits.append(bits); taps_metric.append(m); taps_frames.append(fr); taps_deint.append(de)
    arrays["taps_payload_soft"] = pay
    arrays["taps_deint"] = np.array(taps_deint, np.int8)
    arrays["taps_bits"] = np.array(taps_bits, np.uint8)
    arrays["taps_metric"] = np.array(taps_metric, np.int32)
    arrays["taps_frames"] = np.array(taps_frames, np.uint8)
    arrays["deinterleave_perm"] = perm

    # ---- noisy / offset configurations (inputs DEFINED by oracle_lib.impair) ---------------
    base = np.frombuffer(iq100, np.int16)
    noisy = {}
    for tag, kw in (("p2000_12dB", dict(amp=2000.0, f0_hz=2000.0, ebn0_db=12.0, seed=1)),
                    ("m2000_6dB", dict(amp=2000.0, f0_hz=-2000.0, ebn0_db=6.0, seed=1)),
                    ("p700_16dB", dict(amp=2000.0, f0_hz=700.0, ebn0_db=16.0, seed=7)),
                    ("p2000_clean", dict(amp=2000.0, f0_hz=2000.0, ebn0_db=None, seed=1))):
        x = impair(base, **kw)
        out, err, rc = run_demod(x.tobytes(), ["-s", "-r", "-q"])
        r = ref.receive(x, streaming=True)
        assert out == r["frames"].tobytes()
        assert event_lines(err) == [ln for ln in r["log"].strip().split("\n") if ln]
        noisy[tag] = {"recipe": kw, "input_sha256": sha(x.tobytes()), "frames_sha256": sha(out), "rc": rc,
                      "n_frames": int(len(r["frames"])), "est_offset": r["est_offset"],
                      "final_freq_offset": r["final_freq_offset"], "n_soft": int(len(r["soft"])),
                      "events_sha256": sha(r["log"].strip().encode())}
        arrays[f"n_{tag}_frames"] = r["frames"]
        arrays[f"n_{tag}_metrics"] = r["metrics"]
        arrays[f"n_{tag}_frame_sym"] = r["frame_sym"]
        arrays[f"n_{tag}_soft_strided"] = r["soft"][::97].copy()
        (HERE / f"n_{tag}_events.txt").write_text(r["log"])
    meta["noisy_100"] = noisy

    # ---- constants -------------------------------------------------------------------------
    meta["base40_W5NYV"] = "000003742697"   # reference Makefile:29
    meta["est_offset_c1"] = ref.estimate_offset(iq)

    np.savez_compressed(HERE / "golden.npz", **arrays)
    (HERE / "golden.json").write_text(json.dumps(meta, indent=1, ensure_ascii=False))
    print("wrote", HERE / "golden.npz", (HERE / "golden.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
