"""Level 0 of INTEGRATION.md, tested with the boundary's REAL caller: the reference's own `opv-modem -R -d <path>`
(oracle/_ref/opv-modem, compiled from /root/reference/src/opv-modem.cpp by oracle/Makefile; it travels to the GPU box as
a prebuilt file) forks and execs `<path> -s -r` (src/opv-modem.cpp:696-717), forwards stdin in 16 KB reads (:734,753),
reads 134-byte records non-blocking under select (:765-786) and sends them as UDP datagrams.

  * not gpu: the reference pair (`-d oracle/_ref/opv-demod`) reproduces the committed fixture
    (tests/golden/modem_parent.*, made by tests/golden/make_golden_modem.py) - the harness and the fixture are sound;
  * gpu:     the same parent in front of opv-cxx-demod_amd/bin/opv-demod delivers the same datagrams.
"""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from oracle_lib import Oracle, impair, ref_binary, run_reference_modem_loopback, run_under_reference_modem

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
OURS = ROOT / "opv-cxx-demod_amd" / "bin" / "opv-demod"


@pytest.fixture(scope="module")
def cases():
    meta = json.loads((GOLD / "modem_parent.json").read_text())
    arrays = np.load(GOLD / "modem_parent.npz")
    o = Oracle()
    out = {}
    for name, m in meta.items():
        iq = o.modulate(o.bert_frames(m["frames_sent"], m["callsign"]))
        if m["channel"]:
            iq = impair(iq, **m["channel"])
        assert hashlib.sha256(iq.tobytes()).hexdigest() == m["iq_sha256"], f"{name}: the recipe no longer makes the fixture's input"
        out[name] = (iq, arrays[name], m)
    return out


def _check(child, iq, expected, piece):
    grams, rc, err = run_under_reference_modem(child, iq, piece=piece)
    assert rc == 0, err[-2000:]
    assert all(len(g) == 134 for g in grams), sorted({len(g) for g in grams})
    got = np.frombuffer(b"".join(grams), np.uint8).reshape(-1, 134)
    assert got.shape == expected.shape, (got.shape, expected.shape, err[-1500:])
    assert np.array_equal(got, expected)
    assert f"RX:  {len(expected)} frames" in err            # the parent's own summary (src/opv-modem.cpp:833)


@pytest.mark.skipif(ref_binary("opv-modem") is None or ref_binary("opv-demod") is None, reason="oracle/_ref not built")
@pytest.mark.parametrize("name", ["clean12", "noisy100_14dB"])
def test_reference_pair_reproduces_the_fixture(cases, name, oracle):
    iq, expected, m = cases[name]
    _check(ref_binary("opv-demod"), iq, expected, 16384)
    assert np.array_equal(expected, oracle.receive(iq, streaming=True, want_soft=False)["frames"])   # = the child's stdout records


@pytest.mark.gpu
@pytest.mark.parametrize("name,piece", [("clean12", 16384), ("clean12", 1000), ("noisy100_14dB", 16384), ("noisy100_14dB", 4096)])
def test_reference_modem_drives_our_opv_demod(cases, name, piece):
    """`oracle/_ref/opv-modem -R -r <port> -d opv-cxx-demod_amd/bin/opv-demod`: the capture goes to the parent's stdin in
    pieces of <= 16 KB (and in odd 1000-byte pieces that split samples), the datagrams it sends must be the ones the
    reference's own opv-demod yields through the same parent (fixture), every one exactly 134 bytes."""
    assert ref_binary("opv-modem") is not None, "oracle/_ref/opv-modem did not travel with the snapshot"
    assert OURS.exists()
    iq, expected, m = cases[name]
    _check(OURS, iq, expected, piece)


# ---- the reference's loopback / repeater mode: a PERSISTENT child fed one frame at a time over a live link ---------------
def _loopback(child, oracle):
    sent = oracle.bert_frames(9, "KB5MU", 0x123456, 40)
    grams, err = run_reference_modem_loopback(child, sent)
    assert all(len(g) == 134 for g in grams), sorted({len(g) for g in grams})
    got = np.frombuffer(b"".join(grams), np.uint8).reshape(-1, 134)
    # frame k returns once frame k + 1 has been modulated into the child (the demodulator's one-frame latency,
    # src/opv-demod.cpp:221): 8 of 9 come back, in order, untouched
    assert len(got) == len(sent) - 1 and np.array_equal(got, sent[:-1]), (len(got), err[-1500:])
    assert "TX:  9 frames" in err and "RX:  8 frames" in err, err[-800:]


@pytest.mark.skipif(ref_binary("opv-modem") is None or ref_binary("opv-demod") is None, reason="oracle/_ref not built")
def test_reference_loopback_pair(oracle):
    """`opv-modem -l -d oracle/_ref/opv-demod`: what the reference pair itself does on a live link (the behaviour the GPU
    test below demands of our child)."""
    _loopback(ref_binary("opv-demod"), oracle)


@pytest.mark.gpu
def test_reference_modem_loopback_with_our_opv_demod(oracle):
    """`oracle/_ref/opv-modem -l -d opv-cxx-demod_amd/bin/opv-demod`: the reference's loopback server (src/opv-modem.cpp:855-1000)
    keeps ONE child alive (PersistentDemodulator :348-468: child's stderr to /dev/null, a blocking 347 KB write per frame,
    non-blocking reads), modulates every UDP frame it receives into it and returns what the child decodes. Nine frames sent 120
    ms apart: eight come back, in order, byte for byte - the reference pair's behaviour (test above)."""
    assert ref_binary("opv-modem") is not None, "oracle/_ref/opv-modem did not travel with the snapshot"
    _loopback(OURS, oracle)
