"""CPU-only: pin oracle/opv_oracle.c against fixtures made by the compiled reference
(tests/golden/make_golden.py) and, where oracle/_ref/libopv_ref.so exists, against the
reference classes live. Everything here is bit-exact (==), including fp64 soft symbols."""
import hashlib

import numpy as np
import pytest

from oracle_lib import (CODED_BITS, FRAME_BYTES, Oracle, Reference, format_events, impair)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------ transmit chain
def test_modulator_sha256_10_and_100(oracle, golden, iq10, iq100):
    _, meta = golden
    pins = meta["opv_mod_bert_W5NYV"]
    assert iq10.nbytes == pins["10"]["bytes"] and sha(iq10) == pins["10"]["sha256"]
    assert iq100.nbytes == pins["100"]["bytes"] and sha(iq100) == pins["100"]["sha256"]


def test_base40_and_bert_frame(oracle, golden):
    _, meta = golden
    assert oracle.base40("W5NYV").tobytes().hex() == meta["base40_W5NYV"]
    f = oracle.bert_frames(2, first=5)
    assert f[0, :6].tobytes().hex() == meta["base40_W5NYV"]
    assert f[0, 6:12].tobytes().hex() == "bbaadd000000"
    assert f[1, 12] == 6 and f[1, 133] == (6 + 121) & 0xFF


def test_raw_mode_kat(oracle, golden):
    """reference Makefile:28-33 — three hand-built frames through opv-mod -R | opv-demod -s -r"""
    arrays, meta = golden
    frames = arrays["raw_kat_frames"]
    iq = oracle.modulate(frames)
    assert sha(iq) == meta["raw_kat"]["iq_sha256"]
    r = oracle.receive(iq, streaming=True)
    assert np.array_equal(r["frames"], frames)


def test_lfsr_table_prefix_and_interleaver(oracle, golden):
    arrays, _ = golden
    t = oracle.lfsr_table()
    assert t[:16].tobytes().hex() == "ff1aaf6652231e10a0f9fa8a98677dd2" and t[133] == 0x31
    perm = oracle.deinterleave_perm()
    assert np.array_equal(perm, arrays["deinterleave_perm"])
    assert sorted(perm.tolist()) == list(range(CODED_BITS))
    assert perm[:8].tolist() == [7, 68, 129, 206, 267, 328, 405, 466]


# ------------------------------------------------------------------ receive chain
@pytest.mark.parametrize("mode", ["stream", "batch"])
def test_config1_bit_exact(oracle, golden, iq10, mode):
    arrays, meta = golden
    r = oracle.receive(iq10, streaming=(mode == "stream"))
    m = meta[f"c1_{mode}"]
    assert np.array_equal(r["frames"], arrays[f"c1_{mode}_frames"])
    assert sha(r["frames"]) == m["frames_sha256"]
    assert np.array_equal(r["soft"], arrays[f"c1_{mode}_soft"])  # fp64, bit for bit
    assert np.array_equal(r["metrics"], arrays[f"c1_{mode}_metrics"])
    assert np.array_equal(r["quality"], arrays[f"c1_{mode}_quality"])
    assert np.array_equal(r["frame_sym"], arrays[f"c1_{mode}_frame_sym"])
    assert format_events(r["events"]) == m["events"]
    assert r["est_offset"] == m["est_offset"] == 1430.0
    assert r["final_freq_offset"] == m["final_freq_offset"]
    assert r["final_timing_freq"] == m["final_timing_freq"]
    assert r["final_state"] == m["final_state"]
    ch = arrays[f"c1_{mode}_chunks"]
    assert np.array_equal(r["chunks"][:, [0, 1, 3, 4]], ch[:, [0, 1, 3, 4]])


def test_config1_initial_offset_flag(oracle, golden, iq10):
    arrays, meta = golden
    r = oracle.receive(iq10, streaming=True, init_offset=1000.0)
    assert np.isnan(r["est_offset"])
    assert np.array_equal(r["soft"], arrays["c1_stream_o1000_soft"])
    assert np.array_equal(r["frames"], arrays["c1_stream_o1000_frames"])
    assert format_events(r["events"]) == meta["c1_stream_o1000"]["events"]


def test_100_frames_stream(oracle, golden, iq100):
    arrays, meta = golden
    r = oracle.receive(iq100, streaming=True, want_soft=False)
    assert np.array_equal(r["frames"], arrays["c100_stream_frames"])
    assert sha(r["frames"]) == meta["c100_stream"]["frames_sha256"]
    assert hashlib.sha256("\n".join(format_events(r["events"])).encode()).hexdigest() == \
        meta["c100_stream"]["events_sha256"]
    assert np.array_equal(r["frames"], oracle.bert_frames(100))


def test_frame_decoder_taps(oracle, golden):
    arrays, _ = golden
    for k in range(3):
        d = oracle.frame_decode(arrays["taps_payload_soft"][k])
        assert d["metric"] == arrays["taps_metric"][k]
        assert np.array_equal(d["deint"], arrays["taps_deint"][k])
        assert np.array_equal(d["bits"], arrays["taps_bits"][k])
        assert np.array_equal(d["frame"], arrays["taps_frames"][k])
        m, bits = oracle.viterbi(arrays["taps_deint"][k].astype(np.int32))
        assert m == arrays["taps_metric"][k] and np.array_equal(bits, arrays["taps_bits"][k])


def test_silent_frame_is_dropped(oracle):
    assert oracle.frame_decode(np.zeros(CODED_BITS))["metric"] == -1  # reference :859


@pytest.mark.parametrize("tag", ["p2000_12dB", "m2000_6dB", "p700_16dB", "p2000_clean"])
def test_noisy_configs(oracle, golden, iq100, tag):
    arrays, meta = golden
    m = meta["noisy_100"][tag]
    x = impair(iq100, **m["recipe"])
    assert sha(x) == m["input_sha256"], "impairment generator drifted (numpy RNG?)"
    r = oracle.receive(x, streaming=True)
    assert len(r["frames"]) == m["n_frames"]
    assert np.array_equal(r["frames"], arrays[f"n_{tag}_frames"])
    assert np.array_equal(r["metrics"], arrays[f"n_{tag}_metrics"])
    assert np.array_equal(r["frame_sym"], arrays[f"n_{tag}_frame_sym"])
    assert np.array_equal(r["soft"][::97], arrays[f"n_{tag}_soft_strided"])
    assert r["est_offset"] == m["est_offset"] and r["final_freq_offset"] == m["final_freq_offset"]
    assert hashlib.sha256("\n".join(format_events(r["events"])).encode()).hexdigest() == m["events_sha256"]


def test_short_and_ragged_inputs(oracle, iq10):
    # empty, sub-symbol, sub-chunk (tail path only, no offset search: reference :1088-1113)
    for n in (0, 1, 49, 51, 4000, 86719):
        r = oracle.receive(iq10[: 2 * n], streaming=True)
        assert np.isnan(r["est_offset"])
        assert len(r["frames"]) == 0
    r = oracle.receive(iq10[: 2 * 86720], streaming=True)
    assert r["est_offset"] == 1430.0 and len(r["chunks"]) == 2  # full chunk + 40-sample tail


# ------------------------------------------------------------------ live vs reference classes
@pytest.mark.skipif(not Reference.available(), reason="oracle/_ref/libopv_ref.so not built")
def test_live_against_reference_classes(oracle, iq10):
    ref = Reference()
    x = impair(iq10, amp=3000.0, f0_hz=-1234.0, ebn0_db=9.0, seed=3)
    for streaming in (True, False):
        a = oracle.receive(x, streaming=streaming)
        b = ref.receive(x, streaming=streaming)
        assert np.array_equal(a["soft"], b["soft"])
        assert np.array_equal(a["frames"], b["frames"])
        assert np.array_equal(a["frame_sym"], b["frame_sym"])
        assert format_events(a["events"]) == [ln for ln in b["log"].strip().split("\n") if ln]
    assert oracle.estimate_offset(x) == ref.estimate_offset(x)


# ------------------------------------------------------------------ coherent mode (-c), SURVEY §8f-4
@pytest.mark.parametrize("tag", ["clean", "p700_16dB_pll20"])
def test_coherent_mode_bit_exact(oracle, iq10, tag):
    """oracle's restatement of CoherentMSKDemodulator (reference :365-572) and of the -c batch driver
    (:1144-1161) against tests/golden/coherent.* (made by the compiled reference)."""
    import json
    from pathlib import Path
    g = Path(__file__).parent / "golden"
    meta = json.loads((g / "coherent.json").read_text())[tag]
    arrays = np.load(g / "coherent.npz")
    x = iq10 if tag == "clean" else impair(iq10, 2000.0, 700.0, 16.0, seed=11)
    assert sha(x) == meta["iq_sha256"]
    r = oracle.receive(x, streaming=False, coherent=True, pll_bw=meta["pll_bw"])
    assert np.array_equal(r["soft"], arrays[tag + "_soft"])          # fp64, bit for bit
    assert np.array_equal(r["frames"], arrays[tag + "_frames"]) and len(r["frames"]) == meta["n_frames"]
    assert np.array_equal(r["metrics"], arrays[tag + "_metrics"])
    assert np.array_equal(r["frame_sym"], arrays[tag + "_frame_sym"])
    assert r["est_offset"] == meta["est_offset"] and r["final_freq_offset"] == meta["final_freq_offset"]
    assert sha(r["frames"]) == meta["stdout_sha256"]
    ev = [ln for ln in meta["stderr_lines"] if ln.startswith("[")]
    assert format_events(r["events"]) == ev


def test_coherent_loop_is_chaotic(oracle, iq10):
    """Why §8f-4 has no GPU row: the reference's Costas loop never locks and its
    trajectory is chaotic - 1e-15 rad on the initial carrier phase grows to O(1) differences of the soft
    symbols within ~5 frames, so only arithmetic that is bit-identical to the reference's
    (libm sin/cos/atan2/hypot results included) can follow it. The same probe on the non-coherent
    demodulator stays at the 1e-13 level (its loops are contractive)."""
    est = oracle.estimate_offset(iq10)
    a, _ = oracle.coherent_demodulate(iq10, est)
    b, _ = oracle.coherent_demodulate(iq10, est, carrier_phase0=1e-15)
    e = np.abs(a - b) / np.mean(np.abs(a))
    assert e[:1000].max() < 1e-10
    assert e[-5000:].max() > 1e-2
    d1, d2 = oracle.new_demod(), oracle.new_demod()
    d1.freq_offset = d2.freq_offset = est
    d2.mu = 1e-15
    s1, s2 = oracle.demodulate(d1, iq10), oracle.demodulate(d2, iq10)
    assert len(s1) == len(s2) and np.max(np.abs(s1 - s2)) / np.mean(np.abs(s1)) < 1e-9


@pytest.mark.skipif(not Reference.available(), reason="oracle/_ref/libopv_ref.so not built")
def test_coherent_live_against_reference_class(oracle, iq10):
    x = impair(iq10, amp=5000.0, f0_hz=-300.0, ebn0_db=14.0, seed=5)
    a = oracle.receive(x, streaming=False, coherent=True, pll_bw=35.0, afc_alpha=0.002)
    b = Reference().receive(x, streaming=False, coherent=True, pll_bw=35.0, afc_alpha=0.002)
    assert np.array_equal(a["soft"], b["soft"]) and np.array_equal(a["frames"], b["frames"])
    assert a["final_freq_offset"] == b["final_freq_offset"]
