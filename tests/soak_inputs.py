"""Seeded input generators shared by the soak scripts (scripts/experiments/decoder_soak.py, offset_soak.py: tens of
thousands of cases, run by hand) and by the slices of them that run in every `pytest -m gpu` (tests/test_gpu_parity.py).
Test infrastructure: the inputs come from the CPU oracle's transmit chain and the numpy channel model."""
import numpy as np


def decoder_payloads(n, seed):
    """n x 2144 soft-symbol payloads for FrameDecoder::decode (ref src/opv-demod.cpp:854-898) of eight kinds: real coded
    frames at three noise levels, pure noise, few-level inputs full of trellis ties and quantiser boundaries, scales around
    the 1e-10 drop threshold, huge scales, sparse zeros."""
    from oracle_lib import Oracle
    o = Oracle()
    rng = np.random.default_rng(seed)
    soft = np.empty((n, 2144))
    for k in range(n):
        kind = k % 8
        if kind in (0, 1, 2):            # a real coded frame: bit 1 -> negative soft, plus noise of three strengths
            bits = o.encode_frame(rng.integers(0, 256, 134, dtype=np.uint8)).astype(np.float64)
            s = (1.0 - 2.0 * bits) * 2.4e11
            soft[k] = s + rng.standard_normal(2144) * 2.4e11 * (0.3, 0.8, 1.6)[kind]
        elif kind == 3:
            soft[k] = rng.standard_normal(2144) * 3e10
        elif kind == 4:                  # few levels: exact ties in the trellis and on quantiser boundaries
            soft[k] = rng.integers(-3, 4, 2144) * 1e10
        elif kind == 5:
            soft[k] = rng.standard_normal(2144) * 10.0 ** rng.uniform(-12, 3)      # around the 1e-10 drop threshold
        elif kind == 6:
            soft[k] = rng.standard_normal(2144) * 1e200
        else:
            s = rng.standard_normal(2144) * 1e11
            s[rng.random(2144) < rng.uniform(0.1, 0.99)] = 0.0
            soft[k] = s
    return soft


def oracle_decode_chunk(soft):
    from oracle_lib import Oracle
    o = Oracle()
    return [o.frame_decode(s) for s in soft]


def offset_openings(seed, n_streams=512):
    """n_streams random openings of a capture for estimate_offset (ref src/opv-demod.cpp:131-202): random start inside a BERT
    run, carrier offset in and beyond the +/-1530 Hz span, level, Eb/N0 from 0 dB to clean, lengths from 3000 to 45000 samples
    (the search uses min(N, 40000)), some pure noise and some digital silence with a burst."""
    from oracle_lib import Oracle, impair
    o = Oracle()
    rng = np.random.default_rng(seed)
    base = o.modulate(o.bert_frames(3, "K%d" % (seed % 1000), first=seed))
    caps = []
    for k in range(n_streams):
        n = int(rng.choice([3000, 8000, 20000, 39999, 40000, 40001, 45000]))
        at = int(rng.integers(0, base.size // 2 - n - 1))
        x = base[2 * at: 2 * (at + n)]
        kind = k % 10
        if kind == 8:                                            # noise only
            x = np.zeros_like(x)
            x = impair(x + 1, amp=float(rng.uniform(50, 3000)), ebn0_db=-20.0, seed=seed * 1000 + k)
        elif kind == 9:                                          # digital silence with a burst somewhere
            y = np.zeros_like(x)
            a, b = sorted(int(v) for v in rng.integers(0, n, 2))
            y[2 * a: 2 * b] = x[2 * a: 2 * b]
            x = y
        else:
            ebn0 = None if kind == 0 else float(rng.uniform(0, 25))
            x = impair(x, amp=float(rng.uniform(100, 16000)), f0_hz=float(rng.uniform(-2500, 2500)), ebn0_db=ebn0,
                       seed=seed * 1000 + k)
        caps.append(np.ascontiguousarray(x))
    return caps


def symmetric_openings(seed, n_streams=208):
    """n_streams captures whose offset-search landscape is EXACTLY symmetric, E(-o) = E(+o) (real-valued or purely imaginary
    samples): mirrored candidates tie, and in the reference they tie EXACTLY - its LO increments for -o and +o are exact
    negatives, so c2(-o) = conj(c1(+o)) bit for bit and the strict '>' (src/opv-demod.cpp:161,195) keeps the first, -1500 -
    while a one-pass evaluation sees two energies 1e-13 apart in either order. The landscape of such a capture is a
    smooth even function (every feature is a 54 kHz wide window lobe), so it peaks at o = 0 (no tie) or is convex and the two
    EDGE candidates -1500 / +1500 tie, the coarse winner then dragging the fine pass to -1530 or +1530: real tones outside the
    tone pair (convex), pairs of them, with and without real noise, real noise alone, and - mostly peaked at 0, the controls - a
    tone near the pair or one branch of an MSK capture with a carrier offset; random levels and lengths (the search uses
    min(N, 40 000) samples)."""
    from oracle_lib import Oracle, impair
    o = Oracle()
    rng = np.random.default_rng(seed)
    base = o.modulate(o.bert_frames(2, "T%d" % (seed % 1000), first=seed))
    caps = []
    for k in range(n_streams):
        n = int(rng.choice([4000, 12000, 24000, 39999, 40000, 40040, 44000]))
        t = np.arange(n)
        kind = k % 8
        amp = float(rng.uniform(200, 15000))
        if kind in (0, 1):                                       # a real tone outside the pair: convex landscape, the edges tie
            f = float(rng.uniform(30000, 60000))
            v = amp * np.cos(2 * np.pi * f * t / 2168000.0 + rng.uniform(0, 6.28))
        elif kind == 2:                                          # two of them plus real noise
            v = amp * np.cos(2 * np.pi * rng.uniform(30000, 60000) * t / 2168000.0 + rng.uniform(0, 6.28)) \
                + 0.5 * amp * np.cos(2 * np.pi * rng.uniform(30000, 60000) * t / 2168000.0 + 1.0) + 0.05 * amp * rng.standard_normal(n)
        elif kind == 3:                                          # a tone near the pair (peak at 0) against one outside, plus real noise
            v = rng.uniform(0, 0.3) * amp * np.cos(2 * np.pi * (13550.0 + rng.uniform(-1400, 1400)) * t / 2168000.0) \
                + amp * np.cos(2 * np.pi * rng.uniform(30000, 60000) * t / 2168000.0 + 1.0) + 0.05 * amp * rng.standard_normal(n)
        elif kind == 4:                                          # real noise only
            v = rng.standard_normal(n) * amp * 0.2
        else:                                                    # one branch of an MSK capture with a carrier offset
            at = int(rng.integers(0, base.size // 2 - n - 1))
            x = impair(base[2 * at: 2 * (at + n)], amp=amp, f0_hz=float(rng.uniform(-1500, 1500)),
                       ebn0_db=None if kind == 5 else float(rng.uniform(3, 25)), seed=seed * 1000 + k)
            v = x[(kind & 1)::2].astype(np.float64)              # kind 5, 7: Q branch; kind 6: I branch
        x = np.zeros(2 * n, np.int16)
        x[(k // 8) % 2::2] = np.clip(np.rint(v), -32768, 32767)  # real-valued (Q = 0) or purely imaginary (I = 0)
        caps.append(x)
    return caps


def weak_correlation_openings(seed, n_streams=96):
    """n_streams captures whose correlation with the tone pair is WEAK against their power - the inputs on which the errors of an
    offset-search energy are large relative to the energy itself (they scale with sqrt(energy x power), k_offset_search.hip): a
    complex tone far outside the pair with noise, a strong interferer over a faint MSK signal, a few LSB of noise on a DC offset,
    and - the ones that put two candidates within rounding of each other without an exact mirror - a real tone whose other
    branch carries one LSB of noise. Lengths around the 40 000 samples the search uses."""
    from oracle_lib import Oracle, impair
    o = Oracle()
    rng = np.random.default_rng(seed)
    base = o.modulate(o.bert_frames(2, "W%d" % (seed % 1000), first=seed))
    caps = []
    for k in range(n_streams):
        n = int(rng.choice([12000, 39999, 40000, 40040, 44000]))
        t = np.arange(n)
        kind = k % 4
        amp = float(rng.uniform(500, 15000))
        if kind == 0:                                            # complex tone on a null of one tone's 40-sample window (a sidelobe of the other's) + complex noise
            f = float(rng.choice([-1, 1]) * (13550.0 + 54200.0 * int(rng.integers(1, 18)) + rng.uniform(-800, 800)))
            v = amp * np.exp(2j * np.pi * f * t / 2168000.0 + 1j * rng.uniform(0, 6.28)) \
                + 10.0 ** rng.uniform(-3, -1) * amp * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        elif kind == 1:                                          # strong interferer over a faint MSK signal
            at = int(rng.integers(0, base.size // 2 - n - 1))
            x = impair(base[2 * at: 2 * (at + n)], amp=float(rng.uniform(20, 300)), f0_hz=float(rng.uniform(-1400, 1400)), ebn0_db=None, seed=seed * 1000 + k)
            v = x[0::2] + 1j * x[1::2] + amp * np.exp(2j * np.pi * float(rng.uniform(40000, 300000)) * t / 2168000.0)
        elif kind == 2:                                          # a few LSB of noise on a DC offset
            v = amp * np.exp(1j * rng.uniform(0, 6.28)) + rng.uniform(0.5, 3.0) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        else:                                                    # a real tone, one LSB of noise on the other branch: near, not exact, mirror ties
            f = float(rng.uniform(30000, 60000))
            v = amp * np.cos(2 * np.pi * f * t / 2168000.0 + rng.uniform(0, 6.28)) + 1j * rng.uniform(0.3, 1.0) * rng.standard_normal(n)
        x = np.zeros(2 * n, np.int16)
        x[0::2] = np.clip(np.rint(v.real), -32768, 32767)
        x[1::2] = np.clip(np.rint(v.imag), -32768, 32767)
        caps.append(x)
    return caps


def offset_opening(seed, k):
    """ONE opening for estimate_offset, a function of (seed, k) alone (so that single finds of scripts/experiments/near_tie_hunt.py
    can be regenerated): a random start inside a 3-frame BERT run, carrier offset within and beyond the search span, level,
    Eb/N0 3 .. 22 dB, 20 000 .. 45 000 samples."""
    from oracle_lib import Oracle, impair
    global _OPENING_BASE
    try:
        base = _OPENING_BASE
    except NameError:
        o = Oracle()
        base = _OPENING_BASE = o.modulate(o.bert_frames(3, "NT", first=4242))
    rng = np.random.default_rng([int(seed), int(k)])
    n = int(rng.choice([20000, 30000, 39999, 40000, 40001, 45000]))
    at = int(rng.integers(0, base.size // 2 - n - 1))
    x = impair(base[2 * at: 2 * (at + n)], amp=float(rng.uniform(300, 12000)), f0_hz=float(rng.uniform(-2200, 2200)),
               ebn0_db=float(rng.uniform(3, 22)), seed=int(rng.integers(1, 2 ** 31)))
    return np.ascontiguousarray(x)


def near_tie_class(e, rel=1e-11):
    """Does the strict-'>' scan of estimate_offset over these 134 energies (121 coarse, then 13 fine around the coarse winner) meet
    a NEAR tie - a contender within `rel` of the best energy in play that is not equal to it? Returns "coarse", "fine" or ""."""
    e = np.asarray(e)
    c = e[:121]
    top = c.max()
    near = c[(c >= top * (1 - rel)) & (c != top)]
    if near.size:
        return "coarse"
    best = int(np.argmax(c))                       # first maximum
    f = np.delete(e[121:134], 6)                   # (entry 6 of the fine pass repeats the coarse winner's offset)
    top2 = max(top, f.max())
    pool = np.concatenate([[top], f])
    near = pool[(pool >= top2 * (1 - rel)) & (pool != top2)]
    return "fine" if near.size else ""


def oracle_offset_chunk(caps):
    from oracle_lib import Oracle
    o = Oracle()
    return [o.estimate_offset(c) for c in caps]


def oracle_offset_energies_chunk(caps):
    from oracle_lib import Oracle
    o = Oracle()
    return [o.estimate_offset(c, energies=True) for c in caps]


def oracle_receive_job(x, want_soft=False):
    """one stream through the oracle's whole receive chain (-s semantics); the soft log (8 B per symbol) only on request"""
    from oracle_lib import Oracle
    e = Oracle().receive(x, streaming=True, want_soft=want_soft)
    keys = ("frames", "metrics", "frame_sym", "events", "n_soft", "est_offset", "final_freq_offset") + (("soft",) if want_soft else ())
    return {k: e[k] for k in keys}


def host_workers(cap=16):
    """worker processes for the oracle side: the cores this job may use (a one-GPU box of the pool gives 16)"""
    import os
    return max(1, min(len(os.sched_getaffinity(0)), cap))
