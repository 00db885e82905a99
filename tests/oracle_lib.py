"""ctypes bindings for the TEST-ONLY checkers.

* ``oracle``  -> oracle/_build/libopv_oracle.so  (our plain-C restatement, always available:
                 built on demand with gcc)
* ``ref``     -> oracle/_ref/libopv_ref.so       (window onto the compiled reference classes;
                 present only where oracle/Makefile could see /root/reference, or where the
                 prebuilt file travelled with the snapshot)

Nothing in the product imports this module.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"

SPS = 40
FRAME_BYTES = 134
FRAME_BITS = 1072
CODED_BITS = 2144
FRAME_SYMBOLS = 2168
CHUNK_SAMPLES = 86720

EVENT_DTYPE = np.dtype(
    [("kind", "<i4"), ("count", "<i4"), ("sym_idx", "<u8"), ("corr", "<f8"), ("raw", "<f8")], align=True
)
EV_NAMES = {1: "HUNT_TO_VERIFY", 2: "VERIFY_TO_LOCK", 3: "SYNC_OK", 4: "SYNC_MISS", 5: "LOST_LOCK"}


def build_oracle():
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR), "oracle"], check=True)
    return ORACLE_DIR / "_build" / "libopv_oracle.so"


class _RxCfg(C.Structure):
    _fields_ = [("streaming", C.c_int), ("have_init_offset", C.c_int), ("init_offset", C.c_double),
                ("afc_alpha", C.c_double), ("coherent", C.c_int), ("pll_bw", C.c_double)]


class _RxOut(C.Structure):
    _fields_ = [
        ("frames", C.c_void_p), ("metrics", C.c_void_p), ("quality", C.c_void_p), ("frame_sym", C.c_void_p),
        ("cap_frames", C.c_size_t),
        ("soft", C.c_void_p), ("cap_soft", C.c_size_t),
        ("events", C.c_void_p), ("cap_events", C.c_size_t),
        ("chunk_state", C.c_void_p), ("cap_chunks", C.c_size_t),
        ("n_frames", C.c_size_t), ("n_perfect", C.c_size_t), ("n_soft", C.c_size_t), ("n_events", C.c_size_t),
        ("n_chunks", C.c_size_t),
        ("est_offset", C.c_double), ("final_freq_offset", C.c_double), ("final_timing_freq", C.c_double),
        ("final_state", C.c_int),
    ]


class _Demod(C.Structure):
    _fields_ = [("freq_offset", C.c_double), ("phase_f1", C.c_double), ("phase_f2", C.c_double),
                ("prev1_re", C.c_double), ("prev1_im", C.c_double), ("prev2_re", C.c_double),
                ("prev2_im", C.c_double), ("afc_alpha", C.c_double), ("mu", C.c_double),
                ("timing_freq", C.c_double), ("alpha_timing", C.c_double), ("beta_timing", C.c_double),
                ("leftover", C.c_size_t)]


class _Coh(C.Structure):
    _fields_ = [("freq_offset", C.c_double), ("carrier_phase", C.c_double), ("phase_f1", C.c_double),
                ("phase_f2", C.c_double), ("loop_freq", C.c_double), ("prev_re", C.c_double),
                ("prev_im", C.c_double), ("afc_alpha", C.c_double), ("pll_alpha", C.c_double),
                ("pll_beta", C.c_double)]


def _iq(a):
    a = np.ascontiguousarray(a, dtype=np.int16).reshape(-1)
    assert a.size % 2 == 0
    return a


class Oracle:
    """The plain-C restatement (oracle/opv_oracle.c)."""

    def __init__(self):
        self.lib = C.CDLL(str(build_oracle()))
        L = self.lib
        L.oro_modulated_len.restype = C.c_size_t
        L.oro_modulated_len.argtypes = [C.c_size_t]
        L.oro_modulate_frames.restype = C.c_size_t
        L.oro_modulate_frames.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.oro_bert_frame.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oro_encode_frame.argtypes = [C.c_void_p, C.c_void_p]
        L.oro_lfsr_table.argtypes = [C.c_void_p]
        L.oro_base40_encode.argtypes = [C.c_char_p, C.c_void_p]
        L.oro_estimate_offset.restype = C.c_double
        L.oro_estimate_offset.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.oro_demod_init.argtypes = [C.c_void_p]
        L.oro_demodulate.restype = C.c_size_t
        L.oro_demodulate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.oro_deinterleave_addr.restype = C.c_size_t
        L.oro_deinterleave_addr.argtypes = [C.c_size_t]
        L.oro_viterbi.restype = C.c_int
        L.oro_viterbi.argtypes = [C.c_void_p, C.c_void_p]
        L.oro_frame_decode.restype = C.c_int
        L.oro_frame_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oro_receive.restype = C.c_int
        L.oro_receive.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]

    # ---- transmit ----
    def bert_frames(self, n, callsign="W5NYV", token=0xBBAADD, first=0):
        out = np.zeros((n, FRAME_BYTES), np.uint8)
        for k in range(n):
            self.lib.oro_bert_frame(callsign.encode(), token, first + k, out[k].ctypes.data)
        return out

    def modulate(self, frames):
        frames = np.ascontiguousarray(frames, np.uint8).reshape(-1, FRAME_BYTES)
        n = self.lib.oro_modulated_len(len(frames))
        iq = np.empty(2 * n, np.int16)
        w = self.lib.oro_modulate_frames(frames.ctypes.data, len(frames), iq.ctypes.data)
        assert w == n
        return iq

    def encode_frame(self, payload):
        payload = np.ascontiguousarray(payload, np.uint8)
        out = np.zeros(CODED_BITS, np.uint8)
        self.lib.oro_encode_frame(payload.ctypes.data, out.ctypes.data)
        return out

    def lfsr_table(self):
        out = np.zeros(FRAME_BYTES, np.uint8)
        self.lib.oro_lfsr_table(out.ctypes.data)
        return out

    def base40(self, callsign):
        out = np.zeros(6, np.uint8)
        self.lib.oro_base40_encode(callsign.encode(), out.ctypes.data)
        return out

    # ---- receive pieces ----
    def estimate_offset(self, iq, energies=False):
        iq = _iq(iq)
        e = np.zeros(134, np.float64)
        r = self.lib.oro_estimate_offset(iq.ctypes.data, iq.size // 2, e.ctypes.data)
        return (r, e) if energies else r

    def new_demod(self):
        d = _Demod()
        self.lib.oro_demod_init(C.byref(d))
        return d

    def demodulate(self, d, iq):
        iq = _iq(iq)
        n = iq.size // 2
        soft = np.empty(n // 38 + 8, np.float64)
        ns = self.lib.oro_demodulate(C.byref(d), iq.ctypes.data, n, soft.ctypes.data, soft.size)
        return soft[:ns].copy()

    def coherent_demodulate(self, iq, freq_offset, afc_alpha=0.001, pll_bw=50.0, extra=False, carrier_phase0=0.0):
        """CoherentMSKDemodulator::demodulate from a fresh object (ref:455-543)."""
        iq = _iq(iq)
        n = iq.size // 2
        d = _Coh()
        self.lib.oro_coh_init(C.byref(d))
        d.freq_offset = freq_offset
        d.afc_alpha = afc_alpha
        d.carrier_phase = carrier_phase0
        self.lib.oro_coh_set_pll_bandwidth(C.byref(d), C.c_double(pll_bw))
        soft = np.empty(n // SPS + 1, np.float64)
        ex = np.zeros((n // SPS + 1, 3), np.float64)
        self.lib.oro_coh_demodulate.restype = C.c_size_t
        ns = self.lib.oro_coh_demodulate(C.byref(d), C.c_void_p(iq.ctypes.data), C.c_size_t(n),
                                         C.c_void_p(soft.ctypes.data), C.c_size_t(soft.size),
                                         C.c_void_p(ex.ctypes.data), C.c_size_t(len(ex) if extra else 0))
        return (soft[:ns].copy(), d, ex[:ns].copy()) if extra else (soft[:ns].copy(), d)

    def deinterleave_perm(self):
        return np.array([self.lib.oro_deinterleave_addr(i) for i in range(CODED_BITS)], np.uint16)

    def viterbi(self, q):
        q = np.ascontiguousarray(q, np.int32)
        bits = np.zeros(FRAME_BITS, np.uint8)
        m = self.lib.oro_viterbi(q.ctypes.data, bits.ctypes.data)
        return m, bits

    def frame_decode(self, soft):
        soft = np.ascontiguousarray(soft, np.float64)
        assert soft.size == CODED_BITS
        out = np.zeros(FRAME_BYTES, np.uint8)
        q = np.zeros(CODED_BITS, np.int32)
        de = np.zeros(CODED_BITS, np.int32)
        bits = np.zeros(FRAME_BITS, np.uint8)
        m = self.lib.oro_frame_decode(soft.ctypes.data, out.ctypes.data, q.ctypes.data, de.ctypes.data,
                                      bits.ctypes.data)
        return dict(metric=m, frame=out, q=q, deint=de, bits=bits)

    # ---- whole receiver ----
    def receive(self, iq, streaming=True, init_offset=None, afc_alpha=0.001, want_soft=True,
                coherent=False, pll_bw=50.0):
        iq = _iq(iq)
        n = iq.size // 2
        cap_frames = n // (FRAME_SYMBOLS * 38) + 8
        cap_soft = (n // 38 + 64) if want_soft else 0
        frames = np.zeros((cap_frames, FRAME_BYTES), np.uint8)
        metrics = np.zeros(cap_frames, np.int32)
        quality = np.zeros(cap_frames, np.float64)
        fsym = np.zeros(cap_frames, np.uint64)
        soft = np.zeros(max(cap_soft, 1), np.float64)
        events = np.zeros(4 * cap_frames + 64, EVENT_DTYPE)
        cap_chunks = n // 80000 + 4
        chunks = np.zeros((cap_chunks, 5), np.float64)
        cfg = _RxCfg(int(streaming), int(init_offset is not None), float(init_offset or 0.0), afc_alpha,
                     int(coherent), float(pll_bw))
        out = _RxOut()
        out.frames, out.metrics, out.quality, out.frame_sym = (frames.ctypes.data, metrics.ctypes.data,
                                                               quality.ctypes.data, fsym.ctypes.data)
        out.cap_frames = cap_frames
        out.soft = soft.ctypes.data if want_soft else None
        out.cap_soft = cap_soft
        out.events, out.cap_events = events.ctypes.data, events.size
        out.chunk_state, out.cap_chunks = chunks.ctypes.data, cap_chunks
        rc = self.lib.oro_receive(iq.ctypes.data, n, C.byref(cfg), C.byref(out))
        assert rc == 0
        nf = out.n_frames
        assert nf <= cap_frames and out.n_events <= events.size and out.n_chunks <= cap_chunks
        return dict(
            frames=frames[:nf].copy(), metrics=metrics[:nf].copy(), quality=quality[:nf].copy(),
            frame_sym=fsym[:nf].copy(), n_perfect=out.n_perfect,
            soft=soft[:out.n_soft].copy() if want_soft else None, n_soft=out.n_soft,
            events=events[:out.n_events].copy(), chunks=chunks[:out.n_chunks].copy(),
            est_offset=out.est_offset, final_freq_offset=out.final_freq_offset,
            final_timing_freq=out.final_timing_freq, final_state=out.final_state,
        )


def format_events(events):
    """Render tracker events exactly as SyncTracker::process prints them
    (reference src/opv-demod.cpp:651,677,695,699,705)."""
    lines = []
    for e in events:
        k, idx = int(e["kind"]), int(e["sym_idx"])
        if k == 1:
            lines.append("[%d] HUNTING→VERIFYING (corr=%.3f, raw=%.0f)" % (idx, e["corr"], e["raw"]))
        elif k == 2:
            lines.append("[%d] VERIFYING→LOCKED (frame %d)" % (idx, e["count"]))
        elif k == 3:
            lines.append("[%d] LOCKED: sync OK (corr=%.3f)" % (idx, e["corr"]))
        elif k == 4:
            lines.append("[%d] LOCKED: sync MISS #%d (corr=%.3f)" % (idx, e["count"], e["corr"]))
        elif k == 5:
            lines.append("[%d] LOCKED→HUNTING (lost lock)" % idx)
    return lines


# ---------------------------------------------------------------------------------------
class Reference:
    """Window onto the compiled reference classes (oracle/_ref/libopv_ref.so)."""

    @staticmethod
    def available():
        return (ORACLE_DIR / "_ref" / "libopv_ref.so").exists()

    def __init__(self):
        self.lib = C.CDLL(str(ORACLE_DIR / "_ref" / "libopv_ref.so"))
        L = self.lib
        for f in ("ref_demod_create", "ref_tracker_create", "ref_coh_create"):
            getattr(L, f).restype = C.c_void_p
        for f in ("ref_demod_freq_offset", "ref_demod_timing_freq", "ref_demod_estimate_offset",
                  "ref_coh_freq_offset", "ref_coh_estimate_offset"):
            getattr(L, f).restype = C.c_double
        for f in ("ref_demod_leftover", "ref_demod_demodulate", "ref_deinterleave_addr", "ref_log_take",
                  "ref_coh_demodulate"):
            getattr(L, f).restype = C.c_size_t
        L.ref_demod_destroy.argtypes = [C.c_void_p]
        L.ref_demod_set_freq_offset.argtypes = [C.c_void_p, C.c_double]
        L.ref_demod_set_afc.argtypes = [C.c_void_p, C.c_double]
        L.ref_demod_freq_offset.argtypes = [C.c_void_p]
        L.ref_demod_timing_freq.argtypes = [C.c_void_p]
        L.ref_demod_leftover.argtypes = [C.c_void_p]
        L.ref_demod_estimate_offset.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ref_demod_demodulate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ref_tracker_destroy.argtypes = [C.c_void_p]
        L.ref_coh_destroy.argtypes = [C.c_void_p]
        L.ref_coh_set_freq_offset.argtypes = [C.c_void_p, C.c_double]
        L.ref_coh_set_afc.argtypes = [C.c_void_p, C.c_double]
        L.ref_coh_set_pll.argtypes = [C.c_void_p, C.c_double]
        L.ref_coh_freq_offset.argtypes = [C.c_void_p]
        L.ref_coh_estimate_offset.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ref_coh_demodulate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ref_tracker_state.argtypes = [C.c_void_p]
        L.ref_tracker_process.argtypes = [C.c_void_p, C.c_double, C.c_size_t, C.c_void_p, C.c_void_p]
        L.ref_deinterleave_addr.argtypes = [C.c_size_t]
        L.ref_viterbi.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_frame_decode.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_log_take.argtypes = [C.c_void_p, C.c_size_t]

    def take_log(self):
        buf = C.create_string_buffer(1 << 22)
        n = self.lib.ref_log_take(buf, len(buf))
        return buf.raw[:min(n, len(buf))].decode("utf-8")

    def estimate_offset(self, iq):
        iq = _iq(iq)
        h = self.lib.ref_demod_create()
        r = self.lib.ref_demod_estimate_offset(h, iq.ctypes.data, iq.size // 2)
        self.lib.ref_demod_destroy(h)
        return r

    def frame_decode(self, soft):
        soft = np.ascontiguousarray(soft, np.float64)
        out = np.zeros(FRAME_BYTES, np.uint8)
        m = self.lib.ref_frame_decode(soft.ctypes.data, out.ctypes.data)
        return m, out

    def viterbi(self, q):
        q = np.ascontiguousarray(q, np.int32)
        bits = np.zeros(FRAME_BITS, np.uint8)
        m = self.lib.ref_viterbi(q.ctypes.data, bits.ctypes.data)
        return m, bits

    def deinterleave_perm(self):
        return np.array([self.lib.ref_deinterleave_addr(i) for i in range(CODED_BITS)], np.uint16)

    def receive(self, iq, streaming=True, init_offset=None, afc_alpha=0.001, coherent=False, pll_bw=50.0):
        """Drive the reference classes the way the reference's main() does
        (streaming src/opv-demod.cpp:995-1113, batch :1132-1206, -c :1144-1161)."""
        iq = _iq(iq)
        n = iq.size // 2
        L = self.lib
        self.take_log()
        dm = L.ref_demod_create()
        tr = L.ref_tracker_create()
        softs, frames, metrics, quality, fsym, chunks = [], [], [], [], [], []
        est = float("nan")
        payload = np.zeros(CODED_BITS, np.float64)
        q = C.c_double(0)
        total = 0

        def feed(s):
            nonlocal total
            for i, v in enumerate(s):
                if L.ref_tracker_process(tr, float(v), total + i, payload.ctypes.data, C.byref(q)):
                    m, fr = self.frame_decode(payload)
                    if m >= 0:
                        frames.append(fr)
                        metrics.append(m)
                        quality.append(q.value)
                        fsym.append(total + i)
            total += len(s)

        def demod(view):
            buf = np.empty(view.size // 2 // 38 + 8, np.float64)
            ns = L.ref_demod_demodulate(dm, view.ctypes.data, view.size // 2, buf.ctypes.data, buf.size)
            chunks.append([L.ref_demod_freq_offset(dm), L.ref_demod_timing_freq(dm), float("nan"),
                           float(L.ref_demod_leftover(dm)), float(ns)])
            return buf[:ns].copy()

        if streaming:
            if init_offset is not None:
                L.ref_demod_set_freq_offset(dm, float(init_offset))
            L.ref_demod_set_afc(dm, afc_alpha)
            start, first = 0, True
            while n - start >= CHUNK_SAMPLES:
                view = iq[2 * start: 2 * (start + CHUNK_SAMPLES)]
                if first:
                    if init_offset is None:
                        est = L.ref_demod_estimate_offset(dm, view.ctypes.data, CHUNK_SAMPLES)
                        L.ref_demod_set_freq_offset(dm, est)
                    first = False
                s = demod(view)
                softs.append(s)
                feed(s)
                lo = L.ref_demod_leftover(dm)
                start += CHUNK_SAMPLES - lo if 0 < lo < CHUNK_SAMPLES else CHUNK_SAMPLES
            if n > start:
                s = demod(iq[2 * start:])
                softs.append(s)
                feed(s)
        elif coherent:
            cd = L.ref_coh_create()
            est = L.ref_coh_estimate_offset(cd, iq.ctypes.data, n)
            L.ref_coh_set_freq_offset(cd, est)
            L.ref_coh_set_afc(cd, afc_alpha)
            L.ref_coh_set_pll(cd, pll_bw)
            buf = np.empty(n // SPS + 1, np.float64)
            ns = L.ref_coh_demodulate(cd, iq.ctypes.data, n, buf.ctypes.data, buf.size)
            L.ref_demod_set_freq_offset(dm, L.ref_coh_freq_offset(cd))  # so final_freq_offset below reports it
            chunks.append([L.ref_coh_freq_offset(cd), 0.0, float("nan"), 0.0, float(ns)])
            L.ref_coh_destroy(cd)
            s = buf[:ns].copy()
            softs.append(s)
            feed(s)
        else:
            est = L.ref_demod_estimate_offset(dm, iq.ctypes.data, n)
            L.ref_demod_set_freq_offset(dm, est)
            L.ref_demod_set_afc(dm, afc_alpha)
            s = demod(iq)
            softs.append(s)
            feed(s)
        res = dict(
            frames=np.array(frames, np.uint8).reshape(-1, FRAME_BYTES), metrics=np.array(metrics, np.int32),
            quality=np.array(quality), frame_sym=np.array(fsym, np.uint64),
            soft=np.concatenate(softs) if softs else np.zeros(0), chunks=np.array(chunks).reshape(-1, 5),
            est_offset=est, final_freq_offset=L.ref_demod_freq_offset(dm),
            final_timing_freq=L.ref_demod_timing_freq(dm), final_state=L.ref_tracker_state(tr),
            log=self.take_log(),
        )
        L.ref_demod_destroy(dm)
        L.ref_tracker_destroy(tr)
        return res


def ref_binary(name):
    p = ORACLE_DIR / "_ref" / name
    return p if p.exists() else None


# ---------------------------------------------------------------------------------------
def impair(iq, amp=2000.0, f0_hz=0.0, ebn0_db=None, seed=1, full_scale=16383.0):
    """Seeded channel model for the noisy configurations: rescale to
    amplitude `amp`, rotate by exp(j 2 pi f0 n / Fs), add complex AWGN with total variance
    80*amp^2 / 10^(EbN0/10) (Eb = 2 Es, Es = 40 amp^2), round to nearest, clip to int16.
    The reference has no channel model; this tool DEFINES the inputs of those configs."""
    iq = _iq(iq)
    z = (iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)) * (amp / full_scale)
    n = np.arange(z.size, dtype=np.float64)
    if f0_hz:
        z = z * np.exp(2j * np.pi * f0_hz * n / 2168000.0)
    if ebn0_db is not None:
        rng = np.random.default_rng(seed)
        var = 80.0 * amp * amp / (10.0 ** (ebn0_db / 10.0))
        sd = np.sqrt(var / 2.0)
        z = z + sd * (rng.standard_normal(z.size) + 1j * rng.standard_normal(z.size))
    out = np.empty(2 * z.size, np.int16)
    out[0::2] = np.clip(np.rint(z.real), -32768, 32767).astype(np.int16)
    out[1::2] = np.clip(np.rint(z.imag), -32768, 32767).astype(np.int16)
    return out


def accidents(iq, rng, amp, n_max=4):
    """What a live channel does to a capture besides noise (a test INPUT generator, like impair): 1..n_max of - samples
    dropped or repeated (the timing loop and the tracker's flywheel see a jump), a burst of strong noise, a deep fade, a
    step of the carrier, a stretch of weak noise without a signal (never an exact zero run: that class is covered by the
    silence tests). Returns (int16 IQ, a note naming the accidents)."""
    iq = _iq(iq)
    z = iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)
    note = []
    for _ in range(int(rng.integers(1, n_max + 1))):
        kind = str(rng.choice(["drop", "dup", "burst", "fade", "fstep", "noisegap"]))
        at = int(rng.integers(1000, z.size - 1000))
        if kind == "drop":
            m = int(rng.integers(1, 3000))
            z = np.concatenate([z[:at], z[at + m:]])
        elif kind == "dup":
            m = int(rng.integers(1, 3000))
            z = np.concatenate([z[:at], z[at - min(m, at):at], z[at:]])
        elif kind == "burst":
            e = min(z.size, at + int(rng.integers(1000, 200000)))
            z[at:e] += amp * 30 * (rng.standard_normal(e - at) + 1j * rng.standard_normal(e - at))
        elif kind == "fade":
            e = min(z.size, at + int(rng.integers(1000, 300000)))
            z[at:e] *= 0.03
        elif kind == "fstep":
            df = float(rng.uniform(-500, 500))
            z[at:] *= np.exp(2j * np.pi * df * np.arange(z.size - at) / 2168000.0)
        else:
            e = min(z.size, at + int(rng.integers(1000, 300000)))
            z[at:e] = 3.0 * (rng.standard_normal(e - at) + 1j * rng.standard_normal(e - at)) + (0.4 + 0.3j)
        note.append(kind)
    out = np.empty(2 * z.size, np.int16)
    out[0::2] = np.clip(np.rint(z.real), -32768, 32767).astype(np.int16)
    out[1::2] = np.clip(np.rint(z.imag), -32768, 32767).astype(np.int16)
    out[0::2][(out[0::2] == 0) & (out[1::2] == 0)] = 1     # (a fade over a stretch of weak noise rounds to exact zeros otherwise)
    return out, "+".join(note)


def resample_clock(iq, ppm):
    """Sample-clock error for the timing loop: the capture as an ADC running `ppm`
    parts per million fast would have taken it (linear interpolation, round to nearest). A test
    INPUT generator - any int16 stream is a valid input for the parity tests."""
    iq = _iq(iq)
    z = iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)
    n_out = int(z.size / (1.0 + ppm * 1e-6))
    t = np.arange(n_out, dtype=np.float64) * (1.0 + ppm * 1e-6)
    i = np.minimum(t.astype(np.int64), z.size - 2)
    f = t - i
    y = z[i] * (1.0 - f) + z[i + 1] * f
    out = np.empty(2 * n_out, np.int16)
    out[0::2] = np.clip(np.rint(y.real), -32768, 32767).astype(np.int16)
    out[1::2] = np.clip(np.rint(y.imag), -32768, 32767).astype(np.int16)
    return out


def channel_model(iq, gain=1.0, f0_hz=0.0, sigma=0.0, seed=0):
    """CPU model of the product's device channel tool (csrc/k_channel.hip), operation for
    operation: rotation by exp(j 2 pi f0 n / Fs) with the phase reduced in cycles, gain, counter-based AWGN
    (splitmix64 finaliser keyed by seed ^ n * 0xD1342543DE82EF95 -> two 24-bit uniforms -> float32 Box-Muller),
    round to nearest even, clip. The integer part (hash, uniforms) is exact; the transcendental part uses numpy's
    float32 log / sincos and float64 sincos, which may differ from the device's by an ulp, i.e. the int16 result
    may differ by one LSB where a value sits within ~1e-4 LSB of a rounding boundary (the GPU test counts them).
    This pins what bench.py's and the BER curve's inputs ARE, off-device."""
    iq = _iq(iq)
    n = np.arange(iq.size // 2, dtype=np.uint64)
    t = (f0_hz / 2168000.0) * n.astype(np.float64)
    t = t - np.rint(t)
    sn, cs = np.sin(2.0 * np.pi * t), np.cos(2.0 * np.pi * t)
    # exact values at the points where sincospi is exact (the reduced phase is a multiple of 1/4 cycle)
    q = np.rint(t * 4.0)
    on = (t * 4.0 == q)
    if on.any():
        qi = q[on].astype(np.int64) % 4
        sn[on] = np.array([0.0, 1.0, 0.0, -1.0])[qi]
        cs[on] = np.array([1.0, 0.0, -1.0, 0.0])[qi]
    xr = gain * iq[0::2].astype(np.float64)
    xi = gain * iq[1::2].astype(np.float64)
    yr = xr * cs - xi * sn
    yi = xr * sn + xi * cs
    if sigma > 0.0:
        with np.errstate(over="ignore"):
            z = (np.uint64(seed) ^ (n * np.uint64(0xD1342543DE82EF95))) + np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            h = z ^ (z >> np.uint64(31))
        f32 = np.float32
        u1 = ((h >> np.uint64(40)).astype(np.uint32).astype(f32) + f32(0.5)) * f32(1.0 / 16777216.0)
        u2 = (((h >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.uint32).astype(f32) + f32(0.5)) * f32(1.0 / 16777216.0)
        rad = np.sqrt(f32(-2.0) * np.log(u1))
        a = (f32(2.0) * u2).astype(np.float64) * np.pi        # sincospif(2 u2): evaluated in double, rounded to float32
        s2, c2 = np.sin(a).astype(f32), np.cos(a).astype(f32)
        yr = yr + sigma * (rad * c2).astype(np.float64)
        yi = yi + sigma * (rad * s2).astype(np.float64)
    out = np.empty(iq.size, np.int16)
    out[0::2] = np.clip(np.rint(yr), -32768, 32767).astype(np.int16)
    out[1::2] = np.clip(np.rint(yi), -32768, 32767).astype(np.int16)
    return out


# ---------------------------------------------------------------------------------------
def run_under_reference_modem(child, iq, piece=16384, recv_timeout=60.0, verbose_child=False):
    """The boundary's real caller: the REFERENCE's `opv-modem -R -d <child>` (oracle/_ref/opv-modem, compiled from
    src/opv-modem.cpp) forks, execs `<child> -s -r` on pipes (:696-717), forwards its stdin in 16 KB reads (:734,753),
    reads exactly-134-byte records from the child non-blocking (:765-786) and sends each as one UDP datagram to
    127.0.0.1:<port> (-r). `iq` is written to the parent's stdin in pieces of <= `piece` bytes, then stdin is closed.
    Returns (list of datagrams in arrival order, parent's exit status, parent's stderr text)."""
    import select
    import socket
    import subprocess
    import time
    modem = ref_binary("opv-modem")
    assert modem is not None, "oracle/_ref/opv-modem is missing (make -C oracle ref, in the build container)"
    rx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    rx.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 8 << 20)
    rx.bind(("127.0.0.1", 0))
    port = rx.getsockname()[1]
    rx.setblocking(False)
    args = [str(modem), "-R", "-r", str(port), "-d", str(child)] + (["-v"] if verbose_child else [])
    p = subprocess.Popen(args, stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    data = memoryview(np.ascontiguousarray(_iq(iq)).tobytes())
    grams = []

    def drain():
        while True:
            try:
                grams.append(rx.recv(65536))
            except BlockingIOError:
                return
    err = []
    import threading
    th = threading.Thread(target=lambda: err.append(p.stderr.read()), daemon=True)
    th.start()
    for at in range(0, len(data), piece):
        p.stdin.write(data[at:at + piece])        # (blocks while the child's pipe is full: the parent's own back-pressure)
        drain()
    p.stdin.close()
    t_end = time.time() + recv_timeout
    while p.poll() is None and time.time() < t_end:
        select.select([rx], [], [], 0.05)
        drain()
    if p.poll() is None:
        p.kill()
        p.wait()
        raise AssertionError("opv-modem -R did not finish: the child never closed its stdout?")
    time.sleep(0.05)
    drain()
    th.join(5)
    rx.close()
    return grams, p.returncode, (err[0].decode("utf-8", "replace") if err else "")


def run_reference_modem_loopback(child, frames, gap_s=0.12, settle_s=4.0, start_s=0.4):
    """The boundary's OTHER real caller: the reference's `opv-modem -l -d <child>` (loopback / repeater mode,
    src/opv-modem.cpp:855-1000). The parent takes 134-byte frames by UDP, modulates each with its own persistent modulator,
    writes the frame's 86 720 samples to a PERSISTENT `<child> -s -r` (PersistentDemodulator, :348-468: fork, execlp, child's
    stderr to /dev/null, blocking 347 KB writes, non-blocking reads of the decoded records) and sends every decoded frame back
    to the sender. `frames` ([K, 134] uint8) are sent one datagram at a time, `gap_s` apart - a live link, not a file; returns
    (list of returned datagrams in arrival order, the parent's stderr text). The demodulator's one-frame latency shows here:
    frame k comes back once frame k + 1 has been sent."""
    import signal
    import socket
    import subprocess
    import time
    modem = ref_binary("opv-modem")
    assert modem is not None, "oracle/_ref/opv-modem is missing (make -C oracle ref, in the build container)"
    probe = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    probe.bind(("127.0.0.1", 0))
    port = probe.getsockname()[1]
    probe.close()
    p = subprocess.Popen([str(modem), "-l", "-p", str(port), "-d", str(child)], stdin=subprocess.DEVNULL,
                         stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    so = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    so.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 1 << 22)
    so.bind(("127.0.0.1", 0))
    so.setblocking(False)
    got = []

    def drain():
        while True:
            try:
                got.append(so.recv(4096))
            except BlockingIOError:
                return
    try:
        time.sleep(start_s)                                  # the parent binds its port and starts the child
        assert p.poll() is None, p.stderr.read().decode("utf-8", "replace")
        for f in np.ascontiguousarray(frames, np.uint8).reshape(-1, FRAME_BYTES):
            so.sendto(f.tobytes(), ("127.0.0.1", port))
            t_end = time.time() + gap_s
            while time.time() < t_end:
                drain()
                time.sleep(0.005)
        t_end, n_last = time.time() + settle_s, -1
        while time.time() < t_end:                           # until nothing new has arrived for a while
            drain()
            if len(got) != n_last:
                n_last, t_end = len(got), time.time() + settle_s
            time.sleep(0.02)
    finally:
        p.send_signal(signal.SIGINT)                         # the parent's own shutdown path (it ends its child itself)
        try:
            err = p.communicate(timeout=30)[1]
        except subprocess.TimeoutExpired:
            p.kill()
            err = p.communicate()[1]
        so.close()
    return got, err.decode("utf-8", "replace")
