"""ctypes binding of the C ABI in include/opv_demod.h (libopv_demod_hip.so).

Python is plumbing here (tests, bench, multi-GPU launch); the product is the shared library.
The directory name of this package contains a '-', so import it by path:

    import importlib.util, pathlib
    spec = importlib.util.spec_from_file_location("opv_amd", ".../opv-cxx-demod_amd/opv_amd.py")

or use ``load_opv_amd()`` from the repo-root ``__graft_entry__``.
There is no CPU fallback: if the library is missing or no MI355X is visible, calls raise.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("OPV_LIB", PKG / "libopv_demod_hip.so"))   # (OPV_LIB: dev switch, an experimental build of the same ABI)

SPS = 40
FRAME_BYTES = 134
FRAME_BITS = 1072
ENCODED_BITS = 2144
FRAME_SYMBOLS = 2168
CHUNK_SAMPLES = 86720
SAMPLE_RATE = 2168000.0

EXPORTS = [
    "opv_create", "opv_destroy", "opv_last_error", "opv_abi_version", "opv_push_iq", "opv_push_iq_batch", "opv_push_iq_batch_async", "opv_push_wait", "opv_flush",
    "opv_attach_device_iq", "opv_process", "opv_sync", "opv_set_frontend", "opv_reset_stream", "opv_pop_frames", "opv_pop_events",
    "opv_get_state", "opv_device_frames", "opv_hip_stream", "opv_comm_unique_id", "opv_comm_init", "opv_comm_init_all",
    "opv_comm_destroy", "opv_gather_frames", "opv_gather_frames_all", "opv_tap_soft", "opv_tap_chunks",
    "opv_tap_offset_energies", "opv_offset_ties_on_host", "opv_offset_ties_decided_on_host", "opv_offset_ties_left_to_device", "opv_tap_wave_info", "opv_tap_occupancy", "opv_decode_payloads", "opv_tx_bert_frame", "opv_tx_bert_frames", "opv_tx_modulated_samples",
    "opv_tx_modulate", "opv_tap_tx_checkpoints", "opv_frontend_kernel", "opv_channel_device", "opv_resample_device", "opv_enable_timing", "opv_kernel_times", "opv_tx_modulate_device", "opv_tx_modulate_device_to_host",
    "opv_tx_stream_create", "opv_tx_stream_reset", "opv_tx_stream_frames", "opv_tx_stream_tail", "opv_tx_stream_destroy", "opv_tap_tx_frame",
]


class OpvError(RuntimeError):
    pass


class Cfg(C.Structure):
    _fields_ = [("streaming", C.c_int32), ("have_init_offset", C.c_int32), ("init_offset_hz", C.c_double),
                ("afc_alpha", C.c_double), ("device", C.c_int32), ("coherent", C.c_int32),
                ("max_samples", C.c_uint64), ("pll_bw_hz", C.c_double)]


class FrameMeta(C.Structure):
    _fields_ = [("viterbi_metric", C.c_int32), ("sync_ok", C.c_int32), ("sync_quality", C.c_double),
                ("release_symbol", C.c_uint64), ("payload_symbol", C.c_uint64)]


class StreamState(C.Structure):
    _fields_ = [("freq_offset_hz", C.c_double), ("timing_freq", C.c_double), ("est_offset_hz", C.c_double),
                ("mu", C.c_double), ("total_symbols", C.c_uint64), ("total_samples", C.c_uint64),
                ("chunk_origin", C.c_uint64), ("sync_state", C.c_int32), ("frames_released", C.c_int32),
                ("frames_decoded", C.c_int32), ("frames_perfect", C.c_int32), ("n_chunks", C.c_int32),
                ("flushed", C.c_int32), ("events_dropped", C.c_uint32), ("edge_ties", C.c_uint32),
                ("stalled", C.c_int32), ("offset_ties", C.c_int32)]


EVENT_DTYPE = np.dtype(
    [("kind", "<i4"), ("count", "<i4"), ("sym_idx", "<u8"), ("corr", "<f8"), ("raw", "<f8")], align=True)
META_DTYPE = np.dtype([("viterbi_metric", "<i4"), ("sync_ok", "<i4"), ("sync_quality", "<f8"),
                       ("release_symbol", "<u8"), ("payload_symbol", "<u8")], align=True)


def build(force=False):
    """Compile the library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or not LIB_PATH.exists():
        subprocess.run(["make", "-s", "-C", str(PKG), "-j8", "all"], check=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise OpvError(f"{LIB_PATH} is missing: run `make -C {PKG}` (there is no CPU fallback)")
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7. Two HIP runtimes in one process do not
        # work ("No HIP GPUs are available" in whichever initialises second), and the dynamic loader
        # keys on the soname: whichever copy is loaded first serves both. When torch is present, let
        # it load first so that this library binds to the runtime torch was built for.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(str(LIB_PATH))
        L.opv_last_error.restype = C.c_char_p
        L.opv_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Cfg)]
        L.opv_destroy.argtypes = [C.c_void_p]
        L.opv_destroy.restype = None
        L.opv_push_iq.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.opv_flush.argtypes = [C.c_void_p, C.c_int]
        L.opv_push_iq_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.opv_push_iq_batch_async.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.opv_push_wait.argtypes = [C.c_void_p]
        L.opv_attach_device_iq.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
        L.opv_process.argtypes = [C.c_void_p]
        L.opv_sync.argtypes = [C.c_void_p]
        L.opv_set_frontend.argtypes = [C.c_void_p, C.c_int]
        L.opv_frontend_kernel.restype = C.c_char_p
        L.opv_frontend_kernel.argtypes = [C.c_void_p]
        L.opv_reset_stream.argtypes = [C.c_void_p, C.c_int]
        L.opv_pop_frames.restype = C.c_long
        L.opv_pop_frames.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.opv_pop_events.restype = C.c_long
        L.opv_pop_events.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.opv_get_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(StreamState)]
        L.opv_device_frames.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.opv_hip_stream.restype = C.c_void_p
        L.opv_hip_stream.argtypes = [C.c_void_p]
        L.opv_comm_unique_id.argtypes = [C.c_char_p]
        L.opv_comm_init.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.opv_comm_init_all.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
        L.opv_comm_destroy.restype = None
        L.opv_comm_destroy.argtypes = [C.c_void_p]
        L.opv_gather_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.opv_gather_frames_all.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.opv_tap_soft.restype = C.c_long
        L.opv_tap_soft.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_size_t]
        L.opv_tap_chunks.restype = C.c_long
        L.opv_tap_chunks.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_size_t]
        L.opv_tap_offset_energies.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.opv_tap_wave_info.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.opv_offset_ties_on_host.argtypes = [C.c_void_p]
        L.opv_offset_ties_decided_on_host.restype = C.c_uint64
        L.opv_offset_ties_decided_on_host.argtypes = [C.c_void_p]
        L.opv_offset_ties_left_to_device.restype = C.c_uint64
        L.opv_offset_ties_left_to_device.argtypes = [C.c_void_p]
        L.opv_tap_occupancy.argtypes = [C.c_void_p, C.c_void_p]
        L.opv_decode_payloads.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
        L.opv_tx_bert_frame.restype = None
        L.opv_tx_bert_frame.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.opv_tx_bert_frames.restype = None
        L.opv_tx_bert_frames.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_size_t, C.c_void_p]
        L.opv_tap_tx_checkpoints.restype = None
        L.opv_tap_tx_checkpoints.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p]
        L.opv_tx_modulated_samples.restype = C.c_size_t
        L.opv_tx_modulated_samples.argtypes = [C.c_size_t]
        L.opv_tx_modulate.restype = C.c_size_t
        L.opv_tx_modulate.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.opv_tap_tx_frame.restype = None
        L.opv_tap_tx_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.opv_tx_stream_create.restype = C.c_void_p
        L.opv_tx_stream_create.argtypes = []
        L.opv_tx_stream_reset.restype = None
        L.opv_tx_stream_reset.argtypes = [C.c_void_p]
        L.opv_tx_stream_destroy.restype = None
        L.opv_tx_stream_destroy.argtypes = [C.c_void_p]
        L.opv_tx_stream_frames.restype = C.c_size_t
        L.opv_tx_stream_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.opv_tx_stream_tail.restype = C.c_size_t
        L.opv_tx_stream_tail.argtypes = [C.c_void_p]
        L.opv_channel_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_double,
                                         C.c_double, C.c_uint64]
        L.opv_resample_device.restype = C.c_long
        L.opv_resample_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_double]
        L.opv_enable_timing.argtypes = [C.c_void_p, C.c_int]
        L.opv_tx_modulate_device.restype = C.c_long
        L.opv_tx_modulate_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.opv_kernel_times.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _chk(rc):
    if rc < 0:
        raise OpvError(f"opv error {rc}: {lib().opv_last_error().decode()}")
    return rc


def comm_create(world, rank, device=0, uid=None):
    """RCCL communicator through the library's own binding (opv_comm_unique_id / opv_comm_init); returns (comm, uid)"""
    if uid is None:
        buf = C.create_string_buffer(128)
        _chk(lib().opv_comm_unique_id(buf))
        uid = buf.raw
    comm = C.c_void_p()
    _chk(lib().opv_comm_init(C.byref(comm), world, rank, uid, device))
    return comm, uid


def comm_init_all(devices):
    """opv_comm_init_all: one RCCL communicator per listed GPU inside THIS process (rank i on devices[i]); returns the list"""
    n = len(devices)
    comms = (C.c_void_p * n)()
    devs = (C.c_int * n)(*[int(d) for d in devices])
    _chk(lib().opv_comm_init_all(comms, n, devs))
    return [C.c_void_p(comms[i]) for i in range(n)]


def gather_frames_all(demods, comms, root, d_frames_all, d_counts_all):
    """opv_gather_frames_all: the gathers of every context of this process (demods[i] on comms[i]) in one RCCL group"""
    n = len(demods)
    ctxs = (C.c_void_p * n)(*[d.h.value for d in demods])
    cm = (C.c_void_p * n)(*[c.value for c in comms])
    _chk(lib().opv_gather_frames_all(ctxs, cm, n, root, C.c_void_p(d_frames_all), C.c_void_p(d_counts_all)))


def comm_destroy(comm):
    lib().opv_comm_destroy(comm)


# ---------------------------------------------------------------- transmit side (host)
def bert_frames(n, callsign="W5NYV", token=0xBBAADD, first=0):
    out = np.zeros((n, FRAME_BYTES), np.uint8)
    lib().opv_tx_bert_frames(callsign.encode(), token, first, n, out.ctypes.data)
    return out


def callsign_of(frame):
    """Base-40 station id of a 134-byte frame's first six bytes (what the reference prints as `Station ID`,
    ref src/opv-demod.cpp:907-925 / src/opv-mod.cpp:63-90: big-endian 48 bits, first character least significant)"""
    v = int.from_bytes(bytes(bytearray(frame[:6])), "big")
    out = ""
    while v:
        c = v % 40
        out += "?" if c == 0 else chr(64 + c) if c <= 26 else chr(48 + c - 27) if c <= 36 else "-/."[c - 37]
        v //= 40
    return out


def tx_checkpoints(first, count):
    """(ph1, ph2) of the modulator's NCOs at symbols 128 * (first .. first + count - 1) of a run (parity tap)"""
    out = np.empty((count, 2), np.float64)
    lib().opv_tap_tx_checkpoints(first, count, out.ctypes.data)
    return out


def modulate(frames):
    frames = np.ascontiguousarray(frames, np.uint8).reshape(-1, FRAME_BYTES)
    n = lib().opv_tx_modulated_samples(len(frames))
    iq = np.empty(2 * n, np.int16)
    w = lib().opv_tx_modulate(frames.ctypes.data, len(frames), iq.ctypes.data)
    assert w == n
    return iq


def tx_frame_taps(frame):
    """(randomised bytes, coded bits in encoder order, interleaved bits) of one 134-byte frame (opv_tap_tx_frame)"""
    frame = np.ascontiguousarray(frame, np.uint8).reshape(FRAME_BYTES)
    r, c, i = np.empty(FRAME_BYTES, np.uint8), np.empty(2144, np.uint8), np.empty(2144, np.uint8)
    lib().opv_tap_tx_frame(frame.ctypes.data, r.ctypes.data, c.ctypes.data, i.ctypes.data)
    return r, c, i


class TxStream:
    """the host modulator with its state carried from call to call (opv_tx_stream_*; reference HDLModulator,
    src/opv-mod.cpp:219-291)"""

    def __init__(self):
        self.h = lib().opv_tx_stream_create()
        if not self.h:
            raise OpvError("opv_tx_stream_create failed")

    def reset(self):
        lib().opv_tx_stream_reset(self.h)

    def frames(self, frames):
        frames = np.ascontiguousarray(frames, np.uint8).reshape(-1, FRAME_BYTES)
        iq = np.empty(2 * len(frames) * 2168 * 40, np.int16)
        w = lib().opv_tx_stream_frames(self.h, frames.ctypes.data, len(frames), iq.ctypes.data)
        assert w == iq.size // 2
        return iq

    @staticmethod
    def tail():
        iq = np.empty(2 * 4000, np.int16)
        assert lib().opv_tx_stream_tail(iq.ctypes.data) == 4000
        return iq

    def close(self):
        if self.h:
            lib().opv_tx_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------- receiver
class Demod:
    """n_streams independent receivers on one GPU (mirrors the three reference objects
    MSKDemodulatorAFC + SyncTracker + FrameDecoder per stream)."""

    def __init__(self, n_streams=1, max_samples=1 << 22, streaming=True, init_offset=None, afc_alpha=0.001,
                 device=0, coherent=False, pll_bw=50.0):
        self.n_streams = n_streams
        self.cfg = Cfg(int(streaming), int(init_offset is not None), float(init_offset or 0.0), afc_alpha,
                       device, int(coherent), int(max_samples), float(pll_bw))
        self.h = C.c_void_p()
        _chk(lib().opv_create(C.byref(self.h), n_streams, C.byref(self.cfg)))
        if os.environ.get("OPV_FRONTEND"):      # dev switch: run everything on one mapping (0 / 1 / 4 / 16)
            self.set_frontend(int(os.environ["OPV_FRONTEND"]))

    def close(self):
        if getattr(self, "h", None):
            lib().opv_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    def push(self, stream, iq):
        iq = np.ascontiguousarray(iq, np.int16).reshape(-1)
        _chk(lib().opv_push_iq(self.h, stream, iq.ctypes.data, iq.size // 2))

    def push_batch(self, streams, blocks, wait=True):
        """opv_push_iq_batch: blocks[i] (int16 IQ) goes to streams[i]; one wait for all copies. wait=False:
        opv_push_iq_batch_async - the blocks must stay alive and unchanged until push_wait() (they are kept referenced here)."""
        blocks = [np.ascontiguousarray(b, np.int16).reshape(-1) for b in blocks]
        n = len(blocks)
        ids = (C.c_int * n)(*[int(s) for s in streams])
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in blocks])
        lens = (C.c_size_t * n)(*[b.size // 2 for b in blocks])
        if wait:
            _chk(lib().opv_push_iq_batch(self.h, n, ids, ptrs, lens))
        else:
            held = getattr(self, "_inflight", None)        # the previous batch's blocks stay referenced until the call has waited for them
            _chk(lib().opv_push_iq_batch_async(self.h, n, ids, ptrs, lens))
            self._inflight = blocks
            del held

    def push_wait(self):
        _chk(lib().opv_push_wait(self.h))
        self._inflight = None

    def flush(self, stream):
        _chk(lib().opv_flush(self.h, stream))

    def attach(self, stream, dev_ptr, n_samples, eof=True):
        _chk(lib().opv_attach_device_iq(self.h, stream, C.c_void_p(dev_ptr), n_samples, int(eof)))

    def process(self):
        _chk(lib().opv_process(self.h))

    def sync(self):
        _chk(lib().opv_sync(self.h))

    def set_frontend(self, streams_per_wave):
        _chk(lib().opv_set_frontend(self.h, int(streams_per_wave)))

    def frontend_kernel(self):
        """name of the front-end kernel the last process() launched"""
        return lib().opv_frontend_kernel(self.h).decode()

    def reset(self, stream=-1):
        _chk(lib().opv_reset_stream(self.h, stream))

    def enable_timing(self, on=True):
        _chk(lib().opv_enable_timing(self.h, int(on)))

    def kernel_times(self):
        """ms of the last process(): dict(offset_search, msk_frontend, sync_track, frame_decode)"""
        t = (C.c_float * 4)()
        _chk(lib().opv_kernel_times(self.h, t))
        return dict(offset_search=t[0], msk_frontend=t[1], sync_track=t[2], frame_decode=t[3])

    def pop_frames(self, stream, cap=None):
        cap = cap or (int(self.cfg.max_samples) // (FRAME_SYMBOLS * 38) + 8)
        out = np.zeros((cap, FRAME_BYTES), np.uint8)
        meta = np.zeros(cap, META_DTYPE)
        n = _chk(lib().opv_pop_frames(self.h, stream, out.ctypes.data, cap, meta.ctypes.data))
        return out[:n].copy(), meta[:n].copy()

    def pop_events(self, stream, cap=None):
        cap = cap or (4 * (int(self.cfg.max_samples) // (FRAME_SYMBOLS * 38) + 8) + 64)
        ev = np.zeros(cap, EVENT_DTYPE)
        n = _chk(lib().opv_pop_events(self.h, stream, ev.ctypes.data, cap))
        return ev[:n].copy()

    def state(self, stream):
        s = StreamState()
        _chk(lib().opv_get_state(self.h, stream, C.byref(s)))
        return s

    def soft(self, stream, first=0, cap=None):
        cap = cap or (int(self.cfg.max_samples) // 38 + 128)
        out = np.empty(cap, np.float64)
        n = _chk(lib().opv_tap_soft(self.h, stream, first, out.ctypes.data, cap))
        return out[:n].copy()

    def chunks(self, stream, first=0):
        cap = int(self.cfg.max_samples) // 80000 + 4
        out = np.zeros((cap, 5), np.float64)
        n = _chk(lib().opv_tap_chunks(self.h, stream, first, out.ctypes.data, cap))
        return out[:n].copy()

    def offset_energies(self, stream):
        out = np.zeros(134, np.float64)
        _chk(lib().opv_tap_offset_energies(self.h, stream, out.ctypes.data))
        return out

    def offset_ties_on_host(self):
        """True when the offset search's near-ties are decided with the host's libm (include/opv_demod.h)"""
        return bool(lib().opv_offset_ties_on_host(self.h))

    def offset_ties_decided_on_host(self):
        """streams whose offset-search tie the host's libm has decided so far (final for a round after sync())"""
        return int(lib().opv_offset_ties_decided_on_host(self.h))

    def offset_ties_left_to_device(self):
        """streams listed for the host beyond what a round stages (their device decision stood); 0 in ordinary operation"""
        return int(lib().opv_offset_ties_left_to_device(self.h))

    def wave_info(self, stream):
        """(HW_ID, XCC_ID, shader cycles, 100 MHz ticks) of the wave that ran the stream's last front-end launch"""
        out = (C.c_uint64 * 4)()
        _chk(lib().opv_tap_wave_info(self.h, stream, out))
        return tuple(int(v) for v in out)

    def occupancy(self):
        """workgroups per CU the runtime can keep resident, per hot-path kernel (opv_tap_occupancy)"""
        out = (C.c_int * 6)()
        _chk(lib().opv_tap_occupancy(self.h, out))
        return dict(zip(("k_msk_frontend_rb", "k_msk_frontend_rb_wg4", "k_msk_frontend_x16_wg4", "k_msk_frontend_x4_wg4", "k_frame_decode",
                         "k_frame_scale"), [int(v) for v in out]))

    def decode_payloads(self, soft, taps=False):
        soft = np.ascontiguousarray(soft, np.float64).reshape(-1, ENCODED_BITS)
        n = len(soft)
        out = np.zeros((n, FRAME_BYTES), np.uint8)
        met = np.zeros(n, np.int32)
        q = np.zeros((n, ENCODED_BITS), np.int8) if taps else None
        de = np.zeros((n, ENCODED_BITS), np.int8) if taps else None
        bits = np.zeros((n, FRAME_BITS), np.uint8) if taps else None
        _chk(lib().opv_decode_payloads(self.h, soft.ctypes.data, n, out.ctypes.data, met.ctypes.data,
                                       q.ctypes.data if taps else None, de.ctypes.data if taps else None,
                                       bits.ctypes.data if taps else None))
        return dict(frames=out, metrics=met, q=q, deint=de, bits=bits)

    def device_frames(self):
        f, m, c, cap = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_size_t()
        _chk(lib().opv_device_frames(self.h, C.byref(f), C.byref(m), C.byref(c), C.byref(cap)))
        return f.value, m.value, c.value, cap.value

    def hip_stream(self):
        return lib().opv_hip_stream(self.h)

    def gather_frames(self, comm, root, d_frames_all, d_counts_all):
        """opv_gather_frames: this context's frame buffer + counts to `root` over the RCCL communicator `comm`"""
        _chk(lib().opv_gather_frames(self.h, comm, root, C.c_void_p(d_frames_all), C.c_void_p(d_counts_all)))

    def modulate_device(self, frames, d_out):
        """TX chain into HBM (bit-identical to modulate()); returns samples re-evaluated on the host."""
        frames = np.ascontiguousarray(frames, np.uint8).reshape(-1, FRAME_BYTES)
        return _chk(lib().opv_tx_modulate_device(self.h, frames.ctypes.data, len(frames), C.c_void_p(d_out)))

    def channel(self, d_in, d_out, n_samples, gain=1.0, f0_hz=0.0, sigma=0.0, seed=0):
        _chk(lib().opv_channel_device(self.h, C.c_void_p(d_in), C.c_void_p(d_out), n_samples, gain, f0_hz, sigma,
                                      seed))

    def resample(self, d_in, n_in, d_out, out_cap, clock_ppm):
        """opv_resample_device: returns the number of samples written to d_out"""
        n = lib().opv_resample_device(self.h, C.c_void_p(d_in), n_in, C.c_void_p(d_out), out_cap, clock_ppm)
        _chk(n)
        return n

    # convenience: the whole reference main() for one host capture on stream 0..n-1
    def receive(self, captures):
        """captures: list of int16 IQ arrays, one per stream. Returns per-stream dicts."""
        assert len(captures) == self.n_streams
        for s, iq in enumerate(captures):
            self.push(s, iq)
            self.flush(s)
        self.process()
        self.sync()
        res = []
        for s in range(self.n_streams):
            fr, meta = self.pop_frames(s)
            ev = self.pop_events(s)
            st = self.state(s)
            while st.stalled:                   # back-pressure: pop, run another round, until the stream has finished
                before = (st.total_symbols, st.frames_released, len(fr))
                self.process()
                f2, m2 = self.pop_frames(s)
                fr, meta = np.concatenate([fr, f2]), np.concatenate([meta, m2])
                ev = np.concatenate([ev, self.pop_events(s)])
                st = self.state(s)
                if st.stalled and (st.total_symbols, st.frames_released, len(fr)) == before:
                    raise OpvError(f"stream {s} stalled (0x{st.stalled:x}) without progress")
            res.append(dict(frames=fr, meta=meta, events=ev, soft=self.soft(s), state=st, chunks=self.chunks(s)))
        return res


def format_events(events):
    """The stderr lines of SyncTracker::process (reference src/opv-demod.cpp:651,677,695,699,705)."""
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
