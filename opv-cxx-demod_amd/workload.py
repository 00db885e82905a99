"""Synthetic multi-stream workloads, generated IN HBM by the device modulator and the device channel tool
(SURVEY.md §8d C4/C5 recipe; include/opv_demod.h: opv_tx_modulate_device, opv_channel_device).

Global stream g is its own BERT capture (callsign ``S<g>``, frame numbers from 1000 g: bit-identical to what
``opv-mod -S S<g> -B F`` emits for those frame numbers) at amplitude 2000, carrier offset
f0 = -2000 + 4000 (g mod 64) / 63 Hz - SURVEY.md §8(d) C4 as written: the two edge streams sit ON the AFC clamp
(ref src/opv-demod.cpp:303) and outside the +/-1530 Hz span of the offset search (:135,169) - AWGN at the given
Eb/N0 from the counter-based generator keyed by seed 1000 + g. `clean=True` is C4's "all-clean variant": the same
per-stream BERT captures exactly as the modulator emits them (full scale, no offset, no noise, no channel pass).
bench.py, the multi-rank GPU test and the scripts all build their inputs here, so that what is benchmarked is what
is parity-tested.
"""
import numpy as np

AMP = 2000.0
F0_EDGE_HZ = 2000.0          # f0 runs from -F0_EDGE_HZ (stream 0) to +F0_EDGE_HZ (stream 63) within every 64-stream shard
RECIPE = f"amp {AMP:g}, f0 {-F0_EDGE_HZ:g}..{F0_EDGE_HZ:+g} Hz"


def stream_params(g, ebn0):
    """(callsign, first frame number, f0 in Hz, sigma per component, seed) of global stream g"""
    sigma = 0.0
    if ebn0 is not None and ebn0 > 0:
        # Eb = 2 Es (rate 1/2), Es = 40 A^2  ->  total complex noise variance 80 A^2 / (Eb/N0)
        sigma = float(np.sqrt(80.0 * AMP * AMP / 10.0 ** (ebn0 / 10.0) / 2.0))
    return f"S{g}", 1000 * g, -F0_EDGE_HZ + 2.0 * F0_EDGE_HZ * (g % 64) / 63.0, sigma, 1000 + g


def tx_frames(amd, g, n_frames):
    cs, first, _, _, _ = stream_params(g, None)
    return amd.bert_frames(n_frames, callsign=cs, first=first)


def generate(amd, dm, torch, dev, global_ids, n_frames, ebn0, timing=None, clean=False):
    """Fill HBM with one impaired capture per global stream id. Returns (d_iq [S, 2 n] int16 on `dev`,
    tx [S, n_frames, 134] uint8 numpy, n samples per stream). `dm` is any Demod context on that device (the
    generator kernels run on its HIP stream). timing: a dict that receives generate_s, the seconds spent behind the
    allocation of the (up to 178 GB) capture buffer: BERT frames, device transmit chain, channel tool, for all streams.
    clean: the all-clean variant (the device modulator writes every capture in place, no channel pass)."""
    import time
    n = amd.lib().opv_tx_modulated_samples(n_frames)
    assert n % 4 == 0
    S = len(global_ids)
    d_clean = None if clean else torch.empty(2 * n, dtype=torch.int16, device=dev)
    d_iq = torch.empty((S, 2 * n), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tx = np.empty((S, n_frames, 134), np.uint8)
    for k, g in enumerate(global_ids):
        _, _, f0, sigma, seed = stream_params(g, ebn0)
        tx[k] = tx_frames(amd, g, n_frames)
        if clean:
            dm.modulate_device(tx[k], d_iq[k].data_ptr())
            continue
        dm.modulate_device(tx[k], d_clean.data_ptr())
        dm.channel(d_clean.data_ptr(), d_iq[k].data_ptr(), n, gain=AMP / 16383.0, f0_hz=f0, sigma=sigma, seed=seed)
    dm.sync()
    if timing is not None:
        timing["generate_s"] = time.perf_counter() - t0
    return d_iq, tx, n


class DevPtr:
    """Zero-copy torch view of library-owned device memory (CUDA array interface v2)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (ptr, False), "shape": shape, "typestr": typestr, "version": 2}


def frame_views(dm, torch, dev):
    """(frames [S, cap, 134] uint8, counts [S] int32) as torch tensors aliasing the library's device buffers"""
    fptr, _mptr, cptr, fcap = dm.device_frames()
    S = dm.n_streams
    return (torch.as_tensor(DevPtr(fptr, (S, fcap, 134), "|u1"), device=dev),
            torch.as_tensor(DevPtr(cptr, (S,), "<i4"), device=dev))
