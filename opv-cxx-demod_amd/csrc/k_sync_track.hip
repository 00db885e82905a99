// k_sync_track.hip — 24-bit soft sync-word correlator + HUNTING/VERIFYING/LOCKED flywheel,
// one wavefront per stream over that stream's soft-symbol log.
//
// Replaces SyncTracker::process / soft_correlate (reference src/opv-demod.cpp:615-757).
// The reference is a per-symbol state machine; restated on log positions it only ever
// (a) scans for the first symbol whose 24-symbol window passes the HUNTING thresholds, or
// (b) jumps: VERIFYING releases its payload 2144 symbols after the sync (:658), LOCKED checks
//     the sync 2168 symbols after the previous one (:684) and releases 2144 after it (:720).
// So the wave scans 64 candidate positions per step while HUNTING (one 24-tap window per
// lane, ballot + first-set-bit keeps "first hit wins") and jumps otherwise. Each window sum
// runs oldest->newest exactly like soft_correlate (:747-752) with the +/-1 pattern (:597-600),
// so given the same soft symbols every threshold decision (0.85 / 5000 / 0.70 / energy 100,
// :783-786) is bit-identical to the reference. Frames are recorded as positions into the
// soft log (payload = symbols anchor+1 .. anchor+2144) for k_frame_decode.
//
// Cost: O(symbols) only while hunting; O(frames) when locked. Bytes: 8 B/symbol re-read from
// L2/HBM while hunting, 192 B/frame when locked.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "opv_device.h"

namespace {

struct Corr { double raw, norm; };

// ref :743-757
__device__ inline Corr window_corr(const double* __restrict__ soft, uint32_t mask, uint64_t last) {
    const uint32_t w0 = (uint32_t)(last - (OPV_SYNC_BITS - 1));  // the soft log is a power-of-two ring
    double sum = 0.0, energy = 0.0;
#pragma unroll
    for (int i = 0; i < OPV_SYNC_BITS; ++i) {
        const double s = soft[(w0 + (uint32_t)i) & mask];
        const double pat = ((OPV_SYNC_WORD >> (OPV_SYNC_BITS - 1 - i)) & 1u) ? -1.0 : 1.0;  // ref :597-600
        sum += s * pat;
        energy += fabs(s);
    }
    Corr c;
    c.raw = sum;
    c.norm = energy < 100.0 ? 0.0 : sum / energy;  // ref :755-756
    return c;
}

}  // namespace

extern "C" __global__ __launch_bounds__(64) void k_sync_track(OpvStream* __restrict__ streams) {
    OpvStream& st = streams[blockIdx.x];
    const int lane = threadIdx.x;
    const double* __restrict__ soft = st.soft;
    const uint32_t mask = (uint32_t)(st.cap_soft - 1);
    const uint64_t n = st.n_soft;

    int state = st.trk_state, collecting = st.trk_collecting, misses = st.trk_misses;
    uint64_t anchor = st.trk_anchor, next = st.trk_next;
    double quality = st.trk_quality;
    uint32_t n_frames = st.n_frames, n_events = st.n_events;
    int overflow = st.overflow;
    if (lane == 0) st.dec_from = n_frames;

    // The event log (the reference's stderr lines) is LOSSY: a ring of the most recent cap_events entries; a
    // consumer that never reads it (opv-modem sends the child's stderr to /dev/null) loses the oldest lines and
    // nothing else. Frames are not: a full ring of unpopped frames stops the tracker in front of the release
    // (state untouched) until opv_pop_frames has made room.
    auto event = [&](int kind, int count, uint64_t sym, double corr, double raw) {
        if (lane == 0) {
            OpvEventRec& e = st.events[n_events % st.cap_events];
            e.kind = kind; e.count = count; e.sym_idx = sym; e.corr = corr; e.raw = raw;
        }
        ++n_events;
    };
    int sync_ok = st.trk_sync_ok, stalled = 0;
    auto frames_full = [&]() { return n_frames - st.frames_popped >= st.cap_frames; };
    auto release = [&](uint64_t at) {  // ref :660-668 / :721-729
        if (lane == 0) {
            OpvFrameRec& f = st.frec[n_frames % st.cap_frames];
            f.payload_sym = anchor + 1;
            f.release_sym = at;
            f.quality = quality;
            f.sync_ok = sync_ok;
            f.pad = 0;
            st.metrics[n_frames % st.cap_frames] = INT32_MIN;   // "released, not decoded yet" (the slot is a ring entry)
        }
        ++n_frames;
        collecting = 0;
    };

    for (;;) {
        if (overflow) break;
        if (state == 0) {  // HUNTING (ref :635-655)
            uint64_t pos = next < (OPV_SYNC_BITS - 1) ? (uint64_t)(OPV_SYNC_BITS - 1) : next;  // ref :637
            bool found = false;
            while (pos < n) {
                const uint64_t cand = pos + (uint64_t)lane;
                Corr c = {0.0, 0.0};
                bool ok = false;
                if (cand < n) {
                    c = window_corr(soft, mask, cand);
                    ok = (c.raw >= 5000.0) && (c.norm >= 0.85);  // ref :642
                }
                const unsigned long long m = __ballot(ok);
                if (m) {
                    const int first = __ffsll((long long)m) - 1;
                    const uint64_t s = pos + (uint64_t)first;
                    const double nrm = __shfl(c.norm, first, 64), raw = __shfl(c.raw, first, 64);
                    state = 1;  // VERIFYING
                    sync_ok = 1;
                    quality = nrm;
                    anchor = s;  // symbols_since_sync_ = 0 here (ref :645)
                    collecting = 1;
                    event(1, 0, s, nrm, raw);
                    next = s + 1;
                    found = true;
                    break;
                }
                pos += 64;
            }
            if (!found) { next = n; break; }
        } else if (state == 1) {  // VERIFYING (ref :657-680)
            const uint64_t r = anchor + OPV_CODED;
            if (r >= n) break;
            if (frames_full()) { stalled = 2; break; }
            release(r);
            state = 2;
            misses = 0;
            event(2, (int)n_frames, r, 0.0, 0.0);
            next = r + 1;
        } else {  // LOCKED (ref :682-732)
            if (collecting) {
                const uint64_t r = anchor + OPV_CODED;
                if (r >= n) break;
                if (frames_full()) { stalled = 2; break; }
                release(r);
                next = r + 1;
                continue;
            }
            const uint64_t c = anchor + OPV_FSYMS;  // ref :684
            if (c >= n) break;
            const Corr k = window_corr(soft, mask, c);
            next = c + 1;
            if (k.norm >= 0.70) {  // ref :688
                misses = 0;
                quality = k.norm;
                collecting = 1;
                sync_ok = 1;
                event(3, 0, c, k.norm, k.raw);
            } else {
                ++misses;
                event(4, misses, c, k.norm, k.raw);
                if (misses >= 5) {  // ref :702 (SYNC_MISS_LIMIT :60)
                    state = 0;
                    collecting = 0;
                    event(5, 0, c, k.norm, k.raw);
                    continue;  // ref :706: symbols_since_sync_ is NOT reset
                }
                quality = k.norm;  // flywheel (ref :709-712)
                collecting = 1;
                sync_ok = 0;
            }
            anchor = c;  // symbols_since_sync_ = 0 (ref :716)
        }
    }

    if (lane == 0) {
        st.trk_state = state; st.trk_collecting = collecting; st.trk_misses = misses;
        st.trk_anchor = anchor; st.trk_next = next; st.trk_quality = quality;
        st.n_frames = n_frames; st.n_events = n_events; st.trk_sync_ok = sync_ok;
        st.stalled |= stalled;  // (the front-end, which runs first, rewrote bit 0 this round)
    }
}
