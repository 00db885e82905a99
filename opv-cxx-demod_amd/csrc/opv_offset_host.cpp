// opv_offset_host.cpp — the carrier-offset search's TIE decision on the host.
//
// k_offset_search.hip evaluates the 134 candidates of MSKDemodulatorAFC::estimate_offset (reference
// src/opv-demod.cpp:131-202) from ONE pass over the samples; its energies agree with the reference's to ~1e-13
// relative. When another candidate lies within 1e-11 of the winner the order of the two is decided by the last places
// of sin / cos - i.e. by libm. The reference's libm is the host's, not the device's: for such a stream a kernel behind
// the search copies the first <= 40 000 samples (160 KB) into pinned host memory and a host function enqueued behind THAT
// (hipLaunchHostFunc; opv_capi.hip: opv_process - the caller does not wait) repeats the decision here, the contenders
// evaluated by the reference's own loop (phases accumulated sample by sample from zero, one sin and one cos per LO and
// sample, sums in its order) with the host's sin / cos. Nothing in this file calls HIP. Same idea as the transmit chain's
// ambiguous samples (opv_capi.hip: opv_tx_modulate_device). Compiled with the host compiler, -ffp-contract=off (Makefile: CXXFLAGS).
//
// This is product code: a few candidates of a few streams (the guard fires on <= 4 of 512 ordinary captures), never the
// whole search and never the receive chain - there is no CPU implementation of the hot path in this library.
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "opv_device.h"
#include "opv_offset_host.h"

namespace {
constexpr double kTwoPi = 2.0 * 3.14159265358979323846;  // ref :43-44
constexpr double kFs = 2168000.0;                        // ref :40
constexpr double kFdev = 13550.0;                        // ref :42
constexpr int kTerms = 2 * OPV_OFFS_TERMS - 1;
constexpr double kTieRel = 1e-11;                        // k_offset_search.hip: kTieRel
// k_offset_search.hip: `near` - the same products and the same subtraction, so the same verdicts
inline bool near(double top, double e, double rel, double power) {
    const double d = top - e;
    return d <= rel * top || d * d <= (rel * rel * 40.0) * power * top;
}
}  // namespace

// ref :143-159 for one candidate: energy over nsym fixed 40-sample windows from sample 0. The reference's loop is sequential in
// two cheap things only - the LO phases (one addition per sample, never wrapped, :154-155) and the sum of the window energies
// (:158); both stay sequential here, in its order. The 160 000 sin / cos in between depend on nothing but a window's starting
// phases, so the windows are shared out over a few threads: the same numbers, 3 ms -> ~0.4 ms per candidate.
double opv_offset_candidate_energy(const int16_t* iq, size_t nsym, double offset, bool threads) {
    const double inc1 = kTwoPi * (-kFdev + offset) / kFs;   // ref :137
    const double inc2 = kTwoPi * (+kFdev + offset) / kFs;   // ref :138
    std::vector<double> start(2 * nsym), energy(nsym);
    double a1 = 0.0, a2 = 0.0;
    for (size_t s = 0; s < nsym; ++s) {                     // the phases at every window start, accumulated exactly like the reference's
        start[2 * s] = a1;
        start[2 * s + 1] = a2;
        for (int i = 0; i < OPV_SPS; ++i) { a1 += inc1; a2 += inc2; }
    }
    auto windows = [&](size_t s0, size_t s1) {
        for (size_t s = s0; s < s1; ++s) {
            double ph1 = start[2 * s], ph2 = start[2 * s + 1];
            double c1r = 0.0, c1i = 0.0, c2r = 0.0, c2i = 0.0;
            const int16_t* x = iq + 2 * (size_t)OPV_SPS * s;
            for (int i = 0; i < OPV_SPS; ++i, x += 2) {
                const double re = (double)x[0], im = (double)x[1];
                const double k1 = std::cos(ph1), s1 = std::sin(ph1);
                const double k2 = std::cos(ph2), s2 = std::sin(ph2);
                c1r += re * k1 + im * s1;                    // s * conj(lo) (ref :151-152)
                c1i += im * k1 - re * s1;
                c2r += re * k2 + im * s2;
                c2i += im * k2 - re * s2;
                ph1 += inc1;
                ph2 += inc2;
            }
            energy[s] = (c1r * c1r + c1i * c1i) + (c2r * c2r + c2i * c2i);
        }
    };
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 8) nt = 8;
    if (!threads || nt < 2 || nsym < 256) {
        windows(0, nsym);
    } else {
        std::vector<std::thread> pool;
        const size_t per = (nsym + nt - 1) / nt;
        size_t covered = per < nsym ? per : nsym;             // [0, covered) is this thread's own share
        for (unsigned t = 1; t < nt; ++t) {
            const size_t s0 = t * per, s1 = s0 + per < nsym ? s0 + per : nsym;
            if (s0 >= s1) break;
            try {
                pool.emplace_back(windows, s0, s1);
            } catch (...) {                                   // no more threads to be had: the rest is done here (same numbers)
                break;
            }
            covered = s1;
        }
        windows(0, per < nsym ? per : nsym);
        if (covered < nsym && pool.size() + 1 < nt) windows((pool.size() + 1) * per, nsym);
        for (auto& th : pool) th.join();
    }
    double total = 0.0;
    for (size_t s = 0; s < nsym; ++s) total += energy[s];   // ref :158, in its order
    return total;
}

// The decision of k_offset_search.hip (`decide`, coarse then fine), candidate for candidate, with the contenders
// re-evaluated by opv_offset_candidate_energy. poly: the device's 19 coefficients; the Horner evaluation below is the
// device's own (fma, same operand order), so the polynomial energies and with them the set of contenders are the device's.
double opv_offset_decide_on_host(const int16_t* iq, size_t nsym, const double* poly, double power, double* energies134, uint32_t* ties_out,
                                 bool threads) {
    auto poly_energy = [&](double offset) {
        const double th = kTwoPi * offset / kFs;
        double e = poly[kTerms - 1];
        for (int p = kTerms - 2; p >= 0; --p) e = std::fma(e, th, poly[p]);
        return e;
    };
    double e[134];
    double best_e = 0.0, best = 0.0, fine_best = 0.0;
    uint32_t ties = 0;
    auto decide = [&](int c0, int c1, double base, double step, bool fine) {
        auto off = [&](int c) { return base + step * (double)(c - c0); };
        for (int c = c0; c < c1; ++c) e[c] = poly_energy(off(c));
        double top = best_e;
        for (int c = c0; c < c1; ++c) top = std::fmax(top, e[c]);
        if (top > 0.0) {
            auto in_play = [&](int c) { return near(top, e[c], kTieRel, power) && !(fine && off(c) == best); };
            const bool defend = fine && near(top, best_e, kTieRel, power);   // (decided before any energy is replaced, like the device's uniform flow)
            int contenders = defend ? 1 : 0;
            for (int c = c0; c < c1; ++c) contenders += in_play(c);
            if (contenders > 1) {
                for (int c = c0; c < c1; ++c)
                    if (in_play(c)) { e[c] = opv_offset_candidate_energy(iq, nsym, off(c), threads); ++ties; }
                if (defend) { best_e = opv_offset_candidate_energy(iq, nsym, best, threads); ++ties; }
            }
        }
        for (int c = c0; c < c1; ++c) {
            energies134[c] = e[c];
            if (fine && off(c) == best) continue;                 // the coarse winner's own repeat: never '>' in the reference
            if (e[c] > best_e) {                                  // strict: first maximum wins (ref :161, :195)
                best_e = e[c];
                if (fine) fine_best = off(c);
                else best = off(c);
            }
        }
        if (!fine) fine_best = best;                              // ref :168
    };
    decide(0, 121, -1500.0, 25.0, false);                        // ref :135
    decide(121, 134, best - 30.0, 5.0, true);                    // ref :169
    *ties_out = ties;
    return fine_best;
}

// The slots of one pass (OpvTieStage, pinned host memory; called from the host function opv_process enqueues - no HIP call in
// here). One tied stream: its candidates' windows are shared out over threads (opv_offset_candidate_energy). Several: the
// STREAMS are shared out, a candidate evaluated by one thread from start to end - a context whose inputs tie systematically
// (real-valued captures on every stream) keeps the host's cores busy instead of creating seven threads per candidate.
// Same numbers either way: windows are independent and their energies are added in the reference's order.
void opv_offset_decide_slots(OpvTieSlot* slots, uint32_t n) {
    auto one = [&](OpvTieSlot& sl, bool threads) {
        sl.ties = 0;
        if (sl.nsym == 0 || sl.nsym > 1000) return;          // (cannot happen: the device's decision stands)
        uint32_t ties = 0;
        double energies[134];
        const double est = opv_offset_decide_on_host(sl.iq, sl.nsym, sl.poly, sl.power, energies, &ties, threads);
        if (ties == 0) return;                                // (the same polynomial gives the same contenders: not expected)
        std::memcpy(sl.energies, energies, sizeof energies);
        sl.est = est;
        sl.ties = ties;
    };
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 8) nt = 8;
    if (nt > n) nt = n;
    if (nt < 2) {
        for (uint32_t i = 0; i < n; ++i) one(slots[i], true);
        return;
    }
    std::atomic<uint32_t> next{0};
    auto worker = [&] {
        for (uint32_t i; (i = next.fetch_add(1)) < n;) one(slots[i], false);
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; ++t) {
        try {
            pool.emplace_back(worker);
        } catch (...) {                                       // no more threads to be had: the ones there are share the slots
            break;
        }
    }
    worker();
    for (auto& th : pool) th.join();
}

// Does THIS process's libm reproduce the energy the reference's evaluation gives (glibc 2.35, x86-64; the pinned value is
// checked against the test suite's CPU restatement of estimate_offset, itself bit-identical to the compiled reference:
// tests/test_capi_and_host.py)? 1000 windows of a fixed pseudo-random int16 sequence at +1425 Hz (a coarse candidate, so
// that an energy table of the whole search holds the same number): phases run to ~1.6e3 rad, every quadrant and
// range-reduction path the real evaluation takes. ~3 ms, once per process.
bool opv_offset_host_libm_matches_reference() {
    static const bool ok = [] {
        static int16_t iq[2 * 40000];
        uint32_t v = 0x2545F491u;
        for (int k = 0; k < 2 * 40000; ++k) {
            v = v * 1664525u + 1013904223u;
            iq[k] = (int16_t)((int32_t)(v >> 16) % 4001 - 2000);
        }
        const double e = opv_offset_candidate_energy(iq, 1000, OPV_OFFSET_PROBE_HZ, true);
        return e == OPV_OFFSET_PROBE_ENERGY;
    }();
    return ok;
}
