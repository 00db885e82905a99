/* oracle/opv_oracle.c — TEST INFRASTRUCTURE ONLY (see opv_oracle.h for the rules).
 *
 * CPU restatement of the reference's MSK transmit + receive chains. Every routine cites
 * the reference lines it follows ("ref:" = /root/reference/src/opv-demod.cpp unless
 * another file is named). The arithmetic order of the reference is kept on purpose:
 * sequential sums, per-sample libm sin/cos, phase accumulation by repeated addition —
 * that is what makes the result bit-identical to the compiled reference and therefore a
 * usable checker for the GPU path. Build with -ffp-contract=off (oracle/Makefile).
 *
 * Parity: PINNED against the compiled reference (tests/test_oracle_golden.py).
 */
#include "opv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORO_PI 3.14159265358979323846 /* ref:43 */
static const double TWO_PI = 2.0 * ORO_PI;            /* ref:44 */
static const double FS = 2168000.0;                   /* ref:40 */
static const double FDEV = 13550.0;                   /* ref:42 */
static const double SYM_RATE = 2168000.0 / 40.0;      /* ref:41 */
#define SYNC_WORD 0x02B8DBu                           /* ref:46 */

static inline int parity8(unsigned v) { return __builtin_parity(v & 0xFFu); }

/* ===================================== transmit ===================================== */

/* opv-mod.cpp:82-90 */
static int b40_digit(char c) {
    if (c >= 'A' && c <= 'Z') return c - 'A' + 1;
    if (c >= 'a' && c <= 'z') return c - 'a' + 1;
    if (c >= '0' && c <= '9') return c - '0' + 27;
    if (c == '-') return 37;
    if (c == '/') return 38;
    if (c == '.') return 39;
    return 0;
}

/* opv-mod.cpp:63-79 — first character ends up least significant; 48-bit big-endian */
void oro_base40_encode(const char* callsign, uint8_t out6[6]) {
    size_t len = strlen(callsign);
    if (len > 9) len = 9; /* opv-mod.cpp:451-454 */
    uint64_t v = 0;
    for (size_t k = len; k-- > 0;) v = v * 40u + (uint64_t)b40_digit(callsign[k]);
    for (int b = 0; b < 6; ++b) out6[b] = (uint8_t)(v >> (8 * (5 - b)));
}

/* opv-mod.cpp:339-361 */
void oro_bert_frame(const char* callsign, uint32_t token, uint32_t frame_num,
                    uint8_t out[ORO_FRAME_BYTES]) {
    memset(out, 0, ORO_FRAME_BYTES);
    oro_base40_encode(callsign, out);
    out[6] = (uint8_t)(token >> 16);
    out[7] = (uint8_t)(token >> 8);
    out[8] = (uint8_t)token;
    for (uint32_t i = 0; i < ORO_FRAME_BYTES - 12; ++i) out[12 + i] = (uint8_t)(frame_num + i);
}

/* CCSDS randomiser, x^8+x^7+x^5+x^3+1 style taps on bits 7,6,4,2; state 0xFF per frame.
 * opv-mod.cpp:97-113 and, identically, ref:887-893. */
void oro_lfsr_table(uint8_t out[ORO_FRAME_BYTES]) {
    uint8_t st = 0xFF;
    for (int i = 0; i < ORO_FRAME_BYTES; ++i) {
        uint8_t o = 0;
        for (int b = 7; b >= 0; --b) {
            o |= (uint8_t)(((st >> 7) & 1u) << b);
            uint8_t fb = (uint8_t)(((st >> 7) ^ (st >> 6) ^ (st >> 4) ^ (st >> 2)) & 1u);
            st = (uint8_t)((st << 1) | fb);
        }
        out[i] = o;
    }
}

/* opv-mod.cpp:159-213 (randomise :166-169, encode :186-196, interleave :142-153) */
void oro_encode_frame(const uint8_t payload[ORO_FRAME_BYTES], uint8_t coded[ORO_CODED_BITS]) {
    uint8_t rnd[ORO_FRAME_BYTES], tab[ORO_FRAME_BYTES], lin[ORO_CODED_BITS];
    oro_lfsr_table(tab);
    for (int i = 0; i < ORO_FRAME_BYTES; ++i) rnd[i] = payload[i] ^ tab[i];

    unsigned sr = 0; /* encoder register cleared per frame, opv-mod.cpp:161 */
    size_t o = 0;
    for (int byte = ORO_FRAME_BYTES - 1; byte >= 0; --byte) {      /* last byte first */
        for (int bit = 7; bit >= 0; --bit) {                       /* MSB first       */
            unsigned in = (rnd[byte] >> bit) & 1u;
            unsigned st = (in << 6) | sr;                          /* opv-mod.cpp:125 */
            lin[o++] = (uint8_t)parity8(st & 0x4F);                /* G1, :129        */
            lin[o++] = (uint8_t)parity8(st & 0x6D);                /* G2, :130        */
            sr = ((sr << 1) | in) & 0x3F;                          /* :131            */
        }
    }
    for (size_t i = 0; i < ORO_CODED_BITS; ++i) coded[oro_deinterleave_addr(i)] = lin[i]; /* :144-151 */
}

void oro_mod_reset(oro_mod* m) { m->ph1 = 0.0; m->ph2 = 0.0; m->t = 0; m->bn = 1; } /* opv-mod.cpp:221-226 */

/* opv-mod.cpp:228-284. Behaviour (Appendix A of SURVEY.md): d=+1 for bit 0, -1 for bit 1;
 * bit 0 drives tone 1 with sign T, bit 1 drives tone 2 with sign (+/-)T by symbol parity;
 * T'=d*T (or +1 right after reset); both NCOs free-run and wrap to [-pi, pi]. */
void oro_mod_symbol(oro_mod* m, int bit, int16_t iq[2 * ORO_SPS]) {
    const int d = bit ? -1 : 1;                                    /* :232 */
    const int t_next = (m->t == 0) ? 1 : d * m->t;                 /* :234-239 */
    int s1 = 0, s2 = 0;
    if (d == 1) {
        s1 = m->t;                                                 /* :241-251 (0 when T==0) */
    } else {
        const int neg_enc = (m->bn == 0) ? -1 : 1;                 /* :242-245 */
        s2 = neg_enc * m->t;                                       /* :253-257 */
    }
    const double inc1 = TWO_PI * (-FDEV) / FS;                     /* :259 (F1 = -13550) */
    const double inc2 = TWO_PI * (+FDEV) / FS;                     /* :260 */
    for (int i = 0; i < ORO_SPS; ++i) {
        const double sn1 = sin(m->ph1), cs1 = cos(m->ph1);
        const double sn2 = sin(m->ph2), cs2 = cos(m->ph2);
        const double I = (double)s1 * sn1 + (double)s2 * sn2;      /* :268 */
        const double Q = (double)s1 * cs1 + (double)s2 * cs2;      /* :269 */
        iq[2 * i] = (int16_t)(16383.0 * I);                        /* :271 truncation */
        iq[2 * i + 1] = (int16_t)(16383.0 * Q);                    /* :272 */
        m->ph1 += inc1;
        m->ph2 += inc2;
        while (m->ph1 > ORO_PI) m->ph1 -= TWO_PI;                  /* :276-279 */
        while (m->ph1 < -ORO_PI) m->ph1 += TWO_PI;
        while (m->ph2 > ORO_PI) m->ph2 -= TWO_PI;
        while (m->ph2 < -ORO_PI) m->ph2 += TWO_PI;
    }
    m->t = t_next;                                                 /* :282 */
    m->bn = 1 - m->bn;                                             /* :283 */
}

size_t oro_modulated_len(size_t nframes) {
    return nframes * (size_t)ORO_FRAME_SYMBOLS * ORO_SPS + 100u * ORO_SPS; /* opv-mod.cpp:528-529 */
}

/* One whole opv-mod run: a single modulator reset (opv-mod.cpp:474/:506), then per frame
 * 24 sync bits MSB first (:315-321) and the 2144 interleaved coded bits (:332-335), then
 * 100 symbols of zeros (:528-529). Returns samples written. */
size_t oro_modulate_frames(const uint8_t* frames, size_t nframes, int16_t* iq) {
    oro_mod m;
    oro_mod_reset(&m);
    uint8_t coded[ORO_CODED_BITS];
    size_t w = 0;
    for (size_t f = 0; f < nframes; ++f) {
        oro_encode_frame(frames + f * ORO_FRAME_BYTES, coded);
        for (int b = ORO_SYNC_BITS - 1; b >= 0; --b) {
            oro_mod_symbol(&m, (int)((SYNC_WORD >> b) & 1u), iq + 2 * w);
            w += ORO_SPS;
        }
        for (int k = 0; k < ORO_CODED_BITS; ++k) {
            oro_mod_symbol(&m, coded[k], iq + 2 * w);
            w += ORO_SPS;
        }
    }
    memset(iq + 2 * w, 0, sizeof(int16_t) * 2u * 100u * ORO_SPS);
    w += 100u * ORO_SPS;
    return w;
}

/* ===================================== receive ====================================== */

void oro_demod_init(oro_demod* d) { /* ref:110-119 */
    memset(d, 0, sizeof(*d));
    d->afc_alpha = 0.001;
    d->alpha_timing = 0.005;
    d->beta_timing = 0.00001;
}

/* ref:143-159 — energy of one candidate offset. LO phases start at 0 and accumulate over
 * every sample (never wrapped); symbols are fixed 40-sample windows from sample 0. */
static double candidate_energy(const int16_t* iq, size_t nsym, double offset) {
    double p1 = 0.0, p2 = 0.0;
    const double i1 = TWO_PI * (-FDEV + offset) / FS; /* ref:137 */
    const double i2 = TWO_PI * (+FDEV + offset) / FS; /* ref:138 */
    double total = 0.0;
    for (size_t s = 0; s < nsym; ++s) {
        double a1r = 0, a1i = 0, a2r = 0, a2i = 0;
        for (size_t i = 0; i < ORO_SPS; ++i) {
            const size_t k = s * ORO_SPS + i;
            const double xr = iq[2 * k], xi = iq[2 * k + 1];
            const double c1 = cos(p1), s1 = sin(p1), c2 = cos(p2), s2 = sin(p2);
            /* x * conj(lo): (xr + j xi)(c - j s)  (ref:151-152) */
            a1r += xr * c1 + xi * s1;
            a1i += xi * c1 - xr * s1;
            a2r += xr * c2 + xi * s2;
            a2i += xi * c2 - xr * s2;
            p1 += i1;
            p2 += i2;
        }
        total += (a1r * a1r + a1i * a1i) + (a2r * a2r + a2i * a2i); /* ref:158 */
    }
    return total;
}

/* ref:131-202 */
double oro_estimate_offset(const int16_t* iq, size_t n, double* energies) {
    const size_t test = n < (size_t)ORO_SPS * 1000u ? n : (size_t)ORO_SPS * 1000u; /* ref:141 */
    const size_t nsym = test / ORO_SPS;
    double best = 0.0, best_e = 0.0;
    size_t k = 0;
    for (double off = -1500; off <= 1500; off += 25) { /* ref:135 */
        const double e = candidate_energy(iq, nsym, off);
        if (energies) energies[k] = e;
        ++k;
        if (e > best_e) { best_e = e; best = off; } /* strict >, first maximum wins (ref:161) */
    }
    double fine = best;
    for (double off = best - 30; off <= best + 30; off += 5) { /* ref:169 */
        const double e = candidate_energy(iq, nsym, off);
        if (energies) energies[k] = e;
        ++k;
        if (e > best_e) { best_e = e; fine = off; } /* ref:195-198 */
    }
    return fine;
}

/* ref:122-128 — linear interpolation with index clamped to [0, n-2] */
static inline void lerp_iq(const int16_t* iq, size_t n, double idx, double* re, double* im) {
    if (idx < 0) idx = 0;
    if (idx >= (double)(n - 1)) idx = (double)(n - 2);
    const size_t i = (size_t)idx;
    const double f = idx - (double)i;
    const double g = 1.0 - f;
    *re = (double)iq[2 * i] * g + (double)iq[2 * i + 2] * f;
    *im = (double)iq[2 * i + 1] * g + (double)iq[2 * i + 3] * f;
}

static inline double clampd(double v, double lo, double hi) { return v < lo ? lo : (hi < v ? hi : v); }

/* ref:206-329 */
size_t oro_demodulate(oro_demod* d, const int16_t* iq, size_t n, double* soft, size_t cap) {
    size_t ns = 0;
    double inc1 = TWO_PI * (-FDEV + d->freq_offset) / FS; /* ref:210 */
    double inc2 = TWO_PI * (+FDEV + d->freq_offset) / FS; /* ref:211 */
    const double EL = ORO_SPS / 4.0;                      /* ref:214 */
    double pos = d->mu;                                   /* ref:217 */

    while (pos + (double)ORO_SPS + EL < (double)n) {      /* ref:221 */
        double c1r = 0, c1i = 0, c2r = 0, c2i = 0;        /* on-time */
        double e1r = 0, e1i = 0, e2r = 0, e2i = 0;        /* early   */
        double l1r = 0, l1i = 0, l2r = 0, l2i = 0;        /* late    */
        double ph1 = d->phase_f1, ph2 = d->phase_f2;      /* ref:228 */

        for (size_t i = 0; i < ORO_SPS; ++i) {
            const double p_on = pos + (double)i;          /* ref:232 */
            const double p_e = p_on - EL;
            const double p_l = p_on + EL;
            double onr, oni, er, ei, lr, li;
            lerp_iq(iq, n, p_on, &onr, &oni);
            if (p_e >= 0) lerp_iq(iq, n, p_e, &er, &ei);  /* ref:237 */
            else { er = iq[0]; ei = iq[1]; }
            lerp_iq(iq, n, p_l, &lr, &li);
            const double k1 = cos(ph1), s1 = sin(ph1);    /* ref:240-241 */
            const double k2 = cos(ph2), s2 = sin(ph2);
            c1r += onr * k1 + oni * s1;  c1i += oni * k1 - onr * s1;   /* ref:243-248 */
            c2r += onr * k2 + oni * s2;  c2i += oni * k2 - onr * s2;
            e1r += er * k1 + ei * s1;    e1i += ei * k1 - er * s1;
            e2r += er * k2 + ei * s2;    e2i += ei * k2 - er * s2;
            l1r += lr * k1 + li * s1;    l1i += li * k1 - lr * s1;
            l2r += lr * k2 + li * s2;    l2i += li * k2 - lr * s2;
            ph1 += inc1;                                  /* ref:250-251 */
            ph2 += inc2;
        }
        d->phase_f1 = ph1;
        d->phase_f2 = ph2;
        while (d->phase_f1 > ORO_PI) d->phase_f1 -= TWO_PI;  /* ref:259-262 */
        while (d->phase_f1 < -ORO_PI) d->phase_f1 += TWO_PI;
        while (d->phase_f2 > ORO_PI) d->phase_f2 -= TWO_PI;
        while (d->phase_f2 < -ORO_PI) d->phase_f2 += TWO_PI;

        const double en1 = c1r * c1r + c1i * c1i;         /* ref:264-265 */
        const double en2 = c2r * c2r + c2i * c2i;
        if (soft && ns < cap) soft[ns] = en2 - en1;       /* ref:268 */
        ++ns;

        double ee, el;                                    /* ref:271-280 */
        if (en1 > en2) { ee = e1r * e1r + e1i * e1i; el = l1r * l1r + l1i * l1i; }
        else           { ee = e2r * e2r + e2i * e2i; el = l2r * l2r + l2i * l2i; }
        const double ted = (el - ee) / (el + ee + 1e-10);

        d->timing_freq += d->beta_timing * ted;           /* ref:283-286 */
        d->timing_freq = clampd(d->timing_freq, -0.1, 0.1);
        double adj = d->alpha_timing * ted + d->timing_freq;
        adj = clampd(adj, -2.0, 2.0);

        if (ns > 1) {                                     /* ref:289 — per-call symbol count */
            double dr, di, pr, pi;
            if (en1 > en2) { dr = c1r; di = c1i; pr = d->prev1_re; pi = d->prev1_im; }
            else           { dr = c2r; di = c2i; pr = d->prev2_re; pi = d->prev2_im; }
            /* dom * conj(prev) (ref:299) */
            const double zr = dr * pr + di * pi;
            const double zi = di * pr - dr * pi;
            const double pd = atan2(zi, zr);
            const double ferr = pd * SYM_RATE / TWO_PI;   /* ref:300 */
            d->freq_offset += d->afc_alpha * ferr;        /* ref:302-303 */
            d->freq_offset = clampd(d->freq_offset, -2000.0, 2000.0);
            inc1 = TWO_PI * (-FDEV + d->freq_offset) / FS;
            inc2 = TWO_PI * (+FDEV + d->freq_offset) / FS;
        }
        d->prev1_re = c1r; d->prev1_im = c1i;             /* ref:309-310 */
        d->prev2_re = c2r; d->prev2_im = c2i;

        pos += (double)ORO_SPS + adj;                     /* ref:313 */
    }
    const size_t used = (size_t)pos;                      /* ref:318-328 */
    d->mu = pos - (double)used;
    d->leftover = n - used;
    return ns;
}

/* ------------------------- coherent demodulator (ref:365-572) ----------------------- */
void oro_coh_init(oro_coh* d) { /* ref:367-376 */
    memset(d, 0, sizeof(*d));
    d->afc_alpha = 0.001;
    d->pll_alpha = 0.01;
    d->pll_beta = 0.001;
}

void oro_coh_set_pll_bandwidth(oro_coh* d, double bw) { /* ref:551-558 */
    const double wn = bw * TWO_PI;
    const double zeta = 0.707;
    d->pll_alpha = 2.0 * zeta * wn / SYM_RATE;
    d->pll_beta = wn * wn / (SYM_RATE * SYM_RATE);
}

/* ref:455-543. Fixed 40-sample symbol grid from sample 0 (no timing recovery); every sample is
 * de-rotated by the carrier phase, which advances by loop_freq per SAMPLE; std::complex
 * products are written out in the (ac-bd, ad+bc) order libstdc++ evaluates them in. */
size_t oro_coh_demodulate(oro_coh* d, const int16_t* iq, size_t n, double* soft, size_t cap,
                          double* extra, size_t cap_extra) {
    double inc1 = TWO_PI * (-FDEV + d->freq_offset) / FS; /* ref:459 */
    double inc2 = TWO_PI * (+FDEV + d->freq_offset) / FS; /* ref:460 */
    const size_t nsym = n / ORO_SPS;                      /* ref:462 */
    for (size_t sym = 0; sym < nsym; ++sym) {
        double c1r = 0, c1i = 0, c2r = 0, c2i = 0;
        for (size_t i = 0; i < ORO_SPS; ++i) {
            const size_t k = sym * ORO_SPS + i;
            const double sr = iq[2 * k], si = iq[2 * k + 1];
            const double rr = cos(d->carrier_phase), ri = -sin(d->carrier_phase); /* ref:470 */
            const double xr = sr * rr - si * ri;          /* ref:471 corrected = s * phase_rot */
            const double xi = sr * ri + si * rr;
            const double k1 = cos(d->phase_f1), s1 = sin(d->phase_f1);            /* ref:474-475 */
            const double k2 = cos(d->phase_f2), s2 = sin(d->phase_f2);
            c1r += xr * k1 - xi * (-s1);  c1i += xr * (-s1) + xi * k1;            /* ref:477-478 */
            c2r += xr * k2 - xi * (-s2);  c2i += xr * (-s2) + xi * k2;
            d->phase_f1 += inc1;                          /* ref:480-481 */
            d->phase_f2 += inc2;
            d->carrier_phase += d->loop_freq;             /* ref:484 */
        }
        while (d->phase_f1 > ORO_PI) d->phase_f1 -= TWO_PI;          /* ref:488-493 */
        while (d->phase_f1 < -ORO_PI) d->phase_f1 += TWO_PI;
        while (d->phase_f2 > ORO_PI) d->phase_f2 -= TWO_PI;
        while (d->phase_f2 < -ORO_PI) d->phase_f2 += TWO_PI;
        while (d->carrier_phase > ORO_PI) d->carrier_phase -= TWO_PI;
        while (d->carrier_phase < -ORO_PI) d->carrier_phase += TWO_PI;

        const double en1 = c1r * c1r + c1i * c1i;         /* ref:496-497 */
        const double en2 = c2r * c2r + c2i * c2i;
        if (soft && sym < cap) soft[sym] = c2r - c1r;     /* ref:502-507 */

        double dr, di;                                    /* ref:512 */
        if (en1 > en2) { dr = c1r; di = c1i; } else { dr = c2r; di = c2i; }
        const double mag = hypot(dr, di);                 /* ref:515 std::abs */
        double pe = 0;
        if (mag > 1e-10) pe = di / mag;                   /* ref:517-522 */
        d->loop_freq += d->pll_beta * pe;                 /* ref:526-527 */
        d->carrier_phase += d->pll_alpha * pe;
        d->loop_freq = clampd(d->loop_freq, -0.1, 0.1);   /* ref:530 */

        if (sym > 0) {                                    /* ref:535-543 */
            const double zr = dr * d->prev_re - di * (-d->prev_im);
            const double zi = dr * (-d->prev_im) + di * d->prev_re;
            const double pd = atan2(zi, zr);
            const double ferr = pd * SYM_RATE / TWO_PI;
            d->freq_offset += d->afc_alpha * ferr;
            d->freq_offset = clampd(d->freq_offset, -2000.0, 2000.0);
            inc1 = TWO_PI * (-FDEV + d->freq_offset) / FS;
            inc2 = TWO_PI * (+FDEV + d->freq_offset) / FS;
        }
        d->prev_re = dr; d->prev_im = di;                 /* ref:545 */
        if (extra && sym < cap_extra) {
            extra[3 * sym] = d->carrier_phase; extra[3 * sym + 1] = d->loop_freq;
            extra[3 * sym + 2] = d->freq_offset;
        }
    }
    return nsym;
}

/* ref:591-607 */
void oro_tracker_init(oro_tracker* t) {
    memset(t, 0, sizeof(*t));
    t->state = ORO_HUNTING;
    for (int i = 0; i < ORO_SYNC_BITS; ++i) {
        const int bit = (int)((SYNC_WORD >> (ORO_SYNC_BITS - 1 - i)) & 1u);
        t->pattern[i] = bit ? -1.0 : +1.0;                /* ref:597-600 */
    }
}

/* ref:743-757 — oldest-to-newest over the 24-entry ring */
static double sync_corr(const oro_tracker* t, double* raw) {
    double sum = 0.0, energy = 0.0;
    for (size_t i = 0; i < ORO_SYNC_BITS; ++i) {
        const double s = t->ring[(t->ring_idx + i) % ORO_SYNC_BITS];
        sum += s * t->pattern[i];
        energy += fabs(s);
    }
    *raw = sum;
    if (energy < 100.0) return 0.0;                       /* ref:755, :786 */
    return sum / energy;
}

static void push_event(oro_event* ev, size_t* n_ev, size_t cap, int kind, int count,
                       size_t sym, double corr, double raw) {
    if (!n_ev) return;
    if (ev && *n_ev < cap) {
        ev[*n_ev].kind = kind; ev[*n_ev].count = count; ev[*n_ev].sym_idx = sym;
        ev[*n_ev].corr = corr; ev[*n_ev].raw = raw;
    }
    ++*n_ev;
}

static int release(oro_tracker* t, double* payload, double* quality) {
    if (payload) memcpy(payload, t->pending, sizeof(double) * ORO_CODED_BITS);
    if (quality) *quality = t->quality;
    t->total_frames++;
    t->pending_n = 0;
    t->collecting = 0;
    return 1;
}

/* ref:615-736. The 6504-entry circ_buf_ of the reference (ref:623-624) is never read and
 * is therefore not modelled. */
int oro_tracker_process(oro_tracker* t, double soft, size_t sym_idx, double* payload,
                        double* quality, oro_event* ev, size_t* n_ev, size_t cap_ev) {
    int ready = 0;
    t->ring[t->ring_idx] = soft;                          /* ref:619-620 */
    t->ring_idx = (t->ring_idx + 1) % ORO_SYNC_BITS;
    t->total_symbols++;
    if (t->collecting && t->pending_n < ORO_CODED_BITS)   /* ref:628-630 */
        t->pending[t->pending_n++] = soft;
    t->since_sync++;                                      /* ref:632 */

    switch (t->state) {
    case ORO_HUNTING: {
        if (t->total_symbols < ORO_SYNC_BITS) break;      /* ref:637 */
        double raw;
        const double nc = sync_corr(t, &raw);
        if (raw >= 5000.0 && nc >= 0.85) {                /* ref:642, :783,:785 */
            t->state = ORO_VERIFYING;
            t->quality = nc;
            t->since_sync = 0;
            t->collecting = 1;
            t->pending_n = 0;
            push_event(ev, n_ev, cap_ev, ORO_EV_HUNT_TO_VERIFY, 0, sym_idx, nc, raw);
        }
        break;
    }
    case ORO_VERIFYING: {
        if (t->since_sync >= ORO_CODED_BITS) {            /* ref:658 */
            /* a VERIFYING frame always has exactly 2144 collected symbols */
            ready = release(t, payload, quality);
            t->state = ORO_LOCKED;
            t->misses = 0;
            push_event(ev, n_ev, cap_ev, ORO_EV_VERIFY_TO_LOCK, t->total_frames, sym_idx, 0, 0);
        }
        break;
    }
    case ORO_LOCKED: {
        if (t->since_sync == ORO_FRAME_SYMBOLS) {         /* ref:684 */
            double raw;
            const double c = sync_corr(t, &raw);
            if (c >= 0.70) {                              /* ref:688, :784 */
                t->misses = 0;
                t->quality = c;
                t->collecting = 1;
                t->pending_n = 0;
                push_event(ev, n_ev, cap_ev, ORO_EV_SYNC_OK, 0, sym_idx, c, raw);
            } else {
                t->misses++;
                push_event(ev, n_ev, cap_ev, ORO_EV_SYNC_MISS, t->misses, sym_idx, c, raw);
                if (t->misses >= 5) {                     /* ref:702, :60 */
                    t->state = ORO_HUNTING;
                    t->collecting = 0;
                    push_event(ev, n_ev, cap_ev, ORO_EV_LOST_LOCK, 0, sym_idx, c, raw);
                    break;                                /* ref:706: counter NOT reset */
                }
                t->quality = c;                           /* flywheel, ref:709-712 */
                t->collecting = 1;
                t->pending_n = 0;
            }
            t->since_sync = 0;                            /* ref:716 */
        }
        if (t->collecting && t->pending_n >= ORO_CODED_BITS) /* ref:720 */
            ready = release(t, payload, quality);
        break;
    }
    }
    return ready;
}

size_t oro_deinterleave_addr(size_t i) {                  /* ref:792-795 */
    const size_t p = (i % 32) * 67 + (i / 32);
    return (p / 8) * 8 + (7 - p % 8);
}

/* ref:800-847 */
int oro_viterbi(const int* in, uint8_t* bits) {
    uint8_t dec[ORO_FRAME_BITS][64];
    int metric[64], next[64];
    for (int s = 0; s < 64; ++s) metric[s] = 0x7FFFFFFF;
    metric[0] = 0;
    for (int t = 0; t < ORO_FRAME_BITS; ++t) {
        const int g1 = in[2 * t], g2 = in[2 * t + 1];
        for (int s = 0; s < 64; ++s) {
            const int p0 = s >> 1, p1 = p0 + 32, inb = s & 1;
            const int f0 = (inb << 6) | p0, f1 = (inb << 6) | p1;
            const int bm0 = (parity8(f0 & 0x4F) ? 7 - g1 : g1) + (parity8(f0 & 0x6D) ? 7 - g2 : g2);
            const int bm1 = (parity8(f1 & 0x4F) ? 7 - g1 : g1) + (parity8(f1 & 0x6D) ? 7 - g2 : g2);
            const int m0 = metric[p0] < 0x7FFFFFF0 ? metric[p0] + bm0 : 0x7FFFFFFF; /* ref:826 */
            const int m1 = metric[p1] < 0x7FFFFFF0 ? metric[p1] + bm1 : 0x7FFFFFFF;
            if (m0 <= m1) { next[s] = m0; dec[t][s] = 0; }   /* ties -> lower predecessor */
            else          { next[s] = m1; dec[t][s] = 1; }
        }
        memcpy(metric, next, sizeof(metric));
    }
    int best = 0;
    for (int s = 1; s < 64; ++s) if (metric[s] < metric[best]) best = s; /* first minimum */
    int s = best;
    for (int t = ORO_FRAME_BITS - 1; t >= 0; --t) {
        bits[t] = (uint8_t)(s & 1);
        s = dec[t][s] == 0 ? (s >> 1) : (s >> 1) + 32;
    }
    return metric[best];
}

/* ref:854-898 */
int oro_frame_decode(const double* soft, uint8_t out[ORO_FRAME_BYTES], int* q_tap, int* d_tap,
                     uint8_t* bits_tap) {
    double scale = 0;
    for (size_t i = 0; i < ORO_CODED_BITS; ++i) scale += fabs(soft[i]);   /* ref:857 */
    scale /= ORO_CODED_BITS;
    if (scale < 1e-10) return -1;                                          /* ref:859 */

    int q[ORO_CODED_BITS], de[ORO_CODED_BITS];
    for (size_t i = 0; i < ORO_CODED_BITS; ++i) {
        const double nrm = (-soft[i] / scale) * 3.5 + 3.5;                 /* ref:864 */
        int v = (int)(nrm + 0.5);                                          /* truncation */
        q[i] = v < 0 ? 0 : (v > 7 ? 7 : v);
    }
    for (size_t i = 0; i < ORO_CODED_BITS; ++i) de[i] = q[oro_deinterleave_addr(i)]; /* ref:870-871 */

    uint8_t bits[ORO_FRAME_BITS];
    const int metric = oro_viterbi(de, bits);

    uint8_t tab[ORO_FRAME_BYTES];
    oro_lfsr_table(tab);
    for (size_t i = 0; i < ORO_FRAME_BYTES; ++i) {
        uint8_t b = 0;
        for (int j = 0; j < 8; ++j) b |= (uint8_t)(bits[ORO_FRAME_BITS - 1 - i * 8 - j] << j); /* ref:882 */
        out[i] = b ^ tab[i];                                               /* ref:894 */
    }
    if (q_tap) memcpy(q_tap, q, sizeof(q));
    if (d_tap) memcpy(d_tap, de, sizeof(de));
    if (bits_tap) memcpy(bits_tap, bits, sizeof(bits));
    return metric;
}

/* --------------------------- whole receiver, as main() drives it -------------------- */

typedef struct {
    oro_tracker trk;
    size_t total_symbols;
    oro_rx_out* out;
} rx_sink;

static void feed_symbols(rx_sink* k, const double* soft, size_t ns) {
    oro_rx_out* o = k->out;
    double payload[ORO_CODED_BITS], quality = 0;
    for (size_t i = 0; i < ns; ++i) {
        const size_t idx = k->total_symbols + i;          /* ref:1046 / :1187 */
        if (o->soft && o->n_soft < o->cap_soft) o->soft[o->n_soft] = soft[i];
        o->n_soft++;
        if (oro_tracker_process(&k->trk, soft[i], idx, payload, &quality, o->events, &o->n_events,
                                o->cap_events)) {
            uint8_t fr[ORO_FRAME_BYTES];
            const int metric = oro_frame_decode(payload, fr, NULL, NULL, NULL);
            if (metric >= 0) {                            /* ref:1052 */
                if (o->n_frames < o->cap_frames) {
                    if (o->frames) memcpy(o->frames + o->n_frames * ORO_FRAME_BYTES, fr, ORO_FRAME_BYTES);
                    if (o->metrics) o->metrics[o->n_frames] = metric;
                    if (o->quality) o->quality[o->n_frames] = quality;
                    if (o->frame_sym) o->frame_sym[o->n_frames] = idx;
                }
                o->n_frames++;
                if (metric == 0) o->n_perfect++;
            }
        }
    }
    k->total_symbols += ns;
}

static void note_chunk(oro_rx_out* o, const oro_demod* d, size_t ns) {
    if (o->chunk_state && o->n_chunks < o->cap_chunks) {
        double* c = o->chunk_state + 5 * o->n_chunks;
        c[0] = d->freq_offset; c[1] = d->timing_freq; c[2] = d->mu;
        c[3] = (double)d->leftover; c[4] = (double)ns;
    }
    o->n_chunks++;
}

int oro_receive(const int16_t* iq, size_t n_samples, const oro_rx_cfg* cfg, oro_rx_out* out) {
    double softbuf[4096];
    oro_demod dm;
    oro_demod_init(&dm);
    rx_sink sink;
    oro_tracker_init(&sink.trk);
    sink.total_symbols = 0;
    sink.out = out;
    out->n_frames = out->n_perfect = out->n_soft = out->n_events = out->n_chunks = 0;
    out->est_offset = NAN;

    if (cfg->streaming) {
        /* ref:995-1113. The chunk buffer is refilled to >= 86720 samples; after each call the
         * last `leftover` samples are kept, i.e. the next chunk starts `used` samples later. */
        if (cfg->have_init_offset) dm.freq_offset = cfg->init_offset;  /* ref:1004-1005 */
        dm.afc_alpha = cfg->afc_alpha;                                 /* ref:1009 */
        size_t start = 0;
        int first = 1;
        while (n_samples - start >= ORO_CHUNK_SAMPLES) {
            const int16_t* c = iq + 2 * start;
            if (first) {
                if (!cfg->have_init_offset) {                          /* ref:1030-1036 */
                    out->est_offset = oro_estimate_offset(c, ORO_CHUNK_SAMPLES, NULL);
                    dm.freq_offset = out->est_offset;
                }
                first = 0;
            }
            const size_t ns = oro_demodulate(&dm, c, ORO_CHUNK_SAMPLES, softbuf, 4096);
            note_chunk(out, &dm, ns);
            feed_symbols(&sink, softbuf, ns);
            const size_t lo = dm.leftover;                             /* ref:1070-1076 */
            if (lo > 0 && lo < ORO_CHUNK_SAMPLES) start += ORO_CHUNK_SAMPLES - lo;
            else start += ORO_CHUNK_SAMPLES;
        }
        if (n_samples > start) {                                       /* ref:1088-1113 */
            const size_t ns = oro_demodulate(&dm, iq + 2 * start, n_samples - start, softbuf, 4096);
            note_chunk(out, &dm, ns);
            feed_symbols(&sink, softbuf, ns);
        }
    } else {
        /* ref:1132-1206: estimate on the capture head, ONE demodulate over everything. */
        out->est_offset = oro_estimate_offset(iq, n_samples, NULL);    /* ref:1166 */
        dm.freq_offset = out->est_offset;
        dm.afc_alpha = cfg->afc_alpha;                                 /* ref:1172 */
        /* soft symbols are produced in one call; to bound memory, run the demodulator once
         * into a caller-sized buffer when available, else in a private one. */
        const size_t cap = n_samples / ORO_SPS + 16;
        double* tmp = (out->soft && out->cap_soft >= cap) ? out->soft : NULL;
        double* own = NULL;
        if (!tmp) { own = (double*)malloc(cap * sizeof(double)); tmp = own; }
        if (!tmp) return -1;
        size_t ns;
        if (cfg->coherent) {                                           /* ref:1144-1161 */
            oro_coh cd;
            oro_coh_init(&cd);
            cd.freq_offset = out->est_offset;
            cd.afc_alpha = cfg->afc_alpha;
            oro_coh_set_pll_bandwidth(&cd, cfg->pll_bw);
            ns = oro_coh_demodulate(&cd, iq, n_samples, tmp, cap, NULL, 0);
            dm.freq_offset = cd.freq_offset;                           /* final_offset, ref:1161 */
        } else {
            ns = oro_demodulate(&dm, iq, n_samples, tmp, cap);
        }
        note_chunk(out, &dm, ns);
        feed_symbols(&sink, tmp, ns); /* when tmp == out->soft the tap copy is onto itself */
        free(own);
    }
    out->final_freq_offset = dm.freq_offset;
    out->final_timing_freq = dm.timing_freq;
    out->final_state = sink.trk.state;
    return 0;
}
