"""Experiment (CPU): does a closed-form (per-symbol) evaluation of the coherent demodulator's phases
stay on the reference's trajectory, although that loop never locks? numpy model of the GPU formulation
vs the oracle (bit-identical to the reference)."""
import sys
import numpy as np
sys.path.insert(0, "tests")
from oracle_lib import Oracle, impair

O = Oracle()
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 100
PI = np.pi; TWO_PI = 2 * np.pi; FS = 2168000.0; FDEV = 13550.0; SR = FS / 40

def wrap(p):
    while p > PI: p -= TWO_PI
    while p < -PI: p += TWO_PI
    return p

def model(iq, fo, afc_alpha=0.001, pll_bw=50.0):
    z = iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)
    wn = pll_bw * TWO_PI; pa = 2 * 0.707 * wn / SR; pb = wn * wn / (SR * SR)
    cp = ph1 = ph2 = lf = 0.0; prev = 0j
    i = np.arange(40.0)
    nsym = z.size // 40
    soft = np.empty(nsym)
    for k in range(nsym):
        inc1 = TWO_PI * (-FDEV + fo) / FS; inc2 = TWO_PI * (FDEV + fo) / FS
        s = z[40 * k: 40 * k + 40]
        c1 = np.sum(s * np.exp(-1j * ((cp + ph1) + i * (lf + inc1))))
        c2 = np.sum(s * np.exp(-1j * ((cp + ph2) + i * (lf + inc2))))
        ph1 = wrap(ph1 + 40 * inc1); ph2 = wrap(ph2 + 40 * inc2); cp = wrap(cp + 40 * lf)
        soft[k] = c2.real - c1.real
        dom = c1 if abs(c1) ** 2 > abs(c2) ** 2 else c2
        mag = abs(dom)
        pe = dom.imag / mag if mag > 1e-10 else 0.0
        lf = min(max(lf + pb * pe, -0.1), 0.1)
        cp += pa * pe
        if k > 0:
            pd = np.angle(dom * np.conj(prev))
            fo = min(max(fo + afc_alpha * pd * SR / TWO_PI, -2000.0), 2000.0)
        prev = dom
    return soft

base = O.modulate(O.bert_frames(NF))
for name, cap in (("clean", base), ("16dB+700Hz", impair(base, 2000, 700, 16, seed=2)), ("6dB-2000Hz", impair(base, 2000, -2000, 6, seed=4))):
    est = O.estimate_offset(cap)
    ref, d, ex = O.coherent_demodulate(cap, est, extra=True)
    got = model(cap, est)
    err = np.abs(got - ref) / np.mean(np.abs(ref))
    q = [err[: len(err) * f // 4 or 1].max() for f in (1, 2, 3, 4)]
    print(name, "nsym", len(ref), "max rel err by quarter", ["%.2e" % x for x in q], "loop_freq range", ex[:, 1].min(), ex[:, 1].max(), "fo end", ex[-1, 2])

if len(sys.argv) > 2:
    cap = base
    est = O.estimate_offset(cap)
    ref, d, ex = O.coherent_demodulate(cap, est, extra=True)
    got = model(cap[: 2 * 40 * 4000], est)
    err = np.abs(got - ref[:4000]) / np.mean(np.abs(ref))
    for k in list(range(0, 12)) + list(range(20, 4000, 200)):
        print(k, "%.3e" % err[k], ref[k], got[k], ex[k])
