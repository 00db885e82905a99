"""Experiment (CPU, oracle only): how fast does the MSK front-end recurrence forget its state?
Runs the streaming chunk chain from the true state and from perturbed / cold states."""
import sys, copy, ctypes as C
import numpy as np
sys.path.insert(0, "tests")
from oracle_lib import Oracle, impair, _Demod, CHUNK_SAMPLES

O = Oracle()
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 120
EBN0 = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0
iq = O.modulate(O.bert_frames(NF))
iq = impair(iq, amp=2000.0, f0_hz=700.0, ebn0_db=EBN0, seed=3)
n = iq.size // 2

def clone(d):
    e = _Demod(); C.memmove(C.byref(e), C.byref(d), C.sizeof(d)); return e

def run_chain(d, start, nchunks):
    """returns list of (start, state-copy-at-chunk-start), and softs per chunk"""
    recs = []; softs = []
    for _ in range(nchunks):
        if n - start < CHUNK_SAMPLES: break
        recs.append((start, clone(d)))
        s = O.demodulate(d, iq[2*start:2*(start+CHUNK_SAMPLES)])
        softs.append(s)
        lo = d.leftover
        start += CHUNK_SAMPLES - lo if 0 < lo < CHUNK_SAMPLES else CHUNK_SAMPLES
    return recs, softs

d = O.new_demod()
d.freq_offset = O.estimate_offset(iq[:2*CHUNK_SAMPLES])
print("est offset", d.freq_offset)
true_recs, true_softs = run_chain(d, 0, 10**9)
print("chunks", len(true_recs))
msoft = np.mean(np.abs(np.concatenate(true_softs)))

def compare(recs, softs, c0, label):
    print(label)
    for k, ((st, dd), s) in enumerate(zip(recs, softs)):
        ts, td = true_recs[c0 + k]
        ssame = len(s) == len(true_softs[c0+k])
        ds = np.max(np.abs(s - true_softs[c0+k]))/msoft if ssame else float('nan')
        if k < 12 or k % 5 == 0:
            print(f"  chunk {c0+k:3d} dstart {st-ts:+d} dmu {dd.mu-td.mu:+.3e} dtf {dd.timing_freq-td.timing_freq:+.3e} dfo {dd.freq_offset-td.freq_offset:+.3e} dsoft {ds:.3e}")

# 1. small perturbation on the true grid
c0 = 10
st, d0 = true_recs[c0]
d1 = clone(d0); d1.mu += 1e-3; d1.timing_freq += 1e-6; d1.freq_offset += 1.0
recs, softs = run_chain(d1, st, 80)
compare(recs, softs, c0, "perturbed on true grid (mu+1e-3, tf+1e-6, fo+1)")

# 2. cold start on the true grid position (mu=0, tf=0, fo = stream estimate, prev=0)
d2 = O.new_demod(); d2.freq_offset = true_recs[0][1].freq_offset
recs, softs = run_chain(d2, st, 80)
compare(recs, softs, c0, "cold state at a true chunk start")

# 3. cold start 17 samples off the true grid
d3 = O.new_demod(); d3.freq_offset = true_recs[0][1].freq_offset
recs, softs = run_chain(d3, st + 17, 80)
print("cold start 17 samples off the grid: symbol-boundary position (start+mu) mod 40 vs truth")
for k, (s_, dd) in enumerate(recs):
    ts, td = true_recs[c0 + k]
    if k < 12 or k % 5 == 0:
        print(f"  k {k:3d} start-ts {s_-ts:+d}  boundary diff mod 40: {((s_+dd.mu)-(ts+td.mu)) % 40:.6f}  dtf {dd.timing_freq-td.timing_freq:+.3e} dfo {dd.freq_offset-td.freq_offset:+.3e}")
