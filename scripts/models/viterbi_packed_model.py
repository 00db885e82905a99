"""numpy model of k_frame_decode's packed ACS (round 4): the index algebra, checked against the oracle's ViterbiDecoder
restatement BEFORE the kernel was written. One trellis here (the kernel runs two, in the 16-bit halves of every register):
  * state s at time t lives in lane rotr6(s, t); step t's butterfly partner is lane ^ (1 << K), K = (5 - t) mod 6
  * branch metrics come from a 4-entry table per step, entry j = {bm(j), bm(j ^ flip)}, classes j = (e1 << 1) | e2
  * K < 4 (DPP phases): own = M + T[c].x, oth = M[partner] + T[c].y, raw = sign(own - oth - beta), beta = 1 - u
  * K >= 4 (permlane-swap phases): a = M + T[idx].x, b = M + T[idx].y, swap odd rows of a with even rows of b,
    raw = sign(a' - b' - 1), and the traceback un-flips odd rows
  * metric' = min of the two, 16-bit with a finite sentinel for unreachable states
  * traceback in lane space from per-lane decision bits
usage: python scripts/models/viterbi_packed_model.py [n_trials]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tests"))
SENT = 0x3FF0


def rotl6(x, r):
    r %= 6
    return ((x << r) | (x >> (6 - r))) & 63 if r else x


def par(x):
    return bin(x).count("1") & 1


def lane_tables():
    """per phase PH (= t mod 6) and lane: K, u, table index (own class for DPP phases; un-flipped / flipped for swap phases)"""
    tabs = []
    for ph in range(6):
        K = (5 - ph) % 6
        r = (ph + 1) % 6
        idx = np.zeros(64, np.int64)
        u = np.zeros(64, np.int64)
        for l in range(64):
            st = rotl6(l, r)                       # state this lane holds at time t + 1
            b0 = st & 1
            pown = (st >> 1) | (b0 << 5)
            f = (b0 << 6) | pown
            c = (par(f & 0x4F) << 1) | par(f & 0x6D)
            u[l] = (l >> K) & 1
            assert u[l] == b0
            idx[l] = (c ^ 3) if (K >= 4 and u[l]) else c
        tabs.append((K, u, idx))
    return tabs


def decode(q):
    tabs = lane_tables()
    M = np.full(64, SENT, np.int64)
    M[0] = 0
    raw_bits = np.zeros((1072, 64), np.int64)
    lanes = np.arange(64)
    for t in range(1072):
        K, u, idx = tabs[t % 6]
        sg1, sg2 = int(q[2 * t]), int(q[2 * t + 1])
        bm = np.array([(7 - sg1 if j & 2 else sg1) + (7 - sg2 if j & 1 else sg2) for j in range(4)])
        flip = 3 if K >= 4 else 1
        X, Y = bm[idx], bm[idx ^ flip]             # table entry idx = {bm(idx), bm(idx ^ flip)}
        if K < 4:
            own = M + X
            oth = M[lanes ^ (1 << K)] + Y
            raw = (own - oth - (1 - u)) < 0
            M = np.minimum(own, oth)
        else:
            a, b = M + X, M + Y
            a2, b2 = a.copy(), b.copy()
            odd = ((lanes >> K) & 1) == 1
            # swap: a's odd rows <-> b's even rows (K = 4: rows of 16; K = 5: halves of 32)
            a2[odd] = b[lanes[odd] ^ (1 << K)]
            b2[~odd] = a[lanes[~odd] ^ (1 << K)]
            raw = (a2 - b2 - 1) < 0
            M = np.minimum(a2, b2)
        assert M.max() < 65536 and M.min() >= 0
        raw_bits[t] = raw
    # first minimum in state order
    states = np.array([rotl6(l, 1072 % 6) for l in range(64)])
    order = np.lexsort((states, M))
    cur = int(order[0])
    best = int(M[cur])
    bits = np.zeros(1072, np.uint8)
    for t in range(1071, -1, -1):
        K = (5 - t) % 6
        b = (cur >> K) & 1
        bits[t] = b
        raw = int(raw_bits[t, cur])
        take_own = raw ^ b if K >= 4 else raw
        if not take_own:
            cur ^= 1 << K
    return best, bits


def decode_trace_forward(q):
    """the same trellis with TRACE-FORWARD pointers instead of decision bits (the kernel's final form): every lane carries the
    lane its survivor stood in at the last multiple of six steps; a step moves pointers exactly as it moves metrics (the
    winner's pointer, fetched from the same partner lane), every sixth step the pointer is filed as a 6-bit field and reset to
    the lane's own number. The walk back then hops six steps per field, and the six decoded bits of a hop are the bits of the
    lane it starts from (a step changes bit K of the lane only, K runs 0..5 in walk order from a multiple of six)."""
    tabs = lane_tables()
    M = np.full(64, SENT, np.int64)
    M[0] = 0
    lanes = np.arange(64)
    P = lanes.copy()
    fields = []
    for t in range(1072):
        K, u, idx = tabs[t % 6]
        sg1, sg2 = int(q[2 * t]), int(q[2 * t + 1])
        bm = np.array([(7 - sg1 if j & 2 else sg1) + (7 - sg2 if j & 1 else sg2) for j in range(4)])
        flip = 3 if K >= 4 else 1
        X, Y = bm[idx], bm[idx ^ flip]
        if K < 4:
            x = M + X
            y = M[lanes ^ (1 << K)] + Y
            raw = (x - y - (1 - u)) < 0
            Px, Py = P, P[lanes ^ (1 << K)]
        else:
            a, b = M + X, M + Y
            x, y = a.copy(), b.copy()
            Px, Py = P.copy(), P.copy()
            odd = ((lanes >> K) & 1) == 1
            x[odd] = b[lanes[odd] ^ (1 << K)]
            y[~odd] = a[lanes[~odd] ^ (1 << K)]
            Px[odd] = P[lanes[odd] ^ (1 << K)]
            Py[~odd] = P[lanes[~odd] ^ (1 << K)]
            raw = (x - y - 1) < 0
        M = np.minimum(x, y)
        P = np.where(raw, Px, Py)
        if t % 6 == 5 or t == 1071:
            fields.append(P.copy())
            P = lanes.copy()
    assert len(fields) == 179
    states = np.array([rotl6(l, 1072 % 6) for l in range(64)])
    cur = int(np.lexsort((states, M))[0])
    best = int(M[cur])
    walk = []                                      # decoded bits in walk order (t = 1071 first)
    walk += [(cur >> k) & 1 for k in (2, 3, 4, 5)]  # the 4-step tail group: K = 2, 3, 4, 5
    cur = int(fields[178][cur])
    for g in range(177, -1, -1):
        walk += [(cur >> k) & 1 for k in range(6)]
        cur = int(fields[g][cur])
    assert cur == 0                                 # the encoder starts in state 0 = lane 0 at time 0
    bits = np.array(walk[::-1], np.uint8)
    return best, bits


def main():
    from oracle_lib import Oracle
    o = Oracle()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(5)
    for trial in range(n):
        kind = trial % 4
        if kind == 0:
            q = rng.integers(0, 8, 2144)
        elif kind == 1:
            q = rng.integers(3, 5, 2144)           # many ties
        elif kind == 2:
            q = np.full(2144, 7 * (trial & 1))      # all ties / maximal metrics
        else:
            coded = o.encode_frame(rng.integers(0, 256, 134, dtype=np.uint8))
            perm = o.deinterleave_perm()
            q = np.clip(np.where(coded == 1, 7, 0) + rng.integers(-3, 4, 2144), 0, 7)
        m0, b0 = o.viterbi(q)
        m1, b1 = decode(q)
        assert m0 == m1 and np.array_equal(b0, b1), (trial, kind, m0, m1, int((b0 != b1).sum()))
        m2, b2 = decode_trace_forward(q)
        assert m0 == m2 and np.array_equal(b0, b2), ("trace-forward", trial, kind, m0, m2, int((b0 != b2).sum()))
    print(f"{n} trellises: metric and 1072 bits equal the oracle's, with decision bits and with trace-forward pointers")


if __name__ == "__main__":
    main()
