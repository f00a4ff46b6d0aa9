"""Prototype of the "coarse-DP" band certificate that round 2's DESIGN (section 12) and its review proposed for the
250-bp Smith-Waterman tiers -- and the measurement that it certifies NOTHING the existing certificate does not.

The idea: the band certificate of csrc/sw.hip must exclude every alignment scoring >= S1 (the seed diagonal's plain
score) outside the swept band; with m mismatches it needs 2.5 m diagonals either side although far diagonals match at
25 %.  A dynamic programme over (block of 32 rows, diagonal) with exact block scores for "stay" transitions and
optimistic rewards for blocks in which the path changes diagonal would bound every band-leaving path from above.
Result (python tools/coarse_certificate_prototype.py 250 800; 2 358 candidates): the tier chosen is the old one for EVERY
candidate.  Why: a block in which the path moves must be credited with up to `match x rows` (the path may take the matching
rows of either diagonal), so zig-zagging between two in-band neighbours gains ~3 points per block containing a mismatch
over the exact stay score; that inflation of the IN-band prefix bound alone exceeds the gap cost of leaving the band.
Tightening it needs per-row information, i.e. the banded DP itself.  Kept as evidence; nothing in the product uses it.
Needs the oracle (checker) for the candidate list: a tools/ script, not a test.
"""
import sys, importlib, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
g.load_package()
W = importlib.import_module('kslam_amd.workload')
import oracle as O

L = int(sys.argv[1]) if len(sys.argv) > 1 else 250
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
a, b, gO, gE = 2, 3, 5, 2
dev = torch.device('cpu'); gen = torch.Generator(device=dev); gen.manual_seed(1)
db, offs = W.make_database(dev, gen, 2, 5, 300000)
gen.manual_seed(2)
reads = W.make_reads(dev, gen, db, offs, NP, read_len=L).numpy()
rl = [reads[i].tobytes() for i in range(reads.shape[0])]
dbn = db.numpy()
gl = [dbn[int(offs[i]):int(offs[i+1])].tobytes() for i in range(len(offs)-1)]
O.set_num_threads(8)
al, cg, ph = O.align_to_database(rl, gl)
print('candidates', len(al))
code = np.full(256, 4, dtype=np.int8)
for k, c in enumerate(b'ACGT'): code[c] = k
comp = np.array([3, 2, 1, 0, 4], dtype=np.int8)

def amin_of(score, Lq, Wq):
    if score <= 0: return -1
    Lm = min(Lq, Wq); m00 = (score + a - 1) // a
    if m00 > Lm: return 10**9
    am = m00
    room = Lm * a - score - gO
    if room >= 0:
        gg = 1
        if gE < a: gg = min(room // gE + 1, 2047)
        m0 = (score + gO + (gg - 1) * gE + a - 1) // a
        am = min(am, m0 - gg)
    return am

TIERS = [16, 32, 48, 64, 96, 128]
def tier_old(best, Lq, Wq, d0):
    am = amin_of(best, Lq, Wq)
    for k, ND in enumerate(TIERS):
        dlo = d0 - ND // 2
        if am >= 0 and (am == 10**9 or (am - Lq >= dlo and Wq - am <= dlo + ND - 1)): return k
    return None

T = 32
def coarse_tier(q, w, S1, d0, Rlo, Rhi):
    """smallest tier whose band the coarse DP certifies; None if none"""
    Lq, Wq = len(q), len(w)
    D = Rhi - Rlo + 1
    nb = (Lq + T - 1) // T
    M = np.zeros((D, nb), dtype=np.int32); X = np.zeros((D, nb), dtype=np.int32)
    for x in range(D):
        d = Rlo + x
        i0, i1 = max(0, -d), min(Lq, Wq - d)
        if i1 <= i0: continue
        qi = q[i0:i1]; wj = w[i0 + d:i1 + d]
        valid = (qi < 4) & (wj < 4)
        eq = valid & (qi == wj); ne = valid & (qi != wj)
        blk = np.arange(i0, i1) // T
        M[x] = np.bincount(blk, weights=eq, minlength=nb)[:nb]
        X[x] = np.bincount(blk, weights=ne, minlength=nb)[:nb]
    s = a * M - b * X; p = a * M
    Tb = np.minimum(T, Lq - T * np.arange(nb))
    NEG = -10**6
    def dt(f):   # max-plus with cost gO + (|delta|-1) gE for delta != 0
        out = np.full(D, NEG)
        r = NEG
        for x in range(D):
            out[x] = max(out[x], r)
            r = max(r - gE, f[x] - gO)
        r = NEG
        for x in range(D - 1, -1, -1):
            out[x] = max(out[x], r)
            r = max(r - gE, f[x] - gO)
        return out
    for k, ND in enumerate(TIERS):
        dlo = d0 - ND // 2; dhi = dlo + ND - 1
        inB = (np.arange(Rlo, Rhi + 1) >= dlo) & (np.arange(Rlo, Rhi + 1) <= dhi)
        if inB.all(): return k
        f0 = np.zeros(D); f1 = np.full(D, NEG)      # at a boundary: layer 0 (never outside), layer 1
        f0[~inB] = NEG                              # (a zero-score start on an outside diagonal is layer 1)
        f1[~inB] = 0
        U = NEG
        for be in range(nb):
            rew = a * Tb[be] + (gO - gE)
            # ending inside this block (partial traverse) from previous states, and starting inside
            e1 = np.maximum(f1 + p[:, be], np.where(~inB, f0 + p[:, be], NEG))
            U = max(U, e1.max())
            # stay
            n0 = np.where(inB, f0 + s[:, be], NEG)
            n1 = np.maximum(f1 + s[:, be], np.where(~inB, f0 + s[:, be], NEG))
            # moves
            m0 = dt(f0) + rew                      # one move from layer 0
            m1 = dt(f1) + rew
            out0 = np.where(~inB, m0, NEG)         # landed outside: layer 1
            m01 = dt(out0 - rew) + rew             # ... and a second move anywhere
            n0 = np.maximum(n0, np.where(inB, m0, NEG))
            n1 = np.maximum.reduce([n1, m1, out0, m01])
            # starts inside the block
            n0 = np.maximum(n0, np.where(inB, p[:, be], NEG))
            n1 = np.maximum(n1, np.where(~inB, p[:, be], NEG))
            # a start inside the block followed by moves: reward minus costs from any diagonal
            st = np.zeros(D)
            ms = dt(st) + rew
            n0 = np.maximum(n0, np.where(inB, ms, NEG))   # (start in B, move inside B)
            so = np.where(~inB, 0, NEG)
            n1 = np.maximum(n1, np.maximum(dt(so) + rew, np.where(~inB, ms, NEG)))
            f0, f1 = np.maximum(n0, 0 * n0 + NEG), n1
            U = max(U, f1.max())
        if U < S1: return k
    return None

old = np.zeros(8, dtype=int); new = np.zeros(8, dtype=int); n_done = 0
t0 = time.time()
sel = np.random.default_rng(1).permutation(len(al))[:2500]
move = {}
for ci in sel:
    o = al[ci]
    rd = np.frombuffer(rl[o['read']], dtype=np.uint8)
    q = code[rd]
    ge = np.frombuffer(gl[o['entry']], dtype=np.uint8)
    rel = int(o['rel']); s0 = max(rel, 0)
    win = ge[s0:s0 + len(q)]
    w = code[win]
    if o['revcomp']: w = comp[w][::-1]
    Lq, Wq = len(q), len(w)
    d0 = rel if rel < 0 else 0
    # best of the five plain diagonals d0-2..d0+2
    best = 0
    for d in range(d0 - 2, d0 + 3):
        i0, i1 = max(0, -d), min(Lq, Wq - d)
        if i1 <= i0: continue
        qi = q[i0:i1]; wj = w[i0 + d:i1 + d]
        valid = (qi < 4) & (wj < 4)
        sc = a * int((valid & (qi == wj)).sum()) - b * int((valid & (qi != wj)).sum())
        best = max(best, sc)
    ko = tier_old(best, Lq, Wq, d0)
    am = amin_of(best, Lq, Wq)
    if am < 0 or am == 10**9: continue
    Rlo, Rhi = am - Lq, Wq - am
    if Rhi - Rlo > 400: ko = None
    kn = coarse_tier(q, w, best, d0, max(Rlo, -Lq + 1), min(Rhi, Wq - 1)) if ko is not None else None
    old[ko if ko is not None else 7] += 1; new[kn if kn is not None else 7] += 1
    move[(ko, kn)] = move.get((ko, kn), 0) + 1
    n_done += 1
print('L', L, 'n', n_done, 'seconds', round(time.time() - t0, 1))
print('tiers      ', TIERS, 'unknown')
print('old planned', old)
print('new planned', new)
print(sorted(move.items(), key=lambda kv: -kv[1])[:20])
