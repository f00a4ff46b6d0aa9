#!/usr/bin/env python3
"""Generates tests/golden/sw_tie_cases.json: read / genome pairs whose best local-alignment score is
reached by two alignments ending a few diagonals apart, such that a banded anti-diagonal sweep
that keeps "the first best cell seen" per lane reports a different end than the reference's rule
(highest score, then smallest end column, then smallest end row; src/ssw.c:316-342).

Method: a plain-numpy Smith-Waterman (affine gaps, the reference's scoring 2/3/5/2) gives the
full H matrix; `lane_rule` replays the visiting order of k_sw_band (lane t owns DPL adjacent
diagonals, even diagonals on anti-diagonal k, odd ones on k + 1, one turn = two anti-diagonals) and
compares its pick with the true rule.  Candidates are a unique 48-base prefix (so that the pair
shares 32-mer seeds) + a tandem repeat + junk + repeat, the genome ending inside the repeat.
The expected alignments in the test come from the oracle, not from this model.

    python tests/golden/make_sw_tie_cases.py [seeds...]      (a few minutes per seed)
"""
import json
import os
import sys

import numpy as np

M, X, GO, GE = 2, 3, 5, 2


def sw(read, ref):
    L, W = len(read), len(ref)
    S = np.where((read[:, None] == 4) | (ref[None, :] == 4), 0, np.where(read[:, None] == ref[None, :], M, -X))
    H = np.zeros((L + 1, W + 1), int)
    F = np.zeros(W + 1, int)
    jj = np.arange(W + 1)
    for i in range(1, L + 1):
        F = np.maximum(0, np.maximum(F - GE, H[i - 1] - GO))
        Ht = np.zeros(W + 1, int)
        Ht[1:] = np.maximum(0, np.maximum(H[i - 1, :-1] + S[i - 1], F[1:]))
        acc = np.maximum.accumulate(Ht - GO + jj * GE)          # E[j] = max(0, max_{k<j}(Ht[k] - GO - (j-1-k) GE))
        E = np.zeros(W + 1, int)
        E[1:] = np.maximum(0, acc[:-1] - (jj[1:] - 1) * GE)
        H[i] = np.maximum(Ht, E)
        H[i, 0] = 0
    return H[1:, 1:]


def lane_rule(H, DPL, GL, d0=0):
    ND = DPL * GL
    dlo = d0 - ND // 2
    dhi = dlo + ND - 1
    kmin = 0 if dhi >= 0 else -dhi
    k0 = kmin - ((kmin - dlo) & 1)
    cells = np.argwhere(H == H.max())
    winners = []
    for t in range(GL):
        db = dlo + DPL * t
        cand = []
        for i, j in cells:
            q = (j - i) - db
            if 0 <= q < DPL:
                n = (i + j - k0 - (q & 1)) // 2
                cand.append(((n, q & 1, q), (int(j), int(i))))          # visiting order -> (column, row)
        if cand:
            winners.append(min(cand)[1])
    inband = [(int(j), int(i)) for i, j in cells if dlo <= j - i <= dhi]
    return (min(winners) if winners else None), (min(inband) if inband else None)


def search(seed, trials, prefix=48):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(trials):
        U = rng.integers(0, 4, prefix)
        per = int(rng.integers(2, 6))
        unit = rng.integers(0, 4, per)
        na, nb = int(rng.integers(2, 12)), int(rng.integers(2, 12))
        junk = rng.integers(0, 5, int(rng.integers(1, 8)))
        read = np.concatenate([U, np.resize(unit, na), junk, np.roll(unit, -na % per)[np.arange(nb) % per]])
        ref = np.concatenate([U, np.resize(unit, int(rng.integers(na + nb - 4, na + nb + 12)))])
        for _ in range(int(rng.integers(0, 3))):
            read[prefix + rng.integers(0, len(read) - prefix)] = rng.integers(0, 5)
        H = sw(read, ref)
        for DPL, GL in ((4, 8), (6, 8), (8, 8)):
            a, b = lane_rule(H, DPL, GL)
            if a != b:
                out.append({"dpl": DPL, "lane": a, "true": b, "read": "".join("ACGTN"[c] for c in read),
                            "ref": "".join("ACGTN"[c] for c in ref)})
                break
    return out


if __name__ == "__main__":
    seeds = [int(x) for x in sys.argv[1:]] or [11, 12, 13, 14, 15, 16, 17]
    cases = []
    for s in seeds:
        cases += search(s, 5000)
        print("seed", s, "->", len(cases), "cases so far", flush=True)
    json.dump(cases, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sw_tie_cases.json"), "w"))
