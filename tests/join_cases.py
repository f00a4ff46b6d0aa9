"""Seeded inputs for the join / dedupe / Aligner::Align / SW-driver pins (test data plumbing, numpy only).

Used by tests/test_oracle.py (oracle vs the real reference in oracle/_ref), by tests/golden/make_golden.py (which records
the real reference's answers as join_vectors.npz / align_vectors.npz) and by the -m gpu tests that replay those fixtures
through the C ABI.  What the cases cover:
  * pile-ups with several leading genome records: a segment shared by two entries, a reverse-complement copy in a third,
    a tandem repeat whose period is the genome gap (many genome records of ONE entry under one k-mer)
  * forward / reverse-complement mixes on both sides
  * k-mer 0 runs (poly-A / poly-T, all-N), which the join skips (src/Overlap.h:236-239)
  * reads hanging off either genome end (negative rel, truncated window), reads shorter than 32, exactly 32
  * lower-case, N, U and IUPAC characters in reads (FASTQ bases are used verbatim)
  * a reverse-palindromic 150-base window: the same (read, entry, rel) arrives with revComp 0 AND 1 -- the tie
    overlapSort does not order (src/Overlap.h:87-98); flagged, compared modulo revComp
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTacgt", b"TGCAtgca"):
    _COMP[_a] = _b


def _rb(rng, n):
    return _ACGT[rng.integers(0, 4, n)]


def _rc(a):
    return _COMP[a[::-1]]


def make_join_case(seed, n_reads=260, read_len=150):
    """-> (reads: list[bytes], genomes: list[bytes])"""
    rng = np.random.default_rng(seed)
    g0 = _rb(rng, 5000)
    g1 = _rb(rng, 4200)
    g2 = _rb(rng, 4800)
    g2[1600:3600] = g0[480:2480]                 # shared segment, both offsets multiples of 16 -> same sampled k-mers
    g3 = _rb(rng, 3900)
    g3[800:1800] = _rc(g1[96:1096])              # reverse-complement copy
    unit = _rb(rng, 16)
    g4 = np.concatenate([_rb(rng, 640), np.tile(unit, 40), _rb(rng, 700)])   # tandem repeat, period = genome gap
    x = _rb(rng, read_len // 2)
    pal = np.concatenate([x, _rc(x)])            # reverse-palindromic window
    g5 = np.concatenate([_rb(rng, 1600), pal, _rb(rng, 1200 + (read_len & 1))])
    g6 = np.concatenate([_rb(rng, 300), np.full(200, ord("A"), np.uint8), _rb(rng, 300)])  # poly-A: k-mer 0
    for p in rng.integers(0, len(g1), 6):          # N and IUPAC letters inside a genome (2-bit code 0 in the k-mer, 4 in SW)
        g1[p] = rng.choice(np.frombuffer(b"NNRYK", dtype=np.uint8))
    genomes = [g0, g1, g2, g3, g4, g5, g6]
    reads = []
    for i in range(n_reads):
        gi = int(rng.integers(0, len(genomes)))
        g = genomes[gi]
        kind = rng.random()
        if kind < 0.08:                          # hanging off the left end
            start = -int(rng.integers(1, read_len - 40))
        elif kind < 0.16:                        # hanging off the right end
            start = len(g) - int(rng.integers(40, read_len))
        else:
            start = int(rng.integers(0, len(g) - read_len))
        lo, hi = max(start, 0), min(start + read_len, len(g))
        r = np.concatenate([_rb(rng, lo - start), g[lo:hi], _rb(rng, start + read_len - hi)])
        if rng.random() < 0.5:
            r = _rc(r)
        m = rng.random(len(r)) < 0.012
        r[m] = _rb(rng, int(m.sum()))
        if rng.random() < 0.25:                  # one short indel
            p = int(rng.integers(5, len(r) - 5))
            n = int(rng.integers(1, 4))
            r = np.concatenate([r[:p], r[p + n:]]) if rng.random() < 0.5 else np.concatenate([r[:p], _rb(rng, n), r[p:]])
        if rng.random() < 0.1:
            r[int(rng.integers(0, len(r)))] = ord("N")
        if rng.random() < 0.08:                  # verbatim FASTQ: lower case and odd letters
            p = int(rng.integers(0, len(r) - 12))
            r[p:p + 6] = np.frombuffer(bytes(r[p:p + 6]).lower(), dtype=np.uint8)
            r[p + 8] = rng.choice(np.frombuffer(b"URYKMSWn", dtype=np.uint8))
        reads.append(r.tobytes())
    reads += [pal.tobytes(), _rc(pal).tobytes(),                          # the revComp tie
              np.tile(unit, 10)[:read_len].tobytes(),                    # inside the tandem repeat
              b"A" * read_len, b"T" * read_len, b"N" * read_len,         # k-mer 0 only
              g0[100:131].tobytes(), g0[200:232].tobytes(), b"",         # 31, 32 and 0 bases
              g0[4900:].tobytes() + _rb(rng, 50).tobytes(),              # 100 genome bases then junk
              g1[:64].tobytes()]
    return reads, [g.tobytes() for g in genomes]


def revcomp_tie_rows(raw):
    """Indices i of the DEDUPED list's candidates are ambiguous when the RAW list holds the same (read, entry, rel) with
    both revComp values.  -> set of (read, entry, rel)"""
    seen = {}
    for r, e, l, c in zip(raw["read"], raw["entry"], raw["rel"], raw["revcomp"]):
        seen.setdefault((int(r), int(e), int(l)), set()).add(int(c))
    return {k for k, v in seen.items() if len(v) == 2}


def make_align_cases(seed, n=400):
    """(query, ref, ref_len) ASCII triples for Aligner::Align: substitutions, indels, junk ends, lower case / U / IUPAC /
    N on both sides, ref_len shorter than the string.  -> list[(bytes, bytes, int)]"""
    rng = np.random.default_rng(seed)
    odd = np.frombuffer(b"acgtuUNnRYKMSWBDHVXrykm-*", dtype=np.uint8)
    out = []
    for i in range(n):
        L = int(rng.integers(33, 260))
        ref = _rb(rng, L)
        q = ref.copy()
        k = int(rng.integers(0, max(1, L // 9)))
        q[rng.integers(0, L, k)] = _rb(rng, k)
        for _ in range(int(rng.integers(0, 3))):
            p = int(rng.integers(1, len(q) - 1))
            m = int(rng.integers(1, 5))
            q = np.concatenate([q[:p], q[p + m:]]) if rng.random() < 0.5 else np.concatenate([q[:p], _rb(rng, m), q[p:]])
        if rng.random() < 0.3:
            q = np.concatenate([_rb(rng, int(rng.integers(1, 25))), q])
        if rng.random() < 0.45:
            q = np.concatenate([q, _rb(rng, int(rng.integers(1, 25)))])
        for s in (q, ref):
            if rng.random() < 0.35:
                m = int(rng.integers(1, 6))
                s[rng.integers(0, len(s), m)] = odd[rng.integers(0, len(odd), m)]
        ref_len = len(ref) if rng.random() < 0.7 else int(rng.integers(max(1, len(ref) // 2), len(ref) + 1))
        out.append((q.tobytes(), ref.tobytes(), ref_len))
    return out
