"""Seeded synthetic genomes and paired reads, numpy only.

Host-side test/bench data plumbing: no network, so there is no RefSeq; genomes
are uniform i.i.d. ACGT organised as species x strains, reads are paired
fragments with substitutions and sparse indels.  Layout of a read batch follows
the reference: R1 block then R2 block, mate of i is i + n
(reference src/FASTQsequence.h:111-123).
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = np.arange(256, dtype=np.uint8)
for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
    _COMP[a] = b


def revcomp(a):
    """Reverse complement of a uint8 ASCII array."""
    return _COMP[a[::-1]]


def random_bases(rng, n):
    return _ACGT[rng.integers(0, 4, n, dtype=np.uint8)]


def mutate(rng, seq, sub_rate, indel_rate, max_indel=3):
    """Substitutions at sub_rate per base, indels (1..max_indel) at indel_rate per base."""
    s = seq.copy()
    n = len(s)
    if sub_rate > 0 and n:
        m = rng.random(n) < sub_rate
        # substitute with a DIFFERENT base
        cur = np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), s[m])
        cur = np.where((cur < 4), cur, 0)
        s[m] = _ACGT[(cur + rng.integers(1, 4, m.sum())) % 4]
    if indel_rate > 0 and n:
        k = rng.binomial(n, indel_rate)
        if k:
            pos = np.sort(rng.integers(0, n, k))
            out, last = [], 0
            for p in pos:
                if p < last:
                    continue
                out.append(s[last:p])
                ln = int(rng.integers(1, max_indel + 1))
                if rng.random() < 0.5:
                    out.append(random_bases(rng, ln))  # insertion
                    last = p
                else:
                    last = min(n, p + ln)              # deletion
            out.append(s[last:])
            s = np.concatenate(out)
    return s


def make_genomes(seed, n_species, n_strains, length, strain_sub=0.02, strain_indel=0.0005,
                 shared_segment=0):
    """n_species * n_strains genomes; strains derive from the species root."""
    rng = np.random.default_rng(seed)
    genomes = []
    for _ in range(n_species):
        root = random_bases(rng, length)
        for st in range(n_strains):
            if st == 0:
                genomes.append(root.copy())
            else:
                genomes.append(mutate(rng, root, strain_sub * rng.uniform(0.5, 1.5), strain_indel, 10))
    if shared_segment and len(genomes) >= 2:
        seg = genomes[0][1000:1000 + shared_segment]
        g = genomes[-1]
        p = len(g) // 2
        genomes[-1] = np.concatenate([g[:p], seg, g[p + len(seg):]])[:len(g)]
    return genomes


def make_paired_reads(seed, genomes, n_pairs, read_len=150, frag_mean=350, frag_sd=30,
                      sub_rate=0.01, indel_rate=0.001, unmapped_frac=0.02, n_rate=0.0,
                      edge_frac=0.0):
    """Returns (reads, truth): reads = [R1_0..R1_{n-1}, R2_0..R2_{n-1}] (uint8 arrays);
    truth[i] = (genome index or -1, fragment start, flipped)."""
    rng = np.random.default_rng(seed)
    r1, r2, truth = [], [], []
    glens = np.array([len(g) for g in genomes])
    for _ in range(n_pairs):
        frag_len = int(np.clip(rng.normal(frag_mean, frag_sd), read_len, 1000))
        if rng.random() < unmapped_frac:
            frag = random_bases(rng, frag_len)
            gi, start = -1, 0
        else:
            gi = int(rng.integers(0, len(genomes)))
            G = int(glens[gi])
            if edge_frac and rng.random() < edge_frac:
                # fragment hanging off either end of the genome (window truncation / negative rel)
                start = int(rng.integers(-frag_len + 40, 0)) if rng.random() < 0.5 \
                    else int(rng.integers(G - frag_len, G - 40))
            else:
                start = int(rng.integers(0, max(1, G - frag_len)))
            lo, hi = max(start, 0), min(start + frag_len, G)
            frag = np.concatenate([random_bases(rng, lo - start), genomes[gi][lo:hi],
                                   random_bases(rng, start + frag_len - hi)])
        flip = bool(rng.random() < 0.5)
        if flip:
            frag = revcomp(frag)
        a = mutate(rng, frag[:read_len], sub_rate, indel_rate)[:read_len]
        b = mutate(rng, revcomp(frag)[:read_len], sub_rate, indel_rate)[:read_len]
        if n_rate:
            for x in (a, b):
                m = rng.random(len(x)) < n_rate
                x[m] = ord("N")
        r1.append(a)
        r2.append(b)
        truth.append((gi, start, flip))
    return r1 + r2, truth


def to_bytes(seqs):
    return [s.tobytes() for s in seqs]
