"""The BASELINE workload, generated straight into HBM with torch, together with its ground truth.

bench.py and tests/test_gpu_scale.py both build their inputs here, so what the bench times is what
the scale test checks.  SURVEY.md section 8d: genomes are uniform i.i.d. ACGT organised as species x
strains (strains = the species root with 1-3 % substitutions and sparse 1-10 bp indels), reads are
paired fragments (length ~ N(350, 30)), 1 % substitutions, at most one single-base indel per read,
2 % of the pairs from a genome that is not in the database.  Batch layout as in the reference:
R1 block then R2 block, mate of i is i + n (src/FASTQsequence.h:111-123).

The truth kept per pair is what the reference's own tests plant and expect to recover
(src/Tests.h:161-264: entry, offset, revComp of every read; :321-330: score = 2 x overlap length).
torch is plumbing here (device memory and a random generator), nothing of this is on the product path.
"""
import numpy as np
import torch

READ_LEN = 150
_ACGT = torch.tensor(list(b"ACGT"), dtype=torch.uint8)


def _codes_of(x):
    """ASCII uint8 tensor -> 0..3 for A,C,G,T (others 0)."""
    c = torch.zeros_like(x)
    c[x == ord("C")] = 1
    c[x == ord("G")] = 2
    c[x == ord("T")] = 3
    return c


def _revcomp_rows(x):
    """reverse-complement each row of an ASCII uint8 [n, L] tensor."""
    lut = torch.arange(256, dtype=torch.uint8, device=x.device)
    for a, b in zip(b"ACGT", b"TGCA"):
        lut[a] = b
    return lut[x.flip(1).long()]


def _revcomp_codes(c):
    """reverse complement of a code tensor (A0 C1 G2 T3 as in _ACGT: complement = 3 - code)"""
    return (3 - c).flip(0)


def viral_lengths(dev, gen, n_viral, min_len=5_000, max_len=200_000):
    """lengths of the "+viral" entries of BASELINE configs[2] ("add 10 k genomes of 5-200 kb"):
    log-uniform between the two bounds, as virus genome sizes are -- many small ones, few large ones"""
    u = torch.rand(n_viral, generator=gen, device=dev, dtype=torch.float64)
    ln = torch.exp(np.log(min_len) + u * (np.log(max_len) - np.log(min_len))).long().clamp(min_len, max_len)
    return ln.cpu().numpy().astype(np.int64)


RRNA_LEN, RRNA_COPIES, IS_LEN, IS_COPIES = 1500, 5, 1300, 2


def _sync(dev):
    """the generators' tensors are complete when they return (a CPU device has nothing to wait for)"""
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize(dev)


def make_database(dev, gen, n_species, n_strains, length, n_viral=0, repeats=False):
    """Species x strains database as ONE device byte tensor + host offsets.
    Strains derive from the species root by 1-3 % substitutions + sparse 1-10 bp indels.
    n_viral > 0 (configs[2]): that many unrelated i.i.d. genomes of 5-200 kb appended AFTER the bacterial entries
    (the random draws of the bacterial part are the same with and without them).
    repeats: what real bacterial sets carry and i.i.d. genomes do not -- sequence shared by hundreds of entries.  Every
    species root gets RRNA_COPIES copies of ONE 1.5 kb "rRNA operon" (the species' own variant: 3 % substitutions
    against the universal segment, so that, as with real 16S / 23S genes, conserved 32-mers are shared across species and
    others are not), and every third species IS_COPIES copies of one identical 1.3 kb insertion element; the strains then
    diverge from the root as usual.  A read from such a copy meets thousands of genome k-mers: the regime of the join's
    long pile-ups (nG x nR cross product, src/Overlap.h:163-197), of (read, entry) segments with several copies in the
    dedupe, and of read pairs with hundreds of alignment pairs.  Its own generator (seeded from `gen`'s seed + 7919), so
    that the rest of the database is bit for bit the one without repeats."""
    acgt = _ACGT.to(dev)
    rgen = None
    if repeats:
        rgen = torch.Generator(device=dev)
        rgen.manual_seed(gen.initial_seed() + 7919)
        rrna = torch.randint(0, 4, (RRNA_LEN,), generator=rgen, device=dev, dtype=torch.uint8)
        ins_el = torch.randint(0, 4, (IS_LEN,), generator=rgen, device=dev, dtype=torch.uint8)
    cap = int(n_species * n_strains * length * 1.01) + 1024 + n_viral * 200_000
    db = torch.empty(cap, dtype=torch.uint8, device=dev)
    offs = [0]
    for sp in range(n_species):
        root_codes = torch.randint(0, 4, (length,), generator=gen, device=dev, dtype=torch.uint8)
        if repeats and length > 40 * RRNA_LEN:
            mm = torch.rand(RRNA_LEN, generator=rgen, device=dev) < 0.03
            sh = torch.randint(1, 4, (RRNA_LEN,), generator=rgen, device=dev, dtype=torch.uint8)
            variant = torch.where(mm, (rrna + sh) % 4, rrna)
            root_codes = root_codes.clone()
            # copies at seeded, non-overlapping places: one per tenth of the genome
            slots = torch.randperm(10, generator=rgen, device=dev)[:RRNA_COPIES + IS_COPIES].tolist()
            jit = torch.rand(RRNA_COPIES + IS_COPIES, generator=rgen, device=dev).tolist()
            for k, (slot, u) in enumerate(zip(slots, jit)):
                seg = variant if k < RRNA_COPIES else ins_el
                if k >= RRNA_COPIES and sp % 3 != 0:
                    continue
                at = slot * (length // 10) + int(u * (length // 10 - seg.numel() - 1))
                if (k + sp) % 2:        # either strand
                    seg = _revcomp_codes(seg)
                root_codes[at:at + seg.numel()] = seg
        for st in range(n_strains):
            codes = root_codes
            if st > 0:
                rate = 0.01 + 0.02 * float(torch.rand(1, generator=gen, device=dev))
                m = torch.rand(length, generator=gen, device=dev) < rate
                shift = torch.randint(1, 4, (length,), generator=gen, device=dev, dtype=torch.uint8)
                codes = torch.where(m, (root_codes + shift) % 4, root_codes)
                # sparse indels: ~1 per 2 kb, 1-10 bp
                ev = torch.rand(length, generator=gen, device=dev) < 0.0005
                ln = torch.randint(1, 11, (length,), generator=gen, device=dev)
                is_ins = torch.rand(length, generator=gen, device=dev) < 0.5
                counts = torch.ones(length, dtype=torch.long, device=dev)
                counts = torch.where(ev & is_ins, 1 + ln, counts)
                # deletion of ln bases starting at the event
                del_start = torch.nonzero(ev & ~is_ins).flatten()
                if del_start.numel():
                    dl = ln[del_start]
                    idx = (del_start[:, None] + torch.arange(10, device=dev)[None, :])
                    keep = torch.arange(10, device=dev)[None, :] < dl[:, None]
                    idx = idx[keep]
                    idx = idx[idx < length]
                    counts[idx] = 0
                src = torch.repeat_interleave(torch.arange(length, device=dev), counts)
                out = codes[src]
                dup = torch.zeros_like(src, dtype=torch.bool)
                dup[1:] = src[1:] == src[:-1]
                rnd = torch.randint(0, 4, (src.numel(),), generator=gen, device=dev, dtype=torch.uint8)
                codes = torch.where(dup, rnd, out)
            n = codes.numel()
            db[offs[-1]:offs[-1] + n] = acgt[codes.long()]
            offs.append(offs[-1] + n)
    if n_viral:
        lens = viral_lengths(dev, gen, n_viral)
        total = int(lens.sum())
        db[offs[-1]:offs[-1] + total] = acgt[torch.randint(0, 4, (total,), generator=gen, device=dev)]
        base = offs[-1]
        offs.extend((base + np.cumsum(lens)).tolist())
    # the library reads this tensor on ITS OWN stream (and its own copy of the HIP runtime): nothing orders torch's kernels
    # before it, so the database must be complete when the caller gets it (a test that went straight on to
    # set_index_device built its index from a half-written database once in eight runs)
    _sync(dev)
    return db[:offs[-1]], np.array(offs, dtype=np.uint64)


def make_reads(dev, gen, db, offs, n_pairs, read_len=READ_LEN, sub_rate=0.01, indel_rate=0.001,
               unmapped=0.02, with_truth=False, by_length=False):
    """[2 * n_pairs, read_len] ASCII tensor in the reference batch layout (R1 block | R2 block).
    with_truth: also a dict of device tensors, one row per READ (2 * n_pairs):
      entry      source database entry, -1 for reads of the pairs that are not from the database
      rel        the reference's relativePosition of the read on that entry (start of the read, or
                 of its reverse complement, in entry coordinates)
      revcomp    0 / 1: the read is the reverse complement of the entry
      n_subs     substitutions applied to the read's bases
      has_indel  the read carries the one single-base indel
      seed_ok    an error-free 32-mer of the read sits on a 16-aligned entry offset (the genome
                 sampling of src/SLAM.h:64), so the reference's join must report (entry, rel, revcomp)
    The random draws are the same with and without truth.
    by_length: the source entry of a pair is drawn with probability proportional to its length (a community in which
    every genome has the same coverage) instead of uniformly over the entries -- for databases whose entries differ in
    length by orders of magnitude (configs[2]: 4 Mb bacterial next to 5 kb viral genomes)."""
    acgt = _ACGT.to(dev)
    goff = torch.from_numpy(offs.astype(np.int64)).to(dev)
    glen = goff[1:] - goff[:-1]
    ng = glen.numel()
    if by_length:
        at = (torch.rand(n_pairs, generator=gen, device=dev, dtype=torch.float64) * float(goff[-1])).long()
        g = (torch.searchsorted(goff, at, right=True) - 1).clamp(0, ng - 1)
    else:
        g = torch.randint(0, ng, (n_pairs,), generator=gen, device=dev)
    frag = (350 + 30 * torch.randn(n_pairs, generator=gen, device=dev)).round().long().clamp(read_len + 1, 1000)
    span = (glen[g] - frag - 2).clamp(min=1)
    start = (torch.rand(n_pairs, generator=gen, device=dev, dtype=torch.float64) * span).long()
    W = read_len + 1
    j = torch.arange(W, device=dev)[None, :]
    a_idx = goff[g][:, None] + start[:, None] + j                       # forward window at s
    b_idx = goff[g][:, None] + (start + frag - W)[:, None] + j          # window ending at s + f
    A = db[a_idx]
    B = _revcomp_rows(db[b_idx])
    flip = torch.rand(n_pairs, generator=gen, device=dev) < 0.5
    r1 = torch.where(flip[:, None], B, A)
    r2 = torch.where(flip[:, None], A, B)
    reads = torch.cat([r1, r2], 0)                                      # [2n, W]
    n2 = 2 * n_pairs
    # substitutions
    m = torch.rand(n2, W, generator=gen, device=dev) < sub_rate
    shift = torch.randint(1, 4, (n2, W), generator=gen, device=dev, dtype=torch.uint8)
    reads = torch.where(m, acgt[((_codes_of(reads) + shift) % 4).long()], reads)
    # at most one single-base indel per read
    p_ind = 1.0 - (1.0 - indel_rate) ** read_len
    has = torch.rand(n2, generator=gen, device=dev) < p_ind
    pos = torch.randint(1, read_len - 1, (n2,), generator=gen, device=dev)
    ins = torch.rand(n2, generator=gen, device=dev) < 0.5
    jj = torch.arange(read_len, device=dev)[None, :].expand(n2, read_len)
    d = torch.zeros(n2, read_len, dtype=torch.long, device=dev)
    d = torch.where((has & ins)[:, None] & (jj > pos[:, None]), torch.full_like(d, -1), d)
    d = torch.where((has & ~ins)[:, None] & (jj >= pos[:, None]), torch.full_like(d, 1), d)
    out = torch.gather(reads, 1, jj + d)
    rnd = acgt[torch.randint(0, 4, (n2,), generator=gen, device=dev)]
    at = (has & ins)[:, None] & (jj == pos[:, None])
    out = torch.where(at, rnd[:, None].expand(n2, read_len), out)
    # pairs from a genome that is not in the database
    um = torch.rand(n_pairs, generator=gen, device=dev) < unmapped
    um2 = torch.cat([um, um])
    junk = acgt[torch.randint(0, 4, (n2, read_len), generator=gen, device=dev)]
    out = torch.where(um2[:, None], junk, out)
    out = out.contiguous()
    if not with_truth:
        return out
    # ---- truth, per read ----
    L = read_len
    is_b = torch.cat([flip, ~flip])                    # the read is the B (reverse-complement) mate
    g2, start2, frag2 = torch.cat([g, g]), torch.cat([start, start]), torch.cat([frag, frag])
    rel = torch.where(is_b, start2 + frag2 - L, start2)
    sub = m[:, :L]
    n_subs = sub.sum(1)
    # substitutions in entry order: column j of an A read is entry position rel + j, of a B read
    # rel + L - 1 - j
    err = torch.where(is_b[:, None], sub.flip(1), sub).to(torch.int32)
    cs = torch.zeros(n2, L + 1, dtype=torch.int32, device=dev)
    cs[:, 1:] = torch.cumsum(err, 1)
    first = (rel + 15) // 16 * 16 - rel                # read-relative position of the first sampled entry offset
    seed_ok = torch.zeros(n2, dtype=torch.bool, device=dev)
    for t in range((L + 15) // 16):
        o = first + 16 * t
        valid = o + 32 <= L
        oc = o.clamp(max=L - 32)
        e = torch.gather(cs, 1, (oc + 32)[:, None]).squeeze(1) - torch.gather(cs, 1, oc[:, None]).squeeze(1)
        seed_ok |= valid & (e == 0)
    mapped = ~um2
    truth = {
        "entry": torch.where(mapped, g2, torch.full_like(g2, -1)),
        "rel": rel, "revcomp": is_b.to(torch.int64), "n_subs": n_subs,
        "has_indel": has, "seed_ok": seed_ok & mapped & ~has,
    }
    _sync(dev)   # (see make_database: the library reads `out` on its own stream)
    return out, truth


# ---- checks of a result set against the planted truth (nothing but the generator's own record) ----

def overlap_columns(ov_bytes):
    """uint8 device/host tensor holding n kslam_overlap records (48 B) -> dict of int64 columns."""
    w = ov_bytes.view(torch.int32).view(-1, 12)
    x = w[:, 3].long() & 0xFFFFFFFF
    return {
        "read": w[:, 0].long() & 0xFFFFFFFF, "entry": w[:, 1].long() & 0xFFFFFFFF, "rel": w[:, 2].long(),
        "revcomp": x & 0xFF, "score": x >> 16,
        "ref_begin": w[:, 4].long(), "ref_end": w[:, 5].long(),
        "query_begin": w[:, 6].long(), "query_end": w[:, 7].long(),
        "cigar_len": w[:, 8].long() & 0xFFFFFFFF,
        "cigar_off": (w[:, 10].long() & 0xFFFFFFFF) | (w[:, 11].long() << 32),
    }


def check_against_truth(ov_bytes, cig_words, truth, read_len, match=2, mismatch=3):
    """What src/Tests.h:161-264 and :321-330 expect, on every read of the batch:
      * sortedness: (read, entry, rel) non-decreasing (src/Overlap.h:87-98);
      * every read with an error-free sampled 32-mer and no indel has the overlap
        (entry, rel, revComp) it was planted with;
      * that overlap's score is at least the plain diagonal's (2 L - 5 per substitution) and at most
        2 L; for an error-free read it IS 2 L, the alignment spans the whole read and the CIGAR is <L>M;
      * reads of the pairs that are not from the database have no overlap at all.
    Returns a dict of counts; `ok` is True when nothing is missing or wrong."""
    c = overlap_columns(ov_bytes)
    n = c["read"].numel()
    L = read_len
    # one int64 key per row, (read, entry, rel) in that significance; the field widths follow the data
    # (20 M reads x 1 250 entries x 4 M positions of a 10 M-pair batch need 59 bits)
    n_entry = int(max(int(c["entry"].max()) if n else 0, int(truth["entry"].max()))) + 1
    n_rel = int(max(int(c["rel"].max()) if n else 0, int(truth["rel"].max()))) + 1026
    assert (int(truth["entry"].numel()) + 1) * n_entry * n_rel < 2 ** 63, "key does not fit int64"
    key = (c["read"] * n_entry + c["entry"]) * n_rel + (c["rel"] + 1024)
    unsorted = int((key[1:] < key[:-1]).sum()) if n > 1 else 0
    want = torch.nonzero(truth["seed_ok"]).flatten()
    wkey = (want * n_entry + truth["entry"][want]) * n_rel + (truth["rel"][want] + 1024)
    idx = torch.searchsorted(key, wkey).clamp(max=max(n - 1, 0))
    found = (key[idx] == wkey) if n else torch.zeros_like(wkey, dtype=torch.bool)
    # the reference leaves equal (read, entry, rel) with different revComp to an unstable sort; here
    # revComp = false comes first, so the planted strand may be the second of two equal keys
    rc_want = truth["revcomp"][want]
    idx2 = (idx + 1).clamp(max=max(n - 1, 0))
    second = found & (c["revcomp"][idx] != rc_want) & (key[idx2] == wkey) & (c["revcomp"][idx2] == rc_want)
    idx = torch.where(second, idx2, idx)
    strand_ok = found & (c["revcomp"][idx] == rc_want)
    ns = truth["n_subs"][want]
    sc = c["score"][idx]
    floor = match * L - (match + mismatch) * ns
    score_ok = strand_ok & (sc >= floor) & (sc <= match * L)
    perfect = strand_ok & (ns == 0)
    p_idx = idx[perfect]
    rel_p = truth["rel"][want][perfect]
    perfect_ok = (c["score"][p_idx] == match * L) & (c["ref_begin"][p_idx] == rel_p) & \
                 (c["ref_end"][p_idx] == rel_p + L - 1) & (c["query_begin"][p_idx] == 0) & \
                 (c["query_end"][p_idx] == L - 1)
    if cig_words is not None and cig_words.numel():
        one = c["cigar_len"][p_idx] == 1
        op = cig_words.view(torch.int32)[c["cigar_off"][p_idx].clamp(max=cig_words.numel() - 1)].long()
        perfect_ok &= one & (op == (L << 4))
    unmapped = truth["entry"] < 0
    hits_unmapped = int(unmapped[c["read"]].sum()) if n else 0
    res = {
        "overlaps": n, "unsorted_neighbours": unsorted,
        "planted_expected": int(want.numel()), "planted_found": int(strand_ok.sum()),
        "planted_missing": int(want.numel() - int(strand_ok.sum())),
        "planted_score_out_of_bounds": int((strand_ok & ~score_ok).sum()),
        "error_free_reads": int(perfect.sum()), "error_free_wrong": int((~perfect_ok).sum()),
        "overlaps_on_reads_not_from_db": hits_unmapped,
    }
    res["ok"] = (unsorted == 0 and res["planted_missing"] == 0 and res["planted_score_out_of_bounds"] == 0
                 and res["error_free_wrong"] == 0 and hits_unmapped == 0)
    return res


# ---- the batch as FASTQ text (what the reference's driver reads, src/FASTQsequence.h:129-165) ----

ID_DIGITS = 8


def fastq_text(reads_block, mate, first_pair=0, gen=None, phred=(20, 40)):
    """[n, L] ASCII base rows (device or host tensor) -> uint8 tensor [n, W] on the same device holding the records
    "@p00000123/1\n<bases>\n+\n<quality>\n" (fixed width; the identifier the reference derives is p00000123,
    src/FASTQsequence.h:61-71).  Qualities: uniform Phred 20-40 from `gen`, constant 'I'
    (Phred 40) without a generator.  Returns (text, quality rows [n, L])."""
    n, L = reads_block.shape
    dev = reads_block.device
    Wd = 2 + ID_DIGITS + 3 + L + 3 + L + 1
    a = torch.empty((n, Wd), dtype=torch.uint8, device=dev)
    a[:, 0] = ord("@")
    a[:, 1] = ord("p")
    idx = torch.arange(first_pair, first_pair + n, device=dev, dtype=torch.int64)
    for d in range(ID_DIGITS):
        a[:, 2 + d] = ((idx // 10 ** (ID_DIGITS - 1 - d)) % 10 + ord("0")).to(torch.uint8)
    c = 2 + ID_DIGITS
    a[:, c] = ord("/")
    a[:, c + 1] = ord("0") + mate
    a[:, c + 2] = 10
    a[:, c + 3:c + 3 + L] = reads_block
    a[:, c + 3 + L] = 10
    a[:, c + 4 + L] = ord("+")
    a[:, c + 5 + L] = 10
    if gen is None:
        q = torch.full((n, L), ord("I"), dtype=torch.uint8, device=dev)
    else:
        q = torch.randint(33 + phred[0], 33 + phred[1] + 1, (n, L), generator=gen, device=dev, dtype=torch.uint8)
    a[:, c + 6 + L:c + 6 + 2 * L] = q
    a[:, Wd - 1] = 10
    return a, q


def pair_id(i):
    return b"p%0*d" % (ID_DIGITS, i)


def make_batch_in_pieces(dev, gen, db, offs, total_pairs, read_len, pieces=8, first_piece=0, n_pieces=None, seed_base=2,
                         by_length=False, with_truth=True):
    """A large batch generated piece by piece (the temporaries of make_reads are 8-byte-per-base tensors): pieces
    [first_piece, first_piece + n_pieces) of `pieces`, each seeded on its own so that the batch is the same however it
    is split (bench.py --strong gives every rank its pieces).  -> ([2 m, L] reads in block layout, truth or None)"""
    n_pieces = pieces if n_pieces is None else n_pieces
    piece = total_pairs // pieces
    r1s, r2s, tr = [], [], []
    for pc in range(first_piece, first_piece + n_pieces):
        gen.manual_seed(seed_base + 1000 * pc)
        r = make_reads(dev, gen, db, offs, piece, read_len=read_len, with_truth=with_truth, by_length=by_length)
        if with_truth:
            r, t = r
            tr.append(t)
        r1s.append(r[:piece])
        r2s.append(r[piece:])
    reads = torch.cat(r1s + r2s, 0).contiguous()
    truth = {k: torch.cat([t[k][:piece] for t in tr] + [t[k][piece:] for t in tr]) for k in tr[0]} if with_truth else None
    _sync(dev)   # (see make_database)
    return reads, truth


def taxonomy(n_species, n_strains, n_viral=0):
    """A synthetic <db>/taxDB for the synthetic database (four lines per node: id, parent id, name, rank;
    src/TaxonomyDatabase.h:153-183) and the taxonomy id of every entry.  root 1 -> Bacteria 2 -> genus (two species
    each) -> species -> strain = entry; Viruses 3 -> one species per viral entry."""
    recs = [(1, 1, b"root", b"no rank"), (2, 1, b"Bacteria", b"superkingdom")]
    ids = []
    for s in range(n_species):
        g = 1000 + s // 2
        if s % 2 == 0:
            recs.append((g, 2, b"Genus%d" % g, b"genus"))
        recs.append((10000 + s, g, b"Genus%d species%d" % (g, s), b"species"))
        for k in range(n_strains):
            recs.append((100000 + s * n_strains + k, 10000 + s, b"strain %d.%d" % (s, k), b"strain"))
            ids.append(100000 + s * n_strains + k)
    if n_viral:
        recs.append((3, 1, b"Viruses", b"superkingdom"))
        for v in range(n_viral):
            recs.append((2000000 + v, 3, b"Synthetic virus %d" % v, b"species"))
            ids.append(2000000 + v)
    return b"".join(b"%d\n%d\n%s\n%s\n" % r for r in recs), np.array(ids, dtype=np.uint32)


def ids_view(T, n_pairs, read_len, first_pair=0):
    """kslam_reads_view of a fixed-length batch [R1 block | R2 block] without bases / quality columns (the SAM writer
    gets NM / MD / log-probability from the GPU and never reads them): identifiers p<8 digits>, offsets."""
    ids = np.empty((2 * n_pairs, 1 + ID_DIGITS), dtype=np.uint8)
    ids[:, 0] = ord("p")
    idx = np.tile(np.arange(first_pair, first_pair + n_pairs, dtype=np.int64), 2)
    for d in range(ID_DIGITS):
        ids[:, 1 + d] = (idx // 10 ** (ID_DIGITS - 1 - d)) % 10 + ord("0")
    ids = ids.reshape(-1)
    ioff = np.arange(2 * n_pairs + 1, dtype=np.uint64) * np.uint64(1 + ID_DIGITS)
    boff = np.arange(2 * n_pairs + 1, dtype=np.uint64) * np.uint64(read_len)
    rv = T.ReadsView(2 * n_pairs, None, boff.ctypes.data, None, boff.ctypes.data, ids.ctypes.data, ioff.ctypes.data)
    return type("IdsView", (), {"view": rv, "n_reads": 2 * n_pairs, "_keep": (ids, ioff, boff)})()
