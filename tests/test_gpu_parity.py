"""GPU parity tests proper: the HIP path through the C ABI vs the CPU oracle.

Bit-exact bar: every integer field and every CIGAR op must be identical.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(kslam):
    c = kslam.Context()
    yield c
    c.close()


def _rand_seqs(rng, n, lo, hi, junk=True):
    out = []
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L)].copy()
        if junk and L and rng.random() < 0.3:
            k = rng.integers(0, L, rng.integers(1, 4))
            s[k] = rng.choice(np.frombuffer(b"NnacgtRYU-", dtype=np.uint8), len(k))
        out.append(s.tobytes())
    return out


@pytest.mark.parametrize("is_gb,gap", [(False, 1), (True, 16), (True, 4), (False, 3), (True, 1), (False, 64)])
def test_extract_parity(ctx, oracle, is_gb, gap):
    rng = np.random.default_rng(100 + gap + int(is_gb))
    seqs = _rand_seqs(rng, 300, 0, 400) + [b"", b"A" * 31, b"A" * 32, b"ACGT" * 8, b"N" * 50,
                                           b"acgt" * 20] + _rand_seqs(rng, 3, 3000, 9000)
    exp = oracle.extract_kmers(seqs, is_gb, gap)
    got = ctx.extract_kmers(seqs, is_gb, gap)
    assert len(exp) == len(got)
    assert (exp == got).all()


def test_extract_golden_vector(ctx):
    # reference answers recorded in SURVEY.md section 8c
    s = b"ACGTTGCAAGGCTTAACCGGTTACGATCGATCGGATCCAGTNACGT"
    r = ctx.extract_kmers([b""] * 7 + [s], False, 1)
    assert (int(r[0]["kmer"]), int(r[0]["meta"]), int(r[0]["offset"])) == (0x1eb43da05fa1c9c9, 7, 0)
    assert (int(r[1]["kmer"]), int(r[1]["meta"]), int(r[1]["offset"])) == (0x72727817e835ad07, 0x40000007, 13)
    g = ctx.extract_kmers([b""] * 5 + [s], True, 4)
    assert [(int(x["meta"]), int(x["offset"])) for x in g] == [
        (0x80000005, 0), (0xc0000005, 4), (0x80000005, 8), (0x80000005, 12)]


@pytest.mark.parametrize("n", [0, 1, 63, 4096, 4097, 100000, 1 << 20])
def test_sort_parity(ctx, oracle, kslam, n):
    rng = np.random.default_rng(7 + n)
    recs = np.zeros(n, dtype=kslam.KMER_DT)
    # forced key ties: few distinct k-mers, random meta (all four flag combinations)
    recs["kmer"] = rng.integers(0, max(n // 8, 2), n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    recs["meta"] = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    recs["offset"] = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    got = ctx.sort_kmers(recs)
    exp = oracle.sort_kmers(recs)
    # reference key is (kmer asc, meta desc); offset is not part of it (KMer.h:392-396)
    assert (got["kmer"] == exp["kmer"]).all() and (got["meta"] == exp["meta"]).all()
    # same multiset: normalise the unspecified tie order by offset
    o = np.lexsort((got["offset"], ~got["meta"], got["kmer"]))
    assert (got[o] == exp).all()


def _dataset(synth, seed, n_pairs, n_species=3, n_strains=2, glen=20000, **kw):
    genomes = synth.make_genomes(seed, n_species, n_strains, glen, shared_segment=2000)
    reads, truth = synth.make_paired_reads(seed + 1, genomes, n_pairs, **kw)
    return synth.to_bytes(reads), synth.to_bytes(genomes), truth


def test_find_overlaps_parity(kslam, oracle, synth):
    reads, genomes, _ = _dataset(synth, 11, 500, edge_frac=0.1, n_rate=0.002)
    c = kslam.Context()
    c.set_index(genomes)
    c.load_reads(reads)
    got, raw = c.find_overlaps()
    c.close()
    recs = np.concatenate([oracle.extract_kmers(reads, False, 1), oracle.extract_kmers(genomes, True, 16)])
    exp, exp_raw = oracle.find_overlaps(oracle.sort_kmers(recs), [len(r) for r in reads])
    assert raw == exp_raw
    assert len(got) == len(exp)
    for f in ("read", "entry", "rel", "revcomp"):
        assert (got[f] == exp[f]).all(), f


def _compare_alignments(got, gcig, exp, ecig):
    assert len(got) == len(exp)
    for f in ("read", "entry", "rel", "revcomp", "score", "ref_begin", "ref_end", "query_begin",
              "query_end", "cigar_len"):
        bad = np.nonzero(got[f] != exp[f])[0]
        assert len(bad) == 0, "%s differs at %s: got %s exp %s" % (f, bad[:5], got[bad[:5]], exp[bad[:5]])
    for i in range(len(got)):
        a = gcig[int(got["cigar_off"][i]):int(got["cigar_off"][i]) + int(got["cigar_len"][i])]
        b = ecig[int(exp["cigar_off"][i]):int(exp["cigar_off"][i]) + int(exp["cigar_len"][i])]
        assert (a == b).all(), "cigar %d" % i


@pytest.mark.parametrize("case", ["default", "nocigar", "threshold", "edges", "indels", "len250", "len100"])
def test_align_parity(kslam, oracle, synth, case):
    kw, pk = {}, {}
    if case == "nocigar":
        pk = dict(report_cigar=False)
    if case == "threshold":
        pk = dict(score_threshold=250)
    if case == "edges":
        kw = dict(edge_frac=0.5, n_rate=0.005)
    if case == "indels":
        kw = dict(indel_rate=0.02, sub_rate=0.03)
    if case == "len250":
        kw = dict(read_len=250, frag_mean=450)
    if case == "len100":
        kw = dict(read_len=100, frag_mean=250)
    reads, genomes, _ = _dataset(synth, 21 + len(case), 400, **kw)
    reads = reads + [b"ACGT" * 5, b"", b"N" * 150]  # too short / empty / all-N reads produce nothing
    got, gcig = kslam.align_to_database(reads, genomes, **pk)
    p = oracle.Params.default(report_cigar=pk.get("report_cigar", True),
                              score_threshold=pk.get("score_threshold", 0))
    exp, ecig, _ = oracle.align_to_database(reads, genomes, p)
    assert len(exp) > 300
    _compare_alignments(got, gcig, exp, ecig)


def _revcomp(b):
    return b[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))


@pytest.mark.parametrize("read_len", [150, 400])
def test_wide_bands_every_cigar_bin(kslam, oracle, read_len):
    """banded_sw starts at |refLen - readLen| + 1 (src/ssw.c:616) and doubles: reads with ONE long insertion ask for every
    bin of the CIGAR stage (csrc/cigar.hip cig_bin: 1, 2, 3..4, 5..7, 8..15, 16..31, 32..63) and, at 400 bases, for the
    generic loop beyond (bands of 64 and more, classes by floor(log2)); a second short indel elsewhere makes some attempts
    fall short of the score so that bands double into the next bin.  Both strands; against the oracle."""
    rng = np.random.default_rng(4242 + read_len)
    g = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 60000).tobytes()
    genomes = [g[:30000], g[30000:]]
    ins_lens = [1, 2, 3, 4, 5, 6, 7, 8, 11, 15, 16, 20, 31, 32, 40] if read_len == 150 else \
               [1, 3, 7, 15, 16, 31, 32, 47, 62, 63, 64, 70, 100, 127, 128, 131]
    reads = []
    for I in ins_lens:
        for rep in range(6):
            a = int(rng.integers(100, 29000 - read_len))
            src = genomes[rep & 1]
            cut = int(rng.integers(read_len // 3, read_len // 2))
            body = read_len - I
            junk = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), I).tobytes()
            r = src[a:a + cut] + junk + src[a + cut:a + body]
            if rep >= 2:          # a second, short indel in the longer flank
                at = cut + I + int(rng.integers(20, max(21, body - cut - 20)))
                r = (r[:at] + r[at + 2:] + b"AC") if rep & 1 else (r[:at] + b"G" + r[at:-1])
            assert len(r) == read_len
            reads.append(r if rep % 3 else _revcomp(r))
    got, gcig = kslam.align_to_database(reads, genomes)
    exp, ecig, _ = oracle.align_to_database(reads, genomes, oracle.Params.default())
    assert len(exp) >= len(reads) // 2
    # the bins were really asked for: insertions of every length survive in the CIGARs
    longest = 0
    for i in range(len(exp)):
        ops = ecig[int(exp["cigar_off"][i]):int(exp["cigar_off"][i]) + int(exp["cigar_len"][i])]
        for o in ops:
            if (int(o) & 15) == 1:
                longest = max(longest, int(o) >> 4)
    assert longest >= (32 if read_len == 150 else 100), longest
    _compare_alignments(got, gcig, exp, ecig)


def _sprinkle(rng, seqs, rate, alphabet):
    out = []
    for s in seqs:
        a = np.frombuffer(s, dtype=np.uint8).copy()
        m = rng.random(len(a)) < rate
        a[m] = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), int(m.sum()))
        out.append(a.tobytes())
    return out


@pytest.mark.parametrize("read_len,rate", [(150, 0.004), (100, 0.01), (250, 0.003)])
def test_align_parity_odd_alphabet(kslam, oracle, synth, read_len, rate):
    """Lower case, U/u, IUPAC codes, N and '-' in reads AND entries, on both strands.  The three codings
    of the path disagree exactly on these characters: k-mer coding maps everything but upper-case ACTG
    to A (src/KMer.h:261-263), SSW coding takes lower case and maps U to A, the rest to N
    (src/ssw_cpp.cpp:11-23), and the window reverse-complement only touches upper-case ACGT
    (src/sequenceTools.h:98-116) -- a lower-case 'a' in a reverse-strand window stays an 'a'."""
    rng = np.random.default_rng(4000 + read_len)
    genomes = synth.make_genomes(400 + read_len, 3, 3, 25000, strain_sub=0.02, strain_indel=0.001, shared_segment=2000)
    reads, _ = synth.make_paired_reads(401 + read_len, genomes, 1500, read_len=read_len, frag_mean=2 * read_len + 60,
                                       sub_rate=0.01, indel_rate=0.003, edge_frac=0.1)
    alphabet = b"acgtacgtNnUuRYKMSWBDHVrykm-*."
    gb = _sprinkle(rng, synth.to_bytes(genomes), rate, alphabet)
    rb = _sprinkle(rng, synth.to_bytes(reads), rate * 1.5, alphabet)
    # whole stretches too: a lower-case (soft-masked) region in an entry, an all-lower-case read
    g0 = bytearray(gb[0]); g0[5000:5400] = bytes(g0[5000:5400]).lower(); gb[0] = bytes(g0)
    rb[7] = rb[7].lower()
    rb[8] = rb[8].replace(b"T", b"U")
    got, gcig = kslam.align_to_database(rb, gb)
    exp, ecig, _ = oracle.align_to_database(rb, gb)
    assert len(exp) > 4000 and (exp["revcomp"] == 1).sum() > 1500 and (exp["cigar_len"] > 1).sum() > 300
    _compare_alignments(got, gcig, exp, ecig)


def test_align_chunked_equals_unchunked(kslam, synth):
    reads, genomes, _ = _dataset(synth, 5, 600)
    a, ac = kslam.align_to_database(reads, genomes)
    b, bc = kslam.align_to_database(reads, genomes, max_kmers_per_chunk=20000)
    _compare_alignments(a, ac, b, bc)


def test_resident_path_and_page_locked_results(kslam, synth):
    """load_reads -> align_resident -> take_results (library-owned page-locked buffers, two batches
    outstanding at once) gives what the one-call host-pointer entry point gives."""
    reads, genomes, _ = _dataset(synth, 6, 500)
    exp, ecig = kslam.align_to_database(reads, genomes)
    c = kslam.Context()
    c.set_index(genomes)
    c.load_reads(reads)
    n_out, n_cig = c.align_resident()
    a, ac, rel_a = c.take_results()
    c.align_resident()
    b, bc, rel_b = c.take_results()          # first batch still held
    assert a.ctypes.data != b.ctypes.data and len(a) == n_out and len(ac) == n_cig
    _compare_alignments(a, ac, exp, ecig)
    _compare_alignments(b, bc, exp, ecig)
    f, fc = c.fetch_results(n_out, n_cig)
    _compare_alignments(f, fc, exp, ecig)
    rel_a()
    rel_b()
    c.close()


def test_results_released_from_a_second_thread(kslam, synth):
    """The pipelined callers (bench.py's sam_pipeline) hand batch k's page-locked
    buffers back from a worker thread while the main thread takes batch k+1's: the pool is locked
    (ADVICE r1).  40 rounds with up to three batches outstanding, results identical every time."""
    import queue
    import threading
    reads, genomes, _ = _dataset(synth, 8, 400)
    c = kslam.Context()
    c.set_index(genomes)
    c.load_reads(reads)
    c.align_resident()
    ref_ov, ref_cg, rel = c.take_results()
    ref_ov, ref_cg = ref_ov.copy(), ref_cg.copy()
    rel()
    q, bad = queue.Queue(maxsize=3), []

    def worker():
        while True:
            item = q.get()
            if item is None:
                return
            ov, cg, release = item
            if not ((ov == ref_ov).all() and np.array_equal(cg, ref_cg)):
                bad.append(1)
            release()
    t = threading.Thread(target=worker)
    t.start()
    for _ in range(40):
        c.align_resident()
        q.put(c.take_results())
    q.put(None)
    t.join()
    c.close()
    assert not bad


def test_pipelined_entry_gives_the_same_batches(kslam, synth):
    """kslam_align_batch_async / kslam_wait_batch: three different batches kept in flight over the two
    worker lanes, waited for out of order, repeated; every result equals the synchronous entry's.
    An unsupported batch fails at ITS wait, with its message, and the lanes keep working."""
    genomes = synth.make_genomes(31, 3, 2, 25000, shared_segment=1500)
    batches = []
    for k, n in enumerate([300, 1, 451]):
        reads, _ = synth.make_paired_reads(40 + k, genomes, n, indel_rate=0.004, edge_frac=0.05)
        batches.append(synth.to_bytes(reads))
    batches.append([])                                   # an empty batch is a batch
    gb = synth.to_bytes(genomes)
    c = kslam.Context()
    c.set_index(gb)
    expect = [c.align_batch(b) for b in batches]
    order = [0, 1, 2, 3, 2, 0, 1, 1, 3, 0, 2, 2]
    tickets = [c.submit_batch(batches[i]) for i in order[:3]]
    assert tickets == [0, 1, 2]
    for k in range(3, len(order) + 3):
        if k < len(order):
            tickets.append(c.submit_batch(batches[order[k]]))
        j = k - 3
        ov, cg = c.wait_batch(tickets[j])
        e_ov, e_cg = expect[order[j]]
        assert ov.tobytes() == e_ov.tobytes() and cg.tobytes() == e_cg.tobytes(), j
    with pytest.raises(kslam.KslamError, match="ticket"):
        c.wait_batch(tickets[0])
    # out-of-order waits + an unsupported batch in the middle
    t0 = c.submit_batch(batches[0])
    t_bad = c.submit_batch([b"ACGT" * 2500])             # 10 000 bases: beyond the supported read length (9 000)
    t2 = c.submit_batch(batches[2])
    ov, cg = c.wait_batch(t2)
    assert ov.tobytes() == expect[2][0].tobytes()
    with pytest.raises(kslam.KslamError, match="9000"):
        c.wait_batch(t_bad)
    ov, cg = c.wait_batch(t0)
    assert ov.tobytes() == expect[0][0].tobytes() and cg.tobytes() == expect[0][1].tobytes()
    # a new index while the lanes exist: they see it
    c.set_index(gb[:2])
    e2 = c.align_batch(batches[0])
    got = c.wait_batch(c.submit_batch(batches[0]))
    assert got[0].tobytes() == e2[0].tobytes() and len(e2[0]) < len(expect[0][0])
    c.submit_batch(batches[2])                           # left outstanding on purpose: destroy cleans up
    c.close()


@pytest.mark.parametrize("read_len,odd", [(150, False), (100, True), (250, False)])
def test_row_details_equal_the_reference_walk(kslam, synth, read_len, odd):
    """kslam_row_details (NM, log-probability, MD text per overlap record) against the plain restatement of
    getCigarAndMD's walk (src/SAM.h:101-237, tests/rowdetails_ref.py): indels, clipped ends, reads over
    the entry ends, both strands, odd characters, qualities over the whole phred range; and the pipelined
    entry returns the same rows."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from rowdetails_ref import row_details
    rng = np.random.default_rng(77 + read_len)
    genomes = synth.make_genomes(500 + read_len, 3, 3, 25000, strain_sub=0.02, strain_indel=0.002, shared_segment=2000)
    reads, _ = synth.make_paired_reads(501 + read_len, genomes, 1200, read_len=read_len, frag_mean=2 * read_len + 60,
                                       sub_rate=0.02, indel_rate=0.006, edge_frac=0.1, n_rate=0.002)
    rb, gb = synth.to_bytes(reads), synth.to_bytes(genomes)
    if odd:
        gb = _sprinkle(rng, gb, 0.004, b"acgtNnUuRY-")
        rb = _sprinkle(rng, rb, 0.006, b"acgtNnUuRY-")
    quals = [bytes(rng.integers(33, 33 + 94, len(b), dtype=np.uint8)) for b in rb]
    c = kslam.Context()
    c.set_index(gb)
    c.load_reads(rb)
    n_out, n_cig = c.align_resident()
    ov, cg = c.fetch_results(n_out, n_cig)
    with pytest.raises(kslam.KslamError, match="load_qualities"):
        c.row_details()
    c.load_qualities(quals)
    c.row_details()
    det, md = c.take_row_details(n_out)
    edet, emd = row_details(ov, cg, rb, quals, gb, kslam.ROW_DETAIL_DT)
    assert (ov["revcomp"] == 1).sum() > 1000 and (ov["cigar_len"] > 1).sum() > 500 and (ov["query_begin"] > 0).sum() > 50
    for f in ("nm", "md_len", "md_off", "flags"):
        assert (det[f] == edet[f]).all(), f
    assert (det["logp"].view(np.uint64) == edet["logp"].view(np.uint64)).all()
    assert md.tobytes() == emd.tobytes()
    # the pipelined entry with qualities: same rows
    import ctypes as C
    keep_b = [C.create_string_buffer(b, len(b) + 1) for b in rb]
    keep_q = [C.create_string_buffer(q, len(q) + 1) for q in quals]
    bp = (C.c_char_p * len(rb))(*[C.cast(x, C.c_char_p) for x in keep_b])
    qp = (C.c_char_p * len(rb))(*[C.cast(x, C.c_char_p) for x in keep_q])
    lens = np.array([len(b) for b in rb], dtype=np.uint32)
    t1 = c.submit_batch_full(len(rb), C.cast(bp, C.c_void_p), C.cast(qp, C.c_void_p), lens.ctypes.data)
    t2 = c.submit_batch_full(len(rb), C.cast(bp, C.c_void_p), None, lens.ctypes.data)
    o1, c1, d1, m1, rel1 = c.collect_batch(t1)
    o2, c2, d2, m2, rel2 = c.collect_batch(t2)
    assert o1.tobytes() == ov.tobytes() and c1.tobytes() == cg.tobytes() and o2.tobytes() == ov.tobytes()
    assert d1.tobytes() == det.tobytes() and m1.tobytes() == md.tobytes() and len(d2) == 0 and len(m2) == 0
    rel1()
    rel2()
    # a quality character that is no phred+33 value, in an aligned column: flagged, not fatal here
    bad = list(quals)
    bad[int(ov["read"][0])] = b"\x10" * len(rb[int(ov["read"][0])])
    c.load_qualities(bad)
    c.row_details()
    d3, _ = c.take_row_details(n_out)
    assert (d3["flags"][ov["read"] == ov["read"][0]] & 1).all() and (d3["flags"] & 1).sum() < len(d3)
    c.close()


def test_empty_batch(kslam, synth):
    _, genomes, _ = _dataset(synth, 5, 1)
    ov, cg = kslam.align_to_database([], genomes)
    assert len(ov) == 0 and len(cg) == 0


def test_scoring_the_reference_types_cannot_hold_fails_loudly(kslam):
    """match / mismatch beyond the int8_t score matrix, gap penalties beyond the Aligner's uint8_t (src/ssw_cpp.cpp:25-49,
    src/ssw_cpp.h), match 0"""
    for kw in ({"match": 128}, {"mismatch": 128}, {"gap_open": 256}, {"gap_extend": 300, "gap_open": 301}, {"match": 0}):
        with pytest.raises(kslam.KslamError):
            kslam.Context(**kw)


@pytest.mark.parametrize("scoring", [(2, 9, 5, 2), (5, 4, 10, 10), (2, 8, 2, 3), (1, 1, 1, 1), (3, 7, 4, 4)])
def test_scoring_outside_the_envelope_equals_the_striped_reference(kslam, oracle, synth, scoring):
    """`SLAM --gap-open / --gap-extend` takes any value (src/main.cpp:44-55).  Outside `gapE < gapO, mismatch <= gapO + gapE`
    the reference's answer depends on its striped evaluation order (Lazy-F only extends, E is not refreshed:
    src/ssw.c:274-305, 512-526); such contexts run every candidate through k_sw_striped (the SSE lanes played literally)
    and the literal banded_sw.  Against the oracle's striped emulation, which tests/test_oracle.py holds to the real ssw.c on
    these very scorings: every row field and CIGAR word."""
    m, x, go, ge = scoring
    genomes = synth.make_genomes(91, 2, 3, 12000, strain_sub=0.03, strain_indel=0.002, shared_segment=1500)
    reads, _ = synth.make_paired_reads(92, genomes, 350, read_len=120, sub_rate=0.03, indel_rate=0.01, n_rate=0.003, edge_frac=0.1)
    rb, gb = synth.to_bytes(reads), synth.to_bytes(genomes)
    rb += [b"", b"ACGT" * 10, gb[0][100:131]]
    for thr in (0, 2 * m * 20):
        got, gcig = kslam.align_to_database(rb, gb, match=m, mismatch=x, gap_open=go, gap_extend=ge, score_threshold=thr)
        exp, ecig, _ = oracle.align_to_database(rb, gb, oracle.Params.default(match=m, mismatch=x, gap_open=go, gap_extend=ge,
                                                                              score_threshold=thr))
        assert len(exp) > 800
        _compare_alignments(got, gcig, exp, ecig)


# ---------------------------------------------------------------------------
# committed golden fixtures (made from the real reference pieces in the build container,
# tests/golden/make_golden.py): nothing here needs /root/reference at run time
# ---------------------------------------------------------------------------
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _split(flat, lens):
    out, p = [], 0
    for n in lens:
        out.append(flat[p:p + int(n)])
        p += int(n)
    return out


def test_golden_kmer_vectors(ctx):
    z = np.load(os.path.join(GOLD, "kmer_vectors.npz"))
    seqs = [s.tobytes() for s in _split(z["seqs"], z["seq_len"])]
    rr = ctx.extract_kmers(seqs, False, 1)
    rg = ctx.extract_kmers(seqs, True, 16)
    assert (rr == z["reads_gap1"]).all() and (rg == z["genbank_gap16"]).all()
    srt = ctx.sort_kmers(np.concatenate([rr, rg]))
    assert (srt["kmer"] == z["sorted_kmer"]).all() and (srt["meta"] == z["sorted_meta"]).all()


def test_golden_align_small(kslam):
    z = np.load(os.path.join(GOLD, "align_small.npz"))
    reads = [s.tobytes() for s in _split(z["reads"], z["read_len"])]
    genomes = [s.tobytes() for s in _split(z["genomes"], z["genome_len"])]
    got, gcig = kslam.align_to_database(reads, genomes)
    _compare_alignments(got, gcig, z["alignments"], z["cigars"])


def test_golden_ssw_vectors_through_the_abi(kslam):
    """Each (read, ref) pair of the real-ssw.c vector file becomes a 1-read / 1-entry batch whose
    only candidate is the planted k-mer hit; checks score + coordinates + CIGAR through the ABI.
    Only vectors that share an exact 32-mer at a 16-aligned reference offset produce a candidate."""
    z = np.load(os.path.join(GOLD, "ssw_vectors.npz"))
    reads, refs = _split(z["reads"], z["read_len"]), _split(z["refs"], z["ref_len"])
    cigs = _split(z["cigars"], z["cigar_len"])
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    checked = 0
    ctxs = {}
    for i in range(len(reads)):
        prm = tuple(int(v) for v in z["params"][i])
        rd, rf = lut[reads[i]].tobytes(), lut[refs[i]].tobytes()
        if len(rf) < len(rd) or len(rf) > len(rd):  # window must equal ref: need |ref| == |read| case only
            continue
        if prm not in ctxs:
            ctxs[prm] = kslam.Context(match=prm[0], mismatch=prm[1], gap_open=prm[2], gap_extend=prm[3])
        c = ctxs[prm]
        c.set_index([rf])
        ov, cg = c.align_batch([rd])
        hit = ov[(ov["rel"] == 0) & (ov["revcomp"] == 0)]
        if len(hit) == 0:
            continue
        h = hit[0]
        exp = tuple(int(v) for v in z["results"][i])
        assert (int(h["score"]), int(h["ref_begin"]), int(h["ref_end"]), int(h["query_begin"]),
                int(h["query_end"])) == exp, i
        assert np.array_equal(cg[int(h["cigar_off"]):int(h["cigar_off"]) + int(h["cigar_len"])], cigs[i]), i
        checked += 1
    for c in ctxs.values():
        c.close()
    assert checked >= 20, checked


def _cols(z, name):
    flat, off = z[name], z[name + "_off"]
    return [flat[int(off[i]):int(off[i + 1])].tobytes() for i in range(len(off) - 1)]


def _compare_modulo_revcomp_ties(got, gcig, exp, ecig, ties):
    """exp = answers of the real reference at one thread; overlapSort has no revComp in its key (src/Overlap.h:87-98),
    so where the raw list holds the same (read, entry, rel) with both revComp values the survivor's flag is a tie"""
    assert len(got) == len(exp)
    amb = np.array([(int(r), int(e), int(l)) in ties for r, e, l in zip(exp["read"], exp["entry"], exp["rel"])], dtype=bool)
    for f in ("read", "entry", "rel", "revcomp", "score", "ref_begin", "ref_end", "query_begin", "query_end", "cigar_len"):
        ok = got[f] == exp[f]
        if f == "revcomp":
            ok = ok | amb
        assert ok.all(), (f, np.flatnonzero(~ok)[:5])
    if ecig is not None:
        for i in range(len(got)):
            a = gcig[int(got["cigar_off"][i]):int(got["cigar_off"][i]) + int(got["cigar_len"][i])]
            b = ecig[int(exp["cigar_off"][i]):int(exp["cigar_off"][i]) + int(exp["cigar_len"][i])]
            assert (a == b).all(), "cigar %d" % i
    return int(amb.sum())


def test_golden_join_vectors(kslam, ctx):
    """Through the ABI against the recorded answers of the reference's OWN sortKMers, findOverlaps_parallel and
    alignToDatabase (oracle/_ref/libjoin_ref.so -> tests/golden/join_vectors.npz): pile-ups with several genome records,
    rc/fwd mixes, k-mer 0, genome ends, ragged and empty reads, odd letters, the revComp tie."""
    z = np.load(os.path.join(GOLD, "join_vectors.npz"))
    reads, genomes = _cols(z, "reads"), _cols(z, "genomes")
    ties = {tuple(int(v) for v in t) for t in z["ties"]}
    recs = np.concatenate([ctx.extract_kmers(reads, False, 1), ctx.extract_kmers(genomes, True, 16)])
    srt = ctx.sort_kmers(recs)
    assert (srt["kmer"] == z["sorted"]["kmer"]).all() and (srt["meta"] == z["sorted"]["meta"]).all()
    c = kslam.Context()
    c.set_index(genomes)
    c.load_reads(reads)
    got, raw = c.find_overlaps()
    c.close()
    assert raw == len(z["raw"])
    exp = z["deduped"]
    assert len(got) == len(exp)
    amb = np.array([(int(r), int(e), int(l)) in ties for r, e, l in zip(exp["read"], exp["entry"], exp["rel"])], dtype=bool)
    for f in ("read", "entry", "rel"):
        assert (got[f] == exp[f]).all(), f
    assert ((got["revcomp"] == exp["revcomp"]) | amb).all() and 0 < amb.sum() < 10
    g, gc = kslam.align_to_database(reads, genomes)
    _compare_modulo_revcomp_ties(g, gc, z["alignments_thr0"], z["cigars_thr0"], ties)
    g, gc = kslam.align_to_database(reads, genomes, score_threshold=150)
    _compare_modulo_revcomp_ties(g, gc, z["alignments_thr150"], z["cigars_thr150"], ties)
    g, gc = kslam.align_to_database(reads, genomes, report_cigar=False)
    _compare_modulo_revcomp_ties(g, gc, z["alignments_nocigar"], None, ties)
    assert len(gc) == 0


def test_golden_align_vectors_through_the_abi(kslam):
    """The recorded answers of the reference's OWN Aligner::Align (tests/golden/align_vectors.npz) through the ABI: a case
    (query, ref, ref_len) with ref_len == |ref| <= |query| becomes a 1-read / 1-entry batch; every candidate with
    rel <= 0 on the forward strand has the window entry.substr(0, |query|) == ref (src/SmithWaterman.h:203-208), so its
    answer is the recorded one.  ASCII incl. lower case / U / IUPAC on both sides; thresholded and disabled CIGAR."""
    z = np.load(os.path.join(GOLD, "align_vectors.npz"))
    checked = 0
    assert len(z["param_sets"]) == 5          # two scorings inside the envelope, three outside it (k_sw_striped)
    for params in (tuple(int(v) for v in ps) for ps in z["param_sets"]):
        tag = "p%d%d%d%d" % params
        qs, rs, ns = _cols(z, tag + "_query"), _cols(z, tag + "_ref"), z[tag + "_ref_len"]
        for thr, want in ((0, 1), (120, 1), (0, 0)):
            k = "%s_thr%d_cigar%d" % (tag, thr, want)
            cigs = _split(z[k + "_cigars"], z[k + "_cigar_len"])
            c = kslam.Context(match=params[0], mismatch=params[1], gap_open=params[2], gap_extend=params[3],
                              score_threshold=thr, report_cigar=bool(want))
            for i in range(len(qs)):
                if int(ns[i]) != len(rs[i]) or len(rs[i]) > len(qs[i]):
                    continue
                c.set_index([rs[i]])
                ov, cg = c.align_batch([qs[i]])
                hit = ov[(ov["rel"] <= 0) & (ov["revcomp"] == 0)]
                for h in hit:
                    exp = tuple(int(v) for v in z[k + "_results"][i])
                    assert (int(h["score"]), int(h["ref_begin"]), int(h["ref_end"]), int(h["query_begin"]),
                            int(h["query_end"])) == exp, (k, i)
                    assert np.array_equal(cg[int(h["cigar_off"]):int(h["cigar_off"]) + int(h["cigar_len"])], cigs[i]), (k, i)
                    checked += 1
            c.close()
    assert checked >= 100, checked


# ---------------------------------------------------------------------------
# size-independent properties on a larger batch (the oracle would take minutes here)
# ---------------------------------------------------------------------------
def test_properties_large_batch(kslam, synth):
    genomes = synth.make_genomes(77, 8, 3, 200000)
    reads, truth = synth.make_paired_reads(78, genomes, 20000, sub_rate=0.0, indel_rate=0.0, unmapped_frac=0.05)
    rb, gb = synth.to_bytes(reads), synth.to_bytes(genomes)
    c = kslam.Context(max_kmers_per_chunk=1 << 20)   # forces several internal chunks
    c.set_index(gb)
    ov, cg = c.align_batch(rb)
    ov2, cg2 = c.align_batch(rb)                      # idempotence
    tm = c.timings()
    c.close()
    assert tm["n_chunks"] > 1
    assert (ov == ov2).all() and np.array_equal(cg, cg2)
    # sortedness: (read, entry, rel) non-decreasing, reference order (src/Overlap.h:87-98)
    key = (ov["read"].astype(np.int64) << 40) | (ov["entry"].astype(np.int64) << 24) | (ov["rel"].astype(np.int64) + (1 << 20))
    assert (np.diff(key) >= 0).all()
    # error-free planted reads: best hit on the source genome has score 2 * L and CIGAR "150M"
    n = len(truth)
    best = {}
    for i in np.nonzero(ov["score"] == 300)[0]:
        best.setdefault(int(ov["read"][i]), []).append(i)
    found = 0
    for p, (g, start, flip) in enumerate(truth):
        if g < 0:
            continue
        for rid in (p, p + n):
            hits = [i for i in best.get(rid, []) if int(ov["entry"][i]) == g]
            assert hits, (p, rid)
            h = hits[0]
            assert int(ov["cigar_len"][h]) == 1 and int(cg[int(ov["cigar_off"][h])]) == (150 << 4)
            assert int(ov["ref_end"][h]) - int(ov["ref_begin"][h]) == 149
            # src/Tests.h:161-264: the planted (offset, revComp) comes back.  The fragment starts at
            # `start`; its first mate is the forward strand unless the fragment was flipped.
            rel, rc = int(ov["rel"][h]), int(ov["revcomp"][h])
            first_mate = rid == p
            assert rc == int(first_mate == bool(flip)), (p, rid, rc, flip)
            assert rel == int(ov["ref_begin"][h]) and int(ov["query_begin"][h]) == 0
            if rc == 0:
                assert rel == start, (p, rid, rel, start)
            else:   # the reverse mate ends where the fragment ends: start + fragment length - 150
                assert 150 <= rel - start + 150 <= 1000, (p, rid, rel, start)
            found += 1
    assert found > 30000
    # unmapped pairs (random sequence) have no alignment at all
    for p, (g, _, _) in enumerate(truth):
        if g < 0:
            assert not (ov["read"] == p).any()


@pytest.mark.parametrize("read_len,frag", [(150, 350), (250, 500), (400, 700)])
def test_banded_sw_equals_full_matrix_sw(kslam, synth, monkeypatch, read_len, frag):
    """The provably-banded anti-diagonal SW kernels (32 / 64 / 128 diagonals) and the full-matrix
    kernel must agree on every candidate (divergent strains, indels, N, reads hanging off the
    genome ends)."""
    genomes = synth.make_genomes(91, 6, 4, 60000, strain_sub=0.03, strain_indel=0.002)
    reads, _ = synth.make_paired_reads(92, genomes, 6000 if read_len == 150 else 2500, read_len=read_len,
                                       frag_mean=frag, sub_rate=0.02, indel_rate=0.004, n_rate=0.002,
                                       edge_frac=0.05)
    rb, gb = synth.to_bytes(reads), synth.to_bytes(genomes)
    c = kslam.Context()
    c.set_index(gb)
    a, ac = c.align_batch(rb)
    monkeypatch.setenv("KSLAM_SW_FULL", "1")
    c.reload_tuning()                     # the switches are read at kslam_create, not per batch
    b, bc = c.align_batch(rb)
    monkeypatch.delenv("KSLAM_SW_FULL")
    c.close()
    assert len(a) > 8000
    _compare_alignments(a, ac, b, bc)


@pytest.mark.parametrize("read_len", [150, 250])
def test_every_kernel_variant_gives_the_same_alignments(kslam, synth, monkeypatch, read_len):
    """The hot path picks among several implementations per candidate (band tiers planned from the
    seed diagonal / full matrix for the scores; band in registers / systolic / one lane with LDS rows
    for the CIGAR).  Forcing each choice in turn must not change a single field or CIGAR op."""
    genomes = synth.make_genomes(191, 6, 4, 60000, strain_sub=0.03, strain_indel=0.003)
    reads, _ = synth.make_paired_reads(192, genomes, 5000 if read_len == 150 else 2500, read_len=read_len,
                                       frag_mean=2 * read_len + 50, sub_rate=0.02, indel_rate=0.006,
                                       n_rate=0.002, edge_frac=0.05)
    rb, gb = synth.to_bytes(reads), synth.to_bytes(genomes)
    c = kslam.Context()
    c.set_index(gb)
    base, bcig = c.align_batch(rb)
    assert len(base) > 8000 and (base["cigar_len"] > 1).sum() > 1000
    variants = [{"KSLAM_CIGAR_SYS": "0", "KSLAM_CIGAR_REG": "0"},   # every CIGAR on the literal one-lane kernel
                {"KSLAM_CIGAR_SYS": "255"},                          # systolic for every band class
                {"KSLAM_CIGAR_SYS": "0"},                            # registers for narrow, one-lane for wide
                {"KSLAM_CIGAR_DIRS": "lds", "KSLAM_CIGAR_SYS": "0", "KSLAM_CIGAR_REG": "0"},
                {"KSLAM_CIGAR_TB": "inline"},                        # systolic tracebacks at the end of the DP kernel
                {"KSLAM_CIGAR_TB": "inline", "KSLAM_CIGAR_SYS": "255"},
                {"KSLAM_SW_FULL": "1"},                              # full-matrix scores only
                {"KSLAM_SW_NO48": "1"},                              # tiers 16 / 32 / 64 / 96
                {"KSLAM_SW_NO96": "1"},                              # no 96-diagonal tier
                {"KSLAM_SW_UNKNOWN_ND": "16"},                       # gapped candidates start at the narrowest band
                {"KSLAM_SW_UNKNOWN_ND": "64"},
                {"KSLAM_SORT_DIGIT_BYTES": "0"},                     # radix histograms re-read the records
                {"KSLAM_JOIN_GROUP_ORDER": "0"},                     # overlap keys through all their radix passes (no group ranking)
                {"KSLAM_SWEEP_ROOM": "0"},                           # CIGAR bins / SW tiers sized without room: every candidate sent on takes the left-over rounds
                {"KSLAM_SW_SWEEP": "0"},                             # a read-back in front of every SW tier
                {"KSLAM_JOIN": "merge"},                             # the merge join instead of the probe (join.hip: k_join_merge)
                {"KSLAM_JOIN": "merge", "KSLAM_SORT_BYTES": "1"},    # ... with read records ordered by their top byte only
                {"KSLAM_JOIN": "merge", "KSLAM_SORT_BYTES": "8"}]    # ... and by the whole key
    for env in variants:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c.reload_tuning()                 # the switches are read at kslam_create, not per batch
        got, gcig = c.align_batch(rb)
        if "KSLAM_SWEEP_ROOM" in env:     # (the SW tiers' sweep is sized from the context's previous chunk: once more)
            got, gcig = c.align_batch(rb)
        for k in env:
            monkeypatch.delenv(k)
        _compare_alignments(got, gcig, base, bcig)
    c.close()


def test_sort_selftest_large(ctx):
    """full-size property: 64 M random records come out ordered (stable) after the 8-pass sort"""
    ms, ms_launch, inv = ctx.selftest_sort(1 << 26, 1)
    assert inv == 0


def test_ragged_reads_and_odd_entries(kslam, oracle, synth):
    """ragged read lengths (incl. < 32 and empty), entries shorter than k, an empty entry"""
    rng = np.random.default_rng(321)
    genomes = synth.make_genomes(55, 3, 2, 15000)
    reads, _ = synth.make_paired_reads(56, genomes, 300, read_len=300, frag_mean=600, indel_rate=0.003)
    reads = [r[:int(rng.integers(0, 301))] for r in reads]
    gb = synth.to_bytes(genomes) + [b"ACGT" * 5, b"", b"A" * 40, synth.random_bases(rng, 33).tobytes()]
    rb = synth.to_bytes(reads)
    got, gcig = kslam.align_to_database(rb, gb)
    exp, ecig, _ = oracle.align_to_database(rb, gb)
    assert len(exp) > 200
    _compare_alignments(got, gcig, exp, ecig)


def test_repeated_genomes_long_runs(kslam, oracle, synth):
    """collisions: 120 identical entries make every k-mer run 120 long (whole-workgroup expansion
    path of the join) and every read a 120-way multi-hit"""
    rng = np.random.default_rng(77)
    g = synth.random_bases(rng, 4000)
    genomes = [g.tobytes()] * 120
    reads, _ = synth.make_paired_reads(78, [g], 60, unmapped_frac=0.0)
    rb = synth.to_bytes(reads)
    got, gcig = kslam.align_to_database(rb, genomes)
    exp, ecig, _ = oracle.align_to_database(rb, genomes)
    assert len(exp) >= 120 * 100
    _compare_alignments(got, gcig, exp, ecig)


@pytest.mark.parametrize("period", [7, 2])
def test_long_tandem_repeat_thousands_of_hits_per_read_and_entry(kslam, oracle, synth, period):
    """One entry that is a 21 kb tandem repeat: every read k-mer meets hundreds of genome k-mers, a read has
    thousands of candidate positions on ONE entry.  Period 7: neighbouring hits lie >= 3 apart, every one is
    kept (each is the head of its own run in k_dedupe_flags -- the ADVICE-r1 case that used to walk a whole
    (read, entry) segment on one lane); period 2: dense runs, the greedy rule thins them."""
    rng = np.random.default_rng(31 + period)
    unit = synth.random_bases(rng, period)
    if period == 2:
        unit = np.frombuffer(b"AC", dtype=np.uint8)
    g = np.resize(unit, 21000)
    flank = synth.random_bases(rng, 3000)
    genome = np.concatenate([flank, g, flank])
    reads = []
    for _ in range(24):
        p = int(rng.integers(2900, 3000 + 21000 - 100))
        r = synth.mutate(rng, genome[p:p + 150], 0.01, 0.0)
        reads.append(synth.revcomp(r) if rng.random() < 0.5 else r)
    rb, gb = synth.to_bytes(reads), [genome.tobytes(), synth.random_bases(rng, 5000).tobytes()]
    c = kslam.Context()
    c.set_index(gb)
    c.load_reads(rb)
    got_t, raw = c.find_overlaps()
    recs = np.concatenate([oracle.extract_kmers(rb, False, 1), oracle.extract_kmers(gb, True, 16)])
    exp_t, exp_raw = oracle.find_overlaps(oracle.sort_kmers(recs), [len(r) for r in rb])
    assert raw == exp_raw and raw > 100000
    assert len(got_t) == len(exp_t) and (len(exp_t) > 20000 if period == 7 else len(exp_t) > 3000)
    for f in ("read", "entry", "rel", "revcomp"):
        assert (got_t[f] == exp_t[f]).all(), f
    got, gcig = c.align_batch(rb)
    c.close()
    exp, ecig, _ = oracle.align_to_database(rb, gb)
    _compare_alignments(got, gcig, exp, ecig)


@pytest.mark.parametrize("unknown_nd", [None, "32", "48", "64"])
def test_equal_best_scores_on_neighbouring_diagonals(kslam, oracle, monkeypatch, unknown_nd):
    """tests/golden/sw_tie_cases.json: reads whose best score is reached by two alignments ending a few
    diagonals apart (a tandem repeat after a unique seed; window cut by the genome end), found with
    a model of the band sweep's visiting order.  The reference reports the smallest end column, then
    the smallest end row (ssw.c:316-342) -- also when both ends fall to the same lane of the sweep."""
    import json
    cases = json.load(open(os.path.join(GOLD, "sw_tie_cases.json")))
    if unknown_nd:
        monkeypatch.setenv("KSLAM_SW_UNKNOWN_ND", unknown_nd)
    bad = []
    for k, c in enumerate(cases):
        reads, genomes = [c["read"].encode()], [c["ref"].encode()]
        got, gcig = kslam.align_to_database(reads, genomes)
        exp, ecig, _ = oracle.align_to_database(reads, genomes)
        assert len(exp) == 1 and (int(exp["ref_end"][0]), int(exp["query_end"][0])) == tuple(c["true"])
        try:
            _compare_alignments(got, gcig, exp, ecig)
        except AssertionError as e:
            bad.append((k, c["dpl"], str(e)[:120]))
    assert not bad, bad


def test_overlap_keys_ordered_by_groups_equal_the_full_sort(kslam, oracle, synth, monkeypatch):
    """join.hip group_order: the overlap keys are radix-sorted by their high bytes only and ranked inside each (read, entry)
    group; a group of more than 64 keys sends the chunk -- and the context from then on -- through all the passes.  Reads from
    a plain genome (groups of a few keys), reads that meet an entry at 3 / 5 / 9 loci (rRNA-like copies: groups of tens),
    and reads inside tandem repeats (groups of hundreds: the fallback, in the middle of the context's life), against the
    oracle and against the same context with the switch off; then a plain batch again on the context that has fallen back."""
    rng = np.random.default_rng(77)
    seg = synth.random_bases(rng, 400)
    multi = np.concatenate([np.concatenate([synth.random_bases(rng, 900), synth.mutate(rng, seg, 0.01, 0.0)]) for _ in range(9)])
    plain = [synth.random_bases(rng, 20000), multi]
    unit = synth.random_bases(rng, 5)
    tandem = np.concatenate([synth.random_bases(rng, 300), np.resize(unit, 1200), synth.random_bases(rng, 300)])

    def reads_from(genomes, n, tandem_too=False):
        out = []
        for k in range(n):
            g = genomes[k % len(genomes)]
            at = int(rng.integers(0, len(g) - 150))
            r = synth.mutate(rng, g[at:at + 150], 0.02, 0.003)[:150]
            out.append(synth.revcomp(r) if k % 3 == 0 else r)
        return out
    seg_reads = [synth.mutate(rng, seg[i:i + 150], 0.01, 0.002)[:150] for i in range(0, 240, 8)]      # 9 loci each
    batch1 = synth.to_bytes(reads_from(plain, 600) + seg_reads)
    batch2 = synth.to_bytes(reads_from([tandem, plain[0]], 400))                                       # tandem repeats: long groups
    gb = synth.to_bytes(plain + [tandem])
    c = kslam.Context()
    c.set_index(gb)
    for rb in (batch1, batch2, batch1):
        got, gcig = c.align_batch(rb)
        exp, ecig, _ = oracle.align_to_database(rb, gb)
        _compare_alignments(got, gcig, exp, ecig)
    monkeypatch.setenv("KSLAM_JOIN_GROUP_ORDER", "0")
    c2 = kslam.Context()
    c2.set_index(gb)
    for rb in (batch1, batch2):
        a, ac = c.align_batch(rb)
        b, bc = c2.align_batch(rb)
        _compare_alignments(a, ac, b, bc)
    monkeypatch.delenv("KSLAM_JOIN_GROUP_ORDER")
    c.close()
    c2.close()


def test_merge_join_equals_the_probe_and_the_oracle(kslam, oracle, synth, monkeypatch):
    """join.hip k_join_merge (KSLAM_JOIN=merge): a workgroup streams the genome key range its tile of sorted read records
    meets through LDS in pieces of 4096 keys and places every read key by binary search.  Cases: tiles that span several
    pieces (2 Mb of genome against a few thousand reads), runs of equal genome keys of hundreds (tandem repeats: runs cut by
    piece boundaries, the whole-block expansion), an entry met at nine loci, reads that are not from the database, a batch
    of three reads, the filter switched off (every read k-mer joins: tiles of one piece); against the oracle and the probe."""
    rng = np.random.default_rng(4242)
    seg = synth.random_bases(rng, 400)
    multi = np.concatenate([np.concatenate([synth.random_bases(rng, 900), synth.mutate(rng, seg, 0.01, 0.0)]) for _ in range(9)])
    unit = synth.random_bases(rng, 5)
    tandem = np.concatenate([synth.random_bases(rng, 300), np.resize(unit, 3000), synth.random_bases(rng, 300)])
    big = [synth.random_bases(rng, 700000) for _ in range(3)]
    genomes = big + [multi, tandem]
    reads = []
    for k in range(3000):
        g = genomes[k % len(genomes)]
        at = int(rng.integers(0, len(g) - 150))
        r = synth.mutate(rng, g[at:at + 150], 0.02, 0.003)[:150]
        reads.append(synth.revcomp(r) if k % 3 == 0 else r)
    reads += [synth.random_bases(rng, 150) for _ in range(200)]                        # not from the database
    rb, gb = synth.to_bytes(reads), synth.to_bytes(genomes)
    exp, ecig, _ = oracle.align_to_database(rb, gb)
    few = rb[:3]
    fexp, fecig, _ = oracle.align_to_database(few, gb)
    probe = kslam.Context()
    probe.set_index(gb)
    base, bcig = probe.align_batch(rb)
    _compare_alignments(base, bcig, exp, ecig)
    for env in ({}, {"KSLAM_FILTER_BITS": "0"}, {"KSLAM_SORT_BYTES": "8"}, {"KSLAM_BUCKET_BITS_EXACT": "12"}):
        monkeypatch.setenv("KSLAM_JOIN", "merge")
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = kslam.Context()
        c.set_index(gb)
        got, gcig = c.align_batch(rb)
        _compare_alignments(got, gcig, exp, ecig)
        got, gcig = c.align_batch(few)
        _compare_alignments(got, gcig, fexp, fecig)
        assert c.timings()["n_overlaps_raw"] == probe_raw(probe, few)
        c.close()
        monkeypatch.delenv("KSLAM_JOIN")
        for k in env:
            monkeypatch.delenv(k)
    probe.close()


def probe_raw(ctx, reads):
    ctx.align_batch(reads)
    return ctx.timings()["n_overlaps_raw"]


def test_filter_built_block_by_block_equals_the_atomic_build(kslam, oracle, synth, monkeypatch):
    """filter.hip since round 6: the membership filter is assembled 32 KB block by block in LDS from the keys' probe words
    ordered by block, instead of four scattered atomics per key.  Same bits: the same read k-mers survive (count) and the
    same alignments come out, for a filter of several blocks, for KSLAM_FILTER_BITS at its smallest, for an index with
    repeated keys and k-mer 0 (poly-A), and for an index without a single k-mer."""
    rng = np.random.default_rng(99)
    genomes = [synth.random_bases(rng, 300000) for _ in range(3)] + [np.frombuffer(b"A" * 500, dtype=np.uint8).copy()]
    genomes.append(np.concatenate([genomes[0][1000:3000], genomes[1][500:900]]))             # repeated keys
    reads = []
    for k in range(2500):
        g = genomes[k % 3]
        at = int(rng.integers(0, len(g) - 150))
        reads.append(synth.mutate(rng, g[at:at + 150], 0.02, 0.003)[:150])
    reads += [synth.random_bases(rng, 150) for _ in range(500)] + [np.frombuffer(b"A" * 150, dtype=np.uint8).copy()]
    rb, gb = synth.to_bytes(reads), synth.to_bytes(genomes)
    exp, ecig, _ = oracle.align_to_database(rb, gb)
    for bits in (None, "20", "26"):
        kept = {}
        for how in ("blocks", "atomics"):
            if how == "atomics":
                monkeypatch.setenv("KSLAM_FILTER_BUILD", "atomics")
            if bits:
                monkeypatch.setenv("KSLAM_FILTER_BITS", bits)
            c = kslam.Context()
            c.set_index(gb)
            got, gcig = c.align_batch(rb)
            _compare_alignments(got, gcig, exp, ecig)
            kept[how] = c.timings()["n_kmers_kept"]
            c.close()
            monkeypatch.delenv("KSLAM_FILTER_BUILD", raising=False)
            monkeypatch.delenv("KSLAM_FILTER_BITS", raising=False)
        assert kept["blocks"] == kept["atomics"] and 0 < kept["blocks"] < 0.5 * len(rb) * 119, (bits, kept)
    c = kslam.Context()
    c.set_index([b"ACGT" * 5, b""])                   # no entry reaches 32 bases: no k-mer, an empty filter
    got, gcig = c.align_batch(rb[:50])
    assert len(got) == 0 and len(gcig) == 0
    c.close()


@pytest.mark.parametrize("copies", [[3, 9, 12, 13], [12, 13, 16, 17]])
def test_group_route_at_its_cap(kslam, oracle, synth, copies):
    """Groups of EXACTLY known size around join.hip's cap of 64 keys.  An entry is G copies of a random 48-base unit (copy
    starts are multiples of 16 = sampled genome k-mer offsets); a read of 95 bases over two units has 4 k-mers at offsets 0 / 16
    / 32 / 48 that each match G - 1 sampled genome positions, a read of 96 bases has 5: raw (read, entry) groups of about 4 G
    and 5 G keys (measured with the oracle's pre-dedupe list: 11 ... 59 and EXACTLY 64 in the first list, which stays on the
    group route; 63, 64, 67, 79, 84 in the second, which sends its chunk the long way).
    Plain reads ride along so that blocks hold groups of every size; everything against the oracle."""
    rng = np.random.default_rng(sum(copies))
    genomes, reads = [synth.random_bases(rng, 6000)], []
    for g in copies:
        unit = synth.random_bases(rng, 48)
        genomes.append(np.concatenate([synth.random_bases(rng, 64), np.tile(unit, g), synth.random_bases(rng, 64)]))
        for L in (95, 96):
            for shift in (0, 48):
                r = np.tile(unit, 3)[shift:shift + L].copy()
                reads.append(r)
                reads.append(synth.revcomp(r))
    for k in range(300):
        at = int(rng.integers(0, 5800))
        reads.append(synth.mutate(rng, genomes[0][at:at + 120], 0.02, 0.002)[:120])
    order = rng.permutation(len(reads))
    rb, gb = synth.to_bytes([reads[i] for i in order]), synth.to_bytes(genomes)
    # the raw (read, entry) group sizes, from the oracle's pre-dedupe list: the cap must lie inside their range
    recs = oracle.sort_kmers(np.concatenate([oracle.extract_kmers(rb, False, 1), oracle.extract_kmers(gb, True, 16)]))
    pre = oracle.scan_overlaps(recs, [len(r) for r in rb])
    sizes = np.unique(pre["read"].astype(np.int64) * 64 + pre["entry"], return_counts=True)[1]
    assert sizes.max() == (64 if max(copies) <= 13 else 84) and (sizes == 64).any()
    c = kslam.Context()
    c.set_index(gb)
    got, gcig = c.align_batch(rb)
    again, acig = c.align_batch(rb)          # (second list: this one runs inside the pause the first one's long group started)
    c.close()
    exp, ecig, _ = oracle.align_to_database(rb, gb)
    assert len(exp) > 300 + 4 * sum(copies)
    _compare_alignments(got, gcig, exp, ecig)
    _compare_alignments(again, acig, exp, ecig)


@pytest.mark.parametrize("n_entries,passes", [(1, 9), (9, 9), (128, 9), (129, 10), (300, 10), (40000, 11)])
def test_index_build_stats_and_the_passes_of_the_one_time_sort(kslam, synth, n_entries, passes):
    """kslam_index_build_stats (roofline.index_sort of the bench line): the one-time sort of the genome k-mer records makes 8
    passes over the k-mer and, over the meta word, what can differ in a list of genome records (src/KMer.h:65-67, :388-398):
    a pass per whole byte of the ids while more than 7 id bits remain, then ONE pass over the remaining id bits with the
    revComp bit on top (round 6; a pass per byte until then) -- 9 passes up to 128 entries, 10 up to 32 768, 11 beyond -- and
    the index it leaves behind aligns like before (the rows of a batch against the oracle's)."""
    rng = np.random.default_rng(n_entries)
    genomes = [synth.random_bases(rng, int(rng.integers(400, 900)) if n_entries < 1000 else int(rng.integers(150, 300))) for _ in range(n_entries)]
    reads = []
    for k in range(200):
        g = genomes[int(rng.integers(0, n_entries))]
        at = int(rng.integers(0, len(g) - 120))
        r = synth.mutate(rng, g[at:at + 120], 0.02, 0.003)
        reads.append(synth.revcomp(r) if k & 1 else r)
    rb, gb = synth.to_bytes(reads), synth.to_bytes(genomes)
    c = kslam.Context()
    with pytest.raises(kslam.KslamError):
        c.index_build_stats()                                   # no index yet
    c.set_index(gb)
    st = c.index_build_stats()
    assert st["n_entries"] == n_entries and st["sort_passes"] == passes
    assert st["n_genome_kmers"] == sum((len(g) - 32) // 16 + 1 for g in gb)
    assert st["ms_sort"] > 0 and st["ms_total"] >= st["ms_sort"]
    got, gcig = c.align_batch(rb)
    assert c.timings()["n_genome_kmers"] == st["n_genome_kmers"]
    c.close()
    exp, ecig, _ = oracle_align(rb, gb)
    _compare_alignments(got, gcig, exp, ecig)


@pytest.mark.parametrize("scoring", [None, (1, 3, 5, 2), (1, 4, 6, 1), (3, 2, 4, 1), (2, 6, 5, 2), (5, 4, 6, 3)])
def test_one_mismatch_closed_form_equals_the_dp(kslam, oracle, synth, scoring):
    """k_sw_plan's closed form for candidates with exactly ONE mismatch on the seed diagonal (sw.hip: no DP when the counts on
    the four neighbouring diagonals, the window length and the gap cost rule everything else out).  Reads that are a genome
    window with one substitution at EVERY row in turn (rows 0, 1 and L - 2, L - 1 trim the alignment), both strands, lengths
    40-160; the same inside tandem repeats of period 1 / 2 / 3, where a neighbouring diagonal matches as well and the form
    must stand back; reads over a genome end (window shorter than the read); a read with its mismatch next to an N.  Scorings
    where the whole read ties with one side of the mismatch (match 1, mismatch 3: rows L - 4 and 3) or a gap is nearly free.
    Rows, ends, begins and CIGARs must be the oracle's, which runs the reference's DP."""
    rng = np.random.default_rng(31 + (scoring[0] * 7 + scoring[1] if scoring else 0))
    uniq = synth.random_bases(rng, 6000)
    reps = []
    for period in (1, 2, 3):
        unit = synth.random_bases(rng, period)
        reps.append(np.concatenate([synth.random_bases(rng, 60), np.resize(unit, 400), synth.random_bases(rng, 60)]))
    genomes = [uniq] + reps
    comp = {ord("A"): ord("C"), ord("C"): ord("G"), ord("G"): ord("T"), ord("T"): ord("A")}
    reads = []
    for L in (40, 97, 150, 160):
        at = int(rng.integers(100, 5000))
        for x in list(range(L)) if L in (40, 150) else [0, 1, 2, 3, 4, L // 2, L - 5, L - 4, L - 3, L - 2, L - 1]:
            r = uniq[at:at + L].copy()
            r[x] = comp[int(r[x])]
            reads.append(r if (x + L) % 3 else synth.revcomp(r))
    for g in reps:                                         # seeds in the unique flank, the mismatch inside the repeat
        for x in (50, 70, 100, 140):
            for start in (20, 30, 45):
                r = g[start:start + 150].copy()
                r[x] = comp[int(r[x])]
                reads.append(r if x % 20 else synth.revcomp(r))
    for cut in (10, 40):                                   # over the genome's ends: the window is shorter than the read
        r = np.concatenate([uniq[len(uniq) - 150 + cut:], synth.random_bases(rng, cut)])
        r[70] = comp[int(r[70])]
        reads.append(r)
        r = np.concatenate([synth.random_bases(rng, cut), uniq[:150 - cut]])
        r[90] = comp[int(r[90])]
        reads.append(synth.revcomp(r))
    r = uniq[2000:2150].copy()
    r[60], r[61] = comp[int(r[60])], ord("N")
    reads.append(r)
    rb, gb = synth.to_bytes(reads), synth.to_bytes(genomes)
    kw, p = {}, oracle.Params.default()
    if scoring:
        kw = dict(match=scoring[0], mismatch=scoring[1], gap_open=scoring[2], gap_extend=scoring[3])
        p = oracle.Params.default(**kw)
    got, gcig = kslam.align_to_database(rb, gb, **kw)
    exp, ecig, _ = oracle.align_to_database(rb, gb, p)
    assert len(exp) >= len(reads) - 4
    one = (exp["entry"] == 0) & (exp["cigar_len"] == 1)
    assert one.sum() > 150                                  # the form's own territory: ungapped rows of the unique genome
    _compare_alignments(got, gcig, exp, ecig)


def oracle_align(rb, gb):
    import oracle as O
    return O.align_to_database(rb, gb)


def _low_complexity_dataset(synth, seed, n_genomes, n_reads, read_len):
    """Genomes that alternate unique stretches (seeds) with tandem repeats of period 1..6 (where equal
    best scores, equal-cost gap placements and off-diagonal optima are the rule), reads sampled from
    them on both strands with substitutions, Ns and indels; some reads run over a genome's end."""
    rng = np.random.default_rng(seed)
    genomes = []
    for _ in range(n_genomes):
        parts = []
        while sum(len(x) for x in parts) < 3000:
            parts.append(synth.random_bases(rng, int(rng.integers(40, 90))))
            unit = synth.random_bases(rng, int(rng.integers(1, 7)))
            rep = np.resize(unit, int(rng.integers(20, 120)))
            parts.append(synth.mutate(rng, rep, 0.02, 0.004))
        genomes.append(np.concatenate(parts))
    genomes.append(synth.mutate(rng, genomes[0], 0.02, 0.003, 4))   # a strain: near-identical candidates
    reads = []
    for _ in range(n_reads):
        g = genomes[int(rng.integers(0, len(genomes)))]
        L = int(rng.integers(read_len // 2, read_len + 1))
        p = int(rng.integers(0, len(g) - L // 2))
        r = synth.mutate(rng, g[p:p + L], 0.02, 0.01, 4)
        if rng.random() < 0.2 and len(r) > 3:
            r[rng.integers(0, len(r), int(rng.integers(1, 3)))] = ord("N")
        reads.append(synth.revcomp(r) if rng.random() < 0.5 else r)
    return synth.to_bytes(reads), synth.to_bytes(genomes)


@pytest.mark.parametrize("read_len,scoring", [(150, None), (100, None), (250, None), (150, (1, 4, 6, 1)), (150, (3, 2, 4, 1))])
def test_low_complexity_parity(kslam, oracle, synth, read_len, scoring):
    reads, genomes = _low_complexity_dataset(synth, 900 + read_len + (scoring[0] if scoring else 0), 5, 2500, read_len)
    kw = {}
    p = oracle.Params.default()
    if scoring:
        kw = dict(match=scoring[0], mismatch=scoring[1], gap_open=scoring[2], gap_extend=scoring[3])
        p = oracle.Params.default(**kw)
    got, gcig = kslam.align_to_database(reads, genomes, **kw)
    exp, ecig, _ = oracle.align_to_database(reads, genomes, p)
    assert len(exp) > 2000 and (exp["cigar_len"] > 1).sum() > 300
    _compare_alignments(got, gcig, exp, ecig)


def test_longest_supported_reads_and_the_limit(kslam, oracle, synth):
    """511 bases is the longest read the packed kernels take (9-bit row / column keys in the packed DP
    values): such reads, mixed with short ones, must match the oracle; 512 goes to the long-read kernels
    (test_reads_beyond_the_packed_kernels) and more than 9000 bases must be refused, loudly."""
    genomes = synth.make_genomes(301, 2, 2, 30000, strain_sub=0.02, strain_indel=0.002)
    reads, _ = synth.make_paired_reads(302, genomes, 300, read_len=511, frag_mean=900, sub_rate=0.02,
                                       indel_rate=0.003, n_rate=0.002, edge_frac=0.1)
    short, _ = synth.make_paired_reads(303, genomes, 200, read_len=90, frag_mean=250, sub_rate=0.02, indel_rate=0.004)
    rb, gb = synth.to_bytes(reads) + synth.to_bytes(short), synth.to_bytes(genomes)
    assert max(len(r) for r in rb) == 511
    got, gcig = kslam.align_to_database(rb, gb)
    exp, ecig, _ = oracle.align_to_database(rb, gb)
    assert len(exp) > 800 and (exp["cigar_len"] > 1).sum() > 100
    _compare_alignments(got, gcig, exp, ecig)
    one_more = synth.to_bytes([synth.mutate(np.random.default_rng(5), genomes[0][1000:1540], 0.02, 0.003)[:512]])
    assert len(one_more[0]) == 512
    got, gcig = kslam.align_to_database(rb[:50] + one_more, gb)
    exp, ecig, _ = oracle.align_to_database(rb[:50] + one_more, gb)
    assert (exp["read"] == 50).any()
    _compare_alignments(got, gcig, exp, ecig)
    with pytest.raises(kslam.KslamError, match="9000"):
        kslam.align_to_database([b"ACGT" * 2251], gb)


@pytest.mark.parametrize("scoring", [(16, 10, 12, 4), (8, 6, 9, 2)])
def test_scores_at_the_top_of_the_score_field(kslam, oracle, synth, scoring):
    """Perfect 511-base reads under a large match score: 16 x 511 = 8176 is the largest score the packed
    DP values (and the v_max_f64 pairs that keep the running best) can hold; with (16, ., ., 4) the band
    kernels' offset range is exceeded and everything runs on the full-matrix kernel, with (8, ., ., 2)
    the bands run near the top of theirs.  One more match point sends the batch to the long-read kernels
    (int32 scores), with the same rows."""
    genomes = synth.make_genomes(311, 2, 2, 20000, strain_sub=0.01, strain_indel=0.001)
    reads, _ = synth.make_paired_reads(312, genomes, 150, read_len=511, frag_mean=900, sub_rate=0.0, indel_rate=0.0)
    noisy, _ = synth.make_paired_reads(313, genomes, 150, read_len=511, frag_mean=900, sub_rate=0.02, indel_rate=0.003)
    rb, gb = synth.to_bytes(reads) + synth.to_bytes(noisy), synth.to_bytes(genomes)
    kw = dict(match=scoring[0], mismatch=scoring[1], gap_open=scoring[2], gap_extend=scoring[3])
    got, gcig = kslam.align_to_database(rb, gb, **kw)
    exp, ecig, _ = oracle.align_to_database(rb, gb, oracle.Params.default(**kw))
    assert int(exp["score"].max()) == scoring[0] * 511
    _compare_alignments(got, gcig, exp, ecig)
    kw = dict(match=scoring[0] + 1, mismatch=scoring[1], gap_open=scoring[2], gap_extend=scoring[3])
    got, gcig = kslam.align_to_database(rb[:100], gb, **kw)
    exp, ecig, _ = oracle.align_to_database(rb[:100], gb, oracle.Params.default(**kw))
    assert int(exp["score"].max()) == (scoring[0] + 1) * 511
    _compare_alignments(got, gcig, exp, ecig)


@pytest.mark.parametrize("long_len,layout", [(600, "mixed"), (2000, "mixed"), (700, "all"), (4000, "few")])
def test_reads_beyond_the_packed_kernels(kslam, oracle, synth, long_len, layout):
    """ssw_align takes any read length (src/ssw.c:841-951); the packed SW kernels hold 511 bases.  A batch with longer
    reads -- merged pairs, the odd long read -- is split into runs of short and long reads; the long runs go through
    the unfiltered extraction, k_sw_long (plain two-pass SW, one wavefront per candidate) and the literal banded_sw kernel.
    Rows and CIGARs must be the oracle's for every read of the batch, short and long, in batch order."""
    rng = np.random.default_rng(4000 + long_len)
    genomes = synth.make_genomes(301, 3, 3, 40000, strain_sub=0.02, strain_indel=0.002)
    short, _ = synth.make_paired_reads(302, genomes, 400, read_len=150, sub_rate=0.015, indel_rate=0.003, edge_frac=0.05)
    def long_read(L):
        g = genomes[int(rng.integers(0, len(genomes)))]
        at = int(rng.integers(-200, len(g) - L + 200))             # some hang over either end of the genome
        lo, hi = max(at, 0), min(at + L, len(g))
        frag = np.concatenate([synth.random_bases(rng, lo - at), g[lo:hi], synth.random_bases(rng, at + L - hi)])
        if rng.random() < 0.5:
            frag = synth.revcomp(frag)
        r = synth.mutate(rng, frag, 0.02, 0.004)
        return r[:L + int(rng.integers(-30, 1))]
    n_long = {"mixed": 40, "all": 60, "few": 6}[layout]
    longs = [long_read(long_len) for _ in range(n_long)]
    if layout == "all":
        reads = longs
    else:
        reads = list(short)
        for k, r in enumerate(longs):                              # scattered: runs of one, and one run of several
            reads.insert(int(rng.integers(0, len(reads))) if k % 4 else 17, r)
    rb, gb = synth.to_bytes(reads), synth.to_bytes(genomes)
    assert max(len(r) for r in rb) > 511
    c = kslam.Context()
    c.set_index(gb)
    got, gcig = c.align_batch(rb)
    again, acig = c.align_batch(rb)
    c.close()
    exp, ecig, _ = oracle.align_to_database(rb, gb)
    _compare_alignments(got, gcig, exp, ecig)
    assert got.tobytes() == again.tobytes() and gcig.tobytes() == acig.tobytes()
    lens = np.array([len(r) for r in rb])
    on_long = lens[got["read"]] > 511
    assert on_long.sum() > 2 * n_long and (got["cigar_len"][on_long] > 1).sum() > n_long // 2
    assert got["score"][on_long].max() > 1022                      # beyond what 511 bases can score
