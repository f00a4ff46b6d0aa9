"""CPU tests: the oracle against the golden vectors and (when oracle/_ref exists) the real
reference pieces; structural expectations of the reference's own src/Tests.h."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _split(flat, lens):
    out, p = [], 0
    for n in lens:
        out.append(flat[p:p + int(n)])
        p += int(n)
    return out


def test_survey_vectors(oracle):
    sv = json.load(open(os.path.join(GOLD, "survey_vectors.json")))
    r = oracle.extract_kmers([b""] * sv["read_id"] + [sv["read46"].encode()], False, 1)
    for i, exp in enumerate(sv["read_records"]):
        assert [int(r[i]["kmer"]), int(r[i]["meta"]), int(r[i]["offset"])] == exp
    g = oracle.extract_kmers([b""] * sv["genome_id"] + [sv["read46"].encode()], True, sv["genome_gap"])
    assert [[int(x["meta"]), int(x["offset"])] for x in g] == sv["genome_meta_off"]
    p = oracle.Params.default(match=sv["scoring"][0], mismatch=sv["scoring"][1],
                              gap_open=sv["scoring"][2], gap_extend=sv["scoring"][3])
    for a in sv["align"]:
        for plain in (False, True):
            res, cig = oracle.align(a["query"].encode(), a["ref"].encode()[:a["ref_len"]], p, plain=plain)
            assert res.score1 == a["score"]
            assert oracle.cigar_string(cig) == a["cigar"]
            assert (res.read_begin1, res.read_end1) == (a["q_b"], a["q_e"])
            if "ref_b" in a:
                assert (res.ref_begin1, res.ref_end1) == (a["ref_b"], a["ref_e"])


def test_ssw_golden_vectors(oracle):
    """oracle (striped emulation AND the plain-Gotoh kernel spec) == answers of the real ssw.c"""
    z = np.load(os.path.join(GOLD, "ssw_vectors.npz"))
    reads, refs = _split(z["reads"], z["read_len"]), _split(z["refs"], z["ref_len"])
    cigs = _split(z["cigars"], z["cigar_len"])
    for i in range(len(reads)):
        m, x, go, ge = [int(v) for v in z["params"][i]]
        mat = oracle.build_matrix(m, x)
        for plain in (False, True):
            res, cig = oracle.ssw_align(reads[i], refs[i], mat, go, ge, plain=plain)
            got = (res.score1, res.ref_begin1, res.ref_end1, res.read_begin1, res.read_end1)
            assert got == tuple(int(v) for v in z["results"][i]), (i, plain)
            assert np.array_equal(cig, cigs[i]), (i, plain)


def test_kmer_golden_vectors(oracle):
    z = np.load(os.path.join(GOLD, "kmer_vectors.npz"))
    seqs = [s.tobytes() for s in _split(z["seqs"], z["seq_len"])]
    rr = oracle.extract_kmers(seqs, False, 1)
    rg = oracle.extract_kmers(seqs, True, 16)
    assert (rr == z["reads_gap1"]).all() and (rg == z["genbank_gap16"]).all()
    srt = oracle.sort_kmers(np.concatenate([rr, rg]))
    assert (srt["kmer"] == z["sorted_kmer"]).all() and (srt["meta"] == z["sorted_meta"]).all()


def test_align_small_golden(oracle):
    z = np.load(os.path.join(GOLD, "align_small.npz"))
    reads = [s.tobytes() for s in _split(z["reads"], z["read_len"])]
    genomes = [s.tobytes() for s in _split(z["genomes"], z["genome_len"])]
    for plain in (False, True):
        al, cg, _ = oracle.align_to_database(reads, genomes, plain=plain)
        assert (al == z["alignments"]).all() and np.array_equal(cg, z["cigars"])


def _random_pair(rng):
    L = int(rng.integers(20, 256))
    ref = rng.integers(0, 4, L).astype(np.int8)
    rd = ref.copy()
    k = int(rng.integers(0, max(1, L // 8)))
    rd[rng.integers(0, L, k)] = rng.integers(0, 4, k)
    for _ in range(int(rng.integers(0, 3))):
        p = int(rng.integers(1, len(rd) - 1)); n = int(rng.integers(1, 4))
        rd = np.concatenate([rd[:p], rd[p + n:]]) if rng.random() < 0.5 else \
            np.concatenate([rd[:p], rng.integers(0, 4, n).astype(np.int8), rd[p:]])
    if rng.random() < 0.3:
        rd = np.concatenate([rng.integers(0, 4, rng.integers(1, 20)).astype(np.int8), rd])
    if rng.random() < 0.3:
        rd = np.concatenate([rd, rng.integers(0, 4, rng.integers(1, 20)).astype(np.int8)])
    if rng.random() < 0.2:
        rd[rng.integers(0, len(rd))] = 4
    if rng.random() < 0.2:
        ref[rng.integers(0, len(ref))] = 4
    if rng.random() < 0.2:
        ref = ref[:int(rng.integers(L // 2, L + 1))]
    return rd[:300], ref


@pytest.mark.parametrize("params,n", [((2, 3, 5, 2), 3000), ((1, 4, 6, 1), 800), ((2, 6, 5, 1), 800),
                                      ((3, 2, 4, 3), 800)])
def test_plain_spec_equals_striped(oracle, params, n):
    """The plain-Gotoh spec the HIP kernel implements == the striped emulation, inside the
    scoring envelope kslam_create accepts (gapE < gapO, mismatch <= gapO + gapE)."""
    rng = np.random.default_rng(sum(params))
    mat = oracle.build_matrix(params[0], params[1])
    for i in range(n):
        rd, ref = _random_pair(rng)
        a, ca = oracle.ssw_align(rd, ref, mat, params[2], params[3])
        b, cb = oracle.ssw_align(rd, ref, mat, params[2], params[3], plain=True)
        assert (a.score1, a.ref_begin1, a.ref_end1, a.read_begin1, a.read_end1) == \
               (b.score1, b.ref_begin1, b.ref_end1, b.read_begin1, b.read_end1), i
        assert np.array_equal(ca, cb), i


def test_against_real_ssw_when_present(oracle):
    if not oracle.have_ref_ssw():
        pytest.skip("oracle/_ref/libssw_ref.so not built (no /root/reference)")
    rng = np.random.default_rng(99)
    for params in ((2, 3, 5, 2), (2, 9, 5, 2), (5, 4, 10, 10), (2, 8, 2, 3)):  # incl. sets outside the envelope
        mat = oracle.build_matrix(params[0], params[1])
        for i in range(500):
            rd, ref = _random_pair(rng)
            r0, c0 = oracle.ref_ssw_align(rd, ref, mat, params[2], params[3])
            r1, c1 = oracle.ssw_align(rd, ref, mat, params[2], params[3])
            assert (r1.score1, r1.ref_begin1, r1.ref_end1, r1.read_begin1, r1.read_end1) == r0, (params, i)
            assert np.array_equal(c0, c1), (params, i)


def test_against_real_kmer_code_when_present(oracle):
    if not oracle.have_ref_kmer():
        pytest.skip("oracle/_ref/libkmer_ref.so not built (no /root/reference)")
    rng = np.random.default_rng(5)
    B = np.frombuffer(b"ACGTN", dtype=np.uint8)
    seqs = [B[rng.choice(5, int(rng.integers(0, 500)), p=[.245, .245, .245, .245, .02])].tobytes() for _ in range(200)]
    for is_gb, gap in ((False, 1), (True, 16), (True, 5)):
        assert (oracle.extract_kmers(seqs, is_gb, gap) == oracle.ref_extract_kmers(seqs, is_gb, gap)).all()
    recs = np.concatenate([oracle.extract_kmers(seqs, False, 1), oracle.extract_kmers(seqs, True, 16)])
    a, b = oracle.sort_kmers(recs), oracle.ref_sort_kmers(recs)
    assert (a["kmer"] == b["kmer"]).all() and (a["meta"] == b["meta"]).all()
    assert oracle.ref_kmer3(b"TAG") == (35, 24)  # src/KMer.h:27


def test_encoding_order_and_canonical_choice(oracle):
    """src/Tests.h:334-452: packed k-mers sort in A<C<T<G order; canonical = numerically smaller,
    palindromes take the rc branch (src/KMer.h:173)."""
    r = oracle.extract_kmers([b"A" * 32, b"C" * 32, b"T" * 32, b"G" * 32], False, 1)
    # A*32 / C*32: rc is T*32 / G*32 (larger) so forward; T*32 -> rc A*32 = 0 so rc; G -> rc C
    assert [int(x["kmer"]) for x in r] == [0, 0x5555555555555555, 0, 0x5555555555555555]
    assert [int(x["meta"]) >> 30 for x in r] == [0, 0, 1, 1]
    pal = b"ACGT" * 8  # reverse complement of itself
    p = oracle.extract_kmers([pal], False, 1)
    assert int(p[0]["meta"]) >> 30 == 1 and int(p[0]["offset"]) == 0


def test_dedupe_ladder(oracle):
    """src/Overlap.h:79-85,290: unique compares with the LAST KEPT element: 0,2,4,5,8 -> 0,4,8."""
    K = oracle.KMER_DT
    rels = [0, 2, 4, 5, 8]
    recs = []
    for i, rel in enumerate(rels):
        km = 1000 + i
        recs.append((km, (1 << 31) | 3, 100 + rel))  # genome entry 3 at offset 100+rel
        recs.append((km, 7, 100))                    # read 7 at offset 100 -> rel
    a = np.array(recs, dtype=K)
    ov, raw = oracle.find_overlaps(oracle.sort_kmers(a), [200] * 8)
    assert raw == 5
    assert [int(x) for x in ov["rel"]] == [0, 4, 8]
    assert set(int(x) for x in ov["read"]) == {7} and set(int(x) for x in ov["entry"]) == {3}


def test_planted_overlaps_and_scores(oracle, synth):
    """src/Tests.h:161-333: planted (entry, rel, revComp) recovered; SW score = 2 x overlap length."""
    rng = np.random.default_rng(2)
    genomes = [synth.random_bases(rng, 3000) for _ in range(40)]
    reads, truth = [], []
    for i in range(300):
        g = int(rng.integers(0, 40)); L = 100
        off = int(rng.integers(-30, 3000 - 70))
        lo, hi = max(off, 0), min(off + L, 3000)
        s = np.concatenate([synth.random_bases(rng, lo - off), genomes[g][lo:hi], synth.random_bases(rng, off + L - hi)])
        rc = bool(rng.random() < 0.5)
        reads.append((synth.revcomp(s) if rc else s).tobytes())
        truth.append((g, off, rc, hi - lo))
    al, cg, _ = oracle.align_to_database(reads, [g.tobytes() for g in genomes])
    for i, (g, off, rc, ovl) in enumerate(truth):
        m = al[(al["read"] == i) & (al["entry"] == g)]
        hit = [x for x in m if abs(int(x["rel"]) - off) < 3 and bool(x["revcomp"]) == rc]
        assert hit and int(hit[0]["score"]) == 2 * ovl, i


def test_kmer_zero_never_joins(oracle):
    """src/Overlap.h:236-239: runs with kMerInt == 0 (poly-A / poly-T / all-N) are skipped."""
    al, _, _ = oracle.align_to_database([b"A" * 100, b"N" * 100, b"T" * 100], [b"A" * 500])
    assert len(al) == 0


# ---- pins against the reference's own join / dedupe / aligner / SW driver (oracle/_ref/libjoin_ref.so) ----
def _sorted4(a):
    return np.sort(a[["read", "entry", "rel", "revcomp"]], order=["read", "entry", "rel", "revcomp"])


def _same_modulo_revcomp_ties(got, exp, ties, fields):
    assert len(got) == len(exp)
    amb = np.array([(int(r), int(e), int(l)) in ties for r, e, l in zip(exp["read"], exp["entry"], exp["rel"])], dtype=bool)
    for f in fields:
        ok = got[f] == exp[f]
        if f == "revcomp":
            ok = ok | amb
        assert ok.all(), (f, np.flatnonzero(~ok)[:5])
    return int(amb.sum())


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_against_real_join_when_present(oracle, seed):
    """orc_find_overlaps == the reference's own findOverlaps + findOverlaps_parallel (src/Overlap.h:153-295), compiled in
    place, on pile-ups with several genome records, rc/fwd mixes, k-mer 0, genome ends and the revComp tie."""
    if not oracle.have_ref_join():
        pytest.skip("oracle/_ref/libjoin_ref.so not built (no /root/reference)")
    from join_cases import make_join_case, revcomp_tie_rows
    reads, genomes = make_join_case(seed)
    recs = np.concatenate([oracle.extract_kmers(reads, False, 1), oracle.extract_kmers(genomes, True, 16)])
    srt = oracle.ref_sort_kmers(recs)                       # the reference's own order, ties and all
    lens = [len(r) for r in reads]
    ref_ov, ref_raw = oracle.ref_find_overlaps(srt, lens, want_raw=True)
    raw = oracle.scan_overlaps(srt, lens)
    assert len(raw) == len(ref_raw) > 2000
    assert (raw == ref_raw).all()                           # same list in the same emission order
    ov, nraw = oracle.find_overlaps(srt, lens)
    assert nraw == len(ref_raw)
    ties = revcomp_tie_rows(ref_raw)
    assert ties, "the case is meant to hold a revComp tie"
    n_amb = _same_modulo_revcomp_ties(ov, ref_ov, ties, ("read", "entry", "rel", "revcomp"))
    assert 0 < n_amb < 20
    # the oracle's own sort (offset asc as the last key) instead of the reference's unstable one: same multiset, same dedupe
    ov2, _ = oracle.find_overlaps(oracle.sort_kmers(recs), lens)
    assert (ov2 == ov).all()
    assert (_sorted4(oracle.scan_overlaps(oracle.sort_kmers(recs), lens)) == _sorted4(ref_raw)).all()
    # multi-threaded reference run: chunked findOverlaps + parallel sort give the same list modulo the same ties
    ref_mt, _ = oracle.ref_find_overlaps(srt, lens, one_thread=False)
    _same_modulo_revcomp_ties(ref_mt, ref_ov, ties, ("read", "entry", "rel", "revcomp"))


def test_against_real_dedupe_ladder_when_present(oracle):
    """sort + unique alone (src/Overlap.h:289-291): 0,2,4,5,8 -> 0,4,8; comparison is with the LAST KEPT element."""
    if not oracle.have_ref_join():
        pytest.skip("oracle/_ref/libjoin_ref.so not built (no /root/reference)")
    rng = np.random.default_rng(3)
    rows = [(7, 3, r, 0) for r in (8, 5, 4, 2, 0)] + [(7, 4, r, 1) for r in (-3, -1, 1, 2, 100, 102, 105)]
    rows += [(int(rng.integers(0, 5)), int(rng.integers(0, 3)), int(rng.integers(-20, 20)), 0) for _ in range(400)]
    a = np.zeros(len(rows), dtype=oracle.OVERLAP_DT)
    for i, (r, e, l, c) in enumerate(rows):
        a[i]["read"], a[i]["entry"], a[i]["rel"], a[i]["revcomp"] = r, e, l, c
    got = oracle.ref_sort_unique_overlaps(a)
    lad = got[(got["read"] == 7) & (got["entry"] == 3)]
    assert [int(x) for x in lad["rel"]] == [0, 4, 8]
    # the restatement's unique on the same input
    K = oracle.KMER_DT
    b = np.sort(a, order=["read", "entry", "rel"])
    kept = [0]
    for i in range(1, len(b)):
        x, y = b[kept[-1]], b[i]
        if not (x["read"] == y["read"] and x["entry"] == y["entry"] and abs(int(x["rel"]) - int(y["rel"])) < 3):
            kept.append(i)
    assert (b[kept][["read", "entry", "rel"]] == got[["read", "entry", "rel"]]).all()


@pytest.mark.parametrize("params", [(2, 3, 5, 2), (1, 4, 6, 1), (3, 2, 4, 3)])
def test_against_real_aligner_when_present(oracle, params):
    """orc_align == the reference's own Aligner::Align (src/ssw_cpp.cpp:234-283: kBaseTranslation, BuildSwScoreMatrix,
    SetFlag, ConvertAlignment) on ASCII with lower case / U / IUPAC, ref_len < strlen, thresholded and disabled CIGAR."""
    if not oracle.have_ref_join():
        pytest.skip("oracle/_ref/libjoin_ref.so not built (no /root/reference)")
    from join_cases import make_align_cases
    cases = make_align_cases(sum(params), 300)
    for thr, want_cigar in ((0, True), (120, True), (0, False)):
        p = oracle.Params.default(report_cigar=want_cigar, score_threshold=thr, match=params[0], mismatch=params[1],
                                  gap_open=params[2], gap_extend=params[3])
        n_cig = n_nocig = 0
        for i, (q, r, n) in enumerate(cases):
            exp, ecig = oracle.ref_aligner_align(q, r, p, ref_len=n)
            for plain in (False, True):
                res, cig = oracle.align(q, r[:n], p, plain=plain)
                got = (res.score1, res.ref_begin1, res.ref_end1, res.read_begin1, res.read_end1)
                assert got == exp, (i, thr, want_cigar, plain)
                assert np.array_equal(cig, ecig), (i, thr, want_cigar, plain)
            n_cig += len(ecig) > 0
            n_nocig += len(ecig) == 0
        assert (n_cig > 0) == want_cigar and (n_nocig > 0) == (thr > 0 or not want_cigar)
    # every byte of kBaseTranslation's domain (src/ssw_cpp.cpp:10: 128 entries; a byte >= 128 indexes outside the table in
    # the reference -- undefined, nothing to pin) but NUL, as a query / a reference: the translation table and the 5x5 matrix
    p = oracle.Params.default(match=params[0], mismatch=params[1], gap_open=params[2], gap_extend=params[3])
    for c in range(1, 128):
        for d in b"ACGTN":
            q, r = bytes([c]) * 3, bytes([d]) * 3
            exp, _ = oracle.ref_aligner_align(q, r, p)
            res, _ = oracle.align(q, r, p)
            assert res.score1 == exp[0], (c, d)


@pytest.mark.parametrize("seed", [21, 22])
def test_against_real_align_to_database_when_present(oracle, seed):
    """orc_align_to_database == the reference's own alignToDatabase (src/SLAM.h:59-79: getKMersFromReads, getKMers,
    sortKMers, findOverlaps_parallel, performSmithWatermanOnRange_parallel), and orc_sw_on_overlap == the reference's
    performSmithWatermanOnRange2 (src/SmithWaterman.h:184-233) on the same overlap list."""
    if not oracle.have_ref_join():
        pytest.skip("oracle/_ref/libjoin_ref.so not built (no /root/reference)")
    from join_cases import make_join_case, revcomp_tie_rows
    reads, genomes = make_join_case(seed, n_reads=160)
    recs = np.concatenate([oracle.extract_kmers(reads, False, 1), oracle.extract_kmers(genomes, True, 16)])
    lens = [len(r) for r in reads]
    ties = revcomp_tie_rows(oracle.scan_overlaps(oracle.sort_kmers(recs), lens))
    fields = ("read", "entry", "rel", "revcomp", "score", "ref_begin", "ref_end", "query_begin", "query_end", "cigar_len")
    for thr in (0, 150):
        p = oracle.Params.default(score_threshold=thr)
        exp, ecig = oracle.ref_align_to_database(reads, genomes, p)
        assert len(exp) > 400
        for plain in (False, True):
            got, gcig, _ = oracle.align_to_database(reads, genomes, p, plain=plain)
            _same_modulo_revcomp_ties(got, exp, ties, fields)
            assert np.array_equal(gcig, ecig)
        assert ((exp["cigar_len"] > 0) == (exp["score"] >= thr)).all()
        # the SW driver alone, on the reference's own deduped list
        ov = np.zeros(len(exp), dtype=oracle.OVERLAP_DT)
        for f in ("read", "entry", "rel", "revcomp"):
            ov[f] = exp[f]
        sw, swcig = oracle.ref_sw_on_overlaps(ov, reads, genomes, p)
        assert (sw == exp).all() and np.array_equal(swcig, ecig)
    # multi-threaded reference run == one-thread run modulo the revComp ties
    mt, _ = oracle.ref_align_to_database(reads, genomes, oracle.Params.default(), one_thread=False)
    one, _ = oracle.ref_align_to_database(reads, genomes, oracle.Params.default())
    _same_modulo_revcomp_ties(mt, one, ties, fields)


def test_the_real_align_to_database_template_and_the_full_size_comparison(oracle):
    """oracle/_ref/libslam_ref.so::ref_slam_align_to_database calls the reference's own `alignToDatabase` template
    (src/SLAM.h:59-79, compiled from the header) -- what bench.py times as cpu_baseline.kind "reference" and what the
    full-size -m gpu checker runs.  It equals libjoin_ref.so's statement of the same five calls; and
    compare_with_reference_rows (the comparison both use) accepts exactly the documented revComp ties and nothing else."""
    if not (oracle.have_ref_slam() and oracle.have_ref_join()):
        pytest.skip("oracle/_ref not built (no /root/reference)")
    from join_cases import make_join_case
    reads, genomes = make_join_case(12, n_reads=160)
    oracle.ref_slam_set_index([{"bases": g} for g in genomes])
    flat = np.frombuffer(b"".join(reads), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    one, ocig, sec, _ = oracle.ref_slam_align_to_database(flat, offs, threads=1)
    exp, ecig = oracle.ref_align_to_database(reads, genomes)
    assert len(one) > 400 and (one == exp).all() and np.array_equal(ocig, ecig) and sec > 0
    rb, eb = (lambda i: reads[i]), (lambda j: genomes[j])
    # the restatement against a multi-threaded run of the real thing
    mt, mcig, _, _ = oracle.ref_slam_align_to_database(flat, offs, threads=4)
    got, gcig, _ = oracle.align_to_database(reads, genomes)
    v = oracle.compare_with_reference_rows(got, gcig, mt, mcig, rb, eb)
    assert v["identical"] and v["differing_rows_are_revcomp_ties"], v
    # a planted tie decided the other way is accepted ...
    from join_cases import revcomp_tie_rows
    recs = np.concatenate([oracle.extract_kmers(reads, False, 1), oracle.extract_kmers(genomes, True, 16)])
    ties = revcomp_tie_rows(oracle.scan_overlaps(oracle.sort_kmers(recs), [len(r) for r in reads]))
    assert ties
    flip = got.copy()
    i = [k for k in range(len(flip)) if (int(flip["read"][k]), int(flip["entry"][k]), int(flip["rel"][k])) in ties][0]
    flip["revcomp"][i] = one["revcomp"][i] ^ 1
    v = oracle.compare_with_reference_rows(flip, gcig, one, ocig, rb, eb)
    assert v["identical"] and v["rows_differing"] >= 1, v
    base = oracle.compare_with_reference_rows(got, gcig, one, ocig, rb, eb)
    assert base["identical"], base
    # ... a flipped flag on a row that is no tie, a changed score, a changed CIGAR word are not
    j = [k for k in range(len(got)) if (int(got["read"][k]), int(got["entry"][k]), int(got["rel"][k])) not in ties][5]
    for field, delta in (("revcomp", 1), ("score", 1), ("ref_end", 1)):
        wrong = got.copy()
        wrong[field][j] ^= delta
        v = oracle.compare_with_reference_rows(wrong, gcig, one, ocig, rb, eb)
        assert not v["identical"] and v["rows_differing"] == base["rows_differing"] + 1, (field, v)
    wcig = gcig.copy()
    wcig[0] ^= 16
    v = oracle.compare_with_reference_rows(got, wcig, one, ocig, rb, eb)
    assert not v["identical"] and v["rows_differing"] == base["rows_differing"] + 1, v
    assert not oracle.compare_with_reference_rows(got[:-1], gcig, one, ocig, rb, eb)["identical"]


def test_reference_log_stamps_become_phase_times(oracle, tmp_path):
    """bench.py's cpu_baseline.phases_s: the reference's own log.txt stamps (src/sequenceTools.h:171-179, `[t = 1.23s]\t<text>`) of
    the LAST alignToDatabase call, each phase from its stamp to the next phase's, Smith-Waterman to the end of the call."""
    from oracle.binding import _ref_log_phases
    log = tmp_path / "log.txt"
    log.write_text("[t = 0.00s]\tBuilding taxonomy index\n"
                   "[t = 1.00s]\tAligning reads to database using k = 32\n[t = 1.00s]\tGetting k-mers from reads\n"
                   "[t = 1.50s]\tObtained 5 k-mers\n[t = 1.50s]\tGetting k-mers from index\n[t = 2.00s]\tSorting k-mers\n"
                   "[t = 2.10s]\tFinding overlaps\n[t = 2.20s]\tPerforming pairwise Smith-Waterman\n"
                   "[t = 10.00s]\tAligning reads to database using k = 32\n[t = 10.01s]\tGetting k-mers from reads\n"
                   "[t = 10.35s]\tObtained 238000000 k-mers\n[t = 10.35s]\tGetting k-mers from index\n[t = 12.30s]\tObtained 312498224 k-mers\n"
                   "[t = 12.31s]\tSorting k-mers\n[t = 15.90s]\tFinding overlaps\n[t = 17.10s]\tFound 8153847 k-mer overlaps\n"
                   "[t = 17.15s]\tPerforming pairwise Smith-Waterman\n")
    ph = _ref_log_phases(str(log), 19.5)
    assert ph == {"extract": 0.34, "genome_kmers": 1.96, "sort": 3.59, "join": 1.25, "sw": 12.35}
    assert _ref_log_phases(str(tmp_path / "missing.txt"), 1.0) is None
    (tmp_path / "other.txt").write_text("[t = 0.00s]\tBuilding taxonomy index\n")
    assert _ref_log_phases(str(tmp_path / "other.txt"), 1.0) is None


# ---- the committed answers of the real reference (recorded by tests/golden/make_golden.py --pins): these run anywhere ----
def _cols(z, name):
    flat, off = z[name], z[name + "_off"]
    return [flat[int(off[i]):int(off[i + 1])].tobytes() for i in range(len(off) - 1)]


def test_join_golden_vectors(oracle):
    """oracle == the recorded answers of the reference's own sortKMers / findOverlaps / findOverlaps_parallel /
    alignToDatabase (tests/golden/join_vectors.npz)."""
    z = np.load(os.path.join(GOLD, "join_vectors.npz"))
    reads, genomes = _cols(z, "reads"), _cols(z, "genomes")
    lens = [len(r) for r in reads]
    ties = {tuple(int(v) for v in t) for t in z["ties"]}
    assert len(ties) >= 1
    recs = np.concatenate([oracle.extract_kmers(reads, False, 1), oracle.extract_kmers(genomes, True, 16)])
    srt = oracle.sort_kmers(recs)
    assert (srt["kmer"] == z["sorted"]["kmer"]).all() and (srt["meta"] == z["sorted"]["meta"]).all()
    assert (oracle.scan_overlaps(z["sorted"], lens) == z["raw"]).all()
    assert (_sorted4(oracle.scan_overlaps(srt, lens)) == _sorted4(z["raw"])).all()
    ov, nraw = oracle.find_overlaps(srt, lens)
    assert nraw == len(z["raw"])
    _same_modulo_revcomp_ties(ov, z["deduped"], ties, ("read", "entry", "rel", "revcomp"))
    fields = ("read", "entry", "rel", "revcomp", "score", "ref_begin", "ref_end", "query_begin", "query_end", "cigar_len")
    for key, p in (("thr0", oracle.Params.default()), ("thr150", oracle.Params.default(score_threshold=150))):
        for plain in (False, True):
            got, gcig, _ = oracle.align_to_database(reads, genomes, p, plain=plain)
            _same_modulo_revcomp_ties(got, z["alignments_" + key], ties, fields)
            assert np.array_equal(gcig, z["cigars_" + key])
    got, gcig, _ = oracle.align_to_database(reads, genomes, oracle.Params.default(report_cigar=False))
    _same_modulo_revcomp_ties(got, z["alignments_nocigar"], ties, fields)
    assert len(gcig) == 0


def test_align_golden_vectors(oracle):
    """oracle == the recorded answers of the reference's own Aligner::Align (tests/golden/align_vectors.npz)."""
    z = np.load(os.path.join(GOLD, "align_vectors.npz"))
    assert len(z["param_sets"]) == 5          # two scorings inside the envelope, three outside it
    for params in (tuple(int(v) for v in ps) for ps in z["param_sets"]):
        tag = "p%d%d%d%d" % params
        qs, rs, ns = _cols(z, tag + "_query"), _cols(z, tag + "_ref"), z[tag + "_ref_len"]
        for thr, want in ((0, 1), (120, 1), (0, 0)):
            k = "%s_thr%d_cigar%d" % (tag, thr, want)
            cigs = _split(z[k + "_cigars"], z[k + "_cigar_len"])
            p = oracle.Params.default(report_cigar=bool(want), score_threshold=thr, match=params[0], mismatch=params[1],
                                      gap_open=params[2], gap_extend=params[3])
            # the plain-Gotoh spec equals the striped code only inside the envelope (gapE < gapO, mismatch <= gapO + gapE)
            in_envelope = 1 <= params[3] < params[2] and params[1] <= params[2] + params[3]
            for i in range(len(qs)):
                for plain in ((False, True) if in_envelope else (False,)):
                    res, cig = oracle.align(qs[i], rs[i][:int(ns[i])], p, plain=plain)
                    got = (res.score1, res.ref_begin1, res.ref_end1, res.read_begin1, res.read_end1)
                    assert got == tuple(int(v) for v in z[k + "_results"][i]), (k, i, plain)
                    assert np.array_equal(cig, cigs[i]), (k, i, plain)
