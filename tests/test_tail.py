"""Host tail (include/kslam_tail.h): pairing -> insert-size screen ->
score screen -> pseudo-assembly -> SAM.

The product (k-slam_amd/host/tail.cpp: flat records, threaded) is compared with the serial
object-by-object restatement in oracle/tail_oracle.cpp.  The reference itself cannot be built here
(Boost headers are absent) and its tests hold no vectors for this stage, so the restatement is
pinned by what the output must satisfy on its own:
  * hand-worked pairing cases derived from reference src/PairedOverlap.h:132-242;
  * the SAM records re-create the genome window from read + CIGAR + MD, NM is the edit count,
    flags / mate fields are mutually consistent (SAM definition, reference src/SAM.h:101-433).
These tests are host-only except the last one (hot path on the GPU -> tail).
"""
import importlib
import os
import re

import numpy as np
import pytest


@pytest.fixture(scope="module")
def T(kslam):
    return importlib.import_module("kslam_amd.tail")


def _ov(kslam, rows):
    """rows of (read, entry, rel, revcomp, score, ref_begin, ref_end) -> overlap records, no cigars"""
    a = np.zeros(len(rows), dtype=kslam.OVERLAP_DT)
    for i, r in enumerate(rows):
        a[i]["read"], a[i]["entry"], a[i]["rel"], a[i]["revcomp"], a[i]["score"] = r[:5]
        a[i]["ref_begin"], a[i]["ref_end"] = r[5], r[6]
        a[i]["query_begin"], a[i]["query_end"] = 0, 99
    return a


def _pairs_of(pr):
    return [(int(p["combined_score"]), int(p["entry"]), int(p["ref_start"]), int(p["ref_end"]),
             int(p["insert_size"]), int(p["r1"]), int(p["r2"])) for p in pr]


NO = 0xFFFFFFFF

# (name, overlaps for ONE read pair (reads 0 and 1, 100 bases each), expected alignment pairs)
PAIRING_CASES = [
    ("proper pair: R1 forward then R2 reverse",
     [(0, 0, 100, 0, 200, 100, 199), (1, 0, 300, 1, 190, 300, 399)],
     [(390, 0, 100, 399, 300, 0, 1)]),
    ("same strand never pairs; leftovers flush R2 first (src/PairedOverlap.h:222-241)",
     [(0, 0, 100, 0, 200, 100, 199), (1, 0, 300, 0, 190, 300, 399)],
     [(190, 0, 300, 399, 0, NO, 1), (200, 0, 100, 199, 0, 0, NO)]),
    ("a newer R1 displaces an unused one, which is emitted alone",
     [(0, 0, 100, 0, 200, 100, 199), (0, 0, 150, 0, 180, 150, 249), (1, 0, 320, 1, 190, 320, 419)],
     [(200, 0, 100, 199, 0, 0, NO), (370, 0, 150, 419, 270, 1, 2)]),
    ("tie on (entry, rel): the R1 overlap is taken first",
     [(0, 0, 200, 1, 150, 200, 299), (1, 0, 200, 0, 160, 200, 299)],
     [(310, 0, 200, 299, 100, 0, 1)]),
    ("one R1 pairs with two successive R2 overlaps",
     [(0, 0, 100, 0, 200, 100, 199), (1, 0, 300, 1, 190, 300, 399), (1, 0, 305, 1, 170, 305, 404)],
     [(390, 0, 100, 399, 300, 0, 1), (370, 0, 100, 404, 305, 0, 2)]),
    ("runs are per entry: nothing pairs across entries",
     [(0, 0, 10, 0, 200, 10, 109), (0, 1, 20, 0, 180, 20, 119), (1, 1, 200, 1, 190, 200, 299)],
     [(200, 0, 10, 109, 0, 0, NO), (370, 1, 20, 299, 280, 1, 2)]),
    ("R2 forward first, then R1 reverse: orientation R2....R1 (makePair false)",
     [(0, 0, 400, 1, 200, 400, 499), (1, 0, 150, 0, 190, 150, 249)],
     [(390, 0, 150, 499, 350, 0, 1)]),
]


@pytest.mark.parametrize("name,rows,expected", PAIRING_CASES, ids=[c[0][:40] for c in PAIRING_CASES])
def test_pairing_hand_worked(kslam, oracle, T, name, rows, expected):
    ov = _ov(kslam, rows)
    reads = T.Reads([b"A" * 100, b"C" * 100])
    P = T.TailParams.default(stages=T.STAGE_PAIRING_ONLY, threads=2)
    rp, pr, st = T.tail_pairs(P, reads, ov)
    assert _pairs_of(pr) == expected, name
    assert len(rp) == 1 and (int(rp[0]["r1_read"]), int(rp[0]["r2_read"])) == (0, 1)
    assert int(rp[0]["count"]) == len(expected)
    orp, opr = oracle.tail_pairs(P, reads.view, ov)
    assert _pairs_of(opr) == expected, "restatement: " + name
    assert orp.tobytes() == rp.tobytes()


def test_insert_size_split_and_score_screen(kslam, oracle, T):
    """A pair far beyond the insert-size limit is split into two single alignments
    (src/PairedOverlap.h:396-436), then the 0.95 score fraction keeps only near-best ones."""
    n = 400
    rows1, rows2 = [], []
    for i in range(n):
        far = (i == 7)
        rows1.append((i, 0, 1000 * i, 0, 200, 1000 * i, 1000 * i + 99))
        d = 60000 if far else 200 + (i % 40)
        rows2.append((n + i, 0, 1000 * i + d, 1, 195 if far else 190, 1000 * i + d, 1000 * i + d + 99))
    ov = _ov(kslam, rows1 + rows2)
    reads = T.Reads([b"A" * 100] * (2 * n))
    P = T.TailParams.default(pseudo_assembly=False, threads=3)
    rp, pr, st = T.tail_pairs(P, reads, ov)
    orp, opr = oracle.tail_pairs(P, reads.view, ov)
    assert rp.tobytes() == orp.tobytes() and pr.tobytes() == opr.tobytes()
    assert st.n_insert_sizes == n and 300 < st.max_insert_size < 1000
    g = rp[7]
    mine = _pairs_of(pr[int(g["first"]):int(g["first"]) + int(g["count"])])
    # split: the R2-only half keeps the slot, the R1-only half is appended; both within 95 %
    assert sorted(mine) == sorted([(195, 0, 7000 + 60000, 7000 + 60099, 0, NO, n + 7),
                                   (200, 0, 7000, 7099, 0, 7, NO)])
    others = [int(c) for k, c in enumerate(rp["count"]) if k != 7]
    assert set(others) == {1}


def _fuzz_overlaps(kslam, rng, n_pairs, n_entries, per_read=3.0, paired=True):
    """Random overlap records in alignToDatabase order; many score ties to exercise the unstable sorts."""
    n_reads = 2 * n_pairs if paired else n_pairs
    k = rng.poisson(per_read, n_reads)
    read = np.repeat(np.arange(n_reads), k)
    m = len(read)
    a = np.zeros(m, dtype=kslam.OVERLAP_DT)
    a["read"] = read
    a["entry"] = rng.integers(0, n_entries, m)
    base = rng.integers(0, 5000, n_reads)[read % max(n_pairs, 1)]
    a["rel"] = base + rng.integers(-400, 400, m)
    a["revcomp"] = rng.integers(0, 2, m)
    a["score"] = rng.choice([120, 150, 180, 190, 200], m)
    a["ref_begin"] = np.maximum(a["rel"], 0)
    a["ref_end"] = a["ref_begin"] + rng.integers(60, 100, m)
    a["query_begin"] = 0
    a["query_end"] = 99
    order = np.lexsort((a["rel"], a["entry"], a["read"]))
    return a[order], n_reads


@pytest.mark.parametrize("seed,paired,kw", [
    (1, True, {}),
    (2, True, {"score_threshold": 150, "score_fraction": 0.8}),
    (3, True, {"pseudo_assembly": False}),
    (4, False, {}),
    (5, True, {"stages": 1}),
    (6, True, {"stages": 2}),
    (7, True, {"stages": 4}),
    (8, False, {"score_threshold": 160, "num_sam_alignments": 2}),
])
def test_product_equals_restatement_on_random_overlaps(kslam, oracle, T, seed, paired, kw):
    rng = np.random.default_rng(seed)
    ov, n_reads = _fuzz_overlaps(kslam, rng, 6000, 12, paired=paired)
    reads = T.Reads([b"A" * 100] * n_reads)
    index = T.Index([b"A" * 8000] * 12, taxonomy_ids=list(range(50, 62)),
                    genes=[[(0, 4000, b"g%d" % e, b"", b"left half"), (4000, 8000, b"", b"P%d" % e, b"right")]
                           for e in range(12)])
    P = T.TailParams.default(paired=paired, report_cigar=False, threads=4, **kw)
    rp, pr, st = T.tail_pairs(P, reads, ov)
    orp, opr = oracle.tail_pairs(P, reads.view, ov)
    assert rp.tobytes() == orp.tobytes()
    assert pr.tobytes() == opr.tobytes()
    assert st.n_paired_final == len(pr) and st.n_read_pairs == len(rp)
    sam, _ = T.tail_sam(P, reads, index, ov, np.zeros(0, np.uint32))
    assert sam == oracle.tail_sam(P, reads.view, index.view, ov, np.zeros(0, np.uint32))
    # the two-step form gives the same text as the fused call
    assert T.sam_records(P, reads, index, ov, np.zeros(0, np.uint32), rp, pr) == sam
    # thread count does not change anything
    P1 = T.TailParams.default(paired=paired, report_cigar=False, threads=1, **kw)
    assert T.tail_sam(P1, reads, index, ov, np.zeros(0, np.uint32))[0] == sam


def _aligned_case(oracle, synth, T, seed, n_pairs, **read_kw):
    genomes = synth.make_genomes(seed, 3, 3, 20000, shared_segment=3000)
    reads, _ = synth.make_paired_reads(seed + 1, genomes, n_pairs, read_len=100, frag_mean=300, frag_sd=40,
                                       sub_rate=0.02, indel_rate=0.004, edge_frac=0.05, **read_kw)
    rb, gb = synth.to_bytes(reads), synth.to_bytes(genomes)
    rng = np.random.default_rng(seed)
    quals = [bytes(rng.integers(35, 74, len(b), dtype=np.uint8)) for b in rb]
    ids = [b"frag%d" % (i % n_pairs) for i in range(2 * n_pairs)]
    R = T.Reads(rb, quals, ids)
    genes = [[(100 + 900 * k, 900 * k + 900, b"gene%d" % k, b"WP_%d" % k if k % 3 else b"", b"product %d" % k)
              for k in range(20)] for _ in gb]
    I = T.Index(gb, locus_tags=[b"NC_%06d" % i for i in range(len(gb))],
                taxonomy_ids=[100 + i // 3 for i in range(len(gb))], genes=genes)
    return rb, gb, quals, R, I


_MD_TOKEN = re.compile(rb"(\d+)|\^([A-Z]+)|([A-Z])")
_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _check_row_against_genome(f, tags, read, genome):
    """read + CIGAR + MD must re-create genome[pos-1 ...]; NM must be the edit count."""
    flag, pos, cigar = int(f[1]), int(f[3]), f[5]
    seq = read[::-1].translate(_COMP) if flag & 0x10 else read
    ops = [(int(n), op) for n, op in re.findall(rb"(\d+)([MIDS])", cigar)]
    assert b"".join(b"%d%s" % (n, op) for n, op in ops) == cigar
    assert sum(n for n, op in ops if op in b"MIS") == len(read)
    md = [m.groups() for m in _MD_TOKEN.finditer(tags[b"MD"])]
    assert b"".join((a or b"") + (b"^" + b if b else b"") + (c or b"") for a, b, c in md) == tags[b"MD"]
    # expand MD into per-reference-base instructions
    steps = []
    for num, dele, mis in md:
        if num is not None:
            steps += [None] * int(num)
        elif dele is not None:
            steps += [("D", bytes([c])) for c in dele]
        else:
            steps.append(("X", mis))
    ref, q, s, nm = bytearray(), 0, 0, 0
    for n, op in ops:
        if op == b"S":
            q += n
        elif op == b"I":
            q += n
            nm += n
        elif op == b"M":
            for _ in range(n):
                st = steps[s]
                s += 1
                if st is None:
                    ref.append(seq[q])
                else:
                    assert st[0] == "X" and st[1] != seq[q:q + 1]
                    ref += st[1]
                    nm += 1
                q += 1
        else:
            for _ in range(n):
                st = steps[s]
                s += 1
                assert st is not None and st[0] == "D"
                ref += st[1]
                nm += 1
    assert s == len(steps) and q == len(read)
    assert bytes(ref) == genome[pos - 1:pos - 1 + len(ref)]
    assert nm == int(tags[b"NM"])


def test_sam_text_matches_restatement_and_the_sam_definition(kslam, oracle, synth, T):
    n_pairs = 1200
    rb, gb, quals, R, I = _aligned_case(oracle, synth, T, 11, n_pairs)
    al, cig, _ = oracle.align_to_database(rb, gb, oracle.Params.default())
    P = T.TailParams.default(threads=4)
    sam, st = T.tail_sam(P, R, I, al, cig)
    assert sam == oracle.tail_sam(P, R.view, I.view, al, cig)
    assert T.sam_header(I, b"SLAM --db db r1.fq r2.fq") == oracle.sam_header(I.view, b"SLAM --db db r1.fq r2.fq")
    lines = sam.split(b"\n")
    assert lines[-1] == b"" and len(lines) - 1 == 2 * (len(lines) // 2)
    locus = {b"NC_%06d" % i: i for i in range(len(gb))}
    checked = primaries = 0
    for k in range(0, len(lines) - 1, 2):
        a, b = lines[k].split(b"\t"), lines[k + 1].split(b"\t")
        fa, fb = int(a[1]), int(b[1])
        pid = int(a[0][4:])
        assert a[0] == b[0] and (fa & 0x41) == 0x41 and (fb & 0x81) == 0x81     # paired, first / last
        assert (fa & 0x100) == (fb & 0x100)
        primaries += not (fa & 0x100)
        assert bool(fa & 0x4) == bool(fb & 0x8) and bool(fb & 0x4) == bool(fa & 0x8)  # mate unmapped mirrors
        if not (fa | fb) & 0x4:   # the reference sets the mate-strand bit only when both mates align
            assert bool(fa & 0x10) == bool(fb & 0x20) and bool(fb & 0x10) == bool(fa & 0x20)
            assert (fa & 0x2) and (fb & 0x2)
        assert a[2] == b[2] and a[6] == b"=" and b[6] == b"="
        assert int(a[7]) == int(b[3]) and int(b[7]) == int(a[3])                # pnext = mate pos
        assert int(a[8]) == -int(b[8])                                          # tlen antisymmetric
        for f, flag, read in ((a, fa, rb[pid]), (b, fb, rb[n_pairs + pid])):
            assert 0 <= int(f[4]) <= 50 and f[9] == b"*" and f[10] == b"*"
            if flag & 0x4:
                assert len(f) == 11 and f[5] == b"*"
                continue
            tags = dict((t[:2], t[5:]) for t in f[11:])
            assert [t[:2] for t in f[11:15]] == [b"MD", b"AS", b"XS", b"NM"]
            _check_row_against_genome(f, tags, read, gb[locus[f[2]]])
            assert int(tags[b"XT"]) == 100 + locus[f[2]] // 3
            checked += 1
    assert checked > 2 * n_pairs * 0.8 and primaries == st.n_read_pairs


def test_sam_text_written_through_the_fd_writer(kslam, oracle, synth, T, tmp_path):
    """kslam_write_fd and the background writer (kslam_sam_writer_* / kslam_write_queued: buffers handed over, written by
    the writer's thread in order, reused): a header first, two batches after it, into a file and into a pipe -- always the
    text of kslam_tail_sam; a failing descriptor is reported."""
    import ctypes as C
    import os
    import threading
    rb, gb, quals, R, I = _aligned_case(oracle, synth, T, 13, 900)
    al, cig, _ = oracle.align_to_database(rb, gb, oracle.Params.default())
    P = T.TailParams.default(threads=5)
    exp, _ = T.tail_sam(P, R, I, al, cig)
    L = T.lib()
    ov = np.ascontiguousarray(al, dtype=kslam.OVERLAP_DT)
    pool = np.ascontiguousarray(cig, dtype=np.uint32)
    writer = C.cast(L.kslam_write_fd, T.WRITE_FN)

    def write_to(fd):
        keep = C.c_int(fd)
        st = T.TailStats()
        T._chk(L.kslam_tail_sam_write(C.byref(P), C.byref(R.view), C.byref(I.view), ov.ctypes.data, len(ov), pool.ctypes.data,
                                      len(pool), writer, C.cast(C.pointer(keep), C.c_void_p), C.byref(st)))
        return st
    path = str(tmp_path / "x.sam")
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    os.write(fd, b"@HD\theader\n")
    st = write_to(fd)
    write_to(fd)
    os.close(fd)
    assert open(path, "rb").read() == b"@HD\theader\n" + exp + exp and st.sam_bytes == len(exp) > 100000
    r, w = os.pipe()
    got = []
    reader = threading.Thread(target=lambda: got.append(os.fdopen(r, "rb").read()))
    reader.start()
    write_to(w)
    os.close(w)
    reader.join()
    assert got[0] == exp

    def queued(fd, n_batches):
        wr = T.SamWriter(fd)
        wr.write(b"@HD\theader\n")
        st = T.TailStats()
        for _ in range(n_batches):
            T._chk(L.kslam_tail_sam_write(C.byref(P), C.byref(R.view), C.byref(I.view), ov.ctypes.data, len(ov), pool.ctypes.data,
                                          len(pool), wr.callback, wr._h, C.byref(st)))
        return wr.close()
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    n, sec = queued(fd, 5)
    os.close(fd)
    assert open(path, "rb").read() == b"@HD\theader\n" + exp * 5 and n == 11 + 5 * len(exp) and sec >= 0
    r, w = os.pipe()
    got = []
    reader = threading.Thread(target=lambda: got.append(os.fdopen(r, "rb").read()))
    reader.start()
    queued(w, 2)
    os.close(w)
    reader.join()
    assert got[0] == b"@HD\theader\n" + exp * 2
    rd = os.open(path, os.O_RDONLY)            # not open for writing: the writer's thread fails, close() says so
    with pytest.raises(kslam.KslamError, match="writing the SAM text failed"):
        queued(rd, 1)
    os.close(rd)


@pytest.mark.parametrize("kw", [{}, {"num_sam_alignments": 1}, {"paired": False}, {"score_threshold": 120}])
def test_sam_from_precomputed_row_details_is_the_same_text(kslam, oracle, synth, T, kw):
    """kslam_tail_sam_rows: NM / log-probability / MD handed in per row (what kslam_row_details computes
    on the GPU; here from tests/rowdetails_ref.py) -> byte-identical SAM text, without the writer
    touching the entry bases (the index view it gets holds garbage there)."""
    from rowdetails_ref import row_details
    n_pairs = 700
    rb, gb, quals, R, I = _aligned_case(oracle, synth, T, 51, n_pairs, n_rate=0.004)
    al, cig, _ = oracle.align_to_database(rb, gb, oracle.Params.default(score_threshold=kw.get("score_threshold", 0)))
    det, md = row_details(al, cig, rb, quals, gb, kslam.ROW_DETAIL_DT)
    assert (det["nm"] > 0).sum() > 500 and (det["md_len"] > 3).sum() > 500
    P = T.TailParams.default(threads=3, **kw)
    exp, _ = T.tail_sam(P, R, I, al, cig)
    genes = [[(100 + 900 * k, 900 * k + 900, b"gene%d" % k, b"WP_%d" % k if k % 3 else b"", b"product %d" % k)
              for k in range(20)] for _ in gb]
    scrambled = T.Index([bytes(len(g)) for g in gb], locus_tags=[b"NC_%06d" % i for i in range(len(gb))],
                        taxonomy_ids=[100 + i // 3 for i in range(len(gb))], genes=genes)
    got, st = T.tail_sam_rows(P, R, scrambled, al, cig, det, md)
    assert got == exp and len(got) > 100000
    # errors travel: a row whose CIGAR ran off its read, a quality character that is no phred+33 value
    bad = det.copy()
    bad["flags"][:] = 2
    with pytest.raises(kslam.KslamError, match="cigar runs past"):
        T.tail_sam_rows(P, R, scrambled, al, cig, bad, md)
    bad = det.copy()
    bad["md_off"] += np.uint64(len(md))
    with pytest.raises(kslam.KslamError, match="MD slice"):
        T.tail_sam_rows(P, R, scrambled, al, cig, bad, md)


def test_tail_with_score_threshold_and_n_bases(kslam, oracle, synth, T):
    rb, gb, quals, R, I = _aligned_case(oracle, synth, T, 21, 500, n_rate=0.01)
    al, cig, _ = oracle.align_to_database(rb, gb, oracle.Params.default(score_threshold=150))
    for kw in ({"score_threshold": 150}, {"score_threshold": 150, "sam_xa": True, "num_sam_alignments": 1},
               {"paired": False}):
        P = T.TailParams.default(threads=2, **kw)
        sam, st = T.tail_sam(P, R, I, al, cig)
        assert sam == oracle.tail_sam(P, R.view, I.view, al, cig)
        assert st.n_overlaps_screened == int((al["score"] >= kw.get("score_threshold", 0)).sum())


def test_tail_argument_errors(kslam, T):
    reads = T.Reads([b"A" * 100] * 4)
    P = T.TailParams.default(threads=1)
    good = _ov(kslam, [(0, 0, 5, 0, 100, 5, 104), (2, 0, 200, 1, 100, 200, 299)])
    T.tail_pairs(P, reads, good)
    with pytest.raises(kslam.KslamError) as e:      # not in (read, entry, rel) order
        T.tail_pairs(P, reads, good[::-1].copy())
    assert e.value.status == 1 and "order" in str(e.value)
    bad = good.copy()
    bad["read"][1] = 9
    with pytest.raises(kslam.KslamError) as e:
        T.tail_pairs(P, reads, bad)
    assert e.value.status == 1
    with pytest.raises(kslam.KslamError) as e:      # paired layout needs an even read count
        T.tail_pairs(P, T.Reads([b"A" * 100] * 3), good[:1])
    assert e.value.status == 1
    index = T.Index([b"A" * 50])
    cig = np.array([(200 << 4) | 0], dtype=np.uint32)     # 200M on a 100-base read
    ov = good[:1].copy()
    ov["cigar_len"], ov["cigar_off"] = 1, 0
    with pytest.raises(kslam.KslamError) as e:
        T.tail_sam(P, reads, index, ov, cig)
    assert e.value.status == 1 and "cigar" in str(e.value)
    # empty input is fine
    rp, pr, st = T.tail_pairs(P, reads, good[:0])
    assert len(rp) == 0 and len(pr) == 0
    assert T.tail_sam(P, reads, index, good[:0], cig[:0])[0] == b""


@pytest.mark.gpu
def test_gpu_hot_path_into_tail_matches_cpu_chain(kslam, oracle, synth, T):
    """alignToDatabase on the MI355X -> host tail == CPU alignToDatabase -> serial tail, byte for byte."""
    n_pairs = 4000
    rb, gb, quals, R, I = _aligned_case(oracle, synth, T, 31, n_pairs)
    ctx = kslam.Context(report_cigar=True)
    ctx.set_index(gb)
    al, cig = ctx.align_batch(rb)
    P = T.TailParams.default()
    sam, st = T.tail_sam(P, R, I, al, cig)
    eal, ecig, _ = oracle.align_to_database(rb, gb, oracle.Params.default())
    assert sam == oracle.tail_sam(P, R.view, I.view, eal, ecig)
    assert sam.count(b"\n") >= 2 * st.n_read_pairs > n_pairs


def test_gnu_sort_header_reproduces_std_sort(tmp_path):
    """csrc/gnu_sort.h (the std::sort restatement the device pairing / screens use) against the real
    std::sort of this toolchain: 120 k tie-heavy, adversarial and random arrays, element for element."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    assert gxx
    exe = str(tmp_path / "gnu_sort_check")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([gxx, "-O2", "-std=c++17", os.path.join(root, "tests", "gnu_sort_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "GNU_SORT_OK" in r.stdout, r.stdout + r.stderr


def test_worker_pool_with_concurrent_loops(tmp_path):
    """k-slam_amd/host/workers.hpp: parallel loops started from several threads at once (the batch loop's SAM-text and
    taxonomy threads), from inside a task, and with a failing task -- under ThreadSanitizer where the toolchain has it."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    assert gxx
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "pool_check.cpp")
    exe = str(tmp_path / "pool_check")
    tsan = subprocess.run([gxx, "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", src, "-o", exe], capture_output=True).returncode == 0
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600) if tsan else None
    if r is None or (r.returncode != 0 and "data race" not in r.stderr and "FAILED" not in r.stdout):
        # no sanitizer in this toolchain, or its runtime cannot start here (address-space layout of some containers)
        subprocess.check_call([gxx, "-O2", "-std=c++17", "-pthread", src, "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_finish_prepare_then_read_only_write_equals_the_one_call_finish(kslam, oracle, synth, T):
    """host/stream.cpp's order: kslam_tail_finish_prepare (everything that changes the pair arrays: host pseudo-assembly +
    second screen, writeSAMOutputPairs' per-pair sort) and only then the SAM text and the classification side by side.
    The two-step route gives the one-call route's text and leaves the arrays in the state the one call leaves them in --
    with read pairs of more than 16 alignment pairs (introsort really moves records there) and with the host running the
    pseudo-assembly (the device's fallback)."""
    n_pairs = 500
    # 24 near-identical genomes: a read aligns to all of them -> groups of > 16 alignment pairs with tied scores
    root = synth.make_genomes(5, 1, 1, 6000)[0]
    rng = np.random.default_rng(5)
    genomes = [synth.mutate(rng, root, 0.002, 0.0) for _ in range(24)]
    reads, _ = synth.make_paired_reads(6, genomes, n_pairs, read_len=100, frag_mean=300, frag_sd=40, sub_rate=0.01, indel_rate=0.002)
    rb, gb = synth.to_bytes(reads), synth.to_bytes(genomes)
    R = T.Reads(rb, [b"I" * len(b) for b in rb], [b"f%d" % (i % n_pairs) for i in range(2 * n_pairs)])
    I = T.Index(gb, locus_tags=[b"NC_%06d" % i for i in range(len(gb))], taxonomy_ids=[100 + i for i in range(len(gb))])
    al, cig, _ = oracle.align_to_database(rb, gb, oracle.Params.default())
    for pseudo in (True, False):
        P_screens = T.TailParams.default(threads=4, pseudo_assembly=False, stages=3)      # what the device hands over
        rp, pr, _ = T.tail_pairs(P_screens, R, al)
        assert int(rp["count"].max()) > 16
        P_all = T.TailParams.default(threads=4, pseudo_assembly=pseudo)
        rp1, pr1, one = rp.copy(), pr.copy(), []
        T.tail_finish_rows(P_all, R, I, al, cig, None, None, rp1, pr1, sink=one.append)
        rp2, pr2, two = rp.copy(), pr.copy(), []
        st = T.tail_finish_prepare(P_all, R, al, rp2, pr2, sort_groups=True)
        frozen = (rp2.copy(), pr2.copy())
        P_ro = T.TailParams.default(threads=4, pseudo_assembly=False, stages=7 | 16)
        T.tail_finish_rows(P_ro, R, I, al, cig, None, None, rp2, pr2, sink=two.append)
        assert (rp2 == frozen[0]).all() and (pr2 == frozen[1]).all()       # the write really only read
        assert b"".join(two) == b"".join(one) and len(one) > 0
        assert (rp2 == rp1).all() and (pr2 == pr1).all()
        assert st.n_paired_final == int(rp2["count"].sum())
        # and the whole-chain entry agrees (the reference order: screens, pseudo, screen, sort, text)
        assert b"".join(one) == T.tail_sam(P_all, R, I, al, cig)[0]
    # without a SAM file the reference does not sort: sort_groups=False leaves the screened order alone
    rp3, pr3 = rp.copy(), pr.copy()
    T.tail_finish_prepare(T.TailParams.default(threads=4, pseudo_assembly=False), R, al, rp3, pr3, sort_groups=False)
    assert (rp3 == rp).all() and (pr3 == pr).all()


def test_background_writer_appends_in_order_from_where_the_descriptor_stands(kslam, T, tmp_path):
    """kslam_sam_writer: small and large pieces, an odd starting offset, appended in order; the file is what was queued, byte
    for byte, and the descriptor stands at its end."""
    rng = np.random.default_rng(3)
    pieces = [b"@HD\tVN:1.0\n", rng.integers(0, 256, 3_000_001, dtype=np.uint8).tobytes(), b"x" * 17,
              rng.integers(0, 256, 5_123_457, dtype=np.uint8).tobytes(), rng.integers(0, 256, 1_048_576, dtype=np.uint8).tobytes()]
    path = str(tmp_path / "out.sam")
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    os.write(fd, b"abc")                      # the writer starts where the descriptor stands
    w = T.SamWriter(fd)
    for p in pieces:
        w.write(p)
    n, sec = w.close()
    assert n == sum(len(p) for p in pieces)
    assert os.lseek(fd, 0, os.SEEK_CUR) == 3 + n
    os.close(fd)
    assert open(path, "rb").read() == b"abc" + b"".join(pieces)


def test_small_divider():
    """csrc/sw.hip SmallDiv: (a * (2^20 / b + 1)) >> 20 == a / b for every dividend the fast path admits (a < 4096) and every
    divisor it admits (b in 1..255) -- the arithmetic identity the certificate's three divisions rest on, exhaustively."""
    a = np.arange(4096, dtype=np.uint64)
    for b in range(1, 256):
        m = np.uint64((1 << 20) // b + 1)
        assert np.array_equal((a * m) >> np.uint64(20), a // np.uint64(b)), b
        assert int(a[-1] * m) < 1 << 32   # the product fits the 32-bit multiply
