"""The first half of the host tail on the GPU (csrc/pairs.hip): score
screen, read pairing, insert-size statistics, insert-size screen, score-fraction screen, pseudo-assembly
and the second score screen -- against the
host tail (k-slam_amd/host/tail.cpp, itself compared with the oracle's serial restatement in
tests/test_tail.py), record for record.  Equal keys are the rule in these sorts, so the comparison is on
the exact permutation: the device reproduces libstdc++'s std::sort (csrc/gnu_sort.h,
tests/gnu_sort_check.cpp compares that with the real std::sort on the CPU)."""
import importlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T(kslam):
    return importlib.import_module("kslam_amd.tail")


@pytest.fixture(scope="module")
def ctx(kslam):
    c = kslam.Context()
    yield c
    c.close()


@pytest.mark.parametrize("seed,paired,thr,frac,stages,per_read", [
    (1, True, 0, 0.95, 3, 3.0), (2, True, 150, 0.8, 3, 3.0), (3, True, 0, 0.95, 1, 3.0), (4, True, 0, 0.95, 2, 3.0),
    (5, False, 0, 0.95, 3, 3.0), (6, False, 160, 0.95, 2, 3.0), (7, True, 0, 0.95, 3, 30.0), (8, True, 0, 1.0, 3, 12.0),
    (9, True, 0, 0.5, 0, 3.0), (10, True, 250, 0.95, 3, 3.0),
    # read pairs of hundreds to thousands of alignment pairs (reads in repeats): one wavefront per read pair (k_screen_big)
    (11, True, 0, 0.95, 3, 150.0), (12, True, 0, 0.6, 3, 400.0), (13, False, 0, 0.9, 2, 300.0), (14, True, 0, 0.95, 1, 250.0)])
def test_device_pairing_and_screens_equal_the_host_tail(kslam, T, ctx, seed, paired, thr, frac, stages, per_read):
    """random overlap records full of score ties (read pairs with up to ~100 records reach the introsort
    part of std::sort): read pairs and alignment pairs equal kslam_tail_pairs', byte for byte"""
    from test_tail import _fuzz_overlaps
    rng = np.random.default_rng(100 + seed)
    n_units = 4000 if per_read < 10 else (700 if per_read < 100 else 120)
    ov, n_reads = _fuzz_overlaps(kslam, rng, n_units, 12, per_read=per_read, paired=paired)
    if seed == 7:      # far-apart pairs: a spike in the insert-size ladder (the `limit` branch of the statistics)
        far = rng.random(len(ov)) < 0.03
        ov["rel"][far] += 500000
        ov["ref_begin"][far] += 500000
        ov["ref_end"][far] += 500000
        ov = ov[np.lexsort((ov["rel"], ov["entry"], ov["read"]))]
    reads = T.Reads([b"A" * 100] * n_reads)
    P = T.TailParams.default(paired=paired, report_cigar=False, threads=4, score_threshold=thr, score_fraction=frac,
                             pseudo_assembly=False, stages=stages if stages else 8)
    rp, pr, st = T.tail_pairs(P, reads, ov)
    got = ctx.pair_screen_overlaps(ov, np.full(n_reads, 100, dtype=np.uint32), paired=paired, score_threshold=thr,
                                   score_fraction=frac, stages=stages)
    grp, gpr = ctx.take_pairs()
    assert len(pr) > 500 or thr > 200      # (threshold above every score: nothing is left, on both sides)
    assert grp.tobytes() == rp.tobytes() and gpr.tobytes() == pr.tobytes()
    assert got["n_overlaps_screened"] == st.n_overlaps_screened and got["n_paired_initial"] == st.n_paired_initial
    assert got["n_read_pairs"] == len(rp) and got["n_pairs"] == len(pr)
    if paired and stages & 1:
        assert got["max_insert_size"] == st.max_insert_size and got["n_insert_sizes"] == st.n_insert_sizes


@pytest.mark.parametrize("seed,entries,per_read,stages", [(21, 300, 400.0, 3), (22, 1200, 700.0, 3), (23, 60, 500.0, 1)])
def test_read_pairs_that_meet_every_entry(kslam, T, ctx, seed, entries, per_read, stages):
    """What a read inside an rRNA-like repeat looks like: hundreds of rows per mate, a few per entry, over hundreds of
    entries -- the shape k_pair_big is for (a wavefront per read pair, a lane per entry; csrc/pairs.hip), incl. entries only
    one mate has.  Read pairs and alignment pairs == the host tail's, byte for byte."""
    from test_tail import _fuzz_overlaps
    rng = np.random.default_rng(300 + seed)
    ov, n_reads = _fuzz_overlaps(kslam, rng, 60, entries, per_read=per_read, paired=True)
    # some entries lose one mate's rows altogether (the mate-2-only path of the kernel, and mate-1-only runs)
    drop = (rng.random(len(ov)) < 0.25) & ((ov["entry"] % 5 == 1) == (ov["read"] < n_reads // 2))
    ov = ov[~drop]
    reads = T.Reads([b"A" * 100] * n_reads)
    P = T.TailParams.default(paired=True, report_cigar=False, threads=4, score_threshold=60, score_fraction=0.9,
                             pseudo_assembly=False, stages=stages)
    rp, pr, st = T.tail_pairs(P, reads, ov)
    got = ctx.pair_screen_overlaps(ov, np.full(n_reads, 100, dtype=np.uint32), paired=True, score_threshold=60,
                                   score_fraction=0.9, stages=stages)
    grp, gpr = ctx.take_pairs()
    per_unit = np.bincount(ov["read"] % (n_reads // 2), minlength=n_reads // 2)
    assert per_unit.max() > 256 and len(pr) > 500
    assert grp.tobytes() == rp.tobytes() and gpr.tobytes() == pr.tobytes()
    assert got["n_overlaps_screened"] == st.n_overlaps_screened and got["n_paired_initial"] == st.n_paired_initial
    if stages & 1:
        assert got["max_insert_size"] == st.max_insert_size and got["n_insert_sizes"] == st.n_insert_sizes


def test_wave_sort_produces_the_std_sort_permutation(kslam, ctx, tmp_path):
    """csrc/wave_gnu_sort.h (std::sort by one wavefront: parallel Hoare partition, per-leaf insertion sorts)
    against the real std::sort of this toolchain on ~3 000 arrays: every size from 0 to 300, sizes up to the
    4000 elements that are sorted in LDS, a dozen up to 262 144 that are sorted in global memory, few / many distinct keys, sorted / reversed / organ-pipe / all-equal / median-of-three
    killer inputs (the last drives std::sort into its heap-sort fallback)"""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    assert gxx
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "gnu_sort_check")
    subprocess.check_call([gxx, "-O2", "-std=c++17", os.path.join(root, "tests", "gnu_sort_check.cpp"), "-o", exe])
    rng = np.random.default_rng(5)
    segs = []
    sizes = list(range(0, 301)) * 4 + [int(x) for x in rng.integers(300, 4001, 1500)] + [4000] * 8 + [17, 33, 64, 65, 127, 128, 129]
    sizes += [4001, 4002, 5000, 8191, 8192, 20000, 20001, 65535, 65536, 65537, 100000, 262144]   # sorted in global memory
    for i, n in enumerate(sizes):
        distinct = [1, 2, 3, 20, 100000][i % 5]
        k = rng.integers(0, distinct, n).astype(np.int32)
        shape = i % 7
        if shape == 1:
            k.sort()
        elif shape == 2:
            k = np.sort(k)[::-1].copy()
        elif shape == 3 and n > 2:
            k.sort()
            k[n // 2:] = k[n // 2:][::-1]
        segs.append(k)
    for n in (17, 33, 64, 100, 257, 1000, 4000, 30000):      # Musser's median-of-three killer
        v = np.zeros(n, dtype=np.int32)
        h = n // 2
        for i in range(1, h + 1):
            if i % 2 == 1:
                v[i - 1] = i
                v[i] = h + i
            v[h + i - 1] = 2 * i
        segs.append(v)
    off = np.zeros(len(segs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in segs])
    keys = np.concatenate(segs).astype(np.int32)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as fh:
        fh.write(np.uint64(len(segs)).tobytes() + off.tobytes() + keys.tobytes())
    subprocess.check_call([exe, "perm", fin, fout])
    exp = np.fromfile(fout, dtype=np.uint32)
    got = ctx.debug_wave_sort(keys, off)
    bad = np.nonzero(got != exp)[0]
    assert len(bad) == 0, "first difference in segment %d (size %d)" % (
        np.searchsorted(off, bad[0], side="right") - 1, len(segs[np.searchsorted(off, bad[0], side="right") - 1]))
    # the permutations are not the stable ones (else this test would not see a wrong tie order)
    stable = np.concatenate([np.argsort(x, kind="stable") for x in segs]).astype(np.uint32)
    assert (stable != exp).sum() > 100000


def _compacted(rp, pr):
    """the device leaves the records where they are when the second screen shrinks a group; the host tail
    hands back the survivors packed -- pack the device's the same way"""
    cnt = rp["count"].astype(np.int64)
    starts = np.cumsum(cnt) - cnt
    idx = np.repeat(rp["first"].astype(np.int64) - starts, cnt) + np.arange(int(cnt.sum()))
    out = rp.copy()
    out["first"] = starts
    return out, pr[idx]


@pytest.mark.parametrize("seed,paired,thr,frac,per_read,n_entries,expect_device", [
    (21, True, 0, 0.95, 3.0, 12, True), (22, True, 150, 0.8, 3.0, 12, True), (23, False, 0, 0.95, 3.0, 12, True),
    (24, True, 0, 0.95, 3.0, 300, True), (25, True, 0, 0.5, 12.0, 40, True), (26, True, 0, 0.95, 12.0, 2, True),
    (27, True, 0, 1.0, 3.0, 7, True), (28, True, 0, 0.3, 60.0, 1, False)])
def test_device_pseudo_assembly_equals_the_host_tail(kslam, T, ctx, seed, paired, thr, frac, per_read, n_entries, expect_device):
    """pseudoAssembly + the second score screen on the device (stages = 7): chains of overlapping alignments
    per entry, scores from double sums in std::sort's order of equal starts -> the host tail's records, byte
    for byte; entries too big for a workgroup's LDS (seed 26: ~20 k spans each) are sorted in global memory; an
    entry beyond 262 144 spans (seed 28) leaves the stage to the host, and says so"""
    from test_tail import _fuzz_overlaps
    rng = np.random.default_rng(100 + seed)
    ov, n_reads = _fuzz_overlaps(kslam, rng, 3000, n_entries, per_read=per_read, paired=paired)
    if seed == 24:      # degenerate spans (refEnd == refStart: the reference divides by zero) and reversed ones
        z = rng.random(len(ov)) < 0.02
        ov["ref_end"][z] = ov["ref_begin"][z]
        w = rng.random(len(ov)) < 0.02
        ov["ref_end"][w] = ov["ref_begin"][w] - 5
    reads = T.Reads([b"A" * 100] * n_reads)
    got = ctx.pair_screen_overlaps(ov, np.full(n_reads, 100, dtype=np.uint32), paired=paired, score_threshold=thr,
                                   score_fraction=frac, stages=7)
    grp, gpr = ctx.take_pairs()
    assert bool(got["stages_done"] & 4) == expect_device and got["stages_done"] & 3 == (3 if paired else 2)
    P = T.TailParams.default(paired=paired, report_cigar=False, threads=4, score_threshold=thr, score_fraction=frac,
                             pseudo_assembly=True, stages=7 if expect_device else 3)
    rp, pr, st = T.tail_pairs(P, reads, ov)
    crp, cpr = _compacted(grp, gpr)
    assert len(pr) > 500
    assert crp.tobytes() == rp.tobytes() and cpr.tobytes() == pr.tobytes()
    if expect_device:   # the stage did something: chain scores differ from the pairing's sums
        P3 = T.TailParams.default(paired=paired, report_cigar=False, threads=4, score_threshold=thr, score_fraction=frac,
                                  pseudo_assembly=False, stages=3)
        _, pr3, _ = T.tail_pairs(P3, reads, ov)
        assert len(pr3) != len(pr) or pr3.tobytes() != pr.tobytes()


@pytest.mark.parametrize("pseudo", [False, True])
def test_alignment_to_sam_with_the_tail_front_on_the_gpu(kslam, oracle, synth, T, pseudo):
    """align -> row details -> pairing / screens on the GPU -> host: [pseudo-assembly, second screen,] SAM
    text == the whole tail on the host == the oracle chain; also through the pipelined lanes"""
    from test_tail import _aligned_case
    n_pairs = 3000
    rb, gb, quals, R, I = _aligned_case(oracle, synth, T, 77, n_pairs)
    P = T.TailParams.default(pseudo_assembly=pseudo)
    c = kslam.Context()
    c.set_index(gb)
    c.load_reads(rb)
    n_out, n_cig = c.align_resident()
    ov, cg = c.fetch_results(n_out, n_cig)
    c.load_qualities(quals)
    c.row_details()
    det, md = c.take_row_details(n_out)
    st = c.pair_screen(paired=True)
    rp, pr = c.take_pairs()
    P_front = T.TailParams.default(pseudo_assembly=False, stages=3)
    hrp, hpr, hst = T.tail_pairs(P_front, R, ov)
    assert rp.tobytes() == hrp.tobytes() and pr.tobytes() == hpr.tobytes() and st["max_insert_size"] == hst.max_insert_size
    chunks = []
    fst = T.tail_finish_rows(P, R, I, ov, cg, det, md, rp.copy(), pr.copy(), chunks.append)
    sam = b"".join(chunks)
    # the walk for the rows the pairs refer to only (what the lanes run after the device pairing): those rows'
    # records equal the full walk's, the others are zero, and the SAM text is the same
    c.row_details(of_pairs=True)
    det2, md2 = c.take_row_details(n_out)
    used = np.zeros(n_out, dtype=bool)
    for f in ("r1", "r2"):
        idx = pr[f][pr[f] != kslam.NO_OVERLAP] if hasattr(kslam, "NO_OVERLAP") else pr[f][pr[f] != 0xFFFFFFFF]
        used[idx] = True
    assert 0.1 < used.mean() < 0.9
    for f in ("logp", "nm", "md_len", "flags"):
        assert (det2[f][used].view(np.uint64 if f == "logp" else det2[f].dtype) ==
                det[f][used].view(np.uint64 if f == "logp" else det[f].dtype)).all(), f
    assert not det2["md_len"][~used].any() and not det2["nm"][~used].any() and (det2["logp"][~used] == 0).all()
    for k in np.nonzero(used)[0][:2000]:
        a, b = int(det[k]["md_off"]), int(det2[k]["md_off"])
        assert md[a:a + int(det[k]["md_len"])].tobytes() == md2[b:b + int(det2[k]["md_len"])].tobytes()
    chunks2 = []
    T.tail_finish_rows(P, R, I, ov, cg, det2, md2, rp.copy(), pr.copy(), chunks2.append)
    assert b"".join(chunks2) == sam
    exp, est = T.tail_sam(P, R, I, ov, cg)
    eal, ecig, _ = oracle.align_to_database(rb, gb, oracle.Params.default())
    assert sam == exp == oracle.tail_sam(P, R.view, I.view, eal, ecig)
    assert fst.n_paired_final == est.n_paired_final and len(sam) > 500000
    # the pipelined lanes with the pairing switched on
    import ctypes as C
    keep_b = [C.create_string_buffer(b, len(b) + 1) for b in rb]
    keep_q = [C.create_string_buffer(q, len(q) + 1) for q in quals]
    bp = (C.c_char_p * len(rb))(*[C.cast(x, C.c_char_p) for x in keep_b])
    qp = (C.c_char_p * len(rb))(*[C.cast(x, C.c_char_p) for x in keep_q])
    lens = np.array([len(b) for b in rb], dtype=np.uint32)
    c.set_pairing(paired=True)
    tickets = [c.submit_batch_full(len(rb), C.cast(bp, C.c_void_p), C.cast(qp, C.c_void_p), lens.ctypes.data) for _ in range(3)]
    for t in tickets:
        o, g, d, m, release = c.collect_batch(t)
        lrp, lpr, lst = c.last_pairs
        assert lrp.tobytes() == rp.tobytes() and lpr.tobytes() == pr.tobytes() and lst["n_pairs"] == len(pr)
        out = []
        T.tail_finish_rows(P, R, I, o, g, d, m, lrp.copy(), lpr.copy(), out.append)
        assert b"".join(out) == exp
        release()
    if pseudo:   # pseudo-assembly and the second screen on the device too: the host only writes
        c.set_pairing(paired=True, stages=7)
        o, g, d, m, release = c.collect_batch(c.submit_batch_full(len(rb), C.cast(bp, C.c_void_p), C.cast(qp, C.c_void_p), lens.ctypes.data))
        lrp, lpr, lst = c.last_pairs
        assert lst["stages_done"] == 7
        hrp7, hpr7, _ = T.tail_pairs(P, R, ov)
        crp, cpr = _compacted(lrp, lpr)
        assert crp.tobytes() == hrp7.tobytes() and cpr.tobytes() == hpr7.tobytes()
        out = []
        P_write = T.TailParams.default(pseudo_assembly=False)
        wst = T.tail_finish_rows(P_write, R, I, o, g, d, m, lrp.copy(), lpr.copy(), out.append)
        assert b"".join(out) == exp and wst.n_paired_final == est.n_paired_final
        release()
    c.set_pairing(stages=0)
    o, g, d, m, release = c.collect_batch(c.submit_batch_full(len(rb), C.cast(bp, C.c_void_p), None, lens.ctypes.data))
    assert c.last_pairs is None
    release()
    c.close()
    # a batch the library splits into several internal chunks: the lanes still pair once, on the whole batch, between
    # the chunks' SW stages and their CIGAR stages
    c2 = kslam.Context(max_kmers_per_chunk=60000)
    c2.set_index(gb)
    c2.set_pairing(paired=True, stages=7 if pseudo else 3)
    for _ in range(2):
        o, g, d, m, release = c2.collect_batch(c2.submit_batch_full(len(rb), C.cast(bp, C.c_void_p), C.cast(qp, C.c_void_p), lens.ctypes.data))
        lrp, lpr, lst = c2.last_pairs
        out = []
        Pw = T.TailParams.default(pseudo_assembly=pseudo and not (lst["stages_done"] & 4))
        T.tail_finish_rows(Pw, R, I, o, g, d, m, lrp.copy(), lpr.copy(), out.append)
        assert b"".join(out) == exp
        # CIGARs only where a pair refers to the row
        _, live = _compacted(lrp, lpr)          # (the second screen shrinks groups in place: only their live records count)
        used = np.zeros(len(o), dtype=bool)
        for f in ("r1", "r2"):
            used[live[f][live[f] != 0xFFFFFFFF]] = True
        assert not o["cigar_len"][~used].any() and o["cigar_len"][used].all()
        release()
    c2.close()


def test_round2_entry_points_refuse_misuse_loudly(kslam, synth):
    """state and argument errors of the entry points added this round: a status code and a message, never a
    crash or a silent default (the reference's tail throws / aborts on the same conditions)"""
    c = kslam.Context()
    with pytest.raises(kslam.KslamError, match="no batch loaded|STATE"):
        c.pair_screen(paired=True)                                  # nothing aligned yet
    with pytest.raises(kslam.KslamError, match="kslam_pair_screen has not been called|STATE"):
        c.take_pairs()
    genomes = synth.make_genomes(5, 1, 1, 5000)
    reads, _ = synth.make_paired_reads(6, genomes, 11, read_len=100, frag_mean=250, frag_sd=20)
    rb = synth.to_bytes(reads)
    c.set_index(synth.to_bytes(genomes))
    c.load_reads(rb[:21])                                           # an odd number of reads
    c.align_resident()
    with pytest.raises(kslam.KslamError, match="even, non-zero number of reads"):
        c.pair_screen(paired=True)
    st = c.pair_screen(paired=False, stages=7)                      # single-end: fine, insert screen not applicable
    assert st["stages_done"] == 6
    ov = np.zeros(3, dtype=kslam.OVERLAP_DT)
    with pytest.raises(kslam.KslamError, match="even, non-zero number of reads"):
        c.pair_screen_overlaps(ov, np.full(3, 100, dtype=np.uint32), paired=True)
    # a caller's own rows must come sorted by read and name reads of the batch: the first-row table has no place for others
    ov = np.zeros(6, dtype=kslam.OVERLAP_DT)
    ov["read"] = [0, 2, 1, 4, 5, 5]
    ov["score"] = 100
    with pytest.raises(kslam.KslamError, match="not sorted by read"):
        c.pair_screen_overlaps(ov, np.full(6, 100, dtype=np.uint32), paired=True)
    ov["read"] = [0, 1, 2, 4, 5, 9]
    with pytest.raises(kslam.KslamError, match="not sorted by read"):
        c.pair_screen_overlaps(ov, np.full(6, 100, dtype=np.uint32), paired=True)
    ov["read"] = [0, 1, 2, 4, 5, 5]
    assert c.pair_screen_overlaps(ov, np.full(6, 100, dtype=np.uint32), paired=True)["n_overlaps_screened"] == 6
    # rows that jump back and forth between two reads far apart: every jump forward is a stretch of more than
    # 64 reads without rows -- n / 2 of them, the list of stretches has room for n_reads / 64 + 2 -- and nothing marks them bad
    # until the jump back: the list must be bounded, the call refused, and the context usable afterwards
    big = np.zeros(40000, dtype=kslam.OVERLAP_DT)
    big["read"] = np.tile(np.array([0, 1000], dtype=np.uint32), 20000)
    big["score"] = 100
    for _ in range(3):
        with pytest.raises(kslam.KslamError, match="not sorted by read"):
            c.pair_screen_overlaps(big, np.full(2000, 100, dtype=np.uint32), paired=True)
    ov["read"] = [0, 1, 2, 4, 5, 5]
    assert c.pair_screen_overlaps(ov, np.full(6, 100, dtype=np.uint32), paired=True)["n_overlaps_screened"] == 6
    # the sort hook: descending segment bounds, a segment over the limit
    with pytest.raises(kslam.KslamError, match="segments must be ascending"):
        c.debug_wave_sort(np.zeros(10, dtype=np.int32), np.array([0, 8, 4, 10], dtype=np.uint64))
    with pytest.raises(kslam.KslamError, match="segments must be ascending and hold at most"):
        c.debug_wave_sort(np.zeros(300000, dtype=np.int32), np.array([0, 300000], dtype=np.uint64))
    assert len(c.debug_wave_sort(np.zeros(0, dtype=np.int32), np.array([0], dtype=np.uint64))) == 0
    # a ticket that was never issued, a ticket collected twice
    with pytest.raises(kslam.KslamError):
        c.collect_batch(12345)
    c.close()
