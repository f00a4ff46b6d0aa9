"""The first half of the host tail on the GPU (csrc/pairs.hip): score
screen, read pairing, insert-size statistics, insert-size screen and score-fraction screen -- against the
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
    (9, True, 0, 0.5, 0, 3.0), (10, True, 250, 0.95, 3, 3.0)])
def test_device_pairing_and_screens_equal_the_host_tail(kslam, T, ctx, seed, paired, thr, frac, stages, per_read):
    """random overlap records full of score ties (read pairs with up to ~100 records reach the introsort
    part of std::sort): read pairs and alignment pairs equal kslam_tail_pairs', byte for byte"""
    from test_tail import _fuzz_overlaps
    rng = np.random.default_rng(100 + seed)
    n_units = 4000 if per_read < 10 else 700
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
    c.set_pairing(stages=0)
    o, g, d, m, release = c.collect_batch(c.submit_batch_full(len(rb), C.cast(bp, C.c_void_p), None, lens.ctypes.data))
    assert c.last_pairs is None
    release()
    c.close()
