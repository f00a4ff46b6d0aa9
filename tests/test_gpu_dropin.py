"""THE LITERAL DROP-IN: the reference's own program with the GPU operator in it.

oracle/_ref/libslam_gpu_ref.so is oracle/ref_slam_driver.cpp built with -DKSLAM_REF_GPU_OPERATOR: the reference's
metagenomicAnalysis_Low_Mem (src/SLAM.h:159-268) -- its FASTQ reader, batch loop, score screen, pairing, insert-size
statistics, screens, pseudo-assembly, SAM writer, per-read LCA and reports, all compiled from /root/reference/src where
they lie -- with ONE function swapped: alignToDatabase (src/SLAM.h:59-79) is cut out of the header by a line slice and
defined as INTEGRATION.md tells a maintainer to define it, through k-slam_amd/host/slam_hot_path.hpp over the C ABI of
libkslam_hip.so.  Its four output files must be, byte for byte,
  * the files the UNMODIFIED reference wrote for the same inputs (tests/golden/slam_loop.npz, tests/golden/c1_golden.json:
    recorded from oracle/_ref/libslam_ref.so by tests/golden/make_golden.py), and
  * the files oracle/_ref/libslam_ref.so writes in this very run (both libraries travel to the GPU box prebuilt).
"""
import importlib
import json
import os

import numpy as np
import pytest

import ref_loop_case as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need(oracle):
    if not oracle.have_ref_slam_gpu():
        pytest.fail("oracle/_ref/libslam_gpu_ref.so is missing: __graft_entry__.build() makes it where /root/reference exists, "
                    "and it travels to the GPU box")


@pytest.mark.parametrize("tag", ["a", "b"])
def test_reference_program_with_the_gpu_operator_writes_the_recorded_reference_files(kslam, oracle, tmp_path, tag):
    """slam_loop.npz (420 / 300 pairs x 100 bp, several batches, pseudo-assembly on / off): SAM incl. header, XML report,
    _abbreviated and _PerRead of the patched program == the files the unpatched reference wrote."""
    _need(oracle)
    from test_reference_loop import load_fixture_case
    D = importlib.import_module("kslam_amd.db")
    z = np.load(os.path.join(ROOT, "tests", "golden", "slam_loop.npz"), allow_pickle=False)
    case = load_fixture_case(z, tag)
    dbdir = R.write_case(case, tmp_path, D)
    got = R.run_reference(oracle, case, tmp_path, dbdir, int(z[tag + "_per_batch"]), pseudo=bool(z[tag + "_pseudo"]), gpu_operator=True)
    for k in ("sam", "xml", "abbreviated", "per_read"):
        assert got[k] == z[tag + "_" + k].tobytes(), k
    if got["log"] is not None:      # the reference opens ./log.txt once per process: only the first run of a process has one
        assert b"Aligning reads to database using k = 32" in got["log"] and b"Performing pairwise Smith-Waterman" not in got["log"]


@pytest.mark.parametrize("seed,pseudo", [(1, True), (1, False), (2, True), (2, False)])
def test_dropin_at_c1_size_equals_the_golden_and_the_reference_run_beside_it(kslam, oracle, synth, tmp_path, seed, pseudo):
    """SURVEY 8c golden (5) / BASELINE configs[0]: 10 k pairs x 150 bp vs 3 x 2 Mb with a shared 20 kb segment, one batch.
    The patched program's files == tests/golden/c1_golden.json (md5 of the four files, first / last 200 SAM lines), and ==
    what the unpatched reference (oracle/_ref/libslam_ref.so) writes here for the same files."""
    _need(oracle)
    D = importlib.import_module("kslam_amd.db")
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "c1_golden.json")))["cases"]["seed%d_%s" % (seed, "pseudo" if pseudo else "nopseudo")]
    case = R.make_case_c1(synth, seed)
    assert R.digest_of_inputs(case) == gold["inputs_md5"], "the generator no longer makes the inputs the golden was recorded on"
    dbdir = R.write_case(case, tmp_path, D)
    got = R.run_reference(oracle, case, tmp_path, dbdir, 10_000_000, pseudo=pseudo, gpu_operator=True)
    d = R.digest_of_outputs(got)
    assert d["sam_head"] == gold["sam_head"] and d["sam_tail"] == gold["sam_tail"] and d["sam_lines"] == gold["sam_lines"]
    assert d["bytes"] == gold["bytes"] and d["md5"] == gold["md5"]
    if oracle.have_ref_slam():
        ref = R.run_reference(oracle, case, tmp_path, dbdir, 10_000_000, pseudo=pseudo)
        for k in ("sam", "xml", "abbreviated", "per_read"):
            assert got[k] == ref[k], k


@pytest.mark.parametrize("scoring,kw", [({"match": 1, "mismatch": 4, "gap_open": 6, "gap_extend": 1}, {}),
                                        ({"match": 3, "mismatch": 2, "gap_open": 4, "gap_extend": 3}, {"score_threshold": 150}),
                                        ({}, {"sam_xa": True, "num_alignments": 3}), ({}, {"just_align": True})])
def test_dropin_follows_the_reference_flags(kslam, oracle, synth, tmp_path, scoring, kw):
    """The globals the operator reads (match / misMatch / gapOpen / gapExtend / scoreThreshold, src/Globals.h:27-36) and
    flags of the loop around it (--sam-xa, --num-alignments, --just-align, several batches): patched == unpatched, side by
    side, on 1 500 pairs x 110 bp vs 9 strain genomes."""
    _need(oracle)
    if not oracle.have_ref_slam():
        pytest.skip("oracle/_ref/libslam_ref.so not present")
    D = importlib.import_module("kslam_amd.db")
    case = R.make_case(synth, n_pairs=1500, seed=6301)
    dbdir = R.write_case(case, tmp_path, D)
    got = R.run_reference(oracle, case, tmp_path, dbdir, 400, gpu_operator=True, scoring=scoring, **kw)
    ref = R.run_reference(oracle, case, tmp_path, dbdir, 400, scoring=scoring, **kw)
    for k in ("sam", "xml", "abbreviated", "per_read"):
        assert got[k] == ref[k], k
    assert len(ref["sam"]) > 100 * 1500
