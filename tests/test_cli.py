"""k-slam_amd/SLAM (tools/slam_main.cpp): the reference's command line, src/main.cpp:24-157 -> src/SLAM.h:159-268, over the
C ABI.  CPU: flags that need no GPU (--version, --help, --parse-fasta).  -m gpu: the binary on files against the files the
reference's OWN loop wrote (tests/golden/slam_loop.npz) and, for the other modes, against the oracle chain."""
import importlib
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SLAM = os.path.join(ROOT, "k-slam_amd", "SLAM")


def _run(args, cwd, check=True):
    r = subprocess.run([SLAM] + args, cwd=str(cwd), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    if check:
        assert r.returncode == 0, r.stderr.decode()
    return r


def test_version_and_help(kslam, tmp_path):
    assert os.path.exists(SLAM), "k-slam_amd/SLAM is built by __graft_entry__.build()"
    r = _run(["--version"], tmp_path, check=False)
    assert (r.returncode, r.stdout) == (1, b"1.0\n")                       # src/main.cpp:95-98
    for args in ([], ["--help"]):
        r = _run(args, tmp_path, check=False)
        assert r.returncode == 1 and r.stdout.startswith(b"Usage\tSLAM [option] --db=DATABASE R1FILE R2FILE\n")
        for flag in (b"--db", b"--min-alignment-score arg (=0)", b"--score-fraction-threshold arg (=0.95)", b"--match-score arg (=2)",
                     b"--mismatch-penalty arg (=3)", b"--gap-open arg (=5)", b"--gap-extend arg (=2)", b"--num-reads arg (=4294967295)",
                     b"--num-reads-at-once arg (=10000000)", b"--output-file", b"--sam-file", b"--num-alignments arg (=10)",
                     b"--sam-xa", b"--just-align", b"--no-pseudo-assembly"):
            assert flag in r.stdout, flag                                    # src/main.cpp:36-71
    r = _run(["--no-such-flag"], tmp_path, check=False)
    assert r.returncode != 0 and b"unrecognised option" in r.stderr
    r = _run(["--gap-open", "x", "--db", "d", "r.fq"], tmp_path, check=False)
    assert r.returncode != 0 and b"gap-open" in r.stderr


def test_parse_fasta_builds_the_database_the_loader_reads(kslam, tmp_path):
    """createIndexFromFASTA, src/GenbankTools.h:224-260: locusTag = text between '>' and the first space (EMPTY when the
    header has no space), bases upper-cased and joined over lines, entries without bases dropped, \\r\\n accepted."""
    D = importlib.import_module("kslam_amd.db")
    (tmp_path / "a.fa").write_bytes(b">NC_1.1 first genome\nACGTacgt\nNNAC\n\n>nospace\nGGGG\n>NC_3 empty entry follows\n>NC_4 d\r\nTTtt\r\nAA\r\n")
    (tmp_path / "b.fa").write_bytes(b"> leading space\nCCCC\n>NC_6 last")
    _run(["--parse-fasta", "--output-file", "database", "a.fa", "b.fa"], tmp_path)
    db = D.Database.load(tmp_path / "database")
    from oracle import db_oracle
    _, ents = db_oracle.parse((tmp_path / "database").read_bytes())
    assert [(e["locusTag"], e["bases"]) for e in ents] == [(b"NC_1.1", b"ACGTACGTNNAC"), (b"", b"GGGG"), (b"NC_4", b"TTTTAA"), (b"", b"CCCC")]
    assert all(e["taxonomyID"] == 0 and e["genes"] == [] for e in ents)
    assert db.n_entries == 4
    db.close()
    assert b"Parsing FASTA" in (tmp_path / "log.txt").read_bytes()


def _fixture_case(tag):
    from test_reference_loop import load_fixture_case
    z = np.load(os.path.join(ROOT, "tests", "golden", "slam_loop.npz"), allow_pickle=False)
    return z, load_fixture_case(z, tag)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_binary_on_files_equals_the_reference_loop(kslam, tmp_path, tag):
    """SLAM --db D --sam-file S --output-file O R1 R2 writes the four files the reference wrote for the same inputs
    (tests/golden/slam_loop.npz: recorded from the reference's own metagenomicAnalysis_Low_Mem), byte for byte; the @PG
    line carries this run's command line (src/main.cpp:25-30, src/SAM.h:513-531)."""
    import ref_loop_case as R
    D = importlib.import_module("kslam_amd.db")
    z, case = _fixture_case(tag)
    R.write_case(case, tmp_path, D)
    args = ["--db=db", "--sam-file", "out.sam", "--output-file=out", "--num-reads-at-once", str(int(z[tag + "_per_batch"]))]
    if not bool(z[tag + "_pseudo"]):
        args.append("--no-pseudo-assembly")
    args += ["R1.fq", "R2.fq"]
    _run(args, tmp_path)
    cl = (SLAM + " " + " ".join(args)).encode()
    exp_sam = z[tag + "_sam"].tobytes().replace(b'CL:"SLAM --db db R1.fq R2.fq"', b'CL:"' + cl + b'"')
    assert (tmp_path / "out.sam").read_bytes() == exp_sam
    assert (tmp_path / "out").read_bytes() == z[tag + "_xml"].tobytes()
    assert (tmp_path / "out_abbreviated").read_bytes() == z[tag + "_abbreviated"].tobytes()
    assert (tmp_path / "out_PerRead").read_bytes() == z[tag + "_per_read"].tobytes()
    log = (tmp_path / "log.txt").read_text()
    assert log.startswith("[t = 0.00s]\tPerforming metagenomic analysis\n") and log.rstrip().endswith("Done")
    assert "Processed\t%d\t reads" % case["n_pairs"] in log


@pytest.mark.gpu
@pytest.mark.parametrize("seed,pseudo", [(1, True), (1, False), (2, True), (2, False)])
def test_binary_at_c1_size_equals_the_golden(kslam, synth, tmp_path, seed, pseudo):
    """SURVEY 8c golden (5) / BASELINE configs[0] through the executable: 10 k pairs x 150 bp vs 3 x 2 Mb with a shared 20 kb
    segment, with and without pseudo-assembly.  tests/golden/c1_golden.json holds what the reference's own loop
    (src/SLAM.h:209-239 inside metagenomicAnalysis_Low_Mem, oracle/_ref/libslam_ref.so) wrote for these inputs: md5 and size
    of the four files, the first / last 200 SAM lines.  Only the @PG line differs (this run's command line)."""
    import hashlib
    import json
    import ref_loop_case as R
    D = importlib.import_module("kslam_amd.db")
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "c1_golden.json")))["cases"]["seed%d_%s" % (seed, "pseudo" if pseudo else "nopseudo")]
    case = R.make_case_c1(synth, seed)
    assert R.digest_of_inputs(case) == gold["inputs_md5"], "the generator no longer makes the inputs the golden was recorded on"
    R.write_case(case, tmp_path, D)
    args = ["--db=db", "--sam-file", "out.sam", "--output-file=out"] + ([] if pseudo else ["--no-pseudo-assembly"]) + ["R1.fq", "R2.fq"]
    _run(args, tmp_path)
    cl = (SLAM + " " + " ".join(args)).encode()
    sam = (tmp_path / "out.sam").read_bytes().replace(b'CL:"' + cl + b'"', b'CL:"SLAM --db db R1.fq R2.fq"')
    got = {"sam": sam, "xml": (tmp_path / "out").read_bytes(), "abbreviated": (tmp_path / "out_abbreviated").read_bytes(),
           "per_read": (tmp_path / "out_PerRead").read_bytes()}
    d = R.digest_of_outputs(got)
    assert d["sam_head"] == gold["sam_head"] and d["sam_tail"] == gold["sam_tail"] and d["sam_lines"] == gold["sam_lines"]
    assert d["bytes"] == gold["bytes"] and d["md5"] == gold["md5"]


@pytest.mark.gpu
def test_binary_other_modes_equal_the_oracle_chain(kslam, oracle, synth, tmp_path):
    """--just-align (SAM only), single-end input, XML on stdout without --output-file, --num-reads, --sam-xa,
    --num-alignments, --min-alignment-score: against the oracle chain (itself pinned to the reference's loop)."""
    import ref_loop_case as R
    D = importlib.import_module("kslam_amd.db")
    case = R.make_case(synth, n_pairs=600, seed=6101)
    R.write_case(case, tmp_path, D)
    # --just-align: no report files
    args = ["--db", "db", "--sam-file", "ja.sam", "--just-align", "--num-reads-at-once", "250", "R1.fq", "R2.fq"]
    _run(args, tmp_path)
    cl = (SLAM + " " + " ".join(args)).encode()
    assert (tmp_path / "ja.sam").read_bytes() == R.run_oracle_chain(oracle, case, 250, command_line=cl)["sam"]
    assert not (tmp_path / "_PerRead").exists()
    # flags of the tail + --num-reads cutting the stream short (src/SLAM.h:193, 201-203)
    args = ["--db", "db", "--sam-file", "f.sam", "--output-file", "f", "--num-reads-at-once", "200", "--num-reads", "500", "--sam-xa",
            "--min-alignment-score", "120", "R1.fq", "R2.fq"]
    _run(args, tmp_path)
    cl = (SLAM + " " + " ".join(args)).encode()
    sub = dict(case)
    n = 500
    sub["bases"] = case["bases"][:n] + case["bases"][600:600 + n]
    sub["quals"] = case["quals"][:n] + case["quals"][600:600 + n]
    sub["ids"], sub["n_pairs"] = case["ids"][:n], n
    exp = R.run_oracle_chain(oracle, sub, 200, sam_xa=True, score_threshold=120, command_line=cl)
    assert (tmp_path / "f.sam").read_bytes() == exp["sam"]
    assert (tmp_path / "f_PerRead").read_bytes() == exp["per_read"]
    assert (tmp_path / "f_abbreviated").read_bytes() == exp["abbreviated"]
    # single end, no --output-file: XML on stdout, <out>_PerRead is the file "_PerRead", no _abbreviated (src/SLAM.h:256-266)
    single = R.make_case(synth, n_pairs=400, seed=6202, paired=False)
    t2 = tmp_path / "single"
    t2.mkdir()
    R.write_case(single, t2, D)
    args = ["--db", "db", "--sam-file", "s.sam", "--num-alignments", "3", "R1.fq"]
    r = _run(args, t2)
    cl = (SLAM + " " + " ".join(args)).encode()
    exp = R.run_oracle_chain(oracle, single, 10000000, num_alignments=3, command_line=cl)
    assert (t2 / "s.sam").read_bytes() == exp["sam"]
    assert (t2 / "_PerRead").read_bytes() == exp["per_read"]
    assert b"<taxon>" in r.stdout[:10] and r.stdout.rstrip().endswith(b"</taxon>") and not (t2 / "_abbreviated").exists()
