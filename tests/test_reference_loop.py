"""CPU tests against the reference's OWN batch loop (metagenomicAnalysis_Low_Mem, src/SLAM.h:159-268), compiled in
place as oracle/_ref/libslam_ref.so (oracle/ref_slam_driver.cpp) and run on real files:

  FASTQ pair + GenbankIndex + taxDB  ->  SAM (header included), <out> (XML), <out>_abbreviated, <out>_PerRead

 * the ORACLE chain (kslam_oracle.c -> tail_oracle.cpp -> taxonomy_oracle.cpp, what every -m gpu parity test checks
   the HIP path against) must produce the same bytes: this is the pin of rows a-5..a-9 and N1 of SURVEY.md section 8;
 * the PRODUCT's host-side stages that run without a GPU (host tail, per-read LCA, report writer, SAM header; the
   alignments come from the oracle here) must produce the same bytes too.
Skipped when oracle/_ref/libslam_ref.so is absent (a machine without /root/reference and without the prebuilt file).
The committed fixture tests/golden/slam_loop.npz holds the reference's answers for the -m gpu run of the whole product.
"""
import importlib
import os

import numpy as np
import pytest

import ref_loop_case as R


def _need_ref(oracle):
    if not oracle.have_ref_slam():
        pytest.skip("oracle/_ref/libslam_ref.so not built (no /root/reference)")


def _sam_body(sam):
    """SAM text after the header lines"""
    lines = sam.split(b"\n")
    k = 0
    while k < len(lines) and lines[k].startswith(b"@"):
        k += 1
    return b"\n".join(lines[k:])


@pytest.mark.parametrize("per_batch,pseudo,kw", [
    (200, True, {}),                                  # three batches, pseudo-assembly on
    (500, False, {}),                                 # one batch, --no-pseudo-assembly
    (333, True, {"sam_xa": True}),                    # --sam-xa: primary lines only
    (250, True, {"num_alignments": 2}),               # --num-alignments 2
    (500, True, {"score_threshold": 150}),            # --min-alignment-score 150: CIGAR gate + score screen
])
def test_oracle_chain_equals_the_real_reference_loop(kslam, oracle, synth, tmp_path, per_batch, pseudo, kw):
    _need_ref(oracle)
    D = importlib.import_module("kslam_amd.db")
    case = R.make_case(synth, n_pairs=500)
    dbdir = R.write_case(case, tmp_path, D)
    ref = R.run_reference(oracle, case, tmp_path, dbdir, per_batch, pseudo=pseudo, **kw)
    orc = R.run_oracle_chain(oracle, case, per_batch, pseudo=pseudo, **kw)
    assert ref["sam"].count(b"\n") > 700 and ref["per_read"].count(b"\n") > 400
    assert orc["sam"] == ref["sam"]
    assert orc["per_read"] == ref["per_read"]
    assert orc["abbreviated"] == ref["abbreviated"]
    assert len(set(orc["tax_ids"])) > 4
    # the stamps the reference leaves in log.txt (src/sequenceTools.h:171-179): the counts are checkable facts
    # (the reference's Log is a function-static that opens ./log.txt once per process: only the first run of a process
    # finds the file in its own directory)
    if ref["log"] is not None:
        log = ref["log"].decode()
        assert log.count("Aligning reads to database using k = 32") == -(-500 // per_batch)
        assert "Processed\t500\t reads" in log


def test_oracle_chain_equals_the_real_reference_loop_single_end_and_just_align(kslam, oracle, synth, tmp_path):
    _need_ref(oracle)
    D = importlib.import_module("kslam_amd.db")
    case = R.make_case(synth, n_pairs=400, seed=4202, paired=False)
    dbdir = R.write_case(case, tmp_path, D)
    ref = R.run_reference(oracle, case, tmp_path, dbdir, 150, command_line=b"SLAM --db db R1.fq")
    orc = R.run_oracle_chain(oracle, case, 150, command_line=b"SLAM --db db R1.fq")
    assert ref["sam"].count(b"\n") > 300
    assert orc["sam"] == ref["sam"] and orc["per_read"] == ref["per_read"] and orc["abbreviated"] == ref["abbreviated"]
    # --just-align: SAM only, no report files (src/SLAM.h:240-242, 252-255)
    case2 = R.make_case(synth, n_pairs=300, seed=4303)
    t2 = tmp_path / "ja"
    t2.mkdir()
    dbdir2 = R.write_case(case2, t2, D)
    ref2 = R.run_reference(oracle, case2, t2, dbdir2, 300, just_align=True)
    assert ref2["xml"] is None and ref2["per_read"] is None and ref2["abbreviated"] is None
    assert R.run_oracle_chain(oracle, case2, 300)["sam"] == ref2["sam"]


@pytest.mark.parametrize("per_batch,pseudo", [(200, True), (500, False)])
def test_product_host_stages_equal_the_real_reference_loop(kslam, oracle, synth, tmp_path, per_batch, pseudo):
    """The product's host tail, per-read LCA, report writer and SAM header (host-only entry points of libkslam_hip.so,
    fed with the oracle's alignments because there is no GPU here) against the files the reference wrote."""
    _need_ref(oracle)
    D = importlib.import_module("kslam_amd.db")
    T = importlib.import_module("kslam_amd.tail")
    X = importlib.import_module("kslam_amd.taxonomy")
    case = R.make_case(synth, n_pairs=500, seed=4404)
    dbdir = R.write_case(case, tmp_path, D)
    ref = R.run_reference(oracle, case, tmp_path, dbdir, per_batch, pseudo=pseudo)
    db = D.Database.load(os.path.join(dbdir, "database"))
    tax = X.TaxDB(case["taxdb"])
    report = X.Report()
    P = T.TailParams.default(pseudo_assembly=pseudo)
    n = case["n_pairs"]
    rb, quals, ids = case["bases"], case["quals"], case["ids"]
    gb = [e["bases"] for e in case["entries"]]
    sam, per_read, tax_ids = [T.sam_header(db, b"SLAM --db db R1.fq R2.fq")], [], []
    for lo in range(0, n, per_batch):
        hi = min(n, lo + per_batch)
        b_reads, b_quals, b_ids = rb[lo:hi] + rb[n + lo:n + hi], quals[lo:hi] + quals[n + lo:n + hi], ids[lo:hi] * 2
        al, cig, _ = oracle.align_to_database(b_reads, gb)
        reads = T.Reads(b_reads, b_quals, b_ids)
        al = al.astype(kslam.OVERLAP_DT) if hasattr(kslam, "OVERLAP_DT") and al.dtype != kslam.OVERLAP_DT else al
        sam.append(T.tail_sam(P, reads, db, al, cig)[0])
        rp, pr, _ = T.tail_pairs(P, reads, al)
        t, text = tax.classify(P, reads, db, rp, pr)
        report.add_batch(reads, db, rp, pr, t)
        per_read.append(text)
        tax_ids += t.tolist()
    assert b"".join(sam) == ref["sam"]
    assert b"".join(per_read) == ref["per_read"]
    assert tax.summary(np.asarray(tax_ids, dtype=np.uint32), n) == ref["abbreviated"]
    assert tax.report_xml(report, db, db.gene_extras(), n) == ref["xml"]
    report.close()
    tax.close()
    db.close()


def test_golden_slam_loop_fixture_is_what_the_reference_writes(kslam, oracle, synth, tmp_path):
    """tests/golden/slam_loop.npz (inputs + the reference's output files) regenerated here must be identical: the
    fixture the -m gpu test replays is the real reference's answer, not the restatement's."""
    _need_ref(oracle)
    D = importlib.import_module("kslam_amd.db")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "slam_loop.npz")
    z = np.load(path, allow_pickle=False)
    for tag in ("a", "b"):
        case = load_fixture_case(z, tag)
        t = tmp_path / tag
        t.mkdir()
        dbdir = R.write_case(case, t, D)
        ref = R.run_reference(oracle, case, t, dbdir, int(z[tag + "_per_batch"]), pseudo=bool(z[tag + "_pseudo"]))
        for k in ("sam", "xml", "abbreviated", "per_read"):
            assert ref[k] == z[tag + "_" + k].tobytes(), (tag, k)


def load_fixture_case(z, tag):
    """the `case` dict of ref_loop_case.make_case out of slam_loop.npz"""
    import json
    meta = json.loads(z[tag + "_meta"].tobytes().decode())

    def col(name):
        flat, off = z["%s_%s" % (tag, name)], z["%s_%s_off" % (tag, name)]
        return [flat[int(off[i]):int(off[i + 1])].tobytes() for i in range(len(off) - 1)]
    bases, genes = col("entry_bases"), meta["genes"]
    entries = [{"bases": bases[i], "taxonomyID": e["taxonomyID"], "genbankID": e["genbankID"],
                "locusTag": e["locusTag"].encode(), "isPlasmid": e["isPlasmid"],
                "genes": [{k: (v.encode() if isinstance(v, str) else v) for k, v in g.items()} for g in genes[i]]}
               for i, e in enumerate(meta["entries"])]
    n = meta["n_pairs"]
    return {"entries": entries, "taxdb": z[tag + "_taxdb"].tobytes(), "bases": col("read_bases"), "quals": col("read_quals"),
            "ids": col("read_ids"), "n_pairs": n, "r1": z[tag + "_r1"].tobytes(), "r2": z[tag + "_r2"].tobytes()}


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_chain_equals_the_golden_slam_loop(kslam, oracle, tag):
    """Runs anywhere: the restated chain against the files the real reference wrote (tests/golden/slam_loop.npz)."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "slam_loop.npz"), allow_pickle=False)
    case = load_fixture_case(z, tag)
    orc = R.run_oracle_chain(oracle, case, int(z[tag + "_per_batch"]), pseudo=bool(z[tag + "_pseudo"]))
    assert orc["sam"] == z[tag + "_sam"].tobytes()
    assert orc["per_read"] == z[tag + "_per_read"].tobytes()
    assert orc["abbreviated"] == z[tag + "_abbreviated"].tobytes()


def test_c1_golden_fixture_is_what_the_reference_writes(kslam, oracle, synth, tmp_path):
    """tests/golden/c1_golden.json regenerated here (seed 1, pseudo-assembly on) must be identical: the fixture is the real
    reference's output (oracle/_ref/libslam_ref.so), not a self-made file -- and the ORACLE chain writes the same SAM,
    _PerRead and _abbreviated at this size too (10 k pairs x 150 bp vs 3 x 2 Mb)."""
    import json
    import ref_loop_case as R
    D = importlib.import_module("kslam_amd.db")
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_golden.json")))["cases"]["seed1_pseudo"]
    case = R.make_case_c1(synth, 1)
    assert R.digest_of_inputs(case) == gold["inputs_md5"]
    chain = R.run_oracle_chain(oracle, case, 10_000_000, pseudo=True)
    import hashlib
    for k in ("sam", "per_read", "abbreviated"):
        assert hashlib.md5(chain[k]).hexdigest() == gold["md5"][k], k
    # the PRODUCT's host stages on the oracle's alignments (no GPU here): all four files, the XML report included -- whose
    # first taxon loses the read std::sort leaves in front of the reference's vector (no read pair is unclassified here:
    # combineTaxonomies' quirk, src/MetagenomicResults.h:159-175; kslam_gnu::front_after_sort)
    T = importlib.import_module("kslam_amd.tail")
    X = importlib.import_module("kslam_amd.taxonomy")
    dbdir = R.write_case(case, tmp_path, D)
    db = D.Database.load(os.path.join(dbdir, "database"))
    tax, report = X.TaxDB(case["taxdb"]), X.Report()
    P = T.TailParams.default(pseudo_assembly=True)
    n = case["n_pairs"]
    al, cig, _ = oracle.align_to_database(case["bases"], [e["bases"] for e in case["entries"]])
    al = al.astype(kslam.OVERLAP_DT) if al.dtype != kslam.OVERLAP_DT else al
    reads = T.Reads(case["bases"], case["quals"], case["ids"] * 2)
    sam = T.sam_header(db, b"SLAM --db db R1.fq R2.fq") + T.tail_sam(P, reads, db, al, cig)[0]
    rp, pr, _ = T.tail_pairs(P, reads, al)
    t, text = tax.classify(P, reads, db, rp, pr)
    report.add_batch(reads, db, rp, pr, t)
    got = {"sam": sam, "per_read": text, "abbreviated": tax.summary(np.asarray(t, dtype=np.uint32), n),
           "xml": tax.report_xml(report, db, db.gene_extras(), n)}
    for k in ("sam", "per_read", "abbreviated", "xml"):
        assert hashlib.md5(got[k]).hexdigest() == gold["md5"][k], k
    report.close()
    tax.close()
    db.close()
    if oracle.have_ref_slam():
        ref = R.run_reference(oracle, case, tmp_path, dbdir, 10_000_000, pseudo=True)
        d = R.digest_of_outputs(ref)
        assert {k: d[k] for k in ("md5", "bytes", "sam_lines", "sam_head", "sam_tail")} == {k: gold[k] for k in ("md5", "bytes", "sam_lines", "sam_head", "sam_tail")}

