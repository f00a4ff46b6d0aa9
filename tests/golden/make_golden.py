#!/usr/bin/env python3
"""Generates the committed golden fixtures (run in the build container, where
/root/reference exists and oracle/_ref has been built from it).

 * survey_vectors.json : reference answers recorded in SURVEY.md section 8c (k-mer records of
   a 46-base read / genome, `TAG` at K=3, four Aligner::Align tuples).  Written verbatim, and
   re-checked here against the real reference pieces that can be built (KMer.h, ssw.c).
 * ssw_vectors.npz     : 1500 random (read, ref) code pairs with the answers of the reference's
   own ssw_init + ssw_align (oracle/_ref/libssw_ref.so), scoring (2,3,5,2) and (1,4,6,1).
 * kmer_vectors.npz    : sequences + the records produced by the reference's own
   getKMers_parallel and sortKMers (oracle/_ref/libkmer_ref.so).
 * align_small.npz     : a seeded 150-pair x 6-genome data set with the expected
   alignToDatabase output; produced by the oracle with its SSW core routed through the
   reference's ssw.c AND by the oracle's own restatement, asserted identical.
 * join_vectors.npz    : reads + genomes (tests/join_cases.py) with the answers of the reference's OWN
   sortKMers, findOverlaps (raw list), findOverlaps_parallel (deduped list) and alignToDatabase
   (oracle/_ref/libjoin_ref.so, one OpenMP thread), plus the (read, entry, rel) keys whose revComp is a tie.
 * align_vectors.npz   : ASCII (query, ref, ref_len) triples with the answers of the reference's OWN
   Aligner::Align at three filter settings.
 * slam_loop.npz       : two small FASTQ pairs + databases + taxDB with the files the reference's OWN batch
   loop metagenomicAnalysis_Low_Mem wrote for them (oracle/_ref/libslam_ref.so): SAM, XML report,
   _abbreviated, _PerRead.
Fixtures are data (inputs + expected outputs); no reference source text is stored.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
from conftest import load_kslam  # noqa: E402


def main():
    assert O.have_ref_ssw() and O.have_ref_kmer(), "needs oracle/_ref (reference present)"
    # ---- SURVEY 8c vectors ----
    q = "ACGTTGCAAGGCTTAACCGGTTACGATCGATCGGATCCAGT"
    sv = {
        "read46": "ACGTTGCAAGGCTTAACCGGTTACGATCGATCGGATCCAGTNACGT",
        "read_id": 7, "read_records": [[0x1eb43da05fa1c9c9, 0x00000007, 0], [0x72727817e835ad07, 0x40000007, 13]],
        "genome_id": 5, "genome_gap": 4, "genome_meta_off": [[0x80000005, 0], [0xc0000005, 4], [0x80000005, 8], [0x80000005, 12]],
        "tag_k3": {"seq": "TAG", "fwd": 35, "rc": 24},
        "align": [
            {"query": q, "ref": q, "ref_len": 41, "score": 82, "ref_b": 0, "ref_e": 40, "q_b": 0, "q_e": 40, "cigar": "41M"},
            {"query": q, "ref": "ACGTTGCAAGGCTTAACCGGTTTTACGATCGATCGGATCCAG", "ref_len": 41, "score": 71,
             "ref_b": 0, "ref_e": 40, "q_b": 0, "q_e": 38, "cigar": "20M2D19M"},
            {"query": "ACGTNNGCAAGGCTTAACCGGTTACGATCGATCGGATCCAGT", "ref": q, "ref_len": 41, "score": 75,
             "q_b": 0, "q_e": 41, "cigar": "4M1I37M"},
        ],
        "scoring": [2, 3, 5, 2],
    }
    assert O.ref_kmer3(b"TAG") == (35, 24)
    r = O.ref_extract_kmers([b""] * 7 + [sv["read46"].encode()], False, 1)
    assert [int(r[0]["kmer"]), int(r[0]["meta"]), int(r[0]["offset"])] == sv["read_records"][0]
    json.dump(sv, open(os.path.join(HERE, "survey_vectors.json"), "w"), indent=1)

    # ---- SSW vectors from the real ssw.c ----
    rng = np.random.default_rng(12345)
    cases = []
    for params in ((2, 3, 5, 2), (1, 4, 6, 1)):
        mat = O.build_matrix(params[0], params[1])
        for _ in range(750):
            L = int(rng.integers(20, 256))
            ref = rng.integers(0, 4, L).astype(np.int8)
            rd = ref.copy()
            k = rng.integers(0, max(1, L // 10))
            rd[rng.integers(0, L, k)] = rng.integers(0, 4, k)
            if rng.random() < 0.5:  # indel
                p = int(rng.integers(1, L - 1)); n = int(rng.integers(1, 4))
                rd = np.concatenate([rd[:p], rd[p + n:]]) if rng.random() < 0.5 else \
                    np.concatenate([rd[:p], rng.integers(0, 4, n).astype(np.int8), rd[p:]])
            if rng.random() < 0.3:
                rd = np.concatenate([rng.integers(0, 4, rng.integers(1, 15)).astype(np.int8), rd])
            if rng.random() < 0.2:
                rd[rng.integers(0, len(rd))] = 4
            if rng.random() < 0.2:
                ref = ref[:int(rng.integers(L // 2, L + 1))]
            res, cig = O.ref_ssw_align(rd, ref, mat, params[2], params[3])
            cases.append((params, rd, ref, res, cig))
    np.savez_compressed(
        os.path.join(HERE, "ssw_vectors.npz"),
        params=np.array([c[0] for c in cases], dtype=np.int32),
        reads=np.concatenate([c[1] for c in cases]), read_len=np.array([len(c[1]) for c in cases]),
        refs=np.concatenate([c[2] for c in cases]), ref_len=np.array([len(c[2]) for c in cases]),
        results=np.array([c[3] for c in cases], dtype=np.int32),
        cigars=np.concatenate([c[4] for c in cases]), cigar_len=np.array([len(c[4]) for c in cases]))

    # ---- k-mer vectors from the real KMer.h ----
    B = np.frombuffer(b"ACGTNacgt", dtype=np.uint8)
    seqs = [B[rng.choice(9, int(rng.integers(0, 300)), p=[.24, .24, .24, .24, .01, .0075, .0075, .0075, .0075])].tobytes()
            for _ in range(60)] + [b"A" * 40, b"ACGT" * 10, b"T" * 33]
    rr = O.ref_extract_kmers(seqs, False, 1)
    rg = O.ref_extract_kmers(seqs, True, 16)
    allr = np.concatenate([rr, rg])
    srt = O.ref_sort_kmers(allr)
    np.savez_compressed(os.path.join(HERE, "kmer_vectors.npz"),
                        seqs=np.frombuffer(b"".join(seqs), dtype=np.uint8),
                        seq_len=np.array([len(s) for s in seqs]), reads_gap1=rr, genbank_gap16=rg,
                        sorted_kmer=srt["kmer"], sorted_meta=srt["meta"])

    # ---- small end-to-end alignToDatabase fixture ----
    K = load_kslam()
    import importlib
    synth = importlib.import_module("kslam_amd.synth")
    genomes = synth.make_genomes(41, 3, 2, 12000, shared_segment=1500)
    reads, _ = synth.make_paired_reads(42, genomes, 150, edge_frac=0.15, n_rate=0.003, indel_rate=0.004)
    reads, genomes = synth.to_bytes(reads), synth.to_bytes(genomes)
    a1, c1, _ = O.align_to_database(reads, genomes)
    assert O.use_reference_ssw(True)
    a2, c2, _ = O.align_to_database(reads, genomes)
    O.use_reference_ssw(False)
    assert (a1 == a2).all() and np.array_equal(c1, c2), "oracle restatement != reference ssw core"
    np.savez_compressed(os.path.join(HERE, "align_small.npz"),
                        reads=np.frombuffer(b"".join(reads), dtype=np.uint8),
                        read_len=np.array([len(s) for s in reads]),
                        genomes=np.frombuffer(b"".join(genomes), dtype=np.uint8),
                        genome_len=np.array([len(s) for s in genomes]), alignments=a1, cigars=c1)
    print("fixtures written:", sorted(f for f in os.listdir(HERE) if f.endswith((".npz", ".json"))))


def make_fastq_cases():
    """Inputs + the records the reference's own reader (oracle/_ref/libfastq_ref.so, built from
    src/FASTQsequence.h where it lies) returns for them -> tests/golden/fastq_cases.json."""
    import base64
    import tempfile
    import numpy as np
    import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_fastq import make_text
    assert O.have_ref_fastq(), "needs /root/reference (make -C oracle ref)"
    named = [
        ("plain", b"@r1\nACGT\n+\nIIII\n@r2\nGG\n+\nII\n"),
        ("crlf and lone cr", b"@r1 d/1\r\nACGT\r\n+\r\nIIII\r@r2/2\rAC\r+\rII\r"),
        ("no final newline", b"@r1\nACGT\n+\nIIII"),
        ("missing quality line", b"@r1\nACGT\n+\n"),
        ("identifier rules", b"@ x\nA\n+\nI\n@\nC\n+\nI\n\nG\n+\nI\n@a/b/c d/e\nT\n+\nI\n@/lead\nA\n+\nI\n"),
        ("blank lines inside", b"@r1\n\n+\n\n@r2\nAC\n+\nII\n\n\n"),
        ("truncated record", b"@r1\nACGT\n+\nIIII\n@r2\nAC\n"),
        ("empty", b""),
    ]
    rng = np.random.default_rng(2024)
    for k in range(6):
        named.append(("random %d" % k, make_text(rng, 25, truncate=k % 2 == 1, blank_tail=k % 3)))
    cases = []
    with tempfile.TemporaryDirectory() as d:
        for name, text in named:
            path = os.path.join(d, "c.fq")
            open(path, "wb").write(text)
            b, q, i, _ = O.ref_fastq_read(path)
            cases.append({"name": name, "text_b64": base64.b64encode(text).decode(),
                          "records": [[base64.b64encode(x).decode() for x in r] for r in zip(i, b, q)]})
    json.dump({"source": "reference src/FASTQsequence.h getSequencesFromFASTQFile via oracle/ref_fastq_driver.cpp",
               "cases": cases}, open(os.path.join(HERE, "fastq_cases.json"), "w"), indent=0)
    print("fastq_cases.json:", len(cases), "cases")


def make_taxonomy_cases():
    """A taxDB text + the REAL reference TaxonomyDB's answers (oracle/_ref/libtaxonomy_ref.so, built
    from src/TaxonomyDatabase.h where it lies) -> tests/golden/taxonomy_cases.json."""
    import tempfile
    import numpy as np
    import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_taxonomy import make_tree, query_sets
    assert O.have_ref_taxonomy(), "needs /root/reference (make -C oracle ref)"
    rng = np.random.default_rng(77)
    text, ids = make_tree(rng, 120)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "taxDB")
        open(path, "wb").write(text)
        ref = O.ref_taxonomy_tree(path)
        lca = [[s, ref.lca(s)] for s in query_sets(rng, ids, 300)]
        nodes = [[i, ref.parent(i), ref.text(i, 0).decode(), ref.text(i, 1).decode(), ref.text(i, 2).decode(),
                  ref.is_subspecies(i), ref.at_rank(i, b"species")]
                 for i in list(dict.fromkeys(ids))[:120] + [0, 1, 999999]]
        ref.close()
    json.dump({"source": "reference src/TaxonomyDatabase.h via oracle/ref_taxonomy_driver.cpp",
               "taxdb": text.decode(), "lca": lca, "nodes": nodes},
              open(os.path.join(HERE, "taxonomy_cases.json"), "w"), indent=0)
    print("taxonomy_cases.json:", len(lca), "lca queries,", len(nodes), "nodes")


def _cols(items):
    off = np.zeros(len(items) + 1, dtype=np.uint64)
    np.cumsum([len(x) for x in items], out=off[1:])
    return np.frombuffer(b"".join(items), dtype=np.uint8), off


def make_join_and_align_vectors():
    import oracle as O
    from join_cases import make_join_case, make_align_cases, revcomp_tie_rows
    assert O.have_ref_join(), "needs /root/reference (make -C oracle ref)"
    reads, genomes = make_join_case(20240, n_reads=220)
    recs = np.concatenate([O.ref_extract_kmers(reads, False, 1), O.ref_extract_kmers(genomes, True, 16)])
    srt = O.ref_sort_kmers(recs)
    lens = [len(r) for r in reads]
    dedup, raw = O.ref_find_overlaps(srt, lens, want_raw=True)
    ties = np.array(sorted(revcomp_tie_rows(raw)), dtype=np.int64).reshape(-1, 3)
    out = {"sorted": srt, "raw": raw, "deduped": dedup, "ties": ties}
    out["reads"], out["reads_off"] = _cols(reads)
    out["genomes"], out["genomes_off"] = _cols(genomes)
    for thr in (0, 150):
        al, cg = O.ref_align_to_database(reads, genomes, O.Params.default(score_threshold=thr))
        assert (al[["read", "entry", "rel"]] == dedup[["read", "entry", "rel"]]).all()
        out["alignments_thr%d" % thr], out["cigars_thr%d" % thr] = al, cg
    al, cg = O.ref_align_to_database(reads, genomes, O.Params.default(report_cigar=False))
    assert len(cg) == 0
    out["alignments_nocigar"] = al
    np.savez_compressed(os.path.join(HERE, "join_vectors.npz"), **out)
    print("join_vectors.npz: %d records, %d raw, %d deduped, %d revComp ties" % (len(srt), len(raw), len(dedup), len(ties)))

    out = {}
    # two scorings inside the envelope, three outside it (the striped evaluation order decides there: src/ssw.c:274-305, 512-526)
    param_sets = ((2, 3, 5, 2), (1, 4, 6, 1), (2, 9, 5, 2), (5, 4, 10, 10), (2, 8, 2, 3))
    out["param_sets"] = np.array(param_sets, dtype=np.int32)
    for params in param_sets:
        cases = make_align_cases(777 + params[0] + 10 * params[1], 250 if params in param_sets[:2] else 120)
        tag = "p%d%d%d%d" % params
        out[tag + "_query"], out[tag + "_query_off"] = _cols([c[0] for c in cases])
        out[tag + "_ref"], out[tag + "_ref_off"] = _cols([c[1] for c in cases])
        out[tag + "_ref_len"] = np.array([c[2] for c in cases], dtype=np.int32)
        for thr, want in ((0, 1), (120, 1), (0, 0)):
            p = O.Params.default(report_cigar=bool(want), score_threshold=thr, match=params[0], mismatch=params[1],
                                 gap_open=params[2], gap_extend=params[3])
            res, cigs = [], []
            for q, r, n in cases:
                a, c = O.ref_aligner_align(q, r, p, ref_len=n)
                res.append(a)
                cigs.append(c)
            k = "%s_thr%d_cigar%d" % (tag, thr, want)
            out[k + "_results"] = np.array(res, dtype=np.int32)
            out[k + "_cigars"] = np.concatenate(cigs) if cigs else np.zeros(0, np.uint32)
            out[k + "_cigar_len"] = np.array([len(c) for c in cigs], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "align_vectors.npz"), **out)
    print("align_vectors.npz: %d scorings x 3 filter settings" % len(param_sets))


def make_slam_loop():
    import importlib
    import tempfile
    import oracle as O
    import ref_loop_case as R
    assert O.have_ref_slam(), "needs /root/reference (make -C oracle ref)"
    load_kslam()
    synth = importlib.import_module("kslam_amd.synth")
    D = importlib.import_module("kslam_amd.db")
    out = {}
    for tag, seed, n_pairs, per_batch, pseudo in (("a", 5101, 420, 150, True), ("b", 5202, 300, 300, False)):
        case = R.make_case(synth, n_pairs=n_pairs, seed=seed, genome_len=9000, read_len=100)
        with tempfile.TemporaryDirectory() as t:
            dbdir = R.write_case(case, t, D)
            ref = R.run_reference(O, case, t, dbdir, per_batch, pseudo=pseudo)
        for k in ("sam", "xml", "abbreviated", "per_read"):
            out[tag + "_" + k] = np.frombuffer(ref[k], dtype=np.uint8)
        out[tag + "_per_batch"], out[tag + "_pseudo"] = np.int64(per_batch), np.int64(pseudo)
        for k in ("taxdb", "r1", "r2"):
            out[tag + "_" + k] = np.frombuffer(case[k], dtype=np.uint8)
        for name, items in (("entry_bases", [e["bases"] for e in case["entries"]]), ("read_bases", case["bases"]),
                            ("read_quals", case["quals"]), ("read_ids", case["ids"])):
            out["%s_%s" % (tag, name)], out["%s_%s_off" % (tag, name)] = _cols(items)
        meta = {"n_pairs": n_pairs,
                "entries": [{"taxonomyID": e["taxonomyID"], "genbankID": e["genbankID"], "locusTag": e["locusTag"].decode(),
                             "isPlasmid": bool(e["isPlasmid"])} for e in case["entries"]],
                "genes": [[{k: (v.decode() if isinstance(v, bytes) else v) for k, v in g.items()} for g in e["genes"]]
                          for e in case["entries"]]}
        out[tag + "_meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        print("slam_loop %s: %d pairs, %d SAM bytes, %d _PerRead lines" % (tag, n_pairs, len(ref["sam"]), ref["per_read"].count(b"\n")))
    np.savez_compressed(os.path.join(HERE, "slam_loop.npz"), **out)


def make_c1_golden():
    """SURVEY 8c golden (5): the C1-size end-to-end run -- 10 k pairs x 150 bp vs 3 x 2 Mb, genome 2 carrying a 20 kb copy of
    genome 0 -- through the reference's OWN loop (oracle/_ref/libslam_ref.so, one thread: the XML report's order of equal
    counts depends on the thread count), seeds 1 / 2, with and without pseudo-assembly.  The inputs are regenerated from
    the seeds wherever the fixture is replayed (kslam_amd.synth, numpy only); their md5s are recorded so that a drifting
    generator shows up as that, not as a product failure.  Kept of the outputs: md5 + size of the four files, the first /
    last 200 SAM lines."""
    import importlib
    import tempfile
    import oracle as O
    import ref_loop_case as R
    assert O.have_ref_slam(), "needs /root/reference (make -C oracle ref)"
    load_kslam()
    synth = importlib.import_module("kslam_amd.synth")
    D = importlib.import_module("kslam_amd.db")
    out = {"what": "tests/ref_loop_case.py make_case_c1(seed) through the reference's metagenomicAnalysis_Low_Mem (src/SLAM.h:159-268), "
                   "command line 'SLAM --db db R1.fq R2.fq', one batch, one thread", "cases": {}}
    for seed in (1, 2):
        case = R.make_case_c1(synth, seed)
        for pseudo in (True, False):
            with tempfile.TemporaryDirectory() as t:
                dbdir = R.write_case(case, t, D)
                ref = R.run_reference(O, case, t, dbdir, 10_000_000, pseudo=pseudo)
            d = R.digest_of_outputs(ref)
            d["inputs_md5"] = R.digest_of_inputs(case)
            out["cases"]["seed%d_%s" % (seed, "pseudo" if pseudo else "nopseudo")] = d
            print("c1_golden seed %d pseudo %d: %d SAM lines, md5 %s" % (seed, pseudo, d["sam_lines"], d["md5"]["sam"]))
    json.dump(out, open(os.path.join(HERE, "c1_golden.json"), "w"), indent=0)


if __name__ == "__main__":
    if "--c1" in sys.argv:
        make_c1_golden()
        sys.exit(0)
    if "--pins" in sys.argv:           # only the fixtures recorded from libjoin_ref.so / libslam_ref.so
        make_join_and_align_vectors()
        make_slam_loop()
        make_c1_golden()
        sys.exit(0)
    if "--taxonomy" in sys.argv:
        make_taxonomy_cases()
        sys.exit(0)
    if "--fastq" not in sys.argv:      # --fastq: only (re)write fastq_cases.json
        main()
    make_fastq_cases()
    make_taxonomy_cases()
    make_join_and_align_vectors()
    make_slam_loop()
    make_c1_golden()
