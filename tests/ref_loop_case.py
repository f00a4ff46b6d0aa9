"""A small FASTQ pair + database + taxDB on disk, the ORACLE chain over it, and the REAL reference loop over it.

Test data plumbing shared by tests/test_reference_loop.py (CPU: oracle chain == the reference's own
metagenomicAnalysis_Low_Mem compiled in place, oracle/_ref/libslam_ref.so), tests/golden/make_golden.py (records the
real reference's output files as tests/golden/slam_loop.npz) and the -m gpu test that replays that fixture through the
C ABI.  Nothing here is product code.
"""
import importlib
import os

import numpy as np


def taxdb_text(n_species, n_strains):
    """root 1 -> 2 (Bacteria) -> genus 10+g -> species 100+s -> strain 1000+e; four lines per node
    (src/TaxonomyDatabase.h:153-183)"""
    recs = [(1, 1, b"root", b"no rank"), (2, 1, b"Bacteria", b"superkingdom")]
    for s in range(n_species):
        g = 10 + s // 2
        if s % 2 == 0:
            recs.append((g, 2, b"Genus%d" % g, b"genus"))
        recs.append((100 + s, g, b"Genus%d species%d" % (g, s), b"species"))
        for k in range(n_strains):
            recs.append((1000 + s * n_strains + k, 100 + s, b"strain %d.%d" % (s, k), b"strain"))
    return b"".join(b"%d\n%d\n%s\n%s\n" % r for r in recs)


def fastq_text(bases, quals, ids, mate, eol=b"\n"):
    return b"".join(b"@" + i + b"/%d extra words" % mate + eol + b + eol + b"+" + eol + q + eol
                    for b, q, i in zip(bases, quals, ids))


def make_case(synth, n_pairs=500, seed=4101, read_len=110, genome_len=14000, n_species=3, n_strains=3, paired=True):
    """-> dict(entries, taxdb, r1, r2, bases, quals, ids, n_pairs): entries in the layout kslam_amd.db.write takes"""
    rng = np.random.default_rng(seed)
    genomes = synth.make_genomes(seed % 1000, n_species, n_strains, genome_len, strain_sub=0.02, strain_indel=0.001,
                                 shared_segment=1800)
    reads, _ = synth.make_paired_reads(seed % 1000 + 1, genomes, n_pairs, read_len=read_len, frag_mean=300, frag_sd=45,
                                       sub_rate=0.015, indel_rate=0.003, n_rate=0.001, edge_frac=0.05, unmapped_frac=0.05)
    gb = synth.to_bytes(genomes)
    entries = [{"bases": g, "taxonomyID": 1000 + i if i != 5 else 0, "genbankID": 7000 + i,
                "locusTag": b"NC_%06d.1" % i, "isPlasmid": i % 4 == 3,
                # locus tags make sortResults' key (count, cdsStart, locusTag; src/MetagenomicResults.h:262-271) total
                "genes": [{"geneName": b"gene%d" % k, "proteinID": b"WP_%d.1" % (100 * i + k), "locusTag": b"LT%02d_%02d" % (i, k),
                           "referenceSequence": b"NC_%06d" % i, "product": b"hypothetical protein %d" % k,
                           "start": 400 + 1300 * k, "stop": 1500 + 1300 * k, "geneID": k, "complement": bool(k & 1)}
                          for k in range(9)]}
               for i, g in enumerate(gb)]
    rb = synth.to_bytes(reads)
    quals = [bytes(rng.integers(35, 74, len(b), dtype=np.uint8)) for b in rb]
    ids = [b"frag%05d" % i for i in range(n_pairs)]
    case = {"entries": entries, "taxdb": taxdb_text(n_species, n_strains), "bases": rb, "quals": quals, "ids": ids,
            "n_pairs": n_pairs, "r1": fastq_text(rb[:n_pairs], quals[:n_pairs], ids, 1),
            "r2": fastq_text(rb[n_pairs:], quals[n_pairs:], ids, 2) if paired else b""}
    if not paired:
        case["bases"], case["quals"] = rb[:n_pairs], quals[:n_pairs]
    return case


def make_case_c1(synth, seed=1, n_pairs=10000):
    """BASELINE configs[0] / SURVEY 8c golden (5): n_pairs x 2 x 150 bp vs 3 genomes of 2 Mb, genome 2 carrying a 20 kb copy of
    genome 0 (reads from it hit two entries), fragments ~ N(350, 30), 1 % substitutions, 0.1 % indels, 2 % of the pairs from
    no genome of the database, qualities constant 'I'.  Same dict layout as make_case."""
    genomes = synth.make_genomes(seed, 3, 1, 2_000_000, shared_segment=20000)
    reads, _ = synth.make_paired_reads(seed + 1000, genomes, n_pairs, read_len=150, frag_mean=350, frag_sd=30, sub_rate=0.01,
                                       indel_rate=0.001, unmapped_frac=0.02)
    gb = synth.to_bytes(genomes)
    entries = [{"bases": g, "taxonomyID": 1000 + i, "genbankID": 7000 + i, "locusTag": b"NC_%06d.1" % i, "isPlasmid": False,
                "genes": [{"geneName": b"gene%d" % k, "proteinID": b"WP_%d.1" % (100 * i + k), "locusTag": b"LT%02d_%02d" % (i, k),
                           "referenceSequence": b"NC_%06d" % i, "product": b"hypothetical protein %d" % k,
                           "start": 400 + 130000 * k, "stop": 120000 + 130000 * k, "geneID": k, "complement": bool(k & 1)}
                          for k in range(15)]}
               for i, g in enumerate(gb)]
    rb = synth.to_bytes(reads)
    quals = [b"I" * len(b) for b in rb]
    ids = [b"p%d" % i for i in range(n_pairs)]
    return {"entries": entries, "taxdb": taxdb_text(3, 1), "bases": rb, "quals": quals, "ids": ids, "n_pairs": n_pairs,
            "r1": fastq_text(rb[:n_pairs], quals[:n_pairs], ids, 1), "r2": fastq_text(rb[n_pairs:], quals[n_pairs:], ids, 2)}


def digest_of_outputs(ref, lines=200):
    """what tests/golden/c1_golden.json keeps of a run's four files: md5 of each, and the first / last `lines` SAM lines"""
    import hashlib
    sam = ref["sam"].split(b"\n")
    if sam and sam[-1] == b"":
        sam.pop()
    return {"md5": {k: hashlib.md5(ref[k]).hexdigest() for k in ("sam", "per_read", "xml", "abbreviated")},
            "bytes": {k: len(ref[k]) for k in ("sam", "per_read", "xml", "abbreviated")},
            "sam_lines": len(sam), "sam_head": [x.decode() for x in sam[:lines]], "sam_tail": [x.decode() for x in sam[-lines:]]}


def digest_of_inputs(case):
    import hashlib
    return {"r1": hashlib.md5(case["r1"]).hexdigest(), "r2": hashlib.md5(case["r2"]).hexdigest(),
            "genomes": hashlib.md5(b"".join(e["bases"] for e in case["entries"])).hexdigest(),
            "taxdb": hashlib.md5(case["taxdb"]).hexdigest()}


def write_case(case, tmp_path, db_module):
    """files the product reads: <tmp>/db/{database,taxDB}, <tmp>/R1.fq, <tmp>/R2.fq"""
    dbdir = os.path.join(str(tmp_path), "db")
    os.makedirs(dbdir, exist_ok=True)
    db_module.write(os.path.join(dbdir, "database"), case["entries"])
    open(os.path.join(dbdir, "taxDB"), "wb").write(case["taxdb"])
    open(os.path.join(str(tmp_path), "R1.fq"), "wb").write(case["r1"])
    if case["r2"]:
        open(os.path.join(str(tmp_path), "R2.fq"), "wb").write(case["r2"])
    return dbdir


def run_reference(oracle, case, tmp_path, dbdir, per_batch, pseudo=True, just_align=False, sam_xa=False,
                  num_alignments=10, score_threshold=0, command_line=b"SLAM --db db R1.fq R2.fq", threads=1, gpu_operator=False,
                  scoring=None):
    """The reference's own loop on the files.  -> dict(sam, xml, abbreviated, per_read, log): bytes of the files it wrote.
    gpu_operator: the same loop with alignToDatabase swapped for the GPU operator (oracle/_ref/libslam_gpu_ref.so)."""
    oracle.ref_slam_set_index([{
        "bases": e["bases"], "locus_tag": e["locusTag"], "taxonomy_id": e["taxonomyID"], "genbank_id": e["genbankID"],
        "genes": [{"name": g["geneName"], "locus_tag": g["locusTag"], "protein_id": g["proteinID"], "product": g["product"],
                   "reference": g["referenceSequence"], "gene_id": g["geneID"], "start": g["start"], "stop": g["stop"],
                   "complement": g["complement"]} for g in e["genes"]]} for e in case["entries"]], gpu=gpu_operator)
    p = oracle.RefSlamParams.default(pseudo_assembly=int(pseudo), just_align=int(just_align), sam_xa=int(sam_xa),
                                     num_reads_at_once=per_batch, num_sam_alignments=num_alignments,
                                     score_threshold=score_threshold, threads=threads)
    for k, v in (scoring or {}).items():      # match / mismatch / gap_open / gap_extend
        setattr(p, k, v)
    t = str(tmp_path)
    wd = os.path.join(t, "gpurun" if gpu_operator else "refrun")
    os.makedirs(wd, exist_ok=True)
    out = os.path.join(wd, "out")
    sam = os.path.join(wd, "out.sam")
    run = lambda: oracle.ref_slam_run(os.path.join(t, "R1.fq"), os.path.join(t, "R2.fq") if case["r2"] else "",
                                      dbdir, out, sam, p, command_line, workdir=wd, gpu=gpu_operator)
    if threads == 1:
        oracle.binding._one_thread(run)
    else:
        run()

    def rd(path):
        return open(path, "rb").read() if os.path.exists(path) else None
    return {"sam": rd(sam), "xml": rd(out), "abbreviated": rd(out + "_abbreviated"), "per_read": rd(out + "_PerRead"),
            "log": rd(os.path.join(wd, "log.txt"))}


def run_oracle_chain(oracle, case, per_batch, pseudo=True, sam_xa=False, num_alignments=10, score_threshold=0,
                     command_line=b"SLAM --db db R1.fq R2.fq"):
    """The restated chain, batch by batch with the reference's boundaries: oracle.align_to_database ->
    tail_oracle (pairing .. SAM) -> taxonomy_oracle (per-read LCA, abbreviated report).
    -> dict(sam, per_read, abbreviated, tax_ids, batches)"""
    T = importlib.import_module("kslam_amd.tail")          # ctypes structures only (views / params)
    ents = case["entries"]
    gb = [e["bases"] for e in ents]
    oI = T.Index(gb, locus_tags=[e["locusTag"] for e in ents], taxonomy_ids=[e["taxonomyID"] for e in ents],
                 genes=[[(g["start"], g["stop"], g["geneName"], g["proteinID"], g["product"]) for g in e["genes"]] for e in ents])
    paired = bool(case["r2"])
    P = T.TailParams.default(pseudo_assembly=pseudo, sam_xa=sam_xa, num_sam_alignments=num_alignments,
                             score_threshold=score_threshold, paired=paired)
    otree = oracle.taxonomy_tree(case["taxdb"])
    n = case["n_pairs"]
    rb, quals, ids = case["bases"], case["quals"], case["ids"]
    sam, per_read, tax, batches = [oracle.sam_header(oI.view, command_line)], [], [], []
    for lo in range(0, n, per_batch):
        hi = min(n, lo + per_batch)
        if paired:
            b_reads, b_quals, b_ids = rb[lo:hi] + rb[n + lo:n + hi], quals[lo:hi] + quals[n + lo:n + hi], ids[lo:hi] * 2
        else:
            b_reads, b_quals, b_ids = rb[lo:hi], quals[lo:hi], ids[lo:hi]
        eal, ecig, _ = oracle.align_to_database(b_reads, gb, oracle.Params.default(score_threshold=score_threshold))
        oR = T.Reads(b_reads, b_quals, b_ids)
        sam.append(oracle.tail_sam(P, oR.view, oI.view, eal, ecig))
        erp, epr = oracle.tail_pairs(P, oR.view, eal)
        t = [otree.lca([ents[int(e)]["taxonomyID"] for e in epr["entry"][int(g["first"]):int(g["first"]) + int(g["count"])]])
             for g in erp]
        tax += t
        per_read.append(b"".join(b"%s\t%d\n" % (b_ids[int(g["r1_read"])], x) for g, x in zip(erp, t)))
        batches.append((b_ids, erp, epr))
    out = {"sam": b"".join(sam), "per_read": b"".join(per_read), "tax_ids": tax, "batches": batches,
           "abbreviated": oracle.taxonomy_summary(otree, tax, n)}
    otree.close()
    return out
