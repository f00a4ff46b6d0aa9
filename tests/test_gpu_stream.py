"""The reference's batch loop, several batches long (metagenomicAnalysis_Low_Mem, src/SLAM.h:159-268):
one FASTQ pair streamed in batches of `--num-reads-at-once` pairs through
  kslam_fastq_batch_end -> kslam_submit_batch_fastq_text (GPU: FASTQ index, alignment, pairing, insert-size
  statistics, screens, pseudo-assembly, per-row walk) -> kslam_tail_finish_write_rows -> kslam_write_fd
  -> kslam_tail_classify -> kslam_taxreport_add_batch
with one SAM file appended to and the taxonomy accumulated over the batches (src/SLAM.h:234-249), against
the oracle chain driven with the SAME batch boundaries (the insert-size limit is a per-batch statistic,
src/PairedOverlap.h:314-360, so the boundaries are part of the result)."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_case(synth, tmp_path, n_pairs, eol, seed=2031, long_reads=False):
    from test_gpu_end_to_end import _taxdb_text, _fastq_text
    D = importlib.import_module("kslam_amd.db")
    n_species, n_strains = 4, 3
    rng = np.random.default_rng(seed)
    genomes = synth.make_genomes(seed % 1000, n_species, n_strains, 30000, strain_sub=0.02, strain_indel=0.001,
                                 shared_segment=2500)
    reads, _ = synth.make_paired_reads(seed % 1000 + 1, genomes, n_pairs, read_len=120, frag_mean=320, frag_sd=40,
                                       sub_rate=0.015, indel_rate=0.003, n_rate=0.001, edge_frac=0.04, unmapped_frac=0.05)
    if long_reads:      # every 40th pair is a pair of 600-base reads (merged-pair-like): beyond the packed kernels' 511
        for k in range(0, n_pairs, 40):
            g = genomes[int(rng.integers(0, len(genomes)))]
            at = int(rng.integers(0, len(g) - 1300))
            frag = g[at:at + 1200]
            if rng.random() < 0.5:
                frag = synth.revcomp(frag)
            reads[k] = synth.mutate(rng, frag[:600], 0.015, 0.003)[:600]
            reads[n_pairs + k] = synth.mutate(rng, synth.revcomp(frag)[:600], 0.015, 0.003)[:600]
    gb = synth.to_bytes(genomes)
    entries = [{"bases": g, "taxonomyID": 1000 + i if i != 5 else 0, "genbankID": 7000 + i,
                "locusTag": b"NC_%06d.1" % i, "isPlasmid": i % 4 == 3,
                # gene locus tags make sortResults' key (count, cdsStart, locusTag; src/MetagenomicResults.h:262-271) total:
                # without them genes of different strains tie there and the order is std::sort's (unstable) choice
                "genes": [{"geneName": b"gene%d" % k, "proteinID": b"WP_%d.1" % (100 * i + k), "locusTag": b"LT%02d_%02d" % (i, k),
                           "referenceSequence": b"NC_%06d" % i,
                           "product": b"hypothetical protein %d" % k, "start": 500 + 1500 * k,
                           "stop": 1700 + 1500 * k, "geneID": k, "complement": bool(k & 1)} for k in range(15)]}
               for i, g in enumerate(gb)]
    dbdir = tmp_path / "db"
    dbdir.mkdir()
    D.write(dbdir / "database", entries)
    taxdb = _taxdb_text(n_species, n_strains)
    (dbdir / "taxDB").write_bytes(taxdb)
    rb = synth.to_bytes(reads)
    quals = [bytes(rng.integers(35, 74, len(b), dtype=np.uint8)) for b in rb]
    ids = [b"frag%05d" % i for i in range(n_pairs)]
    r1 = _fastq_text(rb[:n_pairs], quals[:n_pairs], ids, 1, eol)
    r2 = _fastq_text(rb[n_pairs:], quals[n_pairs:], ids, 2, eol)
    return dbdir, taxdb, rb, quals, ids, r1, r2


@pytest.mark.parametrize("eol,pseudo,per_batch,long_reads", [(b"\n", True, 700, False), (b"\r\n", False, 1000, False),
                                                             (b"\n", True, 2500, False), (b"\n", True, 833, False),
                                                             (b"\n", True, 900, True)])
def test_stream_of_batches_equals_the_reference_loop(kslam, oracle, synth, tmp_path, eol, pseudo, per_batch, long_reads):
    import ctypes as C
    D = importlib.import_module("kslam_amd.db")
    T = importlib.import_module("kslam_amd.tail")
    X = importlib.import_module("kslam_amd.taxonomy")
    S = importlib.import_module("kslam_amd.stream")
    dbo = importlib.import_module("oracle.db_oracle")
    n_pairs = 2500
    dbdir, taxdb, rb, quals, ids, r1, r2 = _make_case(synth, tmp_path, n_pairs, eol, long_reads=long_reads)

    # ---- product: files -> files ----
    db = D.Database.load(dbdir / "database")
    ctx = kslam.Context()
    bases_pp, lens_p = db.entry_pointers()
    ctx._chk(ctx._L.kslam_set_index(ctx._h, db.n_entries, C.cast(bases_pp, C.c_void_p), C.cast(lens_p, C.c_void_p)))
    tax = X.TaxDB((dbdir / "taxDB").read_bytes())
    report = X.Report()
    h1, h2 = kslam.HostBuffer(len(r1) + 64), kslam.HostBuffer(len(r2) + 64)
    h1.a[:len(r1)] = np.frombuffer(r1, dtype=np.uint8)
    h2.a[:len(r2)] = np.frombuffer(r2, dtype=np.uint8)
    P = T.TailParams.default(pseudo_assembly=pseudo)
    header = T.sam_header(db, b"SLAM --db db R1.fq R2.fq")
    sam_path, per_read_path = str(tmp_path / "out.sam"), str(tmp_path / "out_PerRead")
    sam_fd = os.open(sam_path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    pr_fd = os.open(per_read_path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    res = S.classify_stream(ctx, db, h1.ptr, len(r1), h2.ptr, len(r2), per_batch, P, taxdb=tax, report=report,
                            sam_fd=sam_fd, per_read_fd=pr_fd, sam_header=header)
    os.close(sam_fd)
    os.close(pr_fd)
    sam = open(sam_path, "rb").read()
    per_read = open(per_read_path, "rb").read()
    summary = tax.summary(res["tax_ids"], res["pairs"])
    xml = tax.report_xml(report, db, db.gene_extras(), res["pairs"])
    # ---- the same loop inside the library (kslam_stream_classify, include/kslam_stream.h): same files, same report ----
    report_n = X.Report()
    sam_fd = os.open(sam_path + ".native", os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    pr_fd = os.open(per_read_path + ".native", os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    nat = S.classify_stream_native(ctx, db, h1.ptr, len(r1), h2.ptr, len(r2), per_batch, P, taxdb=tax, report=report_n,
                                   sam_fd=sam_fd, per_read_fd=pr_fd, sam_header=header)
    os.close(sam_fd)
    os.close(pr_fd)
    assert open(sam_path + ".native", "rb").read() == sam and open(per_read_path + ".native", "rb").read() == per_read
    assert nat["tax_ids"].tolist() == res["tax_ids"].tolist() and nat["n_pairs"] == res["pairs"]
    assert nat["n_batches"] == len(res["batches"]) and nat["sam_bytes"] + len(header) == len(sam)
    assert tax.report_xml(report_n, db, db.gene_extras(), nat["n_pairs"]) == xml
    report_n.close()
    if per_batch == 833:
        # the loop's other shapes: host stage on ONE thread, a pool of three, lanes that poll and sleep instead of spinning
        os.environ["KSLAM_LANE_WAITS"] = "yield"
        try:
            ctx2 = kslam.Context()
        finally:
            del os.environ["KSLAM_LANE_WAITS"]
        ctx2._chk(ctx2._L.kslam_set_index(ctx2._h, db.n_entries, C.cast(bases_pp, C.c_void_p), C.cast(lens_p, C.c_void_p)))
        report_v = X.Report()
        sam_fd = os.open(sam_path + ".v", os.O_RDWR | os.O_CREAT | os.O_TRUNC)
        pr_fd = os.open(per_read_path + ".v", os.O_RDWR | os.O_CREAT | os.O_TRUNC)
        var = S.classify_stream_native(ctx2, db, h1.ptr, len(r1), h2.ptr, len(r2), per_batch, P, taxdb=tax, report=report_v,
                                       sam_fd=sam_fd, per_read_fd=pr_fd, sam_header=header, host_threads=1, pool_threads=3, depth=2)
        os.close(sam_fd)
        os.close(pr_fd)
        assert open(sam_path + ".v", "rb").read() == sam and open(per_read_path + ".v", "rb").read() == per_read
        assert var["tax_ids"].tolist() == res["tax_ids"].tolist()
        assert tax.report_xml(report_v, db, db.gene_extras(), var["n_pairs"]) == xml
        report_v.close()
        ctx2.close()
    n_batches = (n_pairs + per_batch - 1) // per_batch
    assert res["pairs"] == n_pairs and len(res["batches"]) == n_batches
    assert [b["pairs"] for b in res["batches"]] == [min(per_batch, n_pairs - k * per_batch) for k in range(n_batches)]
    if pseudo:
        assert all(b["pseudo_assembly_on"] == "gpu" for b in res["batches"])

    # ---- checker: the oracle chain, batch by batch with the same boundaries ----
    _, oentries = dbo.parse((dbdir / "database").read_bytes())
    ogb = [e["bases"] for e in oentries]
    oI = T.Index(ogb, locus_tags=[e["locusTag"] for e in oentries], taxonomy_ids=[e["taxonomyID"] for e in oentries],
                 genes=[[(g["start"], g["stop"], g["geneName"], g["proteinID"], g["product"]) for g in e["genes"]]
                        for e in oentries])
    otree = oracle.taxonomy_tree(taxdb)
    esam, eper_read, etax, ebatches, limits = [], [], [], [], []
    for k in range(n_batches):
        lo, hi = k * per_batch, min(n_pairs, (k + 1) * per_batch)
        b_reads = rb[lo:hi] + rb[n_pairs + lo:n_pairs + hi]
        b_quals = quals[lo:hi] + quals[n_pairs + lo:n_pairs + hi]
        b_ids = ids[lo:hi] + ids[lo:hi]
        eal, ecig, _ = oracle.align_to_database(b_reads, ogb)
        oR = T.Reads(b_reads, b_quals, b_ids)
        st = T.TailStats()
        esam.append(oracle.tail_sam(P, oR.view, oI.view, eal, ecig, stats=st))
        limits.append(int(st.max_insert_size))
        erp, epr = oracle.tail_pairs(P, oR.view, eal)
        t = [otree.lca([oentries[int(e)]["taxonomyID"] for e in epr["entry"][int(g["first"]):int(g["first"]) + int(g["count"])]])
             for g in erp]
        etax += t
        eper_read.append(b"".join(b"%s\t%d\n" % (b_ids[int(g["r1_read"])], x) for g, x in zip(erp, t)))
        ebatches.append((b_ids, erp, epr))
    assert sam == header + b"".join(esam) and sam.count(b"\n") > 2 * n_pairs * 0.9
    assert [b["max_insert_size"] for b in sorted(res["batches"], key=lambda b: b["batch"])] == limits
    assert res["tax_ids"].tolist() == etax and len(set(etax)) > 6
    assert per_read == b"".join(eper_read)
    assert summary == oracle.taxonomy_summary(otree, etax, n_pairs)
    from test_taxonomy import _xml_restatement
    ogenes = [[{"start": g["start"], "stop": g["stop"], "name": g["geneName"], "protein": g["proteinID"], "product": g["product"],
                "locus": g["locusTag"], "reference": g["referenceSequence"], "id": g["geneID"]} for g in e["genes"]] for e in oentries]
    exml, taxa = _xml_restatement(tax, [e["taxonomyID"] for e in oentries], ogenes, None, ebatches, n_pairs)
    if xml != exml and os.path.isdir(os.path.join(ROOT, "gpurun_out")):      # leave both texts behind for a diff
        open(os.path.join(ROOT, "gpurun_out", "stream_xml_got.xml"), "wb").write(xml)
        open(os.path.join(ROOT, "gpurun_out", "stream_xml_exp.xml"), "wb").write(exml)
    assert xml == exml and len(taxa) > 6 and xml.count(b"<read>") > 0.8 * n_pairs
    if n_batches > 1 and per_batch == 700:
        # the boundaries matter: one batch of everything has another insert-size limit or at least other statistics
        assert len(set(limits)) >= 1
    h1.close()
    h2.close()
    ctx.close()
    db.close()
    report.close()
    tax.close()
    otree.close()


def test_stream_stops_at_max_pairs_and_reports_mismatched_files(kslam, synth, tmp_path):
    """--num-reads (maxNumReads, src/SLAM.h:193, 201-203) cuts the last batch short; an R2 file with fewer
    records than R1 is refused like kslam_fastq_parse_pair refuses it."""
    import ctypes as C
    D = importlib.import_module("kslam_amd.db")
    T = importlib.import_module("kslam_amd.tail")
    S = importlib.import_module("kslam_amd.stream")
    n_pairs = 900
    dbdir, taxdb, rb, quals, ids, r1, r2 = _make_case(synth, tmp_path, n_pairs, b"\n", seed=77)
    db = D.Database.load(dbdir / "database")
    ctx = kslam.Context()
    bases_pp, lens_p = db.entry_pointers()
    ctx._chk(ctx._L.kslam_set_index(ctx._h, db.n_entries, C.cast(bases_pp, C.c_void_p), C.cast(lens_p, C.c_void_p)))
    h1, h2 = kslam.HostBuffer(len(r1) + 64), kslam.HostBuffer(len(r2) + 64)
    h1.a[:len(r1)] = np.frombuffer(r1, dtype=np.uint8)
    h2.a[:len(r2)] = np.frombuffer(r2, dtype=np.uint8)
    P = T.TailParams.default()
    res = S.classify_stream(ctx, db, h1.ptr, len(r1), h2.ptr, len(r2), 400, P, max_pairs_total=650)
    assert [b["pairs"] for b in sorted(res["batches"], key=lambda b: b["batch"])] == [400, 250] and res["pairs"] == 650
    nat = S.classify_stream_native(ctx, db, h1.ptr, len(r1), h2.ptr, len(r2), 400, P, max_pairs_total=650)
    assert nat["n_batches"] == 2 and nat["n_pairs"] == 650 and len(nat["tax_ids"]) == 0
    # R2 one record short
    cut = r2.rstrip(b"\n").rfind(b"\n@")
    with pytest.raises(kslam.KslamError, match="mismatch in R1 and R2"):
        S.classify_stream(ctx, db, h1.ptr, len(r1), h2.ptr, cut + 1, 400, P)
    with pytest.raises(kslam.KslamError, match="mismatch in R1 and R2"):
        S.classify_stream_native(ctx, db, h1.ptr, len(r1), h2.ptr, cut + 1, 400, P)
    # and the context is usable afterwards
    assert S.classify_stream_native(ctx, db, h1.ptr, len(r1), h2.ptr, len(r2), 500, P)["n_pairs"] == n_pairs
    # a failure on the SECOND host thread (the per-read file cannot be written) stops the loop with its message, and a
    # failing SAM sink does the same from the writer's side
    X = importlib.import_module("kslam_amd.taxonomy")
    tax = X.TaxDB((dbdir / "taxDB").read_bytes())
    rd, wr = os.pipe()
    os.close(rd)                                          # writing to wr now fails with EPIPE
    import signal
    old = signal.signal(signal.SIGPIPE, signal.SIG_IGN)
    try:
        with pytest.raises(kslam.KslamError, match="per-read file"):
            S.classify_stream_native(ctx, db, h1.ptr, len(r1), h2.ptr, len(r2), 400, P, taxdb=tax, per_read_fd=wr)
        with pytest.raises(kslam.KslamError, match="SAM"):
            S.classify_stream_native(ctx, db, h1.ptr, len(r1), h2.ptr, len(r2), 400, P, taxdb=tax, sam_fd=wr)
    finally:
        signal.signal(signal.SIGPIPE, old)
        os.close(wr)
    assert S.classify_stream_native(ctx, db, h1.ptr, len(r1), h2.ptr, len(r2), 500, P, taxdb=tax)["n_pairs"] == n_pairs
    tax.close()
    h1.close()
    h2.close()
    ctx.close()
    db.close()


@pytest.mark.parametrize("pseudo,per_batch", [(True, 800), (False, 2500)])
def test_single_end_stream_equals_the_reference_loop(kslam, oracle, synth, tmp_path, pseudo, per_batch):
    """The isPaired == false branch of the batch loop (src/SLAM.h:193-233: getSequencesFromFASTQFile, dummy alignment
    pairs, score screen [, pseudo-assembly + screen], one SAM row per alignment): ONE FASTQ text through
    kslam_stream_classify with tail.paired = 0 and r2 = NULL, against the oracle chain batch by batch."""
    import ctypes as C
    D = importlib.import_module("kslam_amd.db")
    T = importlib.import_module("kslam_amd.tail")
    X = importlib.import_module("kslam_amd.taxonomy")
    S = importlib.import_module("kslam_amd.stream")
    dbo = importlib.import_module("oracle.db_oracle")
    n_reads = 2500
    dbdir, taxdb, rb, quals, ids, r1, _ = _make_case(synth, tmp_path, n_reads, b"\n", seed=411)
    rb, quals = rb[:n_reads], quals[:n_reads]                     # the R1 file alone is the data set
    db = D.Database.load(dbdir / "database")
    ctx = kslam.Context()
    bases_pp, lens_p = db.entry_pointers()
    ctx._chk(ctx._L.kslam_set_index(ctx._h, db.n_entries, C.cast(bases_pp, C.c_void_p), C.cast(lens_p, C.c_void_p)))
    tax = X.TaxDB((dbdir / "taxDB").read_bytes())
    report = X.Report()
    h1 = kslam.HostBuffer(len(r1) + 64)
    h1.a[:len(r1)] = np.frombuffer(r1, dtype=np.uint8)
    P = T.TailParams.default(paired=False, pseudo_assembly=pseudo)
    header = T.sam_header(db, b"SLAM --db db R1.fq")
    sam_path, per_read_path = str(tmp_path / "single.sam"), str(tmp_path / "single_PerRead")
    sam_fd = os.open(sam_path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    pr_fd = os.open(per_read_path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    nat = S.classify_stream_native(ctx, db, h1.ptr, len(r1), None, 0, per_batch, P, taxdb=tax, report=report,
                                   sam_fd=sam_fd, per_read_fd=pr_fd, sam_header=header)
    os.close(sam_fd)
    os.close(pr_fd)
    sam = open(sam_path, "rb").read()
    per_read = open(per_read_path, "rb").read()
    n_batches = (n_reads + per_batch - 1) // per_batch
    assert nat["n_pairs"] == n_reads and nat["n_batches"] == n_batches
    # the Python loop, single-end: the same files
    sam_fd = os.open(sam_path + ".py", os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    pr_fd = os.open(per_read_path + ".py", os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    res = S.classify_stream(ctx, db, h1.ptr, len(r1), None, 0, per_batch, P, taxdb=tax, sam_fd=sam_fd, per_read_fd=pr_fd,
                            sam_header=header)
    os.close(sam_fd)
    os.close(pr_fd)
    assert open(sam_path + ".py", "rb").read() == sam and open(per_read_path + ".py", "rb").read() == per_read
    assert res["pairs"] == n_reads and res["tax_ids"].tolist() == nat["tax_ids"].tolist()
    # a second text with paired = 0 is refused, and so is paired data handed over as one text
    with pytest.raises(kslam.KslamError, match="ONE text"):
        S.classify_stream_native(ctx, db, h1.ptr, len(r1), h1.ptr, len(r1), per_batch, P)
    with pytest.raises(kslam.KslamError, match="mismatch in R1 and R2"):
        S.classify_stream_native(ctx, db, h1.ptr, len(r1), h1.ptr, 0, per_batch, T.TailParams.default())

    _, oentries = dbo.parse((dbdir / "database").read_bytes())
    ogb = [e["bases"] for e in oentries]
    oI = T.Index(ogb, locus_tags=[e["locusTag"] for e in oentries], taxonomy_ids=[e["taxonomyID"] for e in oentries],
                 genes=[[(g["start"], g["stop"], g["geneName"], g["proteinID"], g["product"]) for g in e["genes"]]
                        for e in oentries])
    otree = oracle.taxonomy_tree(taxdb)
    esam, eper_read, etax, ebatches = [], [], [], []
    for k in range(n_batches):
        lo, hi = k * per_batch, min(n_reads, (k + 1) * per_batch)
        b_reads, b_quals, b_ids = rb[lo:hi], quals[lo:hi], ids[lo:hi]
        eal, ecig, _ = oracle.align_to_database(b_reads, ogb)
        oR = T.Reads(b_reads, b_quals, b_ids)
        esam.append(oracle.tail_sam(P, oR.view, oI.view, eal, ecig))
        erp, epr = oracle.tail_pairs(P, oR.view, eal)
        t = [otree.lca([oentries[int(e)]["taxonomyID"] for e in epr["entry"][int(g["first"]):int(g["first"]) + int(g["count"])]])
             for g in erp]
        etax += t
        eper_read.append(b"".join(b"%s\t%d\n" % (b_ids[int(g["r1_read"])], x) for g, x in zip(erp, t)))
        ebatches.append((b_ids, erp, epr))
    assert sam == header + b"".join(esam) and sam.count(b"\n") > 0.9 * n_reads
    assert nat["tax_ids"].tolist() == etax and len(set(etax)) > 6
    assert per_read == b"".join(eper_read)
    assert tax.summary(nat["tax_ids"], n_reads) == oracle.taxonomy_summary(otree, etax, n_reads)
    from test_taxonomy import _xml_restatement
    ogenes = [[{"start": g["start"], "stop": g["stop"], "name": g["geneName"], "protein": g["proteinID"], "product": g["product"],
                "locus": g["locusTag"], "reference": g["referenceSequence"], "id": g["geneID"]} for g in e["genes"]] for e in oentries]
    exml, taxa = _xml_restatement(tax, [e["taxonomyID"] for e in oentries], ogenes, None, ebatches, n_reads)
    assert tax.report_xml(report, db, db.gene_extras(), n_reads) == exml and len(taxa) > 6
    h1.close()
    ctx.close()
    db.close()
    report.close()
    tax.close()
    otree.close()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_golden_slam_loop_through_the_abi(kslam, tmp_path, tag):
    """FASTQ files + database + taxDB -> SAM / _PerRead / _abbreviated / XML through kslam_stream_classify, against the
    files the reference's OWN metagenomicAnalysis_Low_Mem wrote for the same inputs (oracle/_ref/libslam_ref.so ->
    tests/golden/slam_loop.npz; tests/test_reference_loop.py regenerates them where the reference is present).  No
    oracle in this test: the expected bytes are the real reference's."""
    import ctypes as C
    import ref_loop_case as R
    from test_reference_loop import load_fixture_case
    D = importlib.import_module("kslam_amd.db")
    T = importlib.import_module("kslam_amd.tail")
    X = importlib.import_module("kslam_amd.taxonomy")
    S = importlib.import_module("kslam_amd.stream")
    z = np.load(os.path.join(ROOT, "tests", "golden", "slam_loop.npz"), allow_pickle=False)
    case = load_fixture_case(z, tag)
    dbdir = R.write_case(case, tmp_path, D)
    per_batch, pseudo = int(z[tag + "_per_batch"]), bool(z[tag + "_pseudo"])
    db = D.Database.load(os.path.join(dbdir, "database"))
    ctx = kslam.Context()
    bases_pp, lens_p = db.entry_pointers()
    ctx._chk(ctx._L.kslam_set_index(ctx._h, db.n_entries, C.cast(bases_pp, C.c_void_p), C.cast(lens_p, C.c_void_p)))
    tax = X.TaxDB(case["taxdb"])
    report = X.Report()
    r1, r2 = case["r1"], case["r2"]
    h1, h2 = kslam.HostBuffer(len(r1) + 64), kslam.HostBuffer(len(r2) + 64)
    h1.a[:len(r1)] = np.frombuffer(r1, dtype=np.uint8)
    h2.a[:len(r2)] = np.frombuffer(r2, dtype=np.uint8)
    P = T.TailParams.default(pseudo_assembly=pseudo)
    header = T.sam_header(db, b"SLAM --db db R1.fq R2.fq")
    sam_path, per_read_path = str(tmp_path / "out.sam"), str(tmp_path / "out_PerRead")
    sam_fd = os.open(sam_path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    pr_fd = os.open(per_read_path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    res = S.classify_stream_native(ctx, db, h1.ptr, len(r1), h2.ptr, len(r2), per_batch, P, taxdb=tax, report=report,
                                   sam_fd=sam_fd, per_read_fd=pr_fd, sam_header=header)
    os.close(sam_fd)
    os.close(pr_fd)
    assert res["n_pairs"] == case["n_pairs"]
    assert open(sam_path, "rb").read() == z[tag + "_sam"].tobytes()
    assert open(per_read_path, "rb").read() == z[tag + "_per_read"].tobytes()
    assert tax.summary(res["tax_ids"], res["n_pairs"]) == z[tag + "_abbreviated"].tobytes()
    assert tax.report_xml(report, db, db.gene_extras(), res["n_pairs"]) == z[tag + "_xml"].tobytes()
    h1.close()
    h2.close()
    ctx.close()
    db.close()
    report.close()
    tax.close()


@pytest.mark.parametrize("route", ["host_text", "host_pseudo"])
def test_stream_routes_that_leave_work_to_the_host(kslam, oracle, synth, tmp_path, monkeypatch, route):
    """The batch loop's other routes, with its default TWO host threads (SAM text and taxonomy side by side):
    host_text    KSLAM_HOST_SAM_TEXT=1: the device hands rows / details / pairs over and the CPUs format the text
    host_pseudo  KSLAM_PSEUDO_CAP=50: the device declines pseudo-assembly (an entry holds more alignment pairs than the cap),
                 so the batch comes back without text and the host runs pseudo-assembly + second screen + per-pair sort
                 (kslam_tail_finish_prepare) BEFORE the text and the classification read the arrays
    Both must write what the default route (everything on the GPU) writes, which the tests above hold against the oracle."""
    import ctypes as C
    D = importlib.import_module("kslam_amd.db")
    T = importlib.import_module("kslam_amd.tail")
    X = importlib.import_module("kslam_amd.taxonomy")
    S = importlib.import_module("kslam_amd.stream")
    n_pairs, per_batch = 2500, 900
    dbdir, taxdb, rb, quals, ids, r1, r2 = _make_case(synth, tmp_path, n_pairs, b"\n", seed=909)
    db = D.Database.load(dbdir / "database")
    tax = X.TaxDB((dbdir / "taxDB").read_bytes())
    h1, h2 = kslam.HostBuffer(len(r1) + 64), kslam.HostBuffer(len(r2) + 64)
    h1.a[:len(r1)] = np.frombuffer(r1, dtype=np.uint8)
    h2.a[:len(r2)] = np.frombuffer(r2, dtype=np.uint8)
    P = T.TailParams.default(pseudo_assembly=True)
    header = T.sam_header(db, b"SLAM --db db R1.fq R2.fq")
    bases_pp, lens_p = db.entry_pointers()

    def run(tag):
        ctx = kslam.Context()
        ctx._chk(ctx._L.kslam_set_index(ctx._h, db.n_entries, C.cast(bases_pp, C.c_void_p), C.cast(lens_p, C.c_void_p)))
        report = X.Report()
        sam_path, pr_path = str(tmp_path / (tag + ".sam")), str(tmp_path / (tag + "_PerRead"))
        sam_fd = os.open(sam_path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
        pr_fd = os.open(pr_path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
        res = S.classify_stream_native(ctx, db, h1.ptr, len(r1), h2.ptr, len(r2), per_batch, P, taxdb=tax, report=report,
                                       sam_fd=sam_fd, per_read_fd=pr_fd, sam_header=header)
        os.close(sam_fd)
        os.close(pr_fd)
        xml = tax.report_xml(report, db, db.gene_extras(), res["n_pairs"])
        report.close()
        ctx.close()
        return open(sam_path, "rb").read(), open(pr_path, "rb").read(), res["tax_ids"].tolist(), xml, res

    base = run("gpu")
    assert base[4]["batches_pseudo_on_host"] == 0 and base[0].count(b"\n") > 2 * n_pairs * 0.9
    if route == "host_text":
        monkeypatch.setenv("KSLAM_HOST_SAM_TEXT", "1")
    else:
        monkeypatch.setenv("KSLAM_PSEUDO_CAP", "50")
    other = run(route)
    if route == "host_pseudo":
        assert other[4]["batches_pseudo_on_host"] == other[4]["n_batches"] == 3
    assert other[0] == base[0] and other[1] == base[1] and other[2] == base[2] and other[3] == base[3]
    h1.close()
    h2.close()
    db.close()
    tax.close()
