"""FASTQ text + <db>/database + <db>/taxDB in, SAM text + _PerRead + abbreviated report out: the whole
classification flow of the reference's low-memory driver (src/SLAM.h:159-268) with the MI355X hot
path in the middle, against the CPU chain of restatements (oracle/), byte for byte.

  product:  kslam_fastq_parse_pair -> kslam_db_load -> kslam_set_index (the database's own columns)
            -> kslam_load_reads -> kslam_align_resident (HIP) -> kslam_tail_sam / kslam_tail_pairs
            -> kslam_tail_classify (per-read LCA) -> kslam_taxonomy_summary
  checker:  oracle fastq reader -> oracle/db_oracle.py -> oracle alignToDatabase -> oracle tail ->
            oracle taxonomy tree

This is the -m gpu evidence for SURVEY section 8f rows N2 (database load), N3 (FASTQ ingest) and the
per-read LCA of N1.  Also here: the C++ binding header a k-SLAM maintainer would include
(k-slam_amd/host/slam_hot_path.hpp) compiled and run against the library.
"""
import ctypes as C
import importlib
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _taxdb_text(n_species, n_strains):
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


def _fastq_text(bases, quals, ids, mate, eol):
    return b"".join(b"@" + i + b"/%d extra words" % mate + eol + b + eol + b"+" + eol + q + eol
                    for b, q, i in zip(bases, quals, ids))


@pytest.mark.parametrize("eol", [b"\n", b"\r\n"])
def test_fastq_and_database_files_to_sam_and_per_read_taxa(kslam, oracle, synth, tmp_path, eol):
    F = importlib.import_module("kslam_amd.fastq")
    D = importlib.import_module("kslam_amd.db")
    T = importlib.import_module("kslam_amd.tail")
    X = importlib.import_module("kslam_amd.taxonomy")
    dbo = importlib.import_module("oracle.db_oracle")
    n_species, n_strains, n_pairs = 4, 3, 3000
    rng = np.random.default_rng(2024)
    genomes = synth.make_genomes(41, n_species, n_strains, 30000, strain_sub=0.02, strain_indel=0.001,
                                 shared_segment=2500)
    reads, _ = synth.make_paired_reads(42, genomes, n_pairs, read_len=120, frag_mean=320, frag_sd=40, sub_rate=0.015,
                                       indel_rate=0.003, n_rate=0.001, edge_frac=0.04, unmapped_frac=0.05)
    # ---- the files ----
    gb = synth.to_bytes(genomes)
    entries = [{"bases": g, "taxonomyID": 1000 + i if i != 5 else 0, "genbankID": 7000 + i,
                "locusTag": b"NC_%06d.1" % i, "isPlasmid": i % 4 == 3,
                "genes": [{"geneName": b"gene%d" % k, "proteinID": b"WP_%d.1" % (100 * i + k),
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

    # ---- product chain ----
    batch, u1, u2 = F.parse_pair(r1, r2)
    assert (u1, u2) == (len(r1), len(r2)) and batch.n_reads == 2 * n_pairs
    db = D.Database.load(dbdir / "database")
    ctx = kslam.Context()
    bases_pp, lens_p = db.entry_pointers()               # kslam_set_index straight from the loaded columns
    ctx._chk(ctx._L.kslam_set_index(ctx._h, db.n_entries, C.cast(bases_pp, C.c_void_p), C.cast(lens_p, C.c_void_p)))
    cat, off = batch.bases_array()
    ctx.load_reads_arrays(cat, off)
    n_out, n_cig = ctx.align_resident()
    ov, cg, release = ctx.take_results()
    P = T.TailParams.default()
    sam, st = T.tail_sam(P, batch, db, ov, cg)
    rp, pr, _ = T.tail_pairs(P, batch, ov)
    tax = X.TaxDB((dbdir / "taxDB").read_bytes())
    tax_ids, per_read = tax.classify(P, batch, db, rp, pr)
    summary = tax.summary(tax_ids, n_pairs)
    report = X.Report()
    report.add_batch(batch, db, rp, pr, tax_ids)
    xml = tax.report_xml(report, db, db.gene_extras(), n_pairs)
    header = T.sam_header(db, b"SLAM --db db R1.fq R2.fq")

    # ---- checker chain ----
    b1, q1, i1, _ = oracle.fastq_read(r1)
    b2, q2, i2, _ = oracle.fastq_read(r2)
    assert b1 + b2 == rb and q1 + q2 == quals and i1 == ids and i2 == ids
    _, oentries = dbo.parse((dbdir / "database").read_bytes())
    ogb = [e["bases"] for e in oentries]
    assert ogb == gb
    eal, ecig, _ = oracle.align_to_database(b1 + b2, ogb)
    oR = T.Reads(b1 + b2, q1 + q2, i1 + i2)
    oI = T.Index(ogb, locus_tags=[e["locusTag"] for e in oentries], taxonomy_ids=[e["taxonomyID"] for e in oentries],
                 genes=[[(g["start"], g["stop"], g["geneName"], g["proteinID"], g["product"]) for g in e["genes"]]
                        for e in oentries])
    esam = oracle.tail_sam(P, oR.view, oI.view, eal, ecig)
    erp, epr = oracle.tail_pairs(P, oR.view, eal)
    otree = oracle.taxonomy_tree(taxdb)
    etax = [otree.lca([oentries[int(e)]["taxonomyID"] for e in epr["entry"][int(g["first"]):int(g["first"]) + int(g["count"])]])
            for g in erp]
    eper_read = b"".join(b"%s\t%d\n" % (ids[int(g["r1_read"])], t) for g, t in zip(erp, etax))

    assert len(eal) > 2 * n_pairs and (eal["cigar_len"] > 1).sum() > 300
    assert len(ov) == len(eal)
    for f in ("read", "entry", "rel", "revcomp", "score", "ref_begin", "ref_end", "query_begin", "query_end",
              "cigar_len", "cigar_off"):
        assert (ov[f] == eal[f]).all(), f
    assert np.array_equal(cg, ecig)
    assert sam == esam and sam.count(b"\n") > 2 * n_pairs * 0.9
    assert tax_ids.tolist() == etax and len(set(etax)) > 6
    assert per_read == eper_read
    assert summary == oracle.taxonomy_summary(otree, etax, n_pairs)
    # the XML report (writeResults): restated step by step in tests/test_taxonomy.py, from the checker chain's pairs
    from test_taxonomy import _xml_restatement
    ogenes = [[{"start": g["start"], "stop": g["stop"], "name": g["geneName"], "protein": g["proteinID"], "product": g["product"],
                "locus": g["locusTag"], "reference": g["referenceSequence"], "id": g["geneID"]} for g in e["genes"]] for e in oentries]
    exml, taxa = _xml_restatement(tax, [e["taxonomyID"] for e in oentries], ogenes, None, [(ids, erp, epr)], n_pairs)
    assert xml == exml and len(taxa) > 6 and sum(len(t["genes"]) for t in taxa) > 50 and xml.count(b"<read>") > 0.8 * n_pairs
    assert header == oracle.sam_header(oI.view, b"SLAM --db db R1.fq R2.fq")
    # ---- the same SAM text with NM / log-probability / MD computed on the GPU (kslam_row_details):
    # qualities straight from the parsed batch's column, rows checked against the restatement of the
    # reference's walk, and the writer given an index whose bases it must not need ----
    from rowdetails_ref import row_details
    qcol = np.frombuffer(b"".join(batch.quality) + b"\0", dtype=np.uint8)
    ctx.load_qualities_array(qcol)
    ctx.row_details()
    det, md = ctx.take_row_details(len(ov))
    edet, emd = row_details(eal, ecig, b1 + b2, q1 + q2, ogb, kslam.ROW_DETAIL_DT)
    for f in ("nm", "md_len", "md_off", "flags"):
        assert (det[f] == edet[f]).all(), f
    assert (det["logp"].view(np.uint64) == edet["logp"].view(np.uint64)).all()      # the same double, bit for bit
    assert md.tobytes() == emd.tobytes() and (det["nm"] > 0).sum() > 1000
    blind = T.Index([bytes(len(g)) for g in ogb], locus_tags=[e["locusTag"] for e in oentries],
                    taxonomy_ids=[e["taxonomyID"] for e in oentries],
                    genes=[[(g["start"], g["stop"], g["geneName"], g["proteinID"], g["product"]) for g in e["genes"]]
                           for e in oentries])
    sam2, _ = T.tail_sam_rows(P, batch, blind, ov, cg, det, md)
    assert sam2 == esam
    # ---- and through the pipelined column entry: the parser's (page-locked) columns are not copied ----
    cols = batch._cols
    tk = [ctx.submit_batch_columns(batch.n_reads, cols.bases, cols.quality, cols.bases_off) for _ in range(3)]
    for t in tk:
        o3, c3, d3, m3, rel3 = ctx.collect_batch(t)
        assert o3.tobytes() == ov.tobytes() and c3.tobytes() == cg.tobytes()
        assert d3.tobytes() == det.tobytes() and m3.tobytes() == md.tobytes()
        rel3()
    # ---- and straight from the FASTQ TEXT: the host only indexes the records (kslam_fastq_index_pair),
    # the texts go up from page-locked memory and the columns are cut out of them on the device ----
    h1, h2 = kslam.HostBuffer(len(r1) + 64), kslam.HostBuffer(len(r2) + 64)
    h1.a[:len(r1)] = np.frombuffer(r1, dtype=np.uint8)
    h2.a[:len(r2)] = np.frombuffer(r2, dtype=np.uint8)
    ix, v1, v2 = F.index_pair(h1.ptr, len(r1), h2.ptr, len(r2))
    assert (v1, v2) == (len(r1), len(r2)) and ix.n_reads == batch.n_reads and ix.ids == batch.ids
    tk = [ctx.submit_batch_fastq(h1.ptr, len(r1), h2.ptr, len(r2), ix.n_reads, ix._cols.bases_off,
                                 ix.layout.bases_at, ix.layout.quality_at) for _ in range(2)]
    for t in tk:
        o4, c4, d4, m4, rel4 = ctx.collect_batch(t)
        assert o4.tobytes() == ov.tobytes() and c4.tobytes() == cg.tobytes()
        assert d4.tobytes() == det.tobytes() and m4.tobytes() == md.tobytes()
        sam4, _ = T.tail_sam_rows(P, ix, blind, o4, c4, d4, m4)      # the indexed batch as the tail's reads view
        assert sam4 == esam
        rel4()
    ix.close()
    h1.close()
    h2.close()
    release()
    ctx.close()
    batch.close()
    db.close()
    report.close()
    tax.close()
    otree.close()


def test_cpp_binding_header_builds_and_runs(tmp_path):
    """k-slam_amd/host/slam_hot_path.hpp is the binding INTEGRATION.md tells a k-SLAM maintainer to
    include; tests/host_mirror_check.cpp instantiates it with look-alikes of the reference's Overlap /
    Alignment / GenbankEntry / FASTQ types and aligns two planted reads."""
    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the image"
    exe = str(tmp_path / "host_mirror_check")
    libdir = os.path.join(ROOT, "k-slam_amd")
    subprocess.check_call([gxx, "-std=c++11", "-O1", os.path.join(ROOT, "tests", "host_mirror_check.cpp"), "-o", exe,
                           "-L" + libdir, "-lkslam_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("read ")]
    assert len(lines) == 2 and "score 300" in lines[0] and "cigar 150M" in lines[0] and "rel 3216" in lines[1]
    assert "multi identical" in out.stdout


@pytest.mark.parametrize("seed,eol_mix,at_eof,max_pairs,tail_cut", [
    (1, False, True, 0, 0), (2, True, True, 0, 0), (3, True, True, 50, 0), (4, True, False, 0, 0), (5, True, True, 0, 1),
    (6, True, True, 0, 2), (7, True, False, 70, 0), (8, False, True, 0, 3)])
def test_fastq_records_found_on_the_device(kslam, synth, seed, eol_mix, at_eof, max_pairs, tail_cut):
    """kslam_submit_batch_fastq_text (line index, identifiers, offsets and columns all built on the GPU)
    against the host parser (k-slam_amd/host/fastq.cpp, which tests/test_fastq.py pins against the REAL
    reference reader): mixed "\\n" / "\\r\\n", headers with spaces and slashes, empty reads, a stream that ends
    without its last terminator / in the middle of a record / right after a "\\r", a prefix of a longer stream,
    a cap on the number of pairs; and the alignment of that batch equals the alignment of the host's columns."""
    F = importlib.import_module("kslam_amd.fastq")
    rng = np.random.default_rng(900 + seed)
    genomes = synth.make_genomes(70 + seed, 2, 2, 12000)
    reads, _ = synth.make_paired_reads(71 + seed, genomes, 120, read_len=100, frag_mean=260, frag_sd=30)
    rb = synth.to_bytes(reads)
    heads = [b"@r%d", b"@r%d desc text", b"@r%d/1", b"@r%d/2 x/y", b"@ r%d", b"@a/b/c%d", b"@/r%d", b"@r%d\tTAB/9", b"@", b"x"]
    eols = [b"\n", b"\r\n"]

    def text(block):
        out = []
        for k, seq in enumerate(block):
            h = heads[int(rng.integers(0, len(heads)))]
            h = h % k if b"%d" in h else h
            if rng.random() < 0.05:
                seq = b""
            q = bytes(rng.integers(33, 75, len(seq), dtype=np.uint8))
            for line in (h, seq, b"+", q):
                out.append(line + (eols[int(rng.integers(0, 2))] if eol_mix else b"\n"))
        return b"".join(out)
    t1, t2 = text(rb[:120]), text(rb[120:])
    if tail_cut == 1:      # no terminator after the last quality line
        t1, t2 = t1.rstrip(b"\r\n"), t2.rstrip(b"\r\n")
    if tail_cut == 2:      # the last record has no quality line at all: the empty line read at end of stream completes it
        def drop_last_line(t):
            body = t.rstrip(b"\r\n")
            return body[:max(body.rfind(b"\n"), body.rfind(b"\r")) + 1]
        t1, t2 = drop_last_line(t1), drop_last_line(t2)
        # (that record's bases line is then longer than its empty quality line: both parsers must refuse it)
    if tail_cut == 3:      # blank lines after the last record
        t1, t2 = t1 + b"\n\n\n", t2 + b"\n\n\n"
    if not at_eof:
        t1, t2 = t1 + b"@partial\nACG", t2 + b"@partial\r"
    h1, h2 = kslam.HostBuffer(len(t1) + 64), kslam.HostBuffer(len(t2) + 64)
    h1.a[:len(t1)] = np.frombuffer(t1, dtype=np.uint8)
    h2.a[:len(t2)] = np.frombuffer(t2, dtype=np.uint8)
    c = kslam.Context()
    c.set_index(synth.to_bytes(genomes))
    tk = c.submit_batch_fastq_text(h1.ptr, len(t1), h2.ptr, len(t2), max_pairs=max_pairs, at_eof=at_eof)
    try:
        full, f1, f2 = F.parse_pair(t1, t2, max_pairs=max_pairs, at_eof=at_eof)
        ok = all(len(b) == len(q) for b, q in zip(full.bases, full.quality))
    except kslam.KslamError as e:
        full, ok = None, False
        host_err = str(e)
    if not ok:
        with pytest.raises(kslam.KslamError, match="quality line|mismatch in R1 and R2"):
            c.collect_batch(tk)
        assert tail_cut == 2 or full is None
        c.close()
        return
    ov, cg, det, md, release = c.collect_batch(tk)
    R = c.last_reads
    assert R.n_reads == full.n_reads and R.consumed == (f1, f2)
    _, foff = full.bases_array()
    assert (R.bases_off == foff).all()
    ids = [R.ids_bytes[int(R.ids_off[i]):int(R.ids_off[i + 1])].tobytes() for i in range(R.n_reads)]
    assert ids == full.ids
    # the device's columns are the host's: same alignment, same row details
    cols = full._cols
    o2, c2, d2, m2, rel2 = c.collect_batch(c.submit_batch_columns(full.n_reads, cols.bases, cols.quality, cols.bases_off))
    assert ov.tobytes() == o2.tobytes() and cg.tobytes() == c2.tobytes() and det.tobytes() == d2.tobytes() and md.tobytes() == m2.tobytes()
    assert len(ov) > 100 or max_pairs
    release()
    rel2()
    c.close()
    h1.close()
    h2.close()
