"""include/kslam_samtext.h: the SAM records and the <out>_PerRead lines written on the GPU (csrc/samtext.hip) against the host
tail's text for the same rows (kslam_tail_finish_write_rows, kslam_tail_classify -- themselves held against the oracle chain
and the reference's own loop in tests/test_tail.py, tests/test_reference_loop.py).  Byte for byte: flags, mate fields, tlen,
soft clips, MD / NM, XS truncation, X0, XT, the gene tags, the mapping qualities (libm on the host, one byte per row handed
back), --num-alignments, --sam-xa, single-end data, rows without CIGAR (--min-alignment-score), read pairs of more than 16
alignment pairs (the per-pair std::sort really permutes), per-read LCA incl. entries the tree does not know."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(synth, T, seed, n_pairs, many_strains=False, read_len=100):
    rng = np.random.default_rng(seed)
    if many_strains:      # a read aligns to ~20 near-identical genomes: groups of > 16 alignment pairs, tied scores, multi-hit MAPQ
        root = synth.make_genomes(seed, 1, 1, 8000)[0]
        genomes = [synth.mutate(rng, root, 0.003, 0.0002) for _ in range(20)] + synth.make_genomes(seed + 5, 2, 2, 8000)
    else:
        genomes = synth.make_genomes(seed, 3, 3, 20000, shared_segment=3000)
    reads, _ = synth.make_paired_reads(seed + 1, genomes, n_pairs, read_len=read_len, frag_mean=300, frag_sd=40, sub_rate=0.02,
                                       indel_rate=0.004, edge_frac=0.05, n_rate=0.002)
    rb, gb = synth.to_bytes(reads), synth.to_bytes(genomes)
    quals = [bytes(rng.integers(35, 74, len(b), dtype=np.uint8)) for b in rb]
    ids = [b"frag%d/x" % (i % n_pairs) if i % 7 else b"f%d" % (i % n_pairs) for i in range(2 * n_pairs)]
    genes = [[(100 + 900 * k, 900 * k + 900, b"gene%d" % k if k % 4 else b"", b"WP_%d" % k if k % 3 else b"", b"product %d" % k if k % 5 else b"")
              for k in range(8)] for _ in gb]
    tax_ids = [1000 + i if i % 6 != 5 else (0 if i % 12 == 5 else 777777) for i in range(len(gb))]   # 0 and an id the tree lacks
    I = T.Index(gb, locus_tags=[b"NC_%06d.%d" % (i, i % 3) for i in range(len(gb))], taxonomy_ids=tax_ids, genes=genes)
    recs = [(1, 1, b"root", b"no rank"), (2, 1, b"Bacteria", b"superkingdom")]
    for i in range(len(gb)):
        if i % 3 == 0:
            recs.append((100 + i // 3, 2, b"species %d" % (i // 3), b"species"))
        recs.append((1000 + i, 100 + i // 3, b"strain %d" % i, b"strain"))
    taxdb_text = b"".join(b"%d\n%d\n%s\n%s\n" % r for r in recs)
    return rb, gb, quals, ids, I, taxdb_text


def _device_and_host_text(kslam, T, X, ST, rb, gb, quals, ids, I, taxdb_text, paired=True, num_alignments=10, sam_xa=False,
                          score_threshold=0, report_cigar=True, pseudo=True):
    n = len(rb)
    R = T.Reads(rb, quals, ids)
    c = kslam.Context(score_threshold=score_threshold, report_cigar=report_cigar)
    c.set_index(gb)
    c.load_reads(rb)
    n_out, n_cig = c.align_resident()
    c.load_qualities(quals)
    st = c.pair_screen(paired=paired, score_threshold=score_threshold, stages=7 if pseudo else 3)
    assert not pseudo or (st["stages_done"] & 4)
    det = md = None
    if report_cigar:
        c.row_details(of_pairs=True)
        det, md = c.take_row_details(n_out)
    ov, cg = c.fetch_results(n_out, n_cig)
    rp, pr = c.take_pairs()                              # BEFORE the device sorts them: the host does its own sort below
    tax = X.TaxDB(taxdb_text)
    ST.set_annotations(c, I, tax)
    ST.load_read_ids(c, ids)
    sam, per, tids = ST.sam_text(c, paired=paired, num_alignments=num_alignments, sam_xa=sam_xa, want_sam=True, want_per_read=True)
    srp, spr = c.take_pairs()                            # now in writeSAMOutputPairs' order
    # ---- the host's text for the same rows ----
    P = T.TailParams.default(paired=paired, pseudo_assembly=False, num_sam_alignments=num_alignments, sam_xa=sam_xa,
                             score_threshold=score_threshold, report_cigar=report_cigar)
    chunks = []
    hrp, hpr = rp.copy(), pr.copy()
    T.tail_finish_rows(P, R, I, ov, cg, det, md, hrp, hpr, chunks.append)
    exp_tax, exp_per = tax.classify(P, R, I, hrp, hpr)
    tax.close()
    c.close()
    return (sam, per, tids, srp, spr), (b"".join(chunks), exp_per, exp_tax, hrp, hpr), int(rp["count"].max())


@pytest.mark.parametrize("kw", [{}, {"num_alignments": 1}, {"num_alignments": 3, "sam_xa": True}, {"paired": False},
                                {"score_threshold": 150}, {"report_cigar": False}, {"pseudo": False}])
def test_device_text_equals_the_host_text(kslam, synth, kw):
    T = importlib.import_module("kslam_amd.tail")
    X = importlib.import_module("kslam_amd.taxonomy")
    ST = importlib.import_module("kslam_amd.samtext")
    rb, gb, quals, ids, I, taxdb_text = _case(synth, T, 31, 2500)
    if kw.get("paired") is False:
        rb, quals, ids = rb[:2500], quals[:2500], ids[:2500]
    got, exp, _ = _device_and_host_text(kslam, T, X, ST, rb, gb, quals, ids, I, taxdb_text, **kw)
    assert len(exp[0]) > 200000 and exp[0].count(b"\n") > 2000
    assert got[0] == exp[0]
    assert got[1] == exp[1] and got[2].tolist() == exp[2].tolist() and len(set(exp[2].tolist())) > 5
    assert got[3].tobytes() == exp[3].tobytes() and got[4].tobytes() == exp[4].tobytes()     # the pairs come back sorted


def test_device_text_with_large_tied_groups_and_multi_hit_qualities(kslam, synth):
    T = importlib.import_module("kslam_amd.tail")
    X = importlib.import_module("kslam_amd.taxonomy")
    ST = importlib.import_module("kslam_amd.samtext")
    rb, gb, quals, ids, I, taxdb_text = _case(synth, T, 57, 1200, many_strains=True)
    for na in (10, 40):
        got, exp, biggest = _device_and_host_text(kslam, T, X, ST, rb, gb, quals, ids, I, taxdb_text, num_alignments=na, pseudo=False)
        assert biggest > 16
        assert got[0] == exp[0] and got[1] == exp[1] and got[2].tolist() == exp[2].tolist()
        assert got[4].tobytes() == exp[4].tobytes()
        quals_seen = {ln.split(b"\t")[4] for ln in exp[0].split(b"\n") if ln}
        assert len(quals_seen) > 4            # not just 0 and 50: the host-evaluated qualities are exercised
