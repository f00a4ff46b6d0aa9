"""Parity at the bench's own scale: BASELINE configs[1] (1 M x 2 x 150 bp vs the 5.0 Gb database) and
configs[4] (2 x 250 bp), generated exactly as bench.py generates them (k-slam_amd/workload.py).

At this size the library runs in a regime no small test reaches: a 27-bit bucket table, three radix
passes per batch, 312 M genome k-mers, entry byte offsets beyond 2^32, ~25 M raw overlaps through the
single-pass join.  Two checks:

* the reference's own structural expectation (src/Tests.h:161-264, :321-330) on EVERY read of the
  batch: planted (entry, rel, revComp) present, score bounds, error-free reads = <L>M;
* oracle parity on a sub-database: the join, the dedupe and SW are independent per (read, entry)
  (src/Overlap.h:163-197, :79-85; src/SmithWaterman.h:198-231), so the GPU's rows for the reads drawn
  from a few species must equal, field by field and CIGAR op by CIGAR op, what the oracle gives for
  those reads against just those species' entries, after renumbering reads and entries.  The species
  are chosen at the start of the database, across the 2^32 byte boundary and at its end.
"""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPECIES, STRAINS, GENOME_LEN, PAIRS = 250, 5, 4_000_000, 1_000_000


@pytest.fixture(scope="module")
def big(kslam):
    import torch
    # no skip on a GPU box: if torch cannot see the device this must fail, not pass silently
    assert torch.cuda.is_available(), "torch sees no HIP device (did another HIP runtime open it first?)"
    assert torch.cuda.get_device_properties(0).total_memory > 100e9, "needs the 5 Gb database resident (MI355X: 288 GB)"
    W = importlib.import_module("kslam_amd.workload")
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)                                   # bench.py's database seed
    db, offs = W.make_database(dev, gen, SPECIES, STRAINS, GENOME_LEN)
    torch.cuda.synchronize()
    ctx = kslam.Context()
    ctx.set_index_device(len(offs) - 1, db.data_ptr(), offs)
    yield {"W": W, "dev": dev, "gen": gen, "db": db, "offs": offs, "ctx": ctx, "torch": torch}
    ctx.close()


def _align(big, read_len, pairs):
    torch, W, ctx = big["torch"], big["W"], big["ctx"]
    big["gen"].manual_seed(2)                            # bench.py's read seed on rank 0
    reads, truth = W.make_reads(big["dev"], big["gen"], big["db"], big["offs"], pairs, read_len=read_len,
                                with_truth=True)
    torch.cuda.synchronize()
    n_reads = reads.shape[0]
    roffs = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
    ctx.load_reads_device(n_reads, reads.data_ptr(), roffs)
    n_out, n_cig = ctx.align_resident()
    ov = torch.empty(n_out * 48, dtype=torch.uint8, device=big["dev"])
    cg = torch.empty(max(n_cig, 1) * 4, dtype=torch.uint8, device=big["dev"])
    ctx.copy_results_device(ov.data_ptr(), cg.data_ptr())
    return reads, truth, ov, cg[:n_cig * 4].view(torch.int32), ctx.timings()


def _sub_database_parity(big, kslam, oracle, reads, truth, ov_dev, cg_dev, species, max_pairs):
    torch, offs = big["torch"], big["offs"]
    n_pairs = reads.shape[0] // 2
    entries = np.array(sorted(s * STRAINS + k for s in species for k in range(STRAINS)))
    t_entry = truth["entry"][:n_pairs].cpu().numpy()
    from_sub = np.nonzero(np.isin(t_entry, entries))[0][:max_pairs]
    not_in_db = np.nonzero(t_entry < 0)[0][:200]
    pairs = np.sort(np.concatenate([from_sub, not_in_db]))
    m = len(pairs)
    assert m > 1000
    rd = reads.cpu().numpy()
    sub_reads = [rd[i].tobytes() for i in pairs] + [rd[i + n_pairs].tobytes() for i in pairs]
    db = big["db"]
    sub_entries = [db[int(offs[e]):int(offs[e + 1])].cpu().numpy().tobytes() for e in entries]
    assert int(offs[entries[-1]]) > 1 << 32              # the last species sits beyond 4 GiB
    exp, ecig, _ = oracle.align_to_database(sub_reads, sub_entries)

    ov = np.frombuffer(ov_dev.cpu().numpy().tobytes(), dtype=kslam.OVERLAP_DT)
    cg = cg_dev.cpu().numpy().view(np.uint32)
    local = np.full(2 * n_pairs, -1, dtype=np.int64)
    local[pairs] = np.arange(m)
    local[pairs + n_pairs] = m + np.arange(m)
    rows = ov[local[ov["read"]] >= 0]
    eloc = np.full(len(offs) - 1, -1, dtype=np.int64)
    eloc[entries] = np.arange(len(entries))
    # every hit of these reads lies in their own species (a chance 32-mer match elsewhere in 5 Gb has
    # probability ~1e-9 per read): a row outside the sub-database would be a wrong overlap
    assert (eloc[rows["entry"]] >= 0).all()
    got_read, got_entry = local[rows["read"]], eloc[rows["entry"]]
    order = np.lexsort((rows["revcomp"], rows["rel"], got_entry, got_read))
    assert (order == np.arange(len(rows))).all()         # the renumbering is monotone: order is kept
    assert len(rows) == len(exp), (len(rows), len(exp))
    assert (got_read == exp["read"]).all() and (got_entry == exp["entry"]).all()
    for f in ("rel", "revcomp", "score", "ref_begin", "ref_end", "query_begin", "query_end", "cigar_len"):
        bad = np.nonzero(rows[f] != exp[f])[0]
        assert len(bad) == 0, "%s differs at %s: got %s exp %s" % (f, bad[:5], rows[bad[:5]], exp[bad[:5]])
    # CIGAR pools: compare op by op through a flat gather
    ln = rows["cigar_len"].astype(np.int64)
    tot = int(ln.sum())
    start = np.repeat(np.cumsum(ln) - ln, ln)
    inner = np.arange(tot) - start
    a = cg[np.repeat(rows["cigar_off"].astype(np.int64), ln) + inner]
    b = ecig[np.repeat(exp["cigar_off"].astype(np.int64), ln) + inner]
    assert np.array_equal(a, b)
    return len(rows), int((ln > 1).sum())


@pytest.mark.parametrize("read_len", [150, 250])
def test_bench_workload_truth_and_oracle_parity(big, kslam, oracle, read_len):
    reads, truth, ov, cg, tm = _align(big, read_len, PAIRS)
    # the regime: one chunk of 2^30 k-mers at most, 3 radix passes, > 300 M genome k-mers
    assert tm["n_genome_kmers"] > 300_000_000 and tm["sort_passes"] == 3
    assert tm["n_overlaps_raw"] > 20_000_000
    res = big["W"].check_against_truth(ov, cg, truth, read_len)
    print(res)
    assert res["planted_expected"] > 1_300_000
    assert res["ok"], res
    # first species, the species across the 2^32-byte boundary, the last species
    cross = (int(np.searchsorted(big["offs"], np.uint64(1) << np.uint64(32))) - 1) // STRAINS
    n_rows, n_gapped = _sub_database_parity(big, kslam, oracle, reads, truth, ov, cg, [0, cross, SPECIES - 1], 2500)
    assert n_rows > 8000 and n_gapped > 300, (n_rows, n_gapped)


def test_run_to_run_identity_at_scale(big):
    """the single-pass join writes in scheduling order and the band tiers hand candidates on with
    atomics: the RESULT must not depend on either"""
    torch = big["torch"]
    reads, truth, ov, cg, _ = _align(big, 150, PAIRS)
    ctx = big["ctx"]
    n_out, n_cig = ctx.align_resident()
    ov2 = torch.empty(n_out * 48, dtype=torch.uint8, device=big["dev"])
    cg2 = torch.empty(max(n_cig, 1) * 4, dtype=torch.uint8, device=big["dev"])
    ctx.copy_results_device(ov2.data_ptr(), cg2.data_ptr())
    assert ov2.numel() == ov.numel() and torch.equal(ov, ov2)
    assert torch.equal(cg, cg2[:n_cig * 4].view(torch.int32))


def test_repeat_rich_database_truth_and_oracle_parity(kslam, oracle):
    """The same 5 Gb database with what real bacterial sets carry: an rRNA-like 1.5 kb segment in five copies per genome
    (species variants sharing their conserved 32-mers) and an insertion element in every third species
    (k-slam_amd/workload.py).  Reads from such a copy meet thousands of genome k-mers: k_join_fill's long pile-ups and its
    > 4096-overlap workgroups, (read, entry) segments with several copies in k_dedupe_flags, the rerun of the join when its
    output outgrows the buffer sized from the filter's survivors -- none of which the i.i.d. database reaches.  The
    reference emits the full nG x nR cross product there (src/Overlap.h:163-197; removeLowQualityOverlaps is commented
    out, :292), so must this.  Checks: planted truth on every read; and oracle parity, row by row and CIGAR op by CIGAR op,
    of every row that pairs a sampled read -- the reads of three species plus the 1 500 reads with the most hits in the
    batch, wherever they come from -- with an entry of those species (join, dedupe and SW are independent per (read,
    entry): no closure of the sample is needed for the rows)."""
    import torch
    W = importlib.import_module("kslam_amd.workload")
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    db, offs = W.make_database(dev, gen, SPECIES, STRAINS, GENOME_LEN, repeats=True)
    ctx = kslam.Context()
    ctx.set_index_device(len(offs) - 1, db.data_ptr(), offs)
    gen.manual_seed(2)
    n_pairs = 1_000_000
    reads, truth = W.make_reads(dev, gen, db, offs, n_pairs, read_len=150, with_truth=True)
    torch.cuda.synchronize()
    ctx.load_reads_device(2 * n_pairs, reads.data_ptr(), np.arange(2 * n_pairs + 1, dtype=np.uint64) * np.uint64(150))
    n_out, n_cig = ctx.align_resident()
    tm = ctx.timings()
    ov_d = torch.empty(n_out * 48, dtype=torch.uint8, device=dev)
    cg_d = torch.empty(max(n_cig, 1) * 4, dtype=torch.uint8, device=dev)
    ctx.copy_results_device(ov_d.data_ptr(), cg_d.data_ptr())
    res = W.check_against_truth(ov_d, cg_d[:n_cig * 4].view(torch.int32), truth, 150)
    print(res, {k: round(v, 2) if isinstance(v, float) else v for k, v in tm.items()})
    assert res["ok"] and res["planted_expected"] > 1_600_000, res
    ov = np.frombuffer(ov_d.cpu().numpy().tobytes(), dtype=kslam.OVERLAP_DT)
    cg = cg_d[:n_cig * 4].cpu().numpy().view(np.uint32)
    per_read = np.bincount(ov["read"], minlength=2 * n_pairs)
    assert tm["n_overlaps"] > 12_000_000 and per_read.max() > 1000         # the repeat regime is really there
    species = [0, SPECIES // 2, SPECIES - 1]
    entries = np.array(sorted(s * STRAINS + k for s in species for k in range(STRAINS)))
    t_entry = truth["entry"].cpu().numpy()
    hot = np.argsort(-per_read, kind="stable")[:1500]
    from_sub = np.nonzero(np.isin(t_entry, entries))[0][:4000]
    sample = np.unique(np.concatenate([hot, from_sub]))
    rd = reads[torch.from_numpy(sample).to(dev)].cpu().numpy()
    sub_reads = [rd[i].tobytes() for i in range(len(sample))]
    sub_entries = [db[int(offs[e]):int(offs[e + 1])].cpu().numpy().tobytes() for e in entries]
    exp, ecig, _ = oracle.align_to_database(sub_reads, sub_entries)
    local = np.full(2 * n_pairs, -1, dtype=np.int64)
    local[sample] = np.arange(len(sample))
    eloc = np.full(len(offs) - 1, -1, dtype=np.int64)
    eloc[entries] = np.arange(len(entries))
    rows = ov[(local[ov["read"]] >= 0) & (eloc[ov["entry"]] >= 0)]
    assert len(rows) == len(exp) and len(rows) > 30_000, (len(rows), len(exp))
    assert (local[rows["read"]] == exp["read"]).all() and (eloc[rows["entry"]] == exp["entry"]).all()
    for f in ("rel", "revcomp", "score", "ref_begin", "ref_end", "query_begin", "query_end", "cigar_len"):
        bad = np.nonzero(rows[f] != exp[f])[0]
        assert len(bad) == 0, "%s differs at %s" % (f, bad[:5])
    ln = rows["cigar_len"].astype(np.int64)
    inner = np.arange(int(ln.sum())) - np.repeat(np.cumsum(ln) - ln, ln)
    assert np.array_equal(cg[np.repeat(rows["cigar_off"].astype(np.int64), ln) + inner],
                          ecig[np.repeat(exp["cigar_off"].astype(np.int64), ln) + inner])
    # several copies of one (read, entry): the dedupe's "|delta rel| < 3 against the last kept" over long segments
    key = rows["read"].astype(np.int64) * 4096 + rows["entry"]
    _, counts = np.unique(key, return_counts=True)
    assert counts.max() >= 5
    # run to run
    n2, c2 = ctx.align_resident()
    ov2 = torch.empty(n2 * 48, dtype=torch.uint8, device=dev)
    cg2 = torch.empty(max(c2, 1) * 4, dtype=torch.uint8, device=dev)
    ctx.copy_results_device(ov2.data_ptr(), cg2.data_ptr())
    assert n2 == n_out and torch.equal(ov_d, ov2) and torch.equal(cg_d[:n_cig * 4], cg2[:c2 * 4])
    ctx.close()
