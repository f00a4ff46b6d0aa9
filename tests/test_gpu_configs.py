"""BASELINE configs[2], configs[3] (its one-GPU shape) and configs[4] at their full size, through the whole chain:

  configs[2]  10 M x 2 x 150 bp read pairs vs the bacterial + viral database (1 250 genomes of 4 Mb + 10 000 of 5-200 kb,
              SURVEY.md section 8d), full pipeline incl. SAM output, pseudo-assembly on (the reference's default)
  configs[3]  ONE batch of 10 M x 2 x 150 bp pairs (the reference's --num-reads-at-once default; configs[3] is four of them)
              split EIGHT ways, every shard aligned on the box's one GPU in a context of its own; shard -> export in batch
              terms -> the placement of include/kslam_comm.h -> merged arrays; the tail sharded the same way (insert sizes
              of all shards, pseudo-assembly with the entries partitioned over the shards); everything compared with ONE
              context that took the whole batch (test_config3_... at the bottom)
  configs[4]  10 M x 2 x 250 bp read pairs vs the bacterial database

FASTQ text in host memory -> k-slam_amd/stream.py (the reference's batch loop, src/SLAM.h:159-268: ONE batch of 10 M
pairs, the reference's --num-reads-at-once default) -> kslam_submit_batch_fastq_text (GPU: FASTQ index, alignment in
several internal chunks, pairing, insert-size statistics, screens, pseudo-assembly, per-row walk) -> SAM text written to
/dev/shm.  Checked three ways:

* the reference's own structural expectation (src/Tests.h:161-264, :321-330) on EVERY read: planted (entry, rel,
  revComp) found, score bounds, no overlap for reads that are not from the database, rows sorted;
* run-to-run identity of the rows and of the SAM file;
* oracle parity on a sub-database INCLUDING THE SAM TEXT: join, dedupe and SW are independent per (read, entry); pairing,
  the screens and the SAM records are per read pair once the batch's insert-size limit is fixed
  (src/PairedOverlap.h:314-436, src/SAM.h:443-512) and pseudo-assembly is per entry (src/PairedOverlap.h:480-582).  So for
  the read pairs drawn from a few entries -- which are ALL the read pairs that hit those entries -- the rows and the SAM
  lines must be, byte for byte, what the oracle chain gives for those reads against those entries when it is handed the
  batch's limit.
"""
import importlib
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPECIES, STRAINS, GENOME_LEN = 250, 5, 4_000_000
PAIRS = int(os.environ.get("KSLAM_TEST_CONFIG_PAIRS", "10000000"))


def _file_crc(path):
    crc, n = 0, 0
    with open(path, "rb") as f:
        while True:
            b = f.read(64 << 20)
            if not b:
                break
            crc = zlib.crc32(b, crc)
            n += len(b)
    return crc, n


def _run_config(kslam, oracle, tmp_path, n_viral, read_len, by_length, pseudo):
    import torch
    assert torch.cuda.is_available(), "torch sees no HIP device"
    W = importlib.import_module("kslam_amd.workload")
    T = importlib.import_module("kslam_amd.tail")
    S = importlib.import_module("kslam_amd.stream")
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    db, offs = W.make_database(dev, gen, SPECIES, STRAINS, GENOME_LEN, n_viral=n_viral)
    n_entries = len(offs) - 1
    ctx = kslam.Context()
    torch.cuda.synchronize()   # torch wrote the database on its own stream
    ctx.set_index_device(n_entries, db.data_ptr(), offs)
    n_pairs = PAIRS
    reads, truth = W.make_batch_in_pieces(dev, gen, db, offs, n_pairs, read_len, by_length=by_length)
    # ---- the two FASTQ texts, built on the GPU piece by piece, in page-locked host memory ----
    rec_w = 2 + W.ID_DIGITS + 3 + read_len + 3 + read_len + 1
    h = [kslam.HostBuffer(n_pairs * rec_w + 64) for _ in range(2)]
    quals = []
    qgen = torch.Generator(device=dev)
    qgen.manual_seed(77)
    step = 1_250_000
    for mate in (0, 1):
        view = torch.from_numpy(h[mate].a[:n_pairs * rec_w].reshape(n_pairs, rec_w))
        for lo in range(0, n_pairs, step):
            hi = min(n_pairs, lo + step)
            txt, q = W.fastq_text(reads[mate * n_pairs + lo:mate * n_pairs + hi], mate + 1, first_pair=lo, gen=qgen)
            view[lo:hi].copy_(txt)
            quals.append(q)
            del txt
    quals = torch.cat(quals, 0)                                      # [2 n, L], block layout like `reads`
    torch.cuda.synchronize()
    tax_ids = np.arange(1, n_entries + 1, dtype=np.uint32)
    I = T.IndexArrays(np.zeros(1, dtype=np.uint8), offs, taxonomy_ids=tax_ids)   # no copy of the database on the host
    P = T.TailParams.default(pseudo_assembly=pseudo)

    # ---- the sub-database: first species, the species across the 2^32-byte boundary, the last bacterial species,
    # and (configs[2]) viral entries: the first five, the last five, the five shortest, the five longest ----
    cross = (int(np.searchsorted(offs, np.uint64(1) << np.uint64(32))) - 1) // STRAINS
    species = sorted({0, min(cross, SPECIES - 1), SPECIES - 1})
    sub_entries = [s * STRAINS + k for s in species for k in range(STRAINS)]
    if n_viral:
        lens = np.diff(offs.astype(np.int64))[SPECIES * STRAINS:]
        by_len = np.argsort(lens, kind="stable")
        vi = sorted(set(range(5)) | set(range(n_viral - 5, n_viral)) | set(by_len[:5].tolist()) | set(by_len[-5:].tolist()))
        sub_entries += [SPECIES * STRAINS + v for v in vi]
    sub_entries = np.array(sorted(sub_entries))
    t_entry = truth["entry"][:n_pairs].cpu().numpy()
    sub_pairs = np.sort(np.concatenate([np.nonzero(np.isin(t_entry, sub_entries))[0], np.nonzero(t_entry < 0)[0][:200]]))
    m = len(sub_pairs)
    assert m > 1000
    keep = {}

    def before(k, ov, cg, det, md, rp, pr, pst, rv):
        # the sub-sample's SAM text from the batch as the GPU returned it (copies: the writer sorts in place)
        mask = np.isin(rp["r1_read"], sub_pairs)
        out = []
        Pw = T.TailParams.default(pseudo_assembly=False) if pst["stages_done"] & 4 or not pseudo else P
        T.tail_finish_rows(Pw, rv, I, ov, cg, det, md, rp[mask].copy(), pr.copy(), sink=out.append)
        keep.update(sam=b"".join(out), ov=ov.copy(), limit=pst["max_insert_size"], stats=dict(pst),
                    n_groups=int(mask.sum()))

    def run(tag):
        path = "/dev/shm/kslam_test_%s_%d.sam" % (tag, os.getpid())
        fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
        try:
            res = S.classify_stream(ctx, I, h[0].ptr, n_pairs * rec_w, h[1].ptr, n_pairs * rec_w, n_pairs, P, sam_fd=fd,
                                    before_batch=before)
        finally:
            os.close(fd)
        crc = _file_crc(path)
        os.unlink(path)
        return res, crc

    res_a, crc_a = run("a")
    first = dict(keep)
    res_b, crc_b = run("b")
    assert res_a["pairs"] == n_pairs and len(res_a["batches"]) == 1
    b0 = res_a["batches"][0]
    print("batch:", b0, "sam bytes", crc_a[1], "seconds", round(res_a["seconds"], 2), round(res_b["seconds"], 2))
    if pseudo:
        assert b0["pseudo_assembly_on"] == "gpu"
    # ---- run-to-run identity: rows, sub-sample text, the whole SAM file ----
    assert crc_a == crc_b and crc_a[1] > 150 * n_pairs
    assert first["ov"].tobytes() == keep["ov"].tobytes() and first["sam"] == keep["sam"]
    # ---- planted truth on every read ----
    ov_dev = torch.from_numpy(first["ov"].view(np.uint8).reshape(-1)).to(dev)
    verdict = W.check_against_truth(ov_dev, None, truth, read_len)
    print(verdict)
    assert verdict["planted_expected"] > 1.3 * n_pairs and verdict["ok"], verdict
    del ov_dev

    # ---- oracle parity on the sub-database: rows, then the SAM text ----
    idx = torch.from_numpy(np.concatenate([sub_pairs, sub_pairs + n_pairs])).to(dev)
    rd = reads[idx].cpu().numpy()
    qd = quals[idx].cpu().numpy()
    sub_reads = [rd[i].tobytes() for i in range(2 * m)]
    sub_quals = [qd[i].tobytes() for i in range(2 * m)]
    sub_ids = [W.pair_id(int(i)) for i in sub_pairs] * 2
    sub_bases = [db[int(offs[e]):int(offs[e + 1])].cpu().numpy().tobytes() for e in sub_entries]
    exp, ecig, _ = oracle.align_to_database(sub_reads, sub_bases)
    ov = first["ov"]
    local = np.full(2 * n_pairs, -1, dtype=np.int64)
    local[sub_pairs] = np.arange(m)
    local[sub_pairs + n_pairs] = m + np.arange(m)
    rows = ov[local[ov["read"]] >= 0]
    eloc = np.full(n_entries, -1, dtype=np.int64)
    eloc[sub_entries] = np.arange(len(sub_entries))
    assert (eloc[rows["entry"]] >= 0).all()                 # these reads hit nothing outside their own entries
    others = ov[(local[ov["read"]] < 0)]
    assert not np.isin(others["entry"], sub_entries).any()  # and nothing else hits the sub-database: it is closed
    assert len(rows) == len(exp) and len(rows) > 3 * m * 0.5
    assert (local[rows["read"]] == exp["read"]).all() and (eloc[rows["entry"]] == exp["entry"]).all()
    for f in ("rel", "revcomp", "score", "ref_begin", "ref_end", "query_begin", "query_end"):
        bad = np.nonzero(rows[f] != exp[f])[0]
        assert len(bad) == 0, "%s differs at %s" % (f, bad[:5])
    has = rows["cigar_len"] > 0                               # the lanes compute CIGARs for the rows the SAM text needs
    assert (rows["cigar_len"][has] == exp["cigar_len"][has]).all() and has.sum() > 0.2 * len(rows)
    oR = T.Reads(sub_reads, sub_quals, sub_ids)
    oI = T.Index(sub_bases, locus_tags=[b"entry%d" % e for e in sub_entries], taxonomy_ids=tax_ids[sub_entries])
    oracle.tail_force_insert_limit(first["limit"])
    try:
        esam = oracle.tail_sam(P, oR.view, oI.view, exp, ecig)
    finally:
        oracle.tail_force_insert_limit(None)
    got = first["sam"]
    if got != esam:
        gl, el = got.split(b"\n"), esam.split(b"\n")
        for a, b in zip(gl, el):
            if a != b:
                print("first differing SAM line:\n got", a[:300], "\n exp", b[:300])
                break
        print(len(gl), len(el))
    assert got == esam
    n_lines = got.count(b"\n")
    gapped = sum(1 for ln in got.split(b"\n") if b"\t" in ln and (b"I" in ln.split(b"\t")[5] or b"D" in ln.split(b"\t")[5]))
    print("sub-database: %d read pairs, %d rows, %d SAM lines (%d with indels), limit %d" % (m, len(rows), n_lines, gapped, first["limit"]))
    assert n_lines > 1.5 * (m - 200) and gapped > 100
    for x in h:
        x.close()
    ctx.close()
    return b0


def test_config2_bacterial_plus_viral_10m_pairs_full_pipeline(kslam, oracle, tmp_path):
    b0 = _run_config(kslam, oracle, tmp_path, n_viral=10_000, read_len=150, by_length=True, pseudo=True)
    assert b0["overlaps"] > 2 * PAIRS


def test_config4_250bp_10m_pairs_full_pipeline(kslam, oracle, tmp_path):
    b0 = _run_config(kslam, oracle, tmp_path, n_viral=0, read_len=250, by_length=False, pseudo=True)
    assert b0["overlaps"] > 2 * PAIRS


def test_config3_one_10m_pair_batch_split_eight_ways_on_one_gpu(kslam):
    """BASELINE configs[3]'s arithmetic at its real size -- 10 M pairs, ~80 M rows, a multi-GB CIGAR pool, 1.25 M pairs per
    shard -- on the one GPU a box has: eight sibling contexts (they borrow the index) stand for the eight ranks.
      alignment   every shard's records, exported in batch terms (kslam_export_shard_device) into the places
                  kslam_comm_gather_plan assigns, ARE the rows and the CIGAR pool of one context that aligned the whole batch
      rank-0 tail a context that adopts the merged arrays writes the whole batch's SAM text and _PerRead lines
      sharded     phase A per shard, all shards' insert sizes to every shard, phase B, pseudo-assembly with entry e owned by
                  shard e mod 8 (kslam_pseudo_route / _owned / _return; the exchanges are concatenations here, RCCL's part
                  is tests/test_gpu_multi.py's), per-row walk, text: the shards' SAM / _PerRead text in shard order IS
                  the whole batch's (CRC-32 of 2.4 GB).
    What this cannot show is time: the eight shards share one GPU."""
    import torch
    assert torch.cuda.is_available(), "torch sees no HIP device"
    W = importlib.import_module("kslam_amd.workload")
    T = importlib.import_module("kslam_amd.tail")
    X = importlib.import_module("kslam_amd.taxonomy")
    ST = importlib.import_module("kslam_amd.samtext")
    Cm = importlib.import_module("kslam_amd.comm")
    kd = importlib.import_module("kslam_amd.dist")
    dev = torch.device("cuda", 0)
    N, L, n = 8, 150, PAIRS - PAIRS % 8
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    db, offs = W.make_database(dev, gen, SPECIES, STRAINS, GENOME_LEN)
    n_entries = len(offs) - 1
    whole = kslam.Context()
    torch.cuda.synchronize()
    whole.set_index_device(n_entries, db.data_ptr(), offs)
    reads, _ = W.make_batch_in_pieces(dev, gen, db, offs, n, L, pieces=N, with_truth=False)
    qgen = torch.Generator(device=dev)
    qgen.manual_seed(4242)
    qual = torch.randint(33 + 20, 33 + 41, (2 * n * L + 64,), generator=qgen, device=dev, dtype=torch.uint8)
    torch.cuda.synchronize()
    tax_text, entry_tax = W.taxonomy(SPECIES, STRAINS, 0)
    index_view = T.IndexArrays(np.zeros(1, dtype=np.uint8), offs, taxonomy_ids=entry_tax)
    taxdb = X.TaxDB(tax_text)
    keep = []

    def load(c, rd, q, n_loc, first_pair):
        c.load_reads_device(2 * n_loc, rd.data_ptr(), np.arange(2 * n_loc + 1, dtype=np.uint64) * np.uint64(L))
        c.load_qualities_device(q.data_ptr())
        rv = W.ids_view(T, n_loc, L, first_pair=first_pair)
        ST.set_annotations(c, index_view, taxdb)
        c._chk(ST.lib().kslam_load_read_ids(c._h, rv._keep[0].ctypes.data, rv._keep[1].ctypes.data))
        keep.append((rd, q, rv))

    def text_of(contexts, tag):
        """SAM records and per-read lines of the contexts' read pairs, in order, into two files -> their (crc, bytes)"""
        sam_path = "/dev/shm/kslam_test_c3_%s_%d.sam" % (tag, os.getpid())
        fd, pfd = os.open(sam_path, os.O_RDWR | os.O_CREAT | os.O_TRUNC), os.open(sam_path + "_PerRead", os.O_RDWR | os.O_CREAT | os.O_TRUNC)
        wr = T.SamWriter(fd)
        n_tax = 0
        try:
            for c in contexts:
                c.row_details(of_pairs=True)
                _, _, tax = ST.sam_text_to_files(c, wr, pfd, paired=True, num_alignments=10, sam_xa=False, want_per_read=True)
                n_tax += len(tax)
        finally:
            wr.close()
            os.close(fd)
            os.close(pfd)
        out = _file_crc(sam_path), _file_crc(sam_path + "_PerRead"), n_tax
        os.unlink(sam_path)
        os.unlink(sam_path + "_PerRead")
        return out

    # ---- ONE context, the whole batch ----
    load(whole, reads, qual, n, 0)
    n_out, n_cig = whole.align_resident()
    ov_w = torch.empty(n_out * 48, dtype=torch.uint8, device=dev)
    cg_w = torch.empty(max(n_cig, 1) * 4, dtype=torch.uint8, device=dev)
    whole.copy_results_device(ov_w.data_ptr(), cg_w.data_ptr())
    st_w = whole.pair_screen(paired=True, stages=7)
    assert st_w["stages_done"] & 4 and n_out > 7 * n
    exp = text_of([whole], "whole")
    print("one context: %d rows, %d CIGAR words, SAM %d bytes, _PerRead %d bytes, limit %d" % (n_out, n_cig, exp[0][1], exp[1][1], st_w["max_insert_size"]))
    assert exp[0][1] > 200 * n and exp[2] == st_w["n_read_pairs"]

    # ---- the same batch in eight shards, each in its own context ----
    bounds = kd.shard_bounds(n, N)
    shards = []
    for lo, hi in bounds:
        c = whole.sibling()
        rd = torch.cat([reads[lo:hi], reads[n + lo:n + hi]]).contiguous()
        q = torch.cat([qual[lo * L:hi * L], qual[(n + lo) * L:(n + hi) * L], torch.zeros(64, dtype=torch.uint8, device=dev)]).contiguous()
        torch.cuda.synchronize()
        load(c, rd, q, hi - lo, lo)
        c.align_resident()
        shards.append(c)
    counts = [c.shard_counts_device(hi - lo) for c, (lo, hi) in zip(shards, bounds)]
    row1, row2, op1, op2, (tot_rows, tot_ops) = Cm.gather_plan(counts)
    assert (tot_rows, tot_ops) == (n_out, n_cig)
    rows = torch.empty(tot_rows * 48, dtype=torch.uint8, device=dev)
    pool = torch.empty(max(tot_ops, 1) * 4, dtype=torch.uint8, device=dev)
    for r, (c, (lo, hi)) in enumerate(zip(shards, bounds)):
        c.export_shard_device(hi - lo, lo, n, op1[r], op2[r], rows.data_ptr() + 48 * row1[r], rows.data_ptr() + 48 * row2[r],
                              pool.data_ptr() + 4 * op1[r], pool.data_ptr() + 4 * op2[r])
    torch.cuda.synchronize()
    assert torch.equal(rows, ov_w) and torch.equal(pool[:tot_ops * 4], cg_w[:n_cig * 4])
    del ov_w, cg_w

    # ---- rank 0's form: a context adopts the merged arrays and runs the batch-global tail ----
    tail_ctx = whole.sibling()
    load(tail_ctx, reads, qual, n, 0)
    tail_ctx.adopt_results_device(rows.data_ptr(), tot_rows, pool.data_ptr(), tot_ops)
    st_t = tail_ctx.pair_screen(paired=True, stages=7)
    assert {k: st_t[k] for k in ("n_read_pairs", "n_pairs", "max_insert_size", "stages_done")} == \
           {k: st_w[k] for k in ("n_read_pairs", "n_pairs", "max_insert_size", "stages_done")}
    assert text_of([tail_ctx], "adopted") == exp
    tail_ctx.close()
    del rows, pool

    # ---- the tail sharded like the alignment ----
    ins = [kd.device_bytes(*(lambda p, k: (p, k * 4))(*c.pair_phase_a(True, 0)), dev) for c in shards]
    all_ins = torch.cat(ins)
    torch.cuda.synchronize()
    limits = []
    for c in shards:
        stats, _, _ = c.pair_phase_b(all_ins.data_ptr() if all_ins.numel() else None, all_ins.numel() // 4, 0.95, 3)
        limits.append(stats["max_insert_size"])
    assert set(limits) == {st_w["max_insert_size"]}
    routed = []
    for c in shards:
        d_heads, cn = c.pseudo_route(N)
        routed.append((kd.device_bytes(d_heads, sum(cn) * 16, dev), cn))
    back = [[None] * N for _ in range(N)]
    for d, c in enumerate(shards):
        got = torch.cat([h[sum(cn[:d]) * 16:sum(cn[:d + 1]) * 16] for h, cn in routed]).contiguous()
        torch.cuda.synchronize()
        k = got.numel() // 16
        scores = kd.device_bytes(c.pseudo_owned(got.data_ptr() if k else None, k), k * 4, dev)
        at = 0
        for src, (_, cn) in enumerate(routed):
            back[src][d] = scores[at:at + cn[d] * 4]
            at += cn[d] * 4
    owned = [sum(cn[d] for _, cn in routed) for d in range(N)]
    print("alignment pairs per shard:", [sum(cn) for _, cn in routed], "owned after routing:", owned)
    n_rp = 0
    for src, c in enumerate(shards):
        mine = torch.cat(back[src]).contiguous()
        torch.cuda.synchronize()
        s2 = c.pseudo_return(mine.data_ptr() if mine.numel() else None, mine.numel() // 4, 0.95)
        assert s2["stages_done"] & 4
        n_rp += s2["n_read_pairs"]
    assert n_rp == st_w["n_read_pairs"]
    assert max(owned) < 1.5 * sum(owned) / N                    # entry mod N balances the stage
    assert text_of(shards, "sharded") == exp
    for c in shards:
        c.close()
    whole.close()
    taxdb.close()

