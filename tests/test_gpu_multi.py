"""Read-sharded batches on the GPU (BASELINE configs[3]): the single-process
multi-device entry of the C ABI (kslam_multi_*), the device-side merge both hosts share
(kslam_merge_shards_device), and bench.py's --strong mode run through the RCCL code path at world
size 1.  Every result must be BYTE-identical to what one context returns for the whole batch.

The GPU boxes have one device, so the multi-device entry is given the same ordinal several times:
two or three contexts on one GPU, peer copies that are device-to-device copies.  The 8-GPU run itself
is the driver's (SCALE record)."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data(synth, seed, n_pairs, **kw):
    genomes = synth.make_genomes(seed, 3, 3, 30000, strain_sub=0.02, strain_indel=0.001, shared_segment=2000)
    reads, _ = synth.make_paired_reads(seed + 1, genomes, n_pairs, sub_rate=0.015, indel_rate=0.004, edge_frac=0.05, **kw)
    return synth.to_bytes(reads), synth.to_bytes(genomes)


@pytest.mark.parametrize("devices,n_pairs", [([0, 0], 3001), ([0, 0, 0], 2000), ([0], 500), ([0, 0, 0, 0, 0], 3)])
def test_multi_device_entry_equals_single_context(kslam, synth, devices, n_pairs):
    rb, gb = _data(synth, 600 + len(devices), n_pairs)
    exp, ecig = kslam.align_to_database(rb, gb)
    m = kslam.MultiContext(devices)
    m.set_index(gb)
    got, gcig = m.align_batch(rb, paired=True)
    again, acig = m.align_batch(rb, paired=True)
    m.close()
    assert len(exp) > n_pairs
    assert got.tobytes() == exp.tobytes() and gcig.tobytes() == ecig.tobytes()
    assert again.tobytes() == exp.tobytes() and acig.tobytes() == ecig.tobytes()


def test_multi_device_entry_unpaired_and_errors(kslam, synth):
    rb, gb = _data(synth, 650, 1000)
    rb = rb[:1501]                                    # an odd number of single reads
    exp, ecig = kslam.align_to_database(rb, gb)
    m = kslam.MultiContext([0, 0, 0])
    m.set_index(gb)
    got, gcig = m.align_batch(rb, paired=False)
    assert got.tobytes() == exp.tobytes() and gcig.tobytes() == ecig.tobytes()
    with pytest.raises(kslam.KslamError, match="even"):
        m.align_batch(rb, paired=True)
    e, c = m.align_batch([], paired=True)
    assert len(e) == 0 and len(c) == 0
    m.close()
    with pytest.raises(kslam.KslamError):
        kslam.MultiContext([0, 99])                   # no such device


def test_device_merge_equals_host_reassembly(kslam, synth):
    """kslam_merge_shards_device on gathered buffers == kslam_amd.dist.reassemble (numpy, covered on the
    CPU by the gloo test) == the single-context result, for an uneven three-way split."""
    import torch
    kd = importlib.import_module("kslam_amd.dist")
    n_pairs = 2500
    rb, gb = _data(synth, 700, n_pairs)
    c = kslam.Context()
    c.set_index(gb)
    exp, ecig = c.align_batch(rb)
    bounds = [(0, 700), (700, 701), (701, n_pairs)]
    parts = [c.align_batch(kd.local_reads(rb, n_pairs, lo, hi)) for lo, hi in bounds]
    host, hpool = kd.reassemble(parts, bounds, n_pairs, kslam.OVERLAP_DT)
    for f in ("read", "entry", "rel", "revcomp", "score", "ref_begin", "ref_end", "query_begin", "query_end", "cigar_len"):
        assert (host[f] == exp[f]).all(), f
    dev = torch.device("cuda", 0)
    rows = torch.from_numpy(np.concatenate([p[0] for p in parts]).view(np.uint8).copy()).to(dev)
    pool = torch.from_numpy(np.concatenate([p[1] for p in parts]).view(np.uint8).copy()).to(dev)
    out_rows, out_pool = torch.empty_like(rows), torch.empty_like(pool)
    torch.cuda.synchronize()
    c.merge_shards_device([(lo, hi, len(p[0]), len(p[1])) for (lo, hi), p in zip(bounds, parts)], n_pairs,
                          rows.data_ptr(), pool.data_ptr(), out_rows.data_ptr(), out_pool.data_ptr())
    got = np.frombuffer(out_rows.cpu().numpy().tobytes(), dtype=kslam.OVERLAP_DT)
    gpool = out_pool.cpu().numpy().view(np.uint32)
    assert got.tobytes() == exp.tobytes() and gpool.tobytes() == ecig.tobytes()
    with pytest.raises(kslam.KslamError, match="order"):
        c.merge_shards_device([(700, 800, 0, 0), (0, 700, 0, 0)], n_pairs, 0, 0, 0, 0)
    c.close()


def test_bench_strong_mode_through_the_rccl_path_at_world_1():
    """bench.py --strong with KSLAM_BENCH_FORCE_DIST=1: process group (RCCL), count exchange, gather,
    device merge -- everything the 8-GPU run does, with one rank; the merged batch must be byte-identical
    to the single-context result and pass the planted-truth checks."""
    env = dict(os.environ, KSLAM_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577",
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--strong", "--total-pairs", "40000",
                        "--species", "4", "--strains", "3", "--genome-len", "300000", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    v = line["hot_path"]["verified"]
    assert line["scaling"] == "strong" and line["config"]["pairs_per_batch"] == 40000
    assert v["ok"] and v["merged_equals_single_context"] and v["merged_unsorted_neighbours"] == 0
    assert v["planted_missing"] == 0 and v["planted_expected"] > 50000 and v["merged_rows"] == v["overlaps"]


@pytest.mark.parametrize("world,comm", [(2, "torch"), (4, "torch"), (8, "torch"), (2, "kslam"), (8, "kslam")])
def test_bench_strong_mode_with_several_ranks_on_one_gpu(world, comm):
    """bench.py --gpus N --strong, launched the way the driver launches it (torch.distributed.run, one process
    per rank), with the ranks SHARING the box's one GPU (KSLAM_BENCH_SHARE_GPU=1: gloo with host-staged pieces
    instead of RCCL, which refuses two ranks on a device).  Every rank aligns its pairs of the batch in its own
    context; count exchange, export in batch terms, placement on rank 0 and the verification are the code of a
    real N-GPU run.  Rank 0 then aligns the WHOLE batch in one context: the merged result must equal it byte
    for byte."""
    env = dict(os.environ, KSLAM_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "KSLAM_RCCL_LIB"):
        env.pop(k, None)
    if comm == "kslam":          # the data through include/kslam_comm.h, RCCL's entry points played by tests/fake_rccl (see below)
        env["KSLAM_RCCL_LIB"] = _fake_rccl()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29580 + world + (20 if comm == "kslam" else 0)), os.path.join(ROOT, "bench.py"),
                        "--gpus", str(world), "--total-pairs", "48000", "--species", "4", "--strains", "3",
                        "--genome-len", "300000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([x for x in r.stdout.strip().splitlines() if x.startswith("{")][-1])
    v = line["hot_path"]["verified"]
    assert line["scaling"] == "strong" and line["n_gpus"] == world and line["config"]["pairs_per_batch"] == 48000
    assert line["config"]["pairs_per_gpu"] == 48000 // world
    assert v["ok"] and v["merged_equals_single_context"] and v["merged_unsorted_neighbours"] == 0
    assert v["planted_missing"] == 0 and v["planted_expected"] > 60000 and v["merged_rows"] == v["overlaps"]
    assert line["rccl"]["data_path"].startswith("kslam_comm" if comm == "kslam" else "torch.distributed")
    assert line["rccl"]["launched_by"].startswith("external launcher") and line["verified_classified"]
    _n1_point_is_in_the_line(line)


@pytest.mark.parametrize("world,comm", [(2, "torch"), (4, "torch"), (8, "torch"), (2, "kslam"), (4, "kslam"), (8, "kslam")])
def test_bench_starts_its_own_ranks_when_no_launcher_did(world, comm):
    """`python bench.py --gpus N` with WORLD_SIZE unset: bench.py starts the N rank processes itself (never a re-exec of
    a process that has touched the GPU), relays rank 0's line, and the line proves the ranks were there: n_gpus, the
    communicator's world, ranks_seen, bytes gathered per step, per-rank align times, the merged batch equal to one
    context's, and the second clock through the batch-global tail (SAM text + per-read taxa on rank 0).
    comm = kslam: the data moves through the LIBRARY's communicator (include/kslam_comm.h) -- gather_begin / _end under the
    next batch's alignment, kslam_comm_sharded_tail on two contexts in turn -- with tests/fake_rccl standing in for RCCL,
    which refuses ranks that share a device; comm = torch: through torch.distributed (gloo, host-staged)."""
    env = dict(os.environ, KSLAM_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "KSLAM_RCCL_LIB"):
        env.pop(k, None)
    if comm == "kslam":
        env["KSLAM_RCCL_LIB"] = _fake_rccl()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--total-pairs", "48000",
                        "--species", "4", "--strains", "3", "--genome-len", "300000", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--comm", "auto" if comm == "torch" else "kslam"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    v = line["hot_path"]["verified"]
    assert line["scaling"] == "strong" and line["n_gpus"] == world and line["config"]["pairs_per_gpu"] == 48000 // world
    assert line["rccl"]["world"] == world and line["rccl"]["ranks_seen"] == world and line["rccl"]["launched_by"] == "bench.py"
    assert line["rccl"]["bytes_gathered_per_step"] > 48 * v["overlaps"] and len(line["per_rank_align_ms"]) == world
    if comm == "kslam":
        f = line["rccl"]["kslam_comm"]
        assert line["rccl"]["data_path"].startswith("kslam_comm") and "fake_rccl" in f["library"]
        assert f["ncclCommCount_per_rank"] == [world] * world and f["ncclCommUserRank_per_rank"] == list(range(world))
    else:
        assert line["rccl"]["data_path"].startswith("torch.distributed") and line["rccl"]["kslam_comm"] is None and line["rccl"]["data_path_why"]
    assert v["ok"] and v["merged_equals_single_context"] and v["planted_missing"] == 0
    c = line["classified_rank0_tail"]
    assert c["sam_file_bytes"] > 100 * 48000 and c["per_read_lines_per_batch"] > 0.9 * 48000 and c["pseudo_assembly_on"] == "gpu"
    # the tail sharded like the alignment: each rank's SAM / _PerRead part files, in rank order, ARE rank 0's files
    sh = line["classified_sharded"]
    assert line["verified_classified"] and sh["part_files_in_rank_order_equal_rank0_files"] and sh["pseudo_assembly_on"] == "gpu"
    assert sh["bytes_received_from_other_ranks_per_step"] > 4 * 48000 and sh["max_insert_size"] == c["max_insert_size"]
    _n1_point_is_in_the_line(line)
    assert line["n1_same_workload"]["max_insert_size"] == c["max_insert_size"]


def _n1_point_is_in_the_line(line):
    """every --gpus N > 1 line carries the N = 1 point of ITS workload, measured in that run by rank 0 alone, and the ratio"""
    n1 = line["n1_same_workload"]
    assert "error" not in n1, n1
    assert n1["value"] > 0 and n1["steps"] == line["steps"] and n1["hot_path_reads_per_s"] > 0 and n1["pseudo_assembly_on"] == "gpu"
    assert abs(line["speedup_vs_n1_same_workload"] - line["value"] / n1["value"]) < 2e-3
    assert abs(line["hot_path"]["speedup_vs_n1_same_workload"] - line["hot_path"]["reads_per_s"] / n1["hot_path_reads_per_s"]) < 2e-3
    assert line["scaling_curve_origin"].startswith("n1_same_workload.value")
    # the same batch, the same files: what the one GPU wrote is what the ranks wrote together
    c = line["classified_rank0_tail"]
    assert n1["sam_file_bytes"] == c["sam_file_bytes"] and n1["per_read_file_bytes"] == c["per_read_file_bytes"]


def test_strong_line_of_one_rank_through_the_self_launcher_equals_plain_strong():
    """N = 1: `--gpus 1 --strong` (no process group), the same through the library's communicator (real RCCL, opened by the
    library, at world size 1: --comm auto picks it) and the same through torch.distributed classify the same batch: the same
    SAM file and _PerRead file byte for byte (checksums), the same merged rows."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--strong", "--total-pairs", "40000", "--species", "4", "--strains", "3",
            "--genome-len", "300000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "KSLAM_RCCL_LIB"):
        env.pop(k, None)
    a = subprocess.run(base, env=env, capture_output=True, text=True, timeout=900)
    assert a.returncode == 0, a.stdout[-2000:] + a.stderr[-3000:]
    env2 = dict(env, KSLAM_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29591", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    b = subprocess.run(base, env=env2, capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-3000:]
    c = subprocess.run(base + ["--comm", "torch"], env=dict(env2, MASTER_PORT="29592"), capture_output=True, text=True, timeout=900)
    assert c.returncode == 0, c.stdout[-2000:] + c.stderr[-3000:]
    la, lb, lc = (json.loads([x for x in p.stdout.strip().splitlines() if x.startswith("{")][-1]) for p in (a, b, c))
    for k in ("sam_file_bytes", "per_read_file_bytes", "alignment_pairs_per_batch", "max_insert_size", "sam_file_crc32", "per_read_file_crc32"):
        assert la["classified_rank0_tail"][k] == lb["classified_rank0_tail"][k] == lc["classified_rank0_tail"][k], k
    assert la["verified_classified"] and lb["verified_classified"] and lc["verified_classified"]
    assert la["hot_path"]["verified"]["merged_rows"] == lb["hot_path"]["verified"]["merged_rows"] == lc["hot_path"]["verified"]["merged_rows"]
    assert lb["hot_path"]["verified"]["merged_equals_single_context"] and lc["hot_path"]["verified"]["merged_equals_single_context"]
    assert "rccl" in lb and "rccl" in lc and "rccl" not in la
    fb = lb["rccl"]["kslam_comm"]
    assert lb["rccl"]["data_path"].startswith("kslam_comm") and fb["ncclCommCount_per_rank"] == [1] and "rccl" in fb["library"] and fb["ncclGetVersion"] > 0
    assert lc["rccl"]["data_path"].startswith("torch.distributed") and lc["rccl"]["kslam_comm"] is None


@pytest.mark.parametrize("bounds,pseudo", [([(0, 700), (700, 701), (701, 2500)], "routed"), ([(0, 1250), (1250, 2500)], "routed"),
                                           ([(0, 313), (313, 625), (625, 938), (938, 1250), (1250, 1563), (1563, 1875), (1875, 2188), (2188, 2500)], "routed"),
                                           ([(0, 2500)], "routed"),
                                           ([(0, 1000), (1000, 1006), (1006, 2500)], "routed-empty-shard"),
                                           ([(0, 700), (700, 701), (701, 2500)], True), ([(0, 1250), (1250, 2500)], True),
                                           ([(0, 900), (900, 1800), (1800, 2500)], False), ([(0, 2500)], True)])
def test_sharded_tail_equals_one_context(kslam, synth, bounds, pseudo):
    """The tail with the read pairs SHARDED (kslam_pair_phase_a / _b, then kslam_pseudo_route / _owned / _return -- "routed":
    the entries partitioned over the shards -- or kslam_pseudo_merged: every shard runs the stage on all records): every shard
    pairs and screens its own read pairs; the insert-size limit is computed from all shards' insert sizes, pseudo-assembly
    from all shards' alignment-pair records (what bench.py --gpus N moves over RCCL; here the shards are sibling contexts on
    one GPU and the exchanges are concatenations).  The shards' SAM text, concatenated in shard order, must be the text one context produces
    for the whole batch -- pairing, limit, screens, chain scores, per-row NM / MD / log-probability and all."""
    import torch
    T = importlib.import_module("kslam_amd.tail")
    kd = importlib.import_module("kslam_amd.dist")
    n_pairs = 2500
    rng = np.random.default_rng(99)
    rb, gb = _data(synth, 720, n_pairs)
    if pseudo == "routed-empty-shard":       # the middle shard's reads come from nowhere: no rows, no alignment pairs, nothing to route
        for i in list(range(1000, 1006)) + list(range(n_pairs + 1000, n_pairs + 1006)):
            rb[i] = bytes(synth.random_bases(rng, len(rb[i])))
        pseudo = "routed"
    quals = [bytes(rng.integers(35, 74, len(b), dtype=np.uint8)) for b in rb]
    ids = [b"q%05d" % i for i in range(n_pairs)]
    dev = torch.device("cuda", 0)
    I = T.Index(gb, taxonomy_ids=list(range(1, len(gb) + 1)))
    P_write = T.TailParams.default(pseudo_assembly=False)

    def sam_of(c, reads, q, names):
        c.row_details(of_pairs=True)
        ov, cg, rel1 = c.take_results()
        det, md, rel2 = c.take_row_details(len(ov), copy=False)
        rp, pr, rel3 = c.take_pairs(copy=False)
        out = []
        T.tail_finish_rows(P_write, T.Reads(reads, q, names), I, ov, cg, det, md, rp, pr, sink=out.append)
        for r in (rel1, rel2, rel3):
            r()
        return b"".join(out), len(rp), len(pr)

    # ---- one context, the whole batch ----
    whole = kslam.Context()
    whole.set_index(gb)
    whole.load_reads(rb)
    whole.load_qualities(quals)
    whole.align_resident()
    st = whole.pair_screen(paired=True, stages=7 if pseudo else 3)
    assert not pseudo or st["stages_done"] & 4
    exp, n_rp, n_pr = sam_of(whole, rb, quals, ids + ids)

    # ---- the same batch in shards ----
    shards = []
    for lo, hi in bounds:
        c = whole.sibling()
        loc = kd.local_reads(rb, n_pairs, lo, hi)
        c.load_reads(loc)
        c.load_qualities(kd.local_reads(quals, n_pairs, lo, hi))
        c.align_resident()
        shards.append((c, loc, kd.local_reads(quals, n_pairs, lo, hi), ids[lo:hi] + ids[lo:hi]))
    ins = [kd.device_bytes(*(lambda p, n: (p, n * 4))(*c.pair_phase_a(True, 0)), dev) for c, _, _, _ in shards]
    all_ins = torch.cat(ins)
    torch.cuda.synchronize()
    recs, limits = [], []
    for c, _, _, _ in shards:
        stats, d_pairs, n = c.pair_phase_b(all_ins.data_ptr() if all_ins.numel() else None, all_ins.numel() // 4, 0.95, 3)
        recs.append(kd.device_bytes(d_pairs, n * 32, dev))
        limits.append(stats["max_insert_size"])
    assert len(set(limits)) == 1 and limits[0] == st["max_insert_size"]
    if pseudo == "routed":
        # the entries partitioned over the shards (kslam_pseudo_route / _owned / _return): entry e is shard e mod N's; the
        # "all-to-all" lays the pieces a shard receives end to end in source order, the scores go back the same way
        torch.cuda.synchronize()
        N = len(shards)
        routed = []
        for c, _, _, _ in shards:
            d_heads, counts = c.pseudo_route(N)
            routed.append((kd.device_bytes(d_heads, sum(counts) * 16, dev), counts))
        assert [sum(cn) for _, cn in routed] == [r.numel() // 32 for r in recs]
        back = [[None] * N for _ in range(N)]                         # back[source][destination] = its scores
        for d, (c, _, _, _) in enumerate(shards):
            pieces = []
            for src, (heads, counts) in enumerate(routed):
                at = sum(counts[:d]) * 16
                pieces.append(heads[at:at + counts[d] * 16])
            got = torch.cat(pieces).contiguous()
            torch.cuda.synchronize()
            n_recv = got.numel() // 16
            scores = kd.device_bytes(c.pseudo_owned(got.data_ptr() if n_recv else None, n_recv), n_recv * 4, dev)
            at = 0
            for src, (_, counts) in enumerate(routed):
                back[src][d] = scores[at:at + counts[d] * 4]
                at += counts[d] * 4
        for src, (c, _, _, _) in enumerate(shards):
            mine = torch.cat(back[src]).contiguous()
            torch.cuda.synchronize()
            s2 = c.pseudo_return(mine.data_ptr() if mine.numel() else None, mine.numel() // 4, 0.95)
            assert s2["stages_done"] & 4
    elif pseudo:
        torch.cuda.synchronize()
        base = 0
        for (c, _, _, _), mine in zip(shards, recs):
            all_recs = torch.cat(recs)          # (a fresh copy per shard: the stage rewrites the gathered scores in place)
            torch.cuda.synchronize()
            s2 = c.pseudo_merged(all_recs.data_ptr() if all_recs.numel() else None, all_recs.numel() // 32, base, 0.95)
            assert s2["stages_done"] & 4
            base += mine.numel() // 32
    got, g_rp, g_pr = b"", 0, 0
    for c, loc, q, names in shards:
        text, a, b = sam_of(c, loc, q, names)
        got += text
        g_rp += a
        g_pr += b
        c.close()
    whole.close()
    assert (g_rp, g_pr) == (n_rp, n_pr) and len(exp) > 200 * n_pairs
    assert got == exp


def test_default_bench_line_keeps_the_contract(tmp_path):
    """`python bench.py` (N = 1, the driver's command) at a small size: ONE JSON line on stdout and nothing else, the
    contract's fields, `roofline` + `cpu_baseline` + `hot_path.verified`, and the end-to-end legs behind `value` (the
    native batch loop: files written, repetitions identical, the sink probe)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--pairs", "30000", "--species", "4", "--strains", "3",
            "--genome-len", "300000", "--steps", "3", "--warmup", "1", "--cpu-pairs", "3000", "--cpu-genomes", "4"]
    r = subprocess.run(base, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 1 and out[0].startswith("{"), r.stdout[:500]
    line = json.loads(out[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["scaling"] == "weak" and line["higher_is_better"] is True
    assert line["vs_baseline"] is None and "workload" in line["config"] and line["value"] > 0
    rf, cb = line["roofline"], line["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    have_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libslam_ref.so"))
    assert cb["kind"] == ("reference" if have_ref else "port") and cb["cores"] >= 1 and cb["gpu_equals_cpu_on_sample"]["identical"]
    if have_ref:      # the reference's own alignToDatabase timed, the port beside it, the GPU's rows compared with the reference's
        assert cb["gpu_equals_reference"]["identical"] and cb["gpu_equals_reference"]["differing_rows_are_revcomp_ties"]
        assert cb["port"]["kind"] == "port" and cb["port"]["equals_reference"]["identical"] and cb["port"]["n_alignments"] == cb["n_alignments"]
        assert cb["seconds"] > 0 and (cb["phases_s"] is None or set(cb["phases_s"]) == {"extract", "genome_kmers", "sort", "join", "sw"})
    hp = line["hot_path"]
    assert hp["verified"]["ok"] and hp["verified"]["run_to_run_identical"] and hp["verified"]["planted_missing"] == 0
    e = line["e2e"]
    assert line["value"] == e["reads_per_s"] and len(e["repetitions_ms_per_step"]) == 3 and e["verified"]["repetitions_identical"]
    assert e["verified"]["sam_file_bytes"] > 100 * 30000 and e["driver"].startswith("kslam_stream_classify")
    assert e["sink"] == line["sink_probe"]["chosen"] and isinstance(e["cpu_s_by_thread"], dict) and e["host_cpus_usable"] >= 1
    # the optional legs are opt-in (--legs): the driver's run is the contract legs and the curve's origin
    for k in ("e2e_sam_to_dev_null", "e2e_with_pseudo_assembly", "abi_path", "e2e_other_sink", "strong_reference"):
        assert k not in line, k
    n1 = line["strong_n1"]
    assert "error" not in n1, n1
    assert n1["value"] > 0 and n1["hot_path_reads_per_s"] > 0 and n1["pseudo_assembly_on"] == "gpu" and n1["sam_file_bytes"] > 0
    assert line["scaling_curve_origin"].startswith("strong_n1.value")
    # ---- the same with every optional leg switched on
    r = subprocess.run(base + ["--legs", "all", "--strong-n1", "off", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 1 and out[0].startswith("{"), r.stdout[:500]
    line = json.loads(out[0])
    assert line["e2e_sam_to_dev_null"]["reads_per_s"] > 0 and line["e2e_with_pseudo_assembly"]["pseudo_assembly_on"] == "gpu"
    assert line["abi_path"]["equals_resident_result"] and "strong_n1" not in line and line["scaling_curve_origin"] is None


@pytest.mark.parametrize("pseudo", [True, False])
def test_rccl_behind_the_c_abi_world_1(kslam, synth, pseudo):
    """include/kslam_comm.h with no PyTorch in the path: ncclGetUniqueId / ncclCommInitRank (librccl opened by the library),
    the gather (counts all-gather, export in batch terms) and the sharded tail (two variable-length all-gathers), at the
    world size one box allows.  The gathered arrays are the context's own result byte for byte; adopted by a second
    context they are its result; the tail equals kslam_pair_screen's.  N > 1 on hardware: the driver's SCALE record."""
    import ctypes
    Cm = importlib.import_module("kslam_amd.comm")
    n_pairs = 1500
    rb, gb = _data(synth, 910, n_pairs)
    c = kslam.Context()
    c.set_index(gb)
    c.load_reads(rb)
    n_out, n_cig = c.align_resident()
    exp, ecig = c.fetch_results(n_out, n_cig)
    comm = Cm.Comm(c, Cm.unique_id(), 0, 1)
    assert (c._L.kslam_comm_rank(comm._h), c._L.kslam_comm_world(comm._h), c._L.kslam_ctx_device(c._h)) == (0, 1, 0)
    d_rows, n_rows, d_pool, n_ops = comm.gather_batch(n_pairs, 0, n_pairs)
    assert (n_rows, n_ops) == (n_out, n_cig) and n_rows > n_pairs
    # a second context adopts the gathered arrays (what rank 0 does with the batch-global result)
    c2 = c.sibling()
    c2.load_reads(rb)
    c2.adopt_results_device(d_rows, n_rows, d_pool, n_ops)
    got, gcig = c2.fetch_results(n_rows, n_ops)
    assert got.tobytes() == exp.tobytes() and gcig.tobytes() == ecig.tobytes()
    # the tail through the communicator == the one-call tail
    want = c2.pair_screen(True, 0, 0.95, 7 if pseudo else 3)
    erp, epr = c2.take_pairs()
    st, moved = comm.sharded_tail(True, 0, 0.95, pseudo)
    grp, gpr = c.take_pairs()
    assert {k: st[k] for k in ("n_read_pairs", "n_pairs", "max_insert_size", "n_insert_sizes")} == \
           {k: want[k] for k in ("n_read_pairs", "n_pairs", "max_insert_size", "n_insert_sizes")}
    assert grp.tobytes() == erp.tobytes() and gpr.tobytes() == epr.tobytes()
    assert moved == 0                              # bytes that arrived from OTHER ranks
    facts = comm.info()
    assert (facts["comm_count"], facts["comm_rank"], facts["device"]) == (1, 0, 0) and "rccl" in facts["library"] and facts["rccl_version"] > 0
    # a second gather reuses the communicator's buffers
    again = comm.gather_batch(n_pairs, 0, n_pairs)
    assert again[1:4:2] == (n_rows, n_ops)
    with pytest.raises(kslam.KslamError):
        Cm.Comm(c, Cm.unique_id(), 3, 2)          # rank outside the world
    comm.close()
    c2.close()
    c.close()


def _fake_rccl():
    """tests/fake_rccl/libfake_rccl.so (a stand-in for the RCCL entry points comm.cpp resolves: files in /dev/shm as the
    transport), built on first use"""
    d = os.path.join(ROOT, "tests", "fake_rccl")
    so, src = os.path.join(d, "libfake_rccl.so"), os.path.join(d, "fake_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    return so


@pytest.mark.parametrize("world,pseudo,mode", [(2, True, ""), (4, True, ""), (8, True, ""), (3, False, ""), (3, True, "empty"), (2, True, "empty")])
def test_comm_behind_the_c_abi_at_world_n_on_one_gpu(world, pseudo, mode):
    """include/kslam_comm.h at world sizes the box has no GPUs for: N threads are the ranks (tests/comm_world_n.py), RCCL's
    entry points are the test double -- the communicator code of the LIBRARY runs as at N GPUs: rank 0's gathered arrays are
    one context's result byte for byte, the ranks' SAM text in rank order is one context's text (insert-size limit from all
    ranks' insert sizes, pseudo-assembly with the entries partitioned over the ranks)."""
    env = dict(os.environ, KSLAM_RCCL_LIB=_fake_rccl())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "comm_world_n.py"), str(world), "2400", str(int(pseudo))] + ([mode] if mode else []),
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert not line["hung"] and not line["errors"], line
    assert "fake_rccl" in line["library"] and line["comm_counts"] == [world] * world and line["comm_ranks"] == list(range(world))
    assert line["gather_identical"] and line["rows"] > 2400 and line["second_gather_rows"] == line["rows"]
    assert not any(line["tail_errors"]) and line["sam_identical"] and line["counts_identical"] and line["limit_identical"]
    assert all(bool(s & 4) == pseudo for s in line["stages_done"]) and line["sam_bytes"] > (100 if mode == "empty" else 200) * 2400
    assert all(m > 0 for m in line["moved"])          # (an empty rank still receives the others' insert sizes)


def test_comm_ranks_fail_together_when_one_declines():
    """A device stage that declines on SOME ranks must make every rank return KSLAM_ERR_UNSUPPORTED for the batch, and none
    may wait in a collective.  KSLAM_PSEUDO_CAP = 40 makes the stage decline wherever an entry holds more
    than 40 alignment pairs; the database has 9 entries and entry e belongs to rank e mod 12, so ranks 9, 10 and 11 own no
    entry, succeed locally, and learn of the failure only through the status exchange."""
    env = dict(os.environ, KSLAM_RCCL_LIB=_fake_rccl(), KSLAM_PSEUDO_CAP="40", KSLAM_FAKE_RCCL_TIMEOUT="60")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "comm_world_n.py"), "12", "2400", "1", "decline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert not line["hung"] and not line["errors"], line
    assert line["gather_identical"]
    assert line["tail_errors"] == [4] * 12, line["tail_errors"]          # KSLAM_ERR_UNSUPPORTED on every rank

