#!/usr/bin/env python3
"""bench.py -- throughput of the k-SLAM alignment hot path on MI355X.

A "step" is one pass of the hot path (alignToDatabase, reference src/SLAM.h:59-79:
read k-mer extraction -> k-mer sort -> join against the resident genome k-mer list ->
overlap sort/dedupe -> Smith-Waterman -> CIGAR) over one batch of synthetic paired reads
that is already resident in HBM.  Workload at N=1 = BASELINE.json configs[1]:
1M 150 bp read pairs vs a ~5 Gb synthetic bacterial database.  With N > 1 every rank
replicates the database, aligns its own 1M pairs (weak scaling) and the per-read results
are gathered to rank 0 over RCCL inside the timed region.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
READ_LEN = 150
W = entry.load_package() and importlib.import_module("kslam_amd.workload")   # generator + ground truth
make_database, make_reads = W.make_database, W.make_reads


def cpu_baseline(K, db, offs, seed, n_genomes, n_pairs, read_len=READ_LEN, device=0):
    """The oracle's alignToDatabase timed on the host cores, on a bounded sample of the
    same workload (reported baseline; the oracle is the checker, never the product).  The same
    sample then goes through the HIP library and the two result sets are compared record by record:
    the baseline leg doubles as a parity check inside the driver-run record."""
    import oracle as O
    n_genomes = min(n_genomes, len(offs) - 1)
    sub = db[:int(offs[n_genomes])].cpu()
    suboffs = offs[:n_genomes + 1]
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)
    reads = make_reads(torch.device("cpu"), gen, sub, suboffs, n_pairs, read_len=read_len).numpy()
    rl = [reads[i].tobytes() for i in range(reads.shape[0])]
    subn = sub.numpy()
    gl = [subn[int(suboffs[i]):int(suboffs[i + 1])].tobytes() for i in range(n_genomes)]
    kind_ssw = "own scalar SSW restatement"
    if O.use_reference_ssw(True):
        kind_ssw = "SSW core = the reference's own ssw.c (SSE2) from oracle/_ref"
    # one OpenMP thread per CPU the job may use (cgroup quota), not per hardware thread of the host
    cores = O.usable_cpus()
    O.set_num_threads(cores)
    al, cg, ph = O.align_to_database(rl, gl)
    dt = float(ph[5])          # seconds inside the C call (excludes the ctypes marshalling)
    O.use_reference_ssw(False)
    out = {
        "value": round(len(rl) / dt, 1), "unit": "reads/s", "cores": cores, "kind": "port",
        "pairs": n_pairs, "read_len": read_len, "db_genomes": n_genomes, "db_bases": int(suboffs[-1]),
        "seconds": round(dt, 2),
        "phases_s": dict(zip(("extract", "genome_kmers", "sort", "join", "sw"), (round(float(x), 2) for x in ph[:5]))),
        "sample": "%d pairs x %d bp vs the first %d database genomes (%.0f Mb) -- NOT the whole 5 Gb database, "
                  "which the CPU path cannot finish inside the bounded 10-30 s; whole reference batch "
                  "path incl. genome k-mer re-extraction and the (reads+genomes) sort, OpenMP on the CPUs the "
                  "job's cgroup quota allows; %s" % (n_pairs, read_len, n_genomes, float(suboffs[-1]) / 1e6, kind_ssw),
        "n_alignments": int(len(al)),
    }
    try:   # the same sample through the HIP library: identical records and CIGARs?
        c = K.Context(device=device)
        c.set_index(gl)
        c.load_reads_arrays(np.ascontiguousarray(reads).reshape(-1), np.arange(len(rl) + 1, dtype=np.uint64) * np.uint64(read_len))
        n_out, n_cig = c.align_resident()
        gov, gcg = c.fetch_results(n_out, n_cig)
        c.close()
        same = len(gov) == len(al) and all((gov[f] == al[f]).all() for f in (
            "read", "entry", "rel", "revcomp", "score", "ref_begin", "ref_end", "query_begin", "query_end",
            "cigar_len", "cigar_off")) and np.array_equal(gcg, cg)
        out["gpu_equals_cpu_on_sample"] = {"identical": bool(same), "alignments": int(len(gov)), "cigar_ops": int(len(gcg))}
    except Exception as e:   # never lose the bench line over the extra check
        out["gpu_equals_cpu_on_sample"] = {"error": repr(e)}
    return out


def abi_path(K, ctx, reads, read_len, steps):
    """What a k-SLAM host linking the library sees: reads[i].bases in host memory in (char **, lengths),
    overlap records + CIGAR pool back in host memory (page-locked, library-owned), batch after batch
    through kslam_align_batch_async / kslam_wait_batch with two batches in flight -- upload, kernels and
    download of neighbouring batches overlap.  Every PCIe byte is inside this number; it is not `value`."""
    import ctypes as C
    host = np.ascontiguousarray(reads.cpu().numpy())
    n = host.shape[0]
    ptrs = (host.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(read_len)).astype(np.uint64)
    lens = np.full(n, read_len, dtype=np.uint32)
    pp, lp = ptrs.ctypes.data, lens.ctypes.data

    t_sub, t_wait = [], []

    def run(k):
        t0 = time.perf_counter()
        rows = 0
        pend = [ctx.submit_batch_pointers(n, pp, lp)]
        for i in range(k):
            ta = time.perf_counter()
            if i + 1 < k:
                pend.append(ctx.submit_batch_pointers(n, pp, lp))
            tb = time.perf_counter()
            ov, cg, release = ctx.wait_batch(pend.pop(0), copy=False)
            rows = len(ov)
            release()
            t_sub.append(tb - ta)
            t_wait.append(time.perf_counter() - tb)
            done_at.append(time.perf_counter())
        return time.perf_counter() - t0, rows
    done_at = []
    run(3)                                             # lanes, page-locked buffers and work buffers exist now
    del t_sub[:], t_wait[:], done_at[:]
    wall, rows = run(steps)
    steady = (done_at[-1] - done_at[0]) / (len(done_at) - 1)   # batch-to-batch, without the pipeline fill of the first
    t0 = time.perf_counter()
    for _ in range(3):
        ov, cg, release = ctx.align_batch_pointers(n, pp, lp, copy=False)
        if _ < 2:
            release()
    sync_wall = (time.perf_counter() - t0) / 3
    # identity with the resident path
    n_out, n_cig = ctx.align_resident()
    r_ov, r_cg = ctx.fetch_results(n_out, n_cig)
    same = ov.tobytes() == r_ov.tobytes() and cg.tobytes() == r_cg.tobytes()
    release()
    return {
        "ms_in_submit": round(1e3 * sum(t_sub) / max(len(t_sub), 1), 2), "ms_in_wait": round(1e3 * sum(t_wait) / max(len(t_wait), 1), 2),
        "reads_per_s": round(n / steady, 1), "ms_per_batch": round(steady * 1e3, 2), "steps": steps,
        "including_pipeline_fill": {"reads_per_s": round(n * steps / wall, 1), "ms_per_batch": round(wall / steps * 1e3, 2)},
        "one_batch_at_a_time": {"reads_per_s": round(n / sync_wall, 1), "ms_per_batch": round(sync_wall * 1e3, 2)},
        "h2d_mb_per_batch": round(n * read_len / 1e6, 1), "d2h_mb_per_batch": round((rows * 48 + n_cig * 4) / 1e6, 1),
        "equals_resident_result": bool(same),
        "what": "host pointers in -> kslam_align_batch_async / kslam_wait_batch (two batches in flight) -> host "
                "results out; `one_batch_at_a_time` = kslam_align_batch in a loop",
    }


def _host_copy(t, keep):
    """a host copy in a private anonymous mapping advised for transparent huge pages (what kslam_db_load /
    kslam_fastq_parse do for their columns)"""
    import mmap
    n = t.numel() * t.element_size()
    m = mmap.mmap(-1, max((n + (2 << 20) - 1) // (2 << 20) * (2 << 20), 2 << 20),
                  flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    if hasattr(mmap, "MADV_HUGEPAGE"):
        m.madvise(mmap.MADV_HUGEPAGE)
    a = np.frombuffer(m, dtype=np.uint8, count=n).reshape(tuple(t.shape))
    torch.from_numpy(a).copy_(t)
    keep.append(m)
    return a


def sam_pipeline(K, ctx, reads, db, offs, read_len, steps, pseudo_assembly=False):
    """Read columns in host memory -> SAM records on the host, the way a streaming caller runs it: the
    batch goes up through the pipelined entry (kslam_submit_batch_columns: bases + qualities by DMA from
    page-locked columns), comes back as overlap records + CIGARs + per-row NM / log-probability / MD
    (kslam_row_details: the GPU walks every alignment's CIGAR + read + quality + entry window, so the host
    writer formats text and never reads the 5 GB database), and goes through the host tail
    (include/kslam_tail.h: pairing, insert-size / score screens, [pseudo-assembly,] SAM text) on a worker
    thread while the next batches are on the GPU.  Reported next to the headline number; not `value`."""
    import threading
    T = importlib.import_module("kslam_amd.tail")
    n_reads = reads.shape[0]
    t0 = time.time()
    nb = n_reads * read_len
    hb, hq = K.HostBuffer(nb + 64), K.HostBuffer(nb + 64)
    torch.from_numpy(hb.a[:nb].reshape(n_reads, read_len)).copy_(reads)
    hq.a[:] = ord("I")                                             # quality: constant phred 40
    off = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
    R = T.ReadsArrays(hb.a[:nb], read_len, quality_u8=hq.a[:nb])
    # the index view the writer gets: offsets, names, taxonomy ids -- and NO copy of the database
    I = T.IndexArrays(np.zeros(1, dtype=np.uint8), offs, taxonomy_ids=np.arange(1, len(offs), dtype=np.uint32))
    P = T.TailParams.default(pseudo_assembly=pseudo_assembly)
    P_write = T.TailParams.default(pseudo_assembly=False)         # when the GPU has run that stage too
    gpu_stages = 7 if pseudo_assembly and os.environ.get("KSLAM_BENCH_HOST_PSEUDO") != "1" else 3
    t_host_copy = time.time() - t0
    stats, finished = [], []

    def submit():
        return ctx.submit_batch_columns(n_reads, hb.ptr, hq.ptr, off.ctypes.data)

    def collect(tk):
        res = ctx.collect_batch(tk)
        return res + (ctx.last_pairs,)

    def tail(ov, cg, det, md, release, pairs):
        rp, pr, pst = pairs                        # read pairs / alignment pairs from the GPU (views: modified in place)
        st = T.tail_finish_rows(P_write if pst["stages_done"] & 4 else P, R, I, ov, cg, det, md, rp, pr)
        release()                                  # page-locked result buffers back to the library
        d = st.as_dict()
        d["gpu_pairing"] = pst
        stats.append(d)
        finished.append(time.perf_counter())       # batch complete: SAM text written

    # score screen, pairing, insert-size statistics, screens [, pseudo-assembly, second screen]: on the GPU
    ctx.set_pairing(paired=True, stages=gpu_stages)
    for tk in [submit(), submit(), submit()]:       # warm both lanes' buffers and the tail's work buffers
        tail(*collect(tk))
    stats.clear()
    del finished[:]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    worker, done_at = None, []
    pend = [submit(), submit()]                    # two batches on the GPU lanes
    for k in range(steps):
        res = collect(pend.pop(0))                 # batch k: its rows in page-locked buffers
        if k + 2 < steps:
            pend.append(submit())
        if worker is not None:
            worker.join()                          # host tail of batch k-1 must be done
        worker = threading.Thread(target=tail, args=res)
        worker.start()
        done_at.append(time.perf_counter())
    worker.join()
    wall = time.perf_counter() - t0
    tail_ms = [sum(v for k, v in s.items() if k.startswith("ms_")) for s in stats]
    last = stats[-1]
    ctx.set_pairing(stages=0)
    hb.close()
    hq.close()
    # batch-to-batch in the steady state (completion of batch 0 to completion of the last): what a long run of
    # batches sees; the wall clock of a short run also holds the pipeline's fill (first batch: upload + align +
    # download with nothing to overlap) and drain (last host stage)
    steady = (finished[-1] - finished[0]) / (len(finished) - 1) if len(finished) > 1 else wall / steps
    return {
        "reads_per_s": round(n_reads / steady, 1), "ms_per_batch": round(steady * 1e3, 2),
        "including_pipeline_fill_and_drain": {"reads_per_s": round(n_reads * steps / wall, 1),
                                              "ms_per_batch": round(wall / steps * 1e3, 2)},
        "steps": steps, "host_tail_ms": round(sum(tail_ms) / len(tail_ms), 2),
        "host_tail_phases_ms": {k[3:]: round(last[k], 2) for k in last if k.startswith("ms_")},
        "host_threads": int(last["threads"]), "sam_mb_per_batch": round(last["sam_bytes"] / 1e6, 1),
        "alignment_pairs": int(last["n_paired_final"]), "read_pairs_aligned": int(last["n_read_pairs"]),
        "pseudo_assembly": bool(pseudo_assembly), "gpu_pairing": last["gpu_pairing"],
        "pseudo_assembly_on": ("gpu" if last["gpu_pairing"]["stages_done"] & 4 else "host") if pseudo_assembly else None,
        "what": "read columns in page-locked host memory -> kslam_submit_batch_columns (align + per-row NM / "
                "log-probability / MD + score screen / pairing / insert-size statistics / screens%s on the GPU, two "
                "batches in flight) -> kslam_collect_batch -> SAM text (host, discarded by the writer; no host copy "
                "of the database) on a worker thread; one-time host copy of the reads took %.1f s" % (
                    " / pseudo-assembly / second screen" if pseudo_assembly else "", t_host_copy),
    }


def full_pipeline(K, ctx, reads, db, offs, read_len, steps, pseudo_assembly=False):
    """First FASTQ byte to last SAM byte, the way the reference's low-memory driver loops
    (src/SLAM.h:193-241): per batch the two FASTQ texts are parsed on the host (include/kslam_fastq.h),
    bases and qualities go to the GPU through the pipelined entry (kslam_submit_batch_columns: by DMA from
    the parser's page-locked columns), come back as
    overlap records + CIGARs + per-row NM / log-probability / MD (kslam_collect_batch) and go through the
    host tail to SAM text -- parse of batch k+1, GPU work of batch k and tail of batch k-1 at the same
    time.  Synthetic FASTQ text of the bench's own read batch, held in memory; the SAM text is handed to
    a writer that discards it.  Reported next to the headline number; it is not `value`."""
    import threading
    F = importlib.import_module("kslam_amd.fastq")
    T = importlib.import_module("kslam_amd.tail")
    host = reads.cpu().numpy()
    n = host.shape[0] // 2

    def fastq_text(block, mate):
        # fixed-width records "@p0000123/1\n<bases>\n+\n<quality>\n"
        W = 2 + 7 + 3 + read_len + 3 + read_len + 1
        a = np.empty((n, W), dtype=np.uint8)
        a[:, 0:2] = np.frombuffer(b"@p", dtype=np.uint8)
        idx = np.arange(n, dtype=np.int64)
        for d in range(7):
            a[:, 2 + d] = ((idx // 10 ** (6 - d)) % 10 + ord("0")).astype(np.uint8)
        a[:, 9:12] = np.frombuffer(b"/%d\n" % mate, dtype=np.uint8)
        a[:, 12:12 + read_len] = block
        a[:, 12 + read_len:15 + read_len] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        a[:, 15 + read_len:15 + 2 * read_len] = ord("I")
        a[:, W - 1] = ord("\n")
        return a.tobytes()
    r1, r2 = fastq_text(host[:n], 1), fastq_text(host[n:], 2)
    # the two "files" as a host would hold them for this library: read into page-locked buffers
    h1, h2 = K.HostBuffer(len(r1) + 64), K.HostBuffer(len(r2) + 64)
    h1.a[:len(r1)] = np.frombuffer(r1, dtype=np.uint8)
    h2.a[:len(r2)] = np.frombuffer(r2, dtype=np.uint8)
    len1, len2 = len(r1), len(r2)
    del r1, r2
    I = T.IndexArrays(np.zeros(1, dtype=np.uint8), offs, taxonomy_ids=np.arange(1, len(offs), dtype=np.uint32))
    nthr = int(os.environ.get("KSLAM_BENCH_HOST_THREADS", "0"))
    P = T.TailParams.default(threads=nthr, pseudo_assembly=pseudo_assembly)
    P_write = T.TailParams.default(threads=nthr, pseudo_assembly=False)   # when the GPU has run that stage too
    gpu_stages = 7 if pseudo_assembly and os.environ.get("KSLAM_BENCH_HOST_PSEUDO") != "1" else 3
    stats, on_gpu, finished = [], [], []

    def collect(tk):
        res = ctx.collect_batch(tk)
        return res + (ctx.last_pairs, ctx.last_reads)

    def tail(batch, ov, cg, det, md, release, pairs, reads_view):
        rp, pr, pst = pairs
        on_gpu.append(bool(pst["stages_done"] & 4))
        st = T.tail_finish_rows(P_write if on_gpu[-1] else P, reads_view if reads_view is not None else batch, I, ov, cg,
                                det, md, rp, pr)
        release()
        if batch is not None:
            batch.close()
        stats.append(st.as_dict())
        finished.append(time.perf_counter())

    ctx.set_pairing(paired=True, stages=gpu_stages)
    host_index = os.environ.get("KSLAM_BENCH_HOST_FASTQ_INDEX") == "1"   # A/B: the record index on the host (round-2 first form)

    def parse_and_submit():
        t0 = time.perf_counter()
        # the host only INDEXES the records (line ends, identifiers, offsets); the texts go up as they
        # are and the bases / quality columns are cut out of them on the GPU
        if not host_index:
            # nothing is scanned on the host: line index, identifiers, offsets and columns are all made on the GPU
            tk = ctx.submit_batch_fastq_text(h1.ptr, len1, h2.ptr, len2)
            return None, tk, (0.0, time.perf_counter() - t0)
        batch, u1, u2 = F.index_pair(h1.ptr, len1, h2.ptr, len2, threads=nthr)
        t1 = time.perf_counter()
        tk = ctx.submit_batch_fastq(h1.ptr, len1, h2.ptr, len2, batch.n_reads, batch._cols.bases_off,
                                    batch.layout.bases_at, batch.layout.quality_at)
        return batch, tk, (t1 - t0, time.perf_counter() - t1)
    import ctypes as C
    worker, parts, waits, joins = None, [], [], []

    depth = int(os.environ.get("KSLAM_LANES", "2")) + 1

    def run(nsteps):
        nonlocal worker
        queue = [parse_and_submit()]                               # batch 0 on its way
        for k in range(nsteps):
            while len(queue) < depth and k + len(queue) < nsteps:  # more batches queued behind it: one per lane + 1
                queue.append(parse_and_submit())
            cur = queue.pop(0)
            tw = time.perf_counter()
            res = collect(cur[1])                                  # batch k back from the GPU
            waits.append(time.perf_counter() - tw)
            tj = time.perf_counter()
            if worker is not None:
                worker.join()                                      # host stage of batch k-1 done
            joins.append(time.perf_counter() - tj)
            worker = threading.Thread(target=tail, args=(cur[0],) + tuple(res))
            worker.start()
            parts.append(cur[2])
        worker.join()
        worker = None
    run(6)          # warm-up in the same shape: both lanes, the parser's page-locked block cache, the tail's arenas
    stats.clear()
    del parts[:], waits[:], joins[:], finished[:]
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    run(steps)
    wall = time.perf_counter() - t_start
    n_reads = 2 * n
    ms = lambda k: round(1e3 * sum(p[k] for p in parts) / len(parts), 2)   # noqa: E731
    ctx.set_pairing(stages=0)
    h1.close()
    h2.close()
    steady = (finished[-1] - finished[0]) / (len(finished) - 1) if len(finished) > 1 else wall / steps   # see sam_pipeline
    return {
        "reads_per_s": round(n_reads / steady, 1), "ms_per_batch": round(steady * 1e3, 2), "steps": steps,
        "including_pipeline_fill_and_drain": {"reads_per_s": round(n_reads * steps / wall, 1),
                                              "ms_per_batch": round(wall / steps * 1e3, 2)},
        "ms_fastq_parse": ms(0), "ms_submit": ms(1), "ms_waiting_for_gpu": round(1e3 * sum(waits) / len(waits), 2),
        "ms_waiting_for_host_stage": round(1e3 * sum(joins) / len(joins), 2),
        "host_tail_ms": round(sum(sum(v for k, v in s.items() if k.startswith("ms_")) for s in stats) / len(stats), 2),
        "host_tail_phases_ms": {k[3:]: round(stats[-1][k], 2) for k in stats[-1] if k.startswith("ms_")},
        "fastq_mb_per_batch": round((len1 + len2) / 1e6, 1), "sam_mb_per_batch": round(stats[-1]["sam_bytes"] / 1e6, 1),
        "pseudo_assembly": bool(pseudo_assembly),
        "pseudo_assembly_on": ("gpu" if all(on_gpu) else "host") if pseudo_assembly else None,
        "fastq_index": "host" if host_index else "gpu",
        "what": "FASTQ text (2 files, in page-locked memory) -> kslam_submit_batch_fastq_text (texts up by DMA; line index, "
                "identifiers, offsets and the bases / quality columns made on the GPU; align, per-row "
                "NM / log-probability / MD, score screen / pairing / insert-size statistics / screens [/ pseudo-assembly / "
                "second screen]) -> kslam_collect_batch -> SAM text (host, discarded by the "
                "writer; no host copy of the database); three batches in flight, the host stage of batch k-1 under the GPU "
                "work of batches k, k+1",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=1_000_000, help="read pairs per GPU per step")
    ap.add_argument("--species", type=int, default=250)
    ap.add_argument("--strains", type=int, default=5)
    ap.add_argument("--genome-len", type=int, default=4_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=300000)
    ap.add_argument("--cpu-genomes", type=int, default=25)
    ap.add_argument("--no-cigar", action="store_true")
    ap.add_argument("--no-abi-path", action="store_true", help="skip the host-pointers-in / host-results-out leg")
    ap.add_argument("--no-sam-pipeline", action="store_true", help="skip the GPU + host-tail pipeline leg")
    ap.add_argument("--no-full-pipeline", action="store_true", help="skip the FASTQ text -> SAM text leg")
    ap.add_argument("--read-len", type=int, default=READ_LEN, help="150 (BASELINE configs[1..3]) or 250 (configs[4])")
    ap.add_argument("--strong", action="store_true",
                    help="BASELINE configs[3] shape: ONE batch of --total-pairs pairs per step, split over the GPUs, "
                         "timed until rank 0 holds the merged result (default when --gpus > 1)")
    ap.add_argument("--weak", action="store_true", help="with --gpus > 1: --pairs fresh pairs per GPU instead")
    ap.add_argument("--total-pairs", type=int, default=10_000_000,
                    help="pairs per batch in --strong mode (the reference's --num-reads-at-once default, src/main.cpp:56)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    # KSLAM_BENCH_SHARE_GPU=1 (tests only): the ranks share the GPUs that exist, and talk through gloo with
    # host-staged pieces -- RCCL refuses two ranks on one device.  Everything else (sharding, count exchange,
    # export in batch terms, placement, verification) is the code a real N-GPU run executes.
    share = os.environ.get("KSLAM_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if share else dev          # where the communicator's small tensors live
    dist = None
    use_dist = world > 1 or os.environ.get("KSLAM_BENCH_FORCE_DIST") == "1"   # the override runs the RCCL code path at N=1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        limit = datetime.timedelta(minutes=10)     # a rank that has died must not leave the others waiting for half an hour
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=limit)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=limit)

    K = entry.load_package()
    kdist = importlib.import_module("kslam_amd.dist")
    strong = args.strong or (world > 1 and not args.weak)
    PIECES = 8          # the strong batch is generated in 8 fixed pieces, so it is the same batch for N = 1, 2, 4, 8
    if strong and (8 % world or args.total_pairs % PIECES):
        raise SystemExit("--strong needs 1, 2, 4 or 8 ranks and --total-pairs divisible by 8")

    # ---- synthetic inputs, generated straight into HBM (data: synthetic) ----
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)                      # database: same on every rank (replicated index)
    t0 = time.time()
    db, offs = make_database(dev, gen, args.species, args.strains, args.genome_len)
    if strong:
        # this rank's pairs [pair_lo, pair_hi) of the one batch, local layout [R1 of them | R2 of them]
        piece = args.total_pairs // PIECES
        mine = range(rank * PIECES // world, (rank + 1) * PIECES // world)
        pair_lo, pair_hi = mine[0] * piece, (mine[-1] + 1) * piece
        r1s, r2s, tr = [], [], []
        for pc in mine:
            gen.manual_seed(2 + 1000 * pc)
            r, t = make_reads(dev, gen, db, offs, piece, read_len=args.read_len, with_truth=True)
            r1s.append(r[:piece]); r2s.append(r[piece:]); tr.append(t)
        reads = torch.cat(r1s + r2s, 0).contiguous()
        truth = {k: torch.cat([t[k][:piece] for t in tr] + [t[k][piece:] for t in tr]) for k in tr[0]}
        del r1s, r2s, tr
        n_batch_pairs = args.total_pairs
    else:
        gen.manual_seed(2 + 1000 * rank)        # reads: a different shard of pairs per rank
        reads, truth = make_reads(dev, gen, db, offs, args.pairs, read_len=args.read_len, with_truth=True)
        pair_lo, pair_hi = rank * args.pairs, (rank + 1) * args.pairs
        n_batch_pairs = args.pairs * world
    torch.cuda.synchronize()
    t_gen = time.time() - t0

    ctx = K.Context(report_cigar=not args.no_cigar, device=local_rank)
    t0 = time.time()
    ctx.set_index_device(len(offs) - 1, db.data_ptr(), offs)
    t_index = time.time() - t0
    n_reads = reads.shape[0]
    roffs = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(args.read_len))
    ctx.load_reads_device(n_reads, reads.data_ptr(), roffs)

    pending = []   # the gather of the previous batch, still in flight while this one is aligned
    merged = {}    # rank 0, --strong: the batch-global result of the last finished batch (device tensors)

    def drain():
        while pending:
            got = kdist.finish_gather(pending.pop())
            if strong and rank == 0:
                # every transfer landed in its final place (kslam_amd.dist.start_gather_sharded): rank 0
                # now HOLDS the batch-global result in the reference's order, and ran no kernel for it
                merged["ov"], merged["cg"] = got

    split = {"align": 0.0, "wait_for_previous_gather": 0.0, "counts_export_post": 0.0}   # host clock, this rank, timed steps

    def step():
        ta = time.perf_counter()
        n_out, n_cig = ctx.align_resident()
        tb = time.perf_counter()
        split["align"] += tb - ta
        if use_dist and strong:
            # the one exchange of the path (point-to-point over xGMI): count exchange, every rank re-bases
            # its own records on its own GPU, four sends per rank into their final places on rank 0.  The
            # transfer of batch k overlaps the alignment of batch k + 1; it is waited for before the next
            # one starts and before the clock stops.
            drain()
            tc = time.perf_counter()
            pending.append(kdist.start_gather_sharded(ctx, n_reads // 2, pair_lo, args.total_pairs, dev))
            split["wait_for_previous_gather"] += tc - tb
            split["counts_export_post"] += time.perf_counter() - tc
        elif use_dist:
            ov = torch.empty(n_out * 48, dtype=torch.uint8, device=dev)
            cg = torch.empty(n_cig * 4, dtype=torch.uint8, device=dev)
            ctx.copy_results_device(ov.data_ptr(), cg.data_ptr())
            drain()
            pending.append(kdist.start_gather(ov, cg))
        return n_out, n_cig

    def barrier():
        drain()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    acc = {}
    for k in split:
        split[k] = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_out, n_cig = step()
        for k, v in ctx.timings().items():
            acc[k] = acc.get(k, 0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        te = torch.tensor([elapsed] + [split[k] for k in sorted(split)], dtype=torch.float64, device=cdev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te[0].item())
        split_max = {k: float(v) for k, v in zip(sorted(split), te[1:].tolist())}     # slowest rank per part

    # ---- outside the timed region: is what was just timed RIGHT?  (no oracle here: the generator's
    # own ground truth, the reference's structural expectations of src/Tests.h:161-264, :321-330) ----
    def device_results():
        n_out, n_cig = ctx.align_resident()
        ov = torch.empty(n_out * 48, dtype=torch.uint8, device=dev)
        cg = torch.empty(max(n_cig, 1) * 4, dtype=torch.uint8, device=dev)
        ctx.copy_results_device(ov.data_ptr(), cg.data_ptr())
        return ov, cg[:n_cig * 4].view(torch.int32)
    ov_a, cg_a = device_results()
    verified = W.check_against_truth(ov_a, None if args.no_cigar else cg_a, truth, args.read_len)
    ov_b, cg_b = device_results()
    verified["run_to_run_identical"] = bool(ov_a.numel() == ov_b.numel() and torch.equal(ov_a, ov_b)
                                            and torch.equal(cg_a, cg_b))
    verified["ok"] = bool(verified["ok"] and verified["run_to_run_identical"])
    if use_dist:   # every rank checked its own shard: sum the counts, AND the verdicts
        keys = [k for k, v in verified.items() if not isinstance(v, bool)]
        t = torch.tensor([verified[k] for k in keys] + [int(verified["ok"]), int(verified["run_to_run_identical"])],
                         dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        for k, v in zip(keys, t[:len(keys)].tolist()):
            verified[k] = int(v)
        verified["ok"] = bool(int(t[-2]) == world)
        verified["run_to_run_identical"] = bool(int(t[-1]) == world)
    if strong and use_dist and rank == 0 and "ov" in merged:
        # the merged batch: row count, order, and -- when this rank aligned the whole batch itself (one
        # rank) -- byte identity with the single-context result
        mc = W.overlap_columns(merged["ov"])
        n_rel = int(mc["rel"].max()) + 1026 if mc["rel"].numel() else 1026
        mkey = (mc["read"] * (len(offs) - 1) + mc["entry"]) * n_rel + (mc["rel"] + 1024)
        verified["merged_rows"] = int(mkey.numel())
        verified["merged_unsorted_neighbours"] = int((mkey[1:] < mkey[:-1]).sum()) if mkey.numel() > 1 else 0
        if world > 1:
            # the whole batch once more, in THIS rank's context alone (outside the timed region): what the N ranks
            # produced together must be, byte for byte, what one context returns for the batch
            del ov_a, cg_a, ov_b, cg_b
            piece = args.total_pairs // PIECES
            r1s, r2s = [], []
            for pc in range(PIECES):
                gen.manual_seed(2 + 1000 * pc)
                r = make_reads(dev, gen, db, offs, piece, read_len=args.read_len)
                r1s.append(r[:piece]); r2s.append(r[piece:])
            whole = torch.cat(r1s + r2s, 0).contiguous()
            del r1s, r2s
            ctx.load_reads_device(whole.shape[0], whole.data_ptr(),
                                  np.arange(whole.shape[0] + 1, dtype=np.uint64) * np.uint64(args.read_len))
            ov_a, cg_a = device_results()
            ov_b = cg_b = None
            del whole
        verified["merged_equals_single_context"] = bool(
            merged["ov"].numel() == ov_a.numel() and torch.equal(merged["ov"], ov_a) and
            torch.equal(merged["cg"].view(torch.int32), cg_a))
        verified["ok"] = bool(verified["ok"] and verified["merged_equals_single_context"])
        verified["ok"] = bool(verified["ok"] and verified["merged_unsorted_neighbours"] == 0)
    del ov_a, cg_a, ov_b, cg_b

    # RCCL announces itself on stdout through C stdio ("Librccl path : ..."), buffered when piped and
    # otherwise flushed when each rank exits -- after rank 0's JSON.  Every rank pushes it out now,
    # and rank 0 prints only once all have, so that the JSON line is the last line of the output.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if use_dist:
        dist.barrier()

    if rank == 0:
        S = args.steps
        tm = {k: v / S for k, v in acc.items()}
        total_reads = 2 * n_batch_pairs * S
        n_kmers = tm["n_read_kmers"]
        passes = int(round(tm["sort_passes"]))
        launches = max(tm["n_scatter_launches"], 1)
        launch_ms = tm["ms_sort_scatter"] / launches
        n_sorted = tm["n_kmers_kept"]      # the records the per-batch sort moves: read k-mers the genome filter let through
        per_launch_bytes = (n_sorted / max(tm["n_chunks"], 1)) * 16 * 2   # one scatter launch: 16 B in + 16 B out per record
        achieved = per_launch_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("k_scatter_bytes_per_launch")
            except Exception:
                traffic = None
        sort_bytes = n_sorted * 16 * (2 * passes + 1)
        # the SW phase against the VALU issue rate: instruction count of the phase's kernels per alignment call from the
        # committed counter run of this same workload (tools/pmc_valu.sh -> profiles/sw_valu.json), time measured live
        sw_valu = None
        vpath = os.path.join(ROOT, "profiles", "sw_valu.json")
        if os.path.exists(vpath) and args.read_len == READ_LEN and not strong and args.pairs == 1_000_000 and tm["ms_sw"] > 0:
            try:
                vj = json.load(open(vpath))
                instr = float(vj["sw_phase_per_align"]["valu_wave_instr"])
                rate = instr / (tm["ms_sw"] * 1e-3) / 1e9
                sw_valu = {"bound": "valu-issue", "kernels": "k_sw_plan + k_sw_band<...> tiers + k_sw (the SW phase, 61 % of the step)",
                           "achieved": round(rate, 1), "peak": 1228.8, "unit": "G wave-instr/s", "frac": round(rate / 1228.8, 4),
                           "sustained_for_4_cycle_kinds": 575.0, "sustained_for_2_cycle_kinds": 1084.0,
                           "valu_wave_instr_per_step": int(instr), "ms": round(tm["ms_sw"], 3),
                           "note": "peak = 256 CUs x 4 SIMD x 2.4 GHz / 2 cycles; the sweeps' instruction mix is mostly 4-cycle kinds "
                                   "(v_max_i32, VOP3, DPP, v_max_f64: tools/valu_peak.hip, profiles/r01g_valu_peak.txt), whose sustained "
                                   "rate is the realistic ceiling; instruction count from profiles/sw_valu.json (PMC, separate run)"}
            except Exception:
                sw_valu = None
        out = {
            "metric": "paired %dbp reads/sec classified (bit-exact SAM)" % args.read_len,
            "value": round(total_reads / elapsed, 1),
            "unit": "reads/s",
            "n_gpus": world, "steps": S, "warmup": args.warmup,
            "ms_per_step": round(elapsed / S * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u64 k-mers / i32 DP", "data": "synthetic",
            "config": {
                "workload": ("BASELINE configs[3]: ONE batch of %d x 2 x %d bp reads per step, read pairs split over %d GPU(s), "
                             "timed until rank 0 holds the merged result, vs %d-genome " % (
                                 args.total_pairs, args.read_len, world, len(offs) - 1) if strong else
                             "BASELINE configs[%d]: %d x 2 x %d bp reads per GPU vs %d-genome " % (
                                 4 if args.read_len > 150 else 1, args.pairs, args.read_len, len(offs) - 1)) +
                            "(%d species x %d strains x %.1f Mb = %.2f Gb) synthetic bacterial db, "
                            "hot path alignToDatabase incl. CIGAR, inputs resident in HBM" % (
                                args.species, args.strains, args.genome_len / 1e6, float(offs[-1]) / 1e9),
                "pairs_per_batch": n_batch_pairs,
                "pairs_per_gpu": n_reads // 2, "db_bases": int(offs[-1]),
                "parallelism": "read pairs sharded x%d, genome k-mer list replicated, gather to rank 0" % world,
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_scatter<4> (the scatter launch of one radix pass of the read k-mer sort; "
                                          "since round 2 the sort only sees the k-mers the genome filter lets through)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "launch_ms": round(launch_ms, 4), "bytes_per_launch": int(per_launch_bytes),
                "sort_phase": {"passes": passes, "bytes": int(sort_bytes), "ms": round(tm["ms_sort"], 3),
                               "frac": round(sort_bytes / (tm["ms_sort"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                               if tm["ms_sort"] > 0 else 0.0},
                # context, not the roofline: what the best hand-written streaming copy of the same bytes
                # sustained on a bench box (tools/copy_peak.hip, profiles/r01h_copy_peak.txt)
                "streaming_copy_ceiling": {"GB/s": 5590.0, "measured": "profiles/r01h_copy_peak.txt"},
            },
            "roofline_valu": sw_valu,
            "phases_ms": {k: round(tm[k], 3) for k in ("ms_extract", "ms_sort", "ms_join", "ms_sw",
                                                         "ms_cigar", "ms_total")},
            "counts": {"read_kmers": int(n_kmers), "read_kmers_kept_by_filter": int(tm["n_kmers_kept"]),
                       "genome_kmers": int(tm["n_genome_kmers"]),
                       "overlaps_raw": int(tm["n_overlaps_raw"]), "candidates": int(tm["n_overlaps"]),
                       "cigar_ops": int(n_cig), "chunks": int(tm["n_chunks"])},
            "sw_gcups": round(tm["sw_cells"] / ((tm["ms_sw"]) * 1e-3) / 1e9, 1) if tm["ms_sw"] > 0 else 0.0,
            "setup_s": {"generate": round(t_gen, 2), "index_build": round(t_index, 2)},
            "verified": verified,
        }
        if use_dist and strong:
            # where a step's time goes on the host clock (per step; max over ranks, and rank 0 = the collecting rank):
            # the align call, the wait for the previous batch's gather to land, and count exchange + export + posting
            out["strong_step_split_ms"] = {
                "max_over_ranks": {k: round(v / S * 1e3, 3) for k, v in split_max.items()},
                "rank0": {k: round(v / S * 1e3, 3) for k, v in split.items()}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(K, db, offs, 77, args.cpu_genomes, args.cpu_pairs, args.read_len, local_rank)
        torch.cuda.empty_cache()       # what torch's allocator cached while generating the inputs goes back to the device:
                                       # the legs below run two more contexts' worth of library buffers
        if world == 1 and not strong and not args.no_abi_path:
            try:
                out["abi_path"] = abi_path(K, ctx, reads, args.read_len, max(args.steps, 12))
            except Exception as e:   # extra evidence only: never lose the bench line over it
                out["abi_path"] = {"error": repr(e)}
        # The pipeline legs run the host tail with the flags of the configuration they are on: BASELINE
        # configs[1] is quoted with --no-pseudo-assembly; the same legs with pseudo-assembly (the reference's
        # default, configs[2]) are reported next to them.
        if world == 1 and not strong and not args.no_sam_pipeline and not args.no_cigar:
            for key, pa in (("sam_pipeline", False), ("sam_pipeline_with_pseudo_assembly", True)):
                try:
                    out[key] = sam_pipeline(K, ctx, reads, db, offs, args.read_len, max(args.steps, 12), pa)
                except Exception as e:   # extra evidence only: never lose the bench line over it
                    out[key] = {"error": repr(e)}
        if world == 1 and not strong and not args.no_full_pipeline and not args.no_cigar:
            for key, pa in (("full_pipeline", False), ("full_pipeline_with_pseudo_assembly", True)):
                try:
                    out[key] = full_pipeline(K, ctx, reads, db, offs, args.read_len, max(args.steps, 12), pa)
                except Exception as e:
                    out[key] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        # leave at once: with a process group up, a rank that unwinds normally can sit in the communicator's
        # teardown while the other ranks wait for it in a collective
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
