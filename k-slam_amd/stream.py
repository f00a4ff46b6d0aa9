"""The reference's batch loop over the C ABI: metagenomicAnalysis_Low_Mem (reference src/SLAM.h:159-268).

    while reads are left:                                     src/SLAM.h:193
        read --num-reads-at-once pairs from R1 / R2            :201-206  kslam_fastq_batch_end (host: counts line ends)
        alignToDatabase                                        :209      } kslam_submit_batch_fastq_text:
        score screen, pairing, insert-size limit + screen,     :210-229  }   FASTQ index, alignment, pairing, statistics,
          score screen [, pseudo-assembly + score screen]      :230-233  }   screens, pseudo-assembly, per-row NM / MD /
                                                                          }   log-probability -- all on the GPU
        writeSAMOutputPairs                                    :234-239  kslam_tail_finish_write_rows -> kslam_write_fd
        per-read taxonomy (LCA)                                :243-249  kslam_tail_classify, kslam_taxreport_add_batch
    _PerRead, report, _abbreviated                             :255-265  at the end (XML / abbreviated: the caller)

Batches are cut on the host BEFORE they are submitted (only line terminators are counted), so batch k + 1 is on its
way to the GPU while batch k is aligned and the host stage of batch k - 1 (SAM text, LCA) runs on a worker thread.
Batch boundaries are the reference's: `pairs_per_batch` records per stream per batch, which matters because the
insert-size limit is a per-batch statistic (src/PairedOverlap.h:314-360).

ctypes plumbing for tests/ and bench.py; every stage it calls is native code behind include/*.h.
"""
import ctypes as C
import os
import threading
import time

import numpy as np

from . import KslamError
from . import fastq as F
from . import tail as T


# every symbol include/kslam_stream.h declares
EXPORTS = ["kslam_stream_classify"]


class StreamParams(C.Structure):
    """kslam_stream_params"""
    _fields_ = [("pairs_per_batch", C.c_uint64), ("max_pairs_total", C.c_uint64), ("tail", T.TailParams), ("sam_fd", C.c_int32),
                ("per_read_fd", C.c_int32), ("sam_header", C.c_char_p), ("sam_header_len", C.c_uint64), ("depth", C.c_uint32),
                ("host_threads", C.c_uint32), ("pool_threads", C.c_uint32), ("passes", C.c_uint32)]


class StreamStats(C.Structure):
    """kslam_stream_stats"""
    _fields_ = [("n_batches", C.c_uint64), ("n_pairs", C.c_uint64), ("n_overlaps", C.c_uint64), ("n_read_pairs_aligned", C.c_uint64),
                ("n_alignment_pairs", C.c_uint64), ("sam_bytes", C.c_uint64), ("per_read_bytes", C.c_uint64),
                ("first_max_insert_size", C.c_uint32), ("batches_pseudo_on_host", C.c_uint32), ("seconds", C.c_double),
                ("seconds_waiting_for_gpu", C.c_double), ("seconds_waiting_for_host_stage", C.c_double),
                ("seconds_sam_text", C.c_double), ("seconds_classify", C.c_double), ("seconds_report", C.c_double),
                ("seconds_in_write", C.c_double), ("seconds_cutting", C.c_double), ("seconds_submitting", C.c_double),
                ("seconds_closing", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def classify_stream_native(ctx, index, r1_ptr, len1, r2_ptr, len2, pairs_per_batch, params, taxdb=None, report=None,
                           sam_fd=-1, per_read_fd=-1, sam_header=None, max_pairs_total=0, depth=0, passes=1, host_threads=0, pool_threads=0):
    """kslam_stream_classify: the same loop as classify_stream below, inside the library (no Python between the batches).
    Single-end data: params.paired = 0, r2_ptr = None, len2 = 0.
    -> dict of the statistics + tax_ids (uint32 array, empty without a taxdb)"""
    L = T.lib()
    L.kslam_stream_classify.argtypes = [C.c_void_p, C.POINTER(T.IndexView), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                        C.c_void_p, C.c_uint64, C.POINTER(StreamParams), C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_uint64), C.POINTER(StreamStats)]
    P = StreamParams(pairs_per_batch, max_pairs_total, params, sam_fd, per_read_fd, sam_header, len(sam_header) if sam_header else 0,
                     depth, host_threads, pool_threads, passes)
    st, ids, n_ids = StreamStats(), C.c_void_p(), C.c_uint64()
    rc = L.kslam_stream_classify(ctx._h, C.byref(index.view), taxdb._h if taxdb is not None else None,
                                 report._h if report is not None else None, r1_ptr, len1, r2_ptr, len2, C.byref(P),
                                 C.byref(ids), C.byref(n_ids), C.byref(st))
    if rc != 0:
        msg = L.kslam_tail_last_error().decode() or ctx._L.kslam_last_error(ctx._h).decode()
        raise KslamError(rc, msg)
    out = st.as_dict()
    n = int(n_ids.value)
    out["tax_ids"] = np.frombuffer((C.c_char * (4 * n)).from_address(ids.value), dtype=np.uint32).copy() if n else np.zeros(0, dtype=np.uint32)
    if ids.value:
        L.kslam_free(ids)
    return out


def _fd_writer():
    L = T.lib()
    return C.cast(L.kslam_write_fd, T.WRITE_FN)


def finish_rows_fd(params, reads, index, ov, cg, det, md, rp, pr, fd):
    """kslam_tail_finish_write_rows with one of the library's own writers (no Python in the write path): fd = a file
    descriptor (kslam_write_fd, written before the call returns), a kslam_amd.tail.SamWriter (kslam_write_queued: the
    text is handed to its background thread), or < 0: the text is formatted and dropped."""
    L = T.lib()
    st = T.TailStats()
    pool = np.ascontiguousarray(cg, dtype=np.uint32)
    if isinstance(fd, T.SamWriter):
        cb, user = fd.callback, fd._h
    elif fd >= 0:
        keep = C.c_int(fd)
        cb, user = _fd_writer(), C.cast(C.pointer(keep), C.c_void_p)
    else:
        cb, user = T.WRITE_FN(lambda u, d, n: 0), None
    p = lambda a: a.ctypes.data if a is not None and len(a) else None   # noqa: E731
    T._chk(L.kslam_tail_finish_write_rows(C.byref(params), C.byref(reads.view), C.byref(index.view), p(ov), len(ov),
                                          p(pool), len(pool), p(det), p(md), len(md) if md is not None else 0,
                                          p(rp), len(rp), p(pr), len(pr), cb, user, C.byref(st)))
    return st


def cut_batches(r1_ptr, len1, r2_ptr, len2, pairs_per_batch, max_pairs_total=0, threads=0):
    """The windows [(p1, e1, p2, e2, at_eof), ...] the reference's loop would read, batch by batch (a generator:
    the next boundary is looked for when the previous batch has been handed out)."""
    p1 = p2 = 0
    done_pairs = 0
    single = r2_ptr is None                                     # single-end data: one text (src/SLAM.h:198-206)
    while p1 < len1 or p2 < len2:
        want = pairs_per_batch
        if max_pairs_total:
            if done_pairs >= max_pairs_total:
                return
            want = min(want, max_pairs_total - done_pairs)      # readsPerGoTemp, src/SLAM.h:201-203
        e1, c1 = F.batch_end(r1_ptr + p1, len1 - p1, want, True, threads)
        e2 = 0 if single else F.batch_end(r2_ptr + p2, len2 - p2, want, True, threads)[0]
        e1 += p1
        e2 += p2
        last = e1 >= len1 or (not single and e2 >= len2)
        yield p1, e1, p2, e2, last
        done_pairs += want
        p1, p2 = e1, e2
        if last:
            return


def classify_stream(ctx, index, r1_ptr, len1, r2_ptr, len2, pairs_per_batch, params, taxdb=None, report=None,
                    sam_fd=-1, per_read_fd=-1, sam_header=None, max_pairs_total=0, depth=None, host_threads=0,
                    on_batch=None, before_batch=None, windows=None):
    """Runs the loop above.  r1_ptr / r2_ptr: ADDRESSES of the two FASTQ texts (page-locked memory from
    kslam_amd.HostBuffer goes up by DMA), index: a kslam_amd.tail index view (e.g. kslam_amd.db.Database),
    params: kslam_amd.tail.TailParams (paired; pseudo_assembly as wanted), taxdb / report: optional
    kslam_amd.taxonomy.TaxDB / Report.  Returns a dict: pairs, tax_ids (uint32, one per aligned read pair over all
    batches, in order), per-batch statistics and the wall-clock split.
    Test hooks, both called on the host-stage thread: before_batch(k, ov, cg, det, md, rp, pr, pair_stats, reads) sees
    the batch as the GPU returned it (the SAM writer sorts `pr` in place afterwards); on_batch(rec, ov, cg, rp, pr,
    reads) sees it after the host stage."""
    paired = bool(params.paired)
    if not paired and (r2_ptr is not None or len2):
        raise KslamError(2, "classify_stream: single-end data (params.paired == 0) is ONE text: r2_ptr must be None")
    t_start = time.perf_counter()
    stages = 3 | (4 if params.pseudo_assembly else 0)
    ctx.set_pairing(paired=paired, score_threshold=params.score_threshold, score_fraction=params.score_fraction, stages=stages)
    depth = depth or 3
    # the SAM text leaves through a background writer (kslam_sam_writer): the write of batch k runs under the
    # formatting of batch k + 1
    writer = T.SamWriter(sam_fd) if sam_fd >= 0 else None
    if sam_header is not None and writer is not None:
        writer.write(sam_header)
    P_host = T.TailParams.default(paired=paired, report_cigar=bool(params.report_cigar), score_threshold=params.score_threshold,
                                  num_sam_alignments=params.num_sam_alignments, score_fraction=params.score_fraction,
                                  pseudo_assembly=bool(params.pseudo_assembly), sam_xa=bool(params.sam_xa), threads=host_threads)
    P_write = T.TailParams.default(paired=paired, report_cigar=bool(params.report_cigar), score_threshold=params.score_threshold,
                                   num_sam_alignments=params.num_sam_alignments, score_fraction=params.score_fraction,
                                   pseudo_assembly=False, sam_xa=bool(params.sam_xa), threads=host_threads)
    out = {"batches": [], "tax_ids": [], "pairs": 0, "sam_bytes": 0, "per_read_bytes": 0}
    failure = []

    def host_stage(k, ov, cg, det, md, release, pairs, reads):
        try:
            t0 = time.perf_counter()
            rp, pr, pst = pairs
            on_gpu = bool(pst["stages_done"] & 4)
            if before_batch is not None:
                before_batch(k, ov, cg, det, md, rp, pr, pst, reads)
                t0 = time.perf_counter()
            st = finish_rows_fd(P_write if on_gpu or not params.pseudo_assembly else P_host, reads, index, ov, cg, det, md,
                                rp, pr, writer if writer is not None else -1)
            t1 = time.perf_counter()
            rec = {"batch": k, "pairs": reads.n_reads // 2 if paired else reads.n_reads, "overlaps": int(len(ov)), "alignment_pairs": int(st.n_paired_final),
                   "read_pairs_aligned": int(st.n_read_pairs), "max_insert_size": int(pst["max_insert_size"]),
                   "pseudo_assembly_on": ("gpu" if on_gpu else "host") if params.pseudo_assembly else None,
                   "sam_bytes": int(st.sam_bytes), "ms_sam": round((t1 - t0) * 1e3, 2)}
            if taxdb is not None:
                ids, text = taxdb.classify(P_write, reads, index, rp, pr, per_read=True)
                if per_read_fd >= 0 and text:
                    os.write(per_read_fd, text)
                out["per_read_bytes"] += len(text)
                t2 = time.perf_counter()
                if report is not None:
                    report.add_batch(reads, index, rp, pr, ids)
                out["tax_ids"].append(ids)
                rec["ms_classify"] = round((time.perf_counter() - t1) * 1e3, 2)
                rec["ms_report"] = round((time.perf_counter() - t2) * 1e3, 2)
            if on_batch is not None:
                on_batch(rec, ov, cg, rp, pr, reads)
            out["batches"].append(rec)
            out["pairs"] += rec["pairs"]
            out["sam_bytes"] += rec["sam_bytes"]
        except BaseException as e:      # reported by the main thread
            failure.append(e)
        finally:
            release()

    # windows: batch boundaries found earlier with cut_batches (timing runs read one text several times over)
    windows = iter(windows) if windows is not None else cut_batches(r1_ptr, len1, r2_ptr, len2, pairs_per_batch,
                                                                   max_pairs_total, host_threads)
    queue, worker, k = [], None, 0
    t_wait_gpu = t_wait_host = 0.0
    exhausted = False
    try:
        while True:
            while not exhausted and len(queue) < depth:
                w = next(windows, None)
                if w is None:
                    exhausted = True
                    break
                p1, e1, p2, e2, last = w
                # at_eof for inner windows too: a window ends right after a terminator (kslam_fastq_batch_end looked at
                # the byte behind a closing "\r"), so the end-of-stream rule adds nothing but keeps that "\r" a whole
                # terminator
                queue.append(ctx.submit_batch_fastq_text(r1_ptr + p1, e1 - p1, r2_ptr + p2 if paired else None, e2 - p2 if paired else 0,
                                                         max_pairs=0, at_eof=True))
            if not queue:
                break
            ta = time.perf_counter()
            ov, cg, det, md, release = ctx.collect_batch(queue.pop(0))
            pairs, reads = ctx.last_pairs, ctx.last_reads
            tb = time.perf_counter()
            t_wait_gpu += tb - ta
            if worker is not None:
                worker.join()
                worker = None
            t_wait_host += time.perf_counter() - tb
            if failure:
                release()
                break
            if reads is None or reads.n_reads == 0:     # an empty batch ends the loop (src/SLAM.h:207)
                release()
                break
            if pairs is None:
                release()
                raise KslamError(6, "the lane returned no device pairing")
            worker = threading.Thread(target=host_stage, args=(k, ov, cg, det, md, release, pairs, reads))
            worker.start()
            k += 1
    except BaseException as e:
        failure.insert(0, e)
    if worker is not None:
        worker.join()
    for tk in queue:                                  # (after a failure: let the lanes finish and drop their results)
        try:
            ctx.collect_batch(tk)[4]()
        except KslamError:
            pass
    ctx.set_pairing(stages=0)
    if writer is not None:
        try:
            out["sam_bytes_written"], out["s_in_write"] = writer.close()
        except KslamError as e:
            failure.append(e)
    if failure:
        raise failure[0]
    out["tax_ids"] = np.concatenate(out["tax_ids"]) if out["tax_ids"] else np.zeros(0, dtype=np.uint32)
    out["seconds"] = time.perf_counter() - t_start
    out["s_waiting_for_gpu"] = round(t_wait_gpu, 4)
    out["s_waiting_for_host_stage"] = round(t_wait_host, 4)
    return out
