"""bench_legs.py -- the legs of bench.py that are not the contract's skeleton: the end-to-end leg behind `value` (FASTQ text ->
SAM file + _PerRead file), the strong step on one GPU (`strong_n1` / `n1_same_workload`), the optional host-pointer leg and the
sink probes.  bench.py holds the flow (arguments, launcher, the timed hot path, verification, `roofline`, `cpu_baseline`, the
multi-GPU clocks) and imports these.  Nothing here touches oracle/."""
import importlib
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

W = entry.load_package() and importlib.import_module("kslam_amd.workload")   # generator + ground truth
ids_view = W.ids_view      # (k-slam_amd/workload.py: the tests use it too)


# ------------------------------------------------------------------------------------------------ the e2e legs
class FastqFiles:
    """F batches of synthetic pairs as two FASTQ texts in page-locked host memory (what a host holds after reading the
    two files), built on the GPU piece by piece."""

    def __init__(self, K, dev, batches, read_len, first_pair=0):
        n = sum(b.shape[0] // 2 for b in batches)
        self.rec = 2 + W.ID_DIGITS + 3 + read_len + 3 + read_len + 1
        self.n_pairs, self.len = n, n * self.rec
        self.h = [K.HostBuffer(self.len + 64) for _ in range(2)]
        qgen = torch.Generator(device=dev)
        qgen.manual_seed(4242)
        at = 0
        for b in batches:
            m = b.shape[0] // 2
            for mate in (0, 1):
                view = torch.from_numpy(self.h[mate].a[at * self.rec:(at + m) * self.rec].reshape(m, self.rec))
                for lo in range(0, m, 1_250_000):
                    hi = min(m, lo + 1_250_000)
                    txt, _ = W.fastq_text(b[mate * m + lo:mate * m + hi], mate + 1, first_pair=first_pair + at + lo, gen=qgen)
                    view[lo:hi].copy_(txt)
                    del txt
            at += m
        torch.cuda.synchronize()

    def close(self):
        for x in self.h:
            x.close()


def thread_cpu():
    """CPU seconds (user + system) of this process's threads, summed by thread name (/proc/self/task/*/stat)."""
    out, tick = {}, os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open("/proc/self/task/%s/stat" % tid).read()
        except OSError:
            continue
        name = st[st.index("(") + 1:st.rindex(")")]
        f = st[st.rindex(")") + 2:].split()
        out[name] = out.get(name, 0.0) + (int(f[11]) + int(f[12])) / tick
    return out


def usable_cpus():
    """CPUs the process may use: the affinity mask, capped by the cgroup v2 CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cgroup_throttled_ms():
    """Milliseconds this container's CPU quota has stalled it so far (cgroup v2 cpu.stat), or None."""
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("throttled_usec"):
                return int(line.split()[1]) / 1e3
    except OSError:
        pass
    return None


def fresh_file(path):
    if os.path.exists(path):
        os.unlink(path)
    return os.open(path, os.O_RDWR | os.O_CREAT | os.O_EXCL, 0o600)


def probe_sink(d, mb=256):
    """GB/s of `mb` MB of fresh bytes written to a new file in d with write() (8 MB calls) and closed."""
    path = os.path.join(d, "kslam_bench_probe_%d" % os.getpid())
    buf = os.urandom(8 << 20)
    best = 0.0
    for _ in range(2):
        if os.path.exists(path):
            os.unlink(path)
        t0 = time.perf_counter()
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        for _ in range(mb // 8):
            os.write(fd, buf)
        os.close(fd)
        best = max(best, mb * 2 ** 20 / (time.perf_counter() - t0) / 1e9)
        os.unlink(path)
    return round(best, 2)


def choose_sink(out_dir, need_bytes):
    """Where the e2e legs write: --out-dir, or (auto) the candidate -- /dev/shm (tmpfs), the temp directory (the container's
    disk-backed file system) -- whose page cache takes a new file fastest, among those with room for it."""
    if out_dir != "auto":
        return out_dir, None
    import shutil
    import tempfile
    probes = {}
    for d in dict.fromkeys(["/dev/shm", tempfile.gettempdir()]):
        try:
            if os.path.isdir(d) and os.access(d, os.W_OK) and shutil.disk_usage(d).free > 2 * need_bytes + (8 << 30):
                probes[d] = probe_sink(d)
        except OSError:
            pass
    if not probes:
        return "/dev/shm", {}
    return max(probes, key=probes.get), probes


def e2e_leg(K, ctx, files, pairs_per_batch, index_view, taxdb, steps, warmup, pseudo, reps=3, tag="e2e", out_dir="/dev/shm", native=True):
    """K steps of the reference's batch loop, FASTQ text in host memory to SAM + _PerRead files in out_dir."""
    S = importlib.import_module("kslam_amd.stream")
    T = importlib.import_module("kslam_amd.tail")
    X = importlib.import_module("kslam_amd.taxonomy")
    P = T.TailParams.default(pseudo_assembly=pseudo)
    wins = list(S.cut_batches(files.h[0].ptr, files.len, files.h[1].ptr, files.len, pairs_per_batch))
    F = len(wins)
    header = T.sam_header(index_view, b"SLAM --db synthetic R1.fq R2.fq")
    discard = out_dir == "/dev/null"      # the same leg with the SAM text thrown away by the kernel: what the sink costs
    sam_path = "/dev/null" if discard else os.path.join(out_dir, "kslam_bench_%d_%s.sam" % (os.getpid(), tag))
    pr_path = os.path.join("/dev/shm" if discard else out_dir, "kslam_bench_%d_%s_PerRead" % (os.getpid(), tag))

    def run(n_steps, keep_report=False):
        report = X.Report()
        # NEW files every time (unlink, then create): re-opening the previous repetition's file with O_TRUNC makes ext4 /
        # overlay flush the whole file to disk inside close() (its replace-via-truncate heuristic), 0.8 s per 8 GB
        sam_fd = os.open(sam_path, os.O_WRONLY) if discard else fresh_file(sam_path)
        pr_fd = fresh_file(pr_path)
        torch.cuda.synchronize()
        cpu0, thr0 = thread_cpu(), cgroup_throttled_ms()
        t0 = time.perf_counter()
        if native:
            # the loop inside the library (kslam_stream_classify); a text of F batches is read ceil(n_steps / F) times over
            res = S.classify_stream_native(ctx, index_view, files.h[0].ptr, files.len, files.h[1].ptr, files.len, pairs_per_batch, P,
                                           taxdb=taxdb, report=report, sam_fd=sam_fd, per_read_fd=pr_fd, sam_header=header,
                                           max_pairs_total=n_steps * pairs_per_batch, passes=-(-n_steps // F),
                                           host_threads=int(os.environ.get("KSLAM_BENCH_HOST_THREADS", "0")),
                                           pool_threads=int(os.environ.get("KSLAM_BENCH_POOL_THREADS", "0")))
            res.update(pairs=res["n_pairs"], per_read_bytes=res["per_read_bytes"], s_in_write=res["seconds_in_write"],
                       s_waiting_for_gpu=round(res["seconds_waiting_for_gpu"], 4),
                       s_waiting_for_host_stage=round(res["seconds_waiting_for_host_stage"], 4),
                       s_main=dict(cutting=round(res["seconds_cutting"], 4), submitting=round(res["seconds_submitting"], 4),
                                   closing=round(res["seconds_closing"], 4), classify=round(res["seconds_classify"], 4),
                                   report=round(res["seconds_report"], 4)),
                       batches=[{"batch": 0, "ms_sam": res["seconds_sam_text"] * 1e3, "ms_classify": (res["seconds_classify"] + res["seconds_report"]) * 1e3,
                                 "alignment_pairs": res["n_alignment_pairs"], "max_insert_size": res["first_max_insert_size"],
                                 "pseudo_assembly_on": (("host" if res["batches_pseudo_on_host"] else "gpu") if pseudo else None)}])
        else:
            res = S.classify_stream(ctx, index_view, files.h[0].ptr, files.len, files.h[1].ptr, files.len, pairs_per_batch, P,
                                    taxdb=taxdb, report=report, sam_fd=sam_fd, per_read_fd=pr_fd, sam_header=header,
                                    windows=[wins[i % F] for i in range(n_steps)])
        t_call = time.perf_counter() - t0
        os.close(sam_fd)
        os.close(pr_fd)
        torch.cuda.synchronize()
        res["wall"] = time.perf_counter() - t0
        cpu1, thr1 = thread_cpu(), cgroup_throttled_ms()
        res["cgroup_throttled_ms"] = None if thr0 is None or thr1 is None else round(thr1 - thr0, 1)
        res["cpu_s_by_thread"] = {k: round(v - cpu0.get(k, 0.0), 3) for k, v in sorted(cpu1.items()) if v - cpu0.get(k, 0.0) >= 0.005}
        res["s_call_and_close"] = [round(t_call, 4), round(res["wall"] - t_call, 4), round(res.get("seconds", 0.0), 4)]
        res["sam_file_bytes"] = (res.get("sam_bytes_written") or res.get("sam_bytes", 0) + len(header)) if discard else os.path.getsize(sam_path)
        res["per_read_file_bytes"] = os.path.getsize(pr_path)
        if keep_report:
            res["report"] = report
        else:
            report.close()
        return res
    try:
        run(max(warmup, 3))                       # lanes, page-locked buffers, the tail's arenas, the files' pages
        runs = [run(steps, keep_report=(i == reps - 1)) for i in range(reps)]
        walls = sorted(r["wall"] for r in runs)
        med = runs[[r["wall"] for r in runs].index(walls[len(walls) // 2])]
        last = runs[-1]
        # end of run (src/SLAM.h:255-265): the abbreviated table and the XML report over all batches
        t0 = time.perf_counter()
        summary = taxdb.summary(last["tax_ids"], last["pairs"])
        xml = taxdb.report_xml(last["report"], index_view, None, last["pairs"])
        with open(pr_path + ".xml", "wb") as f:
            f.write(xml)
        t_end = time.perf_counter() - t0
        last["report"].close()
        n_reads = 2 * pairs_per_batch * steps
        b = sorted(med["batches"], key=lambda r: r["batch"])
        same = all(r["sam_file_bytes"] == runs[0]["sam_file_bytes"] and r["per_read_file_bytes"] == runs[0]["per_read_file_bytes"]
                   and np.array_equal(r["tax_ids"], runs[0]["tax_ids"]) for r in runs)
        out = {
            "reads_per_s": round(n_reads / med["wall"], 1), "ms_per_step": round(med["wall"] / steps * 1e3, 3), "steps": steps,
            "repetitions_ms_per_step": [round(r["wall"] / steps * 1e3, 3) for r in runs],
            "pairs_per_batch": pairs_per_batch, "distinct_batches_in_the_text": F, "pseudo_assembly": bool(pseudo),
            "pseudo_assembly_on": b[-1]["pseudo_assembly_on"],
            "fastq_mb_per_batch": round(2 * files.len / F / 1e6, 1), "sam_mb_per_batch": round(med["sam_bytes"] / steps / 1e6, 1),
            "per_read_mb_per_batch": round(med["per_read_bytes"] / steps / 1e6, 2),
            "classified_read_pairs_per_batch": int(len(med["tax_ids"]) / steps),
            "alignment_pairs_per_batch": int(sum(r["alignment_pairs"] for r in b) / steps),
            "host_ms_per_batch": {"sam_text": round(sum(r["ms_sam"] for r in b) / steps, 2),
                                  "lca_per_read_and_report": round(sum(r.get("ms_classify", 0.0) for r in b) / steps, 2),
                                  "writer_thread_in_write": round(med.get("s_in_write", 0.0) / steps * 1e3, 2)},
            "sink": sam_path.rsplit("/", 1)[0],
            "s_main_thread_waiting_for_gpu": med["s_waiting_for_gpu"], "s_main_thread_waiting_for_host_stage": med["s_waiting_for_host_stage"],
            "s_main_thread_other": med.get("s_main"), "s_call_close_library": med.get("s_call_and_close"),
            "cpu_s_by_thread": med.get("cpu_s_by_thread"), "cgroup_throttled_ms": med.get("cgroup_throttled_ms"),
            "host_cpus_usable": usable_cpus(),
            "end_of_run_reports_s": round(t_end, 3), "end_of_run_report_bytes": {"abbreviated": len(summary), "xml": len(xml)},
            "including_end_of_run_reports": {"reads_per_s": round(n_reads / (last["wall"] + t_end), 1)},
            "verified": {"repetitions_identical": bool(same), "sam_file_bytes": int(runs[0]["sam_file_bytes"]),
                         "per_read_lines": int(len(runs[0]["tax_ids"])), "max_insert_size": b[0]["max_insert_size"]},
            "driver": "kslam_stream_classify (include/kslam_stream.h: the loop inside the library)" if native else "k-slam_amd/stream.py",
            "what": "FASTQ text (two files' worth, page-locked host memory) -> kslam_fastq_batch_end (batch boundaries) -> "
                    "kslam_submit_batch_fastq_text (GPU: FASTQ record index, alignToDatabase, score screen / pairing / insert-size "
                    "statistics / screens%s, per-row NM / MD / log-probability, the per-pair sort, the SAM records and the <out>_PerRead lines "
                    "with the per-read LCA (include/kslam_samtext.h; the mapping qualities' pow / log10 on the host's libm in between); %d "
                    "batches in flight) -> kslam_collect_batch -> the page-locked SAM block to kslam_sam_writer (background write() into the "
                    "SAM file), the per-read block to its file, kslam_taxreport_add_batch on a second host thread; wall clock of the K steps "
                    "incl. pipeline fill and drain.  host_ms_per_batch.sam_text is what is left of the SAM stage on the CPUs (handing the "
                    "block over); KSLAM_HOST_SAM_TEXT=1 brings the CPU formatter back (A/B)"
                    % (" / pseudo-assembly / second screen" if pseudo else "", 3),
        }
        return out
    finally:
        for pth in (sam_path, pr_path, pr_path + ".xml"):
            try:
                if pth != "/dev/null":
                    os.unlink(pth)
            except OSError:
                pass


def solo_strong(K, ctx, whole, read_len, total_pairs, index_view, tax_text, steps, warmup, out_dir, tag):
    """The strong step (ONE batch of `total_pairs` pairs, configs[3]'s shape) on ONE GPU, through the same stages the sharded
    clock of the --gpus N > 1 lines times on every rank -- alignToDatabase, pairing / insert-size limit / screens /
    pseudo-assembly, per-row walk, SAM text and per-read LCA written on the GPU, both files written -- with two contexts
    taking the steps in turn (the second borrows the index).  This is the N = 1 point of the scaling curve measured IN THE
    SAME RUN: rank 0 runs it alone after the N-rank clocks (the other ranks wait at a barrier), and the default N = 1 line
    runs it after its own legs.  -> dict(value, ms_per_step, steps, hot_path_ms_per_step, ...)"""
    T = importlib.import_module("kslam_amd.tail")
    X = importlib.import_module("kslam_amd.taxonomy")
    ST = importlib.import_module("kslam_amd.samtext")
    dev = whole.device
    n_reads = whole.shape[0]
    roffs = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
    qgen = torch.Generator(device=dev)
    qgen.manual_seed(4242)
    qual = torch.randint(33 + 20, 33 + 41, (n_reads * read_len + 64,), generator=qgen, device=dev, dtype=torch.uint8)
    torch.cuda.synchronize()
    rv = ids_view(T, total_pairs, read_len)
    taxdb = X.TaxDB(tax_text)
    ctx_b = ctx.sibling()
    pair = (ctx, ctx_b)
    for c in pair:
        c.load_reads_device(n_reads, whole.data_ptr(), roffs)
        c.load_qualities_device(qual.data_ptr())
        ST.set_annotations(c, index_view, taxdb)
        c._chk(ST.lib().kslam_load_read_ids(c._h, rv._keep[0].ctypes.data, rv._keep[1].ctypes.data))
    sam_path = os.path.join(out_dir, "kslam_bench_%d_%s.sam" % (os.getpid(), tag))
    pr_path = sam_path + "_PerRead"
    ms = {"align": 0.0, "pairing_screens_pseudo": 0.0, "row_details": 0.0, "sam_text_on_gpu": 0.0}
    seen = {}

    def worker(c, pst, fds, my_turn, next_turn):
        my_turn.wait()                                # blocks join the writer's queue in step order: one file
        t1 = time.perf_counter()
        n_sam, n_pr, tax = ST.sam_text_to_files(c, fds[0], fds[1], paired=True, num_alignments=10, sam_xa=False, want_per_read=True)
        ms["sam_text_on_gpu"] += time.perf_counter() - t1
        seen.update(sam_bytes=n_sam, per_read_lines=int(len(tax)), max_insert_size=int(pst["max_insert_size"]),
                    pseudo_on="gpu" if pst["stages_done"] & 4 else "host")
        next_turn.set()

    def steps_(k):
        sam_fd = fresh_file(sam_path)
        fds = (T.SamWriter(sam_fd), fresh_file(pr_path))
        turn = threading.Event()
        turn.set()
        flights = []
        for i in range(k):
            c = pair[i & 1]
            if i >= 2:
                flights[i - 2].join()                 # this context's previous step has left its buffers and the host
            t1 = time.perf_counter()
            c.align_resident()
            t2 = time.perf_counter()
            pst = c.pair_screen(paired=True, stages=7)
            t3 = time.perf_counter()
            c.row_details(of_pairs=True)
            t4 = time.perf_counter()
            ms["align"] += t2 - t1
            ms["pairing_screens_pseudo"] += t3 - t2
            ms["row_details"] += t4 - t3
            nxt = threading.Event()
            w = threading.Thread(target=worker, args=(c, pst, fds, turn, nxt))
            w.start()
            flights.append(w)
            turn = nxt
        for w in flights:
            w.join()
        fds[0].close()
        os.close(sam_fd)
        os.close(fds[1])
    try:
        steps_(max(warmup, 2))                        # both contexts once: their page-locked result buffers exist afterwards
        for k in ms:
            ms[k] = 0.0
        for pth in (sam_path, pr_path):
            if os.path.exists(pth):
                os.unlink(pth)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps_(steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        sizes = (os.path.getsize(sam_path), os.path.getsize(pr_path))
        # the hot path alone on the whole batch: the N = 1 point of hot_path.reads_per_s
        ctx.align_resident()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.align_resident()
        torch.cuda.synchronize()
        hot = (time.perf_counter() - t0) / steps
    finally:
        for pth in (sam_path, pr_path):
            if os.path.exists(pth):
                os.unlink(pth)
        ctx_b.close()
        taxdb.close()
    return {"what": "ONE batch of %d pairs per step on ONE GPU, the stages of classified_sharded with nothing to exchange: the N = 1 "
                    "point of the strong-scaling curve, measured in this run" % total_pairs,
            "value": round(2 * total_pairs * steps / el, 1), "unit": "reads/s", "ms_per_step": round(el / steps * 1e3, 3), "steps": steps,
            "stage_ms_per_step": {k: round(v / steps * 1e3, 2) for k, v in ms.items()},
            "hot_path_ms_per_step": round(hot * 1e3, 3), "hot_path_reads_per_s": round(2 * total_pairs / hot, 1),
            "sam_file_bytes": sizes[0], "per_read_file_bytes": sizes[1], "sam_mb_per_batch": round(seen["sam_bytes"] / 1e6, 1),
            "max_insert_size": seen["max_insert_size"], "pseudo_assembly_on": seen["pseudo_on"]}


def abi_path(K, ctx, reads, read_len, steps):
    """What a k-SLAM host that only swaps alignToDatabase sees (first sketch): reads[i].bases in host
    memory in (char **, lengths), overlap records + CIGAR pool back in host memory, through kslam_align_batch one
    batch at a time and through kslam_align_batch_async / kslam_wait_batch with two batches in flight."""
    host = np.ascontiguousarray(reads.cpu().numpy())
    n = host.shape[0]
    ptrs = (host.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(read_len)).astype(np.uint64)
    lens = np.full(n, read_len, dtype=np.uint32)
    pp, lp = ptrs.ctypes.data, lens.ctypes.data
    done_at = []

    def run(k):
        t0 = time.perf_counter()
        rows = 0
        pend = [ctx.submit_batch_pointers(n, pp, lp)]
        for i in range(k):
            if i + 1 < k:
                pend.append(ctx.submit_batch_pointers(n, pp, lp))
            ov, cg, release = ctx.wait_batch(pend.pop(0), copy=False)
            rows = len(ov)
            release()
            done_at.append(time.perf_counter())
        return time.perf_counter() - t0, rows
    run(3)
    del done_at[:]
    wall, rows = run(steps)
    steady = (done_at[-1] - done_at[0]) / (len(done_at) - 1)
    ov, cg, release = ctx.align_batch_pointers(n, pp, lp, copy=False)   # untimed: this context's page-locked buffers
    release()                                                           # (the pipelined run used the lanes')
    t0 = time.perf_counter()
    for i in range(3):
        ov, cg, release = ctx.align_batch_pointers(n, pp, lp, copy=False)
        if i < 2:
            release()
    sync_wall = (time.perf_counter() - t0) / 3
    n_out, n_cig = ctx.align_resident()
    r_ov, r_cg = ctx.fetch_results(n_out, n_cig)
    same = ov.tobytes() == r_ov.tobytes() and cg.tobytes() == r_cg.tobytes()
    release()
    return {
        "pipelined": {"reads_per_s": round(n * steps / wall, 1), "ms_per_batch": round(wall / steps * 1e3, 2),
                      "steady_state_ms_per_batch": round(steady * 1e3, 2)},
        "one_batch_at_a_time": {"reads_per_s": round(n / sync_wall, 1), "ms_per_batch": round(sync_wall * 1e3, 2)},
        "steps": steps, "h2d_mb_per_batch": round(n * read_len / 1e6, 1), "d2h_mb_per_batch": round((rows * 48 + n_cig * 4) / 1e6, 1),
        "equals_resident_result": bool(same),
        "what": "host pointers in -> host results out, alignToDatabase only: `one_batch_at_a_time` = kslam_align_batch in a loop (the "
                "drop-in of INTEGRATION.md's first sketch), `pipelined` = kslam_align_batch_async / kslam_wait_batch, two batches in flight",
    }
