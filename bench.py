#!/usr/bin/env python3
"""bench.py -- paired reads/s classified by the MI355X-native k-SLAM path (BASELINE.json's metric).

A "step" is one batch of synthetic paired reads through the reference's per-batch loop body (src/SLAM.h:193-249):
FASTQ text -> alignToDatabase (src/SLAM.h:59-79: k-mer extraction, sort, join, dedupe, Smith-Waterman, CIGAR) ->
score screen, pairing, insert-size statistics, screens [, pseudo-assembly] -> SAM text WRITTEN to a file ->
per-read taxonomy (LCA) WRITTEN to <out>_PerRead.  `value` = 2 x pairs x K / wall clock of K such steps, pipeline
fill and drain included (median of three repetitions), FASTQ text in page-locked host memory when the clock starts,
every PCIe byte inside.  Reported beside it, as `hot_path`: the alignToDatabase operator alone on a batch that is
resident in HBM (what `value` was in rounds 1-2), with the `roofline` objects attached to it.

  --config 1 (default, N = 1)  BASELINE configs[1]: 1 M x 2 x 150 bp pairs per batch vs the 5 Gb bacterial database,
                               --no-pseudo-assembly
  --config 2                   configs[2]: 10 M pairs per batch vs bacterial + 10 k viral genomes, pseudo-assembly on
  --config 4                   configs[4]: 10 M x 2 x 250 bp pairs per batch vs the bacterial database
  --gpus N > 1 (config 3)      ONE batch of --total-pairs pairs per step, read pairs sharded over the N GPUs: `value` = every
                               rank classifies its own read pairs (insert sizes all-gathered, pseudo-assembly by entry over
                               RCCL, its part of the SAM file written); `hot_path` = the merged records gathered to rank 0.
                               The line carries the N = 1 point of this workload measured in the same run
                               (`n1_same_workload`).  Without a launcher (WORLD_SIZE unset) bench.py starts the ranks itself.

The N = 1 line's legs: the hot path + its verification, the e2e leg (`value`), `cpu_baseline` (the reference's own
alignToDatabase compiled in place, timed on the host cores; oracle/ is used by this leg only) and `strong_n1` (the N > 1
lines' workload on this one GPU).  More evidence on request: --legs abi_path,other_sink,null_sink,pseudo | all.
The legs themselves: bench_legs.py.

Prints ONE JSON line on rank 0 (the driver contract of the task statement).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import threading
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
PIECES = 8            # a strong batch is generated in 8 fixed pieces: the same batch for N = 1, 2, 4, 8


# ------------------------------------------------------------------------------------------------ launcher
def kfd_gpu_count():
    """GPUs the amdgpu driver exposes, without touching the HIP runtime: /sys/class/kfd/kfd/topology/nodes/*/properties with
    simd_count > 0 (CPU nodes have 0), cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None when the files are not
    readable."""
    import glob
    n = 0
    files = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not files:
        return None
    for f in files:
        try:
            for line in open(f):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        except (OSError, ValueError):
            return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def device_identity(ordinal):
    """what tells two GPUs apart across processes: the uuid the runtime reports, else the PCI address, else the ordinal"""
    try:
        p = torch.cuda.get_device_properties(ordinal)
    except Exception:
        return "ordinal:%d" % ordinal
    u = getattr(p, "uuid", None)
    if u is not None and str(u).strip("0-") != "":
        return "uuid:%s" % u
    if hasattr(p, "pci_bus_id"):
        return "pci:%s:%s:%s" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
    return "ordinal:%d" % ordinal


def self_launch(args, argv):
    """--gpus N without a launcher: start the N rank processes here.  The parent never calls into the HIP runtime (on this
    pool an exec after the GPU was initialised takes the machine down, and Popen is fork + exec): the GPUs are counted from
    the kernel driver's topology files; when those are not readable the ranks themselves refuse a device that is not there.
    Relays rank 0's JSON line and exits non-zero if any rank dies."""
    n = args.gpus
    share = os.environ.get("KSLAM_BENCH_SHARE_GPU") == "1"
    ndev = kfd_gpu_count()
    if ndev is not None and n > ndev and not share:
        raise SystemExit("--gpus %d but %d device(s) visible (KSLAM_BENCH_SHARE_GPU=1 lets ranks share a GPU: tests only)" % (n, ndev))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), KSLAM_BENCH_LAUNCHED_BY="bench.py")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out_lines = []
    reader = threading.Thread(target=lambda: out_lines.extend(procs[0].stdout.read().splitlines()))
    reader.start()
    failed = None
    while any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0) and failed is None:
                failed = (r, p.returncode)
                for q in procs:                      # the exact children started above, by PID
                    if q.poll() is None:
                        q.terminate()
        time.sleep(0.2)
    reader.join()
    for r, p in enumerate(procs):
        if p.returncode != 0 and failed is None:
            failed = (r, p.returncode)
    js = [x for x in out_lines if x.startswith("{")]
    for x in out_lines:
        if not x.startswith("{"):
            print(x, file=sys.stderr)
    if failed is not None or not js:
        print("bench.py: rank %s exited with code %s" % (failed or ("?", "?")), file=sys.stderr)
        raise SystemExit(1)
    print(js[-1], flush=True)


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(K, db, offs, seed, n_genomes, n_pairs, read_len=READ_LEN, device=0, full=False, same_reads=None, why=None,
                 with_port=True):
    """The CPU path timed on the host cores (a reported baseline, never the product): `kind` "reference" = the reference's
    OWN `alignToDatabase` (src/SLAM.h:59-79, the template compiled from the header where it lies into
    oracle/_ref/libslam_ref.so: KMer.h, Overlap.h, SmithWaterman.h, ssw.c untouched), OpenMP on every CPU the job may use,
    phase times from its own log.txt stamps (src/sequenceTools.h:171-179); `kind` "port" = oracle/'s restatement, taken when
    oracle/_ref is absent (and reported beside the reference as `port` when it is there).
    full + same_reads: the WHOLE workload, on the very batch the hot path timed; otherwise a bounded sample, with `why`.
    The same reads then go through the HIP library: every row and CIGAR word is compared with the baseline's (modulo the
    revComp ties a multi-threaded reference run leaves open) and the GPU's time on them is reported,
    so that `speedup_on_sample` is one workload on both sides."""
    import oracle as O
    import tempfile
    n_genomes = min(n_genomes, len(offs) - 1)
    sub = db[:int(offs[n_genomes])].cpu()
    suboffs = np.asarray(offs[:n_genomes + 1])
    if same_reads is not None:
        reads = same_reads.cpu().numpy()
        n_pairs = reads.shape[0] // 2
    else:
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed)
        reads = make_reads(torch.device("cpu"), gen, sub, suboffs, n_pairs, read_len=read_len).numpy()
    reads = np.ascontiguousarray(reads)
    n_reads = reads.shape[0]
    subn = sub.numpy()
    cores = O.usable_cpus()        # one OpenMP thread per CPU the job may use (cgroup quota), not per hardware thread
    what = ("%d pairs x %d bp vs %s database genomes (%.0f Mb)%s; the whole alignToDatabase batch path incl. genome k-mer "
            "re-extraction and the (reads+genomes) sort, no tail, OpenMP on the CPUs the job's cgroup quota allows" % (
                n_pairs, read_len, "all %d" % n_genomes if full else "the first %d" % n_genomes, float(suboffs[-1]) / 1e6,
                "" if full else " -- NOT the whole database, which the CPU path cannot finish inside the bounded 10-30 s"))
    if full:
        what = "the WHOLE workload of this run%s: " % (", on the batch the hot path timed" if same_reads is not None else "") + what
    out, ref_rows, ref_cig = None, None, None
    if O.have_ref_slam():
        try:
            O.ref_slam_set_index_arrays(subn, suboffs)
            with tempfile.TemporaryDirectory() as wd:
                al, cg, dt, phases = O.ref_slam_align_to_database(reads.reshape(-1), np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len),
                                                                  threads=cores, workdir=wd)
            O.ref_slam_set_index_arrays(subn[:0], suboffs[:1])        # the reference's copy of the database goes back to the host
            out = {"value": round(n_reads / dt, 1), "unit": "reads/s", "cores": cores, "kind": "reference",
                   "what": "the reference's own alignToDatabase (src/SLAM.h:59-79) compiled in place from /root/reference/src "
                           "(oracle/_ref/libslam_ref.so), timed around the call",
                   "seconds": round(dt, 2), "phases_s": phases, "phases_from": "the reference's log.txt stamps (10 ms resolution)",
                   "n_alignments": int(len(al))}
            ref_rows, ref_cig = al, cg
        except Exception as e:     # fall back to the port, and say so
            out = None
            why = (why + "; " if why else "") + "reference run failed: %r" % (e,)
    port = None
    if out is None or with_port:
        rl = [reads[i].tobytes() for i in range(n_reads)]
        gl = [subn[int(suboffs[i]):int(suboffs[i + 1])].tobytes() for i in range(n_genomes)]
        kind_ssw = "own scalar SSW restatement"
        if O.use_reference_ssw(True):
            kind_ssw = "SSW core = the reference's own ssw.c (SSE2) from oracle/_ref"
        O.set_num_threads(cores)
        pal, pcg, ph = O.align_to_database(rl, gl)
        pdt = float(ph[5])              # seconds inside the C call (excludes the ctypes marshalling)
        O.use_reference_ssw(False)
        del rl, gl
        port = {"value": round(n_reads / pdt, 1), "unit": "reads/s", "cores": cores, "kind": "port", "what": "oracle/'s restatement; " + kind_ssw,
                "seconds": round(pdt, 2),
                "phases_s": dict(zip(("extract", "genome_kmers", "sort", "join", "sw"), (round(float(x), 2) for x in ph[:5]))),
                "n_alignments": int(len(pal))}
        if out is None:
            out, ref_rows, ref_cig = port, pal, pcg
            port = None
            if not O.have_ref_slam():
                out["why_not_the_reference"] = "oracle/_ref/libslam_ref.so is absent (it is built where /root/reference exists and travels with the snapshot)"
        else:
            v = O.compare_with_reference_rows(pal, pcg, ref_rows, ref_cig, lambda i: reads[i].tobytes(),
                                              lambda j: subn[int(suboffs[j]):int(suboffs[j + 1])].tobytes())
            port["equals_reference"] = v
            del pal, pcg
    out.update({"pairs": n_pairs, "read_len": read_len, "db_genomes": n_genomes, "db_bases": int(suboffs[-1]),
                "same_batch_as_hot_path": same_reads is not None, "sample": what})
    if port is not None:
        out["port"] = port
    if why:
        out["why"] = why
    try:   # the same sample through the HIP library: identical records and CIGARs?  and how long does the GPU take for it?
        c = K.Context(device=device)
        c.set_index_arrays(subn, suboffs)
        c.load_reads_arrays(reads.reshape(-1), np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len))
        n_out, n_cig = c.align_resident()
        t0 = time.perf_counter()
        for _ in range(3):
            n_out, n_cig = c.align_resident()
        gpu_ms = (time.perf_counter() - t0) / 3 * 1e3
        gov, gcg = c.fetch_results(n_out, n_cig)
        c.close()
        v = O.compare_with_reference_rows(gov, gcg, ref_rows, ref_cig, lambda i: reads[i].tobytes(),
                                          lambda j: subn[int(suboffs[j]):int(suboffs[j + 1])].tobytes())
        out["gpu_equals_reference" if out["kind"] == "reference" else "gpu_equals_cpu_on_sample"] = v
        if out["kind"] == "reference":
            out["gpu_equals_cpu_on_sample"] = v      # (the name rounds 1-5 used; the checker is now the reference itself)
        out["gpu_ms_on_sample"] = round(gpu_ms, 3)
        out["speedup_on_sample"] = round(out["seconds"] * 1e3 / gpu_ms, 1)
    except Exception as e:   # never lose the bench line over the extra check
        out["gpu_equals_cpu_on_sample"] = {"error": repr(e)}
    return out


# ------------------------------------------------------------------------------------------------ the legs (bench_legs.py)
from bench_legs import (FastqFiles, abi_path, cgroup_throttled_ms, choose_sink, e2e_leg, fresh_file, ids_view, solo_strong,  # noqa: E402
                        usable_cpus)



# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="BASELINE configs[k]; default 1 at --gpus 1, 3 (strong) at --gpus > 1")
    ap.add_argument("--pairs", type=int, default=0, help="read pairs per batch (default: 1 M for config 1, 10 M for 2 and 4)")
    ap.add_argument("--species", type=int, default=250)
    ap.add_argument("--strains", type=int, default=5)
    ap.add_argument("--genome-len", type=int, default=4_000_000)
    ap.add_argument("--viral", type=int, default=-1, help="viral genomes appended to the database (default: 10000 for config 2, else 0)")
    ap.add_argument("--repeats", action="store_true",
                    help="repeat-rich database: an rRNA-like 1.5 kb segment x5 in every genome, an insertion element in a third of them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=300000)
    ap.add_argument("--cpu-genomes", type=int, default=25)
    ap.add_argument("--cpu-full", action="store_true", help="cpu_baseline on the WHOLE workload of this run (minutes to hours of CPU time)")
    ap.add_argument("--cpu-baseline", choices=["auto", "full", "sample"], default="auto",
                    help="auto: the whole configs[1] workload on the hot path's own batch when the box has >= 16 usable CPUs and the "
                         "memory (about 30 s), else the bounded sample (--cpu-pairs / --cpu-genomes)")
    ap.add_argument("--no-cigar", action="store_true")
    ap.add_argument("--legs", default="", help="optional evidence legs of the N = 1 line, comma-separated or `all` (none by default: the "
                    "driver's run is the contract legs -- hot path, e2e `value`, cpu_baseline, strong_n1): abi_path (host pointers in / host "
                    "results out), other_sink (the e2e leg on the directory --out-dir auto did not choose), null_sink (SAM text to "
                    "/dev/null), pseudo (the e2e leg with the reference's default, pseudo-assembly on)")
    ap.add_argument("--no-abi-path", action="store_true", help=argparse.SUPPRESS)     # rounds 2-5 spelling: the leg is opt-in now
    ap.add_argument("--no-e2e", action="store_true", help="hot path only: `value` is then the resident-input rate and says so")
    ap.add_argument("--out-dir", default="auto", help="where the e2e legs write their SAM / _PerRead files (auto: the faster of "
                    "/dev/shm and the temp directory by a 256 MB write probe, see sink_probe in the line)")
    ap.add_argument("--no-sam-pipeline", dest="no_e2e", action="store_true", help=argparse.SUPPRESS)    # rounds 1-2 spelling (tools/*.sh)
    ap.add_argument("--no-full-pipeline", dest="no_e2e", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--read-len", type=int, default=0, help="150 (configs[1..3]) or 250 (configs[4])")
    ap.add_argument("--strong", action="store_true",
                    help="configs[3] shape: ONE batch of --total-pairs pairs per step, split over the GPUs (default when --gpus > 1)")
    ap.add_argument("--weak", action="store_true", help="with --gpus > 1: --pairs fresh pairs per GPU instead (hot path only)")
    ap.add_argument("--comm", choices=["auto", "kslam", "torch"], default="auto",
                    help="who moves the data between the ranks: kslam = the library's own RCCL path behind the C ABI (include/kslam_comm.h, "
                         "what a C++ host links: librccl opened by the library, no PyTorch in the transfers); torch = torch.distributed "
                         "(k-slam_amd/dist.py, the same protocol); auto = kslam when its communicator comes up on every rank, else torch")
    ap.add_argument("--strong-n1", choices=["auto", "off"], default="auto",
                    help="N = 1, configs[1]: after the line's own legs, also time the workload of the --gpus N > 1 lines (ONE batch of "
                         "--total-pairs pairs per step) on this one GPU, in this process, and attach it as `strong_n1`: the origin of the "
                         "strong-scaling curve.  Every --gpus N > 1 line measures the same thing itself (`n1_same_workload`)")
    ap.add_argument("--strong-reference", choices=["auto", "on", "off"], default=None, help=argparse.SUPPRESS)   # round 5 spelling of --strong-n1
    ap.add_argument("--total-pairs", type=int, default=10_000_000,
                    help="pairs per batch in --strong mode (the reference's --num-reads-at-once default, src/main.cpp:56)")
    args = ap.parse_args()
    if args.strong_reference == "off":
        args.strong_n1 = "off"
    legs = set(x for x in args.legs.replace("all", "abi_path,other_sink,null_sink,pseudo").split(",") if x)
    unknown = legs - {"abi_path", "other_sink", "null_sink", "pseudo"}
    if unknown:
        raise SystemExit("--legs: unknown leg(s) %s" % ", ".join(sorted(unknown)))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, sys.argv[1:])       # before this process makes any HIP call
    # ONE JSON line on stdout and nothing else: libraries that print to the C-level stdout (RCCL's version banner at
    # init) go to stderr for the whole run; the line itself is written to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    pending_line = [None]          # the graded line once it is complete enough to print, while optional legs still run
    emitted = [False]

    def emit(obj):
        """ONE JSON line, once, on the saved stdout"""
        if emitted[0] or obj is None:
            return
        emitted[0] = True
        try:
            sys.stdout.flush()
        except Exception:
            pass
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    def emit_on_exit():
        # a line that was complete when an optional leg took the process down (SIGTERM from a launcher's timeout, an exception
        # escaping to the interpreter's exit) is still written
        import atexit
        import signal
        atexit.register(lambda: emit(pending_line[0]))

        def on_signal(signum, frame):
            emit(pending_line[0])
            os._exit(128 + signum)
        for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            try:
                signal.signal(sig, on_signal)
            except (ValueError, OSError):
                pass
    emit_on_exit()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    share = os.environ.get("KSLAM_BENCH_SHARE_GPU") == "1"
    if world > torch.cuda.device_count() and not share:
        raise SystemExit("--gpus %d but %d device(s) visible" % (world, torch.cuda.device_count()))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    # KSLAM_BENCH_SHARE_GPU=1 (tests only): the ranks share the GPUs that exist, and talk through gloo with
    # host-staged pieces -- RCCL refuses two ranks on one device.  Everything else (sharding, count exchange,
    # export in batch terms, placement, verification) is the code a real N-GPU run executes.
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

    # ---- one rank per GPU, or the run is not a scaling point: every rank's device identity (uuid when the runtime reports one,
    # else PCI ids, else the ordinal) must be different.  A repeat is a hard failure of every rank -- exit code, not a JSON
    # field -- unless the tests asked for ranks that share a GPU.
    device_ids = [device_identity(local_rank)]
    if use_dist:
        box = [None] * world
        dist.all_gather_object(box, device_ids[0])
        device_ids = box
        if len(set(device_ids)) != world and not share:
            print("bench.py: ranks share a device (%s): not one rank per GPU; KSLAM_BENCH_SHARE_GPU=1 allows it for tests" % device_ids,
                  file=sys.stderr)
            raise SystemExit(3)

    K = entry.load_package()
    kdist = importlib.import_module("kslam_amd.dist")
    Cm = importlib.import_module("kslam_amd.comm")
    T = importlib.import_module("kslam_amd.tail")
    X = importlib.import_module("kslam_amd.taxonomy")
    S = importlib.import_module("kslam_amd.stream")
    strong = args.strong or (world > 1 and not args.weak) or args.config == 3
    config = args.config or (3 if strong else 1)
    if config == 3:
        strong = True
    read_len = args.read_len or (250 if config == 4 else READ_LEN)
    pairs = args.pairs or (1_000_000 if config in (1, 3) else 10_000_000)
    n_viral = args.viral if args.viral >= 0 else (10_000 if config == 2 else 0)
    pseudo = config != 1                     # configs[1] is quoted with --no-pseudo-assembly; the reference's default is on
    by_length = n_viral > 0
    if strong and (8 % world or args.total_pairs % PIECES):
        raise SystemExit("--strong needs 1, 2, 4 or 8 ranks and --total-pairs divisible by 8")
    # the sink of the legs that write files: rank 0 decides for everybody
    sink = [None, None]
    if rank == 0:
        # bytes one clock leaves in the directory: K batches of SAM text + _PerRead (two such files can exist at a time)
        sink = list(choose_sink(args.out_dir, (270 * args.total_pairs if strong else 450 * pairs) * max(args.steps, args.warmup, 3)))
    if use_dist:
        dist.broadcast_object_list(sink, src=0)
    args.out_dir, sink_probe = sink

    # ---- synthetic inputs, generated straight into HBM (data: synthetic) ----
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)                      # database: same on every rank (replicated index)
    t0 = time.time()
    db, offs = make_database(dev, gen, args.species, args.strains, args.genome_len, n_viral=n_viral, repeats=args.repeats)
    n_entries = len(offs) - 1
    if strong:
        # this rank's pairs [pair_lo, pair_hi) of the one batch, local layout [R1 of them | R2 of them]
        piece = args.total_pairs // PIECES
        mine = range(rank * PIECES // world, (rank + 1) * PIECES // world)
        pair_lo, pair_hi = mine[0] * piece, (mine[-1] + 1) * piece
        reads, truth = W.make_batch_in_pieces(dev, gen, db, offs, args.total_pairs, read_len, pieces=PIECES, first_piece=mine[0],
                                              n_pieces=len(mine), by_length=by_length)
        n_batch_pairs = args.total_pairs
    elif pairs > 2_000_000:
        reads, truth = W.make_batch_in_pieces(dev, gen, db, offs, pairs, read_len, pieces=PIECES, seed_base=2 + 1000 * rank,
                                              by_length=by_length)
        pair_lo, pair_hi = rank * pairs, (rank + 1) * pairs
        n_batch_pairs = pairs * world
    else:
        gen.manual_seed(2 + 1000 * rank)        # reads: a different shard of pairs per rank
        reads, truth = make_reads(dev, gen, db, offs, pairs, read_len=read_len, with_truth=True, by_length=by_length)
        pair_lo, pair_hi = rank * pairs, (rank + 1) * pairs
        n_batch_pairs = pairs * world
    torch.cuda.synchronize()
    t_gen = time.time() - t0

    ctx = K.Context(report_cigar=not args.no_cigar, device=local_rank)
    t0 = time.time()
    ctx.set_index_device(n_entries, db.data_ptr(), offs)
    t_index = time.time() - t0
    n_reads = reads.shape[0]
    roffs = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len))
    ctx.load_reads_device(n_reads, reads.data_ptr(), roffs)
    tax_text, entry_tax = W.taxonomy(args.species, args.strains, n_viral)
    index_view = T.IndexArrays(np.zeros(1, dtype=np.uint8), offs, taxonomy_ids=entry_tax)   # no host copy of the database

    # ---- who moves the data: the library's own communicator (RCCL behind the C ABI) or torch.distributed ----
    comm, comm_why = None, None

    def make_comm(c):
        """kslam_comm on context c, on every rank or on none: rank 0's unique id travels through the process group that exists
        anyway, and the ranks agree on the outcome before anybody uses it"""
        box, ok = [None], 1
        if rank == 0:
            try:
                box[0] = Cm.unique_id()
            except K.KslamError as e:
                box[0] = "error: %s" % e
        dist.broadcast_object_list(box, src=0)
        got = None
        if isinstance(box[0], bytes):
            try:
                got = Cm.Comm(c, box[0], rank, world)
            except K.KslamError as e:
                ok, box[0] = 0, "error: %s" % e
        else:
            ok = 0
        t = torch.tensor([ok], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t[0]) == 0:
            if got is not None:
                got.close()
            return None, (box[0] if isinstance(box[0], str) else "another rank could not create its communicator")
        return got, None
    if use_dist and strong and args.comm != "torch":
        if share and not os.environ.get("KSLAM_RCCL_LIB"):
            comm_why = "ranks share a GPU (RCCL refuses two ranks on one device) and KSLAM_RCCL_LIB names no stand-in"
        else:
            comm, comm_why = make_comm(ctx)
        if comm is None and args.comm == "kslam":
            raise SystemExit("--comm kslam: %s" % comm_why)
        if comm is not None:
            # what RCCL itself says about the communicator the data will move through: a count other than N is a hard failure
            mine = comm.info()
            if mine["comm_count"] != world or mine["comm_rank"] != rank:
                print("bench.py: rank %d: ncclCommCount = %d, ncclCommUserRank = %d in a world of %d" % (rank, mine["comm_count"], mine["comm_rank"], world),
                      file=sys.stderr)
                raise SystemExit(4)

    pending = []   # the gather of the previous batch, still in flight while this one is aligned
    merged = {}    # rank 0, --strong: the batch-global result of the last finished batch (device tensors)

    def drain():
        while pending:
            h = pending.pop()
            if h == "kslam_comm":
                # kslam_comm_gather_end: the transfers have landed; rank 0's arrays are the communicator's (no copy: valid until
                # the gather after the next one)
                d_rows, n_rows, d_pool, n_ops = comm.gather_end()
                if rank == 0:
                    merged["ov"] = torch.as_tensor(kdist._DevView(d_rows, max(n_rows, 1) * 48), device=dev)[:n_rows * 48]
                    merged["cg"] = torch.as_tensor(kdist._DevView(d_pool, max(n_ops, 1) * 4), device=dev)[:n_ops * 4]
                continue
            got = kdist.finish_gather(h)
            if strong and rank == 0:
                # every transfer landed in its final place (kslam_amd.dist.start_gather_sharded): rank 0
                # now HOLDS the batch-global result in the reference's order, and ran no kernel for it
                merged["ov"], merged["cg"] = got

    split = {"align": 0.0, "wait_for_previous_gather": 0.0, "counts_export_post": 0.0}   # host clock, this rank, timed steps
    gathered_bytes = [0]

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
            if comm is not None:
                comm.gather_begin(n_reads // 2, pair_lo, args.total_pairs)       # counts, export, one group of ncclSend / ncclRecv posted
                pending.append("kslam_comm")
            else:
                pending.append(kdist.start_gather_sharded(ctx, n_reads // 2, pair_lo, args.total_pairs, dev))
            split["wait_for_previous_gather"] += tc - tb
            split["counts_export_post"] += time.perf_counter() - tc
        elif use_dist:
            ov = torch.empty(n_out * 48, dtype=torch.uint8, device=dev)
            cg = torch.empty(n_cig * 4, dtype=torch.uint8, device=dev)
            ctx.copy_results_device(ov.data_ptr(), cg.data_ptr())
            drain()
            pending.append(kdist.start_gather(ov, cg))
        elif strong:
            ov = torch.empty(n_out * 48, dtype=torch.uint8, device=dev)
            cg = torch.empty(max(n_cig, 1) * 4, dtype=torch.uint8, device=dev)
            ctx.copy_results_device(ov.data_ptr(), cg.data_ptr())
            merged["ov"], merged["cg"] = ov, cg[:n_cig * 4]
        gathered_bytes[0] = n_out * 48 + n_cig * 4
        return n_out, n_cig

    def barrier():
        drain()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ================= the hot path: alignToDatabase on the resident batch (+ the gather when sharded) =================
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
    split_max, per_rank_align = dict(split), None
    if use_dist:
        te = torch.tensor([elapsed] + [split[k] for k in sorted(split)], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(te) for _ in range(world)]
        dist.all_gather(allr, te)
        elapsed = max(float(a[0]) for a in allr)
        split_max = {k: max(float(a[1 + i]) for a in allr) for i, k in enumerate(sorted(split))}
        per_rank_align = [round(float(a[1 + sorted(split).index("align")]) / args.steps * 1e3, 3) for a in allr]

    # ---- outside the timed region: is what was just timed RIGHT?  (no oracle here: the generator's
    # own ground truth, the reference's structural expectations of src/Tests.h:161-264, :321-330) ----
    def device_results():
        n_out, n_cig = ctx.align_resident()
        ov = torch.empty(n_out * 48, dtype=torch.uint8, device=dev)
        cg = torch.empty(max(n_cig, 1) * 4, dtype=torch.uint8, device=dev)
        ctx.copy_results_device(ov.data_ptr(), cg.data_ptr())
        return ov, cg[:n_cig * 4].view(torch.int32)
    ov_a, cg_a = device_results()
    verified = W.check_against_truth(ov_a, None if args.no_cigar else cg_a, truth, read_len)
    ov_b, cg_b = device_results()
    verified["run_to_run_identical"] = bool(ov_a.numel() == ov_b.numel() and torch.equal(ov_a, ov_b)
                                            and torch.equal(cg_a, cg_b))
    verified["ok"] = bool(verified["ok"] and verified["run_to_run_identical"])
    ranks_seen = 1
    if use_dist:   # every rank checked its own shard: sum the counts, AND the verdicts
        keys = [k for k, v in verified.items() if not isinstance(v, bool)]
        t = torch.tensor([verified[k] for k in keys] + [int(verified["ok"]), int(verified["run_to_run_identical"]), 1],
                         dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        for k, v in zip(keys, t[:len(keys)].tolist()):
            verified[k] = int(v)
        verified["ok"] = bool(int(t[-3]) == world)
        verified["run_to_run_identical"] = bool(int(t[-2]) == world)
        ranks_seen = int(t[-1])
    whole = None
    if strong and rank == 0 and "ov" in merged:
        # the merged batch: row count, order, and byte identity with what ONE context returns for the whole batch
        mc = W.overlap_columns(merged["ov"])
        n_rel = int(mc["rel"].max()) + 1026 if mc["rel"].numel() else 1026
        mkey = (mc["read"] * n_entries + mc["entry"]) * n_rel + (mc["rel"] + 1024)
        verified["merged_rows"] = int(mkey.numel())
        verified["merged_unsorted_neighbours"] = int((mkey[1:] < mkey[:-1]).sum()) if mkey.numel() > 1 else 0
        del mc, mkey
        if world > 1:
            # the whole batch once more, in THIS rank's context alone (outside the timed region)
            del ov_a, cg_a, ov_b, cg_b
            whole, _ = W.make_batch_in_pieces(dev, gen, db, offs, args.total_pairs, read_len, pieces=PIECES, by_length=by_length,
                                              with_truth=False)
            ctx.load_reads_device(whole.shape[0], whole.data_ptr(),
                                  np.arange(whole.shape[0] + 1, dtype=np.uint64) * np.uint64(read_len))
            ov_a, cg_a = device_results()
            ov_b = cg_b = None
            ctx.load_reads_device(n_reads, reads.data_ptr(), roffs)      # back to this rank's shard for the second clock
        else:
            whole = reads
        verified["merged_equals_single_context"] = bool(
            merged["ov"].numel() == ov_a.numel() and torch.equal(merged["ov"], ov_a) and
            torch.equal(merged["cg"].view(torch.int32), cg_a))
        verified["ok"] = bool(verified["ok"] and verified["merged_equals_single_context"])
        verified["ok"] = bool(verified["ok"] and verified["merged_unsorted_neighbours"] == 0)
    del ov_a, cg_a, ov_b, cg_b

    # ================= strong mode, second clock: through the batch-global tail to SAM text + per-read taxa on rank 0 =================
    classified = None
    n1_same = None
    if strong and not args.no_e2e and not args.no_cigar:
        taxdb = X.TaxDB(tax_text) if rank == 0 else None
        tail_ctx = rv = None
        if rank == 0:
            # rank 0 keeps the WHOLE batch (bases + qualities) resident in a sibling context that borrows the index: the
            # per-row walk of the SAM writer (NM / MD / log-probability) needs every read of the batch next to the genomes
            tail_ctx = ctx.sibling()
            tail_ctx.load_reads_device(whole.shape[0], whole.data_ptr(), np.arange(whole.shape[0] + 1, dtype=np.uint64) * np.uint64(read_len))
            qgen = torch.Generator(device=dev)
            qgen.manual_seed(4242)
            qual = torch.randint(33 + 20, 33 + 41, (whole.shape[0] * read_len + 64,), generator=qgen, device=dev, dtype=torch.uint8)
            torch.cuda.synchronize()
            tail_ctx.load_qualities_device(qual.data_ptr())
            del qual
            rv = ids_view(T, args.total_pairs, read_len)
        # the ranks of a node share its CPUs: each rank's host stages take their share, not one thread per usable CPU each
        host_threads = max(2, usable_cpus() // world) if world > 1 else 0
        P_write = T.TailParams.default(pseudo_assembly=False, threads=host_threads)
        P_host = T.TailParams.default(pseudo_assembly=True, threads=host_threads)
        sam_path = os.path.join(args.out_dir, "kslam_bench_%d_strong.sam" % os.getpid())
        pr_path = sam_path + "_PerRead"
        tail_ms = {"adopt_pair_screen_details": 0.0, "download": 0.0, "host_sam_and_lca": 0.0}
        tail_out = {}

        def host_stage(ov, cg, det, md, rp, pr, pst, releases, fds):
            t1 = time.perf_counter()
            st = S.finish_rows_fd(P_write if pst["stages_done"] & 4 else P_host, rv, index_view, ov, cg, det, md, rp, pr, fds[0])
            ids, text = taxdb.classify(P_write, rv, index_view, rp, pr, per_read=True)
            os.write(fds[1], text)
            for r in releases:
                r()
            tail_ms["host_sam_and_lca"] += time.perf_counter() - t1
            tail_out.update(sam_bytes=int(st.sam_bytes), per_read_lines=int(len(ids)), alignment_pairs=int(st.n_paired_final),
                            max_insert_size=int(pst["max_insert_size"]), pseudo_on="gpu" if pst["stages_done"] & 4 else "host")

        trace = os.environ.get("KSLAM_BENCH_TRACE") == "1"
        t_origin = [time.perf_counter()]

        def mark(what):
            if trace:
                print("[trace rank %d] %8.1f ms %s" % (rank, (time.perf_counter() - t_origin[0]) * 1e3, what), file=sys.stderr, flush=True)

        dl_done = [None]     # set when the previous batch's download has left tail_ctx's buffers

        def tail_worker(pst, fds, n_rows, downloaded, my_turn, next_turn):
            mark("rank-0 tail: download starts")
            t2 = time.perf_counter()
            ov, cg, rel1 = tail_ctx.take_results()
            det, md, rel2 = tail_ctx.take_row_details(len(ov), copy=False)
            rp, pr, rel3 = tail_ctx.take_pairs(copy=False)
            tail_ms["download"] += time.perf_counter() - t2
            downloaded.set()
            mark("rank-0 tail: download done")
            my_turn.wait()                                # host stages in batch order: one SAM file
            mark("rank-0 tail: host stage starts")
            host_stage(ov, cg, det, md, rp, pr, pst, (rel1, rel2, rel3), fds)
            mark("rank-0 tail: host stage done")
            next_turn.set()

        def tail_step(fds, turn):
            # rank 0, after the gather of this batch has landed: the reference's per-batch steps after alignToDatabase
            # (src/SLAM.h:210-249) on the merged batch.  The GPU part (adopt, pairing / screens / pseudo-assembly, per-row
            # walk) runs here; the download and the host stage run on a worker thread under the NEXT batch's alignment
            # (tail_ctx has its own stream and buffers), host stages one after the other.
            if dl_done[0] is not None:
                dl_done[0].wait()                         # tail_ctx's result buffers are free again
            if len(workers) >= 3:
                workers[-3].join()                        # at most three batches' results wait for the host
            mark("rank-0 tail: adopt starts")
            t1 = time.perf_counter()
            n_rows, n_ops = merged["ov"].numel() // 48, merged["cg"].numel() // 4
            tail_ctx.adopt_results_device(merged["ov"].data_ptr(), n_rows, merged["cg"].data_ptr(), n_ops)
            pst = tail_ctx.pair_screen(paired=True, stages=7)
            tail_ctx.row_details(of_pairs=True)
            tail_ms["adopt_pair_screen_details"] += time.perf_counter() - t1
            mark("rank-0 tail: GPU part done")
            dl_done[0] = threading.Event()
            nxt = threading.Event()
            w = threading.Thread(target=tail_worker, args=(pst, fds, n_rows, dl_done[0], turn, nxt))
            w.start()
            workers.append(w)
            return nxt

        workers = []

        def classified_steps(k):
            fds = None
            turn = threading.Event()
            turn.set()
            if rank == 0:
                sam_fd = fresh_file(sam_path)
                fds = (T.SamWriter(sam_fd), fresh_file(pr_path))
            for _ in range(k):
                step()
                if rank == 0:
                    drain()                               # this batch's rows have landed on rank 0
                    turn = tail_step(fds, turn)
            if rank == 0:
                for w in workers:
                    w.join()
                del workers[:]
                dl_done[0] = None
                fds[0].close()
                os.close(sam_fd)
                os.close(fds[1])
        classified_steps(max(args.warmup, 3))             # three batches in flight: their page-locked result buffers exist afterwards
        for k in tail_ms:
            tail_ms[k] = 0.0
        if rank == 0:
            for pth in (sam_path, pr_path):               # (see the sharded clock: the old files go before the clock starts)
                if os.path.exists(pth):
                    os.unlink(pth)
        barrier()
        t0 = time.perf_counter()
        classified_steps(args.steps)
        barrier()
        el2 = time.perf_counter() - t0
        if use_dist:
            te = torch.tensor([el2], dtype=torch.float64, device=cdev)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el2 = float(te[0])
        if rank == 0:
            classified = {"elapsed": el2, "rank0_tail_ms_per_step": {k: round(v / args.steps * 1e3, 2) for k, v in tail_ms.items()},
                          "sam_mb_per_batch": round(tail_out["sam_bytes"] / 1e6, 1), "per_read_lines_per_batch": tail_out["per_read_lines"],
                          "alignment_pairs_per_batch": tail_out["alignment_pairs"], "max_insert_size": tail_out["max_insert_size"],
                          "pseudo_assembly_on": tail_out["pseudo_on"],
                          "sam_file_bytes": os.path.getsize(sam_path), "per_read_file_bytes": os.path.getsize(pr_path)}
            if os.path.getsize(sam_path) < (1 << 30):      # small runs (the tests): the files' checksums, to compare lines of different runs
                import zlib
                classified["sam_file_crc32"] = zlib.crc32(open(sam_path, "rb").read())
                classified["per_read_file_crc32"] = zlib.crc32(open(pr_path, "rb").read())

        # ================= third clock: the tail SHARDED like the alignment =================
        # every rank: align its pairs -> pairing on its own rows -> all-gather of the insert sizes (the limit is a statistic
        # of the whole batch) -> screens -> all-gather of the alignment-pair records -> pseudo-assembly (per entry over all
        # read pairs) -> per-row NM / MD / log-probability, SAM text and per-read LCA of ITS read pairs into ITS part files.
        # No overlap rows travel; the part files concatenated in rank order are the SAM / _PerRead files.
        n_local = n_reads // 2
        qgen = torch.Generator(device=dev)
        qgen.manual_seed(4242)                      # the whole batch's qualities as rank 0's tail context holds them
        qual_all = torch.randint(33 + 20, 33 + 41, (2 * args.total_pairs * read_len + 64,), generator=qgen, device=dev, dtype=torch.uint8)
        qual_loc = torch.cat([qual_all[pair_lo * read_len:pair_hi * read_len],
                              qual_all[(args.total_pairs + pair_lo) * read_len:(args.total_pairs + pair_hi) * read_len],
                              torch.zeros(64, dtype=torch.uint8, device=dev)]).contiguous()
        del qual_all
        torch.cuda.synchronize()
        ctx.load_qualities_device(qual_loc.data_ptr())
        rv_loc = ids_view(T, n_local, read_len, first_pair=pair_lo)
        taxdb_s = X.TaxDB(tax_text)
        part_sam = os.path.join(args.out_dir, "kslam_bench_%s_part%d.sam" % (os.environ.get("MASTER_PORT", "solo") if use_dist else str(os.getpid()), rank))
        part_pr = part_sam + "_PerRead"
        # two contexts (the second borrows the index) take the steps in turn: the download and the host stage of step k run
        # on a worker thread under the alignment of step k + 1 on the other context; host stages one after the other
        ctx_b = ctx.sibling()
        ctx_b.load_reads_device(n_reads, reads.data_ptr(), roffs)
        ctx_b.load_qualities_device(qual_loc.data_ptr())
        sh_ctx = (ctx, ctx_b)
        comm_b = None
        if comm is not None:
            comm_b, why_b = make_comm(ctx_b)     # the two contexts take the steps in turn: a communicator each
            if comm_b is None:
                raise SystemExit("a second kslam_comm could not be created: %s" % why_b)
        sh_comm = (comm, comm_b)
        sh_ms = {"align": 0.0, "pairing_gathers_pseudo": 0.0, "row_details": 0.0, "download_on_worker": 0.0, "host_sam_and_lca": 0.0}
        sh_out = {"moved": 0}
        # the SAM records and per-read lines written on each rank's GPU (include/kslam_samtext.h); KSLAM_HOST_SAM_TEXT=1: on its CPUs
        ST = importlib.import_module("kslam_amd.samtext")
        device_text = os.environ.get("KSLAM_HOST_SAM_TEXT") != "1"
        if device_text:
            ids_u8, ids_off = rv_loc._keep[0], rv_loc._keep[1]
            for c in sh_ctx:
                ST.set_annotations(c, index_view, taxdb_s)
                c._chk(ST.lib().kslam_load_read_ids(c._h, ids_u8.ctypes.data, ids_off.ctypes.data))
            sh_ms["sam_text_on_gpu"] = 0.0

        def sh_host(ov, cg, det, md, rp, pr, pst, releases, fds):
            t1 = time.perf_counter()
            st = S.finish_rows_fd(P_write if pst["stages_done"] & 4 else P_host, rv_loc, index_view, ov, cg, det, md, rp, pr, fds[0])
            ids, text = taxdb_s.classify(P_write, rv_loc, index_view, rp, pr, per_read=True)
            os.write(fds[1], text)
            for r in releases:
                r()
            sh_ms["host_sam_and_lca"] += time.perf_counter() - t1
            sh_out.update(sam_bytes=int(st.sam_bytes), per_read_lines=int(len(ids)), pseudo_on="gpu" if pst["stages_done"] & 4 else "host",
                          max_insert_size=int(pst["max_insert_size"]))

        def sh_worker_device_text(c, pst, fds, downloaded, my_turn, next_turn):
            # the text is written where the rows lie: nothing but the text (and the taxonomy ids) leaves the GPU
            my_turn.wait()                                # blocks join the writer's queue in step order: one part file
            t1 = time.perf_counter()
            n_sam, n_pr, tax = ST.sam_text_to_files(c, fds[0], fds[1], paired=True, num_alignments=10, sam_xa=False, want_per_read=True)
            sh_ms["sam_text_on_gpu"] += time.perf_counter() - t1
            downloaded.set()
            sh_out.update(sam_bytes=n_sam, per_read_lines=int(len(tax)), pseudo_on="gpu", max_insert_size=int(pst["max_insert_size"]))
            next_turn.set()

        def sh_worker(c, pst, fds, downloaded, my_turn, next_turn):
            if device_text and (pst["stages_done"] & 4):
                return sh_worker_device_text(c, pst, fds, downloaded, my_turn, next_turn)
            mark("download starts")
            t1 = time.perf_counter()
            ov, cg, rel1 = c.take_results()
            det, md, rel2 = c.take_row_details(len(ov), copy=False)
            rp, pr, rel3 = c.take_pairs(copy=False)
            sh_ms["download_on_worker"] += time.perf_counter() - t1
            downloaded.set()
            mark("download done")
            my_turn.wait()                                # host stages in step order: one part file
            mark("host stage starts")
            sh_host(ov, cg, det, md, rp, pr, pst, (rel1, rel2, rel3), fds)
            mark("host stage done")
            next_turn.set()

        def sharded_steps(k):
            sam_fd = fresh_file(part_sam)
            fds = (T.SamWriter(sam_fd), fresh_file(part_pr))
            turn = threading.Event()
            turn.set()
            flights = []                                  # (worker thread, its download-done event) per step
            for i in range(k):
                c = sh_ctx[i & 1]
                if i >= 2:
                    # this context's previous step has left its buffers AND the host: the host stage (SAM text, the writer
                    # behind it) is the slower side, and letting the GPU run further ahead only makes it slower still
                    # (measured: 494 ms per step like this, 590-800 with the GPU three steps ahead)
                    flights[i - 2][0].join()
                mark("step %d: align starts" % i)
                t1 = time.perf_counter()
                c.align_resident()
                t2 = time.perf_counter()
                if use_dist and sh_comm[i & 1] is not None:
                    pst, moved = sh_comm[i & 1].sharded_tail(True, 0, 0.95, True)     # kslam_comm_sharded_tail
                    sh_out["moved"] = moved
                elif use_dist:
                    pst, moved = kdist.sharded_tail(c, dev, True, 0, 0.95, True)
                    sh_out["moved"] = moved
                else:
                    pst = c.pair_screen(paired=True, stages=7)
                t3 = time.perf_counter()
                c.row_details(of_pairs=True)
                t4 = time.perf_counter()
                sh_ms["align"] += t2 - t1
                sh_ms["pairing_gathers_pseudo"] += t3 - t2
                sh_ms["row_details"] += t4 - t3
                mark("step %d: GPU part done" % i)
                nxt, done = threading.Event(), threading.Event()
                w = threading.Thread(target=sh_worker, args=(c, pst, fds, done, turn, nxt))
                w.start()
                flights.append((w, done))
                turn = nxt
            for w, _ in flights:
                w.join()
            fds[0].close()
            os.close(sam_fd)
            os.close(fds[1])
        sharded_steps(max(args.warmup, 2))                # both contexts once: their page-locked result buffers exist afterwards
        # three repetitions of the K steps, the median counts (the host side of a step -- SAM text into the page cache --
        # varies by 20 % between repetitions on the bench boxes); every rank takes the same one: the max over ranks decides
        reps = []
        for _ in range(3):
            for k in sh_ms:
                sh_ms[k] = 0.0
            for pth in (part_sam, part_pr):              # freeing the last repetition's 25 GB of page cache takes a second:
                if os.path.exists(pth):                  # not inside the clock (a run of the tool starts with no file)
                    os.unlink(pth)
            barrier()
            thr0 = cgroup_throttled_ms()
            t0 = time.perf_counter()
            sharded_steps(args.steps)
            barrier()
            el = time.perf_counter() - t0
            thr1 = cgroup_throttled_ms()
            if use_dist:
                te = torch.tensor([el] + [sh_ms[k] for k in sorted(sh_ms)], dtype=torch.float64, device=cdev)
                dist.all_reduce(te, op=dist.ReduceOp.MAX)
                el = float(te[0])
                mx = {k: float(v) for k, v in zip(sorted(sh_ms), te[1:].tolist())}
            else:
                mx = dict(sh_ms)
            reps.append((el, mx, thr0, thr1))
        sh_reps = [round(r[0] / args.steps * 1e3, 1) for r in reps]
        el3, sh_max, thr0, thr1 = sorted(reps, key=lambda r: r[0])[1]
        # two more batches through both forms (one per context of the sharded form), outside the clocks: the part files
        # in rank order must BE rank 0's files
        sharded_steps(2)
        classified_steps(2)
        barrier()
        if rank == 0:
            base = part_sam.rsplit("part0", 1)
            parts_ok = True
            for full, suffix in ((sam_path, ""), (pr_path, "_PerRead")):
                # every file holds the batch twice (a part file: once from each of its rank's two contexts)
                halves = []
                for r in range(world):
                    part = open(base[0] + "part%d" % r + base[1] + suffix, "rb").read()
                    parts_ok = parts_ok and len(part) % 2 == 0 and part[:len(part) // 2] == part[len(part) // 2:]
                    halves.append(part[:len(part) // 2])
                once = b"".join(halves)
                parts_ok = parts_ok and once + once == open(full, "rb").read()
            classified["sharded"] = {
                "elapsed": el3, "repetitions_ms_per_step": sh_reps, "ms_per_step_max_over_ranks": {k: round(v / args.steps * 1e3, 2) for k, v in sh_max.items()},
                "bytes_received_from_other_ranks_per_step": int(sh_out["moved"]), "pseudo_assembly_on": sh_out["pseudo_on"],
                "pseudo_assembly_form": "entries partitioned over the ranks (entry e on rank e mod N): all-to-all of 16-byte heads, 4-byte scores back",
                "host_threads_per_rank": host_threads, "cgroup_throttled_ms": None if thr0 is None or thr1 is None else round(thr1 - thr0, 1),
                "max_insert_size": sh_out["max_insert_size"],
                "part_files_in_rank_order_equal_rank0_files": bool(parts_ok)}
            if os.environ.get("KSLAM_BENCH_KEEP_FILES"):      # debugging: the files side by side
                import shutil
                for pth in (sam_path, pr_path, part_sam, part_pr):
                    shutil.copy(pth, os.path.join(os.environ["KSLAM_BENCH_KEEP_FILES"], os.path.basename(pth).replace(str(os.getpid()), "X")))
            for pth in (sam_path, pr_path):
                os.unlink(pth)
            tail_ctx.close()
            taxdb.close()
        if use_dist:
            dist.barrier()
        for pth in (part_sam, part_pr):
            try:
                os.unlink(pth)
            except OSError:
                pass
        taxdb_s.close()
        if comm_b is not None:
            comm_b.close()
        ctx_b.close()
        del qual_loc
        # ---- the N = 1 point of THIS workload, in THIS run: rank 0 alone repeats the step on the whole batch, the others wait
        if world > 1:
            if rank == 0:
                try:
                    n1_same = solo_strong(K, ctx, whole, read_len, args.total_pairs, index_view, tax_text, args.steps, args.warmup, args.out_dir, "n1")
                except Exception as e:
                    n1_same = {"error": repr(e)}
            dist.barrier()

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

    comm_facts = None
    if comm is not None:
        # what RCCL itself reports on every rank: ncclCommCount / ncclCommUserRank / the device / the library that was opened
        mine = comm.info()
        t = torch.tensor([mine["comm_count"], mine["comm_rank"], mine["device"], mine["rccl_version"]], dtype=torch.int64, device=cdev)
        allf = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allf, t)
        comm_facts = {"ncclCommCount_per_rank": [int(a[0]) for a in allf], "ncclCommUserRank_per_rank": [int(a[1]) for a in allf],
                      "device_per_rank": [int(a[2]) for a in allf], "ncclGetVersion": mine["rccl_version"], "library": mine["library"]}
        comm.close()
        comm = "closed"

    if rank == 0:
        Ksteps = args.steps
        tm = {k: v / Ksteps for k, v in acc.items()}
        total_reads = 2 * n_batch_pairs * Ksteps
        passes = int(round(tm["sort_passes"]))
        launches = max(tm["n_scatter_launches"], 1)
        launch_ms = tm["ms_sort_scatter"] / launches
        n_sorted = tm["n_kmers_kept"]      # the records the per-batch sort moves: read k-mers the genome filter let through
        # one scatter launch: 16 B in + 16 B out per record, + 1 B per record in every launch but the last of a sort (the
        # next pass's digit byte, radix_sort.hip): averaged over the sort's launches
        per_launch_bytes = (n_sorted / max(tm["n_chunks"], 1)) * (32 + (passes - 1) / max(passes, 1))
        achieved = per_launch_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        traffic = sort_pmc = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and config == 1:
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("k_scatter_bytes_per_launch")
                sort_pmc = (tj.get("sort_phase") or {}).get("bytes_per_call")     # PMC, whole phase (profiles/run_profiles.sh)
            except Exception:
                traffic = sort_pmc = None
        sort_bytes = n_sorted * 16 * (2 * passes + 1)
        # the SW phase against the VALU issue rate: instruction count of the phase's kernels per alignment call from the
        # committed counter run of this same workload (tools/pmc_valu.sh -> profiles/sw_valu.json), time measured live
        sw_valu = None
        vpath = os.path.join(ROOT, "profiles", "sw_valu.json")
        if os.path.exists(vpath) and config == 1 and pairs == 1_000_000 and args.species == 250 and tm["ms_sw"] > 0:
            try:
                vj = json.load(open(vpath))
                instr = float(vj["sw_phase_per_align"]["valu_wave_instr"])
                rate = instr / (tm["ms_sw"] * 1e-3) / 1e9
                sw_valu = {"bound": "valu-issue", "kernels": "k_sw_plan + k_sw_band<...> tiers + k_sw (the SW phase, the largest share of the step)",
                           "achieved": round(rate, 1), "peak": 1228.8, "unit": "G wave-instr/s", "frac": round(rate / 1228.8, 4),
                           "sustained_for_4_cycle_kinds": 575.0, "sustained_for_2_cycle_kinds": 1084.0,
                           "valu_wave_instr_per_step": int(instr), "ms": round(tm["ms_sw"], 3),
                           "note": "peak = 256 CUs x 4 SIMD x 2.4 GHz / 2 cycles; the sweeps' instruction mix is mostly 4-cycle kinds "
                                   "(v_max_i32, VOP3, DPP, v_max_f64: tools/valu_peak.hip, profiles/r01g_valu_peak.txt), whose sustained "
                                   "rate is the realistic ceiling; instruction count from profiles/sw_valu.json (PMC, separate run)"}
            except Exception:
                sw_valu = None
        db_desc = "%d-genome (%d species x %d strains x %.1f Mb%s = %.2f Gb) synthetic %s db" % (
            n_entries, args.species, args.strains, args.genome_len / 1e6, " + %d viral genomes of 5-200 kb" % n_viral if n_viral else "",
            float(offs[-1]) / 1e9, ("bacterial + viral" if n_viral else "bacterial") +
            (" REPEAT-RICH (rRNA-like 1.5 kb segment x5 per genome, 1.3 kb insertion element x2 in every third species)" if args.repeats else ""))
        roofline = {
            "bound": "hbm", "kernel": "k_scatter<4> (the scatter launch of one radix pass of the read k-mer sort; the sort only sees "
                                      "the k-mers the genome filter lets through; since round 3 a launch also stores the next pass's "
                                      "digit byte per record, which made the launch 9 % slower and the sort phase 24 % faster)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "launch_ms": round(launch_ms, 4), "bytes_per_launch": int(per_launch_bytes),
            "sort_phase": {"passes": passes, "bytes": int(sort_bytes), "ms": round(tm["ms_sort"], 3),
                           "frac": round(sort_bytes / (tm["ms_sort"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tm["ms_sort"] > 0 else 0.0,
                           # beside the formula value: the HBM bytes the phase's dispatches really moved, from the
                           # FETCH_SIZE / WRITE_SIZE counters of a separate profiled run of this workload, over the live time
                           "pmc_bytes": sort_pmc,
                           "pmc_frac": round(sort_pmc / (tm["ms_sort"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if sort_pmc and tm["ms_sort"] > 0 else None},
            # context, not the roofline: what the best hand-written streaming copy of the same bytes
            # sustained on a bench box (tools/copy_peak.hip, profiles/r01h_copy_peak.txt)
            "streaming_copy_ceiling": {"GB/s": 5590.0, "measured": "profiles/r01h_copy_peak.txt"},
        }
        # the sort north_star calls "giant": the genome k-mer records, sorted ONCE per index here (the reference re-extracts and
        # re-sorts them with every batch, src/SLAM.h:64-65).  SURVEY 8d's formula with the passes executed; device time from
        # HIP events inside kslam_set_index (kslam_index_build_stats); PMC bytes of the same phase from a separate profiled run
        ist = ctx.index_build_stats()
        isort_bytes = ist["n_genome_kmers"] * 16 * (2 * ist["sort_passes"] + 1)
        isort_pmc = None
        if os.path.exists(tpath) and config in (1, 3) and args.species == 250:
            try:
                isort_pmc = (json.load(open(tpath)).get("index_sort") or {}).get("bytes")
            except Exception:
                isort_pmc = None
        roofline["index_sort"] = {
            "what": "one-time radix sort of the genome k-mer records at kslam_set_index (k_tile_hist_setup<4> + k_scatter_setup<4> per pass)",
            "records": ist["n_genome_kmers"], "passes": ist["sort_passes"], "bytes": int(isort_bytes), "ms": round(ist["ms_sort"], 3),
            "achieved": round(isort_bytes / (ist["ms_sort"] * 1e-3) / 1e9, 1) if ist["ms_sort"] > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(isort_bytes / (ist["ms_sort"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ist["ms_sort"] > 0 else 0.0,
            "pmc_bytes": isort_pmc,
            "pmc_frac": round(isort_pmc / (ist["ms_sort"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if isort_pmc and ist["ms_sort"] > 0 else None,
            "index_build_ms": {k: round(ist[k], 2) for k in ("ms_encode_extract", "ms_sort", "ms_tables", "ms_total")}}
        hot = {
            "reads_per_s": round(total_reads / elapsed, 1), "ms_per_step": round(elapsed / Ksteps * 1e3, 3),
            "what": ("ONE batch of %d pairs per step, read pairs split over %d GPU(s), timed until rank 0 holds the merged overlap records"
                     % (args.total_pairs, world)) if strong else
                    "alignToDatabase incl. CIGAR on a batch that is resident in HBM; results stay on the device",
            "phases_ms": {k: round(tm[k], 3) for k in ("ms_extract", "ms_sort", "ms_join", "ms_sw", "ms_cigar", "ms_total")},
            "counts": {"read_kmers": int(tm["n_read_kmers"]), "read_kmers_kept_by_filter": int(tm["n_kmers_kept"]),
                       "genome_kmers": int(tm["n_genome_kmers"]), "overlaps_raw": int(tm["n_overlaps_raw"]),
                       "candidates": int(tm["n_overlaps"]), "cigar_ops": int(n_cig), "chunks": int(tm["n_chunks"])},
            "sw_gcups": round(tm["sw_cells"] / ((tm["ms_sw"]) * 1e-3) / 1e9, 1) if tm["ms_sw"] > 0 else 0.0,
            "verified": verified,
        }
        out = {
            "metric": "paired %dbp reads/sec classified (bit-exact SAM)" % read_len,
            "value": None, "unit": "reads/s",
            "n_gpus": world, "steps": Ksteps, "warmup": args.warmup, "ms_per_step": None,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u64 k-mers / i32 DP", "data": "synthetic",
            "config": {
                "workload": ("BASELINE configs[3]: ONE batch of %d x 2 x %d bp reads per step, read pairs split over %d GPU(s), vs %s"
                             % (args.total_pairs, read_len, world, db_desc)) if strong else
                            ("BASELINE configs[%d]: %d x 2 x %d bp reads per batch%s vs %s, %s" % (
                                config, pairs, read_len, " per GPU" if world > 1 else "", db_desc,
                                "pseudo-assembly on (the reference's default)" if pseudo else "--no-pseudo-assembly")),
                "pairs_per_batch": n_batch_pairs, "pairs_per_gpu": n_reads // 2, "db_bases": int(offs[-1]), "db_entries": n_entries,
                "parallelism": "read pairs sharded x%d, genome k-mer list replicated, gather to rank 0" % world,
            },
            "value_definition": None,
            "hot_path": hot, "roofline": roofline, "roofline_valu": sw_valu,
            "setup_s": {"generate": round(t_gen, 2), "index_build": round(t_index, 2)},
        }
        if use_dist:
            out["rccl"] = {"data_path": "kslam_comm (include/kslam_comm.h: RCCL opened by the library, behind the C ABI)" if comm == "closed"
                           else "torch.distributed (k-slam_amd/dist.py)", "data_path_why": comm_why, "kslam_comm": comm_facts,
                           "backend": dist.get_backend(), "world": world, "ranks_seen": ranks_seen,
                           "bytes_gathered_per_step": int(merged["ov"].numel() + merged["cg"].numel()) if "ov" in merged else 0,
                           "launched_by": os.environ.get("KSLAM_BENCH_LAUNCHED_BY", "external launcher (torch.distributed.run)"),
                           "shared_gpu": bool(share)}
            out["per_rank_align_ms"] = per_rank_align
        if use_dist and strong:
            out["strong_step_split_ms"] = {
                "max_over_ranks": {k: round(v / Ksteps * 1e3, 3) for k, v in split_max.items()},
                "rank0": {k: round(v / Ksteps * 1e3, 3) for k, v in split.items()}}
        if strong and classified is not None:
            sh = classified.pop("sharded")
            out["value"] = round(total_reads / sh["elapsed"], 1)
            out["ms_per_step"] = round(sh["elapsed"] / Ksteps * 1e3, 3)
            del sh["elapsed"]
            classified["reads_per_s"] = round(total_reads / classified["elapsed"], 1)
            classified["ms_per_step"] = round(classified["elapsed"] / Ksteps * 1e3, 3)
            del classified["elapsed"]
            out["classified_sharded"] = sh
            out["classified_rank0_tail"] = classified
            out["verified_classified"] = bool(sh["part_files_in_rank_order_equal_rank0_files"])
            out["value_definition"] = (
                "K steps of ONE batch of --total-pairs read pairs, sharded over the ranks and CLASSIFIED where they are: every "
                "rank aligns its read pairs (resident in its HBM), pairs them, all-gathers the insert sizes (the limit is a "
                "statistic of the whole batch), sends the 16-byte heads of its alignment-pair records to the ranks that own their "
                "entries (pseudo-assembly is per entry over all read pairs: entry e on rank e mod N) and gets the scores back "
                "over RCCL, screens, and writes the SAM text and the per-read LCA of ITS read pairs into its part files (the "
                "parts in rank order are the files: verified_classified); max over ranks.  classified_rank0_tail: the same "
                "result with the overlap records gathered to rank 0 and the whole tail there (the Amdahl form); "
                "hot_path.reads_per_s: clock stopped when rank 0 holds the merged overlap records")
            # the curve's origin, same workload, same run (world 1: this line IS that point)
            if world > 1:
                out["n1_same_workload"] = n1_same
                if n1_same and "value" in n1_same:
                    out["speedup_vs_n1_same_workload"] = round(out["value"] / n1_same["value"], 3)
                    hot["speedup_vs_n1_same_workload"] = round(hot["reads_per_s"] / n1_same["hot_path_reads_per_s"], 3)
                else:
                    out["speedup_vs_n1_same_workload"] = None
            else:
                out["n1_same_workload"] = {"what": "this line is the N = 1 point of the strong workload", "value": out["value"], "unit": "reads/s",
                                           "ms_per_step": out["ms_per_step"], "hot_path_reads_per_s": hot["reads_per_s"]}
                out["speedup_vs_n1_same_workload"] = 1.0
            out["scaling_curve_origin"] = ("n1_same_workload.value: the same ONE-batch-of-%d-pairs step on one GPU, measured in this run by rank 0 "
                                           "alone after the N-rank clocks; speedup_vs_n1_same_workload = value / that" % args.total_pairs)
        else:
            out["value_definition"] = "see e2e" if not args.no_e2e and not strong else "hot path only (--no-e2e): resident-input alignToDatabase"
        torch.cuda.empty_cache()       # what torch's allocator cached while generating the inputs goes back to the device
        if world == 1 and not args.no_cpu_baseline:
            # the apples-to-apples baseline is the WHOLE workload on the batch the hot path timed: ~30 s on 16 CPUs for
            # configs[1] (profiles/r03i_bench_cpu_full.json).  Taken by default when the box can afford it, else the sample.
            import oracle as O
            cpus = O.usable_cpus()
            try:
                free_gb = int([ln for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")][0].split()[1]) / 1e6
            except Exception:
                free_gb = 0.0
            mode, why = args.cpu_baseline, None
            if args.cpu_full:
                mode = "full"
            if mode == "auto":
                if config == 1 and not strong and cpus >= 16 and free_gb >= 60 and pairs <= 1_000_000:
                    mode = "full"
                else:
                    mode = "sample"
                    why = ("auto: the whole workload needs >= 16 usable CPUs (has %d), >= 60 GB of free host memory (has %.0f) and "
                           "the configs[1] batch (this is config %s, %d pairs); --cpu-baseline full forces it" % (cpus, free_gb, config, pairs))
            if mode == "full":
                out["cpu_baseline"] = cpu_baseline(K, db, offs, 2, n_entries, pairs, read_len, local_rank, full=True,
                                                   same_reads=None if strong else reads)
                cands = out.get("hot_path", {}).get("counts", {}).get("candidates")
                out["cpu_baseline"]["n_alignments_equals_hot_path_candidates"] = (cands == out["cpu_baseline"]["n_alignments"])
            else:
                out["cpu_baseline"] = cpu_baseline(K, db, offs, 77, args.cpu_genomes, args.cpu_pairs, read_len, local_rank, why=why)
        if world == 1 and not strong and "abi_path" in legs and pairs <= 2_000_000:
            try:
                out["abi_path"] = abi_path(K, ctx, reads, read_len, max(Ksteps, 12))
            except Exception as e:   # extra evidence only: never lose the bench line over it
                out["abi_path"] = {"error": repr(e)}
        if world == 1 and not strong and not args.no_e2e and not args.no_cigar:
            # ---- `value`: FASTQ text -> SAM file + _PerRead file, K steps of the reference's batch loop ----
            batch_bytes = 2 * pairs * (2 + W.ID_DIGITS + 3 + 2 * read_len + 4)
            F = max(1, min(Ksteps, int(26e9 // batch_bytes)))     # distinct batches in the text (page-locked host memory)
            batches = [reads]
            for b in range(1, F):
                if pairs > 2_000_000:
                    r, _ = W.make_batch_in_pieces(dev, gen, db, offs, pairs, read_len, seed_base=2 + 17 * b, by_length=by_length, with_truth=False)
                else:
                    gen.manual_seed(2 + 17 * b)
                    r = make_reads(dev, gen, db, offs, pairs, read_len=read_len, by_length=by_length)
                batches.append(r)
            files = FastqFiles(K, dev, batches, read_len)
            del batches
            torch.cuda.empty_cache()
            taxdb = X.TaxDB(tax_text)
            e2e = e2e_leg(K, ctx, files, pairs, index_view, taxdb, Ksteps, args.warmup, pseudo, out_dir=args.out_dir)
            out["e2e"] = e2e
            if sink_probe:
                out["sink_probe"] = {"GB_per_s_of_a_new_256MB_file": sink_probe, "chosen": args.out_dir}
            out["value"], out["ms_per_step"] = e2e["reads_per_s"], e2e["ms_per_step"]
            out["value_definition"] = (
                "2 x pairs x K / wall clock of K steps of the reference's batch loop (src/SLAM.h:193-249), pipeline fill and drain "
                "included, median of %d repetitions: FASTQ text in page-locked HOST memory when the clock starts (every PCIe byte "
                "inside), SAM text and <out>_PerRead written to new files in %s when it stops (write() into the page cache, no fsync -- "
                "the reference's ofstream does none; --out-dir auto takes the faster of /dev/shm and the temp directory, see "
                "sink_probe; e2e_other_sink is the same leg on the other one), per-read LCA inside.  The rate with "
                "the batch resident in HBM and the results left on the device (alignToDatabase only) is hot_path.reads_per_s"
                % (3, args.out_dir))
            if sink_probe and len(sink_probe) > 1 and "other_sink" in legs:
                other = [d for d in sink_probe if d != args.out_dir][0]
                try:
                    d = e2e_leg(K, ctx, files, pairs, index_view, taxdb, Ksteps, args.warmup, pseudo, reps=1, tag="other", out_dir=other)
                    out["e2e_other_sink"] = {k: d[k] for k in ("reads_per_s", "ms_per_step", "host_ms_per_batch", "sink")}
                except Exception as e:
                    out["e2e_other_sink"] = {"error": repr(e)}
            if "null_sink" in legs:
                try:                  # what the sink costs: the same leg with the SAM text written to /dev/null
                    d = e2e_leg(K, ctx, files, pairs, index_view, taxdb, Ksteps, args.warmup, pseudo, reps=1, tag="null", out_dir="/dev/null")
                    out["e2e_sam_to_dev_null"] = {k: d[k] for k in ("reads_per_s", "ms_per_step", "host_ms_per_batch")}
                except Exception as e:
                    out["e2e_sam_to_dev_null"] = {"error": repr(e)}
            if config == 1 and "pseudo" in legs:       # the same with the reference's default (pseudo-assembly on)
                try:
                    out["e2e_with_pseudo_assembly"] = e2e_leg(K, ctx, files, pairs, index_view, taxdb, Ksteps, args.warmup, True, tag="pa", out_dir=args.out_dir)
                except Exception as e:
                    out["e2e_with_pseudo_assembly"] = {"error": repr(e)}
            taxdb.close()
            files.close()
        if out["value"] is None:      # --no-e2e / --no-cigar / weak multi-GPU: the hot path is all that was timed
            out["value"], out["ms_per_step"] = hot["reads_per_s"], hot["ms_per_step"]
            out["value_definition"] = "hot path only: alignToDatabase on the resident batch" + (", results gathered to rank 0" if use_dist else "")
        # ---- the origin of the strong-scaling curve: the --gpus N > 1 lines run ONE batch of --total-pairs pairs per step, sharded
        # (configs[3]); this line's `value` is configs[1] (1 M-pair batches, FASTQ text in, files out).  The same strong step runs
        # here on this one GPU, in this process (no child, nothing exec'ed), with this context's index.  Should the leg take the
        # process down, the line as it stands is written by the handlers installed in emit_on_exit().
        if world == 1 and not strong and config == 1 and args.strong_n1 == "auto" and not args.no_e2e and not args.no_cigar:
            n1_pairs = args.total_pairs if pairs == 1_000_000 else max(PIECES, min(args.total_pairs, 8 * pairs) // PIECES * PIECES)
            out["scaling_curve_origin"] = ("strong_n1.value: the workload of the --gpus N > 1 lines (ONE batch of %d pairs per step) on this one "
                                           "GPU; `value` itself is configs[1] and is NOT the N = 1 point of their curve.  Every N > 1 line also "
                                           "measures that point itself (n1_same_workload)" % n1_pairs)
            pending_line[0] = out
            t0 = time.time()
            try:
                del reads
                torch.cuda.empty_cache()
                whole, _ = W.make_batch_in_pieces(dev, gen, db, offs, n1_pairs, read_len, pieces=PIECES, by_length=by_length, with_truth=False)
                torch.cuda.synchronize()
                out["strong_n1"] = solo_strong(K, ctx, whole, read_len, n1_pairs, index_view, tax_text, max(3, min(Ksteps, 10)), min(args.warmup, 2),
                                               args.out_dir, "n1")
                out["strong_n1"]["seconds"] = round(time.time() - t0, 1)
                del whole
            except Exception as e:   # extra evidence only: never lose the bench line over it
                out["strong_n1"] = {"error": repr(e)}
        elif world == 1 and not strong:
            out["scaling_curve_origin"] = None
        emit(out)
    if ctx is not None:
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
        import traceback
        traceback.print_exc()
        try:
            sys.stdout.flush()
            sys.stderr.flush()
        except Exception:
            pass
        os._exit(1)     # a rank that failed must not sit in a collective's destructor while the others wait
