"""Soak of the device tail front against the host tail (k-slam_amd/host/tail.cpp), random configurations until
the time is up:  python tools/soak_tail.py [seconds] [first_seed]
  * pairing / insert-size statistics / screens / pseudo-assembly / second screen on random overlap sets full of
    score and position ties (tests/test_tail.py's generator) -- read pairs and alignment pairs byte for byte;
  * the wavefront std::sort (csrc/wave_gnu_sort.h) against the real std::sort (tests/gnu_sort_check.cpp perm).
Prints one line per round and a summary; exits non-zero on the first difference."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
torch.cuda.init()
from conftest import load_kslam  # noqa: E402
import importlib  # noqa: E402

K = load_kslam()
T = importlib.import_module("kslam_amd.tail")
from test_tail import _fuzz_overlaps  # noqa: E402
from test_gpu_tail import _compacted  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
tmp = tempfile.mkdtemp()
exe = os.path.join(tmp, "gnu_sort_check")
subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "gnu_sort_check.cpp"), "-o", exe])
ctx = K.Context()
t_end = time.time() + budget
rounds = recs = sorted_keys = 0
seed = seed0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    paired = bool(rng.random() < 0.8)
    n_entries = int(rng.choice([1, 2, 5, 12, 40, 300, 3000]))
    per_read = float(rng.choice([1.0, 3.0, 8.0, 20.0]))
    n_units = int(rng.choice([300, 2000, 6000]))
    if rng.random() < 0.25:         # reads inside a repeat: hundreds of rows a mate over many entries (k_pair_big, k_screen_big)
        n_entries = int(rng.choice([12, 60, 300, 1500]))
        per_read = float(rng.choice([150.0, 400.0, 800.0]))
        n_units = int(rng.choice([30, 80]))
    thr = int(rng.choice([0, 0, 150, 185]))
    frac = float(rng.choice([0.95, 0.95, 0.8, 0.5, 1.0]))
    stages = int(rng.choice([7, 7, 7, 3, 5, 6, 4]))
    ov, n_reads = _fuzz_overlaps(K, rng, n_units, n_entries, per_read=per_read, paired=paired)
    if rng.random() < 0.3:          # degenerate and reversed spans
        z = rng.random(len(ov)) < 0.02
        ov["ref_end"][z] = ov["ref_begin"][z] - rng.integers(0, 6, int(z.sum()))
    if rng.random() < 0.3:          # a spike far away in the insert-size ladder
        far = rng.random(len(ov)) < 0.03
        for f in ("rel", "ref_begin", "ref_end"):
            ov[f][far] += 500000
        ov = ov[np.lexsort((ov["rel"], ov["entry"], ov["read"]))]
    reads = T.Reads([b"A" * 100] * n_reads)
    got = ctx.pair_screen_overlaps(ov, np.full(n_reads, 100, dtype=np.uint32), paired=paired, score_threshold=thr,
                                   score_fraction=frac, stages=stages)
    grp, gpr = ctx.take_pairs()
    done = got["stages_done"]
    P = T.TailParams.default(paired=paired, report_cigar=False, threads=4, score_threshold=thr, score_fraction=frac,
                             pseudo_assembly=bool(done & 4), stages=(done & 7) if done & 7 else 8)
    rp, pr, st = T.tail_pairs(P, reads, ov)
    crp, cpr = _compacted(grp, gpr)
    ok = crp.tobytes() == rp.tobytes() and cpr.tobytes() == pr.tobytes()
    ok = ok and (not (paired and stages & 1) or got["max_insert_size"] == st.max_insert_size)
    # the sort
    sizes = [int(x) for x in rng.integers(0, 5000, 40)] + [int(x) for x in rng.integers(4001, 60000, 3)]
    segs = []
    for i, n in enumerate(sizes):
        k = rng.integers(0, int(rng.choice([1, 2, 3, 7, 50, 1000, 1 << 30])), n).astype(np.int32)
        sh = int(rng.integers(0, 5))
        if sh == 1:
            k.sort()
        elif sh == 2:
            k = np.sort(k)[::-1].copy()
        elif sh == 3 and n > 2:
            k.sort()
            k[n // 2:] = k[n // 2:][::-1]
        segs.append(k)
    off = np.zeros(len(segs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in segs])
    keys = np.concatenate(segs).astype(np.int32)
    fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
    with open(fin, "wb") as fh:
        fh.write(np.uint64(len(segs)).tobytes() + off.tobytes() + keys.tobytes())
    subprocess.check_call([exe, "perm", fin, fout])
    ok_sort = np.array_equal(np.fromfile(fout, dtype=np.uint32), ctx.debug_wave_sort(keys, off))
    rounds += 1
    recs += len(pr)
    sorted_keys += len(keys)
    print("seed %d paired %d entries %4d per_read %4.1f units %4d thr %3d frac %.2f stages %d done %d: %7d rows -> %7d alignment pairs %s; sort %s"
          % (seed, paired, n_entries, per_read, n_units, thr, frac, stages, done, len(ov), len(pr), "ok" if ok else "DIFFERENT",
             "ok" if ok_sort else "DIFFERENT"), flush=True)
    if not (ok and ok_sort):
        sys.exit(1)
    seed += 1
print("SOAK_TAIL_OK rounds %d alignment pairs %d sorted keys %d seeds %d..%d" % (rounds, recs, sorted_keys, seed0, seed - 1))
