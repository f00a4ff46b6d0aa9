"""Soak of the GPU text stage (csrc/samtext.hip) against the host formatter, random data sets and flags until the time is up:
    python tools/soak_samtext.py [seconds] [first_seed]
reads + qualities -> alignment, device pairing / screens [/ pseudo-assembly], per-row walk -> SAM records + per-read lines +
taxonomy ids written on the GPU   ==   kslam_tail_finish_write_rows + kslam_tail_classify on the same rows, byte for byte."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
torch.cuda.init()
from conftest import load_kslam        # noqa: E402
import test_gpu_samtext as TS          # noqa: E402

K = load_kslam()
synth = importlib.import_module("kslam_amd.synth")
T = importlib.import_module("kslam_amd.tail")
X = importlib.import_module("kslam_amd.taxonomy")
ST = importlib.import_module("kslam_amd.samtext")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
t_end = time.time() + budget
rounds = total = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    n_pairs = int(rng.choice([400, 1200, 2500]))
    read_len = int(rng.choice([60, 100, 150, 250]))
    many = bool(rng.random() < 0.35)
    rb, gb, quals, ids, I, taxdb_text = TS._case(synth, T, seed, n_pairs, many_strains=many, read_len=read_len)
    kw = dict(paired=bool(rng.random() < 0.8), num_alignments=int(rng.choice([1, 2, 3, 10, 10, 40])), sam_xa=bool(rng.random() < 0.2),
              score_threshold=int(rng.choice([0, 0, 0, read_len, int(1.6 * read_len)])), report_cigar=bool(rng.random() < 0.9),
              pseudo=bool(rng.random() < 0.5))
    if not kw["paired"]:
        rb, quals, ids = rb[:n_pairs], quals[:n_pairs], ids[:n_pairs]
    got, exp, biggest = TS._device_and_host_text(K, T, X, ST, rb, gb, quals, ids, I, taxdb_text, **kw)
    ok = got[0] == exp[0] and got[1] == exp[1] and got[2].tolist() == exp[2].tolist() and got[4].tobytes() == exp[4].tobytes()
    if not ok:
        print("MISMATCH at seed %d: %r read_len %d pairs %d many %s" % (seed, kw, read_len, n_pairs, many))
        sys.exit(1)
    rounds += 1
    total += len(exp[0])
    seed += 1
print("soak_samtext: %d data sets, %.1f MB of SAM text identical (device text == host text), seeds %d..%d" %
      (rounds, total / 1e6, int(sys.argv[2]) if len(sys.argv) > 2 else 7000, seed - 1))
