"""Does re-aligning the SAME resident batch (what bench.py's timed loop does) flatter the number?  Align batch A
ten times, then alternate between two different batches A and B (each loaded right before its alignment), and
compare the device time of the alignment (kslam_timings.ms_total, HIP events around the hot path only)."""
import os
import sys
import importlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

torch.cuda.init()
K = entry.load_package()
W = importlib.import_module("kslam_amd.workload")
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev)
gen.manual_seed(1)
db, offs = W.make_database(dev, gen, 250, 5, 4_000_000)
ctx = K.Context(report_cigar=True, device=0)
ctx.set_index_device(len(offs) - 1, db.data_ptr(), offs)
batches = []
for seed in (2, 1002):
    gen.manual_seed(seed)
    batches.append(W.make_reads(dev, gen, db, offs, 1_000_000))
n = batches[0].shape[0]
roffs = np.arange(n + 1, dtype=np.uint64) * np.uint64(150)


def run(order):
    out = []
    for b in order:
        ctx.load_reads_device(n, batches[b].data_ptr(), roffs)
        ctx.align_resident()
        out.append(ctx.timings()["ms_total"])
    return out


run([0, 1, 0, 1])
same = run([0] * 10)
alt = run([0, 1] * 5)
print("same batch ten times : %.3f ms per alignment (min %.3f max %.3f)" % (sum(same) / 10, min(same), max(same)))
print("two batches alternated: %.3f ms per alignment (min %.3f max %.3f)" % (sum(alt) / 10, min(alt), max(alt)))
