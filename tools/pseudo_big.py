"""Time the device tail front (pairing, screens, pseudo-assembly) on big batches, where an entry holds more
alignment pairs than fit LDS:  python tools/pseudo_big.py [pairs ...]   (bench database, configs[1] shape)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

torch.cuda.init()
K = entry.load_package()
import importlib  # noqa: E402
W = importlib.import_module("kslam_amd.workload")
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev)
gen.manual_seed(1)
db, offs = W.make_database(dev, gen, 250, 5, 4_000_000)
ctx = K.Context(report_cigar=True, device=0)
ctx.set_index_device(len(offs) - 1, db.data_ptr(), offs)
for pairs in [int(x) for x in sys.argv[1:]] or [1_000_000, 4_000_000, 10_000_000]:
    gen.manual_seed(7)
    reads = W.make_reads(dev, gen, db, offs, pairs)
    n = reads.shape[0]
    ctx.load_reads_device(n, reads.data_ptr(), np.arange(n + 1, dtype=np.uint64) * np.uint64(150))
    n_out, n_cig = ctx.align_resident()
    for stages in (3, 7, 7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = ctx.pair_screen(paired=True, stages=stages)
        dt = time.perf_counter() - t0
        print("pairs %9d overlaps %9d stages %d: %.2f ms, alignment pairs %d, stages_done %d" % (
            pairs, n_out, stages, dt * 1e3, st["n_pairs"], st["stages_done"]), flush=True)
    del reads
    torch.cuda.empty_cache()
