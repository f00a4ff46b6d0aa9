import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
torch.cuda.init()
K = entry.load_package()
import importlib
W = importlib.import_module("kslam_amd.workload")
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
db, offs = W.make_database(dev, gen, 250, 5, 4_000_000)
ctx = K.Context(report_cigar=True, device=0)
ctx.set_index_device(len(offs) - 1, db.data_ptr(), offs)
pairs = 10_000_000
gen.manual_seed(7)
reads = W.make_reads(dev, gen, db, offs, pairs)
n = reads.shape[0]
ctx.load_reads_device(n, reads.data_ptr(), np.arange(n + 1, dtype=np.uint64) * np.uint64(150))
n_out, n_cig = ctx.align_resident()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cnt = ctx.shard_counts_device(pairs)
    t1 = time.perf_counter()
    rows = torch.empty(cnt[0] * 48, dtype=torch.uint8, device=dev)
    pool = torch.empty(cnt[2] * 4, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    rp, pp = rows.data_ptr(), pool.data_ptr()
    ctx.export_shard_device(pairs, 0, pairs, 0, cnt[3], rp, rp + 48 * cnt[1], pp, pp + 4 * cnt[3])
    t3 = time.perf_counter()
    print("counts %.2f ms, torch.empty %.2f ms, export %.2f ms (%d rows, %d ops)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, cnt[0], cnt[2]))
    del rows, pool
