import sys, os, json, numpy as np, torch, importlib
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
import bench as B
K = entry.load_package()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
db, offs = B.make_database(dev, gen, 250, 5, 4_000_000)
gen.manual_seed(2)
reads = B.make_reads(dev, gen, db, offs, 1_000_000, read_len=150)
ctx = K.Context(report_cigar=True, device=0)
ctx.set_index_device(len(offs) - 1, db.data_ptr(), offs)
n_reads = reads.shape[0]
roffs = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(150))
ctx.load_reads_device(n_reads, reads.data_ptr(), roffs)
n_out, n_cig = ctx.align_resident()
ov, cg = ctx.fetch_results(n_out, n_cig)
sc = ov["score"].astype(np.int64)
print("results", len(ov))
h = np.bincount(np.minimum(sc // 10, 30))
print("score/10 histogram:", h.tolist())
span_r = ov["ref_end"] - ov["ref_begin"]; span_q = ov["query_end"] - ov["query_begin"]
print("ungapped-looking (equal spans):", float((span_r == span_q).mean()))
print("cigar_len hist:", np.bincount(np.minimum(ov["cigar_len"], 10)).tolist())
print(json.dumps(ctx.timings()))
