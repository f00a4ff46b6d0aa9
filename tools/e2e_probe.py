"""Where the host stage of the batch loop spends its time: the e2e leg of bench.py with the SAM sink switched
(none / /dev/shm file / /dev/null / /tmp file) and the taxonomy stage on or off.  python tools/e2e_probe.py"""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as entry
K = entry.load_package()
W = importlib.import_module("kslam_amd.workload"); T = importlib.import_module("kslam_amd.tail")
X = importlib.import_module("kslam_amd.taxonomy"); S = importlib.import_module("kslam_amd.stream")
import bench
dev = torch.device("cuda", 0); gen = torch.Generator(device=dev); gen.manual_seed(1)
db, offs = W.make_database(dev, gen, 250, 5, 4_000_000)
ctx = K.Context(); ctx.set_index_device(len(offs) - 1, db.data_ptr(), offs)
batches = []
for b in range(4):
    gen.manual_seed(2 + 17 * b); batches.append(W.make_reads(dev, gen, db, offs, 1_000_000))
files = bench.FastqFiles(K, dev, batches, 150); del batches
tax_text, entry_tax = W.taxonomy(250, 5, 0)
I = T.IndexArrays(np.zeros(1, dtype=np.uint8), offs, taxonomy_ids=entry_tax)
taxdb = X.TaxDB(tax_text)
P = T.TailParams.default(pseudo_assembly=False)
wins = list(S.cut_batches(files.h[0].ptr, files.len, files.h[1].ptr, files.len, 1_000_000))
def run(sink, tax, n=10):
    fd = -1
    if sink: fd = os.open(sink, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o600)
    t0 = time.perf_counter()
    rep = X.Report() if tax == 2 else None
    res = S.classify_stream(ctx, I, files.h[0].ptr, files.len, files.h[1].ptr, files.len, 1_000_000, P, taxdb=taxdb if tax else None,
                            report=rep, sam_fd=fd, windows=[wins[i % 4] for i in range(n)])
    if rep is not None: rep.close()
    dt = time.perf_counter() - t0
    if fd >= 0: os.close(fd)
    if sink and os.path.isfile(sink) and not sink.startswith("/dev/null"): os.unlink(sink)
    b = res["batches"]
    return {"sink": sink, "tax": tax, "ms_per_batch": round(dt / n * 1e3, 1), "ms_sam": round(sum(x["ms_sam"] for x in b) / n, 1),
            "ms_classify": round(sum(x.get("ms_classify", 0) for x in b) / n, 1), "ms_report": round(sum(x.get("ms_report", 0) for x in b) / n, 1), "wait_gpu_s": res["s_waiting_for_gpu"], "wait_host_s": res["s_waiting_for_host_stage"]}
run(None, False, 4)
for sink, tax in ((None, 0), ("/dev/null", 0), ("/dev/shm/kslam_probe.sam", 0), (None, 1), (None, 2), ("/dev/shm/kslam_probe.sam", 2)):
    print(json.dumps(run(sink, tax)), flush=True)
