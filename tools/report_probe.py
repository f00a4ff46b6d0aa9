#!/usr/bin/env python3
"""Times the end-of-run reports (kslam_taxonomy_summary, kslam_taxreport_xml) on a synthetic run: N read pairs in
batches of 1 M, one alignment pair each, on the bench's taxonomy (250 species x 5 strains).  CPU only.
usage: tools/report_probe.py [n_pairs] [--no-genes]   (prints the seconds and a digest of the two texts)"""
import hashlib, importlib, sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import importlib.util, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("kslam_amd", os.path.join(root, "k-slam_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(root, "k-slam_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["kslam_amd"] = pkg; spec.loader.exec_module(pkg)
T = importlib.import_module("kslam_amd.tail"); X = importlib.import_module("kslam_amd.taxonomy")
W = importlib.import_module("kslam_amd.workload")

n_total = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 4_000_000
rng = np.random.default_rng(5)
tax_text, entry_tax = W.taxonomy(250, 5, 0)
db = X.TaxDB(tax_text)
n_entries = len(entry_tax)
no_genes = "--no-genes" in sys.argv
genes = [[] for e in range(n_entries)] if no_genes else [[(1000 * k, 1000 * k + 900, b"g%d" % k, b"WP_%d_%d" % (e // 5, k), b"product %d" % k) for k in range(3)] for e in range(n_entries)]
index = T.Index([b"A" * 4000] * n_entries, taxonomy_ids=list(entry_tax), genes=genes)
flat = [g for gl in genes for g in gl]
extras = None if no_genes else X.GeneExtras.from_lists([b"LT_%d" % i for i in range(len(flat))], [b"NC_%06d" % (i // 3) for i in range(len(flat))], list(range(len(flat))))
rep = X.Report()
all_tax = []
per = 1_000_000
t_add = 0.0
for b0 in range(0, n_total, per):
    n = min(per, n_total - b0)
    ids = [b"%08d" % (b0 + i) for i in range(n)]
    reads = T.Reads([b"A"] * (2 * n), ids=ids + ids)
    rp = np.zeros(n, dtype=T.READ_PAIR_DT)
    rp["r1_read"] = np.arange(n); rp["r2_read"] = np.arange(n) + n; rp["first"] = np.arange(n); rp["count"] = 1
    pr = np.zeros(n, dtype=T.PAIRED_OVERLAP_DT)
    e = rng.integers(0, n_entries, n); pr["entry"] = e
    s = rng.integers(0, 3500, n); pr["ref_start"] = s; pr["ref_end"] = s + 300
    sp = np.asarray(entry_tax, dtype=np.uint32)[e]
    tax = np.where(rng.random(n) < 0.02, 0, sp).astype(np.uint32)
    t0 = time.time(); rep.add_batch(reads, index, rp, pr, tax); t_add += time.time() - t0
    all_tax.append(tax)
all_tax = np.concatenate(all_tax)
t0 = time.time(); summ = db.summary(all_tax, n_total); t_sum = time.time() - t0
t0 = time.time(); xml = db.report_xml(rep, index, extras, n_total); t_xml = time.time() - t0
dig = hashlib.sha256(summ).hexdigest()[:16] + " " + hashlib.sha256(xml).hexdigest()[:16]
print(f"pairs {n_total}  add_batch {t_add:.2f} s  summary {t_sum:.2f} s ({len(summ)} B)  xml {t_xml:.2f} s ({len(xml)/1e6:.1f} MB)  digest {dig}")
