"""Load time of a synthetic <db>/database (include/kslam_db.h): python tools/db_bench.py [GB] [dir]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
K = entry.load_package()
D = importlib.import_module("kslam_amd.db")
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
d = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
path = os.path.join(d, "kslam_db_bench.database")
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
n = max(1, int(gb * 1e9 / 4e6))
entries = [{"bases": acgt[rng.integers(0, 4, 4_000_000, dtype=np.uint8)].tobytes(), "taxonomyID": i + 1, "genbankID": i,
            "locusTag": b"NC_%06d.1" % i,
            "genes": [{"geneName": b"g%d" % k, "proteinID": b"NP_%d.1" % k, "product": b"hypothetical protein", "start": 100 * k,
                       "stop": 100 * k + 90} for k in range(50)]} for i in range(n)]
t0 = time.time(); D.write(path, entries); t_write = time.time() - t0
size = os.path.getsize(path)
del entries
for threads in (1, 0):
    t0 = time.time(); db = D.Database.load(path, threads=threads); t = time.time() - t0
    print("load %.2f GB, %d entries, %d genes, threads=%s: %.2f s = %.2f GB/s" % (size / 1e9, db.n_entries, db.n_genes, threads or "all", t, size / 1e9 / t))
    db.close()
print("write: %.2f s" % t_write)
os.remove(path)
