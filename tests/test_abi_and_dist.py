"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/kslam.h
declares (no compute without a GPU), it fails loudly without a device, and the read-pair
sharding + gather of the multi-GPU path is correct (gloo, world_size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="kslam.h"):
    h = open(os.path.join(ROOT, "include", header)).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(kslam_[a-z_0-9]+)\s*\(", h)))


def test_library_exports_every_declared_symbol(kslam):
    import ctypes
    assert os.path.exists(kslam.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(kslam.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(L, name), "missing export " + name
    assert sorted(kslam.EXPORTS) == declared
    assert L.kslam_abi_version() == 10


def test_library_exports_every_tail_symbol(kslam):
    """include/kslam_tail.h (host tail) lives in the same library."""
    import ctypes
    import importlib
    T = importlib.import_module("kslam_amd.tail")
    L = ctypes.CDLL(kslam.LIB_PATH)
    declared = _declared_symbols("kslam_tail.h")
    assert len(declared) == 16
    for name in declared:
        assert hasattr(L, name), "missing export " + name
    assert sorted(T.EXPORTS) == declared
    # struct sizes the header implies (natural alignment, no packing)
    assert ctypes.sizeof(T.TailParams) == 40 and ctypes.sizeof(T.ReadsView) == 56
    assert ctypes.sizeof(T.IndexView) == 128 and ctypes.sizeof(T.TailStats) == 104
    assert T.PAIRED_OVERLAP_DT.itemsize == 32 and T.READ_PAIR_DT.itemsize == 24


@pytest.mark.parametrize("header,module,count", [("kslam_fastq.h", "fastq", 6), ("kslam_taxonomy.h", "taxonomy", 17), ("kslam_samtext.h", "samtext", 4),
                                                 ("kslam_db.h", "db", 9), ("kslam_stream.h", "stream", 1)])
def test_library_exports_every_host_stage_symbol(kslam, header, module, count):
    import ctypes
    import importlib
    M = importlib.import_module("kslam_amd." + module)
    L = ctypes.CDLL(kslam.LIB_PATH)
    declared = [d for d in _declared_symbols(header) if d != "kslam_write_fn"]
    assert len(declared) == count
    for name in declared:
        assert hasattr(L, name), "missing export " + name
    assert sorted(M.EXPORTS) == declared


def test_struct_layouts(kslam):
    assert kslam.KMER_DT.itemsize == 16          # sizeof(KMerAndData<uint64_t,32>), src/KMer.h:103-116
    assert kslam.OVERLAP_TEMP_DT.itemsize == 16  # OverlapTemp, src/Overlap.h:36-52
    assert kslam.OVERLAP_DT.itemsize == 48


def test_no_device_fails_loudly(kslam):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(kslam.KslamError) as e:
        kslam.Context()
    assert e.value.status == 2 and "no CPU path" in str(e.value)


def test_product_never_touches_the_oracle():
    """The product tree must not reference oracle/ in any form."""
    for dp, _, files in os.walk(os.path.join(ROOT, "k-slam_amd")):
        if "build" in dp:
            continue
        for f in files:
            if f.endswith((".hip", ".h", ".hpp", ".py", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), os.path.join(dp, f)


_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist, importlib
from conftest import load_kslam
K = load_kslam(); kd = importlib.import_module("kslam_amd.dist"); synth = importlib.import_module("kslam_amd.synth")
import oracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
genomes = synth.make_genomes(3, 2, 2, 15000, shared_segment=1000)
n_pairs = 101
reads, _ = synth.make_paired_reads(4, genomes, n_pairs, edge_frac=0.1)
reads, genomes = synth.to_bytes(reads), synth.to_bytes(genomes)
bounds = kd.shard_bounds(n_pairs, world)
lo, hi = bounds[rank]
al, cg, _ = O.align_to_database(kd.local_reads(reads, n_pairs, lo, hi), genomes)   # stands in for the HIP path
ov_t = torch.from_numpy(al.view(np.uint8).copy()); cg_t = torch.from_numpy(cg.view(np.uint8).copy())
handle = kd.start_gather(ov_t, cg_t)          # bench.py overlaps the next batch here
parts = kd.finish_gather(handle)
again = kd.gather_to_rank0(ov_t, cg_t)         # the one-call form gives the same
assert (again is None) == (parts is None)
if parts is not None:
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(parts, again))
if rank == 0:
    parts = [(p[0].numpy().view(K.OVERLAP_DT), p[1].numpy().view(np.uint32)) for p in parts]
    got, pool = kd.reassemble(parts, bounds, n_pairs, K.OVERLAP_DT)
    exp, epool, _ = O.align_to_database(reads, genomes)
    assert len(got) == len(exp) and len(got) > 100
    for f in ("read", "entry", "rel", "revcomp", "score", "ref_begin", "ref_end", "query_begin", "query_end", "cigar_len"):
        assert (got[f] == exp[f]).all(), f
    for i in range(len(got)):
        a = pool[int(got["cigar_off"][i]):int(got["cigar_off"][i]) + int(got["cigar_len"][i])]
        b = epool[int(exp["cigar_off"][i]):int(exp["cigar_off"][i]) + int(exp["cigar_len"][i])]
        assert (a == b).all()
    print("GATHER_OK", len(got))

# ---- the gather without a merge step (kslam_amd.dist.start_gather_sharded): the context's two device
# functions restated in numpy on host memory, the protocol itself (count exchange, bases, four pieces
# per rank landing in their final places) is the product's ----
import ctypes
class HostShard:
    def __init__(self, al, cg):
        self.al, self.cg = al, cg
    def shard_counts_device(self, n_loc):
        r1 = self.al["read"] < n_loc
        n1 = int(r1.sum())
        return len(self.al), n1, len(self.cg), int(self.al["cigar_len"][:n1].sum())
    def export_shard_device(self, n_loc, pair_lo, n_total, pb1, pb2, d_r1, d_r2, d_p1, d_p2):
        n, n1, c, c1 = self.shard_counts_device(n_loc)
        o = self.al.copy()
        has = o["cigar_len"] > 0
        o["read"][:n1] += np.uint32(pair_lo)
        o["read"][n1:] = o["read"][n1:] - np.uint32(n_loc) + np.uint32(n_total + pair_lo)
        o["cigar_off"][:n1] = np.where(has[:n1], o["cigar_off"][:n1] + np.uint64(pb1), 0)
        o["cigar_off"][n1:] = np.where(has[n1:], o["cigar_off"][n1:] - np.uint64(c1) + np.uint64(pb2), 0)
        for ptr, data in ((d_r1, o[:n1]), (d_r2, o[n1:]), (d_p1, self.cg[:c1]), (d_p2, self.cg[c1:])):
            b = data.tobytes()
            if b:
                ctypes.memmove(ptr, b, len(b))
h = kd.start_gather_sharded(HostShard(al.view(K.OVERLAP_DT), cg), hi - lo, lo, n_pairs, torch.device("cpu"))
res = kd.finish_gather(h)
if rank == 0:
    rows, pool = res
    exp, epool, _ = O.align_to_database(reads, genomes)
    assert rows.numpy().tobytes() == exp.view(K.OVERLAP_DT).tobytes() and pool.numpy().tobytes() == epool.tobytes()
    print("SHARDED_GATHER_OK")
dist.barrier(); dist.destroy_process_group()
'''


def test_sharded_gather_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GATHER_OK" in r.stdout and "SHARDED_GATHER_OK" in r.stdout


_ROUTED_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist, importlib
from conftest import load_kslam
K = load_kslam(); kd = importlib.import_module("kslam_amd.dist")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
HEAD = np.dtype([("score", "<u4"), ("entry", "<u4"), ("start", "<i4"), ("stop", "<i4")])
def records(r):          # rank r's alignment-pair heads of the batch, in read-pair order (every rank can make every rank's)
    g = np.random.default_rng(100 + r)
    n = [0, 57, 1, 301, 40][r %% 5]
    a = np.zeros(n, dtype=HEAD)
    a["score"] = g.integers(1, 600, n); a["entry"] = g.integers(0, 11, n); a["start"] = g.integers(0, 5000, n); a["stop"] = a["start"] + 150
    return a
def stage(heads):         # stands in for the device stage: per entry, an ORDER-DEPENDENT function of the heads in arrival order
    out = heads["score"].copy()
    for e in np.unique(heads["entry"]):
        idx = np.flatnonzero(heads["entry"] == e)
        acc = np.uint32(17)
        for k in idx:
            acc = np.uint32((int(acc) * 31 + int(heads["score"][k]) + int(heads["start"][k])) & 0xFFFFFFFF)
            out[k] = acc
    return out
class HostCtx:            # kslam_pseudo_route / _owned / _return restated on host memory; the PROTOCOL under test is dist.py's
    def __init__(self, recs): self.recs, self.keep = recs, []
    def pseudo_route(self, world):
        dest = self.recs["entry"] %% world
        self.perm = np.argsort(dest, kind="stable")
        self.sent = np.ascontiguousarray(self.recs[self.perm]); self.keep.append(self.sent)
        return (self.sent.ctypes.data if len(self.sent) else 0), [int((dest == d).sum()) for d in range(world)]
    def pseudo_owned(self, ptr, n):
        import ctypes
        heads = np.frombuffer((ctypes.c_char * (16 * n)).from_address(ptr), dtype=HEAD).copy() if n else np.zeros(0, dtype=HEAD)
        assert n == 0 or (heads["entry"] %% world == rank).all()
        if os.environ.get("DECLINE_ON") == str(rank):
            raise K.KslamError(4, "declined (test)")
        self.scores = np.ascontiguousarray(stage(heads)); self.keep.append(self.scores)
        return self.scores.ctypes.data if n else 0
    def pseudo_return(self, ptr, n, fraction):
        import ctypes
        sc = np.frombuffer((ctypes.c_char * (4 * n)).from_address(ptr), dtype="<u4").copy() if n else np.zeros(0, dtype="<u4")
        self.result = self.recs["score"].copy(); self.result[self.perm] = sc
        return {"stages_done": 7}
mine = records(rank)
ctx = HostCtx(mine)
try:
    stats, moved = kd.routed_pseudo_assembly(ctx, torch.device("cpu"))
    whole = np.concatenate([records(r) for r in range(world)])          # the batch in rank order = read-pair order
    exp = stage(whole)
    at = sum(len(records(r)) for r in range(rank))
    assert (ctx.result == exp[at:at + len(mine)]).all(), "rank %%d: scores differ from the whole batch's" %% rank
    assert moved == 16 * int((whole["entry"] %% world == rank).sum()) + 4 * len(mine)
    print("ROUTED_OK", rank, len(mine))
except K.KslamError as e:
    print("ROUTED_DECLINED", rank, e.status)
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,decline", [(2, None), (3, None), (5, None), (3, 1)])
def test_routed_pseudo_assembly_protocol_gloo(tmp_path, world, decline):
    """k-slam_amd/dist.py routed_pseudo_assembly (the all-to-all of heads to the entries' owners, the scores back, the status
    word) between `world` processes over gloo, with the three device functions restated in numpy and the stage replaced by an
    order-dependent function per entry: every rank must end up with the scores ONE process computes for the whole batch --
    i.e. the pieces arrive in source-rank order and go back to where they came from.  With one rank declining, every rank
    raises and none hangs."""
    script = tmp_path / "routed.py"
    script.write_text(_ROUTED_WORKER % {"root": ROOT})
    port = str(29560 + world + (10 if decline is not None else 0))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    if decline is not None:
        env["DECLINE_ON"] = str(decline)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    if decline is None:
        assert r.stdout.count("ROUTED_OK") == world
    else:
        assert r.stdout.count("ROUTED_DECLINED") == world and "ROUTED_OK" not in r.stdout


@pytest.mark.parametrize("header", ["kslam.h", "kslam_tail.h", "kslam_fastq.h", "kslam_taxonomy.h", "kslam_db.h", "kslam_stream.h", "kslam_comm.h", "kslam_samtext.h"])
def test_headers_are_plain_c(header, tmp_path):
    """The boundary is a C ABI: every header must compile on its own as C99 (pedantic) and as C++11."""
    import shutil
    import subprocess
    src = tmp_path / "h.c"
    src.write_text('#include "%s"\nint main(void) { return 0; }\n' % header)
    inc = os.path.join(ROOT, "include")
    for cc, std, lang in (("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "c++")):
        if not shutil.which(cc):
            pytest.skip(cc + " not available")
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, "-x", lang, str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_library_exports_every_comm_symbol_and_the_gather_plan(kslam):
    """include/kslam_comm.h (RCCL behind the C ABI): exported by the same library, which does NOT link librccl
    (dlopen on first use); the placement arithmetic of the gather equals k-slam_amd/dist.py's (the protocol the gloo
    world-2 test above runs)."""
    import ctypes
    import importlib
    import numpy as np
    Cm = importlib.import_module("kslam_amd.comm")
    L = ctypes.CDLL(kslam.LIB_PATH)
    declared = _declared_symbols("kslam_comm.h")
    assert sorted(Cm.EXPORTS) == declared and len(declared) == 12
    for name in declared:
        assert hasattr(L, name), "missing export " + name
    needed = subprocess.run(["readelf", "-d", kslam.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" not in needed
    rng = np.random.default_rng(8)
    for world in (1, 2, 3, 4, 8):
        cnt = []
        for _ in range(world):
            n, c = int(rng.integers(0, 1000)), int(rng.integers(0, 3000))
            cnt.append((n, int(rng.integers(0, n + 1)), c, int(rng.integers(0, c + 1))))
        row1, row2, op1, op2, tot = Cm.gather_plan(cnt)
        # k-slam_amd/dist.py start_gather_sharded
        rows_r1 = sum(c[1] for c in cnt)
        ops_r1 = sum(c[3] for c in cnt)
        assert row1 == [sum(c[1] for c in cnt[:r]) for r in range(world)]
        assert row2 == [rows_r1 + sum(c[0] - c[1] for c in cnt[:r]) for r in range(world)]
        assert op1 == [sum(c[3] for c in cnt[:r]) for r in range(world)]
        assert op2 == [ops_r1 + sum(c[2] - c[3] for c in cnt[:r]) for r in range(world)]
        assert tot == (sum(c[0] for c in cnt), sum(c[2] for c in cnt))
