"""Helper of tests/test_gpu_multi.py (run as a subprocess, one GPU): include/kslam_comm.h at world size N.

N threads of this process are the N ranks: each owns a sibling context on device 0 and a kslam_comm; the RCCL entry
points are tests/fake_rccl (KSLAM_RCCL_LIB, set by the caller) because RCCL itself refuses two ranks on one device.  What
runs is the library's own communicator code -- count / status exchanges, export in batch terms, the group of sends and
receives into final places, the all-gather of the insert sizes, pseudo-assembly's two all-to-alls -- and everything is
compared with ONE context that aligned the whole batch: rank 0's gathered arrays byte for byte, the ranks' read pairs,
alignment pairs and SAM text in rank order.

usage: comm_world_n.py WORLD N_PAIRS PSEUDO(0/1) [decline | empty]
  decline: the caller set KSLAM_PSEUDO_CAP so low that the device stage declines on the ranks that own entries: EVERY
           rank must come back with KSLAM_ERR_UNSUPPORTED, none may hang.
  empty:   the read pairs of rank 1 come from nowhere: no rows to send, no alignment pairs to route -- zero-length pieces in
           every exchange.
Prints one JSON line."""
import importlib
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    world, n_pairs, pseudo = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
    decline = len(sys.argv) > 4 and sys.argv[4] == "decline"
    empty = len(sys.argv) > 4 and sys.argv[4] == "empty"
    K = entry.load_package()
    synth = importlib.import_module("kslam_amd.synth")
    Cm = importlib.import_module("kslam_amd.comm")
    kd = importlib.import_module("kslam_amd.dist")
    T = importlib.import_module("kslam_amd.tail")
    import numpy as np
    genomes = synth.make_genomes(811, 3, 3, 30000, strain_sub=0.02, strain_indel=0.001, shared_segment=2000)
    reads, _ = synth.make_paired_reads(812, genomes, n_pairs, sub_rate=0.015, indel_rate=0.004, edge_frac=0.05)
    rb, gb = synth.to_bytes(reads), synth.to_bytes(genomes)
    rng = np.random.default_rng(5)
    if empty:
        lo, hi = kd.shard_bounds(n_pairs, world)[1] if world < 3 else (n_pairs // 5, n_pairs // 5 + 1)
        for i in list(range(lo, hi)) + list(range(n_pairs + lo, n_pairs + hi)):
            rb[i] = bytes(synth.random_bases(rng, len(rb[i])))
    quals = [bytes(rng.integers(35, 74, len(b), dtype=np.uint8)) for b in rb]
    ids = [b"q%05d" % i for i in range(n_pairs)]
    I = T.Index(gb, taxonomy_ids=list(range(1, len(gb) + 1)))
    P_write = T.TailParams.default(pseudo_assembly=False)

    def sam_of(c, rd, q, names):
        c.row_details(of_pairs=True)
        ov, cg, rel1 = c.take_results()
        det, md, rel2 = c.take_row_details(len(ov), copy=False)
        rp, pr, rel3 = c.take_pairs(copy=False)
        out = []
        T.tail_finish_rows(P_write, T.Reads(rd, q, names), I, ov, cg, det, md, rp, pr, sink=out.append)
        n = (len(rp), len(pr))
        for r in (rel1, rel2, rel3):
            r()
        return b"".join(out), n

    whole = K.Context()
    whole.set_index(gb)
    whole.load_reads(rb)
    whole.load_qualities(quals)
    n_out, n_cig = whole.align_resident()
    exp_rows, exp_pool = whole.fetch_results(n_out, n_cig)
    want = whole.pair_screen(paired=True, stages=7 if pseudo else 3)
    exp_sam, exp_n = sam_of(whole, rb, quals, ids + ids)

    uid = Cm.unique_id()
    bounds = kd.shard_bounds(n_pairs, world)
    if world >= 3:                       # uneven shards, one of a single read pair
        cut = [0, n_pairs // 5, n_pairs // 5 + 1] + [n_pairs // 5 + 1 + (k * (n_pairs - n_pairs // 5 - 1)) // (world - 2) for k in range(1, world - 1)]
        bounds = list(zip(cut[:-1], cut[1:]))
        assert len(bounds) == world and bounds[-1][1] == n_pairs
    res = [None] * world
    errs = [None] * world

    def rank_main(r):
        try:
            lo, hi = bounds[r]
            c = whole.sibling()
            loc, q = kd.local_reads(rb, n_pairs, lo, hi), kd.local_reads(quals, n_pairs, lo, hi)
            c.load_reads(loc)
            c.load_qualities(q)
            c.align_resident()
            comm = Cm.Comm(c, uid, r, world)
            out = {"info": comm.info()}
            comm.gather_begin(hi - lo, lo, n_pairs)
            d_rows, n_rows, d_pool, n_ops = comm.gather_end()
            if r == 0:
                probe = whole.sibling()
                probe.load_reads(rb)
                probe.adopt_results_device(d_rows, n_rows, d_pool, n_ops)
                g_rows, g_pool = probe.fetch_results(n_rows, n_ops)
                out["gather_identical"] = bool(g_rows.tobytes() == exp_rows.tobytes() and g_pool.tobytes() == exp_pool.tobytes())
                out["rows"] = n_rows
                probe.close()
            else:
                out["gather_identical"] = bool(d_rows is None and n_rows == 0)
            # a second gather in the same communicator (buffers reused, the other pair of arrays on rank 0)
            d2 = comm.gather_batch(hi - lo, lo, n_pairs)
            out["second_gather_rows"] = d2[1]
            try:
                st, moved = comm.sharded_tail(True, 0, 0.95, pseudo)
                out["stats"], out["moved"] = st, moved
                out["sam"], out["n"] = sam_of(c, loc, q, ids[lo:hi] + ids[lo:hi])
            except K.KslamError as e:
                out["tail_error"] = int(e.status)
            comm.close()
            c.close()
            res[r] = out
        except BaseException as e:       # noqa: BLE001 -- reported by the parent thread
            import traceback
            errs[r] = traceback.format_exc()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    hung = [r for r, t in enumerate(th) if t.is_alive()]
    line = {"world": world, "hung": hung, "errors": [e for e in errs if e]}
    if not hung and not line["errors"]:
        line["library"] = res[0]["info"]["library"]
        line["comm_counts"] = [x["info"]["comm_count"] for x in res]
        line["comm_ranks"] = [x["info"]["comm_rank"] for x in res]
        line["gather_identical"] = all(x["gather_identical"] for x in res)
        line["rows"] = res[0]["rows"]
        line["second_gather_rows"] = res[0]["second_gather_rows"]
        line["tail_errors"] = [x.get("tail_error") for x in res]
        if not any(line["tail_errors"]):
            line["sam_identical"] = bool(b"".join(x["sam"] for x in res) == exp_sam)
            line["counts_identical"] = bool((sum(x["n"][0] for x in res), sum(x["n"][1] for x in res)) == exp_n)
            line["limit_identical"] = bool(all(x["stats"]["max_insert_size"] == want["max_insert_size"] for x in res))
            line["stages_done"] = [x["stats"]["stages_done"] for x in res]
            line["moved"] = [x["moved"] for x in res]
            line["sam_bytes"] = len(exp_sam)
    whole.close()
    print(json.dumps(line), flush=True)
    if hung:
        os._exit(3)


if __name__ == "__main__":
    main()
