"""Read-pair sharding across ranks and the end-of-batch gather.

The hot path shards by read pair: k-mer extraction, the join and SW are per read,
the dedupe is per (read, entry).  Each rank keeps the genome index replicated in
its own HBM and aligns pairs [lo, hi) of the batch; there is NO collective on
the data path.  The only exchange is the variable-length gather of the per-read
results to rank 0 at the end of the batch (RCCL point-to-point over xGMI on the
GPU box: each peer has its own direct link to GPU 0; gloo in the CPU tests).

Batch layout (reference src/FASTQsequence.h:111-123): reads = [R1 block | R2
block], mate of i is i + n.  A rank's local batch keeps that layout:
[R1[lo:hi] | R2[lo:hi]].
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_pairs, world):
    return [(r * n_pairs // world, (r + 1) * n_pairs // world) for r in range(world)]


def local_reads(reads, n_pairs, lo, hi):
    """reads: sequence of 2*n_pairs items in block layout -> the rank's local block layout."""
    return list(reads[lo:hi]) + list(reads[n_pairs + lo:n_pairs + hi])


def start_gather(ov_bytes, cig_bytes, group=None):
    """Begin the variable-length gather of (overlap records, cigar pool) byte tensors to rank 0.

    ov_bytes / cig_bytes: 1-D uint8 tensors (device tensors with the nccl backend); they must stay
    alive and unmodified until finish_gather.  Returns a handle for finish_gather.  Between the two
    calls the caller is free to run the next batch: the point-to-point transfers proceed on the
    communicator's own stream."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = ov_bytes.device
    sizes = torch.tensor([ov_bytes.numel(), cig_bytes.numel()], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    # one batched group of point-to-point operations (ncclGroupStart/End under RCCL), so the seven
    # peers stream to rank 0 concurrently, each over its own xGMI link
    ops, parts = [], None
    if rank == 0:
        parts = [(ov_bytes, cig_bytes)]
        for r in range(1, world):
            no, nc = int(all_sizes[r][0]), int(all_sizes[r][1])
            o = torch.empty(no, dtype=torch.uint8, device=dev)
            c = torch.empty(nc, dtype=torch.uint8, device=dev)
            if no:
                ops.append(dist.P2POp(dist.irecv, o, r, group))
            if nc:
                ops.append(dist.P2POp(dist.irecv, c, r, group))
            parts.append((o, c))
    else:
        if ov_bytes.numel():
            ops.append(dist.P2POp(dist.isend, ov_bytes, 0, group))
        if cig_bytes.numel():
            ops.append(dist.P2POp(dist.isend, cig_bytes, 0, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    return {"reqs": reqs, "parts": parts, "keep": (ov_bytes, cig_bytes)}


def start_gather_concat(ov_bytes, cig_bytes, group=None):
    """As start_gather, but rank 0 receives every rank's records straight into ONE row tensor and ONE
    pool tensor, shard after shard -- the layout kslam_merge_shards_device takes.  finish_gather then
    returns on rank 0 (rows_all, pool_all, [(row_bytes, pool_bytes) per rank]); elsewhere None."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = ov_bytes.device
    sizes = torch.tensor([ov_bytes.numel(), cig_bytes.numel()], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    ops, parts = [], None
    if rank == 0:
        sz = [(int(a[0]), int(a[1])) for a in all_sizes]
        rows_all = torch.empty(sum(a for a, _ in sz), dtype=torch.uint8, device=dev)
        pool_all = torch.empty(sum(b for _, b in sz), dtype=torch.uint8, device=dev)
        ro, po = sz[0]
        rows_all[:ro].copy_(ov_bytes)
        pool_all[:po].copy_(cig_bytes)
        for r in range(1, world):
            a, b = sz[r]
            if a:
                ops.append(dist.P2POp(dist.irecv, rows_all[ro:ro + a], r, group))
            if b:
                ops.append(dist.P2POp(dist.irecv, pool_all[po:po + b], r, group))
            ro += a
            po += b
        parts = (rows_all, pool_all, sz)
    else:
        if ov_bytes.numel():
            ops.append(dist.P2POp(dist.isend, ov_bytes, 0, group))
        if cig_bytes.numel():
            ops.append(dist.P2POp(dist.isend, cig_bytes, 0, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    return {"reqs": reqs, "parts": parts, "keep": (ov_bytes, cig_bytes)}


def start_gather_sharded(ctx, n_local_pairs, pair_lo, n_pairs_total, dev, group=None):
    """The gather without a merge step.  Every rank asks its context how its last results split into the
    R1 / R2 blocks (kslam_shard_counts_device), the counts are all-gathered, every rank exports its
    records in BATCH terms (kslam_export_shard_device: read ids and CIGAR offsets re-based on its own
    GPU) and sends four pieces that land in their final places on rank 0.  finish_gather returns on
    rank 0 (rows, pool) -- uint8 device tensors holding the batch-global result, byte for byte what
    one context returns for the whole batch -- elsewhere None.

    With the gloo backend (no device-to-device transport: the CPU tests, and the N = 2 test that runs two
    ranks on ONE GPU) the pieces are staged through host memory; the protocol is the same."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    staged = dist.get_backend(group) == "gloo"
    cdev = torch.device("cpu") if staged else dev        # where the communicator's tensors live
    mine = ctx.shard_counts_device(n_local_pairs)
    t = torch.tensor(mine, dtype=torch.int64, device=cdev)
    allc = [torch.zeros(4, dtype=torch.int64, device=cdev) for _ in range(world)]
    dist.all_gather(allc, t, group=group)
    cnt = [tuple(int(v) for v in a.tolist()) for a in allc]       # (rows, rows_r1, ops, ops_r1) per rank
    rows_r1 = sum(c[1] for c in cnt)
    ops_r1 = sum(c[3] for c in cnt)
    row1 = [sum(c[1] for c in cnt[:r]) for r in range(world)]
    row2 = [rows_r1 + sum(c[0] - c[1] for c in cnt[:r]) for r in range(world)]
    op1 = [sum(c[3] for c in cnt[:r]) for r in range(world)]
    op2 = [ops_r1 + sum(c[2] - c[3] for c in cnt[:r]) for r in range(world)]
    ops, parts, keep, landed = [], None, None, []
    if rank == 0:
        rows = torch.empty(sum(c[0] for c in cnt) * 48, dtype=torch.uint8, device=dev)
        pool = torch.empty(sum(c[2] for c in cnt) * 4, dtype=torch.uint8, device=dev)
        rp, pp = rows.data_ptr(), pool.data_ptr()
        ctx.export_shard_device(n_local_pairs, pair_lo, n_pairs_total, op1[0], op2[0], rp + 48 * row1[0],
                                rp + 48 * row2[0], pp + 4 * op1[0], pp + 4 * op2[0])
        for r in range(1, world):
            n, n1, c, c1 = cnt[r]
            for buf, start, count, unit in ((rows, row1[r], n1, 48), (rows, row2[r], n - n1, 48),
                                            (pool, op1[r], c1, 4), (pool, op2[r], c - c1, 4)):
                if count:
                    dst = buf[start * unit:(start + count) * unit]
                    if staged:
                        h = torch.empty(count * unit, dtype=torch.uint8)
                        landed.append((dst, h))
                        dst = h
                    ops.append(dist.P2POp(dist.irecv, dst, r, group))
        parts = (rows, pool)
    else:
        n, n1, c, c1 = cnt[rank]
        srows = torch.empty(max(n, 1) * 48, dtype=torch.uint8, device=dev)
        spool = torch.empty(max(c, 1) * 4, dtype=torch.uint8, device=dev)
        rp, pp = srows.data_ptr(), spool.data_ptr()
        ctx.export_shard_device(n_local_pairs, pair_lo, n_pairs_total, op1[rank], op2[rank], rp, rp + 48 * n1,
                                pp, pp + 4 * c1)
        if staged:
            srows, spool = srows.cpu(), spool.cpu()
        for buf, start, count, unit in ((srows, 0, n1, 48), (srows, n1, n - n1, 48), (spool, 0, c1, 4), (spool, c1, c - c1, 4)):
            if count:
                ops.append(dist.P2POp(dist.isend, buf[start * unit:(start + count) * unit], 0, group))
        keep = (srows, spool)
    reqs = dist.batch_isend_irecv(ops) if ops else []
    return {"reqs": reqs, "parts": parts, "keep": keep, "landed": landed}


def finish_gather(handle):
    """Wait for a gather begun with start_gather.  Returns on rank 0 a list of (ov, cig) uint8
    tensors per rank, elsewhere None."""
    for q in handle["reqs"]:
        q.wait()
    for dst, h in handle.get("landed", ()):     # host-staged pieces (gloo) into their final places
        dst.copy_(h)
    return handle["parts"]


def gather_to_rank0(ov_bytes, cig_bytes, group=None):
    """start_gather + finish_gather in one call."""
    return finish_gather(start_gather(ov_bytes, cig_bytes, group))


def reassemble(parts, bounds, n_pairs, overlap_dtype):
    """Rank 0: per-rank results (local read ids) -> one batch-global result in the
    reference order (read, entry, rel).  parts[r] = (overlaps ndarray, cigar ndarray)."""
    r1_parts, r2_parts, pools = [], [], []
    base = 0
    for (ov, cig), (lo, hi) in zip(parts, bounds):
        ov = ov.copy()
        n_loc = hi - lo
        ov["cigar_off"] = np.where(ov["cigar_len"] > 0, ov["cigar_off"] + np.uint64(base), np.uint64(0))
        is_r2 = ov["read"] >= n_loc
        ov["read"] = np.where(is_r2, ov["read"] - n_loc + n_pairs + lo, ov["read"] + lo).astype(np.uint32)
        r1_parts.append(ov[~is_r2])
        r2_parts.append(ov[is_r2])
        pools.append(cig)
        base += len(cig)
    out = np.concatenate(r1_parts + r2_parts) if parts else np.zeros(0, dtype=overlap_dtype)
    pool = np.concatenate(pools) if pools else np.zeros(0, dtype=np.uint32)
    return out, pool


# ---- the batch-global steps of the tail when read pairs are sharded (include/kslam.h: kslam_pair_phase_a / _b, kslam_pseudo_merged) ----

class _DevView:
    """n bytes of device memory at `ptr` as an object torch can wrap without copying (__cuda_array_interface__)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def device_bytes(ptr, nbytes, dev):
    """a uint8 device tensor holding a COPY of nbytes at device address ptr (dev = cpu: at host address ptr -- the CPU
    tests' stand-in contexts)"""
    if nbytes == 0 or not ptr:
        return torch.empty(0, dtype=torch.uint8, device=dev)
    if torch.device(dev).type == "cpu":
        import ctypes
        return torch.from_numpy(np.frombuffer((ctypes.c_char * int(nbytes)).from_address(int(ptr)), dtype=np.uint8).copy())
    return torch.as_tensor(_DevView(ptr, nbytes), device=dev).clone()


def _sync(dev):
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize(dev)


def all_gather_bytes(mine, dev, group=None):
    """variable-length all-gather of uint8 device tensors -> (concatenation in rank order on `dev`, [bytes per rank]).
    nccl: device tensors padded to the longest; gloo (tests: ranks sharing one GPU): staged through host memory."""
    world = dist.get_world_size(group)
    staged = dist.get_backend(group) == "gloo"
    cdev = torch.device("cpu") if staged else dev
    n = torch.tensor([mine.numel()], dtype=torch.int64, device=cdev)
    counts = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    longest = max(counts)
    if longest == 0:
        return torch.empty(0, dtype=torch.uint8, device=dev), counts
    pad = torch.zeros(longest, dtype=torch.uint8, device=cdev)
    pad[:mine.numel()] = mine.to(cdev)
    parts = [torch.empty(longest, dtype=torch.uint8, device=cdev) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.cat([p[:c] for p, c in zip(parts, counts)]).to(dev)
    return out, counts


def all_to_all_bytes(mine, send_bytes, dev, group=None):
    """variable-length all-to-all of uint8 device tensors: `mine` holds the piece for rank 0, then the piece for rank 1, ...
    (send_bytes[d] bytes each) -> (what this rank received, pieces end to end in SOURCE-RANK order, on `dev`;
    [bytes received per source rank]).  nccl: all_to_all_single on device tensors (grouped send / recv over xGMI);
    gloo (tests: ranks sharing one GPU): the same call on host copies."""
    world = dist.get_world_size(group)
    staged = dist.get_backend(group) == "gloo"
    cdev = torch.device("cpu") if staged else dev
    n = torch.tensor(send_bytes, dtype=torch.int64, device=cdev)
    counts = [torch.zeros(world, dtype=torch.int64, device=cdev) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    recv_bytes = [int(counts[src][dist.get_rank(group)]) for src in range(world)]
    out = torch.empty(sum(recv_bytes), dtype=torch.uint8, device=cdev)
    dist.all_to_all_single(out, mine.to(cdev), recv_bytes, [int(v) for v in send_bytes], group=group)
    return out.to(dev), recv_bytes


def routed_pseudo_assembly(ctx, dev, score_fraction=0.95, group=None):
    """pseudoAssembly (src/PairedOverlap.h:480-582) with the ENTRIES partitioned over the ranks: entry e belongs to rank
    e mod world.  kslam_pseudo_route -> all-to-all of 16-byte heads -> kslam_pseudo_owned -> all-to-all of the 4-byte scores
    back -> kslam_pseudo_return (include/kslam.h).  A rank whose device stage declines makes EVERY rank raise (a status
    word travels with the scores), none is left waiting.  Returns (stats dict, bytes this rank received)."""
    from . import KslamError
    world = dist.get_world_size(group)
    staged = dist.get_backend(group) == "gloo"
    fdev = torch.device("cpu") if staged else dev

    def agree(status):
        """the worst status of all ranks, before anybody enters a transfer that depends on the step"""
        flag = torch.tensor([status], dtype=torch.int64, device=fdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        return int(flag.item())
    # the routing step can fail on one rank alone (out of memory in the route buffers, a world beyond 256): its status is
    # shared BEFORE the first all-to-all, so that the others do not wait in it for a rank that has left
    status, counts, n_own, heads = 0, [0] * world, 0, torch.empty(0, dtype=torch.uint8, device=dev)
    try:
        d_heads, counts = ctx.pseudo_route(world)
        n_own = sum(counts)
        heads = device_bytes(d_heads, n_own * 16, dev)
    except KslamError as e:
        status = int(e.status) or 1
    _sync(dev)
    worst = agree(status)
    if worst:
        raise KslamError(worst, "the routing step of the partitioned pseudo-assembly failed on some rank: no rank enters the exchange")
    got, recv = all_to_all_bytes(heads, [c * 16 for c in counts], dev, group)
    _sync(dev)
    n_recv = got.numel() // 16
    status, scores = 0, torch.empty(0, dtype=torch.uint8, device=dev)
    try:
        d_scores = ctx.pseudo_owned(got.data_ptr() if n_recv else None, n_recv)
        scores = device_bytes(d_scores, n_recv * 4, dev)
    except KslamError as e:
        status, scores = int(e.status) or 1, torch.zeros(n_recv * 4, dtype=torch.uint8, device=dev)
    _sync(dev)
    # the status of every rank, before anybody depends on the scores
    worst = agree(status)
    back, _ = all_to_all_bytes(scores, [b // 4 for b in recv], dev, group)
    _sync(dev)
    if worst:
        raise KslamError(worst, "the pseudo-assembly of some rank's entries declined on the device: the batch's stage belongs to the host")
    assert back.numel() == n_own * 4
    # ... and the last step: a rank that fails to commit the scores must not leave the others with a batch it does not have
    status, stats = 0, None
    try:
        stats = ctx.pseudo_return(back.data_ptr() if n_own else None, n_own, score_fraction)
    except KslamError as e:
        status = int(e.status) or 1
    worst = agree(status)
    if worst:
        raise KslamError(worst, "kslam_pseudo_return failed on some rank: the batch's pseudo-assembly is void on every rank")
    return stats, got.numel() + back.numel()


def sharded_tail(ctx, dev, paired=True, score_threshold=0, score_fraction=0.95, pseudo_assembly=True, group=None, routed=True):
    """Pairing, insert-size screen, score screen [, pseudo-assembly + second screen] for THIS rank's read pairs of a batch
    sharded over the ranks -- the reference's steps between alignToDatabase and the SAM writer (src/SLAM.h:210-233) -- with
    the two batch-global steps fed from all ranks: the insert sizes (4 bytes per properly paired read pair) are all-gathered;
    pseudo-assembly runs with the entries partitioned over the ranks (routed_pseudo_assembly: 16 bytes out and 4 back per
    alignment pair, 1 / world of the stage per rank) -- routed=False keeps round 3's form, an all-gather of the 32-byte
    records and the whole stage on every rank.  Afterwards the context holds this rank's read pairs / alignment pairs as
    after kslam_pair_screen.  Returns (stats dict, bytes this rank received in the exchanges)."""
    rank = dist.get_rank(group)
    d_ins, n_ins = ctx.pair_phase_a(paired, score_threshold)
    mine = device_bytes(d_ins, n_ins * 4, dev)
    _sync(dev)
    all_ins, c1 = all_gather_bytes(mine, dev, group)
    _sync(dev)
    stats, d_pairs, n_pairs = ctx.pair_phase_b(all_ins.data_ptr() if all_ins.numel() else None, all_ins.numel() // 4, score_fraction, 3)
    moved = sum(c1)
    if pseudo_assembly and routed:
        stats, m2 = routed_pseudo_assembly(ctx, dev, score_fraction, group)
        moved += m2
    elif pseudo_assembly:
        recs = device_bytes(d_pairs, n_pairs * 32, dev)
        _sync(dev)
        all_recs, c2 = all_gather_bytes(recs, dev, group)
        _sync(dev)
        # raises KslamError(KSLAM_ERR_UNSUPPORTED) when the device stage declines (2^28 or more records in the batch):
        # a rank must not fall back to pseudo-assembling its OWN pairs, the stage is batch-global (src/PairedOverlap.h:480-582)
        stats = ctx.pseudo_merged(all_recs.data_ptr() if all_recs.numel() else None, all_recs.numel() // 32, sum(c2[:rank]) // 32,
                                  score_fraction)
        moved += sum(c2)
    return stats, moved
