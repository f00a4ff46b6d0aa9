"""ctypes plumbing for include/kslam_comm.h: the end-of-batch exchanges of a read-sharded batch over RCCL behind the C ABI
(one process per GPU, no PyTorch involved).  k-slam_amd/dist.py is the same protocol through torch.distributed."""
import ctypes as C

import numpy as np

from . import KslamError, PairStats, lib as _base_lib

EXPORTS = ["kslam_comm_last_error", "kslam_comm_unique_id", "kslam_comm_create", "kslam_comm_destroy", "kslam_comm_rank",
           "kslam_comm_world", "kslam_comm_gather_plan", "kslam_comm_gather_batch", "kslam_comm_gather_begin", "kslam_comm_gather_end", "kslam_comm_sharded_tail", "kslam_comm_info"]
ID_BYTES = 128
_ready = False


class CommFacts(C.Structure):
    """kslam_comm_facts (include/kslam_comm.h)"""
    _fields_ = [("comm_count", C.c_int32), ("comm_rank", C.c_int32), ("rccl_version", C.c_int32), ("device", C.c_int32),
                ("library", C.c_char * 240)]


class ShardCounts(C.Structure):
    """kslam_shard_counts (include/kslam.h)"""
    _fields_ = [("n_rows", C.c_uint64), ("n_rows_r1", C.c_uint64), ("n_cigar", C.c_uint64), ("n_cigar_r1", C.c_uint64)]


def lib():
    global _ready
    L = _base_lib()
    if not _ready:
        vp, u64 = C.c_void_p, C.c_uint64
        L.kslam_comm_last_error.restype = C.c_char_p
        L.kslam_comm_unique_id.argtypes = [vp]
        L.kslam_comm_create.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
        L.kslam_comm_destroy.argtypes = [vp]
        L.kslam_comm_rank.argtypes = [vp]
        L.kslam_comm_world.argtypes = [vp]
        L.kslam_comm_gather_plan.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
        L.kslam_comm_gather_plan.restype = None
        L.kslam_comm_gather_batch.argtypes = [vp, u64, u64, u64, C.POINTER(vp), C.POINTER(u64), C.POINTER(vp), C.POINTER(u64)]
        L.kslam_comm_sharded_tail.argtypes = [vp, C.c_int, C.c_uint32, C.c_double, C.c_int, C.POINTER(PairStats), C.POINTER(u64)]
        L.kslam_comm_gather_begin.argtypes = [vp, u64, u64, u64]
        L.kslam_comm_gather_end.argtypes = [vp, C.POINTER(vp), C.POINTER(u64), C.POINTER(vp), C.POINTER(u64)]
        L.kslam_comm_info.argtypes = [vp, C.POINTER(CommFacts)]
        _ready = True
    return L


def _chk(st):
    if st != 0:
        raise KslamError(st, lib().kslam_comm_last_error().decode())


def unique_id():
    """rank 0: ncclGetUniqueId -> 128 bytes to hand to the other ranks"""
    buf = (C.c_uint8 * ID_BYTES)()
    _chk(lib().kslam_comm_unique_id(buf))
    return bytes(buf)


def gather_plan(counts):
    """counts: [(n_rows, n_rows_r1, n_cigar, n_cigar_r1)] per rank -> (row1, row2, op1, op2, (rows, ops))"""
    w = len(counts)
    arr = (ShardCounts * w)(*[ShardCounts(*c) for c in counts])
    out = [np.zeros(w, dtype=np.uint64) for _ in range(4)]
    tot = np.zeros(2, dtype=np.uint64)
    lib().kslam_comm_gather_plan(arr, w, *[o.ctypes.data for o in out], tot.ctypes.data)
    return tuple(o.tolist() for o in out) + ((int(tot[0]), int(tot[1])),)


class Comm:
    """kslam_comm: ncclCommInitRank on the context's device"""

    def __init__(self, ctx, uid, rank, world):
        self._L, self._h, self._ctx = lib(), C.c_void_p(), ctx
        buf = (C.c_uint8 * ID_BYTES).from_buffer_copy(uid)
        _chk(self._L.kslam_comm_create(ctx._h, buf, rank, world, C.byref(self._h)))

    def close(self):
        if self._h:
            self._L.kslam_comm_destroy(self._h)
            self._h = C.c_void_p()

    def info(self):
        """kslam_comm_info -> what RCCL reports about this communicator"""
        f = CommFacts()
        _chk(self._L.kslam_comm_info(self._h, C.byref(f)))
        return {"comm_count": int(f.comm_count), "comm_rank": int(f.comm_rank), "rccl_version": int(f.rccl_version),
                "device": int(f.device), "library": f.library.decode(errors="replace")}

    def gather_batch(self, n_local_pairs, pair_lo, n_pairs_total):
        """-> rank 0: (device pointer of the rows, n_rows, device pointer of the pool, n_ops); elsewhere (None, 0, None, 0)"""
        rows, pool, n, m = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        _chk(self._L.kslam_comm_gather_batch(self._h, n_local_pairs, pair_lo, n_pairs_total, C.byref(rows), C.byref(n),
                                             C.byref(pool), C.byref(m)))
        return rows.value, int(n.value), pool.value, int(m.value)

    def gather_begin(self, n_local_pairs, pair_lo, n_pairs_total):
        """kslam_comm_gather_begin: counts, export, the transfers posted; the context is free for the next batch"""
        _chk(self._L.kslam_comm_gather_begin(self._h, n_local_pairs, pair_lo, n_pairs_total))

    def gather_end(self):
        """kslam_comm_gather_end -> what gather_batch returns"""
        rows, pool, n, m = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        _chk(self._L.kslam_comm_gather_end(self._h, C.byref(rows), C.byref(n), C.byref(pool), C.byref(m)))
        return rows.value, int(n.value), pool.value, int(m.value)

    def sharded_tail(self, paired=True, score_threshold=0, score_fraction=0.95, pseudo_assembly=True):
        st, moved = PairStats(), C.c_uint64()
        _chk(self._L.kslam_comm_sharded_tail(self._h, int(paired), score_threshold, score_fraction, int(pseudo_assembly),
                                             C.byref(st), C.byref(moved)))
        return st.as_dict(), int(moved.value)
