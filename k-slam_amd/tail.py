"""ctypes plumbing over the host-tail entry points of libkslam_hip.so (include/kslam_tail.h).

Pairing -> insert-size screen -> score screen -> pseudo-assembly -> SAM records,
i.e. what the reference runs between alignToDatabase and the taxonomy step
(reference src/SLAM.h:101-133).  Host-only calls: they work without a GPU.
"""
import ctypes as C

import numpy as np

from . import KslamError, OVERLAP_DT, lib as _base_lib

NO_OVERLAP = 0xFFFFFFFF
STAGE_INSERT, STAGE_SCORE, STAGE_PSEUDO, STAGE_ALL, STAGE_PAIRING_ONLY = 1, 2, 4, 7, 8

PAIRED_OVERLAP_DT = np.dtype([("combined_score", "<u4"), ("entry", "<u4"), ("ref_start", "<i4"),
                              ("ref_end", "<i4"), ("insert_size", "<u4"), ("r1", "<u4"),
                              ("r2", "<u4"), ("pad", "<u4")])
READ_PAIR_DT = np.dtype([("r1_read", "<u4"), ("r2_read", "<u4"), ("first", "<u8"), ("count", "<u8")])
assert PAIRED_OVERLAP_DT.itemsize == 32 and READ_PAIR_DT.itemsize == 24

# every symbol include/kslam_tail.h declares
EXPORTS = ["kslam_tail_last_error", "kslam_tail_pairs", "kslam_sam_records", "kslam_tail_sam",
           "kslam_tail_sam_write", "kslam_tail_sam_rows", "kslam_tail_sam_write_rows", "kslam_tail_finish_write_rows",
           "kslam_tail_finish_prepare", "kslam_tail_release_buffers",
           "kslam_sam_header", "kslam_write_fd", "kslam_sam_writer_open", "kslam_write_queued", "kslam_sam_writer_enqueue", "kslam_sam_writer_close"]
WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)

_vp, _u64, _u32, _i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32


class TailParams(C.Structure):
    """kslam_tail_params: the globals of reference src/Globals.h:31-42 the tail reads."""
    _fields_ = [("score_threshold", _u32), ("num_sam_alignments", _u32), ("score_fraction", C.c_double),
                ("pseudo_assembly", _i32), ("sam_xa", _i32), ("report_cigar", _i32), ("paired", _i32),
                ("stages", _u32), ("threads", _i32)]

    @classmethod
    def default(cls, paired=True, report_cigar=True, score_threshold=0, num_sam_alignments=10,
                score_fraction=0.95, pseudo_assembly=True, sam_xa=False, stages=0, threads=0):
        # defaults of reference src/main.cpp:40-97
        return cls(score_threshold, num_sam_alignments, score_fraction, int(pseudo_assembly),
                   int(sam_xa), int(report_cigar), int(paired), stages, threads)


class ReadsView(C.Structure):
    _fields_ = [("n_reads", _u64), ("bases", _vp), ("bases_off", _vp), ("quality", _vp),
                ("quality_off", _vp), ("ids", _vp), ("ids_off", _vp)]


class IndexView(C.Structure):
    _fields_ = [("n_entries", _u64), ("bases", _vp), ("bases_off", _vp), ("locus_tag", _vp),
                ("locus_tag_off", _vp), ("taxonomy_id", _vp), ("n_genes", _u64), ("gene_first", _vp),
                ("gene_start", _vp), ("gene_stop", _vp), ("gene_name", _vp), ("gene_name_off", _vp),
                ("protein_id", _vp), ("protein_id_off", _vp), ("product", _vp), ("product_off", _vp)]


class TailStats(C.Structure):
    _fields_ = [("n_overlaps_in", _u64), ("n_overlaps_screened", _u64), ("n_paired_initial", _u64),
                ("n_paired_final", _u64), ("n_read_pairs", _u64), ("n_insert_sizes", _u64),
                ("max_insert_size", _u32), ("threads", _u32), ("ms_pairing", C.c_double),
                ("ms_insert", C.c_double), ("ms_screens", C.c_double), ("ms_pseudo", C.c_double),
                ("ms_sam", C.c_double), ("sam_bytes", _u64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def _column(items):
    """list of bytes -> (uint8 text array, uint64 offsets)"""
    off = np.zeros(len(items) + 1, dtype=np.uint64)
    if len(items):
        np.cumsum(np.fromiter((len(s) for s in items), dtype=np.uint64, count=len(items)), out=off[1:])
    text = np.frombuffer(b"".join(items) + b"\0", dtype=np.uint8)
    return text, off


def _p(a):
    return a.ctypes.data


class Reads:
    """Column view of a read batch ([R1 block | R2 block] when paired)."""

    def __init__(self, bases, quality=None, ids=None):
        n = len(bases)
        if quality is None:
            quality = [b"I" * len(b) for b in bases]
        if ids is None:
            ids = [b"read%d" % (i % (n // 2) if n >= 2 else i) for i in range(n)]
        self._keep = [_column(bases), _column(quality), _column(ids)]
        (b, bo), (q, qo), (i, io) = self._keep
        self.view = ReadsView(n, _p(b), _p(bo), _p(q), _p(qo), _p(i), _p(io))


class ReadsArrays(Reads):
    """The same view over numpy columns that already exist (no per-read Python objects):
    fixed-length reads as one uint8 array, offsets = arange * read_len."""

    def __init__(self, bases_u8, read_len, quality_u8=None, ids=None):
        bases_u8 = np.ascontiguousarray(bases_u8, dtype=np.uint8).reshape(-1)
        n = len(bases_u8) // read_len
        off = np.arange(n + 1, dtype=np.uint64) * np.uint64(read_len)
        if quality_u8 is None:
            quality_u8 = np.full(n * read_len, ord("I"), dtype=np.uint8)
        quality_u8 = np.ascontiguousarray(quality_u8, dtype=np.uint8).reshape(-1)
        if ids is None:
            half = max(n // 2, 1)
            ids = [b"r%d" % (i % half) for i in range(n)]
        idc = _column(ids)
        self._keep = [bases_u8, off, quality_u8, idc]
        self.view = ReadsView(n, _p(bases_u8), _p(off), _p(quality_u8), _p(off), _p(idc[0]), _p(idc[1]))


class IndexArrays:
    """Index view over one concatenated uint8 genome array + offsets."""

    def __init__(self, bases_u8, offsets, locus_tags=None, taxonomy_ids=None):
        bases_u8 = np.ascontiguousarray(bases_u8, dtype=np.uint8).reshape(-1)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(off) - 1
        if locus_tags is None:
            locus_tags = [b"entry%d" % i for i in range(n)]
        tax = np.zeros(max(n, 1), dtype=np.uint32)
        if taxonomy_ids is not None:
            tax[:n] = taxonomy_ids
        lc = _column(locus_tags)
        self._keep = [bases_u8, off, lc, tax]
        self.view = IndexView(n, _p(bases_u8), _p(off), _p(lc[0]), _p(lc[1]), _p(tax))


class Index:
    """Column view of the GenbankIndex fields the tail reads.

    genes: optional list (per entry) of lists of (start, stop, gene_name, protein_id, product)."""

    def __init__(self, entries, locus_tags=None, taxonomy_ids=None, genes=None):
        n = len(entries)
        if locus_tags is None:
            locus_tags = [b"entry%d" % i for i in range(n)]
        tax = np.zeros(max(n, 1), dtype=np.uint32)
        if taxonomy_ids is not None:
            tax[:n] = taxonomy_ids
        self._keep = [_column(entries), _column(locus_tags), tax]
        (b, bo), (l, lo), _ = self._keep
        v = IndexView(n, _p(b), _p(bo), _p(l), _p(lo), _p(tax))
        if genes:
            first = np.zeros(n + 1, dtype=np.uint64)
            flat = []
            for e in range(n):
                flat.extend(genes[e])
                first[e + 1] = len(flat)
            gs = np.array([g[0] for g in flat], dtype=np.uint32).view(np.int32)
            ge = np.array([g[1] for g in flat], dtype=np.uint32).view(np.int32)
            cols = [_column([g[k] for g in flat]) for k in (2, 3, 4)]
            self._keep += [first, gs, ge, cols]
            v.n_genes = len(flat)
            v.gene_first, v.gene_start, v.gene_stop = _p(first), _p(gs), _p(ge)
            v.gene_name, v.gene_name_off = _p(cols[0][0]), _p(cols[0][1])
            v.protein_id, v.protein_id_off = _p(cols[1][0]), _p(cols[1][1])
            v.product, v.product_off = _p(cols[2][0]), _p(cols[2][1])
        self.view = v


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = _base_lib()
        P = C.POINTER
        L.kslam_tail_last_error.restype = C.c_char_p
        L.kslam_tail_pairs.argtypes = [P(TailParams), P(ReadsView), _vp, _u64, P(_vp), P(_u64), P(_vp),
                                       P(_u64), P(TailStats)]
        L.kslam_sam_records.argtypes = [P(TailParams), P(ReadsView), P(IndexView), _vp, _u64, _vp, _u64,
                                        _vp, _u64, _vp, _u64, P(_vp), P(_u64), P(TailStats)]
        L.kslam_tail_sam.argtypes = [P(TailParams), P(ReadsView), P(IndexView), _vp, _u64, _vp, _u64,
                                     P(_vp), P(_u64), P(TailStats)]
        L.kslam_tail_sam_write.argtypes = [P(TailParams), P(ReadsView), P(IndexView), _vp, _u64, _vp, _u64,
                                           WRITE_FN, _vp, P(TailStats)]
        L.kslam_tail_sam_rows.argtypes = [P(TailParams), P(ReadsView), P(IndexView), _vp, _u64, _vp, _u64,
                                          _vp, _vp, _u64, P(_vp), P(_u64), P(TailStats)]
        L.kslam_tail_sam_write_rows.argtypes = [P(TailParams), P(ReadsView), P(IndexView), _vp, _u64, _vp, _u64,
                                                _vp, _vp, _u64, WRITE_FN, _vp, P(TailStats)]
        L.kslam_tail_finish_write_rows.argtypes = [P(TailParams), P(ReadsView), P(IndexView), _vp, _u64, _vp, _u64,
                                                   _vp, _vp, _u64, _vp, _u64, _vp, _u64, WRITE_FN, _vp, P(TailStats)]
        L.kslam_tail_finish_prepare.argtypes = [P(TailParams), P(ReadsView), _vp, _u64, _vp, _u64, _vp, _u64, C.c_int, P(TailStats)]
        L.kslam_tail_release_buffers.restype = None
        L.kslam_sam_header.argtypes = [P(IndexView), C.c_char_p, P(_vp), P(_u64)]
        L.kslam_sam_writer_open.argtypes = [C.c_int, P(_vp)]
        L.kslam_sam_writer_close.argtypes = [_vp, P(_u64), P(C.c_double)]
        L.kslam_write_queued.argtypes = [_vp, C.c_char_p, _u64]
        _lib = L
    return _lib


def _chk(st):
    if st != 0:
        raise KslamError(st, lib().kslam_tail_last_error().decode())


def _take(ptr, n, dtype):
    """copy a malloc'ed array out and release it"""
    L = lib()
    if n:
        buf = (C.c_char * (int(n) * dtype.itemsize)).from_address(ptr.value)
        out = np.frombuffer(buf, dtype=dtype).copy()
    else:
        out = np.zeros(0, dtype=dtype)
    if ptr.value:
        L.kslam_free(ptr)
    return out


def _ov(overlaps):
    ov = np.ascontiguousarray(overlaps, dtype=OVERLAP_DT)
    return ov, (_p(ov) if len(ov) else None)


def tail_pairs(params, reads, overlaps):
    """kslam_tail_pairs -> (read_pairs, pairs, stats)"""
    L = lib()
    ov, pov = _ov(overlaps)
    rp, pr, nrp, npr, st = _vp(), _vp(), _u64(), _u64(), TailStats()
    _chk(L.kslam_tail_pairs(C.byref(params), C.byref(reads.view), pov, len(ov), C.byref(rp),
                            C.byref(nrp), C.byref(pr), C.byref(npr), C.byref(st)))
    return _take(rp, nrp.value, READ_PAIR_DT), _take(pr, npr.value, PAIRED_OVERLAP_DT), st


def _text(ptr, n):
    L = lib()
    # (ctypes.string_at takes a C int: the XML report of a 100 M-pair run is longer than that)
    out = bytes((C.c_char * n.value).from_address(ptr.value)) if n.value else b""
    if ptr.value:
        L.kslam_free(ptr)
    return out


def tail_sam(params, reads, index, overlaps, cigar_pool):
    """kslam_tail_sam -> (SAM records as bytes, stats)"""
    L = lib()
    ov, pov = _ov(overlaps)
    pool = np.ascontiguousarray(cigar_pool, dtype=np.uint32)
    txt, n, st = _vp(), _u64(), TailStats()
    _chk(L.kslam_tail_sam(C.byref(params), C.byref(reads.view), C.byref(index.view), pov, len(ov),
                          _p(pool) if len(pool) else None, len(pool), C.byref(txt), C.byref(n),
                          C.byref(st)))
    return _text(txt, n), st


def tail_sam_write(params, reads, index, overlaps, cigar_pool, sink):
    """kslam_tail_sam_write: sink(bytes) is called per chunk, in order -> stats"""
    L = lib()
    ov, pov = _ov(overlaps)
    pool = np.ascontiguousarray(cigar_pool, dtype=np.uint32)
    st = TailStats()

    def _cb(user, data, n):
        try:
            sink(C.string_at(data, n))
            return 0
        except Exception:  # reported to the caller as a failed write
            return 1

    cb = WRITE_FN(_cb)
    _chk(L.kslam_tail_sam_write(C.byref(params), C.byref(reads.view), C.byref(index.view), pov, len(ov),
                                _p(pool) if len(pool) else None, len(pool), cb, None, C.byref(st)))
    return st


def tail_sam_discard(params, reads, index, overlaps, cigar_pool):
    """kslam_tail_sam_write with a writer that drops the text (timing runs) -> stats"""
    L = lib()
    ov, pov = _ov(overlaps)
    pool = np.ascontiguousarray(cigar_pool, dtype=np.uint32)
    st = TailStats()
    cb = WRITE_FN(lambda user, data, n: 0)
    _chk(L.kslam_tail_sam_write(C.byref(params), C.byref(reads.view), C.byref(index.view), pov, len(ov),
                                _p(pool) if len(pool) else None, len(pool), cb, None, C.byref(st)))
    return st


def tail_sam_rows(params, reads, index, overlaps, cigar_pool, details, md_pool):
    """kslam_tail_sam_rows: as tail_sam with the per-row details of kslam_row_details (ROW_DETAIL_DT array +
    MD bytes as a uint8 array) -> (SAM records as bytes, stats)"""
    L = lib()
    ov, pov = _ov(overlaps)
    pool = np.ascontiguousarray(cigar_pool, dtype=np.uint32)
    det = np.ascontiguousarray(details)
    md = np.ascontiguousarray(md_pool, dtype=np.uint8)
    txt, n, st = _vp(), _u64(), TailStats()
    _chk(L.kslam_tail_sam_rows(C.byref(params), C.byref(reads.view), C.byref(index.view), pov, len(ov),
                               _p(pool) if len(pool) else None, len(pool), _p(det) if len(det) else None,
                               _p(md) if len(md) else None, len(md), C.byref(txt), C.byref(n), C.byref(st)))
    return _text(txt, n), st


def tail_sam_discard_rows(params, reads, index, overlaps, cigar_pool, details, md_pool):
    """kslam_tail_sam_write_rows with a writer that drops the text (timing runs) -> stats"""
    L = lib()
    ov, pov = _ov(overlaps)
    pool = np.ascontiguousarray(cigar_pool, dtype=np.uint32)
    det = np.ascontiguousarray(details)
    md = np.ascontiguousarray(md_pool, dtype=np.uint8)
    st = TailStats()
    cb = WRITE_FN(lambda user, data, n: 0)
    _chk(L.kslam_tail_sam_write_rows(C.byref(params), C.byref(reads.view), C.byref(index.view), pov, len(ov),
                                     _p(pool) if len(pool) else None, len(pool), _p(det) if len(det) else None,
                                     _p(md) if len(md) else None, len(md), cb, None, C.byref(st)))
    return st


def tail_finish_rows(params, reads, index, overlaps, cigar_pool, details, md_pool, read_pairs, pairs, sink=None):
    """kslam_tail_finish_write_rows: read pairs / alignment pairs from kslam_pair_screen (both arrays are
    MODIFIED in place) -> [pseudo-assembly + second score screen] -> SAM text to sink(bytes) (None: dropped)
    -> stats.  details / md_pool may be None."""
    L = lib()
    ov, pov = _ov(overlaps)
    pool = np.ascontiguousarray(cigar_pool, dtype=np.uint32)
    det = np.ascontiguousarray(details) if details is not None else None
    md = np.ascontiguousarray(md_pool, dtype=np.uint8) if md_pool is not None else np.zeros(0, dtype=np.uint8)
    assert read_pairs.dtype == READ_PAIR_DT and pairs.dtype == PAIRED_OVERLAP_DT
    assert read_pairs.flags["C_CONTIGUOUS"] and pairs.flags["C_CONTIGUOUS"] and read_pairs.flags["WRITEABLE"] and pairs.flags["WRITEABLE"]
    st = TailStats()

    def _cb(user, data, n):
        try:
            if sink is not None:
                sink(C.string_at(data, n))
            return 0
        except Exception:
            return 1
    cb = WRITE_FN(_cb)
    _chk(L.kslam_tail_finish_write_rows(C.byref(params), C.byref(reads.view), C.byref(index.view), pov, len(ov),
                                        _p(pool) if len(pool) else None, len(pool),
                                        _p(det) if det is not None and len(det) else None,
                                        _p(md) if len(md) else None, len(md),
                                        _p(read_pairs) if len(read_pairs) else None, len(read_pairs),
                                        _p(pairs) if len(pairs) else None, len(pairs), cb, None, C.byref(st)))
    return st


def tail_finish_prepare(params, reads, overlaps, read_pairs, pairs, sort_groups=True):
    """kslam_tail_finish_prepare: the part of the finish that CHANGES read_pairs / pairs (host pseudo-assembly + second
    screen when params ask, then writeSAMOutputPairs' per-pair sort), in place.  Afterwards tail_finish_rows with
    pseudo_assembly=False and stages | KSLAM_TAIL_GROUPS_SORTED (16) only reads them."""
    L = lib()
    ov, pov = _ov(overlaps)
    assert read_pairs.dtype == READ_PAIR_DT and pairs.dtype == PAIRED_OVERLAP_DT
    assert read_pairs.flags["C_CONTIGUOUS"] and pairs.flags["C_CONTIGUOUS"] and read_pairs.flags["WRITEABLE"] and pairs.flags["WRITEABLE"]
    st = TailStats()
    _chk(L.kslam_tail_finish_prepare(C.byref(params), C.byref(reads.view), pov, len(ov),
                                     _p(read_pairs) if len(read_pairs) else None, len(read_pairs),
                                     _p(pairs) if len(pairs) else None, len(pairs), int(sort_groups), C.byref(st)))
    return st


def release_buffers():
    lib().kslam_tail_release_buffers()


def sam_records(params, reads, index, overlaps, cigar_pool, read_pairs, pairs):
    """kslam_sam_records on the output of tail_pairs -> SAM records as bytes"""
    L = lib()
    ov, pov = _ov(overlaps)
    pool = np.ascontiguousarray(cigar_pool, dtype=np.uint32)
    rp = np.ascontiguousarray(read_pairs, dtype=READ_PAIR_DT)
    pr = np.array(pairs, dtype=PAIRED_OVERLAP_DT)  # sorted in place by the library
    txt, n, st = _vp(), _u64(), TailStats()
    _chk(L.kslam_sam_records(C.byref(params), C.byref(reads.view), C.byref(index.view), pov, len(ov),
                             _p(pool) if len(pool) else None, len(pool),
                             _p(rp) if len(rp) else None, len(rp), _p(pr) if len(pr) else None, len(pr),
                             C.byref(txt), C.byref(n), C.byref(st)))
    return _text(txt, n)


def sam_header(index, command_line=b""):
    L = lib()
    txt, n = _vp(), _u64()
    _chk(L.kslam_sam_header(C.byref(index.view), command_line, C.byref(txt), C.byref(n)))
    return _text(txt, n)


class SamWriter:
    """kslam_sam_writer: a background thread that writes SAM text to `fd` in order while the next batch is formatted"""

    def __init__(self, fd):
        L = lib()
        self._h = _vp()
        _chk(L.kslam_sam_writer_open(fd, C.byref(self._h)))
        self.callback = C.cast(L.kslam_write_queued, WRITE_FN)

    def write(self, data):
        if lib().kslam_write_queued(self._h, data, len(data)) != 0:
            raise KslamError(1, "the SAM writer reported a failed write")

    def close(self):
        """-> (bytes written, seconds the thread spent in write())"""
        if self._h is None:
            return 0, 0.0
        n, sec = _u64(), C.c_double()
        h, self._h = self._h, None
        _chk(lib().kslam_sam_writer_close(h, C.byref(n), C.byref(sec)))
        return int(n.value), float(sec.value)
