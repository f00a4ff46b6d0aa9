"""ctypes plumbing over the FASTQ ingest entry points of libkslam_hip.so (include/kslam_fastq.h).

File bytes -> read columns (bases / quality / identifiers + offsets), the layout the device
upload and the host tail take.  Host-only calls: they work without a GPU.
"""
import ctypes as C

import numpy as np

from . import KslamError, lib as _base_lib
from .tail import ReadsView

# every symbol include/kslam_fastq.h declares
EXPORTS = ["kslam_fastq_parse", "kslam_fastq_parse_pair", "kslam_reads_free", "kslam_fastq_index_pair",
           "kslam_fastq_layout_free", "kslam_fastq_batch_end"]

_vp, _u64 = C.c_void_p, C.c_uint64


class ReadsColumns(C.Structure):
    """kslam_reads_columns: field for field a kslam_reads_view with library-owned arrays."""
    _fields_ = ReadsView._fields_


class Layout(C.Structure):
    """kslam_fastq_layout"""
    _fields_ = [("n_reads", _u64), ("bases_at", _vp), ("quality_at", _vp)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = _base_lib()
        P = C.POINTER
        L.kslam_fastq_parse.argtypes = [C.c_char_p, _u64, _u64, C.c_int, C.c_int, P(ReadsColumns), P(_u64)]
        L.kslam_fastq_parse_pair.argtypes = [C.c_char_p, _u64, C.c_char_p, _u64, _u64, C.c_int, C.c_int,
                                             P(ReadsColumns), P(_u64), P(_u64)]
        L.kslam_fastq_index_pair.argtypes = [_vp, _u64, _vp, _u64, _u64, C.c_int, C.c_int, P(ReadsColumns),
                                             P(Layout), P(_u64), P(_u64)]
        L.kslam_fastq_layout_free.argtypes = [P(Layout)]
        L.kslam_fastq_batch_end.argtypes = [_vp, _u64, _u64, C.c_int, C.c_int, P(_u64), P(C.c_int)]
        L.kslam_fastq_layout_free.restype = None
        L.kslam_reads_free.argtypes = [P(ReadsColumns)]
        L.kslam_reads_free.restype = None
        L.kslam_tail_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _chk(st):
    if st != 0:
        raise KslamError(st, lib().kslam_tail_last_error().decode())


class Batch:
    """A parsed batch.  `.view` can be passed wherever a tail `Reads` object goes; `.bases`,
    `.quality`, `.ids` are lists of bytes (copied out on first use)."""

    def __init__(self, cols):
        self._cols = cols
        self.n_reads = int(cols.n_reads)
        self.view = ReadsView.from_buffer(cols)
        self._lists = None

    def _column(self, text, off):
        n = self.n_reads
        o = np.frombuffer((C.c_char * (8 * (n + 1))).from_address(off), dtype=np.uint64)
        total = int(o[n])
        raw = C.string_at(text, total) if total else b""
        return [raw[int(o[i]):int(o[i + 1])] for i in range(n)]

    def columns(self):
        if self._lists is None:
            c = self._cols
            self._lists = (self._column(c.bases, c.bases_off), self._column(c.quality, c.quality_off),
                           self._column(c.ids, c.ids_off))
        return self._lists

    @property
    def bases(self):
        return self.columns()[0]

    @property
    def quality(self):
        return self.columns()[1]

    @property
    def ids(self):
        return self.columns()[2]

    def bases_array(self):
        """(uint8 view of the concatenated bases, uint64 offsets): what kslam_load_reads takes"""
        c, n = self._cols, self.n_reads
        off = np.frombuffer((C.c_char * (8 * (n + 1))).from_address(c.bases_off), dtype=np.uint64)
        cat = np.frombuffer((C.c_char * (int(off[n]) + 1)).from_address(c.bases), dtype=np.uint8)
        return cat, off

    def close(self):
        if self._cols is not None:
            lib().kslam_reads_free(C.byref(self._cols))
            self._cols = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def parse(text, max_reads=0, at_eof=True, threads=0):
    """kslam_fastq_parse -> (Batch, bytes consumed)"""
    cols, used = ReadsColumns(), _u64()
    _chk(lib().kslam_fastq_parse(text, len(text), max_reads, int(at_eof), threads, C.byref(cols),
                                 C.byref(used)))
    return Batch(cols), int(used.value)


class IndexedBatch(Batch):
    """kslam_fastq_index_pair: identifiers and offsets on the host, `layout` (where the bases / quality lines
    lie in [r1 | r2]) for kslam_submit_batch_fastq; no bases / quality columns."""

    def __init__(self, cols, layout):
        super().__init__(cols)
        self.layout = layout

    def close(self):
        if self.layout is not None:
            lib().kslam_fastq_layout_free(C.byref(self.layout))
            self.layout = None
        super().close()


def index_pair(r1_ptr, len1, r2_ptr, len2, max_pairs=0, at_eof=True, threads=0):
    """kslam_fastq_index_pair on two text ADDRESSES (e.g. of kslam_amd.HostBuffer) -> (IndexedBatch, consumed1, consumed2)"""
    cols, lay, u1, u2 = ReadsColumns(), Layout(), _u64(), _u64()
    _chk(lib().kslam_fastq_index_pair(r1_ptr, len1, r2_ptr, len2, max_pairs, int(at_eof), threads, C.byref(cols),
                                      C.byref(lay), C.byref(u1), C.byref(u2)))
    return IndexedBatch(cols, lay), int(u1.value), int(u2.value)


def parse_pair(r1, r2, max_pairs=0, at_eof=True, threads=0):
    """kslam_fastq_parse_pair -> (Batch laid out [R1 block | R2 block], consumed1, consumed2)"""
    cols, u1, u2 = ReadsColumns(), _u64(), _u64()
    _chk(lib().kslam_fastq_parse_pair(r1, len(r1), r2, len(r2), max_pairs, int(at_eof), threads,
                                      C.byref(cols), C.byref(u1), C.byref(u2)))
    return Batch(cols), int(u1.value), int(u2.value)


def batch_end(text_ptr, length, max_records, at_eof=True, threads=0):
    """kslam_fastq_batch_end on a text ADDRESS -> (end, complete): the byte after record `max_records`"""
    end, done = _u64(), C.c_int()
    _chk(lib().kslam_fastq_batch_end(text_ptr, length, max_records, int(at_eof), threads, C.byref(end), C.byref(done)))
    return int(end.value), bool(done.value)
