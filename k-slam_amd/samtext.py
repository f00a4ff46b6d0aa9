"""ctypes plumbing for include/kslam_samtext.h: the SAM records and the <out>_PerRead lines of a batch written on the GPU."""
import ctypes as C

import numpy as np

from . import lib as _base_lib

EXPORTS = ["kslam_set_sam_annotations", "kslam_set_sam_text", "kslam_load_read_ids", "kslam_sam_text"]
_ready = False


def lib():
    global _ready
    L = _base_lib()
    if not _ready:
        vp, u64, u32, P = C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER
        L.kslam_set_sam_annotations.argtypes = [vp, vp, vp]
        L.kslam_set_sam_text.argtypes = [vp, C.c_int, C.c_int, u32, C.c_int]
        L.kslam_load_read_ids.argtypes = [vp, vp, vp]
        L.kslam_sam_text.argtypes = [vp, C.c_int, u32, C.c_int, P(vp), P(u64), P(vp), P(u64), P(vp), P(u64)]
        _ready = True
    return L


def set_annotations(ctx, index, taxdb=None):
    """index: a kslam_amd.tail.Index / IndexArrays / kslam_amd.db.Database (anything with .view); taxdb: kslam_amd.taxonomy.TaxDB"""
    ctx._chk(lib().kslam_set_sam_annotations(ctx._h, C.addressof(index.view), taxdb._h if taxdb is not None else None))


def set_sam_text(ctx, want_sam=True, want_per_read=False, num_alignments=10, sam_xa=False):
    ctx._chk(lib().kslam_set_sam_text(ctx._h, int(want_sam), int(want_per_read), num_alignments, int(sam_xa)))


def load_read_ids(ctx, ids):
    """ids: list of bytes, one per read of the loaded batch"""
    off = np.zeros(len(ids) + 1, dtype=np.uint64)
    if ids:
        np.cumsum([len(x) for x in ids], out=off[1:])
    cat = np.frombuffer(b"".join(ids) + b"\0", dtype=np.uint8)
    ctx._chk(lib().kslam_load_read_ids(ctx._h, cat.ctypes.data, off.ctypes.data))


def sam_text(ctx, paired=True, num_alignments=10, sam_xa=False, want_sam=True, want_per_read=False):
    """kslam_sam_text on the context's resident batch -> (sam bytes or None, per-read bytes or None, tax ids or None)"""
    L = lib()
    t, p, x = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nt, np_, nx = C.c_uint64(), C.c_uint64(), C.c_uint64()
    ctx._chk(L.kslam_sam_text(ctx._h, int(paired), num_alignments, int(sam_xa),
                              C.byref(t) if want_sam else None, C.byref(nt) if want_sam else None,
                              C.byref(p) if want_per_read else None, C.byref(np_) if want_per_read else None,
                              C.byref(x) if want_per_read else None, C.byref(nx) if want_per_read else None))
    sam = C.string_at(t.value, nt.value) if want_sam else None
    per = C.string_at(p.value, np_.value) if want_per_read else None
    tax = np.frombuffer((C.c_char * (4 * nx.value)).from_address(x.value), dtype=np.uint32).copy() if want_per_read and nx.value else \
        (np.zeros(0, dtype=np.uint32) if want_per_read else None)
    for q in (t, p, x):
        if q.value:
            L.kslam_free_pinned(ctx._h, q)
    return sam, per, tax


def sam_text_to_files(ctx, writer, per_read_fd, paired=True, num_alignments=10, sam_xa=False, want_per_read=True):
    """kslam_sam_text with nothing copied through Python: the SAM block joins `writer`'s queue (a kslam_amd.tail.SamWriter;
    kslam_sam_writer_enqueue with kslam_free_pinned itself as the release callback), the per-read lines are written to
    per_read_fd straight from the page-locked block.  -> (sam bytes, per-read bytes, taxonomy ids as a numpy copy)"""
    import os
    L = lib()
    L.kslam_sam_writer_enqueue.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    t, p, x = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nt, np_, nx = C.c_uint64(), C.c_uint64(), C.c_uint64()
    ctx._chk(L.kslam_sam_text(ctx._h, int(paired), num_alignments, int(sam_xa), C.byref(t), C.byref(nt),
                              C.byref(p) if want_per_read else None, C.byref(np_) if want_per_read else None,
                              C.byref(x) if want_per_read else None, C.byref(nx) if want_per_read else None))
    release = C.cast(L.kslam_free_pinned, C.c_void_p)          # void (*)(void *user = ctx, void *data)
    if L.kslam_sam_writer_enqueue(writer._h, t, nt.value, release, ctx._h) != 0:
        raise RuntimeError(L.kslam_tail_last_error().decode())
    tax = None
    if want_per_read:
        if np_.value:
            view = memoryview((C.c_char * np_.value).from_address(p.value))
            done = 0
            while done < np_.value:
                done += os.write(per_read_fd, view[done:])
        tax = np.frombuffer((C.c_char * (4 * nx.value)).from_address(x.value), dtype=np.uint32).copy() if nx.value else np.zeros(0, np.uint32)
        for q in (p, x):
            if q.value:
                L.kslam_free_pinned(ctx._h, q)
    return int(nt.value), int(np_.value), tax
