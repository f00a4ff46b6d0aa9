"""ctypes plumbing over the taxonomy entry points of libkslam_hip.so (include/kslam_taxonomy.h).

taxDB text -> tree; LCA / lineage queries; per-read classification of a batch of read pairs and the
end-of-run abbreviated report.  Host-only calls: they work without a GPU.
"""
import ctypes as C

import numpy as np

from . import KslamError, lib as _base_lib
from .tail import IndexView, PAIRED_OVERLAP_DT, READ_PAIR_DT, ReadsView, TailParams

# every symbol include/kslam_taxonomy.h declares
EXPORTS = ["kslam_taxdb_parse", "kslam_taxdb_free", "kslam_taxdb_size", "kslam_taxdb_lca",
           "kslam_taxdb_parent", "kslam_taxdb_at_rank", "kslam_taxdb_is_below",
           "kslam_taxdb_is_subspecies", "kslam_taxdb_text", "kslam_taxdb_dense", "kslam_taxdb_node", "kslam_tail_classify",
           "kslam_taxonomy_summary", "kslam_taxreport_create", "kslam_taxreport_free", "kslam_taxreport_add_batch",
           "kslam_taxreport_xml"]

_vp, _u32, _u64 = C.c_void_p, C.c_uint32, C.c_uint64
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = _base_lib()
        P = C.POINTER
        L.kslam_taxdb_parse.argtypes = [C.c_char_p, _u64, P(_vp)]
        L.kslam_taxdb_free.argtypes = [_vp]
        L.kslam_taxdb_free.restype = None
        L.kslam_taxdb_size.argtypes = [_vp]
        L.kslam_taxdb_size.restype = _u64
        L.kslam_taxdb_lca.argtypes = [_vp, _vp, _u64]
        L.kslam_taxdb_lca.restype = _u32
        L.kslam_taxdb_parent.argtypes = [_vp, _u32]
        L.kslam_taxdb_parent.restype = _u32
        L.kslam_taxdb_at_rank.argtypes = [_vp, _u32, C.c_char_p]
        L.kslam_taxdb_at_rank.restype = _u32
        L.kslam_taxdb_is_below.argtypes = [_vp, _u32, _u32]
        L.kslam_taxdb_is_below.restype = C.c_int32
        L.kslam_taxdb_is_subspecies.argtypes = [_vp, _u32]
        L.kslam_taxdb_is_subspecies.restype = C.c_int32
        L.kslam_taxdb_text.argtypes = [_vp, _u32, C.c_int, P(_vp), P(_u64)]
        L.kslam_tail_classify.argtypes = [P(TailParams), P(ReadsView), P(IndexView), _vp, _vp, _u64, _vp, _u64,
                                          _vp, P(_vp), P(_u64)]
        L.kslam_taxonomy_summary.argtypes = [_vp, _vp, _u64, _u64, P(_vp), P(_u64)]
        L.kslam_taxreport_create.argtypes = [P(_vp)]
        L.kslam_taxreport_free.argtypes = [_vp]
        L.kslam_taxreport_free.restype = None
        L.kslam_taxreport_add_batch.argtypes = [_vp, P(ReadsView), P(IndexView), _vp, _u64, _vp, _u64, _vp]
        L.kslam_taxreport_xml.argtypes = [_vp, P(IndexView), P(GeneExtras), _vp, _u64, P(_vp), P(_u64)]
        L.kslam_tail_last_error.restype = C.c_char_p
        _lib = L
    return _lib


class GeneExtras(C.Structure):
    """kslam_gene_extras: the gene fields the XML report prints beyond the index view"""
    _fields_ = [("gene_locus_tag", _vp), ("gene_locus_tag_off", _vp), ("gene_reference", _vp),
                ("gene_reference_off", _vp), ("gene_id", _vp)]

    @classmethod
    def from_lists(cls, locus_tags, references, gene_ids):
        from .tail import _column
        lt, ltoff = _column(locus_tags)
        rf, rfoff = _column(references)
        ids = np.ascontiguousarray(gene_ids, dtype=np.uint32)
        x = cls(lt.ctypes.data, ltoff.ctypes.data, rf.ctypes.data, rfoff.ctypes.data, ids.ctypes.data)
        x._keep = (lt, ltoff, rf, rfoff, ids)
        return x


def _chk(st):
    if st != 0:
        raise KslamError(st, lib().kslam_tail_last_error().decode())


def _take_text(ptr, n):
    # (ctypes.string_at takes a C int: the XML report of a 100 M-pair run is longer than that)
    out = bytes((C.c_char * n.value).from_address(ptr.value)) if n.value else b""
    if ptr.value:
        lib().kslam_free(ptr)
    return out


class TaxDB:
    NAME, RANK, LINEAGE = 0, 1, 2

    def __init__(self, text):
        h = _vp()
        _chk(lib().kslam_taxdb_parse(text, len(text), C.byref(h)))
        self._h = h

    def __len__(self):
        return int(lib().kslam_taxdb_size(self._h))

    def lca(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.uint32)
        return int(lib().kslam_taxdb_lca(self._h, a.ctypes.data if len(a) else None, len(a)))

    def parent(self, i):
        return int(lib().kslam_taxdb_parent(self._h, i))

    def at_rank(self, i, rank):
        return int(lib().kslam_taxdb_at_rank(self._h, i, rank))

    def is_below(self, upper, lower):
        return int(lib().kslam_taxdb_is_below(self._h, upper, lower))

    def is_subspecies(self, i):
        return int(lib().kslam_taxdb_is_subspecies(self._h, i))

    def text(self, i, which):
        p, n = _vp(), _u64()
        _chk(lib().kslam_taxdb_text(self._h, i, which, C.byref(p), C.byref(n)))
        return _take_text(p, n)

    def classify(self, params, reads, index, read_pairs, pairs, per_read=True):
        """kslam_tail_classify -> (taxonomy id per read pair, per-read text or None)"""
        rp = np.ascontiguousarray(read_pairs, dtype=READ_PAIR_DT)
        pr = np.ascontiguousarray(pairs, dtype=PAIRED_OVERLAP_DT)
        ids = np.zeros(len(rp), dtype=np.uint32)
        p, n = _vp(), _u64()
        _chk(lib().kslam_tail_classify(C.byref(params), C.byref(reads.view), C.byref(index.view), self._h,
                                       rp.ctypes.data if len(rp) else None, len(rp),
                                       pr.ctypes.data if len(pr) else None, len(pr), ids.ctypes.data,
                                       C.byref(p) if per_read else None, C.byref(n) if per_read else None))
        return ids, (_take_text(p, n) if per_read else None)

    def summary(self, tax_ids, num_reads):
        a = np.ascontiguousarray(tax_ids, dtype=np.uint32)
        p, n = _vp(), _u64()
        _chk(lib().kslam_taxonomy_summary(self._h, a.ctypes.data if len(a) else None, len(a), num_reads,
                                          C.byref(p), C.byref(n)))
        return _take_text(p, n)

    def close(self):
        if self._h is not None and self._h.value:
            lib().kslam_taxdb_free(self._h)
            self._h = None

    def report_xml(self, report, index, extras, num_reads):
        """kslam_taxreport_xml -> the text of the XML report"""
        p, n = _vp(), _u64()
        _chk(lib().kslam_taxreport_xml(report._h, C.byref(index.view), C.byref(extras) if extras is not None else None,
                                       self._h, num_reads, C.byref(p), C.byref(n)))
        return _take_text(p, n)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Report:
    """kslam_taxreport: one IdentifiedTaxonomy per read pair, collected batch by batch"""

    def __init__(self):
        h = _vp()
        _chk(lib().kslam_taxreport_create(C.byref(h)))
        self._h = h

    def add_batch(self, reads, index, read_pairs, pairs, tax_ids):
        rp = np.ascontiguousarray(read_pairs, dtype=READ_PAIR_DT)
        pr = np.ascontiguousarray(pairs, dtype=PAIRED_OVERLAP_DT)
        ids = np.ascontiguousarray(tax_ids, dtype=np.uint32)
        _chk(lib().kslam_taxreport_add_batch(self._h, C.byref(reads.view), C.byref(index.view),
                                             rp.ctypes.data if len(rp) else None, len(rp),
                                             pr.ctypes.data if len(pr) else None, len(pr),
                                             ids.ctypes.data if len(ids) else None))

    def close(self):
        if self._h is not None and self._h.value:
            lib().kslam_taxreport_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
