"""ctypes plumbing over the database entry points of libkslam_hip.so (include/kslam_db.h).

<db>/database (Boost.Serialization text archive of a GenbankIndex) -> columns, and back.  Host-only
calls: they work without a GPU.  The columns plug straight into the other stages:
`Database.index_view` is the view the host tail takes, `Database.entries()` / `set_index_on(ctx)`
feed kslam_set_index.
"""
import ctypes as C

import numpy as np

from . import KslamError, lib as _base_lib
from .tail import IndexView, _column, _p

# every symbol include/kslam_db.h declares
EXPORTS = ["kslam_db_parse", "kslam_db_load", "kslam_db_free", "kslam_db_view", "kslam_db_library_version",
           "kslam_db_variant", "kslam_db_entry_bases", "kslam_db_entry_lengths", "kslam_db_write"]

_vp, _u32, _u64 = C.c_void_p, C.c_uint32, C.c_uint64


class DbColumns(C.Structure):
    _fields_ = [("index", IndexView), ("genbank_id", _vp), ("is_plasmid", _vp), ("is_16s", _vp),
                ("gene_locus_tag", _vp), ("gene_locus_tag_off", _vp), ("gene_reference", _vp),
                ("gene_reference_off", _vp), ("gene_id", _vp), ("gene_complement", _vp)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = _base_lib()
        P = C.POINTER
        L.kslam_db_parse.argtypes = [C.c_char_p, _u64, C.c_int, P(_vp)]
        L.kslam_db_load.argtypes = [C.c_char_p, C.c_int, P(_vp)]
        L.kslam_db_free.argtypes = [_vp]
        L.kslam_db_free.restype = None
        L.kslam_db_view.argtypes = [_vp]
        L.kslam_db_view.restype = P(DbColumns)
        L.kslam_db_library_version.argtypes = [_vp]
        L.kslam_db_library_version.restype = _u32
        L.kslam_db_variant.argtypes = [_vp]
        L.kslam_db_variant.restype = _u32
        L.kslam_db_entry_bases.argtypes = [_vp]
        L.kslam_db_entry_bases.restype = _vp
        L.kslam_db_entry_lengths.argtypes = [_vp]
        L.kslam_db_entry_lengths.restype = _vp
        L.kslam_db_write.argtypes = [C.c_char_p, P(DbColumns), _u32]
        L.kslam_tail_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _chk(st):
    if st != 0:
        raise KslamError(st, lib().kslam_tail_last_error().decode())


def _arr(ptr, n, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(int(n),))


def _strings(text_ptr, off_ptr, n):
    off = _arr(off_ptr, n + 1, np.uint64)
    total = int(off[-1]) if n else 0
    text = _arr(text_ptr, total, np.uint8)
    return text, off


class Database:
    """A parsed <db>/database.  All arrays are views into memory the library owns (valid until
    close())."""

    def __init__(self, handle):
        self._h = handle
        v = lib().kslam_db_view(handle).contents
        self.columns = v
        self.index_view = v.index
        self.view = v.index            # a Database goes wherever kslam_amd.tail takes an Index / IndexArrays
        iv = v.index
        n, g = int(iv.n_entries), int(iv.n_genes)
        self.n_entries, self.n_genes = n, g
        self.bases, self.bases_off = _strings(iv.bases, iv.bases_off, n)
        self.locus_tag, self.locus_tag_off = _strings(iv.locus_tag, iv.locus_tag_off, n)
        self.taxonomy_id = _arr(iv.taxonomy_id, n, np.uint32)
        self.genbank_id = _arr(v.genbank_id, n, np.uint32)
        self.is_plasmid = _arr(v.is_plasmid, n, np.uint8)
        self.is_16s = _arr(v.is_16s, n, np.uint8)
        self.gene_first = _arr(iv.gene_first, n + 1, np.uint64)
        self.gene_start = _arr(iv.gene_start, g, np.int32)
        self.gene_stop = _arr(iv.gene_stop, g, np.int32)
        self.gene_id = _arr(v.gene_id, g, np.uint32)
        self.gene_complement = _arr(v.gene_complement, g, np.uint8)
        self.gene_name = _strings(iv.gene_name, iv.gene_name_off, g)
        self.protein_id = _strings(iv.protein_id, iv.protein_id_off, g)
        self.product = _strings(iv.product, iv.product_off, g)
        self.gene_locus_tag = _strings(v.gene_locus_tag, v.gene_locus_tag_off, g)
        self.gene_reference = _strings(v.gene_reference, v.gene_reference_off, g)
        self.library_version = int(lib().kslam_db_library_version(handle))
        self.variant = int(lib().kslam_db_variant(handle))

    def gene_extras(self):
        """kslam_gene_extras over this database's columns (for kslam_amd.taxonomy's XML report)"""
        from .taxonomy import GeneExtras
        c = self.columns
        x = GeneExtras(c.gene_locus_tag, c.gene_locus_tag_off, c.gene_reference, c.gene_reference_off, c.gene_id)
        x._keep = self
        return x

    @classmethod
    def parse(cls, text, threads=0):
        h = _vp()
        _chk(lib().kslam_db_parse(text, len(text), threads, C.byref(h)))
        return cls(h)

    @classmethod
    def load(cls, path, threads=0):
        h = _vp()
        _chk(lib().kslam_db_load(str(path).encode(), threads, C.byref(h)))
        return cls(h)

    def close(self):
        if self._h:
            lib().kslam_db_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @staticmethod
    def _item(col, i):
        text, off = col
        return text[int(off[i]):int(off[i + 1])].tobytes()

    def entry(self, i):
        """Entry i as a dict shaped like the reference's GenbankEntry."""
        genes = []
        for g in range(int(self.gene_first[i]), int(self.gene_first[i + 1])):
            genes.append({"geneName": self._item(self.gene_name, g), "locusTag": self._item(self.gene_locus_tag, g),
                          "proteinID": self._item(self.protein_id, g), "product": self._item(self.product, g),
                          "referenceSequence": self._item(self.gene_reference, g), "geneID": int(self.gene_id[g]),
                          "start": int(np.uint32(self.gene_start[g])), "stop": int(np.uint32(self.gene_stop[g])),
                          "complement": bool(self.gene_complement[g])})
        return {"bases": self._item((self.bases, self.bases_off), i), "taxonomyID": int(self.taxonomy_id[i]),
                "genbankID": int(self.genbank_id[i]), "isPlasmid": bool(self.is_plasmid[i]),
                "is16S": bool(self.is_16s[i]), "locusTag": self._item((self.locus_tag, self.locus_tag_off), i),
                "genes": genes}

    def entries(self):
        """The genome sequences as a list of bytes (what Context.set_index takes)."""
        return [self._item((self.bases, self.bases_off), i) for i in range(self.n_entries)]

    def entry_pointers(self):
        """(char **bases, uint64 *lengths) as the C ABI's kslam_set_index takes them; no copies."""
        return lib().kslam_db_entry_bases(self._h), lib().kslam_db_entry_lengths(self._h)


def write(path, entries, library_version=17):
    """entries: list of dicts shaped like Database.entry() (missing keys default to empty / 0).
    Writes the archive the reference's writeIndexToBoostSerial would."""
    n = len(entries)
    genes = [g for e in entries for g in e.get("genes", [])]
    first = np.zeros(n + 1, dtype=np.uint64)
    for i, e in enumerate(entries):
        first[i + 1] = first[i] + len(e.get("genes", []))
    cols = {k: _column([e.get(k, b"") for e in entries]) for k in ("bases", "locusTag")}
    gcols = {k: _column([g.get(k, b"") for g in genes])
             for k in ("geneName", "locusTag", "proteinID", "product", "referenceSequence")}
    u32 = lambda xs: np.ascontiguousarray(list(xs) + [0], dtype=np.uint32)   # noqa: E731 (never empty)
    u8 = lambda xs: np.ascontiguousarray(list(xs) + [0], dtype=np.uint8)     # noqa: E731
    tax, gid = u32(e.get("taxonomyID", 0) for e in entries), u32(e.get("genbankID", 0) for e in entries)
    pl, s16 = u8(e.get("isPlasmid", False) for e in entries), u8(e.get("is16S", False) for e in entries)
    gs, ge = u32(g.get("start", 0) for g in genes), u32(g.get("stop", 0) for g in genes)
    ggid, gc = u32(g.get("geneID", 0) for g in genes), u8(g.get("complement", False) for g in genes)
    c = DbColumns()
    iv = c.index
    iv.n_entries = n
    iv.bases, iv.bases_off = _p(cols["bases"][0]), _p(cols["bases"][1])
    iv.locus_tag, iv.locus_tag_off = _p(cols["locusTag"][0]), _p(cols["locusTag"][1])
    iv.taxonomy_id = _p(tax)
    iv.n_genes = len(genes)
    iv.gene_first, iv.gene_start, iv.gene_stop = _p(first), _p(gs), _p(ge)
    iv.gene_name, iv.gene_name_off = _p(gcols["geneName"][0]), _p(gcols["geneName"][1])
    iv.protein_id, iv.protein_id_off = _p(gcols["proteinID"][0]), _p(gcols["proteinID"][1])
    iv.product, iv.product_off = _p(gcols["product"][0]), _p(gcols["product"][1])
    c.genbank_id, c.is_plasmid, c.is_16s = _p(gid), _p(pl), _p(s16)
    c.gene_locus_tag, c.gene_locus_tag_off = _p(gcols["locusTag"][0]), _p(gcols["locusTag"][1])
    c.gene_reference, c.gene_reference_off = _p(gcols["referenceSequence"][0]), _p(gcols["referenceSequence"][1])
    c.gene_id, c.gene_complement = _p(ggid), _p(gc)
    _chk(lib().kslam_db_write(str(path).encode(), C.byref(c), library_version))
