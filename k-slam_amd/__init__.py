"""kslam_amd -- Python plumbing over the C ABI of libkslam_hip.so (include/kslam.h).

The product is the HIP library; this module only loads it with ctypes so that
tests/ and bench.py can drive it.  There is no CPU fallback: if the library is
missing it raises, and on a box without a HIP device every call fails with
KSLAM_ERR_NO_DEVICE.

The directory is named ``k-slam_amd`` (not importable by name); load it with
``tests/conftest.py``'s helper or ``importlib`` (see ``load_package`` in
``__graft_entry__.py``).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KSLAM_LIB: another build of the same ABI (tools/kprof.sh loads the measurement-only `make ABLATE=1` library)
LIB_PATH = os.environ.get("KSLAM_LIB") or os.path.join(_HERE, "libkslam_hip.so")

KMER_DT = np.dtype([("kmer", "<u8"), ("meta", "<u4"), ("offset", "<u4")])
OVERLAP_TEMP_DT = np.dtype([("read", "<u4"), ("entry", "<u4"), ("rel", "<i4"),
                            ("revcomp", "u1"), ("pad", "u1", (3,))])
OVERLAP_DT = np.dtype([("read", "<u4"), ("entry", "<u4"), ("rel", "<i4"),
                       ("revcomp", "u1"), ("pad", "u1"), ("score", "<u2"),
                       ("ref_begin", "<i4"), ("ref_end", "<i4"),
                       ("query_begin", "<i4"), ("query_end", "<i4"),
                       ("cigar_len", "<u4"), ("pad2", "<u4"), ("cigar_off", "<u8")])
assert OVERLAP_DT.itemsize == 48
PAIRED_OVERLAP_DT = np.dtype([("combined_score", "<u4"), ("entry", "<u4"), ("ref_start", "<i4"), ("ref_end", "<i4"),
                              ("insert_size", "<u4"), ("r1", "<u4"), ("r2", "<u4"), ("pad", "<u4")])
READ_PAIR_DT = np.dtype([("r1_read", "<u4"), ("r2_read", "<u4"), ("first", "<u8"), ("count", "<u8")])
ROW_DETAIL_DT = np.dtype([("logp", "<f8"), ("md_off", "<u8"), ("md_len", "<u4"), ("nm", "<u4"),
                          ("flags", "<u4"), ("pad", "<u4")])
assert ROW_DETAIL_DT.itemsize == 32

STATUS = {0: "OK", 1: "ERR_ARG", 2: "ERR_NO_DEVICE", 3: "ERR_OOM", 4: "ERR_UNSUPPORTED",
          5: "ERR_STATE", 6: "ERR_INTERNAL"}

# every symbol include/kslam.h declares
EXPORTS = ["kslam_abi_version", "kslam_index_build_stats", "kslam_version", "kslam_check_std_sort", "kslam_create", "kslam_destroy", "kslam_last_error", "kslam_reload_tuning", "kslam_ctx_device", "kslam_create_sibling", "kslam_adopt_results_device",
           "kslam_set_index", "kslam_set_index_device", "kslam_align_batch", "kslam_free_batch",
           "kslam_align_batch_async", "kslam_wait_batch", "kslam_load_qualities", "kslam_load_qualities_device",
           "kslam_row_details", "kslam_take_row_details", "kslam_free_pinned", "kslam_submit_batch",
           "kslam_submit_batch_columns", "kslam_submit_batch_fastq", "kslam_submit_batch_fastq_text", "kslam_collect_batch", "kslam_release_batch", "kslam_host_alloc",
           "kslam_host_free", "kslam_pair_screen", "kslam_pair_phase_a", "kslam_pair_phase_b", "kslam_pseudo_merged", "kslam_pseudo_route", "kslam_pseudo_owned", "kslam_pseudo_return", "kslam_pair_screen_overlaps", "kslam_take_pairs", "kslam_set_pairing", "kslam_debug_wave_sort", "kslam_row_details_of_pairs",
           "kslam_load_reads", "kslam_load_reads_device", "kslam_align_resident",
           "kslam_fetch_results", "kslam_take_results", "kslam_copy_results_device", "kslam_get_timings",
           "kslam_extract_kmers", "kslam_sort_kmers", "kslam_find_overlaps", "kslam_free",
           "kslam_selftest_sort", "kslam_merge_shards_device", "kslam_shard_counts_device",
           "kslam_export_shard_device", "kslam_multi_create", "kslam_multi_destroy",
           "kslam_multi_last_error", "kslam_multi_set_index", "kslam_multi_align_batch", "kslam_multi_free_batch"]


class Params(C.Structure):
    """kslam_params: the scoring globals of reference src/Globals.h:27-36."""
    _fields_ = [("match", C.c_uint32), ("mismatch", C.c_uint32), ("gap_open", C.c_uint32),
                ("gap_extend", C.c_uint32), ("score_threshold", C.c_uint32),
                ("report_cigar", C.c_int32), ("device", C.c_int32),
                ("max_kmers_per_chunk", C.c_uint32)]


class PairStats(C.Structure):
    """kslam_pair_stats"""
    _fields_ = [("n_overlaps_screened", C.c_uint64), ("n_paired_initial", C.c_uint64), ("n_insert_sizes", C.c_uint64),
                ("n_read_pairs", C.c_uint64), ("n_pairs", C.c_uint64), ("max_insert_size", C.c_uint32),
                ("stages_done", C.c_uint32)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class BatchResult(C.Structure):
    """kslam_batch_result"""
    _fields_ = [("overlaps", C.c_void_p), ("n_overlaps", C.c_uint64), ("cigar_pool", C.c_void_p),
                ("n_cigar", C.c_uint64), ("details", C.c_void_p), ("md_pool", C.c_void_p), ("n_md", C.c_uint64),
                ("read_pairs", C.c_void_p), ("n_read_pairs", C.c_uint64), ("pairs", C.c_void_p),
                ("n_pairs", C.c_uint64), ("pair_stats", PairStats),
                ("n_reads", C.c_uint64), ("reads_bases_off", C.c_void_p), ("reads_ids", C.c_void_p),
                ("reads_ids_off", C.c_void_p), ("consumed1", C.c_uint64), ("consumed2", C.c_uint64),
                ("sam_text", C.c_void_p), ("sam_text_len", C.c_uint64), ("per_read_text", C.c_void_p),
                ("per_read_len", C.c_uint64), ("tax_ids", C.c_void_p), ("text_flags", C.c_uint32), ("pad_", C.c_uint32)]


class Timings(C.Structure):
    _fields_ = [("ms_extract", C.c_float), ("ms_sort", C.c_float), ("ms_sort_scatter", C.c_float),
                ("ms_join", C.c_float), ("ms_sw", C.c_float), ("ms_cigar", C.c_float),
                ("ms_total", C.c_float), ("sort_passes", C.c_uint32),
                ("n_read_kmers", C.c_uint64), ("n_genome_kmers", C.c_uint64),
                ("n_overlaps_raw", C.c_uint64), ("n_overlaps", C.c_uint64),
                ("sw_cells", C.c_uint64), ("n_chunks", C.c_uint32),
                ("n_scatter_launches", C.c_uint32), ("n_kmers_kept", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class IndexStats(C.Structure):
    """kslam_index_stats (include/kslam.h)"""
    _fields_ = [("n_genome_kmers", C.c_uint64), ("sort_passes", C.c_uint32), ("n_entries", C.c_uint32),
                ("ms_encode_extract", C.c_float), ("ms_sort", C.c_float), ("ms_tables", C.c_float), ("ms_total", C.c_float)]


class KslamError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("kslam status %s: %s" % (STATUS.get(status, status), msg))
        self.status = status


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of libkslam_hip.so (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "-s", "clean"])
    subprocess.check_call(["make", "-C", csrc, "-s", "-j8"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run __graft_entry__.build() (hipcc) first; "
                              "there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        L.kslam_abi_version.restype = u32
        L.kslam_version.restype = C.c_char_p
        L.kslam_check_std_sort.argtypes = [C.POINTER(u64)]
        L.kslam_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
        L.kslam_destroy.argtypes = [vp]
        L.kslam_last_error.restype = C.c_char_p
        L.kslam_last_error.argtypes = [vp]
        L.kslam_reload_tuning.argtypes = [vp]
        L.kslam_ctx_device.argtypes = [vp]
        L.kslam_ctx_device.restype = C.c_int32
        L.kslam_create_sibling.argtypes = [vp, C.POINTER(vp)]
        L.kslam_adopt_results_device.argtypes = [vp, vp, u64, vp, u64]
        L.kslam_set_index.argtypes = [vp, u64, vp, vp]
        L.kslam_set_index_device.argtypes = [vp, u64, vp, vp]
        L.kslam_align_batch.argtypes = [vp, u64, vp, vp, C.POINTER(vp), C.POINTER(u64),
                                        C.POINTER(vp), C.POINTER(u64)]
        L.kslam_free_batch.argtypes = [vp, vp, vp]
        L.kslam_align_batch_async.argtypes = [vp, u64, vp, vp, C.POINTER(u64)]
        L.kslam_wait_batch.argtypes = [vp, u64, C.POINTER(vp), C.POINTER(u64), C.POINTER(vp), C.POINTER(u64)]
        L.kslam_load_qualities.argtypes = [vp, vp]
        L.kslam_load_qualities_device.argtypes = [vp, vp]
        L.kslam_row_details.argtypes = [vp, C.POINTER(u64)]
        L.kslam_row_details_of_pairs.argtypes = [vp, C.POINTER(u64)]
        L.kslam_take_row_details.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
        L.kslam_free_pinned.argtypes = [vp, vp]
        L.kslam_submit_batch.argtypes = [vp, u64, vp, vp, vp, C.POINTER(u64)]
        L.kslam_pair_screen.argtypes = [vp, C.c_int, u32, C.c_double, u32, C.POINTER(PairStats)]
        L.kslam_pair_phase_a.argtypes = [vp, C.c_int, u32, C.POINTER(vp), C.POINTER(u64)]
        L.kslam_pair_phase_b.argtypes = [vp, vp, u64, C.c_double, u32, C.POINTER(PairStats), C.POINTER(vp), C.POINTER(u64)]
        L.kslam_pseudo_merged.argtypes = [vp, vp, u64, u64, C.c_double, C.POINTER(PairStats)]
        L.kslam_pseudo_route.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u64)]
        L.kslam_pseudo_owned.argtypes = [vp, vp, u64, C.POINTER(vp)]
        L.kslam_pseudo_return.argtypes = [vp, vp, u64, C.c_double, C.POINTER(PairStats)]
        L.kslam_pair_screen_overlaps.argtypes = [vp, vp, u64, vp, u64, C.c_int, u32, C.c_double, u32, C.POINTER(PairStats)]
        L.kslam_take_pairs.argtypes = [vp, C.POINTER(vp), C.POINTER(u64), C.POINTER(vp), C.POINTER(u64)]
        L.kslam_set_pairing.argtypes = [vp, C.c_int, u32, C.c_double, u32]
        L.kslam_debug_wave_sort.argtypes = [vp, vp, vp, u64, vp]
        L.kslam_host_alloc.restype = vp
        L.kslam_host_alloc.argtypes = [u64]
        L.kslam_host_free.argtypes = [vp, u64]
        L.kslam_submit_batch_columns.argtypes = [vp, u64, vp, vp, vp, C.POINTER(u64)]
        L.kslam_submit_batch_fastq.argtypes = [vp, vp, u64, vp, u64, u64, vp, vp, vp, C.POINTER(u64)]
        L.kslam_submit_batch_fastq_text.argtypes = [vp, vp, u64, vp, u64, u64, C.c_int, C.POINTER(u64)]
        L.kslam_collect_batch.argtypes = [vp, u64, C.POINTER(BatchResult)]
        L.kslam_release_batch.argtypes = [vp, C.POINTER(BatchResult)]
        L.kslam_load_reads.argtypes = [vp, u64, vp, vp]
        L.kslam_load_reads_device.argtypes = [vp, u64, vp, vp]
        L.kslam_align_resident.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        L.kslam_fetch_results.argtypes = [vp, vp, vp]
        L.kslam_take_results.argtypes = [vp, C.POINTER(vp), C.POINTER(u64), C.POINTER(vp), C.POINTER(u64)]
        L.kslam_copy_results_device.argtypes = [vp, vp, vp]
        L.kslam_get_timings.argtypes = [vp, C.POINTER(Timings)]
        L.kslam_extract_kmers.argtypes = [vp, u64, vp, vp, C.c_int, u32, vp, u64, C.POINTER(u64)]
        L.kslam_sort_kmers.argtypes = [vp, vp, u64]
        L.kslam_find_overlaps.argtypes = [vp, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
        L.kslam_free.argtypes = [vp]
        L.kslam_selftest_sort.argtypes = [vp, u64, u32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                          C.POINTER(u64)]
        L.kslam_merge_shards_device.argtypes = [vp, u32, vp, u64, vp, vp, vp, vp]
        L.kslam_shard_counts_device.argtypes = [vp, u64, vp]
        L.kslam_export_shard_device.argtypes = [vp, u64, u64, u64, u64, u64, vp, vp, vp, vp]
        L.kslam_multi_create.argtypes = [C.POINTER(Params), vp, u32, C.POINTER(vp)]
        L.kslam_multi_destroy.argtypes = [vp]
        L.kslam_multi_last_error.restype = C.c_char_p
        L.kslam_multi_last_error.argtypes = [vp]
        L.kslam_multi_set_index.argtypes = [vp, u64, vp, vp]
        L.kslam_multi_align_batch.argtypes = [vp, u64, vp, vp, C.c_int, C.POINTER(vp), C.POINTER(u64),
                                              C.POINTER(vp), C.POINTER(u64)]
        L.kslam_multi_free_batch.argtypes = [vp, vp, vp]
        _lib = L
    return _lib


def _seq_arrays(seqs):
    n = len(seqs)
    bufs = [C.create_string_buffer(s, len(s) + 1) for s in seqs]
    ptrs = (C.c_char_p * max(n, 1))(*[C.cast(b, C.c_char_p) for b in bufs])
    return ptrs, bufs


def _concat(seqs):
    lens = np.fromiter((len(s) for s in seqs), dtype=np.uint64, count=len(seqs))
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    cat = np.frombuffer(b"".join(seqs) + b"\0", dtype=np.uint8)
    return cat, off


class Context:
    """One kslam_ctx: one process, one GPU (reference threading model: src/SLAM.h:59 is
    called from the main thread once per batch)."""

    def __init__(self, match=2, mismatch=3, gap_open=5, gap_extend=2, score_threshold=0,
                 report_cigar=True, device=0, max_kmers_per_chunk=0):
        self._L = lib()
        self._h = C.c_void_p()
        p = Params(match, mismatch, gap_open, gap_extend, score_threshold,
                   1 if report_cigar else 0, device, max_kmers_per_chunk)
        st = self._L.kslam_create(C.byref(p), C.byref(self._h))
        if st != 0:
            msg = self._L.kslam_last_error(self._h).decode() if self._h else "create failed"
            if self._h:
                self._L.kslam_destroy(self._h)
                self._h = C.c_void_p()
            raise KslamError(st, msg)

    def close(self):
        if self._h:
            self._L.kslam_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st):
        if st != 0:
            raise KslamError(st, self._L.kslam_last_error(self._h).decode())

    def sibling(self):
        """kslam_create_sibling: a second context on this device that borrows this context's index"""
        h = C.c_void_p()
        self._chk(self._L.kslam_create_sibling(self._h, C.byref(h)))
        c = Context.__new__(Context)
        c._L, c._h = self._L, h
        return c

    def adopt_results_device(self, d_overlaps, n_overlaps, d_cigars, n_cigar):
        """kslam_adopt_results_device: device arrays become this context's last result"""
        self._chk(self._L.kslam_adopt_results_device(self._h, d_overlaps, n_overlaps, d_cigars, n_cigar))

    def reload_tuning(self):
        """kslam_reload_tuning: the KSLAM_* environment switches are read at kslam_create; a test that flips one
        between two batches of the same context calls this"""
        self._chk(self._L.kslam_reload_tuning(self._h))

    # ---- const GenbankIndex& ----
    def set_index(self, entries):
        ptrs, keep = _seq_arrays(entries)
        lens = np.array([len(s) for s in entries], dtype=np.uint64)
        self._chk(self._L.kslam_set_index(self._h, len(entries), C.cast(ptrs, C.c_void_p),
                                          lens.ctypes.data))

    def set_index_device(self, n_entries, dev_ptr, host_offsets):
        off = np.ascontiguousarray(host_offsets, dtype=np.uint64)
        self._chk(self._L.kslam_set_index_device(self._h, n_entries, dev_ptr, off.ctypes.data))

    def set_index_arrays(self, cat_u8, offsets_u64):
        """kslam_set_index on a database held as ONE host uint8 array + entry offsets (no bytes object per entry)."""
        off = np.ascontiguousarray(offsets_u64, dtype=np.uint64)
        n = len(off) - 1
        base = cat_u8.ctypes.data
        ptrs = (C.c_void_p * max(n, 1))(*[base + int(off[i]) for i in range(n)])
        lens = np.ascontiguousarray(np.diff(off), dtype=np.uint64)
        self._chk(self._L.kslam_set_index(self._h, n, C.cast(ptrs, C.c_void_p), lens.ctypes.data))

    # ---- alignToDatabase ----
    def align_batch(self, reads):
        """alignToDatabase(reads, index): returns (overlaps[OVERLAP_DT], cigar_pool[u32])."""
        ptrs, keep = _seq_arrays(reads)
        lens = np.array([len(s) for s in reads], dtype=np.uint32)
        out, cig = C.c_void_p(), C.c_void_p()
        n_out, n_cig = C.c_uint64(), C.c_uint64()
        self._chk(self._L.kslam_align_batch(self._h, len(reads), C.cast(ptrs, C.c_void_p),
                                            lens.ctypes.data, C.byref(out), C.byref(n_out),
                                            C.byref(cig), C.byref(n_cig)))
        n, nc = int(n_out.value), int(n_cig.value)
        ov = np.frombuffer((C.c_char * (n * 48)).from_address(out.value), dtype=OVERLAP_DT).copy() \
            if n else np.zeros(0, dtype=OVERLAP_DT)
        cg = np.frombuffer((C.c_char * (nc * 4)).from_address(cig.value), dtype=np.uint32).copy() \
            if nc else np.zeros(0, dtype=np.uint32)
        self._L.kslam_free_batch(self._h, out, cig)
        return ov, cg

    def align_batch_pointers(self, n_reads, bases_pp, lens_p, copy=True):
        """kslam_align_batch on a ready `const char *const *` / `const uint32_t *` pair; copy=False returns
        views of the library's page-locked buffers plus the function that hands them back"""
        out, cig = C.c_void_p(), C.c_void_p()
        n_out, n_cig = C.c_uint64(), C.c_uint64()
        self._chk(self._L.kslam_align_batch(self._h, n_reads, bases_pp, lens_p, C.byref(out), C.byref(n_out),
                                            C.byref(cig), C.byref(n_cig)))
        n, nc = int(n_out.value), int(n_cig.value)
        ov = np.frombuffer((C.c_char * (n * 48)).from_address(out.value), dtype=OVERLAP_DT) \
            if n else np.zeros(0, dtype=OVERLAP_DT)
        cg = np.frombuffer((C.c_char * (nc * 4)).from_address(cig.value), dtype=np.uint32) \
            if nc else np.zeros(0, dtype=np.uint32)

        def release():
            self._L.kslam_free_batch(self._h, out, cig)
        if copy:
            ov, cg = ov.copy(), cg.copy()
            release()
            return ov, cg
        return ov, cg, release

    def submit_batch(self, reads):
        """kslam_align_batch_async on a list of bytes -> ticket"""
        ptrs, keep = _seq_arrays(reads)
        lens = np.array([len(s) for s in reads], dtype=np.uint32)
        return self.submit_batch_pointers(len(reads), C.cast(ptrs, C.c_void_p), lens.ctypes.data)

    def submit_batch_pointers(self, n_reads, bases_pp, lens_p):
        """kslam_align_batch_async on a ready `const char *const *` / `const uint32_t *` pair"""
        t = C.c_uint64()
        self._chk(self._L.kslam_align_batch_async(self._h, n_reads, bases_pp, lens_p, C.byref(t)))
        return int(t.value)

    def wait_batch(self, ticket, copy=True):
        """kslam_wait_batch -> (overlaps, cigar_pool[, release]); copy=False returns views of the library's
        page-locked buffers plus the function that hands them back"""
        po, pc, no, nc = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        self._chk(self._L.kslam_wait_batch(self._h, ticket, C.byref(po), C.byref(no), C.byref(pc), C.byref(nc)))
        ov = np.frombuffer((C.c_char * (no.value * OVERLAP_DT.itemsize)).from_address(po.value),
                           dtype=OVERLAP_DT) if no.value else np.zeros(0, dtype=OVERLAP_DT)
        cg = np.frombuffer((C.c_char * (nc.value * 4)).from_address(pc.value),
                           dtype=np.uint32) if nc.value else np.zeros(0, dtype=np.uint32)

        def release():
            self._L.kslam_free_batch(self._h, po, pc)
        if copy:
            ov, cg = ov.copy(), cg.copy()
            release()
            return ov, cg
        return ov, cg, release

    # ---- per-row details for the SAM writer ----
    def load_qualities(self, quals):
        """kslam_load_qualities from a list of bytes (same lengths as the loaded reads)"""
        cat = np.frombuffer(b"".join(quals) + b"\0", dtype=np.uint8)
        self._chk(self._L.kslam_load_qualities(self._h, cat.ctypes.data))

    def load_qualities_array(self, cat_u8):
        self._chk(self._L.kslam_load_qualities(self._h, cat_u8.ctypes.data))

    def load_qualities_device(self, dev_ptr):
        self._chk(self._L.kslam_load_qualities_device(self._h, dev_ptr))

    def row_details(self, of_pairs=False):
        """kslam_row_details (every row) / kslam_row_details_of_pairs (the rows the last pair_screen's pairs refer to)"""
        n = C.c_uint64()
        f = self._L.kslam_row_details_of_pairs if of_pairs else self._L.kslam_row_details
        self._chk(f(self._h, C.byref(n)))
        return int(n.value)

    def take_row_details(self, n_rows, copy=True):
        """kslam_take_row_details -> (details[ROW_DETAIL_DT], md_pool[uint8][, release])"""
        pd, pm, nm = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._chk(self._L.kslam_take_row_details(self._h, C.byref(pd), C.byref(pm), C.byref(nm)))
        det = np.frombuffer((C.c_char * (n_rows * 32)).from_address(pd.value), dtype=ROW_DETAIL_DT) \
            if n_rows else np.zeros(0, dtype=ROW_DETAIL_DT)
        md = np.frombuffer((C.c_char * int(nm.value)).from_address(pm.value), dtype=np.uint8) \
            if nm.value else np.zeros(0, dtype=np.uint8)

        def release():
            self._L.kslam_free_pinned(self._h, pd)
            self._L.kslam_free_pinned(self._h, pm)
        if copy:
            det, md = det.copy(), md.copy()
            release()
            return det, md
        return det, md, release

    # ---- the first half of the host tail on the device ----
    def pair_screen(self, paired=True, score_threshold=0, score_fraction=0.95, stages=3):
        """kslam_pair_screen on the last results -> stats dict"""
        st = PairStats()
        self._chk(self._L.kslam_pair_screen(self._h, int(paired), score_threshold, score_fraction, stages, C.byref(st)))
        return st.as_dict()

    def index_build_stats(self):
        """kslam_index_build_stats -> dict (device milliseconds by phase, records and passes of the one-time sort)"""
        st = IndexStats()
        self._chk(self._L.kslam_index_build_stats(self._h, C.byref(st)))
        return {k: (float(getattr(st, k)) if k.startswith("ms_") else int(getattr(st, k))) for k, _ in IndexStats._fields_}

    def pair_phase_a(self, paired=True, score_threshold=0):
        """kslam_pair_phase_a -> (device address of this shard's insert sizes (int32), their number)"""
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self._L.kslam_pair_phase_a(self._h, int(paired), score_threshold, C.byref(p), C.byref(n)))
        return p.value or 0, int(n.value)

    def pair_phase_b(self, d_all_inserts, n_all, score_fraction=0.95, stages=3):
        """kslam_pair_phase_b -> (stats dict, device address of this shard's dense alignment-pair records, their number)"""
        st, p, n = PairStats(), C.c_void_p(), C.c_uint64()
        self._chk(self._L.kslam_pair_phase_b(self._h, d_all_inserts, n_all, score_fraction, stages, C.byref(st), C.byref(p), C.byref(n)))
        return st.as_dict(), p.value or 0, int(n.value)

    def pseudo_merged(self, d_all_pairs, n_all, own_base, score_fraction=0.95):
        """kslam_pseudo_merged -> stats dict"""
        st = PairStats()
        self._chk(self._L.kslam_pseudo_merged(self._h, d_all_pairs, n_all, own_base, score_fraction, C.byref(st)))
        return st.as_dict()

    def pseudo_route(self, world):
        """kslam_pseudo_route -> (device address of this rank's 16-byte heads partitioned by entry mod world, [heads per destination])"""
        p, counts = C.c_void_p(), (C.c_uint64 * world)()
        self._chk(self._L.kslam_pseudo_route(self._h, world, C.byref(p), counts))
        return p.value or 0, [int(v) for v in counts]

    def pseudo_owned(self, d_heads, n):
        """kslam_pseudo_owned on n received heads -> device address of their n new scores (uint32)"""
        p = C.c_void_p()
        self._chk(self._L.kslam_pseudo_owned(self._h, d_heads, n, C.byref(p)))
        return p.value or 0

    def pseudo_return(self, d_scores, n, score_fraction=0.95):
        """kslam_pseudo_return -> stats dict"""
        st = PairStats()
        self._chk(self._L.kslam_pseudo_return(self._h, d_scores, n, score_fraction, C.byref(st)))
        return st.as_dict()

    def pair_screen_overlaps(self, overlaps, read_lens, paired=True, score_threshold=0, score_fraction=0.95, stages=3):
        """kslam_pair_screen_overlaps on host arrays -> stats dict"""
        ov = np.ascontiguousarray(overlaps, dtype=OVERLAP_DT)
        rl = np.ascontiguousarray(read_lens, dtype=np.uint32)
        st = PairStats()
        self._chk(self._L.kslam_pair_screen_overlaps(self._h, ov.ctypes.data if len(ov) else None, len(ov),
                                                     rl.ctypes.data if len(rl) else None, len(rl), int(paired),
                                                     score_threshold, score_fraction, stages, C.byref(st)))
        return st.as_dict()

    def take_pairs(self, copy=True):
        """kslam_take_pairs -> (read_pairs[READ_PAIR_DT], pairs[PAIRED_OVERLAP_DT][, release])"""
        pg, pp, ng, npr = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        self._chk(self._L.kslam_take_pairs(self._h, C.byref(pg), C.byref(ng), C.byref(pp), C.byref(npr)))
        rp = np.frombuffer((C.c_char * (int(ng.value) * 24)).from_address(pg.value), dtype=READ_PAIR_DT) \
            if ng.value else np.zeros(0, dtype=READ_PAIR_DT)
        pr = np.frombuffer((C.c_char * (int(npr.value) * 32)).from_address(pp.value), dtype=PAIRED_OVERLAP_DT) \
            if npr.value else np.zeros(0, dtype=PAIRED_OVERLAP_DT)

        def release():
            self._L.kslam_free_pinned(self._h, pg)
            self._L.kslam_free_pinned(self._h, pp)
        if copy:
            rp, pr = rp.copy(), pr.copy()
            release()
            return rp, pr
        return rp, pr, release

    def debug_wave_sort(self, keys, seg_off):
        """kslam_debug_wave_sort -> the permutation (per segment) the device's std::sort restatement produces"""
        keys = np.ascontiguousarray(keys, dtype=np.int32)
        seg_off = np.ascontiguousarray(seg_off, dtype=np.uint64)
        perm = np.zeros(len(keys), dtype=np.uint32)
        self._chk(self._L.kslam_debug_wave_sort(self._h, keys.ctypes.data, seg_off.ctypes.data, len(seg_off) - 1,
                                                perm.ctypes.data))
        return perm

    def set_pairing(self, paired=True, score_threshold=0, score_fraction=0.95, stages=3):
        """kslam_set_pairing: what the pipelined lanes run after the alignment (stages=0: off)"""
        self._chk(self._L.kslam_set_pairing(self._h, int(paired), score_threshold, score_fraction, stages))

    def submit_batch_full(self, n_reads, bases_pp, quals_pp, lens_p):
        """kslam_submit_batch on ready pointer arrays (quals_pp may be None)"""
        t = C.c_uint64()
        self._chk(self._L.kslam_submit_batch(self._h, n_reads, bases_pp, quals_pp, lens_p, C.byref(t)))
        return int(t.value)

    def submit_batch_columns(self, n_reads, bases_p, quality_p, offsets_p):
        """kslam_submit_batch_columns: addresses of the concatenated bases / qualities and the uint64 offsets;
        they must stay valid until collect_batch"""
        t = C.c_uint64()
        self._chk(self._L.kslam_submit_batch_columns(self._h, n_reads, bases_p, quality_p, offsets_p, C.byref(t)))
        return int(t.value)

    def submit_batch_fastq(self, r1_p, len1, r2_p, len2, n_reads, offsets_p, bases_at_p, quality_at_p):
        """kslam_submit_batch_fastq: addresses of the two texts and of the index arrays of
        kslam_amd.fastq.index_pair; all must stay valid until collect_batch"""
        t = C.c_uint64()
        self._chk(self._L.kslam_submit_batch_fastq(self._h, r1_p, len1, r2_p, len2, n_reads, offsets_p, bases_at_p,
                                                   quality_at_p, C.byref(t)))
        return int(t.value)

    def submit_batch_fastq_text(self, r1_p, len1, r2_p, len2, max_pairs=0, at_eof=True):
        """kslam_submit_batch_fastq_text: the two texts (addresses; valid until collect_batch), nothing else"""
        t = C.c_uint64()
        self._chk(self._L.kslam_submit_batch_fastq_text(self._h, r1_p, len1, r2_p, len2, max_pairs, int(at_eof), C.byref(t)))
        return int(t.value)

    def collect_batch(self, ticket):
        """kslam_collect_batch -> (overlaps, cigar_pool, details, md_pool, release): views of the library's
        page-locked buffers (details / md_pool empty when no qualities were submitted)"""
        r = BatchResult()
        self._chk(self._L.kslam_collect_batch(self._h, ticket, C.byref(r)))

        def view(ptr, n, dt):
            return np.frombuffer((C.c_char * (int(n) * dt.itemsize)).from_address(ptr), dtype=dt) \
                if n and ptr else np.zeros(0, dtype=dt)
        ov = view(r.overlaps, r.n_overlaps, OVERLAP_DT)
        cg = view(r.cigar_pool, r.n_cigar, np.dtype(np.uint32))
        det = view(r.details, r.n_overlaps if r.details else 0, ROW_DETAIL_DT)
        md = view(r.md_pool, r.n_md, np.dtype(np.uint8))
        self.last_pairs = (view(r.read_pairs, r.n_read_pairs, READ_PAIR_DT), view(r.pairs, r.n_pairs, PAIRED_OVERLAP_DT),
                           r.pair_stats.as_dict()) if r.read_pairs or r.pairs else None
        # kslam_submit_batch_fastq_text: the batch's host columns as a reads view for kslam_amd.tail
        self.last_reads = None
        if r.reads_bases_off:
            from . import tail as _T
            n = int(r.n_reads)
            rv = _T.ReadsView(n, None, r.reads_bases_off, None, r.reads_bases_off, r.reads_ids, r.reads_ids_off)
            ids_off = view(r.reads_ids_off, n + 1, np.dtype(np.uint64))
            self.last_reads = type("ReadsFromDevice", (), {
                "view": rv, "n_reads": n, "consumed": (int(r.consumed1), int(r.consumed2)),
                "bases_off": view(r.reads_bases_off, n + 1, np.dtype(np.uint64)), "ids_off": ids_off,
                "ids_bytes": view(r.reads_ids, int(ids_off[n]) if n else 0, np.dtype(np.uint8))})()

        def release():
            self._L.kslam_release_batch(self._h, C.byref(r))
        return ov, cg, det, md, release

    def load_reads(self, reads):
        cat, off = _concat(reads)
        self._chk(self._L.kslam_load_reads(self._h, len(reads), cat.ctypes.data, off.ctypes.data))

    def load_reads_arrays(self, cat_u8, offsets_u64):
        """kslam_load_reads on columns that already exist (e.g. kslam_amd.fastq.Batch.bases_array())."""
        self._chk(self._L.kslam_load_reads(self._h, len(offsets_u64) - 1, cat_u8.ctypes.data, offsets_u64.ctypes.data))

    def load_reads_device(self, n_reads, dev_ptr, host_offsets):
        off = np.ascontiguousarray(host_offsets, dtype=np.uint64)
        self._chk(self._L.kslam_load_reads_device(self._h, n_reads, dev_ptr, off.ctypes.data))

    def align_resident(self):
        n_out, n_cig = C.c_uint64(), C.c_uint64()
        self._chk(self._L.kslam_align_resident(self._h, C.byref(n_out), C.byref(n_cig)))
        return int(n_out.value), int(n_cig.value)

    def fetch_results(self, n_out, n_cig):
        ov = np.zeros(n_out, dtype=OVERLAP_DT)
        cg = np.zeros(n_cig, dtype=np.uint32)
        self._chk(self._L.kslam_fetch_results(self._h, ov.ctypes.data, cg.ctypes.data))
        return ov, cg

    def take_results(self):
        """Last results as numpy views of the library's page-locked buffers (no copy).
        Returns (overlaps, cigar_pool, release); call release() when done with both arrays."""
        po, pc, no, nc = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        self._chk(self._L.kslam_take_results(self._h, C.byref(po), C.byref(no), C.byref(pc), C.byref(nc)))
        ov = np.frombuffer((C.c_char * (no.value * OVERLAP_DT.itemsize)).from_address(po.value),
                           dtype=OVERLAP_DT) if no.value else np.zeros(0, dtype=OVERLAP_DT)
        cg = np.frombuffer((C.c_char * (nc.value * 4)).from_address(pc.value),
                           dtype=np.uint32) if nc.value else np.zeros(0, dtype=np.uint32)

        def release():
            self._L.kslam_free_batch(self._h, po, pc)
        return ov, cg, release

    def copy_results_device(self, d_overlaps, d_cigars):
        self._chk(self._L.kslam_copy_results_device(self._h, d_overlaps, d_cigars))

    def merge_shards_device(self, shards, n_pairs, d_overlaps, d_cigars, d_out_overlaps, d_out_cigars):
        """kslam_merge_shards_device: shards = [(pair_lo, pair_hi, n_rows, n_cigar), ...] in batch order;
        the four pointers are device addresses on this context's GPU."""
        sh = np.array([tuple(int(v) for v in x) for x in shards], dtype=SHARD_DT)
        self._chk(self._L.kslam_merge_shards_device(self._h, len(sh), sh.ctypes.data, n_pairs, d_overlaps, d_cigars,
                                                    d_out_overlaps, d_out_cigars))

    def shard_counts_device(self, n_local_pairs):
        """kslam_shard_counts_device -> (n_rows, n_rows_r1, n_cigar, n_cigar_r1) of the last results"""
        a = np.zeros(4, dtype=np.uint64)
        self._chk(self._L.kslam_shard_counts_device(self._h, n_local_pairs, a.ctypes.data))
        return tuple(int(v) for v in a)

    def export_shard_device(self, n_local_pairs, pair_lo, n_pairs_total, pool_base_r1, pool_base_r2,
                            d_rows_r1, d_rows_r2, d_pool_r1, d_pool_r2):
        self._chk(self._L.kslam_export_shard_device(self._h, n_local_pairs, pair_lo, n_pairs_total, pool_base_r1,
                                                    pool_base_r2, d_rows_r1, d_rows_r2, d_pool_r1, d_pool_r2))

    def timings(self):
        t = Timings()
        self._chk(self._L.kslam_get_timings(self._h, C.byref(t)))
        return t.as_dict()

    # ---- stage-level entry points ----
    def extract_kmers(self, seqs, is_gb, gap):
        ptrs, keep = _seq_arrays(seqs)
        lens = np.array([len(s) for s in seqs], dtype=np.uint64)
        cap = int(sum((int(x) - 32) // gap + 1 for x in lens if x >= 32))
        out = np.zeros(max(cap, 1), dtype=KMER_DT)
        n = C.c_uint64()
        self._chk(self._L.kslam_extract_kmers(self._h, len(seqs), C.cast(ptrs, C.c_void_p),
                                              lens.ctypes.data, int(is_gb), gap, out.ctypes.data,
                                              cap, C.byref(n)))
        assert int(n.value) == cap
        return out[:cap]

    def sort_kmers(self, recs):
        out = np.ascontiguousarray(recs.copy())
        self._chk(self._L.kslam_sort_kmers(self._h, out.ctypes.data, len(out)))
        return out

    def selftest_sort(self, n, iters=3):
        """(ms per 8-pass sort, ms per scatter launch, inversions) on n random records."""
        a, b, inv = C.c_float(), C.c_float(), C.c_uint64()
        self._chk(self._L.kslam_selftest_sort(self._h, n, iters, C.byref(a), C.byref(b), C.byref(inv)))
        return float(a.value), float(b.value), int(inv.value)

    def find_overlaps(self):
        out = C.c_void_p()
        n, raw = C.c_uint64(), C.c_uint64()
        self._chk(self._L.kslam_find_overlaps(self._h, C.byref(out), C.byref(n), C.byref(raw)))
        m = int(n.value)
        ov = np.frombuffer((C.c_char * (m * 16)).from_address(out.value),
                           dtype=OVERLAP_TEMP_DT).copy() if m else np.zeros(0, dtype=OVERLAP_TEMP_DT)
        self._L.kslam_free(out)
        return ov, int(raw.value)


SHARD_DT = np.dtype([("pair_lo", "<u8"), ("pair_hi", "<u8"), ("n_rows", "<u8"), ("n_cigar", "<u8")])


class HostBuffer:
    """kslam_host_alloc: page-locked host memory as a numpy uint8 array (`.a`); free with close()."""

    def __init__(self, nbytes):
        self._L = lib()
        self.nbytes = int(nbytes)
        self.ptr = self._L.kslam_host_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError("kslam_host_alloc(%d) failed" % nbytes)
        self.a = np.frombuffer((C.c_char * self.nbytes).from_address(self.ptr), dtype=np.uint8)

    def close(self):
        if self.ptr:
            self.a = None
            self._L.kslam_host_free(self.ptr, self.nbytes)
            self.ptr = None


class MultiContext:
    """kslam_multi: one process, one context per entry of `devices` (an ordinal may repeat), read
    pairs sharded, results gathered to devices[0] and merged there (include/kslam.h)."""

    def __init__(self, devices, match=2, mismatch=3, gap_open=5, gap_extend=2, score_threshold=0,
                 report_cigar=True, max_kmers_per_chunk=0):
        self._L = lib()
        self._h = C.c_void_p()
        p = Params(match, mismatch, gap_open, gap_extend, score_threshold, 1 if report_cigar else 0, 0,
                   max_kmers_per_chunk)
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        st = self._L.kslam_multi_create(C.byref(p), dv.ctypes.data, len(dv), C.byref(self._h))
        if st != 0:
            msg = self._L.kslam_multi_last_error(self._h).decode() if self._h else "create failed"
            if self._h:
                self._L.kslam_multi_destroy(self._h)
                self._h = C.c_void_p()
            raise KslamError(st, msg)

    def close(self):
        if self._h:
            self._L.kslam_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st):
        if st != 0:
            raise KslamError(st, self._L.kslam_multi_last_error(self._h).decode())

    def set_index(self, entries):
        ptrs, keep = _seq_arrays(entries)
        lens = np.array([len(s) for s in entries], dtype=np.uint64)
        self._chk(self._L.kslam_multi_set_index(self._h, len(entries), C.cast(ptrs, C.c_void_p), lens.ctypes.data))

    def align_batch(self, reads, paired=True):
        ptrs, keep = _seq_arrays(reads)
        lens = np.array([len(s) for s in reads], dtype=np.uint32)
        out, cig = C.c_void_p(), C.c_void_p()
        n_out, n_cig = C.c_uint64(), C.c_uint64()
        self._chk(self._L.kslam_multi_align_batch(self._h, len(reads), C.cast(ptrs, C.c_void_p), lens.ctypes.data,
                                                  1 if paired else 0, C.byref(out), C.byref(n_out),
                                                  C.byref(cig), C.byref(n_cig)))
        n, nc = int(n_out.value), int(n_cig.value)
        ov = np.frombuffer((C.c_char * (n * 48)).from_address(out.value), dtype=OVERLAP_DT).copy() \
            if n else np.zeros(0, dtype=OVERLAP_DT)
        cg = np.frombuffer((C.c_char * (nc * 4)).from_address(cig.value), dtype=np.uint32).copy() \
            if nc else np.zeros(0, dtype=np.uint32)
        self._L.kslam_multi_free_batch(self._h, out, cig)
        return ov, cg


def align_to_database(reads, entries, **params):
    """One-shot mirror of alignToDatabase(reads, genbankIndex), reference src/SLAM.h:59-79."""
    ctx = Context(**params)
    try:
        ctx.set_index(entries)
        return ctx.align_batch(reads)
    finally:
        ctx.close()
