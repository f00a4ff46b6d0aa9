"""ctypes binding of oracle/liboracle.so (+ the optional oracle/_ref libraries).

TEST INFRASTRUCTURE ONLY: the checker, never the thing measured or shipped.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

KMER_DT = np.dtype([("kmer", "<u8"), ("meta", "<u4"), ("offset", "<u4")])
OVERLAP_DT = np.dtype([("read", "<u4"), ("entry", "<u4"), ("rel", "<i4"),
                       ("revcomp", "u1"), ("pad", "u1", (3,))])
ALIGN_DT = np.dtype([("read", "<u4"), ("entry", "<u4"), ("rel", "<i4"),
                     ("revcomp", "u1"), ("pad", "u1"), ("score", "<u2"),
                     ("ref_begin", "<i4"), ("ref_end", "<i4"),
                     ("query_begin", "<i4"), ("query_end", "<i4"),
                     ("cigar_len", "<u4"), ("pad2", "<u4"), ("cigar_off", "<u8")])
assert KMER_DT.itemsize == 16 and OVERLAP_DT.itemsize == 16
assert ALIGN_DT.itemsize == 48


class Params(C.Structure):
    """orc_params: the scoring globals of reference src/Globals.h:27-36."""
    _fields_ = [("match", C.c_uint32), ("mismatch", C.c_uint32),
                ("gap_open", C.c_uint32), ("gap_extend", C.c_uint32),
                ("score_threshold", C.c_uint32), ("report_cigar", C.c_int32)]

    @classmethod
    def default(cls, report_cigar=True, score_threshold=0, match=2, mismatch=3,
                gap_open=5, gap_extend=2):
        # defaults of reference src/main.cpp:44-55
        return cls(match, mismatch, gap_open, gap_extend, score_threshold,
                   1 if report_cigar else 0)


class SswResult(C.Structure):
    _fields_ = [("score1", C.c_uint16), ("ref_begin1", C.c_int32),
                ("ref_end1", C.c_int32), ("read_begin1", C.c_int32),
                ("read_end1", C.c_int32), ("cigar_len", C.c_int32),
                ("status", C.c_int32)]


def build(force=False):
    """Compile the checker libraries (and oracle/_ref when /root/reference exists).  make decides
    what is out of date; on a box without the sources the prebuilt files are used as they are."""
    so = os.path.join(_HERE, "liboracle.so")
    have_all = all(os.path.exists(os.path.join(_HERE, f))
                   for f in ("liboracle.so", "libtail_oracle.so", "libfastq_oracle.so",
                             "libtaxonomy_oracle.so"))
    if force:
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean"])
    if force or not have_all or os.path.isdir("/root/reference/src") or _stale():
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def _stale():
    pairs = [("liboracle.so", "kslam_oracle.c"), ("libtail_oracle.so", "tail_oracle.cpp"),
             ("libfastq_oracle.so", "fastq_oracle.cpp"), ("libtaxonomy_oracle.so", "taxonomy_oracle.cpp")]
    return any(os.path.getmtime(os.path.join(_HERE, a)) < os.path.getmtime(os.path.join(_HERE, b))
               for a, b in pairs)


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = build()
        L = C.CDLL(so)
        u64, u32, i32, vp = C.c_uint64, C.c_uint32, C.c_int32, C.c_void_p
        L.orc_count_kmers.restype = u64
        L.orc_count_kmers.argtypes = [u64, C.c_uint]
        L.orc_extract_all.restype = u64
        L.orc_extract_all.argtypes = [u64, vp, vp, C.c_int, C.c_uint, vp]
        L.orc_sort_kmers.argtypes = [vp, u64]
        L.orc_count_overlaps.restype = u64
        L.orc_count_overlaps.argtypes = [vp, u64]
        L.orc_scan_overlaps.restype = u64
        L.orc_scan_overlaps.argtypes = [vp, u64, vp, vp]
        L.orc_find_overlaps.restype = u64
        L.orc_find_overlaps.argtypes = [vp, u64, vp, vp, vp]
        L.orc_build_matrix.argtypes = [u32, u32, vp]
        L.orc_translate.argtypes = [C.c_char_p, i32, vp]
        for f in (L.orc_ssw_align, L.orc_ssw_align_plain):
            f.argtypes = [vp, i32, vp, i32, vp, C.c_uint8, C.c_uint8, C.c_uint8,
                          C.c_uint16, i32, vp, i32, C.POINTER(SswResult)]
        L.orc_ssw_align_mode.argtypes = [vp, i32, vp, i32, vp, C.c_uint8, C.c_uint8, C.c_uint8,
                                         C.c_uint16, i32, vp, i32, C.POINTER(SswResult), C.c_int]
        L.orc_banded_sw.restype = i32
        L.orc_banded_sw.argtypes = [vp, vp, i32, i32, i32, u32, u32, i32, vp, i32, vp, i32, vp]
        L.orc_align.argtypes = [C.c_char_p, i32, C.c_char_p, i32, C.POINTER(Params),
                                vp, i32, C.POINTER(SswResult), C.c_int]
        L.orc_sw_on_overlap.argtypes = [vp, C.c_char_p, u64, C.c_char_p, u64,
                                        C.POINTER(Params), vp, vp, i32, C.c_int]
        L.orc_align_to_database.restype = C.c_int
        L.orc_align_to_database.argtypes = [u64, vp, vp, u64, vp, vp, C.POINTER(Params),
                                            C.c_int, C.POINTER(vp), C.POINTER(u64),
                                            C.POINTER(vp), C.POINTER(u64), vp]
        L.orc_free.argtypes = [vp]
        L.orc_use_reference_ssw.restype = C.c_int
        L.orc_use_reference_ssw.argtypes = [C.c_char_p]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _seq_arrays(seqs):
    """list[bytes] -> (char** array, uint64 lens, keepalive)."""
    n = len(seqs)
    bufs = [C.create_string_buffer(s, len(s) + 1) for s in seqs]
    ptrs = (C.c_char_p * max(n, 1))(*[C.cast(b, C.c_char_p) for b in bufs])
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    return ptrs, lens, bufs


def extract_kmers(seqs, is_gb, gap):
    """getKMers_parallel (reference src/KMer.h:190-241) over a list of bytes."""
    L = lib()
    ptrs, lens, keep = _seq_arrays(seqs)
    tot = sum(int(L.orc_count_kmers(int(x), gap)) for x in lens)
    out = np.zeros(tot, dtype=KMER_DT)
    got = L.orc_extract_all(len(seqs), C.cast(ptrs, C.c_void_p), lens.ctypes.data,
                            int(is_gb), gap, out.ctypes.data)
    assert got == tot
    return out


def sort_kmers(recs):
    """sortKMers (reference src/KMer.h:388-398); returns a sorted copy."""
    out = np.ascontiguousarray(recs.copy())
    lib().orc_sort_kmers(out.ctypes.data, len(out))
    return out


def find_overlaps(sorted_recs, read_lens):
    """findOverlaps_parallel (reference src/Overlap.h:277-295)."""
    L = lib()
    s = np.ascontiguousarray(sorted_recs)
    rl = np.ascontiguousarray(np.asarray(read_lens, dtype=np.uint64))
    raw = int(L.orc_count_overlaps(s.ctypes.data, len(s)))
    out = np.zeros(raw + 1, dtype=OVERLAP_DT)
    nraw = C.c_uint64(0)
    m = L.orc_find_overlaps(s.ctypes.data, len(s), rl.ctypes.data, out.ctypes.data,
                            C.byref(nraw))
    return out[:m].copy(), int(nraw.value)


def scan_overlaps(sorted_recs, read_lens):
    """findOverlaps (reference src/Overlap.h:230-246) over the whole range: the pre-dedupe list, emission order."""
    L = lib()
    s = np.ascontiguousarray(sorted_recs)
    rl = np.ascontiguousarray(np.asarray(read_lens, dtype=np.uint64))
    raw = int(L.orc_count_overlaps(s.ctypes.data, len(s)))
    out = np.zeros(raw + 1, dtype=OVERLAP_DT)
    m = L.orc_scan_overlaps(s.ctypes.data, len(s), rl.ctypes.data, out.ctypes.data)
    return out[:m].copy()


def build_matrix(match, mismatch):
    m = np.zeros(25, dtype=np.int8)
    lib().orc_build_matrix(match, mismatch, m.ctypes.data)
    return m


def translate(seq):
    out = np.zeros(max(len(seq), 1), dtype=np.int8)
    lib().orc_translate(seq, len(seq), out.ctypes.data)
    return out[:len(seq)]


def ssw_align(read_codes, ref_codes, mat, gap_open, gap_extend, flag=0x0f,
              filters=0, filterd=32767, plain=False):
    """ssw_align (reference src/ssw.c:841-951) on translated sequences."""
    L = lib()
    rd = np.ascontiguousarray(read_codes, dtype=np.int8)
    rf = np.ascontiguousarray(ref_codes, dtype=np.int8)
    cap = 2 * (len(rd) + len(rf)) + 8
    cig = np.zeros(cap, dtype=np.uint32)
    res = SswResult()
    L.orc_ssw_align_mode(rd.ctypes.data, len(rd), rf.ctypes.data, len(rf), mat.ctypes.data, gap_open,
                         gap_extend, flag, filters, filterd, cig.ctypes.data, cap, C.byref(res),
                         int(plain))
    return res, cig[:max(res.cigar_len, 0)].copy()


def align(query, ref, params=None, plain=False):
    """Aligner::Align (reference src/ssw_cpp.cpp:234-283) on ASCII bytes."""
    p = params or Params.default()
    cap = 2 * (len(query) + len(ref)) + 8
    cig = np.zeros(cap, dtype=np.uint32)
    res = SswResult()
    lib().orc_align(query, len(query), ref, len(ref), C.byref(p), cig.ctypes.data, cap,
                    C.byref(res), int(plain))
    return res, cig[:max(res.cigar_len, 0)].copy()


def align_to_database(reads, entries, params=None, plain=False):
    """alignToDatabase (reference src/SLAM.h:59-79).

    Returns (alignments[ALIGN_DT], cigar_pool[uint32], phase_seconds[6])."""
    L = lib()
    p = params or Params.default()
    rp, rl, k1 = _seq_arrays(reads)
    ep, el, k2 = _seq_arrays(entries)
    out, cig = C.c_void_p(), C.c_void_p()
    n_out, n_cig = C.c_uint64(), C.c_uint64()
    ph = np.zeros(6, dtype=np.float64)
    rc = L.orc_align_to_database(len(reads), C.cast(rp, C.c_void_p), rl.ctypes.data,
                                 len(entries), C.cast(ep, C.c_void_p), el.ctypes.data,
                                 C.byref(p), int(plain), C.byref(out), C.byref(n_out),
                                 C.byref(cig), C.byref(n_cig), ph.ctypes.data)
    if rc != 0:
        raise MemoryError("orc_align_to_database failed")
    n, nc = int(n_out.value), int(n_cig.value)
    al = np.frombuffer((C.c_char * (n * ALIGN_DT.itemsize)).from_address(out.value),
                       dtype=ALIGN_DT).copy() if n else np.zeros(0, dtype=ALIGN_DT)
    cg = np.frombuffer((C.c_char * (nc * 4)).from_address(cig.value),
                       dtype=np.uint32).copy() if nc else np.zeros(0, dtype=np.uint32)
    L.orc_free(out)
    L.orc_free(cig)
    return al, cg, ph


def num_threads():
    return int(lib().orc_num_threads())


def usable_cpus():
    """CPUs this process can really use: os.cpu_count() capped by a cgroup-v2 CPU quota
    (the GPU boxes run jobs under cpu.max = 16 CPUs of a 256-thread host)."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


# ---------------------------------------------------------------------------
# the REAL reference, compiled in place into oracle/_ref (may be absent)
# ---------------------------------------------------------------------------
REF_SSW = os.path.join(_HERE, "_ref", "libssw_ref.so")
REF_KMER = os.path.join(_HERE, "_ref", "libkmer_ref.so")


def have_ref_ssw():
    build()
    return os.path.exists(REF_SSW)


def have_ref_kmer():
    build()
    return os.path.exists(REF_KMER)


def use_reference_ssw(enable=True):
    """Route orc_align's SSW core through the real ssw.c (cpu_baseline leg)."""
    if enable and not have_ref_ssw():
        return False
    rc = lib().orc_use_reference_ssw(REF_SSW.encode() if enable else None)
    return rc == 0


class _RefSAlign(C.Structure):  # s_align, reference src/ssw.h:47-57
    _fields_ = [("score1", C.c_uint16), ("score2", C.c_uint16),
                ("ref_begin1", C.c_int32), ("ref_end1", C.c_int32),
                ("read_begin1", C.c_int32), ("read_end1", C.c_int32),
                ("ref_end2", C.c_int32), ("cigar", C.POINTER(C.c_uint32)),
                ("cigarLen", C.c_int32)]


_ref_ssw = None


def ref_ssw_align(read_codes, ref_codes, mat, gap_open, gap_extend, flag=0x0f,
                  filters=0, filterd=32767):
    """Call the reference's own ssw_init + ssw_align (src/ssw.c:808,841)."""
    global _ref_ssw
    if _ref_ssw is None:
        R = C.CDLL(REF_SSW)
        R.ssw_init.restype = C.c_void_p
        R.ssw_init.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int8]
        R.ssw_align.restype = C.POINTER(_RefSAlign)
        R.ssw_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint8, C.c_uint8,
                                C.c_uint8, C.c_uint16, C.c_int32, C.c_int32]
        R.init_destroy.argtypes = [C.c_void_p]
        R.align_destroy.argtypes = [C.POINTER(_RefSAlign)]
        _ref_ssw = R
    R = _ref_ssw
    rd = np.ascontiguousarray(read_codes, dtype=np.int8)
    rf = np.ascontiguousarray(ref_codes, dtype=np.int8)
    prof = R.ssw_init(rd.ctypes.data, len(rd), mat.ctypes.data, 5, 2)
    a = R.ssw_align(prof, rf.ctypes.data, len(rf), gap_open, gap_extend, flag, filters,
                    filterd, len(rd))
    s = a.contents
    cig = np.array([s.cigar[i] for i in range(s.cigarLen)], dtype=np.uint32) \
        if s.cigar else np.zeros(0, dtype=np.uint32)
    res = (s.score1, s.ref_begin1, s.ref_end1, s.read_begin1, s.read_end1)
    R.align_destroy(a)
    R.init_destroy(prof)
    return res, cig


_ref_kmer = None


def _refk():
    global _ref_kmer
    if _ref_kmer is None:
        R = C.CDLL(REF_KMER)
        R.ref_extract_kmers.restype = C.c_uint64
        R.ref_extract_kmers.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_uint, C.c_void_p, C.c_uint64]
        R.ref_sort_kmers.restype = C.c_int
        R.ref_sort_kmers.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p]
        R.ref_kmer3.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _ref_kmer = R
    return _ref_kmer


def ref_extract_kmers(seqs, is_gb, gap):
    """The reference's getKMers_parallel (src/KMer.h:190-241) itself."""
    R = _refk()
    ptrs, lens, keep = _seq_arrays(seqs)
    cap = sum(max(0, (len(s) - 32) // gap + 1) if len(s) >= 32 else 0 for s in seqs) + 16
    out = np.zeros(cap, dtype=KMER_DT)
    n = R.ref_extract_kmers(len(seqs), C.cast(ptrs, C.c_void_p), lens.ctypes.data,
                            int(is_gb), gap, out.ctypes.data, cap)
    assert n <= cap
    return out[:n].copy()


def ref_sort_kmers(recs):
    """The reference's sortKMers (src/KMer.h:388-398) itself."""
    out = np.ascontiguousarray(recs.copy())
    with tempfile.TemporaryDirectory() as d:
        rc = _refk().ref_sort_kmers(out.ctypes.data, len(out), d.encode())
    assert rc == 0
    return out


def ref_kmer3(s):
    f, r = C.c_uint32(), C.c_uint32()
    _refk().ref_kmer3(s, C.byref(f), C.byref(r))
    return int(f.value), int(r.value)


def cigar_string(cig):
    return "".join("%d%s" % (int(c) >> 4, "MID"[int(c) & 15]) for c in cig)


# ---- host tail (oracle/tail_oracle.cpp): pairing .. SAM, serial restatement ----
PAIRED_OVERLAP_DT = np.dtype([("combined_score", "<u4"), ("entry", "<u4"), ("ref_start", "<i4"),
                              ("ref_end", "<i4"), ("insert_size", "<u4"), ("r1", "<u4"),
                              ("r2", "<u4"), ("pad", "<u4")])
READ_PAIR_DT = np.dtype([("r1_read", "<u4"), ("r2_read", "<u4"), ("first", "<u8"), ("count", "<u8")])
_tail = None


def tail_lib():
    global _tail
    if _tail is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "libtail_oracle.so"))
        vp, u64 = C.c_void_p, C.c_uint64
        L.orc_tail_pairs.argtypes = [vp, vp, vp, u64, C.POINTER(vp), C.POINTER(u64), C.POINTER(vp),
                                     C.POINTER(u64), vp]
        L.orc_tail_sam.argtypes = [vp, vp, vp, vp, u64, vp, u64, C.POINTER(vp), C.POINTER(u64), vp]
        L.orc_sam_header.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(u64)]
        L.orc_tail_free.argtypes = [vp]
        L.orc_tail_force_insert_limit.argtypes = [C.c_int64]
        L.orc_tail_force_insert_limit.restype = None
        _tail = L
    return _tail


def _tail_take(ptr, n, dtype):
    if n:
        buf = (C.c_char * (int(n) * dtype.itemsize)).from_address(ptr.value)
        out = np.frombuffer(buf, dtype=dtype).copy()
    else:
        out = np.zeros(0, dtype=dtype)
    tail_lib().orc_tail_free(ptr)
    return out


def tail_pairs(params, reads_view, overlaps, stats=None):
    """params / reads_view / stats: the ctypes structures of include/kslam_tail.h
    (the tests build them with the product's plumbing module).  -> (read_pairs, pairs)"""
    L = tail_lib()
    ov = np.ascontiguousarray(overlaps, dtype=ALIGN_DT)
    rp, pr, nrp, npr = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
    rc = L.orc_tail_pairs(C.addressof(params), C.addressof(reads_view), ov.ctypes.data, len(ov),
                          C.byref(rp), C.byref(nrp), C.byref(pr), C.byref(npr),
                          C.addressof(stats) if stats is not None else None)
    assert rc == 0
    return _tail_take(rp, nrp.value, READ_PAIR_DT), _tail_take(pr, npr.value, PAIRED_OVERLAP_DT)


def tail_sam(params, reads_view, index_view, overlaps, cigar_pool, stats=None):
    L = tail_lib()
    ov = np.ascontiguousarray(overlaps, dtype=ALIGN_DT)
    pool = np.ascontiguousarray(cigar_pool, dtype=np.uint32)
    txt, n = C.c_void_p(), C.c_uint64()
    rc = L.orc_tail_sam(C.addressof(params), C.addressof(reads_view), C.addressof(index_view),
                        ov.ctypes.data, len(ov), pool.ctypes.data if len(pool) else None, len(pool),
                        C.byref(txt), C.byref(n), C.addressof(stats) if stats is not None else None)
    assert rc == 0
    out = C.string_at(txt.value, n.value)
    L.orc_tail_free(txt)
    return out


def tail_force_insert_limit(limit):
    """the next tail_pairs / tail_sam calls screen with this insert-size limit (the whole batch's) instead of the one
    they would compute from their own read pairs; None switches back"""
    tail_lib().orc_tail_force_insert_limit(-1 if limit is None else int(limit))


def sam_header(index_view, command_line=b""):
    L = tail_lib()
    txt, n = C.c_void_p(), C.c_uint64()
    assert L.orc_sam_header(C.addressof(index_view), command_line, C.byref(txt), C.byref(n)) == 0
    out = C.string_at(txt.value, n.value)
    L.orc_tail_free(txt)
    return out


# ---- FASTQ reader (oracle/fastq_oracle.cpp) and the real one (oracle/_ref/libfastq_ref.so) ----
REF_FASTQ = os.path.join(_HERE, "_ref", "libfastq_ref.so")
_fq = None
_fqref = None


def _unflatten(n, ptrs, free):
    out = []
    for text, off in ptrs:
        o = np.frombuffer((C.c_char * (8 * (n + 1))).from_address(off.value), dtype=np.uint64).copy()
        raw = C.string_at(text.value, int(o[n])) if int(o[n]) else b""
        out.append([raw[int(o[i]):int(o[i + 1])] for i in range(n)])
        free(text)
        free(off)
    return out


def fastq_read(text, pos=0, max_reads=0xFFFFFFFF):
    """One getSequencesFromFASTQFile call on the restated reader, starting at byte `pos`.
    -> (bases, quality, ids, position of the stream afterwards)"""
    global _fq
    if _fq is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "libfastq_oracle.so"))
        vp, u64 = C.c_void_p, C.c_uint64
        L.orc_fastq_read.argtypes = [C.c_char_p, u64, C.POINTER(u64), C.c_uint32, C.POINTER(u64)] + \
            [C.POINTER(vp)] * 6
        L.orc_fastq_free.argtypes = [vp]
        _fq = L
    p, n = C.c_uint64(pos), C.c_uint64()
    v = [C.c_void_p() for _ in range(6)]
    assert _fq.orc_fastq_read(text, len(text), C.byref(p), max_reads, C.byref(n),
                              *[C.byref(x) for x in v]) == 0
    b, q, i = _unflatten(int(n.value), [(v[0], v[1]), (v[2], v[3]), (v[4], v[5])], _fq.orc_fastq_free)
    return b, q, i, int(p.value)


def have_ref_fastq():
    build()
    return os.path.exists(REF_FASTQ)


def ref_fastq_read(path, per_call=0xFFFFFFFF):
    """The REAL reference reader on a file, called per_call reads at a time until exhausted.
    -> (bases, quality, ids, reads returned by each call)"""
    global _fqref
    if _fqref is None:
        L = C.CDLL(REF_FASTQ)
        vp, u64 = C.c_void_p, C.c_uint64
        L.ref_fastq_read.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(u64)] + [C.POINTER(vp)] * 6 + \
            [vp, u64, C.POINTER(u64)]
        L.ref_fastq_free.argtypes = [vp]
        _fqref = L
    n, nc = C.c_uint64(), C.c_uint64()
    calls = np.zeros(4096, dtype=np.uint64)
    v = [C.c_void_p() for _ in range(6)]
    rc = _fqref.ref_fastq_read(path.encode(), per_call, C.byref(n), *[C.byref(x) for x in v],
                               calls.ctypes.data, len(calls), C.byref(nc))
    assert rc == 0, "reference reader could not open " + path
    b, q, i = _unflatten(int(n.value), [(v[0], v[1]), (v[2], v[3]), (v[4], v[5])], _fqref.ref_fastq_free)
    return b, q, i, [int(x) for x in calls[:int(nc.value)]]


# ---- taxonomy tree (oracle/taxonomy_oracle.cpp) and the real one (oracle/_ref/libtaxonomy_ref.so) ----
REF_TAXONOMY = os.path.join(_HERE, "_ref", "libtaxonomy_ref.so")


class _TaxTree:
    """Common surface of the restated tree and the reference's TaxonomyDB."""

    def __init__(self, L, prefix, handle, closer):
        self._L, self._p, self._h, self._close = L, prefix, handle, closer

    def _f(self, name):
        return getattr(self._L, self._p + name)

    def lca(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.uint32)
        return int(self._f("taxdb_lca")(self._h, a.ctypes.data, len(a)))

    def parent(self, i):
        return int(self._f("taxdb_parent")(self._h, i))

    def at_rank(self, i, rank):
        return int(self._f("taxdb_at_rank")(self._h, i, rank))

    def is_below(self, upper, lower):
        return int(self._f("taxdb_is_below")(self._h, upper, lower))

    def is_subspecies(self, i):
        return int(self._f("taxdb_is_subspecies")(self._h, i))

    def text(self, i, which):
        p = self._f("taxdb_text")(self._h, i, which)
        s = C.string_at(p)
        self._f("tax_free")(p)
        return s

    def close(self):
        if self._h:
            self._close(self._h)
            self._h = None


def _tax_sigs(L, p):
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    getattr(L, p + "taxdb_lca").restype = u32
    getattr(L, p + "taxdb_lca").argtypes = [vp, vp, u64]
    getattr(L, p + "taxdb_parent").restype = u32
    getattr(L, p + "taxdb_parent").argtypes = [vp, u32]
    getattr(L, p + "taxdb_at_rank").restype = u32
    getattr(L, p + "taxdb_at_rank").argtypes = [vp, u32, C.c_char_p]
    getattr(L, p + "taxdb_is_below").restype = C.c_int32
    getattr(L, p + "taxdb_is_below").argtypes = [vp, u32, u32]
    getattr(L, p + "taxdb_is_subspecies").restype = C.c_int32
    getattr(L, p + "taxdb_is_subspecies").argtypes = [vp, u32]
    getattr(L, p + "taxdb_text").restype = vp
    getattr(L, p + "taxdb_text").argtypes = [vp, u32, C.c_int]
    getattr(L, p + "tax_free").argtypes = [vp]


_taxo = None
_taxref = None


def taxonomy_lib():
    global _taxo
    if _taxo is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "libtaxonomy_oracle.so"))
        _tax_sigs(L, "orc_")
        L.orc_taxdb_parse.restype = C.c_void_p
        L.orc_taxdb_parse.argtypes = [C.c_char_p, C.c_uint64]
        L.orc_taxdb_free.argtypes = [C.c_void_p]
        L.orc_taxonomy_summary.restype = C.c_void_p
        L.orc_taxonomy_summary.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
        _taxo = L
    return _taxo


def taxonomy_tree(text):
    L = taxonomy_lib()
    return _TaxTree(L, "orc_", L.orc_taxdb_parse(text, len(text)), L.orc_taxdb_free)


def taxonomy_summary(tree, ids, num_reads):
    L = taxonomy_lib()
    a = np.ascontiguousarray(ids, dtype=np.uint32)
    p = L.orc_taxonomy_summary(tree._h, a.ctypes.data, len(a), num_reads)
    s = C.string_at(p)
    L.orc_tax_free(p)
    return s


def have_ref_taxonomy():
    build()
    return os.path.exists(REF_TAXONOMY)


def ref_taxonomy_tree(path):
    """The REAL reference TaxonomyDB built from a taxDB file."""
    global _taxref
    if _taxref is None:
        L = C.CDLL(REF_TAXONOMY)
        _tax_sigs(L, "ref_")
        L.ref_taxdb_open.restype = C.c_void_p
        L.ref_taxdb_open.argtypes = [C.c_char_p]
        L.ref_taxdb_close.argtypes = [C.c_void_p]
        L.ref_taxdb_size.restype = C.c_uint64
        L.ref_taxdb_size.argtypes = [C.c_void_p]
        _taxref = L
    h = _taxref.ref_taxdb_open(path.encode())
    assert h, "reference TaxonomyDB could not open " + path
    return _TaxTree(_taxref, "ref_", h, _taxref.ref_taxdb_close)


# ---- the REAL join / dedupe / Aligner::Align / SW driver / alignToDatabase (oracle/_ref/libjoin_ref.so) ----
REF_JOIN = os.path.join(_HERE, "_ref", "libjoin_ref.so")
_refjoin = None


def have_ref_join():
    build()
    return os.path.exists(REF_JOIN)


def _refj():
    global _refjoin
    if _refjoin is None:
        R = C.CDLL(REF_JOIN)
        vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int32
        R.ref_find_overlaps.restype = u64
        R.ref_find_overlaps.argtypes = [vp, u64, u64, vp, vp, u64, C.POINTER(u64), vp, u64, C.c_char_p]
        R.ref_sort_unique_overlaps.restype = u64
        R.ref_sort_unique_overlaps.argtypes = [vp, u64]
        R.ref_aligner_align.restype = i32
        R.ref_aligner_align.argtypes = [C.c_char_p, C.c_char_p, i32, C.POINTER(Params), C.POINTER(C.c_uint16)] + \
            [C.POINTER(i32)] * 4 + [vp, i32]
        R.ref_align_to_database.restype = C.c_int
        R.ref_align_to_database.argtypes = [u64, vp, vp, u64, vp, vp, C.POINTER(Params), C.POINTER(vp),
                                            C.POINTER(u64), C.POINTER(vp), C.POINTER(u64), C.c_char_p]
        R.ref_sw_on_overlaps.restype = C.c_int
        R.ref_sw_on_overlaps.argtypes = [vp, u64, u64, vp, vp, u64, vp, vp, C.POINTER(Params), vp, vp, u64,
                                         C.POINTER(u64)]
        R.ref_free.argtypes = [vp]
        _refjoin = R
    return _refjoin


def _one_thread(fn):
    """overlapSort has no revComp in its key and __gnu_parallel::sort is unstable (src/Overlap.h:87-98, 289): run the
    reference with ONE OpenMP thread so that ties come out the same way every time (libstdc++'s parallel sort falls back
    to the sequential std::sort when the team has one thread)."""
    old = os.environ.get("OMP_NUM_THREADS")
    import ctypes.util
    gomp = C.CDLL(ctypes.util.find_library("gomp") or "libgomp.so.1")
    gomp.omp_get_max_threads.restype = C.c_int
    before = gomp.omp_get_max_threads()
    gomp.omp_set_num_threads(1)
    try:
        return fn()
    finally:
        gomp.omp_set_num_threads(before)
        if old is not None:
            os.environ["OMP_NUM_THREADS"] = old


def ref_find_overlaps(sorted_recs, read_lens, one_thread=True, want_raw=False):
    """The reference's own findOverlaps_parallel (src/Overlap.h:277-295) on a sorted record list.
    -> (deduped overlaps[OVERLAP_DT], raw count), or with want_raw (deduped, raw list[OVERLAP_DT] in emission order)"""
    R = _refj()
    s = np.ascontiguousarray(sorted_recs)
    rl = np.ascontiguousarray(np.asarray(read_lens, dtype=np.uint64))
    cap = int(lib().orc_count_overlaps(s.ctypes.data, len(s))) + 16
    out = np.zeros(cap, dtype=OVERLAP_DT)
    raw = np.zeros(cap, dtype=OVERLAP_DT)
    nraw = C.c_uint64(0)

    def run():
        with tempfile.TemporaryDirectory() as d:
            return R.ref_find_overlaps(s.ctypes.data, len(s), len(rl), rl.ctypes.data, out.ctypes.data, cap,
                                       C.byref(nraw), raw.ctypes.data, cap, d.encode())
    m = _one_thread(run) if one_thread else run()
    assert m <= cap and nraw.value <= cap
    if want_raw:
        return out[:m].copy(), raw[:int(nraw.value)].copy()
    return out[:m].copy(), int(nraw.value)


def ref_sort_unique_overlaps(overlaps):
    """__gnu_parallel::sort(overlapSort) + std::unique(overlapEqual), src/Overlap.h:289-291, one thread."""
    o = np.ascontiguousarray(overlaps.copy())
    m = _one_thread(lambda: _refj().ref_sort_unique_overlaps(o.ctypes.data, len(o)))
    return o[:m].copy()


def ref_aligner_align(query, ref, params=None, ref_len=None):
    """The reference's own Aligner::Align (src/ssw_cpp.cpp:234-283) on ASCII bytes.
    -> ((score, ref_begin, ref_end, query_begin, query_end), cigar[u32])"""
    p = params or Params.default()
    n = len(ref) if ref_len is None else ref_len
    cap = 2 * (len(query) + len(ref)) + 8
    cig = np.zeros(cap, dtype=np.uint32)
    sc = C.c_uint16()
    v = [C.c_int32() for _ in range(4)]
    m = _refj().ref_aligner_align(query, ref, n, C.byref(p), C.byref(sc), *[C.byref(x) for x in v],
                                  cig.ctypes.data, cap)
    assert m <= cap
    return (int(sc.value),) + tuple(int(x.value) for x in v), cig[:m].copy()


def ref_align_to_database(reads, entries, params=None, one_thread=True):
    """The reference's own alignToDatabase (src/SLAM.h:59-79). -> (alignments[ALIGN_DT], cigar_pool[u32])"""
    R = _refj()
    p = params or Params.default()
    rp, rl, k1 = _seq_arrays(reads)
    ep, el, k2 = _seq_arrays(entries)
    out, cig = C.c_void_p(), C.c_void_p()
    n_out, n_cig = C.c_uint64(), C.c_uint64()

    def run():
        with tempfile.TemporaryDirectory() as d:
            return R.ref_align_to_database(len(reads), C.cast(rp, C.c_void_p), rl.ctypes.data, len(entries),
                                           C.cast(ep, C.c_void_p), el.ctypes.data, C.byref(p), C.byref(out),
                                           C.byref(n_out), C.byref(cig), C.byref(n_cig), d.encode())
    rc = _one_thread(run) if one_thread else run()
    assert rc == 0
    n, nc = int(n_out.value), int(n_cig.value)
    al = np.frombuffer((C.c_char * (n * ALIGN_DT.itemsize)).from_address(out.value),
                       dtype=ALIGN_DT).copy() if n else np.zeros(0, dtype=ALIGN_DT)
    cg = np.frombuffer((C.c_char * (nc * 4)).from_address(cig.value),
                       dtype=np.uint32).copy() if nc else np.zeros(0, dtype=np.uint32)
    R.ref_free(out)
    R.ref_free(cig)
    return al, cg


def ref_sw_on_overlaps(overlaps, reads, entries, params=None):
    """The reference's own performSmithWatermanOnRange2 (src/SmithWaterman.h:184-233) on an OVERLAP_DT list
    grouped by read.  -> (alignments[ALIGN_DT], cigar_pool[u32])"""
    R = _refj()
    p = params or Params.default()
    ov = np.ascontiguousarray(overlaps, dtype=OVERLAP_DT)
    rp, rl, k1 = _seq_arrays(reads)
    ep, el, k2 = _seq_arrays(entries)
    out = np.zeros(len(ov) + 1, dtype=ALIGN_DT)
    cap = int(sum(2 * int(rl[int(r)]) + 8 for r in ov["read"])) + 8
    pool = np.zeros(cap, dtype=np.uint32)
    nc = C.c_uint64()
    rc = R.ref_sw_on_overlaps(ov.ctypes.data, len(ov), len(reads), C.cast(rp, C.c_void_p), rl.ctypes.data,
                              len(entries), C.cast(ep, C.c_void_p), el.ctypes.data, C.byref(p), out.ctypes.data,
                              pool.ctypes.data, cap, C.byref(nc))
    assert rc == 0
    return out[:len(ov)].copy(), pool[:int(nc.value)].copy()


ROW_FIELDS = ("read", "entry", "rel", "revcomp", "score", "ref_begin", "ref_end", "query_begin", "query_end", "cigar_len")


def compare_with_reference_rows(got, gcig, exp, ecig, read_bytes, entry_bytes):
    """A result set (rows[ALIGN_DT or the ABI's record], CIGAR pool) against a MULTI-THREADED run of the reference, whose
    only thread-dependent output is documented in DESIGN.md section 2: when the raw overlap list holds one
    (read, entry, rel) with both revComp values, `overlapSort` has no revComp in its key and `__gnu_parallel::sort` is
    unstable (src/Overlap.h:87-98, 289), so which flag survives std::unique -- and with it that row's alignment -- depends
    on the thread count.  Every other row must be identical in every field and CIGAR word.
    read_bytes(i) / entry_bytes(j) -> bytes: only called for the reads / entries of rows that differ, to rebuild THEIR raw
    overlap list with the restatement (extract -> sort -> findOverlaps) and check that each differing row is such a tie.
    -> dict(identical, alignments, cigar_ops, rows_differing, differing_rows_are_revcomp_ties[, why])"""
    out = {"identical": False, "alignments": int(len(got)), "cigar_ops": int(len(gcig)), "rows_differing": None,
           "differing_rows_are_revcomp_ties": None}
    if len(got) != len(exp):
        out["why"] = "row counts differ: %d vs %d" % (len(got), len(exp))
        return out
    for f in ("read", "entry", "rel"):
        if not (got[f] == exp[f]).all():
            out["why"] = "the (read, entry, rel) lists differ in `%s`" % f
            return out
    bad = np.zeros(len(got), dtype=bool)
    for f in ROW_FIELDS[3:]:
        bad |= got[f] != exp[f]
    # CIGAR words of the rows that agree so far (their lengths are equal): drop the differing rows' words on both sides
    glen, elen = got["cigar_len"].astype(np.int64), exp["cigar_len"].astype(np.int64)
    if int(glen.sum()) != len(gcig) or int(elen.sum()) != len(ecig):
        out["why"] = "a CIGAR pool does not have the length its rows add up to"
        return out
    if bad.any():
        gw = np.repeat(~bad, glen)
        ew = np.repeat(~bad, elen)
        ga, ea = np.asarray(gcig)[gw], np.asarray(ecig)[ew]
    else:
        ga, ea = np.asarray(gcig), np.asarray(ecig)
    if len(ga) != len(ea):
        out["why"] = "CIGAR pools of the agreeing rows differ in length"
        return out
    neq = ga != ea
    if neq.any():
        # name the rows: word index -> row
        keep_rows = np.flatnonzero(~bad)
        row_of_word = np.repeat(keep_rows, glen[keep_rows])
        bad[np.unique(row_of_word[neq])] = True
    nbad = int(bad.sum())
    out["rows_differing"] = nbad
    if nbad == 0:
        if not (got["cigar_off"] == exp["cigar_off"]).all():
            out["why"] = "cigar_off differs"
            return out
        out["identical"] = True
        out["differing_rows_are_revcomp_ties"] = True
        return out
    if nbad > 100000:
        out["why"] = "%d rows differ: not a handful of ties" % nbad
        return out
    rows = np.flatnonzero(bad)
    rids = sorted({int(x) for x in got["read"][rows]})
    eids = sorted({int(x) for x in got["entry"][rows]})
    sub_reads = [read_bytes(i) for i in rids]
    sub_entries = [entry_bytes(j) for j in eids]
    recs = np.concatenate([extract_kmers(sub_reads, False, 1), extract_kmers(sub_entries, True, 16)])
    raw = scan_overlaps(sort_kmers(recs), [len(r) for r in sub_reads])
    seen = {}
    for r, e, l, c in zip(raw["read"], raw["entry"], raw["rel"], raw["revcomp"]):
        seen.setdefault((rids[int(r)], eids[int(e)], int(l)), set()).add(int(c))
    # std::unique compares with the last KEPT element: a kept row stands for the raw rows within < 3 of it, so the flag of
    # the kept row is ambiguous when the raw list holds both flags AT the kept rel (the first of the equal-key run)
    ok = all(len(seen.get((int(got["read"][i]), int(got["entry"][i]), int(got["rel"][i])), ())) == 2 and
             int(got["revcomp"][i]) != int(exp["revcomp"][i]) for i in rows)
    out["differing_rows_are_revcomp_ties"] = bool(ok)
    out["identical"] = bool(ok)       # identical modulo the one tie the reference leaves open
    if not ok:
        out["why"] = "rows differ that are not revComp ties; first: %s" % [int(x) for x in rows[:5]]
    return out


# ---- the REAL batch loop metagenomicAnalysis_Low_Mem on files (oracle/_ref/libslam_ref.so) ----
REF_SLAM = os.path.join(_HERE, "_ref", "libslam_ref.so")
# the same program with ONE function swapped: alignToDatabase = the GPU operator behind the C ABI (ref_slam_driver.cpp built
# with -DKSLAM_REF_GPU_OPERATOR: INTEGRATION.md's patch to src/SLAM.h:59-79, compiled into the reference's own loop)
REF_SLAM_GPU = os.path.join(_HERE, "_ref", "libslam_gpu_ref.so")
_refslam = {}


class RefSlamParams(C.Structure):
    """the option globals of src/main.cpp:40-97, with its defaults"""
    _fields_ = [("match", C.c_uint32), ("mismatch", C.c_uint32), ("gap_open", C.c_uint32),
                ("gap_extend", C.c_uint32), ("score_threshold", C.c_uint32), ("num_sam_alignments", C.c_uint32),
                ("score_fraction", C.c_double), ("pseudo_assembly", C.c_int32), ("sam_xa", C.c_int32),
                ("just_align", C.c_int32), ("num_reads", C.c_uint32), ("num_reads_at_once", C.c_uint32),
                ("threads", C.c_int32)]

    @classmethod
    def default(cls, **kw):
        p = cls(2, 3, 5, 2, 0, 10, 0.95, 1, 0, 0, 0xFFFFFFFF, 10000000, 1)
        for k, v in kw.items():
            assert hasattr(p, k), k
            setattr(p, k, v)
        return p


def have_ref_slam():
    build()
    return os.path.exists(REF_SLAM)


def have_ref_slam_gpu():
    return os.path.exists(REF_SLAM_GPU)


def _refs(gpu=False):
    if gpu not in _refslam:
        R = C.CDLL(REF_SLAM_GPU if gpu else REF_SLAM)
        assert R.ref_slam_operator_is_gpu() == int(gpu)
        cp, u32, u64 = C.c_char_p, C.c_uint32, C.c_uint64
        R.ref_slam_index_reset.argtypes = []
        R.ref_slam_index_add_entry.argtypes = [cp, u64, cp, u32, u32]
        R.ref_slam_index_add_gene.argtypes = [cp, cp, cp, cp, cp, u32, u32, u32, C.c_int32]
        R.ref_slam_run.restype = C.c_int
        R.ref_slam_run.argtypes = [cp, cp, cp, cp, cp, cp, C.POINTER(RefSlamParams), cp]
        vp = C.c_void_p
        R.ref_slam_align_to_database.restype = C.c_int
        R.ref_slam_align_to_database.argtypes = [u64, vp, vp, C.POINTER(RefSlamParams), C.c_int32, C.POINTER(vp),
                                                 C.POINTER(u64), C.POINTER(vp), C.POINTER(u64),
                                                 C.POINTER(C.c_double), cp]
        R.ref_slam_free.argtypes = [vp]
        _refslam[gpu] = R
    return _refslam[gpu]


def ref_slam_set_index(entries, gpu=False):
    """entries: list of dicts {bases, locus_tag, taxonomy_id, genbank_id?, genes: [{name, locus_tag?, protein_id,
    product, reference?, gene_id?, start, stop, complement?}]} -- the GenbankIndex the reference's (unbuildable) archive
    reader would have returned.  gpu: for the library whose alignToDatabase is the GPU operator."""
    R = _refs(gpu)
    R.ref_slam_index_reset()
    for e in entries:
        R.ref_slam_index_add_entry(e["bases"], len(e["bases"]), e.get("locus_tag", b""), e.get("taxonomy_id", 0),
                                   e.get("genbank_id", 0))
        for g in e.get("genes", ()):
            R.ref_slam_index_add_gene(g.get("name", b""), g.get("locus_tag", b""), g.get("protein_id", b""),
                                      g.get("product", b""), g.get("reference", b""), g.get("gene_id", 0),
                                      g["start"], g["stop"], int(g.get("complement", 0)))


def ref_slam_set_index_arrays(db, offs, gpu=False):
    """The same injection for a database held as ONE uint8 array + entry offsets (the bench's 5 Gb database: no Python
    bytes object per entry)."""
    R = _refs(gpu)
    R.ref_slam_index_reset()
    db = np.ascontiguousarray(db, dtype=np.uint8)
    add = R.ref_slam_index_add_entry
    old = add.argtypes
    add.argtypes = [C.c_void_p] + list(old[1:])
    try:
        for i in range(len(offs) - 1):
            lo, hi = int(offs[i]), int(offs[i + 1])
            add(db.ctypes.data + lo, hi - lo, b"", 0, 0)
    finally:
        add.argtypes = old


_REF_LOG_PHASES = (("extract", "Getting k-mers from reads"), ("genome_kmers", "Getting k-mers from index"),
                   ("sort", "Sorting k-mers"), ("join", "Finding overlaps"), ("sw", "Performing pairwise Smith-Waterman"))


def _ref_log_phases(path, seconds):
    """Phase times of the LAST alignToDatabase call from the reference's own log.txt stamps `[t = 1.23s]\t<text>`
    (src/sequenceTools.h:171-179; 10 ms resolution): each phase runs from its stamp to the next phase's; the last one
    (Smith-Waterman) to the end of the call, `seconds` after the "Aligning reads to database" stamp.  None when the log is
    not there (the function-static Log opens ./log.txt at the process's FIRST log() call)."""
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return None
    stamps = []
    for ln in lines:
        if ln.startswith("[t = ") and "s]\t" in ln:
            t, _, text = ln[5:].partition("s]\t")
            try:
                stamps.append((float(t), text))
            except ValueError:
                pass
    starts = [i for i, (_, text) in enumerate(stamps) if text.startswith("Aligning reads to database")]
    if not starts:
        return None
    blk = stamps[starts[-1]:]
    t0 = blk[0][0]
    at = {}
    for name, text in _REF_LOG_PHASES:
        hit = [t for t, x in blk if x.startswith(text)]
        if not hit:
            return None
        at[name] = hit[0] - t0
    order = [n for n, _ in _REF_LOG_PHASES]
    out = {}
    for i, n in enumerate(order):
        end = at[order[i + 1]] if i + 1 < len(order) else seconds
        out[n] = round(max(end - at[n], 0.0), 2)
    return out


def ref_slam_align_to_database(reads_flat, read_offs, params=None, report_cigar=True, threads=0, gpu=False,
                               workdir=None):
    """The reference's OWN alignToDatabase (src/SLAM.h:59-79: the template itself, compiled from the header where it lies
    into oracle/_ref/libslam_ref.so) on a batch given as one flat uint8 array + offsets, against the index injected with
    ref_slam_set_index[_arrays].  threads: OpenMP threads (0 = leave; 1 = the deterministic run the goldens use).
    -> (alignments[ALIGN_DT], cigar_pool[u32], seconds inside alignToDatabase, phases from its log.txt or None)"""
    R = _refs(gpu)
    p = params or RefSlamParams.default()
    p.threads = int(threads)
    flat = np.ascontiguousarray(reads_flat, dtype=np.uint8)
    offs = np.ascontiguousarray(read_offs, dtype=np.uint64)
    out, cig = C.c_void_p(), C.c_void_p()
    n_out, n_cig, sec = C.c_uint64(), C.c_uint64(), C.c_double()
    tmp = None
    if workdir is None:
        tmp = tempfile.TemporaryDirectory()
        workdir = tmp.name
    try:
        rc = R.ref_slam_align_to_database(len(offs) - 1, flat.ctypes.data, offs.ctypes.data, C.byref(p),
                                          int(bool(report_cigar)), C.byref(out), C.byref(n_out), C.byref(cig),
                                          C.byref(n_cig), C.byref(sec), workdir.encode())
        assert rc == 0, "the reference's alignToDatabase failed (%d)" % rc
        phases = _ref_log_phases(os.path.join(workdir, "log.txt"), float(sec.value))
    finally:
        if tmp is not None:
            tmp.cleanup()
    n, nc = int(n_out.value), int(n_cig.value)
    al = np.frombuffer((C.c_char * (n * ALIGN_DT.itemsize)).from_address(out.value),
                       dtype=ALIGN_DT).copy() if n else np.zeros(0, dtype=ALIGN_DT)
    cg = np.frombuffer((C.c_char * (nc * 4)).from_address(cig.value),
                       dtype=np.uint32).copy() if nc else np.zeros(0, dtype=np.uint32)
    R.ref_slam_free(out)
    R.ref_slam_free(cig)
    return al, cg, float(sec.value), phases


def ref_slam_run(r1, r2, db_dir, out, sam, params=None, command_line=b"SLAM", workdir=None, gpu=False):
    """src/main.cpp:138-151 -> the reference's own metagenomicAnalysis_Low_Mem.  Paths are str; r2 / out / sam may be
    "".  The index comes from ref_slam_set_index (db_dir only supplies <db_dir>/taxDB)."""
    p = params or RefSlamParams.default()
    with tempfile.TemporaryDirectory() as d:
        rc = _refs(gpu).ref_slam_run(r1.encode(), r2.encode(), db_dir.encode(), out.encode(), sam.encode(),
                                  command_line, C.byref(p), (workdir or d).encode())
    assert rc == 0, "the reference threw"
