/*
 * kslam_oracle.h -- CPU ORACLE for the k-SLAM alignment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (k-slam_amd/csrc, libkslam_hip.so) never
 * links, loads or calls it and fails loudly when its HIP code is missing.
 *
 * What it is: a plain-C restatement, function by function, of the reference's
 * alignToDatabase() path (reference src/SLAM.h:59-79).  Every function cites
 * the reference file:line it follows.  Citations are into /root/reference/.
 *
 * Parity pinning status (see also DESIGN.md "Oracle"): EVERY function below is
 * pinned by the real reference compiled in place (oracle/_ref, built by
 * oracle/Makefile from /root/reference/src; no stand-in headers -- Boost is
 * avoided by two build-time line slices described in the Makefile):
 *     - ssw core: ssw_init / ssw_align / sw_sse2_byte / sw_sse2_word /
 *       banded_sw (src/ssw.c, libssw_ref.so)  -> orc_ssw_align, orc_banded_sw
 *     - k-mer codec, extraction, sort (src/KMer.h, libkmer_ref.so)
 *                                         -> orc_extract_kmers, orc_sort_kmers
 *     - join + dedupe: processPileUp / findOverlaps / findOverlaps_parallel
 *       (src/Overlap.h, included whole in libjoin_ref.so)
 *                                         -> orc_scan_overlaps, orc_find_overlaps
 *     - Aligner::Align + TranslateBase + BuildSwScoreMatrix + SetFlag +
 *       ConvertAlignment (src/ssw_cpp.cpp, libjoin_ref.so)   -> orc_align
 *     - performSmithWatermanOnRange2 (src/SmithWaterman.h, included whole)
 *                                         -> orc_sw_on_overlap
 *     - alignToDatabase (src/SLAM.h:59-79)  -> orc_align_to_database
 *   tests/test_oracle.py::test_against_real_* compare them on seeded cases when
 *   oracle/_ref exists; tests/golden/{ssw,kmer,join,align}_vectors.npz hold the
 *   reference's recorded answers for machines without it.
 *   One caveat the reference itself leaves open: overlapSort has no revComp
 *   in its key and the sort is unstable (src/Overlap.h:87-98, 289), so where
 *   the raw list holds the same (read, entry, rel) with both revComp values
 *   the survivor's flag is a tie.  The oracle (and the product) keep
 *   revComp == false; the comparisons flag those rows.
 *
 * More checkers live beside this one, each with its own status header:
 *   fastq_oracle.cpp  FASTQ reader (src/FASTQsequence.h) -- PINNED by the real
 *                     reference (ref_fastq_driver.cpp -> _ref/libfastq_ref.so)
 *   taxonomy_oracle.cpp  taxonomy tree + LCA -- PINNED (_ref/libtaxonomy_ref.so)
 *   tail_oracle.cpp   pairing .. SAM (src/PairedOverlap.h, src/SAM.h) -- PINNED
 *                     by the reference's own batch loop run on files
 *                     (ref_slam_driver.cpp -> _ref/libslam_ref.so;
 *                     tests/test_reference_loop.py, tests/golden/slam_loop.npz)
 */
#ifndef KSLAM_ORACLE_H_
#define KSLAM_ORACLE_H_
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* KMerAndData<uint64_t,32>, src/KMer.h:58-116 (16 bytes, little endian). */
typedef struct {
  uint64_t kmer;   /* kMerInt */
  uint32_t meta;   /* ID_isFromGB_RC: id | isFromGB<<31 | revComp<<30 */
  uint32_t offset; /* KMerData::offset */
} orc_kmer_rec;

/* OverlapTemp, src/Overlap.h:36-52 */
typedef struct {
  uint32_t read;
  uint32_t entry;
  int32_t rel;
  uint8_t revcomp;
  uint8_t pad[3];
} orc_overlap;

/* Overlap + StripedSmithWaterman::Alignment, src/Overlap.h:53-74,
 * src/ssw_cpp.h:10-87.  cigar lives in a pooled array. */
typedef struct {
  uint32_t read;
  uint32_t entry;
  int32_t rel;
  uint8_t revcomp;
  uint8_t pad;
  uint16_t score;
  int32_t ref_begin, ref_end, query_begin, query_end;
  uint32_t cigar_len;
  uint32_t pad2;
  uint64_t cigar_off;
} orc_alignment;

typedef struct {
  uint32_t match, mismatch, gap_open, gap_extend; /* src/Globals.h:27-31 */
  uint32_t score_threshold;                        /* src/Globals.h:31 */
  int32_t report_cigar;                            /* src/Globals.h:36 */
} orc_params;

/* raw SSW result, src/ssw.h:47-57 */
typedef struct {
  uint16_t score1;
  int32_t ref_begin1, ref_end1, read_begin1, read_end1;
  int32_t cigar_len;   /* number of ops written to cigar_out */
  int32_t status;      /* 0 ok; 1 = reference would hit "Trace back error" */
} orc_ssw_result;

/* ---- a-2, a-3: src/KMer.h:160-181, 246-280 ---- */
uint64_t orc_count_kmers(uint64_t len, unsigned gap);
uint64_t orc_extract_kmers(const char *bases, uint64_t len, int is_gb,
                           uint32_t id, unsigned gap, orc_kmer_rec *out);
uint64_t orc_extract_all(uint64_t n, const char *const *bases,
                         const uint64_t *lens, int is_gb, unsigned gap,
                         orc_kmer_rec *out);
/* ---- a-4: src/KMer.h:388-398 (offset asc added as the final tie-break) */
void orc_sort_kmers(orc_kmer_rec *recs, uint64_t n);
/* ---- a-5, a-6: src/Overlap.h:153-199, 230-246, 277-295 ----
 * returns number of overlaps written (after sort+unique); *n_raw gets the
 * pre-dedupe count.  out must hold orc_count_overlaps() entries. */
uint64_t orc_count_overlaps(const orc_kmer_rec *sorted, uint64_t n);
uint64_t orc_scan_overlaps(const orc_kmer_rec *sorted, uint64_t n,
                           const uint64_t *read_lens, orc_overlap *out);
uint64_t orc_find_overlaps(const orc_kmer_rec *sorted, uint64_t n,
                           const uint64_t *read_lens, orc_overlap *out,
                           uint64_t *n_raw);
/* ---- a-9..a-13: src/ssw_cpp.cpp:11-60,234-283; src/ssw.c ---- */
void orc_build_matrix(uint32_t match, uint32_t mismatch, int8_t mat[25]);
void orc_translate(const char *s, int32_t n, int8_t *out);
/* striped emulation of ssw_align on translated sequences */
void orc_ssw_align(const int8_t *read, int32_t read_len, const int8_t *ref,
                   int32_t ref_len, const int8_t mat[25], uint8_t gap_open,
                   uint8_t gap_extend, uint8_t flag, uint16_t filters,
                   int32_t filterd, uint32_t *cigar_out, int32_t cigar_cap,
                   orc_ssw_result *res);
/* plain (non-striped) Gotoh with the reference tie-breaks: the SPEC the HIP
 * kernels implement; compared against orc_ssw_align / the real ssw.c */
void orc_ssw_align_plain(const int8_t *read, int32_t read_len,
                         const int8_t *ref, int32_t ref_len,
                         const int8_t mat[25], uint8_t gap_open,
                         uint8_t gap_extend, uint8_t flag, uint16_t filters,
                         int32_t filterd, uint32_t *cigar_out,
                         int32_t cigar_cap, orc_ssw_result *res);
/* mode 0 = striped emulation, 1 = plain two-pass Gotoh, 2 = single forward pass
 * with origin tracking (what the HIP kernel runs) */
void orc_ssw_align_mode(const int8_t *read, int32_t read_len, const int8_t *ref,
                        int32_t ref_len, const int8_t mat[25], uint8_t gap_open,
                        uint8_t gap_extend, uint8_t flag, uint16_t filters,
                        int32_t filterd, uint32_t *cigar_out, int32_t cigar_cap,
                        orc_ssw_result *res, int mode);
int32_t orc_banded_sw(const int8_t *ref, const int8_t *read, int32_t ref_len,
                      int32_t read_len, int32_t score, uint32_t gap_open,
                      uint32_t gap_extend, int32_t band_width,
                      const int8_t *mat, int32_t n, uint32_t *cigar_out,
                      int32_t cigar_cap, int32_t *status);
/* Aligner::Align on ASCII, src/ssw_cpp.cpp:234-283 */
void orc_align(const char *query, int32_t query_len, const char *ref,
               int32_t ref_len, const orc_params *p, uint32_t *cigar_out,
               int32_t cigar_cap, orc_ssw_result *res, int plain);
/* ---- a-8: src/SmithWaterman.h:184-233 for ONE overlap ---- */
void orc_sw_on_overlap(const orc_overlap *ov, const char *read,
                       uint64_t read_len, const char *entry,
                       uint64_t entry_len, const orc_params *p,
                       orc_alignment *out, uint32_t *cigar_out,
                       int32_t cigar_cap, int plain);
/* ---- the whole path, src/SLAM.h:59-79 ----
 * Returns 0 on success.  Results are malloc'ed; release with orc_free. */
int orc_align_to_database(uint64_t n_reads, const char *const *reads,
                          const uint64_t *read_lens, uint64_t n_entries,
                          const char *const *entries,
                          const uint64_t *entry_lens, const orc_params *p,
                          int plain, orc_alignment **out, uint64_t *n_out,
                          uint32_t **cigar_pool, uint64_t *n_cigar,
                          double phase_seconds[6]);
void orc_free(void *p);

/* optional hook: route the SSW core through the real reference library
 * (oracle/_ref/libssw_ref.so: ssw_init/ssw_align) for cpu_baseline timing */
int orc_use_reference_ssw(const char *libpath);
int orc_num_threads(void);
void orc_set_num_threads(int n); /* OpenMP team size for the parallel phases */

#ifdef __cplusplus
}
#endif
#endif
