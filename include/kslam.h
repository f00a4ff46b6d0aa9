/*
 * kslam.h -- C ABI of the MI355X-native k-SLAM alignment hot path.
 *
 * One shared library (k-slam_amd/libkslam_hip.so, HIP for gfx950) exports
 * exactly these symbols.  Plain pointers and sizes only: no C++ types, no
 * torch types, no exceptions across the boundary.  Every call returns a
 * kslam_status; kslam_last_error() gives the message.
 *
 * The boundary replaces, in the reference (citations into /root/reference/):
 *
 *   alignToDatabase(reads, genbankIndex) -> std::vector<Overlap>
 *                                                     src/SLAM.h:59-79
 *
 * i.e. getKMersFromReads (src/KMer.h:373-381), GenbankIndex::getKMers
 * (src/GenbankTools.h:211-219), sortKMers (src/KMer.h:388-398),
 * findOverlaps_parallel (src/Overlap.h:277-295) and
 * performSmithWatermanOnRange_parallel (src/SmithWaterman.h:234-249).
 * INTEGRATION.md shows the reference-side binding.
 */
#ifndef KSLAM_H_
#define KSLAM_H_
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KSLAM_ABI_VERSION 10
#define KSLAM_K 32u /* src/Globals.h:25 */

typedef enum {
  KSLAM_OK = 0,
  KSLAM_ERR_ARG = 1,         /* bad argument */
  KSLAM_ERR_NO_DEVICE = 2,   /* no usable HIP device / HIP runtime failure */
  KSLAM_ERR_OOM = 3,         /* host or device allocation failed */
  KSLAM_ERR_UNSUPPORTED = 4, /* outside the supported envelope */
  KSLAM_ERR_STATE = 5,       /* call order (e.g. align before set_index) */
  KSLAM_ERR_INTERNAL = 6     /* device-side consistency check failed */
} kslam_status;

/* The implicit inputs of alignToDatabase: the scoring globals
 * src/Globals.h:27-31,36 as set from the CLI (src/main.cpp:44-55) and
 * reportCigar = (samFile != "") (src/SLAM.h:169). */
typedef struct {
  uint32_t match;           /* --match-score        default 2 */
  uint32_t mismatch;        /* --mismatch-penalty   default 3 */
  uint32_t gap_open;        /* --gap-open           default 5 */
  uint32_t gap_extend;      /* --gap-extend         default 2 */
  uint32_t score_threshold; /* --min-alignment-score default 0 */
  int32_t report_cigar;     /* reportCigar */
  int32_t device;           /* HIP device ordinal this context owns */
  uint32_t max_kmers_per_chunk; /* 0 = default; internal read sub-batching */
} kslam_params;

/* KMerAndData<uint64_t,32>, src/KMer.h:58-116: 16 bytes, little endian.
 * meta = id & 0x3FFFFFFF | isFromGB << 31 | revComp << 30 (src/KMer.h:65-67) */
typedef struct {
  uint64_t kmer;
  uint32_t meta;
  uint32_t offset;
} kslam_kmer;

/* OverlapTemp, src/Overlap.h:36-52 */
typedef struct {
  uint32_t read;   /* readPosInArray */
  uint32_t entry;  /* entryPosInArray */
  int32_t rel;     /* relativePosition */
  uint8_t revcomp; /* revComp */
  uint8_t pad[3];
} kslam_overlap_temp;

/* Overlap (src/Overlap.h:53-74) with its StripedSmithWaterman::Alignment
 * (src/ssw_cpp.h:10-87) flattened; the malloc'ed cigar becomes a slice
 * [cigar_off, cigar_off + cigar_len) of the batch's cigar pool (BAM packing
 * len << 4 | op, op M=0 I=1 D=2). */
typedef struct {
  uint32_t read;
  uint32_t entry;
  int32_t rel;
  uint8_t revcomp;
  uint8_t pad;
  uint16_t score; /* sw_score */
  int32_t ref_begin;
  int32_t ref_end;
  int32_t query_begin;
  int32_t query_end;
  uint32_t cigar_len;
  uint32_t pad2;
  uint64_t cigar_off;
} kslam_overlap;

/* Device time per phase of the last kslam_align_* call, from HIP events on
 * the context's own stream, in milliseconds; plus work counters. */
typedef struct {
  float ms_extract;     /* read k-mer extraction (with the membership filter when it is on) */
  float ms_sort;        /* k-mer radix sort (histograms + scans + scatter passes) */
  float ms_sort_scatter;/* the scatter launches of that sort only */
  float ms_join;        /* lookup join + overlap-key sort + dedupe */
  float ms_sw;          /* Smith-Waterman scores, ends and origins (planning + band tiers + full matrix) */
  float ms_cigar;       /* banded DP + traceback + finalize */
  float ms_total;       /* first kernel to last kernel */
  uint32_t sort_passes; /* radix passes executed per k-mer sort */
  uint64_t n_read_kmers;
  uint64_t n_genome_kmers;
  uint64_t n_overlaps_raw;   /* before sort + unique */
  uint64_t n_overlaps;       /* candidates that went through SW */
  uint64_t sw_cells;         /* forward-pass DP cells (query_len * window_len) */
  uint32_t n_chunks;
  uint32_t n_scatter_launches;
  uint64_t n_kmers_kept;     /* read k-mers that passed the genome-membership filter
                                (= n_read_kmers when the filter is off); these are sorted and joined */
} kslam_timings;

/* Device time of the last kslam_set_index / kslam_set_index_device by phase (HIP events on the context's stream,
 * milliseconds) and the size of its one-time sort -- the sort of the genome k-mer records north_star calls "giant"
 * (src/KMer.h:388-398 run on the genomes' records: once per index here, once per BATCH in the reference,
 * src/SLAM.h:64-65).  Algorithmic HBM bytes of that sort by SURVEY 8d's formula: n_genome_kmers x 16 x (2 x sort_passes + 1). */
typedef struct {
  uint64_t n_genome_kmers;
  uint32_t sort_passes;      /* 8-bit LSD passes executed: 8 over the 64-bit k-mer + those bytes of the meta word that can differ */
  uint32_t n_entries;
  float ms_encode_extract;   /* base coding + genome k-mer extraction (gap k / 2) */
  float ms_sort;             /* the radix sort of the 16-byte records: histograms, scans, scatters */
  float ms_tables;           /* key and {meta, offset} columns, bucket table, membership filter */
  float ms_total;
} kslam_index_stats;

typedef struct kslam_ctx kslam_ctx;

/* ---- lifecycle ------------------------------------------------------- */
uint32_t kslam_abi_version(void);
/* One line: ABI version, host compiler and libstdc++ the library was built with, and whether THAT
 * libstdc++'s std::sort permutes like csrc/gnu_sort.h (the device stages of rows N1 / N4 reproduce the
 * reference's std::sort permutations, src/PairedOverlap.h:369, 403, 527, with it; the host tail calls
 * std::sort itself).  kslam_check_std_sort runs that comparison (once per process: ~6000 tie-heavy,
 * ordered, all-equal and median-of-three-killer arrays) and returns 1 when every array came out
 * identical; when it returns 0, kslam_pair_screen* / kslam_set_pairing refuse the device stages
 * (KSLAM_ERR_UNSUPPORTED) and the tail has to run on the host.  Neither needs a device. */
const char *kslam_version(void);
int kslam_check_std_sort(uint64_t *n_arrays);
kslam_status kslam_create(const kslam_params *params, kslam_ctx **out);
void kslam_destroy(kslam_ctx *ctx);
const char *kslam_last_error(const kslam_ctx *ctx);
/* The KSLAM_* environment switches (kernel choice, buffer sizing, debug output; none of
 * them changes a result) are read ONCE, by kslam_create, never on the batch path.  kslam_reload_tuning reads
 * them again for this context and its worker lanes -- for variant tests and tuning scripts that flip a
 * switch between two batches; a production host never calls it.  Not while batches are in flight. */
kslam_status kslam_reload_tuning(kslam_ctx *ctx);
/* the HIP device ordinal the context owns (-1: none) */
int32_t kslam_ctx_device(const kslam_ctx *ctx);

/* A second context on the same device that BORROWS `primary`'s index (same device pointers; nothing of the
 * index is copied or freed by the sibling) and has its own stream, read batch and work buffers -- what the
 * pipelined lanes are made of.  For a host that keeps two batches resident at once (bench.py: a rank's shard in
 * one context, the whole batch for the batch-global tail in another).  Destroy it before the primary; it sees
 * a later kslam_set_index of the primary only after being re-created. */
kslam_status kslam_create_sibling(kslam_ctx *primary, kslam_ctx **out);

/* ---- the index: const GenbankIndex& (src/GenbankTools.h:187-220) ------
 * entries[j].bases, already upper-cased by the DB builder
 * (src/GenbankTools.h:256-258).  One-time: uploads the bases and builds the
 * resident, sorted genome k-mer list (gap = k/2, src/SLAM.h:64). */
kslam_status kslam_set_index(kslam_ctx *ctx, uint64_t n_entries,
                             const char *const *bases, const uint64_t *lens);
/* same, bases already in device memory: entry j occupies
 * d_bases[h_offsets[j] .. h_offsets[j+1]).  The copy runs on the context's
 * own stream, which knows nothing of the stream that produced d_bases: the
 * caller must have synchronized with the producer (hipStreamSynchronize /
 * hipDeviceSynchronize) before the call.  Same rule for
 * kslam_load_reads_device. */
kslam_status kslam_set_index_device(kslam_ctx *ctx, uint64_t n_entries,
                                    const void *d_bases,
                                    const uint64_t *h_offsets);
/* phases of the last index build on this context (a sibling / lane reports its primary's); KSLAM_ERR_STATE before one */
kslam_status kslam_index_build_stats(const kslam_ctx *ctx, kslam_index_stats *out);

/* ---- the operator: alignToDatabase, src/SLAM.h:59-79 -------------------
 * reads[i].bases used verbatim (src/FASTQsequence.h:46).  Output: overlaps
 * sorted by (read, entry, rel) after the reference's dedupe, each with its
 * alignment; cigar present iff report_cigar && score >= score_threshold
 * (src/ssw.c:924).  Buffers are owned by the library until
 * kslam_free_batch. */
kslam_status kslam_align_batch(kslam_ctx *ctx, uint64_t n_reads,
                               const char *const *bases, const uint32_t *lens,
                               kslam_overlap **out, uint64_t *n_out,
                               uint32_t **cigar_pool, uint64_t *n_cigar);
void kslam_free_batch(kslam_ctx *ctx, kslam_overlap *out, uint32_t *cigar_pool);

/* ---- the same operator, pipelined ---------------------------------------
 * The reference's batch loop (src/SLAM.h:194-241) is: read a batch, align it,
 * post-process it, next.  kslam_align_batch_async takes the batch (the reads are
 * copied out of the caller's memory before it returns) and hands back a ticket;
 * kslam_wait_batch blocks until that batch's result is in page-locked host
 * memory (same ownership as kslam_align_batch: give it back with
 * kslam_free_batch, from any thread).  Batches alternate between two internal
 * worker lanes with their own streams, so with two batches submitted the upload
 * of one and the download of another run under the kernels of a third phase:
 *     submit(0); for k: submit(k + 1); wait(k); <host tail of batch k>
 * Results are those of kslam_align_batch, batch by batch.  Not to be mixed with
 * a kslam_set_index call while tickets are outstanding. */
kslam_status kslam_align_batch_async(kslam_ctx *ctx, uint64_t n_reads,
                                     const char *const *bases, const uint32_t *lens,
                                     uint64_t *ticket);
kslam_status kslam_wait_batch(kslam_ctx *ctx, uint64_t ticket, kslam_overlap **out,
                              uint64_t *n_out, uint32_t **cigar_pool,
                              uint64_t *n_cigar);

/* ---- page-locked host memory ------------------------------------------------
 * Buffers a host fills itself (read columns, FASTQ text) reach the device by DMA, without a staging
 * copy, when they come from here: huge-page backed, registered with the runtime.  Needs no context
 * (but a HIP device).  bytes is what was asked for. */
void *kslam_host_alloc(uint64_t bytes);
void kslam_host_free(void *p, uint64_t bytes);

/* ---- per-row details for the SAM writer (row N1 of SURVEY section 8f) ----
 * getCigarAndMD (src/SAM.h:101-237) walks CIGAR + read + quality + entry bases
 * of every reported alignment; on the host that is a random walk through the
 * whole database.  With the batch's quality strings on the device as well
 * (kslam_load_qualities: same layout and offsets as the loaded reads),
 * kslam_row_details computes for every overlap record of the last result
 *   nm    = sequenceDiff.NM                       (SAM.h:150, 169, 181)
 *   logp  = sequenceDiff.logProbability           (SAM.h:146-152; the tables of
 *           SAM.h:33-48 come from the host's libm, the additions run in column
 *           order, so the double is the reference's)
 *   MD    = sequenceDiff.MD as text, md_pool[md_off .. md_off + md_len)
 * and include/kslam_tail.h takes them instead of walking the database itself.
 * flags: 1 = a quality character outside phred+33 0..99 in an aligned column,
 * 2 = the CIGAR runs past the read or the entry (both are errors the host
 * tail reports when it meets such a row). */
typedef struct {
  double logp;
  uint64_t md_off;
  uint32_t md_len;
  uint32_t nm;
  uint32_t flags;
  uint32_t pad;
} kslam_row_detail;
kslam_status kslam_load_qualities(kslam_ctx *ctx, const char *concat_quality);
kslam_status kslam_load_qualities_device(kslam_ctx *ctx, const void *d_concat_quality);
kslam_status kslam_row_details(kslam_ctx *ctx, uint64_t *n_md);
/* the same after kslam_pair_screen, for the overlap records its surviving alignment pairs refer to only (what the
 * SAM writer will ask for: about a third of the rows); the other rows' records are zero and have no MD text.
 * The pipelined lanes run this form when kslam_set_pairing switched the device pairing on. */
kslam_status kslam_row_details_of_pairs(kslam_ctx *ctx, uint64_t *n_md);
/* page-locked, library-owned copies; hand each back with kslam_free_pinned */
kslam_status kslam_take_row_details(kslam_ctx *ctx, kslam_row_detail **details,
                                    char **md_pool, uint64_t *n_md);
void kslam_free_pinned(kslam_ctx *ctx, void *p);

/* ---- the first half of the host tail on the GPU ----
 * From the overlap records of the last result, still in HBM: the score screen
 * (src/Overlap.h:329-341), read pairing (getPairedOverlaps, src/PairedOverlap.h:107-270; single end:
 * dummy pairs, :280-298), the batch-global insert-size statistics (getMaxAllowedInsertSize, :314-360)
 * and the two per-read-pair screens (:361-436) -- i.e. include/kslam_tail.h's kslam_tail_pairs with
 * the same stages, record for record (the reference's std::sort permutations included) -- and, with
 * KSLAM_TAIL_PSEUDO_ASM, pseudoAssembly (:480-582) and the second score screen as well, in place: the
 * records keep their positions, the groups' counts shrink, combined_score holds the chain scores.  The
 * host is left with the SAM text (kslam_tail_finish_write_rows, with the stages in stages_done masked
 * out of its params).
 * stages: KSLAM_TAIL_INSERT_SCREEN (1) | KSLAM_TAIL_SCORE_SCREEN (2) | KSLAM_TAIL_PSEUDO_ASM (4) of
 * include/kslam_tail.h. */
#define KSLAM_NO_OVERLAP 0xFFFFFFFFu

/* PairedOverlap, src/PairedOverlap.h:32-57; the two Overlap copies become
 * indices into the caller's kslam_overlap array */
typedef struct {
  uint32_t combined_score;
  uint32_t entry;
  int32_t ref_start;
  int32_t ref_end;
  uint32_t insert_size;
  uint32_t r1; /* KSLAM_NO_OVERLAP when hasR1 is false */
  uint32_t r2;
  uint32_t pad;
} kslam_paired_overlap;

/* ReadPairAndOverlaps, src/PairedOverlap.h:62-75: alignmentPairs is
 * pairs[first .. first + count) */
typedef struct {
  uint32_t r1_read;
  uint32_t r2_read;
  uint64_t first;
  uint64_t count;
} kslam_read_pair;

typedef struct {
  uint64_t n_overlaps_screened; /* after the score threshold */
  uint64_t n_paired_initial;    /* alignment pairs out of pairing */
  uint64_t n_insert_sizes;
  uint64_t n_read_pairs;        /* read pairs (or reads) with >= 1 alignment pair left */
  uint64_t n_pairs;             /* alignment pairs left */
  uint32_t max_insert_size;     /* getMaxAllowedInsertSize; UINT32_MAX when not computed */
  uint32_t stages_done;         /* KSLAM_TAIL_* bits of the stages the device ran: the host tail runs the rest
                                   (PSEUDO_ASM is left to the host when one entry holds more than 262144
                                   alignment pairs: one wavefront per entry is the wrong tool there) */
} kslam_pair_stats;
kslam_status kslam_pair_screen(kslam_ctx *ctx, int paired, uint32_t score_threshold,
                               double score_fraction, uint32_t stages, kslam_pair_stats *stats);
/* The same in pieces, for a batch whose read pairs are SHARDED over several GPUs (bench.py --gpus N).
 * Everything in the tail is per read pair except two steps, and those take gathered inputs:
 *   kslam_pair_phase_a   score screen + pairing on this shard's result; *d_inserts / *n_inserts = the shard's non-zero
 *                        insert sizes (device memory, valid until the next pairing call on this context)
 *   <the host gathers every shard's insert sizes into one device array, in any order>
 *   kslam_pair_phase_b   getMaxAllowedInsertSize of ALL of them (the limit is a statistic of the whole batch,
 *                        src/PairedOverlap.h:314-360: every shard computes the same value), then the two screens on this
 *                        shard's read pairs; *d_pairs / *n_pairs = its dense alignment-pair records (device memory)
 *   <the host gathers every shard's records in RANK order = read-pair order, the reference's bucket iteration order>
 *   kslam_pseudo_merged  pseudoAssembly (per entry across all read pairs of the batch, :480-582) on the gathered records
 *                        -- modified in place --, the new scores of records [own_base, own_base + n_pairs) copied back
 *                        into this shard's, second score screen on them.  stages_done lacks PSEUDO_ASM when an entry
 *                        is too large for the device path (the caller then runs that stage on the host, merged).
 * Afterwards kslam_take_pairs / kslam_row_details_of_pairs as after kslam_pair_screen.  The shards' read pairs,
 * alignment pairs and SAM text concatenated in rank order are those of one context that aligned the whole batch
 * (tests/test_gpu_multi.py). */
kslam_status kslam_pair_phase_a(kslam_ctx *ctx, int paired, uint32_t score_threshold, const int32_t **d_inserts,
                                uint64_t *n_inserts);
kslam_status kslam_pair_phase_b(kslam_ctx *ctx, const int32_t *d_all_inserts, uint64_t n_all, double score_fraction,
                                uint32_t stages, kslam_pair_stats *stats, const kslam_paired_overlap **d_pairs,
                                uint64_t *n_pairs);
kslam_status kslam_pseudo_merged(kslam_ctx *ctx, void *d_all_pairs, uint64_t n_all, uint64_t own_base,
                                 double score_fraction, kslam_pair_stats *stats);
/* pseudoAssembly with the ENTRIES partitioned over the `world` ranks instead (what include/kslam_comm.h and
 * k-slam_amd/dist.py run): the stage is independent per entry (src/PairedOverlap.h:495-574 -- one bucket per entry, one
 * walk per bucket), so entry e belongs to rank e mod world alone; 1 / world of the stage's work per rank, and 16 + 4
 * bytes per alignment pair travelling instead of world x 32.  After kslam_pair_phase_b on every rank:
 *   kslam_pseudo_route   *d_heads = this rank's records as 16-byte heads {combined_score, entry, ref_start, ref_end} (the
 *                        first half of kslam_paired_overlap), stably partitioned by entry mod world: counts[d] heads for
 *                        rank d, destination after destination (device memory, valid until the next route)
 *   <all-to-all: rank d lays the pieces it receives end to end in SOURCE-RANK order -- read-pair order, the reference's
 *    bucket iteration order (:486-493), which the tie order of its std::sort by refStart depends on>
 *   kslam_pseudo_owned   the stage on the n heads received (modified in place); *d_scores = their combined scores
 *                        afterwards, in the same order.  KSLAM_ERR_UNSUPPORTED when an entry is too large for the device
 *                        path: every rank must then leave the batch's stage to the host (kslam_comm.h exchanges a status
 *                        word so that all ranks fail together)
 *   <all-to-all back: the same pieces, 4 bytes per head, to where they came from, in sending order>
 *   kslam_pseudo_return  the n = sum(counts) scores into this rank's records, second score screen (src/SLAM.h:226-227)
 * The result equals kslam_pseudo_merged's and one context's (tests/test_gpu_multi.py, tests/test_gpu_tail.py). */
kslam_status kslam_pseudo_route(kslam_ctx *ctx, uint32_t world, const void **d_heads, uint64_t *counts);
kslam_status kslam_pseudo_owned(kslam_ctx *ctx, void *d_heads, uint64_t n, const uint32_t **d_scores);
kslam_status kslam_pseudo_return(kslam_ctx *ctx, const uint32_t *d_scores, uint64_t n, double score_fraction,
                                 kslam_pair_stats *stats);
/* the same on overlap records and read lengths handed in from the host (stage-level parity tests).  The records in
 * alignToDatabase's order -- sorted by read (src/Overlap.h:87-98), reads numbered below n_reads --, else KSLAM_ERR_ARG */
kslam_status kslam_pair_screen_overlaps(kslam_ctx *ctx, const kslam_overlap *overlaps, uint64_t n_overlaps,
                                        const uint32_t *read_lens, uint64_t n_reads, int paired,
                                        uint32_t score_threshold, double score_fraction,
                                        uint32_t stages, kslam_pair_stats *stats);
/* page-locked, library-owned copies of the last kslam_pair_screen* result; kslam_free_pinned each */
kslam_status kslam_take_pairs(kslam_ctx *ctx, kslam_read_pair **read_pairs, uint64_t *n_read_pairs,
                              kslam_paired_overlap **pairs, uint64_t *n_pairs);
/* Test hook for the sort pseudo-assembly stands on (csrc/wave_gnu_sort.h: libstdc++'s std::sort permutation
 * by one wavefront): segment i = keys[seg_off[i] .. seg_off[i+1]), at most 262144 keys (up to 4000 are sorted
 * in LDS, longer ones in global memory, as pseudo-assembly does with small and big entries), is sorted ascending by
 * key; perm[seg_off[i] + k] = the index within the segment of the element that ends up k-th. */
kslam_status kslam_debug_wave_sort(kslam_ctx *ctx, const int32_t *keys, const uint64_t *seg_off,
                                   uint64_t n_seg, uint32_t *perm);
/* what the pipelined lanes run after the alignment: stages == 0 switches the pairing off (default) */
kslam_status kslam_set_pairing(kslam_ctx *ctx, int paired, uint32_t score_threshold,
                               double score_fraction, uint32_t stages);

/* The pipelined entry with everything a SAM-writing host needs in one result:
 * quality may be NULL (then details / md_pool come back NULL). */
typedef struct {
  kslam_overlap *overlaps;
  uint64_t n_overlaps;
  uint32_t *cigar_pool;         /* with the device pairing on and qualities given, the batch gets
                                   the pairing BETWEEN the SW and the CIGAR stage, and CIGARs (like details) only for the
                                   records the alignment pairs refer to; the others carry cigar_len 0.  kslam_align_batch
                                   and the other entries return every CIGAR, as alignToDatabase does. */
  uint64_t n_cigar;
  kslam_row_detail *details;    /* one per overlap record; with the device pairing on, filled in for the records the
                                   alignment pairs refer to (kslam_row_details_of_pairs), zero for the others */
  char *md_pool;
  uint64_t n_md;
  kslam_read_pair *read_pairs;  /* NULL unless kslam_set_pairing switched the device pairing on */
  uint64_t n_read_pairs;
  kslam_paired_overlap *pairs;
  uint64_t n_pairs;
  kslam_pair_stats pair_stats;
  /* kslam_submit_batch_fastq_text only (else 0 / NULL): the batch as the host tail needs it */
  uint64_t n_reads;
  uint64_t *reads_bases_off; /* n_reads + 1 */
  char *reads_ids;
  uint64_t *reads_ids_off;   /* n_reads + 1 */
  uint64_t consumed1, consumed2;
  /* With KSLAM_TEXT_SAM in text_flags, overlaps / cigar_pool / details / md_pool are NULL (their counts are still
   * reported): the SAM writer was their only reader, and the records were written where they lie.
   * kslam_set_sam_text (include/kslam_samtext.h) switched on: the batch's SAM records and <out>_PerRead lines as written
   * on the GPU, the taxonomy id per read pair; NULL / 0 when off, or when this batch's text was left to the host
   * (text_flags says which).  With KSLAM_TEXT_PAIRS_SORTED the read pairs' alignment pairs above are already in
   * writeSAMOutputPairs' order (the device ran the per-pair sort): kslam_tail_finish_write_rows then takes
   * KSLAM_TAIL_GROUPS_SORTED and kslam_tail_finish_prepare must not sort again. */
  char *sam_text;
  uint64_t sam_text_len;
  char *per_read_text;
  uint64_t per_read_len;
  uint32_t *tax_ids;         /* n_read_pairs */
  uint32_t text_flags;       /* KSLAM_TEXT_* */
  uint32_t pad_;
} kslam_batch_result;
#define KSLAM_TEXT_PAIRS_SORTED 1u
#define KSLAM_TEXT_SAM 2u
#define KSLAM_TEXT_PER_READ 4u
kslam_status kslam_submit_batch(kslam_ctx *ctx, uint64_t n_reads, const char *const *bases,
                                const char *const *quality, const uint32_t *lens,
                                uint64_t *ticket);
/* the same for a batch that already lies in columns (what kslam_fastq_parse returns, include/kslam_fastq.h:
 * read i = bases[offsets[i] .. offsets[i+1]), quality likewise or NULL): nothing is copied at submission,
 * the columns go to the device straight from where they are -- by DMA when they are page-locked, which the
 * FASTQ parser's columns are once a context exists -- and must stay valid until the batch is collected. */
kslam_status kslam_submit_batch_columns(kslam_ctx *ctx, uint64_t n_reads, const char *bases,
                                        const char *quality, const uint64_t *offsets,
                                        uint64_t *ticket);
/* and for a batch that is still FASTQ text: the host has only INDEXED it (kslam_fastq_index_pair,
 * include/kslam_fastq.h: offsets = its bases_off, bases_at / quality_at = its layout); the two texts go
 * up as they are (by DMA when they lie in kslam_host_alloc memory) and the bases and quality columns are
 * cut out of them on the device.  Texts and arrays must stay valid until the batch is collected. */
kslam_status kslam_submit_batch_fastq(kslam_ctx *ctx, const char *r1, uint64_t len1,
                                      const char *r2, uint64_t len2, uint64_t n_reads,
                                      const uint64_t *offsets, const uint64_t *bases_at,
                                      const uint64_t *quality_at, uint64_t *ticket);
/* and with the record index built on the device as well: the host hands over the two texts and nothing
 * else.  The batch's read columns the host still needs come back in the result (reads_*: identifiers,
 * their offsets, the base offsets = lengths; read i's bases and quality stay on the device), with
 * consumed1 / consumed2 = where the reference's streams would stand (kslam_fastq_parse_pair's rule:
 * up to max_pairs records per stream, 0 = no limit; at_eof as there).  Same errors as
 * kslam_fastq_index_pair, reported by kslam_collect_batch.  r2 == NULL (and len2 == 0): single-end reads, one stream
 * (getSequencesFromFASTQFile, src/FASTQsequence.h:129-147); max_pairs then counts reads, and kslam_set_pairing's
 * `paired` must be 0. */
kslam_status kslam_submit_batch_fastq_text(kslam_ctx *ctx, const char *r1, uint64_t len1,
                                           const char *r2, uint64_t len2, uint64_t max_pairs,
                                           int at_eof, uint64_t *ticket);
kslam_status kslam_collect_batch(kslam_ctx *ctx, uint64_t ticket, kslam_batch_result *out);
void kslam_release_batch(kslam_ctx *ctx, kslam_batch_result *r);

/* ---- the same operator in three steps, for callers that keep the batch
 * resident in HBM (bench.py, multi-GPU sharding) --------------------------- */
kslam_status kslam_load_reads(kslam_ctx *ctx, uint64_t n_reads,
                              const char *concat, const uint64_t *offsets);
kslam_status kslam_load_reads_device(kslam_ctx *ctx, uint64_t n_reads,
                                     const void *d_concat,
                                     const uint64_t *h_offsets);
kslam_status kslam_align_resident(kslam_ctx *ctx, uint64_t *n_out,
                                  uint64_t *n_cigar);
kslam_status kslam_fetch_results(kslam_ctx *ctx, kslam_overlap *out,
                                 uint32_t *cigar_pool);
/* the last results in page-locked host buffers the library owns and reuses
 * (full PCIe rate, no allocation per batch); hand them back with
 * kslam_free_batch.  Several batches may be outstanding at once, and
 * kslam_free_batch may be called from ANOTHER thread than the one that is
 * inside kslam_take_results / kslam_align_batch (a host-tail worker handing
 * batch k back while batch k+1 is taken): the buffer pool is locked.  Every
 * other entry point of a context must be called from one thread at a time. */
kslam_status kslam_take_results(kslam_ctx *ctx, kslam_overlap **out, uint64_t *n_out,
                                uint32_t **cigar_pool, uint64_t *n_cigar);
/* device-to-device copy of the last results (for a RCCL gather) */
kslam_status kslam_copy_results_device(kslam_ctx *ctx, void *d_overlaps,
                                       void *d_cigar_pool);
kslam_status kslam_get_timings(const kslam_ctx *ctx, kslam_timings *out);
/* The other direction: records (in alignToDatabase's order, read ids and cigar_off in terms of the batch this
 * context has LOADED) and their CIGAR pool, already in device memory of this context's GPU, become its "last
 * result" -- as if kslam_align_resident had produced them.  For the batch-global steps after a read-sharded
 * alignment (src/SLAM.h:210-239 on the merged batch): the collecting rank loads the whole batch's reads and
 * qualities, adopts the gathered rows and runs kslam_pair_screen / kslam_row_details_of_pairs on them.  The
 * caller must have synchronized with whatever produced the two arrays. */
kslam_status kslam_adopt_results_device(kslam_ctx *ctx, const void *d_overlaps, uint64_t n_overlaps,
                                        const void *d_cigar_pool, uint64_t n_cigar);

/* ---- read-sharded batches: several GPUs of one node ----
 * The batch loop of the reference (src/SLAM.h:194-209) hands alignToDatabase
 * readsPerGo pairs at a time; the pairs of one batch are independent, so shard
 * g of N aligns pairs [g n / N, (g + 1) n / N) against its own replica of the
 * index and the only exchange is one gather of the result records.  Local
 * batches keep the block layout: [R1 of the shard's pairs | R2 of them].
 *
 * kslam_merge_shards_device is the collecting side: the shards' records and
 * CIGAR pools, gathered back to back into device memory of ctx's GPU by
 * whatever moved them (hipMemcpyPeerAsync in kslam_multi_*, RCCL send/recv
 * in a one-process-per-GPU host), become the batch-global result in the
 * reference's order -- byte for byte what one context returns for the whole
 * batch.  d_out_overlaps must hold sum(n_rows) records, d_out_cigars
 * sum(n_cigar) words; unpaired batches pass n_pairs = number of reads. */
typedef struct {
  uint64_t pair_lo, pair_hi; /* the batch's pairs [lo, hi) this shard aligned */
  uint64_t n_rows;           /* overlap records it produced */
  uint64_t n_cigar;          /* words of its CIGAR pool */
} kslam_shard;
kslam_status kslam_merge_shards_device(kslam_ctx *ctx, uint32_t n_shards,
                                       const kslam_shard *shards,
                                       uint64_t n_pairs, const void *d_overlaps,
                                       const void *d_cigar_pools,
                                       void *d_out_overlaps, void *d_out_cigars);

/* The sending side, for a gather that needs no merge: the last results of a
 * context whose batch was the local block layout of n_local_pairs pairs
 * ([R1 of them | R2 of them]).  kslam_shard_counts reports how its rows and
 * CIGAR words split into the R1 and the R2 block; once every shard's counts are
 * known (one all-gather of four numbers), kslam_export_shard_device writes the
 * shard's records in BATCH terms -- read ids re-based with pair_lo / n_pairs_total,
 * cigar_off pointing into the batch-global pool whose words of R1 rows start at
 * pool_base_r1 and of R2 rows at pool_base_r2 -- into four device buffers, which
 * may be the final places themselves (the shard on the collecting GPU) or send
 * buffers.  With bases = running sums over the shards (all R1 blocks first, then
 * all R2 blocks) the gathered arrays are byte for byte the single-context result. */
typedef struct {
  uint64_t n_rows, n_rows_r1;   /* overlap records; those of the R1 block */
  uint64_t n_cigar, n_cigar_r1; /* CIGAR words; those of the R1 rows */
} kslam_shard_counts;
kslam_status kslam_shard_counts_device(kslam_ctx *ctx, uint64_t n_local_pairs,
                                       kslam_shard_counts *out);
kslam_status kslam_export_shard_device(kslam_ctx *ctx, uint64_t n_local_pairs,
                                       uint64_t pair_lo, uint64_t n_pairs_total,
                                       uint64_t pool_base_r1, uint64_t pool_base_r2,
                                       void *d_rows_r1, void *d_rows_r2,
                                       void *d_pool_r1, void *d_pool_r2);

/* One process driving several devices: one context per entry of `devices`
 * (params->device is ignored; an ordinal may repeat, which puts two shards on
 * one GPU), the index replicated, kslam_multi_align_batch = shard, align on
 * every device concurrently, count exchange, every shard exports its records
 * in batch terms, peer copies into their final places on devices[0], one copy
 * to the host.  Same result and same ownership rules
 * as kslam_align_batch.  paired != 0: n_reads is even and reads[i], reads[i +
 * n_reads / 2] are mates (they stay on one shard). */
typedef struct kslam_multi kslam_multi;
kslam_status kslam_multi_create(const kslam_params *params, const int32_t *devices,
                                uint32_t n_devices, kslam_multi **out);
void kslam_multi_destroy(kslam_multi *m);
const char *kslam_multi_last_error(const kslam_multi *m);
kslam_status kslam_multi_set_index(kslam_multi *m, uint64_t n_entries,
                                   const char *const *bases, const uint64_t *lens);
kslam_status kslam_multi_align_batch(kslam_multi *m, uint64_t n_reads,
                                     const char *const *bases, const uint32_t *lens,
                                     int paired, kslam_overlap **out, uint64_t *n_out,
                                     uint32_t **cigar_pool, uint64_t *n_cigar);
void kslam_multi_free_batch(kslam_multi *m, kslam_overlap *out, uint32_t *cigar_pool);

/* ---- stage-level entry points (parity tests of SURVEY section 8a rows) - */
/* getKMers_parallel, src/KMer.h:190-241 */
kslam_status kslam_extract_kmers(kslam_ctx *ctx, uint64_t n,
                                 const char *const *bases,
                                 const uint64_t *lens, int is_from_genbank,
                                 uint32_t gap, kslam_kmer *out, uint64_t cap,
                                 uint64_t *n_out);
/* sortKMers, src/KMer.h:388-398 (kmer asc, meta desc; input order kept among
 * exact ties) */
kslam_status kslam_sort_kmers(kslam_ctx *ctx, kslam_kmer *recs, uint64_t n);
/* findOverlaps_parallel, src/Overlap.h:277-295, on the loaded reads against
 * the resident index */
kslam_status kslam_find_overlaps(kslam_ctx *ctx, kslam_overlap_temp **out,
                                 uint64_t *n_out, uint64_t *n_raw);
void kslam_free(void *p);

/* ---- device self-test / micro-benchmark of the k-mer radix sort -------------
 * Sorts n pseudo-random 16-byte records `iters` times (8 passes over the 64-bit
 * k-mer, the hot-path configuration), checks the order on the device and reports
 * milliseconds per sort and per scatter launch (HIP events).  Diagnostic only. */
kslam_status kslam_selftest_sort(kslam_ctx *ctx, uint64_t n, uint32_t iters,
                                 float *ms_per_sort, float *ms_per_scatter_launch,
                                 uint64_t *n_inversions);

#ifdef __cplusplus
}
#endif
#endif /* KSLAM_H_ */
