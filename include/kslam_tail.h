/*
 * kslam_tail.h -- C ABI of the host tail that follows the alignment hot path
 * (SURVEY.md section 8f row N1): score screen -> read pairing -> insert-size
 * screen -> score-fraction screen -> pseudo-assembly -> SAM records.
 *
 * Same library as kslam.h (k-slam_amd/libkslam_hip.so).  These entry points are
 * host-only: they never touch the GPU and take the kslam_overlap array +
 * cigar pool exactly as kslam_align_batch / kslam_fetch_results return them.
 *
 * They replace, in the reference (citations into /root/reference/), the body
 * of metagenomicAnalysis between alignToDatabase and the taxonomy step
 * (src/SLAM.h:101-133, same steps again in the low-memory loop
 * src/SLAM.h:209-239):
 *
 *   screenOverlapsByScoreThreshold          src/Overlap.h:329-341
 *   makePair / getPairsFromRead / getPairedOverlaps src/PairedOverlap.h:107-270
 *   getPerReadOverlaps (paired)             src/PairedOverlap.h:437-470
 *   getPerReadOverlaps (single end)         src/Overlap.h:303-327
 *   getDummyAlignmentPairsFromSingleEndReads src/PairedOverlap.h:280-298
 *   getMaxAllowedInsertSize                 src/PairedOverlap.h:314-360
 *   screenPairedAlignmentsByInsertSize      src/PairedOverlap.h:396-436
 *   screenPairedAlignmentsByScore           src/PairedOverlap.h:361-390
 *   pseudoAssembly                          src/PairedOverlap.h:480-582
 *   getCigarAndMD / getSAMFromPair          src/SAM.h:101-237, 352-433
 *   writeSAMOutputPairs / getHeader         src/SAM.h:443-512, 513-531
 *
 * Where the reference leaves the result to an unstable parallel sort
 * (__gnu_parallel::sort on the partial key (pair, entry, rel) in
 * getPairedOverlaps) this library fixes the order: ties keep the input order
 * (read ascending, i.e. the R1 overlap before the R2 overlap).  DESIGN.md
 * section 9 lists every such point.
 */
#ifndef KSLAM_TAIL_H_
#define KSLAM_TAIL_H_
#include "kslam.h"

#ifdef __cplusplus
extern "C" {
#endif

/* which stages run after pairing; the reference flow is all of them */
#define KSLAM_TAIL_INSERT_SCREEN 1u /* getMaxAllowedInsertSize + screen (paired only) */
#define KSLAM_TAIL_SCORE_SCREEN 2u  /* screenPairedAlignmentsByScore */
#define KSLAM_TAIL_PSEUDO_ASM 4u    /* pseudoAssembly + second score screen */
#define KSLAM_TAIL_ALL 7u
#define KSLAM_TAIL_PAIRING_ONLY 8u /* stop after pairing + grouping (stage-level tests) */
#define KSLAM_TAIL_GROUPS_SORTED 16u /* kslam_tail_finish_write_rows only: every read pair's alignment pairs are already in
                                        writeSAMOutputPairs' order (kslam_tail_finish_prepare put them there): do not sort again */

/* The globals the tail reads (src/Globals.h:31-42, set in src/main.cpp:40-97) */
typedef struct {
  uint32_t score_threshold;    /* --min-alignment-score      default 0 */
  uint32_t num_sam_alignments; /* --num-alignments           default 10 */
  double score_fraction;       /* --score-fraction-threshold default 0.95 */
  int32_t pseudo_assembly;     /* !--no-pseudo-assembly      default 1 */
  int32_t sam_xa;              /* --sam-xa                   default 0 */
  int32_t report_cigar;        /* reportCigar = SAM file requested */
  int32_t paired;              /* pairedData = R2 file given */
  uint32_t stages;             /* KSLAM_TAIL_* mask; 0 means KSLAM_TAIL_ALL
                                  (PSEUDO_ASM still needs pseudo_assembly) */
  int32_t threads;             /* 0 = one per hardware thread */
} kslam_tail_params;

/* std::vector<FASTQSequence> (src/FASTQsequence.h:38-50) by columns: item i of
 * a column is text[off[i] .. off[i+1]).  Paired batches are [R1 block | R2
 * block], mate of i is i + n_reads/2 (src/FASTQsequence.h:111-123). */
typedef struct {
  uint64_t n_reads;
  const char *bases;
  const uint64_t *bases_off;
  const char *quality; /* same lengths as bases (phred+33) */
  const uint64_t *quality_off;
  const char *ids; /* sequenceIdentifier, already stripped */
  const uint64_t *ids_off;
} kslam_reads_view;

/* The GenbankIndex fields the tail reads (src/GenbankTools.h:136-164): bases,
 * locusTag, taxonomyID and the gene list (codingSequence start/stop, geneName,
 * proteinID, product; src/GenbankTools.h:47-100; start/stop are the CDS's
 * uint32 fields reinterpreted as int, as getGene does, src/GenbankTools.h:170-185), genes in CSR form by entry.
 * n_genes == 0: all gene pointers may be NULL. */
typedef struct {
  uint64_t n_entries;
  const char *bases;
  const uint64_t *bases_off;
  const char *locus_tag;
  const uint64_t *locus_tag_off;
  const uint32_t *taxonomy_id;
  uint64_t n_genes;
  const uint64_t *gene_first; /* n_entries + 1 */
  const int32_t *gene_start;
  const int32_t *gene_stop;
  const char *gene_name;
  const uint64_t *gene_name_off; /* n_genes + 1 */
  const char *protein_id;
  const uint64_t *protein_id_off;
  const char *product;
  const uint64_t *product_off;
} kslam_index_view;

/* kslam_paired_overlap, kslam_read_pair and KSLAM_NO_OVERLAP: include/kslam.h */

typedef struct {
  uint64_t n_overlaps_in;
  uint64_t n_overlaps_screened; /* after the score threshold */
  uint64_t n_paired_initial;    /* PairedOverlaps out of pairing */
  uint64_t n_paired_final;
  uint64_t n_read_pairs;        /* read pairs (or reads) with >= 1 alignment */
  uint64_t n_insert_sizes;
  uint32_t max_insert_size;     /* getMaxAllowedInsertSize result */
  uint32_t threads;
  double ms_pairing;
  double ms_insert;
  double ms_screens;
  double ms_pseudo;
  double ms_sam;
  uint64_t sam_bytes;           /* SAM text produced */
} kslam_tail_stats;

/* message of the calling thread's last failed kslam_tail_* / kslam_sam_* call */
const char *kslam_tail_last_error(void);

/* src/SLAM.h:102-128 up to (not including) the SAM writer.  Outputs are malloc'ed,
 * release with kslam_free.  The overlaps must be in alignToDatabase's order
 * (read, entry, rel). */
kslam_status kslam_tail_pairs(const kslam_tail_params *params,
                              const kslam_reads_view *reads,
                              const kslam_overlap *overlaps, uint64_t n_overlaps,
                              kslam_read_pair **read_pairs, uint64_t *n_read_pairs,
                              kslam_paired_overlap **pairs, uint64_t *n_pairs,
                              kslam_tail_stats *stats);

/* writeSAMOutputPairs over every read pair, src/SAM.h:443-512 (the text the
 * reference streams to the SAM file after the header).  Sorts each read pair's
 * alignment pairs in place, as the reference does. */
kslam_status kslam_sam_records(const kslam_tail_params *params,
                               const kslam_reads_view *reads,
                               const kslam_index_view *index,
                               const kslam_overlap *overlaps, uint64_t n_overlaps,
                               const uint32_t *cigar_pool, uint64_t n_cigar,
                               const kslam_read_pair *read_pairs, uint64_t n_read_pairs,
                               kslam_paired_overlap *pairs, uint64_t n_pairs,
                               char **text, uint64_t *text_len, kslam_tail_stats *stats);

/* both of the above in one call: overlaps -> SAM records */
kslam_status kslam_tail_sam(const kslam_tail_params *params,
                            const kslam_reads_view *reads,
                            const kslam_index_view *index,
                            const kslam_overlap *overlaps, uint64_t n_overlaps,
                            const uint32_t *cigar_pool, uint64_t n_cigar,
                            char **text, uint64_t *text_len, kslam_tail_stats *stats);

/* The same with the text handed to a writer instead of one buffer: the drop-in
 * for the reference's `outFile << entry` stream (src/SAM.h:507-509).  `write` is
 * called on the calling thread, chunk after chunk in output order, each chunk a
 * whole number of lines; a non-zero return aborts with KSLAM_ERR_ARG.  The chunks
 * live in buffers the library keeps between calls, so a steady stream of batches
 * allocates nothing. */
typedef int (*kslam_write_fn)(void *user, const char *data, uint64_t len);
/* a ready-made writer: `user` points to an int holding an open file descriptor (the SAM file) */
int kslam_write_fd(void *user, const char *data, uint64_t len);
/* The same in the background: a writer object owns a thread that write()s to `fd`, batch after batch in order, while
 * the caller formats the next batch.  Pass kslam_write_queued as `write` and the writer as `user`: the
 * kslam_tail_*_write* entries then hand the batch's text buffers over instead of copying them (written buffers are
 * reused; at most two batches wait in the queue) and return as soon as the text is formatted.  Called directly,
 * kslam_write_queued copies `data` into the queue (a header).  kslam_sam_writer_close waits until everything is
 * written and reports a failed write (message in kslam_tail_last_error()); it does not close fd.
 * Why a thread: writes into one file serialise in the kernel whoever issues them (inode lock; ~6 GB/s into tmpfs on the
 * MI355X boxes = 70 ms per batch of 1 M read pairs, more than the formatting on sixteen CPUs), so the only way to hide
 * them is to overlap them with the next batch's work. */
typedef struct kslam_sam_writer kslam_sam_writer;
kslam_status kslam_sam_writer_open(int fd, kslam_sam_writer **out);
int kslam_write_queued(void *user, const char *data, uint64_t len);
/* text another stage produced (the GPU formatter's page-locked block, include/kslam_samtext.h) joins the queue as it is, in
 * order with everything else; release(user, data) is called once it has been written.  Blocks while two batches wait.
 * The block is consumed on EVERY path: a failing call (null writer, an earlier write error) releases it before it returns. */
kslam_status kslam_sam_writer_enqueue(kslam_sam_writer *writer, char *data, uint64_t len,
                                      void (*release)(void *user, void *data), void *user);
kslam_status kslam_sam_writer_close(kslam_sam_writer *writer, uint64_t *bytes_written, double *seconds_writing);
kslam_status kslam_tail_sam_write(const kslam_tail_params *params,
                                  const kslam_reads_view *reads,
                                  const kslam_index_view *index,
                                  const kslam_overlap *overlaps, uint64_t n_overlaps,
                                  const uint32_t *cigar_pool, uint64_t n_cigar,
                                  kslam_write_fn write, void *user, kslam_tail_stats *stats);

/* The same two with the per-row details of kslam_row_details (include/kslam.h):
 * NM, the log-probability and the MD text of every overlap record were computed
 * on the GPU, so the writer formats text and never reads index->bases (whose
 * 3 M scattered 150-byte windows per batch are what bounds the plain entry on
 * a multi-gigabyte database).  details[i] belongs to overlaps[i]; identical
 * output.  details == NULL: the plain walk. */
kslam_status kslam_tail_sam_rows(const kslam_tail_params *params,
                                 const kslam_reads_view *reads,
                                 const kslam_index_view *index,
                                 const kslam_overlap *overlaps, uint64_t n_overlaps,
                                 const uint32_t *cigar_pool, uint64_t n_cigar,
                                 const kslam_row_detail *details, const char *md_pool,
                                 uint64_t n_md, char **text, uint64_t *text_len,
                                 kslam_tail_stats *stats);
kslam_status kslam_tail_sam_write_rows(const kslam_tail_params *params,
                                       const kslam_reads_view *reads,
                                       const kslam_index_view *index,
                                       const kslam_overlap *overlaps, uint64_t n_overlaps,
                                       const uint32_t *cigar_pool, uint64_t n_cigar,
                                       const kslam_row_detail *details, const char *md_pool,
                                       uint64_t n_md, kslam_write_fn write, void *user,
                                       kslam_tail_stats *stats);

/* Everything up to the SAM writer's input computed on the GPU (kslam_pair_screen, include/kslam.h:
 * score screen, pairing, insert-size statistics, the two per-read-pair screens): this entry takes those
 * read pairs / alignment pairs (both are modified: pseudo-assembly rewrites scores, the sorts run in
 * place), runs pseudo-assembly + the second score screen when params ask for them, and writes the SAM
 * records.  details may be NULL (then the writer walks the database itself). */
kslam_status kslam_tail_finish_write_rows(const kslam_tail_params *params,
                                          const kslam_reads_view *reads,
                                          const kslam_index_view *index,
                                          const kslam_overlap *overlaps, uint64_t n_overlaps,
                                          const uint32_t *cigar_pool, uint64_t n_cigar,
                                          const kslam_row_detail *details, const char *md_pool,
                                          uint64_t n_md, kslam_read_pair *read_pairs,
                                          uint64_t n_read_pairs, kslam_paired_overlap *pairs,
                                          uint64_t n_pairs, kslam_write_fn write, void *user,
                                          kslam_tail_stats *stats);

/* The part of kslam_tail_finish_write_rows that CHANGES read_pairs / pairs, on its own, so that a caller can finish it
 * before anything reads the arrays concurrently (the batch loop's taxonomy thread, host/stream.cpp): pseudo-assembly +
 * the second score screen when params ask for them (pass params->pseudo_assembly = 0 when the device already ran them),
 * then -- when sort_groups != 0 -- the per-read-pair std::sort by combinedScore descending that writeSAMOutputPairs
 * starts with (src/SAM.h:446-450; an in-place sort in the reference too, so the classification that follows sees this
 * order, src/SLAM.h:234-246; without a SAM file the reference does not sort, hence the switch).  Afterwards call
 * kslam_tail_finish_write_rows with pseudo_assembly = 0 and stages | KSLAM_TAIL_GROUPS_SORTED: it then only reads.
 * stats (may be NULL): ms_pseudo, n_read_pairs, n_paired_final. */
kslam_status kslam_tail_finish_prepare(const kslam_tail_params *params, const kslam_reads_view *reads,
                                       const kslam_overlap *overlaps, uint64_t n_overlaps,
                                       kslam_read_pair *read_pairs, uint64_t n_read_pairs,
                                       kslam_paired_overlap *pairs, uint64_t n_pairs, int sort_groups,
                                       kslam_tail_stats *stats);

/* The tail keeps its work buffers (a few hundred bytes per overlap) between
 * calls; this returns them to the allocator.  Calls are serialised internally:
 * each one already spreads over all worker threads. */
void kslam_tail_release_buffers(void);

/* getHeader, src/SAM.h:513-531 */
kslam_status kslam_sam_header(const kslam_index_view *index, const char *command_line,
                              char **text, uint64_t *text_len);

#ifdef __cplusplus
}
#endif
#endif /* KSLAM_TAIL_H_ */
