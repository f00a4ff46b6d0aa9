/*
 * kslam_fastq.h -- C ABI of the FASTQ ingest in front of the alignment hot path
 * (SURVEY.md section 8f row N3).  Same library as kslam.h; host-only.
 *
 * Replaces, in the reference (citations into /root/reference/):
 *
 *   getSequencesFromFASTQFile         src/FASTQsequence.h:129-165
 *   getPairedSequencesFromFASTQFiles  src/FASTQsequence.h:110-123
 *   FASTQSequence::FASTQSequence      src/FASTQsequence.h:61-71  (identifier rule)
 *   safeGetline                       src/sequenceTools.h:45-73  (line endings)
 *
 * The reference pulls lines from a std::ifstream into a vector of three-string
 * objects; here the caller hands over the file's bytes (read or mmap'ed however
 * it likes) and gets the batch as three columns, which is the layout
 * kslam_load_reads (bases + bases_off) and the tail's kslam_reads_view take
 * as they are.  Records are found from a parallel line index, not by a serial
 * getline loop; the result is the reference's, including its corner cases:
 *   - a line ends at "\n", "\r\n" or a lone "\r";
 *   - records are exactly four lines, whatever the lines contain (no '@'/'+'
 *     check), and at the true end of the stream one more, empty, line is read
 *     (src/sequenceTools.h:65-68 sets only eofbit), which can complete a
 *     record whose quality line is missing;
 *   - identifier = header line without its first character, cut at the first
 *     space, then at the first '/'; a header of fewer than 2 characters gives
 *     an empty identifier; bases and quality are kept verbatim.
 */
#ifndef KSLAM_FASTQ_H_
#define KSLAM_FASTQ_H_
#include "kslam_tail.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A parsed batch; field for field a kslam_reads_view whose arrays the library
 * allocated (cast to const kslam_reads_view * to pass it on).  Release with
 * kslam_reads_free. */
typedef struct {
  uint64_t n_reads;
  char *bases;
  uint64_t *bases_off; /* n_reads + 1 */
  char *quality;
  uint64_t *quality_off;
  char *ids;
  uint64_t *ids_off;
} kslam_reads_columns;

/* One stream.  Parses up to max_reads records from text[0..len) (0 = no
 * limit).  at_eof != 0: text ends at the end of the file (the reference's
 * end-of-stream behaviour applies); at_eof == 0: text is a prefix of a longer
 * stream, only records whose four lines are all terminated are taken.
 * *consumed = where the reference's stream would stand for the next
 * getSequencesFromFASTQFile call (the low-memory loop, src/SLAM.h:193-207):
 * the end of the last record taken, or len when at_eof and fewer than
 * max_reads records were left.  threads: 0 = all usable CPUs.  Errors leave
 * their message in kslam_tail_last_error(). */
kslam_status kslam_fastq_parse(const char *text, uint64_t len, uint64_t max_reads, int at_eof,
                               int threads, kslam_reads_columns *out, uint64_t *consumed);

/* Two streams into one batch laid out [R1 block | R2 block], mate of i is
 * i + n_reads/2 (src/FASTQsequence.h:110-123).  Up to max_pairs records are
 * taken from each stream.  The reference only checks (n1 + n2) / n1 == 2 in
 * integer arithmetic and would mis-pair a longer R2 file; this call requires
 * n1 == n2 and returns KSLAM_ERR_ARG ("mismatch in R1 and R2 size") otherwise. */
kslam_status kslam_fastq_parse_pair(const char *r1, uint64_t len1, const char *r2, uint64_t len2,
                                    uint64_t max_pairs, int at_eof, int threads,
                                    kslam_reads_columns *out, uint64_t *consumed1,
                                    uint64_t *consumed2);

void kslam_reads_free(kslam_reads_columns *cols);

/* The same batch WITHOUT its two big columns: the records are indexed, identifiers and offsets are
 * built (out->ids, ids_off, bases_off, quality_off; out->bases and out->quality stay NULL), and
 * `layout` says where every read's bases and quality line lie in the two texts taken as one,
 * [r1 | r2] (positions of R2 fields are len1 + their position in r2).  kslam_submit_batch_fastq
 * (include/kslam.h) then uploads the TEXTS and cuts the bases / quality columns out on the GPU, where
 * they are needed, instead of the host copying 2 x 300 MB per batch into columns first; with the row
 * details of kslam_row_details the host tail never needs them.  A read whose quality line is not as
 * long as its bases line is refused here (KSLAM_ERR_ARG; the plain parser keeps both and the tail
 * rejects the batch).  Release out with kslam_reads_free and layout with kslam_fastq_layout_free. */
typedef struct {
  uint64_t n_reads;
  uint64_t *bases_at;   /* n_reads */
  uint64_t *quality_at; /* n_reads */
} kslam_fastq_layout;
kslam_status kslam_fastq_index_pair(const char *r1, uint64_t len1, const char *r2, uint64_t len2,
                                    uint64_t max_pairs, int at_eof, int threads,
                                    kslam_reads_columns *out, kslam_fastq_layout *layout,
                                    uint64_t *consumed1, uint64_t *consumed2);
void kslam_fastq_layout_free(kslam_fastq_layout *layout);

/* Batch boundaries without parsing: where the reference's stream would stand after reading max_records more
 * records from text[0..len) (getSequencesFromFASTQFile's loop, src/FASTQsequence.h:129-165, as the batch loop of
 * src/SLAM.h:193-207 calls it) -- the byte after the fourth line of record max_records.  *complete = 1 and
 * *end = that position when the text holds that many whole records; otherwise *end = len and *complete = at_eof
 * (at the true end of the stream the rest is the last, shorter batch; of a prefix, more bytes are needed).
 * A streaming host cuts its batches with this (terminators are only counted, 64 MiB per round on all usable
 * CPUs) and hands each window to kslam_submit_batch_fastq_text (include/kslam.h), which indexes it on the
 * device; batch k+1 can then be submitted before batch k's index exists. */
kslam_status kslam_fastq_batch_end(const char *text, uint64_t len, uint64_t max_records, int at_eof, int threads,
                                   uint64_t *end, int *complete);

#ifdef __cplusplus
}
#endif
#endif /* KSLAM_FASTQ_H_ */
