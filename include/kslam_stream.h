/*
 * kslam_stream.h -- the reference's batch loop as ONE call (the caller of the hot path).
 *
 * Replaces, in the reference (citations into /root/reference/), the body of
 *
 *   metagenomicAnalysis_Low_Mem(R1, R2, db, out, sam, readsPerGo, maxNumReads)      src/SLAM.h:159-268
 *
 * between opening the files and writing the end-of-run reports: the while loop of :193-250 --
 *   getPairedSequencesFromFASTQFiles (readsPerGo pairs)          :201-206   kslam_fastq_batch_end: the batch's bytes
 *   alignToDatabase                                              :209       }
 *   screenOverlapsByScoreThreshold, getPairedOverlaps,           :210-229   } kslam_submit_batch_fastq_text +
 *     getPerReadOverlaps, getMaxAllowedInsertSize, the screens,             }   kslam_set_pairing: on the GPU
 *     [pseudoAssembly + screen]                                  :230-233   }
 *   writeSAMOutputPairs for every read pair                      :234-239   kslam_tail_finish_write_rows -> the
 *                                                                            background writer (kslam_sam_writer)
 *   convertAlignmentsToIdentifiedTaxonomies_parallel, append     :243-249   kslam_tail_classify, kslam_taxreport_add_batch
 * -- with batch k+1 uploading while batch k is aligned and the host stage of batch k-1 (SAM text, LCA) runs on a
 * worker thread.  Batch boundaries are the reference's (pairs_per_batch records per stream per batch: the
 * insert-size limit is a per-batch statistic).  The end-of-run outputs stay with the caller, as in the reference:
 * <out>_PerRead is the concatenation of what this call writes to per_read_fd; kslam_taxonomy_summary and
 * kslam_taxreport_xml (include/kslam_taxonomy.h) give <out>_abbreviated and <out>.
 *
 * Same library as kslam.h; needs a context with an index (kslam_set_index).  Single-end data (the reference's
 * isPaired == false branch: getSequencesFromFASTQFile, getDummyAlignmentPairsFromSingleEndReads, :198-206, :228-233):
 * tail.paired = 0, r2 = NULL, len2 = 0; "pairs" in the fields below then reads "reads".  k-slam_amd/stream.py is the same loop in Python with
 * hooks for the tests; tests/test_gpu_stream.py holds both against the oracle chain and against each other.
 */
#ifndef KSLAM_STREAM_H_
#define KSLAM_STREAM_H_
#include "kslam_taxonomy.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  uint64_t pairs_per_batch; /* --num-reads-at-once, src/main.cpp:56 (default there: 10 000 000) */
  uint64_t max_pairs_total; /* --num-reads (maxNumReads); 0 = the whole files */
  kslam_tail_params tail;   /* the globals the tail reads; tail.paired = 0: single-end data, one text */
  int32_t sam_fd;           /* open descriptor of the SAM file, or -1 (the text is formatted and dropped) */
  int32_t per_read_fd;      /* open descriptor of <out>_PerRead, or -1 */
  const char *sam_header;   /* getHeader's text (kslam_sam_header), written first when sam_fd >= 0; may be NULL */
  uint64_t sam_header_len;
  uint32_t depth;           /* batches in flight; 0 = 3 */
  uint32_t host_threads;    /* 0 or 2: a batch's SAM text and its taxonomy part (LCA, _PerRead, report) on two threads at the same
                               time; 1: one after the other on one thread */
  uint32_t pool_threads;    /* threads of the host stages' parallel loops while the call runs; 0 = the CPUs the process may use
                               minus four (the SAM writer, the two lanes and the second host thread are busy next to the loops:
                               more runnable threads than a cgroup CPU quota has CPUs get the whole process throttled) */
  uint32_t passes;          /* 0 or 1: the files once; n: the two texts read n times over, as if they were n copies long
                               (timing runs: a longer stream without a longer text; max_pairs_total counts over all passes) */
} kslam_stream_params;

typedef struct {
  uint64_t n_batches;
  uint64_t n_pairs;               /* read pairs read from the files */
  uint64_t n_overlaps;            /* alignToDatabase rows over all batches */
  uint64_t n_read_pairs_aligned;  /* read pairs with at least one alignment pair left = _PerRead lines */
  uint64_t n_alignment_pairs;
  uint64_t sam_bytes, per_read_bytes;
  uint32_t first_max_insert_size; /* the first batch's insert-size limit */
  uint32_t batches_pseudo_on_host; /* batches whose pseudo-assembly the device left to the host (an entry too large) */
  double seconds;                 /* the whole call */
  double seconds_waiting_for_gpu, seconds_waiting_for_host_stage; /* main thread */
  double seconds_sam_text, seconds_classify, seconds_report;      /* host stage, summed over the batches (the SAM text on one
                                                                     thread, classification + report on another, concurrently) */
  double seconds_in_write;        /* writer thread inside write() */
  double seconds_cutting, seconds_submitting; /* main thread: kslam_fastq_batch_end, kslam_submit_batch_fastq_text */
  double seconds_closing;         /* main thread, after the last batch: the writer's queue drained, the file complete */
} kslam_stream_stats;

/* tax_ids: one taxonomy id per aligned read pair over all batches, in order (malloc'ed, kslam_free; NULL when taxdb
 * is NULL = --just-align).  report may be NULL.  Errors: those of the calls above; the message is in
 * kslam_last_error(ctx) or kslam_tail_last_error(), whichever stage failed (both are tried by the Python plumbing). */
kslam_status kslam_stream_classify(kslam_ctx *ctx, const kslam_index_view *index, const kslam_taxdb *taxdb,
                                   kslam_taxreport *report, const char *r1, uint64_t len1, const char *r2,
                                   uint64_t len2, const kslam_stream_params *params, uint32_t **tax_ids,
                                   uint64_t *n_tax_ids, kslam_stream_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* KSLAM_STREAM_H_ */
