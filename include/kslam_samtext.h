/*
 * kslam_samtext.h -- the SAM records and the <out>_PerRead lines of a batch written ON THE GPU (SURVEY.md section 8f
 * row N1: the host tail's last stage moved next to the data).  Same library as kslam.h.
 *
 * Replaces, in the reference (citations into /root/reference/):
 *   writeSAMOutputPairs for every read pair     src/SLAM.h:234-239, src/SAM.h:278-305, 339-433, 443-512
 *   convertAlignmentsToIdentifiedTaxonomies_parallel's taxonomy id per read pair
 *                                               src/SLAM.h:243-246, src/MetagenomicResults.h:88-112,
 *                                               src/TaxonomyDatabase.h:185-223
 *   writePerReadResults                         src/MetagenomicResults.h:455-463
 * i.e. what kslam_tail_finish_write_rows and kslam_tail_classify (kslam_tail.h, kslam_taxonomy.h) do on the CPUs, with
 * the same bytes out.  The mapping quality's pow / log10 / ceil stay with the host's libm (src/SAM.h:464-499): the
 * library brings the log-probabilities of the rows whose quality depends on them to the host, evaluates them there and
 * hands one byte per row back before the text is written (csrc/samtext.hip).
 *
 * Use:  kslam_set_sam_annotations once (what the lines quote from the index; the taxonomy tree), kslam_set_sam_text to
 * switch the stage on for the pipelined lanes -- kslam_collect_batch then returns sam_text / per_read_text / tax_ids
 * (kslam_batch_result, kslam.h) and the read pairs' alignment pairs already in writeSAMOutputPairs' order -- or
 * kslam_sam_text for one resident batch.
 */
#ifndef KSLAM_SAMTEXT_H_
#define KSLAM_SAMTEXT_H_
#include "kslam_taxonomy.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Copies to the device what the SAM lines quote from the index (locus tags, taxonomy ids, the gene columns of
 * kslam_index_view) and, when taxdb is not NULL, the taxonomy tree for the per-read LCA.  index->bases is not read.
 * Needs an index of the same n_entries in the context. */
kslam_status kslam_set_sam_annotations(kslam_ctx *ctx, const kslam_index_view *index, const kslam_taxdb *taxdb);

/* What the pipelined lanes add after pairing + per-row walk (kslam_set_pairing must be on, qualities given):
 * want_sam != 0: the SAM records; want_per_read != 0: taxonomy ids + <out>_PerRead lines (needs a taxonomy tree in the
 * annotations).  num_alignments / sam_xa: --num-alignments / --sam-xa.  Both 0 switches the stage off (default).
 * A batch whose pseudo-assembly the device left to the host (kslam_pair_stats.stages_done) comes back without text. */
kslam_status kslam_set_sam_text(kslam_ctx *ctx, int want_sam, int want_per_read, uint32_t num_alignments, int sam_xa);

/* read identifiers for a batch that was loaded by columns (kslam_load_reads*): read i = concat[offsets[i] .. offsets[i+1]).
 * (kslam_submit_batch_fastq_text batches carry theirs.) */
kslam_status kslam_load_read_ids(kslam_ctx *ctx, const char *concat, const uint64_t *offsets);

/* One resident batch, after kslam_pair_screen (or the phased calls) and -- when the context reports CIGARs --
 * kslam_row_details_of_pairs: when the SAM records are asked for, sorts every read pair's alignment pairs in place (the
 * reference's per-pair std::sort: it only runs when a SAM file is written), then returns page-locked, library-owned copies (kslam_free_pinned each; an output pointer may be NULL to skip it):
 * the SAM records, the per-read lines and the taxonomy id per read pair. */
kslam_status kslam_sam_text(kslam_ctx *ctx, int paired, uint32_t num_alignments, int sam_xa, char **sam_text,
                            uint64_t *sam_len, char **per_read_text, uint64_t *per_read_len, uint32_t **tax_ids,
                            uint64_t *n_tax_ids);

#ifdef __cplusplus
}
#endif
#endif /* KSLAM_SAMTEXT_H_ */
