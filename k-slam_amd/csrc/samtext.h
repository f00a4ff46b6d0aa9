// samtext.h -- interface of samtext.hip (the SAM records and <out>_PerRead lines of a batch written on the GPU)
#pragma once
#include "common.h"

namespace kslam {

// device copies of the columns the lines quote from the index (kslam_index_view, include/kslam_tail.h) and of the taxonomy
// tree (host/taxonomy.cpp's dense form), uploaded once per context
struct SamAnnot {
  uint64_t n_entries = 0, n_genes = 0;
  const uint8_t *locus = nullptr;
  const uint64_t *locus_off = nullptr;
  const uint32_t *tax = nullptr;          // taxonomy id per entry
  const uint64_t *gene_first = nullptr;   // n_entries + 1
  const int32_t *gene_start = nullptr, *gene_stop = nullptr;
  const uint8_t *gname = nullptr, *prot = nullptr, *prod = nullptr;
  const uint64_t *gname_off = nullptr, *prot_off = nullptr, *prod_off = nullptr;
  // taxonomy tree (absent with --just-align): node numbers, 0xFFFFFFFF = none
  uint64_t n_nodes = 0;
  const uint32_t *up = nullptr, *depth = nullptr, *node_tax = nullptr;
  const uint32_t *entry_node = nullptr;   // node of the entry's taxonomy id
};

struct SamInputs {   // one batch, all device pointers
  const kslam_overlap *ov = nullptr;
  const uint32_t *pool = nullptr;            // CIGAR words; nullptr when no CIGAR was asked for
  const kslam_row_detail *det = nullptr;     // per overlap record (the rows the alignment pairs refer to are filled in)
  const uint8_t *md_pool = nullptr;
  const uint8_t *ids = nullptr;              // read identifiers, ids_off[read] .. ids_off[read + 1]
  const uint64_t *ids_off = nullptr;
  const uint64_t *read_off = nullptr;        // base offsets = read lengths
};

struct SamParams {
  uint32_t num_alignments = 10;   // --num-alignments
  int32_t paired = 1, sam_xa = 0, report_cigar = 1;
  int32_t sort_groups = 1;        // run writeSAMOutputPairs' per-pair sort (the reference only sorts when it writes a SAM file)
  uint32_t mapq_unique = 50;      // the host libm's ceil(-10 log10(1e-5)): quality of a mate with ONE reported row
};

struct SamPlan {   // per read pair
  uint32_t n_rows, use1, use2, need;   // need: bit 0 = R1's qualities come from the host, bit 1 = R2's
};

struct SamWork {
  DevBuf plan, cnt_vals, cnt_segs, val_off, seg_off, scan_tmp, totals, vals, seg_len, mapq, text_len, text_off, text;
  DevBuf tax_ids, pr_len, pr_off, pr_text;
};

// First half: sorts every read pair's alignment pairs IN PLACE (the reference's per-pair std::sort), plans the rows and
// compacts the log-probabilities the host must turn into mapping qualities: W.vals (double, n_vals), W.seg_len (u32, n_segs).
// err_flags: bits of kslam_row_detail.flags met on a row that is reported (2) or whose probability matters (1).
void sam_plan(kslam_paired_overlap *d_recs, const kslam_read_pair *d_groups, uint64_t n_groups, const SamInputs &in, const SamParams &P,
              SamWork &W, uint64_t *n_vals, uint64_t *n_segs, uint32_t *err_flags, hipStream_t s);
// Second half, after W.mapq (u8, n_vals; same layout as W.vals) has been filled: the text into W.text.
void sam_format(const kslam_paired_overlap *d_recs, const kslam_read_pair *d_groups, uint64_t n_groups, const SamInputs &in,
                const SamAnnot &A, const SamParams &P, SamWork &W, uint64_t *text_bytes, hipStream_t s);
// per-read LCA into W.tax_ids (u32 per read pair) and the <out>_PerRead lines into W.pr_text
void per_read_device(const kslam_paired_overlap *d_recs, const kslam_read_pair *d_groups, uint64_t n_groups, const SamInputs &in,
                     const SamAnnot &A, SamWork &W, uint64_t *text_bytes, hipStream_t s);

}  // namespace kslam
