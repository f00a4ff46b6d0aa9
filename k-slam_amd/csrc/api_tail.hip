// api_tail.hip -- the C ABI, part 4: the widened path on the device: quality columns, per-row details
// (NM / MD / log-probability), pairing / insert-size statistics / screens / pseudo-assembly in one piece and in the pieces a
// sharded batch needs, SAM records and per-read lines written on the GPU (include/kslam_samtext.h), annotations, read ids.
#include "context.h"

namespace kslam_api {

// ---- the SAM records / per-read lines on the device (include/kslam_samtext.h, csrc/samtext.hip) ---------------------
// ceil(-10 log10(t)) stored into a uint8_t, src/SAM.h:502-506, with THIS host's libm (host/tail.cpp: mapq_of, same code)
inline uint8_t mapq_of(double prob, double sum) {
  double t = 1.0 - prob / sum;
  if (t <= 0.00001) t = 0.00001;
  double q = ceil(-10.0 * std::log10(t));
  if (std::isnan(q)) return 0;
  return (uint8_t)q;
}

void sam_stage_free(kslam_ctx *c, SamStage &S) {
  if (S.h_vals) pinned_put(c, S.h_vals);
  if (S.h_seg) pinned_put(c, S.h_seg);
  if (S.h_mapq) pinned_put(c, S.h_mapq);
  S.h_vals = nullptr;
  S.h_seg = nullptr;
  S.h_mapq = nullptr;
}

// first half (GPU): the per-pair sort, the plan, the log-probabilities the host must evaluate brought over
void sam_stage_plan(kslam_ctx *c, const kslam_ctx *owner, int paired, uint32_t num_alignments, int sam_xa, bool sort_groups, SamStage &S) {
  if (!(c->have_pairs && c->pairs_of_result))
    throw StatusError{KSLAM_ERR_STATE, "kslam_pair_screen has not been called for this result"};
  if (!owner->have_annot) throw StatusError{KSLAM_ERR_STATE, "kslam_set_sam_annotations has not been called"};
  if (owner->annot.n_entries != c->n_entries) throw StatusError{KSLAM_ERR_STATE, "the annotations belong to another index"};
  if (!c->have_ids) throw StatusError{KSLAM_ERR_STATE, "the batch has no read identifiers on the device (kslam_load_read_ids)"};
  if (c->prm.report_cigar && c->n_cig && !c->have_details)
    throw StatusError{KSLAM_ERR_STATE, "kslam_row_details_of_pairs has not been called for this result"};
  S.in.ov = c->res_ov.as<kslam_overlap>();
  S.in.pool = (c->prm.report_cigar && c->n_cig) ? c->res_cig.as<uint32_t>() : nullptr;
  S.in.det = c->have_details ? c->res_det.as<kslam_row_detail>() : nullptr;
  S.in.md_pool = c->d_md_pool;
  S.in.ids = c->d_ids;
  S.in.ids_off = c->d_ids_off;
  S.in.read_off = c->r_off.as<uint64_t>();
  S.P.num_alignments = num_alignments;
  S.P.paired = paired ? 1 : 0;
  S.P.sam_xa = sam_xa ? 1 : 0;
  S.P.report_cigar = c->prm.report_cigar ? 1 : 0;
  S.P.mapq_unique = mapq_of(1.0, 1.0);
  S.P.sort_groups = sort_groups ? 1 : 0;
  S.d_recs = const_cast<kslam_paired_overlap *>(c->pres.d_pairs);
  S.d_groups = c->pres.d_groups;
  S.n_groups = c->pres.n_read_pairs;
  uint32_t err = 0;
  sam_plan(S.d_recs, S.d_groups, S.n_groups, S.in, S.P, c->samw, &S.n_vals, &S.n_segs, &err, c->stream);
  if (err & 2u) throw StatusError{KSLAM_ERR_ARG, "cigar runs past the end of the read or the entry"};
  if (err & 1u) throw StatusError{KSLAM_ERR_ARG, "quality character outside phred+33 0..99"};
  S.h_vals = (double *)pinned_get(c, (S.n_vals + 1) * sizeof(double));
  S.h_seg = (uint32_t *)pinned_get(c, (S.n_segs + 1) * sizeof(uint32_t));
  S.h_mapq = (uint8_t *)pinned_get(c, S.n_vals + 16);
  if (S.n_vals) HIPCHK(hipMemcpyAsync(S.h_vals, c->samw.vals.p, S.n_vals * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (S.n_segs) HIPCHK(hipMemcpyAsync(S.h_seg, c->samw.seg_len.p, S.n_segs * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(stream_wait(c->stream));
}

// the host step: 10^logp, the sum in row order, the quality (src/SAM.h:464-499; host/tail.cpp: write_group's sums)
void sam_stage_mapq(SamStage &S) {
  if (!S.n_segs) return;
  std::vector<uint64_t> at(S.n_segs + 1, 0);
  for (uint64_t i = 0; i < S.n_segs; i++) at[i + 1] = at[i] + S.h_seg[i];
  const uint64_t grain = 2048, n_tasks = (S.n_segs + grain - 1) / grain;
  kslam_host::Pool::get().tasks(kslam_host::usable_cpus(), n_tasks, [&](size_t t) {
    for (uint64_t i = t * grain; i < std::min<uint64_t>(S.n_segs, (t + 1) * grain); i++) {
      const double *v = S.h_vals + at[i];
      uint8_t *q = S.h_mapq + at[i];
      const uint32_t n = S.h_seg[i];
      double prob[64], *pr = prob;
      std::vector<double> big;
      if (n > 64) {
        big.resize(n);
        pr = big.data();
      }
      double sum = 0;
      for (uint32_t k = 0; k < n; k++) {
        pr[k] = std::isinf(v[k]) ? 0.0 : std::pow(10, v[k]);   // a row without this mate: probability 0
        sum += pr[k];
      }
      for (uint32_t k = 0; k < n; k++) q[k] = mapq_of(pr[k], sum);
    }
  });
}

// second half (GPU): the qualities go up, the text is written; per-read lines
void sam_stage_kernels(kslam_ctx *c, const kslam_ctx *owner, SamStage &S, bool want_sam, bool want_per_read) {
  hipStream_t s = c->stream;
  S.text_bytes = S.pr_bytes = 0;
  if (want_sam) {
    if (S.n_vals) HIPCHK(hipMemcpyAsync(c->samw.mapq.p, S.h_mapq, S.n_vals, hipMemcpyHostToDevice, s));
    sam_format(S.d_recs, S.d_groups, S.n_groups, S.in, owner->annot, S.P, c->samw, &S.text_bytes, s);
  }
  if (want_per_read) {
    if (!owner->annot.up) throw StatusError{KSLAM_ERR_STATE, "the annotations hold no taxonomy tree"};
    per_read_device(S.d_recs, S.d_groups, S.n_groups, S.in, owner->annot, c->samw, &S.pr_bytes, s);
  }
  HIPCHK(stream_wait(s));
}
// ... and everything copied to page-locked memory (outside the lanes' compute token: the copy engine's work)
void sam_stage_fetch(kslam_ctx *c, SamStage &S, bool want_sam, bool want_per_read, char **sam_text, uint64_t *sam_len, char **pr_text,
                     uint64_t *pr_len, uint32_t **tax, uint64_t *n_tax) {
  hipStream_t s = c->stream;
  const uint64_t text_bytes = S.text_bytes, pr_bytes = S.pr_bytes;
  char *ht = nullptr, *hp = nullptr;
  uint32_t *hx = nullptr;
  try {
    if (want_sam) {
      ht = (char *)pinned_get(c, text_bytes + 64);
      if (text_bytes) HIPCHK(hipMemcpyAsync(ht, c->samw.text.p, text_bytes, hipMemcpyDeviceToHost, s));
    }
    if (want_per_read) {
      hp = (char *)pinned_get(c, pr_bytes + 64);
      hx = (uint32_t *)pinned_get(c, (S.n_groups + 1) * sizeof(uint32_t));
      if (pr_bytes) HIPCHK(hipMemcpyAsync(hp, c->samw.pr_text.p, pr_bytes, hipMemcpyDeviceToHost, s));
      if (S.n_groups) HIPCHK(hipMemcpyAsync(hx, c->samw.tax_ids.p, S.n_groups * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(stream_wait(s));
  } catch (...) {
    if (ht) pinned_put(c, ht);
    if (hp) pinned_put(c, hp);
    if (hx) pinned_put(c, hx);
    throw;
  }
  if (sam_text) *sam_text = ht; else if (ht) pinned_put(c, ht);
  if (sam_len) *sam_len = text_bytes;
  if (pr_text) *pr_text = hp; else if (hp) pinned_put(c, hp);
  if (pr_len) *pr_len = pr_bytes;
  if (tax) *tax = hx; else if (hx) pinned_put(c, hx);
  if (n_tax) *n_tax = want_per_read ? S.n_groups : 0;
}

template <typename T>
const T *annot_upload(kslam_ctx *c, const T *src, uint64_t n, hipStream_t s) {
  c->annot_bufs.emplace_back();
  DevBuf &b = c->annot_bufs.back();
  b.ensure((n + 1) * sizeof(T));
  if (n && src) HIPCHK(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, s));
  else if (n) HIPCHK(hipMemsetAsync(b.p, 0, n * sizeof(T), s));
  return b.as<T>();
}

void fill_pair_stats(const PairResult &r, kslam_pair_stats *st) {
  if (!st) return;
  memset(st, 0, sizeof *st);
  st->n_overlaps_screened = r.n_overlaps_screened; st->n_paired_initial = r.n_paired_initial;
  st->n_insert_sizes = r.n_insert_sizes; st->n_read_pairs = r.n_read_pairs; st->n_pairs = r.n_pairs;
  st->max_insert_size = r.max_insert_size;
  st->stages_done = r.stages_done;
}

}  // namespace kslam_api

extern "C" {

kslam_status kslam_load_qualities(kslam_ctx *c, const char *concat_quality) {
  return guarded(c, [&] {
    if (!c->have_reads) throw StatusError{KSLAM_ERR_STATE, "kslam_load_reads first: the quality strings share its offsets"};
    const uint64_t total = c->h_roff[c->n_reads];
    if (total && !concat_quality) throw StatusError{KSLAM_ERR_ARG, "null quality"};
    c->r_qual.ensure(total + 64);
    if (total) HIPCHK(hipMemcpyAsync(c->r_qual.p, concat_quality, total, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->r_qual.as<uint8_t>() + total, 0, 64, c->stream));
    HIPCHK(stream_wait(c->stream));
    c->have_qual = true;
    c->have_details = false;
  });
}

kslam_status kslam_load_qualities_device(kslam_ctx *c, const void *d_concat_quality) {
  return guarded(c, [&] {
    if (!c->have_reads) throw StatusError{KSLAM_ERR_STATE, "kslam_load_reads first: the quality strings share its offsets"};
    const uint64_t total = c->h_roff[c->n_reads];
    if (total && !d_concat_quality) throw StatusError{KSLAM_ERR_ARG, "null quality"};
    c->r_qual.ensure(total + 64);
    if (total) HIPCHK(hipMemcpyAsync(c->r_qual.p, d_concat_quality, total, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->r_qual.as<uint8_t>() + total, 0, 64, c->stream));
    HIPCHK(stream_wait(c->stream));
    c->have_qual = true;
    c->have_details = false;
  });
}

static kslam_status row_details_impl(kslam_ctx *c, uint64_t *n_md, bool of_pairs) {
  return guarded(c, [&] {
    if (!c->have_qual) throw StatusError{KSLAM_ERR_STATE, "kslam_load_qualities has not been called for this batch"};
    if (of_pairs && !(c->have_pairs && c->pairs_of_result))
      throw StatusError{KSLAM_ERR_STATE, "kslam_pair_screen has not been called for this result (pairs of records handed "
                                         "in through kslam_pair_screen_overlaps do not refer to its rows)"};
    if (!c->d_tables.p) {
      // matchTable / misMatchTable of src/SAM.h:33-48, with the host's libm (the values the host tail uses)
      double t[200];
      t[0] = std::log10(1.0 - std::pow(10.0, 1.0 / -10.0));
      t[100] = 1 / -10.0;
      for (int i = 1; i < 100; i++) {
        t[i] = std::log10(1.0 - std::pow(10.0, i / -10.0));
        t[100 + i] = i / -10.0;
      }
      c->d_tables.ensure(sizeof t);
      HIPCHK(hipMemcpyAsync(c->d_tables.p, t, sizeof t, hipMemcpyHostToDevice, c->stream));
      HIPCHK(stream_wait(c->stream));
    }
    c->res_det.ensure((c->n_res + 1) * sizeof(kslam_row_detail));
    const uint32_t *d_list = nullptr;
    uint64_t n_list = 0;
    if (of_pairs) {
      referenced_rows(c->pw, &c->pres, c->n_res, &d_list, &n_list, c->stream);
      if (!d_list) d_list = reinterpret_cast<const uint32_t *>(c->res_det.p);   // (no rows at all: any non-null list of length 0)
    }
    row_details(c->res_ov.as<kslam_overlap>(), c->n_res, c->res_cig.as<uint32_t>(), c->r_bases.as<uint8_t>(),
                c->r_qual.as<uint8_t>(), c->r_off.as<uint64_t>(), c->g_bases.as<uint8_t>(), c->g_off.as<uint64_t>(),
                c->d_tables.as<double>(), c->res_det.as<kslam_row_detail>(), c->detw, &c->d_md_pool, &c->n_md,
                &c->det_flags, c->stream, d_list, n_list);
    HIPCHK(stream_wait(c->stream));
    c->have_details = true;
    if (n_md) *n_md = c->n_md;
  });
}

kslam_status kslam_row_details(kslam_ctx *c, uint64_t *n_md) { return row_details_impl(c, n_md, false); }
kslam_status kslam_row_details_of_pairs(kslam_ctx *c, uint64_t *n_md) { return row_details_impl(c, n_md, true); }

kslam_status kslam_take_row_details(kslam_ctx *c, kslam_row_detail **details, char **md_pool, uint64_t *n_md) {
  if (!c || !details || !md_pool || !n_md) return KSLAM_ERR_ARG;
  *details = nullptr; *md_pool = nullptr; *n_md = 0;
  kslam_row_detail *hd = nullptr;
  char *hm = nullptr;
  kslam_status st = guarded(c, [&] {
    if (!c->have_details) throw StatusError{KSLAM_ERR_STATE, "kslam_row_details has not been called for this result"};
    hd = (kslam_row_detail *)pinned_get(c, (c->n_res + 1) * sizeof(kslam_row_detail));
    hm = (char *)pinned_get(c, c->n_md + 64);
    if (c->n_res) HIPCHK(hipMemcpyAsync(hd, c->res_det.p, c->n_res * sizeof(kslam_row_detail), hipMemcpyDeviceToHost, c->stream));
    if (c->n_md) HIPCHK(hipMemcpyAsync(hm, c->d_md_pool, c->n_md, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(stream_wait(c->stream));
  });
  if (st != KSLAM_OK) {
    if (hd) pinned_put(c, hd);
    if (hm) pinned_put(c, hm);
    return st;
  }
  *details = hd; *md_pool = hm; *n_md = c->n_md;
  return KSLAM_OK;
}


// the device stages stand on csrc/gnu_sort.h being this build's std::sort (host/selfcheck.cpp)
static void require_std_sort_parity() {
  if (!kslam_check_std_sort(nullptr))
    throw StatusError{KSLAM_ERR_UNSUPPORTED, std::string("this build's std::sort does not permute like csrc/gnu_sort.h (") +
                                                 kslam_version() + "): run pairing and screens on the host (include/kslam_tail.h)"};
}

kslam_status kslam_pair_screen(kslam_ctx *c, int paired, uint32_t score_threshold, double score_fraction, uint32_t stages,
                               kslam_pair_stats *stats) {
  return guarded(c, [&] {
    require_std_sort_parity();
    if (!c->have_reads) throw StatusError{KSLAM_ERR_STATE, "no batch loaded"};
    if (paired && (c->n_reads < 2 || (c->n_reads & 1)))
      throw StatusError{KSLAM_ERR_ARG, "paired data needs an even, non-zero number of reads ([R1 block | R2 block])"};
    if (c->n_res >= (1ull << 30)) throw StatusError{KSLAM_ERR_UNSUPPORTED, "2^30 or more overlaps in one batch"};
    c->have_pairs = c->pairs_of_result = c->phase_a_done = false;
    pair_and_screen(c->res_ov.as<kslam_overlap>(), c->n_res, c->r_len.as<uint32_t>(), c->n_reads, paired ? 1 : 0,
                    score_threshold, score_fraction, (stages & 1u) != 0, (stages & 2u) != 0, c->pw, c->sortws, &c->pres,
                    c->stream);
    if (stages & 4u) pseudo_and_rescreen(c->pw, &c->pres, score_fraction, c->sortws, c->stream);
    HIPCHK(stream_wait(c->stream));
    c->have_pairs = c->pairs_of_result = true;
    fill_pair_stats(c->pres, stats);
  });
}

// ---- the same in pieces, for read pairs sharded over several GPUs: the two batch-global steps take gathered inputs ----
kslam_status kslam_pair_phase_a(kslam_ctx *c, int paired, uint32_t score_threshold, const int32_t **d_inserts, uint64_t *n_inserts) {
  return guarded(c, [&] {
    require_std_sort_parity();
    if (!d_inserts || !n_inserts) throw StatusError{KSLAM_ERR_ARG, "null argument"};
    if (!c->have_reads) throw StatusError{KSLAM_ERR_STATE, "no batch loaded"};
    if (paired && (c->n_reads < 2 || (c->n_reads & 1)))
      throw StatusError{KSLAM_ERR_ARG, "paired data needs an even, non-zero number of reads ([R1 block | R2 block])"};
    if (c->n_res >= (1ull << 30)) throw StatusError{KSLAM_ERR_UNSUPPORTED, "2^30 or more overlaps in one batch"};
    c->have_pairs = c->pairs_of_result = c->phase_a_done = false;
    c->pw.route_n = ~0ull;
    pair_phase_a(c->res_ov.as<kslam_overlap>(), c->n_res, c->r_len.as<uint32_t>(), c->n_reads, paired ? 1 : 0, score_threshold, c->pw,
                 &c->pres, c->stream);
    HIPCHK(stream_wait(c->stream));
    c->phase_a_done = true;
    *d_inserts = c->pw.inserts.as<int32_t>();
    *n_inserts = c->pres.n_insert_sizes;
  });
}

kslam_status kslam_pair_phase_b(kslam_ctx *c, const int32_t *d_all_inserts, uint64_t n_all, double score_fraction, uint32_t stages,
                                kslam_pair_stats *stats, const kslam_paired_overlap **d_pairs, uint64_t *n_pairs) {
  return guarded(c, [&] {
    if (!c->phase_a_done) throw StatusError{KSLAM_ERR_STATE, "kslam_pair_phase_a has not been called for this result"};
    if (n_all && !d_all_inserts) throw StatusError{KSLAM_ERR_ARG, "null insert sizes"};
    c->phase_a_done = false;
    uint32_t limit = 0xFFFFFFFFu;
    const bool do_insert = (stages & 1u) != 0;
    if (do_insert && c->pw.paired) limit = insert_limit_device(d_all_inserts, n_all, c->pw, c->sortws, c->stream);
    pair_phase_b(c->res_ov.as<kslam_overlap>(), limit, score_fraction, do_insert, (stages & 2u) != 0, c->pw, &c->pres, c->stream);
    HIPCHK(stream_wait(c->stream));
    c->pres.n_insert_sizes = n_all;
    c->have_pairs = c->pairs_of_result = true;
    fill_pair_stats(c->pres, stats);
    if (d_pairs) *d_pairs = c->pres.d_pairs;
    if (n_pairs) *n_pairs = c->pres.n_pairs;
  });
}

kslam_status kslam_pseudo_merged(kslam_ctx *c, void *d_all_pairs, uint64_t n_all, uint64_t own_base, double score_fraction,
                                 kslam_pair_stats *stats) {
  return guarded(c, [&] {
    if (!(c->have_pairs && c->pairs_of_result)) throw StatusError{KSLAM_ERR_STATE, "kslam_pair_phase_b has not been called for this result"};
    if (n_all && !d_all_pairs) throw StatusError{KSLAM_ERR_ARG, "null records"};
    if (!pseudo_merged(c->pw, &c->pres, d_all_pairs, n_all, own_base, score_fraction, c->sortws, c->stream))
      throw StatusError{KSLAM_ERR_UNSUPPORTED, "the batch-global pseudo-assembly declined (2^28 or more alignment pairs in one batch): "
                                               "nothing was changed; gather the pairs on one host and run kslam_tail_finish_prepare there"};
    HIPCHK(stream_wait(c->stream));
    fill_pair_stats(c->pres, stats);
  });
}

kslam_status kslam_pseudo_route(kslam_ctx *c, uint32_t world, const void **d_heads, uint64_t *counts) {
  return guarded(c, [&] {
    if (!d_heads || !counts) throw StatusError{KSLAM_ERR_ARG, "null argument"};
    if (!(c->have_pairs && c->pairs_of_result)) throw StatusError{KSLAM_ERR_STATE, "kslam_pair_phase_b has not been called for this result"};
    pseudo_route(c->pw, &c->pres, world, d_heads, counts, c->sortws, c->stream);
    HIPCHK(stream_wait(c->stream));
  });
}

kslam_status kslam_pseudo_owned(kslam_ctx *c, void *d_heads, uint64_t n, const uint32_t **d_scores) {
  return guarded(c, [&] {
    require_std_sort_parity();
    if (!d_scores || (n && !d_heads)) throw StatusError{KSLAM_ERR_ARG, "null argument"};
    if (!pseudo_owned(c->pw, d_heads, n, d_scores, c->sortws, c->stream))
      throw StatusError{KSLAM_ERR_UNSUPPORTED, "the pseudo-assembly of this rank's entries declined (an entry with more than 262144 alignment pairs, or "
                                               "2^28 or more of them on one rank): nothing was changed; every rank must give the batch's stage to the host"};
    HIPCHK(stream_wait(c->stream));
  });
}

kslam_status kslam_pseudo_return(kslam_ctx *c, const uint32_t *d_scores, uint64_t n, double score_fraction, kslam_pair_stats *stats) {
  return guarded(c, [&] {
    if (!(c->have_pairs && c->pairs_of_result)) throw StatusError{KSLAM_ERR_STATE, "kslam_pair_phase_b has not been called for this result"};
    if (c->pw.route_n == ~0ull) throw StatusError{KSLAM_ERR_STATE, "kslam_pseudo_route has not been called for this result"};
    if (n && !d_scores) throw StatusError{KSLAM_ERR_ARG, "null scores"};
    pseudo_return(c->pw, &c->pres, d_scores, n, score_fraction, c->stream);
    HIPCHK(stream_wait(c->stream));
    fill_pair_stats(c->pres, stats);
  });
}

kslam_status kslam_pair_screen_overlaps(kslam_ctx *c, const kslam_overlap *overlaps, uint64_t n_overlaps,
                                        const uint32_t *read_lens, uint64_t n_reads, int paired, uint32_t score_threshold,
                                        double score_fraction, uint32_t stages, kslam_pair_stats *stats) {
  return guarded(c, [&] {
    require_std_sort_parity();
    if ((n_overlaps && !overlaps) || (n_reads && !read_lens)) throw StatusError{KSLAM_ERR_ARG, "null argument"};
    if (paired && (n_reads < 2 || (n_reads & 1)))
      throw StatusError{KSLAM_ERR_ARG, "paired data needs an even, non-zero number of reads ([R1 block | R2 block])"};
    if (n_overlaps >= (1ull << 30)) throw StatusError{KSLAM_ERR_UNSUPPORTED, "2^30 or more overlaps in one batch"};
    c->have_pairs = c->pairs_of_result = c->phase_a_done = false;
    c->pr_ov.ensure((n_overlaps + 1) * sizeof(kslam_overlap));
    c->pr_len.ensure((n_reads + 1) * sizeof(uint32_t));
    if (n_overlaps)
      HIPCHK(hipMemcpyAsync(c->pr_ov.p, overlaps, n_overlaps * sizeof(kslam_overlap), hipMemcpyHostToDevice, c->stream));
    if (n_reads) HIPCHK(hipMemcpyAsync(c->pr_len.p, read_lens, n_reads * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    pair_and_screen(c->pr_ov.as<kslam_overlap>(), n_overlaps, c->pr_len.as<uint32_t>(), n_reads, paired ? 1 : 0,
                    score_threshold, score_fraction, (stages & 1u) != 0, (stages & 2u) != 0, c->pw, c->sortws, &c->pres,
                    c->stream);
    if (stages & 4u) pseudo_and_rescreen(c->pw, &c->pres, score_fraction, c->sortws, c->stream);
    HIPCHK(stream_wait(c->stream));
    c->have_pairs = true;
    fill_pair_stats(c->pres, stats);
  });
}

kslam_status kslam_set_sam_annotations(kslam_ctx *c, const kslam_index_view *iv, const kslam_taxdb *taxdb) {
  return guarded(c, [&] {
    if (!iv) throw StatusError{KSLAM_ERR_ARG, "null index view"};
    if (!c->have_index) throw StatusError{KSLAM_ERR_STATE, "kslam_set_index has not been called"};
    if (iv->n_entries != c->n_entries) throw StatusError{KSLAM_ERR_ARG, "the index view has another number of entries than the index"};
    if (!iv->locus_tag_off || !iv->taxonomy_id) throw StatusError{KSLAM_ERR_ARG, "index view needs locus tags and taxonomy ids"};
    if (iv->n_genes && (!iv->gene_first || !iv->gene_start || !iv->gene_stop || !iv->gene_name_off || !iv->protein_id_off || !iv->product_off))
      throw StatusError{KSLAM_ERR_ARG, "index view has n_genes > 0 but no gene columns"};
    for (auto &b : c->annot_bufs) b.release();
    c->annot_bufs.clear();
    c->annot_bufs.reserve(24);
    c->have_annot = false;
    hipStream_t s = c->stream;
    const uint64_t E = iv->n_entries, G = iv->n_genes;
    SamAnnot A;
    A.n_entries = E;
    A.n_genes = G;
    A.locus = annot_upload(c, (const uint8_t *)iv->locus_tag, iv->locus_tag_off[E], s);
    A.locus_off = annot_upload(c, iv->locus_tag_off, E + 1, s);
    A.tax = annot_upload(c, iv->taxonomy_id, E, s);
    if (G) {
      A.gene_first = annot_upload(c, iv->gene_first, E + 1, s);
      A.gene_start = annot_upload(c, iv->gene_start, G, s);
      A.gene_stop = annot_upload(c, iv->gene_stop, G, s);
      A.gname = annot_upload(c, (const uint8_t *)iv->gene_name, iv->gene_name_off[G], s);
      A.gname_off = annot_upload(c, iv->gene_name_off, G + 1, s);
      A.prot = annot_upload(c, (const uint8_t *)iv->protein_id, iv->protein_id_off[G], s);
      A.prot_off = annot_upload(c, iv->protein_id_off, G + 1, s);
      A.prod = annot_upload(c, (const uint8_t *)iv->product, iv->product_off[G], s);
      A.prod_off = annot_upload(c, iv->product_off, G + 1, s);
    }
    std::vector<uint32_t> entry_node;
    if (taxdb) {
      uint64_t n_nodes = 0;
      const uint32_t *up = nullptr, *depth = nullptr, *node_tax = nullptr;
      if (kslam_taxdb_dense(taxdb, &n_nodes, &up, &depth, &node_tax) != KSLAM_OK) throw StatusError{KSLAM_ERR_ARG, kslam_tail_last_error()};
      A.n_nodes = n_nodes;
      A.up = annot_upload(c, up, n_nodes, s);
      A.depth = annot_upload(c, depth, n_nodes, s);
      A.node_tax = annot_upload(c, node_tax, n_nodes, s);
      entry_node.resize(E + 1);
      for (uint64_t e = 0; e < E; e++) entry_node[e] = kslam_taxdb_node(taxdb, iv->taxonomy_id[e]);
      A.entry_node = annot_upload(c, entry_node.data(), E, s);
    }
    HIPCHK(stream_wait(s));
    c->annot = A;
    c->have_annot = true;
  });
}

kslam_status kslam_set_sam_text(kslam_ctx *c, int want_sam, int want_per_read, uint32_t num_alignments, int sam_xa) {
  return guarded(c, [&] {
    if ((want_sam || want_per_read) && !c->have_annot) throw StatusError{KSLAM_ERR_STATE, "kslam_set_sam_annotations has not been called"};
    if (want_per_read && !c->annot.up) throw StatusError{KSLAM_ERR_STATE, "the annotations hold no taxonomy tree"};
    c->samtext.sam = want_sam != 0;
    c->samtext.per_read = want_per_read != 0;
    c->samtext.num_alignments = num_alignments;
    c->samtext.sam_xa = sam_xa;
  });
}

kslam_status kslam_load_read_ids(kslam_ctx *c, const char *concat, const uint64_t *offsets) {
  return guarded(c, [&] {
    if (!c->have_reads) throw StatusError{KSLAM_ERR_STATE, "no batch loaded"};
    if (!offsets || (c->n_reads && offsets[c->n_reads] && !concat)) throw StatusError{KSLAM_ERR_ARG, "null argument"};
    const uint64_t n = c->n_reads, bytes = n ? offsets[n] : 0;
    c->ids_buf.ensure(bytes + 64);
    c->ids_off_buf.ensure((n + 1) * sizeof(uint64_t));
    if (bytes) HIPCHK(hipMemcpyAsync(c->ids_buf.p, concat, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->ids_off_buf.p, offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(stream_wait(c->stream));
    c->d_ids = c->ids_buf.as<uint8_t>();
    c->d_ids_off = c->ids_off_buf.as<uint64_t>();
    c->have_ids = true;
  });
}

kslam_status kslam_sam_text(kslam_ctx *c, int paired, uint32_t num_alignments, int sam_xa, char **sam_text, uint64_t *sam_len,
                            char **per_read_text, uint64_t *per_read_len, uint32_t **tax_ids, uint64_t *n_tax_ids) {
  if (sam_text) *sam_text = nullptr;
  if (per_read_text) *per_read_text = nullptr;
  if (tax_ids) *tax_ids = nullptr;
  if (sam_len) *sam_len = 0;
  if (per_read_len) *per_read_len = 0;
  if (n_tax_ids) *n_tax_ids = 0;
  SamStage S;
  const kslam_status st = guarded(c, [&] {
    if (sam_text && !sam_len) throw StatusError{KSLAM_ERR_ARG, "sam_text without sam_len"};
    if ((per_read_text && !per_read_len) || (tax_ids && !n_tax_ids)) throw StatusError{KSLAM_ERR_ARG, "an output without its length"};
    sam_stage_plan(c, c, paired, num_alignments, sam_xa, sam_text != nullptr, S);
    sam_stage_mapq(S);
    const bool want_sam = sam_text != nullptr, want_pr = per_read_text != nullptr || tax_ids != nullptr;
    sam_stage_kernels(c, c, S, want_sam, want_pr);
    sam_stage_fetch(c, S, want_sam, want_pr, sam_text, sam_len, per_read_text, per_read_len, tax_ids, n_tax_ids);
  });
  if (c) sam_stage_free(c, S);
  return st;
}

kslam_status kslam_take_pairs(kslam_ctx *c, kslam_read_pair **read_pairs, uint64_t *n_read_pairs, kslam_paired_overlap **pairs,
                              uint64_t *n_pairs) {
  if (!c || !read_pairs || !n_read_pairs || !pairs || !n_pairs) return KSLAM_ERR_ARG;
  *read_pairs = nullptr; *pairs = nullptr; *n_read_pairs = 0; *n_pairs = 0;
  kslam_read_pair *hg = nullptr;
  kslam_paired_overlap *hp = nullptr;
  kslam_status st = guarded(c, [&] {
    if (!c->have_pairs) throw StatusError{KSLAM_ERR_STATE, "kslam_pair_screen has not been called for this result"};
    hg = (kslam_read_pair *)pinned_get(c, (c->pres.n_read_pairs + 1) * sizeof(kslam_read_pair));
    hp = (kslam_paired_overlap *)pinned_get(c, (c->pres.n_pairs + 1) * sizeof(kslam_paired_overlap));
    if (c->pres.n_read_pairs)
      HIPCHK(hipMemcpyAsync(hg, c->pres.d_groups, c->pres.n_read_pairs * sizeof(kslam_read_pair), hipMemcpyDeviceToHost, c->stream));
    if (c->pres.n_pairs)
      HIPCHK(hipMemcpyAsync(hp, c->pres.d_pairs, c->pres.n_pairs * sizeof(kslam_paired_overlap), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(stream_wait(c->stream));
  });
  if (st != KSLAM_OK) {
    if (hg) pinned_put(c, hg);
    if (hp) pinned_put(c, hp);
    return st;
  }
  *read_pairs = hg; *n_read_pairs = c->pres.n_read_pairs; *pairs = hp; *n_pairs = c->pres.n_pairs;
  return KSLAM_OK;
}

kslam_status kslam_debug_wave_sort(kslam_ctx *c, const int32_t *keys, const uint64_t *seg_off, uint64_t n_seg, uint32_t *perm) {
  return guarded(c, [&] {
    if (!seg_off || (n_seg && seg_off[n_seg] && (!keys || !perm))) throw StatusError{KSLAM_ERR_ARG, "null argument"};
    debug_wave_sort(keys, seg_off, n_seg, perm, c->stream);
  });
}

kslam_status kslam_set_pairing(kslam_ctx *c, int paired, uint32_t score_threshold, double score_fraction, uint32_t stages) {
  if (!c) return KSLAM_ERR_ARG;
  if (stages & 7u) {
    const kslam_status st = guarded(c, [&] { require_std_sort_parity(); });
    if (st != KSLAM_OK) return st;
  }
  std::lock_guard<std::mutex> lk(c->as_mu);
  c->pairing.paired = paired; c->pairing.thr = score_threshold; c->pairing.fraction = score_fraction;
  c->pairing.stages = stages & 7u;
  return KSLAM_OK;
}

}  // extern "C"
