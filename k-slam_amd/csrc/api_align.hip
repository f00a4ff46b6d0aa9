// api_align.hip -- the C ABI, part 3: alignToDatabase (reference src/SLAM.h:59-79) on the resident read batch, chunk by chunk:
//   extract read k-mers (+ membership filter) -> radix sort -> join against the resident sorted genome k-mer list ->
//   overlap sort + dedupe -> SW scores / ends -> banded CIGAR [-> pairing hook of the pipelined lanes].
// All device work runs on the context's own HIP stream; phases are bracketed with HIP events (kslam_timings).
#include "context.h"

namespace kslam_api {

__global__ void k_lens(const uint64_t *off, uint64_t n, uint32_t *len) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) len[i] = (uint32_t)(off[i + 1] - off[i]);
}

void finish_load_reads(kslam_ctx *c) {
  hipStream_t s = c->stream;
  const uint64_t n = c->n_reads;
  if (n >= (1ull << 30)) throw StatusError{KSLAM_ERR_UNSUPPORTED, "more than 2^30 reads (KMer.h:65 id field)"};
  uint64_t mx = 0, mx_at = 0;
  // per-read k-mer and segment counts as prefix sums, so that an align call plans its chunks with a
  // binary search instead of walking every read while the GPU waits
  c->h_kpre.assign(n + 1, 0);
  c->h_spre.assign(n + 1, 0);
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t len = c->h_roff[i + 1] - c->h_roff[i];
    if (len > mx) { mx = len; mx_at = i; }
    const uint64_t k = len >= KSLAM_K ? len - KSLAM_K + 1 : 0;  // gap 1, KMer.h:378
    c->h_kpre[i + 1] = c->h_kpre[i] + k;
    c->h_spre[i + 1] = c->h_spre[i] + (k + SEG_KMERS - 1) / SEG_KMERS;
  }
  // 13-bit score field of the packed DP values (and v_max_f64 reading 8188 and above as NaN patterns), 9-bit row / column
  // fields of the origin key: reads beyond either go through the plain kernels, in chunks of their own
  c->short_cap = (uint32_t)std::min<uint64_t>(511, 8187 / std::max<uint64_t>(1, (uint64_t)c->prm.match + 2 * c->prm.gap_extend));
  if (!scoring_in_envelope(c->prm)) c->short_cap = 0;   // scoring outside the envelope: every read is of the class the literal kernels take
  if (mx > 9000)
    throw StatusError{KSLAM_ERR_UNSUPPORTED, "reads longer than 9000 bases are not supported (read " + std::to_string(mx_at) +
                                                 " of the batch has " + std::to_string(mx) + ")"};
  c->class_runs.clear();
  c->max_short_len = 0;
  if (mx > c->short_cap) {
    bool prev_long = false;
    for (uint64_t i = 0; i < n; i++) {
      const uint64_t len = c->h_roff[i + 1] - c->h_roff[i];
      const bool is_long = len > c->short_cap;
      if (!is_long) c->max_short_len = std::max<uint32_t>(c->max_short_len, (uint32_t)len);
      if (i && is_long != prev_long) c->class_runs.push_back(i);
      prev_long = is_long;
    }
  } else {
    c->max_short_len = (uint32_t)mx;
  }
  c->max_read_len = (uint32_t)mx;
  c->r_off.ensure((n + 1) * sizeof(uint64_t));
  HIPCHK(hipMemcpyAsync(c->r_off.p, c->h_roff.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
  c->r_codes.ensure(c->h_roff[n] + 64);
  encode_bases(c->r_bases.as<uint8_t>(), c->r_codes.as<uint8_t>(), c->h_roff[n] + 48, s);
  c->r_len.ensure((n + 1) * sizeof(uint32_t));
  if (n) hipLaunchKernelGGL(k_lens, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->r_off.as<uint64_t>(), n,
                            c->r_len.as<uint32_t>());
  HIPCHK(stream_wait(s));
  c->have_reads = true;
  c->have_qual = false;      // a new batch: its quality strings have not been loaded
  c->have_details = false;
  c->have_ids = false;
  c->n_res = 0;
  c->n_cig = 0;
}

float ev_ms(hipEvent_t a, hipEvent_t b) {
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

// What the pipelined lanes ask align_resident to do between the SW stage and the CIGAR stage of a batch that is ONE
// chunk: the device pairing (and screens, pseudo-assembly) on the records with their final coordinates, so that
// the CIGAR stage -- and later the per-row walk -- only runs for the rows some surviving alignment pair refers
// to (36 % of the rows of the bench workload; the SAM writer asks for no others).

// the hot path on the resident reads; stop_after_join: only rows a-3..a-6
void align_resident(kslam_ctx *c, bool stop_after_join, uint64_t *n_raw_out, PairingHook *hook) {
  if (!c->have_index) throw StatusError{KSLAM_ERR_STATE, "kslam_set_index has not been called"};
  if (!c->have_reads) throw StatusError{KSLAM_ERR_STATE, "no reads loaded"};
  hipStream_t s = c->stream;
  kslam_timings tm{};
  tm.n_genome_kmers = c->n_gk;
  c->n_res = 0;
  c->n_cig = 0;
  c->have_details = false;
  c->have_pairs = false;
  c->pairs_of_result = false;
  c->phase_a_done = false;
  c->cells.ensure(sizeof(uint64_t));
  HIPCHK(hipMemsetAsync(c->cells.p, 0, sizeof(uint64_t), s));
  uint64_t n_raw_total = 0;

  // overlap key layout: read (chunk local) | entry | rel + bias | revcomp
  OverlapKeyLayout lay;
  lay.bits_entry = bits_for(c->n_entries ? c->n_entries - 1 : 0);
  lay.rel_bias = c->max_read_len;
  lay.bits_rel = bits_for(c->max_entry_len + c->max_read_len);
  if (lay.bits_entry + lay.bits_rel + 1 > 56)
    throw StatusError{KSLAM_ERR_UNSUPPORTED, "entry count x entry length too large for the packed overlap key"};
  const uint32_t max_bits_read = 63 - lay.bits_entry - lay.bits_rel;
  const uint64_t max_chunk_reads = max_bits_read >= 31 ? (1ull << 31) : (1ull << max_bits_read);
  // default: 2^30 read k-mers (4.2 M 150-bp pairs) per chunk = 32 GB of sort buffers, a ninth of the HBM
  const uint64_t max_chunk_kmers = c->prm.max_kmers_per_chunk ? c->prm.max_kmers_per_chunk : (1ull << 30);

  GenomeIndexDev g;
  g.key = c->gk_key.as<uint64_t>(); g.mo = c->gk_meta.as<uint2>();
  g.bucket = c->g_bucket.as<uint32_t>(); g.bucket_bits = c->bucket_bits; g.n = (uint32_t)c->n_gk;
  SwInputs in;
  in.read_bases = c->r_bases.as<uint8_t>(); in.read_off = c->r_off.as<uint64_t>();
  in.genome_bases = c->g_bases.as<uint8_t>(); in.genome_off = c->g_off.as<uint64_t>();
  in.read_codes = c->r_codes.as<uint8_t>(); in.genome_codes = c->g_codes.as<uint8_t>();
  SwParams sp;
  sp.match = (int32_t)c->prm.match; sp.mismatch = (int32_t)c->prm.mismatch;
  sp.gap_open = (int32_t)c->prm.gap_open; sp.gap_extend = (int32_t)c->prm.gap_extend;
  sp.score_threshold = c->prm.score_threshold; sp.report_cigar = c->prm.report_cigar;
  sp.striped = scoring_in_envelope(c->prm) ? 0 : 1;

  // The join looks every read k-mer up in the resident genome list (bucket table over the top
  // bucket_bits key bits + binary search), so the read list only has to be ordered as far as that
  // lookup benefits from locality: by the top key bytes covering the bucket bits.  Lower bytes
  // would only order records inside one bucket, which no later stage observes (the overlap list
  // is re-sorted by (read, entry, rel)).  KSLAM_SORT_BYTES overrides (8 = full 64-bit order).
  std::vector<SortPass> kpasses;
  {
    uint32_t nbytes = (std::min(c->bucket_bits, 24u) + 7) / 8;
    if (c->tune.sort_bytes >= 0) nbytes = (uint32_t)c->tune.sort_bytes;
    nbytes = std::min(8u, std::max(c->filter_bits ? 0u : 1u, nbytes));   // 0: look the survivors up unsorted
    for (uint32_t b = 8 - nbytes; b < 8; b++) kpasses.push_back(SortPass{b / 4, 8 * (b % 4), 0});
  }
  tm.sort_passes = (uint32_t)kpasses.size();

  uint64_t r0 = 0;
  const uint64_t n = c->n_reads;
  uint32_t tb_err_total = 0;
  // the lanes' order (PairingHook): every chunk up to its SW stage first, then the pairing on the whole batch, then
  // the CIGAR stage chunk by chunk for the rows the pairs refer to
  const bool lazy = hook && !stop_after_join && sp.report_cigar && !(hook->paired && (c->n_reads < 2 || (c->n_reads & 1)));
  struct Deferred { uint64_t first, m; uint32_t lmax; bool long_chunk; };
  std::vector<Deferred> deferred;
  while (r0 < n) {
    // ---- chunk [r0, r1) ----
    // as many reads as fit max_chunk_reads and max_chunk_kmers, at least one
    const uint64_t hi = std::min<uint64_t>(n, r0 + max_chunk_reads);
    const uint64_t *kp = c->h_kpre.data();
    uint64_t r1 = (uint64_t)(std::upper_bound(kp + r0 + 1, kp + hi + 1, kp[r0] + max_chunk_kmers) - kp) - 1;
    r1 = std::max(r1, r0 + 1);
    // a chunk holds reads of one class: short (the packed kernels) or long (the plain ones)
    const bool long_chunk = c->h_roff[r0 + 1] - c->h_roff[r0] > c->short_cap;
    if (!c->class_runs.empty()) {
      auto nx = std::upper_bound(c->class_runs.begin(), c->class_runs.end(), r0);
      if (nx != c->class_runs.end()) r1 = std::min<uint64_t>(r1, *nx);
    }
    uint32_t lmax_chunk = c->max_short_len;
    if (long_chunk) {
      lmax_chunk = 0;
      for (uint64_t i = r0; i < r1; i++) lmax_chunk = std::max<uint32_t>(lmax_chunk, (uint32_t)(c->h_roff[i + 1] - c->h_roff[i]));
    }
    Tuning tune_chunk = c->tune;
    if (long_chunk) {   // every CIGAR of such a chunk on the literal one-lane kernel (the others are sized by template)
      tune_chunk.cigar_sys_mask = 0;
      tune_chunk.cigar_reg = false;
      tune_chunk.cigar_dirs_lds = false;
    }
    const uint64_t nk_all = kp[r1] - kp[r0], nsegs = c->h_spre[r1] - c->h_spre[r0];
    if (nk_all >= (1ull << 32)) throw StatusError{KSLAM_ERR_UNSUPPORTED, "a single read chunk exceeds 2^32 k-mers"};
    const uint64_t nr = r1 - r0;
    tm.n_chunks++;
    tm.n_read_kmers += nk_all;
    lay.bits_read = bits_for(nr ? nr - 1 : 0);
    const uint64_t *d_off = c->r_off.as<uint64_t>() + r0;
    uint64_t *d_tot = c->totals.as<uint64_t>();

    // ---- a-3: read k-mer extraction ----
    HIPCHK(hipEventRecord(c->ev[0], s));
    uint64_t nk = nk_all;
    const bool use_filter = c->filter_bits && !long_chunk;   // (k_extract_filter packs a read into 36 words)
    if (use_filter && nk_all) {
      // only the k-mers the genome filter lets through are written; buffer sized from the last chunk,
      // rerun once with the exact size if it was too small
      c->totals.ensure(8 * sizeof(uint64_t));
      d_tot = c->totals.as<uint64_t>();
      uint64_t cap = std::max<uint64_t>(c->kept_last + c->kept_last / 4, nk_all / 8) + 4096;   // (a context's first batch: 10.6 % of the k-mers of reads that come from the database survive the filter; / 12 meant a rerun)
      cap = std::min(cap, nk_all);
      // the extraction also writes the first radix pass's digit of every survivor (radix_sort.hip: digit bytes)
      const bool with_digits = c->tune.sort_digit_bytes && kpasses.size() > 1 && kpasses[0].word < 2 && !kpasses[0].invert;
      for (int attempt = 0; attempt < 2; attempt++) {
        c->recs_a.ensure((cap + 1) * sizeof(uint4));
        if (with_digits) c->sortws.digits.ensure(cap + 64);
        extract_filtered(c->r_bases.as<uint8_t>(), d_off, (uint32_t)nr, c->g_filter.p, c->filter_bits,
                         c->recs_a.as<uint4>(), d_tot + 2, cap, c->tune, s, with_digits ? c->sortws.digits.as<uint8_t>() : nullptr,
                         with_digits ? kpasses[0].word : 0u, with_digits ? kpasses[0].shift : 0u);
        read_back(&nk, d_tot + 2, sizeof nk, s);
        if (nk <= cap) break;
        cap = nk;
      }
      c->kept_last = nk;
      c->recs_b.ensure((nk + 1) * sizeof(uint4));
      c->sortws.first_digits_ready = with_digits;
    } else {
      c->recs_a.ensure((nk + 1) * sizeof(uint4));
      c->recs_b.ensure((nk + 1) * sizeof(uint4));
      run_extract(c, c->r_bases.as<uint8_t>(), d_off, nr, 1, 0, nsegs, c->recs_a.as<uint4>());
    }
    tm.n_kmers_kept += nk;
    HIPCHK(hipEventRecord(c->ev[1], s));
    // ---- a-4: sort by k-mer ----
    c->sortws.use_digit_bytes = c->tune.sort_digit_bytes;
    c->sortws.ev_sc0 = c->evs0; c->sortws.ev_sc1 = c->evs1;
    const uint4 *sorted = (const uint4 *)radix_sort(c->recs_a.p, c->recs_b.p, nk, 4, kpasses.data(),
                                                    (int)kpasses.size(), c->sortws, s, c->ev[2], c->ev[3],
                                                    &tm.n_scatter_launches);
    c->sortws.ev_sc0 = nullptr; c->sortws.ev_sc1 = nullptr;
    c->sortws.first_digits_ready = false;
    HIPCHK(hipEventRecord(c->ev[4], s));
    // ---- a-5: join ----
    const uint64_t n_tiles = (nk + JOIN_TILE - 1) / JOIN_TILE;
    c->block_tot.ensure((n_tiles + 1) * sizeof(uint32_t));
    c->block_base.ensure((n_tiles + 1) * sizeof(uint64_t));
    c->scan_tmp.ensure(scan_tmp_bytes(std::max<uint64_t>(n_tiles, 1)));
    c->totals.ensure(8 * sizeof(uint64_t));
    d_tot = c->totals.as<uint64_t>();
    uint64_t raw = 0;
    if (nk) {
      // single-pass join into a buffer sized from the last batch; rerun once if it was too small
      const uint64_t have_cap = c->ovk_a.cap / sizeof(uint64_t);
      const uint64_t guess = (use_filter ? 4 * nk : nk / 6) + 1024;
      uint64_t cap = have_cap > guess ? have_cap - 1 : guess;   // never grows a big-enough buffer
      for (int attempt = 0; attempt < 2; attempt++) {
        c->ovk_a.ensure((cap + 1) * sizeof(uint64_t));
        if (c->tune.join_merge && !kpasses.empty())
          join_fill_merge(sorted, (uint32_t)nk, g, 8u * (uint32_t)kpasses.size(), c->r_len.as<uint32_t>() + r0, d_tot, cap, lay,
                          c->ovk_a.as<uint64_t>(), s);
        else
          join_fill_single_pass(sorted, (uint32_t)nk, g, c->r_len.as<uint32_t>() + r0, d_tot, cap, lay,
                                c->ovk_a.as<uint64_t>(), s);
        read_back(&raw, d_tot, sizeof raw, s);
        if (raw <= cap) break;
        cap = raw + raw / 8;
      }
    }
    n_raw_total += raw;
    if (raw >= (1ull << 32)) throw StatusError{KSLAM_ERR_UNSUPPORTED, "more than 2^32 raw overlaps in one chunk; lower max_kmers_per_chunk"};
    uint64_t m = 0;
    if (raw) {
      c->ovk_b.ensure((raw + 1) * sizeof(uint64_t));
      // ---- a-6: sort by (read, entry, rel[, revcomp]) + unique ----
      const uint32_t key_bits = lay.bits_read + lay.bits_entry + lay.bits_rel + 1;
      const uint32_t key_bytes = (key_bits + 7) / 8;
      // the low bits -- rel and revComp -- only order the keys inside a (read, entry) group of a few keys: radix passes over the
      // bits above them, then join.hip's group_order -- unless a recent chunk of this context has shown groups too long for that
      // (reads in tandem repeats), or the switch is off
      const uint32_t low_bits = lay.bits_rel + 1;
      bool grouped = c->tune.join_group_order && c->group_route_pause == 0 && low_bits >= 16 && low_bits < key_bits;
      if (c->group_route_pause) c->group_route_pause--;
      c->flags.ensure((raw + 1) * sizeof(uint32_t));
      c->pos.ensure((raw + 1) * sizeof(uint32_t));
      c->scan_tmp.ensure(scan_tmp_bytes(raw));
      const uint64_t *keys = nullptr;
      for (int attempt = 0; attempt < 2; attempt++) {
        std::vector<SortPass> op;
        if (grouped)   // digits of the whole 64-bit key from bit low_bits on (radix_sort.hip: word 2)
          for (uint32_t sh = low_bits; sh < key_bits; sh += 8) op.push_back(SortPass{2u, sh, 0});
        else
          for (uint32_t b = 0; b < key_bytes; b++) op.push_back(SortPass{b / 4, 8 * (b % 4), 0});
        void *src = attempt == 0 ? c->ovk_a.p : const_cast<uint64_t *>(keys);           // (second attempt: any order of the same keys will do)
        void *dst = src == c->ovk_a.p ? c->ovk_b.p : c->ovk_a.p;
        keys = (const uint64_t *)radix_sort(src, dst, raw, 2, op.data(), (int)op.size(), c->sortws, s, nullptr, nullptr, nullptr);
        uint32_t *d_big = reinterpret_cast<uint32_t *>(d_tot + 3);
        const uint64_t *sorted_by_high = keys;
        if (grouped) {   // ordered keys AND flags in one kernel, into the other buffer (the sort's output stays intact)
          uint64_t *other = keys == c->ovk_a.as<uint64_t>() ? c->ovk_b.as<uint64_t>() : c->ovk_a.as<uint64_t>();
          HIPCHK(hipMemsetAsync(d_big, 0, sizeof(uint64_t), s));
          group_order(keys, raw, lay, other, c->flags.as<uint32_t>(), d_big, s);
          keys = other;
        } else {
          dedupe_flags(keys, raw, lay, c->flags.as<uint32_t>(), s);
        }
        exclusive_scan_u32(c->flags.as<uint32_t>(), c->pos.as<uint32_t>(), raw, d_tot, c->scan_tmp.p, s);
        uint64_t back[4] = {0, 0, 0, 0};
        read_back(back, d_tot, sizeof back, s);      // [0] survivors, [3] "a group was too long"
        m = back[0];
        if (!grouped || back[3] == 0) break;
        c->group_route_pause = 32;                      // this chunk again, all passes; the next 32 chunks go there directly
        grouped = false;
        keys = sorted_by_high;                          // (a permutation of the chunk's keys, untouched by the attempt)
      }
      ensure_keep(c->res_ov, (c->n_res + m + 1) * sizeof(kslam_overlap), c->n_res * sizeof(kslam_overlap), s);
      dedupe_compact(keys, c->flags.as<uint32_t>(), c->pos.as<uint32_t>(), raw, lay, (uint32_t)r0,
                     c->res_ov.as<kslam_overlap>() + c->n_res, s);
    }
    HIPCHK(hipEventRecord(c->ev[5], s));
    uint64_t ncig = 0;
    if (m && !stop_after_join) {
      kslam_overlap *cand = c->res_ov.as<kslam_overlap>() + c->n_res;
      // ---- a-8..a-12: scores and ends ----
      uint32_t *band0;
      if (lazy) {   // the batch's band array: one slice per chunk, kept until the CIGAR stage runs
        ensure_keep(c->band0_all, (c->n_res + m + 1) * sizeof(uint32_t), c->n_res * sizeof(uint32_t), s);
        band0 = c->band0_all.as<uint32_t>() + c->n_res;
      } else {
        c->band0.ensure((m + 1) * sizeof(uint32_t));
        band0 = c->band0.as<uint32_t>();
        cigar_prepare(c->cig, m, s);
      }
      uint64_t n_full = 0;
      sw_scores(cand, m, in, sp, lmax_chunk, band0, c->sww, &n_full, c->tune, s, long_chunk);
      if (c->tune.debug) fprintf(stderr, "[kslam] SW: %llu candidates, %llu needed the full-matrix kernel\n", (unsigned long long)m, (unsigned long long)n_full);
      HIPCHK(hipEventRecord(c->ev[6], s));
      if (lazy) {
        deferred.push_back(Deferred{c->n_res, m, lmax_chunk, long_chunk});
      } else {
        // ---- a-13: cigar ----
        uint32_t tb_err = 0;
        cigar_traceback(cand, m, in, sp, lmax_chunk, band0, c->cig, &ncig, &tb_err, tune_chunk, s);
        tb_err_total += tb_err;
        ensure_keep(c->res_cig, (c->n_cig + ncig + 1) * sizeof(uint32_t), c->n_cig * sizeof(uint32_t), s);
        cigar_finalize(cand, m, in, lmax_chunk, c->cig, band0, c->res_cig.as<uint32_t>(), c->n_cig, c->cells.as<uint64_t>(), s);
      }
    } else {
      HIPCHK(hipEventRecord(c->ev[6], s));
    }
    HIPCHK(hipEventRecord(c->ev[7], s));
    HIPCHK(stream_wait(s));
    tm.ms_extract += ev_ms(c->ev[0], c->ev[1]);
    tm.ms_sort += ev_ms(c->ev[1], c->ev[4]);
    if (nk) for (size_t q = 0; q < kpasses.size(); q++) tm.ms_sort_scatter += ev_ms(c->evs0[q], c->evs1[q]);
    tm.ms_join += ev_ms(c->ev[4], c->ev[5]);
    tm.ms_sw += ev_ms(c->ev[5], c->ev[6]);
    tm.ms_cigar += ev_ms(c->ev[6], c->ev[7]);
    tm.ms_total += ev_ms(c->ev[0], c->ev[7]);
    c->n_res += m;
    c->n_cig += ncig;
    r0 = r1;
  }
  if (lazy && c->n_res) {
    HIPCHK(hipEventRecord(c->ev[6], s));
    const uint64_t nr = c->n_res;
    if (nr < (1ull << 30)) {   // (else: no device pairing possible; every CIGAR, the caller pairs on the host)
      c->fin_copy.ensure((nr + 1) * sizeof(kslam_overlap));
      final_coords_copy(c->res_ov.as<kslam_overlap>(), nr, in, c->fin_copy.as<kslam_overlap>(), s);
      pair_and_screen(c->fin_copy.as<kslam_overlap>(), nr, c->r_len.as<uint32_t>(), c->n_reads, hook->paired ? 1 : 0, hook->thr,
                      hook->fraction, (hook->stages & 1u) != 0, (hook->stages & 2u) != 0, c->pw, c->sortws, &c->pres, s);
      if (hook->stages & 4u) pseudo_and_rescreen(c->pw, &c->pres, hook->fraction, c->sortws, s);
      const uint32_t *list = nullptr;
      uint64_t n_list = 0;
      referenced_rows(c->pw, &c->pres, nr, &list, &n_list, s);      // leaves the per-row flags in c->pw.flags
      drop_unreferenced_cigars(c->res_ov.as<kslam_overlap>(), c->band0_all.as<uint32_t>(), c->pw.flags.as<uint32_t>(), nr, s);
      hook->ran = true;
    }
    for (const Deferred &d : deferred) {
      kslam_overlap *cand = c->res_ov.as<kslam_overlap>() + d.first;
      uint32_t *band0 = c->band0_all.as<uint32_t>() + d.first;
      uint64_t ncig = 0;
      uint32_t tb_err = 0;
      cigar_prepare(c->cig, d.m, s);
      Tuning tune_d = c->tune;
      if (d.long_chunk) { tune_d.cigar_sys_mask = 0; tune_d.cigar_reg = false; tune_d.cigar_dirs_lds = false; }
      cigar_traceback(cand, d.m, in, sp, d.lmax, band0, c->cig, &ncig, &tb_err, tune_d, s);
      tb_err_total += tb_err;
      ensure_keep(c->res_cig, (c->n_cig + ncig + 1) * sizeof(uint32_t), c->n_cig * sizeof(uint32_t), s);
      cigar_finalize(cand, d.m, in, d.lmax, c->cig, band0, c->res_cig.as<uint32_t>(), c->n_cig, c->cells.as<uint64_t>(), s);
      c->n_cig += ncig;
    }
    HIPCHK(hipEventRecord(c->ev[7], s));
    HIPCHK(stream_wait(s));
    tm.ms_cigar += ev_ms(c->ev[6], c->ev[7]);
    tm.ms_total += ev_ms(c->ev[6], c->ev[7]);
  }
  if (hook && hook->ran) c->have_pairs = c->pairs_of_result = true;   // c->pres: pairs of THIS result (row numbers and coordinates are the final ones)
  tm.n_overlaps_raw = n_raw_total;
  tm.n_overlaps = c->n_res;
  read_back(&tm.sw_cells, c->cells.p, sizeof(uint64_t), s);
  c->tm = tm;
  if (n_raw_out) *n_raw_out = n_raw_total;
  if (tb_err_total)
    throw StatusError{KSLAM_ERR_INTERNAL, std::to_string(tb_err_total) +
                                              " candidates hit the reference's 'Trace back error' path"};
}

// The scoring a context takes is what `SLAM --match-score / --mismatch-penalty / --gap-open / --gap-extend` takes
// (src/main.cpp:44-55) as far as the reference's own types hold it: the Aligner stores the four as uint8_t, the score
// matrix is int8_t (src/ssw_cpp.cpp:25-49).  Inside the ENVELOPE the fast kernels apply; outside it
// every candidate goes through the literal striped kernel (sw.hip: k_sw_striped) and the literal banded_sw.

}  // namespace kslam_api
