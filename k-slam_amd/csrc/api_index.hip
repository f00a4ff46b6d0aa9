// api_index.hip -- the C ABI, part 2: kslam_set_index (genome k-mer extraction, the ONE-TIME sort of the genome k-mer list,
// key / {meta, offset} columns, bucket table, membership filter: build_index) and the stage-level entry points the parity tests
// use (kslam_extract_kmers / _sort_kmers / _selftest_sort / _find_overlaps).  Replaces GenbankIndex::getKMers + sortKMers of
// every batch (reference src/GenbankTools.h:211-219, src/KMer.h:388-398, src/SLAM.h:64-65) by one build per index.
#include "context.h"

namespace kslam_api {

__global__ void k_split_soa(const uint4 *__restrict__ recs, uint32_t n, uint64_t *__restrict__ key, uint2 *__restrict__ mo) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint4 r = recs[i];
  key[i] = ((uint64_t)r.y << 32) | r.x;
  mo[i] = make_uint2(r.z, r.w);
}

__global__ void k_fill_random(uint4 *recs, uint32_t n, uint64_t seed) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t z = seed + (uint64_t)i * 0x9E3779B97F4A7C15ull;   // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  recs[i] = make_uint4((uint32_t)z, (uint32_t)(z >> 32), i, ~i);
}
__global__ void k_count_inversions(const uint4 *recs, uint32_t n, unsigned long long *out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i + 1 >= n) return;
  const uint4 a = recs[i], b = recs[i + 1];
  const uint64_t ka = ((uint64_t)a.y << 32) | a.x, kb = ((uint64_t)b.y << 32) | b.x;
  // stable LSD: equal keys keep their input order (z = original index)
  if (ka > kb || (ka == kb && a.z > b.z)) atomicAdd(out, 1ull);
}

__global__ void k_to_temp(const kslam_overlap *__restrict__ in, uint64_t n, kslam_overlap_temp *__restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  kslam_overlap o = in[i];
  kslam_overlap_temp t;
  t.read = o.read; t.entry = o.entry; t.rel = o.rel; t.revcomp = o.revcomp;
  t.pad[0] = t.pad[1] = t.pad[2] = 0;
  out[i] = t;
}

// passes over the 64-bit k-mer (words x, y of the record), least significant first
void kmer_passes(std::vector<SortPass> &v) {
  for (uint32_t w = 0; w < 2; w++)
    for (uint32_t b = 0; b < 4; b++) v.push_back(SortPass{w, 8 * b, 0});
}
// sortKMers key (KMer.h:392-396): kmer asc, meta desc -> LSD: ~meta bytes, then kmer bytes
void full_key_passes(std::vector<SortPass> &v) {
  for (uint32_t b = 0; b < 4; b++) v.push_back(SortPass{2, 8 * b, 0xFFFFFFFFu});
  kmer_passes(v);
}

struct Planned {
  uint64_t n_kmers = 0, n_segs = 0;
};
Planned plan_host(const uint64_t *off, uint64_t n, uint32_t gap) {
  Planned p;
  for (uint64_t i = 0; i < n; i++) {
    uint64_t len = off[i + 1] - off[i];
    uint64_t k = len >= KSLAM_K ? (len - KSLAM_K) / gap + 1 : 0;
    p.n_kmers += k;
    p.n_segs += (k + SEG_KMERS - 1) / SEG_KMERS;
  }
  return p;
}

// extraction of n sequences d_off[0..n] into d_out (AoS records)
void run_extract(kslam_ctx *c, const uint8_t *d_bases, const uint64_t *d_off, uint64_t n, uint32_t gap, int is_gb,
                 uint64_t n_segs, uint4 *d_out, uint8_t *d_digits, const SortPass *first_pass) {
  hipStream_t s = c->stream;
  c->nk.ensure(n * sizeof(uint32_t) + 4);
  c->nseg.ensure(n * sizeof(uint32_t) + 4);
  c->rec_start.ensure(n * sizeof(uint64_t) + 8);
  c->seg_start.ensure(n * sizeof(uint64_t) + 8);
  c->scan_tmp.ensure(scan_tmp_bytes(n));
  c->totals.ensure(8 * sizeof(uint64_t));
  c->segs.ensure((n_segs + 1) * sizeof(SegEntry));
  extract_plan(d_off, n, gap, c->nk.as<uint32_t>(), c->nseg.as<uint32_t>(), c->rec_start.as<uint64_t>(),
               c->seg_start.as<uint64_t>(), c->totals.as<uint64_t>(), c->scan_tmp.p, s);
  extract_fill_segments(c->nk.as<uint32_t>(), c->rec_start.as<uint64_t>(), c->seg_start.as<uint64_t>(), n, gap,
                        c->segs.as<SegEntry>(), s, n_segs);
  extract_kmers_launch(d_bases, d_off, c->segs.as<SegEntry>(), n_segs, gap, is_gb, 0, d_out, s, d_digits, first_pass);
}

void build_index(kslam_ctx *c) {
  hipStream_t s = c->stream;
  const uint64_t n = c->n_entries;
  c->max_entry_len = 0;
  for (uint64_t i = 0; i < n; i++) c->max_entry_len = std::max(c->max_entry_len, c->h_goff[i + 1] - c->h_goff[i]);
  if (n >= (1ull << 30)) throw StatusError{KSLAM_ERR_UNSUPPORTED, "more than 2^30 entries (KMer.h:65 id field)"};
  if (c->max_entry_len >= (1ull << 32)) throw StatusError{KSLAM_ERR_UNSUPPORTED, "entry longer than 2^32 bases"};
  c->g_off.ensure((n + 1) * sizeof(uint64_t));
  HIPCHK(hipMemcpyAsync(c->g_off.p, c->h_goff.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
  struct Events {       // destroyed on every way out of this function
    hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
    ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
  } evs;
  for (hipEvent_t &x : evs.e) HIPCHK(hipEventCreate(&x));
  hipEvent_t ev_begin = evs.e[0], e1 = evs.e[1], e2 = evs.e[2], e3 = evs.e[3];
  HIPCHK(hipEventRecord(ev_begin, s));
  c->g_codes.ensure(c->h_goff[n] + 64);
  encode_bases(c->g_bases.as<uint8_t>(), c->g_codes.as<uint8_t>(), c->h_goff[n] + 48, s);
  Planned pl = plan_host(c->h_goff.data(), n, KSLAM_K / 2);  // gap k/2, SLAM.h:64
  if (pl.n_kmers >= (1ull << 32)) throw StatusError{KSLAM_ERR_UNSUPPORTED, "more than 2^32 genome k-mers"};
  c->n_gk = pl.n_kmers;
  const uint64_t m = pl.n_kmers;
  c->recs_a.ensure((m + 1) * sizeof(uint4));
  c->recs_b.ensure((m + 1) * sizeof(uint4));
  // sortKMers' order (src/KMer.h:388-398): k-mer ascending, then the meta word DESCENDING.  LSD passes over the meta word
  // first -- but only over what can differ in a list of genome records: the id (n entries: bits 0 .. b - 1), isFromGB = 1
  // everywhere, revComp in bit 30 -- then the 8 bytes of the k-mer.  The id's bytes below its top one take a pass each; its
  // top bits (at most 7 of them) share ONE pass with the revComp bit above them (SortPass::hi_bits): ids of 1 250 entries
  // are 11 bits, so the meta word takes two passes -- id bits 0-7, then {revComp, id bits 8-14} -- and the sort ten.
  // (Round 5: a pass per byte that can differ, three for this database.)
  std::vector<SortPass> passes;
  {
    const uint32_t id_bits = (uint32_t)bits_for(n ? n - 1 : 0);     // <= 30
    uint32_t at = 0;
    while (id_bits - at > 7) {                                        // whole bytes of the id while more than 7 bits remain
      passes.push_back(SortPass{2, at, 0xFFFFFFFFu});
      at += 8;
    }
    SortPass top{2, at, 0xFFFFFFFFu};                                 // the rest of the id below the revComp bit
    top.hi_shift = 30;
    top.hi_bits = 1;
    passes.push_back(top);
  }
  kmer_passes(passes);
  // the extraction writes the first pass's digit of every record next to it: the sort's first histogram reads 1 byte per
  // record instead of 16
  const bool first_digits = c->tune.sort_digit_bytes && passes.size() > 1;
  if (first_digits) c->sortws.digits.ensure(m + 64);
  run_extract(c, c->g_bases.as<uint8_t>(), c->g_off.as<uint64_t>(), n, KSLAM_K / 2, 1, pl.n_segs,
              c->recs_a.as<uint4>(), first_digits ? c->sortws.digits.as<uint8_t>() : nullptr, first_digits ? &passes[0] : nullptr);
  c->sortws.use_digit_bytes = c->tune.sort_digit_bytes;
  c->sortws.first_digits_ready = first_digits;
  c->sortws.meta_digits_in_runs = n && m / n >= 64;     // (a database of many tiny entries has as many ids as records in a tile)
  HIPCHK(hipEventRecord(e1, s));
  void *sorted = radix_sort(c->recs_a.p, c->recs_b.p, m, 4, passes.data(), (int)passes.size(), c->sortws, s,
                            nullptr, nullptr, nullptr, /*setup=*/true);
  c->sortws.first_digits_ready = false;
  c->sortws.meta_digits_in_runs = false;
  HIPCHK(hipEventRecord(e2, s));
  c->gk_key.ensure((m + 1) * sizeof(uint64_t));
  c->gk_meta.ensure((m + 1) * sizeof(uint2));   // {meta, offset} pairs
  uint32_t bits = 8, max_bits = (uint32_t)c->tune.bucket_bits_max;   // 27: ~2.3 genome k-mers per bucket for a 5 Gb database (537 MB table)
  while (bits < max_bits && (m >> (bits + 2)) != 0) bits++;   // 2 to 4 keys per bucket (measured: 3.06 ms at 27 bits, 3.24 at 26, 3.13 at 28)
  if (c->tune.bucket_bits_exact) bits = (uint32_t)c->tune.bucket_bits_exact;   // tuning
  c->bucket_bits = bits;
  c->g_bucket.ensure(((1ull << bits) + 2) * sizeof(uint32_t));
  // membership filter for the read extraction: ~14 bits per genome k-mer (9.3 keys per 128-bit piece),
  // 2^32 bits = 512 MiB for the 312 M k-mers of a 5 Gb database.  KSLAM_FILTER_BITS: log2 of the size
  // in bits, 0 = extract, sort and look up every read k-mer as the reference does.
  uint32_t fb = 20;
  while (fb < 35 && ((uint64_t)1 << fb) < m * 12) fb++;
  if (c->tune.filter_bits >= 0) fb = (uint32_t)c->tune.filter_bits;
  c->filter_bits = fb;
  if (fb) c->g_filter.ensure(filter_bytes(fb));
  // the probe words of the filter's build go through the record buffers of the sort that has just finished: the one that does not
  // hold the sorted list takes them first
  void *other = sorted == c->recs_a.p ? c->recs_b.p : c->recs_a.p;
  bool fused = false;
  if (fb && c->tune.filter_build_sorted)
    fused = split_columns_and_tables(sorted, (uint32_t)m, c->gk_key.as<uint64_t>(), c->gk_meta.p, bits, c->g_bucket.as<uint32_t>(), fb, other, c->sortws, s);
  if (!fused) {
    if (m) hipLaunchKernelGGL(k_split_soa, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, (const uint4 *)sorted,
                              (uint32_t)m, c->gk_key.as<uint64_t>(), c->gk_meta.as<uint2>());
    build_bucket_table(c->gk_key.as<uint64_t>(), (uint32_t)m, bits, c->g_bucket.as<uint32_t>(), s);
  }
  if (fb) {
    if (c->tune.filter_build_sorted) {
      c->pos.ensure((filter_bytes(fb) / 32768 + 2) * sizeof(uint32_t));
      filter_build_sorted(c->gk_key.as<uint64_t>(), (uint32_t)m, fb, c->g_filter.p, other, sorted, c->pos.as<uint32_t>(), c->sortws, s, fused);
    } else {
      filter_build(c->gk_key.as<uint64_t>(), (uint32_t)m, fb, c->g_filter.p, s);
    }
  }
  HIPCHK(hipEventRecord(e3, s));
  HIPCHK(stream_wait(s));
  {
    kslam_index_stats &st = c->index_stats;
    memset(&st, 0, sizeof st);
    st.n_genome_kmers = m;
    st.sort_passes = (uint32_t)passes.size();
    st.n_entries = (uint32_t)n;
    (void)hipEventElapsedTime(&st.ms_encode_extract, ev_begin, e1);
    (void)hipEventElapsedTime(&st.ms_sort, e1, e2);
    (void)hipEventElapsedTime(&st.ms_tables, e2, e3);
    (void)hipEventElapsedTime(&st.ms_total, ev_begin, e3);
    // the one-time sorts' digit bytes (one per genome k-mer: 312 MB for the 5 Gb database) are not kept for the context's life:
    // a batch's sort allocates what its own record count needs
    c->sortws.digits.release();
  }
  c->kept_last = 0;
  c->have_index = true;
  for (auto *l : c->lanes) share_index(l->c, c);   // (no batch may be in flight across kslam_set_index)
}

}  // namespace kslam_api

extern "C" {

kslam_status kslam_index_build_stats(const kslam_ctx *c, kslam_index_stats *out) {
  if (!c || !out) return KSLAM_ERR_ARG;
  if (!c->have_index) return KSLAM_ERR_STATE;
  *out = c->index_stats;
  return KSLAM_OK;
}

kslam_status kslam_set_index(kslam_ctx *c, uint64_t n_entries, const char *const *bases, const uint64_t *lens) {
  return guarded(c, [&] {
    if (n_entries && (!bases || !lens)) throw StatusError{KSLAM_ERR_ARG, "null bases/lens"};
    c->have_index = false;
    c->n_entries = n_entries;
    c->h_goff.assign(n_entries + 1, 0);
    for (uint64_t i = 0; i < n_entries; i++) c->h_goff[i + 1] = c->h_goff[i] + lens[i];
    const uint64_t total = c->h_goff[n_entries];
    c->g_bases.ensure(total + 64);
    for (uint64_t i = 0; i < n_entries; i++)
      if (lens[i])
        HIPCHK(hipMemcpyAsync(c->g_bases.as<uint8_t>() + c->h_goff[i], bases[i], lens[i], hipMemcpyHostToDevice,
                              c->stream));
    HIPCHK(hipMemsetAsync(c->g_bases.as<uint8_t>() + total, 0, 64, c->stream));
    HIPCHK(stream_wait(c->stream));
    build_index(c);
  });
}

kslam_status kslam_set_index_device(kslam_ctx *c, uint64_t n_entries, const void *d_bases,
                                    const uint64_t *h_offsets) {
  return guarded(c, [&] {
    if (n_entries && (!d_bases || !h_offsets)) throw StatusError{KSLAM_ERR_ARG, "null bases/offsets"};
    c->have_index = false;
    c->n_entries = n_entries;
    c->h_goff.assign(n_entries + 1, 0);
    const uint64_t o0 = n_entries ? h_offsets[0] : 0;
    for (uint64_t i = 0; i <= n_entries && n_entries; i++) c->h_goff[i] = h_offsets[i] - o0;
    const uint64_t total = c->h_goff[n_entries];
    c->g_bases.ensure(total + 64);
    if (total)
      HIPCHK(hipMemcpyAsync(c->g_bases.p, (const uint8_t *)d_bases + o0, total, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->g_bases.as<uint8_t>() + total, 0, 64, c->stream));
    build_index(c);
  });
}

kslam_status kslam_extract_kmers(kslam_ctx *c, uint64_t n, const char *const *bases, const uint64_t *lens,
                                 int is_from_genbank, uint32_t gap, kslam_kmer *out, uint64_t cap, uint64_t *n_out) {
  return guarded(c, [&] {
    if (!n_out) throw StatusError{KSLAM_ERR_ARG, "null n_out"};
    if (gap == 0 || gap > MAX_GAP) throw StatusError{KSLAM_ERR_UNSUPPORTED, "gap must be in 1..64"};
    if (n >= (1ull << 30)) throw StatusError{KSLAM_ERR_UNSUPPORTED, "more than 2^30 sequences"};
    std::vector<uint64_t> off(n + 1, 0);
    for (uint64_t i = 0; i < n; i++) off[i + 1] = off[i] + lens[i];
    Planned pl = plan_host(off.data(), n, gap);
    *n_out = pl.n_kmers;
    if (pl.n_kmers > cap || pl.n_kmers == 0) return;
    hipStream_t s = c->stream;
    DevBuf db, doff, drec;
    db.ensure(off[n] + 64);
    doff.ensure((n + 1) * sizeof(uint64_t));
    drec.ensure(pl.n_kmers * sizeof(uint4));
    for (uint64_t i = 0; i < n; i++)
      if (lens[i]) HIPCHK(hipMemcpyAsync(db.as<uint8_t>() + off[i], bases[i], lens[i], hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(db.as<uint8_t>() + off[n], 0, 64, s));
    HIPCHK(hipMemcpyAsync(doff.p, off.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    run_extract(c, db.as<uint8_t>(), doff.as<uint64_t>(), n, gap, is_from_genbank, pl.n_segs, drec.as<uint4>());
    HIPCHK(hipMemcpyAsync(out, drec.p, pl.n_kmers * sizeof(uint4), hipMemcpyDeviceToHost, s));
    HIPCHK(stream_wait(s));
    db.release(); doff.release(); drec.release();
  });
}

kslam_status kslam_sort_kmers(kslam_ctx *c, kslam_kmer *recs, uint64_t n) {
  return guarded(c, [&] {
    if (n == 0) return;
    if (!recs) throw StatusError{KSLAM_ERR_ARG, "null recs"};
    hipStream_t s = c->stream;
    DevBuf a, b;
    a.ensure(n * sizeof(uint4));
    b.ensure(n * sizeof(uint4));
    HIPCHK(hipMemcpyAsync(a.p, recs, n * sizeof(uint4), hipMemcpyHostToDevice, s));
    std::vector<SortPass> passes;
    full_key_passes(passes);
    void *sorted = radix_sort(a.p, b.p, n, 4, passes.data(), (int)passes.size(), c->sortws, s, nullptr, nullptr,
                              nullptr);
    HIPCHK(hipMemcpyAsync(recs, sorted, n * sizeof(uint4), hipMemcpyDeviceToHost, s));
    HIPCHK(stream_wait(s));
    a.release(); b.release();
  });
}

kslam_status kslam_selftest_sort(kslam_ctx *c, uint64_t n, uint32_t iters, float *ms_per_sort,
                                 float *ms_per_scatter_launch, uint64_t *n_inversions) {
  return guarded(c, [&] {
    if (n == 0 || n >= (1ull << 32) || iters == 0) throw StatusError{KSLAM_ERR_ARG, "bad n / iters"};
    hipStream_t s = c->stream;
    c->recs_a.ensure((n + 1) * sizeof(uint4));
    c->recs_b.ensure((n + 1) * sizeof(uint4));
    c->cells.ensure(sizeof(uint64_t));
    std::vector<SortPass> passes;
    kmer_passes(passes);
    float tot = 0, tot_sc = 0;
    uint32_t launches = 0;
    const void *sorted = nullptr;
    for (uint32_t it = 0; it < iters; it++) {
      hipLaunchKernelGGL(k_fill_random, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->recs_a.as<uint4>(),
                         (uint32_t)n, 0x1234567ull + it);
      HIPCHK(hipEventRecord(c->ev[0], s));
      c->sortws.ev_sc0 = c->evs0; c->sortws.ev_sc1 = c->evs1;
      sorted = radix_sort(c->recs_a.p, c->recs_b.p, n, 4, passes.data(), (int)passes.size(), c->sortws, s, c->ev[2],
                          c->ev[3], &launches);
      c->sortws.ev_sc0 = nullptr; c->sortws.ev_sc1 = nullptr;
      HIPCHK(hipEventRecord(c->ev[1], s));
      HIPCHK(stream_wait(s));
      tot += ev_ms(c->ev[0], c->ev[1]);
      for (size_t q = 0; q < passes.size(); q++) tot_sc += ev_ms(c->evs0[q], c->evs1[q]);
    }
    HIPCHK(hipMemsetAsync(c->cells.p, 0, sizeof(uint64_t), s));
    hipLaunchKernelGGL(k_count_inversions, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint4 *)sorted,
                       (uint32_t)n, c->cells.as<unsigned long long>());
    uint64_t inv = 0;
    read_back(&inv, c->cells.p, sizeof inv, s);
    if (ms_per_sort) *ms_per_sort = tot / iters;
    if (ms_per_scatter_launch) *ms_per_scatter_launch = launches ? tot_sc / launches : 0.f;
    if (n_inversions) *n_inversions = inv;
  });
}

kslam_status kslam_find_overlaps(kslam_ctx *c, kslam_overlap_temp **out, uint64_t *n_out, uint64_t *n_raw) {
  if (!out || !n_out) return KSLAM_ERR_ARG;
  *out = nullptr; *n_out = 0;
  return guarded(c, [&] {
    uint64_t raw = 0;
    align_resident(c, true, &raw);
    if (n_raw) *n_raw = raw;
    const uint64_t m = c->n_res;
    kslam_overlap_temp *h = (kslam_overlap_temp *)malloc((m + 1) * sizeof(kslam_overlap_temp));
    if (!h) throw StatusError{KSLAM_ERR_OOM, "host allocation failed"};
    if (m) {
      c->res_tmp.ensure(m * sizeof(kslam_overlap_temp));
      hipLaunchKernelGGL(k_to_temp, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream,
                         c->res_ov.as<kslam_overlap>(), m, c->res_tmp.as<kslam_overlap_temp>());
      HIPCHK(hipMemcpyAsync(h, c->res_tmp.p, m * sizeof(kslam_overlap_temp), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(stream_wait(c->stream));
    }
    *out = h;
    *n_out = m;
  });
}

}  // extern "C"
