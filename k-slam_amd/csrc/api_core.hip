// api_core.hip -- the C ABI (include/kslam.h), part 1: context life cycle, tuning switches, page-locked pools, loading reads,
// fetching results and the operator entry point.  The hot path is alignToDatabase (reference src/SLAM.h:59-79) as
//   extract read k-mers -> radix sort -> join against the resident sorted genome
//   k-mer list (a probe; a merge behind KSLAM_JOIN=merge) -> overlap sort + dedupe -> SW scores -> banded CIGAR
// (api_align.hip).
// All device work runs on the context's own HIP stream; phases are bracketed
// with HIP events.  No CPU fallback exists: without a HIP device every entry
// point fails with KSLAM_ERR_NO_DEVICE.
#include <sys/mman.h>

#include "context.h"

namespace kslam {
Tuning read_tuning() {
  Tuning t;
  auto flag = [](const char *name) { return getenv(name) != nullptr; };
  auto num = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
  auto starts = [](const char *name, char ch) { const char *e = getenv(name); return e && e[0] == ch; };
  t.debug = flag("KSLAM_DEBUG");
  t.sw_full = starts("KSLAM_SW_FULL", '1');
  t.sw_no48 = flag("KSLAM_SW_NO48");
  t.sw_no96 = flag("KSLAM_SW_NO96");
  t.sw_unknown_nd = num("KSLAM_SW_UNKNOWN_ND", 0);
  t.cigar_sys_mask = num("KSLAM_CIGAR_SYS", 0xF8);
  t.cigar_reg = !starts("KSLAM_CIGAR_REG", '0');
  t.plan_blocks_per_cu = std::min(256, std::max(1, num("KSLAM_PLAN_BLOCKS", 64)));
  t.cigar_dirs_lds = starts("KSLAM_CIGAR_DIRS", 'l');
  t.cigar_tb_inline = starts("KSLAM_CIGAR_TB", 'i');
  t.bucket_bits_max = std::min(28, std::max(8, num("KSLAM_BUCKET_BITS", 27)));
  t.bucket_bits_exact = flag("KSLAM_BUCKET_BITS_EXACT") ? std::min(28, std::max(8, num("KSLAM_BUCKET_BITS_EXACT", 0))) : 0;
  if (flag("KSLAM_FILTER_BITS")) {
    const int v = num("KSLAM_FILTER_BITS", 0);
    t.filter_bits = v <= 0 ? 0 : std::min(36, std::max(20, v));
  }
  if (flag("KSLAM_SORT_BYTES")) t.sort_bytes = std::max(0, num("KSLAM_SORT_BYTES", 0));
  t.sort_digit_bytes = !starts("KSLAM_SORT_DIGIT_BYTES", '0');
  t.lanes = std::min(8, std::max(1, num("KSLAM_LANES", 2)));
  t.eager_cigar = flag("KSLAM_EAGER_CIGAR");
  t.lane_waits_yield = starts("KSLAM_LANE_WAITS", 'y');
  t.pageable_columns = flag("KSLAM_PAGEABLE_COLUMNS");
  t.pseudo_cap = std::max(0, num("KSLAM_PSEUDO_CAP", 0));
  t.join_group_order = starts("KSLAM_JOIN_GROUP_ORDER", '0') ? 0 : 1;
  t.join_merge = starts("KSLAM_JOIN", 'm') ? 1 : 0;
  t.sw_sweep = !starts("KSLAM_SW_SWEEP", '0');
  t.sweep_room = !starts("KSLAM_SWEEP_ROOM", '0');
  t.filter_build_sorted = !starts("KSLAM_FILTER_BUILD", 'a');      // =atomics: the scattered read-modify-write build (A/B)
  t.details_in_token = !starts("KSLAM_DETAILS_IN_TOKEN", '0');
#ifdef KSLAM_ABLATE
  t.sw_ablate = (uint32_t)num("KSLAM_SW_ABLATE", 0);
  t.cigar_variant = (uint32_t)num("KSLAM_CIGAR_VARIANT", 0);
  t.filter_ablate = (uint32_t)num("KSLAM_FILTER_ABLATE", 0);
#endif
  return t;
}
}  // namespace kslam

namespace kslam_api {

uint32_t bits_for(uint64_t max_value) {
  uint32_t b = 1;
  while (b < 64 && (max_value >> b) != 0) b++;
  return b;
}

// grow a device buffer while keeping its first `used` bytes
void ensure_keep(DevBuf &b, size_t bytes, size_t used, hipStream_t s) {
  if (bytes <= b.cap) return;
  DevBuf nb;
  nb.ensure(bytes + bytes / 2);
  if (used && b.p) {
    HIPCHK(hipMemcpyAsync(nb.p, b.p, used, hipMemcpyDeviceToDevice, s));
    HIPCHK(stream_wait(s));
  }
  b = std::move(nb);   // frees the old block, takes the new one
}

// Page-locked host memory for the result / staging buffers: a private anonymous mapping advised for
// transparent huge pages, then registered with the runtime.  (hipHostMalloc gives 4 KiB pages; the
// host tail reads the overlap records in it at random, 3 M of them per batch, a TLB miss each.)
// blocks that came from hipHostMalloc (KSLAM_PINNED_PLAIN, or registering a mapping failed)
static std::vector<void *> &plain_blocks() { static std::vector<void *> v; return v; }
static std::mutex &plain_mutex() { static std::mutex m; return m; }
static void *pinned_plain_alloc(size_t bytes) {
  void *q = nullptr;
  if (hipHostMalloc(&q, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lk(plain_mutex());
  plain_blocks().push_back(q);
  return q;
}
void *pinned_alloc(size_t bytes);
void pinned_free(void *p, size_t bytes);
// the allocator the host-side FASTQ parser uses for its big column blocks while a GPU context exists
// (workers.hpp: big_alloc_hook): installed by the first kslam_create, removed by the last kslam_destroy
const kslam_host::BigAlloc pinned_hook{pinned_alloc, pinned_free};
std::mutex &hook_mutex() { static std::mutex m; return m; }
int &hook_users() { static int n = 0; return n; }

void *pinned_alloc(size_t bytes) {
  static const bool plain = getenv("KSLAM_PINNED_PLAIN") != nullptr;
  if (plain) return pinned_plain_alloc(bytes);
  const size_t HP = 2u << 20, len = (bytes + HP - 1) / HP * HP;
  void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) return pinned_plain_alloc(bytes);
#ifdef MADV_HUGEPAGE
  (void)madvise(p, len, MADV_HUGEPAGE);
#endif
  if (hipHostRegister(p, len, hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();
    munmap(p, len);
    return pinned_plain_alloc(bytes);
  }
  return p;
}
void pinned_free(void *p, size_t bytes) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(plain_mutex());
    auto &v = plain_blocks();
    auto it = std::find(v.begin(), v.end(), p);
    if (it != v.end()) {
      v.erase(it);
      (void)hipHostFree(p);
      return;
    }
  }
  const size_t HP = 2u << 20, len = (bytes + HP - 1) / HP * HP;
  (void)hipHostUnregister(p);
  munmap(p, len);
}

// pinned host buffers from a small per-context pool (pinning is expensive; reuse across batches)
void *pinned_get(kslam_ctx *c, size_t bytes) {
  std::lock_guard<std::mutex> lk(c->pin_mu);
  // best fit: a small request must not take the buffer a large one of the same batch needs (first fit
  // did, and the large request then re-pinned ~60 MB -- ~10 ms -- on every batch)
  kslam_ctx::Pinned *best = nullptr;
  for (auto &b : c->pinned)
    if (!b.in_use && b.cap >= bytes && (!best || b.cap < best->cap)) best = &b;
  if (best) { best->in_use = true; return best->p; }
  if (c->pinned.size() >= 64)
    for (auto &b : c->pinned)   // many buffers already: replace a free one that is too small
      if (!b.in_use) {
        pinned_free(b.p, b.cap);
        b.p = nullptr; b.cap = 0;
        size_t want = bytes + bytes / 4 + 4096;
        if (!(b.p = pinned_alloc(want))) throw StatusError{KSLAM_ERR_OOM, "page-locked host allocation failed"};
        b.cap = want; b.in_use = true;
        return b.p;
      }
  kslam_ctx::Pinned nb{nullptr, 0, true};
  size_t want = bytes + bytes / 4 + 4096;
  if (!(nb.p = pinned_alloc(want))) throw StatusError{KSLAM_ERR_OOM, "page-locked host allocation failed"};
  nb.cap = want;
  c->pinned.push_back(nb);
  return nb.p;
}
bool pinned_put(kslam_ctx *c, void *p) {
  std::lock_guard<std::mutex> lk(c->pin_mu);
  for (auto &b : c->pinned)
    if (b.p == p) { b.in_use = false; return true; }
  return false;
}

bool scoring_in_envelope(const kslam_params &p) {
  return p.match >= 1 && p.match <= 31 && p.mismatch <= 32 && p.gap_extend >= 1 && p.gap_extend < p.gap_open &&
         p.mismatch <= p.gap_open + p.gap_extend;
}
void validate_params(const kslam_params &p) {
  if (p.match > 127 || p.mismatch > 127) throw StatusError{KSLAM_ERR_UNSUPPORTED, "match score and mismatch penalty must fit the reference's int8_t score matrix (<= 127)"};
  if (p.gap_open > 255 || p.gap_extend > 255) throw StatusError{KSLAM_ERR_UNSUPPORTED, "gap penalties must fit uint8_t (ssw_cpp.h Aligner)"};
  if (p.match == 0) throw StatusError{KSLAM_ERR_UNSUPPORTED, "match score 0: no k-mer seed could ever score"};
  if (p.score_threshold > 65535) throw StatusError{KSLAM_ERR_UNSUPPORTED, "score_threshold must fit uint16_t (ssw_cpp.h Filter)"};
}

void share_index(kslam_ctx *dst, const kslam_ctx *src) {
  dst->borrowed_index = true;
  dst->have_index = src->have_index;
  dst->index_stats = src->index_stats;
  dst->n_entries = src->n_entries; dst->max_entry_len = src->max_entry_len; dst->h_goff = src->h_goff;
  // views, not owners (DevBuf::borrow frees what dst owned before: an index of its own, if it had one)
  dst->g_bases.borrow(src->g_bases); dst->g_off.borrow(src->g_off); dst->g_codes.borrow(src->g_codes);
  dst->n_gk = src->n_gk; dst->gk_key.borrow(src->gk_key); dst->gk_meta.borrow(src->gk_meta); dst->gk_off.borrow(src->gk_off);
  dst->g_bucket.borrow(src->g_bucket); dst->bucket_bits = src->bucket_bits;
  dst->g_filter.borrow(src->g_filter); dst->filter_bits = src->filter_bits;
  dst->kept_last = 0;
  dst->pairing = src->pairing;
}


// The batch's bases and quality columns cut out of the two FASTQ texts on the device (the host only
// indexed the records: kslam_fastq_index_pair).  Leaves the context as kslam_load_reads +

}  // namespace kslam_api

extern "C" {

uint32_t kslam_abi_version(void) { return KSLAM_ABI_VERSION; }

kslam_status kslam_create(const kslam_params *params, kslam_ctx **out) {
  if (!params || !out) return KSLAM_ERR_ARG;
  *out = nullptr;
  kslam_ctx *c = new (std::nothrow) kslam_ctx();
  if (!c) return KSLAM_ERR_OOM;
  c->prm = *params;
  c->device = params->device;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0 || c->device < 0 || c->device >= ndev) {
    // no device: hand back a context that only carries the message (so the caller can read it)
    c->err = "no usable HIP device (hipGetDeviceCount: " + std::string(hipGetErrorString(e)) + ", count " +
             std::to_string(ndev) + ", requested " + std::to_string(c->device) + "); this library has no CPU path";
    (void)hipGetLastError();
    *out = c;
    c->device = -1;
    return KSLAM_ERR_NO_DEVICE;
  }
  kslam_status st = guarded(c, [&] {
    validate_params(c->prm);
    c->tune = read_tuning();
    c->pw.pseudo_cap = (uint32_t)c->tune.pseudo_cap;
    // while a context exists the FASTQ parser's big column blocks are page-locked (DMA-able as they stand)
    if (!c->tune.pageable_columns) {
      std::lock_guard<std::mutex> lk(hook_mutex());
      if (hook_users()++ == 0) kslam_host::big_alloc_hook().store(&pinned_hook, std::memory_order_release);
      c->holds_hook = true;
    }
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto &ev : c->ev) HIPCHK(hipEventCreate(&ev));
    for (auto &ev : c->evs0) HIPCHK(hipEventCreate(&ev));
    for (auto &ev : c->evs1) HIPCHK(hipEventCreate(&ev));
  });
  *out = c;
  return st;
}

void kslam_destroy(kslam_ctx *c) {
  if (!c) return;
  if (!c->lanes.empty()) stop_lanes(c);   // workers first: they use this context's index
  if (c->holds_hook) {
    std::lock_guard<std::mutex> lk(hook_mutex());
    if (--hook_users() == 0) kslam_host::big_alloc_hook().store(nullptr, std::memory_order_release);
    c->holds_hook = false;
  }
  if (c->device >= 0) {
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    // (a lane's / sibling's view of its primary's index is not freed: DevBuf::borrowed)
    DevBuf *bufs[] = {&c->g_codes, &c->r_codes, &c->g_bases, &c->g_off, &c->gk_key, &c->gk_meta, &c->gk_off, &c->g_bucket, &c->g_filter, &c->r_bases,
                      &c->r_off, &c->r_len, &c->nk, &c->nseg, &c->rec_start, &c->seg_start, &c->segs, &c->scan_tmp,
                      &c->totals, &c->recs_a, &c->recs_b, &c->block_tot, &c->block_base, &c->ovk_a, &c->ovk_b,
                      &c->flags, &c->pos, &c->band0, &c->sortws.hist, &c->sortws.status, &c->sortws.tickets, &c->sortws.digits,
                      &c->cig.flags, &c->cig.pos, &c->cig.list, &c->cig.bmax, &c->cig.needbig,
                      &c->cig.scan_tmp, &c->cig.totals, &c->cig.cig_off, &c->cig.tmp, &c->cig.tmp_big,
                      &c->cig.big_pos, &c->cig.scratch, &c->sww.flags, &c->sww.pos, &c->sww.list, &c->sww.list2, &c->sww.scan_tmp, &c->sww.totals, &c->cells, &c->res_ov, &c->res_cig, &c->res_tmp, &c->r_qual, &c->d_tables, &c->res_det, &c->fq_text, &c->fq_bases_at, &c->fq_qual_at, &c->fqw.tile_count, &c->fqw.tile_base, &c->fqw.scan_tmp,
                      &c->fqw.totals, &c->fqw.ev[0], &c->fqw.ev[1], &c->fqw.bases_at, &c->fqw.quality_at, &c->fqw.blen, &c->fqw.id_at,
                      &c->fqw.id_len, &c->fqw.bases_off, &c->fqw.ids_off, &c->fqw.ids, &c->detw.lens, &c->detw.off,
                      &c->detw.slots, &c->detw.scan_tmp, &c->detw.totals, &c->detw.md_pool, &c->pw.recs, &c->pw.count, &c->pw.base,
                      &c->pw.inserts, &c->pw.flags, &c->pw.gpos, &c->pw.rpos, &c->pw.scan_tmp, &c->pw.totals, &c->pw.groups,
                      &c->pw.dense, &c->pw.sort_a, &c->pw.sort_b, &c->pw.idx, &c->pw.picked, &c->pw.row_list, &c->pw.row_start, &c->pw.gaps, &c->pr_ov, &c->pr_len, &c->mg_shards, &c->mg_lens, &c->mg_off, &c->mg_scan};
    for (DevBuf *b : bufs) b->release();
    for (auto &b : c->annot_bufs) b.release();
    DevBuf *sam_bufs[] = {&c->samw.plan, &c->samw.cnt_vals, &c->samw.cnt_segs, &c->samw.val_off, &c->samw.seg_off, &c->samw.scan_tmp,
                          &c->samw.totals, &c->samw.vals, &c->samw.seg_len, &c->samw.mapq, &c->samw.text_len, &c->samw.text_off,
                          &c->samw.text, &c->samw.tax_ids, &c->samw.pr_len, &c->samw.pr_off, &c->samw.pr_text, &c->ids_buf, &c->ids_off_buf};
    for (DevBuf *b : sam_bufs) b->release();
    {
      std::lock_guard<std::mutex> lk(c->pin_mu);
      for (auto &b : c->pinned) pinned_free(b.p, b.cap);
      c->pinned.clear();
    }
    for (auto &ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : c->evs0) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : c->evs1) if (ev) (void)hipEventDestroy(ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
  }
  delete c;
}

const char *kslam_last_error(const kslam_ctx *c) { return c ? c->err.c_str() : "null context"; }

kslam_status kslam_create_sibling(kslam_ctx *primary, kslam_ctx **out) {
  if (!primary || !out) return KSLAM_ERR_ARG;
  *out = nullptr;
  if (primary->device < 0) return KSLAM_ERR_NO_DEVICE;
  kslam_ctx *c = nullptr;
  const kslam_status st = kslam_create(&primary->prm, &c);
  if (st != KSLAM_OK) {
    if (c) primary->err = c->err;
    kslam_destroy(c);
    return st;
  }
  c->tune = primary->tune;
  c->pw.pseudo_cap = (uint32_t)c->tune.pseudo_cap;
  share_index(c, primary);
  *out = c;
  return KSLAM_OK;
}

kslam_status kslam_adopt_results_device(kslam_ctx *c, const void *d_overlaps, uint64_t n_overlaps, const void *d_cigar_pool,
                                        uint64_t n_cigar) {
  return guarded(c, [&] {
    if (!c->have_reads) throw StatusError{KSLAM_ERR_STATE, "no batch loaded: the records refer to the reads of a loaded batch"};
    if ((n_overlaps && !d_overlaps) || (n_cigar && !d_cigar_pool)) throw StatusError{KSLAM_ERR_ARG, "null argument"};
    hipStream_t s = c->stream;
    c->have_details = false;
    c->have_pairs = c->pairs_of_result = c->phase_a_done = false;
    c->res_ov.ensure((n_overlaps + 1) * sizeof(kslam_overlap));
    c->res_cig.ensure((n_cigar + 1) * sizeof(uint32_t));
    if (n_overlaps) HIPCHK(hipMemcpyAsync(c->res_ov.p, d_overlaps, n_overlaps * sizeof(kslam_overlap), hipMemcpyDeviceToDevice, s));
    if (n_cigar) HIPCHK(hipMemcpyAsync(c->res_cig.p, d_cigar_pool, n_cigar * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    HIPCHK(stream_wait(s));
    c->n_res = n_overlaps;
    c->n_cig = n_cigar;
  });
}

int32_t kslam_ctx_device(const kslam_ctx *c) { return c ? c->device : -1; }

kslam_status kslam_reload_tuning(kslam_ctx *c) {
  if (!c) return KSLAM_ERR_ARG;
  std::lock_guard<std::mutex> lk(c->as_mu);
  const Tuning t = read_tuning();
  const int lanes = c->lanes.empty() ? t.lanes : c->tune.lanes;   // the number of lanes is fixed once they exist
  c->tune = t;
  c->tune.lanes = lanes;
  c->pw.pseudo_cap = (uint32_t)c->tune.pseudo_cap;
  for (auto *l : c->lanes) { l->c->tune = c->tune; l->c->pw.pseudo_cap = (uint32_t)c->tune.pseudo_cap; }
  return KSLAM_OK;
}

kslam_status kslam_load_reads(kslam_ctx *c, uint64_t n_reads, const char *concat, const uint64_t *offsets) {
  return guarded(c, [&] {
    if (n_reads && (!concat || !offsets)) throw StatusError{KSLAM_ERR_ARG, "null reads/offsets"};
    c->have_reads = false;
    c->n_reads = n_reads;
    c->h_roff.assign(n_reads + 1, 0);
    const uint64_t o0 = n_reads ? offsets[0] : 0;
    for (uint64_t i = 0; i <= n_reads && n_reads; i++) c->h_roff[i] = offsets[i] - o0;
    const uint64_t total = c->h_roff[n_reads];
    c->r_bases.ensure(total + 64);
    if (total) HIPCHK(hipMemcpyAsync(c->r_bases.p, concat + o0, total, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->r_bases.as<uint8_t>() + total, 0, 64, c->stream));
    finish_load_reads(c);
  });
}

kslam_status kslam_load_reads_device(kslam_ctx *c, uint64_t n_reads, const void *d_concat,
                                     const uint64_t *h_offsets) {
  return guarded(c, [&] {
    if (n_reads && (!d_concat || !h_offsets)) throw StatusError{KSLAM_ERR_ARG, "null reads/offsets"};
    c->have_reads = false;
    c->n_reads = n_reads;
    c->h_roff.assign(n_reads + 1, 0);
    const uint64_t o0 = n_reads ? h_offsets[0] : 0;
    for (uint64_t i = 0; i <= n_reads && n_reads; i++) c->h_roff[i] = h_offsets[i] - o0;
    const uint64_t total = c->h_roff[n_reads];
    c->r_bases.ensure(total + 64);
    if (total)
      HIPCHK(hipMemcpyAsync(c->r_bases.p, (const uint8_t *)d_concat + o0, total, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->r_bases.as<uint8_t>() + total, 0, 64, c->stream));
    finish_load_reads(c);
  });
}

kslam_status kslam_align_resident(kslam_ctx *c, uint64_t *n_out, uint64_t *n_cigar) {
  return guarded(c, [&] {
    align_resident(c, false, nullptr);
    if (n_out) *n_out = c->n_res;
    if (n_cigar) *n_cigar = c->n_cig;
  });
}

kslam_status kslam_fetch_results(kslam_ctx *c, kslam_overlap *out, uint32_t *cigar_pool) {
  return guarded(c, [&] {
    if (c->n_res && out)
      HIPCHK(hipMemcpyAsync(out, c->res_ov.p, c->n_res * sizeof(kslam_overlap), hipMemcpyDeviceToHost, c->stream));
    if (c->n_cig && cigar_pool)
      HIPCHK(hipMemcpyAsync(cigar_pool, c->res_cig.p, c->n_cig * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(stream_wait(c->stream));
  });
}

kslam_status kslam_take_results(kslam_ctx *c, kslam_overlap **out, uint64_t *n_out, uint32_t **cigar_pool,
                                uint64_t *n_cigar) {
  if (!c || !out || !n_out || !cigar_pool || !n_cigar) return KSLAM_ERR_ARG;
  *out = nullptr; *cigar_pool = nullptr; *n_out = 0; *n_cigar = 0;
  kslam_overlap *ho = nullptr;
  uint32_t *hc = nullptr;
  kslam_status st = guarded(c, [&] {   // pinned: D2H at full PCIe rate, reused by later batches
    ho = (kslam_overlap *)pinned_get(c, (c->n_res + 1) * sizeof(kslam_overlap));
    hc = (uint32_t *)pinned_get(c, (c->n_cig + 1) * sizeof(uint32_t));
  });
  if (st == KSLAM_OK) st = kslam_fetch_results(c, ho, hc);
  if (st != KSLAM_OK) {
    if (ho) pinned_put(c, ho);
    if (hc) pinned_put(c, hc);
    return st;
  }
  *out = ho; *n_out = c->n_res; *cigar_pool = hc; *n_cigar = c->n_cig;
  return KSLAM_OK;
}

kslam_status kslam_copy_results_device(kslam_ctx *c, void *d_overlaps, void *d_cigar_pool) {
  return guarded(c, [&] {
    if (c->n_res && d_overlaps)
      HIPCHK(hipMemcpyAsync(d_overlaps, c->res_ov.p, c->n_res * sizeof(kslam_overlap), hipMemcpyDeviceToDevice,
                            c->stream));
    if (c->n_cig && d_cigar_pool)
      HIPCHK(hipMemcpyAsync(d_cigar_pool, c->res_cig.p, c->n_cig * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                            c->stream));
    HIPCHK(stream_wait(c->stream));
  });
}

kslam_status kslam_get_timings(const kslam_ctx *c, kslam_timings *out) {
  if (!c || !out) return KSLAM_ERR_ARG;
  *out = c->tm;
  return KSLAM_OK;
}

kslam_status kslam_align_batch(kslam_ctx *c, uint64_t n_reads, const char *const *bases, const uint32_t *lens,
                               kslam_overlap **out, uint64_t *n_out, uint32_t **cigar_pool, uint64_t *n_cigar) {
  if (!c || !out || !n_out || !cigar_pool || !n_cigar) return KSLAM_ERR_ARG;
  *out = nullptr; *cigar_pool = nullptr; *n_out = 0; *n_cigar = 0;
  std::vector<uint64_t> off(n_reads + 1, 0);
  char *cat = nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  kslam_status st = guarded(c, [&] {
    if (n_reads && (!bases || !lens)) throw StatusError{KSLAM_ERR_ARG, "null bases/lens"};
    for (uint64_t i = 0; i < n_reads; i++) off[i + 1] = off[i] + lens[i];
    // gather the reads into one pinned buffer, in parallel (2 M small copies per 1 M pairs)
    cat = (char *)pinned_get(c, off[n_reads] + 64);
    unsigned nt = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (n_reads < 100000) nt = 1;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) {
      const uint64_t lo = n_reads * t / nt, hi = n_reads * (t + 1) / nt;
      auto work = [=, &off] { for (uint64_t i = lo; i < hi; i++) memcpy(cat + off[i], bases[i], lens[i]); };
      if (nt == 1) work(); else th.emplace_back(work);
    }
    for (auto &x : th) x.join();
  });
  if (st != KSLAM_OK) { if (cat) pinned_put(c, cat); return st; }
  const double t1 = now();
  st = kslam_load_reads(c, n_reads, cat, off.data());
  pinned_put(c, cat);
  if (st != KSLAM_OK) return st;
  const double t2 = now();
  st = kslam_align_resident(c, nullptr, nullptr);
  if (st != KSLAM_OK) return st;
  const double t3 = now();
  st = kslam_take_results(c, out, n_out, cigar_pool, n_cigar);
  if (c->tune.debug)
    fprintf(stderr, "[kslam] align_batch: gather %.2f ms, load_reads %.2f, align %.2f, take_results %.2f\n", t1 - t0, t2 - t1, t3 - t2,
            now() - t3);
  return st;
}

void kslam_free_batch(kslam_ctx *c, kslam_overlap *out, uint32_t *cigar_pool) {
  if (!c) return;
  auto give_back = [&](void *p) {
    if (!p || pinned_put(c, p)) return;
    for (auto *l : c->lanes)
      if (pinned_put(l->c, p)) return;
    free(p);
  };
  give_back(out);
  give_back(cigar_pool);
}

void kslam_free_pinned(kslam_ctx *c, void *p) {
  if (!c || !p) return;
  if (pinned_put(c, p)) return;
  for (auto *l : c->lanes)
    if (pinned_put(l->c, p)) return;
}

void kslam_free(void *p) { free(p); }
void *kslam_host_alloc(uint64_t bytes) { return pinned_alloc((size_t)bytes); }
void kslam_host_free(void *p, uint64_t bytes) { pinned_free(p, (size_t)bytes); }

}  // extern "C"
