// kslam_api.hip -- the C ABI (include/kslam.h) and the host orchestration of
// the hot path: alignToDatabase (reference src/SLAM.h:59-79) as
//   extract read k-mers -> radix sort -> merge-join against the resident sorted
//   genome k-mer list -> overlap sort + dedupe -> SW scores -> banded CIGAR.
// All device work runs on the context's own HIP stream; phases are bracketed
// with HIP events.  No CPU fallback exists: without a HIP device every entry
// point fails with KSLAM_ERR_NO_DEVICE.
#include <sys/mman.h>

#include "common.h"
#include "samtext.h"
#include "../../include/kslam_samtext.h"
#include "../host/workers.hpp"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <sys/prctl.h>
#include <thread>

using namespace kslam;

struct kslam_ctx {
  kslam_params prm{};
  Tuning tune;                   // the KSLAM_* environment switches as they stood at kslam_create
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  hipEvent_t ev[16]{};
  hipEvent_t evs0[12]{}, evs1[12]{};   // per-pass events around the k-mer scatter kernel

  // ---- index (const GenbankIndex&) ----
  uint32_t group_route_pause = 0;    // chunks left on the long route after a chunk's overlap keys held a (read, entry) group too
                                     // long for join.hip's group_order (a read in a tandem repeat): such data comes in stretches,
                                     // and a chunk that tries the short route in vain pays for 4 radix passes too many
  bool have_index = false;
  kslam_index_stats index_stats{};   // phases of the last build_index (siblings / lanes: a copy of the primary's)
  uint64_t n_entries = 0;
  uint64_t max_entry_len = 0;
  std::vector<uint64_t> h_goff;  // [n_entries + 1]
  DevBuf g_bases, g_off, g_codes;   // g_codes: encode_bases(g_bases)
  uint64_t n_gk = 0;
  DevBuf gk_key, gk_meta, gk_off, g_bucket;
  uint32_t bucket_bits = 8;
  DevBuf g_filter;            // membership filter over the genome k-mers (filter.hip); filter_bits = 0: off
  uint32_t filter_bits = 0;
  uint64_t kept_last = 0;     // survivors of the last chunk (sizes the next chunk's record buffers)

  // ---- resident read batch ----
  bool have_reads = false;
  uint64_t n_reads = 0;
  uint32_t max_read_len = 0;
  // reads the packed SW / extraction kernels cannot hold (more than short_cap bases) are aligned in chunks of their own,
  // by the plain kernels (sw.hip: k_sw_long): class_runs = the read numbers at which the class (short / long) changes
  uint32_t short_cap = 511, max_short_len = 0;
  std::vector<uint64_t> class_runs;
  std::vector<uint64_t> h_roff;  // [n_reads + 1]
  std::vector<uint64_t> h_kpre, h_spre;   // [n_reads + 1] k-mers / extraction segments of the reads before i (chunk planning)
  DevBuf r_bases, r_off, r_len, r_codes;

  // ---- work buffers ----
  DevBuf nk, nseg, rec_start, seg_start, segs, scan_tmp, totals;
  DevBuf recs_a, recs_b, block_tot, block_base, ovk_a, ovk_b, flags, pos, band0;
  SortWorkspace sortws;
  CigarWork cig;
  SwWork sww;
  DevBuf cells;

  // ---- pinned host staging (host-pointer entry point): reused across batches ----
  // kslam_free_batch may run on another thread than the one taking results (a host-tail worker
  // hands buffers back while the main thread takes the next batch's): the pool has its own lock
  struct Pinned { void *p; size_t cap; bool in_use; };
  std::vector<Pinned> pinned;
  std::mutex pin_mu;

  // ---- per-row details for the SAM writer (details.hip) ----
  DevBuf r_qual, d_tables, res_det;
  DevBuf fq_text, fq_bases_at, fq_qual_at;   // kslam_submit_batch_fastq: the uploaded texts and field positions
  FastqWork fqw;                              // kslam_submit_batch_fastq_text: the record index built on the device
  bool have_qual = false, have_details = false;
  uint8_t *d_md_pool = nullptr;   // inside detw.md_pool
  uint64_t n_md = 0;
  uint32_t det_flags = 0;
  DetailWork detw;

  // ---- SAM records / per-read lines on the device (samtext.hip, include/kslam_samtext.h) ----
  SamAnnot annot{};               // device pointers; the primary context owns the buffers, its lanes read them
  std::vector<DevBuf> annot_bufs;
  bool have_annot = false;
  SamWork samw;
  struct { bool sam = false, per_read = false; uint32_t num_alignments = 10; int sam_xa = 0; } samtext;   // for the lanes
  const uint8_t *d_ids = nullptr;       // read identifiers of the loaded batch (fqw.ids, or ids_buf)
  const uint64_t *d_ids_off = nullptr;
  DevBuf ids_buf, ids_off_buf;
  bool have_ids = false;

  // ---- device pairing / screens (pairs.hip) ----
  PairWork pw;
  PairResult pres{};
  bool have_pairs = false;        // c->pres holds pairs (of res_ov, or of records handed in)
  bool pairs_of_result = false;   // ... and they index the rows of the current res_ov (kslam_pair_screen / the lane hook)
  bool phase_a_done = false;      // kslam_pair_phase_a ran on the current result, kslam_pair_phase_b has not yet
  DevBuf pr_ov, pr_len;          // kslam_pair_screen_overlaps: the records and read lengths handed in
  struct { int paired = 1; uint32_t thr = 0; double fraction = 0.95; uint32_t stages = 0; } pairing;   // for the lanes

  // ---- pipelined entry (kslam_align_batch_async): worker lanes, each a sibling context that BORROWS
  // this context's index (same device pointers, never freed by the sibling) ----
  bool borrowed_index = false;
  bool holds_hook = false;       // this context counts towards the page-locked column allocator being installed
  struct AsyncJob {
    uint64_t ticket = 0;
    uint64_t n_reads = 0;
    char *cat = nullptr;              // pinned, from the lane context's pool
    char *qcat = nullptr;             // the quality strings, same layout (optional)
    bool borrowed = false;            // cat / qcat / off_ptr are the caller's columns (kslam_submit_batch_columns)
    const uint64_t *off_ptr = nullptr;
    // kslam_submit_batch_fastq: the two texts (cat = r1, qcat = r2) and where the fields lie in [r1 | r2]
    bool fastq = false, fastq_text = false;   // fastq_text: the index is built on the device too
    bool single = false;                      // fastq_text with ONE stream (r2 == NULL): single-end reads
    uint64_t max_pairs = 0; int at_eof = 1;
    uint64_t r_n = 0, *r_off = nullptr, *r_ids_off = nullptr; char *r_ids = nullptr; uint64_t consumed[2] = {0, 0};
    uint64_t len1 = 0, len2 = 0;
    const uint64_t *bases_at = nullptr, *quality_at = nullptr;
    std::vector<uint64_t> off;
    bool done = false;
    kslam_status st = KSLAM_OK;
    std::string err;
    kslam_overlap *out = nullptr; uint64_t n_out = 0;
    uint32_t *pool = nullptr; uint64_t n_cig = 0;
    kslam_row_detail *det = nullptr; char *md = nullptr; uint64_t n_md = 0;
    kslam_read_pair *rp = nullptr; uint64_t n_rp = 0; kslam_paired_overlap *pr = nullptr; uint64_t n_pr = 0;
    kslam_pair_stats pstats{};
    char *sam_text = nullptr; uint64_t sam_len = 0; char *pr_text = nullptr; uint64_t pr_len = 0; uint32_t *tax = nullptr;
    uint32_t text_flags = 0;
  };
  struct AsyncLane {
    kslam_ctx *c = nullptr;
    std::thread th;
    std::deque<AsyncJob *> q;
  };
  std::vector<AsyncLane *> lanes;
  std::mutex as_mu, as_compute;
  std::condition_variable as_cv;
  std::map<uint64_t, AsyncJob *> jobs;   // submitted, not yet waited for
  uint64_t next_ticket = 0;
  bool as_stop = false;

  // ---- merge of gathered shard results (merge.hip) ----
  DevBuf mg_shards, mg_lens, mg_off, mg_scan;

  // ---- results of the last align ----
  DevBuf res_ov, res_cig, res_tmp, fin_copy, band0_all;
  uint64_t n_res = 0, n_cig = 0;
  kslam_timings tm{};
};

struct kslam_multi {
  std::vector<kslam_ctx *> ctx;
  std::string err;
  DevBuf rows_out, pool_out;     // on ctx[0]'s device: the batch-global result
  std::vector<DevBuf> send;      // per shard, on its own device: its records in batch terms, ready to copy
};

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

namespace {

void share_index(kslam_ctx *dst, const kslam_ctx *src);

template <typename F> kslam_status guarded(kslam_ctx *ctx, F &&f) {
  if (!ctx) return KSLAM_ERR_ARG;
  try {
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) {
      ctx->err = std::string("hipSetDevice failed: ") + hipGetErrorString(e);
      return KSLAM_ERR_NO_DEVICE;
    }
    f();
    return KSLAM_OK;
  } catch (const StatusError &se) {
    ctx->err = se.msg;
    return se.st;
  } catch (const HipError &he) {
    ctx->err = std::string("HIP error ") + hipGetErrorString(he.code) + " at " + he.file + ":" +
               std::to_string(he.line) + " in " + he.what;
    (void)hipGetLastError();
    return he.code == hipErrorOutOfMemory ? KSLAM_ERR_OOM : KSLAM_ERR_NO_DEVICE;
  } catch (const std::bad_alloc &) {
    ctx->err = "host allocation failed";
    return KSLAM_ERR_OOM;
  }
}

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

__global__ void k_lens(const uint64_t *off, uint64_t n, uint32_t *len) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) len[i] = (uint32_t)(off[i + 1] - off[i]);
}

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
                 uint64_t n_segs, uint4 *d_out, uint8_t *d_digits = nullptr, const SortPass *first_pass = nullptr) {
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
  HIPCHK(hipEventRecord(e1, s));
  void *sorted = radix_sort(c->recs_a.p, c->recs_b.p, m, 4, passes.data(), (int)passes.size(), c->sortws, s,
                            nullptr, nullptr, nullptr, /*setup=*/true);
  c->sortws.first_digits_ready = false;
  HIPCHK(hipEventRecord(e2, s));
  c->gk_key.ensure((m + 1) * sizeof(uint64_t));
  c->gk_meta.ensure((m + 1) * sizeof(uint2));   // {meta, offset} pairs
  if (m) hipLaunchKernelGGL(k_split_soa, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, (const uint4 *)sorted,
                            (uint32_t)m, c->gk_key.as<uint64_t>(), c->gk_meta.as<uint2>());
  uint32_t bits = 8, max_bits = (uint32_t)c->tune.bucket_bits_max;   // 27: ~2.3 genome k-mers per bucket for a 5 Gb database (537 MB table)
  while (bits < max_bits && (m >> (bits + 2)) != 0) bits++;   // 2 to 4 keys per bucket (measured: 3.06 ms at 27 bits, 3.24 at 26, 3.13 at 28)
  if (c->tune.bucket_bits_exact) bits = (uint32_t)c->tune.bucket_bits_exact;   // tuning
  c->bucket_bits = bits;
  c->g_bucket.ensure(((1ull << bits) + 2) * sizeof(uint32_t));
  build_bucket_table(c->gk_key.as<uint64_t>(), (uint32_t)m, bits, c->g_bucket.as<uint32_t>(), s);
  // membership filter for the read extraction: ~14 bits per genome k-mer (9.3 keys per 128-bit piece),
  // 2^32 bits = 512 MiB for the 312 M k-mers of a 5 Gb database.  KSLAM_FILTER_BITS: log2 of the size
  // in bits, 0 = extract, sort and look up every read k-mer as the reference does.
  {
    uint32_t fb = 20;
    while (fb < 35 && ((uint64_t)1 << fb) < m * 12) fb++;
    if (c->tune.filter_bits >= 0) fb = (uint32_t)c->tune.filter_bits;
    c->filter_bits = fb;
    if (fb) {
      c->g_filter.ensure(filter_bytes(fb));
      if (c->tune.filter_build_sorted) {
        // the probe words take the record buffers of the sort that has just finished (k_split_soa, queued above, was their last reader)
        c->pos.ensure((filter_bytes(fb) / 32768 + 2) * sizeof(uint32_t));
        filter_build_sorted(c->gk_key.as<uint64_t>(), (uint32_t)m, fb, c->g_filter.p, c->recs_a.p, c->recs_b.p, c->pos.as<uint32_t>(), c->sortws, s);
      } else {
        filter_build(c->gk_key.as<uint64_t>(), (uint32_t)m, fb, c->g_filter.p, s);
      }
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

bool scoring_in_envelope(const kslam_params &p);

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
struct PairingHook {
  int paired;
  uint32_t thr;
  double fraction;
  uint32_t stages;
  bool ran = false;    // false: the batch had several chunks (or nothing to pair): the caller pairs afterwards
};

// the hot path on the resident reads; stop_after_join: only rows a-3..a-6
void align_resident(kslam_ctx *c, bool stop_after_join, uint64_t *n_raw_out, PairingHook *hook = nullptr) {
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

kslam_status multi_fail(kslam_multi *m, kslam_status st, const std::string &msg) {
  m->err = msg;
  return st;
}

// run f(k) for every shard on its own host thread (every entry point of a context blocks on its stream)
template <typename F> kslam_status multi_for_each(kslam_multi *m, F &&f) {
  const size_t n = m->ctx.size();
  std::vector<kslam_status> st(n, KSLAM_OK);
  std::vector<std::thread> th;
  for (size_t k = 1; k < n; k++) th.emplace_back([&, k] { st[k] = f(k); });
  st[0] = f(0);
  for (auto &t : th) t.join();
  for (size_t k = 0; k < n; k++)
    if (st[k] != KSLAM_OK) return multi_fail(m, st[k], "shard " + std::to_string(k) + ": " + kslam_last_error(m->ctx[k]));
  return KSLAM_OK;
}


// a sibling context sees the primary's index through the same device pointers
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
// kslam_load_qualities would.
kslam_status load_reads_from_fastq(kslam_ctx *c, uint64_t n_reads, const char *r1, uint64_t len1, const char *r2,
                                   uint64_t len2, const uint64_t *offsets, const uint64_t *bases_at,
                                   const uint64_t *quality_at) {
  return guarded(c, [&] {
    if (n_reads && (!offsets || !bases_at || !quality_at || (len1 && !r1) || (len2 && !r2)))
      throw StatusError{KSLAM_ERR_ARG, "null argument"};
    hipStream_t s = c->stream;
    c->have_reads = false;
    c->n_reads = n_reads;
    c->h_roff.assign(n_reads + 1, 0);
    const uint64_t o0 = n_reads ? offsets[0] : 0;
    for (uint64_t i = 0; i <= n_reads && n_reads; i++) c->h_roff[i] = offsets[i] - o0;
    const uint64_t total = c->h_roff[n_reads];
    c->fq_text.ensure(len1 + len2 + 64);
    if (len1) HIPCHK(hipMemcpyAsync(c->fq_text.p, r1, len1, hipMemcpyHostToDevice, s));
    if (len2) HIPCHK(hipMemcpyAsync(c->fq_text.as<uint8_t>() + len1, r2, len2, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(c->fq_text.as<uint8_t>() + len1 + len2, 0, 64, s));
    c->fq_bases_at.ensure((n_reads + 1) * sizeof(uint64_t));
    c->fq_qual_at.ensure((n_reads + 1) * sizeof(uint64_t));
    c->r_off.ensure((n_reads + 1) * sizeof(uint64_t));
    if (n_reads) {
      HIPCHK(hipMemcpyAsync(c->fq_bases_at.p, bases_at, n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, s));
      HIPCHK(hipMemcpyAsync(c->fq_qual_at.p, quality_at, n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipMemcpyAsync(c->r_off.p, c->h_roff.data(), (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    c->r_bases.ensure(total + 64);
    c->r_qual.ensure(total + 64);
    // a field that reaches past the end of the texts would be a broken index: check on the host
    for (uint64_t i = 0; i < n_reads; i++) {
      const uint64_t len = c->h_roff[i + 1] - c->h_roff[i];
      if (bases_at[i] + len > len1 + len2 || quality_at[i] + len > len1 + len2)
        throw StatusError{KSLAM_ERR_ARG, "field " + std::to_string(i) + " lies outside the FASTQ texts"};
    }
    gather_fields(c->fq_text.as<uint8_t>(), c->fq_bases_at.as<uint64_t>(), c->fq_qual_at.as<uint64_t>(),
                  c->r_off.as<uint64_t>(), n_reads, c->r_bases.as<uint8_t>(), c->r_qual.as<uint8_t>(), s);
    HIPCHK(hipMemsetAsync(c->r_bases.as<uint8_t>() + total, 0, 64, s));
    HIPCHK(hipMemsetAsync(c->r_qual.as<uint8_t>() + total, 0, 64, s));
    finish_load_reads(c);
    c->have_qual = true;
  });
}


// kslam_submit_batch_fastq_text: texts up, record index + columns on the device, the host's columns back
kslam_status load_reads_from_fastq_text(kslam_ctx *c, kslam_ctx::AsyncJob *job) {
  return guarded(c, [&] {
    const char *r1 = job->cat, *r2 = job->qcat;
    const uint64_t len1 = job->len1, len2 = job->len2;
    if ((len1 && !r1) || (len2 && !r2)) throw StatusError{KSLAM_ERR_ARG, "null text"};
    hipStream_t s = c->stream;
    c->have_reads = false;
    c->fq_text.ensure(len1 + len2 + 64);
    if (len1) HIPCHK(hipMemcpyAsync(c->fq_text.p, r1, len1, hipMemcpyHostToDevice, s));
    if (len2) HIPCHK(hipMemcpyAsync(c->fq_text.as<uint8_t>() + len1, r2, len2, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(c->fq_text.as<uint8_t>() + len1 + len2, 0, 64, s));
    FastqIndexResult ix;
    fastq_index_device(c->fq_text.as<uint8_t>(), len1, len2, len1 ? (const uint8_t *)r1 + len1 - 1 : nullptr,
                       len2 ? (const uint8_t *)r2 + len2 - 1 : nullptr, job->max_pairs, job->at_eof != 0, c->fqw, &ix, s, job->single);
    const uint64_t n = ix.n_reads;
    // the host's columns: offsets (= lengths), identifiers
    job->r_n = n;
    job->r_off = (uint64_t *)pinned_get(c, (n + 2) * sizeof(uint64_t));
    job->r_ids_off = (uint64_t *)pinned_get(c, (n + 2) * sizeof(uint64_t));
    job->r_ids = (char *)pinned_get(c, ix.ids_total + 64);
    HIPCHK(hipMemcpyAsync(job->r_off, ix.d_bases_off, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(job->r_ids_off, ix.d_ids_off, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    if (ix.ids_total) HIPCHK(hipMemcpyAsync(job->r_ids, ix.d_ids, ix.ids_total, hipMemcpyDeviceToHost, s));
    job->consumed[0] = ix.consumed[0];
    job->consumed[1] = ix.consumed[1];
    // the device's columns
    c->n_reads = n;
    c->r_off.ensure((n + 1) * sizeof(uint64_t));
    HIPCHK(hipMemcpyAsync(c->r_off.p, ix.d_bases_off, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
    c->r_bases.ensure(ix.bases_total + 64);
    c->r_qual.ensure(ix.bases_total + 64);
    gather_fields(c->fq_text.as<uint8_t>(), ix.d_bases_at, ix.d_quality_at, c->r_off.as<uint64_t>(), n,
                  c->r_bases.as<uint8_t>(), c->r_qual.as<uint8_t>(), s);
    HIPCHK(hipMemsetAsync(c->r_bases.as<uint8_t>() + ix.bases_total, 0, 64, s));
    HIPCHK(hipMemsetAsync(c->r_qual.as<uint8_t>() + ix.bases_total, 0, 64, s));
    HIPCHK(stream_wait(s));
    job->r_ids[ix.ids_total] = 0;
    c->h_roff.assign(job->r_off, job->r_off + n + 1);
    finish_load_reads(c);
    c->have_qual = true;
    c->d_ids = ix.d_ids;                 // in c->fqw: valid until this context indexes its next batch
    c->d_ids_off = ix.d_ids_off;
    c->have_ids = true;
    job->n_reads = n;
  });
}

// ---- the SAM records / per-read lines on the device (include/kslam_samtext.h, csrc/samtext.hip) ---------------------
// ceil(-10 log10(t)) stored into a uint8_t, src/SAM.h:502-506, with THIS host's libm (host/tail.cpp: mapq_of, same code)
inline uint8_t mapq_of(double prob, double sum) {
  double t = 1.0 - prob / sum;
  if (t <= 0.00001) t = 0.00001;
  double q = ceil(-10.0 * std::log10(t));
  if (std::isnan(q)) return 0;
  return (uint8_t)q;
}

struct SamStage {   // one batch's way through the stage
  SamInputs in;
  SamParams P;
  kslam_paired_overlap *d_recs = nullptr;
  const kslam_read_pair *d_groups = nullptr;
  uint64_t n_groups = 0, n_vals = 0, n_segs = 0, text_bytes = 0, pr_bytes = 0;
  double *h_vals = nullptr;       // pinned
  uint32_t *h_seg = nullptr;      // pinned
  uint8_t *h_mapq = nullptr;      // pinned
};

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

void lane_main(kslam_ctx *primary, kslam_ctx::AsyncLane *lane) {
  kslam_host::name_thread("kslam-lane");
  wait_mode().yield = primary->tune.lane_waits_yield;   // common.h: stream_wait
  if (wait_mode().yield) (void)prctl(PR_SET_TIMERSLACK, 5000UL, 0, 0, 0);   // its 20 us sleeps mean 25, not 70
  for (;;) {
    kslam_ctx::AsyncJob *job = nullptr;
    {
      std::unique_lock<std::mutex> lk(primary->as_mu);
      primary->as_cv.wait(lk, [&] { return primary->as_stop || !lane->q.empty(); });
      if (lane->q.empty()) return;   // stop requested and nothing left
      job = lane->q.front();
      lane->q.pop_front();
    }
    kslam_ctx *c = lane->c;
    const bool dbg = primary->tune.debug;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    double t1 = 0, t2 = 0, t3 = 0;
    kslam_status st;
    if (job->fastq_text)
      st = load_reads_from_fastq_text(c, job);
    else if (job->fastq)
      st = load_reads_from_fastq(c, job->n_reads, job->cat, job->len1, job->qcat, job->len2, job->off_ptr,
                                 job->bases_at, job->quality_at);
    else
      st = kslam_load_reads(c, job->n_reads, job->cat, job->borrowed ? job->off_ptr : job->off.data());
    if (st == KSLAM_OK && job->qcat && !job->fastq)
      st = kslam_load_qualities(c, job->borrowed && job->n_reads ? job->qcat + job->off_ptr[0] : job->qcat);
    if (job->qcat && !job->borrowed) { pinned_put(c, job->qcat); }
    t1 = now();
    if (!job->borrowed) pinned_put(c, job->cat);
    job->cat = nullptr;
    SamStage sam;
    bool sam_planned = false, text_wanted = false, details_wanted = false;
    auto want_details_or_no_cigar = [](kslam_ctx *cc, bool wd) { return wd || !cc->prm.report_cigar; };
    if (st == KSLAM_OK) {
      // one lane computes at a time: the kernels of a batch fill the chip, so two batches computing at
      // once only time-slice -- and, worse, fall into step, both lanes copying while the GPU idles and
      // both computing afterwards (measured: 32.6 ms per batch against 27.3 resident).  With the token
      // the lanes run in anti-phase: one computes while the other downloads its last result and
      // uploads its next batch.
      std::lock_guard<std::mutex> compute(primary->as_compute);
      t2 = now();
      const bool want_details = job->qcat || job->fastq;
      PairingHook hook{primary->pairing.paired, primary->pairing.thr, primary->pairing.fraction, primary->pairing.stages};
      const bool eager = primary->tune.eager_cigar;   // A/B: every CIGAR, pairing afterwards
      const bool use_hook = primary->pairing.stages && want_details && !eager;
      st = guarded(c, [&] { align_resident(c, false, nullptr, use_hook ? &hook : nullptr); });
      if (st == KSLAM_OK && primary->pairing.stages) {
        if (hook.ran) fill_pair_stats(c->pres, &job->pstats);
        else st = kslam_pair_screen(c, primary->pairing.paired, primary->pairing.thr, primary->pairing.fraction,
                                    primary->pairing.stages, &job->pstats);
      }
      details_wanted = st == KSLAM_OK && want_details;
      text_wanted = st == KSLAM_OK && want_details_or_no_cigar(c, want_details);
      if (details_wanted && primary->tune.details_in_token) {   // default; KSLAM_DETAILS_IN_TOKEN=0 moves it out (measured: 40.5 against 43.1 M reads/s)
        st = primary->pairing.stages ? kslam_row_details_of_pairs(c, nullptr) : kslam_row_details(c, nullptr);
        details_wanted = false;
        if (st != KSLAM_OK) text_wanted = false;
      }
    }
    // (KSLAM_DETAILS_IN_TOKEN=0, A/B: the per-row walk outside the token as well -- slower: its 150-byte windows of the index
    // compete with the other lane's staging for L2)
    if (details_wanted) {
      st = primary->pairing.stages ? kslam_row_details_of_pairs(c, nullptr) : kslam_row_details(c, nullptr);
      if (st != KSLAM_OK) text_wanted = false;
    }
    // The SAM records / per-read lines on the device (kslam_set_sam_text): the reference's per-pair sort (in place: the pairs
    // go back to the host in that order), the rows to report, the log-probabilities the host evaluates with its libm, then the
    // text.  OUTSIDE the compute token: these kernels are bound by the latency of dependent gathers (samtext.hip), not by
    // ALU work or bandwidth, so they run next to the other lane's alignment kernels instead of in front of them.
    // Not for a batch whose pseudo-assembly the device left to the host: its scores are not final yet.
    const bool text_on = primary->samtext.sam || primary->samtext.per_read;
    const bool pseudo_left = (primary->pairing.stages & 4u) && !(job->pstats.stages_done & 4u);
    if (text_wanted && text_on && primary->pairing.stages && c->have_ids && !pseudo_left) {
      st = guarded(c, [&] {
        sam_stage_plan(c, primary, primary->pairing.paired, primary->samtext.num_alignments, primary->samtext.sam_xa, primary->samtext.sam, sam);
      });
      sam_planned = st == KSLAM_OK;
    }
    if (sam_planned) {
      sam_stage_mapq(sam);   // pow / log10 / ceil with the host's libm
      st = guarded(c, [&] { sam_stage_kernels(c, primary, sam, primary->samtext.sam, primary->samtext.per_read); });
      uint64_t n_tax = 0;
      if (st == KSLAM_OK)
        st = guarded(c, [&] {
          sam_stage_fetch(c, sam, primary->samtext.sam, primary->samtext.per_read, &job->sam_text, &job->sam_len, &job->pr_text, &job->pr_len,
                          &job->tax, &n_tax);
        });
      if (st == KSLAM_OK)
        job->text_flags = (primary->samtext.sam ? (KSLAM_TEXT_PAIRS_SORTED | KSLAM_TEXT_SAM) : 0u) | (primary->samtext.per_read ? KSLAM_TEXT_PER_READ : 0u);
    }
    sam_stage_free(c, sam);
    t3 = now();
    // with the SAM records written on the device the host has no use for the rows, the CIGAR pool, the per-row details and
    // the MD text (0.7 GB per batch of configs[1]): they stay where they are, only their counts travel
    const bool text_sam = (job->text_flags & KSLAM_TEXT_SAM) != 0;
    if (st == KSLAM_OK && !text_sam) st = kslam_take_results(c, &job->out, &job->n_out, &job->pool, &job->n_cig);
    if (st == KSLAM_OK && text_sam) { job->n_out = c->n_res; job->n_cig = c->n_cig; }
    const double t4 = now();
    if (st == KSLAM_OK && (job->qcat || job->fastq) && !text_sam) st = kslam_take_row_details(c, &job->det, &job->md, &job->n_md);
    const double t5 = now();
    if (st == KSLAM_OK && primary->pairing.stages) st = kslam_take_pairs(c, &job->rp, &job->n_rp, &job->pr, &job->n_pr);
    if (dbg) fprintf(stderr, "[kslam]   align phases: extract %.2f sort %.2f join %.2f sw %.2f cigar %.2f total %.2f ms\n", c->tm.ms_extract,
                     c->tm.ms_sort, c->tm.ms_join, c->tm.ms_sw, c->tm.ms_cigar, c->tm.ms_total);
    if (dbg) fprintf(stderr, "[kslam] t=%.1f lane %p ticket %llu: upload %.2f, token wait %.2f, align %.2f, download %.2f (rows %.2f, details %.2f, pairs %.2f) ms\n",
                     fmod(t0, 100000.0), (void *)lane, (unsigned long long)job->ticket, t1 - t0, t2 - t1, t3 - t2, now() - t3, t4 - t3, t5 - t4, now() - t5);
    {
      std::lock_guard<std::mutex> lk(primary->as_mu);
      job->st = st;
      if (st != KSLAM_OK) job->err = c->err;
      job->done = true;
    }
    primary->as_cv.notify_all();
  }
}


void ensure_lanes(kslam_ctx *c) {
  if (!c->lanes.empty()) return;
  const int n_lanes = c->tune.lanes;
  // built aside and published only when every lane has its context AND its thread: a failure half way
  // (page-locked or device memory) must not leave lanes without workers behind, to which the next
  // submit would queue a job nobody ever runs
  std::vector<kslam_ctx::AsyncLane *> fresh;
  auto undo = [&] {
    for (auto *l : fresh) { kslam_destroy(l->c); delete l; }
    fresh.clear();
  };
  for (int k = 0; k < n_lanes; k++) {
    kslam_ctx *lc = nullptr;
    const kslam_status s1 = kslam_create(&c->prm, &lc);
    if (s1 != KSLAM_OK) {
      const std::string msg = lc ? lc->err : "lane context";
      kslam_destroy(lc);
      undo();
      throw StatusError{s1, msg};
    }
    lc->tune = c->tune;
    lc->pw.pseudo_cap = (uint32_t)c->tune.pseudo_cap;
    share_index(lc, c);
    auto *l = new kslam_ctx::AsyncLane();
    l->c = lc;
    fresh.push_back(l);
  }
  size_t started = 0;
  try {
    for (auto *l : fresh) { l->th = std::thread(lane_main, c, l); started++; }
  } catch (const std::exception &e) {
    { std::lock_guard<std::mutex> lk(c->as_mu); c->as_stop = true; }
    c->as_cv.notify_all();
    for (size_t k = 0; k < started; k++) fresh[k]->th.join();
    c->as_stop = false;
    undo();
    throw StatusError{KSLAM_ERR_OOM, std::string("could not start a lane thread: ") + e.what()};
  }
  c->lanes = std::move(fresh);
}

void stop_lanes(kslam_ctx *c) {
  {
    std::lock_guard<std::mutex> lk(c->as_mu);
    c->as_stop = true;
  }
  c->as_cv.notify_all();
  for (auto *l : c->lanes) {
    if (l->th.joinable()) l->th.join();
    kslam_destroy(l->c);
    delete l;
  }
  c->lanes.clear();
  for (auto &kv : c->jobs) {   // results nobody waited for
    kslam_ctx::AsyncJob *j = kv.second;
    delete j;
  }
  c->jobs.clear();
  c->as_stop = false;
}

}  // namespace

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

kslam_status kslam_index_build_stats(const kslam_ctx *c, kslam_index_stats *out) {
  if (!c || !out) return KSLAM_ERR_ARG;
  if (!c->have_index) return KSLAM_ERR_STATE;
  *out = c->index_stats;
  return KSLAM_OK;
}

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

void kslam_free_pinned(kslam_ctx *c, void *p) {
  if (!c || !p) return;
  if (pinned_put(c, p)) return;
  for (auto *l : c->lanes)
    if (pinned_put(l->c, p)) return;
}

// ---- the operator, pipelined: batches alternate between two worker lanes (a host thread + a sibling
// context with its own stream and work buffers each), so the upload of batch k+1 and the download of
// batch k-1 run under the kernels of batch k, and one lane's host read-backs are covered by the other
// lane's kernels ----
kslam_status kslam_align_batch_async(kslam_ctx *c, uint64_t n_reads, const char *const *bases, const uint32_t *lens,
                                     uint64_t *ticket) {
  return kslam_submit_batch(c, n_reads, bases, nullptr, lens, ticket);
}

kslam_status kslam_submit_batch(kslam_ctx *c, uint64_t n_reads, const char *const *bases, const char *const *quality,
                                const uint32_t *lens, uint64_t *ticket) {
  if (!c || !ticket) return KSLAM_ERR_ARG;
  kslam_ctx::AsyncJob *job = nullptr;
  kslam_status st = guarded(c, [&] {
    if (n_reads && (!bases || !lens)) throw StatusError{KSLAM_ERR_ARG, "null bases/lens"};
    if (!c->have_index) throw StatusError{KSLAM_ERR_STATE, "kslam_set_index has not been called"};
    ensure_lanes(c);
    job = new kslam_ctx::AsyncJob();
    job->n_reads = n_reads;
    job->off.assign(n_reads + 1, 0);
    for (uint64_t i = 0; i < n_reads; i++) job->off[i + 1] = job->off[i] + lens[i];
  });
  if (st != KSLAM_OK) { delete job; return st; }
  // the reads leave the caller's memory now (parallel gather into a page-locked buffer of the lane that
  // will run the batch): the caller may reuse its buffers as soon as this call returns
  uint64_t tk;
  {
    std::lock_guard<std::mutex> lk(c->as_mu);
    tk = c->next_ticket++;
  }
  kslam_ctx::AsyncLane *lane = c->lanes[tk % c->lanes.size()];
  st = guarded(c, [&] {
    job->cat = (char *)pinned_get(lane->c, job->off[n_reads] + 64);
    if (quality) job->qcat = (char *)pinned_get(lane->c, job->off[n_reads] + 64);
    unsigned nt = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (n_reads < 100000) nt = 1;
    std::vector<std::thread> th;
    char *cat = job->cat, *qcat = job->qcat;
    const std::vector<uint64_t> &off = job->off;
    for (unsigned t = 0; t < nt; t++) {
      const uint64_t lo = n_reads * t / nt, hi = n_reads * (t + 1) / nt;
      auto work = [=, &off] {
        for (uint64_t i = lo; i < hi; i++) memcpy(cat + off[i], bases[i], lens[i]);
        if (qcat) for (uint64_t i = lo; i < hi; i++) memcpy(qcat + off[i], quality[i], lens[i]);
      };
      if (nt == 1) work(); else th.emplace_back(work);
    }
    for (auto &x : th) x.join();
  });
  if (st != KSLAM_OK) {
    if (job->cat) pinned_put(lane->c, job->cat);
    if (job->qcat) pinned_put(lane->c, job->qcat);
    delete job;
    return st;
  }
  job->ticket = tk;
  {
    std::lock_guard<std::mutex> lk(c->as_mu);
    c->jobs[tk] = job;
    lane->q.push_back(job);
  }
  c->as_cv.notify_all();
  *ticket = tk;
  return KSLAM_OK;
}

kslam_status kslam_submit_batch_columns(kslam_ctx *c, uint64_t n_reads, const char *bases, const char *quality,
                                        const uint64_t *offsets, uint64_t *ticket) {
  if (!c || !ticket) return KSLAM_ERR_ARG;
  kslam_ctx::AsyncJob *job = nullptr;
  kslam_status st = guarded(c, [&] {
    if (n_reads && (!bases || !offsets)) throw StatusError{KSLAM_ERR_ARG, "null bases/offsets"};
    if (!c->have_index) throw StatusError{KSLAM_ERR_STATE, "kslam_set_index has not been called"};
    ensure_lanes(c);
    job = new kslam_ctx::AsyncJob();
    job->n_reads = n_reads;
    job->borrowed = true;
    job->cat = const_cast<char *>(bases);
    job->qcat = const_cast<char *>(quality);
    job->off_ptr = offsets;
  });
  if (st != KSLAM_OK) { delete job; return st; }
  uint64_t tk;
  {
    std::lock_guard<std::mutex> lk(c->as_mu);
    tk = c->next_ticket++;
    job->ticket = tk;
    c->jobs[tk] = job;
    c->lanes[tk % c->lanes.size()]->q.push_back(job);
  }
  c->as_cv.notify_all();
  *ticket = tk;
  return KSLAM_OK;
}

kslam_status kslam_submit_batch_fastq(kslam_ctx *c, const char *r1, uint64_t len1, const char *r2, uint64_t len2,
                                      uint64_t n_reads, const uint64_t *offsets, const uint64_t *bases_at,
                                      const uint64_t *quality_at, uint64_t *ticket) {
  if (!c || !ticket) return KSLAM_ERR_ARG;
  kslam_ctx::AsyncJob *job = nullptr;
  kslam_status st = guarded(c, [&] {
    if (n_reads && (!offsets || !bases_at || !quality_at)) throw StatusError{KSLAM_ERR_ARG, "null layout"};
    if (!c->have_index) throw StatusError{KSLAM_ERR_STATE, "kslam_set_index has not been called"};
    ensure_lanes(c);
    job = new kslam_ctx::AsyncJob();
    job->n_reads = n_reads;
    job->borrowed = true;
    job->fastq = true;
    job->cat = const_cast<char *>(r1);
    job->qcat = const_cast<char *>(r2);
    job->len1 = len1; job->len2 = len2;
    job->off_ptr = offsets;
    job->bases_at = bases_at; job->quality_at = quality_at;
  });
  if (st != KSLAM_OK) { delete job; return st; }
  uint64_t tk;
  {
    std::lock_guard<std::mutex> lk(c->as_mu);
    tk = c->next_ticket++;
    job->ticket = tk;
    c->jobs[tk] = job;
    c->lanes[tk % c->lanes.size()]->q.push_back(job);
  }
  c->as_cv.notify_all();
  *ticket = tk;
  return KSLAM_OK;
}

kslam_status kslam_submit_batch_fastq_text(kslam_ctx *c, const char *r1, uint64_t len1, const char *r2, uint64_t len2,
                                           uint64_t max_pairs, int at_eof, uint64_t *ticket) {
  if (!c || !ticket) return KSLAM_ERR_ARG;
  kslam_ctx::AsyncJob *job = nullptr;
  kslam_status st = guarded(c, [&] {
    if ((len1 && !r1) || (len2 && !r2)) throw StatusError{KSLAM_ERR_ARG, "null text"};
    if (!c->have_index) throw StatusError{KSLAM_ERR_STATE, "kslam_set_index has not been called"};
    ensure_lanes(c);
    job = new kslam_ctx::AsyncJob();
    job->borrowed = true;
    job->fastq = true;
    job->fastq_text = true;
    job->cat = const_cast<char *>(r1);
    job->qcat = const_cast<char *>(r2);
    job->len1 = len1; job->len2 = len2;
    job->max_pairs = max_pairs; job->at_eof = at_eof;
    job->single = r2 == nullptr && len2 == 0;
  });
  if (st != KSLAM_OK) { delete job; return st; }
  uint64_t tk;
  {
    std::lock_guard<std::mutex> lk(c->as_mu);
    tk = c->next_ticket++;
    job->ticket = tk;
    c->jobs[tk] = job;
    c->lanes[tk % c->lanes.size()]->q.push_back(job);
  }
  c->as_cv.notify_all();
  *ticket = tk;
  return KSLAM_OK;
}

kslam_status kslam_wait_batch(kslam_ctx *c, uint64_t ticket, kslam_overlap **out, uint64_t *n_out, uint32_t **cigar_pool,
                              uint64_t *n_cigar) {
  if (!c || !out || !n_out || !cigar_pool || !n_cigar) return KSLAM_ERR_ARG;
  *out = nullptr; *cigar_pool = nullptr; *n_out = 0; *n_cigar = 0;
  kslam_batch_result r;
  const kslam_status st = kslam_collect_batch(c, ticket, &r);
  if (st != KSLAM_OK) return st;
  kslam_free_pinned(c, r.details);
  kslam_free_pinned(c, r.md_pool);
  kslam_free_pinned(c, r.read_pairs);
  kslam_free_pinned(c, r.pairs);
  kslam_free_pinned(c, r.reads_bases_off);
  kslam_free_pinned(c, r.reads_ids_off);
  kslam_free_pinned(c, r.reads_ids);
  *out = r.overlaps; *n_out = r.n_overlaps; *cigar_pool = r.cigar_pool; *n_cigar = r.n_cigar;
  return KSLAM_OK;
}

void kslam_release_batch(kslam_ctx *c, kslam_batch_result *r) {
  if (!c || !r) return;
  kslam_free_batch(c, r->overlaps, r->cigar_pool);
  kslam_free_pinned(c, r->details);
  kslam_free_pinned(c, r->md_pool);
  kslam_free_pinned(c, r->read_pairs);
  kslam_free_pinned(c, r->pairs);
  kslam_free_pinned(c, r->reads_bases_off);
  kslam_free_pinned(c, r->reads_ids_off);
  kslam_free_pinned(c, r->reads_ids);
  kslam_free_pinned(c, r->sam_text);
  kslam_free_pinned(c, r->per_read_text);
  kslam_free_pinned(c, r->tax_ids);
  memset(r, 0, sizeof *r);
}

kslam_status kslam_collect_batch(kslam_ctx *c, uint64_t ticket, kslam_batch_result *res) {
  if (!c || !res) return KSLAM_ERR_ARG;
  memset(res, 0, sizeof *res);
  kslam_ctx::AsyncJob *job = nullptr;
  {
    std::unique_lock<std::mutex> lk(c->as_mu);
    auto it = c->jobs.find(ticket);
    if (it == c->jobs.end()) { c->err = "no such ticket (already waited for?)"; return KSLAM_ERR_ARG; }
    job = it->second;
    c->as_cv.wait(lk, [&] { return job->done; });
    c->jobs.erase(it);
  }
  const kslam_status st = job->st;
  if (st == KSLAM_OK) {
    res->overlaps = job->out; res->n_overlaps = job->n_out; res->cigar_pool = job->pool; res->n_cigar = job->n_cig;
    res->details = job->det; res->md_pool = job->md; res->n_md = job->n_md;
    res->read_pairs = job->rp; res->n_read_pairs = job->n_rp; res->pairs = job->pr; res->n_pairs = job->n_pr;
    res->pair_stats = job->pstats;
    res->n_reads = job->r_n; res->reads_bases_off = job->r_off; res->reads_ids = job->r_ids; res->reads_ids_off = job->r_ids_off;
    res->consumed1 = job->consumed[0]; res->consumed2 = job->consumed[1];
    res->sam_text = job->sam_text; res->sam_text_len = job->sam_len; res->per_read_text = job->pr_text; res->per_read_len = job->pr_len;
    res->tax_ids = job->tax; res->text_flags = job->text_flags;
  } else {
    kslam_free_pinned(c, job->sam_text);
    kslam_free_pinned(c, job->pr_text);
    kslam_free_pinned(c, job->tax);
    c->err = job->err;
    kslam_free_batch(c, job->out, job->pool);
    kslam_free_pinned(c, job->det);
    kslam_free_pinned(c, job->md);
    kslam_free_pinned(c, job->rp);
    kslam_free_pinned(c, job->pr);
    kslam_free_pinned(c, job->r_off);
    kslam_free_pinned(c, job->r_ids_off);
    kslam_free_pinned(c, job->r_ids);
  }
  delete job;
  return st;
}
void kslam_free(void *p) { free(p); }
void *kslam_host_alloc(uint64_t bytes) { return pinned_alloc((size_t)bytes); }
void kslam_host_free(void *p, uint64_t bytes) { pinned_free(p, (size_t)bytes); }

kslam_status kslam_merge_shards_device(kslam_ctx *c, uint32_t n_shards, const kslam_shard *shards, uint64_t n_pairs,
                                       const void *d_overlaps, const void *d_cigar_pools, void *d_out_overlaps,
                                       void *d_out_cigars) {
  return guarded(c, [&] {
    if (!shards || n_shards == 0 || n_shards > MERGE_MAX_SHARDS) throw StatusError{KSLAM_ERR_ARG, "1..256 shards"};
    std::vector<MergeShard> h(n_shards);
    uint64_t rows = 0, ops = 0;
    for (uint32_t k = 0; k < n_shards; k++) {
      const kslam_shard &sh = shards[k];
      if (sh.pair_hi < sh.pair_lo || sh.pair_hi > n_pairs || sh.pair_hi - sh.pair_lo >= (1ull << 31))
        throw StatusError{KSLAM_ERR_ARG, "shard " + std::to_string(k) + ": bad pair range"};
      if (k && sh.pair_lo < shards[k - 1].pair_hi) throw StatusError{KSLAM_ERR_ARG, "shards must be in batch order"};
      memset(&h[k], 0, sizeof(MergeShard));
      h[k].pair_lo = sh.pair_lo; h[k].pair_hi = sh.pair_hi;
      h[k].row_base = rows; h[k].n_rows = sh.n_rows; h[k].pool_base = ops;
      rows += sh.n_rows; ops += sh.n_cigar;
    }
    if (n_pairs >= (1ull << 31)) throw StatusError{KSLAM_ERR_UNSUPPORTED, "more than 2^31 pairs in one batch"};
    if (rows && (!d_overlaps || !d_out_overlaps)) throw StatusError{KSLAM_ERR_ARG, "null overlap buffers"};
    if (ops && (!d_cigar_pools || !d_out_cigars)) throw StatusError{KSLAM_ERR_ARG, "null cigar buffers"};
    hipStream_t s = c->stream;
    c->mg_shards.ensure(n_shards * sizeof(MergeShard));
    c->mg_lens.ensure((rows + 1) * sizeof(uint32_t));
    c->mg_off.ensure((rows + 1) * sizeof(uint64_t));
    c->mg_scan.ensure(scan_tmp_bytes(std::max<uint64_t>(rows, 1)));
    c->totals.ensure(8 * sizeof(uint64_t));
    HIPCHK(hipMemcpyAsync(c->mg_shards.p, h.data(), n_shards * sizeof(MergeShard), hipMemcpyHostToDevice, s));
    merge_shards((const kslam_overlap *)d_overlaps, rows, (const uint32_t *)d_cigar_pools, c->mg_shards.as<MergeShard>(),
                 n_shards, n_pairs, (kslam_overlap *)d_out_overlaps, (uint32_t *)d_out_cigars, c->mg_lens.as<uint32_t>(),
                 c->mg_off.as<uint64_t>(), c->totals.as<uint64_t>() + 4, c->mg_scan.p, s);
    HIPCHK(stream_wait(s));   // h[] is read by the copy above
  });
}

kslam_status kslam_shard_counts_device(kslam_ctx *c, uint64_t n_local_pairs, kslam_shard_counts *out) {
  return guarded(c, [&] {
    if (!out) throw StatusError{KSLAM_ERR_ARG, "null out"};
    if (n_local_pairs >= (1ull << 31)) throw StatusError{KSLAM_ERR_ARG, "n_local_pairs"};
    c->totals.ensure(8 * sizeof(uint64_t));
    uint64_t *d = c->totals.as<uint64_t>() + 4;
    shard_counts(c->res_ov.as<kslam_overlap>(), c->n_res, (uint32_t)n_local_pairs, c->n_cig, d, c->stream);
    uint64_t h[2] = {0, 0};
    read_back(h, d, sizeof h, c->stream);
    out->n_rows = c->n_res; out->n_rows_r1 = h[0];
    out->n_cigar = c->n_cig; out->n_cigar_r1 = h[1] == ~0ull ? c->n_cig : h[1];
  });
}

kslam_status kslam_export_shard_device(kslam_ctx *c, uint64_t n_local_pairs, uint64_t pair_lo, uint64_t n_pairs_total,
                                       uint64_t pool_base_r1, uint64_t pool_base_r2, void *d_rows_r1, void *d_rows_r2,
                                       void *d_pool_r1, void *d_pool_r2) {
  return guarded(c, [&] {
    kslam_shard_counts sc;
    // (the split again: cheap, and the caller cannot hand in counts that do not match the results)
    c->totals.ensure(8 * sizeof(uint64_t));
    uint64_t *d = c->totals.as<uint64_t>() + 4;
    shard_counts(c->res_ov.as<kslam_overlap>(), c->n_res, (uint32_t)n_local_pairs, c->n_cig, d, c->stream);
    uint64_t h[2] = {0, 0};
    read_back(h, d, sizeof h, c->stream);
    sc.n_rows = c->n_res; sc.n_rows_r1 = h[0]; sc.n_cigar = c->n_cig; sc.n_cigar_r1 = h[1] == ~0ull ? c->n_cig : h[1];
    if (pair_lo + n_local_pairs > n_pairs_total || n_pairs_total >= (1ull << 31))
      throw StatusError{KSLAM_ERR_ARG, "pair range outside the batch"};
    if ((sc.n_rows_r1 && !d_rows_r1) || (sc.n_rows > sc.n_rows_r1 && !d_rows_r2) || (sc.n_cigar_r1 && !d_pool_r1) ||
        (sc.n_cigar > sc.n_cigar_r1 && !d_pool_r2))
      throw StatusError{KSLAM_ERR_ARG, "null destination"};
    export_rows(c->res_ov.as<kslam_overlap>(), sc.n_rows, sc.n_rows_r1, (uint32_t)n_local_pairs, pair_lo, n_pairs_total,
                sc.n_cigar_r1, pool_base_r1, pool_base_r2, (kslam_overlap *)d_rows_r1, (kslam_overlap *)d_rows_r2,
                c->stream);
    if (sc.n_cigar_r1)
      HIPCHK(hipMemcpyAsync(d_pool_r1, c->res_cig.p, sc.n_cigar_r1 * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    if (sc.n_cigar > sc.n_cigar_r1)
      HIPCHK(hipMemcpyAsync(d_pool_r2, c->res_cig.as<uint32_t>() + sc.n_cigar_r1,
                            (sc.n_cigar - sc.n_cigar_r1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(stream_wait(c->stream));
  });
}

// ---- one process, several devices ---------------------------------------------------------------
kslam_status kslam_multi_create(const kslam_params *params, const int32_t *devices, uint32_t n_devices, kslam_multi **out) {
  if (!params || !devices || !out || n_devices == 0 || n_devices > MERGE_MAX_SHARDS) return KSLAM_ERR_ARG;
  kslam_multi *m = new (std::nothrow) kslam_multi();
  if (!m) return KSLAM_ERR_OOM;
  *out = m;
  for (uint32_t k = 0; k < n_devices; k++) {
    kslam_params p = *params;
    p.device = devices[k];
    kslam_ctx *c = nullptr;
    const kslam_status st = kslam_create(&p, &c);
    if (st != KSLAM_OK) {
      m->err = "device " + std::to_string(devices[k]) + ": " + (c ? kslam_last_error(c) : "create failed");
      kslam_destroy(c);
      return st;   // the caller reads the message and destroys m
    }
    m->ctx.push_back(c);
  }
  m->send.resize(n_devices);
  // peer access from the collecting device to the others (hipMemcpyPeerAsync works without it, through
  // the host; with it the copy goes over xGMI)
  for (uint32_t k = 1; k < n_devices; k++) {
    if (devices[k] == devices[0]) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, devices[0], devices[k]) == hipSuccess && can) {
      (void)hipSetDevice(devices[0]);
      (void)hipDeviceEnablePeerAccess(devices[k], 0);
      (void)hipGetLastError();   // "already enabled" is fine
    }
  }
  return KSLAM_OK;
}

void kslam_multi_destroy(kslam_multi *m) {
  if (!m) return;
  if (!m->ctx.empty() && m->ctx[0]->device >= 0) {
    (void)hipSetDevice(m->ctx[0]->device);
    m->rows_out.release(); m->pool_out.release();
  }
  for (size_t k = 0; k < m->send.size() && k < m->ctx.size(); k++)
    if (m->ctx[k]->device >= 0) { (void)hipSetDevice(m->ctx[k]->device); m->send[k].release(); }
  for (kslam_ctx *c : m->ctx) kslam_destroy(c);
  delete m;
}

const char *kslam_multi_last_error(const kslam_multi *m) { return m ? m->err.c_str() : "null handle"; }

kslam_status kslam_multi_set_index(kslam_multi *m, uint64_t n_entries, const char *const *bases, const uint64_t *lens) {
  if (!m || m->ctx.empty()) return KSLAM_ERR_ARG;
  return multi_for_each(m, [&](size_t k) { return kslam_set_index(m->ctx[k], n_entries, bases, lens); });
}

kslam_status kslam_multi_align_batch(kslam_multi *m, uint64_t n_reads, const char *const *bases, const uint32_t *lens,
                                     int paired, kslam_overlap **out, uint64_t *n_out, uint32_t **cigar_pool,
                                     uint64_t *n_cigar) {
  if (!m || m->ctx.empty() || !out || !n_out || !cigar_pool || !n_cigar) return KSLAM_ERR_ARG;
  *out = nullptr; *cigar_pool = nullptr; *n_out = 0; *n_cigar = 0;
  if (n_reads && (!bases || !lens)) return multi_fail(m, KSLAM_ERR_ARG, "null bases/lens");
  if (paired && (n_reads & 1)) return multi_fail(m, KSLAM_ERR_ARG, "a paired batch has an even number of reads");
  const uint64_t n_units = paired ? n_reads / 2 : n_reads;   // what is sharded: pairs, or single reads
  const size_t N = m->ctx.size();
  std::vector<kslam_shard> sh(N);
  // ---- shard + align, every device at once ----
  kslam_status st = multi_for_each(m, [&](size_t k) -> kslam_status {
    const uint64_t lo = n_units * k / N, hi = n_units * (k + 1) / N, nl = hi - lo;
    kslam_ctx *c = m->ctx[k];
    std::vector<uint64_t> off((paired ? 2 : 1) * nl + 1, 0);
    for (uint64_t i = 0; i < nl; i++) off[i + 1] = off[i] + lens[lo + i];
    if (paired) for (uint64_t i = 0; i < nl; i++) off[nl + i + 1] = off[nl + i] + lens[n_units + lo + i];
    const uint64_t n_loc = off.size() - 1;
    char *cat = nullptr;
    kslam_status s1 = guarded(c, [&] {
      cat = (char *)pinned_get(c, off[n_loc] + 64);
      for (uint64_t i = 0; i < nl; i++) memcpy(cat + off[i], bases[lo + i], lens[lo + i]);
      if (paired) for (uint64_t i = 0; i < nl; i++) memcpy(cat + off[nl + i], bases[n_units + lo + i], lens[n_units + lo + i]);
    });
    if (s1 == KSLAM_OK) s1 = kslam_load_reads(c, n_loc, cat, off.data());
    if (cat) pinned_put(c, cat);
    if (s1 == KSLAM_OK) s1 = kslam_align_resident(c, &sh[k].n_rows, &sh[k].n_cigar);
    sh[k].pair_lo = lo; sh[k].pair_hi = hi;
    return s1;
  });
  if (st != KSLAM_OK) return st;
  // ---- count exchange: where every shard's R1 rows, R2 rows and CIGAR words go in the batch ----
  std::vector<kslam_shard_counts> cnt(N);
  st = multi_for_each(m, [&](size_t k) { return kslam_shard_counts_device(m->ctx[k], sh[k].pair_hi - sh[k].pair_lo, &cnt[k]); });
  if (st != KSLAM_OK) return st;
  kslam_ctx *c0 = m->ctx[0];
  uint64_t rows = 0, ops = 0, rows_r1 = 0, ops_r1 = 0;
  for (size_t k = 0; k < N; k++) { rows += cnt[k].n_rows; ops += cnt[k].n_cigar; rows_r1 += cnt[k].n_rows_r1; ops_r1 += cnt[k].n_cigar_r1; }
  std::vector<uint64_t> row1(N), row2(N), op1(N), op2(N);
  {
    uint64_t a = 0, b = rows_r1, c = 0, d = ops_r1;
    for (size_t k = 0; k < N; k++) {
      row1[k] = a; a += cnt[k].n_rows_r1;
      row2[k] = b; b += cnt[k].n_rows - cnt[k].n_rows_r1;
      op1[k] = c; c += cnt[k].n_cigar_r1;
      op2[k] = d; d += cnt[k].n_cigar - cnt[k].n_cigar_r1;
    }
  }
  kslam_overlap *ho = nullptr;
  uint32_t *hc = nullptr;
  st = guarded(c0, [&] {
    m->rows_out.ensure((rows + 1) * sizeof(kslam_overlap));
    m->pool_out.ensure((ops + 1) * sizeof(uint32_t));
  });
  if (st != KSLAM_OK) return multi_fail(m, st, kslam_last_error(c0));
  // ---- the one exchange of the path: every shard re-bases its own records (its own GPU, all at once);
  // the shard on the collecting device writes straight into the final arrays, the others into a send
  // buffer that one peer copy per piece moves into place ----
  st = multi_for_each(m, [&](size_t k) -> kslam_status {
    kslam_ctx *ck = m->ctx[k];
    kslam_overlap *fo = m->rows_out.as<kslam_overlap>();
    uint32_t *fp = m->pool_out.as<uint32_t>();
    const uint64_t n1 = cnt[k].n_rows_r1, n2 = cnt[k].n_rows - n1, c1 = cnt[k].n_cigar_r1, c2 = cnt[k].n_cigar - c1;
    const uint64_t nl = sh[k].pair_hi - sh[k].pair_lo;
    if (k == 0)
      return kslam_export_shard_device(ck, nl, sh[k].pair_lo, n_units, op1[k], op2[k], fo + row1[k], fo + row2[k],
                                       fp + op1[k], fp + op2[k]);
    DevBuf &sb = m->send[k];
    kslam_status s1 = guarded(ck, [&] { sb.ensure((n1 + n2 + 1) * sizeof(kslam_overlap) + (c1 + c2 + 1) * sizeof(uint32_t)); });
    if (s1 != KSLAM_OK) return s1;
    kslam_overlap *so = sb.as<kslam_overlap>();
    uint32_t *sp = reinterpret_cast<uint32_t *>(so + n1 + n2);
    s1 = kslam_export_shard_device(ck, nl, sh[k].pair_lo, n_units, op1[k], op2[k], so, so + n1, sp, sp + c1);
    if (s1 != KSLAM_OK) return s1;
    return guarded(ck, [&] {
      if (n1) HIPCHK(hipMemcpyPeerAsync(fo + row1[k], c0->device, so, ck->device, n1 * sizeof(kslam_overlap), ck->stream));
      if (n2) HIPCHK(hipMemcpyPeerAsync(fo + row2[k], c0->device, so + n1, ck->device, n2 * sizeof(kslam_overlap), ck->stream));
      if (c1) HIPCHK(hipMemcpyPeerAsync(fp + op1[k], c0->device, sp, ck->device, c1 * sizeof(uint32_t), ck->stream));
      if (c2) HIPCHK(hipMemcpyPeerAsync(fp + op2[k], c0->device, sp + c1, ck->device, c2 * sizeof(uint32_t), ck->stream));
      HIPCHK(stream_wait(ck->stream));
    });
  });
  if (st != KSLAM_OK) return st;
  st = guarded(c0, [&] {
    ho = (kslam_overlap *)pinned_get(c0, (rows + 1) * sizeof(kslam_overlap));
    hc = (uint32_t *)pinned_get(c0, (ops + 1) * sizeof(uint32_t));
    if (rows) HIPCHK(hipMemcpyAsync(ho, m->rows_out.p, rows * sizeof(kslam_overlap), hipMemcpyDeviceToHost, c0->stream));
    if (ops) HIPCHK(hipMemcpyAsync(hc, m->pool_out.p, ops * sizeof(uint32_t), hipMemcpyDeviceToHost, c0->stream));
    HIPCHK(stream_wait(c0->stream));
  });
  if (st != KSLAM_OK) {
    if (ho) pinned_put(c0, ho);
    if (hc) pinned_put(c0, hc);
    return multi_fail(m, st, kslam_last_error(c0));
  }
  *out = ho; *n_out = rows; *cigar_pool = hc; *n_cigar = ops;
  return KSLAM_OK;
}

void kslam_multi_free_batch(kslam_multi *m, kslam_overlap *out, uint32_t *cigar_pool) {
  if (m && !m->ctx.empty()) kslam_free_batch(m->ctx[0], out, cigar_pool);
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
