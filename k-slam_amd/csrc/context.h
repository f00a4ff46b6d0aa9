// context.h -- what the translation units behind the C ABI (include/kslam.h) share: the context itself and the internal
// helpers that cross file boundaries.  The ABI's host side was one file of 2 650 lines until round 6 (kslam_api.hip); it is now split
// by phase:
//   api_core.hip   context life cycle, tuning switches, page-locked pools, loading reads, fetching results, the operator entry
//   api_index.hip  kslam_set_index: extraction + the one-time sort + tables (build_index), and the stage-level entry points
//   api_align.hip  alignToDatabase on the resident batch (the chunk loop: extract -> sort -> join -> dedupe -> SW -> CIGAR)
//   api_tail.hip   the widened path on the device: qualities, per-row details, pairing / screens / pseudo-assembly, SAM text
//   api_lanes.hip  the pipelined entry (worker lanes, FASTQ text in)
//   api_multi.hip  several devices: shard export / merge, kslam_multi_*
#pragma once
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

namespace kslam_api {

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

// ---- api_core.hip
uint32_t bits_for(uint64_t max_value);
// grow a device buffer while keeping its first `used` bytes
void ensure_keep(DevBuf &b, size_t bytes, size_t used, hipStream_t s);
void *pinned_alloc(size_t bytes);
void pinned_free(void *p, size_t bytes);
// pinned host buffers from a small per-context pool (pinning is expensive; reuse across batches)
void *pinned_get(kslam_ctx *c, size_t bytes);
bool pinned_put(kslam_ctx *c, void *p);
bool scoring_in_envelope(const kslam_params &p);
// a sibling context sees the primary's index through the same device pointers
void share_index(kslam_ctx *dst, const kslam_ctx *src);

// ---- api_index.hip
// extraction of n sequences d_off[0..n] into d_out (AoS records) [+ the first radix pass's digit of every record]
void run_extract(kslam_ctx *c, const uint8_t *d_bases, const uint64_t *d_off, uint64_t n, uint32_t gap, int is_gb,
                 uint64_t n_segs, uint4 *d_out, uint8_t *d_digits = nullptr, const SortPass *first_pass = nullptr);

// ---- api_align.hip
void finish_load_reads(kslam_ctx *c);
float ev_ms(hipEvent_t a, hipEvent_t b);
struct PairingHook {
  int paired;
  uint32_t thr;
  double fraction;
  uint32_t stages;
  bool ran = false;    // false: the batch had several chunks (or nothing to pair): the caller pairs afterwards
};
// the hot path on the resident reads; stop_after_join: only rows a-3..a-6
void align_resident(kslam_ctx *c, bool stop_after_join, uint64_t *n_raw_out, PairingHook *hook = nullptr);

// ---- api_tail.hip

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

void sam_stage_free(kslam_ctx *c, SamStage &S);
void sam_stage_plan(kslam_ctx *c, const kslam_ctx *owner, int paired, uint32_t num_alignments, int sam_xa, bool sort_groups, SamStage &S);
void sam_stage_mapq(SamStage &S);
void sam_stage_kernels(kslam_ctx *c, const kslam_ctx *owner, SamStage &S, bool want_sam, bool want_per_read);

void sam_stage_fetch(kslam_ctx *c, SamStage &S, bool want_sam, bool want_per_read, char **sam_text, uint64_t *sam_len, char **pr_text,
                     uint64_t *pr_len, uint32_t **tax, uint64_t *n_tax);
void fill_pair_stats(const PairResult &r, kslam_pair_stats *st);

// ---- api_lanes.hip
void stop_lanes(kslam_ctx *c);

}  // namespace kslam_api

using namespace kslam_api;
