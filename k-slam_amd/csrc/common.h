// common.h -- shared declarations of the HIP implementation (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <time.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/kslam.h"

namespace kslam {

constexpr int WAVE = 64;

struct HipError {
  hipError_t code;
  const char *what;
  const char *file;
  int line;
};

#define HIPCHK(expr)                                                      \
  do {                                                                    \
    hipError_t _e = (expr);                                               \
    if (_e != hipSuccess) throw ::kslam::HipError{_e, #expr, __FILE__, __LINE__}; \
  } while (0)

struct StatusError {
  kslam_status st;
  std::string msg;
};

// The KSLAM_* environment switches.  They are read ONCE per context, when
// kslam_create builds it (kslam_reload_tuning re-reads them: variant tests and tuning scripts), never on
// the batch path, and none of them changes a result.  The measurement-only ablations that DO change
// results (parts of kernels switched off) only exist in a -DKSLAM_ABLATE build (make ABLATE=1).
struct Tuning {
  bool debug = false;                 // KSLAM_DEBUG: per-stage counts on stderr
  bool sw_full = false;               // KSLAM_SW_FULL=1: every candidate on the full-matrix scoring kernel
  bool sw_no48 = false, sw_no96 = false;   // KSLAM_SW_NO48 / _NO96: drop a band tier
  int sw_unknown_nd = 0;              // KSLAM_SW_UNKNOWN_ND: where gapped candidates start (0: by read length)
  int cigar_sys_mask = 0xF8;          // KSLAM_CIGAR_SYS: band-width bins (cigar.hip: cig_bin) that run on the systolic kernel, one bit each
  bool cigar_reg = true;              // KSLAM_CIGAR_REG=0: no band-in-registers kernel
  bool cigar_dirs_lds = false;        // KSLAM_CIGAR_DIRS=lds: direction words in LDS
  bool cigar_tb_inline = false;       // KSLAM_CIGAR_TB=inline: systolic tracebacks at the end of the DP kernel
  int bucket_bits_max = 27;           // KSLAM_BUCKET_BITS
  int bucket_bits_exact = 0;          // KSLAM_BUCKET_BITS_EXACT (0: sized from the index)
  int filter_bits = -1;               // KSLAM_FILTER_BITS (-1: sized from the index, 0: no filter)
  int sort_bytes = -1;                // KSLAM_SORT_BYTES (-1: what the bucket table needs)
  bool sort_digit_bytes = true;       // KSLAM_SORT_DIGIT_BYTES=0: histograms re-read the records
  int lanes = 2;                      // KSLAM_LANES
  bool eager_cigar = false;           // KSLAM_EAGER_CIGAR
  bool lane_waits_yield = false;      // KSLAM_LANE_WAITS=yield: the pipeline lanes poll + sleep instead of busy-waiting for the GPU (stream_wait)
  bool pageable_columns = false;      // KSLAM_PAGEABLE_COLUMNS
  bool details_in_token = true;       // KSLAM_DETAILS_IN_TOKEN=0 (A/B): the per-row walk outside the lanes' compute token
  int plan_blocks_per_cu = 64;        // KSLAM_PLAN_BLOCKS: workgroups of k_sw_plan per CU (its waves walk through the candidates)
  int join_group_order = 1;           // KSLAM_JOIN_GROUP_ORDER=0: the overlap keys go through all their radix passes (join.hip: group_order)
  bool sw_sweep = true;               // KSLAM_SW_SWEEP=0: a read-back in front of every SW tier (as until round 5)
  bool sweep_room = true;             // KSLAM_SWEEP_ROOM=0 (tests): the CIGAR bins' and SW tiers' launches get NO room for what earlier ones send
                                      // on, so that every such candidate takes the left-over rounds
  int join_merge = 0;                 // KSLAM_JOIN=merge: k_join_merge instead of the probe k_join_fill (join.hip)
  bool filter_build_sorted = true;    // KSLAM_FILTER_BUILD=atomics: the membership filter by scattered atomics instead of block by block (filter.hip)
  int pseudo_cap = 0;                 // KSLAM_PSEUDO_CAP (tests): alignment pairs of one entry beyond which pseudo-assembly is left to the host; 0 = 262144
#ifdef KSLAM_ABLATE
  uint32_t sw_ablate = 0, cigar_variant = 0, filter_ablate = 0;   // KSLAM_SW_ABLATE / _CIGAR_VARIANT / _FILTER_ABLATE
#endif
};
Tuning read_tuning();   // kslam_api.hip

// grow-only device buffer.  Owns its block unless it was made a VIEW of another buffer with borrow() (a sibling context's
// index, share_index): a view never frees.  No copy-assignment: `a = b` between owners would free the block twice.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  bool borrowed = false;
  DevBuf() = default;
  ~DevBuf() { if (p && !borrowed) (void)hipFree(p); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap), borrowed(o.borrowed) { o.p = nullptr; o.cap = 0; o.borrowed = false; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) {
      release();
      p = o.p; cap = o.cap; borrowed = o.borrowed;
      o.p = nullptr; o.cap = 0; o.borrowed = false;
    }
    return *this;
  }
  // a non-owning view of `o`'s block (whatever this buffer owned before is freed)
  void borrow(const DevBuf &o) {
    release();
    p = o.p; cap = o.cap; borrowed = true;
  }
  void ensure(size_t bytes) {
    if (bytes <= cap && !borrowed) return;      // (a view is never written into as if it were ours: it becomes an owner)
    release();
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      p = nullptr;
      (void)hipGetLastError();
      throw StatusError{KSLAM_ERR_OOM, "hipMalloc of " + std::to_string(want) + " bytes failed"};
    }
    cap = want;
  }
  void release() {
    if (p && !borrowed) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    borrowed = false;
  }
  template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// ---- base coding shared by the SW and cigar kernels (branch-free) ---------------------------
#ifdef __HIPCC__
// kBaseTranslation (reference src/ssw_cpp.cpp:11-23): A/a 0, C/c 1, G/g 2, T/t 3, U/u 0, else 4
__device__ inline uint32_t ssw_code(uint32_t c) {
  const uint32_t idx = c & 31u;                                        // A 1, C 3, G 7, T 20, U 21
  const bool known = ((c & 0xC0u) == 0x40u) && ((0x0030008Au >> idx) & 1u);
  const uint32_t code = (uint32_t)(((1ull << 6) | (2ull << 14) | (3ull << 40)) >> (2u * idx)) & 3u;
  return known ? code : 4u;
}
// the same after inPlaceReverseComplement's per-base step (reference src/sequenceTools.h:98-116):
// only UPPER-case A/C/G/T are complemented, every other character is left as it is
__device__ inline uint32_t ssw_code_complemented(uint32_t c) {
  const uint32_t idx = c & 31u;
  const uint32_t code = ssw_code(c);
  const bool upper_acgt = ((c & 0xE0u) == 0x40u) && ((0x0010008Au >> idx) & 1u);
  return upper_acgt ? 3u - code : code;
}
#endif

// ---------------------------------------------------------------- scan.hip
// exclusive scan of n u32 values; out may be u32 or u64. total (u64) is
// written to d_total[0].  tmp must hold scan_tmp_bytes(n).
size_t scan_tmp_bytes(uint64_t n);
void exclusive_scan_u32(const uint32_t *d_in, uint32_t *d_out, uint64_t n,
                        uint64_t *d_total, void *d_tmp, hipStream_t s);
void exclusive_scan_u32_to_u64(const uint32_t *d_in, uint64_t *d_out, uint64_t n,
                               uint64_t *d_total, void *d_tmp, hipStream_t s);

// ------------------------------------------------------------- extract.hip
struct SegEntry {   // one wave-sized unit of extraction work
  uint32_t seq;     // sequence index (read id / entry id)
  uint32_t q0;      // first k-mer index of the segment inside the sequence
  uint64_t out;     // record index of k-mer q0 in the output
};
constexpr uint32_t SEG_KMERS = 128;  // k-mers per segment
constexpr uint32_t MAX_GAP = 64;

struct ExtractPlan {
  uint64_t n_seqs = 0;
  uint64_t n_kmers = 0;
  uint64_t n_segs = 0;
};
// Sizes the work (device scans) and fills the segment table.
// d_nk / d_nseg: u32[n] scratch; d_rec_start: u64[n]; d_seg_start: u64[n]
void extract_plan(const uint64_t *d_offsets, uint64_t n_seqs, uint32_t gap,
                  uint32_t *d_nk, uint32_t *d_nseg, uint64_t *d_rec_start,
                  uint64_t *d_seg_start, uint64_t *d_totals /*[2]*/, void *d_scan_tmp,
                  hipStream_t s);
void extract_fill_segments(const uint32_t *d_nk, const uint64_t *d_rec_start,
                           const uint64_t *d_seg_start, uint64_t n_seqs, uint32_t gap,
                           SegEntry *d_segs, hipStream_t s, uint64_t n_segs = ~0ull);
// AoS output (reference record layout); id_base is added to the sequence index.  d_digits != nullptr: also the digit of
// *first_pass of every record, one byte at the record's index (the first histogram of the sort that follows reads those
// instead of the records: radix_sort.hip, SortWorkspace::first_digits_ready)
struct SortPass;
void extract_kmers_launch(const uint8_t *d_bases, const uint64_t *d_offsets,
                          const SegEntry *d_segs, uint64_t n_segs, uint32_t gap,
                          int is_gb, uint32_t id_base, uint4 *d_out, hipStream_t s,
                          uint8_t *d_digits = nullptr, const SortPass *first_pass = nullptr);

// -------------------------------------------------------------- filter.hip
// Blocked Bloom filter over the genome k-mer set (2^log2_bits bits, 128-byte lines chosen by the
// k-mer's minimizer) and the read extraction kernel that keeps only the k-mers the filter lets through.
size_t filter_bytes(uint32_t log2_bits);
void filter_build(const uint64_t *d_sorted_keys, uint32_t n, uint32_t log2_bits, void *d_filter, hipStream_t s);
// the same filter, built from the keys' probe words ordered by 32 KB filter block, each block assembled in LDS (filter.hip);
// d_words_a / _b: n + 1 u64 each; d_block_start: (filter bytes / 32 KB) + 2 u32
struct SortWorkspace;
void filter_build_sorted(const uint64_t *d_sorted_keys, uint32_t n, uint32_t log2_bits, void *d_filter, void *d_words_a, void *d_words_b,
                         uint32_t *d_block_start, SortWorkspace &ws, hipStream_t s, bool words_ready = false);
// One pass over the sorted genome records for all of: the key column, the {meta, offset} column, the bucket table (2^bucket_bits + 1
// lower bounds) and the filter's probe words in d_words (+ their first sort digit in ws.digits) -- then filter_build_sorted(...,
// words_ready = true).  false: not applicable (empty list, filter smaller than a block), nothing was done.
bool split_columns_and_tables(const void *d_sorted_recs, uint32_t n, uint64_t *d_key, void *d_meta_off, uint32_t bucket_bits, uint32_t *d_bucket,
                              uint32_t log2_bits, void *d_words, SortWorkspace &ws, hipStream_t s);
// reads d_off[0..n_reads] (gap 1, ids = position in d_off): surviving records appended at *d_cursor
// (which ends as their number); nothing is written beyond `cap` (the caller reruns with a larger buffer)
// d_digits != nullptr: also byte digit_word / digit_shift of every record written (the first radix pass's digit), at
// the record's index
void extract_filtered(const uint8_t *d_bases, const uint64_t *d_off, uint32_t n_reads, const void *d_filter,
                      uint32_t log2_bits, uint4 *d_out, uint64_t *d_cursor, uint64_t cap, const Tuning &tune, hipStream_t s,
                      uint8_t *d_digits = nullptr, uint32_t digit_word = 0, uint32_t digit_shift = 0);

// ---------------------------------------------------------- radix_sort.hip
struct SortPass {
  uint32_t word;    // which 32-bit word of the record holds the digit
  uint32_t shift;   // bit shift inside the word
  uint32_t invert;  // XOR mask applied to the word first (descending keys)
  // a digit made of TWO bit fields of the word (hi_bits > 0): bits [shift, shift + 8 - hi_bits) below bits
  // [hi_shift, hi_shift + hi_bits) -- the meta word's high id bits and its revComp bit in one pass (kslam_api.hip: build_index)
  uint32_t hi_shift = 0, hi_bits = 0;
};
#ifdef __HIPCC__
__host__ __device__ inline uint32_t sort_pass_digit(uint32_t word_value, const SortPass &p) {
  const uint32_t v = word_value ^ p.invert;
  if (p.hi_bits == 0) return (v >> p.shift) & 0xFFu;
  const uint32_t lo_bits = 8u - p.hi_bits;
  return ((v >> p.shift) & ((1u << lo_bits) - 1u)) | (((v >> p.hi_shift) & ((1u << p.hi_bits) - 1u)) << lo_bits);
}
#endif
// Waiting for a stream.  hipStreamSynchronize spins on the completion signal: the fastest wake-up, and what a caller
// that has nothing else to do wants (kslam_align_batch, the resident bench).  A pipeline lane waits while the host
// stage of an earlier batch needs every CPU the process may use (a 16-CPU cgroup quota on the bench boxes), and its
// ~40 waits per batch add up to most of the batch's GPU time: measured with tools/cpu_sampler.c, the two lanes spent
// 1.1 CPUs inside the runtime's busy-wait loop -- with hipEventBlockingSync events too, which this runtime spins on
// just the same.  With KSLAM_LANE_WAITS=yield a lane thread sets `yield`, and its waits poll an event with
// hipEventQuery: at once, after a few short spins, then between sleeps of 20-80 microseconds (lane CPU 1.3 -> 0.2 s
// per 20 batches; the GPU-bound pipelined ABI path loses 1 ms per batch to the later wake-ups, and on the bench box the
// end-to-end loop did not get faster for the freed CPU, so the default stays the spin).
struct WaitMode {
  bool yield = false;
  hipEvent_t ev = nullptr;
  ~WaitMode() { if (ev) (void)hipEventDestroy(ev); }
};
inline WaitMode &wait_mode() {
  static thread_local WaitMode w;
  return w;
}
inline hipError_t stream_wait(hipStream_t s) {
  WaitMode &w = wait_mode();
  if (!w.yield) return hipStreamSynchronize(s);
  hipError_t e = hipSuccess;
  if (!w.ev) e = hipEventCreateWithFlags(&w.ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventRecord(w.ev, s);
  if (e != hipSuccess) return e;
  for (int spin = 0;; spin++) {
    e = hipEventQuery(w.ev);
    if (e != hipErrorNotReady) return e;
    if (spin < 8) continue;
    const long us = spin < 40 ? 20 : 80;
    struct timespec ts = {0, us * 1000};
    nanosleep(&ts, nullptr);
  }
}

// Device -> host copy of a few bytes the host has to look at before it can launch the next kernel
// (list sizes, counts).  Through a small page-locked buffer: a copy into pageable memory goes through
// the runtime's staging path and costs tens of microseconds more per look, with the GPU idle.
inline void read_back(void *dst, const void *d_src, size_t bytes, hipStream_t s) {
  static thread_local void *pinned = nullptr;
  constexpr size_t CAP = 256;
  if (bytes > CAP) throw StatusError{KSLAM_ERR_INTERNAL, "read_back: more than 256 bytes"};
  if (!pinned) HIPCHK(hipHostMalloc(&pinned, CAP, hipHostMallocDefault));
  HIPCHK(hipMemcpyAsync(pinned, d_src, bytes, hipMemcpyDeviceToHost, s));
  HIPCHK(stream_wait(s));
  memcpy(dst, pinned, bytes);
}

constexpr int SORT_TILE = 4096;
struct SortWorkspace {
  DevBuf hist;      // u32 [chunks][256] per-chunk digit totals -> bases
  DevBuf status;    // u32 [tiles][256] per-tile digit counts -> in-chunk prefixes
  DevBuf tickets;   // u32 [256] per-digit totals -> global bin bases
  DevBuf digits;    // u8 [n] the next pass's digit of every record, written by the scatter (radix_sort.hip)
  bool use_digit_bytes = true;   // KSLAM_SORT_DIGIT_BYTES=0: every histogram re-reads the records (A/B)
  bool first_digits_ready = false;   // `digits` already holds the FIRST pass's digit of every record (the producer of the
                                     // records wrote them: extract_filtered); the caller sets and clears it around one sort
  bool meta_digits_in_runs = false;  // set around the one-time sort of genome records when entries hold many k-mers each: the meta word's
                                     // digits then come in long runs of few values, and their histograms take the kernel made for that
  hipEvent_t *ev_sc0 = nullptr, *ev_sc1 = nullptr;   // optional per-pass events around k_scatter
  uint32_t epoch = 0;
};
// Sorts n records of REC_WORDS 32-bit words (4 = k-mer record, 2 = u64 key) by
// the given passes (LSD order: passes[0] is the least significant digit).
// Result ends in `a` if the number of passes is even, else in `b`; returns the
// pointer holding the sorted data.  ev0/ev1 bracket the scatter passes.
void *radix_sort(void *a, void *b, uint64_t n, int rec_words, const SortPass *passes,
                 int n_passes, SortWorkspace &ws, hipStream_t s, hipEvent_t ev_scatter0,
                 hipEvent_t ev_scatter1, uint32_t *n_launches, bool setup = false);   // setup: one-time sort (kernel names *_setup)

// ---------------------------------------------------------------- join.hip
struct GenomeIndexDev {
  const uint64_t *key;   // sorted genome k-mers
  const uint2 *mo;       // {ID_isFromGB_RC, offset} of the key at the same position: ONE 8-byte gather per hit (two 4-byte
                         // columns cost two cache lines per hit; the join is bound by the lines it touches, not by ALU work)
  const uint32_t *bucket;  // [2^bits + 1] lower bounds by top `bits` bits of the key
  uint32_t bucket_bits;
  uint32_t n;
};
struct OverlapKeyLayout {  // packed u64 overlap: read | entry | rel + bias | revcomp
  uint32_t bits_read, bits_entry, bits_rel;
  uint32_t rel_bias;
};
constexpr int JOIN_TILE = 1024;
void build_bucket_table(const uint64_t *d_keys, uint32_t n, uint32_t bits, uint32_t *d_bucket,
                        hipStream_t s);
// lower bounds of the nb values `key >> shift` takes (keys ordered by it): d_table[0 .. nb]
void build_offsets_table(const uint64_t *d_keys, uint32_t n, uint32_t shift, uint32_t nb, uint32_t *d_table, hipStream_t s);
// single-pass join: output ranges reserved with one atomic per workgroup; *d_cursor ends as the
// total number of overlaps; nothing beyond `cap` is written (caller reruns with a larger buffer)
void join_fill_single_pass(const uint4 *d_read_recs, uint32_t n_r, GenomeIndexDev g, const uint32_t *d_read_len,
                           uint64_t *d_cursor, uint64_t cap, OverlapKeyLayout lay, uint64_t *d_out, hipStream_t s);
// the same contract by a merge (join.hip: k_join_merge): the read records must be ordered by their top `sorted_top_bits` key bits
void join_fill_merge(const uint4 *d_read_recs, uint32_t n_r, GenomeIndexDev g, uint32_t sorted_top_bits, const uint32_t *d_read_len,
                     uint64_t *d_cursor, uint64_t cap, OverlapKeyLayout lay, uint64_t *d_out, hipStream_t s);
// overlap keys radix-sorted by the bits above rel / revComp only: finish every (read, entry) group -- order its few keys by the
// low bits, apply std::unique's "within 3 of the last kept" (Overlap.h:79-85) -- writing ordered keys and flags (join.hip).
// *d_big is set when a group holds more than 64 keys (a read in a tandem repeat): d_out / d_flags are then incomplete and the
// caller sorts the chunk the long way from d_keys, which is only read here.
void group_order(const uint64_t *d_keys, uint64_t n, OverlapKeyLayout lay, uint64_t *d_out, uint32_t *d_flags, uint32_t *d_big, hipStream_t s);
void dedupe_flags(const uint64_t *d_keys, uint64_t n, OverlapKeyLayout lay, uint32_t *d_flags,
                  hipStream_t s);
void dedupe_compact(const uint64_t *d_keys, const uint32_t *d_flags, const uint32_t *d_pos,
                    uint64_t n, OverlapKeyLayout lay, uint32_t read_id_base, kslam_overlap *d_out,
                    hipStream_t s);

// --------------------------------------------------------------- merge.hip
constexpr uint32_t MERGE_MAX_SHARDS = 256;
struct MergeShard {            // one read shard of a batch, as the collecting device sees it
  uint64_t pair_lo, pair_hi;   // the batch's pairs [lo, hi) this shard aligned (local ids: R1 block | R2 block)
  uint64_t row_base, n_rows;   // its rows inside the gathered row array
  uint64_t pool_base;          // its CIGAR pool inside the gathered pools
  uint64_t n_r1, out_r1, out_r2;   // filled on the device: rows of its R1 block, output positions of its two blocks
};
// d_shards: n_shards entries with the host-known fields set.  d_out / d_pool_out receive the batch-global
// result in the reference's order with the pool in row order; *d_total ends as the number of CIGAR ops.
void merge_shards(const kslam_overlap *d_rows, uint64_t n_rows, const uint32_t *d_pool_in, MergeShard *d_shards,
                  uint32_t n_shards, uint64_t n_pairs, kslam_overlap *d_out, uint32_t *d_pool_out, uint32_t *d_lens,
                  uint64_t *d_new_off, uint64_t *d_total, void *d_scan_tmp, hipStream_t s);

// d_out2[0] = rows of the R1 block, d_out2[1] = CIGAR words of those rows
void shard_counts(const kslam_overlap *d_rows, uint64_t n, uint32_t n_local_pairs, uint64_t n_cigar, uint64_t *d_out2,
                  hipStream_t s);
void export_rows(const kslam_overlap *d_rows, uint64_t n, uint64_t n_r1, uint32_t n_local_pairs, uint64_t pair_lo,
                 uint64_t n_pairs_total, uint64_t n_cigar_r1, uint64_t pool_base_r1, uint64_t pool_base_r2,
                 kslam_overlap *d_out_r1, kslam_overlap *d_out_r2, hipStream_t s);

// ------------------------------------------------------------------ sw.hip
// ---- 48-byte records, a wave at a time ----------------------------------------------------------------------------
// kslam_overlap is 48 bytes and a thread that loads "its" record issues three 16-byte loads at a 48-byte stride: every
// instruction of the wave touches 64 different pieces spread over 3 KB.  These helpers move the 64 records of a wave
// (3072 contiguous bytes) with three fully coalesced 16-byte accesses per lane and transpose them through 3 KB of LDS
// per wave.  All 64 lanes must call; records at or beyond n are not read / written (their lanes get zeros).
static_assert(sizeof(kslam_overlap) == 48, "the record helpers move 48-byte records");
__device__ inline kslam_overlap wave_load_records(const kslam_overlap *rows, uint64_t first, uint64_t n, uint4 *lds_wave) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint4 *src = reinterpret_cast<const uint4 *>(rows + first);
  const uint64_t pieces = first < n ? (n - first < 64 ? (n - first) * 3 : 192) : 0;   // 16-byte pieces that exist
#pragma unroll
  for (uint32_t k = 0; k < 3; k++) {
    const uint32_t x = lane + 64u * k;
    lds_wave[x] = x < pieces ? src[x] : make_uint4(0u, 0u, 0u, 0u);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  union { uint4 q[3]; kslam_overlap o; } u;
#pragma unroll
  for (uint32_t k = 0; k < 3; k++) u.q[k] = lds_wave[3u * lane + k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return u.o;
}
__device__ inline void wave_store_records(kslam_overlap *rows, uint64_t first, uint64_t n, const kslam_overlap &mine, uint4 *lds_wave) {
  const uint32_t lane = threadIdx.x & 63u;
  union { uint4 q[3]; kslam_overlap o; } u;
  u.o = mine;
#pragma unroll
  for (uint32_t k = 0; k < 3; k++) lds_wave[3u * lane + k] = u.q[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  uint4 *dst = reinterpret_cast<uint4 *>(rows + first);
  const uint64_t pieces = first < n ? (n - first < 64 ? (n - first) * 3 : 192) : 0;
#pragma unroll
  for (uint32_t k = 0; k < 3; k++) {
    const uint32_t x = lane + 64u * k;
    if (x < pieces) dst[x] = lds_wave[x];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the kept records of a wave into consecutive slots: lane `keep`s record `mine` for slot base + rank (rank = number of
// kept lanes below it); `count` = kept lanes of the wave.  All 64 lanes must call.
__device__ inline void wave_store_records_compact(kslam_overlap *rows, uint64_t base, uint32_t count, bool keep, uint32_t rank,
                                                  const kslam_overlap &mine, uint4 *lds_wave) {
  const uint32_t lane = threadIdx.x & 63u;
  union { uint4 q[3]; kslam_overlap o; } u;
  u.o = mine;
  if (keep) {
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) lds_wave[3u * rank + k] = u.q[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  uint4 *dst = reinterpret_cast<uint4 *>(rows + base);
#pragma unroll
  for (uint32_t k = 0; k < 3; k++) {
    const uint32_t x = lane + 64u * k;
    if (x < 3u * count) dst[x] = lds_wave[x];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct SwParams {
  int32_t match, mismatch, gap_open, gap_extend;
  uint32_t score_threshold;
  int32_t report_cigar;
  int32_t striped = 0;   // scoring outside the envelope: every candidate through k_sw_striped + the literal banded_sw
#ifdef KSLAM_ABLATE
  uint32_t ablate = 0;   // KSLAM_SW_ABLATE (measurement only): 1 = no sweep, 2 = no staging either, 3 = one turn, result accepted
#else
  static constexpr uint32_t ablate = 0;   // the ablations are compiled out of the product build
#endif
};
struct SwInputs {
  const uint8_t *read_bases;
  const uint64_t *read_off;   // [n_reads + 1]
  const uint8_t *genome_bases;
  const uint64_t *genome_off; // [n_entries + 1]
  // the same two arrays as base codes (encode_bases): what the SW kernels stage from
  const uint8_t *read_codes = nullptr;
  const uint8_t *genome_codes = nullptr;
};
// One byte per base: bits 0..2 = SSW code 0..4 (ssw_code), bit 3 = "upper-case A/C/G/T", i.e. the
// bases inPlaceReverseComplement complements (ssw_code_complemented(c) = bit 3 ? 3 - code : code).
// Done once per index and once per read batch so that no SW kernel decodes ASCII again.
void encode_bases(const uint8_t *d_src, uint8_t *d_dst, uint64_t n, hipStream_t s);
// forward + reverse passes for n candidates (in place on d_ov: window-relative,
// unflipped coordinates); d_band0[i] = initial band width for banded_sw
// (0 = no cigar wanted, ssw.c:924-927)
struct SwWork {
  DevBuf flags, pos, list, list2, scan_tmp, totals, tier_list[6];
  // what the last chunk's tiers received from the tiers before them (sw.hip: the one sweep over the tiers is sized from it)
  uint64_t last_n = 0;
  uint32_t last_inflow[6] = {0, 0, 0, 0, 0, 0};
  int last_tiers = 0, last_lm = -1;
};
// *n_full_out: candidates that needed the full-matrix kernel (the rest ran in a proven band)
void sw_scores(kslam_overlap *d_ov, uint64_t n, SwInputs in, SwParams p, uint32_t max_read_len,
               uint32_t *d_band0, SwWork &W, uint64_t *n_full_out, const Tuning &tune, hipStream_t s,
               bool long_reads = false);   // long_reads: the chunk's reads exceed the packed kernels (sw.hip: k_sw_long)

// Stable 8-way partition of the element numbers 0..n-1 by d_bins[i] (bins >= 8 are left out):
// d_lists[k] receives bin k's numbers in order, d_counts[k] its size (sw.hip).
void partition_bins(const uint8_t *d_bins, uint64_t n, uint32_t *const d_lists[8], uint32_t *d_counts, DevBuf &pos,
                    hipStream_t s);

// --------------------------------------------------------------- cigar.hip
struct CigarWork {
  DevBuf flags, pos, list, bmax, needbig, scan_tmp, totals, cig_off, tmp, tmp_big, big_pos, scratch, cls, cls_list[8],
      special, counters, tb_list;
};
constexpr uint32_t CIG_CAP = 24;  // ops per small temp cigar slot
// allocates and clears the per-candidate cigar state; call before sw_scores
void cigar_prepare(CigarWork &W, uint64_t n, hipStream_t s);
// the records as cigar_finalize will leave them, cigars aside, into d_out (for the pairing between SW and the cigar stage);
// and: no cigar for the rows whose d_referenced flag is 0
void final_coords_copy(const kslam_overlap *d_ov, uint64_t n, SwInputs in, kslam_overlap *d_out, hipStream_t s);
void drop_unreferenced_cigars(kslam_overlap *d_ov, uint32_t *d_bw, const uint32_t *d_referenced, uint64_t n, hipStream_t s);
// banded DP + traceback into temp slots; returns the total number of cigar ops
// and the number of "Trace back error" cases (reference would abort there)
void cigar_traceback(kslam_overlap *d_ov, uint64_t n, SwInputs in, SwParams p, uint32_t lmax, uint32_t *d_bw,
                     CigarWork &W, uint64_t *n_cigar_out, uint32_t *n_tb_err, const Tuning &tune, hipStream_t s);
// un-flip + absolute coordinates, cigar gather into pool[pool_base ...)
void cigar_finalize(kslam_overlap *d_ov, uint64_t n, SwInputs in, uint32_t lmax, CigarWork &W,
                    const uint32_t *d_bw, uint32_t *d_pool, uint64_t pool_base, uint64_t *d_cells, hipStream_t s);

// ------------------------------------------------------------- details.hip
struct DetailWork {
  DevBuf lens, off, slots, scan_tmp, totals, md_pool;
};
// NM / log-probability / MD text of every row (see details.hip); d_tables = matchTable[100] then
// misMatchTable[100] of reference src/SAM.h:33-48, computed by the host
void row_details(const kslam_overlap *d_ov, uint64_t n, const uint32_t *d_pool, const uint8_t *d_rbases,
                 const uint8_t *d_rqual, const uint64_t *d_roff, const uint8_t *d_gbases, const uint64_t *d_goff,
                 const double *d_tables, kslam_row_detail *d_out, DetailWork &W, uint8_t **d_md_pool_out,
                 uint64_t *n_md_out, uint32_t *flags_out, hipStream_t s, const uint32_t *d_rows = nullptr,
                 uint64_t n_list = 0);   // d_rows: only these rows are walked, the others get zero records

// --------------------------------------------------------------- pairs.hip
struct PairWork {
  DevBuf recs, count, base, inserts, flags, gpos, rpos, scan_tmp, totals, groups, dense, sort_a, sort_b, idx, picked, row_list, row_start, gaps;
  DevBuf route_a, route_b, route_heads, route_counts, route_scores;   // pseudo_route / pseudo_owned (entries partitioned over ranks)
  const void *route_sorted = nullptr;   // {destination, record} of this rank's records in sending order (route_a or route_b)
  uint64_t route_n = ~0ull;             // records routed by the last pseudo_route, ~0 when none is outstanding
  uint32_t route_world = 0;
  uint64_t units = 0, mid = 0;   // between pair_phase_a and pair_phase_b: read pairs (or reads) of the batch, its R1 block
  uint32_t pseudo_cap = 0;       // 0 = the default (pairs.hip: PSEUDO_CAP_GLOBAL); tests lower it to reach the host fallback
  int paired = 0;
};
struct PairResult {
  uint64_t n_overlaps_screened, n_paired_initial, n_insert_sizes, n_read_pairs, n_pairs;
  uint32_t max_insert_size;
  uint32_t stages_done;                  // KSLAM_TAIL_* bits of the stages the device ran
  const kslam_read_pair *d_groups;       // n_read_pairs, `first` indexes d_pairs
  const kslam_paired_overlap *d_pairs;   // n_pairs, dense
};
// score screen + pairing + insert-size limit + the two per-read-pair screens (see pairs.hip); d_ov in
// alignToDatabase order, d_read_len[read]; blocks on the stream (small read-backs between the kernels)
void pair_and_screen(const kslam_overlap *d_ov, uint64_t n, const uint32_t *d_read_len, uint64_t n_reads, int paired,
                     uint32_t score_threshold, double score_fraction, int do_insert, int do_score, PairWork &W,
                     SortWorkspace &sortws, PairResult *res, hipStream_t s);

// --------------------------------------------------------- fastq_index.hip
struct FastqWork {
  DevBuf tile_count, tile_base, scan_tmp, totals, ev[2], bases_at, quality_at, blen, id_at, id_len, bases_off, ids_off, ids;
};
struct FastqIndexResult {
  uint64_t n_reads, bases_total, ids_total;
  uint64_t consumed[2];
  const uint64_t *d_bases_at, *d_quality_at;   // n_reads: positions in [r1 | r2]
  const uint64_t *d_bases_off, *d_ids_off;     // n_reads + 1
  const uint8_t *d_ids;                        // ids_total bytes
};
// d_text = [r1 | r2] on the device; h_last1 / h_last2: the streams' last bytes on the host (or nullptr for
// an empty stream).  Semantics of host/fastq.cpp's index (kslam_fastq_index_pair).
void fastq_index_device(const uint8_t *d_text, uint64_t len1, uint64_t len2, const uint8_t *h_last1, const uint8_t *h_last2,
                        uint64_t max_pairs, bool at_eof, FastqWork &W, FastqIndexResult *res, hipStream_t s, bool single = false);

// pair_and_screen in its two halves, for read pairs sharded over several GPUs: phase A pairs per
// read pair and collects the shard's insert sizes (W.inserts, res->n_insert_sizes); the caller gathers every shard's
// insert sizes and computes the batch's limit from all of them (insert_limit_device: the statistic is batch-global,
// src/PairedOverlap.h:314-360); phase B screens with that limit.  pseudo_merged: pseudo-assembly on the dense
// alignment-pair records of ALL shards (d_all, gathered in rank order; this shard's are [own_base, own_base + n_pairs)),
// new scores copied back into this shard's records, second screen on them.
void pair_phase_a(const kslam_overlap *d_ov, uint64_t n, const uint32_t *d_read_len, uint64_t n_reads, int paired,
                  uint32_t score_threshold, PairWork &W, PairResult *res, hipStream_t s);
uint32_t insert_limit_device(const int32_t *d_ins, uint64_t n, PairWork &W, SortWorkspace &sortws, hipStream_t s);
void pair_phase_b(const kslam_overlap *d_ov, uint32_t limit, double score_fraction, int do_insert, int do_score, PairWork &W,
                  PairResult *res, hipStream_t s);
bool pseudo_merged(PairWork &W, PairResult *res, void *d_all, uint64_t n_all, uint64_t own_base, double score_fraction,
                   SortWorkspace &sortws, hipStream_t s);

// the overlap records the result's alignment pairs refer to, ascending, as a list in W (valid until the next pairs call)
void referenced_rows(PairWork &W, const PairResult *res, uint64_t n_rows, const uint32_t **d_list, uint64_t *n_list, hipStream_t s);

// pseudoAssembly + the second score screen on pair_and_screen's result, in place; false (nothing changed)
// when an entry has more spans than a workgroup's LDS holds: the host then runs that stage itself
bool pseudo_and_rescreen(PairWork &W, PairResult *res, double score_fraction, SortWorkspace &sortws, hipStream_t s);
// the same stage with the ENTRIES partitioned over the ranks of a sharded batch (pairs.hip, bottom): route -> all-to-all ->
// owned -> all-to-all back -> return
void pseudo_route(PairWork &W, const PairResult *res, uint32_t world, const void **d_heads, uint64_t *counts, SortWorkspace &sortws,
                  hipStream_t s);
bool pseudo_owned(PairWork &W, void *d_heads, uint64_t n, const uint32_t **d_scores, SortWorkspace &sortws, hipStream_t s);
void pseudo_return(PairWork &W, PairResult *res, const uint32_t *d_scores, uint64_t n, double score_fraction, hipStream_t s);

// test hook for wave_gnu_sort.h (kslam_debug_wave_sort)
void debug_wave_sort(const int32_t *keys, const uint64_t *seg_off, uint64_t n_seg, uint32_t *perm, hipStream_t s);

// bases / quality columns cut out of FASTQ text on the device: read i = text[bases_at[i] ..) and
// text[quality_at[i] ..), d_off[i + 1] - d_off[i] bytes each, to d_bases / d_quality + d_off[i]
void gather_fields(const uint8_t *d_text, const uint64_t *d_bases_at, const uint64_t *d_quality_at, const uint64_t *d_off,
                   uint64_t n_reads, uint8_t *d_bases, uint8_t *d_quality, hipStream_t s);

}  // namespace kslam
