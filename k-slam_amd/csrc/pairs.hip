// pairs.hip -- the host tail up to the SAM formatter on the device: score screen, read pairing, insert-size
// statistics, insert-size screen, score-fraction screen, pseudo-assembly and the second score screen
// (SURVEY.md section 8f rows N1 / N4), straight from the overlap records the hot path left in HBM.
//
// Replaces, in the reference:
//   screenOverlapsByScoreThreshold                      src/Overlap.h:329-341
//   getPairedOverlaps / getPairsFromRead / makePair     src/PairedOverlap.h:107-270
//   getPerReadOverlaps (paired and single end)          src/PairedOverlap.h:437-470, src/Overlap.h:303-327
//   getDummyAlignmentPairsFromSingleEndReads            src/PairedOverlap.h:280-298
//   getMaxAllowedInsertSize                             src/PairedOverlap.h:314-360
//   screenPairedAlignmentsByInsertSize(replace = true)  src/PairedOverlap.h:396-436
//   screenPairedAlignmentsByScore                       src/PairedOverlap.h:361-390
//   pseudoAssembly (+ the screen by score once more)    src/PairedOverlap.h:480-582, src/SLAM.h:220-227
// and follows k-slam_amd/host/tail.cpp (pair_stage, max_allowed_insert, screen_stage) decision for
// decision, including the points where the reference leaves the result to an unstable sort: the pairing
// "sort" is a merge of the R1 and R2 rows with ties in input order (R1 first), and the two per-read-pair
// std::sort calls are reproduced with the permutation libstdc++ produces (gnu_sort.h).
//
// MI355X: one thread per read pair.  A pair's rows are two short runs of the overlap array (found through
// a first-row-per-read table, k_row_starts), its alignment pairs go to a private region of 4 x (its rows) records -- pairing emits
// at most 2 per row, the insert-size screen at most doubles that -- so nothing is counted twice and no
// thread waits for another; the survivors are compacted with two scans.  The insert sizes are appended
// with one atomic per workgroup and sorted with the library's radix sort; quartiles and the percentile ladder
// are read from the sorted array by index, the sums are exact 64-bit integers (and fall back to the
// reference's sequential double accumulation on the host when they could exceed 2^53).
#include <cmath>
#include <cstddef>
#include <type_traits>

#include "common.h"
#include "gnu_sort.h"
#include "wave_gnu_sort.h"

namespace kslam {

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;   // KSLAM_NO_OVERLAP
using Rec = kslam_paired_overlap;

__device__ inline Rec single_rec(const kslam_overlap &o, uint32_t idx, bool is_r1) {
  Rec r;
  r.combined_score = o.score;
  r.entry = o.entry;
  r.ref_start = o.ref_begin;
  r.ref_end = o.ref_end;
  r.insert_size = 0;
  r.r1 = is_r1 ? idx : NONE;
  r.r2 = is_r1 ? NONE : idx;
  r.pad = 0;
  return r;
}

// first row of every read (the rows are sorted by read): row_start[r] for r in 0 .. n_reads, row_start[n_reads] = n.  One
// thread per row fills the entries between its predecessor's read and its own -- k_pair then finds a read pair's two runs
// with four loads instead of four binary searches of 23 dependent probes each (0.7 of its 1.2 ms).
__global__ void k_row_starts(const kslam_overlap *__restrict__ ov, uint64_t n, uint64_t n_reads, uint32_t *__restrict__ row_start,
                             uint4 *__restrict__ gaps, uint32_t gap_cap, uint32_t *__restrict__ n_gaps, uint32_t *__restrict__ bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  const int64_t prev = i > 0 ? (int64_t)ov[i - 1].read : -1;
  const int64_t cur = min(i < n ? (int64_t)ov[i].read : (int64_t)n_reads, (int64_t)n_reads);
  // rows that are not sorted by read, or name a read the batch does not have (a caller's own rows, kslam_pair_screen_rows):
  // the table would have holes; the host fails the call (k_pair clamps what it reads).  Sorted rows have at most
  // n_reads / 64 + 1 stretches of more than 64 reads = gap_cap; rows that jump back and forth (0, 100, 0, 100, ...) would
  // list one per row: those beyond the capacity are dropped and the call is marked bad, nothing is written out of bounds
  if (prev > cur || (i < n && (uint64_t)ov[i].read >= n_reads)) atomicOr(bad, 1u);
  if (cur - prev > 64) {   // a long stretch of reads without rows (an empty result: all of them): k_fill_gaps, in parallel
    const uint32_t g = atomicAdd(n_gaps, 1u);
    if (g < gap_cap) gaps[g] = make_uint4((uint32_t)(prev + 1), (uint32_t)cur, (uint32_t)i, 0u);
    else atomicOr(bad, 1u);
    return;
  }
  for (int64_t r = prev + 1; r <= cur; r++) row_start[r] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void k_fill_gaps(const uint4 *__restrict__ gaps, const uint32_t *__restrict__ n_gaps, uint32_t gap_cap,
                                                    uint32_t *__restrict__ row_start) {
  const uint32_t ng = min(*n_gaps, gap_cap);
  for (uint32_t g = blockIdx.x; g < ng; g += gridDim.x) {
    const uint4 e = gaps[g];
    for (uint64_t r = (uint64_t)e.x + threadIdx.x; r <= e.y; r += 256) row_start[r] = e.z;
  }
}

struct PairArgs {
  const kslam_overlap *ov;
  uint64_t n;              // overlap records
  const uint32_t *row_start;   // [n_reads + 1] (k_row_starts)
  const uint32_t *read_len;
  uint64_t units, mid;     // read pairs (or reads); reads per block
  uint32_t thr;            // score threshold
  int paired;
  Rec *recs;               // 4 x n records of room
  uint32_t *count;         // per unit
  uint64_t *base;          // per unit: where its region starts
  int32_t *inserts;        // n x 2 of room
  unsigned long long *n_inserts, *n_kept, *n_initial;
  uint32_t *big_list, *n_big;   // read pairs left to k_pair_big (nullptr: none are)
};

// getPairsFromRead as a streaming state (host/tail.cpp: Pairer): one candidate slot per (mate, strand).  The rows
// [i, i1) of mate 1 and [j, j1) of mate 2 are merged by (entry, rel), ties in input order (R1 first); every record goes to
// emit(const Rec &).  Runs of different entries do not interact (the slots are emptied when the entry changes), so walking
// ONE entry's rows of the two mates gives exactly that entry's records: k_pair_big below does that, a lane per entry.
template <class Emit>
__device__ inline uint32_t pair_walk(const kslam_overlap *__restrict__ ov, uint64_t i, uint64_t i1, uint64_t j, uint64_t j1, uint32_t thr,
                                     const uint32_t *__restrict__ read_len, Emit &&emit) {
  uint32_t kept = 0;
  uint32_t slot[2][2] = {{NONE, NONE}, {NONE, NONE}};
  bool used[2][2] = {{false, false}, {false, false}};
  bool open = false;
  uint32_t cur_entry = 0;
  auto emit_single = [&](uint32_t idx, bool is_r1) { emit(single_rec(ov[idx], idx, is_r1)); };
  auto close_run = [&]() {   // the order of src/PairedOverlap.h:217-240
    if (!used[1][0] && slot[1][0] != NONE) emit_single(slot[1][0], false);
    if (!used[1][1] && slot[1][1] != NONE) emit_single(slot[1][1], false);
    if (!used[0][0] && slot[0][0] != NONE) emit_single(slot[0][0], true);
    if (!used[0][1] && slot[0][1] != NONE) emit_single(slot[0][1], true);
    for (int m = 0; m < 2; m++)
      for (int s = 0; s < 2; s++) { slot[m][s] = NONE; used[m][s] = false; }
  };
  while (i < i1 || j < j1) {
    bool take1;
    if (j >= j1) take1 = true;
    else if (i >= i1) take1 = false;
    else {
      const kslam_overlap &x = ov[i], &y = ov[j];
      take1 = x.entry != y.entry ? x.entry < y.entry : x.rel <= y.rel;
    }
    const uint32_t idx = (uint32_t)(take1 ? i++ : j++);
    const kslam_overlap o = ov[idx];
    if (o.score < thr) continue;   // src/Overlap.h:329-341
    kept++;
    if (open && o.entry != cur_entry) close_run();
    open = true;
    cur_entry = o.entry;
    const int s = o.revcomp ? 1 : 0, m = take1 ? 0 : 1;
    if (!used[m][s] && slot[m][s] != NONE) emit_single(slot[m][s], m == 0);
    slot[m][s] = idx;
    used[m][s] = false;
    const uint32_t other = slot[1 - m][1 - s];
    if (other != NONE) {   // makePair, src/PairedOverlap.h:107-125
      const uint32_t i1x = m == 0 ? idx : other, i2x = m == 0 ? other : idx;
      const kslam_overlap &p = ov[i1x], &q = ov[i2x];
      const bool r1_first = m != 0;
      const uint32_t ins = r1_first ? (uint32_t)((int64_t)q.rel - p.rel + read_len[q.read])
                                    : (uint32_t)((int64_t)p.rel - q.rel + read_len[p.read]);
      Rec r;
      r.combined_score = (uint16_t)(p.score + q.score);   // PairedOverlap's uint16_t parameter
      r.entry = q.entry;
      r.ref_start = min(p.ref_begin, q.ref_begin);
      r.ref_end = max(p.ref_end, q.ref_end);
      r.insert_size = ins;
      r.r1 = i1x;
      r.r2 = i2x;
      r.pad = 0;
      emit(r);
      used[m][s] = true;
      used[1 - m][1 - s] = true;
    }
  }
  if (open) close_run();
  return kept;
}

// Read pairs with more than PAIR_BIG rows (reads inside an rRNA-like repeat meet every genome of the database: ~17 000 rows a
// pair) are left to k_pair_big: one thread walking them record by record, a dependent 48-byte gather each, was 11 ms of the
// repeat-rich bench's 75 ms step -- the longest thread's time, the chip waiting.
constexpr uint32_t PAIR_BIG = 256;

__global__ __launch_bounds__(256) void k_pair(PairArgs a) {
  const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t n_recs = 0, kept = 0;
  Rec *out = nullptr;
  if (u < a.units) {
    const kslam_overlap *ov = a.ov;
    if (a.paired) {
      const uint64_t split = min<uint64_t>(a.row_start[a.mid], a.n);
      uint64_t i = min<uint64_t>(a.row_start[u], split), i1 = min<uint64_t>(max<uint64_t>(a.row_start[u + 1], i), split);
      uint64_t j = min<uint64_t>(max<uint64_t>(a.row_start[a.mid + u], split), a.n),
               j1 = min<uint64_t>(max<uint64_t>(a.row_start[a.mid + u + 1], j), a.n);   // (clamps: see k_row_starts)
      const uint64_t base = 4 * (i + (j - split));
      a.base[u] = base;
      out = a.recs + base;
      if ((i1 - i) + (j1 - j) > PAIR_BIG && a.big_list) {   // a wavefront's job: count[u], the inserts and the counters come from k_pair_big
        a.big_list[atomicAdd(a.n_big, 1u)] = (uint32_t)u;
        out = nullptr;
      } else {
        kept = pair_walk(ov, i, i1, j, j1, a.thr, a.read_len, [&](const Rec &r) { out[n_recs++] = r; });
      }
    } else {
      // getPerReadOverlaps (single end) + dummy pairs: every overlap of the read as an R1-only record
      uint64_t i = min<uint64_t>(a.row_start[u], a.n), i1 = min<uint64_t>(max<uint64_t>(a.row_start[u + 1], i), a.n);
      const uint64_t base = 4 * i;
      a.base[u] = base;
      out = a.recs + base;
      for (; i < i1; i++) {
        const kslam_overlap o = ov[i];
        if (o.score < a.thr) continue;
        kept++;
        out[n_recs++] = single_rec(o, (uint32_t)i, true);
      }
    }
    if (out) a.count[u] = n_recs;
  }
  // insert sizes of this thread's records, appended with ONE returning atomic per workgroup (and one each for the
  // two counters): same-address atomics take ~12 ns apiece on this chip whoever issues them, and one per wave --
  // 47 k for a 2 M-read batch -- was 0.4 ms of this kernel's 1.3
  __shared__ uint32_t s_ins[4], s_kept[4], s_recs[4];
  __shared__ unsigned long long s_base;
  uint32_t n_ins = 0;
  for (uint32_t k = 0; k < n_recs; k++) n_ins += out[k].insert_size != 0;
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t incl = n_ins, tot_k = kept, tot_r = n_recs;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(incl, d, 64);
    if (lane >= (uint32_t)d) incl += t;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    tot_k += __shfl_down(tot_k, d, 64);
    tot_r += __shfl_down(tot_r, d, 64);
  }
  const uint32_t wave_total = __shfl(incl, 63, 64);
  if (lane == 0) { s_ins[wv] = wave_total; s_kept[wv] = tot_k; s_recs[wv] = tot_r; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t bi = s_ins[0] + s_ins[1] + s_ins[2] + s_ins[3], bk = s_kept[0] + s_kept[1] + s_kept[2] + s_kept[3],
                   br = s_recs[0] + s_recs[1] + s_recs[2] + s_recs[3];
    s_base = bi ? atomicAdd(a.n_inserts, (unsigned long long)bi) : 0ull;
    if (bk) atomicAdd(a.n_kept, (unsigned long long)bk);
    if (br) atomicAdd(a.n_initial, (unsigned long long)br);
  }
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t w = 0; w < wv; w++) before += s_ins[w];
  unsigned long long at = s_base + before + (incl - n_ins);
  for (uint32_t k = 0; k < n_recs; k++)
    if (out[k].insert_size != 0) a.inserts[at++] = (int32_t)out[k].insert_size;
}

// first row in [lo, hi) whose entry is not below e
__device__ inline uint64_t first_entry_at_least(const kslam_overlap *__restrict__ ov, uint64_t lo, uint64_t hi, uint32_t e) {
  while (lo < hi) {
    const uint64_t m = (lo + hi) >> 1;
    if (ov[m].entry < e) lo = m + 1; else hi = m;
  }
  return lo;
}

// One big read pair by one wavefront.  A lane takes an ENTRY: the entry's rows of mate 1 (found where a run of equal entries
// starts in the mate's row range) and of mate 2 (binary search), walked by pair_walk -- first to count the entry's records,
// then, after a scan of the counts over the merged row order (an entry's first merged position is (rows of mate 1 with a
// smaller entry) + (rows of mate 2 with a smaller entry)), to write them where the one-thread walk puts them.  The counts
// and the entries' bounds live in the upper half of the read pair's own record region (4 records of room per row, at most 2 used).
__global__ __launch_bounds__(64) void k_pair_big(PairArgs a) {
  const uint32_t lane = threadIdx.x;
  const uint32_t n_big = *a.n_big;
  const kslam_overlap *ov = a.ov;
  for (uint32_t b = blockIdx.x; b < n_big; b += gridDim.x) {
    const uint64_t u = a.big_list[b];
    const uint64_t split = min<uint64_t>(a.row_start[a.mid], a.n);
    const uint64_t i = min<uint64_t>(a.row_start[u], split), i1 = min<uint64_t>(max<uint64_t>(a.row_start[u + 1], i), split);
    const uint64_t j = min<uint64_t>(max<uint64_t>(a.row_start[a.mid + u], split), a.n),
                   j1 = min<uint64_t>(max<uint64_t>(a.row_start[a.mid + u + 1], j), a.n);
    const uint64_t n1 = i1 - i, n2 = j1 - j, nm = n1 + n2;
    Rec *out = a.recs + a.base[u];
    uint32_t *cnt = reinterpret_cast<uint32_t *>(out + 2 * nm);   // (nm + 1) counts in the unused upper half of the region
    for (uint64_t k = lane; k <= nm; k += 64) cnt[k] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // an entry of the pair, as seen from a row: (mate-1 rows, mate-2 rows, first merged position); false: not this row's job
    auto entry_of_row = [&](uint64_t k, uint64_t *a1, uint64_t *b1, uint64_t *a2, uint64_t *b2, uint64_t *pos) -> bool {
      if (k < n1) {                        // a row of mate 1: the head of its entry's run there
        const uint64_t r = i + k;
        const uint32_t e = ov[r].entry;
        if (r > i && ov[r - 1].entry == e) return false;
        uint64_t z = r + 1;
        while (z < i1 && ov[z].entry == e) z++;
        *a1 = r; *b1 = z;
        *a2 = first_entry_at_least(ov, j, j1, e);
        *b2 = *a2;
        while (*b2 < j1 && ov[*b2].entry == e) (*b2)++;
      } else {                             // a row of mate 2: only entries mate 1 does not have
        const uint64_t r = j + (k - n1);
        const uint32_t e = ov[r].entry;
        if (r > j && ov[r - 1].entry == e) return false;
        const uint64_t lb = first_entry_at_least(ov, i, i1, e);
        if (lb < i1 && ov[lb].entry == e) return false;
        uint64_t z = r + 1;
        while (z < j1 && ov[z].entry == e) z++;
        *a1 = lb; *b1 = lb; *a2 = r; *b2 = z;
      }
      *pos = (*a1 - i) + (*a2 - j);
      return true;
    };
    // the rows that speak for an entry, compacted (one row in seven does: walking the rows themselves left most lanes idle),
    // with the entry's four bounds; both in the unused upper half of the region as well
    uint4 *bounds = reinterpret_cast<uint4 *>(out + 3 * nm);   // 16 bytes per head, at most nm heads: bytes 96 nm .. 112 nm of the region's 128 nm (the counts: 64 nm .. 68 nm)
    uint32_t n_heads = 0;
    for (uint64_t k0 = 0; k0 < nm; k0 += 64) {
      const uint64_t k = k0 + lane;
      uint64_t a1 = 0, b1 = 0, a2 = 0, b2 = 0, pos = 0;
      const bool head = k < nm && entry_of_row(k, &a1, &b1, &a2, &b2, &pos);
      const uint64_t m = __ballot(head);
      if (head) bounds[n_heads + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] =
          make_uint4((uint32_t)(a1 - i), (uint32_t)(b1 - a1), (uint32_t)(a2 - j), (uint32_t)(b2 - a2));
      n_heads += (uint32_t)__popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint32_t kept = 0;
    for (uint32_t h0 = 0; h0 < n_heads; h0 += 64) {
      const uint32_t h = h0 + lane;
      if (h < n_heads) {
        const uint4 e = bounds[h];
        uint32_t c = 0;
        kept += pair_walk(ov, i + e.x, i + e.x + e.y, j + e.z, j + e.z + e.w, a.thr, a.read_len, [&](const Rec &) { c++; });
        cnt[e.x + e.z] = c;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // exclusive scan of the counts in merged row order
    uint32_t carry = 0;
    for (uint64_t k0 = 0; k0 < nm; k0 += 64) {
      const uint64_t k = k0 + lane;
      const uint32_t c = k < nm ? cnt[k] : 0u;
      uint32_t incl = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += t;
      }
      if (k < nm) cnt[k] = carry + incl - c;
      carry += __shfl(incl, 63, 64);
    }
    const uint32_t n_recs = carry;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (uint32_t h0 = 0; h0 < n_heads; h0 += 64) {
      const uint32_t h = h0 + lane;
      if (h < n_heads) {
        const uint4 e = bounds[h];
        uint32_t at = cnt[e.x + e.z];
        (void)pair_walk(ov, i + e.x, i + e.x + e.y, j + e.z, j + e.z + e.w, a.thr, a.read_len, [&](const Rec &r) { out[at++] = r; });
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // the counters and the non-zero insert sizes (their order does not matter: the statistics sort them)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) kept += __shfl_down(kept, d, 64);
    if (lane == 0) {
      a.count[u] = n_recs;
      if (kept) atomicAdd(a.n_kept, (unsigned long long)kept);
      if (n_recs) atomicAdd(a.n_initial, (unsigned long long)n_recs);
    }
    for (uint32_t k0 = 0; k0 < n_recs; k0 += 64) {
      const uint32_t k = k0 + lane;
      const uint32_t ins = k < n_recs ? out[k].insert_size : 0u;
      const uint64_t m = __ballot(ins != 0);
      if (!m) continue;
      unsigned long long at = 0;
      if (lane == 0) at = atomicAdd(a.n_inserts, (unsigned long long)__popcll(m));
      at = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(at >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)at);
      if (ins != 0) a.inserts[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (int32_t)ins;
    }
  }
}

__global__ void k_widen(const int32_t *__restrict__ v, uint64_t n, uint2 *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = make_uint2((uint32_t)v[i], 0u);
}
__global__ void k_pick(const uint2 *__restrict__ sorted, const uint64_t *__restrict__ idx, uint32_t k, int32_t *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k) out[i] = (int32_t)sorted[idx[i]].x;
}
// sums over the kept values lo <= v <= hi: sum, sum of squares (the reference multiplies in int: the
// wrap-around is kept), sum of |squares| (to know whether the double accumulation stays exact), count
__global__ __launch_bounds__(256) void k_insert_sums(const uint2 *__restrict__ sorted, uint64_t n, int32_t lo, int32_t hi,
                                                     long long *__restrict__ out /*[4]*/) {
  long long s1 = 0, s2 = 0, mag = 0, cnt = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const int32_t v = (int32_t)sorted[i].x;
    if (v < lo || v > hi) continue;
    const int32_t sq = (int32_t)((uint32_t)v * (uint32_t)v);
    s1 += v;
    s2 += sq;
    mag += sq < 0 ? -(long long)sq : (long long)sq;
    cnt++;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    s1 += __shfl_down(s1, d, 64);
    s2 += __shfl_down(s2, d, 64);
    mag += __shfl_down(mag, d, 64);
    cnt += __shfl_down(cnt, d, 64);
  }
  __shared__ long long part[4][4];
  if ((threadIdx.x & 63) == 0) {
    const uint32_t wv = threadIdx.x >> 6;
    part[wv][0] = s1; part[wv][1] = s2; part[wv][2] = mag; part[wv][3] = cnt;
  }
  __syncthreads();
  if (threadIdx.x < 4) {    // one atomic per workgroup and sum (the grid is at most 256 workgroups)
    const long long v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    atomicAdd(reinterpret_cast<unsigned long long *>(out + threadIdx.x), (unsigned long long)v);
  }
}

struct ByInsert {
  __host__ __device__ bool operator()(const Rec &x, const Rec &y) const { return x.insert_size < y.insert_size; }
};
struct ByScoreDesc {
  __host__ __device__ bool operator()(const Rec &x, const Rec &y) const { return x.combined_score > y.combined_score; }
};

// screenPairedAlignmentsByInsertSize(replace = true) then screenPairedAlignmentsByScore on one read
// pair's records (host/tail.cpp: insert_screen, score_screen)
// Read pairs with more than SCREEN_BIG alignment pairs (reads in rRNA-like repeats pair up on every genome of the database:
// thousands) are left to k_screen_big: one thread sorting thousands of records while its 63 neighbours wait made this kernel
// 55 ms of a 130 ms step on the repeat-rich database.
constexpr uint32_t SCREEN_BIG = 96;
__global__ __launch_bounds__(256) void k_screen(const kslam_overlap *__restrict__ ov, Rec *__restrict__ recs,
                                                const uint64_t *__restrict__ base, uint32_t *__restrict__ count,
                                                uint64_t units, int do_insert, uint32_t limit, int do_score,
                                                double fraction, uint32_t *__restrict__ flags, uint32_t *__restrict__ big_list,
                                                uint32_t *__restrict__ n_big) {
  const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) return;
  uint32_t n = count[u];
  if (n > SCREEN_BIG && (do_insert || do_score)) {   // a wavefront's job: flags[u] and count[u] come from k_screen_big
    big_list[atomicAdd(n_big, 1u)] = (uint32_t)u;
    return;
  }
  if (n) {
    Rec *v = recs + base[u];
    if (do_insert) {
      kslam_gnu::sort(v, v + n, ByInsert());
      uint32_t cut = 0;
      while (cut < n && !(v[cut].insert_size > limit)) cut++;
      uint32_t end = n;
      for (uint32_t i = cut; i < n; i++) {
        const kslam_overlap &o1 = ov[v[i].r1], &o2 = ov[v[i].r2];
        Rec r;
        r.combined_score = o1.score;
        r.entry = v[i].entry;
        r.ref_start = o1.ref_begin;
        r.ref_end = o1.ref_end;
        r.insert_size = 0;
        r.r1 = v[i].r1;
        r.r2 = NONE;
        r.pad = 0;
        v[end++] = r;
        Rec &c = v[i];
        c.combined_score = o2.score;
        c.insert_size = 0;
        c.r1 = NONE;
        c.ref_start = o2.ref_begin;
        c.ref_end = o2.ref_end;
      }
      n = end;
    }
    if (do_score) {
      kslam_gnu::sort(v, v + n, ByScoreDesc());
      const unsigned top = v[0].combined_score;
      const double bar = top * fraction;
      uint32_t k = 0;
      while (k < n && !((double)v[k].combined_score < bar)) k++;
      n = k;
    }
    count[u] = n;
  }
  flags[u] = n ? 1u : 0u;
}

// The same two screens for ONE big read pair by one wavefront: both std::sorts with libstdc++'s permutation produced by the
// 64 lanes together (wave_gnu_sort.h), the split of the records beyond the insert-size limit and the cut of the score
// screen lane-parallel.  Statement for statement what the thread above does.
__global__ __launch_bounds__(64) void k_screen_big(const kslam_overlap *__restrict__ ov, Rec *__restrict__ recs,
                                                   const uint64_t *__restrict__ base, uint32_t *__restrict__ count,
                                                   const uint32_t *__restrict__ big_list, int do_insert, uint32_t limit, int do_score,
                                                   double fraction, uint32_t *__restrict__ flags) {
  __shared__ kslam_gnu::WaveSortLds S;
  const uint32_t u = big_list[blockIdx.x], lane = threadIdx.x;
  uint32_t n = count[u];
  Rec *v = recs + base[u];
  if (do_insert) {
    kslam_gnu::wave_sort(v, n, ByInsert(), S);
    __syncthreads();
    // cut = the first record beyond the limit (the records are sorted by insert size)
    uint32_t mine = n;
    for (uint32_t i = lane; i < n; i += 64)
      if (v[i].insert_size > limit) { mine = i; break; }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) mine = min(mine, (uint32_t)__shfl_xor((int)mine, m, 64));
    const uint32_t cut = mine;
    for (uint32_t i = cut + lane; i < n; i += 64) {   // record i becomes its R2 half, its R1 half is appended at n + (i - cut)
      const kslam_overlap &o1 = ov[v[i].r1], &o2 = ov[v[i].r2];
      Rec r;
      r.combined_score = o1.score;
      r.entry = v[i].entry;
      r.ref_start = o1.ref_begin;
      r.ref_end = o1.ref_end;
      r.insert_size = 0;
      r.r1 = v[i].r1;
      r.r2 = NONE;
      r.pad = 0;
      v[n + (i - cut)] = r;
      Rec &c = v[i];
      c.combined_score = o2.score;
      c.insert_size = 0;
      c.r1 = NONE;
      c.ref_start = o2.ref_begin;
      c.ref_end = o2.ref_end;
    }
    n += n - cut;
    __syncthreads();
  }
  if (do_score) {
    kslam_gnu::wave_sort(v, n, ByScoreDesc(), S);
    __syncthreads();
    const unsigned top = v[0].combined_score;
    const double bar = top * fraction;
    uint32_t mine = n;   // the first record under the bar (sorted by score, descending)
    for (uint32_t i = lane; i < n; i += 64)
      if ((double)v[i].combined_score < bar) { mine = i; break; }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) mine = min(mine, (uint32_t)__shfl_xor((int)mine, m, 64));
    n = mine;
  }
  if (lane == 0) {
    count[u] = n;
    flags[u] = n ? 1u : 0u;
  }
}

__global__ __launch_bounds__(256) void k_emit_groups(const Rec *__restrict__ recs, const uint64_t *__restrict__ base,
                                                     const uint32_t *__restrict__ count, const uint32_t *__restrict__ gpos,
                                                     const uint64_t *__restrict__ rpos, uint64_t units, uint64_t mid,
                                                     int paired, kslam_read_pair *__restrict__ groups,
                                                     Rec *__restrict__ dense) {
  const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) return;
  const uint32_t n = count[u];
  if (!n) return;
  kslam_read_pair g;
  g.r1_read = (uint32_t)u;
  g.r2_read = paired ? (uint32_t)(u + mid) : 0u;
  g.first = rpos[u];
  g.count = n;
  groups[gpos[u]] = g;
  const Rec *src = recs + base[u];
  Rec *dst = dense + rpos[u];
  for (uint32_t k = 0; k < n; k++) dst[k] = src[k];
}

// ---- pseudoAssembly (src/PairedOverlap.h:480-582; host/tail.cpp: pseudo_stage, chain_entry) -----------
// The reference buckets the alignment pairs per entry in iteration order, sorts each bucket by refStart
// with std::sort and walks it once: alignments that overlap along the genome (the next one starts more
// than 20 bases before the highest position reached) form a chain, and every member of a chain of two or
// more gets the score  coverage x average score per base x length.  The sums of that walk are doubles
// added in the SORTED order, and equal starts are common, so the permutation of ties is part of the
// result: the sort is libstdc++'s, reproduced by one wave per entry with the entry's spans in LDS
// (wave_gnu_sort.h; ~1 300 spans per entry on the bench workload, all entries at once across the chip).
struct PSpan {   // host: Span {start, stop, rec} + the record's score (abs(stop - start) is its span).  The layout
  uint32_t score;  // is the bucket sort's record {entry, start, stop, rec} with the score in the entry's place,
  int32_t start, stop;   // so that a big entry can be worked on where the sort left it
  uint32_t rec;
};
struct ByStart {
  __host__ __device__ bool operator()(const PSpan &a, const PSpan &b) const { return a.start < b.start; }
};
constexpr uint32_t PSEUDO_CAP = 4000;         // spans per entry that fit a workgroup's LDS (64 000 B)
constexpr uint32_t PSEUDO_CAP_GLOBAL = 1u << 18;   // beyond that one wave per entry is the wrong tool: the host runs the stage

__global__ void k_spans(const Rec *__restrict__ recs, const kslam_read_pair *__restrict__ groups, uint64_t n_groups,
                        uint4 *__restrict__ out, uint32_t *__restrict__ max_entry) {
  // one thread per read pair: its records, in order, as {entry, start, stop, rec} (the bucket sort is stable,
  // so the buckets come out in the reference's iteration order)
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t mx = 0;
  if (g < n_groups) {
    const uint64_t first = groups[g].first, cnt = groups[g].count;
    for (uint64_t k = 0; k < cnt; k++) {
      const Rec r = recs[first + k];
      out[first + k] = make_uint4(r.entry, (uint32_t)r.ref_start, (uint32_t)r.ref_end, (uint32_t)(first + k));
      mx = max(mx, r.entry);
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
  if ((threadIdx.x & 63) == 0 && mx) atomicMax(max_entry, mx);   // one atomic per wave
}

__global__ void k_entry_runs(const uint4 *__restrict__ sorted, uint64_t n, uint32_t *__restrict__ flags) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = (i == 0 || sorted[i].x != sorted[i - 1].x) ? 1u : 0u;
}
__global__ void k_run_starts(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ pos, uint64_t n,
                             uint32_t *__restrict__ run_start) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flags[i]) return;
  run_start[pos[i]] = (uint32_t)i;
}
__global__ void k_run_longest(const uint32_t *__restrict__ run_start, uint32_t n_runs, uint32_t n, uint32_t *__restrict__ longest) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_runs) return;
  const uint32_t len = (r + 1 < n_runs ? run_start[r + 1] : n) - run_start[r];
  atomicMax(longest, len);
}

// the reference's `uint32_t s = score;` as x86-64 evaluates it (cvttsd2si to 64 bits, low half kept):
// NaN, infinities and values outside int64 give 0
__device__ inline uint32_t cvt_u32_like_x86(double d) {
  if (!(d > -9223372036854775808.0 && d < 9223372036854775808.0)) return 0u;
  return (uint32_t)(uint64_t)(long long)d;
}

// What the stage reads and writes of an alignment-pair record is its first 16 bytes.  When the entries are partitioned
// over the ranks of a sharded batch (pseudo_route / pseudo_owned / pseudo_return below) only those travel, and the kernels
// of the stage run on them as they run on whole records: R = Rec or RecHead.
struct RecHead {
  uint32_t combined_score;
  uint32_t entry;
  int32_t ref_start;
  int32_t ref_end;
};
static_assert(sizeof(RecHead) == 16 && offsetof(Rec, combined_score) == 0 && offsetof(Rec, entry) == 4 &&
              offsetof(Rec, ref_start) == 8 && offsetof(Rec, ref_end) == 12, "RecHead is the first half of kslam_paired_overlap");

// one chain of an entry's sorted spans: [from, to); host/tail.cpp chain_entry's sums, operation for operation
template <typename P, typename R>
__device__ inline void score_chain(P v, uint32_t from, uint32_t to, R *__restrict__ recs) {
#pragma clang fp contract(off)
  const long len = (long)to - (long)from;
  if (len <= 1) return;
  int reach = v[from].stop;
  int span = abs(v[from].stop - v[from].start);
  double per_base = v[from].score * 1.0 / span;
  uint32_t bases = (uint32_t)span;
  for (uint32_t i = from + 1; i < to; i++) {
    span = abs(v[i].stop - v[i].start);
    if (v[i].stop > reach) reach = v[i].stop;
    per_base += v[i].score * 1.0 / span;
    bases += (uint32_t)span;
  }
  const double length = reach - v[from].start;
  const double coverage = bases / length;
  const double avg = per_base / len;
  const double score = coverage * avg * length;
  const uint32_t s = cvt_u32_like_x86(score);
  for (uint32_t k = from; k < to; k++) recs[v[k].rec & 0x7FFFFFFFu].combined_score = s;
}

// One wave per entry.  (1) its spans into LDS -- or, for an entry with more spans than LDS takes, left where
// the bucket sort put them in global memory (BIG) --; (2) std::sort by start, the permutation included, by the
// whole wave (wave_gnu_sort.h); (3) chain starts: the reference starts a chain where start > reach - 20 with
// reach = the highest stop of the chain so far -- and, the starts being sorted, every stop of an earlier
// chain is below this chain's first start + 20, so reach can be the running maximum over ALL earlier spans:
// a prefix maximum, 64 spans a step; (4) one lane per chain adds up the chain in order (the chains are
// independent; the additions inside one are the reference's sequence) and writes the members' new scores.
template <typename P, typename R>
__device__ inline void pseudo_entry_body(P v, uint32_t cnt, R *__restrict__ recs, kslam_gnu::WaveSortLds &S) {
  const uint32_t lane = threadIdx.x;
  kslam_gnu::wave_sort(v, cnt, ByStart(), S);
  __syncthreads();
  int carry = -1000000;
  for (uint32_t base = 0; base < cnt; base += 64) {
    const uint32_t i = base + lane;
    const int stop = i < cnt ? v[i].stop : INT32_MIN;
    int incl = stop;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= (uint32_t)d) incl = max(incl, t);
    }
    int before = __shfl_up(incl, 1, 64);
    before = lane ? max(before, carry) : carry;
    if (i < cnt && v[i].start > before - 20) v[i].rec |= 0x80000000u;   // a chain starts here
    carry = max(carry, __shfl(incl, 63, 64));
  }
  __syncthreads();
  for (uint32_t i = lane; i < cnt; i += 64) {
    if (!(v[i].rec & 0x80000000u)) continue;
    uint32_t to = i + 1;
    while (to < cnt && !(v[to].rec & 0x80000000u)) to++;
    score_chain(v, i, to, recs);
  }
}

template <bool BIG, typename R>
__global__ __launch_bounds__(64) void k_pseudo_entry(uint4 *__restrict__ sorted, const uint32_t *__restrict__ run_start,
                                                     uint32_t n_runs, uint32_t n, R *__restrict__ recs) {
  extern __shared__ PSpan v_lds[];
  __shared__ kslam_gnu::WaveSortLds S;
  const uint32_t r = blockIdx.x, lane = threadIdx.x;
  const uint32_t lo = run_start[r], cnt = (r + 1 < n_runs ? run_start[r + 1] : n) - lo;
  if ((cnt > PSEUDO_CAP) != BIG) return;     // the other launch's entry
  PSpan *g = reinterpret_cast<PSpan *>(sorted + lo);
  for (uint32_t k = lane; k < cnt; k += 64) {
    PSpan p = g[k];                     // {entry, start, stop, rec}
    p.score = recs[p.rec].combined_score;
    if (BIG) g[k] = p; else v_lds[k] = p;
  }
  __syncthreads();
  if (BIG) pseudo_entry_body(g, cnt, recs, S);
  else pseudo_entry_body(v_lds, cnt, recs, S);
}

// kslam_debug_wave_sort: segments of keys, sorted by key with wave_sort (in LDS up to PSEUDO_CAP keys, in
// global memory beyond, as k_pseudo_entry does); out = the permutation (element ids)
__global__ __launch_bounds__(64) void k_debug_wave_sort(const int32_t *__restrict__ keys, const uint64_t *__restrict__ seg_off,
                                                        uint32_t *__restrict__ perm, PSpan *__restrict__ scratch) {
  extern __shared__ PSpan v_lds[];
  __shared__ kslam_gnu::WaveSortLds S;
  const uint64_t lo = seg_off[blockIdx.x];
  const uint32_t cnt = (uint32_t)(seg_off[blockIdx.x + 1] - lo);
  const bool big = cnt > PSEUDO_CAP;
  PSpan *g = scratch + lo;
  for (uint32_t k = threadIdx.x; k < cnt; k += 64) {
    PSpan p;
    p.start = keys[lo + k];
    p.stop = 0;
    p.rec = k;
    p.score = 0;
    if (big) g[k] = p; else v_lds[k] = p;
  }
  __syncthreads();
  if (big) kslam_gnu::wave_sort(g, cnt, ByStart(), S);
  else kslam_gnu::wave_sort(v_lds, cnt, ByStart(), S);
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < cnt; k += 64) perm[lo + k] = big ? g[k].rec : v_lds[k].rec;
}

// screenPairedAlignmentsByScore once more, in place on the dense records (host/tail.cpp: rescreen_stage)
__global__ __launch_bounds__(256) void k_rescreen(Rec *__restrict__ recs, kslam_read_pair *__restrict__ groups, uint64_t n_groups,
                                                  double fraction) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const uint32_t n = (uint32_t)groups[g].count;
  if (!n) return;
  Rec *v = recs + groups[g].first;
  kslam_gnu::sort(v, v + n, ByScoreDesc());
  const unsigned top = v[0].combined_score;
  const double bar = top * fraction;
  uint32_t k = 0;
  while (k < n && !((double)v[k].combined_score < bar)) k++;
  groups[g].count = k;
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void k_mark_rows(const Rec *__restrict__ recs, const kslam_read_pair *__restrict__ groups,
                                                   uint64_t n_groups, uint64_t n_rows, uint32_t *__restrict__ flags) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const uint64_t first = groups[g].first, cnt = groups[g].count;
  for (uint64_t k = 0; k < cnt; k++) {
    const Rec r = recs[first + k];
    if (r.r1 != NONE && r.r1 < n_rows) flags[r.r1] = 1u;
    if (r.r2 != NONE && r.r2 < n_rows) flags[r.r2] = 1u;
  }
}
__global__ void k_list_rows(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ pos, uint64_t n,
                            uint32_t *__restrict__ list) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) list[pos[i]] = (uint32_t)i;
}
}  // namespace

// Which overlap records do the surviving alignment pairs refer to?  (The SAM writer's per-row walk is only needed
// for those: ~36 % of the rows of the bench workload.)
void referenced_rows(PairWork &W, const PairResult *res, uint64_t n_rows, const uint32_t **d_list, uint64_t *n_list, hipStream_t s) {
  *d_list = nullptr;
  *n_list = 0;
  if (n_rows == 0) return;
  W.flags.ensure((n_rows + 1) * sizeof(uint32_t));
  W.gpos.ensure((n_rows + 1) * sizeof(uint32_t));
  W.scan_tmp.ensure(scan_tmp_bytes(n_rows));
  W.row_list.ensure((n_rows + 1) * sizeof(uint32_t));
  uint64_t *tot = W.totals.as<uint64_t>();
  HIPCHK(hipMemsetAsync(W.flags.p, 0, n_rows * sizeof(uint32_t), s));
  if (res->n_read_pairs)
    hipLaunchKernelGGL(k_mark_rows, dim3((unsigned)((res->n_read_pairs + 255) / 256)), dim3(256), 0, s, res->d_pairs, res->d_groups,
                       res->n_read_pairs, n_rows, W.flags.as<uint32_t>());
  exclusive_scan_u32(W.flags.as<uint32_t>(), W.gpos.as<uint32_t>(), n_rows, tot + 15, W.scan_tmp.p, s);
  hipLaunchKernelGGL(k_list_rows, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, s, W.flags.as<uint32_t>(), W.gpos.as<uint32_t>(),
                     n_rows, W.row_list.as<uint32_t>());
  HIPCHK(hipGetLastError());
  uint64_t cnt = 0;
  read_back(&cnt, tot + 15, sizeof cnt, s);
  *d_list = W.row_list.as<uint32_t>();
  *n_list = cnt;
}

template <typename R>
__global__ void k_spans_flat(const R *__restrict__ recs, uint64_t n, uint4 *__restrict__ out, uint32_t *__restrict__ max_entry) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t mx = 0;
  if (i < n) {
    const R r = recs[i];
    out[i] = make_uint4(r.entry, (uint32_t)r.ref_start, (uint32_t)r.ref_end, (uint32_t)i);
    mx = r.entry;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
  if ((threadIdx.x & 63) == 0 && mx) atomicMax(max_entry, mx);
}
__global__ void k_take_scores(const Rec *__restrict__ all, uint64_t base, uint64_t n, Rec *__restrict__ own) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) own[i].combined_score = all[base + i].combined_score;
}


// pseudo-assembly + second score screen on the result of pair_and_screen, in place.  Returns false, having
// changed nothing, when an entry holds more spans than one workgroup's LDS takes (the host then runs the stage).
// spans of `recs` (in record order, or group by group: the same order, the dense array is laid out by group) ->
// stable bucket sort by entry -> one wavefront per entry: std::sort by start, chains, new scores into recs
template <typename R>
static bool pseudo_on_records(R *recs, uint64_t n, bool from_groups, const kslam_read_pair *groups, uint64_t n_groups,
                              PairWork &W, SortWorkspace &sortws, hipStream_t s) {
  W.sort_a.ensure((n + 1) * sizeof(uint4));
  W.sort_b.ensure((n + 1) * sizeof(uint4));
  W.flags.ensure((n + 1) * sizeof(uint32_t));
  W.gpos.ensure((n + 1) * sizeof(uint32_t));
  W.count.ensure((n + 2) * sizeof(uint32_t));          // run starts
  W.scan_tmp.ensure(scan_tmp_bytes(n));
  W.totals.ensure(16 * sizeof(uint64_t));
  uint64_t *tot = W.totals.as<uint64_t>();
  HIPCHK(hipMemsetAsync(tot + 12, 0, 2 * sizeof(uint64_t), s));
  uint32_t *d_max_entry = reinterpret_cast<uint32_t *>(tot + 12), *d_longest = reinterpret_cast<uint32_t *>(tot + 13);
  if constexpr (std::is_same<R, Rec>::value) {
    if (from_groups)
      hipLaunchKernelGGL(k_spans, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, s, recs, groups, n_groups,
                         W.sort_a.as<uint4>(), d_max_entry);
  }
  if (!from_groups)
    hipLaunchKernelGGL(k_spans_flat<R>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const R *)recs, n, W.sort_a.as<uint4>(), d_max_entry);
  uint32_t max_entry = 0;
  read_back(&max_entry, d_max_entry, sizeof max_entry, s);
  uint32_t bits = 1;
  while (bits < 32 && (max_entry >> bits)) bits++;
  SortPass passes[4];
  int np = 0;
  for (uint32_t b = 0; b < (bits + 7) / 8; b++) passes[np++] = SortPass{0, 8 * b, 0};
  const uint4 *sorted = (const uint4 *)radix_sort(W.sort_a.p, W.sort_b.p, n, 4, passes, np, sortws, s, nullptr, nullptr, nullptr);
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_entry_runs, dim3(nb), dim3(256), 0, s, sorted, n, W.flags.as<uint32_t>());
  exclusive_scan_u32(W.flags.as<uint32_t>(), W.gpos.as<uint32_t>(), n, tot + 14, W.scan_tmp.p, s);
  uint64_t n_runs64 = 0;
  read_back(&n_runs64, tot + 14, sizeof n_runs64, s);
  const uint32_t n_runs = (uint32_t)n_runs64;
  hipLaunchKernelGGL(k_run_starts, dim3(nb), dim3(256), 0, s, W.flags.as<uint32_t>(), W.gpos.as<uint32_t>(), n,
                     W.count.as<uint32_t>());
  hipLaunchKernelGGL(k_run_longest, dim3((n_runs + 255) / 256), dim3(256), 0, s, W.count.as<uint32_t>(), n_runs, (uint32_t)n, d_longest);
  uint32_t longest = 0;
  read_back(&longest, d_longest, sizeof longest, s);
  if (longest > (W.pseudo_cap ? W.pseudo_cap : PSEUDO_CAP_GLOBAL)) return false;
  uint4 *work = const_cast<uint4 *>(sorted);
  hipLaunchKernelGGL((k_pseudo_entry<false, R>), dim3(n_runs), dim3(64), (size_t)std::min(longest, PSEUDO_CAP) * sizeof(PSpan), s, work,
                     W.count.as<uint32_t>(), n_runs, (uint32_t)n, recs);
  if (longest > PSEUDO_CAP)   // entries too big for LDS: the same wave algorithm on the spans where they lie
    hipLaunchKernelGGL((k_pseudo_entry<true, R>), dim3(n_runs), dim3(64), 0, s, work, W.count.as<uint32_t>(), n_runs, (uint32_t)n, recs);
  HIPCHK(hipGetLastError());
  return true;
}

bool pseudo_and_rescreen(PairWork &W, PairResult *res, double score_fraction, SortWorkspace &sortws, hipStream_t s) {
  const uint64_t n = res->n_pairs, n_groups = res->n_read_pairs;
  if (n == 0) { res->stages_done |= 4u; return true; }
  if (n >= (1ull << 28)) return false;
  Rec *recs = const_cast<Rec *>(res->d_pairs);
  kslam_read_pair *groups = const_cast<kslam_read_pair *>(res->d_groups);
  if (!pseudo_on_records(recs, n, true, groups, n_groups, W, sortws, s)) return false;
  hipLaunchKernelGGL(k_rescreen, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, s, recs, groups, n_groups, score_fraction);
  HIPCHK(hipGetLastError());
  res->stages_done |= 4u;
  return true;
}

void debug_wave_sort(const int32_t *keys, const uint64_t *seg_off, uint64_t n_seg, uint32_t *perm, hipStream_t s) {
  if (!n_seg) return;
  const uint64_t n = seg_off[n_seg];
  uint32_t longest = 0;
  for (uint64_t i = 0; i < n_seg; i++) {
    if (seg_off[i + 1] < seg_off[i] || seg_off[i + 1] - seg_off[i] > PSEUDO_CAP_GLOBAL)
      throw StatusError{KSLAM_ERR_ARG, "segments must be ascending and hold at most 262144 keys each"};
    longest = std::max<uint32_t>(longest, (uint32_t)(seg_off[i + 1] - seg_off[i]));
  }
  DevBuf dk, doff, dp, dscratch;
  dscratch.ensure((n + 1) * sizeof(PSpan));
  dk.ensure((n + 1) * sizeof(int32_t));
  doff.ensure((n_seg + 1) * sizeof(uint64_t));
  dp.ensure((n + 1) * sizeof(uint32_t));
  HIPCHK(hipMemcpyAsync(dk.p, keys, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(doff.p, seg_off, (n_seg + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_debug_wave_sort, dim3((unsigned)n_seg), dim3(64), (size_t)std::min(longest, PSEUDO_CAP) * sizeof(PSpan), s,
                     dk.as<int32_t>(), doff.as<uint64_t>(), dp.as<uint32_t>(), dscratch.as<PSpan>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(perm, dp.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIPCHK(stream_wait(s));
}

// getMaxAllowedInsertSize on the device-resident insert sizes (host/tail.cpp: max_allowed_insert)
static uint32_t max_allowed_insert_device(int32_t *d_ins, uint64_t n, PairWork &W, SortWorkspace &sortws, hipStream_t s) {
  if (!n) return 0xFFFFFFFFu;
  W.sort_a.ensure((n + 1) * sizeof(uint2));
  W.sort_b.ensure((n + 1) * sizeof(uint2));
  hipLaunchKernelGGL(k_widen, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_ins, n, W.sort_a.as<uint2>());
  SortPass passes[4];
  for (uint32_t b = 0; b < 4; b++) passes[b] = SortPass{0, 8 * b, 0x80000000u};   // signed order
  const uint2 *sorted = (const uint2 *)radix_sort(W.sort_a.p, W.sort_b.p, n, 2, passes, 4, sortws, s, nullptr, nullptr, nullptr);
  // the percentile ladder, the quartiles: 102 reads by index (indices as the reference computes them, in double)
  uint64_t idx[102];
  for (int i = 0; i < 100; i++) idx[i] = (uint64_t)std::floor(n * (i) / 100.0);
  idx[100] = (uint64_t)std::floor(n * 0.25);
  idx[101] = (uint64_t)std::floor(n * 0.75);
  // (floor(n * (i + 1) / 100.0) for i = 98 is index 99's value: idx[99]; index 100 = n would be past the end
  // and the reference never reads it: its loop stops at i = 98)
  W.idx.ensure(sizeof idx);
  W.picked.ensure(102 * sizeof(int32_t) + 8 * sizeof(long long));
  HIPCHK(hipMemcpyAsync(W.idx.p, idx, sizeof idx, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_pick, dim3(1), dim3(128), 0, s, sorted, W.idx.as<uint64_t>(), 102u, W.picked.as<int32_t>());
  int32_t v[102];
  HIPCHK(hipMemcpyAsync(v, W.picked.p, sizeof v, hipMemcpyDeviceToHost, s));
  HIPCHK(stream_wait(s));
  int32_t limit = 0;
  for (int i = 0; i < 99; i++)
    if (v[i + 1] - v[i] > 1000) {
      limit = v[i];   // sz[floor(n * i / 100)]: integer division in the reference, same index for n < 2^53
      break;
    }
  const int32_t lq = v[100], uq = v[101];
  const int32_t lo = 0;
  int32_t hi = uq + 2 * (uq - lq);
  if (limit) hi = limit;
  if (hi == 0) hi = INT32_MAX;
  long long *d_sums = reinterpret_cast<long long *>(W.picked.as<int32_t>() + 104);
  HIPCHK(hipMemsetAsync(d_sums, 0, 4 * sizeof(long long), s));
  const unsigned nb = (unsigned)std::min<uint64_t>((n + 255) / 256, 256);
  hipLaunchKernelGGL(k_insert_sums, dim3(nb), dim3(256), 0, s, sorted, n, lo, hi, d_sums);
  long long h[4];
  HIPCHK(hipMemcpyAsync(h, d_sums, sizeof h, hipMemcpyDeviceToHost, s));
  HIPCHK(stream_wait(s));
  const long long t1 = h[0], t2 = h[1], tm = h[2], kept = h[3];
  double sum, sq;
  if (tm < (1ll << 53) && std::llabs(t1) < (1ll << 53)) {
    sum = (double)t1;   // every partial sum of the reference's sequential accumulation is exact too
    sq = (double)t2;
  } else {
    // beyond 2^53 the reference's result depends on its order of additions: do them in that order
    std::vector<uint2> hs(n);
    HIPCHK(hipMemcpyAsync(hs.data(), sorted, n * sizeof(uint2), hipMemcpyDeviceToHost, s));
    HIPCHK(stream_wait(s));
    sum = 0;
    sq = 0;
    for (uint64_t i = 0; i < n; i++) {
      const int32_t x = (int32_t)hs[i].x;
      if (x < lo || x > hi) continue;
      sum += x;
      sq = sq + (int32_t)((uint32_t)x * (uint32_t)x);
    }
  }
  const double mean = sum / kept;
  const double sd = std::sqrt(sq / kept - mean * mean);
  const double r = std::floor(mean + 6 * sd);
  return std::isnan(r) ? 0xFFFFFFFFu : (uint32_t)r;
}

// Phase A of pair_and_screen: score screen + pairing per read pair (k_pair); leaves the alignment pairs in their
// per-read-pair regions and the batch's non-zero insert sizes in W.inserts (res->n_insert_sizes of them).
void pair_phase_a(const kslam_overlap *d_ov, uint64_t n, const uint32_t *d_read_len, uint64_t n_reads, int paired,
                  uint32_t score_threshold, PairWork &W, PairResult *res, hipStream_t s) {
  memset(res, 0, sizeof *res);
  res->max_insert_size = 0xFFFFFFFFu;
  const uint64_t units = paired ? n_reads / 2 : n_reads, mid = n_reads / 2;
  W.units = units; W.mid = mid; W.paired = paired;
  if (units == 0) return;
  W.recs.ensure((4 * n + 4) * sizeof(Rec));
  W.count.ensure((units + 1) * sizeof(uint32_t));
  W.base.ensure((units + 1) * sizeof(uint64_t));
  W.inserts.ensure((2 * n + 2) * sizeof(int32_t));
  W.flags.ensure((units + 1) * sizeof(uint32_t));
  W.gpos.ensure((units + 1) * sizeof(uint32_t));
  W.rpos.ensure((units + 1) * sizeof(uint64_t));
  W.scan_tmp.ensure(scan_tmp_bytes(units));
  W.totals.ensure(16 * sizeof(uint64_t));
  uint64_t *tot = W.totals.as<uint64_t>();
  HIPCHK(hipMemsetAsync(tot, 0, 16 * sizeof(uint64_t), s));
  // tot[4] inserts, [5] kept, [6] pairs after pairing
  if (n >= (1ull << 32)) throw StatusError{KSLAM_ERR_UNSUPPORTED, ">= 2^32 overlap records in one batch"};
  W.row_start.ensure((n_reads + 2) * sizeof(uint32_t));
  const uint32_t gap_cap = (uint32_t)(n_reads / 64 + 2);   // stretches of more than 64 reads without rows: at most that many
  W.gaps.ensure((size_t)gap_cap * sizeof(uint4));
  uint32_t *d_ngaps = reinterpret_cast<uint32_t *>(tot + 11);
  HIPCHK(hipMemsetAsync(W.row_start.p, 0, (n_reads + 2) * sizeof(uint32_t), s));   // (unsorted rows leave holes: zeros, not stale numbers)
  uint32_t *d_bad = reinterpret_cast<uint32_t *>(tot + 7);
  hipLaunchKernelGGL(k_row_starts, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, s, d_ov, n, n_reads, W.row_start.as<uint32_t>(),
                     W.gaps.as<uint4>(), gap_cap, d_ngaps, d_bad);
  hipLaunchKernelGGL(k_fill_gaps, dim3(64), dim3(256), 0, s, W.gaps.as<uint4>(), d_ngaps, gap_cap, W.row_start.as<uint32_t>());
  PairArgs a;
  a.ov = d_ov; a.n = n; a.row_start = W.row_start.as<uint32_t>(); a.read_len = d_read_len; a.units = units; a.mid = mid;
  a.thr = score_threshold; a.paired = paired;
  a.recs = W.recs.as<Rec>(); a.count = W.count.as<uint32_t>(); a.base = W.base.as<uint64_t>();
  a.inserts = W.inserts.as<int32_t>();
  a.n_inserts = reinterpret_cast<unsigned long long *>(tot + 4);
  a.n_kept = reinterpret_cast<unsigned long long *>(tot + 5);
  a.n_initial = reinterpret_cast<unsigned long long *>(tot + 6);
  // read pairs with very many rows: listed by k_pair, done by k_pair_big (a fixed grid walks the list: no count comes back)
  W.picked.ensure((units + 1) * sizeof(uint32_t));
  a.big_list = paired ? W.picked.as<uint32_t>() : nullptr;
  a.n_big = reinterpret_cast<uint32_t *>(tot + 3);
  const unsigned nb = (unsigned)((units + 255) / 256);
  hipLaunchKernelGGL(k_pair, dim3(nb), dim3(256), 0, s, a);
  if (paired) hipLaunchKernelGGL(k_pair_big, dim3(2048), dim3(64), 0, s, a);
  uint64_t h[4];
  HIPCHK(hipMemcpyAsync(h, tot + 4, sizeof h, hipMemcpyDeviceToHost, s));
  HIPCHK(stream_wait(s));
  if (h[3]) throw StatusError{KSLAM_ERR_ARG, "the overlap records are not sorted by read (or name reads the batch does not have)"};
  res->n_insert_sizes = h[0];
  res->n_overlaps_screened = h[1];
  res->n_paired_initial = h[2];
}

// getMaxAllowedInsertSize (src/PairedOverlap.h:314-360) of any device array of insert sizes: a batch's own
// (W.inserts), or the insert sizes of all shards of a batch gathered by the caller (the statistics are those of the
// sorted values: where each value came from does not matter).  d_ins is used as scratch input only (not modified).
uint32_t insert_limit_device(const int32_t *d_ins, uint64_t n, PairWork &W, SortWorkspace &sortws, hipStream_t s) {
  return max_allowed_insert_device(const_cast<int32_t *>(d_ins), n, W, sortws, s);
}

// Phase B: the two per-read-pair screens with the given insert-size limit, then the dense groups / pairs arrays.
void pair_phase_b(const kslam_overlap *d_ov, uint32_t limit, double score_fraction, int do_insert, int do_score, PairWork &W,
                  PairResult *res, hipStream_t s) {
  const uint64_t units = W.units, mid = W.mid;
  const int paired = W.paired;
  if (units == 0) return;
  uint64_t *tot = W.totals.as<uint64_t>();
  const unsigned nb = (unsigned)((units + 255) / 256);
  if (do_insert && paired) res->max_insert_size = limit;
  else limit = 0xFFFFFFFFu;
  W.picked.ensure((units + 1) * sizeof(uint32_t));        // the big read pairs' numbers (free at this point of the stage)
  uint32_t *d_nbig = reinterpret_cast<uint32_t *>(tot + 10);
  HIPCHK(hipMemsetAsync(d_nbig, 0, sizeof(uint64_t), s));
  hipLaunchKernelGGL(k_screen, dim3(nb), dim3(256), 0, s, d_ov, W.recs.as<Rec>(), W.base.as<uint64_t>(),
                     W.count.as<uint32_t>(), units, (do_insert && paired) ? 1 : 0, limit, do_score ? 1 : 0, score_fraction,
                     W.flags.as<uint32_t>(), W.picked.as<uint32_t>(), d_nbig);
  uint32_t n_big = 0;
  read_back(&n_big, d_nbig, sizeof n_big, s);
  if (n_big)
    hipLaunchKernelGGL(k_screen_big, dim3(n_big), dim3(64), 0, s, d_ov, W.recs.as<Rec>(), W.base.as<uint64_t>(), W.count.as<uint32_t>(),
                       W.picked.as<uint32_t>(), (do_insert && paired) ? 1 : 0, limit, do_score ? 1 : 0, score_fraction,
                       W.flags.as<uint32_t>());
  exclusive_scan_u32(W.flags.as<uint32_t>(), W.gpos.as<uint32_t>(), units, tot + 8, W.scan_tmp.p, s);
  exclusive_scan_u32_to_u64(W.count.as<uint32_t>(), W.rpos.as<uint64_t>(), units, tot + 9, W.scan_tmp.p, s);
  uint64_t g2[2];
  HIPCHK(hipMemcpyAsync(g2, tot + 8, sizeof g2, hipMemcpyDeviceToHost, s));
  HIPCHK(stream_wait(s));
  res->n_read_pairs = g2[0];
  res->n_pairs = g2[1];
  W.groups.ensure((g2[0] + 1) * sizeof(kslam_read_pair));
  W.dense.ensure((g2[1] + 1) * sizeof(Rec));
  hipLaunchKernelGGL(k_emit_groups, dim3(nb), dim3(256), 0, s, W.recs.as<Rec>(), W.base.as<uint64_t>(), W.count.as<uint32_t>(),
                     W.gpos.as<uint32_t>(), W.rpos.as<uint64_t>(), units, mid, paired, W.groups.as<kslam_read_pair>(),
                     W.dense.as<Rec>());
  HIPCHK(hipGetLastError());
  res->d_groups = W.groups.as<kslam_read_pair>();
  res->d_pairs = W.dense.as<Rec>();
  res->stages_done = ((do_insert && paired) ? 1u : 0u) | (do_score ? 2u : 0u);
}

void pair_and_screen(const kslam_overlap *d_ov, uint64_t n, const uint32_t *d_read_len, uint64_t n_reads, int paired,
                     uint32_t score_threshold, double score_fraction, int do_insert, int do_score, PairWork &W,
                     SortWorkspace &sortws, PairResult *res, hipStream_t s) {
  pair_phase_a(d_ov, n, d_read_len, n_reads, paired, score_threshold, W, res, s);
  if (W.units == 0) return;
  uint32_t limit = 0xFFFFFFFFu;
  if (do_insert && paired) limit = max_allowed_insert_device(W.inserts.as<int32_t>(), res->n_insert_sizes, W, sortws, s);
  pair_phase_b(d_ov, limit, score_fraction, do_insert, do_score, W, res, s);
}

// ---- pseudo-assembly over the alignment pairs of SEVERAL shards -------------------------------------------------------
// pseudoAssembly works per entry across all read pairs of the batch (src/PairedOverlap.h:480-582).  When the read pairs
// are sharded over GPUs, every shard's dense alignment-pair records are gathered (rank order = read-pair order, which
// is the reference's bucket iteration order) into d_all[0 .. n_all); the stage then runs on that array exactly as
// pseudo_and_rescreen runs it on one batch's -- spans in record order, stable bucket sort by entry, std::sort
// permutation and chains per entry -- and leaves the new combined scores in d_all.  Returns false (nothing changed) when
// an entry holds more spans than one wavefront should sort: the caller runs the stage on the host.
bool pseudo_merged(PairWork &W, PairResult *res, void *d_all, uint64_t n_all, uint64_t own_base, double score_fraction,
                   SortWorkspace &sortws, hipStream_t s) {
  const uint64_t n_own = res->n_pairs, n_groups = res->n_read_pairs;
  if (own_base + n_own > n_all) throw StatusError{KSLAM_ERR_ARG, "this shard's records lie outside the gathered array"};
  if (n_all == 0) { res->stages_done |= 4u; return true; }
  if (n_all >= (1ull << 28)) return false;
  Rec *all = static_cast<Rec *>(d_all);
  if (!pseudo_on_records(all, n_all, false, nullptr, 0, W, sortws, s)) return false;
  if (n_own) {
    Rec *own = const_cast<Rec *>(res->d_pairs);
    hipLaunchKernelGGL(k_take_scores, dim3((unsigned)((n_own + 255) / 256)), dim3(256), 0, s, all, own_base, n_own, own);
    hipLaunchKernelGGL(k_rescreen, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, s, own,
                       const_cast<kslam_read_pair *>(res->d_groups), n_groups, score_fraction);
  }
  HIPCHK(hipGetLastError());
  res->stages_done |= 4u;
  return true;
}

// ---- pseudo-assembly with the ENTRIES partitioned over the ranks of a sharded batch ---------------------------------------
// pseudo_merged has every rank run the stage on every record of the batch (N x the work, N x the traffic of an all-gather
// of 32-byte records).  The stage is independent per entry (src/PairedOverlap.h:495-574: one bucket per entry, one walk per
// bucket), so entry e can belong to rank e mod N alone:
//   pseudo_route    this rank's records as 16-byte heads {score, entry, start, end}, stably partitioned by e mod N
//                   (one 8-bit radix pass over {destination, index}); counts[d] = heads bound for rank d
//   <all-to-all: rank d receives the pieces in SOURCE-RANK order = read-pair order, the reference's bucket iteration order>
//   pseudo_owned    the stage on the received heads (same kernels, R = RecHead) -> their new scores, 4 bytes each
//   <all-to-all back, same pieces, 4 bytes per record>
//   pseudo_return   scores into this rank's records through the partition's permutation, second score screen
// Per rank and batch: 16 + 4 bytes per record sent (N - 1) / N of the time, and 1 / N of the stage's work.
__global__ void k_route_keys(const Rec *__restrict__ recs, uint64_t n, uint32_t world, uint2 *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = make_uint2(recs[i].entry % world, (uint32_t)i);
}
__global__ void k_route_gather(const Rec *__restrict__ recs, const uint2 *__restrict__ sorted, uint64_t n, RecHead *__restrict__ heads,
                               unsigned long long *__restrict__ counts) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint2 k = sorted[i];
  heads[i] = *reinterpret_cast<const RecHead *>(recs + k.y);
  if (i + 1 == n || sorted[i + 1].x != k.x) atomicAdd(counts + k.x, (unsigned long long)(i + 1));   // end of destination k.x's run
  if (i + 1 < n && sorted[i + 1].x != k.x) atomicAdd(counts + sorted[i + 1].x, (unsigned long long)0 - (i + 1));   // start of the next
}
__global__ void k_head_scores(const RecHead *__restrict__ heads, uint64_t n, uint32_t *__restrict__ scores) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) scores[i] = heads[i].combined_score;
}
__global__ void k_scores_home(const uint32_t *__restrict__ scores, const uint2 *__restrict__ sorted, uint64_t n, Rec *__restrict__ own) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) own[sorted[i].y].combined_score = scores[i];
}

void pseudo_route(PairWork &W, const PairResult *res, uint32_t world, const void **d_heads, uint64_t *counts, SortWorkspace &sortws,
                  hipStream_t s) {
  const uint64_t n = res->n_pairs;
  if (world == 0 || world > 256) throw StatusError{KSLAM_ERR_ARG, "1 to 256 ranks"};   // one 8-bit radix pass
  if (n >= (1ull << 32)) throw StatusError{KSLAM_ERR_UNSUPPORTED, ">= 2^32 alignment pairs on one rank"};
  for (uint32_t d = 0; d < world; d++) counts[d] = 0;
  W.route_n = n;
  W.route_world = world;
  *d_heads = nullptr;
  if (n == 0) return;
  W.route_a.ensure((n + 1) * sizeof(uint2));
  W.route_b.ensure((n + 1) * sizeof(uint2));
  W.route_heads.ensure((n + 1) * sizeof(RecHead));
  W.totals.ensure(16 * sizeof(uint64_t));
  W.route_counts.ensure(256 * sizeof(uint64_t));
  HIPCHK(hipMemsetAsync(W.route_counts.p, 0, 256 * sizeof(uint64_t), s));
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_route_keys, dim3(nb), dim3(256), 0, s, res->d_pairs, n, world, W.route_a.as<uint2>());
  const SortPass pass{0, 0, 0};
  const uint2 *sorted = (const uint2 *)radix_sort(W.route_a.p, W.route_b.p, n, 2, &pass, 1, sortws, s, nullptr, nullptr, nullptr);
  W.route_sorted = sorted;
  hipLaunchKernelGGL(k_route_gather, dim3(nb), dim3(256), 0, s, res->d_pairs, sorted, n, W.route_heads.as<RecHead>(),
                     W.route_counts.as<unsigned long long>());
  HIPCHK(hipGetLastError());
  for (uint32_t d = 0; d < world; d += 32)   // read_back moves up to 256 bytes
    read_back(counts + d, W.route_counts.as<uint64_t>() + d, std::min(32u, world - d) * sizeof(uint64_t), s);
  *d_heads = W.route_heads.p;
}

// the stage on heads received from all ranks (source-rank order); *d_scores = their scores afterwards, in the same order.
// false: declined (an entry too large for one wavefront, or 2^28 or more heads), nothing usable was produced.
bool pseudo_owned(PairWork &W, void *d_heads, uint64_t n, const uint32_t **d_scores, SortWorkspace &sortws, hipStream_t s) {
  *d_scores = nullptr;
  if (n == 0) return true;
  if (n >= (1ull << 28)) return false;
  RecHead *heads = static_cast<RecHead *>(d_heads);
  if (!pseudo_on_records<RecHead>(heads, n, false, nullptr, 0, W, sortws, s)) return false;
  W.route_scores.ensure((n + 1) * sizeof(uint32_t));
  hipLaunchKernelGGL(k_head_scores, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, heads, n, W.route_scores.as<uint32_t>());
  HIPCHK(hipGetLastError());
  *d_scores = W.route_scores.as<uint32_t>();
  return true;
}

void pseudo_return(PairWork &W, PairResult *res, const uint32_t *d_scores, uint64_t n, double score_fraction, hipStream_t s) {
  if (n != res->n_pairs || n != W.route_n) throw StatusError{KSLAM_ERR_ARG, "as many scores as kslam_pseudo_route sent heads"};
  if (n) {
    Rec *own = const_cast<Rec *>(res->d_pairs);
    const uint64_t n_groups = res->n_read_pairs;
    hipLaunchKernelGGL(k_scores_home, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_scores, (const uint2 *)W.route_sorted, n, own);
    hipLaunchKernelGGL(k_rescreen, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, s, own,
                       const_cast<kslam_read_pair *>(res->d_groups), n_groups, score_fraction);
    HIPCHK(hipGetLastError());
  }
  W.route_n = ~0ull;
  res->stages_done |= 4u;
}

}  // namespace kslam
