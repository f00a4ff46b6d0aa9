// join.hip -- sorted read k-mers x resident sorted genome k-mers -> candidate
// overlaps, then the reference's sort + "unique within 3" dedupe.
//
// Replaces findOverlaps / processPileUp / findOverlaps_parallel (reference
// src/Overlap.h:153-199, 230-246, 277-295).  Semantics kept:
//   * k-mer value 0 never joins (Overlap.h:236-239);
//   * a run of equal k-mers yields |genome records| x |read records| overlaps
//     (Overlap.h:163-197); runs without a genome record yield nothing (:157);
//   * off = g.rc ? L_read - r.offset - 32 : r.offset; rel = int32(g.offset - off);
//     revComp = (g.rc != r.rc)                                   (Overlap.h:177-193)
//   * overlaps sorted by (read, entry, rel) (Overlap.h:87-98); then std::unique
//     with "same read & entry & |delta rel| < 3" against the LAST KEPT element
//     (Overlap.h:79-85, 290).  revComp is appended as the least significant
//     key bit (false first) so the order is total; the reference leaves such
//     ties to an unstable sort.
//
// MI355X design: the reference re-sorts reads + genomes together every batch;
// here the genome list is sorted once and stays in HBM (a key column and a
// {meta, offset} column) with a 2^b-entry bucket table over the top key bits.
// After the membership filter one read k-mer in ten is left (25.3 M against
// 312 M genome keys per configs[1] batch).  Two ways to meet the two lists, both
// built, both measured on that batch (profiles/r06_join_merge.json):
//   * THE PROBE (k_join_fill, the default): every sorted survivor reads its two
//     table bounds, the bucket's handful of keys in one round trip, and one
//     8-byte {meta, offset} per hit.  0.98 ms; 46.8 M 64-byte requests = 2.92 GB
//     by FETCH_SIZE -- MORE than the 2.5 GB key column it declines to stream: the
//     kernel is bound by scattered sectors (3.0 TB/s of them), VALU 0.07.
//   * THE MERGE (k_join_merge, KSLAM_JOIN=merge): the shape of the reference's
//     findOverlaps (src/Overlap.h:230-246), cut into one segment per workgroup: the
//     key range a tile of read records meets is streamed through LDS once, 16 bytes
//     per lane.  0.93 ms on configs[1], 1.04 against 1.00 on the repeat-rich
//     database, 1.71 against 1.69 per chunk on configs[2]: the stream is cheap
//     (2.5 GB sequential) but the {meta, offset} gathers of 23 M hits remain, and
//     they are half of the probe's sectors.  A wash within 5 % either way, and the
//     probe does not care how sparse a batch is (the merge streams the column even
//     for a handful of reads): the probe stays the default.
// An AoS index ({key, meta, offset} in 16 bytes, one gather instead of two) was
// priced and not built: keys and payloads of neighbouring buckets share sectors in
// the two columns more often than a 16-byte record shares its sector with the next
// probe's (42 M against 46.8 M requests expected: -10 %).
// Output is a packed u64 per overlap (read | entry | rel + bias | revcomp) so that
// the overlap sort is a keys-only radix sort over just the populated bytes.
#include "common.h"

namespace kslam {

namespace {

constexpr int JB = 256;                   // threads per block
constexpr int JI = JOIN_TILE / JB;        // records per thread
constexpr uint32_t BIG = 48;              // runs longer than this are expanded by the whole block
constexpr int BIGQ = 64;

__global__ void k_bucket(const uint64_t *__restrict__ keys, uint32_t n, uint32_t sh, uint32_t nb,
                         uint32_t *__restrict__ bucket) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t b = (uint32_t)(keys[i] >> sh);
  int64_t bp = i > 0 ? (int64_t)(keys[i - 1] >> sh) : -1;
  for (int64_t x = bp + 1; x <= (int64_t)b; x++) bucket[x] = i;
  if (i == n - 1)
    for (uint32_t x = b + 1; x <= nb; x++) bucket[x] = n;
}
__global__ void k_bucket_empty(uint32_t nb, uint32_t *bucket) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= nb) bucket[i] = 0;
}

struct Run {
  uint32_t lo, cnt;
};

__device__ inline Run find_run(uint64_t key, const GenomeIndexDev &g) {
  Run r{0, 0};
  if (key == 0) return r;  // Overlap.h:236
  const uint32_t b = (uint32_t)(key >> (64 - g.bucket_bits));
  uint32_t lo = g.bucket[b], hi = g.bucket[b + 1];
  const uint32_t end = hi;
  if (hi - lo <= 16u) {
    // The usual bucket holds a handful of keys: fetch them all at once (independent loads, one
    // round trip) and count instead of chasing a binary search through ten dependent loads.
    const uint32_t nb = hi - lo;
    uint32_t lt = 0, le = 0;
#pragma unroll
    for (uint32_t half = 0; half < 2; half++) {
      if (half * 8u < nb) {
        uint64_t k[8];
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) k[i] = half * 8u + i < nb ? g.key[lo + half * 8u + i] : ~0ull;
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
          const bool in = half * 8u + i < nb;
          lt += (in && k[i] < key) ? 1u : 0u;
          le += (in && k[i] <= key) ? 1u : 0u;
        }
      }
    }
    r.lo = lo + lt;
    r.cnt = le - lt;
    return r;
  }
  while (lo < hi) {  // lower_bound
    uint32_t mid = lo + ((hi - lo) >> 1);
    if (g.key[mid] < key) lo = mid + 1; else hi = mid;
  }
  if (lo >= end || g.key[lo] != key) return r;
  uint32_t a = lo + 1, z = end;
  while (a < z) {  // upper_bound
    uint32_t mid = a + ((z - a) >> 1);
    if (g.key[mid] <= key) a = mid + 1; else z = mid;
  }
  r.lo = lo;
  r.cnt = a - lo;
  return r;
}

__device__ inline uint64_t make_overlap(uint32_t rmeta, uint32_t roff, uint32_t gmeta, uint32_t goff,
                                        const uint32_t *read_len, const OverlapKeyLayout &lay) {
  const uint32_t rid = rmeta & 0x3FFFFFFFu, gid = gmeta & 0x3FFFFFFFu;
  const uint32_t rrc = (rmeta >> 30) & 1u, grc = (gmeta >> 30) & 1u;
  const uint32_t off = grc ? (read_len[rid] - roff - KSLAM_K) : roff;  // Overlap.h:179-183
  const int32_t rel = (int32_t)(goff - off);                            // Overlap.h:186
  const uint64_t relb = (uint64_t)(uint32_t)(rel + (int32_t)lay.rel_bias);
  return ((uint64_t)rid << (lay.bits_entry + lay.bits_rel + 1)) | ((uint64_t)gid << (lay.bits_rel + 1)) |
         (relb << 1) | (uint64_t)(grc != rrc);
}

// Single pass: the workgroup reserves its output range with one atomic add on cursor[0] and skips
// its writes when the range would exceed `cap` (the host then reruns with a larger buffer); the
// output ORDER depends on scheduling, which is harmless because the overlap keys are totally ordered
// by the sort that follows (equal keys are indistinguishable).
//
// flat emission: the block's runs {rmeta, roff, run.lo, first output slot} and, per output slot,
// the run it belongs to -- so that every thread then writes ONE overlap (parallel gathers of the
// genome records, consecutive stores) instead of walking its own runs with the wave idling
constexpr uint32_t FLAT_MAX = 4096;
struct JoinSmem {
  uint32_t wsum[JB / 64];
  unsigned long long s_base;
  uint32_t bigq_n;
  uint4 bigq[BIGQ];          // {rmeta, roff, run.lo, run.cnt}
  uint32_t bigq_out[BIGQ];   // block-relative output offset
  uint4 runq[JOIN_TILE];
  uint16_t owner[FLAT_MAX];
  uint32_t runq_n;
};

// what both join kernels do once every thread knows the run of equal genome keys of each of its JI read records
// (sm.bigq_n and sm.runq_n zeroed by thread 0 before the first barrier the caller passed)
__device__ __forceinline__ void emit_runs(JoinSmem &sm, const uint4 (&r)[JI], const Run (&run)[JI], uint32_t mine, const GenomeIndexDev &g,
                                          const uint32_t *__restrict__ read_len, unsigned long long *__restrict__ cursor, uint64_t cap,
                                          const OverlapKeyLayout &lay, uint64_t *__restrict__ out) {
  // block exclusive scan of `mine`
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) sm.wsum[w] = inc;
  __syncthreads();
  uint32_t ex = inc - mine;
  for (int i = 0; i < w; i++) ex += sm.wsum[i];
  uint64_t bb;
  uint32_t tot = 0;
  for (int i = 0; i < JB / 64; i++) tot += sm.wsum[i];
  if (threadIdx.x == 0) sm.s_base = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull;
  __syncthreads();
  bb = sm.s_base;
  if (bb + tot > cap) return;   // does not fit: only the cursor matters now (block-uniform exit)
  if (tot <= FLAT_MAX) {   // (block-uniform)
#pragma unroll
    for (int it = 0; it < JI; it++) {
      const uint32_t c = run[it].cnt;
      if (c == 0) continue;
      const uint32_t slot = atomicAdd(&sm.runq_n, 1u);
      sm.runq[slot] = make_uint4(r[it].z, r[it].w, run[it].lo, ex);
      for (uint32_t j = 0; j < c; j++) sm.owner[ex + j] = (uint16_t)slot;
      ex += c;
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < tot; k += JB) {
      const uint4 e = sm.runq[sm.owner[k]];
      const uint32_t gi = e.z + (k - e.w);
      const uint2 mo = g.mo[gi];
      out[bb + k] = make_overlap(e.x, e.y, mo.x, mo.y, read_len, lay);
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < JI; it++) {
    const uint32_t c = run[it].cnt;
    if (c == 0) continue;
    bool queued = false;
    if (c > BIG) {
      uint32_t slot = atomicAdd(&sm.bigq_n, 1u);
      if (slot < BIGQ) {
        sm.bigq[slot] = make_uint4(r[it].z, r[it].w, run[it].lo, c);
        sm.bigq_out[slot] = ex;
        queued = true;
      }
    }
    if (!queued) {
      for (uint32_t j = 0; j < c; j++) {
        const uint32_t gi = run[it].lo + j;
        const uint2 mo = g.mo[gi];
        out[bb + ex + j] = make_overlap(r[it].z, r[it].w, mo.x, mo.y, read_len, lay);
      }
    }
    ex += c;
  }
  __syncthreads();
  const uint32_t nq = min(sm.bigq_n, (uint32_t)BIGQ);
  for (uint32_t q = 0; q < nq; q++) {
    const uint4 e = sm.bigq[q];
    const uint64_t ob = bb + sm.bigq_out[q];
    for (uint32_t j = threadIdx.x; j < e.w; j += JB) {
      const uint32_t gi = e.z + j;
      const uint2 mo = g.mo[gi];
      out[ob + j] = make_overlap(e.x, e.y, mo.x, mo.y, read_len, lay);
    }
  }
}

// THE PROBE (default): every sorted read record looks its key up through the bucket table
__global__ __launch_bounds__(JB) void k_join_fill(const uint4 *__restrict__ recs, uint32_t n,
                                                  GenomeIndexDev g, const uint32_t *__restrict__ read_len,
                                                  unsigned long long *__restrict__ cursor, uint64_t cap,
                                                  OverlapKeyLayout lay, uint64_t *__restrict__ out) {
  __shared__ JoinSmem sm;
  const uint32_t base = blockIdx.x * JOIN_TILE;
  if (threadIdx.x == 0) {
    sm.bigq_n = 0;
    sm.runq_n = 0;
  }
  uint4 r[JI];
  Run run[JI];
  uint32_t mine = 0;
#pragma unroll
  for (int it = 0; it < JI; it++) {
    uint32_t i = base + it * JB + threadIdx.x;
    run[it] = Run{0, 0};
    if (i < n) {
      r[it] = recs[i];
      run[it] = find_run(((uint64_t)r[it].y << 32) | r[it].x, g);
    }
    mine += run[it].cnt;
  }
  emit_runs(sm, r, run, mine, g, read_len, cursor, cap, lay, out);
}

// THE MERGE (KSLAM_JOIN=merge): the shape of the reference's findOverlaps (src/Overlap.h:230-246) -- two sorted lists walked
// side by side -- cut into segments of one workgroup each.  The read records are ordered by the top `gb` bits of their key
// (the radix passes the per-batch sort executes), so a tile of JOIN_TILE consecutive read records meets ONE contiguous range of
// the sorted genome key column, [bucket(first record's group), bucket(last record's group + 1)) -- two table reads per
// WORKGROUP instead of two per record.  The workgroup streams that range through LDS in pieces of MP keys with 16-byte
// coalesced loads (the next piece is in flight in registers while the current one is searched); every thread places each of
// its JI read keys in the piece by a binary search in LDS (the JI searches in lock step: independent LDS reads), and the run
// of equal keys is counted where it lies (a run cut by a piece boundary continues in the next piece: the column is sorted, the
// parts are adjacent).  Key 0 never joins (Overlap.h:236).  Then the same flat emission as the probe: {meta, offset} is
// gathered only for hits.  HBM: the key column once, sequentially (2.5 GB for the 5 Gb database whatever the batch), against the
// probe's scattered 64-byte sectors (table rows + keys: ~2 per record); measured side by side in profiles/r06_join_merge.json.
constexpr uint32_t MP = 4096;                    // keys per piece: 32 KB of LDS
constexpr uint32_t MPV = MP / 2 / JB;            // 16-byte loads per thread and piece
__global__ __launch_bounds__(JB) void k_join_merge(const uint4 *__restrict__ recs, uint32_t n, GenomeIndexDev g, uint32_t gb,
                                                   const uint32_t *__restrict__ read_len, unsigned long long *__restrict__ cursor,
                                                   uint64_t cap, OverlapKeyLayout lay, uint64_t *__restrict__ out) {
  // the staged piece and the emission's queues are never live together: one 32 KB block of LDS (5 workgroups per CU)
  static_assert(sizeof(JoinSmem) <= MP * sizeof(uint64_t), "the emission's LDS must fit under the piece");
  __shared__ __attribute__((aligned(16))) unsigned char raw[MP * sizeof(uint64_t)];
  __shared__ uint32_t s_range[2];
  JoinSmem &sm = *reinterpret_cast<JoinSmem *>(raw);
  uint64_t *piece = reinterpret_cast<uint64_t *>(raw);
  const uint32_t base = blockIdx.x * JOIN_TILE;
  const uint32_t tile_end = min(n, base + (uint32_t)JOIN_TILE);
  if (threadIdx.x == 0) {
    const uint4 a = recs[base], z = recs[tile_end - 1];
    const uint64_t ka = ((uint64_t)a.y << 32) | a.x, kz = ((uint64_t)z.y << 32) | z.x;
    const uint32_t up = g.bucket_bits - gb;
    s_range[0] = g.bucket[(uint32_t)(ka >> (64 - gb)) << up] & ~1u;        // (16-byte loads: an even start)
    s_range[1] = g.bucket[((uint32_t)(kz >> (64 - gb)) + 1u) << up];
  }
  uint4 r[JI];
  uint64_t key[JI];
  Run run[JI];
#pragma unroll
  for (int it = 0; it < JI; it++) {
    const uint32_t i = base + it * JB + threadIdx.x;
    run[it] = Run{0, 0};
    key[it] = 0;
    if (i < n) {
      r[it] = recs[i];
      key[it] = ((uint64_t)r[it].y << 32) | r[it].x;
    }
  }
  __syncthreads();
  const uint32_t glo = s_range[0], ghi = s_range[1];
  const ulonglong2 *col = reinterpret_cast<const ulonglong2 *>(g.key);
  const uint32_t n_even = (g.n + 1u) & ~1u;       // the column's allocation is padded (kslam_api.hip: n_gk + 2 keys)
  ulonglong2 nxt[MPV];
  auto fetch = [&](uint32_t p) {
#pragma unroll
    for (uint32_t v = 0; v < MPV; v++) {
      const uint32_t at = p + 2u * (v * JB + threadIdx.x);
      nxt[v] = at < ghi && at < n_even ? col[at >> 1] : make_ulonglong2(~0ull, ~0ull);
    }
  };
  if (glo < ghi) fetch(glo);
  for (uint32_t p = glo; p < ghi; p += MP) {
    const uint32_t cnt = min((uint32_t)MP, ghi - p);
#pragma unroll
    for (uint32_t v = 0; v < MPV; v++) reinterpret_cast<ulonglong2 *>(piece)[v * JB + threadIdx.x] = nxt[v];
    __syncthreads();
    if (p + MP < ghi) fetch(p + MP);                 // in flight while this piece is searched
    const uint64_t first = piece[0], last = piece[cnt - 1];
    // lower bounds of the JI keys in piece[0, cnt), in lock step
    uint32_t lo[JI], hi[JI];
    bool in[JI];
    bool any = false;
#pragma unroll
    for (int it = 0; it < JI; it++) {
      in[it] = key[it] != 0 && key[it] >= first && key[it] <= last;
      lo[it] = 0;
      hi[it] = in[it] ? cnt : 0;
      any |= in[it];
    }
    if (__any(any)) {
      for (uint32_t step = 0; step < 13; step++) {   // 2^12 = MP
        bool more = false;
#pragma unroll
        for (int it = 0; it < JI; it++) {
          if (lo[it] < hi[it]) {
            const uint32_t mid = lo[it] + ((hi[it] - lo[it]) >> 1);
            if (piece[mid] < key[it]) lo[it] = mid + 1; else hi[it] = mid;
            more |= lo[it] < hi[it];
          }
        }
        if (!__any(more)) break;
      }
#pragma unroll
      for (int it = 0; it < JI; it++) {
        if (!in[it]) continue;
        uint32_t e = lo[it];
        // the run's end: a few steps (runs are short), then a second binary search (tandem repeats: thousands)
        uint32_t c = 0;
        while (c < 4 && e + c < cnt && piece[e + c] == key[it]) c++;
        if (c == 4 && e + 4 < cnt && piece[e + 4] == key[it]) {
          uint32_t a = e + 5, z = cnt;
          while (a < z) {
            const uint32_t mid = a + ((z - a) >> 1);
            if (piece[mid] <= key[it]) a = mid + 1; else z = mid;
          }
          c = a - e;
        }
        // the padding of an odd start / the column's end reads as ~0 and a real key of ~0 cannot exist below index n
        if (c && p + e + c > g.n) c = g.n > p + e ? g.n - (p + e) : 0;
        if (c) {
          if (run[it].cnt == 0) run[it].lo = p + e;
          run[it].cnt += c;
        }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {      // (the loop's last barrier has passed: nobody reads the piece any more)
    sm.bigq_n = 0;
    sm.runq_n = 0;
  }
  uint32_t mine = 0;
#pragma unroll
  for (int it = 0; it < JI; it++) mine += run[it].cnt;
  emit_runs(sm, r, run, mine, g, read_len, cursor, cap, lay, out);
}

// std::unique with "same read & entry & |delta rel| < 3 against the LAST KEPT element" (Overlap.h:79-85, 290)
// is a greedy scan, but it only carries state across neighbours that are closer than 3: an element whose
// predecessor (same read and entry) lies 3 or more below it is kept whatever happened before -- the last
// kept element is at most that predecessor.  So the list falls into independent RUNS (chains of
// neighbours less than 3 apart) and one lane walks one run, not one whole (read, entry) segment: hits
// along a tandem repeat of period >= 3, thousands per segment, are all run heads and resolve in parallel;
// only dense runs (consecutive positions: homopolymers) are walked serially, and those are bounded by
// what one read can seed.
__global__ void k_dedupe_flags(const uint64_t *__restrict__ keys, uint64_t n, uint32_t seg_shift,
                               uint64_t rel_mask, uint32_t *__restrict__ flags) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys[i];
  const uint64_t seg = k >> seg_shift;
  const int64_t rel = (int64_t)((k >> 1) & rel_mask);
  if (i > 0) {
    const uint64_t kp = keys[i - 1];
    if ((kp >> seg_shift) == seg && rel - (int64_t)((kp >> 1) & rel_mask) < 3) return;  // inside a run: its head decides
  }
  int64_t last = rel, prev = rel;
  flags[i] = 1;
  for (uint64_t j = i + 1; j < n; j++) {
    const uint64_t kj = keys[j];
    if ((kj >> seg_shift) != seg) break;
    const int64_t rj = (int64_t)((kj >> 1) & rel_mask);
    if (rj - prev >= 3) break;           // the next run: its own head
    prev = rj;
    if (rj - last < 3) flags[j] = 0;     // Overlap.h:83 vs the last kept element
    else { flags[j] = 1; last = rj; }
  }
}

// The radix sort of the overlap keys would spend 3 of its 7 passes on the low bits -- rel and revComp -- which only order the
// keys INSIDE a (read, entry) group, and such a group is a handful of keys (3.2 on the bench workload: the k-mers a read
// shares with one locus).  So the keys are radix-sorted by the bits above them only (4 passes), and the groups are finished
// here.  A group of more than GROUP_CAP keys -- a read lying in a tandem repeat seeds hundreds of positions on one entry --
// is not: the flag tells the host to sort that chunk the long way (and the context to stop trying).
constexpr uint32_t GROUP_CAP = 64;
// One lane per GROUP: the lane whose key is the first of its group (its left neighbour has other high bits) finds the group's
// end, orders the group's keys by their low bits where they lie in LDS (insertion sort: 3 keys on average), and then walks the
// ordered group once for std::unique's rule -- keep a key iff its rel lies 3 or more above the last KEPT one (Overlap.h:79-85,
// 290; a group is one (read, entry), so the rule never looks across groups) -- writing the ordered keys and their flags.  So
// the separate flags kernel is not needed on this route.  A block owns the groups that START in its 256 keys; its LDS tile
// reaches GROUP_CAP + 1 keys further so that such a group is whole.  A longer group raises *big and is left alone: the host
// redoes the chunk from the radix sort's output, which this kernel only reads.
__global__ __launch_bounds__(256) void k_group_order(const uint64_t *__restrict__ keys, uint64_t n, uint32_t shift, uint64_t rel_mask,
                                                     uint64_t *__restrict__ out, uint32_t *__restrict__ flags, uint32_t *__restrict__ big) {
  constexpr uint32_t HALO = GROUP_CAP + 1, TILE = 1 + 256 + HALO;
  __shared__ uint64_t sk[TILE];
  const uint64_t base = (uint64_t)blockIdx.x * 256;
  for (uint32_t x = threadIdx.x; x < TILE; x += 256) {
    const int64_t g = (int64_t)base - 1 + x;
    sk[x] = (g >= 0 && (uint64_t)g < n) ? keys[g] : 0;
  }
  __syncthreads();
  const uint64_t i = base + threadIdx.x;
  const uint32_t me = 1 + threadIdx.x;
  const uint64_t hi = sk[me] >> shift;
  const bool head = i < n && (i == 0 || (sk[me - 1] >> shift) != hi);   // the first key of its group
  __syncthreads();                                                     // every lane has looked before any group is reordered
  if (!head) return;
  const uint32_t last = (uint32_t)min((uint64_t)TILE, n - base + 1);  // sk[last - 1] = the last real key in reach
  uint32_t up = me + 1;
  while (up < last && up - me <= GROUP_CAP && (sk[up] >> shift) == hi) up++;
  const uint32_t g = up - me;
  if (g > GROUP_CAP) {
    atomicOr(big, 1u);
    return;
  }
  const uint64_t lowmask = (1ull << shift) - 1ull;
  for (uint32_t a = me + 1; a < up; a++) {                           // insertion sort by the low bits (ties: any order, the keys are equal)
    const uint64_t v = sk[a];
    uint32_t b = a;
    while (b > me && (sk[b - 1] & lowmask) > (v & lowmask)) {
      sk[b] = sk[b - 1];
      b--;
    }
    sk[b] = v;
  }
  int64_t kept = 0;
  for (uint32_t a = me; a < up; a++) {
    const uint64_t v = sk[a];
    const int64_t rel = (int64_t)((v >> 1) & rel_mask);
    const bool keep = a == me || rel - kept >= 3;
    if (keep) kept = rel;
    out[i + (a - me)] = v;
    flags[i + (a - me)] = keep ? 1u : 0u;
  }
}

__global__ __launch_bounds__(256) void k_dedupe_compact(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ flags,
                                                        const uint32_t *__restrict__ pos, uint64_t n, OverlapKeyLayout lay,
                                                        uint32_t read_id_base, kslam_overlap *__restrict__ out) {
  __shared__ uint4 s_rec[4][192];
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool keep = i < n && flags[i];
  const uint64_t kept = __ballot(keep);
  if (!kept) return;   // (wave-uniform)
  kslam_overlap o;
  memset(&o, 0, sizeof o);
  uint32_t at = 0;
  if (keep) {
    const uint64_t k = keys[i];
    o.revcomp = (uint8_t)(k & 1u);
    o.rel = (int32_t)((k >> 1) & ((1ull << lay.bits_rel) - 1)) - (int32_t)lay.rel_bias;
    o.entry = (uint32_t)((k >> (lay.bits_rel + 1)) & ((1ull << lay.bits_entry) - 1));
    o.read = (uint32_t)(k >> (lay.bits_entry + lay.bits_rel + 1)) + read_id_base;
    at = pos[i];
  }
  // the wave's survivors go to consecutive records: one coalesced burst (common.h)
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t base = (uint32_t)__shfl((int)at, (int)__builtin_ctzll(kept), 64);
  wave_store_records_compact(out, base, (uint32_t)__popcll(kept), keep, (uint32_t)__popcll(kept & ((1ull << lane) - 1ull)), o,
                             s_rec[threadIdx.x >> 6]);
}

}  // namespace

void build_bucket_table(const uint64_t *d_keys, uint32_t n, uint32_t bits, uint32_t *d_bucket,
                        hipStream_t s) {
  const uint32_t nb = 1u << bits;
  if (n == 0) {
    hipLaunchKernelGGL(k_bucket_empty, dim3((nb + 256) / 256), dim3(256), 0, s, nb, d_bucket);
  } else {
    hipLaunchKernelGGL(k_bucket, dim3((n + 255) / 256), dim3(256), 0, s, d_keys, n, 64 - bits, nb, d_bucket);
  }
  HIPCHK(hipGetLastError());
}

void build_offsets_table(const uint64_t *d_keys, uint32_t n, uint32_t shift, uint32_t nb, uint32_t *d_table, hipStream_t s) {
  if (n == 0) hipLaunchKernelGGL(k_bucket_empty, dim3((nb + 256) / 256), dim3(256), 0, s, nb, d_table);
  else hipLaunchKernelGGL(k_bucket, dim3((n + 255) / 256), dim3(256), 0, s, d_keys, n, shift, nb, d_table);
  HIPCHK(hipGetLastError());
}

void join_fill_single_pass(const uint4 *d_read_recs, uint32_t n_r, GenomeIndexDev g, const uint32_t *d_read_len,
                           uint64_t *d_cursor, uint64_t cap, OverlapKeyLayout lay, uint64_t *d_out, hipStream_t s) {
  HIPCHK(hipMemsetAsync(d_cursor, 0, sizeof(uint64_t), s));
  if (n_r == 0) return;
  unsigned blocks = (n_r + JOIN_TILE - 1) / JOIN_TILE;
  hipLaunchKernelGGL(k_join_fill, dim3(blocks), dim3(JB), 0, s, d_read_recs, n_r, g, d_read_len,
                     reinterpret_cast<unsigned long long *>(d_cursor), cap, lay, d_out);
  HIPCHK(hipGetLastError());
}

void join_fill_merge(const uint4 *d_read_recs, uint32_t n_r, GenomeIndexDev g, uint32_t sorted_top_bits, const uint32_t *d_read_len,
                     uint64_t *d_cursor, uint64_t cap, OverlapKeyLayout lay, uint64_t *d_out, hipStream_t s) {
  HIPCHK(hipMemsetAsync(d_cursor, 0, sizeof(uint64_t), s));
  if (n_r == 0) return;
  const uint32_t gb = std::min(sorted_top_bits, g.bucket_bits);
  if (gb == 0 || gb > 31) throw StatusError{KSLAM_ERR_STATE, "the merge join needs read records ordered by 1..31 top key bits"};
  unsigned blocks = (n_r + JOIN_TILE - 1) / JOIN_TILE;
  hipLaunchKernelGGL(k_join_merge, dim3(blocks), dim3(JB), 0, s, d_read_recs, n_r, g, gb, d_read_len,
                     reinterpret_cast<unsigned long long *>(d_cursor), cap, lay, d_out);
  HIPCHK(hipGetLastError());
}

void group_order(const uint64_t *d_keys, uint64_t n, OverlapKeyLayout lay, uint64_t *d_out, uint32_t *d_flags, uint32_t *d_big, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_group_order, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_keys, n, lay.bits_rel + 1, (1ull << lay.bits_rel) - 1, d_out,
                     d_flags, d_big);
  HIPCHK(hipGetLastError());
}

void dedupe_flags(const uint64_t *d_keys, uint64_t n, OverlapKeyLayout lay, uint32_t *d_flags, hipStream_t s) {
  if (n == 0) return;
  unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_dedupe_flags, dim3(blocks), dim3(256), 0, s, d_keys, n, lay.bits_rel + 1,
                     (1ull << lay.bits_rel) - 1, d_flags);
  HIPCHK(hipGetLastError());
}

void dedupe_compact(const uint64_t *d_keys, const uint32_t *d_flags, const uint32_t *d_pos, uint64_t n,
                    OverlapKeyLayout lay, uint32_t read_id_base, kslam_overlap *d_out, hipStream_t s) {
  if (n == 0) return;
  unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_dedupe_compact, dim3(blocks), dim3(256), 0, s, d_keys, d_flags, d_pos, n, lay,
                     read_id_base, d_out);
  HIPCHK(hipGetLastError());
}

}  // namespace kslam
