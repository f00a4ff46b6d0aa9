// radix_sort.hip -- LSD radix sort of k-mer records / packed overlap keys.
//
// Replaces __gnu_parallel::sort in sortKMers (reference src/KMer.h:388-398,
// key = kMerInt asc then ID_isFromGB_RC desc) and in findOverlaps_parallel
// (reference src/Overlap.h:289, key = read, entry, relativePosition).
//
// MI355X design (HBM-bound: every pass moves each record once in, once out):
//   * per pass three streaming steps, no inter-workgroup waiting anywhere:
//       k_tile_hist  reads the tile (16 B/lane coalesced) and writes its 256 digit counts,
//       k_chunk_scan / k_top_scan turn the [tile][digit] counts into global bin offsets,
//       k_scatter    re-reads the tile, ranks records per wavefront with ballot match masks
//                    (stable, wave64), reorders the tile through a 64 KB LDS stage and writes
//                    bin-contiguous runs (avg 256 B per bin per tile);
//     a single-pass "onesweep" with decoupled look-back was measured first: on this chip the
//     look-back words sit behind the fabric and the scatter workgroups spent ~45 % of their life
//     waiting on them (2.2 ms vs 1.2 ms per pass); paying one extra streaming read is cheaper;
//   * algorithmic traffic of the scatter kernel = 32 B per record per launch.
#include "common.h"

namespace kslam {

namespace {

constexpr int RS_BLOCK = 512;
constexpr int RS_WAVES = RS_BLOCK / 64;
constexpr int RS_ITEMS = SORT_TILE / RS_BLOCK;  // 8
constexpr int MAX_PASSES = 12;

struct PassList {
  SortPass p[MAX_PASSES];
  int n;
};

template <int RW> struct RecT;
template <> struct RecT<4> { using type = uint4; };
template <> struct RecT<2> { using type = uint2; };

__device__ inline uint32_t rec_word(const uint4 &r, uint32_t w) {
  return w == 0 ? r.x : (w == 1 ? r.y : (w == 2 ? r.z : r.w));
}
__device__ inline uint32_t rec_word(const uint2 &r, uint32_t w) { return w == 0 ? r.x : r.y; }

template <typename T> __device__ inline uint32_t digit_of(const T &r, const SortPass &p) {
  return sort_pass_digit(rec_word(r, p.word), p);
}
// a two-word record is also a 64-bit key: word 2 = "the digit at bit `shift` of the whole key", whatever words it straddles
__device__ inline uint32_t digit_of(const uint2 &r, const SortPass &p) {
  if (p.word == 2u) return (uint32_t)(((((uint64_t)r.y) << 32) | r.x) >> p.shift) & 0xFFu;
  return sort_pass_digit(rec_word(r, p.word), p);
}

// ---- per-tile digit histogram of one pass ----------------------------------------------------
// One workgroup per 4096-record tile: 16 B/lane coalesced loads, LDS atomic histogram, 1 KiB out.
template <int RW>
__device__ __forceinline__ void tile_hist_body(const typename RecT<RW>::type *__restrict__ in, uint32_t n,
                                               SortPass pass, uint32_t *__restrict__ tile_hist) {
  __shared__ uint32_t h[256];
  const uint32_t tid = threadIdx.x;
  if (tid < 256) h[tid] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * SORT_TILE;
#pragma unroll
  for (int it = 0; it < RS_ITEMS; it++) {
    const uint32_t i = base + it * RS_BLOCK + tid;
    if (i < n) atomicAdd(&h[digit_of(in[i], pass)], 1u);
  }
  __syncthreads();
  if (tid < 256) tile_hist[(uint64_t)blockIdx.x * 256 + tid] = h[tid];
}

// The same histogram from the pass's DIGIT BYTES: the scatter of pass p also stores every record's digit of pass
// p + 1 at the record's destination (one byte next to the 8 / 16 the record takes), so that pass p + 1's
// histogram reads 1 byte per record instead of the whole record -- the histogram read had been a third of a
// pass's traffic.  One workgroup of 256 threads per tile, 16 digits per thread.
__device__ __forceinline__ void tile_hist_bytes_body(const uint8_t *__restrict__ dig, uint32_t n,
                                                      uint32_t *__restrict__ tile_hist) {
  __shared__ uint32_t h[256];
  const uint32_t tid = threadIdx.x;
  h[tid] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * SORT_TILE + tid * 16;
  if (base + 16 <= n) {
    const uint4 v = *reinterpret_cast<const uint4 *>(dig + base);   // tiles start at multiples of 4096: aligned
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int b = 0; b < 4; b++) atomicAdd(&h[(w[k] >> (8 * b)) & 0xFFu], 1u);
  } else {
    for (uint32_t i = base; i < n && i < base + 16; i++) atomicAdd(&h[dig[i]], 1u);
  }
  __syncthreads();
  tile_hist[(uint64_t)blockIdx.x * 256 + tid] = h[tid];
}
__global__ __launch_bounds__(256) void k_tile_hist_bytes(const uint8_t *__restrict__ dig, uint32_t n,
                                                         uint32_t *__restrict__ tile_hist) {
  tile_hist_bytes_body(dig, n, tile_hist);
}
// The one-time sorts' byte histograms (312 M digit bytes per pass for the 5 Gb database), two kernels the host picks from per
// pass.  Both: HT = 4 tiles per workgroup, their 16-byte loads all in flight before the first LDS operation, a histogram per tile.
constexpr uint32_t HT = 4;
// (1) digits of the K-MER bytes are spread over the 256 values: an LDS atomic per digit (0.10 ms per pass).
__global__ __launch_bounds__(256) void k_tile_hist_bytes_setup(const uint8_t *__restrict__ dig, uint32_t n, uint32_t n_tiles,
                                                               uint32_t *__restrict__ tile_hist) {
  __shared__ uint32_t h[HT][256];
  const uint32_t tid = threadIdx.x;
#pragma unroll
  for (uint32_t t = 0; t < HT; t++) h[t][tid] = 0;
  __syncthreads();
  const uint32_t tile0 = blockIdx.x * HT;
  uint4 v[HT];
#pragma unroll
  for (uint32_t t = 0; t < HT; t++) {
    const uint32_t base = (tile0 + t) * SORT_TILE + tid * 16;
    v[t] = (tile0 + t < n_tiles && base + 16 <= n) ? *reinterpret_cast<const uint4 *>(dig + base) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (uint32_t t = 0; t < HT; t++) {
    const uint32_t base = (tile0 + t) * SORT_TILE + tid * 16;
    if (tile0 + t >= n_tiles) break;
    if (base + 16 <= n) {
      const uint32_t w[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int b = 0; b < 4; b++) atomicAdd(&h[t][(w[k] >> (8 * b)) & 0xFFu], 1u);
    } else {
      for (uint32_t i = base; i < n && i < base + 16; i++) atomicAdd(&h[t][dig[i]], 1u);
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t t = 0; t < HT; t++)
    if (tile0 + t < n_tiles) tile_hist[(uint64_t)(tile0 + t) * 256 + tid] = h[t][tid];
}
// (2) digits of the META word take a handful of values -- the low id byte is ONE value over a whole tile of records in
// extraction order, the {revComp, high id bits} digit two to ten -- and 64 lanes adding to one or two LDS addresses serialise
// (1.09 ms per such pass with kernel (1); 0.8 with 64 private copies of the histogram, whose zeroing and summing cost more
// than the atomics they spared).  Here a thread first counts its OWN 16 bytes per distinct value (byte-parallel compare
// against the first byte not yet counted: one or two rounds), the wave then adds up the counts of the lanes that hold
// the same value (ballot + butterfly), and ONE lane per value and wave touches the histogram.  Only right for such digits:
// 16 distinct bytes in a thread would be 16 rounds.
__global__ __launch_bounds__(256) void k_tile_hist_bytes_skew_setup(const uint8_t *__restrict__ dig, uint32_t n, uint32_t n_tiles,
                                                                    uint32_t *__restrict__ tile_hist) {
  __shared__ uint32_t h[HT][256];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
#pragma unroll
  for (uint32_t t = 0; t < HT; t++) h[t][tid] = 0;
  __syncthreads();
  const uint32_t tile0 = blockIdx.x * HT;
  uint4 v[HT];
#pragma unroll
  for (uint32_t t = 0; t < HT; t++) {
    const uint32_t base = (tile0 + t) * SORT_TILE + tid * 16;
    v[t] = (tile0 + t < n_tiles && base + 16 <= n) ? *reinterpret_cast<const uint4 *>(dig + base) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (uint32_t t = 0; t < HT; t++) {
    const uint32_t base = (tile0 + t) * SORT_TILE + tid * 16;
    if (tile0 + t >= n_tiles) break;
    const bool whole = base + 16 <= n;
    if (!whole)    // (the list's last, partial 16 bytes: one thread of the whole launch)
      for (uint32_t i = base; i < n && i < base + 16; i++) atomicAdd(&h[t][dig[i]], 1u);
    const uint32_t w[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
    uint32_t done[4] = {0, 0, 0, 0};   // 0x80 in every byte already counted
    bool more = whole;
    while (__any(more)) {
      uint32_t val = 0, cnt = 0;
      if (more) {
        // the first byte not yet counted
        const uint32_t r0 = ~done[0] & 0x80808080u, r1 = ~done[1] & 0x80808080u, r2 = ~done[2] & 0x80808080u, r3 = ~done[3] & 0x80808080u;
        const uint32_t k = r0 ? 0u : (r1 ? 1u : (r2 ? 2u : 3u));
        const uint32_t rk = r0 ? r0 : (r1 ? r1 : (r2 ? r2 : r3)), wk = k == 0 ? w[0] : (k == 1 ? w[1] : (k == 2 ? w[2] : w[3]));
        val = (wk >> (((uint32_t)__builtin_ctz(rk)) & 24u)) & 0xFFu;
        const uint32_t pat = val * 0x01010101u;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t x = w[q] ^ pat;
          uint32_t eq = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
          eq = ~(eq | x | 0x7F7F7F7Fu);                 // 0x80 exactly in the bytes of w[q] that equal val
          cnt += (uint32_t)__builtin_popcount(eq);
          done[q] |= eq;
        }
        more = (done[0] & done[1] & done[2] & done[3]) != 0x80808080u;
      }
      // the wave's lanes that hold the same value add up; one lane per value touches LDS
      uint64_t todo = __ballot(cnt != 0);
      while (todo) {
        const int leader = (int)__builtin_ctzll(todo);
        const uint32_t vl = (uint32_t)__shfl((int)val, leader, 64);
        const bool mine = cnt != 0 && val == vl;
        const uint64_t m = __ballot(mine);
        uint32_t part = mine ? cnt : 0u;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += (uint32_t)__shfl_xor((int)part, d, 64);
        if ((int)lane == leader) atomicAdd(&h[t][vl], part);
        todo &= ~m;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t t = 0; t < HT; t++)
    if (tile0 + t < n_tiles) tile_hist[(uint64_t)(tile0 + t) * 256 + tid] = h[t][tid];
}

// ---- scan of the tile histograms (per digit, over tiles) --------------------------------------
constexpr int CHUNK_TILES = 64;
// level 1: one workgroup per chunk of 64 tiles, thread = digit: in-chunk exclusive prefix in place
__global__ __launch_bounds__(256) void k_chunk_scan(uint32_t *__restrict__ tile_hist, uint32_t n_tiles,
                                                    uint32_t *__restrict__ chunk_tot) {
  const uint32_t d = threadIdx.x;
  const uint32_t t0 = blockIdx.x * CHUNK_TILES, t1 = min(t0 + CHUNK_TILES, n_tiles);
  uint32_t acc = 0;
  // 32 loads in flight per thread: a load-add-store loop over the 64 tiles was a chain of 64 round trips (23 us per
  // launch, twelve launches per batch: a fifth of the sort phase's time went into this small kernel)
  constexpr uint32_t U = 32;
  for (uint32_t t = t0; t < t1; t += U) {
    uint32_t v[U];
#pragma unroll
    for (uint32_t k = 0; k < U; k++) v[k] = t + k < t1 ? tile_hist[(uint64_t)(t + k) * 256 + d] : 0u;
#pragma unroll
    for (uint32_t k = 0; k < U; k++)
      if (t + k < t1) {
        tile_hist[(uint64_t)(t + k) * 256 + d] = acc;
        acc += v[k];
      }
  }
  chunk_tot[(uint64_t)blockIdx.x * 256 + d] = acc;
}
// level 2: one workgroup per digit: exclusive prefix of that digit's chunk totals (column scan)
__global__ __launch_bounds__(256) void k_col_scan(uint32_t *__restrict__ chunk_tot, uint32_t n_chunks,
                                                  uint32_t *__restrict__ digit_tot) {
  __shared__ uint32_t ws[4];
  const uint32_t d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t per = (n_chunks + 255) / 256;
  const uint32_t c0 = tid * per, c1 = min(c0 + per, n_chunks);
  uint32_t sum = 0;
  for (uint32_t c = c0; c < c1; c++) sum += chunk_tot[(uint64_t)c * 256 + d];
  uint32_t inc = sum;
#pragma unroll
  for (int dd = 1; dd < 64; dd <<= 1) {
    uint32_t t = __shfl_up(inc, dd, 64);
    if (lane >= (uint32_t)dd) inc += t;
  }
  if (lane == 63) ws[w] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (uint32_t i = 0; i < 4; i++) { if (i < w) base += ws[i]; tot += ws[i]; }
  uint32_t acc = base + inc - sum;
  for (uint32_t c = c0; c < c1; c++) {
    const uint32_t v = chunk_tot[(uint64_t)c * 256 + d];
    chunk_tot[(uint64_t)c * 256 + d] = acc;
    acc += v;
  }
  if (tid == 0) digit_tot[d] = tot;
}
// level 3: exclusive scan of the 256 digit totals -> global bin bases
__global__ __launch_bounds__(256) void k_bin_scan(uint32_t *__restrict__ digit_tot) {
  __shared__ uint32_t ws[4];
  const uint32_t d = threadIdx.x, lane = d & 63, w = d >> 6;
  const uint32_t v = digit_tot[d];
  uint32_t inc = v;
#pragma unroll
  for (int dd = 1; dd < 64; dd <<= 1) {
    uint32_t t = __shfl_up(inc, dd, 64);
    if (lane >= (uint32_t)dd) inc += t;
  }
  if (lane == 63) ws[w] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t i = 0; i < w; i++) base += ws[i];
  digit_tot[d] = base + inc - v;
}

template <int RW>
__global__ __launch_bounds__(RS_BLOCK) void k_tile_hist(const typename RecT<RW>::type *__restrict__ in, uint32_t n,
                                                        SortPass pass, uint32_t *__restrict__ tile_hist) {
  tile_hist_body<RW>(in, n, pass, tile_hist);
}
// The same kernel under another name for the one-time sorts (index build: 312 M records, 12 passes),
// so that a profile's per-kernel averages of k_tile_hist / k_scatter are those of the per-batch sort.
template <int RW>
__global__ __launch_bounds__(RS_BLOCK) void k_tile_hist_setup(const typename RecT<RW>::type *__restrict__ in, uint32_t n,
                                                              SortPass pass, uint32_t *__restrict__ tile_hist) {
  tile_hist_body<RW>(in, n, pass, tile_hist);
}

// ---- scatter of one LSD pass --------------------------------------------------------------------
// Workgroup = tile.  Stable per-wave ranking with ballot match masks, per-wave LDS digit counters,
// tile-local bin starts, then the tile is reordered through a 64 KB LDS stage so that every digit's
// records leave as one contiguous run at  chunk_base[chunk][d] + tile_prefix[tile][d].
template <int RW>
__device__ __forceinline__ void scatter_body(const typename RecT<RW>::type *__restrict__ in,
                                             typename RecT<RW>::type *__restrict__ out, uint32_t n,
                                             const uint32_t *__restrict__ tile_prefix,
                                             const uint32_t *__restrict__ chunk_base,
                                             const uint32_t *__restrict__ bin_base, SortPass pass,
                                             uint8_t *__restrict__ next_digits, SortPass next_pass) {
  using T = typename RecT<RW>::type;
  __shared__ T stage[SORT_TILE];
  __shared__ uint32_t wave_hist[RS_WAVES][256];
  __shared__ uint32_t tile_off[256];
  __shared__ uint32_t glob_delta[256];
  __shared__ uint32_t wsum[4];

  const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < RS_WAVES * 256; i += RS_BLOCK) (&wave_hist[0][0])[i] = 0;
  __syncthreads();
  // XCD-aware tile mapping: workgroups b, b+8, b+16, ... share an XCD (and its L2), so give each
  // XCD a CONTIGUOUS range of tiles: the runs it appends to a bin are then adjacent in memory and
  // merge in that L2 before they are written back, instead of leaving as isolated 256-byte pieces
  const uint32_t nb = gridDim.x, xq = nb >> 3, xr = nb & 7u, xcd = blockIdx.x & 7u;
  const uint32_t tile = xcd * xq + min(xcd, xr) + (blockIdx.x >> 3);
  const uint32_t tile_base = tile * SORT_TILE;
  const uint32_t count = min((uint32_t)SORT_TILE, n - tile_base);

  // coalesced loads: within a wave, item `it` of lane l is record w*512 + it*64 + l
  T item[RS_ITEMS];
  uint32_t rank[RS_ITEMS];
  const uint32_t wbase = w * (64 * RS_ITEMS);
#pragma unroll
  for (int it = 0; it < RS_ITEMS; it++) {
    uint32_t loc = wbase + it * 64 + lane;
    if (loc < count) item[it] = in[tile_base + loc];
  }
  uint32_t gbase = 0;
  if (tid < 256)
    gbase = bin_base[tid] + chunk_base[(uint64_t)(tile / CHUNK_TILES) * 256 + tid] + tile_prefix[(uint64_t)tile * 256 + tid];
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int it = 0; it < RS_ITEMS; it++) {
    const bool valid = (wbase + it * 64 + lane) < count;
    const uint32_t d = valid ? digit_of(item[it], pass) : 0u;
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (d >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const uint32_t pre = wave_hist[w][d];
    const uint32_t rnk = __popcll(m & lt_mask);
    __builtin_amdgcn_wave_barrier();
    if (valid && rnk == 0) wave_hist[w][d] = pre + (uint32_t)__popcll(m);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    rank[it] = pre + rnk;
  }
  __syncthreads();
  // per digit: exclusive prefix over waves, tile total; exclusive scan over digits -> tile-local bin starts
  uint32_t total = 0;
  if (tid < 256) {
#pragma unroll
    for (int i = 0; i < RS_WAVES; i++) {
      uint32_t t = wave_hist[i][tid];
      wave_hist[i][tid] = total;
      total += t;
    }
  }
  uint32_t inc = total;
  if (tid < 256) {
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      uint32_t t = __shfl_up(inc, dd, 64);
      if (lane >= (uint32_t)dd) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
  }
  __syncthreads();
  if (tid < 256) {
    uint32_t base = 0;
    for (uint32_t i = 0; i < w; i++) base += wsum[i];
    const uint32_t toff = base + inc - total;
    tile_off[tid] = toff;
    glob_delta[tid] = gbase - toff;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < RS_ITEMS; it++) {
    if ((wbase + it * 64 + lane) < count) {
      const uint32_t d = digit_of(item[it], pass);
      stage[tile_off[d] + wave_hist[w][d] + rank[it]] = item[it];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const uint32_t p = j * RS_BLOCK + tid;
    if (p < count) {
      const T r = stage[p];
      const uint32_t dest = glob_delta[digit_of(r, pass)] + p;
      out[dest] = r;
      if (next_digits) next_digits[dest] = (uint8_t)digit_of(r, next_pass);   // the next pass's histogram reads these
    }
  }
}

template <int RW>
__global__ __launch_bounds__(RS_BLOCK) void k_scatter(const typename RecT<RW>::type *__restrict__ in,
                                                      typename RecT<RW>::type *__restrict__ out, uint32_t n,
                                                      const uint32_t *__restrict__ tile_prefix,
                                                      const uint32_t *__restrict__ chunk_base,
                                                      const uint32_t *__restrict__ bin_base, SortPass pass,
                                                      uint8_t *__restrict__ next_digits, SortPass next_pass) {
  scatter_body<RW>(in, out, n, tile_prefix, chunk_base, bin_base, pass, next_digits, next_pass);
}
template <int RW>
__global__ __launch_bounds__(RS_BLOCK) void k_scatter_setup(const typename RecT<RW>::type *__restrict__ in,
                                                            typename RecT<RW>::type *__restrict__ out, uint32_t n,
                                                            const uint32_t *__restrict__ tile_prefix,
                                                            const uint32_t *__restrict__ chunk_base,
                                                            const uint32_t *__restrict__ bin_base, SortPass pass,
                                                            uint8_t *__restrict__ next_digits, SortPass next_pass) {
  scatter_body<RW>(in, out, n, tile_prefix, chunk_base, bin_base, pass, next_digits, next_pass);
}

template <int RW, bool SETUP>
void sort_impl(void *a, void *b, uint32_t n, const PassList &pl, SortWorkspace &ws, hipStream_t s,
               hipEvent_t ev0, hipEvent_t ev1, uint32_t *n_launches, void **result) {
  using T = typename RecT<RW>::type;
  const uint32_t tiles = (n + SORT_TILE - 1) / SORT_TILE;
  const uint32_t chunks = (tiles + CHUNK_TILES - 1) / CHUNK_TILES;
  uint32_t *tile_hist = ws.status.as<uint32_t>();
  uint32_t *chunk_tot = ws.hist.as<uint32_t>();
  uint32_t *digit_tot = ws.tickets.as<uint32_t>();
  T *src = (T *)a, *dst = (T *)b;
  // digit bytes of the NEXT pass, written by each scatter next to the records (ws.digits: n bytes): a pass then reads
  // 1 + 16 bytes per record and writes 16 + 1 instead of reading 16 + 16 (the one-time index sort too, since round 5:
  // its 312 M records x 11 passes are the largest sort of the path)
  uint8_t *digits = (pl.n > 1 && ws.use_digit_bytes) ? ws.digits.as<uint8_t>() : nullptr;
  if (ev0) HIPCHK(hipEventRecord(ev0, s));
  for (int p = 0; p < pl.n; p++) {
    if (SETUP && digits && (p > 0 || ws.first_digits_ready)) {
      // a digit of few values (a pass over the meta word of k-mer records): the conflict-free shape
      if (RW == 4 && pl.p[p].word == 2 && ws.meta_digits_in_runs) hipLaunchKernelGGL(k_tile_hist_bytes_skew_setup, dim3((tiles + HT - 1) / HT), dim3(256), 0, s, (const uint8_t *)digits, n, tiles, tile_hist);
      else hipLaunchKernelGGL(k_tile_hist_bytes_setup, dim3((tiles + HT - 1) / HT), dim3(256), 0, s, (const uint8_t *)digits, n, tiles, tile_hist);
    }
    else if (SETUP) hipLaunchKernelGGL(k_tile_hist_setup<RW>, dim3(tiles), dim3(RS_BLOCK), 0, s, (const T *)src, n, pl.p[p], tile_hist);
    else if (digits && (p > 0 || ws.first_digits_ready)) hipLaunchKernelGGL(k_tile_hist_bytes, dim3(tiles), dim3(256), 0, s, (const uint8_t *)digits, n, tile_hist);
    else hipLaunchKernelGGL(k_tile_hist<RW>, dim3(tiles), dim3(RS_BLOCK), 0, s, (const T *)src, n, pl.p[p], tile_hist);
    hipLaunchKernelGGL(k_chunk_scan, dim3(chunks), dim3(256), 0, s, tile_hist, tiles, chunk_tot);
    hipLaunchKernelGGL(k_col_scan, dim3(256), dim3(256), 0, s, chunk_tot, chunks, digit_tot);
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(256), 0, s, digit_tot);
    if (ws.ev_sc0) HIPCHK(hipEventRecord(ws.ev_sc0[p], s));
    uint8_t *nd = (digits && p + 1 < pl.n) ? digits : nullptr;   // (pass p's histogram has read the array by now)
    const SortPass np = pl.p[p + 1 < pl.n ? p + 1 : p];
    if (SETUP) hipLaunchKernelGGL(k_scatter_setup<RW>, dim3(tiles), dim3(RS_BLOCK), 0, s, (const T *)src, dst, n, tile_hist,
                                  chunk_tot, digit_tot, pl.p[p], nd, np);
    else hipLaunchKernelGGL(k_scatter<RW>, dim3(tiles), dim3(RS_BLOCK), 0, s, (const T *)src, dst, n, tile_hist,
                            chunk_tot, digit_tot, pl.p[p], nd, np);
    if (ws.ev_sc0) HIPCHK(hipEventRecord(ws.ev_sc1[p], s));
    T *t = src; src = dst; dst = t;
    if (n_launches) (*n_launches)++;
  }
  if (ev1) HIPCHK(hipEventRecord(ev1, s));
  HIPCHK(hipGetLastError());
  *result = src;
}

}  // namespace

void *radix_sort(void *a, void *b, uint64_t n, int rec_words, const SortPass *passes, int n_passes,
                 SortWorkspace &ws, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, uint32_t *n_launches, bool setup) {
  if (n_passes > MAX_PASSES) throw StatusError{KSLAM_ERR_ARG, "too many radix passes"};
  if (n >= (1ull << 32)) throw StatusError{KSLAM_ERR_UNSUPPORTED, "radix sort of >= 2^32 records"};
  if (n == 0 || n_passes == 0) {
    if (ev0) HIPCHK(hipEventRecord(ev0, s));
    if (ev1) HIPCHK(hipEventRecord(ev1, s));
    return a;
  }
  PassList pl;
  pl.n = n_passes;
  for (int i = 0; i < n_passes; i++) pl.p[i] = passes[i];
  const uint64_t tiles = (n + SORT_TILE - 1) / SORT_TILE;
  const uint64_t chunks = (tiles + CHUNK_TILES - 1) / CHUNK_TILES;
  ws.status.ensure(tiles * 256 * sizeof(uint32_t));   // per-tile digit histograms / prefixes
  ws.hist.ensure(chunks * 256 * sizeof(uint32_t));    // per-chunk totals / bases
  ws.tickets.ensure(256 * sizeof(uint32_t));           // per-digit totals -> bin bases
  if (n_passes > 1 && ws.use_digit_bytes) ws.digits.ensure(n + 64);   // next-pass digit of every record
  void *res = nullptr;
  if (rec_words == 4 && setup) sort_impl<4, true>(a, b, (uint32_t)n, pl, ws, s, ev0, ev1, n_launches, &res);
  else if (rec_words == 4) sort_impl<4, false>(a, b, (uint32_t)n, pl, ws, s, ev0, ev1, n_launches, &res);
  else if (rec_words == 2 && setup) sort_impl<2, true>(a, b, (uint32_t)n, pl, ws, s, ev0, ev1, n_launches, &res);
  else if (rec_words == 2) sort_impl<2, false>(a, b, (uint32_t)n, pl, ws, s, ev0, ev1, n_launches, &res);
  else throw StatusError{KSLAM_ERR_ARG, "unsupported record width"};
  return res;
}

}  // namespace kslam
