// scan.hip -- device-wide exclusive scan (plumbing for the join / dedupe /
// cigar-pool layout).  Three launches: tile sums, scan of the tile sums by one
// workgroup, tile-local scan + tile prefix.  Everything is HBM-streaming:
// 4 B read + 4/8 B written per element, plus a second 4 B read.
#include "common.h"

namespace kslam {

namespace {
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ inline uint64_t wave_incl_scan_u64(uint64_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// block-wide exclusive scan of one u64 per thread; returns exclusive prefix,
// total in *total (all threads)
__device__ inline uint64_t block_excl_scan(uint64_t v, uint64_t *total, uint64_t *sm /*[5]*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint64_t inc = wave_incl_scan_u64(v);
  if (lane == 63) sm[w] = inc;
  __syncthreads();
  uint64_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_BLOCK / 64; i++) {
    uint64_t x = sm[i];
    if (i < w) base += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_tile_sums(const uint32_t *in, uint64_t n,
                                                          uint64_t *tile_sum) {
  __shared__ uint64_t sm[8];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    uint64_t idx = base + (uint64_t)i * SCAN_BLOCK + threadIdx.x;
    if (idx < n) acc += in[idx];
  }
  uint64_t tot;
  block_excl_scan(acc, &tot, sm);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_tile_sums(uint64_t *tile_sum, uint64_t n_tiles,
                                                               uint64_t *total_out) {
  __shared__ uint64_t sm[8];
  uint64_t carry = 0;
  for (uint64_t base = 0; base < n_tiles; base += SCAN_BLOCK) {
    uint64_t idx = base + threadIdx.x;
    uint64_t v = idx < n_tiles ? tile_sum[idx] : 0;
    uint64_t tot;
    uint64_t ex = block_excl_scan(v, &tot, sm);
    if (idx < n_tiles) tile_sum[idx] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

// Tile-local scan + tile prefix.  A thread takes FOUR consecutive elements per round (one 16-byte load; a wave's loads and
// stores are then contiguous) over four rounds of 1 024 elements, instead of sixteen consecutive elements of its own
// (64-byte stride between lanes: every line fetched in four pieces, 1.45 TB/s); the block scan runs once per round with
// the carry of the rounds before it.
template <typename OutT>
__global__ __launch_bounds__(SCAN_BLOCK) void k_tile_scan(const uint32_t *in, OutT *out, uint64_t n,
                                                          const uint64_t *tile_prefix) {
  __shared__ uint64_t sm[8];
  constexpr int ROUNDS = SCAN_ITEMS / 4;
  const uint64_t tile = (uint64_t)blockIdx.x * SCAN_TILE;
  uint32_t v[ROUNDS][4];
  const bool in16 = (reinterpret_cast<uintptr_t>(in) & 15u) == 0, out16 = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const uint64_t idx = tile + (uint64_t)r * (SCAN_BLOCK * 4) + (uint64_t)threadIdx.x * 4;
    if (idx + 4 <= n && in16) {
      const uint4 q = *reinterpret_cast<const uint4 *>(in + idx);   // (tile and thread offsets are multiples of 4 elements)
      v[r][0] = q.x; v[r][1] = q.y; v[r][2] = q.z; v[r][3] = q.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) v[r][j] = idx + j < n ? in[idx + j] : 0;
    }
  }
  uint64_t carry = tile_prefix[blockIdx.x];
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const uint64_t mine = (uint64_t)v[r][0] + v[r][1] + v[r][2] + v[r][3];
    uint64_t tot;
    uint64_t ex = block_excl_scan(mine, &tot, sm) + carry;
    carry += tot;
    const uint64_t idx = tile + (uint64_t)r * (SCAN_BLOCK * 4) + (uint64_t)threadIdx.x * 4;
    OutT o[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      o[j] = (OutT)ex;
      ex += v[r][j];
    }
    if (idx + 4 <= n && out16) {
      if (sizeof(OutT) == 4) {
        *reinterpret_cast<uint4 *>(out + idx) = make_uint4((uint32_t)o[0], (uint32_t)o[1], (uint32_t)o[2], (uint32_t)o[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) out[idx + j] = o[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (idx + j < n) out[idx + j] = o[j];
    }
  }
}

template <typename OutT>
void scan_impl(const uint32_t *d_in, OutT *d_out, uint64_t n, uint64_t *d_total, void *d_tmp,
               hipStream_t s) {
  uint64_t *tile_sum = reinterpret_cast<uint64_t *>(d_tmp);
  if (n == 0) {
    if (d_total) HIPCHK(hipMemsetAsync(d_total, 0, sizeof(uint64_t), s));
    return;
  }
  uint64_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(k_tile_sums, dim3((unsigned)n_tiles), dim3(SCAN_BLOCK), 0, s, d_in, n, tile_sum);
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(SCAN_BLOCK), 0, s, tile_sum, n_tiles, d_total);
  hipLaunchKernelGGL(k_tile_scan<OutT>, dim3((unsigned)n_tiles), dim3(SCAN_BLOCK), 0, s, d_in, d_out, n,
                     tile_sum);
  HIPCHK(hipGetLastError());
}
}  // namespace

size_t scan_tmp_bytes(uint64_t n) { return ((n + SCAN_TILE - 1) / SCAN_TILE + 1) * sizeof(uint64_t); }

void exclusive_scan_u32(const uint32_t *d_in, uint32_t *d_out, uint64_t n, uint64_t *d_total,
                        void *d_tmp, hipStream_t s) {
  scan_impl<uint32_t>(d_in, d_out, n, d_total, d_tmp, s);
}
void exclusive_scan_u32_to_u64(const uint32_t *d_in, uint64_t *d_out, uint64_t n, uint64_t *d_total,
                               void *d_tmp, hipStream_t s) {
  scan_impl<uint64_t>(d_in, d_out, n, d_total, d_tmp, s);
}

}  // namespace kslam
