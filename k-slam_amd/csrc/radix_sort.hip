// radix_sort.hip -- LSD radix sort of k-mer records / packed overlap keys.
//
// Replaces __gnu_parallel::sort in sortKMers (reference src/KMer.h:388-398,
// key = kMerInt asc then ID_isFromGB_RC desc) and in findOverlaps_parallel
// (reference src/Overlap.h:289, key = read, entry, relativePosition).
//
// MI355X design (HBM-bound: every pass moves each record once in, once out):
//   * one up-front histogram kernel reads the keys ONCE and builds the digit
//     histograms of all passes in LDS (8-bit digits, 256 bins per pass);
//   * each pass is a single "onesweep" kernel: a workgroup takes a 4096-record
//     tile (dynamic ticket), loads it with 16-byte-per-lane coalesced loads,
//     ranks records per wavefront with ballot match masks (stable, wave64),
//     turns the per-wave LDS digit histograms into tile offsets, resolves the
//     tile's global bin offsets by decoupled look-back over 8-byte
//     {epoch,flag,count} words (agent-scope relaxed atomics: one self-contained
//     word per digit, so no payload ordering is needed), reorders the tile
//     through LDS and writes bin-contiguous runs (avg 256 B per bin per tile);
//   * algorithmic traffic = (2 * passes + 1) * record bytes per record.
#include "common.h"

namespace kslam {

namespace {

constexpr int RS_BLOCK = 512;
constexpr int RS_WAVES = RS_BLOCK / 64;
constexpr int RS_ITEMS = SORT_TILE / RS_BLOCK;  // 8
constexpr int MAX_PASSES = 12;
constexpr uint32_t FLAG_AGG = 1, FLAG_INCL = 2;
constexpr uint32_t SPIN_LIMIT = 1u << 24;

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
  return ((rec_word(r, p.word) ^ p.invert) >> p.shift) & 0xFFu;
}

// ---- histograms of all passes in one read of the data ---------------------
template <int RW>
__global__ __launch_bounds__(RS_BLOCK) void k_hist(const typename RecT<RW>::type *__restrict__ in, uint32_t n,
                                                   PassList pl, uint32_t *__restrict__ ghist) {
  __shared__ uint32_t h[MAX_PASSES * 256];
  for (int i = threadIdx.x; i < pl.n * 256; i += RS_BLOCK) h[i] = 0;
  __syncthreads();
  const uint32_t stride = gridDim.x * RS_BLOCK;
  for (uint32_t i = blockIdx.x * RS_BLOCK + threadIdx.x; i < n; i += stride) {
    typename RecT<RW>::type r = in[i];
    for (int p = 0; p < pl.n; p++) atomicAdd(&h[p * 256 + digit_of(r, pl.p[p])], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < pl.n * 256; i += RS_BLOCK) {
    uint32_t v = h[i];
    if (v) atomicAdd(&ghist[i], v);
  }
}

// exclusive scan of each pass's 256 bins (one 256-thread block per pass)
__global__ __launch_bounds__(256) void k_hist_scan(uint32_t *ghist) {
  __shared__ uint32_t ws[4];
  uint32_t *h = ghist + blockIdx.x * 256;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t v = h[threadIdx.x], inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) ws[w] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int i = 0; i < w; i++) base += ws[i];
  h[threadIdx.x] = base + inc - v;
}

__device__ inline uint64_t pack_status(uint32_t epoch, uint32_t flag, uint32_t value) {
  return ((uint64_t)epoch << 34) | ((uint64_t)flag << 32) | value;
}

// ---- one LSD pass ---------------------------------------------------------
template <int RW>
__global__ __launch_bounds__(RS_BLOCK) void k_onesweep(const typename RecT<RW>::type *__restrict__ in,
                                                       typename RecT<RW>::type *__restrict__ out, uint32_t n,
                                                       const uint32_t *__restrict__ bin_base,
                                                       uint64_t *status, uint32_t *ticket, uint32_t epoch,
                                                       SortPass pass, uint32_t *errflag) {
  using T = typename RecT<RW>::type;
  __shared__ T stage[SORT_TILE];
  __shared__ uint32_t wave_hist[RS_WAVES][256];
  __shared__ uint32_t tile_off[256];
  __shared__ uint32_t glob_delta[256];
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t s_tile;

  const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) s_tile = atomicAdd(ticket, 1u);
  for (int i = tid; i < RS_WAVES * 256; i += RS_BLOCK) (&wave_hist[0][0])[i] = 0;
  __syncthreads();
  const uint32_t tile = s_tile;
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
  // stable per-wave ranking with ballot match masks
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

  // per digit: exclusive prefix over waves, tile total, publish aggregate
  uint32_t total = 0;
  if (tid < 256) {
#pragma unroll
    for (int i = 0; i < RS_WAVES; i++) {
      uint32_t t = wave_hist[i][tid];
      wave_hist[i][tid] = total;
      total += t;
    }
    __hip_atomic_store(&status[(uint64_t)tile * 256 + tid],
                       pack_status(epoch, tile == 0 ? FLAG_INCL : FLAG_AGG, total), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
  // exclusive scan of the 256 digit totals -> tile-local bin starts
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
    // decoupled look-back for this digit
    uint32_t prev = 0;
    if (tile > 0) {
      uint32_t t = tile - 1;
      uint32_t spins = 0;
      while (true) {
        uint64_t v = __hip_atomic_load(&status[(uint64_t)t * 256 + tid], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(v >> 34) != epoch) {
          if (++spins > SPIN_LIMIT) { atomicExch(errflag, 1u); break; }
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        prev += (uint32_t)v;
        if (((uint32_t)(v >> 32) & 3u) == FLAG_INCL) break;
        t--;
      }
      __hip_atomic_store(&status[(uint64_t)tile * 256 + tid], pack_status(epoch, FLAG_INCL, prev + total),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    glob_delta[tid] = bin_base[tid] + prev - toff;
  }
  __syncthreads();

  // reorder through LDS so that each bin's records are contiguous
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
      out[glob_delta[digit_of(r, pass)] + p] = r;
    }
  }
}

template <int RW>
void sort_impl(void *a, void *b, uint32_t n, const PassList &pl, SortWorkspace &ws, hipStream_t s,
               hipEvent_t ev0, hipEvent_t ev1, uint32_t *n_launches, void **result) {
  using T = typename RecT<RW>::type;
  const uint32_t tiles = (n + SORT_TILE - 1) / SORT_TILE;
  uint32_t *hist = ws.hist.as<uint32_t>();
  uint32_t *tickets = ws.tickets.as<uint32_t>();
  HIPCHK(hipMemsetAsync(hist, 0, (size_t)pl.n * 256 * sizeof(uint32_t), s));
  HIPCHK(hipMemsetAsync(tickets, 0, (MAX_PASSES + 1) * sizeof(uint32_t), s));
  unsigned hblocks = (unsigned)std::min<uint64_t>(2048, ((uint64_t)n + RS_BLOCK - 1) / RS_BLOCK);
  hipLaunchKernelGGL(k_hist<RW>, dim3(hblocks), dim3(RS_BLOCK), 0, s, (const T *)a, n, pl, hist);
  hipLaunchKernelGGL(k_hist_scan, dim3(pl.n), dim3(256), 0, s, hist);
  T *src = (T *)a, *dst = (T *)b;
  if (ev0) HIPCHK(hipEventRecord(ev0, s));
  for (int p = 0; p < pl.n; p++) {
    ws.epoch++;
    if (ws.epoch >= (1u << 30)) ws.epoch = 1;  // wrapped: stale words from 2^30 passes ago cannot survive
    hipLaunchKernelGGL(k_onesweep<RW>, dim3(tiles), dim3(RS_BLOCK), 0, s, (const T *)src, dst, n,
                       hist + p * 256, ws.status.as<uint64_t>(), tickets + p, ws.epoch, pl.p[p],
                       ws.errflag.as<uint32_t>());
    T *t = src; src = dst; dst = t;
    if (n_launches) (*n_launches)++;
  }
  if (ev1) HIPCHK(hipEventRecord(ev1, s));
  HIPCHK(hipGetLastError());
  *result = src;
}

}  // namespace

void *radix_sort(void *a, void *b, uint64_t n, int rec_words, const SortPass *passes, int n_passes,
                 SortWorkspace &ws, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, uint32_t *n_launches) {
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
  ws.hist.ensure((size_t)MAX_PASSES * 256 * sizeof(uint32_t));
  ws.tickets.ensure((MAX_PASSES + 1) * sizeof(uint32_t));
  if (!ws.errflag.p) {
    ws.errflag.ensure(sizeof(uint32_t));
    HIPCHK(hipMemsetAsync(ws.errflag.p, 0, sizeof(uint32_t), s));
  }
  size_t need = tiles * 256 * sizeof(uint64_t);
  if (need > ws.status.cap) {
    ws.status.ensure(need);
    HIPCHK(hipMemsetAsync(ws.status.p, 0, ws.status.cap, s));  // epoch 0 is never issued
  }
  void *res = nullptr;
  if (rec_words == 4) sort_impl<4>(a, b, (uint32_t)n, pl, ws, s, ev0, ev1, n_launches, &res);
  else if (rec_words == 2) sort_impl<2>(a, b, (uint32_t)n, pl, ws, s, ev0, ev1, n_launches, &res);
  else throw StatusError{KSLAM_ERR_ARG, "unsupported record width"};
  return res;
}

}  // namespace kslam
