// merge.hip -- per-shard results of a read-sharded batch -> one batch-global result, on the device.
//
// The hot path shards by read pair: shard s aligns pairs [lo_s, hi_s) of the
// batch against its own replica of the index, with local read ids in the reference's block layout
// ([R1 of its pairs | R2 of its pairs], src/FASTQsequence.h:111-123).  After the one gather of the
// path the collecting device holds the shards' records back to back; this file turns them into what
// ONE context would have returned for the whole batch, byte for byte:
//   * read ids re-based to the batch (R1 of pair p is p, R2 is n_pairs + p);
//   * rows in the reference's order (read, entry, rel) (src/Overlap.h:87-98): every shard's rows are
//     already sorted by local read id, i.e. its R1 rows come first, so the merged list is
//     [R1 rows of shard 0 | R1 rows of shard 1 | ... | R2 rows of shard 0 | ...];
//   * the CIGAR pool re-laid in row order (an exclusive scan of cigar_len), which is the layout the
//     single-context path produces (cigar.hip k_finalize).
// All of it is streaming: every row is read once and written once, every CIGAR op likewise.
//
// The second half of the file is the cheaper protocol the library's own multi-GPU paths use: after a
// count exchange every shard knows where its R1 rows, R2 rows and CIGAR words go in the batch-global
// arrays, re-bases its own records (k_export_rows, on its own GPU, all shards in parallel) and the
// transfers land in their final place -- the collecting device runs no kernel at all.
#include <cstddef>

#include "common.h"

namespace kslam {

namespace {

__global__ void k_merge_plan(const kslam_overlap *__restrict__ rows, MergeShard *__restrict__ sh, uint32_t n_shards) {
  // one thread per shard: first row whose local read id is in the R2 block
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n_shards) {
    const kslam_overlap *r = rows + sh[s].row_base;
    uint64_t lo = 0, hi = sh[s].n_rows;
    const uint32_t n_loc = (uint32_t)(sh[s].pair_hi - sh[s].pair_lo);
    while (lo < hi) {
      const uint64_t mid = lo + ((hi - lo) >> 1);
      if (r[mid].read < n_loc) lo = mid + 1; else hi = mid;
    }
    sh[s].n_r1 = lo;
  }
  __syncthreads();   // (single block: n_shards <= 256)
  if (s == 0) {
    uint64_t a = 0;
    for (uint32_t k = 0; k < n_shards; k++) { sh[k].out_r1 = a; a += sh[k].n_r1; }
    for (uint32_t k = 0; k < n_shards; k++) { sh[k].out_r2 = a; a += sh[k].n_rows - sh[k].n_r1; }
  }
}

__global__ __launch_bounds__(256) void k_merge_rows(const kslam_overlap *__restrict__ rows, uint64_t n,
                                                    const MergeShard *__restrict__ sh, uint32_t n_shards,
                                                    uint64_t n_pairs, kslam_overlap *__restrict__ out,
                                                    uint32_t *__restrict__ lens) {
  __shared__ MergeShard s_sh[MERGE_MAX_SHARDS];
  for (uint32_t k = threadIdx.x; k < n_shards; k += blockDim.x) s_sh[k] = sh[k];
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s = 0;
  while (s + 1 < n_shards && i >= s_sh[s + 1].row_base) s++;
  const MergeShard m = s_sh[s];
  kslam_overlap o = rows[i];
  const uint64_t k = i - m.row_base;
  const uint32_t n_loc = (uint32_t)(m.pair_hi - m.pair_lo);
  const bool r2 = o.read >= n_loc;
  o.read = r2 ? (uint32_t)(n_pairs + m.pair_lo + (o.read - n_loc)) : (uint32_t)(m.pair_lo + o.read);
  o.cigar_off = o.cigar_len ? m.pool_base + o.cigar_off : 0;   // where the ops sit in the gathered pools
  const uint64_t pos = r2 ? m.out_r2 + (k - m.n_r1) : m.out_r1 + k;
  out[pos] = o;
  lens[pos] = o.cigar_len;
}

__global__ __launch_bounds__(256) void k_merge_cigars(kslam_overlap *__restrict__ out, uint64_t n,
                                                      const uint64_t *__restrict__ new_off,
                                                      const uint32_t *__restrict__ pool_in, uint32_t *__restrict__ pool_out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t len = out[i].cigar_len;
  if (!len) return;
  const uint64_t src = out[i].cigar_off, dst = new_off[i];
  for (uint32_t j = 0; j < len; j++) pool_out[dst + j] = pool_in[src + j];
  out[i].cigar_off = dst;
}

// ---- the sending side: a shard's own results, already in batch terms --------------------------
// Split of the rows into the R1 / R2 blocks (binary search: rows are sorted by local read id) and the
// CIGAR words that belong to the R1 rows: the pool is in row order, so that is the offset of the
// first R2 row that has a CIGAR (rows without one carry offset 0, hence no binary search there).
__global__ void k_shard_split(const kslam_overlap *__restrict__ rows, uint64_t n, uint32_t n_local_pairs,
                              uint64_t *__restrict__ out /*[0] rows of R1, [1] = ~0 (no R2 row with a CIGAR seen yet)*/) {
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if (rows[mid].read < n_local_pairs) lo = mid + 1; else hi = mid;
  }
  out[0] = lo;
  out[1] = ~0ull;
}
__global__ __launch_bounds__(256) void k_shard_first_cigar(const kslam_overlap *__restrict__ rows, uint64_t n,
                                                           uint64_t *__restrict__ out) {
  // cigar_off of the first R2 row that has a CIGAR (offsets grow with the row number).  ONE workgroup walks
  // forward from the split, 256 rows a step; nearly always the first step finds it.  (The first version put a
  // thread on every R2 row and took the minimum with one atomic per wave: 640 k same-address atomics for a
  // 10 M-pair batch, 7.3 ms -- the atomics, not the reads.)
  __shared__ unsigned long long s_min[4];
  for (uint64_t base = out[0]; base < n; base += 256) {
    const uint64_t i = base + threadIdx.x;
    unsigned long long v = ~0ull;
    if (i < n && rows[i].cigar_len != 0) v = rows[i].cigar_off;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const unsigned long long o = __shfl_down(v, d, 64);
      v = o < v ? o : v;
    }
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long m = s_min[0];
    for (int k = 1; k < 4; k++) m = s_min[k] < m ? s_min[k] : m;
    __syncthreads();
    if (m != ~0ull) {
      if (threadIdx.x == 0) out[1] = m;
      return;
    }
  }
}

// A record is three 16-byte pieces: {read, entry, rel, strand / score}, the four spans, {cigar_len, pad, cigar_off}.
// One thread per PIECE: loads and stores of a wave are contiguous kilobytes (a thread per 48-byte record made
// every load instruction a stride-48 gather: 0.83 TB/s; now the copy runs at the streaming rate).
__global__ __launch_bounds__(256) void k_export_rows(const uint4 *__restrict__ rows, uint64_t n, uint64_t n_r1,
                                                     uint32_t n_local_pairs, uint64_t pair_lo, uint64_t n_pairs_total,
                                                     uint64_t n_cigar_r1, uint64_t pool_base_r1, uint64_t pool_base_r2,
                                                     uint4 *__restrict__ out_r1, uint4 *__restrict__ out_r2) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= 3 * n) return;
  const uint64_t i = g / 3;
  const uint32_t part = (uint32_t)(g - 3 * i);
  uint4 v = rows[g];
  const bool r1 = i < n_r1;
  if (part == 0) {
    v.x = r1 ? (uint32_t)(pair_lo + v.x) : (uint32_t)(n_pairs_total + pair_lo + (v.x - n_local_pairs));
  } else if (part == 2) {
    const uint64_t off = (uint64_t)v.z | ((uint64_t)v.w << 32);
    const uint64_t noff = v.x ? (r1 ? pool_base_r1 + off : pool_base_r2 + (off - n_cigar_r1)) : 0;
    v.z = (uint32_t)noff;
    v.w = (uint32_t)(noff >> 32);
  }
  if (r1) out_r1[g] = v; else out_r2[g - 3 * n_r1] = v;
}

}  // namespace

void shard_counts(const kslam_overlap *d_rows, uint64_t n, uint32_t n_local_pairs, uint64_t n_cigar, uint64_t *d_out2,
                  hipStream_t s) {
  hipLaunchKernelGGL(k_shard_split, dim3(1), dim3(1), 0, s, d_rows, n, n_local_pairs, d_out2);
  if (n && n_cigar) hipLaunchKernelGGL(k_shard_first_cigar, dim3(1), dim3(256), 0, s, d_rows, n, d_out2);
  HIPCHK(hipGetLastError());
}

void export_rows(const kslam_overlap *d_rows, uint64_t n, uint64_t n_r1, uint32_t n_local_pairs, uint64_t pair_lo,
                 uint64_t n_pairs_total, uint64_t n_cigar_r1, uint64_t pool_base_r1, uint64_t pool_base_r2,
                 kslam_overlap *d_out_r1, kslam_overlap *d_out_r2, hipStream_t s) {
  if (n == 0) return;
  static_assert(sizeof(kslam_overlap) == 48 && offsetof(kslam_overlap, cigar_len) == 32 && offsetof(kslam_overlap, cigar_off) == 40,
                "k_export_rows works on the three 16-byte pieces of a record");
  hipLaunchKernelGGL(k_export_rows, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const uint4 *>(d_rows), n,
                     n_r1, n_local_pairs, pair_lo, n_pairs_total, n_cigar_r1, pool_base_r1, pool_base_r2,
                     reinterpret_cast<uint4 *>(d_out_r1), reinterpret_cast<uint4 *>(d_out_r2));
  HIPCHK(hipGetLastError());
}

void merge_shards(const kslam_overlap *d_rows, uint64_t n_rows, const uint32_t *d_pool_in, MergeShard *d_shards,
                  uint32_t n_shards, uint64_t n_pairs, kslam_overlap *d_out, uint32_t *d_pool_out, uint32_t *d_lens,
                  uint64_t *d_new_off, uint64_t *d_total, void *d_scan_tmp, hipStream_t s) {
  if (n_shards == 0 || n_shards > MERGE_MAX_SHARDS) throw StatusError{KSLAM_ERR_ARG, "merge: 1..256 shards"};
  hipLaunchKernelGGL(k_merge_plan, dim3(1), dim3(256), 0, s, d_rows, d_shards, n_shards);
  if (n_rows) {
    const unsigned nb = (unsigned)((n_rows + 255) / 256);
    hipLaunchKernelGGL(k_merge_rows, dim3(nb), dim3(256), 0, s, d_rows, n_rows, d_shards, n_shards, n_pairs, d_out, d_lens);
    exclusive_scan_u32_to_u64(d_lens, d_new_off, n_rows, d_total, d_scan_tmp, s);
    hipLaunchKernelGGL(k_merge_cigars, dim3(nb), dim3(256), 0, s, d_out, n_rows, d_new_off, d_pool_in, d_pool_out);
  }
  HIPCHK(hipGetLastError());
}

}  // namespace kslam
