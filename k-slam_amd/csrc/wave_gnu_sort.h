// wave_gnu_sort.h -- libstdc++'s std::sort permutation (see gnu_sort.h) produced by the 64 lanes of one
// wavefront together, for an array in LDS.
//
// gnu_sort.h runs the algorithm of bits/stl_algo.h on one lane.  That is right for the many short arrays of
// the per-read-pair screens, and slow for the few long ones of pseudo-assembly (one array per database
// entry, thousands of elements, every step a dependent LDS access).  Three observations make the same
// PERMUTATION a data-parallel job:
//
//  1. __unguarded_partition is a Hoare partition: it swaps the k-th element from the left that is not less
//     than the pivot with the k-th element from the right that the pivot is not less than, for k = 1, 2, ...
//     while the two have not met.  The elements it passes over are never looked at again, so the k-th
//     stoppers of both sides are properties of the array before the call: a wave reads a window of up to
//     64 elements at each end, ballots the stoppers, ranks them with popcounts and performs
//     min(#left, #right) swaps at once -- exactly the swaps the sequential loop makes, in a state the
//     sequential loop passes through (windows never overlap, so the pairs have not met).  The last one or
//     zero unexamined elements give the cut by the sequential loop's own exit rule.
//  2. __final_insertion_sort is a stable insertion sort of an array whose introsort leaves (<= 16
//     elements) are already ordered among each other, so it equals a stable insertion sort of every leaf by
//     itself: one lane per leaf, 64 leaves at a time.
//  3. The two parts a partition leaves are disjoint: the order in which they are finished does not matter.
//
// Median-of-three and the heap-sort fallback (depth limit 2 floor(log2 n), practically never reached) stay
// on lane 0.  All lanes of the wave must call wave_sort with the same arguments; the workgroup is ONE wave
// (the barriers below are workgroup barriers).  kslam_debug_wave_sort (include/kslam.h) exposes it to the
// tests, which compare the permutation with the real std::sort's on tie-heavy arrays of every size class.
#pragma once
#include <hip/hip_runtime.h>

#include "gnu_sort.h"

namespace kslam_gnu {

struct WaveSortLds {
  uint32_t pos_l[64], pos_r[64];
  uint32_t leaf[64];          // lo | (len - 1) << 28   (len 2..16, lo < 2^28)
  uint32_t stack_lo[64], stack_hi[64];
  uint8_t stack_depth[64];
};
constexpr uint32_t WAVE_SORT_MAX = 1u << 28;   // elements

// __unguarded_partition(a + first, a + last, pivot) by one wave; returns the cut
template <typename T, typename Less>
__device__ inline uint32_t wave_partition(T *a, uint32_t first, uint32_t last, const T pivot, Less less, WaveSortLds &S) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t below = (1ull << lane) - 1;
  uint32_t f = first, l = last;   // [f, l): not examined yet; everything before f is <= pivot, everything from l on >= pivot
  while (l - f >= 2) {
    const uint32_t r = l - f;
    const uint32_t wl = min(64u, (r + 1) / 2), wr = min(64u, r - wl);
    bool sl = false, sr = false;
    if (lane < wl) sl = !less(a[f + lane], pivot);           // stops the scan from the left
    if (lane < wr) sr = !less(pivot, a[l - 1 - lane]);       // stops the scan from the right (lane k: k-th from the right)
    const uint64_t ml = __ballot(sl), mr = __ballot(sr);
    const uint32_t cl = __popcll(ml), cr = __popcll(mr), m = min(cl, cr);
    if (sl) S.pos_l[__popcll(ml & below)] = f + lane;
    if (sr) S.pos_r[__popcll(mr & below)] = l - 1 - lane;
    __syncthreads();
    if (lane < m) {
      const uint32_t p = S.pos_l[lane], q = S.pos_r[lane];
      const T u = a[p], v = a[q];
      a[p] = v;
      a[q] = u;
    }
    uint32_t nf, nl;
    if (cl > cr) {          // the right window is used up; the left scan waits at its next stopper
      nf = S.pos_l[m];
      nl = l - wr;
    } else if (cl < cr) {
      nf = f + wl;
      nl = (uint32_t)S.pos_r[m] + 1;
    } else {
      nf = f + wl;
      nl = l - wr;
    }
    __syncthreads();
    f = nf;
    l = nl;
  }
  if (l - f == 1) return less(a[f], pivot) ? f + 1 : f;
  return f;
}

template <typename T, typename Less>
__device__ inline void wave_sort_flush_leaves(T *a, uint32_t n_leaves, Less less, WaveSortLds &S) {
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  if (lane < n_leaves) {
    const uint32_t lo = S.leaf[lane] & 0x0FFFFFFFu, len = (S.leaf[lane] >> 28) + 1;
    insertion_sort(a + lo, a + lo + len, less);
  }
  __syncthreads();
}

// std::sort(a, a + n, less) for n < WAVE_SORT_MAX elements in LDS, or in global memory (the array's home
// decides the instructions after inlining; the barriers order one wave's accesses either way)
template <typename T, typename Less>
__device__ inline void wave_sort(T *a, uint32_t n, Less less, WaveSortLds &S) {
  if (n < 2) return;
  const uint32_t lane = threadIdx.x & 63;
  int lg = 0;
  for (uint32_t k = n; k > 1; k >>= 1) lg++;
  uint32_t sp = 0, n_leaves = 0;
  auto add_leaf = [&](uint32_t lo, uint32_t hi) {
    if (hi - lo < 2) return;
    if (lane == 0) S.leaf[n_leaves] = lo | ((hi - lo - 1) << 28);
    if (++n_leaves == 64) {
      wave_sort_flush_leaves(a, 64, less, S);
      n_leaves = 0;
    }
  };
  uint32_t lo = 0, hi = n;
  int depth = 2 * lg;
  while (true) {
    while (hi - lo > 16) {
      if (depth == 0) {
        __syncthreads();
        if (lane == 0) heap_sort(a + lo, a + hi, less);
        __syncthreads();
        lo = hi;   // sorted: nothing left for the insertion sort to do here
        break;
      }
      --depth;
      const uint32_t mid = lo + (hi - lo) / 2;
      __syncthreads();
      if (lane == 0) move_median_to_first(a + lo, a + lo + 1, a + mid, a + hi - 1, less);
      __syncthreads();
      const T pivot = a[lo];
      const uint32_t cut = wave_partition(a, lo + 1, hi, pivot, less, S);
      if (hi - cut > 16) {
        if (lane == 0) {
          S.stack_lo[sp] = cut;
          S.stack_hi[sp] = hi;
          S.stack_depth[sp] = (uint8_t)depth;
        }
        sp++;
      } else {
        add_leaf(cut, hi);
      }
      hi = cut;
    }
    add_leaf(lo, hi);
    if (!sp) break;
    __syncthreads();
    --sp;
    lo = S.stack_lo[sp];
    hi = S.stack_hi[sp];
    depth = S.stack_depth[sp];
  }
  wave_sort_flush_leaves(a, n_leaves, less, S);
}

}  // namespace kslam_gnu
