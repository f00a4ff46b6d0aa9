// gnu_sort.h -- the permutation libstdc++'s std::sort produces, reproduced step by step.
//
// The reference screens and ranks a read pair's alignment pairs with std::sort on PARTIAL keys
// (insert size, combined score: src/PairedOverlap.h:369, 403; src/SAM.h:448).  Equal keys are the rule
// there, std::sort is not stable, and which of two equal elements comes first decides which alignments
// survive the screens and in which order they are written.  The host tail gets the reference's
// permutation by calling the same std::sort on the same element order (an introsort's permutation depends
// only on the outcomes of its comparisons).  Device code has no std::sort, so this header restates the
// algorithm of GCC's bits/stl_algo.h / bits/stl_heap.h (libstdc++ 11; unchanged in this part since 4.x):
// introsort loop with median-of-three to the front and unguarded partition down to 16 elements, heap sort
// when the depth limit 2 floor(log2 n) runs out, one final insertion sort (guarded for the first 16
// elements, unguarded after).  tests/gnu_sort_check.cpp compares it with the real std::sort element for
// element on tie-heavy, adversarial and random inputs; the same header is compiled for the device.
#pragma once
#include <cstddef>

#if defined(__HIPCC__)
#define KSLAM_HD __host__ __device__
#else
#define KSLAM_HD
#endif

namespace kslam_gnu {

template <typename T>
KSLAM_HD inline void swap_(T &a, T &b) {
  T t = a;
  a = b;
  b = t;
}

template <typename T, typename Less>
KSLAM_HD inline void unguarded_linear_insert(T *last, Less less) {
  T val = *last;
  T *next = last;
  --next;
  while (less(val, *next)) {
    *last = *next;
    last = next;
    --next;
  }
  *last = val;
}

template <typename T, typename Less>
KSLAM_HD inline void insertion_sort(T *first, T *last, Less less) {
  if (first == last) return;
  for (T *i = first + 1; i != last; ++i) {
    if (less(*i, *first)) {
      T val = *i;
      for (T *p = i; p != first; --p) *p = *(p - 1);   // move_backward(first, i, i + 1)
      *first = val;
    } else {
      unguarded_linear_insert(i, less);
    }
  }
}

template <typename T, typename Less>
KSLAM_HD inline void final_insertion_sort(T *first, T *last, Less less) {
  if (last - first > 16) {
    insertion_sort(first, first + 16, less);
    for (T *i = first + 16; i != last; ++i) unguarded_linear_insert(i, less);
  } else {
    insertion_sort(first, last, less);
  }
}

// ---- heap sort (the depth-limit fallback: __partial_sort(first, last, last)) --------------------
template <typename T, typename Less>
KSLAM_HD inline void push_heap_(T *first, ptrdiff_t hole, ptrdiff_t top, T value, Less less) {
  ptrdiff_t parent = (hole - 1) / 2;
  while (hole > top && less(first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}

template <typename T, typename Less>
KSLAM_HD inline void adjust_heap(T *first, ptrdiff_t hole, ptrdiff_t len, T value, Less less) {
  const ptrdiff_t top = hole;
  ptrdiff_t child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less(first[child], first[child - 1])) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  push_heap_(first, hole, top, value, less);
}

template <typename T, typename Less>
KSLAM_HD inline void heap_sort(T *first, T *last, Less less) {
  const ptrdiff_t len = last - first;
  if (len >= 2) {   // __make_heap
    ptrdiff_t parent = (len - 2) / 2;
    while (true) {
      T value = first[parent];
      adjust_heap(first, parent, len, value, less);
      if (parent == 0) break;
      parent--;
    }
  }
  // (__heap_select's loop over [middle, last) is empty: middle == last)
  while (last - first > 1) {   // __sort_heap: __pop_heap(first, last - 1, last - 1)
    --last;
    T value = *last;
    *last = *first;
    adjust_heap(first, (ptrdiff_t)0, (ptrdiff_t)(last - first), value, less);
  }
}

template <typename T, typename Less>
KSLAM_HD inline void move_median_to_first(T *result, T *a, T *b, T *c, Less less) {
  if (less(*a, *b)) {
    if (less(*b, *c)) swap_(*result, *b);
    else if (less(*a, *c)) swap_(*result, *c);
    else swap_(*result, *a);
  } else if (less(*a, *c)) {
    swap_(*result, *a);
  } else if (less(*b, *c)) {
    swap_(*result, *c);
  } else {
    swap_(*result, *b);
  }
}

template <typename T, typename Less>
KSLAM_HD inline T *unguarded_partition(T *first, T *last, T *pivot, Less less) {
  while (true) {
    while (less(*first, *pivot)) ++first;
    --last;
    while (less(*pivot, *last)) --last;
    if (!(first < last)) return first;
    swap_(*first, *last);
    ++first;
  }
}

// std::sort(first, last, less)
template <typename T, typename Less>
KSLAM_HD inline void sort(T *first, T *last, Less less) {
  if (first == last) return;
  ptrdiff_t n = last - first;
  int lg = 0;
  for (ptrdiff_t k = n; k > 1; k >>= 1) lg++;   // std::__lg
  // __introsort_loop(first, last, 2 lg): the recursion on the right part becomes a stack; a part is pushed
  // with the depth limit it was called with
  struct Frame { T *first, *last; int depth; };
  Frame stack[64];
  int sp = 0;
  stack[sp++] = Frame{first, last, 2 * lg};
  while (sp) {
    Frame f = stack[--sp];
    T *lo = f.first, *hi = f.last;
    int depth = f.depth;
    while (hi - lo > 16) {
      if (depth == 0) {
        heap_sort(lo, hi, less);
        break;
      }
      --depth;
      T *mid = lo + (hi - lo) / 2;
      move_median_to_first(lo, lo + 1, mid, hi - 1, less);
      T *cut = unguarded_partition(lo + 1, hi, lo, less);
      // the reference recurses into [cut, hi) first and then loops on [lo, cut); the two parts are
      // disjoint, so the order in which they are finished does not change the result
      if (sp < 64) stack[sp++] = Frame{cut, hi, depth};
      hi = cut;
    }
  }
  final_insertion_sort(first, last, less);
}

// WHICH element std::sort(first, last, less) leaves at *first, without sorting the rest: the chain of partitions that
// leads to the leftmost block, and nothing else.  A partition only moves elements inside its range, and the recursion into
// the right part [cut, hi) never touches [lo, cut) -- so following only the left parts reproduces the leftmost block
// exactly (O(n) instead of O(n log n): n + n/2 + ... element visits).  The final insertion sort never lets an element pass
// an equal one and every element right of the leftmost block compares >= everything in it, so the front element is the first
// least element of that block.  Destroys the order of [first, last).  Returns a pointer into the range (first == last: last).
// Used where the reference's output depends on nothing but the front of a std::sort by a partial key
// (combineTaxonomies, src/MetagenomicResults.h:149-177: host/taxonomy.cpp).
template <typename T, typename Less>
KSLAM_HD inline T *front_after_sort(T *first, T *last, Less less) {
  if (first == last) return last;
  ptrdiff_t n = last - first;
  int lg = 0;
  for (ptrdiff_t k = n; k > 1; k >>= 1) lg++;
  int depth = 2 * lg;
  T *lo = first, *hi = last;
  while (hi - lo > 16) {
    if (depth == 0) {
      heap_sort(lo, hi, less);
      break;
    }
    --depth;
    T *mid = lo + (hi - lo) / 2;
    move_median_to_first(lo, lo + 1, mid, hi - 1, less);
    hi = unguarded_partition(lo + 1, hi, lo, less);
  }
  T *best = lo;
  for (T *p = lo + 1; p != hi; ++p)
    if (less(*p, *best)) best = p;
  return best;
}

}  // namespace kslam_gnu
