// selfcheck.cpp -- does THIS build's std::sort permute like csrc/gnu_sort.h says it does?
//
// The reference ranks and screens a read pair's alignment pairs with std::sort on partial keys
// (src/PairedOverlap.h:369, 403, 527; src/SAM.h:448): equal keys are the rule, std::sort is unstable, and
// its permutation decides which alignments survive.  The host tail (tail.cpp) calls the std::sort of the
// libstdc++ it is compiled with; the device stages (pairs.hip) run csrc/gnu_sort.h / wave_gnu_sort.h, a
// restatement of GCC's introsort (threshold 16, median of three, depth limit 2 log2 n, heap-sort fallback).
// The two only agree as long as the toolchain's <algorithm> is that algorithm, so the library checks it
// itself: kslam_check_std_sort() compares them element for element on tie-heavy, ordered, organ-pipe,
// all-equal and median-of-three-killer arrays.  __graft_entry__.build() runs it and refuses a library that
// fails; kslam_version() reports the libstdc++ it was verified against; kslam_pair_screen / kslam_set_pairing
// refuse the device stages when it fails (kslam_api.hip), so that a mismatch can never pass silently.
// This file is compiled with the same host compiler and flags as tail.cpp.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kslam.h"
#include "../csrc/gnu_sort.h"

namespace {

struct El { uint32_t key, id; };   // partial key: the id shows which of two equal elements came first

uint64_t splitmix(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <class Less> bool same(const std::vector<El> &v, Less less) {
  std::vector<El> a = v, b = v;
  std::sort(a.begin(), a.end(), less);
  kslam_gnu::sort(b.data(), b.data() + b.size(), less);
  for (size_t i = 0; i < a.size(); i++)
    if (a[i].key != b[i].key || a[i].id != b[i].id) return false;
  // and the shortcut that finds only the element std::sort leaves in front (host/taxonomy.cpp)
  if (!v.empty()) {
    std::vector<El> c = v;
    const El *f = kslam_gnu::front_after_sort(c.data(), c.data() + c.size(), less);
    if (f->key != a[0].key || f->id != a[0].id) return false;
  }
  return true;
}

// Musser's median-of-three killer: drives an introsort into its heap-sort fallback
std::vector<El> killer(size_t n) {
  std::vector<El> v(n);
  const size_t k = n / 2;
  for (size_t i = 1; i <= k; i++) {
    if (i % 2 == 1) { v[i - 1].key = (uint32_t)i; v[i].key = (uint32_t)(k + i); }
    v[k + i - 1].key = (uint32_t)(2 * i);
  }
  for (size_t i = 0; i < n; i++) v[i].id = (uint32_t)i;
  return v;
}

struct Verdict { bool ok = false; uint64_t arrays = 0; };

Verdict run_check() {
  Verdict r;
  uint64_t seed = 12345;
  auto asc = [](const El &a, const El &b) { return a.key < b.key; };
  auto desc = [](const El &a, const El &b) { return a.key > b.key; };
  for (int round = 0; round < 3000; round++) {
    const size_t n = round < 1200 ? splitmix(seed) % 40 : (round < 2800 ? splitmix(seed) % 300 : splitmix(seed) % 5000);
    const uint32_t distinct = 1 + (uint32_t)(splitmix(seed) % (round % 3 == 0 ? 3 : (round % 3 == 1 ? 20 : 100000)));
    std::vector<El> v(n);
    for (size_t i = 0; i < n; i++) v[i] = El{(uint32_t)(splitmix(seed) % distinct), (uint32_t)i};
    const int shape = (int)(splitmix(seed) % 6);
    if (shape == 1) std::stable_sort(v.begin(), v.end(), asc);
    if (shape == 2) std::stable_sort(v.begin(), v.end(), desc);
    if (shape == 3 && n > 2) { std::stable_sort(v.begin(), v.end(), asc); std::reverse(v.begin() + n / 2, v.end()); }
    if (!same(v, asc) || !same(v, desc)) return r;
    r.arrays += 2;
  }
  for (size_t n : {17u, 33u, 64u, 100u, 257u, 1000u, 4096u, 20001u}) {
    if (!same(killer(n), asc)) return r;
    std::vector<El> all(n);
    for (size_t i = 0; i < n; i++) all[i] = El{7, (uint32_t)i};
    if (!same(all, asc) || !same(all, desc)) return r;
    r.arrays += 3;
  }
  r.ok = true;
  return r;
}

const Verdict &verdict() {
  static Verdict v;
  static std::once_flag once;
  std::call_once(once, [] { v = run_check(); });
  return v;
}

}  // namespace

extern "C" {

int kslam_check_std_sort(uint64_t *n_arrays) {
  const Verdict &v = verdict();
  if (n_arrays) *n_arrays = v.arrays;
  return v.ok ? 1 : 0;
}

const char *kslam_version(void) {
  static std::string s;
  static std::once_flag once;
  std::call_once(once, [] {
    const Verdict &v = verdict();
    char buf[512];
#if defined(__GLIBCXX__)
    const long glibcxx = __GLIBCXX__;
#else
    const long glibcxx = 0;
#endif
#if defined(_GLIBCXX_RELEASE)
    const int rel = _GLIBCXX_RELEASE;
#else
    const int rel = 0;
#endif
    snprintf(buf, sizeof buf,
             "kslam-mi355x abi %u; gfx950; host compiler %s; libstdc++ release %d (__GLIBCXX__ %ld); "
             "std::sort permutation == csrc/gnu_sort.h: %s (%llu arrays compared when the library was loaded)",
             (unsigned)KSLAM_ABI_VERSION, __VERSION__, rel, glibcxx, v.ok ? "verified" : "MISMATCH",
             (unsigned long long)v.arrays);
    s = buf;
  });
  return s.c_str();
}

}  // extern "C"
