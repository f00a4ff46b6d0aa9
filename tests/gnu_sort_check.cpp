// Compares k-slam_amd/csrc/gnu_sort.h with the real std::sort, element for element (keys are partial:
// the payload shows which of two equal elements came first).  Build: g++ -O2 -std=c++17 gnu_sort_check.cpp
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <string>
#include <vector>
#include "../k-slam_amd/csrc/gnu_sort.h"

struct El { uint32_t key, id; };
static uint64_t g_cmp = 0;

template <class Less> bool same(std::vector<El> v, Less less) {
  std::vector<El> a = v, b = v;
  std::sort(a.begin(), a.end(), less);
  kslam_gnu::sort(b.data(), b.data() + b.size(), less);
  for (size_t i = 0; i < a.size(); i++)
    if (a[i].key != b[i].key || a[i].id != b[i].id) return false;
  return true;
}

// a sequence on which median-of-three quicksort degrades (Musser): drives std::sort into its heap-sort fallback
static std::vector<El> killer(size_t n) {
  std::vector<El> v(n);
  size_t k = n / 2;
  for (size_t i = 1; i <= k; i++) {
    if (i % 2 == 1) { v[i - 1].key = (uint32_t)i; v[i].key = (uint32_t)(k + i); }
    v[k + i - 1].key = (uint32_t)(2 * i);
  }
  for (size_t i = 0; i < n; i++) v[i].id = (uint32_t)i;
  return v;
}

// `gnu_sort_check perm IN OUT`: IN = u64 n_seg, u64 seg_off[n_seg + 1], i32 keys[]; OUT = u32 perm[]: what the
// real std::sort does to each segment of {key, id} elements compared by key (tests/test_gpu_tail.py compares
// the device's wave sort with it)
static int perm_mode(const char *in, const char *out) {
  FILE *f = std::fopen(in, "rb");
  if (!f) return 2;
  uint64_t n_seg = 0;
  if (std::fread(&n_seg, 8, 1, f) != 1) return 2;
  std::vector<uint64_t> off(n_seg + 1);
  if (std::fread(off.data(), 8, n_seg + 1, f) != n_seg + 1) return 2;
  std::vector<int32_t> keys(off[n_seg]);
  if (!keys.empty() && std::fread(keys.data(), 4, keys.size(), f) != keys.size()) return 2;
  std::fclose(f);
  struct SEl { int32_t key; uint32_t id; };
  std::vector<uint32_t> perm(keys.size());
  for (uint64_t s = 0; s < n_seg; s++) {
    std::vector<SEl> v(off[s + 1] - off[s]);
    for (size_t i = 0; i < v.size(); i++) v[i] = SEl{keys[off[s] + i], (uint32_t)i};
    std::sort(v.begin(), v.end(), [](const SEl &a, const SEl &b) { return a.key < b.key; });
    for (size_t i = 0; i < v.size(); i++) perm[off[s] + i] = v[i].id;
  }
  f = std::fopen(out, "wb");
  if (!f) return 2;
  std::fwrite(perm.data(), 4, perm.size(), f);
  std::fclose(f);
  return 0;
}

int main(int argc, char **argv) {
  if (argc == 4 && std::string(argv[1]) == "perm") return perm_mode(argv[2], argv[3]);
  std::mt19937_64 rng(12345);
  auto asc = [](const El &a, const El &b) { return a.key < b.key; };
  auto desc = [](const El &a, const El &b) { return a.key > b.key; };
  uint64_t cases = 0;
  for (int round = 0; round < 60000; round++) {
    size_t n = round < 20000 ? rng() % 40 : (round < 50000 ? rng() % 300 : rng() % 5000);
    uint32_t distinct = 1 + (uint32_t)(rng() % (round % 3 == 0 ? 3 : (round % 3 == 1 ? 20 : 100000)));
    std::vector<El> v(n);
    for (size_t i = 0; i < n; i++) v[i] = El{(uint32_t)(rng() % distinct), (uint32_t)i};
    int shape = (int)(rng() % 6);
    if (shape == 1) std::sort(v.begin(), v.end(), asc);
    if (shape == 2) std::sort(v.begin(), v.end(), desc);
    if (shape == 3 && n > 2) { std::sort(v.begin(), v.end(), asc); std::reverse(v.begin() + n / 2, v.end()); }
    if (!same(v, asc) || !same(v, desc)) { std::printf("MISMATCH round %d n %zu\n", round, n); return 1; }
    cases += 2;
  }
  for (size_t n : {17u, 33u, 64u, 100u, 257u, 1000u, 4096u, 20001u, 100000u}) {
    if (!same(killer(n), asc)) { std::printf("MISMATCH killer n %zu\n", n); return 1; }
    std::vector<El> all(n);
    for (size_t i = 0; i < n; i++) all[i] = El{7, (uint32_t)i};
    if (!same(all, asc) || !same(all, desc)) { std::printf("MISMATCH all-equal n %zu\n", n); return 1; }
    cases += 3;
  }
  // does the killer really reach the heap-sort fallback?  count comparisons: far above n log n if it degraded
  {
    std::vector<El> v = killer(100000);
    uint64_t c = 0;
    std::sort(v.begin(), v.end(), [&](const El &a, const El &b) { c++; return a.key < b.key; });
    std::printf("killer(100000): %llu comparisons by std::sort\n", (unsigned long long)c);
  }
  std::printf("GNU_SORT_OK %llu cases\n", (unsigned long long)cases);
  return 0;
}
