// std_sort_front.cpp -- test helper: which record does THE REAL std::sort leave in front?
// Reads n little-endian uint32 keys from the file named on the command line, sorts {key, index} records by key alone with
// libstdc++'s std::sort (the permutation of equal keys is the algorithm's), prints the index of the record at position 0.
// tests/test_taxonomy.py uses it for the one place where the reference's XML report depends on that (combineTaxonomies,
// src/MetagenomicResults.h:149-177, sorted by one thread); the product finds the same record without sorting
// (kslam_gnu::front_after_sort, k-slam_amd/csrc/gnu_sort.h).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
int main(int argc, char **argv) {
  if (argc < 2) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 3;
  struct Rec { uint32_t key, idx; };
  std::vector<Rec> v;
  uint32_t k;
  while (fread(&k, 4, 1, f) == 1) v.push_back(Rec{k, (uint32_t)v.size()});
  fclose(f);
  std::sort(v.begin(), v.end(), [](const Rec &a, const Rec &b) { return a.key < b.key; });
  printf("%u\n", v.empty() ? 0u : v[0].idx);
  return 0;
}
