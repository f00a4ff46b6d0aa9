// abi_bench.cpp -- times the HOST-POINTER entry point kslam_align_batch (what the reference-side
// binding calls): read concatenation + H2D + hot path + D2H, vs the device-resident hot path alone.
// Build + run on the GPU box:
//   g++ -O2 -std=c++11 tools/abi_bench.cpp -o /tmp/abi_bench -Lk-slam_amd -lkslam_hip -Wl,-rpath,$PWD/k-slam_amd && /tmp/abi_bench
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../include/kslam.h"

static uint64_t rng_state = 88172645463325252ull;
static inline uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }

int main(int argc, char** argv) {
  const size_t n_pairs = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000;
  const size_t n_genomes = 50, glen = 4000000, L = 150;
  std::vector<std::string> genomes(n_genomes);
  for (auto& g : genomes) { g.resize(glen); for (auto& c : g) c = "ACGT"[rnd() & 3]; }
  std::vector<std::string> reads(2 * n_pairs);
  const char* comp = "TGCA";  // complement of ACGT by index
  for (size_t p = 0; p < n_pairs; p++) {
    const std::string& g = genomes[rnd() % n_genomes];
    size_t s = rnd() % (glen - 400), f = 300 + rnd() % 100;
    reads[p] = g.substr(s, L);
    std::string r2(L, 'A');
    for (size_t k = 0; k < L; k++) { char c = g[s + f - 1 - k]; r2[k] = comp[c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3]; }
    reads[n_pairs + p] = r2;
    for (int m = 0; m < 2; m++) { reads[p][rnd() % L] = "ACGT"[rnd() & 3]; reads[n_pairs + p][rnd() % L] = "ACGT"[rnd() & 3]; }
  }
  kslam_params prm{2, 3, 5, 2, 0, 1, 0, 0};
  kslam_ctx* ctx = nullptr;
  if (kslam_create(&prm, &ctx) != KSLAM_OK) { std::printf("create: %s\n", ctx ? kslam_last_error(ctx) : "?"); return 2; }
  std::vector<const char*> gp(n_genomes); std::vector<uint64_t> gl(n_genomes);
  for (size_t i = 0; i < n_genomes; i++) { gp[i] = genomes[i].data(); gl[i] = genomes[i].size(); }
  auto t0 = std::chrono::steady_clock::now();
  if (kslam_set_index(ctx, n_genomes, gp.data(), gl.data()) != KSLAM_OK) { std::printf("index: %s\n", kslam_last_error(ctx)); return 2; }
  auto t1 = std::chrono::steady_clock::now();
  std::printf("set_index (200 Mb, host pointers): %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
  std::vector<const char*> rp(reads.size()); std::vector<uint32_t> rl(reads.size());
  for (size_t i = 0; i < reads.size(); i++) { rp[i] = reads[i].data(); rl[i] = (uint32_t)reads[i].size(); }
  for (int it = 0; it < 4; it++) {
    kslam_overlap* ov = nullptr; uint32_t* pool = nullptr; uint64_t n = 0, nc = 0;
    auto a = std::chrono::steady_clock::now();
    if (kslam_align_batch(ctx, reads.size(), rp.data(), rl.data(), &ov, &n, &pool, &nc) != KSLAM_OK) { std::printf("align: %s\n", kslam_last_error(ctx)); return 2; }
    auto b = std::chrono::steady_clock::now();
    kslam_timings tm; kslam_get_timings(ctx, &tm);
    std::printf("kslam_align_batch %zu reads: wall %.1f ms (device hot path %.1f ms), %llu overlaps, %llu cigar ops -> %.2f M reads/s incl. host<->device\n",
                reads.size(), std::chrono::duration<double, std::milli>(b - a).count(), tm.ms_total, (unsigned long long)n,
                (unsigned long long)nc, reads.size() / std::chrono::duration<double>(b - a).count() / 1e6);
    kslam_free_batch(ctx, ov, pool);
  }
  kslam_destroy(ctx);
  return 0;
}
