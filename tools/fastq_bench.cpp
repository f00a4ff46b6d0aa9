// fastq_bench.cpp -- host-only timing of the FASTQ ingest (include/kslam_fastq.h).
//   g++ -O3 -std=c++17 -pthread tools/fastq_bench.cpp k-slam_amd/host/fastq.cpp -o /tmp/fastq_bench
//   /tmp/fastq_bench [n_pairs] [threads] [iters]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../include/kslam_fastq.h"

extern "C" const char *kslam_tail_last_error(void) { return "(see status)"; }  // lives in tail.cpp in the library

int main(int argc, char **argv) {
  const uint64_t n_pairs = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000;
  const int threads = argc > 2 ? atoi(argv[2]) : 0, iters = argc > 3 ? atoi(argv[3]) : 5;
  std::string r1, r2;
  const std::string seq(150, 'A'), qual(150, 'I');
  for (uint64_t i = 0; i < n_pairs; i++) {
    std::string h = "@SRR000001." + std::to_string(i) + " 071112_SLXA-EAS1_s_7:5:1:817:345 length=150";
    r1 += h + "/1\n" + seq + "\n+\n" + qual + "\n";
    r2 += h + "/2\n" + seq + "\n+\n" + qual + "\n";
  }
  printf("%llu pairs, 2 x %.1f MB of FASTQ text\n", (unsigned long long)n_pairs, r1.size() / 1e6);
  for (int it = 0; it < iters; it++) {
    kslam_reads_columns cols;
    uint64_t u1, u2;
    auto t0 = std::chrono::steady_clock::now();
    kslam_status rc = kslam_fastq_parse_pair(r1.data(), r1.size(), r2.data(), r2.size(), 0, 1, threads, &cols, &u1, &u2);
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc) { printf("error %d\n", rc); return 1; }
    printf("iter %d: %.1f ms  %.2f GB/s of text  %.1f M reads/s  (%llu reads)\n", it, ms,
           (r1.size() + r2.size()) / ms / 1e6, cols.n_reads / ms / 1e3, (unsigned long long)cols.n_reads);
    kslam_reads_free(&cols);
  }
  return 0;
}
