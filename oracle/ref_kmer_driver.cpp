// ref_kmer_driver.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref).
//
// Thin extern "C" driver around the REAL reference k-mer code, compiled from
// the reference header where it lies (-I/root/reference/src, src/KMer.h and
// its only project include src/sequenceTools.h).  No reference source is
// copied and no stand-in header is supplied: KMer.h needs nothing the image
// lacks.  The standard headers below are pre-included because KMer.h relies
// on transitive includes (omp_get_max_threads, std::array, numeric_limits).
//
// Used to pin oracle/kslam_oracle.c's orc_extract_kmers / orc_sort_kmers
// (and through them the HIP extraction + radix sort) to the reference's
// getKMers_parallel (src/KMer.h:190-241) and sortKMers (src/KMer.h:388-398).
#include <omp.h>
#include <array>
#include <limits>
#include <vector>
#include <string>
#include <cstdint>
#include <cstring>
#include <unistd.h>
#include "KMer.h"

namespace {
struct Seq {
  std::string bases;  // the only member getKMers_parallel touches
};
typedef SLAM::KMerAndData<uint64_t, 32> Rec;
static_assert(sizeof(Rec) == 16, "reference record is 16 bytes");
}

extern "C" {

// returns the number of records written (<= cap), or the needed count if > cap
uint64_t ref_extract_kmers(uint64_t n, const char *const *seqs,
                           const uint64_t *lens, int is_gb, unsigned gap,
                           void *out, uint64_t cap) {
  std::vector<Seq> entries(n);
  for (uint64_t i = 0; i < n; i++) entries[i].bases.assign(seqs[i], lens[i]);
  std::vector<Rec> kmers;
  SLAM::getKMers_parallel(entries, kmers, is_gb != 0, gap);
  if (kmers.size() <= cap && !kmers.empty())
    std::memcpy(out, kmers.data(), kmers.size() * sizeof(Rec));
  return kmers.size();
}

// sortKMers() logs through a function-static Log that opens ./log.txt, so run
// it from `workdir` (a scratch directory).
int ref_sort_kmers(void *recs, uint64_t n, const char *workdir) {
  char cwd[4096];
  if (!getcwd(cwd, sizeof cwd)) return -1;
  if (workdir && chdir(workdir) != 0) return -2;
  std::vector<Rec> v(n);
  if (n) std::memcpy(v.data(), recs, n * sizeof(Rec));
  SLAM::sortKMers(v);
  if (n) std::memcpy(recs, v.data(), n * sizeof(Rec));
  if (workdir && chdir(cwd) != 0) return -3;
  return 0;
}

// addBaseToKMers<uint32_t,3> on a short string: pins the 2-bit code
// (src/KMer.h:27 "TAG" example)
void ref_kmer3(const char *s, uint32_t *fwd, uint32_t *rc) {
  uint32_t k = 0, r = 0;
  for (const char *p = s; *p; ++p) SLAM::addBaseToKMers<uint32_t, 3>(*p, k, r);
  *fwd = k;
  *rc = r;
}
}
