// ref_fastq_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin C entry point around the REAL reference FASTQ reader: it includes the reference's
// src/FASTQsequence.h (and through it src/sequenceTools.h) where they lie under /root/reference
// -- nothing is copied -- and is compiled by oracle/Makefile into oracle/_ref/libfastq_ref.so.
// Reads the file the way the low-memory loop does (src/SLAM.h:193-207): one std::ifstream,
// getSequencesFromFASTQFile called with per_call reads at a time until it returns nothing.
#include <inttypes.h>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "FASTQsequence.h"

extern "C" {

// calls[] receives the number of reads each call returned (up to max_calls entries)
int ref_fastq_read(const char *path, uint32_t per_call, uint64_t *n_out, char **bases, uint64_t **bases_off,
                   char **qual, uint64_t **qual_off, char **ids, uint64_t **ids_off, uint64_t *calls,
                   uint64_t max_calls, uint64_t *n_calls) {
  std::ifstream in(path, std::ios::binary);
  if (!in.good()) return 1;
  std::vector<SLAM::FASTQSequence> all;
  *n_calls = 0;
  for (;;) {
    std::vector<SLAM::FASTQSequence> reads;
    SLAM::getSequencesFromFASTQFile(in, reads, per_call);
    if (reads.empty()) break;
    if (*n_calls < max_calls) calls[(*n_calls)++] = reads.size();
    all.insert(all.end(), reads.begin(), reads.end());
  }
  *n_out = all.size();
  std::string cb, cq, ci;
  *bases_off = (uint64_t *)malloc(8 * (all.size() + 1));
  *qual_off = (uint64_t *)malloc(8 * (all.size() + 1));
  *ids_off = (uint64_t *)malloc(8 * (all.size() + 1));
  for (size_t i = 0; i < all.size(); i++) {
    (*bases_off)[i] = cb.size();
    (*qual_off)[i] = cq.size();
    (*ids_off)[i] = ci.size();
    cb += all[i].bases;
    cq += all[i].quality;
    ci += all[i].sequenceIdentifier;
  }
  (*bases_off)[all.size()] = cb.size();
  (*qual_off)[all.size()] = cq.size();
  (*ids_off)[all.size()] = ci.size();
  auto dup = [](const std::string &x) {
    char *p = (char *)malloc(x.size() + 1);
    memcpy(p, x.data(), x.size());
    p[x.size()] = 0;
    return p;
  };
  *bases = dup(cb);
  *qual = dup(cq);
  *ids = dup(ci);
  return 0;
}

void ref_fastq_free(void *p) { free(p); }
}
