// ref_slam_driver.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref/libslam_ref.so).
//
// The REAL reference batch loop, metagenomicAnalysis_Low_Mem
// (src/SLAM.h:159-268), compiled from the reference sources where they lie
// and run on real files: FASTQ in; SAM, <out>, <out>_abbreviated,
// <out>_PerRead and log.txt out.  src/SLAM.h is included WHOLE, and through it
// KMer.h, FASTQsequence.h, Overlap.h, SmithWaterman.h, PairedOverlap.h,
// SAM.h, MetagenomicResults.h, TaxonomyDatabase.h ... all unmodified.
// ssw.c and ssw_cpp.cpp are linked (see ref_join_driver.cpp for how the
// Makefile builds them).  Nothing of the reference is copied into the repo.
//
// Build-time accommodations (oracle/Makefile, in a mktemp directory deleted
// after the compile):
//   (1) ssw_cpp_noboost.h  = `sed 7d src/ssw_cpp.h` (drops the unused
//       `#include <boost/optional.hpp>`); included first so the include guard
//       makes every later `#include "ssw_cpp.h"` a no-op.
//   (2) GenbankTools_noboost.h = `sed -n '18,23p;28,30p;32,200p;206,219p'
//       src/GenbankTools.h` + `}` + `#endif`: the guard, the non-Boost
//       includes, classes CDS / Gene / GenbankEntry / GenbankIndex (minus
//       writeIndexToBoostSerial), getGene, GenbankIndex::getKMers.  Keeps
//       the guard GENBANKTOOLS_H_, so SLAM.h's own include is skipped.
//   (3) getIndexFromBoostSerial (src/GenbankTools.h:336-344) is the one
//       function of the path that cannot exist without Boost.Serialization.
//       This file DEFINES it as a test hook that returns the index the test
//       injected through ref_slam_index_* (the same entries the test writes
//       with the product's kslam_db_write for the product run).  So the
//       archive GRAMMAR is not pinned by this library;
//       everything downstream of the loaded GenbankIndex is.
//   src/main.cpp (boost::program_options) is not built: the globals it sets
//   (src/main.cpp:40-97) are set here from ref_slam_params.
//
// Built a second time as oracle/_ref/libslam_gpu_ref.so with -DKSLAM_REF_GPU_OPERATOR: THE LITERAL DROP-IN.  The same
// program -- the reference's own FASTQ reader, batch loop, pairing, screens, pseudo-assembly, SAM writer and reports,
// compiled from where they lie -- with ONE function swapped: alignToDatabase (src/SLAM.h:59-79) is cut out of the
// header by a third line slice (`sed 59,79d src/SLAM.h`: its declaration at :54-56 stays) and defined below exactly as
// INTEGRATION.md tells a maintainer to define it, through k-slam_amd/host/slam_hot_path.hpp over the C ABI of
// libkslam_hip.so.  tests/test_gpu_dropin.py runs both libraries on the same files and compares the four outputs byte
// for byte.  Test infrastructure like everything under oracle/: the product never links or loads it.
#include <omp.h>
#include <array>
#include <limits>
#include <vector>
#include <string>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <algorithm>
#include <numeric>
#include <thread>
#include <mutex>
#include <unordered_map>
#include <fstream>
#include <sstream>
#include <iostream>
#include <climits>
#include <chrono>
#include <unistd.h>
#include "ssw_cpp_noboost.h"
#include "Globals.h"
#include "sequenceTools.h"
#include "KMer.h"
#include "ParallelTools.h"
#include "FASTQsequence.h"
#include "MetagenomicFASTQSequence.h"
#include "TaxonomyDatabase.h"
#include "GenbankTools_noboost.h"
namespace SLAM {
GenbankIndex getIndexFromBoostSerial(const std::string serialFileName);
}
#ifdef KSLAM_REF_GPU_OPERATOR
#include "SLAM_minus_alignToDatabase.h"     // the Makefile's slice: src/SLAM.h without lines 59-79
#include "../k-slam_amd/host/slam_hot_path.hpp"
namespace SLAM {
static kslam_host::HotPath *gpuPath = nullptr;   // one per process, like the reference's globals
// INTEGRATION.md, "The reference-side binding": the replacement a maintainer would write in src/SLAM.h
template <typename FASTQType>
inline std::vector<Overlap> alignToDatabase(const std::vector<FASTQType> &reads, const GenbankIndex &genbankIndex) {
  log("Aligning reads to database using k = " + std::to_string(k));
  if (!gpuPath) {                       // first batch: upload genomes, build the resident k-mer list
    gpuPath = new kslam_host::HotPath(match, misMatch, gapOpen, gapExtend, scoreThreshold, reportCigar);
    gpuPath->setIndex(genbankIndex);    // uses genbankIndex.entries[j].bases
  }
  return gpuPath->alignToDatabase<Overlap>(reads);   // same order, same Alignment ownership
}
static void dropGpuPath() {   // (the test harness changes index and scoring between runs; a k-SLAM process would not)
  delete gpuPath;
  gpuPath = nullptr;
}
}  // namespace SLAM
#else
#include "SLAM.h"
namespace SLAM {
static void dropGpuPath() {}
}
#endif

namespace {
SLAM::GenbankIndex g_index;
}
namespace SLAM {
GenbankIndex getIndexFromBoostSerial(const std::string) { return g_index; }
}

extern "C" {

struct ref_slam_params {  // the option globals of src/main.cpp:40-97
  uint32_t match, mismatch, gap_open, gap_extend;
  uint32_t score_threshold;     // --min-alignment-score
  uint32_t num_sam_alignments;  // --num-alignments
  double score_fraction;        // --score-fraction-threshold
  int32_t pseudo_assembly;      // !--no-pseudo-assembly
  int32_t sam_xa;               // --sam-xa
  int32_t just_align;           // --just-align
  uint32_t num_reads;           // --num-reads
  uint32_t num_reads_at_once;   // --num-reads-at-once
  int32_t threads;              // OMP_NUM_THREADS (0 = leave)
};

void ref_slam_index_reset(void) {
  g_index.entries.clear();
  SLAM::dropGpuPath();
}
// 1 when alignToDatabase is the GPU operator behind the C ABI (libslam_gpu_ref.so), 0 for the reference's own
int ref_slam_operator_is_gpu(void) {
#ifdef KSLAM_REF_GPU_OPERATOR
  return 1;
#else
  return 0;
#endif
}

// appends a GenbankEntry (fields the archive carries, src/GenbankTools.h:155-163)
void ref_slam_index_add_entry(const char *bases, uint64_t len, const char *locus_tag,
                              uint32_t taxonomy_id, uint32_t genbank_id) {
  SLAM::GenbankEntry e;
  e.bases.assign(bases, len);
  e.locusTag = locus_tag;
  e.taxonomyID = taxonomy_id;
  e.genbankID = genbank_id;
  g_index.entries.push_back(e);
}

// appends a Gene to the last entry (fields of src/GenbankTools.h:101-109)
void ref_slam_index_add_gene(const char *gene_name, const char *locus_tag,
                             const char *protein_id, const char *product,
                             const char *reference_sequence, uint32_t gene_id,
                             uint32_t start, uint32_t stop, int32_t complement) {
  SLAM::Gene g(gene_name, locus_tag, protein_id, product, reference_sequence,
               SLAM::CDS(start, stop, complement != 0));
  g.geneID = gene_id;
  g_index.entries.back().genes.push_back(g);
}

// src/main.cpp:24-29,138-151: sets the globals, then metagenomicAnalysis_Low_Mem.
// r2 may be "" (single end); out / sam may be "".  Runs in `workdir` (log.txt
// lands there).  Returns 0, or 1 when the reference threw.
int ref_slam_run(const char *r1, const char *r2, const char *db_dir, const char *out,
                 const char *sam, const char *command_line, const ref_slam_params *p,
                 const char *workdir) {
  char old[4096];
  if (!getcwd(old, sizeof old)) return 2;
  if (workdir && chdir(workdir) != 0) return 3;
  match = p->match;
  misMatch = p->mismatch;
  gapOpen = p->gap_open;
  gapExtend = p->gap_extend;
  scoreThreshold = p->score_threshold;
  numSAMAlignments = p->num_sam_alignments;
  scoreFractionThreshold = p->score_fraction;
  performPseudoAssembly = p->pseudo_assembly != 0;
  SAMXA = p->sam_xa != 0;
  justAlign = p->just_align != 0;
  reportCigar = false;  // set by the analysis when a SAM file is asked for (src/SLAM.h:169)
  pairedData = true;
  commandLine = command_line;
  if (p->threads > 0) omp_set_num_threads(p->threads);
  int rc = 0;
  SLAM::dropGpuPath();   // the scoring globals may have changed since the last run
  try {
    SLAM::metagenomicAnalysis_Low_Mem(r1, r2, db_dir, out, sam, p->num_reads_at_once,
                                      p->num_reads);
  } catch (const std::exception &e) {
    std::cerr << "reference threw: " << e.what() << std::endl;
    rc = 1;
  }
  SLAM::dropGpuPath();
  if (workdir && chdir(old) != 0) abort();
  return rc;
}

// THE OPERATOR ALONE: the reference's own `alignToDatabase` (src/SLAM.h:59-79, the template compiled from the header where
// it lies -- in libslam_ref.so it is the reference's code, in libslam_gpu_ref.so the swapped-in GPU operator) on a batch
// handed over as arrays: reads = one flat byte buffer + n_reads + 1 offsets (R1 block then R2 block, as
// getPairedSequencesFromFASTQFiles leaves them, src/FASTQsequence.h:111-123), genomes = the index injected through
// ref_slam_index_add_entry.  bench.py times this as `cpu_baseline.kind = "reference"` and the -m gpu tests use it as the
// full-size checker.  Runs in `workdir`: the reference's log() stamps (src/sequenceTools.h:171-179) land in
// <workdir>/log.txt when this is the process's first log() call.  *seconds = wall clock of the call to alignToDatabase
// alone (building the std::vector<MetagenomicFASTQSequence> and flattening the result are outside).  Results are
// malloc'ed (free with ref_slam_free): 56-byte records == oracle/kslam_oracle.h orc_alignment, CIGAR words pooled.
struct ref_slam_alignment {
  uint32_t read, entry;
  int32_t rel;
  uint8_t revcomp, pad;
  uint16_t score;
  int32_t ref_begin, ref_end, query_begin, query_end;
  uint32_t cigar_len, pad2;
  uint64_t cigar_off;
};

int ref_slam_align_to_database(uint64_t n_reads, const char *bases, const uint64_t *offs,
                               const ref_slam_params *p, int32_t report_cigar, void **out, uint64_t *n_out,
                               uint32_t **cigar_pool, uint64_t *n_cigar, double *seconds, const char *workdir) {
  char old[4096];
  if (!getcwd(old, sizeof old)) return 2;
  if (workdir && chdir(workdir) != 0) return 3;
  match = p->match;
  misMatch = p->mismatch;
  gapOpen = p->gap_open;
  gapExtend = p->gap_extend;
  scoreThreshold = p->score_threshold;
  reportCigar = report_cigar != 0;
  pairedData = true;
  if (p->threads > 0) omp_set_num_threads(p->threads);
  SLAM::dropGpuPath();
  int rc = 0;
  std::vector<SLAM::Overlap> overlaps;
  try {
    std::vector<SLAM::MetagenomicFASTQSequence> reads(n_reads);
    for (uint64_t i = 0; i < n_reads; i++) reads[i].bases.assign(bases + offs[i], offs[i + 1] - offs[i]);
    const auto t0 = std::chrono::steady_clock::now();
    overlaps = SLAM::alignToDatabase(reads, g_index);
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  } catch (const std::exception &e) {
    std::cerr << "reference threw: " << e.what() << std::endl;
    rc = 1;
  }
  SLAM::dropGpuPath();
  if (workdir && chdir(old) != 0) abort();
  if (rc) return rc;
  uint64_t nc = 0;
  for (auto &o : overlaps)
    if (o.alignment.cigar) nc += o.alignment.cigarLen;
  ref_slam_alignment *res = static_cast<ref_slam_alignment *>(std::calloc(overlaps.size() + 1, sizeof(ref_slam_alignment)));
  uint32_t *pool = static_cast<uint32_t *>(std::malloc(sizeof(uint32_t) * (nc + 1)));
  if (!res || !pool) return 4;
  uint64_t off = 0;
  for (size_t i = 0; i < overlaps.size(); i++) {
    const SLAM::Overlap &o = overlaps[i];
    ref_slam_alignment &r = res[i];
    r.read = o.readPosInArray;
    r.entry = o.entryPosInArray;
    r.rel = o.relativePosition;
    r.revcomp = o.revComp;
    r.score = o.alignment.sw_score;
    r.ref_begin = o.alignment.ref_begin;
    r.ref_end = o.alignment.ref_end;
    r.query_begin = o.alignment.query_begin;
    r.query_end = o.alignment.query_end;
    r.cigar_off = off;
    r.cigar_len = o.alignment.cigar ? o.alignment.cigarLen : 0;
    if (r.cigar_len) {
      std::memcpy(pool + off, o.alignment.cigar, sizeof(uint32_t) * r.cigar_len);
      off += r.cigar_len;
    }
  }
  *out = res;
  *n_out = overlaps.size();
  *cigar_pool = pool;
  *n_cigar = nc;
  return 0;
}
void ref_slam_free(void *q) { std::free(q); }
}
