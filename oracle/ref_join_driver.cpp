// ref_join_driver.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref/libjoin_ref.so).
//
// Thin extern "C" driver around the REAL reference join / dedupe / aligner
// code, compiled from the reference sources where they lie
// (-I/root/reference/src).  Nothing of the reference is copied into the repo.
//
// What is real here (compiled unmodified from /root/reference/src):
//   KMer.h           getKMers_parallel, sortKMers
//   Overlap.h        the WHOLE header: OverlapTemp, Overlap, overlapSort,
//                    overlapEqual, processPileUp, findOverlaps,
//                    findOverlaps_parallel, screenOverlapsByScoreThreshold
//   ParallelTools.h  parallelize / getStartPositions / parallelForEachWithSplit
//   ssw.c            (linked: the Makefile compiles it beside this file)
//   ssw_cpp.cpp      Aligner, TranslateBase, BuildSwScoreMatrix, SetFlag,
//                    ConvertAlignment, Aligner::Align
//
// Two build-time accommodations, both made by oracle/Makefile in a mktemp
// directory that is deleted after the compile (nothing lands in the repo or
// in oracle/_ref except the .so):
//   (1) src/ssw_cpp.h line 7 is `#include <boost/optional.hpp>`, which this
//       image lacks and which NOTHING in ssw_cpp.h / ssw_cpp.cpp uses.  The
//       Makefile writes `sed 7d src/ssw_cpp.h` as ssw_cpp_noboost.h; this TU
//       includes it first, so Overlap.h's own `#include "ssw_cpp.h"` finds
//       the original, sees the include guard already defined and skips it.
//       ssw_cpp.cpp is compiled from stdin with the same header named
//       ssw_cpp.h on the include path.  No stand-in for Boost is written.
//   (2) src/GenbankTools.h includes Boost serialisation + progress headers
//       (lines 24-27, 31) for its archive writer/reader and the FASTA / GBFF
//       database builders.  The Makefile writes
//       `sed -n '18,23p;28,30p;32,200p;206,219p'` (+ the closing `}` and
//       `#endif`) as GenbankTools_noboost.h: the include guard, the non-Boost
//       includes, the classes CDS / Gene / GenbankEntry / GenbankIndex with
//       every member except writeIndexToBoostSerial (201-205), getGene, and
//       GenbankIndex::getKMers -- the reference's own lines, none written
//       here.  It keeps the guard GENBANKTOOLS_H_, so when SmithWaterman.h
//       (included WHOLE) asks for "GenbankTools.h" the original is skipped.
//       What this leaves out of the pin: the on-disk archive grammar
//       (getIndexFromBoostSerial), which host/db.cpp restates.
//
// Pins: orc_find_overlaps (a-5, a-6), orc_align (a-9), orc_sw_on_overlap
// (a-8), orc_align_to_database (src/SLAM.h:59-79).
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
#include <climits>
#include <unistd.h>
#include "ssw_cpp_noboost.h"
#include "Globals.h"
#include "sequenceTools.h"
#include "KMer.h"
#include "ParallelTools.h"
#include "FASTQsequence.h"
#include "TaxonomyDatabase.h"
#include "GenbankTools_noboost.h"
#include "Overlap.h"
#include "SmithWaterman.h"

namespace {
struct Seq {
  std::string bases;
};
typedef SLAM::KMerAndData<uint64_t, 32> Rec;
static_assert(sizeof(Rec) == 16, "reference record is 16 bytes");

struct OutOverlap {  // == orc_overlap
  uint32_t read, entry;
  int32_t rel;
  uint8_t revcomp, pad[3];
};
struct OutAlignment {  // == orc_alignment
  uint32_t read, entry;
  int32_t rel;
  uint8_t revcomp, pad;
  uint16_t score;
  int32_t ref_begin, ref_end, query_begin, query_end;
  uint32_t cigar_len, pad2;
  uint64_t cigar_off;
};
struct Params {  // == orc_params
  uint32_t match, mismatch, gap_open, gap_extend, score_threshold;
  int32_t report_cigar;
};

struct Cwd {  // the reference logs through a function-static Log that opens ./log.txt
  char old[4096];
  bool moved = false;
  explicit Cwd(const char *dir) {
    if (dir && getcwd(old, sizeof old) && chdir(dir) == 0) moved = true;
  }
  ~Cwd() {
    if (moved && chdir(old) != 0) abort();
  }
};

void set_globals(const Params *p) {
  match = p->match;
  misMatch = p->mismatch;
  gapOpen = p->gap_open;
  gapExtend = p->gap_extend;
  scoreThreshold = p->score_threshold;
  reportCigar = p->report_cigar != 0;
}

std::vector<Seq> make_seqs(uint64_t n, const char *const *s, const uint64_t *lens) {
  std::vector<Seq> v(n);
  for (uint64_t i = 0; i < n; i++) v[i].bases.assign(s[i], lens[i]);
  return v;
}
}  // namespace

extern "C" {

// findOverlaps_parallel (src/Overlap.h:277-295) on an already sorted record
// list; read lengths come from reads[].bases.size() as in the reference.
// Returns the deduped count; *n_raw = the pre-sort/unique count, obtained by
// running findOverlaps (src/Overlap.h:230-246) over the whole range.
uint64_t ref_find_overlaps(const void *sorted, uint64_t n, uint64_t n_reads,
                           const uint64_t *read_lens, void *out, uint64_t cap,
                           uint64_t *n_raw, void *out_raw, uint64_t cap_raw,
                           const char *workdir) {
  Cwd cwd(workdir);
  std::vector<Seq> reads(n_reads);
  for (uint64_t i = 0; i < n_reads; i++) reads[i].bases.assign(read_lens[i], 'A');
  std::vector<Rec> v(n);
  if (n) std::memcpy(v.data(), sorted, n * sizeof(Rec));
  {
    // the pre-dedupe list, in the order findOverlaps emits it over the whole range
    std::vector<SLAM::OverlapTemp> raw = SLAM::findOverlaps(v.begin(), v.end(), reads);
    if (n_raw) *n_raw = raw.size();
    if (out_raw && raw.size() <= cap_raw) {
      OutOverlap *o = static_cast<OutOverlap *>(out_raw);
      for (size_t i = 0; i < raw.size(); i++) {
        o[i].read = raw[i].readPosInArray;
        o[i].entry = raw[i].entryPosInArray;
        o[i].rel = raw[i].relativePosition;
        o[i].revcomp = raw[i].revComp;
        o[i].pad[0] = o[i].pad[1] = o[i].pad[2] = 0;
      }
    }
  }
  std::vector<SLAM::OverlapTemp> ov =
      SLAM::findOverlaps_parallel(v.begin(), v.end(), reads);
  if (ov.size() <= cap) {
    OutOverlap *o = static_cast<OutOverlap *>(out);
    for (size_t i = 0; i < ov.size(); i++) {
      o[i].read = ov[i].readPosInArray;
      o[i].entry = ov[i].entryPosInArray;
      o[i].rel = ov[i].relativePosition;
      o[i].revcomp = ov[i].revComp;
      o[i].pad[0] = o[i].pad[1] = o[i].pad[2] = 0;
    }
  }
  return ov.size();
}

// the dedupe alone: __gnu_parallel::sort(overlapSort) + std::unique(overlapEqual)
// exactly as src/Overlap.h:289-291 on a caller-supplied OverlapTemp list
uint64_t ref_sort_unique_overlaps(void *ovs, uint64_t n) {
  OutOverlap *o = static_cast<OutOverlap *>(ovs);
  std::vector<SLAM::OverlapTemp> v;
  v.reserve(n);
  for (uint64_t i = 0; i < n; i++)
    v.emplace_back(o[i].read, o[i].entry, o[i].rel, o[i].revcomp != 0);
  __gnu_parallel::sort(v.begin(), v.end(), SLAM::overlapSort());
  auto it = std::unique(v.begin(), v.end(), SLAM::overlapEqual());
  v.resize(std::distance(v.begin(), it));
  for (size_t i = 0; i < v.size(); i++) {
    o[i].read = v[i].readPosInArray;
    o[i].entry = v[i].entryPosInArray;
    o[i].rel = v[i].relativePosition;
    o[i].revcomp = v[i].revComp;
  }
  return v.size();
}

// Aligner::Align (src/ssw_cpp.cpp:234-283) with the filter the SW driver sets
// (src/SmithWaterman.h:191-197).  query and ref are NUL-terminated ASCII.
// Returns the cigar length (ops copied to cigar_out when they fit).
int32_t ref_aligner_align(const char *query, const char *ref, int32_t ref_len,
                          const Params *p, uint16_t *score, int32_t *ref_begin,
                          int32_t *ref_end, int32_t *query_begin,
                          int32_t *query_end, uint32_t *cigar_out,
                          int32_t cigar_cap) {
  const StripedSmithWaterman::Aligner aligner(p->match, p->mismatch, p->gap_open,
                                              p->gap_extend);
  StripedSmithWaterman::Filter filter;
  filter.report_begin_position = true;
  filter.report_cigar = p->report_cigar != 0;
  filter.score_filter = p->score_threshold;
  StripedSmithWaterman::Alignment a;
  aligner.Align(query, ref, ref_len, filter, &a);
  *score = a.sw_score;
  *ref_begin = a.ref_begin;
  *ref_end = a.ref_end;
  *query_begin = a.query_begin;
  *query_end = a.query_end;
  if (a.cigar && a.cigarLen <= cigar_cap)
    std::memcpy(cigar_out, a.cigar, sizeof(uint32_t) * a.cigarLen);
  return a.cigar ? a.cigarLen : 0;
}

// TranslateBase table (src/ssw_cpp.cpp:11-23) read back through Align's own
// matrix is not observable; expose the 5x5 matrix the Aligner builds instead
// by aligning single characters: score(query=c1, ref=c2).
// (kept minimal: tests use ref_aligner_align with 1-char strings)

// The whole hot path, src/SLAM.h:59-79, with the real pieces:
// getKMersFromReads + getKMers_parallel(entries, gap k/2) + sortKMers +
// findOverlaps_parallel + performSmithWatermanOnRange_parallel.
// Results are malloc'ed; free with ref_free.
int ref_align_to_database(uint64_t n_reads, const char *const *reads_p,
                          const uint64_t *read_lens, uint64_t n_entries,
                          const char *const *entries_p,
                          const uint64_t *entry_lens, const Params *p,
                          void **out, uint64_t *n_out, uint32_t **cigar_pool,
                          uint64_t *n_cigar, const char *workdir) {
  Cwd cwd(workdir);
  set_globals(p);
  std::vector<Seq> reads = make_seqs(n_reads, reads_p, read_lens);
  SLAM::GenbankIndex index;
  index.entries.resize(n_entries);
  for (uint64_t i = 0; i < n_entries; i++)
    index.entries[i].bases.assign(entries_p[i], entry_lens[i]);
  std::vector<Rec> kMers;
  SLAM::getKMersFromReads(reads, kMers);
  index.getKMers<KMerInt, k>(kMers, k / 2);
  SLAM::sortKMers(kMers);
  std::vector<SLAM::OverlapTemp> tmp =
      SLAM::findOverlaps_parallel(kMers.begin(), kMers.end(), reads);
  std::vector<SLAM::Overlap> overlaps;
  overlaps.reserve(tmp.size());
  for (auto &o : tmp) overlaps.emplace_back(o);
  kMers.resize(0);
  kMers.shrink_to_fit();
  SLAM::performSmithWatermanOnRange_parallel(overlaps.begin(), overlaps.end(),
                                             reads, index);
  uint64_t nc = 0;
  for (auto &o : overlaps)
    if (o.alignment.cigar) nc += o.alignment.cigarLen;
  OutAlignment *res =
      static_cast<OutAlignment *>(std::calloc(overlaps.size() + 1, sizeof(OutAlignment)));
  uint32_t *pool = static_cast<uint32_t *>(std::malloc(sizeof(uint32_t) * (nc + 1)));
  if (!res || !pool) return 1;
  uint64_t off = 0;
  for (size_t i = 0; i < overlaps.size(); i++) {
    const SLAM::Overlap &o = overlaps[i];
    OutAlignment &r = res[i];
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

// performSmithWatermanOnRange2 (src/SmithWaterman.h:184-233) on a caller's
// overlap list (must be grouped by read as the deduped list is)
int ref_sw_on_overlaps(const void *ovs, uint64_t n, uint64_t n_reads,
                       const char *const *reads_p, const uint64_t *read_lens,
                       uint64_t n_entries, const char *const *entries_p,
                       const uint64_t *entry_lens, const Params *p, void *out,
                       uint32_t *cigar_pool, uint64_t cigar_cap,
                       uint64_t *n_cigar) {
  set_globals(p);
  std::vector<Seq> reads = make_seqs(n_reads, reads_p, read_lens);
  SLAM::GenbankIndex index;
  index.entries.resize(n_entries);
  for (uint64_t i = 0; i < n_entries; i++)
    index.entries[i].bases.assign(entries_p[i], entry_lens[i]);
  const OutOverlap *in = static_cast<const OutOverlap *>(ovs);
  std::vector<SLAM::Overlap> overlaps;
  overlaps.reserve(n);
  for (uint64_t i = 0; i < n; i++)
    overlaps.emplace_back(in[i].read, in[i].entry, in[i].rel, in[i].revcomp != 0);
  SLAM::performSmithWatermanOnRange2(overlaps.begin(), overlaps.end(), reads, index);
  OutAlignment *res = static_cast<OutAlignment *>(out);
  uint64_t off = 0;
  for (size_t i = 0; i < overlaps.size(); i++) {
    const SLAM::Overlap &o = overlaps[i];
    OutAlignment &r = res[i];
    std::memset(&r, 0, sizeof r);
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
      if (off + r.cigar_len > cigar_cap) return 2;
      std::memcpy(cigar_pool + off, o.alignment.cigar, sizeof(uint32_t) * r.cigar_len);
      off += r.cigar_len;
    }
  }
  *n_cigar = off;
  return 0;
}

void ref_free(void *p) { std::free(p); }
}
