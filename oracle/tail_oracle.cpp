// tail_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/kslam_oracle.h header).
//
// Serial CPU restatement of the reference's host tail between alignToDatabase
// and the taxonomy step, object by object the way
// the reference holds the data (vectors of records carrying copies of their
// overlaps, std::string fields), so that the product's flat, threaded
// implementation is checked against an independently shaped one.
//
// PINNED by the reference's own batch loop (metagenomicAnalysis_Low_Mem,
// src/SLAM.h:159-268, with src/PairedOverlap.h, src/SAM.h and
// src/MetagenomicResults.h included whole) compiled in place as
// oracle/_ref/libslam_ref.so (oracle/ref_slam_driver.cpp) and run on real
// files: tests/test_reference_loop.py compares the SAM text (header included)
// byte for byte over paired / single-end data, with and without
// pseudo-assembly, --sam-xa, --num-alignments, --min-alignment-score and
// several batch sizes; tests/golden/slam_loop.npz holds the reference's files
// for machines without /root/reference.  tests/test_tail.py additionally checks
// the SAM text against the SAM definition itself (CIGAR + MD re-create the
// reference window, NM equals the edit count) and hand-worked pairing cases.
//
// Where the reference's result depends on an unstable parallel sort the order
// is fixed here exactly as include/kslam_tail.h states (ties keep input order).
// Everywhere else the same libstdc++ algorithm (std::sort, std::remove_if,
// std::find_if) is applied to the same element order with the same comparator,
// because unstable std::sort on partial keys decides which alignments survive.
#include <algorithm>
#include <cmath>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

#include "../include/kslam_tail.h"

namespace {

struct OvRec {  // Overlap, src/Overlap.h:53-74, + where it came from
  uint32_t read = 0, entry = 0;
  int32_t rel = 0;
  bool revcomp = false;
  uint16_t score = 0;
  int32_t ref_begin = 0, ref_end = 0, query_begin = 0, query_end = 0;
  const uint32_t *cigar = nullptr;
  uint32_t cigar_len = 0;
  uint32_t src = KSLAM_NO_OVERLAP;
};

struct PairRec {  // PairedOverlap, src/PairedOverlap.h:32-57
  uint32_t combined = 0, entry = 0;
  int ref_start = 0, ref_end = 0;
  uint32_t insert = 0;
  bool has1 = false, has2 = false;
  OvRec o1, o2;
};

struct ReadPair {  // ReadPairAndOverlaps, src/PairedOverlap.h:62-75
  uint32_t r1 = 0, r2 = 0;
  std::vector<PairRec> pairs;
};

struct Reads {
  const kslam_reads_view *v;
  size_t size() const { return v->n_reads; }
  std::string bases(uint32_t i) const {
    return std::string(v->bases + v->bases_off[i], v->bases + v->bases_off[i + 1]);
  }
  size_t length(uint32_t i) const { return v->bases_off[i + 1] - v->bases_off[i]; }
  std::string quality(uint32_t i) const {
    return std::string(v->quality + v->quality_off[i], v->quality + v->quality_off[i + 1]);
  }
  std::string id(uint32_t i) const {
    return std::string(v->ids + v->ids_off[i], v->ids + v->ids_off[i + 1]);
  }
};

PairRec single(const OvRec &o, bool is_r1) {
  PairRec p;
  p.combined = o.score;
  p.entry = o.entry;
  p.ref_start = o.ref_begin;
  p.ref_end = o.ref_end;
  p.insert = 0;
  p.has1 = is_r1;
  p.has2 = !is_r1;
  (is_r1 ? p.o1 : p.o2) = o;
  return p;
}

// makePair, src/PairedOverlap.h:107-125.  r1_first: the pair lies R1....R2.
PairRec both(const OvRec &a, const OvRec &b, bool r1_first, const Reads &reads) {
  PairRec p;
  p.combined = (uint16_t)(a.score + b.score);
  p.entry = b.entry;
  p.ref_start = std::min(a.ref_begin, b.ref_begin);
  p.ref_end = std::max(a.ref_end, b.ref_end);
  p.insert = r1_first ? (uint32_t)(b.rel - a.rel + (int64_t)reads.length(b.read))
                      : (uint32_t)(a.rel - b.rel + (int64_t)reads.length(a.read));
  p.has1 = p.has2 = true;
  p.o1 = a;
  p.o2 = b;
  return p;
}

// getPairsFromRead, src/PairedOverlap.h:132-242: one (read pair, entry) run.
// slot[mate][strand] = the latest overlap of that mate and strand; an overlap
// pairs with the latest one of the other mate on the opposite strand.
size_t pair_run(const std::vector<OvRec> &ov, size_t first, const Reads &reads,
                std::vector<PairRec> &out) {
  const size_t last = ov.size();
  if (first == last) return last;
  const uint32_t mid = reads.size() / 2;
  const uint32_t pair_id = ov[first].read % mid, entry = ov[first].entry;
  size_t slot[2][2] = {{last, last}, {last, last}};
  bool used[2][2] = {{false, false}, {false, false}};
  size_t cur = first;
  for (; cur != last && ov[cur].read % mid == pair_id && ov[cur].entry == entry; cur++) {
    const int m = ov[cur].read < mid ? 0 : 1, s = ov[cur].revcomp ? 1 : 0;
    if (!used[m][s] && slot[m][s] != last) out.push_back(single(ov[slot[m][s]], m == 0));
    slot[m][s] = cur;
    used[m][s] = false;
    const size_t mate = slot[1 - m][1 - s];
    if (mate != last) {
      out.push_back(m == 0 ? both(ov[cur], ov[mate], false, reads)
                           : both(ov[mate], ov[cur], true, reads));
      used[m][s] = true;
      used[1 - m][1 - s] = true;
    }
  }
  const int flush[4][2] = {{1, 0}, {1, 1}, {0, 0}, {0, 1}};
  for (auto &f : flush)
    if (!used[f[0]][f[1]] && slot[f[0]][f[1]] != last)
      out.push_back(single(ov[slot[f[0]][f[1]]], f[0] == 0));
  return cur;
}

// getPairedOverlaps, src/PairedOverlap.h:243-270 (sort) + the per-run loop
std::vector<PairRec> pair_all(std::vector<OvRec> &ov, const Reads &reads) {
  const uint32_t mid = reads.size() / 2;
  std::stable_sort(ov.begin(), ov.end(), [&](const OvRec &a, const OvRec &b) {
    if (a.read % mid != b.read % mid) return a.read % mid < b.read % mid;
    if (a.entry != b.entry) return a.entry < b.entry;
    return a.rel < b.rel;
  });
  std::vector<PairRec> out;
  size_t at = 0;
  while (at != ov.size()) at = pair_run(ov, at, reads, out);
  return out;
}

// getPerReadOverlaps, src/PairedOverlap.h:437-470
std::vector<ReadPair> group_pairs(const std::vector<PairRec> &pairs, uint32_t mid) {
  std::vector<ReadPair> out;
  ReadPair acc;
  uint32_t at = 0;
  for (auto &p : pairs) {
    uint32_t here = p.has1 ? p.o1.read : p.o2.read - mid;
    if (here != at) {
      if (!acc.pairs.empty()) {
        out.push_back(acc);
        acc.pairs.clear();
      }
      at = here;
    }
    acc.pairs.push_back(p);
    acc.r1 = here;
    acc.r2 = here + mid;
  }
  if (!acc.pairs.empty()) out.push_back(acc);
  return out;
}

// single end: getPerReadOverlaps src/Overlap.h:303-327 +
// getDummyAlignmentPairsFromSingleEndReads src/PairedOverlap.h:280-298
std::vector<ReadPair> group_single(const std::vector<OvRec> &ov) {
  std::vector<ReadPair> out;
  ReadPair acc;
  uint32_t at = 0;
  for (auto &o : ov) {
    if (o.read != at) {
      if (!acc.pairs.empty()) {
        out.push_back(acc);
        acc.pairs.clear();
      }
      at = o.read;
    }
    acc.pairs.push_back(single(o, true));
    acc.r1 = o.read;
    acc.r2 = 0;
  }
  if (!acc.pairs.empty()) out.push_back(acc);
  return out;
}

// getMaxAllowedInsertSize, src/PairedOverlap.h:314-360
uint32_t max_insert(const std::vector<ReadPair> &rps, uint64_t *n_sizes) {
  std::vector<int32_t> sz;
  for (auto &rp : rps)
    for (auto &p : rp.pairs)
      if (p.insert != 0) sz.push_back(p.insert);
  *n_sizes = sz.size();
  if (sz.empty()) return UINT32_MAX;
  std::sort(sz.begin(), sz.end());
  int32_t limit = 0;
  for (int i = 0; i < 99; i++) {
    if (sz[floor(sz.size() * (i + 1) / 100.0)] - sz[floor(sz.size() * (i) / 100.0)] > 1000) {
      limit = sz[floor(sz.size() * (i) / 100)];
      break;
    }
  }
  int32_t lq = sz[floor(sz.size() * 0.25)];
  int32_t uq = sz[floor(sz.size() * 0.75)];
  int32_t lo = 0;
  int32_t hi = uq + 2 * (uq - lq);
  if (limit) hi = limit;
  if (hi == 0) hi = INT32_MAX;
  sz.erase(std::remove_if(sz.begin(), sz.end(), [&](int32_t v) { return v < lo || v > hi; }),
           sz.end());
  double sum = std::accumulate(sz.begin(), sz.end(), 0.0);
  double mean = sum / sz.size();
  double sq = std::inner_product(sz.begin(), sz.end(), sz.begin(), 0.0);
  double sd = std::sqrt(sq / sz.size() - mean * mean);
  double r = floor(mean + 6 * sd);
  return std::isnan(r) ? UINT_MAX : (uint32_t)r;
}

// screenPairedAlignmentsByInsertSize(..., replace = true), src/PairedOverlap.h:396-436
void screen_insert(std::vector<ReadPair> &rps, uint32_t limit) {
  for (auto &rp : rps) {
    auto &v = rp.pairs;
    std::sort(v.begin(), v.end(),
              [](const PairRec &a, const PairRec &b) { return a.insert < b.insert; });
    size_t cut = std::find_if(v.begin(), v.end(),
                              [&](const PairRec &p) { return p.insert > limit; }) -
                 v.begin();
    size_t old_end = v.size();
    v.reserve(old_end + (old_end - cut));
    for (size_t i = cut; i < old_end; i++) {
      v.push_back(single(v[i].o1, true));
      v.back().entry = v[i].entry;
      PairRec &c = v[i];
      c.combined = c.o2.score;
      c.has1 = false;
      c.insert = 0;
      c.o1 = OvRec();
      c.ref_start = c.o2.ref_begin;
      c.ref_end = c.o2.ref_end;
    }
  }
}

// screenPairedAlignmentsByScore, src/PairedOverlap.h:361-390
void screen_score(std::vector<ReadPair> &rps, double fraction) {
  for (auto &rp : rps) {
    auto &v = rp.pairs;
    if (v.empty()) continue;
    std::sort(v.begin(), v.end(),
              [](const PairRec &a, const PairRec &b) { return a.combined > b.combined; });
    unsigned top = v[0].combined;
    auto cut = std::find_if(v.begin(), v.end(),
                            [&](const PairRec &p) { return p.combined < top * fraction; });
    v.erase(cut, v.end());
  }
}

// pseudoAssembly, src/PairedOverlap.h:480-582
void pseudo_assembly(std::vector<ReadPair> &rps) {
  struct Item {
    int start, stop;
    PairRec *p;
  };
  std::unordered_map<uint32_t, std::vector<Item>> by_entry;
  for (auto &rp : rps)
    for (auto &p : rp.pairs) by_entry[p.entry].push_back(Item{p.ref_start, p.ref_end, &p});
  for (auto &kv : by_entry) {
    auto &v = kv.second;
    std::sort(v.begin(), v.end(), [](const Item &a, const Item &b) { return a.start < b.start; });
    size_t chain = 0;
    int reach = -1000000;
    uint32_t bases = 0;
    double per_base = 0;
    auto close = [&](size_t end) {
      long n = (long)(end - chain);
      if (n > 1) {
        double length = reach - v[chain].start;
        double coverage = bases / length;
        double avg = per_base / n;
        double score = coverage * avg * length;
        for (size_t k = chain; k < end; k++) v[k].p->combined = score;
      }
    };
    for (size_t i = 0; i < v.size(); i++) {
      PairRec *p = v[i].p;
      uint32_t span = abs(p->ref_end - p->ref_start);
      if (v[i].start > reach - 20) {
        close(i);
        chain = i;
        reach = v[i].stop;
        per_base = p->combined * 1.0 / abs(p->ref_end - p->ref_start);
        bases = span;
      } else {
        if (v[i].stop > reach) reach = v[i].stop;
        per_base += p->combined * 1.0 / abs(p->ref_end - p->ref_start);
        bases += span;
      }
    }
    close(v.size());
  }
}

// ---------------------------------------------------------------- SAM ---------
std::string revcomp_text(const std::string &s) {  // src/sequenceTools.h:77-97
  std::string r(s.rbegin(), s.rend());
  for (auto &c : r) {
    if (c == 'A') c = 'T';
    else if (c == 'T') c = 'A';
    else if (c == 'C') c = 'G';
    else if (c == 'G') c = 'C';
  }
  return r;
}

struct Index {
  const kslam_index_view *v;
  const char *bases(uint32_t e) const { return v->bases + v->bases_off[e]; }
  std::string locus(uint32_t e) const {
    return std::string(v->locus_tag + v->locus_tag_off[e], v->locus_tag + v->locus_tag_off[e + 1]);
  }
  // GenbankEntry::getGene, src/GenbankTools.h:170-185; returns gene number or -1
  int64_t gene(uint32_t e, int32_t start, int32_t stop) const {
    if (!v->n_genes) return -1;
    int64_t best = -1;
    int32_t widest = 0;
    for (uint64_t g = v->gene_first[e]; g < v->gene_first[e + 1]; g++) {
      int32_t shared = std::min<int>(stop, v->gene_stop[g]) - std::max<int>(start, v->gene_start[g]);
      if (shared > widest) {
        best = (int64_t)g;
        widest = shared;
      }
    }
    return best;
  }
  static std::string col(const char *t, const uint64_t *off, uint64_t i) {
    return std::string(t + off[i], t + off[i + 1]);
  }
};

struct Diff {  // SequenceDifference, src/SAM.h:26-32
  std::string cigar, md;
  uint32_t nm = 0;
  double logp = 0;
};

const std::vector<double> &match_table() {  // src/SAM.h:33-40
  static std::vector<double> t = [] {
    std::vector<double> v;
    v.push_back(std::log10(1.0 - std::pow(10.0, 1.0 / -10.0)));
    for (int i = 1; i < 100; i++) v.push_back(std::log10(1.0 - std::pow(10.0, i / -10.0)));
    return v;
  }();
  return t;
}
const std::vector<double> &mismatch_table() {  // src/SAM.h:41-48
  static std::vector<double> t = [] {
    std::vector<double> v;
    v.push_back(1 / -10.0);
    for (int i = 1; i < 100; i++) v.push_back(i / -10.0);
    return v;
  }();
  return t;
}

// getCigarAndMD, src/SAM.h:101-237
Diff cigar_and_md(const OvRec &o, const Reads &reads, const Index &index) {
  Diff d;
  std::vector<std::string> parts;
  const char *ref = index.bases(o.entry);
  std::string query = o.revcomp ? revcomp_text(reads.bases(o.read)) : reads.bases(o.read);
  std::string qual = reads.quality(o.read);
  if (o.revcomp) std::reverse(qual.begin(), qual.end());
  if (!o.cigar) return d;
  int rp = o.ref_begin, qp = 0;
  if (o.query_begin > 0) {
    d.cigar += std::to_string(o.query_begin) + "S";
    qp += o.query_begin;
  }
  for (uint32_t k = 0; k < o.cigar_len; k++) {
    uint32_t len = o.cigar[k] >> 4, op = o.cigar[k] & 15;
    d.cigar += std::to_string(len);
    if (op == 0) {
      d.cigar.push_back('M');
      int run = 0;
      for (uint32_t i = 0; i < len; i++, rp++, qp++) {
        if (ref[rp] == query[qp]) {
          run++;
          d.logp += match_table()[qual[qp] - 33];
        } else {
          d.nm++;
          if (run) parts.push_back(std::to_string(run));
          parts.push_back(std::string(1, ref[rp]));
          d.logp += mismatch_table()[qual[qp] - 33];
          run = 0;
        }
      }
      if (run) parts.push_back(std::to_string(run));
    } else if (op == 1) {
      d.cigar.push_back('I');
      d.nm += len;
      qp += len;
    } else if (op == 2) {
      d.cigar.push_back('D');
      parts.push_back("^");
      std::string gone;
      for (uint32_t i = 0; i < len; i++, rp++) {
        gone.push_back(ref[rp]);
        d.nm++;
      }
      parts.push_back(gone);
    }
  }
  int tail = (int)query.size() - o.query_end - 1;
  if (tail > 0) d.cigar += std::to_string(tail) + "S";
  bool after_del = false;
  for (size_t i = 0; i < parts.size();) {
    if (parts[i] == "^") {
      d.md += parts[i++];
      d.md += parts[i++];
      after_del = true;
    } else if (isdigit((unsigned char)parts[i][0])) {
      int total = 0;
      while (i < parts.size() && isdigit((unsigned char)parts[i][0])) total += std::stoi(parts[i++]);
      d.md += std::to_string(total);
      after_del = false;
    } else {
      if (after_del) {
        d.md += "0";
        after_del = false;
      }
      d.md += parts[i++];
    }
  }
  return d;
}

struct SamRow {  // SAMEntry, src/SAM.h:238-277
  std::string qname, rname, cigar = "*", rnext = "=", md, xg, xp, xr;
  uint32_t pos = 0, pnext = 0, nm = 0, xo = 0, xt = 0;
  uint8_t mapq = 255;
  int32_t tlen = 0;
  bool multi = false, all_aligned = false, unmapped = false, next_unmapped = false, rc = false,
       next_rc = false, first = false, secondary = true;
  uint16_t as = 0, xs = 0;
  double prob = 0;
};

struct Cfg {
  bool paired, report_cigar, sam_xa;
  uint32_t n_sam;
};

uint16_t flag_of(const SamRow &r, const Cfg &c) {  // src/SAM.h:306-323
  uint16_t f = 0;
  if (r.multi) f |= 0x1;
  if (r.all_aligned) f |= 0x2;
  if (r.unmapped) f |= 0x4;
  if (r.next_unmapped) f |= 0x8;
  if (r.rc) f |= 0x10;
  if (r.next_rc) f |= 0x20;
  if (c.paired) f |= r.first ? 0x40 : 0x80;
  if (r.secondary) f |= 0x100;
  return f;
}

std::string line_of(const SamRow &r, const Cfg &c) {  // src/SAM.h:278-305
  std::string out = r.qname + '\t' + std::to_string(flag_of(r, c)) + '\t' + r.rname + '\t' +
                    std::to_string(r.pos) + '\t' + std::to_string(r.mapq) + '\t' +
                    (c.report_cigar ? r.cigar : "*") + '\t' + r.rnext + '\t' +
                    std::to_string(r.pnext) + '\t' + std::to_string(r.tlen) + "\t*\t*";
  if (r.unmapped) return out;
  if (c.report_cigar) out += "\tMD:Z:" + r.md;
  out += "\tAS:i:" + std::to_string(r.as) + "\tXS:i:" + std::to_string(r.xs) +
         "\tNM:i:" + std::to_string(r.nm) + "\tX0:i:" + std::to_string(r.xo);
  if (r.xt != 0) out += "\tXT:i:" + std::to_string(r.xt);
  if (!r.xg.empty()) out += "\tXG:Z:" + r.xg;
  if (!r.xp.empty()) out += "\tXP:Z:" + r.xp;
  if (!r.xr.empty()) out += "\tXR:Z:\"" + r.xr + "\"";
  return out;
}

void init_row(SamRow &r, const OvRec &o, const Reads &reads, const Index &index) {  // src/SAM.h:339-351
  Diff d = cigar_and_md(o, reads, index);
  r.cigar = d.cigar;
  r.md = d.md;
  r.nm = d.nm;
  r.prob = std::pow(10, d.logp);
  r.rname = index.locus(o.entry);
  r.pos = o.ref_begin + 1;
  r.as = o.score;
}

// getSAMFromPair, src/SAM.h:352-433
std::pair<SamRow, SamRow> rows_of(const PairRec &p, const Reads &reads, const Index &index,
                                  const Cfg &c) {
  SamRow a, b;
  a.first = true;
  b.first = false;
  int64_t g = index.gene(p.entry, p.ref_start, p.ref_end);
  if (g >= 0) {
    a.xg = b.xg = Index::col(index.v->gene_name, index.v->gene_name_off, g);
    a.xp = b.xp = Index::col(index.v->protein_id, index.v->protein_id_off, g);
    a.xr = b.xr = Index::col(index.v->product, index.v->product_off, g);
  }
  a.xt = b.xt = index.v->taxonomy_id[p.entry];
  bool conventional = true;
  if (c.paired) a.multi = b.multi = true;
  if (p.has1 && p.has2) {
    a.all_aligned = b.all_aligned = true;
    conventional = p.o1.ref_begin < p.o2.ref_begin;
    if (p.o1.revcomp) a.rc = b.next_rc = true;
    if (p.o2.revcomp) b.rc = a.next_rc = true;
  } else if (p.has1) {
    a.next_unmapped = true;
    b.unmapped = true;
    if (p.o1.revcomp) a.rc = true;
  } else if (p.has2) {
    b.next_unmapped = true;
    a.unmapped = true;
    if (p.o2.revcomp) b.rc = true;
  }
  if (p.has1) init_row(a, p.o1, reads, index);
  if (p.has2) init_row(b, p.o2, reads, index);
  a.pnext = b.pos;
  b.pnext = a.pos;
  if (!p.has1) {
    a.rname = b.rname;
    a.pos = b.pos;
    b.pnext = b.pos;
    a.pnext = b.pos;
  }
  if (!p.has2) {
    b.rname = a.rname;
    b.pos = a.pos;
    a.pnext = a.pos;
    b.pnext = a.pos;
  }
  if (!c.paired) {
    a.rnext = "*";
    a.pnext = 0;
    a.next_unmapped = false;
  }
  int32_t tlen = p.ref_end - p.ref_start + 1;
  if (!(p.has1 || p.has2)) tlen = 0;
  if (!conventional) tlen *= -1;
  a.tlen = tlen;
  b.tlen = tlen * -1;
  a.xs = b.xs = p.combined;
  return {a, b};
}

// ceil(-10 log10(t)) stored into a uint8_t, src/SAM.h:502-506.  When the sum
// of probabilities is 0 the reference divides 0 by 0; converting the resulting
// NaN to an integer is undefined in C++ and yields 0 in the low byte with
// x86-64 gcc (cvttsd2si -> 0x80000000): that observable value is kept.
uint8_t mapq_of(double prob, double sum) {
  double t = 1.0 - prob / sum;
  if (t <= 0.00001) t = 0.00001;
  double q = ceil(-10.0 * std::log10(t));
  if (std::isnan(q)) return 0;
  return (uint8_t)q;
}

// writeSAMOutputPairs, src/SAM.h:443-512
void write_pairs(std::string &out, ReadPair &rp, const Reads &reads, const Index &index,
                 const Cfg &c) {
  std::sort(rp.pairs.begin(), rp.pairs.end(),
            [](const PairRec &a, const PairRec &b) { return a.combined > b.combined; });
  std::vector<std::pair<SamRow, SamRow>> rows;
  uint32_t hits1 = 0, hits2 = 0;
  for (auto &p : rp.pairs) {
    if (p.has1) hits1++;
    if (p.has2) hits2++;
    rows.push_back(rows_of(p, reads, index, c));
    if (rows.size() >= c.n_sam) break;
  }
  if (rows.empty()) return;
  double sum1 = 0, sum2 = 0;
  for (auto &r : rows) {
    r.first.qname = reads.id(rp.r1);
    r.second.qname = reads.id(rp.r2);
    sum1 += r.first.prob;
    sum2 += r.second.prob;
    r.first.xo = hits1;
    r.second.xo = hits2;
  }
  rows[0].first.secondary = false;
  rows[0].second.secondary = false;
  for (auto &r : rows) {
    r.first.mapq = mapq_of(r.first.prob, sum1);
    r.second.mapq = mapq_of(r.second.prob, sum2);
    out += line_of(r.first, c) + "\n";
    if (c.paired) out += line_of(r.second, c) + "\n";
    if (c.sam_xa) break;
  }
}

thread_local std::string g_err;

std::vector<OvRec> load(const kslam_overlap *ov, uint64_t n, const uint32_t *pool, uint32_t thr) {
  std::vector<OvRec> v;
  for (uint64_t i = 0; i < n; i++) {
    if (ov[i].score < thr) continue;  // screenOverlapsByScoreThreshold, src/Overlap.h:329-341
    OvRec o;
    o.read = ov[i].read;
    o.entry = ov[i].entry;
    o.rel = ov[i].rel;
    o.revcomp = ov[i].revcomp != 0;
    o.score = ov[i].score;
    o.ref_begin = ov[i].ref_begin;
    o.ref_end = ov[i].ref_end;
    o.query_begin = ov[i].query_begin;
    o.query_end = ov[i].query_end;
    o.cigar = (pool && ov[i].cigar_len) ? pool + ov[i].cigar_off : nullptr;
    o.cigar_len = ov[i].cigar_len;
    o.src = (uint32_t)i;
    v.push_back(o);
  }
  return v;
}

// A sub-sample of a batch is run with the insert-size limit of the WHOLE batch (getMaxAllowedInsertSize is a
// batch-global statistic, src/PairedOverlap.h:314-360; everything after it is per read pair or per entry):
// orc_tail_force_insert_limit(limit) makes the next calls use `limit` instead of computing it; -1 switches back.
static int64_t g_forced_insert_limit = -1;

// src/SLAM.h:102-128
std::vector<ReadPair> run_tail(const kslam_tail_params *p, const Reads &reads,
                               std::vector<OvRec> &ov, kslam_tail_stats *st) {
  uint32_t stages = p->stages ? p->stages : KSLAM_TAIL_ALL;
  std::vector<ReadPair> rps;
  if (p->paired) {
    auto pairs = pair_all(ov, reads);
    st->n_paired_initial = pairs.size();
    rps = group_pairs(pairs, reads.size() / 2);
    if (stages & KSLAM_TAIL_INSERT_SCREEN) {
      st->max_insert_size = max_insert(rps, &st->n_insert_sizes);
      if (g_forced_insert_limit >= 0) st->max_insert_size = (uint32_t)g_forced_insert_limit;
      screen_insert(rps, st->max_insert_size);
    }
    if (stages & KSLAM_TAIL_SCORE_SCREEN) screen_score(rps, p->score_fraction);
  } else {
    rps = group_single(ov);
    for (auto &rp : rps) st->n_paired_initial += rp.pairs.size();
    if (stages & KSLAM_TAIL_SCORE_SCREEN) screen_score(rps, p->score_fraction);
  }
  if (p->pseudo_assembly && (stages & KSLAM_TAIL_PSEUDO_ASM)) {
    pseudo_assembly(rps);
    screen_score(rps, p->score_fraction);
  }
  st->n_read_pairs = rps.size();
  for (auto &rp : rps) st->n_paired_final += rp.pairs.size();
  return rps;
}

char *dup_text(const std::string &s) {
  char *t = (char *)malloc(s.size() + 1);
  memcpy(t, s.data(), s.size());
  t[s.size()] = 0;
  return t;
}

}  // namespace

extern "C" {

const char *orc_tail_last_error(void) { return g_err.c_str(); }

void orc_tail_force_insert_limit(int64_t limit) { g_forced_insert_limit = limit; }

int orc_tail_pairs(const kslam_tail_params *params, const kslam_reads_view *reads_v,
                   const kslam_overlap *overlaps, uint64_t n_overlaps,
                   kslam_read_pair **read_pairs, uint64_t *n_read_pairs,
                   kslam_paired_overlap **pairs, uint64_t *n_pairs, kslam_tail_stats *stats) {
  kslam_tail_stats st;
  memset(&st, 0, sizeof st);
  Reads reads{reads_v};
  auto ov = load(overlaps, n_overlaps, nullptr, params->score_threshold);
  st.n_overlaps_in = n_overlaps;
  st.n_overlaps_screened = ov.size();
  auto rps = run_tail(params, reads, ov, &st);
  *n_read_pairs = rps.size();
  *n_pairs = st.n_paired_final;
  *read_pairs = (kslam_read_pair *)malloc(sizeof(kslam_read_pair) * (rps.size() + 1));
  *pairs = (kslam_paired_overlap *)malloc(sizeof(kslam_paired_overlap) * (st.n_paired_final + 1));
  uint64_t at = 0;
  for (size_t i = 0; i < rps.size(); i++) {
    (*read_pairs)[i] = kslam_read_pair{rps[i].r1, rps[i].r2, at, rps[i].pairs.size()};
    for (auto &p : rps[i].pairs)
      (*pairs)[at++] = kslam_paired_overlap{p.combined, p.entry, p.ref_start, p.ref_end, p.insert,
                                            p.has1 ? p.o1.src : KSLAM_NO_OVERLAP,
                                            p.has2 ? p.o2.src : KSLAM_NO_OVERLAP, 0};
  }
  if (stats) *stats = st;
  return 0;
}

int orc_tail_sam(const kslam_tail_params *params, const kslam_reads_view *reads_v,
                 const kslam_index_view *index_v, const kslam_overlap *overlaps,
                 uint64_t n_overlaps, const uint32_t *cigar_pool, uint64_t n_cigar, char **text,
                 uint64_t *text_len, kslam_tail_stats *stats) {
  (void)n_cigar;
  kslam_tail_stats st;
  memset(&st, 0, sizeof st);
  Reads reads{reads_v};
  Index index{index_v};
  auto ov = load(overlaps, n_overlaps, cigar_pool, params->score_threshold);
  st.n_overlaps_in = n_overlaps;
  st.n_overlaps_screened = ov.size();
  auto rps = run_tail(params, reads, ov, &st);
  Cfg c{params->paired != 0, params->report_cigar != 0, params->sam_xa != 0,
        params->num_sam_alignments};
  std::string out;
  for (auto &rp : rps) write_pairs(out, rp, reads, index, c);
  *text = dup_text(out);
  *text_len = out.size();
  if (stats) *stats = st;
  return 0;
}

// getHeader, src/SAM.h:513-531
int orc_sam_header(const kslam_index_view *index_v, const char *command_line, char **text,
                   uint64_t *text_len) {
  Index index{index_v};
  std::string h = "@HD\tVN:1.0\tSO:unsorted\n";
  for (uint64_t e = 0; e < index_v->n_entries; e++) {
    h += "@SQ\tSN:" + index.locus(e) + "\tLN:" +
         std::to_string(index_v->bases_off[e + 1] - index_v->bases_off[e]);
    if (index_v->taxonomy_id[e]) h += "\tSP:" + std::to_string(index_v->taxonomy_id[e]);
    h += "\n";
  }
  h += "@PG\tID:SLAM\tPN:SLAM\tVN:1.0\tCL:\"" + std::string(command_line) + "\"\n";
  *text = dup_text(h);
  *text_len = h.size();
  return 0;
}

void orc_tail_free(void *p) { free(p); }
}
