// samtext.hip -- the SAM records and the <out>_PerRead lines of a batch, written on the GPU.
//
// Replaces, in the reference (citations into /root/reference/), what host/tail.cpp's sam_stage and host/taxonomy.cpp's
// kslam_tail_classify do on the CPUs:
//   writeSAMOutputPairs              src/SAM.h:443-512    per read pair: std::sort by combinedScore descending, the first
//                                                         --num-alignments records, mapping qualities, the lines
//   getSAMFromPair / SAMEntry::init  src/SAM.h:339-433    flags, rname / pos / pnext / tlen, XS, X0
//   SAMEntry::getEntry               src/SAM.h:278-305    the line itself
//   getResultFromPairedOverlaps      src/MetagenomicResults.h:88-112   LCA of the entries' taxonomy ids
//   getLowestCommonAncestor          src/TaxonomyDatabase.h:185-223
//   writePerReadResults              src/MetagenomicResults.h:455-463  "identifier \t taxonomy id \n"
//
// Everything a line needs is in HBM when the per-row walk (details.hip) has run: the overlap records with their final
// coordinates, the CIGAR pool, NM / MD / log-probability per row, the alignment-pair records grouped by read pair, the read
// identifiers (fastq_index.hip) and -- uploaded once, kslam_set_sam_annotations -- the index's locus tags, taxonomy ids
// and gene columns and the taxonomy tree.  One thing is NOT computed here: 10^logp and log10 come from the host's libm
// (the mapping quality is ceil(-10 log10(1 - p / sum p)), src/SAM.h:464-499, and its bits are libm's).  So the stage
// runs in two halves around a small host step:
//   sam_plan     per read pair: the reference's sort (gnu_sort.h: libstdc++'s permutation), the rows to report, and for
//                every mate whose mapping quality depends on probabilities (more than one reported row carries it, or
//                10^logp could underflow) the log-probabilities of its rows, compacted
//   <host>       pow / sum in row order / log10 / ceil on those (a few hundred thousand values), one byte back per row
//   sam_format   lengths -> exclusive scan -> the text, one thread per read pair writing its lines back to back
// A mate with a single reported row has quality ceil(-10 log10(1 - p / p)) whatever p is: the host hands that constant
// in (mapq_unique = its libm's value, 50).
//
// HBM-bound byte work (integer / byte paths are not reshaped into GEMMs): a batch of configs[1] is 400 MB of
// text from ~1 GB of inputs.  The writer runs twice over the same code, templated on the sink (CountSink adds lengths,
// ByteSink stores), so length pass and write pass cannot disagree.
#include "common.h"
#include "gnu_sort.h"
#include "samtext.h"

namespace kslam {

namespace {
constexpr uint32_t NONE = KSLAM_NO_OVERLAP;
typedef kslam_paired_overlap Rec;

struct ByScoreDesc {
  __host__ __device__ bool operator()(const Rec &a, const Rec &b) const { return a.combined_score > b.combined_score; }
};

struct CountSink {
  uint64_t n = 0;
  __device__ void ch(uint8_t) { n++; }
  __device__ void bytes(const uint8_t *, uint64_t k) { n += k; }
  __device__ void lit(const char *, uint32_t k) { n += k; }
  __device__ void skip(uint32_t k) { n += k; }
};
// Bytes collect in a 64-bit register and leave as ONE 8-byte store (unaligned stores are fine on gfx950): a line of ~200
// bytes is ~25 store instructions instead of ~200, and neighbouring threads' lines are neighbours in memory, so the stores
// of a wave fall into a few hundred consecutive cache lines that L2 merges.
struct ByteSink {
  uint8_t *w;
  uint64_t acc = 0;
  uint32_t k = 0;   // bytes waiting in acc
  __device__ explicit ByteSink(uint8_t *at) : w(at) {}
  __device__ void ch(uint8_t c) {
    acc |= (uint64_t)c << (8 * k);
    if (++k == 8) {
      __builtin_memcpy(w, &acc, 8);
      w += 8;
      acc = 0;
      k = 0;
    }
  }
  __device__ void bytes(const uint8_t *s, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) ch(s[i]);
  }
  __device__ void lit(const char *s, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) ch((uint8_t)s[i]);
  }
  __device__ void flush() {
    for (uint32_t i = 0; i < k; i++) w[i] = (uint8_t)(acc >> (8 * i));
    w += k;
    acc = 0;
    k = 0;
  }
};

__device__ inline uint32_t digits_u64(uint64_t v) {
  uint32_t d = 1;
  while (v >= 10) {
    v /= 10;
    d++;
  }
  return d;
}
template <class Sink>
__device__ inline void put_num(Sink &o, uint64_t v);
template <>
__device__ inline void put_num<CountSink>(CountSink &o, uint64_t v) { o.n += digits_u64(v); }
template <>
__device__ inline void put_num<ByteSink>(ByteSink &o, uint64_t v) {
  // the digits most significant first, from a register: up to 8 digits per 64-bit word (numbers here are < 2^32: 10 digits)
  uint32_t d = 0;
  uint64_t lo = 0, hi = 0;   // digit j of the reversed number in byte j
  do {
    const uint64_t q = v / 10;
    const uint64_t c = '0' + (v - q * 10);
    if (d < 8) lo |= c << (8 * d); else hi |= c << (8 * (d - 8));
    v = q;
    d++;
  } while (v);
  for (uint32_t j = d; j-- > 0;) o.ch((uint8_t)((j < 8 ? lo >> (8 * j) : hi >> (8 * (j - 8))) & 0xFF));
}
template <class Sink>
__device__ inline void put_snum(Sink &o, int64_t v) {
  if (v < 0) {
    o.ch('-');
    put_num(o, (uint64_t)(-v));
  } else {
    put_num(o, (uint64_t)v);
  }
}
#define LIT(o, s) (o).lit(s, (uint32_t)(sizeof(s) - 1))

// GenbankEntry::getGene, src/GenbankTools.h:170-185 (host/tail.cpp: best_gene)
__device__ inline int64_t best_gene(const SamAnnot &A, uint32_t e, int32_t start, int32_t stop) {
  if (!A.n_genes) return -1;
  int64_t best = -1;
  int32_t widest = 0;
  for (uint64_t g = A.gene_first[e]; g < A.gene_first[e + 1]; g++) {
    const int32_t shared = min(stop, A.gene_stop[g]) - max(start, A.gene_start[g]);
    if (shared > widest) {
      best = (int64_t)g;
      widest = shared;
    }
  }
  return best;
}

struct Row {   // SAMEntry, src/SAM.h:238-277 (host/tail.cpp: Row)
  bool mapped = false;
  uint32_t rname_entry = 0, pos = 0, pnext = 0, nm = 0, ov = NONE;
  int32_t tlen = 0;
  uint16_t as = 0, xs = 0, flag = 0;
};

// getSAMFromPair, src/SAM.h:352-433 (host/tail.cpp: write_group's row construction), for one alignment-pair record
__device__ inline void make_rows(const Rec &p, const kslam_overlap *ov, bool paired, Row &a, Row &b) {
  const bool has1 = p.r1 != NONE, has2 = p.r2 != NONE;
  uint16_t fa = 0x40, fb = 0x80;
  if (!paired) fa = fb = 0;
  bool a_next_unmapped = false;
  if (paired) {
    fa |= 0x1;
    fb |= 0x1;
  }
  bool conventional = true;
  const kslam_overlap *o1 = has1 ? &ov[p.r1] : nullptr, *o2 = has2 ? &ov[p.r2] : nullptr;
  if (has1 && has2) {
    fa |= 0x2;
    fb |= 0x2;
    conventional = o1->ref_begin < o2->ref_begin;
    if (o1->revcomp) {
      fa |= 0x10;
      fb |= 0x20;
    }
    if (o2->revcomp) {
      fb |= 0x10;
      fa |= 0x20;
    }
  } else if (has1) {
    a_next_unmapped = true;
    fb |= 0x4;
    if (o1->revcomp) fa |= 0x10;
  } else if (has2) {
    fb |= 0x8;
    fa |= 0x4;
    if (o2->revcomp) fb |= 0x10;
  }
  if (has1) {
    a.mapped = true;
    a.ov = p.r1;
    a.rname_entry = o1->entry;
    a.pos = (uint32_t)(o1->ref_begin + 1);
    a.as = o1->score;
  }
  if (has2) {
    b.mapped = true;
    b.ov = p.r2;
    b.rname_entry = o2->entry;
    b.pos = (uint32_t)(o2->ref_begin + 1);
    b.as = o2->score;
  }
  a.pnext = b.pos;
  b.pnext = a.pos;
  if (!has1) {
    a.rname_entry = b.rname_entry;
    a.pos = b.pos;
    b.pnext = b.pos;
    a.pnext = b.pos;
  }
  if (!has2) {
    b.rname_entry = a.rname_entry;
    b.pos = a.pos;
    a.pnext = a.pos;
    b.pnext = a.pos;
  }
  if (!paired) {
    a.pnext = 0;
    a_next_unmapped = false;
  }
  if (a_next_unmapped) fa |= 0x8;
  int32_t tlen = p.ref_end - p.ref_start + 1;
  if (!(has1 || has2)) tlen = 0;
  if (!conventional) tlen *= -1;
  a.tlen = tlen;
  b.tlen = tlen * -1;
  a.xs = b.xs = (uint16_t)p.combined_score;
  a.flag = fa | 0x100;
  b.flag = fb | 0x100;
}

// A row without CIGAR (score under --min-alignment-score, or no SAM file) was not walked: probability 10^0, NM 0, no MD
// (host/tail.cpp: cigar_and_md returns before it looks at the details)
__device__ inline bool walked(const SamInputs &in, uint32_t row) { return in.pool && in.det && in.ov[row].cigar_len != 0; }
__device__ inline double row_logp(const SamInputs &in, uint32_t row) { return walked(in, row) ? in.det[row].logp : 0.0; }
__device__ inline uint32_t row_flags(const SamInputs &in, uint32_t row) { return walked(in, row) ? in.det[row].flags : 0u; }

// SAMEntry::getEntry, src/SAM.h:278-305 (host/tail.cpp: put_line)
template <class Sink>
__device__ inline void put_line(Sink &o, const SamInputs &in, const SamAnnot &A, const SamParams &P, const Row &r, uint32_t qname_read,
                                uint8_t mapq, uint32_t xo, int64_t gene, uint32_t xt) {
  o.bytes(in.ids + in.ids_off[qname_read], in.ids_off[qname_read + 1] - in.ids_off[qname_read]);
  o.ch('\t');
  put_num(o, r.flag);
  o.ch('\t');
  o.bytes(A.locus + A.locus_off[r.rname_entry], A.locus_off[r.rname_entry + 1] - A.locus_off[r.rname_entry]);
  o.ch('\t');
  put_num(o, r.pos);
  o.ch('\t');
  put_num(o, mapq);
  o.ch('\t');
  const kslam_overlap *ov = r.mapped ? &in.ov[r.ov] : nullptr;
  if (!P.report_cigar || !r.mapped) {
    o.ch('*');
  } else if (in.pool && ov->cigar_len) {   // getCigarAndMD's CIGAR text, src/SAM.h:120-125, 185-191 (host/tail.cpp: cigar_and_md)
    if (ov->query_begin > 0) {
      put_num(o, (uint64_t)ov->query_begin);
      o.ch('S');
    }
    for (uint32_t k = 0; k < ov->cigar_len; k++) {
      const uint32_t c = in.pool[ov->cigar_off + k], len = c >> 4, op = c & 15;
      put_num(o, len);
      if (op < 3) o.ch((uint8_t)"MID"[op]);
    }
    const int64_t L = (int64_t)(in.read_off[ov->read + 1] - in.read_off[ov->read]);
    const int64_t tail = L - ov->query_end - 1;
    if (tail > 0) {
      put_num(o, (uint64_t)tail);
      o.ch('S');
    }
  }
  o.ch('\t');
  o.ch(P.paired ? '=' : '*');
  o.ch('\t');
  put_num(o, r.pnext);
  o.ch('\t');
  put_snum(o, r.tlen);
  LIT(o, "\t*\t*");
  if (r.mapped) {
    uint32_t nm = 0;
    if (P.report_cigar) {
      LIT(o, "\tMD:Z:");
      if (walked(in, r.ov)) {
        const kslam_row_detail &d = in.det[r.ov];
        o.bytes(in.md_pool + d.md_off, d.md_len);
        nm = d.nm;
      }
    }
    LIT(o, "\tAS:i:");
    put_num(o, r.as);
    LIT(o, "\tXS:i:");
    put_num(o, r.xs);
    LIT(o, "\tNM:i:");
    put_num(o, nm);
    LIT(o, "\tX0:i:");
    put_num(o, xo);
    if (xt != 0) {
      LIT(o, "\tXT:i:");
      put_num(o, xt);
    }
    if (gene >= 0) {
      if (A.gname_off[gene + 1] > A.gname_off[gene]) {
        LIT(o, "\tXG:Z:");
        o.bytes(A.gname + A.gname_off[gene], A.gname_off[gene + 1] - A.gname_off[gene]);
      }
      if (A.prot_off[gene + 1] > A.prot_off[gene]) {
        LIT(o, "\tXP:Z:");
        o.bytes(A.prot + A.prot_off[gene], A.prot_off[gene + 1] - A.prot_off[gene]);
      }
      if (A.prod_off[gene + 1] > A.prod_off[gene]) {
        LIT(o, "\tXR:Z:\"");
        o.bytes(A.prod + A.prod_off[gene], A.prod_off[gene + 1] - A.prod_off[gene]);
        o.ch('"');
      }
    }
  }
  o.ch('\n');
}

// ---- plan: the reference's per-pair sort, the rows to report, which mates need libm ----------------------------------
__global__ __launch_bounds__(256) void k_sam_plan(Rec *__restrict__ recs, const kslam_read_pair *__restrict__ groups, uint64_t n_groups,
                                                  SamInputs in, SamParams P, SamPlan *__restrict__ plan, uint32_t *__restrict__ n_vals,
                                                  uint32_t *__restrict__ n_segs, uint32_t *__restrict__ err) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const uint32_t cnt = (uint32_t)groups[g].count;
  if (!cnt) {
    plan[g] = SamPlan{0, 0, 0, 0};
    n_vals[g] = 0;
    n_segs[g] = 0;
    return;
  }
  Rec *v = recs + groups[g].first;
  if (P.sort_groups) kslam_gnu::sort(v, v + cnt, ByScoreDesc());   // writeSAMOutputPairs' first statement, src/SAM.h:446-450
  const uint32_t n_rows = min(cnt, max(P.num_alignments, 1u));
  uint32_t use1 = 0, use2 = 0, bad = 0;
  double lone1 = 0, lone2 = 0;
  for (uint32_t k = 0; k < n_rows; k++) {
    const Rec r = v[k];
    if (r.r1 != NONE) {
      use1++;
      lone1 = row_logp(in, r.r1);
      bad |= row_flags(in, r.r1) & 2u;
    }
    if (r.r2 != NONE) {
      use2++;
      lone2 = row_logp(in, r.r2);
      bad |= row_flags(in, r.r2) & 2u;
    }
  }
  // 10^logp matters when it is summed with other rows' or when it could underflow to 0 (host/tail.cpp: cigar_and_md)
  const bool need1 = use1 > 1 || (use1 == 1 && lone1 <= -300.0), need2 = use2 > 1 || (use2 == 1 && lone2 <= -300.0);
  if (need1 || need2)
    for (uint32_t k = 0; k < n_rows; k++) {
      const Rec r = v[k];
      if (need1 && r.r1 != NONE) bad |= row_flags(in, r.r1) & 1u;
      if (need2 && r.r2 != NONE) bad |= row_flags(in, r.r2) & 1u;
    }
  if (bad) atomicOr(err, bad);
  plan[g] = SamPlan{n_rows, use1, use2, (uint32_t)need1 | ((uint32_t)need2 << 1)};
  n_segs[g] = (uint32_t)need1 + (uint32_t)need2;
  n_vals[g] = ((uint32_t)need1 + (uint32_t)need2) * n_rows;
}

// the log-probabilities the host needs, mate by mate: segment = n_rows slots in row order, +infinity where the row lacks the mate
__global__ __launch_bounds__(256) void k_sam_collect(const Rec *__restrict__ recs, const kslam_read_pair *__restrict__ groups,
                                                     uint64_t n_groups, SamInputs in, const SamPlan *__restrict__ plan,
                                                     const uint64_t *__restrict__ val_off, const uint64_t *__restrict__ seg_off,
                                                     double *__restrict__ vals, uint32_t *__restrict__ seg_len) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const SamPlan pl = plan[g];
  const uint32_t n_rows = pl.n_rows;
  const bool need1 = pl.need & 1u, need2 = pl.need & 2u;
  if (!(need1 || need2)) return;
  const Rec *v = recs + groups[g].first;
  uint64_t at = val_off[g], sg = seg_off[g];
  const double INF = __longlong_as_double(0x7FF0000000000000ll);
  if (need1) {
    for (uint32_t k = 0; k < n_rows; k++) vals[at + k] = v[k].r1 != NONE ? row_logp(in, v[k].r1) : INF;
    seg_len[sg++] = n_rows;
    at += n_rows;
  }
  if (need2) {
    for (uint32_t k = 0; k < n_rows; k++) vals[at + k] = v[k].r2 != NONE ? row_logp(in, v[k].r2) : INF;
    seg_len[sg] = n_rows;
  }
}

// the text of one read pair's lines (host/tail.cpp: write_group from the mapping qualities on)
template <class Sink>
__device__ inline void write_group(Sink &o, const Rec *v, const kslam_read_pair &grp, const SamPlan pl, const SamInputs &in, const SamAnnot &A,
                                   const SamParams &P, const uint8_t *mapq_vals, uint64_t val_at) {
  const uint32_t n_rows = pl.n_rows, use1 = pl.use1, use2 = pl.use2;
  const bool need1 = pl.need & 1u, need2 = pl.need & 2u;
  const uint64_t at1 = val_at, at2 = val_at + (need1 ? n_rows : 0);
  for (uint32_t k = 0; k < n_rows; k++) {
    const Rec p = v[k];
    Row a, b;
    make_rows(p, in.ov, P.paired != 0, a, b);
    if (k == 0) {
      a.flag &= ~0x100;
      b.flag &= ~0x100;
    }
    // mapq_of, src/SAM.h:502-506: a mate the row lacks has probability 0 -> quality 0; a lone mapped row -> the constant
    const uint8_t q1 = a.mapped ? (need1 ? mapq_vals[at1 + k] : (uint8_t)P.mapq_unique) : 0;
    const uint8_t q2 = b.mapped ? (need2 ? mapq_vals[at2 + k] : (uint8_t)P.mapq_unique) : 0;
    const int64_t gene = best_gene(A, p.entry, p.ref_start, p.ref_end);
    const uint32_t xt = A.tax[p.entry];
    put_line(o, in, A, P, a, grp.r1_read, q1, use1, gene, xt);
    if (P.paired) put_line(o, in, A, P, b, grp.r2_read, q2, use2, gene, xt);
    if (P.sam_xa) break;
  }
}

__global__ __launch_bounds__(256) void k_sam_lengths(const Rec *__restrict__ recs, const kslam_read_pair *__restrict__ groups,
                                                     uint64_t n_groups, SamInputs in, SamAnnot A, SamParams P,
                                                     const SamPlan *__restrict__ plan, const uint64_t *__restrict__ val_off,
                                                     const uint8_t *__restrict__ mapq_vals, uint32_t *__restrict__ text_len) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const SamPlan pl = plan[g];
  if (!pl.n_rows) {
    text_len[g] = 0;
    return;
  }
  CountSink o;
  write_group(o, recs + groups[g].first, groups[g], pl, in, A, P, mapq_vals, val_off[g]);
  text_len[g] = (uint32_t)o.n;
}
__global__ __launch_bounds__(256) void k_sam_write(const Rec *__restrict__ recs, const kslam_read_pair *__restrict__ groups,
                                                   uint64_t n_groups, SamInputs in, SamAnnot A, SamParams P,
                                                   const SamPlan *__restrict__ plan, const uint64_t *__restrict__ val_off,
                                                   const uint8_t *__restrict__ mapq_vals, const uint64_t *__restrict__ text_off,
                                                   uint8_t *__restrict__ text) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const SamPlan pl = plan[g];
  if (!pl.n_rows) return;
  ByteSink o(text + text_off[g]);
  write_group(o, recs + groups[g].first, groups[g], pl, in, A, P, mapq_vals, val_off[g]);
  o.flush();
}

// ---- per-read taxonomy: getLowestCommonAncestor over the read pair's entries (host/taxonomy.cpp: lca_ids) ----------
__device__ inline uint32_t lca_nodes(const SamAnnot &A, uint32_t a, uint32_t b) {
  while (A.depth[a] > A.depth[b]) a = A.up[a];
  while (A.depth[b] > A.depth[a]) b = A.up[b];
  while (a != b) {
    a = A.up[a];
    b = A.up[b];
    if (a == NONE || b == NONE) return NONE;
  }
  return a;
}
__global__ __launch_bounds__(256) void k_lca(const Rec *__restrict__ recs, const kslam_read_pair *__restrict__ groups, uint64_t n_groups,
                                             SamInputs in, SamAnnot A, uint32_t *__restrict__ tax_ids, uint32_t *__restrict__ line_len) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const uint32_t cnt = (uint32_t)groups[g].count;
  const Rec *v = recs + groups[g].first;
  uint32_t acc = NONE, lone = 0, result = 0;
  bool have_lone = false, zero = cnt == 0;
  for (uint32_t k = 0; k < cnt && !zero; k++) {
    const uint32_t e = v[k].entry, id = A.tax[e];
    if (id == 0) {   // an empty path: nothing in common
      zero = true;
      break;
    }
    const uint32_t nd = A.entry_node[e];
    if (nd == NONE) {   // an id the tree does not know: a path of just itself
      if (acc != NONE || (have_lone && lone != id)) {
        zero = true;
        break;
      }
      lone = id;
      have_lone = true;
      continue;
    }
    if (have_lone) {
      zero = true;
      break;
    }
    acc = acc == NONE ? nd : lca_nodes(A, acc, nd);
    if (acc == NONE) zero = true;
  }
  if (!zero) result = have_lone ? lone : (acc == NONE ? 0 : A.node_tax[acc]);
  tax_ids[g] = result;
  uint32_t len = 0;
  if (cnt) {   // (a result without alignments has no read name: no line)
    const uint32_t r = groups[g].r1_read;
    len = (uint32_t)(in.ids_off[r + 1] - in.ids_off[r]) + 1 + digits_u64(result) + 1;
  }
  line_len[g] = len;
}
__global__ __launch_bounds__(256) void k_per_read_write(const kslam_read_pair *__restrict__ groups, uint64_t n_groups, SamInputs in,
                                                        const uint32_t *__restrict__ tax_ids, const uint64_t *__restrict__ off,
                                                        uint8_t *__restrict__ text) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups || !groups[g].count) return;
  const uint32_t r = groups[g].r1_read;
  ByteSink o(text + off[g]);
  o.bytes(in.ids + in.ids_off[r], in.ids_off[r + 1] - in.ids_off[r]);
  o.ch('\t');
  put_num(o, tax_ids[g]);
  o.ch('\n');
  o.flush();
}

inline unsigned blocks_for(uint64_t n) { return (unsigned)((n + 255) / 256); }
}  // namespace

void sam_plan(kslam_paired_overlap *d_recs, const kslam_read_pair *d_groups, uint64_t n_groups, const SamInputs &in, const SamParams &P,
              SamWork &W, uint64_t *n_vals, uint64_t *n_segs, uint32_t *err_flags, hipStream_t s) {
  *n_vals = *n_segs = 0;
  *err_flags = 0;
  if (!n_groups) return;
  W.plan.ensure(n_groups * sizeof(SamPlan));
  W.cnt_vals.ensure(n_groups * 4);
  W.cnt_segs.ensure(n_groups * 4);
  W.val_off.ensure((n_groups + 1) * 8);
  W.seg_off.ensure((n_groups + 1) * 8);
  W.scan_tmp.ensure(scan_tmp_bytes(n_groups));
  W.totals.ensure(64);
  HIPCHK(hipMemsetAsync(W.totals.p, 0, 64, s));
  uint32_t *d_err = W.totals.as<uint32_t>() + 8;
  hipLaunchKernelGGL(k_sam_plan, dim3(blocks_for(n_groups)), dim3(256), 0, s, d_recs, d_groups, n_groups, in, P, W.plan.as<SamPlan>(),
                     W.cnt_vals.as<uint32_t>(), W.cnt_segs.as<uint32_t>(), d_err);
  HIPCHK(hipGetLastError());
  exclusive_scan_u32_to_u64(W.cnt_vals.as<uint32_t>(), W.val_off.as<uint64_t>(), n_groups, W.totals.as<uint64_t>(), W.scan_tmp.p, s);
  exclusive_scan_u32_to_u64(W.cnt_segs.as<uint32_t>(), W.seg_off.as<uint64_t>(), n_groups, W.totals.as<uint64_t>() + 1, W.scan_tmp.p, s);
  uint64_t t[5];
  read_back(t, W.totals.p, sizeof t, s);
  *n_vals = t[0];
  *n_segs = t[1];
  *err_flags = (uint32_t)t[4];
  W.vals.ensure((t[0] + 1) * 8);
  W.seg_len.ensure((t[1] + 1) * 4);
  W.mapq.ensure(t[0] + 16);
  if (t[0])
    hipLaunchKernelGGL(k_sam_collect, dim3(blocks_for(n_groups)), dim3(256), 0, s, d_recs, d_groups, n_groups, in, W.plan.as<SamPlan>(),
                       W.val_off.as<uint64_t>(), W.seg_off.as<uint64_t>(), W.vals.as<double>(), W.seg_len.as<uint32_t>());
  HIPCHK(hipGetLastError());
}

void sam_format(const kslam_paired_overlap *d_recs, const kslam_read_pair *d_groups, uint64_t n_groups, const SamInputs &in,
                const SamAnnot &A, const SamParams &P, SamWork &W, uint64_t *text_bytes, hipStream_t s) {
  *text_bytes = 0;
  if (!n_groups) return;
  W.text_len.ensure(n_groups * 4);
  W.text_off.ensure((n_groups + 1) * 8);
  hipLaunchKernelGGL(k_sam_lengths, dim3(blocks_for(n_groups)), dim3(256), 0, s, d_recs, d_groups, n_groups, in, A, P, W.plan.as<SamPlan>(),
                     W.val_off.as<uint64_t>(), W.mapq.as<uint8_t>(), W.text_len.as<uint32_t>());
  HIPCHK(hipGetLastError());
  exclusive_scan_u32_to_u64(W.text_len.as<uint32_t>(), W.text_off.as<uint64_t>(), n_groups, W.totals.as<uint64_t>() + 2, W.scan_tmp.p, s);
  uint64_t total = 0;
  read_back(&total, W.totals.as<uint64_t>() + 2, sizeof total, s);
  W.text.ensure(total + 64);
  if (total)
    hipLaunchKernelGGL(k_sam_write, dim3(blocks_for(n_groups)), dim3(256), 0, s, d_recs, d_groups, n_groups, in, A, P, W.plan.as<SamPlan>(),
                       W.val_off.as<uint64_t>(), W.mapq.as<uint8_t>(), W.text_off.as<uint64_t>(), W.text.as<uint8_t>());
  HIPCHK(hipGetLastError());
  *text_bytes = total;
}

void per_read_device(const kslam_paired_overlap *d_recs, const kslam_read_pair *d_groups, uint64_t n_groups, const SamInputs &in,
                     const SamAnnot &A, SamWork &W, uint64_t *text_bytes, hipStream_t s) {
  *text_bytes = 0;
  if (!n_groups) return;
  W.tax_ids.ensure(n_groups * 4);
  W.pr_len.ensure(n_groups * 4);
  W.pr_off.ensure((n_groups + 1) * 8);
  W.scan_tmp.ensure(scan_tmp_bytes(n_groups));
  W.totals.ensure(64);
  hipLaunchKernelGGL(k_lca, dim3(blocks_for(n_groups)), dim3(256), 0, s, d_recs, d_groups, n_groups, in, A, W.tax_ids.as<uint32_t>(),
                     W.pr_len.as<uint32_t>());
  HIPCHK(hipGetLastError());
  exclusive_scan_u32_to_u64(W.pr_len.as<uint32_t>(), W.pr_off.as<uint64_t>(), n_groups, W.totals.as<uint64_t>() + 3, W.scan_tmp.p, s);
  uint64_t total = 0;
  read_back(&total, W.totals.as<uint64_t>() + 3, sizeof total, s);
  W.pr_text.ensure(total + 64);
  if (total)
    hipLaunchKernelGGL(k_per_read_write, dim3(blocks_for(n_groups)), dim3(256), 0, s, d_groups, n_groups, in, W.tax_ids.as<uint32_t>(),
                       W.pr_off.as<uint64_t>(), W.pr_text.as<uint8_t>());
  HIPCHK(hipGetLastError());
  *text_bytes = total;
}

}  // namespace kslam
