// sw.hip -- Smith-Waterman validation of candidate overlaps (scores + ends).
//
// Replaces, per overlap, performSmithWatermanOnRange2 (reference
// src/SmithWaterman.h:184-233) -> Aligner::Align (src/ssw_cpp.cpp:234-283) ->
// ssw_align's forward and reverse passes (src/ssw.c:841-923) with the SSE2
// kernels sw_sse2_byte / sw_sse2_word (src/ssw.c:143-383, 408-592).
//
// Semantics kept:
//   window = entry.bases.substr(max(rel,0), L), reverse-complemented with
//   A<->T, C<->G only when revComp (SmithWaterman.h:204-207);
//   ASCII -> {A0 C1 G2 T3 U0 else 4} (ssw_cpp.cpp:11-23); 5x5 matrix with
//   +match / -mismatch and zeros on row/column 4 (ssw_cpp.cpp:25-49);
//   H = max(0, Hdiag + s, E, F), E' = max(0, E - gE, H - gO), F' likewise;
//   end_ref = first column (scan order) whose maximum strictly exceeds the
//   running maximum, end_read = smallest read index holding that maximum
//   (ssw.c:316-342, 536-557); the reverse pass runs over ref[0..end_ref]
//   right-to-left against reversed read[0..end_read] and stops after the first
//   column whose maximum equals the forward score (ssw.c:330, 545, 906-923).
// The 8-bit pass and its overflow re-run give the same triple as the 16-bit
// pass, so one int32 DP reproduces both.  The reverse pass is not run at all: the
// forward pass carries each alignment's start cell along with its score (see
// sw_origin_pass), which provably selects the same begin as the reference's
// reverse scan (the test tree holds a scalar CPU statement of it that
// equals the striped emulation on 80k random + 60k low-complexity trials).  The striped Lazy-F evaluation order
// is only observable when a gap pair can beat a mismatch or when gapE >= gapO: scoring outside that
// envelope goes, candidate by candidate, through k_sw_striped below, which plays the reference's SSE
// lanes literally (kslam_create accepts whatever the reference's flags accept).
//
// MI355X design: integer ALU work, no MFMA; the kernels are bound by VALU issue cycles
// (tools/valu_peak.hip), so the code is written against the instruction
// count.  Candidates go through exact banded tiers (k_sw_band: 8 or 16 lanes per candidate,
// 2..8 adjacent diagonals per lane, swept by anti-diagonals; a result is accepted only with a
// certificate that the band held every optimal alignment), picked per candidate by k_sw_plan from
// the seed diagonal; what no band can certify runs on the full-matrix kernel (k_sw: 16 lanes per
// candidate, lane t owns R consecutive read rows and computes column s - t at step s).  Everything
// is in registers (H, E, F per diagonal or row; a 6-bit packed score row per read base so a score
// is one v_bfe_i32); neighbours are one DPP row shift away; sequences sit in LDS as 1 byte/base
// codes staged from pre-encoded arrays.  The lane-local best cell is kept by ONE v_max_f64 per
// cell over a (score | inverted position, H) pair, which is the reference's tie rule (highest
// score, first column, smallest row) whatever order the cells are visited in.
#include <type_traits>

#include "common.h"
#include "stage.h"

namespace kslam {

namespace {

constexpr int KB = 18;                    // low bits of a packed DP value: origin key (col << 9 | row)
constexpr int32_t KEYMASK = (1 << KB) - 1;

typedef int32_t v2i32 __attribute__((ext_vector_type(2)));
// a + b / a - b as exactly one all-VGPR instruction: keeps the compiler from re-deriving running
// values as sums of several induction variables or folding an SGPR into a three-operand form
// (both cost more issue cycles than they save; tools/valu_peak.hip)
__device__ inline int32_t add_vv(int32_t a, int32_t b) {
  int32_t r;
  asm("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline int32_t sub_vv(int32_t a, int32_t b) {
  int32_t r;
  asm("v_sub_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

struct PassResult {
  int32_t score, end_col, end_row, beg_col, beg_row;
};

// group-wide selection of the best cell from the lane-local bests: max score, then smallest
// column, then smallest row (ssw.c:316-342); also hands out its origin key
template <int GL = 16>
__device__ inline PassResult reduce_best(int32_t lbV, int32_t lbZ) {
  const int32_t lane = threadIdx.x & 63;
  const int32_t sc = lbV >> KB;
  const int32_t ecol = (lbZ >> 9) - 1, erow = (lbZ & 511) - 1;
  const int32_t G = sc > 0 ? ((sc << KB) | ((511 - ecol) << 9) | (511 - erow)) : 0;
  int32_t Gm = G;
#pragma unroll
  for (int m = 1; m < GL; m <<= 1) Gm = max(Gm, __shfl_xor(Gm, m, GL));
  const uint64_t bal = __ballot(G == Gm && Gm != 0);
  const int32_t gbase = lane & ~(GL - 1);
  const uint32_t grp_bits = (uint32_t)(bal >> gbase) & ((1u << GL) - 1u);
  PassResult res{0, 0, 0, 0, 0};
  const int32_t src = gbase | (grp_bits ? __builtin_ctz(grp_bits) : 0);
  const int32_t wV = __shfl(lbV, src, 64), wZ = __shfl(lbZ, src, 64);
  if (grp_bits) {
    res.score = wV >> KB;
    res.end_col = (wZ >> 9) - 1;
    res.end_row = (wZ & 511) - 1;
    res.beg_col = (wV & KEYMASK) >> 9;
    res.beg_row = wV & 511;
  }
  return res;
}

// One forward SW pass with origin tracking for the 16-lane group this lane belongs to.
// Every DP value is score * 2^18 + origin key, so v_max_i32 is a lexicographic
// (score, start column, start row) maximum: among the optimal alignments ending at the
// best cell the one starting at the largest column, then the largest row survives -- the
// alignment the reference's reverse pass (ssw.c:906-923) reports.  A cell whose score is 0
// holds the key of its diagonal successor (Z), so a fresh alignment inherits its own first
// cell.  E and F are kept unclamped (max(0, E) is what the reference's saturating
// arithmetic holds; negative values never beat Z).  Result valid in every lane of the group.
template <int R>
__device__ inline PassResult sw_origin_pass(const uint8_t *qcodes, int32_t qlen, const uint8_t *wcodes,
                                            int32_t ncols, const SwParams &p) {
  const int32_t lane = threadIdx.x & 63;
  const int32_t t = lane & 15;
  uint32_t tab[R];
  int32_t H[R], E[R], rowkey[R];
  const int32_t gO = in_vgpr(p.gap_open << KB), gE = in_vgpr(p.gap_extend << KB);   // VGPR operands: 2-cycle subtractions
  const int32_t NEG = -((p.gap_open + p.gap_extend + 1) << KB);
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int32_t i = t * R + r;
    uint32_t tb = 0;
    if (i < qlen) {
      const uint32_t q = qcodes[i];
#pragma unroll
      for (uint32_t c = 0; c < 4; c++) {
        const int32_t s = q > 3u ? 0 : (q == c ? p.match : -p.mismatch);
        tb |= ((uint32_t)s & 63u) << (6 * c);
      }                                   // column code 4 (N) scores 0: bits 24..29 stay clear
    } else {
      tb = 0x20820820u;                   // padding row: -32 against every column code
    }
    tab[r] = tb;
    H[r] = i + 1;                         // virtual cell (i, -1): score 0, successor (i + 1, 0)
    E[r] = 512 | (i + 1);                 // no gap yet: the floor Z of (i, 0) rides on E (see the cell)
    rowkey[r] = i + 1;
  }
  int32_t prev_hl = t * R;                // virtual cell (tR - 1, -1): successor (tR, 0)
  int32_t out_h = 0, out_f = 0;
  // lane-local best cell as a 64-bit (G, H) pair kept by one v_max_f64 per cell, G = (H | KEYMASK) - Z =
  // score over the inverted position key (see k_sw_band); here G >= 0 always (no offset on the score)
  double best = 0.0;
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0");   // MODE.IEEE = 0: NaN patterns are passed over
  const int32_t nsteps = ncols > 0 ? ncols + 15 : 0;
  for (int32_t step = 0;; step++) {
    if (__ballot(step < nsteps) == 0ull) break;   // the four groups of the wave iterate together
    const int32_t in_h = dpp_row_shr1(out_h);
    const int32_t in_f = dpp_row_shr1(out_f);
    const int32_t c = step - t;
    if (step < nsteps && c >= 0 && c < ncols) {
      const uint32_t shift = (uint32_t)wcodes[c] * 6u;
      int32_t diag = t == 0 ? (c << 9) : prev_hl;   // row -1: successor (0, c)
      prev_hl = in_h;
      int32_t F = t == 0 ? NEG : in_f;
      const int32_t colkey2 = (c + 2) << 9;
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int32_t s = __builtin_amdgcn_sbfe(tab[r], shift, 6);
        // As in k_sw_band the zero floor rides on E: a cell hands max(E, Z of its right-hand neighbour)
        // on, so H is one max3 (a floor carried further loses gE per column and is dominated).
        const int32_t Zr = colkey2 | rowkey[r];           // Z of (i, c + 1) = Z of this cell + 512
        const int32_t h = max(max(diag + (s << KB), E[r]), F);
        diag = H[r];
        H[r] = h;
        const int32_t hg = h - gO;
        E[r] = max(max(E[r] - gE, hg), Zr);
        F = max(F - gE, hg);
        const v2i32 gh = {h, sub_vv(h | KEYMASK, Zr)};   // highest score, then first column, then smallest row
        const double cand = __builtin_bit_cast(double, gh);
        asm("v_max_f64 %0, %1, %2" : "=v"(best) : "v"(best), "v"(cand));
      }
      out_h = H[R - 1];
      out_f = F;
    }
  }
  const v2i32 bb = __builtin_bit_cast(v2i32, best);
  // G = score * 2^18 + KEYMASK - 512 - Z(cell): back to the score and the cell's position key
  const int32_t gv = bb.y - (KEYMASK - 512);
  const int32_t gsc = (bb.x | bb.y) == 0 ? 0 : (gv + KEYMASK) >> KB;
  const int32_t lbV = gsc > 0 ? bb.x : 0;                       // score and origin key of the best cell
  const int32_t lbZ = gsc > 0 ? (gsc << KB) - gv : 0;          // its position key
  return reduce_best(lbV, lbZ);
}

// ---- shared pieces of the two SW kernels ------------------------------------------------------
// stage read + window of candidate `o` as SSW codes (all 16 lanes of the group cooperate)
template <int GL = 16>
__device__ inline void stage_candidate(const kslam_overlap &o, const SwInputs &in, int32_t t, uint8_t *sq,
                                       uint8_t *sw, int32_t *L_out, int32_t *wlen_out) {
  const uint64_t ro = in.read_off[o.read];
  const int32_t L = (int32_t)(in.read_off[o.read + 1] - ro);
  const uint64_t go = in.genome_off[o.entry];
  const uint64_t G = in.genome_off[o.entry + 1] - go;
  const int64_t s0 = o.rel > 0 ? o.rel : 0;                           // SmithWaterman.h:204
  const int32_t wlen = (int32_t)min((uint64_t)L, G - (uint64_t)s0);    // substr, :205-206
  for (int32_t i = t; i < L; i += GL) sq[i] = (uint8_t)ssw_code(in.read_bases[ro + i]);
  for (int32_t j = t; j < wlen; j += GL) {
    sw[j] = (uint8_t)(o.revcomp ? ssw_code_complemented(in.genome_bases[go + s0 + (wlen - 1 - j)])  // :207
                                : ssw_code(in.genome_bases[go + s0 + j]));
  }
  *L_out = L;
  *wlen_out = wlen;
}

struct Staged {
  int32_t L, W;        // read length, window length
  int32_t qoff, woff;  // where read base 0 / window base 0 sit in the staged buffers
};
// read (codes x 1, + score rows when tab != nullptr) and window (codes x WS) of candidate o
template <int GL, int WS>
__device__ inline Staged stage_candidate_wide(const kslam_overlap &o, const SwInputs &in, int32_t t, uint8_t *sq,
                                              uint8_t *sw, uint32_t *tab, const SwParams &p, int32_t bias = 0) {
  const uint64_t ro = in.read_off[o.read];
  const int32_t L = (int32_t)(in.read_off[o.read + 1] - ro);
  const uint64_t go = in.genome_off[o.entry];
  const uint64_t G = in.genome_off[o.entry + 1] - go;
  const int64_t s0 = o.rel > 0 ? o.rel : 0;                           // SmithWaterman.h:204
  const int32_t wlen = (int32_t)min((uint64_t)L, G - (uint64_t)s0);    // substr, :205-206
  Staged st;
  st.L = L;
  st.W = wlen;
  st.qoff = stage_span<GL, 1>(in.read_codes + ro, L, false, t, sq, tab, p, bias);
  st.woff = stage_span<GL, WS>(in.genome_codes + go + s0, wlen, o.revcomp != 0, t, sw, nullptr, p);   // :207
  return st;
}

// result record + band request for banded_sw, ssw.c:924-935 (flag 0x0f: score and distance filters)
template <int GL = 16, int WS = 1>   // WS: the window codes in sw[] are stored multiplied by WS
__device__ inline void sw_epilogue(kslam_overlap *ov, uint64_t ci, bool have, int32_t t, int32_t L,
                                   const PassResult &f, const uint8_t *sq, const uint8_t *sw, const SwParams &p,
                                   uint32_t *band0) {
  const bool ok = have && f.score > 0;
  const int32_t refLen = f.end_col - f.beg_col + 1, readLen = f.end_row - f.beg_row + 1;
  const bool want = p.report_cigar && ok && (uint32_t)f.score >= (p.score_threshold & 0xFFFFu) &&
                    refLen - 1 <= 32767 && readLen - 1 <= 32767;
  // Ungapped shortcut: when both spans are equal and the plain diagonal already scores
  // `score`, banded_sw's first attempt (band 1) reaches it on the main diagonal, every H
  // direction there is "diagonal" (ties prefer it, ssw.c:686) and the cigar is <n>M.
  int32_t dsum = 0;
  if (want && refLen == readLen) {
    for (int32_t k = t; k < readLen; k += GL) {
      const uint32_t q = sq[f.beg_row + k], c = sw[f.beg_col + k];
      dsum += (q > 3u || c > 3u * WS) ? 0 : (q * WS == c ? p.match : -p.mismatch);
    }
  }
#pragma unroll
  for (int m = 1; m < GL; m <<= 1) dsum += __shfl_xor(dsum, m, GL);
  if (have && t == 0) {
    kslam_overlap o = ov[ci];
    o.score = (uint16_t)f.score;
    o.ref_begin = ok ? f.beg_col : -1;   // window-relative, unflipped; finalised after the cigar stage
    o.ref_end = ok ? f.end_col : 0;
    o.query_begin = ok ? f.beg_row : -1;
    o.query_end = ok ? f.end_row : L - 1;
    o.cigar_len = 0;
    o.cigar_off = 0;
    uint32_t bw = 0;
    if (want) {
      if (refLen == readLen && dsum == f.score && !p.striped) {   // (an envelope argument: a gap pair may beat a mismatch outside it)
        bw = 0x80000000u | (uint32_t)readLen;   // inline <n>M, no banded DP needed
        o.cigar_len = 1;
      } else {
        bw = (uint32_t)(refLen > readLen ? refLen - readLen : readLen - refLen) + 1u;
      }
    }
    ov[ci] = o;
    band0[ci] = bw;
  }
}

// ---- full-matrix kernel (any candidate) --------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void k_sw(kslam_overlap *__restrict__ ov, uint64_t n, SwInputs in, SwParams p,
                                            uint32_t *__restrict__ band0, const uint32_t *__restrict__ list) {
  constexpr int LMAX = R * 16;
  __shared__ uint8_t s_q[16][LMAX];
  __shared__ uint8_t s_w[16][LMAX];
  const int32_t lane = threadIdx.x & 63;
  const int32_t t = lane & 15;
  const int32_t grp = threadIdx.x >> 4;
  const uint64_t gi = (uint64_t)blockIdx.x * 16 + grp;
  const bool have = gi < n;
  const uint64_t ci = have ? (list ? list[gi] : gi) : 0;
  int32_t L = 0, wlen = 0;
  if (have) stage_candidate(ov[ci], in, t, s_q[grp], s_w[grp], &L, &wlen);
  __syncthreads();
  // forward pass (ssw.c:870-877) and, by origin tracking, the result of the reverse pass (:906-923)
  const PassResult f = sw_origin_pass<R>(s_q[grp], L, s_w[grp], have ? wlen : 0, p);
  sw_epilogue(ov, ci, have, t, L, f, s_q[grp], s_w[grp], p, band0);
}

// ---- exactness certificate of a band -----------------------------------------------------------
// `score` is the score of some real alignment of the candidate (a lower bound of the optimum).  Any
// alignment scoring S >= score with g gap bases pays at least cost(g) = gO + (g-1) gE (gE < gO), so
// it has m >= m0(g) = ceil((score + cost(g)) / match) matches, uses >= m0(g) rows and columns,
// starts on a diagonal d = j - i in [-(L - m0(g)), W - m0(g)] and stays within g of it.  True when
// the union of those ranges over all feasible g lies inside [dlo, dlo + ND - 1].
// the part of the certificate that does not depend on the band: the smallest A(g) = m0(g) - g over the feasible
// g; INT32_MAX when nothing needs bounding, -1 when the score certifies nothing
// a / b for the small dividends of the certificate (scores, lengths x match) without the ~40 instructions of an integer
// division: (a * (2^20 / b + 1)) >> 20 is exact for a <= 4282 and every b in 1..255 (checked exhaustively,
// tests/test_tail.py::test_small_divider); larger dividends take the real division
struct SmallDiv {
  uint32_t m;
  int32_t b;
  __device__ explicit SmallDiv(int32_t bb) : m((1u << 20) / (uint32_t)bb + 1u), b(bb) {}
  __device__ int32_t operator()(int32_t a) const {
    return ((uint32_t)a < 4096u && (uint32_t)b < 256u) ? (int32_t)(((uint32_t)a * m) >> 20) : a / b;
  }
};
// certificate_amin below with the two dividers made once per wave
__device__ inline int32_t certificate_amin_fast(int32_t score, int32_t L, int32_t W, const SwParams &p, const SmallDiv &by_match,
                                                const SmallDiv &by_gap_extend) {
  if (score <= 0) return -1;
  const int32_t Lm = min(L, W), ma = p.match;
  const int32_t m00 = by_match(score + ma - 1);
  if (m00 > Lm) return INT32_MAX;
  int32_t amin = m00;
  const int32_t room = Lm * ma - score - p.gap_open;
  if (room >= 0) {
    int32_t g = 1;
    if (p.gap_extend < ma) g = min(by_gap_extend(room) + 1, 2047);
    const int32_t m0 = by_match(score + p.gap_open + (g - 1) * p.gap_extend + ma - 1);
    amin = min(amin, m0 - g);
  }
  return amin;
}
__device__ inline int32_t certificate_amin(int32_t score, int32_t L, int32_t W, const SwParams &p) {
  if (score <= 0) return -1;
  const int32_t Lm = min(L, W), ma = p.match;
  const int32_t m00 = (score + ma - 1) / ma;
  if (m00 > Lm) return INT32_MAX;
  int32_t amin = m00;
  const int32_t room = Lm * ma - score - p.gap_open;
  if (room >= 0) {
    int32_t g = 1;
    if (p.gap_extend < ma) g = min(room / p.gap_extend + 1, 2047);
    const int32_t m0 = (score + p.gap_open + (g - 1) * p.gap_extend + ma - 1) / ma;
    amin = min(amin, m0 - g);
  }
  return amin;
}
__device__ inline bool band_holds(int32_t amin, int32_t L, int32_t W, int32_t dlo, int32_t ND) {
  if (amin < 0) return false;
  if (amin == INT32_MAX) return true;
  return amin - L >= dlo && W - amin <= dlo + ND - 1;
}

__device__ inline bool band_certifies(int32_t score, int32_t L, int32_t W, int32_t dlo, int32_t ND,
                                      const SwParams &p) {
  if (score <= 0) return false;
  // With A(g) = m0(g) - g the ranges are [-(L - A(g)), W - A(g)]: all that matters is the smallest
  // A(g) over the feasible g (m0(g) <= min(L, W)).  For g >= 1, A(g) = ceil((c + g gE) / match) - g
  // never decreases with g when gE >= match and never increases when gE < match, so the minimum
  // sits at g = 0, g = 1 or the largest feasible g: no loop.
  const int32_t Lm = min(L, W), ma = p.match;
  const int32_t m00 = (score + ma - 1) / ma;
  if (m00 > Lm) return true;   // (cannot happen for a score some alignment reached; nothing to bound)
  int32_t amin = m00;
  const int32_t room = Lm * ma - score - p.gap_open;   // >= 0 iff one gap base is affordable
  if (room >= 0) {
    int32_t g = 1;
    if (p.gap_extend < ma) g = min(room / p.gap_extend + 1, 2047);
    const int32_t m0 = (score + p.gap_open + (g - 1) * p.gap_extend + ma - 1) / ma;
    amin = min(amin, m0 - g);
  }
  return amin - L >= dlo && W - amin <= dlo + ND - 1;
}

// The banded tiers of one launch configuration: diagonals swept per tier, the list each tier works
// through and the counters behind the lists (counts[k]: entries of list k; counts[NT_FULL]: entries of
// the full-matrix list).
constexpr int NT_MAX = 6, NT_FULL = 7;
struct Tiers {
  int n;                  // tiers in use
  int unknown;            // tier a candidate starts in when its seed diagonals certify nothing
  int nd[NT_MAX];         // diagonals of tier k, ascending
  uint32_t *list[NT_MAX];
  uint32_t *full_list;
  uint32_t *counts;
};

// ---- tier planning ------------------------------------------------------------------------------
// Which band does a candidate need?  The certificate only asks for a lower bound of the optimal
// score, and the cheapest real alignment to score is the seed itself: the read laid on one of the
// diagonals the join merged into this candidate (|rel difference| < 3), no gaps, full length.  The
// best of those five plain diagonal sums picks the narrowest band that is certain to certify;
// gapped alignments, whose diagonal sums are poor, start in the 32-diagonal band and move up on
// failure as before.
// Software-pipelined: the way to a candidate's bases is three dependent loads long (its record -> the offsets of its read
// and its entry -> the 16-byte chunks of the two spans) and the arithmetic behind them is short, so a workgroup that
// took 32 candidates through "load, barrier, count, exit" spent its life waiting (1.25 ms for 8 M candidates, VALU 0.43).
// Here a WAVE walks through octets of candidates (eight lanes each; nothing is shared between waves, so there is no
// block barrier) and every level of the chain is issued one octet ahead of the level that consumes it: while octet i is
// counted, the chunks of octet i + 1, the offsets of i + 2 and the record of i + 3 are on their way.
template <int CPL>
struct SpanFetch {   // stage_span split in two: the loads ...
  uint4 v[CPL];
  int32_t nch, shift, len;
  bool rc;
};
template <int GL, int CPL>
__device__ inline void span_fetch(const uint8_t *src, int32_t len, bool rc, int32_t t, SpanFetch<CPL> &S) {
  S.shift = (int32_t)(reinterpret_cast<uintptr_t>(src) & 15u);
  S.len = len;
  S.rc = rc;
  S.nch = (S.shift + len + 15) >> 4;
  const uint4 *base = reinterpret_cast<const uint4 *>(src - S.shift);
#pragma unroll
  for (int j = 0; j < CPL; j++) {
    const int32_t k = t + GL * j;
    S.v[j] = k < S.nch ? base[k] : make_uint4(0u, 0u, 0u, 0u);
  }
}
template <int GL, int WS, int CPL>   // ... and the conversion + LDS stores; returns where element 0 sits in dst
__device__ inline int32_t span_store(const SpanFetch<CPL> &S, int32_t t, uint8_t *dst) {
#pragma unroll
  for (int j = 0; j < CPL; j++) {
    const int32_t k = t + GL * j;
    if (k < S.nch) {
      const uint4 v = S.v[j];
      uint4 c;
      int32_t at;
      if (!S.rc) {
        c.x = codes_of_dword<WS>(v.x, false);
        c.y = codes_of_dword<WS>(v.y, false);
        c.z = codes_of_dword<WS>(v.z, false);
        c.w = codes_of_dword<WS>(v.w, false);
        at = k;
      } else {
        c.x = __builtin_bswap32(codes_of_dword<WS>(v.w, true));
        c.y = __builtin_bswap32(codes_of_dword<WS>(v.z, true));
        c.z = __builtin_bswap32(codes_of_dword<WS>(v.y, true));
        c.w = __builtin_bswap32(codes_of_dword<WS>(v.x, true));
        at = S.nch - 1 - k;
      }
      reinterpret_cast<uint4 *>(dst)[at] = c;
    }
  }
  return S.rc ? 16 * S.nch - S.shift - S.len : S.shift;
}
__device__ inline void wave_lds_fence() {   // LDS traffic of ONE wave is in order; this only pins the compiler
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int LMAX>
__global__ __launch_bounds__(256) void k_sw_plan(kslam_overlap *__restrict__ ov, uint64_t n, SwInputs in,
                                                 SwParams p, Tiers T, uint8_t *__restrict__ tier,
                                                 uint32_t *__restrict__ band0) {
  constexpr int GL = 8, NG = 256 / GL, PW = 16;   // PW: bytes of "N" padding either side of a span
  constexpr int CPL = (((LMAX + 30) >> 4) + GL - 1) / GL;   // chunks of a span per lane
  __shared__ __attribute__((aligned(16))) uint8_t s_q[NG][LMAX + STAGE_PAD + 2 * PW];
  __shared__ __attribute__((aligned(16))) uint8_t s_w[NG][LMAX + STAGE_PAD + 2 * PW];
  const int32_t lane = threadIdx.x & 63;
  const int32_t t = lane & (GL - 1);
  const int32_t grp = threadIdx.x / GL;
  const uint64_t octets = (n + 7) / 8;
  const uint64_t nw = (uint64_t)gridDim.x * 4, w0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t gsub = (uint32_t)lane >> 3;
  struct C1 { bool have; uint32_t read, entry; int32_t rel; uint32_t rc; };
  struct C2 { bool have; int32_t rel; uint32_t rc; uint64_t ro, ro1, go, go1; };
  struct C3 { bool have; int32_t rel, L, W; SpanFetch<CPL> q, w; };
  auto s1 = [&](uint64_t oct) {   // the candidate's record
    C1 c{false, 0u, 0u, 0, 0u};
    const uint64_t gi = oct * 8 + gsub;
    if (oct < octets && gi < n) {
      const kslam_overlap *o = ov + gi;
      c.have = true;
      c.read = o->read;
      c.entry = o->entry;
      c.rel = o->rel;
      c.rc = o->revcomp;
    }
    return c;
  };
  auto s2 = [&](const C1 &a) {    // where its read and its entry lie
    C2 c{a.have, a.rel, a.rc, 0ull, 0ull, 0ull, 0ull};
    if (a.have) {
      c.ro = in.read_off[a.read];
      c.ro1 = in.read_off[a.read + 1];
      c.go = in.genome_off[a.entry];
      c.go1 = in.genome_off[a.entry + 1];
    }
    return c;
  };
  auto s3 = [&](const C2 &a) {    // the chunks of the two spans (stage_candidate_wide's arithmetic)
    C3 c;
    c.have = a.have;
    c.rel = a.rel;
    c.L = (int32_t)(a.ro1 - a.ro);
    const uint64_t G = a.go1 - a.go;
    const int64_t s0 = a.rel > 0 ? a.rel : 0;                                 // SmithWaterman.h:204
    c.W = a.have ? (int32_t)min((uint64_t)c.L, G - (uint64_t)s0) : 0;        // substr, :205-206
    if (a.have) {
      span_fetch<GL, CPL>(in.read_codes + a.ro, c.L, false, t, c.q);
      span_fetch<GL, CPL>(in.genome_codes + a.go + s0, c.W, a.rc != 0, t, c.w);   // :207
    } else {
      c.q.nch = c.w.nch = 0;
      c.q.shift = c.w.shift = c.q.len = c.w.len = 0;
      c.q.rc = c.w.rc = false;
    }
    return c;
  };
  const SmallDiv by_match(p.match), by_gap_extend(p.gap_extend);
  // the tier a lane speaks for: lane t of a group tests tier t's band, the first that holds wins
  int32_t nd_mine = 0;
#pragma unroll
  for (int k = 0; k < NT_MAX; k++) nd_mine = (t == k && k < T.n) ? T.nd[k] : nd_mine;
  C3 z0 = s3(s2(s1(w0)));
  C2 y1 = s2(s1(w0 + nw));
  C1 x2 = s1(w0 + 2 * nw);
  for (uint64_t oct = w0; oct < octets; oct += nw) {
  const C3 z1 = s3(y1);
  const C2 y2 = s2(x2);
  const C1 x3 = s1(oct + 3 * nw);
  const uint64_t gi = oct * 8 + gsub;
  const bool have = z0.have;
  const int32_t L = z0.L, W = z0.W, rel = z0.rel;
  uint8_t *qc = s_q[grp] + PW, *wc = s_w[grp] + PW;
  wave_lds_fence();   // the previous octet's reads of these buffers are done
  if (have) {
    qc += span_store<GL, 1, CPL>(z0.q, t, qc);
    wc += span_store<GL, 1, CPL>(z0.w, t, wc);
  }
  wave_lds_fence();
  for (int32_t x = t; x < PW; x += GL) {   // code 4 scores 0 against everything
    qc[-1 - x] = 4;
    qc[L + x] = 4;
    wc[-1 - x] = 4;
    wc[W + x] = 4;
  }
  wave_lds_fence();
  // Four bases per step: read word (LDS-aligned) against three window words, matches and mismatches counted with
  // byte-parallel arithmetic (codes are 0..4, so x + 0x7F sets bit 7 of a byte iff it is non-zero; code 4 = bit 2 =
  // "scores 0").  WHICH three diagonals: the dedupe keeps the smallest rel of a chain and drops what lies up to 2 above
  // it (Overlap.h:79-85 on the ascending order of :87-98), so the seeds merged into this candidate sit on rel, rel + 1,
  // rel + 2 -- window diagonals d0 .. d0 + 2, or d0 - 2 .. d0 in the flipped window of a revComp candidate.  (Five
  // diagonals, d0 - 2 .. d0 + 2 for everybody, planned the same tiers: the other two never held the best sum.  Any
  // subset is SOUND -- the sums are only a lower bound of the optimum that picks the starting tier.)
  const int32_t d0 = rel < 0 ? rel : 0;
  const int32_t qa = (int32_t)(reinterpret_cast<uintptr_t>(qc) & 3u);
  // read indices i = 4 m - qa; keep every window byte touched inside its padding
  const int32_t i_lo = max(-qa, ((-d0 - 8) & ~3) - qa), i_hi = min(L, W - d0 + 8);
  uint32_t nm[3] = {0, 0, 0}, nx[3] = {0, 0, 0};
  constexpr uint32_t B7 = 0x80808080u, LO7 = 0x7F7F7F7Fu;
  const int32_t dfirst = d0 - (z0.w.rc ? 2 : 0);   // the lowest of the three diagonals
  for (int32_t i = i_lo + 4 * t; i < i_hi; i += 4 * GL) {
    const uint32_t q = *reinterpret_cast<const uint32_t *>(qc + i);
    const uint32_t qn = (q << 5) & B7;   // bit 7 where the read base is N / padding
    const uint8_t *wa = wc + (i + dfirst);
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(wa) & 3u);
    const uint32_t *wb = reinterpret_cast<const uint32_t *>(wa - sh);
    // the eight window bytes from wa on, byte-aligned (v_alignbyte_b32); diagonal k then starts at byte k
    const uint32_t a0 = __builtin_amdgcn_alignbyte(wb[1], wb[0], sh), a1 = __builtin_amdgcn_alignbyte(wb[2], wb[1], sh);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const uint32_t w = k == 0 ? a0 : __builtin_amdgcn_alignbyte(a1, a0, (uint32_t)k);
      const uint32_t x = q ^ w;
      const uint32_t ne = (x + LO7) & B7;            // bytes that differ
      const uint32_t inv = qn | ((w << 5) & B7);     // bytes that score 0
      nx[k] += (uint32_t)__popc(ne & (inv ^ B7));    // mismatches
      nm[k] += (uint32_t)__popc((ne | inv) ^ B7);    // matches
    }
  }
  int32_t best = 0, full = 0, full_x = 0, near_m = 0;   // near_m: most matches on one of the two other diagonals counted
#pragma unroll
  for (int k = 0; k < 3; k++) {
    uint32_t c = nm[k] | (nx[k] << 16);   // both counts of a diagonal in one register (each < 2^10)
#pragma unroll
    for (int m = 1; m < GL; m <<= 1) c += (uint32_t)__shfl_xor((int)c, m, GL);
    best = max(best, (int32_t)(c & 0xFFFFu) * p.match - (int32_t)(c >> 16) * p.mismatch);
    const bool seed = z0.w.rc ? k == 2 : k == 0;   // the seed diagonal d0 itself
    if (seed) {
      full = (int32_t)(c & 0xFFFFu);
      full_x = (int32_t)(c >> 16);
    } else {
      near_m = max(near_m, (int32_t)(c & 0xFFFFu));
    }
  }
  // A read that matches its whole window base for base (seed diagonal, W = L, every column a real
  // match: no N) needs no DP at all: score match x L; any other alignment has fewer matched pairs or
  // pays for a gap, so it is the unique optimum -- end (L-1, L-1), begin (0, 0), CIGAR <L>M -- and
  // the reference's tie rules never come into play.  ~5 % of the candidates of the bench workload.
  bool perfect = have && rel >= 0 && W == L && L > 0 && full == L && p.ablate == 0;
  PassResult f{p.match * L, L - 1, L - 1, 0, 0};
  // ONE mismatch on the seed diagonal, every other column a real match (8.8 % of the bench workload's candidates): no DP
  // either, when four counts say that nothing else can reach the diagonal's own best score S1.  With x the mismatch's row,
  // the diagonal offers three maximal runs -- the whole read: match (L - 1) - mismatch; rows [0, x): match x; rows (x, L):
  // match (L - 1 - x) -- and S1 is the largest, required to be STRICTLY the largest (a tie is left to the DP and the
  // reference's tie rules).  Everything else is below S1 when
  //   (a) match (L - 3) < S1               an ungapped alignment on a diagonal 3 or more away has at most L - 3 pairs;
  //   (b) match (L - 1) - gapO < S1        an alignment with g >= 1 gap bases has I inserted and D deleted bases, at most
  //                                        L - I pairs by its rows and at most W - D = L - D by its columns, so at most
  //                                        L - 1, and pays gapO at least: the window is exactly as long as the read
  //                                        (substr(s, L), src/SmithWaterman.h:204-206), that is what kills the gapped ones;
  //   (c) match M_k < S1 for k = +-1, +-2  an ungapped alignment on diagonal d0 + k scores at most match x (the matches
  //                                        M_k on that diagonal): two of the four were counted above, the other two are
  //                                        counted here, by the lanes of the few candidates that get this far.
  // Then the run is the unique optimum: its end cell is the only cell holding S1 (ssw.c:316-342 has nothing to choose),
  // the reverse pass finds its start (:906-923), the spans are equal and the diagonal sums to S1, so the CIGAR is <n>M
  // (sw_epilogue).  tests: test_one_mismatch_closed_form_equals_the_dp (every mismatch row, tandem repeats that fail (c),
  // scorings that fail (a) / (b) or tie), and every parity test of the suite, whose batches are full of such candidates.
  const bool one = have && !perfect && rel >= 0 && W == L && L > 4 && full == L - 1 && full_x == 1 && p.ablate == 0;
  if (__ballot(one)) {
    int32_t far_m[2] = {0, 0}, xrow = -1;
    if (one) {
      const int32_t dsec = z0.w.rc ? 1 : -2;   // the two diagonals not counted above: d0 - 2, d0 - 1, or d0 + 1, d0 + 2 when flipped
      for (int32_t i = -qa + 4 * t; i < L; i += 4 * GL) {
        const uint32_t q = *reinterpret_cast<const uint32_t *>(qc + i);
        const uint32_t qn = (q << 5) & B7;
        const uint8_t *wa = wc + (i + dsec);
        const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(wa) & 3u);
        const uint32_t *wb = reinterpret_cast<const uint32_t *>(wa - sh);
        const uint32_t a0 = __builtin_amdgcn_alignbyte(wb[1], wb[0], sh), a1 = __builtin_amdgcn_alignbyte(wb[2], wb[1], sh);
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const uint32_t w = k == 0 ? a0 : __builtin_amdgcn_alignbyte(a1, a0, 1u);
          const uint32_t ne = ((q ^ w) + LO7) & B7;
          far_m[k] += (uint32_t)__popc((ne | qn | ((w << 5) & B7)) ^ B7);
        }
        // the seed diagonal once more, for the row of its one mismatch
        const uint8_t *w0p = wc + i;
        const uint32_t sh0 = (uint32_t)(reinterpret_cast<uintptr_t>(w0p) & 3u);
        const uint32_t *w0b = reinterpret_cast<const uint32_t *>(w0p - sh0);
        const uint32_t w0 = __builtin_amdgcn_alignbyte(w0b[1], w0b[0], sh0);
        const uint32_t mm = ((q ^ w0) + LO7) & B7 & ~(qn | ((w0 << 5) & B7));
        if (mm) xrow = i + ((int32_t)__builtin_ctz(mm) >> 3);
      }
    }
#pragma unroll
    for (int m = 1; m < GL; m <<= 1) {
      far_m[0] += __shfl_xor(far_m[0], m, GL);
      far_m[1] += __shfl_xor(far_m[1], m, GL);
      xrow = max(xrow, __shfl_xor(xrow, m, GL));
    }
    if (one && xrow >= 0 && xrow < L) {
      const int32_t ma = p.match, whole = ma * (L - 1) - p.mismatch, left = ma * xrow, right = ma * (L - 1 - xrow);
      int32_t S1, b, e;
      if (whole > left && whole > right) { S1 = whole; b = 0; e = L - 1; }
      else if (left > whole && left > right) { S1 = left; b = 0; e = xrow - 1; }
      else if (right > whole && right > left) { S1 = right; b = xrow + 1; e = L - 1; }
      else { S1 = -1; b = e = 0; }
      const int32_t others = max(max(ma * (L - 3), ma * (L - 1) - p.gap_open), ma * max(near_m, max(far_m[0], far_m[1])));
      if (S1 > 0 && others < S1) {
        perfect = true;   // (for the tier list: this candidate is in none)
        f = PassResult{S1, e, e, b, b};
      }
    }
  }
  sw_epilogue<GL, 1>(ov, gi, perfect, t, L, f, qc, wc, p, band0);
  {
    // every lane of the group has `best`; lane k asks whether tier k's band holds, the narrowest that does is the choice
    // (no diagonal certifies anything -- a gapped alignment: T.unknown, see sw_scores)
    const int32_t amin = certificate_amin_fast(best, L, W, p, by_match, by_gap_extend);
    const bool holds = nd_mine > 0 && band_holds(amin, L, W, d0 - nd_mine / 2, nd_mine);
    const uint32_t mine = (uint32_t)(__ballot(holds) >> (8u * gsub)) & 0xFFu;
    const int choice = mine ? (int)__builtin_ctz(mine) : T.unknown;
    if (have && t == 0) tier[gi] = perfect ? (uint8_t)255 : (uint8_t)choice;   // 255: in no tier's list
  }
  z0 = z1;
  y1 = y2;
  x2 = x3;
  }
}

// NT-way stable partition of the candidate numbers by tier: per-block counts, one small scan,
// then a scatter that ranks within the block by ballots.
constexpr int TIER_ITEMS = 4096;   // candidates per block
constexpr int NT = 8;              // tier bins (up to 6 used)
__global__ __launch_bounds__(256) void k_tier_hist(const uint8_t *__restrict__ tier, uint64_t n,
                                                   uint32_t *__restrict__ block_hist, uint32_t n_blocks) {
  __shared__ uint32_t h[NT];
  if (threadIdx.x < NT) h[threadIdx.x] = 0;
  __syncthreads();
  uint32_t c[NT] = {0, 0, 0, 0, 0, 0, 0, 0};
  const uint64_t base = (uint64_t)blockIdx.x * TIER_ITEMS;
  for (uint32_t k = threadIdx.x; k < TIER_ITEMS; k += 256) {
    const uint64_t i = base + k;
    if (i < n && tier[i] < NT) c[tier[i]]++;
  }
#pragma unroll
  for (int k = 0; k < NT; k++) {
    uint32_t v = c[k];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&h[k], v);
  }
  __syncthreads();
  if (threadIdx.x < NT) block_hist[threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of block_hist per tier (each tier's list starts at 0); totals[k] = tier size
__global__ __launch_bounds__(1024) void k_tier_scan(uint32_t *__restrict__ block_hist, uint32_t n_blocks,
                                                    uint32_t *__restrict__ totals) {
  __shared__ uint32_t part[1024];
  const uint32_t k = blockIdx.x;   // tier
  uint32_t *a = block_hist + (size_t)k * n_blocks;
  const uint32_t per = (n_blocks + 1023) / 1024;
  const uint32_t lo = min(n_blocks, threadIdx.x * per), hi = min(n_blocks, lo + per);
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; i++) sum += a[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (uint32_t i = 0; i < 1024; i++) {
      const uint32_t v = part[i];
      part[i] = run;
      run += v;
    }
    totals[k] = run;
  }
  __syncthreads();
  uint32_t run = part[threadIdx.x];
  for (uint32_t i = lo; i < hi; i++) {
    const uint32_t v = a[i];
    a[i] = run;
    run += v;
  }
}

struct TierLists {
  uint32_t *list[NT];
};
__global__ __launch_bounds__(256) void k_tier_scatter(const uint8_t *__restrict__ tier, uint64_t n,
                                                      const uint32_t *__restrict__ block_hist, uint32_t n_blocks,
                                                      TierLists out) {
  __shared__ uint32_t base[NT];      // running position of each tier inside this block
  __shared__ uint32_t wave_cnt[NT][4];
  if (threadIdx.x < NT) base[threadIdx.x] = block_hist[threadIdx.x * n_blocks + blockIdx.x];
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t b0 = (uint64_t)blockIdx.x * TIER_ITEMS;
  for (uint32_t r = 0; r < TIER_ITEMS / 256; r++) {
    const uint64_t i = b0 + (uint64_t)r * 256 + threadIdx.x;
    const int tk = (i < n && tier[i] < NT) ? (int)tier[i] : -1;   // bins >= NT: not listed
    uint32_t rank = 0;
#pragma unroll
    for (int k = 0; k < NT; k++) {
      const uint64_t m = __ballot(tk == k);
      if (tk == k) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wave_cnt[k][wv] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (tk >= 0) {
      uint32_t off = base[tk] + rank;
      for (uint32_t w = 0; w < wv; w++) off += wave_cnt[tk][w];
      out.list[tk][off] = (uint32_t)i;
    }
    __syncthreads();
    if (threadIdx.x < NT)
      base[threadIdx.x] += wave_cnt[threadIdx.x][0] + wave_cnt[threadIdx.x][1] + wave_cnt[threadIdx.x][2] +
                           wave_cnt[threadIdx.x][3];
    __syncthreads();
  }
}

// ---- banded anti-diagonal kernel ---------------------------------------------------------------
// Exact pruning with an a-posteriori certificate.  The group sweeps the 64 diagonals around the
// seed diagonal; the best score S1 found there is the score of a real alignment, hence a lower
// bound of the optimum.  Any alignment with score S >= S1 and g gap bases pays at least
// cost(g) = gO + (g-1) gE (gE < gO), so it has m >= m0(g) = ceil((S1 + cost(g)) / match) matches,
// uses >= m0(g) rows and columns, starts on a diagonal d = j - i in [-(L - m0(g)), W - m0(g)] and
// stays within g of it.  If the union of those ranges over all feasible g lies inside the swept
// band, the band contained EVERY alignment scoring >= S1 -- in particular all optimal ones, which
// is all the reference's answer depends on (cells fed from outside the band can only come out
// lower, never higher, so they cannot win a maximum or a tie) -- and the result is exact.
// Otherwise the candidate is flagged for the full-matrix kernel.
// Sweep: GL lanes x DPL adjacent diagonals; on every anti-diagonal step k = i + j a lane computes
// the DPL / 2 cells of its diagonals with the parity of k.  A cell takes E from diagonal d-1 and F
// from diagonal d+1 (both from step k-1: own registers, or one DPP row shift at the lane
// boundary) and its own diagonal's H from step k-2.
// GL = lanes per candidate: 8 (32 diagonals, 8 candidates per wave) or 16 (64 diagonals, 4 per wave).
// `list` (optional) maps work items to candidates; todo[] is indexed by work item.
template <int LMAX, int GL, int DPL, int BS = 256>
__global__ __launch_bounds__(BS) void k_sw_band(kslam_overlap *__restrict__ ov, uint64_t n_cap, SwInputs in, SwParams p,
                                                 uint32_t *__restrict__ band0, const uint32_t *__restrict__ list,
                                                 Tiers T, int self, const uint32_t *__restrict__ n_dev, uint32_t first,
                                                 uint32_t n_sure) {
  // n_dev: the list's length on the device, read now (the list was still growing when the host sized this launch for n_cap
  // entries from `first` on); a workgroup beyond it leaves at once.  n_sure entries were there when the host looked: a
  // workgroup inside them does not wait for the load
  uint64_t n = n_cap;
  if (n_dev && (uint64_t)(blockIdx.x + 1) * (BS / GL) > n_sure) {
    const uint32_t tot = *n_dev;
    n = min(n_cap, (uint64_t)(tot > first ? tot - first : 0u));
  }
  if ((uint64_t)blockIdx.x * (BS / GL) >= n) return;   // (block-uniform)
  list += first;
  constexpr int NG = BS / GL;           // candidates per block
  constexpr int ND = DPL * GL;          // diagonals swept (DPL adjacent diagonals per lane)
  // The sweep also computes cells that lie outside the matrix near its corners (no per-cell range
  // test): the score rows and the window are padded by PADM entries of "scores 0 against
  // everything" on both sides.  Such cells can only hold values derived from real ones by standing
  // still or paying for gaps, so they never beat a real maximum (strictly), and nothing flows from
  // them back into the matrix: before the matrix they hold exactly the zero-score value Z a fresh
  // alignment starts from, after it every dependency points further out.
  constexpr int PADM = ND == 16 ? 16 : (ND <= 48 ? 32 : (ND == 64 ? 48 : 80));   // >= ND / 2 + 2, x16
  constexpr int ROW = LMAX + STAGE_PAD + 2 * PADM;
  __shared__ __attribute__((aligned(16))) uint8_t s_q[NG][LMAX + STAGE_PAD];
  __shared__ __attribute__((aligned(16))) uint8_t s_w[NG][ROW];
  __shared__ __attribute__((aligned(16))) uint32_t s_tab[NG][ROW];
  const int32_t lane = threadIdx.x & 63;
  const int32_t t = lane & (GL - 1);
  const int32_t grp = threadIdx.x / GL;
  const uint64_t gi = (uint64_t)blockIdx.x * NG + grp;
  const bool have = gi < n;
  const uint64_t ci = have ? (list ? list[gi] : gi) : 0;
  int32_t L = 0, W = 0, rel = 0;
  const uint8_t *qc = s_q[grp];
  uint8_t *wc = s_w[grp] + PADM;         // window codes x 6 (the bfe offset)
  uint32_t *tab = s_tab[grp] + PADM;     // 6-bit packed score row per read base
  if (have && p.ablate < 2) {
    const kslam_overlap o = ov[ci];
    rel = o.rel;
    const Staged st = stage_candidate_wide<GL, 6>(o, in, t, s_q[grp], wc, tab, p, 2 * p.gap_extend);
    L = st.L;
    W = st.W;
    qc += st.qoff;
    wc += st.woff;
    tab += st.qoff;
  }
  __syncthreads();
  const uint32_t pad_row = score_row(4u, p, 2 * p.gap_extend);
  for (int32_t x = t; x < PADM; x += GL) {   // the padding, after the chunk stores it overlaps
    tab[x - PADM] = pad_row;
    tab[L + x] = pad_row;
    wc[x - PADM] = 24;                       // code 4 x 6: field 4 of every score row
    wc[W + x] = 24;
  }
  __syncthreads();
  // Gap extension for free: every value is held with gE x (its anti-diagonal k = i + j) added to the
  // score.  Then E' = max(E', H' - (gO - gE)) needs no subtraction for the extension, a diagonal step
  // adds 2 gE (folded into the score table) and the zero floor Z grows by 2 gE per turn like
  // everything else.  The running best is kept relative to the current anti-diagonal's offset.
  const int32_t gE18 = p.gap_extend << KB, gOE = (p.gap_open - p.gap_extend) << KB;
  const int32_t NEG = -((p.gap_open + p.gap_extend + 1) << KB);
  const int32_t d0 = rel < 0 ? rel : 0;          // seed diagonal: read base i sits on window base i + d0
  const int32_t dlo = d0 - ND / 2, dhi = dlo + ND - 1;
  // anti-diagonals k = i + j that hold a matrix cell of some band diagonal: diagonal d spans
  // k = |d| .. (d <= W - L ? 2L - 2 + d : 2W - 2 - d)
  const int32_t kmin = dhi >= 0 ? 0 : -dhi;
  const int32_t dstar = min(max(W - L, dlo), dhi);
  const int32_t kmax = dstar <= W - L ? 2 * L - 2 + dstar : 2 * W - 2 - dstar;
  // diagonals q = 0, 2 have the parity of dlo, q = 1, 3 the other one; phase A runs at k, phase B
  // at k + 1, so the first k has the parity of dlo
  const int32_t k0 = kmin - ((kmin - dlo) & 1);
  // The lane's DPL diagonals d = db + q come in pairs (q = 2h, 2h + 1).  On turn n pair h sits on
  // read row i = ib - h + n (both of its cells: phase B runs one anti-diagonal later) and on window
  // columns jb + h + n and jb + h + 1 + n.  So one score-row pointer and one window pointer per lane
  // serve all cells with constant offsets, and every Z = ((j + 1) << 9) | (i + 1) = 513 i + 512 d + 513
  // is a constant away from the lane's Z of diagonal 0.
  const int32_t db = dlo + DPL * t;
  const int32_t ib = (k0 - db) >> 1;                 // exact: k0 and dlo have the same parity, DPL is even
  // One byte offset into each LDS array per lane, advanced by a single all-VGPR add per turn; the
  // loads use it with immediate offsets (score row of pair h: dword DPL/2 - 1 - h; window codes of
  // pair h: bytes h and h + 1).
  uint8_t *const tab0 = reinterpret_cast<uint8_t *>(&s_tab[0][0]);
  uint8_t *const win0 = &s_w[0][0];
  int32_t tofs = (int32_t)(reinterpret_cast<const uint8_t *>(tab + (ib - (DPL / 2 - 1))) - tab0);
  int32_t wofs = (int32_t)((wc + (ib + db)) - win0);
  const int32_t two_v = in_vgpr(2), eight_v = in_vgpr(8);
  const int32_t Zb0 = 513 * ib + 512 * db + 513 + gE18 * k0;   // Z' of diagonal 0 of the lane on anti-diagonal k0
  // Loop-invariant operands of the sweep live in VGPRs on purpose: on gfx950 the plain 32-bit
  // add / sub / or / and issue at twice the rate when every source is a VGPR or a literal (an SGPR
  // source halves it; tools/valu_peak.hip).
  const int32_t gOEv = in_vgpr(gOE), dZv = in_vgpr(512 + gE18), zincv = in_vgpr(2 * gE18 + 513);
  // ZrA / ZrB: Z' of the right-hand neighbour (i, j + 1, one anti-diagonal on) of the lane's
  // diagonal-0 cell in phase A (anti-diagonal k) and phase B (k + 1)
  int32_t ZrA = Zb0 + 512 + gE18, ZrB = ZrA + gE18;
  int32_t Hd[DPL], Eo[DPL], Fo[DPL];
#pragma unroll
  for (int q = 0; q < DPL; q++) {
    const int h = q >> 1;
    Hd[q] = Zb0 + ((q & 1) ? gE18 : 0) + (-513 * h + 512 * q) - 513 - 2 * gE18;   // the cell before the first one
    Eo[q] = Zb0 + (-513 * ((q + 1) >> 1) + 512 * (q + 1));   // no gap yet: the floor of phase A's cell q + 1 (odd q)
    Fo[q] = NEG;
  }
  // Running best of the lane, by the reference's rule -- highest score, then smallest column, then
  // smallest row (ssw.c:316-342) -- whatever order the cells are visited in (a lane meets column j + 1
  // of one diagonal before column j of the next).  G = (H' | KEYMASK) - Zr is the plain score (the
  // anti-diagonal offset cancels) over the INVERTED position key, so one signed compare decides.
  // The H' that came with the best G (score and origin key) rides along as the low half of a 64-bit
  // (G, H') pair whose maximum is ONE v_max_f64: bit patterns of non-negative doubles order like
  // integers, and with the IEEE bit of the MODE register cleared every NaN pattern -- here: G in
  // [-2^20, 0), only cells scoring <= gE + 3 -- is simply passed over, like the negative ones.
  const int32_t G0 = KEYMASK - 512 - gE18;   // G = G0 + score * 2^18 - key(cell)
  double best = 0.0;                         // (G, H') of the lane's best cell; +0.0: none yet
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0");   // MODE.IEEE = 0 (no float math in this kernel)
  int32_t nturns = have && p.ablate != 1 && L > 0 && W > 0 && kmax >= k0 ? ((kmax - k0) >> 1) + 1 : 0;
  if (p.ablate == 3) nturns = min(nturns, 1);   // measurement only: everything but the sweep
  uint32_t trow[DPL / 2], wcode[DPL / 2 + 1];
  auto cell = [&](int q, int32_t Ein, int32_t Fin) {
    const int h = q >> 1;
    const int32_t s = __builtin_amdgcn_sbfe(trow[h], wcode[h + (q & 1)], 6);
    // The zero floor rides on E: what a cell hands to its right-hand neighbour is max(E, that
    // neighbour's Z), so the neighbour's max3 below already holds its floor and needs no fourth
    // operand.  Older floors carried along are lower in the score field than the newest one, i.e.
    // dominated: every H is the same value as with a separate floor.
    const int32_t Zr = ((q & 1) ? ZrB : ZrA) + (-513 * h + 512 * q);
    const int32_t hv = max(max(Hd[q] + (s << KB), Ein), Fin);
    Hd[q] = hv;
    const int32_t hg = hv - gOEv;
    Eo[q] = max(max(Ein, hg), Zr);
    Fo[q] = max(Fin, hg);
    const int32_t G = sub_vv(hv | KEYMASK, Zr);
    const v2i32 gh = {hv, G};                              // low half, high half
    const double cand = __builtin_bit_cast(double, gh);
    asm("v_max_f64 %0, %1, %2" : "=v"(best) : "v"(best), "v"(cand));
  };
  // All lanes of the wave run the turns of its longest candidate.  A candidate that is done has only
  // cells outside the matrix left (every one has a padding row or a padding column: "scores 0", so no
  // new maximum); it stops advancing its pointers one turn past its last real one, so that it neither
  // re-scores real cells nor runs on into its neighbours' buffers.  The turns every candidate of the
  // wave still needs run without that bookkeeping.
  int32_t nmax = nturns, nmin = nturns;
#pragma unroll
  for (int m = GL; m < 64; m <<= 1) {
    nmax = max(nmax, __shfl_xor(nmax, m, 64));
    nmin = min(nmin, __shfl_xor(nmin, m, 64));
  }
  nmax = __builtin_amdgcn_readfirstlane(nmax);
  nmin = __builtin_amdgcn_readfirstlane(nmin);
  // the DP part of a turn: trow[] / wcode[] hold its score rows and window codes
  auto sweep = [&](int32_t zinc) {
    {  // phase A: the lane's even diagonals; E comes from the odd diagonal below, F from the one above
      const int32_t ein = dpp_row_shr1(Eo[DPL - 1]);
      int32_t e[DPL / 2], f[DPL / 2];
#pragma unroll
      for (int h = 0; h < DPL / 2; h++) {
        e[h] = h == 0 ? (t == 0 ? ZrA - dZv : ein) : Eo[2 * h - 1];   // band edge: no E, just the floor
        f[h] = Fo[2 * h + 1];
      }
#pragma unroll
      for (int h = 0; h < DPL / 2; h++) cell(2 * h, e[h], f[h]);
    }
    {  // phase B: the odd diagonals
      const int32_t fin = __builtin_amdgcn_update_dpp(0, Fo[0], 0x101, 0xF, 0xF, true);  // row_shl:1
      int32_t e[DPL / 2], f[DPL / 2];
#pragma unroll
      for (int h = 0; h < DPL / 2; h++) {
        e[h] = Eo[2 * h];
        f[h] = h == DPL / 2 - 1 ? (t == GL - 1 ? NEG : fin) : Fo[2 * h + 2];
      }
#pragma unroll
      for (int h = 0; h < DPL / 2; h++) cell(2 * h + 1, e[h], f[h]);
    }
    ZrA = add_vv(ZrA, zinc);
    ZrB = add_vv(ZrB, zinc);
  };
  // Two turns per trip while every candidate of the wave still advances: the score rows and window
  // codes of both are fetched together up front (the second turn's are the first's shifted by one
  // entry), so the LDS latency is paid once per two turns and overlaps the first turn's arithmetic.
  int32_t tn = 0;
  for (; tn + 2 <= nmin; tn += 2) {
    uint32_t TT[DPL / 2 + 1], WW[DPL / 2 + 2];
#pragma unroll
    for (int j = 0; j <= DPL / 2; j++) TT[j] = *reinterpret_cast<const uint32_t *>(tab0 + tofs + 4 * j);
#pragma unroll
    for (int x = 0; x <= DPL / 2 + 1; x++) WW[x] = win0[wofs + x];
    tofs = add_vv(tofs, eight_v);
    wofs = add_vv(wofs, two_v);
#pragma unroll
    for (int u = 0; u < 2; u++) {
#pragma unroll
      for (int h = 0; h < DPL / 2; h++) trow[h] = TT[DPL / 2 - 1 - h + u];
#pragma unroll
      for (int h = 0; h <= DPL / 2; h++) wcode[h] = WW[h + u];
      sweep(zincv);
    }
  }
  for (; tn < nmax; tn++) {
#pragma unroll
    for (int h = 0; h < DPL / 2; h++) trow[h] = *reinterpret_cast<const uint32_t *>(tab0 + tofs + 4 * (DPL / 2 - 1 - h));
#pragma unroll
    for (int h = 0; h <= DPL / 2; h++) wcode[h] = win0[wofs + h];
    const int32_t adv = tn < nturns ? 1 : 0;
    tofs += 4 * adv;
    wofs += adv;
    sweep(2 * gE18 + (adv ? 513 : 0));   // a frozen candidate's position key stays put
  }
  // back to the lane's best as (score, origin key) and the cell's position key
  const v2i32 bb = __builtin_bit_cast(v2i32, best);
  const int32_t Gb = bb.y, lbO = bb.x;
  const bool none = (Gb | lbO) == 0;
  const int32_t gv = Gb - G0;                       // score * 2^18 - key(cell), key(cell) in (0, 2^18)
  const int32_t lsc = none ? 0 : (gv + KEYMASK) >> KB;
  const int32_t lbV = (lsc << KB) | (lbO & KEYMASK), lbZ = none ? 0 : (lsc << KB) - gv;
  const PassResult f = reduce_best<GL>(lbV, lbZ);
  // certificate: every alignment scoring >= f.score lies inside [dlo, dlo + ND - 1]
  const bool exact = have && (p.ablate == 3 || band_certifies(f.score, L, W, dlo, ND, p));
  {  // The others move on.  What this band found is a real alignment's score, i.e. a lower bound:
     // it picks the narrowest later tier that is certain to certify (or the full matrix) directly.
     // One atomic per wave and destination reserves the list slots.
    const bool fail = have && t == 0 && !exact;
    int dest = -1;
    if (fail) {
      dest = NT_FULL;
      const int32_t amin = certificate_amin(f.score, L, W, p);
      for (int k = self + 1; k < T.n; k++)
        if (f.score <= 0 || band_holds(amin, L, W, d0 - T.nd[k] / 2, T.nd[k])) {
          dest = k;   // (nothing found at all: just try the next band)
          break;
        }
    }
    for (int k = self + 1; k <= NT_FULL; k++) {
      if (k >= T.n && k != NT_FULL) continue;
      const uint64_t m = __ballot(dest == k);
      if (!m) continue;
      uint32_t base = 0;
      if (lane == (int32_t)__builtin_ctzll(m)) base = atomicAdd(T.counts + k, (uint32_t)__popcll(m));
      base = __shfl(base, __builtin_ctzll(m), 64);
      uint32_t *dl = k == NT_FULL ? T.full_list : T.list[k];
      if (dest == k) dl[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)ci;
    }
  }
  sw_epilogue<GL, 6>(ov, ci, exact, t, L, f, qc, wc, p, band0);
}

}  // namespace

void partition_bins(const uint8_t *d_bins, uint64_t n, uint32_t *const d_lists[8], uint32_t *d_counts, DevBuf &pos,
                    hipStream_t s) {
  if (n == 0) return;
  const uint32_t n_blocks = (uint32_t)((n + TIER_ITEMS - 1) / TIER_ITEMS);
  pos.ensure((size_t)NT * n_blocks * sizeof(uint32_t));
  TierLists TL;
  for (int k = 0; k < NT; k++) TL.list[k] = d_lists[k];
  hipLaunchKernelGGL(k_tier_hist, dim3(n_blocks), dim3(256), 0, s, d_bins, n, pos.as<uint32_t>(), n_blocks);
  hipLaunchKernelGGL(k_tier_scan, dim3(NT), dim3(1024), 0, s, pos.as<uint32_t>(), n_blocks, d_counts);
  hipLaunchKernelGGL(k_tier_scatter, dim3(n_blocks), dim3(256), 0, s, d_bins, n, pos.as<uint32_t>(), n_blocks, TL);
  HIPCHK(hipGetLastError());
}

namespace {
__global__ __launch_bounds__(256) void k_encode(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint64_t n16) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n16) return;
  const uint4 v = src[i];
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t o[4];
#pragma unroll
  for (int d = 0; d < 4; d++) {
    uint32_t out = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const uint32_t ch = (w[d] >> (8 * b)) & 0xFFu;
      const uint32_t code = ssw_code(ch);
      const uint32_t flag = ssw_code_complemented(ch) != code ? 8u : 0u;
      out |= (code | flag) << (8 * b);
    }
    o[d] = out;
  }
  dst[i] = make_uint4(o[0], o[1], o[2], o[3]);
}
}  // namespace


// ---- reads beyond the packed kernels' reach (more than 511 bases, or match x length beyond the 13-bit score field) ----
// ssw_align (src/ssw.c:841-951) takes any length; merged read pairs and the odd long read are real inputs.  This kernel is
// the reference's two passes stated plainly: one wavefront per candidate, 32-bit scores, the matrix swept by
// anti-diagonals with H / E / F of the last two anti-diagonals in LDS (indexed by read row, updated in place from the
// highest row down, so that a row's upper neighbours are still the previous anti-diagonal's).  Forward pass
// (:870-877): highest score, then smallest end column, then smallest end row (:316-342); reverse pass (:906-923) over
// the two prefixes read backwards: the first column, then the smallest row, at which the score is reached again.
// Slow (a 2 000-base read: ~1 ms of one wavefront per candidate) and exact; only the chunks of a batch that hold such reads
// come here (kslam_api.hip splits a batch into runs of short and long reads).
struct LongBest {
  long long key;   // score << 40 | (0xFFFFF - column) << 20 | (0xFFFFF - row): one signed maximum is the reference's rule
};
__device__ inline long long long_pass(const uint8_t *sq, const uint8_t *sw, int32_t L, int32_t W, bool rev, int32_t q_last,
                                      int32_t w_last, const SwParams &p, int32_t *A2, int32_t *A1, int32_t *E, int32_t *F,
                                      int32_t lane) {
  // rev: row i reads sq[q_last - i], column j reads sw[w_last - j] (the reverse pass over the prefixes)
  constexpr int32_t NEG = -(1 << 29);
  for (int32_t i = lane; i <= L; i += 64) { A2[i] = 0; A1[i] = 0; E[i] = NEG; F[i] = NEG; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  long long best = 0;
  const int32_t gO = p.gap_open, gE = p.gap_extend;
  // arrays are indexed by row + 1 (slot 0 = the row above the matrix: H = 0, F = NEG)
  for (int32_t k = 0; k <= L + W - 2; k++) {
    const int32_t lo = max(0, k - (W - 1)), hi = min(L - 1, k);       // rows of this anti-diagonal
    // from the highest rows down: a chunk's upper neighbour row belongs to the chunk processed after it
    for (int32_t top = hi; top >= lo; top -= 64) {
      const int32_t i = top - lane;
      const bool live = i >= lo;
      int32_t h = 0, e = NEG, f = NEG;
      if (live) {
        const int32_t j = k - i;
        const uint32_t qc = sq[rev ? q_last - i : i], wc = sw[rev ? w_last - j : j];
        const int32_t sc = (qc > 3u || wc > 3u) ? 0 : (qc == wc ? p.match : -p.mismatch);
        const int32_t hd = (i > 0 && j > 0) ? A2[i] : 0;             // H(i - 1, j - 1): anti-diagonal k - 2, row i - 1
        const int32_t hl = j > 0 ? A1[i + 1] : 0, el = j > 0 ? E[i + 1] : NEG;   // (i, j - 1): k - 1, row i
        const int32_t hu = i > 0 ? A1[i] : 0, fu = i > 0 ? F[i] : NEG;           // (i - 1, j): k - 1, row i - 1
        e = max(el - gE, hl - gO);
        f = max(fu - gE, hu - gO);
        h = max(max(hd + sc, 0), max(e, f));
        const long long key = ((long long)h << 40) | ((long long)(0xFFFFF - j) << 20) | (long long)(0xFFFFF - i);
        best = h > 0 && key > best ? key : best;
      }
      __builtin_amdgcn_wave_barrier();      // every lane has read its neighbours of the previous anti-diagonals
      if (live) { A2[i + 1] = h; E[i + 1] = e; F[i + 1] = f; }   // A2 becomes this anti-diagonal's H row
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    int32_t *t = A2; A2 = A1; A1 = t;       // k - 1 becomes k - 2, the rows just written are k - 1
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const long long o = __shfl_xor(best, m, 64);
    best = o > best ? o : best;
  }
  return best;
}

__global__ __launch_bounds__(64) void k_sw_long(kslam_overlap *__restrict__ ov, uint64_t n, SwInputs in, SwParams p,
                                                uint32_t *__restrict__ band0, uint32_t lcap) {
  extern __shared__ __attribute__((aligned(16))) uint8_t long_lds[];
  const int32_t lane = threadIdx.x;
  const uint64_t ci = blockIdx.x;
  if (ci >= n) return;
  const uint32_t rows = lcap + 2;
  int32_t *A2 = reinterpret_cast<int32_t *>(long_lds), *A1 = A2 + rows, *E = A1 + rows, *F = E + rows;
  uint8_t *sq = reinterpret_cast<uint8_t *>(F + rows), *sw = sq + ((lcap + 16) & ~15u);
  int32_t L = 0, W = 0;
  stage_candidate<64>(ov[ci], in, lane, sq, sw, &L, &W);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  PassResult f{0, 0, 0, 0, 0};
  if (L > 0 && W > 0) {
    // A2 / A1 swap inside the pass: which array ends as which does not matter, both are re-initialised by the next pass
    const long long fw = long_pass(sq, sw, L, W, false, 0, 0, p, A2, A1, E, F, lane);
    if (fw > 0) {
      f.score = (int32_t)(fw >> 40);
      f.end_col = 0xFFFFF - (int32_t)((fw >> 20) & 0xFFFFF);
      f.end_row = 0xFFFFF - (int32_t)(fw & 0xFFFFF);
      const long long bw = long_pass(sq, sw, f.end_row + 1, f.end_col + 1, true, f.end_row, f.end_col, p, A2, A1, E, F, lane);
      // the reverse pass reaches the forward score (the same alignment read backwards); its first column / smallest row
      f.beg_col = f.end_col - (0xFFFFF - (int32_t)((bw >> 20) & 0xFFFFF));
      f.beg_row = f.end_row - (0xFFFFF - (int32_t)(bw & 0xFFFFF));
    }
  }
  sw_epilogue<64, 1>(ov, ci, true, lane, L, f, sq, sw, p, band0);
}

// ---- scoring outside the envelope: the reference's striped kernels evaluated LITERALLY ------------------------------
// Inside `1 <= gapE < gapO, mismatch <= gapO + gapE` the striped evaluation order of sw_sse2_byte / sw_sse2_word is not
// observable and every kernel above computes a plain Gotoh recurrence.  Outside it the reference's answer depends on its
// layout: the Lazy-F loop only EXTENDS F and E is never refreshed after the correction (src/ssw.c:274-305, 512-526), so a
// candidate's score is what 16 (byte) or 8 (word) SSE lanes of segLen = ceil(readLen / lanes) cells each produce in that
// order.  `SLAM --gap-open / --gap-extend` takes any value (src/main.cpp:44-55), so this kernel plays the lanes: one
// wavefront per candidate, lane l < W is SSE lane l (the others idle), pvHStore / pvHLoad / pvE / pvHmax are LDS arrays of
// segLen x W cells, the query profile is computed on the fly.  It follows sw_sse2_byte (src/ssw.c:143-383) and sw_sse2_word
// (:408-592) statement by statement -- the tests hold it to the real ssw.c on scorings outside the envelope -- and
// ssw_align's sequence (:870-923): byte pass, word pass when the byte pass saturates, reverse pass over the reversed
// prefixes with `terminate = score1`.  Slow (~20 us of one wavefront per candidate) and exact.
struct StripedEnd { int32_t score, ref, read; };

template <int W>
__device__ inline int32_t lanes_max(int32_t v, int32_t lane) {   // over SSE lanes 0 .. W-1, result in every lane
  v = lane < W ? v : INT32_MIN;
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = max(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ inline int32_t sat_u8(int32_t v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ inline int32_t subs_u16(int32_t a, int32_t b) { return a > b ? a - b : 0; }          // on values in 0..65535
__device__ inline int32_t adds_i16(int32_t a, int32_t b) { const int32_t v = a + b; return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }

// sw_sse2_byte (BYTE) / sw_sse2_word over ref[0 .. refLen) in direction ref_dir against rd[0 .. readLen): sq_at(j) gives the
// read code of profile position j, rf_at(i) the window code of column i.  HS / HL / EE / HM: LDS, segLen x W uint16 each.
template <bool BYTE, typename ReadAt, typename RefAt>
__device__ inline StripedEnd striped_pass(ReadAt rd_at, int32_t readLen, RefAt rf_at, int32_t refLen, int ref_dir, const SwParams &p,
                                          int32_t bias, int32_t terminate, uint16_t *HS, uint16_t *HL, uint16_t *EE, uint16_t *HM,
                                          int32_t lane) {
  constexpr int W = BYTE ? 16 : 8;
  const int32_t segLen = (readLen + W - 1) / W;
  const bool on = lane < W;
  const int32_t gO = p.gap_open, gE = p.gap_extend;
  for (int32_t k = lane; k < segLen * W; k += 64) { HS[k] = 0; HL[k] = 0; EE[k] = 0; HM[k] = 0; }
  wave_sync();
  int32_t vMaxScore = 0, vMaxMark = 0;      // per SSE lane
  int32_t best = 0;
  int32_t end_read = readLen - 1, end_ref = BYTE ? -1 : 0;
  int32_t begin = 0, end = refLen, step = 1;
  if (ref_dir == 1) { begin = refLen - 1; end = -1; step = -1; }
  auto score_of = [&](uint32_t rc, int32_t j) -> int32_t {   // qP_byte / qP_word: mat[ref][read[j]] (+ bias), padding bias / 0
    if (j >= readLen) return BYTE ? bias : 0;
    const uint32_t qc = rd_at(j);
    const int32_t m = (qc > 3u || rc > 3u) ? 0 : (qc == rc ? p.match : -p.mismatch);
    return BYTE ? (int32_t)(uint8_t)(int8_t)(m + bias) : m;
  };
  for (int32_t i = begin; i != end; i += step) {
    int32_t vF = 0, vMaxColumn = 0;
    int32_t vH = 0;
    if (on && lane > 0) vH = HS[(segLen - 1) * W + lane - 1];   // pvHStore[segLen - 1] shifted by one lane
    wave_sync();
    { uint16_t *t = HL; HL = HS; HS = t; }
    const uint32_t rc = rf_at(i);
    for (int32_t j = 0; j < segLen; j++) {
      if (on) {
        const int32_t pv = score_of(rc, j + lane * segLen);
        int32_t h;
        if (BYTE) { h = sat_u8(vH + pv); h = sat_u8(h - bias); }
        else h = adds_i16(vH, pv);
        const int32_t e = EE[j * W + lane];
        h = max(h, e);
        h = max(h, vF);
        vMaxColumn = max(vMaxColumn, h);
        HS[j * W + lane] = (uint16_t)h;
        const int32_t hg = BYTE ? sat_u8(h - gO) : subs_u16(h, gO);
        int32_t ee = BYTE ? sat_u8(e - gE) : subs_u16(e, gE);
        ee = max(ee, hg);
        EE[j * W + lane] = (uint16_t)ee;
        int32_t ff = BYTE ? sat_u8(vF - gE) : subs_u16(vF, gE);
        ff = max(ff, hg);
        vF = ff;
        vH = HL[j * W + lane];
      }
    }
    wave_sync();
    if (BYTE) {   // Lazy_F, src/ssw.c:274-305
      int32_t j = 0;
      vH = on ? HS[lane] : 0;
      { const int32_t up = __shfl_up(vF, 1, 64); vF = (on && lane > 0) ? up : 0; }
      for (;;) {
        const int32_t t = sat_u8(vF - sat_u8(vH - gO));
        if (!__any(on && t != 0)) break;
        if (on) {
          vH = max(vH, vF);
          vMaxColumn = max(vMaxColumn, vH);
          HS[j * W + lane] = (uint16_t)vH;
          vF = sat_u8(vF - gE);
        }
        j++;
        if (j >= segLen) {
          j = 0;
          const int32_t up = __shfl_up(vF, 1, 64);
          vF = (on && lane > 0) ? up : 0;
        }
        if (on) vH = HS[j * W + lane];
      }
    } else {      // Lazy_F, src/ssw.c:514-526: vMaxColumn is not refreshed here
      bool done = false;
      for (int k = 0; k < W && !done; k++) {
        { const int32_t up = __shfl_up(vF, 1, 64); vF = (on && lane > 0) ? up : 0; }
        for (int32_t j = 0; j < segLen; j++) {
          bool mine = false;
          if (on) {
            int32_t h = HS[j * W + lane];
            h = max(h, vF);
            HS[j * W + lane] = (uint16_t)h;
            const int32_t hg = subs_u16(h, gO);
            vF = subs_u16(vF, gE);
            mine = vF > hg;
          }
          if (!__any(mine)) { done = true; break; }
        }
      }
    }
    wave_sync();
    // src/ssw.c:307-325 / :528-543
    vMaxScore = max(vMaxScore, vMaxColumn);
    const bool differs = __any(on && vMaxMark != vMaxScore);
    bool stop = false;
    if (differs) {
      vMaxMark = vMaxScore;
      const int32_t temp = lanes_max<W>(vMaxScore, lane);
      if (temp > best) {
        best = temp;
        if (BYTE && best + bias >= 255) stop = true;   // overflow: the byte pass gives up (:318)
        else {
          end_ref = i;
          for (int32_t k = lane; k < segLen * W; k += 64) HM[k] = HS[k];
          wave_sync();
        }
      }
    }
    if (stop) break;
    const int32_t colmax = lanes_max<W>(vMaxColumn, lane);
    if (colmax == terminate) break;                    // :330 / :545
  }
  wave_sync();
  // src/ssw.c:334-342 / :549-557: the smallest read index at which the end column holds the maximum
  int32_t er = end_read;
  for (int32_t k = lane; k < segLen * W; k += 64)
    if ((int32_t)HM[k] == best) er = min(er, k / W + (k % W) * segLen);
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) er = min(er, __shfl_xor(er, m, 64));
  StripedEnd r;
  r.score = BYTE ? (best + bias >= 255 ? 255 : best) : best;
  r.ref = end_ref;
  r.read = er;
  return r;
}

__global__ __launch_bounds__(64) void k_sw_striped(kslam_overlap *__restrict__ ov, uint64_t n, SwInputs in, SwParams p,
                                                   uint32_t *__restrict__ band0, uint32_t lcap) {
  extern __shared__ __attribute__((aligned(16))) uint8_t striped_lds[];
  const int32_t lane = threadIdx.x;
  const uint64_t ci = blockIdx.x;
  if (ci >= n) return;
  const uint32_t cells = lcap + 16;   // segLen x W <= readLen + W - 1
  uint16_t *HS = reinterpret_cast<uint16_t *>(striped_lds), *HL = HS + cells, *EE = HL + cells, *HM = EE + cells;
  uint8_t *sq = reinterpret_cast<uint8_t *>(HM + cells), *sw = sq + ((lcap + 16) & ~15u);
  int32_t L = 0, Wn = 0;
  stage_candidate<64>(ov[ci], in, lane, sq, sw, &L, &Wn);
  wave_sync();
  PassResult f{0, 0, 0, 0, 0};
  if (L > 0 && Wn > 0) {
    const int32_t bias = max(p.mismatch, 0);   // |min(mat)|, src/ssw.c:819-822 (the matrix's only negative entry is -mismatch)
    auto rd_fw = [&](int32_t j) -> uint32_t { return sq[j]; };
    auto rf = [&](int32_t i) -> uint32_t { return sw[i]; };
    StripedEnd b = striped_pass<true>(rd_fw, L, rf, Wn, 0, p, bias, 255 /* (uint8_t)-1 */, HS, HL, EE, HM, lane);
    bool word = false;
    if (b.score == 255) {   // :873-877
      b = striped_pass<false>(rd_fw, L, rf, Wn, 0, p, 0, 65535 /* (uint16_t)-1 */, HS, HL, EE, HM, lane);
      word = true;
    }
    if (b.score > 0) {
      f.score = b.score;
      f.end_col = b.ref;
      f.end_row = b.read;
      // reverse pass (:906-923): the read prefix reversed, the window prefix scanned right to left, stop at score1
      const int32_t q_last = b.read;
      auto rd_rv = [&](int32_t j) -> uint32_t { return sq[q_last - j]; };
      const StripedEnd r = word ? striped_pass<false>(rd_rv, b.read + 1, rf, b.ref + 1, 1, p, 0, b.score, HS, HL, EE, HM, lane)
                                : striped_pass<true>(rd_rv, b.read + 1, rf, b.ref + 1, 1, p, bias, b.score, HS, HL, EE, HM, lane);
      f.beg_col = r.ref;
      f.beg_row = b.read - r.read;
    }
  }
  sw_epilogue<64, 1>(ov, ci, true, lane, L, f, sq, sw, p, band0);
}

void encode_bases(const uint8_t *d_src, uint8_t *d_dst, uint64_t n, hipStream_t s) {
  const uint64_t n16 = (n + 15) / 16;   // both arrays carry 64 bytes of slack
  if (n16) hipLaunchKernelGGL(k_encode, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s,
                              reinterpret_cast<const uint4 *>(d_src), reinterpret_cast<uint4 *>(d_dst), n16);
  HIPCHK(hipGetLastError());
}

void sw_scores(kslam_overlap *d_ov, uint64_t n, SwInputs in, SwParams p, uint32_t max_read_len,
               uint32_t *d_band0, SwWork &W, uint64_t *n_full_out, const Tuning &tune, hipStream_t s, bool long_reads) {
  if (n_full_out) *n_full_out = 0;
  if (n == 0) return;
  if (long_reads) {   // a chunk of reads the packed kernels cannot hold: every candidate through the plain two-pass kernel
    const uint32_t lcap = (max_read_len + 15u) & ~15u;
    const size_t lds = (size_t)4 * (lcap + 2) * sizeof(int32_t) + 2 * ((size_t)lcap + 16);
    if (lds > 160 * 1024) throw StatusError{KSLAM_ERR_UNSUPPORTED, "reads longer than 9000 bases are not supported"};
    if (lds > 64 * 1024)
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sw_long), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (p.striped) {   // scoring outside the envelope: the reference's striped evaluation, literally
      const size_t lds2 = (size_t)4 * (lcap + 16) * sizeof(uint16_t) + 2 * ((size_t)lcap + 16);
      if (lds2 > 64 * 1024)
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sw_striped), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      for (uint64_t lo = 0; lo < n; lo += 1u << 30) {
        const uint64_t m = std::min<uint64_t>(n - lo, 1u << 30);
        hipLaunchKernelGGL(k_sw_striped, dim3((unsigned)m), dim3(64), lds2, s, d_ov + lo, m, in, p, d_band0 + lo, lcap);
      }
      HIPCHK(hipGetLastError());
      if (n_full_out) *n_full_out = n;
      return;
    }
    for (uint64_t lo = 0; lo < n; lo += 1u << 30) {
      const uint64_t m = std::min<uint64_t>(n - lo, 1u << 30);
      hipLaunchKernelGGL(k_sw_long, dim3((unsigned)m), dim3(64), lds, s, d_ov + lo, m, in, p, d_band0 + lo, lcap);
    }
    HIPCHK(hipGetLastError());
    if (n_full_out) *n_full_out = n;
    return;
  }
  if (max_read_len > 511) throw StatusError{KSLAM_ERR_UNSUPPORTED, "a chunk of short reads holds a read longer than 511 bases"};
  if (n >= (1ull << 32)) throw StatusError{KSLAM_ERR_UNSUPPORTED, ">= 2^32 candidates in one chunk"};
  const int lm = max_read_len <= 160 ? 0 : (max_read_len <= 256 ? 1 : 2);
  const uint32_t *full_list = nullptr;
  uint64_t n_full = n;
  // the band kernels carry gE x (i + j) on top of the score and 2 gE inside the 6-bit table fields
  const bool band_ok = (int64_t)(p.match + 2 * p.gap_extend) * (int64_t)max_read_len <= 8187 &&
                       p.match + 2 * p.gap_extend <= 31;
  const bool debug = tune.debug;
#ifdef KSLAM_ABLATE
  p.ablate = tune.sw_ablate;
#endif
  if (!tune.sw_full && band_ok) {
    // banded tiers of 16 / 32 / 48 / 64 / 96 (/ 128 for reads > 160 bases) diagonals; k_sw_plan sends each
    // candidate to the narrowest one its seed diagonal already certifies, the others start at 48 (64 for the
    // longer reads); whatever fails a tier's certificate is appended to the list of the tier its score
    // certifies, and what no tier certifies goes to the full-matrix kernel
    Tiers T;
    memset(&T, 0, sizeof T);
    {
      const int nd0[5] = {16, 32, 48, 64, 96}, nd1[6] = {16, 32, 48, 64, 96, 128};
      const bool no48 = tune.sw_no48, no96 = tune.sw_no96;   // A/B of the tier set
      T.n = 0;
      for (int k = 0; k < (lm == 0 ? 5 : 6); k++) {
        const int nd = lm == 0 ? nd0[k] : nd1[k];
        if ((nd == 48 && no48) || (nd == 96 && no96)) continue;
        T.nd[T.n++] = nd;
      }
    }
    // Candidates whose diagonal sums certify nothing are the gapped ones; on the bench workload 62 % of
    // them end up needing more than 32 diagonals, so starting them at 48 (3.1 us) is cheaper than 32
    // first (2.25 us) and 48 again for most.
    {
      // (reads of 161-256 bases: 64 -- 83.2 against 85.0 ms per 1 M pairs of 250 bp, 89.5 when started at 32)
      const int unk = tune.sw_unknown_nd ? tune.sw_unknown_nd : (lm == 0 ? 48 : 64);
      T.unknown = 1;
      for (int k = 0; k < T.n; k++) if (T.nd[k] <= unk) T.unknown = k;
    }
    W.flags.ensure(n);                                   // tier per candidate (u8)
    for (int k = 0; k < T.n; k++) W.tier_list[k].ensure((n + 1) * sizeof(uint32_t));
    W.list.ensure((n + 1) * sizeof(uint32_t));           // list for the full-matrix kernel
    const uint32_t n_blocks = (uint32_t)((n + TIER_ITEMS - 1) / TIER_ITEMS);
    W.pos.ensure((size_t)NT * n_blocks * sizeof(uint32_t));
    W.totals.ensure(16 * sizeof(uint32_t));
    uint8_t *tier = W.flags.as<uint8_t>();
    uint32_t *counts = W.totals.as<uint32_t>();          // [k] tier sizes, [NT_FULL] full-matrix list size
    HIPCHK(hipMemsetAsync(counts, 0, 16 * sizeof(uint32_t), s));
    TierLists TL;
    for (int k = 0; k < NT; k++) TL.list[k] = W.tier_list[std::min(k, NT_MAX - 1)].as<uint32_t>();
    for (int k = 0; k < T.n; k++) T.list[k] = W.tier_list[k].as<uint32_t>();
    T.full_list = W.list.as<uint32_t>();
    T.counts = counts;
    const unsigned pb = (unsigned)std::min<uint64_t>((n + 31) / 32, 256 * (uint64_t)tune.plan_blocks_per_cu);   // its waves walk through the candidates
    if (lm == 0) hipLaunchKernelGGL(k_sw_plan<160>, dim3(pb), dim3(256), 0, s, d_ov, n, in, p, T, tier, d_band0);
    else if (lm == 1) hipLaunchKernelGGL(k_sw_plan<256>, dim3(pb), dim3(256), 0, s, d_ov, n, in, p, T, tier, d_band0);
    else hipLaunchKernelGGL(k_sw_plan<512>, dim3(pb), dim3(256), 0, s, d_ov, n, in, p, T, tier, d_band0);
    hipLaunchKernelGGL(k_tier_hist, dim3(n_blocks), dim3(256), 0, s, tier, n, W.pos.as<uint32_t>(), n_blocks);
    hipLaunchKernelGGL(k_tier_scan, dim3(NT), dim3(1024), 0, s, W.pos.as<uint32_t>(), n_blocks, counts);
    hipLaunchKernelGGL(k_tier_scatter, dim3(n_blocks), dim3(256), 0, s, tier, n, W.pos.as<uint32_t>(), n_blocks, TL);
    uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (debug) {
      read_back(h, counts, sizeof h, s);
      fprintf(stderr, "[kslam] SW planned: %u / %u / %u / %u / %u / %u\n", h[0], h[1], h[2], h[3], h[4], h[5]);
    }
    // One sweep over the tiers without a read-back in front of each (round 6; ~25 us of idle GPU at each of them).  Tier k's
    // list = what k_sw_plan put there + what the tiers before it send on -- unknown to the host when it queues tier k, known
    // to the kernel when it runs (n_dev).  The launch is sized for planned + 1.2 x the inflow the LAST chunk of this context
    // saw at that tier (scaled by the chunks' sizes) + 4096: chunks of one run are statistically alike.  What did not fit,
    // and what that sends on, is left to rounds after the one read-back: each launches the part of every list nobody has
    // run yet.  A context's first chunk (no history) and a changed tier set go tier by tier as before.
    auto launch_tier = [&](int k, uint64_t m, const uint32_t *n_dev, uint32_t first, uint32_t n_sure = 0) {
      const int nd = T.nd[k];
      const uint32_t *list = T.list[k];
#define KSLAM_BAND(LM, GLV, DPLV, BSV) \
  hipLaunchKernelGGL((k_sw_band<LM, GLV, DPLV, BSV>), dim3((unsigned)((m + (BSV / GLV) - 1) / (BSV / GLV))), dim3(BSV), 0, s, \
                     d_ov, m, in, p, d_band0, list, T, k, n_dev, first, n_sure)
#define KSLAM_BAND_LM(GLV, DPLV, BSV) \
  do { if (lm == 0) KSLAM_BAND(160, GLV, DPLV, BSV); else if (lm == 1) KSLAM_BAND(256, GLV, DPLV, BSV); \
       else KSLAM_BAND(512, GLV, DPLV, BSV); } while (0)
      // lanes x diagonals per lane, measured on the bench workload: 8 x 8 beats 16 x 4 for the
      // 64-diagonal tier (9.7 against 10.5 ms); 4-lane shapes lose (half the waves per LDS byte)
      if (nd == 16) KSLAM_BAND_LM(8, 2, 256);
      else if (nd == 32) KSLAM_BAND_LM(8, 4, 256);
      else if (nd == 48) KSLAM_BAND_LM(8, 6, 128);
      else if (nd == 64) KSLAM_BAND_LM(8, 8, 128);
      else if (nd == 96) KSLAM_BAND_LM(16, 6, 256);
      else KSLAM_BAND_LM(16, 8, 256);
#undef KSLAM_BAND_LM
#undef KSLAM_BAND
    };
    uint32_t planned[NT_MAX] = {0, 0, 0, 0, 0, 0}, done[NT_MAX] = {0, 0, 0, 0, 0, 0};
    read_back(h, counts, sizeof h, s);
    for (int k = 0; k < T.n; k++) planned[k] = h[k];
    const bool history = W.last_n && W.last_tiers == T.n && W.last_lm == lm && tune.sw_sweep;
    if (history) {
      const double scale = (double)n / (double)W.last_n;
      for (int k = 0; k < T.n; k++) {
        const uint64_t room = k && tune.sweep_room ? (uint64_t)(1.2 * scale * W.last_inflow[k]) + 4096 : 0;
        const uint64_t cap = std::min<uint64_t>(n, planned[k] + room);
        if (debug) fprintf(stderr, "[kslam] SW tier %d (%d diagonals): %u planned, sized for %llu\n", k, T.nd[k], planned[k], (unsigned long long)cap);
        if (cap) launch_tier(k, cap, counts + k, 0, planned[k]);
        done[k] = (uint32_t)cap;
      }
    }
    for (int round = 0;; round++) {
      if (round || history) read_back(h, counts, sizeof h, s);
      bool progressed = false;
      for (int k = 0; k < T.n; k++) {
        done[k] = std::min(done[k], h[k]);
        if (h[k] <= done[k]) continue;
        const uint64_t m = h[k] - done[k];
        if (debug) fprintf(stderr, "[kslam] SW round %d tier %d (%d diagonals): %llu candidates%s\n", round, k, T.nd[k], (unsigned long long)m, history ? " left over" : "");
        launch_tier(k, m, nullptr, done[k]);
        done[k] = h[k];
        progressed = true;
        if (!history) {    // tier by tier: the next tier's size is known once this one has run
          read_back(h, counts, sizeof h, s);
        }
      }
      if (!progressed) break;
    }
    W.last_n = n;
    W.last_tiers = T.n;
    W.last_lm = lm;
    for (int k = 0; k < T.n; k++) W.last_inflow[k] = h[k] - planned[k];
    read_back(h, counts, sizeof h, s);      // (the full-matrix list's length: the last round's launches may have added to it)
    n_full = h[NT_FULL];
    full_list = W.list.as<uint32_t>();
    if (debug) fprintf(stderr, "[kslam] SW full matrix: %llu candidates\n", (unsigned long long)n_full);
  }
  if (n_full_out) *n_full_out = n_full;
  if (n_full) {  // full matrix for the rest
    const uint64_t m = n_full;
    const unsigned b2 = (unsigned)((m + 15) / 16);
    if (lm == 0) hipLaunchKernelGGL(k_sw<10>, dim3(b2), dim3(256), 0, s, d_ov, m, in, p, d_band0, full_list);
    else if (lm == 1) hipLaunchKernelGGL(k_sw<16>, dim3(b2), dim3(256), 0, s, d_ov, m, in, p, d_band0, full_list);
    else hipLaunchKernelGGL(k_sw<32>, dim3(b2), dim3(256), 0, s, d_ov, m, in, p, d_band0, full_list);
  }
  HIPCHK(hipGetLastError());
}

}  // namespace kslam
