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
// reverse scan; oracle/kslam_oracle.c origin_pass is the CPU statement of it
// (80k random + 60k low-complexity trials equal to the striped emulation).  The striped Lazy-F evaluation order
// is only observable when a gap pair can beat a mismatch or when gapE >= gapO;
// kslam_create rejects such scoring.
//
// MI355X design: integer ALU work, no MFMA.  A candidate is mapped onto one
// DPP row (16 lanes) of a wavefront, four candidates per wave.  Lane t owns R
// consecutive query rows; the 16 lanes sweep the DP matrix as a systolic
// anti-diagonal: at step s lane t computes column s - t for its rows, entirely
// in registers (H, E per row; 6-bit packed score table per row so the score is
// one v_bfe_i32).  The only cross-lane traffic per step is two row_shr:1 DPP
// moves: {H of the last row, F leaving the strip} and the running column
// maximum packed as (H << 16 | 0xFFFF - row) so one v_max_u32 both maximises H
// and minimises the row index.  Sequences sit in LDS as 1 byte/base codes.
#include "common.h"

namespace kslam {

namespace {

__device__ inline uint32_t translate_base(uint32_t c) {  // ssw_cpp.cpp:11-23
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case 'U': case 'u': return 0;
    default: return 4;
  }
}
__device__ inline uint32_t complement_base(uint32_t c) {  // sequenceTools.h:98-116
  switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'T': return 'A';
    case 'G': return 'C';
    default: return c;
  }
}

__device__ inline int32_t dpp_row_shr1(int32_t v) {
  // lane i of each 16-lane row receives lane i-1; lane 0 receives 0
  return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
}

constexpr int KB = 18;                    // low bits of a packed DP value: origin key (col << 9 | row)
constexpr int32_t KEYMASK = (1 << KB) - 1;

struct PassResult {
  int32_t score, end_col, end_row, beg_col, beg_row;
};

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
  const int32_t gO = p.gap_open << KB, gE = p.gap_extend << KB;
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
    E[r] = NEG;
    rowkey[r] = i + 1;
  }
  int32_t prev_hl = t * R;                // virtual cell (tR - 1, -1): successor (tR, 0)
  int32_t out_h = 0, out_f = 0;
  int32_t lbV = 0, lbZ = 0;               // lane-local best cell: packed value and its Z (= position + 1)
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
      const int32_t colkey = (c + 1) << 9;
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int32_t s = __builtin_amdgcn_sbfe(tab[r], shift, 6);
        const int32_t Z = colkey | rowkey[r];
        int32_t h = max(max(diag + (s << KB), E[r]), F);
        h = max(h, Z);
        diag = H[r];
        H[r] = h;
        const int32_t hg = h - gO;
        E[r] = max(E[r] - gE, hg);
        F = max(F - gE, hg);
        const bool up = h > (lbV | KEYMASK);       // strictly larger score: first column, smallest row win
        lbV = up ? h : lbV;
        lbZ = up ? Z : lbZ;
      }
      out_h = H[R - 1];
      out_f = F;
    }
  }
  // group reduction: max score, then smallest column, then smallest row
  const int32_t sc = lbV >> KB;
  const int32_t ecol = (lbZ >> 9) - 1, erow = (lbZ & 511) - 1;
  int32_t G = sc > 0 ? ((sc << KB) | ((511 - ecol) << 9) | (511 - erow)) : 0;
  int32_t Gm = G;
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) Gm = max(Gm, __shfl_xor(Gm, m, 16));
  const uint64_t bal = __ballot(G == Gm && Gm != 0);
  const uint32_t grp_bits = (uint32_t)(bal >> (lane & 48)) & 0xFFFFu;
  PassResult res{0, 0, 0, 0, 0};
  const int32_t src = (lane & 48) | (grp_bits ? __builtin_ctz(grp_bits) : 0);
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

template <int R>
__global__ __launch_bounds__(256) void k_sw(kslam_overlap *__restrict__ ov, uint64_t n, SwInputs in, SwParams p,
                                            uint32_t *__restrict__ band0) {
  constexpr int LMAX = R * 16;
  __shared__ uint8_t s_q[16][LMAX];
  __shared__ uint8_t s_w[16][LMAX];
  const int32_t lane = threadIdx.x & 63;
  const int32_t t = lane & 15;
  const int32_t grp = threadIdx.x >> 4;
  const uint64_t ci = (uint64_t)blockIdx.x * 16 + grp;
  const bool have = ci < n;
  int32_t L = 0, wlen = 0;
  if (have) {
    const kslam_overlap o = ov[ci];
    const uint64_t ro = in.read_off[o.read];
    L = (int32_t)(in.read_off[o.read + 1] - ro);
    const uint64_t go = in.genome_off[o.entry];
    const uint64_t G = in.genome_off[o.entry + 1] - go;
    const int64_t s0 = o.rel > 0 ? o.rel : 0;                     // SmithWaterman.h:204
    wlen = (int32_t)min((uint64_t)L, G - (uint64_t)s0);            // substr, :205-206
    const uint32_t revcomp = o.revcomp;
    for (int32_t i = t; i < L; i += 16) s_q[grp][i] = (uint8_t)translate_base(in.read_bases[ro + i]);
    for (int32_t j = t; j < wlen; j += 16) {
      uint32_t ch;
      if (!revcomp) ch = in.genome_bases[go + s0 + j];
      else ch = complement_base(in.genome_bases[go + s0 + (wlen - 1 - j)]);  // :207
      s_w[grp][j] = (uint8_t)translate_base(ch);
    }
  }
  __syncthreads();
  // forward pass (ssw.c:870-877) and, by origin tracking, the result of the reverse pass (:906-923)
  const PassResult f = sw_origin_pass<R>(s_q[grp], L, s_w[grp], have ? wlen : 0, p);
  const bool ok = have && f.score > 0;
  // band request for banded_sw, ssw.c:924-935 (flag 0x0f: score and distance filters)
  const int32_t refLen = f.end_col - f.beg_col + 1, readLen = f.end_row - f.beg_row + 1;
  const bool want = p.report_cigar && ok && (uint32_t)f.score >= (p.score_threshold & 0xFFFFu) &&
                    refLen - 1 <= 32767 && readLen - 1 <= 32767;
  // Ungapped shortcut: when both spans are equal and the plain diagonal already scores
  // `score`, banded_sw's first attempt (band 1) reaches it on the main diagonal, every H
  // direction there is "diagonal" (ties prefer it, ssw.c:686) and the cigar is <n>M.
  int32_t dsum = 0;
  if (want && refLen == readLen) {
    for (int32_t k = t; k < readLen; k += 16) {
      const uint32_t q = s_q[grp][f.beg_row + k], c = s_w[grp][f.beg_col + k];
      dsum += (q > 3u || c > 3u) ? 0 : (q == c ? p.match : -p.mismatch);
    }
  }
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) dsum += __shfl_xor(dsum, m, 16);
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
      if (refLen == readLen && dsum == f.score) {
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

}  // namespace

void sw_scores(kslam_overlap *d_ov, uint64_t n, SwInputs in, SwParams p, uint32_t max_read_len,
               uint32_t *d_band0, hipStream_t s) {
  if (n == 0) return;
  unsigned blocks = (unsigned)((n + 15) / 16);
  if (max_read_len <= 160)
    hipLaunchKernelGGL(k_sw<10>, dim3(blocks), dim3(256), 0, s, d_ov, n, in, p, d_band0);
  else if (max_read_len <= 256)
    hipLaunchKernelGGL(k_sw<16>, dim3(blocks), dim3(256), 0, s, d_ov, n, in, p, d_band0);
  else if (max_read_len <= 511)
    hipLaunchKernelGGL(k_sw<32>, dim3(blocks), dim3(256), 0, s, d_ov, n, in, p, d_band0);
  else
    throw StatusError{KSLAM_ERR_UNSUPPORTED, "reads longer than 511 bases are not supported yet"};
  HIPCHK(hipGetLastError());
}

}  // namespace kslam
