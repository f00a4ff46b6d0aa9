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
// pass, so one int32 DP reproduces both.  The striped Lazy-F evaluation order
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

struct PassResult {
  int32_t score, pos, row;
};

// One SW pass for the 16-lane group this lane belongs to.
//   qcodes[qbase + qdir * i] is query row i (i < qlen), wcodes[col0 + cdir * c]
//   is reference column c (c < ncols).  Result valid in every lane of the group.
template <int R>
__device__ inline PassResult sw_pass(const uint8_t *qcodes, int32_t qlen, int32_t qbase, int32_t qdir,
                                     const uint8_t *wcodes, int32_t ncols, int32_t col0, int32_t cdir,
                                     int32_t terminate, const SwParams &p) {
  const int32_t lane = threadIdx.x & 63;
  const int32_t t = lane & 15;
  const int32_t gsrc = (lane & ~15) | 15;  // sink lane of my group
  uint32_t tab[R];
  int32_t H[R], E[R];
  uint32_t rowtag[R];  // 0xFFFF - row for valid rows, 0 for padding rows
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int32_t i = t * R + r;
    const uint32_t q = i < qlen ? qcodes[qbase + qdir * i] : 4u;
    uint32_t tb = 0;
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
      const int32_t s = q > 3u ? 0 : (q == c ? p.match : -p.mismatch);
      tb |= ((uint32_t)s & 63u) << (6 * c);
    }
    tab[r] = tb;  // column code 4 (N) scores 0: bits 24..29 stay clear
    H[r] = 0;
    E[r] = 0;
    rowtag[r] = i < qlen ? (uint32_t)(0xFFFF - i) : 0u;
  }
  int32_t prev_hl = 0;      // H[last row of lane t-1] one column back = my diagonal
  int32_t out_hf = 0;       // {H last row | F leaving} of the column I just finished
  uint32_t out_cm = 0;      // running column maximum up to and including my rows
  int32_t best = 0, best_pos = cdir > 0 ? 0 : 0, best_row = qlen - 1;
  bool done = false;
  const int32_t nsteps = ncols > 0 ? ncols + 15 : 0;
  // all four groups of the wave iterate together; a group idles once finished
  for (int32_t step = 0;; step++) {
    const bool grp_run = step < nsteps && !done;
    if (__ballot(grp_run) == 0ull) break;
    const int32_t in_hf = dpp_row_shr1(out_hf);
    const uint32_t in_cm = (uint32_t)dpp_row_shr1((int32_t)out_cm);
    const int32_t c = step - t;
    if (grp_run && c >= 0 && c < ncols) {
      const uint32_t refc = wcodes[col0 + cdir * c];
      const uint32_t shift = refc * 6u;
      int32_t diag = prev_hl;
      prev_hl = in_hf & 0xFFFF;
      int32_t F = (int32_t)((uint32_t)in_hf >> 16);
      uint32_t cm = in_cm;
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int32_t s = __builtin_amdgcn_sbfe(tab[r], shift, 6);
        int32_t h = max(max(diag + s, E[r]), F);
        diag = H[r];
        H[r] = h;
        const int32_t tt = max(h - p.gap_open, 0);
        E[r] = max(E[r] - p.gap_extend, tt);
        F = max(F - p.gap_extend, tt);
        const uint32_t tag = rowtag[r];
        cm = max(cm, tag ? (((uint32_t)h << 16) | tag) : 0u);
      }
      out_hf = (H[R - 1] & 0xFFFF) | (F << 16);
      out_cm = cm;
      if (t == 15) {  // the column is complete
        const int32_t cmv = (int32_t)(cm >> 16);
        if (cmv > best) {
          best = cmv;
          best_pos = col0 + cdir * c;
          best_row = 0xFFFF - (int32_t)(cm & 0xFFFFu);
        }
        if (cmv == terminate) done = true;
      }
    }
    done = __shfl((int)done, gsrc, 64) != 0;
  }
  PassResult res;
  res.score = __shfl(best, gsrc, 64);
  res.pos = __shfl(best_pos, gsrc, 64);
  res.row = __shfl(best_row, gsrc, 64);
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
  int64_t s0 = 0;
  uint32_t revcomp = 0;
  if (have) {
    const kslam_overlap o = ov[ci];
    const uint64_t ro = in.read_off[o.read];
    L = (int32_t)(in.read_off[o.read + 1] - ro);
    const uint64_t go = in.genome_off[o.entry];
    const uint64_t G = in.genome_off[o.entry + 1] - go;
    s0 = o.rel > 0 ? o.rel : 0;                                   // SmithWaterman.h:204
    wlen = (int32_t)min((uint64_t)L, G - (uint64_t)s0);            // substr, :205-206
    revcomp = o.revcomp;
    for (int32_t i = t; i < L; i += 16) s_q[grp][i] = (uint8_t)translate_base(in.read_bases[ro + i]);
    for (int32_t j = t; j < wlen; j += 16) {
      uint32_t ch;
      if (!revcomp) ch = in.genome_bases[go + s0 + j];
      else ch = complement_base(in.genome_bases[go + s0 + (wlen - 1 - j)]);  // :207
      s_w[grp][j] = (uint8_t)translate_base(ch);
    }
  }
  __syncthreads();
  // forward pass, ssw.c:870-877 (terminate = -1: never)
  PassResult f = sw_pass<R>(s_q[grp], L, 0, 1, s_w[grp], have ? wlen : 0, 0, 1, -1, p);
  // reverse pass, ssw.c:906-923
  const int32_t rl = f.row + 1, rcols = f.pos + 1;
  const bool rev_ok = have && f.score > 0;
  PassResult b = sw_pass<R>(s_q[grp], rev_ok ? rl : 0, f.row, -1, s_w[grp], rev_ok ? rcols : 0, f.pos, -1,
                            f.score, p);
  if (have && t == 0) {
    kslam_overlap o = ov[ci];
    int32_t ref_begin = -1, read_begin = -1, ref_end = f.pos, read_end = f.row;
    if (rev_ok) {
      ref_begin = b.pos;
      read_begin = f.row - b.row;
    } else {
      ref_end = 0;
    }
    o.score = (uint16_t)f.score;
    o.ref_begin = ref_begin;   // window-relative, unflipped; finalised after the cigar stage
    o.ref_end = ref_end;
    o.query_begin = read_begin;
    o.query_end = read_end;
    o.cigar_len = 0;
    o.cigar_off = 0;
    ov[ci] = o;
    // band request for banded_sw, ssw.c:924-935 (flag 0x0f: score and distance filters)
    uint32_t bw = 0;
    if (p.report_cigar && rev_ok && (uint32_t)f.score >= (p.score_threshold & 0xFFFFu) &&
        ref_end - ref_begin <= 32767 && read_end - read_begin <= 32767) {
      const int32_t a = ref_end - ref_begin + 1, c = read_end - read_begin + 1;
      bw = (uint32_t)(a > c ? a - c : c - a) + 1u;
    }
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
  else if (max_read_len <= 512)
    hipLaunchKernelGGL(k_sw<32>, dim3(blocks), dim3(256), 0, s, d_ov, n, in, p, d_band0);
  else
    throw StatusError{KSLAM_ERR_UNSUPPORTED, "reads longer than 512 bases are not supported yet"};
  HIPCHK(hipGetLastError());
}

}  // namespace kslam
