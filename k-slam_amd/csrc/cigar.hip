// cigar.hip -- banded traceback -> CIGAR, and the final coordinate fix-up.
//
// Replaces banded_sw (reference src/ssw.c:594-792) as called from ssw_align
// (src/ssw.c:924-946) and the revComp un-flip of performSmithWatermanOnRange2
// (reference src/SmithWaterman.h:211-229).
//
// banded_sw is restated step for step, including its three row arrays
// h_b / e_b / h_c with the set_u indexing (ssw.c:56-62) and the sentinel
// assignment h_b[edge] = e_b[edge] = 0 (ssw.c:655) that clobbers a live cell
// when the band is clipped by the reference end -- the CIGAR depends on it.
// Band doubling (ssw.c:693-694) is driven from the host: candidates are binned
// by band class (bw <= 2^c), a class launch runs one attempt for each of its
// candidates and failed ones move to the next class.
//
// MI355X design: O(L * band) scalar work per candidate (about 1 % of the DP
// cells of the scoring passes), so one candidate per lane.  Each wavefront owns
// a scratch slab laid out [element][lane]: lanes run the same row/column loop
// in near lock step, so the row arrays and the 1-byte-per-cell direction matrix
// are touched with coalesced 64-lane accesses that stay in L2.
#include "common.h"
#include "banded_core.h"

namespace kslam {

namespace {

__device__ inline uint32_t band_class(uint32_t bw) {
  uint32_t c = 0;
  while ((1u << c) < bw) c++;
  return c;
}

__global__ void k_class_flags(const uint32_t *__restrict__ bw, const uint8_t *__restrict__ needbig, uint64_t n,
                              uint32_t cls, uint32_t big, uint32_t *__restrict__ flags) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t b = bw[i];
  uint32_t f = 0;
  if (b != 0 && !(b >> 31)) {   // bit 31: inline <n>M set by the SW kernel
    if (big) f = needbig[i] ? 1u : 0u;
    else f = (!needbig[i] && band_class(b) == cls) ? 1u : 0u;
  }
  flags[i] = f;
}

__global__ void k_scatter_list(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ pos, uint64_t n,
                               uint32_t *__restrict__ list) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) list[pos[i]] = (uint32_t)i;
}

__global__ void k_max_bw(const uint32_t *__restrict__ bw, const uint32_t *__restrict__ list, uint32_t m,
                         uint32_t *__restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t v = i < m ? bw[list[i]] : 0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_down((int)v, d, 64));
  if ((threadIdx.x & 63) == 0 && v) atomicMax(out, v);
}

__global__ void k_max_all(const uint32_t *__restrict__ bw, uint64_t n, uint32_t *__restrict__ out) {
  __shared__ uint32_t sm[4];
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t v = i < n ? bw[i] : 0;
  if (v >> 31) v = 0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_down((int)v, d, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    v = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
    if (v) atomicMax(out, v);
  }
}

struct CigJob {
  kslam_overlap *ov;
  uint32_t *bw;
  int32_t *bmax;
  uint8_t *needbig;
  const uint32_t *list;
  uint32_t m;           // list entries in this launch
  uint32_t list_base;   // first list entry of this launch
  uint32_t slot_bw;     // band width the scratch slab is sized for
  uint32_t lmax;        // max read length (row count bound)
  uint32_t cap;         // cigar ops per temp slot
  uint32_t *tmp;        // temp cigar slots: normal: [candidate][cap]; big: [list pos][cap]
  uint32_t big;
  uint8_t *scratch;
  uint64_t wave_slab;   // scratch bytes per wavefront
  uint32_t *err;        // [0] traceback errors
  uint32_t variant;     // timing ablations (KSLAM_CIGAR_VARIANT), 0 in production
};

// banded_sw (ssw.c:594-792): one attempt with J.bw[ci], then traceback when max >= score.
// One candidate per lane; everything the DP touches lives in LDS laid out [element][lane]
// (NL lanes per block, fewer for wide bands): translated read / reference spans, the three
// row arrays, one direction byte per band cell.  The serial chain (f and the left H) runs in
// registers; the previous-row values and the reference code of the next cell are fetched one
// iteration ahead so LDS latency hides behind the chain.
struct LdsLayout {
  uint32_t nl;       // lanes (candidates) per block
  uint32_t lmax;     // rows / columns the sequence buffers are sized for
  uint32_t W1;       // row array length (2 * slot_bw + 4)
  uint32_t wd;       // direction cells per row the global slab is sized for (2 * slot_bw + 1)
  uint32_t wpr;      // direction words per row kept in LDS (six 5-bit cells per word); 0: global slab
};

// Wave-cooperative staging of one span per lane: for each lane c of the wave in turn, all lanes
// fetch c's span with coalesced aligned dword loads and store the SSW codes transposed into
// dst as 4-bit codes, byte (k / 2) * NS + c, nibble k & 1 (NS = NL + 1 spreads the strided
// accesses over the LDS banks; the buffer is zeroed first and filled with LDS atomic ORs).  When
// `rev` the bases are complemented and written back to front (window of a revComp overlap).
__device__ inline void stage_codes_wave(const uint8_t *src, int32_t len, bool rev, uint8_t *dst, uint32_t NS,
                                        uint32_t nl_active) {
  const uint32_t lane = threadIdx.x;
  const uintptr_t a0 = reinterpret_cast<uintptr_t>(src);
  const int32_t maxlen = [&] {   // longest span in the wave (wave-uniform loop bound)
    int32_t m = len;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    return m;
  }();
  const int32_t nchunk = (maxlen + 3 + 255) / 256;   // 64 lanes x 4 bytes per chunk (+3 for misalignment)
  for (int32_t ch = 0; ch < nchunk; ch++) {
    for (uint32_t c0 = 0; c0 < nl_active; c0 += 8) {
      uint32_t v[8];
      int32_t clen[8], shift0[8];
      uint32_t crev[8];
#pragma unroll
      for (uint32_t u = 0; u < 8; u++) {            // 8 candidates' loads in flight together
        const uint32_t c = min(c0 + u, 63u);
        const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)a0, c);
        const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(a0 >> 32), c);
        clen[u] = c0 + u < nl_active ? (int32_t)__builtin_amdgcn_readlane((uint32_t)len, c) : 0;
        crev[u] = __builtin_amdgcn_readlane((uint32_t)rev, c);
        const uintptr_t ca = ((uintptr_t)hi << 32) | lo;
        const uintptr_t al = ca & ~(uintptr_t)3;
        shift0[u] = (int32_t)(ca - al);
        const int32_t nw = (shift0[u] + clen[u] + 3) >> 2;
        const int32_t x = ch * 64 + (int32_t)lane;
        v[u] = x < nw ? reinterpret_cast<const uint32_t *>(al)[x] : 0u;
      }
#pragma unroll
      for (uint32_t u = 0; u < 8; u++) {
        const uint32_t c = c0 + u;
        const int32_t x = ch * 64 + (int32_t)lane;
#pragma unroll
        for (int32_t b = 0; b < 4; b++) {
          const int32_t k = x * 4 + b - shift0[u];
          if (k >= 0 && k < clen[u]) {
            const uint32_t chh = (v[u] >> (8 * b)) & 0xFFu;
            const uint32_t code = crev[u] ? ssw_code_complemented(chh) : ssw_code(chh);
            const uint32_t kk = (uint32_t)(crev[u] ? clen[u] - 1 - k : k);
            const uint32_t byte_addr = (kk >> 1) * NS + c;              // two 4-bit codes per byte
            atomicOr(reinterpret_cast<uint32_t *>(dst) + (byte_addr >> 2),
                     code << (8u * (byte_addr & 3u) + 4u * (kk & 1u)));
          }
        }
      }
    }
  }
}

// One candidate per lane.  LDS ([element][lane]): translated spans + the three row arrays, and,
// for the narrow bands nearly every candidate needs, the direction matrix itself: 5 bits per band
// cell, six cells per word, so the traceback -- a chain of dependent loads, one per step -- runs
// at LDS latency.  Wider bands keep the directions in a global slab per block ([cell][lane],
// coalesced, L2 resident), written fire-and-forget during the DP.
__global__ __launch_bounds__(64) void k_banded_lds(CigJob J, SwInputs in, SwParams p, LdsLayout Y) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  if (J.variant == 3) return;   // ablation: launch floor
  const uint32_t lane = threadIdx.x;
  const uint32_t NL = Y.nl, NS = NL + 1;
  const uint32_t li = blockIdx.x * NL + lane;
  const bool have = lane < NL && li < J.m;
  const uint32_t n_here = min(NL, J.m - blockIdx.x * NL);
  const uint32_t half = (((Y.lmax + 1) / 2) * NS + 3) & ~3u;   // bytes per packed sequence buffer
  uint8_t *SQ = lds_raw;
  uint8_t *SR = SQ + half;
  int16_t *S = reinterpret_cast<int16_t *>(lds_raw + (((size_t)2 * half + 15) & ~(size_t)15));
  uint32_t *DW = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(S) +
                                              (((size_t)3 * Y.W1 * NL * sizeof(int16_t) + 15) & ~(size_t)15));
  for (uint32_t x = lane; x < half / 2; x += 64) reinterpret_cast<uint32_t *>(lds_raw)[x] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  uint8_t *D = J.scratch + (uint64_t)blockIdx.x * J.wave_slab;

  uint32_t ci = 0;
  kslam_overlap o;
  memset(&o, 0, sizeof o);
  int32_t band_width = 1, refLen = 0, readLen = 0;
  bool skip = !have;
  const uint8_t *qsrc = in.read_bases, *rsrc = in.genome_bases;
  if (have) {
    ci = J.list[J.list_base + li];
    o = J.ov[ci];
    band_width = (int32_t)J.bw[ci];
    refLen = o.ref_end - o.ref_begin + 1;    // ssw.c:930-931
    readLen = o.query_end - o.query_begin + 1;
    // direction buffer growth check of the reference, ssw.c:631-642
    if ((int64_t)(band_width * 2 + 1) * readLen * 3 >= ((int64_t)1 << 30)) {
      o.score = 0;                                          // ssw.c:941-944
      o.cigar_len = 0;
      J.ov[ci] = o;
      J.bw[ci] = 0;
      skip = true;
    } else {
      const uint64_t ro = in.read_off[o.read];
      const uint64_t L = in.read_off[o.read + 1] - ro;
      const uint64_t go = in.genome_off[o.entry];
      const uint64_t G = in.genome_off[o.entry + 1] - go;
      const int64_t s0 = o.rel > 0 ? o.rel : 0;
      const int64_t wlen = (int64_t)min(L, G - (uint64_t)s0);
      qsrc = in.read_bases + ro + o.query_begin;
      // window position x of a flipped (revComp) window is genome position wlen - 1 - x
      rsrc = in.genome_bases + go + s0 + (o.revcomp ? (wlen - 1 - o.ref_end) : (int64_t)o.ref_begin);
    }
  }
  if (J.variant == 4) return;   // ablation: candidate header loads only
  if (J.variant == 5) { qsrc = in.read_bases + 64 * lane; rsrc = in.genome_bases + 64 * lane; }   // ablation: cache-resident sources
  // stage the two spans as SSW codes (ssw_cpp.cpp:11-23), cooperatively and coalesced
  stage_codes_wave(qsrc, skip ? 0 : readLen, false, SQ, NS, n_here);
  stage_codes_wave(rsrc, skip ? 0 : refLen, !skip && o.revcomp != 0, SR, NS, n_here);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (skip || J.variant == 2 || J.variant == 5) return;
  const int32_t score = o.score;
  struct Acc {
    int16_t *S; uint8_t *SQ, *SR, *D;   // row arrays as int16: |values| < 2^13 (14-bit score field)
    uint32_t NL, lane, W1, width_d;
    __device__ int16_t &hb(int32_t k) { return S[(uint32_t)k * NL + lane]; }
    __device__ int16_t &eb(int32_t k) { return S[(W1 + (uint32_t)k) * NL + lane]; }
    __device__ int16_t &hc(int32_t k) { return S[(2 * W1 + (uint32_t)k) * NL + lane]; }
    __device__ uint32_t q(int32_t i) { return (SQ[((uint32_t)i >> 1) * (NL + 1) + lane] >> (4 * (i & 1))) & 15u; }
    __device__ uint32_t r(int32_t j) { return (SR[((uint32_t)j >> 1) * (NL + 1) + lane] >> (4 * (j & 1))) & 15u; }
    uint32_t *DW, wpr, acc;             // LDS direction words: [row * wpr + word][lane]
    __device__ void set_dir(int32_t i, int32_t col, uint32_t v) {
      if (wpr) {
        const uint32_t w = (uint32_t)col / 6u, sh = 5u * ((uint32_t)col - 6u * w);
        acc = sh ? (acc | (v << sh)) : v;   // cells of a row arrive in column order: the word is complete
        DW[((uint32_t)i * wpr + w) * NL + lane] = acc;   // after its last cell, no read-modify-write needed
      } else
        D[((size_t)i * width_d + (uint32_t)col) * NL + lane] = (uint8_t)v;
    }
    __device__ uint32_t get_dir(int32_t i, int32_t col) {
      if (wpr) {
        const uint32_t w = (uint32_t)col / 6u, sh = 5u * ((uint32_t)col - 6u * w);
        return (DW[((uint32_t)i * wpr + w) * NL + lane] >> sh) & 31u;
      }
      return D[((size_t)i * width_d + (uint32_t)col) * NL + lane];
    }
  } A{S, SQ, SR, D, NL, lane, Y.W1, (uint32_t)(band_width * 2 + 1), DW, Y.wpr, 0u};
  (void)score;
  const int32_t mx = banded_attempt(A, refLen, readLen, band_width, p, J.bmax[ci]);
  if (J.variant == 1) return;   // ablation: no traceback
  J.bmax[ci] = mx;
  if (mx < score) {               // ssw.c:693-694: retry with twice the band
    J.bw[ci] = (uint32_t)band_width * 2u;
    return;
  }
  uint32_t *tmp = J.tmp + (uint64_t)(J.big ? (J.list_base + li) : ci) * J.cap;
  bool ovf = false;
  const int32_t l = banded_traceback(A, refLen, readLen, band_width, tmp, J.cap, &ovf);
  if (l < 0) {
    atomicAdd(&J.err[0], 1u);
    o.cigar_len = 0;
    J.ov[ci] = o;
    J.bw[ci] = 0;
    return;
  }
  if (ovf) {
    J.needbig[ci] = 1;  // rerun with a full-size temp slot
    return;
  }
  o.cigar_len = (uint32_t)l;
  J.ov[ci] = o;
  J.bw[ci] = 0;
  J.needbig[ci] = J.big ? 2 : 0;  // 2: ops live in the big temp area
}

__global__ void k_cigar_lens(const kslam_overlap *__restrict__ ov, uint64_t n, uint32_t *__restrict__ lens) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) lens[i] = ov[i].cigar_len;
}

// Un-flip + absolute coordinates (SmithWaterman.h:211-229), cigar gather.
// tmp holds the ops in TRACEBACK order; the reference reverses them once
// (ssw.c:773-784) and once more for revComp overlaps (SmithWaterman.h:212-216).
__global__ __launch_bounds__(256) void k_finalize(kslam_overlap *__restrict__ ov, uint64_t n, SwInputs in,
                                                  const uint64_t *__restrict__ cig_off,
                                                  const uint32_t *__restrict__ tmp, uint32_t cap,
                                                  const uint32_t *__restrict__ tmp_big, uint32_t cap_big,
                                                  const uint32_t *__restrict__ big_pos,
                                                  const uint8_t *__restrict__ needbig,
                                                  const uint32_t *__restrict__ bw,
                                                  uint32_t *__restrict__ pool, uint64_t pool_base, unsigned long long *cells) {
  __shared__ unsigned long long sm[4];
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long mycells = 0;
  if (i < n) {
    kslam_overlap o = ov[i];
    const uint64_t L = in.read_off[o.read + 1] - in.read_off[o.read];
    const uint64_t G = in.genome_off[o.entry + 1] - in.genome_off[o.entry];
    const int64_t s0 = o.rel > 0 ? o.rel : 0;
    const int64_t wlen = (int64_t)min(L, G - (uint64_t)s0);
    mycells = L * (unsigned long long)wlen;
    const uint32_t cl = o.cigar_len;
    if (cl && (bw[i] >> 31)) {
      pool[pool_base + cig_off[i]] = (bw[i] & 0x7FFFFFFFu) << 4;   // <n>M
    } else if (cl) {
      const uint32_t *src = (needbig[i] == 2) ? tmp_big + (uint64_t)big_pos[i] * cap_big
                                                          : tmp + i * (uint64_t)cap;
      uint32_t *dst = pool + pool_base + cig_off[i];
      if (o.revcomp) for (uint32_t k = 0; k < cl; k++) dst[k] = src[k];
      else for (uint32_t k = 0; k < cl; k++) dst[k] = src[cl - 1 - k];
    }
    o.cigar_off = cl ? pool_base + cig_off[i] : 0;
    if (o.revcomp) {
      const int32_t rb = o.ref_begin, qb = o.query_begin;
      o.ref_begin = (int32_t)(wlen - (int64_t)(o.ref_end + 1));
      o.ref_end = (int32_t)(wlen - (int64_t)(rb + 1));
      o.query_begin = (int32_t)((int64_t)L - (int64_t)(o.query_end + 1));
      o.query_end = (int32_t)((int64_t)L - (int64_t)(qb + 1));
    }
    o.ref_begin += (int32_t)s0;
    o.ref_end += (int32_t)s0;
    ov[i] = o;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mycells += __shfl_down(mycells, d, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mycells;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(cells, sm[0] + sm[1] + sm[2] + sm[3]);
}

}  // namespace

// ---------------------------------------------------------------------------
// host driver of the cigar stage
// ---------------------------------------------------------------------------
void cigar_prepare(CigarWork &W, uint64_t n, hipStream_t s) {
  if (n == 0) return;
  W.bmax.ensure(n * sizeof(int32_t));
  W.needbig.ensure(n);
  W.tmp.ensure(n * (uint64_t)CIG_CAP * sizeof(uint32_t));
  HIPCHK(hipMemsetAsync(W.bmax.p, 0, n * sizeof(int32_t), s));
  HIPCHK(hipMemsetAsync(W.needbig.p, 0, n, s));
}

void cigar_traceback(kslam_overlap *d_ov, uint64_t n, SwInputs in, SwParams p, uint32_t lmax, uint32_t *d_bw,
                     CigarWork &W, uint64_t *n_cigar_out, uint32_t *n_tb_err, hipStream_t s) {
  *n_cigar_out = 0;
  *n_tb_err = 0;
  if (n == 0) return;
  if (n >= (1ull << 32)) throw StatusError{KSLAM_ERR_UNSUPPORTED, ">= 2^32 candidates in one chunk"};
  const unsigned nb = (unsigned)((n + 255) / 256);
  W.flags.ensure(n * sizeof(uint32_t));
  W.pos.ensure(n * sizeof(uint32_t));
  W.list.ensure(n * sizeof(uint32_t));
  W.big_pos.ensure(n * sizeof(uint32_t));
  W.scan_tmp.ensure(scan_tmp_bytes(n));
  W.totals.ensure(4 * sizeof(uint64_t));
  W.cig_off.ensure(n * sizeof(uint64_t));
  W.tmp_big.ensure(256);
  uint64_t *d_tot = W.totals.as<uint64_t>();
  uint32_t *d_err = reinterpret_cast<uint32_t *>(d_tot + 2);
  HIPCHK(hipMemsetAsync(d_tot, 0, 4 * sizeof(uint64_t), s));
  const uint32_t cap_big = 2 * lmax + 4;
  if (p.report_cigar) {
    auto run_lists = [&](uint32_t cls, bool big) -> uint64_t {
      hipLaunchKernelGGL(k_class_flags, dim3(nb), dim3(256), 0, s, d_bw, W.needbig.as<uint8_t>(), n, cls,
                         big ? 1u : 0u, W.flags.as<uint32_t>());
      exclusive_scan_u32(W.flags.as<uint32_t>(), W.pos.as<uint32_t>(), n, d_tot, W.scan_tmp.p, s);
      uint64_t m = 0;
      HIPCHK(hipMemcpyAsync(&m, d_tot, sizeof m, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      if (m) hipLaunchKernelGGL(k_scatter_list, dim3(nb), dim3(256), 0, s, W.flags.as<uint32_t>(),
                                W.pos.as<uint32_t>(), n, W.list.as<uint32_t>());
      return m;
    };
    auto launch = [&](uint64_t m, uint32_t slot_bw, bool big) {
      LdsLayout Y;
      Y.lmax = lmax; Y.W1 = slot_bw * 2 + 4; Y.wd = slot_bw * 2 + 1;
      // Directions: global slab.  Keeping them in LDS (KSLAM_CIGAR_DIRS=lds) was measured: the DP is
      // LDS-instruction bound, not bound by the traceback's loads, and the extra 38 KB per block cost
      // more occupancy than the traceback gained (class 1: 7.9 ms against 2.8 ms).
      const size_t base_lane = (size_t)lmax + 2 + (size_t)3 * Y.W1 * sizeof(int16_t);
      const uint32_t wpr_fit = (Y.wd + 5) / 6;
      const char *force = getenv("KSLAM_CIGAR_DIRS");
      const bool dir_in_lds = force && force[0] == 'l' &&
                              (base_lane + (size_t)lmax * wpr_fit * 4) * 16 + 64 <= 64 * 1024;
      Y.wpr = dir_in_lds ? wpr_fit : 0;
      const size_t per_lane = base_lane + (size_t)lmax * Y.wpr * 4;
      uint32_t nl = 64;
      while (nl > 1 && per_lane * nl + 64 > 64 * 1024) nl >>= 1;   // <= 64 KB: at least two blocks per CU
      Y.nl = nl;
      const size_t half = ((((size_t)lmax + 1) / 2) * (nl + 1) + 3) & ~(size_t)3;
      const size_t rows_bytes = ((size_t)3 * Y.W1 * nl * sizeof(int16_t) + 15) & ~(size_t)15;
      const size_t lds = ((2 * half + 15) & ~(size_t)15) + rows_bytes + (size_t)lmax * Y.wpr * 4 * nl;
      if (lds > 160 * 1024) throw StatusError{KSLAM_ERR_UNSUPPORTED, "banded traceback band does not fit LDS"};
      if (lds > 64 * 1024)
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_banded_lds),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      uint64_t slab = Y.wpr ? 256 : (uint64_t)lmax * Y.wd * nl;
      slab = (slab + 255) & ~255ull;
      const uint64_t SCRATCH_BUDGET = 3ull << 30;
      const uint64_t blocks_per_launch = std::max<uint64_t>(1, SCRATCH_BUDGET / slab);
      const uint64_t n_blocks = (m + nl - 1) / nl;
      W.scratch.ensure(std::min<uint64_t>(n_blocks, blocks_per_launch) * slab);
      for (uint64_t b0 = 0; b0 < n_blocks; b0 += blocks_per_launch) {
        CigJob J;
        J.ov = d_ov; J.bw = d_bw; J.bmax = W.bmax.as<int32_t>(); J.needbig = W.needbig.as<uint8_t>();
        J.list = W.list.as<uint32_t>();
        const uint64_t nb_here = std::min<uint64_t>(blocks_per_launch, n_blocks - b0);
        J.list_base = (uint32_t)(b0 * nl);
        J.m = (uint32_t)std::min<uint64_t>(nb_here * nl, m - b0 * nl);
        J.slot_bw = slot_bw; J.lmax = lmax;
        J.cap = big ? cap_big : CIG_CAP;
        J.tmp = big ? W.tmp_big.as<uint32_t>() : W.tmp.as<uint32_t>();
        J.big = big ? 1 : 0;
        J.scratch = W.scratch.as<uint8_t>();
        J.wave_slab = slab;
        J.err = d_err;
        { const char *cv = getenv("KSLAM_CIGAR_VARIANT"); J.variant = cv ? (uint32_t)atoi(cv) : 0u; }
        hipLaunchKernelGGL(k_banded_lds, dim3((unsigned)nb_here), dim3(64), lds, s, J, in, p, Y);
      }
      HIPCHK(hipGetLastError());
    };
    // largest initial band class present
    hipLaunchKernelGGL(k_max_all, dim3(nb), dim3(256), 0, s, d_bw, n, reinterpret_cast<uint32_t *>(d_tot + 1));
    uint64_t mb0 = 0;
    HIPCHK(hipMemcpyAsync(&mb0, d_tot + 1, sizeof mb0, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    uint32_t last_cls = 0;
    while ((1ull << last_cls) < mb0) last_cls++;
    // failures move up exactly one class, so the sweep ends at the first empty class above last_cls
    for (uint32_t cls = 0; cls < 31 && mb0 > 0; cls++) {
      uint64_t m = run_lists(cls, false);
      if (m == 0) {
        if (cls > last_cls) break;
        continue;
      }
      if (getenv("KSLAM_DEBUG")) fprintf(stderr, "[kslam] cigar class %u (band <= %u): %llu candidates\n", cls, 1u << cls, (unsigned long long)m);
      launch(m, 1u << cls, false);
      if (cls >= last_cls) last_cls = cls + 1;
    }
    // candidates whose cigar did not fit the small temp slot: rerun with full-size slots
    uint64_t n_big = run_lists(0, true);
    if (n_big) {
      W.tmp_big.ensure(n_big * (uint64_t)cap_big * sizeof(uint32_t));
      HIPCHK(hipMemcpyAsync(W.big_pos.p, W.pos.p, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemsetAsync(d_tot + 1, 0, sizeof(uint64_t), s));
      hipLaunchKernelGGL(k_max_bw, dim3((unsigned)((n_big + 255) / 256)), dim3(256), 0, s, d_bw,
                         W.list.as<uint32_t>(), (uint32_t)n_big, reinterpret_cast<uint32_t *>(d_tot + 1));
      uint64_t mb = 0;
      HIPCHK(hipMemcpyAsync(&mb, d_tot + 1, sizeof mb, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      launch(n_big, (uint32_t)mb, true);
    }
  }
  // cigar pool layout
  hipLaunchKernelGGL(k_cigar_lens, dim3(nb), dim3(256), 0, s, d_ov, n, W.flags.as<uint32_t>());
  exclusive_scan_u32_to_u64(W.flags.as<uint32_t>(), W.cig_off.as<uint64_t>(), n, d_tot, W.scan_tmp.p, s);
  uint64_t host_tot[3] = {0, 0, 0};
  HIPCHK(hipMemcpyAsync(host_tot, d_tot, sizeof host_tot, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  *n_cigar_out = host_tot[0];
  *n_tb_err = (uint32_t)(host_tot[2] & 0xFFFFFFFFu);
}

void cigar_finalize(kslam_overlap *d_ov, uint64_t n, SwInputs in, uint32_t lmax, CigarWork &W,
                    const uint32_t *d_bw, uint32_t *d_pool, uint64_t pool_base, uint64_t *d_cells, hipStream_t s) {
  if (n == 0) return;
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_finalize, dim3(nb), dim3(256), 0, s, d_ov, n, in, W.cig_off.as<uint64_t>(),
                     W.tmp.as<uint32_t>(), CIG_CAP, W.tmp_big.as<uint32_t>(), 2 * lmax + 4,
                     W.big_pos.as<uint32_t>(), W.needbig.as<uint8_t>(), d_bw, d_pool, pool_base,
                     reinterpret_cast<unsigned long long *>(d_cells));
  HIPCHK(hipGetLastError());
}

}  // namespace kslam
