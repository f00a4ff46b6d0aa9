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

namespace kslam {

namespace {

__device__ inline uint32_t tr_base(uint32_t c) {  // ssw_cpp.cpp:11-23
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case 'U': case 'u': return 0;
    default: return 4;
  }
}
__device__ inline uint32_t comp_base(uint32_t c) {  // sequenceTools.h:98-116
  switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'T': return 'A';
    case 'G': return 'C';
    default: return c;
  }
}
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
};

__global__ __launch_bounds__(64) void k_banded(CigJob J, SwInputs in, SwParams p) {
  const uint32_t lane = threadIdx.x;
  const uint32_t li = blockIdx.x * 64 + lane;
  if (li >= J.m) return;
  const uint32_t ci = J.list[J.list_base + li];
  kslam_overlap o = J.ov[ci];
  int32_t band_width = (int32_t)J.bw[ci];
  const int32_t score = o.score;
  const int32_t refLen = o.ref_end - o.ref_begin + 1;    // ssw.c:930-931
  const int32_t readLen = o.query_end - o.query_begin + 1;
  // direction buffer growth check of the reference, ssw.c:631-642
  if ((int64_t)(band_width * 2 + 1) * readLen * 3 >= ((int64_t)1 << 30)) {
    o.score = 0;                                          // ssw.c:941-944
    o.cigar_len = 0;
    J.ov[ci] = o;
    J.bw[ci] = 0;
    return;
  }
  const uint64_t ro = in.read_off[o.read];
  const uint64_t L = in.read_off[o.read + 1] - ro;
  const uint64_t go = in.genome_off[o.entry];
  const uint64_t G = in.genome_off[o.entry + 1] - go;
  const int64_t s0 = o.rel > 0 ? o.rel : 0;
  const int64_t wlen = (int64_t)min(L, G - (uint64_t)s0);
  const uint8_t *rd = in.read_bases + ro + o.query_begin;
  const uint8_t *gw = in.genome_bases + go + s0;
  const int32_t rb = o.ref_begin;
  const bool rc = o.revcomp != 0;

  const int32_t width = band_width * 2 + 3, width_d = band_width * 2 + 1;
  const int32_t W1 = (int32_t)J.slot_bw * 2 + 4;  // row array length the slab was sized for
  uint8_t *slab = J.scratch + (uint64_t)blockIdx.x * J.wave_slab;
  int32_t *S = reinterpret_cast<int32_t *>(slab);
  uint8_t *D = slab + (uint64_t)3 * W1 * 64 * sizeof(int32_t);
#define HB(k) S[(uint32_t)(k) * 64u + lane]
#define EB(k) S[(uint32_t)(W1 + (k)) * 64u + lane]
#define HC(k) S[(uint32_t)(2 * W1 + (k)) * 64u + lane]
#define DIR(i, col) D[((uint64_t)(i) * (uint32_t)width_d + (uint32_t)(col)) * 64u + lane]
  int32_t mx = J.bmax[ci];
  for (int32_t k = 0; k <= width; k++) { HB(k) = 0; EB(k) = 0; HC(k) = 0; }
  for (int32_t i = 0; i < readLen; i++) {
    int32_t beg = 0, end = refLen - 1, u = 0, edge, f;
    int32_t j = i - band_width;
    beg = beg > j ? beg : j;
    j = i + band_width;
    end = end < j ? end : j;
    edge = end + 1 < width - 1 ? end + 1 : width - 1;      // ssw.c:654
    f = 0;
    HB(0) = 0; EB(0) = 0; HB(edge) = 0; EB(edge) = 0; HC(0) = 0;  // ssw.c:655
    const uint32_t qc = tr_base(rd[i]);
    const int32_t xi = i - band_width > 0 ? i - band_width : 0;
    const int32_t xim = i - 1 - band_width > 0 ? i - 1 - band_width : 0;
    for (j = beg; j <= end; j++) {
      const int32_t e = j - xim + 1;        // set_u(e, w, i-1, j)
      const int32_t b = j - 1 - xi + 1;     // set_u(b, w, i, j-1)
      const int32_t d = j - 1 - xim + 1;    // set_u(d, w, i-1, j-1)
      u = j - xi + 1;                       // set_u(u, w, i, j)
      const int32_t pos = rb + j;
      const uint32_t ch = rc ? comp_base(gw[wlen - 1 - pos]) : gw[pos];
      const uint32_t rcode = tr_base(ch);
      const int32_t sc = (qc > 3u || rcode > 3u) ? 0 : (qc == rcode ? p.match : -p.mismatch);
      int32_t t1 = i == 0 ? -p.gap_open : HB(e) - p.gap_open;     // ssw.c:668-671
      int32_t t2 = i == 0 ? -p.gap_extend : EB(e) - p.gap_extend;
      const int32_t ev = t1 > t2 ? t1 : t2;
      EB(u) = ev;
      const uint32_t de = t1 > t2 ? 3u : 2u;
      t1 = HC(b) - p.gap_open;                                     // ssw.c:673-676
      t2 = f - p.gap_extend;
      f = t1 > t2 ? t1 : t2;
      const uint32_t df = t1 > t2 ? 5u : 4u;
      const int32_t e1 = ev > 0 ? ev : 0, f1 = f > 0 ? f : 0;      // ssw.c:678-682
      t1 = e1 > f1 ? e1 : f1;
      t2 = HB(d) + sc;
      const int32_t hv = t1 > t2 ? t1 : t2;
      HC(u) = hv;
      if (hv > mx) mx = hv;                                        // ssw.c:684
      const uint32_t dh = t1 <= t2 ? 1u : (e1 > f1 ? de : df);     // ssw.c:686-690
      DIR(i, j - xi) = (uint8_t)((de - 2u) | ((df - 4u) << 1) | (dh << 2));
    }
    for (j = 1; j <= u; j++) HB(j) = HC(j);                        // ssw.c:692
  }
  J.bmax[ci] = mx;
  if (mx < score) {               // ssw.c:693-694: retry with twice the band
    J.bw[ci] = (uint32_t)band_width * 2u;
    return;
  }
  // traceback, ssw.c:698-771
  int32_t i = readLen - 1, j = refLen - 1, cnt = 0, l = 0, op = 0, cur = 0, plane = 2;
  uint32_t *tmp = J.tmp + (uint64_t)(J.big ? (J.list_base + li) : ci) * J.cap;
  bool bad = false, ovf = false;
  while (i > 0) {
    const int32_t xi = i - band_width > 0 ? i - band_width : 0;
    const int32_t col = j - xi;
    const int32_t jend = (refLen - 1) < (i + band_width) ? (refLen - 1) : (i + band_width);
    uint32_t dir = 0;
    if (col >= 0 && j <= jend && j >= 0) {
      const uint32_t bb = DIR(i, col);
      dir = plane == 2 ? ((bb >> 2) & 7u) : (plane == 0 ? 2u + (bb & 1u) : 4u + ((bb >> 1) & 1u));
    }
    switch (dir) {
      case 1: --i; --j; plane = 2; op = 0; break;
      case 2: --i; plane = 0; op = 1; break;
      case 3: --i; plane = 2; op = 1; break;
      case 4: --j; plane = 1; op = 2; break;
      case 5: --j; plane = 2; op = 2; break;
      default: bad = true; break;
    }
    if (bad) break;
    if (op == cur) ++cnt;
    else {
      if ((uint32_t)l < J.cap) tmp[l] = (uint32_t)cnt << 4 | (uint32_t)cur; else ovf = true;
      ++l;
      cur = op;
      cnt = 1;
    }
  }
  if (bad) {
    atomicAdd(&J.err[0], 1u);
    o.cigar_len = 0;
    J.ov[ci] = o;
    J.bw[ci] = 0;
    return;
  }
  if (op == 0) {                                                    // ssw.c:754-761
    if ((uint32_t)l < J.cap) tmp[l] = (uint32_t)(cnt + 1) << 4; else ovf = true;
    ++l;
  } else {
    if ((uint32_t)l + 1 < J.cap) { tmp[l] = (uint32_t)cnt << 4 | (uint32_t)op; tmp[l + 1] = 16u; } else ovf = true;
    l += 2;
  }
  if (ovf) {
    J.needbig[ci] = 1;  // rerun with a full-size temp slot
    return;
  }
  o.cigar_len = (uint32_t)l;
  J.ov[ci] = o;
  J.bw[ci] = 0;
  J.needbig[ci] = J.big ? 2 : 0;  // 2: ops live in the big temp area
#undef HB
#undef EB
#undef HC
#undef DIR
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
namespace {
constexpr uint32_t CIG_CAP = 24;  // ops per small temp slot
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
  W.bmax.ensure(n * sizeof(int32_t));
  W.needbig.ensure(n);
  W.big_pos.ensure(n * sizeof(uint32_t));
  W.scan_tmp.ensure(scan_tmp_bytes(n));
  W.totals.ensure(4 * sizeof(uint64_t));
  W.cig_off.ensure(n * sizeof(uint64_t));
  W.tmp.ensure(n * (uint64_t)CIG_CAP * sizeof(uint32_t));
  W.tmp_big.ensure(256);
  uint64_t *d_tot = W.totals.as<uint64_t>();
  uint32_t *d_err = reinterpret_cast<uint32_t *>(d_tot + 2);
  HIPCHK(hipMemsetAsync(W.bmax.p, 0, n * sizeof(int32_t), s));
  HIPCHK(hipMemsetAsync(W.needbig.p, 0, n, s));
  HIPCHK(hipMemsetAsync(d_tot, 0, 4 * sizeof(uint64_t), s));
  const uint32_t cap_big = 2 * lmax + 4;
  if (p.report_cigar) {
    const uint64_t SCRATCH_BUDGET = 1ull << 31;
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
      const uint64_t W1 = (uint64_t)slot_bw * 2 + 4;
      uint64_t slab = 64ull * (3 * W1 * sizeof(int32_t) + (uint64_t)(2 * slot_bw + 1) * lmax);
      slab = (slab + 255) & ~255ull;
      const uint64_t waves_per_launch = std::max<uint64_t>(1, SCRATCH_BUDGET / slab);
      const uint64_t per_launch = waves_per_launch * 64;
      W.scratch.ensure(std::min<uint64_t>((m + 63) / 64, waves_per_launch) * slab);
      for (uint64_t base = 0; base < m; base += per_launch) {
        CigJob J;
        J.ov = d_ov; J.bw = d_bw; J.bmax = W.bmax.as<int32_t>(); J.needbig = W.needbig.as<uint8_t>();
        J.list = W.list.as<uint32_t>();
        J.m = (uint32_t)std::min<uint64_t>(per_launch, m - base);
        J.list_base = (uint32_t)base;
        J.slot_bw = slot_bw; J.lmax = lmax;
        J.cap = big ? cap_big : CIG_CAP;
        J.tmp = big ? W.tmp_big.as<uint32_t>() : W.tmp.as<uint32_t>();
        J.big = big ? 1 : 0;
        J.scratch = W.scratch.as<uint8_t>();
        J.wave_slab = slab;
        J.err = d_err;
        hipLaunchKernelGGL(k_banded, dim3((J.m + 63) / 64), dim3(64), 0, s, J, in, p);
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
