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
// by band width (cig_bin below), a bin's launch runs one attempt for each of its
// candidates and a failed one moves to the bin of its doubled band.
//
// MI355X design: three implementations of one attempt that leave the same direction
// bits -- the band in registers, one candidate per lane (bands up to 4 / 7); a candidate
// over 8 or 16 lanes swept by anti-diagonals like the scoring kernels (bands up to 127);
// the literal row arrays in LDS, one candidate per lane (the rest).  Direction bits go to a
// slab in global memory packed along band diagonals, six cells per word, so that the
// traceback -- a chain of dependent loads -- takes a whole run of matches per word.
#include "common.h"
#include "banded_core.h"
#include "stage.h"

namespace kslam {

namespace {

// Bins of the band widths.  banded_sw starts at |refLen - readLen| + 1 and doubles (ssw.c:616, 693-694), so the widths a batch
// asks for are 1 + the indel lengths -- NOT powers of two -- and a kernel that gives every candidate of a bin the slots of the
// bin's widest member should not have "one more than a power of two" as that member: 2 bw + 1 diagonals for bw = 8 are 17,
// one too many for 16 slots.  The bins therefore end at 2^k - 1 (2 bw + 1 <= 2^(k+1) - 1 slots), except the two narrow ones,
// which are exact because nearly every gapped read of a batch is a one-base indel (bw = 2):
//   0: 1   1: 2   2: 3..4   3: 5..7   4: 8..15   5: 16..31   6: 32..63   7: >= 64 (the generic loop, by floor(log2 bw))
// A failed attempt doubles bw, which always lands in a LATER bin.
__host__ __device__ inline uint32_t cig_bin(uint32_t bw) {
  return bw <= 2 ? bw - 1 : (bw <= 4 ? 2u : (bw <= 7 ? 3u : (bw <= 15 ? 4u : (bw <= 31 ? 5u : (bw <= 63 ? 6u : 7u)))));
}
constexpr uint32_t CIG_BIN_MAX_BW[7] = {1, 2, 4, 7, 15, 31, 63};
__device__ inline uint32_t band_class(uint32_t bw) {   // floor(log2 bw): the classes of the generic loop (bw >= 64)
  return 31u - (uint32_t)__builtin_clz(bw | 1u);
}

__global__ void k_class_flags(const uint32_t *__restrict__ bw, const uint8_t *__restrict__ needbig, uint64_t n,
                              uint32_t cls, uint32_t big, uint32_t *__restrict__ flags) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t b = bw[i];
  uint32_t f = 0;
  if (b != 0 && !(b >> 31)) {   // bit 31: inline <n>M set by the SW kernel
    if (big == 1) f = needbig[i] ? 1u : 0u;
    else if (big == 2) f = (needbig[i] == 3 && band_class(b) == cls) ? 1u : 0u;   // sent back by the systolic kernel
    else f = (!needbig[i] && band_class(b) == cls) ? 1u : 0u;
  }
  flags[i] = f;
}

// which classes (floor(log2 bw)) still have unfinished candidates: one bit each
__global__ void k_class_mask(const uint32_t *__restrict__ bw, const uint8_t *__restrict__ needbig, uint64_t n, uint32_t *__restrict__ mask) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t v = 0;
  if (i < n) {
    const uint32_t b = bw[i];
    if (b != 0 && !(b >> 31) && !needbig[i]) v = 1u << band_class(b);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v |= (uint32_t)__shfl_down((int)v, d, 64);
  if ((threadIdx.x & 63) == 0 && v) atomicOr(mask, v);
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

struct CigJob {
  kslam_overlap *ov;
  uint32_t *bw;
  int32_t *bmax;
  uint8_t *needbig;
  const uint32_t *list;
  uint32_t m;           // list entries in this launch (with m_dev: the launch's CAPACITY)
  const uint32_t *m_dev = nullptr;   // the list's length on the device, read when the kernel runs: this launch covers entries
                                     // [list_base, min(list_base + m, *m_dev)) -- a bin's list is still growing when the host sizes its launch
  uint32_t m_sure = 0;               // entries of this launch the host KNEW to exist: a workgroup inside them does not wait for *m_dev
  uint32_t list_base;   // first list entry of this launch
  uint32_t slot_bw;     // band width the scratch slab is sized for
  uint32_t lmax;        // max read length (row count bound)
  uint32_t cap;         // cigar ops per temp slot
  uint32_t *tmp;        // temp cigar slots: normal: [candidate][cap]; big: [list pos][cap]
  uint32_t big;
  uint8_t *scratch;
  uint64_t wave_slab;   // scratch bytes per wavefront
  uint32_t *err;        // [0] traceback errors
#ifdef KSLAM_ABLATE
  uint32_t variant;     // timing ablations (KSLAM_CIGAR_VARIANT): measurement-only build
#else
  static constexpr uint32_t variant = 0;   // compiled out of the product build
#endif
  // where candidates go that are not finished by this launch (nullptr: the host re-lists by flags)
  uint32_t *bin_list[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // band doubled: the list of its new bin
  uint32_t *bin_count = nullptr;                                // [8] (nullptr: the host re-lists by flags)
  uint32_t *special_list = nullptr, *special_count = nullptr;   // handed back by the systolic kernel
  uint32_t *big_count = nullptr;                                // cigar longer than the small temp slot
  // systolic kernels: which list positions' attempts reached the score, for k_systolic_traceback (nullptr: the
  // group's lane 0 walks the traceback itself, at the end of the DP kernel)
  uint8_t *tb_flag = nullptr;   // [m], zeroed: 1 = this list position's attempt reached the score
};

// list entries this launch really has (CigJob::m_dev)
// (hi: one past the last entry the calling workgroup looks at)
__device__ inline uint32_t live_entries(const CigJob &J, uint32_t hi) {
  if (!J.m_dev || hi <= J.m_sure) return J.m;
  const uint32_t tot = *J.m_dev;
  return min(J.m, tot > J.list_base ? tot - J.list_base : 0u);
}

// wave-aggregated append of candidate ci to a list (one atomic per wave)
__device__ inline void append_candidate(bool want, uint32_t ci, uint32_t *list, uint32_t *count) {
  const uint64_t m = __ballot(want);
  if (!m) return;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t leader = (uint32_t)__builtin_ctzll(m);
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
  base = __shfl((int)base, (int)leader, 64);
  if (want) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = ci;
}

// the same for a candidate whose band doubled: into the list of its new bin (lanes of a wave may differ in that)
__device__ inline void append_doubled(const CigJob &J, bool want, uint32_t ci, uint32_t new_bw) {
  if (!J.bin_count) return;
  const uint32_t b = cig_bin(new_bw);
  for (;;) {
    const uint64_t m = __ballot(want);
    if (!m) break;
    const uint32_t b0 = (uint32_t)__shfl((int)b, (int)__builtin_ctzll(m), 64);
    append_candidate(want && b == b0, ci, J.bin_list[b0], J.bin_count + b0);
    want = want && b != b0;
  }
}

__global__ void k_cig_class(const uint32_t *__restrict__ bw, uint64_t n, uint8_t *__restrict__ cls) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t b = bw[i];
  cls[i] = (b != 0 && !(b >> 31)) ? (uint8_t)cig_bin(b) : (uint8_t)255;
}

// banded_sw (ssw.c:594-792): one attempt with J.bw[ci], then traceback when max >= score.
// One candidate per lane; everything the DP touches lives in LDS laid out [element][lane]
// (NL lanes per block, fewer for wide bands): translated read / reference spans, the three
// row arrays, one direction byte per band cell.  The serial chain (f and the left H) runs in
// registers; the previous-row values and the reference code of the next cell are fetched one
// iteration ahead so LDS latency hides behind the chain.
struct LdsLayout {
  uint32_t nl;       // lanes (candidates) per block
  uint32_t lmax;     // rows / columns the sequence buffers are sized for
  uint32_t nch;      // 16-base chunks a sequence buffer holds per lane (stage_codes_wave)
  uint32_t W1;       // row array length (2 * slot_bw + 4)
  uint32_t wd;       // direction cells per row the global slab is sized for (2 * slot_bw + 1)
  uint32_t wpr;      // direction words per row kept in LDS (six 5-bit cells per word); 0: global slab
};

// Staging for the one-candidate-per-lane kernels, by the whole wave.  (It used to be lane-private: every lane walked
// its own two spans, 16 bytes per load, and a load instruction of the wave touched 64 different cache lines of which the
// L1 kept next to none until the lane came back for the next 16 bytes -- by ablation 0.56 of the 1.3 ms of the bw = 2
// launch.)  Here eight lanes serve one candidate at a time, as stage_span does for the systolic kernels: the span is
// fetched as the 16-byte ALIGNED chunks that cover it, a chunk per lane and a whole line per eight lanes, converted in
// registers to two 4-bit codes per byte (pre-encoded bases, see encode_bases: bits 0-2 SSW code, bit 3 complementable)
// and stored with one 8-byte LDS write at the chunk's own place in the candidate's column: granule (chunk, lane) of dst
// is the 8 bytes at ((chunk * NL + lane) * 8).  The buffer thus holds the codes of the covering chunks, not re-aligned,
// and the caller gets the position of its element 0.  `rev` (window of a revComp overlap): complemented, the chunks in
// reverse order with their bytes reversed, which lands the reversed span at a (different) position of the same column.
// Both base arrays are the library's own copies: 256-byte aligned starts, 64 bytes of slack at the end.
__device__ inline uint32_t pack_nibbles(uint32_t c) {   // four codes, one per byte -> 16 bits
  const uint32_t t = (c | (c >> 4)) & 0x00FF00FFu;
  return (t | (t >> 8)) & 0xFFFFu;
}
__device__ inline int32_t stage_codes_wave(const uint8_t *codes, int32_t len, bool rev, uint8_t *dst, uint32_t NL,
                                           uint32_t lane) {
  const uint32_t shift = (uint32_t)(reinterpret_cast<uintptr_t>(codes) & 15u);
  const int32_t nch = len > 0 ? ((int32_t)shift + len + 15) >> 4 : 0;
  const uint64_t base = reinterpret_cast<uintptr_t>(codes) - shift;
  const uint32_t t = lane & 7u, g = lane >> 3;
#pragma unroll 2
  for (uint32_t r = 0; r < 8; r++) {
    const int c = (int)(8u * r + g);   // the candidate (lane) this group of eight serves in round r
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)base, c, 64), hi = (uint32_t)__shfl((int)(uint32_t)(base >> 32), c, 64);
    const int32_t n = __shfl(nch, c, 64);
    const bool rv = __shfl((int)rev, c, 64) != 0;
    const uint4 *b = reinterpret_cast<const uint4 *>(((uint64_t)hi << 32) | lo);
    for (int32_t k = (int32_t)t; k < n; k += 8) {
      const uint4 v = b[k];
      uint2 w;
      int32_t at;
      if (!rv) {
        w.x = pack_nibbles(codes_of_dword<1>(v.x, false)) | (pack_nibbles(codes_of_dword<1>(v.y, false)) << 16);
        w.y = pack_nibbles(codes_of_dword<1>(v.z, false)) | (pack_nibbles(codes_of_dword<1>(v.w, false)) << 16);
        at = k;
      } else {
        w.x = pack_nibbles(__builtin_bswap32(codes_of_dword<1>(v.w, true))) | (pack_nibbles(__builtin_bswap32(codes_of_dword<1>(v.z, true))) << 16);
        w.y = pack_nibbles(__builtin_bswap32(codes_of_dword<1>(v.y, true))) | (pack_nibbles(__builtin_bswap32(codes_of_dword<1>(v.x, true))) << 16);
        at = n - 1 - k;
      }
      *reinterpret_cast<uint2 *>(dst + ((uint32_t)at * NL + (uint32_t)c) * 8u) = w;
    }
  }
  return rev ? 16 * nch - (int32_t)shift - len : (int32_t)shift;
}

// ---- narrow bands: the band lives in registers ------------------------------------------------------
// One attempt of banded_sw (ssw.c:645-693) for a lane whose band width is at most BW.  Slot x of the
// band is diagonal j - i = x - bw; H and E of the previous row sit in registers indexed by slot
// (statically, the slot loop is unrolled), a row is swept left to right with the F chain and the left
// H in registers, and the five direction bits of its cells leave as one or two words.  Cells that do
// not exist -- slots beyond 2 bw, columns outside [0, refLen) -- hold H = E = 0 and pass H = F = 0 to
// the right, which is what the reference's zeroed rows and its sentinels present (see the systolic
// kernel below for the same argument).  Caller guarantees refLen > 2 bw + 1.
template <int BW, class Seq>
__device__ inline int32_t banded_attempt_reg(Seq &m, uint32_t *DW, uint32_t NL, uint32_t lane, int32_t refLen,
                                             int32_t readLen, int32_t bw, const SwParams &p, int32_t mx) {
  constexpr int NX = 2 * BW + 1;
  int32_t H[NX + 1], E[NX + 1];
  // direction words packed along a diagonal: word (i / 6, slot) holds rows 6 (i / 6) .. + 5 of the slot,
  // so that the traceback's diagonal steps stay inside the word it already fetched
  uint32_t dwx[NX];
#pragma unroll
  for (int x = 0; x < NX; x++) dwx[x] = 0;
  uint32_t R[NX];
#pragma unroll
  for (int x = 0; x <= NX; x++) { H[x] = 0; E[x] = 0; }
#pragma unroll
  for (int x = 0; x < NX; x++) {
    const int32_t j = x - bw;
    R[x] = (uint32_t)j < (uint32_t)refLen ? 6u * m.r(j) : 24u;   // window code x 6: the field of a packed score row
  }
  const int32_t live = 2 * bw;   // last live slot
  const int32_t gO = in_vgpr(p.gap_open), gE = in_vgpr(p.gap_extend);
  for (int32_t i = 0; i < readLen; i++) {
    const uint32_t srow = score_row(m.q(i), p, 0);   // the read base against the five window codes, 6 bits each
    int32_t hleft = 0, f = 0;
    const uint32_t i6 = (uint32_t)i / 6u, sh = 5u * ((uint32_t)i - 6u * i6);
    // Interior rows -- every slot of the register band is a real cell: band as wide as the template's, row far enough
    // from both ends of the window -- need none of the "does this cell exist" masking, and they are nearly all rows
    // (all but the first and last BW of ~150).  The choice is made per WAVE (no lane still running may be at an edge
    // row), so the lanes never split over the two bodies.
    const bool interior = bw == BW && i >= BW && i + BW < refLen;
    if (__ballot(!interior) == 0) {
#pragma unroll
      for (int x = 0; x < NX; x++) {
        const int32_t sc = __builtin_amdgcn_sbfe(srow, R[x], 6);
        int32_t t1 = H[x + 1] - gO, t2 = E[x + 1] - gE;                       // ssw.c:668-671
        const int32_t ev = max(t1, t2);
        const uint32_t de = (uint32_t)(t2 - t1) >> 31;
        t1 = hleft - gO;                                                     // ssw.c:673-676
        t2 = f - gE;
        const int32_t fv = max(t1, t2);
        const uint32_t df = (uint32_t)(t2 - t1) >> 31;
        const int32_t e1 = max(ev, 0), f1 = max(fv, 0);                      // ssw.c:678-682
        const int32_t m1 = max(e1, f1), dg = H[x] + sc;
        const int32_t hv = max(m1, dg);
        const uint32_t dh = m1 <= dg ? 1u : (e1 > f1 ? 2u + de : 4u + df);   // ssw.c:686-690
        H[x] = hv;
        E[x] = ev;
        hleft = hv;
        f = fv;
        mx = max(mx, hv);
        dwx[x] |= (de | (df << 1) | (dh << 2)) << sh;
      }
    } else {
#pragma unroll
    for (int x = 0; x < NX; x++) {
      const int32_t j = i + x - bw;
      const bool valid = x <= live && (uint32_t)j < (uint32_t)refLen;
      const int32_t vm = valid ? -1 : 0;   // one select, then masks: a cell that does not exist leaves zeros
      const int32_t sc = __builtin_amdgcn_sbfe(srow, R[x], 6);
      int32_t t1 = H[x + 1] - gO, t2 = E[x + 1] - gE;                       // ssw.c:668-671
      const int32_t ev = max(t1, t2);
      const uint32_t de = (uint32_t)(t2 - t1) >> 31;                       // t1 > t2 (values are small: no overflow)
      t1 = hleft - gO;                                                     // ssw.c:673-676
      t2 = f - gE;
      const int32_t fv = max(t1, t2);
      const uint32_t df = (uint32_t)(t2 - t1) >> 31;
      const int32_t e1 = max(ev, 0), f1 = max(fv, 0);                      // ssw.c:678-682
      const int32_t m1 = max(e1, f1), dg = H[x] + sc;
      const int32_t hv = max(m1, dg);
      const uint32_t dh = m1 <= dg ? 1u : (e1 > f1 ? 2u + de : 4u + df);   // ssw.c:686-690
      H[x] = hv & vm;
      E[x] = ev & vm;
      hleft = H[x];
      f = fv & vm;
      mx = max(mx, H[x]);                                                  // ssw.c:684 (hv >= 0)
      dwx[x] |= ((de | (df << 1) | (dh << 2)) & (uint32_t)vm) << sh;
    }
    }
    if (sh == 25u || i == readLen - 1) {
#pragma unroll
      for (int x = 0; x < NX; x++) {
        DW[(i6 * NX + x) * NL + lane] = dwx[x];
        dwx[x] = 0;
      }
    }
#pragma unroll
    for (int x = 0; x + 1 < NX; x++) R[x] = R[x + 1];
    const int32_t jn = i + 1 + (NX - 1) - bw;
    R[NX - 1] = (uint32_t)jn < (uint32_t)refLen ? 6u * m.r(jn) : 24u;
  }
  return mx;
}

// One candidate per lane.  LDS ([element][lane]): translated spans + the three row arrays, and,
// for the narrow bands nearly every candidate needs, the direction matrix itself: 5 bits per band
// cell, six cells per word, so the traceback -- a chain of dependent loads, one per step -- runs
// at LDS latency.  Wider bands keep the directions in a global slab per block ([cell][lane],
// coalesced, L2 resident), written fire-and-forget during the DP.
template <int REG_BW>   // 0: rows in LDS, any band; 1 / 2 / 4: bands up to that width in registers
__global__ __launch_bounds__(64) void k_banded_lds(CigJob J, SwInputs in, SwParams p, LdsLayout Y) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  if (J.variant == 3) return;   // ablation: launch floor
  const uint32_t lane = threadIdx.x;
  const uint32_t NL = Y.nl;
  const uint32_t li = blockIdx.x * NL + lane;
  const bool have = lane < NL && li < live_entries(J, (blockIdx.x + 1) * NL);
  const uint32_t half = Y.nch * NL * 8u;   // bytes per packed sequence buffer
  uint8_t *SQ = lds_raw;
  uint8_t *SR = SQ + half;
  int16_t *S = reinterpret_cast<int16_t *>(lds_raw + (size_t)2 * half);
  uint32_t *DW = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(S) +
                                              (((size_t)3 * Y.W1 * NL * sizeof(int16_t) + 15) & ~(size_t)15));
  uint8_t *D = J.scratch + (uint64_t)blockIdx.x * J.wave_slab;

  uint32_t ci = 0;
  kslam_overlap o;
  memset(&o, 0, sizeof o);
  int32_t band_width = 1, refLen = 0, readLen = 0;
  bool skip = !have;
  const uint8_t *qsrc = in.read_codes, *rsrc = in.genome_codes;
  if (have) {
    ci = J.list[J.list_base + li];
    o = J.ov[ci];
    if (J.needbig[ci] == 3) J.needbig[ci] = 0;   // handed over by the systolic kernel
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
      qsrc = in.read_codes + ro + o.query_begin;
      // window position x of a flipped (revComp) window is genome position wlen - 1 - x
      rsrc = in.genome_codes + go + s0 + (o.revcomp ? (wlen - 1 - o.ref_end) : (int64_t)o.ref_begin);
    }
  }
  if constexpr (REG_BW > 0) {
    // a window no longer than its band does not fit the register sweep (banded_attempt_reg wants refLen > 2 bw + 1), and this
    // instantiation keeps no row arrays: to the one-lane kernel (the whole wave takes part in the append)
    const bool hand = !skip && !(band_width <= REG_BW && refLen > 2 * band_width + 1);
    if (hand) J.needbig[ci] = 3;
    append_candidate(hand, ci, J.special_list, J.special_count);
    skip = skip || hand;
  }
  if (J.variant == 4) return;   // ablation: candidate header loads only
  // stage the two spans as SSW codes (ssw_cpp.cpp:11-23): the wave together, eight lanes per candidate
  const int32_t oq = stage_codes_wave(qsrc, skip ? 0 : readLen, false, SQ, NL, lane);
  const int32_t orf = stage_codes_wave(rsrc, skip ? 0 : refLen, !skip && o.revcomp != 0, SR, NL, lane);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (skip || J.variant == 2) return;
  const int32_t score = o.score;
  struct Acc {
    int16_t *S; uint8_t *SQ, *SR, *D;   // row arrays as int16: |values| < 2^13 (14-bit score field)
    uint32_t NL, lane, W1, width_d;
    __device__ int16_t &hb(int32_t k) { return S[(uint32_t)k * NL + lane]; }
    __device__ int16_t &eb(int32_t k) { return S[(W1 + (uint32_t)k) * NL + lane]; }
    __device__ int16_t &hc(int32_t k) { return S[(2 * W1 + (uint32_t)k) * NL + lane]; }
    int32_t oq, orf;                    // where element 0 of each span sits in its column (stage_codes_wave)
    __device__ static uint32_t code(const uint8_t *B, uint32_t s, uint32_t NL, uint32_t lane) {
      return ((uint32_t)B[((s >> 4) * NL + lane) * 8u + ((s >> 1) & 7u)] >> (4u * (s & 1u))) & 15u;
    }
    __device__ uint32_t q(int32_t i) { return code(SQ, (uint32_t)(i + oq), NL, lane); }
    __device__ uint32_t r(int32_t j) { return code(SR, (uint32_t)(j + orf), NL, lane); }
    uint32_t *DW, wpr, acc;             // LDS direction words: [row * wpr + word][lane]
    __device__ void set_dir(int32_t i, int32_t col, uint32_t v) {
      if (wpr) {
        const uint32_t w = (uint32_t)col / 6u, sh = 5u * ((uint32_t)col - 6u * w);
        acc = sh ? (acc | (v << sh)) : v;   // cells of a row arrive in column order: the word is complete
        DW[((uint32_t)i * wpr + w) * NL + lane] = acc;   // after its last cell, no read-modify-write needed
      } else
        D[((size_t)i * width_d + (uint32_t)col) * NL + lane] = (uint8_t)v;
    }
    int32_t reg_bw = 0;                 // > 0: the words hold band slots (banded_attempt_reg), not columns
    uint32_t nx = 0, have_idx = 0xFFFFFFFFu, have_word = 0;   // register variant: slots per row; the word fetched last
    __device__ int32_t diag_run(int32_t i, int32_t j) {   // (banded_core.h) only the register variant packs along diagonals
      if (reg_bw <= 0) return 0;
      const uint32_t i6 = (uint32_t)i / 6u, r = (uint32_t)i - 6u * i6;
      const uint32_t idx = (i6 * nx + (uint32_t)(j - i + reg_bw)) * NL + lane;
      if (idx != have_idx) {
        have_idx = idx;
        have_word = DW[idx];
      }
      return diag_cells_down_from(have_word, r);
    }
    __device__ uint32_t get_dir(int32_t i, int32_t col) {
      if (reg_bw > 0) {                 // words (i / 6, slot): six rows of one band slot
        if (i < reg_bw) col += reg_bw - i;   // slot = j - i + bw, col = j - max(0, i - bw)
        const uint32_t i6 = (uint32_t)i / 6u, r = (uint32_t)i - 6u * i6;
        const uint32_t idx = (i6 * nx + (uint32_t)col) * NL + lane;
        if (idx != have_idx) {
          have_idx = idx;
          have_word = DW[idx];
        }
        return (have_word >> (5u * r)) & 31u;
      }
      if (wpr) {
        if (reg_bw > 0 && i < reg_bw) col += reg_bw - i;   // slot = j - i + bw, col = j - max(0, i - bw)
        const uint32_t w = (uint32_t)col / 6u, sh = 5u * ((uint32_t)col - 6u * w);
        return (DW[((uint32_t)i * wpr + w) * NL + lane] >> sh) & 31u;
      }
      return D[((size_t)i * width_d + (uint32_t)col) * NL + lane];
    }
  } A{S, SQ, SR, D, NL, lane, Y.W1, (uint32_t)(band_width * 2 + 1), oq, orf, DW, Y.wpr, 0u};
  (void)score;
  int32_t mx;
  if constexpr (REG_BW > 0) {
    // direction words in the global slab, [row * WPR + word][lane]; slot x of row i
    constexpr uint32_t WPR = (2 * (REG_BW > 0 ? REG_BW : 1) + 1 + 5) / 6;
    mx = banded_attempt_reg<(REG_BW > 0 ? REG_BW : 1)>(A, reinterpret_cast<uint32_t *>(D), NL, lane, refLen,
                                                        readLen, band_width, p, J.bmax[ci]);
    A.DW = reinterpret_cast<uint32_t *>(D);
    A.wpr = WPR;
    A.reg_bw = band_width;
    A.nx = 2 * (REG_BW > 0 ? REG_BW : 1) + 1;
  } else {
    mx = banded_attempt(A, refLen, readLen, band_width, p, J.bmax[ci]);
  }
  if (J.variant == 1) return;   // ablation: no traceback
  J.bmax[ci] = mx;
  if (mx < score) {               // ssw.c:693-694: retry with twice the band
    J.bw[ci] = (uint32_t)band_width * 2u;
    append_doubled(J, true, ci, (uint32_t)band_width * 2u);
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
    if (J.big_count) atomicAdd(J.big_count, 1u);
    return;
  }
  o.cigar_len = (uint32_t)l;
  J.ov[ci] = o;
  J.bw[ci] = 0;
  J.needbig[ci] = J.big ? 2 : 0;  // 2: ops live in the big temp area
}

// traceback of one candidate from the direction words a systolic attempt left in its slab
template <int GL, int DPL>
__device__ inline void systolic_traceback(const CigJob &J, uint32_t li, uint32_t ci, kslam_overlap o, int32_t bw,
                                          int32_t refLen, int32_t readLen, const uint32_t *D) {
  const int32_t k0 = (bw & 1) ? -1 : 0;
  struct Acc {
    const uint32_t *D;
    int32_t bw, k0;
    uint32_t have_idx, have_word;   // the word fetched last: a diagonal run reuses it
    __device__ int32_t diag_run(int32_t i, int32_t j) {   // (banded_core.h)
      const int32_t x = j - i + bw;
      const int32_t tt = x / DPL, q = x - tt * DPL;
      const uint32_t n = (uint32_t)((i + j - k0 - (q & 1)) >> 1);
      const uint32_t m = n / 6u, r = n - 6u * m;
      const uint32_t idx = (m * GL + (uint32_t)tt) * DPL + (uint32_t)q;
      if (idx != have_idx) {
        have_idx = idx;
        have_word = D[idx];
      }
      return diag_cells_down_from(have_word, r);
    }
    __device__ uint32_t get_dir(int32_t i, int32_t col) {
      const int32_t j = col + (i - bw > 0 ? i - bw : 0);
      const int32_t x = j - i + bw;
      const int32_t tt = x / DPL, q = x - tt * DPL;
      const uint32_t n = (uint32_t)((i + j - k0 - (q & 1)) >> 1);
      const uint32_t m = n / 6u, r = n - 6u * m;
      const uint32_t idx = (m * GL + (uint32_t)tt) * DPL + (uint32_t)q;
      if (idx != have_idx) {
        have_idx = idx;
        have_word = D[idx];
      }
      return (have_word >> (5u * r)) & 31u;
    }
  } A{D, bw, k0, 0xFFFFFFFFu, 0u};
  uint32_t *tmp = J.tmp + (uint64_t)(J.big ? (J.list_base + li) : ci) * J.cap;
  bool ovf = false;
  const int32_t l = banded_traceback(A, refLen, readLen, bw, tmp, J.cap, &ovf);
  if (l < 0) {
    atomicAdd(&J.err[0], 1u);
    o.cigar_len = 0;
    J.ov[ci] = o;
    J.bw[ci] = 0;
    return;
  }
  if (ovf) {
    J.needbig[ci] = 1;  // rerun with a full-size temp slot
    if (J.big_count) atomicAdd(J.big_count, 1u);
    return;
  }
  o.cigar_len = (uint32_t)l;
  J.ov[ci] = o;
  J.bw[ci] = 0;
  J.needbig[ci] = J.big ? 2 : 0;  // 2: ops live in the big temp area
}

// ---- systolic banded attempt ----------------------------------------------------------------------
// The same recurrence as banded_attempt (ssw.c:645-693), but a candidate is spread over GL lanes the
// way the scoring kernels are: lane t owns DPL adjacent diagonals d = j - i of the band
// [-bw, +bw], and on every anti-diagonal k = i + j the diagonals of one parity advance by a cell.
// A cell needs (H, E) of the cell above it -- diagonal d + 1, one anti-diagonal earlier -- (H, F) of
// the cell to its left -- diagonal d - 1, one anti-diagonal earlier -- and its own diagonal's H from
// two anti-diagonals earlier: all in this lane's registers or one DPP shift away.  Cells that do
// not exist (outside the band, outside the matrix, not reached yet) read as H = E = F = 0, which is
// what the reference's zero-initialised rows and its h_b/e_b sentinels (ssw.c:645, 655) present;
// here a diagonal simply keeps its zero registers until its first real cell.  The 5 direction bits
// of a cell go to a per-candidate slab, one word per lane and turn, where the traceback (same
// banded_traceback as everywhere) finds cell (i, j) by arithmetic.
// Not handled here, sent back to the one-lane-per-candidate kernel (needbig = 3): spans the band
// covers completely (refLen <= 2 bw + 1), where the sentinel write of ssw.c:655 lands on a live cell.
template <int LMAX, int GL, int DPL, int BS>
__global__ __launch_bounds__(BS) void k_cigar_systolic(CigJob J, SwInputs in, SwParams p) {
  constexpr int NG = BS / GL, ND = GL * DPL;
  __shared__ __attribute__((aligned(16))) uint8_t s_w[NG][LMAX + STAGE_PAD];
  __shared__ __attribute__((aligned(16))) uint32_t s_tab[NG][LMAX + STAGE_PAD];
  const int32_t lane = threadIdx.x & 63;
  const int32_t t = lane & (GL - 1);
  const int32_t grp = threadIdx.x / GL;
  const uint32_t li = blockIdx.x * NG + grp;
  bool have = li < live_entries(J, (blockIdx.x + 1) * NG), special = false;
  uint32_t ci = 0;
  kslam_overlap o;
  memset(&o, 0, sizeof o);
  int32_t bw = 1, refLen = 0, readLen = 0;
  const uint8_t *wc = s_w[grp];
  const uint32_t *tab = s_tab[grp];
  if (have) {
    ci = J.list[J.list_base + li];
    o = J.ov[ci];
    bw = (int32_t)J.bw[ci];
    refLen = o.ref_end - o.ref_begin + 1;    // ssw.c:930-931
    readLen = o.query_end - o.query_begin + 1;
    if (2 * bw + 1 > ND || refLen <= 2 * bw + 1 || readLen > LMAX || refLen > LMAX || readLen < 1) {
      if (t == 0) J.needbig[ci] = 3;
      special = true;
      have = false;
    }
  }
  if (J.special_list) append_candidate(special && t == 0, ci, J.special_list, J.special_count);
  if (have) {
    const uint64_t ro = in.read_off[o.read];
    const uint64_t L = in.read_off[o.read + 1] - ro;
    const uint64_t go = in.genome_off[o.entry];
    const uint64_t G = in.genome_off[o.entry + 1] - go;
    const int64_t s0 = o.rel > 0 ? o.rel : 0;
    const int64_t wlen = (int64_t)min(L, G - (uint64_t)s0);
    // window position x of a flipped (revComp) window is genome position wlen - 1 - x
    const uint8_t *rsrc = in.genome_codes + go + s0 + (o.revcomp ? (wlen - 1 - o.ref_end) : (int64_t)o.ref_begin);
    tab += stage_span<GL, 1>(in.read_codes + ro + o.query_begin, readLen, false, t, s_w[grp], s_tab[grp], p);
    __builtin_amdgcn_wave_barrier();   // the read's codes were only a vehicle for the score rows
    wc += stage_span<GL, 6>(rsrc, refLen, o.revcomp != 0, t, s_w[grp], nullptr, p);
  }
  __syncthreads();
  // ---- the attempt
  const int32_t gO = in_vgpr(p.gap_open), gE = in_vgpr(p.gap_extend);
  const int32_t k0 = (bw & 1) ? -1 : 0;               // first anti-diagonal: parity of the lowest diagonal -bw
  const int32_t db = -bw + DPL * t;
  const int32_t ib = (k0 - db) >> 1;
  int32_t H[DPL], E[DPL], F[DPL], vs[DPL];
  uint32_t vl[DPL];
#pragma unroll
  for (int q = 0; q < DPL; q++) {
    const int32_t d = db + q, i0 = ib - (q >> 1);
    const int32_t lo = d < 0 ? -d : 0, hi = min(readLen, refLen - d);
    vs[q] = lo - i0;
    vl[q] = (have && DPL * t + q <= 2 * bw) ? (uint32_t)max(hi - lo, 0) : 0u;
    H[q] = 0;
    E[q] = 0;
    F[q] = 0;
  }
  const uint32_t *tp = tab + (ib - (DPL / 2 - 1));
  const uint8_t *wp = wc + (ib + db);
  int32_t mx = 0;
  const int32_t nturns = have && J.variant != 2 ? ((readLen + refLen - 2 - k0) >> 1) + 1 : 0;   // (variant 2: staging only)
  uint32_t *D = reinterpret_cast<uint32_t *>(J.scratch) + ((uint64_t)blockIdx.x * NG + grp) * (J.wave_slab / 4);
  // Direction words: six 5-bit cells per word, packed ALONG a diagonal -- word (m, lane, q) holds the
  // cells of the lane's diagonal q on turns 6m .. 6m + 5 -- because that is how the traceback walks:
  // a diagonal step stays in the same word five times out of six, so its chain of dependent loads
  // is a fifth as long as with one word per turn.
  uint32_t dw[DPL];
#pragma unroll
  for (int q = 0; q < DPL; q++) dw[q] = 0;
  int32_t sh = 0, grp6 = 0;   // 5 x (turn mod 6), turn / 6: wave-uniform
  // score rows and window codes of a turn are fetched up front for all of the lane's cells, existing
  // or not (an address outside the candidate's rows reads a neighbour's or nothing; the value is
  // not used): inside the per-cell branches the loads would each be waited for in turn
  uint32_t trow[DPL / 2], wcode[DPL / 2 + 1];
  int32_t nv = 0;   // the turn counter in a VGPR
  auto cell = [&](int q, int32_t Hu, int32_t Eu, int32_t Hl, int32_t Fl) {
    if ((uint32_t)(nv - vs[q]) < vl[q]) {
      const int h = q >> 1;
      const int32_t s = __builtin_amdgcn_sbfe(trow[h], wcode[h + (q & 1)], 6);
      int32_t t1 = Hu - gO, t2 = Eu - gE;                  // ssw.c:668-671
      const int32_t ev = max(t1, t2);
      const uint32_t de = t1 > t2 ? 1u : 0u;
      t1 = Hl - gO;                                        // ssw.c:673-676
      t2 = Fl - gE;
      const int32_t fv = max(t1, t2);
      const uint32_t df = t1 > t2 ? 1u : 0u;
      const int32_t e1 = max(ev, 0), f1 = max(fv, 0);      // ssw.c:678-682
      const int32_t m1 = max(e1, f1), dg = H[q] + s;
      const int32_t hv = max(m1, dg);
      mx = max(mx, hv);                                    // ssw.c:684
      const uint32_t dh = m1 <= dg ? 1u : (e1 > f1 ? 2u + de : 4u + df);   // ssw.c:686-690
      H[q] = hv;
      E[q] = ev;
      F[q] = fv;
      dw[q] |= (de | (df << 1) | (dh << 2)) << sh;
    }
  };
  for (int32_t n = 0;; n++) {
    if (__ballot(n < nturns) == 0ull) break;
#pragma unroll
    for (int h = 0; h < DPL / 2; h++) trow[h] = tp[DPL / 2 - 1 - h];
#pragma unroll
    for (int h = 0; h <= DPL / 2; h++) wcode[h] = wp[h];
    {  // phase A: the even diagonals of the lane; left neighbour of q = 0 lives in lane t - 1
      const int32_t hl = dpp_row_shr1(H[DPL - 1]), fl = dpp_row_shr1(F[DPL - 1]);
      int32_t Hl[DPL / 2], Fl[DPL / 2], Hu[DPL / 2], Eu[DPL / 2];
#pragma unroll
      for (int h = 0; h < DPL / 2; h++) {
        Hl[h] = h == 0 ? (t == 0 ? 0 : hl) : H[2 * h - 1];
        Fl[h] = h == 0 ? (t == 0 ? 0 : fl) : F[2 * h - 1];
        Hu[h] = H[2 * h + 1];
        Eu[h] = E[2 * h + 1];
      }
#pragma unroll
      for (int h = 0; h < DPL / 2; h++) cell(2 * h, Hu[h], Eu[h], Hl[h], Fl[h]);
    }
    {  // phase B: the odd diagonals; upper neighbour of q = DPL - 1 lives in lane t + 1
      const int32_t hu = dpp_row_shl1(H[0]), eu = dpp_row_shl1(E[0]);
      int32_t Hl[DPL / 2], Fl[DPL / 2], Hu[DPL / 2], Eu[DPL / 2];
#pragma unroll
      for (int h = 0; h < DPL / 2; h++) {
        Hl[h] = H[2 * h];
        Fl[h] = F[2 * h];
        Hu[h] = h == DPL / 2 - 1 ? (t == GL - 1 ? 0 : hu) : H[2 * h + 2];
        Eu[h] = h == DPL / 2 - 1 ? (t == GL - 1 ? 0 : eu) : E[2 * h + 2];
      }
#pragma unroll
      for (int h = 0; h < DPL / 2; h++) cell(2 * h + 1, Hu[h], Eu[h], Hl[h], Fl[h]);
    }
    sh += 5;
    if (sh == 30) {   // six turns gathered: one DPL-word store per lane
      if (6 * grp6 < nturns) {
#pragma unroll
        for (int q = 0; q < DPL; q++) D[((uint32_t)grp6 * GL + (uint32_t)t) * DPL + q] = dw[q];
      }
#pragma unroll
      for (int q = 0; q < DPL; q++) dw[q] = 0;
      sh = 0;
      grp6++;
    }
    tp += 1;
    wp += 1;
    nv += 1;
  }
  if (sh != 0 && 6 * grp6 < nturns) {
#pragma unroll
    for (int q = 0; q < DPL; q++) D[((uint32_t)grp6 * GL + (uint32_t)t) * DPL + q] = dw[q];
  }
#pragma unroll
  for (int m = 1; m < GL; m <<= 1) mx = max(mx, __shfl_xor(mx, m, GL));
  __threadfence_block();   // the direction words of the whole group, visible to its lane 0
  if (!have || t != 0 || J.variant == 1) return;   // (variant 1, measurement only: no traceback)
  const int32_t best = max(mx, J.bmax[ci]);   // `max` is carried across attempts, ssw.c:684
  J.bmax[ci] = best;
  if (best < (int32_t)o.score) {               // ssw.c:693-694: retry with twice the band
    J.bw[ci] = (uint32_t)bw * 2u;
    append_doubled(J, true, ci, (uint32_t)bw * 2u);
    return;
  }
  if (J.tb_flag) {   // the walk is a chain of dependent loads: it runs in its own kernel, 64 candidates to the wave
    // (a flag per list position, not a compacted list: one returning atomic per wave on ONE counter -- 20-odd thousand a
    // launch at ~12 ns apiece -- was two thirds of the 16-slot launch's 0.38 ms)
    J.tb_flag[li] = 1;
    return;
  }
  systolic_traceback<GL, DPL>(J, li, ci, o, bw, refLen, readLen, D);
}

// The tracebacks of a systolic launch, one candidate per LANE.  At the end of the DP kernel only lane 0 of a
// group of 8 or 16 walked -- a chain of ~28 dependent loads per candidate with an eighth of the wave's lanes
// busy, which was ~40 % of those kernels' time; here a wave has 64 walks in flight.
template <int GL, int DPL>
__global__ __launch_bounds__(256) void k_systolic_traceback(CigJob J) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= live_entries(J, (blockIdx.x + 1) * blockDim.x) || !J.tb_flag[x]) return;
  const uint32_t li = x;
  const uint32_t ci = J.list[J.list_base + li];
  kslam_overlap o = J.ov[ci];
  const int32_t bw = (int32_t)J.bw[ci];
  const int32_t refLen = o.ref_end - o.ref_begin + 1, readLen = o.query_end - o.query_begin + 1;
  const uint32_t *D = reinterpret_cast<const uint32_t *>(J.scratch) + (uint64_t)li * (J.wave_slab / 4);
  systolic_traceback<GL, DPL>(J, li, ci, o, bw, refLen, readLen, D);
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
  __shared__ uint4 s_rec[4][192];
  unsigned long long mycells = 0;
  // a workgroup walks through tiles of 256 rows: the DP-cell count it ends with is ONE atomic on `cells` (a same-address
  // atomic takes ~12 ns whoever issues it; one per 256 rows -- 32 k of them -- was most of this kernel's 0.41 ms)
  for (uint64_t tile = blockIdx.x; tile * 256 < n; tile += gridDim.x) {
  const uint64_t i = tile * 256 + threadIdx.x;
  const uint64_t first = i - (threadIdx.x & 63u);   // the wave's records travel together (common.h: wave_load_records)
  kslam_overlap o = wave_load_records(ov, first, n, s_rec[threadIdx.x >> 6]);
  if (i < n) {
    const uint64_t L = in.read_off[o.read + 1] - in.read_off[o.read];
    const uint64_t G = in.genome_off[o.entry + 1] - in.genome_off[o.entry];
    const int64_t s0 = o.rel > 0 ? o.rel : 0;
    const int64_t wlen = (int64_t)min(L, G - (uint64_t)s0);
    mycells += L * (unsigned long long)wlen;
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
  }
  wave_store_records(ov, first, n, o, s_rec[threadIdx.x >> 6]);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mycells += __shfl_down(mycells, d, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mycells;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(cells, sm[0] + sm[1] + sm[2] + sm[3]);
}

// The records of k_finalize WITHOUT their cigars, into a second array: what the device pairing needs (final
// coordinates) before the cigar stage has run.  Same arithmetic as k_finalize below.
__global__ __launch_bounds__(256) void k_final_coords(const kslam_overlap *__restrict__ ov, uint64_t n, SwInputs in,
                                                      kslam_overlap *__restrict__ out) {
  __shared__ uint4 s_rec[4][192];
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t first = i - (threadIdx.x & 63u);   // the wave's first record (a wave is past the end as a whole or not at all)
  if (first >= n) return;
  kslam_overlap o = wave_load_records(ov, first, n, s_rec[threadIdx.x >> 6]);
  if (i < n) {
    const uint64_t L = in.read_off[o.read + 1] - in.read_off[o.read];
    const uint64_t G = in.genome_off[o.entry + 1] - in.genome_off[o.entry];
    const int64_t s0 = o.rel > 0 ? o.rel : 0;
    const int64_t wlen = (int64_t)min(L, G - (uint64_t)s0);
    if (o.revcomp) {
      const int32_t rb = o.ref_begin, qb = o.query_begin;
      o.ref_begin = (int32_t)(wlen - (int64_t)(o.ref_end + 1));
      o.ref_end = (int32_t)(wlen - (int64_t)(rb + 1));
      o.query_begin = (int32_t)((int64_t)L - (int64_t)(o.query_end + 1));
      o.query_end = (int32_t)((int64_t)L - (int64_t)(qb + 1));
    }
    o.ref_begin += (int32_t)s0;
    o.ref_end += (int32_t)s0;
    o.cigar_len = 0;
    o.cigar_off = 0;
  }
  wave_store_records(out, first, n, o, s_rec[threadIdx.x >> 6]);
}
// rows no alignment pair refers to: no cigar (neither the inline <n>M nor a banded one)
__global__ void k_drop_unreferenced(kslam_overlap *__restrict__ ov, uint32_t *__restrict__ bw, const uint32_t *__restrict__ referenced,
                                    uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || referenced[i]) return;
  bw[i] = 0;
  if (ov[i].cigar_len) ov[i].cigar_len = 0;
}

}  // namespace

void final_coords_copy(const kslam_overlap *d_ov, uint64_t n, SwInputs in, kslam_overlap *d_out, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_final_coords, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_ov, n, in, d_out);
  HIPCHK(hipGetLastError());
}
void drop_unreferenced_cigars(kslam_overlap *d_ov, uint32_t *d_bw, const uint32_t *d_referenced, uint64_t n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_drop_unreferenced, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_ov, d_bw, d_referenced, n);
  HIPCHK(hipGetLastError());
}

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
                     CigarWork &W, uint64_t *n_cigar_out, uint32_t *n_tb_err, const Tuning &tune, hipStream_t s) {
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
    W.counters.ensure(16 * sizeof(uint32_t));
    uint32_t *cnt = W.counters.as<uint32_t>();   // [0..7] bin list sizes, [8] handed back, [9] long cigars, [10] tracebacks of a systolic launch, [11] classes present (generic loop)
    auto run_lists = [&](uint32_t cls, uint32_t mode) -> uint64_t {   // mode: 0 class, 1 big rerun, 2 handed back
      hipLaunchKernelGGL(k_class_flags, dim3(nb), dim3(256), 0, s, d_bw, W.needbig.as<uint8_t>(), n, cls,
                         mode, W.flags.as<uint32_t>());
      exclusive_scan_u32(W.flags.as<uint32_t>(), W.pos.as<uint32_t>(), n, d_tot, W.scan_tmp.p, s);
      uint64_t m = 0;
      read_back(&m, d_tot, sizeof m, s);
      if (m) hipLaunchKernelGGL(k_scatter_list, dim3(nb), dim3(256), 0, s, W.flags.as<uint32_t>(),
                                W.pos.as<uint32_t>(), n, W.list.as<uint32_t>());
      return m;
    };
    // one attempt for the m candidates of W.list with the systolic kernel; false when the band is
    // too wide for it
    // bit b: bin b (cig_bin) runs on the systolic kernel.  Measured on the bench workload: bins 0-2 (bands up to 4) are
    // cheaper on the register kernel (5 of 16 diagonal slots live on the systolic one, 3.9 against 5.4 ms in round 2);
    // bin 3 (bands 5..7) costs the same on both (register kernel <8> with mixed widths in a wave: 4.43 against 4.36 ms
    // for the stage); a band of 16 in registers needs 256 VGPRs + AGPR spills and loses (5.3 ms).
    const int sys_mask = tune.cigar_sys_mask;
    struct Route {
      const uint32_t *list = nullptr;
      uint32_t *const *bin_list = nullptr;   // [8]: where a candidate goes whose band doubled (nullptr: the host re-lists by flags)
      uint32_t *bin_count = nullptr, *special_list = nullptr, *special_count = nullptr, *big_count = nullptr;
      const uint32_t *count_dev = nullptr;   // the list's length where the kernels read it (CigJob::m_dev); then `m` is a capacity
      uint32_t sure = 0;                     // ... and this many entries of the list are known to exist
      bool no_reg = false;                   // the one-lane kernel whatever the band (the candidates the others handed over)
      uint32_t first = 0;                    // the launch starts at this entry of the list
    };
    auto launch_systolic = [&](uint64_t m, uint32_t slot_bw, uint32_t bin, const Route &R) -> bool {
      const uint32_t need = 2 * slot_bw + 1;
      if (need > 256 || !((sys_mask >> std::min(bin, 7u)) & 1)) return false;
      const int lm = lmax <= 160 ? 0 : (lmax <= 256 ? 1 : 2);
      const uint32_t LM = lm == 0 ? 160 : (lm == 1 ? 256 : 512);
      // lanes x diagonals per lane: 8 x 2 / 4 / 8 up to 64 slots, 16 x 8 / 16 up to 256 (a handful of
      // candidates per batch ever need more than 64, but one wave of the one-lane kernel sweeping
      // 129-cell rows alone takes milliseconds, with the whole chip waiting for it)
      const uint32_t GL = need > 64 ? 16 : 8;
      const uint32_t DPL = GL == 16 ? (need <= 128 ? 8 : 16) : (need <= 16 ? 2 : (need <= 32 ? 4 : 8));
      const uint32_t NG = 128 / GL;
      uint64_t slab = (uint64_t)((LM + 2 + 5) / 6 + 1) * GL * DPL * 4;   // direction words of one candidate: one per lane, diagonal and six turns
      slab = (slab + 255) & ~255ull;
      const uint64_t SCRATCH_BUDGET = 3ull << 30;
      const uint64_t groups_per_launch = std::max<uint64_t>(NG, (SCRATCH_BUDGET / slab) / NG * NG);
      W.scratch.ensure(std::min<uint64_t>((m + NG - 1) / NG * NG, groups_per_launch) * slab);
      for (uint64_t g0 = 0; g0 < m; g0 += groups_per_launch) {
        CigJob J;
        J.ov = d_ov; J.bw = d_bw; J.bmax = W.bmax.as<int32_t>(); J.needbig = W.needbig.as<uint8_t>();
        J.list = R.list ? R.list : W.list.as<uint32_t>();
        if (R.bin_list) for (int k = 0; k < 8; k++) J.bin_list[k] = R.bin_list[k];
        J.bin_count = R.bin_list ? R.bin_count : nullptr;
        J.special_list = R.special_list; J.special_count = R.special_count; J.big_count = R.big_count;
        J.list_base = R.first + (uint32_t)g0;
        J.m_dev = R.count_dev;
        J.m_sure = R.sure > J.list_base ? R.sure - J.list_base : 0;
        J.m = (uint32_t)std::min<uint64_t>(groups_per_launch, m - g0);
        J.slot_bw = slot_bw; J.lmax = lmax;
        J.cap = CIG_CAP;
        J.tmp = W.tmp.as<uint32_t>();
        J.big = 0;
        J.scratch = W.scratch.as<uint8_t>();
        J.wave_slab = slab;
        J.err = d_err;
#ifdef KSLAM_ABLATE
        J.variant = tune.cigar_variant;
#endif
        const unsigned nb = (unsigned)((J.m + NG - 1) / NG);
        // the tracebacks of this launch in a kernel of their own (KSLAM_CIGAR_TB=inline: at the end of the DP kernel)
        const bool tb_inline = tune.cigar_tb_inline;
        if (!tb_inline) {
          W.tb_list.ensure((uint64_t)J.m + 64);
          J.tb_flag = W.tb_list.as<uint8_t>();
          HIPCHK(hipMemsetAsync(J.tb_flag, 0, J.m, s));
        }
        const unsigned nb_tb = (unsigned)(((uint64_t)J.m + 255) / 256);
#define KSLAM_SYS(LMV, GLV, DPLV) \
  do { hipLaunchKernelGGL((k_cigar_systolic<LMV, GLV, DPLV, 128>), dim3(nb), dim3(128), 0, s, J, in, p); \
       if (J.tb_flag) hipLaunchKernelGGL((k_systolic_traceback<GLV, DPLV>), dim3(nb_tb), dim3(256), 0, s, J); } while (0)
#define KSLAM_SYS_LM(GLV, DPLV) \
  do { if (lm == 0) KSLAM_SYS(160, GLV, DPLV); else if (lm == 1) KSLAM_SYS(256, GLV, DPLV); else KSLAM_SYS(512, GLV, DPLV); } while (0)
        if (GL == 16 && DPL == 16) KSLAM_SYS_LM(16, 16);
        else if (GL == 16) KSLAM_SYS_LM(16, 8);
        else if (DPL == 2) KSLAM_SYS_LM(8, 2);
        else if (DPL == 4) KSLAM_SYS_LM(8, 4);
        else KSLAM_SYS_LM(8, 8);
#undef KSLAM_SYS_LM
#undef KSLAM_SYS
      }
      HIPCHK(hipGetLastError());
      return true;
    };
    auto launch = [&](uint64_t m, uint32_t slot_bw, bool big, const Route &R) {
      LdsLayout Y;
      Y.lmax = lmax; Y.W1 = slot_bw * 2 + 4; Y.wd = slot_bw * 2 + 1;
      // Directions: global slab.  Keeping them in LDS (KSLAM_CIGAR_DIRS=lds) was measured: the DP is
      // LDS-instruction bound, not bound by the traceback's loads, and the extra 38 KB per block cost
      // more occupancy than the traceback gained (class 1: 7.9 ms against 2.8 ms).
      Y.nch = ((lmax + 30) >> 4) + 1;   // 15 bytes of misalignment in front, up to 15 behind
      const uint32_t wpr_fit = (Y.wd + 5) / 6;
      const bool dir_in_lds = tune.cigar_dirs_lds &&
                              ((size_t)2 * Y.nch * 8 + (size_t)3 * Y.W1 * sizeof(int16_t) + (size_t)lmax * wpr_fit * 4) * 16 + 64 <= 64 * 1024;
      Y.wpr = dir_in_lds ? wpr_fit : 0;
      // narrow bands: the band in registers (KSLAM_CIGAR_REG=0 turns it off)
      const bool use_reg = tune.cigar_reg && !R.no_reg;
      // (a class-0 candidate has band 1: three slots, not the five of the band-2 instantiation it used to share)
      const uint32_t reg_bw = (!use_reg || big || Y.wpr) ? 0u : (slot_bw <= 1 ? 1u : (slot_bw <= 2 ? 2u : (slot_bw <= 4 ? 4u : (slot_bw <= 8 ? 8u : (slot_bw <= 16 ? 16u : 0u)))));
      // The register kernels keep no row arrays in LDS (round 6): a candidate they cannot hold -- a window no longer than its
      // band -- goes to the one-lane kernel through the same list the systolic kernels hand theirs over on.  3-5 KB less per
      // workgroup: 13 instead of 10 workgroups per CU for band 2.
      if (reg_bw) {
        if (!R.special_list) throw StatusError{KSLAM_ERR_STATE, "a register-band launch needs a hand-over list"};
        Y.W1 = 0;
      }
      const size_t base_lane = (size_t)2 * Y.nch * 8 + (size_t)3 * Y.W1 * sizeof(int16_t);
      const size_t per_lane = base_lane + (size_t)lmax * Y.wpr * 4;
      uint32_t nl = 64;
      while (nl > 1 && per_lane * nl + 64 > 64 * 1024) nl >>= 1;   // <= 64 KB: at least two blocks per CU
      Y.nl = nl;
      const size_t half = (size_t)Y.nch * nl * 8;
      const size_t rows_bytes = ((size_t)3 * Y.W1 * nl * sizeof(int16_t) + 15) & ~(size_t)15;
      const size_t lds = 2 * half + rows_bytes + (size_t)lmax * Y.wpr * 4 * nl;
      if (lds > 160 * 1024) throw StatusError{KSLAM_ERR_UNSUPPORTED, "banded traceback band does not fit LDS"};
      if (lds > 64 * 1024)
        for (const void *f : {reinterpret_cast<const void *>(&k_banded_lds<0>),
                              reinterpret_cast<const void *>(&k_banded_lds<1>),
                              reinterpret_cast<const void *>(&k_banded_lds<2>),
                              reinterpret_cast<const void *>(&k_banded_lds<4>),
                              reinterpret_cast<const void *>(&k_banded_lds<8>),
                              reinterpret_cast<const void *>(&k_banded_lds<16>)})
          HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      uint64_t slab = Y.wpr ? 256 : (uint64_t)lmax * std::max<uint32_t>(Y.wd, reg_bw ? 8u : 0u) * nl;   // direction
      slab = (slab + 255) & ~255ull;   // bytes per block; the register variant stores <= 8 bytes per row and lane
      const uint64_t SCRATCH_BUDGET = 3ull << 30;
      const uint64_t blocks_per_launch = std::max<uint64_t>(1, SCRATCH_BUDGET / slab);
      const uint64_t n_blocks = (m + nl - 1) / nl;
      W.scratch.ensure(std::min<uint64_t>(n_blocks, blocks_per_launch) * slab);
      for (uint64_t b0 = 0; b0 < n_blocks; b0 += blocks_per_launch) {
        CigJob J;
        J.ov = d_ov; J.bw = d_bw; J.bmax = W.bmax.as<int32_t>(); J.needbig = W.needbig.as<uint8_t>();
        J.list = R.list ? R.list : W.list.as<uint32_t>();
        if (R.bin_list) for (int k = 0; k < 8; k++) J.bin_list[k] = R.bin_list[k];
        J.bin_count = R.bin_list ? R.bin_count : nullptr;
        J.big_count = R.big_count;
        J.special_list = reg_bw ? R.special_list : nullptr;
        J.special_count = reg_bw ? R.special_count : nullptr;
        const uint64_t nb_here = std::min<uint64_t>(blocks_per_launch, n_blocks - b0);
        J.list_base = R.first + (uint32_t)(b0 * nl);
        J.m_dev = R.count_dev;
        J.m_sure = R.sure > J.list_base ? R.sure - J.list_base : 0;
        J.m = (uint32_t)std::min<uint64_t>(nb_here * nl, m - b0 * nl);
        J.slot_bw = slot_bw; J.lmax = lmax;
        J.cap = big ? cap_big : CIG_CAP;
        J.tmp = big ? W.tmp_big.as<uint32_t>() : W.tmp.as<uint32_t>();
        J.big = big ? 1 : 0;
        J.scratch = W.scratch.as<uint8_t>();
        J.wave_slab = slab;
        J.err = d_err;
#ifdef KSLAM_ABLATE
        J.variant = tune.cigar_variant;
#endif
        if (reg_bw == 1) hipLaunchKernelGGL(k_banded_lds<1>, dim3((unsigned)nb_here), dim3(64), lds, s, J, in, p, Y);
        else if (reg_bw == 2) hipLaunchKernelGGL(k_banded_lds<2>, dim3((unsigned)nb_here), dim3(64), lds, s, J, in, p, Y);
        else if (reg_bw == 4) hipLaunchKernelGGL(k_banded_lds<4>, dim3((unsigned)nb_here), dim3(64), lds, s, J, in, p, Y);
        else if (reg_bw == 8) hipLaunchKernelGGL(k_banded_lds<8>, dim3((unsigned)nb_here), dim3(64), lds, s, J, in, p, Y);
        else if (reg_bw == 16) hipLaunchKernelGGL(k_banded_lds<16>, dim3((unsigned)nb_here), dim3(64), lds, s, J, in, p, Y);
        else hipLaunchKernelGGL(k_banded_lds<0>, dim3((unsigned)nb_here), dim3(64), lds, s, J, in, p, Y);
      }
      HIPCHK(hipGetLastError());
    };
    // Bins 0..6 (bw < 64, cig_bin above): one partition of the candidates by bin, then bin after bin; an attempt that falls
    // short of the score doubles its band, which moves it to a later bin, and the kernels append such candidates to that
    // bin's list themselves.
    const bool debug = tune.debug;
    W.cls.ensure(n);
    for (int k = 0; k < 8; k++) W.cls_list[k].ensure((n + 1) * sizeof(uint32_t));
    W.special.ensure((n + 1) * sizeof(uint32_t));
    HIPCHK(hipMemsetAsync(cnt, 0, 16 * sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_cig_class, dim3(nb), dim3(256), 0, s, d_bw, n, W.cls.as<uint8_t>());
    uint32_t *lists[8];
    for (int k = 0; k < 8; k++) lists[k] = W.cls_list[k].as<uint32_t>();
    partition_bins(W.cls.as<uint8_t>(), n, lists, cnt, W.pos, s);
    // ONE sweep over the bins without a read-back in front of each (round 6; the GPU used to idle 25-60 us at each of ~11 of
    // them per chunk).  A bin's list grows while the earlier bins run (a failed attempt appends the candidate whose band
    // doubled), so the host cannot know its length when it queues the bin's launch -- but the kernels can: a launch is
    // sized for a CAPACITY, what the partition put into the bin plus room for an eighth of everything in the bins before
    // it (all of it while that is little), and its workgroups read the list's real length when they run (CigJob::m_dev; failures are rare, the spare
    // workgroups leave at once).  Then one read-back.  What did not fit a capacity, what the systolic launches handed to
    // the one-lane kernel (spans the band covers completely; laid out for the widest band among the bins that handed over)
    // and whatever THOSE append is left to rounds: each launches the part of every list nobody has run yet, then reads
    // the counters again, until nothing is new.
    uint32_t hc[10];
    uint32_t done[7] = {0, 0, 0, 0, 0, 0, 0};
    read_back(hc, cnt, sizeof hc, s);
    if (debug) fprintf(stderr, "[kslam] cigar bins as the SW stage asked for them: %u %u %u %u %u %u %u %u\n", hc[0], hc[1], hc[2], hc[3], hc[4], hc[5], hc[6], hc[7]);
    uint32_t special_bw = 0;
    {
      uint64_t before = 0;
      for (uint32_t bin = 0; bin < 7; bin++) {
        const uint64_t room = !tune.sweep_room ? 0 : (before <= 65536 ? before : std::max<uint64_t>(65536, before / 8));   // (spare workgroups cost ~1 ns each)
        const uint64_t cap = hc[bin] + room;
        before += hc[bin];
        if (cap == 0) continue;
        const uint32_t slot_bw = CIG_BIN_MAX_BW[bin];
        Route R;
        R.list = lists[bin];
        R.bin_list = lists;
        R.bin_count = cnt;
        R.special_list = W.special.as<uint32_t>();
        R.special_count = cnt + 8;
        R.big_count = cnt + 9;
        R.count_dev = cnt + bin;
        R.sure = hc[bin];
        if (!launch_systolic(cap, slot_bw, bin, R)) launch(cap, slot_bw, false, R);
        special_bw = std::max(special_bw, slot_bw);    // (register bins hand over too: windows no longer than their band)
        done[bin] = (uint32_t)cap;       // (clamped to the list's length below)
      }
    }
    for (int round = 0;; round++) {
      read_back(hc, cnt, sizeof hc, s);
      for (uint32_t bin = 0; bin < 7; bin++) done[bin] = std::min(done[bin], hc[bin]);
      bool progressed = false;
      if (hc[8]) {   // the few the systolic launches handed back
        if (debug) fprintf(stderr, "[kslam]   handed to the one-lane kernel: %u\n", hc[8]);
        Route R2;
        R2.list = W.special.as<uint32_t>();
        R2.bin_list = lists;
        R2.bin_count = cnt;
        R2.big_count = cnt + 9;
        R2.no_reg = true;
        launch(hc[8], special_bw, false, R2);
        HIPCHK(hipMemsetAsync(cnt + 8, 0, sizeof(uint32_t), s));
        special_bw = 0;
        progressed = true;
      }
      for (uint32_t bin = 0; bin < 7; bin++) {
        if (hc[bin] <= done[bin]) continue;
        const uint64_t m = hc[bin] - done[bin];
        const uint32_t slot_bw = CIG_BIN_MAX_BW[bin];
        if (debug) fprintf(stderr, "[kslam] cigar round %d bin %u (band <= %u): %llu candidates left over\n", round, bin, slot_bw, (unsigned long long)m);
        Route R;
        R.list = lists[bin];
        R.first = done[bin];
        R.bin_list = lists;
        R.bin_count = cnt;
        R.special_list = W.special.as<uint32_t>();
        R.special_count = cnt + 8;
        R.big_count = cnt + 9;
        if (!launch_systolic(m, slot_bw, bin, R)) launch(m, slot_bw, false, R);
        special_bw = std::max(special_bw, slot_bw);
        done[bin] = hc[bin];
        progressed = true;
      }
      if (!progressed) break;
    }
    // 64 and wider (never seen on real reads): the flag / scan / scatter loop, class c = floor(log2 bw) after class
    // (hc is current: the loop above ends on a round that found nothing new after its last read-back)
    if (hc[7]) {
      HIPCHK(hipMemsetAsync(cnt + 11, 0, sizeof(uint32_t), s));
      hipLaunchKernelGGL(k_class_mask, dim3(nb), dim3(256), 0, s, d_bw, W.needbig.as<uint8_t>(), n, cnt + 11);
      uint32_t present = 0;
      read_back(&present, cnt + 11, sizeof present, s);
      for (uint32_t cls = 6; cls < 31; cls++) {
        if (!((present >> cls) & 1u)) continue;
        uint64_t m = run_lists(cls, 0);
        if (m == 0) continue;
        present |= 2u << cls;   // what fails here doubles its band
        const uint32_t slot_bw = (2u << cls) - 1u;
        if (debug) fprintf(stderr, "[kslam] cigar class %u (band <= %u): %llu candidates\n", cls, slot_bw, (unsigned long long)m);
        Route R;
        R.big_count = cnt + 9;
        R.special_list = W.special.as<uint32_t>();
        R.special_count = cnt + 8;
        if (launch_systolic(m, slot_bw, 7, R)) {
          read_back(hc, cnt, sizeof hc, s);
          if (hc[8]) {
            Route R2 = R;
            R2.list = W.special.as<uint32_t>();
            R2.special_list = nullptr;
            R2.no_reg = true;
            launch(hc[8], slot_bw, false, R2);
            HIPCHK(hipMemsetAsync(cnt + 8, 0, sizeof(uint32_t), s));
          }
        } else {
          launch(m, slot_bw, false, R);
        }
      }
      read_back(hc, cnt, sizeof hc, s);
    }
    // candidates whose cigar did not fit the small temp slot: rerun with full-size slots
    uint64_t n_big = hc[9] ? run_lists(0, 1) : 0;
    if (n_big) {
      W.tmp_big.ensure(n_big * (uint64_t)cap_big * sizeof(uint32_t));
      HIPCHK(hipMemcpyAsync(W.big_pos.p, W.pos.p, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemsetAsync(d_tot + 1, 0, sizeof(uint64_t), s));
      hipLaunchKernelGGL(k_max_bw, dim3((unsigned)((n_big + 255) / 256)), dim3(256), 0, s, d_bw,
                         W.list.as<uint32_t>(), (uint32_t)n_big, reinterpret_cast<uint32_t *>(d_tot + 1));
      uint64_t mb = 0;
      read_back(&mb, d_tot + 1, sizeof mb, s);
      launch(n_big, (uint32_t)mb, true, Route());
    }
  }
  // cigar pool layout
  hipLaunchKernelGGL(k_cigar_lens, dim3(nb), dim3(256), 0, s, d_ov, n, W.flags.as<uint32_t>());
  exclusive_scan_u32_to_u64(W.flags.as<uint32_t>(), W.cig_off.as<uint64_t>(), n, d_tot, W.scan_tmp.p, s);
  uint64_t host_tot[3] = {0, 0, 0};
  read_back(host_tot, d_tot, sizeof host_tot, s);
  *n_cigar_out = host_tot[0];
  *n_tb_err = (uint32_t)(host_tot[2] & 0xFFFFFFFFu);
}

void cigar_finalize(kslam_overlap *d_ov, uint64_t n, SwInputs in, uint32_t lmax, CigarWork &W,
                    const uint32_t *d_bw, uint32_t *d_pool, uint64_t pool_base, uint64_t *d_cells, hipStream_t s) {
  if (n == 0) return;
  const unsigned nb = (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 8);
  hipLaunchKernelGGL(k_finalize, dim3(nb), dim3(256), 0, s, d_ov, n, in, W.cig_off.as<uint64_t>(),
                     W.tmp.as<uint32_t>(), CIG_CAP, W.tmp_big.as<uint32_t>(), 2 * lmax + 4,
                     W.big_pos.as<uint32_t>(), W.needbig.as<uint8_t>(), d_bw, d_pool, pool_base,
                     reinterpret_cast<unsigned long long *>(d_cells));
  HIPCHK(hipGetLastError());
}

}  // namespace kslam
