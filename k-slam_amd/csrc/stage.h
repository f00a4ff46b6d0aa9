// stage.h -- staging of base spans into LDS, shared by the SW kernels (sw.hip) and the systolic
// CIGAR kernel (cigar.hip).
#pragma once
#include "common.h"

namespace kslam {

// A wave-uniform value in a vector register (kept there: the compiler would hold it in an SGPR).
// On gfx950 the plain 32-bit add / sub / and / or issue in 2 cycles per wave when every source is a
// VGPR or a literal and in 4 with an SGPR source (tools/valu_peak.hip), so loop-invariant operands
// of the DP sweeps are parked in VGPRs.
__device__ inline int32_t in_vgpr(int32_t x) {
  int32_t v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x));
  return v;
}

__device__ inline int32_t dpp_row_shr1(int32_t v) {
  // lane i of each 16-lane row receives lane i-1; lane 0 receives 0
  return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
}
__device__ inline int32_t dpp_row_shl1(int32_t v) {
  // lane i of each 16-lane row receives lane i+1; lane 15 receives 0
  return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xF, 0xF, true);
}

// ---- wide staging --------------------------------------------------------------------------------
// Byte-at-a-time staging (stage_candidate in sw.hip, kept for the full-matrix kernel) cost as much as
// a 16-diagonal band sweep: one memory instruction and one LDS store per base.  Here a span is fetched as 16-byte chunks from the 16-byte-aligned
// address below it, a chunk per lane, converted in registers and stored with one 16-byte LDS
// write (the sources are the pre-encoded base arrays, see encode_bases) -- at the chunk's own
// position, NOT re-aligned: the buffer holds the codes of the aligned
// chunks and the caller gets the offset at which its span starts.  A reverse-complemented window
// stores the chunks in reverse order with their bytes reversed, which lands the reversed span at a
// (different) offset of the same buffer.  Both base arrays are the library's own copies: 256-byte
// aligned starts and 64 bytes of slack at the end, so the aligned reads never leave them.
constexpr int STAGE_PAD = 32;   // bytes a span buffer needs beyond the longest span

// four encoded bases (see encode_bases) -> four SSW codes x WS, optionally complemented
template <int WS>
__device__ inline uint32_t codes_of_dword(uint32_t v, bool comp) {
  uint32_t c = v & 0x07070707u;
  if (comp) c ^= ((v >> 3) & 0x01010101u) * 3u;   // 3 - code where the base is complementable
  if (WS == 6) c = (c << 2) + (c << 1);           // codes <= 4: no carry between the bytes
  return c;
}

// 6-bit packed score row (one field per reference code 0..3; code 4 = N / padding reads field 4).
// `bias` is added to every score, also to the zeros of N: the band kernels fold a constant per
// diagonal step into the table (see k_sw_band).
__device__ inline uint32_t score_row(uint32_t q, const SwParams &p, int32_t bias) {
  const uint32_t mis = (uint32_t)(bias - p.mismatch) & 63u, mat = (uint32_t)(bias + p.match) & 63u;
  const uint32_t zero = (uint32_t)bias & 63u;
  const uint32_t all_mis = mis | (mis << 6) | (mis << 12) | (mis << 18) | (zero << 24);
  const uint32_t all_zero = zero | (zero << 6) | (zero << 12) | (zero << 18) | (zero << 24);
  return q > 3u ? all_zero : (all_mis ^ ((mis ^ mat) << (6u * q)));
}

// GL lanes stage src[0..len) into dst (16-byte aligned, >= len + STAGE_PAD bytes) as WS x code;
// when tab != nullptr also the score row of every base.  Returns the offset of element 0 in dst.
template <int GL, int WS>
__device__ inline int32_t stage_span(const uint8_t *src, int32_t len, bool rc, int32_t t, uint8_t *dst,
                                     uint32_t *tab, const SwParams &p, int32_t bias = 0) {
  const uint32_t shift = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 15u);
  const uint4 *base = reinterpret_cast<const uint4 *>(src - shift);
  const int32_t nch = ((int32_t)shift + len + 15) >> 4;
  for (int32_t k = t; k < nch; k += GL) {
    const uint4 v = base[k];
    uint4 c;
    int32_t at;
    if (!rc) {
      c.x = codes_of_dword<WS>(v.x, false);
      c.y = codes_of_dword<WS>(v.y, false);
      c.z = codes_of_dword<WS>(v.z, false);
      c.w = codes_of_dword<WS>(v.w, false);
      at = k;
    } else {   // reversed: last chunk first, bytes of a chunk back to front
      c.x = __builtin_bswap32(codes_of_dword<WS>(v.w, true));
      c.y = __builtin_bswap32(codes_of_dword<WS>(v.z, true));
      c.z = __builtin_bswap32(codes_of_dword<WS>(v.y, true));
      c.w = __builtin_bswap32(codes_of_dword<WS>(v.x, true));
      at = nch - 1 - k;
    }
    reinterpret_cast<uint4 *>(dst)[at] = c;
    if (tab) {
      const uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
      for (int d = 0; d < 4; d++) {
        uint4 r;
        r.x = score_row(((w[d]) & 0xFFu) / (uint32_t)WS, p, bias);
        r.y = score_row(((w[d] >> 8) & 0xFFu) / (uint32_t)WS, p, bias);
        r.z = score_row(((w[d] >> 16) & 0xFFu) / (uint32_t)WS, p, bias);
        r.w = score_row((w[d] >> 24) / (uint32_t)WS, p, bias);
        reinterpret_cast<uint4 *>(tab)[at * 4 + d] = r;
      }
    }
  }
  return rc ? 16 * nch - (int32_t)shift - len : (int32_t)shift;
}


}  // namespace kslam
