// extract.hip -- canonical 2-bit 32-mer extraction.
//
// Replaces getKMers_parallel / splitIntoKMersAndAddToVector / addBaseToKMers
// (reference src/KMer.h:160-181, 190-241, 246-280).  Semantics kept:
//   A=0 C=1 T=2 G=3, anything else 0 (KMer.h:246-268);
//   k-mer i starts at base i*gap; the canonical record is the forward k-mer iff
//   fwd < rc (palindromes take the rc branch, KMer.h:173);
//   rc offset is len-32-pos for reads and pos for genomes (KMer.h:176);
//   records land in sequence order at deterministic positions.
//
// MI355X design: HBM-bound (L bytes in, 16 B per k-mer out).  One wavefront per
// segment of up to 128 k-mers of one sequence: the wave loads the segment's
// bytes as aligned dwords (coalesced), packs them to 2 bits/base MSB-first in
// LDS, then every lane cuts its 64-bit k-mer out of three LDS words with a
// funnel shift, derives the reverse complement with v_bfrev (no per-base
// loop) and stores one 16-byte record -- a wave stores 1 KiB contiguous.
#include "common.h"

namespace kslam {

namespace {

__global__ void k_plan(const uint64_t *off, uint64_t n, uint32_t gap, uint32_t *nk, uint32_t *nseg) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t len = off[i + 1] - off[i];
  uint32_t k = len >= KSLAM_K ? (uint32_t)((len - KSLAM_K) / gap + 1) : 0;  // KMer.h:203
  nk[i] = k;
  nseg[i] = (k + SEG_KMERS - 1) / SEG_KMERS;
}

// one wave per sequence: lanes stride over the sequence's segments
__global__ void k_fill_segments(const uint32_t *nk, const uint64_t *rec_start, const uint64_t *seg_start,
                                uint64_t n, SegEntry *segs) {
  uint64_t seq = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (seq >= n) return;
  uint32_t lane = threadIdx.x & 63;
  uint32_t k = nk[seq];
  uint32_t ns = (k + SEG_KMERS - 1) / SEG_KMERS;
  uint64_t s0 = seg_start[seq], r0 = rec_start[seq];
  for (uint32_t j = lane; j < ns; j += 64) {
    SegEntry e;
    e.seq = (uint32_t)seq;
    e.q0 = j * SEG_KMERS;
    e.out = r0 + (uint64_t)j * SEG_KMERS;
    segs[s0 + j] = e;
  }
}

// short sequences (reads: one or two segments each): one thread per sequence
__global__ void k_fill_segments_short(const uint32_t *nk, const uint64_t *rec_start, const uint64_t *seg_start,
                                      uint64_t n, SegEntry *segs) {
  const uint64_t seq = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (seq >= n) return;
  const uint32_t k = nk[seq];
  const uint32_t ns = (k + SEG_KMERS - 1) / SEG_KMERS;
  const uint64_t s0 = seg_start[seq], r0 = rec_start[seq];
  for (uint32_t j = 0; j < ns; j++) {
    SegEntry e;
    e.seq = (uint32_t)seq;
    e.q0 = j * SEG_KMERS;
    e.out = r0 + (uint64_t)j * SEG_KMERS;
    segs[s0 + j] = e;
  }
}

// 4 ASCII bytes (little endian dword) -> 8 bits, first base in bits 7:6
__device__ inline uint32_t pack4(uint32_t x) {
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t c = (x >> (8 * j)) & 0xFFu;
    uint32_t code = (c >> 1) & 3u;                       // A 0x41, C 0x43, T 0x54, G 0x47
    uint32_t cand = (0x47544341u >> (8 * code)) & 0xFFu; // the one letter with that code
    code = (cand == c) ? code : 0u;                      // everything else encodes as A
    r |= code << (6 - 2 * j);
  }
  return r;
}

__device__ inline uint64_t revcomp64(uint64_t fwd) {
  uint64_t x = fwd ^ 0xAAAAAAAAAAAAAAAAull;  // complement: flip the high bit of each base
  x = __brevll(x);                           // reverses base order and the bits inside each base
  return ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);
}

__global__ __launch_bounds__(256) void k_extract(const uint8_t *__restrict__ bases,
                                                 const uint64_t *__restrict__ off,
                                                 const SegEntry *__restrict__ segs, uint64_t n_segs,
                                                 uint32_t gap, uint32_t is_gb, uint32_t id_base,
                                                 uint32_t words_per_wave, uint4 *__restrict__ out,
                                                 uint8_t *__restrict__ digits, SortPass dp) {
  extern __shared__ uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t seg_i = (uint64_t)blockIdx.x * 4 + w;
  if (seg_i >= n_segs) return;
  uint32_t *my = lds + (size_t)w * words_per_wave;
  uint8_t *my8 = reinterpret_cast<uint8_t *>(my);

  const SegEntry sg = segs[seg_i];
  const uint64_t s0 = off[sg.seq];
  const uint64_t len = off[sg.seq + 1] - s0;
  const uint32_t nk_total = (uint32_t)((len - KSLAM_K) / gap + 1);
  const uint32_t nk = min(SEG_KMERS, nk_total - sg.q0);
  const uint64_t p0 = (uint64_t)sg.q0 * gap;
  const uint32_t span = (nk - 1) * gap + KSLAM_K;
  const uint64_t a0 = s0 + p0;
  const uint64_t a_al = a0 & ~3ull;
  const uint32_t m = (uint32_t)(a0 & 3ull);
  const uint32_t ndw = (m + span + 3) >> 2;
  const uint32_t *src = reinterpret_cast<const uint32_t *>(bases + a_al);
  for (uint32_t d = lane; d < ndw; d += 64) {
    uint32_t x = src[d];
    my8[(d & ~3u) | (3u - (d & 3u))] = (uint8_t)pack4(x);  // word holds 16 bases MSB-first
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const uint32_t idbits = ((sg.seq + id_base) & 0x3FFFFFFFu) | (is_gb << 31);  // KMer.h:65-67
  for (uint32_t q = lane; q < nk; q += 64) {
    const uint32_t sidx = m + q * gap;
    const uint32_t wi = sidx >> 4, sh = (sidx & 15u) * 2u;
    const uint32_t W0 = my[wi], W1 = my[wi + 1], W2 = my[wi + 2];
    const uint64_t a = ((uint64_t)W0 << 32) | W1, b = ((uint64_t)W1 << 32) | W2;
    const uint64_t fwd = ((a << sh) & 0xFFFFFFFF00000000ull) | ((b << sh) >> 32);
    const uint64_t rc = revcomp64(fwd);
    const uint64_t pos = p0 + (uint64_t)q * gap;
    uint4 rec;
    if (fwd < rc) {  // KMer.h:173
      rec.x = (uint32_t)fwd; rec.y = (uint32_t)(fwd >> 32);
      rec.z = idbits;
      rec.w = (uint32_t)pos;
    } else {
      rec.x = (uint32_t)rc; rec.y = (uint32_t)(rc >> 32);
      rec.z = idbits | (1u << 30);
      rec.w = is_gb ? (uint32_t)pos : (uint32_t)(len - KSLAM_K - pos);  // KMer.h:176: len-1-i
    }
    out[sg.out + q] = rec;
    if (digits) {
      const uint32_t wv = dp.word == 0 ? rec.x : (dp.word == 1 ? rec.y : (dp.word == 2 ? rec.z : rec.w));
      digits[sg.out + q] = (uint8_t)sort_pass_digit(wv, dp);
    }
  }
}

}  // namespace

void extract_plan(const uint64_t *d_offsets, uint64_t n_seqs, uint32_t gap, uint32_t *d_nk,
                  uint32_t *d_nseg, uint64_t *d_rec_start, uint64_t *d_seg_start, uint64_t *d_totals,
                  void *d_scan_tmp, hipStream_t s) {
  if (n_seqs == 0) {
    HIPCHK(hipMemsetAsync(d_totals, 0, 2 * sizeof(uint64_t), s));
    return;
  }
  unsigned blocks = (unsigned)((n_seqs + 255) / 256);
  hipLaunchKernelGGL(k_plan, dim3(blocks), dim3(256), 0, s, d_offsets, n_seqs, gap, d_nk, d_nseg);
  exclusive_scan_u32_to_u64(d_nk, d_rec_start, n_seqs, d_totals, d_scan_tmp, s);
  exclusive_scan_u32_to_u64(d_nseg, d_seg_start, n_seqs, d_totals + 1, d_scan_tmp, s);
  HIPCHK(hipGetLastError());
}

void extract_fill_segments(const uint32_t *d_nk, const uint64_t *d_rec_start, const uint64_t *d_seg_start,
                           uint64_t n_seqs, uint32_t gap, SegEntry *d_segs, hipStream_t s, uint64_t n_segs) {
  (void)gap;
  if (n_seqs == 0) return;
  if (n_segs <= 4 * n_seqs) {   // reads: a wave per sequence would write one entry with 63 idle lanes
    hipLaunchKernelGGL(k_fill_segments_short, dim3((unsigned)((n_seqs + 255) / 256)), dim3(256), 0, s, d_nk,
                       d_rec_start, d_seg_start, n_seqs, d_segs);
    HIPCHK(hipGetLastError());
    return;
  }
  uint64_t threads = n_seqs * 64;
  unsigned blocks = (unsigned)((threads + 255) / 256);
  hipLaunchKernelGGL(k_fill_segments, dim3(blocks), dim3(256), 0, s, d_nk, d_rec_start, d_seg_start, n_seqs,
                     d_segs);
  HIPCHK(hipGetLastError());
}

void extract_kmers_launch(const uint8_t *d_bases, const uint64_t *d_offsets, const SegEntry *d_segs,
                          uint64_t n_segs, uint32_t gap, int is_gb, uint32_t id_base, uint4 *d_out,
                          hipStream_t s, uint8_t *d_digits, const SortPass *first_pass) {
  if (n_segs == 0) return;
  uint32_t span = (SEG_KMERS - 1) * gap + KSLAM_K + 3;
  uint32_t words = (span + 15) / 16 + 3;
  size_t lds = (size_t)4 * words * sizeof(uint32_t);
  unsigned blocks = (unsigned)((n_segs + 3) / 4);
  hipLaunchKernelGGL(k_extract, dim3(blocks), dim3(256), lds, s, d_bases, d_offsets, d_segs, n_segs, gap,
                     (uint32_t)(is_gb != 0), id_base, words, d_out, first_pass ? d_digits : nullptr,
                     first_pass ? *first_pass : SortPass{0, 0, 0});
  HIPCHK(hipGetLastError());
}

}  // namespace kslam
