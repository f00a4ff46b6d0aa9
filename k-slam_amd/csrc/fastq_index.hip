// fastq_index.hip -- FASTQ records found on the device.
//
// The reference pulls lines out of a std::ifstream one getline at a time
// (getSequencesFromFASTQFile, src/FASTQsequence.h:129-165; safeGetline, src/sequenceTools.h:45-73); the
// host parser of this library (host/fastq.cpp) indexes the text in parallel on the CPU.  For the pipelined
// entry the texts are uploaded anyway -- the bases and quality columns are cut out of them on the GPU --
// so the line index is built there too and the host is left with nothing to scan:
//   1. every 4 KiB tile (one wavefront) counts its line terminators; an exclusive scan numbers them;
//   2. the tiles write the terminator positions, ev[line];
//   3. one thread per record takes lines 4r .. 4r+3: where its bases and quality lie, how long they
//      are, and its identifier (header minus its first character, cut at the first space, then at the
//      first '/', src/FASTQsequence.h:61-71);
//   4. scans of the lengths give the column offsets; the identifiers are copied out; bases / quality
//      are gathered by k_gather_fields (details.hip).
// Semantics are those of host/fastq.cpp (which tests/test_fastq.py pins against the REAL reference
// reader): a line ends at "\n", "\r\n" or a lone "\r"; records are exactly four lines whatever they
// contain; at the true end of the stream the unterminated rest (if any) and then one more, empty, line
// are read, which can complete a record; `consumed` is where the reference's stream would stand.
#include "common.h"

namespace kslam {

namespace {

constexpr uint32_t FQ_TILE = 4096;   // bytes per tile: ONE WAVE, 64 lanes x 4 pieces of 16 bytes (piece k of lane l = bytes 1024 k + 16 l ..)

struct __attribute__((packed, aligned(1))) B16 {
  uint32_t w[4];
};

// bit j (j = 0..3) set iff byte j of w equals the byte repeated in pat: the exact zero-byte test on w ^ pat, then the four
// 0x80 flags gathered into a nibble by one multiplication (no flag reaches bits 28..31, no mask needed)
__device__ inline uint32_t eq_mask4(uint32_t w, uint32_t pat) {
  const uint32_t t = w ^ pat;
  const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
  return ((z >> 7) * 0x01020408u) >> 24;
}
// A line terminator starts at p: "\r" (alone or followed by "\n"), or "\n" not preceded by "\r"
// (src/sequenceTools.h:57-64; host/fastq.cpp: is_event).  The 16 bytes of a piece as two 16-bit masks.
__device__ inline void cr_lf_masks16(const B16 &v, uint32_t *cr, uint32_t *lf) {
  uint32_t c = 0, l = 0;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    c |= eq_mask4(v.w[d], 0x0D0D0D0Du) << (4 * d);
    l |= eq_mask4(v.w[d], 0x0A0A0A0Au) << (4 * d);
  }
  *cr = c;
  *lf = l;
}

// The terminators of the wave's tile: mask[k] = those in the lane's piece k (bit j = byte j of the piece).  A lane loads
// its four pieces with the loads in flight together (coalesced per piece: 64 lanes x 16 bytes = 1 KiB); whether the byte
// in front of a piece is "\r" comes from the neighbouring lane's mask (from lane 63's previous piece for lane 0; one byte
// load for the tile's first piece), not from a byte load per thread.  4 KiB tiles, 256 threads each with a barrier and 16
// bytes of work, ran at 1.2 TB/s.
__device__ inline void tile_event_masks(const uint8_t *t, uint64_t tile_start, uint64_t scan_len, uint32_t lane, uint32_t mask[4]) {
  B16 v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint64_t p = tile_start + 1024ull * k + 16ull * lane;
    if (p < scan_len) v[k] = *reinterpret_cast<const B16 *>(t + p);   // (up to 15 bytes past scan_len: the buffer's slack)
    else v[k].w[0] = v[k].w[1] = v[k].w[2] = v[k].w[3] = 0;
  }
  uint32_t before = (lane == 0 && tile_start > 0 && tile_start < scan_len) ? (t[tile_start - 1] == '\r' ? 1u : 0u) : 0u;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint64_t p = tile_start + 1024ull * k + 16ull * lane;
    uint32_t cr, lf;
    cr_lf_masks16(v[k], &cr, &lf);
    const uint32_t valid = p >= scan_len ? 0u : (scan_len - p >= 16 ? 0xFFFFu : ((1u << (uint32_t)(scan_len - p)) - 1u));
    cr &= valid;
    lf &= valid;
    const uint32_t last_is_cr = cr >> 15;
    const uint32_t from_left = (uint32_t)__shfl_up((int)last_is_cr, 1, 64);          // lane l - 1's piece k ends just before mine
    const uint32_t prev_cr = lane == 0 ? before : from_left;
    mask[k] = cr | (lf & ~(((cr << 1) | prev_cr) & 0xFFFFu));
    before = (uint32_t)__shfl((int)last_is_cr, 63, 64);                               // lane 0's piece k + 1 follows lane 63's piece k
  }
}

__global__ __launch_bounds__(256) void k_fq_count(const uint8_t *__restrict__ t, uint64_t scan_len, uint64_t tiles,
                                                  uint32_t *__restrict__ tile_count) {
  const uint64_t tile = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= tiles) return;   // (whole waves)
  const uint32_t lane = threadIdx.x & 63;
  uint32_t m[4];
  tile_event_masks(t, tile * FQ_TILE, scan_len, lane, m);
  uint32_t c = (uint32_t)(__popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]));
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
  if (lane == 0) tile_count[tile] = c;
}

__global__ __launch_bounds__(256) void k_fq_events(const uint8_t *__restrict__ t, uint64_t scan_len, uint64_t tiles,
                                                   const uint32_t *__restrict__ tile_base, uint64_t *__restrict__ ev) {
  const uint64_t tile = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= tiles) return;
  const uint32_t lane = threadIdx.x & 63;
  uint32_t m[4];
  tile_event_masks(t, tile * FQ_TILE, scan_len, lane, m);
  uint32_t at = tile_base[tile];
#pragma unroll
  for (int k = 0; k < 4; k++) {   // positions ascend with (piece, lane, byte)
    const uint32_t c = (uint32_t)__popc(m[k]);
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t x = __shfl_up(incl, d, 64);
      if (lane >= (uint32_t)d) incl += x;
    }
    uint32_t mine = at + incl - c;
    const uint64_t p0 = tile * FQ_TILE + 1024ull * k + 16ull * lane;
    uint32_t mm = m[k];
    while (mm) {
      ev[mine++] = p0 + (uint32_t)__builtin_ctz(mm);
      mm &= mm - 1;
    }
    at += (uint32_t)__shfl((int)incl, 63, 64);
  }
}

__device__ inline uint64_t line_after(const uint8_t *t, uint64_t len, uint64_t p) {
  return (t[p] == '\r' && p + 1 < len && t[p + 1] == '\n') ? p + 2 : p + 1;
}

struct FqStream {
  const uint8_t *text;   // the stream's first byte (device)
  uint64_t len;
  const uint64_t *ev;    // terminator positions
  uint64_t terminated;   // their number
  uint64_t rest_start;   // text after the last terminator
  uint64_t n;            // records taken
  uint64_t shift;        // position of the stream inside [r1 | r2]
  uint64_t first;        // number of its first record in the batch
};

// line `l` of the stream: [start, end) and where the next line starts (host/fastq.cpp: index_stream)
__device__ inline void line_span(const FqStream &s, uint64_t l, uint64_t *start, uint64_t *end, uint64_t *next) {
  if (l < s.terminated) {
    *start = l == 0 ? 0 : line_after(s.text, s.len, s.ev[l - 1]);
    *end = s.ev[l];
    *next = line_after(s.text, s.len, s.ev[l]);
  } else if (l == s.terminated && s.rest_start < s.len) {
    *start = s.rest_start; *end = s.len; *next = s.len;     // the unterminated rest
  } else {
    *start = s.len; *end = s.len; *next = s.len;            // the empty line read at end of stream
  }
}

// first k in [0, n) with h[k] == c, or n: 16 bytes a step (one unaligned load), equal bytes found with the exact
// zero-byte test on h ^ cccc.  Reads up to 15 bytes past h + n: the text buffer has 64 spare bytes after its end.
struct __attribute__((packed, aligned(1))) FqBytes16 {
  uint32_t w[4];
};
__device__ inline uint64_t find_byte(const uint8_t *h, uint64_t n, uint32_t c) {
  const uint32_t pat = c * 0x01010101u;
  for (uint64_t k0 = 0; k0 < n; k0 += 16) {
    const FqBytes16 v = *reinterpret_cast<const FqBytes16 *>(h + k0);
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const uint32_t t = v.w[w] ^ pat;
      const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);   // 0x80 in the bytes that are equal
      if (z) {
        const uint64_t k = k0 + 4u * (uint32_t)w + ((uint32_t)__builtin_ctz(z) >> 3);
        return k < n ? k : n;
      }
    }
  }
  return n;
}

__global__ __launch_bounds__(256) void k_fq_fields(FqStream s, uint64_t *__restrict__ bases_at, uint64_t *__restrict__ quality_at,
                                                   uint32_t *__restrict__ blen, uint64_t *__restrict__ id_at,
                                                   uint32_t *__restrict__ id_len, uint64_t *__restrict__ misc /*[0] after_quality of the last record, [1] error flags*/) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= s.n) return;
  uint64_t a, b, nx;
  // identifier: FASTQSequence::FASTQSequence, src/FASTQsequence.h:61-71 (host/fastq.cpp: identifier_of)
  line_span(s, 4 * r, &a, &b, &nx);
  uint64_t istart = 0;
  uint32_t ilen = 0;
  if (b - a > 1) {
    const uint8_t *h = s.text + a;
    const uint64_t flen = b - a;
    const uint64_t sp = find_byte(h, flen, ' ');
    const uint64_t end = sp == flen ? flen : (sp == 0 ? 1 : sp);
    istart = a + 1;
    ilen = (uint32_t)find_byte(h + 1, end - 1, '/');
  }
  line_span(s, 4 * r + 1, &a, &b, &nx);
  const uint64_t bs = a, bl = b - a;
  line_span(s, 4 * r + 3, &a, &b, &nx);
  const uint64_t o = s.first + r;
  bases_at[o] = s.shift + bs;
  quality_at[o] = s.shift + a;
  blen[o] = (uint32_t)bl;
  id_at[o] = s.shift + istart;
  id_len[o] = ilen;
  if (b - a != bl || bl > 0xFFFFFFFFull) atomicOr(reinterpret_cast<unsigned long long *>(misc + 1), 1ull);   // quality line of another length
  if (r + 1 == s.n) misc[0] = nx;
}

__global__ __launch_bounds__(256) void k_fq_ids(const uint8_t *__restrict__ text, const uint64_t *__restrict__ id_at,
                                                const uint32_t *__restrict__ id_len, const uint64_t *__restrict__ ids_off,
                                                uint64_t n, uint8_t *__restrict__ ids) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint8_t *src = text + id_at[r];
  uint8_t *dst = ids + ids_off[r];
  for (uint32_t k = 0; k < id_len[r]; k++) dst[k] = src[k];
}

// terminators of one stream -> ev; returns their number
uint64_t index_events(const uint8_t *d_text, uint64_t scan_len, FastqWork &W, DevBuf &ev, hipStream_t s) {
  if (scan_len == 0) return 0;
  const uint64_t tiles = (scan_len + FQ_TILE - 1) / FQ_TILE;
  if (tiles >= (1ull << 31)) throw StatusError{KSLAM_ERR_UNSUPPORTED, "FASTQ text of 8 TiB or more in one call"};
  W.tile_count.ensure((tiles + 1) * sizeof(uint32_t));
  W.tile_base.ensure((tiles + 1) * sizeof(uint32_t));
  W.scan_tmp.ensure(scan_tmp_bytes(tiles));
  W.totals.ensure(8 * sizeof(uint64_t));
  hipLaunchKernelGGL(k_fq_count, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, d_text, scan_len, tiles, W.tile_count.as<uint32_t>());
  exclusive_scan_u32(W.tile_count.as<uint32_t>(), W.tile_base.as<uint32_t>(), tiles, W.totals.as<uint64_t>(), W.scan_tmp.p, s);
  uint64_t total = 0;
  read_back(&total, W.totals.p, sizeof total, s);
  if (total >= (1ull << 32)) throw StatusError{KSLAM_ERR_UNSUPPORTED, "2^32 or more lines in one FASTQ call"};
  ev.ensure((total + 1) * sizeof(uint64_t));
  hipLaunchKernelGGL(k_fq_events, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, d_text, scan_len, tiles, W.tile_base.as<uint32_t>(),
                     ev.as<uint64_t>());
  HIPCHK(hipGetLastError());
  return total;
}

}  // namespace

void fastq_index_device(const uint8_t *d_text, uint64_t len1, uint64_t len2, const uint8_t *h_tail1, const uint8_t *h_tail2,
                        uint64_t max_pairs, bool at_eof, FastqWork &W, FastqIndexResult *res, hipStream_t s, bool single) {
  memset(res, 0, sizeof *res);
  FqStream st[2];
  const uint64_t lens[2] = {len1, len2};
  const uint8_t *tails[2] = {h_tail1, h_tail2};
  for (int k = 0; k < 2; k++) {
    FqStream &q = st[k];
    q.text = d_text + (k ? len1 : 0);
    q.len = lens[k];
    q.shift = k ? len1 : 0;
    // without the rest of the stream a trailing "\r" may or may not be half of "\r\n"
    uint64_t scan_len = q.len;
    if (!at_eof && q.len && tails[k] && tails[k][0] == '\r') scan_len = q.len - 1;   // tails[k][0]: the stream's last byte
    q.terminated = index_events(q.text, scan_len, W, W.ev[k], s);
    q.ev = W.ev[k].as<uint64_t>();
    uint64_t last_ev = 0;
    q.rest_start = 0;
    if (q.terminated) {
      read_back(&last_ev, W.ev[k].as<uint64_t>() + (q.terminated - 1), sizeof last_ev, s);
      // line_after(last_ev): needs the byte at last_ev and the one after it: the host has the stream's last two bytes
      // only when the terminator is at the very end; read both from the device instead
      uint8_t two[2] = {0, 0};
      const uint64_t nb = last_ev + 1 < q.len ? 2 : 1;
      read_back(two, q.text + last_ev, nb, s);
      q.rest_start = (two[0] == '\r' && nb == 2 && two[1] == '\n') ? last_ev + 2 : last_ev + 1;
    }
    const uint64_t rest_lines = at_eof ? (q.rest_start < q.len ? 2 : 1) : 0;
    uint64_t n = (q.terminated + rest_lines) / 4;
    if (max_pairs && n > max_pairs) n = max_pairs;
    q.n = n;
  }
  // single end (getSequencesFromFASTQFile, src/FASTQsequence.h:129-147): one stream, max_pairs counts reads
  if (!single && st[0].n != st[1].n) throw StatusError{KSLAM_ERR_ARG, "mismatch in R1 and R2 size"};   // src/FASTQsequence.h:118-122
  const uint64_t n = st[0].n + st[1].n;
  st[0].first = 0;
  st[1].first = st[0].n;
  W.bases_at.ensure((n + 1) * sizeof(uint64_t));
  W.quality_at.ensure((n + 1) * sizeof(uint64_t));
  W.blen.ensure((n + 1) * sizeof(uint32_t));
  W.id_at.ensure((n + 1) * sizeof(uint64_t));
  W.id_len.ensure((n + 1) * sizeof(uint32_t));
  W.bases_off.ensure((n + 2) * sizeof(uint64_t));
  W.ids_off.ensure((n + 2) * sizeof(uint64_t));
  W.scan_tmp.ensure(scan_tmp_bytes(std::max<uint64_t>(n, 1)));
  W.totals.ensure(8 * sizeof(uint64_t));
  uint64_t *tot = W.totals.as<uint64_t>();
  HIPCHK(hipMemsetAsync(tot, 0, 8 * sizeof(uint64_t), s));
  uint64_t after[2] = {0, 0};
  for (int k = 0; k < 2; k++) {
    if (!st[k].n) continue;
    hipLaunchKernelGGL(k_fq_fields, dim3((unsigned)((st[k].n + 255) / 256)), dim3(256), 0, s, st[k], W.bases_at.as<uint64_t>(),
                       W.quality_at.as<uint64_t>(), W.blen.as<uint32_t>(), W.id_at.as<uint64_t>(), W.id_len.as<uint32_t>(),
                       tot + 2 + 2 * k);
  }
  uint64_t h[8];
  read_back(h, tot, sizeof h, s);
  if (h[3] | h[5]) throw StatusError{KSLAM_ERR_ARG, "a read's quality line is not as long as its bases line"};
  after[0] = h[2];
  after[1] = h[4];
  uint64_t b_total = 0, i_total = 0;
  if (n) {
    exclusive_scan_u32_to_u64(W.blen.as<uint32_t>(), W.bases_off.as<uint64_t>(), n, tot, W.scan_tmp.p, s);
    exclusive_scan_u32_to_u64(W.id_len.as<uint32_t>(), W.ids_off.as<uint64_t>(), n, tot + 1, W.scan_tmp.p, s);
    uint64_t t2[2];
    read_back(t2, tot, sizeof t2, s);
    b_total = t2[0];
    i_total = t2[1];
  }
  // the closing entries of the offset arrays
  HIPCHK(hipMemcpyAsync(W.bases_off.as<uint64_t>() + n, &b_total, sizeof b_total, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(W.ids_off.as<uint64_t>() + n, &i_total, sizeof i_total, hipMemcpyHostToDevice, s));
  W.ids.ensure(i_total + 64);
  if (n) hipLaunchKernelGGL(k_fq_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_text, W.id_at.as<uint64_t>(),
                            W.id_len.as<uint32_t>(), W.ids_off.as<uint64_t>(), n, W.ids.as<uint8_t>());
  HIPCHK(hipGetLastError());
  HIPCHK(stream_wait(s));   // b_total / i_total are read by the copies above
  for (int k = 0; k < 2; k++) {
    // short of max_pairs at the true end of the stream, the reference's loop has read on to the end
    if (at_eof && (!max_pairs || st[k].n < max_pairs)) res->consumed[k] = st[k].len;
    else res->consumed[k] = st[k].n ? after[k] : 0;
  }
  res->n_reads = n;
  res->bases_total = b_total;
  res->ids_total = i_total;
  res->d_bases_at = W.bases_at.as<uint64_t>();
  res->d_quality_at = W.quality_at.as<uint64_t>();
  res->d_bases_off = W.bases_off.as<uint64_t>();
  res->d_ids_off = W.ids_off.as<uint64_t>();
  res->d_ids = W.ids.as<uint8_t>();
}

}  // namespace kslam
