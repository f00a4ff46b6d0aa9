// details.hip -- what the SAM writer needs to know about an alignment beyond its CIGAR: the MD tag,
// the edit distance NM and the log-probability of the read given the alignment.
//
// Replaces the walk of getCigarAndMD (reference src/SAM.h:101-237) over CIGAR + read + quality + entry
// bases.  On the host that walk is bound by scattered reads into a multi-gigabyte database (3 M random
// 150-byte windows per batch); here reads, qualities, entries and CIGARs are in HBM
// already, one thread walks one alignment, and the host tail receives per row
//   nm      -- mismatching M columns + inserted + deleted bases                 (SAM.h:150,169,181)
//   logp    -- sum over the M columns, IN COLUMN ORDER, of matchTable[q] / misMatchTable[q]
//              (SAM.h:146-152; the tables of SAM.h:33-48 are computed on the host with libm and handed
//              in, and a chain of IEEE double additions in a fixed order is the same on every machine)
//   MD text -- the merged components of SAM.h:204-235: match counts added up, mismatched reference bases,
//              ^deleted bases, a "0" between a deletion and a mismatch
// Semantics kept: the query is the read, or its reverse complement with only upper-case A/C/G/T
// complemented (reverseComplement, src/sequenceTools.h:77-116) and the quality string reversed
// (SAM.h:113-116); columns compare raw characters (SAM.h:144); the walk starts at ref_begin and at
// query position query_begin (SAM.h:118-125).
//
// MI355X: each lane streams three byte sequences (entry window, read, quality) with one unaligned
// 16-byte load per 16 columns and stream; MD bytes go to a 48-byte slot per row first (it fits for all
// but pathological rows), an exclusive scan of the lengths lays out the pool, and a gather pass moves the
// slots into it (rows that did not fit are walked again, straight into the pool).
#include "common.h"

namespace kslam {

namespace {

constexpr uint32_t MD_SLOT = 48;

struct __attribute__((packed, aligned(1))) Bytes16 {
  uint32_t w[4];
};
__device__ inline uint32_t byte_of(const Bytes16 &v, int j) { return (v.w[j >> 2] >> (8 * (j & 3))) & 0xFFu; }

// complement of reverseComplement: A<->T, C<->G, upper case only (src/sequenceTools.h:77-97)
__host__ __device__ inline uint32_t complement(uint32_t c) {
  const uint32_t at = (c == 'A' || c == 'T') ? (uint32_t)('A' ^ 'T') : 0u;
  const uint32_t cg = (c == 'C' || c == 'G') ? (uint32_t)('C' ^ 'G') : 0u;
  return c ^ at ^ cg;
}

// What a block keeps in LDS for the walk:
//   tab[0..99] matchTable, tab[100] = matchTable[0] (a quality character outside 0..99 is flagged and walks
//   on as 0, as before), tab[101..200] misMatchTable, tab[201] = misMatchTable[0], tab[202] = 0.0 (columns
//   past the end of a run add nothing: x + 0.0 is x);
//   lut[0..255] identity, lut[256..511] complement: a lane reads lut[its strand's half + base], so the
//   complement costs one LDS byte read and no VALU work.
constexpr uint32_t TAB_MM = 101, TAB_ZERO = 202, TAB_N = 203, Q_CLAMP = 100;
struct WalkLds {
  double tab[TAB_N];
  uint8_t lut[512];
};
__device__ inline void fill_walk_lds(WalkLds &S, const double *__restrict__ tables) {
  for (uint32_t k = threadIdx.x; k < TAB_N; k += blockDim.x) {
    double v = 0.0;
    if (k < 100) v = tables[k];
    else if (k == 100) v = tables[0];
    else if (k <= 200) v = tables[100 + (k - TAB_MM)];
    else if (k == 201) v = tables[100];
    S.tab[k] = v;
  }
  for (uint32_t k = threadIdx.x; k < 512; k += blockDim.x) S.lut[k] = (uint8_t)(k < 256 ? k : complement(k - 256));
  __syncthreads();
}

struct MdOut {          // the streaming form of SAM.h:204-235 (host: MdWriter in host/tail.cpp)
  uint8_t *dst;
  uint32_t cap, n = 0;
  uint32_t pending = 0;
  bool have_pending = false, after_del = false;
  __device__ void put(uint32_t c) {
    if (n < cap) dst[n] = (uint8_t)c;
    n++;
  }
  __device__ void num(uint32_t v) {
    if (v < 1000u) {            // a run of matches is at most a read long: up to three digits, constant divisions
      const uint32_t h = v / 100u, t = (v - 100u * h) / 10u, u = v - 100u * h - 10u * t;
      if (v >= 100u) put('0' + h);
      if (v >= 10u) put('0' + t);
      put('0' + u);
      return;
    }
    uint32_t p = 1000u;
    while (v / p >= 10u) p *= 10u;
    while (p) {
      put('0' + v / p);
      v %= p;
      p /= 10u;
    }
  }
  __device__ void matches(uint32_t run) {
    if (run) { pending += run; have_pending = true; }
  }
  __device__ void flush() {
    if (have_pending) { num(pending); pending = 0; have_pending = false; after_del = false; }
  }
  __device__ void mismatch(uint32_t ref_base) {
    flush();
    if (after_del) { put('0'); after_del = false; }
    put(ref_base);
  }
};

struct WalkResult {
  double logp;
  uint32_t nm, md_len, flags;
};

// flags: 1 = a quality character outside phred+33 0..99 (the host tail rejects the batch when it needs
// that probability), 2 = the CIGAR runs past the read or the entry (the host tail rejects the batch)
__device__ inline uint32_t bswap_(uint32_t x) { return __builtin_bswap32(x); }

__device__ inline WalkResult walk_row(const kslam_overlap &o, const uint32_t *__restrict__ pool,
                                      const uint8_t *__restrict__ rbases, const uint8_t *__restrict__ rqual,
                                      const uint64_t *__restrict__ roff, const uint8_t *__restrict__ gbases,
                                      const uint64_t *__restrict__ goff, const WalkLds &S, uint8_t *md_dst, uint32_t md_cap) {
  WalkResult res{0.0, 0u, 0u, 0u};
  if (o.cigar_len == 0) return res;
  const uint64_t rb = roff[o.read];
  const int64_t L = (int64_t)(roff[o.read + 1] - rb);
  const uint64_t gb = goff[o.entry];
  const int64_t ref_len = (int64_t)(goff[o.entry + 1] - gb);
  const uint8_t *ref = gbases + gb;
  const bool rc = o.revcomp != 0;
  const uint8_t *lut = S.lut + (rc ? 256 : 0);
  MdOut w;
  w.dst = md_dst;
  w.cap = md_cap;
  int64_t rp = o.ref_begin, qp = o.query_begin > 0 ? o.query_begin : 0;
  double logp = 0.0;
  uint32_t nm = 0, worst = 0;
  // ONE loop over 16-column chunks for the whole row, whatever CIGAR operation a chunk belongs to: the lanes of
  // a wave walk different CIGARs, and a loop per operation made the wave run every operation slot's chunk loop
  // to the longest run any lane has there (~20 chunk rounds per wave for ~10 chunks per row).  A lane that has
  // used up its M run fetches operations until the next one (I and D are handled on the way).
  uint32_t k = 0, m_len = 0, i0 = 0, run = 0;
  int64_t at = 0;
  for (;;) {
    if (i0 >= m_len) {
      bool got = false;
      while (k < o.cigar_len) {
        const uint32_t c = pool[o.cigar_off + k], len = c >> 4, op = c & 15u;
        k++;
        if (op == 0) {
          if (rp < 0 || qp < 0 || rp + (int64_t)len > ref_len || qp + (int64_t)len > L) {
            res.flags |= 2u;
            k = o.cigar_len;
            break;
          }
          if (len) {
            at = rc ? L - 1 - qp : qp;         // index of column 0's base in the read
            run = 0;
            m_len = len;
            i0 = 0;
            got = true;
            break;
          }
        } else if (op == 1) {
          nm += len;
          qp += len;
        } else if (op == 2) {
          if (rp < 0 || rp + (int64_t)len > ref_len) {
            res.flags |= 2u;
            k = o.cigar_len;
            break;
          }
          w.flush();
          w.put('^');
          for (uint32_t i = 0; i < len; i++) w.put(ref[rp + i]);
          w.after_del = true;
          rp += len;
          nm += len;
        }
      }
      if (!got) break;
    }
    {
      const uint32_t len = m_len;
      {
        const uint32_t nn = min(16u, len - i0);
        const Bytes16 R = *reinterpret_cast<const Bytes16 *>(ref + rp + i0);
        // forward: bytes at .. at + 15; reverse: the 16 bytes ENDING at `at - i0`, read back to front
        const int64_t first = rc ? (int64_t)rb + at - (int64_t)i0 - 15 : (int64_t)rb + at + (int64_t)i0;
        Bytes16 B, Q;
        if (first >= 0) {
          B = *reinterpret_cast<const Bytes16 *>(rbases + first);
          Q = *reinterpret_cast<const Bytes16 *>(rqual + first);
        } else {            // (reverse strand at the very start of the batch's first read: stay inside the array)
#pragma unroll
          for (int x = 0; x < 4; x++) B.w[x] = Q.w[x] = 0;
          for (int j = 0; j < 16; j++)
            if (first + j >= 0) {
              B.w[j >> 2] |= (uint32_t)rbases[first + j] << (8 * (j & 3));
              Q.w[j >> 2] |= (uint32_t)rqual[first + j] << (8 * (j & 3));
            }
        }
        // reverse strand: byte j of the chunk is byte 15 - j of what was loaded
        {
          const uint32_t b0 = bswap_(B.w[3]), b1 = bswap_(B.w[2]), b2 = bswap_(B.w[1]), b3 = bswap_(B.w[0]);
          const uint32_t q0 = bswap_(Q.w[3]), q1 = bswap_(Q.w[2]), q2 = bswap_(Q.w[1]), q3 = bswap_(Q.w[0]);
          B.w[0] = rc ? b0 : B.w[0]; B.w[1] = rc ? b1 : B.w[1]; B.w[2] = rc ? b2 : B.w[2]; B.w[3] = rc ? b3 : B.w[3];
          Q.w[0] = rc ? q0 : Q.w[0]; Q.w[1] = rc ? q1 : Q.w[1]; Q.w[2] = rc ? q2 : Q.w[2]; Q.w[3] = rc ? q3 : Q.w[3];
        }
        // Columns first, without a branch: every lane runs all 16 columns of the chunk, the ones past the end
        // of its run add tab[TAB_ZERO] = 0.0 and set no bit (the walk had 66 VALU instructions and 25 scalar
        // ones per column when the columns were predicated and both strands' byte picks computed).
        // The probability chain's additions stay in column order.  The MD bookkeeping then runs once per
        // MISMATCH, not once per column.
        uint32_t miss = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const bool valid = (uint32_t)j < nn;
          const uint32_t r = byte_of(R, j);
          const uint32_t qc = lut[byte_of(B, j)];
          const uint32_t qq = min(byte_of(Q, j) - 33u, Q_CLAMP);     // (below 33 wraps around: clamped too)
          worst = max(worst, valid ? qq : 0u);
          const bool mm = valid && r != qc;
          miss |= (mm ? 1u : 0u) << j;
          const uint32_t idx = valid ? qq + (mm ? TAB_MM : 0u) : TAB_ZERO;
          logp += S.tab[idx];
        }
        uint32_t from = 0;
        while (miss) {
          const uint32_t kk = (uint32_t)__builtin_ctz(miss);
          const uint32_t word = kk < 4 ? R.w[0] : (kk < 8 ? R.w[1] : (kk < 12 ? R.w[2] : R.w[3]));
          run += kk - from;
          nm++;
          w.matches(run);
          w.mismatch((word >> (8 * (kk & 3u))) & 0xFFu);
          run = 0;
          from = kk + 1;
          miss &= miss - 1;
        }
        run += nn - from;
      }
      i0 += 16;
      if (i0 >= len) {      // the run is done
        w.matches(run);
        rp += len;
        qp += len;
      }
    }
  }
  w.flush();
  res.flags |= worst == Q_CLAMP ? 1u : 0u;
  res.logp = logp;
  res.nm = nm;
  res.md_len = w.n;
  return res;
}

__global__ __launch_bounds__(256) void k_row_details(const kslam_overlap *__restrict__ ov, uint64_t n,
                                                     const uint32_t *__restrict__ pool, const uint8_t *__restrict__ rbases,
                                                     const uint8_t *__restrict__ rqual, const uint64_t *__restrict__ roff,
                                                     const uint8_t *__restrict__ gbases, const uint64_t *__restrict__ goff,
                                                     const double *__restrict__ tables, kslam_row_detail *__restrict__ out,
                                                     uint32_t *__restrict__ md_lens, uint8_t *__restrict__ slots,
                                                     uint32_t *__restrict__ flags_or, const uint32_t *__restrict__ rows) {
  __shared__ WalkLds S;
  fill_walk_lds(S, tables);
  const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n) return;
  const uint64_t i = rows ? rows[x] : x;      // a list of rows, or all of them
  const kslam_overlap o = ov[i];
  const WalkResult r = walk_row(o, pool, rbases, rqual, roff, gbases, goff, S, slots + i * MD_SLOT, MD_SLOT);
  kslam_row_detail d;
  d.logp = r.logp;
  d.md_off = 0;
  d.md_len = r.md_len;
  d.nm = r.nm;
  d.flags = r.flags;
  d.pad = 0;
  out[i] = d;
  md_lens[i] = r.md_len;
  if (r.flags) atomicOr(flags_or, r.flags);
}

__global__ __launch_bounds__(256) void k_md_gather(const kslam_overlap *__restrict__ ov, uint64_t n,
                                                   const uint32_t *__restrict__ pool, const uint8_t *__restrict__ rbases,
                                                   const uint8_t *__restrict__ rqual, const uint64_t *__restrict__ roff,
                                                   const uint8_t *__restrict__ gbases, const uint64_t *__restrict__ goff,
                                                   const double *__restrict__ tables, kslam_row_detail *__restrict__ out,
                                                   const uint64_t *__restrict__ md_off, const uint8_t *__restrict__ slots,
                                                   uint8_t *__restrict__ md_pool, const uint32_t *__restrict__ rows) {
  __shared__ WalkLds S;
  fill_walk_lds(S, tables);
  const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n) return;
  const uint64_t i = rows ? rows[x] : x;
  const uint32_t len = out[i].md_len;
  const uint64_t off = md_off[i];
  out[i].md_off = off;
  if (len <= MD_SLOT) {
    const uint8_t *src = slots + i * MD_SLOT;
    for (uint32_t j = 0; j < len; j++) md_pool[off + j] = src[j];
  } else {   // did not fit its slot: once more, straight into the pool
    const kslam_overlap o = ov[i];
    (void)walk_row(o, pool, rbases, rqual, roff, gbases, goff, S, md_pool + off, len);
  }
}

}  // namespace

// ---- bases / quality columns cut out of uploaded FASTQ text (kslam_submit_batch_fastq) ----------------
namespace {
// 16 lanes per read: lane j moves bytes [16 j, 16 j + 16) of the read's bases line and of its quality
// line from the text into the two columns (unaligned 16-byte loads and stores; the last piece byte-wise)
__global__ __launch_bounds__(256) void k_gather_fields(const uint8_t *__restrict__ text, const uint64_t *__restrict__ bases_at,
                                                       const uint64_t *__restrict__ quality_at,
                                                       const uint64_t *__restrict__ off, uint64_t n_reads,
                                                       uint8_t *__restrict__ bases, uint8_t *__restrict__ quality) {
  const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  if (i >= n_reads) return;
  const uint32_t j = threadIdx.x & 15u;
  const uint64_t o = off[i], len = off[i + 1] - o;
  const uint8_t *sb = text + bases_at[i], *sq = text + quality_at[i];
  for (uint64_t at = 16ull * j; at < len; at += 256) {
    if (at + 16 <= len) {
      *reinterpret_cast<Bytes16 *>(bases + o + at) = *reinterpret_cast<const Bytes16 *>(sb + at);
      *reinterpret_cast<Bytes16 *>(quality + o + at) = *reinterpret_cast<const Bytes16 *>(sq + at);
    } else {
      for (uint64_t b = at; b < len; b++) {
        bases[o + b] = sb[b];
        quality[o + b] = sq[b];
      }
    }
  }
}
}  // namespace

void gather_fields(const uint8_t *d_text, const uint64_t *d_bases_at, const uint64_t *d_quality_at, const uint64_t *d_off,
                   uint64_t n_reads, uint8_t *d_bases, uint8_t *d_quality, hipStream_t s) {
  if (n_reads == 0) return;
  const uint64_t threads = n_reads * 16;
  hipLaunchKernelGGL(k_gather_fields, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, d_text, d_bases_at,
                     d_quality_at, d_off, n_reads, d_bases, d_quality);
  HIPCHK(hipGetLastError());
}

void row_details(const kslam_overlap *d_ov, uint64_t n, const uint32_t *d_pool, const uint8_t *d_rbases,
                 const uint8_t *d_rqual, const uint64_t *d_roff, const uint8_t *d_gbases, const uint64_t *d_goff,
                 const double *d_tables, kslam_row_detail *d_out, DetailWork &W, uint8_t **d_md_pool_out,
                 uint64_t *n_md_out, uint32_t *flags_out, hipStream_t s, const uint32_t *d_rows, uint64_t n_list) {
  *n_md_out = 0;
  *flags_out = 0;
  *d_md_pool_out = nullptr;
  if (n == 0) return;
  W.lens.ensure((n + 1) * sizeof(uint32_t));
  W.off.ensure((n + 1) * sizeof(uint64_t));
  W.slots.ensure(n * MD_SLOT + 64);
  W.scan_tmp.ensure(scan_tmp_bytes(n));
  W.totals.ensure(4 * sizeof(uint64_t));
  uint64_t *d_tot = W.totals.as<uint64_t>();
  HIPCHK(hipMemsetAsync(d_tot, 0, 4 * sizeof(uint64_t), s));
  const uint64_t m = d_rows ? n_list : n;      // rows walked
  if (d_rows) {   // the rows that are not on the list: zero records, no MD text
    HIPCHK(hipMemsetAsync(d_out, 0, n * sizeof(kslam_row_detail), s));
    HIPCHK(hipMemsetAsync(W.lens.p, 0, n * sizeof(uint32_t), s));
  }
  const unsigned nb = (unsigned)((m + 255) / 256);
  if (m)
    hipLaunchKernelGGL(k_row_details, dim3(nb), dim3(256), 0, s, d_ov, m, d_pool, d_rbases, d_rqual, d_roff, d_gbases, d_goff,
                       d_tables, d_out, W.lens.as<uint32_t>(), W.slots.as<uint8_t>(), reinterpret_cast<uint32_t *>(d_tot + 1), d_rows);
  exclusive_scan_u32_to_u64(W.lens.as<uint32_t>(), W.off.as<uint64_t>(), n, d_tot, W.scan_tmp.p, s);
  uint64_t h[2] = {0, 0};
  read_back(h, d_tot, sizeof h, s);
  W.md_pool.ensure(h[0] + 64);
  if (m)
    hipLaunchKernelGGL(k_md_gather, dim3(nb), dim3(256), 0, s, d_ov, m, d_pool, d_rbases, d_rqual, d_roff, d_gbases, d_goff,
                       d_tables, d_out, W.off.as<uint64_t>(), W.slots.as<uint8_t>(), W.md_pool.as<uint8_t>(), d_rows);
  HIPCHK(hipGetLastError());
  *d_md_pool_out = W.md_pool.as<uint8_t>();
  *n_md_out = h[0];
  *flags_out = (uint32_t)h[1];
}

}  // namespace kslam
