// filter.hip -- read k-mer extraction fused with a membership test against the genome k-mer set.
//
// The reference writes every read k-mer into the one list it sorts (src/KMer.h:373-381, src/SLAM.h:63-66)
// although a read k-mer without an equal genome k-mer can never produce an overlap: processPileUp
// skips every run that does not start with a genome record (src/Overlap.h:157-162).  The genome list is
// sampled at every 16th offset, so of a read's 119 k-mers only the ~7 in phase with the sampling can
// match at all -- on the BASELINE workload 97 % of the 238 M read k-mers per batch were written
// (3.8 GB), sorted and looked up for nothing.  Here the extraction kernel asks a filter built once per
// index ("is this k-mer possibly in the genome list?") and keeps only the survivors.  False positives
// are harmless (the join finds no run for them), false negatives cannot happen, so the overlaps are
// exactly the reference's.
//
// Filter layout (MI355X): a blocked Bloom filter of 2^b bits in 128-byte lines of eight 16-byte
// pieces.  A k-mer's LINE is chosen by its canonical minimizer (the smallest 16-mer over both strands of
// the 32-mer), its piece and its four bits (one per dword of the piece) by a hash of the whole k-mer.
// Consecutive k-mers of a read share their minimizer for ~9 positions on average, so the 119 probes of a
// read touch ~14 distinct lines instead of 119: the probe traffic drops from 238 M to ~28 M line fetches
// per batch, and each probe is ONE 16-byte load.  512 MiB (b = 32) holds the 312 M keys of the 5 Gb
// database at 9.3 keys per piece: false-positive rate ~0.5 %.
//
// Survivors are staged per workgroup in LDS and appended to the output with one atomic; their order
// is scheduling dependent, which no later stage observes (they are sorted / looked up individually and
// the overlap list is re-sorted by (read, entry, rel)).
#include "common.h"

namespace kslam {

namespace {

__device__ inline uint64_t revcomp64(uint64_t fwd) {
  uint64_t x = fwd ^ 0xAAAAAAAAAAAAAAAAull;
  x = __brevll(x);
  return ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);
}

// 4 ASCII bytes (little endian dword) -> 8 bits, first base in bits 7:6 (A 0, C 1, T 2, G 3, else 0:
// reference src/KMer.h:246-268)
__device__ inline uint32_t pack4(uint32_t x) {
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t c = (x >> (8 * j)) & 0xFFu;
    uint32_t code = (c >> 1) & 3u;
    uint32_t cand = (0x47544341u >> (8 * code)) & 0xFFu;
    code = (cand == c) ? code : 0u;
    r |= code << (6 - 2 * j);
  }
  return r;
}

// smallest 32-bit window (16 bases) of the 64-bit word, over all 17 base-aligned positions
__device__ inline uint32_t min_window16(uint64_t v) {
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  uint32_t m = min(lo, hi);
#pragma unroll
  for (int j = 1; j < 16; j += 2) {
    const uint32_t a = __builtin_amdgcn_alignbit(hi, lo, 2 * j);
    const uint32_t b = j + 1 < 16 ? __builtin_amdgcn_alignbit(hi, lo, 2 * (j + 1)) : a;
    m = min(m, min(a, b));
  }
  return m;
}

struct Probe {
  uint32_t piece;          // index of the 16-byte piece
  uint32_t s0, s1, s2, s3; // bit number inside each dword of the piece
};

// `kmer` and `rc` are the two strands of one 32-mer (either order): the result is the same for both.  `mini`: the k-mer's
// canonical minimizer = the smallest 16-mer over both strands = min(min_window16(kmer), min_window16(rc))
__device__ inline Probe probe_with_minimizer(uint64_t kmer, uint64_t rc, uint32_t mini, uint32_t line_bits) {
  const uint64_t canon = kmer < rc ? kmer : rc;
  const uint32_t h = ((uint32_t)canon * 0x9E3779B1u) ^ ((uint32_t)(canon >> 32) * 0x85EBCA77u);
  const uint32_t g = h ^ (h >> 15);
  Probe p;
  const uint32_t line = (mini * 0x9E3779B1u) >> (32 - line_bits);
  p.piece = (line << 3) | (g >> 29);
  p.s0 = g & 31u; p.s1 = (g >> 5) & 31u; p.s2 = (g >> 10) & 31u; p.s3 = (g >> 15) & 31u;
  return p;
}
__device__ inline Probe probe_of(uint64_t kmer, uint64_t rc, uint32_t line_bits) {
  return probe_with_minimizer(kmer, rc, min(min_window16(kmer), min_window16(rc)), line_bits);
}
// reverse complement of a 16-mer held as 32 bits (the 64-bit form: revcomp64)
__device__ inline uint32_t revcomp32(uint32_t f) {
  uint32_t x = __brev(f ^ 0xAAAAAAAAu);
  return ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
}
// min over the lanes of the 16-lane row at and below / at and above this one (DPP row shifts; a lane without a source takes the identity)
// (`old` = the identity of min, so that the compiler can fold the shift into the v_min_u32 itself)
__device__ inline uint32_t row_prefix_min(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xF, 0xF, false));   // row_shr:1
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xF, 0xF, false));
  return v;
}
__device__ inline uint32_t row_suffix_min(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x101, 0xF, 0xF, false));   // row_shl:1
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x102, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x104, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x108, 0xF, 0xF, false));
  return v;
}

__global__ void k_filter_build(const uint64_t *__restrict__ keys, uint32_t n, uint32_t line_bits,
                               uint32_t *__restrict__ filter) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys[i];
  if (k == 0) return;                       // k-mer 0 never joins (src/Overlap.h:236)
  if (i > 0 && keys[i - 1] == k) return;    // sorted list: one insert per distinct key
  const Probe p = probe_of(k, revcomp64(k), line_bits);
  uint32_t *w = filter + (size_t)p.piece * 4;
  atomicOr(w + 0, 1u << p.s0);
  atomicOr(w + 1, 1u << p.s1);
  atomicOr(w + 2, 1u << p.s2);
  atomicOr(w + 3, 1u << p.s3);
}

// ---- the build since round 6: the keys' probes ordered by filter block, each block assembled in LDS ----------------
// k_filter_build above is 4 scattered read-modify-writes per key over the whole filter (312 M keys, 512 MB: 45 ms, 58 %
// of kslam_set_index).  Instead: (1) k_filter_words turns every distinct non-zero key into its probe, one 64-bit word
// `piece << 20 | s3 s2 s1 s0` (a key that is zero or repeats its left neighbour becomes a word beyond every block);
// (2) the words go through the radix passes that cover the bits above a BLOCK of FBLK pieces (32 KB of filter: two 8-bit
// passes for the 5 Gb database); (3) k_filter_fill: one workgroup per block ORs its words into 32 KB of LDS and stores
// the block with 16-byte coalesced writes -- every filter byte written once, no read-modify-write in HBM.
constexpr uint32_t FBLK_BITS = 11, FBLK = 1u << FBLK_BITS;     // 2048 pieces of 16 bytes
__global__ __launch_bounds__(256) void k_filter_words(const uint64_t *__restrict__ keys, uint32_t n, uint32_t line_bits,
                                                      uint64_t *__restrict__ words) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys[i];
  uint64_t w = 1ull << (20 + 3 + line_bits);    // beyond the last piece: sorts behind every block, belongs to none
  if (k != 0 && !(i > 0 && keys[i - 1] == k)) {  // k-mer 0 never joins (src/Overlap.h:236); one insert per distinct key
    const Probe p = probe_of(k, revcomp64(k), line_bits);
    w = ((uint64_t)p.piece << 20) | ((uint64_t)p.s3 << 15) | (p.s2 << 10) | (p.s1 << 5) | p.s0;
  }
  words[i] = w;
}
// The three tables of the index from ONE pass over the sorted genome records (round 6, second half): the key column and the
// {meta, offset} column (what k_split_soa wrote), the bucket table over the top key bits (what join.hip's k_bucket computed from
// the key column: lower bounds, written where the bucket number changes), the key's probe word (k_filter_words) and the first
// digit byte of the probe words' sort.  The three kernels each streamed the 312 M keys again.
__global__ __launch_bounds__(256) void k_split_tables(const uint4 *__restrict__ recs, uint32_t n, uint64_t *__restrict__ key,
                                                      uint2 *__restrict__ mo, uint32_t bucket_shift, uint32_t nb,
                                                      uint32_t *__restrict__ bucket, uint32_t line_bits, uint64_t *__restrict__ words,
                                                      uint8_t *__restrict__ digits, uint32_t digit_shift) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4 r = recs[i];
  const uint64_t k = ((uint64_t)r.y << 32) | r.x;
  bool have_prev = i > 0;
  uint64_t kp = 0;
  if (have_prev) {
    const uint2 p = *reinterpret_cast<const uint2 *>(recs + (i - 1));   // the left neighbour's key (same cache line three times in four)
    kp = ((uint64_t)p.y << 32) | p.x;
  }
  key[i] = k;
  mo[i] = make_uint2(r.z, r.w);
  {   // bucket[x] = first index whose key's top bits are >= x
    const uint32_t b = (uint32_t)(k >> bucket_shift);
    const int64_t bp = have_prev ? (int64_t)(kp >> bucket_shift) : -1;
    for (int64_t x = bp + 1; x <= (int64_t)b; x++) bucket[x] = i;
    if (i == n - 1)
      for (uint32_t x = b + 1; x <= nb; x++) bucket[x] = n;
  }
  uint64_t w = 1ull << (20 + 3 + line_bits);
  if (k != 0 && !(have_prev && kp == k)) {
    const Probe p = probe_of(k, revcomp64(k), line_bits);
    w = ((uint64_t)p.piece << 20) | ((uint64_t)p.s3 << 15) | (p.s2 << 10) | (p.s1 << 5) | p.s0;
  }
  words[i] = w;
  digits[i] = (uint8_t)((w >> digit_shift) & 0xFFu);
}

// block_start[b] = index of the first word of block b (words ordered by block; join.hip's offsets kernel)
__global__ __launch_bounds__(256) void k_filter_fill(const uint64_t *__restrict__ words, const uint32_t *__restrict__ block_start,
                                                     uint4 *__restrict__ filter) {
  __shared__ uint32_t blk[FBLK * 4];
  for (uint32_t x = threadIdx.x; x < FBLK * 4; x += 256) blk[x] = 0;
  __syncthreads();
  const uint32_t lo = block_start[blockIdx.x], hi = block_start[blockIdx.x + 1];
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
    const uint64_t w = words[i];
    uint32_t *q = blk + (((uint32_t)(w >> 20)) & (FBLK - 1u)) * 4;
    const uint32_t g = (uint32_t)w;
    atomicOr(q + 0, 1u << (g & 31u));
    atomicOr(q + 1, 1u << ((g >> 5) & 31u));
    atomicOr(q + 2, 1u << ((g >> 10) & 31u));
    atomicOr(q + 3, 1u << ((g >> 15) & 31u));
  }
  __syncthreads();
  uint4 *dst = filter + (size_t)blockIdx.x * FBLK;
  for (uint32_t x = threadIdx.x; x < FBLK; x += 256) dst[x] = reinterpret_cast<const uint4 *>(blk)[x];
}

constexpr int FW = 8;              // waves per workgroup
constexpr int STAGE = 2048;        // survivor records staged per workgroup (128 reads x ~12.6 on the bench workload)
constexpr uint32_t LWORDS = 36;    // packed-base words per wave: reads up to 511 bases + alignment slack
constexpr uint32_t RPW = 16;       // reads per wave (their offsets are fetched with one load)

struct Cut {        // one k-mer of the read, cut out of the packed words
  uint64_t fwd, rc;
  bool valid;
};

// One wave per read, RPW reads in turn.  Record layout and canonical choice as k_extract
// (extract.hip): the survivors are bit-identical to the records the unfiltered kernel writes.
// Not bound by bytes (150 B in, ~12 survivors out per read) but by round trips: the wave's read
// offsets come with one load up front, the bases of read r+1 are in flight while read r is
// processed, the filter probes of 128 k-mers (two per lane) are issued together -- and the output
// cursor is touched ONCE per workgroup: a first version appended per wave (250 k returning atomic adds
// on one address per batch) and ran at exactly 12 ns per atomic, 3.1 ms, whatever else it did.
__global__ __launch_bounds__(FW * 64) void k_extract_filter(const uint8_t *__restrict__ bases,
                                                            const uint64_t *__restrict__ off, uint32_t n_reads,
                                                            const uint4 *__restrict__ filter, uint32_t line_bits,
                                                            uint4 *__restrict__ out,
                                                            unsigned long long *__restrict__ cursor, uint64_t cap,
                                                            uint32_t ablate, uint8_t *__restrict__ digits,
                                                            uint32_t digit_word, uint32_t digit_shift) {
#ifdef KSLAM_ABLATE
#define KSLAM_FILTER_ABLATED(bit) ((ablate & (bit)) != 0)   // measurement-only build: parts of the kernel switched off
#else
#define KSLAM_FILTER_ABLATED(bit) false
#endif
  __shared__ uint32_t packed[FW][LWORDS];
  __shared__ uint4 stage[STAGE];
  __shared__ uint32_t staged;              // slots handed out (may count past STAGE)
  __shared__ uint32_t stage_valid;         // first slot that was handed out but not written (stage full)
  __shared__ unsigned long long wg_base;
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t *my = packed[w];
  uint8_t *my8 = reinterpret_cast<uint8_t *>(my);
  if (threadIdx.x == 0) { staged = 0; stage_valid = STAGE; }
  __syncthreads();
  const uint32_t r_begin = (blockIdx.x * FW + w) * RPW;
  const uint32_t r_cnt = r_begin < n_reads ? min(n_reads - r_begin, RPW) : 0u;
  const uint64_t my_off = r_cnt ? off[r_begin + min(lane, r_cnt)] : 0ull;   // lanes 0..r_cnt: this wave's read offsets

  // the first radix pass's digit of a record (the sort's first histogram then reads these bytes, not the records)
  auto digit_of_rec = [&](const uint4 &r) -> uint8_t { return (uint8_t)(((digit_word ? r.y : r.x) >> digit_shift) & 0xFFu); };
  auto offset_of = [&](uint32_t i) -> uint64_t {   // i is wave-uniform
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)my_off, i);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(my_off >> 32), i);
    return ((uint64_t)hi << 32) | lo;
  };
  // the dwords of one read (up to 3 per lane: 511 bases + 3 of misalignment = 129 dwords)
  uint32_t nx0 = 0, nx1 = 0, nx2 = 0;
  auto fetch = [&](uint32_t i) {
    const uint64_t s0 = offset_of(i);
    const uint32_t len = (uint32_t)(offset_of(i + 1) - s0);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(bases + (s0 & ~3ull));
    const uint32_t ndw = ((uint32_t)(s0 & 3ull) + len + 3) >> 2;
    if (lane < ndw) nx0 = src[lane];
    if (lane + 64 < ndw) nx1 = src[lane + 64];
    if (lane + 128 < ndw) nx2 = src[lane + 128];
  };
  if (r_cnt) fetch(0);

  // the workgroup's stage goes out (one atomic) and is reset; called by every thread at the same point
  auto flush_stage = [&]() {
    __syncthreads();
    const uint32_t n_st = min(staged, stage_valid);
    if (threadIdx.x == 0) wg_base = n_st ? atomicAdd(cursor, (unsigned long long)n_st) : 0ull;
    __syncthreads();
    const unsigned long long gb = wg_base;
    if (gb + n_st <= cap)
      for (uint32_t k = threadIdx.x; k < n_st; k += FW * 64) {
        const uint4 r = stage[k];
        out[gb + k] = r;
        if (digits) digits[gb + k] = digit_of_rec(r);
      }
    __syncthreads();
    if (threadIdx.x == 0) { staged = 0; stage_valid = STAGE; }
    __syncthreads();
  };

  for (uint32_t i = 0; i < RPW; i++) {
    // Every wave of the workgroup takes its i-th read in the same trip, so that the stage can be emptied
    // between trips when it is half full: reads with many genome k-mers (long reads, repeats) then cost
    // one more atomic per workgroup and trip instead of one per wave and 64 k-mers -- 250-bp reads ran
    // 26 x slower through that overflow path (28.9 ms) than they do now.
    if (i) {
      // The decision must be the same in every wave: all of them read `staged` between two barriers, so
      // that no wave is already adding to it (keep_record of trip i) while another has yet to read it --
      // waves that disagreed would pair the barriers of flush_stage with the wrong ones.
      __syncthreads();
      const bool half_full = staged >= STAGE / 2;
      __syncthreads();
      if (half_full) flush_stage();
    }
    if (i >= r_cnt) continue;
    const uint64_t s0 = offset_of(i);
    const uint32_t len = (uint32_t)(offset_of(i + 1) - s0);
    const uint32_t m = (uint32_t)(s0 & 3ull);
    const uint32_t ndw = (m + len + 3) >> 2;
    __builtin_amdgcn_wave_barrier();       // the previous read's words are no longer needed
    if (lane < ndw) my8[(lane & ~3u) | (3u - (lane & 3u))] = (uint8_t)pack4(nx0);
    if (lane + 64 < ndw) my8[((lane + 64) & ~3u) | (3u - (lane & 3u))] = (uint8_t)pack4(nx1);
    if (lane + 128 < ndw) my8[((lane + 128) & ~3u) | (3u - (lane & 3u))] = (uint8_t)pack4(nx2);
    if (i + 1 < r_cnt) fetch(i + 1);       // in flight while this read is processed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (len < KSLAM_K) continue;           // src/KMer.h:167
    const uint32_t nk = len - KSLAM_K + 1; // gap 1 (src/KMer.h:378)
    const uint32_t idbits = (r_begin + i) & 0x3FFFFFFFu;

    auto keep_record = [&](const Cut &c, uint32_t q, bool pass) {
      const bool is_fwd = c.fwd < c.rc;    // src/KMer.h:173 (palindromes take the rc branch)
      const uint64_t kmer = is_fwd ? c.fwd : c.rc;
      const bool keep = c.valid && pass && kmer != 0;   // k-mer 0 never joins (src/Overlap.h:236)
      const unsigned long long mask = __ballot(keep);
      if (mask == 0) return;               // (wave-uniform)
      const uint32_t cnt = (uint32_t)__popcll(mask);
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&staged, cnt);   // LDS
      base = __builtin_amdgcn_readfirstlane(base);
      uint4 rec;
      rec.x = (uint32_t)kmer; rec.y = (uint32_t)(kmer >> 32);
      rec.z = is_fwd ? idbits : (idbits | (1u << 30));
      rec.w = is_fwd ? q : (len - KSLAM_K - q);   // src/KMer.h:176
      const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      if (base + cnt <= STAGE) {
        if (keep) stage[slot] = rec;
      } else {
        // the workgroup's stage is full (reads with many genome k-mers): this wave's records go out
        // with an atomic of their own, and so will everything handed out after them
        if (lane == 0) atomicMin(&stage_valid, base);
        unsigned long long gb = 0;
        if (lane == 0) gb = atomicAdd(cursor, (unsigned long long)cnt);
        gb = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(gb >> 32)) << 32) |
             __builtin_amdgcn_readfirstlane((uint32_t)gb);
        if (keep && gb + cnt <= cap) {
          out[gb + (slot - base)] = rec;
          if (digits) digits[gb + (slot - base)] = digit_of_rec(rec);
        }
      }
    };
    // The minimizers of a chunk of 128 k-mers from the read's 16-MERS (round 6).  A k-mer's canonical minimizer is the smallest of
    // its 17 windows over both strands, and window j of k-mer q is the 16-mer at read position q + j whichever k-mer looks at
    // it: so c(p) = min(16-mer at p, its reverse complement) is computed ONCE per position (a lane takes positions q0 + lane,
    // + 64, and the first 16 lanes + 128), and k-mer q's minimizer is the minimum of c over [q, q + 16] -- by van Herk's
    // blocks: positions fall into the wave's rows of 16 lanes, and 17 consecutive ones are the suffix of one row from q on
    // plus the prefix of the next row up to q + 16 (the same lane of the next row).  Row prefix / suffix minima are four
    // DPP shifts each.  (probe_of computes the same value from the two 64-bit strands, 17 windows each: 66 operations per
    // k-mer, of which this keeps about 35 per PAIR of k-mers.)
    // ... and the k-mers themselves are two such 16-mers: fwd(q) = F(q) : F(q + 16), its reverse complement
    // R(q + 16) : R(q) with R = the 16-mer's own reverse complement -- the values c() was made of, one row up.
    struct Mer16 { uint32_t f, r, c; };
    auto mer16 = [&](uint32_t pos) -> Mer16 {
      Mer16 x{0u, 0u, 0xFFFFFFFFu};
      if (pos + 16u > len) return x;                        // beyond the read's last 16-mer: no k-mer's window
      const uint32_t sidx = m + pos, wi = sidx >> 4, sh = (sidx & 15u) * 2u;
      x.f = (uint32_t)((((((uint64_t)my[wi]) << 32) | my[wi + 1]) << sh) >> 32);
      x.r = revcomp32(x.f);
      x.c = min(x.f, x.r);
      return x;
    };
    for (uint32_t q0 = 0; q0 < nk; q0 += 128) {
      const uint32_t qa = q0 + lane, qb = q0 + 64 + lane;
      Cut ca, cb;
      uint32_t mini_a, mini_b;
      {
        const Mer16 x0 = mer16(qa), x1 = mer16(qb), x2 = lane < 16u ? mer16(q0 + 128u + lane) : Mer16{0u, 0u, 0xFFFFFFFFu};
        const uint32_t p0 = row_prefix_min(x0.c), p1 = row_prefix_min(x1.c), p2 = row_prefix_min(x2.c);
        const int up = (int)((lane + 16u) & 63u);             // the same lane of the next row (of the next register for the last row)
        const bool same = lane < 48u;
        auto up_a = [&](uint32_t v0, uint32_t v1) { const uint32_t a = (uint32_t)__shfl((int)v0, up, 64), b = (uint32_t)__shfl((int)v1, up, 64); return same ? a : b; };
        mini_a = min(row_suffix_min(x0.c), up_a(p0, p1));
        mini_b = min(row_suffix_min(x1.c), up_a(p1, p2));
        ca.valid = qa < nk;
        cb.valid = qb < nk;
        ca.fwd = ((uint64_t)x0.f << 32) | up_a(x0.f, x1.f);
        ca.rc = ((uint64_t)up_a(x0.r, x1.r) << 32) | x0.r;
        cb.fwd = ((uint64_t)x1.f << 32) | up_a(x1.f, x2.f);
        cb.rc = ((uint64_t)up_a(x1.r, x2.r) << 32) | x1.r;
      }
      Probe pa, pb;
      if (KSLAM_FILTER_ABLATED(2u)) {   // measurement only: no minimizer (a pseudo-random line per k-mer)
        pa.piece = (uint32_t)((ca.fwd * 0x9E3779B97F4A7C15ull) >> (64 - line_bits - 3)); pa.s0 = pa.s1 = pa.s2 = pa.s3 = (uint32_t)ca.rc & 31u;
        pb.piece = (uint32_t)((cb.fwd * 0x9E3779B97F4A7C15ull) >> (64 - line_bits - 3)); pb.s0 = pb.s1 = pb.s2 = pb.s3 = (uint32_t)cb.rc & 31u;
      } else {
        pa = probe_with_minimizer(ca.fwd, ca.rc, mini_a, line_bits); pb = probe_with_minimizer(cb.fwd, cb.rc, mini_b, line_bits);
      }
      uint4 fa, fb;
      if (KSLAM_FILTER_ABLATED(1u)) {   // measurement only: no probe load
        fa = make_uint4(pa.piece, pa.piece >> 3, pa.piece >> 5, 0x11111111u); fb = make_uint4(pb.piece, pb.piece >> 2, pb.piece >> 7, 0x11111111u);
      } else {
        fa = filter[pa.piece];
        fb = filter[pb.piece];   // (a lane beyond the read probes k-mer 0 of it: in bounds, ignored)
      }
      keep_record(ca, qa, ((fa.x >> pa.s0) & (fa.y >> pa.s1) & (fa.z >> pa.s2) & (fa.w >> pa.s3) & 1u) != 0);
      keep_record(cb, qb, ((fb.x >> pb.s0) & (fb.y >> pb.s1) & (fb.z >> pb.s2) & (fb.w >> pb.s3) & 1u) != 0);
    }
  }
  flush_stage();
#undef KSLAM_FILTER_ABLATED
}

}  // namespace

size_t filter_bytes(uint32_t log2_bits) { return (size_t)1 << (log2_bits - 3); }

void filter_build(const uint64_t *d_sorted_keys, uint32_t n, uint32_t log2_bits, void *d_filter, hipStream_t s) {
  HIPCHK(hipMemsetAsync(d_filter, 0, filter_bytes(log2_bits), s));
  if (n == 0) return;
  hipLaunchKernelGGL(k_filter_build, dim3((n + 255) / 256), dim3(256), 0, s, d_sorted_keys, n, log2_bits - 10,
                     (uint32_t *)d_filter);
  HIPCHK(hipGetLastError());
}

bool split_columns_and_tables(const void *d_sorted_recs, uint32_t n, uint64_t *d_key, void *d_meta_off, uint32_t bucket_bits, uint32_t *d_bucket,
                              uint32_t log2_bits, void *d_words, SortWorkspace &ws, hipStream_t s) {
  const uint32_t line_bits = log2_bits - 10, piece_bits = line_bits + 3;
  if (n == 0 || log2_bits < 20 || piece_bits < FBLK_BITS) return false;     // the caller takes the separate kernels
  ws.digits.ensure((size_t)n + 64);
  hipLaunchKernelGGL(k_split_tables, dim3((n + 255) / 256), dim3(256), 0, s, (const uint4 *)d_sorted_recs, n, d_key, (uint2 *)d_meta_off,
                     64 - bucket_bits, 1u << bucket_bits, d_bucket, line_bits, (uint64_t *)d_words, ws.digits.as<uint8_t>(), 20 + FBLK_BITS);
  HIPCHK(hipGetLastError());
  return true;
}

void filter_build_sorted(const uint64_t *d_sorted_keys, uint32_t n, uint32_t log2_bits, void *d_filter, void *d_words_a, void *d_words_b,
                         uint32_t *d_block_start, SortWorkspace &ws, hipStream_t s, bool words_ready) {
  const uint32_t line_bits = log2_bits - 10, piece_bits = line_bits + 3;
  if (n == 0 || piece_bits < FBLK_BITS) {         // (a filter smaller than one block: never chosen by kslam_set_index, fb >= 20)
    filter_build(d_sorted_keys, n, log2_bits, d_filter, s);
    return;
  }
  const uint32_t n_blocks = 1u << (piece_bits - FBLK_BITS);
  if (!words_ready) hipLaunchKernelGGL(k_filter_words, dim3((n + 255) / 256), dim3(256), 0, s, d_sorted_keys, n, line_bits, (uint64_t *)d_words_a);
  // radix passes over the block number and the "belongs to no block" bit above it: key bits [20 + FBLK_BITS, 20 + piece_bits]
  SortPass passes[8];
  int np = 0;
  for (uint32_t sh = 20 + FBLK_BITS; sh <= 20 + piece_bits; sh += 8) passes[np++] = SortPass{2u, sh, 0};
  const bool keep = ws.use_digit_bytes;
  ws.use_digit_bytes = true;
  ws.first_digits_ready = words_ready;      // (k_split_tables wrote the first pass's digit of every word)
  const uint64_t *words = (const uint64_t *)radix_sort(d_words_a, d_words_b, n, 2, passes, np, ws, s, nullptr, nullptr, nullptr, /*setup=*/true);
  ws.first_digits_ready = false;
  ws.use_digit_bytes = keep;
  build_offsets_table(words, n, 20 + FBLK_BITS, n_blocks + 1, d_block_start, s);   // [0, n_blocks]: starts; the words of no block follow
  hipLaunchKernelGGL(k_filter_fill, dim3(n_blocks), dim3(256), 0, s, words, d_block_start, (uint4 *)d_filter);
  HIPCHK(hipGetLastError());
}

void extract_filtered(const uint8_t *d_bases, const uint64_t *d_off, uint32_t n_reads, const void *d_filter,
                      uint32_t log2_bits, uint4 *d_out, uint64_t *d_cursor, uint64_t cap, const Tuning &tune, hipStream_t s,
                      uint8_t *d_digits, uint32_t digit_word, uint32_t digit_shift) {
  HIPCHK(hipMemsetAsync(d_cursor, 0, sizeof(uint64_t), s));
  if (n_reads == 0) return;
  const uint32_t per_block = FW * RPW;
  hipLaunchKernelGGL(k_extract_filter, dim3((n_reads + per_block - 1) / per_block), dim3(FW * 64), 0, s, d_bases, d_off,
                     n_reads, (const uint4 *)d_filter, log2_bits - 10, d_out,
                     reinterpret_cast<unsigned long long *>(d_cursor), cap,
#ifdef KSLAM_ABLATE
                     tune.filter_ablate,
#else
                     0u,
#endif
                     d_digits, digit_word, digit_shift);
  (void)tune;
  HIPCHK(hipGetLastError());
}

}  // namespace kslam
