// tail.cpp -- the host tail behind include/kslam_tail.h.
//
// What the reference does with vectors of objects that each carry two Overlap
// copies (src/PairedOverlap.h, src/SAM.h), this file does over the flat
// kslam_overlap array the hot path returns: alignment pairs are 32-byte records
// holding two indices, every stage is a parallel sweep over read-pair ranges on
// a persistent worker pool, and the SAM text is formatted straight into
// per-task buffers.  The observable result is the reference's:
//  * the pairing sort (src/PairedOverlap.h:247-257) is a merge of the R1 and R2
//    halves of the already (read, entry, rel)-sorted input, ties R1 first;
//  * wherever the reference calls std::sort with a partial key
//    (src/PairedOverlap.h:369,403,527, src/SAM.h:448) the same libstdc++
//    std::sort runs on the same element order with the same comparator: the
//    permutation an introsort produces depends only on comparison outcomes, so
//    sorting 32-byte records gives the order the reference gets;
//  * the insert-size statistics (src/PairedOverlap.h:314-360) are taken from a
//    radix-sorted copy, with the sums formed in integers (exact, hence equal to
//    the reference's sequential double accumulation while they stay < 2^53).
#include <algorithm>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kslam_tail.h"
#include <cerrno>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>

#include "workers.hpp"

namespace {

typedef kslam_paired_overlap Rec;
typedef kslam_read_pair Group;

using namespace kslam_host;
typedef HostError TailError;

int thread_count(const kslam_tail_params *p) {
  int n = p->threads > 0 ? p->threads : usable_cpus();
  return std::max(1, std::min(n, 512));
}


// ---------------------------------------------------------------- storage ------
// Uninitialised storage that is kept between calls.  A batch touches several
// hundred MB of intermediates; handing that back to the allocator and faulting
// fresh pages in again on every batch costs more than the work itself, so the
// arena only ever grows (kslam_tail_release_buffers() frees it).
template <typename T>
struct Buf {
  T *p = nullptr;
  size_t cap = 0;
  T *ensure(size_t n) {
    if (n > cap) {
      size_t want = std::max(n, cap + cap / 2);
      T *q = (T *)realloc((void *)p, want * sizeof(T));
      if (!q) fail(KSLAM_ERR_OOM, "out of host memory for the tail's work buffers");
      advise_huge(q, want * sizeof(T));
      p = q;
      cap = want;
    }
    return p;
  }
  void release() {
    free((void *)p);
    p = nullptr;
    cap = 0;
  }
};

struct Text {  // growable byte buffer (realloc: large blocks grow by mremap, no copy, no zero fill)
  char *p = nullptr;
  size_t cap = 0, n = 0;
  // a BORROWED block (kslam_sam_writer_enqueue: text another stage produced, e.g. the GPU formatter's page-locked copy):
  // never grown, handed back through ext_release instead of free()
  void (*ext_release)(void *user, void *data) = nullptr;
  void *ext_user = nullptr;
  Text() {}
  Text(const Text &) = delete;
  Text &operator=(const Text &) = delete;
  Text(Text &&o) noexcept : p(o.p), cap(o.cap), n(o.n), ext_release(o.ext_release), ext_user(o.ext_user) {
    o.p = nullptr;
    o.cap = o.n = 0;
    o.ext_release = nullptr;
  }
  void drop() {
    if (ext_release) {
      if (p) ext_release(ext_user, p);
    } else {
      free(p);
    }
    p = nullptr;
    cap = n = 0;
    ext_release = nullptr;
  }
  Text &operator=(Text &&o) noexcept {
    if (this != &o) {
      drop();
      p = o.p;
      cap = o.cap;
      n = o.n;
      ext_release = o.ext_release;
      ext_user = o.ext_user;
      o.p = nullptr;
      o.cap = o.n = 0;
      o.ext_release = nullptr;
    }
    return *this;
  }
  ~Text() { drop(); }
  void reserve(size_t want) {
    if (want <= cap) return;
    if (ext_release) fail(KSLAM_ERR_INTERNAL, "a borrowed text block cannot grow");
    char *q = (char *)realloc(p, want);
    if (!q) fail(KSLAM_ERR_OOM, "out of host memory for SAM text");
    advise_huge(q, want);
    p = q;
    cap = want;
  }
  char *need(size_t k) {
    if (n + k > cap) reserve(std::max(cap * 2, n + k + 4096));
    return p + n;
  }
  void release() { drop(); }
  void put(const char *s, size_t k) {
    memcpy(need(k), s, k);
    n += k;
  }
  void put(char c) {
    *need(1) = c;
    n++;
  }
  void lit(const char *s) { put(s, strlen(s)); }
  void num(uint64_t v) {
    char *d = need(20);
    if (v < 10) {   // CIGAR and MD numbers are mostly one to three digits
      d[0] = (char)('0' + v);
      n += 1;
      return;
    }
    char t[24];
    int k = 24;
    do {
      t[--k] = (char)('0' + v % 10);
      v /= 10;
    } while (v);
    memcpy(d, t + k, 24 - k);
    n += 24 - k;
  }
  void snum(int64_t v) {
    if (v < 0) {
      put('-');
      num((uint64_t)(-v));
    } else
      num((uint64_t)v);
  }
};

struct Span {
  int start, stop;
  uint32_t rec;
};

struct Arena {
  std::mutex call;  // one tail call at a time: each call spreads over all workers anyway
  Buf<Rec> paired;  // alignment pairs out of pairing: one region per task, gaps between regions
  Buf<Rec> screened;  // ... after the per-read-pair screens, same layout
  Buf<Group> groups_by_task, groups;
  Buf<int32_t> inserts_by_task, inserts, inserts_tmp;
  Buf<size_t> radix_hist;
  Buf<uint32_t> entry_hist;
  Buf<size_t> entry_start;
  Buf<Span> spans;
  std::vector<Text> sam_parts;
  void release() {
    paired.release();
    screened.release();
    groups_by_task.release();
    groups.release();
    inserts_by_task.release();
    inserts.release();
    inserts_tmp.release();
    radix_hist.release();
    entry_hist.release();
    entry_start.release();
    spans.release();
    std::vector<Text>().swap(sam_parts);
  }
};
Arena &arena() {
  static Arena *a = new Arena();
  return *a;
}

// ---------------------------------------------------------------- inputs ------
struct Input {
  const kslam_tail_params *p;
  const kslam_reads_view *reads;
  const kslam_overlap *ov;
  uint64_t n;
  uint32_t mid;  // paired: n_reads / 2
  int threads;
  uint32_t stages;
};

inline uint32_t read_len(const kslam_reads_view *r, uint32_t i) {
  return (uint32_t)(r->bases_off[i + 1] - r->bases_off[i]);
}

void check_input(const Input &in) {
  if (in.n >= (1ull << 30)) fail(KSLAM_ERR_UNSUPPORTED, "2^30 or more overlaps in one batch");
  if (in.reads->n_reads >= (1ull << 32)) fail(KSLAM_ERR_UNSUPPORTED, "2^32 or more reads in one batch");
  if (in.p->paired && (in.reads->n_reads < 2 || (in.reads->n_reads & 1)))
    fail(KSLAM_ERR_ARG, "paired data needs an even, non-zero number of reads ([R1 block | R2 block])");
  const uint64_t n_reads = in.reads->n_reads;
  std::atomic<int> bad(0);
  const size_t chunk = 1 << 16, n_tasks = (in.n + chunk - 1) / chunk;
  Pool::get().tasks(in.threads, n_tasks, [&](size_t t) {
    size_t lo = t * chunk, hi = std::min<size_t>(in.n, lo + chunk);
    for (size_t i = lo; i < hi; i++) {
      const kslam_overlap &a = in.ov[i];
      if (a.read >= n_reads) {
        bad = 1;
        return;
      }
      if (i + 1 < in.n) {
        const kslam_overlap &b = in.ov[i + 1];
        bool ok = a.read != b.read ? a.read < b.read
                                   : (a.entry != b.entry ? a.entry < b.entry : a.rel <= b.rel);
        if (!ok) {
          bad = 2;
          return;
        }
      }
    }
  });
  if (bad == 1) fail(KSLAM_ERR_ARG, "overlap refers to a read outside the batch");
  if (bad == 2)
    fail(KSLAM_ERR_ARG, "overlaps are not in alignToDatabase order (read, entry, rel)");
}

// ---------------------------------------------------------------- pairing -----
// What one pairing task produced, written straight into its regions of the arena.
struct PairOut {
  Rec *recs;          // region base (absolute: arena.paired.p + rec_base)
  size_t rec_base;    // offset of the region in arena.paired
  int32_t *inserts;   // same offsets in arena.inserts_by_task
  Group *groups;      // region in arena.groups_by_task
  size_t n_recs = 0, n_inserts = 0, n_groups = 0;
  uint64_t kept = 0;  // overlaps that passed the score threshold
  uint32_t max_entry = 0;
};

inline Rec single_rec(const kslam_overlap &o, uint32_t idx, bool is_r1) {
  return Rec{o.score, o.entry, o.ref_begin, o.ref_end, 0,
             is_r1 ? idx : KSLAM_NO_OVERLAP, is_r1 ? KSLAM_NO_OVERLAP : idx, 0};
}

// getPairsFromRead (src/PairedOverlap.h:132-242) as a streaming state: one
// candidate slot per (mate, strand); an overlap pairs with the latest overlap of
// the other mate on the opposite strand.  A run of k overlaps emits at most k
// single records and k pairs, which bounds a task's region at 2 x its overlaps.
struct RunState {
  uint32_t slot[2][2];
  bool used[2][2];
  void reset() {
    for (int m = 0; m < 2; m++)
      for (int s = 0; s < 2; s++) {
        slot[m][s] = KSLAM_NO_OVERLAP;
        used[m][s] = false;
      }
  }
};

struct Pairer {
  const Input &in;
  PairOut &out;
  RunState st;
  uint32_t cur_pid = 0, cur_entry = 0;
  bool open = false;
  size_t group_first = 0;

  Pairer(const Input &i, PairOut &o) : in(i), out(o) { st.reset(); }

  void emit(const Rec &r) {
    out.recs[out.n_recs++] = r;
    if (r.insert_size) out.inserts[out.n_inserts++] = (int32_t)r.insert_size;
    if (r.entry > out.max_entry) out.max_entry = r.entry;
  }
  void emit_single(uint32_t idx, bool is_r1) { emit(single_rec(in.ov[idx], idx, is_r1)); }
  // makePair, src/PairedOverlap.h:107-125
  void emit_both(uint32_t i1, uint32_t i2, bool r1_first) {
    const kslam_overlap &a = in.ov[i1], &b = in.ov[i2];
    uint32_t ins = r1_first ? (uint32_t)((int64_t)b.rel - a.rel + read_len(in.reads, b.read))
                            : (uint32_t)((int64_t)a.rel - b.rel + read_len(in.reads, a.read));
    emit(Rec{(uint16_t)(a.score + b.score), b.entry, std::min(a.ref_begin, b.ref_begin),
             std::max(a.ref_end, b.ref_end), ins, i1, i2, 0});
  }
  void close_run() {
    static const int order[4][2] = {{1, 0}, {1, 1}, {0, 0}, {0, 1}};
    for (auto &f : order)
      if (!st.used[f[0]][f[1]] && st.slot[f[0]][f[1]] != KSLAM_NO_OVERLAP)
        emit_single(st.slot[f[0]][f[1]], f[0] == 0);
    st.reset();
  }
  void close_group() {
    if (out.n_recs > group_first)
      out.groups[out.n_groups++] =
          Group{cur_pid, cur_pid + in.mid, out.rec_base + group_first, out.n_recs - group_first};
    group_first = out.n_recs;
  }
  void feed(uint32_t idx, uint32_t pid, int mate) {
    const kslam_overlap &o = in.ov[idx];
    if (open && (pid != cur_pid || o.entry != cur_entry)) {
      close_run();
      if (pid != cur_pid) close_group();
    }
    open = true;
    cur_pid = pid;
    cur_entry = o.entry;
    const int s = o.revcomp ? 1 : 0, m = mate;
    if (!st.used[m][s] && st.slot[m][s] != KSLAM_NO_OVERLAP) emit_single(st.slot[m][s], m == 0);
    st.slot[m][s] = idx;
    st.used[m][s] = false;
    uint32_t other = st.slot[1 - m][1 - s];
    if (other != KSLAM_NO_OVERLAP) {
      if (m == 0)
        emit_both(idx, other, false);
      else
        emit_both(other, idx, true);
      st.used[m][s] = true;
      st.used[1 - m][1 - s] = true;
    }
  }
  void finish() {
    if (open) {
      close_run();
      close_group();
    }
  }
};

size_t first_read_at_least(const kslam_overlap *ov, size_t lo, size_t hi, uint64_t read) {
  while (lo < hi) {
    size_t m = (lo + hi) / 2;
    if (ov[m].read < read)
      lo = m + 1;
    else
      hi = m;
  }
  return lo;
}

// The flat state between stages: dense read-pair groups whose `first` points
// into a sparse record array.
struct TailState {
  Rec *recs = nullptr;
  Group *groups = nullptr;
  size_t n_groups = 0;
  int32_t *inserts = nullptr;
  size_t n_inserts = 0;
  uint64_t n_recs = 0;  // records in use (sum of group counts)
  uint32_t max_entry = 0;
  size_t rec_extent = 0;  // highest record index + 1
};

void pair_stage(const Input &in, Arena &A, TailState &ts, uint64_t *kept) {
  const uint32_t thr = in.p->score_threshold;
  const size_t n_tasks = std::max<size_t>(1, std::min<size_t>((size_t)in.threads * 8, in.n / 2048 + 1));
  const uint64_t units = in.p->paired ? in.mid : in.reads->n_reads;  // group ids
  Rec *rec_arena = A.paired.ensure(2 * in.n + 1);
  int32_t *ins_arena = A.inserts_by_task.ensure(2 * in.n + 1);
  Group *grp_arena = A.groups_by_task.ensure(units + 1);
  std::vector<PairOut> outs(n_tasks);
  const size_t split = in.p->paired ? first_read_at_least(in.ov, 0, in.n, in.mid) : in.n;
  Pool::get().tasks(in.threads, n_tasks, [&](size_t t) {
    const uint64_t u0 = units * t / n_tasks, u1 = units * (t + 1) / n_tasks;
    PairOut o;  // a local: neighbouring elements of outs[] share cache lines and are bumped per record
    o.groups = grp_arena + u0;
    if (in.p->paired) {
      size_t i = first_read_at_least(in.ov, 0, split, u0), i1 = first_read_at_least(in.ov, 0, split, u1);
      size_t j = first_read_at_least(in.ov, split, in.n, in.mid + u0),
             j1 = first_read_at_least(in.ov, split, in.n, in.mid + u1);
      o.rec_base = 2 * (i + (j - split));
      o.recs = rec_arena + o.rec_base;
      o.inserts = ins_arena + o.rec_base;
      Pairer pr(in, o);
      while (i < i1 || j < j1) {
        bool take1;
        if (j >= j1)
          take1 = true;
        else if (i >= i1)
          take1 = false;
        else {
          const kslam_overlap &a = in.ov[i], &b = in.ov[j];
          uint32_t pa = a.read, pb = b.read - in.mid;
          take1 = pa != pb ? pa < pb : (a.entry != b.entry ? a.entry < b.entry : a.rel <= b.rel);
        }
        size_t idx = take1 ? i++ : j++;
        if (in.ov[idx].score < thr) continue;  // src/Overlap.h:329-341
        o.kept++;
        pr.feed((uint32_t)idx, take1 ? in.ov[idx].read : in.ov[idx].read - in.mid, take1 ? 0 : 1);
      }
      pr.finish();
    } else {
      // getPerReadOverlaps (src/Overlap.h:303-327) + dummy pairs (src/PairedOverlap.h:280-298)
      size_t i = first_read_at_least(in.ov, 0, in.n, u0), i1 = first_read_at_least(in.ov, 0, in.n, u1);
      o.rec_base = 2 * i;
      o.recs = rec_arena + o.rec_base;
      o.inserts = ins_arena + o.rec_base;
      size_t first = 0;
      uint32_t cur = 0;
      for (; i < i1; i++) {
        const kslam_overlap &a = in.ov[i];
        if (a.score < thr) continue;
        o.kept++;
        if (o.n_recs > first && a.read != cur) {
          o.groups[o.n_groups++] = Group{cur, 0, o.rec_base + first, o.n_recs - first};
          first = o.n_recs;
        }
        cur = a.read;
        o.recs[o.n_recs++] = single_rec(a, (uint32_t)i, true);
        if (a.entry > o.max_entry) o.max_entry = a.entry;
      }
      if (o.n_recs > first) o.groups[o.n_groups++] = Group{cur, 0, o.rec_base + first, o.n_recs - first};
    }
    outs[t] = o;
  });
  // dense group and insert-size lists (the records stay where the tasks wrote them)
  std::vector<size_t> g_at(n_tasks + 1, 0), i_at(n_tasks + 1, 0);
  ts.n_recs = 0;
  ts.max_entry = 0;
  ts.rec_extent = 0;
  for (size_t t = 0; t < n_tasks; t++) {
    g_at[t + 1] = g_at[t] + outs[t].n_groups;
    i_at[t + 1] = i_at[t] + outs[t].n_inserts;
    ts.n_recs += outs[t].n_recs;
    ts.max_entry = std::max(ts.max_entry, outs[t].max_entry);
    if (outs[t].n_recs) ts.rec_extent = std::max(ts.rec_extent, outs[t].rec_base + outs[t].n_recs);
    *kept += outs[t].kept;
  }
  ts.groups = A.groups.ensure(g_at.back() + 1);
  ts.inserts = A.inserts.ensure(i_at.back() + 1);
  Pool::get().tasks(in.threads, n_tasks, [&](size_t t) {
    if (outs[t].n_groups) memcpy(ts.groups + g_at[t], outs[t].groups, outs[t].n_groups * sizeof(Group));
    if (outs[t].n_inserts) memcpy(ts.inserts + i_at[t], outs[t].inserts, outs[t].n_inserts * sizeof(int32_t));
  });
  ts.recs = rec_arena;
  ts.n_groups = g_at.back();
  ts.n_inserts = i_at.back();
}

// ---------------------------------------------------------------- insert size -
// parallel LSD radix sort of int32 (4 x 8 bits, sign bit flipped)
void sort_i32(int threads, Arena &A, int32_t *v, size_t n) {
  if (n < (1 << 15) || threads == 1) {
    std::sort(v, v + n);
    return;
  }
  const int parts = std::min(threads, 32);
  int32_t *src = v, *dst = A.inserts_tmp.ensure(n);
  size_t *hist = A.radix_hist.ensure((size_t)parts * 256);
  for (int pass = 0; pass < 4; pass++) {
    const int shift = pass * 8;
    auto digit = [shift](int32_t x) { return (((uint32_t)x ^ 0x80000000u) >> shift) & 255u; };
    Pool::get().tasks(threads, parts, [&](size_t p) {
      size_t lo = n * p / parts, hi = n * (p + 1) / parts;
      size_t *h = &hist[p * 256];
      std::fill(h, h + 256, 0);
      for (size_t i = lo; i < hi; i++) h[digit(src[i])]++;
    });
    size_t run = 0;
    for (int d = 0; d < 256; d++)
      for (int p = 0; p < parts; p++) {
        size_t c = hist[(size_t)p * 256 + d];
        hist[(size_t)p * 256 + d] = run;
        run += c;
      }
    Pool::get().tasks(threads, parts, [&](size_t p) {
      size_t lo = n * p / parts, hi = n * (p + 1) / parts;
      size_t *h = &hist[p * 256];
      for (size_t i = lo; i < hi; i++) dst[h[digit(src[i])]++] = src[i];
    });
    std::swap(src, dst);
  }
  // 4 passes: the data is back in v
}

// getMaxAllowedInsertSize, src/PairedOverlap.h:314-360
uint32_t max_allowed_insert(int threads, Arena &A, int32_t *sz, size_t n) {
  if (!n) return UINT32_MAX;
  sort_i32(threads, A, sz, n);
  int32_t limit = 0;
  for (int i = 0; i < 99; i++) {
    if (sz[(size_t)floor(n * (i + 1) / 100.0)] - sz[(size_t)floor(n * (i) / 100.0)] > 1000) {
      limit = sz[(size_t)floor(n * (i) / 100)];
      break;
    }
  }
  int32_t lq = sz[(size_t)floor(n * 0.25)];
  int32_t uq = sz[(size_t)floor(n * 0.75)];
  int32_t lo = 0;
  int32_t hi = uq + 2 * (uq - lq);
  if (limit) hi = limit;
  if (hi == 0) hi = INT32_MAX;
  // the kept values lo <= v <= hi are one run of the sorted array
  size_t a = std::lower_bound(sz, sz + n, lo) - sz;
  size_t b = hi < lo ? a : (size_t)(std::upper_bound(sz, sz + n, hi) - sz);
  const size_t kept = b - a;
  // sum and sum of squares (the reference multiplies in int: wrap-around kept)
  const int parts = std::min<int>(threads, 64);
  std::vector<int64_t> s1(parts, 0), s2(parts, 0), mag(parts, 0);
  Pool::get().tasks(threads, parts, [&](size_t p) {
    size_t lo_i = a + kept * p / parts, hi_i = a + kept * (p + 1) / parts;
    int64_t x = 0, y = 0, m = 0;
    for (size_t i = lo_i; i < hi_i; i++) {
      int32_t v = sz[i];
      int32_t sq = (int32_t)((uint32_t)v * (uint32_t)v);
      x += v;
      y += sq;
      m += sq < 0 ? -(int64_t)sq : sq;
    }
    s1[p] = x;
    s2[p] = y;
    mag[p] = m;
  });
  int64_t t1 = 0, t2 = 0, tm = 0;
  for (int p = 0; p < parts; p++) {
    t1 += s1[p];
    t2 += s2[p];
    tm += mag[p];
  }
  double sum, sq;
  if (tm < (1ll << 53) && std::llabs(t1) < (1ll << 53)) {
    sum = (double)t1;  // every partial sum of the sequential accumulation is exact too
    sq = (double)t2;
  } else {
    sum = 0;
    sq = 0;
    for (size_t i = a; i < b; i++) {
      sum += sz[i];
      sq = sq + (int32_t)((uint32_t)sz[i] * (uint32_t)sz[i]);
    }
  }
  double mean = sum / kept;
  double sd = std::sqrt(sq / kept - mean * mean);
  double r = floor(mean + 6 * sd);
  return std::isnan(r) ? UINT_MAX : (uint32_t)r;
}

// ---------------------------------------------------------------- screens -----
inline bool by_insert(const Rec &a, const Rec &b) { return a.insert_size < b.insert_size; }
inline bool by_score_desc(const Rec &a, const Rec &b) { return a.combined_score > b.combined_score; }

// screenPairedAlignmentsByInsertSize(replace = true), src/PairedOverlap.h:396-436,
// on one read pair's records v[0..n); appends the split halves, returns the new n
// (at most 2n: the caller's region has that room).
size_t insert_screen(const kslam_overlap *ov, Rec *v, size_t n, uint32_t limit) {
  std::sort(v, v + n, by_insert);
  size_t cut = std::find_if(v, v + n, [&](const Rec &r) { return r.insert_size > limit; }) - v;
  size_t end = n;
  for (size_t i = cut; i < n; i++) {
    const kslam_overlap &o1 = ov[v[i].r1], &o2 = ov[v[i].r2];
    v[end++] = Rec{o1.score, v[i].entry, o1.ref_begin, o1.ref_end, 0, v[i].r1, KSLAM_NO_OVERLAP, 0};
    Rec &c = v[i];
    c.combined_score = o2.score;
    c.insert_size = 0;
    c.r1 = KSLAM_NO_OVERLAP;
    c.ref_start = o2.ref_begin;
    c.ref_end = o2.ref_end;
  }
  return end;
}

// screenPairedAlignmentsByScore on one read pair, src/PairedOverlap.h:366-380
Rec *score_screen(Rec *first, Rec *last, double fraction) {
  if (first == last) return last;
  std::sort(first, last, by_score_desc);
  unsigned top = first->combined_score;
  return std::find_if(first, last, [&](const Rec &r) { return r.combined_score < top * fraction; });
}

// ranges of groups with about equal record counts; rec_at[t] = records before range t
std::vector<size_t> group_ranges(const Group *groups, size_t n_groups, size_t n_tasks,
                                 std::vector<size_t> *rec_at = nullptr) {
  std::vector<size_t> cut(n_tasks + 1, n_groups);
  if (rec_at) rec_at->assign(n_tasks + 1, 0);
  uint64_t total = 0;
  for (size_t i = 0; i < n_groups; i++) total += groups[i].count + 1;
  uint64_t acc = 0, recs = 0;
  size_t t = 0;
  cut[0] = 0;
  for (size_t i = 0; i < n_groups; i++) {
    acc += groups[i].count + 1;
    recs += groups[i].count;
    while (t + 1 < n_tasks && acc >= total * (t + 1) / n_tasks) {
      cut[++t] = i + 1;
      if (rec_at) (*rec_at)[t] = recs;
    }
  }
  for (size_t k = t + 1; k <= n_tasks; k++) {
    cut[k] = n_groups;
    if (rec_at) (*rec_at)[k] = recs;
  }
  return cut;
}

size_t task_count(int threads, size_t items, size_t grain) {
  return std::max<size_t>(1, std::min<size_t>((size_t)threads * 8, items / grain + 1));
}

// both per-read-pair screens in one sweep; output into arena.screened, each task's
// region starting at 2 x (records before it)
void screen_stage(const Input &in, Arena &A, TailState &ts, bool do_insert, uint32_t limit, bool do_score) {
  if (!do_insert && !do_score) return;
  const size_t n_tasks = task_count(in.threads, ts.n_groups, 512);
  std::vector<size_t> rec_at;
  auto cut = group_ranges(ts.groups, ts.n_groups, n_tasks, &rec_at);
  Rec *out = A.screened.ensure(2 * ts.n_recs + 1);
  std::vector<size_t> used(n_tasks, 0);
  Pool::get().tasks(in.threads, n_tasks, [&](size_t t) {
    size_t at = 2 * rec_at[t];
    for (size_t g = cut[t]; g < cut[t + 1]; g++) {
      Group &gr = ts.groups[g];
      Rec *v = out + at;
      memcpy(v, ts.recs + gr.first, gr.count * sizeof(Rec));
      size_t n = gr.count;
      if (do_insert) n = insert_screen(in.ov, v, n, limit);
      if (do_score) n = score_screen(v, v + n, in.p->score_fraction) - v;
      gr.first = at;
      gr.count = n;
      at += n;
    }
    used[t] = at;
  });
  ts.recs = out;
  ts.n_recs = 0;
  ts.rec_extent = 0;
  for (size_t t = 0; t < n_tasks; t++) {
    ts.n_recs += used[t] - 2 * rec_at[t];
    if (used[t] > 2 * rec_at[t]) ts.rec_extent = std::max(ts.rec_extent, used[t]);
  }
}

// second score screen: shrinks groups in place
void rescreen_stage(const Input &in, TailState &ts) {
  const size_t n_tasks = task_count(in.threads, ts.n_groups, 512);
  auto cut = group_ranges(ts.groups, ts.n_groups, n_tasks);
  Pool::get().tasks(in.threads, n_tasks, [&](size_t t) {
    for (size_t g = cut[t]; g < cut[t + 1]; g++) {
      Rec *f = ts.recs + ts.groups[g].first;
      ts.groups[g].count = score_screen(f, f + ts.groups[g].count, in.p->score_fraction) - f;
    }
  });
}

// ---------------------------------------------------------------- pseudo-assembly
// pseudoAssembly, src/PairedOverlap.h:480-582.  The reference buckets pointers
// per entry in iteration order; here: a parallel stable counting sort of record
// numbers by entry, then entries are processed independently.
void chain_entry(Rec *recs, Span *v, size_t n) {
  std::sort(v, v + n, [](const Span &a, const Span &b) { return a.start < b.start; });
  size_t chain = 0;
  int reach = -1000000;
  uint32_t bases = 0;
  double per_base = 0;
  auto close = [&](size_t end) {
    long len = (long)(end - chain);
    if (len > 1) {
      double length = reach - v[chain].start;
      double coverage = bases / length;
      double avg = per_base / len;
      double score = coverage * avg * length;
      uint32_t s = score;
      for (size_t k = chain; k < end; k++) recs[v[k].rec].combined_score = s;
    }
  };
  for (size_t i = 0; i < n; i++) {
    const Rec &r = recs[v[i].rec];
    const int span = abs(r.ref_end - r.ref_start);
    if (v[i].start > reach - 20) {
      close(i);
      chain = i;
      reach = v[i].stop;
      per_base = r.combined_score * 1.0 / span;
      bases = span;
    } else {
      if (v[i].stop > reach) reach = v[i].stop;
      per_base += r.combined_score * 1.0 / span;
      bases += span;
    }
  }
  close(n);
}

void pseudo_stage(const Input &in, Arena &A, TailState &ts) {
  if (!ts.n_groups) return;
  if (ts.rec_extent >= (1ull << 32)) fail(KSLAM_ERR_UNSUPPORTED, "too many alignment pairs in one batch");
  const size_t n_entries = (size_t)ts.max_entry + 1;
  const size_t parts = std::max<size_t>(1, std::min<size_t>(std::min(in.threads, 32), ts.n_recs / 4096 + 1));
  auto part_cut = group_ranges(ts.groups, ts.n_groups, parts);
  uint32_t *hist = A.entry_hist.ensure(parts * n_entries);
  Pool::get().tasks(in.threads, parts, [&](size_t p) {
    uint32_t *h = hist + p * n_entries;
    std::fill(h, h + n_entries, 0u);
    for (size_t g = part_cut[p]; g < part_cut[p + 1]; g++) {
      const Rec *r = ts.recs + ts.groups[g].first;
      for (size_t k = 0; k < ts.groups[g].count; k++) h[r[k].entry]++;
    }
  });
  size_t *start = A.entry_start.ensure(n_entries + 1);
  size_t run = 0;
  for (size_t e = 0; e < n_entries; e++) {
    start[e] = run;
    for (size_t p = 0; p < parts; p++) {
      uint32_t c = hist[p * n_entries + e];
      hist[p * n_entries + e] = (uint32_t)(run - start[e]);
      run += c;
    }
  }
  start[n_entries] = run;
  const size_t n = run;
  Span *spans = A.spans.ensure(n + 1);
  Pool::get().tasks(in.threads, parts, [&](size_t p) {
    uint32_t *h = hist + p * n_entries;
    for (size_t g = part_cut[p]; g < part_cut[p + 1]; g++) {
      const size_t first = ts.groups[g].first;
      for (size_t k = 0; k < ts.groups[g].count; k++) {
        const Rec &r = ts.recs[first + k];
        spans[start[r.entry] + h[r.entry]++] = Span{r.ref_start, r.ref_end, (uint32_t)(first + k)};
      }
    }
  });
  // entries in chunks of roughly equal record counts
  const size_t n_tasks = task_count(in.threads, n, 1024);
  std::vector<size_t> cut(n_tasks + 1, n_entries);
  cut[0] = 0;
  {
    size_t t = 0;
    for (size_t e = 0; e < n_entries && t + 1 < n_tasks; e++)
      while (t + 1 < n_tasks && start[e + 1] >= n * (t + 1) / n_tasks) cut[++t] = e + 1;
    for (size_t k = t + 1; k <= n_tasks; k++) cut[k] = n_entries;
  }
  Pool::get().tasks(in.threads, n_tasks, [&](size_t t) {
    for (size_t e = cut[t]; e < cut[t + 1]; e++)
      if (start[e + 1] > start[e]) chain_entry(ts.recs, spans + start[e], start[e + 1] - start[e]);
  });
}

// ---------------------------------------------------------------- the flow -----
// src/SLAM.h:102-128
void run_tail(const Input &in, Arena &A, TailState &ts, kslam_tail_stats &st) {
  check_input(in);
  double t0 = now_ms();
  pair_stage(in, A, ts, &st.n_overlaps_screened);
  st.n_overlaps_in = in.n;
  st.n_paired_initial = ts.n_recs;
  double t1 = now_ms();
  st.ms_pairing = t1 - t0;
  const bool do_insert = in.p->paired && (in.stages & KSLAM_TAIL_INSERT_SCREEN);
  uint32_t limit = UINT32_MAX;
  if (do_insert) {
    st.n_insert_sizes = ts.n_inserts;
    limit = max_allowed_insert(in.threads, A, ts.inserts, ts.n_inserts);
    st.max_insert_size = limit;
  }
  double t2 = now_ms();
  st.ms_insert = t2 - t1;
  screen_stage(in, A, ts, do_insert, limit, (in.stages & KSLAM_TAIL_SCORE_SCREEN) != 0);
  double t3 = now_ms();
  st.ms_screens = t3 - t2;
  if (in.p->pseudo_assembly && (in.stages & KSLAM_TAIL_PSEUDO_ASM)) {
    pseudo_stage(in, A, ts);
    rescreen_stage(in, ts);
  }
  st.ms_pseudo = now_ms() - t3;
  st.n_read_pairs = ts.n_groups;
  uint64_t total = 0;
  for (size_t g = 0; g < ts.n_groups; g++) total += ts.groups[g].count;
  st.n_paired_final = total;
  st.threads = in.threads;
}


// ---------------------------------------------------------------- SAM ---------
struct LogTables {  // src/SAM.h:33-48
  double match[100], mismatch[100];
  LogTables() {
    match[0] = std::log10(1.0 - std::pow(10.0, 1.0 / -10.0));
    mismatch[0] = 1 / -10.0;
    for (int i = 1; i < 100; i++) {
      match[i] = std::log10(1.0 - std::pow(10.0, i / -10.0));
      mismatch[i] = i / -10.0;
    }
  }
};
const LogTables &tables() {
  static const LogTables t;
  return t;
}

struct SamInput {
  const kslam_tail_params *p;
  const kslam_reads_view *reads;
  const kslam_index_view *index;
  const kslam_overlap *ov;
  uint64_t n_ov;
  const uint32_t *pool;
  uint64_t n_pool;
  // per-row NM / log-probability / MD computed on the GPU (kslam_row_details, include/kslam.h); when
  // present the writer never reads the entry bases
  const kslam_row_detail *det = nullptr;
  const char *md_pool = nullptr;
  uint64_t n_md = 0;
  bool groups_sorted = false;  // KSLAM_TAIL_GROUPS_SORTED: writeSAMOutputPairs' per-pair sort already done
};

struct Row {  // SAMEntry, src/SAM.h:238-277, text fields as slices of the task's scratch
  bool mapped = false;
  uint32_t rname_entry = 0;
  uint32_t pos = 0, pnext = 0, nm = 0, xo = 0;
  int32_t tlen = 0;
  uint16_t as = 0, xs = 0, flag = 0;
  uint8_t mapq = 0;
  double prob = 0;
  size_t cigar_at = 0, cigar_len = 0, md_at = 0, md_len = 0;
  bool cigar_star = true;
};

struct ComplementLut {  // src/sequenceTools.h:77-97: A<->T, C<->G, everything else unchanged
  char t[256];
  ComplementLut() {
    for (int i = 0; i < 256; i++) t[i] = (char)i;
    t['A'] = 'T';
    t['T'] = 'A';
    t['C'] = 'G';
    t['G'] = 'C';
  }
};
const ComplementLut &complement_lut() {
  static const ComplementLut l;
  return l;
}

// MD text under construction: the reference collects components and merges them
// afterwards (src/SAM.h:204-235: adjacent counts add up, "0" between a deletion
// and a mismatch); here the same rules are applied while streaming.
struct MdWriter {
  Text &md;
  uint64_t pending = 0;
  bool have_pending = false, after_del = false;
  explicit MdWriter(Text &t) : md(t) {}
  void matches(uint32_t run) {
    if (run) {
      pending += run;
      have_pending = true;
    }
  }
  void flush() {
    if (have_pending) {
      md.num(pending);
      pending = 0;
      have_pending = false;
      after_del = false;
    }
  }
  void mismatch(char ref_base) {
    flush();
    if (after_del) {
      md.put('0');
      after_del = false;
    }
    md.put(ref_base);
  }
  void deletion(const char *ref, uint32_t len) {
    flush();
    md.put('^');
    md.put(ref, len);
    after_del = true;
  }
};

#if defined(__SSE2__)
// 16 query bases for columns i .. i + 15 of an M operation.  RC: the 16 read bytes ENDING at
// base_at - i, reversed, with A<->T and C<->G swapped (upper case only, everything else unchanged:
// the table of complement_lut()).
template <bool RC>
inline __m128i query16(const char *base_at, uint32_t i) {
  if (!RC) return _mm_loadu_si128(reinterpret_cast<const __m128i *>(base_at + i));
  __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(base_at - (ptrdiff_t)i - 15));
  // reverse the 16 bytes with SSE2 only: reverse the 16-bit units, then swap the bytes inside them
  v = _mm_shuffle_epi32(v, _MM_SHUFFLE(0, 1, 2, 3));
  v = _mm_shufflelo_epi16(v, _MM_SHUFFLE(2, 3, 0, 1));
  v = _mm_shufflehi_epi16(v, _MM_SHUFFLE(2, 3, 0, 1));
  v = _mm_or_si128(_mm_slli_epi16(v, 8), _mm_srli_epi16(v, 8));
  // complement: x ^ ('A' ^ 'T') where x is A or T, x ^ ('C' ^ 'G') where x is C or G
  const __m128i at = _mm_or_si128(_mm_cmpeq_epi8(v, _mm_set1_epi8('A')), _mm_cmpeq_epi8(v, _mm_set1_epi8('T')));
  const __m128i cg = _mm_or_si128(_mm_cmpeq_epi8(v, _mm_set1_epi8('C')), _mm_cmpeq_epi8(v, _mm_set1_epi8('G')));
  const __m128i flip = _mm_or_si128(_mm_and_si128(at, _mm_set1_epi8('A' ^ 'T')), _mm_and_si128(cg, _mm_set1_epi8('C' ^ 'G')));
  return _mm_xor_si128(v, flip);
}
#endif

// One M operation of `len` columns.  RC: the query is the reverse complement of
// the read, walked backwards through bases/qual.  PROB: also accumulate the log
// probability (a serial chain of double additions in column order -- its
// rounding is part of the result, so it cannot be reassociated).
template <bool RC, bool PROB>
inline void match_columns(const char *ref, const char *base_at, const char *qual_at, uint32_t len,
                          MdWriter &w, uint32_t &nm_io, double &logp_io) {
  const char *lut = complement_lut().t;
  const LogTables &tb = tables();
  // accumulators in locals: through the references they would live in memory and
  // every column would pay a store-to-load round trip on the addition chain
  double logp = logp_io;
  uint32_t nm = nm_io, run = 0;
  uint32_t i = 0;
#if defined(__SSE2__)
  if (!PROB) {
    // without the probability chain a column is a byte compare: 16 at a time, and only the
    // mismatching columns (a handful per read) are visited one by one
    for (; i + 16 <= len; i += 16) {
      const __m128i r = _mm_loadu_si128(reinterpret_cast<const __m128i *>(ref + i));
      uint32_t miss = 0xFFFFu ^ (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(r, query16<RC>(base_at, i)));
      uint32_t from = 0;
      while (miss) {
        const uint32_t k = (uint32_t)__builtin_ctz(miss);
        run += k - from;
        nm++;
        w.matches(run);
        w.mismatch(ref[i + k]);
        run = 0;
        from = k + 1;
        miss &= miss - 1;
      }
      run += 16 - from;
    }
    base_at += RC ? -(ptrdiff_t)i : (ptrdiff_t)i;
    qual_at += RC ? -(ptrdiff_t)i : (ptrdiff_t)i;
  }
#endif
  for (; i < len; i++) {
    const char qc = RC ? lut[(unsigned char)*base_at] : *base_at;
    int q = 0;
    if (PROB) {
      q = (unsigned char)*qual_at - 33;
      if (__builtin_expect((unsigned)q >= 100u, 0))
        fail(KSLAM_ERR_ARG, "quality character outside phred+33 0..99");
    }
    if (__builtin_expect(ref[i] == qc, 1)) {
      run++;
      if (PROB) logp += tb.match[q];
    } else {
      nm++;
      w.matches(run);
      w.mismatch(ref[i]);
      if (PROB) logp += tb.mismatch[q];
      run = 0;
    }
    base_at += RC ? -1 : 1;
    qual_at += RC ? -1 : 1;
  }
  w.matches(run);
  nm_io = nm;
  logp_io = logp;
}

// getCigarAndMD, src/SAM.h:101-237, streamed.  want_prob = false skips the
// probability (the caller knows it cancels out of the mapping quality); it
// returns false, having done nothing, when 10^logp could underflow to 0 -- the
// one way the probability's value would still matter.
bool cigar_and_md(const SamInput &in, const kslam_overlap &o, Text &scratch, Row &r, bool want_prob) {
  r.cigar_at = scratch.n;
  r.cigar_len = 0;
  r.md_at = scratch.n;
  r.md_len = 0;
  r.nm = 0;
  r.prob = 1.0;  // pow(10, 0)
  if (!in.pool || o.cigar_len == 0) return true;
  if (o.cigar_off + o.cigar_len > in.n_pool) fail(KSLAM_ERR_ARG, "cigar slice outside the cigar pool");
  const uint64_t rb = in.reads->bases_off[o.read], L = in.reads->bases_off[o.read + 1] - rb;
  const uint64_t qb = in.reads->quality_off[o.read];
  if (in.reads->quality_off[o.read + 1] - qb != L)
    fail(KSLAM_ERR_ARG, "quality string length differs from the read length");
  if (in.det) {
    // The walk over CIGAR + read + quality + entry bases was done on the GPU: NM, the log-probability
    // (same tables, same order of additions) and the MD text come with the row; only the CIGAR text is
    // formatted here, from the ops alone.
    const kslam_row_detail &d = in.det[&o - in.ov];
    if (d.flags & 2u) fail(KSLAM_ERR_ARG, "cigar runs past the end of the read or the entry");
    if (d.md_off + d.md_len > in.n_md) fail(KSLAM_ERR_ARG, "MD slice outside the MD pool");
    Text &cg = scratch;
    if (o.query_begin > 0) {
      cg.num((uint64_t)o.query_begin);
      cg.put('S');
    }
    for (uint32_t k = 0; k < o.cigar_len; k++) {
      const uint32_t c = in.pool[o.cigar_off + k], len = c >> 4, op = c & 15;
      cg.num(len);
      if (op < 3) cg.put("MID"[op]);
    }
    const int64_t tail = (int64_t)L - o.query_end - 1;
    if (tail > 0) {
      cg.num((uint64_t)tail);
      cg.put('S');
    }
    r.cigar_len = cg.n - r.cigar_at;
    r.md_at = cg.n;
    cg.put(in.md_pool + d.md_off, d.md_len);
    r.md_len = d.md_len;
    r.nm = d.nm;
    // 10^logp only where its value matters: when it is summed with other rows' (want_prob), or when
    // it could underflow to 0 (then prob / prob is 0 / 0 in the reference, not 1)
    if (want_prob || d.logp <= -300.0) {
      if (d.flags & 1u) fail(KSLAM_ERR_ARG, "quality character outside phred+33 0..99");
      r.prob = std::pow(10, d.logp);
    }
    return true;
  }
  const char *bases = in.reads->bases + rb, *qual = in.reads->quality + qb;
  const char *ref = in.index->bases + in.index->bases_off[o.entry];
  const int64_t ref_len = (int64_t)(in.index->bases_off[o.entry + 1] - in.index->bases_off[o.entry]);
  const bool rc = o.revcomp != 0;
  const size_t mark = scratch.n;
  Text &cg = scratch;
  int64_t rp = o.ref_begin, qp = 0;
  if (o.query_begin > 0) {
    cg.num((uint64_t)o.query_begin);
    cg.put('S');
    qp += o.query_begin;
  }
  static thread_local Text md;
  md.n = 0;
  MdWriter w(md);
  double logp = 0;
  uint32_t nm = 0, m_columns = 0, m_mismatches = 0;
  for (uint32_t k = 0; k < o.cigar_len; k++) {
    const uint32_t c = in.pool[o.cigar_off + k], len = c >> 4, op = c & 15;
    cg.num(len);
    if (op == 0) {
      cg.put('M');
      if (rp < 0 || qp < 0 || rp + len > ref_len || qp + len > (int64_t)L)
        fail(KSLAM_ERR_ARG, "cigar runs past the end of the read or the entry");
      const int64_t at = rc ? (int64_t)L - 1 - qp : qp;
      const uint32_t before = nm;
      if (rc) {
        if (want_prob)
          match_columns<true, true>(ref + rp, bases + at, qual + at, len, w, nm, logp);
        else
          match_columns<true, false>(ref + rp, bases + at, qual + at, len, w, nm, logp);
      } else {
        if (want_prob)
          match_columns<false, true>(ref + rp, bases + at, qual + at, len, w, nm, logp);
        else
          match_columns<false, false>(ref + rp, bases + at, qual + at, len, w, nm, logp);
      }
      m_columns += len;
      m_mismatches += nm - before;
      rp += len;
      qp += len;
    } else if (op == 1) {
      cg.put('I');
      nm += len;
      qp += len;
    } else if (op == 2) {
      cg.put('D');
      if (rp < 0 || rp + len > ref_len) fail(KSLAM_ERR_ARG, "cigar runs past the end of the entry");
      w.deletion(ref + rp, len);
      rp += len;
      nm += len;
    }
  }
  w.flush();
  if (!want_prob) {
    // |log10 P| <= 0.687 per matching column (phred 0/1, src/SAM.h:33-40) + 9.9 per
    // mismatch (phred 99, src/SAM.h:41-48); 10^x is non-zero down to x = -323.3
    if (0.687 * (m_columns - m_mismatches) + 9.9 * m_mismatches >= 320.0) {
      scratch.n = mark;
      return false;
    }
  }
  const int64_t tail = (int64_t)L - o.query_end - 1;
  if (tail > 0) {
    cg.num((uint64_t)tail);
    cg.put('S');
  }
  r.cigar_len = cg.n - r.cigar_at;
  r.md_at = cg.n;
  cg.put(md.p, md.n);
  r.md_len = md.n;
  r.nm = nm;
  if (want_prob) r.prob = std::pow(10, logp);
  return true;
}

// GenbankEntry::getGene, src/GenbankTools.h:170-185
int64_t best_gene(const kslam_index_view *ix, uint32_t e, int32_t start, int32_t stop) {
  if (!ix->n_genes) return -1;
  int64_t best = -1;
  int32_t widest = 0;
  for (uint64_t g = ix->gene_first[e]; g < ix->gene_first[e + 1]; g++) {
    int32_t shared = std::min<int>(stop, ix->gene_stop[g]) - std::max<int>(start, ix->gene_start[g]);
    if (shared > widest) {
      best = (int64_t)g;
      widest = shared;
    }
  }
  return best;
}

// ceil(-10 log10(t)) stored into a uint8_t, src/SAM.h:502-506.  When no row of
// this mate has a probability the reference divides 0 by 0; converting that NaN
// to an integer is undefined in C++ and gives 0 in the low byte with x86-64 gcc
// (cvttsd2si -> 0x80000000): that observable value is kept.
inline uint8_t mapq_of(double prob, double sum) {
  double t = 1.0 - prob / sum;
  if (t <= 0.00001) t = 0.00001;
  double q = ceil(-10.0 * std::log10(t));
  if (std::isnan(q)) return 0;
  return (uint8_t)q;
}

void put_col(Text &out, const char *t, const uint64_t *off, uint64_t i) {
  out.put(t + off[i], off[i + 1] - off[i]);
}

// Appends to a Text whose capacity for the whole line was reserved up front: no capacity check
// per field, literals with compile-time lengths, two digits per division.
struct LineWriter {
  char *w;
  explicit LineWriter(char *at) : w(at) {}
  // Short fields (read names, locus tags, CIGAR and MD strings of a few characters) are the rule: up to 16 bytes go as
  // two overlapping 8-byte moves instead of a call into memcpy, which was 5 % of the stage.  The reserve of 384 bytes
  // per line covers the 8 bytes a short copy may write past its field; reading 8 bytes of a shorter field is avoided.
  void bytes(const char *s, size_t k) {
    if (k >= 8 && k <= 16) {
      uint64_t a, b;
      memcpy(&a, s, 8);
      memcpy(&b, s + k - 8, 8);
      memcpy(w, &a, 8);
      memcpy(w + k - 8, &b, 8);
    } else if (k < 8) {
      for (size_t i = 0; i < k; i++) w[i] = s[i];
    } else {
      memcpy(w, s, k);
    }
    w += k;
  }
  void col(const char *t, const uint64_t *off, uint64_t i) { bytes(t + off[i], off[i + 1] - off[i]); }
  void ch(char c) { *w++ = c; }
  template <size_t N>
  void lit(const char (&s)[N]) {
    memcpy(w, s, N - 1);
    w += N - 1;
  }
  void num(uint64_t v) {
    static const char pairs[201] =
        "00010203040506070809101112131415161718192021222324252627282930313233343536373839"
        "40414243444546474849505152535455565758596061626364656667686970717273747576777879"
        "8081828384858687888990919293949596979899";
    if (v < 100) {                       // flags, mapping qualities, NM, X0: one or two digits
      if (v >= 10) {
        memcpy(w, pairs + 2 * v, 2);
        w += 2;
      } else {
        *w++ = (char)('0' + v);
      }
      return;
    }
    if (v < 10000) {                     // scores, template lengths, short positions: three or four digits
      const uint64_t q = v / 100, r = v - q * 100;
      if (q >= 10) {
        memcpy(w, pairs + 2 * q, 2);
        w += 2;
      } else {
        *w++ = (char)('0' + q);
      }
      memcpy(w, pairs + 2 * r, 2);
      w += 2;
      return;
    }
    char t[24];
    int k = 24;
    while (v >= 100) {
      const uint64_t q = v / 100, r = v - q * 100;
      k -= 2;
      memcpy(t + k, pairs + 2 * r, 2);
      v = q;
    }
    if (v >= 10) {
      k -= 2;
      memcpy(t + k, pairs + 2 * v, 2);
    } else {
      t[--k] = (char)('0' + v);
    }
    const int n = 24 - k;                // 3..20 digits: two overlapping 8-byte moves up to 16
    if (n <= 8) {
      for (int i = 0; i < n; i++) w[i] = t[k + i];
    } else {
      memcpy(w, t + k, (size_t)n);
    }
    w += n;
  }
  void snum(int64_t v) {
    if (v < 0) {
      ch('-');
      num((uint64_t)(-v));
    } else
      num((uint64_t)v);
  }
};

// SAMEntry::getEntry, src/SAM.h:278-305
void put_line(const SamInput &in, Text &out, const Text &scratch, const Row &r, uint32_t qname_read,
              int64_t gene, uint32_t xt, bool paired) {
  const kslam_index_view *ix = in.index;
  const kslam_reads_view *rd = in.reads;
  // everything variable in the line + room for the fixed text and 12 numbers of <= 20 digits
  size_t bound = (rd->ids_off[qname_read + 1] - rd->ids_off[qname_read]) +
                 (ix->locus_tag_off[r.rname_entry + 1] - ix->locus_tag_off[r.rname_entry]) + r.cigar_len + r.md_len + 384;
  if (gene >= 0)
    bound += (ix->gene_name_off[gene + 1] - ix->gene_name_off[gene]) + (ix->protein_id_off[gene + 1] - ix->protein_id_off[gene]) +
             (ix->product_off[gene + 1] - ix->product_off[gene]);
  LineWriter o(out.need(bound));
  o.col(rd->ids, rd->ids_off, qname_read);
  o.ch('\t');
  o.num(r.flag);
  o.ch('\t');
  o.col(ix->locus_tag, ix->locus_tag_off, r.rname_entry);
  o.ch('\t');
  o.num(r.pos);
  o.ch('\t');
  o.num(r.mapq);
  o.ch('\t');
  if (!in.p->report_cigar || r.cigar_star)
    o.ch('*');
  else
    o.bytes(scratch.p + r.cigar_at, r.cigar_len);
  o.ch('\t');
  o.ch(paired ? '=' : '*');  // single end prints only the R1 row, whose rnext is "*" (src/SAM.h:416-420)
  o.ch('\t');
  o.num(r.pnext);
  o.ch('\t');
  o.snum(r.tlen);
  o.lit("\t*\t*");
  if (r.mapped) {
    if (in.p->report_cigar) {
      o.lit("\tMD:Z:");
      o.bytes(scratch.p + r.md_at, r.md_len);
    }
    o.lit("\tAS:i:");
    o.num(r.as);
    o.lit("\tXS:i:");
    o.num(r.xs);
    o.lit("\tNM:i:");
    o.num(r.nm);
    o.lit("\tX0:i:");
    o.num(r.xo);
    if (xt != 0) {
      o.lit("\tXT:i:");
      o.num(xt);
    }
    if (gene >= 0) {
      if (ix->gene_name_off[gene + 1] > ix->gene_name_off[gene]) {
        o.lit("\tXG:Z:");
        o.col(ix->gene_name, ix->gene_name_off, gene);
      }
      if (ix->protein_id_off[gene + 1] > ix->protein_id_off[gene]) {
        o.lit("\tXP:Z:");
        o.col(ix->protein_id, ix->protein_id_off, gene);
      }
      if (ix->product_off[gene + 1] > ix->product_off[gene]) {
        o.lit("\tXR:Z:\"");
        o.col(ix->product, ix->product_off, gene);
        o.ch('"');
      }
    }
  }
  o.ch('\n');
  out.n = (size_t)(o.w - out.p);
}

// writeSAMOutputPairs (src/SAM.h:443-512) with getSAMFromPair (src/SAM.h:352-433)
void write_group(const SamInput &in, const Group &g, Rec *recs, Text &out, Text &scratch,
                 std::vector<Row> &rows, std::vector<int64_t> &genes) {
  if (!g.count) return;
  const bool paired = in.p->paired != 0;
  if (!in.groups_sorted) std::sort(recs, recs + g.count, by_score_desc);
  scratch.n = 0;
  rows.clear();
  genes.clear();
  uint32_t hits1 = 0, hits2 = 0;
  size_t n_rows = 0;
  // A mate with a single aligned row among those reported has mapping quality
  // ceil(-10 log10(1 - p/p)): its probability p cancels (as long as p > 0), so the
  // serial log-probability sum is not needed for it.
  const size_t n_use = std::min<size_t>(g.count, std::max<uint32_t>(in.p->num_sam_alignments, 1));
  uint32_t use1 = 0, use2 = 0;
  for (size_t k = 0; k < n_use; k++) {
    use1 += recs[k].r1 != KSLAM_NO_OVERLAP;
    use2 += recs[k].r2 != KSLAM_NO_OVERLAP;
  }
  for (size_t k = 0; k < g.count; k++) {
    const Rec &p = recs[k];
    const bool has1 = p.r1 != KSLAM_NO_OVERLAP, has2 = p.r2 != KSLAM_NO_OVERLAP;
    if ((has1 && p.r1 >= in.n_ov) || (has2 && p.r2 >= in.n_ov) || p.entry >= in.index->n_entries)
      fail(KSLAM_ERR_ARG, "alignment pair refers outside the overlap array or the index");
    if (has1) hits1++;
    if (has2) hits2++;
    Row a, b;
    uint16_t fa = 0x40, fb = 0x80;
    if (!paired) fa = fb = 0;
    bool a_next_unmapped = false;
    if (paired) {
      fa |= 0x1;
      fb |= 0x1;
    }
    bool conventional = true;
    const kslam_overlap *o1 = has1 ? &in.ov[p.r1] : nullptr, *o2 = has2 ? &in.ov[p.r2] : nullptr;
    if (o1 && o1->entry >= in.index->n_entries) fail(KSLAM_ERR_ARG, "overlap entry outside the index");
    if (o2 && o2->entry >= in.index->n_entries) fail(KSLAM_ERR_ARG, "overlap entry outside the index");
    if (has1 && has2) {
      fa |= 0x2;
      fb |= 0x2;
      conventional = o1->ref_begin < o2->ref_begin;
      if (o1->revcomp) {
        fa |= 0x10;
        fb |= 0x20;
      }
      if (o2->revcomp) {
        fb |= 0x10;
        fa |= 0x20;
      }
    } else if (has1) {
      a_next_unmapped = true;
      fb |= 0x4;
      if (o1->revcomp) fa |= 0x10;
    } else if (has2) {
      fb |= 0x8;
      fa |= 0x4;
      if (o2->revcomp) fb |= 0x10;
    }
    if (has1) {  // SAMEntry::init, src/SAM.h:339-351
      if (use1 > 1 || !cigar_and_md(in, *o1, scratch, a, false)) cigar_and_md(in, *o1, scratch, a, true);
      a.cigar_star = false;
      a.mapped = true;
      a.rname_entry = o1->entry;
      a.pos = (uint32_t)(o1->ref_begin + 1);
      a.as = o1->score;
    }
    if (has2) {
      if (use2 > 1 || !cigar_and_md(in, *o2, scratch, b, false)) cigar_and_md(in, *o2, scratch, b, true);
      b.cigar_star = false;
      b.mapped = true;
      b.rname_entry = o2->entry;
      b.pos = (uint32_t)(o2->ref_begin + 1);
      b.as = o2->score;
    }
    a.pnext = b.pos;
    b.pnext = a.pos;
    if (!has1) {
      a.rname_entry = b.rname_entry;
      a.pos = b.pos;
      b.pnext = b.pos;
      a.pnext = b.pos;
    }
    if (!has2) {
      b.rname_entry = a.rname_entry;
      b.pos = a.pos;
      a.pnext = a.pos;
      b.pnext = a.pos;
    }
    if (!paired) {
      a.pnext = 0;
      a_next_unmapped = false;
    }
    if (a_next_unmapped) fa |= 0x8;
    int32_t tlen = p.ref_end - p.ref_start + 1;
    if (!(has1 || has2)) tlen = 0;
    if (!conventional) tlen *= -1;
    a.tlen = tlen;
    b.tlen = tlen * -1;
    a.xs = b.xs = (uint16_t)p.combined_score;
    a.flag = fa | 0x100;  // secondary until chosen as primary
    b.flag = fb | 0x100;
    rows.push_back(a);
    rows.push_back(b);
    genes.push_back(best_gene(in.index, p.entry, p.ref_start, p.ref_end));
    n_rows++;
    if (n_rows >= in.p->num_sam_alignments) break;
  }
  double sum1 = 0, sum2 = 0;
  for (size_t k = 0; k < n_rows; k++) {
    sum1 += rows[2 * k].prob;
    sum2 += rows[2 * k + 1].prob;
  }
  rows[0].flag &= ~0x100;
  rows[1].flag &= ~0x100;
  for (size_t k = 0; k < n_rows; k++) {
    Row &a = rows[2 * k], &b = rows[2 * k + 1];
    a.xo = hits1;
    b.xo = hits2;
    a.mapq = mapq_of(a.prob, sum1);
    b.mapq = mapq_of(b.prob, sum2);
    const uint32_t xt = in.index->taxonomy_id[recs[k].entry];
    put_line(in, out, scratch, a, g.r1_read, genes[k], xt, paired);
    if (paired) put_line(in, out, scratch, b, g.r2_read, genes[k], xt, paired);
    if (in.p->sam_xa) break;
  }
}

// The genome window of an alignment is a random access into a multi-GB index:
// three or four cache lines that nothing has touched recently.  Formatting a row
// takes about as long as one DRAM round trip, so the windows of the read pairs a
// few steps ahead are requested while the current one is written (and, one level
// earlier, the overlap records that say where those windows are).
inline void prefetch_overlaps(const SamInput &in, const Group &g, const Rec *recs) {
  const size_t n = std::min<size_t>(g.count, std::max<uint32_t>(in.p->num_sam_alignments, 1));
  for (size_t k = 0; k < n; k++) {
    if (recs[k].r1 != KSLAM_NO_OVERLAP && recs[k].r1 < in.n_ov) __builtin_prefetch(&in.ov[recs[k].r1]);
    if (recs[k].r2 != KSLAM_NO_OVERLAP && recs[k].r2 < in.n_ov) __builtin_prefetch(&in.ov[recs[k].r2]);
  }
}
inline void prefetch_windows(const SamInput &in, const Group &g, const Rec *recs) {
  if (!in.pool) return;
  if (in.det) {   // no entry window to fetch: the row's detail record, its MD bytes and its CIGAR ops
    const size_t n = std::min<size_t>(g.count, std::max<uint32_t>(in.p->num_sam_alignments, 1));
    for (size_t k = 0; k < n; k++)
      for (uint32_t idx : {recs[k].r1, recs[k].r2}) {
        if (idx == KSLAM_NO_OVERLAP || idx >= in.n_ov) continue;
        __builtin_prefetch(&in.det[idx]);
        __builtin_prefetch(in.pool + in.ov[idx].cigar_off);
      }
    return;
  }
  const size_t n = std::min<size_t>(g.count, std::max<uint32_t>(in.p->num_sam_alignments, 1));
  for (size_t k = 0; k < n; k++)
    for (uint32_t idx : {recs[k].r1, recs[k].r2}) {
      if (idx == KSLAM_NO_OVERLAP || idx >= in.n_ov) continue;
      const kslam_overlap &o = in.ov[idx];
      if (o.entry >= in.index->n_entries || o.ref_begin < 0 || o.ref_end < o.ref_begin) continue;
      const char *w = in.index->bases + in.index->bases_off[o.entry] + o.ref_begin;
      const int64_t span = std::min<int64_t>((int64_t)o.ref_end - o.ref_begin + 1, 1024);
      for (int64_t at = 0; at < span + 63; at += 64) __builtin_prefetch(w + at);
      __builtin_prefetch(in.pool + o.cigar_off);
    }
}

void check_sam_views(const SamInput &in) {
  if (!in.index || !in.index->bases_off || !in.index->locus_tag_off || !in.index->taxonomy_id)
    fail(KSLAM_ERR_ARG, "index view is incomplete");
  if (!in.reads->ids_off || !in.reads->quality_off) fail(KSLAM_ERR_ARG, "reads view needs ids and quality");
  if (in.index->n_genes && (!in.index->gene_first || !in.index->gene_start || !in.index->gene_stop ||
                            !in.index->gene_name_off || !in.index->protein_id_off || !in.index->product_off))
    fail(KSLAM_ERR_ARG, "index view has n_genes > 0 but no gene columns");
}

}  // namespace

// The background SAM writer (include/kslam_tail.h: kslam_sam_writer_*): one thread that write()s whole batches of
// chunks in order while the next batch is being formatted.  The formatter's per-task buffers are handed over, not
// copied; written sets come back through `spare` and are reused (fresh memory for 400 MB of text per batch costs more
// in page faults than the formatting).
struct kslam_sam_writer {
  int fd = -1;
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::vector<Text>> queue;
  std::vector<std::vector<Text>> spare;
  bool stop = false, busy = false;
  std::atomic<int> error{0};           // errno of the first failed write
  uint64_t bytes = 0;
  double write_s = 0;
  double starved_s = 0;                // with an empty queue, after the first block (KSLAM_DEBUG prints it at close)
  uint64_t blocks = 0;
  // Tried and removed (round 4): copying a batch into a shared MAPPING of the file's next region with four threads, to get
  // around the inode lock that serialises write() into one file (11 GB/s here: 37-41 ms per 404 MB batch).  On the bench
  // boxes' file system the write faults cost far more than the lock: 150-200 ms per batch against 45 (13 against 45 M reads/s).
  void run() {
    name_thread("kslam-writer");
    for (;;) {
      std::vector<Text> set;
      {
        std::unique_lock<std::mutex> lk(m);
        const double tw = now_ms();
        const bool was_empty = queue.empty();
        cv.wait(lk, [&] { return stop || !queue.empty(); });
        if (was_empty && blocks && !queue.empty()) starved_s += (now_ms() - tw) * 1e-3;
        if (queue.empty()) return;
        set = std::move(queue.front());
        queue.pop_front();
        busy = true;
      }
      const double t0 = now_ms();
      uint64_t done = 0;
      for (Text &t : set) {
        const char *p = t.p;
        size_t left = t.n;
        while (left && !error) {
          const ssize_t w = ::write(fd, p, std::min<size_t>(left, (size_t)1 << 30));
          if (w < 0) {
            if (errno == EINTR) continue;
            error = errno ? errno : EIO;
            break;
          }
          p += w;
          left -= (size_t)w;
          done += (uint64_t)w;
        }
        t.n = 0;
      }
      bool borrowed = false;
      for (Text &t : set)
        if (t.ext_release) {
          t.drop();            // back to its owner as soon as it is written
          borrowed = true;
        }
      {
        std::lock_guard<std::mutex> lk(m);
        bytes += done;
        blocks++;
        write_s += (now_ms() - t0) * 1e-3;
        busy = false;
        if (!borrowed && spare.size() < 3) spare.push_back(std::move(set));
      }
      cv.notify_all();
    }
  }
};

namespace {

// where the SAM text goes: one malloc'ed buffer, or a writer called chunk by chunk in order
struct SamSink {
  char **text = nullptr;
  uint64_t *text_len = nullptr;
  kslam_write_fn write = nullptr;
  void *user = nullptr;
};

void sam_stage(const SamInput &in, Arena &A, int threads, const Group *groups, size_t n_groups, Rec *recs,
               const SamSink &sink, uint64_t *bytes) {
  check_sam_views(in);
  const size_t n_tasks = task_count(threads, n_groups, 256);
  auto cut = group_ranges(groups, n_groups, n_tasks);
  if (A.sam_parts.size() < n_tasks) A.sam_parts.resize(n_tasks);
  std::vector<Text> &parts = A.sam_parts;
  const uint64_t n_reads = in.reads->n_reads;
  Pool::get().tasks(threads, n_tasks, [&](size_t t) {
    static thread_local Text scratch;
    static thread_local std::vector<Row> rows;
    static thread_local std::vector<int64_t> genes;
    Text out(std::move(parts[t]));  // a local: the write cursor must not share a cache line with other tasks
    out.n = 0;
    // one allocation per task in the common case: lines x (fixed fields + id + cigar/MD/tags)
    size_t lines = 0;
    for (size_t g = cut[t]; g < cut[t + 1]; g++)
      lines += std::min<size_t>(groups[g].count, std::max<uint32_t>(in.p->num_sam_alignments, 1));
    if (cut[t + 1] > cut[t] && groups[cut[t]].r1_read < n_reads) {
      const Group &g0 = groups[cut[t]];
      size_t id_len = in.reads->ids_off[g0.r1_read + 1] - in.reads->ids_off[g0.r1_read];
      out.reserve(lines * (in.p->paired ? 2 : 1) * (id_len + (in.p->report_cigar ? 220 : 140)) + 4096);
    }
    const size_t near = 6, far = 14;  // read pairs ahead: genome windows / overlap records
    for (size_t g = cut[t]; g < cut[t + 1]; g++) {
      if (groups[g].r1_read >= n_reads || groups[g].r2_read >= n_reads)
        fail(KSLAM_ERR_ARG, "read pair refers to a read outside the batch");
      if (g + far < cut[t + 1]) prefetch_overlaps(in, groups[g + far], recs + groups[g + far].first);
      if (g + near < cut[t + 1]) prefetch_windows(in, groups[g + near], recs + groups[g + near].first);
      write_group(in, groups[g], recs + groups[g].first, out, scratch, rows, genes);
    }
    parts[t] = std::move(out);
  });
  std::vector<size_t> at(n_tasks + 1, 0);
  for (size_t t = 0; t < n_tasks; t++) at[t + 1] = at[t] + parts[t].n;
  *bytes = at.back();
  if (sink.write == &kslam_write_queued && sink.user) {
    // the background writer: this batch's buffers join its queue (at most two batches wait there: a writer that falls
    // behind holds the formatter back instead of piling up text), and a written set takes their place in the arena
    kslam_sam_writer *w = static_cast<kslam_sam_writer *>(sink.user);
    std::vector<Text> next;
    {
      std::unique_lock<std::mutex> lk(w->m);
      w->cv.wait(lk, [&] { return w->error || w->queue.size() < 2; });
      if (w->error) fail(KSLAM_ERR_ARG, std::string("writing the SAM text failed: ") + strerror(w->error));
      if (!w->spare.empty()) {
        next = std::move(w->spare.back());
        w->spare.pop_back();
      }
      std::vector<Text> mine(std::make_move_iterator(parts.begin()), std::make_move_iterator(parts.begin() + n_tasks));
      w->queue.push_back(std::move(mine));
    }
    w->cv.notify_all();
    for (size_t t = 0; t < n_tasks && t < next.size(); t++) parts[t] = std::move(next[t]);
    return;
  }
  if (sink.write) {
    for (size_t t = 0; t < n_tasks; t++)
      if (parts[t].n && sink.write(sink.user, parts[t].p, parts[t].n) != 0)
        fail(KSLAM_ERR_ARG, "the SAM writer callback reported a failure");
    return;
  }
  char *buf = (char *)malloc(at.back() + 1);
  if (!buf) fail(KSLAM_ERR_OOM, "out of host memory for the SAM text");
  Pool::get().tasks(threads, n_tasks, [&](size_t t) {
    if (parts[t].n) memcpy(buf + at[t], parts[t].p, parts[t].n);
  });
  buf[at.back()] = 0;
  *sink.text = buf;
  *sink.text_len = at.back();
}

Input make_input(const kslam_tail_params *p, const kslam_reads_view *reads, const kslam_overlap *ov,
                 uint64_t n) {
  if (!p || !reads || (!ov && n)) fail(KSLAM_ERR_ARG, "null argument");
  if (!reads->bases_off) fail(KSLAM_ERR_ARG, "reads view needs bases_off");
  Input in{p, reads, ov, n, (uint32_t)(reads->n_reads / 2), thread_count(p),
           p->stages ? p->stages : KSLAM_TAIL_ALL};
  return in;
}

void tail_to_sam(const kslam_tail_params *params, const kslam_reads_view *reads,
                 const kslam_index_view *index, const kslam_overlap *overlaps, uint64_t n_overlaps,
                 const uint32_t *cigar_pool, uint64_t n_cigar, const SamSink &sink,
                 kslam_tail_stats *stats, const kslam_row_detail *det = nullptr, const char *md_pool = nullptr,
                 uint64_t n_md = 0) {
  Input in = make_input(params, reads, overlaps, n_overlaps);
  Arena &A = arena();
  std::lock_guard<std::mutex> one(A.call);
  TailState ts;
  kslam_tail_stats st;
  memset(&st, 0, sizeof st);
  run_tail(in, A, ts, st);
  SamInput si{params, reads, index, overlaps, n_overlaps, cigar_pool, n_cigar};
  if (det && cigar_pool) {
    if (!md_pool && n_md) fail(KSLAM_ERR_ARG, "null MD pool");
    si.det = det;
    si.md_pool = md_pool;
    si.n_md = n_md;
  }
  double t0 = now_ms();
  sam_stage(si, A, in.threads, ts.groups, ts.n_groups, ts.recs, sink, &st.sam_bytes);
  st.ms_sam = now_ms() - t0;
  if (stats) *stats = st;
}

}  // namespace

extern "C" {

const char *kslam_tail_last_error(void) { return g_err.c_str(); }

kslam_status kslam_tail_pairs(const kslam_tail_params *params, const kslam_reads_view *reads,
                              const kslam_overlap *overlaps, uint64_t n_overlaps,
                              kslam_read_pair **read_pairs, uint64_t *n_read_pairs,
                              kslam_paired_overlap **pairs, uint64_t *n_pairs,
                              kslam_tail_stats *stats) {
  return guarded([&] {
    if (!read_pairs || !n_read_pairs || !pairs || !n_pairs) fail(KSLAM_ERR_ARG, "null output argument");
    Input in = make_input(params, reads, overlaps, n_overlaps);
    Arena &A = arena();
    std::lock_guard<std::mutex> one(A.call);
    TailState ts;
    kslam_tail_stats st;
    memset(&st, 0, sizeof st);
    run_tail(in, A, ts, st);
    // dense copies for the caller
    Group *g_out = (Group *)malloc(sizeof(Group) * (ts.n_groups + 1));
    Rec *r_out = (Rec *)malloc(sizeof(Rec) * (st.n_paired_final + 1));
    if (!g_out || !r_out) {
      free(g_out);
      free(r_out);
      fail(KSLAM_ERR_OOM, "out of host memory");
    }
    uint64_t at = 0;
    for (size_t g = 0; g < ts.n_groups; g++) {
      g_out[g] = ts.groups[g];
      g_out[g].first = at;
      at += ts.groups[g].count;
    }
    const size_t chunk = 4096, n_tasks = (ts.n_groups + chunk - 1) / chunk;
    Pool::get().tasks(in.threads, n_tasks, [&](size_t t) {
      for (size_t g = t * chunk; g < std::min(ts.n_groups, (t + 1) * chunk); g++)
        memcpy(r_out + g_out[g].first, ts.recs + ts.groups[g].first, ts.groups[g].count * sizeof(Rec));
    });
    *read_pairs = g_out;
    *pairs = r_out;
    *n_read_pairs = ts.n_groups;
    *n_pairs = st.n_paired_final;
    if (stats) *stats = st;
  });
}

kslam_status kslam_sam_records(const kslam_tail_params *params, const kslam_reads_view *reads,
                               const kslam_index_view *index, const kslam_overlap *overlaps,
                               uint64_t n_overlaps, const uint32_t *cigar_pool, uint64_t n_cigar,
                               const kslam_read_pair *read_pairs, uint64_t n_read_pairs,
                               kslam_paired_overlap *pairs, uint64_t n_pairs, char **text,
                               uint64_t *text_len, kslam_tail_stats *stats) {
  return guarded([&] {
    if (!text || !text_len || (!read_pairs && n_read_pairs) || (!pairs && n_pairs))
      fail(KSLAM_ERR_ARG, "null argument");
    Input in = make_input(params, reads, overlaps, n_overlaps);
    for (uint64_t g = 0; g < n_read_pairs; g++)
      if (read_pairs[g].first + read_pairs[g].count > n_pairs)
        fail(KSLAM_ERR_ARG, "read pair slice outside the pairs array");
    Arena &A = arena();
    std::lock_guard<std::mutex> one(A.call);
    SamInput si{params, reads, index, overlaps, n_overlaps, cigar_pool, n_cigar};
    SamSink sink;
    sink.text = text;
    sink.text_len = text_len;
    uint64_t bytes = 0;
    double t0 = now_ms();
    sam_stage(si, A, in.threads, read_pairs, n_read_pairs, pairs, sink, &bytes);
    if (stats) {
      stats->ms_sam = now_ms() - t0;
      stats->sam_bytes = bytes;
    }
  });
}

kslam_status kslam_tail_sam(const kslam_tail_params *params, const kslam_reads_view *reads,
                            const kslam_index_view *index, const kslam_overlap *overlaps,
                            uint64_t n_overlaps, const uint32_t *cigar_pool, uint64_t n_cigar,
                            char **text, uint64_t *text_len, kslam_tail_stats *stats) {
  return guarded([&] {
    if (!text || !text_len) fail(KSLAM_ERR_ARG, "null output argument");
    SamSink sink;
    sink.text = text;
    sink.text_len = text_len;
    tail_to_sam(params, reads, index, overlaps, n_overlaps, cigar_pool, n_cigar, sink, stats);
  });
}

kslam_status kslam_tail_sam_write(const kslam_tail_params *params, const kslam_reads_view *reads,
                                  const kslam_index_view *index, const kslam_overlap *overlaps,
                                  uint64_t n_overlaps, const uint32_t *cigar_pool, uint64_t n_cigar,
                                  kslam_write_fn write, void *user, kslam_tail_stats *stats) {
  return guarded([&] {
    if (!write) fail(KSLAM_ERR_ARG, "null writer");
    SamSink sink;
    sink.write = write;
    sink.user = user;
    tail_to_sam(params, reads, index, overlaps, n_overlaps, cigar_pool, n_cigar, sink, stats);
  });
}

kslam_status kslam_tail_sam_rows(const kslam_tail_params *params, const kslam_reads_view *reads,
                                 const kslam_index_view *index, const kslam_overlap *overlaps,
                                 uint64_t n_overlaps, const uint32_t *cigar_pool, uint64_t n_cigar,
                                 const kslam_row_detail *details, const char *md_pool, uint64_t n_md,
                                 char **text, uint64_t *text_len, kslam_tail_stats *stats) {
  return guarded([&] {
    if (!text || !text_len) fail(KSLAM_ERR_ARG, "null output argument");
    SamSink sink;
    sink.text = text;
    sink.text_len = text_len;
    tail_to_sam(params, reads, index, overlaps, n_overlaps, cigar_pool, n_cigar, sink, stats, details, md_pool, n_md);
  });
}

kslam_status kslam_tail_sam_write_rows(const kslam_tail_params *params, const kslam_reads_view *reads,
                                       const kslam_index_view *index, const kslam_overlap *overlaps,
                                       uint64_t n_overlaps, const uint32_t *cigar_pool, uint64_t n_cigar,
                                       const kslam_row_detail *details, const char *md_pool, uint64_t n_md,
                                       kslam_write_fn write, void *user, kslam_tail_stats *stats) {
  return guarded([&] {
    if (!write) fail(KSLAM_ERR_ARG, "null writer");
    SamSink sink;
    sink.write = write;
    sink.user = user;
    tail_to_sam(params, reads, index, overlaps, n_overlaps, cigar_pool, n_cigar, sink, stats, details, md_pool, n_md);
  });
}

kslam_status kslam_tail_finish_prepare(const kslam_tail_params *params, const kslam_reads_view *reads,
                                       const kslam_overlap *overlaps, uint64_t n_overlaps,
                                       kslam_read_pair *read_pairs, uint64_t n_read_pairs,
                                       kslam_paired_overlap *pairs, uint64_t n_pairs, int sort_groups,
                                       kslam_tail_stats *stats) {
  return guarded([&] {
    if ((!read_pairs && n_read_pairs) || (!pairs && n_pairs)) fail(KSLAM_ERR_ARG, "null argument");
    Input in = make_input(params, reads, overlaps, n_overlaps);
    for (uint64_t g = 0; g < n_read_pairs; g++)
      if (read_pairs[g].first + read_pairs[g].count > n_pairs) fail(KSLAM_ERR_ARG, "read pair slice outside the pairs array");
    Arena &A = arena();
    std::lock_guard<std::mutex> one(A.call);
    kslam_tail_stats st;
    memset(&st, 0, sizeof st);
    st.n_overlaps_in = n_overlaps;
    TailState ts;
    ts.recs = pairs;
    ts.groups = read_pairs;
    ts.n_groups = n_read_pairs;
    ts.n_recs = n_pairs;
    ts.rec_extent = n_pairs;
    const double t0 = now_ms();
    if (in.p->pseudo_assembly && (in.stages & KSLAM_TAIL_PSEUDO_ASM)) {
      uint32_t mx = 0;
      for (uint64_t k = 0; k < n_pairs; k++) mx = std::max(mx, pairs[k].entry);
      ts.max_entry = mx;
      pseudo_stage(in, A, ts);
      rescreen_stage(in, ts);
    }
    st.ms_pseudo = now_ms() - t0;
    if (sort_groups) {   // writeSAMOutputPairs' first statement, src/SAM.h:446-450
      const size_t n_tasks = task_count(in.threads, ts.n_groups, 512);
      auto cut = group_ranges(ts.groups, ts.n_groups, n_tasks);
      Pool::get().tasks(in.threads, n_tasks, [&](size_t t) {
        for (size_t g = cut[t]; g < cut[t + 1]; g++)
          if (ts.groups[g].count > 1)
            std::sort(ts.recs + ts.groups[g].first, ts.recs + ts.groups[g].first + ts.groups[g].count, by_score_desc);
      });
    }
    st.n_read_pairs = ts.n_groups;
    uint64_t total = 0;
    for (size_t g = 0; g < ts.n_groups; g++) total += ts.groups[g].count;
    st.n_paired_final = total;
    st.threads = in.threads;
    if (stats) *stats = st;
  });
}

kslam_status kslam_tail_finish_write_rows(const kslam_tail_params *params, const kslam_reads_view *reads,
                                          const kslam_index_view *index, const kslam_overlap *overlaps,
                                          uint64_t n_overlaps, const uint32_t *cigar_pool, uint64_t n_cigar,
                                          const kslam_row_detail *details, const char *md_pool, uint64_t n_md,
                                          kslam_read_pair *read_pairs, uint64_t n_read_pairs,
                                          kslam_paired_overlap *pairs, uint64_t n_pairs, kslam_write_fn write,
                                          void *user, kslam_tail_stats *stats) {
  return guarded([&] {
    if (!write) fail(KSLAM_ERR_ARG, "null writer");
    if ((!read_pairs && n_read_pairs) || (!pairs && n_pairs)) fail(KSLAM_ERR_ARG, "null argument");
    Input in = make_input(params, reads, overlaps, n_overlaps);
    for (uint64_t g = 0; g < n_read_pairs; g++)
      if (read_pairs[g].first + read_pairs[g].count > n_pairs) fail(KSLAM_ERR_ARG, "read pair slice outside the pairs array");
    Arena &A = arena();
    std::lock_guard<std::mutex> one(A.call);
    kslam_tail_stats st;
    memset(&st, 0, sizeof st);
    st.n_overlaps_in = n_overlaps;
    TailState ts;
    ts.recs = pairs;
    ts.groups = read_pairs;
    ts.n_groups = n_read_pairs;
    ts.n_recs = n_pairs;
    ts.rec_extent = n_pairs;
    double t0 = now_ms();
    if (in.p->pseudo_assembly && (in.stages & KSLAM_TAIL_PSEUDO_ASM)) {
      uint32_t mx = 0;
      for (uint64_t k = 0; k < n_pairs; k++) mx = std::max(mx, pairs[k].entry);
      ts.max_entry = mx;
      pseudo_stage(in, A, ts);
      rescreen_stage(in, ts);
    }
    st.ms_pseudo = now_ms() - t0;
    st.n_read_pairs = ts.n_groups;
    uint64_t total = 0;
    for (size_t g = 0; g < ts.n_groups; g++) total += ts.groups[g].count;
    st.n_paired_final = total;
    st.threads = in.threads;
    SamInput si{params, reads, index, overlaps, n_overlaps, cigar_pool, n_cigar};
    si.groups_sorted = (params->stages & KSLAM_TAIL_GROUPS_SORTED) != 0;
    if (details && cigar_pool) {
      if (!md_pool && n_md) fail(KSLAM_ERR_ARG, "null MD pool");
      si.det = details;
      si.md_pool = md_pool;
      si.n_md = n_md;
    }
    SamSink sink;
    sink.write = write;
    sink.user = user;
    double t1 = now_ms();
    sam_stage(si, A, in.threads, ts.groups, ts.n_groups, ts.recs, sink, &st.sam_bytes);
    st.ms_sam = now_ms() - t1;
    if (stats) *stats = st;
  });
}

// a kslam_write_fn that streams to a file descriptor: `user` points to the int
int kslam_write_fd(void *user, const char *data, uint64_t len) {
  if (!user) return 1;
  const int fd = *static_cast<const int *>(user);
  while (len) {
    const ssize_t w = ::write(fd, data, (size_t)std::min<uint64_t>(len, 1ull << 30));
    if (w < 0) {
      if (errno == EINTR) continue;
      return 1;
    }
    data += w;
    len -= (uint64_t)w;
  }
  return 0;
}

kslam_status kslam_sam_writer_open(int fd, kslam_sam_writer **out) {
  return guarded([&] {
    if (!out || fd < 0) fail(KSLAM_ERR_ARG, "bad argument");
    kslam_sam_writer *w = new kslam_sam_writer();
    w->fd = fd;
    w->th = std::thread([w] { w->run(); });
    *out = w;
  });
}

// the writer as a plain kslam_write_fn (a caller's own text, e.g. the header): copied into a queued buffer
int kslam_write_queued(void *user, const char *data, uint64_t len) {
  kslam_sam_writer *w = static_cast<kslam_sam_writer *>(user);
  if (!w) return 1;
  std::vector<Text> one(1);
  try {
    one[0].put(data, (size_t)len);
  } catch (...) {
    return 1;
  }
  {
    std::unique_lock<std::mutex> lk(w->m);
    w->cv.wait(lk, [&] { return w->error || w->queue.size() < 2; });
    if (w->error) return 1;
    w->queue.push_back(std::move(one));
  }
  w->cv.notify_all();
  return 0;
}

// a block another stage produced goes into the queue as it is (no copy); release(user, data) is called on the writer thread
// once it has been written (or at close, if a write failed before its turn), and before returning when the call fails
kslam_status kslam_sam_writer_enqueue(kslam_sam_writer *w, char *data, uint64_t len, void (*release)(void *user, void *data),
                                      void *user) {
  return guarded([&] {
    // the block is CONSUMED whatever happens: from here on `t` hands it back through release() on every path out,
    // the argument failures included (the caller has given it away and must not release it a second time)
    Text t;
    if (data && release) {
      t.p = data;
      t.n = t.cap = (size_t)len;
      t.ext_release = release;
      t.ext_user = user;
    }
    if (!w || (!data && len) || !release) fail(KSLAM_ERR_ARG, "null argument");
    std::vector<Text> one;
    one.push_back(std::move(t));
    {
      std::unique_lock<std::mutex> lk(w->m);
      static const size_t room = [] { const char *e = getenv("KSLAM_WRITER_QUEUE"); return (size_t)(e ? std::max(1, std::min(16, atoi(e))) : 2); }();
      w->cv.wait(lk, [&] { return w->error || w->queue.size() < room; });
      if (w->error) fail(KSLAM_ERR_ARG, std::string("writing the SAM text failed: ") + strerror(w->error));   // `one` releases the block
      w->queue.push_back(std::move(one));
    }
    w->cv.notify_all();
  });
}

kslam_status kslam_sam_writer_close(kslam_sam_writer *w, uint64_t *bytes_written, double *seconds_writing) {
  if (!w) return KSLAM_ERR_ARG;
  {
    std::lock_guard<std::mutex> lk(w->m);
    w->stop = true;
  }
  w->cv.notify_all();
  if (w->th.joinable()) w->th.join();      // drains the queue first
  const int err = w->error;
  if (getenv("KSLAM_DEBUG"))
    fprintf(stderr, "[kslam] SAM writer: %llu blocks, %.1f MB, %.1f ms in write(), %.1f ms with an empty queue after its first block\n",
            (unsigned long long)w->blocks, w->bytes / 1e6, w->write_s * 1e3, w->starved_s * 1e3);
  if (bytes_written) *bytes_written = w->bytes;
  if (seconds_writing) *seconds_writing = w->write_s;
  delete w;
  if (err) {
    g_err = std::string("writing the SAM text failed: ") + strerror(err);
    return KSLAM_ERR_ARG;
  }
  return KSLAM_OK;
}

void kslam_tail_release_buffers(void) {
  Arena &A = arena();
  std::lock_guard<std::mutex> one(A.call);
  A.release();
}

kslam_status kslam_sam_header(const kslam_index_view *index, const char *command_line, char **text,
                              uint64_t *text_len) {
  return guarded([&] {
    if (!index || !text || !text_len || !index->bases_off || !index->locus_tag_off || !index->taxonomy_id)
      fail(KSLAM_ERR_ARG, "null argument");
    Text h;
    h.lit("@HD\tVN:1.0\tSO:unsorted\n");
    for (uint64_t e = 0; e < index->n_entries; e++) {
      h.lit("@SQ\tSN:");
      put_col(h, index->locus_tag, index->locus_tag_off, e);
      h.lit("\tLN:");
      h.num(index->bases_off[e + 1] - index->bases_off[e]);
      if (index->taxonomy_id[e]) {
        h.lit("\tSP:");
        h.num(index->taxonomy_id[e]);
      }
      h.put('\n');
    }
    h.lit("@PG\tID:SLAM\tPN:SLAM\tVN:1.0\tCL:\"");
    h.lit(command_line ? command_line : "");
    h.lit("\"\n");
    char *buf = (char *)malloc(h.n + 1);
    if (!buf) fail(KSLAM_ERR_OOM, "out of host memory");
    memcpy(buf, h.p, h.n);
    buf[h.n] = 0;
    *text = buf;
    *text_len = h.n;
  });
}
}
