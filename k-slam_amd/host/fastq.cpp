// fastq.cpp -- FASTQ ingest behind include/kslam_fastq.h.
//
// The reference reads a FASTQ stream with a getline loop that builds one object
// of three std::strings per read (src/FASTQsequence.h:129-165).  Here the bytes
// of the file are indexed in parallel -- count line terminators per chunk, prefix
// sum, then every chunk knows the global number of each line it closes, and line
// i belongs to record i / 4 as field i % 4 -- and the three fields are copied
// into column arrays (bases / quality / identifiers + offsets), which is what the
// device upload (kslam_load_reads) and the host tail (kslam_reads_view) consume.
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <vector>

#include "../../include/kslam_fastq.h"
#include "workers.hpp"

namespace {
using namespace kslam_host;

// A line terminator starts at p: "\r" (alone or followed by "\n"), or "\n" not
// preceded by "\r" (src/sequenceTools.h:57-64).
inline bool is_event(const char *t, uint64_t p) {
  return t[p] == '\r' || (t[p] == '\n' && !(p > 0 && t[p - 1] == '\r'));
}
inline uint64_t line_after(const char *t, uint64_t len, uint64_t p) {
  return (t[p] == '\r' && p + 1 < len && t[p + 1] == '\n') ? p + 2 : p + 1;
}

struct Field {
  uint64_t start = 0;
  uint32_t len = 0;
};

// Uninitialised storage kept between calls (same reasoning as the column block cache below).
template <typename T>
struct Buf {
  T *p = nullptr;
  size_t cap = 0;
  T *ensure(size_t n) {
    if (n > cap) {
      size_t want = std::max(n, cap + cap / 2);
      T *q = (T *)realloc((void *)p, want * sizeof(T));
      if (!q) fail(KSLAM_ERR_OOM, "out of host memory for the FASTQ line index");
      p = q;
      cap = want;
    }
    return p;
  }
};

struct IndexArena {  // one per stream slot (R1 / R2)
  Buf<uint64_t> events;
  Buf<Field> id, bases, qual;
};
std::mutex g_parse_call;  // one parse at a time: each spreads over all workers anyway
IndexArena g_arena[2];

// where the three kept fields of every record of one stream are
struct StreamIndex {
  const char *text = nullptr;
  uint64_t n = 0;         // records taken
  uint64_t consumed = 0;  // where the reference's stream would stand afterwards
  Field *id = nullptr, *bases = nullptr, *qual = nullptr;
};

// FASTQSequence::FASTQSequence, src/FASTQsequence.h:61-71: drop the first
// character, cut at the first space, then at the first '/'
inline Field identifier_of(const char *t, uint64_t start, uint64_t len) {
  Field f;
  if (len <= 1) return f;
  const char *h = t + start;
  uint64_t end = len;
  if (const void *sp = memchr(h, ' ', len)) {
    uint64_t space = (const char *)sp - h;
    // substr(1, space - 1); a space at index 0 gives substr(1, 0) = ""
    end = space == 0 ? 1 : space;
  }
  f.start = start + 1;
  uint64_t n = end - 1;
  if (const void *sl = memchr(h + 1, '/', n)) n = (const char *)sl - (h + 1);
  if (n > 0xFFFFFFFFull) fail(KSLAM_ERR_UNSUPPORTED, "FASTQ header line longer than 4 GiB");
  f.len = (uint32_t)n;
  return f;
}

void index_stream(const char *text, uint64_t len, uint64_t max_reads, bool at_eof, int threads,
                  IndexArena &A, StreamIndex &ix) {
  ix.text = text;
  if (len && !text) fail(KSLAM_ERR_ARG, "null text");
  // without the rest of the stream a trailing "\r" may or may not be half of "\r\n"
  uint64_t scan_len = len;
  if (!at_eof && len && text[len - 1] == '\r') scan_len = len - 1;
  const uint64_t chunk = 1 << 20;
  const size_t n_chunks = (size_t)((scan_len + chunk - 1) / chunk);
  std::vector<uint64_t> count(n_chunks + 1, 0), last_event(n_chunks, UINT64_MAX);
  // one scan of the text: each chunk lists where its line terminators start, in its own slice of
  // a shared array (one slot per 8 bytes of text; a chunk with denser terminators than that --
  // not FASTQ -- spills to a vector of its own)
  const uint64_t slots = chunk / 8;
  uint64_t *ev_arena = A.events.ensure(n_chunks * slots + 1);
  std::vector<std::vector<uint64_t>> spill(n_chunks);
  Pool::get().tasks(threads, n_chunks, [&](size_t c) {
    const uint64_t lo = c * chunk, hi = std::min(scan_len, lo + chunk);
    uint64_t *ev = ev_arena + c * slots, k = 0;
    std::vector<uint64_t> &extra = spill[c];
    auto push = [&](uint64_t p) {
      if (k < slots)
        ev[k] = p;
      else
        extra.push_back(p);
      k++;
    };
    uint64_t p = lo;
    const uint64_t ones = 0x0101010101010101ull, high = 0x8080808080808080ull;
    while (p < hi) {
      // skip 8 bytes at a time while none of them is LF (0x0A) or CR (0x0D)
      while (p + 8 <= hi) {
        uint64_t w;
        memcpy(&w, text + p, 8);
        const uint64_t a = w ^ (ones * 0x0A), b = w ^ (ones * 0x0D);
        if ((((a - ones) & ~a) | ((b - ones) & ~b)) & high) break;
        p += 8;
      }
      const uint64_t stop = std::min(hi, p + 8);
      for (; p < stop; p++)
        if ((text[p] == '\n' || text[p] == '\r') && is_event(text, p)) push(p);
    }
    count[c + 1] = k;
    last_event[c] = !k ? UINT64_MAX : (k <= slots ? ev[k - 1] : extra.back());
  });
  for (size_t c = 0; c < n_chunks; c++) count[c + 1] += count[c];
  const uint64_t terminated = count[n_chunks];
  // start of the line each chunk's first event closes
  std::vector<uint64_t> carry(n_chunks + 1, 0);
  for (size_t c = 0; c < n_chunks; c++)
    carry[c + 1] = last_event[c] == UINT64_MAX ? carry[c] : line_after(text, len, last_event[c]);
  const uint64_t rest_start = carry[n_chunks];  // text after the last terminator
  // at the true end of the stream the reference reads the unterminated rest (if any) and then
  // one more, empty, line before the stream fails (src/sequenceTools.h:65-68)
  const uint64_t rest_lines = at_eof ? (rest_start < len ? 2 : 1) : 0;
  uint64_t n = (terminated + rest_lines) / 4;
  if (max_reads && n > max_reads) n = max_reads;
  ix.n = n;
  ix.id = A.id.ensure(n + 1);
  ix.bases = A.bases.ensure(n + 1);
  ix.qual = A.qual.ensure(n + 1);
  std::vector<uint64_t> after_quality(1, 0);  // next-line start after the last record's 4th line
  const uint64_t want_lines = 4 * n;
  auto store = [&](uint64_t line, uint64_t start, uint64_t end, uint64_t next) {
    const uint64_t rec = line / 4, flen = end - start;
    if (flen > 0xFFFFFFFFull) fail(KSLAM_ERR_UNSUPPORTED, "FASTQ line longer than 4 GiB");
    switch (line % 4) {
      case 0: ix.id[rec] = identifier_of(text, start, flen); break;
      case 1: ix.bases[rec] = Field{start, (uint32_t)flen}; break;
      case 3:
        ix.qual[rec] = Field{start, (uint32_t)flen};
        if (rec + 1 == n) after_quality[0] = next;
        break;
      default: break;
    }
  };
  Pool::get().tasks(threads, n_chunks, [&](size_t c) {
    uint64_t line = count[c];
    uint64_t start = carry[c];
    const uint64_t k = count[c + 1] - count[c];
    const uint64_t *ev = ev_arena + c * slots;
    for (uint64_t j = 0; j < k && line < want_lines; j++) {
      const uint64_t p = j < slots ? ev[j] : spill[c][j - slots];
      const uint64_t next = line_after(text, len, p);
      store(line, start, p, next);
      start = next;
      line++;
    }
  });
  // the lines past the last terminator (only ever part of the final record)
  for (uint64_t line = terminated; line < want_lines; line++) {
    if (line == terminated && rest_start < len)
      store(line, rest_start, len, len);
    else
      store(line, len, len, len);  // the empty line read at end of stream
  }
  // short of max_reads at the true end of the stream, the reference's loop has read on to the end
  if (at_eof && (!max_reads || n < max_reads))
    ix.consumed = len;
  else
    ix.consumed = n ? after_quality[0] : 0;
}

// Column arrays are recycled: a batch is a few hundred MB, and faulting that much fresh
// memory in for every batch costs more than parsing it.  kslam_reads_free parks the
// blocks here (a handful at most); the next parse takes the ones that are big enough.
struct BlockCache {
  struct Block {
    void *p;
    size_t cap;
    bool in_use;
    void (*release)(void *, size_t);   // nullptr: free()
  };
  static void drop(const Block &b) {
    if (b.release) b.release(b.p, b.cap); else free(b.p);
  }
  std::mutex m;
  std::vector<Block> blocks;
  void *get(size_t bytes) {
    std::lock_guard<std::mutex> lk(m);
    size_t best = SIZE_MAX;
    for (size_t i = 0; i < blocks.size(); i++)
      if (!blocks[i].in_use && blocks[i].cap >= bytes && (best == SIZE_MAX || blocks[i].cap < blocks[best].cap))
        best = i;
    if (best != SIZE_MAX) {
      blocks[best].in_use = true;
      return blocks[best].p;
    }
    const size_t cap = bytes + bytes / 8 + 64;
    // big blocks (bases, qualities, identifiers of a batch) come from the library's page-locked allocator
    // once a GPU context has installed it (workers.hpp: big_alloc_hook), so they reach the device by DMA
    const BigAlloc *hook = big_alloc_hook().load(std::memory_order_acquire);
    void *p = nullptr;
    void (*rel)(void *, size_t) = nullptr;
    if (hook && hook->alloc && cap >= (8u << 20)) {
      p = hook->alloc(cap);
      if (p) rel = hook->release;
    }
    if (!p) {
      p = malloc(cap);
      if (!p) fail(KSLAM_ERR_OOM, "out of host memory for the read columns");
      advise_huge(p, cap);   // the tail walks these columns per alignment
    }
    // replace a parked block that was too small, so the cache does not grow without bound
    for (auto &b : blocks)
      if (!b.in_use) {
        drop(b);
        b = Block{p, cap, true, rel};
        return p;
      }
    blocks.push_back(Block{p, cap, true, rel});
    return p;
  }
  void put(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(m);
    size_t parked = 0;
    for (auto &b : blocks) parked += !b.in_use;
    for (size_t i = 0; i < blocks.size(); i++)
      if (blocks[i].p == p) {
        if (parked >= 12) {  // two batches' worth of columns is plenty
          drop(blocks[i]);
          blocks.erase(blocks.begin() + i);
        } else
          blocks[i].in_use = false;
        return;
      }
    free(p);
  }
};
BlockCache &cache() {
  static BlockCache *c = new BlockCache();
  return *c;
}

template <typename T>
T *alloc(uint64_t n) {
  return (T *)cache().get(sizeof(T) * (n + 1));
}

void free_columns(kslam_reads_columns *c) {
  cache().put(c->bases);
  cache().put(c->bases_off);
  cache().put(c->quality);
  cache().put(c->quality_off);
  cache().put(c->ids);
  cache().put(c->ids_off);
  memset(c, 0, sizeof *c);
}

// streams[k]'s records become reads [first[k], first[k] + n_k) of the batch
void emit_columns(const std::vector<const StreamIndex *> &streams, int threads, kslam_reads_columns *out) {
  uint64_t n = 0;
  for (auto s : streams) n += s->n;
  memset(out, 0, sizeof *out);
  out->n_reads = n;
  try {
    out->bases_off = alloc<uint64_t>(n + 1);
    out->quality_off = alloc<uint64_t>(n + 1);
    out->ids_off = alloc<uint64_t>(n + 1);
    uint64_t b = 0, q = 0, i = 0, r = 0;
    for (auto s : streams)
      for (uint64_t k = 0; k < s->n; k++, r++) {
        out->bases_off[r] = b;
        out->quality_off[r] = q;
        out->ids_off[r] = i;
        b += s->bases[k].len;
        q += s->qual[k].len;
        i += s->id[k].len;
      }
    out->bases_off[n] = b;
    out->quality_off[n] = q;
    out->ids_off[n] = i;
    out->bases = alloc<char>(b + 64);  // slack: the device upload reads whole words
    out->quality = alloc<char>(q + 1);
    out->ids = alloc<char>(i + 1);
    memset(out->bases + b, 0, 64);
    out->quality[q] = 0;
    out->ids[i] = 0;
    uint64_t first = 0;
    for (auto s : streams) {
      const uint64_t grain = 8192, n_tasks = (s->n + grain - 1) / grain;
      Pool::get().tasks(threads, n_tasks, [&](size_t t) {
        for (uint64_t k = t * grain; k < std::min(s->n, (t + 1) * grain); k++) {
          const uint64_t rr = first + k;
          memcpy(out->bases + out->bases_off[rr], s->text + s->bases[k].start, s->bases[k].len);
          memcpy(out->quality + out->quality_off[rr], s->text + s->qual[k].start, s->qual[k].len);
          memcpy(out->ids + out->ids_off[rr], s->text + s->id[k].start, s->id[k].len);
        }
      });
      first += s->n;
    }
  } catch (...) {
    free_columns(out);
    throw;
  }
}

int thread_count(int threads) { return std::max(1, std::min(threads > 0 ? threads : usable_cpus(), 512)); }

}  // namespace

extern "C" {

kslam_status kslam_fastq_parse(const char *text, uint64_t len, uint64_t max_reads, int at_eof, int threads,
                               kslam_reads_columns *out, uint64_t *consumed) {
  return guarded([&] {
    if (!out) fail(KSLAM_ERR_ARG, "null output argument");
    memset(out, 0, sizeof *out);
    const int nt = thread_count(threads);
    std::lock_guard<std::mutex> one(g_parse_call);
    StreamIndex ix;
    index_stream(text, len, max_reads, at_eof != 0, nt, g_arena[0], ix);
    emit_columns({&ix}, nt, out);
    if (consumed) *consumed = ix.consumed;
  });
}

kslam_status kslam_fastq_parse_pair(const char *r1, uint64_t len1, const char *r2, uint64_t len2,
                                    uint64_t max_pairs, int at_eof, int threads, kslam_reads_columns *out,
                                    uint64_t *consumed1, uint64_t *consumed2) {
  return guarded([&] {
    if (!out) fail(KSLAM_ERR_ARG, "null output argument");
    memset(out, 0, sizeof *out);
    const int nt = thread_count(threads);
    std::lock_guard<std::mutex> one(g_parse_call);
    StreamIndex a, b;
    index_stream(r1, len1, max_pairs, at_eof != 0, nt, g_arena[0], a);
    index_stream(r2, len2, max_pairs, at_eof != 0, nt, g_arena[1], b);
    if (a.n != b.n) fail(KSLAM_ERR_ARG, "mismatch in R1 and R2 size");  // src/FASTQsequence.h:118-122
    emit_columns({&a, &b}, nt, out);
    if (consumed1) *consumed1 = a.consumed;
    if (consumed2) *consumed2 = b.consumed;
  });
}

void kslam_reads_free(kslam_reads_columns *cols) {
  if (cols) free_columns(cols);
}

kslam_status kslam_fastq_index_pair(const char *r1, uint64_t len1, const char *r2, uint64_t len2,
                                    uint64_t max_pairs, int at_eof, int threads, kslam_reads_columns *out,
                                    kslam_fastq_layout *layout, uint64_t *consumed1, uint64_t *consumed2) {
  return guarded([&] {
    if (!out || !layout) fail(KSLAM_ERR_ARG, "null output argument");
    memset(out, 0, sizeof *out);
    memset(layout, 0, sizeof *layout);
    const int nt = thread_count(threads);
    std::lock_guard<std::mutex> one(g_parse_call);
    StreamIndex a, b;
    index_stream(r1, len1, max_pairs, at_eof != 0, nt, g_arena[0], a);
    index_stream(r2, len2, max_pairs, at_eof != 0, nt, g_arena[1], b);
    if (a.n != b.n) fail(KSLAM_ERR_ARG, "mismatch in R1 and R2 size");  // src/FASTQsequence.h:118-122
    const uint64_t n = a.n + b.n;
    try {
      out->n_reads = n;
      out->bases_off = alloc<uint64_t>(n + 1);
      out->quality_off = alloc<uint64_t>(n + 1);
      out->ids_off = alloc<uint64_t>(n + 1);
      layout->n_reads = n;
      layout->bases_at = alloc<uint64_t>(n + 1);
      layout->quality_at = alloc<uint64_t>(n + 1);
      uint64_t bsum = 0, isum = 0, r = 0;
      for (const StreamIndex *s : {&a, &b}) {
        const uint64_t shift = s == &a ? 0 : len1;
        for (uint64_t k = 0; k < s->n; k++, r++) {
          if (s->qual[k].len != s->bases[k].len)
            fail(KSLAM_ERR_ARG, "a read's quality line is not as long as its bases line");
          out->bases_off[r] = bsum;
          out->quality_off[r] = bsum;
          out->ids_off[r] = isum;
          layout->bases_at[r] = shift + s->bases[k].start;
          layout->quality_at[r] = shift + s->qual[k].start;
          bsum += s->bases[k].len;
          isum += s->id[k].len;
        }
      }
      out->bases_off[n] = out->quality_off[n] = bsum;
      out->ids_off[n] = isum;
      out->ids = alloc<char>(isum + 1);
      out->ids[isum] = 0;
      uint64_t first = 0;
      for (const StreamIndex *s : {&a, &b}) {
        const uint64_t grain = 16384, n_tasks = (s->n + grain - 1) / grain;
        Pool::get().tasks(nt, n_tasks, [&](size_t t) {
          for (uint64_t k = t * grain; k < std::min(s->n, (t + 1) * grain); k++)
            memcpy(out->ids + out->ids_off[first + k], s->text + s->id[k].start, s->id[k].len);
        });
        first += s->n;
      }
    } catch (...) {
      free_columns(out);
      kslam_fastq_layout_free(layout);
      throw;
    }
    if (consumed1) *consumed1 = a.consumed;
    if (consumed2) *consumed2 = b.consumed;
  });
}

// Where the batch loop of src/SLAM.h:193-207 would leave a stream after one more batch: the byte after the
// max_records-th record's fourth line.  Only terminators are COUNTED here (no fields, no index): 64 MiB of
// text per round over all workers, stopping in the round that holds the last line wanted.
kslam_status kslam_fastq_batch_end(const char *text, uint64_t len, uint64_t max_records, int at_eof, int threads,
                                   uint64_t *end, int *complete) {
  return guarded([&] {
    if (!end || !complete) fail(KSLAM_ERR_ARG, "null output argument");
    if (len && !text) fail(KSLAM_ERR_ARG, "null text");
    *end = len;
    *complete = 0;
    if (max_records == 0) { *complete = at_eof != 0; return; }
    uint64_t scan_len = len;   // (a trailing "\r" of a prefix may be half of "\r\n": index_stream's rule)
    if (!at_eof && len && text[len - 1] == '\r') scan_len = len - 1;
    const int nt = thread_count(threads);
    const uint64_t want = 4 * max_records, chunk = 1 << 20, round_chunks = 64;
    uint64_t have = 0;
    for (uint64_t base = 0; base < scan_len; base += chunk * round_chunks) {
      const uint64_t stop = std::min(scan_len, base + chunk * round_chunks);
      const size_t n_chunks = (size_t)((stop - base + chunk - 1) / chunk);
      std::vector<uint64_t> count(n_chunks, 0);
      Pool::get().tasks(nt, n_chunks, [&](size_t c) {
        const uint64_t lo = base + c * chunk, hi = std::min(stop, lo + chunk);
        uint64_t k = 0, p = lo;
#if defined(__SSE2__)
        // 16 bytes at a time: the positions of LF / CR as a bit mask, visited one by one (a line end every 40-150 bytes)
        const __m128i lf = _mm_set1_epi8('\n'), cr = _mm_set1_epi8('\r');
        for (; p + 16 <= hi; p += 16) {
          const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(text + p));
          unsigned m = (unsigned)_mm_movemask_epi8(_mm_or_si128(_mm_cmpeq_epi8(v, lf), _mm_cmpeq_epi8(v, cr)));
          while (m) {
            const unsigned b = (unsigned)__builtin_ctz(m);
            m &= m - 1;
            if (is_event(text, p + b)) k++;
          }
        }
#endif
        for (; p < hi; p++)
          if ((text[p] == '\n' || text[p] == '\r') && is_event(text, p)) k++;
        count[c] = k;
      });
      for (size_t c = 0; c < n_chunks; c++) {
        if (have + count[c] >= want) {   // the line wanted ends in this chunk
          const uint64_t lo = base + c * chunk, hi = std::min(stop, lo + chunk);
          for (uint64_t p = lo; p < hi; p++)
            if ((text[p] == '\n' || text[p] == '\r') && is_event(text, p) && ++have == want) {
              *end = line_after(text, len, p);
              *complete = 1;
              return;
            }
          fail(KSLAM_ERR_INTERNAL, "kslam_fastq_batch_end: count and locate disagree");
        }
        have += count[c];
      }
    }
    // fewer than 4 x max_records terminated lines: at the true end of the stream the rest is the last batch
    // (the reference reads on to the end: index_stream's `consumed = len`); of a prefix, more bytes are needed
    *complete = at_eof != 0;
  });
}

void kslam_fastq_layout_free(kslam_fastq_layout *layout) {
  if (!layout) return;
  cache().put(layout->bases_at);
  cache().put(layout->quality_at);
  memset(layout, 0, sizeof *layout);
}
}
