/*
 * kslam_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See kslam_oracle.h for the scope statement and the parity-pinning status.
 *
 * Plain-C restatement of the reference hot path alignToDatabase()
 * (reference src/SLAM.h:59-79).  All file:line citations are into
 * /root/reference/.  Nothing here is copied from the reference: the SSE2
 * kernels of src/ssw.c are restated as scalar loops over an explicit
 * [segment][lane] layout so that the striped evaluation order (which the
 * Lazy-F shortcut makes observable) is reproduced exactly.
 */
#define _GNU_SOURCE
#include "kslam_oracle.h"
#include <dlfcn.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_K 32u /* src/Globals.h:25 */

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------------
 * a-2: getTwoBits, src/KMer.h:246-268.  A=0 C=1 T=2 G=3, anything else 0.
 * ---------------------------------------------------------------------- */
static inline uint64_t two_bits(char b) {
  switch (b) {
    case 'A': return 0;
    case 'C': return 1;
    case 'T': return 2;
    case 'G': return 3;
    default: return 0;
  }
}

/* number of k-mers of one sequence, src/KMer.h:203 */
uint64_t orc_count_kmers(uint64_t len, unsigned gap) {
  return len >= ORC_K ? (len - ORC_K) / gap + 1 : 0;
}

/* ------------------------------------------------------------------------
 * a-3: splitIntoKMersAndAddToVector, src/KMer.h:160-181 with
 * addBaseToKMers src/KMer.h:272-280 (K = 32 so the mask is all ones).
 * ---------------------------------------------------------------------- */
uint64_t orc_extract_kmers(const char *bases, uint64_t len, int is_gb,
                           uint32_t id, unsigned gap, orc_kmer_rec *out) {
  uint64_t kmer = 0, rc = 0, n = 0;
  if (len < ORC_K) return 0; /* :167 */
  for (uint64_t i = 0; i < len; i++) {
    uint64_t b = two_bits(bases[i]);
    kmer = (kmer << 2) | b;                   /* :274-278, mask = ~0 */
    rc = (rc >> 2) | ((b ^ 2u) << (2 * (ORC_K - 1))); /* :275,279 */
    if (i < ORC_K - 1) continue;              /* :171 */
    uint64_t pos = i - (ORC_K - 1);
    if (pos % gap != 0) continue;             /* :172 */
    orc_kmer_rec r;
    uint32_t idbits = id & 0x3FFFFFFFu;       /* :65 */
    if (kmer < rc) {                          /* :173 forward wins only if < */
      r.kmer = kmer;
      r.meta = idbits | ((uint32_t)(is_gb != 0) << 31);
      r.offset = (uint32_t)pos;
    } else {                                  /* palindromes land here */
      r.kmer = rc;
      r.meta = idbits | ((uint32_t)(is_gb != 0) << 31) | (1u << 30);
      r.offset = (uint32_t)(is_gb ? pos : len - 1 - i); /* :176 */
    }
    out[n++] = r;
  }
  return n;
}

/* getKMers_parallel, src/KMer.h:190-241: records land in entry order at
 * deterministic positions whatever the thread count. */
uint64_t orc_extract_all(uint64_t n, const char *const *bases,
                         const uint64_t *lens, int is_gb, unsigned gap,
                         orc_kmer_rec *out) {
  uint64_t *start = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
  uint64_t tot = 0;
  for (uint64_t i = 0; i < n; i++) {
    start[i] = tot;
    tot += orc_count_kmers(lens[i], gap);
  }
  start[n] = tot;
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < (int64_t)n; i++)
    orc_extract_kmers(bases[i], lens[i], is_gb, (uint32_t)i, gap,
                      out + start[i]);
  free(start);
  return tot;
}

/* ------------------------------------------------------------------------
 * a-4: sortKMers, src/KMer.h:388-398: kMerInt ascending, then
 * ID_isFromGB_RC DESCENDING.  The reference sort is unstable and offset is
 * not part of its key; the oracle adds offset ascending to make the order
 * total (exact ties are reference-nondeterministic).
 * ---------------------------------------------------------------------- */
static int cmp_kmer(const void *a, const void *b) {
  const orc_kmer_rec *x = (const orc_kmer_rec *)a, *y = (const orc_kmer_rec *)b;
  if (x->kmer != y->kmer) return x->kmer < y->kmer ? -1 : 1;
  if (x->meta != y->meta) return x->meta > y->meta ? -1 : 1;
  if (x->offset != y->offset) return x->offset < y->offset ? -1 : 1;
  return 0;
}

/* bottom-up merge of two sorted runs */
static void merge_runs(const char *a, size_t na, const char *b, size_t nb,
                       char *dst, size_t sz,
                       int (*cmp)(const void *, const void *)) {
  size_t i = 0, j = 0, o = 0;
  while (i < na && j < nb) {
    if (cmp(b + j * sz, a + i * sz) < 0) memcpy(dst + (o++) * sz, b + (j++) * sz, sz);
    else memcpy(dst + (o++) * sz, a + (i++) * sz, sz);
  }
  if (i < na) memcpy(dst + o * sz, a + i * sz, (na - i) * sz);
  o += na - i;
  if (j < nb) memcpy(dst + o * sz, b + j * sz, (nb - j) * sz);
}

/* qsort per chunk in parallel, then pairwise merges (comparators are total) */
static void par_sort(void *base, size_t n, size_t sz,
                     int (*cmp)(const void *, const void *)) {
  int nt = orc_num_threads();
  if (n < 1u << 16 || nt < 2) {
    qsort(base, n, sz, cmp);
    return;
  }
  int chunks = 1;
  while (chunks < nt) chunks <<= 1;
  size_t *bound = (size_t *)malloc((chunks + 1) * sizeof(size_t));
  for (int c = 0; c <= chunks; c++) bound[c] = n * (size_t)c / chunks;
#pragma omp parallel for schedule(dynamic, 1)
  for (int c = 0; c < chunks; c++)
    qsort((char *)base + bound[c] * sz, bound[c + 1] - bound[c], sz, cmp);
  char *tmp = (char *)malloc(n * sz);
  char *src = (char *)base, *dst = tmp;
  for (int width = 1; width < chunks; width <<= 1) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int c = 0; c < chunks; c += 2 * width) {
      size_t lo = bound[c], mid = bound[c + width], hi = bound[c + 2 * width];
      merge_runs(src + lo * sz, mid - lo, src + mid * sz, hi - mid,
                 dst + lo * sz, sz, cmp);
    }
    char *t = src; src = dst; dst = t;
  }
  if (src != (char *)base) memcpy(base, src, n * sz);
  free(tmp);
  free(bound);
}

void orc_sort_kmers(orc_kmer_rec *recs, uint64_t n) {
  par_sort(recs, n, sizeof(orc_kmer_rec), cmp_kmer);
}

/* ------------------------------------------------------------------------
 * a-5: findOverlaps (src/Overlap.h:230-246) + processPileUp (:153-199).
 * One scanner with an optional sink: count-only when out == NULL.
 * ---------------------------------------------------------------------- */
static inline int is_gb(const orc_kmer_rec *r) { return (r->meta >> 31) & 1; }
static inline int is_rc(const orc_kmer_rec *r) { return (r->meta >> 30) & 1; }
static inline uint32_t rec_id(const orc_kmer_rec *r) { return r->meta & 0x3FFFFFFFu; }

static uint64_t scan_pileups(const orc_kmer_rec *s, uint64_t n,
                             const uint64_t *read_lens, orc_overlap *out) {
  uint64_t cnt = 0, first = 0;
  while (first < n) {
    if (s[first].kmer == 0) { first++; continue; }          /* :236-239 */
    /* std::adjacent_find on kMerInt, :240-242 */
    uint64_t j = first;
    while (j + 1 < n && s[j].kmer != s[j + 1].kmer) j++;
    if (j + 1 >= n) break; /* adjacent_find returned last */
    first = j;
    /* processPileUp(first, last) */
    if (!is_gb(&s[first])) {                                /* :157-162 */
      uint64_t r = first;
      while (r < n && s[r].kmer == s[first].kmer) r++;
      first = r;
      continue;
    }
    uint64_t r = first;
    for (; r < n; r++) {                                    /* :163 */
      if (s[r].kmer != s[first].kmer) break;                /* :164 */
      if (is_gb(&s[r])) continue;                           /* :165 */
      for (uint64_t g = first; g != r; g++) {               /* :175 */
        if (!is_gb(&s[g])) break;                           /* :176 */
        if (out) {
          int same = is_rc(&s[g]) == is_rc(&s[r]);          /* :177-178 */
          uint32_t off = !is_rc(&s[g])                      /* :179-183 */
                             ? s[r].offset
                             : (uint32_t)(read_lens[rec_id(&s[r])] -
                                          s[r].offset - ORC_K);
          orc_overlap o;
          memset(&o, 0, sizeof o);
          o.read = rec_id(&s[r]);
          o.entry = rec_id(&s[g]);
          o.rel = (int32_t)(s[g].offset - off);             /* :186 u32 wrap */
          o.revcomp = (uint8_t)!same;
          out[cnt] = o;
        }
        cnt++;
      }
    }
    first = r;
  }
  return cnt;
}

uint64_t orc_count_overlaps(const orc_kmer_rec *sorted, uint64_t n) {
  return scan_pileups(sorted, n, NULL, NULL);
}
/* the pre-dedupe list in emission order (findOverlaps over the whole range) */
uint64_t orc_scan_overlaps(const orc_kmer_rec *sorted, uint64_t n,
                           const uint64_t *read_lens, orc_overlap *out) {
  return scan_pileups(sorted, n, read_lens, out);
}

/* overlapSort, src/Overlap.h:87-98: (read, entry, rel); revComp is NOT in the
 * reference key -- the oracle appends it (false first) to make ties total. */
static int cmp_overlap(const void *a, const void *b) {
  const orc_overlap *x = (const orc_overlap *)a, *y = (const orc_overlap *)b;
  if (x->read != y->read) return x->read < y->read ? -1 : 1;
  if (x->entry != y->entry) return x->entry < y->entry ? -1 : 1;
  if (x->rel != y->rel) return x->rel < y->rel ? -1 : 1;
  if (x->revcomp != y->revcomp) return x->revcomp < y->revcomp ? -1 : 1;
  return 0;
}

/* a-6: findOverlaps_parallel, src/Overlap.h:277-295: sort, then std::unique
 * with overlapEqual (:79-85) -- each element is compared with the LAST KEPT
 * one (libstdc++ std::unique), |delta rel| < 3. */
uint64_t orc_find_overlaps(const orc_kmer_rec *sorted, uint64_t n,
                           const uint64_t *read_lens, orc_overlap *out,
                           uint64_t *n_raw) {
  uint64_t m = scan_pileups(sorted, n, read_lens, out);
  if (n_raw) *n_raw = m;
  par_sort(out, m, sizeof(orc_overlap), cmp_overlap);
  if (m == 0) return 0;
  uint64_t kept = 0;
  for (uint64_t i = 1; i < m; i++) {
    const orc_overlap *a = &out[kept], *b = &out[i];
    int64_t d = (int64_t)a->rel - (int64_t)b->rel;
    if (d < 0) d = -d;
    int eq = a->read == b->read && a->entry == b->entry && d < 3;
    if (!eq) out[++kept] = *b;
  }
  return kept + 1;
}

/* ------------------------------------------------------------------------
 * a-9: ssw_cpp wrapper pieces, src/ssw_cpp.cpp:11-49.
 * kBaseTranslation: A/a 0, C/c 1, G/g 2, T/t 3, U/u 0, everything else 4.
 * ---------------------------------------------------------------------- */
static inline int8_t translate_base(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case 'U': case 'u': return 0; /* table slot 85/117 holds 0, :17,21 */
    default: return 4;
  }
}
void orc_translate(const char *s, int32_t n, int8_t *out) {
  for (int32_t i = 0; i < n; i++) out[i] = translate_base(s[i]);
}
/* BuildSwScoreMatrix, src/ssw_cpp.cpp:25-49 (arguments arrive as uint8_t) */
void orc_build_matrix(uint32_t match, uint32_t mismatch, int8_t mat[25]) {
  uint8_t m = (uint8_t)match, x = (uint8_t)mismatch;
  int id = 0;
  for (int i = 0; i < 4; i++) {
    for (int j = 0; j < 4; j++) mat[id++] = (i == j) ? (int8_t)m : (int8_t)(-x);
    mat[id++] = 0;
  }
  for (int i = 0; i < 5; i++) mat[id++] = 0;
}

/* ------------------------------------------------------------------------
 * a-10/a-11: striped SW, scalar emulation with an explicit [seg][lane] layout.
 * lanes = 16 -> sw_sse2_byte (src/ssw.c:143-383, profile qP_byte :105-133)
 * lanes = 8  -> sw_sse2_word (src/ssw.c:408-592, profile qP_word :385-406)
 * ---------------------------------------------------------------------- */
typedef struct { uint16_t score; int32_t ref; int32_t read; } sw_end;

static inline int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

static sw_end striped_byte(const int8_t *ref, int ref_dir, int32_t refLen,
                           const int8_t *read, int32_t readLen,
                           const int8_t *mat, int32_t n, uint8_t gapO,
                           uint8_t gapE, uint8_t terminate, uint8_t bias) {
  enum { W = 16 };
  int32_t segLen = (readLen + W - 1) / W;
  size_t vsz = (size_t)(segLen > 0 ? segLen : 1) * W;
  uint8_t *prof = (uint8_t *)malloc((size_t)n * vsz);
  /* qP_byte :116-131 */
  for (int32_t nt = 0; nt < n; nt++)
    for (int32_t i = 0; i < segLen; i++)
      for (int32_t l = 0; l < W; l++) {
        int32_t j = i + l * segLen;
        prof[(size_t)nt * vsz + (size_t)i * W + l] =
            (uint8_t)(int8_t)(j >= readLen ? bias : mat[nt * n + read[j]] + bias);
      }
  uint8_t *HStore = (uint8_t *)calloc(vsz, 1), *HLoad = (uint8_t *)calloc(vsz, 1);
  uint8_t *E = (uint8_t *)calloc(vsz, 1), *Hmax = (uint8_t *)calloc(vsz, 1);
  uint8_t vMaxScore[W] = {0}, vMaxMark[W] = {0};
  uint8_t max = 0;
  int32_t end_read = readLen - 1, end_ref = -1;           /* :166-169 */
  int32_t begin = 0, end = refLen, step = 1;
  if (ref_dir == 1) { begin = refLen - 1; end = -1; step = -1; } /* :209-213 */
  for (int32_t i = begin; i != end; i += step) {
    uint8_t vF[W] = {0}, vMaxColumn[W] = {0}, vH[W], e[W];
    /* vH = pvHStore[segLen-1] << 1 lane, :225-226 */
    vH[0] = 0;
    for (int l = 1; l < W; l++) vH[l] = HStore[(size_t)(segLen - 1) * W + l - 1];
    const uint8_t *vP = prof + (size_t)ref[i] * vsz;      /* :228 */
    uint8_t *pv = HLoad; HLoad = HStore; HStore = pv;     /* :231-233 */
    for (int32_t j = 0; j < segLen; j++) {                /* :236-271 */
      for (int l = 0; l < W; l++) {
        int h = sat_u8(vH[l] + vP[(size_t)j * W + l]);    /* adds_epu8 */
        h = sat_u8(h - bias);                             /* subs_epu8 */
        e[l] = E[(size_t)j * W + l];
        if (e[l] > h) h = e[l];
        if (vF[l] > h) h = vF[l];
        if (h > vMaxColumn[l]) vMaxColumn[l] = (uint8_t)h;
        HStore[(size_t)j * W + l] = (uint8_t)h;
        int hg = sat_u8(h - gapO);
        int ee = sat_u8(e[l] - gapE);
        if (hg > ee) ee = hg;
        E[(size_t)j * W + l] = (uint8_t)ee;
        int ff = sat_u8(vF[l] - gapE);
        if (hg > ff) ff = hg;
        vF[l] = (uint8_t)ff;
        vH[l] = HLoad[(size_t)j * W + l];
      }
    }
    /* Lazy_F, :274-305 */
    int32_t j = 0;
    for (int l = 0; l < W; l++) vH[l] = HStore[l];
    for (int l = W - 1; l > 0; l--) vF[l] = vF[l - 1];
    vF[0] = 0;
    for (;;) {
      int all = 1;
      for (int l = 0; l < W; l++) {
        int t = sat_u8(vH[l] - gapO);
        t = sat_u8(vF[l] - t);
        if (t != 0) all = 0;
      }
      if (all) break;
      for (int l = 0; l < W; l++) {
        if (vF[l] > vH[l]) vH[l] = vF[l];
        if (vH[l] > vMaxColumn[l]) vMaxColumn[l] = vH[l];
        HStore[(size_t)j * W + l] = vH[l];
        vF[l] = (uint8_t)sat_u8(vF[l] - gapE);
      }
      j++;
      if (j >= segLen) {
        j = 0;
        for (int l = W - 1; l > 0; l--) vF[l] = vF[l - 1];
        vF[0] = 0;
      }
      for (int l = 0; l < W; l++) vH[l] = HStore[(size_t)j * W + l];
    }
    /* :307-325 */
    int differs = 0;
    for (int l = 0; l < W; l++) {
      if (vMaxColumn[l] > vMaxScore[l]) vMaxScore[l] = vMaxColumn[l];
      if (vMaxMark[l] != vMaxScore[l]) differs = 1;
    }
    if (differs) {
      uint8_t temp = 0;
      for (int l = 0; l < W; l++) {
        vMaxMark[l] = vMaxScore[l];
        if (vMaxScore[l] > temp) temp = vMaxScore[l];
      }
      if (temp > max) {
        max = temp;
        if (max + bias >= 255) break;                     /* :318 overflow */
        end_ref = i;
        memcpy(Hmax, HStore, vsz);
      }
    }
    uint8_t colmax = 0;
    for (int l = 0; l < W; l++) if (vMaxColumn[l] > colmax) colmax = vMaxColumn[l];
    if (colmax == terminate) break;                       /* :330 */
  }
  /* :334-342 */
  for (int32_t i = 0; i < segLen * W; i++)
    if (Hmax[i] == max) {
      int32_t temp = i / W + i % W * segLen;
      if (temp < end_read) end_read = temp;
    }
  free(prof); free(HStore); free(HLoad); free(E); free(Hmax);
  sw_end r;
  r.score = (uint16_t)(max + bias >= 255 ? 255 : max);    /* :351 */
  r.ref = end_ref;
  r.read = end_read;
  return r;
}

static inline uint16_t subs_u16(uint16_t a, uint16_t b) { return a > b ? (uint16_t)(a - b) : 0; }
static inline int16_t adds_i16(int16_t a, int16_t b) {
  int v = (int)a + (int)b;
  return (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
}

static sw_end striped_word(const int8_t *ref, int ref_dir, int32_t refLen,
                           const int8_t *read, int32_t readLen,
                           const int8_t *mat, int32_t n, uint8_t gapO,
                           uint8_t gapE, uint16_t terminate) {
  enum { W = 8 };
  int32_t segLen = (readLen + W - 1) / W;
  size_t vsz = (size_t)(segLen > 0 ? segLen : 1) * W;
  int16_t *prof = (int16_t *)malloc((size_t)n * vsz * sizeof(int16_t));
  for (int32_t nt = 0; nt < n; nt++)                      /* qP_word :397-404 */
    for (int32_t i = 0; i < segLen; i++)
      for (int32_t l = 0; l < W; l++) {
        int32_t j = i + l * segLen;
        prof[(size_t)nt * vsz + (size_t)i * W + l] =
            (int16_t)(j >= readLen ? 0 : mat[nt * n + read[j]]);
      }
  int16_t *HStore = (int16_t *)calloc(vsz, 2), *HLoad = (int16_t *)calloc(vsz, 2);
  int16_t *E = (int16_t *)calloc(vsz, 2), *Hmax = (int16_t *)calloc(vsz, 2);
  int16_t vMaxScore[W] = {0}, vMaxMark[W] = {0};
  uint16_t max = 0;
  int32_t end_read = readLen - 1, end_ref = 0;            /* :422-425 */
  int32_t begin = 0, end = refLen, step = 1;
  if (ref_dir == 1) { begin = refLen - 1; end = -1; step = -1; }
  for (int32_t i = begin; i != end; i += step) {
    int16_t vF[W] = {0}, vMaxColumn[W] = {0}, vH[W], e[W];
    vH[0] = 0;
    for (int l = 1; l < W; l++) vH[l] = HStore[(size_t)(segLen - 1) * W + l - 1];
    const int16_t *vP = prof + (size_t)ref[i] * vsz;
    int16_t *pv = HLoad; HLoad = HStore; HStore = pv;
    for (int32_t j = 0; j < segLen; j++) {                /* :486-510 */
      for (int l = 0; l < W; l++) {
        int16_t h = adds_i16(vH[l], vP[(size_t)j * W + l]);
        e[l] = E[(size_t)j * W + l];
        if (e[l] > h) h = e[l];
        if (vF[l] > h) h = vF[l];
        if (h > vMaxColumn[l]) vMaxColumn[l] = h;
        HStore[(size_t)j * W + l] = h;
        int16_t hg = (int16_t)subs_u16((uint16_t)h, gapO);
        int16_t ee = (int16_t)subs_u16((uint16_t)e[l], gapE);
        if (hg > ee) ee = hg;
        E[(size_t)j * W + l] = ee;
        int16_t ff = (int16_t)subs_u16((uint16_t)vF[l], gapE);
        if (hg > ff) ff = hg;
        vF[l] = ff;
        vH[l] = HLoad[(size_t)j * W + l];
      }
    }
    /* Lazy_F, :514-526: vMaxColumn is NOT refreshed here (unlike byte) */
    for (int k = 0, done = 0; k < W && !done; k++) {
      for (int l = W - 1; l > 0; l--) vF[l] = vF[l - 1];
      vF[0] = 0;
      for (int32_t j = 0; j < segLen; j++) {
        int any = 0;
        for (int l = 0; l < W; l++) {
          int16_t h = HStore[(size_t)j * W + l];
          if (vF[l] > h) h = vF[l];
          HStore[(size_t)j * W + l] = h;
          int16_t hg = (int16_t)subs_u16((uint16_t)h, gapO);
          vF[l] = (int16_t)subs_u16((uint16_t)vF[l], gapE);
          if (vF[l] > hg) any = 1;
        }
        if (!any) { done = 1; break; }
      }
    }
    int differs = 0;                                      /* :528-543 */
    for (int l = 0; l < W; l++) {
      if (vMaxColumn[l] > vMaxScore[l]) vMaxScore[l] = vMaxColumn[l];
      if (vMaxMark[l] != vMaxScore[l]) differs = 1;
    }
    if (differs) {
      int16_t temp = vMaxScore[0];
      for (int l = 0; l < W; l++) {
        vMaxMark[l] = vMaxScore[l];
        if (vMaxScore[l] > temp) temp = vMaxScore[l];
      }
      if ((uint16_t)temp > max) {
        max = (uint16_t)temp;
        end_ref = i;
        memcpy(Hmax, HStore, vsz * sizeof(int16_t));
      }
    }
    int16_t colmax = vMaxColumn[0];
    for (int l = 0; l < W; l++) if (vMaxColumn[l] > colmax) colmax = vMaxColumn[l];
    if ((uint16_t)colmax == terminate) break;             /* :545 */
  }
  for (int32_t i = 0; i < segLen * W; i++)                /* :549-557 */
    if ((uint16_t)Hmax[i] == max) {
      int32_t temp = i / W + i % W * segLen;
      if (temp < end_read) end_read = temp;
    }
  free(prof); free(HStore); free(HLoad); free(E); free(Hmax);
  sw_end r;
  r.score = max;
  r.ref = end_ref;
  r.read = end_read;
  return r;
}

/* ------------------------------------------------------------------------
 * The SPEC the HIP kernels implement: plain column-major Gotoh,
 *   H = max(0, Hdiag + s, E, F); E' = max(0, E-gE, H-gO); F' likewise,
 * end_ref = first column (scan order) whose column max strictly exceeds the
 * running max, end_read = smallest read index with H == max in that column,
 * stop after the column whose max == terminate.  Equal to the striped code
 * whenever a diagonal step is never worse than an adjacent gap pair
 * (mismatch <= gapO + min(gapO, gapE)); tests/ checks this against the
 * striped emulation and against the real ssw.c.
 * ---------------------------------------------------------------------- */
static sw_end plain_pass(const int8_t *ref, int ref_dir, int32_t refLen,
                         const int8_t *read, int32_t readLen,
                         const int8_t *mat, int32_t n, int32_t gapO,
                         int32_t gapE, int32_t terminate) {
  int32_t *H = (int32_t *)calloc((size_t)readLen + 1, sizeof(int32_t));
  int32_t *E = (int32_t *)calloc((size_t)readLen + 1, sizeof(int32_t));
  int32_t max = 0, end_ref = 0, end_read = readLen - 1;
  int32_t begin = 0, end = refLen, step = 1;
  if (ref_dir == 1) { begin = refLen - 1; end = -1; step = -1; }
  for (int32_t c = begin; c != end; c += step) {
    int32_t F = 0, diag = 0, colmax = 0, colrow = 0;
    const int8_t *mrow = mat + ref[c] * n;
    for (int32_t i = 0; i < readLen; i++) {
      int32_t h = diag + mrow[read[i]];
      if (E[i] > h) h = E[i];
      if (F > h) h = F;
      if (h < 0) h = 0;
      diag = H[i];
      H[i] = h;
      if (h > colmax) { colmax = h; colrow = i; }
      int32_t t = h - gapO; if (t < 0) t = 0;
      int32_t e = E[i] - gapE; if (e < 0) e = 0;
      E[i] = e > t ? e : t;
      int32_t f = F - gapE; if (f < 0) f = 0;
      F = f > t ? f : t;
    }
    if (colmax > max) { max = colmax; end_ref = c; end_read = colrow; }
    if (colmax == terminate) break;
  }
  free(H); free(E);
  sw_end r;
  r.score = (uint16_t)max;
  r.ref = end_ref;
  r.read = end_read;
  return r;
}

/* ------------------------------------------------------------------------
 * Second formulation of the same SPEC, the one the HIP kernel actually runs:
 * ONE forward pass that carries, next to every score, the start cell of the
 * alignment it belongs to ("origin"), packed as score * 2^18 + (col << 9 | row)
 * so that an integer max is a lexicographic (score, col, row) max.  Among all
 * optimal alignments ending at (end_ref, end_read) it therefore reports the one
 * starting at the largest column, then the largest row -- exactly what the
 * reference's reverse pass finds (first column from the right whose maximum
 * equals the score, then the smallest reversed read index, ssw.c:906-923).
 * A cell whose score is 0 stores the key of its diagonal successor, so a fresh
 * alignment picks up its own first cell as origin.
 * ---------------------------------------------------------------------- */
static void origin_pass(const int8_t *ref, int32_t refLen, const int8_t *read,
                        int32_t readLen, const int8_t *mat, int32_t n,
                        int32_t gapO, int32_t gapE, int32_t *score,
                        int32_t *end_ref, int32_t *end_read, int32_t *beg_ref,
                        int32_t *beg_read) {
  const int KB = 18;
  int64_t *H = (int64_t *)calloc((size_t)readLen + 1, sizeof(int64_t));
  int64_t *E = (int64_t *)calloc((size_t)readLen + 1, sizeof(int64_t));
  const int64_t NEG = -((int64_t)(gapO + gapE + 1) << KB);
  for (int32_t i = 0; i < readLen; i++) { H[i] = ((int64_t)0 << 9) | (i + 1); E[i] = NEG; } /* virtual cell (i, -1): diagonal successor (i + 1, 0) */
  int64_t bestV = 0; int32_t best = 0, bc = 0, br = readLen - 1;
  for (int32_t c = 0; c < refLen; c++) {
    int64_t F = NEG;
    int64_t diag = ((int64_t)c << 9) | 0;          /* virtual row -1: successor (0, c) */
    const int8_t *mrow = mat + ref[c] * n;
    for (int32_t i = 0; i < readLen; i++) {
      int64_t h = diag + ((int64_t)mrow[read[i]] << KB);
      if (E[i] > h) h = E[i];
      if (F > h) h = F;
      const int64_t Z = ((int64_t)(c + 1) << 9) | (int64_t)(i + 1);
      if (Z > h) h = Z;
      diag = H[i];
      H[i] = h;
      const int32_t sc = (int32_t)(h >> KB);
      if (sc > best) { best = sc; bestV = h; bc = c; br = i; }
      const int64_t hg = h - ((int64_t)gapO << KB);
      int64_t e = E[i] - ((int64_t)gapE << KB);
      E[i] = e > hg ? e : hg;
      int64_t f = F - ((int64_t)gapE << KB);
      F = f > hg ? f : hg;
    }
  }
  free(H); free(E);
  *score = best; *end_ref = bc; *end_read = br;
  const int32_t key = (int32_t)(bestV & ((1 << KB) - 1));
  *beg_ref = key >> 9; *beg_read = key & 511;
}

/* ------------------------------------------------------------------------
 * a-13: banded_sw, src/ssw.c:594-792, restated with the same three row
 * arrays and index helpers (set_u :56-62, set_d :66-72) so that the edge
 * sentinels (:655) behave identically.  Returns the cigar length, writes the
 * ops (already reversed into alignment order) to cigar_out.  INT32_MAX means
 * the reference's direction buffer size check overflowed (:631-642).
 * *status = 1 when the reference would print "Trace back error".
 * ---------------------------------------------------------------------- */
static inline int32_t band_u(int32_t w, int32_t i, int32_t j) {
  int32_t x = i - w; x = x > 0 ? x : 0; return j - x + 1;
}
static inline int32_t band_d(int32_t w, int32_t i, int32_t j, int32_t p) {
  int32_t x = i - w; x = x > 0 ? x : 0; x = j - x; return x * 3 + p;
}

int32_t orc_banded_sw(const int8_t *ref, const int8_t *read, int32_t refLen,
                      int32_t readLen, int32_t score, uint32_t gapO,
                      uint32_t gapE, int32_t band_width, const int8_t *mat,
                      int32_t n, uint32_t *cigar_out, int32_t cigar_cap,
                      int32_t *status) {
  int32_t max = 0, width, width_d, *h_b = NULL, *e_b = NULL, *h_c = NULL;
  int8_t *direction = NULL;
  int32_t wgO = (int32_t)gapO, wgE = (int32_t)gapE;
  if (status) *status = 0;
  do {
    width = band_width * 2 + 3; width_d = band_width * 2 + 1;
    /* reference grows s2 by powers of two and bails out when it goes
     * negative, :631-642: that happens once width_d*readLen*3 >= 2^30 */
    if ((int64_t)width_d * readLen * 3 >= ((int64_t)1 << 30)) {
      free(h_b); free(e_b); free(h_c); free(direction);
      return INT32_MAX;
    }
    h_b = (int32_t *)realloc(h_b, (size_t)(width + 1) * sizeof(int32_t));
    e_b = (int32_t *)realloc(e_b, (size_t)(width + 1) * sizeof(int32_t));
    h_c = (int32_t *)realloc(h_c, (size_t)(width + 1) * sizeof(int32_t));
    direction = (int8_t *)realloc(direction, (size_t)width_d * readLen * 3 + 3);
    memset(direction, 0, (size_t)width_d * readLen * 3 + 3);
    memset(e_b, 0, (size_t)(width + 1) * sizeof(int32_t));
    memset(h_c, 0, (size_t)(width + 1) * sizeof(int32_t));
    for (int32_t j = 0; j <= width; j++) h_b[j] = 0;      /* :645 (1..width-2) */
    for (int32_t i = 0; i < readLen; i++) {
      int32_t beg = 0, end = refLen - 1, u = 0, edge, f;
      int32_t j = i - band_width;
      beg = beg > j ? beg : j;
      j = i + band_width;
      end = end < j ? end : j;
      edge = end + 1 < width - 1 ? end + 1 : width - 1;   /* :654 */
      f = h_b[0] = e_b[0] = h_b[edge] = e_b[edge] = h_c[0] = 0; /* :655 */
      int8_t *dl = direction + (size_t)width_d * i * 3;
      for (j = beg; j <= end; j++) {
        int32_t e = band_u(band_width, i - 1, j);
        int32_t b = band_u(band_width, i, j - 1);
        int32_t d = band_u(band_width, i - 1, j - 1);
        u = band_u(band_width, i, j);
        int32_t de = band_d(band_width, i, j, 0);
        int32_t df = band_d(band_width, i, j, 1);
        int32_t dh = band_d(band_width, i, j, 2);
        int32_t t1 = i == 0 ? -wgO : h_b[e] - wgO;        /* :668-671 */
        int32_t t2 = i == 0 ? -wgE : e_b[e] - wgE;
        e_b[u] = t1 > t2 ? t1 : t2;
        dl[de] = t1 > t2 ? 3 : 2;
        t1 = h_c[b] - wgO;                                /* :673-676 */
        t2 = f - wgE;
        f = t1 > t2 ? t1 : t2;
        dl[df] = t1 > t2 ? 5 : 4;
        int32_t e1 = e_b[u] > 0 ? e_b[u] : 0;             /* :678-682 */
        int32_t f1 = f > 0 ? f : 0;
        t1 = e1 > f1 ? e1 : f1;
        t2 = h_b[d] + mat[ref[j] * n + read[i]];
        h_c[u] = t1 > t2 ? t1 : t2;
        if (h_c[u] > max) max = h_c[u];                   /* :684 */
        if (t1 <= t2) dl[dh] = 1;                         /* :686-690 */
        else dl[dh] = e1 > f1 ? dl[de] : dl[df];
      }
      for (j = 1; j <= u; j++) h_b[j] = h_c[j];           /* :692 */
    }
    band_width *= 2;
  } while (max < score);                                  /* :694-695 */
  band_width /= 2;
  width_d = band_width * 2 + 1;

  /* traceback, :698-771; ops collected in reverse then flipped (:773-784) */
  int32_t i = readLen - 1, j = refLen - 1, cnt = 0, l = 0, op = 0, cur = 0, plane = 2;
  uint32_t *c = (uint32_t *)malloc(((size_t)readLen + refLen + 4) * sizeof(uint32_t));
  int bad = 0;
  while (i > 0) {
    int32_t x = i - band_width; x = x > 0 ? x : 0;
    int32_t col = j - x;
    int8_t dir = 0;
    if (col >= 0 && col < width_d && j >= 0) dir = direction[(size_t)width_d * i * 3 + col * 3 + plane];
    switch (dir) {
      case 1: --i; --j; plane = 2; op = 0; break;
      case 2: --i; plane = 0; op = 1; break;
      case 3: --i; plane = 2; op = 1; break;
      case 4: --j; plane = 1; op = 2; break;
      case 5: --j; plane = 2; op = 2; break;
      default: bad = 1; break;
    }
    if (bad) break;
    if (op == cur) ++cnt;
    else {
      ++l;
      c[l - 1] = (uint32_t)cnt << 4 | (uint32_t)cur;
      cur = op;
      cnt = 1;
    }
  }
  int32_t len = 0;
  if (bad) {
    if (status) *status = 1;
  } else {
    if (op == 0) { ++l; c[l - 1] = (uint32_t)(cnt + 1) << 4; }   /* :754-761 */
    else { l += 2; c[l - 2] = (uint32_t)cnt << 4 | (uint32_t)op; c[l - 1] = 16; }
    len = l;
    for (int32_t s = 0; s < l && s < cigar_cap; s++) cigar_out[s] = c[l - 1 - s];
    if (l > cigar_cap && status) *status = 2;
  }
  free(c); free(direction); free(h_c); free(e_b); free(h_b);
  return len;
}

/* ------------------------------------------------------------------------
 * a-12: ssw_align, src/ssw.c:841-951 (+ ssw_init :808-833 bias, seq_reverse
 * :794-806).  `plain` selects the plain-Gotoh passes instead of the striped
 * emulation; everything else is shared.
 * ---------------------------------------------------------------------- */
static void ssw_align_impl(const int8_t *read, int32_t readLen,
                           const int8_t *ref, int32_t refLen,
                           const int8_t *mat, uint8_t gapO, uint8_t gapE,
                           uint8_t flag, uint16_t filters, int32_t filterd,
                           uint32_t *cigar_out, int32_t cigar_cap,
                           orc_ssw_result *r, int plain) {
  const int32_t n = 5;
  memset(r, 0, sizeof *r);
  r->ref_begin1 = -1;
  r->read_begin1 = -1;
  int32_t bias = 0;
  for (int i = 0; i < n * n; i++) if (mat[i] < bias) bias = mat[i]; /* :819-822 */
  bias = abs(bias);
  int word = 0;
  sw_end best;
  if (plain == 2 && (flag & 0x08) && readLen < 512 && refLen < 512) {
    int32_t sc, er, eq, brf, brd;
    origin_pass(ref, refLen, read, readLen, mat, n, gapO, gapE, &sc, &er, &eq, &brf, &brd);
    r->score1 = (uint16_t)sc; r->ref_end1 = er; r->read_end1 = eq;
    r->ref_begin1 = brf; r->read_begin1 = brd;
    if (sc == 0) { r->ref_begin1 = -1; r->read_begin1 = -1; return; }
    goto cigar_stage;
  }
  if (plain) {
    best = plain_pass(ref, 0, refLen, read, readLen, mat, n, gapO, gapE, -1);
    word = 1;
  } else {
    best = striped_byte(ref, 0, refLen, read, readLen, mat, n, gapO, gapE,
                        (uint8_t)-1, (uint8_t)bias);       /* :870-872 */
    if (best.score == 255) {                               /* :873-877 */
      best = striped_word(ref, 0, refLen, read, readLen, mat, n, gapO, gapE,
                          (uint16_t)-1);
      word = 1;
    }
  }
  r->score1 = best.score;
  r->ref_end1 = best.ref;
  r->read_end1 = best.read;
  if (flag == 0 || (flag == 2 && r->score1 < filters)) return;  /* :904 */

  int32_t rl = r->read_end1 + 1;                           /* :906-923 */
  int8_t *rev = (int8_t *)calloc((size_t)(rl > 0 ? rl : 1), 1);
  for (int32_t s = 0; s < rl; s++) rev[s] = read[r->read_end1 - s];
  sw_end br;
  if (plain)
    br = plain_pass(ref, 1, r->ref_end1 + 1, rev, rl, mat, n, gapO, gapE, r->score1);
  else if (word == 0)
    br = striped_byte(ref, 1, r->ref_end1 + 1, rev, rl, mat, n, gapO, gapE,
                      (uint8_t)r->score1, (uint8_t)bias);
  else
    br = striped_word(ref, 1, r->ref_end1 + 1, rev, rl, mat, n, gapO, gapE,
                      r->score1);
  free(rev);
  r->ref_begin1 = br.ref;
  r->read_begin1 = r->read_end1 - br.read;
cigar_stage:
  if ((7 & flag) == 0 || ((2 & flag) != 0 && r->score1 < filters) ||
      ((4 & flag) != 0 && (r->ref_end1 - r->ref_begin1 > filterd ||
                           r->read_end1 - r->read_begin1 > filterd)))
    return;                                                /* :924-927 */
  int32_t rfl = r->ref_end1 - r->ref_begin1 + 1;           /* :930-935 */
  int32_t rdl = r->read_end1 - r->read_begin1 + 1;
  int32_t bw = abs(rfl - rdl) + 1;
  int32_t st = 0;
  int32_t len = orc_banded_sw(ref + r->ref_begin1, read + r->read_begin1, rfl,
                              rdl, r->score1, gapO, gapE, bw, mat, n,
                              cigar_out, cigar_cap, &st);
  if (len == INT32_MAX) { r->cigar_len = 0; r->score1 = 0; }   /* :941-944 */
  else r->cigar_len = len;
  r->status = st;
}

void orc_ssw_align(const int8_t *read, int32_t read_len, const int8_t *ref,
                   int32_t ref_len, const int8_t mat[25], uint8_t gap_open,
                   uint8_t gap_extend, uint8_t flag, uint16_t filters,
                   int32_t filterd, uint32_t *cigar_out, int32_t cigar_cap,
                   orc_ssw_result *res) {
  ssw_align_impl(read, read_len, ref, ref_len, mat, gap_open, gap_extend, flag,
                 filters, filterd, cigar_out, cigar_cap, res, 0);
}
void orc_ssw_align_mode(const int8_t *read, int32_t read_len, const int8_t *ref,
                        int32_t ref_len, const int8_t mat[25], uint8_t gap_open,
                        uint8_t gap_extend, uint8_t flag, uint16_t filters,
                        int32_t filterd, uint32_t *cigar_out, int32_t cigar_cap,
                        orc_ssw_result *res, int mode) {
  ssw_align_impl(read, read_len, ref, ref_len, mat, gap_open, gap_extend, flag,
                 filters, filterd, cigar_out, cigar_cap, res, mode);
}
void orc_ssw_align_plain(const int8_t *read, int32_t read_len,
                         const int8_t *ref, int32_t ref_len,
                         const int8_t mat[25], uint8_t gap_open,
                         uint8_t gap_extend, uint8_t flag, uint16_t filters,
                         int32_t filterd, uint32_t *cigar_out,
                         int32_t cigar_cap, orc_ssw_result *res) {
  ssw_align_impl(read, read_len, ref, ref_len, mat, gap_open, gap_extend, flag,
                 filters, filterd, cigar_out, cigar_cap, res, 1);
}

/* ---- optional: the REAL reference ssw core (oracle/_ref/libssw_ref.so) ---- */
typedef struct {  /* s_align, src/ssw.h:47-57 */
  uint16_t score1, score2;
  int32_t ref_begin1, ref_end1, read_begin1, read_end1, ref_end2;
  uint32_t *cigar;
  int32_t cigarLen;
} ref_s_align;
typedef void *(*fn_ssw_init)(const int8_t *, int32_t, const int8_t *, int32_t, int8_t);
typedef ref_s_align *(*fn_ssw_align)(const void *, const int8_t *, int32_t, uint8_t, uint8_t, uint8_t, uint16_t, int32_t, int32_t);
typedef void (*fn_init_destroy)(void *);
typedef void (*fn_align_destroy)(ref_s_align *);
static fn_ssw_init g_ssw_init;
static fn_ssw_align g_ssw_align;
static fn_init_destroy g_init_destroy;
static fn_align_destroy g_align_destroy;

int orc_use_reference_ssw(const char *libpath) {
  if (!libpath) { g_ssw_init = NULL; return 0; }
  void *h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
  if (!h) return -1;
  g_ssw_align = (fn_ssw_align)dlsym(h, "ssw_align");
  g_init_destroy = (fn_init_destroy)dlsym(h, "init_destroy");
  g_align_destroy = (fn_align_destroy)dlsym(h, "align_destroy");
  fn_ssw_init f = (fn_ssw_init)dlsym(h, "ssw_init");
  if (!f || !g_ssw_align || !g_init_destroy || !g_align_destroy) return -2;
  g_ssw_init = f;
  return 0;
}

/* ------------------------------------------------------------------------
 * Aligner::Align(query, ref, ref_len, filter, alignment),
 * src/ssw_cpp.cpp:234-283 with SetFlag :90-93 and the Filter the SW driver
 * builds (src/SmithWaterman.h:193-197, src/ssw_cpp.h:107-111).
 * ---------------------------------------------------------------------- */
void orc_align(const char *query, int32_t query_len, const char *ref,
               int32_t ref_len, const orc_params *p, uint32_t *cigar_out,
               int32_t cigar_cap, orc_ssw_result *res, int plain) {
  int8_t mat[25];
  orc_build_matrix(p->match, p->mismatch, mat);
  int8_t *tq = (int8_t *)malloc((size_t)(query_len > 0 ? query_len : 1));
  int8_t *tr = (int8_t *)malloc((size_t)(ref_len > 0 ? ref_len : 1));
  orc_translate(query, query_len, tq);
  orc_translate(ref, ref_len, tr);
  uint8_t flag = 0x08;                       /* report_begin_position */
  if (p->report_cigar) flag |= 0x0f;
  uint16_t filters = (uint16_t)p->score_threshold;
  if (g_ssw_init && !plain) {
    void *prof = g_ssw_init(tq, query_len, mat, 5, 2);
    ref_s_align *a = g_ssw_align(prof, tr, ref_len, (uint8_t)p->gap_open,
                                 (uint8_t)p->gap_extend, flag, filters, 32767,
                                 query_len);
    memset(res, 0, sizeof *res);
    res->score1 = a->score1;
    res->ref_begin1 = a->ref_begin1; res->ref_end1 = a->ref_end1;
    res->read_begin1 = a->read_begin1; res->read_end1 = a->read_end1;
    res->cigar_len = a->cigar ? a->cigarLen : 0;
    for (int32_t i = 0; i < res->cigar_len && i < cigar_cap; i++) cigar_out[i] = a->cigar[i];
    g_align_destroy(a);
    g_init_destroy(prof);
  } else {
    ssw_align_impl(tq, query_len, tr, ref_len, mat, (uint8_t)p->gap_open,
                   (uint8_t)p->gap_extend, flag, filters, 32767, cigar_out,
                   cigar_cap, res, plain);
  }
  free(tq);
  free(tr);
}

/* inPlaceReverseComplement, src/sequenceTools.h:98-116: only upper-case
 * A/C/G/T are complemented, everything else is left as is. */
static void revcomp_window(char *s, uint64_t n) {
  for (uint64_t i = 0; i < n / 2; i++) { char t = s[i]; s[i] = s[n - 1 - i]; s[n - 1 - i] = t; }
  for (uint64_t i = 0; i < n; i++)
    switch (s[i]) {
      case 'A': s[i] = 'T'; break;
      case 'C': s[i] = 'G'; break;
      case 'T': s[i] = 'A'; break;
      case 'G': s[i] = 'C'; break;
      default: break;
    }
}

/* ------------------------------------------------------------------------
 * a-8: performSmithWatermanOnRange2 body for one overlap,
 * src/SmithWaterman.h:200-230.
 * ---------------------------------------------------------------------- */
void orc_sw_on_overlap(const orc_overlap *ov, const char *read,
                       uint64_t read_len, const char *entry,
                       uint64_t entry_len, const orc_params *p,
                       orc_alignment *out, uint32_t *cigar_out,
                       int32_t cigar_cap, int plain) {
  int64_t s = ov->rel > 0 ? ov->rel : 0;                  /* :204 */
  uint64_t wlen = 0;
  if ((uint64_t)s <= entry_len) {                         /* substr :205-206 */
    wlen = entry_len - (uint64_t)s;
    if (wlen > read_len) wlen = read_len;
  }
  char *win = (char *)malloc(wlen + 1);
  memcpy(win, entry + s, wlen);
  win[wlen] = 0;
  if (ov->revcomp) revcomp_window(win, wlen);             /* :207 */
  orc_ssw_result r;
  orc_align(read, (int32_t)read_len, win,
            (int32_t)(read_len < wlen ? read_len : wlen), p, cigar_out,
            cigar_cap, &r, plain);                        /* :208-210 */
  memset(out, 0, sizeof *out);
  out->read = ov->read; out->entry = ov->entry; out->rel = ov->rel;
  out->revcomp = ov->revcomp;
  out->score = r.score1;
  out->ref_begin = r.ref_begin1; out->ref_end = r.ref_end1;
  out->query_begin = r.read_begin1; out->query_end = r.read_end1;
  out->cigar_len = (uint32_t)r.cigar_len;
  if (ov->revcomp) {                                      /* :211-226 */
    if (p->report_cigar && r.cigar_len > 0)
      for (int32_t a = 0, b = r.cigar_len - 1; a < b; a++, b--) {
        uint32_t t = cigar_out[a]; cigar_out[a] = cigar_out[b]; cigar_out[b] = t;
      }
    int32_t tmp = out->ref_begin;
    out->ref_begin = (int32_t)(wlen - (uint64_t)(out->ref_end + 1));
    out->ref_end = (int32_t)(wlen - (uint64_t)(tmp + 1));
    tmp = out->query_begin;
    out->query_begin = (int32_t)(read_len - (uint64_t)(out->query_end + 1));
    out->query_end = (int32_t)(read_len - (uint64_t)(tmp + 1));
  }
  out->ref_begin += (int32_t)s;                           /* :227-228 */
  out->ref_end += (int32_t)s;
  free(win);
}

/* ------------------------------------------------------------------------
 * alignToDatabase, src/SLAM.h:59-79.
 * phase_seconds: [0] read k-mers, [1] genome k-mers, [2] sort, [3] join +
 * overlap sort/unique, [4] Smith-Waterman, [5] total.
 * ---------------------------------------------------------------------- */
int orc_align_to_database(uint64_t n_reads, const char *const *reads,
                          const uint64_t *read_lens, uint64_t n_entries,
                          const char *const *entries,
                          const uint64_t *entry_lens, const orc_params *p,
                          int plain, orc_alignment **out, uint64_t *n_out,
                          uint32_t **cigar_pool, uint64_t *n_cigar,
                          double phase_seconds[6]) {
  double t0 = now_s(), t;
  double ph[6] = {0, 0, 0, 0, 0, 0};
  uint64_t nr = 0, ng = 0;
  for (uint64_t i = 0; i < n_reads; i++) nr += orc_count_kmers(read_lens[i], 1);
  for (uint64_t i = 0; i < n_entries; i++) ng += orc_count_kmers(entry_lens[i], ORC_K / 2);
  orc_kmer_rec *recs = (orc_kmer_rec *)malloc((nr + ng + 1) * sizeof(orc_kmer_rec));
  if (!recs) return -1;
  t = now_s();
  orc_extract_all(n_reads, reads, read_lens, 0, 1, recs);               /* :63 */
  ph[0] = now_s() - t; t = now_s();
  orc_extract_all(n_entries, entries, entry_lens, 1, ORC_K / 2, recs + nr); /* :64 */
  ph[1] = now_s() - t; t = now_s();
  orc_sort_kmers(recs, nr + ng);                                        /* :65 */
  ph[2] = now_s() - t; t = now_s();
  uint64_t raw = orc_count_overlaps(recs, nr + ng);
  orc_overlap *ov = (orc_overlap *)malloc((raw + 1) * sizeof(orc_overlap));
  if (!ov) { free(recs); return -1; }
  uint64_t m = orc_find_overlaps(recs, nr + ng, read_lens, ov, NULL);   /* :66-67 */
  free(recs);                                                           /* :74-75 */
  ph[3] = now_s() - t; t = now_s();

  orc_alignment *al = (orc_alignment *)calloc(m + 1, sizeof(orc_alignment));
  uint64_t *coff = (uint64_t *)malloc((m + 1) * sizeof(uint64_t));
  uint64_t ctot = 0;
  for (uint64_t i = 0; i < m; i++) {
    coff[i] = ctot;
    ctot += p->report_cigar ? 2 * read_lens[ov[i].read] + 8 : 1;
  }
  coff[m] = ctot;
  uint32_t *ctmp = (uint32_t *)malloc((ctot + 1) * sizeof(uint32_t));
  if (!al || !coff || !ctmp) return -1;
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t i = 0; i < (int64_t)m; i++)                              /* :76-77 */
    orc_sw_on_overlap(&ov[i], reads[ov[i].read], read_lens[ov[i].read],
                      entries[ov[i].entry], entry_lens[ov[i].entry], p, &al[i],
                      ctmp + coff[i], (int32_t)(coff[i + 1] - coff[i]), plain);
  uint64_t cn = 0;
  for (uint64_t i = 0; i < m; i++) {
    al[i].cigar_off = cn;
    memmove(ctmp + cn, ctmp + coff[i], al[i].cigar_len * sizeof(uint32_t));
    cn += al[i].cigar_len;
  }
  ph[4] = now_s() - t;
  ph[5] = now_s() - t0;
  free(ov);
  free(coff);
  *out = al; *n_out = m; *cigar_pool = ctmp; *n_cigar = cn;
  if (phase_seconds) memcpy(phase_seconds, ph, sizeof ph);
  return 0;
}
