// banded_core.h -- banded_sw (reference src/ssw.c:594-792) restated ONCE, against a small
// memory-accessor interface, so that the in-kernel fast path (sw.hip, bands 1-2) and the
// stand-alone kernel (cigar.hip, any band) run literally the same code.
//
// Kept from the reference: the three row arrays h_b / e_b / h_c with set_u indexing
// (ssw.c:56-62), the sentinel assignment h_b[edge] = e_b[edge] = 0 (ssw.c:655) that can clobber a
// live cell when the band is clipped by the reference end, the tie rules `t1 > t2 ? open :
// extend` (ssw.c:669-675) and `t1 <= t2 -> diagonal` (ssw.c:685-689), `max` carried across
// attempts (ssw.c:684), the `while (i > 0)` traceback with its final-element fix-up
// (ssw.c:698-771).  The serial chain (f, left H) and the diagonal H run in registers; the
// upper-row values and the next reference code are fetched one cell ahead.
//
// Accessor M: int32_t& hb(k), eb(k), hc(k); uint32_t q(i), r(j) (SSW codes 0..4 of the aligned
// read / reference spans); void set_dir(i, col, v); uint32_t get_dir(i, col).
#pragma once
#include <type_traits>
#include <utility>
#include "common.h"

namespace kslam {

// one attempt with `band_width`; returns the (carried) maximum, ssw.c:645-693
template <class M>
__device__ inline int32_t banded_attempt(M &m, int32_t refLen, int32_t readLen, int32_t band_width,
                                         const SwParams &p, int32_t mx) {
  const int32_t width = band_width * 2 + 3;
  for (int32_t k = 0; k <= width; k++) { m.hb(k) = 0; m.eb(k) = 0; m.hc(k) = 0; }
  for (int32_t i = 0; i < readLen; i++) {
    int32_t beg = 0, end = refLen - 1, u = 0, edge;
    int32_t j = i - band_width;
    beg = beg > j ? beg : j;
    j = i + band_width;
    end = end < j ? end : j;
    edge = end + 1 < width - 1 ? end + 1 : width - 1;               // ssw.c:654
    m.hb(0) = 0; m.eb(0) = 0; m.hb(edge) = 0; m.eb(edge) = 0; m.hc(0) = 0;  // ssw.c:655
    const uint32_t qc = m.q(i);
    const int32_t xi = i - band_width > 0 ? i - band_width : 0;
    const int32_t xim = i - 1 - band_width > 0 ? i - 1 - band_width : 0;
    int32_t f = 0;
    int32_t e = beg - xim + 1;            // set_u(e, w, i-1, j)
    int32_t hb_d = m.hb(e - 1);           // set_u(d, w, i-1, j-1) = e - 1
    int32_t hleft = m.hc(beg - xi);       // set_u(b, w, i, j-1) = u - 1; first one is h_c[0] = 0
    int32_t hb_e = m.hb(e), eb_e = m.eb(e);
    uint32_t rcode = m.r(beg);
    for (j = beg; j <= end; j++, e++) {
      int32_t n_hb = 0, n_eb = 0;
      uint32_t n_rc = 0;
      if (j < end) { n_hb = m.hb(e + 1); n_eb = m.eb(e + 1); n_rc = m.r(j + 1); }
      u = j - xi + 1;                     // set_u(u, w, i, j)
      const int32_t sc = (qc > 3u || rcode > 3u) ? 0 : (qc == rcode ? p.match : -p.mismatch);
      int32_t t1 = i == 0 ? -p.gap_open : hb_e - p.gap_open;        // ssw.c:668-671
      int32_t t2 = i == 0 ? -p.gap_extend : eb_e - p.gap_extend;
      const int32_t ev = t1 > t2 ? t1 : t2;
      m.eb(u) = ev;
      const uint32_t de = t1 > t2 ? 3u : 2u;
      t1 = hleft - p.gap_open;                                       // ssw.c:673-676
      t2 = f - p.gap_extend;
      f = t1 > t2 ? t1 : t2;
      const uint32_t df = t1 > t2 ? 5u : 4u;
      const int32_t e1 = ev > 0 ? ev : 0, f1 = f > 0 ? f : 0;      // ssw.c:678-682
      t1 = e1 > f1 ? e1 : f1;
      t2 = hb_d + sc;
      const int32_t hv = t1 > t2 ? t1 : t2;
      m.hc(u) = hv;
      if (hv > mx) mx = hv;                                        // ssw.c:684
      const uint32_t dh = t1 <= t2 ? 1u : (e1 > f1 ? de : df);     // ssw.c:686-690
      m.set_dir(i, j - xi, (de - 2u) | ((df - 4u) << 1) | (dh << 2));
      hleft = hv;
      hb_d = hb_e;
      hb_e = n_hb; eb_e = n_eb; rcode = n_rc;
    }
    for (j = 1; j <= u; j++) m.hb(j) = m.hc(j);                    // ssw.c:692
  }
  return mx;
}

// An accessor may offer `int32_t diag_run(int32_t i, int32_t j)`: how many of the cells (i, j), (i - 1, j - 1), ... say
// "diagonal" in the H plane, as far as the direction word that holds (i, j) goes (0: look at the cell the slow way).  The
// kernels that pack a band diagonal's cells into one word answer that with one load and a bit scan, and the walk then
// takes a whole run of matches -- most of an alignment -- per dependent load instead of one step.
template <class M, class = void>
struct has_diag_run : std::false_type {};
template <class M>
struct has_diag_run<M, std::void_t<decltype(std::declval<M &>().diag_run(0, 0))>> : std::true_type {};
// the bit scan for six 5-bit cells per word, cell r = bits 5 r .. 5 r + 4, H direction in bits 2..4 of a cell (1 = diagonal):
// consecutive cells r, r - 1, ... whose H direction is "diagonal"
__device__ inline int32_t diag_cells_down_from(uint32_t word, uint32_t r) {
  constexpr uint32_t ONES = 1u | (1u << 5) | (1u << 10) | (1u << 15) | (1u << 20) | (1u << 25);
  uint32_t x = (word & (0x1Cu * ONES)) ^ (0x04u * ONES);   // zero fields where the cell says diagonal
  x &= (2u << (5u * r + 4u)) - 1u;                          // cells 0 .. r
  if (x == 0) return (int32_t)r + 1;
  const uint32_t p = 31u - (uint32_t)__builtin_clz(x);      // highest cell that says something else
  return (int32_t)r - (int32_t)((p * 13u) >> 6);            // p / 5 for p < 30
}

// traceback, ssw.c:698-771.  Ops are written in TRACEBACK order into tmp[0..cap); returns the op
// count (may exceed cap: *ovf), or -1 on the reference's "Trace back error" path.
template <class M>
__device__ inline int32_t banded_traceback(M &m, int32_t refLen, int32_t readLen, int32_t band_width,
                                           uint32_t *tmp, uint32_t cap, bool *ovf) {
  int32_t i = readLen - 1, j = refLen - 1, cnt = 0, l = 0, op = 0, cur = 0, plane = 2;
  *ovf = false;
  while (i > 0) {
    const int32_t xi = i - band_width > 0 ? i - band_width : 0;
    const int32_t col = j - xi;
    const int32_t jend = (refLen - 1) < (i + band_width) ? (refLen - 1) : (i + band_width);
    uint32_t dir = 0;
    if constexpr (has_diag_run<M>::value) {
      if (plane == 2 && col >= 0 && j <= jend && j >= 0) {
        // a run of direction 1 (--i --j, M, stay in H; ssw.c:704-708).  Along a band diagonal a cell that follows an
        // existing cell exists as long as its column does (col and jend move with the row), and the loop ends at i == 0
        const int32_t run = min(m.diag_run(i, j), min(i, j + 1));
        if (run > 0) {
          if (cur == 0) cnt += run;
          else {
            if ((uint32_t)l < cap) tmp[l] = (uint32_t)cnt << 4 | (uint32_t)cur; else *ovf = true;
            ++l;
            cur = 0;
            cnt = run;
          }
          op = 0;
          i -= run;
          j -= run;
          continue;
        }
      }
    }
    if (col >= 0 && j <= jend && j >= 0) {
      const uint32_t bb = m.get_dir(i, col);
      // plane 2 (H): bits 2..4 as they are; plane 0 (E): 2 + bit 0; plane 1 (F): 4 + bit 1 -- the base and
      // the field mask of each plane sit in two small tables, so that lanes in different planes run
      // the same instructions
      dir = ((0x042u >> (4 * plane)) & 15u) + ((bb >> plane) & ((0x711u >> (4 * plane)) & 15u));
    }
    if (dir - 1u > 4u) return -1;   // the reference's "Trace back error" (ssw.c:747-750)
    // what a direction does (ssw.c:704-745), one byte per direction 1..5:
    // bit 0 --i, bit 1 --j, bits 2-3 the cigar op (M 0, I 1, D 2), bits 4-5 the next plane
    //   1: --i --j, M, H    2: --i, I, E    3: --i, I, H    4: --j, D, F    5: --j, D, H
    const uint32_t step = (uint32_t)(0x2A1A250523ull >> (8u * (dir - 1u)));
    i -= (int32_t)(step & 1u);
    j -= (int32_t)((step >> 1) & 1u);
    op = (int32_t)((step >> 2) & 3u);
    plane = (int32_t)((step >> 4) & 3u);
    if (op == cur) ++cnt;
    else {
      if ((uint32_t)l < cap) tmp[l] = (uint32_t)cnt << 4 | (uint32_t)cur; else *ovf = true;
      ++l;
      cur = op;
      cnt = 1;
    }
  }
  if (op == 0) {                                                    // ssw.c:754-761
    if ((uint32_t)l < cap) tmp[l] = (uint32_t)(cnt + 1) << 4; else *ovf = true;
    ++l;
  } else {
    if ((uint32_t)l + 1 < cap) { tmp[l] = (uint32_t)cnt << 4 | (uint32_t)op; tmp[l + 1] = 16u; } else *ovf = true;
    l += 2;
  }
  return l;
}

}  // namespace kslam
