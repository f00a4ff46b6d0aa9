// stream.cpp -- include/kslam_stream.h: the reference's batch loop (metagenomicAnalysis_Low_Mem, src/SLAM.h:193-250)
// over this library's own C ABI.  Plain C++: every stage it calls is an exported entry point, so a k-SLAM host
// could write the same loop itself (INTEGRATION.md shows it); this file is that loop, tested and timed.
//
//   main thread:  cut batch k+d (kslam_fastq_batch_end) -> kslam_submit_batch_fastq_text ... kslam_collect_batch(k)
//   worker:       host stage of batch k-1: the SAM records and the per-read lines arrive WRITTEN (include/kslam_samtext.h: on the
//                 GPU, inside the lane) and go to kslam_sam_writer (its own thread) / per_read_fd as they are; what is left for
//                 the CPUs is kslam_taxreport_add_batch.  A batch whose text the device left to the host (pseudo-assembly
//                 fallback), or KSLAM_HOST_SAM_TEXT=1: kslam_tail_finish_prepare -> kslam_tail_finish_write_rows,
//                 kslam_tail_classify
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/kslam_fastq.h"
#include "../../include/kslam_stream.h"
#include "../../include/kslam_samtext.h"
#include "workers.hpp"

namespace {
using namespace kslam_host;

struct Window { uint64_t p1, e1, p2, e2; };

bool write_all(int fd, const char *p, uint64_t n) {
  while (n) {
    const ssize_t w = ::write(fd, p, (size_t)std::min<uint64_t>(n, 1ull << 30));
    if (w < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += w;
    n -= (uint64_t)w;
  }
  return true;
}

}  // namespace

extern "C" kslam_status kslam_stream_classify(kslam_ctx *ctx, const kslam_index_view *index, const kslam_taxdb *taxdb,
                                              kslam_taxreport *report, const char *r1, uint64_t len1, const char *r2,
                                              uint64_t len2, const kslam_stream_params *P, uint32_t **tax_ids_out,
                                              uint64_t *n_tax_ids, kslam_stream_stats *stats) {
  if (tax_ids_out) *tax_ids_out = nullptr;
  if (n_tax_ids) *n_tax_ids = 0;
  kslam_stream_stats st;
  memset(&st, 0, sizeof st);
  st.first_max_insert_size = 0xFFFFFFFFu;
  const double t_begin = now_ms();
  kslam_sam_writer *writer = nullptr;
  std::thread worker;
  kslam_status worker_status = KSLAM_OK;
  std::string worker_error;
  std::deque<uint64_t> tickets;
  std::vector<uint32_t> all_ids;
  bool pairing_set = false, text_set = false;

  const int pool_cap = P && P->pool_threads ? (int)P->pool_threads : std::max(2, usable_cpus() - 4);
  Pool::get().add_cap(pool_cap);
  // whatever happens: the worker joined, the tickets collected and released, the writer closed, the pairing switched off
  auto wind_down = [&]() -> kslam_status {
    if (worker.joinable()) worker.join();
    for (uint64_t tk : tickets) {
      kslam_batch_result r;
      if (kslam_collect_batch(ctx, tk, &r) == KSLAM_OK) kslam_release_batch(ctx, &r);
    }
    tickets.clear();
    kslam_status w = KSLAM_OK;
    if (writer) {
      uint64_t bytes = 0;
      double sec = 0;
      w = kslam_sam_writer_close(writer, &bytes, &sec);
      st.seconds_in_write = sec;
      writer = nullptr;
    }
    if (text_set) kslam_set_sam_text(ctx, 0, 0, 10, 0);
    if (pairing_set) kslam_set_pairing(ctx, 1, 0, 0.95, 0);
    Pool::get().remove_cap(pool_cap);
    return w;
  };

  const kslam_status status = guarded([&] {
    if (!ctx || !index || !P || !stats) fail(KSLAM_ERR_ARG, "null argument");
    if ((len1 && !r1) || (len2 && !r2)) fail(KSLAM_ERR_ARG, "null text");
    const bool paired = P->tail.paired != 0;
    if (!paired && (r2 || len2)) fail(KSLAM_ERR_ARG, "single-end data (tail.paired == 0) is ONE text: r2 must be NULL");
    if (P->pairs_per_batch == 0) fail(KSLAM_ERR_ARG, "pairs_per_batch must be positive");
    if (taxdb && !tax_ids_out) fail(KSLAM_ERR_ARG, "tax_ids output missing");
    uint32_t depth = P->depth ? P->depth : 3;
    if (const char *e = getenv("KSLAM_STREAM_DEPTH")) depth = (uint32_t)std::max(1, std::min(8, atoi(e)));   // A/B
    const uint32_t stages = KSLAM_TAIL_INSERT_SCREEN | KSLAM_TAIL_SCORE_SCREEN | (P->tail.pseudo_assembly ? KSLAM_TAIL_PSEUDO_ASM : 0u);
    if (kslam_set_pairing(ctx, paired ? 1 : 0, P->tail.score_threshold, P->tail.score_fraction, stages) != KSLAM_OK)
      fail(KSLAM_ERR_UNSUPPORTED, kslam_last_error(ctx));
    pairing_set = true;
    // the SAM records and the per-read lines written on the GPU (include/kslam_samtext.h); KSLAM_HOST_SAM_TEXT=1 keeps the
    // host formatter for everything (A/B, and the route of a batch the device hands back without text)
    const bool device_text = !(getenv("KSLAM_HOST_SAM_TEXT") && getenv("KSLAM_HOST_SAM_TEXT")[0] == '1') && (P->sam_fd >= 0 || taxdb);
    if (device_text) {
      if (kslam_set_sam_annotations(ctx, index, taxdb) != KSLAM_OK ||
          kslam_set_sam_text(ctx, P->sam_fd >= 0 ? 1 : 0, taxdb ? 1 : 0, P->tail.num_sam_alignments, P->tail.sam_xa) != KSLAM_OK)
        fail(KSLAM_ERR_STATE, kslam_last_error(ctx));
      text_set = true;
    }
    if (P->sam_fd >= 0) {
      if (kslam_sam_writer_open(P->sam_fd, &writer) != KSLAM_OK) fail(KSLAM_ERR_ARG, "could not start the SAM writer");
      if (P->sam_header && P->sam_header_len && kslam_write_queued(writer, P->sam_header, P->sam_header_len) != 0)
        fail(KSLAM_ERR_ARG, "writing the SAM header failed");
    }
    kslam_tail_params host_all = P->tail, host_write = P->tail, host_sorted = P->tail;   // what the host stage still has to run
    host_write.pseudo_assembly = 0;
    host_sorted.pseudo_assembly = 0;
    host_sorted.stages = (P->tail.stages ? P->tail.stages : KSLAM_TAIL_ALL) | KSLAM_TAIL_GROUPS_SORTED;

    // ---- batch boundaries, found ahead of the submission (src/SLAM.h:193, 201-206) ----
    uint64_t p1 = 0, p2 = 0, done_pairs = 0;
    uint32_t passes_left = P->passes > 1 ? P->passes - 1 : 0;
    bool exhausted = false;
    auto next_window = [&](Window *w) -> bool {
      if ((exhausted || !(p1 < len1 || p2 < len2)) && passes_left && len1 && (len2 || !paired)) {   // the texts once more
        passes_left--;
        p1 = p2 = 0;
        exhausted = false;
      }
      if (exhausted || !(p1 < len1 || p2 < len2)) return false;
      uint64_t want = P->pairs_per_batch;
      if (P->max_pairs_total) {
        if (done_pairs >= P->max_pairs_total) return false;
        want = std::min(want, P->max_pairs_total - done_pairs);   // readsPerGoTemp
      }
      uint64_t e1 = 0, e2 = 0;
      int c1 = 0, c2 = 0;
      const double tc = now_ms();
      if (kslam_fastq_batch_end(r1 + p1, len1 - p1, want, 1, P->tail.threads, &e1, &c1) != KSLAM_OK ||
          (paired && kslam_fastq_batch_end(r2 + p2, len2 - p2, want, 1, P->tail.threads, &e2, &c2) != KSLAM_OK))
        fail(KSLAM_ERR_ARG, kslam_tail_last_error());
      st.seconds_cutting += (now_ms() - tc) * 1e-3;
      *w = Window{p1, p1 + e1, p2, p2 + e2};
      done_pairs += want;
      p1 += e1;
      p2 += e2;
      if (p1 >= len1 || (paired && p2 >= len2)) exhausted = true;
      return true;
    };

    // The host stage of one batch, on the worker thread (it owns `res`), in the reference's order (src/SLAM.h:228-246):
    //   (1) everything that CHANGES res.read_pairs / res.pairs: what the device left of pseudo-assembly + the second score
    //       screen, then writeSAMOutputPairs' per-pair sort (only when there is a SAM file: without one the reference does
    //       not sort, and the classification sees the unsorted order) -- kslam_tail_finish_prepare;
    //   (2) the SAM text on this thread and the taxonomy part (per-read LCA, <out>_PerRead, the report's records) on a
    //       second one at the same time.  From here on both only READ `res`; the pool shares its workers between their
    //       loops, and the serial stretches of one (offsets, buffer growth, the per-read file's write) run under the
    //       other's loops instead of leaving the workers idle.
    auto host_stage = [&](kslam_batch_result res) {
      name_thread("kslam-host");
      kslam_reads_view reads = {res.n_reads, nullptr, res.reads_bases_off, nullptr, res.reads_bases_off, res.reads_ids, res.reads_ids_off};
      kslam_status tax_status = KSLAM_OK;
      std::string tax_error;
      std::thread tax_thread;
      auto tax_part = [&] {
          tax_status = guarded([&] {
            const double t1 = now_ms();
            const size_t base = all_ids.size();
            all_ids.resize(base + res.n_read_pairs);
            uint64_t tlen = 0;
            bool wrote = true;
            if (res.text_flags & KSLAM_TEXT_PER_READ) {   // taxonomy ids and lines came with the batch (GPU)
              if (res.n_read_pairs) memcpy(all_ids.data() + base, res.tax_ids, sizeof(uint32_t) * res.n_read_pairs);
              tlen = res.per_read_len;
              wrote = P->per_read_fd < 0 || write_all(P->per_read_fd, res.per_read_text, tlen);
            } else {
              char *text = nullptr;
              const kslam_status b = kslam_tail_classify(&host_write, &reads, index, taxdb, res.read_pairs, res.n_read_pairs, res.pairs,
                                                         res.n_pairs, all_ids.data() + base, &text, &tlen);
              if (b != KSLAM_OK) fail(b, kslam_tail_last_error());
              wrote = P->per_read_fd < 0 || write_all(P->per_read_fd, text, tlen);
              kslam_free(text);
            }
            if (!wrote) fail(KSLAM_ERR_ARG, std::string("writing the per-read file failed: ") + strerror(errno));
            st.per_read_bytes += tlen;
            const double t2 = now_ms();
            st.seconds_classify += (t2 - t1) * 1e-3;
            if (report) {
              const kslam_status c = kslam_taxreport_add_batch(report, &reads, index, res.read_pairs, res.n_read_pairs, res.pairs,
                                                               res.n_pairs, all_ids.data() + base);
              if (c != KSLAM_OK) fail(c, kslam_tail_last_error());
              st.seconds_report += (now_ms() - t2) * 1e-3;
            }
          });
          if (tax_status != KSLAM_OK) tax_error = g_err;
      };
      kslam_status s = guarded([&] {
        const bool on_gpu = (res.pair_stats.stages_done & KSLAM_TAIL_PSEUDO_ASM) != 0;
        if (P->tail.pseudo_assembly && !on_gpu) st.batches_pseudo_on_host++;
        const kslam_tail_params *tp = (on_gpu || !P->tail.pseudo_assembly) ? &host_write : &host_all;
        kslam_tail_stats ps;
        memset(&ps, 0, sizeof ps);
        const double t0 = now_ms();
        if (res.text_flags & KSLAM_TEXT_PAIRS_SORTED) {   // the device finished every stage and ran the per-pair sort: nothing to change
          ps.n_read_pairs = res.n_read_pairs;
          for (uint64_t g = 0; g < res.n_read_pairs; g++) ps.n_paired_final += res.read_pairs[g].count;
        } else {
          const kslam_status a = kslam_tail_finish_prepare(tp, &reads, res.overlaps, res.n_overlaps, res.read_pairs, res.n_read_pairs,
                                                           res.pairs, res.n_pairs, writer ? 1 : 0, &ps);
          if (a != KSLAM_OK) fail(a, kslam_tail_last_error());
        }
        st.seconds_sam_text += (now_ms() - t0) * 1e-3;
        st.n_alignment_pairs += ps.n_paired_final;
        st.n_read_pairs_aligned += ps.n_read_pairs;
        st.n_overlaps += res.n_overlaps;
        if (st.n_batches == 0) st.first_max_insert_size = res.pair_stats.max_insert_size;
        st.n_batches++;
        st.n_pairs += P->tail.paired ? res.n_reads / 2 : res.n_reads;
      });
      std::string err = s != KSLAM_OK ? g_err : std::string();
      const bool two_threads = P->host_threads != 1;
      if (s == KSLAM_OK && taxdb && two_threads) tax_thread = std::thread([&] { name_thread("kslam-tax"); tax_part(); });
      if (s == KSLAM_OK && writer && (res.text_flags & KSLAM_TEXT_SAM)) {
        s = guarded([&] {   // written on the GPU: the page-locked block joins the writer's queue as it is and goes back to the
                            // context's pool once it is in the file
          const double t0 = now_ms();
          char *block = res.sam_text;
          const uint64_t len = res.sam_text_len;
          res.sam_text = nullptr;   // the writer owns it now
          static const auto give_back = [](void *user, void *data) { kslam_free_pinned(static_cast<kslam_ctx *>(user), data); };
          if (kslam_sam_writer_enqueue(writer, block, len, +give_back, ctx) != KSLAM_OK) fail(KSLAM_ERR_ARG, kslam_tail_last_error());
          st.seconds_sam_text += (now_ms() - t0) * 1e-3;
          st.sam_bytes += len;
        });
        if (s != KSLAM_OK) err = g_err;
      } else if (s == KSLAM_OK && writer) {
        s = guarded([&] {
          kslam_tail_stats ts;
          memset(&ts, 0, sizeof ts);
          const double t0 = now_ms();
          const kslam_status a = kslam_tail_finish_write_rows(&host_sorted, &reads, index, res.overlaps, res.n_overlaps, res.cigar_pool,
                                                              res.n_cigar, res.details, res.md_pool, res.n_md, res.read_pairs,
                                                              res.n_read_pairs, res.pairs, res.n_pairs, kslam_write_queued,
                                                              (void *)writer, &ts);
          if (a != KSLAM_OK) fail(a, kslam_tail_last_error());
          st.seconds_sam_text += (now_ms() - t0) * 1e-3;
          st.sam_bytes += ts.sam_bytes;
        });
        if (s != KSLAM_OK) err = g_err;
      }
      if (tax_thread.joinable()) tax_thread.join();
      else if (taxdb && s == KSLAM_OK) tax_part();
      if (s == KSLAM_OK && tax_status != KSLAM_OK) {
        s = tax_status;
        err = tax_error;
      }
      kslam_release_batch(ctx, &res);
      if (s != KSLAM_OK && worker_status == KSLAM_OK) {
        worker_status = s;
        worker_error = err;            // (the failing thread's thread-local message)
      }
    };

    for (;;) {
      Window w;
      while (tickets.size() < depth && next_window(&w)) {
        uint64_t tk = 0;
        const double tsub = now_ms();
        // "at end of stream" for inner windows too: a window ends right after a terminator (kslam_fastq_batch_end looked at
        // the byte behind a closing "\r"), so the end-of-stream rule adds nothing and keeps that "\r" a whole terminator
        // (single end: r2 == NULL is how the library is told that there is one stream)
        if (kslam_submit_batch_fastq_text(ctx, r1 + w.p1, w.e1 - w.p1, paired ? r2 + w.p2 : nullptr, paired ? w.e2 - w.p2 : 0, 0, 1,
                                          &tk) != KSLAM_OK)
          fail(KSLAM_ERR_STATE, kslam_last_error(ctx));
        st.seconds_submitting += (now_ms() - tsub) * 1e-3;
        tickets.push_back(tk);
      }
      if (tickets.empty()) break;
      const double ta = now_ms();
      kslam_batch_result res;
      const uint64_t tk = tickets.front();
      tickets.pop_front();
      const kslam_status cs = kslam_collect_batch(ctx, tk, &res);
      const double tb = now_ms();
      st.seconds_waiting_for_gpu += (tb - ta) * 1e-3;
      if (worker.joinable()) worker.join();
      st.seconds_waiting_for_host_stage += (now_ms() - tb) * 1e-3;
      if (cs != KSLAM_OK) fail(cs, kslam_last_error(ctx));
      if (worker_status != KSLAM_OK) {
        kslam_release_batch(ctx, &res);
        fail(worker_status, worker_error);
      }
      if (res.n_reads == 0) {          // an empty batch ends the loop (src/SLAM.h:207)
        kslam_release_batch(ctx, &res);
        break;
      }
      if (!res.read_pairs && res.n_overlaps) {
        kslam_release_batch(ctx, &res);
        fail(KSLAM_ERR_INTERNAL, "the lane returned no device pairing");
      }
      worker = std::thread(host_stage, res);
    }
    if (worker.joinable()) worker.join();
    if (worker_status != KSLAM_OK) fail(worker_status, worker_error);
  });

  const double t_close = now_ms();
  const kslam_status closing = wind_down();
  st.seconds_closing = (now_ms() - t_close) * 1e-3;
  st.seconds = (now_ms() - t_begin) * 1e-3;
  if (stats) *stats = st;
  if (status != KSLAM_OK) return status;
  if (closing != KSLAM_OK) return closing;
  if (taxdb && tax_ids_out) {
    uint32_t *out = (uint32_t *)malloc(sizeof(uint32_t) * (all_ids.size() + 1));
    if (!out) return KSLAM_ERR_OOM;
    if (!all_ids.empty()) memcpy(out, all_ids.data(), sizeof(uint32_t) * all_ids.size());
    *tax_ids_out = out;
    if (n_tax_ids) *n_tax_ids = all_ids.size();
  }
  return KSLAM_OK;
}
