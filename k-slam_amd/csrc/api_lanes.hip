// api_lanes.hip -- the C ABI, part 5: the operator pipelined.  Batches alternate between worker lanes (a host thread + a
// sibling context that borrows the index each); columns, FASTQ fields or whole FASTQ texts in, results (+ pairs, details, SAM text)
// out through kslam_wait_batch / kslam_collect_batch.
#include "context.h"

namespace kslam_api {

kslam_status load_reads_from_fastq(kslam_ctx *c, uint64_t n_reads, const char *r1, uint64_t len1, const char *r2,
                                   uint64_t len2, const uint64_t *offsets, const uint64_t *bases_at,
                                   const uint64_t *quality_at) {
  return guarded(c, [&] {
    if (n_reads && (!offsets || !bases_at || !quality_at || (len1 && !r1) || (len2 && !r2)))
      throw StatusError{KSLAM_ERR_ARG, "null argument"};
    hipStream_t s = c->stream;
    c->have_reads = false;
    c->n_reads = n_reads;
    c->h_roff.assign(n_reads + 1, 0);
    const uint64_t o0 = n_reads ? offsets[0] : 0;
    for (uint64_t i = 0; i <= n_reads && n_reads; i++) c->h_roff[i] = offsets[i] - o0;
    const uint64_t total = c->h_roff[n_reads];
    c->fq_text.ensure(len1 + len2 + 64);
    if (len1) HIPCHK(hipMemcpyAsync(c->fq_text.p, r1, len1, hipMemcpyHostToDevice, s));
    if (len2) HIPCHK(hipMemcpyAsync(c->fq_text.as<uint8_t>() + len1, r2, len2, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(c->fq_text.as<uint8_t>() + len1 + len2, 0, 64, s));
    c->fq_bases_at.ensure((n_reads + 1) * sizeof(uint64_t));
    c->fq_qual_at.ensure((n_reads + 1) * sizeof(uint64_t));
    c->r_off.ensure((n_reads + 1) * sizeof(uint64_t));
    if (n_reads) {
      HIPCHK(hipMemcpyAsync(c->fq_bases_at.p, bases_at, n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, s));
      HIPCHK(hipMemcpyAsync(c->fq_qual_at.p, quality_at, n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipMemcpyAsync(c->r_off.p, c->h_roff.data(), (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    c->r_bases.ensure(total + 64);
    c->r_qual.ensure(total + 64);
    // a field that reaches past the end of the texts would be a broken index: check on the host
    for (uint64_t i = 0; i < n_reads; i++) {
      const uint64_t len = c->h_roff[i + 1] - c->h_roff[i];
      if (bases_at[i] + len > len1 + len2 || quality_at[i] + len > len1 + len2)
        throw StatusError{KSLAM_ERR_ARG, "field " + std::to_string(i) + " lies outside the FASTQ texts"};
    }
    gather_fields(c->fq_text.as<uint8_t>(), c->fq_bases_at.as<uint64_t>(), c->fq_qual_at.as<uint64_t>(),
                  c->r_off.as<uint64_t>(), n_reads, c->r_bases.as<uint8_t>(), c->r_qual.as<uint8_t>(), s);
    HIPCHK(hipMemsetAsync(c->r_bases.as<uint8_t>() + total, 0, 64, s));
    HIPCHK(hipMemsetAsync(c->r_qual.as<uint8_t>() + total, 0, 64, s));
    finish_load_reads(c);
    c->have_qual = true;
  });
}


// kslam_submit_batch_fastq_text: texts up, record index + columns on the device, the host's columns back
kslam_status load_reads_from_fastq_text(kslam_ctx *c, kslam_ctx::AsyncJob *job) {
  return guarded(c, [&] {
    const char *r1 = job->cat, *r2 = job->qcat;
    const uint64_t len1 = job->len1, len2 = job->len2;
    if ((len1 && !r1) || (len2 && !r2)) throw StatusError{KSLAM_ERR_ARG, "null text"};
    hipStream_t s = c->stream;
    c->have_reads = false;
    c->fq_text.ensure(len1 + len2 + 64);
    if (len1) HIPCHK(hipMemcpyAsync(c->fq_text.p, r1, len1, hipMemcpyHostToDevice, s));
    if (len2) HIPCHK(hipMemcpyAsync(c->fq_text.as<uint8_t>() + len1, r2, len2, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(c->fq_text.as<uint8_t>() + len1 + len2, 0, 64, s));
    FastqIndexResult ix;
    fastq_index_device(c->fq_text.as<uint8_t>(), len1, len2, len1 ? (const uint8_t *)r1 + len1 - 1 : nullptr,
                       len2 ? (const uint8_t *)r2 + len2 - 1 : nullptr, job->max_pairs, job->at_eof != 0, c->fqw, &ix, s, job->single);
    const uint64_t n = ix.n_reads;
    // the host's columns: offsets (= lengths), identifiers
    job->r_n = n;
    job->r_off = (uint64_t *)pinned_get(c, (n + 2) * sizeof(uint64_t));
    job->r_ids_off = (uint64_t *)pinned_get(c, (n + 2) * sizeof(uint64_t));
    job->r_ids = (char *)pinned_get(c, ix.ids_total + 64);
    HIPCHK(hipMemcpyAsync(job->r_off, ix.d_bases_off, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(job->r_ids_off, ix.d_ids_off, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    if (ix.ids_total) HIPCHK(hipMemcpyAsync(job->r_ids, ix.d_ids, ix.ids_total, hipMemcpyDeviceToHost, s));
    job->consumed[0] = ix.consumed[0];
    job->consumed[1] = ix.consumed[1];
    // the device's columns
    c->n_reads = n;
    c->r_off.ensure((n + 1) * sizeof(uint64_t));
    HIPCHK(hipMemcpyAsync(c->r_off.p, ix.d_bases_off, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
    c->r_bases.ensure(ix.bases_total + 64);
    c->r_qual.ensure(ix.bases_total + 64);
    gather_fields(c->fq_text.as<uint8_t>(), ix.d_bases_at, ix.d_quality_at, c->r_off.as<uint64_t>(), n,
                  c->r_bases.as<uint8_t>(), c->r_qual.as<uint8_t>(), s);
    HIPCHK(hipMemsetAsync(c->r_bases.as<uint8_t>() + ix.bases_total, 0, 64, s));
    HIPCHK(hipMemsetAsync(c->r_qual.as<uint8_t>() + ix.bases_total, 0, 64, s));
    HIPCHK(stream_wait(s));
    job->r_ids[ix.ids_total] = 0;
    c->h_roff.assign(job->r_off, job->r_off + n + 1);
    finish_load_reads(c);
    c->have_qual = true;
    c->d_ids = ix.d_ids;                 // in c->fqw: valid until this context indexes its next batch
    c->d_ids_off = ix.d_ids_off;
    c->have_ids = true;
    job->n_reads = n;
  });
}

void lane_main(kslam_ctx *primary, kslam_ctx::AsyncLane *lane) {
  kslam_host::name_thread("kslam-lane");
  wait_mode().yield = primary->tune.lane_waits_yield;   // common.h: stream_wait
  if (wait_mode().yield) (void)prctl(PR_SET_TIMERSLACK, 5000UL, 0, 0, 0);   // its 20 us sleeps mean 25, not 70
  for (;;) {
    kslam_ctx::AsyncJob *job = nullptr;
    {
      std::unique_lock<std::mutex> lk(primary->as_mu);
      primary->as_cv.wait(lk, [&] { return primary->as_stop || !lane->q.empty(); });
      if (lane->q.empty()) return;   // stop requested and nothing left
      job = lane->q.front();
      lane->q.pop_front();
    }
    kslam_ctx *c = lane->c;
    const bool dbg = primary->tune.debug;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    double t1 = 0, t2 = 0, t3 = 0;
    kslam_status st;
    if (job->fastq_text)
      st = load_reads_from_fastq_text(c, job);
    else if (job->fastq)
      st = load_reads_from_fastq(c, job->n_reads, job->cat, job->len1, job->qcat, job->len2, job->off_ptr,
                                 job->bases_at, job->quality_at);
    else
      st = kslam_load_reads(c, job->n_reads, job->cat, job->borrowed ? job->off_ptr : job->off.data());
    if (st == KSLAM_OK && job->qcat && !job->fastq)
      st = kslam_load_qualities(c, job->borrowed && job->n_reads ? job->qcat + job->off_ptr[0] : job->qcat);
    if (job->qcat && !job->borrowed) { pinned_put(c, job->qcat); }
    t1 = now();
    if (!job->borrowed) pinned_put(c, job->cat);
    job->cat = nullptr;
    SamStage sam;
    bool sam_planned = false, text_wanted = false, details_wanted = false;
    auto want_details_or_no_cigar = [](kslam_ctx *cc, bool wd) { return wd || !cc->prm.report_cigar; };
    if (st == KSLAM_OK) {
      // one lane computes at a time: the kernels of a batch fill the chip, so two batches computing at
      // once only time-slice -- and, worse, fall into step, both lanes copying while the GPU idles and
      // both computing afterwards (measured: 32.6 ms per batch against 27.3 resident).  With the token
      // the lanes run in anti-phase: one computes while the other downloads its last result and
      // uploads its next batch.
      std::lock_guard<std::mutex> compute(primary->as_compute);
      t2 = now();
      const bool want_details = job->qcat || job->fastq;
      PairingHook hook{primary->pairing.paired, primary->pairing.thr, primary->pairing.fraction, primary->pairing.stages};
      const bool eager = primary->tune.eager_cigar;   // A/B: every CIGAR, pairing afterwards
      const bool use_hook = primary->pairing.stages && want_details && !eager;
      st = guarded(c, [&] { align_resident(c, false, nullptr, use_hook ? &hook : nullptr); });
      if (st == KSLAM_OK && primary->pairing.stages) {
        if (hook.ran) fill_pair_stats(c->pres, &job->pstats);
        else st = kslam_pair_screen(c, primary->pairing.paired, primary->pairing.thr, primary->pairing.fraction,
                                    primary->pairing.stages, &job->pstats);
      }
      details_wanted = st == KSLAM_OK && want_details;
      text_wanted = st == KSLAM_OK && want_details_or_no_cigar(c, want_details);
      if (details_wanted && primary->tune.details_in_token) {   // default; KSLAM_DETAILS_IN_TOKEN=0 moves it out (measured: 40.5 against 43.1 M reads/s)
        st = primary->pairing.stages ? kslam_row_details_of_pairs(c, nullptr) : kslam_row_details(c, nullptr);
        details_wanted = false;
        if (st != KSLAM_OK) text_wanted = false;
      }
    }
    // (KSLAM_DETAILS_IN_TOKEN=0, A/B: the per-row walk outside the token as well -- slower: its 150-byte windows of the index
    // compete with the other lane's staging for L2)
    if (details_wanted) {
      st = primary->pairing.stages ? kslam_row_details_of_pairs(c, nullptr) : kslam_row_details(c, nullptr);
      if (st != KSLAM_OK) text_wanted = false;
    }
    // The SAM records / per-read lines on the device (kslam_set_sam_text): the reference's per-pair sort (in place: the pairs
    // go back to the host in that order), the rows to report, the log-probabilities the host evaluates with its libm, then the
    // text.  OUTSIDE the compute token: these kernels are bound by the latency of dependent gathers (samtext.hip), not by
    // ALU work or bandwidth, so they run next to the other lane's alignment kernels instead of in front of them.
    // Not for a batch whose pseudo-assembly the device left to the host: its scores are not final yet.
    const bool text_on = primary->samtext.sam || primary->samtext.per_read;
    const bool pseudo_left = (primary->pairing.stages & 4u) && !(job->pstats.stages_done & 4u);
    if (text_wanted && text_on && primary->pairing.stages && c->have_ids && !pseudo_left) {
      st = guarded(c, [&] {
        sam_stage_plan(c, primary, primary->pairing.paired, primary->samtext.num_alignments, primary->samtext.sam_xa, primary->samtext.sam, sam);
      });
      sam_planned = st == KSLAM_OK;
    }
    if (sam_planned) {
      sam_stage_mapq(sam);   // pow / log10 / ceil with the host's libm
      st = guarded(c, [&] { sam_stage_kernels(c, primary, sam, primary->samtext.sam, primary->samtext.per_read); });
      uint64_t n_tax = 0;
      if (st == KSLAM_OK)
        st = guarded(c, [&] {
          sam_stage_fetch(c, sam, primary->samtext.sam, primary->samtext.per_read, &job->sam_text, &job->sam_len, &job->pr_text, &job->pr_len,
                          &job->tax, &n_tax);
        });
      if (st == KSLAM_OK)
        job->text_flags = (primary->samtext.sam ? (KSLAM_TEXT_PAIRS_SORTED | KSLAM_TEXT_SAM) : 0u) | (primary->samtext.per_read ? KSLAM_TEXT_PER_READ : 0u);
    }
    sam_stage_free(c, sam);
    t3 = now();
    // with the SAM records written on the device the host has no use for the rows, the CIGAR pool, the per-row details and
    // the MD text (0.7 GB per batch of configs[1]): they stay where they are, only their counts travel
    const bool text_sam = (job->text_flags & KSLAM_TEXT_SAM) != 0;
    if (st == KSLAM_OK && !text_sam) st = kslam_take_results(c, &job->out, &job->n_out, &job->pool, &job->n_cig);
    if (st == KSLAM_OK && text_sam) { job->n_out = c->n_res; job->n_cig = c->n_cig; }
    const double t4 = now();
    if (st == KSLAM_OK && (job->qcat || job->fastq) && !text_sam) st = kslam_take_row_details(c, &job->det, &job->md, &job->n_md);
    const double t5 = now();
    if (st == KSLAM_OK && primary->pairing.stages) st = kslam_take_pairs(c, &job->rp, &job->n_rp, &job->pr, &job->n_pr);
    if (dbg) fprintf(stderr, "[kslam]   align phases: extract %.2f sort %.2f join %.2f sw %.2f cigar %.2f total %.2f ms\n", c->tm.ms_extract,
                     c->tm.ms_sort, c->tm.ms_join, c->tm.ms_sw, c->tm.ms_cigar, c->tm.ms_total);
    if (dbg) fprintf(stderr, "[kslam] t=%.1f lane %p ticket %llu: upload %.2f, token wait %.2f, align %.2f, download %.2f (rows %.2f, details %.2f, pairs %.2f) ms\n",
                     fmod(t0, 100000.0), (void *)lane, (unsigned long long)job->ticket, t1 - t0, t2 - t1, t3 - t2, now() - t3, t4 - t3, t5 - t4, now() - t5);
    {
      std::lock_guard<std::mutex> lk(primary->as_mu);
      job->st = st;
      if (st != KSLAM_OK) job->err = c->err;
      job->done = true;
    }
    primary->as_cv.notify_all();
  }
}


void ensure_lanes(kslam_ctx *c) {
  if (!c->lanes.empty()) return;
  const int n_lanes = c->tune.lanes;
  // built aside and published only when every lane has its context AND its thread: a failure half way
  // (page-locked or device memory) must not leave lanes without workers behind, to which the next
  // submit would queue a job nobody ever runs
  std::vector<kslam_ctx::AsyncLane *> fresh;
  auto undo = [&] {
    for (auto *l : fresh) { kslam_destroy(l->c); delete l; }
    fresh.clear();
  };
  for (int k = 0; k < n_lanes; k++) {
    kslam_ctx *lc = nullptr;
    const kslam_status s1 = kslam_create(&c->prm, &lc);
    if (s1 != KSLAM_OK) {
      const std::string msg = lc ? lc->err : "lane context";
      kslam_destroy(lc);
      undo();
      throw StatusError{s1, msg};
    }
    lc->tune = c->tune;
    lc->pw.pseudo_cap = (uint32_t)c->tune.pseudo_cap;
    share_index(lc, c);
    auto *l = new kslam_ctx::AsyncLane();
    l->c = lc;
    fresh.push_back(l);
  }
  size_t started = 0;
  try {
    for (auto *l : fresh) { l->th = std::thread(lane_main, c, l); started++; }
  } catch (const std::exception &e) {
    { std::lock_guard<std::mutex> lk(c->as_mu); c->as_stop = true; }
    c->as_cv.notify_all();
    for (size_t k = 0; k < started; k++) fresh[k]->th.join();
    c->as_stop = false;
    undo();
    throw StatusError{KSLAM_ERR_OOM, std::string("could not start a lane thread: ") + e.what()};
  }
  c->lanes = std::move(fresh);
}

void stop_lanes(kslam_ctx *c) {
  {
    std::lock_guard<std::mutex> lk(c->as_mu);
    c->as_stop = true;
  }
  c->as_cv.notify_all();
  for (auto *l : c->lanes) {
    if (l->th.joinable()) l->th.join();
    kslam_destroy(l->c);
    delete l;
  }
  c->lanes.clear();
  for (auto &kv : c->jobs) {   // results nobody waited for
    kslam_ctx::AsyncJob *j = kv.second;
    delete j;
  }
  c->jobs.clear();
  c->as_stop = false;
}

}  // namespace kslam_api

extern "C" {

// ---- the operator, pipelined: batches alternate between two worker lanes (a host thread + a sibling
// context with its own stream and work buffers each), so the upload of batch k+1 and the download of
// batch k-1 run under the kernels of batch k, and one lane's host read-backs are covered by the other
// lane's kernels ----
kslam_status kslam_align_batch_async(kslam_ctx *c, uint64_t n_reads, const char *const *bases, const uint32_t *lens,
                                     uint64_t *ticket) {
  return kslam_submit_batch(c, n_reads, bases, nullptr, lens, ticket);
}

kslam_status kslam_submit_batch(kslam_ctx *c, uint64_t n_reads, const char *const *bases, const char *const *quality,
                                const uint32_t *lens, uint64_t *ticket) {
  if (!c || !ticket) return KSLAM_ERR_ARG;
  kslam_ctx::AsyncJob *job = nullptr;
  kslam_status st = guarded(c, [&] {
    if (n_reads && (!bases || !lens)) throw StatusError{KSLAM_ERR_ARG, "null bases/lens"};
    if (!c->have_index) throw StatusError{KSLAM_ERR_STATE, "kslam_set_index has not been called"};
    ensure_lanes(c);
    job = new kslam_ctx::AsyncJob();
    job->n_reads = n_reads;
    job->off.assign(n_reads + 1, 0);
    for (uint64_t i = 0; i < n_reads; i++) job->off[i + 1] = job->off[i] + lens[i];
  });
  if (st != KSLAM_OK) { delete job; return st; }
  // the reads leave the caller's memory now (parallel gather into a page-locked buffer of the lane that
  // will run the batch): the caller may reuse its buffers as soon as this call returns
  uint64_t tk;
  {
    std::lock_guard<std::mutex> lk(c->as_mu);
    tk = c->next_ticket++;
  }
  kslam_ctx::AsyncLane *lane = c->lanes[tk % c->lanes.size()];
  st = guarded(c, [&] {
    job->cat = (char *)pinned_get(lane->c, job->off[n_reads] + 64);
    if (quality) job->qcat = (char *)pinned_get(lane->c, job->off[n_reads] + 64);
    unsigned nt = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (n_reads < 100000) nt = 1;
    std::vector<std::thread> th;
    char *cat = job->cat, *qcat = job->qcat;
    const std::vector<uint64_t> &off = job->off;
    for (unsigned t = 0; t < nt; t++) {
      const uint64_t lo = n_reads * t / nt, hi = n_reads * (t + 1) / nt;
      auto work = [=, &off] {
        for (uint64_t i = lo; i < hi; i++) memcpy(cat + off[i], bases[i], lens[i]);
        if (qcat) for (uint64_t i = lo; i < hi; i++) memcpy(qcat + off[i], quality[i], lens[i]);
      };
      if (nt == 1) work(); else th.emplace_back(work);
    }
    for (auto &x : th) x.join();
  });
  if (st != KSLAM_OK) {
    if (job->cat) pinned_put(lane->c, job->cat);
    if (job->qcat) pinned_put(lane->c, job->qcat);
    delete job;
    return st;
  }
  job->ticket = tk;
  {
    std::lock_guard<std::mutex> lk(c->as_mu);
    c->jobs[tk] = job;
    lane->q.push_back(job);
  }
  c->as_cv.notify_all();
  *ticket = tk;
  return KSLAM_OK;
}

kslam_status kslam_submit_batch_columns(kslam_ctx *c, uint64_t n_reads, const char *bases, const char *quality,
                                        const uint64_t *offsets, uint64_t *ticket) {
  if (!c || !ticket) return KSLAM_ERR_ARG;
  kslam_ctx::AsyncJob *job = nullptr;
  kslam_status st = guarded(c, [&] {
    if (n_reads && (!bases || !offsets)) throw StatusError{KSLAM_ERR_ARG, "null bases/offsets"};
    if (!c->have_index) throw StatusError{KSLAM_ERR_STATE, "kslam_set_index has not been called"};
    ensure_lanes(c);
    job = new kslam_ctx::AsyncJob();
    job->n_reads = n_reads;
    job->borrowed = true;
    job->cat = const_cast<char *>(bases);
    job->qcat = const_cast<char *>(quality);
    job->off_ptr = offsets;
  });
  if (st != KSLAM_OK) { delete job; return st; }
  uint64_t tk;
  {
    std::lock_guard<std::mutex> lk(c->as_mu);
    tk = c->next_ticket++;
    job->ticket = tk;
    c->jobs[tk] = job;
    c->lanes[tk % c->lanes.size()]->q.push_back(job);
  }
  c->as_cv.notify_all();
  *ticket = tk;
  return KSLAM_OK;
}

kslam_status kslam_submit_batch_fastq(kslam_ctx *c, const char *r1, uint64_t len1, const char *r2, uint64_t len2,
                                      uint64_t n_reads, const uint64_t *offsets, const uint64_t *bases_at,
                                      const uint64_t *quality_at, uint64_t *ticket) {
  if (!c || !ticket) return KSLAM_ERR_ARG;
  kslam_ctx::AsyncJob *job = nullptr;
  kslam_status st = guarded(c, [&] {
    if (n_reads && (!offsets || !bases_at || !quality_at)) throw StatusError{KSLAM_ERR_ARG, "null layout"};
    if (!c->have_index) throw StatusError{KSLAM_ERR_STATE, "kslam_set_index has not been called"};
    ensure_lanes(c);
    job = new kslam_ctx::AsyncJob();
    job->n_reads = n_reads;
    job->borrowed = true;
    job->fastq = true;
    job->cat = const_cast<char *>(r1);
    job->qcat = const_cast<char *>(r2);
    job->len1 = len1; job->len2 = len2;
    job->off_ptr = offsets;
    job->bases_at = bases_at; job->quality_at = quality_at;
  });
  if (st != KSLAM_OK) { delete job; return st; }
  uint64_t tk;
  {
    std::lock_guard<std::mutex> lk(c->as_mu);
    tk = c->next_ticket++;
    job->ticket = tk;
    c->jobs[tk] = job;
    c->lanes[tk % c->lanes.size()]->q.push_back(job);
  }
  c->as_cv.notify_all();
  *ticket = tk;
  return KSLAM_OK;
}

kslam_status kslam_submit_batch_fastq_text(kslam_ctx *c, const char *r1, uint64_t len1, const char *r2, uint64_t len2,
                                           uint64_t max_pairs, int at_eof, uint64_t *ticket) {
  if (!c || !ticket) return KSLAM_ERR_ARG;
  kslam_ctx::AsyncJob *job = nullptr;
  kslam_status st = guarded(c, [&] {
    if ((len1 && !r1) || (len2 && !r2)) throw StatusError{KSLAM_ERR_ARG, "null text"};
    if (!c->have_index) throw StatusError{KSLAM_ERR_STATE, "kslam_set_index has not been called"};
    ensure_lanes(c);
    job = new kslam_ctx::AsyncJob();
    job->borrowed = true;
    job->fastq = true;
    job->fastq_text = true;
    job->cat = const_cast<char *>(r1);
    job->qcat = const_cast<char *>(r2);
    job->len1 = len1; job->len2 = len2;
    job->max_pairs = max_pairs; job->at_eof = at_eof;
    job->single = r2 == nullptr && len2 == 0;
  });
  if (st != KSLAM_OK) { delete job; return st; }
  uint64_t tk;
  {
    std::lock_guard<std::mutex> lk(c->as_mu);
    tk = c->next_ticket++;
    job->ticket = tk;
    c->jobs[tk] = job;
    c->lanes[tk % c->lanes.size()]->q.push_back(job);
  }
  c->as_cv.notify_all();
  *ticket = tk;
  return KSLAM_OK;
}

kslam_status kslam_wait_batch(kslam_ctx *c, uint64_t ticket, kslam_overlap **out, uint64_t *n_out, uint32_t **cigar_pool,
                              uint64_t *n_cigar) {
  if (!c || !out || !n_out || !cigar_pool || !n_cigar) return KSLAM_ERR_ARG;
  *out = nullptr; *cigar_pool = nullptr; *n_out = 0; *n_cigar = 0;
  kslam_batch_result r;
  const kslam_status st = kslam_collect_batch(c, ticket, &r);
  if (st != KSLAM_OK) return st;
  kslam_free_pinned(c, r.details);
  kslam_free_pinned(c, r.md_pool);
  kslam_free_pinned(c, r.read_pairs);
  kslam_free_pinned(c, r.pairs);
  kslam_free_pinned(c, r.reads_bases_off);
  kslam_free_pinned(c, r.reads_ids_off);
  kslam_free_pinned(c, r.reads_ids);
  *out = r.overlaps; *n_out = r.n_overlaps; *cigar_pool = r.cigar_pool; *n_cigar = r.n_cigar;
  return KSLAM_OK;
}

void kslam_release_batch(kslam_ctx *c, kslam_batch_result *r) {
  if (!c || !r) return;
  kslam_free_batch(c, r->overlaps, r->cigar_pool);
  kslam_free_pinned(c, r->details);
  kslam_free_pinned(c, r->md_pool);
  kslam_free_pinned(c, r->read_pairs);
  kslam_free_pinned(c, r->pairs);
  kslam_free_pinned(c, r->reads_bases_off);
  kslam_free_pinned(c, r->reads_ids_off);
  kslam_free_pinned(c, r->reads_ids);
  kslam_free_pinned(c, r->sam_text);
  kslam_free_pinned(c, r->per_read_text);
  kslam_free_pinned(c, r->tax_ids);
  memset(r, 0, sizeof *r);
}

kslam_status kslam_collect_batch(kslam_ctx *c, uint64_t ticket, kslam_batch_result *res) {
  if (!c || !res) return KSLAM_ERR_ARG;
  memset(res, 0, sizeof *res);
  kslam_ctx::AsyncJob *job = nullptr;
  {
    std::unique_lock<std::mutex> lk(c->as_mu);
    auto it = c->jobs.find(ticket);
    if (it == c->jobs.end()) { c->err = "no such ticket (already waited for?)"; return KSLAM_ERR_ARG; }
    job = it->second;
    c->as_cv.wait(lk, [&] { return job->done; });
    c->jobs.erase(it);
  }
  const kslam_status st = job->st;
  if (st == KSLAM_OK) {
    res->overlaps = job->out; res->n_overlaps = job->n_out; res->cigar_pool = job->pool; res->n_cigar = job->n_cig;
    res->details = job->det; res->md_pool = job->md; res->n_md = job->n_md;
    res->read_pairs = job->rp; res->n_read_pairs = job->n_rp; res->pairs = job->pr; res->n_pairs = job->n_pr;
    res->pair_stats = job->pstats;
    res->n_reads = job->r_n; res->reads_bases_off = job->r_off; res->reads_ids = job->r_ids; res->reads_ids_off = job->r_ids_off;
    res->consumed1 = job->consumed[0]; res->consumed2 = job->consumed[1];
    res->sam_text = job->sam_text; res->sam_text_len = job->sam_len; res->per_read_text = job->pr_text; res->per_read_len = job->pr_len;
    res->tax_ids = job->tax; res->text_flags = job->text_flags;
  } else {
    kslam_free_pinned(c, job->sam_text);
    kslam_free_pinned(c, job->pr_text);
    kslam_free_pinned(c, job->tax);
    c->err = job->err;
    kslam_free_batch(c, job->out, job->pool);
    kslam_free_pinned(c, job->det);
    kslam_free_pinned(c, job->md);
    kslam_free_pinned(c, job->rp);
    kslam_free_pinned(c, job->pr);
    kslam_free_pinned(c, job->r_off);
    kslam_free_pinned(c, job->r_ids_off);
    kslam_free_pinned(c, job->r_ids);
  }
  delete job;
  return st;
}

}  // extern "C"
