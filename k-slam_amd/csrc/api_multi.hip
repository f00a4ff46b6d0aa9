// api_multi.hip -- the C ABI, part 6: more than one device.  Export of a shard's rows in batch terms, the device merge of gathered
// shards, and kslam_multi_* (one process driving several GPUs, read pairs sharded, index replicated: SURVEY 8e).
#include "context.h"

namespace kslam_api {

kslam_status multi_fail(kslam_multi *m, kslam_status st, const std::string &msg) {
  m->err = msg;
  return st;
}

// run f(k) for every shard on its own host thread (every entry point of a context blocks on its stream)
template <typename F> kslam_status multi_for_each(kslam_multi *m, F &&f) {
  const size_t n = m->ctx.size();
  std::vector<kslam_status> st(n, KSLAM_OK);
  std::vector<std::thread> th;
  for (size_t k = 1; k < n; k++) th.emplace_back([&, k] { st[k] = f(k); });
  st[0] = f(0);
  for (auto &t : th) t.join();
  for (size_t k = 0; k < n; k++)
    if (st[k] != KSLAM_OK) return multi_fail(m, st[k], "shard " + std::to_string(k) + ": " + kslam_last_error(m->ctx[k]));
  return KSLAM_OK;
}


// a sibling context sees the primary's index through the same device pointers

}  // namespace kslam_api

extern "C" {

kslam_status kslam_merge_shards_device(kslam_ctx *c, uint32_t n_shards, const kslam_shard *shards, uint64_t n_pairs,
                                       const void *d_overlaps, const void *d_cigar_pools, void *d_out_overlaps,
                                       void *d_out_cigars) {
  return guarded(c, [&] {
    if (!shards || n_shards == 0 || n_shards > MERGE_MAX_SHARDS) throw StatusError{KSLAM_ERR_ARG, "1..256 shards"};
    std::vector<MergeShard> h(n_shards);
    uint64_t rows = 0, ops = 0;
    for (uint32_t k = 0; k < n_shards; k++) {
      const kslam_shard &sh = shards[k];
      if (sh.pair_hi < sh.pair_lo || sh.pair_hi > n_pairs || sh.pair_hi - sh.pair_lo >= (1ull << 31))
        throw StatusError{KSLAM_ERR_ARG, "shard " + std::to_string(k) + ": bad pair range"};
      if (k && sh.pair_lo < shards[k - 1].pair_hi) throw StatusError{KSLAM_ERR_ARG, "shards must be in batch order"};
      memset(&h[k], 0, sizeof(MergeShard));
      h[k].pair_lo = sh.pair_lo; h[k].pair_hi = sh.pair_hi;
      h[k].row_base = rows; h[k].n_rows = sh.n_rows; h[k].pool_base = ops;
      rows += sh.n_rows; ops += sh.n_cigar;
    }
    if (n_pairs >= (1ull << 31)) throw StatusError{KSLAM_ERR_UNSUPPORTED, "more than 2^31 pairs in one batch"};
    if (rows && (!d_overlaps || !d_out_overlaps)) throw StatusError{KSLAM_ERR_ARG, "null overlap buffers"};
    if (ops && (!d_cigar_pools || !d_out_cigars)) throw StatusError{KSLAM_ERR_ARG, "null cigar buffers"};
    hipStream_t s = c->stream;
    c->mg_shards.ensure(n_shards * sizeof(MergeShard));
    c->mg_lens.ensure((rows + 1) * sizeof(uint32_t));
    c->mg_off.ensure((rows + 1) * sizeof(uint64_t));
    c->mg_scan.ensure(scan_tmp_bytes(std::max<uint64_t>(rows, 1)));
    c->totals.ensure(8 * sizeof(uint64_t));
    HIPCHK(hipMemcpyAsync(c->mg_shards.p, h.data(), n_shards * sizeof(MergeShard), hipMemcpyHostToDevice, s));
    merge_shards((const kslam_overlap *)d_overlaps, rows, (const uint32_t *)d_cigar_pools, c->mg_shards.as<MergeShard>(),
                 n_shards, n_pairs, (kslam_overlap *)d_out_overlaps, (uint32_t *)d_out_cigars, c->mg_lens.as<uint32_t>(),
                 c->mg_off.as<uint64_t>(), c->totals.as<uint64_t>() + 4, c->mg_scan.p, s);
    HIPCHK(stream_wait(s));   // h[] is read by the copy above
  });
}

kslam_status kslam_shard_counts_device(kslam_ctx *c, uint64_t n_local_pairs, kslam_shard_counts *out) {
  return guarded(c, [&] {
    if (!out) throw StatusError{KSLAM_ERR_ARG, "null out"};
    if (n_local_pairs >= (1ull << 31)) throw StatusError{KSLAM_ERR_ARG, "n_local_pairs"};
    c->totals.ensure(8 * sizeof(uint64_t));
    uint64_t *d = c->totals.as<uint64_t>() + 4;
    shard_counts(c->res_ov.as<kslam_overlap>(), c->n_res, (uint32_t)n_local_pairs, c->n_cig, d, c->stream);
    uint64_t h[2] = {0, 0};
    read_back(h, d, sizeof h, c->stream);
    out->n_rows = c->n_res; out->n_rows_r1 = h[0];
    out->n_cigar = c->n_cig; out->n_cigar_r1 = h[1] == ~0ull ? c->n_cig : h[1];
  });
}

kslam_status kslam_export_shard_device(kslam_ctx *c, uint64_t n_local_pairs, uint64_t pair_lo, uint64_t n_pairs_total,
                                       uint64_t pool_base_r1, uint64_t pool_base_r2, void *d_rows_r1, void *d_rows_r2,
                                       void *d_pool_r1, void *d_pool_r2) {
  return guarded(c, [&] {
    kslam_shard_counts sc;
    // (the split again: cheap, and the caller cannot hand in counts that do not match the results)
    c->totals.ensure(8 * sizeof(uint64_t));
    uint64_t *d = c->totals.as<uint64_t>() + 4;
    shard_counts(c->res_ov.as<kslam_overlap>(), c->n_res, (uint32_t)n_local_pairs, c->n_cig, d, c->stream);
    uint64_t h[2] = {0, 0};
    read_back(h, d, sizeof h, c->stream);
    sc.n_rows = c->n_res; sc.n_rows_r1 = h[0]; sc.n_cigar = c->n_cig; sc.n_cigar_r1 = h[1] == ~0ull ? c->n_cig : h[1];
    if (pair_lo + n_local_pairs > n_pairs_total || n_pairs_total >= (1ull << 31))
      throw StatusError{KSLAM_ERR_ARG, "pair range outside the batch"};
    if ((sc.n_rows_r1 && !d_rows_r1) || (sc.n_rows > sc.n_rows_r1 && !d_rows_r2) || (sc.n_cigar_r1 && !d_pool_r1) ||
        (sc.n_cigar > sc.n_cigar_r1 && !d_pool_r2))
      throw StatusError{KSLAM_ERR_ARG, "null destination"};
    export_rows(c->res_ov.as<kslam_overlap>(), sc.n_rows, sc.n_rows_r1, (uint32_t)n_local_pairs, pair_lo, n_pairs_total,
                sc.n_cigar_r1, pool_base_r1, pool_base_r2, (kslam_overlap *)d_rows_r1, (kslam_overlap *)d_rows_r2,
                c->stream);
    if (sc.n_cigar_r1)
      HIPCHK(hipMemcpyAsync(d_pool_r1, c->res_cig.p, sc.n_cigar_r1 * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    if (sc.n_cigar > sc.n_cigar_r1)
      HIPCHK(hipMemcpyAsync(d_pool_r2, c->res_cig.as<uint32_t>() + sc.n_cigar_r1,
                            (sc.n_cigar - sc.n_cigar_r1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(stream_wait(c->stream));
  });
}

// ---- one process, several devices ---------------------------------------------------------------
kslam_status kslam_multi_create(const kslam_params *params, const int32_t *devices, uint32_t n_devices, kslam_multi **out) {
  if (!params || !devices || !out || n_devices == 0 || n_devices > MERGE_MAX_SHARDS) return KSLAM_ERR_ARG;
  kslam_multi *m = new (std::nothrow) kslam_multi();
  if (!m) return KSLAM_ERR_OOM;
  *out = m;
  for (uint32_t k = 0; k < n_devices; k++) {
    kslam_params p = *params;
    p.device = devices[k];
    kslam_ctx *c = nullptr;
    const kslam_status st = kslam_create(&p, &c);
    if (st != KSLAM_OK) {
      m->err = "device " + std::to_string(devices[k]) + ": " + (c ? kslam_last_error(c) : "create failed");
      kslam_destroy(c);
      return st;   // the caller reads the message and destroys m
    }
    m->ctx.push_back(c);
  }
  m->send.resize(n_devices);
  // peer access from the collecting device to the others (hipMemcpyPeerAsync works without it, through
  // the host; with it the copy goes over xGMI)
  for (uint32_t k = 1; k < n_devices; k++) {
    if (devices[k] == devices[0]) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, devices[0], devices[k]) == hipSuccess && can) {
      (void)hipSetDevice(devices[0]);
      (void)hipDeviceEnablePeerAccess(devices[k], 0);
      (void)hipGetLastError();   // "already enabled" is fine
    }
  }
  return KSLAM_OK;
}

void kslam_multi_destroy(kslam_multi *m) {
  if (!m) return;
  if (!m->ctx.empty() && m->ctx[0]->device >= 0) {
    (void)hipSetDevice(m->ctx[0]->device);
    m->rows_out.release(); m->pool_out.release();
  }
  for (size_t k = 0; k < m->send.size() && k < m->ctx.size(); k++)
    if (m->ctx[k]->device >= 0) { (void)hipSetDevice(m->ctx[k]->device); m->send[k].release(); }
  for (kslam_ctx *c : m->ctx) kslam_destroy(c);
  delete m;
}

const char *kslam_multi_last_error(const kslam_multi *m) { return m ? m->err.c_str() : "null handle"; }

kslam_status kslam_multi_set_index(kslam_multi *m, uint64_t n_entries, const char *const *bases, const uint64_t *lens) {
  if (!m || m->ctx.empty()) return KSLAM_ERR_ARG;
  return multi_for_each(m, [&](size_t k) { return kslam_set_index(m->ctx[k], n_entries, bases, lens); });
}

kslam_status kslam_multi_align_batch(kslam_multi *m, uint64_t n_reads, const char *const *bases, const uint32_t *lens,
                                     int paired, kslam_overlap **out, uint64_t *n_out, uint32_t **cigar_pool,
                                     uint64_t *n_cigar) {
  if (!m || m->ctx.empty() || !out || !n_out || !cigar_pool || !n_cigar) return KSLAM_ERR_ARG;
  *out = nullptr; *cigar_pool = nullptr; *n_out = 0; *n_cigar = 0;
  if (n_reads && (!bases || !lens)) return multi_fail(m, KSLAM_ERR_ARG, "null bases/lens");
  if (paired && (n_reads & 1)) return multi_fail(m, KSLAM_ERR_ARG, "a paired batch has an even number of reads");
  const uint64_t n_units = paired ? n_reads / 2 : n_reads;   // what is sharded: pairs, or single reads
  const size_t N = m->ctx.size();
  std::vector<kslam_shard> sh(N);
  // ---- shard + align, every device at once ----
  kslam_status st = multi_for_each(m, [&](size_t k) -> kslam_status {
    const uint64_t lo = n_units * k / N, hi = n_units * (k + 1) / N, nl = hi - lo;
    kslam_ctx *c = m->ctx[k];
    std::vector<uint64_t> off((paired ? 2 : 1) * nl + 1, 0);
    for (uint64_t i = 0; i < nl; i++) off[i + 1] = off[i] + lens[lo + i];
    if (paired) for (uint64_t i = 0; i < nl; i++) off[nl + i + 1] = off[nl + i] + lens[n_units + lo + i];
    const uint64_t n_loc = off.size() - 1;
    char *cat = nullptr;
    kslam_status s1 = guarded(c, [&] {
      cat = (char *)pinned_get(c, off[n_loc] + 64);
      for (uint64_t i = 0; i < nl; i++) memcpy(cat + off[i], bases[lo + i], lens[lo + i]);
      if (paired) for (uint64_t i = 0; i < nl; i++) memcpy(cat + off[nl + i], bases[n_units + lo + i], lens[n_units + lo + i]);
    });
    if (s1 == KSLAM_OK) s1 = kslam_load_reads(c, n_loc, cat, off.data());
    if (cat) pinned_put(c, cat);
    if (s1 == KSLAM_OK) s1 = kslam_align_resident(c, &sh[k].n_rows, &sh[k].n_cigar);
    sh[k].pair_lo = lo; sh[k].pair_hi = hi;
    return s1;
  });
  if (st != KSLAM_OK) return st;
  // ---- count exchange: where every shard's R1 rows, R2 rows and CIGAR words go in the batch ----
  std::vector<kslam_shard_counts> cnt(N);
  st = multi_for_each(m, [&](size_t k) { return kslam_shard_counts_device(m->ctx[k], sh[k].pair_hi - sh[k].pair_lo, &cnt[k]); });
  if (st != KSLAM_OK) return st;
  kslam_ctx *c0 = m->ctx[0];
  uint64_t rows = 0, ops = 0, rows_r1 = 0, ops_r1 = 0;
  for (size_t k = 0; k < N; k++) { rows += cnt[k].n_rows; ops += cnt[k].n_cigar; rows_r1 += cnt[k].n_rows_r1; ops_r1 += cnt[k].n_cigar_r1; }
  std::vector<uint64_t> row1(N), row2(N), op1(N), op2(N);
  {
    uint64_t a = 0, b = rows_r1, c = 0, d = ops_r1;
    for (size_t k = 0; k < N; k++) {
      row1[k] = a; a += cnt[k].n_rows_r1;
      row2[k] = b; b += cnt[k].n_rows - cnt[k].n_rows_r1;
      op1[k] = c; c += cnt[k].n_cigar_r1;
      op2[k] = d; d += cnt[k].n_cigar - cnt[k].n_cigar_r1;
    }
  }
  kslam_overlap *ho = nullptr;
  uint32_t *hc = nullptr;
  st = guarded(c0, [&] {
    m->rows_out.ensure((rows + 1) * sizeof(kslam_overlap));
    m->pool_out.ensure((ops + 1) * sizeof(uint32_t));
  });
  if (st != KSLAM_OK) return multi_fail(m, st, kslam_last_error(c0));
  // ---- the one exchange of the path: every shard re-bases its own records (its own GPU, all at once);
  // the shard on the collecting device writes straight into the final arrays, the others into a send
  // buffer that one peer copy per piece moves into place ----
  st = multi_for_each(m, [&](size_t k) -> kslam_status {
    kslam_ctx *ck = m->ctx[k];
    kslam_overlap *fo = m->rows_out.as<kslam_overlap>();
    uint32_t *fp = m->pool_out.as<uint32_t>();
    const uint64_t n1 = cnt[k].n_rows_r1, n2 = cnt[k].n_rows - n1, c1 = cnt[k].n_cigar_r1, c2 = cnt[k].n_cigar - c1;
    const uint64_t nl = sh[k].pair_hi - sh[k].pair_lo;
    if (k == 0)
      return kslam_export_shard_device(ck, nl, sh[k].pair_lo, n_units, op1[k], op2[k], fo + row1[k], fo + row2[k],
                                       fp + op1[k], fp + op2[k]);
    DevBuf &sb = m->send[k];
    kslam_status s1 = guarded(ck, [&] { sb.ensure((n1 + n2 + 1) * sizeof(kslam_overlap) + (c1 + c2 + 1) * sizeof(uint32_t)); });
    if (s1 != KSLAM_OK) return s1;
    kslam_overlap *so = sb.as<kslam_overlap>();
    uint32_t *sp = reinterpret_cast<uint32_t *>(so + n1 + n2);
    s1 = kslam_export_shard_device(ck, nl, sh[k].pair_lo, n_units, op1[k], op2[k], so, so + n1, sp, sp + c1);
    if (s1 != KSLAM_OK) return s1;
    return guarded(ck, [&] {
      if (n1) HIPCHK(hipMemcpyPeerAsync(fo + row1[k], c0->device, so, ck->device, n1 * sizeof(kslam_overlap), ck->stream));
      if (n2) HIPCHK(hipMemcpyPeerAsync(fo + row2[k], c0->device, so + n1, ck->device, n2 * sizeof(kslam_overlap), ck->stream));
      if (c1) HIPCHK(hipMemcpyPeerAsync(fp + op1[k], c0->device, sp, ck->device, c1 * sizeof(uint32_t), ck->stream));
      if (c2) HIPCHK(hipMemcpyPeerAsync(fp + op2[k], c0->device, sp + c1, ck->device, c2 * sizeof(uint32_t), ck->stream));
      HIPCHK(stream_wait(ck->stream));
    });
  });
  if (st != KSLAM_OK) return st;
  st = guarded(c0, [&] {
    ho = (kslam_overlap *)pinned_get(c0, (rows + 1) * sizeof(kslam_overlap));
    hc = (uint32_t *)pinned_get(c0, (ops + 1) * sizeof(uint32_t));
    if (rows) HIPCHK(hipMemcpyAsync(ho, m->rows_out.p, rows * sizeof(kslam_overlap), hipMemcpyDeviceToHost, c0->stream));
    if (ops) HIPCHK(hipMemcpyAsync(hc, m->pool_out.p, ops * sizeof(uint32_t), hipMemcpyDeviceToHost, c0->stream));
    HIPCHK(stream_wait(c0->stream));
  });
  if (st != KSLAM_OK) {
    if (ho) pinned_put(c0, ho);
    if (hc) pinned_put(c0, hc);
    return multi_fail(m, st, kslam_last_error(c0));
  }
  *out = ho; *n_out = rows; *cigar_pool = hc; *n_cigar = ops;
  return KSLAM_OK;
}

void kslam_multi_free_batch(kslam_multi *m, kslam_overlap *out, uint32_t *cigar_pool) {
  if (m && !m->ctx.empty()) kslam_free_batch(m->ctx[0], out, cigar_pool);
}


}  // extern "C"
