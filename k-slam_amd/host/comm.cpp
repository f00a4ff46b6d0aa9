// comm.cpp -- include/kslam_comm.h: the end-of-batch exchanges of a read-sharded batch over RCCL, one process per GPU,
// on top of the library's own C ABI (kslam_shard_counts_device / kslam_export_shard_device / kslam_pair_phase_a, _b /
// kslam_pseudo_merged) and the HIP runtime.  librccl.so is opened with dlopen on first use.
//
// The protocol is the one k-slam_amd/dist.py runs through torch.distributed: counts by
// ncclAllGather, then grouped ncclSend / ncclRecv whose pieces land in their final places on rank 0; the variable-length
// all-gathers of the sharded tail pad every rank's piece to the longest.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/kslam_comm.h"

namespace {

thread_local std::string g_err;

struct Fail {
  kslam_status st;
  std::string msg;
};
[[noreturn]] void fail(kslam_status st, const std::string &m) { throw Fail{st, m}; }

template <class F>
kslam_status guarded(F &&f) {
  try {
    f();
    return KSLAM_OK;
  } catch (const Fail &e) {
    g_err = e.msg;
    return e.st;
  } catch (const std::bad_alloc &) {
    g_err = "out of host memory";
    return KSLAM_ERR_OOM;
  }
}

void hipchk(hipError_t e, const char *what) {
  if (e != hipSuccess) fail(KSLAM_ERR_NO_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

// ---- the handful of RCCL entry points, resolved at run time (rccl.h's signatures) ----
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[KSLAM_COMM_ID_BYTES]; } ncclUniqueId;
enum { ncclUint8 = 1, ncclUint64 = 5 };
struct Rccl {
  void *so = nullptr;
  int (*GetUniqueId)(ncclUniqueId *) = nullptr;
  int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
Rccl &rccl() {
  static Rccl r;
  if (r.so) return r;
  const char *names[] = {getenv("KSLAM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    if (!n || !*n) continue;
    r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.so) break;
  }
  if (!r.so) fail(KSLAM_ERR_UNSUPPORTED, std::string("librccl.so could not be opened: ") + dlerror());
  auto sym = [&](const char *n) {
    void *p = dlsym(r.so, n);
    if (!p) fail(KSLAM_ERR_UNSUPPORTED, std::string("librccl.so lacks ") + n);
    return p;
  };
  r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
  r.Send = (decltype(r.Send))sym("ncclSend");
  r.Recv = (decltype(r.Recv))sym("ncclRecv");
  r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  return r;
}
void ncchk(int rc, const char *what) {
  if (rc != 0) fail(KSLAM_ERR_INTERNAL, std::string(what) + ": " + rccl().GetErrorString(rc));
}

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  void *ensure(size_t bytes) {
    if (bytes > cap) {
      if (p) (void)hipFree(p);
      p = nullptr;
      cap = 0;
      const size_t want = bytes + bytes / 4 + 256;
      hipchk(hipMalloc(&p, want), "hipMalloc (communicator buffer)");
      cap = want;
    }
    return p;
  }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};

}  // namespace

struct kslam_comm {
  kslam_ctx *ctx = nullptr;
  int rank = 0, world = 1, device = 0;
  ncclComm_t nccl = nullptr;
  hipStream_t stream = nullptr;
  DevBuf counts_mine, counts_all, rows, pool, srows, spool, pad, gathered, all_bytes;
};

extern "C" {

const char *kslam_comm_last_error(void) { return g_err.c_str(); }

kslam_status kslam_comm_unique_id(uint8_t id[KSLAM_COMM_ID_BYTES]) {
  return guarded([&] {
    if (!id) fail(KSLAM_ERR_ARG, "null id");
    ncclUniqueId u;
    ncchk(rccl().GetUniqueId(&u), "ncclGetUniqueId");
    memcpy(id, u.internal, KSLAM_COMM_ID_BYTES);
  });
}

kslam_status kslam_comm_create(kslam_ctx *ctx, const uint8_t id[KSLAM_COMM_ID_BYTES], int rank, int world, kslam_comm **out) {
  if (out) *out = nullptr;
  return guarded([&] {
    if (!ctx || !id || !out) fail(KSLAM_ERR_ARG, "null argument");
    if (world < 1 || rank < 0 || rank >= world) fail(KSLAM_ERR_ARG, "rank outside [0, world)");
    const int device = kslam_ctx_device(ctx);
    if (device < 0) fail(KSLAM_ERR_NO_DEVICE, "the context has no device");
    hipchk(hipSetDevice(device), "hipSetDevice");
    kslam_comm *c = new kslam_comm;
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    c->device = device;
    try {
      hipchk(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate");
      ncclUniqueId u;
      memcpy(u.internal, id, KSLAM_COMM_ID_BYTES);
      ncchk(rccl().CommInitRank(&c->nccl, world, u, rank), "ncclCommInitRank");
    } catch (...) {
      if (c->stream) (void)hipStreamDestroy(c->stream);
      delete c;
      throw;
    }
    *out = c;
  });
}

void kslam_comm_destroy(kslam_comm *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->nccl) (void)rccl().CommDestroy(c->nccl);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int kslam_comm_rank(const kslam_comm *c) { return c ? c->rank : -1; }
int kslam_comm_world(const kslam_comm *c) { return c ? c->world : 0; }

void kslam_comm_gather_plan(const kslam_shard_counts *cnt, int world, uint64_t *row1, uint64_t *row2, uint64_t *op1,
                            uint64_t *op2, uint64_t totals[2]) {
  uint64_t rows_r1 = 0, ops_r1 = 0, rows = 0, ops = 0;
  for (int r = 0; r < world; r++) {
    rows_r1 += cnt[r].n_rows_r1;
    ops_r1 += cnt[r].n_cigar_r1;
    rows += cnt[r].n_rows;
    ops += cnt[r].n_cigar;
  }
  uint64_t a1 = 0, a2 = rows_r1, b1 = 0, b2 = ops_r1;   // all R1 blocks in rank order, then all R2 blocks
  for (int r = 0; r < world; r++) {
    row1[r] = a1;
    row2[r] = a2;
    op1[r] = b1;
    op2[r] = b2;
    a1 += cnt[r].n_rows_r1;
    a2 += cnt[r].n_rows - cnt[r].n_rows_r1;
    b1 += cnt[r].n_cigar_r1;
    b2 += cnt[r].n_cigar - cnt[r].n_cigar_r1;
  }
  totals[0] = rows;
  totals[1] = ops;
}

kslam_status kslam_comm_gather_batch(kslam_comm *c, uint64_t n_local_pairs, uint64_t pair_lo, uint64_t n_pairs_total,
                                     void **d_rows, uint64_t *n_rows, void **d_pool, uint64_t *n_ops) {
  if (d_rows) *d_rows = nullptr;
  if (d_pool) *d_pool = nullptr;
  if (n_rows) *n_rows = 0;
  if (n_ops) *n_ops = 0;
  return guarded([&] {
    if (!c || !d_rows || !n_rows || !d_pool || !n_ops) fail(KSLAM_ERR_ARG, "null argument");
    hipchk(hipSetDevice(c->device), "hipSetDevice");
    Rccl &R = rccl();
    const int W = c->world, me = c->rank;
    // ---- four counts per rank ----
    kslam_shard_counts mine;
    if (kslam_shard_counts_device(c->ctx, n_local_pairs, &mine) != KSLAM_OK) fail(KSLAM_ERR_STATE, kslam_last_error(c->ctx));
    uint64_t *dm = (uint64_t *)c->counts_mine.ensure(sizeof mine);
    uint64_t *da = (uint64_t *)c->counts_all.ensure(sizeof mine * W);
    hipchk(hipMemcpyAsync(dm, &mine, sizeof mine, hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync");
    ncchk(R.AllGather(dm, da, 4, ncclUint64, c->nccl, c->stream), "ncclAllGather (counts)");
    std::vector<kslam_shard_counts> cnt(W);
    hipchk(hipMemcpyAsync(cnt.data(), da, sizeof mine * W, hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync");
    hipchk(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    std::vector<uint64_t> row1(W), row2(W), op1(W), op2(W);
    uint64_t tot[2];
    kslam_comm_gather_plan(cnt.data(), W, row1.data(), row2.data(), op1.data(), op2.data(), tot);
    const size_t RB = sizeof(kslam_overlap), OB = sizeof(uint32_t);
    // ---- every rank writes its records in batch terms; rank 0 straight into the final arrays ----
    char *rows = nullptr, *pool = nullptr, *srows = nullptr, *spool = nullptr;
    const uint64_t n = cnt[me].n_rows, n1 = cnt[me].n_rows_r1, g = cnt[me].n_cigar, g1 = cnt[me].n_cigar_r1;
    if (me == 0) {
      rows = (char *)c->rows.ensure((tot[0] + 1) * RB);
      pool = (char *)c->pool.ensure((tot[1] + 1) * OB);
      if (kslam_export_shard_device(c->ctx, n_local_pairs, pair_lo, n_pairs_total, op1[0], op2[0], rows + RB * row1[0],
                                    rows + RB * row2[0], pool + OB * op1[0], pool + OB * op2[0]) != KSLAM_OK)
        fail(KSLAM_ERR_STATE, kslam_last_error(c->ctx));
    } else {
      srows = (char *)c->srows.ensure((n + 1) * RB);
      spool = (char *)c->spool.ensure((g + 1) * OB);
      if (kslam_export_shard_device(c->ctx, n_local_pairs, pair_lo, n_pairs_total, op1[me], op2[me], srows, srows + RB * n1, spool,
                                    spool + OB * g1) != KSLAM_OK)
        fail(KSLAM_ERR_STATE, kslam_last_error(c->ctx));
    }
    // ---- one group of point-to-point transfers: each peer over its own xGMI link to rank 0 ----
    if (W > 1) {
      ncchk(R.GroupStart(), "ncclGroupStart");
      if (me == 0) {
        for (int r = 1; r < W; r++) {
          const uint64_t m = cnt[r].n_rows, m1 = cnt[r].n_rows_r1, q = cnt[r].n_cigar, q1 = cnt[r].n_cigar_r1;
          if (m1) ncchk(R.Recv(rows + RB * row1[r], m1 * RB, ncclUint8, r, c->nccl, c->stream), "ncclRecv");
          if (m - m1) ncchk(R.Recv(rows + RB * row2[r], (m - m1) * RB, ncclUint8, r, c->nccl, c->stream), "ncclRecv");
          if (q1) ncchk(R.Recv(pool + OB * op1[r], q1 * OB, ncclUint8, r, c->nccl, c->stream), "ncclRecv");
          if (q - q1) ncchk(R.Recv(pool + OB * op2[r], (q - q1) * OB, ncclUint8, r, c->nccl, c->stream), "ncclRecv");
        }
      } else {
        if (n1) ncchk(R.Send(srows, n1 * RB, ncclUint8, 0, c->nccl, c->stream), "ncclSend");
        if (n - n1) ncchk(R.Send(srows + RB * n1, (n - n1) * RB, ncclUint8, 0, c->nccl, c->stream), "ncclSend");
        if (g1) ncchk(R.Send(spool, g1 * OB, ncclUint8, 0, c->nccl, c->stream), "ncclSend");
        if (g - g1) ncchk(R.Send(spool + OB * g1, (g - g1) * OB, ncclUint8, 0, c->nccl, c->stream), "ncclSend");
      }
      ncchk(R.GroupEnd(), "ncclGroupEnd");
    }
    hipchk(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    if (me == 0) {
      *d_rows = rows;
      *d_pool = pool;
      *n_rows = tot[0];
      *n_ops = tot[1];
    }
  });
}

namespace {
// variable-length all-gather of device bytes: counts first, then every piece padded to the longest.  Returns a device
// pointer to the concatenation in rank order (in c->all_bytes) and the per-rank byte counts.
char *all_gather_bytes(kslam_comm *c, const void *d_mine, uint64_t n_mine, std::vector<uint64_t> *counts) {
  Rccl &R = rccl();
  const int W = c->world;
  uint64_t *dm = (uint64_t *)c->counts_mine.ensure(sizeof(uint64_t));
  uint64_t *da = (uint64_t *)c->counts_all.ensure(sizeof(uint64_t) * W);
  hipchk(hipMemcpyAsync(dm, &n_mine, sizeof n_mine, hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync");
  ncchk(R.AllGather(dm, da, 1, ncclUint64, c->nccl, c->stream), "ncclAllGather (counts)");
  counts->assign(W, 0);
  hipchk(hipMemcpyAsync(counts->data(), da, sizeof(uint64_t) * W, hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync");
  hipchk(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
  uint64_t longest = 0, total = 0;
  for (uint64_t v : *counts) {
    longest = std::max(longest, v);
    total += v;
  }
  char *out = (char *)c->all_bytes.ensure(total + 16);
  if (longest == 0) return out;
  if (W == 1) {
    hipchk(hipMemcpyAsync(out, d_mine, n_mine, hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
    hipchk(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    return out;
  }
  char *pad = (char *)c->pad.ensure(longest);
  char *all = (char *)c->gathered.ensure(longest * W);
  if (n_mine) hipchk(hipMemcpyAsync(pad, d_mine, n_mine, hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
  ncchk(R.AllGather(pad, all, longest, ncclUint8, c->nccl, c->stream), "ncclAllGather (pieces)");
  uint64_t at = 0;
  for (int r = 0; r < W; r++) {
    if ((*counts)[r]) hipchk(hipMemcpyAsync(out + at, all + longest * r, (*counts)[r], hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
    at += (*counts)[r];
  }
  hipchk(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
  return out;
}
}  // namespace

kslam_status kslam_comm_sharded_tail(kslam_comm *c, int paired, uint32_t score_threshold, double score_fraction, int pseudo_assembly,
                                     kslam_pair_stats *stats, uint64_t *bytes_received) {
  if (bytes_received) *bytes_received = 0;
  return guarded([&] {
    if (!c || !stats) fail(KSLAM_ERR_ARG, "null argument");
    hipchk(hipSetDevice(c->device), "hipSetDevice");
    const int32_t *d_ins = nullptr;
    uint64_t n_ins = 0, moved = 0;
    if (kslam_pair_phase_a(c->ctx, paired, score_threshold, &d_ins, &n_ins) != KSLAM_OK) fail(KSLAM_ERR_STATE, kslam_last_error(c->ctx));
    std::vector<uint64_t> cnt;
    char *all_ins = all_gather_bytes(c, d_ins, n_ins * 4, &cnt);
    uint64_t ins_bytes = 0;
    for (uint64_t v : cnt) ins_bytes += v;
    moved += ins_bytes;
    const kslam_paired_overlap *d_pairs = nullptr;
    uint64_t n_pairs = 0;
    if (kslam_pair_phase_b(c->ctx, ins_bytes ? (const int32_t *)all_ins : nullptr, ins_bytes / 4, score_fraction, 3, stats, &d_pairs,
                           &n_pairs) != KSLAM_OK)
      fail(KSLAM_ERR_STATE, kslam_last_error(c->ctx));
    if (pseudo_assembly) {
      char *all_recs = all_gather_bytes(c, d_pairs, n_pairs * sizeof(kslam_paired_overlap), &cnt);
      uint64_t rec_bytes = 0, before = 0;
      for (int r = 0; r < c->world; r++) {
        if (r < c->rank) before += cnt[r];
        rec_bytes += cnt[r];
      }
      moved += rec_bytes;
      // KSLAM_ERR_UNSUPPORTED when the device stage declines: the stage is batch-global (src/PairedOverlap.h:480-582), a
      // rank must not fall back to its own pairs
      const kslam_status s = kslam_pseudo_merged(c->ctx, rec_bytes ? all_recs : nullptr, rec_bytes / sizeof(kslam_paired_overlap),
                                                 before / sizeof(kslam_paired_overlap), score_fraction, stats);
      if (s != KSLAM_OK) fail(s, kslam_last_error(c->ctx));
    }
    if (bytes_received) *bytes_received = moved;
  });
}

}  // extern "C"
