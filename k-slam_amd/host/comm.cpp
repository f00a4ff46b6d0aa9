// comm.cpp -- include/kslam_comm.h: the end-of-batch exchanges of a read-sharded batch over RCCL, one process per GPU,
// on top of the library's own C ABI (kslam_shard_counts_device / kslam_export_shard_device / kslam_pair_phase_a, _b /
// kslam_pseudo_route, _owned, _return) and the HIP runtime.  librccl.so is opened with dlopen on first use.
//
// The protocol is the one k-slam_amd/dist.py runs through torch.distributed: counts by
// ncclAllGather, then grouped ncclSend / ncclRecv whose pieces land in their final places on rank 0; the variable-length
// all-gather of the insert sizes pads every rank's piece to the longest; pseudo-assembly's two all-to-alls are one group of
// ncclSend / ncclRecv each (every pair of GPUs has its own xGMI link).
//
// Failing together: a step that can fail on ONE rank (an export, an allocation, a device stage that declines) is followed
// by an exchange of status words before any rank enters a transfer that depends on it -- the words ride on the count
// exchanges that exist anyway, or are one 8-byte all-gather -- so that every rank returns the failure and none waits in a
// collective for a peer that has left.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstring>
#include <exception>
#include <mutex>
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
  } catch (const std::exception &e) {   // nothing may leave through extern "C"
    g_err = std::string("unexpected exception: ") + e.what();
    return KSLAM_ERR_INTERNAL;
  } catch (...) {
    g_err = "unexpected exception";
    return KSLAM_ERR_INTERNAL;
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
  int (*CommAbort)(ncclComm_t) = nullptr;
  int (*CommCount)(const ncclComm_t, int *) = nullptr;
  int (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  int (*GetVersion)(int *) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
// Resolved into a local table and published only when complete, once, under a lock: a failed attempt leaves nothing
// half-filled behind (the next call tries again and fails the same way), and two threads cannot race on the first call.
Rccl &rccl() {
  static Rccl table;
  static bool ready = false;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (ready) return table;
  Rccl r;
  const char *names[] = {getenv("KSLAM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  std::string tried;
  for (const char *n : names) {
    if (!n || !*n) continue;
    r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.so) break;
    const char *e = dlerror();
    tried += std::string(tried.empty() ? "" : "; ") + (e ? e : n);
  }
  if (!r.so) fail(KSLAM_ERR_UNSUPPORTED, "librccl.so could not be opened: " + tried);
  auto sym = [&](const char *n) {
    void *p = dlsym(r.so, n);
    if (!p) {
      dlclose(r.so);
      fail(KSLAM_ERR_UNSUPPORTED, std::string("librccl.so lacks ") + n);
    }
    return p;
  };
  r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
  r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
  r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
  r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
  r.Send = (decltype(r.Send))sym("ncclSend");
  r.Recv = (decltype(r.Recv))sym("ncclRecv");
  r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
  table = r;
  ready = true;
  return table;
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

// what one rank knows about its own last local step
struct Status {
  kslam_status st = KSLAM_OK;
  std::string msg;
  void set(kslam_status s, const std::string &m) {
    st = s == KSLAM_OK ? KSLAM_ERR_INTERNAL : s;
    msg = m;
  }
};
template <class F>
Status attempt(F &&f) {
  Status s;
  try {
    f();
  } catch (const Fail &e) {
    s.set(e.st, e.msg);
  } catch (const std::bad_alloc &) {
    s.set(KSLAM_ERR_OOM, "out of host memory");
  }
  return s;
}

}  // namespace

struct kslam_comm {
  kslam_ctx *ctx = nullptr;
  int rank = 0, world = 1, device = 0;
  ncclComm_t nccl = nullptr;
  hipStream_t stream = nullptr;
  bool dead = false;   // a transfer failed half-way: the communicator was aborted (ncclCommAbort), only destroy is left
  DevBuf counts_mine, counts_all, rows[2], pool[2], srows, spool, pad, gathered, all_bytes, a2a_in, a2a_back;
  int flip = 0;
  bool gather_pending = false;   // between kslam_comm_gather_begin and _end
  char *got_rows = nullptr, *got_pool = nullptr;
  uint64_t got_n_rows = 0, got_n_ops = 0;
};

namespace {
// A transfer (collective or group of point-to-point operations) that fails half-way leaves the peers inside it: the
// communicator is aborted so that they come out with an error instead of waiting, and is dead afterwards.
template <class F>
void transfer(kslam_comm *c, F &&f) {
  try {
    f();
  } catch (...) {
    if (c->nccl) (void)rccl().CommAbort(c->nccl);
    c->nccl = nullptr;
    c->dead = true;
    throw;
  }
}

// k 64-bit words of every rank, in rank order
void all_gather_words(kslam_comm *c, const uint64_t *mine, int k, std::vector<uint64_t> *all) {
  Rccl &R = rccl();
  const int W = c->world;
  uint64_t *dm = nullptr, *da = nullptr;
  transfer(c, [&] {   // (an allocation of a few bytes failing here is as fatal to the peers as a failing collective)
    dm = (uint64_t *)c->counts_mine.ensure(sizeof(uint64_t) * k);
    da = (uint64_t *)c->counts_all.ensure(sizeof(uint64_t) * k * W);
    all->assign((size_t)k * W, 0);
    hipchk(hipMemcpyAsync(dm, mine, sizeof(uint64_t) * k, hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync");
    ncchk(R.AllGather(dm, da, (size_t)k, ncclUint64, c->nccl, c->stream), "ncclAllGather (counts)");
    hipchk(hipMemcpyAsync(all->data(), da, sizeof(uint64_t) * k * W, hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync");
    hipchk(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
  });
}
// after an exchange whose word `at` (of `stride` per rank) is each rank's status: every rank fails if any did
void together(kslam_comm *c, const std::vector<uint64_t> &all, int stride, int at, const Status &local, const char *what) {
  if (local.st != KSLAM_OK) fail(local.st, std::string(what) + " on this rank (" + std::to_string(c->rank) + "): " + local.msg);
  for (int r = 0; r < c->world; r++) {
    const kslam_status st = (kslam_status)all[(size_t)r * stride + at];
    if (st != KSLAM_OK) fail(st, std::string(what) + " failed on rank " + std::to_string(r) + " (status " + std::to_string((int)st) + "): every rank gives the batch up");
  }
}
// one 8-byte all-gather for a local step that has no count exchange to ride on
void agree(kslam_comm *c, const Status &local, const char *what) {
  const uint64_t w = (uint64_t)local.st;
  std::vector<uint64_t> all;
  all_gather_words(c, &w, 1, &all);
  together(c, all, 1, 0, local, what);
}
}  // namespace

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
  if (c->stream && !c->dead) (void)hipStreamSynchronize(c->stream);   // a gather begun and never ended: its buffers go away below
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

kslam_status kslam_comm_gather_begin(kslam_comm *c, uint64_t n_local_pairs, uint64_t pair_lo, uint64_t n_pairs_total) {
  return guarded([&] {
    if (!c) fail(KSLAM_ERR_ARG, "null argument");
    hipchk(hipSetDevice(c->device), "hipSetDevice");
    Rccl &R = rccl();
    const int W = c->world, me = c->rank;
    if (c->dead) fail(KSLAM_ERR_STATE, "this communicator was aborted by an earlier failure");
    if (c->gather_pending) fail(KSLAM_ERR_STATE, "a gather is in flight: kslam_comm_gather_end first");
    c->flip ^= 1;   // rank 0 receives into the other pair of arrays: what the last gather returned stays valid meanwhile
    DevBuf &rows_buf = c->rows[c->flip], &pool_buf = c->pool[c->flip];
    // ---- four counts per rank, and whether the rank could count at all ----
    kslam_shard_counts mine;
    memset(&mine, 0, sizeof mine);
    Status local;
    if (kslam_shard_counts_device(c->ctx, n_local_pairs, &mine) != KSLAM_OK) local.set(KSLAM_ERR_STATE, kslam_last_error(c->ctx));
    uint64_t words[5] = {mine.n_rows, mine.n_rows_r1, mine.n_cigar, mine.n_cigar_r1, (uint64_t)local.st};
    std::vector<uint64_t> all;
    all_gather_words(c, words, 5, &all);
    together(c, all, 5, 4, local, "kslam_shard_counts_device");
    std::vector<kslam_shard_counts> cnt(W);
    for (int r = 0; r < W; r++) cnt[r] = kslam_shard_counts{all[5 * r], all[5 * r + 1], all[5 * r + 2], all[5 * r + 3]};
    std::vector<uint64_t> row1(W), row2(W), op1(W), op2(W);
    uint64_t tot[2];
    kslam_comm_gather_plan(cnt.data(), W, row1.data(), row2.data(), op1.data(), op2.data(), tot);
    const size_t RB = sizeof(kslam_overlap), OB = sizeof(uint32_t);
    // ---- every rank writes its records in batch terms; rank 0 straight into the final arrays ----
    char *rows = nullptr, *pool = nullptr, *srows = nullptr, *spool = nullptr;
    const uint64_t n = cnt[me].n_rows, n1 = cnt[me].n_rows_r1, g = cnt[me].n_cigar, g1 = cnt[me].n_cigar_r1;
    local = attempt([&] {
      if (me == 0) {
        rows = (char *)rows_buf.ensure((tot[0] + 1) * RB);
        pool = (char *)pool_buf.ensure((tot[1] + 1) * OB);
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
    });
    agree(c, local, "kslam_export_shard_device");
    // ---- one group of point-to-point transfers: each peer over its own xGMI link to rank 0 ----
    transfer(c, [&] {
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
    });
    // the transfers are in flight on the communicator's stream; the sending side's copies (srows / spool) and rank 0's
    // arrays belong to them until kslam_comm_gather_end
    c->gather_pending = true;
    c->got_rows = me == 0 ? rows : nullptr;
    c->got_pool = me == 0 ? pool : nullptr;
    c->got_n_rows = me == 0 ? tot[0] : 0;
    c->got_n_ops = me == 0 ? tot[1] : 0;
  });
}

kslam_status kslam_comm_gather_end(kslam_comm *c, void **d_rows, uint64_t *n_rows, void **d_pool, uint64_t *n_ops) {
  if (d_rows) *d_rows = nullptr;
  if (d_pool) *d_pool = nullptr;
  if (n_rows) *n_rows = 0;
  if (n_ops) *n_ops = 0;
  return guarded([&] {
    if (!c || !d_rows || !n_rows || !d_pool || !n_ops) fail(KSLAM_ERR_ARG, "null argument");
    if (c->dead) fail(KSLAM_ERR_STATE, "this communicator was aborted by an earlier failure");
    if (!c->gather_pending) fail(KSLAM_ERR_STATE, "no gather in flight");
    hipchk(hipSetDevice(c->device), "hipSetDevice");
    c->gather_pending = false;
    transfer(c, [&] { hipchk(hipStreamSynchronize(c->stream), "hipStreamSynchronize"); });
    *d_rows = c->got_rows;
    *d_pool = c->got_pool;
    *n_rows = c->got_n_rows;
    *n_ops = c->got_n_ops;
  });
}

kslam_status kslam_comm_gather_batch(kslam_comm *c, uint64_t n_local_pairs, uint64_t pair_lo, uint64_t n_pairs_total,
                                     void **d_rows, uint64_t *n_rows, void **d_pool, uint64_t *n_ops) {
  if (d_rows) *d_rows = nullptr;
  if (d_pool) *d_pool = nullptr;
  if (n_rows) *n_rows = 0;
  if (n_ops) *n_ops = 0;
  if (!c || !d_rows || !n_rows || !d_pool || !n_ops) {
    g_err = "null argument";
    return KSLAM_ERR_ARG;
  }
  const kslam_status s = kslam_comm_gather_begin(c, n_local_pairs, pair_lo, n_pairs_total);
  return s != KSLAM_OK ? s : kslam_comm_gather_end(c, d_rows, n_rows, d_pool, n_ops);
}

namespace {
// variable-length all-gather of device bytes: counts first (with this rank's status word beside its count), then every piece
// padded to the longest.  Returns a device pointer to the concatenation in rank order (in c->all_bytes) and the per-rank
// byte counts.
char *all_gather_bytes(kslam_comm *c, const void *d_mine, uint64_t n_mine, std::vector<uint64_t> *counts, const Status &local,
                       const char *what) {
  Rccl &R = rccl();
  const int W = c->world;
  const uint64_t words[2] = {local.st == KSLAM_OK ? n_mine : 0, (uint64_t)local.st};
  std::vector<uint64_t> all;
  all_gather_words(c, words, 2, &all);
  together(c, all, 2, 1, local, what);
  counts->assign(W, 0);
  uint64_t longest = 0, total = 0;
  for (int r = 0; r < W; r++) {
    (*counts)[r] = all[2 * r];
    longest = std::max(longest, all[2 * r]);
    total += all[2 * r];
  }
  char *out = nullptr, *pad = nullptr, *gathered = nullptr;
  agree(c, attempt([&] {
          out = (char *)c->all_bytes.ensure(total + 16);
          if (W > 1 && longest) {
            pad = (char *)c->pad.ensure(longest);
            gathered = (char *)c->gathered.ensure(longest * W);
          }
        }),
        "communicator buffers");
  if (longest == 0) return out;
  if (W == 1) {
    hipchk(hipMemcpyAsync(out, d_mine, n_mine, hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
    hipchk(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    return out;
  }
  transfer(c, [&] {
    if (n_mine) hipchk(hipMemcpyAsync(pad, d_mine, n_mine, hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
    ncchk(R.AllGather(pad, gathered, longest, ncclUint8, c->nccl, c->stream), "ncclAllGather (pieces)");
    uint64_t at = 0;
    for (int r = 0; r < W; r++) {
      if ((*counts)[r])
        hipchk(hipMemcpyAsync(out + at, gathered + longest * r, (*counts)[r], hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
      at += (*counts)[r];
    }
    hipchk(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
  });
  return out;
}

// all-to-all of device bytes: the piece for rank r is send[send_off[r] .. + send_n[r]), what rank r sends here lands at
// recv[recv_off[r] .. + recv_n[r]).  One group of ncclSend / ncclRecv (every pair of GPUs has its own xGMI link); this rank's
// own piece is a device copy.
void all_to_all(kslam_comm *c, const char *send, const std::vector<uint64_t> &send_off, const std::vector<uint64_t> &send_n, char *recv,
                const std::vector<uint64_t> &recv_off, const std::vector<uint64_t> &recv_n) {
  Rccl &R = rccl();
  const int W = c->world, me = c->rank;
  transfer(c, [&] {
    if (send_n[me] != recv_n[me]) fail(KSLAM_ERR_INTERNAL, "all_to_all: this rank's own piece has two sizes");
    if (send_n[me])
      hipchk(hipMemcpyAsync(recv + recv_off[me], send + send_off[me], send_n[me], hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
    if (W > 1) {
      ncchk(R.GroupStart(), "ncclGroupStart");
      for (int r = 0; r < W; r++) {
        if (r == me) continue;
        if (send_n[r]) ncchk(R.Send(send + send_off[r], send_n[r], ncclUint8, r, c->nccl, c->stream), "ncclSend");
        if (recv_n[r]) ncchk(R.Recv(recv + recv_off[r], recv_n[r], ncclUint8, r, c->nccl, c->stream), "ncclRecv");
      }
      ncchk(R.GroupEnd(), "ncclGroupEnd");
    }
    hipchk(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
  });
}
}  // namespace

kslam_status kslam_comm_sharded_tail(kslam_comm *c, int paired, uint32_t score_threshold, double score_fraction, int pseudo_assembly,
                                     kslam_pair_stats *stats, uint64_t *bytes_received) {
  if (bytes_received) *bytes_received = 0;
  return guarded([&] {
    if (!c || !stats) fail(KSLAM_ERR_ARG, "null argument");
    if (c->dead) fail(KSLAM_ERR_STATE, "this communicator was aborted by an earlier failure");
    if (c->gather_pending) fail(KSLAM_ERR_STATE, "a gather is in flight: kslam_comm_gather_end first");
    hipchk(hipSetDevice(c->device), "hipSetDevice");
    const int W = c->world, me = c->rank;
    // ---- pairing on this rank's rows; the insert sizes of every rank (getMaxAllowedInsertSize is a statistic of the batch) ----
    const int32_t *d_ins = nullptr;
    uint64_t n_ins = 0, moved = 0;
    Status local;
    if (kslam_pair_phase_a(c->ctx, paired, score_threshold, &d_ins, &n_ins) != KSLAM_OK) local.set(KSLAM_ERR_STATE, kslam_last_error(c->ctx));
    std::vector<uint64_t> cnt;
    char *all_ins = all_gather_bytes(c, d_ins, n_ins * 4, &cnt, local, "kslam_pair_phase_a");
    uint64_t ins_bytes = 0;
    for (uint64_t v : cnt) ins_bytes += v;
    moved += ins_bytes - cnt[me];
    const kslam_paired_overlap *d_pairs = nullptr;
    uint64_t n_pairs = 0;
    local = Status();
    if (kslam_pair_phase_b(c->ctx, ins_bytes ? (const int32_t *)all_ins : nullptr, ins_bytes / 4, score_fraction, 3, stats, &d_pairs,
                           &n_pairs) != KSLAM_OK)
      local.set(KSLAM_ERR_STATE, kslam_last_error(c->ctx));
    if (!pseudo_assembly) {
      agree(c, local, "kslam_pair_phase_b");
      if (bytes_received) *bytes_received = moved;
      return;
    }
    // ---- pseudoAssembly, entry e on rank e mod W (src/PairedOverlap.h:480-582; include/kslam.h: kslam_pseudo_route) ----
    const void *d_heads = nullptr;
    std::vector<uint64_t> to(W, 0);
    if (local.st == KSLAM_OK && kslam_pseudo_route(c->ctx, (uint32_t)W, &d_heads, to.data()) != KSLAM_OK)
      local.set(KSLAM_ERR_STATE, kslam_last_error(c->ctx));
    std::vector<uint64_t> words(W + 1, 0), matrix;
    for (int r = 0; r < W; r++) words[r] = local.st == KSLAM_OK ? to[r] : 0;
    words[W] = (uint64_t)local.st;
    all_gather_words(c, words.data(), W + 1, &matrix);
    together(c, matrix, W + 1, W, local, "kslam_pair_phase_b / kslam_pseudo_route");
    const uint64_t HB = 16, SB = 4;   // a head, a score
    std::vector<uint64_t> send_off(W), send_n(W), recv_off(W), recv_n(W);
    uint64_t n_own = 0, n_recv = 0;
    for (int r = 0; r < W; r++) {
      send_off[r] = n_own;
      send_n[r] = to[r];
      n_own += to[r];
      recv_off[r] = n_recv;
      recv_n[r] = matrix[(size_t)r * (W + 1) + me];   // what rank r holds for the entries that are mine
      n_recv += recv_n[r];
    }
    char *in = nullptr, *back = nullptr;
    agree(c, attempt([&] {
            in = (char *)c->a2a_in.ensure(n_recv * HB + 16);
            back = (char *)c->a2a_back.ensure(n_own * SB + 16);
          }),
          "communicator buffers");
    auto scaled = [](std::vector<uint64_t> v, uint64_t k) {
      for (auto &x : v) x *= k;
      return v;
    };
    all_to_all(c, (const char *)d_heads, scaled(send_off, HB), scaled(send_n, HB), in, scaled(recv_off, HB), scaled(recv_n, HB));
    moved += (n_recv - recv_n[me]) * HB;
    const uint32_t *d_scores = nullptr;
    local = Status();
    {
      const kslam_status s = kslam_pseudo_owned(c->ctx, n_recv ? in : nullptr, n_recv, &d_scores);
      // KSLAM_ERR_UNSUPPORTED when the device stage declines: the stage is batch-global, no rank may go on with scores of its own
      if (s != KSLAM_OK) local.set(s, kslam_last_error(c->ctx));
    }
    agree(c, local, "kslam_pseudo_owned");
    all_to_all(c, (const char *)d_scores, scaled(recv_off, SB), scaled(recv_n, SB), back, scaled(send_off, SB), scaled(send_n, SB));
    moved += (n_own - send_n[me]) * SB;
    // the last step is agreed like the others: a rank that cannot commit the scores must not leave its peers with a batch
    // whose pseudo-assembly only they hold
    local = Status();
    if (kslam_pseudo_return(c->ctx, n_own ? (const uint32_t *)back : nullptr, n_own, score_fraction, stats) != KSLAM_OK)
      local.set(KSLAM_ERR_STATE, kslam_last_error(c->ctx));
    agree(c, local, "kslam_pseudo_return");
    if (bytes_received) *bytes_received = moved;
  });
}

kslam_status kslam_comm_info(const kslam_comm *c, kslam_comm_facts *out) {
  return guarded([&] {
    if (!c || !out) fail(KSLAM_ERR_ARG, "null argument");
    memset(out, 0, sizeof *out);
    if (c->dead || !c->nccl) fail(KSLAM_ERR_STATE, "this communicator was aborted by an earlier failure");
    Rccl &R = rccl();
    int v = 0, n = 0, r = -1;
    ncchk(R.GetVersion(&v), "ncclGetVersion");
    ncchk(R.CommCount(c->nccl, &n), "ncclCommCount");
    ncchk(R.CommUserRank(c->nccl, &r), "ncclCommUserRank");
    out->rccl_version = v;
    out->comm_count = n;
    out->comm_rank = r;
    out->device = c->device;
    Dl_info di;
    if (dladdr((void *)R.AllGather, &di) && di.dli_fname) {
      strncpy(out->library, di.dli_fname, sizeof out->library - 1);
    }
  });
}

}  // extern "C"
