// fake_rccl.cpp -- TEST INFRASTRUCTURE, not part of the product: a stand-in for the dozen RCCL entry points that
// k-slam_amd/host/comm.cpp resolves with dlsym, so that the library's own communicator code (include/kslam_comm.h) can run
// at world sizes 2 / 4 / 8 on a box with ONE GPU, where RCCL refuses two ranks on one device.  Selected by the tests with
// KSLAM_RCCL_LIB=<this .so>; kslam_comm_info reports the path, so a record made with it says so.
//
// Transport: files under /dev/shm/kslam_fake_rccl_<id>/ -- a send copies the device bytes to the host and publishes them
// as m_<src>_<dst>_<seq> (written under a temporary name, renamed when complete); a receive polls for its file, copies it
// to the device and unlinks it.  Works between threads of one process and between processes.  Every operation first waits
// for the stream it was given and completes before it returns, which is a legal (if slow) execution of the stream order
// RCCL promises.  Nothing here says anything about RCCL's performance or about xGMI; it exercises the CALLER's protocol:
// counts, offsets, piece order, group composition, status exchange.
#include <dirent.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {
enum { OK = 0, ERR_SYSTEM = 2, ERR_INTERNAL = 3, ERR_ARG = 4 };
struct Id { char internal[128]; };
struct Comm {
  std::string dir;
  int rank = 0, world = 1;
  std::vector<uint64_t> sent, received;   // per peer
};
struct Op {
  bool send;
  void *p;
  size_t bytes;
  int peer;
  Comm *c;
  hipStream_t s;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

size_t width(int dtype) {   // ncclDataType_t: int8 0, uint8 1, int32 2, uint32 3, int64 4, uint64 5, half 6, float 7, double 8
  static const size_t w[] = {1, 1, 4, 4, 8, 8, 2, 4, 8};
  return dtype >= 0 && dtype <= 8 ? w[dtype] : 1;
}
double timeout_s() {
  const char *e = getenv("KSLAM_FAKE_RCCL_TIMEOUT");
  return e ? atof(e) : 120.0;
}
std::string name(const Comm *c, int src, int dst, uint64_t seq) {
  char b[96];
  snprintf(b, sizeof b, "/m_%d_%d_%llu", src, dst, (unsigned long long)seq);
  return c->dir + b;
}
int do_send(const Op &o) {
  Comm *c = o.c;
  std::vector<char> host(o.bytes);
  if (o.bytes && hipMemcpy(host.data(), o.p, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ERR_SYSTEM;
  const std::string fin = name(c, c->rank, o.peer, c->sent[o.peer]++), tmp = fin + ".part";
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
  if (fd < 0) return ERR_SYSTEM;
  size_t at = 0;
  while (at < o.bytes) {
    const ssize_t k = write(fd, host.data() + at, o.bytes - at);
    if (k <= 0) { close(fd); return ERR_SYSTEM; }
    at += (size_t)k;
  }
  close(fd);
  return rename(tmp.c_str(), fin.c_str()) == 0 ? OK : ERR_SYSTEM;
}
int do_recv(const Op &o) {
  Comm *c = o.c;
  const std::string fin = name(c, o.peer, c->rank, c->received[o.peer]++);
  const auto t0 = std::chrono::steady_clock::now();
  int fd;
  while ((fd = open(fin.c_str(), O_RDONLY)) < 0) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) {
      fprintf(stderr, "[fake_rccl] rank %d: no message %s after %.0f s\n", c->rank, fin.c_str(), timeout_s());
      return ERR_INTERNAL;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  struct stat st;
  if (fstat(fd, &st) != 0 || (size_t)st.st_size != o.bytes) {
    fprintf(stderr, "[fake_rccl] rank %d: %s holds %lld bytes, the receive wants %zu\n", c->rank, fin.c_str(), (long long)st.st_size, o.bytes);
    close(fd);
    return ERR_ARG;
  }
  std::vector<char> host(o.bytes);
  size_t at = 0;
  while (at < o.bytes) {
    const ssize_t k = read(fd, host.data() + at, o.bytes - at);
    if (k <= 0) { close(fd); return ERR_SYSTEM; }
    at += (size_t)k;
  }
  close(fd);
  unlink(fin.c_str());
  if (o.bytes && hipMemcpy(o.p, host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ERR_SYSTEM;
  return OK;
}
int run(std::vector<Op> &ops) {   // sends are buffered (files), so all sends first can never wait for a peer
  for (const Op &o : ops)
    if (hipStreamSynchronize(o.s) != hipSuccess) return ERR_SYSTEM;
  for (const Op &o : ops)
    if (o.send) { const int r = do_send(o); if (r) return r; }
  for (const Op &o : ops)
    if (!o.send) { const int r = do_recv(o); if (r) return r; }
  return OK;
}
int post(Op o) {
  if (!o.c || o.peer < 0 || o.peer >= o.c->world) return ERR_ARG;
  if (g_depth) { g_ops.push_back(o); return OK; }
  std::vector<Op> one{o};
  return run(one);
}
}  // namespace

extern "C" {
int ncclGetVersion(int *v) { if (!v) return ERR_ARG; *v = 0; return OK; }   // 0: no RCCL release calls itself that
const char *ncclGetErrorString(int r) {
  switch (r) {
    case OK: return "no error";
    case ERR_SYSTEM: return "fake_rccl: system / HIP call failed";
    case ERR_INTERNAL: return "fake_rccl: a message never arrived";
    case ERR_ARG: return "fake_rccl: invalid argument or size mismatch between a send and its receive";
    default: return "fake_rccl: unknown";
  }
}
int ncclGetUniqueId(Id *id) {
  if (!id) return ERR_ARG;
  memset(id->internal, 0, sizeof id->internal);
  std::random_device rd;
  snprintf(id->internal, sizeof id->internal, "%08x%08x%08x_%d", rd(), rd(), rd(), (int)getpid());
  return OK;
}
int ncclCommInitRank(Comm **out, int world, Id id, int rank) {
  if (!out || world < 1 || rank < 0 || rank >= world) return ERR_ARG;
  id.internal[127] = 0;
  Comm *c = new Comm;
  c->dir = std::string("/dev/shm/kslam_fake_rccl_") + id.internal;
  c->rank = rank;
  c->world = world;
  c->sent.assign(world, 0);
  c->received.assign(world, 0);
  mkdir(c->dir.c_str(), 0700);
  const std::string hello = c->dir + "/hello_" + std::to_string(rank);
  const int fd = open(hello.c_str(), O_WRONLY | O_CREAT, 0600);
  if (fd < 0) { delete c; return ERR_SYSTEM; }
  close(fd);
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {   // every rank has arrived
    int seen = 0;
    for (int r = 0; r < world; r++) seen += access((c->dir + "/hello_" + std::to_string(r)).c_str(), F_OK) == 0;
    if (seen == world) break;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) { delete c; return ERR_INTERNAL; }
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  *out = c;
  return OK;
}
static int leave(Comm *c) {
  if (!c) return OK;
  // a rank leaves a marker; the last one to leave removes the directory (files of a failed run included)
  const int fd = open((c->dir + "/bye_" + std::to_string(c->rank)).c_str(), O_WRONLY | O_CREAT, 0600);
  if (fd >= 0) close(fd);
  int gone = 0;
  for (int r = 0; r < c->world; r++) gone += access((c->dir + "/bye_" + std::to_string(r)).c_str(), F_OK) == 0;
  if (gone == c->world) {
    if (DIR *d = opendir(c->dir.c_str())) {
      while (dirent *e = readdir(d))
        if (e->d_name[0] != '.') unlink((c->dir + "/" + e->d_name).c_str());
      closedir(d);
    }
    rmdir(c->dir.c_str());
  }
  delete c;
  return OK;
}
int ncclCommDestroy(Comm *c) { return leave(c); }
int ncclCommAbort(Comm *c) { return leave(c); }
int ncclCommCount(const Comm *c, int *n) { if (!c || !n) return ERR_ARG; *n = c->world; return OK; }
int ncclCommUserRank(const Comm *c, int *r) { if (!c || !r) return ERR_ARG; *r = c->rank; return OK; }
int ncclGroupStart() { g_depth++; return OK; }
int ncclGroupEnd() {
  if (g_depth <= 0) return ERR_ARG;
  if (--g_depth) return OK;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run(ops);
}
int ncclSend(const void *p, size_t count, int dtype, int peer, Comm *c, hipStream_t s) {
  return post(Op{true, const_cast<void *>(p), count * width(dtype), peer, c, s});
}
int ncclRecv(void *p, size_t count, int dtype, int peer, Comm *c, hipStream_t s) {
  return post(Op{false, p, count * width(dtype), peer, c, s});
}
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, Comm *c, hipStream_t s) {
  if (!c) return ERR_ARG;
  const size_t bytes = count * width(dtype);
  std::vector<Op> ops;
  for (int r = 0; r < c->world; r++)
    if (r != c->rank) ops.push_back(Op{true, const_cast<void *>(send), bytes, r, c, s});
  for (int r = 0; r < c->world; r++)
    if (r != c->rank) ops.push_back(Op{false, (char *)recv + bytes * r, bytes, r, c, s});
  if (hipStreamSynchronize(s) != hipSuccess) return ERR_SYSTEM;
  if (bytes && hipMemcpy((char *)recv + bytes * c->rank, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ERR_SYSTEM;
  return run(ops);
}
}  // extern "C"
