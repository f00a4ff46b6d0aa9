// How fast can 400 MB of fresh text reach the page cache of ONE new file on this box?  g++ -O2 -pthread tools/pagecache_probe.cpp
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const char *path = argc > 1 ? argv[1] : "/dev/shm/kslam_pc_probe";
  const size_t N = 400u << 20;
  const int T = argc > 2 ? atoi(argv[2]) : 16;
  char *src = (char *)malloc(N);
  memset(src, 'x', N);
  auto par = [&](auto f) { std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back(f, t); for (auto &x : th) x.join(); };
  for (int mode = 0; mode < 12; mode++) {
    unlink(path);
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
    double t0 = now();
    const char *name = "";
    if (mode == 0) { name = "write() 1 thread"; size_t o = 0; while (o < N) o += write(fd, src + o, std::min<size_t>(N - o, 8 << 20)); }
    if (mode == 1) { name = "pwrite() T threads"; par([&](int t) { size_t lo = N / T * t, hi = t == T - 1 ? N : N / T * (t + 1); while (lo < hi) lo += pwrite(fd, src + lo, std::min<size_t>(hi - lo, 8 << 20), lo); }); }
    if (mode == 2 || mode == 3) {
      name = mode == 2 ? "ftruncate + mmap + memcpy T threads" : "same + MADV_HUGEPAGE";
      if (ftruncate(fd, N)) return 1;
      char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      if (mode == 3) printf("  madvise -> %d\n", madvise(m, N, MADV_HUGEPAGE));
      par([&](int t) { size_t lo = N / T * t, hi = t == T - 1 ? N : N / T * (t + 1); memcpy(m + lo, src + lo, hi - lo); });
      munmap(m, N);
    }
    if (mode == 4) { name = "fallocate only"; if (fallocate(fd, 0, 0, N)) perror("fallocate"); }
    if (mode == 5) {
      name = "fallocate, then mmap + memcpy T threads";
      if (fallocate(fd, 0, 0, N)) perror("fallocate");
      double t1 = now();
      char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      par([&](int t) { size_t lo = N / T * t, hi = t == T - 1 ? N : N / T * (t + 1); memcpy(m + lo, src + lo, hi - lo); });
      munmap(m, N);
      printf("  (copy part: %.1f ms)\n", (now() - t1) * 1e3);
    }
    if (mode == 6 || mode == 7) {
      name = mode == 6 ? "fallocate, then write() 1 thread" : "fallocate, then pwrite() T threads";
      if (fallocate(fd, 0, 0, N)) perror("fallocate");
      double t1 = now();
      if (mode == 6) { size_t o = 0; while (o < N) o += write(fd, src + o, std::min<size_t>(N - o, 8 << 20)); }
      else par([&](int t) { size_t lo = N / T * t, hi = t == T - 1 ? N : N / T * (t + 1); while (lo < hi) lo += pwrite(fd, src + lo, std::min<size_t>(hi - lo, 8 << 20), lo); });
      printf("  (write part: %.1f ms)\n", (now() - t1) * 1e3);
    }
    if (mode >= 8 && mode <= 11) {
      // 8: ftruncate + mmap(MAP_POPULATE) + copy; 9: fallocate + the same; 10: ftruncate + mmap + MADV_POPULATE_WRITE per thread + copy;
      // 11: fallocate + the same
      const char *names[4] = {"ftruncate + mmap(MAP_POPULATE) + memcpy T", "fallocate + mmap(MAP_POPULATE) + memcpy T",
                              "ftruncate + mmap + MADV_POPULATE_WRITE/thread", "fallocate + mmap + MADV_POPULATE_WRITE/thread"};
      name = names[mode - 8];
      if (mode & 1) { if (fallocate(fd, 0, 0, N)) perror("fallocate"); } else if (ftruncate(fd, N)) return 1;
      double t1 = now();
      char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED | (mode < 10 ? MAP_POPULATE : 0), fd, 0);
      if (m == MAP_FAILED) { perror("mmap"); return 1; }
      double t2 = now();
      int bad = 0;
      par([&](int t) {
        size_t lo = N / T * t, hi = t == T - 1 ? N : N / T * (t + 1);
        lo &= ~(size_t)4095; if (t != T - 1) hi &= ~(size_t)4095;
#ifdef MADV_POPULATE_WRITE
        if (mode >= 10 && madvise(m + lo, hi - lo, MADV_POPULATE_WRITE)) bad = 1;
#else
        if (mode >= 10) bad = 2;
#endif
        memcpy(m + lo, src + lo, hi - lo);
      });
      munmap(m, N);
      printf("  (mmap %.1f ms, populate+copy %.1f ms, madvise %s)\n", (t2 - t1) * 1e3, (now() - t2) * 1e3, bad == 0 ? "ok" : (bad == 1 ? "FAILED" : "not compiled"));
    }
    double dt = now() - t0;
    printf("%-44s %7.1f ms  %.2f GB/s\n", name, dt * 1e3, N / dt / 1e9);
    // overwrite in place (pages exist): what the same copy costs without allocation
    if (mode == 2) {
      char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      t0 = now();
      par([&](int t) { size_t lo = N / T * t, hi = t == T - 1 ? N : N / T * (t + 1); memcpy(m + lo, src + lo, hi - lo); });
      dt = now() - t0;
      printf("%-44s %7.1f ms  %.2f GB/s\n", "  second copy into the same (existing) pages", dt * 1e3, N / dt / 1e9);
      munmap(m, N);
    }
    close(fd);
  }
  unlink(path);
  return 0;
}
