// sink_probe.cpp -- how fast can ONE new file take text on this box?  write() from one thread (what kslam_sam_writer does)
// against a shared mapping filled by T threads (page faults run in parallel; write() holds the inode lock for the copy).
//   g++ -O2 -pthread -o /tmp/sink_probe tools/sink_probe.cpp && /tmp/sink_probe /tmp 404 8
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const size_t mb = argc > 2 ? atol(argv[2]) : 404, blocks = 6;
  const int T = argc > 3 ? atoi(argv[3]) : 8;
  const size_t n = mb << 20;
  char *src = (char *)aligned_alloc(4096, n);
  memset(src, 'x', n);
  const std::string path = dir + "/sink_probe.bin";
  {
    unlink(path.c_str());
    int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
    const double t0 = now();
    for (size_t b = 0; b < blocks; b++) {
      size_t done = 0;
      while (done < n) { ssize_t w = write(fd, src + done, n - done); if (w <= 0) { perror("write"); return 1; } done += w; }
    }
    const double t = now() - t0;
    close(fd);
    printf("write(), 1 thread:            %6.2f GB/s (%zu x %zu MB)\n", blocks * n / t / 1e9, blocks, mb);
  }
  for (int threads : {1, 2, 4, T}) {
    for (int populate = 0; populate < 2; populate++) {
      unlink(path.c_str());
      int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0644);
      const double t0 = now();
      for (size_t b = 0; b < blocks; b++) {
        if (ftruncate(fd, (b + 1) * n)) { perror("ftruncate"); return 1; }
        char *m = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, b * n);
        if (m == MAP_FAILED) { perror("mmap"); return 1; }
        std::vector<std::thread> th;
        for (int k = 0; k < threads; k++)
          th.emplace_back([&, k] {
            const size_t lo = (n / threads * k) & ~(size_t)4095, hi = k == threads - 1 ? n : (n / threads * (k + 1)) & ~(size_t)4095;
#ifdef MADV_POPULATE_WRITE
            if (populate) madvise(m + lo, hi - lo, MADV_POPULATE_WRITE);
#endif
            memcpy(m + lo, src + lo, hi - lo);
          });
        for (auto &x : th) x.join();
        munmap(m, n);
      }
      const double t = now() - t0;
      close(fd);
      printf("mmap + memcpy, %d thread(s)%s: %6.2f GB/s\n", threads, populate ? ", MADV_POPULATE_WRITE" : "                     ", blocks * n / t / 1e9);
    }
  }
  // pwrite from T threads into disjoint ranges of one file (serialised by the inode lock?)
  {
    unlink(path.c_str());
    int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
    const double t0 = now();
    for (size_t b = 0; b < blocks; b++) {
      std::vector<std::thread> th;
      for (int k = 0; k < T; k++)
        th.emplace_back([&, k] {
          const size_t lo = n / T * k, hi = k == T - 1 ? n : n / T * (k + 1);
          size_t done = lo;
          while (done < hi) { ssize_t w = pwrite(fd, src + done, hi - done, b * n + done); if (w <= 0) break; done += w; }
        });
      for (auto &x : th) x.join();
    }
    const double t = now() - t0;
    close(fd);
    printf("pwrite(), %d threads:          %6.2f GB/s\n", T, blocks * n / t / 1e9);
  }
  unlink(path.c_str());
  return 0;
}
