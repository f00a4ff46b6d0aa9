// d2h_probe.hip -- device-to-host and host-to-device copy rates into page-locked memory of the two kinds the
// library can use (hipHostMalloc; mmap + MADV_HUGEPAGE + hipHostRegister), with the GPU idle and with a
// chip-filling kernel running on another stream.   hipcc --offload-arch=gfx950 -O3 tools/d2h_probe.hip -o /tmp/d2h_probe
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void busy(float *p, int iters) {
  float a = p[threadIdx.x], b = 1.0001f;
  for (int i = 0; i < iters; i++) a = a * b + 0.5f;
  if (a == 12345.f) p[0] = a;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t N = 64u << 20;
  void *d = nullptr, *h1 = nullptr;
  float *dj = nullptr;
  CK(hipMalloc(&d, N)); CK(hipMalloc(&dj, 4096)); CK(hipMemset(d, 1, N)); CK(hipMemset(dj, 0, 4096));
  CK(hipHostMalloc(&h1, N, hipHostMallocDefault));
  void *h2 = mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  madvise(h2, N, MADV_HUGEPAGE);
  memset(h2, 0, N);
  CK(hipHostRegister(h2, N, hipHostRegisterDefault));
  memset(h1, 0, N);
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  for (int load = 0; load < 2; load++)
    for (int kind = 0; kind < 2; kind++)
      for (int dir = 0; dir < 2; dir++) {
        void *h = kind ? h2 : h1;
        double best = 1e9, sum = 0;
        for (int rep = 0; rep < 6; rep++) {
          if (load) hipLaunchKernelGGL(busy, dim3(256 * 16), dim3(256), 0, s2, dj, 4000000);   // ~tens of ms, fills the chip
          const double t0 = now();
          if (dir == 0) CK(hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost, s1));
          else CK(hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s1));
          CK(hipStreamSynchronize(s1));
          const double t = now() - t0;
          CK(hipStreamSynchronize(s2));
          if (rep) { best = t < best ? t : best; sum += t; }
        }
        printf("%-22s %-26s %s: best %.2f ms (%.1f GB/s), mean %.2f ms\n", load ? "under a busy kernel" : "idle GPU",
               kind ? "mmap+hipHostRegister" : "hipHostMalloc", dir ? "H2D" : "D2H", best, N / best / 1e6, sum / 5);
      }
  // the same 64 MB split over several streams (several SDMA engines?): device -> hipHostMalloc memory, idle GPU
  hipStream_t ss[8];
  for (auto &x : ss) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  for (int dir = 0; dir < 2; dir++)
    for (int parts : {1, 2, 4, 8}) {
      double best = 1e9;
      for (int rep = 0; rep < 6; rep++) {
        const double t0 = now();
        const size_t step = N / parts;
        for (int k = 0; k < parts; k++) {
          if (dir == 0) CK(hipMemcpyAsync((char *)h1 + k * step, (char *)d + k * step, step, hipMemcpyDeviceToHost, ss[k]));
          else CK(hipMemcpyAsync((char *)d + k * step, (char *)h1 + k * step, step, hipMemcpyHostToDevice, ss[k]));
        }
        for (int k = 0; k < parts; k++) CK(hipStreamSynchronize(ss[k]));
        const double t = now() - t0;
        if (rep) best = t < best ? t : best;
      }
      printf("%s in %d parts on %d streams: best %.2f ms (%.1f GB/s)\n", dir ? "H2D" : "D2H", parts, parts, best, N / best / 1e6);
    }
  return 0;
}
