// copy_probe.hip -- does a page-locked H2D / D2H copy on one stream slow down a compute kernel on another?
// (rocprofv3 shows the library's copies as __amd_rocclr_copyBuffer KERNELS on the bench boxes: shader copies, not SDMA.)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/copy_probe tools/copy_probe.hip && /tmp/copy_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void spin(float *out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; i++) a = a * b + 1e-6f;
  if (a == 12345.f) out[0] = a;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t N = 600ull << 20;
  void *h, *h2, *d, *d2; float *o;
  CK(hipHostMalloc(&h, N)); CK(hipHostMalloc(&h2, N)); CK(hipMalloc(&d, N)); CK(hipMalloc(&d2, N)); CK(hipMalloc(&o, 64));
  memset(h, 1, N); memset(h2, 2, N);
  hipStream_t sc, sk, sc2; CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc2, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto kernel_ms = [&](int blocks) { CK(hipEventRecord(e0, sk)); hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, sk, o, 200000); CK(hipEventRecord(e1, sk)); };
  for (int blocks : {256 * 8, 256 * 32}) {
    for (int rep = 0; rep < 2; rep++) {
      kernel_ms(blocks); CK(hipStreamSynchronize(sk)); float k0; CK(hipEventElapsedTime(&k0, e0, e1));
      double t = now(); CK(hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, sc)); CK(hipStreamSynchronize(sc)); const double h2d = now() - t;
      t = now(); CK(hipMemcpyAsync(h2, d2, N, hipMemcpyDeviceToHost, sc)); CK(hipStreamSynchronize(sc)); const double d2h = now() - t;
      // both directions at once, with the kernel running
      t = now();
      CK(hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, sc)); CK(hipMemcpyAsync(h2, d2, N, hipMemcpyDeviceToHost, sc2));
      kernel_ms(blocks);
      CK(hipStreamSynchronize(sk)); const double tk = now() - t; CK(hipStreamSynchronize(sc)); CK(hipStreamSynchronize(sc2)); const double tall = now() - t;
      float k1; CK(hipEventElapsedTime(&k1, e0, e1));
      printf("blocks %5d: kernel alone %.2f ms; H2D %.1f GB/s, D2H %.1f GB/s (600 MB each, alone); kernel next to both copies %.2f ms (host saw %.2f), all done after %.2f ms\n",
             blocks, k0, N / h2d / 1e6, N / d2h / 1e6, k1, tk, tall);
    }
  }
  return 0;
}
