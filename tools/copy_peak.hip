// What a hand-written streaming copy sustains on this chip (read + write bytes per second): the
// practical ceiling for a kernel that, like the radix scatter, reads every byte once and writes it
// once.  tools/, not product.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/copy_peak.hip -o /tmp/copy_peak && /tmp/copy_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int U>
__global__ void __launch_bounds__(256) k_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n) {
  // a block owns U * 256 consecutive uint4 per trip; grid-stride over the buffer
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (base + (size_t)u * 256 < n) v[u] = in[base + (size_t)u * 256];
#pragma unroll
    for (int u = 0; u < U; u++) if (base + (size_t)u * 256 < n) out[base + (size_t)u * 256] = v[u];
  }
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ void __launch_bounds__(256) k_copy_nt(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (base + (size_t)u * 256 < n) v[u] = __builtin_nontemporal_load(&in[base + (size_t)u * 256]);
#pragma unroll
    for (int u = 0; u < U; u++) if (base + (size_t)u * 256 < n) __builtin_nontemporal_store(v[u], &out[base + (size_t)u * 256]);
  }
}
__global__ void __launch_bounds__(256) k_read(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n) {
  uint4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const uint4 v = in[i];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_write(uint4 *__restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = uint4{1u, 2u, 3u, (uint32_t)i};
}

template <class F>
static double time_ms(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  const size_t bytes = 3808000000ull;   // one radix pass of the bench: 238 M records x 16 B
  const size_t n = bytes / 16;
  uint4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
    double t;
    t = time_ms([&] { k_copy<1><<<blocks, 256>>>(a, b, n); }); printf("copy U=1 grid %5d: %.3f ms  %.2f TB/s read+write\n", blocks, t, 2.0 * bytes / t / 1e9);
    t = time_ms([&] { k_copy<4><<<blocks, 256>>>(a, b, n); }); printf("copy U=4 grid %5d: %.3f ms  %.2f TB/s\n", blocks, t, 2.0 * bytes / t / 1e9);
    t = time_ms([&] { k_copy<8><<<blocks, 256>>>(a, b, n); }); printf("copy U=8 grid %5d: %.3f ms  %.2f TB/s\n", blocks, t, 2.0 * bytes / t / 1e9);
    t = time_ms([&] { k_copy_nt<4><<<blocks, 256>>>((const u32x4 *)a, (u32x4 *)b, n); }); printf("copy U=4 nontemporal grid %5d: %.3f ms  %.2f TB/s\n", blocks, t, 2.0 * bytes / t / 1e9);
  }
  {
    const int blocks = (int)((n + 255) / 256 / 4);
    double t = time_ms([&] { k_copy<4><<<blocks, 256>>>(a, b, n); }); printf("copy U=4 one trip per block (grid %d): %.3f ms  %.2f TB/s\n", blocks, t, 2.0 * bytes / t / 1e9);
    t = time_ms([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }); printf("hipMemcpyAsync D2D: %.3f ms  %.2f TB/s\n", t, 2.0 * bytes / t / 1e9);
    t = time_ms([&] { k_read<<<256 * 16, 256>>>(a, b, n); }); printf("read only: %.3f ms  %.2f TB/s\n", t, 1.0 * bytes / t / 1e9);
    t = time_ms([&] { k_write<<<256 * 16, 256>>>(b, n); }); printf("write only: %.3f ms  %.2f TB/s\n", t, 1.0 * bytes / t / 1e9);
  }
  return 0;
}
