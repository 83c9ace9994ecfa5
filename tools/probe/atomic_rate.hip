// Microbenchmark: global float atomic-add throughput on gfx950 as a function of active lanes /
// segment alignment, against plain stores.  Build: hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* buf, size_t n, int L, int iters, int rowstride, int misalign, int slots) {
  const int lane = threadIdx.x & 63;
  const size_t wid = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  // each wave walks "rows" of a private window: base + i*rowstride
  const int wp = 64 / slots;  // lanes per slot
  const int slot = lane / wp, col = lane % wp;
  size_t base = (wid * 2654435761ull) % (n - (size_t)iters * rowstride - 64 * 70000ull);
  base = (base & ~31ull) + misalign + (size_t)slot * 67200;  // slot = another channel plane
  const bool on = col < L;
  for (int i = 0; i < iters; ++i) {
    float* p = buf + base + (size_t)i * rowstride + col;
    if (on) {
      if (MODE == 0) unsafeAtomicAdd(p, 1.0f);
      else if (MODE == 1) *p = 1.0f;
      else if (MODE == 2) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
int main() {
  const size_t n = 300ull << 20;  // 1.2 GB of floats? no: 300M floats = 1.2 GB
  float* buf;
  hipMalloc(&buf, n * 4);
  hipMemset(buf, 0, n * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int blocks = 8192, iters = 64;
  struct Cfg { int L, mis, slots; };
  const Cfg cfgs[] = {{64, 0, 1}, {64, 5, 1}, {32, 0, 1}, {20, 0, 1}, {20, 5, 1}, {8, 0, 1}, {20, 5, 2}, {16, 3, 4}, {8, 3, 8}, {1, 0, 1}};
  for (int mode = 0; mode < 3; ++mode)
    for (const Cfg& c : cfgs) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        if (mode == 0) k<0><<<blocks, 256>>>(buf, n, c.L, iters, 336, c.mis, c.slots);
        if (mode == 1) k<1><<<blocks, 256>>>(buf, n, c.L, iters, 336, c.mis, c.slots);
        if (mode == 2) k<2><<<blocks, 256>>>(buf, n, c.L, iters, 336, c.mis, c.slots);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      const double instr = (double)blocks * 4 * iters;
      const double lanes = instr * c.L * c.slots;
      printf("mode %d L=%2d mis=%d slots=%d: %.3f ms  %.1f G lane-ops/s  %.2f G wave-instr/s\n", mode, c.L, c.mis, c.slots, best,
             lanes / best / 1e6, instr / best / 1e6);
    }
  return 0;
}
