// Probe: does hipExtAnyOrderLaunch let a kernel overlap its predecessor ON THE SAME STREAM on gfx950 (ROCm 7)?
// hip_ext.h says the flag is "not supported on AMD GFX9xx boards" for hipExtModuleLaunchKernel; this measures it.
//   hipcc --offload-arch=gfx950 -O2 -o any_order any_order.hip && ./any_order
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void spin(unsigned long long* out, int slot, int us) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[2 * slot] = t0;
    out[2 * slot + 1] = wall_clock64();
  }
}

int main() {
  unsigned long long *d, h[8];
  hipMalloc(&d, sizeof(h));
  hipStream_t s;
  hipStreamCreate(&s);
  for (int variant = 0; variant < 3; ++variant) {
    hipMemset(d, 0, sizeof(h));
    hipDeviceSynchronize();
    // K0: 300 us on one workgroup; K1: 20 us; K2: 20 us
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 0, 300);
    if (variant == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 1, 20);
    else hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, 1, 20);
    if (variant == 2) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, 2, 20);
    else hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 2, 20);
    hipStreamSynchronize(s);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const double u = 0.01;
    printf("variant %d (%s): K0 [0, %.1f] us  K1 [%.1f, %.1f]  K2 [%.1f, %.1f]\n", variant,
           variant == 0 ? "all ordered" : (variant == 1 ? "K1 any-order, K2 ordered" : "K1 and K2 any-order"),
           (h[1] - h[0]) * u, ((long long)(h[2] - h[0])) * u, ((long long)(h[3] - h[0])) * u, ((long long)(h[4] - h[0])) * u,
           ((long long)(h[5] - h[0])) * u);
  }
  return 0;
}
