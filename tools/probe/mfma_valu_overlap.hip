// Do VALU instructions of another wave run under the MFMAs of a SIMD's matrix pipe?  12 waves per CU: 8 issue
// v_mfma_f32_32x32x2_f32 back to back (2 per SIMD), 4 (one per SIMD) issue dependent-free v_fma_f32 (mode 1), LDS reads (mode 2)
// or nothing (mode 0).  Reports the MFMA rate and the time of the VALU waves alone / together.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu_overlap.hip -o tools/probe/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned long long g_t[8];

template <int MODE, bool MFMA_ON>
__global__ __launch_bounds__(768) void k(float* sink, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 768) lds[i] = (float)(i & 255) * 0.001f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long t0 = __builtin_readcyclecounter();
  float r = 0.f;
  if (wave < 8) {
    if (MFMA_ON) {
      f32x16 acc[2];
      for (int m = 0; m < 2; ++m)
        for (int q = 0; q < 16; ++q) acc[m][q] = 0.f;
      const float a = 1.f + lane * 0.001f, b = 0.5f;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
      for (int m = 0; m < 2; ++m)
        for (int q = 0; q < 16; ++q) r += acc[m][q];
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) g_t[0] = __builtin_readcyclecounter() - t0;
  } else {
    if (MODE == 1) {   // 256 independent-ish FMAs per iteration (8 chains)
      float x[8];
      for (int c = 0; c < 8; ++c) x[c] = lane * 0.01f + c;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int s = 0; s < 32; ++s)
#pragma unroll
          for (int c = 0; c < 8; ++c) x[c] = __builtin_fmaf(x[c], 0.999f, 0.001f);
      for (int c = 0; c < 8; ++c) r += x[c];
    } else if (MODE == 2) {   // 64 LDS reads per iteration
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int s = 0; s < 64; ++s) r += lds[(s * 64 + lane + it) & 8191];
    }
    if (threadIdx.x == 512 && blockIdx.x == 0) g_t[1] = __builtin_readcyclecounter() - t0;
  }
  sink[blockIdx.x * 768 + threadIdx.x] = r;
}

template <int MODE, bool MFMA_ON>
void run(float* sink, int iters, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    k<MODE, MFMA_ON><<<256, 768>>>(sink, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t[8];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof(t));
  printf("%-44s %.3f ms   mfma waves %.1f cycles / MFMA   other wave %.1f cycles / iteration\n", what, ms,
         MFMA_ON ? (double)t[0] / (iters * 32.0) : 0.0, MODE ? (double)t[1] / iters : 0.0);
}

int main() {
  float* sink;
  hipMalloc(&sink, 256 * 768 * 4);
  const int iters = 2048;
  run<0, true>(sink, iters, "MFMA alone (2 waves / SIMD)");
  run<1, false>(sink, iters, "256 v_fma per iteration alone (1 wave / SIMD)");
  run<1, true>(sink, iters, "MFMA + 256 v_fma per iteration");
  run<2, false>(sink, iters, "64 ds_read per iteration alone");
  run<2, true>(sink, iters, "MFMA + 64 ds_read per iteration");
  return 0;
}
