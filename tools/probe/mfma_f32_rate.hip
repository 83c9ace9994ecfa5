// Issue rate of v_mfma_f32_32x32x2_f32 on gfx950: cycles per MFMA for one / two waves per SIMD with the operands
//   mode 0: in registers
//   mode 1: read from LDS with ds_read_b32, the compiler's own schedule
//   mode 2: ds_read_b32, operands two k-steps ahead, order pinned with sched_barrier (the fp32 deform_conv2d inner loop)
//   mode 3: ds_read_b128 from a k-contiguous layout: one read feeds four k-steps of one operand
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_f32_rate.hip -o tools/probe/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned long long g_out[16];

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += NT) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    lds[i] = iters < 0 ? (float)i * 1e-6f : ((float)(h & 0xffffff) / 8388608.f - 1.f);   // iters < 0: smooth tiny values; else random in [-1, 1)
  }
  if (iters < 0) iters = -iters;
  __syncthreads();
  f32x16 acc[2];
  for (int m = 0; m < 2; ++m)
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, kq = lane >> 5;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (MODE == 0) {
    float a[2] = {1.f, 2.f}, b = 0.5f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int st = 0; st < 16; ++st)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b, acc[m], 0, 0, 0);
  } else if (MODE == 1) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        float a[2], b;
#pragma unroll
        for (int m = 0; m < 2; ++m) a[m] = lds[(2 * st + kq) * 256 + ((wave & 3) * 2 + m) * 32 + l31];
        b = lds[8192 + (2 * st + kq) * 64 + (wave & 1) * 32 + l31];
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b, acc[m], 0, 0, 0);
      }
  } else if (MODE == 2 || MODE == 4) {
    float aq[4][2], bq[4];
    for (int s = 0; s < 4; ++s) aq[s][0] = aq[s][1] = bq[s] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if (MODE == 4) __syncthreads();   // a workgroup barrier per 32 MFMAs, as between the rounds of the kernel
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int m = 0; m < 2; ++m) aq[s][m] = lds[(2 * s + kq) * 256 + ((wave & 3) * 2 + m) * 32 + l31];
        bq[s] = lds[8192 + (2 * s + kq) * 64 + (wave & 1) * 32 + l31];
      }
#pragma unroll
      for (int st = 0; st < 16; ++st) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          if (st + 2 < 16) {
            if (m == 0) {
#pragma unroll
              for (int m2 = 0; m2 < 2; ++m2) aq[(st + 2) % 4][m2] = lds[(2 * (st + 2) + kq) * 256 + ((wave & 3) * 2 + m2) * 32 + l31];
            } else {
              bq[(st + 2) % 4] = lds[8192 + (2 * (st + 2) + kq) * 64 + (wave & 1) * 32 + l31];
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st % 4][m], bq[st % 4], acc[m], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else if (MODE == 5) {   // barrier per 32 MFMAs; the last two k-steps of a round are multiplied behind the next round's first reads
    float aq[4][2], bq[4];
    for (int s = 0; s < 4; ++s) aq[s][0] = aq[s][1] = bq[s] = 0.f;
    for (int it = 0; it < iters; ++it) {
      __syncthreads();
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int m = 0; m < 2; ++m) aq[s][m] = lds[(2 * s + kq) * 256 + ((wave & 3) * 2 + m) * 32 + l31];
        bq[s] = lds[8192 + (2 * s + kq) * 64 + (wave & 1) * 32 + l31];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i2 = 0; i2 < 16; ++i2) {
        const int st = i2 < 2 ? i2 + 2 : i2 - 2;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          if (i2 >= 2 && st + 2 < 16) {
            if (m == 0) {
#pragma unroll
              for (int m2 = 0; m2 < 2; ++m2) aq[(st + 2) % 4][m2] = lds[(2 * (st + 2) + kq) * 256 + ((wave & 3) * 2 + m2) * 32 + l31];
            } else {
              bq[(st + 2) % 4] = lds[8192 + (2 * (st + 2) + kq) * 64 + (wave & 1) * 32 + l31];
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[st % 4][m], bq[st % 4], acc[m], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else {
    // k-contiguous: A[m][36], B[n][36] (pitch 36 floats); lane (l31, kq) owns k = 16 kq + 0..15 of its row: 4 b128 per operand
    const float4* A4 = reinterpret_cast<const float4*>(lds);
    for (int it = 0; it < iters; ++it) {
      float4 a4[2][4], b4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int m = 0; m < 2; ++m) a4[m][q] = A4[((((wave & 3) * 2 + m) * 32 + l31) * 36 + 16 * kq) / 4 + q];
        b4[q] = A4[(9216 + ((wave & 1) * 32 + l31) * 36 + 16 * kq) / 4 + q];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float av[2][4] = {{a4[0][q].x, a4[0][q].y, a4[0][q].z, a4[0][q].w}, {a4[1][q].x, a4[1][q].y, a4[1][q].z, a4[1][q].w}};
        const float bv[4] = {b4[q].x, b4[q].y, b4[q].z, b4[q].w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][c], bv[c], acc[m], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int m = 0; m < 2; ++m)
    for (int r = 0; r < 16; ++r) s += acc[m][r];
  sink[blockIdx.x * NT + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) g_out[MODE] = t1 - t0;
}

template <int MODE, int NT>
void run(float* sink, int wgs, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    k<MODE, NT><<<wgs, NT>>>(sink, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c[16];
  hipMemcpyFromSymbol(c, HIP_SYMBOL(g_out), sizeof(c));
  const double mf = (double)iters * 32;
  printf("wgs %d x %d threads mode %d: %.3f ms  %.1f cycles per own MFMA (wave 0)  clock %.2f GHz  %.1f TFLOP/s\n", wgs, NT, MODE, ms,
         (double)c[MODE] / mf, (double)c[MODE] / (ms * 1e6), (double)wgs * (NT / 64) * mf * 4096 / (ms * 1e9));
}

int main() {
  float* sink;
  hipMalloc(&sink, 1024 * 512 * 4);
  const int iters = 512;
  run<0, 256>(sink, 256, iters);
  run<1, 256>(sink, 256, iters);
  run<2, 256>(sink, 256, iters);
  run<3, 256>(sink, 256, iters);
  run<0, 512>(sink, 256, iters);
  run<1, 512>(sink, 256, iters);
  run<2, 512>(sink, 256, iters);
  run<3, 512>(sink, 256, iters);
  run<4, 512>(sink, 256, iters);
  run<5, 512>(sink, 256, iters);
  printf("smooth tiny operands:\n");
  run<5, 512>(sink, 256, -iters);
  run<2, 512>(sink, 256, -iters);
  return 0;
}
