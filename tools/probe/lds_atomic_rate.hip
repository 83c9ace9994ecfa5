// Microbenchmark: LDS float-add throughput on gfx950 per CU, as the deform_conv2d backward window uses it.
// One 512-thread workgroup per CU (108 KB of LDS keeps it alone there), every wave issues `iters` batches of 64 adds.
//   mode 0  ds_add_f32, lanes on consecutive addresses of one channel row (no bank conflict)
//   mode 1  ds_add_f32, lanes on random addresses of one channel row (window-shaped: what a lane = pixel layout does)
//   mode 2  ds_add_f32, [pos][64 ch] layout, lane = channel (32 consecutive banks per half wave), random pos per half
//   mode 3  as 2 without atomics: ds_read_b32 + v_add + ds_write_b32 (legal when a wave owns its channels)
//   mode 4  as 3 with 2 channels per lane: ds_read_b64 / ds_write_b64
//   mode 5  as 3 with 4 channels per lane: ds_read_b128 / ds_write_b128
//   mode 6  ds_add_f32, all 64 lanes on the same address
//   mode 7  ds_pk_add_f16-free control: plain ds_write_b32 at random addresses (no read)
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ void lds_add(float* p, float v) {
  __hip_atomic_fetch_add((lds_float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
constexpr int WSZ = 425, CH = 64;
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, const int* rnd, int iters) {
  extern __shared__ __attribute__((aligned(16))) float win[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < CH * WSZ; e += 512) win[e] = 0.f;
  __syncthreads();
  int r[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) r[q] = rnd[(blockIdx.x * 512 + tid) * 16 + q];   // random position in [0, WSZ)
  float v = 1.f + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int c = (wave * 8 + q * 4 + kk) & (CH - 1);
        if (MODE == 0) lds_add(win + c * WSZ + ((lane + q * 7 + kk + it) % WSZ), v);
        if (MODE == 1) lds_add(win + c * WSZ + ((r[q] + kk * 26 + it) % WSZ), v);
        if (MODE == 6) lds_add(win + c * WSZ + ((q + kk + it) % WSZ), v);
        if (MODE == 7) win[c * WSZ + ((r[q] + kk * 26 + it) % WSZ)] = v;
        if (MODE == 2 || MODE == 3) {
          // position uniform per half wave (the pixel of accumulator register q), lane = channel
          const int pos = (__shfl(r[q], lane & 32) + kk * 26 + it) % WSZ;
          float* p = win + pos * CH + ((wave & 1) * 32 + (lane & 31));
          if (MODE == 2) lds_add(p, v);
          else *p = *p + v;
        }
        if (MODE == 4) {
          const int pos = (__shfl(r[q], lane & 32) + kk * 26 + it) % WSZ;
          float2* p = reinterpret_cast<float2*>(win + pos * CH) + (lane & 31);
          float2 t = *p;
          t.x += v;
          t.y += v;
          *p = t;
        }
        if (MODE == 5) {
          const int pos = (__shfl(r[q], lane & 48) + kk * 26 + it) % WSZ;
          float4* p = reinterpret_cast<float4*>(win + pos * CH) + (lane & 15);
          float4 t = *p;
          t.x += v;
          t.y += v;
          t.z += v;
          t.w += v;
          *p = t;
        }
      }
    }
  }
  __syncthreads();
  float s = 0.f;
  for (int e = tid; e < CH * WSZ; e += 512) s += win[e];
  out[blockIdx.x * 512 + tid] = s;
}
template <int MODE>
void run(float* out, const int* rnd, const char* what, int lanes_per_instr_factor) {
  const int blocks = 256, iters = 200;
  const size_t lds = (size_t)CH * WSZ * 4;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    k<MODE><<<blocks, 512, lds>>>(out, rnd, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double instr_per_cu = 8.0 * iters * 64;   // wave-level add batches (one LDS update instruction or pair each)
  const double cyc = best * 1e-3 * 2.4e9;
  printf("mode %d %-58s %.3f ms  %.1f cycles per wave-level update per CU (%d floats each)  %.2f floats/clk/CU\n", MODE, what, best,
         cyc / instr_per_cu, 64 * lanes_per_instr_factor, 64.0 * lanes_per_instr_factor * instr_per_cu / cyc);
}
int main() {
  float* out;
  int* rnd;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&rnd, 256 * 512 * 16 * 4);
  int* h = (int*)malloc(256 * 512 * 16 * 4);
  srand(1);
  for (int i = 0; i < 256 * 512 * 16; ++i) h[i] = rand() % WSZ;
  hipMemcpy(rnd, h, 256 * 512 * 16 * 4, hipMemcpyHostToDevice);
  run<0>(out, rnd, "ds_add_f32 consecutive lanes", 1);
  run<1>(out, rnd, "ds_add_f32 random positions of a channel row", 1);
  run<6>(out, rnd, "ds_add_f32 one address for the wave", 1);
  run<2>(out, rnd, "ds_add_f32 [pos][ch], lane = channel", 1);
  run<3>(out, rnd, "read+add+write b32 [pos][ch], lane = channel", 1);
  run<4>(out, rnd, "read+add+write b64 [pos][ch], lane = 2 channels", 2);
  run<5>(out, rnd, "read+add+write b128 [pos][ch], lane = 4 channels", 4);
  run<7>(out, rnd, "ds_write_b32 random positions (control)", 1);
  return 0;
}
