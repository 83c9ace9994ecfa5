// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access pattern of the RoIAlign DMA kernel:
// a streaming read of a known number of bytes with global_load_lds_dwordx4 (16 B per lane, lane-contiguous = 1 KiB
// per wave instruction) — kernel `calib_dma16` — and with 16 B per lane at a 28-byte lane stride (row fragments,
// unaligned) — kernel `calib_dma16_frag`; plus a plain float4 copy (`calib_copy`) whose read AND write bytes are known.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__global__ __launch_bounds__(256) void calib_dma16(const float* in, float* out, size_t n_float4_per_wave_iter, int iters) {
  __shared__ __attribute__((aligned(16))) float buf[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gw = (size_t)blockIdx.x * 4 + wave, nw = (size_t)gridDim.x * 4;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const float* src = in + ((size_t)it * nw + gw) * 256 + lane * 4;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)buf[wave], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += buf[wave][lane];
  }
  if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_dma16_frag(const float* in, float* out, int iters) {
  // each wave instruction: 12 "rows" of 5 lanes x 16 B = 80-byte fragments, rows 1344 B apart, start unaligned by 4 B
  __shared__ __attribute__((aligned(16))) float buf[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gw = (size_t)blockIdx.x * 4 + wave, nw = (size_t)gridDim.x * 4;
  const int r = lane / 5 > 11 ? 11 : lane / 5, q = lane % 5;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const float* src = in + ((size_t)it * nw + gw) * 336 * 12 + (size_t)r * 336 + 1 + q * 4;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)buf[wave], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += buf[wave][lane];
  }
  if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_copy(const float4* in, float4* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
int main() {
  const size_t bytes = 1ull << 30;  // 1 GiB source (> 256 MiB infinity cache)
  float *in, *out;
  hipMalloc(&in, bytes + (1 << 20)); hipMalloc(&out, bytes);
  hipMemset(in, 0, bytes); hipMemset(out, 0, bytes);
  const int blocks = 4096, iters = (int)(bytes / ((size_t)blocks * 4 * 1024));
  calib_dma16<<<blocks, 256>>>(in, out, 0, iters);
  hipDeviceSynchronize();
  printf("calib_dma16: read %zu bytes\n", (size_t)blocks * 4 * 1024 * iters);
  const int iters2 = (int)(bytes / ((size_t)blocks * 4 * 336 * 12 * 4));
  calib_dma16_frag<<<blocks, 256>>>(in, out, iters2);
  hipDeviceSynchronize();
  printf("calib_dma16_frag: requested %zu bytes (12x80B fragments per instr), touched-64B-sector bytes ~%zu\n",
         (size_t)blocks * 4 * iters2 * 12 * 80, (size_t)blocks * 4 * iters2 * 12 * 128);
  calib_copy<<<8192, 256>>>((const float4*)in, (float4*)out, bytes / 16);
  hipDeviceSynchronize();
  printf("calib_copy: read %zu bytes, wrote %zu bytes\n", bytes, bytes);
  return 0;
}
