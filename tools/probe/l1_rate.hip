// Microbenchmark: per-CU load rate for L2/L1-resident data on gfx950:
//  mode 0: global_load_dwordx4 -> VGPR     mode 1: global_load_lds_dwordx4 (LDS-DMA, 16 B/lane)
//  mode 2: global_load_lds_dword (4 B/lane) mode 3: global_load_dword -> VGPR
//  pattern: contiguous 1 KiB per instruction (frag=0) or row fragments of `frag` bytes from rows `pitch` bytes apart.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gptr_t;
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ buf, float* __restrict__ out, int iters, int frag_q, int pitch_f,
                                         int foot_rows) {
  __shared__ __attribute__((aligned(16))) float lds[4][2][64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wid = (size_t)blockIdx.x * 4 + wave;
  // lane -> (row, quad) of a window whose rows hold frag_q quads (16 B each)
  const int row = lane / frag_q, q = lane % frag_q;
  const int rows_per_instr = 64 / frag_q;
  const float* base = buf + (wid % 512) * 4096 * 0 + (size_t)(blockIdx.x % 64) * 1024 * 64;
  float4 acc = {0, 0, 0, 0};
  float a1 = 0;
  for (int i = 0; i < iters; ++i) {
    const int r = (i * rows_per_instr + row) % foot_rows;
    const float* p = base + (size_t)r * pitch_f + q * 4;
    if (MODE == 0) {
      const float4 v = *(const float4*)p;
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    } else if (MODE == 1) {
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lds_ptr_t)&lds[wave][i & 1][0], 16, 0, 0);
    } else if (MODE == 2) {
      __builtin_amdgcn_global_load_lds((gptr_t)(p), (lds_ptr_t)&lds[wave][i & 1][0], 4, 0, 0);
    } else {
      a1 += *p;
    }
  }
  if (MODE == 1 || MODE == 2) {
    __builtin_amdgcn_s_waitcnt(0);
    a1 = lds[wave][0][lane];
  }
  if (acc.x + acc.y + acc.z + acc.w + a1 == 12345.f) out[0] = 1.f;
}
int main() {
  float *buf, *out;
  const size_t n = 64ull << 20;
  hipMalloc(&buf, n * 4);
  hipMemset(buf, 0, n * 4);
  hipMalloc(&out, 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int blocks = 256 * 8, iters = 2048;
  struct Cfg { int frag_q, pitch_f, foot_rows; };
  // contiguous 1 KiB (frag_q=64 quads => one row of 1 KiB), 80-B fragments (5 quads), 128-B (8), 32 B (2)
  const Cfg cfgs[] = {{64, 256, 16}, {16, 336, 32}, {8, 336, 32}, {5, 336, 40}, {4, 336, 32}, {2, 336, 32}, {1, 336, 64}};
  for (int mode = 0; mode < 4; ++mode)
    for (const Cfg& c : cfgs) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        if (mode == 0) k<0><<<blocks, 256>>>(buf, out, iters, c.frag_q, c.pitch_f, c.foot_rows);
        if (mode == 1) k<1><<<blocks, 256>>>(buf, out, iters, c.frag_q, c.pitch_f, c.foot_rows);
        if (mode == 2) k<2><<<blocks, 256>>>(buf, out, iters, c.frag_q, c.pitch_f, c.foot_rows);
        if (mode == 3) k<3><<<blocks, 256>>>(buf, out, iters, c.frag_q, c.pitch_f, c.foot_rows);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      const double instr = (double)blocks * 4 * iters;
      const int lanes_used = (64 / c.frag_q) * c.frag_q;
      const double bytes = instr * lanes_used * ((mode == 0 || mode == 1) ? 16 : 4);
      printf("mode %d frag %3d B pitch %d: %.3f ms  %.2f TB/s  (%.1f B/clk/CU @2.4GHz)  %.1f clk/instr/CU\n", mode, c.frag_q * 16,
             c.pitch_f * 4, best, bytes / best / 1e9, bytes / best / 1e9 * 1e12 / 256 / 2.4e9 / 1e0 / 1e0 / 1.0 / 1.0 / 1.0 / 1.0 / 1.0 / 1.0 * 1e0 / 1e0 / 1.0 / 1.0 / 1.0 * 1.0 / 1.0,
             best * 1e-3 * 2.4e9 / (instr / 256));
    }
  return 0;
}
