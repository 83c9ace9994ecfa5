// Probe: what does the texture path charge a global_load_lds_dwordx4 for — lanes, quads of lanes, or the 64-byte sectors a quad
// touches?  A wave stages windows of `wrows` rows x `wcols` floats out of 64 channel planes (H x W fp32, L2-resident, a new plane
// per instruction like the RoIAlign forward) in three lane layouts:
//   0 "pieces"  : the shipped layout — a row = lpr (odd) 16-byte pieces from the 16-byte boundary at or below x0, rows packed
//                 back to back over the 64 lanes (quads of lanes cross rows and sectors)
//   1 "sectors" : a row = spr aligned 64-byte sectors, one sector per quad of lanes (4 lanes x 16 B), 16 / spr rows per instruction
//   2 "pieces8" : like 0 with an EVEN lpr rounded to a multiple of 4 (quads never cross rows, sectors not aligned)
//   3 "dword"   : global_load_lds_dword, one float per lane, rows packed back to back (no padding at all)
// Prints cycles per instruction per CU and per staged row, at 16 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gptr_t;

struct Cfg {
  int layout, H, W, wcols, spr;
};

template <int DW>
__global__ __launch_bounds__(256) void k(const float* __restrict__ maps, float* __restrict__ out, int iters, Cfg c, int* rows_out) {
  __shared__ __attribute__((aligned(16))) float lds[4][2][64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned wid = blockIdx.x * 4 + wave;
  const unsigned h = wid * 2654435761u;
  const int x0 = (int)((h >> 8) % (unsigned)(c.W - c.wcols - 20));
  const int y0 = (int)((h >> 20) % (unsigned)(c.H - 34));
  // lane -> element offset inside a plane
  int off, rows_per_instr;
  if (c.layout == 0 || c.layout == 2) {
    const int xal = x0 - (x0 & 3);
    int nq = (x0 + c.wcols - 1 - xal + 4) / 4;
    int lpr = c.layout == 0 ? (nq | 1) : ((nq + 3) & ~3);
    rows_per_instr = 64 / lpr;
    const int row = min(lane / lpr, rows_per_instr - 1), q = min(lane % lpr, nq - 1);
    off = (y0 + row) * c.W + xal + 4 * q;
  } else if (c.layout == 1) {
    rows_per_instr = 16 / c.spr;
    const int quad = lane >> 2, p = lane & 3;
    const int row = min(quad / c.spr, rows_per_instr - 1), sj = quad % c.spr;
    const int e0 = (y0 + row) * c.W + x0;            // first wanted element of the row
    const int s0 = e0 & ~15;                          // its 64-byte sector (planes are 64-byte aligned: H*W % 16 == 0 or not — see main)
    off = s0 + 16 * sj + 4 * p;
  } else {
    rows_per_instr = 64 / c.wcols;
    const int row = min(lane / c.wcols, rows_per_instr - 1), x = lane % c.wcols;
    off = (y0 + row) * c.W + x0 + x;
  }
  if (lane == 0) atomicAdd(rows_out, rows_per_instr);
  const size_t plane = (size_t)c.H * c.W;
  const int step_rows = rows_per_instr * c.W;
  const int pm = c.W > 200 ? 7 : 31;   // 16 or 64 planes: 4.3 MB either way (one XCD's L2)
  for (int i = 0; i < iters; ++i) {
    // 32 channels of one row group, then the next row group (3 groups), then over again from another plane set
    const float* p = maps + (size_t)((i & pm) + (pm + 1) * ((i >> 7) & 1)) * plane + off + ((i >> 5) & 3) * step_rows;
    if (DW == 4)
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lds_ptr_t)&lds[wave][i & 1][0], 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lds_ptr_t)&lds[wave][i & 1][0], 4, 0, 0);
    if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  __builtin_amdgcn_s_waitcnt(0);
  if (lds[wave][0][lane] == 12345.f) out[0] = 1.f;
}

int main() {
  float *maps, *out;
  int* rows;
  const size_t n = 64ull * 200 * 336 + 4096;
  hipMalloc(&maps, n * 4);
  hipMemset(maps, 0, n * 4);
  hipMalloc(&out, 4);
  hipMalloc(&rows, 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int blocks = 256 * 16, iters = 1024;
  const Cfg cfgs[] = {
      // P3-like planes (100 x 168: row pitch 672 B = 10.5 sectors), 23-float rows
      {0, 100, 168, 23, 0}, {2, 100, 168, 23, 0}, {1, 100, 168, 23, 2}, {1, 100, 168, 23, 3}, {3, 100, 168, 23, 0},
      // P2-like planes (200 x 336: 21 sectors per row)
      {0, 200, 336, 23, 0}, {2, 200, 336, 23, 0}, {1, 200, 336, 23, 2}, {1, 200, 336, 23, 3}, {3, 200, 336, 23, 0},
      // narrower / wider windows
      {0, 100, 168, 15, 0}, {1, 100, 168, 15, 2}, {3, 100, 168, 15, 0},
      {0, 100, 168, 31, 0}, {1, 100, 168, 31, 3}, {3, 100, 168, 31, 0},
      {1, 100, 168, 31, 4}, {1, 100, 168, 23, 1}, {1, 100, 168, 23, 4},
  };
  const char* names[] = {"pieces ", "sectors", "pieces4", "dword  "};
  for (const Cfg& c : cfgs) {
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      hipMemset(rows, 0, 4);
      hipEventRecord(a);
      if (c.layout == 3) k<1><<<blocks, 256>>>(maps, out, iters, c, rows);
      else k<4><<<blocks, 256>>>(maps, out, iters, c, rows);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    int rsum = 0;
    hipMemcpy(&rsum, rows, 4, hipMemcpyDeviceToHost);
    const double rpi = (double)rsum / ((double)blocks * 4);
    const double instr_per_cu = (double)blocks * 4 * iters / 256;
    const double clk = best * 1e-3 * 2.25e9 / instr_per_cu;
    printf("%s map %3dx%3d row %2d floats spr %d: %.3f ms  %6.1f clk/instr/CU  rows/instr %5.2f  %5.2f clk/row\n",
           names[c.layout], c.H, c.W, c.wcols, c.spr, best, clk, rpi, clk / rpi);
  }
  return 0;
}
