// Probe: s_load_dwordx4 / s_load_dword from 4-byte-aligned (not 16-byte-aligned) addresses on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const int* in, int* out, int shift) {
  const int* p = in + shift;
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  i32x4_t v;
  int w;
  asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dword %1, %2, 0x10\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v), "=&s"(w) : "s"(p) : "memory");
  if (threadIdx.x == 0) {
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3]; out[4] = w;
  }
}
int main() {
  int *in, *out;
  hipMalloc(&in, 4096); hipMalloc(&out, 64);
  int h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int shift = 0; shift < 9; ++shift) {
    k<<<1, 64>>>(in, out, shift * 5);
    int o[5]; hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
    printf("shift %d (byte %d): %d %d %d %d %d (want %d..)\n", shift * 5, shift * 20, o[0], o[1], o[2], o[3], o[4], shift * 5);
  }
  return 0;
}
