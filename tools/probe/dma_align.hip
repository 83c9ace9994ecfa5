// Probe: does global_load_lds_dwordx4 accept 4-byte-aligned (not 16-byte-aligned) per-lane global addresses?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__global__ void k(const float* in, float* out, int shift) {
  __shared__ __attribute__((aligned(16))) float buf[256];
  const int lane = threadIdx.x;
  const float* src = in + shift + lane * 7;  // arbitrary 4-byte aligned, 28-byte stride per lane
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)buf, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = buf[lane * 4 + j];
}
int main() {
  float *in, *out;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 4);
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)i;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int shift = 0; shift < 4; ++shift) {
    k<<<1, 64>>>(in, out, shift);
    float o[256]; hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (o[l * 4 + j] != (float)(shift + l * 7 + j)) ++bad;
    printf("shift %d: bad=%d  lane1 got %g %g %g %g (want %d..)\n", shift, bad, o[4], o[5], o[6], o[7], shift + 7);
  }
  return 0;
}
