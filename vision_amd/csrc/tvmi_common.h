// tvmi_common.h — shared device/host helpers for the gfx950 kernels.
// Only HIP runtime headers are used here (no torch): the kernels library is built with
// hipcc 7.2 while torch ships its own ROCm 7.0 runtime, so the two sides only exchange
// raw pointers and a hipStream_t (SURVEY.md §7 "hard parts" (2)).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/tvmi.h"

namespace tvmi {

constexpr int kWave = 64;  // gfx950 wavefront width; hard-coded on purpose.

// Records a failure for tvmi_last_error() and returns its code.
int set_error(int code, const char* what);

#define TVMI_CHECK_ARG(cond, msg)                                  \
  do {                                                             \
    if (!(cond)) return ::tvmi::set_error(hipErrorInvalidValue, msg); \
  } while (0)

#define TVMI_RETURN_LAUNCH_STATUS(name)                       \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return ::tvmi::set_error((int)e__, name); \
    return 0;                                                 \
  } while (0)

// Storage type <-> accumulation type.  fp16/bf16 tensors are read and written in their
// own format and accumulated in fp32; fp64 stays fp64.
template <typename T>
struct Acc {
  using type = float;
};
template <>
struct Acc<double> {
  using type = double;
};

template <typename T>
__device__ __forceinline__ typename Acc<T>::type ld(const T* p) {
  return static_cast<typename Acc<T>::type>(*p);
}
template <>
__device__ __forceinline__ float ld<__half>(const __half* p) {
  return __half2float(*p);
}
template <>
__device__ __forceinline__ float ld<__hip_bfloat16>(const __hip_bfloat16* p) {
  return __bfloat162float(*p);
}

template <typename T>
__device__ __forceinline__ void st(T* p, typename Acc<T>::type v) {
  *p = static_cast<T>(v);
}
template <>
__device__ __forceinline__ void st<__half>(__half* p, float v) {
  *p = __float2half(v);
}
template <>
__device__ __forceinline__ void st<__hip_bfloat16>(__hip_bfloat16* p, float v) {
  *p = __float2bfloat16(v);
}

// Atomic accumulate into a tensor element of storage type T.
// LDS float add that stays a ds_add_f32: an `atomicAdd` on a generic pointer that the optimiser merges with a global atomic of
// the same value (select of pointers) becomes ONE flat atomic, which resolves its aperture per lane in the texture path —
// measured ~170 cycles per wave instruction in dcn_bwd_data_mfma_win against a few cycles for the LDS form.
__device__ __forceinline__ void lds_add_f32(float* p, float v) {
  typedef __attribute__((address_space(3))) float lds_float;
  __hip_atomic_fetch_add((lds_float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void atomic_accum(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_accum(double* p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_accum(__half* p, float v) {
  // 16-bit CAS on the containing 32-bit word.
  unsigned int* base = reinterpret_cast<unsigned int*>(reinterpret_cast<uintptr_t>(p) & ~uintptr_t(3));
  const bool hi = (reinterpret_cast<uintptr_t>(p) & 2) != 0;
  unsigned int old = *base, assumed;
  do {
    assumed = old;
    unsigned short h = hi ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xffffu);
    __half cur = __ushort_as_half(h);
    unsigned short nh = __half_as_ushort(__float2half(__half2float(cur) + v));
    unsigned int repl = hi ? ((assumed & 0x0000ffffu) | ((unsigned int)nh << 16))
                           : ((assumed & 0xffff0000u) | nh);
    old = atomicCAS(base, assumed, repl);
  } while (old != assumed);
}
__device__ __forceinline__ void atomic_accum(__hip_bfloat16* p, float v) {
  unsigned int* base = reinterpret_cast<unsigned int*>(reinterpret_cast<uintptr_t>(p) & ~uintptr_t(3));
  const bool hi = (reinterpret_cast<uintptr_t>(p) & 2) != 0;
  unsigned int old = *base, assumed;
  do {
    assumed = old;
    unsigned short h = hi ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xffffu);
    float cur = __uint_as_float((unsigned int)h << 16);
    __hip_bfloat16 nb = __float2bfloat16(cur + v);
    unsigned short nh = *reinterpret_cast<unsigned short*>(&nb);
    unsigned int repl = hi ? ((assumed & 0x0000ffffu) | ((unsigned int)nh << 16))
                           : ((assumed & 0xffff0000u) | nh);
    old = atomicCAS(base, assumed, repl);
  } while (old != assumed);
}

// Exact unsigned division of n < 2^16 by d in [1, 2^16) with one mul-hi.
// m = floor(2^32/d)+1 is exact on that domain; d == 1 is handled by the caller path.
struct FastDiv16 {
  unsigned int m;
  unsigned int d;
  __device__ __forceinline__ void init(unsigned int div) {
    d = div;
    m = (div <= 1u) ? 0u : (unsigned int)(0x100000000ull / div) + 1u;
  }
  __device__ __forceinline__ unsigned int div(unsigned int n) const {
    return d <= 1u ? n : __umulhi(n, m);
  }
};

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

#define TVMI_DISPATCH_FLOAT(dt, NAME, ...)                                   \
  switch (dt) {                                                              \
    case TVMI_F32: {                                                         \
      using scalar_t = float;                                                \
      __VA_ARGS__;                                                           \
      break;                                                                 \
    }                                                                        \
    case TVMI_F64: {                                                         \
      using scalar_t = double;                                               \
      __VA_ARGS__;                                                           \
      break;                                                                 \
    }                                                                        \
    case TVMI_F16: {                                                         \
      using scalar_t = __half;                                               \
      __VA_ARGS__;                                                           \
      break;                                                                 \
    }                                                                        \
    case TVMI_BF16: {                                                        \
      using scalar_t = __hip_bfloat16;                                       \
      __VA_ARGS__;                                                           \
      break;                                                                 \
    }                                                                        \
    default:                                                                 \
      return ::tvmi::set_error(hipErrorInvalidValue, NAME ": unsupported dtype"); \
  }

}  // namespace tvmi
