// resize.hip — 2-D interpolation (nearest / nearest-exact / bilinear / bicubic, with and
// without anti-aliasing) over contiguous NCHW planes, for gfx950.
//
// The reference's resize (torchvision/transforms/v2/functional/_geometry.py:283-362,
// transforms/_functional_tensor.py:441-474, models/detection/transform.py:65-72) is a thin
// wrapper over torch.nn.functional.interpolate; the arithmetic is PyTorch core's
// (third-party to the reference, pinned here at torch 2.10.0):
//   ATen/native/UpSample.h:259-340  scale / source-index rules (half-pixel, align_corners,
//                                   linear modes clamp the source index at 0, cubic not)
//   ATen/native/UpSample.h:400-435  cubic convolution coefficients, A = -0.75
//   ATen/native/UpSample.h:442-476  guard_index_and_lambda / index+lambda for linear
//   ATen/native/cuda/UpSample.cuh:263-358  anti-aliased (Pillow) filters: triangle support 1,
//                                   cubic a = -0.5 support 2, span and weight normalisation
// Kernels are output-stationary row tiles: a 256-lane workgroup owns a run of output pixels
// of one output row (coalesced stores), computes the source indices / weights of that run
// ONCE and then walks the N*C planes, so the index arithmetic is amortised over channels and
// every plane contributes independent loads in flight.  The anti-aliased kernels read
// per-axis weight tables built by a tiny pre-kernel (separable Pillow weights, exactly the
// reference's normalisation) instead of re-evaluating the filter per tap.
#include <algorithm>
#include <cmath>

#include "tvmi_common.h"

namespace tvmi {
namespace {

constexpr int kThreads = 256;
enum { MODE_BILINEAR = 0, MODE_BICUBIC = 1 };

// UpSample.h:259-287
__device__ __host__ inline float compute_scale(int64_t in, int64_t out, bool align, double scale_arg) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return scale_arg > 0. ? (float)(1.0 / scale_arg) : (float)in / (float)out;
}

// UpSample.h:289-318
__device__ __forceinline__ float source_index(float scale, int dst, bool align, bool cubic) {
  if (align) return scale * (float)dst;
  const float s = scale * ((float)dst + 0.5f) - 0.5f;
  return (!cubic && s < 0.f) ? 0.f : s;
}

__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2) * x - (A + 3)) * x * x + 1; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A; }

struct Lin {
  int i0, i1;
  float l0, l1;
};

// UpSample.h:450-476
__device__ __forceinline__ Lin linear_index(float scale, int dst, int in, int out, bool align) {
  Lin r;
  if (in == out) {
    r.i0 = r.i1 = dst;
    r.l0 = 1.f;
    r.l1 = 0.f;
    return r;
  }
  const float real = source_index(scale, dst, align, false);
  r.i0 = min((int)floorf(real), in - 1);
  r.l1 = fminf(fmaxf(real - (float)r.i0, 0.f), 1.f);
  r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
  r.l0 = 1.f - r.l1;
  return r;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void bilinear2d_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                              int NC, int IH, int IW, int OH, int OW, float sh,
                                                              float sw, int align, int nc_per_block) {
  const int ox = blockIdx.x * kThreads + threadIdx.x;
  const int oy = blockIdx.y;
  if (ox >= OW) return;
  const Lin y = linear_index(sh, oy, IH, OH, align != 0);
  const Lin x = linear_index(sw, ox, IW, OW, align != 0);
  const int nc0 = blockIdx.z * nc_per_block, nc1 = min(NC, nc0 + nc_per_block);
  const int64_t iplane = (int64_t)IH * IW, oplane = (int64_t)OH * OW;
  const int64_t o00 = (int64_t)y.i0 * IW + x.i0, o01 = (int64_t)y.i0 * IW + x.i1;
  const int64_t o10 = (int64_t)y.i1 * IW + x.i0, o11 = (int64_t)y.i1 * IW + x.i1;
  for (int nc = nc0; nc < nc1; ++nc) {
    const T* p = in + nc * iplane;
    const float v = y.l0 * (x.l0 * ld(p + o00) + x.l1 * ld(p + o01)) + y.l1 * (x.l0 * ld(p + o10) + x.l1 * ld(p + o11));
    st(out + nc * oplane + (int64_t)oy * OW + ox, v);
  }
}

// ---- bilinear, LDS-tiled: one wave = 256 output columns x TY output rows of a few planes ----------------------------
// The kernel above issues four 4-byte gathers and one 4-byte store per output; the texture path retires a wave load
// in ~16 cycles whatever its width, so at 8x3x1080x1920 -> 800x1422 it sits at 3 TB/s like ATen's kernel.  Here the
// input patch of the tile (<= 10 rows x 384 columns) is staged with 16-byte loads into LDS, every lane produces FOUR
// consecutive outputs of each of the TY rows from LDS (same operations in the same order as above: identical
// results) and stores them as one 16-byte non-temporal run.  Used when the patch fits (scale_w <= 1.49, scale_h <=
// 2.3: every detection-transform resize), fp32 / fp16 / bf16.
constexpr int kTileW = 256, kTileRows = 4, kPatchCols = 384, kPatchRows = 10;

template <typename T>
struct Vec4;
template <>
struct Vec4<float> {
  typedef float raw __attribute__((ext_vector_type(4))) __attribute__((aligned(4)));
  static __device__ __forceinline__ float up(float v) { return v; }
  static __device__ __forceinline__ float down(float v) { return v; }
};
template <>
struct Vec4<__half> {
  typedef unsigned short raw __attribute__((ext_vector_type(4))) __attribute__((aligned(2)));
  static __device__ __forceinline__ float up(unsigned short v) { return __half2float(__ushort_as_half(v)); }
  static __device__ __forceinline__ unsigned short down(float v) { return __half_as_ushort(__float2half(v)); }
};
template <>
struct Vec4<__hip_bfloat16> {
  typedef unsigned short raw __attribute__((ext_vector_type(4))) __attribute__((aligned(2)));
  static __device__ __forceinline__ float up(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
  static __device__ __forceinline__ unsigned short down(float v) {
    const __hip_bfloat16 b = __float2bfloat16(v);
    unsigned short u;
    __builtin_memcpy(&u, &b, 2);
    return u;
  }
};

template <typename T>
__global__ __launch_bounds__(64) void bilinear2d_tile_kernel(const T* __restrict__ in, T* __restrict__ out, int NC, int IH,
                                                             int IW, int OH, int OW, float sh, float sw, int align,
                                                             int nc_per_block) {
  typedef typename Vec4<T>::raw raw4;
  __shared__ float patch[kPatchRows][kPatchCols];
  const int lane = threadIdx.x;
  const int ox0 = blockIdx.x * kTileW, oy0 = blockIdx.y * kTileRows;
  const bool al = align != 0;
  // the lane's four output columns
  Lin xs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) xs[j] = linear_index(sw, min(ox0 + 4 * lane + j, OW - 1), IW, OW, al);
  const int nx = min(4, OW - (ox0 + 4 * lane));  // valid outputs of this lane (<= 0: none)
  // patch origin / extent (wave-uniform)
  const int xin0 = __builtin_amdgcn_readfirstlane(linear_index(sw, ox0, IW, OW, al).i0);
  const int xin1 = __builtin_amdgcn_readfirstlane(linear_index(sw, min(ox0 + kTileW - 1, OW - 1), IW, OW, al).i1);
  Lin ys[kTileRows];
#pragma unroll
  for (int r = 0; r < kTileRows; ++r) ys[r] = linear_index(sh, min(oy0 + r, OH - 1), IH, OH, al);
  const int yin0 = ys[0].i0, yin1 = ys[kTileRows - 1].i1;
  const int nrows = __builtin_amdgcn_readfirstlane(yin1 - yin0 + 1), nq = (xin1 - xin0 + 4) >> 2;
  const int nc0 = blockIdx.z * nc_per_block, nc1 = min(NC, nc0 + nc_per_block);
  const int64_t iplane = (int64_t)IH * IW, oplane = (int64_t)OH * OW;
  for (int nc = nc0; nc < nc1; ++nc) {
    const T* p = in + nc * iplane + (int64_t)yin0 * IW;
    for (int q = lane; q < nq; q += 64) {
      // a quad never leaves its row: the last one is shifted left, re-written columns carry identical values
      const int xsrc = min(xin0 + 4 * q, IW - 4);
      const int c = xsrc - xin0;
      raw4 v[kPatchRows];
#pragma unroll
      for (int r = 0; r < kPatchRows; ++r)  // all rows of the patch in flight (row count is wave-uniform)
        if (r < nrows) v[r] = *reinterpret_cast<const raw4*>(p + (int64_t)r * IW + xsrc);
#pragma unroll
      for (int r = 0; r < kPatchRows; ++r)
        if (r < nrows) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c + j >= 0) patch[r][c + j] = Vec4<T>::up(v[r][j]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (nx > 0) {
#pragma unroll
      for (int r = 0; r < kTileRows; ++r) {
        if (oy0 + r < OH) {
          const float* ra = patch[ys[r].i0 - yin0];
          const float* rb = patch[ys[r].i1 - yin0];
          raw4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c0 = xs[j].i0 - xin0, c1 = xs[j].i1 - xin0;
            const float v = ys[r].l0 * (xs[j].l0 * ra[c0] + xs[j].l1 * ra[c1]) + ys[r].l1 * (xs[j].l0 * rb[c0] + xs[j].l1 * rb[c1]);
            o[j] = Vec4<T>::down(v);
          }
          T* dst = out + nc * oplane + (int64_t)(oy0 + r) * OW + ox0 + 4 * lane;
          if (nx == 4) {
            __builtin_nontemporal_store(o, reinterpret_cast<raw4*>(dst));
          } else {
            for (int j = 0; j < nx; ++j) reinterpret_cast<decltype(Vec4<T>::down(0.f))*>(dst)[j] = o[j];
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void bicubic2d_kernel(const T* __restrict__ in, T* __restrict__ out, int NC,
                                                             int IH, int IW, int OH, int OW, float sh, float sw,
                                                             int align, int nc_per_block) {
  const int ox = blockIdx.x * kThreads + threadIdx.x;
  const int oy = blockIdx.y;
  if (ox >= OW) return;
  const int nc0 = blockIdx.z * nc_per_block, nc1 = min(NC, nc0 + nc_per_block);
  const int64_t iplane = (int64_t)IH * IW, oplane = (int64_t)OH * OW;
  if (IH == OH && IW == OW) {
    for (int nc = nc0; nc < nc1; ++nc) {
      const int64_t o = (int64_t)oy * OW + ox;
      out[nc * oplane + o] = in[nc * iplane + o];
    }
    return;
  }
  const float A = -0.75f;
  const float ry = source_index(sh, oy, align != 0, true);
  const float rx = source_index(sw, ox, align != 0, true);
  const int iy = min((int)floorf(ry), IH - 1), ix = min((int)floorf(rx), IW - 1);
  const float ty = fminf(fmaxf(ry - (float)iy, 0.f), 1.f), tx = fminf(fmaxf(rx - (float)ix, 0.f), 1.f);
  float cy[4], cx[4];
  cy[0] = cubic2(ty + 1.f, A);
  cy[1] = cubic1(ty, A);
  cy[2] = cubic1(1.f - ty, A);
  cy[3] = cubic2(1.f - ty + 1.f, A);
  cx[0] = cubic2(tx + 1.f, A);
  cx[1] = cubic1(tx, A);
  cx[2] = cubic1(1.f - tx, A);
  cx[3] = cubic2(1.f - tx + 1.f, A);
  int yy[4], xx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    yy[k] = max(min(iy - 1 + k, IH - 1), 0);
    xx[k] = max(min(ix - 1 + k, IW - 1), 0);
  }
  for (int nc = nc0; nc < nc1; ++nc) {
    const T* p = in + nc * iplane;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const T* row = p + (int64_t)yy[k] * IW;
      const float r = ld(row + xx[0]) * cx[0] + ld(row + xx[1]) * cx[1] + ld(row + xx[2]) * cx[2] + ld(row + xx[3]) * cx[3];
      acc += r * cy[k];
    }
    st(out + nc * oplane + (int64_t)oy * OW + ox, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void nearest2d_kernel(const T* __restrict__ in, T* __restrict__ out, int NC,
                                                             int IH, int IW, int OH, int OW, float sh, float sw,
                                                             int exact, int nc_per_block) {
  const int ox = blockIdx.x * kThreads + threadIdx.x;
  const int oy = blockIdx.y;
  if (ox >= OW) return;
  // UpSample.h:320-343 (floorf on float scale; "exact" adds the half-pixel offset)
  const int iy = exact ? min((int)floorf(((float)oy + 0.5f) * sh), IH - 1) : min((int)floorf((float)oy * sh), IH - 1);
  const int ix = exact ? min((int)floorf(((float)ox + 0.5f) * sw), IW - 1) : min((int)floorf((float)ox * sw), IW - 1);
  const int nc0 = blockIdx.z * nc_per_block, nc1 = min(NC, nc0 + nc_per_block);
  const int64_t iplane = (int64_t)IH * IW, oplane = (int64_t)OH * OW;
  const int64_t src = (int64_t)iy * IW + ix, dst = (int64_t)oy * OW + ox;
  for (int nc = nc0; nc < nc1; ++nc) out[nc * oplane + dst] = in[nc * iplane + src];
}

// ---- anti-aliased: per-axis tables.  Layout per output index i: [xmin, xsize, w[0..taps)]
__device__ __forceinline__ float aa_filter(float x, int mode) {
  if (x < 0.f) x = -x;
  if (mode == MODE_BILINEAR) return x < 1.f ? 1.f - x : 0.f;
  const float a = -0.5f;
  if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
  if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
  return 0.f;
}

__global__ void aa_table_kernel(float* __restrict__ table, int out_size, int in_size, float scale, int mode, int taps,
                                int align) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= out_size) return;
  float* row = table + (int64_t)i * (taps + 2);
  const int interp = mode == MODE_BILINEAR ? 2 : 4;
  const float support = scale >= 1.f ? (interp * 0.5f) * scale : interp * 0.5f;
  // UpSample.cuh:303-315: the centre is scale*(i+0.5) whatever align_corners says (align only
  // changes `scale`), as in ATen's anti-aliased CPU/CUDA kernels.
  (void)align;
  const float center = scale * ((float)i + 0.5f);
  const int xmin = max((int)(center - support + 0.5f), 0);
  const int xsize = max(min((int)(center + support + 0.5f), in_size) - xmin, 0);
  const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
  float total = 0.f;
  for (int j = 0; j < xsize && j < taps; ++j) {
    const float w = aa_filter(((float)j + (float)xmin - center + 0.5f) * invscale, mode);
    row[2 + j] = w;
    total += w;
  }
  for (int j = 0; j < taps; ++j) {
    if (j < xsize) {
      if (total != 0.f) row[2 + j] /= total;
    } else {
      row[2 + j] = 0.f;
    }
  }
  row[0] = __int_as_float(xmin);
  row[1] = __int_as_float(min(xsize, taps));
}

// ---- anti-aliased, LDS-tiled: one wave = 256 output columns x 4 output rows of a few planes -------------------------
// Separable inside the tile: the input patch (<= 16 rows x 380 columns) is staged with 16-byte loads, every lane forms
// the HORIZONTAL sums of its four output columns for every patch row (taps from LDS, weights in registers — the same
// operations in the same order as aa2d_kernel), then the four output rows are the VERTICAL combinations of those row
// sums; the vertical weights of the tile are one 4 x 16 matrix held across the lanes of a VGPR and enter as scalars
// (v_readlane), rows outside an output row's support are skipped (their weight is exactly 0).  The per-output
// kernels issue taps_y x taps_x gathers per output (20-56 at a 1.35x downscale) and are bound by the texture path;
// here a patch pixel is fetched once per tile.  Both axes <= 8 taps (scale <= 3.5 bilinear, <= 1.75 bicubic).
constexpr int kAATaps = 8, kAARows = 16;

// One axis of a separable filter: output index -> first source index, number of taps, weights (zero padded to kAATaps).
// TableAxis reads the Pillow tables of aa_table_kernel; CubicAxis evaluates torch's non-anti-aliased bicubic
// (UpSample.h:400-435, A = -0.75; taps clamped into the image are folded onto the border pixel: the weights of a
// repeated pixel are added first, a ~1e-7 relative change at the image border only).
struct TableAxis {
  const float* tab;
  int taps;
  __device__ __forceinline__ int first(int i) const { return __float_as_int(tab[(int64_t)i * (taps + 2)]); }
  __device__ __forceinline__ int end(int i) const {
    const float* r = tab + (int64_t)i * (taps + 2);
    return __float_as_int(r[0]) + __float_as_int(r[1]);
  }
  __device__ __forceinline__ void weights(int i, float (&w)[kAATaps]) const {
    const float* r = tab + (int64_t)i * (taps + 2);
#pragma unroll
    for (int k = 0; k < kAATaps; ++k) w[k] = k < taps ? r[2 + k] : 0.f;
  }
};
struct CubicAxis {
  float scale;
  int in;
  bool align;
  __device__ __forceinline__ void place(int i, int& ix, float& t) const {
    const float real = source_index(scale, i, align, true);
    ix = min((int)floorf(real), in - 1);
    t = fminf(fmaxf(real - (float)ix, 0.f), 1.f);
  }
  __device__ __forceinline__ int first(int i) const {
    int ix;
    float t;
    place(i, ix, t);
    return max(min(ix - 1, in - 1), 0);
  }
  __device__ __forceinline__ int end(int i) const {
    int ix;
    float t;
    place(i, ix, t);
    return max(min(ix + 2, in - 1), 0) + 1;
  }
  __device__ __forceinline__ void weights(int i, float (&w)[kAATaps]) const {
    int ix;
    float t;
    place(i, ix, t);
    const float A = -0.75f;
    const float c[4] = {cubic2(t + 1.f, A), cubic1(t, A), cubic1(1.f - t, A), cubic2(1.f - t + 1.f, A)};
    const int f = max(min(ix - 1, in - 1), 0);
#pragma unroll
    for (int k = 0; k < kAATaps; ++k) w[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int d = max(min(ix - 1 + k, in - 1), 0) - f;  // 0..3
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (d == q) w[q] += c[k];
    }
  }
};

template <typename T, typename AX, int TAPS>  // TAPS: horizontal taps evaluated per output (4 or 8)
__global__ __launch_bounds__(64) void aa2d_tile_kernel(const T* __restrict__ in, T* __restrict__ out, AX yax, AX xax, int NC,
                                                       int IH, int IW, int OH, int OW, int nc_per_block) {
  typedef typename Vec4<T>::raw raw4;
  __shared__ float patch[kAARows][kPatchCols];
  const int lane = threadIdx.x;
  const int ox0 = blockIdx.x * kTileW, oy0 = blockIdx.y * kTileRows;
  // cells right of the image edge are read with weight 0 (the weight rows are zero padded): make them finite once
  for (int i = lane; i < kAARows * kPatchCols; i += 64) (&patch[0][0])[i] = 0.f;
  // the lane's four output columns: first tap and weights
  int cx[4];
  float wx[4][TAPS];
  const int nx = min(4, OW - (ox0 + 4 * lane));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ox = min(ox0 + 4 * lane + j, OW - 1);
    cx[j] = xax.first(ox);
    float w[kAATaps];
    xax.weights(ox, w);
#pragma unroll
    for (int i = 0; i < TAPS; ++i) wx[j][i] = w[i];
  }
  const int xin0 = __builtin_amdgcn_readfirstlane(xax.first(ox0));
  const int xlast = __builtin_amdgcn_readfirstlane(xax.first(min(ox0 + kTileW - 1, OW - 1)));
  const int ncols = min(xlast + TAPS, IW) - xin0;  // staged columns (every tap index below stays inside the patch)
  const int nq = (ncols + 3) >> 2;
#pragma unroll
  for (int j = 0; j < 4; ++j) cx[j] -= xin0;
  // rows: patch origin, and the 4 x 16 matrix of vertical weights: lane (ro * 16 + r) holds the weight of patch row r
  // in output row ro (0 outside its support)
  const int yin0 = __builtin_amdgcn_readfirstlane(yax.first(min(oy0, OH - 1)));
  const int nrows = __builtin_amdgcn_readfirstlane(yax.end(min(oy0 + kTileRows - 1, OH - 1))) - yin0;
  float wv = 0.f;
  {
    const int ro = lane >> 4, r = lane & 15;
    const int oy = min(oy0 + ro, OH - 1);
    float wy[kAATaps];
    yax.weights(oy, wy);
    const int t = r - (yax.first(oy) - yin0);
#pragma unroll
    for (int k = 0; k < kAATaps; ++k)
      if (t == k) wv = wy[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int nc0 = blockIdx.z * nc_per_block, nc1 = min(NC, nc0 + nc_per_block);
  const int64_t iplane = (int64_t)IH * IW, oplane = (int64_t)OH * OW;
  for (int nc = nc0; nc < nc1; ++nc) {
    const T* p = in + nc * iplane + (int64_t)yin0 * IW;
    for (int q = lane; q < nq; q += 64) {
      // a quad never leaves its row: the last one is shifted left, re-written columns carry identical values
      const int xsrc = min(xin0 + 4 * q, IW - 4);
      const int c = xsrc - xin0;
#pragma unroll
      for (int r0 = 0; r0 < kAARows; r0 += 8) {  // eight rows in flight
        raw4 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if (r0 + r < nrows) v[r] = *reinterpret_cast<const raw4*>(p + (int64_t)(r0 + r) * IW + xsrc);
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if (r0 + r < nrows) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (c + j >= 0) patch[r0 + r][c + j] = Vec4<T>::up(v[r][j]);
          }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float acc[kTileRows][4];
#pragma unroll
    for (int ro = 0; ro < kTileRows; ++ro)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[ro][j] = 0.f;
#pragma unroll
    for (int r = 0; r < kAARows; ++r) {
      if (r < nrows) {
        float h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float* row = &patch[r][cx[j]];
          float t = 0.f;
#pragma unroll
          for (int i = 0; i < TAPS; ++i) t += row[i] * wx[j][i];
          h[j] = t;
        }
#pragma unroll
        for (int ro = 0; ro < kTileRows; ++ro) {
          const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv), ro * 16 + r));
          if (w != 0.f) {  // wave-uniform: outside the output row's support
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[ro][j] += h[j] * w;
          }
        }
      }
    }
    if (nx > 0) {
#pragma unroll
      for (int ro = 0; ro < kTileRows; ++ro) {
        if (oy0 + ro < OH) {
          raw4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = Vec4<T>::down(acc[ro][j]);
          T* dst = out + nc * oplane + (int64_t)(oy0 + ro) * OW + ox0 + 4 * lane;
          if (nx == 4) {
            __builtin_nontemporal_store(o, reinterpret_cast<raw4*>(dst));
          } else {
            for (int j = 0; j < nx; ++j) reinterpret_cast<decltype(Vec4<T>::down(0.f))*>(dst)[j] = o[j];
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// Output-stationary, one lane per output column.  The x weights of the lane's column live in registers for the
// whole block (MAXT compile-time taps, zero beyond xsize — the table rows are zero padded), the y weights of the
// block's output row are wave-uniform (scalar loads), and the tap loop is fully unrolled: ysize x MAXT independent
// loads + FMAs per output and plane, no table traffic inside the loops.  Indices past the row end are clamped (their
// weight is zero).  Neighbouring lanes' windows overlap by (1 - 1/scale), so the taps are L1 hits.
template <typename T, int MAXT>
__global__ __launch_bounds__(kThreads) void aa2d_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                        const float* __restrict__ ytab, const float* __restrict__ xtab,
                                                        int NC, int IH, int IW, int OH, int OW, int ytaps, int xtaps,
                                                        int nc_per_block) {
  const int ox = blockIdx.x * kThreads + threadIdx.x;
  const int oy = blockIdx.y;
  if (ox >= OW) return;
  const float* yr = ytab + (int64_t)oy * (ytaps + 2);
  const float* xr = xtab + (int64_t)ox * (xtaps + 2);
  const int ymin = __float_as_int(yr[0]), ysize = __float_as_int(yr[1]);
  const int xmin = __float_as_int(xr[0]);
  float wx[MAXT];
  int xo[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    wx[i] = i < xtaps ? xr[2 + i] : 0.f;
    xo[i] = min(xmin + i, IW - 1);
  }
  const int nc0 = blockIdx.z * nc_per_block, nc1 = min(NC, nc0 + nc_per_block);
  const int64_t iplane = (int64_t)IH * IW, oplane = (int64_t)OH * OW;
  for (int nc = nc0; nc < nc1; ++nc) {
    const T* p = in + nc * iplane + (int64_t)ymin * IW;
    float acc = 0.f;
    for (int j = 0; j < ysize; ++j) {
      const T* row = p + (int64_t)j * IW;
      float r = 0.f;
#pragma unroll
      for (int i = 0; i < MAXT; ++i) r += ld(row + xo[i]) * wx[i];
      acc += r * yr[2 + j];
    }
    st(out + nc * oplane + (int64_t)oy * OW + ox, acc);
  }
}

// arbitrary tap counts (downscale factors above ~7.5x bicubic / 15x bilinear): same loops with run-time bounds
template <typename T>
__global__ __launch_bounds__(kThreads) void aa2d_kernel_any(const T* __restrict__ in, T* __restrict__ out,
                                                            const float* __restrict__ ytab, const float* __restrict__ xtab,
                                                            int NC, int IH, int IW, int OH, int OW, int ytaps, int xtaps,
                                                            int nc_per_block) {
  const int ox = blockIdx.x * kThreads + threadIdx.x;
  const int oy = blockIdx.y;
  if (ox >= OW) return;
  const float* yr = ytab + (int64_t)oy * (ytaps + 2);
  const float* xr = xtab + (int64_t)ox * (xtaps + 2);
  const int ymin = __float_as_int(yr[0]), ysize = __float_as_int(yr[1]);
  const int xmin = __float_as_int(xr[0]), xsize = __float_as_int(xr[1]);
  const int nc0 = blockIdx.z * nc_per_block, nc1 = min(NC, nc0 + nc_per_block);
  const int64_t iplane = (int64_t)IH * IW, oplane = (int64_t)OH * OW;
  for (int nc = nc0; nc < nc1; ++nc) {
    const T* p = in + nc * iplane + (int64_t)ymin * IW + xmin;
    float acc = 0.f;
    for (int j = 0; j < ysize; ++j) {
      const T* row = p + (int64_t)j * IW;
      float r = 0.f;
      for (int i = 0; i < xsize; ++i) r += ld(row + i) * xr[2 + i];
      acc += r * yr[2 + j];
    }
    st(out + nc * oplane + (int64_t)oy * OW + ox, acc);
  }
}

inline int taps_for(int mode, float scale) {
  const int interp = mode == MODE_BILINEAR ? 2 : 4;
  const float support = scale >= 1.f ? (interp * 0.5f) * scale : interp * 0.5f;
  return (int)ceilf(support) * 2 + 1;
}


// ---------------------------------------------------------------------------------------
// Batched paste_masks_in_image (models/detection/roi_heads.py:378-437,486-500): for every
// detection n, zero-pad the M x M mask by `padding`, expand the box by (M+2p)/M, truncate to
// integers, bilinearly resize the padded mask to the integer box size (align_corners=False)
// and paste it into an im_h x im_w canvas.  The reference does this with a Python loop of
// F.interpolate + zeros + slice-assign + stack (its eval loop carries a FIXME about exactly
// that); here ONE output-stationary launch writes every canvas pixel once: a pixel outside its
// detection's clipped box is 0, a pixel inside evaluates its 4 taps straight from the
// (L2-resident) M x M mask.  The kernel is a pure HBM write stream (N*im_h*im_w elements).
// Box arithmetic follows expand_boxes op by op in float (this TU is built with
// -ffp-contract=off) so the integer box — and therefore the paste rectangle — is bit-exact.
template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void paste_masks_kernel(const T* __restrict__ masks, const float* __restrict__ boxes,
                                                               T* __restrict__ out, int M, int im_h, int im_w, int pad,
                                                               float scale) {
  const int n = blockIdx.y;
  const int64_t plane = (int64_t)im_h * im_w;
  const float* b = boxes + (int64_t)n * 4;
  const float b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
  float w_half = (b2 - b0) * 0.5f, h_half = (b3 - b1) * 0.5f;
  const float x_c = (b2 + b0) * 0.5f, y_c = (b3 + b1) * 0.5f;
  w_half *= scale;
  h_half *= scale;
  const long long e0 = (long long)(x_c - w_half), e2 = (long long)(x_c + w_half);
  const long long e1 = (long long)(y_c - h_half), e3 = (long long)(y_c + h_half);
  const long long w = max(e2 - e0 + 1, 1ll), h = max(e3 - e1 + 1, 1ll);
  const long long x_0 = max(e0, 0ll), x_1 = min(e2 + 1, (long long)im_w);
  const long long y_0 = max(e1, 0ll), y_1 = min(e3 + 1, (long long)im_h);
  const bool sane = h < (1ll << 30) && w < (1ll << 30);
  const int IN = M + 2 * pad;
  const float sh = (float)IN / (float)h, sw = (float)IN / (float)w;
  const T* m = masks + (int64_t)n * M * M;
  T* o = out + (int64_t)n * plane;
  auto pixel = [&](int y, int x) -> float {
    if (!(y >= y_0 && y < y_1 && x >= x_0 && x < x_1 && sane)) return 0.f;
    const Lin ly = linear_index(sh, (int)(y - e1), IN, (int)h, false);
    const Lin lx = linear_index(sw, (int)(x - e0), IN, (int)w, false);
    auto tap = [&](int iy, int ix) -> float {
      iy -= pad;
      ix -= pad;
      return (iy < 0 || iy >= M || ix < 0 || ix >= M) ? 0.f : ld(m + iy * M + ix);
    };
    return ly.l0 * (lx.l0 * tap(ly.i0, lx.i0) + lx.l1 * tap(ly.i0, lx.i1)) +
           ly.l1 * (lx.l0 * tap(ly.i1, lx.i0) + lx.l1 * tap(ly.i1, lx.i1));
  };
  // VEC consecutive canvas elements per lane = one 16-byte store (the canvas is a pure write stream)
  const int64_t nvec = plane / VEC;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * kThreads) {
    const int64_t e = i * VEC;
    int y = (int)(e / im_w), x = (int)(e - (int64_t)y * im_w);
    struct alignas(sizeof(T) * VEC) Pack { T v[VEC]; } pk;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      T t;
      st(&t, pixel(y, x));
      pk.v[j] = t;
      if (++x == im_w) {
        x = 0;
        ++y;
      }
    }
    if constexpr (sizeof(T) * VEC == 16) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      u32x4 u;
      __builtin_memcpy(&u, &pk, 16);
      __builtin_nontemporal_store(u, reinterpret_cast<u32x4*>(o + e));
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[e + j] = pk.v[j];
    }
  }
}

// ---------------------------------------------------------------------------------------
// GeneralizedRCNNTransform.forward for a batch of images (models/detection/transform.py:119-255):
// per image normalize ((x - mean) / std, :154-165), bilinear resize to the size the reference's
// scale rule gives (_resize_image_and_masks :25-72, F.interpolate align_corners=False), then
// zero-padded batching to a common size divisible by 32 (batch_images :228-246).  The reference
// runs ~6 launches and two host->device copies PER IMAGE; here one output-stationary launch
// writes the whole [B,3,Hp,Wp] batch once.  Each of the 4 taps is normalised exactly like the
// reference normalises the source pixel before the interpolation consumes it.
constexpr int kXformMaxImages = 64;
struct XformImages {
  const void* ptr[kXformMaxImages];
  int H[kXformMaxImages], W[kXformMaxImages];    // source size
  int OH[kXformMaxImages], OW[kXformMaxImages];  // resized size (<= Hp, Wp)
  float mean[4], stdv[4];
  int C;
};

template <typename T>
__global__ __launch_bounds__(kThreads) void normalize_resize_batch_kernel(XformImages im, T* __restrict__ out, int Hp, int Wp) {
  const int x = blockIdx.x * kThreads + threadIdx.x;
  const int y = blockIdx.y, b = blockIdx.z;
  if (x >= Wp) return;
  const int H = im.H[b], W = im.W[b], OH = im.OH[b], OW = im.OW[b];
  T* o = out + ((int64_t)b * im.C * Hp + y) * Wp + x;
  const int64_t oplane = (int64_t)Hp * Wp;
  if (y >= OH || x >= OW) {
    for (int c = 0; c < im.C; ++c) st(o + c * oplane, 0.f);
    return;
  }
  const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
  const Lin ly = linear_index(sh, y, H, OH, false);
  const Lin lx = linear_index(sw, x, W, OW, false);
  const T* src = static_cast<const T*>(im.ptr[b]);
  const int64_t iplane = (int64_t)H * W;
  for (int c = 0; c < im.C; ++c) {
    const T* p = src + c * iplane;
    const float m = im.mean[c], sd = im.stdv[c];
    const float v00 = (ld(p + (int64_t)ly.i0 * W + lx.i0) - m) / sd, v01 = (ld(p + (int64_t)ly.i0 * W + lx.i1) - m) / sd;
    const float v10 = (ld(p + (int64_t)ly.i1 * W + lx.i0) - m) / sd, v11 = (ld(p + (int64_t)ly.i1 * W + lx.i1) - m) / sd;
    st(o + c * oplane, ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11));
  }
}

// ---------------------------------------------------------------------------------------
// Backward of all six interpolation modes (aten::upsample_{nearest2d,bilinear2d,bicubic2d}_backward,
// _upsample_nearest_exact2d_backward, _upsample_{bilinear2d,bicubic2d}_aa_backward; ATen/native/cuda/UpSampleBilinear2d.cu,
// UpSampleBicubic2d.cu, UpSampleNearest2d.cu of torch 2.10).  ATen's bilinear / bicubic / anti-aliased backward kernels are
// output-stationary and scatter with atomicAdd (non-deterministic sums; torch warns under use_deterministic_algorithms).
// Here the gradient is GATHERED: every mode is separable, so one axis is described by the forward table
//   output index o -> [first source index, number of taps, weights]            (bwd_axis_table_kernel / aa_table_kernel)
// and, because the first and last source index of an output are non-decreasing in o, the outputs that touch input index i
// form ONE contiguous range [lo(i), hi(i)) (bwd_axis_ranges_kernel, two binary searches).  A lane owns an input pixel and
// sums wy(oy, iy) * wx(ox, ix) * grad_out[oy][ox] over its two ranges in a fixed order: deterministic, no atomics, no
// zero-fill of grad_input, fp32 accumulation rounded once for the 16-bit types.  Taps that the forward clamps onto the
// same border pixel have their weights added in the table (CubicAxis does the same for the forward tile kernel).
enum { BWD_NEAREST = 0, BWD_NEAREST_EXACT = 1, BWD_BILINEAR = 2, BWD_BICUBIC = 3 };

__global__ void bwd_axis_table_kernel(float* __restrict__ table, int out_size, int in_size, float scale, int kind, int taps,
                                      int align) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= out_size) return;
  float* row = table + (int64_t)o * (taps + 2);
  int first = 0, n = 1;
  float w[4] = {1.f, 0.f, 0.f, 0.f};
  if (kind == BWD_NEAREST) {            // UpSample.h:320-343
    first = min((int)floorf((float)o * scale), in_size - 1);
  } else if (kind == BWD_NEAREST_EXACT) {
    first = min((int)floorf(((float)o + 0.5f) * scale), in_size - 1);
  } else if (kind == BWD_BILINEAR) {
    const Lin l = linear_index(scale, o, in_size, out_size, align != 0);
    first = l.i0;
    n = l.i1 - l.i0 + 1;
    w[0] = n == 1 ? l.l0 + l.l1 : l.l0;
    w[1] = n == 1 ? 0.f : l.l1;
  } else {  // bicubic, A = -0.75; source index not clamped at 0, taps clamped into the image (UpSample.h:400-435)
    {
      const float real = source_index(scale, o, align != 0, true);
      const int ix = min((int)floorf(real), in_size - 1);
      const float t = fminf(fmaxf(real - (float)ix, 0.f), 1.f), A = -0.75f;
      const float c[4] = {cubic2(t + 1.f, A), cubic1(t, A), cubic1(1.f - t, A), cubic2(1.f - t + 1.f, A)};
      first = max(min(ix - 1, in_size - 1), 0);
      const int last = max(min(ix + 2, in_size - 1), 0);
      n = last - first + 1;
      w[0] = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int d = max(min(ix - 1 + k, in_size - 1), 0) - first;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (d == q) w[q] += c[k];
      }
    }
  }
  row[0] = __int_as_float(first);
  row[1] = __int_as_float(n);
  for (int k = 0; k < taps; ++k) row[2 + k] = k < 4 ? w[k] : 0.f;
}

// range[i] = {first o whose taps end beyond i, first o whose taps start beyond i}
__global__ void bwd_axis_ranges_kernel(const float* __restrict__ table, int taps, int out_size, int in_size,
                                       int2* __restrict__ range) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= in_size) return;
  const int64_t pitch = taps + 2;
  int lo = 0, hi = out_size;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const float* r = table + mid * pitch;
    if (__float_as_int(r[0]) + __float_as_int(r[1]) > i) hi = mid; else lo = mid + 1;
  }
  const int first = lo;
  hi = out_size;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (__float_as_int(table[mid * pitch]) > i) hi = mid; else lo = mid + 1;
  }
  range[i] = make_int2(first, lo);
}

__device__ __forceinline__ float bwd_weight(const float* __restrict__ table, int taps, int o, int i) {
  const float* r = table + (int64_t)o * (taps + 2);
  const int d = i - __float_as_int(r[0]);
  return (d >= 0 && d < __float_as_int(r[1])) ? r[2 + d] : 0.f;
}

constexpr int kBwdPlanes = 4, kBwdCols = 8, kBwdRows = 8;

template <typename T>
__global__ __launch_bounds__(kThreads) void upsample2d_bwd_kernel(const T* __restrict__ gout, T* __restrict__ gin,
                                                                  const float* __restrict__ ytab,
                                                                  const float* __restrict__ xtab,
                                                                  const int2* __restrict__ yrange,
                                                                  const int2* __restrict__ xrange, int NC, int IH, int IW,
                                                                  int OH, int OW, int yt, int xt, int nc_per_block) {
  const int ix = blockIdx.x * kThreads + threadIdx.x;
  const int iy = blockIdx.y;
  if (ix >= IW) return;
  const int2 yr = yrange[iy], xr = xrange[ix];
  const int nc0 = blockIdx.z * nc_per_block, nc1 = min(NC, nc0 + nc_per_block);
  const int64_t iplane = (int64_t)IH * IW, oplane = (int64_t)OH * OW;
  // the weights of the first kBwdRows x kBwdCols outputs of the two ranges (all of them up to a 4x up-scale of the
  // interpolating modes, any down-scale) are looked up once and serve every plane; longer ranges re-read the tables
  const int cnt0 = min(kBwdCols, xr.y - xr.x), rows0 = min(kBwdRows, yr.y - yr.x);
  float wx0[kBwdCols], wy0[kBwdRows];
#pragma unroll
  for (int k = 0; k < kBwdCols; ++k) wx0[k] = k < cnt0 ? bwd_weight(xtab, xt, xr.x + k, ix) : 0.f;
#pragma unroll
  for (int j = 0; j < kBwdRows; ++j) wy0[j] = j < rows0 ? bwd_weight(ytab, yt, yr.x + j, iy) : 0.f;
  for (int nc = nc0; nc < nc1; nc += kBwdPlanes) {
    const int np = min(kBwdPlanes, nc1 - nc);
    float acc[kBwdPlanes];
#pragma unroll
    for (int p = 0; p < kBwdPlanes; ++p) acc[p] = 0.f;
    for (int xc = xr.x; xc < xr.y; xc += kBwdCols) {
      const int cnt = min(kBwdCols, xr.y - xc);
      float wx[kBwdCols];
      if (xc == xr.x) {
#pragma unroll
        for (int k = 0; k < kBwdCols; ++k) wx[k] = wx0[k];
      } else {
#pragma unroll
        for (int k = 0; k < kBwdCols; ++k) wx[k] = k < cnt ? bwd_weight(xtab, xt, xc + k, ix) : 0.f;
      }
      const T* g0 = gout + (int64_t)nc * oplane + (int64_t)yr.x * OW + xc;
#pragma unroll
      for (int j = 0; j < kBwdRows; ++j) {
        if (j < rows0) {
          const T* g = g0 + (int64_t)j * OW;
#pragma unroll
          for (int k = 0; k < kBwdCols; ++k) {
            if (k < cnt) {
              const float w = wy0[j] * wx[k];
#pragma unroll
              for (int p = 0; p < kBwdPlanes; ++p)
                if (p < np) acc[p] += w * ld(g + p * oplane + k);
            }
          }
        }
      }
      for (int oy = yr.x + kBwdRows; oy < yr.y; ++oy) {
        const float wy = bwd_weight(ytab, yt, oy, iy);
        const T* g = gout + (int64_t)nc * oplane + (int64_t)oy * OW + xc;
#pragma unroll
        for (int k = 0; k < kBwdCols; ++k) {
          if (k < cnt) {
            const float w = wy * wx[k];
#pragma unroll
            for (int p = 0; p < kBwdPlanes; ++p)
              if (p < np) acc[p] += w * ld(g + p * oplane + k);
          }
        }
      }
    }
#pragma unroll
    for (int p = 0; p < kBwdPlanes; ++p)
      if (p < np) st(gin + (int64_t)(nc + p) * iplane + (int64_t)iy * IW + ix, acc[p]);
  }
}

// The same sum with the loads of a block issued together: a block is R output rows x CC output columns (ONE CC-element load
// per row, at element alignment) of P planes — R*P independent loads in flight per lane before the first use, where the
// kernel above waits for every (row, column) group of 4 (its ranges differ per lane, so its loads sit under divergent
// branches).  Loads are unconditional: a block that overhangs the range is shifted back inside the tensor / re-reads the
// last row, and a term outside the lane's range is dropped with a select (never "multiplied by a zero weight": 0 * inf).
// Any range length is correct (the blocks loop); the host picks (R, CC, P) from an estimate of the longest range of each axis
// so that the common cases (2x nearest / bilinear up-scales, every down-scale) are a single block.
template <typename T, int CC>
struct RawVec;
template <int CC>
struct RawVec<float, CC> {
  typedef float type __attribute__((ext_vector_type(CC))) __attribute__((aligned(4)));
};
template <int CC>
struct RawVec<__half, CC> {
  typedef unsigned short type __attribute__((ext_vector_type(CC))) __attribute__((aligned(2)));
};
template <int CC>
struct RawVec<__hip_bfloat16, CC> {
  typedef unsigned short type __attribute__((ext_vector_type(CC))) __attribute__((aligned(2)));
};

template <typename T, int R, int CC, int P>
__global__ __launch_bounds__(kThreads) void upsample2d_bwd_vec_kernel(const T* __restrict__ gout, T* __restrict__ gin,
                                                                      const float* __restrict__ ytab,
                                                                      const float* __restrict__ xtab,
                                                                      const int2* __restrict__ yrange,
                                                                      const int2* __restrict__ xrange, int NC, int IH,
                                                                      int IW, int OH, int OW, int yt, int xt,
                                                                      int nc_per_block) {
  typedef typename RawVec<T, CC>::type vec;
  const int ix = blockIdx.x * kThreads + threadIdx.x;
  const int iy = blockIdx.y;
  if (ix >= IW) return;
  const int2 yr = yrange[iy], xr = xrange[ix];
  const int nc0 = blockIdx.z * nc_per_block, nc1 = min(NC, nc0 + nc_per_block);
  const int64_t iplane = (int64_t)IH * IW, oplane = (int64_t)OH * OW;
  // weights of the first block of each axis: looked up once for all planes (the only block up to a CC x R range)
  float wx0[CC], wy0[R];
  {
    const int xs = min(xr.x, OW - CC);
#pragma unroll
    for (int m = 0; m < CC; ++m) wx0[m] = (xs + m >= xr.x && xs + m < min(xr.x + CC, xr.y)) ? bwd_weight(xtab, xt, xs + m, ix) : 0.f;
#pragma unroll
    for (int j = 0; j < R; ++j) wy0[j] = yr.x + j < yr.y ? bwd_weight(ytab, yt, yr.x + j, iy) : 0.f;
  }
  for (int nc = nc0; nc < nc1; nc += P) {
    float acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) acc[p] = 0.f;
    const T* gp[P];
#pragma unroll
    for (int p = 0; p < P; ++p) gp[p] = gout + (int64_t)min(nc + p, nc1 - 1) * oplane;   // surplus planes re-read the last one
    for (int xc = xr.x; xc < xr.y; xc += CC) {
      const int xs = min(xc, OW - CC), xe = min(xc + CC, xr.y);
      float wx[CC];
      bool okx[CC];
#pragma unroll
      for (int m = 0; m < CC; ++m) {
        okx[m] = xs + m >= xc && xs + m < xe;
        wx[m] = xc == xr.x ? wx0[m] : (okx[m] ? bwd_weight(xtab, xt, xs + m, ix) : 0.f);
      }
      for (int yc = yr.x; yc < yr.y; yc += R) {
        float wy[R];
        int row[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
          wy[j] = yc == yr.x ? wy0[j] : (yc + j < yr.y ? bwd_weight(ytab, yt, yc + j, iy) : 0.f);
          row[j] = min(yc + j, yr.y - 1) * OW + xs;
        }
        vec v[P][R];
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
          for (int j = 0; j < R; ++j) v[p][j] = *reinterpret_cast<const vec*>(gp[p] + row[j]);
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const bool oky = yc + j < yr.y;
#pragma unroll
          for (int m = 0; m < CC; ++m) {
            const float w = wy[j] * wx[m];
            const bool ok = oky && okx[m];
#pragma unroll
            for (int p = 0; p < P; ++p) {
              const float t = acc[p] + w * Vec4<T>::up(v[p][j][m]);
              acc[p] = ok ? t : acc[p];
            }
          }
        }
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p)
      if (nc + p < nc1) st(gin + (int64_t)(nc + p) * iplane + (int64_t)iy * IW + ix, acc[p]);
  }
}

// nearest / nearest-exact forward for any element type: a copy of B-byte elements
template <typename U>
__global__ __launch_bounds__(kThreads) void nearest2d_bytes_kernel(const U* __restrict__ in, U* __restrict__ out, int NC,
                                                                   int IH, int IW, int OH, int OW, float sh, float sw,
                                                                   int exact, int nc_per_block) {
  const int ox = blockIdx.x * kThreads + threadIdx.x;
  const int oy = blockIdx.y;
  if (ox >= OW) return;
  const int iy = exact ? min((int)floorf(((float)oy + 0.5f) * sh), IH - 1) : min((int)floorf((float)oy * sh), IH - 1);
  const int ix = exact ? min((int)floorf(((float)ox + 0.5f) * sw), IW - 1) : min((int)floorf((float)ox * sw), IW - 1);
  const int nc0 = blockIdx.z * nc_per_block, nc1 = min(NC, nc0 + nc_per_block);
  const int64_t iplane = (int64_t)IH * IW, oplane = (int64_t)OH * OW;
  const int64_t src = (int64_t)iy * IW + ix, dst = (int64_t)oy * OW + ox;
  for (int nc = nc0; nc < nc1; ++nc) out[nc * oplane + dst] = in[nc * iplane + src];
}

// ---- channels_last (NHWC in memory) forward, all six modes: lanes run along (output column, channel) — the contiguous axis
// of both tensors — and the two axes come from the same [first, taps, weights] tables as the backward.  ATen keeps the memory
// format of a channels_last input (`suggest_memory_format`), the reference's resize_image goes out of its way to preserve it
// (_geometry.py:324-338); the planar kernels would need a layout copy on either side.
// CPT channels per lane: 4 when C % 4 == 0 (one 16-byte / 8-byte load per tap), 3 for RGB images (C == 3), 1 otherwise — the axis
// tables are looked up once per lane, not once per channel.  MAXT > 0: both axes have at most MAXT taps (nearest, bilinear,
// bicubic without anti-aliasing) and the tap loops are unrolled with their loads issued together; MAXT == 0: any tap count.
template <typename T, int CPT, int MAXT>
__global__ __launch_bounds__(kThreads) void upsample2d_nhwc_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                                   const float* __restrict__ ytab,
                                                                   const float* __restrict__ xtab, int C, int IH, int IW,
                                                                   int OH, int OW, int yt, int xt) {
  const int cgs = C / CPT;                              // channel groups per pixel
  const int e = blockIdx.x * kThreads + threadIdx.x;   // ox * cgs + channel group
  const int oy = blockIdx.y, n = blockIdx.z;
  if (e >= OW * cgs) return;
  const int ox = e / cgs, c0 = (e - ox * cgs) * CPT;
  const float* yrow = ytab + (int64_t)oy * (yt + 2);
  const float* xrow = xtab + (int64_t)ox * (xt + 2);
  const int y0 = __float_as_int(yrow[0]), ny = __float_as_int(yrow[1]);
  const int x0 = __float_as_int(xrow[0]), nx = __float_as_int(xrow[1]);
  const T* base = in + (int64_t)n * IH * IW * C + c0;
  float acc[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) acc[c] = 0.f;
  auto tap = [&](const T* p, float (&v)[CPT]) {
    if constexpr (CPT == 4) {
      const typename RawVec<T, 4>::type q = *reinterpret_cast<const typename RawVec<T, 4>::type*>(p);
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = Vec4<T>::up(q[c]);
    } else {
#pragma unroll
      for (int c = 0; c < CPT; ++c) v[c] = ld(p + c);
    }
  };
  if constexpr (MAXT > 0) {
    float wy[MAXT], wx[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      wy[j] = j < ny ? yrow[2 + j] : 0.f;
      wx[j] = j < nx ? xrow[2 + j] : 0.f;
    }
    float v[MAXT][MAXT][CPT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
#pragma unroll
      for (int k = 0; k < MAXT; ++k)   // taps beyond the range re-read the last one (never used: selected away below)
        tap(base + ((int64_t)(y0 + max(min(j, ny - 1), 0)) * IW + x0 + max(min(k, nx - 1), 0)) * C, v[j][k]);
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      float r[CPT];
#pragma unroll
      for (int c = 0; c < CPT; ++c) r[c] = 0.f;
#pragma unroll
      for (int k = 0; k < MAXT; ++k)
#pragma unroll
        for (int c = 0; c < CPT; ++c) r[c] = k < nx ? r[c] + wx[k] * v[j][k][c] : r[c];
#pragma unroll
      for (int c = 0; c < CPT; ++c) acc[c] = j < ny ? acc[c] + wy[j] * r[c] : acc[c];
    }
  } else {
    for (int j = 0; j < ny; ++j) {
      const T* row = base + ((int64_t)(y0 + j) * IW + x0) * C;
      float r[CPT];
#pragma unroll
      for (int c = 0; c < CPT; ++c) r[c] = 0.f;
      for (int k = 0; k < nx; ++k) {
        float v[CPT];
        tap(row + (int64_t)k * C, v);
#pragma unroll
        for (int c = 0; c < CPT; ++c) r[c] += xrow[2 + k] * v[c];
      }
#pragma unroll
      for (int c = 0; c < CPT; ++c) acc[c] += yrow[2 + j] * r[c];
    }
  }
  T* dst = out + (((int64_t)n * OH + oy) * OW + ox) * C + c0;
  if constexpr (CPT == 4) {
    typename RawVec<T, 4>::type q;
#pragma unroll
    for (int c = 0; c < 4; ++c) q[c] = Vec4<T>::down(acc[c]);
    *reinterpret_cast<typename RawVec<T, 4>::type*>(dst) = q;
  } else {
#pragma unroll
    for (int c = 0; c < CPT; ++c) st(dst + c, acc[c]);
  }
}

template <typename U>
__global__ __launch_bounds__(kThreads) void nearest2d_nhwc_bytes_kernel(const U* __restrict__ in, U* __restrict__ out, int C,
                                                                        int IH, int IW, int OH, int OW, float sh, float sw,
                                                                        int exact) {
  const int e = blockIdx.x * kThreads + threadIdx.x;
  const int oy = blockIdx.y, n = blockIdx.z;
  if (e >= OW * C) return;
  const int ox = e / C, c = e - ox * C;
  const int iy = exact ? min((int)floorf(((float)oy + 0.5f) * sh), IH - 1) : min((int)floorf((float)oy * sh), IH - 1);
  const int ix = exact ? min((int)floorf(((float)ox + 0.5f) * sw), IW - 1) : min((int)floorf((float)ox * sw), IW - 1);
  out[((int64_t)n * OH + oy) * OW * C + e] = in[(((int64_t)n * IH + iy) * IW + ix) * C + c];
}

struct Launch {
  dim3 grid;
  int nc_per_block;
};
inline Launch plan(int64_t NC, int64_t OH, int64_t OW) {
  // enough workgroups to fill 256 CUs several times over, otherwise keep planes together
  const int64_t xy = ceil_div(OW, kThreads) * OH;
  int64_t zsplit = std::min<int64_t>(NC, std::max<int64_t>(1, 4096 / std::max<int64_t>(xy, 1)));
  zsplit = std::min<int64_t>(zsplit, 65535);
  const int ncpb = (int)ceil_div(NC, zsplit);
  return {dim3((unsigned)ceil_div(OW, kThreads), (unsigned)OH, (unsigned)ceil_div(NC, ncpb)), ncpb};
}

}  // namespace
}  // namespace tvmi

using namespace tvmi;

#define TVMI_RESIZE_PROLOGUE(name)                                                                   \
  if (NC * OH * OW == 0) return 0;                                                                   \
  TVMI_CHECK_ARG(input && output, name ": null pointer");                                            \
  TVMI_CHECK_ARG(IH > 0 && IW > 0, name ": input spatial size must be positive");                    \
  TVMI_CHECK_ARG(OH <= 65535 && IH * IW < (1ll << 31) && OH * OW < (1ll << 31), name ": size too large"); \
  hipStream_t s = static_cast<hipStream_t>(stream);                                                  \
  const Launch L = plan(NC, OH, OW);

extern "C" int tvmi_upsample_bilinear2d(const void* input, void* output, tvmi_dtype dt, int64_t NC, int64_t IH,
                                        int64_t IW, int64_t OH, int64_t OW, int align_corners, double scale_h,
                                        double scale_w, void* stream) {
  TVMI_RESIZE_PROLOGUE("upsample_bilinear2d");
  const float sh = compute_scale(IH, OH, align_corners, scale_h), sw = compute_scale(IW, OW, align_corners, scale_w);
  // LDS-tiled kernel when the input patch of a 256 x 4 output tile fits its LDS block (see bilinear2d_tile_kernel)
  const double need_cols = std::ceil((double)(kTileW - 1) * (double)sw) + 3.0, need_rows = std::ceil((double)(kTileRows - 1) * (double)sh) + 3.0;
  if (dt != TVMI_F64 && IW >= 4 && need_cols <= (double)(kPatchCols - 4) && need_rows <= (double)kPatchRows) {
    const int64_t tiles = ceil_div(OW, kTileW) * ceil_div(OH, kTileRows);
    // below ~16 k tile-planes the chip is not full and the load -> barrier -> compute phases of a wave are exposed:
    // the per-output kernel is faster there (measured: 3x480x640 -> 800x1067, 9.3 vs 7.0 us)
    const int per = (int)std::max<int64_t>(1, std::min<int64_t>(8, NC * tiles / 8192));
    const dim3 grid((unsigned)ceil_div(OW, kTileW), (unsigned)ceil_div(OH, kTileRows), (unsigned)ceil_div(NC, per));
    if (grid.z <= 65535 && NC * tiles >= 16384) {
#define TVMI_BILINEAR_TILE(scalar_t)                                                                                   \
  bilinear2d_tile_kernel<scalar_t><<<grid, dim3(64), 0, s>>>((const scalar_t*)input, (scalar_t*)output, (int)NC, (int)IH, \
                                                            (int)IW, (int)OH, (int)OW, sh, sw, align_corners, per)
      if (dt == TVMI_F32) TVMI_BILINEAR_TILE(float);
      else if (dt == TVMI_F16) TVMI_BILINEAR_TILE(__half);
      else if (dt == TVMI_BF16) TVMI_BILINEAR_TILE(__hip_bfloat16);
      else return ::tvmi::set_error(hipErrorInvalidValue, "upsample_bilinear2d: unsupported dtype");
#undef TVMI_BILINEAR_TILE
      TVMI_RETURN_LAUNCH_STATUS("tvmi_upsample_bilinear2d");
    }
  }
  TVMI_DISPATCH_FLOAT(dt, "upsample_bilinear2d",
                      bilinear2d_kernel<scalar_t><<<L.grid, dim3(kThreads), 0, s>>>(
                          (const scalar_t*)input, (scalar_t*)output, (int)NC, (int)IH, (int)IW, (int)OH, (int)OW, sh,
                          sw, align_corners, L.nc_per_block));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_upsample_bilinear2d");
}

extern "C" int tvmi_upsample_bicubic2d(const void* input, void* output, tvmi_dtype dt, int64_t NC, int64_t IH,
                                       int64_t IW, int64_t OH, int64_t OW, int align_corners, double scale_h,
                                       double scale_w, void* stream) {
  TVMI_RESIZE_PROLOGUE("upsample_bicubic2d");
  const float sh = compute_scale(IH, OH, align_corners, scale_h), sw = compute_scale(IW, OW, align_corners, scale_w);
  {  // LDS-tiled separable kernel (aa2d_tile_kernel with on-the-fly cubic weights) when the tile's patch fits
    const double need_cols = std::ceil((double)(kTileW - 1) * (double)sw) + kAATaps + 3.0;
    const double need_rows = std::ceil((double)(kTileRows - 1) * (double)sh) + 4 + 3.0;
    const int64_t tiles = ceil_div(OW, kTileW) * ceil_div(OH, kTileRows);
    const int per = (int)std::max<int64_t>(1, std::min<int64_t>(8, NC * tiles / 8192));
    const dim3 grid((unsigned)ceil_div(OW, kTileW), (unsigned)ceil_div(OH, kTileRows), (unsigned)ceil_div(NC, per));
    if (dt != TVMI_F64 && IW >= 4 && !(IH == OH && IW == OW) && need_cols <= (double)(kPatchCols - 4) &&
        need_rows <= (double)kAARows && grid.z <= 65535 && NC * tiles >= 16384) {
      const CubicAxis yax{sh, (int)IH, align_corners != 0}, xax{sw, (int)IW, align_corners != 0};
#define TVMI_CUBIC_TILE(scalar_t)                                                                                      \
  aa2d_tile_kernel<scalar_t, CubicAxis, 4><<<grid, dim3(64), 0, s>>>((const scalar_t*)input, (scalar_t*)output, yax, xax, (int)NC, \
                                                                 (int)IH, (int)IW, (int)OH, (int)OW, per)
      if (dt == TVMI_F32) TVMI_CUBIC_TILE(float);
      else if (dt == TVMI_F16) TVMI_CUBIC_TILE(__half);
      else if (dt == TVMI_BF16) TVMI_CUBIC_TILE(__hip_bfloat16);
      else return ::tvmi::set_error(hipErrorInvalidValue, "upsample_bicubic2d: unsupported dtype");
#undef TVMI_CUBIC_TILE
      TVMI_RETURN_LAUNCH_STATUS("tvmi_upsample_bicubic2d");
    }
  }
  TVMI_DISPATCH_FLOAT(dt, "upsample_bicubic2d",
                      bicubic2d_kernel<scalar_t><<<L.grid, dim3(kThreads), 0, s>>>(
                          (const scalar_t*)input, (scalar_t*)output, (int)NC, (int)IH, (int)IW, (int)OH, (int)OW, sh,
                          sw, align_corners, L.nc_per_block));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_upsample_bicubic2d");
}

extern "C" int tvmi_upsample_nearest2d(const void* input, void* output, tvmi_dtype dt, int64_t NC, int64_t IH,
                                       int64_t IW, int64_t OH, int64_t OW, int exact, double scale_h, double scale_w,
                                       void* stream) {
  TVMI_RESIZE_PROLOGUE("upsample_nearest2d");
  const float sh = compute_scale(IH, OH, false, scale_h), sw = compute_scale(IW, OW, false, scale_w);
  TVMI_DISPATCH_FLOAT(dt, "upsample_nearest2d",
                      nearest2d_kernel<scalar_t><<<L.grid, dim3(kThreads), 0, s>>>(
                          (const scalar_t*)input, (scalar_t*)output, (int)NC, (int)IH, (int)IW, (int)OH, (int)OW, sh,
                          sw, exact, L.nc_per_block));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_upsample_nearest2d");
}

extern "C" size_t tvmi_upsample_aa2d_workspace_bytes(int mode, int64_t IH, int64_t IW, int64_t OH, int64_t OW,
                                                     int align_corners, double scale_h, double scale_w) {
  if (OH <= 0 || OW <= 0 || IH <= 0 || IW <= 0) return 0;
  const float sh = compute_scale(IH, OH, align_corners, scale_h), sw = compute_scale(IW, OW, align_corners, scale_w);
  return ((size_t)OH * (taps_for(mode, sh) + 2) + (size_t)OW * (taps_for(mode, sw) + 2)) * sizeof(float);
}

extern "C" int tvmi_upsample_aa2d(const void* input, void* output, tvmi_dtype dt, int mode, int64_t NC, int64_t IH,
                                  int64_t IW, int64_t OH, int64_t OW, int align_corners, double scale_h,
                                  double scale_w, void* workspace, size_t workspace_bytes, void* stream) {
  TVMI_RESIZE_PROLOGUE("upsample_aa2d");
  TVMI_CHECK_ARG(mode == MODE_BILINEAR || mode == MODE_BICUBIC, "upsample_aa2d: mode must be 0 (bilinear) or 1 (bicubic)");
  TVMI_CHECK_ARG(workspace && workspace_bytes >= tvmi_upsample_aa2d_workspace_bytes(mode, IH, IW, OH, OW, align_corners,
                                                                                    scale_h, scale_w),
                 "upsample_aa2d: workspace too small");
  const float sh = compute_scale(IH, OH, align_corners, scale_h), sw = compute_scale(IW, OW, align_corners, scale_w);
  const int yt = taps_for(mode, sh), xt = taps_for(mode, sw);
  float* ytab = static_cast<float*>(workspace);
  float* xtab = ytab + (size_t)OH * (yt + 2);
  aa_table_kernel<<<dim3((unsigned)ceil_div(OH, 128)), dim3(128), 0, s>>>(ytab, (int)OH, (int)IH, sh, mode, yt,
                                                                         align_corners);
  aa_table_kernel<<<dim3((unsigned)ceil_div(OW, 128)), dim3(128), 0, s>>>(xtab, (int)OW, (int)IW, sw, mode, xt,
                                                                         align_corners);
  // LDS-tiled kernel: both axes within 8 taps, the patch of a 256 x 4 output tile within its LDS block, enough tiles
  {
    const double need_cols = std::ceil((double)(kTileW - 1) * (double)sw) + kAATaps + 2.0;
    const double need_rows = std::ceil((double)(kTileRows - 1) * (double)sh) + yt + 2.0;
    const int64_t tiles = ceil_div(OW, kTileW) * ceil_div(OH, kTileRows);
    const int per = (int)std::max<int64_t>(1, std::min<int64_t>(8, NC * tiles / 8192));
    const dim3 grid((unsigned)ceil_div(OW, kTileW), (unsigned)ceil_div(OH, kTileRows), (unsigned)ceil_div(NC, per));
    if (dt != TVMI_F64 && IW >= 4 && xt <= kAATaps && yt <= kAATaps && need_cols <= (double)(kPatchCols - 4) &&
        need_rows <= (double)kAARows && grid.z <= 65535 && NC * tiles >= 16384) {
#define TVMI_AA_TILE(scalar_t)                                                                                          \
  aa2d_tile_kernel<scalar_t, TableAxis, 8><<<grid, dim3(64), 0, s>>>((const scalar_t*)input, (scalar_t*)output, TableAxis{ytab, yt}, \
                                                                 TableAxis{xtab, xt}, (int)NC, (int)IH, (int)IW, (int)OH,         \
                                                                 (int)OW, per)
      if (dt == TVMI_F32) TVMI_AA_TILE(float);
      else if (dt == TVMI_F16) TVMI_AA_TILE(__half);
      else if (dt == TVMI_BF16) TVMI_AA_TILE(__hip_bfloat16);
      else return ::tvmi::set_error(hipErrorInvalidValue, "upsample_aa2d: unsupported dtype");
#undef TVMI_AA_TILE
      TVMI_RETURN_LAUNCH_STATUS("tvmi_upsample_aa2d");
    }
  }
#define TVMI_AA(KERNEL)                                                                                       \
  KERNEL<<<L.grid, dim3(kThreads), 0, s>>>((const scalar_t*)input, (scalar_t*)output, ytab, xtab, (int)NC, (int)IH, \
                                           (int)IW, (int)OH, (int)OW, yt, xt, L.nc_per_block)
  TVMI_DISPATCH_FLOAT(dt, "upsample_aa2d", {
    if (xt <= 4)
      TVMI_AA((aa2d_kernel<scalar_t, 4>));
    else if (xt <= 8)
      TVMI_AA((aa2d_kernel<scalar_t, 8>));
    else if (xt <= 16)
      TVMI_AA((aa2d_kernel<scalar_t, 16>));
    else
      TVMI_AA(aa2d_kernel_any<scalar_t>);
  });
#undef TVMI_AA
  TVMI_RETURN_LAUNCH_STATUS("tvmi_upsample_aa2d");
}

namespace {
struct BwdPlan {
  int kind_y, kind_x, yt, xt;  // kind < 0: anti-aliased table of -(kind) - 1
  float sh, sw;
  size_t ytab_off, xtab_off, yr_off, xr_off, bytes;
};
// forward geometry: input [IH, IW] -> output [OH, OW]; the backward reads grad_output [OH, OW] and writes grad_input [IH, IW]
BwdPlan bwd_plan(int mode, int antialias, int64_t IH, int64_t IW, int64_t OH, int64_t OW, int align, double scale_h,
                 double scale_w) {
  BwdPlan p{};
  const bool lin = mode >= 2;
  p.sh = compute_scale(IH, OH, lin && align, scale_h);
  p.sw = compute_scale(IW, OW, lin && align, scale_w);
  if (lin && antialias) {
    p.kind_y = p.kind_x = -(mode - 2) - 1;
    p.yt = taps_for(mode - 2, p.sh);
    p.xt = taps_for(mode - 2, p.sw);
  } else {
    p.kind_y = p.kind_x = mode;
    p.yt = p.xt = mode == BWD_BILINEAR ? 2 : (mode == BWD_BICUBIC ? 4 : 1);
    if (mode == BWD_BICUBIC && IH == OH && IW == OW) {  // the forward is a copy then (bicubic2d_kernel), whatever the scales say
      p.kind_y = p.kind_x = BWD_NEAREST;
      p.sh = p.sw = 1.f;
    }
  }
  size_t f = 0;
  p.ytab_off = f;
  f += (size_t)OH * (p.yt + 2);
  p.xtab_off = f;
  f += (size_t)OW * (p.xt + 2);
  f = (f + 1) & ~(size_t)1;  // the int2 ranges start 8-byte aligned
  p.yr_off = f * sizeof(float);
  p.xr_off = p.yr_off + (size_t)IH * sizeof(int2);
  p.bytes = p.xr_off + (size_t)IW * sizeof(int2);
  p.ytab_off *= sizeof(float);
  p.xtab_off *= sizeof(float);
  return p;
}
}  // namespace

extern "C" size_t tvmi_upsample2d_backward_workspace_bytes(int mode, int antialias, int64_t IH, int64_t IW, int64_t OH,
                                                           int64_t OW, int align_corners, double scale_h, double scale_w) {
  if (OH <= 0 || OW <= 0 || IH <= 0 || IW <= 0 || mode < 0 || mode > 3) return 0;
  return bwd_plan(mode, antialias, IH, IW, OH, OW, align_corners, scale_h, scale_w).bytes;
}

extern "C" int tvmi_upsample2d_backward(const void* grad_output, void* grad_input, tvmi_dtype dt, int mode, int antialias,
                                        int64_t NC, int64_t IH, int64_t IW, int64_t OH, int64_t OW, int align_corners,
                                        double scale_h, double scale_w, void* workspace, size_t workspace_bytes,
                                        void* stream) {
  if (NC * IH * IW == 0) return 0;
  TVMI_CHECK_ARG(mode >= 0 && mode <= 3, "upsample2d_backward: mode must be 0 nearest, 1 nearest-exact, 2 bilinear, 3 bicubic");
  TVMI_CHECK_ARG(!antialias || mode >= 2, "upsample2d_backward: anti-aliasing is for the bilinear and bicubic modes");
  TVMI_CHECK_ARG(OH > 0 && OW > 0, "upsample2d_backward: output spatial size must be positive");
  TVMI_CHECK_ARG(grad_output && grad_input, "upsample2d_backward: null pointer");
  TVMI_CHECK_ARG(dt == TVMI_F32 || dt == TVMI_F16 || dt == TVMI_BF16, "upsample2d_backward: float32 / float16 / bfloat16 only");
  TVMI_CHECK_ARG(IH <= 65535 && IH * IW < (1ll << 31) && OH * OW < (1ll << 31), "upsample2d_backward: size too large");
  const BwdPlan p = bwd_plan(mode, antialias, IH, IW, OH, OW, align_corners, scale_h, scale_w);
  TVMI_CHECK_ARG(workspace && workspace_bytes >= p.bytes, "upsample2d_backward: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  float* ytab = reinterpret_cast<float*>(ws + p.ytab_off);
  float* xtab = reinterpret_cast<float*>(ws + p.xtab_off);
  int2* yr = reinterpret_cast<int2*>(ws + p.yr_off);
  int2* xr = reinterpret_cast<int2*>(ws + p.xr_off);
  const int al = (mode >= 2 && align_corners) ? 1 : 0;
  if (p.kind_y < 0) {
    aa_table_kernel<<<dim3((unsigned)ceil_div(OH, 128)), dim3(128), 0, s>>>(ytab, (int)OH, (int)IH, p.sh, -p.kind_y - 1, p.yt, al);
    aa_table_kernel<<<dim3((unsigned)ceil_div(OW, 128)), dim3(128), 0, s>>>(xtab, (int)OW, (int)IW, p.sw, -p.kind_x - 1, p.xt, al);
  } else {
    bwd_axis_table_kernel<<<dim3((unsigned)ceil_div(OH, 128)), dim3(128), 0, s>>>(ytab, (int)OH, (int)IH, p.sh, p.kind_y, p.yt, al);
    bwd_axis_table_kernel<<<dim3((unsigned)ceil_div(OW, 128)), dim3(128), 0, s>>>(xtab, (int)OW, (int)IW, p.sw, p.kind_x, p.xt, al);
  }
  bwd_axis_ranges_kernel<<<dim3((unsigned)ceil_div(IH, 128)), dim3(128), 0, s>>>(ytab, p.yt, (int)OH, (int)IH, yr);
  bwd_axis_ranges_kernel<<<dim3((unsigned)ceil_div(IW, 128)), dim3(128), 0, s>>>(xtab, p.xt, (int)OW, (int)IW, xr);
  const Launch L = plan(NC, IH, IW);
  // estimated longest range per axis: outputs per input pixel = filter footprint / scale (see upsample2d_bwd_vec_kernel)
  auto est = [&](float scale, int taps_of_mode) -> int {
    if (antialias && mode >= 2) return scale >= 1.f ? taps_of_mode + 1 : (int)std::ceil((double)(taps_of_mode + 1) / (double)scale);
    return scale > 0.f ? (int)std::min(4096.0, std::ceil((double)taps_of_mode / (double)scale)) : 4096;
  };
  const int foot = mode == BWD_BILINEAR ? 2 : (mode == BWD_BICUBIC ? 4 : 1);
  const int ex = est(p.sw, foot), ey = est(p.sh, foot);
#define TVMI_BWD_VEC(scalar_t, R, CC, P)                                                                                 \
  upsample2d_bwd_vec_kernel<scalar_t, R, CC, P><<<L.grid, dim3(kThreads), 0, s>>>(                                          \
      (const scalar_t*)grad_output, (scalar_t*)grad_input, ytab, xtab, yr, xr, (int)NC, (int)IH, (int)IW, (int)OH, (int)OW, p.yt, \
      p.xt, L.nc_per_block)
#define TVMI_BWD(scalar_t)                                                                                              \
  if (OW >= 4 && ex <= 2 && ey <= 2) TVMI_BWD_VEC(scalar_t, 2, 2, 8);                                                   \
  else if (OW >= 4 && ex <= 2) TVMI_BWD_VEC(scalar_t, 4, 2, 4);                                                         \
  else if (OW >= 4 && ey <= 2) TVMI_BWD_VEC(scalar_t, 2, 4, 4);                                                         \
  else if (OW >= 4) TVMI_BWD_VEC(scalar_t, 4, 4, 4);                                                                    \
  else                                                                                                                  \
    upsample2d_bwd_kernel<scalar_t><<<L.grid, dim3(kThreads), 0, s>>>((const scalar_t*)grad_output, (scalar_t*)grad_input, ytab, \
                                                                      xtab, yr, xr, (int)NC, (int)IH, (int)IW, (int)OH, (int)OW,  \
                                                                      p.yt, p.xt, L.nc_per_block)
  if (dt == TVMI_F32) { TVMI_BWD(float); }
  else if (dt == TVMI_F16) { TVMI_BWD(__half); }
  else { TVMI_BWD(__hip_bfloat16); }
#undef TVMI_BWD
#undef TVMI_BWD_VEC
  TVMI_RETURN_LAUNCH_STATUS("tvmi_upsample2d_backward");
}

extern "C" size_t tvmi_upsample2d_nhwc_workspace_bytes(int mode, int antialias, int64_t IH, int64_t IW, int64_t OH, int64_t OW,
                                                       int align_corners, double scale_h, double scale_w) {
  if (OH <= 0 || OW <= 0 || IH <= 0 || IW <= 0 || mode < 0 || mode > 3) return 0;
  return bwd_plan(mode, antialias, IH, IW, OH, OW, align_corners, scale_h, scale_w).yr_off;   // the two tables, no ranges
}

extern "C" int tvmi_upsample2d_nhwc(const void* input, void* output, tvmi_dtype dt, int mode, int antialias, int64_t N,
                                    int64_t C, int64_t IH, int64_t IW, int64_t OH, int64_t OW, int align_corners,
                                    double scale_h, double scale_w, void* workspace, size_t workspace_bytes, void* stream) {
  if (N * C * OH * OW == 0) return 0;
  TVMI_CHECK_ARG(mode >= 0 && mode <= 3, "upsample2d_nhwc: mode must be 0 nearest, 1 nearest-exact, 2 bilinear, 3 bicubic");
  TVMI_CHECK_ARG(!antialias || mode >= 2, "upsample2d_nhwc: anti-aliasing is for the bilinear and bicubic modes");
  TVMI_CHECK_ARG(input && output, "upsample2d_nhwc: null pointer");
  TVMI_CHECK_ARG(IH > 0 && IW > 0, "upsample2d_nhwc: input spatial size must be positive");
  TVMI_CHECK_ARG(dt == TVMI_F32 || dt == TVMI_F16 || dt == TVMI_BF16, "upsample2d_nhwc: float32 / float16 / bfloat16 only");
  TVMI_CHECK_ARG(OH <= 65535 && N <= 65535 && OW * C < (1ll << 31) && IH * IW * C < (1ll << 40), "upsample2d_nhwc: size too large");
  const BwdPlan p = bwd_plan(mode, antialias, IH, IW, OH, OW, align_corners, scale_h, scale_w);
  TVMI_CHECK_ARG(workspace && workspace_bytes >= p.yr_off, "upsample2d_nhwc: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  float* ytab = reinterpret_cast<float*>(ws + p.ytab_off);
  float* xtab = reinterpret_cast<float*>(ws + p.xtab_off);
  const int al = (mode >= 2 && align_corners) ? 1 : 0;
  if (p.kind_y < 0) {
    aa_table_kernel<<<dim3((unsigned)ceil_div(OH, 128)), dim3(128), 0, s>>>(ytab, (int)OH, (int)IH, p.sh, -p.kind_y - 1, p.yt, al);
    aa_table_kernel<<<dim3((unsigned)ceil_div(OW, 128)), dim3(128), 0, s>>>(xtab, (int)OW, (int)IW, p.sw, -p.kind_x - 1, p.xt, al);
  } else {
    bwd_axis_table_kernel<<<dim3((unsigned)ceil_div(OH, 128)), dim3(128), 0, s>>>(ytab, (int)OH, (int)IH, p.sh, p.kind_y, p.yt, al);
    bwd_axis_table_kernel<<<dim3((unsigned)ceil_div(OW, 128)), dim3(128), 0, s>>>(xtab, (int)OW, (int)IW, p.sw, p.kind_x, p.xt, al);
  }
  const int cpt = C % 4 == 0 ? 4 : (C == 3 ? 3 : 1);
  const int maxt = std::max(p.yt, p.xt);   // taps of the tables: 1 nearest, 2 bilinear, 4 bicubic; anti-aliased: by scale
  const dim3 grid((unsigned)ceil_div(OW * (C / cpt), kThreads), (unsigned)OH, (unsigned)N);
#define TVMI_NHWC_K(scalar_t, CPT, MAXT)                                                                                 \
  upsample2d_nhwc_kernel<scalar_t, CPT, MAXT><<<grid, dim3(kThreads), 0, s>>>((const scalar_t*)input, (scalar_t*)output, ytab, xtab, \
                                                                             (int)C, (int)IH, (int)IW, (int)OH, (int)OW, p.yt, p.xt)
#define TVMI_NHWC_T(scalar_t, CPT)                                                                                       \
  if (maxt <= 1) TVMI_NHWC_K(scalar_t, CPT, 1);                                                                         \
  else if (maxt <= 2) TVMI_NHWC_K(scalar_t, CPT, 2);                                                                    \
  else if (maxt <= 4) TVMI_NHWC_K(scalar_t, CPT, 4);                                                                    \
  else TVMI_NHWC_K(scalar_t, CPT, 0)
#define TVMI_NHWC(scalar_t)                                                                                              \
  if (cpt == 4) { TVMI_NHWC_T(scalar_t, 4); }                                                                           \
  else if (cpt == 3) { TVMI_NHWC_T(scalar_t, 3); }                                                                      \
  else { TVMI_NHWC_T(scalar_t, 1); }
  if (dt == TVMI_F32) { TVMI_NHWC(float); }
  else if (dt == TVMI_F16) { TVMI_NHWC(__half); }
  else { TVMI_NHWC(__hip_bfloat16); }
#undef TVMI_NHWC
#undef TVMI_NHWC_T
#undef TVMI_NHWC_K
  TVMI_RETURN_LAUNCH_STATUS("tvmi_upsample2d_nhwc");
}

extern "C" int tvmi_upsample_nearest2d_nhwc_any(const void* input, void* output, int64_t elem_bytes, int64_t N, int64_t C,
                                                int64_t IH, int64_t IW, int64_t OH, int64_t OW, int exact, double scale_h,
                                                double scale_w, void* stream) {
  if (N * C * OH * OW == 0) return 0;
  TVMI_CHECK_ARG(input && output, "upsample_nearest2d_nhwc_any: null pointer");
  TVMI_CHECK_ARG(IH > 0 && IW > 0, "upsample_nearest2d_nhwc_any: input spatial size must be positive");
  TVMI_CHECK_ARG(OH <= 65535 && N <= 65535 && OW * C < (1ll << 31), "upsample_nearest2d_nhwc_any: size too large");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float sh = compute_scale(IH, OH, false, scale_h), sw = compute_scale(IW, OW, false, scale_w);
  const dim3 grid((unsigned)ceil_div(OW * C, kThreads), (unsigned)OH, (unsigned)N);
#define TVMI_NEAREST_NHWC(U)                                                                                           \
  nearest2d_nhwc_bytes_kernel<U><<<grid, dim3(kThreads), 0, s>>>((const U*)input, (U*)output, (int)C, (int)IH, (int)IW, (int)OH, \
                                                                (int)OW, sh, sw, exact)
  switch (elem_bytes) {
    case 1: TVMI_NEAREST_NHWC(unsigned char); break;
    case 2: TVMI_NEAREST_NHWC(unsigned short); break;
    case 4: TVMI_NEAREST_NHWC(unsigned int); break;
    case 8: TVMI_NEAREST_NHWC(unsigned long long); break;
    default: return ::tvmi::set_error(hipErrorInvalidValue, "upsample_nearest2d_nhwc_any: element size must be 1, 2, 4 or 8 bytes");
  }
#undef TVMI_NEAREST_NHWC
  TVMI_RETURN_LAUNCH_STATUS("tvmi_upsample_nearest2d_nhwc_any");
}

extern "C" int tvmi_upsample_nearest2d_any(const void* input, void* output, int64_t elem_bytes, int64_t NC, int64_t IH,
                                           int64_t IW, int64_t OH, int64_t OW, int exact, double scale_h, double scale_w,
                                           void* stream) {
  TVMI_RESIZE_PROLOGUE("upsample_nearest2d_any");
  const float sh = compute_scale(IH, OH, false, scale_h), sw = compute_scale(IW, OW, false, scale_w);
#define TVMI_NEAREST_ANY(U)                                                                                            \
  nearest2d_bytes_kernel<U><<<L.grid, dim3(kThreads), 0, s>>>((const U*)input, (U*)output, (int)NC, (int)IH, (int)IW, (int)OH, \
                                                              (int)OW, sh, sw, exact, L.nc_per_block)
  switch (elem_bytes) {
    case 1: TVMI_NEAREST_ANY(unsigned char); break;
    case 2: TVMI_NEAREST_ANY(unsigned short); break;
    case 4: TVMI_NEAREST_ANY(unsigned int); break;
    case 8: TVMI_NEAREST_ANY(unsigned long long); break;
    default: return ::tvmi::set_error(hipErrorInvalidValue, "upsample_nearest2d_any: element size must be 1, 2, 4 or 8 bytes");
  }
#undef TVMI_NEAREST_ANY
  TVMI_RETURN_LAUNCH_STATUS("tvmi_upsample_nearest2d_any");
}

extern "C" int tvmi_paste_masks(const void* masks, const float* boxes, void* output, tvmi_dtype dt, int64_t N,
                                int64_t M, int64_t im_h, int64_t im_w, int64_t padding, void* stream) {
  if (N * im_h * im_w == 0) return 0;
  TVMI_CHECK_ARG(masks && boxes && output, "paste_masks: null pointer");
  TVMI_CHECK_ARG(M > 0 && padding >= 0 && M + 2 * padding < 32768, "paste_masks: mask size out of range");
  TVMI_CHECK_ARG(N <= 65535 && im_h * im_w < (1ll << 31), "paste_masks: size too large");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float scale = (float)((double)(M + 2 * padding) / (double)M);
  // 16-byte packs when every canvas starts 16-byte aligned, else scalar stores
#define TVMI_PASTE(VEC)                                                                                        \
  paste_masks_kernel<scalar_t, VEC><<<dim3((unsigned)std::min<int64_t>(ceil_div(im_h * im_w / (VEC), kThreads),  \
                                                                      std::max<int64_t>(1, 32768 / N)),        \
                                           (unsigned)N),                                                       \
                                      dim3(kThreads), 0, s>>>((const scalar_t*)masks, boxes, (scalar_t*)output, (int)M, \
                                                              (int)im_h, (int)im_w, (int)padding, scale)
  TVMI_DISPATCH_FLOAT(dt, "paste_masks", {
    constexpr int V = 16 / (int)sizeof(scalar_t);
    if ((im_h * im_w) % V == 0 && reinterpret_cast<uintptr_t>(output) % 16 == 0)
      TVMI_PASTE(V);
    else
      TVMI_PASTE(1);
  });
#undef TVMI_PASTE
  TVMI_RETURN_LAUNCH_STATUS("tvmi_paste_masks");
}

extern "C" int tvmi_normalize_resize_batch(const void* const* images, const int64_t* heights, const int64_t* widths,
                                           const int64_t* out_heights, const int64_t* out_widths, int64_t num_images,
                                           int64_t channels, const float* mean, const float* stdv, void* output,
                                           tvmi_dtype dt, int64_t padded_h, int64_t padded_w, void* stream) {
  TVMI_CHECK_ARG(num_images >= 0 && num_images <= tvmi::kXformMaxImages, "normalize_resize_batch: at most 64 images per call");
  if (num_images * channels * padded_h * padded_w == 0) return 0;
  TVMI_CHECK_ARG(images && heights && widths && out_heights && out_widths && mean && stdv && output,
                 "normalize_resize_batch: null pointer");
  TVMI_CHECK_ARG(channels >= 1 && channels <= 4, "normalize_resize_batch: 1..4 channels");
  TVMI_CHECK_ARG(padded_h <= 65535 && padded_h * padded_w < (1ll << 31), "normalize_resize_batch: size too large");
  tvmi::XformImages im;
  for (int i = 0; i < tvmi::kXformMaxImages; ++i) {
    const int j = i < num_images ? i : 0;
    TVMI_CHECK_ARG(images[j] && heights[j] > 0 && widths[j] > 0 && heights[j] * widths[j] < (1ll << 31) &&
                       out_heights[j] > 0 && out_widths[j] > 0 && out_heights[j] <= padded_h && out_widths[j] <= padded_w,
                   "normalize_resize_batch: bad image geometry");
    im.ptr[i] = images[j];
    im.H[i] = (int)heights[j];
    im.W[i] = (int)widths[j];
    im.OH[i] = (int)out_heights[j];
    im.OW[i] = (int)out_widths[j];
  }
  for (int c = 0; c < 4; ++c) {
    im.mean[c] = c < channels ? mean[c] : 0.f;
    im.stdv[c] = c < channels ? stdv[c] : 1.f;
  }
  im.C = (int)channels;
  const dim3 grid((unsigned)ceil_div(padded_w, kThreads), (unsigned)padded_h, (unsigned)num_images);
  TVMI_DISPATCH_FLOAT(dt, "normalize_resize_batch",
                      normalize_resize_batch_kernel<scalar_t><<<grid, dim3(kThreads), 0, static_cast<hipStream_t>(stream)>>>(
                          im, static_cast<scalar_t*>(output), (int)padded_h, (int)padded_w));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_normalize_resize_batch");
}
