// roi_pool.hip — RoIPool, PSRoIAlign, PSRoIPool (forward + backward) for gfx950.
//
// Semantics (incl. the reference's quirks, kept on purpose):
//   roi_pool     torchvision/csrc/ops/cpu/roi_pool_kernel.cpp:24-134  (round()ed RoI, +1
//                inclusive size, bins [floor(p*bin), ceil((p+1)*bin)) clipped to [0,H],
//                max initialised with -FLT_MAX, empty bin -> 0 / argmax -1)
//   ps_roi_align cpu/ps_roi_align_kernel.cpp:17-151,219-313 (always -0.5 shift, no >=1
//                clamp, count = gh*gw with no max(.,1), c_in = (c_out*PH+ph)*PW+pw)
//   ps_roi_pool  cpu/ps_roi_pool_kernel.cpp:22-155 (size = max(end-start,1) without +1;
//                forward clips bins to H-1/W-1, backward clips to H/W; backward rounds the
//                RoI with roundf)
// Work decomposition: one lane per pooled output element with `pw` fastest, so a wave
// covers one or more complete pooled rows of one (roi, channel) — output / argmax /
// channel_mapping stores are contiguous, and the lanes of a wave read neighbouring input
// bins of the same plane (shared cache lines).  Backward kernels scatter with hardware
// float atomics (`alertNotDeterministic` is raised by the dispatcher glue).
#include <float.h>

#include <algorithm>
#include <type_traits>

#include "tvmi_common.h"

namespace tvmi {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <typename T>
__global__ __launch_bounds__(kThreads) void roi_pool_fwd(const T* __restrict__ input,
                                                         const T* __restrict__ rois,
                                                         T* __restrict__ output, int* __restrict__ argmax,
                                                         int64_t total, int C, int H, int W, int PH,
                                                         int PW, double spatial_scale) {
  using A = typename Acc<T>::type;
  const A scale = (A)spatial_scale;
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * kThreads) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / ((int64_t)PW * PH)) % C);
    const int64_t n = idx / ((int64_t)PW * PH * C);
    const T* roi = rois + n * 5;
    const int b = (int)ld(roi);
    const int rsw = (int)round(ld(roi + 1) * scale);
    const int rsh = (int)round(ld(roi + 2) * scale);
    const int rew = (int)round(ld(roi + 3) * scale);
    const int reh = (int)round(ld(roi + 4) * scale);
    const int rw = max(rew - rsw + 1, 1), rh = max(reh - rsh + 1, 1);
    const A bin_h = (A)rh / (A)PH, bin_w = (A)rw / (A)PW;
    int hs = (int)floor((A)ph * bin_h), ws = (int)floor((A)pw * bin_w);
    int he = (int)ceil((A)(ph + 1) * bin_h), we = (int)ceil((A)(pw + 1) * bin_w);
    hs = clampi(hs + rsh, 0, H);
    he = clampi(he + rsh, 0, H);
    ws = clampi(ws + rsw, 0, W);
    we = clampi(we + rsw, 0, W);
    const bool empty = (he <= hs) || (we <= ws);
    A maxval = empty ? (A)0 : (A)-FLT_MAX;
    int maxidx = -1;
    const T* plane = input + ((int64_t)b * C + c) * H * W;
    for (int h = hs; h < he; ++h) {
      for (int w = ws; w < we; ++w) {
        const A v = ld(plane + h * W + w);
        if (v > maxval) {
          maxval = v;
          maxidx = h * W + w;
        }
      }
    }
    st(output + idx, maxval);
    argmax[idx] = maxidx;
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void roi_pool_bwd(const T* __restrict__ grad,
                                                         const T* __restrict__ rois,
                                                         const int* __restrict__ argmax,
                                                         T* __restrict__ grad_input, int64_t total, int C,
                                                         int H, int W, int PH, int PW, int64_t ns,
                                                         int64_t cs, int64_t hs, int64_t ws) {
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * kThreads) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / ((int64_t)PW * PH)) % C);
    const int64_t n = idx / ((int64_t)PW * PH * C);
    const int am = argmax[idx];
    if (am == -1) continue;
    const int b = (int)ld(rois + n * 5);
    atomic_accum(grad_input + ((int64_t)b * C + c) * H * W + am, ld(grad + n * ns + c * cs + ph * hs + pw * ws));
  }
}

// ---- bilinear helpers of the PS-RoIAlign CPU kernel (cpu/ps_roi_align_kernel.cpp:17-70,153-217)
template <typename A>
__device__ __forceinline__ bool bilinear_setup(int H, int W, A y, A x, int& yl, int& yh, int& xl, int& xh,
                                               A& w1, A& w2, A& w3, A& w4) {
  if (y < (A)-1.0 || y > (A)H || x < (A)-1.0 || x > (A)W) return false;
  if (y <= (A)0) y = (A)0;
  if (x <= (A)0) x = (A)0;
  yl = (int)y;
  xl = (int)x;
  if (yl >= H - 1) {
    yh = yl = H - 1;
    y = (A)yl;
  } else {
    yh = yl + 1;
  }
  if (xl >= W - 1) {
    xh = xl = W - 1;
    x = (A)xl;
  } else {
    xh = xl + 1;
  }
  const A ly = y - (A)yl, lx = x - (A)xl;
  const A hy = (A)1. - ly, hx = (A)1. - lx;
  w1 = hy * hx;
  w2 = hy * lx;
  w3 = ly * hx;
  w4 = ly * lx;
  return true;
}

template <typename T, bool kBackward>
__global__ __launch_bounds__(kThreads) void ps_roi_align_kernel(
    const T* __restrict__ data /* input (fwd) | grad_output (bwd) */, const T* __restrict__ rois,
    T* __restrict__ out /* output (fwd) | grad_input (bwd) */, int* __restrict__ channel_mapping,
    int64_t total, int C, int H, int W, int PH, int PW, int C_out, double spatial_scale, int sr) {
  using A = typename Acc<T>::type;
  const A scale = (A)spatial_scale;
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * kThreads) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c_out = (int)((idx / ((int64_t)PW * PH)) % C_out);
    const int64_t n = idx / ((int64_t)PW * PH * C_out);
    const T* roi = rois + n * 5;
    const int b = (int)ld(roi);
    const A rsw = ld(roi + 1) * scale - (A)0.5;
    const A rsh = ld(roi + 2) * scale - (A)0.5;
    const A rew = ld(roi + 3) * scale - (A)0.5;
    const A reh = ld(roi + 4) * scale - (A)0.5;
    const A rw = rew - rsw, rh = reh - rsh;
    const A bin_h = rh / (A)PH, bin_w = rw / (A)PW;
    const int c_in = kBackward ? channel_mapping[idx] : (c_out * PH + ph) * PW + pw;
    const A hstart = (A)ph * bin_h + rsh;
    const A wstart = (A)pw * bin_w + rsw;
    const int gh = sr > 0 ? sr : (int)ceil(rh / (A)PH);
    const int gw = sr > 0 ? sr : (int)ceil(rw / (A)PW);
    const A count = (A)(gh * gw);
    const int64_t plane_off = ((int64_t)b * C + c_in) * H * W;
    A acc = (A)0;
    A go = (A)0;
    if (kBackward) go = ld(data + idx);
    for (int iy = 0; iy < gh; ++iy) {
      const A y = hstart + (A)((float)iy + .5f) * bin_h / (A)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const A x = wstart + (A)((float)ix + .5f) * bin_w / (A)gw;
        int yl, yh, xl, xh;
        A w1, w2, w3, w4;
        if (!bilinear_setup<A>(H, W, y, x, yl, yh, xl, xh, w1, w2, w3, w4)) continue;
        if (kBackward) {
          T* gi = out + plane_off;
          atomic_accum(gi + yl * W + xl, go * w1 / count);
          atomic_accum(gi + yl * W + xh, go * w2 / count);
          atomic_accum(gi + yh * W + xl, go * w3 / count);
          atomic_accum(gi + yh * W + xh, go * w4 / count);
        } else {
          const T* p = data + plane_off;
          const A v1 = ld(p + yl * W + xl), v2 = ld(p + yl * W + xh);
          const A v3 = ld(p + yh * W + xl), v4 = ld(p + yh * W + xh);
          acc += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
        }
      }
    }
    if (!kBackward) {
      st(out + idx, acc / count);
      channel_mapping[idx] = c_in;
    }
  }
}

template <typename T, bool kBackward>
__global__ __launch_bounds__(kThreads) void ps_roi_pool_kernel(
    const T* __restrict__ data, const T* __restrict__ rois, T* __restrict__ out,
    int* __restrict__ channel_mapping, int64_t total, int C, int H, int W, int PH, int PW, int C_out,
    double spatial_scale) {
  using A = typename Acc<T>::type;
  const A scale = (A)spatial_scale;
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * kThreads) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c_out = (int)((idx / ((int64_t)PW * PH)) % C_out);
    const int64_t n = idx / ((int64_t)PW * PH * C_out);
    const T* roi = rois + n * 5;
    const int b = (int)ld(roi);
    int rsw, rsh, rew, reh;
    if (kBackward) {  // cpu/ps_roi_pool_kernel.cpp:109-112 uses roundf
      rsw = (int)roundf((float)(ld(roi + 1) * scale));
      rsh = (int)roundf((float)(ld(roi + 2) * scale));
      rew = (int)roundf((float)(ld(roi + 3) * scale));
      reh = (int)roundf((float)(ld(roi + 4) * scale));
    } else {
      rsw = (int)round(ld(roi + 1) * scale);
      rsh = (int)round(ld(roi + 2) * scale);
      rew = (int)round(ld(roi + 3) * scale);
      reh = (int)round(ld(roi + 4) * scale);
    }
    const int rw = max(rew - rsw, 1), rh = max(reh - rsh, 1);
    const A bin_h = (A)rh / (A)PH, bin_w = (A)rw / (A)PW;
    int hs = (int)floor((A)ph * bin_h), ws = (int)floor((A)pw * bin_w);
    int he = (int)ceil((A)(ph + 1) * bin_h), we = (int)ceil((A)(pw + 1) * bin_w);
    const int hmax = kBackward ? H : H - 1, wmax = kBackward ? W : W - 1;
    hs = clampi(hs + rsh, 0, hmax);
    he = clampi(he + rsh, 0, hmax);
    ws = clampi(ws + rsw, 0, wmax);
    we = clampi(we + rsw, 0, wmax);
    const bool empty = (he <= hs) || (we <= ws);
    const int c_in = kBackward ? channel_mapping[idx] : (c_out * PH + ph) * PW + pw;
    const int64_t plane_off = ((int64_t)b * C + c_in) * H * W;
    const A bin_area = (A)((he - hs) * (we - ws));
    if (kBackward) {
      const A diff = empty ? (A)0 : ld(data + idx) / bin_area;
      T* gi = out + plane_off;
      for (int h = hs; h < he; ++h)
        for (int w = ws; w < we; ++w) atomic_accum(gi + h * W + w, diff);
    } else {
      const T* p = data + plane_off;
      A sum = (A)0;
      for (int h = hs; h < he; ++h)
        for (int w = ws; w < we; ++w) sum += ld(p + h * W + w);
      st(out + idx, empty ? (A)0 : sum / bin_area);
      channel_mapping[idx] = c_in;
    }
  }
}

inline dim3 grid_for(int64_t total) {
  return dim3((unsigned)std::min<int64_t>(ceil_div(total, kThreads), 1 << 20));
}

}  // namespace
}  // namespace tvmi

using tvmi::grid_for;
using tvmi::kThreads;

extern "C" int tvmi_roi_pool_forward(const void* input, const void* rois, void* output, int32_t* argmax,
                                     tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                                     int64_t pooled_h, int64_t pooled_w, double spatial_scale,
                                     void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "roi_pool: pooled size must be positive");
  const int64_t total = K * C * pooled_h * pooled_w;
  if (total == 0) return 0;
  TVMI_CHECK_ARG(input && rois && output && argmax, "roi_pool: null pointer");
  TVMI_CHECK_ARG(H * W < (1ll << 31), "roi_pool: plane too large");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "roi_pool_forward",
                      tvmi::roi_pool_fwd<scalar_t><<<grid_for(total), dim3(kThreads), 0, s>>>(
                          (const scalar_t*)input, (const scalar_t*)rois, (scalar_t*)output, argmax, total,
                          (int)C, (int)H, (int)W, (int)pooled_h, (int)pooled_w, spatial_scale));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_roi_pool_forward");
}

extern "C" int tvmi_roi_pool_backward(const void* grad, const void* rois, const int32_t* argmax,
                                      void* grad_input, tvmi_dtype dt, int64_t N, int64_t C, int64_t H,
                                      int64_t W, int64_t K, int64_t pooled_h, int64_t pooled_w,
                                      int64_t n_stride, int64_t c_stride, int64_t h_stride,
                                      int64_t w_stride, void* stream) {
  const int64_t total = K * C * pooled_h * pooled_w;
  if (total == 0 || N * C * H * W == 0) return 0;
  TVMI_CHECK_ARG(grad && rois && argmax && grad_input, "roi_pool_backward: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "roi_pool_backward",
                      tvmi::roi_pool_bwd<scalar_t><<<grid_for(total), dim3(kThreads), 0, s>>>(
                          (const scalar_t*)grad, (const scalar_t*)rois, argmax, (scalar_t*)grad_input, total,
                          (int)C, (int)H, (int)W, (int)pooled_h, (int)pooled_w, n_stride, c_stride, h_stride,
                          w_stride));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_roi_pool_backward");
}

extern "C" int tvmi_ps_roi_align_forward(const void* input, const void* rois, void* output,
                                         int32_t* channel_mapping, tvmi_dtype dt, int64_t N, int64_t C,
                                         int64_t H, int64_t W, int64_t K, int64_t pooled_h,
                                         int64_t pooled_w, double spatial_scale, int64_t sampling_ratio,
                                         void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "ps_roi_align: pooled size must be positive");
  TVMI_CHECK_ARG(C % (pooled_h * pooled_w) == 0,
                 "input channels must be a multiple of pooling height * pooling width");
  const int64_t C_out = C / (pooled_h * pooled_w);
  const int64_t total = K * C_out * pooled_h * pooled_w;
  if (total == 0) return 0;
  TVMI_CHECK_ARG(input && rois && output && channel_mapping, "ps_roi_align: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "ps_roi_align_forward",
                      (tvmi::ps_roi_align_kernel<scalar_t, false><<<grid_for(total), dim3(kThreads), 0, s>>>(
                          (const scalar_t*)input, (const scalar_t*)rois, (scalar_t*)output, channel_mapping,
                          total, (int)C, (int)H, (int)W, (int)pooled_h, (int)pooled_w, (int)C_out,
                          spatial_scale, (int)sampling_ratio)));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_ps_roi_align_forward");
}

extern "C" int tvmi_ps_roi_align_backward(const void* grad, const void* rois,
                                          const int32_t* channel_mapping, void* grad_input, tvmi_dtype dt,
                                          int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                                          int64_t pooled_h, int64_t pooled_w, double spatial_scale,
                                          int64_t sampling_ratio, void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "ps_roi_align_backward: pooled size must be positive");
  const int64_t C_out = C / (pooled_h * pooled_w);
  const int64_t total = K * C_out * pooled_h * pooled_w;
  if (total == 0 || N * C * H * W == 0) return 0;
  TVMI_CHECK_ARG(grad && rois && channel_mapping && grad_input, "ps_roi_align_backward: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "ps_roi_align_backward",
                      (tvmi::ps_roi_align_kernel<scalar_t, true><<<grid_for(total), dim3(kThreads), 0, s>>>(
                          (const scalar_t*)grad, (const scalar_t*)rois, (scalar_t*)grad_input,
                          const_cast<int32_t*>(channel_mapping), total, (int)C, (int)H, (int)W, (int)pooled_h,
                          (int)pooled_w, (int)C_out, spatial_scale, (int)sampling_ratio)));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_ps_roi_align_backward");
}

extern "C" int tvmi_ps_roi_pool_forward(const void* input, const void* rois, void* output,
                                        int32_t* channel_mapping, tvmi_dtype dt, int64_t N, int64_t C,
                                        int64_t H, int64_t W, int64_t K, int64_t pooled_h, int64_t pooled_w,
                                        double spatial_scale, void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "ps_roi_pool: pooled size must be positive");
  TVMI_CHECK_ARG(C % (pooled_h * pooled_w) == 0,
                 "input channels must be a multiple of pooling height * pooling width");
  const int64_t C_out = C / (pooled_h * pooled_w);
  const int64_t total = K * C_out * pooled_h * pooled_w;
  if (total == 0) return 0;
  TVMI_CHECK_ARG(input && rois && output && channel_mapping, "ps_roi_pool: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "ps_roi_pool_forward",
                      (tvmi::ps_roi_pool_kernel<scalar_t, false><<<grid_for(total), dim3(kThreads), 0, s>>>(
                          (const scalar_t*)input, (const scalar_t*)rois, (scalar_t*)output, channel_mapping,
                          total, (int)C, (int)H, (int)W, (int)pooled_h, (int)pooled_w, (int)C_out,
                          spatial_scale)));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_ps_roi_pool_forward");
}

extern "C" int tvmi_ps_roi_pool_backward(const void* grad, const void* rois, const int32_t* channel_mapping,
                                         void* grad_input, tvmi_dtype dt, int64_t N, int64_t C, int64_t H,
                                         int64_t W, int64_t K, int64_t pooled_h, int64_t pooled_w,
                                         double spatial_scale, void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "ps_roi_pool_backward: pooled size must be positive");
  const int64_t C_out = C / (pooled_h * pooled_w);
  const int64_t total = K * C_out * pooled_h * pooled_w;
  if (total == 0 || N * C * H * W == 0) return 0;
  TVMI_CHECK_ARG(grad && rois && channel_mapping && grad_input, "ps_roi_pool_backward: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "ps_roi_pool_backward",
                      (tvmi::ps_roi_pool_kernel<scalar_t, true><<<grid_for(total), dim3(kThreads), 0, s>>>(
                          (const scalar_t*)grad, (const scalar_t*)rois, (scalar_t*)grad_input,
                          const_cast<int32_t*>(channel_mapping), total, (int)C, (int)H, (int)W, (int)pooled_h,
                          (int)pooled_w, (int)C_out, spatial_scale)));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_ps_roi_pool_backward");
}
