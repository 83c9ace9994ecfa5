// quantized.hip — the two "quantized" operators of the reference: CPU-only there, integer tensors with
// explicit (scale, zero point) arguments in this version of the reference.
//   qroi_align : torchvision/csrc/ops/quantized/cpu/qroi_align_kernel.cpp:22-178 — RoIs dequantised on load, the bilinear sums
//                taken on the RAW integer pixels (`output_val += w1 * v1 + ...`, `sum_w += w1 + w2 + w3 + w4`), dequantised once
//                (`scale * (output_val - zero_point * sum_w)`), averaged, re-quantised with round-half-even (std::nearbyint)
//                and saturated to the integer type.  Batch index is always 0 (the reference accepts one image only, :206-207).
//   qnms       : quantized/cpu/qnms_kernel.cpp:22-120 evaluates the boxes as float32 with exactly the arithmetic of
//                cpu/nms_kernel.cpp (the scale cancels in the IoU), so the dispatcher glue widens the boxes to fp32 and calls
//                tvmi_nms_blocking with the stable descending order of the INTEGER scores — no kernel of its own.
// One lane per pooled output, pw fastest; sample arithmetic shared with the float kernels (roi_common.h: axis_sample), the
// same operations in the same order as the reference (this TU is built with -ffp-contract=off), so results are bit-identical.
#include <limits>

#include "roi_common.h"

namespace tvmi {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void qroi_align_kernel(const T* __restrict__ input, const T* __restrict__ rois, T* __restrict__ output,
                                                         int64_t total, int C, int H, int W, int PH, int PW, float in_scale,
                                                         int64_t in_zp, float roi_scale, int64_t roi_zp, float spatial_scale,
                                                         int sr, int aligned) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / ((int64_t)PW * PH)) % C);
    const int64_t k = idx / ((int64_t)PW * PH * C);
    const T* r = rois + k * 5;
    // dequantize_val: (float(value) - zero_point) * scale  (int64 zero point converted to float by the subtraction)
    const float offset = aligned ? 0.5f : 0.f;
    const float sw = ((float)r[1] - (float)roi_zp) * roi_scale * spatial_scale - offset;
    const float sh = ((float)r[2] - (float)roi_zp) * roi_scale * spatial_scale - offset;
    const float ew = ((float)r[3] - (float)roi_zp) * roi_scale * spatial_scale - offset;
    const float eh = ((float)r[4] - (float)roi_zp) * roi_scale * spatial_scale - offset;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) {
      rw = fmaxf(rw, 1.f);
      rh = fmaxf(rh, 1.f);
    }
    const float bin_h = rh / (float)PH, bin_w = rw / (float)PW;
    const int gh = sr > 0 ? sr : (int)ceilf(rh / (float)PH);
    const int gw = sr > 0 ? sr : (int)ceilf(rw / (float)PW);
    const float count = (float)max(gh * gw, 1);
    const T* plane = input + (int64_t)c * H * W;   // roi_batch_ind = 0 (:58-59)
    float out_val = 0.f, sum_w = 0.f;
    for (int iy = 0; iy < gh; ++iy) {
      int ylo, yhi;
      float ly, hy;
      const bool vy = axis_sample<float>(H, sh, bin_h, gh, ph, iy, ylo, yhi, ly, hy);
      for (int ix = 0; ix < gw; ++ix) {
        int xlo, xhi;
        float lx, hx;
        const bool vx = axis_sample<float>(W, sw, bin_w, gw, pw, ix, xlo, xhi, lx, hx);
        if (!(vy && vx)) continue;   // the reference's pre_calc holds zero weights and position 0 here: adds exact zeros
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        out_val += w1 * (float)plane[(int64_t)ylo * W + xlo] + w2 * (float)plane[(int64_t)ylo * W + xhi] +
                   w3 * (float)plane[(int64_t)yhi * W + xlo] + w4 * (float)plane[(int64_t)yhi * W + xhi];
        sum_w += w1 + w2 + w3 + w4;
      }
    }
    out_val = in_scale * (out_val - (float)in_zp * sum_w);
    out_val /= count;
    const float inv_scale = 1.0f / in_scale;
    // int64(zero_point + nearbyint(x)): the sum is formed in float (the int64 zero point is converted), then truncated
    const float q = (float)in_zp + nearbyintf(out_val * inv_scale);
    int64_t qval = (int64_t)q;
    const int64_t qmin = (int64_t)std::numeric_limits<T>::min(), qmax = (int64_t)std::numeric_limits<T>::max();
    qval = qval < qmin ? qmin : qval;
    qval = qval > qmax ? qmax : qval;
    output[idx] = (T)qval;
  }
}

template <typename T>
int launch_q(const void* input, const void* rois, void* output, int64_t C, int64_t H, int64_t W, int64_t K, int64_t PH, int64_t PW,
             double in_scale, int64_t in_zp, double roi_scale, int64_t roi_zp, double spatial_scale, int64_t sr, int aligned,
             hipStream_t stream) {
  const int64_t total = K * C * PH * PW;
  const int64_t blocks = std::min<int64_t>(ceil_div(total, (int64_t)256), 1 << 20);
  qroi_align_kernel<T><<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(
      static_cast<const T*>(input), static_cast<const T*>(rois), static_cast<T*>(output), total, (int)C, (int)H, (int)W, (int)PH,
      (int)PW, (float)in_scale, in_zp, (float)roi_scale, roi_zp, (float)spatial_scale, (int)sr, aligned);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_qroi_align_forward");
}

}  // namespace
}  // namespace tvmi

extern "C" int tvmi_qroi_align_forward(const void* input, const void* rois, void* output, tvmi_int_dtype dt, int64_t C, int64_t H,
                                       int64_t W, int64_t K, int64_t pooled_h, int64_t pooled_w, double input_scale,
                                       int64_t input_zero_point, double rois_scale, int64_t rois_zero_point, double spatial_scale,
                                       int64_t sampling_ratio, int aligned, void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0 && C >= 0 && H >= 0 && W >= 0 && K >= 0, "qroi_align: bad sizes");
  if (K * C * pooled_h * pooled_w == 0) return 0;
  TVMI_CHECK_ARG(input && rois && output, "qroi_align: null pointer");
  TVMI_CHECK_ARG(H * W < (1ll << 31), "qroi_align: plane exceeds 32-bit indexing");
  hipStream_t s = static_cast<hipStream_t>(stream);
#define TVMI_Q(T_) return tvmi::launch_q<T_>(input, rois, output, C, H, W, K, pooled_h, pooled_w, input_scale, input_zero_point, \
                                             rois_scale, rois_zero_point, spatial_scale, sampling_ratio, aligned, s)
  switch (dt) {
    case TVMI_U8: TVMI_Q(uint8_t);
    case TVMI_I8: TVMI_Q(int8_t);
    case TVMI_I16: TVMI_Q(int16_t);
    case TVMI_I32: TVMI_Q(int32_t);
    case TVMI_I64: TVMI_Q(int64_t);
  }
#undef TVMI_Q
  return tvmi::set_error((int)hipErrorInvalidValue, "qroi_align: unsupported integer dtype");
}
