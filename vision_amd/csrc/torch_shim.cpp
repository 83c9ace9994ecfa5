// torch_shim.cpp — dispatcher glue: binds the C-ABI launchers of libtvmi_kernels.so
// (include/tvmi.h) to the reference's `torchvision::` operator schemas on the CUDA (= HIP on
// PyTorch-ROCm) dispatch key.  Built with g++ against torch headers only — no HIP headers,
// no kernels; the only things that cross into the kernels library are raw device pointers,
// sizes and the current hipStream_t.
//
// Schemas are the verbatim strings of the reference:
//   torchvision/csrc/ops/nms.cpp:27, roi_align.cpp:75,77, roi_pool.cpp:68,70,
//   ps_roi_align.cpp:75,77, ps_roi_pool.cpp:71,73, deform_conv2d.cpp:102,104,
//   box_iou_rotated.cpp:31, quantized/cpu/qnms_kernel.cpp:149,
//   quantized/cpu/qroi_align_kernel.cpp:236, vision.cpp:31
// so that torchvision/_meta_registrations.py, _autograd_registrations.py and every
// torchvision.ops wrapper bind to this library unchanged.
// Define TVMI_NO_SCHEMA_DEFS when another library in the process already owns the m.def()s.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <torch/csrc/inductor/aoti_torch/c/shim.h>
#include <torch/library.h>

#include <tuple>

#include "../../include/tvmi.h"

namespace tvmi_shim {
namespace {

void* current_stream(const at::Tensor& t) {
  void* s = nullptr;
  TORCH_CHECK(aoti_torch_get_current_cuda_stream(t.get_device(), &s) == 0,
              "tvmi: cannot query the current HIP stream");
  return s;
}

tvmi_dtype dtype_of(const at::Tensor& t, const char* op) {
  switch (t.scalar_type()) {
    case at::kFloat:
      return TVMI_F32;
    case at::kDouble:
      return TVMI_F64;
    case at::kHalf:
      return TVMI_F16;
    case at::kBFloat16:
      return TVMI_BF16;
    default:
      TORCH_CHECK(false, op, ": unsupported dtype ", t.scalar_type());
  }
}

void check_status(int status, const char* op) {
  TORCH_CHECK(status == 0, op, " failed: ", tvmi_last_error());
}

// ---- nms: cuda/nms_kernel.cu:166-258 (checks and messages), cpu/nms_kernel.cpp (semantics)
at::Tensor nms_segmented(const at::Tensor& dets, const at::Tensor& scores,
                         const c10::optional<at::Tensor>& seg, double iou_threshold) {
  TORCH_CHECK(dets.is_cuda(), "dets must be a CUDA tensor");
  TORCH_CHECK(scores.is_cuda(), "scores must be a CUDA tensor");
  TORCH_CHECK(dets.dim() == 2, "boxes should be a 2d tensor, got ", dets.dim(), "D");
  TORCH_CHECK(dets.size(1) == 4, "boxes should have 4 elements in dimension 1, got ", dets.size(1));
  TORCH_CHECK(scores.dim() == 1, "scores should be a 1d tensor, got ", scores.dim(), "D");
  TORCH_CHECK(dets.size(0) == scores.size(0),
              "boxes and scores should have same number of elements in ", "dimension 0, got ",
              dets.size(0), " and ", scores.size(0));
  c10::DeviceGuard guard(dets.device());
  const int64_t n = dets.size(0);
  if (dets.numel() == 0) return at::empty({0}, dets.options().dtype(at::kLong));

  // Half/BFloat16 boxes are evaluated in fp32, as cuda/nms_kernel.cu:32-53 does for Half.
  at::Tensor boxes = dets;
  if (dets.scalar_type() == at::kHalf || dets.scalar_type() == at::kBFloat16) boxes = dets.to(at::kFloat);
  TORCH_CHECK(boxes.scalar_type() == at::kFloat || boxes.scalar_type() == at::kDouble,
              "nms: boxes must be a floating point tensor");
  boxes = boxes.contiguous();
  at::Tensor order = std::get<1>(at::sort(scores, /*stable=*/true, /*dim=*/0, /*descending=*/true));
  at::Tensor seg_c;
  const int64_t* seg_ptr = nullptr;
  if (seg.has_value() && seg->defined()) {
    TORCH_CHECK(seg->dim() == 1 && seg->size(0) == n, "idxs must be a 1d tensor with one entry per box");
    seg_c = seg->to(at::kLong).contiguous();
    seg_ptr = seg_c.const_data_ptr<int64_t>();
  }
  const size_t ws_bytes = tvmi_nms_workspace_bytes(n);
  at::Tensor workspace = at::empty({(int64_t)ws_bytes}, dets.options().dtype(at::kByte));
  at::Tensor keep = at::empty({n}, dets.options().dtype(at::kLong));
  at::Tensor num = at::empty({1}, dets.options().dtype(at::kLong));
  check_status(tvmi_nms(boxes.const_data_ptr(), order.const_data_ptr<int64_t>(), seg_ptr, n, iou_threshold,
                        dtype_of(boxes, "nms"), workspace.mutable_data_ptr(), ws_bytes,
                        keep.mutable_data_ptr<int64_t>(), num.mutable_data_ptr<int64_t>(),
                        current_stream(dets)),
               "nms");
  const int64_t num_keep = num.item<int64_t>();  // the one host sync (data-dependent size)
  return keep.narrow(0, 0, num_keep);
}

at::Tensor nms(const at::Tensor& dets, const at::Tensor& scores, double iou_threshold) {
  return nms_segmented(dets, scores, c10::nullopt, iou_threshold);
}

// ---- roi_align: cuda/roi_align_kernel.cu:334-466
at::Tensor roi_align_forward(const at::Tensor& input, const at::Tensor& rois, double spatial_scale,
                             int64_t pooled_height, int64_t pooled_width, int64_t sampling_ratio,
                             bool aligned) {
  TORCH_CHECK(input.is_cuda(), "input must be a CUDA tensor");
  TORCH_CHECK(rois.is_cuda(), "rois must be a CUDA tensor");
  TORCH_CHECK(rois.dim() == 2 && rois.size(1) == 5, "rois must have shape as Tensor[K, 5]");
  TORCH_CHECK(input.dim() == 4, "input must be a 4d tensor [N, C, H, W]");
  TORCH_CHECK(input.device() == rois.device(), "roi_align_forward_kernel: input and rois must be on the same GPU");
  TORCH_CHECK(input.scalar_type() == rois.scalar_type(),
              "roi_align_forward_kernel: Expected tensor for argument #1 'input' to have the same type as "
              "tensor for argument #2 'rois'; but type ", input.scalar_type(), " does not equal ",
              rois.scalar_type());
  c10::DeviceGuard guard(input.device());
  const int64_t K = rois.size(0), C = input.size(1), H = input.size(2), W = input.size(3);
  at::Tensor output = at::empty({K, C, pooled_height, pooled_width}, input.options());
  if (output.numel() == 0) return output;
  at::Tensor input_ = input.contiguous(), rois_ = rois.contiguous();
  check_status(tvmi_roi_align_forward(input_.const_data_ptr(), rois_.const_data_ptr(), output.mutable_data_ptr(),
                                      dtype_of(input, "roi_align"), input.size(0), C, H, W, K, pooled_height,
                                      pooled_width, spatial_scale, sampling_ratio, aligned ? 1 : 0,
                                      current_stream(input)),
               "roi_align");
  return output;
}

at::Tensor roi_align_backward(const at::Tensor& grad, const at::Tensor& rois, double spatial_scale,
                              int64_t pooled_height, int64_t pooled_width, int64_t batch_size,
                              int64_t channels, int64_t height, int64_t width, int64_t sampling_ratio,
                              bool aligned) {
  TORCH_CHECK(grad.is_cuda(), "grad must be a CUDA tensor");
  TORCH_CHECK(rois.is_cuda(), "rois must be a CUDA tensor");
  TORCH_CHECK(grad.device() == rois.device(), "roi_align_backward_kernel: grad and rois must be on the same GPU");
  TORCH_CHECK(grad.scalar_type() == rois.scalar_type(),
              "roi_align_backward_kernel: grad and rois must have the same type");
  c10::DeviceGuard guard(grad.device());
  at::Tensor grad_input = at::zeros({batch_size, channels, height, width}, grad.options());
  if (grad.numel() == 0) return grad_input;
  at::globalContext().alertNotDeterministic("roi_align_backward_kernel");
  at::Tensor rois_ = rois.contiguous();
  check_status(tvmi_roi_align_backward(grad.const_data_ptr(), rois_.const_data_ptr(), grad_input.mutable_data_ptr(),
                                       dtype_of(grad, "_roi_align_backward"), batch_size, channels, height,
                                       width, rois.size(0), pooled_height, pooled_width, spatial_scale,
                                       sampling_ratio, aligned ? 1 : 0, grad.stride(0), grad.stride(1),
                                       grad.stride(2), grad.stride(3), current_stream(grad)),
               "_roi_align_backward");
  return grad_input;
}

int64_t cuda_version() { return -1; }  // vision.cpp:21-28 without WITH_CUDA; ROCm never checks it
int64_t tvmi_abi_version() { return tvmi_version(); }

}  // namespace

#ifndef TVMI_NO_SCHEMA_DEFS
TORCH_LIBRARY_FRAGMENT(torchvision, m) {
  m.def("nms(Tensor dets, Tensor scores, float iou_threshold) -> Tensor");
  m.def(
      "roi_align(Tensor input, Tensor rois, float spatial_scale, SymInt pooled_height, SymInt pooled_width, int sampling_ratio, bool aligned) -> Tensor");
  m.def(
      "_roi_align_backward(Tensor grad, Tensor rois, float spatial_scale, SymInt pooled_height, SymInt pooled_width, SymInt batch_size, SymInt channels, SymInt height, SymInt width, int sampling_ratio, bool aligned) -> Tensor");
  m.def(
      "roi_pool(Tensor input, Tensor rois, float spatial_scale, SymInt pooled_height, SymInt pooled_width) -> (Tensor, Tensor)");
  m.def(
      "_roi_pool_backward(Tensor grad, Tensor rois, Tensor argmax, float spatial_scale, SymInt pooled_height, SymInt pooled_width, SymInt batch_size, SymInt channels, SymInt height, SymInt width) -> Tensor");
  m.def(
      "ps_roi_align(Tensor input, Tensor rois, float spatial_scale, SymInt pooled_height, SymInt pooled_width, int sampling_ratio) -> (Tensor, Tensor)");
  m.def(
      "_ps_roi_align_backward(Tensor grad, Tensor rois, Tensor channel_mapping, float spatial_scale, SymInt pooled_height, SymInt pooled_width, int sampling_ratio, SymInt batch_size, SymInt channels, SymInt height, SymInt width) -> Tensor");
  m.def(
      "ps_roi_pool(Tensor input, Tensor rois, float spatial_scale, SymInt pooled_height, SymInt pooled_width) -> (Tensor, Tensor)");
  m.def(
      "_ps_roi_pool_backward(Tensor grad, Tensor rois, Tensor channel_mapping, float spatial_scale, SymInt pooled_height, SymInt pooled_width, SymInt batch_size, SymInt channels, SymInt height, SymInt width) -> Tensor");
  m.def(
      "deform_conv2d(Tensor input, Tensor weight, Tensor offset, Tensor mask, Tensor bias, SymInt stride_h, SymInt stride_w, SymInt pad_h, SymInt pad_w, SymInt dilation_h, SymInt dilation_w, SymInt groups, SymInt offset_groups, bool use_mask) -> Tensor");
  m.def(
      "_deform_conv2d_backward(Tensor grad, Tensor input, Tensor weight, Tensor offset, Tensor mask, Tensor bias, SymInt stride_h, SymInt stride_w, SymInt pad_h, SymInt pad_w, SymInt dilation_h, SymInt dilation_w, SymInt groups, SymInt offset_groups, bool use_mask) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("box_iou_rotated(Tensor boxes1, Tensor boxes2) -> Tensor");
  m.def("qnms(Tensor dets, Tensor scores, float iou_threshold) -> Tensor");
  m.def(
      "qroi_align(Tensor input, Tensor rois, float input_scale, int input_zero_point, float rois_scale, int rois_zero_point, float spatial_scale, SymInt pooled_height, SymInt pooled_width, int sampling_ratio, bool aligned) -> Tensor");
  m.def("_cuda_version", &cuda_version);
}
#endif

// Extra entry points that have no reference schema (native fused forms of python-level
// loops in the reference); they live in their own namespace.
TORCH_LIBRARY(tvmi, m) {
  m.def("abi_version", &tvmi_abi_version);
  // batched NMS without the per-class python loop of torchvision/ops/boxes.py:113-126
  m.def("nms_segmented(Tensor dets, Tensor scores, Tensor? idxs, float iou_threshold) -> Tensor");
}

TORCH_LIBRARY_IMPL(torchvision, CUDA, m) {
  m.impl("nms", &nms);
  m.impl("roi_align", &roi_align_forward);
  m.impl("_roi_align_backward", &roi_align_backward);
}

TORCH_LIBRARY_IMPL(tvmi, CUDA, m) { m.impl("nms_segmented", &nms_segmented); }

}  // namespace tvmi_shim
