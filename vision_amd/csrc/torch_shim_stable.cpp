// torch_shim_stable.cpp — the STABLE-ABI half of the dispatcher glue (`_C_stable` of torchvision/extension.py:30).
//
// The reference keeps the CUDA kernels of `torchvision::nms` and `torchvision::box_iou_rotated` on PyTorch's stable ABI
// (cuda/nms_kernel.cu:262-264, cuda/box_iou_rotated_kernel.cu:192-194: STABLE_TORCH_LIBRARY_IMPL + TORCH_BOX, tensors as
// torch::stable::Tensor, helpers through torch_call_dispatcher) and everything else on the classic one.  Same split
// here: this translation unit includes only torch/csrc/stable + torch/headeronly + the C shim (no ATen, no c10 classes,
// no libstdc++ types across the boundary), registers those two kernels on the CUDA key and calls the C ABI of
// include/tvmi.h; tvmi_torch.so (torch_shim.cpp) defines the schemas and holds the rest.  A binary built from this
// file keeps loading across PyTorch releases that keep the stable ABI, which is the point of the reference's move.
#include <torch/csrc/inductor/aoti_torch/c/shim.h>
#include <torch/csrc/stable/accelerator.h>
#include <torch/csrc/stable/library.h>
#include <torch/csrc/stable/ops.h>
#include <torch/csrc/stable/tensor.h>
#include <torch/headeronly/core/ScalarType.h>
#include <torch/headeronly/util/Exception.h>

#include <array>
#include <optional>
#include <tuple>
#include <vector>

#include "tvmi.h"

namespace {

using torch::headeronly::ScalarType;
using torch::stable::Tensor;

void* current_stream(const Tensor& t) {
  void* stream = nullptr;
  TORCH_ERROR_CODE_CHECK(aoti_torch_get_current_cuda_stream(t.get_device_index(), &stream));
  return stream;
}

tvmi_dtype dtype_of(const Tensor& t, const char* op) {
  switch (t.scalar_type()) {
    case ScalarType::Float:
      return TVMI_F32;
    case ScalarType::Double:
      return TVMI_F64;
    case ScalarType::Half:
      return TVMI_F16;
    case ScalarType::BFloat16:
      return TVMI_BF16;
    default:
      STD_TORCH_CHECK(false, op, ": unsupported dtype");
  }
  return TVMI_F32;
}

void check_status(int st, const char* op) { STD_TORCH_CHECK(st == 0, op, " failed: ", tvmi_last_error()); }

Tensor empty_like_device(const Tensor& like, std::initializer_list<int64_t> sizes, ScalarType dtype) {
  const std::vector<int64_t> s(sizes);
  return torch::stable::new_empty(like, torch::headeronly::IntHeaderOnlyArrayRef(s.data(), s.size()), dtype);
}

// aten::sort.stable through the dispatcher (the stable ABI has no sort wrapper of its own): indices of the stable
// descending order
Tensor stable_descending_order(const Tensor& scores) {
  std::array<StableIValue, 4> stack{torch::stable::detail::from(scores), torch::stable::detail::from(std::optional<bool>(true)),
                                    torch::stable::detail::from(int64_t(0)), torch::stable::detail::from(true)};
  TORCH_ERROR_CODE_CHECK(torch_call_dispatcher("aten::sort", "stable", stack.data(), TORCH_ABI_VERSION));
  Tensor values = torch::stable::detail::to<Tensor>(stack[0]);
  (void)values;
  return torch::stable::detail::to<Tensor>(stack[1]);
}

// ---- nms: cuda/nms_kernel.cu:166-258 ------------------------------------------------------------------------
Tensor nms_kernel(const Tensor& dets, const Tensor& scores, double iou_threshold) {
  STD_TORCH_CHECK(dets.is_cuda(), "dets must be a CUDA tensor");
  STD_TORCH_CHECK(scores.is_cuda(), "scores must be a CUDA tensor");
  STD_TORCH_CHECK(dets.dim() == 2, "boxes should be a 2d tensor, got ", dets.dim(), "D");
  STD_TORCH_CHECK(dets.size(1) == 4, "boxes should have 4 elements in dimension 1, got ", dets.size(1));
  STD_TORCH_CHECK(scores.dim() == 1, "scores should be a 1d tensor, got ", scores.dim(), "D");
  STD_TORCH_CHECK(dets.size(0) == scores.size(0), "boxes and scores should have same number of elements in ",
                  "dimension 0, got ", dets.size(0), " and ", scores.size(0));
  const torch::stable::accelerator::DeviceGuard guard(dets.get_device_index());
  const int64_t n = dets.size(0);
  if (dets.numel() == 0) return empty_like_device(dets, {0}, ScalarType::Long);

  // Half / BFloat16 boxes are evaluated in fp32, as cuda/nms_kernel.cu:32-53 does for Half
  Tensor boxes = dets;
  if (dets.scalar_type() == ScalarType::Half || dets.scalar_type() == ScalarType::BFloat16)
    boxes = torch::stable::to(dets, ScalarType::Float);
  STD_TORCH_CHECK(boxes.scalar_type() == ScalarType::Float || boxes.scalar_type() == ScalarType::Double,
                  "nms: boxes must be a floating point tensor");
  boxes = torch::stable::contiguous(boxes);
  Tensor order;
  if (scores.scalar_type() == ScalarType::Float && n <= 4096) {
    // detector-step sizes: the stable descending order in one launch (tvmi_sort_scores_desc)
    const Tensor sc = torch::stable::contiguous(scores);
    order = empty_like_device(dets, {n}, ScalarType::Long);
    check_status(tvmi_sort_scores_desc(static_cast<const float*>(sc.const_data_ptr()), n,
                                       static_cast<int64_t*>(order.mutable_data_ptr()), current_stream(dets)),
                 "sort_scores_desc");
  } else if (scores.scalar_type() == ScalarType::Float && n < (1ll << 31)) {
    // larger lists: key pass + rocPRIM radix sort of (key, 32-bit index) pairs (score_sort.hip)
    const Tensor sc = torch::stable::contiguous(scores);
    order = empty_like_device(dets, {n}, ScalarType::Long);
    const size_t sb = tvmi_sort_scores_desc_workspace_bytes(n);
    Tensor sws = empty_like_device(dets, {(int64_t)sb}, ScalarType::Byte);
    check_status(tvmi_sort_scores_desc_large(static_cast<const float*>(sc.const_data_ptr()), n,
                                             static_cast<int64_t*>(order.mutable_data_ptr()), sws.mutable_data_ptr(), sb,
                                             current_stream(dets)),
                 "sort_scores_desc_large");
  } else {
    order = torch::stable::contiguous(stable_descending_order(scores));
  }
  Tensor keep = empty_like_device(dets, {n}, ScalarType::Long);
  Tensor num = empty_like_device(dets, {1}, ScalarType::Long);
  const size_t ws_bytes = tvmi_nms_workspace_bytes(n);
  Tensor workspace = empty_like_device(dets, {(int64_t)ws_bytes}, ScalarType::Byte);
  check_status(tvmi_nms_blocking(boxes.const_data_ptr(), static_cast<const int64_t*>(order.const_data_ptr()), nullptr, n, iou_threshold,
                        dtype_of(boxes, "nms"), workspace.mutable_data_ptr(), ws_bytes,
                        static_cast<int64_t*>(keep.mutable_data_ptr()), static_cast<int64_t*>(num.mutable_data_ptr()),
                        current_stream(dets)),
               "nms");
  // the one host synchronisation of the op: the data-dependent output size (the reference takes it in masked_select)
  int64_t count = 0;
  TORCH_ERROR_CODE_CHECK(aoti_torch_item_int64(num.get(), &count));
  return torch::stable::narrow(keep, 0, 0, count);
}

// ---- box_iou_rotated: cuda/box_iou_rotated_kernel.cu:92-188 ----------------------------------------------------
Tensor box_iou_rotated_kernel(const Tensor& boxes1, const Tensor& boxes2) {
  STD_TORCH_CHECK(boxes1.is_cuda(), "boxes1 must be a CUDA tensor");
  STD_TORCH_CHECK(boxes2.is_cuda(), "boxes2 must be a CUDA tensor");
  STD_TORCH_CHECK(boxes1.dim() == 2 && boxes1.size(1) == 5, "boxes1 must have shape (N, 5)");
  STD_TORCH_CHECK(boxes2.dim() == 2 && boxes2.size(1) == 5, "boxes2 must have shape (M, 5)");
  STD_TORCH_CHECK(boxes1.scalar_type() == boxes2.scalar_type(), "boxes1 and boxes2 must have the same dtype");
  const torch::stable::accelerator::DeviceGuard guard(boxes1.get_device_index());
  Tensor b1 = torch::stable::contiguous(boxes1), b2 = torch::stable::contiguous(boxes2);
  if (b1.scalar_type() == ScalarType::Half || b1.scalar_type() == ScalarType::BFloat16) {
    b1 = torch::stable::to(b1, ScalarType::Float);
    b2 = torch::stable::to(b2, ScalarType::Float);
  }
  const int64_t N = b1.size(0), M = b2.size(0);
  Tensor ious = empty_like_device(b1, {N, M}, ScalarType::Float);
  if (N > 0 && M > 0)
    check_status(tvmi_box_iou_rotated(b1.const_data_ptr(), b2.const_data_ptr(), static_cast<float*>(ious.mutable_data_ptr()),
                                      dtype_of(b1, "box_iou_rotated"), N, M, current_stream(boxes1)),
                 "box_iou_rotated");
  return ious;
}

}  // namespace

STABLE_TORCH_LIBRARY_IMPL(torchvision, CUDA, m) {
  m.impl("nms", TORCH_BOX(&nms_kernel));
  m.impl("box_iou_rotated", TORCH_BOX(&box_iou_rotated_kernel));
}
