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
#include <ATen/ops/_upsample_bicubic2d_aa_cuda_dispatch.h>
#include <ATen/ops/_upsample_bilinear2d_aa_cuda_dispatch.h>
#include <ATen/ops/_upsample_nearest_exact2d_cuda_dispatch.h>
#include <ATen/ops/upsample_bicubic2d_cuda_dispatch.h>
#include <ATen/ops/upsample_bilinear2d_cuda_dispatch.h>
#include <ATen/ops/upsample_nearest2d_cuda_dispatch.h>
#include <ATen/ops/_upsample_bicubic2d_aa_backward_cuda_dispatch.h>
#include <ATen/ops/_upsample_bilinear2d_aa_backward_cuda_dispatch.h>
#include <ATen/ops/_upsample_nearest_exact2d_backward_cuda_dispatch.h>
#include <ATen/ops/upsample_bicubic2d_backward_cuda_dispatch.h>
#include <ATen/ops/upsample_bilinear2d_backward_cuda_dispatch.h>
#include <ATen/ops/upsample_nearest2d_backward_cuda_dispatch.h>
#include <ATen/native/Resize.h>
#include <atomic>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <string>
#include <memory>
#include <mutex>
#include <c10/core/DeviceGuard.h>
#include <torch/csrc/inductor/aoti_torch/c/shim.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

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

// partition of the score order by segment id + the segment-major NMS on it (n_live: device count of the masked form, or null)
int segment_major_nms(const at::Tensor& boxes, const at::Tensor& order, const at::Tensor& seg, const int64_t* n_live,
                      int64_t num_segments, double iou_threshold, at::Tensor& nms_ws, at::Tensor& keep, at::Tensor& num) {
  const int64_t n = boxes.size(0);
  const auto lopt = boxes.options().dtype(at::kLong);
  at::Tensor keys = at::empty({n}, lopt), perm = at::empty({n}, lopt), flag = at::empty({1}, boxes.options().dtype(at::kInt));
  const size_t pb = tvmi_partition_by_segment_workspace_bytes(n);
  at::Tensor pws = at::empty({(int64_t)pb}, boxes.options().dtype(at::kByte));
  int st = tvmi_partition_by_segment(order.const_data_ptr<int64_t>(), seg.const_data_ptr<int64_t>(), n, n_live, num_segments,
                                     keys.mutable_data_ptr<int64_t>(), perm.mutable_data_ptr<int64_t>(), flag.mutable_data_ptr<int>(),
                                     pws.mutable_data_ptr(), pb, current_stream(boxes));
  if (st != 0) return st;
  return tvmi_nms_segmented_devcount(boxes.const_data_ptr(), order.const_data_ptr<int64_t>(), keys.const_data_ptr<int64_t>(),
                                     perm.const_data_ptr<int64_t>(), n, n_live, flag.const_data_ptr<int>(), iou_threshold,
                                     dtype_of(boxes, "nms"), nms_ws.mutable_data_ptr(), (size_t)nms_ws.numel(),
                                     keep.mutable_data_ptr<int64_t>(), num.mutable_data_ptr<int64_t>(), current_stream(boxes));
}

// ---- nms: cuda/nms_kernel.cu:166-258 (checks and messages), cpu/nms_kernel.cpp (semantics)
// Returns (keep [n] with a valid prefix, num [1] int64 on the device).  `allow_sync`: the segment-major path needs
// one look at `num` to detect an over-long segment; without it (graph-capturable form) n > 4096 still takes that path
// and reports -1 in `num` for such inputs.
std::tuple<at::Tensor, at::Tensor> nms_impl(const at::Tensor& dets, const at::Tensor& scores,
                                            const c10::optional<at::Tensor>& seg, double iou_threshold, int64_t num_segments,
                                            bool allow_sync) {
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
  if (dets.numel() == 0)
    return std::make_tuple(at::empty({0}, dets.options().dtype(at::kLong)), at::zeros({1}, dets.options().dtype(at::kLong)));

  // Half/BFloat16 boxes are evaluated in fp32, as cuda/nms_kernel.cu:32-53 does for Half.
  at::Tensor boxes = dets;
  if (dets.scalar_type() == at::kHalf || dets.scalar_type() == at::kBFloat16) boxes = dets.to(at::kFloat);
  TORCH_CHECK(boxes.scalar_type() == at::kFloat || boxes.scalar_type() == at::kDouble,
              "nms: boxes must be a floating point tensor");
  boxes = boxes.contiguous();
  at::Tensor seg_c;
  const int64_t* seg_ptr = nullptr;
  if (seg.has_value() && seg->defined()) {
    TORCH_CHECK(seg->dim() == 1 && seg->size(0) == n, "idxs must be a 1d tensor with one entry per box");
    seg_c = seg->to(at::kLong).contiguous();
    seg_ptr = seg_c.const_data_ptr<int64_t>();
  }
  at::Tensor keep = at::empty({n}, dets.options().dtype(at::kLong));
  at::Tensor num = at::empty({1}, dets.options().dtype(at::kLong));
  // detector-step sizes: score order, per-segment tiles, sweeps and the global-order compaction as ONE launch (round 6)
  int64_t step_on = 1;
  tvmi_get_option("nms.step_fused", &step_on);
  if (step_on && boxes.scalar_type() == at::kFloat && scores.scalar_type() == at::kFloat && n <= 4096 &&
      (seg_ptr ? (num_segments >= 1 && num_segments <= 64) : n <= 1024)) {
    at::Tensor sc = scores.contiguous();
    const int64_t S = seg_ptr ? num_segments : 1;
    const size_t sb = tvmi_nms_step_workspace_bytes(n, S);
    at::Tensor sws = at::empty({(int64_t)sb}, dets.options().dtype(at::kByte));
    check_status(tvmi_nms_step(boxes.const_data_ptr<float>(), sc.const_data_ptr<float>(), seg_ptr, n, S, iou_threshold,
                               sws.mutable_data_ptr(), sb, keep.mutable_data_ptr<int64_t>(), num.mutable_data_ptr<int64_t>(), nullptr,
                               nullptr, 0, 0, nullptr, 0, nullptr, 0, current_stream(dets)),
                 "nms_step");
    if (!allow_sync || num.item<int64_t>() >= 0) return std::make_tuple(keep, num);
    // a segment above 1024 boxes or an id outside [0, num_segments): the paths below
  }
  at::Tensor order;
  if (scores.scalar_type() == at::kFloat && n <= 4096) {
    // detector-step sizes: the stable descending order in one launch instead of torch's sort + arange + copies
    at::Tensor sc = scores.contiguous();
    order = at::empty({n}, dets.options().dtype(at::kLong));
    check_status(tvmi_sort_scores_desc(sc.const_data_ptr<float>(), n, order.mutable_data_ptr<int64_t>(), current_stream(dets)),
                 "sort_scores_desc");
  } else if (scores.scalar_type() == at::kFloat && n < (1ll << 31)) {
    // larger lists: key pass + rocPRIM radix sort of (key, 32-bit index) pairs (score_sort.hip), a third of the time of
    // at::sort's (float, int64) merge sort at 100k scores
    at::Tensor sc = scores.contiguous();
    order = at::empty({n}, dets.options().dtype(at::kLong));
    const size_t sb = tvmi_sort_scores_desc_workspace_bytes(n);
    at::Tensor sws = at::empty({(int64_t)sb}, dets.options().dtype(at::kByte));
    check_status(tvmi_sort_scores_desc_large(sc.const_data_ptr<float>(), n, order.mutable_data_ptr<int64_t>(),
                                             sws.mutable_data_ptr(), sb, current_stream(dets)),
                 "sort_scores_desc_large");
  } else {
    order = std::get<1>(at::sort(scores, /*stable=*/true, /*dim=*/0, /*descending=*/true));
  }
  if (seg_ptr && n <= 4096 && num_segments >= 1 && num_segments <= 1024) {
    // detector-step sizes with a known id range: per-segment tiles + concurrent per-segment sweeps, no second sort
    const size_t sb = tvmi_nms_small_segments_workspace_bytes(n, num_segments);
    at::Tensor sws = at::empty({(int64_t)sb}, dets.options().dtype(at::kByte));
    check_status(tvmi_nms_small_segments(boxes.const_data_ptr(), order.const_data_ptr<int64_t>(), seg_ptr, n, num_segments,
                                         iou_threshold, dtype_of(boxes, "nms"), sws.mutable_data_ptr(), sb,
                                         keep.mutable_data_ptr<int64_t>(), num.mutable_data_ptr<int64_t>(),
                                         current_stream(dets)),
                 "nms_small_segments");
    if (!allow_sync || num.item<int64_t>() >= 0) return std::make_tuple(keep, num);
    // a segment above 1024 boxes or an id outside [0, num_segments): general path below
  }
  if (seg_ptr && n > 4096) {  // up to 4096 boxes the single-launch global sweep is as fast
    // segment-major path: stable partition of the score order by segment (a radix sort over the id bits of the sequence that
    // is already in score order), block-diagonal masks, one sweep workgroup per segment
    const size_t sb = tvmi_nms_segmented_workspace_bytes(n);
    at::Tensor sws = at::empty({(int64_t)sb}, dets.options().dtype(at::kByte));
    check_status(segment_major_nms(boxes, order, seg_c, nullptr, num_segments, iou_threshold, sws, keep, num), "nms_segmented");
    if (!allow_sync || num.item<int64_t>() >= 0) return std::make_tuple(keep, num);
    // a segment above 8,192 boxes: fall through to the global-order pipeline
  }
  const size_t ws_bytes = tvmi_nms_workspace_bytes(n);
  at::Tensor workspace = at::empty({(int64_t)ws_bytes}, dets.options().dtype(at::kByte));
  // callers that read the size on the host anyway take the re-planning form (one stream sync per re-plan above
  // "nms.replan_min_boxes" boxes); the padded / capturable form never synchronises
  check_status((allow_sync ? tvmi_nms_blocking : tvmi_nms)(boxes.const_data_ptr(), order.const_data_ptr<int64_t>(), seg_ptr, n,
                                                            iou_threshold, dtype_of(boxes, "nms"), workspace.mutable_data_ptr(),
                                                            ws_bytes, keep.mutable_data_ptr<int64_t>(),
                                                            num.mutable_data_ptr<int64_t>(), current_stream(dets)),
               "nms");
  return std::make_tuple(keep, num);
}

at::Tensor nms_segmented(const at::Tensor& dets, const at::Tensor& scores, const c10::optional<at::Tensor>& seg,
                         double iou_threshold, int64_t num_segments) {
  auto r = nms_impl(dets, scores, seg, iou_threshold, num_segments, /*allow_sync=*/true);
  if (std::get<0>(r).numel() == 0) return std::get<0>(r);
  const int64_t num_keep = std::get<1>(r).item<int64_t>();  // the one host sync (data-dependent size)
  return std::get<0>(r).narrow(0, 0, num_keep);
}

// ---- qnms (quantized/cpu/qnms_kernel.cpp:22-146; CPU-only in the reference): integer boxes / scores.  The reference evaluates
// the boxes as float32 with the arithmetic of cpu/nms_kernel.cpp (the quantisation scale cancels in the IoU) in the stable
// descending order of the INTEGER scores — i.e. our nms on the widened boxes with aten's stable sort of the integer scores.
at::Tensor qnms_forward(const at::Tensor& dets, const at::Tensor& scores, double iou_threshold) {
  TORCH_CHECK(dets.is_cuda(), "dets must be a CUDA tensor");
  TORCH_CHECK(scores.is_cuda(), "scores must be a CUDA tensor");
  TORCH_CHECK(dets.scalar_type() == scores.scalar_type(), "dets should have the same type as scores");
  TORCH_CHECK(c10::isIntegralType(dets.scalar_type(), /*includeBool=*/false), "qnms: integer tensors expected (the reference dispatches AT_INTEGRAL_TYPES)");
  return nms_segmented(dets.to(at::kFloat), scores, c10::nullopt, iou_threshold, -1);
}

// no host synchronisation: (keep [n] whose first num[0] entries are valid, num [1] int64 on the device)
std::tuple<at::Tensor, at::Tensor> nms_segmented_padded(const at::Tensor& dets, const at::Tensor& scores,
                                                        const c10::optional<at::Tensor>& seg, double iou_threshold,
                                                        int64_t num_segments) {
  return nms_impl(dets, scores, seg, iou_threshold, num_segments, /*allow_sync=*/false);
}



// Masked form (round 5): `valid` [n] uint8 marks the candidates that take part; the others leave the problem ON THE DEVICE
// (score -inf / key INT64_MAX -> behind every live candidate in both sorts; the kernels read the live count from memory), so
// the detector post-processing needs no `nonzero` compaction and no host read between the candidate kernel and the packing
// launch.  keep holds indices into the UNcompacted candidate list, in the reference's order (descending score, stable).
// Limits (checked on the device, reported as num = -1): a segment above 8,192 live boxes on the segment-major path; an id outside
// [0, num_segments) or a segment above 1,024 live boxes on the small path.  `max_segment_size` is the caller's STATIC bound on the
// boxes of one segment (-1 = unknown -> n): the small path is taken only when the bound fits it, so with n <= 8192 or
// max_segment_size <= 8192 (and ids in range) the op cannot come back with -1 (the same data used to fail at n = 4096
// and pass at n = 4097).  `vision_amd.detection_post._masked_nms` routes anything beyond these limits to the compacting form.
std::tuple<at::Tensor, at::Tensor> nms_segmented_masked(const at::Tensor& dets, const at::Tensor& scores, const at::Tensor& seg,
                                                        const at::Tensor& valid, double iou_threshold, int64_t num_segments,
                                                        int64_t max_segment_size) {
  TORCH_CHECK(dets.is_cuda() && scores.is_cuda() && seg.is_cuda() && valid.is_cuda(), "nms_segmented_masked: CUDA tensors expected");
  TORCH_CHECK(dets.dim() == 2 && dets.size(1) == 4, "boxes should be a 2d tensor [N, 4]");
  const int64_t n = dets.size(0);
  TORCH_CHECK(scores.dim() == 1 && seg.dim() == 1 && valid.dim() == 1 && scores.size(0) == n && seg.size(0) == n && valid.size(0) == n,
              "nms_segmented_masked: scores, idxs and valid must have one entry per box");
  TORCH_CHECK(dets.scalar_type() == at::kFloat && scores.scalar_type() == at::kFloat, "nms_segmented_masked: float32 boxes and scores");
  TORCH_CHECK(valid.scalar_type() == at::kByte || valid.scalar_type() == at::kBool, "nms_segmented_masked: valid must be uint8 or bool");
  TORCH_CHECK(n < (1ll << 31), "nms_segmented_masked: too many boxes");
  c10::DeviceGuard guard(dets.device());
  at::Tensor keep = at::zeros({n}, dets.options().dtype(at::kLong));   // the tail behind num stays 0 (a deterministic output)
  at::Tensor num = at::zeros({1}, dets.options().dtype(at::kLong));
  if (n == 0) return std::make_tuple(keep, num);
  at::Tensor boxes = dets.contiguous(), sc = scores.contiguous(), sg = seg.to(at::kLong).contiguous(), vl = valid.contiguous();
  at::Tensor sc_m = at::empty_like(sc), sg_m = at::empty_like(sg), n_live = at::empty({1}, dets.options().dtype(at::kLong));
  check_status(tvmi_nms_mask_inputs(sc.const_data_ptr<float>(), sg.const_data_ptr<int64_t>(),
                                    static_cast<const uint8_t*>(vl.const_data_ptr()), n, sc_m.mutable_data_ptr<float>(),
                                    sg_m.mutable_data_ptr<int64_t>(), n_live.mutable_data_ptr<int64_t>(), current_stream(dets)),
               "nms_mask_inputs");
  at::Tensor order = at::empty({n}, dets.options().dtype(at::kLong));
  if (n <= 4096) {
    check_status(tvmi_sort_scores_desc(sc_m.const_data_ptr<float>(), n, order.mutable_data_ptr<int64_t>(), current_stream(dets)),
                 "sort_scores_desc");
  } else {
    const size_t sb = tvmi_sort_scores_desc_workspace_bytes(n);
    at::Tensor sws = at::empty({(int64_t)sb}, dets.options().dtype(at::kByte));
    check_status(tvmi_sort_scores_desc_large(sc_m.const_data_ptr<float>(), n, order.mutable_data_ptr<int64_t>(),
                                             sws.mutable_data_ptr(), sb, current_stream(dets)),
                 "sort_scores_desc_large");
  }
  const int64_t seg_bound = (max_segment_size < 0 || max_segment_size > n) ? n : max_segment_size;
  if (n <= 4096 && num_segments >= 1 && num_segments <= 1024 && seg_bound <= 1024) {
    const size_t sb = tvmi_nms_small_segments_workspace_bytes(n, num_segments);
    at::Tensor sws = at::empty({(int64_t)sb}, dets.options().dtype(at::kByte));
    check_status(tvmi_nms_small_segments_devcount(boxes.const_data_ptr(), order.const_data_ptr<int64_t>(), sg_m.const_data_ptr<int64_t>(),
                                                  n, n_live.const_data_ptr<int64_t>(), num_segments, iou_threshold, TVMI_F32,
                                                  sws.mutable_data_ptr(), sb, keep.mutable_data_ptr<int64_t>(),
                                                  num.mutable_data_ptr<int64_t>(), current_stream(dets)),
                 "nms_small_segments");
    return std::make_tuple(keep, num);
  }
  const size_t sb = tvmi_nms_segmented_workspace_bytes(n);
  at::Tensor sws = at::empty({(int64_t)sb}, dets.options().dtype(at::kByte));
  // (the unmasked ids go in: dead ranks are recognised by their position behind *n_live, not by their key)
  check_status(segment_major_nms(boxes, order, sg, n_live.const_data_ptr<int64_t>(), num_segments, iou_threshold, sws, keep, num),
               "nms_segmented");
  return std::make_tuple(keep, num);
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
  at::Tensor output = at::empty({K, C, pooled_height, pooled_width}, input.options().memory_format(at::MemoryFormat::Contiguous));
  if (output.numel() == 0) return output;
  if ((input.scalar_type() == at::kFloat ||
       ((input.scalar_type() == at::kHalf || input.scalar_type() == at::kBFloat16) && C % 2 == 0)) &&
      pooled_height == 7 && pooled_width == 7 && sampling_ratio == 2 && C > 1 &&
      H >= 2 && W >= 2 && !input.is_contiguous() && input.is_contiguous(at::MemoryFormat::ChannelsLast) &&
      H * W * C < (1ll << 31)) {
    // channels_last input: native NHWC kernel instead of the reference's input.contiguous() copy
    const void* ptr = input.const_data_ptr();
    const double scale = spatial_scale;
    at::Tensor r = rois.to(at::kFloat).contiguous();  // the multi-scale entries take float32 RoIs (exact upcast)
    at::Tensor order_ws = at::empty({K}, input.options().dtype(at::kInt));   // launch order of the RoIs
    check_status(tvmi_multiscale_roi_align_forward_nhwc(&ptr, &H, &W, &scale, 1, r.const_data_ptr(), output.mutable_data_ptr(),
                                                        dtype_of(input, "roi_align"), input.size(0), C, K, 7, 7, 2, aligned ? 1 : 0, 0, 0, 224.0,
                                                        4.0, 1e-6, order_ws.mutable_data_ptr(), (size_t)K * sizeof(int), current_stream(input)),
                 "roi_align");
    return output;
  }
  at::Tensor input_ = input.contiguous(), rois_ = rois.contiguous();
  // "declined by the LDS-DMA kernel" flags + the tables of the shared-staging kernel
  const size_t ws_bytes = tvmi_roi_align_forward_workspace_bytes(K, pooled_height, pooled_width, sampling_ratio);
  at::Tensor ws = at::empty({(int64_t)ws_bytes}, input.options().dtype(at::kByte));
  check_status(tvmi_roi_align_forward(input_.const_data_ptr(), rois_.const_data_ptr(), output.mutable_data_ptr(),
                                      dtype_of(input, "roi_align"), input.size(0), C, H, W, K, pooled_height,
                                      pooled_width, spatial_scale, sampling_ratio, aligned ? 1 : 0,
                                      ws.mutable_data_ptr(), ws_bytes, current_stream(input)),
               "roi_align");
  return output;
}

// The tile-owner backward keeps ~9 KB (7x7) / ~18 KB (14x14) of coefficient tables per RoI.  That scratch is absent in the
// reference: it is only taken while it stays within 2x the tensors the call moves anyway (or 256 MiB) — a call with very many
// RoIs on small maps takes the atomic regime instead of gigabytes of tables.
bool owner_workspace_is_reasonable(size_t ws_bytes, const at::Tensor& grad, int64_t grad_input_numel) {
  const size_t moved = (size_t)(grad.numel() + grad_input_numel) * (size_t)grad.element_size();
  return ws_bytes <= std::max<size_t>(size_t(256) << 20, 2 * moved);
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
  const bool low = grad.scalar_type() == at::kHalf || grad.scalar_type() == at::kBFloat16;
  if (batch_size * channels * height * width == 0 || rois.size(0) == 0)
    return at::zeros({batch_size, channels, height, width}, grad.options());
  at::Tensor rois_ = rois.contiguous();
  // The tile-owner backward (fp32 / fp16 / bf16 grads, 7x7 / 14x14 bins) wants the bins of a channel contiguous; it reads
  // the grads in their own type, accumulates in fp32 registers and writes every pixel of grad_input exactly once, rounded
  // once (no zero-fill, no atomics, deterministic).  Everything else accumulates atomically into a zero-filled tensor like
  // the reference (cuda/roi_align_kernel.cu:440); 16-bit grads then go through fp32 (native float atomics instead of
  // 16-bit CAS loops, one rounding at the end).
  at::Tensor g = grad;
  size_t ws_bytes = tvmi_roi_align_backward_workspace_bytes(batch_size, rois.size(0), pooled_height, pooled_width);
  if (!owner_workspace_is_reasonable(ws_bytes, grad, batch_size * channels * height * width)) ws_bytes = 0;
  if (ws_bytes != 0 && (grad.scalar_type() == at::kFloat || low) &&
      !(grad.stride(3) == 1 && grad.stride(2) == pooled_width && grad.stride(1) == pooled_height * pooled_width))
    g = grad.contiguous();
  const tvmi_dtype dt = dtype_of(g, "_roi_align_backward");
  const bool overwrites = ws_bytes != 0 && tvmi_roi_align_backward_overwrites(dt, batch_size, channels, height, width, rois.size(0),
                                                                              pooled_height, pooled_width, g.stride(1), g.stride(2),
                                                                              g.stride(3), ws_bytes) != 0;
  if (low && !overwrites)
    return roi_align_backward(grad.to(at::kFloat), rois.to(at::kFloat), spatial_scale, pooled_height, pooled_width,
                              batch_size, channels, height, width, sampling_ratio, aligned)
        .to(grad.scalar_type());
  at::Tensor grad_input = overwrites ? at::empty({batch_size, channels, height, width}, g.options())
                                     : at::zeros({batch_size, channels, height, width}, g.options());
  if (!overwrites) at::globalContext().alertNotDeterministic("roi_align_backward_kernel");
  at::Tensor ws = at::empty({(int64_t)(overwrites ? ws_bytes : 0)}, g.options().dtype(at::kByte));
  check_status(tvmi_roi_align_backward(g.const_data_ptr(), rois_.const_data_ptr(), grad_input.mutable_data_ptr(), dt, batch_size,
                                       channels, height, width, rois.size(0), pooled_height, pooled_width, spatial_scale,
                                       sampling_ratio, aligned ? 1 : 0, g.stride(0), g.stride(1), g.stride(2), g.stride(3),
                                       overwrites ? ws.mutable_data_ptr() : nullptr, overwrites ? ws_bytes : 0, current_stream(grad)),
               "_roi_align_backward");
  return grad_input;
}

// ---- roi_pool: cuda/roi_pool_kernel.cu:127-260
std::tuple<at::Tensor, at::Tensor> roi_pool_forward(const at::Tensor& input, const at::Tensor& rois,
                                                    double spatial_scale, int64_t pooled_height,
                                                    int64_t pooled_width) {
  TORCH_CHECK(input.is_cuda(), "input must be a CUDA tensor");
  TORCH_CHECK(rois.is_cuda(), "rois must be a CUDA tensor");
  TORCH_CHECK(rois.dim() == 2 && rois.size(1) == 5, "Tensor rois should have shape as Tensor[K, 5]");
  TORCH_CHECK(input.dim() == 4, "input must be a 4d tensor [N, C, H, W]");
  TORCH_CHECK(input.device() == rois.device(), "roi_pool_forward_kernel: input and rois must be on the same GPU");
  TORCH_CHECK(input.scalar_type() == rois.scalar_type(), "roi_pool_forward_kernel: input and rois must have the same type");
  c10::DeviceGuard guard(input.device());
  const int64_t K = rois.size(0), C = input.size(1), H = input.size(2), W = input.size(3);
  at::Tensor output = at::empty({K, C, pooled_height, pooled_width}, input.options());
  at::Tensor argmax = at::empty({K, C, pooled_height, pooled_width}, input.options().dtype(at::kInt));
  if (output.numel() == 0) return std::make_tuple(output, argmax);
  at::Tensor input_ = input.contiguous(), rois_ = rois.contiguous();
  check_status(tvmi_roi_pool_forward(input_.const_data_ptr(), rois_.const_data_ptr(), output.mutable_data_ptr(),
                                     argmax.mutable_data_ptr<int32_t>(), dtype_of(input, "roi_pool"), input.size(0),
                                     C, H, W, K, pooled_height, pooled_width, spatial_scale, current_stream(input)),
               "roi_pool");
  return std::make_tuple(output, argmax);
}

at::Tensor roi_pool_backward(const at::Tensor& grad, const at::Tensor& rois, const at::Tensor& argmax,
                             double spatial_scale, int64_t pooled_height, int64_t pooled_width,
                             int64_t batch_size, int64_t channels, int64_t height, int64_t width) {
  TORCH_CHECK(grad.is_cuda(), "grad must be a CUDA tensor");
  TORCH_CHECK(rois.is_cuda(), "rois must be a CUDA tensor");
  TORCH_CHECK(argmax.is_cuda(), "argmax must be a CUDA tensor");
  TORCH_CHECK(grad.scalar_type() == rois.scalar_type(), "roi_pool_backward_kernel: grad and rois must have the same type");
  c10::DeviceGuard guard(grad.device());
  if (grad.numel() == 0) return at::zeros({batch_size, channels, height, width}, grad.options());
  // plane-owner regime (a gradient plane fits a CU's LDS): every pixel is written, fixed summation order
  const bool overwrites = tvmi_roi_pool_backward_overwrites(dtype_of(grad, "_roi_pool_backward"), batch_size, channels, height, width) != 0;
  at::Tensor grad_input = overwrites ? at::empty({batch_size, channels, height, width}, grad.options())
                                     : at::zeros({batch_size, channels, height, width}, grad.options());
  if (!overwrites) at::globalContext().alertNotDeterministic("roi_pool_backward_kernel");
  at::Tensor argmax_ = argmax.to(at::kInt).contiguous(), rois_ = rois.contiguous();
  check_status(tvmi_roi_pool_backward(grad.const_data_ptr(), rois_.const_data_ptr(), argmax_.const_data_ptr<int32_t>(),
                                      grad_input.mutable_data_ptr(), dtype_of(grad, "_roi_pool_backward"), batch_size,
                                      channels, height, width, rois.size(0), pooled_height, pooled_width,
                                      grad.stride(0), grad.stride(1), grad.stride(2), grad.stride(3),
                                      current_stream(grad)),
               "_roi_pool_backward");
  return grad_input;
}

// ---- ps_roi_align / ps_roi_pool: cuda/ps_roi_align_kernel.cu, cuda/ps_roi_pool_kernel.cu host parts
std::tuple<at::Tensor, at::Tensor> ps_roi_common_forward(const char* name, bool align, const at::Tensor& input,
                                                         const at::Tensor& rois, double spatial_scale,
                                                         int64_t pooled_height, int64_t pooled_width,
                                                         int64_t sampling_ratio) {
  TORCH_CHECK(input.is_cuda(), "input must be a CUDA tensor");
  TORCH_CHECK(rois.is_cuda(), "rois must be a CUDA tensor");
  TORCH_CHECK(rois.dim() == 2 && rois.size(1) == 5, "Tensor rois should have shape as Tensor[K, 5]");
  TORCH_CHECK(input.dim() == 4, "input must be a 4d tensor [N, C, H, W]");
  TORCH_CHECK(input.device() == rois.device(), name, ": input and rois must be on the same GPU");
  TORCH_CHECK(input.scalar_type() == rois.scalar_type(), name, ": input and rois must have the same type");
  c10::DeviceGuard guard(input.device());
  const int64_t K = rois.size(0), C = input.size(1), H = input.size(2), W = input.size(3);
  TORCH_CHECK(C % (pooled_height * pooled_width) == 0,
              "input channels must be a multiple of pooling height * pooling width");
  const int64_t C_out = C / (pooled_height * pooled_width);
  at::Tensor output = at::empty({K, C_out, pooled_height, pooled_width}, input.options());
  at::Tensor mapping = at::empty({K, C_out, pooled_height, pooled_width}, input.options().dtype(at::kInt));
  if (output.numel() == 0) return std::make_tuple(output, mapping);
  at::Tensor input_ = input.contiguous(), rois_ = rois.contiguous();
  int st = align ? tvmi_ps_roi_align_forward(input_.const_data_ptr(), rois_.const_data_ptr(), output.mutable_data_ptr(),
                                             mapping.mutable_data_ptr<int32_t>(), dtype_of(input, name), input.size(0),
                                             C, H, W, K, pooled_height, pooled_width, spatial_scale, sampling_ratio,
                                             current_stream(input))
                 : tvmi_ps_roi_pool_forward(input_.const_data_ptr(), rois_.const_data_ptr(), output.mutable_data_ptr(),
                                            mapping.mutable_data_ptr<int32_t>(), dtype_of(input, name), input.size(0),
                                            C, H, W, K, pooled_height, pooled_width, spatial_scale,
                                            current_stream(input));
  check_status(st, name);
  return std::make_tuple(output, mapping);
}

at::Tensor ps_roi_common_backward(const char* name, bool align, const at::Tensor& grad, const at::Tensor& rois,
                                  const at::Tensor& channel_mapping, double spatial_scale, int64_t pooled_height,
                                  int64_t pooled_width, int64_t sampling_ratio, int64_t batch_size,
                                  int64_t channels, int64_t height, int64_t width) {
  TORCH_CHECK(grad.is_cuda(), "grad must be a CUDA tensor");
  TORCH_CHECK(rois.is_cuda(), "rois must be a CUDA tensor");
  TORCH_CHECK(channel_mapping.is_cuda(), "channel_mapping must be a CUDA tensor");
  TORCH_CHECK(grad.scalar_type() == rois.scalar_type(), name, ": grad and rois must have the same type");
  c10::DeviceGuard guard(grad.device());
  if (grad.numel() == 0) return at::zeros({batch_size, channels, height, width}, grad.options());
  // plane-owner regime (a gradient plane fits a CU's LDS): every pixel is written, fixed summation order — the same
  // predicate as roi_pool's
  const bool overwrites = tvmi_ps_roi_backward_overwrites(dtype_of(grad, name), batch_size, channels, height, width) != 0;
  at::Tensor grad_input = overwrites ? at::empty({batch_size, channels, height, width}, grad.options())
                                     : at::zeros({batch_size, channels, height, width}, grad.options());
  if (!overwrites) at::globalContext().alertNotDeterministic(name);
  at::Tensor grad_ = grad.contiguous(), rois_ = rois.contiguous(), map_ = channel_mapping.to(at::kInt).contiguous();
  int st = align ? tvmi_ps_roi_align_backward(grad_.const_data_ptr(), rois_.const_data_ptr(),
                                              map_.const_data_ptr<int32_t>(), grad_input.mutable_data_ptr(),
                                              dtype_of(grad, name), batch_size, channels, height, width, rois.size(0),
                                              pooled_height, pooled_width, spatial_scale, sampling_ratio,
                                              current_stream(grad))
                 : tvmi_ps_roi_pool_backward(grad_.const_data_ptr(), rois_.const_data_ptr(),
                                             map_.const_data_ptr<int32_t>(), grad_input.mutable_data_ptr(),
                                             dtype_of(grad, name), batch_size, channels, height, width, rois.size(0),
                                             pooled_height, pooled_width, spatial_scale, current_stream(grad));
  check_status(st, name);
  return grad_input;
}

std::tuple<at::Tensor, at::Tensor> ps_roi_align_forward(const at::Tensor& input, const at::Tensor& rois,
                                                        double spatial_scale, int64_t pooled_height,
                                                        int64_t pooled_width, int64_t sampling_ratio) {
  return ps_roi_common_forward("ps_roi_align_forward_kernel", true, input, rois, spatial_scale, pooled_height,
                               pooled_width, sampling_ratio);
}
at::Tensor ps_roi_align_backward(const at::Tensor& grad, const at::Tensor& rois, const at::Tensor& channel_mapping,
                                 double spatial_scale, int64_t pooled_height, int64_t pooled_width,
                                 int64_t sampling_ratio, int64_t batch_size, int64_t channels, int64_t height,
                                 int64_t width) {
  return ps_roi_common_backward("ps_roi_align_backward_kernel", true, grad, rois, channel_mapping, spatial_scale,
                                pooled_height, pooled_width, sampling_ratio, batch_size, channels, height, width);
}
std::tuple<at::Tensor, at::Tensor> ps_roi_pool_forward(const at::Tensor& input, const at::Tensor& rois,
                                                       double spatial_scale, int64_t pooled_height,
                                                       int64_t pooled_width) {
  return ps_roi_common_forward("ps_roi_pool_forward_kernel", false, input, rois, spatial_scale, pooled_height,
                               pooled_width, 0);
}
at::Tensor ps_roi_pool_backward(const at::Tensor& grad, const at::Tensor& rois, const at::Tensor& channel_mapping,
                                double spatial_scale, int64_t pooled_height, int64_t pooled_width,
                                int64_t batch_size, int64_t channels, int64_t height, int64_t width) {
  return ps_roi_common_backward("ps_roi_pool_backward_kernel", false, grad, rois, channel_mapping, spatial_scale,
                                pooled_height, pooled_width, 0, batch_size, channels, height, width);
}

// ---- deform_conv2d: checks/messages of cpu/deform_conv2d_kernel.cpp:921-1046 (= cuda :1035-1160)
struct DcnShape {
  int64_t B, C, H, W, OC, kh, kw, oh, ow;
};

DcnShape dcn_check(const at::Tensor& input, const at::Tensor& weight, const at::Tensor& offset,
                   const at::Tensor& mask, const at::Tensor& bias, int64_t stride_h, int64_t stride_w,
                   int64_t pad_h, int64_t pad_w, int64_t dilation_h, int64_t dilation_w, int64_t n_weight_grps,
                   int64_t n_offset_grps, bool use_mask) {
  TORCH_CHECK(input.dim() == 4);
  TORCH_CHECK(offset.dim() == 4);
  TORCH_CHECK(!use_mask || mask.dim() == 4);
  TORCH_CHECK(weight.dim() == 4);
  TORCH_CHECK(input.is_cuda(), "input must be a CUDA tensor");
  DcnShape s;
  s.B = input.size(0);
  s.C = input.size(1);
  s.H = input.size(2);
  s.W = input.size(3);
  s.OC = weight.size(0);
  s.kh = weight.size(2);
  s.kw = weight.size(3);
  const int64_t ker_h = dilation_h * (s.kh - 1) + 1, ker_w = dilation_w * (s.kw - 1) + 1;
  s.oh = ((s.H + 2 * pad_h - ker_h) / stride_h) + 1;
  s.ow = ((s.W + 2 * pad_w - ker_w) / stride_w) + 1;
  TORCH_CHECK(s.kh > 0 && s.kw > 0, "weight_h: ", s.kh, " weight_w: ", s.kw);
  TORCH_CHECK(stride_h > 0 && stride_w > 0, "stride_h: ", stride_h, " stride_w: ", stride_w);
  TORCH_CHECK(pad_h >= 0 && pad_w >= 0, "pad_h: ", pad_h, " pad_w: ", pad_w);
  TORCH_CHECK(dilation_h > 0 && dilation_w > 0, "dilation_h: ", dilation_h, " dilation_w: ", dilation_w);
  TORCH_CHECK(weight.size(1) * n_weight_grps == input.size(1));
  TORCH_CHECK(weight.size(0) % n_weight_grps == 0);
  TORCH_CHECK((offset.size(1) == n_offset_grps * 2 * s.kh * s.kw), "offset.shape[1] is not valid: got: ",
              offset.size(1), " expected: ", n_offset_grps * 2 * s.kh * s.kw);
  TORCH_CHECK((!use_mask || mask.size(1) == n_offset_grps * s.kh * s.kw), "mask.shape[1] is not valid: got: ",
              mask.size(1), " expected: ", n_offset_grps * s.kh * s.kw);
  TORCH_CHECK(input.size(1) % n_offset_grps == 0);
  TORCH_CHECK((offset.size(0) == input.size(0)), "invalid batch size of offset");
  TORCH_CHECK((offset.size(2) == s.oh && offset.size(3) == s.ow), "offset output dims: (", offset.size(2), ", ",
              offset.size(3), ") - ", "computed output dims: (", s.oh, ", ", s.ow, ")");
  TORCH_CHECK((mask.size(0) == input.size(0)), "invalid batch size of mask");
  TORCH_CHECK((!use_mask || (mask.size(2) == s.oh && mask.size(3) == s.ow)), "mask output dims: (", mask.size(2),
              ", ", mask.size(3), ") - ", "computed output dims: (", s.oh, ", ", s.ow, ")");
  TORCH_CHECK(s.oh > 0 && s.ow > 0, "Calculated output size too small - out_h: ", s.oh, " out_w: ", s.ow);
  return s;
}

at::Tensor deform_conv2d_forward(const at::Tensor& input, const at::Tensor& weight, const at::Tensor& offset,
                                 const at::Tensor& mask, const at::Tensor& bias, int64_t stride_h,
                                 int64_t stride_w, int64_t pad_h, int64_t pad_w, int64_t dilation_h,
                                 int64_t dilation_w, int64_t n_weight_grps, int64_t n_offset_grps, bool use_mask) {
  at::Tensor input_c = input.contiguous(), offset_c = offset.contiguous(), weight_c = weight.contiguous();
  at::Tensor mask_c = mask.contiguous(), bias_c = bias.contiguous();
  const DcnShape s = dcn_check(input_c, weight_c, offset_c, mask_c, bias_c, stride_h, stride_w, pad_h, pad_w,
                               dilation_h, dilation_w, n_weight_grps, n_offset_grps, use_mask);
  c10::DeviceGuard guard(input.device());
  // 16-bit tensors run natively: v_mfma_f32_32x32x16_{f16,bf16} with fp32 accumulation for real channel contractions
  // (the sampled values are rounded to 16 bits like the reference's `columns`, the sum stays fp32), the lane = pixel LDS
  // kernel for depthwise 3x3, the direct kernel otherwise — no widening copies.
  at::Tensor out = at::empty({s.B, s.OC, s.oh, s.ow}, input_c.options());
  if (out.numel() == 0) return out;
  const tvmi_dtype dt = dtype_of(input_c, "deform_conv2d");
  const size_t ws_bytes = tvmi_deform_conv2d_forward_workspace_bytes(dt, s.B, s.C, s.H, s.W, s.OC, s.kh, s.kw, n_weight_grps,
                                                                     n_offset_grps);
  at::Tensor ws = at::empty({(int64_t)ws_bytes}, input_c.options().dtype(at::kByte));
  check_status(tvmi_deform_conv2d_forward(input_c.const_data_ptr(), weight_c.const_data_ptr(),
                                          offset_c.const_data_ptr(), mask_c.const_data_ptr(), bias_c.const_data_ptr(),
                                          out.mutable_data_ptr(), dt, s.B, s.C, s.H, s.W, s.OC, s.kh, s.kw, stride_h,
                                          stride_w, pad_h, pad_w, dilation_h, dilation_w, n_weight_grps, n_offset_grps,
                                          use_mask ? 1 : 0, ws.mutable_data_ptr(), ws_bytes, current_stream(input)),
               "deform_conv2d");
  return out;
}

// Backward (cpu/deform_conv2d_kernel.cpp:1153-1226, cuda/deform_conv2d_kernel.cu:752-1033,1257-1330): ONE call into the kernels
// library — two fused matrix-core kernels (or the direct ones), no materialised `columns`, no library GEMM
// (deform_conv2d_bwd.hip).
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> deform_conv2d_backward(
    const at::Tensor& grad, const at::Tensor& input, const at::Tensor& weight, const at::Tensor& offset,
    const at::Tensor& mask, const at::Tensor& bias, int64_t stride_h, int64_t stride_w, int64_t pad_h, int64_t pad_w,
    int64_t dilation_h, int64_t dilation_w, int64_t n_weight_grps, int64_t n_offset_grps, bool use_mask) {
  at::Tensor grad_c = grad.contiguous(), input_c = input.contiguous(), offset_c = offset.contiguous();
  at::Tensor weight_c = weight.contiguous(), mask_c = mask.contiguous(), bias_c = bias.contiguous();
  const DcnShape s = dcn_check(input_c, weight_c, offset_c, mask_c, bias_c, stride_h, stride_w, pad_h, pad_w,
                               dilation_h, dilation_w, n_weight_grps, n_offset_grps, use_mask);
  // the kernels read grad through its raw pointer as [B, OC, oh, ow] of the input's type
  TORCH_CHECK(grad.is_cuda() && grad.device() == input.device(), "_deform_conv2d_backward: grad must be on the input's device");
  TORCH_CHECK(grad.scalar_type() == input.scalar_type(), "_deform_conv2d_backward: grad must have the input's dtype, got ",
              grad.scalar_type(), " and ", input.scalar_type());
  TORCH_CHECK(grad.dim() == 4 && grad.size(0) == s.B && grad.size(1) == s.OC && grad.size(2) == s.oh && grad.size(3) == s.ow,
              "_deform_conv2d_backward: grad must be [", s.B, ", ", s.OC, ", ", s.oh, ", ", s.ow, "], got ", grad.sizes());
  c10::DeviceGuard guard(input.device());
  at::globalContext().alertNotDeterministic("deform_conv2d_backward_kernel");
  // every output is fully overwritten by the call (it zero-fills what it accumulates into)
  at::Tensor grad_input = at::empty_like(input_c), grad_offset = at::empty_like(offset_c);
  at::Tensor grad_mask = use_mask ? at::empty_like(mask_c) : at::zeros_like(mask_c);
  at::Tensor grad_weight = at::empty_like(weight_c);
  at::Tensor grad_bias = at::ones_like(bias_c);
  if (s.B == 0 || input_c.numel() == 0 || weight_c.numel() == 0)
    // grad_bias = ones * grad.sum((0, 2, 3)) in the reference (cpu/deform_conv2d_kernel.cpp:1211-1213): zero for an empty problem
    return std::make_tuple(at::zeros_like(input_c), at::zeros_like(weight_c), at::zeros_like(offset_c), at::zeros_like(mask_c),
                           at::zeros_like(bias_c));
  const tvmi_dtype dt = dtype_of(input_c, "_deform_conv2d_backward");
  const size_t ws_bytes = tvmi_deform_conv2d_backward_workspace_bytes(dt, s.B, s.C, s.H, s.W, s.OC, s.kh, s.kw, stride_h, stride_w,
                                                                      pad_h, pad_w, dilation_h, dilation_w, n_weight_grps,
                                                                      n_offset_grps);
  at::Tensor ws = at::empty({(int64_t)ws_bytes}, input_c.options().dtype(at::kByte));
  check_status(tvmi_deform_conv2d_backward(grad_c.const_data_ptr(), input_c.const_data_ptr(), weight_c.const_data_ptr(),
                                           offset_c.const_data_ptr(), mask_c.const_data_ptr(), grad_input.mutable_data_ptr(),
                                           grad_weight.mutable_data_ptr(), grad_offset.mutable_data_ptr(),
                                           grad_mask.mutable_data_ptr(), grad_bias.mutable_data_ptr(), dt, s.B, s.C, s.H, s.W,
                                           s.OC, s.kh, s.kw, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                                           n_weight_grps, n_offset_grps, use_mask ? 1 : 0,
                                           ws_bytes ? ws.mutable_data_ptr() : nullptr, ws_bytes, current_stream(input)),
               "_deform_conv2d_backward");
  return std::make_tuple(grad_input, grad_weight, grad_offset, grad_mask, grad_bias);
}

// ---- box_iou_rotated, nms: stable-ABI kernels, torch_shim_stable.cpp (as in the reference: cuda/box_iou_rotated_kernel.cu:192,
// cuda/nms_kernel.cu:262)

// ---- resize (aten::upsample_* arithmetic; see include/tvmi.h).  mode: 0 nearest, 1 nearest-exact,
// 2 bilinear, 3 bicubic.  Input [N,C,H,W] (any strides; made contiguous), output [N,C,OH,OW].
at::Tensor interpolate2d(const at::Tensor& input, int64_t out_h, int64_t out_w, int64_t mode, bool align_corners,
                         bool antialias, double scale_h, double scale_w) {
  TORCH_CHECK(input.is_cuda(), "input must be a CUDA tensor");
  TORCH_CHECK(input.dim() == 4, "interpolate2d expects a 4d tensor [N, C, H, W]");
  TORCH_CHECK(out_h >= 0 && out_w >= 0, "output size must be non-negative");
  TORCH_CHECK(mode >= 0 && mode <= 3, "unknown interpolation mode ", mode);
  TORCH_CHECK(!antialias || mode >= 2, "Anti-alias option is restricted to bilinear and bicubic modes");
  TORCH_CHECK((input.size(2) > 0 && input.size(3) > 0) || input.numel() == 0,
              "Input and output sizes should be greater than 0");
  c10::DeviceGuard guard(input.device());
  const int64_t N = input.size(0), C = input.size(1), IH = input.size(2), IW = input.size(3);
  void* stream = current_stream(input);
  const auto stype = input.scalar_type();
  const bool is_fp = stype == at::kFloat || stype == at::kDouble || stype == at::kHalf || stype == at::kBFloat16;
  // channels_last in, channels_last out (ATen's `suggest_memory_format` rule), no layout copy: the NHWC kernels of resize.hip
  if (!input.is_contiguous() && input.is_contiguous(at::MemoryFormat::ChannelsLast) && input.numel() > 0 && out_h > 0 && out_w > 0 &&
      out_h <= 65535 && N <= 65535 && out_w * C < (1ll << 31) && (stype == at::kFloat || stype == at::kHalf || stype == at::kBFloat16 || (mode <= 1 && !is_fp))) {
    at::Tensor out = at::empty({N, C, out_h, out_w}, input.options().memory_format(at::MemoryFormat::ChannelsLast));
    if (mode <= 1 && !is_fp) {
      check_status(tvmi_upsample_nearest2d_nhwc_any(input.const_data_ptr(), out.mutable_data_ptr(), (int64_t)input.element_size(), N, C, IH,
                                                    IW, out_h, out_w, mode == 1, scale_h, scale_w, stream),
                   "interpolate2d");
      return out;
    }
    const size_t wb = tvmi_upsample2d_nhwc_workspace_bytes((int)mode, antialias, IH, IW, out_h, out_w, align_corners, scale_h, scale_w);
    at::Tensor ws = at::empty({(int64_t)wb}, input.options().dtype(at::kByte).memory_format(at::MemoryFormat::Contiguous));
    check_status(tvmi_upsample2d_nhwc(input.const_data_ptr(), out.mutable_data_ptr(), dtype_of(input, "interpolate2d"), (int)mode, antialias,
                                      N, C, IH, IW, out_h, out_w, align_corners, scale_h, scale_w, ws.mutable_data_ptr(), wb, stream),
                 "interpolate2d");
    return out;
  }
  // channels_last inputs the NHWC kernels do not take (float64, sizes beyond their grid limits): computed in NCHW and handed
  // back channels_last, so the output layout depends on the input layout alone (the fake kernel promises exactly that)
  if (!input.is_contiguous() && input.is_contiguous(at::MemoryFormat::ChannelsLast) && input.numel() > 0 && out_h > 0 && out_w > 0)
    return interpolate2d(input.contiguous(), out_h, out_w, mode, align_corners, antialias, scale_h, scale_w)
        .contiguous(at::MemoryFormat::ChannelsLast);
  at::Tensor in_c = input.contiguous();
  at::Tensor out = at::empty({N, C, out_h, out_w}, in_c.options());
  if (out.numel() == 0) return out;
  if (mode <= 1 && !is_fp) {
    // nearest / nearest-exact are copies: any element type (uint8 images and masks, _geometry.py:316-323), by element size
    check_status(tvmi_upsample_nearest2d_any(in_c.const_data_ptr(), out.mutable_data_ptr(), (int64_t)in_c.element_size(), N * C, IH,
                                             IW, out_h, out_w, mode == 1, scale_h, scale_w, stream),
                 "interpolate2d");
    return out;
  }
  const tvmi_dtype dt = dtype_of(in_c, "interpolate2d");
  int st = 0;
  if (mode <= 1) {
    st = tvmi_upsample_nearest2d(in_c.const_data_ptr(), out.mutable_data_ptr(), dt, N * C, IH, IW, out_h, out_w,
                                 mode == 1, scale_h, scale_w, stream);
  } else if (antialias) {
    const size_t wb = tvmi_upsample_aa2d_workspace_bytes((int)mode - 2, IH, IW, out_h, out_w, align_corners, scale_h,
                                                         scale_w);
    at::Tensor ws = at::empty({(int64_t)wb}, in_c.options().dtype(at::kByte));
    st = tvmi_upsample_aa2d(in_c.const_data_ptr(), out.mutable_data_ptr(), dt, (int)mode - 2, N * C, IH, IW, out_h,
                            out_w, align_corners, scale_h, scale_w, ws.mutable_data_ptr(), wb, stream);
  } else if (mode == 2) {
    st = tvmi_upsample_bilinear2d(in_c.const_data_ptr(), out.mutable_data_ptr(), dt, N * C, IH, IW, out_h, out_w,
                                  align_corners, scale_h, scale_w, stream);
  } else {
    st = tvmi_upsample_bicubic2d(in_c.const_data_ptr(), out.mutable_data_ptr(), dt, N * C, IH, IW, out_h, out_w,
                                 align_corners, scale_h, scale_w, stream);
  }
  check_status(st, "interpolate2d");
  return out;
}

// Gradient of interpolate2d: grad_output [N,C,out_h,out_w] -> grad_input [N,C,in_h,in_w] (gather form, deterministic;
// include/tvmi.h: tvmi_upsample2d_backward).  float32 / float16 / bfloat16.
at::Tensor interpolate2d_backward(const at::Tensor& grad_output, int64_t in_h, int64_t in_w, int64_t mode, bool align_corners,
                                  bool antialias, double scale_h, double scale_w) {
  TORCH_CHECK(grad_output.is_cuda(), "grad_output must be a CUDA tensor");
  TORCH_CHECK(grad_output.dim() == 4, "interpolate2d_backward expects a 4d tensor [N, C, H, W]");
  TORCH_CHECK(in_h >= 0 && in_w >= 0, "input size must be non-negative");
  TORCH_CHECK(mode >= 0 && mode <= 3, "unknown interpolation mode ", mode);
  TORCH_CHECK(!antialias || mode >= 2, "Anti-alias option is restricted to bilinear and bicubic modes");
  const auto t = grad_output.scalar_type();
  TORCH_CHECK(t == at::kFloat || t == at::kHalf || t == at::kBFloat16, "interpolate2d_backward: float32 / float16 / bfloat16 only");
  c10::DeviceGuard guard(grad_output.device());
  at::Tensor g = grad_output.contiguous();
  const int64_t N = g.size(0), C = g.size(1), OH = g.size(2), OW = g.size(3);
  at::Tensor grad_input = at::empty({N, C, in_h, in_w}, g.options());
  if (grad_input.numel() == 0) return grad_input;
  if (g.numel() == 0) return grad_input.zero_();
  const size_t wb = tvmi_upsample2d_backward_workspace_bytes((int)mode, antialias, in_h, in_w, OH, OW, align_corners, scale_h, scale_w);
  at::Tensor ws = at::empty({(int64_t)wb}, g.options().dtype(at::kByte));
  check_status(tvmi_upsample2d_backward(g.const_data_ptr(), grad_input.mutable_data_ptr(), dtype_of(g, "interpolate2d_backward"),
                                        (int)mode, antialias, N * C, in_h, in_w, OH, OW, align_corners, scale_h, scale_w,
                                        ws.mutable_data_ptr(), wb, current_stream(g)),
               "interpolate2d_backward");
  return grad_input;
}

// ---- the resize boundary of the reference: torchvision resizes through F.interpolate
// (models/detection/transform.py:65-72, ops/feature_pyramid_network.py:194, models/detection/roi_heads.py:427,453,
// transforms/v2/functional/_geometry.py:344-350), i.e. through aten::upsample_*.  `tvmi::override_aten_upsample(True)`
// (opt-in; TVMI_OVERRIDE_ATEN_UPSAMPLE=1 at import) puts our kernels on the CUDA key of those aten ops, functional and
// .out overloads, so the unchanged reference python lands in resize.hip.  Round 5: the six `*_backward` ops are taken over
// too (gather kernels, deterministic), the nearest modes serve integer / bool tensors and channels_last inputs have their
// own kernels; inputs our kernels do not serve (other strides, integer dtypes in the interpolating modes, float64, 0-sized)
// go to ATen's own CUDA kernel through its static-dispatch entry.  The handle is dropped again by override_aten_upsample(False).
namespace upsample_override {

std::unique_ptr<torch::Library> g_lib;
std::mutex g_mutex;
std::atomic<int64_t> g_calls{0};  // how many aten::upsample_* calls were served by tvmi kernels (tests read it)

// What the override serves itself.  Everything else falls through to ATen's own kernel: float64 (resize.hip computes scales,
// weights and sums in float — fine for the 16/32-bit types, but a process-wide override must not cost fp64 callers their
// precision, e.g. gradcheck), strided inputs other than channels_last, integer types outside the nearest modes, empty tensors, and shapes beyond the limits
// of the resize launchers (TVMI_RESIZE_PROLOGUE in resize.hip: OH <= 65535, IH*IW and OH*OW below 2^31) — those must reach
// ATen, not raise "size too large".
bool ours(const at::Tensor& self, at::IntArrayRef size, bool nearest = false) {
  const auto t = self.scalar_type();
  // nearest modes are copies: every non-complex, non-quantized element type of 1 / 2 / 4 bytes or int64 is served
  const bool any_ok = nearest && (c10::isIntegralType(t, /*includeBool=*/true));
  if (!(self.is_cuda() && self.dim() == 4 && size.size() == 2 && self.numel() > 0 && size[0] > 0 && size[1] > 0 &&
        (t == at::kFloat || t == at::kHalf || t == at::kBFloat16 || any_ok)))
    return false;
  const int64_t IH = self.size(2), IW = self.size(3), OH = size[0], OW = size[1];
  if (!self.is_contiguous())   // channels_last: the NHWC kernels (output channels_last, like ATen); other strides: ATen
    return self.is_contiguous(at::MemoryFormat::ChannelsLast) && OH <= 65535 && self.size(0) <= 65535 && OW * self.size(1) < (1ll << 31);
  return OH <= 65535 && IH * IW < (1ll << 31) && OH * OW < (1ll << 31);
}

double sc(const std::optional<double>& v) { return v.has_value() ? *v : -1.0; }

#define TVMI_UPSAMPLE_LINEAR(NAME, ATEN, MODE, AA)                                                                      \
  at::Tensor NAME(const at::Tensor& self, at::IntArrayRef size, bool align_corners, std::optional<double> sh,          \
                  std::optional<double> sw) {                                                                          \
    if (!ours(self, size)) return at::cuda::ATEN(self, size, align_corners, sh, sw);                                   \
    ++g_calls;                                                                                                         \
    return interpolate2d(self, size[0], size[1], MODE, align_corners, AA, sc(sh), sc(sw));                             \
  }                                                                                                                    \
  at::Tensor& NAME##_out(const at::Tensor& self, at::IntArrayRef size, bool align_corners, std::optional<double> sh,   \
                         std::optional<double> sw, at::Tensor& out) {                                                  \
    if (!ours(self, size)) return at::cuda::ATEN##_outf(self, size, align_corners, sh, sw, out);                       \
    ++g_calls;                                                                                                         \
    at::Tensor r = interpolate2d(self, size[0], size[1], MODE, align_corners, AA, sc(sh), sc(sw));                     \
    at::native::resize_output(out, r.sizes());                                                                         \
    out.copy_(r);                                                                                                      \
    return out;                                                                                                        \
  }
#define TVMI_UPSAMPLE_NEAREST(NAME, ATEN, MODE)                                                                        \
  at::Tensor NAME(const at::Tensor& self, at::IntArrayRef size, std::optional<double> sh, std::optional<double> sw) {  \
    if (!ours(self, size, true)) return at::cuda::ATEN(self, size, sh, sw);                                            \
    ++g_calls;                                                                                                         \
    return interpolate2d(self, size[0], size[1], MODE, false, false, sc(sh), sc(sw));                                  \
  }                                                                                                                    \
  at::Tensor& NAME##_out(const at::Tensor& self, at::IntArrayRef size, std::optional<double> sh,                       \
                         std::optional<double> sw, at::Tensor& out) {                                                  \
    if (!ours(self, size, true)) return at::cuda::ATEN##_outf(self, size, sh, sw, out);                                \
    ++g_calls;                                                                                                         \
    at::Tensor r = interpolate2d(self, size[0], size[1], MODE, false, false, sc(sh), sc(sw));                          \
    at::native::resize_output(out, r.sizes());                                                                         \
    out.copy_(r);                                                                                                      \
    return out;                                                                                                        \
  }

TVMI_UPSAMPLE_LINEAR(bilinear, upsample_bilinear2d, 2, false)
TVMI_UPSAMPLE_LINEAR(bicubic, upsample_bicubic2d, 3, false)
TVMI_UPSAMPLE_LINEAR(bilinear_aa, _upsample_bilinear2d_aa, 2, true)
TVMI_UPSAMPLE_LINEAR(bicubic_aa, _upsample_bicubic2d_aa, 3, true)
TVMI_UPSAMPLE_NEAREST(nearest, upsample_nearest2d, 0)
TVMI_UPSAMPLE_NEAREST(nearest_exact, _upsample_nearest_exact2d, 1)
#undef TVMI_UPSAMPLE_LINEAR
#undef TVMI_UPSAMPLE_NEAREST

// The backward ops (what autograd calls for the six forwards above): served for float32 / float16 / bfloat16 gradients of
// 4-d problems within the launch limits (made contiguous first, grad_input contiguous), everything else reaches ATen's kernel.
bool ours_bwd(const at::Tensor& g, at::IntArrayRef out_size, at::IntArrayRef in_size) {
  const auto t = g.scalar_type();
  if (!(g.is_cuda() && g.dim() == 4 && out_size.size() == 2 && in_size.size() == 4 && g.numel() > 0 &&
        (t == at::kFloat || t == at::kHalf || t == at::kBFloat16)))   // any strides: `sum().backward()` hands an expanded grad
    return false;
  // channels_last gradients keep ATen's channels_last backward kernels (ours would pay a layout copy and return NCHW)
  if (!g.is_contiguous() && g.is_contiguous(at::MemoryFormat::ChannelsLast)) return false;
  const int64_t IH = in_size[2], IW = in_size[3], OH = out_size[0], OW = out_size[1];
  return g.size(0) == in_size[0] && g.size(1) == in_size[1] && g.size(2) == OH && g.size(3) == OW && IH > 0 && IW > 0 &&
         IH <= 65535 && IH * IW < (1ll << 31) && OH * OW < (1ll << 31);
}

#define TVMI_UPSAMPLE_LINEAR_BWD(NAME, ATEN, MODE, AA)                                                                  \
  at::Tensor NAME(const at::Tensor& g, at::IntArrayRef osz, at::IntArrayRef isz, bool align_corners,                   \
                  std::optional<double> sh, std::optional<double> sw) {                                                \
    if (!ours_bwd(g, osz, isz)) return at::cuda::ATEN(g, osz, isz, align_corners, sh, sw);                             \
    ++g_calls;                                                                                                         \
    return interpolate2d_backward(g, isz[2], isz[3], MODE, align_corners, AA, sc(sh), sc(sw));                         \
  }                                                                                                                    \
  at::Tensor& NAME##_out(const at::Tensor& g, at::IntArrayRef osz, at::IntArrayRef isz, bool align_corners,            \
                         std::optional<double> sh, std::optional<double> sw, at::Tensor& grad_input) {                 \
    if (!ours_bwd(g, osz, isz)) return at::cuda::ATEN##_outf(g, osz, isz, align_corners, sh, sw, grad_input);          \
    ++g_calls;                                                                                                         \
    at::Tensor r = interpolate2d_backward(g, isz[2], isz[3], MODE, align_corners, AA, sc(sh), sc(sw));                 \
    at::native::resize_output(grad_input, r.sizes());                                                                  \
    grad_input.copy_(r);                                                                                               \
    return grad_input;                                                                                                 \
  }
#define TVMI_UPSAMPLE_NEAREST_BWD(NAME, ATEN, MODE)                                                                    \
  at::Tensor NAME(const at::Tensor& g, at::IntArrayRef osz, at::IntArrayRef isz, std::optional<double> sh,             \
                  std::optional<double> sw) {                                                                          \
    if (!ours_bwd(g, osz, isz)) return at::cuda::ATEN(g, osz, isz, sh, sw);                                            \
    ++g_calls;                                                                                                         \
    return interpolate2d_backward(g, isz[2], isz[3], MODE, false, false, sc(sh), sc(sw));                              \
  }                                                                                                                    \
  at::Tensor& NAME##_out(const at::Tensor& g, at::IntArrayRef osz, at::IntArrayRef isz, std::optional<double> sh,      \
                         std::optional<double> sw, at::Tensor& grad_input) {                                           \
    if (!ours_bwd(g, osz, isz)) return at::cuda::ATEN##_outf(g, osz, isz, sh, sw, grad_input);                         \
    ++g_calls;                                                                                                         \
    at::Tensor r = interpolate2d_backward(g, isz[2], isz[3], MODE, false, false, sc(sh), sc(sw));                      \
    at::native::resize_output(grad_input, r.sizes());                                                                  \
    grad_input.copy_(r);                                                                                               \
    return grad_input;                                                                                                 \
  }

TVMI_UPSAMPLE_LINEAR_BWD(bilinear_bwd, upsample_bilinear2d_backward, 2, false)
TVMI_UPSAMPLE_LINEAR_BWD(bicubic_bwd, upsample_bicubic2d_backward, 3, false)
TVMI_UPSAMPLE_LINEAR_BWD(bilinear_aa_bwd, _upsample_bilinear2d_aa_backward, 2, true)
TVMI_UPSAMPLE_LINEAR_BWD(bicubic_aa_bwd, _upsample_bicubic2d_aa_backward, 3, true)
TVMI_UPSAMPLE_NEAREST_BWD(nearest_bwd, upsample_nearest2d_backward, 0)
TVMI_UPSAMPLE_NEAREST_BWD(nearest_exact_bwd, _upsample_nearest_exact2d_backward, 1)
#undef TVMI_UPSAMPLE_LINEAR_BWD
#undef TVMI_UPSAMPLE_NEAREST_BWD

bool set(bool enable) {
  std::lock_guard<std::mutex> lock(g_mutex);
  const bool was = g_lib != nullptr;
  if (enable && !g_lib) {
    g_lib = std::make_unique<torch::Library>(torch::Library::IMPL, "aten", c10::make_optional(c10::DispatchKey::CUDA), __FILE__,
                                             __LINE__);
    g_lib->impl("upsample_bilinear2d", &bilinear);
    g_lib->impl("upsample_bilinear2d.out", &bilinear_out);
    g_lib->impl("upsample_bicubic2d", &bicubic);
    g_lib->impl("upsample_bicubic2d.out", &bicubic_out);
    g_lib->impl("_upsample_bilinear2d_aa", &bilinear_aa);
    g_lib->impl("_upsample_bilinear2d_aa.out", &bilinear_aa_out);
    g_lib->impl("_upsample_bicubic2d_aa", &bicubic_aa);
    g_lib->impl("_upsample_bicubic2d_aa.out", &bicubic_aa_out);
    g_lib->impl("upsample_nearest2d", &nearest);
    g_lib->impl("upsample_nearest2d.out", &nearest_out);
    g_lib->impl("_upsample_nearest_exact2d", &nearest_exact);
    g_lib->impl("_upsample_nearest_exact2d.out", &nearest_exact_out);
    g_lib->impl("upsample_bilinear2d_backward", &bilinear_bwd);
    g_lib->impl("upsample_bilinear2d_backward.grad_input", &bilinear_bwd_out);
    g_lib->impl("upsample_bicubic2d_backward", &bicubic_bwd);
    g_lib->impl("upsample_bicubic2d_backward.grad_input", &bicubic_bwd_out);
    g_lib->impl("_upsample_bilinear2d_aa_backward", &bilinear_aa_bwd);
    g_lib->impl("_upsample_bilinear2d_aa_backward.grad_input", &bilinear_aa_bwd_out);
    g_lib->impl("_upsample_bicubic2d_aa_backward", &bicubic_aa_bwd);
    g_lib->impl("_upsample_bicubic2d_aa_backward.grad_input", &bicubic_aa_bwd_out);
    g_lib->impl("upsample_nearest2d_backward", &nearest_bwd);
    g_lib->impl("upsample_nearest2d_backward.grad_input", &nearest_bwd_out);
    g_lib->impl("_upsample_nearest_exact2d_backward", &nearest_exact_bwd);
    g_lib->impl("_upsample_nearest_exact2d_backward.grad_input", &nearest_exact_bwd_out);
  } else if (!enable) {
    g_lib.reset();
  }
  return was;
}

int64_t calls() { return g_calls.load(); }

}  // namespace upsample_override

// ---- multi-scale RoIAlign in one launch (torchvision/ops/poolers.py:147-227)
at::Tensor multiscale_roi_align(at::TensorList features, const at::Tensor& rois, at::ArrayRef<double> scales,
                                int64_t pooled_height, int64_t pooled_width, int64_t sampling_ratio, bool aligned,
                                int64_t k_min, int64_t k_max, double canonical_scale, double canonical_level,
                                double eps) {
  TORCH_CHECK(features.size() >= 1 && features.size() <= 8, "multiscale_roi_align: 1..8 feature levels supported");
  TORCH_CHECK(features.size() == scales.size(), "multiscale_roi_align: one scale per feature level");
  TORCH_CHECK(rois.is_cuda() && rois.dim() == 2 && rois.size(1) == 5, "rois must be a CUDA tensor of shape [K, 5]");
  const at::Tensor& f0 = features[0];
  TORCH_CHECK(f0.is_cuda() && f0.dim() == 4, "features must be 4d CUDA tensors");
  c10::DeviceGuard guard(f0.device());
  std::vector<at::Tensor> keep;
  std::vector<const void*> ptrs;
  std::vector<int64_t> hs, ws;
  // channels_last maps (NHWC in memory) have their own kernel: lane = channel, no layout copy (the reference's
  // input.contiguous() would rewrite every map)
  const bool low = f0.scalar_type() == at::kHalf || f0.scalar_type() == at::kBFloat16;
  bool nhwc = (f0.scalar_type() == at::kFloat || (low && f0.size(1) % 2 == 0)) && pooled_height == 7 && pooled_width == 7 &&
              sampling_ratio == 2 && f0.size(1) > 1;
  for (const at::Tensor& f : features)
    nhwc = nhwc && f.dim() == 4 && !f.is_contiguous() && f.is_contiguous(at::MemoryFormat::ChannelsLast) && f.size(2) >= 2 &&
           f.size(3) >= 2 && f.numel() / std::max<int64_t>(f.size(0), 1) < (1ll << 31);
  if (nhwc) {
    for (const at::Tensor& f : features) {
      TORCH_CHECK(f.is_cuda() && f.size(0) == f0.size(0) && f.size(1) == f0.size(1) && f.scalar_type() == f0.scalar_type() &&
                      f.device() == f0.device(),
                  "multiscale_roi_align: feature levels must share device, dtype, batch and channel sizes");
      ptrs.push_back(f.const_data_ptr());
      hs.push_back(f.size(2));
      ws.push_back(f.size(3));
    }
    at::Tensor r = rois.to(at::kFloat).contiguous();
    at::Tensor out = at::empty({rois.size(0), f0.size(1), pooled_height, pooled_width}, f0.options().memory_format(at::MemoryFormat::Contiguous));
    if (out.numel() == 0) return out;
    at::Tensor order_ws = at::empty({rois.size(0)}, f0.options().dtype(at::kInt));   // launch order of the RoIs
    check_status(tvmi_multiscale_roi_align_forward_nhwc(ptrs.data(), hs.data(), ws.data(), scales.data(),
                                                        (int64_t)features.size(), r.const_data_ptr(), out.mutable_data_ptr(),
                                                        dtype_of(f0, "multiscale_roi_align"), f0.size(0), f0.size(1), rois.size(0), pooled_height,
                                                        pooled_width, sampling_ratio, aligned ? 1 : 0, k_min, k_max,
                                                        canonical_scale, canonical_level, eps, order_ws.mutable_data_ptr(),
                                                        (size_t)rois.size(0) * sizeof(int), current_stream(f0)),
                 "multiscale_roi_align");
    return out;
  }
  for (const at::Tensor& f : features) {
    TORCH_CHECK(f.is_cuda() && f.dim() == 4 && f.size(0) == f0.size(0) && f.size(1) == f0.size(1) &&
                    f.scalar_type() == f0.scalar_type() && f.device() == f0.device(),
                "multiscale_roi_align: feature levels must share device, dtype, batch and channel sizes");
    keep.push_back(f.contiguous());
    ptrs.push_back(keep.back().const_data_ptr());
    hs.push_back(f.size(2));
    ws.push_back(f.size(3));
  }
  // RoIs stay float32 whatever the feature dtype: levels and sample coordinates are computed from the fp32 boxes,
  // as the reference does (poolers.py:199-222; under autocast _autograd_registrations.py:246 casts them to fp32)
  at::Tensor rois_ = rois.to(at::kFloat).contiguous();
  const int64_t K = rois.size(0), C = f0.size(1);
  at::Tensor output = at::empty({K, C, pooled_height, pooled_width}, f0.options());
  if (output.numel() == 0) return output;
  // "declined by the LDS-DMA kernel" flags + the tables of the shared-staging kernel
  const size_t fwd_ws_bytes = tvmi_roi_align_forward_workspace_bytes(K, pooled_height, pooled_width, sampling_ratio);
  at::Tensor order_ws = at::empty({(int64_t)fwd_ws_bytes}, f0.options().dtype(at::kByte));
  check_status(tvmi_multiscale_roi_align_forward(ptrs.data(), hs.data(), ws.data(), scales.data(),
                                                 (int64_t)features.size(), rois_.const_data_ptr(),
                                                 output.mutable_data_ptr(), dtype_of(f0, "multiscale_roi_align"),
                                                 f0.size(0), C, K, pooled_height, pooled_width, sampling_ratio,
                                                 aligned ? 1 : 0, k_min, k_max, canonical_scale, canonical_level, eps,
                                                 order_ws.mutable_data_ptr(), fwd_ws_bytes, current_stream(f0)),
               "multiscale_roi_align");
  return output;
}

// The same op taking the per-image box lists (the argument MultiScaleRoIAlign.forward receives, ops/poolers.py:289-321): the
// [K,5] rows of convert_boxes_to_roi_format are written by the launch-order pre-pass of the call (roi_align.hip, round 6) instead
// of a launch of their own.  Returns (output, rois): the rows are what the backward takes.  NCHW float32 / float16 / bfloat16
// maps and float32 boxes; everything else goes through boxes_to_rois + multiscale_roi_align in the python mirror.
std::tuple<at::Tensor, at::Tensor> multiscale_roi_align_boxes(at::TensorList features, at::TensorList boxes, at::ArrayRef<double> scales,
                                                              int64_t pooled_height, int64_t pooled_width, int64_t sampling_ratio,
                                                              bool aligned, int64_t k_min, int64_t k_max, double canonical_scale,
                                                              double canonical_level, double eps) {
  TORCH_CHECK(features.size() >= 1 && features.size() <= 8, "multiscale_roi_align: 1..8 feature levels supported");
  TORCH_CHECK(features.size() == scales.size(), "multiscale_roi_align: one scale per feature level");
  TORCH_CHECK(boxes.size() >= 1 && boxes.size() <= 64, "multiscale_roi_align_boxes: 1..64 box lists (one per image)");
  const at::Tensor& f0 = features[0];
  TORCH_CHECK(f0.is_cuda() && f0.dim() == 4, "features must be 4d CUDA tensors");
  c10::DeviceGuard guard(f0.device());
  std::vector<at::Tensor> keep, bkeep;
  std::vector<const void*> ptrs, bptrs;
  std::vector<int64_t> hs, ws, counts;
  int64_t K = 0;
  for (const at::Tensor& b : boxes) {
    TORCH_CHECK(b.is_cuda() && b.dim() == 2 && b.size(1) == 4 && b.scalar_type() == at::kFloat && b.device() == f0.device(),
                "multiscale_roi_align_boxes: float32 CUDA boxes [n_i, 4] expected");
    bkeep.push_back(b.contiguous());
    bptrs.push_back(b.size(0) ? bkeep.back().const_data_ptr() : nullptr);
    counts.push_back(b.size(0));
    K += b.size(0);
  }
  for (const at::Tensor& f : features) {
    TORCH_CHECK(f.is_cuda() && f.dim() == 4 && f.size(0) == f0.size(0) && f.size(1) == f0.size(1) &&
                    f.scalar_type() == f0.scalar_type() && f.device() == f0.device(),
                "multiscale_roi_align: feature levels must share device, dtype, batch and channel sizes");
    keep.push_back(f.contiguous());
    ptrs.push_back(keep.back().const_data_ptr());
    hs.push_back(f.size(2));
    ws.push_back(f.size(3));
  }
  const int64_t C = f0.size(1);
  at::Tensor rois = at::empty({K, 5}, f0.options().dtype(at::kFloat));
  at::Tensor output = at::empty({K, C, pooled_height, pooled_width}, f0.options());
  if (K == 0) return std::make_tuple(output, rois);
  const size_t fwd_ws_bytes = tvmi_roi_align_forward_workspace_bytes(K, pooled_height, pooled_width, sampling_ratio);
  at::Tensor order_ws = at::empty({(int64_t)fwd_ws_bytes}, f0.options().dtype(at::kByte));
  check_status(tvmi_multiscale_roi_align_forward_boxes(ptrs.data(), hs.data(), ws.data(), scales.data(), (int64_t)features.size(),
                                                       bptrs.data(), counts.data(), (int64_t)boxes.size(), rois.mutable_data_ptr(),
                                                       output.mutable_data_ptr(), dtype_of(f0, "multiscale_roi_align"), f0.size(0), C,
                                                       pooled_height, pooled_width, sampling_ratio, aligned ? 1 : 0, k_min, k_max,
                                                       canonical_scale, canonical_level, eps, order_ws.mutable_data_ptr(), fwd_ws_bytes,
                                                       current_stream(f0)),
               "multiscale_roi_align_boxes");
  return std::make_tuple(output, rois);
}

// backward of the fused multi-scale op: one launch scatters into every level's gradient map
std::vector<at::Tensor> multiscale_roi_align_backward(const at::Tensor& grad, const at::Tensor& rois, at::IntArrayRef heights,
                                                      at::IntArrayRef widths, at::ArrayRef<double> scales, int64_t batch_size,
                                                      int64_t pooled_height, int64_t pooled_width, int64_t sampling_ratio,
                                                      bool aligned, int64_t k_min, int64_t k_max, double canonical_scale,
                                                      double canonical_level, double eps) {
  TORCH_CHECK(grad.is_cuda() && rois.is_cuda() && grad.dim() == 4 && rois.dim() == 2 && rois.size(1) == 5 &&
                  grad.size(0) == rois.size(0),
              "multiscale_roi_align_backward: grad [K,C,PH,PW] and rois [K,5] CUDA tensors expected");
  TORCH_CHECK(heights.size() >= 1 && heights.size() <= 8 && heights.size() == widths.size() && heights.size() == scales.size(),
              "multiscale_roi_align_backward: 1..8 levels with one height / width / scale each");
  c10::DeviceGuard guard(grad.device());
  const bool low = grad.scalar_type() == at::kHalf || grad.scalar_type() == at::kBFloat16;
  TORCH_CHECK(grad.scalar_type() == at::kFloat || low, "multiscale_roi_align_backward: float32 / float16 / bfloat16 only");
  at::Tensor r = rois.to(at::kFloat).contiguous();
  const int64_t K = grad.size(0), C = grad.size(1);
  std::vector<int64_t> hs(heights.begin(), heights.end()), ws(widths.begin(), widths.end());
  size_t ws_bytes = tvmi_roi_align_backward_workspace_bytes(batch_size, K, pooled_height, pooled_width);
  {
    int64_t map_numel = 0;
    for (size_t i = 0; i < heights.size(); ++i) map_numel += batch_size * C * heights[i] * widths[i];
    if (!owner_workspace_is_reasonable(ws_bytes, grad, map_numel)) ws_bytes = 0;
  }
  // 16-bit gradients are read natively by the tile-owner kernel (fp32 accumulation, one rounding per pixel); when that
  // regime does not apply they are widened and take the fp32 atomic path (see roi_align_backward)
  auto run = [&](at::Tensor g) {
    if (ws_bytes != 0 && !(g.stride(3) == 1 && g.stride(2) == pooled_width && g.stride(1) == pooled_height * pooled_width))
      g = g.contiguous();
    const tvmi_dtype dt = dtype_of(g, "multiscale_roi_align_backward");
    const bool overwrites = ws_bytes != 0 && K != 0 && C != 0 && batch_size != 0 &&
                            tvmi_multiscale_roi_align_backward_overwrites(dt, batch_size, C, K, hs.data(), ws.data(), (int64_t)hs.size(),
                                                                          pooled_height, pooled_width, g.stride(1), g.stride(2),
                                                                          g.stride(3), ws_bytes) != 0;
    std::vector<at::Tensor> outs;
    if (!overwrites && g.scalar_type() != at::kFloat) return outs;   // caller widens
    std::vector<void*> ptrs;
    for (size_t i = 0; i < heights.size(); ++i) {
      outs.push_back(overwrites ? at::empty({batch_size, C, heights[i], widths[i]}, g.options())
                                : at::zeros({batch_size, C, heights[i], widths[i]}, g.options()));
      ptrs.push_back(outs.back().mutable_data_ptr());
    }
    if (g.numel() != 0 && batch_size != 0) {
      if (!overwrites) at::globalContext().alertNotDeterministic("roi_align_backward_kernel");
      at::Tensor wsp = at::empty({(int64_t)(overwrites ? ws_bytes : 0)}, g.options().dtype(at::kByte));
      check_status(tvmi_multiscale_roi_align_backward(g.const_data_ptr(), r.const_data_ptr(), ptrs.data(), hs.data(), ws.data(),
                                                      scales.data(), (int64_t)heights.size(), dt, batch_size, C, K, pooled_height,
                                                      pooled_width, sampling_ratio, aligned ? 1 : 0, k_min, k_max, canonical_scale,
                                                      canonical_level, eps, g.stride(0), g.stride(1), g.stride(2), g.stride(3),
                                                      overwrites ? wsp.mutable_data_ptr() : nullptr, overwrites ? ws_bytes : 0,
                                                      current_stream(grad)),
                   "multiscale_roi_align_backward");
    }
    return outs;
  };
  std::vector<at::Tensor> outs = run(grad);
  if (outs.empty() && heights.size() > 0) {   // 16-bit grads outside the owner regime
    outs = run(grad.to(at::kFloat));
    for (auto& o : outs) o = o.to(grad.scalar_type());
  }
  return outs;
}

// ---- detection payload packing (one launch; see include/tvmi.h)
std::tuple<at::Tensor, at::Tensor> pack_detections(const at::Tensor& boxes, const at::Tensor& scores,
                                                   const c10::optional<at::Tensor>& labels, const at::Tensor& image_idx,
                                                   const at::Tensor& keep, int64_t num_images, int64_t max_dets) {
  TORCH_CHECK(boxes.is_cuda() && scores.is_cuda() && image_idx.is_cuda() && keep.is_cuda(), "pack_detections: CUDA tensors expected");
  TORCH_CHECK(boxes.dim() == 2 && boxes.size(1) == 4 && scores.dim() == 1 && scores.size(0) == boxes.size(0) &&
                  image_idx.dim() == 1 && image_idx.size(0) == boxes.size(0) && keep.dim() == 1,
              "pack_detections: boxes [N,4], scores [N], image_idx [N], keep [M] expected");
  c10::DeviceGuard guard(boxes.device());
  at::Tensor b = boxes.to(at::kFloat).contiguous(), s = scores.to(at::kFloat).contiguous();
  at::Tensor ii = image_idx.to(at::kLong).contiguous(), k = keep.to(at::kLong).contiguous();
  at::Tensor lab;
  const int64_t* lab_ptr = nullptr;
  if (labels.has_value() && labels->defined()) {
    lab = labels->to(at::kLong).contiguous();
    lab_ptr = lab.const_data_ptr<int64_t>();
  }
  at::Tensor dets = at::empty({num_images, max_dets, 6}, b.options());
  at::Tensor counts = at::empty({num_images}, b.options().dtype(at::kInt));
  check_status(tvmi_pack_detections(b.const_data_ptr<float>(), s.const_data_ptr<float>(), lab_ptr,
                                    ii.const_data_ptr<int64_t>(), k.const_data_ptr<int64_t>(), k.size(0), num_images,
                                    max_dets, dets.mutable_data_ptr<float>(), counts.mutable_data_ptr<int32_t>(),
                                    current_stream(boxes)),
               "pack_detections");
  return std::make_tuple(dets, counts);
}

// ---- batched paste_masks_in_image (roi_heads.py:486-500) in one launch
at::Tensor paste_masks(const at::Tensor& masks, const at::Tensor& boxes, int64_t im_h, int64_t im_w, int64_t padding) {
  TORCH_CHECK(masks.is_cuda() && boxes.is_cuda(), "paste_masks: CUDA tensors expected");
  TORCH_CHECK(masks.dim() == 4 && masks.size(1) == 1 && masks.size(2) == masks.size(3),
              "paste_masks: masks must have shape [N, 1, M, M]");
  TORCH_CHECK(boxes.dim() == 2 && boxes.size(1) == 4 && boxes.size(0) == masks.size(0),
              "paste_masks: boxes must have shape [N, 4]");
  TORCH_CHECK(im_h >= 0 && im_w >= 0 && padding >= 0, "paste_masks: negative size");
  c10::DeviceGuard guard(masks.device());
  const int64_t N = masks.size(0), M = masks.size(2);
  at::Tensor out = at::empty({N, 1, im_h, im_w}, masks.options());
  if (out.numel() == 0) return out;
  at::Tensor m = masks.contiguous(), b = boxes.to(at::kFloat).contiguous();
  check_status(tvmi_paste_masks(m.const_data_ptr(), b.const_data_ptr<float>(), out.mutable_data_ptr(),
                                dtype_of(masks, "paste_masks"), N, M, im_h, im_w, padding, current_stream(masks)),
               "paste_masks");
  return out;
}

// ---- candidate generation for the two post-processing stages (postprocess.hip)
std::tuple<at::Tensor, at::Tensor, at::Tensor> detection_candidates(const at::Tensor& class_logits,
                                                                    const at::Tensor& box_regression,
                                                                    const at::Tensor& proposals, const at::Tensor& row_image,
                                                                    const at::Tensor& image_hw, at::ArrayRef<double> weights,
                                                                    double bbox_xform_clip, double score_thresh,
                                                                    double min_size) {
  TORCH_CHECK(class_logits.is_cuda() && box_regression.is_cuda() && proposals.is_cuda() && row_image.is_cuda() &&
                  image_hw.is_cuda(),
              "detection_candidates: CUDA tensors expected");
  TORCH_CHECK(class_logits.dim() == 2, "detection_candidates: class_logits must be [R, C]");
  const int64_t R = class_logits.size(0), C = class_logits.size(1);
  TORCH_CHECK(box_regression.dim() == 2 && box_regression.size(0) == R && box_regression.size(1) == 4 * C,
              "detection_candidates: box_regression must be [R, 4*C]");
  TORCH_CHECK(proposals.dim() == 2 && proposals.size(0) == R && proposals.size(1) == 4 && row_image.numel() == R,
              "detection_candidates: proposals [R,4] and row_image [R] expected");
  TORCH_CHECK(image_hw.dim() == 2 && image_hw.size(1) == 2 && image_hw.size(0) >= 1, "detection_candidates: image_hw must be [B, 2]");
  TORCH_CHECK(weights.size() == 4, "detection_candidates: 4 box coder weights expected");
  c10::DeviceGuard guard(class_logits.device());
  at::Tensor lg = class_logits.to(at::kFloat).contiguous(), rg = box_regression.to(at::kFloat).contiguous();
  at::Tensor pr = proposals.to(at::kFloat).contiguous(), ri = row_image.to(at::kInt).contiguous();
  at::Tensor hw = image_hw.to(at::kFloat).contiguous();
  const int64_t Cm = C > 0 ? C - 1 : 0;
  at::Tensor cb = at::empty({R, Cm, 4}, lg.options()), cs = at::empty({R, Cm}, lg.options());
  at::Tensor cv = at::empty({R, Cm}, lg.options().dtype(at::kByte));
  check_status(tvmi_detection_candidates(lg.const_data_ptr<float>(), rg.const_data_ptr<float>(), pr.const_data_ptr<float>(),
                                         ri.const_data_ptr<int32_t>(), hw.const_data_ptr<float>(), R, C, hw.size(0),
                                         (float)weights[0], (float)weights[1], (float)weights[2], (float)weights[3],
                                         (float)bbox_xform_clip, (float)score_thresh, (float)min_size,
                                         cb.mutable_data_ptr<float>(), cs.mutable_data_ptr<float>(),
                                         cv.mutable_data_ptr<uint8_t>(), current_stream(class_logits)),
               "detection_candidates");
  return std::make_tuple(cb, cs, cv);
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> rpn_candidates(
    const at::Tensor& objectness, const at::Tensor& boxes, const c10::optional<at::Tensor>& deltas, const at::Tensor& top_idx,
    const at::Tensor& level_offsets, const at::Tensor& image_hw, double bbox_xform_clip, double score_thresh,
    double min_size) {
  TORCH_CHECK(objectness.is_cuda() && boxes.is_cuda() && top_idx.is_cuda() && level_offsets.is_cuda() && image_hw.is_cuda(),
              "rpn_candidates: CUDA tensors expected");
  TORCH_CHECK(objectness.dim() == 2 && boxes.dim() == 3 && boxes.size(0) == objectness.size(0) &&
                  boxes.size(1) == objectness.size(1) && boxes.size(2) == 4,
              "rpn_candidates: objectness [B,A] and boxes [B,A,4] expected");
  const int64_t B = objectness.size(0), A = objectness.size(1);
  TORCH_CHECK(top_idx.dim() == 2 && top_idx.size(0) == B, "rpn_candidates: top_idx must be [B, T]");
  TORCH_CHECK(image_hw.dim() == 2 && image_hw.size(0) == B && image_hw.size(1) == 2, "rpn_candidates: image_hw must be [B, 2]");
  TORCH_CHECK(level_offsets.dim() == 1 && level_offsets.numel() >= 1, "rpn_candidates: level_offsets must be [L]");
  c10::DeviceGuard guard(objectness.device());
  const int64_t T = top_idx.size(1);
  at::Tensor ob = objectness.to(at::kFloat).contiguous(), bx = boxes.to(at::kFloat).contiguous();
  at::Tensor ti = top_idx.to(at::kLong).contiguous(), lo = level_offsets.to(at::kLong).contiguous();
  at::Tensor hw = image_hw.to(at::kFloat).contiguous(), dl;
  const float* dptr = nullptr;
  if (deltas.has_value() && deltas->defined()) {
    TORCH_CHECK(deltas->sizes() == boxes.sizes(), "rpn_candidates: deltas must match boxes");
    dl = deltas->to(at::kFloat).contiguous();
    dptr = dl.const_data_ptr<float>();
  }
  at::Tensor ob_ = at::empty({B, T, 4}, ob.options()), os = at::empty({B, T}, ob.options());
  at::Tensor ol = at::empty({B, T}, ob.options().dtype(at::kLong)), ov = at::empty({B, T}, ob.options().dtype(at::kByte));
  check_status(tvmi_rpn_candidates(ob.const_data_ptr<float>(), bx.const_data_ptr<float>(), dptr, ti.const_data_ptr<int64_t>(),
                                   lo.const_data_ptr<int64_t>(), hw.const_data_ptr<float>(), B, A, T, lo.numel(),
                                   (float)bbox_xform_clip, (float)score_thresh, (float)min_size,
                                   ob_.mutable_data_ptr<float>(), os.mutable_data_ptr<float>(), ol.mutable_data_ptr<int64_t>(),
                                   ov.mutable_data_ptr<uint8_t>(), current_stream(objectness)),
               "rpn_candidates");
  return std::make_tuple(ob_, os, ol, ov);
}

std::tuple<at::Tensor, at::Tensor> pack_detections_devcount(const at::Tensor& boxes, const at::Tensor& scores,
                                                            const c10::optional<at::Tensor>& labels,
                                                            const at::Tensor& image_idx, const at::Tensor& keep,
                                                            const at::Tensor& num_keep, int64_t num_images, int64_t max_dets) {
  TORCH_CHECK(boxes.is_cuda() && scores.is_cuda() && image_idx.is_cuda() && keep.is_cuda() && num_keep.is_cuda(),
              "pack_detections: CUDA tensors expected");
  TORCH_CHECK(boxes.dim() == 2 && boxes.size(1) == 4 && scores.dim() == 1 && scores.size(0) == boxes.size(0) &&
                  image_idx.dim() == 1 && image_idx.size(0) == boxes.size(0) && keep.dim() == 1 && num_keep.numel() == 1 &&
                  num_keep.scalar_type() == at::kLong,
              "pack_detections: boxes [N,4], scores [N], image_idx [N], keep [M], num_keep [1] int64 expected");
  c10::DeviceGuard guard(boxes.device());
  at::Tensor b = boxes.to(at::kFloat).contiguous(), s = scores.to(at::kFloat).contiguous();
  at::Tensor ii = image_idx.to(at::kLong).contiguous(), k = keep.to(at::kLong).contiguous();
  at::Tensor lab;
  const int64_t* lab_ptr = nullptr;
  if (labels.has_value() && labels->defined()) {
    lab = labels->to(at::kLong).contiguous();
    lab_ptr = lab.const_data_ptr<int64_t>();
  }
  at::Tensor dets = at::empty({num_images, max_dets, 6}, b.options());
  at::Tensor counts = at::empty({num_images}, b.options().dtype(at::kInt));
  check_status(tvmi_pack_detections_devcount(b.const_data_ptr<float>(), s.const_data_ptr<float>(), lab_ptr,
                                             ii.const_data_ptr<int64_t>(), k.const_data_ptr<int64_t>(), k.size(0),
                                             num_keep.const_data_ptr<int64_t>(), num_images, max_dets,
                                             dets.mutable_data_ptr<float>(), counts.mutable_data_ptr<int32_t>(),
                                             current_stream(boxes)),
               "pack_detections");
  return std::make_tuple(dets, counts);
}

// The collective payload written in place: [B, max_dets * 6 + 1] floats, the count of an image in the last column.
at::Tensor pack_detections_payload(const at::Tensor& boxes, const at::Tensor& scores, const c10::optional<at::Tensor>& labels,
                                   const at::Tensor& image_idx, const at::Tensor& keep, const at::Tensor& num_keep,
                                   int64_t num_images, int64_t max_dets) {
  TORCH_CHECK(boxes.is_cuda() && scores.is_cuda() && image_idx.is_cuda() && keep.is_cuda() && num_keep.is_cuda(),
              "pack_detections: CUDA tensors expected");
  TORCH_CHECK(boxes.dim() == 2 && boxes.size(1) == 4 && scores.dim() == 1 && scores.size(0) == boxes.size(0) &&
                  image_idx.dim() == 1 && image_idx.size(0) == boxes.size(0) && keep.dim() == 1 && num_keep.numel() == 1 &&
                  num_keep.scalar_type() == at::kLong,
              "pack_detections: boxes [N,4], scores [N], image_idx [N], keep [M], num_keep [1] int64 expected");
  c10::DeviceGuard guard(boxes.device());
  at::Tensor b = boxes.to(at::kFloat).contiguous(), s = scores.to(at::kFloat).contiguous();
  at::Tensor ii = image_idx.to(at::kLong).contiguous(), k = keep.to(at::kLong).contiguous();
  at::Tensor lab;
  const int64_t* lab_ptr = nullptr;
  if (labels.has_value() && labels->defined()) {
    lab = labels->to(at::kLong).contiguous();
    lab_ptr = lab.const_data_ptr<int64_t>();
  }
  at::Tensor payload = at::empty({num_images, max_dets * 6 + 1}, b.options());
  check_status(tvmi_pack_detections_payload(b.const_data_ptr<float>(), s.const_data_ptr<float>(), lab_ptr,
                                            ii.const_data_ptr<int64_t>(), k.const_data_ptr<int64_t>(), k.size(0),
                                            num_keep.const_data_ptr<int64_t>(), num_images, max_dets,
                                            payload.mutable_data_ptr<float>(), max_dets * 6 + 1, nullptr, current_stream(boxes)),
               "pack_detections_payload");
  return payload;
}

// The detector step's batched NMS + padded top-k payload as ONE launch (include/tvmi.h: tvmi_nms_step).  Returns
// (keep [n] with num[0] valid entries in the reference's order, num [1] int64 on the device, payload [B, max_dets * 6 + 1]):
// what nms_segmented_padded + pack_detections_payload return, without a host read and without the launches in between.
std::tuple<at::Tensor, at::Tensor, at::Tensor> nms_step(const at::Tensor& dets, const at::Tensor& scores, const at::Tensor& idxs,
                                                        double iou_threshold, int64_t num_segments, const at::Tensor& image_idx,
                                                        const c10::optional<at::Tensor>& labels, int64_t num_images, int64_t max_dets) {
  TORCH_CHECK(dets.is_cuda() && scores.is_cuda() && idxs.is_cuda() && image_idx.is_cuda(), "nms_step: CUDA tensors expected");
  TORCH_CHECK(dets.dim() == 2 && dets.size(1) == 4 && dets.scalar_type() == at::kFloat && scores.scalar_type() == at::kFloat,
              "nms_step: float32 boxes [N,4] and scores [N] expected");
  const int64_t n = dets.size(0);
  TORCH_CHECK(scores.dim() == 1 && scores.size(0) == n && idxs.dim() == 1 && idxs.size(0) == n && image_idx.dim() == 1 &&
                  image_idx.size(0) == n,
              "nms_step: scores, idxs and image_idx must have one entry per box");
  TORCH_CHECK(n >= 1 && n <= 4096 && num_segments >= 1 && num_segments <= 64 && num_images >= 1 && num_images <= 16 && max_dets >= 1,
              "nms_step: 1 <= n <= 4096, num_segments <= 64, num_images <= 16 (larger problems: nms_segmented_padded + pack_detections_payload)");
  c10::DeviceGuard guard(dets.device());
  at::Tensor b = dets.contiguous(), sc = scores.contiguous(), sg = idxs.to(at::kLong).contiguous(), ii = image_idx.to(at::kLong).contiguous();
  at::Tensor lab;
  const int64_t* lab_ptr = nullptr;
  if (labels.has_value() && labels->defined()) {
    lab = labels->to(at::kLong).contiguous();
    lab_ptr = lab.const_data_ptr<int64_t>();
  }
  at::Tensor keep = at::empty({n}, dets.options().dtype(at::kLong));
  at::Tensor num = at::empty({1}, dets.options().dtype(at::kLong));
  at::Tensor payload = at::empty({num_images, max_dets * 6 + 1}, b.options());
  const size_t sb = tvmi_nms_step_workspace_bytes(n, num_segments);
  at::Tensor sws = at::empty({(int64_t)sb}, dets.options().dtype(at::kByte));
  check_status(tvmi_nms_step(b.const_data_ptr<float>(), sc.const_data_ptr<float>(), sg.const_data_ptr<int64_t>(), n, num_segments,
                             iou_threshold, sws.mutable_data_ptr(), sb, keep.mutable_data_ptr<int64_t>(), num.mutable_data_ptr<int64_t>(),
                             ii.const_data_ptr<int64_t>(), lab_ptr, num_images, max_dets, payload.mutable_data_ptr<float>(),
                             max_dets * 6 + 1, nullptr, 1, current_stream(dets)),
               "nms_step");
  return std::make_tuple(keep, num, payload);
}

// The detector step as ONE call (include/tvmi.h: tvmi_multiscale_roi_align_forward_boxes_with_nms_step): multiscale_roi_align_boxes
// and nms_step with their own arguments and their own results — (output, rois, keep, num, payload) — but where the RoIAlign call takes
// its one-launch 7 x 7 route the NMS workgroups ride in front of its grid: one launch on one stream instead of two launches that
// need two streams, a fork and a join to overlap.
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> roi_align_boxes_nms_step(
    at::TensorList features, at::TensorList boxes, at::ArrayRef<double> scales, int64_t pooled_height, int64_t pooled_width,
    int64_t sampling_ratio, bool aligned, int64_t k_min, int64_t k_max, double canonical_scale, double canonical_level, double eps,
    const at::Tensor& dets, const at::Tensor& scores, const at::Tensor& idxs, double iou_threshold, int64_t num_segments,
    const at::Tensor& image_idx, const c10::optional<at::Tensor>& labels, int64_t num_images, int64_t max_dets) {
  TORCH_CHECK(features.size() >= 1 && features.size() <= 8, "multiscale_roi_align: 1..8 feature levels supported");
  TORCH_CHECK(features.size() == scales.size(), "multiscale_roi_align: one scale per feature level");
  TORCH_CHECK(boxes.size() >= 1 && boxes.size() <= 64, "multiscale_roi_align_boxes: 1..64 box lists (one per image)");
  const at::Tensor& f0 = features[0];
  TORCH_CHECK(f0.is_cuda() && f0.dim() == 4, "features must be 4d CUDA tensors");
  TORCH_CHECK(dets.is_cuda() && scores.is_cuda() && idxs.is_cuda() && image_idx.is_cuda() && dets.device() == f0.device(),
              "roi_align_boxes_nms_step: CUDA tensors on one device expected");
  TORCH_CHECK(dets.dim() == 2 && dets.size(1) == 4 && dets.scalar_type() == at::kFloat && scores.scalar_type() == at::kFloat,
              "nms_step: float32 boxes [N,4] and scores [N] expected");
  const int64_t n = dets.size(0);
  TORCH_CHECK(scores.dim() == 1 && scores.size(0) == n && idxs.dim() == 1 && idxs.size(0) == n && image_idx.dim() == 1 &&
                  image_idx.size(0) == n,
              "nms_step: scores, idxs and image_idx must have one entry per box");
  TORCH_CHECK(n >= 1 && n <= 4096 && num_segments >= 1 && num_segments <= 64 && num_images >= 1 && num_images <= 16 && max_dets >= 1,
              "nms_step: 1 <= n <= 4096, num_segments <= 64, num_images <= 16 (larger problems: nms_segmented_padded + pack_detections_payload)");
  c10::DeviceGuard guard(f0.device());
  std::vector<at::Tensor> keepalive, bkeep;
  std::vector<const void*> ptrs, bptrs;
  std::vector<int64_t> hs, ws, counts;
  int64_t K = 0;
  for (const at::Tensor& b : boxes) {
    TORCH_CHECK(b.is_cuda() && b.dim() == 2 && b.size(1) == 4 && b.scalar_type() == at::kFloat && b.device() == f0.device(),
                "multiscale_roi_align_boxes: float32 CUDA boxes [n_i, 4] expected");
    bkeep.push_back(b.contiguous());
    bptrs.push_back(b.size(0) ? bkeep.back().const_data_ptr() : nullptr);
    counts.push_back(b.size(0));
    K += b.size(0);
  }
  for (const at::Tensor& f : features) {
    TORCH_CHECK(f.is_cuda() && f.dim() == 4 && f.size(0) == f0.size(0) && f.size(1) == f0.size(1) &&
                    f.scalar_type() == f0.scalar_type() && f.device() == f0.device(),
                "multiscale_roi_align: feature levels must share device, dtype, batch and channel sizes");
    keepalive.push_back(f.contiguous());
    ptrs.push_back(keepalive.back().const_data_ptr());
    hs.push_back(f.size(2));
    ws.push_back(f.size(3));
  }
  const int64_t C = f0.size(1);
  at::Tensor rois = at::empty({K, 5}, f0.options().dtype(at::kFloat));
  at::Tensor output = at::empty({K, C, pooled_height, pooled_width}, f0.options());
  at::Tensor b = dets.contiguous(), sc = scores.contiguous(), sg = idxs.to(at::kLong).contiguous(), ii = image_idx.to(at::kLong).contiguous();
  at::Tensor lab;
  const int64_t* lab_ptr = nullptr;
  if (labels.has_value() && labels->defined()) {
    lab = labels->to(at::kLong).contiguous();
    lab_ptr = lab.const_data_ptr<int64_t>();
  }
  at::Tensor keep = at::empty({n}, dets.options().dtype(at::kLong));
  at::Tensor num = at::empty({1}, dets.options().dtype(at::kLong));
  at::Tensor payload = at::empty({num_images, max_dets * 6 + 1}, b.options());
  const size_t sb = tvmi_nms_step_workspace_bytes(n, num_segments);
  at::Tensor sws = at::empty({(int64_t)sb}, dets.options().dtype(at::kByte));
  const size_t fwd_ws_bytes = tvmi_roi_align_forward_workspace_bytes(K, pooled_height, pooled_width, sampling_ratio);
  at::Tensor order_ws = at::empty({(int64_t)fwd_ws_bytes}, f0.options().dtype(at::kByte));
  check_status(tvmi_multiscale_roi_align_forward_boxes_with_nms_step(
                   ptrs.data(), hs.data(), ws.data(), scales.data(), (int64_t)features.size(), bptrs.data(), counts.data(),
                   (int64_t)boxes.size(), rois.mutable_data_ptr(), output.mutable_data_ptr(), dtype_of(f0, "multiscale_roi_align"), f0.size(0),
                   C, pooled_height, pooled_width, sampling_ratio, aligned ? 1 : 0, k_min, k_max, canonical_scale, canonical_level, eps,
                   order_ws.mutable_data_ptr(), fwd_ws_bytes, b.const_data_ptr<float>(), sc.const_data_ptr<float>(),
                   sg.const_data_ptr<int64_t>(), n, num_segments, iou_threshold, sws.mutable_data_ptr(), sb, keep.mutable_data_ptr<int64_t>(),
                   num.mutable_data_ptr<int64_t>(), ii.const_data_ptr<int64_t>(), lab_ptr, num_images, max_dets,
                   payload.mutable_data_ptr<float>(), max_dets * 6 + 1, nullptr, 1, current_stream(f0)),
               "roi_align_boxes_nms_step");
  return std::make_tuple(output, rois, keep, num, payload);
}

// ---- qroi_align (quantized/cpu/qroi_align_kernel.cpp:182-231: checks and messages; :22-178: arithmetic)
at::Tensor qroi_align_forward(const at::Tensor& input, const at::Tensor& rois, double input_scale, int64_t input_zero_point,
                              double rois_scale, int64_t rois_zero_point, double spatial_scale, c10::SymInt pooled_height,
                              c10::SymInt pooled_width, int64_t sampling_ratio, bool aligned) {
  TORCH_CHECK(input.is_cuda(), "input must be a CUDA tensor");
  TORCH_CHECK(rois.is_cuda(), "rois must be a CUDA tensor");
  TORCH_CHECK(rois.dim() == 2 && rois.size(1) == 5, "rois must have shape as Tensor[K, 5]");
  TORCH_CHECK(input.dim() == 4 && input.size(0) == 1, "Only one image per batch is allowed in qroi_align.");
  TORCH_CHECK(input.scalar_type() == rois.scalar_type(), "input should have the same type as rois");
  tvmi_int_dtype dt;
  switch (input.scalar_type()) {
    case at::kByte: dt = TVMI_U8; break;
    case at::kChar: dt = TVMI_I8; break;
    case at::kShort: dt = TVMI_I16; break;
    case at::kInt: dt = TVMI_I32; break;
    case at::kLong: dt = TVMI_I64; break;
    default: TORCH_CHECK(false, "qroi_align: integer tensors expected (the reference dispatches AT_INTEGRAL_TYPES)");
  }
  c10::DeviceGuard guard(input.device());
  const int64_t ph = pooled_height.expect_int(), pw = pooled_width.expect_int();
  at::Tensor in = input.contiguous(), r = rois.contiguous();
  at::Tensor out = at::empty({rois.size(0), input.size(1), ph, pw}, input.options());
  if (out.numel() == 0) return out;
  check_status(tvmi_qroi_align_forward(in.const_data_ptr(), r.const_data_ptr(), out.mutable_data_ptr(), dt, input.size(1), input.size(2),
                                       input.size(3), rois.size(0), ph, pw, input_scale, input_zero_point, rois_scale,
                                       rois_zero_point, spatial_scale, sampling_ratio, aligned ? 1 : 0, current_stream(input)),
               "qroi_align");
  return out;
}

// ---- convert_boxes_to_roi_format (ops/_utils.py:18-25) in one launch
at::Tensor boxes_to_rois(at::TensorList boxes) {
  TORCH_CHECK(boxes.size() >= 1 && boxes.size() <= 64, "boxes_to_rois: 1..64 box lists supported");
  const at::Tensor& b0 = boxes[0];
  TORCH_CHECK(b0.is_cuda(), "boxes_to_rois: CUDA tensors expected");
  c10::DeviceGuard guard(b0.device());
  std::vector<at::Tensor> keep;
  std::vector<const void*> ptrs;
  std::vector<int64_t> counts;
  int64_t total = 0;
  for (const at::Tensor& b : boxes) {
    TORCH_CHECK(b.is_cuda() && b.dim() == 2 && b.size(1) == 4 && b.scalar_type() == b0.scalar_type() && b.device() == b0.device(),
                "boxes_to_rois: every entry must be a [n, 4] tensor of the same dtype and device");
    keep.push_back(b.contiguous());
    ptrs.push_back(keep.back().const_data_ptr());
    counts.push_back(b.size(0));
    total += b.size(0);
  }
  at::Tensor rois = at::empty({total, 5}, b0.options());
  if (total == 0) return rois;
  check_status(tvmi_boxes_to_rois(ptrs.data(), counts.data(), (int64_t)boxes.size(), rois.mutable_data_ptr(),
                                  dtype_of(b0, "boxes_to_rois"), current_stream(b0)),
               "boxes_to_rois");
  return rois;
}

// ---- pairwise IoU / GIoU / DIoU / CIoU in one launch (ops/boxes.py:314-391, 409-436, 439-515)
at::Tensor box_iou_pairwise(const at::Tensor& boxes1, const at::Tensor& boxes2, int64_t mode, double eps) {
  TORCH_CHECK(mode >= 0 && mode <= 3, "box_iou_pairwise: mode 0 IoU, 1 generalized, 2 distance, 3 complete");
  TORCH_CHECK(boxes1.is_cuda() && boxes2.is_cuda() && boxes1.dim() == 2 && boxes2.dim() == 2 && boxes1.size(1) == 4 &&
                  boxes2.size(1) == 4,
              "box_iou_pairwise: boxes1 [N,4] and boxes2 [M,4] CUDA tensors expected");
  c10::DeviceGuard guard(boxes1.device());
  // _upcast (ops/_utils.py:72-84) + type promotion of the two inputs
  auto dt = at::promote_types(boxes1.scalar_type(), boxes2.scalar_type());
  // modes 2 / 3 upcast the boxes before any arithmetic (distance_box_iou / complete_box_iou start with _upcast)
  const int src16 = mode >= 2 ? 0 : (dt == at::kHalf ? 1 : (dt == at::kBFloat16 ? 2 : 0));
  if (dt != at::kDouble) dt = at::kFloat;
  at::Tensor a = boxes1.to(dt).contiguous(), b = boxes2.to(dt).contiguous();
  at::Tensor out = at::empty({a.size(0), b.size(0)}, a.options());
  if (out.numel() == 0) return out;
  check_status(tvmi_box_iou_pairwise(a.const_data_ptr(), b.const_data_ptr(), out.mutable_data_ptr(), dtype_of(a, "box_iou"),
                                     a.size(0), b.size(0), (int)mode, src16, eps, current_stream(boxes1)),
               "box_iou_pairwise");
  return out;
}

// ---- GeneralizedRCNNTransform.forward for a batch in one launch (resize.hip)
at::Tensor normalize_resize_batch(at::TensorList images, at::IntArrayRef out_heights, at::IntArrayRef out_widths,
                                  at::ArrayRef<double> mean, at::ArrayRef<double> stdv, int64_t padded_h, int64_t padded_w) {
  TORCH_CHECK(images.size() >= 1 && images.size() <= 64, "normalize_resize_batch: 1..64 images supported");
  TORCH_CHECK(images.size() == out_heights.size() && images.size() == out_widths.size(),
              "normalize_resize_batch: one output size per image");
  const at::Tensor& i0 = images[0];
  TORCH_CHECK(i0.is_cuda() && i0.dim() == 3, "normalize_resize_batch: images must be 3d CUDA tensors [C, H, W]");
  const int64_t C = i0.size(0);
  TORCH_CHECK((int64_t)mean.size() == C && (int64_t)stdv.size() == C && C <= 4, "normalize_resize_batch: one mean / std per channel (<= 4)");
  c10::DeviceGuard guard(i0.device());
  std::vector<at::Tensor> keep;
  std::vector<const void*> ptrs;
  std::vector<int64_t> hs, ws, ohs(out_heights.begin(), out_heights.end()), ows(out_widths.begin(), out_widths.end());
  for (const at::Tensor& im : images) {
    TORCH_CHECK(im.is_cuda() && im.dim() == 3 && im.size(0) == C && im.scalar_type() == i0.scalar_type() && im.device() == i0.device(),
                "normalize_resize_batch: images must share channel count, dtype and device");
    TORCH_CHECK(im.is_floating_point(), "Expected input images to be of floating type (in range [0, 1]), but found type ",
                im.scalar_type(), " instead");
    keep.push_back(im.contiguous());
    ptrs.push_back(keep.back().const_data_ptr());
    hs.push_back(im.size(1));
    ws.push_back(im.size(2));
  }
  float m[4] = {0, 0, 0, 0}, sd[4] = {1, 1, 1, 1};
  for (int64_t c = 0; c < C; ++c) {
    // torch.as_tensor(image_mean, dtype=image.dtype): the constants are rounded to the image dtype first
    m[c] = (float)mean[c];
    sd[c] = (float)stdv[c];
  }
  at::Tensor out = at::empty({(int64_t)images.size(), C, padded_h, padded_w}, i0.options());
  check_status(tvmi_normalize_resize_batch(ptrs.data(), hs.data(), ws.data(), ohs.data(), ows.data(), (int64_t)images.size(), C, m,
                                           sd, out.mutable_data_ptr(), dtype_of(i0, "normalize_resize_batch"), padded_h,
                                           padded_w, current_stream(i0)),
               "normalize_resize_batch");
  return out;
}

at::Tensor sort_scores_desc(const at::Tensor& scores) {
  TORCH_CHECK(scores.is_cuda() && scores.dim() == 1 && scores.scalar_type() == at::kFloat && scores.size(0) < (1ll << 31),
              "sort_scores_desc: a 1d float32 CUDA tensor with fewer than 2^31 elements expected");
  c10::DeviceGuard guard(scores.device());
  at::Tensor sc = scores.contiguous();
  const int64_t n = sc.size(0);
  at::Tensor order = at::empty({n}, sc.options().dtype(at::kLong));
  if (n <= 4096) {
    check_status(tvmi_sort_scores_desc(sc.const_data_ptr<float>(), n, order.mutable_data_ptr<int64_t>(), current_stream(scores)),
                 "sort_scores_desc");
  } else {
    const size_t sb = tvmi_sort_scores_desc_workspace_bytes(n);
    at::Tensor sws = at::empty({(int64_t)sb}, sc.options().dtype(at::kByte));
    check_status(tvmi_sort_scores_desc_large(sc.const_data_ptr<float>(), n, order.mutable_data_ptr<int64_t>(),
                                             sws.mutable_data_ptr(), sb, current_stream(scores)),
                 "sort_scores_desc_large");
  }
  return order;
}

int64_t cuda_version() { return -1; }  // vision.cpp:21-28 without WITH_CUDA; ROCm never checks it
int64_t tvmi_abi_version() {
  // the glue was compiled against this header: refuse to run on a kernels library of another ABI generation
  TORCH_CHECK(tvmi_version() == TVMI_ABI_VERSION, "libtvmi_kernels.so has ABI ", tvmi_version(), ", the glue was built for ", TVMI_ABI_VERSION);
  return tvmi_version();
}
int64_t tvmi_get_option_op(const std::string& name) {
  int64_t v = 0;
  check_status(tvmi_get_option(name.c_str(), &v), "get_option");
  return v;
}

bool tvmi_set_option_op(const std::string& name, int64_t value) {
  check_status(tvmi_set_option(name.c_str(), value), "set_option");
  return true;
}

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
// ---- TVMI_AUTOFUSE=1: opt-in class-level swap of the fused pieces into the reference's python (vision_amd/autofuse.py).
// This library is what torchvision/extension.py:8-33 loads as `_C`, i.e. its static initialisers are the one piece of our code
// that runs when the UNCHANGED reference package is imported.  When the variable is set and the process hosts a CPython
// interpreter, execute vision_amd/autofuse.py (found next to this file's directory; stdlib imports only — nothing of this
// library is called back while dlopen is still running) as the module `vision_amd.autofuse` and call its install().  The
// interpreter's entry points are looked up in the running process, so a C++ host without python is simply left alone.
namespace {
int autofuse_anchor = 0;
struct AutoFuseBootstrap {
  AutoFuseBootstrap() {
    const char* flag = std::getenv("TVMI_AUTOFUSE");
    if (flag == nullptr || flag[0] == '\0' || (flag[0] == '0' && flag[1] == '\0')) return;
    using is_init_t = int (*)();
    using ensure_t = int (*)();
    using release_t = void (*)(int);
    using run_t = int (*)(const char*);
    auto is_init = reinterpret_cast<is_init_t>(dlsym(RTLD_DEFAULT, "Py_IsInitialized"));
    auto ensure = reinterpret_cast<ensure_t>(dlsym(RTLD_DEFAULT, "PyGILState_Ensure"));
    auto release = reinterpret_cast<release_t>(dlsym(RTLD_DEFAULT, "PyGILState_Release"));
    auto run = reinterpret_cast<run_t>(dlsym(RTLD_DEFAULT, "PyRun_SimpleString"));
    if (!is_init || !ensure || !release || !run || !is_init()) return;
    Dl_info info;
    if (dladdr(&autofuse_anchor, &info) == 0 || info.dli_fname == nullptr) return;
    char real[PATH_MAX];
    if (realpath(info.dli_fname, real) == nullptr) return;      // the overlay's `_C.so` is a symlink to vision_amd/_lib/tvmi_torch.so
    std::string dir(real);
    for (int up = 0; up < 2; ++up) {                             // .../vision_amd/_lib/tvmi_torch.so -> .../vision_amd
      const size_t cut = dir.rfind('/');
      if (cut == std::string::npos) return;
      dir.resize(cut);
    }
    const std::string code =
        "import sys as _s, importlib.util as _u\n"
        "if 'vision_amd.autofuse' not in _s.modules:\n"
        "    _sp = _u.spec_from_file_location('vision_amd.autofuse', r\"\"\"" + dir + "/autofuse.py\"\"\")\n"
        "    _m = _u.module_from_spec(_sp); _s.modules['vision_amd.autofuse'] = _m; _sp.loader.exec_module(_m)\n"
        "_s.modules['vision_amd.autofuse'].install()\n";
    const int gil = ensure();
    if (run(code.c_str()) != 0) std::fprintf(stderr, "[tvmi] TVMI_AUTOFUSE=1: installing vision_amd.autofuse failed (see the traceback above)\n");
    release(gil);
  }
} autofuse_bootstrap;
}  // namespace

TORCH_LIBRARY(tvmi, m) {
  m.def("abi_version", &tvmi_abi_version);
  m.def("get_option", &tvmi_get_option_op);
  m.def("set_option", &tvmi_set_option_op);   // process-wide kernel switches (include/tvmi.h: tvmi_set_option)
  // opt-in: our resize kernels on the CUDA key of aten::upsample_* (returns the previous state)
  m.def("override_aten_upsample", &upsample_override::set);
  m.def("aten_upsample_calls", &upsample_override::calls);
  // batched NMS without the per-class python loop of torchvision/ops/boxes.py:113-126; num_segments > 0 promises
  // ids in [0, num_segments) and unlocks the single-launch path for n <= 4096
  m.def("nms_segmented(Tensor dets, Tensor scores, Tensor? idxs, float iou_threshold, int num_segments=-1) -> Tensor");
  // the same without the host sync on the result size: (keep [n] with a valid prefix, num [1] on the device)
  m.def("nms_segmented_padded(Tensor dets, Tensor scores, Tensor? idxs, float iou_threshold, int num_segments=-1) -> (Tensor, Tensor)");
  m.def("nms_segmented_masked(Tensor dets, Tensor scores, Tensor idxs, Tensor valid, float iou_threshold, int num_segments=-1, int max_segment_size=-1) -> (Tensor, Tensor)");
  m.def(
      "pack_detections_devcount(Tensor boxes, Tensor scores, Tensor? labels, Tensor image_idx, Tensor keep, Tensor num_keep, int num_images, int max_dets) -> (Tensor, Tensor)");
  // the same launch writing the all-gather payload [B, max_dets * 6 + 1] in place (the count in the last column)
  m.def(
      "pack_detections_payload(Tensor boxes, Tensor scores, Tensor? labels, Tensor image_idx, Tensor keep, Tensor num_keep, int num_images, int max_dets) -> Tensor");
  // aten::upsample_* arithmetic on our kernels (mode 0 nearest, 1 nearest-exact, 2 bilinear, 3 bicubic;
  // scale_* <= 0 means "not given")
  m.def(
      "pack_detections(Tensor boxes, Tensor scores, Tensor? labels, Tensor image_idx, Tensor keep, int num_images, int max_dets) -> (Tensor, Tensor)");
  // batched NMS of a detector step + the padded top-k payload in ONE launch (n <= 4096, <= 64 segments, <= 16 images)
  m.def(
      "nms_step(Tensor dets, Tensor scores, Tensor idxs, float iou_threshold, int num_segments, Tensor image_idx, Tensor? labels, int num_images, int max_dets) -> (Tensor, Tensor, Tensor)");
  m.def(
      "multiscale_roi_align(Tensor[] features, Tensor rois, float[] scales, int pooled_height, int pooled_width, int sampling_ratio, bool aligned, int k_min, int k_max, float canonical_scale, float canonical_level, float eps) -> Tensor");
  m.def(
      "multiscale_roi_align_boxes(Tensor[] features, Tensor[] boxes, float[] scales, int pooled_height, int pooled_width, int sampling_ratio, bool aligned, int k_min, int k_max, float canonical_scale, float canonical_level, float eps) -> (Tensor, Tensor)");
  m.def(
      "roi_align_boxes_nms_step(Tensor[] features, Tensor[] boxes, float[] scales, int pooled_height, int pooled_width, int sampling_ratio, bool aligned, int k_min, int k_max, float canonical_scale, float canonical_level, float eps, Tensor dets, Tensor scores, Tensor idxs, float iou_threshold, int num_segments, Tensor image_idx, Tensor? labels, int num_images, int max_dets) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def(
      "multiscale_roi_align_backward(Tensor grad, Tensor rois, int[] heights, int[] widths, float[] scales, int batch_size, int pooled_height, int pooled_width, int sampling_ratio, bool aligned, int k_min, int k_max, float canonical_scale, float canonical_level, float eps) -> Tensor[]");
  // roi_heads.py:680-722 / rpn.py:266-286 up to the NMS, batched over images (one launch each)
  m.def(
      "detection_candidates(Tensor class_logits, Tensor box_regression, Tensor proposals, Tensor row_image, Tensor image_hw, float[] weights, float bbox_xform_clip, float score_thresh, float min_size) -> (Tensor, Tensor, Tensor)");
  m.def(
      "rpn_candidates(Tensor objectness, Tensor boxes, Tensor? deltas, Tensor top_idx, Tensor level_offsets, Tensor image_hw, float bbox_xform_clip, float score_thresh, float min_size) -> (Tensor, Tensor, Tensor, Tensor)");
  // models/detection/transform.py:119-255 (normalize + resize + zero-padded batching) as one launch
  m.def(
      "normalize_resize_batch(Tensor[] images, int[] out_heights, int[] out_widths, float[] mean, float[] std, int padded_h, int padded_w) -> Tensor");
  // indices of aten::sort(scores, stable=True, descending=True) for float32 scores (one launch up to 4096, radix sort above)
  m.def("sort_scores_desc(Tensor scores) -> Tensor");
  // ops/boxes.py:314-391 / 409-436 (box_iou / generalized_box_iou of xyxy boxes) as one launch
  m.def("box_iou_pairwise(Tensor boxes1, Tensor boxes2, int mode, float eps=1e-07) -> Tensor");
  // ops/_utils.py:18-25 (cat + full_like per image + 2 cats) as one launch
  m.def("boxes_to_rois(Tensor[] boxes) -> Tensor");
  // python loop of roi_heads.py:486-500 (pad + expand + resize + paste per detection) as one launch
  m.def("paste_masks(Tensor masks, Tensor boxes, int im_h, int im_w, int padding) -> Tensor");
  m.def(
      "interpolate2d(Tensor input, int out_h, int out_w, int mode, bool align_corners, bool antialias, float scale_h, float scale_w) -> Tensor");
  // its gradient, gathered (deterministic): grad_output [N,C,OH,OW] -> grad_input [N,C,in_h,in_w]
  m.def(
      "interpolate2d_backward(Tensor grad_output, int in_h, int in_w, int mode, bool align_corners, bool antialias, float scale_h, float scale_w) -> Tensor");
}

TORCH_LIBRARY_IMPL(torchvision, CUDA, m) {
  m.impl("roi_align", &roi_align_forward);
  m.impl("_roi_align_backward", &roi_align_backward);
  m.impl("roi_pool", &roi_pool_forward);
  m.impl("_roi_pool_backward", &roi_pool_backward);
  m.impl("ps_roi_align", &ps_roi_align_forward);
  m.impl("_ps_roi_align_backward", &ps_roi_align_backward);
  m.impl("ps_roi_pool", &ps_roi_pool_forward);
  m.impl("_ps_roi_pool_backward", &ps_roi_pool_backward);
  m.impl("deform_conv2d", &deform_conv2d_forward);
  m.impl("_deform_conv2d_backward", &deform_conv2d_backward);
  m.impl("qnms", &qnms_forward);
  m.impl("qroi_align", &qroi_align_forward);
}

TORCH_LIBRARY_IMPL(tvmi, CUDA, m) {
  m.impl("nms_segmented", &nms_segmented);
  m.impl("nms_segmented_padded", &nms_segmented_padded);
  m.impl("nms_segmented_masked", &nms_segmented_masked);
  m.impl("nms_step", &nms_step);
  m.impl("pack_detections_devcount", &pack_detections_devcount);
  m.impl("pack_detections_payload", &pack_detections_payload);
  m.impl("interpolate2d", &interpolate2d);
  m.impl("interpolate2d_backward", &interpolate2d_backward);
  m.impl("multiscale_roi_align", &multiscale_roi_align);
  m.impl("multiscale_roi_align_boxes", &multiscale_roi_align_boxes);
  m.impl("roi_align_boxes_nms_step", &roi_align_boxes_nms_step);
  m.impl("multiscale_roi_align_backward", &multiscale_roi_align_backward);
  m.impl("pack_detections", &pack_detections);
  m.impl("paste_masks", &paste_masks);
  m.impl("boxes_to_rois", &boxes_to_rois);
  m.impl("sort_scores_desc", &sort_scores_desc);
  m.impl("normalize_resize_batch", &normalize_resize_batch);
  m.impl("box_iou_pairwise", &box_iou_pairwise);
  m.impl("detection_candidates", &detection_candidates);
  m.impl("rpn_candidates", &rpn_candidates);
}

}  // namespace tvmi_shim
