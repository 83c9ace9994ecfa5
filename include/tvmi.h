/*
 * tvmi.h — C ABI of libtvmi_kernels.so: hand-written gfx950 (MI355X / CDNA4) kernels
 * for the torchvision custom-operator hot path.
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes plain
 * device pointers, sizes and a `hipStream_t` (passed as void*), enqueues work on that
 * stream, never synchronises the host (except where stated) and returns a hipError_t
 * value as int (0 == hipSuccess).  No torch types appear here; the dispatcher glue
 * that binds these launchers to the `torchvision::` schemas lives in
 * vision_amd/csrc/torch_shim.cpp.
 *
 * Each function cites the reference interface it replaces (paths relative to the
 * pytorch/vision tree).  "rois" are always [K,5] = (batch_index, x1, y1, x2, y2) in the
 * same dtype as the feature tensor, feature tensors are contiguous NCHW.
 */
#ifndef TVMI_H_
#define TVMI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Element types of feature / box tensors. */
typedef enum tvmi_dtype {
  TVMI_F32 = 0,
  TVMI_F64 = 1,
  TVMI_F16 = 2,
  TVMI_BF16 = 3
} tvmi_dtype;

/* Library / ABI version (major*10000 + minor*100 + patch). */
int tvmi_version(void);
/* Static string of the gfx arch the kernels were compiled for ("gfx950"). */
const char* tvmi_arch(void);
/* Human-readable text for the last non-zero status returned on this thread. */
const char* tvmi_last_error(void);

/* ------------------------------------------------------------------ NMS ----------
 * Replaces: torchvision/csrc/ops/cuda/nms_kernel.cu:56-148 (nms_kernel_impl,
 * gather_keep_from_mask) and the host sequence at :166-258; semantics (and the
 * bit-exact arithmetic) are those of torchvision/csrc/ops/cpu/nms_kernel.cpp:17-95.
 *
 *   dets   [n,4] xyxy, dtype dt (F32 or F64), contiguous, UNSORTED
 *   order  [n]   int64: indices of dets in processing order (stable descending score)
 *   seg    [n]   optional int64 segment (class / level / image) id per ORIGINAL box, or
 *                NULL.  Boxes in different segments never suppress each other (this is
 *                what torchvision.ops.boxes._batched_nms_vanilla computes, ops/boxes.py
 *                :113-126, without its per-class python loop).
 *   keep_out [n] int64: original indices of kept boxes, in `order` order
 *   num_keep_out [1] int64 (device): number of kept boxes
 *   workspace: device scratch of at least tvmi_nms_workspace_bytes(n) bytes.
 */
size_t tvmi_nms_workspace_bytes(int64_t n);
int tvmi_nms(const void* dets, const int64_t* order, const int64_t* seg, int64_t n,
             double iou_threshold, tvmi_dtype dt, void* workspace, size_t workspace_bytes,
             int64_t* keep_out, int64_t* num_keep_out, void* stream);

/* ------------------------------------------------------------- RoIAlign ----------
 * Replaces: torchvision/csrc/ops/cuda/roi_align_kernel.cu:68-143,334-394 (forward),
 * :204-332,396-466 (backward); arithmetic follows
 * torchvision/csrc/ops/cpu/roi_align_kernel.cpp:18-115,183-289 and
 * cpu/roi_align_common.h:32-124.
 *   input  [N,C,H,W]  output [K,C,PH,PW] (fully overwritten, no pre-zero needed)
 * backward: grad [K,C,PH,PW] read with the given element strides; grad_input
 * [N,C,H,W] must be zero-filled by the caller (the launcher accumulates atomically).
 * F16/BF16 accumulate in fp32.
 */
int tvmi_roi_align_forward(const void* input, const void* rois, void* output, tvmi_dtype dt,
                           int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                           int64_t pooled_h, int64_t pooled_w, double spatial_scale,
                           int64_t sampling_ratio, int aligned, void* stream);
int tvmi_roi_align_backward(const void* grad, const void* rois, void* grad_input, tvmi_dtype dt,
                            int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                            int64_t pooled_h, int64_t pooled_w, double spatial_scale,
                            int64_t sampling_ratio, int aligned, int64_t n_stride,
                            int64_t c_stride, int64_t h_stride, int64_t w_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TVMI_H_ */
