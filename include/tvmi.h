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
 * same dtype as the feature tensor (float32 for the multi-scale entries), feature tensors are contiguous NCHW.
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

/* integer element types of the quantized entries (the reference dispatches AT_INTEGRAL_TYPES, quantized/cpu/qroi_align_kernel.cpp:252) */
typedef enum {
  TVMI_U8 = 0,
  TVMI_I8 = 1,
  TVMI_I16 = 2,
  TVMI_I32 = 3,
  TVMI_I64 = 4
} tvmi_int_dtype;

/* Library / ABI version (major*10000 + minor*100 + patch).  300 = round 3: the RoI backward entries OVERWRITE grad_input
 * in the owner regimes, tvmi_roi_align_backward_workspace_bytes takes (N, K, PH, PW), tvmi_box_iou_pairwise has `eps`,
 * the RoIAlign forward workspace grew (tvmi_roi_align_forward_workspace_bytes).  A caller built against a 100-series
 * header must not call this library: check TVMI_ABI_VERSION == tvmi_version(). */
#define TVMI_ABI_VERSION 307
int tvmi_version(void);
/* Process-wide tuning switches (thread-safe to read concurrently with launches; set them before use).  Returns 0, or an
 * error for an unknown name.
 *   "roi_align.pin_chunks"       1 (default) / 0: the fp32 / 16-bit LDS-DMA forward kernels pin channel chunks to XCDs (every
 *                                byte of a feature map is then wanted by one private L2 only) when the chunk count is a
 *                                multiple of 8; 0 = every XCD walks a contiguous RoI range chunk by chunk
 *   "roi_align.order"            1 (default) / 0: with pinned chunks, RoIs start in (image, level, window-top band) order
 *                                (a one-workgroup counting sort in front of the launch; needs the forward workspace)
 *   "roi_align.order_bands"      bands per (image, level) in that order key (default 16, 1..64)
 *   "dcn.channels_last_gather"   1 (default) / 0: the 16-bit MFMA deform_conv2d kernel samples a [B, H*W, C] copy of the input
 *   "dcn.bwd_mfma"               1 (default) / 0: tvmi_deform_conv2d_backward contracts on the matrix cores where the shapes allow
 *                                (0 = the direct kernels for every problem)
 *   "dcn.bwd_window"             1 (default) / 0: its data-gradient kernel adds the grad_input contributions of a pixel tile in an
 *                                LDS window and flushes it once per channel chunk (0 = one global atomic per contribution)
 *   "dcn.bwd_owner"              1 (default) / 0: 3 x 3 problems with whole 64-channel chunks take the owner form of that kernel
 *                                (lane = channel, plain LDS read-add-write instead of LDS atomics, nine taps' accumulators resident)
 *   "nms.replan_min_boxes"       tvmi_nms_blocking re-plans problems of at least this many boxes on their survivors
 *                                (default 24576; 0 = never)
 *   "nms.replan_divisor"         share of the row chunks swept before a re-plan (default 16 = the first sixteenth)
 *   "nms.replan_max"             re-plans per call (default 3)
 *   "nms.device_handoff"         1 (default) / 0: resolver <-> push hand-offs of the large path of tvmi_nms_blocking through memory
 *                                words (agent-scope atomics, no fences) instead of stream events; one call per device at a time,
 *                                never under capture, never in tvmi_nms (which may not synchronise).  Needs the call's three
 *                                streams to run concurrently: not taken when the environment says kernels are serialised
 *                                (ROCPROF_COUNTER_COLLECTION / ROCPROF_COUNTERS / ROCP_METRICS / ROCP_INPUT, HIP_LAUNCH_BLOCKING,
 *                                AMD_SERIALIZE_KERNEL).  Anything else that keeps the launches from running side by side is caught
 *                                on the device: a poll gives up after about a second, the call is re-run with stream events —
 *                                a slower correct answer — and the process stops using the hand-offs (reads back as 0) until the
 *                                option is set to 1 again
 *   "nms.handoff_lose_flag"      test hook, 0 (default) / 1: the resolver of the first chunk does not announce itself, so the
 *                                recovery path above runs for real
 *   "nms.mask_lds_bytes"         dynamic LDS per mask workgroup of the large path (all chunks but the first) — an
 *                                occupancy cap that keeps wave slots free for the sweep's 16-wave workgroup
 *                                (default 36000 = four workgroups per CU; 0 = no cap)
 *   "nms.step_fused"             1 (default) / 0: detector-step sizes (n <= 4096, <= 64 segments; plain nms up to 1024 boxes) take the
 *                                one-launch kernel of tvmi_nms_step inside the torch glue; 0 = the launch chain (score sort, collect,
 *                                tiles, sweep) — kept as the route for more segments / float64 and as the A/B reference
 *   "roi_align.inline_mop"       1 (default) / 0: in the 7 x 7 multi-scale forward the units the LDS-DMA path declines take the wave path
 *                                inside the same launch instead of a worklist + a mop-up launch
 *   "roi_align.carry_step"       1 (default) / 0: tvmi_multiscale_roi_align_forward_boxes_with_nms_step puts the NMS workgroups of the
 *                                detector step in front of the RoIAlign grid (one launch); 0 = the two entries one after the other
 *   "roi_align.fold_order"       1 (default) / 0: in the 7 x 7 multi-scale forward that starts from box lists (<= 4096 boxes, stream not
 *                                being captured) the launch-order pre-pass runs as ONE WORKGROUP OF THE LAUNCH instead of a launch
 *                                in front of it; the first round of units runs in input order meanwhile.  Same results either way
 *   "roi_align.fold_first_round_pct"  100 (default), 0..400: how many positions stay in input order, in percent of the workgroups the
 *                                chip holds at once (tuning value of the folded form) */
int tvmi_set_option(const char* name, int64_t value);
/* Current value of a switch of tvmi_set_option (0, or an error for an unknown name). */
int tvmi_get_option(const char* name, int64_t* value);
/* Static string of the gfx arch the kernels were compiled for ("gfx950"). */
const char* tvmi_arch(void);
/* Human-readable text for the last non-zero status returned on this thread. */
const char* tvmi_last_error(void);

/* `waiter` (hipStream_t) waits for everything enqueued on `signaler` so far: torch's Stream.wait_stream with an event that
 * releases to DEVICE scope (hipEventReleaseToDevice) instead of system scope — the fork / join of the step's two streams.
 * tvmi_stream_event_scope: 0 system scope (torch's events), 1 device scope (default), 2 no fence from the event itself. */
int tvmi_stream_wait_stream(void* waiter, void* signaler);
int tvmi_stream_event_scope(int scope);

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
/* Same operation, same arguments, same result, for callers that read the result on the host anyway (the reference's
 * nms ends in a device-to-host copy, cuda/nms_kernel.cu:226-236): above "nms.replan_min_boxes" boxes the call sweeps
 * the first chunk(s) of the sorted list, drops every box they removed and starts over on the survivors.  That saves the
 * removed fraction in BOTH dimensions of the pair tests and shortens the serial sweep, and costs one
 * host synchronisation per re-plan (the survivor count sizes the next grids).  100k boxes: 2.2 -> 1.27 ms (38 % kept),
 * 1.9 -> 0.69 ms (9 % kept).  Never re-plans under stream capture; tvmi_nms never synchronises. */
int tvmi_nms_blocking(const void* dets, const int64_t* order, const int64_t* seg, int64_t n,
                      double iou_threshold, tvmi_dtype dt, void* workspace, size_t workspace_bytes,
                      int64_t* keep_out, int64_t* num_keep_out, void* stream);

/* The `order` input of the entries above for small inputs: indices of
 * aten::sort(scores, stable=True, descending=True) (NaN first, ties by ascending index, -0 == +0) for
 * float32 scores, n <= 4096, in one single-workgroup launch (the reference sorts with
 * `scores.sort(0, descending=True)`, cuda/nms_kernel.cu:186-187).
 */
int tvmi_sort_scores_desc(const float* scores, int64_t n, int64_t* order, void* stream);
/* The same order for any n < 2^31 (score_sort.hip): a key pass that encodes the NaN / signed-zero rules, rocPRIM's radix
 * sort of (key, index) pairs, an index-widening pass.  workspace: 256-byte aligned device scratch of at least
 * tvmi_sort_scores_desc_workspace_bytes(n) bytes (the query asks rocPRIM for its temporary storage on the CURRENT device:
 * it returns 0 for n <= 0, n >= 2^31 and when no device is available). */
size_t tvmi_sort_scores_desc_workspace_bytes(int64_t n);
int tvmi_sort_scores_desc_large(const float* scores, int64_t n, int64_t* order, void* workspace,
                                size_t workspace_bytes, void* stream);

/* Segment-major form of the same operation for batched_nms (ops/boxes.py:57-126): the caller
 * additionally provides the STABLE partition of the score order by segment id —
 *   perm     [n] int64: perm[p] = rank in `order` of the box at segment-major position p
 *   seg_keys [n] int64: segment id at position p, ascending (i.e. the values / indices of a
 *                       stable ascending sort of seg[order[.]])
 * Only tiles whose two 64-box blocks share a segment are evaluated and every segment is swept
 * by its own workgroup; keep_out is still in global score order, identical to tvmi_nms with
 * `seg`.  A segment may span at most 128 64-box blocks (8,192 boxes; beyond that one workgroup
 * per segment is the wrong shape); otherwise num_keep_out is set to -1 and keep_out is
 * unspecified — callers fall back to tvmi_nms, whose sweep is parallel over column blocks.
 */
size_t tvmi_nms_segmented_workspace_bytes(int64_t n);
int tvmi_nms_segmented(const void* dets, const int64_t* order, const int64_t* seg_keys, const int64_t* perm, int64_t n,
                       double iou_threshold, tvmi_dtype dt, void* workspace, size_t workspace_bytes, int64_t* keep_out,
                       int64_t* num_keep_out, void* stream);

/* Detector-step sizes in two launches and without the second sort: n <= 4096 boxes, segment ids
 * in [0, num_segments), num_segments <= 1024, every segment <= 1024 boxes (per-image / per-level
 * / per-class lists of one batch).  Same inputs and result as tvmi_nms with `seg`; per-segment
 * tiles instead of an N x N mask, one sweep chain per segment, all segments concurrently.
 * Inputs outside those limits give num_keep_out = -1 (fall back to tvmi_nms).
 */
size_t tvmi_nms_small_segments_workspace_bytes(int64_t n, int64_t num_segments);
int tvmi_nms_small_segments(const void* dets, const int64_t* order, const int64_t* seg, int64_t n,
                            int64_t num_segments, double iou_threshold, tvmi_dtype dt, void* workspace,
                            size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out, void* stream);

/* Device-count forms of the two batched paths (round 5: sync-free detector post-processing, SURVEY.md §8f-1 — replaces
 * the `torch.where` / boolean-index compaction before every batched_nms of models/detection/roi_heads.py:716-729,
 * rpn.py:268-283, retinanet.py:537-565, each of which reads a count on the host).  The lists are laid out for `capacity`
 * candidates; the ones that take part are the first *n_dev entries of `order` (and, segment-major form, of `seg_keys` /
 * `perm`): tvmi_nms_mask_inputs gives masked-out candidates the score -inf and the key INT64_MAX, so the caller's two
 * stable sorts put them behind every live candidate, and counts the live ones on the device (n_live [1] int64, zeroed
 * by the call).  keep_out / num_keep_out as in the host-count forms; launch grids are sized for the capacity and
 * surplus workgroups retire at once.  No entry synchronises. */
int tvmi_nms_mask_inputs(const float* scores, const int64_t* seg, const uint8_t* valid, int64_t n, float* scores_out,
                         int64_t* seg_out, int64_t* n_live, void* stream);
int tvmi_nms_segmented_devcount(const void* dets, const int64_t* order, const int64_t* seg_keys, const int64_t* perm,
                                int64_t capacity, const int64_t* n_dev, const int* partition_flag, double iou_threshold,
                                tvmi_dtype dt, void* workspace, size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out,
                                void* stream);
/* The stable partition of the score order by segment id that the segment-major forms take (`seg_keys`, `perm`): a stable LSD
 * radix sort over the id bits of the sequence that is already in score order — the permutation of the reference's
 * formulation (ops/boxes.py:113-126: per-class index lists in score order).  Ids must lie in [0, num_segments) when
 * num_segments > 0 (the sort then covers ceil(log2) + 1 bits), in [0, 2^31) otherwise; an id outside raises *flag_out (device
 * int, zeroed by the call), which tvmi_nms_segmented_devcount turns into num_keep_out = -1.  With n_dev, ranks >= *n_dev are
 * dead (masked-out candidates) and are placed behind every live one.  n_dev and partition_flag of
 * tvmi_nms_segmented_devcount may be NULL (all boxes live / ids known to be in range). */
size_t tvmi_partition_by_segment_workspace_bytes(int64_t n);
int tvmi_partition_by_segment(const int64_t* order, const int64_t* seg, int64_t n, const int64_t* n_dev, int64_t num_segments,
                              int64_t* keys_out, int64_t* perm_out, int* flag_out, void* workspace, size_t workspace_bytes,
                              void* stream);
int tvmi_nms_small_segments_devcount(const void* dets, const int64_t* order, const int64_t* seg, int64_t capacity,
                                     const int64_t* n_dev, int64_t num_segments, double iou_threshold, tvmi_dtype dt,
                                     void* workspace, size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out,
                                     void* stream);

/* The detector step's batched NMS (+ padded top-k payload) in ONE launch (round 6).  Replaces, on the GPU, the python chain
 * torchvision/ops/boxes.py:57-126 (`batched_nms`: coordinate trick or per-class loop -> torchvision::nms -> index) followed
 * by the per-image split / top-k of models/detection/roi_heads.py:716-737 — and this library's own 4-5 launch chain for the
 * same sizes (score sort, collect, tiles, sweep, pack).
 *   dets [n,4] / scores [n] float32, seg [n] int64 in [0, num_segments) (NULL with num_segments = 1: plain nms),
 *   1 <= n <= 4096, num_segments <= 64, every segment <= 1024 boxes (else *num_keep_out = -1, keep_out unspecified).
 *   keep_out [n] int64: indices of the kept boxes in descending score order over all segments (ties by index), the
 *   reference's result; num_keep_out [1] int64 on the device.  No host synchronisation; one memset + one kernel.
 *   payload != NULL: additionally row b of payload (row_stride floats apart) = the first max_dets kept boxes of image b as
 *   (x1, y1, x2, y2, score, label) zero-padded, + the count behind them when count_in_row; counts [num_images] int32
 *   optional; image_idx [n] int64 gives the image of every box (may be `seg` itself), labels optional; num_images <= 16.
 *   The rows are what tvmi_pack_detections_payload writes from (keep_out, num_keep_out). */
size_t tvmi_nms_step_workspace_bytes(int64_t n, int64_t num_segments);
int tvmi_nms_step(const float* dets, const float* scores, const int64_t* seg, int64_t n, int64_t num_segments,
                  double iou_threshold, void* workspace, size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out,
                  const int64_t* image_idx, const int64_t* labels, int64_t num_images, int64_t max_dets, float* payload,
                  int64_t row_stride, int32_t* counts, int count_in_row, void* stream);

/* ------------------------------------------------------------- RoIAlign ----------
 * Replaces: torchvision/csrc/ops/cuda/roi_align_kernel.cu:68-143,334-394 (forward),
 * :204-332,396-466 (backward); arithmetic follows
 * torchvision/csrc/ops/cpu/roi_align_kernel.cpp:18-115,183-289 and
 * cpu/roi_align_common.h:32-124.
 *   input  [N,C,H,W]  output [K,C,PH,PW] (fully overwritten, no pre-zero needed)
 *   workspace (optional, may be NULL): tvmi_roi_align_forward_workspace_bytes(K, PH, PW, sampling_ratio) bytes of
 *   device scratch: the work list of the RoIs the LDS-DMA kernel declines (windows that do not fit its LDS blocks; a small
 *   fixed-grid mop-up launch serves them) and the launch order of the counting-sort pre-pass (RoIs of one image / level /
 *   band next to each other).  Results do not depend on it, only which kernels run.
 * backward: grad [K,C,PH,PW] read with the given element strides.  Two regimes:
 *   - TILE-OWNER path (deterministic; what torchvision/ops/roi_align.py:276-281 reroutes to python for):
 *     float32, 7x7 or 14x14 bins (any sampling_ratio), the [C,PH,PW] block of a RoI contiguous (w_stride 1,
 *     h_stride PW, c_stride PH*PW), H, W <= 4096, and a workspace of tvmi_roi_align_backward_workspace_bytes(N, K, PH, PW) bytes.
 *     Every 16x16 tile of grad_input is accumulated in registers by ONE workgroup and written exactly once
 *     with plain stores: grad_input is FULLY OVERWRITTEN (no zero-fill needed), no atomics, bit-reproducible.
 *     tvmi_roi_align_backward_overwrites(...) tells the caller whether a call takes this path.
 *   - otherwise (fp64, 16-bit, other bin shapes, no workspace): accumulates with hardware atomics into a
 *     grad_input the CALLER zero-filled, like cuda/roi_align_kernel.cu:304-327,440.
 */
size_t tvmi_roi_align_forward_workspace_bytes(int64_t K, int64_t pooled_h, int64_t pooled_w, int64_t sampling_ratio);
int tvmi_roi_align_forward(const void* input, const void* rois, void* output, tvmi_dtype dt,
                           int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                           int64_t pooled_h, int64_t pooled_w, double spatial_scale,
                           int64_t sampling_ratio, int aligned, void* workspace, size_t workspace_bytes,
                           void* stream);
/* 0 when the pooled shape has no tile-owner kernel. */
size_t tvmi_roi_align_backward_workspace_bytes(int64_t N, int64_t K, int64_t pooled_h, int64_t pooled_w);
/* 1 if tvmi_roi_align_backward with these arguments overwrites grad_input (tile-owner path), 0 if it accumulates. */
int tvmi_roi_align_backward_overwrites(tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                                       int64_t pooled_h, int64_t pooled_w, int64_t c_stride, int64_t h_stride,
                                       int64_t w_stride, size_t workspace_bytes);
int tvmi_roi_align_backward(const void* grad, const void* rois, void* grad_input, tvmi_dtype dt,
                            int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                            int64_t pooled_h, int64_t pooled_w, double spatial_scale,
                            int64_t sampling_ratio, int aligned, int64_t n_stride,
                            int64_t c_stride, int64_t h_stride, int64_t w_stride, void* workspace,
                            size_t workspace_bytes, void* stream);

/* Multi-scale RoIAlign (FPN): replaces the per-level python loop of
 * torchvision/ops/poolers.py:147-227 (_multiscale_roi_align: LevelMapper -> torch.where ->
 * roi_align -> index_put per level) with ONE launch.  `inputs[l]` is level l's [N,C,H_l,W_l]
 * feature map (same N, C, dtype), `rois` [K,5] are FLOAT32 image coordinates whatever the feature dtype
 * (levels and sample positions are computed from the fp32 boxes, as the reference does); the level of a RoI
 * is floor(canonical_level + log2(sqrt(area)/canonical_scale) + eps) clamped to
 * [k_min, k_max], minus k_min (poolers.py:73-84).  output [K,C,PH,PW], fully overwritten.
 * F32 / F16 / BF16.
 */
int tvmi_multiscale_roi_align_forward(const void* const* inputs, const int64_t* heights,
                                      const int64_t* widths, const double* spatial_scales,
                                      int64_t n_levels, const void* rois, void* output, tvmi_dtype dt,
                                      int64_t N, int64_t C, int64_t K, int64_t pooled_h, int64_t pooled_w,
                                      int64_t sampling_ratio, int aligned, int64_t k_min, int64_t k_max,
                                      double canonical_scale, double canonical_level, double eps,
                                      void* workspace, size_t workspace_bytes, void* stream);
/* Backward of the multi-scale form in one launch (float32 grads and RoIs): grad [K,C,PH,PW] (element strides
 * given) goes into the gradient map of each RoI's level, grad_inputs[l] = [N,C,H_l,W_l] contiguous.  Same two
 * regimes as tvmi_roi_align_backward: with the tile-owner workspace every map is fully overwritten and the
 * result is deterministic; otherwise the maps must be zero-filled by the caller.  Replaces the per-level
 * autograd loop the reference runs through poolers.py:199-222 + _roi_align_backward.
 */
int tvmi_multiscale_roi_align_backward_overwrites(tvmi_dtype dt, int64_t N, int64_t C, int64_t K, const int64_t* heights,
                                                  const int64_t* widths, int64_t n_levels, int64_t pooled_h,
                                                  int64_t pooled_w, int64_t c_stride, int64_t h_stride, int64_t w_stride,
                                                  size_t workspace_bytes);
int tvmi_multiscale_roi_align_backward(const void* grad, const void* rois, void* const* grad_inputs,
                                       const int64_t* heights, const int64_t* widths, const double* spatial_scales,
                                       int64_t n_levels, tvmi_dtype dt, int64_t N, int64_t C, int64_t K, int64_t pooled_h,
                                       int64_t pooled_w, int64_t sampling_ratio, int aligned, int64_t k_min, int64_t k_max,
                                       double canonical_scale, double canonical_level, double eps, int64_t n_stride,
                                       int64_t c_stride, int64_t h_stride, int64_t w_stride, void* workspace,
                                       size_t workspace_bytes, void* stream);

/* tvmi_multiscale_roi_align_forward taking the PER-IMAGE BOX LISTS MultiScaleRoIAlign.forward receives (ops/poolers.py:289-321)
 * instead of [K,5] rows (round 6): boxes[i] = [counts[i], 4] float32 device boxes of image i, num_images <= 64.  The rows of
 * convert_boxes_to_roi_format (ops/_utils.py:18-25) are written to rois_out [K,5] float32 (K = sum of counts; the backward takes
 * them) by the launch-order pre-pass of this call where that runs (7x7 / 14x14 bins with sampling_ratio 2, a multiple of 256
 * channels, workspace given) — by the units of the launch itself where the pre-pass is folded into it ("roi_align.fold_order") —
 * otherwise by tvmi_boxes_to_rois in front of the plain entry.  Same result either way. */
int tvmi_multiscale_roi_align_forward_boxes(const void* const* inputs, const int64_t* heights, const int64_t* widths,
                                            const double* spatial_scales, int64_t n_levels, const void* const* boxes,
                                            const int64_t* counts, int64_t num_images, void* rois_out, void* output, tvmi_dtype dt,
                                            int64_t N, int64_t C, int64_t pooled_h, int64_t pooled_w, int64_t sampling_ratio,
                                            int aligned, int64_t k_min, int64_t k_max, double canonical_scale, double canonical_level,
                                            double eps, void* workspace, size_t workspace_bytes, void* stream);

/* The detector step as ONE call (ABI 307): tvmi_multiscale_roi_align_forward_boxes (arguments up to workspace_bytes) AND
 * tvmi_nms_step (the arguments from nms_dets on, in that entry's order and meaning: the reference's batched_nms of
 * ops/boxes.py:50-102 + the padded per-image top-k of models/detection/roi_heads.py:668-723 over the step's proposals) — two
 * independent jobs of one detector step (neither reads what the other writes) that the reference issues as separate launches.
 * Where the RoIAlign call takes its one-launch route (7x7 bins, sampling_ratio 2, a multiple of 256 channels, workspace given) the
 * workgroups of the NMS ride in front of the RoIAlign grid: one launch on one stream, where two launches need two streams, a fork
 * and a join to overlap (~40 us of a 265 us step on the MI355X).  Anywhere else the two entries run one after the other on
 * `stream`.  Results are those of the two entries, bit for bit. */
int tvmi_multiscale_roi_align_forward_boxes_with_nms_step(
    const void* const* inputs, const int64_t* heights, const int64_t* widths, const double* spatial_scales, int64_t n_levels,
    const void* const* boxes, const int64_t* counts, int64_t num_images, void* rois_out, void* output, tvmi_dtype dt, int64_t N, int64_t C,
    int64_t pooled_h, int64_t pooled_w, int64_t sampling_ratio, int aligned, int64_t k_min, int64_t k_max, double canonical_scale,
    double canonical_level, double eps, void* workspace, size_t workspace_bytes, const float* nms_dets, const float* nms_scores,
    const int64_t* nms_seg, int64_t nms_n, int64_t nms_segments, double iou_threshold, void* nms_workspace, size_t nms_workspace_bytes,
    int64_t* keep_out, int64_t* num_keep_out, const int64_t* image_idx, const int64_t* labels, int64_t det_images, int64_t max_dets,
    float* payload, int64_t row_stride, int32_t* det_counts, int count_in_row, void* stream);

/* The same operation on channels_last feature maps (element (n,c,y,x) at ((n*H+y)*W+x)*C+c;
 * SURVEY.md §8f-2): lane = channel, taps are coalesced loads off a scalar base, no LDS window.
 * Output is still the reference's NCHW-contiguous [K,C,PH,PW]; rois are float32.  float32 maps (or float16 / bfloat16 with
 * an even channel count: two channels per lane), 7x7 bins, sampling_ratio 2, every level H,W >= 2; a single level with k_min == k_max is plain
 * roi_align.  The reference instead copies every map to NCHW first
 * (cuda/roi_align_kernel.cu:365 `input.contiguous()`).  `workspace` (optional, ABI 304): K ints of scratch for the launch
 * order of the RoIs — with it the units start sorted by (image, level, window-top band) and every XCD serves a contiguous
 * eighth of that order; NULL / too small: input order.  Results do not depend on it.
 */
int tvmi_multiscale_roi_align_forward_nhwc(const void* const* inputs, const int64_t* heights, const int64_t* widths,
                                           const double* spatial_scales, int64_t n_levels, const void* rois,
                                           void* output, tvmi_dtype dt, int64_t N, int64_t C, int64_t K,
                                           int64_t pooled_h, int64_t pooled_w, int64_t sampling_ratio, int aligned,
                                           int64_t k_min, int64_t k_max, double canonical_scale,
                                           double canonical_level, double eps, void* workspace, size_t workspace_bytes,
                                           void* stream);

/* ----------------------------------------------- RoIPool / PSRoIAlign / PSRoIPool ------
 * Replaces: cuda/roi_pool_kernel.cu:15-125,127-260, cuda/ps_roi_align_kernel.cu,
 * cuda/ps_roi_pool_kernel.cu:14-140; arithmetic follows cpu/roi_pool_kernel.cpp:24-134,
 * cpu/ps_roi_align_kernel.cpp:73-151,219-313, cpu/ps_roi_pool_kernel.cpp:22-155.
 * Side outputs (argmax, channel_mapping) are int32 [K, C_out, PH, PW]; PS variants need
 * C % (PH*PW) == 0 and produce C_out = C / (PH*PW) channels.  Backward launchers
 * accumulate into a caller-zeroed grad_input; PS backward reads a contiguous grad.
 */
int tvmi_roi_pool_forward(const void* input, const void* rois, void* output, int32_t* argmax,
                          tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                          int64_t pooled_h, int64_t pooled_w, double spatial_scale, void* stream);
int tvmi_roi_pool_backward(const void* grad, const void* rois, const int32_t* argmax, void* grad_input,
                           tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                           int64_t pooled_h, int64_t pooled_w, int64_t n_stride, int64_t c_stride,
                           int64_t h_stride, int64_t w_stride, void* stream);
/* roi_pool backward has two regimes, like roi_align's: when a gradient plane (H*W fp32) fits the LDS of a CU and the
 * dtype is F32 / F16 / BF16, one wave owns each plane, accumulates in LDS in a fixed order and WRITES every pixel of
 * grad_input (deterministic, no global atomics, no zero-fill needed) — tvmi_roi_pool_backward_overwrites(...) == 1;
 * otherwise it adds into a caller-zeroed grad_input with hardware atomics (cuda/roi_pool_kernel.cu:80-125).  A plane
 * above one LDS plane (36,864 pixels) is cut into up to 8 strips of whole rows with one owner wave each (roi_pool only);
 * tvmi_ps_roi_align_backward / tvmi_ps_roi_pool_backward own whole planes only: tvmi_ps_roi_backward_overwrites. */
int tvmi_roi_pool_backward_overwrites(tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W);
/* The same question for tvmi_ps_roi_align_backward / tvmi_ps_roi_pool_backward (whole planes only). */
int tvmi_ps_roi_backward_overwrites(tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W);
int tvmi_ps_roi_align_forward(const void* input, const void* rois, void* output,
                              int32_t* channel_mapping, tvmi_dtype dt, int64_t N, int64_t C, int64_t H,
                              int64_t W, int64_t K, int64_t pooled_h, int64_t pooled_w,
                              double spatial_scale, int64_t sampling_ratio, void* stream);
int tvmi_ps_roi_align_backward(const void* grad, const void* rois, const int32_t* channel_mapping,
                               void* grad_input, tvmi_dtype dt, int64_t N, int64_t C, int64_t H,
                               int64_t W, int64_t K, int64_t pooled_h, int64_t pooled_w,
                               double spatial_scale, int64_t sampling_ratio, void* stream);
int tvmi_ps_roi_pool_forward(const void* input, const void* rois, void* output, int32_t* channel_mapping,
                             tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                             int64_t pooled_h, int64_t pooled_w, double spatial_scale, void* stream);
int tvmi_ps_roi_pool_backward(const void* grad, const void* rois, const int32_t* channel_mapping,
                              void* grad_input, tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W,
                              int64_t K, int64_t pooled_h, int64_t pooled_w, double spatial_scale,
                              void* stream);

/* ------------------------------------------------------------ deform_conv2d ------------
 * Replaces: cuda/deform_conv2d_kernel.cu:136-209 (deformable_im2col) + the per-group
 * addmm_ loop at :1035-1255 with ONE fused offset-gather -> LDS -> fp32-MFMA kernel (fp32),
 * or a direct kernel (other dtypes / tiny channel counts); semantics of
 * cpu/deform_conv2d_kernel.cpp:95-209,921-1151.  Tensors: input [B,C,H,W],
 * weight [OC,C/groups,kh,kw], offset [B,2*og*kh*kw,oh,ow], mask [B,og*kh*kw,oh,ow]
 * (ignored unless use_mask), bias [OC], output [B,OC,oh,ow] (fully overwritten).
 * `workspace` holds the re-laid-out weights (tvmi_deform_conv2d_workspace_bytes).
 */
size_t tvmi_deform_conv2d_workspace_bytes(tvmi_dtype dt, int64_t C, int64_t OC, int64_t kh, int64_t kw,
                                          int64_t groups);
/* Workspace of the forward call for a given input: the re-laid-out weights and, for fp16 / bf16 inputs whose channel counts
 * (C, C / groups, C / offset_groups) are multiples of 8, a [B, H*W, C] copy of the input that the fused MFMA kernel samples
 * with one 16-byte load per corner and channel octet, in 32-deep K slabs.  A caller that only provides
 * tvmi_deform_conv2d_workspace_bytes() gets the planar-gather kernel: same values (the same bits when C / offset_groups is a
 * multiple of 16), slower. */
size_t tvmi_deform_conv2d_forward_workspace_bytes(tvmi_dtype dt, int64_t B, int64_t C, int64_t H, int64_t W, int64_t OC,
                                                  int64_t kh, int64_t kw, int64_t groups, int64_t offset_groups);
int tvmi_deform_conv2d_forward(const void* input, const void* weight, const void* offset, const void* mask,
                               const void* bias, void* output, tvmi_dtype dt, int64_t B, int64_t C,
                               int64_t H, int64_t W, int64_t OC, int64_t kh, int64_t kw, int64_t stride_h,
                               int64_t stride_w, int64_t pad_h, int64_t pad_w, int64_t dil_h, int64_t dil_w,
                               int64_t groups, int64_t offset_groups, int use_mask, void* workspace,
                               size_t workspace_bytes, void* stream);
/* The whole backward pass in one call, no materialised `columns` and no library GEMM.  Replaces the reference's
 * backward_gradient_inputs + backward_gradient_parameters (cpu/deform_conv2d_kernel.cpp:554-919, 1153-1226;
 * cuda/deform_conv2d_kernel.cu:752-1033, 1257-1330: a GEMM per weight group into [C*kh*kw, B*oh*ow], col2im_coord, col2im,
 * im2col into the same buffer, a second GEMM).  grad_out [B,OC,oh,ow]; grad_input / grad_weight / grad_offset / grad_mask
 * (ignored unless use_mask) / grad_bias (may be NULL) have the shapes of input / weight / offset / mask / [OC] and are FULLY
 * overwritten.  fp32 / fp16 / bf16 problems with at least 16 channels per weight group on both sides (and whole 32-channel
 * blocks per offset group) contract on the fp32 matrix cores inside two fused kernels — sums in fp32, 16-bit results rounded
 * once; depthwise problems (groups = C = OC, 64 / 128 / 256 channels per offset group) take a channels-last kernel, everything
 * else (fp64, tiny channel counts) direct kernels with the same fusion.  grad_input is accumulated with float adds whose order
 * is not fixed, like the reference's (not bit-reproducible run to run).  `workspace`:
 * tvmi_deform_conv2d_backward_workspace_bytes (re-laid-out weights, the [tap][oc][ic] weight-gradient sums, fp32 sums of
 * 16-bit problems, channels-last copies of input / grad_out and channels-last grad_input sums where a kernel wants them; 0 for
 * fp64).  The query takes the geometry of the call (ABI 304: stride / padding / dilation instead of the output size), because
 * the route — and with it the buffers — depends on it: a strided or dilated 3 x 3 problem whose window no longer fits the
 * owner kernel's LDS asks for no channels-last buffers. */
size_t tvmi_deform_conv2d_backward_workspace_bytes(tvmi_dtype dt, int64_t B, int64_t C, int64_t H, int64_t W, int64_t OC,
                                                   int64_t kh, int64_t kw, int64_t stride_h, int64_t stride_w, int64_t pad_h,
                                                   int64_t pad_w, int64_t dil_h, int64_t dil_w, int64_t groups,
                                                   int64_t offset_groups);
int tvmi_deform_conv2d_backward(const void* grad_out, const void* input, const void* weight, const void* offset,
                                const void* mask, void* grad_input, void* grad_weight, void* grad_offset, void* grad_mask,
                                void* grad_bias, tvmi_dtype dt, int64_t B, int64_t C, int64_t H, int64_t W, int64_t OC,
                                int64_t kh, int64_t kw, int64_t stride_h, int64_t stride_w, int64_t pad_h, int64_t pad_w,
                                int64_t dil_h, int64_t dil_w, int64_t groups, int64_t offset_groups, int use_mask,
                                void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------- detection post-processing --------
 * One launch from the score-ordered keep list of a (batched) NMS to the fixed-shape payload
 * that is all-gathered across GPUs: dets [num_images, max_dets, 6] fp32 = (x1,y1,x2,y2,score,
 * label) zero padded, counts [num_images] int32.  Replaces the index / split / top-k glue of
 * torchvision/models/detection/roi_heads.py:720-735 and the pickled all_gather_object of
 * references/detection/utils.py:70-83.  boxes [N,4] fp32, scores [N] fp32, labels [N] int64 or
 * NULL, image_idx [N] int64, keep [num_keep] int64 (descending score).  One workgroup per image; num_images <= 65535.
 */
int tvmi_pack_detections(const float* boxes, const float* scores, const int64_t* labels,
                         const int64_t* image_idx, const int64_t* keep, int64_t num_keep, int64_t num_images,
                         int64_t max_dets, float* dets, int32_t* counts, void* stream);
/* Same, with the length of `keep` read on the device (num_keep_dev [1] int64, e.g. tvmi_nms's
 * num_keep_out on the same stream; clamped to [0, keep_capacity]): NMS -> packed payload with
 * no host synchronisation in between, so the chain can be captured in a hipGraph.  A NEGATIVE
 * length — the error sentinel of the sync-free NMS entries (segment above its size limit, id
 * outside the promised range) — is passed on: counts[b] = -1 for every image, zero payload. */
int tvmi_pack_detections_devcount(const float* boxes, const float* scores, const int64_t* labels,
                                  const int64_t* image_idx, const int64_t* keep, int64_t keep_capacity,
                                  const int64_t* num_keep_dev, int64_t num_images, int64_t max_dets, float* dets,
                                  int32_t* counts, void* stream);
/* The same launch writing the COLLECTIVE PAYLOAD itself: row b of `payload` ([num_images, row_stride] floats, row_stride >=
 * max_dets * 6 + 1) = the max_dets x 6 detection block of image b followed by its count as a float — what
 * vision_amd/sharding.py all-gathers (replaces the pickled all_gather_object of references/detection/utils.py:70-83) with no
 * assembly launches between the NMS and the collective.  `counts` (int32) is optional (NULL: not written). */
int tvmi_pack_detections_payload(const float* boxes, const float* scores, const int64_t* labels, const int64_t* image_idx,
                                 const int64_t* keep, int64_t keep_capacity, const int64_t* num_keep_dev, int64_t num_images,
                                 int64_t max_dets, float* payload, int64_t row_stride, int32_t* counts, void* stream);

/* convert_boxes_to_roi_format (torchvision/ops/_utils.py:18-25) in one launch: per-image box
 * lists [n_i,4] (dtype dt, contiguous) -> rois [sum n_i, 5] = (image index, x1, y1, x2, y2).
 * num_images <= 64. */
int tvmi_boxes_to_rois(const void* const* boxes, const int64_t* counts, int64_t num_images, void* rois,
                       tvmi_dtype dt, void* stream);

/* Candidate generation for the detector's two post-processing stages, batched over images
 * (one launch each; the segmented NMS that follows is tvmi_nms with segment ids, the top-k
 * packing is tvmi_pack_detections):
 *  tvmi_detection_candidates — RoIHeads.postprocess_detections, models/detection/roi_heads.py:
 *    680-722: softmax over C classes, BoxCoder.decode_single per class (_utils.py:183-224,
 *    weights are divisors), clip_boxes_to_image (ops/boxes.py:171-199), background dropped,
 *    valid = score > score_thresh && w >= min_size && h >= min_size (ops/boxes.py:148-168).
 *    class_logits [R,C], box_regression [R,4C], proposals [R,4], row_image [R] int32,
 *    image_hw [B,2] = (height,width) -> cand_boxes [R,C-1,4], cand_scores [R,C-1],
 *    cand_valid [R,C-1] uint8 (class c at column c-1).
 *  tvmi_rpn_candidates — RegionProposalNetwork.filter_proposals, models/detection/rpn.py:
 *    266-286 for the per-level top-k survivors `top_idx` [B,T] (indices into the A anchors):
 *    gather, sigmoid, clip, valid = w,h >= min_size && prob >= score_thresh, level id from
 *    level_offsets [L] (first anchor of each level).  boxes_in [B,A,4] are the decoded
 *    proposals, or — when `deltas` [B,A,4] is not NULL — the anchors, decoded here with
 *    weights (1,1,1,1) for the survivors only (rpn.py:364-366 decodes every anchor).
 */
int tvmi_detection_candidates(const float* class_logits, const float* box_regression, const float* proposals,
                              const int32_t* row_image, const float* image_hw, int64_t R, int64_t C, int64_t B,
                              float wx, float wy, float ww, float wh, float bbox_xform_clip, float score_thresh,
                              float min_size, float* cand_boxes, float* cand_scores, uint8_t* cand_valid, void* stream);
int tvmi_rpn_candidates(const float* objectness, const float* boxes_in, const float* deltas, const int64_t* top_idx,
                        const int64_t* level_offsets, const float* image_hw, int64_t B, int64_t A, int64_t T, int64_t L,
                        float bbox_xform_clip, float score_thresh, float min_size, float* out_boxes, float* out_scores,
                        int64_t* out_levels, uint8_t* out_valid, void* stream);

/* Pairwise axis-aligned IoU of xyxy boxes: boxes1 [N,4], boxes2 [M,4] -> out [N,M], dt F32 or F64.
 *   mode 0  box_iou               torchvision/ops/boxes.py:314-391
 *   mode 1  generalized_box_iou   :409-436
 *   mode 2  distance_box_iou      :469-515 (`eps` added to the squared enclosing diagonal)
 *   mode 3  complete_box_iou      :439-466 (alpha = v / (1 - iou + v + eps))
 * 16-bit boxes are upcast by the caller like ops/_utils.py:72-84; for modes 0 / 1 source_16bit = 1 (float16) / 2
 * (bfloat16) makes the kernel round rb - lt to that type first, as the reference's `_upcast(rb - lt)` does (modes
 * 2 / 3 upcast the boxes before any arithmetic: pass 0).  One launch instead of ~10-25 broadcast tensor ops; same
 * operations, same order.
 */
int tvmi_box_iou_pairwise(const void* boxes1, const void* boxes2, void* out, tvmi_dtype dt, int64_t N, int64_t M,
                          int mode, int source_16bit, double eps, void* stream);

/* ---------------------------------------------------------- box_iou_rotated ------------
 * Replaces: cuda/box_iou_rotated_kernel.cu:41-188; semantics of box_iou_rotated_utils.h:67-383
 * and cpu/box_iou_rotated_kernel.cpp:29-115.  boxes [N,5]/[M,5] = (cx,cy,w,h,angle_deg),
 * F32 or F64; ious [N,M] is always float32.
 */
int tvmi_box_iou_rotated(const void* boxes1, const void* boxes2, float* ious, tvmi_dtype dt, int64_t N,
                         int64_t M, void* stream);

/* ------------------------------------------------------------------- resize ------------
 * The reference's resize path (transforms/v2/functional/_geometry.py:283-362,
 * transforms/_functional_tensor.py:441-474, models/detection/transform.py:65-72) ends in
 * aten::upsample_{bilinear2d,bicubic2d,nearest2d}, _upsample_nearest_exact2d and
 * _upsample_{bilinear2d,bicubic2d}_aa; these launchers implement that arithmetic
 * (ATen/native/UpSample.h, ATen/native/cuda/UpSample.cuh of torch 2.10) for contiguous
 * planes: input [NC,IH,IW] -> output [NC,OH,OW].  scale_h/scale_w <= 0 mean "not given"
 * (scale = in/out), otherwise they are the user scale factors exactly like the aten ops'
 * optional `scales_*` arguments.  mode for the anti-aliased entry: 0 bilinear, 1 bicubic.
 */
int tvmi_upsample_bilinear2d(const void* input, void* output, tvmi_dtype dt, int64_t NC, int64_t IH, int64_t IW,
                             int64_t OH, int64_t OW, int align_corners, double scale_h, double scale_w,
                             void* stream);
int tvmi_upsample_bicubic2d(const void* input, void* output, tvmi_dtype dt, int64_t NC, int64_t IH, int64_t IW,
                            int64_t OH, int64_t OW, int align_corners, double scale_h, double scale_w,
                            void* stream);
int tvmi_upsample_nearest2d(const void* input, void* output, tvmi_dtype dt, int64_t NC, int64_t IH, int64_t IW,
                            int64_t OH, int64_t OW, int exact, double scale_h, double scale_w, void* stream);
size_t tvmi_upsample_aa2d_workspace_bytes(int mode, int64_t IH, int64_t IW, int64_t OH, int64_t OW,
                                          int align_corners, double scale_h, double scale_w);
int tvmi_upsample_aa2d(const void* input, void* output, tvmi_dtype dt, int mode, int64_t NC, int64_t IH,
                       int64_t IW, int64_t OH, int64_t OW, int align_corners, double scale_h, double scale_w,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Nearest / nearest-exact for ANY element type (elem_bytes 1, 2, 4 or 8: the op is a copy) — the reference resizes uint8
 * images and masks with InterpolationMode.NEAREST without a cast (transforms/v2/functional/_geometry.py:316-323), i.e.
 * through aten::upsample_nearest2d on integer tensors. */
int tvmi_upsample_nearest2d_any(const void* input, void* output, int64_t elem_bytes, int64_t NC, int64_t IH, int64_t IW,
                                int64_t OH, int64_t OW, int exact, double scale_h, double scale_w, void* stream);

/* The six modes on channels_last tensors: input [N,IH,IW,C] -> output [N,OH,OW,C] in memory (what ATen returns for a
 * channels_last input; resize_image preserves the format on purpose, _geometry.py:324-338).  mode: 0 nearest, 1 nearest-exact,
 * 2 bilinear, 3 bicubic; antialias only with 2 / 3.  F32 / F16 / BF16; the nearest modes for any element size through
 * tvmi_upsample_nearest2d_nhwc_any. */
size_t tvmi_upsample2d_nhwc_workspace_bytes(int mode, int antialias, int64_t IH, int64_t IW, int64_t OH, int64_t OW,
                                            int align_corners, double scale_h, double scale_w);
int tvmi_upsample2d_nhwc(const void* input, void* output, tvmi_dtype dt, int mode, int antialias, int64_t N, int64_t C,
                         int64_t IH, int64_t IW, int64_t OH, int64_t OW, int align_corners, double scale_h, double scale_w,
                         void* workspace, size_t workspace_bytes, void* stream);
int tvmi_upsample_nearest2d_nhwc_any(const void* input, void* output, int64_t elem_bytes, int64_t N, int64_t C, int64_t IH,
                                     int64_t IW, int64_t OH, int64_t OW, int exact, double scale_h, double scale_w,
                                     void* stream);

/* Backward of the six modes above — aten::upsample_{nearest2d,bilinear2d,bicubic2d}_backward,
 * _upsample_nearest_exact2d_backward, _upsample_{bilinear2d,bicubic2d}_aa_backward (the autograd formulas of the aten ops
 * torchvision's resize / FPN top-down path / segmentation heads go through: ops/feature_pyramid_network.py:194,
 * models/segmentation/_utils.py:27,33).  mode: 0 nearest, 1 nearest-exact, 2 bilinear, 3 bicubic; antialias only with 2 / 3.
 * The sizes are the FORWARD's: grad_output [NC,OH,OW] -> grad_input [NC,IH,IW], every element of grad_input written (no
 * zero fill needed).  Gather formulation with a fixed summation order: bit-reproducible, unlike ATen's atomicAdd kernels.
 * F32 / F16 / BF16 (fp32 accumulation, rounded once). */
size_t tvmi_upsample2d_backward_workspace_bytes(int mode, int antialias, int64_t IH, int64_t IW, int64_t OH, int64_t OW,
                                                int align_corners, double scale_h, double scale_w);
int tvmi_upsample2d_backward(const void* grad_output, void* grad_input, tvmi_dtype dt, int mode, int antialias, int64_t NC,
                             int64_t IH, int64_t IW, int64_t OH, int64_t OW, int align_corners, double scale_h,
                             double scale_w, void* workspace, size_t workspace_bytes, void* stream);

/* GeneralizedRCNNTransform.forward for a batch (models/detection/transform.py:119-255) in one
 * launch: images[i] is [C,H_i,W_i] (dt, contiguous); every output pixel of image i inside
 * [out_h_i, out_w_i] is the bilinear (align_corners=False) sample of ((x - mean_c) / std_c), the
 * rest of the [num_images, C, padded_h, padded_w] batch is zero.  The caller applies the
 * reference's size rules (_resize_image_and_masks :25-72, batch_images :228-246) on the host.
 */
int tvmi_normalize_resize_batch(const void* const* images, const int64_t* heights, const int64_t* widths,
                                const int64_t* out_heights, const int64_t* out_widths, int64_t num_images,
                                int64_t channels, const float* mean, const float* stdv, void* output, tvmi_dtype dt,
                                int64_t padded_h, int64_t padded_w, void* stream);

/* ------------------------------------------------------------------- mask paste --------
 * Batched replacement of paste_masks_in_image (models/detection/roi_heads.py:486-500, with
 * expand_masks :404-413, expand_boxes :378-395 and paste_mask_in_image :416-437 folded in):
 * masks [N,M,M] (dt), boxes [N,4] float32 xyxy in image pixels, output [N,im_h,im_w] (dt),
 * every element written.  The reference loops over detections in Python; this is one launch.
 */
int tvmi_paste_masks(const void* masks, const float* boxes, void* output, tvmi_dtype dt, int64_t N, int64_t M,
                     int64_t im_h, int64_t im_w, int64_t padding, void* stream);

/* ------------------------------------------------------------------ quantized ------
 * torchvision::qroi_align (quantized/cpu/qroi_align_kernel.cpp:22-178, CPU-only in the reference): `input` [1, C, H, W] and
 * `rois` [K, 5] are integer tensors of type `dt` with explicit (scale, zero point); bilinear sums on the raw integers,
 * dequantised once, averaged, re-quantised with round-half-even and saturated to `dt`; batch index 0 for every RoI.
 * Bit-identical to the reference kernel.  (torchvision::qnms needs no entry of its own: its arithmetic is tvmi_nms_blocking on
 * the boxes widened to float32 — quantized/cpu/qnms_kernel.cpp:60-120 — see the dispatcher glue.) */
int tvmi_qroi_align_forward(const void* input, const void* rois, void* output, tvmi_int_dtype dt, int64_t C, int64_t H, int64_t W,
                            int64_t K, int64_t pooled_h, int64_t pooled_w, double input_scale, int64_t input_zero_point,
                            double rois_scale, int64_t rois_zero_point, double spatial_scale, int64_t sampling_ratio, int aligned,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TVMI_H_ */
