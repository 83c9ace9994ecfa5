// postprocess.hip — detection post-processing glue that the reference does with a dozen tiny
// torch kernels per image (models/detection/roi_heads.py:680-737 `postprocess_detections`:
// index by keep -> per-image split -> top-k; references/detection/utils.py:70-83 then pickles
// the per-image dicts for all_gather_object).  Here ONE launch turns the score-ordered keep list
// of a (batched) NMS into the fixed-shape payload that is all-gathered over RCCL:
//   dets[b, r, :] = (x1, y1, x2, y2, score, label) of the r-th best kept box of image b,
//   zero padded to `max_dets`; counts[b] = min(#kept in image b, max_dets).
// Order inside an image is the order of `keep` (descending score), i.e. a stable partition.
#include "tvmi_common.h"

namespace tvmi {
namespace {

constexpr int kPackThreads = 1024;
constexpr int kPackWaves = kPackThreads / 64;

// One workgroup PER IMAGE: workgroup b walks the score-ordered keep list, takes the entries of image b in order and
// stops as soon as it holds max_dets of them — in a detector step (tens of thousands of kept candidates for a payload
// of 100 per image) that is after the first chunk or two.  (Round 2 ran the whole batch through ONE workgroup: 92 us in
// the Mask R-CNN step, more than the RoIAlign launches it follows.)
__global__ __launch_bounds__(kPackThreads) void pack_detections_kernel(
    const float* __restrict__ boxes, const float* __restrict__ scores, const int64_t* __restrict__ labels,
    const int64_t* __restrict__ image_idx, const int64_t* __restrict__ keep, int64_t num_keep,
    const int64_t* __restrict__ num_keep_dev, int num_images, int max_dets, float* __restrict__ dets,
    int32_t* __restrict__ counts, int64_t row_stride, int count_in_row) {
  // the keep list may come straight from an NMS launch on the same stream: its length then lives on the device.
  // A negative length is the sync-free NMS's error sentinel (a segment above its size limit, an id outside the
  // promised range): the keep list is then unspecified, and the sentinel is passed on as counts[b] = -1 (zero
  // payload) instead of being mistaken for "nothing kept".
  const bool poisoned = num_keep_dev && *num_keep_dev < 0;
  if (num_keep_dev) num_keep = min(max(*num_keep_dev, (int64_t)0), num_keep);
  __shared__ int s_wave[kPackWaves];   // this chunk: entries of this image per wave
  __shared__ int s_cnt;                // entries of this image so far
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int b = blockIdx.x;
  float* my = dets + (int64_t)b * row_stride;   // row_stride = max_dets * 6 for the plain [B, D, 6] layout
  for (int i = tid; i < max_dets * 6; i += kPackThreads) my[i] = 0.f;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  for (int64_t c0 = 0; c0 < num_keep; c0 += kPackThreads) {
    const int64_t e = c0 + tid;
    int64_t src = 0;
    bool mine = false;
    if (e < num_keep) {
      src = keep[e];
      mine = image_idx[src] == (int64_t)b;
    }
    const unsigned long long bal = __ballot(mine);
    if (lane == 0) s_wave[wave] = __popcll(bal);
    const int before = s_cnt;            // stable until the barrier below
    __syncthreads();
    if (mine) {
      int r = before + __popcll(bal & ((1ull << lane) - 1ull));
      for (int w = 0; w < wave; ++w) r += s_wave[w];
      if (r < max_dets) {
        float* d = my + (int64_t)r * 6;
        const float4 bx = *reinterpret_cast<const float4*>(boxes + src * 4);
        d[0] = bx.x;
        d[1] = bx.y;
        d[2] = bx.z;
        d[3] = bx.w;
        d[4] = scores[src];
        d[5] = labels ? (float)labels[src] : 0.f;
      }
    }
    __syncthreads();
    if (tid == 0) {
      int add = 0;
      for (int w = 0; w < kPackWaves; ++w) add += s_wave[w];
      s_cnt = before + add;
    }
    __syncthreads();
    // the keep list is in score order: once this image holds max_dets detections nothing later can enter its payload
    if (s_cnt >= max_dets) break;
  }
  if (tid == 0) {
    const int n = poisoned ? -1 : min(s_cnt, max_dets);
    if (counts) counts[b] = n;
    if (count_in_row) my[(int64_t)max_dets * 6] = (float)n;   // collective payload: the count rides in the row
  }
}


// ---------------------------------------------------------------------------------------
// Candidate generation for the two post-processing stages of a two-stage detector, batched
// over the images of a step.  Both replace a chain of ~15 elementwise /
// indexing torch kernels PER IMAGE in the reference with one launch per batch; what follows
// them (class- / level-segmented NMS, top-k packing) are tvmi_nms with segment ids and
// tvmi_pack_detections.  Arithmetic follows the reference op by op in fp32 (this TU is built
// with -ffp-contract=off): BoxCoder.decode_single (models/detection/_utils.py:183-224),
// clip_boxes_to_image (ops/boxes.py:171-199), remove_small_boxes (ops/boxes.py:148-168).

// decode_single for one (box, code) pair; weights are divisors exactly like the reference
__device__ __forceinline__ float4 decode_box(const float4 box, const float4 code, float wx, float wy, float ww, float wh,
                                             float clip) {
  const float width = box.z - box.x, height = box.w - box.y;
  const float ctr_x = box.x + 0.5f * width, ctr_y = box.y + 0.5f * height;
  const float dx = code.x / wx, dy = code.y / wy;
  const float dw = fminf(code.z / ww, clip), dh = fminf(code.w / wh, clip);
  const float pcx = dx * width + ctr_x, pcy = dy * height + ctr_y;
  const float pw = expf(dw) * width, ph = expf(dh) * height;
  const float hw = 0.5f * pw, hh = 0.5f * ph;
  return make_float4(pcx - hw, pcy - hh, pcx + hw, pcy + hh);
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// roi_heads.py:680-737 up to (not including) batched_nms: softmax over classes, per-class
// decode, clip, drop background / low score / small boxes.  One wave per RoI row.
__global__ __launch_bounds__(256) void det_candidates_kernel(const float* __restrict__ logits, const float* __restrict__ reg,
                                                             const float* __restrict__ props,
                                                             const int32_t* __restrict__ row_image,
                                                             const float* __restrict__ image_hw, int R, int C, int B,
                                                             float wx, float wy, float ww, float wh, float clip,
                                                             float score_thresh, float min_size,
                                                             float* __restrict__ cand_boxes, float* __restrict__ cand_scores,
                                                             uint8_t* __restrict__ cand_valid) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (r >= R) return;
  const float* lg = logits + (int64_t)r * C;
  float mx = -INFINITY;
  for (int c = lane; c < C; c += 64) mx = fmaxf(mx, lg[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int c = lane; c < C; c += 64) sum += expf(lg[c] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float4 box = *reinterpret_cast<const float4*>(props + (int64_t)r * 4);
  int img = row_image[r];
  img = min(max(img, 0), B - 1);
  const float im_h = image_hw[2 * img], im_w = image_hw[2 * img + 1];
  for (int c = 1 + lane; c < C; c += 64) {
    const float score = expf(lg[c] - mx) / sum;
    const float4 code = *reinterpret_cast<const float4*>(reg + ((int64_t)r * C + c) * 4);
    float4 b = decode_box(box, code, wx, wy, ww, wh, clip);
    b.x = clampf(b.x, 0.f, im_w);
    b.z = clampf(b.z, 0.f, im_w);
    b.y = clampf(b.y, 0.f, im_h);
    b.w = clampf(b.w, 0.f, im_h);
    const bool ok = score > score_thresh && (b.z - b.x) >= min_size && (b.w - b.y) >= min_size;
    const int64_t o = (int64_t)r * (C - 1) + (c - 1);
    *reinterpret_cast<float4*>(cand_boxes + o * 4) = b;
    cand_scores[o] = score;
    cand_valid[o] = ok ? 1 : 0;
  }
}

// rpn.py:242-286 up to (not including) batched_nms, for the per-level top-k survivors:
// gather, (optionally decode anchors + deltas: rpn.py:364-366 does it for EVERY anchor before
// the top-k, here only the survivors are decoded — same values, elementwise), sigmoid, clip,
// small-box and score filter, level id.  One lane per (image, survivor).
__global__ __launch_bounds__(256) void rpn_candidates_kernel(const float* __restrict__ objectness,
                                                             const float* __restrict__ boxes_in,
                                                             const float* __restrict__ deltas,
                                                             const int64_t* __restrict__ top_idx,
                                                             const int64_t* __restrict__ level_offsets,
                                                             const float* __restrict__ image_hw, int B, int64_t A, int T,
                                                             int L, float clip, float score_thresh, float min_size,
                                                             float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                             int64_t* __restrict__ out_levels, uint8_t* __restrict__ out_valid) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * T) return;
  const int b = (int)(i / T);
  int64_t a = top_idx[i];
  a = min(max(a, (int64_t)0), A - 1);
  float4 box = *reinterpret_cast<const float4*>(boxes_in + ((int64_t)b * A + a) * 4);
  if (deltas) box = decode_box(box, *reinterpret_cast<const float4*>(deltas + ((int64_t)b * A + a) * 4), 1.f, 1.f, 1.f, 1.f, clip);
  const float im_h = image_hw[2 * b], im_w = image_hw[2 * b + 1];
  box.x = clampf(box.x, 0.f, im_w);
  box.z = clampf(box.z, 0.f, im_w);
  box.y = clampf(box.y, 0.f, im_h);
  box.w = clampf(box.w, 0.f, im_h);
  const float prob = 1.f / (1.f + expf(-objectness[(int64_t)b * A + a]));
  int lvl = 0;
  for (int l = 1; l < L; ++l) lvl += a >= level_offsets[l] ? 1 : 0;
  *reinterpret_cast<float4*>(out_boxes + i * 4) = box;
  out_scores[i] = prob;
  out_levels[i] = lvl;
  out_valid[i] = ((box.z - box.x) >= min_size && (box.w - box.y) >= min_size && prob >= score_thresh) ? 1 : 0;
}

// convert_boxes_to_roi_format (ops/_utils.py:18-25: cat(boxes), a full_like id column per image, two more
// cats = 7 tiny launches for a batch of 4) as one launch: rois[k] = (image index, x1, y1, x2, y2).
constexpr int kRoiFmtMaxImages = 64;
struct BoxLists {
  const void* ptr[kRoiFmtMaxImages];
  int end[kRoiFmtMaxImages];  // exclusive prefix end of every image's boxes in the concatenation
  int n;
};
template <typename T>
__global__ __launch_bounds__(256) void boxes_to_rois_kernel(BoxLists bl, T* __restrict__ rois, int K) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  int img = 0;
  for (int i = 0; i < bl.n - 1; ++i) img += k >= bl.end[i] ? 1 : 0;
  const int first = img > 0 ? bl.end[img - 1] : 0;
  const T* b = static_cast<const T*>(bl.ptr[img]) + (int64_t)(k - first) * 4;
  T* r = rois + (int64_t)k * 5;
  st(r, (float)img);
  r[1] = b[0];
  r[2] = b[1];
  r[3] = b[2];
  r[4] = b[3];
}

// Pairwise axis-aligned IoU / generalized / distance / complete IoU (ops/boxes.py:314-391, 409-436, 439-515): the reference builds
// [N,M,2] temporaries with ~10 elementwise launches; here one thread per (i, j), j fastest, same
// operations in the same order (this TU has no FP contraction), so values are bit-identical to the
// reference's CPU tensor math.
// `src16`: the boxes came from float16 (1) / bfloat16 (2) tensors — the reference then forms rb - lt in that type
// BEFORE upcasting (`_upcast(rb - lt)`), so the difference is rounded to 16 bits here too.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void box_iou_pairwise_kernel(const T* __restrict__ b1, const T* __restrict__ b2,
                                                               T* __restrict__ out, int N, int M, int src16, T eps) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= M) return;
  const T ax1 = b1[i * 4 + 0], ay1 = b1[i * 4 + 1], ax2 = b1[i * 4 + 2], ay2 = b1[i * 4 + 3];
  const T bx1 = b2[(int64_t)j * 4 + 0], by1 = b2[(int64_t)j * 4 + 1], bx2 = b2[(int64_t)j * 4 + 2], by2 = b2[(int64_t)j * 4 + 3];
  const T area1 = (ax2 - ax1) * (ay2 - ay1), area2 = (bx2 - bx1) * (by2 - by1);
  auto tmax = [](T a, T b) { return (a != a || a > b) ? a : b; };  // torch.max / torch.min propagate NaN
  auto tmin = [](T a, T b) { return (a != a || a < b) ? a : b; };
  auto clamp0 = [](T v) { return v < (T)0 ? (T)0 : v; };           // clamp(min=0) keeps NaN
  auto r16 = [src16](T v) -> T {
    if (src16 == 1) return (T)__half2float(__float2half_rn((float)v));
    if (src16 == 2) return (T)__bfloat162float(__float2bfloat16((float)v));
    return v;
  };
  const T w = clamp0(r16(tmin(ax2, bx2) - tmax(ax1, bx1))), h = clamp0(r16(tmin(ay2, by2) - tmax(ay1, by1)));
  const T inter = w * h;
  const T uni = area1 + area2 - inter;
  T r = inter / uni;
  if (MODE == 1) {
    const T wi = clamp0(r16(tmax(ax2, bx2) - tmin(ax1, bx1))), hi = clamp0(r16(tmax(ay2, by2) - tmin(ay1, by1)));
    const T areai = wi * hi;
    r = r - (areai - uni) / areai;
  }
  if (MODE >= 2) {
    // _box_diou_iou (ops/boxes.py:494-515), the boxes are already upcast there: no 16-bit rounding of differences.
    // torch.min / max of the corner tensors: lti = min(lt1, lt2), rbi = max(rb1, rb2); whi = clamp(rbi - lti, 0)
    const T wi = clamp0(tmax(ax2, bx2) - tmin(ax1, bx1)), hi = clamp0(tmax(ay2, by2) - tmin(ay1, by1));
    const T diag2 = (wi * wi) + (hi * hi) + eps;
    const T xp = (ax1 + ax2) / (T)2, yp = (ay1 + ay2) / (T)2, xg = (bx1 + bx2) / (T)2, yg = (by1 + by2) / (T)2;
    const T dx = xp - xg, dy = yp - yg;
    const T cdist2 = (dx * dx) + (dy * dy);
    const T iou = r;
    r = iou - (cdist2 / diag2);
    if (MODE == 3) {
      // complete_box_iou (:439-466): v = 4/pi^2 * (atan(w_p/h_p) - atan(w_g/h_g))^2, alpha = v / (1 - iou + v + eps)
      const T wp = ax2 - ax1, hp = ay2 - ay1, wg = bx2 - bx1, hg = by2 - by1;
      const T d = atan(wp / hp) - atan(wg / hg);
      const T v = (T)(4.0 / (3.141592653589793 * 3.141592653589793)) * (d * d);
      const T alpha = v / ((T)1 - iou + v + eps);
      r = r - alpha * v;
    }
  }
  out[(int64_t)i * M + j] = r;
}

}  // namespace
}  // namespace tvmi

extern "C" int tvmi_pack_detections(const float* boxes, const float* scores, const int64_t* labels,
                                    const int64_t* image_idx, const int64_t* keep, int64_t num_keep, int64_t num_images,
                                    int64_t max_dets, float* dets, int32_t* counts, void* stream) {
  TVMI_CHECK_ARG(num_images >= 0 && max_dets >= 0 && num_keep >= 0, "pack_detections: negative size");
  if (num_images == 0) return 0;
  TVMI_CHECK_ARG(num_images <= 65535, "pack_detections: at most 65535 images per call");
  TVMI_CHECK_ARG(dets && counts && (num_keep == 0 || (boxes && scores && image_idx && keep)),
                 "pack_detections: null pointer");
  tvmi::pack_detections_kernel<<<dim3((unsigned)num_images), dim3(tvmi::kPackThreads), 0, static_cast<hipStream_t>(stream)>>>(
      boxes, scores, labels, image_idx, keep, num_keep, nullptr, (int)num_images, (int)max_dets, dets, counts, max_dets * 6, 0);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_pack_detections");
}

extern "C" int tvmi_pack_detections_devcount(const float* boxes, const float* scores, const int64_t* labels,
                                             const int64_t* image_idx, const int64_t* keep, int64_t keep_capacity,
                                             const int64_t* num_keep_dev, int64_t num_images, int64_t max_dets,
                                             float* dets, int32_t* counts, void* stream) {
  TVMI_CHECK_ARG(num_images >= 0 && max_dets >= 0 && keep_capacity >= 0, "pack_detections: negative size");
  if (num_images == 0) return 0;
  TVMI_CHECK_ARG(num_images <= 65535, "pack_detections: at most 65535 images per call");
  TVMI_CHECK_ARG(dets && counts && num_keep_dev && (keep_capacity == 0 || (boxes && scores && image_idx && keep)),
                 "pack_detections: null pointer");
  tvmi::pack_detections_kernel<<<dim3((unsigned)num_images), dim3(tvmi::kPackThreads), 0, static_cast<hipStream_t>(stream)>>>(
      boxes, scores, labels, image_idx, keep, keep_capacity, num_keep_dev, (int)num_images, (int)max_dets, dets, counts,
      max_dets * 6, 0);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_pack_detections_devcount");
}

extern "C" int tvmi_pack_detections_payload(const float* boxes, const float* scores, const int64_t* labels,
                                            const int64_t* image_idx, const int64_t* keep, int64_t keep_capacity,
                                            const int64_t* num_keep_dev, int64_t num_images, int64_t max_dets,
                                            float* payload, int64_t row_stride, int32_t* counts, void* stream) {
  TVMI_CHECK_ARG(num_images >= 0 && max_dets >= 0 && keep_capacity >= 0, "pack_detections: negative size");
  if (num_images == 0) return 0;
  TVMI_CHECK_ARG(num_images <= 65535, "pack_detections: at most 65535 images per call");
  TVMI_CHECK_ARG(row_stride >= max_dets * 6 + 1, "pack_detections_payload: a row holds max_dets * 6 floats and the count");
  TVMI_CHECK_ARG(payload && num_keep_dev && (keep_capacity == 0 || (boxes && scores && image_idx && keep)),
                 "pack_detections: null pointer");
  tvmi::pack_detections_kernel<<<dim3((unsigned)num_images), dim3(tvmi::kPackThreads), 0, static_cast<hipStream_t>(stream)>>>(
      boxes, scores, labels, image_idx, keep, keep_capacity, num_keep_dev, (int)num_images, (int)max_dets, payload, counts,
      row_stride, 1);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_pack_detections_payload");
}

extern "C" int tvmi_detection_candidates(const float* class_logits, const float* box_regression, const float* proposals,
                                         const int32_t* row_image, const float* image_hw, int64_t R, int64_t C,
                                         int64_t B, float wx, float wy, float ww, float wh, float bbox_xform_clip,
                                         float score_thresh, float min_size, float* cand_boxes, float* cand_scores,
                                         uint8_t* cand_valid, void* stream) {
  TVMI_CHECK_ARG(R >= 0 && C >= 1 && B >= 1, "detection_candidates: bad sizes");
  if (R == 0 || C == 1) return 0;
  TVMI_CHECK_ARG(class_logits && box_regression && proposals && row_image && image_hw && cand_boxes && cand_scores && cand_valid,
                 "detection_candidates: null pointer");
  TVMI_CHECK_ARG(R * C < (1ll << 31), "detection_candidates: size too large");
  tvmi::det_candidates_kernel<<<dim3((unsigned)((R + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(
      class_logits, box_regression, proposals, row_image, image_hw, (int)R, (int)C, (int)B, wx, wy, ww, wh, bbox_xform_clip,
      score_thresh, min_size, cand_boxes, cand_scores, cand_valid);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_detection_candidates");
}

extern "C" int tvmi_rpn_candidates(const float* objectness, const float* boxes_in, const float* deltas,
                                   const int64_t* top_idx, const int64_t* level_offsets, const float* image_hw, int64_t B,
                                   int64_t A, int64_t T, int64_t L, float bbox_xform_clip, float score_thresh,
                                   float min_size, float* out_boxes, float* out_scores, int64_t* out_levels,
                                   uint8_t* out_valid, void* stream) {
  TVMI_CHECK_ARG(B >= 0 && A >= 0 && T >= 0 && L >= 1, "rpn_candidates: bad sizes");
  if (B * T == 0) return 0;
  TVMI_CHECK_ARG(A > 0 && objectness && boxes_in && top_idx && level_offsets && image_hw && out_boxes && out_scores &&
                     out_levels && out_valid,
                 "rpn_candidates: null pointer");
  TVMI_CHECK_ARG(B * T < (1ll << 31) && B <= (1 << 20), "rpn_candidates: size too large");
  tvmi::rpn_candidates_kernel<<<dim3((unsigned)((B * T + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(
      objectness, boxes_in, deltas, top_idx, level_offsets, image_hw, (int)B, A, (int)T, (int)L, bbox_xform_clip,
      score_thresh, min_size, out_boxes, out_scores, out_levels, out_valid);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_rpn_candidates");
}

extern "C" int tvmi_boxes_to_rois(const void* const* boxes, const int64_t* counts, int64_t num_images, void* rois,
                                  tvmi_dtype dt, void* stream) {
  TVMI_CHECK_ARG(num_images >= 0 && num_images <= tvmi::kRoiFmtMaxImages, "boxes_to_rois: at most 64 images per call");
  if (num_images == 0) return 0;
  TVMI_CHECK_ARG(boxes && counts && rois, "boxes_to_rois: null pointer");
  tvmi::BoxLists bl;
  int64_t run = 0;
  for (int i = 0; i < tvmi::kRoiFmtMaxImages; ++i) {
    if (i < num_images) {
      TVMI_CHECK_ARG(counts[i] >= 0 && (counts[i] == 0 || boxes[i]), "boxes_to_rois: bad box list");
      run += counts[i];
    }
    bl.ptr[i] = i < num_images ? boxes[i] : nullptr;
    TVMI_CHECK_ARG(run < (1ll << 31), "boxes_to_rois: too many boxes");
    bl.end[i] = (int)run;
  }
  bl.n = (int)num_images;
  if (run == 0) return 0;
  const dim3 grid((unsigned)((run + 255) / 256));
  TVMI_DISPATCH_FLOAT(dt, "boxes_to_rois",
                      tvmi::boxes_to_rois_kernel<scalar_t><<<grid, dim3(256), 0, static_cast<hipStream_t>(stream)>>>(
                          bl, static_cast<scalar_t*>(rois), (int)run));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_boxes_to_rois");
}

extern "C" int tvmi_box_iou_pairwise(const void* boxes1, const void* boxes2, void* out, tvmi_dtype dt, int64_t N, int64_t M,
                                     int mode, int source_16bit, double eps, void* stream) {
  TVMI_CHECK_ARG(N >= 0 && M >= 0, "box_iou_pairwise: negative size");
  if (N * M == 0) return 0;
  TVMI_CHECK_ARG(boxes1 && boxes2 && out, "box_iou_pairwise: null pointer");
  TVMI_CHECK_ARG(dt == TVMI_F32 || dt == TVMI_F64, "box_iou_pairwise: float32 / float64 (upcast 16-bit boxes first)");
  TVMI_CHECK_ARG(mode >= 0 && mode <= 3, "box_iou_pairwise: mode 0 IoU, 1 generalized, 2 distance, 3 complete");
  TVMI_CHECK_ARG(N <= 65535 * 64ll && M < (1ll << 31), "box_iou_pairwise: size too large");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((M + 255) / 256), (unsigned)std::min<int64_t>(N, 65535));
  // rows beyond the y-limit of the grid are covered by repeated launches on row slices
  for (int64_t r0 = 0; r0 < N; r0 += 65535) {
    const int rows = (int)std::min<int64_t>(65535, N - r0);
    const dim3 g((unsigned)((M + 255) / 256), (unsigned)rows);
#define TVMI_IOU(T_, MODE_)                                                                                         \
  tvmi::box_iou_pairwise_kernel<T_, MODE_><<<g, dim3(256), 0, s>>>(static_cast<const T_*>(boxes1) + r0 * 4,          \
                                                                   static_cast<const T_*>(boxes2),                   \
                                                                   static_cast<T_*>(out) + r0 * M, rows, (int)M, source_16bit, (T_)eps)
#define TVMI_IOU_MODES(T_)                      \
  switch (mode) {                               \
    case 0: TVMI_IOU(T_, 0); break;             \
    case 1: TVMI_IOU(T_, 1); break;             \
    case 2: TVMI_IOU(T_, 2); break;             \
    default: TVMI_IOU(T_, 3); break;            \
  }
    if (dt == TVMI_F32) {
      TVMI_IOU_MODES(float)
    } else {
      TVMI_IOU_MODES(double)
    }
#undef TVMI_IOU_MODES
#undef TVMI_IOU
  }
  (void)grid;
  TVMI_RETURN_LAUNCH_STATUS("tvmi_box_iou_pairwise");
}
