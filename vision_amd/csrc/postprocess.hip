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
constexpr int kPackMaxImages = 256;

__global__ __launch_bounds__(kPackThreads) void pack_detections_kernel(
    const float* __restrict__ boxes, const float* __restrict__ scores, const int64_t* __restrict__ labels,
    const int64_t* __restrict__ image_idx, const int64_t* __restrict__ keep, int64_t num_keep, int num_images,
    int max_dets, float* __restrict__ dets, int32_t* __restrict__ counts) {
  __shared__ int s_cnt[kPackMaxImages];                 // kept so far per image
  __shared__ int s_wave[kPackWaves][kPackMaxImages];    // this chunk: kept per (wave, image)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int64_t total = (int64_t)num_images * max_dets * 6;
  for (int64_t i = tid; i < total; i += kPackThreads) dets[i] = 0.f;
  for (int i = tid; i < num_images; i += kPackThreads) s_cnt[i] = 0;
  for (int i = tid; i < kPackWaves * kPackMaxImages; i += kPackThreads) (&s_wave[0][0])[i] = 0;
  __syncthreads();
  for (int64_t c0 = 0; c0 < num_keep; c0 += kPackThreads) {
    const int64_t e = c0 + tid;
    const bool valid = e < num_keep;
    int64_t src = 0;
    int img = -1;
    if (valid) {
      src = keep[e];
      img = (int)image_idx[src];
      if (img < 0 || img >= num_images) img = -1;
    }
    // rank among equal image ids inside the wave: peel off one distinct id per iteration
    int rank_in_wave = 0;
    unsigned long long todo = __ballot(img >= 0);
    while (todo) {
      const int leader = __builtin_ctzll(todo);
      const int cur = __builtin_amdgcn_readlane(img, leader);
      const unsigned long long same = __ballot(img == cur);
      if (img == cur) rank_in_wave = __popcll(same & ((1ull << lane) - 1ull));
      if (lane == leader) s_wave[wave][cur] = __popcll(same);
      todo &= ~same;
    }
    __syncthreads();
    if (img >= 0) {
      int base = s_cnt[img];
      for (int w = 0; w < wave; ++w) base += s_wave[w][img];
      const int r = base + rank_in_wave;
      if (r < max_dets) {
        float* d = dets + ((int64_t)img * max_dets + r) * 6;
        const float4 b = *reinterpret_cast<const float4*>(boxes + src * 4);
        d[0] = b.x;
        d[1] = b.y;
        d[2] = b.z;
        d[3] = b.w;
        d[4] = scores[src];
        d[5] = labels ? (float)labels[src] : 0.f;
      }
    }
    __syncthreads();
    for (int i = tid; i < num_images; i += kPackThreads) {
      int add = 0;
      for (int w = 0; w < kPackWaves; ++w) {
        add += s_wave[w][i];
        s_wave[w][i] = 0;
      }
      s_cnt[i] += add;
    }
    __syncthreads();
  }
  for (int i = tid; i < num_images; i += kPackThreads) counts[i] = min(s_cnt[i], max_dets);
}

}  // namespace
}  // namespace tvmi

extern "C" int tvmi_pack_detections(const float* boxes, const float* scores, const int64_t* labels,
                                    const int64_t* image_idx, const int64_t* keep, int64_t num_keep, int64_t num_images,
                                    int64_t max_dets, float* dets, int32_t* counts, void* stream) {
  TVMI_CHECK_ARG(num_images >= 0 && max_dets >= 0 && num_keep >= 0, "pack_detections: negative size");
  if (num_images == 0) return 0;
  TVMI_CHECK_ARG(num_images <= tvmi::kPackMaxImages, "pack_detections: at most 256 images per call");
  TVMI_CHECK_ARG(dets && counts && (num_keep == 0 || (boxes && scores && image_idx && keep)),
                 "pack_detections: null pointer");
  tvmi::pack_detections_kernel<<<dim3(1), dim3(tvmi::kPackThreads), 0, static_cast<hipStream_t>(stream)>>>(
      boxes, scores, labels, image_idx, keep, num_keep, (int)num_images, (int)max_dets, dets, counts);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_pack_detections");
}
