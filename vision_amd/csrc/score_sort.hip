// score_sort.hip — the processing order of large NMS problems: indices of
// aten::sort(scores, stable=True, descending=True) for float32 scores, any n (the reference sorts with
// `scores.sort(0, descending=True)`, cuda/nms_kernel.cu:186-187; NaN first, ties by ascending index, -0 == +0).
//
// Up to 4096 scores nms.hip's single-workgroup bitonic kernel is the faster form.  Above that torch's sort is a radix
// block sort plus seven merge passes over (float, int64) pairs, an index arange and two copies — 0.10 ms for 100k
// scores, 7 % of the whole NMS.  Here: one pass turns every score into a 32-bit key whose ASCENDING unsigned order is
// the wanted order (the NaN / signed-zero rules live in that key, nowhere else), rocPRIM's radix sort orders
// (key, 32-bit index) pairs — a plain library sort, stable by construction — and one pass widens the indices.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "tvmi_common.h"

namespace tvmi {
namespace {

__device__ __forceinline__ unsigned order_key(float f) {  // ascending key == descending score (same rule as nms.hip)
  if (f != f) return 0u;  // NaN is the greatest value for aten::sort
  unsigned b = __builtin_bit_cast(unsigned, f);
  if (f == 0.f) b = 0u;  // -0 == +0
  const unsigned asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  const unsigned d = ~asc;
  return d == 0u ? 1u : d;  // cannot happen for non-NaN values (asc of +inf is 0xFF800000), kept for safety
}

__global__ __launch_bounds__(256) void score_keys(const float* __restrict__ scores, int n, unsigned* __restrict__ keys,
                                                  unsigned* __restrict__ index) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    keys[i] = order_key(scores[i]);
    index[i] = (unsigned)i;
  }
}
__global__ __launch_bounds__(256) void widen_index(const unsigned* __restrict__ index, int n, int64_t* __restrict__ order) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) order[i] = (int64_t)index[i];
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// (rocPRIM sorts up to a million items with a block sort + log2(n / 1024) merge passes: 7 launches of ~5.6 us at 100k items, 50 us a
// sort.  Forcing Onesweep — radix_sort_config<..., MergeSortLimit = 0> — was measured in round 5: 8 iteration launches of 22 us
// and 20 buffer fills per call, 430 us instead of 235 for batched_nms at 100k x 80; profiles/r05_bnms_*.)

// ---- stable partition of the score order by segment id (the second sort of a batched NMS).  Round 4 used aten's stable
// sort of the int64 ids (a merge sort: 87 us of the 264 us of batched_nms at 100k boxes x 80 classes); a stable LSD radix
// sort over just the id bits, applied to the sequence that is already in score order, is the same permutation.
__global__ __launch_bounds__(256) void partition_keys(const int64_t* __restrict__ order, const int64_t* __restrict__ seg, int n,
                                                      const int64_t* __restrict__ n_dev, unsigned dead_key, unsigned* __restrict__ keys,
                                                      unsigned* __restrict__ ranks, int* __restrict__ flag) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= n) return;
  const bool live = n_dev == nullptr || (int64_t)g < *n_dev;
  unsigned k = dead_key;   // behind every live key (the device-count forms never look at these positions)
  if (live) {
    const int64_t sg = seg[order[g]];
    if (sg < 0 || sg >= (int64_t)dead_key) atomicOr(flag, 1);   // outside the promised range: the caller takes the general path
    else k = (unsigned)sg;
  }
  keys[g] = k;
  ranks[g] = (unsigned)g;
}
__global__ __launch_bounds__(256) void partition_widen(const unsigned* __restrict__ keys, const unsigned* __restrict__ ranks, int n,
                                                       int64_t* __restrict__ keys_out, int64_t* __restrict__ perm_out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < n) {
    keys_out[p] = (int64_t)keys[p];
    perm_out[p] = (int64_t)ranks[p];
  }
}
inline int partition_bits(int64_t num_segments) {   // live ids < 2^bits; the dead key is 2^bits
  if (num_segments <= 0 || num_segments > (1ll << 30)) return 31;
  int b = 1;
  while ((1ll << b) < num_segments) ++b;
  return b;
}
inline hipError_t partition_temp_bytes(int64_t n, size_t* bytes) {
  *bytes = 0;
  return rocprim::radix_sort_pairs(nullptr, *bytes, static_cast<unsigned*>(nullptr), static_cast<unsigned*>(nullptr),
                                                  static_cast<unsigned*>(nullptr), static_cast<unsigned*>(nullptr), (size_t)n, 0, 32,
                                                  static_cast<hipStream_t>(nullptr), false);
}

inline hipError_t radix_temp_bytes(int64_t n, size_t* bytes) {
  *bytes = 0;
  return rocprim::radix_sort_pairs(nullptr, *bytes, static_cast<unsigned*>(nullptr), static_cast<unsigned*>(nullptr),
                                                  static_cast<unsigned*>(nullptr), static_cast<unsigned*>(nullptr), (size_t)n, 0, 32,
                                                  static_cast<hipStream_t>(nullptr), false);
}

}  // namespace
}  // namespace tvmi

extern "C" size_t tvmi_sort_scores_desc_workspace_bytes(int64_t n) {
  if (n <= 0 || n >= (1ll << 31)) return 0;
  size_t temp = 0;
  if (tvmi::radix_temp_bytes(n, &temp) != hipSuccess) return 0;
  return 4 * tvmi::align256((size_t)n * sizeof(unsigned)) + tvmi::align256(temp);
}

extern "C" int tvmi_sort_scores_desc_large(const float* scores, int64_t n, int64_t* order, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  TVMI_CHECK_ARG(n >= 0 && n < (1ll << 31), "sort_scores_desc_large: 0 <= n < 2^31");
  if (n == 0) return 0;
  TVMI_CHECK_ARG(scores && order && workspace, "sort_scores_desc_large: null pointer");
  size_t temp = 0;
  hipError_t e = tvmi::radix_temp_bytes(n, &temp);
  if (e != hipSuccess) return tvmi::set_error((int)e, "tvmi_sort_scores_desc_large: temporary storage query");
  const size_t col = tvmi::align256((size_t)n * sizeof(unsigned));
  TVMI_CHECK_ARG(workspace_bytes >= 4 * col + tvmi::align256(temp), "sort_scores_desc_large: workspace too small");
  TVMI_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "sort_scores_desc_large: workspace must be 256-byte aligned");
  char* base = static_cast<char*>(workspace);
  unsigned* keys_in = reinterpret_cast<unsigned*>(base);
  unsigned* keys_out = reinterpret_cast<unsigned*>(base + col);
  unsigned* idx_in = reinterpret_cast<unsigned*>(base + 2 * col);
  unsigned* idx_out = reinterpret_cast<unsigned*>(base + 3 * col);
  void* temp_storage = base + 4 * col;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)tvmi::ceil_div(n, (int64_t)256));
  tvmi::score_keys<<<grid, dim3(256), 0, s>>>(scores, (int)n, keys_in, idx_in);
  e = rocprim::radix_sort_pairs(temp_storage, temp, keys_in, keys_out, idx_in, idx_out, (size_t)n, 0, 32, s, false);
  if (e != hipSuccess) return tvmi::set_error((int)e, "tvmi_sort_scores_desc_large: radix sort");
  tvmi::widen_index<<<grid, dim3(256), 0, s>>>(idx_out, (int)n, order);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_sort_scores_desc_large");
}

extern "C" size_t tvmi_partition_by_segment_workspace_bytes(int64_t n) {
  if (n <= 0 || n >= (1ll << 31)) return 0;
  size_t temp = 0;
  if (tvmi::partition_temp_bytes(n, &temp) != hipSuccess) return 0;
  return 4 * tvmi::align256((size_t)n * sizeof(unsigned)) + tvmi::align256(temp) + 256;
}

extern "C" int tvmi_partition_by_segment(const int64_t* order, const int64_t* seg, int64_t n, const int64_t* n_dev,
                                         int64_t num_segments, int64_t* keys_out, int64_t* perm_out, int* flag_out, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  TVMI_CHECK_ARG(n >= 0 && n < (1ll << 31), "partition_by_segment: 0 <= n < 2^31");
  TVMI_CHECK_ARG(flag_out != nullptr, "partition_by_segment: flag_out is null");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(flag_out, 0, sizeof(int), s);
  if (e != hipSuccess) return tvmi::set_error((int)e, "tvmi_partition_by_segment: memset");
  if (n == 0) return 0;
  TVMI_CHECK_ARG(order && seg && keys_out && perm_out && workspace, "partition_by_segment: null pointer");
  size_t temp = 0;
  e = tvmi::partition_temp_bytes(n, &temp);
  if (e != hipSuccess) return tvmi::set_error((int)e, "tvmi_partition_by_segment: temporary storage query");
  const size_t col = tvmi::align256((size_t)n * sizeof(unsigned));
  TVMI_CHECK_ARG(workspace_bytes >= 4 * col + tvmi::align256(temp), "partition_by_segment: workspace too small");
  TVMI_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "partition_by_segment: workspace must be 256-byte aligned");
  char* base = static_cast<char*>(workspace);
  unsigned* keys_in = reinterpret_cast<unsigned*>(base);
  unsigned* keys_sorted = reinterpret_cast<unsigned*>(base + col);
  unsigned* rank_in = reinterpret_cast<unsigned*>(base + 2 * col);
  unsigned* rank_sorted = reinterpret_cast<unsigned*>(base + 3 * col);
  void* temp_storage = base + 4 * col;
  const int bits = tvmi::partition_bits(num_segments);
  const dim3 grid((unsigned)tvmi::ceil_div(n, (int64_t)256));
  tvmi::partition_keys<<<grid, dim3(256), 0, s>>>(order, seg, (int)n, n_dev, 1u << bits, keys_in, rank_in, flag_out);
  e = rocprim::radix_sort_pairs(temp_storage, temp, keys_in, keys_sorted, rank_in, rank_sorted, (size_t)n, 0,
                                                     (unsigned)bits + 1, s, false);
  if (e != hipSuccess) return tvmi::set_error((int)e, "tvmi_partition_by_segment: radix sort");
  tvmi::partition_widen<<<grid, dim3(256), 0, s>>>(keys_sorted, rank_sorted, (int)n, keys_out, perm_out);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_partition_by_segment");
}
