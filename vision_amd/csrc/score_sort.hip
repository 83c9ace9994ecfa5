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
