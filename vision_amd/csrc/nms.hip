// nms.hip — greedy IoU non-maximum suppression for gfx950 (MI355X), wave64 bitmask form.
//
// Semantics and arithmetic: torchvision/csrc/ops/cpu/nms_kernel.cpp:17-95.  The result is
// the same index list, bit for bit, as the reference CPU kernel:
//   * areas[k] = (x2-x1)*(y2-y1); inter = max(0,xx2-xx1)*max(0,yy2-yy1);
//     ovr = inter / ((iarea + areas[j]) - inter), every operation individually rounded
//     (this TU is compiled with -ffp-contract=off; `/` is the IEEE-correct division);
//   * the test is `(double)ovr > iou_threshold` exactly as cpu/nms_kernel.cpp:88 promotes
//     it (the reference CUDA kernel compares against a float-rounded threshold instead,
//     cuda/nms_kernel.cu:45 — that is NOT what we reproduce);
//   * 0/0 = NaN compares false, so degenerate boxes neither suppress nor get suppressed.
//
// Structure (differs from cuda/nms_kernel.cu:56-148):
//   K1 nms_mask_kernel : one wave per (64-row block) x (8 x 64-column blocks).  Lane = column
//      box (registers); the 64 row boxes sit in LDS and are broadcast.  For every row the
//      64-lane predicate is turned into the 64-bit suppression word by the compare itself
//      (__ballot, an SGPR pair) and parked in lane `row`; after 8 column blocks each lane
//      owns one full 64-byte line of the row-major mask and stores it with 4 dwordx4.
//      Boxes are gathered through `order` in-kernel (no index_select pass), areas are
//      recomputed from the box (same value the reference caches).
//   K2 nms_sweep_kernel: ONE 1024-thread workgroup walks the 64-box blocks in order.  Wave 0
//      resolves the diagonal 64x64 tile with a register-only scalar loop (v_readlane), the
//      whole workgroup then ORs the kept rows into the LDS-resident `removed` bit-vector with
//      coalesced row reads, and wave 0 appends the kept original indices (in score order)
//      and finally the count — no N-step single-thread loop, no masked_select pass.
#include <algorithm>
#include <type_traits>

#include "tvmi_common.h"

namespace tvmi {
namespace {

constexpr int kColGroup = 8;  // column blocks per wave task (8 x 8 B = one 64-B mask line)
constexpr int kMaskWaves = 4; // waves per workgroup in K1
constexpr int kSweepThreads = 1024;

template <typename T>
struct Box {
  T x1, y1, x2, y2;
};

template <typename T>
__device__ __forceinline__ Box<T> load_box(const T* dets, int64_t i) {
  Box<T> b;
  if constexpr (std::is_same<T, float>::value) {
    const float4 v = *reinterpret_cast<const float4*>(dets + i * 4);
    b.x1 = v.x;
    b.y1 = v.y;
    b.x2 = v.z;
    b.y2 = v.w;
  } else {
    b.x1 = dets[i * 4 + 0];
    b.y1 = dets[i * 4 + 1];
    b.x2 = dets[i * 4 + 2];
    b.y2 = dets[i * 4 + 3];
  }
  return b;
}

// mask layout: row-major [n][cb_pad] u64, cb_pad = col_blocks rounded up to kColGroup.
template <typename T>
__global__ __launch_bounds__(kMaskWaves * kWave) void nms_mask_kernel(
    const T* __restrict__ dets, const int64_t* __restrict__ order, const int64_t* __restrict__ seg,
    int n, int col_blocks, int cb_pad, double thr, unsigned long long* __restrict__ mask) {
  __shared__ T s_row[kMaskWaves][64][5];       // x1,y1,x2,y2,area of the wave's row block
  __shared__ long long s_seg[kMaskWaves][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row_blk = blockIdx.y;
  const int col_grp = blockIdx.x * kMaskWaves + wave;
  const int cb0 = col_grp * kColGroup;
  if (cb0 >= cb_pad) return;
  const int row0 = row_blk * 64;
  const int my_row = row0 + lane;
  unsigned long long words[kColGroup];
#pragma unroll
  for (int q = 0; q < kColGroup; ++q) words[q] = 0ull;

  // Column groups entirely left of the diagonal hold no (j > i) pair
  // (the sweep never reads them, so they are not even written)
  const bool active = (cb0 + kColGroup - 1 >= row_blk) && (cb0 < col_blocks);
  if (!active) return;
  {
    // stage the 64 row boxes (wave-private LDS region; no workgroup barrier needed)
    {
      T x1 = 0, y1 = 0, x2 = 0, y2 = 0;
      long long sg = 0;
      if (my_row < n) {
        const int64_t oi = order[my_row];
        const Box<T> b = load_box<T>(dets, oi);
        x1 = b.x1;
        y1 = b.y1;
        x2 = b.x2;
        y2 = b.y2;
        if (seg) sg = seg[oi];
      }
      s_row[wave][lane][0] = x1;
      s_row[wave][lane][1] = y1;
      s_row[wave][lane][2] = x2;
      s_row[wave][lane][3] = y2;
      s_row[wave][lane][4] = (x2 - x1) * (y2 - y1);
      s_seg[wave][lane] = sg;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int rows_here = min(64, n - row0);
#pragma unroll
    for (int q = 0; q < kColGroup; ++q) {
      const int cb = cb0 + q;
      if (cb < row_blk || cb >= col_blocks) continue;  // wave-uniform
      const int j = cb * 64 + lane;
      T jx1 = 0, jy1 = 0, jx2 = 0, jy2 = 0;
      long long jseg = 0;
      const bool jvalid = j < n;
      if (jvalid) {
        const int64_t oj = order[j];
        const Box<T> b = load_box<T>(dets, oj);
        jx1 = b.x1;
        jy1 = b.y1;
        jx2 = b.x2;
        jy2 = b.y2;
        if (seg) jseg = seg[oj];
      }
      const T jarea = (jx2 - jx1) * (jy2 - jy1);
      const bool diag = cb == row_blk;
      unsigned long long mine = 0ull;
      for (int i = 0; i < rows_here; ++i) {
        const T ix1 = s_row[wave][i][0], iy1 = s_row[wave][i][1];
        const T ix2 = s_row[wave][i][2], iy2 = s_row[wave][i][3];
        const T iarea = s_row[wave][i][4];
        const T xx1 = ix1 > jx1 ? ix1 : jx1;  // std::max(ix1, x1[j])
        const T yy1 = iy1 > jy1 ? iy1 : jy1;
        const T xx2 = jx2 < ix2 ? jx2 : ix2;  // std::min(ix2, x2[j])
        const T yy2 = jy2 < iy2 ? jy2 : iy2;
        const T dw = xx2 - xx1, dh = yy2 - yy1;
        const T w = (T)0 < dw ? dw : (T)0;    // std::max(0, xx2-xx1)
        const T h = (T)0 < dh ? dh : (T)0;
        const T inter = w * h;
        const T ovr = inter / (iarea + jarea - inter);
        bool p = ((double)ovr > thr) && jvalid;
        if (diag) p = p && (lane > i);
        if (seg) p = p && (jseg == s_seg[wave][i]);
        const unsigned long long word = __ballot(p);
        if (lane == i) mine = word;
      }
      words[q] = mine;
    }
  }
  if (my_row < n) {
    unsigned long long* dst = mask + (size_t)my_row * cb_pad + cb0;
    ulonglong2* d2 = reinterpret_cast<ulonglong2*>(dst);
#pragma unroll
    for (int q = 0; q < kColGroup / 2; ++q) d2[q] = make_ulonglong2(words[2 * q], words[2 * q + 1]);
  }
}

__global__ __launch_bounds__(kSweepThreads) void nms_sweep_kernel(
    const unsigned long long* __restrict__ mask, const int64_t* __restrict__ order, int n,
    int col_blocks, int cb_pad, int64_t* __restrict__ keep_out, int64_t* __restrict__ num_keep) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_removed[];  // [col_blocks + 2]
  unsigned long long* s_keepbits = s_removed + col_blocks;                        // [1]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  for (int c = tid; c < col_blocks; c += kSweepThreads) s_removed[c] = 0ull;
  __syncthreads();
  int64_t count = 0;  // meaningful in wave 0 only
  for (int b = 0; b < col_blocks; ++b) {
    const int row0 = b * 64;
    if (tid < 64) {
      const int row = row0 + lane;
      unsigned long long diag = 0ull;
      if (row < n) diag = mask[(size_t)row * cb_pad + b];
      const unsigned long long rem_v = s_removed[b];
      // wave-uniform: keep the running word in SGPRs so the 64-step resolve is scalar code
      unsigned long long rem =
          ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(rem_v >> 32)) << 32) |
          (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)rem_v);
      const int rows_here = min(64, n - row0);
      if (rows_here < 64) rem |= ~0ull << rows_here;  // rows past n are "removed"
      unsigned long long keep = 0ull;
      const unsigned int dlo = (unsigned int)diag, dhi = (unsigned int)(diag >> 32);
#pragma unroll
      for (int kbit = 0; kbit < 64; ++kbit) {
        const unsigned long long rk =
            ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dhi, kbit) << 32) |
            (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dlo, kbit);
        if (!((rem >> kbit) & 1ull)) {  // wave-uniform
          keep |= 1ull << kbit;
          rem |= rk;
        }
      }
      if (lane == 0) *s_keepbits = keep;
      // append kept original indices in order
      const bool kept = (keep >> lane) & 1ull;
      if (kept) {
        const unsigned long long below = keep & ((1ull << lane) - 1ull);
        keep_out[count + __popcll(below)] = order[row0 + lane];
      }
      count += __popcll(keep);
    }
    __syncthreads();
    const unsigned long long keep = *s_keepbits;
    // OR the kept rows of this block into removed[] for all later column blocks.
    for (int c = b + 1 + tid; c < col_blocks; c += kSweepThreads) {
      unsigned long long acc = 0ull;
      unsigned long long kb = keep;
      while (kb) {
        const int kbit = __ffsll((long long)kb) - 1;
        kb &= kb - 1ull;
        acc |= mask[(size_t)(row0 + kbit) * cb_pad + c];
      }
      if (acc) s_removed[c] |= acc;
    }
    __syncthreads();
  }
  if (tid == 0) *num_keep = count;
}

template <typename T>
int launch(const void* dets, const int64_t* order, const int64_t* seg, int64_t n, double thr,
           void* workspace, int64_t* keep_out, int64_t* num_keep, hipStream_t stream) {
  const int col_blocks = (int)ceil_div(n, 64);
  const int cb_pad = (int)(ceil_div(col_blocks, kColGroup) * kColGroup);
  unsigned long long* mask = static_cast<unsigned long long*>(workspace);
  const int col_groups = cb_pad / kColGroup;
  const dim3 grid((unsigned)ceil_div(col_groups, kMaskWaves), (unsigned)col_blocks);
  nms_mask_kernel<T><<<grid, dim3(kMaskWaves * kWave), 0, stream>>>(
      static_cast<const T*>(dets), order, seg, (int)n, col_blocks, cb_pad, thr, mask);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, "tvmi_nms: mask kernel launch");
  const size_t lds = (size_t)(col_blocks + 2) * sizeof(unsigned long long);
  nms_sweep_kernel<<<dim3(1), dim3(kSweepThreads), lds, stream>>>(mask, order, (int)n, col_blocks,
                                                                  cb_pad, keep_out, num_keep);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_nms: sweep kernel launch");
}

}  // namespace
}  // namespace tvmi

extern "C" size_t tvmi_nms_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  const int64_t col_blocks = tvmi::ceil_div(n, 64);
  const int64_t cb_pad = tvmi::ceil_div(col_blocks, tvmi::kColGroup) * tvmi::kColGroup;
  return (size_t)n * (size_t)cb_pad * sizeof(unsigned long long);
}

extern "C" int tvmi_nms(const void* dets, const int64_t* order, const int64_t* seg, int64_t n,
                        double iou_threshold, tvmi_dtype dt, void* workspace,
                        size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out,
                        void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_CHECK_ARG(n >= 0, "nms: negative box count");
  TVMI_CHECK_ARG(num_keep_out != nullptr, "nms: num_keep_out is null");
  if (n == 0) {
    hipError_t e = hipMemsetAsync(num_keep_out, 0, sizeof(int64_t), s);
    return e == hipSuccess ? 0 : tvmi::set_error((int)e, "tvmi_nms: memset");
  }
  TVMI_CHECK_ARG(dets && order && keep_out && workspace, "nms: null pointer");
  TVMI_CHECK_ARG(n <= 1200000, "nms: more than 1.2M boxes is not supported by the bitmask path");
  TVMI_CHECK_ARG(workspace_bytes >= tvmi_nms_workspace_bytes(n), "nms: workspace too small");
  TVMI_CHECK_ARG(dt == TVMI_F32 || dt == TVMI_F64, "nms: dets must be float32 or float64");
  if (dt == TVMI_F32)
    return tvmi::launch<float>(dets, order, seg, n, iou_threshold, workspace, keep_out, num_keep_out, s);
  return tvmi::launch<double>(dets, order, seg, n, iou_threshold, workspace, keep_out, num_keep_out, s);
}
