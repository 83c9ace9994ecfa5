// roi_common.h — RoI geometry shared by the RoIAlign forward (roi_align.hip) and backward
// (roi_align_bwd.hip) translation units.  Arithmetic follows the reference's CPU kernels op by op
// (torchvision/csrc/ops/cpu/roi_align_kernel.cpp:36-66, cpu/roi_align_common.h:50-103); both TUs are
// built with -ffp-contract=off so that sample coordinates round like the reference's x86 build.
#pragma once

#include "tvmi_common.h"

namespace tvmi {

constexpr int kMaxLevels = 8;  // FPN levels served by one multi-scale launch

template <typename A>
struct RoiGeom {
  A start_h, start_w, bin_h, bin_w, count;
  int gh, gw, batch;
};

// cpu/roi_align_kernel.cpp:36-66
template <typename T, typename A>
__device__ __forceinline__ RoiGeom<A> roi_geom(const T* roi, A scale, int PH, int PW, int sr, bool aligned) {
  RoiGeom<A> g;
  g.batch = (int)ld(roi);
  const A offset = aligned ? (A)0.5 : (A)0.0;
  const A sw = ld(roi + 1) * scale - offset;
  const A sh = ld(roi + 2) * scale - offset;
  const A ew = ld(roi + 3) * scale - offset;
  const A eh = ld(roi + 4) * scale - offset;
  A rw = ew - sw;
  A rh = eh - sh;
  if (!aligned) {
    rw = rw > (A)1. ? rw : (A)1.;  // std::max(roi_width, 1)
    rh = rh > (A)1. ? rh : (A)1.;
  }
  g.start_h = sh;
  g.start_w = sw;
  g.bin_h = rh / (A)PH;
  g.bin_w = rw / (A)PW;
  g.gh = sr > 0 ? sr : (int)ceil(rh / (A)PH);
  g.gw = sr > 0 ? sr : (int)ceil(rw / (A)PW);
  const int cnt = g.gh * g.gw;
  g.count = (A)(cnt > 1 ? cnt : 1);
  return g;
}

// One axis of cpu/roi_align_common.h:50-103.  Returns false when the coordinate is
// outside [-1, dim] (the sample then contributes zero).
template <typename A>
__device__ __forceinline__ bool axis_sample(int dim, A start, A bin, int grid, int p, int i, int& lo, int& hi, A& l,
                                            A& h) {
  A c = start + (A)p * bin + (A)((float)i + .5f) * bin / (A)grid;
  if (c < (A)-1.0 || c > (A)dim) {
    lo = hi = 0;
    l = h = (A)0;
    return false;
  }
  if (c <= (A)0) c = (A)0;
  lo = (int)c;
  if (lo >= dim - 1) {
    hi = lo = dim - 1;
    c = (A)lo;
  } else {
    hi = lo + 1;
  }
  l = c - (A)lo;
  h = (A)1. - l;
  return true;
}

// lo/l/h of one axis sample in "shifted" form (dim >= 2): a sample on the last row / column (where the
// reference uses x_high = x_low with weight 0) is re-expressed on the pair (dim-2, dim-1) with factors
// (0, 1) — the same value, and lo + 1 is always inside the map.
__device__ __forceinline__ bool axis_sample_shifted(int dim, float start, float bin, int grid, int p, int i, int& lo,
                                                    float& l, float& h) {
  // branch-free on purpose: written with early returns / conditional stores through the references, the compiler kept
  // {l, h} in a SCRATCH pair addressed by a run-time offset, and the scratch loads' vmcnt dependency then forced a
  // `s_waitcnt vmcnt(0)` into the DMA kernels' double-buffered channel loop (guarded by tests/test_isa_guards.py)
  int lo_, hi_;
  float l_, h_;
  const bool v = axis_sample<float>(dim, start, bin, grid, p, i, lo_, hi_, l_, h_);
  const bool edge = lo_ > dim - 2;  // lo == dim-1: value is in[dim-1]
  lo = v ? (edge ? dim - 2 : lo_) : 0;
  l = v ? (edge ? 1.f : l_) : 0.f;
  h = v ? (edge ? 0.f : h_) : 0.f;
  return v;
}

// v_mul_legacy_f32 (0 * x = 0 for every x, NaN and Inf included) as the LLVM intrinsic itself — clang 22 has no builtin
// for it; an inline-asm form would pin the surrounding LDS reads in place.  The forward kernels use it for the ONE product
// of a tap pair whose factor can be an exact zero on a pixel the reference never reads (the x edge re-expressed on the
// pair (W-2, W-1) with factors (0, 1)): a NaN / Inf there must not reach the output (cpu/roi_align_common.h:78-90 reads
// pixel W-1 twice instead).
extern "C" __device__ float tvmi_fmul_legacy(float, float) __asm("llvm.amdgcn.fmul.legacy");
__device__ __forceinline__ float mul_legacy(float a, float b) { return tvmi_fmul_legacy(a, b); }

// Multi-scale (FPN) launches: the level of every RoI is chosen IN the kernel
// (torchvision/ops/poolers.py:47-84, LevelMapper: floor(k0 + log2(sqrt(area)/s0) + eps) clamped to
// [k_min, k_max]).  RoIs of the multi-scale entries are always float32 image coordinates, whatever the
// feature dtype: the reference computes levels and sample coordinates from the fp32 boxes too
// (poolers.py:199-222, and _autograd_registrations.py:246 under autocast).
struct MsLevels {
  const void* ptr[kMaxLevels];
  int H[kMaxLevels];
  int W[kMaxLevels];
  float scale[kMaxLevels];
  int n_levels;
  int k_min, k_max;
  float s0, lvl0, eps;
};

template <typename R>
__device__ __forceinline__ int fpn_level(const R* roi, const MsLevels& lv) {
  const float x1 = ld(roi + 1), y1 = ld(roi + 2), x2 = ld(roi + 3), y2 = ld(roi + 4);
  const float s = sqrtf((x2 - x1) * (y2 - y1));
  float t = floorf(lv.lvl0 + log2f(s / lv.s0) + lv.eps);
  t = fminf(fmaxf(t, (float)lv.k_min), (float)lv.k_max);
  int l = (t == t) ? (int)t - lv.k_min : 0;  // NaN area (torch.clamp keeps NaN, .to(int64) of NaN is the minimum): level 0
  return min(max(l, 0), lv.n_levels - 1);
}

int set_roi_option(const char* name, int64_t value);  // roi_align.hip
int get_roi_option(const char* name, int64_t* value);
int get_nms_option(const char* name, int64_t* value);
int get_dcn_option(const char* name, int64_t* value);
int set_nms_option(const char* name, int64_t value);  // nms.hip
int set_dcn_option(const char* name, int64_t value);  // deform_conv2d.hip
int get_dcn_bwd_option(const char* name, int64_t* value);
int set_dcn_bwd_option(const char* name, int64_t value);  // deform_conv2d_bwd.hip

}  // namespace tvmi
