// dcn_common.h — what the forward (deform_conv2d.hip) and the backward (deform_conv2d_bwd.hip) kernels of deform_conv2d share:
// the problem description, the sampling-location arithmetic of cpu/deform_conv2d_kernel.cpp:95-132 and the argument checks.
#pragma once

#include <algorithm>

#include "tvmi_common.h"

namespace tvmi {

struct DcnParams {
  int B, C, H, W;        // input
  int OC, kh, kw;        // weight [OC, C/groups, kh, kw]
  int oh, ow;            // output spatial
  int sh, sw, ph, pw, dh, dw;
  int groups, ogroups;   // weight groups, offset groups
  int use_mask;
  int ICg, OCg;          // channels per weight group
  int cpog;              // channels per offset group
  int xcd_tiles;         // option "dcn.xcd_tiles": pixel tiles dealt to the XCDs in contiguous ranges (see tile_of_block)
};

// One sampling location: 4 corner offsets (clamped to a valid address) and 4 weights
// (zeroed for corners outside the image), per cpu/deform_conv2d_kernel.cpp:95-132.
template <typename A>
struct Tap {
  int o1, o2, o3, o4;
  A w1, w2, w3, w4;
  A m;  // modulation mask (1 when unused)
};

template <typename A>
__device__ __forceinline__ void make_tap(Tap<A>& t, int H, int W, A h, A w, A mask) {
  // branch-free on purpose (round 4): with an early return for the out-of-image case the compiler kept the Tap of the fused
  // kernels in SCRATCH (conditional stores through the reference) and re-loaded it inside the slab loop
  const bool inside = !(h <= (A)-1 || (A)H <= h || w <= (A)-1 || (A)W <= w);
  const A hc = inside ? h : (A)0, wc = inside ? w : (A)0;
  const int hl = (int)floor(hc), wl = (int)floor(wc);
  const int hh_ = hl + 1, wh_ = wl + 1;
  const A lh = hc - (A)hl, lw = wc - (A)wl;
  const A hh = (A)1 - lh, hw = (A)1 - lw;
  const bool v_hl = hl >= 0, v_wl = wl >= 0, v_hh = hh_ <= H - 1, v_wh = wh_ <= W - 1;
  const int chl = v_hl ? hl : 0, cwl = v_wl ? wl : 0, chh = v_hh ? hh_ : H - 1, cwh = v_wh ? wh_ : W - 1;
  t.m = mask;
  t.o1 = inside ? chl * W + cwl : 0;
  t.o2 = inside ? chl * W + cwh : 0;
  t.o3 = inside ? chh * W + cwl : 0;
  t.o4 = inside ? chh * W + cwh : 0;
  t.w1 = (inside && v_hl && v_wl) ? hh * hw : (A)0;
  t.w2 = (inside && v_hl && v_wh) ? hh * lw : (A)0;
  t.w3 = (inside && v_hh && v_wl) ? lh * hw : (A)0;
  t.w4 = (inside && v_hh && v_wh) ? lh * lw : (A)0;
}

// The raw (offset_h, offset_w, mask) of one tap of one pixel, and the Tap they give: split so that the fused kernels can
// fetch the raw values of the NEXT offset-group segment while the current one is consumed (load_tap below does both at
// once: a dependent global-load round trip in front of the first gathers of every segment, nine times per tile at config 4).
// (the raw values stay in the tensor's type until they are used: converting a 16-bit value at load time is a use, and the
// wait it needs would sit right behind the load)
template <typename T>
struct TapRaw {
  T off_h, off_w, m;
};
template <typename T>
__device__ __forceinline__ TapRaw<T> tap_raw_identity() {
  TapRaw<T> r;
  st(&r.off_h, 0.f);
  st(&r.off_w, 0.f);
  st(&r.m, 1.f);
  return r;
}
template <typename T>
__device__ __forceinline__ void load_tap_raw(TapRaw<T>& r, const DcnParams& p, const T* __restrict__ offset,
                                             const T* __restrict__ mask, int b, int og, int tap, int oy, int ox) {
  const int64_t plane = (int64_t)p.oh * p.ow;
  const int64_t pix = (int64_t)oy * p.ow + ox;
  const T* optr = offset + ((int64_t)(b * p.ogroups + og) * 2 * p.kh * p.kw) * plane;
  r.off_h = optr[(int64_t)(2 * tap) * plane + pix];
  r.off_w = optr[(int64_t)(2 * tap + 1) * plane + pix];
  if (p.use_mask) r.m = mask[((int64_t)(b * p.ogroups + og) * p.kh * p.kw + tap) * plane + pix];
}
template <typename T, typename A>
__device__ __forceinline__ void tap_from_raw(Tap<A>& t, const DcnParams& p, const TapRaw<T>& r, int tap, int oy, int ox) {
  const int i = tap / p.kw, j = tap - i * p.kw;
  const A y = (A)(oy * p.sh - p.ph) + (A)(i * p.dh) + (A)ld(&r.off_h);
  const A x = (A)(ox * p.sw - p.pw) + (A)(j * p.dw) + (A)ld(&r.off_w);
  make_tap<A>(t, p.H, p.W, y, x, (A)ld(&r.m));
}

template <typename T, typename A>
__device__ __forceinline__ void load_tap(Tap<A>& t, const DcnParams& p, const T* __restrict__ offset,
                                         const T* __restrict__ mask, int b, int og, int tap, int oy, int ox) {
  const int i = tap / p.kw, j = tap - i * p.kw;
  const int64_t plane = (int64_t)p.oh * p.ow;
  const int64_t pix = (int64_t)oy * p.ow + ox;
  const T* optr = offset + ((int64_t)(b * p.ogroups + og) * 2 * p.kh * p.kw) * plane;
  const A off_h = ld(optr + (int64_t)(2 * tap) * plane + pix);
  const A off_w = ld(optr + (int64_t)(2 * tap + 1) * plane + pix);
  A mval = (A)1;
  if (p.use_mask) mval = ld(mask + ((int64_t)(b * p.ogroups + og) * p.kh * p.kw + tap) * plane + pix);
  const A y = (A)(oy * p.sh - p.ph) + (A)(i * p.dh) + off_h;
  const A x = (A)(ox * p.sw - p.pw) + (A)(j * p.dw) + off_w;
  make_tap<A>(t, p.H, p.W, y, x, mval);
}
template <typename T, typename A>
__device__ __forceinline__ A sample_tap(const Tap<A>& t, const T* __restrict__ plane) {
  const A v1 = ld(plane + t.o1), v2 = ld(plane + t.o2), v3 = ld(plane + t.o3), v4 = ld(plane + t.o4);
  return t.m * (t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4);
}

inline int fill_params_common(DcnParams& p, int64_t B, int64_t C, int64_t H, int64_t W, int64_t OC, int64_t kh, int64_t kw,
                int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw, int64_t groups,
                int64_t ogroups, int use_mask) {
  TVMI_CHECK_ARG(kh > 0 && kw > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && ph >= 0 && pw >= 0,
                 "deform_conv2d: invalid kernel/stride/pad/dilation");
  TVMI_CHECK_ARG(groups > 0 && ogroups > 0 && C % groups == 0 && OC % groups == 0 && C % ogroups == 0,
                 "deform_conv2d: channels not divisible by groups");
  p.B = (int)B;
  p.C = (int)C;
  p.H = (int)H;
  p.W = (int)W;
  p.OC = (int)OC;
  p.kh = (int)kh;
  p.kw = (int)kw;
  p.sh = (int)sh;
  p.sw = (int)sw;
  p.ph = (int)ph;
  p.pw = (int)pw;
  p.dh = (int)dh;
  p.dw = (int)dw;
  p.groups = (int)groups;
  p.ogroups = (int)ogroups;
  p.use_mask = use_mask;
  p.oh = (int)((H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1);
  p.ow = (int)((W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1);
  p.ICg = (int)(C / groups);
  p.OCg = (int)(OC / groups);
  p.cpog = (int)(C / ogroups);
  p.xcd_tiles = 0;
  TVMI_CHECK_ARG(p.oh > 0 && p.ow > 0, "deform_conv2d: calculated output size too small");
  TVMI_CHECK_ARG(H * W < (1ll << 31), "deform_conv2d: plane too large");
  return 0;
}

inline int dcn_round_up(int v, int m) { return (v + m - 1) / m * m; }
inline size_t dcn_align256(size_t b) { return (b + 255) & ~(size_t)255; }
inline dim3 dcn_grid1d(int64_t total) { return dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 1 << 20)); }

}  // namespace tvmi
