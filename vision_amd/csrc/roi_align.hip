// roi_align.hip — RoIAlign forward for gfx950 (MI355X).  (Backward: roi_align_bwd.hip.)
//
// Semantics: torchvision/csrc/ops/cpu/roi_align_kernel.cpp:18-115 and cpu/roi_align_common.h:32-124
// (sample -> 4 taps + weights).  The TU is built with -ffp-contract=off so the sample-coordinate
// arithmetic rounds exactly like the reference's x86 build; fused multiply-adds are only used where
// written explicitly.
//
// Design (not the reference's one-thread-per-output gather, cuda/roi_align_kernel.cu:68):
//   * unit = one wave64 = (RoI, channel chunk), no workgroup barrier anywhere;
//   * 7x7 / 14x14 bins with sampling_ratio 2: lane = output bin, its 4 samples live in registers, the
//     RoI's window rows go HBM/L2 -> LDS by 16-byte LDS-DMA (double-buffered), taps are read from LDS
//     (roi_align_fwd_dma / roi_align_fwd_ms_dma); what that kernel declines (windows above 8 DMA blocks
//     per channel) is mopped up by a register-staged wave kernel;
//   * other pooled shapes: per-wave axis tables in LDS + register-staged window (roi_align_fwd_wave);
//   * channels_last maps: lane = channel kernel (roi_align_fwd_nhwc), no layout copy;
//   * fp64: one thread per output element, arithmetic on the fly.
// Multi-scale (FPN) entries pick the level of every RoI in the kernel and serve all levels with one launch.
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>
#include <type_traits>

#include "roi_common.h"
#include "nms_step_device.h"

namespace tvmi {
namespace {

constexpr int kThreads = 256;

// ---------------------------------------------------------------------------------------
// Generic forward: one thread per output element, arithmetic on the fly (any grid size,
// any dtype incl. fp64).  Also the per-workgroup fallback of the tiled kernel.
template <typename T, typename A>
__device__ __forceinline__ A roi_align_point(const T* plane, int H, int W, const RoiGeom<A>& g,
                                             int ph, int pw) {
  A acc = (A)0;
  for (int iy = 0; iy < g.gh; ++iy) {
    int ylo, yhi;
    A ly, hy;
    const bool vy = axis_sample<A>(H, g.start_h, g.bin_h, g.gh, ph, iy, ylo, yhi, ly, hy);
    for (int ix = 0; ix < g.gw; ++ix) {
      int xlo, xhi;
      A lx, hx;
      const bool vx = axis_sample<A>(W, g.start_w, g.bin_w, g.gw, pw, ix, xlo, xhi, lx, hx);
      if (!(vy && vx)) continue;
      const A w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
      const A v1 = ld(plane + (int64_t)ylo * W + xlo);
      const A v2 = ld(plane + (int64_t)ylo * W + xhi);
      const A v3 = ld(plane + (int64_t)yhi * W + xlo);
      const A v4 = ld(plane + (int64_t)yhi * W + xhi);
      acc += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
    }
  }
  return acc / g.count;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_generic(
    const T* __restrict__ input, const T* __restrict__ rois, T* __restrict__ output, int64_t total,
    int C, int H, int W, int PH, int PW, double spatial_scale, int sr, int aligned) {
  using A = typename Acc<T>::type;
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * kThreads) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / ((int64_t)PW * PH)) % C);
    const int64_t k = idx / ((int64_t)PW * PH * C);
    const RoiGeom<A> g = roi_geom<T, A>(rois + k * 5, (A)spatial_scale, PH, PW, sr, aligned != 0);
    const T* plane = input + ((int64_t)g.batch * C + c) * H * W;
    st(output + idx, roi_align_point<T, A>(plane, H, W, g, ph, pw));
  }
}

// ---------------------------------------------------------------------------------------
// Wave-autonomous forward ("v3"): every wave owns one (RoI, channel chunk) unit end to end —
// its own axis tables, its own LDS window, no workgroup barrier anywhere, so the 16 resident
// waves of a CU drift apart and hide each other's memory latency.  The window is staged with a
// 2-D lane mapping (lane = (row-in-group, column), column count padded to a power of two) so
// that no integer division sits on the staging path, and out-of-tensor pad cells read the
// CLAMPED element — exactly what the reference multiplies its zero weight with.
constexpr int kWaveWin = 2048;   // floats of LDS window per wave (8 KiB)
constexpr int kWaveTab = 64;     // samples per axis held in the per-wave tables

struct WaveShared {
  float4 ytab[kWaveTab];  // {bits(lo), l, h, -}: one 16-byte LDS read per sample row
  float4 xtab[kWaveTab];
  float win[kWaveWin];
};


// Stages gc channel windows (rows y0.., cols x0.., clamped into the tensor), RG row groups of CH channels at a time:
// RG*CH (12..16) independent loads are in flight per lane before the first LDS write.  Every load and every LDS write is
// UNCONDITIONAL: a row group past the window (RG is the row-group count rounded up to the next instantiation) and a channel
// past the group re-read the last row / channel — same address, same value, same LDS slot.  (Rounds 1-4 guarded them with the
// wave-uniform `if (rg < nrg)`: the compiler turned every guarded load into a branch, a load and an `s_waitcnt vmcnt(0)` for
// the phi — 7 of 16 loads synchronous; profiles/r04_isa_waits.txt, tests/test_isa_guards.py.)  Lanes beyond the window edge
// re-read the edge element, so there is no predication and no integer division anywhere on this path; the channel base
// pointer is wave-uniform (SGPR) and the per-lane offsets are 32-bit.
template <typename T, int RG, int CH>
__device__ __forceinline__ void stage_window(float* __restrict__ win, const T* __restrict__ in_cg, int64_t plane_sz,
                                             int gc, int wsz, int rpi, int rsub, int colc, int gxc, int y0, int wh,
                                             int wstride, int H, int W) {
  unsigned gbyte[RG];   // zero-extended byte offsets off the uniform channel pointer: one VGPR per row group, SGPR-base loads
  int loff[RG];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    const int r = min(rg * rpi + rsub, wh - 1);
    gbyte[rg] = (unsigned)(min(y0 + r, H - 1) * W + gxc) * (unsigned)sizeof(T);
    loff[rg] = r * wstride + colc;
  }
  for (int ch0 = 0; ch0 < gc; ch0 += CH) {
    float v[CH][RG];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const char* chp = reinterpret_cast<const char*>(in_cg + (int64_t)min(ch0 + c, gc - 1) * plane_sz);
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        // the empty asm keeps the zero-extension of the offset in THIS block: hoisted out of the channel loop it becomes a 64-bit
        // VGPR pair per row group and the load loses its SGPR-base + 32-bit-offset addressing form (instruction selection works
        // per block) — 32 more VGPRs at RG = 16: the 7x7 wave kernels went from 126 to 243 VGPRs and the scheduler serialised the loads
        unsigned o = gbyte[rg];
        asm("" : "+v"(o));
        v[c][rg] = ld(reinterpret_cast<const T*>(chp + o));
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float* wch = win + min(ch0 + c, gc - 1) * wsz;
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) wch[loff[rg]] = v[c][rg];
    }
  }
}

// nrg (1..16 row groups, wave-uniform) -> the instantiation with the fewest re-read rows (none up to 8, at most one beyond)
template <typename T>
__device__ __forceinline__ void stage_window_setup(float* __restrict__ win, const T* __restrict__ in_cg,
                                                   int64_t plane_sz, int gc, int nrg, int wsz, int rpi, int rsub,
                                                   int colc, int gxc, int y0, int wh, int wstride, int H, int W) {
#define TVMI_STAGE(RG, CH) stage_window<T, RG, CH>(win, in_cg, plane_sz, gc, wsz, rpi, rsub, colc, gxc, y0, wh, wstride, H, W)
  switch (nrg) {
    case 1: TVMI_STAGE(1, 16); break;
    case 2: TVMI_STAGE(2, 8); break;
    case 3: TVMI_STAGE(3, 5); break;
    case 4: TVMI_STAGE(4, 4); break;
    case 5: TVMI_STAGE(5, 3); break;
    case 6: TVMI_STAGE(6, 2); break;
    case 7: TVMI_STAGE(7, 2); break;
    case 8: TVMI_STAGE(8, 2); break;
    case 9: case 10: TVMI_STAGE(10, 1); break;
    case 11: case 12: TVMI_STAGE(12, 1); break;
    case 13: case 14: TVMI_STAGE(14, 1); break;
    default: TVMI_STAGE(16, 1); break;
  }
#undef TVMI_STAGE
}

template <typename T, typename R, int PHT, int PWT, int SRT>
__device__ __forceinline__ void roi_align_fwd_wave_unit(WaveShared& s, const T* __restrict__ input,
                                                        const R* __restrict__ rois, T* __restrict__ output,
                                                        int C, int H, int W, int PH_, int PW_, float spatial_scale,
                                                        int sr_, int aligned, int k, int c0, int chunk) {
  const int PH = PHT > 0 ? PHT : PH_;
  const int PW = PWT > 0 ? PWT : PW_;
  const int sr = SRT > 0 ? SRT : sr_;
  const int PHW = PH * PW;
  const int lane = threadIdx.x & 63;
  const int cc = min(chunk, C - c0);
  RoiGeom<float> g = roi_geom<R, float>(rois + (int64_t)k * 5, spatial_scale, PH, PW, sr, aligned != 0);
  if (SRT > 0) {
    g.gh = SRT;
    g.gw = SRT;
  }
  T* out = output + ((int64_t)k * C + c0) * PHW;
  const T* in0 = input + ((int64_t)g.batch * C + c0) * H * W;
  const int64_t plane_sz = (int64_t)H * W;
  const int gh = g.gh, gw = g.gw;
  const int ny = PH * gh, nx = PW * gw;
  if (gh <= 0 || gw <= 0) {
    for (int o = lane; o < cc * PHW; o += 64) st(out + o, 0.f);
    return;
  }
  if (ny > kWaveTab || nx > kWaveTab) {  // tables do not fit: arithmetic on the fly
    for (int o = lane; o < cc * PHW; o += 64) {
      const int c = o / PHW, bin = o - c * PHW;
      const int ph = bin / PW, pw = bin - ph * PW;
      st(out + o, roi_align_point<T, float>(in0 + c * plane_sz, H, W, g, ph, pw));
    }
    return;
  }
  // ---- axis tables (one lane per sample) and the window they span
  int y0, y1, x0, x1;
  {
    int lo = 0, hi = 0;
    float l = 0.f, h = 0.f;
    bool v = false;
    if (lane < ny) v = axis_sample<float>(H, g.start_h, g.bin_h, gh, lane / gh, lane % gh, lo, hi, l, h);
    const unsigned long long bal = __ballot(v);
    if (bal == 0ull) {
      for (int o = lane; o < cc * PHW; o += 64) st(out + o, 0.f);
      return;
    }
    {  // sample coordinates are monotonic in the lane index (either direction: malformed RoIs with
       // aligned=True have negative bins), so the extremes sit at the first / last valid lane
      const int fl = __builtin_ctzll(bal), ll = 63 - __builtin_clzll(bal);
      y0 = min(__builtin_amdgcn_readlane(lo, fl), __builtin_amdgcn_readlane(lo, ll));
      y1 = max(__builtin_amdgcn_readlane(hi, fl), __builtin_amdgcn_readlane(hi, ll));
    }
    if (lane < ny) s.ytab[lane] = make_float4(__int_as_float(v ? lo : y0), l, h, v ? 1.f : 0.f);
  }
  {
    int lo = 0, hi = 0;
    float l = 0.f, h = 0.f;
    bool v = false;
    if (lane < nx) v = axis_sample<float>(W, g.start_w, g.bin_w, gw, lane / gw, lane % gw, lo, hi, l, h);
    const unsigned long long bal = __ballot(v);
    if (bal == 0ull) {
      for (int o = lane; o < cc * PHW; o += 64) st(out + o, 0.f);
      return;
    }
    {
      const int fl = __builtin_ctzll(bal), ll = 63 - __builtin_clzll(bal);
      x0 = min(__builtin_amdgcn_readlane(lo, fl), __builtin_amdgcn_readlane(lo, ll));
      x1 = max(__builtin_amdgcn_readlane(hi, fl), __builtin_amdgcn_readlane(hi, ll));
    }
    if (lane < nx) s.xtab[lane] = make_float4(__int_as_float(v ? lo : x0), l, h, v ? 1.f : 0.f);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // divide like the reference; a power-of-two count (the usual 2x2 grid) is an exact multiply
  const float count = g.count;
  const int icount = gh * gw;
  const bool pow2 = icount > 0 && (icount & (icount - 1)) == 0;
  const float inv_count = 1.f / count;
  const int wh = y1 - y0 + 2, ww = x1 - x0 + 2;  // +1 pad row/col: only ever met by a zero weight
  int wpad = 1;
  while (wpad < ww) wpad <<= 1;
  const int wstride = ww | 1;
  const int wsz = wh * wstride;
  const int rpi = wpad <= 64 ? 64 / wpad : 1;  // window rows staged per wave-instruction
  const int nrg = (wh + rpi - 1) / rpi;
  int G = (ww <= 64 && nrg <= 16) ? kWaveWin / wsz : 0;
  if (G > cc) G = cc;

  if (G < 1) {
    // window too large (or staging disabled): table-driven gathers straight from global memory
    for (int o = lane; o < cc * PHW; o += 64) {
      const int c = o / PHW, bin = o - c * PHW;
      const int ph = bin / PW, pw = bin - ph * PW;
      const T* plane = in0 + c * plane_sz;
      float acc = 0.f;
      for (int iy = 0; iy < gh; ++iy) {
        const float4 ye = s.ytab[ph * gh + iy];
        const int ylo = __float_as_int(ye.x);
        const float ly = ye.y, hy = ye.z;
        const T* r0 = plane + (int64_t)ylo * W;
        const T* r1 = plane + (int64_t)min(ylo + 1, H - 1) * W;
        for (int ix = 0; ix < gw; ++ix) {
          const float4 xe = s.xtab[pw * gw + ix];
          const int xlo = __float_as_int(xe.x), xhi = min(xlo + 1, W - 1);
          const float lx = xe.y, hx = xe.z;
          // a sample the reference skips (roi_align_common.h:60-73) contributes an exact zero; a VALID sample keeps the
          // reference's own 0 * pixel products (a NaN under a zero weight is a NaN there too)
          const float sv = (hy * hx) * ld(r0 + xlo) + (hy * lx) * ld(r0 + xhi) + (ly * hx) * ld(r1 + xlo) + (ly * lx) * ld(r1 + xhi);
          acc += (ye.w != 0.f && xe.w != 0.f) ? sv : 0.f;
        }
      }
      st(out + o, pow2 ? acc * inv_count : acc / count);
    }
    return;
  }

  const int rsub = lane / wpad, col = lane - rsub * wpad;
  const int colc = min(col, ww - 1);
  const int gxc = min(x0 + colc, W - 1);
  const int wbias = y0 * wstride + x0;  // tables hold absolute coordinates
  for (int cg = 0; cg < cc; cg += G) {
    const int gc = min(G, cc - cg);
    const T* in_cg = in0 + (int64_t)cg * plane_sz;
    stage_window_setup<T>(s.win, in_cg, plane_sz, gc, nrg, wsz, rpi, rsub, colc, gxc, y0, wh, wstride, H, W);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nout = gc * PHW;
    for (int o = lane; o < nout; o += 64) {
      const int ch = o / PHW, bin = o - ch * PHW;
      const int ph = bin / PW, pw = bin - ph * PW;
      const float* wbase = s.win + (ch * wsz - wbias);
      float acc = 0.f;
      for (int iy = 0; iy < gh; ++iy) {
        const float4 ye = s.ytab[ph * gh + iy];
        const float* row = wbase + __float_as_int(ye.x) * wstride;
        const float ly = ye.y, hy = ye.z;
        for (int ix = 0; ix < gw; ++ix) {
          const float4 xe = s.xtab[pw * gw + ix];
          const float* p = row + __float_as_int(xe.x);
          const float lx = xe.y, hx = xe.z;
          // a sample the reference skips (roi_align_common.h:60-73) contributes an exact zero (its taps point at the window
          // origin); valid samples keep the reference's 0 * pixel products; the x / y edges are exact (the pad column / row
          // holds the clamped pixel the reference reads twice)
          const float sv = (hy * hx) * p[0] + (hy * lx) * p[1] + (ly * hx) * p[wstride] + (ly * lx) * p[wstride + 1];
          acc += (ye.w != 0.f && xe.w != 0.f) ? sv : 0.f;
        }
      }
      st(out + cg * PHW + o, pow2 ? acc * inv_count : acc / count);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ---------------------------------------------------------------------------------------
// Fast wave unit for the compile-time shapes (7x7 / 14x14, sampling_ratio 2): lane = output
// bin.  Each lane keeps the 4 samples of its bin(s) in REGISTERS for the whole unit — one LDS
// offset and the separable bilinear factors (ly,hy / lx,hx) per sample — so the per-channel
// inner loop is nothing but 2 ds_read2_b32 + 6 multiply-adds per sample and one coalesced
// store of the PH*PW outputs of that channel.  The bilinear form hy*(hx*v1+lx*v2) +
// ly*(hx*v3+lx*v4) is evaluated with explicit fused multiply-adds; it differs from the
// reference's (hy*hx)*v1+... association by rounding only (<1e-6 relative).
template <typename T, typename R, int PHT, int PWT, int SRT>
__device__ __forceinline__ void roi_align_fwd_wave_fast(WaveShared& s, const T* __restrict__ input,
                                                        const R* __restrict__ rois, T* __restrict__ output,
                                                        int C, int H, int W, float spatial_scale, int aligned,
                                                        int k, int c0, int chunk) {
  constexpr int PHW = PHT * PWT;
  constexpr int NB = (PHW + 63) / 64;  // bins per lane
  constexpr int NS = SRT * SRT;        // samples per bin
  constexpr int ny = PHT * SRT, nx = PWT * SRT;
  static_assert(ny <= 64 && nx <= 64, "axis samples must fit one wave");
  const int lane = threadIdx.x & 63;
  const int cc = min(chunk, C - c0);
  RoiGeom<float> g = roi_geom<R, float>(rois + (int64_t)k * 5, spatial_scale, PHT, PWT, SRT, aligned != 0);
  T* out = output + ((int64_t)k * C + c0) * PHW;
  const T* in0 = input + ((int64_t)g.batch * C + c0) * H * W;
  const int64_t plane_sz = (int64_t)H * W;

  // ---- window bounds: one lane per axis sample, first/last valid sample via ballot
  int y0, y1, x0, x1;
  {
    int lo = 0, hi = 0;
    float l, h;
    bool v = false;
    if (lane < ny) v = axis_sample<float>(H, g.start_h, g.bin_h, SRT, lane / SRT, lane % SRT, lo, hi, l, h);
    const unsigned long long by = __ballot(v);
    int lo2 = 0, hi2 = 0;
    bool v2 = false;
    if (lane < nx) v2 = axis_sample<float>(W, g.start_w, g.bin_w, SRT, lane / SRT, lane % SRT, lo2, hi2, l, h);
    const unsigned long long bx = __ballot(v2);
    if (by == 0ull || bx == 0ull) {  // every sample of one axis is outside: all outputs are 0
      for (int o = lane; o < cc * PHW; o += 64) st(out + o, 0.f);
      return;
    }
    {  // monotonic in the lane index, either direction (negative bins of malformed RoIs)
      const int fy_ = __builtin_ctzll(by), ly_ = 63 - __builtin_clzll(by);
      const int fx_ = __builtin_ctzll(bx), lx_ = 63 - __builtin_clzll(bx);
      y0 = min(__builtin_amdgcn_readlane(lo, fy_), __builtin_amdgcn_readlane(lo, ly_));
      y1 = max(__builtin_amdgcn_readlane(hi, fy_), __builtin_amdgcn_readlane(hi, ly_));
      x0 = min(__builtin_amdgcn_readlane(lo2, fx_), __builtin_amdgcn_readlane(lo2, lx_));
      x1 = max(__builtin_amdgcn_readlane(hi2, fx_), __builtin_amdgcn_readlane(hi2, lx_));
    }
  }
  const int wh = y1 - y0 + 2, ww = x1 - x0 + 2;  // +1 pad row/col: only ever met by a zero weight
  int wpad = 1;
  while (wpad < ww) wpad <<= 1;
  const int wstride = ww | 1;
  const int wsz = wh * wstride;
  const int rpi = wpad <= 64 ? 64 / wpad : 1;
  const int nrg = (wh + rpi - 1) / rpi;
  int G = (ww <= 64 && nrg <= 16) ? kWaveWin / wsz : 0;
  if (G > cc) G = cc;

  // ---- per-lane sample set-up (registers)
  int off[NB][NS];
  float fy[NB][SRT][2], fx[NB][SRT][2];  // {l, h} per axis sample
  int gy[NB][SRT], gxx[NB][SRT];         // absolute low indices (global-gather fallback)
  unsigned vmask[NB];                    // bit iy*SRT+ix: the sample is one the reference evaluates (not skipped)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int bin = min(lane + 64 * b, PHW - 1);
    const int ph = bin / PWT, pw = bin - ph * PWT;
    int ylo[SRT], xlo[SRT];
    bool vys[SRT], vxs[SRT];
#pragma unroll
    for (int i = 0; i < SRT; ++i) {
      int hi;
      float l, h;
      const bool vy = axis_sample<float>(H, g.start_h, g.bin_h, SRT, ph, i, ylo[i], hi, l, h);
      fy[b][i][0] = l;
      fy[b][i][1] = h;
      if (!vy) ylo[i] = y0;
      gy[b][i] = ylo[i];
      vys[i] = vy;
      const bool vx = axis_sample<float>(W, g.start_w, g.bin_w, SRT, pw, i, xlo[i], hi, l, h);
      fx[b][i][0] = l;
      fx[b][i][1] = h;
      if (!vx) xlo[i] = x0;
      gxx[b][i] = xlo[i];
      vxs[i] = vx;
    }
    vmask[b] = 0;
#pragma unroll
    for (int iy = 0; iy < SRT; ++iy)
#pragma unroll
      for (int ix = 0; ix < SRT; ++ix) vmask[b] |= (vys[iy] && vxs[ix]) ? (1u << (iy * SRT + ix)) : 0u;
#pragma unroll
    for (int iy = 0; iy < SRT; ++iy)
#pragma unroll
      for (int ix = 0; ix < SRT; ++ix) off[b][iy * SRT + ix] = (ylo[iy] - y0) * wstride + (xlo[ix] - x0);
  }
  const float inv_count = 1.f / (float)NS;  // SRT*SRT is a power of two here or handled below
  constexpr bool kPow2 = (NS & (NS - 1)) == 0;

  if (G < 1) {
    // window too large for LDS: same register set-up, taps straight from global memory
    for (int c = 0; c < cc; ++c) {
      const T* plane = in0 + (int64_t)c * plane_sz;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float acc = 0.f;
#pragma unroll
        for (int iy = 0; iy < SRT; ++iy) {
          const T* r0 = plane + (int64_t)gy[b][iy] * W;
          const T* r1 = plane + (int64_t)min(gy[b][iy] + 1, H - 1) * W;
#pragma unroll
          for (int ix = 0; ix < SRT; ++ix) {
            const int xl = gxx[b][ix], xh = min(xl + 1, W - 1);
            const float t0 = __builtin_fmaf(fx[b][ix][0], ld(r0 + xh), fx[b][ix][1] * ld(r0 + xl));
            const float t1 = __builtin_fmaf(fx[b][ix][0], ld(r1 + xh), fx[b][ix][1] * ld(r1 + xl));
            float a2 = __builtin_fmaf(fy[b][iy][1], t0, acc);
            a2 = __builtin_fmaf(fy[b][iy][0], t1, a2);
            acc = ((vmask[b] >> (iy * SRT + ix)) & 1u) ? a2 : acc;   // a sample the reference skips adds an exact zero
          }
        }
        const int bin = lane + 64 * b;
        if (bin < PHW) st(out + c * PHW + bin, kPow2 ? acc * inv_count : acc / (float)NS);
      }
    }
    return;
  }

  const int rsub = lane / wpad, col = lane - rsub * wpad;
  const int colc = min(col, ww - 1);
  const int gxc = min(x0 + colc, W - 1);
  for (int cg = 0; cg < cc; cg += G) {
    const int gc = min(G, cc - cg);
    const T* in_cg = in0 + (int64_t)cg * plane_sz;
    stage_window_setup<T>(s.win, in_cg, plane_sz, gc, nrg, wsz, rpi, rsub, colc, gxc, y0, wh, wstride, H, W);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int ch = 0; ch < gc; ++ch) {
      const float* wbase = s.win + ch * wsz;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float acc = 0.f;
#pragma unroll
        for (int iy = 0; iy < SRT; ++iy) {
#pragma unroll
          for (int ix = 0; ix < SRT; ++ix) {
            // a skipped sample (cpu/roi_align_common.h:60-73) has its taps pointed at the window origin: it adds an exact
            // zero (select, not a product — a NaN / Inf there must not reach the sum); valid samples keep the reference's
            // products, zero weights included; the x / y edges are exact already (the pad column / row holds the clamped
            // pixel the reference reads twice)
            const float* p = wbase + off[b][iy * SRT + ix];
            const float t0 = __builtin_fmaf(fx[b][ix][0], p[1], fx[b][ix][1] * p[0]);
            const float t1 = __builtin_fmaf(fx[b][ix][0], p[wstride + 1], fx[b][ix][1] * p[wstride]);
            float a2 = __builtin_fmaf(fy[b][iy][1], t0, acc);
            a2 = __builtin_fmaf(fy[b][iy][0], t1, a2);
            acc = ((vmask[b] >> (iy * SRT + ix)) & 1u) ? a2 : acc;
          }
        }
        const int bin = lane + 64 * b;
        if (bin < PHW) st(out + (cg + ch) * PHW + bin, kPow2 ? acc * inv_count : acc / (float)NS);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ---------------------------------------------------------------------------------------
// LDS-DMA wave unit (fp32, compile-time shapes): the window rows go HBM/L2 -> LDS directly
// (global_load_lds_dword: per-lane global address, LDS slot = wave base + lane*4), 16 DMA
// instructions per pass, DOUBLE-BUFFERED — the DMAs of pass p+1 are in flight while pass p is
// gathered from LDS, and a counted `s_waitcnt vmcnt(16)` (never 0 inside the loop) retires
// exactly the previous pass.  No staging VGPRs, no ds_write, no workgroup barrier; the window
// image is [channel][row][wpad] with wpad = 64 / rows-per-instruction, so a lane's slot is a
// pure function of its lane id, and lanes past the window edge fetch the clamped edge element
// (the very element the reference multiplies its zero weight with).
// DMA instructions (x 1 KiB) per pass and buffer.  (Measured, round 4: three blocks per buffer at 7 x 7 — 7.8 KB of LDS per
// wave, FIVE workgroups = 20 waves per CU, windows of 4 .. 6 row groups single-buffered — is 2 % SLOWER than four blocks and four
// workgroups, 0.228 vs 0.224 ms box-normalised: the kernel does not want occupancy, it wants its prefetch depth.)
constexpr int dma_per_pass(int /*pht*/) { return 4; }
// (Measured and removed, round 4: "packed" staging — the 64 lanes of a DMA instruction assigned to a flat (channel of the pass,
// window row, piece) index, so that a small window puts several channels into one instruction and a pass always fills its four
// blocks: 2.05 -> 1.75 DMA instructions per RoI-channel on the FPN workload, up to 8x more channels in flight per wave.  Correct
// (every route test green) and 12 % SLOWER, 0.248 vs 0.221 ms (fp32 7 x 7; bf16 0.199 vs 0.176): an instruction whose lanes span
// several channel planes touches as many cache lines as the instructions it replaces — the texture path is paid per line, not
// per instruction — and the per-lane 64-bit plane offsets plus the per-channel tap address arithmetic come on top.
// profiles/r04_roi_variants_packed.json.)
constexpr int kDmaBlk = 260;                     // LDS floats per DMA instruction block (256 + 4 skew, 16-B aligned)
constexpr int dma_buf_floats(int pht) { return dma_per_pass(pht) * kDmaBlk; }   // floats per buffer

constexpr int kDmaStage = 400;   // floats: finished [channel][bin] rows of a wave, parked until they leave as 16-byte stores
template <int PHT>
struct DmaShared {
  __attribute__((aligned(16))) float buf[2 * dma_buf_floats(PHT)];
  __attribute__((aligned(16))) float stage[kDmaStage];
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Window image in LDS, built by 16-byte LDS-DMA pieces (global_load_lds_dwordx4: every lane
// moves 4 consecutive floats of a window row; 4-byte source alignment is enough on gfx950).
// One DMA instruction = 64 lanes = `rpi` rows of `lpr` (odd) quads; its 256-float block starts
// every 260 floats.  slot(r, c) = (r / rpi) * 260 + (r % rpi) * 4*lpr + c.
// The window is [y0, y0+wh) x [x0, x0+4*nq): no pad row/column is needed because a sample that
// sits on the last row/column (where the reference uses x_high = x_low with weight 0) is
// re-expressed on the pair (dim-2, dim-1) with factors (0, 1) — the same value, never an
// out-of-row read; the window is shifted left when it would run past the row end.
// n / d for 0 <= n < 1024 and 1 <= d <= 64 as a multiply-shift: m = ceil(65536 / d) (v_rcp_f32 is exact for powers of two — the
// only d with an integral 65536 / d — and its 1-ulp error is far below the 1/64 that separates any other quotient from an integer),
// and n * m >> 16 = floor(n / d) because n * (m - 65536 / d) / 65536 < 1 / d.  The unit set-up divides by two wave-uniform
// run-time values (pieces per row, rows per DMA instruction) seven times per lane: 2 reciprocals instead of 7 u32 divisions.
__device__ __forceinline__ unsigned small_div_magic(int d) { return (unsigned)ceilf(65536.f * __builtin_amdgcn_rcpf((float)d)); }
__device__ __forceinline__ int small_div(int n, unsigned m) { return (int)(((unsigned)n * m) >> 16); }

struct DmaWindow {
  int state, y0, x0, wh, nq, lpr, rpi, nrg;
  unsigned m_lpr, m_rpi;   // small_div_magic of lpr / rpi
  int sparse;  // 1: the LDS image holds only the SAMPLED rows — slot 2s / 2s+1 = low / high tap row of y-sample s
};

// EPP = elements per 16-byte DMA piece (4 for fp32, 8 for fp16/bf16); nq then counts pieces.
template <int PHT, int PWT, int SRT, int EPP = 4>
__device__ __forceinline__ DmaWindow dma_window(const RoiGeom<float>& g, int H, int W) {
  constexpr int ny = PHT * SRT, nx = PWT * SRT;
  const int lane = threadIdx.x & 63;
  DmaWindow w;
  w.state = 2;
  w.y0 = w.x0 = w.wh = w.nq = w.lpr = w.rpi = w.nrg = w.sparse = 0;
  w.m_lpr = w.m_rpi = 0;
  if (H < 2 || W < EPP) return w;
  int lo = 0, hi = 0, lo2 = 0;
  float l, h;
  bool v = false, v2 = false;
  // y: the reference's own (low, high) rows — on the bottom edge high = low, both tap rows are then row H-1;
  // x: the shifted form (the pair (W-2, W-1) with factors (0, 1) on the right edge) so that a tap pair is one LDS read
  if (lane < ny) v = axis_sample<float>(H, g.start_h, g.bin_h, SRT, lane / SRT, lane % SRT, lo, hi, l, h);
  if (lane < nx) v2 = axis_sample_shifted(W, g.start_w, g.bin_w, SRT, lane / SRT, lane % SRT, lo2, l, h);
  const unsigned long long by = __ballot(v), bx = __ballot(v2);
  if (by == 0ull || bx == 0ull) {  // every sample of one axis is outside the map: all outputs are zero
    w.state = 0;
    return w;
  }
  // a RoI with SOME samples outside [-1, dim] (skipped by the reference, roi_align_common.h:60-73) is left to the
  // mop-up kernel, which gives skipped samples an exact zero; here every sample has a real window position
  if (by != (ny == 64 ? ~0ull : (1ull << ny) - 1ull) || bx != (nx == 64 ? ~0ull : (1ull << nx) - 1ull)) return w;
  // sample coordinates are monotonic in the lane index, in either direction (malformed RoIs with
  // aligned=True have negative bins): the extremes sit at the first / last valid lane
  const int ya = __builtin_amdgcn_readlane(lo, __builtin_ctzll(by)), yb = __builtin_amdgcn_readlane(lo, 63 - __builtin_clzll(by));
  const int yha = __builtin_amdgcn_readlane(hi, __builtin_ctzll(by)), yhb = __builtin_amdgcn_readlane(hi, 63 - __builtin_clzll(by));
  const int xa = __builtin_amdgcn_readlane(lo2, __builtin_ctzll(bx)), xb = __builtin_amdgcn_readlane(lo2, 63 - __builtin_clzll(bx));
  w.y0 = min(ya, yb);
  const int y1 = max(yha, yhb);
  const int x0 = min(xa, xb);
  const int x1 = max(xa, xb) + 1;
  w.wh = y1 - w.y0 + 1;
  // pieces start on 16-byte boundaries whenever the row pitch allows it (measured: no effect on
  // the texture-addresser cost either way; kept because it never costs more than one extra piece)
  const int xal = (W % EPP) == 0 ? x0 - (x0 % EPP) : x0;
  w.nq = (x1 - xal + EPP) / EPP;        // pieces covering [xal, x1]
  w.x0 = min(xal, W - EPP * w.nq);      // keep every piece inside the row (no-op in the aligned case)
  if (w.x0 < 0) return w;               // map narrower than the window image
  w.lpr = w.nq | 1;
  if (w.lpr > 64) return w;
  w.m_lpr = small_div_magic(w.lpr);
  w.rpi = small_div(64, w.m_lpr);
  w.m_rpi = small_div_magic(w.rpi);
  // A window taller than 2 rows per y-sample contains rows no sample touches (bins > 2 px): stage only the
  // sampled rows.  The kernel is bound by the bytes that cross L2 -> L1, and this drops ~1/3 of them on the
  // FPN workload (and pulls RoIs that were too tall for 8 DMA blocks per channel back onto this path).
  if (w.wh > 2 * ny) {
    w.sparse = 1;
    w.wh = 2 * ny;
  }
  w.nrg = small_div(w.wh + w.rpi - 1, w.m_rpi);
  // 1 = DMA path (<= 8 DMA instructions per channel), 2 = register-staged / global-gather path
  w.state = w.nrg <= 2 * dma_per_pass(PHT) ? 1 : 2;
  return w;
}

constexpr int kDmaBlkBytes = kDmaBlk * 4;  // 1040
constexpr int kMopHeader = 4;              // worklist of declined RoIs: [count, -, -, -][RoI indices ...]

template <typename T, int NRG, int kDmaPerPass>
__device__ __forceinline__ void dma_issue_pass(const T* __restrict__ in_pass, int64_t plane_sz, int gc,
                                               const int (&goff)[NRG], char* __restrict__ dst) {
  constexpr int G = NRG <= kDmaPerPass ? kDmaPerPass / NRG : 1;
#pragma unroll
  for (int c = 0; c < G; ++c) {
    const T* chp = in_pass + (int64_t)min(c, gc - 1) * plane_sz;  // tail: re-fetch the last channel
#pragma unroll
    for (int rg = 0; rg < NRG; ++rg) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(chp + goff[rg]), (lds_ptr_t)(dst + (c * NRG + rg) * kDmaBlkBytes), 16, 0, 0);
    }
  }
}

// LDS accesses the COMPILER must not see (fp32 staged outputs).  An LDS-DMA is a pending LDS write on the VM counter, and
// the waitcnt pass cannot tell the `stage` block from the DMA buffers: a plain C++ store into `stage` made it put
// `s_waitcnt vmcnt(0)` in front of every channel's arithmetic — i.e. wait for the DMAs of the NEXT pass that had just been
// issued, which serialised the double buffer (ISA of round 3: `s_waitcnt vmcnt(0) lgkmcnt(0)` in every pass loop; guarded
// now by tests/test_isa_guards.py).  The wave's own program order already orders these accesses against each other, and
// nothing DMA-written is ever touched here.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(lds_ptr_t)const_cast<void*>(p);
}
template <int IMM = 0>
__device__ __forceinline__ void lds_store_f32_opaque(unsigned addr, float v) {
  asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(IMM) : "memory");
}
__device__ __forceinline__ f32x4_t lds_load_f32x4_opaque(unsigned addr) {
  f32x4_t v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ float lds_load_f32_opaque(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// one 32-bit word of channels -> CPL floats
template <typename T>
__device__ __forceinline__ void unpack_word(unsigned w, float (&v)[4 / (int)sizeof(T)]) {
  if constexpr (std::is_same<T, float>::value) {
    v[0] = __builtin_bit_cast(float, w);
  } else if constexpr (std::is_same<T, __half>::value) {
    v[0] = __half2float(__ushort_as_half((unsigned short)(w & 0xffffu)));
    v[1] = __half2float(__ushort_as_half((unsigned short)(w >> 16)));
  } else {  // bfloat16: the upper half of an fp32
    v[0] = __builtin_bit_cast(float, w << 16);
    v[1] = __builtin_bit_cast(float, w & 0xffff0000u);
  }
}

// (Measured and removed, round 4: a tap pair as ONE ds_read_b64 at 4-byte alignment — or one 2-byte-aligned ds_read_b32 for
// the 16-bit types — is correct on gfx950 but 2.6x slower for the whole launch, 0.68 vs 0.26 ms: misaligned LDS reads are
// replayed; profiles/r04_roi_variants_v1.json.)
// `setup(off, fy, fx)`: the per-lane tap set-up of the unit (LDS slots of the tap rows, separable factors).  It runs AFTER the first
// pass of DMAs has been issued (round 6): in-kernel stamps (profiles/r06_roi_fwd_bound/stamps.patch) put a unit's first DMA 3.4 us
// after its start and the first pass's latency at 1.2 us — the taps are not needed before the pass has landed.
template <typename T, int PHT, int PWT, int SRT, int NRG, typename Setup>
__device__ __forceinline__ void roi_align_dma_passes(DmaShared<PHT>& s, const T* __restrict__ in0, T* __restrict__ out,
                                                     int64_t plane_sz, int cc, int H, int W, const DmaWindow& dw,
                                                     const RoiGeom<float>& g, Setup&& setup) {
  constexpr int PHW = PHT * PWT;
  constexpr int NB = (PHW + 63) / 64;
  constexpr int NS = SRT * SRT;
  constexpr int EPP = 16 / (int)sizeof(T);
  constexpr int kDmaPerPass = dma_per_pass(PHT), kDmaBuf = dma_buf_floats(PHT);
  static_assert(NRG <= 2 * kDmaPerPass, "row groups beyond both buffers");
  constexpr bool kDouble = NRG <= kDmaPerPass;        // more row groups use both buffers as one
  constexpr int G = kDouble ? kDmaPerPass / NRG : 1;
  constexpr bool kPow2 = (NS & (NS - 1)) == 0;
  const float inv_count = 1.f / (float)NS;
  const int lane = threadIdx.x & 63;
  const int lrow = small_div(lane, dw.m_lpr);
  const int rsub = min(lrow, dw.rpi - 1);
  const int q = min(lane - lrow * dw.lpr, dw.nq - 1);
  const int gx = dw.x0 + EPP * q;
  char* const bytes = reinterpret_cast<char*>(s.buf);
  int goff[NRG];
#pragma unroll
  for (int rg = 0; rg < NRG; ++rg) {
    const int t = min(rg * dw.rpi + rsub, dw.wh - 1);  // row slot of this lane in row group rg
    int y = dw.y0 + t;
    if (dw.sparse) {
      int lo, hi;
      float l, h;
      axis_sample<float>(H, g.start_h, g.bin_h, SRT, (t >> 1) / SRT, (t >> 1) % SRT, lo, hi, l, h);
      y = (t & 1) ? hi : lo;
    }
    goff[rg] = min(y, H - 1) * W + gx;
  }
  if (kDouble) dma_issue_pass<T, NRG, kDmaPerPass>(in0, plane_sz, min(G, cc), goff, bytes);
  int off[NB][NS][2];
  float fy[NB][SRT][2], fx[NB][SRT][2];
  setup(off, fy, fx);
  // Output rows are parked in LDS and leave SC channels at a time: the [channel][bin] block of a wave unit is contiguous
  // in the NCHW output, so 8 channels of 7 x 7 bins (1568 B) are two full-width 16-byte store instructions instead of eight
  // 49-lane dword stores — the texture path charges ~20 cycles per wave instruction whatever its width, and the stores were
  // a quarter of this kernel's VMEM instructions.  Same wave writes and reads the block: LDS order is program order.
  constexpr bool kStaged = sizeof(T) == 4;
  constexpr int SC = (kDmaStage * 4 / (int)sizeof(T)) / PHW >= 8 ? 8 : ((kDmaStage * 4 / (int)sizeof(T)) / PHW >= 2 ? 2 : 1);
  constexpr int kGroupVec = SC * PHW * (int)sizeof(T) / 16;          // 16-byte pieces of a full group
  static_assert(SC * PHW * (int)sizeof(T) % 16 == 0 && SC * PHW * (int)sizeof(T) <= kDmaStage * 4, "stage block");
  const unsigned stage_addr = lds_addr(s.stage);
  const bool vec_ok = (reinterpret_cast<uintptr_t>(out) & 15) == 0;  // wave-uniform (the unit's output base)
  int staged = 0;                                                    // channels parked in the stage block
  auto flush = [&](int first_ch, int nch) {
    if (nch == SC && vec_ok) {
      f32x4_t* dst = reinterpret_cast<f32x4_t*>(out + (int64_t)first_ch * PHW);
#pragma unroll
      for (int i = 0; i < (kGroupVec + 63) / 64; ++i) {
        const int v = lane + 64 * i;
        if (v < kGroupVec) __builtin_nontemporal_store(lds_load_f32x4_opaque(stage_addr + 16u * (unsigned)v), dst + v);
      }
    } else {
      for (int i = lane; i < nch * PHW; i += 64)
        st(out + (int64_t)first_ch * PHW + i, lds_load_f32_opaque(stage_addr + 4u * (unsigned)i));
    }
  };
  // Tap-pair LDS addresses.  With ONE channel per pass (NRG >= 3) and one bin per lane (7 x 7) the byte address of every
  // pair is a per-unit constant for each of the two buffers: both sets are formed once (16 VGPRs) and the channel loop
  // carries no address arithmetic at all; otherwise the wave-uniform (buffer, channel) base is added per pair.
  constexpr bool kPreAddr = NB == 1 && G == 1;
  constexpr int NBUF = kDouble ? 2 : 1;
  typedef const __attribute__((address_space(3))) T* lds_cptr_t;
  unsigned tap[kPreAddr ? NBUF : 1][NB][NS][2];
  if constexpr (kPreAddr) {
    const unsigned base = lds_addr(bytes);
#pragma unroll
    for (int u = 0; u < NBUF; ++u)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          tap[u][b][i][0] = base + (unsigned)(u * kDmaBuf * 4) + (unsigned)sizeof(T) * (unsigned)off[b][i][0];
          tap[u][b][i][1] = base + (unsigned)(u * kDmaBuf * 4) + (unsigned)sizeof(T) * (unsigned)off[b][i][1];
        }
  }
  const unsigned st_lane = stage_addr + 4u * (unsigned)lane;   // this lane's slot of channel 0 in the stage block (bin b: + 256 b)
  // one pass = gc channels of buffer `buf` (compile-time in the pre-addressed form)
  auto run_channels = [&](auto buf_tag, int cg, int gc, unsigned dyn_base) {
    constexpr int BUF = decltype(buf_tag)::value;
    for (int ch = 0; ch < gc; ++ch) {
      const unsigned chan = dyn_base + (unsigned)(ch * (NRG * kDmaBlkBytes));   // wave-uniform; unused when pre-addressed
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int bin = lane + 64 * b;
        // the WHOLE bin under the lane mask: loads hoisted above it were split off their pair partners
        if (b + 1 < NB || bin < PHW) {
          float acc = 0.f;
#pragma unroll
          for (int iy = 0; iy < SRT; ++iy) {
#pragma unroll
            for (int ix = 0; ix < SRT; ++ix) {
              const int i = iy * SRT + ix;
              // integer -> LDS pointer -> generic pointer: the address-space inference turns the loads back into ds_reads
              const T *q0, *q1;
              if constexpr (kPreAddr) {
                q0 = (const T*)reinterpret_cast<lds_cptr_t>((uintptr_t)tap[BUF][b][i][0]);
                q1 = (const T*)reinterpret_cast<lds_cptr_t>((uintptr_t)tap[BUF][b][i][1]);
              } else {
                const T* wbase = reinterpret_cast<const T*>(bytes + chan);
                q0 = wbase + off[b][i][0];
                q1 = wbase + off[b][i][1];
              }
              const float t0 = __builtin_fmaf(fx[b][ix][0], ld(q0 + 1), mul_legacy(fx[b][ix][1], ld(q0)));   // x edge: 0 * (pixel W-2) = 0
              const float t1 = __builtin_fmaf(fx[b][ix][0], ld(q1 + 1), mul_legacy(fx[b][ix][1], ld(q1)));
              acc = __builtin_fmaf(fy[b][iy][1], t0, acc);
              acc = __builtin_fmaf(fy[b][iy][0], t1, acc);
            }
          }
          const float r = kPow2 ? acc * inv_count : acc / (float)NS;
          if constexpr (kStaged) {
            const unsigned sa = st_lane + (unsigned)(staged * (PHW * 4));
            if (b == 0) lds_store_f32_opaque<0>(sa, r);
            else if (b == 1) lds_store_f32_opaque<256>(sa, r);
            else if (b == 2) lds_store_f32_opaque<512>(sa, r);
            else lds_store_f32_opaque<768>(sa, r);
          } else {
            st(out + (cg + ch) * PHW + bin, r);   // 16-bit outputs: sub-dword LDS writes cost more than the stores save (measured)
          }
          // 14 x 14: one bin's 16 taps at a time — with the four bins of a lane interleaved the kernel needs > 128 VGPRs and
          // loses the fourth wave per SIMD the LDS footprint allows
          if constexpr (NB > 1) __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (kStaged) {
        if (++staged == SC) {
          flush(cg + ch + 1 - SC, SC);
          staged = 0;
        }
      }
    }
  };
  const int npass = (cc + G - 1) / G;
  for (int p = 0; p < npass; ++p) {
    const int cg = p * G;
    const int gc = min(G, cc - cg);
    if (kDouble) {
      if (p + 1 < npass) {
        dma_issue_pass<T, NRG, kDmaPerPass>(in0 + (int64_t)(cg + G) * plane_sz, plane_sz, min(G, cc - cg - G), goff,
                               bytes + ((p + 1) & 1) * (kDmaBuf * 4));
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * NRG) : "memory");  // everything older than the DMAs just issued has landed
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if constexpr (kPreAddr) {
        if (p & 1) run_channels(std::integral_constant<int, NBUF - 1>{}, cg, gc, 0u);
        else run_channels(std::integral_constant<int, 0>{}, cg, gc, 0u);
      } else {
        run_channels(std::integral_constant<int, 0>{}, cg, gc, (unsigned)((p & 1) * (kDmaBuf * 4)));
      }
    } else {
      dma_issue_pass<T, NRG, kDmaPerPass>(in0 + (int64_t)cg * plane_sz, plane_sz, gc, goff, bytes);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      run_channels(std::integral_constant<int, 0>{}, cg, gc, 0u);
    }
    // the ds_reads above are complete (their results were consumed) before the next DMAs may
    // overwrite this buffer
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (kStaged && staged) flush(cc - staged, staged);
}

// Returns true when the unit was left to the caller (declined == nullptr: the inline form, round 6 — the caller runs the wave
// path for it in the same launch instead of a worklist entry for a mop-up launch).
template <typename T, typename R, int PHT, int PWT, int SRT>
__device__ __forceinline__ bool roi_align_fwd_wave_dma(DmaShared<PHT>& s, const T* __restrict__ input,
                                                       const R* __restrict__ rois, T* __restrict__ output,
                                                       int C, int H, int W, float spatial_scale, int aligned, int k,
                                                       int c0, int chunk, int* __restrict__ declined, int k_id = -1) {
  constexpr int PHW = PHT * PWT;
  constexpr int NB = (PHW + 63) / 64;
  constexpr int NS = SRT * SRT;
  constexpr int EPP = 16 / (int)sizeof(T);
  constexpr int BLK = kDmaBlkBytes / (int)sizeof(T);
  const int lane = threadIdx.x & 63;
  const int cc = min(chunk, C - c0);
  const RoiGeom<float> g = roi_geom<R, float>(rois + (int64_t)k * 5, spatial_scale, PHT, PWT, SRT, aligned != 0);
  T* out = output + ((int64_t)k * C + c0) * PHW;
  const T* in0 = input + ((int64_t)g.batch * C + c0) * H * W;
  const int64_t plane_sz = (int64_t)H * W;
  const DmaWindow dw = dma_window<PHT, PWT, SRT, EPP>(g, H, W);
  if (dw.state == 2) {  // left to the mop-up launch: one worklist entry per RoI (the wave of channel chunk 0 reports it)
    if (declined == nullptr) return true;
    if (c0 == 0 && lane == 0) declined[kMopHeader + atomicAdd(declined, 1)] = k_id >= 0 ? k_id : k;   // (k_id: `rois` is the row itself)
    return false;
  }
  if (dw.state == 0) {
    for (int o = lane; o < cc * PHW; o += 64) st(out + o, 0.f);
    return false;
  }
  // ---- per-lane sample set-up (registers): LDS slots of the two tap rows + separable factors; run by roi_align_dma_passes
  // once the first DMAs are on their way
  auto setup = [&](int (&off)[NB][NS][2], float (&fy)[NB][SRT][2], float (&fx)[NB][SRT][2]) {
  const int rstride = EPP * dw.lpr;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int bin = min(lane + 64 * b, PHW - 1);
    const int ph = bin / PWT, pw = bin - ph * PWT;
    int rlo[SRT][2], xlo[SRT];
#pragma unroll
    for (int i = 0; i < SRT; ++i) {
      int lo, hi;
      float l, h;
      const bool vy = axis_sample<float>(H, g.start_h, g.bin_h, SRT, ph, i, lo, hi, l, h);   // every sample is valid here (dma_window)
      fy[b][i][0] = l;
      fy[b][i][1] = h;
      const int r = dw.sparse ? 2 * (ph * SRT + i) : (vy ? lo - dw.y0 : 0);
      const int r1 = dw.sparse ? r + 1 : r + (hi - lo);   // bottom edge: the high tap row IS the low one, like the reference
      const int g0 = small_div(r, dw.m_rpi), g1 = small_div(r1, dw.m_rpi);
      rlo[i][0] = g0 * BLK + (r - g0 * dw.rpi) * rstride;
      rlo[i][1] = g1 * BLK + (r1 - g1 * dw.rpi) * rstride;
      const bool vx = axis_sample_shifted(W, g.start_w, g.bin_w, SRT, pw, i, lo, l, h);
      fx[b][i][0] = l;
      fx[b][i][1] = h;
      xlo[i] = vx ? lo - dw.x0 : 0;
    }
#pragma unroll
    for (int iy = 0; iy < SRT; ++iy)
#pragma unroll
      for (int ix = 0; ix < SRT; ++ix) {
        off[b][iy * SRT + ix][0] = rlo[iy][0] + xlo[ix];
        off[b][iy * SRT + ix][1] = rlo[iy][1] + xlo[ix];
      }
  }
  };
  if (dw.nrg <= 1)
    roi_align_dma_passes<T, PHT, PWT, SRT, 1>(s, in0, out, plane_sz, cc, H, W, dw, g, setup);
  else if (dw.nrg <= 2)
    roi_align_dma_passes<T, PHT, PWT, SRT, 2>(s, in0, out, plane_sz, cc, H, W, dw, g, setup);
  else if (dw.nrg <= 3)   // 28 sampled rows of <= 5 pieces: three DMA instructions per channel, not four
    roi_align_dma_passes<T, PHT, PWT, SRT, 3>(s, in0, out, plane_sz, cc, H, W, dw, g, setup);
  else if (dw.nrg <= 4)
    roi_align_dma_passes<T, PHT, PWT, SRT, 4>(s, in0, out, plane_sz, cc, H, W, dw, g, setup);
  else
    roi_align_dma_passes<T, PHT, PWT, SRT, 2 * dma_per_pass(PHT)>(s, in0, out, plane_sz, cc, H, W, dw, g, setup);
  return false;
}

template <typename T, typename R, int PHT, int PWT, int SRT>
__device__ __forceinline__ void roi_align_wave_dispatch(WaveShared& s, const T* __restrict__ input,
                                                        const R* __restrict__ rois, T* __restrict__ output, int C,
                                                        int H, int W, int PH_, int PW_, float spatial_scale, int sr_,
                                                        int aligned, int k, int c0, int chunk) {
  if constexpr (PHT > 0 && PWT > 0 && SRT > 0) {
    roi_align_fwd_wave_fast<T, R, PHT, PWT, SRT>(s, input, rois, output, C, H, W, spatial_scale, aligned, k, c0, chunk);
  }
  else
    roi_align_fwd_wave_unit<T, R, PHT, PWT, SRT>(s, input, rois, output, C, H, W, PH_, PW_, spatial_scale, sr_, aligned,
                                              k, c0, chunk);
}

// XCD-aware work placement.  Workgroup b is observed to run on XCD b % 8 (a speed assumption only — nothing depends on
// it for correctness).  A unit is (RoI, channel chunk); two placements:
//   pinned (nchunks a multiple of 8): XCD x owns the channel chunks {x, x + 8, ...} of EVERY RoI and walks the RoIs in
//     `perm` order (roi_fwd_order: image, level, window rows — neighbours in that order overlap in the map and run at
//     about the same time).  A byte of a feature map is then wanted by ONE private 4 MiB L2 only, and the slice that L2 is
//     asked for at any moment is a channel chunk of a row band of one plane — the overlap between RoI windows turns into
//     L2 hits instead of a second / third fetch by another XCD (the RoIPool forward does the same, roi_pool.hip).
//   ranges (any chunk count): XCD x gets a contiguous range of RoIs and walks it chunk by chunk.
struct UnitMap {
  const int* perm;  // nullptr: identity
  int pinned;
};

// Position -> RoI index and the [5] row of a float RoI list as SCALAR loads (wave-uniform addresses, data of the launch in front):
// a vector load of a unit's first instructions queues behind the window DMAs of the CU's other waves (~1 us under these
// kernels' own traffic) — twice in a row for position -> index -> row; the scalar cache has its own path to L2.
__device__ __forceinline__ int perm_at(const int* __restrict__ perm, int64_t kk) {
  int k;
  const int* pp = perm + kk;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(k) : "s"(pp) : "memory");
  return k;
}
// (64-bit pairs, not one dwordx4: element extracts of a 4-vector inline-asm result in SGPRs all came back as element 0)
__device__ __forceinline__ void roi_row_scalar(const float* __restrict__ rp, float (&row)[5]) {
  unsigned long long r01, r23;
  int r4;
  asm volatile("s_load_dwordx2 %0, %3, 0x0\n\ts_load_dwordx2 %1, %3, 0x8\n\ts_load_dword %2, %3, 0x10\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(r01), "=&s"(r23), "=&s"(r4) : "s"(rp) : "memory");
  row[0] = __builtin_bit_cast(float, (unsigned)r01);
  row[1] = __builtin_bit_cast(float, (unsigned)(r01 >> 32));
  row[2] = __builtin_bit_cast(float, (unsigned)r23);
  row[3] = __builtin_bit_cast(float, (unsigned)(r23 >> 32));
  row[4] = __builtin_bit_cast(float, r4);
}

// (`block`: the workgroup's number within the units' grid — blockIdx.x, or blockIdx.x less the workgroups of another job that the
// launch carries in front, a multiple of 8 so that the XCD of a unit stays what it was)
__device__ __forceinline__ bool wave_unit_at(unsigned block, int64_t K, int nchunks, const UnitMap& um, int& k, int& chunk_idx,
                                             int wpb = kThreads / 64) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: keep unit math scalar
  const int xcd = block & 7;
  const int64_t j = block >> 3;
  const int64_t local = j * wpb + wave;
  if (um.pinned) {
    const int slot = (int)(local / K);
    chunk_idx = xcd + 8 * slot;
    if (chunk_idx >= nchunks) return false;
    const int64_t kk = local - (int64_t)slot * K;
    k = um.perm ? perm_at(um.perm, kk) : (int)kk;
    return true;
  }
  const int64_t kbase = K >> 3, krem = K & 7;
  const int64_t Kx = kbase + (xcd < krem ? 1 : 0);
  const int64_t kstart = xcd * kbase + (xcd < krem ? xcd : krem);
  if (local >= Kx * nchunks) return false;
  chunk_idx = (int)(local / Kx);
  const int64_t kk = kstart + (local - (int64_t)chunk_idx * Kx);
  k = um.perm ? perm_at(um.perm, kk) : (int)kk;
  return true;
}

__device__ __forceinline__ bool wave_unit(int64_t K, int nchunks, const UnitMap& um, int& k, int& chunk_idx,
                                          int wpb = kThreads / 64) {
  return wave_unit_at(blockIdx.x, K, nchunks, um, k, chunk_idx, wpb);
}

__host__ __device__ inline bool unit_map_can_pin(int nchunks) { return nchunks >= 8 && nchunks % 8 == 0; }

// Units of the channels_last kernel: (RoI, group of 64 lanes' channels).  XCD x serves the x-th EIGHTH OF THE RoIs IN LAUNCH ORDER
// (sorted by image, level, window-top band: the eighths are contiguous runs of that order, i.e. different images / levels /
// bands) with all their channel groups, group-major: an XCD's L2 sees one region of the maps, neighbouring windows one after
// the other.  Round 4 split the RoIs in INPUT order (every XCD touched every map: 1.77 x the algorithmic bytes fetched, now
// 1.05 x).  Measured and rejected on the way: one channel group per XCD pair (the NCHW kernel's pinning) — a group is a 256-byte
// slice of every 1 KB pixel, so an XCD then reads through a quarter of its L2 channels: same traffic, 0.21 instead of 0.18 ms.
__device__ __forceinline__ bool wave_unit_nhwc(int64_t K, int ngroups, const int* __restrict__ perm, int& k, int& gi) {
  return wave_unit(K, ngroups, UnitMap{perm, 0}, k, gi);
}
inline unsigned wave_unit_grid(int64_t K, int nchunks, bool pinned = false, int wpb = kThreads / 64) {
  const int64_t per_xcd = pinned ? ceil_div(K * (nchunks / 8), wpb) : ceil_div(ceil_div(K, 8) * nchunks, wpb);
  return (unsigned)(8 * per_xcd);
}
inline unsigned wave_unit_grid_nhwc(int64_t K, int ngroups, bool) { return wave_unit_grid(K, ngroups); }

// Launch order of the RoIs (forward, pinned placement): a one-workgroup counting sort by (image, level, window-top band).
// Only the ORDER in which units start depends on it — never a result — so the key uses the hardware's approximate
// log2 / reciprocal (the kernels that compute results evaluate the reference's level formula exactly, roi_common.h).
// Keys and in-bucket ranks stay in registers between the histogram and the scatter (kOrderPerThread RoIs per thread up to
// 4096 RoIs; beyond that the key is re-evaluated).  Also clears the mop-up worklist counter of the launch that follows.
// Buckets: N * L * bands <= kOrderBuckets.
constexpr int kOrderThreads = 1024;
constexpr int kOrderBuckets = 4096;
constexpr int kOrderPerThread = 4;
constexpr int64_t kOrderMaxRois = 1 << 16;   // beyond that one workgroup is the wrong shape: identity order
constexpr int kOrderMaxFold = 4096;          // RoIs the 256-thread sort inside a launch keeps in registers (16 per thread)

// Round 6: the pre-pass can also BUILD the [K,5] RoI rows from the per-image box lists (convert_boxes_to_roi_format,
// ops/_utils.py:18-25 — a launch of its own until now: tvmi::boxes_to_rois, 4.8 us + a launch gap in front of every
// MultiScaleRoIAlign call): it touches every box anyway.  BOXES: row k comes from (image of k, box k - first of that image) and is
// written to `rois_out`, which the kernels of the launch that follows read.
constexpr int kOrderMaxImages = 64;
struct MsBoxLists {
  const float* ptr[kOrderMaxImages];
  int end[kOrderMaxImages];   // exclusive prefix end of every image's boxes in the concatenation
  int n;
};

// Row k of the concatenated box lists as a [K,5] RoI row (convert_boxes_to_roi_format).
__device__ __forceinline__ void row_from_boxes(const MsBoxLists& bl, int k, float (&v)[5]) {
  int img = 0;
  for (int i = 0; i < bl.n - 1; ++i) img += k >= bl.end[i] ? 1 : 0;
  int first = 0;
  const float* base = bl.ptr[0];
#pragma unroll 1
  for (int i = 1; i < bl.n; ++i)       // run-time index into the by-value struct would go through scratch: walk it
    if (i == img) {
      first = bl.end[i - 1];
      base = bl.ptr[i];
    }
  const float* b = base + (int64_t)(k - first) * 4;
  v[0] = (float)img;
  v[1] = b[0];
  v[2] = b[1];
  v[3] = b[2];
  v[4] = b[3];
}

struct OrderShared {
  int hist[kOrderBuckets];
  int wsum[16];
  float band_scale[kMaxLevels];   // window-top row (level pixels) -> band index
  float lvl_scale[kMaxLevels];
};

// The counting sort as a workgroup body of NT threads (NT * PER >= K for the register form): the launch of its own
// (roi_fwd_order, 1024 threads) and, round 6, ONE WORKGROUP INSIDE the 7 x 7 multi-scale launch (256 threads, "fold": see
// FoldArgs).  n0: the first n0 RoIs keep their input position (they are not counted) and positions [n0, K) are sorted.
// perm64 != nullptr: positions are handed to the other workgroups of the SAME launch as self-validating 64-bit entries
// (epoch << 32 | RoI index, agent-scope stores): a reader polls ITS entry until it carries the launch's epoch.
template <int NT, int PER, typename R, bool BOXES, int BATCH = 4>
__device__ __forceinline__ void order_body(OrderShared& sh, const MsLevels& lv, const R* __restrict__ rois, int K, int N,
                                           int multiscale, int bands, int* __restrict__ perm, int* __restrict__ mop_counter,
                                           const MsBoxLists& bl, float* __restrict__ rois_out, int n0,
                                           unsigned long long* __restrict__ perm64, int epoch) {
  static_assert(NT % 64 == 0 && NT / 64 <= 16 && kOrderBuckets % NT == 0 && PER % BATCH == 0, "order workgroup shape");
  // inside a launch whose other waves saturate the CU's memory path and issue slots, the one workgroup every later unit waits
  // for goes first (measured without: the sort took ~37 us instead of 10 and the launch grew by what the pre-pass launch had cost)
  if (perm64) __builtin_amdgcn_s_setprio(3);
  constexpr int BPT = kOrderBuckets / NT;   // buckets per thread in the scan
  int* hist = sh.hist;
  const int tid = threadIdx.x;
  const int L = multiscale ? lv.n_levels : 1;
  const int nb = N * L * bands;
  for (int i = tid; i < nb; i += NT) hist[i] = 0;
  if (tid < 16) sh.wsum[tid] = 0;
  if (tid < kMaxLevels) {
    float sc = lv.scale[0];
    int Hl = lv.H[0];
#pragma unroll
    for (int i = 1; i < kMaxLevels; ++i)   // constant indices: a run-time index would put the by-value struct into scratch
      if (i == tid) {
        sc = lv.scale[i];
        Hl = lv.H[i];
      }
    sh.lvl_scale[tid] = sc;
    sh.band_scale[tid] = sc * (float)bands * __builtin_amdgcn_rcpf((float)max(Hl, 1));
  }
  if (tid == 0 && mop_counter) *mop_counter = 0;
  __syncthreads();
  const float inv_s0 = __builtin_amdgcn_rcpf(lv.s0);
  auto key_of = [&](float bf, float x1, float y1, float x2, float y2) {
    int l = 0;
    if (multiscale) {
      // floor(lvl0 + log2(sqrt(area) / s0) + eps), approximately: v_log_f32 of the area, halved
      float t = floorf(lv.lvl0 + 0.5f * __builtin_amdgcn_logf((x2 - x1) * (y2 - y1) * inv_s0 * inv_s0) + lv.eps);
      t = fminf(fmaxf(t, (float)lv.k_min), (float)lv.k_max);
      l = (t == t) ? min(max((int)t - lv.k_min, 0), L - 1) : 0;
    }
    const int b = (bf == bf) ? min(max((int)bf, 0), N - 1) : 0;
    const float yb = y1 * sh.band_scale[l];
    const int band = (yb == yb) ? min(max((int)yb, 0), bands - 1) : 0;
    return (b * L + l) * bands + band;
  };
  auto load_row = [&](int k, float (&v)[5], bool write) {
    if constexpr (BOXES) {
      row_from_boxes(bl, k, v);
      // (inside a launch — perm64 — the rows are written by the units themselves: every vector memory instruction of this
      // workgroup waits a full turn of the CU's saturated texture path, and 5 row stores per RoI were 80 of its 110 per wave:
      // the sort took ~40 us instead of 10)
      if (write && !perm64) {
        float* r = rois_out + (int64_t)k * 5;
#pragma unroll
        for (int e = 0; e < 5; ++e) r[e] = v[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 5; ++e) v[e] = (float)ld(rois + (int64_t)k * 5 + e);
    }
  };
  auto load_key = [&](int k, bool write) {
    float v[5];
    load_row(k, v, write);
    return key_of(v[0], v[1], v[2], v[3], v[4]);
  };
  auto put = [&](int pos, int k) {
    if (perm64) st_agent64(&perm64[pos], ((unsigned long long)(unsigned)epoch << 32) | (unsigned)k);
    else perm[pos] = k;
  };
  const bool in_regs = K <= NT * PER;
  int key[PER], rank[PER];
  if (in_regs) {
#pragma unroll
    for (int j0 = 0; j0 < PER; j0 += BATCH) {         // BATCH rows of the thread in flight together
      float v[BATCH][5];
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int k = min(tid + (j0 + j) * NT, K - 1);
        load_row(k, v[j], tid + (j0 + j) * NT < K);
      }
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int k = tid + (j0 + j) * NT;
        key[j0 + j] = key_of(v[j][0], v[j][1], v[j][2], v[j][3], v[j][4]);
        rank[j0 + j] = (k < K && k >= n0) ? atomicAdd(&hist[key[j0 + j]], 1) : 0;
      }
    }
  } else {
    for (int k = tid; k < K; k += NT) {
      const int ky = load_key(k, true);
      if (k >= n0) atomicAdd(&hist[ky], 1);
    }
  }
  __syncthreads();
  // exclusive scan of the bucket counts: BPT buckets per thread, wave scan, wave totals
  int c[BPT], tsum = 0;
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int idx = tid * BPT + i;
    c[i] = idx < nb ? hist[idx] : 0;
    tsum += c[i];
  }
  int incl = tsum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if ((tid & 63) >= d) incl += o;
  }
  if ((tid & 63) == 63) sh.wsum[tid >> 6] = incl;
  __syncthreads();
  int wbase = 0;
  {
    const int w = tid >> 6;
    int part = (tid & 63) < w ? sh.wsum[tid & 15] : 0;   // the totals of the waves in front: a butterfly instead of a serial walk
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m);
    wbase = part;
  }
  int run = n0 + wbase + incl - tsum;
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int idx = tid * BPT + i;
    if (idx < nb) hist[idx] = run;
    run += c[i];
  }
  __syncthreads();
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int k = tid + j * NT;
      if (k < K) put(k >= n0 ? hist[key[j]] + rank[j] : k, k);
    }
  } else {
    for (int k = tid; k < K; k += NT) put(k >= n0 ? atomicAdd(&hist[load_key(k, false)], 1) : k, k);
  }
}

template <typename R, bool BOXES = false>
__global__ __launch_bounds__(kOrderThreads) void roi_fwd_order(MsLevels lv, const R* __restrict__ rois, int K, int N,
                                                                int multiscale, int bands, int* __restrict__ perm,
                                                                int* __restrict__ mop_counter, MsBoxLists bl = MsBoxLists{},
                                                                float* __restrict__ rois_out = nullptr) {
  __shared__ OrderShared sh;
  order_body<kOrderThreads, kOrderPerThread, R, BOXES>(sh, lv, rois, K, N, multiscale, bands, perm, mop_counter, bl, rois_out, 0,
                                                       nullptr, 0);
}

// Mop-up launch: a FIXED small grid whose waves walk the worklist the DMA launch filled (RoI x channel chunk units).
// With an empty list — the usual case — every wave reads one word and leaves.
constexpr int kMopBlocks = 512;
__device__ __forceinline__ bool mop_unit(const int* __restrict__ mop, int nchunks, int64_t& u, int& k, int& ci) {
  const int n = __builtin_amdgcn_readfirstlane(mop[0]);
  if (u >= (int64_t)n * nchunks) return false;
  const int e = (int)(u / nchunks);
  ci = (int)(u - (int64_t)e * nchunks);
  k = __builtin_amdgcn_readfirstlane(mop[kMopHeader + e]);
  return true;
}

// (Measured on the ISA, round 5: asking for four waves per SIMD with __launch_bounds__(kThreads, 4) — the 7 x 7 / generic forms
// need 121-129 VGPRs — makes the scheduler serialise the staging loads of some instantiations again to stay under 128.  Left to
// the register allocator: three or four waves, every staging load asynchronous; tests/test_isa_guards.py.)
template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_wave(const T* __restrict__ input, const T* __restrict__ rois,
                                                               T* __restrict__ output, int C, int H, int W, int PH_,
                                                               int PW_, float spatial_scale, int sr_, int aligned,
                                                               int nchunks, int chunk, int64_t nunits,
                                                               const int* __restrict__ mop) {
  __shared__ WaveShared s[kThreads / 64];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: keep unit math scalar
  int k, ci;
  if (mop) {   // mop-up form: fixed grid, the waves stride over the worklist
    for (int64_t u = (int64_t)blockIdx.x * (kThreads / 64) + wave; mop_unit(mop, nchunks, u, k, ci); u += (int64_t)gridDim.x * (kThreads / 64))
      roi_align_wave_dispatch<T, T, PHT, PWT, SRT>(s[wave], input, rois, output, C, H, W, PH_, PW_, spatial_scale, sr_,
                                                   aligned, k, ci * chunk, chunk);
    return;
  }
  if (!wave_unit(nunits / nchunks, nchunks, UnitMap{nullptr, 0}, k, ci)) return;
  roi_align_wave_dispatch<T, T, PHT, PWT, SRT>(s[wave], input, rois, output, C, H, W, PH_, PW_, spatial_scale, sr_,
                                               aligned, k, ci * chunk, chunk);
}

template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_dma(const T* __restrict__ input,
                                                              const T* __restrict__ rois, T* __restrict__ output,
                                                              int C, int H, int W, float spatial_scale, int aligned,
                                                              int nchunks, int chunk, int64_t nunits, int* __restrict__ mop,
                                                              UnitMap um) {
  __shared__ DmaShared<PHT> s[kThreads / 64];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: keep unit math scalar
  int k, ci;
  if (!wave_unit(nunits / nchunks, nchunks, um, k, ci)) return;
  roi_align_fwd_wave_dma<T, T, PHT, PWT, SRT>(s[wave], input, rois, output, C, H, W, spatial_scale, aligned, k, ci * chunk,
                                              chunk, mop);
}

// Multi-scale entries: RoIs are float32 image coordinates whatever the feature dtype (roi_common.h).
template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_ms_dma(MsLevels lv, const float* __restrict__ rois,
                                                                 T* __restrict__ output, int C, int aligned,
                                                                 int nchunks, int chunk, int64_t nunits, int* __restrict__ mop,
                                                                 UnitMap um) {
  __shared__ DmaShared<PHT> s[kThreads / 64];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: keep unit math scalar
  int k, ci;
  if (!wave_unit(nunits / nchunks, nchunks, um, k, ci)) return;
  float row[5];
  roi_row_scalar(rois + (int64_t)k * 5, row);
  const int l = fpn_level<float>(row, lv);
  roi_align_fwd_wave_dma<T, float, PHT, PWT, SRT>(s[wave], static_cast<const T*>(lv.ptr[l]), row, output + (int64_t)k * C * (PHT * PWT), C,
                                                  lv.H[l], lv.W[l], lv.scale[l], aligned, 0, ci * chunk, chunk, mop, k);
}

// Round 6, "fold": the order pre-pass INSIDE the launch (roi_align.fold_order, calls that start from box lists).  One workgroup in
// front of the units' grid runs the counting sort (order_body, 256 threads) while the units of the launch's FIRST ROUND — positions
// below n0 = what the chip holds at once — take the RoIs in input order straight from the box lists; a unit at a later position
// polls ITS 64-bit entry of the order (written long before the unit is dispatched: the sort takes ~10 us, a unit ~25) with an
// agent-scope load until the entry carries this launch's epoch, and takes its RoI index from the same word: no flag word, no
// second dependent load, nothing to reset — the entries live in a block only these sort workgroups write (FoldBlocks, per stream)
// and the block's epoch only grows, so an entry an earlier launch left there cannot be mistaken for this launch's.  Saves the pre-pass launch and the gap behind it on the critical path
// of every detector step for ~1/8 of the units in input order.  Not used under graph capture (the epoch would be frozen into the
// node and a replay could read the entries of the replay before it): the two-launch form runs there.
// Only the ORDER of the units depends on any of this — never a result.
struct FoldArgs {
  unsigned long long* perm64;   // nullptr: no fold — positions come from um.perm (the pre-pass launch), rows from `rois`
  int epoch, n0, bands, N;
  float* rois_out;
  MsBoxLists bl;
};
constexpr unsigned kFoldBlocks = 8;   // workgroups in front of the units (one sorts, seven leave): a multiple of 8 keeps the units' XCDs

template <int PHT>
union MsWaveLds {
  DmaShared<PHT> d;
  WaveShared w;
};

template <typename T, int PHT, int PWT, int SRT>
__device__ __forceinline__ void ms_inl_unit(MsWaveLds<PHT>& s, const MsLevels& lv, const float* __restrict__ rois,
                                            T* __restrict__ output, int C, int aligned, int nchunks, int chunk, int64_t nunits,
                                            const UnitMap& um, const FoldArgs& fa, unsigned block, int wpb = kThreads / 64) {
  constexpr int PHW = PHT * PWT;
  int k, ci;
  float row[5];
  if (fa.perm64) {
    int kk;
    if (!wave_unit_at(block, nunits / nchunks, nchunks, UnitMap{nullptr, 1}, kk, ci, wpb)) return;
    k = kk;
    if (kk >= fa.n0) {
      // fast path: a SCALAR load (the vector path queues behind the CU's window DMAs, ~1 us).  An entry that carries this launch's
      // epoch IS this launch's entry whatever cache it came from (epochs are never reused); anything else — not written yet, or a
      // line some cache took before it was written — falls back to the agent-scope poll, which no cache can satisfy
      unsigned long long e;
      const unsigned long long* ep = fa.perm64 + kk;
      asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(e) : "s"(ep) : "memory");
      if ((int)(e >> 32) != fa.epoch) e = ld_agent64(fa.perm64 + kk);
      for (int polls = 0; (int)(e >> 32) != fa.epoch; ++polls) {
        __builtin_amdgcn_s_sleep(8);
        if (polls > (1 << 22)) __builtin_trap();   // the sort workgroup is resident before any unit: seconds of polling mean a broken launch
        e = ld_agent64(fa.perm64 + kk);
      }
      k = __builtin_amdgcn_readfirstlane((int)(unsigned)e);
    }
    {   // the wave-uniform form of row_from_boxes: the box as one scalar load
      int img = 0;
      for (int i = 0; i < fa.bl.n - 1; ++i) img += k >= fa.bl.end[i] ? 1 : 0;
      int first = 0;
      const float* base = fa.bl.ptr[0];
#pragma unroll 1
      for (int i = 1; i < fa.bl.n; ++i)
        if (i == img) {
          first = fa.bl.end[i - 1];
          base = fa.bl.ptr[i];
        }
      const float* bp = base + (int64_t)(k - first) * 4;
      unsigned long long b01, b23;
      asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %2, 0x8\n\ts_waitcnt lgkmcnt(0)" : "=&s"(b01), "=&s"(b23) : "s"(bp) : "memory");
      row[0] = (float)img;
      row[1] = __builtin_bit_cast(float, (unsigned)b01);
      row[2] = __builtin_bit_cast(float, (unsigned)(b01 >> 32));
      row[3] = __builtin_bit_cast(float, (unsigned)b23);
      row[4] = __builtin_bit_cast(float, (unsigned)(b23 >> 32));
      if (ci == 0) {   // the [K,5] row the caller asked for (the backward reads it): one store by the unit of channel chunk 0
        const int lane = threadIdx.x & 63;
        const float v = lane == 0 ? row[0] : lane == 1 ? row[1] : lane == 2 ? row[2] : lane == 3 ? row[3] : row[4];
        if (lane < 5) fa.rois_out[(int64_t)k * 5 + lane] = v;
      }
    }
  } else {
    if (!wave_unit_at(block, nunits / nchunks, nchunks, um, k, ci, wpb)) return;
    roi_row_scalar(rois + (int64_t)k * 5, row);
  }
  const int l = fpn_level<float>(row, lv);
  T* outk = output + (int64_t)k * C * PHW;
  if (roi_align_fwd_wave_dma<T, float, PHT, PWT, SRT>(s.d, static_cast<const T*>(lv.ptr[l]), row, outk, C, lv.H[l], lv.W[l],
                                                      lv.scale[l], aligned, 0, ci * chunk, chunk, nullptr))
    roi_align_fwd_wave_fast<T, float, PHT, PWT, SRT>(s.w, static_cast<const T*>(lv.ptr[l]), row, outk, C, lv.H[l], lv.W[l],
                                                     lv.scale[l], aligned, 0, ci * chunk, chunk);
}

// The same launch WITHOUT a mop-up launch behind it (round 6, "roi_align.inline_mop"): a unit the DMA path declines (a window
// above 8 DMA blocks per channel, samples the reference skips: 0-2 % of the RoIs of a detector step) takes the wave path right
// here, in the LDS of the same wave (the two per-wave images are a union: 10 KB, still four workgroups per CU).  One launch and
// one launch gap less per call; the price is the wave path's registers in this kernel's allocation (127 instead of 73: still four
// waves per SIMD).
template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_ms_dma_inl(MsLevels lv, const float* __restrict__ rois,
                                                                     T* __restrict__ output, int C, int aligned, int nchunks,
                                                                     int chunk, int64_t nunits, UnitMap um, FoldArgs fa) {
  union Lds {
    MsWaveLds<PHT> s[kThreads / 64];
    OrderShared order;
  };
  __shared__ Lds sh;
  unsigned block = blockIdx.x;
  if (fa.perm64) {
    if (block < kFoldBlocks) {
      if (block == 0)
        order_body<kThreads, kOrderMaxFold / kThreads, float, true, 8>(sh.order, lv, nullptr, (int)(nunits / nchunks), fa.N, 1, fa.bands,
                                                                    nullptr, nullptr, fa.bl, fa.rois_out, fa.n0, fa.perm64, fa.epoch);
      return;
    }
    block -= kFoldBlocks;
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  ms_inl_unit<T, PHT, PWT, SRT>(sh.s[wave], lv, rois, output, C, aligned, nchunks, chunk, nunits, um, fa, block);
}

// (Measured and removed, round 6: the same units as ONE-WAVE workgroups — a slot refilled when its own unit ends instead of when
// the slowest of four ends — 0.2494 against 0.2452 ms for the op on one box: the coupling of the four waves costs less than four
// times the dispatches.)
// The detector step in ONE launch (round 6): the workgroups of the step's NMS (nms_step_body, nms_step_device.h: tiles, rank
// counting, sweeps, keep list and payload of <= 4096 boxes) ride in front of the RoIAlign grid.  As two launches the two jobs need
// two streams to overlap, and the fork, the join and the second queue cost the step ~40 us more than the RoIAlign launch takes on
// its own (bench.py: 0.265 ms against 0.220); in one grid the NMS workgroups are simply the first to be dispatched, finish within
// the first fifth of the launch and give their slots to RoIAlign units.  `nms_blocks` is padded to a multiple of 8 (the units'
// XCD placement keys on the workgroup number modulo 8).  The LDS of the two jobs is one union (40 KB: four workgroups per CU as
// before), the register allocation the larger of the two.
template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void roi_align_fwd_ms_dma_inl_step(MsLevels lv, const float* __restrict__ rois,
                                                                          T* __restrict__ output, int C, int aligned, int nchunks,
                                                                          int chunk, int64_t nunits, UnitMap um, StepArgs na,
                                                                          unsigned nms_blocks, FoldArgs fa) {
  static_assert(kThreads == kStepThreads, "one workgroup shape for both jobs");
  union Lds {
    MsWaveLds<PHT> s[kThreads / 64];
    StepShared nms;
    OrderShared order;
  };
  __shared__ Lds sh;
  if (blockIdx.x < nms_blocks) {
    const int id = (int)blockIdx.x;
    if (id < na.S * na.gdim_y)
      nms_step_body(sh.nms, id % na.S, id / na.S, na.gdim_y, na.dets, na.scores, na.seg, na.n, na.S, na.G, na.thr, na.band, na.ws,
                    na.keep_out, na.num_keep, na.pk);
    return;
  }
  unsigned block = blockIdx.x - nms_blocks;
  if (fa.perm64) {
    if (block < kFoldBlocks) {
      if (block == 0)
        order_body<kThreads, kOrderMaxFold / kThreads, float, true, 8>(sh.order, lv, nullptr, (int)(nunits / nchunks), fa.N, 1, fa.bands,
                                                                    nullptr, nullptr, fa.bl, fa.rois_out, fa.n0, fa.perm64, fa.epoch);
      return;
    }
    block -= kFoldBlocks;
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  ms_inl_unit<T, PHT, PWT, SRT>(sh.s[wave], lv, rois, output, C, aligned, nchunks, chunk, nunits, um, fa, block);
}

template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_ms_wave(MsLevels lv, const float* __restrict__ rois,
                                                                  T* __restrict__ output, int C, int PH_, int PW_,
                                                                  int sr_, int aligned, int nchunks, int chunk,
                                                                  int64_t nunits, const int* __restrict__ mop) {
  __shared__ WaveShared s[kThreads / 64];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: keep unit math scalar
  int k, ci;
  if (mop) {
    for (int64_t u = (int64_t)blockIdx.x * (kThreads / 64) + wave; mop_unit(mop, nchunks, u, k, ci); u += (int64_t)gridDim.x * (kThreads / 64)) {
      const int l = fpn_level<float>(rois + (int64_t)k * 5, lv);
      roi_align_wave_dispatch<T, float, PHT, PWT, SRT>(s[wave], static_cast<const T*>(lv.ptr[l]), rois, output, C, lv.H[l],
                                                       lv.W[l], PH_, PW_, lv.scale[l], sr_, aligned, k, ci * chunk, chunk);
    }
    return;
  }
  if (!wave_unit(nunits / nchunks, nchunks, UnitMap{nullptr, 0}, k, ci)) return;
  const int l = fpn_level<float>(rois + (int64_t)k * 5, lv);
  roi_align_wave_dispatch<T, float, PHT, PWT, SRT>(s[wave], static_cast<const T*>(lv.ptr[l]), rois, output, C, lv.H[l],
                                                   lv.W[l], PH_, PW_, lv.scale[l], sr_, aligned, k, ci * chunk, chunk);
}

// Workspace of the forward entries: [mop-up worklist: kMopHeader + K ints, padded to 16 bytes][perm: K ints].  A caller
// that passes only the first part gets the identity launch order; no workspace at all: the register-staged wave kernels alone.
inline size_t fwd_mop_bytes(int64_t K) { return ((size_t)(K + kMopHeader) * sizeof(int) + 15) & ~(size_t)15; }
inline int* mop_part(void* ws, size_t ws_bytes, int64_t K) {
  return (ws && K > 0 && K < (1 << 30) && ws_bytes >= fwd_mop_bytes(K)) ? static_cast<int*>(ws) : nullptr;
}
inline int* perm_part(void* ws, size_t ws_bytes, int64_t K) {
  if (!ws || K <= 0 || K > kOrderMaxRois || ws_bytes < fwd_mop_bytes(K) + (size_t)K * sizeof(int)) return nullptr;
  return reinterpret_cast<int*>(static_cast<char*>(ws) + fwd_mop_bytes(K));
}

// Process-wide switches of the forward (tvmi_set_option): measured defaults, the other settings stay reachable for the
// variant table of HISTORY.md §4.1 and for tests that force every route.
struct FwdOptions {
  int pin_chunks = 1;   // "roi_align.pin_chunks": channel chunks pinned to XCDs when the chunk count allows it
  int order = 1;        // "roi_align.order": launch order from roi_fwd_order (needs the pinned placement + workspace)
  int bands = 16;       // "roi_align.order_bands": window-top bands per (image, level) in the order key
  int carry_step = 1;   // "roi_align.carry_step": tvmi_multiscale_roi_align_forward_boxes_with_nms_step puts the NMS workgroups into the RoIAlign launch
  int fold_pct = 100;   // "roi_align.fold_first_round_pct": positions left in input order, in percent of what the chip holds at once
  int fold_order = 1;   // "roi_align.fold_order": the order pre-pass as a workgroup of the 7 x 7 multi-scale launch (calls from box lists)
  int inline_mop = 1;   // "roi_align.inline_mop": declined units take the wave path inside the DMA launch (multi-scale 7 x 7 entries)
};
FwdOptions g_fwd_opt;

int fill_levels(MsLevels& lv, const void* const* ptrs, const int64_t* heights, const int64_t* widths, const double* scales,
                int64_t n_levels, int64_t k_min, int64_t k_max, double s0, double lvl0, double eps) {
  for (int i = 0; i < kMaxLevels; ++i) {
    const int j = i < n_levels ? i : 0;
    lv.ptr[i] = ptrs[j];
    lv.H[i] = (int)heights[j];
    lv.W[i] = (int)widths[j];
    lv.scale[i] = (float)scales[j];
  }
  lv.n_levels = (int)n_levels;
  lv.k_min = (int)k_min;
  lv.k_max = (int)k_max;
  lv.s0 = (float)s0;
  lv.lvl0 = (float)lvl0;
  lv.eps = (float)eps;
  return 0;
}

// Decides placement and order for one launch of the DMA kernels and clears the mop-up worklist counter — in the order
// pre-pass when that runs, with a 4-byte memset otherwise.
// `bl` (optional): the RoI rows do not exist yet — the pre-pass writes them to `rois_out` from the box lists; the caller has
// checked with plan_units_orders() that the pre-pass runs.
inline bool plan_units_orders(int64_t N, int64_t L, int nchunks, const int* perm) {
  return g_fwd_opt.pin_chunks && unit_map_can_pin(nchunks) && g_fwd_opt.order && perm && N >= 1 && N * L <= kOrderBuckets;
}
template <typename R>
int plan_units(UnitMap& um, const MsLevels& lv, const R* rois, int64_t N, int64_t K, int nchunks, int multiscale, int* mop,
               int* perm, hipStream_t stream, const MsBoxLists* bl = nullptr, float* rois_out = nullptr) {
  um = UnitMap{nullptr, 0};
  um.pinned = (g_fwd_opt.pin_chunks && unit_map_can_pin(nchunks)) ? 1 : 0;
  const int64_t L = multiscale ? lv.n_levels : 1;
  if (plan_units_orders(N, L, nchunks, perm)) {
    const int bands = (int)std::max<int64_t>(1, std::min<int64_t>(g_fwd_opt.bands, kOrderBuckets / (N * L)));
    if constexpr (std::is_same<R, float>::value) {
      if (bl) {
        roi_fwd_order<float, true><<<dim3(1), dim3(kOrderThreads), 0, stream>>>(lv, nullptr, (int)K, (int)N, multiscale, bands, perm,
                                                                                 mop, *bl, rois_out);
        um.perm = perm;
        return 0;
      }
    }
    roi_fwd_order<R><<<dim3(1), dim3(kOrderThreads), 0, stream>>>(lv, rois, (int)K, (int)N, multiscale, bands, perm, mop);
    um.perm = perm;
    return 0;
  }
  return (int)hipMemsetAsync(mop, 0, sizeof(int), stream);
}

constexpr int kUnitChunk = 32;  // channels per wave unit
constexpr int kMopChunk = 64;   // channels per unit of the launch that mops up what the DMA kernel declined (almost always empty)

template <typename T>
int launch_fwd(const void* input, const void* rois, void* output, int64_t N, int64_t C, int64_t H,
               int64_t W, int64_t K, int64_t PH, int64_t PW, double scale, int64_t sr, int aligned,
               int* mop, int* perm, hipStream_t stream) {
  const T* in = static_cast<const T*>(input);
  const T* r = static_cast<const T*>(rois);
  T* out = static_cast<T*>(output);
  const int64_t total = K * C * PH * PW;
  if constexpr (std::is_same<T, double>::value) {
    const int64_t blocks = std::min<int64_t>(ceil_div(total, kThreads), 1 << 20);
    roi_align_fwd_generic<T><<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(
        in, r, out, total, (int)C, (int)H, (int)W, (int)PH, (int)PW, scale, (int)sr, aligned);
  } else {
    const float fs = (float)scale;
    const int nchunks = (int)ceil_div(C, kUnitChunk), mop_nchunks = (int)ceil_div(C, kMopChunk);
    const int64_t nunits = K * nchunks, mop_nunits = K * mop_nchunks;
    const bool fast_shape = (PH == 7 && PW == 7 && sr == 2) || (PH == 14 && PW == 14 && sr == 2);
    UnitMap um{nullptr, 0};
    if (fast_shape && mop) {
      MsLevels one;
      const void* ptrs[1] = {input};
      const int64_t hs[1] = {H}, ws_[1] = {W};
      const double sc[1] = {scale};
      fill_levels(one, ptrs, hs, ws_, sc, 1, 0, 0, 1.0, 0.0, 0.0);
      const int st = plan_units<T>(um, one, r, N, K, nchunks, /*multiscale=*/0, mop, perm, stream);
      if (st != 0) return set_error(st, "tvmi_roi_align_forward: clearing the worklist failed");
    }
    const dim3 dma_grid(wave_unit_grid(K, nchunks, um.pinned != 0)), grid(wave_unit_grid(K, nchunks)),
        mop_grid((unsigned)std::min<int64_t>(kMopBlocks, ceil_div(mop_nunits, kThreads / 64))), block(kThreads);
#define TVMI_FWD_DMA(PHT, PWT, SRT)                                                                                 \
  roi_align_fwd_dma<T, PHT, PWT, SRT><<<dma_grid, block, 0, stream>>>(in, r, out, (int)C, (int)H, (int)W, fs, aligned, \
                                                                            nchunks, kUnitChunk, nunits, mop, um)
#define TVMI_FWD(PHT, PWT, SRT)                                                                                   \
  if (mop) {                                                                                                 \
    TVMI_FWD_DMA(PHT, PWT, SRT);                                                                                  \
    roi_align_fwd_wave<T, PHT, PWT, SRT><<<mop_grid, block, 0, stream>>>(in, r, out, (int)C, (int)H, (int)W, (int)PH, \
                                                                         (int)PW, fs, (int)sr, aligned, mop_nchunks, \
                                                                         kMopChunk, mop_nunits, mop);             \
  } else                                                                                                          \
    roi_align_fwd_wave<T, PHT, PWT, SRT><<<grid, block, 0, stream>>>(in, r, out, (int)C, (int)H, (int)W, (int)PH,  \
                                                                     (int)PW, fs, (int)sr, aligned, nchunks,      \
                                                                     kUnitChunk, nunits, nullptr)
    if (PH == 7 && PW == 7 && sr == 2) {
      TVMI_FWD(7, 7, 2);
    } else if (PH == 14 && PW == 14 && sr == 2) {
      TVMI_FWD(14, 14, 2);
    } else {
      roi_align_fwd_wave<T, 0, 0, 0><<<grid, block, 0, stream>>>(in, r, out, (int)C, (int)H, (int)W, (int)PH, (int)PW, fs,
                                                                 (int)sr, aligned, nchunks, kUnitChunk, nunits, nullptr);
    }
#undef TVMI_FWD
#undef TVMI_FWD_DMA
  }
  TVMI_RETURN_LAUNCH_STATUS("tvmi_roi_align_forward");
}

// ---------------------------------------------------------------------------------------
// channels_last (NHWC) forward.  With the channel as the fastest dimension a
// bilinear tap is 64*CPL CONTIGUOUS floats for the lanes of a wave and its address and weight
// are wave-uniform: lane = channel, the tap is one coalesced load off a scalar base, the
// weight is an SGPR operand of the FMA, and there is no LDS window, no per-channel loop and no
// bank conflict anywhere on the read side.  A wave owns (RoI, 64 channels), walks the bin rows
// with PW accumulators per lane and parks each finished row in a [channel][bin] LDS block; that
// block is contiguous in the NCHW output, so it leaves as one linear store stream.  Arithmetic per output is identical to the NCHW kernels (same separable
// factors, same FMA order).  The reference has no such path: cuda/roi_align_kernel.cu:365
// calls input.contiguous(), i.e. pays a full NHWC->NCHW copy of every feature map first.
template <typename T, int PHT, int PWT, int SRT>
struct NhwcShared {
  // [channel][bin] block of one unit (64 lanes x 4 bytes of channels: 64 fp32 or 128 16-bit channels), pitch PH*PW
  // (odd for 7x7: conflict-free both ways)
  T t[64 * (4 / (int)sizeof(T)) * (PHT * PWT)];
};

template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_nhwc(MsLevels lv, const float* __restrict__ rois,
                                                               T* __restrict__ output, int C, int aligned,
                                                               int ngroups, int64_t nunits, const int* __restrict__ perm) {
  constexpr int PHW = PHT * PWT;
  constexpr int NS = SRT * SRT;
  constexpr int CPL = 4 / (int)sizeof(T);  // channels per lane: every tap is one 32-bit load per lane
  constexpr int GC = 64 * CPL;             // channels per unit
  __shared__ NhwcShared<T, PHT, PWT, SRT> sh[kThreads / 64];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  int k, gi;
  if (!wave_unit_nhwc(nunits / ngroups, ngroups, perm, k, gi)) return;
  const int c0 = gi * GC;
  // level / batch index are wave-uniform: say so, the tap base then lives in SGPRs
  float row[5];
  roi_row_scalar(rois + (int64_t)k * 5, row);
  const int l = __builtin_amdgcn_readfirstlane(fpn_level<float>(row, lv));
  const int H = lv.H[l], W = lv.W[l];
  const RoiGeom<float> g = roi_geom<float, float>(row, lv.scale[l], PHT, PWT, SRT, aligned != 0);
  const int batch = __builtin_amdgcn_readfirstlane(g.batch);
  // ---- per-RoI sample tables, one sample per lane: element offset of the low tap + the two factors
  // y: the reference's (low, high) rows (high = low on the bottom edge); x: shifted pairs + a legacy multiply for the
  // zero factor; a skipped sample (outside [-1, dim], roi_align_common.h:60-73) is wave-uniform here and simply not visited
  int yoff = 0, ystep = 0, xoff = 0;
  float yl = 0.f, yh = 0.f, xl = 0.f, xh = 0.f;
  bool yv = false, xv = false;
  const int rowC = W * C;
  if (lane < PHT * SRT) {
    int lo, hi;
    yv = axis_sample<float>(H, g.start_h, g.bin_h, SRT, lane / SRT, lane % SRT, lo, hi, yl, yh);
    yoff = lo * rowC;
    ystep = (hi - lo) * rowC;
  }
  if (lane < PWT * SRT) {
    int lo;
    xv = axis_sample_shifted(W, g.start_w, g.bin_w, SRT, lane / SRT, lane % SRT, lo, xl, xh);
    xoff = lo * C;
  }
  const unsigned long long yvalid = __ballot(yv), xvalid = __ballot(xv);
  // byte offset of this lane's channel word (lanes past the channel count re-read the last word): the only
  // per-lane part of a tap address, everything else is scalar
  const unsigned cl4 = (unsigned)sizeof(T) * (unsigned)min(c0 + lane * CPL, C - CPL);
  const T* nbase = static_cast<const T*>(lv.ptr[l]) + (int64_t)batch * H * W * C;
  constexpr bool kPow2 = (NS & (NS - 1)) == 0;
  const float inv_count = 1.f / (float)NS;
  T* tl = sh[wave].t;
  // x-sample parameters are the same for every bin row: SGPRs for the whole unit
  int sx0[PWT * SRT];
  float shx[PWT * SRT], slx[PWT * SRT];
#pragma unroll
  for (int j = 0; j < PWT * SRT; ++j) {
    sx0[j] = __builtin_amdgcn_readlane(xoff, j);
    shx[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xh), j));
    slx[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xl), j));
  }
  // A RoI with samples outside [-1, dim] (skipped by the reference) takes the branchy form of the loop: every skipped
  // sample row / column is a wave-uniform `continue`.  The usual RoI (every sample valid) takes the branch-free form, whose
  // 28 tap loads per bin row the compiler batches — a uniform branch per sample would serialise them (measured: 0.19 -> 0.49 ms).
  const bool all_valid = yvalid == ((1ull << (PHT * SRT)) - 1ull) && xvalid == ((1ull << (PWT * SRT)) - 1ull);
  auto bin_rows = [&](auto safe_tag) {
    constexpr bool kSafe = decltype(safe_tag)::value;
    for (int ph = 0; ph < PHT; ++ph) {
      float acc[PWT][CPL];
  #pragma unroll
      for (int pw = 0; pw < PWT; ++pw)
  #pragma unroll
        for (int e = 0; e < CPL; ++e) acc[pw][e] = 0.f;
  #pragma unroll
      for (int iy = 0; iy < SRT; ++iy) {
        const int sy = ph * SRT + iy;
        if constexpr (kSafe) { if (!((yvalid >> sy) & 1ull)) continue; }   // skipped sample row
        const int r0 = __builtin_amdgcn_readlane(yoff, sy);
        const int rstep = __builtin_amdgcn_readlane(ystep, sy);
        const float hy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yh), sy));
        const float ly = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yl), sy));
        const T* row0 = nbase + r0;
        const T* row1 = row0 + rstep;
  #pragma unroll
        for (int j = 0; j < PWT * SRT; ++j) {
          if constexpr (kSafe) { if (!((xvalid >> j) & 1ull)) continue; }   // skipped sample column
          const char* p0 = reinterpret_cast<const char*>(row0 + sx0[j]);
          const char* p1 = reinterpret_cast<const char*>(row1 + sx0[j]);
          const char* q0 = reinterpret_cast<const char*>(row0 + sx0[j] + C);
          const char* q1 = reinterpret_cast<const char*>(row1 + sx0[j] + C);
          float v00[CPL], v01[CPL], v10[CPL], v11[CPL];
          unpack_word<T>(*reinterpret_cast<const unsigned*>(p0 + cl4), v00);
          unpack_word<T>(*reinterpret_cast<const unsigned*>(q0 + cl4), v01);
          unpack_word<T>(*reinterpret_cast<const unsigned*>(p1 + cl4), v10);
          unpack_word<T>(*reinterpret_cast<const unsigned*>(q1 + cl4), v11);
  #pragma unroll
          for (int e = 0; e < CPL; ++e) {
            const float t0 = __builtin_fmaf(slx[j], v01[e], mul_legacy(shx[j], v00[e]));
            const float t1 = __builtin_fmaf(slx[j], v11[e], mul_legacy(shx[j], v10[e]));
            float& a = acc[j / SRT][e];
            a = __builtin_fmaf(hy, t0, a);
            a = __builtin_fmaf(ly, t1, a);
          }
        }
      }
  #pragma unroll
      for (int pw = 0; pw < PWT; ++pw)
  #pragma unroll
        for (int e = 0; e < CPL; ++e)
          st(tl + (lane * CPL + e) * PHW + ph * PWT + pw, kPow2 ? acc[pw][e] * inv_count : acc[pw][e] / (float)NS);
    }
  };
  if (all_valid) bin_rows(std::false_type{});
  else bin_rows(std::true_type{});
  // ---- the [channel][bin] block is contiguous in the NCHW output: one linear stream of 32-bit words
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  unsigned* outw = reinterpret_cast<unsigned*>(output + ((int64_t)k * C + c0) * PHW);
  const unsigned* tlw = reinterpret_cast<const unsigned*>(tl);
  const int nwords = min(GC, C - c0) * PHW / CPL;  // C even for the 16-bit types (checked by the launcher)
#pragma unroll 7
  for (int it = 0; it < PHW; ++it) {
    const int idx = it * 64 + lane;
    if (idx < nwords) __builtin_nontemporal_store(tlw[idx], outw + idx);
  }
}

template <typename T>
int launch_ms_fwd_nhwc(const MsLevels& lv, const void* rois, void* output, int64_t N, int64_t C, int64_t K, int64_t PH, int64_t PW,
                       int64_t sr, int aligned, int* perm, hipStream_t stream) {
  constexpr int GC = 64 * (4 / (int)sizeof(T));
  const int ngroups = (int)ceil_div(C, GC);
  const int64_t nunits = K * ngroups;
  if (!(PH == 7 && PW == 7 && sr == 2))
    return set_error((int)hipErrorInvalidValue,
                     "roi_align (channels_last): only 7x7 bins with sampling_ratio 2 have a native NHWC kernel");
  // launch order (the NCHW kernel's one-workgroup counting sort by image, level, window-top band) when the caller gave scratch
  const int* order = nullptr;
  const int64_t L = lv.n_levels;
  if (perm && g_fwd_opt.order && N >= 1 && N * L <= kOrderBuckets && K <= kOrderMaxRois) {
    const int bands = (int)std::max<int64_t>(1, std::min<int64_t>(g_fwd_opt.bands, kOrderBuckets / (N * L)));
    roi_fwd_order<float><<<dim3(1), dim3(kOrderThreads), 0, stream>>>(lv, static_cast<const float*>(rois), (int)K, (int)N, 1, bands, perm,
                                                                        nullptr);
    order = perm;
  }
  const dim3 grid(wave_unit_grid_nhwc(K, ngroups, order != nullptr)), block(kThreads);
  roi_align_fwd_nhwc<T, 7, 7, 2><<<grid, block, 0, stream>>>(lv, static_cast<const float*>(rois), static_cast<T*>(output), (int)C,
                                                             aligned, ngroups, nunits, order);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_multiscale_roi_align_forward_nhwc");
}

// Order entries of the folded pre-pass (FoldArgs): a block of kOrderMaxFold 64-bit words the LIBRARY owns, one per (device,
// stream), zeroed once on that stream, and an epoch per block that only grows.  The block is written by nothing but the sort
// workgroups of launches on that stream — `epoch << 32 | RoI` — and launches of a stream run one after the other, so a word that
// carries this launch's epoch can only be this launch's entry.  (The first version kept the entries in the CALLER's workspace:
// recycled memory holds arbitrary bits, and the fuzzer produced a leftover pair of ints whose upper half happened to equal the
// epoch — one RoI pooled twice, another never.  A tag cannot vouch for memory other writers touch.)  At the wrap of the epoch
// (2^31 launches) the block is zeroed again.  A capturing stream gets no block (the epoch would be frozen into the graph node):
// the caller takes the two-launch form.
struct FoldBlocks {
  std::mutex mu;
  struct Block {
    int dev;
    hipStream_t stream;
    unsigned long long* words;
    int epoch;
  };
  std::vector<Block> blocks;
  int resident = 0;   // workgroups of the 7 x 7 kernels the chip holds at once (4 per CU)
  bool take(hipStream_t s, unsigned long long** words, int* epoch, int* resident_wgs) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return false;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (resident == 0) {
      int cus = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
      resident = 4 * cus;
    }
    *resident_wgs = resident;
    constexpr size_t kBytes = (size_t)kOrderMaxFold * sizeof(unsigned long long);
    for (auto& b : blocks)
      if (b.dev == dev && b.stream == s) {
        if (b.epoch == 0x7fffffff) {
          if (hipMemsetAsync(b.words, 0, kBytes, s) != hipSuccess) return false;
          b.epoch = 0;
        }
        *words = b.words;
        *epoch = ++b.epoch;
        return true;
      }
    unsigned long long* p = nullptr;
    // zeroed ON THE STREAM of the launches that will use it (hipMemset on the null stream may return before the fill has run)
    if (hipMalloc(reinterpret_cast<void**>(&p), kBytes) != hipSuccess || hipMemsetAsync(p, 0, kBytes, s) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    blocks.push_back(Block{dev, s, p, 1});
    *words = p;
    *epoch = 1;
    return true;
  }
};
inline FoldBlocks& fold_blocks() {
  static FoldBlocks* f = new FoldBlocks();   // never destroyed: the runtime may be gone at exit
  return *f;
}

template <typename T>
int launch_ms_fwd(const tvmi::MsLevels& lv, const void* rois, void* output, int64_t N, int64_t C, int64_t K, int64_t PH,
                  int64_t PW, int64_t sr, int aligned, int* mop, int* perm, hipStream_t stream, const MsBoxLists* bl = nullptr,
                  const StepArgs* carried = nullptr) {
  const float* r = static_cast<const float*>(rois);
  T* out = static_cast<T*>(output);
  const int nchunks = (int)ceil_div(C, kUnitChunk), mop_nchunks = (int)ceil_div(C, kMopChunk);
  const int64_t nunits = K * nchunks, mop_nunits = K * mop_nchunks;
  const bool fast_shape = (PH == 7 && PW == 7 && sr == 2) || (PH == 14 && PW == 14 && sr == 2);
  UnitMap um{nullptr, 0};
  FoldArgs fa{};
  unsigned front = carried ? (unsigned)((carried->S * carried->gdim_y + 7) & ~7) : 0u;   // workgroups in front of the units
  if (fast_shape && mop && bl && PH == 7 && g_fwd_opt.inline_mop && g_fwd_opt.fold_order && K <= kOrderMaxFold &&
      plan_units_orders(N, lv.n_levels, nchunks, perm)) {
    // the order pre-pass as a workgroup of the launch: worth it when the sorted part is most of the RoIs
    int epoch = 0, resident = 0;
    unsigned long long* p64 = nullptr;
    if (fold_blocks().take(stream, &p64, &epoch, &resident)) {
      const int64_t n0 = 4 * (std::max<int64_t>(0, ((int64_t)resident - front - kFoldBlocks) / 8) * g_fwd_opt.fold_pct / 100);   // positions of the first round
      if (2 * n0 <= K) {
        fa.perm64 = p64;
        fa.epoch = epoch;
        fa.n0 = (int)n0;
        fa.bands = (int)std::max<int64_t>(1, std::min<int64_t>(g_fwd_opt.bands, kOrderBuckets / (N * lv.n_levels)));
        fa.N = (int)N;
        fa.rois_out = const_cast<float*>(r);
        fa.bl = *bl;
        um = UnitMap{nullptr, 1};
        front += kFoldBlocks;
      }
    }
  }
  if (fast_shape && mop && !fa.perm64) {
    const int st = plan_units<float>(um, lv, r, N, K, nchunks, /*multiscale=*/1, mop, perm, stream, bl,
                                     const_cast<float*>(r));   // with box lists `rois` is the buffer the pre-pass fills
    if (st != 0) return set_error(st, "tvmi_multiscale_roi_align_forward: clearing the worklist failed");
  }
  const dim3 dma_grid(wave_unit_grid(K, nchunks, um.pinned != 0)), grid(wave_unit_grid(K, nchunks)),
      mop_grid((unsigned)std::min<int64_t>(kMopBlocks, ceil_div(mop_nunits, kThreads / 64))), block(kThreads);
#define TVMI_MS_DMA(PHT, PWT, SRT)                                                                                  \
  roi_align_fwd_ms_dma<T, PHT, PWT, SRT><<<dma_grid, block, 0, stream>>>(lv, r, out, (int)C, aligned, nchunks, kUnitChunk, \
                                                                               nunits, mop, um)
#define TVMI_MS(PHT, PWT, SRT)                                                                                      \
  if (mop && g_fwd_opt.inline_mop && PHT == 7) {   /* 14 x 14: the union of both paths needs 221 VGPRs, two waves per SIMD */ \
    if constexpr (PHT == 7) {                                                                                         \
      if (carried) {   /* the step's NMS workgroups in front of the grid (ms_fwd_can_carry_step) */                    \
        const unsigned nb = (unsigned)((carried->S * carried->gdim_y + 7) & ~7);                                      \
        roi_align_fwd_ms_dma_inl_step<T, PHT, PWT, SRT><<<dim3(dma_grid.x + front), block, 0, stream>>>(             \
            lv, r, out, (int)C, aligned, nchunks, kUnitChunk, nunits, um, *carried, nb, fa);                          \
      } else {                                                                                                        \
        roi_align_fwd_ms_dma_inl<T, PHT, PWT, SRT><<<dim3(dma_grid.x + front), block, 0, stream>>>(                  \
            lv, r, out, (int)C, aligned, nchunks, kUnitChunk, nunits, um, fa);                                        \
      }                                                                                                               \
    }                                                                                                                 \
  } else if (mop) {                                                                                                   \
    TVMI_MS_DMA(PHT, PWT, SRT);                                                                                     \
    roi_align_fwd_ms_wave<T, PHT, PWT, SRT><<<mop_grid, block, 0, stream>>>(lv, r, out, (int)C, (int)PH, (int)PW, (int)sr, \
                                                                            aligned, mop_nchunks, kMopChunk, mop_nunits, \
                                                                            mop);                                   \
  } else                                                                                                            \
    roi_align_fwd_ms_wave<T, PHT, PWT, SRT><<<grid, block, 0, stream>>>(lv, r, out, (int)C, (int)PH, (int)PW, (int)sr, \
                                                                        aligned, nchunks, kUnitChunk, nunits, nullptr)
  if (PH == 7 && PW == 7 && sr == 2) {
    TVMI_MS(7, 7, 2);
  } else if (PH == 14 && PW == 14 && sr == 2) {
    TVMI_MS(14, 14, 2);
  } else {
    roi_align_fwd_ms_wave<T, 0, 0, 0><<<grid, block, 0, stream>>>(lv, r, out, (int)C, (int)PH, (int)PW, (int)sr, aligned,
                                                                  nchunks, kUnitChunk, nunits, nullptr);
  }
#undef TVMI_MS
#undef TVMI_MS_DMA
  TVMI_RETURN_LAUNCH_STATUS("tvmi_multiscale_roi_align_forward");
}

}  // namespace

int set_roi_option(const char* name, int64_t value) {
  if (!name) return -1;
  const std::string n(name);
  if (n == "roi_align.pin_chunks") g_fwd_opt.pin_chunks = value != 0;
  else if (n == "roi_align.order") g_fwd_opt.order = value != 0;
  else if (n == "roi_align.inline_mop") g_fwd_opt.inline_mop = value != 0;
  else if (n == "roi_align.carry_step") g_fwd_opt.carry_step = value != 0;
  else if (n == "roi_align.fold_order") g_fwd_opt.fold_order = value != 0;
  else if (n == "roi_align.fold_first_round_pct") g_fwd_opt.fold_pct = (int)std::max<int64_t>(0, std::min<int64_t>(value, 400));
  else if (n == "roi_align.order_bands") g_fwd_opt.bands = (int)std::max<int64_t>(1, std::min<int64_t>(value, 64));
  else return -1;
  return 0;
}

int get_roi_option(const char* name, int64_t* value) {
  if (!name) return -1;
  const std::string n(name);
  if (n == "roi_align.pin_chunks") *value = g_fwd_opt.pin_chunks;
  else if (n == "roi_align.order") *value = g_fwd_opt.order;
  else if (n == "roi_align.order_bands") *value = g_fwd_opt.bands;
  else if (n == "roi_align.inline_mop") *value = g_fwd_opt.inline_mop;
  else if (n == "roi_align.carry_step") *value = g_fwd_opt.carry_step;
  else if (n == "roi_align.fold_order") *value = g_fwd_opt.fold_order;
  else if (n == "roi_align.fold_first_round_pct") *value = g_fwd_opt.fold_pct;
  else return -1;
  return 0;
}

}  // namespace tvmi

extern "C" size_t tvmi_roi_align_forward_workspace_bytes(int64_t K, int64_t pooled_h, int64_t pooled_w, int64_t sampling_ratio) {
  if (K <= 0) return 0;
  (void)pooled_h; (void)pooled_w; (void)sampling_ratio;
  return tvmi::fwd_mop_bytes(K) + (size_t)K * sizeof(int);   // [mop-up worklist][launch order]
}

extern "C" int tvmi_get_option(const char* name, int64_t* value) {
  if (name && value) {
    if (tvmi::get_roi_option(name, value) == 0) return 0;
    if (tvmi::get_nms_option(name, value) == 0) return 0;
    if (tvmi::get_dcn_option(name, value) == 0) return 0;
    if (tvmi::get_dcn_bwd_option(name, value) == 0) return 0;
  }
  return tvmi::set_error((int)hipErrorInvalidValue, "tvmi_get_option: unknown option");
}

extern "C" int tvmi_set_option(const char* name, int64_t value) {
  if (tvmi::set_roi_option(name, value) == 0) return 0;
  if (tvmi::set_nms_option(name, value) == 0) return 0;
  if (tvmi::set_dcn_option(name, value) == 0) return 0;
  if (tvmi::set_dcn_bwd_option(name, value) == 0) return 0;
  return tvmi::set_error((int)hipErrorInvalidValue, "tvmi_set_option: unknown option");
}

extern "C" int tvmi_roi_align_forward(const void* input, const void* rois, void* output,
                                      tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W,
                                      int64_t K, int64_t pooled_h, int64_t pooled_w,
                                      double spatial_scale, int64_t sampling_ratio, int aligned,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "roi_align: pooled size must be positive");
  int* declined = tvmi::mop_part(workspace, workspace_bytes, K);
  int* perm = tvmi::perm_part(workspace, workspace_bytes, K);
  TVMI_CHECK_ARG(N >= 0 && C >= 0 && H >= 0 && W >= 0 && K >= 0, "roi_align: negative size");
  if (K * C * pooled_h * pooled_w == 0) return 0;
  TVMI_CHECK_ARG(input && rois && output, "roi_align: null pointer");
  // in-plane BYTE offsets of the wave kernels are unsigned 32-bit (stage_window)
  TVMI_CHECK_ARG(H * W < (1ll << 31) && (dt != TVMI_F32 || H * W < (1ll << 30)) && K * tvmi::ceil_div(C, 32) < (1ll << 31),
                 "roi_align: size exceeds 32-bit launch limits");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "roi_align_forward",
                      return tvmi::launch_fwd<scalar_t>(input, rois, output, N, C, H, W, K, pooled_h,
                                                        pooled_w, spatial_scale, sampling_ratio,
                                                        aligned, declined, perm, s));
  return 0;
}

extern "C" int tvmi_multiscale_roi_align_forward(const void* const* inputs, const int64_t* heights,
                                                 const int64_t* widths, const double* spatial_scales,
                                                 int64_t n_levels, const void* rois, void* output, tvmi_dtype dt,
                                                 int64_t N, int64_t C, int64_t K, int64_t pooled_h,
                                                 int64_t pooled_w, int64_t sampling_ratio, int aligned,
                                                 int64_t k_min, int64_t k_max, double canonical_scale,
                                                 double canonical_level, double eps, void* workspace,
                                                 size_t workspace_bytes, void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "multiscale_roi_align: pooled size must be positive");
  TVMI_CHECK_ARG(n_levels >= 1 && n_levels <= tvmi::kMaxLevels, "multiscale_roi_align: 1..8 levels supported");
  if (K * C * pooled_h * pooled_w == 0) return 0;
  TVMI_CHECK_ARG(inputs && heights && widths && spatial_scales && rois && output, "multiscale_roi_align: null pointer");
  TVMI_CHECK_ARG(dt == TVMI_F32 || dt == TVMI_F16 || dt == TVMI_BF16,
                 "multiscale_roi_align: float32 / float16 / bfloat16 only");
  TVMI_CHECK_ARG(K * tvmi::ceil_div(C, 32) < (1ll << 31), "multiscale_roi_align: size exceeds 32-bit launch limits");
  for (int64_t i = 0; i < n_levels; ++i)
    TVMI_CHECK_ARG(inputs[i] != nullptr && heights[i] * widths[i] < (1ll << 30), "multiscale_roi_align: bad level");
  tvmi::MsLevels lv;
  tvmi::fill_levels(lv, inputs, heights, widths, spatial_scales, n_levels, k_min, k_max, canonical_scale, canonical_level, eps);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int* declined = tvmi::mop_part(workspace, workspace_bytes, K);
  int* perm = tvmi::perm_part(workspace, workspace_bytes, K);
  switch (dt) {
    case TVMI_F32:
      return tvmi::launch_ms_fwd<float>(lv, rois, output, N, C, K, pooled_h, pooled_w, sampling_ratio, aligned, declined, perm, s);
    case TVMI_F16:
      return tvmi::launch_ms_fwd<__half>(lv, rois, output, N, C, K, pooled_h, pooled_w, sampling_ratio, aligned, declined, perm, s);
    default:
      return tvmi::launch_ms_fwd<__hip_bfloat16>(lv, rois, output, N, C, K, pooled_h, pooled_w, sampling_ratio, aligned, declined,
                                                 perm, s);
  }
}

// The same call taking the per-image box lists instead of [K,5] rows: where the order pre-pass runs (7x7 / 14x14 bins,
// sampling_ratio 2, a multiple of 8 channel chunks, workspace given) it builds the rows itself; otherwise the rows are built by
// tvmi_boxes_to_rois first.  `rois_out` [K,5] float32 receives them either way (the backward needs them).
static int ms_fwd_boxes(const tvmi::StepArgs* carried, const void* const* inputs, const int64_t* heights, const int64_t* widths,
                                                       const double* spatial_scales, int64_t n_levels, const void* const* boxes,
                                                       const int64_t* counts, int64_t num_images, void* rois_out, void* output,
                                                       tvmi_dtype dt, int64_t N, int64_t C, int64_t pooled_h, int64_t pooled_w,
                                                       int64_t sampling_ratio, int aligned, int64_t k_min, int64_t k_max,
                                                       double canonical_scale, double canonical_level, double eps, void* workspace,
                                                       size_t workspace_bytes, void* stream) {
  TVMI_CHECK_ARG(num_images >= 0 && num_images <= tvmi::kOrderMaxImages, "multiscale_roi_align (boxes): at most 64 images per call");
  TVMI_CHECK_ARG(num_images == 0 || (boxes && counts), "multiscale_roi_align (boxes): null pointer");
  tvmi::MsBoxLists bl;
  int64_t K = 0;
  bl.n = (int)num_images;
  for (int64_t i = 0; i < num_images; ++i) {
    TVMI_CHECK_ARG(counts[i] >= 0 && (counts[i] == 0 || boxes[i]), "multiscale_roi_align (boxes): bad box list");
    K += counts[i];
    TVMI_CHECK_ARG(K < (1ll << 31), "multiscale_roi_align (boxes): too many boxes");
    bl.ptr[i] = static_cast<const float*>(boxes[i]);
    bl.end[i] = (int)K;
  }
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "multiscale_roi_align: pooled size must be positive");
  TVMI_CHECK_ARG(n_levels >= 1 && n_levels <= tvmi::kMaxLevels, "multiscale_roi_align: 1..8 levels supported");
  if (K == 0) return 0;
  TVMI_CHECK_ARG(rois_out != nullptr, "multiscale_roi_align (boxes): rois_out is null");
  const int nchunks = (int)tvmi::ceil_div(C, tvmi::kUnitChunk);
  int* declined = tvmi::mop_part(workspace, workspace_bytes, K);
  int* perm = tvmi::perm_part(workspace, workspace_bytes, K);
  const bool fast_shape = (pooled_h == 7 && pooled_w == 7 && sampling_ratio == 2) || (pooled_h == 14 && pooled_w == 14 && sampling_ratio == 2);
  const bool fused = fast_shape && declined && K <= tvmi::kOrderMaxRois && C * pooled_h * pooled_w > 0 &&
                     tvmi::plan_units_orders(N, n_levels, nchunks, perm) && (dt == TVMI_F32 || dt == TVMI_F16 || dt == TVMI_BF16);
  if (carried && !(fused && pooled_h == 7 && tvmi::g_fwd_opt.inline_mop))   // (the caller asked ms_fwd_boxes_carries first)
    return tvmi::set_error((int)hipErrorInvalidValue, "multiscale_roi_align (boxes): this call cannot carry the NMS step");
  if (!fused) {
    const int st = tvmi_boxes_to_rois(boxes, counts, num_images, rois_out, TVMI_F32, stream);
    if (st != 0) return st;
    return tvmi_multiscale_roi_align_forward(inputs, heights, widths, spatial_scales, n_levels, rois_out, output, dt, N, C, K, pooled_h,
                                             pooled_w, sampling_ratio, aligned, k_min, k_max, canonical_scale, canonical_level, eps,
                                             workspace, workspace_bytes, stream);
  }
  TVMI_CHECK_ARG(inputs && heights && widths && spatial_scales && output, "multiscale_roi_align: null pointer");
  TVMI_CHECK_ARG(K * tvmi::ceil_div(C, 32) < (1ll << 31), "multiscale_roi_align: size exceeds 32-bit launch limits");
  for (int64_t i = 0; i < n_levels; ++i)
    TVMI_CHECK_ARG(inputs[i] != nullptr && heights[i] * widths[i] < (1ll << 30), "multiscale_roi_align: bad level");
  tvmi::MsLevels lv;
  tvmi::fill_levels(lv, inputs, heights, widths, spatial_scales, n_levels, k_min, k_max, canonical_scale, canonical_level, eps);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dt) {
    case TVMI_F32:
      return tvmi::launch_ms_fwd<float>(lv, rois_out, output, N, C, K, pooled_h, pooled_w, sampling_ratio, aligned, declined, perm, s, &bl, carried);
    case TVMI_F16:
      return tvmi::launch_ms_fwd<__half>(lv, rois_out, output, N, C, K, pooled_h, pooled_w, sampling_ratio, aligned, declined, perm, s, &bl, carried);
    default:
      return tvmi::launch_ms_fwd<__hip_bfloat16>(lv, rois_out, output, N, C, K, pooled_h, pooled_w, sampling_ratio, aligned, declined,
                                                 perm, s, &bl, carried);
  }
}


extern "C" int tvmi_multiscale_roi_align_forward_boxes(const void* const* inputs, const int64_t* heights, const int64_t* widths,
                                                       const double* spatial_scales, int64_t n_levels, const void* const* boxes,
                                                       const int64_t* counts, int64_t num_images, void* rois_out, void* output,
                                                       tvmi_dtype dt, int64_t N, int64_t C, int64_t pooled_h, int64_t pooled_w,
                                                       int64_t sampling_ratio, int aligned, int64_t k_min, int64_t k_max,
                                                       double canonical_scale, double canonical_level, double eps, void* workspace,
                                                       size_t workspace_bytes, void* stream) {
  return ms_fwd_boxes(nullptr, inputs, heights, widths, spatial_scales, n_levels, boxes, counts, num_images, rois_out, output, dt, N, C,
                      pooled_h, pooled_w, sampling_ratio, aligned, k_min, k_max, canonical_scale, canonical_level, eps, workspace,
                      workspace_bytes, stream);
}

// The detector step as one call: MultiScaleRoIAlign over the box lists AND the step's NMS (tvmi_nms_step: same arguments, same
// results).  Where the RoIAlign call takes the one-launch 7 x 7 route, the NMS workgroups ride in front of its grid
// (roi_align_fwd_ms_dma_inl_step; option "roi_align.carry_step", default 1); anywhere else the two entries run one after the other on
// `stream`.  The two jobs share nothing but the launch: neither reads what the other writes.
extern "C" int tvmi_multiscale_roi_align_forward_boxes_with_nms_step(
    const void* const* inputs, const int64_t* heights, const int64_t* widths, const double* spatial_scales, int64_t n_levels,
    const void* const* boxes, const int64_t* counts, int64_t num_images, void* rois_out, void* output, tvmi_dtype dt, int64_t N, int64_t C,
    int64_t pooled_h, int64_t pooled_w, int64_t sampling_ratio, int aligned, int64_t k_min, int64_t k_max, double canonical_scale,
    double canonical_level, double eps, void* workspace, size_t workspace_bytes, const float* nms_dets, const float* nms_scores,
    const int64_t* nms_seg, int64_t nms_n, int64_t nms_segments, double iou_threshold, void* nms_workspace, size_t nms_workspace_bytes,
    int64_t* keep_out, int64_t* num_keep_out, const int64_t* image_idx, const int64_t* labels, int64_t det_images, int64_t max_dets,
    float* payload, int64_t row_stride, int32_t* det_counts, int count_in_row, void* stream) {
  int64_t K = 0;
  for (int64_t i = 0; counts && i < num_images && i < tvmi::kOrderMaxImages; ++i) K += counts[i] > 0 ? counts[i] : 0;
  const int nchunks = (int)tvmi::ceil_div(C, tvmi::kUnitChunk);
  const bool carries = tvmi::g_fwd_opt.carry_step && tvmi::g_fwd_opt.inline_mop && pooled_h == 7 && pooled_w == 7 && sampling_ratio == 2 &&
                       K > 0 && K <= tvmi::kOrderMaxRois && C > 0 && num_images <= tvmi::kOrderMaxImages && rois_out != nullptr &&
                       tvmi::mop_part(workspace, workspace_bytes, K) != nullptr &&
                       tvmi::plan_units_orders(N, n_levels, nchunks, tvmi::perm_part(workspace, workspace_bytes, K)) &&
                       (dt == TVMI_F32 || dt == TVMI_F16 || dt == TVMI_BF16);
  if (!carries) {
    const int st = tvmi_nms_step(nms_dets, nms_scores, nms_seg, nms_n, nms_segments, iou_threshold, nms_workspace, nms_workspace_bytes,
                                 keep_out, num_keep_out, image_idx, labels, det_images, max_dets, payload, row_stride, det_counts,
                                 count_in_row, stream);
    if (st != 0) return st;
    return ms_fwd_boxes(nullptr, inputs, heights, widths, spatial_scales, n_levels, boxes, counts, num_images, rois_out, output, dt, N, C,
                        pooled_h, pooled_w, sampling_ratio, aligned, k_min, k_max, canonical_scale, canonical_level, eps, workspace,
                        workspace_bytes, stream);
  }
  tvmi::StepArgs a;
  const int st = tvmi::step_prepare(&a, nms_dets, nms_scores, nms_seg, nms_n, nms_segments, iou_threshold, nms_workspace,
                                    nms_workspace_bytes, keep_out, num_keep_out, image_idx, labels, det_images, max_dets, payload,
                                    row_stride, det_counts, count_in_row, static_cast<hipStream_t>(stream));
  if (st != 0) return st;
  return ms_fwd_boxes(&a, inputs, heights, widths, spatial_scales, n_levels, boxes, counts, num_images, rois_out, output, dt, N, C,
                      pooled_h, pooled_w, sampling_ratio, aligned, k_min, k_max, canonical_scale, canonical_level, eps, workspace,
                      workspace_bytes, stream);
}

extern "C" int tvmi_multiscale_roi_align_forward_nhwc(const void* const* inputs, const int64_t* heights,
                                                      const int64_t* widths, const double* spatial_scales,
                                                      int64_t n_levels, const void* rois, void* output, tvmi_dtype dt,
                                                      int64_t N, int64_t C, int64_t K, int64_t pooled_h, int64_t pooled_w,
                                                      int64_t sampling_ratio, int aligned, int64_t k_min, int64_t k_max,
                                                      double canonical_scale, double canonical_level, double eps,
                                                      void* workspace, size_t workspace_bytes, void* stream) {
  TVMI_CHECK_ARG(n_levels >= 1 && n_levels <= tvmi::kMaxLevels, "roi_align (channels_last): 1..8 levels supported");
  if (K * C * pooled_h * pooled_w == 0) return 0;
  TVMI_CHECK_ARG(inputs && heights && widths && spatial_scales && rois && output, "roi_align (channels_last): null pointer");
  TVMI_CHECK_ARG(dt == TVMI_F32 || ((dt == TVMI_F16 || dt == TVMI_BF16) && C % 2 == 0),
                 "roi_align (channels_last): float32, or float16 / bfloat16 with an even channel count");
  TVMI_CHECK_ARG(pooled_h == 7 && pooled_w == 7 && sampling_ratio == 2,
                 "roi_align (channels_last): only 7x7 bins with sampling_ratio 2 have a native NHWC kernel");
  TVMI_CHECK_ARG(K * tvmi::ceil_div(C, 64) < (1ll << 31), "roi_align (channels_last): size exceeds 32-bit launch limits");
  for (int64_t i = 0; i < n_levels; ++i)
    TVMI_CHECK_ARG(inputs[i] != nullptr && heights[i] >= 2 && widths[i] >= 2 && heights[i] * widths[i] * C < (1ll << 31),
                   "roi_align (channels_last): every level needs H, W >= 2 and H*W*C < 2^31");
  tvmi::MsLevels lv;
  tvmi::fill_levels(lv, inputs, heights, widths, spatial_scales, n_levels, k_min, k_max, canonical_scale, canonical_level, eps);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // optional scratch: K ints for the launch order (tvmi_roi_align_forward_workspace_bytes covers it); without it the RoIs run in
  // input order
  int* perm = (workspace && workspace_bytes >= (size_t)K * sizeof(int) && (reinterpret_cast<uintptr_t>(workspace) & 3) == 0)
                  ? static_cast<int*>(workspace) : nullptr;
  if (dt == TVMI_F32) return tvmi::launch_ms_fwd_nhwc<float>(lv, rois, output, N, C, K, pooled_h, pooled_w, sampling_ratio, aligned, perm, s);
  if (dt == TVMI_F16) return tvmi::launch_ms_fwd_nhwc<__half>(lv, rois, output, N, C, K, pooled_h, pooled_w, sampling_ratio, aligned, perm, s);
  return tvmi::launch_ms_fwd_nhwc<__hip_bfloat16>(lv, rois, output, N, C, K, pooled_h, pooled_w, sampling_ratio, aligned, perm, s);
}
