// roi_align.hip — RoIAlign forward / backward for gfx950 (MI355X).
//
// Semantics: torchvision/csrc/ops/cpu/roi_align_kernel.cpp:18-115 (forward), :117-289
// (backward) and cpu/roi_align_common.h:32-124 (sample -> 4 taps + weights).  The TU is
// built with -ffp-contract=off so the sample-coordinate arithmetic rounds exactly like the
// reference's x86 build; fused multiply-adds are only used where written explicitly.
//
// Design (not the reference's one-thread-per-output gather, cuda/roi_align_kernel.cu:68):
//   * one 256-thread workgroup per (RoI, chunk of CCH channels);
//   * the RoI's PH*gh y-samples and PW*gw x-samples are decomposed ONCE into two small
//     1-D tables in LDS (low index, low/high weight) — the bilinear weights are separable,
//     so the reference's PH*PW*gh*gw PreCalc array is never materialised;
//   * the RoI's bounding window of the feature map is staged per channel group into LDS
//     with row-contiguous (coalesced along W) loads, and all 4 taps of every sample are
//     gathered from LDS (ds_read2_b32 pairs), not from L1/L2;
//   * outputs of a channel group are written as one contiguous run (K,C,PH,PW is
//     contiguous in (c,ph,pw) for fixed k), fully coalesced.
//   * RoIs whose tables or windows do not fit fall back, per workgroup, to table-driven
//     global gathers or to on-the-fly arithmetic.
// Backward mirrors this: gradients of a channel group are accumulated into the LDS window
// with ds_add_f32 and flushed with ONE global atomic per touched pixel instead of 4 per
// sample.
#include <stdlib.h>

#include <type_traits>

#include "tvmi_common.h"

namespace tvmi {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxTab = 128;      // max PH*gh (and PW*gw) samples per axis kept in LDS
constexpr int kWinFloats = 8192;  // LDS window capacity in floats (32 KiB)
constexpr int kChunk = 32;        // channels per workgroup
constexpr int kMaxLevels = 8;     // FPN levels served by one multi-scale launch

template <typename A>
struct RoiGeom {
  A start_h, start_w, bin_h, bin_w, count;
  int gh, gw, batch;
};

// cpu/roi_align_kernel.cpp:36-66
template <typename T, typename A>
__device__ __forceinline__ RoiGeom<A> roi_geom(const T* roi, A scale, int PH, int PW, int sr,
                                               bool aligned) {
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
__device__ __forceinline__ bool axis_sample(int dim, A start, A bin, int grid, int p, int i, int& lo,
                                            int& hi, A& l, A& h) {
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
// Tiled forward (fp32 accumulate): see the header comment.
struct AxisTab {
  int lo[kMaxTab];
  int hi[kMaxTab];
  float l[kMaxTab];
  float h[kMaxTab];
};

struct TileShared {
  AxisTab y, x;
  int bounds[4];  // ymin, ymax, xmin, xmax over valid samples
  float win[kWinFloats];
};

enum { MODE_LDS = 0, MODE_TAB = 1, MODE_GEN = 2, MODE_ZERO = 3 };

// Builds both axis tables for this workgroup's RoI; returns the processing mode.
// On MODE_LDS the table `lo` entries are rewritten as window-relative offsets
// (y: row*wstride, x: col) and *G is the number of channels staged per pass.
template <typename T>
__device__ __forceinline__ int build_tables(TileShared& s, const RoiGeom<float>& g, int H, int W,
                                            int PH, int PW, int chunk_c, int& y0, int& x0,
                                            int& wh, int& ww, int& wstride, int& G) {
  const int tid = threadIdx.x;
  const int ny = PH * g.gh, nx = PW * g.gw;
  if (ny > kMaxTab || nx > kMaxTab || g.gh <= 0 || g.gw <= 0) {
    return (g.gh <= 0 || g.gw <= 0) ? MODE_ZERO : MODE_GEN;
  }
  if (tid == 0) {
    s.bounds[0] = 0x7fffffff;
    s.bounds[1] = -1;
    s.bounds[2] = 0x7fffffff;
    s.bounds[3] = -1;
  }
  __syncthreads();
  if (tid < ny) {
    int lo, hi;
    float l, h;
    const bool v = axis_sample<float>(H, g.start_h, g.bin_h, g.gh, tid / g.gh, tid % g.gh, lo, hi, l, h);
    s.y.lo[tid] = v ? lo : -1;
    s.y.hi[tid] = hi;
    s.y.l[tid] = l;
    s.y.h[tid] = h;
    if (v) {
      atomicMin(&s.bounds[0], lo);
      atomicMax(&s.bounds[1], hi);
    }
  } else if (tid >= kMaxTab && tid - kMaxTab < nx) {
    const int t = tid - kMaxTab;
    int lo, hi;
    float l, h;
    const bool v = axis_sample<float>(W, g.start_w, g.bin_w, g.gw, t / g.gw, t % g.gw, lo, hi, l, h);
    s.x.lo[t] = v ? lo : -1;
    s.x.hi[t] = hi;
    s.x.l[t] = l;
    s.x.h[t] = h;
    if (v) {
      atomicMin(&s.bounds[2], lo);
      atomicMax(&s.bounds[3], hi);
    }
  }
  __syncthreads();
  y0 = s.bounds[0];
  x0 = s.bounds[2];
  const int y1 = s.bounds[1], x1 = s.bounds[3];
  if (y1 < 0 || x1 < 0) return MODE_ZERO;  // every sample of one axis is out of range
  // Rows y0..y1+1 and cols x0..x1+1 are staged; the +1 pad only ever meets a zero weight.
  wh = y1 - y0 + 2;
  ww = x1 - x0 + 2;
  wstride = ww | 1;
  const int wsz = wh * wstride;
  G = kWinFloats / wsz;
  if (G > chunk_c) G = chunk_c;
  const int mode = G >= 1 ? MODE_LDS : MODE_TAB;
  // Second pass (same thread that wrote the entry): invalid samples get lo = origin and
  // zero weights; LDS mode turns lo into a window-relative offset.
  if (tid < ny) {
    int lo = s.y.lo[tid];
    if (lo < 0) {
      lo = y0;
      s.y.hi[tid] = y0;
    }
    s.y.lo[tid] = mode == MODE_LDS ? (lo - y0) * wstride : lo;
  } else if (tid >= kMaxTab && tid - kMaxTab < nx) {
    const int t = tid - kMaxTab;
    int lo = s.x.lo[t];
    if (lo < 0) {
      lo = x0;
      s.x.hi[t] = x0;
    }
    s.x.lo[t] = mode == MODE_LDS ? (lo - x0) : lo;
  }
  __syncthreads();
  return mode;
}

// One workgroup = one (RoI k, channel chunk starting at c0) of one feature map.
template <typename T, int PHT, int PWT, int SRT>
__device__ __forceinline__ void roi_align_fwd_block(TileShared& s, const T* __restrict__ input,
                                                    const T* __restrict__ rois, T* __restrict__ output,
                                                    int C, int H, int W, int PH_, int PW_,
                                                    float spatial_scale, int sr_, int aligned, int k, int c0) {
  const int PH = PHT > 0 ? PHT : PH_;
  const int PW = PWT > 0 ? PWT : PW_;
  const int sr = SRT > 0 ? SRT : sr_;
  const int PHW = PH * PW;
  const int tid = threadIdx.x;
  const int cc = min(kChunk, C - c0);

  RoiGeom<float> g = roi_geom<T, float>(rois + (int64_t)k * 5, spatial_scale, PH, PW, sr, aligned != 0);
  if (SRT > 0) {
    g.gh = SRT;
    g.gw = SRT;
  }
  T* out = output + ((int64_t)k * C + c0) * PHW;
  const T* in0 = input + ((int64_t)g.batch * C + c0) * H * W;
  const int64_t plane_sz = (int64_t)H * W;

  int y0 = 0, x0 = 0, wh = 0, ww = 0, wstride = 0, G = 0;
  const int mode = build_tables<T>(s, g, H, W, PH, PW, cc, y0, x0, wh, ww, wstride, G);

  if (mode == MODE_ZERO) {
    for (int o = tid; o < cc * PHW; o += kThreads) st(out + o, 0.f);
    return;
  }
  if (mode == MODE_GEN) {
    for (int o = tid; o < cc * PHW; o += kThreads) {
      const int c = o / PHW, bin = o - c * PHW;
      const int ph = bin / PW, pw = bin - ph * PW;
      st(out + o, roi_align_point<T, float>(in0 + c * plane_sz, H, W, g, ph, pw));
    }
    return;
  }
  const float inv_count = g.count;  // divide (not multiply) to round like the reference
  const int gh = g.gh, gw = g.gw;

  if (mode == MODE_TAB) {
    // Window too large for LDS: table-driven gathers straight from global memory.
    for (int o = tid; o < cc * PHW; o += kThreads) {
      const int c = o / PHW, bin = o - c * PHW;
      const int ph = bin / PW, pw = bin - ph * PW;
      const T* plane = in0 + c * plane_sz;
      float acc = 0.f;
      for (int iy = 0; iy < gh; ++iy) {
        const int ty = ph * gh + iy;
        const int ylo = s.y.lo[ty], yhi = s.y.hi[ty];
        const float ly = s.y.l[ty], hy = s.y.h[ty];
        for (int ix = 0; ix < gw; ++ix) {
          const int tx = pw * gw + ix;
          const int xlo = s.x.lo[tx], xhi = s.x.hi[tx];
          const float lx = s.x.l[tx], hx = s.x.h[tx];
          const float v1 = ld(plane + (int64_t)ylo * W + xlo), v2 = ld(plane + (int64_t)ylo * W + xhi);
          const float v3 = ld(plane + (int64_t)yhi * W + xlo), v4 = ld(plane + (int64_t)yhi * W + xhi);
          acc += (hy * hx) * v1 + (hy * lx) * v2 + (ly * hx) * v3 + (ly * lx) * v4;
        }
      }
      st(out + o, acc / inv_count);
    }
    return;
  }

  // MODE_LDS
  const int wsz = wh * wstride;
  const int wpix = wh * ww;
  FastDiv16 div_wpix, div_ww;
  div_wpix.init((unsigned)wpix);
  div_ww.init((unsigned)ww);
  for (int cg = 0; cg < cc; cg += G) {
    const int gc = min(G, cc - cg);
    // ---- stage gc channel windows (rows y0.., cols x0..; out-of-tensor pad = 0)
    const int total = gc * wpix;
    for (int e = tid; e < total; e += kThreads) {
      const int ch = (int)div_wpix.div((unsigned)e);
      const int rem = e - ch * wpix;
      const int r = (int)div_ww.div((unsigned)rem);
      const int col = rem - r * ww;
      const int gy = y0 + r, gx = x0 + col;
      float v = 0.f;
      if (gy < H && gx < W) v = ld(in0 + (cg + ch) * plane_sz + (int64_t)gy * W + gx);
      s.win[ch * wsz + r * wstride + col] = v;
    }
    __syncthreads();
    // ---- gather from LDS
    const int nout = gc * PHW;
    for (int o = tid; o < nout; o += kThreads) {
      const int ch = o / PHW, bin = o - ch * PHW;
      const int ph = bin / PW, pw = bin - ph * PW;
      const float* wbase = s.win + ch * wsz;
      float acc = 0.f;
      for (int iy = 0; iy < gh; ++iy) {
        const int ty = ph * gh + iy;
        const float* row = wbase + s.y.lo[ty];
        const float ly = s.y.l[ty], hy = s.y.h[ty];
        for (int ix = 0; ix < gw; ++ix) {
          const int tx = pw * gw + ix;
          const float* p = row + s.x.lo[tx];
          const float lx = s.x.l[tx], hx = s.x.h[tx];
          const float v1 = p[0], v2 = p[1], v3 = p[wstride], v4 = p[wstride + 1];
          acc += (hy * hx) * v1 + (hy * lx) * v2 + (ly * hx) * v3 + (ly * lx) * v4;
        }
      }
      st(out + cg * PHW + o, acc / inv_count);
    }
    __syncthreads();
  }
}

template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_tile(
    const T* __restrict__ input, const T* __restrict__ rois, T* __restrict__ output, int C, int H,
    int W, int PH_, int PW_, float spatial_scale, int sr_, int aligned, int nchunks) {
  __shared__ TileShared s;
  const int k = blockIdx.x / nchunks;
  const int c0 = (blockIdx.x - k * nchunks) * kChunk;
  roi_align_fwd_block<T, PHT, PWT, SRT>(s, input, rois, output, C, H, W, PH_, PW_, spatial_scale, sr_,
                                        aligned, k, c0);
}

// ---------------------------------------------------------------------------------------
// Multi-scale forward: the FPN level of every RoI is chosen IN the kernel
// (torchvision/ops/poolers.py:47-84, LevelMapper: floor(k0 + log2(sqrt(area)/s0) + eps) clamped
// to [k_min, k_max]) and all levels are served by ONE launch that writes straight into the
// [K,C,PH,PW] result — no torch.where / index / index_put round trips per level
// (poolers.py:199-222).
struct MsLevels {
  const void* ptr[kMaxLevels];
  int H[kMaxLevels];
  int W[kMaxLevels];
  float scale[kMaxLevels];
  int n_levels;
  int k_min, k_max;
  float s0, lvl0, eps;
};

template <typename T>
__device__ __forceinline__ int fpn_level(const T* roi, const MsLevels& lv) {
  const float x1 = ld(roi + 1), y1 = ld(roi + 2), x2 = ld(roi + 3), y2 = ld(roi + 4);
  const float s = sqrtf((x2 - x1) * (y2 - y1));
  float t = floorf(lv.lvl0 + log2f(s / lv.s0) + lv.eps);
  t = fminf(fmaxf(t, (float)lv.k_min), (float)lv.k_max);  // NaN area -> k_min like torch.clamp? (clamp keeps NaN)
  int l = (t == t) ? (int)t - lv.k_min : 0;
  return min(max(l, 0), lv.n_levels - 1);
}

template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_ms_tile(MsLevels lv, const T* __restrict__ rois,
                                                                 T* __restrict__ output, int C, int PH_,
                                                                 int PW_, int sr_, int aligned, int nchunks) {
  __shared__ TileShared s;
  const int k = blockIdx.x / nchunks;
  const int c0 = (blockIdx.x - k * nchunks) * kChunk;
  const int l = fpn_level<T>(rois + (int64_t)k * 5, lv);
  roi_align_fwd_block<T, PHT, PWT, SRT>(s, static_cast<const T*>(lv.ptr[l]), rois, output, C, lv.H[l], lv.W[l],
                                        PH_, PW_, lv.scale[l], sr_, aligned, k, c0);
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

__device__ int g_roi_force_mode = 0;  // debug knob (tvmi_debug_set): 0 auto, 1 never stage through LDS
__device__ int g_cfg_dma_sparse = 1;  // stage only the sampled rows of tall windows (TVMI_ROI_DMA_SPARSE=0: whole window)

// Stages CH channel windows (rows y0.., cols x0.., clamped into the tensor) with RG row-groups
// each: RG*CH independent loads are in flight per lane before the first LDS write.  Lanes
// beyond the window edge re-read the edge element (same address, same value), so there is no
// predication and no integer division anywhere on this path; the channel base pointer is
// wave-uniform (SGPR) and the per-lane offsets are 32-bit.
template <typename T, int RG, int CH>
__device__ __forceinline__ void stage_window(float* __restrict__ win, const T* __restrict__ in_cg, int64_t plane_sz,
                                             int gc, int nrg, int wsz, const int (&goff)[RG], const int (&loff)[RG]) {
  for (int ch0 = 0; ch0 < gc; ch0 += CH) {
    float v[CH][RG];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = min(ch0 + c, gc - 1);  // tail: duplicate the last channel (idempotent)
      const T* chp = in_cg + (int64_t)ch * plane_sz;
#pragma unroll
      for (int rg = 0; rg < RG; ++rg)
        if (rg < nrg) v[c][rg] = ld(chp + goff[rg]);
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = min(ch0 + c, gc - 1);
      float* wch = win + ch * wsz;
#pragma unroll
      for (int rg = 0; rg < RG; ++rg)
        if (rg < nrg) wch[loff[rg]] = v[c][rg];
    }
  }
}

template <typename T, int RG, int CH>
__device__ __forceinline__ void stage_window_setup(float* __restrict__ win, const T* __restrict__ in_cg,
                                                   int64_t plane_sz, int gc, int nrg, int wsz, int rpi, int rsub,
                                                   int colc, int gxc, int y0, int wh, int wstride, int H, int W) {
  int goff[RG], loff[RG];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    const int r = min(rg * rpi + rsub, wh - 1);
    goff[rg] = min(y0 + r, H - 1) * W + gxc;
    loff[rg] = r * wstride + colc;
  }
  stage_window<T, RG, CH>(win, in_cg, plane_sz, gc, nrg, wsz, goff, loff);
}

template <typename T, int PHT, int PWT, int SRT>
__device__ __forceinline__ void roi_align_fwd_wave_unit(WaveShared& s, const T* __restrict__ input,
                                                        const T* __restrict__ rois, T* __restrict__ output,
                                                        int C, int H, int W, int PH_, int PW_, float spatial_scale,
                                                        int sr_, int aligned, int k, int c0, int chunk, int force_mode) {
  const int PH = PHT > 0 ? PHT : PH_;
  const int PW = PWT > 0 ? PWT : PW_;
  const int sr = SRT > 0 ? SRT : sr_;
  const int PHW = PH * PW;
  const int lane = threadIdx.x & 63;
  const int cc = min(chunk, C - c0);
  RoiGeom<float> g = roi_geom<T, float>(rois + (int64_t)k * 5, spatial_scale, PH, PW, sr, aligned != 0);
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
    if (lane < ny) s.ytab[lane] = make_float4(__int_as_float(v ? lo : y0), l, h, 0.f);
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
    if (lane < nx) s.xtab[lane] = make_float4(__int_as_float(v ? lo : x0), l, h, 0.f);
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
  int G = (ww <= 64 && nrg <= 16 && force_mode != 1) ? kWaveWin / wsz : 0;
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
          acc += (hy * hx) * ld(r0 + xlo) + (hy * lx) * ld(r0 + xhi) + (ly * hx) * ld(r1 + xlo) + (ly * lx) * ld(r1 + xhi);
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
    if (force_mode == 2) {
    } else if (nrg <= 4)
      stage_window_setup<T, 4, 4>(s.win, in_cg, plane_sz, gc, nrg, wsz, rpi, rsub, colc, gxc, y0, wh, wstride, H, W);
    else if (nrg <= 8)
      stage_window_setup<T, 8, 2>(s.win, in_cg, plane_sz, gc, nrg, wsz, rpi, rsub, colc, gxc, y0, wh, wstride, H, W);
    else
      stage_window_setup<T, 16, 1>(s.win, in_cg, plane_sz, gc, nrg, wsz, rpi, rsub, colc, gxc, y0, wh, wstride, H, W);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nout = force_mode == 3 ? 0 : gc * PHW;
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
          acc += (hy * hx) * p[0] + (hy * lx) * p[1] + (ly * hx) * p[wstride] + (ly * lx) * p[wstride + 1];
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
template <typename T, int PHT, int PWT, int SRT>
__device__ __forceinline__ void roi_align_fwd_wave_fast(WaveShared& s, const T* __restrict__ input,
                                                        const T* __restrict__ rois, T* __restrict__ output,
                                                        int C, int H, int W, float spatial_scale, int aligned,
                                                        int k, int c0, int chunk, int force_mode) {
  constexpr int PHW = PHT * PWT;
  constexpr int NB = (PHW + 63) / 64;  // bins per lane
  constexpr int NS = SRT * SRT;        // samples per bin
  constexpr int ny = PHT * SRT, nx = PWT * SRT;
  static_assert(ny <= 64 && nx <= 64, "axis samples must fit one wave");
  const int lane = threadIdx.x & 63;
  const int cc = min(chunk, C - c0);
  RoiGeom<float> g = roi_geom<T, float>(rois + (int64_t)k * 5, spatial_scale, PHT, PWT, SRT, aligned != 0);
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
  int G = (ww <= 64 && nrg <= 16 && force_mode != 1) ? kWaveWin / wsz : 0;
  if (G > cc) G = cc;

  // ---- per-lane sample set-up (registers)
  int off[NB][NS];
  float fy[NB][SRT][2], fx[NB][SRT][2];  // {l, h} per axis sample
  int gy[NB][SRT], gxx[NB][SRT];         // absolute low indices (global-gather fallback)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int bin = min(lane + 64 * b, PHW - 1);
    const int ph = bin / PWT, pw = bin - ph * PWT;
    int ylo[SRT], xlo[SRT];
#pragma unroll
    for (int i = 0; i < SRT; ++i) {
      int hi;
      float l, h;
      const bool vy = axis_sample<float>(H, g.start_h, g.bin_h, SRT, ph, i, ylo[i], hi, l, h);
      fy[b][i][0] = l;
      fy[b][i][1] = h;
      if (!vy) ylo[i] = y0;
      gy[b][i] = ylo[i];
      const bool vx = axis_sample<float>(W, g.start_w, g.bin_w, SRT, pw, i, xlo[i], hi, l, h);
      fx[b][i][0] = l;
      fx[b][i][1] = h;
      if (!vx) xlo[i] = x0;
      gxx[b][i] = xlo[i];
    }
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
            acc = __builtin_fmaf(fy[b][iy][1], t0, acc);
            acc = __builtin_fmaf(fy[b][iy][0], t1, acc);
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
    if (nrg <= 2)
      stage_window_setup<T, 2, 8>(s.win, in_cg, plane_sz, gc, nrg, wsz, rpi, rsub, colc, gxc, y0, wh, wstride, H, W);
    else if (nrg <= 4)
      stage_window_setup<T, 4, 4>(s.win, in_cg, plane_sz, gc, nrg, wsz, rpi, rsub, colc, gxc, y0, wh, wstride, H, W);
    else if (nrg <= 8)
      stage_window_setup<T, 8, 2>(s.win, in_cg, plane_sz, gc, nrg, wsz, rpi, rsub, colc, gxc, y0, wh, wstride, H, W);
    else
      stage_window_setup<T, 16, 1>(s.win, in_cg, plane_sz, gc, nrg, wsz, rpi, rsub, colc, gxc, y0, wh, wstride, H, W);
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
            const float* p = wbase + off[b][iy * SRT + ix];
            const float t0 = __builtin_fmaf(fx[b][ix][0], p[1], fx[b][ix][1] * p[0]);
            const float t1 = __builtin_fmaf(fx[b][ix][0], p[wstride + 1], fx[b][ix][1] * p[wstride]);
            acc = __builtin_fmaf(fy[b][iy][1], t0, acc);
            acc = __builtin_fmaf(fy[b][iy][0], t1, acc);
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
constexpr int kDmaPerPass = 4;                   // DMA instructions (x 1 KiB) per pass and buffer
constexpr int kDmaBlk = 260;                     // LDS floats per DMA instruction block (256 + 4 skew, 16-B aligned)
constexpr int kDmaBuf = kDmaPerPass * kDmaBlk;   // floats per buffer

struct DmaShared {
  __attribute__((aligned(16))) float buf[2 * kDmaBuf];
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
struct DmaWindow {
  int state, y0, x0, wh, nq, lpr, rpi, nrg;
  int sparse;  // 1: the LDS image holds only the SAMPLED rows — slot 2s / 2s+1 = low / high tap row of y-sample s
};

// lo/l/h of one axis sample in the "shifted" form described above (dim >= 2)
__device__ __forceinline__ bool axis_sample_shifted(int dim, float start, float bin, int grid, int p, int i, int& lo,
                                                    float& l, float& h) {
  int hi;
  const bool v = axis_sample<float>(dim, start, bin, grid, p, i, lo, hi, l, h);
  if (!v) {
    lo = 0;
    l = h = 0.f;
    return false;
  }
  if (lo > dim - 2) {  // lo == dim-1: value is in[dim-1]
    lo = dim - 2;
    l = 1.f;
    h = 0.f;
  }
  return true;
}

// EPP = elements per 16-byte DMA piece (4 for fp32, 8 for fp16/bf16); nq then counts pieces.
template <int PHT, int PWT, int SRT, int EPP = 4>
__device__ __forceinline__ DmaWindow dma_window(const RoiGeom<float>& g, int H, int W) {
  constexpr int ny = PHT * SRT, nx = PWT * SRT;
  const int lane = threadIdx.x & 63;
  DmaWindow w;
  w.state = 2;
  w.y0 = w.x0 = w.wh = w.nq = w.lpr = w.rpi = w.nrg = w.sparse = 0;
  if (H < 2 || W < EPP) return w;
  int lo = 0, lo2 = 0;
  float l, h;
  bool v = false, v2 = false;
  if (lane < ny) v = axis_sample_shifted(H, g.start_h, g.bin_h, SRT, lane / SRT, lane % SRT, lo, l, h);
  if (lane < nx) v2 = axis_sample_shifted(W, g.start_w, g.bin_w, SRT, lane / SRT, lane % SRT, lo2, l, h);
  const unsigned long long by = __ballot(v), bx = __ballot(v2);
  if (by == 0ull || bx == 0ull) {  // every sample of one axis is outside the map: all outputs are zero
    w.state = 0;
    return w;
  }
  // sample coordinates are monotonic in the lane index, in either direction (malformed RoIs with
  // aligned=True have negative bins): the extremes sit at the first / last valid lane
  const int ya = __builtin_amdgcn_readlane(lo, __builtin_ctzll(by)), yb = __builtin_amdgcn_readlane(lo, 63 - __builtin_clzll(by));
  const int xa = __builtin_amdgcn_readlane(lo2, __builtin_ctzll(bx)), xb = __builtin_amdgcn_readlane(lo2, 63 - __builtin_clzll(bx));
  w.y0 = min(ya, yb);
  const int y1 = max(ya, yb) + 1;
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
  w.rpi = 64 / w.lpr;
  // A window taller than 2 rows per y-sample contains rows no sample touches (bins > 2 px): stage only the
  // sampled rows.  The kernel is bound by the bytes that cross L2 -> L1, and this drops ~1/3 of them on the
  // FPN workload (and pulls RoIs that were too tall for 8 DMA blocks per channel back onto this path).
  if (g_cfg_dma_sparse && w.wh > 2 * ny) {
    w.sparse = 1;
    w.wh = 2 * ny;
  }
  w.nrg = (w.wh + w.rpi - 1) / w.rpi;
  // 1 = DMA path (<= 8 DMA instructions per channel), 2 = register-staged / global-gather path
  w.state = w.nrg <= 2 * kDmaPerPass ? 1 : 2;
  return w;
}

constexpr int kDmaBlkBytes = kDmaBlk * 4;  // 1040

template <typename T, int NRG>
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

template <typename T, int PHT, int PWT, int SRT, int NRG>
__device__ __forceinline__ void roi_align_dma_passes(DmaShared& s, const T* __restrict__ in0, T* __restrict__ out,
                                                     int64_t plane_sz, int cc, int H, int W, const DmaWindow& dw,
                                                     const RoiGeom<float>& g,
                                                     const int (&off)[(PHT * PWT + 63) / 64][SRT * SRT][2],
                                                     const float (&fy)[(PHT * PWT + 63) / 64][SRT][2],
                                                     const float (&fx)[(PHT * PWT + 63) / 64][SRT][2]) {
  constexpr int PHW = PHT * PWT;
  constexpr int NB = (PHW + 63) / 64;
  constexpr int NS = SRT * SRT;
  constexpr int EPP = 16 / (int)sizeof(T);
  constexpr int BLK = kDmaBlkBytes / (int)sizeof(T);  // elements per DMA block incl. skew
  constexpr bool kDouble = NRG <= kDmaPerPass;        // 8 row groups use both buffers as one
  constexpr int G = kDouble ? kDmaPerPass / NRG : 1;
  constexpr bool kPow2 = (NS & (NS - 1)) == 0;
  const float inv_count = 1.f / (float)NS;
  const int lane = threadIdx.x & 63;
  const int rsub = min(lane / dw.lpr, dw.rpi - 1);
  const int q = min(lane - (lane / dw.lpr) * dw.lpr, dw.nq - 1);
  const int gx = dw.x0 + EPP * q;
  char* const bytes = reinterpret_cast<char*>(s.buf);
  int goff[NRG];
#pragma unroll
  for (int rg = 0; rg < NRG; ++rg) {
    const int t = min(rg * dw.rpi + rsub, dw.wh - 1);  // row slot of this lane in row group rg
    int y = dw.y0 + t;
    if (dw.sparse) {
      int lo;
      float l, h;
      axis_sample_shifted(H, g.start_h, g.bin_h, SRT, (t >> 1) / SRT, (t >> 1) % SRT, lo, l, h);
      y = lo + (t & 1);
    }
    goff[rg] = min(y, H - 1) * W + gx;
  }
  const int npass = (cc + G - 1) / G;
  if (kDouble) dma_issue_pass<T, NRG>(in0, plane_sz, min(G, cc), goff, bytes);
  for (int p = 0; p < npass; ++p) {
    const int cg = p * G;
    const int gc = min(G, cc - cg);
    const T* cur;
    if (kDouble) {
      if (p + 1 < npass) {
        dma_issue_pass<T, NRG>(in0 + (int64_t)(cg + G) * plane_sz, plane_sz, min(G, cc - cg - G), goff,
                               bytes + ((p + 1) & 1) * (kDmaBuf * 4));
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // everything older than those 4 DMAs has landed
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      cur = reinterpret_cast<const T*>(bytes + (p & 1) * (kDmaBuf * 4));
    } else {
      dma_issue_pass<T, NRG>(in0 + (int64_t)cg * plane_sz, plane_sz, gc, goff, bytes);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      cur = reinterpret_cast<const T*>(bytes);
    }
    for (int ch = 0; ch < gc; ++ch) {
      const T* wbase = cur + ch * (NRG * BLK);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float acc = 0.f;
#pragma unroll
        for (int iy = 0; iy < SRT; ++iy) {
#pragma unroll
          for (int ix = 0; ix < SRT; ++ix) {
            const T* q0 = wbase + off[b][iy * SRT + ix][0];
            const T* q1 = wbase + off[b][iy * SRT + ix][1];
            const float t0 = __builtin_fmaf(fx[b][ix][0], ld(q0 + 1), fx[b][ix][1] * ld(q0));
            const float t1 = __builtin_fmaf(fx[b][ix][0], ld(q1 + 1), fx[b][ix][1] * ld(q1));
            acc = __builtin_fmaf(fy[b][iy][1], t0, acc);
            acc = __builtin_fmaf(fy[b][iy][0], t1, acc);
          }
        }
        const int bin = lane + 64 * b;
        if (bin < PHW) {
          const float r = kPow2 ? acc * inv_count : acc / (float)NS;
          if constexpr (std::is_same<T, float>::value)
            __builtin_nontemporal_store(r, out + (cg + ch) * PHW + bin);
          else
            st(out + (cg + ch) * PHW + bin, r);
        }
      }
    }
    // the ds_reads above are complete (their results were consumed) before the next DMAs may
    // overwrite this buffer
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <typename T, int PHT, int PWT, int SRT>
__device__ __forceinline__ void roi_align_fwd_wave_dma(DmaShared& s, const T* __restrict__ input,
                                                       const T* __restrict__ rois, T* __restrict__ output,
                                                       int C, int H, int W, float spatial_scale, int aligned, int k,
                                                       int c0, int chunk, int* __restrict__ declined) {
  constexpr int PHW = PHT * PWT;
  constexpr int NB = (PHW + 63) / 64;
  constexpr int NS = SRT * SRT;
  constexpr int EPP = 16 / (int)sizeof(T);
  constexpr int BLK = kDmaBlkBytes / (int)sizeof(T);
  const int lane = threadIdx.x & 63;
  const int cc = min(chunk, C - c0);
  const RoiGeom<float> g = roi_geom<T, float>(rois + (int64_t)k * 5, spatial_scale, PHT, PWT, SRT, aligned != 0);
  T* out = output + ((int64_t)k * C + c0) * PHW;
  const T* in0 = input + ((int64_t)g.batch * C + c0) * H * W;
  const int64_t plane_sz = (int64_t)H * W;
  const DmaWindow dw = dma_window<PHT, PWT, SRT, EPP>(g, H, W);
  if (c0 == 0 && lane == 0) declined[k] = dw.state == 2;  // tells the fallback launch what is left
  if (dw.state == 2) return;
  if (dw.state == 0) {
    for (int o = lane; o < cc * PHW; o += 64) st(out + o, 0.f);
    return;
  }
  // ---- per-lane sample set-up (registers): LDS slots of the two tap rows + separable factors
  int off[NB][NS][2];
  float fy[NB][SRT][2], fx[NB][SRT][2];
  const int rstride = EPP * dw.lpr;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int bin = min(lane + 64 * b, PHW - 1);
    const int ph = bin / PWT, pw = bin - ph * PWT;
    int rlo[SRT][2], xlo[SRT];
#pragma unroll
    for (int i = 0; i < SRT; ++i) {
      int lo;
      float l, h;
      const bool vy = axis_sample_shifted(H, g.start_h, g.bin_h, SRT, ph, i, lo, l, h);
      fy[b][i][0] = l;
      fy[b][i][1] = h;
      const int r = dw.sparse ? 2 * (ph * SRT + i) : (vy ? lo - dw.y0 : 0);
      rlo[i][0] = (r / dw.rpi) * BLK + (r % dw.rpi) * rstride;
      rlo[i][1] = ((r + 1) / dw.rpi) * BLK + ((r + 1) % dw.rpi) * rstride;
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
  if (dw.nrg <= 1)
    roi_align_dma_passes<T, PHT, PWT, SRT, 1>(s, in0, out, plane_sz, cc, H, W, dw, g, off, fy, fx);
  else if (dw.nrg <= 2)
    roi_align_dma_passes<T, PHT, PWT, SRT, 2>(s, in0, out, plane_sz, cc, H, W, dw, g, off, fy, fx);
  else if (dw.nrg <= 4)
    roi_align_dma_passes<T, PHT, PWT, SRT, 4>(s, in0, out, plane_sz, cc, H, W, dw, g, off, fy, fx);
  else
    roi_align_dma_passes<T, PHT, PWT, SRT, 8>(s, in0, out, plane_sz, cc, H, W, dw, g, off, fy, fx);
}

template <typename T, int PHT, int PWT, int SRT>
__device__ __forceinline__ void roi_align_wave_dispatch(WaveShared& s, const T* __restrict__ input,
                                                        const T* __restrict__ rois, T* __restrict__ output, int C,
                                                        int H, int W, int PH_, int PW_, float spatial_scale, int sr_,
                                                        int aligned, int k, int c0, int chunk, int force_mode,
                                                        const int* __restrict__ declined) {
  if constexpr (PHT > 0 && PWT > 0 && SRT > 0) {
    // when the DMA launch ran first, this launch only mops up what it declined
    if (declined && declined[k] == 0) return;
    roi_align_fwd_wave_fast<T, PHT, PWT, SRT>(s, input, rois, output, C, H, W, spatial_scale, aligned, k, c0, chunk,
                                              force_mode);
  }
  else
    roi_align_fwd_wave_unit<T, PHT, PWT, SRT>(s, input, rois, output, C, H, W, PH_, PW_, spatial_scale, sr_, aligned,
                                              k, c0, chunk, force_mode);
}

// XCD-aware work placement.  Workgroup b is observed to run on XCD b % 8 (a speed assumption
// only — nothing depends on it for correctness).  Each XCD gets a CONTIGUOUS range of RoIs and
// walks it channel-chunk by channel-chunk, so at any moment an XCD's private 4 MiB L2 is asked
// for one thin channel slice of the feature maps (which fits) by ALL of its RoIs, and the
// overlap between neighbouring RoI windows turns into L2 hits instead of HBM re-reads.
__device__ __forceinline__ bool wave_unit(int64_t K, int nchunks, const int* __restrict__ order, int& k,
                                          int& chunk_idx, int wpb = kThreads / 64) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: keep unit math scalar
  const int xcd = blockIdx.x & 7;
  const int64_t j = blockIdx.x >> 3;
  const int64_t kbase = K >> 3, krem = K & 7;
  const int64_t Kx = kbase + (xcd < krem ? 1 : 0);
  const int64_t kstart = xcd * kbase + (xcd < krem ? xcd : krem);
  const int64_t local = j * wpb + wave;
  if (local >= Kx * nchunks) return false;
  chunk_idx = (int)(local / Kx);
  k = (int)(kstart + (local - (int64_t)chunk_idx * Kx));
  if (order) k = order[k];
  return true;
}

// Locality order of the RoIs: a stable-enough counting sort by (image, 32-pixel band of the box
// centre).  Cutting the sorted list into 8 contiguous ranges (one per XCD, see wave_unit) gives
// every XCD a horizontal band of one image, i.e. a slice of every feature map that fits its L2.
constexpr int kOrderBins = 4096;
template <typename T>
__global__ __launch_bounds__(1024) void roi_locality_order(const T* __restrict__ rois, int K, int* __restrict__ order) {
  __shared__ int hist[kOrderBins];
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < kOrderBins; i += 1024) hist[i] = 0;
  __syncthreads();
  auto bin_of = [&](int k) {
    const T* r = rois + (int64_t)k * 5;
    int b = (int)ld(r);
    const float cy = 0.5f * ((float)ld(r + 2) + (float)ld(r + 4));
    int band = (cy == cy) ? (int)fminf(fmaxf(cy * (1.f / 32.f), 0.f), 63.f) : 0;
    b = min(max(b, 0), 63);
    return b * 64 + band;
  };
  for (int k = tid; k < K; k += 1024) atomicAdd(&hist[bin_of(k)], 1);
  __syncthreads();
  // exclusive scan of 4096 bins: 4 per thread + block scan
  int loc[4], sum = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    loc[q] = sum;
    sum += hist[tid * 4 + q];
  }
  part[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  const int base = part[tid] - sum;
#pragma unroll
  for (int q = 0; q < 4; ++q) hist[tid * 4 + q] = base + loc[q];
  __syncthreads();
  for (int k = tid; k < K; k += 1024) order[atomicAdd(&hist[bin_of(k)], 1)] = k;
}

inline unsigned wave_unit_grid(int64_t K, int nchunks, int wpb = kThreads / 64) {
  const int64_t per_xcd = ceil_div(ceil_div(K, 8) * nchunks, wpb);
  return (unsigned)(8 * per_xcd);
}

template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_wave(const T* __restrict__ input, const T* __restrict__ rois,
                                                               T* __restrict__ output, int C, int H, int W, int PH_,
                                                               int PW_, float spatial_scale, int sr_, int aligned,
                                                               int nchunks, int chunk, int64_t nunits,
                                                               const int* __restrict__ declined,
                                                               const int* __restrict__ order) {
  __shared__ WaveShared s[kThreads / 64];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: keep unit math scalar
  int k, ci;
  if (!wave_unit(nunits / nchunks, nchunks, order, k, ci)) return;
  const int c0 = ci * chunk;
  roi_align_wave_dispatch<T, PHT, PWT, SRT>(s[wave], input, rois, output, C, H, W, PH_, PW_, spatial_scale, sr_,
                                            aligned, k, c0, chunk, g_roi_force_mode, declined);
}

template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_dma(const T* __restrict__ input,
                                                              const T* __restrict__ rois, T* __restrict__ output,
                                                              int C, int H, int W, float spatial_scale, int aligned,
                                                              int nchunks, int chunk, int64_t nunits, int* __restrict__ declined,
                                                              const int* __restrict__ order) {
  __shared__ DmaShared s[kThreads / 64];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: keep unit math scalar
  int k, ci;
  if (!wave_unit(nunits / nchunks, nchunks, order, k, ci)) return;
  roi_align_fwd_wave_dma<T, PHT, PWT, SRT>(s[wave], input, rois, output, C, H, W, spatial_scale, aligned, k, ci * chunk,
                                        chunk, declined);
}

template <typename T, int PHT, int PWT, int SRT, int WPB = kThreads / 64>
__global__ __launch_bounds__(64 * WPB) void roi_align_fwd_ms_dma(MsLevels lv, const T* __restrict__ rois,
                                                                 T* __restrict__ output, int C, int aligned,
                                                                 int nchunks, int chunk, int64_t nunits, int* __restrict__ declined,
                                                                 const int* __restrict__ order) {
  __shared__ DmaShared s[WPB];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: keep unit math scalar
  int k, ci;
  if (!wave_unit(nunits / nchunks, nchunks, order, k, ci, WPB)) return;
  const int l = fpn_level<T>(rois + (int64_t)k * 5, lv);
  roi_align_fwd_wave_dma<T, PHT, PWT, SRT>(s[wave], static_cast<const T*>(lv.ptr[l]), rois, output, C, lv.H[l], lv.W[l],
                                        lv.scale[l], aligned, k, ci * chunk, chunk, declined);
}

template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_ms_wave(MsLevels lv, const T* __restrict__ rois,
                                                                  T* __restrict__ output, int C, int PH_, int PW_,
                                                                  int sr_, int aligned, int nchunks, int chunk,
                                                                  int64_t nunits, const int* __restrict__ declined,
                                                                  const int* __restrict__ order) {
  __shared__ WaveShared s[kThreads / 64];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: keep unit math scalar
  int k, ci;
  if (!wave_unit(nunits / nchunks, nchunks, order, k, ci)) return;
  const int c0 = ci * chunk;
  const int l = fpn_level<T>(rois + (int64_t)k * 5, lv);
  roi_align_wave_dispatch<T, PHT, PWT, SRT>(s[wave], static_cast<const T*>(lv.ptr[l]), rois, output, C, lv.H[l],
                                            lv.W[l], PH_, PW_, lv.scale[l], sr_, aligned, k, c0, chunk,
                                            g_roi_force_mode, declined);
}

// ---------------------------------------------------------------------------------------
// Backward.  cpu/roi_align_kernel.cpp:183-289: every sample adds grad*w_i/count to its 4
// taps.  Generic version: one thread per grad element, global atomics.
template <typename T>
__global__ __launch_bounds__(kThreads) void roi_align_bwd_generic(
    const T* __restrict__ grad, const T* __restrict__ rois, T* __restrict__ grad_input,
    int64_t total, int C, int H, int W, int PH, int PW, double spatial_scale, int sr, int aligned,
    int64_t ns, int64_t cs, int64_t hs, int64_t ws) {
  using A = typename Acc<T>::type;
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * kThreads) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / ((int64_t)PW * PH)) % C);
    const int64_t k = idx / ((int64_t)PW * PH * C);
    const RoiGeom<A> g = roi_geom<T, A>(rois + k * 5, (A)spatial_scale, PH, PW, sr, aligned != 0);
    T* plane = grad_input + ((int64_t)g.batch * C + c) * H * W;
    const A go = ld(grad + k * ns + c * cs + ph * hs + pw * ws);
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
        atomic_accum(plane + (int64_t)ylo * W + xlo, go * w1 / g.count);
        atomic_accum(plane + (int64_t)ylo * W + xhi, go * w2 / g.count);
        atomic_accum(plane + (int64_t)yhi * W + xlo, go * w3 / g.count);
        atomic_accum(plane + (int64_t)yhi * W + xhi, go * w4 / g.count);
      }
    }
  }
}

// Tiled backward: accumulate a channel group's window in LDS, flush once per pixel.
template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) void roi_align_bwd_tile(
    const T* __restrict__ grad, const T* __restrict__ rois, T* __restrict__ grad_input, int C, int H,
    int W, int PH_, int PW_, float spatial_scale, int sr_, int aligned, int nchunks, int64_t ns,
    int64_t cs, int64_t hs, int64_t ws, const int* __restrict__ declined, MsLevels lv, int use_ms) {
  __shared__ TileShared s;
  const int PH = PHT > 0 ? PHT : PH_;
  const int PW = PWT > 0 ? PWT : PW_;
  const int sr = SRT > 0 ? SRT : sr_;
  const int PHW = PH * PW;
  const int tid = threadIdx.x;
  const int k = blockIdx.x / nchunks;
  if (declined && declined[k] == 0) return;  // this RoI was handled by the wave kernel
  if (use_ms) {  // multi-scale form: the RoI picks its level's gradient map
    const int l = fpn_level<T>(rois + (int64_t)k * 5, lv);
    grad_input = static_cast<T*>(const_cast<void*>(lv.ptr[l]));
    H = lv.H[l];
    W = lv.W[l];
    spatial_scale = lv.scale[l];
  }
  const int c0 = (blockIdx.x - k * nchunks) * kChunk;
  const int cc = min(kChunk, C - c0);

  RoiGeom<float> g = roi_geom<T, float>(rois + (int64_t)k * 5, spatial_scale, PH, PW, sr, aligned != 0);
  if (SRT > 0) {
    g.gh = SRT;
    g.gw = SRT;
  }
  const T* gk = grad + (int64_t)k * ns + (int64_t)c0 * cs;
  T* gi0 = grad_input + ((int64_t)g.batch * C + c0) * H * W;
  const int64_t plane_sz = (int64_t)H * W;

  int y0 = 0, x0 = 0, wh = 0, ww = 0, wstride = 0, G = 0;
  const int mode = build_tables<T>(s, g, H, W, PH, PW, cc, y0, x0, wh, ww, wstride, G);
  if (mode == MODE_ZERO) return;
  const float count = g.count;
  const int gh = g.gh, gw = g.gw;

  if (mode == MODE_GEN || mode == MODE_TAB) {
    for (int o = tid; o < cc * PHW; o += kThreads) {
      const int c = o / PHW, bin = o - c * PHW;
      const int ph = bin / PW, pw = bin - ph * PW;
      T* plane = gi0 + c * plane_sz;
      const float go = ld(gk + c * cs + ph * hs + pw * ws);
      for (int iy = 0; iy < gh; ++iy) {
        int ylo, yhi;
        float ly, hy;
        const bool vy = axis_sample<float>(H, g.start_h, g.bin_h, gh, ph, iy, ylo, yhi, ly, hy);
        for (int ix = 0; ix < gw; ++ix) {
          int xlo, xhi;
          float lx, hx;
          const bool vx = axis_sample<float>(W, g.start_w, g.bin_w, gw, pw, ix, xlo, xhi, lx, hx);
          if (!(vy && vx)) continue;
          atomic_accum(plane + (int64_t)ylo * W + xlo, go * (hy * hx) / count);
          atomic_accum(plane + (int64_t)ylo * W + xhi, go * (hy * lx) / count);
          atomic_accum(plane + (int64_t)yhi * W + xlo, go * (ly * hx) / count);
          atomic_accum(plane + (int64_t)yhi * W + xhi, go * (ly * lx) / count);
        }
      }
    }
    return;
  }

  const int wsz = wh * wstride;
  const int wpix = wh * ww;
  FastDiv16 div_wpix, div_ww;
  div_wpix.init((unsigned)wpix);
  div_ww.init((unsigned)ww);
  for (int cg = 0; cg < cc; cg += G) {
    const int gc = min(G, cc - cg);
    for (int e = tid; e < gc * wsz; e += kThreads) s.win[e] = 0.f;
    __syncthreads();
    const int nout = gc * PHW;
    for (int o = tid; o < nout; o += kThreads) {
      const int ch = o / PHW, bin = o - ch * PHW;
      const int ph = bin / PW, pw = bin - ph * PW;
      const float go = ld(gk + (int64_t)(cg + ch) * cs + ph * hs + pw * ws);
      float* wbase = s.win + ch * wsz;
      for (int iy = 0; iy < gh; ++iy) {
        const int ty = ph * gh + iy;
        float* row = wbase + s.y.lo[ty];
        const float ly = s.y.l[ty], hy = s.y.h[ty];
        for (int ix = 0; ix < gw; ++ix) {
          const int tx = pw * gw + ix;
          float* p = row + s.x.lo[tx];
          const float lx = s.x.l[tx], hx = s.x.h[tx];
          atomicAdd(p, go * (hy * hx) / count);
          atomicAdd(p + 1, go * (hy * lx) / count);
          atomicAdd(p + wstride, go * (ly * hx) / count);
          atomicAdd(p + wstride + 1, go * (ly * lx) / count);
        }
      }
    }
    __syncthreads();
    const int total = gc * wpix;
    for (int e = tid; e < total; e += kThreads) {
      const int ch = (int)div_wpix.div((unsigned)e);
      const int rem = e - ch * wpix;
      const int r = (int)div_ww.div((unsigned)rem);
      const int col = rem - r * ww;
      const int gy = y0 + r, gx = x0 + col;
      const float v = s.win[ch * wsz + r * wstride + col];
      if (gy < H && gx < W && v != 0.f) atomic_accum(gi0 + (cg + ch) * plane_sz + (int64_t)gy * W + gx, v);
    }
    __syncthreads();
  }
}

// debug / tuning knobs (tvmi_debug_set): [0] kernel variant 0 = block-tiled, 1 = wave-autonomous;
// [1] channels per unit for the wave variant; [2] force mode (device side)
int g_cfg_variant = 1;
int g_cfg_chunk = 32;
int g_cfg_dma = 1;
int g_cfg_order = 0;
int g_cfg_bwd_dense = 1;
int g_cfg_dma_wpb = 4;
int g_cfg_dma_extra_lds = 0;  // analysis knob: unused dynamic LDS per workgroup (lowers occupancy)

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
static void load_env_cfg() {
  static bool done = false;
  if (done) return;
  done = true;
  g_cfg_variant = env_int("TVMI_ROI_VARIANT", g_cfg_variant);
  g_cfg_chunk = env_int("TVMI_ROI_CHUNK", g_cfg_chunk);
  g_cfg_dma = env_int("TVMI_ROI_DMA", g_cfg_dma);
  g_cfg_order = env_int("TVMI_ROI_ORDER", g_cfg_order);
  g_cfg_bwd_dense = env_int("TVMI_ROI_BWD_DENSE", g_cfg_bwd_dense);
  g_cfg_dma_wpb = env_int("TVMI_ROI_DMA_WPB", g_cfg_dma_wpb);
  {
    const int v = env_int("TVMI_ROI_DMA_SPARSE", -1);
    if (v >= 0) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cfg_dma_sparse), &v, sizeof(int));
  }
  g_cfg_dma_extra_lds = env_int("TVMI_ROI_DMA_EXTRA_LDS", g_cfg_dma_extra_lds);
}

// ---------------------------------------------------------------------------------------
// Wave-autonomous backward (fp32, compile-time shapes): same unit decomposition and LDS window
// image as the DMA forward.  Lane = bin with its 4 samples in registers; per channel the wave
// loads the 49 (196) grads coalesced, scatters the 16 tap contributions of every bin into the
// zeroed LDS window with ds_add_f32, then flushes the window ROW-WISE: one lane = 4 consecutive
// pixels, so the global float atomics of a wave hit whole row segments (few cache lines per
// instruction) and there is ONE atomic per touched pixel per RoI instead of 4 per sample
// (reference: cuda/roi_align_kernel.cu:304-327).
template <int PHT, int PWT, int SRT, int NRG>
__device__ __forceinline__ void roi_align_bwd_passes(DmaShared& s, const float* __restrict__ gk, float* __restrict__ gi0,
                                                     int64_t plane_sz, int cc, int H, int W, const DmaWindow& dw,
                                                     int64_t cs, const int (&goffs)[(PHT * PWT + 63) / 64],
                                                     const int (&off)[(PHT * PWT + 63) / 64][SRT * SRT][2],
                                                     const float (&fy)[(PHT * PWT + 63) / 64][SRT][2],
                                                     const float (&fx)[(PHT * PWT + 63) / 64][SRT][2]) {
  constexpr int PHW = PHT * PWT;
  constexpr int NB = (PHW + 63) / 64;
  constexpr int NS = SRT * SRT;
  constexpr int G = NRG <= 2 * kDmaPerPass ? (2 * kDmaPerPass) / NRG : 1;  // channels per pass (both buffers)
  const float inv_count = 1.f / (float)NS;
  const int lane = threadIdx.x & 63;
  const int rsub = lane / dw.lpr, q = lane - rsub * dw.lpr;
  const bool flush_lane = rsub < dw.rpi && q < dw.nq;
  const int gx = dw.x0 + 4 * q;
  for (int cg = 0; cg < cc; cg += G) {
    const int gc = min(G, cc - cg);
    // zero the pass' window images
    for (int i = lane; i < gc * NRG * (kDmaBlk / 4); i += 64) reinterpret_cast<float4*>(s.buf)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int ch = 0; ch < gc; ++ch) {
      float* wbase = s.buf + ch * (NRG * kDmaBlk);
      const float* gch = gk + (int64_t)(cg + ch) * cs;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int bin = lane + 64 * b;
        if (bin >= PHW) continue;
        const float gsc = gch[goffs[b]] * inv_count;  // count is 4 here: exact
#pragma unroll
        for (int iy = 0; iy < SRT; ++iy) {
          const float a = gsc * fy[b][iy][1], bb = gsc * fy[b][iy][0];
#pragma unroll
          for (int ix = 0; ix < SRT; ++ix) {
            float* q0 = wbase + off[b][iy * SRT + ix][0];
            float* q1 = wbase + off[b][iy * SRT + ix][1];
            atomicAdd(q0, a * fx[b][ix][1]);
            atomicAdd(q0 + 1, a * fx[b][ix][0]);
            atomicAdd(q1, bb * fx[b][ix][1]);
            atomicAdd(q1 + 1, bb * fx[b][ix][0]);
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // flush: lane (rsub, q) owns 4 consecutive pixels of window row rg*rpi+rsub
    if (flush_lane) {
      for (int ch = 0; ch < gc; ++ch) {
        float* plane = gi0 + (int64_t)(cg + ch) * plane_sz;
#pragma unroll
        for (int rg = 0; rg < NRG; ++rg) {
          const int r = rg * dw.rpi + rsub;
          if (r < dw.wh) {
            const float4 v = *reinterpret_cast<const float4*>(s.buf + (ch * NRG + rg) * kDmaBlk + rsub * 4 * dw.lpr + 4 * q);
            float* dst = plane + (int64_t)(dw.y0 + r) * W + gx;
            if (v.x != 0.f) unsafeAtomicAdd(dst, v.x);
            if (v.y != 0.f) unsafeAtomicAdd(dst + 1, v.y);
            if (v.z != 0.f) unsafeAtomicAdd(dst + 2, v.z);
            if (v.w != 0.f) unsafeAtomicAdd(dst + 3, v.w);
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ---------------------------------------------------------------------------------------
// Dense-separable backward (fp32, compile-time shapes) — no LDS atomics.
// ds_add_f32 turned out to be the bottleneck of every scatter-style backward on this chip
// (measured ~3 cycles per lane-operation per CU: 4.5-7 ms on config 2 in three different
// kernels).  The gradient of a RoI window is a separable product,
//     dWindow[r][c] = sum_ph AyD[r][ph] * ( sum_pw G[ph][pw] * AxD[pw][c] ),
// with AyD / AxD the (tiny, mostly zero) matrices of bilinear row / column factors summed over
// the samples of a bin and divided by count.  A wave handles (RoI, channel chunk): lane = window
// column (several channels side by side when the window is narrow), AxD's column lives in
// registers, AyD rows and the grads sit in LDS and are read as broadcasts; every window pixel
// is produced by exactly one lane with plain FMAs and leaves with ONE global float atomic,
// issued as contiguous row segments.
constexpr int kDenseMaxWH = 48;

template <int PHT, int PWT, int SRT>
struct DenseShared {
  static constexpr int PHP = (PHT + 3) & ~3;   // padded to float4
  static constexpr int PWP = (PWT + 3) & ~3;
  __attribute__((aligned(16))) float ayd[kDenseMaxWH][PHP];
  __attribute__((aligned(16))) float gs[8][PHT][PWP];  // up to 8 channels side by side
};

template <int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) void roi_align_bwd_dense(const float* __restrict__ grad, const float* __restrict__ rois,
                                                                float* __restrict__ grad_input, int C, int H, int W,
                                                                float spatial_scale, int aligned, int nchunks, int chunk,
                                                                int64_t nunits, int64_t ns, int64_t cs, int64_t hs,
                                                                int64_t ws, int* __restrict__ declined, MsLevels lv,
                                                                int use_ms) {
  using DS = DenseShared<PHT, PWT, SRT>;
  __shared__ DS sh[kThreads / 64];
  constexpr int ny = PHT * SRT, nx = PWT * SRT;
  constexpr int NS = SRT * SRT;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  int k, ci;
  if (!wave_unit(nunits / nchunks, nchunks, nullptr, k, ci)) return;
  const int c0 = ci * chunk;
  const int cc = min(chunk, C - c0);
  if (use_ms) {  // multi-scale form: the RoI picks its level's gradient map (poolers.py:73-84)
    const int l = fpn_level<float>(rois + (int64_t)k * 5, lv);
    grad_input = static_cast<float*>(const_cast<void*>(lv.ptr[l]));
    H = lv.H[l];
    W = lv.W[l];
    spatial_scale = lv.scale[l];
  }
  const RoiGeom<float> g = roi_geom<float, float>(rois + (int64_t)k * 5, spatial_scale, PHT, PWT, SRT, aligned != 0);
  // ---- window bounds (shifted samples: lo + 1 is always inside the map)
  int y0 = 0, wh = 0, x0 = 0, ww = 0, state = 2;
  if (H >= 2 && W >= 2) {
    int lo = 0, lo2 = 0;
    float l, h;
    bool v = false, v2 = false;
    if (lane < ny) v = axis_sample_shifted(H, g.start_h, g.bin_h, SRT, lane / SRT, lane % SRT, lo, l, h);
    if (lane < nx) v2 = axis_sample_shifted(W, g.start_w, g.bin_w, SRT, lane / SRT, lane % SRT, lo2, l, h);
    const unsigned long long by = __ballot(v), bx = __ballot(v2);
    if (by == 0ull || bx == 0ull) {
      state = 0;  // no sample inside the map: no gradient
    } else {
      const int ya = __builtin_amdgcn_readlane(lo, __builtin_ctzll(by)), yb = __builtin_amdgcn_readlane(lo, 63 - __builtin_clzll(by));
      const int xa = __builtin_amdgcn_readlane(lo2, __builtin_ctzll(bx)), xb = __builtin_amdgcn_readlane(lo2, 63 - __builtin_clzll(bx));
      y0 = min(ya, yb);
      wh = max(ya, yb) + 1 - y0 + 1;
      x0 = min(xa, xb);
      ww = max(xa, xb) + 1 - x0 + 1;
      state = (wh <= kDenseMaxWH && ww <= 64) ? 1 : 2;
    }
  }
  if (c0 == 0 && lane == 0) declined[k] = state == 2;
  if (state != 1) return;
  int wpad = 4;
  while (wpad < ww) wpad <<= 1;
  const int cpl = min(64 / wpad, 8);  // channels processed side by side
  const int slot = lane / wpad, col = lane - slot * wpad;
  const bool col_ok = col < ww && slot < cpl;
  const float inv_count = 1.f / (float)NS;
  DS& s = sh[wave];
  // ---- AxD column of this lane (registers) and AyD rows (LDS), both already divided by count
  float axd[PWT];
  {
    const int gcol = x0 + col;
#pragma unroll
    for (int pw = 0; pw < PWT; ++pw) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < SRT; ++i) {
        int lo;
        float l, h;
        if (axis_sample_shifted(W, g.start_w, g.bin_w, SRT, pw, i, lo, l, h)) {
          if (lo == gcol) a += h;
          if (lo + 1 == gcol) a += l;
        }
      }
      axd[pw] = col_ok ? a : 0.f;
    }
    if (lane < wh) {
      const int grow = y0 + lane;
#pragma unroll
      for (int ph = 0; ph < DS::PHP; ++ph) {
        float a = 0.f;
        if (ph < PHT) {
#pragma unroll
          for (int i = 0; i < SRT; ++i) {
            int lo;
            float l, h;
            if (axis_sample_shifted(H, g.start_h, g.bin_h, SRT, ph, i, lo, l, h)) {
              if (lo == grow) a += h;
              if (lo + 1 == grow) a += l;
            }
          }
        }
        s.ayd[lane][ph] = a * inv_count;
      }
    }
  }
  const float* gk = grad + (int64_t)k * ns + (int64_t)c0 * cs;
  float* gi0 = grad_input + ((int64_t)g.batch * C + c0) * H * W;
  const int64_t plane_sz = (int64_t)H * W;
  for (int cg = 0; cg < cc; cg += cpl) {
    const int gc = min(cpl, cc - cg);
    // 1. grads of gc channels -> LDS (coalesced within a channel)
    for (int e = lane; e < gc * PHT * PWT; e += 64) {
      const int ch = e / (PHT * PWT), bin = e - ch * (PHT * PWT);
      const int ph = bin / PWT, pw = bin - ph * PWT;
      s.gs[ch][ph][pw] = gk[(int64_t)(cg + ch) * cs + ph * hs + pw * ws];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // 2. t[ph] = sum_pw G[ph][pw] * AxD[pw][col]
    float t[PHT];
    const int myslot = min(slot, gc - 1);
#pragma unroll
    for (int ph = 0; ph < PHT; ++ph) {
      float a = 0.f;
#pragma unroll
      for (int pw = 0; pw < PWT; ++pw) a = __builtin_fmaf(s.gs[myslot][ph][pw], axd[pw], a);
      t[ph] = a;
    }
    // 3. window rows: one global atomic per pixel, contiguous along the row
    const bool write_ok = col_ok && slot < gc;
    float* plane = gi0 + (int64_t)(cg + myslot) * plane_sz + (int64_t)y0 * W + x0 + col;
    for (int r = 0; r < wh; ++r) {
      float v = 0.f;
#pragma unroll
      for (int ph = 0; ph < PHT; ++ph) v = __builtin_fmaf(s.ayd[r][ph], t[ph], v);
      if (write_ok && v != 0.f) unsafeAtomicAdd(plane + (int64_t)r * W, v);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

template <typename T>
int launch_fwd(const void* input, const void* rois, void* output, int64_t N, int64_t C, int64_t H,
               int64_t W, int64_t K, int64_t PH, int64_t PW, double scale, int64_t sr, int aligned,
               int* order, int* declined, hipStream_t stream) {
  load_env_cfg();
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
    const bool wavev = g_cfg_variant == 1;
    if (order && wavev && g_cfg_order) {
      roi_locality_order<T><<<dim3(1), dim3(1024), 0, stream>>>(r, (int)K, order);
    } else {
      order = nullptr;
    }
    const int chunk = wavev ? g_cfg_chunk : kChunk;
    const int nchunks = (int)ceil_div(C, chunk);
    const int64_t nunits = K * nchunks;
    const dim3 grid(wavev ? wave_unit_grid(K, nchunks) : (unsigned)nunits), block(kThreads);
    // fallback launch (what the DMA launch declined): coarser units, it is almost always empty
    const int fb_chunk = 64, fb_nchunks = (int)ceil_div(C, fb_chunk);
    const int64_t fb_nunits = K * fb_nchunks;
    const dim3 fb_grid(wave_unit_grid(K, fb_nchunks));
#define TVMI_FWD(PHT, PWT, SRT)                                                                              \
  if (wavev) {                                                                                               \
    bool dma = false;                                                                                        \
    if constexpr ((PHT) > 0) {                                                                               \
      if (g_cfg_dma && declined) {                                                                           \
        roi_align_fwd_dma<T, PHT, PWT, SRT><<<grid, block, 0, stream>>>(in, r, out, (int)C, (int)H, (int)W, fs, \
                                                                     aligned, nchunks, chunk, nunits, declined, order); \
        dma = true;                                                                                          \
      }                                                                                                      \
    }                                                                                                        \
    if (dma)                                                                                                 \
      roi_align_fwd_wave<T, PHT, PWT, SRT><<<fb_grid, block, 0, stream>>>(                                   \
          in, r, out, (int)C, (int)H, (int)W, (int)PH, (int)PW, fs, (int)sr, aligned, fb_nchunks, fb_chunk,  \
          fb_nunits, declined, order);                                                                       \
    else                                                                                                     \
      roi_align_fwd_wave<T, PHT, PWT, SRT><<<grid, block, 0, stream>>>(in, r, out, (int)C, (int)H, (int)W,   \
                                                                       (int)PH, (int)PW, fs, (int)sr, aligned, \
                                                                       nchunks, chunk, nunits, nullptr, order); \
  } else                                                                                                     \
    roi_align_fwd_tile<T, PHT, PWT, SRT><<<grid, block, 0, stream>>>(in, r, out, (int)C, (int)H, (int)W,     \
                                                                     (int)PH, (int)PW, fs, (int)sr, aligned, nchunks)
    if (PH == 7 && PW == 7 && sr == 2) {
      TVMI_FWD(7, 7, 2);
    } else if (PH == 14 && PW == 14 && sr == 2) {
      TVMI_FWD(14, 14, 2);
    } else {
      TVMI_FWD(0, 0, 0);
    }
#undef TVMI_FWD
  }
  TVMI_RETURN_LAUNCH_STATUS("tvmi_roi_align_forward");
}

template <typename T>
int launch_bwd(const void* grad, const void* rois, void* grad_input, int64_t N, int64_t C, int64_t H,
               int64_t W, int64_t K, int64_t PH, int64_t PW, double scale, int64_t sr, int aligned,
               int64_t ns, int64_t cs, int64_t hs, int64_t ws, int* declined, hipStream_t stream,
               const MsLevels* ms = nullptr) {
  load_env_cfg();
  const T* g = static_cast<const T*>(grad);
  const T* r = static_cast<const T*>(rois);
  T* gi = static_cast<T*>(grad_input);
  const int64_t total = K * C * PH * PW;
  MsLevels lv{};
  const int use_ms = ms != nullptr;
  if (ms) lv = *ms;
  if constexpr (std::is_same<T, double>::value) {
    if (ms) return set_error((int)hipErrorInvalidValue, "multiscale_roi_align_backward: float64 is not supported");
    const int64_t blocks = std::min<int64_t>(ceil_div(total, kThreads), 1 << 20);
    roi_align_bwd_generic<T><<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(
        g, r, gi, total, (int)C, (int)H, (int)W, (int)PH, (int)PW, scale, (int)sr, aligned, ns, cs,
        hs, ws);
  } else {
    const int nchunks = (int)ceil_div(C, kChunk);
    const dim3 grid((unsigned)(K * nchunks)), block(kThreads);
    const float fs = (float)scale;
    const int wchunk = g_cfg_chunk, wnchunks = (int)ceil_div(C, wchunk);
    const int64_t wnunits = K * wnchunks;
#define TVMI_BWD(PHT, PWT, SRT)                                                                            \
  {                                                                                                        \
    const int* dflags = nullptr;                                                                           \
    if constexpr (std::is_same<T, float>::value && (PHT) > 0) {                                            \
      if (g_cfg_bwd_dense && declined && H * W * C < (1ll << 31)) {                                        \
        roi_align_bwd_dense<PHT, PWT, SRT><<<dim3(wave_unit_grid(K, wnchunks)), block, 0, stream>>>(       \
            g, r, gi, (int)C, (int)H, (int)W, fs, aligned, wnchunks, wchunk, wnunits, ns, cs, hs, ws, declined, lv, use_ms); \
        dflags = declined;                                                                                 \
      }                                                                                                    \
    }                                                                                                      \
    roi_align_bwd_tile<T, PHT, PWT, SRT><<<grid, block, 0, stream>>>(g, r, gi, (int)C, (int)H, (int)W,     \
                                                                     (int)PH, (int)PW, fs, (int)sr, aligned, \
                                                                     nchunks, ns, cs, hs, ws, dflags, lv, use_ms); \
  }
    if (PH == 7 && PW == 7 && sr == 2) {
      TVMI_BWD(7, 7, 2);
    } else if (PH == 14 && PW == 14 && sr == 2) {
      TVMI_BWD(14, 14, 2);
    } else {
      TVMI_BWD(0, 0, 0);
    }
#undef TVMI_BWD
  }
  TVMI_RETURN_LAUNCH_STATUS("tvmi_roi_align_backward");
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

template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_nhwc(MsLevels lv, const T* __restrict__ rois,
                                                               T* __restrict__ output, int C, int aligned,
                                                               int ngroups, int64_t nunits) {
  constexpr int PHW = PHT * PWT;
  constexpr int NS = SRT * SRT;
  constexpr int CPL = 4 / (int)sizeof(T);  // channels per lane: every tap is one 32-bit load per lane
  constexpr int GC = 64 * CPL;             // channels per unit
  __shared__ NhwcShared<T, PHT, PWT, SRT> sh[kThreads / 64];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  int k, gi;
  if (!wave_unit(nunits / ngroups, ngroups, nullptr, k, gi)) return;
  const int c0 = gi * GC;
  // level / batch index are wave-uniform: say so, the tap base then lives in SGPRs
  const int l = __builtin_amdgcn_readfirstlane(fpn_level<T>(rois + (int64_t)k * 5, lv));
  const int H = lv.H[l], W = lv.W[l];
  const RoiGeom<float> g = roi_geom<T, float>(rois + (int64_t)k * 5, lv.scale[l], PHT, PWT, SRT, aligned != 0);
  const int batch = __builtin_amdgcn_readfirstlane(g.batch);
  // ---- per-RoI sample tables, one sample per lane: element offset of the low tap + the two factors
  int yoff = 0, xoff = 0;
  float yl = 0.f, yh = 0.f, xl = 0.f, xh = 0.f;
  if (lane < PHT * SRT) {
    int lo;
    axis_sample_shifted(H, g.start_h, g.bin_h, SRT, lane / SRT, lane % SRT, lo, yl, yh);
    yoff = lo * W * C;
  }
  if (lane < PWT * SRT) {
    int lo;
    axis_sample_shifted(W, g.start_w, g.bin_w, SRT, lane / SRT, lane % SRT, lo, xl, xh);
    xoff = lo * C;
  }
  const int rowC = W * C;
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
  for (int ph = 0; ph < PHT; ++ph) {
    float acc[PWT][CPL];
#pragma unroll
    for (int pw = 0; pw < PWT; ++pw)
#pragma unroll
      for (int e = 0; e < CPL; ++e) acc[pw][e] = 0.f;
#pragma unroll
    for (int iy = 0; iy < SRT; ++iy) {
      const int sy = ph * SRT + iy;
      const int r0 = __builtin_amdgcn_readlane(yoff, sy);
      const float hy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yh), sy));
      const float ly = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yl), sy));
      const T* row0 = nbase + r0;
      const T* row1 = row0 + rowC;
#pragma unroll
      for (int j = 0; j < PWT * SRT; ++j) {
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
          const float t0 = __builtin_fmaf(slx[j], v01[e], shx[j] * v00[e]);
          const float t1 = __builtin_fmaf(slx[j], v11[e], shx[j] * v10[e]);
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
int launch_ms_fwd_nhwc(const MsLevels& lv, const void* rois, void* output, int64_t C, int64_t K, int64_t PH, int64_t PW,
                       int64_t sr, int aligned, hipStream_t stream) {
  load_env_cfg();
  constexpr int GC = 64 * (4 / (int)sizeof(T));
  const int ngroups = (int)ceil_div(C, GC);
  const int64_t nunits = K * ngroups;
  const dim3 grid(wave_unit_grid(K, ngroups)), block(kThreads);
  if (!(PH == 7 && PW == 7 && sr == 2))
    return set_error((int)hipErrorInvalidValue,
                     "roi_align (channels_last): only 7x7 bins with sampling_ratio 2 have a native NHWC kernel");
  roi_align_fwd_nhwc<T, 7, 7, 2><<<grid, block, 0, stream>>>(lv, static_cast<const T*>(rois), static_cast<T*>(output), (int)C,
                                                             aligned, ngroups, nunits);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_multiscale_roi_align_forward_nhwc");
}

template <typename T>
int launch_ms_fwd(const tvmi::MsLevels& lv, const void* rois, void* output, int64_t C, int64_t K, int64_t PH,
                  int64_t PW, int64_t sr, int aligned, int* order, int* declined, hipStream_t stream) {
  load_env_cfg();
  const T* r = static_cast<const T*>(rois);
  T* out = static_cast<T*>(output);
  const bool wavev = g_cfg_variant == 1;
  if (order && wavev && g_cfg_order) {
    roi_locality_order<T><<<dim3(1), dim3(1024), 0, stream>>>(r, (int)K, order);
  } else {
    order = nullptr;
  }
  const int chunk = wavev ? g_cfg_chunk : kChunk;
  const int nchunks = (int)ceil_div(C, chunk);
  const int64_t nunits = K * nchunks;
  const dim3 grid(wavev ? wave_unit_grid(K, nchunks) : (unsigned)nunits), block(kThreads);
  const int fb_chunk = 64, fb_nchunks = (int)ceil_div(C, fb_chunk);
  const int64_t fb_nunits = K * fb_nchunks;
  const dim3 fb_grid(wave_unit_grid(K, fb_nchunks));
#define TVMI_MS(PHT, PWT, SRT)                                                                                  \
  if (wavev) {                                                                                                  \
    bool dma = false;                                                                                           \
    if constexpr ((PHT) > 0) {                                                                                  \
      if (g_cfg_dma && declined) {                                                                              \
        if (g_cfg_dma_wpb == 1)                                                                                 \
          roi_align_fwd_ms_dma<T, PHT, PWT, SRT, 1><<<dim3(wave_unit_grid(K, nchunks, 1)), dim3(64), 0, stream>>>( \
              lv, r, out, (int)C, aligned, nchunks, chunk, nunits, declined, order);                            \
        else if (g_cfg_dma_wpb == 2)                                                                            \
          roi_align_fwd_ms_dma<T, PHT, PWT, SRT, 2><<<dim3(wave_unit_grid(K, nchunks, 2)), dim3(128), 0, stream>>>( \
              lv, r, out, (int)C, aligned, nchunks, chunk, nunits, declined, order);                            \
        else                                                                                                    \
          roi_align_fwd_ms_dma<T, PHT, PWT, SRT><<<grid, block, g_cfg_dma_extra_lds, stream>>>(lv, r, out, (int)C, aligned, nchunks, \
                                                                          chunk, nunits, declined, order);      \
        dma = true;                                                                                             \
      }                                                                                                         \
    }                                                                                                           \
    if (dma)                                                                                                    \
      roi_align_fwd_ms_wave<T, PHT, PWT, SRT><<<fb_grid, block, 0, stream>>>(                                   \
          lv, r, out, (int)C, (int)PH, (int)PW, (int)sr, aligned, fb_nchunks, fb_chunk, fb_nunits, declined, order); \
    else                                                                                                        \
      roi_align_fwd_ms_wave<T, PHT, PWT, SRT><<<grid, block, 0, stream>>>(                                      \
          lv, r, out, (int)C, (int)PH, (int)PW, (int)sr, aligned, nchunks, chunk, nunits, nullptr, order);      \
  } else                                                                                                        \
    roi_align_fwd_ms_tile<T, PHT, PWT, SRT><<<grid, block, 0, stream>>>(lv, r, out, (int)C, (int)PH, (int)PW,   \
                                                                        (int)sr, aligned, nchunks)
  if (PH == 7 && PW == 7 && sr == 2) {
    TVMI_MS(7, 7, 2);
  } else if (PH == 14 && PW == 14 && sr == 2) {
    TVMI_MS(14, 14, 2);
  } else {
    TVMI_MS(0, 0, 0);
  }
#undef TVMI_MS
  TVMI_RETURN_LAUNCH_STATUS("tvmi_multiscale_roi_align_forward");
}

}  // namespace
}  // namespace tvmi

extern "C" int tvmi_roi_align_forward(const void* input, const void* rois, void* output,
                                      tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W,
                                      int64_t K, int64_t pooled_h, int64_t pooled_w,
                                      double spatial_scale, int64_t sampling_ratio, int aligned,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "roi_align: pooled size must be positive");
  int* order = (workspace && workspace_bytes >= 2 * (size_t)K * sizeof(int) && K < (1 << 30)) ? static_cast<int*>(workspace) : nullptr;
  int* declined = order ? order + K : nullptr;
  TVMI_CHECK_ARG(N >= 0 && C >= 0 && H >= 0 && W >= 0 && K >= 0, "roi_align: negative size");
  if (K * C * pooled_h * pooled_w == 0) return 0;
  TVMI_CHECK_ARG(input && rois && output, "roi_align: null pointer");
  TVMI_CHECK_ARG(H * W < (1ll << 31) && K * tvmi::ceil_div(C, 32) < (1ll << 31),
                 "roi_align: size exceeds 32-bit launch limits");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "roi_align_forward",
                      return tvmi::launch_fwd<scalar_t>(input, rois, output, N, C, H, W, K, pooled_h,
                                                        pooled_w, spatial_scale, sampling_ratio,
                                                        aligned, order, declined, s));
  return 0;
}

extern "C" size_t tvmi_roi_align_backward_workspace_bytes(int64_t N, int64_t H, int64_t W, int64_t K) {
  if (N <= 0 || H <= 0 || W <= 0 || K <= 0) return 0;
  return (size_t)K * sizeof(int);  // one "left to the fallback launch" flag per RoI
}

extern "C" int tvmi_roi_align_backward(const void* grad, const void* rois, void* grad_input,
                                       tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W,
                                       int64_t K, int64_t pooled_h, int64_t pooled_w,
                                       double spatial_scale, int64_t sampling_ratio, int aligned,
                                       int64_t n_stride, int64_t c_stride, int64_t h_stride,
                                       int64_t w_stride, void* workspace, size_t workspace_bytes, void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "roi_align: pooled size must be positive");
  int* declined = (workspace && workspace_bytes >= (size_t)K * sizeof(int)) ? static_cast<int*>(workspace) : nullptr;
  if (K * C * pooled_h * pooled_w == 0 || N * H * W == 0) return 0;
  TVMI_CHECK_ARG(grad && rois && grad_input, "roi_align_backward: null pointer");
  TVMI_CHECK_ARG(H * W < (1ll << 31) && K * tvmi::ceil_div(C, 32) < (1ll << 31),
                 "roi_align_backward: size exceeds 32-bit launch limits");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "roi_align_backward",
                      return tvmi::launch_bwd<scalar_t>(grad, rois, grad_input, N, C, H, W, K,
                                                        pooled_h, pooled_w, spatial_scale,
                                                        sampling_ratio, aligned, n_stride, c_stride,
                                                        h_stride, w_stride, declined, s));
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
  tvmi::MsLevels lv;
  for (int i = 0; i < tvmi::kMaxLevels; ++i) {
    const int j = i < n_levels ? i : 0;
    TVMI_CHECK_ARG(inputs[j] != nullptr && heights[j] * widths[j] < (1ll << 31), "multiscale_roi_align: bad level");
    lv.ptr[i] = inputs[j];
    lv.H[i] = (int)heights[j];
    lv.W[i] = (int)widths[j];
    lv.scale[i] = (float)spatial_scales[j];
  }
  lv.n_levels = (int)n_levels;
  lv.k_min = (int)k_min;
  lv.k_max = (int)k_max;
  lv.s0 = (float)canonical_scale;
  lv.lvl0 = (float)canonical_level;
  lv.eps = (float)eps;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int* order = (workspace && workspace_bytes >= 2 * (size_t)K * sizeof(int) && K < (1 << 30)) ? static_cast<int*>(workspace) : nullptr;
  int* declined = order ? order + K : nullptr;
  switch (dt) {
    case TVMI_F32:
      return tvmi::launch_ms_fwd<float>(lv, rois, output, C, K, pooled_h, pooled_w, sampling_ratio, aligned, order, declined, s);
    case TVMI_F16:
      return tvmi::launch_ms_fwd<__half>(lv, rois, output, C, K, pooled_h, pooled_w, sampling_ratio, aligned, order, declined, s);
    default:
      return tvmi::launch_ms_fwd<__hip_bfloat16>(lv, rois, output, C, K, pooled_h, pooled_w, sampling_ratio, aligned,
                                                 order, declined, s);
  }
}

extern "C" int tvmi_multiscale_roi_align_backward(const void* grad, const void* rois, void* const* grad_inputs,
                                                  const int64_t* heights, const int64_t* widths,
                                                  const double* spatial_scales, int64_t n_levels, tvmi_dtype dt, int64_t N,
                                                  int64_t C, int64_t K, int64_t pooled_h, int64_t pooled_w,
                                                  int64_t sampling_ratio, int aligned, int64_t k_min, int64_t k_max,
                                                  double canonical_scale, double canonical_level, double eps,
                                                  int64_t n_stride, int64_t c_stride, int64_t h_stride, int64_t w_stride,
                                                  void* workspace, size_t workspace_bytes, void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "multiscale_roi_align_backward: pooled size must be positive");
  TVMI_CHECK_ARG(n_levels >= 1 && n_levels <= tvmi::kMaxLevels, "multiscale_roi_align_backward: 1..8 levels supported");
  if (K * C * pooled_h * pooled_w == 0 || N == 0) return 0;
  TVMI_CHECK_ARG(grad && rois && grad_inputs && heights && widths && spatial_scales, "multiscale_roi_align_backward: null pointer");
  TVMI_CHECK_ARG(dt == TVMI_F32 || dt == TVMI_F16 || dt == TVMI_BF16, "multiscale_roi_align_backward: float32 / float16 / bfloat16 only");
  TVMI_CHECK_ARG(K * tvmi::ceil_div(C, 32) < (1ll << 31), "multiscale_roi_align_backward: size exceeds 32-bit launch limits");
  tvmi::MsLevels lv;
  int64_t hmax = 1, wmax = 1;
  for (int i = 0; i < tvmi::kMaxLevels; ++i) {
    const int j = i < n_levels ? i : 0;
    TVMI_CHECK_ARG(grad_inputs[j] != nullptr && heights[j] > 0 && widths[j] > 0 && heights[j] * widths[j] * C < (1ll << 31),
                   "multiscale_roi_align_backward: bad level");
    lv.ptr[i] = grad_inputs[j];
    lv.H[i] = (int)heights[j];
    lv.W[i] = (int)widths[j];
    lv.scale[i] = (float)spatial_scales[j];
    hmax = std::max(hmax, heights[j]);
    wmax = std::max(wmax, widths[j]);
  }
  lv.n_levels = (int)n_levels;
  lv.k_min = (int)k_min;
  lv.k_max = (int)k_max;
  lv.s0 = (float)canonical_scale;
  lv.lvl0 = (float)canonical_level;
  lv.eps = (float)eps;
  int* declined = (workspace && workspace_bytes >= (size_t)K * sizeof(int)) ? static_cast<int*>(workspace) : nullptr;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // H / W / scale / grad_input of the single-level signature are placeholders: every RoI takes them from its level
  switch (dt) {
    case TVMI_F32:
      return tvmi::launch_bwd<float>(grad, rois, grad_inputs[0], N, C, hmax, wmax, K, pooled_h, pooled_w, spatial_scales[0],
                                     sampling_ratio, aligned, n_stride, c_stride, h_stride, w_stride, declined, s, &lv);
    case TVMI_F16:
      return tvmi::launch_bwd<__half>(grad, rois, grad_inputs[0], N, C, hmax, wmax, K, pooled_h, pooled_w, spatial_scales[0],
                                      sampling_ratio, aligned, n_stride, c_stride, h_stride, w_stride, declined, s, &lv);
    default:
      return tvmi::launch_bwd<__hip_bfloat16>(grad, rois, grad_inputs[0], N, C, hmax, wmax, K, pooled_h, pooled_w,
                                              spatial_scales[0], sampling_ratio, aligned, n_stride, c_stride, h_stride,
                                              w_stride, declined, s, &lv);
  }
}

extern "C" int tvmi_multiscale_roi_align_forward_nhwc(const void* const* inputs, const int64_t* heights,
                                                      const int64_t* widths, const double* spatial_scales,
                                                      int64_t n_levels, const void* rois, void* output, tvmi_dtype dt,
                                                      int64_t N, int64_t C, int64_t K, int64_t pooled_h, int64_t pooled_w,
                                                      int64_t sampling_ratio, int aligned, int64_t k_min, int64_t k_max,
                                                      double canonical_scale, double canonical_level, double eps,
                                                      void* stream) {
  TVMI_CHECK_ARG(n_levels >= 1 && n_levels <= tvmi::kMaxLevels, "roi_align (channels_last): 1..8 levels supported");
  if (K * C * pooled_h * pooled_w == 0) return 0;
  TVMI_CHECK_ARG(inputs && heights && widths && spatial_scales && rois && output, "roi_align (channels_last): null pointer");
  TVMI_CHECK_ARG(dt == TVMI_F32 || ((dt == TVMI_F16 || dt == TVMI_BF16) && C % 2 == 0),
                 "roi_align (channels_last): float32, or float16 / bfloat16 with an even channel count");
  TVMI_CHECK_ARG(pooled_h == 7 && pooled_w == 7 && sampling_ratio == 2,
                 "roi_align (channels_last): only 7x7 bins with sampling_ratio 2 have a native NHWC kernel");
  TVMI_CHECK_ARG(K * tvmi::ceil_div(C, 64) < (1ll << 31), "roi_align (channels_last): size exceeds 32-bit launch limits");
  tvmi::MsLevels lv;
  for (int i = 0; i < tvmi::kMaxLevels; ++i) {
    const int j = i < n_levels ? i : 0;
    TVMI_CHECK_ARG(inputs[j] != nullptr && heights[j] >= 2 && widths[j] >= 2 && heights[j] * widths[j] * C < (1ll << 31),
                   "roi_align (channels_last): every level needs H, W >= 2 and H*W*C < 2^31");
    lv.ptr[i] = inputs[j];
    lv.H[i] = (int)heights[j];
    lv.W[i] = (int)widths[j];
    lv.scale[i] = (float)spatial_scales[j];
  }
  lv.n_levels = (int)n_levels;
  lv.k_min = (int)k_min;
  lv.k_max = (int)k_max;
  lv.s0 = (float)canonical_scale;
  lv.lvl0 = (float)canonical_level;
  lv.eps = (float)eps;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dt == TVMI_F32) return tvmi::launch_ms_fwd_nhwc<float>(lv, rois, output, C, K, pooled_h, pooled_w, sampling_ratio, aligned, s);
  if (dt == TVMI_F16) return tvmi::launch_ms_fwd_nhwc<__half>(lv, rois, output, C, K, pooled_h, pooled_w, sampling_ratio, aligned, s);
  return tvmi::launch_ms_fwd_nhwc<__hip_bfloat16>(lv, rois, output, C, K, pooled_h, pooled_w, sampling_ratio, aligned, s);
}

// Tuning/debug knob, NOT part of the supported ABI (not declared in include/tvmi.h).
extern "C" int tvmi_debug_set(int key, int value) {
  if (key == 0) tvmi::g_cfg_variant = value;
  if (key == 1 && value > 0) tvmi::g_cfg_chunk = value;
  if (key == 3) tvmi::g_cfg_dma = value;
  if (key == 4) tvmi::g_cfg_order = value;
  if (key == 2) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(tvmi::g_roi_force_mode), &value, sizeof(int));
    return (int)e;
  }
  return 0;
}
