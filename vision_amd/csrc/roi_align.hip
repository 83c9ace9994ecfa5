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
#include "tvmi_common.h"

namespace tvmi {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxTab = 128;      // max PH*gh (and PW*gw) samples per axis kept in LDS
constexpr int kWinFloats = 8192;  // LDS window capacity in floats (32 KiB)
constexpr int kChunk = 32;        // channels per workgroup

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

template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_tile(
    const T* __restrict__ input, const T* __restrict__ rois, T* __restrict__ output, int C, int H,
    int W, int PH_, int PW_, float spatial_scale, int sr_, int aligned, int nchunks) {
  __shared__ TileShared s;
  const int PH = PHT > 0 ? PHT : PH_;
  const int PW = PWT > 0 ? PWT : PW_;
  const int sr = SRT > 0 ? SRT : sr_;
  const int PHW = PH * PW;
  const int tid = threadIdx.x;
  const int k = blockIdx.x / nchunks;
  const int c0 = (blockIdx.x - k * nchunks) * kChunk;
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
    int64_t cs, int64_t hs, int64_t ws) {
  __shared__ TileShared s;
  const int PH = PHT > 0 ? PHT : PH_;
  const int PW = PWT > 0 ? PWT : PW_;
  const int sr = SRT > 0 ? SRT : sr_;
  const int PHW = PH * PW;
  const int tid = threadIdx.x;
  const int k = blockIdx.x / nchunks;
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

template <typename T>
int launch_fwd(const void* input, const void* rois, void* output, int64_t N, int64_t C, int64_t H,
               int64_t W, int64_t K, int64_t PH, int64_t PW, double scale, int64_t sr, int aligned,
               hipStream_t stream) {
  const T* in = static_cast<const T*>(input);
  const T* r = static_cast<const T*>(rois);
  T* out = static_cast<T*>(output);
  const int64_t total = K * C * PH * PW;
  if constexpr (std::is_same<T, double>::value) {
    const int64_t blocks = std::min<int64_t>(ceil_div(total, kThreads), 1 << 20);
    roi_align_fwd_generic<T><<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(
        in, r, out, total, (int)C, (int)H, (int)W, (int)PH, (int)PW, scale, (int)sr, aligned);
  } else {
    const int nchunks = (int)ceil_div(C, kChunk);
    const dim3 grid((unsigned)(K * nchunks)), block(kThreads);
    const float fs = (float)scale;
#define TVMI_FWD(PHT, PWT, SRT)                                                                  \
  roi_align_fwd_tile<T, PHT, PWT, SRT><<<grid, block, 0, stream>>>(in, r, out, (int)C, (int)H,   \
                                                                   (int)W, (int)PH, (int)PW, fs, \
                                                                   (int)sr, aligned, nchunks)
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
               int64_t ns, int64_t cs, int64_t hs, int64_t ws, hipStream_t stream) {
  const T* g = static_cast<const T*>(grad);
  const T* r = static_cast<const T*>(rois);
  T* gi = static_cast<T*>(grad_input);
  const int64_t total = K * C * PH * PW;
  if constexpr (std::is_same<T, double>::value) {
    const int64_t blocks = std::min<int64_t>(ceil_div(total, kThreads), 1 << 20);
    roi_align_bwd_generic<T><<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(
        g, r, gi, total, (int)C, (int)H, (int)W, (int)PH, (int)PW, scale, (int)sr, aligned, ns, cs,
        hs, ws);
  } else {
    const int nchunks = (int)ceil_div(C, kChunk);
    const dim3 grid((unsigned)(K * nchunks)), block(kThreads);
    const float fs = (float)scale;
#define TVMI_BWD(PHT, PWT, SRT)                                                                   \
  roi_align_bwd_tile<T, PHT, PWT, SRT><<<grid, block, 0, stream>>>(g, r, gi, (int)C, (int)H,      \
                                                                   (int)W, (int)PH, (int)PW, fs,  \
                                                                   (int)sr, aligned, nchunks, ns, \
                                                                   cs, hs, ws)
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

}  // namespace
}  // namespace tvmi

extern "C" int tvmi_roi_align_forward(const void* input, const void* rois, void* output,
                                      tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W,
                                      int64_t K, int64_t pooled_h, int64_t pooled_w,
                                      double spatial_scale, int64_t sampling_ratio, int aligned,
                                      void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "roi_align: pooled size must be positive");
  TVMI_CHECK_ARG(N >= 0 && C >= 0 && H >= 0 && W >= 0 && K >= 0, "roi_align: negative size");
  if (K * C * pooled_h * pooled_w == 0) return 0;
  TVMI_CHECK_ARG(input && rois && output, "roi_align: null pointer");
  TVMI_CHECK_ARG(H * W < (1ll << 31) && K * tvmi::ceil_div(C, 32) < (1ll << 31),
                 "roi_align: size exceeds 32-bit launch limits");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "roi_align_forward",
                      return tvmi::launch_fwd<scalar_t>(input, rois, output, N, C, H, W, K, pooled_h,
                                                        pooled_w, spatial_scale, sampling_ratio,
                                                        aligned, s));
  return 0;
}

extern "C" int tvmi_roi_align_backward(const void* grad, const void* rois, void* grad_input,
                                       tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W,
                                       int64_t K, int64_t pooled_h, int64_t pooled_w,
                                       double spatial_scale, int64_t sampling_ratio, int aligned,
                                       int64_t n_stride, int64_t c_stride, int64_t h_stride,
                                       int64_t w_stride, void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "roi_align: pooled size must be positive");
  if (K * C * pooled_h * pooled_w == 0 || N * H * W == 0) return 0;
  TVMI_CHECK_ARG(grad && rois && grad_input, "roi_align_backward: null pointer");
  TVMI_CHECK_ARG(H * W < (1ll << 31) && K * tvmi::ceil_div(C, 32) < (1ll << 31),
                 "roi_align_backward: size exceeds 32-bit launch limits");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "roi_align_backward",
                      return tvmi::launch_bwd<scalar_t>(grad, rois, grad_input, N, C, H, W, K,
                                                        pooled_h, pooled_w, spatial_scale,
                                                        sampling_ratio, aligned, n_stride, c_stride,
                                                        h_stride, w_stride, s));
  return 0;
}
