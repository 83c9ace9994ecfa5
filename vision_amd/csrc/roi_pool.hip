// roi_pool.hip — RoIPool, PSRoIAlign, PSRoIPool (forward + backward) for gfx950.
//
// Semantics (incl. the reference's quirks, kept on purpose):
//   roi_pool     torchvision/csrc/ops/cpu/roi_pool_kernel.cpp:24-134  (round()ed RoI, +1
//                inclusive size, bins [floor(p*bin), ceil((p+1)*bin)) clipped to [0,H],
//                max initialised with -FLT_MAX, empty bin -> 0 / argmax -1)
//   ps_roi_align cpu/ps_roi_align_kernel.cpp:17-151,219-313 (always -0.5 shift, no >=1
//                clamp, count = gh*gw with no max(.,1), c_in = (c_out*PH+ph)*PW+pw)
//   ps_roi_pool  cpu/ps_roi_pool_kernel.cpp:22-155 (size = max(end-start,1) without +1;
//                forward clips bins to H-1/W-1, backward clips to H/W; backward rounds the
//                RoI with roundf)
// Work decomposition.
//   roi_pool forward, 7x7, fp32 / fp16 / bf16 (`roi_pool_fwd_cols`): one wave64 per (RoI, 32-channel chunk), LANE =
//   WINDOW COLUMN.  The reference shape — one thread per output scanning its own bin — leaves a wave executing the
//   longest bin of its 64 lanes with dependent scalar-width loads; here the rows of the RoI window are walked once
//   per pooled row `ph` (a window row is contiguous: one coalesced load per row, 2-8 channels side by side when the
//   window is narrower than 32 / 16 / 8 columns, four rows in flight), every lane keeps the running (max, first
//   index) of ITS column for the 7 pooled rows in registers, and the columns of a bin are then folded across lanes
//   through a per-wave LDS transposition by the lane that owns the output (ties -> smaller index, which is the
//   reference's first-in-scan-order).  Outputs and argmax leave as contiguous 49-element runs.  Windows wider than
//   64 columns, other pooled shapes and fp64 take the lane-per-output kernel `roi_pool_fwd`.
//   Everything else: one lane per pooled output element with `pw` fastest, so a wave covers one or more complete
//   pooled rows of one (roi, channel) — output / argmax / channel_mapping stores are contiguous, and the lanes of a
//   wave read neighbouring input bins of the same plane (shared cache lines).  Backward kernels scatter with hardware
//   float atomics (`alertNotDeterministic` is raised by the dispatcher glue).
#include <float.h>

#include <algorithm>
#include <type_traits>

#include "tvmi_common.h"

namespace tvmi {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <typename T>
__global__ __launch_bounds__(kThreads) void roi_pool_fwd(const T* __restrict__ input,
                                                         const T* __restrict__ rois,
                                                         T* __restrict__ output, int* __restrict__ argmax,
                                                         int64_t total, int C, int H, int W, int PH,
                                                         int PW, double spatial_scale) {
  using A = typename Acc<T>::type;
  const A scale = (A)spatial_scale;
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * kThreads) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / ((int64_t)PW * PH)) % C);
    const int64_t n = idx / ((int64_t)PW * PH * C);
    const T* roi = rois + n * 5;
    const int b = (int)ld(roi);
    const int rsw = (int)round(ld(roi + 1) * scale);
    const int rsh = (int)round(ld(roi + 2) * scale);
    const int rew = (int)round(ld(roi + 3) * scale);
    const int reh = (int)round(ld(roi + 4) * scale);
    const int rw = max(rew - rsw + 1, 1), rh = max(reh - rsh + 1, 1);
    const A bin_h = (A)rh / (A)PH, bin_w = (A)rw / (A)PW;
    int hs = (int)floor((A)ph * bin_h), ws = (int)floor((A)pw * bin_w);
    int he = (int)ceil((A)(ph + 1) * bin_h), we = (int)ceil((A)(pw + 1) * bin_w);
    hs = clampi(hs + rsh, 0, H);
    he = clampi(he + rsh, 0, H);
    ws = clampi(ws + rsw, 0, W);
    we = clampi(we + rsw, 0, W);
    const bool empty = (he <= hs) || (we <= ws);
    A maxval = empty ? (A)0 : (A)-FLT_MAX;
    int maxidx = -1;
    const T* plane = input + ((int64_t)b * C + c) * H * W;
    for (int h = hs; h < he; ++h) {
      for (int w = ws; w < we; ++w) {
        const A v = ld(plane + h * W + w);
        if (v > maxval) {
          maxval = v;
          maxidx = h * W + w;
        }
      }
    }
    st(output + idx, maxval);
    argmax[idx] = maxidx;
  }
}

// ---- roi_pool forward, lane = window column (see the header) -------------------------------------------------
constexpr int kPoolChunk = 32;  // channels per wave

// one window row: every lane loads the pixel of its column (SGPR row pointer + a fixed 32-bit lane offset: no vector
// address arithmetic per row) and keeps (max, ROW of the first maximum) — the row is a scalar operand of the select
template <typename T>
__device__ __forceinline__ float ld_row(const T* __restrict__ rowp, unsigned vbyte) {
  // uniform 64-bit base + zero-extended 32-bit lane byte offset = the SGPR-base form of global_load
  return ld(reinterpret_cast<const T*>(reinterpret_cast<const char*>(rowp) + vbyte));
}
template <typename T>
__device__ __forceinline__ void pool_row(const T* __restrict__ rowp, unsigned vbyte, int h, float& av, int& ah) {
  const float v = ld_row(rowp, vbyte);
  const bool gt = v > av;  // strict, NaN never wins: cpu/roi_pool_kernel.cpp:83-88
  ah = gt ? h : ah;
  av = gt ? v : av;
}

template <typename T, int PH, int PW>
__global__ __launch_bounds__(kThreads) void roi_pool_fwd_cols(const T* __restrict__ input, const T* __restrict__ rois,
                                                              T* __restrict__ output, int* __restrict__ argmax, int K,
                                                              int C, int H, int W, double spatial_scale) {
  constexpr int kBins = PH * PW;
  constexpr int kWaves = kThreads / 64;
  __shared__ float2 s_col[kWaves][PH][64];  // per column: (max, bit pattern of its first index)
  __shared__ int s_w[kWaves][8];            // per pw: w0 | w1 << 16 (window-relative), bit 31: bin row/col range empty
  __shared__ int s_h[kWaves][8];            // per ph: 1 if the row range is empty
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // unit -> (RoI k, channel chunk): the feature planes of one channel chunk of one image are a few MB — they fit the
  // 4 MB L2 of an XCD, the whole map does not.  Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8 —
  // observed, a speed assumption only), so partition p = blockIdx % 8 owns the chunks c = p (mod 8) and walks the RoIs
  // (sorted by image in every caller we know) in order, 4 consecutive RoIs per workgroup: an XCD's L2 then serves
  // every re-read of a window instead of the fabric (measured: L2 hit 20 % -> see DESIGN.md).  With fewer than 8
  // chunks the RoI list is split into 8 / chunks contiguous slices instead.
  const int chunks = (C + kPoolChunk - 1) / kPoolChunk;
  int k, chunk;
  {
    const int part = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (chunks >= 8) {
      const int mine = (chunks - part + 7) >> 3;           // chunks owned by this partition: part, part + 8, ...
      const int64_t e = (int64_t)slot * kWaves + wave;     // entry in the partition's list, RoI fastest
      if (e >= (int64_t)mine * K) return;
      chunk = part + 8 * (int)(e / K);
      k = (int)(e % K);
    } else if ((8 % chunks) == 0) {
      const int slices = 8 / chunks, slice = part / chunks;
      const int per = (K + slices - 1) / slices;
      const int64_t e = (int64_t)slot * kWaves + wave;
      if (e >= per) return;
      chunk = part % chunks;
      k = slice * per + (int)e;
      if (k >= K) return;
    } else {
      const int64_t unit = (int64_t)blockIdx.x * kWaves + wave;
      if (unit >= (int64_t)K * chunks) return;
      k = (int)(unit / chunks);
      chunk = (int)(unit % chunks);
    }
  }
  const int c0 = chunk * kPoolChunk;
  // RoI geometry exactly as roi_pool_fwd above (cpu/roi_pool_kernel.cpp:39-73), wave-uniform
  const float scale = (float)spatial_scale;
  const T* roi = rois + (int64_t)k * 5;
  const int b = __builtin_amdgcn_readfirstlane((int)ld(roi));
  const int rsw = __builtin_amdgcn_readfirstlane((int)round(ld(roi + 1) * scale));
  const int rsh = __builtin_amdgcn_readfirstlane((int)round(ld(roi + 2) * scale));
  const int rew = __builtin_amdgcn_readfirstlane((int)round(ld(roi + 3) * scale));
  const int reh = __builtin_amdgcn_readfirstlane((int)round(ld(roi + 4) * scale));
  const int rw = max(rew - rsw + 1, 1), rh = max(reh - rsh + 1, 1);
  const float bin_h = (float)rh / (float)PH, bin_w = (float)rw / (float)PW;
  int hs[PH], he[PH];
#pragma unroll
  for (int ph = 0; ph < PH; ++ph) {
    hs[ph] = __builtin_amdgcn_readfirstlane(clampi((int)floor((float)ph * bin_h) + rsh, 0, H));
    he[ph] = __builtin_amdgcn_readfirstlane(clampi((int)ceil((float)(ph + 1) * bin_h) + rsh, 0, H));
  }
  const int X0 = __builtin_amdgcn_readfirstlane(clampi((int)floor(0.f * bin_w) + rsw, 0, W));
  const int X1 = __builtin_amdgcn_readfirstlane(clampi((int)ceil((float)PW * bin_w) + rsw, 0, W));
  const int ww = X1 - X0;  // columns any bin can touch (bins are monotone in pw)
  const int64_t out0 = ((int64_t)k * C + c0) * kBins;
  const T* img = input + (int64_t)b * C * H * W;  // wave-uniform base; offsets below fit 32 bits (checked by the launcher)

  if (ww > 64) {
    // wide window: lane = output bin, the reference's scan (rare: RoIs above 64 feature columns)
    for (int cc = 0; cc < kPoolChunk && c0 + cc < C; ++cc) {
      if (lane < kBins) {
        const int ph = lane / PW, pw = lane % PW;
        const int h0 = clampi((int)floor((float)ph * bin_h) + rsh, 0, H), h1 = clampi((int)ceil((float)(ph + 1) * bin_h) + rsh, 0, H);
        const int w0 = clampi((int)floor((float)pw * bin_w) + rsw, 0, W), w1 = clampi((int)ceil((float)(pw + 1) * bin_w) + rsw, 0, W);
        const bool empty = (h1 <= h0) || (w1 <= w0);
        float best = empty ? 0.f : -FLT_MAX;
        int besti = -1;
        const T* plane = img + (int64_t)(c0 + cc) * H * W;
        for (int h = h0; h < h1; ++h)
          for (int w = w0; w < w1; ++w) {
            const float v = ld(plane + h * W + w);
            if (v > best) {
              best = v;
              besti = h * W + w;
            }
          }
        st(output + out0 + (int64_t)cc * kBins + lane, best);
        argmax[out0 + (int64_t)cc * kBins + lane] = besti;
      }
    }
    return;
  }

  // bin tables of this RoI (lanes 0..PW-1 / 0..PH-1 compute one entry each)
  if (lane < 8) {
    const int w0 = clampi((int)floor((float)lane * bin_w) + rsw, 0, W), w1 = clampi((int)ceil((float)(lane + 1) * bin_w) + rsw, 0, W);
    const int h0 = clampi((int)floor((float)lane * bin_h) + rsh, 0, H), h1 = clampi((int)ceil((float)(lane + 1) * bin_h) + rsh, 0, H);
    s_w[wave][lane] = w1 <= w0 ? (int)0x80000000 : ((w0 - X0) | ((w1 - X0) << 16));
    s_h[wave][lane] = h1 <= h0;
  }
  // G channels side by side: lane = (channel slot g, column)
  const int wp = ww <= 8 ? 8 : (ww <= 16 ? 16 : (ww <= 32 ? 32 : 64));
  const int G = 64 / wp;
  const int g = lane / wp, col = lane % wp;
  // lanes beyond the window / beyond C read a valid pixel (column X0 / channel c0) whose result is never consumed
  const int x = X0 + (col < ww ? col : 0);
  const int xs = min(x, W - 1);
  for (int cc = 0; cc < kPoolChunk; cc += G) {
    const int c = c0 + cc + g;
    const unsigned voff = (unsigned)((c < C ? c : c0) * H * W + xs) * (unsigned)sizeof(T);  // bytes
    float accv[PH];
    int acch[PH];
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
      float av = -FLT_MAX;
      int ah = -1;
      int h = hs[ph];
      const int e = he[ph];
      const T* rowp = img + (int64_t)h * W;
      for (; h + 4 <= e; h += 4) {  // four independent loads in flight
        const float v0 = ld_row(rowp, voff), v1 = ld_row(rowp + W, voff), v2 = ld_row(rowp + 2 * W, voff), v3 = ld_row(rowp + 3 * W, voff);
        bool gt = v0 > av;
        ah = gt ? h : ah;
        av = gt ? v0 : av;
        gt = v1 > av;
        ah = gt ? h + 1 : ah;
        av = gt ? v1 : av;
        gt = v2 > av;
        ah = gt ? h + 2 : ah;
        av = gt ? v2 : av;
        gt = v3 > av;
        ah = gt ? h + 3 : ah;
        av = gt ? v3 : av;
        rowp += 4 * (int64_t)W;
      }
      for (; h < e; ++h) {
        pool_row(rowp, voff, h, av, ah);
        rowp += W;
      }
      accv[ph] = av;
      acch[ph] = ah;
    }
#pragma unroll
    for (int ph = 0; ph < PH; ++ph)
      s_col[wave][ph][lane] = make_float2(accv[ph], __int_as_float(acch[ph] < 0 ? -1 : acch[ph] * W + x));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // fold the columns of each bin: lane = (channel slot, bin)
    const int nout = kBins * G;
    for (int o = lane; o < nout; o += 64) {
      const int g2 = o / kBins, bin = o % kBins;
      const int ph = bin / PW, pw = bin % PW;
      const int wr = s_w[wave][pw];
      const bool empty = wr < 0 || s_h[wave][ph] != 0;
      float best = empty ? 0.f : -FLT_MAX;
      int besti = -1;
      if (!empty) {
        const float2* sc = &s_col[wave][ph][g2 * wp];
        const int q1 = wr >> 16;
        for (int q = wr & 0xffff; q < q1; ++q) {
          const float2 e = sc[q];
          const int i = __float_as_int(e.y);
          // larger value wins; equal values: the smaller index = the earlier position of the reference's (h, w) scan
          if (e.x > best || (e.x == best && (unsigned)i < (unsigned)besti)) {
            best = e.x;
            besti = i;
          }
        }
      }
      if (c0 + cc + g2 < C) {
        // write-once streams (2 x 200 MB at the measured shape) must not evict the feature planes from L2
        T outv;
        st(&outv, best);
        if constexpr (sizeof(T) == 2) {
          unsigned short bits;
          __builtin_memcpy(&bits, &outv, 2);
          __builtin_nontemporal_store(bits, reinterpret_cast<unsigned short*>(output + out0 + (int64_t)cc * kBins + o));
        } else {
          __builtin_nontemporal_store(outv, output + out0 + (int64_t)cc * kBins + o);
        }
        __builtin_nontemporal_store(besti, argmax + out0 + (int64_t)cc * kBins + o);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void roi_pool_bwd(const T* __restrict__ grad,
                                                         const T* __restrict__ rois,
                                                         const int* __restrict__ argmax,
                                                         T* __restrict__ grad_input, int64_t total, int C,
                                                         int H, int W, int PH, int PW, int64_t ns,
                                                         int64_t cs, int64_t hs, int64_t ws) {
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * kThreads) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / ((int64_t)PW * PH)) % C);
    const int64_t n = idx / ((int64_t)PW * PH * C);
    const int am = argmax[idx];
    if (am == -1) continue;
    const int b = (int)ld(rois + n * 5);
    atomic_accum(grad_input + ((int64_t)b * C + c) * H * W + am, ld(grad + n * ns + c * cs + ph * hs + pw * ws));
  }
}

// ---- bilinear helpers of the PS-RoIAlign CPU kernel (cpu/ps_roi_align_kernel.cpp:17-70,153-217)
template <typename A>
__device__ __forceinline__ bool bilinear_setup(int H, int W, A y, A x, int& yl, int& yh, int& xl, int& xh,
                                               A& w1, A& w2, A& w3, A& w4) {
  if (y < (A)-1.0 || y > (A)H || x < (A)-1.0 || x > (A)W) return false;
  if (y <= (A)0) y = (A)0;
  if (x <= (A)0) x = (A)0;
  yl = (int)y;
  xl = (int)x;
  if (yl >= H - 1) {
    yh = yl = H - 1;
    y = (A)yl;
  } else {
    yh = yl + 1;
  }
  if (xl >= W - 1) {
    xh = xl = W - 1;
    x = (A)xl;
  } else {
    xh = xl + 1;
  }
  const A ly = y - (A)yl, lx = x - (A)xl;
  const A hy = (A)1. - ly, hx = (A)1. - lx;
  w1 = hy * hx;
  w2 = hy * lx;
  w3 = ly * hx;
  w4 = ly * lx;
  return true;
}

template <typename T, bool kBackward>
__global__ __launch_bounds__(kThreads) void ps_roi_align_kernel(
    const T* __restrict__ data /* input (fwd) | grad_output (bwd) */, const T* __restrict__ rois,
    T* __restrict__ out /* output (fwd) | grad_input (bwd) */, int* __restrict__ channel_mapping,
    int64_t total, int C, int H, int W, int PH, int PW, int C_out, double spatial_scale, int sr) {
  using A = typename Acc<T>::type;
  const A scale = (A)spatial_scale;
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * kThreads) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c_out = (int)((idx / ((int64_t)PW * PH)) % C_out);
    const int64_t n = idx / ((int64_t)PW * PH * C_out);
    const T* roi = rois + n * 5;
    const int b = (int)ld(roi);
    const A rsw = ld(roi + 1) * scale - (A)0.5;
    const A rsh = ld(roi + 2) * scale - (A)0.5;
    const A rew = ld(roi + 3) * scale - (A)0.5;
    const A reh = ld(roi + 4) * scale - (A)0.5;
    const A rw = rew - rsw, rh = reh - rsh;
    const A bin_h = rh / (A)PH, bin_w = rw / (A)PW;
    const int c_in = kBackward ? channel_mapping[idx] : (c_out * PH + ph) * PW + pw;
    const A hstart = (A)ph * bin_h + rsh;
    const A wstart = (A)pw * bin_w + rsw;
    const int gh = sr > 0 ? sr : (int)ceil(rh / (A)PH);
    const int gw = sr > 0 ? sr : (int)ceil(rw / (A)PW);
    const A count = (A)(gh * gw);
    const int64_t plane_off = ((int64_t)b * C + c_in) * H * W;
    A acc = (A)0;
    A go = (A)0;
    if (kBackward) go = ld(data + idx);
    for (int iy = 0; iy < gh; ++iy) {
      const A y = hstart + (A)((float)iy + .5f) * bin_h / (A)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const A x = wstart + (A)((float)ix + .5f) * bin_w / (A)gw;
        int yl, yh, xl, xh;
        A w1, w2, w3, w4;
        if (!bilinear_setup<A>(H, W, y, x, yl, yh, xl, xh, w1, w2, w3, w4)) continue;
        if (kBackward) {
          T* gi = out + plane_off;
          atomic_accum(gi + yl * W + xl, go * w1 / count);
          atomic_accum(gi + yl * W + xh, go * w2 / count);
          atomic_accum(gi + yh * W + xl, go * w3 / count);
          atomic_accum(gi + yh * W + xh, go * w4 / count);
        } else {
          const T* p = data + plane_off;
          const A v1 = ld(p + yl * W + xl), v2 = ld(p + yl * W + xh);
          const A v3 = ld(p + yh * W + xl), v4 = ld(p + yh * W + xh);
          acc += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
        }
      }
    }
    if (!kBackward) {
      st(out + idx, acc / count);
      channel_mapping[idx] = c_in;
    }
  }
}

template <typename T, bool kBackward>
__global__ __launch_bounds__(kThreads) void ps_roi_pool_kernel(
    const T* __restrict__ data, const T* __restrict__ rois, T* __restrict__ out,
    int* __restrict__ channel_mapping, int64_t total, int C, int H, int W, int PH, int PW, int C_out,
    double spatial_scale) {
  using A = typename Acc<T>::type;
  const A scale = (A)spatial_scale;
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * kThreads) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c_out = (int)((idx / ((int64_t)PW * PH)) % C_out);
    const int64_t n = idx / ((int64_t)PW * PH * C_out);
    const T* roi = rois + n * 5;
    const int b = (int)ld(roi);
    int rsw, rsh, rew, reh;
    if (kBackward) {  // cpu/ps_roi_pool_kernel.cpp:109-112 uses roundf
      rsw = (int)roundf((float)(ld(roi + 1) * scale));
      rsh = (int)roundf((float)(ld(roi + 2) * scale));
      rew = (int)roundf((float)(ld(roi + 3) * scale));
      reh = (int)roundf((float)(ld(roi + 4) * scale));
    } else {
      rsw = (int)round(ld(roi + 1) * scale);
      rsh = (int)round(ld(roi + 2) * scale);
      rew = (int)round(ld(roi + 3) * scale);
      reh = (int)round(ld(roi + 4) * scale);
    }
    const int rw = max(rew - rsw, 1), rh = max(reh - rsh, 1);
    const A bin_h = (A)rh / (A)PH, bin_w = (A)rw / (A)PW;
    int hs = (int)floor((A)ph * bin_h), ws = (int)floor((A)pw * bin_w);
    int he = (int)ceil((A)(ph + 1) * bin_h), we = (int)ceil((A)(pw + 1) * bin_w);
    const int hmax = kBackward ? H : H - 1, wmax = kBackward ? W : W - 1;
    hs = clampi(hs + rsh, 0, hmax);
    he = clampi(he + rsh, 0, hmax);
    ws = clampi(ws + rsw, 0, wmax);
    we = clampi(we + rsw, 0, wmax);
    const bool empty = (he <= hs) || (we <= ws);
    const int c_in = kBackward ? channel_mapping[idx] : (c_out * PH + ph) * PW + pw;
    const int64_t plane_off = ((int64_t)b * C + c_in) * H * W;
    const A bin_area = (A)((he - hs) * (we - ws));
    if (kBackward) {
      const A diff = empty ? (A)0 : ld(data + idx) / bin_area;
      T* gi = out + plane_off;
      for (int h = hs; h < he; ++h)
        for (int w = ws; w < we; ++w) atomic_accum(gi + h * W + w, diff);
    } else {
      const T* p = data + plane_off;
      A sum = (A)0;
      for (int h = hs; h < he; ++h)
        for (int w = ws; w < we; ++w) sum += ld(p + h * W + w);
      st(out + idx, empty ? (A)0 : sum / bin_area);
      channel_mapping[idx] = c_in;
    }
  }
}

inline dim3 grid_for(int64_t total) {
  return dim3((unsigned)std::min<int64_t>(ceil_div(total, kThreads), 1 << 20));
}

}  // namespace
}  // namespace tvmi

using tvmi::grid_for;
using tvmi::kThreads;

extern "C" int tvmi_roi_pool_forward(const void* input, const void* rois, void* output, int32_t* argmax,
                                     tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                                     int64_t pooled_h, int64_t pooled_w, double spatial_scale,
                                     void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "roi_pool: pooled size must be positive");
  const int64_t total = K * C * pooled_h * pooled_w;
  if (total == 0) return 0;
  TVMI_CHECK_ARG(input && rois && output && argmax, "roi_pool: null pointer");
  TVMI_CHECK_ARG(H * W < (1ll << 31), "roi_pool: plane too large");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (pooled_h == 7 && pooled_w == 7 && dt != TVMI_F64 && C * H * W < (1ll << 29) && W >= 1 && H >= 1 &&
      K * ((C + tvmi::kPoolChunk - 1) / tvmi::kPoolChunk) < (1ll << 31)) {
    // grid: 8 partitions (see the kernel) x the workgroups of the longest partition
    const int64_t chunks = (C + tvmi::kPoolChunk - 1) / tvmi::kPoolChunk, wpb = kThreads / 64;
    int64_t blocks;
    if (chunks >= 8) blocks = 8 * ((((chunks + 7) / 8) * K + wpb - 1) / wpb);
    else if (8 % chunks == 0) blocks = 8 * (((K + (8 / chunks) - 1) / (8 / chunks) + wpb - 1) / wpb);
    else blocks = (K * chunks + wpb - 1) / wpb;
    const dim3 grid((unsigned)blocks);
#define TVMI_POOL_COLS(scalar_t)                                                                                      \
  tvmi::roi_pool_fwd_cols<scalar_t, 7, 7><<<grid, dim3(kThreads), 0, s>>>((const scalar_t*)input, (const scalar_t*)rois, \
                                                                          (scalar_t*)output, argmax, (int)K, (int)C,    \
                                                                          (int)H, (int)W, spatial_scale)
    if (dt == TVMI_F32) TVMI_POOL_COLS(float);
    else if (dt == TVMI_F16) TVMI_POOL_COLS(__half);
    else if (dt == TVMI_BF16) TVMI_POOL_COLS(__hip_bfloat16);
    else return ::tvmi::set_error(hipErrorInvalidValue, "roi_pool_forward: unsupported dtype");
#undef TVMI_POOL_COLS
    TVMI_RETURN_LAUNCH_STATUS("tvmi_roi_pool_forward");
  }
  TVMI_DISPATCH_FLOAT(dt, "roi_pool_forward",
                      tvmi::roi_pool_fwd<scalar_t><<<grid_for(total), dim3(kThreads), 0, s>>>(
                          (const scalar_t*)input, (const scalar_t*)rois, (scalar_t*)output, argmax, total,
                          (int)C, (int)H, (int)W, (int)pooled_h, (int)pooled_w, spatial_scale));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_roi_pool_forward");
}

extern "C" int tvmi_roi_pool_backward(const void* grad, const void* rois, const int32_t* argmax,
                                      void* grad_input, tvmi_dtype dt, int64_t N, int64_t C, int64_t H,
                                      int64_t W, int64_t K, int64_t pooled_h, int64_t pooled_w,
                                      int64_t n_stride, int64_t c_stride, int64_t h_stride,
                                      int64_t w_stride, void* stream) {
  const int64_t total = K * C * pooled_h * pooled_w;
  if (total == 0 || N * C * H * W == 0) return 0;
  TVMI_CHECK_ARG(grad && rois && argmax && grad_input, "roi_pool_backward: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "roi_pool_backward",
                      tvmi::roi_pool_bwd<scalar_t><<<grid_for(total), dim3(kThreads), 0, s>>>(
                          (const scalar_t*)grad, (const scalar_t*)rois, argmax, (scalar_t*)grad_input, total,
                          (int)C, (int)H, (int)W, (int)pooled_h, (int)pooled_w, n_stride, c_stride, h_stride,
                          w_stride));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_roi_pool_backward");
}

extern "C" int tvmi_ps_roi_align_forward(const void* input, const void* rois, void* output,
                                         int32_t* channel_mapping, tvmi_dtype dt, int64_t N, int64_t C,
                                         int64_t H, int64_t W, int64_t K, int64_t pooled_h,
                                         int64_t pooled_w, double spatial_scale, int64_t sampling_ratio,
                                         void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "ps_roi_align: pooled size must be positive");
  TVMI_CHECK_ARG(C % (pooled_h * pooled_w) == 0,
                 "input channels must be a multiple of pooling height * pooling width");
  const int64_t C_out = C / (pooled_h * pooled_w);
  const int64_t total = K * C_out * pooled_h * pooled_w;
  if (total == 0) return 0;
  TVMI_CHECK_ARG(input && rois && output && channel_mapping, "ps_roi_align: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "ps_roi_align_forward",
                      (tvmi::ps_roi_align_kernel<scalar_t, false><<<grid_for(total), dim3(kThreads), 0, s>>>(
                          (const scalar_t*)input, (const scalar_t*)rois, (scalar_t*)output, channel_mapping,
                          total, (int)C, (int)H, (int)W, (int)pooled_h, (int)pooled_w, (int)C_out,
                          spatial_scale, (int)sampling_ratio)));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_ps_roi_align_forward");
}

extern "C" int tvmi_ps_roi_align_backward(const void* grad, const void* rois,
                                          const int32_t* channel_mapping, void* grad_input, tvmi_dtype dt,
                                          int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                                          int64_t pooled_h, int64_t pooled_w, double spatial_scale,
                                          int64_t sampling_ratio, void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "ps_roi_align_backward: pooled size must be positive");
  const int64_t C_out = C / (pooled_h * pooled_w);
  const int64_t total = K * C_out * pooled_h * pooled_w;
  if (total == 0 || N * C * H * W == 0) return 0;
  TVMI_CHECK_ARG(grad && rois && channel_mapping && grad_input, "ps_roi_align_backward: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "ps_roi_align_backward",
                      (tvmi::ps_roi_align_kernel<scalar_t, true><<<grid_for(total), dim3(kThreads), 0, s>>>(
                          (const scalar_t*)grad, (const scalar_t*)rois, (scalar_t*)grad_input,
                          const_cast<int32_t*>(channel_mapping), total, (int)C, (int)H, (int)W, (int)pooled_h,
                          (int)pooled_w, (int)C_out, spatial_scale, (int)sampling_ratio)));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_ps_roi_align_backward");
}

extern "C" int tvmi_ps_roi_pool_forward(const void* input, const void* rois, void* output,
                                        int32_t* channel_mapping, tvmi_dtype dt, int64_t N, int64_t C,
                                        int64_t H, int64_t W, int64_t K, int64_t pooled_h, int64_t pooled_w,
                                        double spatial_scale, void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "ps_roi_pool: pooled size must be positive");
  TVMI_CHECK_ARG(C % (pooled_h * pooled_w) == 0,
                 "input channels must be a multiple of pooling height * pooling width");
  const int64_t C_out = C / (pooled_h * pooled_w);
  const int64_t total = K * C_out * pooled_h * pooled_w;
  if (total == 0) return 0;
  TVMI_CHECK_ARG(input && rois && output && channel_mapping, "ps_roi_pool: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "ps_roi_pool_forward",
                      (tvmi::ps_roi_pool_kernel<scalar_t, false><<<grid_for(total), dim3(kThreads), 0, s>>>(
                          (const scalar_t*)input, (const scalar_t*)rois, (scalar_t*)output, channel_mapping,
                          total, (int)C, (int)H, (int)W, (int)pooled_h, (int)pooled_w, (int)C_out,
                          spatial_scale)));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_ps_roi_pool_forward");
}

extern "C" int tvmi_ps_roi_pool_backward(const void* grad, const void* rois, const int32_t* channel_mapping,
                                         void* grad_input, tvmi_dtype dt, int64_t N, int64_t C, int64_t H,
                                         int64_t W, int64_t K, int64_t pooled_h, int64_t pooled_w,
                                         double spatial_scale, void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "ps_roi_pool_backward: pooled size must be positive");
  const int64_t C_out = C / (pooled_h * pooled_w);
  const int64_t total = K * C_out * pooled_h * pooled_w;
  if (total == 0 || N * C * H * W == 0) return 0;
  TVMI_CHECK_ARG(grad && rois && channel_mapping && grad_input, "ps_roi_pool_backward: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_DISPATCH_FLOAT(dt, "ps_roi_pool_backward",
                      (tvmi::ps_roi_pool_kernel<scalar_t, true><<<grid_for(total), dim3(kThreads), 0, s>>>(
                          (const scalar_t*)grad, (const scalar_t*)rois, (scalar_t*)grad_input,
                          const_cast<int32_t*>(channel_mapping), total, (int)C, (int)H, (int)W, (int)pooled_h,
                          (int)pooled_w, (int)C_out, spatial_scale)));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_ps_roi_pool_backward");
}
