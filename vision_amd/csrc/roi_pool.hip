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
//   FOUR WINDOW COLUMNS of one of 4 / 8 channels.  The reference shape — one thread per output scanning its own bin —
//   leaves a wave executing the longest bin of its 64 lanes with dependent 4-byte loads (measured: VALU-bound on
//   574 M instructions, 1.29 ms).  Here the rows of the RoI window are walked once per pooled row `ph` (a window row is
//   contiguous: one 16-byte load per lane and row, four rows in flight), every lane keeps the running (max, first
//   row) of its four columns in registers, the columns of a pooled row are folded into its 7 bins across lanes through
//   a 2 KB per-wave LDS transposition (double-buffered, so the fold of row ph overlaps the walk of ph + 1; ties ->
//   smaller index, which is the reference's first-in-scan-order), and a lane keeps its 7 results until the pass ends.
//   Channel chunks are pinned to XCDs (a chunk's planes fit one 4 MB L2), outputs leave as non-temporal stores.
//   Windows wider than 64 columns, other pooled shapes and fp64 take the lane-per-output kernel `roi_pool_fwd`.
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

// ---- roi_pool forward, lane = four window columns (see the header) -------------------------------------------
constexpr int kPoolChunk = 32;     // channels per wave
constexpr int kPoolThreads = 256;

// four consecutive pixels of a row with ONE load per lane (16 bytes fp32 / 8 bytes 16-bit; element alignment is enough
// on this memory path): the texture path retires a wave load in ~16 cycles whatever its width, so bytes per load are
// what the row walk is bound by (measured with dword loads: data-return path 94 % busy)
template <typename T>
struct Quad;
template <>
struct Quad<float> {
  typedef float raw __attribute__((ext_vector_type(4))) __attribute__((aligned(4)));
  static __device__ __forceinline__ float cvt(float v) { return v; }
};
template <>
struct Quad<__half> {
  typedef unsigned short raw __attribute__((ext_vector_type(4))) __attribute__((aligned(2)));
  static __device__ __forceinline__ float cvt(unsigned short v) { return __half2float(__ushort_as_half(v)); }
};
template <>
struct Quad<__hip_bfloat16> {
  typedef unsigned short raw __attribute__((ext_vector_type(4))) __attribute__((aligned(2)));
  static __device__ __forceinline__ float cvt(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
};
template <typename T>
__device__ __forceinline__ typename Quad<T>::raw ld_quad(const T* __restrict__ rowp, unsigned vbyte) {
  // uniform 64-bit row pointer + zero-extended 32-bit lane byte offset
  return *reinterpret_cast<const typename Quad<T>::raw*>(reinterpret_cast<const char*>(rowp) + vbyte);
}
// (max, ROW of the first maximum) of the lane's four columns; strict compare, NaN never wins (cpu/roi_pool_kernel.cpp:83-88)
template <typename T>
__device__ __forceinline__ void pool_quad(const typename Quad<T>::raw q, int h, float (&av)[4], int (&ah)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v = Quad<T>::cvt(q[j]);
    const bool gt = v > av[j];
    ah[j] = gt ? h : ah[j];
    av[j] = gt ? v : av[j];
  }
}

template <typename T, int PH, int PW>
__global__ __launch_bounds__(kPoolThreads) void roi_pool_fwd_cols(const T* __restrict__ input, const T* __restrict__ rois,
                                                                  T* __restrict__ output, int* __restrict__ argmax, int K,
                                                                  int C, int H, int W, double spatial_scale) {
  constexpr int kBins = PH * PW;
  constexpr int kWaves = kPoolThreads / 64;
  __shared__ float2 s_col[kWaves][2][256];   // per (channel slot, column): (max, bit pattern of its first index); one pooled row, double-buffered
  __shared__ int s_w[kWaves][8];             // per pw: w0 | w1 << 16 (window-relative), negative: empty column range
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // unit -> (RoI k, channel chunk): the feature planes of one channel chunk of one image are a few MB — they fit the
  // 4 MB L2 of an XCD, the whole map does not.  Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8 —
  // observed, a speed assumption only), so partition p = blockIdx % 8 owns the chunks c = p (mod 8) and walks the RoIs
  // (grouped by image in every caller we know) in order: an XCD's L2 then serves every re-read of a window instead of
  // the fabric (measured: L2 hit 20 % -> 91 %, HISTORY.md 4.4).  With fewer than 8 chunks the RoI list is split into
  // 8 / chunks contiguous slices instead.
  const int chunks = (C + kPoolChunk - 1) / kPoolChunk;
  int k, chunk;
  {
    const int part = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (chunks >= 8) {
      const int mine = (chunks - part + 7) >> 3;        // chunks owned by this partition: part, part + 8, ...
      const int64_t e = (int64_t)slot * kWaves + wave;  // entry in the partition's list, RoI fastest
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

  // bin tables of this RoI (lanes 0..7 compute one entry each)
  if (lane < 8) {
    const int w0 = clampi((int)floor((float)lane * bin_w) + rsw, 0, W), w1 = clampi((int)ceil((float)(lane + 1) * bin_w) + rsw, 0, W);
    s_w[wave][lane] = w1 <= w0 ? (int)0x80000000 : ((w0 - X0) | ((w1 - X0) << 16));
  }
  // G channels side by side: lane = (channel slot g, column quad q); a channel has cw = 4 * wpq column slots in LDS
  const int wq = (ww + 3) >> 2;
  const int wpq = wq <= 8 ? 8 : 16;
  const int G = 64 / wpq, cw = wpq * 4;  // 8 or 4 channels per pass
  const int g = lane / wpq, q = lane % wpq;
  // a quad never leaves its row (the last one is shifted left; re-read columns carry identical values); lanes beyond
  // the window / beyond C read a valid quad whose result is never stored
  const bool qvalid = q < wq;
  const int xl = min(X0 + (qvalid ? 4 * q : 0), W - 4);
  const int slot0 = xl - X0;  // window-relative column of the lane's first pixel (may be negative after the shift)
  // fold stage: lane = (channel slot g2, pooled column pw) for lanes below 7 G (<= 56)
  const int g2 = lane / PW, pw2 = lane % PW;
  const bool folds = lane < PW * G;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int wr = s_w[wave][folds ? pw2 : 0];
  const int fq0 = wr & 0xffff, fq1 = wr < 0 ? 0 : (wr >> 16);  // the bin's window-relative column range (empty: none)
  for (int cc = 0; cc < kPoolChunk; cc += G) {
    const int c = c0 + cc + g;
    const unsigned voff = (unsigned)((c < C ? c : c0) * H * W + xl) * (unsigned)sizeof(T);  // bytes
    float resv[PH];
    int resi[PH];
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
      float av[4] = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
      int ah[4] = {-1, -1, -1, -1};
      int h = hs[ph];
      const int e = he[ph];
      const T* rowp = img + (int64_t)h * W;
      for (; h + 4 <= e; h += 4) {  // four independent loads in flight
        const auto q0 = ld_quad(rowp, voff), q1 = ld_quad(rowp + W, voff), q2 = ld_quad(rowp + 2 * W, voff),
                   q3 = ld_quad(rowp + 3 * W, voff);
        pool_quad<T>(q0, h, av, ah);
        pool_quad<T>(q1, h + 1, av, ah);
        pool_quad<T>(q2, h + 2, av, ah);
        pool_quad<T>(q3, h + 3, av, ah);
        rowp += 4 * (int64_t)W;
      }
      for (; h < e; ++h) {
        pool_quad<T>(ld_quad(rowp, voff), h, av, ah);
        rowp += W;
      }
      float2* dst = &s_col[wave][ph & 1][g * cw];
      if (qvalid) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (slot0 + j >= 0) dst[slot0 + j] = make_float2(av[j], __int_as_float(ah[j] < 0 ? -1 : ah[j] * W + xl + j));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // fold the columns of bin (ph, pw2) of channel slot g2; the other buffer is being filled by the next pooled row
      const bool empty = wr < 0 || e <= hs[ph];
      float best = empty ? 0.f : -FLT_MAX;
      int besti = -1;
      if (folds && !empty) {
        const float2* sc = &s_col[wave][ph & 1][g2 * cw];
        for (int qq = fq0; qq < fq1; ++qq) {
          const float2 t = sc[qq];
          const int i = __float_as_int(t.y);
          // larger value wins; equal values: the smaller index = the earlier position of the reference's (h, w) scan
          if (t.x > best || (t.x == best && (unsigned)i < (unsigned)besti)) {
            best = t.x;
            besti = i;
          }
        }
      }
      resv[ph] = best;
      resi[ph] = besti;
    }
    if (folds && c0 + cc + g2 < C) {
      // write-once streams (2 x 200 MB at the measured shape) must not evict the feature planes from L2.  The seven
      // stores of a lane are 28 bytes apart; together the wave fills G x 196 contiguous bytes.
      // (Measured, round 4: parking a pass's [channel][bin] results in LDS and storing them as 16-byte pieces — 2 + 2 store
      // instructions per pass instead of 7 + 7, what gained the RoIAlign forward 5 % — made THIS kernel 20 % slower, 0.565 vs
      // 0.469 ms: two more wave barriers and an LDS round trip per 8-channel pass, 29 KB of LDS per workgroup.
      // profiles/r04_matrix_roipool_staged_stores.json)
      const int64_t o0 = out0 + (int64_t)(cc + g2) * kBins + pw2;
#pragma unroll
      for (int ph = 0; ph < PH; ++ph) {
        T outv;
        st(&outv, resv[ph]);
        if constexpr (sizeof(T) == 2) {
          unsigned short bits;
          __builtin_memcpy(&bits, &outv, 2);
          __builtin_nontemporal_store(bits, reinterpret_cast<unsigned short*>(output + o0 + ph * PW));
        } else {
          __builtin_nontemporal_store(outv, output + o0 + ph * PW);
        }
        __builtin_nontemporal_store(resi[ph], argmax + o0 + ph * PW);
      }
    }
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

// ---- roi_pool backward, one wave per feature PLANE ------------------------------------------------------------
// The reference scatters every pooled gradient to its argmax pixel with a global atomic (cuda/roi_pool_kernel.cu:
// 80-125; 50 M atomics at 4x256x100x168 / 4000 RoIs, overlapping RoIs hammer the same 64-byte granules: 1.85 ms,
// non-deterministic).  A gfx950 CU has 160 KB of LDS — a whole gradient plane (H*W floats, 67 KB here) fits.  So one
// wave owns plane (image b, channel c): it lists the RoIs of image b, walks their pooled gradients as one flat stream
// (coalesced argmax / grad reads, 16 x 64 elements in flight), accumulates into the LDS plane with ds_add_f32 in
// PROGRAM ORDER — one wave, in-order LDS: the same summation order every run, no global atomics — and writes the
// plane once (every pixel, zeros included: the caller's zero-fill is not needed).  16-bit gradients accumulate in
// fp32 and are rounded once.  Planes above kPlaneMaxPixels keep the atomic kernel above.
constexpr int kPlaneMaxPixels = 36864;  // 144 KB of LDS for the plane + 4 KB for the RoI list
constexpr int kPlaneList = 1024;

// A plane above kPlaneMaxPixels (a 200x336 FPN level) is cut into `nstrips` strips of `strip_px` pixels (whole rows); the
// wave that owns strip s of plane (b, c) walks the same gradient stream and keeps the contributions whose argmax pixel
// lies in its strip — still one owner per pixel, still program order, still no global atomics.
template <typename T>
__global__ __launch_bounds__(64) void roi_pool_bwd_plane(const T* __restrict__ grad, const T* __restrict__ rois,
                                                         const int* __restrict__ argmax, T* __restrict__ grad_input, int K,
                                                         int C, int HW_all, int strip_px, int nstrips, int bins, int PW,
                                                         int64_t ns, int64_t cs, int64_t hs, int64_t ws) {
  extern __shared__ float smem[];
  float* plane = smem;
  int* list = reinterpret_cast<int*>(smem + strip_px);
  const int lane = threadIdx.x;
  const int strip = blockIdx.x % nstrips, bc = blockIdx.x / nstrips;
  const int b = bc / C, c = bc % C;
  const int p0 = strip * strip_px, HW = min(strip_px, HW_all - p0);   // this wave's pixels: [p0, p0 + HW)
  for (int i = lane; i < HW; i += 64) plane[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += kPlaneList) {
    // RoIs of image b among [k0, k0 + kPlaneList), in index order
    // (eight 64-RoI groups of batch indices in flight, every load unconditional with a clamped index: one dependent load per
    // group — `kk < K && load == b` — made the scan of 4000 RoIs 63 round trips per plane)
    int cnt = 0;
    for (int j0 = 0; j0 < kPlaneList && k0 + j0 < K; j0 += 512) {
      float bi[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) bi[u] = ld(rois + (int64_t)min(k0 + j0 + u * 64 + lane, K - 1) * 5);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + u * 64;
        const int kk = k0 + j + lane;
        const bool mine = j < kPlaneList && kk < K && (int)bi[u] == b;
        const unsigned long long m = __ballot(mine);
        if (mine) list[cnt + __popcll(m & ((1ull << lane) - 1ull))] = kk;
        cnt += __popcll(m);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int total = cnt * bins;
    constexpr int U = 16;
    for (int e0 = 0; e0 < total; e0 += 64 * U) {
      int am[U];
      T graw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * 64 + lane;
        // unconditional (the stream position clamped to its last element, the surplus dropped below): behind `if (e < total)` each
        // of the 2 x 16 loads was followed by its own s_waitcnt vmcnt(0) — the phi copy of `am = -1; if (..) am = load`
        const int ec = min(e, total - 1);
        const int r = ec / bins, bin = ec - r * bins;
        const int64_t k = list[r];
        am[u] = argmax[(k * C + c) * bins + bin];   // raw: arithmetic on it here would put a wait between the loads
        graw[u] = grad[k * ns + c * cs + (bin / PW) * hs + (bin % PW) * ws];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int a = am[u] - p0;   // argmax -1 (empty bin) and pixels of other strips stay out
        if (e0 + u * 64 + lane < total && am[u] >= 0 && a >= 0 && a < HW) atomicAdd(&plane[a], (float)ld(&graw[u]));  // ds_add_f32, program order
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  T* dst = grad_input + ((int64_t)b * C + c) * HW_all + p0;
  for (int i = lane; i < HW; i += 64) st(dst + i, plane[i]);
}

inline bool roi_pool_bwd_plane_applies(tvmi_dtype dt, int64_t H, int64_t W, int64_t N, int64_t C) {
  return dt != TVMI_F64 && H * W <= kPlaneMaxPixels && H * W > 0 && N * C < (1ll << 31);
}
// RoIPool only: planes up to 8 strips of whole rows (the PS variants keep the single-plane regime)
constexpr int kPoolMaxStrips = 8;
inline int roi_pool_strips(int64_t H, int64_t W) {
  if (H * W <= kPlaneMaxPixels) return 1;
  if (W <= 0 || W > kPlaneMaxPixels) return 0;
  const int64_t rows = kPlaneMaxPixels / W;              // rows per strip
  const int64_t n = (H + rows - 1) / rows;
  return n <= kPoolMaxStrips ? (int)n : 0;
}
inline bool roi_pool_bwd_strips_apply(tvmi_dtype dt, int64_t H, int64_t W, int64_t N, int64_t C) {
  return dt != TVMI_F64 && H * W > 0 && roi_pool_strips(H, W) > 0 && N * C * kPoolMaxStrips < (1ll << 31);
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

// ---- ps_roi_align / ps_roi_pool backward, one wave per feature PLANE (same idea as roi_pool_bwd_plane) -----------
// Input channel c_in receives gradient from exactly one pooled cell (c_out, ph, pw) of every RoI of its image
// (c_in = (c_out*PH + ph)*PW + pw), so the wave that owns plane (b, c_in) walks the RoIs of image b — 64 at a time with
// lane = RoI for the loads and the per-RoI geometry, then one RoI after the other (values broadcast with v_readlane)
// with lane = sample / pixel of the cell — and accumulates in its LDS plane in program order: deterministic, no
// global atomics (the reference: cuda/ps_roi_align_kernel.cu:218-330, cuda/ps_roi_pool_kernel.cu:94-140).
// Arithmetic of the cell is the atomic kernels' above, expression for expression.
__device__ __forceinline__ float bcast(float v, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); }
__device__ __forceinline__ int bcast(int v, int j) { return __builtin_amdgcn_readlane(v, j); }

template <typename T, bool ALIGN>
__global__ __launch_bounds__(64) void ps_bwd_plane(const T* __restrict__ grad, const T* __restrict__ rois,
                                                   const int* __restrict__ channel_mapping, T* __restrict__ grad_input, int K,
                                                   int C, int H, int W, int PH, int PW, int C_out, double spatial_scale,
                                                   int sr) {
  extern __shared__ float smem[];
  const int HW = H * W;
  float* plane = smem;
  int* list = reinterpret_cast<int*>(smem + HW);
  const int lane = threadIdx.x;
  const int b = blockIdx.x / C, c_in = blockIdx.x % C;
  const int c_out = c_in / (PH * PW), ph = (c_in / PW) % PH, pw = c_in % PW;
  const float scale = (float)spatial_scale;
  for (int i = lane; i < HW; i += 64) plane[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += kPlaneList) {
    // (eight 64-RoI groups of batch indices in flight, every load unconditional with a clamped index: one dependent load per
    // group — `kk < K && load == b` — made the scan of 4000 RoIs 63 round trips per plane)
    int cnt = 0;
    for (int j0 = 0; j0 < kPlaneList && k0 + j0 < K; j0 += 512) {
      float bi[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) bi[u] = ld(rois + (int64_t)min(k0 + j0 + u * 64 + lane, K - 1) * 5);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + u * 64;
        const int kk = k0 + j + lane;
        const bool mine = j < kPlaneList && kk < K && (int)bi[u] == b;
        const unsigned long long m = __ballot(mine);
        if (mine) list[cnt + __popcll(m & ((1ull << lane) - 1ull))] = kk;
        cnt += __popcll(m);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int r0 = 0; r0 < cnt; r0 += 64) {
      // lane = RoI: loads and geometry
      const int nr = min(64, cnt - r0);
      const bool on = lane < nr;
      const int64_t k = on ? list[r0 + lane] : 0;
      const int64_t idx = ((k * C_out + c_out) * PH + ph) * PW + pw;
      const bool use = on && channel_mapping[idx] == c_in;  // the mapping tensor is the forward's formula
      const float go = use ? ld(grad + idx) : 0.f;
      const T* roi = rois + k * 5;
      float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f, f4 = 0.f;
      int i0 = 0, i1 = 0, i2 = 0, i3 = 0;
      if (ALIGN) {  // ps_roi_align_kernel above
        const float rsw = ld(roi + 1) * scale - 0.5f, rsh = ld(roi + 2) * scale - 0.5f;
        const float rew = ld(roi + 3) * scale - 0.5f, reh = ld(roi + 4) * scale - 0.5f;
        const float rw = rew - rsw, rh = reh - rsh;
        const float bin_h = rh / (float)PH, bin_w = rw / (float)PW;
        f0 = (float)ph * bin_h + rsh;  // hstart
        f1 = (float)pw * bin_w + rsw;  // wstart
        f2 = bin_h;
        f3 = bin_w;
        f4 = go;
        i0 = sr > 0 ? sr : (int)ceil(rh / (float)PH);  // gh
        i1 = sr > 0 ? sr : (int)ceil(rw / (float)PW);  // gw
        if (!use) i0 = 0;
      } else {  // ps_roi_pool_kernel above, backward flavour (roundf, clip to H / W)
        const int rsw = (int)roundf((float)(ld(roi + 1) * scale)), rsh = (int)roundf((float)(ld(roi + 2) * scale));
        const int rew = (int)roundf((float)(ld(roi + 3) * scale)), reh = (int)roundf((float)(ld(roi + 4) * scale));
        const int rw = max(rew - rsw, 1), rh = max(reh - rsh, 1);
        const float bin_h = (float)rh / (float)PH, bin_w = (float)rw / (float)PW;
        i0 = clampi((int)floor((float)ph * bin_h) + rsh, 0, H);        // hs
        i1 = clampi((int)ceil((float)(ph + 1) * bin_h) + rsh, 0, H);   // he
        i2 = clampi((int)floor((float)pw * bin_w) + rsw, 0, W);        // ws
        i3 = clampi((int)ceil((float)(pw + 1) * bin_w) + rsw, 0, W);   // we
        const bool empty = (i1 <= i0) || (i3 <= i2);
        f0 = empty ? 0.f : go / (float)((i1 - i0) * (i3 - i2));       // diff
        if (!use || empty) i1 = i0;
      }
      if (ALIGN) {
        // small sampling grids (the usual sampling_ratio 2): lane = (RoI slot, sample), 64 / S RoIs per step; the taps of
        // different RoIs may meet on a pixel inside one ds_add — the LDS serialises them in lane order
        int smax = i0 * i1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) smax = max(smax, __shfl_xor(smax, o));
        if (smax <= 16) {
          const int S = smax <= 1 ? 1 : (smax <= 2 ? 2 : (smax <= 4 ? 4 : (smax <= 8 ? 8 : 16)));
          const int per = 64 / S, slot = lane / S, sidx = lane % S;
          for (int j0 = 0; j0 < nr; j0 += per) {
            const int j = j0 + slot;
            const int src = min(j, 63);
            const int gh = __shfl(i0, src), gw = __shfl(i1, src);
            const float hstart = __shfl(f0, src), wstart = __shfl(f1, src), bin_h = __shfl(f2, src), bin_w = __shfl(f3, src);
            const float g = __shfl(f4, src);
            if (j < nr && sidx < gh * gw) {
              const float count = (float)(gh * gw);
              const int iy = sidx / gw, ix = sidx - iy * gw;
              const float y = hstart + (float)((float)iy + .5f) * bin_h / (float)gh;
              const float x = wstart + (float)((float)ix + .5f) * bin_w / (float)gw;
              int yl, yh, xl, xh;
              float w1, w2, w3, w4;
              if (bilinear_setup<float>(H, W, y, x, yl, yh, xl, xh, w1, w2, w3, w4)) {
                atomicAdd(&plane[yl * W + xl], g * w1 / count);
                atomicAdd(&plane[yl * W + xh], g * w2 / count);
                atomicAdd(&plane[yh * W + xl], g * w3 / count);
                atomicAdd(&plane[yh * W + xh], g * w4 / count);
              }
            }
          }
          continue;
        }
      }
      // one RoI after the other: lane = sample (align, large adaptive grids) / pixel (pool) of its cell
      for (int j = 0; j < nr; ++j) {
        if (ALIGN) {
          const int gh = bcast(i0, j), gw = bcast(i1, j);
          if (gh <= 0 || gw <= 0) continue;
          const float hstart = bcast(f0, j), wstart = bcast(f1, j), bin_h = bcast(f2, j), bin_w = bcast(f3, j), g = bcast(f4, j);
          const float count = (float)(gh * gw);
          for (int sidx = lane; sidx < gh * gw; sidx += 64) {
            const int iy = sidx / gw, ix = sidx - iy * gw;
            const float y = hstart + (float)((float)iy + .5f) * bin_h / (float)gh;
            const float x = wstart + (float)((float)ix + .5f) * bin_w / (float)gw;
            int yl, yh, xl, xh;
            float w1, w2, w3, w4;
            if (!bilinear_setup<float>(H, W, y, x, yl, yh, xl, xh, w1, w2, w3, w4)) continue;
            atomicAdd(&plane[yl * W + xl], g * w1 / count);
            atomicAdd(&plane[yl * W + xh], g * w2 / count);
            atomicAdd(&plane[yh * W + xl], g * w3 / count);
            atomicAdd(&plane[yh * W + xh], g * w4 / count);
          }
        } else {
          const int hs_ = bcast(i0, j), he_ = bcast(i1, j);
          if (he_ <= hs_) continue;
          const int ws_ = bcast(i2, j), we_ = bcast(i3, j);
          const float diff = bcast(f0, j);
          const int bw = we_ - ws_, npx = (he_ - hs_) * bw;
          for (int px = lane; px < npx; px += 64) {
            const int dh = px / bw, dw = px - dh * bw;
            plane[(hs_ + dh) * W + ws_ + dw] += diff;  // pixels of one cell are distinct: no conflict inside a step
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  T* dst = grad_input + ((int64_t)b * C + c_in) * HW;
  for (int i = lane; i < HW; i += 64) st(dst + i, plane[i]);
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
  if (pooled_h == 7 && pooled_w == 7 && dt != TVMI_F64 && C * H * W < (1ll << 29) && W >= 4 && H >= 1 &&
      K * ((C + tvmi::kPoolChunk - 1) / tvmi::kPoolChunk) < (1ll << 31)) {
    // grid: 8 partitions (see the kernel) x the workgroups of the longest partition
    const int64_t chunks = (C + tvmi::kPoolChunk - 1) / tvmi::kPoolChunk, wpb = tvmi::kPoolThreads / 64;
    int64_t blocks;
    if (chunks >= 8) blocks = 8 * ((((chunks + 7) / 8) * K + wpb - 1) / wpb);
    else if (8 % chunks == 0) blocks = 8 * (((K + (8 / chunks) - 1) / (8 / chunks) + wpb - 1) / wpb);
    else blocks = (K * chunks + wpb - 1) / wpb;
    const dim3 grid((unsigned)blocks);
#define TVMI_POOL_COLS(scalar_t)                                                                                      \
  tvmi::roi_pool_fwd_cols<scalar_t, 7, 7><<<grid, dim3(tvmi::kPoolThreads), 0, s>>>((const scalar_t*)input, (const scalar_t*)rois, \
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


namespace tvmi {
namespace {
// K == 0 in the plane-owner regime: the `*_backward_overwrites` queries still answer 1 (they do not know K), so a C-ABI
// caller may have handed over an uninitialised buffer — honour the contract and write the zeros.
inline int zero_planes_if_owner(tvmi_dtype dt, void* grad_input, int64_t N, int64_t C, int64_t H, int64_t W, hipStream_t s,
                                const char* what) {
  if (N * C * H * W == 0 || !roi_pool_bwd_plane_applies(dt, H, W, N, C)) return 0;
  if (!grad_input) return set_error((int)hipErrorInvalidValue, what);
  const size_t esz = dt == TVMI_F32 ? 4 : dt == TVMI_F64 ? 8 : 2;
  const hipError_t e = hipMemsetAsync(grad_input, 0, (size_t)(N * C * H * W) * esz, s);
  return e == hipSuccess ? 0 : set_error((int)e, what);
}
}  // namespace
}  // namespace tvmi

extern "C" int tvmi_roi_pool_backward(const void* grad, const void* rois, const int32_t* argmax,
                                      void* grad_input, tvmi_dtype dt, int64_t N, int64_t C, int64_t H,
                                      int64_t W, int64_t K, int64_t pooled_h, int64_t pooled_w,
                                      int64_t n_stride, int64_t c_stride, int64_t h_stride,
                                      int64_t w_stride, void* stream) {
  const int64_t total = K * C * pooled_h * pooled_w;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (total == 0) {   // K == 0: honour the "overwrites" contract of the owner regime
    if (N * C * H * W == 0 || !tvmi::roi_pool_bwd_strips_apply(dt, H, W, N, C)) return 0;
    TVMI_CHECK_ARG(grad_input, "roi_pool_backward: null pointer");
    const size_t esz = dt == TVMI_F32 ? 4 : 2;
    const hipError_t e = hipMemsetAsync(grad_input, 0, (size_t)(N * C * H * W) * esz, s);
    return e == hipSuccess ? 0 : ::tvmi::set_error((int)e, "roi_pool_backward: zero fill");
  }
  if (N * C * H * W == 0) return 0;
  TVMI_CHECK_ARG(grad && rois && argmax && grad_input, "roi_pool_backward: null pointer");
  if (tvmi::roi_pool_bwd_strips_apply(dt, H, W, N, C)) {
    const int nstrips = tvmi::roi_pool_strips(H, W);
    const int64_t strip_px = nstrips == 1 ? H * W : ((H + nstrips - 1) / nstrips) * W;   // whole rows, evenly
    const size_t lds = (size_t)(strip_px + tvmi::kPlaneList) * sizeof(float);
#define TVMI_POOL_PLANE(scalar_t)                                                                                      \
  do {                                                                                                                 \
    auto kern = tvmi::roi_pool_bwd_plane<scalar_t>;                                                                    \
    if (lds > 64 * 1024)                                                                                               \
      TVMI_CHECK_ARG(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess, \
                     "roi_pool_backward: cannot reserve the LDS plane");                                               \
    kern<<<dim3((unsigned)(N * C * nstrips)), dim3(64), lds, s>>>((const scalar_t*)grad, (const scalar_t*)rois, argmax, \
                                                        (scalar_t*)grad_input, (int)K, (int)C, (int)(H * W),          \
                                                        (int)strip_px, nstrips,                                        \
                                                        (int)(pooled_h * pooled_w), (int)pooled_w, n_stride, c_stride, \
                                                        h_stride, w_stride);                                           \
  } while (0)
    if (dt == TVMI_F32) TVMI_POOL_PLANE(float);
    else if (dt == TVMI_F16) TVMI_POOL_PLANE(__half);
    else if (dt == TVMI_BF16) TVMI_POOL_PLANE(__hip_bfloat16);
    else return ::tvmi::set_error(hipErrorInvalidValue, "roi_pool_backward: unsupported dtype");
#undef TVMI_POOL_PLANE
    TVMI_RETURN_LAUNCH_STATUS("tvmi_roi_pool_backward");
  }
  TVMI_DISPATCH_FLOAT(dt, "roi_pool_backward",
                      tvmi::roi_pool_bwd<scalar_t><<<grid_for(total), dim3(kThreads), 0, s>>>(
                          (const scalar_t*)grad, (const scalar_t*)rois, argmax, (scalar_t*)grad_input, total,
                          (int)C, (int)H, (int)W, (int)pooled_h, (int)pooled_w, n_stride, c_stride, h_stride,
                          w_stride));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_roi_pool_backward");
}

extern "C" int tvmi_roi_pool_backward_overwrites(tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W) {
  return tvmi::roi_pool_bwd_strips_apply(dt, H, W, N, C) ? 1 : 0;
}
// the position-sensitive backward entries own whole planes only (no strips)
extern "C" int tvmi_ps_roi_backward_overwrites(tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W) {
  return tvmi::roi_pool_bwd_plane_applies(dt, H, W, N, C) ? 1 : 0;
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
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (total == 0) return tvmi::zero_planes_if_owner(dt, grad_input, N, C, H, W, s, "ps_roi_align_backward: zero fill");
  if (N * C * H * W == 0) return 0;
  TVMI_CHECK_ARG(grad && rois && channel_mapping && grad_input, "ps_roi_align_backward: null pointer");
  if (tvmi::roi_pool_bwd_plane_applies(dt, H, W, N, C)) {  // plane-owner regime: see ps_bwd_plane
    const size_t lds = (size_t)(H * W + tvmi::kPlaneList) * sizeof(float);
#define TVMI_PS_PLANE(scalar_t)                                                                                       \
  do {                                                                                                                \
    auto kern = tvmi::ps_bwd_plane<scalar_t, true>;                                                                  \
    if (lds > 64 * 1024)                                                                                              \
      TVMI_CHECK_ARG(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess, \
                     "ps_roi_align_backward: cannot reserve the LDS plane");                                           \
    kern<<<dim3((unsigned)(N * C)), dim3(64), lds, s>>>((const scalar_t*)grad, (const scalar_t*)rois, channel_mapping, \
                                                        (scalar_t*)grad_input, (int)K, (int)C, (int)H, (int)W,        \
                                                        (int)pooled_h, (int)pooled_w, (int)C_out, spatial_scale, (int)sampling_ratio); \
  } while (0)
    if (dt == TVMI_F32) TVMI_PS_PLANE(float);
    else if (dt == TVMI_F16) TVMI_PS_PLANE(__half);
    else if (dt == TVMI_BF16) TVMI_PS_PLANE(__hip_bfloat16);
    else return ::tvmi::set_error(hipErrorInvalidValue, "ps_roi_align_backward: unsupported dtype");
#undef TVMI_PS_PLANE
    TVMI_RETURN_LAUNCH_STATUS("tvmi_ps_roi_align_backward");
  }
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
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (total == 0) return tvmi::zero_planes_if_owner(dt, grad_input, N, C, H, W, s, "ps_roi_pool_backward: zero fill");
  if (N * C * H * W == 0) return 0;
  TVMI_CHECK_ARG(grad && rois && channel_mapping && grad_input, "ps_roi_pool_backward: null pointer");
  if (tvmi::roi_pool_bwd_plane_applies(dt, H, W, N, C)) {  // plane-owner regime: see ps_bwd_plane
    const size_t lds = (size_t)(H * W + tvmi::kPlaneList) * sizeof(float);
#define TVMI_PS_PLANE(scalar_t)                                                                                       \
  do {                                                                                                                \
    auto kern = tvmi::ps_bwd_plane<scalar_t, false>;                                                                  \
    if (lds > 64 * 1024)                                                                                              \
      TVMI_CHECK_ARG(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess, \
                     "ps_roi_pool_backward: cannot reserve the LDS plane");                                           \
    kern<<<dim3((unsigned)(N * C)), dim3(64), lds, s>>>((const scalar_t*)grad, (const scalar_t*)rois, channel_mapping, \
                                                        (scalar_t*)grad_input, (int)K, (int)C, (int)H, (int)W,        \
                                                        (int)pooled_h, (int)pooled_w, (int)C_out, spatial_scale, 0); \
  } while (0)
    if (dt == TVMI_F32) TVMI_PS_PLANE(float);
    else if (dt == TVMI_F16) TVMI_PS_PLANE(__half);
    else if (dt == TVMI_BF16) TVMI_PS_PLANE(__hip_bfloat16);
    else return ::tvmi::set_error(hipErrorInvalidValue, "ps_roi_pool_backward: unsupported dtype");
#undef TVMI_PS_PLANE
    TVMI_RETURN_LAUNCH_STATUS("tvmi_ps_roi_pool_backward");
  }
  TVMI_DISPATCH_FLOAT(dt, "ps_roi_pool_backward",
                      (tvmi::ps_roi_pool_kernel<scalar_t, true><<<grid_for(total), dim3(kThreads), 0, s>>>(
                          (const scalar_t*)grad, (const scalar_t*)rois, (scalar_t*)grad_input,
                          const_cast<int32_t*>(channel_mapping), total, (int)C, (int)H, (int)W, (int)pooled_h,
                          (int)pooled_w, (int)C_out, spatial_scale)));
  TVMI_RETURN_LAUNCH_STATUS("tvmi_ps_roi_pool_backward");
}
