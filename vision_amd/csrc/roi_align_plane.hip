// roi_align_plane.hip — RoIAlign forward, SHARED-STAGING kernels for gfx950 (7x7 / 14x14 bins, sampling_ratio 2).
//
// Semantics: torchvision/csrc/ops/cpu/roi_align_kernel.cpp:18-115 + cpu/roi_align_common.h:32-124, like roi_align.hip.
// Reference structure being replaced: cuda/roi_align_kernel.cu:68-143 (one thread per output element, 16 dependent
// scattered loads each).  roi_align.hip's wave kernel stages ONE RoI's window per (RoI, 32 channels) wave: at the FPN
// workload every pixel of the coarse levels is wanted by 6-30 RoIs and is re-staged every time, out of 60-160-byte row
// fragments that cost the texture addresser ~2.4 accesses per useful 64 bytes (profiles/r01_pmc_roi_align_dma_summary.txt).
//
// Here the MAP is staged, not the RoI:
//   * a 512-thread workgroup owns (level, image, row band, channel group): it copies the band — full-width rows, one
//     contiguous run per channel — HBM -> LDS with 16-byte LDS-DMA pieces (global_load_lds_dwordx4; every piece a useful,
//     aligned 16 bytes), then its 8 waves serve EVERY RoI of that image and level whose sampled rows lie in the band;
//   * a level whose whole plane fits the LDS budget is one band (P3: 1 channel, P4: 4, P5: 18 channels per workgroup at
//     the 800x1344 FPN shapes); a larger map (P2) is cut into half-overlapping bands of full-width rows;
//   * lane = output bin (as in the wave kernel): 8 ds_read2_b32 + the separable FMAs per channel, one coalesced
//     non-temporal store of the 49 outputs;
//   * a pre-pass (one 32/64-lane group per RoI) computes, once per RoI instead of once per (RoI, channel group): the FPN
//     level, the band the RoI belongs to (or "not eligible": the wave kernel of roi_align.hip keeps those), and the table
//     of its 28 (56) axis samples {low index, fraction} that the lanes of the serving wave then simply load;
//   * whether a level is served here at all is decided ON THE DEVICE from the pre-pass's per-level sum of window pixels
//     (staging a map pays when the RoIs would otherwise stage more than the map): no host synchronisation, and both
//     kernels read the same integers, so they always agree on who owns a RoI.
// Zero-weight taps (roi_align_common.h:60-73 `continue`, and the x_high = x_low edge): a skipped sample reads a
// zeroed 16-byte cell, the y edge uses the reference's own y_high = y_low row, and the x edge — re-expressed on the pair
// (W-2, W-1) with factors (0, 1) so the pair stays one LDS read — multiplies its untouched pixel with v_mul_legacy_f32
// (0 * anything = 0): a NaN / Inf pixel the reference never reads cannot leak into the result.
#include <algorithm>
#include <string>
#include <type_traits>

#include "roi_common.h"

namespace tvmi {
namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ int level_window_px(const int* __restrict__ blocksum, int l) {
  // every consumer sums the same integers: the result does not depend on the order
  const int lane = threadIdx.x & 63;
  int v = lane < kPlanePreBlocks ? blocksum[lane * kMaxLevels + l] : 0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// ---------------------------------------------------------------------------------------
// Pre-pass: LP lanes per RoI (32 for 7x7, 64 for 14x14), one axis sample per lane.
template <typename R, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kPlanePreThreads) void roi_fwd_prepass(MsLevels lv, PlanePlan plan, const R* __restrict__ rois,
                                                                    int K, int aligned, int multiscale,
                                                                    int* __restrict__ key, float2* __restrict__ axis,
                                                                    int* __restrict__ blocksum) {
  constexpr int NY = PHT * SRT, NX = PWT * SRT, LP = plane_lanes_per_roi(PHT, PWT, SRT);
  constexpr int RPB = kPlanePreThreads / LP;  // RoIs per block and iteration
  __shared__ int s_px[kMaxLevels];
  if (threadIdx.x < kMaxLevels) s_px[threadIdx.x] = 0;
  __syncthreads();
  const int sl = threadIdx.x % LP;
  for (int k0 = blockIdx.x * RPB; k0 < K; k0 += gridDim.x * RPB) {
    const int k = k0 + threadIdx.x / LP;
    const bool live = k < K;
    const R* roi = rois + (int64_t)(live ? k : K - 1) * 5;
    const int l = multiscale ? fpn_level<R>(roi, lv) : 0;
    int H = 0, W = 0;
    float scale = 0.f;
    PlaneLevel pl = plan.lv[0];
#pragma unroll
    for (int i = 0; i < kMaxLevels; ++i)
      if (i == l) {
        H = lv.H[i];
        W = lv.W[i];
        scale = lv.scale[i];
        pl = plan.lv[i];
      }
    const RoiGeom<float> g = roi_geom<R, float>(roi, scale, PHT, PWT, SRT, aligned != 0);
    int lo = -1, hi = -1;
    float fr = 0.f;
    if (sl < NY) {
      int a, b;
      float l_, h_;
      if (axis_sample<float>(H, g.start_h, g.bin_h, SRT, sl / SRT, sl % SRT, a, b, l_, h_)) {
        lo = a;
        hi = b;
        fr = l_;
      }
    } else if (sl < NY + NX) {
      const int s = sl - NY;
      int a;
      float l_, h_;
      if (W >= 2 && axis_sample_shifted(W, g.start_w, g.bin_w, SRT, s / SRT, s % SRT, a, l_, h_)) {
        lo = a;
        hi = a + 1;
        fr = l_;
      }
    }
    const bool isy = sl < NY, isx = !isy && sl < NY + NX;
    int ymin = (isy && lo >= 0) ? lo : 0x7fffffff, ymax = (isy && lo >= 0) ? hi : -1;
    int xmin = (isx && lo >= 0) ? lo : 0x7fffffff, xmax = (isx && lo >= 0) ? hi : -1;
#pragma unroll
    for (int m = LP / 2; m >= 1; m >>= 1) {
      ymin = min(ymin, __shfl_xor(ymin, m));
      ymax = max(ymax, __shfl_xor(ymax, m));
      xmin = min(xmin, __shfl_xor(xmin, m));
      xmax = max(xmax, __shfl_xor(xmax, m));
    }
    bool ok = pl.enabled != 0 && g.batch >= 0 && g.batch < plan.N && g.batch < 32768 && W >= 2 && H >= 1;
    int band = 0, px = 0;
    if (ok && ymax >= 0 && xmax >= 0) {
      band = min(ymin / pl.S, pl.nbands - 1);
      const int r0 = min(band * pl.S, H - pl.B);
      ok = ymin >= r0 && ymax < r0 + pl.B;
      // what the per-RoI stager would move for this RoI and one channel (only the sampled rows of a tall window)
      px = min(ymax - ymin + 1, 2 * NY) * (xmax - xmin + 1);
    }
    if (live && sl == 0) {
      key[k] = ok ? ((g.batch << 16) | (l << 12) | band) : -1;
      if (ok && px > 0) atomicAdd(&s_px[l], min(px, 1 << 14));
    }
    if (live && sl < NY + NX) axis[(int64_t)k * LP + sl] = make_float2(__int_as_float(lo), fr);
  }
  __syncthreads();
  if (threadIdx.x < kMaxLevels) blocksum[blockIdx.x * kMaxLevels + threadIdx.x] = min(s_px[threadIdx.x], 1 << 24);
}

// ---------------------------------------------------------------------------------------
// The serving kernel.
template <typename T>
__device__ __forceinline__ float tap_pair(const char* __restrict__ p, float l, float h) {
  // h * p[0] + l * p[1]; the h product is the one that may carry an exact-zero weight onto a pixel the reference does
  // not read (shifted x edge): v_mul_legacy_f32 makes 0 * NaN = 0
  const T* q = reinterpret_cast<const T*>(p);
  return __builtin_fmaf(l, ld(q + 1), mul_legacy(h, ld(q)));
}

// per-lane sample set-up of one RoI from its axis table
template <typename T, int PHT, int PWT, int SRT>
struct LaneSetup {
  static constexpr int PHW = PHT * PWT, NB = (PHW + 63) / 64, NS = SRT * SRT;
  int off[NB][NS][2];
  float fy[NB][SRT][2], fx[NB][SRT][2];
};

template <int PHT, int PWT, int SRT>
struct AxisEntries {  // the raw table entries a lane needs (loaded one RoI ahead)
  static constexpr int NB = (PHT * PWT + 63) / 64;
  float2 ey[NB][SRT], ex[NB][SRT];
};

template <int PHT, int PWT, int SRT>
__device__ __forceinline__ void load_axis(AxisEntries<PHT, PWT, SRT>& a, const float2* __restrict__ ax, int lane) {
  constexpr int PHW = PHT * PWT, NB = (PHW + 63) / 64, NY = PHT * SRT;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int bin = min(lane + 64 * b, PHW - 1);
    const int ph = bin / PWT, pw = bin - ph * PWT;
#pragma unroll
    for (int i = 0; i < SRT; ++i) {
      a.ey[b][i] = ax[ph * SRT + i];
      a.ex[b][i] = ax[NY + pw * SRT + i];
    }
  }
}

template <typename T, int PHT, int PWT, int SRT>
__device__ __forceinline__ void make_setup(LaneSetup<T, PHT, PWT, SRT>& s, const AxisEntries<PHT, PWT, SRT>& a, int r0, int H, int rowb) {
  constexpr int NB = (PHT * PWT + 63) / 64;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    int row0[SRT], row1[SRT], col[SRT];
    bool vy[SRT], vx[SRT];
#pragma unroll
    for (int i = 0; i < SRT; ++i) {
      const int ylo = __float_as_int(a.ey[b][i].x), xlo = __float_as_int(a.ex[b][i].x);
      vy[i] = ylo >= 0;
      vx[i] = xlo >= 0;
      row0[i] = 16 + (ylo - r0) * rowb;
      row1[i] = row0[i] + (ylo < H - 1 ? rowb : 0);  // y edge: y_high = y_low, like the reference
      col[i] = xlo * (int)sizeof(T);
      s.fy[b][i][0] = a.ey[b][i].y;
      s.fy[b][i][1] = 1.f - a.ey[b][i].y;
      s.fx[b][i][0] = a.ex[b][i].y;
      s.fx[b][i][1] = 1.f - a.ex[b][i].y;
    }
#pragma unroll
    for (int iy = 0; iy < SRT; ++iy)
#pragma unroll
      for (int ix = 0; ix < SRT; ++ix) {
        const bool v = vy[iy] && vx[ix];
        s.off[b][iy * SRT + ix][0] = v ? row0[iy] + col[ix] : 0;  // skipped sample: the zero cell
        s.off[b][iy * SRT + ix][1] = v ? row1[iy] + col[ix] : 0;
      }
  }
}

// Key scan: thread t of a batch starting at kb owns the keys kb + 4 * (i * kPlaneThreads + t) + {0..3}, i = 0 .. KPT/4 - 1
// (one 16-byte load each; the key array is 16-byte aligned and padded to a multiple of 4 by the workspace layout).
template <int KPT>
__device__ __forceinline__ int key_index(int kb, int j, int tid) {
  return kb + 4 * ((j >> 2) * kPlaneThreads + tid) + (j & 3);
}
template <int KPT>
__device__ __forceinline__ void load_keys(int (&kv)[KPT], const int* __restrict__ key, int kb, int K, int tid) {
  static_assert(KPT % 4 == 0, "keys in quads");
#pragma unroll
  for (int i = 0; i < KPT / 4; ++i) {
    const int k0 = kb + 4 * (i * kPlaneThreads + tid);
    int4 v = make_int4(-1, -1, -1, -1);
    if (k0 < K) v = *reinterpret_cast<const int4*>(key + k0);   // the tail quad is padded inside the workspace
    kv[4 * i + 0] = k0 + 0 < K ? v.x : -1;
    kv[4 * i + 1] = k0 + 1 < K ? v.y : -1;
    kv[4 * i + 2] = k0 + 2 < K ? v.z : -1;
    kv[4 * i + 3] = k0 + 3 < K ? v.w : -1;
  }
}

template <typename T, int PHT, int PWT, int SRT>
__global__ __launch_bounds__(kPlaneThreads) void roi_align_fwd_plane(MsLevels lv, PlanePlan plan, const int* __restrict__ key,
                                                                     const float2* __restrict__ axis,
                                                                     const int* __restrict__ blocksum, T* __restrict__ output,
                                                                     int C, int K) {
  constexpr int PHW = PHT * PWT, NB = (PHW + 63) / 64, NS = SRT * SRT;
  constexpr int LP = plane_lanes_per_roi(PHT, PWT, SRT);
  constexpr int EPP = 16 / (int)sizeof(T);
  constexpr int KPT = 8;  // keys per thread and scan batch, fetched as two 16-byte loads (VMEM instructions are what this
                          // kernel is short of: the texture path retires a wave's load in ~20 cycles whatever its width)
  static_assert(SRT == 2, "separable 2x2 sampling");
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  int* const s_list = reinterpret_cast<int*>(smem_all);  // [kPlaneListCap] RoIs this workgroup serves
  char* const smem = smem_all + kPlaneListCap * 4;        // the staged band
  __shared__ int s_cnt[64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- which (level, image, band, channel group) is this workgroup?  Blocks of one level are numbered so that block b
  // (observed on XCD b % 8 — speed only) gets every band of the channel groups gid = x (mod 8): neighbouring bands, which
  // share half their rows, meet in one L2.
  int l = -1;
#pragma unroll
  for (int i = 0; i < kMaxLevels; ++i)
    if (plan.lv[i].enabled && (int)blockIdx.x >= plan.lv[i].block_base && (int)blockIdx.x < plan.lv[i].block_base + plan.lv[i].nblocks) l = i;
  if (l < 0) return;
  PlaneLevel pl = plan.lv[0];
  int H = 0, W = 0;
  const T* src = nullptr;
#pragma unroll
  for (int i = 0; i < kMaxLevels; ++i)
    if (i == l) {
      pl = plan.lv[i];
      H = lv.H[i];
      W = lv.W[i];
      src = static_cast<const T*>(lv.ptr[i]);
    }
  const int q = (int)blockIdx.x - pl.block_base;
  const int r = q >> 3;
  const int split = r % pl.nsplit, r2 = r / pl.nsplit;
  const int gid = (r2 / pl.nbands) * 8 + (q & 7), band = r2 % pl.nbands;
  if (gid >= plan.N * pl.ngroups) return;
  // the device-side decision "is this level served here": same integers, same rule as the wave kernel (roi_align.hip)
  if (!plane_level_active(level_window_px(blocksum, l), pl, plan.gain_x16, plan.N, H, W)) return;
  const int img = gid / pl.ngroups, cgi = gid - img * pl.ngroups;
  const int c0 = cgi * pl.cg, cc = min(pl.cg, C - c0);
  const int r0 = min(band * pl.S, H - pl.B);
  const int n = pl.B * W;                                   // elements of one channel's band
  const int chs = 16 + ((n * (int)sizeof(T) + 15) & ~15);  // bytes per channel region: [16-byte zero cell][band]
  const int want = (img << 16) | (l << 12) | band;

  // ---- the first batch of keys is requested before the band, so that the list is being built while the band lands
  int kv[KPT];
  load_keys<KPT>(kv, key, 0, K, tid);
  // ---- stage the band: per channel one contiguous run of n elements, 16-byte LDS-DMA pieces, instruction f = (c, i)
  // issued by wave f % 8; lanes past the run are masked off (they must not write the next channel's region)
  {
    const int npf = n / EPP, ipc = (npf + 63) >> 6;
    const T* band0 = src + (((int64_t)img * C + c0) * H + r0) * W;
    for (int f = wave; f < cc * ipc; f += kPlaneThreads / 64) {
      const int c = f / ipc, i = f - c * ipc;
      const int piece = i * 64 + lane;
      if (piece < npf)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(band0 + (int64_t)c * H * W + (int64_t)piece * EPP),
                                         (lds_ptr_t)(smem + c * chs + 16 + i * 1024), 16, 0, 0);
    }
    const int tail = n - npf * EPP;  // run length not a multiple of a piece: the last 1..EPP-1 elements by hand
    if (tid < cc * tail) {
      const int c = tid / tail, e = npf * EPP + tid % tail;
      reinterpret_cast<T*>(smem + c * chs + 16)[e] = band0[(int64_t)c * H * W + e];
    }
    if (tid < cc * 4) reinterpret_cast<int*>(smem + (tid >> 2) * chs)[tid & 3] = 0;  // the zero cells
  }
  // ---- the RoIs of this (image, level, band): compacted into the LDS list in a FIXED order (batch, key slot j, wave,
  // lane) — the nsplit workgroups that share a band each build the list for themselves and must agree on which entry
  // is whose.  Beyond kPlaneListCap entries a thread remembers what it could not list; split 0 serves those afterwards.
  unsigned unlisted = 0;
  int n_all = 0;
  for (int kb = 0; kb < K; kb += KPT * kPlaneThreads) {
    if (kb > 0) load_keys<KPT>(kv, key, kb, K, tid);
    unsigned long long bals[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      bals[j] = __ballot(kv[j] == want);
      if (lane == 0) s_cnt[j * (kPlaneThreads / 64) + wave] = __popcll(bals[j]);
    }
    __syncthreads();
    static_assert(KPT * (kPlaneThreads / 64) == 64, "one count per lane");
    const int cnt = s_cnt[lane];
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    const int excl = incl - cnt;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int base = n_all + __builtin_amdgcn_readlane(excl, j * (kPlaneThreads / 64) + wave);
      if (kv[j] == want) {
        const int pos = base + __popcll(bals[j] & ((1ull << lane) - 1ull));
        if (pos < kPlaneListCap) s_list[pos] = key_index<KPT>(kb, j, tid);
        else if (kb == 0) unlisted |= 1u << j;   // only the first batch is remembered; later batches are re-derived below
      }
    }
    n_all += __builtin_amdgcn_readlane(incl, 63);
    __syncthreads();   // s_cnt is reused by the next batch
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int n_list = min(n_all, kPlaneListCap);

  const int rowb = W * (int)sizeof(T);
  auto serve = [&](const LaneSetup<T, PHT, PWT, SRT>& su, int k) {
    T* out = output + ((int64_t)k * C + c0) * PHW;
    for (int c = 0; c < cc; ++c) {
      const char* cb = smem + c * chs;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int bin = lane + 64 * b;
        if (bin < PHW) {  // the whole bin under the mask: idle lanes issue no LDS reads, and the tap pairs stay in one block
          float acc = 0.f;
#pragma unroll
          for (int iy = 0; iy < SRT; ++iy)
#pragma unroll
            for (int ix = 0; ix < SRT; ++ix) {
              const float t0 = tap_pair<T>(cb + su.off[b][iy * SRT + ix][0], su.fx[b][ix][0], su.fx[b][ix][1]);
              const float t1 = tap_pair<T>(cb + su.off[b][iy * SRT + ix][1], su.fx[b][ix][0], su.fx[b][ix][1]);
              acc = __builtin_fmaf(su.fy[b][iy][1], t0, acc);
              acc = __builtin_fmaf(su.fy[b][iy][0], t1, acc);
            }
          const float res = acc * (1.f / (float)NS);
          if constexpr (std::is_same<T, float>::value)
            __builtin_nontemporal_store(res, out + c * PHW + bin);
          else
            st(out + c * PHW + bin, res);
        }
      }
    }
  };

  // ---- serve the list: entry e goes to wave e % 8 (equal cost per RoI); the axis-table entries of the next RoI are
  // requested before the current one is served (one RoI = ~50 instructions per channel: a dependent L2 round trip per
  // RoI would show)
  {
    AxisEntries<PHT, PWT, SRT> cur, nxt;
    const int estep = (kPlaneThreads / 64) * pl.nsplit;   // this workgroup serves entries e = split (mod nsplit)
    int e = wave * pl.nsplit + split;
    if (e < n_list) load_axis<PHT, PWT, SRT>(cur, axis + (int64_t)s_list[e] * LP, lane);
    for (; e < n_list; e += estep) {
      const int k = s_list[e];
      const int en = e + estep;
      if (en < n_list) load_axis<PHT, PWT, SRT>(nxt, axis + (int64_t)s_list[en] * LP, lane);
      LaneSetup<T, PHT, PWT, SRT> su;
      make_setup<T, PHT, PWT, SRT>(su, cur, r0, H, rowb);
      serve(su, k);
      cur = nxt;
    }
  }
  // ---- overflow of the list (more than kPlaneListCap RoIs on one band of one image): every thread re-derives its own
  // matches in the same order and serves those that did not get a slot — wave by wave, correct and unhurried
  if (n_all > kPlaneListCap && split == 0) {
#pragma unroll 1
    for (int kb = 0; kb < K; kb += KPT * kPlaneThreads) {
#pragma unroll 1
      for (int j = 0; j < KPT; ++j) {
        const int k = key_index<KPT>(kb, j, tid);
        bool mine = false;
        if (kb == 0) mine = (unlisted >> j) & 1u;
        else if (k < K && key[k] == want) {   // later batches: listed iff its index is in the list
          mine = true;
          for (int t = 0; t < kPlaneListCap; ++t) mine = mine && s_list[t] != k;
        }
        unsigned long long bal = __ballot(mine);
        while (bal) {
          const int src_lane = __builtin_ctzll(bal);
          bal &= bal - 1;
          const int kk = __builtin_amdgcn_readlane(k, src_lane);
          AxisEntries<PHT, PWT, SRT> a;
          load_axis<PHT, PWT, SRT>(a, axis + (int64_t)kk * LP, lane);
          LaneSetup<T, PHT, PWT, SRT> su;
          make_setup<T, PHT, PWT, SRT>(su, a, r0, H, rowb);
          serve(su, kk);
        }
      }
    }
  }
}

// Measured on BASELINE config 2 (profiles/r03_roi_align_staging_*.txt): staging the map cuts the HBM fetch
// of the forward 4x (to ~the algorithmic bytes) but the launch is then bound by the lane = bin LDS gather (8-9 LDS cycles
// per ds_read2_b32 under bank conflicts, ~70 cycles per RoI-channel) and by the texture path's ~20 cycles per VMEM
// instruction, and ends up 5-25 % SLOWER than the per-RoI LDS-DMA kernel.  It therefore ships switched OFF; tests force
// it on (tests/test_gpu_roi_plane.py) and tvmi_set_option("roi_align.shared_staging", 1) selects it.
struct PlaneOptions {
  int enabled = 0;
  int min_band_rows = 32;   // a map that does not fit is cut into bands only if a band holds at least this many rows
  int gain_x16 = 32;        // a level is served by staging the map when 16 * (window pixels the RoIs would stage) * gain/16 >= map pixels * overlap
  int whole_planes = 1;     // stage levels whose plane fits the LDS budget (at least twice)
  int band_channels = 2;    // channels per workgroup of a banded level
};
PlaneOptions g_opt;

}  // namespace

PlanePlan make_plane_plan(const MsLevels& lv, int64_t N, int64_t C, int64_t K, int esize, int64_t PH, int64_t PW, int64_t sr) {
  PlanePlan plan;
  plan.N = (int)N;
  plan.total_blocks = 0;
  plan.gain_x16 = g_opt.gain_x16;
  for (int i = 0; i < kMaxLevels; ++i) plan.lv[i] = PlaneLevel{0, 1, 1, 1, 1, 1, 1, 0, 0};
  const bool shape_ok = sr == 2 && ((PH == 7 && PW == 7) || (PH == 14 && PW == 14));
  if (!g_opt.enabled || !shape_ok || (esize != 4 && esize != 2) || N <= 0 || N >= 32768 || C <= 0 || K <= 0) return plan;
  int64_t blocks = 0;
  // coarse levels first: their workgroups (whole planes, several channels) are the long ones, the bands of a fine level the
  // short ones — the launch should end on short workgroups
  for (int i = lv.n_levels - 1; i >= 0; --i) {
    const int64_t H = lv.H[i], W = lv.W[i];
    if (H < 1 || W < 2 || H > 32767 || W > 32767) continue;
    PlaneLevel& pl = plan.lv[i];
    const int64_t whole = 16 + ((H * W * esize + 15) & ~(int64_t)15);
    const int64_t nb = (PH * PW + 63) / 64;
    // The per-RoI set-up (4 table loads per lane) and the texture path's ~20 cycles per VMEM instruction are paid once per
    // (RoI, workgroup): staging the map only pays with at least two channels per workgroup.  A plane that fits twice is
    // staged whole, otherwise two channels of a row band (half-overlapping bands of full-width rows).
    if (2 * whole <= kPlaneImageBytes) {
      if (!g_opt.whole_planes) continue;
      pl.B = pl.S = (int)H;
      pl.nbands = 1;
      pl.cg = (int)std::min<int64_t>(std::min<int64_t>(kPlaneImageBytes / whole, C), 32);
    } else {
      const int64_t cgb = std::min<int64_t>(std::max(1, g_opt.band_channels), C);
      const int64_t rows = std::min<int64_t>((kPlaneImageBytes / cgb - 32) / (W * esize), H);
      if (rows < g_opt.min_band_rows || rows < 4 || g_opt.min_band_rows <= 0) continue;
      pl.B = (int)(rows == H ? H : (rows & ~(int64_t)1));
      pl.S = std::max(1, pl.B / 2);
      pl.nbands = pl.B >= H ? 1 : (int)ceil_div(H - pl.B, pl.S) + 1;
      pl.cg = (int)cgb;
      if (pl.nbands > 4095) continue;
    }
    // a workgroup is bound by ONE CU's LDS and texture path: split a band's RoI list over several workgroups (each stages
    // the band again — cheap, it is L2-resident by then) so that none holds much more than ~192 (RoI, channel, bin-group) units
    {
      const int64_t rois_est = std::max<int64_t>(1, K / (N * std::max<int64_t>(1, lv.n_levels) * pl.nbands));
      pl.nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(32, ceil_div(rois_est * pl.cg * nb, 192)));
    }
    pl.ngroups = (int)ceil_div(C, pl.cg);
    const int64_t nblk = 8 * ceil_div(N * pl.ngroups, 8) * pl.nbands * pl.nsplit;
    if (blocks + nblk > (1ll << 30)) continue;
    pl.enabled = 1;
    pl.block_base = (int)blocks;
    pl.nblocks = (int)nblk;
    blocks += nblk;
  }
  plan.total_blocks = (int)blocks;
  return plan;
}

size_t plane_workspace_bytes(int64_t K, int64_t PH, int64_t PW, int64_t sr) {
  if (!(sr == 2 && ((PH == 7 && PW == 7) || (PH == 14 && PW == 14))) || K <= 0) return 0;
  const int LP = plane_lanes_per_roi((int)PH, (int)PW, (int)sr);

  // [key: K ints, padded to a multiple of 4][blocksum: kPlanePreBlocks * kMaxLevels ints][axis: K * LP float2]
  const size_t b = (((size_t)K + 3) & ~(size_t)3) * 4 + (size_t)kPlanePreBlocks * kMaxLevels * 4;
  return b + (size_t)K * LP * 8;
}

PlaneBuffers plane_buffers(void* ws, int64_t K) {
  PlaneBuffers pb;
  char* p = static_cast<char*>(ws);
  pb.key = reinterpret_cast<int*>(p);
  pb.blocksum = pb.key + (((size_t)K + 3) & ~(size_t)3);
  const size_t b = (((size_t)K + 3) & ~(size_t)3) * 4 + (size_t)kPlanePreBlocks * kMaxLevels * 4;
  pb.axis = reinterpret_cast<float2*>(p + b);
  return pb;
}

namespace {
template <typename T, typename R, int PHT, int PWT, int SRT>
int launch_plane_t(const MsLevels& lv, const PlanePlan& plan, const void* rois, void* output, int64_t C, int64_t K, int aligned,
                   int multiscale, const PlaneBuffers& pb, hipStream_t s) {
  constexpr int LP = plane_lanes_per_roi(PHT, PWT, SRT);
  const int pre_blocks = (int)std::min<int64_t>(kPlanePreBlocks, ceil_div(K * LP, kPlanePreThreads));
  if (pre_blocks < kPlanePreBlocks) {  // unused block sums must read as zero
    const hipError_t e = hipMemsetAsync(pb.blocksum, 0, sizeof(int) * kPlanePreBlocks * kMaxLevels, s);
    if (e != hipSuccess) return set_error((int)e, "roi_align: plane pre-pass memset");
  }
  roi_fwd_prepass<R, PHT, PWT, SRT><<<dim3((unsigned)pre_blocks), dim3(kPlanePreThreads), 0, s>>>(
      lv, plan, static_cast<const R*>(rois), (int)K, aligned, multiscale, pb.key, pb.axis, pb.blocksum);
  auto kern = roi_align_fwd_plane<T, PHT, PWT, SRT>;
  static bool attr_set[64] = {};  // per instantiation and device (the attribute belongs to the device's code object);
  int dev = 0;                    // racing threads set the same value
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kPlaneImageBytes + kPlaneListCap * 4) != hipSuccess)
      return set_error((int)hipErrorInvalidValue, "roi_align: cannot reserve the LDS band");
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  kern<<<dim3((unsigned)plan.total_blocks), dim3(kPlaneThreads), kPlaneImageBytes + kPlaneListCap * 4, s>>>(
      lv, plan, pb.key, pb.axis, pb.blocksum, static_cast<T*>(output), (int)C, (int)K);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_roi_align_forward (shared staging)");
}
}  // namespace

template <typename T, typename R>
int launch_plane(const MsLevels& lv, const PlanePlan& plan, const void* rois, void* output, int64_t C, int64_t K, int64_t PH,
                 int aligned, int multiscale, const PlaneBuffers& pb, hipStream_t s) {
  if (PH == 7) return launch_plane_t<T, R, 7, 7, 2>(lv, plan, rois, output, C, K, aligned, multiscale, pb, s);
  return launch_plane_t<T, R, 14, 14, 2>(lv, plan, rois, output, C, K, aligned, multiscale, pb, s);
}

template int launch_plane<float, float>(const MsLevels&, const PlanePlan&, const void*, void*, int64_t, int64_t, int64_t, int, int,
                                        const PlaneBuffers&, hipStream_t);
template int launch_plane<__half, float>(const MsLevels&, const PlanePlan&, const void*, void*, int64_t, int64_t, int64_t, int, int,
                                         const PlaneBuffers&, hipStream_t);
template int launch_plane<__hip_bfloat16, float>(const MsLevels&, const PlanePlan&, const void*, void*, int64_t, int64_t, int64_t, int,
                                                 int, const PlaneBuffers&, hipStream_t);
template int launch_plane<__half, __half>(const MsLevels&, const PlanePlan&, const void*, void*, int64_t, int64_t, int64_t, int, int,
                                          const PlaneBuffers&, hipStream_t);
template int launch_plane<__hip_bfloat16, __hip_bfloat16>(const MsLevels&, const PlanePlan&, const void*, void*, int64_t, int64_t,
                                                          int64_t, int, int, const PlaneBuffers&, hipStream_t);

int set_plane_option(const char* name, int64_t value) {
  if (!name) return -1;
  const std::string n(name);
  if (n == "roi_align.shared_staging") g_opt.enabled = value != 0;
  else if (n == "roi_align.min_band_rows") g_opt.min_band_rows = (int)value;
  else if (n == "roi_align.stage_whole_planes") g_opt.whole_planes = value != 0;
  else if (n == "roi_align.band_channels") g_opt.band_channels = (int)std::max<int64_t>(1, std::min<int64_t>(value, 32));
  else if (n == "roi_align.staging_gain_x16") g_opt.gain_x16 = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 20));
  else return -1;
  return 0;
}

}  // namespace tvmi
