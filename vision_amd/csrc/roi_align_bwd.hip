// roi_align_bwd.hip — RoIAlign backward for gfx950 (MI355X).
//
// Semantics: torchvision/csrc/ops/cpu/roi_align_kernel.cpp:117-289 (every sample adds
// grad * w_i / count to its 4 taps); the reference GPU kernel does that with 4 atomics per sample
// (cuda/roi_align_kernel.cu:204-332) and is therefore non-deterministic — torchvision/ops/roi_align.py
// :276-281 reroutes to a pure-python implementation under torch.use_deterministic_algorithms(True).
//
// Design here: TILE OWNERSHIP, no atomics, no zero-fill pass, deterministic by construction.
//   * every 16x16-pixel tile of every gradient map (all FPN levels in one launch) x 32 channels is owned
//     by ONE workgroup, which accumulates the contributions of all RoIs whose window overlaps the tile in
//     registers and writes each pixel exactly once with plain stores (zeros where no RoI reaches);
//   * the gradient of a RoI window is a separable product,
//         dWin[r][c] = sum_ph AyD[r][ph] * ( sum_pw G[ph][pw] * AxD[pw][c] ),
//     AyD / AxD being the (tiny) matrices of bilinear row / column factors summed over the samples of a bin
//     and divided by the grid size.  A pre-pass expands them once per RoI into window-relative tables
//     (<= 128 rows / columns; larger windows are evaluated on the fly by a second pass), together with the RoI's level,
//     tile rectangle and window;
//   * lane = (column pair, channel slot): G (the PH*PW grads of the lane's channel) and the AxD columns live
//     in VGPRs, the AyD row of a tile row is WAVE-UNIFORM and is fetched with scalar loads (SGPR operands
//     of packed-fp32 FMAs: two columns per instruction);
//   * the RoI list of a tile is built in the kernel (ballot compaction over the per-RoI descriptors of
//     the tile's image, ascending RoI index), so the summation order is fixed: bit-reproducible.
// Everything this path does not cover (fp64, pooled shapes other than 7x7 / 14x14, maps above 4096 px)
// goes to the atomic kernels at the end of this file (callers zero-fill for those).
#include <type_traits>

#include "roi_common.h"

namespace tvmi {
namespace {

constexpr int kThreads = 256;
constexpr int kTile = 16;      // tile edge (pixels) owned by one workgroup
constexpr int kOwnChunk = 32;  // channels per workgroup: 4 waves x 8 channel slots x 1 channel per lane
constexpr int kRCap = 128;     // window rows / columns with precomputed coefficient rows
constexpr int kAyRows = kTile + kRCap + kTile;  // rows of a RoI's AyD table (zero rows on both sides: any tile row offset is readable)
constexpr int kScanChunk = 1024;  // RoI descriptors scanned per list round (256 per wave)
constexpr int kBigBit = 1 << 30;   // descriptor key bit: window above the table capacity (second pass)
constexpr int kBigCap = 1024;       // oversized-window RoIs listed for the second pass (more: it scans every descriptor)
constexpr int kUnset = 0x7f7f7f7f;  // memset pattern of the per-image RoI ranges and of the oversized-window counter

typedef float v2f __attribute__((ext_vector_type(2)));
// Global-memory vector loads whose addresses are only dword-aligned (a channel's grads start at k*C*PH*PW + c*PH*PW floats,
// a lane's table columns at an odd offset): the types say so — gfx950 serves dword-aligned 8 / 16-byte global
// loads in one instruction either way.
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) F2u { float x, y; };
__device__ __forceinline__ float4 ld4u(const float* p) { const F4u v = *reinterpret_cast<const F4u*>(p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ v2f ld2u(const float* p) { const F2u v = *reinterpret_cast<const F2u*>(p); return v2f{v.x, v.y}; }

// ---------------------------------------------------------------------------------------
// workspace layout (all device memory, written by the pre-pass, read-only for the owner kernel)
struct OwnWorkspace {
  int2* scan;     // [K] {key = (level << 24) | batch (| kBigBit), or -1; tile rect ty0 | ty1<<8 | tx0<<16 | tx1<<24}
  int4* win;      // [K] {y0, x0, wh, ww}
  int* imgrange;  // [2N + 1] {first RoI index of image n, -(last index + 1)} ..., kUnset + count of oversized windows; memset to kUnset
  int* biglist;   // [kBigCap] indices of the RoIs with oversized windows (unordered; the second pass sorts them)
  float* ayt;     // [K][kTile + kRCap + kTile][PH]  AyD rows, window-relative, 16 zero rows before and behind
  float* axt;     // [K][PW][kTile + kRCap + kTile]  AxD rows (column index fastest), window-relative, zero columns on both sides
  float* roisf;   // [K][5] the RoIs as float32 (single-level calls hand 16-bit RoIs over with 16-bit gradients; the owner
                  //        kernels only ever read these)
};

constexpr int pad4(int v) { return (v + 3) & ~3; }

inline size_t own_workspace_bytes(int64_t N, int64_t K, int PH, int PW) {
  const size_t tab = (size_t)K * ((size_t)kAyRows * PH + (size_t)kAyRows * PW) * sizeof(float);
  return (size_t)K * (sizeof(int2) + sizeof(int4) + 5 * sizeof(float)) + (size_t)(2 * N + 4 + kBigCap) * sizeof(int) + tab + 576;
}

inline OwnWorkspace carve_workspace(void* base, int64_t N, int64_t K, int PH, int PW) {
  OwnWorkspace w;
  char* p = static_cast<char*>(base);
  p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 63) & ~uintptr_t(63));
  w.win = reinterpret_cast<int4*>(p);
  p += (size_t)K * sizeof(int4);
  w.scan = reinterpret_cast<int2*>(p);
  p += (size_t)K * sizeof(int2);
  p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 15) & ~uintptr_t(15));
  w.imgrange = reinterpret_cast<int*>(p);
  p += (size_t)(2 * N + 4) * sizeof(int);
  w.biglist = reinterpret_cast<int*>(p);
  p += (size_t)kBigCap * sizeof(int);
  p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 63) & ~uintptr_t(63));
  w.ayt = reinterpret_cast<float*>(p);
  p += (size_t)K * kAyRows * PH * sizeof(float);
  p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 63) & ~uintptr_t(63));
  w.axt = reinterpret_cast<float*>(p);
  p += (size_t)K * kAyRows * PW * sizeof(float);
  p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 63) & ~uintptr_t(63));
  w.roisf = reinterpret_cast<float*>(p);
  return w;
}

// Level geometry of one launch (by value).  Tiles are enumerated level by level, COARSEST LEVEL FIRST
// (its tiles carry the longest RoI lists), image-major inside a level.
struct OwnLevels {
  MsLevels ms;
  int tiles_x[kMaxLevels], tiles_y[kMaxLevels];
  int tile_end[kMaxLevels];  // exclusive prefix end of level (in launch order: level n_levels-1 first)
  int use_ms;                // 0: single map (ms.ptr[0]), every RoI belongs to level 0
  int N;
};

// One axis of AyD / AxD for the map row (column) `pos`: sum over the samples of bin p of the factor that
// lands on `pos`, divided by the grid size (the reference divides every tap by gh*gw).
__device__ __forceinline__ float axis_coef(int dim, float start, float bin, int grid, int p, int pos) {
  float a = 0.f;
  for (int i = 0; i < grid; ++i) {
    int lo, hi;
    float l, h;
    if (axis_sample<float>(dim, start, bin, grid, p, i, lo, hi, l, h)) {
      if (lo == pos) a += h;
      if (hi == pos) a += l;
    }
  }
  return a / (float)grid;
}

// [min lo, max hi] over the valid samples of one axis (wave-cooperative); false when no sample is valid.
__device__ __forceinline__ bool axis_window(int dim, float start, float bin, int grid, int P, int& w0, int& w1) {
  const int lane = threadIdx.x & 63;
  int mn = 0x7fffffff, mx = -1;
  const int n = P * grid;
  for (int s = lane; s < n; s += 64) {
    int lo, hi;
    float l, h;
    if (axis_sample<float>(dim, start, bin, grid, s / grid, s % grid, lo, hi, l, h)) {
      mn = min(mn, lo);
      mx = max(mx, hi);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    mn = min(mn, __shfl_xor(mn, d));
    mx = max(mx, __shfl_xor(mx, d));
  }
  w0 = mn;
  w1 = mx;
  return mx >= 0;
}

// Pre-pass: one wave per RoI.
template <typename RT, int PH, int PW>
__global__ __launch_bounds__(kThreads) void roi_bwd_prepass(const RT* __restrict__ rois, int K, OwnLevels lv, int sr,
                                                            int aligned, OwnWorkspace ws) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (k >= K) return;
  const RT* roi = rois + (int64_t)k * 5;
  if (lane < 5) ws.roisf[(int64_t)k * 5 + lane] = ld(roi + lane);   // what the owner kernels read
  const int l = lv.use_ms ? fpn_level<RT>(roi, lv.ms) : 0;
  const int H = lv.ms.H[l], W = lv.ms.W[l];
  const RoiGeom<float> g = roi_geom<RT, float>(roi, lv.ms.scale[l], PH, PW, sr, aligned != 0);
  int y0 = 0, y1 = -1, x0 = 0, x1 = -1;
  const bool batch_ok = g.batch >= 0 && g.batch < lv.N;
  if (batch_ok && lane == 0) {
    // RoI index range of every image, from the run boundaries of the batch column only: RoI lists are normally
    // grouped by image, and thousands of same-address atomics would cost more than the rest of this kernel
    const int prevb = k > 0 ? (int)ld(rois + (int64_t)(k - 1) * 5) : -1;
    const int nextb = k + 1 < K ? (int)ld(rois + (int64_t)(k + 1) * 5) : -1;
    if (prevb != g.batch) atomicMin(&ws.imgrange[2 * g.batch], k);
    if (nextb != g.batch) atomicMin(&ws.imgrange[2 * g.batch + 1], -(k + 1));
  }
  bool ok = batch_ok && g.gh > 0 && g.gw > 0;
  ok = ok && axis_window(H, g.start_h, g.bin_h, g.gh, PH, y0, y1);
  ok = ok && axis_window(W, g.start_w, g.bin_w, g.gw, PW, x0, x1);
  if (!ok) {  // no sample inside the map (or a batch index outside the tensor): no gradient
    if (lane == 0) {
      ws.scan[k] = make_int2(-1, 0);
      ws.win[k] = make_int4(0, 0, 0, 0);
    }
    return;
  }
  const int wh = y1 - y0 + 1, ww = x1 - x0 + 1;
  const int big = (wh > kRCap || ww > kRCap) ? 1 : 0;
  if (lane == 0) {
    const int rect = (y0 / kTile) | ((y1 / kTile) << 8) | ((x0 / kTile) << 16) | ((x1 / kTile) << 24);
    ws.scan[k] = make_int2((l << 24) | g.batch | (big ? kBigBit : 0), rect);
    ws.win[k] = make_int4(y0, x0, wh, ww);
    if (big) {
      const int slot = atomicAdd(&ws.imgrange[2 * lv.N], 1) - kUnset;
      if (slot >= 0 && slot < kBigCap) ws.biglist[slot] = k;
    }
  }
  if (big) return;
  // table row t <-> window row t - 16; rows past wh + 16 + 15 are never read (the tile overlaps the window)
  for (int t = lane; t < wh + 2 * kTile; t += 64) {
    float* row = ws.ayt + ((int64_t)k * kAyRows + t) * PH;
    const int r = t - kTile;
    const bool in = r >= 0 && r < wh;
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) row[ph] = in ? axis_coef(H, g.start_h, g.bin_h, g.gh, ph, y0 + r) : 0.f;
  }
  // x table: [pw][column], table column t <-> window column t - 16 (zeros outside [0, ww): a lane's two adjacent
  // columns are ONE 8-byte load per pw, without range tests); columns past ww + 16 + 15 are never read
  for (int t = lane; t < ww + 2 * kTile; t += 64) {
    const int c = t - kTile;
    const bool in = c >= 0 && c < ww;
#pragma unroll
    for (int pw = 0; pw < PW; ++pw)
      ws.axt[((int64_t)k * PW + pw) * kAyRows + t] = in ? axis_coef(W, g.start_w, g.bin_w, g.gw, pw, x0 + c) : 0.f;
  }
}

// ---------------------------------------------------------------------------------------
// packed-fp32 FMA with a wave-uniform (SGPR) coefficient: acc.xy += c * t.xy
__device__ __forceinline__ v2f pk_fma_bcast(float c, v2f t, v2f acc) {
  v2f cc = {c, c};
  return __builtin_elementwise_fma(cc, t, acc);
}

// ---- pieces of the owner kernel's inner loop
template <int N>
__device__ __forceinline__ void load_coefs(float (&cf)[N], const float* __restrict__ p) {
#pragma unroll
  for (int q = 0; q < N; ++q) cf[q] = p[q];
}

// 16-bit storage -> float, from raw bits
template <typename GT>
__device__ __forceinline__ float from16(unsigned short h) {
  if constexpr (std::is_same<GT, __half>::value) return __half2float(__ushort_as_half(h));
  else return __uint_as_float((unsigned)h << 16);   // bfloat16 = the upper half of an fp32
}
struct __attribute__((packed, aligned(4))) U4u { unsigned x, y, z, w; };
struct __attribute__((packed, aligned(4))) U2u { unsigned x, y; };

// 16-bit grads of one channel: NG elements starting at gp.  The element offset of a channel's grads is (k*C + c)*PH*PW:
// a multiple of 4 elements for 14x14, whose bin rows are fetched 8 + 6 at a time (112 and 84 elements: every batch starts
// 8-byte aligned -> 8-byte loads); any parity for 7x7 (element loads; only the rare oversized-window pass reads 7x7 grads
// through this routine, the main 7x7 path stages whole 8-channel runs, see issue_run).
template <int NG, typename GT, bool kAligned8>
__device__ __forceinline__ void load_grads16(v2f (&G2)[(NG + 1) / 2], const GT* __restrict__ gp16) {
  const unsigned short* gp = reinterpret_cast<const unsigned short*>(gp16);
  if constexpr (kAligned8) {   // the caller guarantees an 8-byte aligned start and NG % 4 == 0
    static_assert(NG % 4 == 0, "whole 8-byte pieces");
#pragma unroll
    for (int e = 0; e + 4 <= NG; e += 4) {
      const U2u v = *reinterpret_cast<const U2u*>(gp + e);
      G2[e / 2] = v2f{from16<GT>((unsigned short)(v.x & 0xffffu)), from16<GT>((unsigned short)(v.x >> 16))};
      G2[e / 2 + 1] = v2f{from16<GT>((unsigned short)(v.y & 0xffffu)), from16<GT>((unsigned short)(v.y >> 16))};
    }
  } else {
#pragma unroll
    for (int e = 0; e + 2 <= NG; e += 2) G2[e / 2] = v2f{from16<GT>(gp[e]), from16<GT>(gp[e + 1])};
    if constexpr (NG & 1) G2[NG / 2] = v2f{from16<GT>(gp[NG - 1]), 0.f};
  }
}

// the PH*PW grads of one channel as pairs (the packed FMAs pick the low / high half by op_sel)
template <int NG>
__device__ __forceinline__ void load_grads(v2f (&G2)[(NG + 1) / 2], const float* __restrict__ gp) {
  constexpr int NG2 = (NG + 1) / 2;
#pragma unroll
  for (int e4 = 0; e4 + 4 <= NG; e4 += 4) {
    const float4 v = ld4u(gp + e4);
    G2[e4 / 2] = v2f{v.x, v.y};
    G2[e4 / 2 + 1] = v2f{v.z, v.w};
  }
  if constexpr ((NG & 3) == 1) G2[NG2 - 1] = v2f{gp[NG - 1], 0.f};
  if constexpr ((NG & 3) == 2) G2[NG2 - 1] = ld2u(gp + NG - 2);
  if constexpr ((NG & 3) == 3) {
    G2[NG2 - 2] = ld2u(gp + NG - 3);
    G2[NG2 - 1] = v2f{gp[NG - 1], 0.f};
  }
}

// t[ph] = sum_pw G[ph][pw] * AxD[pw][two columns of this lane] for the NPH bin rows whose grads are in G2
template <int NPH, int PW>
__device__ __forceinline__ void contract_x(v2f* t, const v2f (&G2)[(NPH * PW + 1) / 2], const v2f (&axd)[PW], bool ch_ok) {
#pragma unroll
  for (int ph = 0; ph < NPH; ++ph) {
    v2f a = {0.f, 0.f};
#pragma unroll
    for (int pw = 0; pw < PW; ++pw) {
      const int e1 = ph * PW + pw;
      const v2f gg = G2[e1 / 2];
      const v2f gs = (e1 & 1) ? __builtin_shufflevector(gg, gg, 1, 1) : __builtin_shufflevector(gg, gg, 0, 0);
      a = __builtin_elementwise_fma(gs, axd[pw], a);
    }
    t[ph] = ch_ok ? a : v2f{0.f, 0.f};
  }
}

// all PH bin rows, the grads fetched GB rows at a time (14x14: two batches of 7 rows, so that the grads of a
// channel never occupy more than ~50 VGPRs)
template <int PH, int PW, typename GT>
__device__ __forceinline__ void grads_times_axd(v2f (&t)[PH], const GT* __restrict__ gp, const v2f (&axd)[PW], bool ch_ok) {
  if constexpr (std::is_same<GT, float>::value) {
    constexpr int GB = PH <= 7 ? PH : 7;
    static_assert(PH % GB == 0, "bin rows must split evenly into grad batches");
#pragma unroll
    for (int p0 = 0; p0 < PH; p0 += GB) {
      v2f G2[(GB * PW + 1) / 2];
      load_grads<GB * PW>(G2, gp + p0 * PW);
      contract_x<GB, PW>(t + p0, G2, axd, ch_ok);
      if (p0 + GB < PH) __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (PH == 14 && PW == 14) {
    {
      v2f G2[(8 * PW) / 2];
      load_grads16<8 * PW, GT, true>(G2, gp);
      contract_x<8, PW>(t, G2, axd, ch_ok);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      v2f G2[(6 * PW) / 2];
      load_grads16<6 * PW, GT, true>(G2, gp + 8 * PW);
      contract_x<6, PW>(t + 8, G2, axd, ch_ok);
    }
  } else {
    v2f G2[(PH * PW + 1) / 2];
    load_grads16<PH * PW, GT, false>(G2, gp);
    contract_x<PH, PW>(t, G2, axd, ch_ok);
  }
}

template <int RB, int PH>
__device__ __forceinline__ void rows_fma(v2f* acc, const float (&cf)[RB * PH], const v2f (&t)[PH]) {
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) acc[r] = pk_fma_bcast(cf[r * PH + ph], t[ph], acc[r]);
}

// One entry of a tile's RoI list, copied into LDS by the scanning lane: everything the accumulation needs
// that is per-RoI (so that the walk over the list has ONE dependent LDS read in front of its loads).
struct OwnEntry {
  int k, y0, x0, wh;
};

// gstage: per wave, the grads of its 8 channels for one RoI as fp32 (8 * PH*PW floats: 392 for 7x7, 1568 for 14x14).  7x7 keeps
// two buffers; 14x14 one (the wave writes entry e + 1 only after it has consumed entry e — program order, in-order LDS).
template <int PH, int PW>
struct OwnShared {
  static constexpr int kRun = (8 * PH * PW + 15) & ~15;
  static constexpr int kBufs = PH <= 7 ? 2 : 1;
  OwnEntry list[4][kScanChunk / 4];  // per-wave segments of the tile's RoI list (ascending RoI index)
  int count[4];
  __attribute__((aligned(16))) float gstage[4][kBufs][kRun];
};

#define TVMI_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)

// Owner work item: (tile, 32-channel chunk).  item = tile * nchunks + chunk, so that with 8 chunks (C = 256) every
// XCD (workgroup index % 8) serves ONE channel chunk of all tiles and the grads of a (RoI, chunk) are fetched into
// that XCD's L2 once for the ~5 tiles that need them.
// grad is [K, C, PH, PW] with contiguous bins (element strides ns, cs for RoI and channel).
// kBig = false: RoIs whose window fits the coefficient tables; the tile is WRITTEN (zeros where nothing reaches).
// kBig = true : the RoIs with larger windows (rare); factors are evaluated here and the tile is read-add-written —
//               still one owner per pixel, still a fixed order (this pass runs after the first one).
template <typename GT, bool kBig, int PH, int PW>
__device__ __forceinline__ void owner_item(OwnShared<PH, PW>& sh, int item, const GT* __restrict__ grad, const float* __restrict__ rois,
                                           const OwnLevels& lv, int C, int K, int nchunks, int sr, int aligned, int64_t ns,
                                           int64_t cs, const OwnWorkspace& ws, const int* klist = nullptr, int nlist = 0) {
  constexpr int RB = PH <= 7 ? 4 : 2;  // tile rows per scalar-load batch (RB * PH coefficient SGPRs, two batches in flight)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = TVMI_UNIFORM(tid >> 6);
  const int chunk = item % nchunks;
  int tile = item / nchunks;
  // ---- which tile: levels in launch order (coarsest first)
  int l = 0, tbase = 0;
  for (int i = lv.ms.n_levels - 1; i >= 0; --i) {
    if (tile < lv.tile_end[i]) {
      l = i;
      break;
    }
    tbase = lv.tile_end[i];
  }
  tile -= tbase;
  const int H = lv.ms.H[l], W = lv.ms.W[l];
  const int txn = lv.tiles_x[l], tyn = lv.tiles_y[l];
  const int n = tile / (txn * tyn);
  const int trem = tile - n * (txn * tyn);
  const int ty = trem / txn, tx = trem - ty * txn;
  const int ybase = ty * kTile, xbase = tx * kTile;
  const int key = (l << 24) | n | (kBig ? kBigBit : 0);
  // ---- lane roles
  const int cp = lane & 7, cslot = lane >> 3;
  const int ch = chunk * kOwnChunk + wave * 8 + cslot;
  const bool ch_ok = ch < C;
  const int chc = ch_ok ? ch : C - 1;
  const int xl = xbase + 2 * cp;  // this lane's two columns: xl, xl + 1
  v2f acc[kTile];
#pragma unroll
  for (int r = 0; r < kTile; ++r) acc[r] = v2f{0.f, 0.f};
  int touched = 0;

  int rs = ws.imgrange[2 * n], re = -ws.imgrange[2 * n + 1];
  if (re <= 0 || rs > K) {  // no RoI on this image
    rs = 0;
    re = 0;
  }
  if (klist) {  // scan domain = positions of a sorted RoI index list instead of the image's index range
    rs = 0;
    re = nlist;
  }
  rs = TVMI_UNIFORM(rs);
  re = TVMI_UNIFORM(re);

  for (int base = rs; base < re; base += kScanChunk) {
    // ---- scan: wave w looks at descriptors [base + 256 w, base + 256 (w + 1)), 64 at a time, and appends
    // the hits to its own list segment (ascending index)
    // Both tables are read UNCONDITIONALLY (positions past the range clamped to its last descriptor, the window of a RoI that
    // does not hit read anyway — 16 KB per image, L2-resident) and all four 64-descriptor groups are in flight together: written
    // as `if (pos < re) load scan; if (hit) load win` per group, the ISA was eight dependent round trips — load, s_waitcnt
    // vmcnt(0), conditional load, s_waitcnt vmcnt(0), four times — in front of every workgroup's first FMA.
    constexpr int NG = kScanChunk / 256;
    int cnt = 0;
    int kk[NG];
    int2 dsc[NG];
    int4 wnd[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int pos = min(base + wave * (kScanChunk / 4) + j * 64 + lane, re - 1);
      kk[j] = klist ? klist[pos] : pos;
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) dsc[j] = ws.scan[kk[j]];
#pragma unroll
    for (int j = 0; j < NG; ++j) wnd[j] = ws.win[kk[j]];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int pos = base + wave * (kScanChunk / 4) + j * 64 + lane;
      const int r = dsc[j].y;
      const bool hit = pos < re && dsc[j].x == key && ty >= (r & 255) && ty <= ((r >> 8) & 255) && tx >= ((r >> 16) & 255) &&
                       tx <= ((r >> 24) & 255);
      const unsigned long long b = __ballot(hit);
      if (hit) {
        OwnEntry e;
        e.k = kk[j];
        e.y0 = wnd[j].x;
        e.x0 = wnd[j].y;
        e.wh = wnd[j].z;
        sh.list[wave][cnt + __builtin_popcountll(b & ((1ull << lane) - 1ull))] = e;
      }
      cnt += __builtin_popcountll(b);
    }
    if (lane == 0) sh.count[wave] = cnt;
    __syncthreads();
    // ---- accumulate the listed RoIs (every wave walks all four segments, in order)
    if constexpr (!kBig) {
      {
        // The grads of the wave's 8 channels are ONE contiguous run of 8 * PH*PW elements: the wave fetches it with coalesced
        // 16-byte-per-lane loads into its private LDS region (as fp32, whatever the storage type) and every lane reads its
        // channel from there row by row — instead of per-lane loads whose 64 lanes ask for 8 scattered 16-byte pieces each
        // (104 line requests per RoI and wave at the texture addresser for 7x7) and, for 14x14, instead of holding 98 grads
        // per lane in registers (194 VGPRs, two waves per SIMD).  The loads of entry e + 1 are issued as soon as entry e
        // sits in LDS, so their latency runs under the packed FMAs of entry e.
        constexpr int NGW = 8 * PH * PW;                            // elements per wave and RoI
        constexpr int EPP = std::is_same<GT, float>::value ? 4 : 8; // elements per 16-byte piece
        constexpr int NPIECE = NGW / EPP;                           // 98 / 392 (fp32), 49 / 196 (16-bit)
        constexpr int NLD = (NPIECE + 63) / 64;                     // load instructions per lane
        constexpr int RUN = OwnShared<PH, PW>::kRun, BUFS = OwnShared<PH, PW>::kBufs;
        static_assert(NGW % 8 == 0 && NGW <= RUN, "whole 16-byte pieces, run fits the staging buffer");
        const int n0 = TVMI_UNIFORM(sh.count[0]), n1 = n0 + TVMI_UNIFORM(sh.count[1]), n2 = n1 + TVMI_UNIFORM(sh.count[2]);
        const int total = n2 + TVMI_UNIFORM(sh.count[3]);
        touched += total;
        const int ch0w = chunk * kOwnChunk + wave * 8;            // first channel of this wave
        const int nvalid = max(0, min(8, C - ch0w)) * PH * PW;    // elements of the run that exist
        float* gl = &sh.gstage[wave][0][0];
        auto entry_at = [&](int e, int& k, int& y0, int& x0, int& wh) {
          const int seg = (e >= n0 ? 1 : 0) + (e >= n1 ? 1 : 0) + (e >= n2 ? 1 : 0);
          const int idx = e - (seg == 0 ? 0 : seg == 1 ? n0 : seg == 2 ? n1 : n2);
          const OwnEntry en = sh.list[seg][idx];
          k = TVMI_UNIFORM(en.k);
          y0 = TVMI_UNIFORM(en.y0);
          x0 = TVMI_UNIFORM(en.x0);
          wh = TVMI_UNIFORM(en.wh);
        };
        U4u gin[NLD];   // raw 16-byte pieces (4 floats, or 8 16-bit elements; C is even on the 16-bit path: dword-aligned runs)
        const GT* run_held = grad;   // the run `gin` was loaded from (stage_run patches the straddling piece from it)
        // EVERY lane loads a whole 16-byte piece, unconditionally: its own when that lies inside the run, piece 0 of the run
        // otherwise (zeroed / patched in stage_run, where the values are used anyway).  Written as `if (inside) v = load; else
        // if (straddles) {...}` the loaded value was one input of a phi, and the copy that resolves it put an s_waitcnt
        // vmcnt(0) right behind the load (ISA) — the prefetch of entry e + 1 never ran under the FMAs of entry e.
        auto issue_run = [&](int k) {
          const GT* run = grad + (int64_t)k * ns + (int64_t)(nvalid > 0 ? ch0w : 0) * cs;
          run_held = run;
#pragma unroll
          for (int u = 0; u < NLD; ++u) {
            const int f0 = EPP * (u * 64 + lane);  // first element of this lane's piece
            gin[u] = *reinterpret_cast<const U4u*>(run + (f0 + EPP <= nvalid ? f0 : 0));
          }
        };
        auto straddling_piece = [&](int f0) {   // the piece that straddles the end of the run (never read past the tensor); rare
          const GT* run = run_held;
          U4u v{0u, 0u, 0u, 0u};
          if constexpr (std::is_same<GT, float>::value) {
            v.x = __float_as_uint(run[f0]);
            if (f0 + 1 < nvalid) v.y = __float_as_uint(run[f0 + 1]);
            if (f0 + 2 < nvalid) v.z = __float_as_uint(run[f0 + 2]);
          } else {
            const unsigned short* r16 = reinterpret_cast<const unsigned short*>(run);
            unsigned short e[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) e[q] = f0 + q < nvalid ? r16[f0 + q] : (unsigned short)0;
            v.x = e[0] | ((unsigned)e[1] << 16);
            v.y = e[2] | ((unsigned)e[3] << 16);
            v.z = e[4] | ((unsigned)e[5] << 16);
            v.w = e[6] | ((unsigned)e[7] << 16);
          }
          return v;
        };
        auto stage_run = [&](float* gb) {   // registers -> this wave's LDS run, as fp32
#pragma unroll
          for (int u = 0; u < NLD; ++u) {
            const int piece = u * 64 + lane;
            if (piece < NPIECE) {
              const int f0 = EPP * piece;
              U4u v = gin[u];
              if (f0 + EPP > nvalid) {   // not a whole piece of the run: nothing, or the straddling one
                v = U4u{0u, 0u, 0u, 0u};
                if ((nvalid % EPP) != 0 && f0 < nvalid) v = straddling_piece(f0);   // (first test wave-uniform, false for whole 8-channel runs)
              }
              if constexpr (std::is_same<GT, float>::value) {
                *reinterpret_cast<float4*>(gb + 4 * piece) =
                    make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
              } else {
                *reinterpret_cast<float4*>(gb + 8 * piece) =
                    make_float4(from16<GT>((unsigned short)(v.x & 0xffffu)), from16<GT>((unsigned short)(v.x >> 16)),
                                from16<GT>((unsigned short)(v.y & 0xffffu)), from16<GT>((unsigned short)(v.y >> 16)));
                *reinterpret_cast<float4*>(gb + 8 * piece + 4) =
                    make_float4(from16<GT>((unsigned short)(v.z & 0xffffu)), from16<GT>((unsigned short)(v.z >> 16)),
                                from16<GT>((unsigned short)(v.w & 0xffffu)), from16<GT>((unsigned short)(v.w >> 16)));
              }
            }
          }
        };
        int k = 0, y0 = 0, x0 = 0, wh = 0;
        if (total > 0) {
          entry_at(0, k, y0, x0, wh);
          issue_run(k);
        }
        for (int e = 0; e < total; ++e) {
          float* gb = gl + (BUFS == 2 ? (e & 1) : 0) * RUN;
          stage_run(gb);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          // -- coefficient rows (scalar loads), AxD columns (vector loads), grads of this lane's channel (LDS)
          const float* arow = ws.ayt + ((int64_t)k * kAyRows + (ybase - y0 + kTile)) * PH;
          float cfa[RB * PH], cfb[RB * PH];
          load_coefs<RB * PH>(cfa, arow);
          load_coefs<RB * PH>(cfb, arow + RB * PH);
          // AxD of this lane's two adjacent pixels: one 8-byte load per pw from the zero-padded [pw][column] table
          v2f axd[PW];
          const float* xt = ws.axt + (int64_t)k * PW * kAyRows + (xl - x0 + kTile);
#pragma unroll
          for (int pw = 0; pw < PW; ++pw) axd[pw] = ld2u(xt + pw * kAyRows);
          // the next entry's grads AFTER this entry's AxD loads: vector loads return in order, so the wait for AxD (needed at
          // once) would otherwise also be a wait for the prefetch issued in front of it
          // ... and UNCONDITIONALLY (the last entry re-loads itself): behind an `if (e + 1 < total)` the compiler's counted waits
          // must assume the loads were not issued, and the last of them becomes a vmcnt(0) that includes the prefetch
          int kn = 0, y0n = 0, x0n = 0, whn = 0;
          entry_at(min(e + 1, total - 1), kn, y0n, x0n, whn);
          issue_run(kn);
          // grads row by row out of LDS, one row ahead of the FMAs that consume it (14 VGPRs instead of 49)
          const float* gch = gb + cslot * (PH * PW);
          v2f t[PH];
          float grow[2][PW];
#pragma unroll
          for (int pw = 0; pw < PW; ++pw) grow[0][pw] = gch[pw];
#pragma unroll
          for (int ph = 0; ph < PH; ++ph) {
            if (ph + 1 < PH) {
#pragma unroll
              for (int pw = 0; pw < PW; ++pw) grow[(ph + 1) & 1][pw] = gch[(ph + 1) * PW + pw];
            }
            __builtin_amdgcn_sched_barrier(0);
            v2f a = {0.f, 0.f};
#pragma unroll
            for (int pw = 0; pw < PW; ++pw) a = pk_fma_bcast(grow[ph & 1][pw], axd[pw], a);
            t[ph] = a;  // lanes of channels past C carry zeros (staged run) and are never stored
            __builtin_amdgcn_sched_barrier(0);
          }
          // groups of RB tile rows that lie entirely outside the window have all-zero coefficient rows: their FMAs are
          // skipped (uniform branch); the scalar loads stay unconditional so that they remain two batches ahead
          const int rrel0 = ybase - y0;
#pragma unroll
          for (int r0 = 0; r0 < kTile; r0 += 2 * RB) {
            if (rrel0 + r0 + RB > 0 && rrel0 + r0 < wh) rows_fma<RB, PH>(acc + r0, cfa, t);
            if (r0 + 2 * RB < kTile) load_coefs<RB * PH>(cfa, arow + (r0 + 2 * RB) * PH);
            __builtin_amdgcn_sched_barrier(0);
            if (rrel0 + r0 + 2 * RB > 0 && rrel0 + r0 + RB < wh) rows_fma<RB, PH>(acc + r0 + RB, cfb, t);
            if (r0 + 3 * RB < kTile) load_coefs<RB * PH>(cfb, arow + (r0 + 3 * RB) * PH);
            __builtin_amdgcn_sched_barrier(0);
          }
          k = kn;
          y0 = y0n;
          x0 = x0n;
          wh = whn;
        }
        __syncthreads();
        continue;
      }
    }
    for (int seg = 0; seg < 4; ++seg) {
      const int nseg = TVMI_UNIFORM(sh.count[seg]);
      touched += nseg;
      for (int i = 0; i < nseg; ++i) {
        const OwnEntry e = sh.list[seg][i];
        const int k = TVMI_UNIFORM(e.k), y0 = TVMI_UNIFORM(e.y0), x0 = TVMI_UNIFORM(e.x0), wh = TVMI_UNIFORM(e.wh);
        const GT* gp = grad + (int64_t)k * ns + (int64_t)chc * cs;
        if constexpr (!kBig) {
          // tile row r <-> table row (ybase - y0) + 16 + r: one base address per RoI, immediate offsets per row.
          // The AyD rows are wave-uniform: scalar loads, RB rows per batch, two batches in flight.
          const float* arow = ws.ayt + ((int64_t)k * kAyRows + (ybase - y0 + kTile)) * PH;
          float cfa[RB * PH], cfb[RB * PH];
          load_coefs<RB * PH>(cfa, arow);
          load_coefs<RB * PH>(cfb, arow + RB * PH);
          // -- AxD columns of this lane's two pixels (table rows; zero outside the window)
          // AxD of this lane's two adjacent pixels: one 8-byte load per pw from the zero-padded [pw][column] table
          v2f axd[PW];
          const float* xt = ws.axt + (int64_t)k * PW * kAyRows + (xl - x0 + kTile);
#pragma unroll
          for (int pw = 0; pw < PW; ++pw) axd[pw] = ld2u(xt + pw * kAyRows);
          v2f t[PH];
          grads_times_axd<PH, PW, GT>(t, gp, axd, ch_ok);
          // -- acc[r] += sum_ph AyD[r][ph] * t[ph]
          // groups of RB tile rows that lie entirely outside the window have all-zero coefficient rows: their FMAs are
          // skipped (uniform branch); the scalar loads stay unconditional so that they remain two batches ahead
          const int rrel0 = ybase - y0;
#pragma unroll
          for (int r0 = 0; r0 < kTile; r0 += 2 * RB) {
            if (rrel0 + r0 + RB > 0 && rrel0 + r0 < wh) rows_fma<RB, PH>(acc + r0, cfa, t);
            if (r0 + 2 * RB < kTile) load_coefs<RB * PH>(cfa, arow + (r0 + 2 * RB) * PH);
            __builtin_amdgcn_sched_barrier(0);
            if (rrel0 + r0 + 2 * RB > 0 && rrel0 + r0 + RB < wh) rows_fma<RB, PH>(acc + r0 + RB, cfb, t);
            if (r0 + 3 * RB < kTile) load_coefs<RB * PH>(cfb, arow + (r0 + 3 * RB) * PH);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          // window above 64 rows / columns: evaluate the factors here (same arithmetic as the pre-pass)
          const float* roi = rois + (int64_t)k * 5;
          const RoiGeom<float> g = roi_geom<float, float>(roi, lv.ms.scale[l], PH, PW, sr, aligned != 0);
          v2f axd[PW];
#pragma unroll
          for (int pw = 0; pw < PW; ++pw)
            axd[pw] = v2f{axis_coef(W, g.start_w, g.bin_w, g.gw, pw, xl), axis_coef(W, g.start_w, g.bin_w, g.gw, pw, xl + 1)};
          v2f t[PH];
          grads_times_axd<PH, PW, GT>(t, gp, axd, ch_ok);
          float ayv[PH];  // AyD row of tile row `lane` (lanes 0..15)
#pragma unroll
          for (int ph = 0; ph < PH; ++ph) ayv[ph] = axis_coef(H, g.start_h, g.bin_h, g.gh, ph, ybase + (lane & 15));
#pragma unroll
          for (int r = 0; r < kTile; ++r) {
#pragma unroll
            for (int ph = 0; ph < PH; ++ph)
              acc[r] = pk_fma_bcast(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ayv[ph]), r)),
                                    t[ph], acc[r]);
          }
        }
      }
    }
    __syncthreads();
  }
  if (kBig && touched == 0) return;  // nothing to add to this tile
  // ---- every pixel of the tile is written exactly once per pass
  if (ch_ok && xl < W) {
    GT* plane = static_cast<GT*>(const_cast<void*>(lv.ms.ptr[l])) + ((int64_t)n * C + ch) * H * W;
    const bool both = xl + 1 < W;
#pragma unroll
    for (int r = 0; r < kTile; ++r) {
      const int y = ybase + r;
      if (y < H) {
        GT* p = plane + (int64_t)y * W + xl;
        v2f v = acc[r];
        if constexpr (kBig) {
          v.x += ld(p);
          if (both) v.y += ld(p + 1);
        }
        if constexpr (std::is_same<GT, float>::value) {
          if (both)
            *reinterpret_cast<v2f*>(p) = v;
          else
            p[0] = v.x;
        } else {   // fp32 sums rounded to the 16-bit type ONCE (the oversized-window pass, if any, rounds a second time)
          st(p, v.x);
          if (both) st(p + 1, v.y);
        }
      }
    }
  }
}

// 14x14: at most 168 VGPRs, so that three waves share a SIMD (the LDS-staged grads brought it down from 194)
template <typename GT, int PH, int PW>
__global__ __launch_bounds__(kThreads, (PH <= 7 ? 1 : 3)) void roi_align_bwd_owner(const GT* __restrict__ grad, OwnLevels lv, int C, int K,
                                                                int nchunks, int sr, int aligned, int64_t ns, int64_t cs,
                                                                OwnWorkspace ws) {
  __shared__ OwnShared<PH, PW> sh;
  owner_item<GT, false, PH, PW>(sh, (int)blockIdx.x, grad, ws.roisf, lv, C, K, nchunks, sr, aligned, ns, cs, ws);
}

// second pass: only does anything when the pre-pass counted RoIs with oversized windows
template <typename GT, int PH, int PW>
__global__ __launch_bounds__(kThreads) void roi_align_bwd_owner_big(const GT* __restrict__ grad, OwnLevels lv, int C, int K,
                                                                    int nchunks, int nitems, int sr, int aligned, int64_t ns,
                                                                    int64_t cs, OwnWorkspace ws) {
  __shared__ OwnShared<PH, PW> sh;
  __shared__ int s_raw[kBigCap], s_sorted[kBigCap];
  const int nbig = ws.imgrange[2 * lv.N] - kUnset;
  if (nbig <= 0) return;  // no oversized window anywhere
  const bool listed = nbig <= kBigCap;
  if (listed) {  // ascending RoI index = the fixed summation order: rank sort of the (short) list
    for (int i = threadIdx.x; i < nbig; i += kThreads) s_raw[i] = ws.biglist[i];
    __syncthreads();
    for (int i = threadIdx.x; i < nbig; i += kThreads) {
      const int v = s_raw[i];
      int rank = 0;
      for (int j = 0; j < nbig; ++j) rank += s_raw[j] < v ? 1 : 0;
      s_sorted[rank] = v;
    }
    __syncthreads();
  }
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    owner_item<GT, true, PH, PW>(sh, item, grad, ws.roisf, lv, C, K, nchunks, sr, aligned, ns, cs, ws, listed ? s_sorted : nullptr, nbig);
    __syncthreads();
  }
}

// =======================================================================================
// Atomic fallback (fp64, other pooled shapes, non-contiguous bins, maps above 4096 px): grad_input is
// zero-filled by the caller and accumulated with hardware float atomics, like the reference GPU kernel.
constexpr int kMaxTab = 128;      // max PH*gh (and PW*gw) samples per axis kept in LDS
constexpr int kWinFloats = 8192;  // LDS window capacity in floats (32 KiB)
constexpr int kChunk = 32;        // channels per workgroup

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
template <typename T>
__global__ __launch_bounds__(kThreads) void roi_align_bwd_tile(
    const T* __restrict__ grad, const T* __restrict__ rois, T* __restrict__ grad_input, int C, int H,
    int W, int PH, int PW, float spatial_scale, int sr, int aligned, int nchunks, int64_t ns,
    int64_t cs, int64_t hs, int64_t ws, MsLevels lv, int use_ms) {
  __shared__ TileShared s;
  const int PHW = PH * PW;
  const int tid = threadIdx.x;
  const int k = blockIdx.x / nchunks;
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


bool owner_shape(int64_t PH, int64_t PW) { return (PH == 7 && PW == 7) || (PH == 14 && PW == 14); }

// Does a backward call with these arguments take the tile-owner path (grad_input fully overwritten,
// deterministic)?  The launchers and the `*_overwrites` queries share this predicate.
bool owner_applies(tvmi_dtype dt, int64_t N, int64_t C, int64_t K, int64_t PH, int64_t PW, const int64_t* heights,
                   const int64_t* widths, int64_t n_levels, int64_t cs, int64_t hs, int64_t ws, size_t workspace_bytes) {
  if (!(dt == TVMI_F32 || dt == TVMI_F16 || dt == TVMI_BF16) || !owner_shape(PH, PW) || N <= 0 || N >= (1 << 24) || C <= 0 || K < 0 ||
      K >= (1ll << 30))
    return false;
  if (dt != TVMI_F32 && (C & 1)) return false;  // 16-bit grads are fetched in 16-byte pieces that must start dword-aligned
  if (ws != 1 || hs != PW || cs != PH * PW) return false;  // the [C, PH, PW] block of a RoI must be contiguous
  if (n_levels < 1 || n_levels > kMaxLevels) return false;
  int64_t tiles = 0;
  for (int64_t i = 0; i < n_levels; ++i) {
    if (heights[i] <= 0 || widths[i] <= 0 || heights[i] > 4096 || widths[i] > 4096) return false;
    tiles += N * ceil_div(heights[i], kTile) * ceil_div(widths[i], kTile);
  }
  if (tiles * ceil_div(C, kOwnChunk) >= (1ll << 31)) return false;
  return workspace_bytes >= own_workspace_bytes(N, K, (int)PH, (int)PW);
}

template <typename GT, typename RT, int PH, int PW>
int launch_owner(const GT* grad, const RT* rois, OwnLevels lv, int64_t N, int64_t C, int64_t K, int sr, int aligned,
                 int64_t ns, int64_t cs, void* workspace, hipStream_t stream) {
  const OwnWorkspace w = carve_workspace(workspace, N, K, PH, PW);
  hipError_t e = hipMemsetAsync(w.imgrange, 0x7f, (size_t)(2 * N + 1) * sizeof(int), stream);
  if (e != hipSuccess) return set_error((int)e, "roi_align_backward: memset");
  if (K > 0)
    roi_bwd_prepass<RT, PH, PW><<<dim3((unsigned)ceil_div(K, kThreads / 64)), dim3(kThreads), 0, stream>>>(rois, (int)K, lv, sr, aligned, w);
  const int nchunks = (int)ceil_div(C, kOwnChunk);
  const int64_t tiles = lv.tile_end[0];
  const int nitems = (int)(tiles * nchunks);
  roi_align_bwd_owner<GT, PH, PW><<<dim3((unsigned)nitems), dim3(kThreads), 0, stream>>>(grad, lv, (int)C, (int)K, nchunks, sr,
                                                                                         aligned, ns, cs, w);
  if (K > 0)
    roi_align_bwd_owner_big<GT, PH, PW><<<dim3((unsigned)std::min(nitems, 2048)), dim3(kThreads), 0, stream>>>(
        grad, lv, (int)C, (int)K, nchunks, nitems, sr, aligned, ns, cs, w);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_roi_align_backward");
}

OwnLevels make_levels(const MsLevels& ms, int64_t N, int use_ms) {
  OwnLevels lv{};
  lv.ms = ms;
  lv.use_ms = use_ms;
  lv.N = (int)N;
  int acc = 0;
  for (int i = ms.n_levels - 1; i >= 0; --i) {  // launch order: coarsest level first
    lv.tiles_x[i] = (int)ceil_div(ms.W[i], kTile);
    lv.tiles_y[i] = (int)ceil_div(ms.H[i], kTile);
    acc += (int)N * lv.tiles_x[i] * lv.tiles_y[i];
    lv.tile_end[i] = acc;
  }
  return lv;
}

// grads (and the maps written) are `dt`; the RoIs are `dt` too in the single-level call and float32 in the multi-scale one
template <typename GT, typename RT>
int dispatch_owner_t(const void* grad, const void* rois, const OwnLevels& lv, int64_t N, int64_t C, int64_t K, int64_t PH,
                     int64_t sr, int aligned, int64_t ns, int64_t cs, void* workspace, hipStream_t stream) {
  const GT* g = static_cast<const GT*>(grad);
  const RT* r = static_cast<const RT*>(rois);
  if (PH == 7) return launch_owner<GT, RT, 7, 7>(g, r, lv, N, C, K, (int)sr, aligned, ns, cs, workspace, stream);
  return launch_owner<GT, RT, 14, 14>(g, r, lv, N, C, K, (int)sr, aligned, ns, cs, workspace, stream);
}

int dispatch_owner(tvmi_dtype dt, bool rois_f32, const void* grad, const void* rois, const MsLevels& ms, int use_ms, int64_t N,
                   int64_t C, int64_t K, int64_t PH, int64_t PW, int64_t sr, int aligned, int64_t ns, int64_t cs, void* workspace,
                   hipStream_t stream) {
  const OwnLevels lv = make_levels(ms, N, use_ms);
  if (dt == TVMI_F32) return dispatch_owner_t<float, float>(grad, rois, lv, N, C, K, PH, sr, aligned, ns, cs, workspace, stream);
  if (dt == TVMI_F16)
    return rois_f32 ? dispatch_owner_t<__half, float>(grad, rois, lv, N, C, K, PH, sr, aligned, ns, cs, workspace, stream)
                    : dispatch_owner_t<__half, __half>(grad, rois, lv, N, C, K, PH, sr, aligned, ns, cs, workspace, stream);
  return rois_f32 ? dispatch_owner_t<__hip_bfloat16, float>(grad, rois, lv, N, C, K, PH, sr, aligned, ns, cs, workspace, stream)
                  : dispatch_owner_t<__hip_bfloat16, __hip_bfloat16>(grad, rois, lv, N, C, K, PH, sr, aligned, ns, cs, workspace, stream);
}

template <typename T>
int launch_atomic(const void* grad, const void* rois, void* grad_input, int64_t C, int64_t H, int64_t W, int64_t K,
                  int64_t PH, int64_t PW, double scale, int64_t sr, int aligned, int64_t ns, int64_t cs, int64_t hs,
                  int64_t ws, hipStream_t stream, const MsLevels* ms) {
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
    roi_align_bwd_generic<T><<<dim3((unsigned)blocks), dim3(kThreads), 0, stream>>>(g, r, gi, total, (int)C, (int)H, (int)W, (int)PH,
                                                                                   (int)PW, scale, (int)sr, aligned, ns, cs, hs, ws);
  } else {
    const int nchunks = (int)ceil_div(C, kChunk);
    const dim3 grid((unsigned)(K * nchunks)), block(kThreads);
    roi_align_bwd_tile<T><<<grid, block, 0, stream>>>(g, r, gi, (int)C, (int)H, (int)W, (int)PH, (int)PW, (float)scale, (int)sr,
                                                      aligned, nchunks, ns, cs, hs, ws, lv, use_ms);
  }
  TVMI_RETURN_LAUNCH_STATUS("tvmi_roi_align_backward");
}

int fill_levels(MsLevels& lv, void* const* ptrs, const int64_t* heights, const int64_t* widths, const double* scales,
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

}  // namespace
}  // namespace tvmi

extern "C" size_t tvmi_roi_align_backward_workspace_bytes(int64_t N, int64_t K, int64_t pooled_h, int64_t pooled_w) {
  if (N <= 0 || K < 0 || !tvmi::owner_shape(pooled_h, pooled_w)) return 0;
  const size_t b = tvmi::own_workspace_bytes(N, K, (int)pooled_h, (int)pooled_w);
  return b <= (size_t(4) << 30) ? b : 0;  // beyond 4 GiB of coefficient tables the atomic path is the saner choice
}

extern "C" int tvmi_roi_align_backward_overwrites(tvmi_dtype dt, int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                                                  int64_t pooled_h, int64_t pooled_w, int64_t c_stride, int64_t h_stride,
                                                  int64_t w_stride, size_t workspace_bytes) {
  return tvmi::owner_applies(dt, N, C, K, pooled_h, pooled_w, &H, &W, 1, c_stride, h_stride, w_stride, workspace_bytes) ? 1 : 0;
}

extern "C" int tvmi_roi_align_backward(const void* grad, const void* rois, void* grad_input, tvmi_dtype dt, int64_t N,
                                       int64_t C, int64_t H, int64_t W, int64_t K, int64_t pooled_h, int64_t pooled_w,
                                       double spatial_scale, int64_t sampling_ratio, int aligned, int64_t n_stride,
                                       int64_t c_stride, int64_t h_stride, int64_t w_stride, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  TVMI_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "roi_align: pooled size must be positive");
  TVMI_CHECK_ARG(N >= 0 && C >= 0 && H >= 0 && W >= 0 && K >= 0, "roi_align_backward: negative size");
  if (N * C * H * W == 0) return 0;
  TVMI_CHECK_ARG(grad_input && (K == 0 || (grad && rois)), "roi_align_backward: null pointer");
  TVMI_CHECK_ARG(H * W < (1ll << 31) && K * tvmi::ceil_div(C, 32) < (1ll << 31), "roi_align_backward: size exceeds 32-bit launch limits");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (workspace && tvmi::owner_applies(dt, N, C, K, pooled_h, pooled_w, &H, &W, 1, c_stride, h_stride, w_stride, workspace_bytes)) {
    TVMI_CHECK_ARG(dt == TVMI_F32 || (reinterpret_cast<uintptr_t>(grad) & 7) == 0, "roi_align_backward: 16-bit grads must be 8-byte aligned");
    tvmi::MsLevels ms;
    void* ptr = grad_input;
    tvmi::fill_levels(ms, &ptr, &H, &W, &spatial_scale, 1, 0, 0, 224.0, 4.0, 1e-6);
    return tvmi::dispatch_owner(dt, /*rois_f32=*/false, grad, rois, ms, 0, N, C, K, pooled_h, pooled_w, sampling_ratio, aligned, n_stride,
                                c_stride, workspace, s);
  }
  if (K * C == 0) return 0;
  TVMI_DISPATCH_FLOAT(dt, "roi_align_backward",
                      return tvmi::launch_atomic<scalar_t>(grad, rois, grad_input, C, H, W, K, pooled_h, pooled_w, spatial_scale,
                                                           sampling_ratio, aligned, n_stride, c_stride, h_stride, w_stride, s,
                                                           nullptr));
  return 0;
}

extern "C" int tvmi_multiscale_roi_align_backward_overwrites(tvmi_dtype dt, int64_t N, int64_t C, int64_t K,
                                                             const int64_t* heights, const int64_t* widths, int64_t n_levels,
                                                             int64_t pooled_h, int64_t pooled_w, int64_t c_stride,
                                                             int64_t h_stride, int64_t w_stride, size_t workspace_bytes) {
  if (!heights || !widths) return 0;
  return tvmi::owner_applies(dt, N, C, K, pooled_h, pooled_w, heights, widths, n_levels, c_stride, h_stride, w_stride, workspace_bytes) ? 1 : 0;
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
  if (C == 0 || N == 0) return 0;
  TVMI_CHECK_ARG(grad_inputs && heights && widths && spatial_scales && (K == 0 || (grad && rois)),
                 "multiscale_roi_align_backward: null pointer");
  TVMI_CHECK_ARG(dt == TVMI_F32 || dt == TVMI_F16 || dt == TVMI_BF16,
                 "multiscale_roi_align_backward: float32 / float16 / bfloat16 gradients (RoIs are always float32)");
  TVMI_CHECK_ARG(K * tvmi::ceil_div(C, 32) < (1ll << 31), "multiscale_roi_align_backward: size exceeds 32-bit launch limits");
  for (int64_t i = 0; i < n_levels; ++i)
    TVMI_CHECK_ARG(grad_inputs[i] != nullptr && heights[i] > 0 && widths[i] > 0 && heights[i] * widths[i] * C < (1ll << 31),
                   "multiscale_roi_align_backward: bad level");
  tvmi::MsLevels ms;
  tvmi::fill_levels(ms, grad_inputs, heights, widths, spatial_scales, n_levels, k_min, k_max, canonical_scale, canonical_level, eps);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (workspace && tvmi::owner_applies(dt, N, C, K, pooled_h, pooled_w, heights, widths, n_levels, c_stride, h_stride, w_stride, workspace_bytes)) {
    TVMI_CHECK_ARG(dt == TVMI_F32 || (reinterpret_cast<uintptr_t>(grad) & 7) == 0, "multiscale_roi_align_backward: 16-bit grads must be 8-byte aligned");
    return tvmi::dispatch_owner(dt, /*rois_f32=*/true, grad, rois, ms, 1, N, C, K, pooled_h, pooled_w, sampling_ratio, aligned, n_stride,
                                c_stride, workspace, s);
  }
  if (K == 0) return 0;
  TVMI_CHECK_ARG(dt == TVMI_F32, "multiscale_roi_align_backward: 16-bit gradients need the tile-owner regime (workspace, contiguous "
                                 "7x7 / 14x14 bins, even channel count)");
  // H / W / scale / grad_input of the single-level signature are placeholders: every RoI takes them from its level
  return tvmi::launch_atomic<float>(grad, rois, grad_inputs[0], C, heights[0], widths[0], K, pooled_h, pooled_w, spatial_scales[0],
                                    sampling_ratio, aligned, n_stride, c_stride, h_stride, w_stride, s, &ms);
}
