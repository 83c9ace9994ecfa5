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
// Structure (differs from cuda/nms_kernel.cu:56-148, whose second kernel is one block doing
// N serial steps with a barrier each):
//   K1 nms_mask_tiles  : one wave per upper-triangular 64x64 tile.  Lane = column box (in
//      registers), the 64 row boxes sit in LDS and are broadcast; for every row the 64-lane
//      predicate becomes the 64-bit suppression word through the compare itself (__ballot =
//      an SGPR pair) and is parked in lane `row`.  The mask is stored TILE-major: tile
//      (rb, cb) is 64 consecutive words, so both this store and every later read of a tile
//      is one coalesced 512-byte access.  Boxes are gathered through `order` in-kernel.
//      Two rows per step with packed fp32 and a division-free band test (see thr_band below); above 4096 boxes the
//      kernel runs in chunks of 64 row blocks and drops the rows the sweep already knows to be suppressed: the row block
//      is compacted as it is staged, the pair loop covers the survivors only.
//   The greedy sweep is blocked and runs UNDER the mask kernels on side streams (see launch()):
//   K2 nms_colreduce   : removed[cb] |= OR over the rows kept in a range of row blocks of tile(rb, cb)[row] —
//      thousands of independent waves, each a masked 512-byte load + a DPP OR-reduction + one atomic.  Used as the
//      PUSH of a resolved chunk's suppression to the later column blocks (near part on the sweep stream, far part on
//      its own stream).
//   K3 nms_resolve_wide: one 16-wave workgroup per chunk of 64 blocks, walked in mini super-blocks of 16.  Wave c owns
//      column block c: it pulls the tiles of the earlier mini super-blocks (keep bits final, in LDS), then the 16
//      diagonal blocks are resolved by parallel fixed-point rounds (exact when two rounds agree) with the serial walk
//      over the not-yet-removed bits (s_ff1 + v_readlane) as fallback.  Kept original indices are written in score
//      order and the running count stays on the device — no masked_select pass.
//   Large problems are RE-PLANNED on their survivors (tvmi_nms_blocking, see launch()): after the first chunks have been swept
//   and pushed to every later column, the score order is compacted to the boxes still alive and the pipeline starts over
//   on that list.  The resolver and the push kernels hand over through memory words (agent-scope atomics, no fences:
//   SweepSync) instead of stream events wherever the three streams are known to run concurrently.
//   Up to 4096 boxes one mask launch + nms_sweep_small; batched NMS: segment-major kernels further down.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include <memory>
#include <mutex>
#include <vector>

#include "tvmi_common.h"

namespace tvmi {
namespace {

constexpr int kMaskWaves = 4;    // waves (= tiles of one row block) per workgroup in K1
constexpr int kSuper = 16;       // 64-box blocks per super-block
constexpr int kReduceRows = 8;   // row blocks folded per wave in K2
constexpr int kWide = 64;        // 64-box blocks per chunk of the large path (one resolve launch)
typedef unsigned long long u64;

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

// ---- wave-wide OR of a 32-bit value: DPP inside each 16-lane row, v_readlane across rows
__device__ __forceinline__ unsigned int wave_or32(unsigned int v) {
  v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
  return (unsigned int)__builtin_amdgcn_readlane((int)v, 0) | (unsigned int)__builtin_amdgcn_readlane((int)v, 16) |
         (unsigned int)__builtin_amdgcn_readlane((int)v, 32) | (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ u64 wave_or64(u64 v) {
  return ((u64)wave_or32((unsigned int)(v >> 32)) << 32) | (u64)wave_or32((unsigned int)v);
}
__device__ __forceinline__ u64 uniform64(u64 v) {  // value is wave-uniform: move it to SGPRs
  return ((u64)(unsigned int)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
         (u64)(unsigned int)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ u64 readlane64(u64 v, int lane) {
  return ((u64)(unsigned int)__builtin_amdgcn_readlane((int)(v >> 32), lane) << 32) |
         (u64)(unsigned int)__builtin_amdgcn_readlane((int)v, lane);
}

typedef float v2f __attribute__((ext_vector_type(2)));

// The suppression predicate `(double)(inter / union) > thr` (cpu/nms_kernel.cpp:88) without the IEEE division in
// the common case.  With u = the smallest float above thr and d = the largest float at or below it, RN(q) > thr
// <=> RN(q) >= u, which q >= u guarantees, and RN(q) <= d is guaranteed by q <= d.  Take hi >= u (1 + 2^-20),
// lo <= d (1 - 2^-20), c = (hi + lo) / 2, r = 0.75 (hi - lo) and evaluate, in float,
//       t = fma(-c, union, inter)        (one rounding),        ru = r * union:
//   t >  ru  =>  inter > (c + r (1 - 2^-22)) union >= hi union > u union   => certainly suppressed,
//   t < -ru  =>  inter < (c - r (1 - 2^-22)) union <= lo union < d union   => certainly not,
// (the extra quarter of the band width in r swallows the roundings of c, t and ru: they are ~2^-24 relative, the band is
// 2^-19 relative), and only pairs with |t| <= ru — IoU within ~2^-19 of the threshold — need the exact division.  The
// relative-error argument needs normal numbers and a positive union: the fast path is only taken for a tile whose boxes
// all have an area in [2^-60, 2^60] (then every coordinate is finite, inter <= min(area_i, area_j) by monotonic
// rounding, union >= max(area_i, area_j) (1 - 2^-23) > 0 and r * union >= 2^-101), and for thresholds in
// [2^-20, 2^20]; everything else is evaluated with the reference's expression.  The result is bit-identical to
// evaluating the division everywhere; the mask kernels are VALU-bound and the division + double compare were ~40 % of
// their instructions.
struct ThrBand {
  float c, r;
};
inline ThrBand thr_band(double thr) {
  ThrBand b{0.f, INFINITY};  // = "always take the exact path" (|t| <= inf)
  if (!(thr >= 1.0 / 1048576.0 && thr <= 1048576.0)) return b;
  float d = (float)thr;
  if ((double)d > thr) d = nextafterf(d, -INFINITY);  // largest float <= thr
  const float u = nextafterf(d, INFINITY);            // smallest float > thr
  const float hi = nextafterf((float)((double)u * (1.0 + 1.0 / 1048576.0)), INFINITY);
  const float lo = nextafterf((float)((double)d * (1.0 - 1.0 / 1048576.0)), -INFINITY);
  b.c = 0.5f * (hi + lo);
  b.r = 0.75f * (hi - lo);
  return b;
}
constexpr float kAreaMin = 0x1p-60f, kAreaMax = 0x1p60f;  // fast-path range of box areas (see above)

// One 64x64 suppression tile: lane = column box (registers), the row boxes come from LDS in component-major form
// (`rows[k * RS + i]`, k = x1,y1,x2,y2,area, i = row: two consecutive rows are one 8-byte read; `row_keys` = their
// segment ids or nullptr).  `nrows` = rows of the block that exist (the others hold zeros or anything at all: their
// words are never consumed).  Returns, in lane r, the 64-bit word of row r.
//
// exact form: the reference's expression for every pair (also the only form for float64 boxes)
template <typename T, int RS>
__device__ __forceinline__ u64 suppression_tile_exact(const T* __restrict__ rows, const long long* __restrict__ row_keys,
                                                      T jx1, T jy1, T jx2, T jy2, T jarea, long long jkey, u64 valid_cols,
                                                      bool diag, double thr, u64 skip_rows) {
  const int lane = threadIdx.x & 63;
  u64 mine = 0ull;
  for (int i = 0; i < 64; ++i) {
    if ((skip_rows >> i) & 1ull) continue;  // wave-uniform: a row nobody will read (see nms_mask_tiles)
    const T ix1 = rows[0 * RS + i], iy1 = rows[1 * RS + i], ix2 = rows[2 * RS + i], iy2 = rows[3 * RS + i];
    const T iarea = rows[4 * RS + i];
    const T xx1 = ix1 > jx1 ? ix1 : jx1;  // std::max(ix1, x1[j])
    const T yy1 = iy1 > jy1 ? iy1 : jy1;
    const T xx2 = jx2 < ix2 ? jx2 : ix2;  // std::min(ix2, x2[j])
    const T yy2 = jy2 < iy2 ? jy2 : iy2;
    const T dw = xx2 - xx1, dh = yy2 - yy1;
    const T w = (T)0 < dw ? dw : (T)0;    // std::max(0, xx2 - xx1)
    const T h = (T)0 < dh ? dh : (T)0;
    const T inter = w * h;
    bool p = (double)(inter / (iarea + jarea - inter)) > thr;
    if (diag) p = p && (lane > i);
    if (row_keys) p = p && (jkey == row_keys[i]);
    const u64 word = __ballot(p) & valid_cols;
    if (lane == i) mine = word;
  }
  return mine;
}

// fast form (float32): fully unrolled, TWO rows per step, written for the machine's real bottleneck.  The mask kernels
// are bound by instruction issue (a wave64 VALU instruction occupies its SIMD for 4 cycles, and a CU has ONE scalar ALU
// for its four SIMDs, so a scalar instruction costs as much issue time as a vector one).  With the rows of a pair side
// by side in a register pair, everything except min / max / compare runs as packed fp32 (v_pk_add / v_pk_mul /
// v_pk_fma: two rows per instruction): the differences, the product, the union and the band test of thr_band().  The
// "undecided" bookkeeping stays on the vector side too: each lane keeps the minimum of |t| - r*union over the rows
// (positive <=> every pair of the lane was decided) — one v_sub with |.| per row and one v_min3 per row pair, no
// scalar mask arithmetic.  Raw v_max / v_min (the builtins would canonicalise each LDS operand with an extra
// instruction; they differ from std::max / std::min only for NaN operands, which the area check of the fast path
// excludes).  The row word is parked with v_writelane (lane select through M0).  Per row pair: 8 min/max, 4 max(0,.),
// 7 packed ops, 2 compares, 2 |t| - ru, 1 min3, 4 writelanes, 2 M0 moves; a tile with any undecided pair is simply
// redone in the exact form, so the result is bit-identical to evaluating the division everywhere.
template <int I, bool DIAG, bool KEYS, int RS>
__device__ __forceinline__ void suppression_row_pair(const float* __restrict__ rows, const long long* __restrict__ row_keys,
                                                     float jx1, float jy1, float jx2, float jy2, v2f jarea2, long long jkey,
                                                     v2f negc, v2f r2, int& mine_lo, int& mine_hi, float& margin) {
  const v2f ix1 = *reinterpret_cast<const v2f*>(rows + 0 * RS + I), iy1 = *reinterpret_cast<const v2f*>(rows + 1 * RS + I);
  const v2f ix2 = *reinterpret_cast<const v2f*>(rows + 2 * RS + I), iy2 = *reinterpret_cast<const v2f*>(rows + 3 * RS + I);
  const v2f iarea = *reinterpret_cast<const v2f*>(rows + 4 * RS + I);
  float a0, a1, b0, b1, c0, c1, d0, d1;
  asm("v_max_f32 %0, %1, %2" : "=v"(a0) : "v"(ix1.x), "v"(jx1));
  asm("v_max_f32 %0, %1, %2" : "=v"(a1) : "v"(ix1.y), "v"(jx1));
  asm("v_max_f32 %0, %1, %2" : "=v"(b0) : "v"(iy1.x), "v"(jy1));
  asm("v_max_f32 %0, %1, %2" : "=v"(b1) : "v"(iy1.y), "v"(jy1));
  asm("v_min_f32 %0, %1, %2" : "=v"(c0) : "v"(ix2.x), "v"(jx2));
  asm("v_min_f32 %0, %1, %2" : "=v"(c1) : "v"(ix2.y), "v"(jx2));
  asm("v_min_f32 %0, %1, %2" : "=v"(d0) : "v"(iy2.x), "v"(jy2));
  asm("v_min_f32 %0, %1, %2" : "=v"(d1) : "v"(iy2.y), "v"(jy2));
  const v2f dw = v2f{c0, c1} - v2f{a0, a1}, dh = v2f{d0, d1} - v2f{b0, b1};
  v2f w, h;
  w.x = 0.f < dw.x ? dw.x : 0.f;
  w.y = 0.f < dw.y ? dw.y : 0.f;
  h.x = 0.f < dh.x ? dh.x : 0.f;
  h.y = 0.f < dh.y ? dh.y : 0.f;
  const v2f inter = w * h;
  const v2f uni = (iarea + jarea2) - inter;
  const v2f t = __builtin_elementwise_fma(negc, uni, inter);
  const v2f ru = r2 * uni;
  u64 word0 = __ballot(t.x > ru.x), word1 = __ballot(t.y > ru.y);
  const float m0 = __builtin_fabsf(t.x) - ru.x, m1 = __builtin_fabsf(t.y) - ru.y;
  asm("v_min3_f32 %0, %0, %1, %2" : "+v"(margin) : "v"(m0), "v"(m1));
  if (DIAG) {
    word0 &= ~0ull << ((I + 1) & 63);                          // only columns after the row
    word1 &= I + 1 < 63 ? (~0ull << ((I + 2) & 63)) : 0ull;
  }
  if (KEYS) {
    word0 &= __ballot(jkey == row_keys[I]);
    word1 &= __ballot(jkey == row_keys[I + 1]);
  }
  // gfx9 takes the lane select of v_writelane from an SGPR or M0 (an inline constant assembles but selects the wrong
  // lane for I >= 32 — measured), and only one SGPR may sit on the constant bus: the row index goes through M0,
  // which the caller saves and restores around the 64 rows.  (M0 is a reserved register for this compiler: naming it
  // in the clobber list is rejected with "clobber list contains reserved registers", so the save / restore pair it is;
  // nothing between the pair can be given an M0 use by the compiler — this TU has no LDS-DMA, movrel or GWS code.)
  asm volatile("s_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %4, m0"
               : "+v"(mine_lo), "+v"(mine_hi)
               : "s"((int)(unsigned)word0), "n"(I), "s"((int)(unsigned)(word0 >> 32)));
  asm volatile("s_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %4, m0"
               : "+v"(mine_lo), "+v"(mine_hi)
               : "s"((int)(unsigned)word1), "n"(I + 1), "s"((int)(unsigned)(word1 >> 32)));
}

template <int I, bool DIAG, bool KEYS, int RS>
__device__ __forceinline__ void suppression_row_pair_unless_skipped(const float* __restrict__ rows,
                                                                    const long long* __restrict__ row_keys, float jx1, float jy1,
                                                                    float jx2, float jy2, v2f jarea2, long long jkey, v2f negc,
                                                                    v2f r2, int& mine_lo, int& mine_hi, float& margin,
                                                                    u64 skip_rows) {
  // wave-uniform: both rows of the pair are already known to be suppressed — nobody reads their words
  if (((skip_rows >> I) & 3ull) != 3ull)
    suppression_row_pair<I, DIAG, KEYS, RS>(rows, row_keys, jx1, jy1, jx2, jy2, jarea2, jkey, negc, r2, mine_lo, mine_hi, margin);
}

template <bool DIAG, bool KEYS, int RS, int... Is>
__device__ __forceinline__ void suppression_rows(std::integer_sequence<int, Is...>, const float* __restrict__ rows,
                                                 const long long* __restrict__ row_keys, float jx1, float jy1, float jx2,
                                                 float jy2, v2f jarea2, long long jkey, v2f negc, v2f r2, int& mine_lo,
                                                 int& mine_hi, float& margin, u64 skip_rows) {
  (suppression_row_pair_unless_skipped<2 * Is, DIAG, KEYS, RS>(rows, row_keys, jx1, jy1, jx2, jy2, jarea2, jkey, negc, r2,
                                                               mine_lo, mine_hi, margin, skip_rows),
   ...);
}

template <typename T, int RS>
__device__ __forceinline__ u64 suppression_tile(const T* __restrict__ rows, const long long* __restrict__ row_keys, int nrows,
                                                T jx1, T jy1, T jx2, T jy2, T jarea, long long jkey, bool jvalid, bool diag,
                                                double thr, ThrBand band, u64 skip_rows = 0ull) {
  const u64 valid_cols = __ballot(jvalid);
  if constexpr (std::is_same<T, float>::value) {
    const int lane = threadIdx.x & 63;
    const float rarea = rows[4 * RS + lane];
    const bool out_of_range = (jvalid && !(jarea >= kAreaMin && jarea <= kAreaMax)) ||
                              (lane < nrows && !(rarea >= kAreaMin && rarea <= kAreaMax));
    if (__ballot(out_of_range) == 0ull) {
      int mine_lo = 0, mine_hi = 0, m0_save;
      float margin = INFINITY;
      const v2f jarea2 = {jarea, jarea}, negc = {-band.c, -band.c}, r2 = {band.r, band.r};
      const auto seq = std::make_integer_sequence<int, 32>{};
      asm volatile("s_mov_b32 %0, m0" : "=s"(m0_save));
      if (row_keys) {
        if (diag) suppression_rows<true, true, RS>(seq, rows, row_keys, jx1, jy1, jx2, jy2, jarea2, jkey, negc, r2, mine_lo, mine_hi, margin, skip_rows);
        else suppression_rows<false, true, RS>(seq, rows, row_keys, jx1, jy1, jx2, jy2, jarea2, jkey, negc, r2, mine_lo, mine_hi, margin, skip_rows);
      } else {
        if (diag) suppression_rows<true, false, RS>(seq, rows, row_keys, jx1, jy1, jx2, jy2, jarea2, jkey, negc, r2, mine_lo, mine_hi, margin, skip_rows);
        else suppression_rows<false, false, RS>(seq, rows, row_keys, jx1, jy1, jx2, jy2, jarea2, jkey, negc, r2, mine_lo, mine_hi, margin, skip_rows);
      }
      asm volatile("s_mov_b32 m0, %0" : : "s"(m0_save));
      // rows that do not exist hold zeros in the mask kernels (decided: inter = 0) or leftovers in the small-segment
      // kernel (may or may not be decided: at worst a spurious exact evaluation of a last block)
      if ((__ballot(!(margin > 0.f)) & valid_cols) == 0ull)
        return (((u64)(unsigned)mine_hi << 32) | (u64)(unsigned)mine_lo) & valid_cols;
    }
  }
  return suppression_tile_exact<T, RS>(rows, row_keys, jx1, jy1, jx2, jy2, jarea, jkey, valid_cols, diag, thr, skip_rows);
}

// mask layout: tile (rb, cb) = 64 words at mask + (rb*CB + cb)*64; word r = row rb*64+r.
template <typename T>
__global__ __launch_bounds__(kMaskWaves * kWave) void nms_mask_tiles(
    const T* __restrict__ dets, const int64_t* __restrict__ order, const int64_t* __restrict__ seg, int n, int CB,
    double thr, ThrBand band, u64* __restrict__ mask, int rb0, const u64* removed) {
  __shared__ __attribute__((aligned(16))) T s_row[5][64];  // x1,y1,x2,y2,area of the row block, component-major
  __shared__ long long s_seg[64];
  __shared__ u64 s_skip;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int rb = rb0 + blockIdx.y;  // large problems are launched in chunks of row blocks (see launch())
  const int cb = blockIdx.x * kMaskWaves + wave;
  if ((int)(blockIdx.x * kMaskWaves + kMaskWaves - 1) < rb) return;  // whole workgroup left of the diagonal
  const int row0 = rb * 64;
  if (threadIdx.x < 64) {
    // Rows already known to be suppressed (by kept boxes of chunks the sweep has finished) are never read by anybody: a
    // kept row is by definition not removed, and both the column reduction and the resolve step only use kept rows.  The
    // word may be stale (the sweep runs concurrently on other streams and only ever ADDS bits): stale = fewer rows
    // dropped.  The surviving rows are COMPACTED to the front of the LDS block (the dropped ones and the rows past n
    // become zero boxes behind them), so that the tile loop below runs over ceil(survivors / 2) row pairs instead of
    // testing pair by pair whether both rows happen to be dropped; every wave of the workgroup uses the one word read here.
    u64 skip = 0ull;
    if (removed) skip = uniform64(__hip_atomic_load(&removed[rb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const int r = row0 + lane;
    const u64 below = (1ull << lane) - 1ull;
    const u64 surv = ~skip & __ballot(r < n);
    const bool live = (surv >> lane) & 1ull;
    const int slot = live ? __popcll(surv & below) : __popcll(surv) + __popcll(~surv & below);
    T x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    long long sg = 0;
    if (live) {
      const int64_t oi = order[r];
      const Box<T> b = load_box<T>(dets, oi);
      x1 = b.x1;
      y1 = b.y1;
      x2 = b.x2;
      y2 = b.y2;
      if (seg) sg = seg[oi];
    }
    s_row[0][slot] = x1;
    s_row[1][slot] = y1;
    s_row[2][slot] = x2;
    s_row[3][slot] = y2;
    s_row[4][slot] = (x2 - x1) * (y2 - y1);
    s_seg[slot] = sg;
    if (lane == 0) s_skip = ~surv;  // dropped or non-existent
  }
  __syncthreads();
  if (cb < rb || cb >= CB) return;
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
  const u64 dropped = uniform64(s_skip);
  const int nsurv = 64 - __popcll(dropped);
  // compact row k sits in lane k of `packed`; the row pairs at and beyond nsurv are skipped as a whole
  const u64 packed = suppression_tile<T, 64>(&s_row[0][0], seg ? s_seg : nullptr, nsurv, jx1, jy1, jx2, jy2, jarea, jseg, jvalid,
                                             false, thr, band, nsurv >= 64 ? 0ull : ~0ull << nsurv);
  u64 mine = packed;
  if (dropped) {  // wave-uniform: back to one word per ORIGINAL row (lane r <- compact slot of row r; dropped rows: 0)
    const int src = __popcll(~dropped & ((1ull << lane) - 1ull)) << 2;
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)packed);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)(packed >> 32));
    mine = ((dropped >> lane) & 1ull) ? 0ull : (((u64)hi << 32) | (u64)lo);
  }
  if (cb == rb) mine &= lane < 63 ? ~0ull << (lane + 1) : 0ull;  // diagonal tile: only the columns after the row
  mask[((size_t)rb * CB + cb) * 64 + lane] = mine;
}

// K2: fold the rows kept in a range of row blocks into removed[cb] for a range of column blocks (see launch()).
__global__ __launch_bounds__(256) void nms_colreduce(const u64* __restrict__ mask, const u64* __restrict__ keepbits,
                                                     u64* __restrict__ removed, int CB, int rlo, int rhi, int c0, int c1) {
  // removed[cb] |= OR over the rows kept in row blocks [rlo, rhi) of tile(rb, cb)[row], for cb in [c0, c1)
  __builtin_amdgcn_s_setprio(2);  // on the sweep's critical path, under the mask kernels of later chunks
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int cb = c0 + blockIdx.x;
  if (cb >= c1) return;
  const int rb_begin = rlo + (blockIdx.y * 4 + wave) * kReduceRows;
  u64 acc = 0ull;
#pragma unroll
  for (int q = 0; q < kReduceRows; ++q) {
    const int rb = rb_begin + q;
    if (rb < rhi) {
      const u64 w = mask[((size_t)rb * CB + cb) * 64 + lane];
      const u64 kb = keepbits[rb];
      if ((kb >> lane) & 1ull) acc |= w;
    }
  }
  const u64 red = wave_or64(acc);
  if (lane == 0 && red) atomicOr(&removed[cb], red);
}

// ---- device-side hand-offs of the sweep ("nms.device_handoff", see launch()).  The resolve workgroup of chunk c and the push
// kernel of chunk c run in DIFFERENT launches on different streams and meet through two words per chunk: `flag` (set by the
// resolver once the chunk's keep bits are in memory) and the arrival counter of the near-push kernel, which the resolver of
// the next chunk polls.  No fences: on the 8-XCD part an agent-scope release is a write-back
// of the whole L2 slice — which the mask kernel keeps full of dirty tiles — and cost more than the event hops it replaced
// (measured).  Instead everything that crosses between the two launches is an agent-scope (write-through /
// L2-bypassing) atomic — keepbits[], removed[], the three words — and a signal is preceded by `s_waitcnt vmcnt(0)`: the
// writer's stores and atomics are acknowledged before the word that announces them is written (the idiom nms_small_seg_sweep's
// ticket already uses).  The mask tiles a push kernel reads were written by a mask kernel that completed before the resolver
// started (stream event), and nobody reads a tile before it is written, so no stale copy of one can sit in any L2.
// A poll that outlives its bound (about a second: the launches it waits for are not running beside it — a tool that
// serialises kernels, a debugger, a co-tenant that keeps the resolver off the CUs) GIVES UP: it raises the call's `failed`
// word (one more SweepSync behind the per-chunk ones) and goes on with whatever is in memory; every later poll of the call
// sees the word and returns at once, so the launch chain drains in bounded time.  The host reads the word when the call has
// finished (tvmi_nms_blocking synchronises anyway) and re-runs the call in the stream-event form: a slower correct answer
// instead of a GPU fault (VERDICT r03 item 4 — this used to be __builtin_trap()).
struct SweepSync {  // one per chunk, zeroed with removed[] at the start of a level
  int flag, near_done, pad[2];
};
__device__ __forceinline__ void poll_until_equal(const int* p, int want, int* failed) {
  int polls = 0;
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
    __builtin_amdgcn_s_sleep(4);
    ++polls;
    if ((polls & 1023) == 0 && __hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    if (polls > (1 << 20)) {
      __hip_atomic_store(failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
}
__device__ __forceinline__ void signal_after_my_stores(int* p) {  // one thread, after the workgroup's barrier
  __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The NEAR push of chunk [b0, b1) with device-side hand-offs: a fixed grid of 4-wave workgroups that is ALREADY RESIDENT
// (polling the chunk's flag) when the resolver finishes; folds the chunk's kept rows into removed[] of the next chunk's column
// blocks [b1, b2) — all the next resolver still lacks — and reports.  The FAR push (every later column block) is the wide
// nms_colreduce launch right behind it on the same stream, whose workgroups report to the chunk's far counter.
constexpr int kPushGroups = 128;
__global__ __launch_bounds__(256) void nms_push(const u64* __restrict__ mask, const u64* keepbits, u64* __restrict__ removed, int CB,
                                                int b0, int b1, int b2, SweepSync* sync, int* failed) {
  __shared__ u64 s_kb[kWide];
  __builtin_amdgcn_s_setprio(2);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (threadIdx.x == 0) poll_until_equal(&sync->flag, 1, failed);
  __syncthreads();
  // this kernel started before the resolver wrote them: agent-scope loads (the far push, a later launch, reads them plainly)
  if ((int)threadIdx.x < b1 - b0) s_kb[threadIdx.x] = __hip_atomic_load(&keepbits[b0 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int groups = (b1 - b0 + kReduceRows - 1) / kReduceRows;
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  for (int item = gw; item < (b2 - b1) * groups; item += nw) {
    const int cb = b1 + item / groups, g = item % groups;
    u64 acc = 0ull;
#pragma unroll
    for (int q = 0; q < kReduceRows; ++q) {
      const int r = g * kReduceRows + q;   // row block within the chunk
      if (r < b1 - b0) {
        const u64 w = mask[((size_t)(b0 + r) * CB + cb) * 64 + lane];
        if ((s_kb[r] >> lane) & 1ull) acc |= w;
      }
    }
    const u64 red = wave_or64(acc);
    if (lane == 0 && red) atomicOr(&removed[cb], red);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my atomics are acknowledged
  __syncthreads();
  if (threadIdx.x == 0) signal_after_my_stores(&sync->near_done);
}

// K3 for large problems: ONE workgroup resolves a WIDE super-block (chunk) of up to kWide 64-box blocks [b0, b1): it
// walks it in mini super-blocks of kSuper blocks; removed[] (filled by the pushes of the earlier chunks, nms_colreduce)
// carries what every earlier chunk suppresses, the rows of earlier mini super-blocks of THIS chunk are pulled here by
// the wave that owns the column (keep bits in LDS), then the diagonal blocks are resolved in registers.
constexpr int kJacobiRounds = 24;  // parallel fixed-point rounds tried per mini super-block before the serial walk
__global__ __launch_bounds__(kSuper * kWave) void nms_resolve_wide(const u64* __restrict__ mask,
                                                                   const int64_t* __restrict__ order,
                                                                   const u64* __restrict__ removed,
                                                                   u64* __restrict__ keepbits, int n, int CB, int b0, int b1,
                                                                   int64_t* __restrict__ keep_out,
                                                                   int64_t* __restrict__ num_keep, SweepSync* sync, int chunk,
                                                                   int* failed, int lose_flag) {
  __shared__ u64 s_keep[kWide];
  __shared__ u64 s_jac[2][kSuper];
  __shared__ int s_changed[3];
  __shared__ int s_base[kWide + 1];
  // this workgroup is the serial link of the sweep and shares the chip with the mask kernels of later chunks:
  // its waves take issue priority over whatever else is resident on the CU
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63;
  const int c_loc = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nb = b1 - b0;
  if (sync) {  // device hand-offs: removed[b0..b1) is complete once the push kernels of the two previous chunks have reported
    if (threadIdx.x == 0) {
      // (the far push of chunk - 2 precedes the near push of chunk - 1 on their in-order stream: one poll covers both)
      if (chunk >= 1) poll_until_equal(&sync[chunk - 1].near_done, kPushGroups, failed);
    }
    __syncthreads();
  }
  for (int m0 = 0; m0 < nb; m0 += kSuper) {
    const int lb = m0 + c_loc;  // local block of this wave in the wide super-block
    const int cb = b0 + lb;
    const bool have = lb < nb;
    u64 diag = 0ull, above[kSuper - 1];
    u64 rem = 0ull;
    if (have) {
      diag = mask[((size_t)cb * CB + cb) * 64 + lane];
      // written by atomics of push kernels that may still be running for LATER chunks: read at agent scope
      rem = __hip_atomic_load(&removed[cb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int q = 0; q < kSuper - 1; ++q) {
      above[q] = 0ull;
      if (have && q < c_loc) above[q] = mask[((size_t)(b0 + m0 + q) * CB + cb) * 64 + lane];
    }
    u64 acc = 0ull;
    if (have) {
      // earlier mini super-blocks of this wide block: keep bits are final, in LDS.  16 independent tile loads per
      // round (one coalesced 512-byte access each) so that the memory latency is paid once per round, not per tile.
      // (Requesting all of them up front was measured: 128 VGPRs + spills at 1024 threads, slower.)
      for (int r0 = 0; r0 < m0; r0 += kSuper) {
        u64 w[kSuper];
#pragma unroll
        for (int q = 0; q < kSuper; ++q) w[q] = mask[((size_t)(b0 + r0 + q) * CB + cb) * 64 + lane];
#pragma unroll
        for (int q = 0; q < kSuper; ++q) acc |= ((s_keep[r0 + q] >> lane) & 1ull) ? w[q] : 0ull;
      }
    }
    rem = uniform64(rem) | (m0 > 0 ? wave_or64(acc) : 0ull);
    const int rows_here = have ? min(64, n - cb * 64) : 0;
    const u64 valid = rows_here >= 64 ? ~0ull : ((1ull << rows_here) - 1ull);
    // ---- the kSuper x kSuper tiles of this mini super-block form a strictly upper-triangular system
    //          keep_c = ~(rem_c | OR over kept rows r of blocks q <= c of tile(q, c)[r]) & valid,
    // whose unique solution is the greedy sweep's answer.  Instead of walking its 16 blocks one barrier-separated step
    // at a time, iterate it in parallel (all 16 waves at once, Jacobi): after t rounds every box whose chain of
    // "suppressed by a box that is itself suppressed by ..." is at most t long is final, and a round without change
    // is the fixed point.  Real detections need a handful of rounds; a chain longer than kJacobiRounds falls back to
    // the serial walk below (same answer, old speed).
    u64 my_keep = ~rem & valid;
    bool converged = false;
    if (lane == 0) s_jac[0][c_loc] = have ? my_keep : 0ull;
    if (threadIdx.x == 0) s_changed[0] = 0;
    for (int it = 0; it < kJacobiRounds; ++it) {
      if (threadIdx.x == 0) s_changed[(it + 1) % 3] = 0;
      __syncthreads();
      const u64* cur = s_jac[it & 1];
      u64 contrib = 0ull;
#pragma unroll
      for (int q = 0; q < kSuper - 1; ++q) contrib |= ((cur[q] >> lane) & 1ull) ? above[q] : 0ull;  // above[q] = 0 for q >= c_loc
      contrib |= ((cur[c_loc] >> lane) & 1ull) ? diag : 0ull;
      const u64 nk = have ? (~(rem | wave_or64(contrib)) & valid) : 0ull;
      if (lane == 0) {
        s_jac[(it + 1) & 1][c_loc] = nk;
        if (nk != my_keep) s_changed[it % 3] = 1;
      }
      my_keep = nk;
      __syncthreads();
      if (s_changed[it % 3] == 0) {
        converged = true;
        break;
      }
    }
    if (converged) {
      if (lane == 0 && have) {
        s_keep[lb] = my_keep;
        __hip_atomic_store(&keepbits[cb], my_keep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      continue;
    }
#pragma unroll
    for (int step = 0; step < kSuper; ++step) {
      if (step == c_loc && have) {
        u64 r = uniform64(rem);
        u64 active = uniform64(__ballot(diag != 0ull)) & ~r & valid;
        while (active) {
          const int k = __builtin_ctzll(active);
          r |= readlane64(diag, k);
          active &= ~(r | (1ull << k));
        }
        if (lane == 0) {
          s_keep[lb] = ~r & valid;
          __hip_atomic_store(&keepbits[cb], ~r & valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
      if (step < kSuper - 1 && have && step < c_loc) {
        const u64 kb = s_keep[m0 + step];
        const u64 contrib = ((kb >> lane) & 1ull) ? above[step] : 0ull;
        rem |= wave_or64(contrib);
      }
    }
  }
  if (sync) {  // the keep bits of the chunk are acknowledged stores: release the push kernel before the index list is emitted
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // lose_flag: test hook ("nms.handoff_lose_flag"): chunk 0 never announces itself, as if this workgroup had not run
    // beside the push kernel — the recovery path (poll gives up, host re-runs with stream events) is then exercised for real
    if (threadIdx.x == 0 && !(lose_flag && chunk == 0)) signal_after_my_stores(&sync[chunk].flag);
  }
  // append the kept original indices in score order
  if (threadIdx.x == 0) {
    int run = 0;
    for (int b = 0; b < nb; ++b) {
      s_base[b] = run;
      run += __popcll(s_keep[b]);
    }
    s_base[nb] = run;
  }
  __syncthreads();
  const int64_t base = *num_keep;
  for (int b = c_loc; b < nb; b += kSuper) {
    const u64 kb = s_keep[b];
    if ((kb >> lane) & 1ull) {
      const u64 below = kb & ((1ull << lane) - 1ull);
      keep_out[base + s_base[b] + __popcll(below)] = order[(int64_t)(b0 + b) * 64 + lane];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) *num_keep = base + s_base[nb];
}

// K3': the whole sweep of a small problem (CB <= kSmallCB column blocks, i.e. N <= 4096) in ONE
// workgroup and ONE launch: the super-blocks are walked in order, the column reduction over
// earlier super-blocks is done by the wave that owns the column (its lanes accumulate the masked
// tile words, one DPP reduction at the end) with the keep bits held in LDS, then the same
// register-resident resolve chain as nms_resolve.  Removes 2 memsets + 2 launches per
// super-block from the latency-bound case (RPN / box-head sizes).
constexpr int kSmallCB = 64;
__global__ __launch_bounds__(kSuper * kWave) void nms_sweep_small(const u64* __restrict__ mask,
                                                                  const int64_t* __restrict__ order, int n, int CB,
                                                                  int64_t* __restrict__ keep_out,
                                                                  int64_t* __restrict__ num_keep, bool append) {
  __shared__ u64 s_keepbits[kSmallCB];
  __shared__ int s_base[kSmallCB + 1];
  const int lane = threadIdx.x & 63;
  const int c_loc = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (int b0 = 0; b0 < CB; b0 += kSuper) {
    const int b1 = min(CB, b0 + kSuper);
    const int cb = b0 + c_loc;
    const bool have = cb < b1;
    u64 diag = 0ull, above[kSuper - 1];
    if (have) diag = mask[((size_t)cb * CB + cb) * 64 + lane];
#pragma unroll
    for (int q = 0; q < kSuper - 1; ++q) {
      above[q] = 0ull;
      if (have && q < c_loc) above[q] = mask[((size_t)(b0 + q) * CB + cb) * 64 + lane];
    }
    // pull from every earlier super-block (their keep bits are final and in LDS)
    u64 acc = 0ull;
    if (have) {
      for (int rb = 0; rb < b0; ++rb) {
        const u64 w = mask[((size_t)rb * CB + cb) * 64 + lane];
        if ((s_keepbits[rb] >> lane) & 1ull) acc |= w;
      }
    }
    u64 rem = b0 > 0 ? wave_or64(acc) : 0ull;
    const int rows_here = have ? min(64, n - cb * 64) : 0;
    const u64 valid = rows_here >= 64 ? ~0ull : ((1ull << rows_here) - 1ull);
#pragma unroll
    for (int step = 0; step < kSuper; ++step) {
      if (step == c_loc && have) {
        u64 r = uniform64(rem);
        u64 active = uniform64(__ballot(diag != 0ull)) & ~r & valid;
        while (active) {
          const int k = __builtin_ctzll(active);
          r |= readlane64(diag, k);
          active &= ~(r | (1ull << k));
        }
        if (lane == 0) s_keepbits[cb] = ~r & valid;
      }
      __syncthreads();
      if (step < kSuper - 1 && have && step < c_loc) {
        const u64 kb = s_keepbits[b0 + step];
        const u64 contrib = ((kb >> lane) & 1ull) ? above[step] : 0ull;
        rem |= wave_or64(contrib);
      }
    }
  }
  // exclusive prefix of the per-block keep counts, then every wave appends its blocks' indices
  if (threadIdx.x == 0) {
    int run = 0;
    for (int b = 0; b < CB; ++b) {
      s_base[b] = run;
      run += __popcll(s_keepbits[b]);
    }
    s_base[CB] = run;
  }
  const int64_t base = append ? *num_keep : 0;  // append: the survivors of a re-planned large problem (see launch())
  __syncthreads();
  for (int b = c_loc; b < CB; b += kSuper) {
    const u64 kb = s_keepbits[b];
    if ((kb >> lane) & 1ull) {
      const u64 below = kb & ((1ull << lane) - 1ull);
      keep_out[base + s_base[b] + __popcll(below)] = order[(int64_t)b * 64 + lane];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) *num_keep = base + s_base[CB];
}

// ---- re-planning on the survivors (see launch()): the boxes of column blocks >= pb that no kept box has removed,
// in score order.  Two small kernels: per-block survivor counts -> exclusive offsets (+ the total, also written to
// pinned host memory), then one wave per block copies its surviving entries of `order`.
__global__ __launch_bounds__(1024) void nms_survivor_offsets(const u64* __restrict__ removed, int n, int CB, int pb,
                                                             int* __restrict__ offsets, int* __restrict__ total_host) {
  __shared__ int s_part[1024];
  const int words = CB - pb;
  const int per = (words + 1023) / 1024;
  const int w0 = threadIdx.x * per, w1 = min(words, w0 + per);
  int sum = 0;
  for (int w = w0; w < w1; ++w) {
    const int rows = min(64, n - (pb + w) * 64);
    const u64 valid = rows >= 64 ? ~0ull : ((1ull << rows) - 1ull);
    sum += __popcll(~removed[pb + w] & valid);
  }
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // inclusive Hillis-Steele scan of the 1024 partial sums
    const int v = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
    __syncthreads();
    s_part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = s_part[threadIdx.x] - sum;
  for (int w = w0; w < w1; ++w) {
    const int rows = min(64, n - (pb + w) * 64);
    const u64 valid = rows >= 64 ? ~0ull : ((1ull << rows) - 1ull);
    offsets[w] = run;
    run += __popcll(~removed[pb + w] & valid);
  }
  if (threadIdx.x == 1023) *total_host = s_part[1023];
}
__global__ __launch_bounds__(256) void nms_compact_order(const u64* __restrict__ removed, const int64_t* __restrict__ order,
                                                         int n, int CB, int pb, const int* __restrict__ offsets,
                                                         int64_t* __restrict__ order_out) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= CB - pb) return;
  const int rows = min(64, n - (pb + w) * 64);
  const u64 valid = rows >= 64 ? ~0ull : ((1ull << rows) - 1ull);
  const u64 surv = ~removed[pb + w] & valid;
  if ((surv >> lane) & 1ull) order_out[offsets[w] + __popcll(surv & ((1ull << lane) - 1ull))] = order[(int64_t)(pb + w) * 64 + lane];
}

__global__ void nms_report_failed(const int* __restrict__ failed, int* __restrict__ host_word) { *host_word = *failed; }

// Fork / join helpers of the large-problem path.  The mask kernel (throughput-bound, fills the chip) is launched in
// chunks of kWide row blocks on one internal stream; the sweep of chunk c (latency-bound: one resolve workgroup plus
// the near push) runs on a second, higher-priority stream as soon as chunk c of the mask is complete — i.e. UNDER the
// mask kernels of the later chunks instead of after them; the far pushes run on a third stream (see launch()).  All
// three are forked from and joined back into the caller's stream with events, so the caller sees ordinary
// stream-ordered behaviour (and the pattern is capturable).  Streams / events are cached per host thread and device.
struct SweepStreams {
  hipStream_t mask_stream = nullptr, sweep_stream = nullptr, far_stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr, join_far = nullptr;
  std::vector<hipEvent_t> chunk_done, resolved, far_done;
  int* host_count = nullptr;  // pinned host word the survivor count of a re-planned problem is read back through
  bool ready = false;
  bool ensure(int nchunks) {
    if (!ready) {  // first use on this thread for this device (the caller has made it current)
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      if (hipStreamCreateWithPriority(&mask_stream, hipStreamNonBlocking, lo) != hipSuccess) return false;
      if (hipStreamCreateWithPriority(&sweep_stream, hipStreamNonBlocking, hi) != hipSuccess) return false;
      if (hipStreamCreateWithPriority(&far_stream, hipStreamNonBlocking, hi) != hipSuccess) return false;
      if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return false;
      if (hipEventCreateWithFlags(&join, hipEventDisableTiming) != hipSuccess) return false;
      if (hipEventCreateWithFlags(&join_far, hipEventDisableTiming) != hipSuccess) return false;
      if (hipHostMalloc(reinterpret_cast<void**>(&host_count), 64, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        host_count = nullptr;  // no re-planning, everything else works
      }
      ready = true;
    }
    for (std::vector<hipEvent_t>* v : {&chunk_done, &resolved, &far_done})
      while ((int)v->size() < nchunks) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
        v->push_back(e);
      }
    return true;
  }
  // Handles are destroyed when the owning host thread exits.  Errors are ignored on purpose: at process teardown the
  // HIP runtime may already be gone (hipErrorDeinitialized), and a handle whose work is still queued is released by
  // the runtime once that work has drained.
  ~SweepStreams() {
    for (std::vector<hipEvent_t>* v : {&chunk_done, &resolved, &far_done})
      for (hipEvent_t e : *v) (void)hipEventDestroy(e);
    for (hipEvent_t e : {fork, join, join_far})
      if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {mask_stream, sweep_stream, far_stream})
      if (st) (void)hipStreamDestroy(st);
    if (host_count) (void)hipHostFree(host_count);
  }
};
// one set per (host thread, device): a thread that alternates between GPUs keeps both sets instead of re-creating
// (and leaking) streams on every switch
struct SweepStreamsByDevice {
  std::vector<std::unique_ptr<SweepStreams>> per_device;
  SweepStreams* current() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
    if ((int)per_device.size() <= dev) per_device.resize(dev + 1);
    if (!per_device[dev]) per_device[dev] = std::make_unique<SweepStreams>();
    return per_device[dev].get();
  }
};
thread_local SweepStreamsByDevice g_sweep_streams;

// ONE call per device at a time may use the device-side hand-offs.  Every poll in them targets work that the SAME host thread
// enqueued earlier, so one call can never block itself however its streams are mapped to hardware queues — but two host
// threads whose polling kernels sit at the heads of each other's shared in-order queues could.  A second call that arrives
// while the first one's work is still on the device therefore takes the stream-event form (whose kernels never wait).
struct HandoffClaim {
  std::mutex mu;
  bool enqueuing = false;     // a call that took the hand-off form is still enqueueing its launches
  hipEvent_t done = nullptr;  // recorded behind the last such call
};
HandoffClaim g_handoff_claim[64];
// A profiler that collects hardware counters per dispatch runs ONE kernel at a time (rocprofv3 --pmc: measured — the polling
// push kernel then waits for a resolver that is never started).  rocprofv3 announces that mode to the process it launches.
// HIP_LAUNCH_BLOCKING / AMD_SERIALIZE_KERNEL serialise every launch of the process, and the older rocprof generations announce
// counter collection through ROCP_* variables.  What none of these catch (a debugger, a co-tenant that starves the resolver) is
// caught on the device: a poll gives up after about a second, the call is re-run in the stream-event form and the process
// stops using the hand-offs (g_handoff_broken).
inline bool env_nonzero(const char* name) {
  const char* v = std::getenv(name);
  return v != nullptr && v[0] != '\0' && !(v[0] == '0' && v[1] == '\0');
}
inline bool kernels_are_serialised() {
  static const bool yes = std::getenv("ROCPROF_COUNTER_COLLECTION") != nullptr || std::getenv("ROCPROF_COUNTERS") != nullptr ||
                          std::getenv("ROCP_METRICS") != nullptr || std::getenv("ROCP_INPUT") != nullptr ||
                          env_nonzero("HIP_LAUNCH_BLOCKING") || env_nonzero("AMD_SERIALIZE_KERNEL") ||
                          env_nonzero("CUDA_LAUNCH_BLOCKING") || env_nonzero("TVMI_TOOL_SERIALIZED_PROFILER");
  return yes;
}
std::atomic<bool> g_handoff_broken{false};   // a poll of this process has timed out once: stream events from now on
inline bool claim_handoff(int dev) {
  if (dev < 0 || dev >= 64 || kernels_are_serialised() || g_handoff_broken.load(std::memory_order_relaxed)) return false;
  HandoffClaim& h = g_handoff_claim[dev];
  std::lock_guard<std::mutex> lock(h.mu);
  if (h.enqueuing) return false;
  if (h.done && hipEventQuery(h.done) != hipSuccess) {
    (void)hipGetLastError();   // hipErrorNotReady is not an error of this call
    return false;
  }
  h.enqueuing = true;
  return true;
}
inline void release_handoff(int dev, hipStream_t stream) {
  HandoffClaim& h = g_handoff_claim[dev];
  std::lock_guard<std::mutex> lock(h.mu);
  if (!h.done && hipEventCreateWithFlags(&h.done, hipEventDisableTiming) != hipSuccess) h.done = nullptr;
  if (h.done) (void)hipEventRecord(h.done, stream);
  h.enqueuing = false;
}

// Options of the large path (tvmi_set_option): "nms.replan_min_boxes" — problems at least this large are re-planned on
// their survivors (0 = never); "nms.replan_divisor" — the first 1/divisor of the row chunks is swept before the re-plan;
// "nms.replan_max" — how many times one call may re-plan; "nms.mask_lds_bytes" — dynamic LDS per mask workgroup.
std::atomic<int64_t> g_replan_min_boxes{24576};
std::atomic<int> g_replan_divisor{16}, g_replan_max{3}, g_mask_lds_bytes{36000}, g_device_handoff{1}, g_handoff_lose_flag{0};
std::atomic<bool> g_step_fused{true};    // "nms.step_fused": detector-step sizes through the one-launch kernel (read by the glue)

// Workspace of the large path: mask tiles | removed[CB] | keepbits[CB] | survivor offsets[CB] (int) | two score-order
// buffers of n indices (re-planning ping-pongs between them).
struct LargeWorkspace {
  u64 *mask, *removed, *keepbits;
  SweepSync* sync;   // one per chunk, directly behind keepbits[] (cleared with them)
  int* offsets;
  int64_t* order_buf[2];
  int* failed;       // raised by a hand-off poll that gave up (sticky for the whole call: outside the per-level clears)
};
inline size_t large_state_bytes(size_t n) {
  const size_t CB = ceil_div(n, (size_t)64);
  return 2 * CB * sizeof(u64) + ceil_div(CB, (size_t)kWide) * sizeof(SweepSync) + ceil_div(CB, (size_t)2) * 2 * sizeof(int) +
         2 * n * sizeof(int64_t) + 16;
}
inline size_t sweep_state_bytes(size_t CB) { return 2 * CB * sizeof(u64) + ceil_div(CB, (size_t)kWide) * sizeof(SweepSync); }
inline LargeWorkspace carve(void* workspace, size_t n) {
  const size_t CB = ceil_div(n, (size_t)64);
  LargeWorkspace w;
  w.mask = static_cast<u64*>(workspace);
  w.removed = w.mask + CB * CB * 64;
  w.keepbits = w.removed + CB;
  w.sync = reinterpret_cast<SweepSync*>(w.keepbits + CB);
  w.offsets = reinterpret_cast<int*>(w.sync + ceil_div(CB, (size_t)kWide));
  w.order_buf[0] = reinterpret_cast<int64_t*>(w.offsets + ceil_div(CB, (size_t)2) * 2);
  w.order_buf[1] = w.order_buf[0] + n;
  w.failed = reinterpret_cast<int*>(w.order_buf[1] + n);
  return w;
}

template <typename T>
int launch(const void* dets, const int64_t* order, const int64_t* seg, int64_t n, double thr, void* workspace,
           int64_t* keep_out, int64_t* num_keep, hipStream_t stream, bool may_sync, bool allow_handoff = true) {
  const T* d = static_cast<const T*>(dets);
  const int64_t n_call = n;   // a hand-off that gave up re-runs the call from here
  const ThrBand band = thr_band(thr);
  if (ceil_div(n, 64) <= kSmallCB) {  // latency-bound sizes: one mask launch, the whole sweep is one more
    const int CB = (int)ceil_div(n, 64);
    u64* mask = static_cast<u64*>(workspace);
    const dim3 grid((unsigned)ceil_div(CB, kMaskWaves), (unsigned)CB);
    nms_mask_tiles<T><<<grid, dim3(kMaskWaves * kWave), 0, stream>>>(d, order, seg, (int)n, CB, thr, band, mask, 0, nullptr);
    nms_sweep_small<<<dim3(1), dim3(kSuper * kWave), 0, stream>>>(mask, order, (int)n, CB, keep_out, num_keep, false);
    TVMI_RETURN_LAUNCH_STATUS("tvmi_nms");
  }
  const LargeWorkspace ws = carve(workspace, (size_t)n);
  u64 *mask = ws.mask, *removed = ws.removed, *keepbits = ws.keepbits;
  const size_t state_bytes = sweep_state_bytes((size_t)ceil_div(n, 64));   // removed[] + keepbits[] + the hand-off words
  hipError_t e = hipMemsetAsync(num_keep, 0, sizeof(int64_t), stream);
  if (e == hipSuccess) e = hipMemsetAsync(ws.failed, 0, sizeof(int), stream);
  if (e != hipSuccess) return set_error((int)e, "tvmi_nms: memset");
  static SweepStreams no_streams;  // never ensure()d: only names the members below when the device query failed
  SweepStreams* ssp = g_sweep_streams.current();
  SweepStreams& ss = ssp ? *ssp : no_streams;
  // RE-PLANNING ON THE SURVIVORS.  The pair tests of a sorted list are wasted on boxes that an early, high-scoring
  // box has already removed: after the first eighth of the rows has been swept (and its removals pushed to EVERY later
  // column) typically half (sparse scenes) to nine tenths (dense scenes) of the remaining boxes are gone.  Dropping
  // the removed ROWS of later tiles (nms_mask_tiles) only saves that fraction once; restarting on the compacted list
  // of survivors saves it in both dimensions, and shortens the serial sweep by the same factor.  The price is one
  // host synchronisation per re-plan (the survivor count sizes the next grids), so it is only done for large
  // problems, never under stream capture, and `tvmi_set_option("nms.replan_min_boxes", 0)` turns it off.
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  const bool capturing = !(hipStreamIsCapturing(stream, &capture) == hipSuccess && capture == hipStreamCaptureStatusNone);
  const bool may_replan = may_sync && ssp && !capturing;
  const size_t mask_lds = (size_t)g_mask_lds_bytes.load(std::memory_order_relaxed);
  const int64_t replan_min = g_replan_min_boxes.load(std::memory_order_relaxed);
  const int divisor = std::max(2, g_replan_divisor.load(std::memory_order_relaxed));
  int replans_left = g_replan_max.load(std::memory_order_relaxed);
  const int64_t* cur = order;
  int flip = 0;
  bool ok = true;
  bool claim_tried = false, claimed = false;   // the device-side hand-offs of this call (see HandoffClaim)
  int claim_dev = -1;
  struct ClaimRelease {   // on every way out: "done" is recorded behind whatever this call enqueued on the caller's stream
    bool& claimed;
    int& dev;
    hipStream_t stream;
    ~ClaimRelease() {
      if (claimed) release_handoff(dev, stream);
    }
  } claim_release{claimed, claim_dev, stream};
  bool side_streams_open = false;  // a re-plan leaves the three side streams forked, idle and ahead of `stream`
  // Everything the side streams did is ordered before what the caller enqueues next: the sweep stream has waited for every
  // mask chunk, so joining it and the far stream joins all three.  BOTH on every way out: with device hand-offs
  // the far stream is released by the resolver's flag, which is raised BEFORE the resolver appends to keep_out / num_keep.
  auto join_side_streams = [&]() {
    bool j = hipEventRecord(ss.join, ss.sweep_stream) == hipSuccess && hipStreamWaitEvent(stream, ss.join, 0) == hipSuccess;
    j = (hipEventRecord(ss.join_far, ss.far_stream) == hipSuccess && hipStreamWaitEvent(stream, ss.join_far, 0) == hipSuccess) && j;
    side_streams_open = false;
    return j;
  };
  while (true) {
    const int CB = (int)ceil_div(n, 64);
    if (CB <= kSmallCB) {  // what survived a re-plan fits the one-workgroup sweep: append to the list (streams are idle)
      const dim3 grid((unsigned)ceil_div(CB, kMaskWaves), (unsigned)CB);
      nms_mask_tiles<T><<<grid, dim3(kMaskWaves * kWave), 0, stream>>>(d, cur, seg, (int)n, CB, thr, band, mask, 0, nullptr);
      nms_sweep_small<<<dim3(1), dim3(kSuper * kWave), 0, stream>>>(mask, cur, (int)n, CB, keep_out, num_keep, true);
      break;
    }
    const int nchunks = (int)ceil_div(CB, kWide);
    bool forked = side_streams_open;
    if (!forked) {
      e = hipMemsetAsync(removed, 0, state_bytes, stream);
      if (e != hipSuccess) return set_error((int)e, "tvmi_nms: memset");   // (the side streams are not forked here)
      forked = ssp && ss.ensure(nchunks) && hipEventRecord(ss.fork, stream) == hipSuccess &&
               hipStreamWaitEvent(ss.mask_stream, ss.fork, 0) == hipSuccess &&
               hipStreamWaitEvent(ss.sweep_stream, ss.fork, 0) == hipSuccess &&
               hipStreamWaitEvent(ss.far_stream, ss.fork, 0) == hipSuccess;
    }
    hipStream_t ms = forked ? ss.mask_stream : stream, sw = forked ? ss.sweep_stream : stream, fs = forked ? ss.far_stream : stream;
    // device-side hand-offs need the three streams to really run concurrently: not under capture (a graph may serialise
    // its branches — may_sync is false there anyway only for tvmi_nms, so ask), and not without the side streams
    // ... and only where the call may synchronise at its end to learn whether a poll gave up (tvmi_nms_blocking)
    if (!claim_tried && forked && !capturing && may_sync && allow_handoff && ss.host_count &&
        g_device_handoff.load(std::memory_order_relaxed) != 0) {
      claim_tried = true;
      (void)hipGetDevice(&claim_dev);
      claimed = claim_handoff(claim_dev);
    }
    const bool handoff = forked && claimed;
    const bool replan = may_replan && forked && ss.host_count && replans_left > 0 && replan_min > 0 && n >= replan_min;
    const int limit = replan ? std::max(1, (nchunks + divisor / 2) / divisor) : nchunks;  // row chunks swept at this level
    auto mask_chunk = [&](int c, bool skip_known) {
      const int r0 = c * kWide, r1 = std::min(CB, r0 + kWide);
      // row block r only has tiles for column blocks >= r: workgroups left of the chunk's first row block exit at once
      const dim3 grid((unsigned)ceil_div(CB, kMaskWaves), (unsigned)(r1 - r0));
      // From the second chunk on a resolve step runs under the mask kernel.  Its ONE 16-wave workgroup can only start on
      // a CU with four free wave slots on every SIMD, which a chip saturated by 4-wave workgroups offers rarely
      // (kernel-trace: resolve 33 us alone, 110-180 us under a full mask kernel).  Dynamic LDS the kernel never touches
      // caps these launches at four workgroups (16 of 32 wave slots) per CU: the resolver starts at once (37-47 us),
      // and the rows its pushes remove are dropped by the mask workgroups that start after them.  The cap costs the
      // mask kernel ~13 % of its throughput, so the first chunk — which runs alone — is launched without it.
      nms_mask_tiles<T><<<grid, dim3(kMaskWaves * kWave), c >= 1 ? mask_lds : 0, ms>>>(d, cur, seg, (int)n, CB, thr, band, mask, r0,
                                                                                      skip_known ? removed : nullptr);
    };
    auto push = [&](hipStream_t st, int rlo, int rhi, int c0, int c1) {  // rows kept in [rlo, rhi) -> removed[c0..c1)
      if (c1 <= c0 || rhi <= rlo) return;
      const dim3 rgrid((unsigned)(c1 - c0), (unsigned)ceil_div(rhi - rlo, 4 * kReduceRows));
      nms_colreduce<<<rgrid, dim3(256), 0, st>>>(mask, keepbits, removed, CB, rlo, rhi, c0, c1);
    };
    auto resolve = [&](int c) {
      const int b0 = c * kWide, b1 = std::min(CB, b0 + kWide);
      nms_resolve_wide<<<dim3(1), dim3(kSuper * kWave), 0, sw>>>(mask, cur, removed, keepbits, (int)n, CB, b0, b1, keep_out,
                                                                 num_keep, handoff ? ws.sync : nullptr, c, ws.failed,
                                                                 handoff ? g_handoff_lose_flag.load(std::memory_order_relaxed) : 0);
    };
    if (forked) {
      // PUSH pipeline on three streams.  Chunk c: its mask tiles (mask stream), then — sweep stream, the serial link —
      // resolve(c) and the NEAR push (rows kept in chunk c -> removed[] of chunk c + 1, all resolve(c + 1) still lacks),
      // then — far stream, off the critical path — the FAR push to every later column block, which has the whole of
      // resolve(c + 1) to finish before resolve(c + 2) needs it.  Because removed[] of a chunk is now filled as the
      // sweep advances (not just before its own resolve), the mask kernel of chunk c can DROP the rows that are already
      // known to be suppressed; it is held back until the far push of chunk c - 3 is done, so that it sees (at least)
      // every removal caused by chunks <= c - 3 — in a sorted list that is nearly all of them.
      for (int c = 0; c < limit; ++c) {
        const int b0 = c * kWide, b1 = std::min(CB, b0 + kWide), b2 = std::min(CB, b1 + kWide);
        const bool last_before_replan = replan && c == limit - 1 && limit < nchunks;
        if (c >= 3) ok = ok && hipStreamWaitEvent(ms, ss.far_done[c - 3], 0) == hipSuccess;
        mask_chunk(c, true);
        ok = ok && hipEventRecord(ss.chunk_done[c], ms) == hipSuccess && hipStreamWaitEvent(sw, ss.chunk_done[c], 0) == hipSuccess;
        if (handoff) {
          // Device-side hand-offs: the sweep stream carries ONE event wait per chunk (the chunk's tiles) and back-to-back
          // resolve launches; the push kernel of the chunk is already resident on the far stream, polling the resolver's
          // flag, and the resolvers of the next two chunks poll its two arrival counters.  An event hop between streams
          // costs ~10 us of command-processor latency here and the event form has two of them on the serial chain of
          // every chunk (58 us per chunk for 33 us of resolve).  Every wait targets work that was enqueued earlier, so
          // any interleaving of the three queues — even a single shared one — makes progress.
          resolve(c);
          if (b1 < CB) nms_push<<<dim3(kPushGroups), dim3(256), 0, fs>>>(mask, keepbits, removed, CB, b0, b1, b2, ws.sync + c, ws.failed);
          // the FAR push needs no counter: the far stream is in order, so the near push of chunk c + 1 — whose counter the
          // resolver of chunk c + 2 polls — cannot even start before this launch has finished
          push(fs, b0, b1, b2, CB);
          ok = ok && hipEventRecord(ss.far_done[c], fs) == hipSuccess;
          (void)last_before_replan;
          continue;
        }
        if (c >= 2) ok = ok && hipStreamWaitEvent(sw, ss.far_done[c - 2], 0) == hipSuccess;
        resolve(c);
        if (!last_before_replan) push(sw, b0, b1, b1, b2);
        ok = ok && hipEventRecord(ss.resolved[c], sw) == hipSuccess && hipStreamWaitEvent(fs, ss.resolved[c], 0) == hipSuccess;
        push(fs, b0, b1, last_before_replan ? b1 : b2, CB);  // nobody waits for a near part before a re-plan: one launch
        ok = ok && hipEventRecord(ss.far_done[c], fs) == hipSuccess;
      }
    } else {  // no side streams (creation failed): same kernels, serially on the caller's stream
      for (int c = 0; c < nchunks; ++c) mask_chunk(c, false);
      for (int c = 0; c < nchunks; ++c) {
        const int b0 = c * kWide, b1 = std::min(CB, b0 + kWide);
        resolve(c);
        push(stream, b0, b1, b1, CB);
      }
    }
    if (!ok || limit >= nchunks) {
      if (forked) ok = join_side_streams() && ok;
      break;
    }
    // Every kept row of blocks < pb has been pushed to every column once the far stream is through: compact the rest
    // there (no event hop), clear the sweep state for the next level, and wait for exactly that stream — everything
    // the other two did at this level precedes it.  The side streams stay forked: the next level's launches go
    // straight to them, an idle machine needs no fork.
    const int pb = limit * kWide;
    int64_t* next = ws.order_buf[flip];
    flip ^= 1;
    nms_survivor_offsets<<<dim3(1), dim3(1024), 0, fs>>>(removed, (int)n, CB, pb, ws.offsets, ss.host_count);
    nms_compact_order<<<dim3((unsigned)ceil_div(CB - pb, 4)), dim3(256), 0, fs>>>(removed, cur, (int)n, CB, pb, ws.offsets, next);
    e = hipMemsetAsync(removed, 0, state_bytes, fs);
    if (e == hipSuccess) e = hipStreamSynchronize(fs);
    if (e != hipSuccess) {
      (void)join_side_streams();   // work may still be queued on the side streams while the caller frees the workspace
      return set_error((int)e, "tvmi_nms: synchronising for the survivor count");
    }
    side_streams_open = true;
    const int survivors = *static_cast<volatile int*>(ss.host_count);
    cur = next;
    n = survivors;
    --replans_left;
    if (survivors <= 0) {  // nothing left: close the fork
      ok = join_side_streams() && ok;
      break;
    }
    if (ceil_div(n, 64) <= kSmallCB) {  // the one-workgroup sweep runs on the caller's stream: it follows BOTH side streams
      ok = join_side_streams() && ok;   // (it appends to keep_out / num_keep, which the last resolver writes after its flag)
    }
  }
  if (!ok) return set_error((int)hipErrorUnknown, "tvmi_nms: stream fork / join failed");
  if (claimed) {
    // Did a hand-off poll give up?  The call may synchronise (its caller reads num_keep next): read the word through the
    // pinned host slot; if it is raised the result is unspecified — re-run the whole call in the stream-event form.
    nms_report_failed<<<dim3(1), dim3(1), 0, stream>>>(ws.failed, ss.host_count + 1);
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return set_error((int)e, "tvmi_nms: synchronising for the hand-off status");
    if (static_cast<volatile int*>(ss.host_count)[1] != 0) {
      g_handoff_broken.store(true, std::memory_order_relaxed);
      release_handoff(claim_dev, stream);
      claimed = false;
      return launch<T>(dets, order, seg, n_call, thr, workspace, keep_out, num_keep, stream, may_sync, /*allow_handoff=*/false);
    }
  }
  TVMI_RETURN_LAUNCH_STATUS("tvmi_nms");
}


// ---------------------------------------------------------------------------------------
// Segment-major batched NMS.  With segment ids (classes / levels / images) the suppression
// matrix is block diagonal once the boxes are laid out segment by segment (score order kept
// inside a segment): only tiles whose two 64-box blocks share a segment are evaluated, and
// every segment is swept by its own workgroup, concurrently — instead of N x N pair tests
// and ONE serial sweep over all N boxes in global score order.  The caller provides the
// stable partition of the score order by segment (`perm`, `keys`: a second stable sort);
// the result is emitted in global score order, identical to the unsegmented formulation.
//   position p (segment-major)  --perm-->  g (rank in global score order)  --order-->  box
// Blocks are NOT padded per segment: a 64-box block may hold the tail of one segment and the
// heads of others; cross-segment mask bits are zero, so a workgroup sweeping the block range
// of "its" segments may resolve foreign rows of its first block incorrectly — it never writes
// them (each keep bit is published by exactly one workgroup, with an atomicOr).
constexpr int kSegMaxBlocks = 128;  // blocks one sweep workgroup handles (8,192 boxes per segment); its pull over
                                   // earlier super-blocks is O(blocks^2) tile reads by ONE workgroup, so larger
                                   // segments are sent back to the global-order pipeline (colreduce + resolve)

// The same bound found by a whole wave (all 64 lanes call it with the same arguments): 64 probes per round trip instead of
// one — 3 dependent loads for 100k keys instead of 17.  (Round 4's lane-0 binary searches, two per row block, were most of
// nms_mask_tiles_seg's 70 us at 100k boxes x 80 classes: every workgroup began with 34 serial global round trips.)
__device__ __forceinline__ int upper_bound_key_wave(const int64_t* __restrict__ keys, int n, int64_t v) {
  const int lane = threadIdx.x & 63;
  int lo = 0, hi = n;   // invariant: every p < lo has keys[p] <= v, every p >= hi has keys[p] > v
  while (hi - lo > 64) {
    const int step = (hi - lo + 63) >> 6;
    const int p = lo + lane * step;
    const bool le = p < hi && keys[p] <= v;
    const int cnt = __builtin_popcountll(__ballot(le));   // keys ascending: the lanes with le form a prefix
    const int nlo = cnt > 0 ? lo + (cnt - 1) * step + 1 : lo;   // probe cnt - 1 holds <= v
    const int nhi = min(hi, lo + cnt * step);                   // probe cnt (if any) holds > v
    lo = __builtin_amdgcn_readfirstlane(nlo);
    hi = __builtin_amdgcn_readfirstlane(nhi);
  }
  const int p = lo + lane;
  const bool le = p < hi && keys[p] <= v;
  return lo + __builtin_popcountll(__ballot(le));
}

__device__ __forceinline__ int upper_bound_key(const int64_t* __restrict__ keys, int n, int64_t v) {
  int lo = 0, hi = n;  // first p with keys[p] > v
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] <= v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// `n_dev` (all kernels of the segment-major and the small-segment path): the boxes that take part, read on the device.
// The caller's lists are laid out for the CAPACITY n, and the boxes that take part are the first *n_dev of BOTH the
// score order and the segment-major order (masked-out candidates carry score -inf and the largest key, so both sorts
// put them last): every kernel clamps its n and the launch grids, sized for the capacity, retire their surplus blocks.
__device__ __forceinline__ int live_boxes(int n, const int64_t* __restrict__ n_dev) {
  if (n_dev == nullptr) return n;
  const int64_t v = *n_dev;
  return (int)min(max(v, (int64_t)0), (int64_t)n);
}

__global__ __launch_bounds__(256) void nms_seg_layout(const int64_t* __restrict__ order, const int64_t* __restrict__ perm,
                                                      int n, const int64_t* __restrict__ n_dev, int64_t* __restrict__ oidx,
                                                      int* __restrict__ invperm) {
  n = live_boxes(n, n_dev);
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int64_t g = perm[p];
  oidx[p] = order[g];
  invperm[g] = p;
}

// one workgroup per row block; its waves walk the column blocks that share a segment with it
template <typename T>
__global__ __launch_bounds__(kMaskWaves * kWave) void nms_mask_tiles_seg(const T* __restrict__ dets,
                                                                         const int64_t* __restrict__ oidx,
                                                                         const int64_t* __restrict__ keys, int n,
                                                                         const int64_t* __restrict__ n_dev, int CB,
                                                                         double thr, ThrBand band, u64* __restrict__ mask) {
  __shared__ __attribute__((aligned(16))) T s_row[5][64];
  __shared__ long long s_key[64];
  __shared__ int s_cbmax;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int rb = blockIdx.x;
  const int row0 = rb * 64;
  n = live_boxes(n, n_dev);
  if (row0 >= n) return;
  if (threadIdx.x < 64) {
    const int r = row0 + lane;
    T x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    long long k = 0;
    if (r < n) {
      const Box<T> b = load_box<T>(dets, oidx[r]);
      x1 = b.x1;
      y1 = b.y1;
      x2 = b.x2;
      y2 = b.y2;
      k = keys[r];
    }
    s_row[0][lane] = x1;
    s_row[1][lane] = y1;
    s_row[2][lane] = x2;
    s_row[3][lane] = y2;
    s_row[4][lane] = (x2 - x1) * (y2 - y1);
    s_key[lane] = k;
    {
      // column blocks that share a segment with this row block end with the segment of its last row; a
      // segment beyond the sweep limit is going to be redone by the global-order pipeline: skip its tiles
      const int64_t klast = keys[min(n - 1, row0 + 63)];
      const int cbm = (upper_bound_key_wave(keys, n, klast) - 1) >> 6;
      const int first_blk = upper_bound_key_wave(keys, n, klast - 1) >> 6;  // ids are integers: first p with key >= klast
      if (lane == 0) s_cbmax = (cbm - first_blk + 1 > kSegMaxBlocks) ? rb - 1 : cbm;
    }
  }
  __syncthreads();
  const int cbmax = s_cbmax;
  for (int cb = rb + wave; cb <= cbmax; cb += kMaskWaves) {
    const int j = cb * 64 + lane;
    T jx1 = 0, jy1 = 0, jx2 = 0, jy2 = 0;
    long long jkey = 0;
    const bool jvalid = j < n;
    if (jvalid) {
      const Box<T> b = load_box<T>(dets, oidx[j]);
      jx1 = b.x1;
      jy1 = b.y1;
      jx2 = b.x2;
      jy2 = b.y2;
      jkey = keys[j];
    }
    const T jarea = (jx2 - jx1) * (jy2 - jy1);
    const u64 mine = suppression_tile<T, 64>(&s_row[0][0], s_key, min(64, n - row0), jx1, jy1, jx2, jy2, jarea, jkey, jvalid,
                                             cb == rb, thr, band);
    mask[((size_t)rb * CB + (cb - rb)) * 64 + lane] = mine;
  }
}

// one workgroup per 64-box block in which at least one segment starts: it sweeps the blocks
// from its own to the end of the last segment that starts in it
__global__ __launch_bounds__(kSuper * kWave) void nms_sweep_seg(const u64* __restrict__ mask,
                                                                const int64_t* __restrict__ keys, int n,
                                                                const int64_t* __restrict__ n_dev, int CB,
                                                                u64* __restrict__ keepbits, int* __restrict__ err) {
  __shared__ u64 s_keepbits[kSegMaxBlocks];
  __shared__ int s_info[3];  // first start position, end position, number of blocks
  const int lane = threadIdx.x & 63;
  const int c_loc = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int B0 = blockIdx.x;
  n = live_boxes(n, n_dev);
  if (B0 * 64 >= n) return;
  if (threadIdx.x < 64) {
    const int p = B0 * 64 + lane;
    const bool start = p < n && (p == 0 || keys[p] != keys[p - 1]);
    const u64 starts = __ballot(start);
    if (starts == 0ull) {   // (wave-uniform)
      if (lane == 0) s_info[2] = 0;
    } else {
      const int end = upper_bound_key_wave(keys, n, keys[min(n - 1, B0 * 64 + 63)]);
      if (lane == 0) {
        s_info[0] = B0 * 64 + __builtin_ctzll(starts);
        s_info[1] = end;
        s_info[2] = ((end - 1) >> 6) + 1 - B0;
      }
    }
  }
  __syncthreads();
  const int nb = s_info[2];
  if (nb == 0) return;
  if (nb > kSegMaxBlocks) {
    if (threadIdx.x == 0) *err = 1;
    return;
  }
  const int first = s_info[0], end = s_info[1];
  for (int b0 = 0; b0 < nb; b0 += kSuper) {
    const int b1 = min(nb, b0 + kSuper);
    const int lb = b0 + c_loc;  // local block of this wave
    const int cb = B0 + lb;
    const bool have = lb < b1;
    u64 diag = 0ull, above[kSuper - 1];
    if (have) diag = mask[((size_t)cb * CB) * 64 + lane];
#pragma unroll
    for (int q = 0; q < kSuper - 1; ++q) {
      above[q] = 0ull;
      if (have && q < c_loc) above[q] = mask[((size_t)(B0 + b0 + q) * CB + (cb - (B0 + b0 + q))) * 64 + lane];
    }
    u64 acc = 0ull;
    if (have) {
      for (int rl = 0; rl < b0; ++rl) {
        const u64 w = mask[((size_t)(B0 + rl) * CB + (cb - (B0 + rl))) * 64 + lane];
        if ((s_keepbits[rl] >> lane) & 1ull) acc |= w;
      }
    }
    u64 rem = b0 > 0 ? wave_or64(acc) : 0ull;
    const int rows_here = have ? min(64, n - cb * 64) : 0;
    const u64 valid = rows_here >= 64 ? ~0ull : ((1ull << rows_here) - 1ull);
#pragma unroll
    for (int step = 0; step < kSuper; ++step) {
      if (step == c_loc && have) {
        u64 r = uniform64(rem);
        u64 active = uniform64(__ballot(diag != 0ull)) & ~r & valid;
        while (active) {
          const int k = __builtin_ctzll(active);
          r |= readlane64(diag, k);
          active &= ~(r | (1ull << k));
        }
        if (lane == 0) s_keepbits[lb] = ~r & valid;
      }
      __syncthreads();
      if (step < kSuper - 1 && have && step < c_loc) {
        const u64 kb = s_keepbits[b0 + step];
        const u64 contrib = ((kb >> lane) & 1ull) ? above[step] : 0ull;
        rem |= wave_or64(contrib);
      }
    }
  }
  // publish the keep bits of the rows this workgroup owns: positions [first, end)
  for (int lb = threadIdx.x; lb < nb; lb += kSuper * kWave) {
    const int p0 = (B0 + lb) * 64;
    u64 own = ~0ull;
    if (first > p0) own &= ~((1ull << (first - p0)) - 1ull);         // first - p0 < 64: only in the first block
    if (end < p0 + 64) own &= (1ull << (end - p0)) - 1ull;           // end - p0 >= 1
    const u64 bits = s_keepbits[lb] & own;
    if (bits) atomicOr(&keepbits[B0 + lb], bits);
  }
}

// emission in GLOBAL score order: g -> position -> keep bit, two-pass compaction
__device__ __forceinline__ bool seg_kept(const u64* __restrict__ keepbits, const int* __restrict__ invperm, int g, int n) {
  if (g >= n) return false;
  const int p = invperm[g];
  return (keepbits[p >> 6] >> (p & 63)) & 1ull;
}

__global__ __launch_bounds__(1024) void nms_seg_count(const u64* __restrict__ keepbits, const int* __restrict__ invperm, int n,
                                                      const int64_t* __restrict__ n_dev, int* __restrict__ counts) {
  __shared__ int s_w[16];
  n = live_boxes(n, n_dev);
  const int g = blockIdx.x * 1024 + threadIdx.x;
  const u64 bal = __ballot(seg_kept(keepbits, invperm, g, n));
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < 16; ++w) t += s_w[w];
    counts[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(1024) void nms_seg_emit(const u64* __restrict__ keepbits, const int* __restrict__ invperm,
                                                     const int64_t* __restrict__ order, const int* __restrict__ counts, int n,
                                                     const int64_t* __restrict__ n_dev, const int* __restrict__ err,
                                                     const int* __restrict__ ext_err, int64_t* __restrict__ keep_out,
                                                     int64_t* __restrict__ num_keep) {
  __shared__ int s_w[16];
  __shared__ int s_part[16];
  n = live_boxes(n, n_dev);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // exclusive prefix of the chunk counts before this chunk
  int part = 0;
  for (int c = threadIdx.x; c < (int)blockIdx.x; c += 1024) part += counts[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  if (lane == 0) s_part[wave] = part;
  const int g = blockIdx.x * 1024 + threadIdx.x;
  const bool kept = seg_kept(keepbits, invperm, g, n);
  const u64 bal = __ballot(kept);
  if (lane == 0) s_w[wave] = __popcll(bal);
  __syncthreads();
  int base = 0;
  for (int w = 0; w < 16; ++w) base += s_part[w];
  int before = 0;
  for (int w = 0; w < wave; ++w) before += s_w[w];
  if (kept) keep_out[base + before + __popcll(bal & ((1ull << lane) - 1ull))] = order[g];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    int tot = base;
    for (int w = 0; w < 16; ++w) tot += s_w[w];
    *num_keep = (*err || (ext_err && *ext_err)) ? -1 : tot;   // ext_err: the partition saw an id outside its promised range
  }
}


// ---------------------------------------------------------------------------------------
// Small batched NMS (n <= 4096 boxes, ids in [0, S), every segment <= 1024 boxes: the RPN /
// box-head / per-image sizes of a detector step) in TWO launches and no second sort:
//  A. nms_small_seg_tiles: workgroup (s, y) collects segment s from the score-ordered list
//     (stable; every workgroup of the segment repeats this cheap scan), and each of its waves
//     builds one upper-triangular 64x64 suppression tile of the segment -> tiles[s][t][64]
//     (136 tiles per segment at most, spread over 9 workgroups: the pair tests use the chip,
//     not one CU per segment); workgroup (s, 0) also stores the segment's global ranks.
//  B. nms_small_seg_sweep: workgroup s sweeps its segment (one super-block: no pull phase),
//     publishes one flag per box, and the LAST workgroup to finish (device-scope fence +
//     ticket) compacts the flags in global score order into keep_out / num_keep.
// Versus mask kernel + global sweep: no N x N mask, S resolve chains run concurrently.
constexpr int kSmallSegBoxes = 1024;
constexpr int kSmallSegBlocks = kSmallSegBoxes / 64;                          // 16 = kSuper
constexpr int kSmallSegTiles = kSmallSegBlocks * (kSmallSegBlocks + 1) / 2;  // 136

struct SmallSegWorkspace {
  int* sync_words;       // [0] finished workgroups, [1] error flag
  int* gcnt;             // [S] boxes per segment
  int* glist;            // [S][1024] global rank of the i-th box of the segment
  int* flags;            // [n] keep flag per global rank
  u64* tiles;            // [S][136][64]
};
inline size_t small_seg_workspace_layout(int64_t n, int64_t S, char* base, SmallSegWorkspace* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* ptr = base ? base + off : nullptr;
    off += (bytes + 255) & ~(size_t)255;
    return ptr;
  };
  char* sw = take(2 * sizeof(int));
  char* gc = take((size_t)S * sizeof(int));
  char* gl = take((size_t)S * kSmallSegBoxes * sizeof(int));
  char* fl = take((size_t)n * sizeof(int));
  char* tl = take((size_t)S * kSmallSegTiles * 64 * sizeof(u64));
  if (w) {
    w->sync_words = reinterpret_cast<int*>(sw);
    w->gcnt = reinterpret_cast<int*>(gc);
    w->glist = reinterpret_cast<int*>(gl);
    w->flags = reinterpret_cast<int*>(fl);
    w->tiles = reinterpret_cast<u64*>(tl);
  }
  return off;
}

// Step A as TWO launches whose workgroups fit next to a chip-filling neighbour (round 5; the one-launch form that staged a whole
// segment in 21 KB of LDS per 1024-lane workgroup — it waited 100-195 us in the dispatcher under the RoIAlign forward, which owns
// 156 of the 160 KB of every CU — was deleted in round 6).  A1: nms_small_seg_collect records the segment's members (global ranks, in score order) and
// its size — 64 bytes of LDS.  B: nms_small_seg_tiles4, 256-lane workgroups of four tiles: a wave fetches its 64 column boxes
// straight into registers (rank -> order -> box), the at most three row blocks of a workgroup's four consecutive tiles are
// staged once in 3.75 KB of LDS.  Same pair arithmetic (suppression_tile), same tiles in ws.tiles, the sweep is unchanged.
template <typename T>
__global__ __launch_bounds__(1024) void nms_small_seg_collect(const int64_t* __restrict__ order, const int64_t* __restrict__ seg, int n,
                                                              const int64_t* __restrict__ n_dev, int S, SmallSegWorkspace ws) {
  __shared__ int s_wcnt[16];
  n = live_boxes(n, n_dev);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int me = blockIdx.x;
  int* glist = ws.glist + (size_t)me * kSmallSegBoxes;
  int cnt = 0;
  bool bad = false;
  for (int base = 0; base < n; base += 1024) {
    const int g = base + tid;
    bool mine = false;
    if (g < n) {
      const int64_t sg = seg[order[g]];
      mine = sg == me;
      bad |= sg < 0 || sg >= S;
    }
    const u64 bal = __ballot(mine);
    if (lane == 0) s_wcnt[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
      const int c = s_wcnt[w];
      before += w < wave ? c : 0;
      total += c;
    }
    const int pos = cnt + before + __popcll(bal & ((1ull << lane) - 1ull));
    if (mine && pos < kSmallSegBoxes) glist[pos] = g;
    cnt += total;
    __syncthreads();
  }
  const bool any_bad = me == 0 ? __syncthreads_or(bad) : false;
  if (tid == 0) {
    if (cnt > kSmallSegBoxes || any_bad) ws.sync_words[1] = 1;
    ws.gcnt[me] = min(cnt, kSmallSegBoxes);
  }
}

constexpr int kTiles4 = 4;   // tiles (= waves) per workgroup of nms_small_seg_tiles4

template <typename T>
__global__ __launch_bounds__(kTiles4 * 64) void nms_small_seg_tiles4(const T* __restrict__ dets, const int64_t* __restrict__ order,
                                                                     double thr, ThrBand band, SmallSegWorkspace ws) {
  __shared__ __attribute__((aligned(16))) T s_row[3][5][64];   // the row blocks of this workgroup's tiles, component-major
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int me = blockIdx.x;
  const int cnt = ws.gcnt[me];
  const int nb = (cnt + 63) >> 6, ntiles = nb * (nb + 1) / 2;
  const int t0 = blockIdx.y * kTiles4;
  if (t0 >= ntiles) return;   // the whole workgroup: the grid is sized for the longest possible segment
  const int* glist = ws.glist + (size_t)me * kSmallSegBoxes;
  auto row_of = [&](int t, int& rb, int& cb) {   // tile index -> (row block, column block) of the upper triangle, row-major
    rb = 0;
    int rem = t;
    while (rem >= nb - rb) {
      rem -= nb - rb;
      ++rb;
    }
    cb = rb + rem;
  };
  int rb0, cb0;
  row_of(t0, rb0, cb0);
  // four consecutive tiles span at most three row blocks (a row of the triangle has at least one tile, the last rows 3, 2, 1)
  if (wave < 3 && rb0 + wave < nb) {
    const int p = (rb0 + wave) * 64 + lane;
    T x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    if (p < cnt) {
      const Box<T> b = load_box<T>(dets, order[glist[p]]);
      x1 = b.x1;
      y1 = b.y1;
      x2 = b.x2;
      y2 = b.y2;
    }
    s_row[wave][0][lane] = x1;
    s_row[wave][1][lane] = y1;
    s_row[wave][2][lane] = x2;
    s_row[wave][3][lane] = y2;
    s_row[wave][4][lane] = (x2 - x1) * (y2 - y1);
  }
  __syncthreads();
  const int t = t0 + wave;
  if (t >= ntiles) return;
  int rb, cb;
  row_of(t, rb, cb);
  const int j = cb * 64 + lane;
  const bool jvalid = j < cnt;
  T jx1 = 0, jy1 = 0, jx2 = 0, jy2 = 0;
  {
    const Box<T> b = load_box<T>(dets, order[glist[jvalid ? j : 0]]);
    jx1 = b.x1;
    jy1 = b.y1;
    jx2 = b.x2;
    jy2 = b.y2;
  }
  const T jarea = (jx2 - jx1) * (jy2 - jy1);
  const u64 mine = suppression_tile<T, 64>(&s_row[rb - rb0][0][0], nullptr, min(64, cnt - rb * 64), jx1, jy1, jx2, jy2, jarea, 0,
                                           jvalid, cb == rb, thr, band);
  ws.tiles[((size_t)me * kSmallSegTiles + t) * 64 + lane] = mine;
}

__global__ __launch_bounds__(kSuper * kWave) void nms_small_seg_sweep(const int64_t* __restrict__ order, int n,
                                                                      const int64_t* __restrict__ n_dev,
                                                                      SmallSegWorkspace ws, int64_t* __restrict__ keep_out,
                                                                      int64_t* __restrict__ num_keep) {
  static_assert(kSmallSegBlocks == kSuper, "one super-block per segment");
  n = live_boxes(n, n_dev);
  __shared__ u64 s_keepbits[kSmallSegBlocks];
  __shared__ int s_wcnt[16];
  __shared__ int s_flag;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int me = blockIdx.x;
  const int cnt = ws.gcnt[me];
  const int nb = (cnt + 63) >> 6;
  const u64* tiles = ws.tiles + (size_t)me * kSmallSegTiles * 64;
  {
    auto tile_index = [nb](int rb, int cb) { return rb * nb - (rb * (rb - 1)) / 2 + (cb - rb); };
    const int c_loc = wave;
    const bool have = c_loc < nb;
    u64 diag = 0ull, above[kSuper - 1];
    if (have) diag = tiles[tile_index(c_loc, c_loc) * 64 + lane];
#pragma unroll
    for (int q = 0; q < kSuper - 1; ++q) {
      above[q] = 0ull;
      if (have && q < c_loc) above[q] = tiles[tile_index(q, c_loc) * 64 + lane];
    }
    u64 rem = 0ull;
    const int rows_here = have ? min(64, cnt - c_loc * 64) : 0;
    const u64 valid = rows_here >= 64 ? ~0ull : ((1ull << rows_here) - 1ull);
#pragma unroll
    for (int step = 0; step < kSuper; ++step) {
      if (step == c_loc && have) {
        u64 r = uniform64(rem);
        u64 active = uniform64(__ballot(diag != 0ull)) & ~r & valid;
        while (active) {
          const int k = __builtin_ctzll(active);
          r |= readlane64(diag, k);
          active &= ~(r | (1ull << k));
        }
        if (lane == 0) s_keepbits[c_loc] = ~r & valid;
      }
      __syncthreads();
      if (step < kSuper - 1 && have && step < c_loc) {
        const u64 kb = s_keepbits[step];
        const u64 contrib = ((kb >> lane) & 1ull) ? above[step] : 0ull;
        rem |= wave_or64(contrib);
      }
    }
  }
  // one flag per box of my segment, at its global rank
  const int* glist = ws.glist + (size_t)me * kSmallSegBoxes;
  // Published with agent-scope (write-through) stores and read back with agent-scope loads instead of release /
  // acquire fences: on the 8-XCD part a device-scope fence is a write-back / invalidate of the whole L2 slice,
  // which costs more than this kernel's actual work.
  for (int i = tid; i < cnt; i += kSuper * kWave)
    __hip_atomic_store(&ws.flags[glist[i]], (int)((s_keepbits[i >> 6] >> (i & 63)) & 1ull), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  // last workgroup out compacts the flags in global score order
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my flag stores are acknowledged
  __syncthreads();
  if (tid == 0) s_flag = atomicAdd(&ws.sync_words[0], 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_flag) return;
  int run = 0;
  for (int base = 0; base < n; base += 1024) {
    const int g = base + tid;
    const bool kept = g < n && __hip_atomic_load(&ws.flags[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    const u64 bal = __ballot(kept);
    if (lane == 0) s_wcnt[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
      const int c = s_wcnt[w];
      before += w < wave ? c : 0;
      total += c;
    }
    if (kept) keep_out[run + before + __popcll(bal & ((1ull << lane) - 1ull))] = order[g];
    run += total;
    __syncthreads();
  }
  if (tid == 0) *num_keep = *(volatile int*)&ws.sync_words[1] ? -1 : run;
}


// ---------------------------------------------------------------------------------------
// Score order for small inputs (n <= 4096, float32): order = aten::sort(scores, stable=True, descending=True)
// indices — NaN first, ties by ascending index, -0 == +0 — from 64-bit keys (order-preserving score bits << 32 | index): the
// unique index in the low word makes the rank of a key exact.  At these sizes torch's path is a radix-sort kernel plus an index
// arange plus two copies (~40 us of launches for 4000 scores).
constexpr int kSortMax = 4096;

// (ascending key) == (descending score, ties by ascending index): NaN is the greatest value for aten::sort, -0 == +0
__device__ __forceinline__ u64 score_key(float f, int i) {
  unsigned b = __builtin_bit_cast(unsigned, f);
  unsigned d;
  if (f != f) {
    d = 0u;
  } else {
    if (f == 0.f) b = 0u;
    const unsigned asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    d = ~asc;
    if (d == 0u) d = 1u;  // cannot happen for non-NaN values (asc of +inf is 0xFF800000), kept for safety
  }
  return ((u64)d << 32) | (unsigned)i;
}

// RANK COUNTING (round 5; the one-workgroup bitonic network over 32 KB of LDS it replaced was deleted in round 6): the keys are
// unique, so the position of element i is the number of keys below its own — n^2 compares (16 M at n = 4096) spread over n / 64
// workgroups instead of 78 compare-exchange passes of ONE workgroup.  Lane = element, the four waves of a workgroup count over one quarter of the keys
// each, 64 keys at a time through a 512-byte wave-private LDS slice (built once per 64 compares, read as broadcasts).  3 KB of
// LDS and 256 lanes per workgroup: it starts on any CU, also next to a launch that owns the CU's LDS (the bitonic kernel's 32 KB
// do not; HISTORY.md 6.0), and it is faster on an idle chip as well.  Identical output (the rank of a unique key is exact).
__global__ __launch_bounds__(256) void sort_scores_desc_rank(const float* __restrict__ scores, int n, int64_t* __restrict__ order) {
  __shared__ u64 s_keys[4][64];
  __shared__ int s_cnt[4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i = blockIdx.x * 64 + lane;
  const u64 ki = i < n ? score_key(scores[i], i) : ~0ull;
  const int per = ((n + 255) >> 8) << 6;            // keys per quarter: a multiple of 64
  const int j_begin = wave * per, j_end = min(n, j_begin + per);
  int cnt = 0;
  for (int j0 = j_begin; j0 < j_end; j0 += 64) {
    const int j = j0 + lane;
    s_keys[wave][lane] = j < n ? score_key(scores[j], j) : ~0ull;   // padding: above every real key, never counted
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < 64; ++t) cnt += s_keys[wave][t] < ki ? 1 : 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  s_cnt[wave][lane] = cnt;
  __syncthreads();
  if (wave == 0 && i < n) order[s_cnt[0][lane] + s_cnt[1][lane] + s_cnt[2][lane] + s_cnt[3][lane]] = i;
}

// ---------------------------------------------------------------------------------------
// Detector-step batched NMS (+ the padded top-k payload) as ONE launch (round 6).
// The chain it replaces for n <= 4096 boxes in <= 64 segments of <= 1024 boxes: memset, score sort -> collect -> tiles4 -> sweep
// (-> pack_detections): 5 kernels + a fill, 49 us of kernel time + 6 launch gaps.  Here the same steps are PHASES of one grid of
// 256-lane workgroups handed over through memory words (agent-scope stores / loads, the hand-off idiom of nms_push above):
// tile workgroups (segment s = blockIdx.x, j = blockIdx.y < G) and counting-only workgroups (j >= G).
//   A1  every tile workgroup: members of segment s as keys (score word << 32 | box index) in LDS, ascending box index — all G
//       workgroups of a segment build the same list (redundant work on an idle chip instead of a publish / poll round trip).
//   A3  workgroups j < 16: score order of the segment by RANK COUNTING — 64 members per workgroup (lane = member), its four waves
//       count over a quarter of the segment's keys each; the member's rank r is its place: box and index go to sbox[s][r] / sidx[s][r].
//   A2  counting-only workgroups: partial GLOBAL score ranks (over all n boxes, for the output order) — 64 boxes x a quarter of the
//       keys per workgroup, combined in LDS: part[q][box], q < 4.  16M key pairs at n = 4000: every SIMD of the chip counts.
//   B   every tile workgroup: waits for the publishers' flags; upper-triangular 64x64 suppression tiles t = 4j + wave (+ 4G ...) from
//       the sorted boxes — suppression_tile, the bit-exact reference predicate — to tiles[s][t]; raises its flag.
//   C   workgroup (s, 0): waits for the G tile flags; greedy sweep of the <= 16 blocks with 4 waves (wave w owns column blocks
//       w, w + 4, ...: the tiles above them in registers, a block's pending removals OR-reduced once, when its turn comes); waits
//       for the counting flags; slot[global rank] = kept ? (index + 1) | image << 13 : 0 for ITS boxes; takes a ticket.
//   D   the LAST sweeper compacts slot[] in rank order into keep_out / num_keep — the reference's output order: descending score
//       over ALL segments — writes the per-image padded top-`max_dets` rows (what tvmi_pack_detections_payload writes) in the same
//       pass, and puts the hand-over words back to zero.
// Counters shared by many producers are avoided (one flag word per producer): same-address agent-scope atomics retire ~150 ns
// apart.  Counting compares 32-bit score words with saturating subtracts (count_below_group): the condition-code round trip of a
// v_cmp -> v_addc chain cost ~45 cycles per key with one wave per SIMD.  Phase times on the MI355X at 4 x 1000 boxes (s_memrealtime,
// -DTVMI_STEP_TIMING): A1 3.9, A3 5.3, wait 2, B 5.6, tile loads 1, C 4.6, slots 4.3, ticket 1.5, D 11 = 39 us in ONE launch,
// against 49 us of kernels + 4.5 us fill + 6 gaps for the chain.
// Forward progress: workgroup (s, 0) and the publishers have the lowest workgroup ids of a launch capped at 512 workgroups (a
// fraction of the chip's resident capacity), so a poller never keeps its producer off the chip; every poll is bounded and a
// timeout surfaces as num_keep = -1 (as do an over-long segment and an id outside [0, S)).
constexpr int kStepThreads = 256;
constexpr int kStepMaxSegments = 64;
constexpr int kStepMaxGroups = 512;
constexpr int kStepMaxImages = 16;
// hand-over words of a launch (ints): [0] sweepers finished (ticket), [1] input error, [2] a poll gave up (never cleared),
// then ONE FLAG PER PRODUCER — publishers [S][16], tile workgroups [S * G], counting workgroups — instead of shared counters:
// agent-scope atomics on one address retire one after the other (~150 ns each on the MI355X: 250 counting workgroups reporting
// to one counter took 35 us), flag stores go out in parallel and a poll is one load per lane.
constexpr int kStepFlagsPub = 4;
constexpr int kStepFlagsTile = kStepFlagsPub + kStepMaxSegments * kSmallSegBlocks;
constexpr int kStepFlagsRank = kStepFlagsTile + kStepMaxGroups;
constexpr int kStepSyncWords = kStepFlagsRank + kStepMaxGroups;

struct StepWorkspace {
  int* sync;     // kStepSyncWords ints, all zero when the launch starts; the last sweeper re-zeroes them (see step_sync_block)
  int* sidx;     // [S][1024] box index of the p-th best box of segment s
  int* part;     // [4][n] partial global ranks
  int* slot;     // [n] by global rank: 0, or (kept box index + 1) | image << 13
  u64* tiles;    // [S][136][64]
  u64* sbox;     // [S][1024][2] the boxes of a segment in score order
};
inline size_t step_workspace_layout(int64_t n, int64_t S, char* base, StepWorkspace* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* ptr = base ? base + off : nullptr;
    off += (bytes + 255) & ~(size_t)255;
    return ptr;
  };
  char* sy = take((size_t)kStepSyncWords * sizeof(int));
  char* si = take((size_t)S * kSmallSegBoxes * sizeof(int));
  char* pa = take((size_t)4 * n * sizeof(int));
  char* sl = take((size_t)n * sizeof(int));
  char* tl = take((size_t)S * kSmallSegTiles * 64 * sizeof(u64));
  char* sb = take((size_t)S * kSmallSegBoxes * 2 * sizeof(u64));
  if (w) {
    w->sync = reinterpret_cast<int*>(sy);
    w->sidx = reinterpret_cast<int*>(si);
    w->part = reinterpret_cast<int*>(pa);
    w->slot = reinterpret_cast<int*>(sl);
    w->tiles = reinterpret_cast<u64*>(tl);
    w->sbox = reinterpret_cast<u64*>(sb);
  }
  return off;
}
inline int step_groups_per_segment(int64_t n, int64_t S) {
  const int nbmax = (int)std::min<int64_t>(kSmallSegBlocks, ceil_div(n, 64));
  const int want = (int)ceil_div(nbmax * (nbmax + 1) / 2, 4);
  return (int)std::max<int64_t>(1, std::min<int64_t>(want, kStepMaxGroups / S));
}

// The hand-over words of the step kernel live in a block the LIBRARY owns, one per (device, stream): zero when it is created,
// and put back to zero by the last workgroup of every launch — launches on one stream run one after the other, so the next one
// finds it clean and the call needs no memset (a fill kernel of its own: 4.5 us on the MI355X, more than any phase of the step).
// A stream that is being captured into a graph gets the caller's workspace words + a memset node instead (no allocation inside a
// capture).  A launch that gave up on a poll leaves the error word set: every later call on that stream reports num = -1.
struct StepSyncBlocks {
  std::mutex mu;
  std::vector<std::pair<std::pair<int, hipStream_t>, int*>> blocks;
  int* get(hipStream_t s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    for (auto& b : blocks)
      if (b.first.first == dev && b.first.second == s) return b.second;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return nullptr;
    }
    int* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), kStepSyncWords * sizeof(int)) != hipSuccess ||
        hipMemset(p, 0, kStepSyncWords * sizeof(int)) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    blocks.push_back({{dev, s}, p});
    return p;
  }
};
inline StepSyncBlocks& step_sync_blocks() {
  static StepSyncBlocks* b = new StepSyncBlocks();   // never destroyed: the runtime may be gone at exit
  return *b;
}

__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_agent64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent64(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every flag of flags[0 .. count) is non-zero: the whole workgroup polls (one load per lane and round), bounded like
// poll_until_equal; returns with the give-up word raised when the producers do not show up
__device__ __forceinline__ void poll_flags(const int* flags, int count, int* gave_up) {
  for (int polls = 0;; ++polls) {
    bool ok = true;
    for (int i = threadIdx.x; i < count; i += blockDim.x) ok = ok && ld_agent(&flags[i]) != 0;
    if (__syncthreads_and(ok)) return;
    __builtin_amdgcn_s_sleep(4);
    if ((polls & 255) == 255 && ld_agent(gave_up) != 0) return;
    if (polls > (1 << 18)) {
      if (threadIdx.x == 0) st_agent(gave_up, 1);
      return;
    }
  }
}

struct StepPack {   // optional payload of the last phase (payload == nullptr: keep list only)
  const int64_t* image_idx;
  const int64_t* labels;
  float* payload;
  int32_t* counts;
  int64_t row_stride;
  int num_images, max_dets, count_in_row;
};

// tiles above the column blocks of one sweep wave: block w + 4k has w + 4k earlier blocks, at most 3 + 4k
template <int K>
struct StepAbove {
  u64 v[4 * K + 3];
};

// number of the 64 keys held one per lane in `kk` that are below `ki` (per lane): the keys are read lane by lane into SGPRs
// (v_readlane) — no LDS round trip, no barrier
// Number of the 64 score words held one per lane in `dk` (a group of 64 consecutive boxes) that rank before the lane's own box
// (score word `di`).  Keys are (score word, box index) pairs and the groups are in index order, so against a whole group only the
// score words need comparing: `rel` < 0 — the group's boxes all have a smaller index, a tie counts (d <= di, i.e. d < di + 1: no
// live score word is 0xffffffff); `rel` > 0 — a larger index, a tie does not.  One full-rate v_cmp_lt_u32 against an SGPR per key
// (a 64-bit integer compare is several times slower); the loops stay rolled (x8): this code runs once per launch, straight-line
// unrolling of everything made the kernel 130 KB of instructions that were all fetched cold.
// max(a - b, 0) for unsigned a (VGPR), b (wave-uniform): v_sub_u32 with the integer clamp.  Opaque on purpose — written as
// __builtin_elementwise_sub_sat the compiler folds min(sat(thr - d), 1) back into v_cmp_lt + v_cndmask / v_addc.
__device__ __forceinline__ unsigned sub_sat_u32(unsigned a, unsigned b_uniform) {
  unsigned r;
  asm("v_sub_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "s"(b_uniform));
  return r;
}
__device__ __forceinline__ int count_below_group(unsigned dk, unsigned di, int rel) {
  // [d < thr] as min(sat(thr - d), 1): plain VALU results, no VCC / SGPR-pair round trip between the compare and the add (a
  // v_cmp -> v_addc chain stalls on the condition code for every key: ~45 cycles per key measured, one wave per SIMD)
  const unsigned thr = di + (rel < 0 ? 1u : 0u);
  unsigned c0 = 0, c1 = 0;
#pragma unroll 4
  for (int t = 0; t < 64; t += 2) {
    const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)dk, t), d1 = (unsigned)__builtin_amdgcn_readlane((int)dk, t + 1);
    c0 += min(sub_sat_u32(thr, d0), 1u);
    c1 += min(sub_sat_u32(thr, d1), 1u);
  }
  return (int)(c0 + c1);
}
// ... and against the lane's OWN group: key t of the group against lane `lane` — a tie counts when t < lane, i.e. the
// threshold of key t is di + 1 for the lanes above t
__device__ __forceinline__ int count_below_own(unsigned dk, unsigned di) {
  const int lane = threadIdx.x & 63;
  unsigned c0 = 0, c1 = 0;
#pragma unroll 4
  for (int t = 0; t < 64; t += 2) {
    const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)dk, t), d1 = (unsigned)__builtin_amdgcn_readlane((int)dk, t + 1);
    c0 += min(sub_sat_u32(di + (t < lane ? 1u : 0u), d0), 1u);
    c1 += min(sub_sat_u32(di + (t + 1 < lane ? 1u : 0u), d1), 1u);
  }
  return (int)(c0 + c1);
}

#ifdef TVMI_STEP_TIMING
__device__ unsigned long long g_step_stamp[64];
#define STEP_STAMP(slot, cond) do { if ((cond) && threadIdx.x == 0) g_step_stamp[slot] = wall_clock64(); } while (0)
#else
#define STEP_STAMP(slot, cond) do { } while (0)
#endif
__global__ __launch_bounds__(kStepThreads) void nms_step_fused(const float* __restrict__ dets, const float* __restrict__ scores,
                                                               const int64_t* __restrict__ seg, int n, int S, int G, double thr,
                                                               ThrBand band, StepWorkspace ws, int64_t* __restrict__ keep_out,
                                                               int64_t* __restrict__ num_keep, StepPack pk) {
  constexpr int kPer = kSortMax / kStepThreads;                          // boxes per thread of a pass over all n (16)
  __shared__ u64 s_keys[kSmallSegBoxes];                                 // members of my segment (ascending box index)
  __shared__ __attribute__((aligned(16))) float s_row[3][5][64];         // row blocks of a workgroup's four tiles
  __shared__ u64 s_keepbits[kSmallSegBlocks];
  __shared__ int s_pre[kStepMaxImages + 1][kPer * 4];                    // counts -> exclusive prefixes per (round, wave)
  __shared__ int s_tot[kStepMaxImages + 1];
  __shared__ int s_lr[4][64];                                            // local-rank partials of a member group
  __shared__ unsigned s_rk[4][8][64];                                    // global rank counting: eight groups of score words per wave
  __shared__ int s_flag;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int s = blockIdx.x, j = blockIdx.y;
  int* const err = ws.sync + 1;
  int* const gave_up = ws.sync + 2;
  int* const pub_flags = ws.sync + kStepFlagsPub + s * kSmallSegBlocks;   // [16] of my segment
  int* const tile_flags = ws.sync + kStepFlagsTile + s * G;               // [G] of my segment
  int* const rank_flags = ws.sync + kStepFlagsRank;                       // one per counting workgroup of the launch
  const u64 lt_mask = (1ull << lane) - 1ull;
  STEP_STAMP(0, s == 0 && j == 0);

  // The counting is spread over the counting-only workgroups of the launch (blockIdx.y >= G: a launch has up to kStepMaxGroups
  // workgroups, the tile workgroups use S * G of them — 16M key pairs at 4000 boxes want every SIMD of the chip) or, without
  // them, over all workgroups before their tiles.  A counting workgroup takes 64 boxes (lane = box) and a Q-th of the keys, its
  // four waves a quarter of that each; the four counts meet in LDS, so a box has Q <= 4 partials to add up later (16 scattered
  // words per box made the sweepers' gather the longest phase of the kernel).
  const int n_groups = (n + 63) >> 6;
  const int rank_j0 = (int)gridDim.y > G ? G : 0;
  const int n_rank_wgs = ((int)gridDim.y - rank_j0) * S;
  const int Q = min(4, max(1, n_rank_wgs / n_groups));
  const int per = ((n_groups + 4 * Q - 1) / (4 * Q)) << 6;   // keys per wave, a multiple of 64
  auto rank_share = [&]() {
    for (int rw = (j - rank_j0) * S + s; rw < n_groups * Q; rw += n_rank_wgs) {   // workgroup-uniform
      const int eg = rw % n_groups, q = rw / n_groups;
      const int i = eg * 64 + lane;
      const unsigned di = i < n ? (unsigned)(score_key(scores[i], 0) >> 32) : 0xffffffffu;
      const int j_begin = (q * 4 + wave) * per, j_end = min(n, j_begin + per);
      int c = 0;
      for (int jb = j_begin; jb < j_end; jb += 8 * 64) {   // eight 64-key chunks per batch: their loads are in flight together
        float sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int qq = jb + u * 64 + lane;
          sv[u] = scores[min(qq, n - 1)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {                        // parked in LDS so that the counting loop below can stay rolled
          const int qq = jb + u * 64 + lane;
          s_rk[wave][u][lane] = qq < j_end ? (unsigned)(score_key(sv[u], 0) >> 32) : 0xffffffffu;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int u = 0; u < 8 && jb + u * 64 < j_end; ++u) {
          const int chunk = (jb >> 6) + u;
          const unsigned dk = s_rk[wave][u][lane];
          c += chunk == eg ? count_below_own(dk, di) : count_below_group(dk, di, chunk < eg ? -1 : 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      s_lr[wave][lane] = c;
      __syncthreads();
      if (wave == 0 && i < n) st_agent(&ws.part[(size_t)q * n + i], s_lr[0][lane] + s_lr[1][lane] + s_lr[2][lane] + s_lr[3][lane]);
      __syncthreads();
    }
  };
  if (j >= G) {   // counting-only workgroup
    STEP_STAMP(21, s == 0 && j == G);
    STEP_STAMP(23, s == 0 && j == (int)gridDim.y - 1);
    rank_share();
    STEP_STAMP(22, s == 0 && j == G);
    STEP_STAMP(24, s == 0 && j == (int)gridDim.y - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) st_agent(&rank_flags[(j - G) * S + s], 1);
    return;
  }
  // ---- A1: the members of my segment, as keys (score bits << 32 | box index) in ascending box order.  EVERY workgroup of the
  // segment builds the same list (redundant work on an idle chip instead of a publish / poll round trip).
  int cnt;
  {
    int64_t sg[kPer];
    float sc[kPer];
#pragma unroll
    for (int r = 0; r < kPer; ++r) {           // all loads of the pass in flight together
      const int g = min(r * kStepThreads + tid, n - 1);
      sg[r] = seg ? seg[g] : 0;
      sc[r] = scores[g];
    }
    u64 bal[kPer];
    bool bad = false;
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      const int g = r * kStepThreads + tid;
      const bool in = g < n;
      bad |= in && (sg[r] < 0 || sg[r] >= (int64_t)S);
      bal[r] = __ballot(in && sg[r] == (int64_t)s);
      if (lane == 0) s_pre[0][r * 4 + wave] = __popcll(bal[r]);
    }
    __syncthreads();
    if (wave == 0) {
      const int c = s_pre[0][lane];
      int incl = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
      }
      s_pre[1][lane] = incl - c;
      if (lane == 63) s_tot[0] = incl;
    }
    const bool any_bad = __syncthreads_or(bad);
    cnt = s_tot[0];
    if ((any_bad || cnt > kSmallSegBoxes) && tid == 0 && j == 0) st_agent(err, 1);
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      const int g = r * kStepThreads + tid;
      if ((bal[r] >> lane) & 1ull) {
        const int pos = s_pre[1][r * 4 + wave] + __popcll(bal[r] & lt_mask);
        if (pos < kSmallSegBoxes) s_keys[pos] = score_key(sc[r], g);
      }
    }
    cnt = min(cnt, kSmallSegBoxes);
    __syncthreads();
  }
  STEP_STAMP(1, s == 0 && j == 0);
  STEP_STAMP(16, s == 0 && j == 16);
  const int nb = (cnt + 63) >> 6, ntiles = nb * (nb + 1) / 2;
  const int npub = min(G, nb);               // workgroups that publish member groups
  // ---- A3: score order of the segment by rank counting: workgroup j takes the member groups j, j + G, ... (64 members each),
  // its four waves count over a quarter of the segment's keys each; the lane that holds member p then knows its rank r and
  // writes the box (and its index) to position r of the segment's sorted lists.
  for (int mg = j; mg < nb; mg += G) {       // workgroup-uniform
    const int p = mg * 64 + lane;
    const bool have = p < cnt;
    const u64 ki = have ? s_keys[p] : ~0ull;
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have && wave == 0) bx = *reinterpret_cast<const float4*>(dets + (int64_t)(unsigned)ki * 4);   // in flight under the counting
    const int cpw = (nb + 3) >> 2;            // 64-key chunks per wave
    const unsigned di = (unsigned)(ki >> 32);
    int c = 0;
    for (int ch = wave * cpw; ch < min(nb, (wave + 1) * cpw); ++ch) {
      const int q = ch * 64 + lane;
      const unsigned dk = q < cnt ? (unsigned)(s_keys[q] >> 32) : 0xffffffffu;   // padding: above every live score word
      c += ch == mg ? count_below_own(dk, di) : count_below_group(dk, di, ch < mg ? -1 : 1);
    }
    s_lr[wave][lane] = c;
    __syncthreads();
    if (wave == 0 && have) {
      const int r = s_lr[0][lane] + s_lr[1][lane] + s_lr[2][lane] + s_lr[3][lane];
      u64* dst = ws.sbox + ((size_t)s * kSmallSegBoxes + r) * 2;
      st_agent64(dst, ((u64)__builtin_bit_cast(unsigned, bx.y) << 32) | __builtin_bit_cast(unsigned, bx.x));
      st_agent64(dst + 1, ((u64)__builtin_bit_cast(unsigned, bx.w) << 32) | __builtin_bit_cast(unsigned, bx.z));
      st_agent(&ws.sidx[(size_t)s * kSmallSegBoxes + r], (int)(unsigned)ki);
    }
    __syncthreads();
  }
  if (j < npub) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my part of the sorted lists is acknowledged
    __syncthreads();
    if (tid == 0) st_agent(&pub_flags[j], 1);
  }
  STEP_STAMP(2, s == 0 && j == 0);
  if ((int)gridDim.y == G) rank_share();   // no counting-only workgroups in this launch: everybody takes a share first
  STEP_STAMP(3, s == 0 && j == 0);
  STEP_STAMP(17, s == 0 && j == 16);
  // ---- B: suppression tiles of my segment, from the sorted boxes
  poll_flags(pub_flags, npub, gave_up);
  STEP_STAMP(4, s == 0 && j == 0);
  STEP_STAMP(18, s == 0 && j == 16);
  {
    const u64* my_sbox = ws.sbox + (size_t)s * kSmallSegBoxes * 2;
    auto box_at = [&](int p, float& x1, float& y1, float& x2, float& y2) {
      const u64 lo = ld_agent64(&my_sbox[2 * p]), hi = ld_agent64(&my_sbox[2 * p + 1]);
      x1 = __builtin_bit_cast(float, (unsigned)lo);
      y1 = __builtin_bit_cast(float, (unsigned)(lo >> 32));
      x2 = __builtin_bit_cast(float, (unsigned)hi);
      y2 = __builtin_bit_cast(float, (unsigned)(hi >> 32));
    };
    auto row_of = [&](int t, int& rb, int& cb) {   // tile index -> (row block, column block) of the upper triangle, row-major
      rb = 0;
      int rem = t;
      while (rem >= nb - rb) {
        rem -= nb - rb;
        ++rb;
      }
      cb = rb + rem;
    };
    u64* tiles = ws.tiles + (size_t)s * kSmallSegTiles * 64;
    for (int t0 = j * 4; t0 < ntiles; t0 += G * 4) {   // workgroup-uniform
      int rb0, cb0;
      row_of(t0, rb0, cb0);
      const int t = t0 + wave;
      int rb = 0, cb = 0;
      if (t < ntiles) row_of(t, rb, cb);
      // column box of this lane and (waves 0..2) one of the at most three row blocks of the four tiles: loads issued together
      const int c = cb * 64 + lane;
      const bool jvalid = t < ntiles && c < cnt;
      float jx1 = 0, jy1 = 0, jx2 = 0, jy2 = 0, x1 = 0, y1 = 0, x2 = 0, y2 = 0;
      const bool stage = wave < 3 && rb0 + wave < nb;
      const int p = (rb0 + wave) * 64 + lane;
      if (jvalid) box_at(c, jx1, jy1, jx2, jy2);
      if (stage && p < cnt) box_at(p, x1, y1, x2, y2);
      if (stage) {
        s_row[wave][0][lane] = x1;
        s_row[wave][1][lane] = y1;
        s_row[wave][2][lane] = x2;
        s_row[wave][3][lane] = y2;
        s_row[wave][4][lane] = (x2 - x1) * (y2 - y1);
      }
      __syncthreads();
      if (t < ntiles) {
        const float jarea = (jx2 - jx1) * (jy2 - jy1);
        const u64 mine = suppression_tile<float, 64>(&s_row[rb - rb0][0][0], nullptr, min(64, cnt - rb * 64), jx1, jy1, jx2, jy2,
                                                     jarea, 0, jvalid, cb == rb, thr, band);
        st_agent64(&tiles[(size_t)t * 64 + lane], mine);
      }
      __syncthreads();
    }
  }
  STEP_STAMP(19, s == 0 && j == 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my tiles and rank partials are acknowledged
  __syncthreads();
  STEP_STAMP(20, s == 0 && j == 16);
  if (tid == 0) {
    st_agent(&tile_flags[j], 1);
    if ((int)gridDim.y == G) st_agent(&rank_flags[j * S + s], 1);   // this launch has no counting-only workgroups: my share is in
  }
  if (j != 0) return;
  // workgroup (s, 0) sweeps the segment once every tile workgroup has reported
  poll_flags(tile_flags, G, gave_up);
  STEP_STAMP(5, s == 0);

  // ---- C: sweep of my segment (4 waves; wave w owns column blocks w, w + 4, w + 8, w + 12)
  int midx[4];                                    // box index of my members p = tid + 256 r (score order), loaded under the sweep
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int p = tid + kStepThreads * r;
    midx[r] = p < cnt ? ld_agent(&ws.sidx[(size_t)s * kSmallSegBoxes + p]) : 0;
  }
  {
    const u64* tiles = ws.tiles + (size_t)s * kSmallSegTiles * 64;
    auto tile_index = [nb](int rb, int cb) { return rb * nb - (rb * (rb - 1)) / 2 + (cb - rb); };
    u64 diag[4];
    StepAbove<0> a0;
    StepAbove<1> a1;
    StepAbove<2> a2;
    StepAbove<3> a3;
    auto load_col = [&](int k, u64* above, int cap) {
      const int c = wave + 4 * k;
      const bool have = c < nb;
      diag[k] = have ? ld_agent64(&tiles[(size_t)tile_index(c, c) * 64 + lane]) : 0ull;
#pragma unroll
      for (int q = 0; q < 15; ++q)
        if (q < cap) above[q] = (have && q < c) ? ld_agent64(&tiles[(size_t)tile_index(q, c) * 64 + lane]) : 0ull;
    };
    load_col(0, a0.v, 3);
    load_col(1, a1.v, 7);
    load_col(2, a2.v, 11);
    load_col(3, a3.v, 15);
    STEP_STAMP(6, s == 0);
    u64 pend[4] = {0ull, 0ull, 0ull, 0ull};   // per-lane pending removals of my blocks (OR-reduced when the block's turn comes)
#pragma unroll
    for (int step = 0; step < kSmallSegBlocks; ++step) {
      if (step < nb) {   // workgroup-uniform
        const int ow = step & 3, ok = step >> 2;
        if (wave == ow) {
          const int rows_here = min(64, cnt - step * 64);
          const u64 valid = rows_here >= 64 ? ~0ull : ((1ull << rows_here) - 1ull);
          u64 r = wave_or64(pend[ok]);
          const u64 dg = diag[ok];
          u64 active = uniform64(__ballot(dg != 0ull)) & ~r & valid;
          while (active) {
            const int k = __builtin_ctzll(active);
            r |= readlane64(dg, k);
            active &= ~(r | (1ull << k));
          }
          if (lane == 0) s_keepbits[step] = ~r & valid;
        }
        __syncthreads();
        const bool kept_lane = (s_keepbits[step] >> lane) & 1ull;
        // rows of block `step` that were kept push their suppression words to my later column blocks
        if (step < 3) pend[0] |= kept_lane ? a0.v[step < 3 ? step : 0] : 0ull;
        if (step < 7) pend[1] |= kept_lane ? a1.v[step < 7 ? step : 0] : 0ull;
        if (step < 11) pend[2] |= kept_lane ? a2.v[step < 11 ? step : 0] : 0ull;
        if (step < 15) pend[3] |= kept_lane ? a3.v[step < 15 ? step : 0] : 0ull;
      }
    }
  }
  __syncthreads();
  STEP_STAMP(7, s == 0);
  // ---- slots of my boxes, by global rank (needs the rank partials of every workgroup of the launch)
  poll_flags(rank_flags, n_rank_wgs, gave_up);
  STEP_STAMP(14, s == 0);
  {
    const bool pack = pk.payload != nullptr;
    int g[4] = {0, 0, 0, 0}, img[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool have = tid + kStepThreads * r < cnt;
#pragma unroll
      for (int q = 0; q < 4; ++q)                         // all loads in flight together
        if (q < Q && have) g[r] += ld_agent(&ws.part[(size_t)q * n + midx[r]]);
      if (pack && have) img[r] = (int)pk.image_idx[midx[r]];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = tid + kStepThreads * r;
      if (p < cnt) {
        const bool kept = (s_keepbits[p >> 6] >> (p & 63)) & 1ull;
        const int im = (img[r] >= 0 && img[r] < kStepMaxImages) ? img[r] : kStepMaxImages;   // outside the payload: no row
        st_agent(&ws.slot[min(max(g[r], 0), n - 1)], kept ? ((midx[r] + 1) | (im << 13)) : 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  STEP_STAMP(8, s == 0);
  if (tid == 0) s_flag = __hip_atomic_fetch_add(&ws.sync[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1;
  __syncthreads();
  if (!s_flag) return;
  STEP_STAMP(9, true);

  // ---- D: last sweeper out — keep list in global score order (+ payload).  Round r covers the ranks [256 r, 256 r + 256); all
  // slot words are loaded together, the ballots of the 16 rounds go to LDS as counts per (round, wave), one wave-scan per list
  // turns them into exclusive prefixes, and the stores follow with one more round of loads (the payload fields).
  const bool pack = pk.payload != nullptr;
  const bool failed = ld_agent(err) != 0 || ld_agent(gave_up) != 0;
  const int B = pack ? pk.num_images : 0;
  int run = 0;
  if (!failed) {
    int v[kPer];
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      const int g = r * kStepThreads + tid;
      v[r] = g < n ? ld_agent(&ws.slot[g]) : 0;
    }
    STEP_STAMP(11, v[0] != 0x7fffffff);
    u64 bal[kPer], ibal[kPer];
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      const int img = v[r] >> 13;
      bal[r] = __ballot(v[r] != 0);
      ibal[r] = 0ull;
      if (lane == 0) s_pre[kStepMaxImages][r * 4 + wave] = __popcll(bal[r]);
      for (int b = 0; b < B; ++b) {
        const u64 bb = __ballot(v[r] != 0 && img == b);
        if (lane == 0) s_pre[b][r * 4 + wave] = __popcll(bb);
        if (img == b) ibal[r] = bb;
      }
    }
    __syncthreads();
    // exclusive scan of the 64 (round, wave) counts of every list: lane = entry, wave w takes lists w, w + 4, ...
    for (int l = wave; l <= kStepMaxImages; l += 4) {
      if (l < B || l == kStepMaxImages) {
        const int c = s_pre[l][lane];
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int o = __shfl_up(incl, d);
          if (lane >= d) incl += o;
        }
        s_pre[l][lane] = incl - c;
        if (lane == 63) s_tot[l] = incl;
      }
    }
    __syncthreads();
    STEP_STAMP(12, true);
    run = s_tot[kStepMaxImages];
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      if (v[r] != 0) {
        const int src = (v[r] & 0x1fff) - 1, img = v[r] >> 13;
        keep_out[s_pre[kStepMaxImages][r * 4 + wave] + __popcll(bal[r] & lt_mask)] = (int64_t)src;
        if (img < B) {
          const int rr = s_pre[img][r * 4 + wave] + __popcll(ibal[r] & lt_mask);
          if (rr < pk.max_dets) {
            float* d = pk.payload + (int64_t)img * pk.row_stride + (int64_t)rr * 6;
            const float4 bx = *reinterpret_cast<const float4*>(dets + (int64_t)src * 4);
            d[0] = bx.x;
            d[1] = bx.y;
            d[2] = bx.z;
            d[3] = bx.w;
            d[4] = scores[src];
            d[5] = pk.labels ? (float)pk.labels[src] : 0.f;
          }
        }
      }
    }
  }
  STEP_STAMP(13, true);
  if (tid == 0) *num_keep = failed ? -1 : run;
  for (int i = run + tid; i < n; i += kStepThreads) keep_out[i] = 0;   // the tail behind num: a deterministic output
  for (int b = 0; b < B; ++b) {
    const int c = failed ? 0 : min(s_tot[b], pk.max_dets);
    float* my = pk.payload + (int64_t)b * pk.row_stride;
    for (int i = c * 6 + tid; i < pk.max_dets * 6; i += kStepThreads) my[i] = 0.f;
    if (tid == 0) {
      const int cn = failed ? -1 : c;
      if (pk.counts) pk.counts[b] = cn;
      if (pk.count_in_row) my[(int64_t)pk.max_dets * 6] = (float)cn;
    }
  }
  // nobody reads the hand-over words any more (every workgroup has passed its last poll: the tickets say so): clean for the
  // next launch on this stream.  The word of a poll that gave up stays: such a stream keeps answering num = -1.
  for (int i = tid; i < kStepSyncWords; i += kStepThreads)
    if (i != 2) st_agent(&ws.sync[i], 0);
  STEP_STAMP(10, true);
}

struct SegWorkspace {
  u64* mask;
  u64* keepbits;
  int64_t* oidx;
  int* invperm;
  int* counts;
  int* err;
};
inline size_t seg_workspace_layout(int64_t n, char* base, SegWorkspace* w) {
  const size_t CB = (size_t)ceil_div(n, 64), NC = (size_t)ceil_div(n, 1024);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* ptr = base ? base + off : nullptr;
    off += (bytes + 255) & ~(size_t)255;
    return ptr;
  };
  char* m = take(CB * std::min<size_t>(CB, (size_t)kSegMaxBlocks) * 64 * sizeof(u64));  // banded: tile (rb, cb) at row rb, slot cb - rb
  char* kb = take((CB + 1) * sizeof(u64));  // keep bits + (err, pad) right behind them: one memset
  char* oi = take((size_t)n * sizeof(int64_t));
  char* ip = take((size_t)n * sizeof(int));
  char* ct = take(NC * sizeof(int));
  if (w) {
    w->mask = reinterpret_cast<u64*>(m);
    w->keepbits = reinterpret_cast<u64*>(kb);
    w->err = reinterpret_cast<int*>(kb + CB * sizeof(u64));
    w->oidx = reinterpret_cast<int64_t*>(oi);
    w->invperm = reinterpret_cast<int*>(ip);
    w->counts = reinterpret_cast<int*>(ct);
  }
  return off;
}

template <typename T>
int launch_seg(const void* dets, const int64_t* order, const int64_t* keys, const int64_t* perm, int64_t n,
               const int64_t* n_dev, const int* ext_err, double thr, void* workspace, int64_t* keep_out, int64_t* num_keep,
               hipStream_t stream) {
  const int CB = (int)ceil_div(n, 64), NC = (int)ceil_div(n, 1024);
  // the mask is banded: a row block only meets column blocks of its own segments, at most kSegMaxBlocks ahead, so a row
  // of tiles is min(CB, kSegMaxBlocks) slots wide (n x 1 KB instead of n^2 / 8 bytes: 0.37 GB instead of 16 GB at 360k boxes)
  const int W = std::min(CB, kSegMaxBlocks);
  SegWorkspace w;
  seg_workspace_layout(n, static_cast<char*>(workspace), &w);
  hipError_t e = hipMemsetAsync(w.keepbits, 0, sizeof(u64) * ((size_t)CB + 1), stream);
  if (e != hipSuccess) return set_error((int)e, "tvmi_nms_segmented: memset");
  nms_seg_layout<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream>>>(order, perm, (int)n, n_dev, w.oidx, w.invperm);
  nms_mask_tiles_seg<T><<<dim3((unsigned)CB), dim3(kMaskWaves * kWave), 0, stream>>>(static_cast<const T*>(dets), w.oidx, keys,
                                                                                  (int)n, n_dev, W, thr, thr_band(thr), w.mask);
  nms_sweep_seg<<<dim3((unsigned)CB), dim3(kSuper * kWave), 0, stream>>>(w.mask, keys, (int)n, n_dev, W, w.keepbits, w.err);
  nms_seg_count<<<dim3((unsigned)NC), dim3(1024), 0, stream>>>(w.keepbits, w.invperm, (int)n, n_dev, w.counts);
  nms_seg_emit<<<dim3((unsigned)NC), dim3(1024), 0, stream>>>(w.keepbits, w.invperm, order, w.counts, (int)n, n_dev, w.err,
                                                            ext_err, keep_out, num_keep);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_nms_segmented");
}

}  // namespace

int set_nms_option(const char* name, int64_t value) {
  if (std::strcmp(name, "nms.replan_min_boxes") == 0) {
    g_replan_min_boxes.store(std::max<int64_t>(0, value), std::memory_order_relaxed);
    return 0;
  }
  if (std::strcmp(name, "nms.replan_divisor") == 0) {
    g_replan_divisor.store((int)std::max<int64_t>(2, std::min<int64_t>(value, 1024)), std::memory_order_relaxed);
    return 0;
  }
  if (std::strcmp(name, "nms.device_handoff") == 0) {
    g_device_handoff.store(value != 0, std::memory_order_relaxed);
    if (value != 0) g_handoff_broken.store(false, std::memory_order_relaxed);   // switching it on again re-arms it after a give-up
    return 0;
  }
  if (std::strcmp(name, "nms.handoff_lose_flag") == 0) {
    g_handoff_lose_flag.store(value != 0, std::memory_order_relaxed);
    return 0;
  }
  if (std::strcmp(name, "nms.mask_lds_bytes") == 0) {
    g_mask_lds_bytes.store((int)std::max<int64_t>(0, std::min<int64_t>(value, 60000)), std::memory_order_relaxed);
    return 0;
  }
  if (std::strcmp(name, "nms.replan_max") == 0) {
    g_replan_max.store((int)std::max<int64_t>(0, std::min<int64_t>(value, 16)), std::memory_order_relaxed);
    return 0;
  }
  if (std::strcmp(name, "nms.step_fused") == 0) {    // 1 (default): n <= 4096 / <= 64 segments as ONE launch (tvmi_nms_step)
    g_step_fused.store(value != 0, std::memory_order_relaxed);
    return 0;
  }
  return -1;
}

int get_nms_option(const char* name, int64_t* value) {
  if (std::strcmp(name, "nms.step_fused") == 0) {
    *value = g_step_fused.load(std::memory_order_relaxed) ? 1 : 0;
    return 0;
  }
  if (std::strcmp(name, "nms.replan_min_boxes") == 0) *value = (int64_t)g_replan_min_boxes.load(std::memory_order_relaxed);
  else if (std::strcmp(name, "nms.replan_divisor") == 0) *value = g_replan_divisor.load(std::memory_order_relaxed);
  else if (std::strcmp(name, "nms.device_handoff") == 0)
    *value = (g_device_handoff.load(std::memory_order_relaxed) && !g_handoff_broken.load(std::memory_order_relaxed)) ? 1 : 0;
  else if (std::strcmp(name, "nms.handoff_lose_flag") == 0) *value = g_handoff_lose_flag.load(std::memory_order_relaxed);
  else if (std::strcmp(name, "nms.mask_lds_bytes") == 0) *value = g_mask_lds_bytes.load(std::memory_order_relaxed);
  else if (std::strcmp(name, "nms.replan_max") == 0) *value = g_replan_max.load(std::memory_order_relaxed);
  else return -1;
  return 0;
}
}  // namespace tvmi

extern "C" size_t tvmi_nms_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  const size_t CB = (size_t)tvmi::ceil_div(n, 64);
  if (CB <= (size_t)tvmi::kSmallCB) return CB * CB * 64 * sizeof(unsigned long long);
  return CB * CB * 64 * sizeof(unsigned long long) + tvmi::large_state_bytes((size_t)n);
}

namespace tvmi {
namespace {
int nms_entry(const void* dets, const int64_t* order, const int64_t* seg, int64_t n, double iou_threshold, tvmi_dtype dt,
              void* workspace, size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out, void* stream, bool may_sync) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_CHECK_ARG(n >= 0, "nms: negative box count");
  TVMI_CHECK_ARG(num_keep_out != nullptr, "nms: num_keep_out is null");
  if (n == 0) {
    hipError_t e = hipMemsetAsync(num_keep_out, 0, sizeof(int64_t), s);
    return e == hipSuccess ? 0 : set_error((int)e, "tvmi_nms: memset");
  }
  TVMI_CHECK_ARG(dets && order && keep_out && workspace, "nms: null pointer");
  TVMI_CHECK_ARG(n <= 1200000, "nms: more than 1.2M boxes is not supported by the bitmask path");
  TVMI_CHECK_ARG(workspace_bytes >= tvmi_nms_workspace_bytes(n), "nms: workspace too small");
  TVMI_CHECK_ARG(dt == TVMI_F32 || dt == TVMI_F64, "nms: dets must be float32 or float64");
  if (dt == TVMI_F32) return launch<float>(dets, order, seg, n, iou_threshold, workspace, keep_out, num_keep_out, s, may_sync);
  return launch<double>(dets, order, seg, n, iou_threshold, workspace, keep_out, num_keep_out, s, may_sync);
}
}  // namespace
}  // namespace tvmi

extern "C" int tvmi_nms(const void* dets, const int64_t* order, const int64_t* seg, int64_t n,
                        double iou_threshold, tvmi_dtype dt, void* workspace,
                        size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out,
                        void* stream) {
  return tvmi::nms_entry(dets, order, seg, n, iou_threshold, dt, workspace, workspace_bytes, keep_out, num_keep_out, stream, false);
}
extern "C" int tvmi_nms_blocking(const void* dets, const int64_t* order, const int64_t* seg, int64_t n,
                                 double iou_threshold, tvmi_dtype dt, void* workspace,
                                 size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out,
                                 void* stream) {
  return tvmi::nms_entry(dets, order, seg, n, iou_threshold, dt, workspace, workspace_bytes, keep_out, num_keep_out, stream, true);
}

extern "C" size_t tvmi_nms_segmented_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return tvmi::seg_workspace_layout(n, nullptr, nullptr);
}

namespace tvmi {
namespace {
int nms_segmented_entry(const void* dets, const int64_t* order, const int64_t* seg_keys, const int64_t* perm, int64_t n,
                        const int64_t* n_dev, const int* ext_err, double iou_threshold, tvmi_dtype dt, void* workspace,
                        size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_CHECK_ARG(n >= 0, "nms_segmented: negative box count");
  TVMI_CHECK_ARG(num_keep_out != nullptr, "nms_segmented: num_keep_out is null");
  if (n == 0) {
    hipError_t e = hipMemsetAsync(num_keep_out, 0, sizeof(int64_t), s);
    return e == hipSuccess ? 0 : set_error((int)e, "tvmi_nms_segmented: memset");
  }
  TVMI_CHECK_ARG(dets && order && seg_keys && perm && keep_out && workspace, "nms_segmented: null pointer");
  TVMI_CHECK_ARG(n <= 1200000, "nms_segmented: more than 1.2M boxes is not supported by the bitmask path");
  TVMI_CHECK_ARG(workspace_bytes >= tvmi_nms_segmented_workspace_bytes(n), "nms_segmented: workspace too small");
  TVMI_CHECK_ARG(dt == TVMI_F32 || dt == TVMI_F64, "nms_segmented: dets must be float32 or float64");
  if (dt == TVMI_F32)
    return launch_seg<float>(dets, order, seg_keys, perm, n, n_dev, ext_err, iou_threshold, workspace, keep_out, num_keep_out, s);
  return launch_seg<double>(dets, order, seg_keys, perm, n, n_dev, ext_err, iou_threshold, workspace, keep_out, num_keep_out, s);
}

int nms_small_segments_entry(const void* dets, const int64_t* order, const int64_t* seg, int64_t n, const int64_t* n_dev,
                             int64_t num_segments, double iou_threshold, tvmi_dtype dt, void* workspace,
                             size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_CHECK_ARG(n >= 0 && num_keep_out != nullptr, "nms_small_segments: bad arguments");
  if (n == 0) {
    hipError_t e = hipMemsetAsync(num_keep_out, 0, sizeof(int64_t), s);
    return e == hipSuccess ? 0 : set_error((int)e, "tvmi_nms_small_segments: memset");
  }
  TVMI_CHECK_ARG(dets && order && seg && keep_out && workspace, "nms_small_segments: null pointer");
  TVMI_CHECK_ARG(n <= 4096 && num_segments >= 1 && num_segments <= 1024,
                 "nms_small_segments: needs n <= 4096 and 1 <= num_segments <= 1024");
  TVMI_CHECK_ARG(workspace_bytes >= tvmi_nms_small_segments_workspace_bytes(n, num_segments),
                 "nms_small_segments: workspace too small");
  TVMI_CHECK_ARG(dt == TVMI_F32 || dt == TVMI_F64, "nms_small_segments: dets must be float32 or float64");
  SmallSegWorkspace w;
  small_seg_workspace_layout(n, num_segments, static_cast<char*>(workspace), &w);
  hipError_t e = hipMemsetAsync(w.sync_words, 0, 2 * sizeof(int), s);
  if (e != hipSuccess) return set_error((int)e, "tvmi_nms_small_segments: memset");
  // a segment of m boxes has ceil(m/64)*(ceil(m/64)+1)/2 tiles; m <= min(n, 1024).  Two launches whose workgroups fit next to a
  // launch that owns the LDS of every CU (see nms_small_seg_collect)
  const int nbmax = (int)std::min<int64_t>(kSmallSegBlocks, ceil_div(n, 64));
  const dim3 grid4((unsigned)num_segments, (unsigned)ceil_div(nbmax * (nbmax + 1) / 2, kTiles4));
  if (dt == TVMI_F32) {
    nms_small_seg_collect<float><<<dim3((unsigned)num_segments), dim3(1024), 0, s>>>(order, seg, (int)n, n_dev, (int)num_segments, w);
    nms_small_seg_tiles4<float><<<grid4, dim3(kTiles4 * 64), 0, s>>>(static_cast<const float*>(dets), order, iou_threshold,
                                                                   thr_band(iou_threshold), w);
  } else {
    nms_small_seg_collect<double><<<dim3((unsigned)num_segments), dim3(1024), 0, s>>>(order, seg, (int)n, n_dev, (int)num_segments, w);
    nms_small_seg_tiles4<double><<<grid4, dim3(kTiles4 * 64), 0, s>>>(static_cast<const double*>(dets), order, iou_threshold,
                                                                    thr_band(iou_threshold), w);
  }
  nms_small_seg_sweep<<<dim3((unsigned)num_segments), dim3(kSuper * kWave), 0, s>>>(order, (int)n, n_dev, w, keep_out, num_keep_out);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_nms_small_segments");
}

// masked candidates -> inputs of the device-count forms: score -inf and the largest key for a masked-out candidate (both sorts
// put it behind every live one), and the number of live candidates.  n_live must be zero before the launch.
__global__ __launch_bounds__(256) void nms_mask_inputs_kernel(const float* __restrict__ scores, const int64_t* __restrict__ seg,
                                                              const uint8_t* __restrict__ valid, int64_t n,
                                                              float* __restrict__ scores_out, int64_t* __restrict__ seg_out,
                                                              int64_t* __restrict__ n_live) {
  __shared__ int s_cnt[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n && valid[i] != 0;
  if (i < n) {
    scores_out[i] = live ? scores[i] : -INFINITY;
    seg_out[i] = live ? seg[i] : INT64_MAX;
  }
  const u64 bal = __ballot(live);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (t) atomicAdd(reinterpret_cast<unsigned long long*>(n_live), (unsigned long long)t);
  }
}
}  // namespace
}  // namespace tvmi

extern "C" int tvmi_nms_segmented(const void* dets, const int64_t* order, const int64_t* seg_keys, const int64_t* perm,
                                  int64_t n, double iou_threshold, tvmi_dtype dt, void* workspace, size_t workspace_bytes,
                                  int64_t* keep_out, int64_t* num_keep_out, void* stream) {
  return tvmi::nms_segmented_entry(dets, order, seg_keys, perm, n, nullptr, nullptr, iou_threshold, dt, workspace, workspace_bytes,
                                   keep_out, num_keep_out, stream);
}

extern "C" int tvmi_nms_segmented_devcount(const void* dets, const int64_t* order, const int64_t* seg_keys, const int64_t* perm,
                                           int64_t capacity, const int64_t* n_dev, const int* partition_flag, double iou_threshold,
                                           tvmi_dtype dt, void* workspace, size_t workspace_bytes, int64_t* keep_out,
                                           int64_t* num_keep_out, void* stream) {
  return tvmi::nms_segmented_entry(dets, order, seg_keys, perm, capacity, n_dev, partition_flag, iou_threshold, dt, workspace,
                                   workspace_bytes, keep_out, num_keep_out, stream);
}

extern "C" int tvmi_nms_mask_inputs(const float* scores, const int64_t* seg, const uint8_t* valid, int64_t n, float* scores_out,
                                    int64_t* seg_out, int64_t* n_live, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_CHECK_ARG(n >= 0 && n_live != nullptr, "nms_mask_inputs: bad arguments");
  hipError_t e = hipMemsetAsync(n_live, 0, sizeof(int64_t), s);
  if (e != hipSuccess) return tvmi::set_error((int)e, "tvmi_nms_mask_inputs: memset");
  if (n == 0) return 0;
  TVMI_CHECK_ARG(scores && seg && valid && scores_out && seg_out, "nms_mask_inputs: null pointer");
  tvmi::nms_mask_inputs_kernel<<<dim3((unsigned)tvmi::ceil_div(n, 256)), dim3(256), 0, s>>>(scores, seg, valid, n, scores_out, seg_out,
                                                                                         n_live);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_nms_mask_inputs");
}

extern "C" size_t tvmi_nms_small_segments_workspace_bytes(int64_t n, int64_t num_segments) {
  if (n <= 0 || num_segments <= 0) return 0;
  return tvmi::small_seg_workspace_layout(n, num_segments, nullptr, nullptr);
}

extern "C" int tvmi_nms_small_segments(const void* dets, const int64_t* order, const int64_t* seg, int64_t n,
                                       int64_t num_segments, double iou_threshold, tvmi_dtype dt, void* workspace,
                                       size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out, void* stream) {
  return tvmi::nms_small_segments_entry(dets, order, seg, n, nullptr, num_segments, iou_threshold, dt, workspace, workspace_bytes,
                                        keep_out, num_keep_out, stream);
}

extern "C" int tvmi_nms_small_segments_devcount(const void* dets, const int64_t* order, const int64_t* seg, int64_t capacity,
                                                const int64_t* n_dev, int64_t num_segments, double iou_threshold, tvmi_dtype dt,
                                                void* workspace, size_t workspace_bytes, int64_t* keep_out,
                                                int64_t* num_keep_out, void* stream) {
  TVMI_CHECK_ARG(n_dev != nullptr, "nms_small_segments_devcount: n_dev is null");
  return tvmi::nms_small_segments_entry(dets, order, seg, capacity, n_dev, num_segments, iou_threshold, dt, workspace,
                                        workspace_bytes, keep_out, num_keep_out, stream);
}

extern "C" size_t tvmi_nms_step_workspace_bytes(int64_t n, int64_t num_segments) {
  if (n <= 0 || num_segments <= 0) return 0;
  return tvmi::step_workspace_layout(n, num_segments, nullptr, nullptr);
}

extern "C" int tvmi_nms_step(const float* dets, const float* scores, const int64_t* seg, int64_t n, int64_t num_segments,
                             double iou_threshold, void* workspace, size_t workspace_bytes, int64_t* keep_out,
                             int64_t* num_keep_out, const int64_t* image_idx, const int64_t* labels, int64_t num_images,
                             int64_t max_dets, float* payload, int64_t row_stride, int32_t* counts, int count_in_row, void* stream) {
  using namespace tvmi;
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_CHECK_ARG(n >= 1 && n <= kSortMax, "nms_step: 1 <= n <= 4096");
  TVMI_CHECK_ARG(num_segments >= 1 && num_segments <= kStepMaxSegments, "nms_step: 1 <= num_segments <= 64");
  TVMI_CHECK_ARG(dets && scores && keep_out && num_keep_out && workspace, "nms_step: null pointer");
  TVMI_CHECK_ARG(seg != nullptr || num_segments == 1, "nms_step: segment ids are needed for more than one segment");
  TVMI_CHECK_ARG(workspace_bytes >= tvmi_nms_step_workspace_bytes(n, num_segments), "nms_step: workspace too small");
  StepPack pk{};
  if (payload) {
    TVMI_CHECK_ARG(image_idx != nullptr, "nms_step: the payload needs the image index of every box");
    TVMI_CHECK_ARG(num_images >= 1 && num_images <= kStepMaxImages, "nms_step: 1 <= num_images <= 16 for the fused payload");
    TVMI_CHECK_ARG(max_dets >= 1 && row_stride >= max_dets * 6 + (count_in_row ? 1 : 0), "nms_step: bad payload shape");
    pk.image_idx = image_idx;
    pk.labels = labels;
    pk.payload = payload;
    pk.counts = counts;
    pk.row_stride = row_stride;
    pk.num_images = (int)num_images;
    pk.max_dets = (int)max_dets;
    pk.count_in_row = count_in_row;
  }
  StepWorkspace w;
  step_workspace_layout(n, num_segments, static_cast<char*>(workspace), &w);
  if (int* sync = step_sync_blocks().get(s)) {
    w.sync = sync;                       // the library's block of this stream: clean, re-armed by the kernel
  } else {                               // graph capture (or no memory): the caller's words + a memset node
    hipError_t e = hipMemsetAsync(w.sync, 0, (size_t)kStepSyncWords * sizeof(int), s);
    if (e != hipSuccess) return set_error((int)e, "tvmi_nms_step: memset");
  }
  const int G = step_groups_per_segment(n, num_segments);
  // counting-only workgroups: enough for 16 key splits per 64 boxes, inside the cap on the launch's workgroups
  const int64_t want_rank = ceil_div(ceil_div(n, 64) * 4, num_segments);
  const int Rg = (int)std::max<int64_t>(0, std::min<int64_t>(want_rank, (kStepMaxGroups - num_segments * G) / num_segments));
  nms_step_fused<<<dim3((unsigned)num_segments, (unsigned)(G + Rg)), dim3(kStepThreads), 0, s>>>(
      dets, scores, seg, (int)n, (int)num_segments, G, iou_threshold, thr_band(iou_threshold), w, keep_out, num_keep_out, pk);
#ifdef TVMI_STEP_TIMING
  {
    static int calls = 0;
    if (++calls % 10 == 0) {
      (void)hipStreamSynchronize(s);
      unsigned long long h[64];
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(tvmi::g_step_stamp), sizeof(h));
      fprintf(stderr, "[step timing, 10 ns ticks]");
      for (int i = 1; i <= 14; ++i) fprintf(stderr, " %d:%lld", i, (long long)(h[i] - h[0]));
      fprintf(stderr, " | counting(0,G) %lld..%lld (0,last) %lld..%lld", (long long)(h[21] - h[0]), (long long)(h[22] - h[0]), (long long)(h[23] - h[0]), (long long)(h[24] - h[0]));
      fprintf(stderr, " | worker(0,16):");
      for (int i = 16; i <= 20; ++i) fprintf(stderr, " %d:%lld", i, (long long)(h[i] - h[0]));
      fprintf(stderr, "\n");
    }
  }
#endif
  TVMI_RETURN_LAUNCH_STATUS("tvmi_nms_step");
}

extern "C" int tvmi_sort_scores_desc(const float* scores, int64_t n, int64_t* order, void* stream) {
  TVMI_CHECK_ARG(n >= 0 && n <= tvmi::kSortMax, "sort_scores_desc: 0 <= n <= 4096");
  if (n == 0) return 0;
  TVMI_CHECK_ARG(scores && order, "sort_scores_desc: null pointer");
  tvmi::sort_scores_desc_rank<<<dim3((unsigned)tvmi::ceil_div(n, 64)), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(scores, (int)n, order);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_sort_scores_desc");
}
