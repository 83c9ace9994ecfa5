// nms_device.h — device helpers of the NMS kernels that more than one translation unit needs: nms.hip, and roi_align.hip for the
// launch that carries the detector step's NMS workgroups in front of the RoIAlign grid (round 6).  Moved verbatim from nms.hip.
#pragma once

#include <utility>

#include "tvmi_common.h"

namespace tvmi {
namespace {

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


constexpr int kSmallSegBoxes = 1024;
constexpr int kSmallSegBlocks = kSmallSegBoxes / 64;                          // 16 = kSuper
constexpr int kSmallSegTiles = kSmallSegBlocks * (kSmallSegBlocks + 1) / 2;  // 136

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

}  // namespace
}  // namespace tvmi
