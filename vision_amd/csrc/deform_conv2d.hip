// deform_conv2d.hip — deformable convolution v1/v2 for gfx950 (MI355X).
//
// Semantics: torchvision/csrc/ops/cpu/deform_conv2d_kernel.cpp
//   bilinear_interpolate :95-132 (zero outside (-1,H)x(-1,W), per-corner validity),
//   deformable_im2col_kernel :134-209, deformable_col2im_kernel :274-348,
//   get_coordinate_weight :407-438, deformable_col2im_coord_kernel :440-551,
//   forward host :921-1151.
//
// Forward, fp32 (the hot path): ONE fused kernel — the reference materialises
// columns[C*kh*kw, B*oh*ow] in memory (250 MB at 2x256x100x136, k=3) and then calls one GEMM
// per weight group (cuda/deform_conv2d_kernel.cu:1035-1255).  Here a 256-thread workgroup
// owns an (out-channel tile x pixel tile) of the output and walks K = (tap, in-channel) in
// slabs of 16: the offset-gather + bilinear "im2col" values of a slab are produced straight
// into LDS (they never touch HBM), the matching weight slab comes from a [tap][ic][oc]
// re-layout of the weights (coalesced, L2 resident), and the contraction runs on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate).  Sampling
// coordinates / bilinear weights are computed once per (pixel, tap, offset group) and reused
// for every input channel of the group.  Global gathers for slab s+1 are issued before the
// MFMAs of slab s (register staging, double-buffered LDS, one barrier per slab); bias is
// added in the epilogue and the result is stored coalesced along the pixel dimension.
// Other dtypes / tiny channel counts use a direct (non-MFMA) kernel with the same maths.
// (round 5: the im2col / col2im / col2im_coord building blocks of the round-3 library-GEMM backward are gone — the
// backward is deform_conv2d_bwd.hip)  Backward pieces were separate kernels combined with plain
// library GEMMs by the dispatcher glue.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "dcn_common.h"

namespace tvmi {
namespace {

// Workgroup b of a launch runs on XCD b % 8 (private 4 MB L2).  Pixel tiles are dealt in CONTIGUOUS ranges — XCD x owns tiles
// [start(x), start(x) + count(x)) — so that the taps of neighbouring tiles meet in one L2 instead of every L2 pulling the
// whole input.  A bijection of [0, ntile) for any ntile; only used when the grid is one-dimensional (y = z = 1).
__device__ __forceinline__ int tile_of_block(const DcnParams& p) {
  int tile = blockIdx.x;
  if (p.xcd_tiles && gridDim.y * gridDim.z == 1) {
    const int ntile = gridDim.x, x = tile & 7, j = tile >> 3, base = ntile >> 3, rem = ntile & 7;
    tile = x * base + min(x, rem) + j;
  }
  return tile;
}

// The 4 corners of a tap are two horizontally adjacent pairs: (o1, o2) and (o3, o4) with o2 - o1 = o4 - o3 in
// {0, 1}.  The fused kernel is bound by the rate at which the texture addresser retires SCATTERED loads (~1 lane per
// clock: 62.7 M im2col elements x 4 corners at config 4), so it fetches each pair with ONE 8-byte load.  `sel` says
// which halves are (v_left, v_right): 0 = (lo, hi) regular; 1 = (lo, lo) and 2 = (hi, hi) when the column was
// clamped (both taps are the same pixel) — in the last column the pair starts one element earlier so that it never
// leaves the row.  Needs W >= 2.
struct PairPlan {
  int b0, b1, sel;
};
struct __attribute__((packed, aligned(4))) F32Pair {
  float lo, hi;
};
__device__ __forceinline__ PairPlan make_pair_plan(const Tap<float>& t, int W) {
  PairPlan q;
  q.b0 = t.o1;
  q.b1 = t.o3;
  q.sel = 0;
  if (t.o2 == t.o1) {
    const bool last_col = (t.o1 % W) == W - 1;
    q.sel = last_col ? 2 : 1;
    q.b0 -= last_col ? 1 : 0;
    q.b1 -= last_col ? 1 : 0;
  }
  return q;
}

// ------------------------------------------------------------------ direct forward
// One thread per output element; taps outer, channels inner (coordinates reused).
template <typename T>
__global__ __launch_bounds__(256) void dcn_fwd_direct(const T* __restrict__ input, const T* __restrict__ weight,
                                                      const T* __restrict__ offset, const T* __restrict__ mask,
                                                      const T* __restrict__ bias, T* __restrict__ out,
                                                      DcnParams p, int64_t total) {
  using A = typename Acc<T>::type;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int ox = (int)(idx % p.ow);
    const int oy = (int)((idx / p.ow) % p.oh);
    const int oc = (int)((idx / ((int64_t)p.ow * p.oh)) % p.OC);
    const int b = (int)(idx / ((int64_t)p.ow * p.oh * p.OC));
    const int g = oc / p.OCg;
    const int64_t plane = (int64_t)p.H * p.W;
    const int KK = p.kh * p.kw;
    A acc = (A)0;
    for (int tap = 0; tap < KK; ++tap) {
      int ic = 0;
      while (ic < p.ICg) {
        const int c_abs = g * p.ICg + ic;
        const int og = c_abs / p.cpog;
        const int seg_end = min(p.ICg, (og + 1) * p.cpog - g * p.ICg);
        Tap<A> t;
        load_tap<T, A>(t, p, offset, mask, b, og, tap, oy, ox);
        for (; ic < seg_end; ++ic) {
          const T* pl = input + ((int64_t)b * p.C + g * p.ICg + ic) * plane;
          const A wv = ld(weight + ((int64_t)oc * p.ICg + ic) * KK + tap);
          acc += wv * sample_tap<T, A>(t, pl);
        }
      }
    }
    st(out + idx, acc + ld(bias + oc));
  }
}

std::atomic<int> g_cl_pipelined{1};  // option "dcn.cl_variant": 0 forces the non-pipelined channels-last kernel — the route of copies of
                                     // 4 GB and more (the pipelined kernel forms 32-bit lane offsets) — so that tests can reach it
std::atomic<int> g_xcd_tiles{1};   // option "dcn.xcd_tiles": contiguous pixel-tile ranges per XCD (tile_of_block)
std::atomic<int> g_cl_gather{1};  // option "dcn.channels_last_gather": the 16-bit MFMA kernel samples a [B, H*W, C] copy

// ------------------------------------------------------------------ fused MFMA forward (fp32)
constexpr int kBK = 16;

// weight [OC, ICg, kh, kw] -> wt [groups][tap][ICg_pad][OCg_pad] (zero padded)
__global__ void dcn_weight_relayout(const float* __restrict__ w, float* __restrict__ wt, DcnParams p,
                                    int ICg_pad, int OCg_pad) {
  const int KK = p.kh * p.kw;
  const int64_t total = (int64_t)p.groups * KK * ICg_pad * OCg_pad;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int oc = (int)(idx % OCg_pad);
    const int ic = (int)((idx / OCg_pad) % ICg_pad);
    const int tap = (int)((idx / ((int64_t)OCg_pad * ICg_pad)) % KK);
    const int g = (int)(idx / ((int64_t)OCg_pad * ICg_pad * KK));
    float v = 0.f;
    if (oc < p.OCg && ic < p.ICg) v = w[(((int64_t)(g * p.OCg + oc)) * p.ICg + ic) * KK + tap];
    wt[idx] = v;
  }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// WM x WN waves along M (out channels) and N (pixels), each owning MI x NI MFMA blocks of 32 x 32.
// More, smaller waves for the same workgroup tile (e.g. 8 waves of 64 x 32 instead of 4 of 64 x 64) double the
// waves per SIMD on problems that only fill the chip once (config 4: 425 tiles for 256 CUs) — the gathers of the
// next slab then have another wave's MFMAs to hide behind.
template <int WM, int WN, int MI, int NI>
__global__ __launch_bounds__(64 * WM * WN) void dcn_fwd_mfma_f32(const float* __restrict__ input,
                                                        const float* __restrict__ wt,
                                                        const float* __restrict__ offset,
                                                        const float* __restrict__ mask,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        DcnParams p, int ICg_pad, int OCg_pad) {
  constexpr int NT = 64 * WM * WN;      // threads per workgroup
  constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
  constexpr int NSUB = NT / BN;         // channel subsets among the B-tile producers
  constexpr int BPT = kBK / NSUB;       // B elements per thread per slab
  constexpr int AQ = kBK * BM / 4;               // float4 pieces of the A slab
  constexpr int AV = (AQ + NT - 1) / NT;         // pieces per thread (the last round may be partial)
  static_assert(NT % BN == 0 && kBK % NSUB == 0 && BM % 4 == 0, "tile shape does not divide evenly");
  __shared__ __attribute__((aligned(16))) float As[2][kBK][BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][kBK][BN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int g = blockIdx.z;
  const int oc0 = blockIdx.y * BM;  // within group
  const int64_t npix = (int64_t)p.B * p.oh * p.ow;
  const int64_t pix0 = (int64_t)tile_of_block(p) * BN;
  const int KK = p.kh * p.kw;
  const int64_t in_plane = (int64_t)p.H * p.W;

  // this thread's producer pixel
  const int pn = tid % BN, csub = tid / BN;
  const int64_t my_pix = pix0 + pn;
  const bool pix_ok = my_pix < npix;
  int pb = 0, poy = 0, pox = 0;
  if (pix_ok) {
    pox = (int)(my_pix % p.ow);
    poy = (int)((my_pix / p.ow) % p.oh);
    pb = (int)(my_pix / ((int64_t)p.ow * p.oh));
  }
  const float* in_b = input + ((int64_t)pb * p.C + (int64_t)g * p.ICg) * in_plane;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  float bv[BPT][4];   // staged corner values of the next slab
  float4 av[AV];      // staged weights of the next slab
  Tap<float> tap_cur;
  tap_cur.o1 = tap_cur.o2 = tap_cur.o3 = tap_cur.o4 = 0;
  tap_cur.w1 = tap_cur.w2 = tap_cur.w3 = tap_cur.w4 = 0.f;
  tap_cur.m = 0.f;
  PairPlan plan;
  plan.b0 = plan.b1 = 0;
  plan.sel = 1;

  // Slab iteration space: tap (outer) x offset-group segment x ic0 (inner).
  int s_tap = 0, s_ic = 0, s_seg_end = 0;
  bool have = KK > 0 && p.ICg > 0;
  // raw offsets / mask of the segment AFTER the current one, fetched when the current one begins (a whole segment ahead)
  TapRaw<float> raw_next = tap_raw_identity<float>();
  auto fetch_raw = [&](int tap, int ic) {
    if (pix_ok) load_tap_raw<float>(raw_next, p, offset, mask, pb, (g * p.ICg + ic) / p.cpog, tap, poy, pox);
  };
  auto begin_segment = [&](int tap, int ic) {
    const int og = (g * p.ICg + ic) / p.cpog;
    s_seg_end = min(p.ICg, (og + 1) * p.cpog - g * p.ICg);
    if (pix_ok) {
      tap_from_raw<float, float>(tap_cur, p, raw_next, tap, poy, pox);
      plan = make_pair_plan(tap_cur, p.W);
    }
    int t2 = tap, i2 = s_seg_end;
    if (i2 >= p.ICg) {
      i2 = 0;
      t2 = tap + 1;
    }
    if (t2 < KK) fetch_raw(t2, i2);
  };
  auto issue_loads = [&](int tap, int ic0, int kmax) {
#pragma unroll
    for (int e = 0; e < BPT; ++e) {
      const int kk = csub + e * NSUB;
      if (kk < kmax && pix_ok) {
        const float* pl = in_b + (int64_t)(ic0 + kk) * in_plane;
        const F32Pair r0 = *reinterpret_cast<const F32Pair*>(pl + plan.b0);
        const F32Pair r1 = *reinterpret_cast<const F32Pair*>(pl + plan.b1);
        bv[e][0] = r0.lo;  // raw halves; commit() picks (left, right) by plan.sel once the data is needed
        bv[e][1] = r0.hi;
        bv[e][2] = r1.lo;
        bv[e][3] = r1.hi;
      } else {
        bv[e][0] = bv[e][1] = bv[e][2] = bv[e][3] = 0.f;
      }
    }
    const float* wsrc = wt + (((int64_t)g * KK + tap) * ICg_pad + ic0) * OCg_pad + oc0;
#pragma unroll
    for (int e = 0; e < AV; ++e) {
      const int lin = (tid + e * NT) * 4;
      const int kk = lin / BM, m = lin - kk * BM;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kk < kmax && oc0 + m < OCg_pad) v = *reinterpret_cast<const float4*>(wsrc + (int64_t)kk * OCg_pad + m);
      av[e] = v;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int e = 0; e < BPT; ++e) {
      const int kk = csub + e * NSUB;
      const float v1 = plan.sel == 2 ? bv[e][1] : bv[e][0], v2 = plan.sel == 1 ? bv[e][0] : bv[e][1];
      const float v3 = plan.sel == 2 ? bv[e][3] : bv[e][2], v4 = plan.sel == 1 ? bv[e][2] : bv[e][3];
      Bs[buf][kk][pn] = tap_cur.m * (tap_cur.w1 * v1 + tap_cur.w2 * v2 + tap_cur.w3 * v3 + tap_cur.w4 * v4);
    }
#pragma unroll
    for (int e = 0; e < AV; ++e) {
      const int lin = (tid + e * NT) * 4;
      const int kk = lin / BM, m = lin - kk * BM;
      if (AQ % NT == 0 || kk < kBK) *reinterpret_cast<float4*>(&As[buf][kk][m]) = av[e];
    }
  };

  int buf = 0;
  if (have) {
    fetch_raw(0, 0);
    begin_segment(0, 0);
    issue_loads(0, 0, min(kBK, s_seg_end));
  }
  while (have) {
    commit(buf);  // uses tap_cur of the slab that was loaded
    __syncthreads();
    // advance to the next slab and start its loads before the MFMAs
    int n_tap = s_tap, n_ic = s_ic + kBK;
    bool n_have = true;
    if (n_ic >= s_seg_end) {
      n_ic = s_seg_end;
      if (n_ic >= p.ICg) {
        n_ic = 0;
        n_tap = s_tap + 1;
        if (n_tap >= KK) n_have = false;
      }
      if (n_have) begin_segment(n_tap, n_ic);
    }
    if (n_have) issue_loads(n_tap, n_ic, min(kBK, s_seg_end - n_ic));
    // MFMA over this slab
    const int kq = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int kk = 0; kk < kBK; kk += 2) {
      float a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = As[buf][kk + kq][(wm * MI + mi) * 32 + l31];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = Bs[buf][kk + kq][(wn * NI + ni) * 32 + l31];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
    s_tap = n_tap;
    s_ic = n_ic;
    have = n_have;
    buf ^= 1;
  }

  // epilogue: D[row][col], col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (oc)
  const int l31 = lane & 31, kq = lane >> 5;
  // the bias of this lane's 16 MI output rows as ONE batch of independent loads (clamped row index).  Round 3 loaded
  // `bias[oc]` inside the guarded store: 32 dependent load -> wait -> store round trips at the end of every tile.
  float brow[MI][16];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      brow[mi][r] = bias[g * p.OCg + min(oc0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq, p.OCg - 1)];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int64_t pix = pix0 + (wn * NI + ni) * 32 + l31;
    if (pix >= npix) continue;
    const int ox = (int)(pix % p.ow);
    const int oy = (int)((pix / p.ow) % p.oh);
    const int b = (int)(pix / ((int64_t)p.ow * p.oh));
    float* obase = out + ((int64_t)b * p.OC + (int64_t)g * p.OCg) * p.oh * p.ow + (int64_t)oy * p.ow + ox;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = oc0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
        if (oc < p.OCg) obase[(int64_t)oc * p.oh * p.ow] = acc[mi][ni][r] + brow[mi][r];
      }
    }
  }
}

// ------------------------------------------------------------------ fused MFMA forward, 16-bit tensors
// fp16 / bf16 tensors on v_mfma_f32_32x32x16_{f16,bf16}: the same fused structure as the fp32 kernel above — offset-gather +
// bilinear "im2col" values of a K slab produced straight into LDS, weights from a [tap][ic slab][oc][16] re-layout — but
// ONE matrix instruction per 32x32 block and 16-deep slab (16x the fp32 MFMA rate), fp32 accumulation.  The sampled values
// are rounded to the 16-bit type when they enter LDS: exactly what the reference does (its `columns` tensor has the input
// dtype, cuda/deform_conv2d_kernel.cu:1234-1239), while its addmm_ also ACCUMULATES in 16 bits — the products here are
// exact (16-bit x 16-bit fits fp32) and the sum is fp32, so the result sits inside the reference's own 16-bit error.
// LDS rows are [row][k] with a 48-byte pitch: a lane's 8 consecutive k of one row is a single 16-byte read, and 16 lanes x
// 48 bytes tile the 64 banks without conflict.
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kRowPitch16 = 24;   // 16-bit elements per LDS row: 16 k-values + 8 of padding (48 bytes)

template <typename T>
__device__ __forceinline__ f32x16v mfma16(uint4 a, uint4 b, f32x16v c) {
  if constexpr (std::is_same<T, __half>::value)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ unsigned short to16(float v) {
  // the value is an fp32 RESULT (mask x bilinear sum, rounded to fp32) that is then rounded to 16 bits: keep the backend from
  // folding the last multiply into v_fma_mixlo_f16, which rounds the exact product ONCE — the planar and the channels-last
  // kernel must produce the same bits (test_deform_conv2d_channels_last_gather_is_bit_identical), and round 4's branch-free
  // make_tap made that fold possible in one of them
  asm volatile("" : "+v"(v));
  if constexpr (std::is_same<T, __half>::value) return __half_as_ushort(__float2half(v));
  else {
    const __hip_bfloat16 b = __float2bfloat16(v);
    return *reinterpret_cast<const unsigned short*>(&b);
  }
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ f32x2 unpack16x2(unsigned u) {   // the two 16-bit values of a dword, as fp32
  if constexpr (std::is_same<T, __half>::value) return f32x2{__half2float(__ushort_as_half((unsigned short)(u & 0xffffu))), __half2float(__ushort_as_half((unsigned short)(u >> 16)))};
  else return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
}
template <typename T>
__device__ __forceinline__ float from16bits(unsigned short h) {
  if constexpr (std::is_same<T, __half>::value) return __half2float(__ushort_as_half(h));
  else return __uint_as_float((unsigned)h << 16);
}

// weight [OC, ICg, kh, kw] (T) -> wt16 [groups][tap][ICg_pad / 16][OCg_pad][16] (T, zero padded)
template <typename T>
__global__ void dcn_weight_relayout16(const T* __restrict__ w, T* __restrict__ wt, DcnParams p, int ICg_pad, int OCg_pad) {
  const int KK = p.kh * p.kw;
  const int64_t total = (int64_t)p.groups * KK * ICg_pad * OCg_pad;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % 16);
    const int oc = (int)((idx / 16) % OCg_pad);
    const int slab = (int)((idx / ((int64_t)16 * OCg_pad)) % (ICg_pad / 16));
    const int tap = (int)((idx / ((int64_t)OCg_pad * ICg_pad)) % KK);
    const int g = (int)(idx / ((int64_t)OCg_pad * ICg_pad * KK));
    const int ic = slab * 16 + k;
    float v = 0.f;
    if (oc < p.OCg && ic < p.ICg) v = ld(w + (((int64_t)(g * p.OCg + oc)) * p.ICg + ic) * KK + tap);
    st(wt + idx, v);
  }
}

template <typename T, int WM, int WN, int MI, int NI>
__global__ __launch_bounds__(64 * WM * WN) void dcn_fwd_mfma_16(const T* __restrict__ input, const T* __restrict__ wt,
                                                                const T* __restrict__ offset, const T* __restrict__ mask,
                                                                const T* __restrict__ bias, T* __restrict__ out, DcnParams p,
                                                                int ICg_pad, int OCg_pad) {
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
  constexpr int NSUB = NT / BN;          // channel subsets among the B-tile producers
  constexpr int BPT = kBK / NSUB;        // consecutive channels per thread and slab (one packed LDS write)
  constexpr int AQ = BM * 2;             // 16-byte pieces of the A slab ([BM][16] 16-bit)
  constexpr int AV = (AQ + NT - 1) / NT;
  static_assert(NT % BN == 0 && kBK % NSUB == 0 && BPT % 2 == 0 && kBK == 16, "tile shape does not divide evenly");
  __shared__ __attribute__((aligned(16))) unsigned short As[2][BM][kRowPitch16];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[2][BN][kRowPitch16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int g = blockIdx.z;
  const int oc0 = blockIdx.y * BM;
  const int64_t npix = (int64_t)p.B * p.oh * p.ow;
  const int64_t pix0 = (int64_t)tile_of_block(p) * BN;
  const int KK = p.kh * p.kw;
  const int64_t in_plane = (int64_t)p.H * p.W;
  const int nslab = ICg_pad / 16;

  const int pn = tid % BN, csub = tid / BN;
  const int64_t my_pix = pix0 + pn;
  const bool pix_ok = my_pix < npix;
  int pb = 0, poy = 0, pox = 0;
  if (pix_ok) {
    pox = (int)(my_pix % p.ow);
    poy = (int)((my_pix / p.ow) % p.oh);
    pb = (int)(my_pix / ((int64_t)p.ow * p.oh));
  }
  const T* in_b = input + ((int64_t)pb * p.C + (int64_t)g * p.ICg) * in_plane;

  f32x16v acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  unsigned bv[BPT][2];   // staged corner pairs (two 16-bit pixels per dword) of the next slab: top pair, bottom pair
  uint4 av[AV];          // staged weights of the next slab
  Tap<float> tap_cur;
  tap_cur.o1 = tap_cur.o2 = tap_cur.o3 = tap_cur.o4 = 0;
  tap_cur.w1 = tap_cur.w2 = tap_cur.w3 = tap_cur.w4 = 0.f;
  tap_cur.m = 0.f;
  PairPlan plan;
  plan.b0 = plan.b1 = 0;
  plan.sel = 1;

  int s_tap = 0, s_ic = 0, s_seg_end = 0;
  bool have = KK > 0 && p.ICg > 0;
  // raw offsets / mask of the segment AFTER the current one, fetched when the current one begins (a whole segment ahead)
  TapRaw<T> raw_next = tap_raw_identity<T>();
  auto fetch_raw = [&](int tap, int ic) {
    if (pix_ok) load_tap_raw<T>(raw_next, p, offset, mask, pb, (g * p.ICg + ic) / p.cpog, tap, poy, pox);
  };
  auto begin_segment = [&](int tap, int ic) {
    const int og = (g * p.ICg + ic) / p.cpog;
    s_seg_end = min(p.ICg, (og + 1) * p.cpog - g * p.ICg);
    if (pix_ok) {
      tap_from_raw<T, float>(tap_cur, p, raw_next, tap, poy, pox);
      plan = make_pair_plan(tap_cur, p.W);
    }
    int t2 = tap, i2 = s_seg_end;
    if (i2 >= p.ICg) {
      i2 = 0;
      t2 = tap + 1;
    }
    if (t2 < KK) fetch_raw(t2, i2);
  };
  auto issue_loads = [&](int tap, int ic0, int kmax) {
#pragma unroll
    for (int e = 0; e < BPT; ++e) {
      const int kk = csub * BPT + e;
      bv[e][0] = bv[e][1] = 0u;
      if (kk < kmax && pix_ok) {
        const T* pl = in_b + (int64_t)(ic0 + kk) * in_plane;
        __builtin_memcpy(&bv[e][0], pl + plan.b0, 4);   // two adjacent pixels: ONE dword load at 2-byte alignment
        __builtin_memcpy(&bv[e][1], pl + plan.b1, 4);
      }
    }
    // a slab never straddles an offset-group segment boundary in the re-laid-out weights: the slab index is ic0 / 16 only
    // when ic0 is a multiple of 16; segments that start elsewhere are served from the slab that contains them with the
    // rows before ic0 zeroed by kmax / the B side (their im2col values are zero)
    const T* wsrc = wt + ((((int64_t)g * KK + tap) * nslab + ic0 / 16) * OCg_pad + oc0) * 16;
#pragma unroll
    for (int e = 0; e < AV; ++e) {
      const int piece = tid + e * NT;
      const int m = piece >> 1, half = piece & 1;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (piece < AQ && oc0 + m < OCg_pad) v = *reinterpret_cast<const uint4*>(wsrc + (int64_t)m * 16 + half * 8);
      av[e] = v;
    }
  };
  auto commit = [&](int buf, int kbase, int len) {
    unsigned short hv[BPT];
#pragma unroll
    for (int e = 0; e < BPT; ++e) {
      const float lo0 = from16bits<T>((unsigned short)(bv[e][0] & 0xffffu)), hi0 = from16bits<T>((unsigned short)(bv[e][0] >> 16));
      const float lo1 = from16bits<T>((unsigned short)(bv[e][1] & 0xffffu)), hi1 = from16bits<T>((unsigned short)(bv[e][1] >> 16));
      const float v1 = plan.sel == 2 ? hi0 : lo0, v2 = plan.sel == 1 ? lo0 : hi0;
      const float v3 = plan.sel == 2 ? hi1 : lo1, v4 = plan.sel == 1 ? lo1 : hi1;
      hv[e] = to16<T>(tap_cur.m * (tap_cur.w1 * v1 + tap_cur.w2 * v2 + tap_cur.w3 * v3 + tap_cur.w4 * v4));
    }
    if (len == kBK) {   // whole slab (kbase = 0): this thread's BPT consecutive k as packed dwords
      unsigned* dst = reinterpret_cast<unsigned*>(&Bs[buf][pn][csub * BPT]);
#pragma unroll
      for (int e = 0; e < BPT / 2; ++e) dst[e] = (unsigned)hv[2 * e] | ((unsigned)hv[2 * e + 1] << 16);
    } else {            // piece of a slab: only its k positions, element by element (the rest of the row was zeroed)
#pragma unroll
      for (int e = 0; e < BPT; ++e)
        if (csub * BPT + e < len) Bs[buf][pn][kbase + csub * BPT + e] = hv[e];
    }
#pragma unroll
    for (int e = 0; e < AV; ++e) {
      const int piece = tid + e * NT;
      if (piece < AQ) *reinterpret_cast<uint4*>(&As[buf][piece >> 1][(piece & 1) * 8]) = av[e];
    }
  };

  // A segment that does not start on a slab boundary (offset groups whose channel count is not a multiple of 16) is walked
  // in pieces that stay inside one weight slab: ic0 advances to the next multiple of 16, the k positions of the piece inside
  // the slab are kbase = ic0 % 16 .. and the B rows outside it are zeroed (zero x weight = nothing).
  auto slab_len = [&](int ic) { return min(min(kBK - (ic & 15), s_seg_end - ic), kBK); };
  int buf = 0;
  int cur_len = 0, cur_base = 0;
  if (have) {
    fetch_raw(0, 0);
    begin_segment(0, 0);
    cur_len = slab_len(0);
    cur_base = 0;
    issue_loads(0, 0, cur_len);
  }
  // zero both B buffers once: rows of a partial slab that no producer writes must read as zero
  for (int i = tid; i < 2 * BN * kRowPitch16 / 2; i += NT) reinterpret_cast<unsigned*>(&Bs[0][0][0])[i] = 0u;
  __syncthreads();
  while (have) {
    const bool partial = cur_len < kBK;
    if (partial) {   // clear this buffer's B rows first (another slab's values may sit in the k positions this one skips)
      for (int i = tid; i < BN * kRowPitch16 / 2; i += NT) reinterpret_cast<unsigned*>(&Bs[buf][0][0])[i] = 0u;
      __syncthreads();
    }
    commit(buf, cur_base, cur_len);
    __syncthreads();
    int n_tap = s_tap, n_ic = s_ic + cur_len;
    bool n_have = true;
    if (n_ic >= s_seg_end) {
      n_ic = s_seg_end;
      if (n_ic >= p.ICg) {
        n_ic = 0;
        n_tap = s_tap + 1;
        if (n_tap >= KK) n_have = false;
      }
      if (n_have) begin_segment(n_tap, n_ic);
    }
    int n_len = 0, n_base = 0;
    if (n_have) {
      n_len = slab_len(n_ic);
      n_base = n_ic & 15;
      issue_loads(n_tap, n_ic, n_len);
    }
    const int kq = lane >> 5, l31 = lane & 31;
    uint4 a[MI], b[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const uint4*>(&As[buf][(wm * MI + mi) * 32 + l31][kq * 8]);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const uint4*>(&Bs[buf][(wn * NI + ni) * 32 + l31][kq * 8]);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mfma16<T>(a[mi], b[ni], acc[mi][ni]);
    s_tap = n_tap;
    s_ic = n_ic;
    cur_len = n_len;
    cur_base = n_base;
    have = n_have;
    buf ^= 1;
  }

  // epilogue: D[row][col], col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (oc)
  const int l31 = lane & 31, kq = lane >> 5;
  float brow[MI][16];   // one batch of independent bias loads (see the fp32 kernel's epilogue)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      brow[mi][r] = ld(bias + g * p.OCg + min(oc0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq, p.OCg - 1));
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int64_t pix = pix0 + (wn * NI + ni) * 32 + l31;
    if (pix >= npix) continue;
    const int ox = (int)(pix % p.ow);
    const int oy = (int)((pix / p.ow) % p.oh);
    const int b = (int)(pix / ((int64_t)p.ow * p.oh));
    T* obase = out + ((int64_t)b * p.OC + (int64_t)g * p.OCg) * p.oh * p.ow + (int64_t)oy * p.ow + ox;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = oc0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
        if (oc < p.OCg) st(obase + (int64_t)oc * p.oh * p.ow, acc[mi][ni][r] + brow[mi][r]);
      }
    }
  }
}

// ------------------------------------------------------------------ fused MFMA forward, 16-bit tensors, channels-last gather
// What bounds dcn_fwd_mfma_16 at config 4 (0.20 ms) is neither the texture addresser nor the matrix pipe: a channels-last
// gather with a QUARTER of the scattered lane-loads ran in the same time (profiles/r03_dcn16_channels_last_k16_variants.json).
// Its loop has ONE 16-deep slab of global loads in flight per workgroup, the matrix work of a slab is a handful of MFMAs, so
// every one of the 144 iterations of a tile costs a full L2 round trip (~0.7 us).  This kernel does the same contraction
// with half the iterations and a quarter of the lane-loads:
//   * a pre-pass rewrites the input as [B, H*W, C] (dcn_to_channels_last: 14 MB at config 4, ~10 us); one 16-byte lane-load
//     then fetches a corner for EIGHT channels — producer item = (pixel, channel octet), 4 loads and one 16-byte LDS write;
//   * K slabs are 32 deep (two v_mfma_f32_32x32x16 steps per 32x32 block) in [row][k] LDS rows of 80 bytes (16 lanes x 80 B
//     tile the 64 banks without conflict);
//   * weights come from an octet-major re-layout [tap][ic / 8][oc][8], so a slab may start at any multiple of 8 channels
//     (offset-group boundaries cut slabs, never octets) and the positions past its end are written as zeros — no partial-slab
//     passes.
// Same corner values and the same rounding to the 16-bit type as the planar kernel; the k products are exact and summed in
// fp32 — grouped in 16s from the start of each offset-group segment, which is the planar kernel's grouping when the channels
// per offset group are a multiple of 16 (then the two kernels agree bit for bit).  Needs C, C / groups and C / offset_groups
// divisible by 8.
template <typename T>
__global__ __launch_bounds__(256) void dcn_to_channels_last(const T* __restrict__ in, T* __restrict__ out, int C, int HW) {
  __shared__ T tile[32][33];
  const int b = blockIdx.z, hw0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const T* src = in + (int64_t)b * C * HW;
  T* dst = out + (int64_t)b * HW * C;
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (c0 + r < C && hw0 + tx < HW) tile[r][tx] = src[(int64_t)(c0 + r) * HW + hw0 + tx];
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (hw0 + r < HW && c0 + tx < C) dst[(int64_t)(hw0 + r) * C + c0 + tx] = tile[tx][r];
}

// weight [OC, ICg, kh, kw] (T) -> wt8 [groups][tap][ICg / 8][OCg_pad][8] (T, zero padded in oc)
template <typename T>
__global__ void dcn_weight_relayout8(const T* __restrict__ w, T* __restrict__ wt, DcnParams p, int OCg_pad) {
  const int KK = p.kh * p.kw;
  const int64_t total = (int64_t)p.groups * KK * p.ICg * OCg_pad;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % 8);
    const int oc = (int)((idx / 8) % OCg_pad);
    const int oct = (int)((idx / ((int64_t)8 * OCg_pad)) % (p.ICg / 8));
    const int tap = (int)((idx / ((int64_t)OCg_pad * p.ICg)) % KK);
    const int g = (int)(idx / ((int64_t)OCg_pad * p.ICg * KK));
    const int ic = oct * 8 + k;
    float v = 0.f;
    if (oc < p.OCg) v = ld(w + (((int64_t)(g * p.OCg + oc)) * p.ICg + ic) * KK + tap);
    st(wt + idx, v);
  }
}

// BK: slab depth (32 channels as shipped); LDS rows are [row][k] with 8 elements of padding (80 bytes: 16 lanes tile the 64 banks)
template <typename T, int WM, int WN, int MI, int NI, int BK>
__global__ __launch_bounds__(64 * WM * WN) void dcn_fwd_mfma_16_cl(const T* __restrict__ input_cl, const T* __restrict__ wt8,
                                                                   const T* __restrict__ offset, const T* __restrict__ mask,
                                                                   const T* __restrict__ bias, T* __restrict__ out, DcnParams p,
                                                                   int OCg_pad) {
  constexpr int kClBK = BK, kClOct = BK / 8, kClPitch = BK + 8;
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
  constexpr int PI = (kClOct * BN + NT - 1) / NT;   // producer items (pixel, octet) per thread and slab
  constexpr int AQ = BM * kClOct;                   // 16-byte pieces of the A slab
  constexpr int AV = (AQ + NT - 1) / NT;
  static_assert(NT % BN == 0, "a thread's producer items must share one pixel");
  extern __shared__ __attribute__((aligned(16))) unsigned short dcn16_cl_lds[];
  unsigned short(*As)[BM][kClPitch] = reinterpret_cast<unsigned short(*)[BM][kClPitch]>(dcn16_cl_lds);
  unsigned short(*Bs)[BN][kClPitch] = reinterpret_cast<unsigned short(*)[BN][kClPitch]>(dcn16_cl_lds + 2 * BM * kClPitch);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int g = blockIdx.z;
  const int oc0 = blockIdx.y * BM;
  const int64_t npix = (int64_t)p.B * p.oh * p.ow;
  const int64_t pix0 = (int64_t)tile_of_block(p) * BN;
  const int KK = p.kh * p.kw;
  const int noct = p.ICg / 8;

  const int pn = tid % BN, oct0 = tid / BN;   // item r of this thread: octet oct0 + r * (NT / BN)
  const int64_t my_pix = pix0 + pn;
  const bool pix_ok = my_pix < npix;
  int pb = 0, poy = 0, pox = 0;
  if (pix_ok) {
    pox = (int)(my_pix % p.ow);
    poy = (int)((my_pix / p.ow) % p.oh);
    pb = (int)(my_pix / ((int64_t)p.ow * p.oh));
  }
  const T* in_b = input_cl + (int64_t)pb * p.H * p.W * p.C + (int64_t)g * p.ICg;

  f32x16v acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  uint4 cv[PI][4];   // the four corners of the next slab's octets
  uint4 av[AV];      // staged weights of the next slab
  Tap<float> tap_cur;
  tap_cur.o1 = tap_cur.o2 = tap_cur.o3 = tap_cur.o4 = 0;
  tap_cur.w1 = tap_cur.w2 = tap_cur.w3 = tap_cur.w4 = 0.f;
  tap_cur.m = 0.f;

  int s_tap = 0, s_ic = 0, s_seg_end = 0;
  bool have = KK > 0 && p.ICg > 0;
  // raw offsets / mask of the segment AFTER the current one, fetched when the current one begins (a whole segment ahead)
  TapRaw<T> raw_next = tap_raw_identity<T>();
  auto fetch_raw = [&](int tap, int ic) {
    if (pix_ok) load_tap_raw<T>(raw_next, p, offset, mask, pb, (g * p.ICg + ic) / p.cpog, tap, poy, pox);
  };
  auto begin_segment = [&](int tap, int ic) {
    const int og = (g * p.ICg + ic) / p.cpog;
    s_seg_end = min(p.ICg, (og + 1) * p.cpog - g * p.ICg);
    if (pix_ok) {
      tap_from_raw<T, float>(tap_cur, p, raw_next, tap, poy, pox);
    }
    int t2 = tap, i2 = s_seg_end;
    if (i2 >= p.ICg) {
      i2 = 0;
      t2 = tap + 1;
    }
    if (t2 < KK) fetch_raw(t2, i2);
  };
  auto issue_loads = [&](int tap, int ic0, int kmax) {   // kmax: channels of the slab (a multiple of 8)
#pragma unroll
    for (int r = 0; r < PI; ++r) {
      const int oct = oct0 + r * (NT / BN);
      cv[r][0] = cv[r][1] = cv[r][2] = cv[r][3] = make_uint4(0u, 0u, 0u, 0u);
      if (oct < kClOct && 8 * oct < kmax && pix_ok) {
        const T* base = in_b + ic0 + 8 * oct;
        cv[r][0] = *reinterpret_cast<const uint4*>(base + (int64_t)tap_cur.o1 * p.C);
        cv[r][1] = *reinterpret_cast<const uint4*>(base + (int64_t)tap_cur.o2 * p.C);
        cv[r][2] = *reinterpret_cast<const uint4*>(base + (int64_t)tap_cur.o3 * p.C);
        cv[r][3] = *reinterpret_cast<const uint4*>(base + (int64_t)tap_cur.o4 * p.C);
      }
    }
    const T* wsrc = wt8 + ((((int64_t)g * KK + tap) * noct + ic0 / 8) * OCg_pad + oc0) * 8;
#pragma unroll
    for (int e = 0; e < AV; ++e) {
      const int piece = tid + e * NT;
      const int m = piece % BM, oct = piece / BM;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (piece < AQ && 8 * oct < kmax && oc0 + m < OCg_pad) v = *reinterpret_cast<const uint4*>(wsrc + ((int64_t)oct * OCg_pad + m) * 8);
      av[e] = v;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int r = 0; r < PI; ++r) {
      const int oct = oct0 + r * (NT / BN);
      if (oct >= kClOct) continue;
      const unsigned* c0 = reinterpret_cast<const unsigned*>(&cv[r][0]);
      const unsigned* c1 = reinterpret_cast<const unsigned*>(&cv[r][1]);
      const unsigned* c2 = reinterpret_cast<const unsigned*>(&cv[r][2]);
      const unsigned* c3 = reinterpret_cast<const unsigned*>(&cv[r][3]);
      unsigned packed[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        unsigned short h[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int sh = 16 * half;
          const float v1 = from16bits<T>((unsigned short)(c0[d] >> sh)), v2 = from16bits<T>((unsigned short)(c1[d] >> sh));
          const float v3 = from16bits<T>((unsigned short)(c2[d] >> sh)), v4 = from16bits<T>((unsigned short)(c3[d] >> sh));
          h[half] = to16<T>(tap_cur.m * (tap_cur.w1 * v1 + tap_cur.w2 * v2 + tap_cur.w3 * v3 + tap_cur.w4 * v4));
        }
        packed[d] = (unsigned)h[0] | ((unsigned)h[1] << 16);
      }
      *reinterpret_cast<uint4*>(&Bs[buf][pn][8 * oct]) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    }
#pragma unroll
    for (int e = 0; e < AV; ++e) {
      const int piece = tid + e * NT;
      if (piece < AQ) *reinterpret_cast<uint4*>(&As[buf][piece % BM][8 * (piece / BM)]) = av[e];
    }
  };

  int buf = 0;
  if (have) {
    fetch_raw(0, 0);
    begin_segment(0, 0);
    issue_loads(0, 0, min(kClBK, s_seg_end));
  }
  while (have) {
    commit(buf);   // uses tap_cur of the slab that was loaded
    __syncthreads();
    int n_tap = s_tap, n_ic = s_ic + kClBK;
    bool n_have = true;
    if (n_ic >= s_seg_end) {
      n_ic = s_seg_end;
      if (n_ic >= p.ICg) {
        n_ic = 0;
        n_tap = s_tap + 1;
        if (n_tap >= KK) n_have = false;
      }
      if (n_have) begin_segment(n_tap, n_ic);
    }
    if (n_have) issue_loads(n_tap, n_ic, min(kClBK, s_seg_end - n_ic));
    const int kq = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int step = 0; step < kClBK / 16; ++step) {
      uint4 a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const uint4*>(&As[buf][(wm * MI + mi) * 32 + l31][16 * step + kq * 8]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const uint4*>(&Bs[buf][(wn * NI + ni) * 32 + l31][16 * step + kq * 8]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mfma16<T>(a[mi], b[ni], acc[mi][ni]);
    }
    s_tap = n_tap;
    s_ic = n_ic;
    have = n_have;
    buf ^= 1;
  }

  const int l31 = lane & 31, kq = lane >> 5;
  float brow[MI][16];   // one batch of independent bias loads (see the fp32 kernel's epilogue)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      brow[mi][r] = ld(bias + g * p.OCg + min(oc0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq, p.OCg - 1));
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int64_t pix = pix0 + (wn * NI + ni) * 32 + l31;
    if (pix >= npix) continue;
    const int ox = (int)(pix % p.ow);
    const int oy = (int)((pix / p.ow) % p.oh);
    const int b = (int)(pix / ((int64_t)p.ow * p.oh));
    T* obase = out + ((int64_t)b * p.OC + (int64_t)g * p.OCg) * p.oh * p.ow + (int64_t)oy * p.ow + ox;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = oc0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
        if (oc < p.OCg) st(obase + (int64_t)oc * p.oh * p.ow, acc[mi][ni][r] + brow[mi][r]);
      }
    }
  }
}

// ------------------------------------------------------------------ round 4: the channels-last kernel, pipelined
// Same tile, arithmetic and LDS layout as dcn_fwd_mfma_16_cl above, two changes that the counters of round 3 asked for
// (the kernel was neither MFMA- nor HBM-bound: ~1600 cycles per 32-channel slab against 256 cycles of MFMA work):
//  * the loads of D slabs are in flight, not one: D register stages (4 corner octets + the weight pieces + the 5 tap
//    weights of the slab they belong to), the loop unrolled D times so that every stage index is static and the compiler's
//    counted `s_waitcnt vmcnt(N)` leaves the D-1 newer stages in flight.  Every load is unconditional (addresses clamped,
//    values zeroed at commit) and the slab iterator SATURATES at the last slab (D-1 redundant re-loads per tile) — a
//    conditional load would make the counted wait conservative;
//  * producer lanes are (pixel, octet) with the OCTET fastest: kOct neighbouring lanes read one 16*kOct-byte run of a
//    pixel's channel vector, so a gather instruction touches 64/kOct cache lines instead of 64 (the texture path is paid
//    per line: profiles/r04_pmc_sq_roi7_summary.txt).
// The workgroup -> pixel-tile map deals CONTIGUOUS tile ranges to the 8 XCDs (workgroup b runs on XCD b % 8), so that the
// taps of neighbouring tiles meet in one private L2.
template <typename T, int WM, int WN, int MI, int NI, int BK, int D, int WPE>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void dcn_fwd_mfma_16_clp(const T* __restrict__ input_cl, const T* __restrict__ wt8,
                                                                    const T* __restrict__ offset, const T* __restrict__ mask,
                                                                    const T* __restrict__ bias, T* __restrict__ out, DcnParams p,
                                                                    int OCg_pad) {
  constexpr int kOct = BK / 8, kPitch = BK + 8;
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
  constexpr int NITEM = BN * kOct;                  // producer items (pixel, octet) of a slab
  constexpr int PI = (NITEM + NT - 1) / NT;
  constexpr int AQ = BM * kOct;                     // 16-byte pieces of the A slab
  constexpr int AV = (AQ + NT - 1) / NT;
  static_assert(D >= 2 && D <= 4, "pipeline depth");
  extern __shared__ __attribute__((aligned(16))) unsigned short dcn16_clp_lds[];
  unsigned short(*As)[BM][kPitch] = reinterpret_cast<unsigned short(*)[BM][kPitch]>(dcn16_clp_lds);
  unsigned short(*Bs)[BN][kPitch] = reinterpret_cast<unsigned short(*)[BN][kPitch]>(dcn16_clp_lds + 2 * BM * kPitch);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int g = blockIdx.z;
  const int oc0 = blockIdx.y * BM;
  const int64_t npix = (int64_t)p.B * p.oh * p.ow;
  const int64_t pix0 = (int64_t)tile_of_block(p) * BN;
  const int KK = p.kh * p.kw;
  const int noct = p.ICg / 8;

  // producer items of this thread
  int pn[PI], poct[PI], pb[PI], poy[PI], pox[PI];
  bool item_ok[PI], pix_ok[PI];
  const T* in_b[PI];
#pragma unroll
  for (int r = 0; r < PI; ++r) {
    const int item = tid + r * NT;
    item_ok[r] = item < NITEM;
    pn[r] = item_ok[r] ? item / kOct : 0;
    poct[r] = item % kOct;
    const int64_t my_pix = pix0 + pn[r];
    pix_ok[r] = item_ok[r] && my_pix < npix;
    pb[r] = poy[r] = pox[r] = 0;
    if (pix_ok[r]) {
      pox[r] = (int)(my_pix % p.ow);
      poy[r] = (int)((my_pix / p.ow) % p.oh);
      pb[r] = (int)(my_pix / ((int64_t)p.ow * p.oh));
    }
    in_b[r] = input_cl + (int64_t)pb[r] * p.H * p.W * p.C + (int64_t)g * p.ICg;
  }
  // this thread's pieces of the A slab
  int a_m[AV], a_oct[AV];
  bool a_ok[AV];
#pragma unroll
  for (int e = 0; e < AV; ++e) {
    const int piece = tid + e * NT;
    a_ok[e] = piece < AQ && oc0 + piece % BM < OCg_pad;
    a_m[e] = a_ok[e] ? piece % BM : 0;
    a_oct[e] = a_ok[e] ? piece / BM : 0;
  }

  f32x16v acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  struct Stage {
    uint4 cv[PI][4];   // the four corner octets of the items
    uint4 av[AV];      // the weight pieces
    // the weights and the mask of the slab's tap, as genuine PAIRS: a scalar broadcast by op_sel makes v_pk_mul_f32 name the
    // odd neighbour register as well, and when that neighbour holds a still-pending load (the next segment's raw offsets)
    // the slab loop waits for vmcnt(0) on a value it never reads
    f32x2 w1[PI], w2[PI], w3[PI], w4[PI], wm[PI];
    int kmax;          // channels of the slab (a multiple of 8)
  };
  Stage stg[D];

  Tap<float> tap_cur[PI];
  TapRaw<T> raw_next[PI];
#pragma unroll
  for (int r = 0; r < PI; ++r) {
    tap_cur[r].o1 = tap_cur[r].o2 = tap_cur[r].o3 = tap_cur[r].o4 = 0;
    tap_cur[r].w1 = tap_cur[r].w2 = tap_cur[r].w3 = tap_cur[r].w4 = 0.f;
    tap_cur[r].m = 0.f;
    raw_next[r] = tap_raw_identity<T>();
  }

  // slab iterator of the ISSUE side: tap (outer) x offset-group segment x ic0 (inner); saturates at the last slab
  int i_tap = 0, i_ic = 0, i_seg_end = 0;
  bool i_have = KK > 0 && p.ICg > 0;
  int total = 0;
  {
    int per_tap = 0;
    for (int ic = 0; ic < p.ICg;) {
      const int og = (g * p.ICg + ic) / p.cpog;
      const int end = min(p.ICg, (og + 1) * p.cpog - g * p.ICg);
      per_tap += (end - ic + BK - 1) / BK;
      ic = end;
    }
    total = per_tap * KK;
  }
  // addresses: a UNIFORM base (the group's channel run of the slab) + a 32-bit per-lane byte offset (pixel corner, octet),
  // re-formed per segment only — the launcher sends inputs of 4 GiB and more to the round-3 kernel
  auto corner_off = [&](int r, int o) { return (unsigned)((((int64_t)pb[r] * p.H * p.W + o) * p.C + 8 * poct[r]) * 2); };
  unsigned coff[PI][4];   // of the current segment's tap (lanes without a pixel: corner 0 of image 0)
#pragma unroll
  for (int r = 0; r < PI; ++r) coff[r][0] = coff[r][1] = coff[r][2] = coff[r][3] = corner_off(r, 0);
  unsigned a_off[AV];
#pragma unroll
  for (int e = 0; e < AV; ++e) a_off[e] = (unsigned)((a_oct[e] * OCg_pad + a_m[e]) * 16);
  auto fetch_raw = [&](int tap, int ic) {
#pragma unroll
    for (int r = 0; r < PI; ++r)
      if (pix_ok[r]) load_tap_raw<T>(raw_next[r], p, offset, mask, pb[r], (g * p.ICg + ic) / p.cpog, tap, poy[r], pox[r]);
  };
  auto begin_segment = [&](int tap, int ic) {
    const int og = (g * p.ICg + ic) / p.cpog;
    i_seg_end = min(p.ICg, (og + 1) * p.cpog - g * p.ICg);
#pragma unroll
    for (int r = 0; r < PI; ++r)
      if (pix_ok[r]) {
        tap_from_raw<T, float>(tap_cur[r], p, raw_next[r], tap, poy[r], pox[r]);
        coff[r][0] = corner_off(r, tap_cur[r].o1);
        coff[r][1] = corner_off(r, tap_cur[r].o2);
        coff[r][2] = corner_off(r, tap_cur[r].o3);
        coff[r][3] = corner_off(r, tap_cur[r].o4);
      }
    int t2 = tap, i2 = i_seg_end;
    if (i2 >= p.ICg) {
      i2 = 0;
      t2 = tap + 1;
    }
    if (t2 < KK) fetch_raw(t2, i2);
  };
  auto issue = [&](Stage& sg) {
    const int kmax = min(BK, i_seg_end - i_ic);
    sg.kmax = kmax;
    const char* gbase = reinterpret_cast<const char*>(input_cl + (int64_t)g * p.ICg + i_ic);
    const char* wbase = reinterpret_cast<const char*>(wt8 + ((((int64_t)g * KK + i_tap) * noct + i_ic / 8) * OCg_pad + oc0) * 8);
    const bool whole = kmax == BK;   // uniform; both arms issue the same loads (the counted waits stay exact)
#pragma unroll
    for (int r = 0; r < PI; ++r) {
      if (whole) {
#pragma unroll
        for (int k = 0; k < 4; ++k) sg.cv[r][k] = *reinterpret_cast<const uint4*>(gbase + coff[r][k]);
      } else {
        const unsigned back = 8 * poct[r] < kmax ? 0u : 16u * poct[r];   // a partial slab: octets past its end read octet 0
#pragma unroll
        for (int k = 0; k < 4; ++k) sg.cv[r][k] = *reinterpret_cast<const uint4*>(gbase + (coff[r][k] - back));
      }
      sg.w1[r] = f32x2{tap_cur[r].w1, tap_cur[r].w1};
      sg.w2[r] = f32x2{tap_cur[r].w2, tap_cur[r].w2};
      sg.w3[r] = f32x2{tap_cur[r].w3, tap_cur[r].w3};
      sg.w4[r] = f32x2{tap_cur[r].w4, tap_cur[r].w4};
      sg.wm[r] = f32x2{tap_cur[r].m, tap_cur[r].m};
      // (opaque, or the backend folds the pairs back into op_sel broadcasts of one register)
      asm volatile("" : "+v"(sg.w1[r]), "+v"(sg.w2[r]), "+v"(sg.w3[r]), "+v"(sg.w4[r]), "+v"(sg.wm[r]));
    }
    if (whole) {
#pragma unroll
      for (int e = 0; e < AV; ++e) sg.av[e] = *reinterpret_cast<const uint4*>(wbase + a_off[e]);
    } else {
#pragma unroll
      for (int e = 0; e < AV; ++e) {
        const unsigned back = 8 * a_oct[e] < kmax ? 0u : (unsigned)(a_oct[e] * OCg_pad * 16);
        sg.av[e] = *reinterpret_cast<const uint4*>(wbase + (a_off[e] - back));
      }
    }
    if (i_have) {
      int n_tap = i_tap, n_ic = i_ic + BK;
      bool n_have = true;
      if (n_ic >= i_seg_end) {
        n_ic = i_seg_end;
        if (n_ic >= p.ICg) {
          n_ic = 0;
          n_tap = i_tap + 1;
          if (n_tap >= KK) n_have = false;
        }
        if (n_have) begin_segment(n_tap, n_ic);
      }
      if (n_have) {
        i_tap = n_tap;
        i_ic = n_ic;
      }
      i_have = n_have;
    }
  };
  // mask x bilinear sum of one dword's two channels, as packed fp32 (v_pk_mul_f32 / v_pk_add_f32: the same roundings, in the same
  // order, as the scalar expression of the other kernels — the commit is VALU-bound: 4 cycles per wave64 instruction)
  auto blend2 = [&](f32x2 w1, f32x2 w2, f32x2 w3, f32x2 w4, f32x2 m, unsigned u1, unsigned u2, unsigned u3, unsigned u4) {
    const f32x2 r = m * (w1 * unpack16x2<T>(u1) + w2 * unpack16x2<T>(u2) + w3 * unpack16x2<T>(u3) + w4 * unpack16x2<T>(u4));
    return (unsigned)to16<T>(r.x) | ((unsigned)to16<T>(r.y) << 16);
  };
  auto commit = [&](const Stage& sg, int buf) {
    const bool whole = sg.kmax == BK;   // uniform: a whole slab needs no zero-fill (columns of pixels past the end are never stored)
#pragma unroll
    for (int r = 0; r < PI; ++r) {
      if constexpr (NITEM != PI * NT) {   // (no exec-mask branch otherwise: the blend shares a basic block with the MFMAs)
        if (!item_ok[r]) continue;
      }
      const unsigned* c0 = reinterpret_cast<const unsigned*>(&sg.cv[r][0]);
      const unsigned* c1 = reinterpret_cast<const unsigned*>(&sg.cv[r][1]);
      const unsigned* c2 = reinterpret_cast<const unsigned*>(&sg.cv[r][2]);
      const unsigned* c3 = reinterpret_cast<const unsigned*>(&sg.cv[r][3]);
      unsigned packed[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) packed[d] = blend2(sg.w1[r], sg.w2[r], sg.w3[r], sg.w4[r], sg.wm[r], c0[d], c1[d], c2[d], c3[d]);
      if (!whole) {
        if (!(8 * poct[r] < sg.kmax)) packed[0] = packed[1] = packed[2] = packed[3] = 0u;
      }
      *reinterpret_cast<uint4*>(&Bs[buf][pn[r]][8 * poct[r]]) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    }
#pragma unroll
    for (int e = 0; e < AV; ++e) {
      const int piece = tid + e * NT;
      if constexpr (AQ != AV * NT) {
        if (piece >= AQ) continue;
      }
      unsigned short* dst = &As[buf][piece % BM][8 * (piece / BM)];
      if (whole) {   // (rows past OCg are zero in the re-laid-out weights)
        *reinterpret_cast<uint4*>(dst) = sg.av[e];
      } else {
        const bool zero = !(8 * (piece / BM) < sg.kmax);
        *reinterpret_cast<uint4*>(dst) = make_uint4(zero ? 0u : sg.av[e].x, zero ? 0u : sg.av[e].y, zero ? 0u : sg.av[e].z, zero ? 0u : sg.av[e].w);
      }
    }
  };
  const int kq = lane >> 5, l31 = lane & 31;
  auto multiply = [&](int buf) {
#pragma unroll
    for (int step = 0; step < BK / 16; ++step) {
      uint4 a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const uint4*>(&As[buf][(wm * MI + mi) * 32 + l31][16 * step + kq * 8]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const uint4*>(&Bs[buf][(wn * NI + ni) * 32 + l31][16 * step + kq * 8]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mfma16<T>(a[mi], b[ni], acc[mi][ni]);
    }
  };

  // One barrier per slab.  Between two barriers a wave multiplies slab s (LDS buffer s & 1) AND blends slab s + 1 into the
  // other buffer — one basic block, so the matrix pipe runs under the blend's VALU work instead of after it (the counters of
  // the first pipelined version: VALU 43 %, MFMA 17 %, nothing else above 50 % — the phases of the 8 lock-stepped waves did not
  // overlap; profiles/r04_pmc_sq_dcn_bf16_clp_summary.txt).
  if (total > 0) {
    fetch_raw(0, 0);
    begin_segment(0, 0);
#pragma unroll
    for (int u = 0; u < D; ++u) issue(stg[u]);
    int buf = 0;
    commit(stg[0], 0);
    auto step = [&](Stage& refill, const Stage& next) {
      __syncthreads();
      issue(refill);            // its slab went to LDS in the previous step
      multiply(buf);
      commit(next, buf ^ 1);
      buf ^= 1;
    };
    const int steps = total - 1, full = steps / D * D;
    for (int s = 0; s < full; s += D) {
#pragma unroll
      for (int u = 0; u < D; ++u) step(stg[u], stg[(u + 1) % D]);
    }
#pragma unroll
    for (int u = 0; u < D - 1; ++u)
      if (full + u < steps) step(stg[u], stg[(u + 1) % D]);
    __syncthreads();
    multiply(buf);
  }

  float brow[MI][16];   // one batch of independent bias loads (see the fp32 kernel's epilogue)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      brow[mi][r] = ld(bias + g * p.OCg + min(oc0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq, p.OCg - 1));
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int64_t pix = pix0 + (wn * NI + ni) * 32 + l31;
    if (pix >= npix) continue;
    const int ox = (int)(pix % p.ow);
    const int oy = (int)((pix / p.ow) % p.oh);
    const int b = (int)(pix / ((int64_t)p.ow * p.oh));
    T* obase = out + ((int64_t)b * p.OC + (int64_t)g * p.OCg) * p.oh * p.ow + (int64_t)oy * p.ow + ox;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = oc0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
        if (oc < p.OCg) st(obase + (int64_t)oc * p.oh * p.ow, acc[mi][ni][r] + brow[mi][r]);
      }
    }
  }
}

// ------------------------------------------------------------------ depthwise forward (groups == C == OC, 3x3)
// Reference shape: cuda/deform_conv2d_kernel.cu:136-209 writes columns[C*9, B*oh*ow] and runs C GEMMs of 1 x 9.  The
// generic direct kernel above re-derives the tap geometry per (pixel, channel) and gathers 36 scattered dwords per output
// from L2 (0.249 ms at config 4 = 0.03 of the HBM roofline: the texture path retires ~1 scattered lane per clock).
// Here:  workgroup = (image, 8 x 64 output tile, offset group, channel range).  lane = pixel: the 9 taps of a pixel
// {LDS offset of the top-left corner, lw, (1-lh)*mask, lh*mask} are computed ONCE and kept in registers for every channel
// of the range; the input window of the tile (+/- kHalo px of offset reach, zero-filled outside the image: the
// reference's zero-padding semantics, cpu/deform_conv2d_kernel.cpp:95-132, become plain reads) is staged per chunk of
// channels in LDS, double-buffered through registers, and the 4 corners of a tap are two LDS pair reads.  A tap that leaves
// the staged window (offset beyond the halo) is flagged per lane and served from global memory with the exact
// reference arithmetic — correct for any offset, fast for offsets within the halo.  Stores are 256-byte rows.
constexpr int kDwTH = 8, kDwTW = 64, kDwThreads = kDwTH * kDwTW, kDwHalo = 4, kDwMaxStage = 12;

struct DwGeom {
  int tile_h, tile_w, tile_sz;   // staged input window of one output tile (rows, cols, elements)
  int ntx, nty;                  // output tiles
  int CB, nstage;                // channels per chunk, staged elements per thread and chunk
  int csplit, cper;              // channel ranges per offset group, channels per range
};

// TWC: the staged window's row pitch as a compile-time constant (75 for stride 1 / dilation 1: the second tap row is then an
// immediate offset of the LDS pair read) or 0 = run-time pitch.
template <typename T, int TWC>
__global__ __launch_bounds__(kDwThreads, 4) void dcn_fwd_depthwise3x3(const T* __restrict__ input, const T* __restrict__ weight,
                                                                     const T* __restrict__ offset, const T* __restrict__ mask,
                                                                     const T* __restrict__ bias, T* __restrict__ out,
                                                                     DcnParams p, DwGeom g) {
  constexpr int KK = 9;
  extern __shared__ __attribute__((aligned(16))) float dw_lds[];   // [2][CB][tile_h][tile_w]
  const int tile_w = TWC > 0 ? TWC : g.tile_w;
  const int tid = threadIdx.x;
  const int px = tid & (kDwTW - 1), py = tid >> 6;
  const int tx0 = (blockIdx.x % g.ntx) * kDwTW, ty0 = (blockIdx.x / g.ntx) * kDwTH;
  const int og = blockIdx.y / g.csplit, cs = blockIdx.y - og * g.csplit;
  const int b = blockIdx.z;
  const int c_begin = og * p.cpog + cs * g.cper, c_end = min((og + 1) * p.cpog, c_begin + g.cper);
  if (c_begin >= c_end) return;
  const int ox = tx0 + px, oy = ty0 + py;
  const bool live = ox < p.ow && oy < p.oh;
  const int in_y0 = ty0 * p.sh - p.ph - kDwHalo, in_x0 = tx0 * p.sw - p.pw - kDwHalo;
  const int64_t plane = (int64_t)p.H * p.W, oplane = (int64_t)p.oh * p.ow;

  // ---- tap geometry of this lane's pixel, once for all channels
  int toff[KK];
  float lw[KK], hhm[KK], lhm[KK];
  unsigned far = 0;
  // the 27 raw values first, every load unconditional (pixel clamped, the mask read from the offset tensor when there is none
  // and replaced by 1 below): written per tap under `if (live)` / `if (use_mask)` the nine taps were nine dependent round
  // trips in front of the first chunk (s_waitcnt vmcnt(0) per tap in the ISA)
  T raw_h[KK], raw_w[KK], raw_m[KK];
  {
    const int64_t pixc = (int64_t)min(oy, p.oh - 1) * p.ow + min(ox, p.ow - 1);
    const T* optr = offset + ((int64_t)(b * p.ogroups + og) * 2 * KK) * oplane + pixc;
    const T* mptr = p.use_mask ? mask + ((int64_t)(b * p.ogroups + og) * KK) * oplane + pixc : optr;
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      raw_h[t] = optr[(int64_t)(2 * t) * oplane];
      raw_w[t] = optr[(int64_t)(2 * t + 1) * oplane];
      raw_m[t] = mptr[(int64_t)t * oplane];
    }
  }
#pragma unroll
  for (int t = 0; t < KK; ++t) {
    toff[t] = 0;
    lw[t] = hhm[t] = lhm[t] = 0.f;
    if (live) {
      const int i = t / 3, j = t - 3 * i;
      const float off_h = ld(&raw_h[t]), off_w = ld(&raw_w[t]);
      const float m = p.use_mask ? (float)ld(&raw_m[t]) : 1.f;
      const float y = (float)(oy * p.sh - p.ph) + (float)(i * p.dh) + off_h;
      const float x = (float)(ox * p.sw - p.pw) + (float)(j * p.dw) + off_w;
      if (!(y <= -1.f || (float)p.H <= y || x <= -1.f || (float)p.W <= x)) {   // else: the sample is zero (reference :99-101)
        const float fy = floorf(y), fx = floorf(x);
        const int ty = (int)fy - in_y0, tx = (int)fx - in_x0;
        if (ty >= 0 && ty + 1 < g.tile_h && tx >= 0 && tx + 1 < tile_w) {
          const float lh = y - fy;
          toff[t] = ty * tile_w + tx;
          lw[t] = x - fx;
          hhm[t] = (1.f - lh) * m;
          lhm[t] = lh * m;
        } else {
          far |= 1u << t;   // inside the image but outside the staged window
        }
      }
    }
  }
  const bool any_far = __ballot(far != 0) != 0ull;

  // ---- staging plan of this thread: element e = tid + i * threads of the chunk image [CB][tile_h][tile_w]
  int gsrc[kDwMaxStage];   // offset inside the chunk's first input plane (cb * plane + y * W + x), -1 = zero fill / nothing
#pragma unroll
  for (int i = 0; i < kDwMaxStage; ++i) {
    gsrc[i] = -1;
    const int e = tid + i * kDwThreads;
    if (i < g.nstage && e < g.CB * g.tile_sz) {
      const int cb = e / g.tile_sz, rem = e - cb * g.tile_sz;
      const int r = rem / tile_w, c = rem - r * tile_w;
      const int iy = in_y0 + r, ix = in_x0 + c;
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) gsrc[i] = cb * (int)plane + iy * p.W + ix;
    }
  }
  const int nchunks = (c_end - c_begin + g.CB - 1) / g.CB;
  // Round 4, from the ISA: (1) the staging loads were conditional (`stage = 0; if (inside) stage = load`), (2) the 16-bit weight
  // vector of a channel was loaded inside the channel loop — its wait is a wait for the whole prefetch issued in front of it —
  // and (3) the output stores of a chunk sat between the prefetch and the s_waitcnt vmcnt(0) of `park` (stores count in vmcnt
  // on this ISA: every chunk waited for its own write acknowledgements).  Now every load is unconditional (clamped address,
  // zeroed at park), the weight vectors of the NEXT chunk travel with its window, and a chunk's results stay in registers
  // until its successor is parked.
  constexpr int kCBMax = 8;   // (depthwise_geom caps CB at 8)
  T stage[kDwMaxStage];
  float wv_cur[kCBMax], wv_nxt[kCBMax];
#pragma unroll
  for (int cb = 0; cb < kCBMax; ++cb) wv_cur[cb] = wv_nxt[cb] = 0.f;
  auto fetch = [&](int chunk) {
    const int c0 = c_begin + chunk * g.CB;
    const T* src = input + ((int64_t)b * p.C + c0) * plane;
    const int lim = (c_end - c0) * (int)plane;   // channels past the range are not touched
#pragma unroll
    for (int i = 0; i < kDwMaxStage; ++i) {
      const int o = (gsrc[i] >= 0 && gsrc[i] < lim) ? gsrc[i] : 0;   // (element 0 of the chunk's first plane: always legal)
      stage[i] = src[o];
    }
    if constexpr (!std::is_same<T, float>::value) {
      // 16-bit types have no scalar load: ONE vector load per channel (lane t = weight t, lane 9 = bias), readlane broadcasts
      const int wl_ = min(tid & 63, KK);
#pragma unroll
      for (int cb = 0; cb < kCBMax; ++cb) {
        const int c = min(c0 + cb, c_end - 1);
        T raw = wl_ < KK ? weight[(int64_t)c * KK + wl_] : bias[c];
        wv_nxt[cb] = (float)ld(&raw);
      }
    }
  };
  auto park = [&](int chunk) {
    float* dst = dw_lds + (chunk & 1) * g.CB * g.tile_sz;
    const int lim = (c_end - (c_begin + chunk * g.CB)) * (int)plane;
#pragma unroll
    for (int i = 0; i < kDwMaxStage; ++i) {
      const int e = tid + i * kDwThreads;
      if (i < g.nstage && e < g.CB * g.tile_sz) dst[e] = (gsrc[i] >= 0 && gsrc[i] < lim) ? (float)ld(&stage[i]) : 0.f;
    }
#pragma unroll
    for (int cb = 0; cb < kCBMax; ++cb) wv_cur[cb] = wv_nxt[cb];
  };
  fetch(0);
  park(0);
  __syncthreads();
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    if (chunk + 1 < nchunks) fetch(chunk + 1);   // the next chunk's window is in flight while this one is used
    const float* tile = dw_lds + (chunk & 1) * g.CB * g.tile_sz;
    const int c0 = c_begin + chunk * g.CB, nc = min(g.CB, c_end - c0);
    float res[kCBMax];
#pragma unroll
    for (int cb = 0; cb < kCBMax; ++cb) {
      res[cb] = 0.f;
      if (cb < nc) {   // (uniform)
        const int c = c0 + cb;   // depthwise: output channel = input channel = weight row
        const float* tc = tile + cb * g.tile_sz;
        const T* wrow = weight + (int64_t)c * KK;
        // the 9 weights and the bias of this channel are wave-uniform: fp32 -> scalar loads; 16-bit: the prefetched vector
        const float wv = wv_cur[cb];
        auto wt = [&](int t) -> float {
          if constexpr (std::is_same<T, float>::value) return t < KK ? wrow[t] : bias[c];
          else return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wv), t));
        };
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < KK; ++t) {
          const float* q = tc + toff[t];
          const float top = __builtin_fmaf(lw[t], q[1], (1.f - lw[t]) * q[0]);
          const float bot = __builtin_fmaf(lw[t], q[tile_w + 1], (1.f - lw[t]) * q[tile_w]);
          const float val = __builtin_fmaf(lhm[t], bot, hhm[t] * top);
          acc = __builtin_fmaf(wt(t), val, acc);
        }
        if (any_far) {   // taps outside the staged window: the reference arithmetic on global memory
          const T* pl = input + ((int64_t)b * p.C + c) * plane;
          for (int t = 0; t < KK; ++t)
            if ((far >> t) & 1u) {
              Tap<float> tp;
              load_tap<T, float>(tp, p, offset, mask, b, og, t, oy, ox);
              acc = __builtin_fmaf(wt(t), sample_tap<T, float>(tp, pl), acc);
            }
        }
        res[cb] = acc + wt(KK);
      }
    }
    if (chunk + 1 < nchunks) park(chunk + 1);
    if (live) {
#pragma unroll
      for (int cb = 0; cb < kCBMax; ++cb)
        if (cb < nc) st(out + ((int64_t)b * p.OC + c0 + cb) * oplane + (int64_t)oy * p.ow + ox, res[cb]);
    }
    __syncthreads();
  }
}

// geometry of the depthwise kernel for a problem, or .CB = 0 when it does not apply
inline DwGeom depthwise_geom(const DcnParams& p, tvmi_dtype dt) {
  DwGeom g{};
  g.CB = 0;
  if (!(dt == TVMI_F32 || dt == TVMI_F16 || dt == TVMI_BF16)) return g;
  if (!(p.ICg == 1 && p.OCg == 1 && p.kh == 3 && p.kw == 3)) return g;
  g.tile_h = (kDwTH - 1) * p.sh + 2 * p.dh + 2 * kDwHalo + 2;
  g.tile_w = (kDwTW - 1) * p.sw + 2 * p.dw + 2 * kDwHalo + 2;
  g.tile_sz = g.tile_h * g.tile_w;
  const int64_t budget = 48 * 1024 / 4;   // floats of LDS per workgroup (3 workgroups per CU), two buffers
  const int64_t cb = std::min<int64_t>(std::min<int64_t>(budget / (2 * (int64_t)g.tile_sz), 8), p.cpog);
  if (cb < 1 || (int64_t)p.H * p.W * cb >= (1ll << 31)) return g;
  g.nstage = (int)ceil_div(cb * g.tile_sz, kDwThreads);
  if (g.nstage > kDwMaxStage) return g;
  g.ntx = (int)ceil_div(p.ow, kDwTW);
  g.nty = (int)ceil_div(p.oh, kDwTH);
  // enough workgroups to fill the chip ~3 times, but at least 16 channels per range so the tap set-up amortises
  const int64_t tiles = (int64_t)g.ntx * g.nty * p.B * p.ogroups;
  int64_t csplit = std::max<int64_t>(1, std::min<int64_t>(ceil_div(3 * 256, tiles), std::max<int64_t>(1, p.cpog / 16)));
  g.cper = (int)ceil_div(ceil_div(p.cpog, csplit), cb) * (int)cb;   // whole chunks per range
  g.csplit = (int)ceil_div(p.cpog, g.cper);
  if ((int64_t)g.csplit * p.ogroups > 65535 || p.B > 65535) return g;
  g.CB = (int)cb;
  return g;
}

// ------------------------------------------------------------------ depthwise forward, packed form (round 5)
// The kernel above is bound by issue slots, not by data (rocprofv3 r03/r04: waves waiting 65 %, LDS 30 %, VALU 23 %): per
// (channel, tap) it issues two LDS pair reads and eight scalar-width VALU ops, and its 780 workgroups are 12 more than
// the 768 it can have resident — a second, nearly empty round.  This form keeps the workgroup shape (lane = pixel, 8 x 64
// output tile, window of the tile in LDS, double-buffered through registers) and changes what a wave issues:
//   * the window holds FOUR channels interleaved ([y][x][4]): one ds_read_b128 fetches a corner of four channels, i.e. one
//     LDS instruction per (channel, tap) instead of two, and the same 16 bytes;
//   * the four bilinear corner weights of a tap (mask folded in) are per-lane constants kept as two register PAIRS
//     (c00, c01), (c10, c11); the blend of four channels is 8 packed ops (v_pk_mul_f32 / v_pk_fma_f32, two channels per
//     instruction, the corner weight broadcast from one half of its pair by op_sel) + 4 weight FMAs with a scalar
//     operand: 3 VALU instructions per (channel, tap) instead of 8;
//   * the channel ranges are sized so that every workgroup is resident at once (2 per CU at <= 128 VGPRs).
// Values: the blend is c00*q00 + c01*q01 + c10*q10 + c11*q11 with c = (1-lh | lh) * (1-lw | lw) * mask — the reference's
// own formula (cpu/deform_conv2d_kernel.cpp:117-131: w1..w4 * v1..v4), other rounding order than the kernel above.
// Needs whole chunks: channels per offset group % 4 == 0.
typedef float dw_f32x2 __attribute__((ext_vector_type(2)));
typedef float dw_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kDwPkCB = 4, kDwPkMaxStage = 4;

// d = broadcast(c.lo | c.hi) * q (+ d): the packed-fp32 ops take either half of a 64-bit source for both results (op_sel)
__device__ __forceinline__ dw_f32x2 pk_mul_bl(dw_f32x2 c, dw_f32x2 q) {
  dw_f32x2 d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(c), "v"(q));
  return d;
}
__device__ __forceinline__ dw_f32x2 pk_fma_bh(dw_f32x2 c, dw_f32x2 q, dw_f32x2 d) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(d) : "v"(c), "v"(q));
  return d;
}
__device__ __forceinline__ dw_f32x2 pk_fma_bl(dw_f32x2 c, dw_f32x2 q, dw_f32x2 d) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(d) : "v"(c), "v"(q));
  return d;
}

// NS: staged window pixels per thread (3 for the stride-1 / dilation-1 window of 19 x 80); TG: taps whose corner reads are issued
// together (3: 12 reads in flight, the register budget of the NS = 3 shape; 1 for the general shape)
template <typename T, int TWC, int NS, int TG>
__global__ __launch_bounds__(kDwThreads, 4) void dcn_fwd_depthwise3x3_pk(const T* __restrict__ input, const T* __restrict__ weight,
                                                                        const T* __restrict__ offset, const T* __restrict__ mask,
                                                                        const T* __restrict__ bias, T* __restrict__ out,
                                                                        DcnParams p, DwGeom g) {
  constexpr int KK = 9, CB = kDwPkCB;
  extern __shared__ __attribute__((aligned(16))) float dw_lds[];   // [2][tile_h][tile_w][4]
  dw_f32x4* const lds4 = reinterpret_cast<dw_f32x4*>(dw_lds);
  const int tile_w = TWC > 0 ? TWC : g.tile_w;
  const int tid = threadIdx.x;
  const int px = tid & (kDwTW - 1), py = tid >> 6;
  const int tx0 = (blockIdx.x % g.ntx) * kDwTW, ty0 = (blockIdx.x / g.ntx) * kDwTH;
  const int og = blockIdx.y / g.csplit, cs = blockIdx.y - og * g.csplit;
  const int b = blockIdx.z;
  const int c_begin = og * p.cpog + cs * g.cper, c_end = min((og + 1) * p.cpog, c_begin + g.cper);   // whole chunks of 4
  if (c_begin >= c_end) return;
  const int ox = tx0 + px, oy = ty0 + py;
  const bool live = ox < p.ow && oy < p.oh;
  const int in_y0 = ty0 * p.sh - p.ph - kDwHalo, in_x0 = tx0 * p.sw - p.pw - kDwHalo;
  const int64_t plane = (int64_t)p.H * p.W, oplane = (int64_t)p.oh * p.ow;

  // ---- tap geometry of this lane's pixel, once for all channels: window pixel of the top-left corner + 4 corner weights
  unsigned tpack[(KK + 1) / 2];   // BYTE offset of a tap's top-left corner in a window buffer (< 32 KB), taps 2k | 2k+1 in the halves
  dw_f32x2 ctop[KK], cbot[KK];   // (c00, c01), (c10, c11)
  unsigned far = 0;
  T raw_h[KK], raw_w[KK], raw_m[KK];
  {
    const int64_t pixc = (int64_t)min(oy, p.oh - 1) * p.ow + min(ox, p.ow - 1);
    const T* optr = offset + ((int64_t)(b * p.ogroups + og) * 2 * KK) * oplane + pixc;
    const T* mptr = p.use_mask ? mask + ((int64_t)(b * p.ogroups + og) * KK) * oplane + pixc : optr;
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      raw_h[t] = optr[(int64_t)(2 * t) * oplane];
      raw_w[t] = optr[(int64_t)(2 * t + 1) * oplane];
      raw_m[t] = mptr[(int64_t)t * oplane];
    }
  }
#pragma unroll
  for (int k = 0; k < (KK + 1) / 2; ++k) tpack[k] = 0u;
#pragma unroll
  for (int t = 0; t < KK; ++t) {
    ctop[t] = cbot[t] = dw_f32x2{0.f, 0.f};
    if (live) {
      const int i = t / 3, j = t - 3 * i;
      const float off_h = ld(&raw_h[t]), off_w = ld(&raw_w[t]);
      const float m = p.use_mask ? (float)ld(&raw_m[t]) : 1.f;
      const float y = (float)(oy * p.sh - p.ph) + (float)(i * p.dh) + off_h;
      const float x = (float)(ox * p.sw - p.pw) + (float)(j * p.dw) + off_w;
      if (!(y <= -1.f || (float)p.H <= y || x <= -1.f || (float)p.W <= x)) {   // else: the sample is zero (reference :99-101)
        const float fy = floorf(y), fx = floorf(x);
        const int ty = (int)fy - in_y0, tx = (int)fx - in_x0;
        if (ty >= 0 && ty + 1 < g.tile_h && tx >= 0 && tx + 1 < tile_w) {
          const float lh = y - fy, lw = x - fx;
          const float hh = 1.f - lh, hw = 1.f - lw;
          tpack[t >> 1] |= (unsigned)((ty * tile_w + tx) * 16) << ((t & 1) * 16);
          ctop[t] = dw_f32x2{hh * hw * m, hh * lw * m};
          cbot[t] = dw_f32x2{lh * hw * m, lh * lw * m};
        } else {
          far |= 1u << t;   // inside the image but outside the staged window
        }
      }
    }
  }
  const bool any_far = __ballot(far != 0) != 0ull;

  // ---- staging plan: window pixel e = tid + i * threads, its four channels by one thread (4 plane loads -> one 16-byte LDS store)
  int gsrc[NS];   // y * W + x inside a plane, -1 = outside the image (zero) / no such pixel
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    gsrc[i] = -1;
    const int e = tid + i * kDwThreads;
    if (e < g.tile_sz) {
      const int r = e / tile_w, c = e - r * tile_w;
      const int iy = in_y0 + r, ix = in_x0 + c;
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) gsrc[i] = iy * p.W + ix;
    }
  }
  const int nchunks = (c_end - c_begin) / CB;
  T stage[NS][CB];
  float wv_cur = 0.f, wv_nxt = 0.f;   // lane cb * 16 + t: weight t (t < 9) / bias (t = 9) of channel c0 + cb
  auto fetch = [&](int chunk) {
    const int c0 = c_begin + chunk * CB;
    const T* src = input + ((int64_t)b * p.C + c0) * plane;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int o = gsrc[i] >= 0 ? gsrc[i] : 0;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) stage[i][cb] = src[(int64_t)cb * plane + o];   // every load unconditional
    }
    // the 9 weights + the bias of the chunk's four channels: ONE vector load (lane cb * 16 + t) that travels with the window;
    // v_readlane broadcasts them in the tap loop.  (Scalar loads inside the loop share lgkmcnt with the LDS reads and return
    // out of order: every wait would have to drain the LDS queue.)
    {
      const int l = tid & 63, cb = l >> 4, t = min(l & 15, KK);
      T raw = t < KK ? weight[(int64_t)(c0 + cb) * KK + t] : bias[c0 + cb];
      wv_nxt = (float)ld(&raw);
    }
  };
  auto park = [&](int chunk) {
    dw_f32x4* dst = lds4 + (chunk & 1) * g.tile_sz;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int e = tid + i * kDwThreads;
      if (e < g.tile_sz) {
        const bool in = gsrc[i] >= 0;
        dw_f32x4 v;
        v.x = in ? (float)ld(&stage[i][0]) : 0.f;
        v.y = in ? (float)ld(&stage[i][1]) : 0.f;
        v.z = in ? (float)ld(&stage[i][2]) : 0.f;
        v.w = in ? (float)ld(&stage[i][3]) : 0.f;
        dst[e] = v;
      }
    }
    wv_cur = wv_nxt;
  };
  fetch(0);
  park(0);
  __syncthreads();
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    if (chunk + 1 < nchunks) fetch(chunk + 1);   // the next chunk's window is in flight while this one is used
    const dw_f32x4* tile = lds4 + (chunk & 1) * g.tile_sz;
    const int c0 = c_begin + chunk * CB;
    auto wt = [&](int cb, int t) -> float {
      return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wv_cur), cb * 16 + t));
    };
    float acc[CB] = {0.f, 0.f, 0.f, 0.f};
    // three taps at a time: their 12 corner reads are issued back to back (the LDS reaches its rate only with many reads in
    // flight per wait — with two per wait, as the compiler scheduled the plain loop, the kernel sat at 0.045 ms), then blended
#pragma unroll
    for (int t0 = 0; t0 < KK; t0 += TG) {
      dw_f32x4 q[TG][4];
#pragma unroll
      for (int u = 0; u < TG; ++u) {
        const int t = t0 + u;
        const dw_f32x4* qp = reinterpret_cast<const dw_f32x4*>(reinterpret_cast<const char*>(tile) + ((tpack[t >> 1] >> ((t & 1) * 16)) & 0xffffu));
        q[u][0] = qp[0];
        q[u][1] = qp[1];
        q[u][2] = qp[tile_w];
        q[u][3] = qp[tile_w + 1];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < TG; ++u) {
        const int t = t0 + u;
        dw_f32x2 lo = pk_mul_bl(ctop[t], q[u][0].xy), hi = pk_mul_bl(ctop[t], q[u][0].zw);
        lo = pk_fma_bh(ctop[t], q[u][1].xy, lo);
        hi = pk_fma_bh(ctop[t], q[u][1].zw, hi);
        lo = pk_fma_bl(cbot[t], q[u][2].xy, lo);
        hi = pk_fma_bl(cbot[t], q[u][2].zw, hi);
        lo = pk_fma_bh(cbot[t], q[u][3].xy, lo);
        hi = pk_fma_bh(cbot[t], q[u][3].zw, hi);
        acc[0] = __builtin_fmaf(wt(0, t), lo.x, acc[0]);
        acc[1] = __builtin_fmaf(wt(1, t), lo.y, acc[1]);
        acc[2] = __builtin_fmaf(wt(2, t), hi.x, acc[2]);
        acc[3] = __builtin_fmaf(wt(3, t), hi.y, acc[3]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (any_far) {   // taps outside the staged window: the reference arithmetic on global memory
      for (int t = 0; t < KK; ++t)
        if ((far >> t) & 1u) {
          Tap<float> tp;
          load_tap<T, float>(tp, p, offset, mask, b, og, t, oy, ox);
          const float w0 = wt(0, t), w1 = wt(1, t), w2 = wt(2, t), w3 = wt(3, t);   // (static indices: a run-time one puts acc[] into scratch)
          const T* pl = input + ((int64_t)b * p.C + c0) * plane;
          acc[0] = __builtin_fmaf(w0, sample_tap<T, float>(tp, pl), acc[0]);
          acc[1] = __builtin_fmaf(w1, sample_tap<T, float>(tp, pl + plane), acc[1]);
          acc[2] = __builtin_fmaf(w2, sample_tap<T, float>(tp, pl + 2 * plane), acc[2]);
          acc[3] = __builtin_fmaf(w3, sample_tap<T, float>(tp, pl + 3 * plane), acc[3]);
        }
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[cb] += wt(cb, KK);
    if (chunk + 1 < nchunks) park(chunk + 1);
    if (live) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) st(out + ((int64_t)b * p.OC + c0 + cb) * oplane + (int64_t)oy * p.ow + ox, acc[cb]);
    }
    __syncthreads();
  }
}

// geometry of the packed depthwise kernel (.CB = 0: does not apply): four interleaved channels per chunk, two window buffers,
// two workgroups per CU, every workgroup resident at once where the problem allows it
inline DwGeom depthwise_pk_geom(const DcnParams& p, tvmi_dtype dt) {
  DwGeom g{};
  g.CB = 0;
  if (!(dt == TVMI_F32 || dt == TVMI_F16 || dt == TVMI_BF16)) return g;
  if (!(p.ICg == 1 && p.OCg == 1 && p.kh == 3 && p.kw == 3) || p.cpog % kDwPkCB) return g;
  g.tile_h = (kDwTH - 1) * p.sh + 2 * p.dh + 2 * kDwHalo + 2;
  // row pitch = a multiple of 16 pixels (16 x 16 bytes = all 64 banks): a ds_read_b128 is served in groups of 16 lanes, whose
  // pixels then keep distinct bank slots whatever row their offsets send them to; what is left are the collisions of the
  // horizontal jitter (simulated with N(0,1) offsets: 2.37 instead of 2.66 LDS cycles per group at pitch 75)
  g.tile_w = ((kDwTW - 1) * p.sw + 2 * p.dw + 2 * kDwHalo + 2 + 15) / 16 * 16;
  g.tile_sz = g.tile_h * g.tile_w;
  if ((size_t)2 * g.tile_sz * kDwPkCB * sizeof(float) > (size_t)64 * 1024) return g;   // two workgroups per CU
  if ((int64_t)p.H * p.W >= (1ll << 31) / 8) return g;
  g.nstage = (int)ceil_div(g.tile_sz, kDwThreads);
  if (g.nstage > kDwPkMaxStage) return g;
  g.ntx = (int)ceil_div(p.ow, kDwTW);
  g.nty = (int)ceil_div(p.oh, kDwTH);
  const int64_t tiles = (int64_t)g.ntx * g.nty * p.B * p.ogroups;
  const int64_t slots = 2 * 256;
  int64_t csplit = std::max<int64_t>(1, std::min<int64_t>(slots / std::max<int64_t>(tiles, 1), std::max<int64_t>(1, p.cpog / 16)));
  g.cper = (int)ceil_div(ceil_div(p.cpog, csplit), kDwPkCB) * kDwPkCB;   // whole chunks per range
  g.csplit = (int)ceil_div(p.cpog, g.cper);
  if ((int64_t)g.csplit * p.ogroups > 65535 || p.B > 65535) return g;
  g.CB = kDwPkCB;
  return g;
}

int fill_params(DcnParams& p, int64_t B, int64_t C, int64_t H, int64_t W, int64_t OC, int64_t kh, int64_t kw,
                int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw, int64_t groups,
                int64_t ogroups, int use_mask) {
  if (int e = fill_params_common(p, B, C, H, W, OC, kh, kw, sh, sw, ph, pw, dh, dw, groups, ogroups, use_mask)) return e;
  p.xcd_tiles = g_xcd_tiles.load(std::memory_order_relaxed);
  return 0;
}

inline bool use_mfma(const DcnParams& p, tvmi_dtype dt) {
  return dt == TVMI_F32 && p.OCg >= 16 && p.ICg >= 4 && p.W >= 2;  // W >= 2: the corner pairs are 8-byte loads
}
inline bool use_mfma16(const DcnParams& p, tvmi_dtype dt) {
  return (dt == TVMI_F16 || dt == TVMI_BF16) && p.OCg >= 16 && p.ICg >= 4 && p.W >= 2;
}
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline bool can_gather_channels_last(const DcnParams& p, tvmi_dtype dt) {   // 16-byte octets of 16-bit channels
  return use_mfma16(p, dt) && p.C % 8 == 0 && p.ICg % 8 == 0 && p.cpog % 8 == 0;
}
inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

template <typename T, int WM, int WN, int MI, int NI, int BK>
int launch_cl16(const T* input_cl, const T* wt8, const T* offset, const T* mask, const T* bias, T* out, const DcnParams& p,
                int OCg_pad, hipStream_t s) {
  constexpr int NT = 64 * WM * WN, BM = 32 * MI * WM, BN = 32 * NI * WN;
  constexpr size_t lds = (size_t)2 * (BM + BN) * (BK + 8) * sizeof(unsigned short);
  auto kern = dcn_fwd_mfma_16_cl<T, WM, WN, MI, NI, BK>;
  if (lds > 64 * 1024) {
    static bool attr_set[64] = {};  // per instantiation and device; racing threads set the same value
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return set_error((int)hipErrorInvalidValue, "deform_conv2d: cannot reserve the LDS slabs");
      if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
  }
  const int64_t npix = (int64_t)p.B * p.oh * p.ow;
  kern<<<dim3((unsigned)ceil_div(npix, BN), (unsigned)ceil_div(p.OCg, BM), (unsigned)p.groups), dim3(NT), lds, s>>>(
      input_cl, wt8, offset, mask, bias, out, p, OCg_pad);
  return 0;
}
template <typename T, int WM, int WN, int MI, int NI, int BK, int D, int WPE = 1>
int launch_clp16(const T* input_cl, const T* wt8, const T* offset, const T* mask, const T* bias, T* out, const DcnParams& p,
                 int OCg_pad, hipStream_t s) {
  constexpr int NT = 64 * WM * WN, BM = 32 * MI * WM, BN = 32 * NI * WN;
  constexpr size_t lds = (size_t)2 * (BM + BN) * (BK + 8) * sizeof(unsigned short);
  auto kern = dcn_fwd_mfma_16_clp<T, WM, WN, MI, NI, BK, D, WPE>;
  if (lds > 64 * 1024) {
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return set_error((int)hipErrorInvalidValue, "deform_conv2d: cannot reserve the LDS slabs");
      if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
  }
  const int64_t npix = (int64_t)p.B * p.oh * p.ow;
  kern<<<dim3((unsigned)ceil_div(npix, BN), (unsigned)ceil_div(p.OCg, BM), (unsigned)p.groups), dim3(NT), lds, s>>>(
      input_cl, wt8, offset, mask, bias, out, p, OCg_pad);
  return 0;
}
inline dim3 grid1d(int64_t total) { return dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 1 << 20)); }

}  // namespace

int set_dcn_option(const char* name, int64_t value) {
  if (std::strcmp(name, "dcn.channels_last_gather") == 0) {
    g_cl_gather.store(value != 0, std::memory_order_relaxed);
    return 0;
  }
  if (std::strcmp(name, "dcn.cl_variant") == 0) {
    g_cl_pipelined.store(value != 0, std::memory_order_relaxed);
    return 0;
  }
  if (std::strcmp(name, "dcn.xcd_tiles") == 0) {
    g_xcd_tiles.store(value != 0, std::memory_order_relaxed);
    return 0;
  }
  return -1;
}

int get_dcn_option(const char* name, int64_t* value) {
  if (std::strcmp(name, "dcn.channels_last_gather") == 0) {
    *value = g_cl_gather.load(std::memory_order_relaxed) ? 1 : 0;
    return 0;
  }
  if (std::strcmp(name, "dcn.cl_variant") == 0) {
    *value = g_cl_pipelined.load(std::memory_order_relaxed);
    return 0;
  }
  if (std::strcmp(name, "dcn.xcd_tiles") == 0) {
    *value = g_xcd_tiles.load(std::memory_order_relaxed);
    return 0;
  }
  return -1;
}
}  // namespace tvmi

using namespace tvmi;

extern "C" size_t tvmi_deform_conv2d_workspace_bytes(tvmi_dtype dt, int64_t C, int64_t OC, int64_t kh, int64_t kw,
                                                     int64_t groups) {
  if (!(dt == TVMI_F32 || dt == TVMI_F16 || dt == TVMI_BF16) || groups <= 0 || C <= 0 || OC <= 0) return 0;
  const int ICg_pad = round_up((int)(C / groups), kBK), OCg_pad = round_up((int)(OC / groups), 64);
  return (size_t)groups * kh * kw * ICg_pad * OCg_pad * (dt == TVMI_F32 ? sizeof(float) : 2);   // the re-laid-out weights
}

extern "C" size_t tvmi_deform_conv2d_forward_workspace_bytes(tvmi_dtype dt, int64_t B, int64_t C, int64_t H, int64_t W,
                                                             int64_t OC, int64_t kh, int64_t kw, int64_t groups,
                                                             int64_t offset_groups) {
  const size_t weights = tvmi_deform_conv2d_workspace_bytes(dt, C, OC, kh, kw, groups);
  if (!(dt == TVMI_F16 || dt == TVMI_BF16) || groups <= 0 || offset_groups <= 0 || B <= 0 || H <= 0 || W <= 0) return weights;
  if (C % 8 != 0 || (C / groups) % 8 != 0 || (C / offset_groups) % 8 != 0) return weights;
  return align256(weights) + (size_t)B * C * H * W * 2;   // + the [B, H*W, C] copy the 16-bit MFMA kernel samples
}

extern "C" int tvmi_deform_conv2d_forward(const void* input, const void* weight, const void* offset,
                                          const void* mask, const void* bias, void* output, tvmi_dtype dt,
                                          int64_t B, int64_t C, int64_t H, int64_t W, int64_t OC, int64_t kh,
                                          int64_t kw, int64_t stride_h, int64_t stride_w, int64_t pad_h,
                                          int64_t pad_w, int64_t dil_h, int64_t dil_w, int64_t groups,
                                          int64_t offset_groups, int use_mask, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  DcnParams p;
  if (int e = fill_params(p, B, C, H, W, OC, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, groups,
                          offset_groups, use_mask))
    return e;
  const int64_t total = (int64_t)B * OC * p.oh * p.ow;
  if (total == 0) return 0;
  TVMI_CHECK_ARG(input && weight && offset && bias && output && (!use_mask || mask), "deform_conv2d: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (use_mfma(p, dt)) {
    const int ICg_pad = round_up(p.ICg, kBK), OCg_pad = round_up(p.OCg, 64);
    TVMI_CHECK_ARG(workspace && workspace_bytes >= tvmi_deform_conv2d_workspace_bytes(dt, C, OC, kh, kw, groups),
                   "deform_conv2d: workspace too small");
    float* wt = static_cast<float*>(workspace);
    const int64_t wtotal = (int64_t)p.groups * kh * kw * ICg_pad * OCg_pad;
    dcn_weight_relayout<<<grid1d(wtotal), dim3(256), 0, s>>>((const float*)weight, wt, p, ICg_pad, OCg_pad);
    const int64_t npix = (int64_t)B * p.oh * p.ow;
#define TVMI_DCN(WM, WN, MI, NI)                                                                                  \
  dcn_fwd_mfma_f32<WM, WN, MI, NI>                                                                                \
      <<<dim3((unsigned)ceil_div(npix, 32 * NI * WN), (unsigned)ceil_div(p.OCg, 32 * MI * WM), (unsigned)p.groups), \
         dim3(64 * WM * WN), 0, s>>>((const float*)input, wt, (const float*)offset, (const float*)mask,              \
                                     (const float*)bias, (float*)output, p, ICg_pad, OCg_pad)
    // 256x64 / 128x128 / 64x256 workgroup tiles; when the tiles fill the chip less than ~3 times, 8 smaller waves
    // per tile instead of 4 (more waves per SIMD to hide the gather latency behind)
    const int64_t ntiles = ceil_div(npix, p.OCg > 128 ? 64 : (p.OCg > 64 ? 128 : 256)) * ceil_div(p.OCg, p.OCg > 128 ? 256 : (p.OCg > 64 ? 128 : 64)) * p.groups;
    const bool eight = ntiles < 3 * 768;
    // (round 4: a 256 x 128 tile was measured as well — 0.435 ms against 0.413 at config 4, profiles/r04_dcn_variants.json)
    if (p.OCg > 128) {
      if (eight) TVMI_DCN(4, 2, 2, 1); else TVMI_DCN(4, 1, 2, 2);
    } else if (p.OCg > 64) {
      if (eight) TVMI_DCN(2, 4, 2, 1); else TVMI_DCN(2, 2, 2, 2);
    } else {
      if (eight) TVMI_DCN(1, 8, 2, 1); else TVMI_DCN(1, 4, 2, 2);
    }
#undef TVMI_DCN
  } else if (use_mfma16(p, dt)) {
    const int ICg_pad = round_up(p.ICg, kBK), OCg_pad = round_up(p.OCg, 64);
    TVMI_CHECK_ARG(workspace && workspace_bytes >= tvmi_deform_conv2d_workspace_bytes(dt, C, OC, kh, kw, groups),
                   "deform_conv2d: workspace too small");
    const int64_t wtotal = (int64_t)p.groups * kh * kw * ICg_pad * OCg_pad;
    const int64_t npix = (int64_t)B * p.oh * p.ow;
    const size_t cl_at = align256((size_t)wtotal * 2), cl_bytes = (size_t)B * C * H * W * 2;
    if (g_cl_gather.load(std::memory_order_relaxed) && can_gather_channels_last(p, dt) && workspace_bytes >= cl_at + cl_bytes) {
      void* in_cl = static_cast<char*>(workspace) + cl_at;
      const int64_t w8total = (int64_t)p.groups * kh * kw * p.ICg * OCg_pad;   // <= wtotal
#define TVMI_CLARGS(scalar_t) icl, w8, (const scalar_t*)offset, (const scalar_t*)mask, (const scalar_t*)bias, (scalar_t*)output, p, OCg_pad, s
#define TVMI_DCN16_CL(scalar_t)                                                                                        \
  do {                                                                                                                 \
    dcn_weight_relayout8<scalar_t><<<grid1d(w8total), dim3(256), 0, s>>>((const scalar_t*)weight, (scalar_t*)workspace, p, OCg_pad); \
    dcn_to_channels_last<scalar_t><<<dim3((unsigned)ceil_div(H * W, 32), (unsigned)ceil_div(C, 32), (unsigned)B), dim3(256), 0, s>>>( \
        (const scalar_t*)input, (scalar_t*)in_cl, (int)C, (int)(H * W));                                               \
    const scalar_t* icl = (const scalar_t*)in_cl;                                                                      \
    const scalar_t* w8 = (const scalar_t*)workspace;                                                                   \
    const bool pipelined = cl_bytes < ((size_t)1 << 32) && g_cl_pipelined.load(std::memory_order_relaxed);   /* 32-bit lane offsets */ \
    /* measured at config 4 (profiles/r04_dcn_variants.json): 256 x 128 tile, 8 waves of 64 x 64 — 0.098 ms against 0.125 ms   \
       for the round-3 kernel; 4 waves 0.104; 256 x 64 tiles 0.151; three stages 0.102.  OC = 128: 128 x 128 tile 0.096        \
       against 0.137 (round 3) and 0.150 (128 x 256 tile: 107 workgroups) */                                                 \
    if (p.OCg > 128 && pipelined) st = launch_clp16<scalar_t, 4, 2, 2, 2, 32, 2>(TVMI_CLARGS(scalar_t));              \
    else if (p.OCg > 64 && p.OCg <= 128 && pipelined) st = launch_clp16<scalar_t, 2, 4, 2, 1, 32, 2>(TVMI_CLARGS(scalar_t)); \
    else                                                                                                               \
    if (p.OCg > 128) st = launch_cl16<scalar_t, 4, 2, 2, 1, 32>(icl, w8, (const scalar_t*)offset, (const scalar_t*)mask, (const scalar_t*)bias, (scalar_t*)output, p, OCg_pad, s); \
    else if (p.OCg > 64) st = launch_cl16<scalar_t, 2, 4, 2, 1, 32>(icl, w8, (const scalar_t*)offset, (const scalar_t*)mask, (const scalar_t*)bias, (scalar_t*)output, p, OCg_pad, s); \
    else st = launch_cl16<scalar_t, 1, 8, 2, 1, 32>(icl, w8, (const scalar_t*)offset, (const scalar_t*)mask, (const scalar_t*)bias, (scalar_t*)output, p, OCg_pad, s); \
  } while (0)
      // (64-deep slabs — 92 KB of LDS, one workgroup per CU — were measured too: 0.185 ms against 0.154 ms at config 4)
      int st = 0;
      if (dt == TVMI_F16) TVMI_DCN16_CL(__half);
      else TVMI_DCN16_CL(__hip_bfloat16);
      if (st) return st;
#undef TVMI_DCN16_CL
      TVMI_RETURN_LAUNCH_STATUS("tvmi_deform_conv2d_forward (channels-last gather)");
    }
#define TVMI_DCN16_T(scalar_t, WM, WN, MI, NI)                                                                         \
  dcn_fwd_mfma_16<scalar_t, WM, WN, MI, NI>                                                                             \
      <<<dim3((unsigned)ceil_div(npix, 32 * NI * WN), (unsigned)ceil_div(p.OCg, 32 * MI * WM), (unsigned)p.groups),     \
         dim3(64 * WM * WN), 0, s>>>((const scalar_t*)input, (const scalar_t*)workspace, (const scalar_t*)offset,      \
                                     (const scalar_t*)mask, (const scalar_t*)bias, (scalar_t*)output, p, ICg_pad, OCg_pad)
#define TVMI_DCN16(scalar_t)                                                                                           \
  do {                                                                                                                 \
    dcn_weight_relayout16<scalar_t><<<grid1d(wtotal), dim3(256), 0, s>>>((const scalar_t*)weight, (scalar_t*)workspace, p, \
                                                                         ICg_pad, OCg_pad);                            \
    /* the matrix work is 16x cheaper than in fp32: the kernel is gather-bound, so always 8 waves per tile */          \
    if (p.OCg > 128) TVMI_DCN16_T(scalar_t, 4, 2, 2, 1);                                                               \
    else if (p.OCg > 64) TVMI_DCN16_T(scalar_t, 2, 4, 2, 1);                                                           \
    else TVMI_DCN16_T(scalar_t, 1, 8, 2, 1);                                                                           \
  } while (0)
    if (dt == TVMI_F16) TVMI_DCN16(__half);
    else TVMI_DCN16(__hip_bfloat16);
#undef TVMI_DCN16
#undef TVMI_DCN16_T
  } else if (const DwGeom pg = depthwise_pk_geom(p, dt); pg.CB > 0) {
    const dim3 grid((unsigned)(pg.ntx * pg.nty), (unsigned)(pg.csplit * p.ogroups), (unsigned)p.B);
    const size_t lds = (size_t)2 * pg.tile_sz * kDwPkCB * sizeof(float);
#define TVMI_DWPK(scalar_t)                                                                                       \
  do {                                                                                                            \
    if (pg.tile_w == 80 && pg.nstage == 3)                                                                        \
      dcn_fwd_depthwise3x3_pk<scalar_t, 80, 3, 3><<<grid, dim3(kDwThreads), lds, s>>>(                               \
          (const scalar_t*)input, (const scalar_t*)weight, (const scalar_t*)offset, (const scalar_t*)mask,        \
          (const scalar_t*)bias, (scalar_t*)output, p, pg);                                                       \
    else                                                                                                          \
      dcn_fwd_depthwise3x3_pk<scalar_t, 0, kDwPkMaxStage, 1><<<grid, dim3(kDwThreads), lds, s>>>(                    \
          (const scalar_t*)input, (const scalar_t*)weight, (const scalar_t*)offset, (const scalar_t*)mask,        \
          (const scalar_t*)bias, (scalar_t*)output, p, pg);                                                       \
  } while (0)
    if (dt == TVMI_F32) TVMI_DWPK(float);
    else if (dt == TVMI_F16) TVMI_DWPK(__half);
    else TVMI_DWPK(__hip_bfloat16);
#undef TVMI_DWPK
  } else if (const DwGeom dg = depthwise_geom(p, dt); dg.CB > 0) {
    const dim3 grid((unsigned)(dg.ntx * dg.nty), (unsigned)(dg.csplit * p.ogroups), (unsigned)p.B);
    const size_t lds = (size_t)2 * dg.CB * dg.tile_sz * sizeof(float);
#define TVMI_DW(scalar_t)                                                                                         \
  do {                                                                                                            \
    if (dg.tile_w == 75)                                                                                          \
      dcn_fwd_depthwise3x3<scalar_t, 75><<<grid, dim3(kDwThreads), lds, s>>>(                                     \
          (const scalar_t*)input, (const scalar_t*)weight, (const scalar_t*)offset, (const scalar_t*)mask,        \
          (const scalar_t*)bias, (scalar_t*)output, p, dg);                                                       \
    else                                                                                                          \
      dcn_fwd_depthwise3x3<scalar_t, 0><<<grid, dim3(kDwThreads), lds, s>>>(                                      \
          (const scalar_t*)input, (const scalar_t*)weight, (const scalar_t*)offset, (const scalar_t*)mask,        \
          (const scalar_t*)bias, (scalar_t*)output, p, dg);                                                       \
  } while (0)
    if (dt == TVMI_F32) TVMI_DW(float);
    else if (dt == TVMI_F16) TVMI_DW(__half);
    else TVMI_DW(__hip_bfloat16);
#undef TVMI_DW
  } else {
    TVMI_DISPATCH_FLOAT(dt, "deform_conv2d_forward",
                        dcn_fwd_direct<scalar_t><<<grid1d(total), dim3(256), 0, s>>>(
                            (const scalar_t*)input, (const scalar_t*)weight, (const scalar_t*)offset,
                            (const scalar_t*)mask, (const scalar_t*)bias, (scalar_t*)output, p, total));
  }
  TVMI_RETURN_LAUNCH_STATUS("tvmi_deform_conv2d_forward");
}

