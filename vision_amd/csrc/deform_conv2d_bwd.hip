// deform_conv2d_bwd.hip — the backward pass of deform_conv2d for gfx950 (MI355X), fused.
//
// Semantics: torchvision/csrc/ops/cpu/deform_conv2d_kernel.cpp
//   deformable_col2im_kernel :274-348 (grad_input), get_coordinate_weight :407-438 and
//   deformable_col2im_coord_kernel :440-551 (grad_offset, grad_mask), backward host code :1153-1226;
//   the device pattern it replaces: cuda/deform_conv2d_kernel.cu:752-1033 (compute_grad_input / compute_grad_offset_and_mask:
//   one library GEMM `columns = W^T x grad_out` per weight group into a materialised [C*kh*kw, B*oh*ow] buffer — 250 MB at
//   2x256x100x136, k = 3 — read back by a col2im and a col2im_coord kernel; backward_gradient_parameters: im2col into the
//   same buffer + a second library GEMM).
//
// Here `columns` never exists in memory and no library GEMM is called:
//   * dcn_bwd_data_mfma   workgroup = 64 output pixels.  For every (tap, 256 input channels) it contracts
//         gcol[(tap, c), n] = sum_o W[o, c, tap] * grad_out[o, n]      (K = out channels of the weight group)
//     on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, both operands staged through double-buffered LDS slabs, one barrier
//     per 16-deep slab) and consumes the 32x32 accumulator blocks WHERE THEY ARE: a lane holds 16 channels of ONE pixel, whose
//     sampling location it computed once per tap; per value it gathers the 4 corners of the input (for the coordinate /
//     mask gradients), scatters `mask * bilinear weight * gcol` into grad_input with global float atomics (what the
//     reference does, cuda/deform_conv2d_kernel.cu:319-389) and adds to its private grad_offset / grad_mask sums, which leave
//     the wave as one atomic per (tap, pixel, component).
//   * dcn_bwd_weight_mfma  workgroup = (tap, pixel range, 256 out channels x 128 in channels).  It contracts
//         grad_weight[o, c, tap] = sum_n grad_out[o, n] * col[(c, tap), n]   (K = output pixels)
//     with the offset-gather + bilinear "im2col" values of a 16-pixel slab produced straight into LDS (as the forward kernel
//     does), and adds its partial block to a [tap][o][c] fp32 buffer (coalesced atomics), which a last small kernel
//     transposes into the weight layout.
//   * 16-bit tensors are read natively and contracted on v_mfma_f32_32x32x16_{f16,bf16} in the owner form of the data kernel
//     and in the weight kernel (round 5: operands in the tensor type as in the reference's 16-bit GEMMs, fp32 accumulation;
//     1.43 -> 1.17 ms at config 4), sums in fp32 buffers, rounded once at the end — the reference accumulates grad_input in
//     16-bit atomics; fp64, depthwise and tiny channel counts take the direct kernels below (same fusion, no matrix cores).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <type_traits>

#include "dcn_common.h"

namespace tvmi {
namespace {

std::atomic<int> g_bwd_mfma{1};   // option "dcn.bwd_mfma": 0 sends every problem to the direct kernels
std::atomic<int> g_bwd_window{1}; // option "dcn.bwd_window": 0 = the data-gradient kernel scatters with global atomics only
std::atomic<int> g_bwd_owner{1};  // option "dcn.bwd_owner": 0 = never the owner form of the data-gradient kernel (dcn_bwd_data_own)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16-bit tensors contract on v_mfma_f32_32x32x16_{f16,bf16} (round 5; the reference runs the backward GEMMs in the 16-bit type
// too, cuda/deform_conv2d_kernel.cu:855,1021): a lane supplies 8 consecutive k of one row as one 16-byte LDS read from
// [row][k] rows of 48 bytes (16 k + 8 pad: 16 lanes x 48 B tile the 64 banks without conflict).  fp32 accumulation.
typedef _Float16 bw_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bw_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kBwPitch16 = 24;
template <typename T>
__device__ __forceinline__ f32x16 bw_mfma16(uint4 a, uint4 b, f32x16 c) {
  if constexpr (std::is_same<T, __half>::value)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(bw_f16x8, a), __builtin_bit_cast(bw_f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bw_bf16x8, a), __builtin_bit_cast(bw_bf16x8, b), c, 0, 0, 0);
}
template <typename T>
__device__ __forceinline__ unsigned short bw_bits16(float v) {
  if constexpr (std::is_same<T, __half>::value) return __half_as_ushort(__float2half(v));
  else {
    const __hip_bfloat16 b = __float2bfloat16(v);
    return *reinterpret_cast<const unsigned short*>(&b);
  }
}

// One sampling location as the backward needs it: the 4 corner offsets (clamped to a legal address), the bilinear weights of
// the forward (zero for corners outside the image and for locations outside (-1, H) x (-1, W): deform_conv2d_kernel.cpp:95-132
// — the weights col2im scatters with, :334-343), and what get_coordinate_weight (:407-438) uses: the corner validity on its
// own (NOT gated by the location test: y == -1 exactly has a valid lower row there) and the fractions.
template <typename A>
struct BwdTap {
  int o[4];
  A bw[4];        // bilinear weights, no mask
  A m;            // modulation mask (1 when unused)
  A dy, dx;
  bool cv[4];     // corner validity of get_coordinate_weight
};

template <typename A>
__device__ __forceinline__ void make_bwd_tap(BwdTap<A>& t, int H, int W, A y, A x, A mask) {
  const bool inside = !(y <= (A)-1 || (A)H <= y || x <= (A)-1 || (A)W <= x);
  // far-away (or NaN) coordinates: every corner is invalid; keep the int conversion defined
  const A yc = fmin(fmax(y, (A)-2), (A)H + (A)1), xc = fmin(fmax(x, (A)-2), (A)W + (A)1);
  const int yl = (int)floor(yc), xl = (int)floor(xc);
  const int yh = yl + 1, xh = xl + 1;
  const bool vyl = 0 <= yl && yl < H, vyh = 0 <= yh && yh < H, vxl = 0 <= xl && xl < W, vxh = 0 <= xh && xh < W;
  const int cyl = min(max(yl, 0), H - 1), cyh = min(max(yh, 0), H - 1), cxl = min(max(xl, 0), W - 1), cxh = min(max(xh, 0), W - 1);
  t.dy = yc - (A)yl;
  t.dx = xc - (A)xl;
  t.m = mask;
  t.o[0] = cyl * W + cxl;
  t.o[1] = cyl * W + cxh;
  t.o[2] = cyh * W + cxl;
  t.o[3] = cyh * W + cxh;
  t.cv[0] = vyl && vxl;
  t.cv[1] = vyl && vxh;
  t.cv[2] = vyh && vxl;
  t.cv[3] = vyh && vxh;
  const A hh = (A)1 - t.dy, hw = (A)1 - t.dx;
  t.bw[0] = (inside && t.cv[0]) ? hh * hw : (A)0;
  t.bw[1] = (inside && t.cv[1]) ? hh * t.dx : (A)0;
  t.bw[2] = (inside && t.cv[2]) ? t.dy * hw : (A)0;
  t.bw[3] = (inside && t.cv[3]) ? t.dy * t.dx : (A)0;
}

template <typename T, typename A>
__device__ __forceinline__ void load_bwd_tap(BwdTap<A>& t, const DcnParams& p, const T* __restrict__ offset,
                                             const T* __restrict__ mask, int b, int og, int tap, int oy, int ox) {
  const int i = tap / p.kw, j = tap - i * p.kw;
  const int64_t plane = (int64_t)p.oh * p.ow;
  const int64_t pix = (int64_t)oy * p.ow + ox;
  const T* optr = offset + ((int64_t)(b * p.ogroups + og) * 2 * p.kh * p.kw) * plane;
  const A off_h = ld(optr + (int64_t)(2 * tap) * plane + pix);
  const A off_w = ld(optr + (int64_t)(2 * tap + 1) * plane + pix);
  A mval = (A)1;
  if (p.use_mask) mval = ld(mask + ((int64_t)(b * p.ogroups + og) * p.kh * p.kw + tap) * plane + pix);
  const A y = (A)(oy * p.sh - p.ph) + (A)(i * p.dh) + off_h;
  const A x = (A)(ox * p.sw - p.pw) + (A)(j * p.dw) + off_w;
  make_bwd_tap<A>(t, p.H, p.W, y, x, mval);
}

// What one gcol value v = columns[(c, tap), n] contributes, given the 4 corner values of input channel c (read at the clamped
// offsets): the scatter into grad_input (col2im), and the three sums of col2im_coord.
template <typename A>
struct CoordSums {
  A gy, gx, gm;
};
template <typename A>
__device__ __forceinline__ void accumulate_coord(CoordSums<A>& s, const BwdTap<A>& t, A v, A x0, A x1, A x2, A x3, bool use_mask) {
  const A v_yx = t.cv[0] ? x0 : (A)0, v_yX = t.cv[1] ? x1 : (A)0, v_Yx = t.cv[2] ? x2 : (A)0, v_YX = t.cv[3] ? x3 : (A)0;
  const A wy = t.dx * (v_YX - v_yX) + ((A)1 - t.dx) * (v_Yx - v_yx);   // get_coordinate_weight, y direction
  const A wx = t.dy * (v_YX - v_Yx) + ((A)1 - t.dy) * (v_yX - v_yx);   // x direction
  s.gy += t.m * wy * v;
  s.gx += t.m * wx * v;
  if (use_mask) s.gm += v * (t.bw[0] * x0 + t.bw[1] * x1 + t.bw[2] * x2 + t.bw[3] * x3);
}

// load_tap_raw of dcn_common.h with every load unconditional: `if (use_mask) r.m = mask[...]` (and an `if (pixel exists)` around
// the call) makes the loaded value one input of a phi, the copy that resolves it is a use, and the s_waitcnt vmcnt(0) it needs
// lands wherever the copy does — in dcn_bwd_weight_mfma in front of the MFMAs of the slab, with the whole next slab of gathers
// in flight.  Here the mask is read from a legal address in any case (the offset tensor when there is no mask) and the caller
// selects 1 at the point of use; the pixel must be a legal one.
template <typename T>
__device__ __forceinline__ void load_tap_raw_u(TapRaw<T>& r, const DcnParams& p, const T* __restrict__ offset,
                                               const T* __restrict__ mask, int b, int og, int tap, int oy, int ox) {
  const int64_t plane = (int64_t)p.oh * p.ow;
  const int64_t pix = (int64_t)oy * p.ow + ox;
  const T* optr = offset + ((int64_t)(b * p.ogroups + og) * 2 * p.kh * p.kw + 2 * tap) * plane + pix;
  const T* mptr = p.use_mask ? mask + ((int64_t)(b * p.ogroups + og) * p.kh * p.kw + tap) * plane + pix : optr;
  r.off_h = optr[0];
  r.off_w = optr[plane];
  r.m = mptr[0];
}
template <typename T>
__device__ __forceinline__ float raw_mask(const DcnParams& p, const TapRaw<T>& r) {
  return p.use_mask ? (float)ld(&r.m) : 1.f;
}

// ------------------------------------------------------------------ small kernels around the two contractions
// weight [OC, ICg, kh, kw] (T) -> wtb [groups][tap][OCg_pad][ICg_pad] fp32 (ic fastest, zero padded): the A operand of
// the data-gradient contraction, rows = out channels (K), columns = in channels (M)
template <typename T>
__global__ void dcn_weight_relayout_bwd(const T* __restrict__ w, float* __restrict__ wtb, DcnParams p, int OCg_pad, int ICg_pad) {
  const int KK = p.kh * p.kw;
  const int64_t total = (int64_t)p.groups * KK * OCg_pad * ICg_pad;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int ic = (int)(idx % ICg_pad);
    const int oc = (int)((idx / ICg_pad) % OCg_pad);
    const int tap = (int)((idx / ((int64_t)ICg_pad * OCg_pad)) % KK);
    const int g = (int)(idx / ((int64_t)ICg_pad * OCg_pad * KK));
    float v = 0.f;
    if (oc < p.OCg && ic < p.ICg) v = ld(w + (((int64_t)(g * p.OCg + oc)) * p.ICg + ic) * KK + tap);
    wtb[idx] = v;
  }
}

// gw_ws [groups][tap][OCg][ICg] fp32 -> grad_weight [OC, ICg, kh, kw] (T)
template <typename T>
__global__ void dcn_bwd_weight_finish(const float* __restrict__ ws, T* __restrict__ gw, DcnParams p) {
  const int KK = p.kh * p.kw;
  const int64_t total = (int64_t)p.OC * p.ICg * KK;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(idx % KK);
    const int ic = (int)((idx / KK) % p.ICg);
    const int ocg = (int)(idx / ((int64_t)KK * p.ICg));   // g * OCg + oc
    const int g = ocg / p.OCg, oc = ocg - g * p.OCg;
    st(gw + idx, ws[(((int64_t)g * KK + tap) * p.OCg + oc) * p.ICg + ic]);
  }
}

template <typename T>
__global__ void dcn_round_from_f32(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) st(dst + i, src[i]);
}

// grad_bias[oc] = sum over images and pixels of grad_out (the reference: grad.sum({0, 2, 3}))
// One workgroup per out channel; 16-byte pieces, four per thread in flight (round 5: the element-at-a-time form took 27 us for
// the 14 MB of config 4 — a chain of dependent 2-byte loads; fixed summation order: deterministic).
template <typename T>
__global__ __launch_bounds__(256) void dcn_bwd_bias(const T* __restrict__ gout, T* __restrict__ gbias, int B, int OC, int64_t plane) {
  using A = typename Acc<T>::type;
  constexpr int EPP = 16 / (int)sizeof(T);   // elements per 16-byte piece
  struct alignas(16) Piece {
    T v[EPP];
  };
  __shared__ A part[4];
  const int oc = blockIdx.x;
  A s = (A)0;
  for (int b = 0; b < B; ++b) {
    const T* src = gout + ((int64_t)b * OC + oc) * plane;
    // leading elements up to the first 16-byte boundary, whole pieces, trailing elements
    const int64_t lead = min<int64_t>(plane, (int64_t)(((16 - (reinterpret_cast<uintptr_t>(src) & 15)) & 15) / sizeof(T)));
    const int64_t npiece = (plane - lead) / EPP;
    const Piece* pp = reinterpret_cast<const Piece*>(src + lead);
    for (int64_t i = threadIdx.x; i < npiece; i += 4 * 256) {
      Piece q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = pp[min<int64_t>(i + u * 256, npiece - 1)];   // (unconditional loads, masked below)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i + u * 256 < npiece) {
#pragma unroll
          for (int e = 0; e < EPP; ++e) s += (A)ld(&q[u].v[e]);
        }
    }
    const int64_t tail0 = lead + npiece * EPP;
    for (int64_t i = threadIdx.x; i < lead + (plane - tail0); i += 256) s += (A)ld(src + (i < lead ? i : tail0 + (i - lead)));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) st(gbias + oc, part[0] + part[1] + part[2] + part[3]);
}

// ------------------------------------------------------------------ data gradients on the matrix cores
constexpr int kBwBK = 16;

// 8 waves: 4 along M (in channels, 64 each = 2 blocks of 32) x 2 along N (pixels, 32 each).  The lane <-> (pixel, channel)
// map of an accumulator block is the forward kernel's: D[row][col], col = lane & 31 = pixel, row = (r & 3) + 8 (r >> 2) +
// 4 (lane >> 5) = channel.  GI: fp32 accumulation buffers (the outputs themselves for fp32 tensors).
template <typename T>
__global__ __launch_bounds__(512) void dcn_bwd_data_mfma(const T* __restrict__ input, const float* __restrict__ wtb,
                                                         const T* __restrict__ offset, const T* __restrict__ mask,
                                                         const T* __restrict__ gout, float* __restrict__ gi,
                                                         float* __restrict__ goff, float* __restrict__ gmask, DcnParams p,
                                                         int OCg_pad, int ICg_pad) {
  constexpr int WM = 4, WN = 2, MI = 2;
  constexpr int NT = 64 * WM * WN;       // 512
  constexpr int BM = 32 * MI * WM;       // 256 in channels
  constexpr int BN = 32 * WN;            // 64 pixels
  constexpr int AV = kBwBK * BM / 4 / NT;   // float4 pieces of the A slab per thread (2)
  constexpr int BV = kBwBK * BN / NT;       // B values per thread (2)
  static_assert(AV * NT * 4 == kBwBK * BM && BV * NT == kBwBK * BN, "slab shapes");
  __shared__ __attribute__((aligned(16))) float As[2][kBwBK][BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][kBwBK][BN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int kq = lane >> 5, l31 = lane & 31;
  const int g = blockIdx.z;
  const int KK = p.kh * p.kw;
  const int64_t oplane = (int64_t)p.oh * p.ow, iplane = (int64_t)p.H * p.W;
  const int64_t npix = (int64_t)p.B * oplane;
  const int64_t pix0 = (int64_t)blockIdx.x * BN;

  // producer side of the B slab: this thread's pixel and its rows of the slab
  const int pn = tid % BN, ksub = tid / BN;   // 8 row subsets
  const int64_t prod_pix = pix0 + pn;
  const bool prod_ok = prod_pix < npix;
  const int64_t prod_b = prod_ok ? prod_pix / oplane : 0, prod_in = prod_ok ? prod_pix - prod_b * oplane : 0;
  const T* gout_p = gout + ((int64_t)prod_b * p.OC + (int64_t)g * p.OCg) * oplane + prod_in;

  // consumer side: the pixel of this lane's accumulator column
  const int64_t my_pix = pix0 + wn * 32 + l31;
  const bool pix_ok = my_pix < npix;
  int mb = 0, moy = 0, mox = 0;
  if (pix_ok) {
    mox = (int)(my_pix % p.ow);
    moy = (int)((my_pix / p.ow) % p.oh);
    mb = (int)(my_pix / oplane);
  }
  const int64_t my_in = (int64_t)moy * p.ow + mox;

  const int nks = OCg_pad / kBwBK;                 // slabs per (tap, channel chunk)
  const int ncc = (p.ICg + BM - 1) / BM;           // channel chunks per tap
  const int total = KK * ncc * nks;

  f32x16 acc[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;

  float4 av[AV];
  float bv[BV];
  auto issue = [&](int tap, int c0, int oc0) {
    const float* wsrc = wtb + (((int64_t)g * KK + tap) * OCg_pad + oc0) * ICg_pad + c0;
#pragma unroll
    for (int e = 0; e < AV; ++e) {
      const int lin = (tid + e * NT) * 4;
      const int kk = lin / BM, m = lin - kk * BM;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c0 + m < ICg_pad) v = *reinterpret_cast<const float4*>(wsrc + (int64_t)kk * ICg_pad + m);
      av[e] = v;
    }
#pragma unroll
    for (int e = 0; e < BV; ++e) {
      const int oc = oc0 + ksub + e * (NT / BN);
      bv[e] = (prod_ok && oc < p.OCg) ? (float)ld(gout_p + (int64_t)oc * oplane) : 0.f;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int e = 0; e < AV; ++e) {
      const int lin = (tid + e * NT) * 4;
      const int kk = lin / BM, m = lin - kk * BM;
      *reinterpret_cast<float4*>(&As[buf][kk][m]) = av[e];
    }
#pragma unroll
    for (int e = 0; e < BV; ++e) Bs[buf][ksub + e * (NT / BN)][pn] = bv[e];
  };

  // grad_offset / grad_mask sums of this lane's pixel for the current (tap, offset group)
  BwdTap<float> tp;
  CoordSums<float> sums{0.f, 0.f, 0.f};
  int cur_tap = -1, cur_og = -1;
  auto flush = [&]() {
    if (cur_tap < 0) return;
    // the two lane halves hold different channels of the same pixel
    sums.gy += __shfl_xor(sums.gy, 32);
    sums.gx += __shfl_xor(sums.gx, 32);
    sums.gm += __shfl_xor(sums.gm, 32);
    if (pix_ok && kq == 0) {
      float* o = goff + ((int64_t)(mb * p.ogroups + cur_og) * 2 * KK + 2 * cur_tap) * oplane + my_in;
      unsafeAtomicAdd(o, sums.gy);
      unsafeAtomicAdd(o + oplane, sums.gx);
      if (p.use_mask) unsafeAtomicAdd(gmask + ((int64_t)(mb * p.ogroups + cur_og) * KK + cur_tap) * oplane + my_in, sums.gm);
    }
    sums.gy = sums.gx = sums.gm = 0.f;
  };
  auto epilogue = [&](int tap, int c0) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int ct = c0 + (wm * MI + mi) * 32;   // first channel of the block, inside the weight group
      if (ct < p.ICg) {                          // (wave-uniform)
        const int og = (g * p.ICg + ct) / p.cpog;   // one offset group per block: the launcher checks the alignment
        if (og != cur_og || tap != cur_tap) {       // (wave-uniform)
          flush();
          cur_og = og;
          cur_tap = tap;
          if (pix_ok) load_bwd_tap<T, float>(tp, p, offset, mask, mb, og, tap, moy, mox);
        }
        if (pix_ok) {
          const int cbase = ct + 4 * kq;
          const T* in_c = input + ((int64_t)mb * p.C + (int64_t)g * p.ICg) * iplane;
          float* gi_c = gi + ((int64_t)mb * p.C + (int64_t)g * p.ICg) * iplane;
          float xv[16][4];
          // all corner reads of the block first (they only feed the coordinate sums) ...
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = min(cbase + (r & 3) + 8 * (r >> 2), p.ICg - 1);
            const T* pl = in_c + (int64_t)c * iplane;
#pragma unroll
            for (int k = 0; k < 4; ++k) xv[r][k] = (float)ld(pl + tp.o[k]);
          }
          // ... the scatter, which needs none of them, under their latency ...
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = cbase + (r & 3) + 8 * (r >> 2);
            if (c < p.ICg) {
              float* pl = gi_c + (int64_t)c * iplane;
              const float v = acc[mi][r];
#pragma unroll
              for (int k = 0; k < 4; ++k)
                if (tp.bw[k] != 0.f) unsafeAtomicAdd(pl + tp.o[k], tp.m * tp.bw[k] * v);
            }
          }
          // ... then the coordinate / mask sums
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = cbase + (r & 3) + 8 * (r >> 2);
            if (c < p.ICg) accumulate_coord<float>(sums, tp, acc[mi][r], xv[r][0], xv[r][1], xv[r][2], xv[r][3], p.use_mask != 0);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
    }
  };

  if (total > 0) {
    int s_tap = 0, s_cc = 0, s_ks = 0;
    issue(0, 0, 0);
    int buf = 0;
    for (int it = 0; it < total; ++it) {
      commit(buf);
      __syncthreads();
      int n_tap = s_tap, n_cc = s_cc, n_ks = s_ks + 1;
      if (n_ks == nks) {
        n_ks = 0;
        if (++n_cc == ncc) {
          n_cc = 0;
          ++n_tap;
        }
      }
      if (it + 1 < total) issue(n_tap, n_cc * BM, n_ks * kBwBK);
#pragma unroll
      for (int kk = 0; kk < kBwBK; kk += 2) {
        float a[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = As[buf][kk + kq][(wm * MI + mi) * 32 + l31];
        const float b = Bs[buf][kk + kq][wn * 32 + l31];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b, acc[mi], 0, 0, 0);
      }
      if (s_ks == nks - 1) epilogue(s_tap, s_cc * BM);
      s_tap = n_tap;
      s_cc = n_cc;
      s_ks = n_ks;
      buf ^= 1;
    }
    flush();
  }
}

// ------------------------------------------------------------------ data gradients on the matrix cores, grad_input through an LDS window
// dcn_bwd_data_mfma above spends ~5 of its 5.5 ms at config 4 in the scatter: 250 M global float atomics whose addresses follow
// the (random) offsets — one 64-byte granule per lane (profiles/r04_dcn_bwd_*.json: the round-3 route with library GEMMs, the
// fused kernel and the depthwise direct kernel all land at ~6 ms).  Here the scatter goes to LDS:
//   workgroup = an 8 x 16 tile of output pixels of one image; channel chunk (64) OUTER, tap inner.  The 36 contributions
//   (9 taps x 4 corners) a chunk's accumulator blocks make per pixel are added with ds_add_f32 into a window of the input
//   plane — the tile's footprint plus kWinR pixels of offset reach — kept in LDS per channel of the chunk; after the last tap
//   the window is flushed ONCE with row-contiguous global atomics (neighbouring tiles overlap in their halos) and zeroed.
//   A corner outside the window (an offset beyond the reach) falls back to a global atomic: correct for any offset, fast
//   for |offset| <= kWinR.  grad_offset / grad_mask: as above, one atomic per (tap, chunk, pixel, component) and wave.
// Lane <-> pixel: accumulator column block wn holds tile rows 2 wn, 2 wn + 1 (16 pixels each).
constexpr int kWinTH = 8, kWinTW = 16, kWinR = 3, kWinCH = 64;

struct WinGeom {
  int wh, ww, wsz;   // window rows, columns, elements per channel
  int ntx, nty;      // tiles per image
};

template <typename T>
__global__ __launch_bounds__(512) void dcn_bwd_data_mfma_win(const T* __restrict__ input, const float* __restrict__ wtb,
                                                             const T* __restrict__ offset, const T* __restrict__ mask,
                                                             const T* __restrict__ gout, float* __restrict__ gi,
                                                             float* __restrict__ goff, float* __restrict__ gmask, DcnParams p,
                                                             int OCg_pad, int ICg_pad, WinGeom wg) {
  constexpr int WM = 2, WN = 4;
  constexpr int NT = 64 * WM * WN;       // 512
  constexpr int BM = kWinCH;             // 64 in channels = 2 blocks of 32
  constexpr int BN = kWinTH * kWinTW;    // 128 pixels = 4 blocks of 32
  constexpr int AQ = kBwBK * BM / 4;     // 256 float4 pieces of the A slab: threads 0..255 take one each
  constexpr int BV = kBwBK * BN / NT;    // 4 B values per thread
  static_assert(BM == 32 * WM && BN == 32 * WN && AQ <= NT && BV * NT == kBwBK * BN, "tile shapes");
  extern __shared__ __attribute__((aligned(16))) float dcn_bwin_lds[];
  float(*As)[kBwBK][BM] = reinterpret_cast<float(*)[kBwBK][BM]>(dcn_bwin_lds);
  float(*Bs)[kBwBK][BN] = reinterpret_cast<float(*)[kBwBK][BN]>(dcn_bwin_lds + 2 * kBwBK * BM);
  float* win = dcn_bwin_lds + 2 * kBwBK * (BM + BN);   // [kWinCH][wsz]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int kq = lane >> 5, l31 = lane & 31;
  const int g = blockIdx.z, b = blockIdx.y;
  const int KK = p.kh * p.kw;
  const int64_t oplane = (int64_t)p.oh * p.ow, iplane = (int64_t)p.H * p.W;
  const int ty0 = (blockIdx.x / wg.ntx) * kWinTH, tx0 = (blockIdx.x % wg.ntx) * kWinTW;
  const int wy0 = ty0 * p.sh - p.ph - kWinR, wx0 = tx0 * p.sw - p.pw - kWinR;   // image coordinates of window element (0, 0)

  // producer side of the B slab
  const int pn = tid % BN, ksub = tid / BN;   // 4 row subsets
  const int poy = ty0 + (pn >> 4), pox = tx0 + (pn & 15);
  const bool prod_ok = poy < p.oh && pox < p.ow;
  const T* gout_p = gout + ((int64_t)b * p.OC + (int64_t)g * p.OCg) * oplane + (prod_ok ? (int64_t)poy * p.ow + pox : 0);

  // consumer side: the pixel of this lane's accumulator column
  const int moy = ty0 + 2 * wn + (l31 >> 4), mox = tx0 + (l31 & 15);
  const bool pix_ok = moy < p.oh && mox < p.ow;
  const int64_t my_in = pix_ok ? (int64_t)moy * p.ow + mox : 0;

  const int nks = OCg_pad / kBwBK;
  const int ncc = (p.ICg + BM - 1) / BM;
  const int total = ncc * KK * nks;

  for (int e = tid; e < kWinCH * wg.wsz; e += NT) win[e] = 0.f;   // (the first barrier of the slab loop orders this)

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
  float bv[BV];
  const int a_kk = (tid * 4) / BM, a_m = (tid * 4) % BM;   // (threads >= AQ: unused)
  auto issue = [&](int tap, int c0, int oc0) {
    if (tid < AQ) {
      av = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c0 + a_m < ICg_pad)
        av = *reinterpret_cast<const float4*>(wtb + (((int64_t)g * KK + tap) * OCg_pad + oc0 + a_kk) * ICg_pad + c0 + a_m);
    }
#pragma unroll
    for (int e = 0; e < BV; ++e) {
      const int oc = oc0 + ksub + e * (NT / BN);
      bv[e] = (prod_ok && oc < p.OCg) ? (float)ld(gout_p + (int64_t)oc * oplane) : 0.f;
    }
  };
  auto commit = [&](int buf) {
    if (tid < AQ) *reinterpret_cast<float4*>(&As[buf][a_kk][a_m]) = av;
#pragma unroll
    for (int e = 0; e < BV; ++e) Bs[buf][ksub + e * (NT / BN)][pn] = bv[e];
  };

  // the raw (offset_h, offset_w, mask) of this lane's pixel for the tap whose slabs are being contracted: fetched when the
  // tap begins, used in its epilogue
  TapRaw<T> raw = tap_raw_identity<T>();
  auto epilogue = [&](int tap, int c0) {
    const int ct = c0 + wm * 32;   // first channel of this wave's block, inside the weight group
    if (ct < p.ICg && pix_ok) {
      const int og = (g * p.ICg + ct) / p.cpog;   // one offset group per block: the launcher checks the alignment
      BwdTap<float> tp;
      {
        const int i = tap / p.kw, j = tap - i * p.kw;
        const float y = (float)(moy * p.sh - p.ph) + (float)(i * p.dh) + (float)ld(&raw.off_h);
        const float x = (float)(mox * p.sw - p.pw) + (float)(j * p.dw) + (float)ld(&raw.off_w);
        make_bwd_tap<float>(tp, p.H, p.W, y, x, (float)ld(&raw.m));
      }
      // window offsets of the 4 corners.  Every lane issues every ds_add_f32 (no exec juggling around 64 LDS instructions):
      // a corner with zero weight or outside the window adds 0.0 to an element of the lane's own (row 0 of the channel: 64
      // distinct banks); a weighted corner outside the window — an offset beyond the reach — is rare and goes to the cold
      // loop of global atomics below.
      int wo[4];
      float wgt[4];
      bool inw[4], far_k[4], far = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int cy = tp.o[k] / p.W, cx = tp.o[k] - cy * p.W;
        const int wy = cy - wy0, wx = cx - wx0;
        const bool in_window = wy >= 0 && wy < wg.wh && wx >= 0 && wx < wg.ww;
        const bool weighted = tp.bw[k] != 0.f;
        inw[k] = in_window && weighted;
        far_k[k] = weighted && !in_window;
        far = far || far_k[k];
        wo[k] = inw[k] ? wy * wg.ww + wx : lane;
        wgt[k] = tp.m * tp.bw[k];
      }
      const int cbase = ct + 4 * kq;
      // 32-bit element offsets from the (image, weight group) base: the launcher sends larger planes to the kernel above
      const T* in_c = input + ((int64_t)b * p.C + (int64_t)g * p.ICg) * iplane;
      float* gi_c = gi + ((int64_t)b * p.C + (int64_t)g * p.ICg) * iplane;
      const int ipl = (int)iplane;
      CoordSums<float> sums{0.f, 0.f, 0.f};
      // 8 accumulator rows at a time: corner reads first (they only feed the coordinate sums), the scatter under their
      // latency, then the sums
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float xv[8][4];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = 8 * half + q;
          const int c = min(cbase + (r & 3) + 8 * (r >> 2), p.ICg - 1);
          const int coff = c * ipl;
#pragma unroll
          for (int k = 0; k < 4; ++k) xv[q][k] = (float)ld(in_c + (coff + tp.o[k]));
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = 8 * half + q;
          const int c = cbase + (r & 3) + 8 * (r >> 2);
          if (c < p.ICg) {
            float* wrow = win + (c - c0) * wg.wsz;
            const float v = acc[r];
#pragma unroll
            for (int k = 0; k < 4; ++k) lds_add_f32(wrow + wo[k], inw[k] ? wgt[k] * v : 0.f);
          }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = 8 * half + q;
          const int c = cbase + (r & 3) + 8 * (r >> 2);
          if (c < p.ICg) accumulate_coord<float>(sums, tp, acc[r], xv[q][0], xv[q][1], xv[q][2], xv[q][3], p.use_mask != 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (far) {
#pragma unroll 1
        for (int r = 0; r < 16; ++r) {
          const int c = cbase + (r & 3) + 8 * (r >> 2);
          if (c >= p.ICg) continue;
          float v = 0.f;   // (acc[r] with a run-time r: select, not a scratch array)
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) v = rr == r ? acc[rr] : v;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (far_k[k]) unsafeAtomicAdd(gi_c + (c * ipl + tp.o[k]), wgt[k] * v);
        }
      }
      // the two lane halves hold different channels of the same pixel (both halves are active here: pix_ok is per pixel)
      sums.gy += __shfl_xor(sums.gy, 32);
      sums.gx += __shfl_xor(sums.gx, 32);
      sums.gm += __shfl_xor(sums.gm, 32);
      if (kq == 0) {
        float* o = goff + ((int64_t)(b * p.ogroups + og) * 2 * KK + 2 * tap) * oplane + my_in;
        unsafeAtomicAdd(o, sums.gy);
        unsafeAtomicAdd(o + oplane, sums.gx);
        if (p.use_mask) unsafeAtomicAdd(gmask + ((int64_t)(b * p.ogroups + og) * KK + tap) * oplane + my_in, sums.gm);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  };
  auto flush_window = [&](int c0) {
    for (int ch = wave; ch < kWinCH; ch += NT / 64) {
      if (c0 + ch >= p.ICg) break;
      float* wrow = win + ch * wg.wsz;
      float* gpl = gi + ((int64_t)b * p.C + (int64_t)g * p.ICg + c0 + ch) * iplane;
      for (int idx = lane; idx < wg.wsz; idx += 64) {
        const float v = wrow[idx];
        wrow[idx] = 0.f;
        const int wy = idx / wg.ww, wx = idx - wy * wg.ww;
        const int iy = wy0 + wy, ix = wx0 + wx;
        if (v != 0.f && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) unsafeAtomicAdd(gpl + (int64_t)iy * p.W + ix, v);
      }
    }
  };
  auto fetch_raw = [&](int tap, int c0) {
    const int ct = c0 + wm * 32;
    if (ct < p.ICg && pix_ok) load_tap_raw<T>(raw, p, offset, mask, b, (g * p.ICg + ct) / p.cpog, tap, moy, mox);
  };

  if (total > 0) {
    int s_cc = 0, s_tap = 0, s_ks = 0;
    issue(0, 0, 0);
    fetch_raw(0, 0);
    int buf = 0;
    for (int it = 0; it < total; ++it) {
      commit(buf);
      __syncthreads();
      int n_cc = s_cc, n_tap = s_tap, n_ks = s_ks + 1;
      if (n_ks == nks) {
        n_ks = 0;
        if (++n_tap == KK) {
          n_tap = 0;
          ++n_cc;
        }
      }
      if (it + 1 < total) issue(n_tap, n_cc * BM, n_ks * kBwBK);
#pragma unroll
      for (int kk = 0; kk < kBwBK; kk += 2) {
        const float a = As[buf][kk + kq][wm * 32 + l31];
        const float bb = Bs[buf][kk + kq][wn * 32 + l31];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
      }
      if (s_ks == nks - 1) {
        epilogue(s_tap, s_cc * BM);
        if (it + 1 < total) fetch_raw(n_tap, n_cc * BM);   // (after the epilogue used the current tap's)
        if (s_tap == KK - 1) {
          __syncthreads();   // every wave's window adds of this chunk are done
          flush_window(s_cc * BM);
        }
      }
      s_cc = n_cc;
      s_tap = n_tap;
      s_ks = n_ks;
      buf ^= 1;
    }
  }
}

inline WinGeom win_geom(const DcnParams& p) {
  WinGeom w{};
  w.wh = (kWinTH - 1) * p.sh + (p.kh - 1) * p.dh + 2 * kWinR + 2;
  w.ww = (kWinTW - 1) * p.sw + (p.kw - 1) * p.dw + 2 * kWinR + 2;
  w.wsz = w.wh * w.ww;
  w.ntx = (int)ceil_div(p.ow, kWinTW);
  w.nty = (int)ceil_div(p.oh, kWinTH);
  return w;
}
inline size_t win_lds_bytes(const WinGeom& w) {
  return ((size_t)2 * kBwBK * (kWinCH + kWinTH * kWinTW) + (size_t)kWinCH * w.wsz) * sizeof(float);
}

// ------------------------------------------------------------------ data gradients, owner form (lane = channel)
// What the probe tools/probe/lds_atomic_rate.hip measured on gfx950: a wave-level ds_add_f32 costs ~193 cycles per CU whatever its
// addresses (3 cycles per lane: the LDS float atomic is serialised), a ds_read_b32 + v_add + ds_write_b32 on the same 64
// addresses ~15.  dcn_bwd_data_mfma_win above is bound by exactly that (1.4 of its 2.2 ms at config 4 are the 250 M LDS
// atomic lane-operations), and by its slab loop: one 16-deep slab of global loads in flight and 8 MFMAs per wave to cover it.
// This kernel removes both:
//   * accumulator blocks are D[pixel][channel]: lane & 31 = CHANNEL, the 16 registers x 2 lane halves = 32 pixels (two rows of
//     the 8 x 16 pixel tile; the halves hold pixels 8 columns apart).  The LDS window of the grad_input tile is [position][64
//     channels]: one scatter instruction adds to 32 consecutive channels of 2 positions that can never coincide (|offset| <=
//     kOwnR, 8 columns apart), so a plain read + add + write is exact.  Waves that hold the same channels (the 4 pixel blocks)
//     take turns, a barrier between them; program order inside a wave does the rest (the LDS is in-order per wave).
//   * channel chunk (64) outer, the 9 taps in groups of 5 + 4 whose accumulators are resident together (80 registers), K = out
//     channels in 16-deep slabs: a grad_out slab serves five taps, an iteration is 40 MFMAs per wave (2560 matrix-pipe cycles
//     against one global-load round trip), and the epilogues of a group run back to back.  (All nine taps resident — 144
//     registers — was measured first: the allocator spills the slab prefetch registers, i.e. a vmcnt(0) behind every load:
//     1.85 ms for the whole backward at config 4 instead of 1.64.)
//   * input and the grad_input sums are channels-last ([B, H*W, C]; a transposing pre-pass and a transposing finish): corner
//     reads, far-offset atomics and the window flush are 128 contiguous bytes per half wave.
//   * grad_offset / grad_mask: per pixel 3 products per lane, reduced over the 32 channel lanes by a reduce-scatter butterfly
//     (30 exchanges per epilogue instead of 3 x 16 x 5), one atomic per (tap, chunk half, pixel, component).
constexpr int kOwnTH = 8, kOwnTW = 16, kOwnR = 3, kOwnCH = 64, kOwnKB = 16, kOwnTG = 5;
constexpr int kOwnAP = kOwnTH * kOwnTW + 32;   // A slab row pitch in floats: row k + 1 starts 32 banks further
constexpr int kOwnTabDw = 16;                  // dwords per (pixel, tap) table entry

inline WinGeom own_geom(const DcnParams& p) {
  WinGeom w{};
  w.wh = (kOwnTH - 1) * p.sh + (p.kh - 1) * p.dh + 2 * kOwnR + 2;
  w.ww = (kOwnTW - 1) * p.sw + (p.kw - 1) * p.dw + 2 * kOwnR + 2;
  w.wsz = w.wh * w.ww;
  w.ntx = (int)ceil_div(p.ow, kOwnTW);
  w.nty = (int)ceil_div(p.oh, kOwnTH);
  return w;
}
inline size_t own_lds_bytes(const WinGeom& w) {
  return ((size_t)kOwnKB * kOwnAP + (size_t)kOwnKB * (kOwnTG * kOwnCH + 32) + (size_t)8 * 32 * kOwnTabDw + (size_t)(w.wsz + 1) * kOwnCH) *
         sizeof(float);
}

// in [N][R][S] -> out [N][S][R] (64 x 64 tiles through LDS)
template <typename Tin, typename Tout>
__global__ __launch_bounds__(256) void dcn_transpose_planes(const Tin* __restrict__ in, Tout* __restrict__ out, int R, int S) {
  __shared__ float tile[64][65];
  const int64_t n = blockIdx.z;
  const int r0 = blockIdx.y * 64, s0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const Tin* src = in + n * (int64_t)R * S;
  Tout* dst = out + n * (int64_t)R * S;
  for (int j = ty; j < 64; j += 4) {
    const int r = r0 + j, s = s0 + tx;
    tile[j][tx] = (r < R && s < S) ? (float)ld(src + (int64_t)r * S + s) : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    const int s = s0 + j, r = r0 + tx;
    if (s < S && r < R) st(dst + (int64_t)s * R + r, tile[tx][j]);
  }
}

template <typename T, int KK>
__global__ __launch_bounds__(512) void dcn_bwd_data_own(const T* __restrict__ xt, const float* __restrict__ wtb,
                                                        const T* __restrict__ offset, const T* __restrict__ mask,
                                                        const T* __restrict__ gout, float* git, float* __restrict__ goff,
                                                        float* __restrict__ gmask, DcnParams p, int OCg_pad, int ICg_pad,
                                                        WinGeom wg) {
  constexpr int NT = 512;
  constexpr int NPX = kOwnTH * kOwnTW;          // 128 pixels = 4 blocks of 32
  constexpr int TG = kOwnTG;                    // taps per group (accumulators resident at a time)
  constexpr int BP = TG * kOwnCH + 32;          // B slab row pitch
  constexpr int BQ = kOwnKB * TG * (kOwnCH / 4);   // float4 pieces of a B slab
  constexpr int BE = (BQ + NT - 1) / NT;
  constexpr int AE = kOwnKB * NPX / NT;         // 4 grad_out values per thread and slab
  static_assert(AE * NT == kOwnKB * NPX, "A slab shape");
  extern __shared__ __attribute__((aligned(16))) float dcn_own_lds[];
  float* As = dcn_own_lds;                                    // [KB][AP]
  float* Bs = As + kOwnKB * kOwnAP;                           // [KB][BP]
  float* tab = Bs + kOwnKB * BP;                              // [8 waves][32 pixels][12]
  float* win = tab + 8 * 32 * kOwnTabDw;                      // [wsz + 1][64]; the last row is the lanes' dummy target

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = wave >> 1, wc = wave & 1;
  const int kq = lane >> 5, l31 = lane & 31;
  const int g = blockIdx.z, b = blockIdx.y;
  const int64_t oplane = (int64_t)p.oh * p.ow;
  const int HW = p.H * p.W;
  const int ty0 = (blockIdx.x / wg.ntx) * kOwnTH, tx0 = (blockIdx.x % wg.ntx) * kOwnTW;
  const int wy0 = ty0 * p.sh - p.ph - kOwnR, wx0 = tx0 * p.sw - p.pw - kOwnR;   // image coordinates of window element (0, 0)

  // block row i (0..31) -> pixel of the wave's two tile rows: bit 2 of i (the lane half of an accumulator) moves x by 8
  auto row_x = [](int i) { return (i & 3) | (((i >> 3) & 1) << 2) | (((i >> 2) & 1) << 3); };

  // A producer: column m of the slab = (pixel block m >> 5, block row m & 31); rows k = tid / 128 + 4 e.  Addresses are a
  // uniform 64-bit base plus a 32-bit lane offset (per-lane 64-bit pointers here were what the allocator spilled first).
  const int am = tid & (NPX - 1);
  const int ak = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int a_oy = ty0 + 2 * (am >> 5) + ((am & 31) >> 4), a_ox = tx0 + row_x(am & 31);
  const bool a_ok = a_oy < p.oh && a_ox < p.ow;
  const int a_off = a_ok ? a_oy * p.ow + a_ox : 0;
  const T* gout_g = gout + ((int64_t)b * p.OC + (int64_t)g * p.OCg) * oplane;
  // B producer: float4 piece j = tid + 512 e of the slab = (row k, tap, channel quad)
  int b_src[BE], b_dst[BE], b_u[BE];
#pragma unroll
  for (int e = 0; e < BE; ++e) {
    const int j = tid + e * NT;
    const int k = j / (TG * 16), rem = j - k * (TG * 16), u = rem >> 4, c4 = rem & 15;
    b_u[e] = j < BQ ? u : TG;   // tap of the group; TG = no piece
    b_src[e] = (u * OCg_pad + k) * ICg_pad + 4 * c4;
    b_dst[e] = k * BP + u * kOwnCH + 4 * c4;
  }
  const float* wtb_g = wtb + (int64_t)g * KK * OCg_pad * ICg_pad;

  // the pixel whose table entry this lane builds (block row l31; both lane halves compute it, half 0 writes)
  const int t_oy = ty0 + 2 * wp + (l31 >> 4), t_ox = tx0 + row_x(l31);
  const bool t_ok = t_oy < p.oh && t_ox < p.ow;

  const int nks = OCg_pad / kOwnKB;
  const int ncc = p.ICg / kOwnCH;
  const T* xt_b = xt + (int64_t)b * HW * p.C;
  float* git_b = git + (int64_t)b * HW * p.C;

  for (int e = tid; e < (wg.wsz + 1) * kOwnCH; e += NT) win[e] = 0.f;   // (ordered by the barriers of the first slab)

  f32x16 acc[TG];
  T av[AE];
  float4 bv[BE];
  auto issue = [&](int c0, int oc0, int tg, int nt) {
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      const int oc = oc0 + ak + 4 * e;   // (scalar)
      if (a_ok && oc < p.OCg) av[e] = (gout_g + (int64_t)oc * oplane)[a_off];
      else st(&av[e], 0.f);
    }
    const float* wsrc = wtb_g + (((int64_t)tg * OCg_pad + oc0) * ICg_pad + c0);
#pragma unroll
    for (int e = 0; e < BE; ++e) {
      bv[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b_u[e] < nt) bv[e] = *reinterpret_cast<const float4*>(wsrc + b_src[e]);
    }
  };
  // 16-bit tensors: the slabs as 16-bit [row][k] rows of 48 bytes in the same two regions (A: 128 pixels, B: TG x 64 channels;
  // the 16 out channels of a slab = ONE 32x32x16 step per tap), the weights back in the tensor type (an exact conversion:
  // wtb is their fp32 re-layout)
  constexpr bool k16 = sizeof(T) == 2;
  unsigned short* const As16 = reinterpret_cast<unsigned short*>(As);
  unsigned short* const Bs16 = reinterpret_cast<unsigned short*>(Bs);
  auto commit = [&]() {
    if constexpr (k16) {
#pragma unroll
      for (int e = 0; e < AE; ++e) As16[am * kBwPitch16 + ak + 4 * e] = *reinterpret_cast<const unsigned short*>(&av[e]);
#pragma unroll
      for (int e = 0; e < BE; ++e)
        if (b_u[e] < TG) {
          const int k = b_dst[e] / BP, row = b_dst[e] - k * BP;   // (row = tap * 64 + channel)
          unsigned short* d = Bs16 + row * kBwPitch16 + k;
          d[0] = bw_bits16<T>(bv[e].x);
          d[kBwPitch16] = bw_bits16<T>(bv[e].y);
          d[2 * kBwPitch16] = bw_bits16<T>(bv[e].z);
          d[3 * kBwPitch16] = bw_bits16<T>(bv[e].w);
        }
      return;
    }
#pragma unroll
    for (int e = 0; e < AE; ++e) As[(ak + 4 * e) * kOwnAP + am] = (float)ld(&av[e]);
#pragma unroll
    for (int e = 0; e < BE; ++e)
      if (b_u[e] < TG) *reinterpret_cast<float4*>(&Bs[b_dst[e]]) = bv[e];
  };

  float* mytab = tab + wave * 32 * kOwnTabDw;
  TapRaw<T> raw = tap_raw_identity<T>();
  bool far_any = false;   // (wave-uniform) some pixel of the wave's block has a weighted corner beyond the reach, current tap
  auto fetch_raw = [&](int tap, int og) {
    load_tap_raw_u<T>(raw, p, offset, mask, b, og, tap, min(t_oy, p.oh - 1), min(t_ox, p.ow - 1));
  };
  // table entry of (pixel l31, tap): {window offsets x4 (float index of channel 0), image positions x4, dy, dx, m, flags,
  // scatter weights x4}
  // flags: bits 0-3 corner validity of get_coordinate_weight, 4-7 corner added in the window, 8-11 corner added with global
  // atomics (weighted but beyond the reach), 12 location inside (-1, H) x (-1, W)
  auto build_table = [&](int tap) {
    const int i = tap / p.kw, j = tap - i * p.kw;
    const float oh_ = (float)ld(&raw.off_h), ow_ = (float)ld(&raw.off_w);
    const float y = (float)(t_oy * p.sh - p.ph) + (float)(i * p.dh) + oh_;
    const float x = (float)(t_ox * p.sw - p.pw) + (float)(j * p.dw) + ow_;
    BwdTap<float> tp;
    make_bwd_tap<float>(tp, p.H, p.W, y, x, raw_mask<T>(p, raw));
    const bool reach = fabsf(oh_) <= (float)kOwnR && fabsf(ow_) <= (float)kOwnR;   // (false for NaN)
    const bool inside = !(y <= -1.f || (float)p.H <= y || x <= -1.f || (float)p.W <= x);
    int flags = inside ? (1 << 12) : 0;
    int wo[4];
    float gw_[4];   // what the scatter multiplies the accumulator with: mask x bilinear weight, 0 for a corner it does not add
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cy = tp.o[k] / p.W, cx = tp.o[k] - cy * p.W;
      const int wy = cy - wy0, wx = cx - wx0;
      const bool in_window = reach && wy >= 0 && wy < wg.wh && wx >= 0 && wx < wg.ww;
      const bool weighted = t_ok && tp.bw[k] != 0.f;
      if (tp.cv[k]) flags |= 1 << k;
      if (weighted && in_window) flags |= 16 << k;
      if (weighted && !in_window) flags |= 256 << k;
      wo[k] = ((weighted && in_window) ? wy * wg.ww + wx : wg.wsz) * kOwnCH;
      gw_[k] = (weighted && in_window) ? tp.m * tp.bw[k] : 0.f;
    }
    far_any = __any(t_ok && (flags & 0xF00));
    if (kq == 0) {
      int4* e = reinterpret_cast<int4*>(mytab + l31 * kOwnTabDw);
      e[0] = make_int4(wo[0], wo[1], wo[2], wo[3]);
      e[1] = make_int4(tp.o[0], tp.o[1], tp.o[2], tp.o[3]);
      e[2] = make_int4(__float_as_int(tp.dy), __float_as_int(tp.dx), __float_as_int(t_ok ? tp.m : 0.f), t_ok ? flags : 0);
      e[3] = make_int4(__float_as_int(gw_[0]), __float_as_int(gw_[1]), __float_as_int(gw_[2]), __float_as_int(gw_[3]));
    }
  };
  // (a wave writes its table and reads it with other lanes: the LDS is in-order per wave, the fences keep the compiler's order)
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // The epilogue's lane-derived values (table addresses, channel offsets, pixel coordinates) are rebuilt from an OPAQUE copy of
  // the lane id inside the tap loop: left visible, they are hoisted out of the chunk loop, stay live across the slab loop next to
  // the 144 accumulators, and the register allocator answers by spilling the PREFETCH registers of the slab loop — a
  // vmcnt(0) and a scratch store right behind every global load (measured: 2.16 ms instead of 1.85 with a few more of them).
  struct LaneView {
    int kq, l31;
    const float* tabq;   // table entry of block row 4 kq
    float* wl;           // window element (position 0, this lane's channel)
    int cl;              // this lane's channel inside the chunk
  };
  auto lane_view = [&]() {
    int lo = lane;
    asm volatile("" : "+v"(lo));
    LaneView v;
    v.kq = lo >> 5;
    v.l31 = lo & 31;
    v.tabq = mytab + 4 * v.kq * kOwnTabDw;
    v.cl = wc * 32 + v.l31;
    v.wl = win + v.cl;
    return v;
  };
  // table entry of accumulator register r (block row 8 (r >> 2) + 4 kq + (r & 3))
  auto entry_of = [&](const LaneView& lv, int r) {
    return reinterpret_cast<const int4*>(lv.tabq + (8 * (r >> 2) + (r & 3)) * kOwnTabDw);
  };

  // grad_offset / grad_mask of one tap: products per pixel, reduce-scatter over the 32 channel lanes.  The 16 registers are
  // taken in 4 batches of 4 (block rows 8 bi + 4 kq + q), one copy of the code: `bi` is wave-uniform but not a constant, the
  // state of the binary-counter merge (L2, L3) lives across the calls.
  struct CoordState {
    float L2[3], L3[3];
  };
  auto coord_batch = [&](const LaneView& lv, CoordState& cs, int bi, const f32x16& a, int c0) {
    const T* xc = xt_b + (g * p.ICg + c0 + lv.cl);
    auto merge = [&](const float* lo, const float* hi, float* out, int level) {
      const bool s = (lv.l31 >> (level + 1)) & 1;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float keep = s ? hi[q] : lo[q], send = s ? lo[q] : hi[q];
        out[q] = keep + __shfl_xor(send, 2 << level);
      }
    };
    float xv[4][4];
    int4 e2[4];
    const float* eb = lv.tabq + 8 * bi * kOwnTabDw;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int4* e = reinterpret_cast<const int4*>(eb + q * kOwnTabDw);
      const int4 gp = e[1];
      e2[q] = e[2];
      xv[q][0] = (float)ld(xc + (int64_t)gp.x * p.C);
      xv[q][1] = (float)ld(xc + (int64_t)gp.y * p.C);
      xv[q][2] = (float)ld(xc + (int64_t)gp.z * p.C);
      xv[q][3] = (float)ld(xc + (int64_t)gp.w * p.C);
    }
    float P[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float dy = __int_as_float(e2[q].x), dx = __int_as_float(e2[q].y), m = __int_as_float(e2[q].z);
      const int fl = e2[q].w;
      const float v = bi == 0 ? a[q] : (bi == 1 ? a[4 + q] : (bi == 2 ? a[8 + q] : a[12 + q]));
      const float x0 = xv[q][0], x1 = xv[q][1], x2 = xv[q][2], x3 = xv[q][3];
      const float v0 = (fl & 1) ? x0 : 0.f, v1 = (fl & 2) ? x1 : 0.f, v2 = (fl & 4) ? x2 : 0.f, v3 = (fl & 8) ? x3 : 0.f;
      const float wy = dx * (v3 - v1) + (1.f - dx) * (v2 - v0);   // get_coordinate_weight, y direction
      const float wx = dy * (v3 - v2) + (1.f - dy) * (v1 - v0);   // x direction
      const bool inside = (fl >> 12) & 1;
      const float hh = 1.f - dy, hw = 1.f - dx;
      const float b0 = (inside && (fl & 1)) ? hh * hw : 0.f, b1 = (inside && (fl & 2)) ? hh * dx : 0.f;
      const float b2 = (inside && (fl & 4)) ? dy * hw : 0.f, b3 = (inside && (fl & 8)) ? dy * dx : 0.f;
      P[q][0] = m * wy * v;
      P[q][1] = m * wx * v;
      P[q][2] = v * (b0 * x0 + b1 * x1 + b2 * x2 + b3 * x3);
    }
    // register bit 0 <-> lane bit 1, bit 1 <-> lane bit 2, bit 2 <-> lane bit 3, bit 3 <-> lane bit 4
    float Ma[3], Mb[3], M1[3];
    merge(P[0], P[1], Ma, 0);
    merge(P[2], P[3], Mb, 0);
    merge(Ma, Mb, M1, 1);
    if ((bi & 1) == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) cs.L2[c] = M1[c];
    } else {
      float M2[3];
      merge(cs.L2, M1, M2, 2);
      if ((bi & 2) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) cs.L3[c] = M2[c];
      } else {
        merge(cs.L3, M2, cs.L3, 3);
      }
    }
  };
  auto coord_finish = [&](const LaneView& lv, CoordState& cs, int tap, int c0) {
    const int og = (g * p.ICg + c0 + wc * 32) / p.cpog;
    // L3: the sums (over this lane pair's channels so far) of register r = lane bits 1..4
#pragma unroll
    for (int c = 0; c < 3; ++c) cs.L3[c] += __shfl_xor(cs.L3[c], 1);
    if ((lv.l31 & 1) == 0) {
      const int r = lv.l31 >> 1;
      const int i = 8 * (r >> 2) + 4 * lv.kq + (r & 3);
      const int oy = ty0 + 2 * wp + (i >> 4), ox = tx0 + row_x(i);
      if (oy < p.oh && ox < p.ow) {
        const int64_t pix = (int64_t)oy * p.ow + ox;
        float* o = goff + ((int64_t)(b * p.ogroups + og) * 2 * KK + 2 * tap) * oplane + pix;
        unsafeAtomicAdd(o, cs.L3[0]);
        unsafeAtomicAdd(o + oplane, cs.L3[1]);
        if (p.use_mask) unsafeAtomicAdd(gmask + ((int64_t)(b * p.ogroups + og) * KK + tap) * oplane + pix, cs.L3[2]);
      }
    }
  };

  // the grad_input contributions of one tap of this wave's block: read + add + write per pixel, in register order.  The
  // table entry of the NEXT register is read before the window writes of the current one (the compiler cannot know that
  // table and window never alias; written in this order it need not).
  auto scatter = [&](const LaneView& lv, const f32x16& a) {
    // (the phase is bound by instruction issue — two waves of eight are at work — so the weights come ready from the table:
    // ~16 vector instructions per register instead of ~40 when they were rebuilt from dy, dx and the flags)
    int4 wo = entry_of(lv, 0)[0], gq = entry_of(lv, 0)[3];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int4 wo_c = wo, g_c = gq;
      if (r + 1 < 16) {
        wo = entry_of(lv, r + 1)[0];
        gq = entry_of(lv, r + 1)[3];
      }
      const float v = a[r];
      const float t0 = lv.wl[wo_c.x], t1 = lv.wl[wo_c.y], t2 = lv.wl[wo_c.z], t3 = lv.wl[wo_c.w];
      lv.wl[wo_c.x] = t0 + __int_as_float(g_c.x) * v;
      lv.wl[wo_c.y] = t1 + __int_as_float(g_c.y) * v;
      lv.wl[wo_c.z] = t2 + __int_as_float(g_c.z) * v;
      lv.wl[wo_c.w] = t3 + __int_as_float(g_c.w) * v;
    }
  };
  // a weighted corner beyond the reach of the window: rare; channels-last global atomics, one pixel (register) at a time
  auto scatter_far = [&](const LaneView& lv, const f32x16& a, int c0) {
    float* gc = git_b + (g * p.ICg + c0 + lv.cl);
#pragma unroll 1
    for (int r = 0; r < 16; ++r) {
      const int4* e = entry_of(lv, r);
      const int4 e2 = e[2];
      const int fl = e2.w;
      if (!(fl & 0xF00)) continue;
      float v = 0.f;   // (a[r] with a run-time r: selects, not a scratch array)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) v = rr == r ? a[rr] : v;
      const int4 gp = e[1];
      const float dy = __int_as_float(e2.x), dx = __int_as_float(e2.y), m = __int_as_float(e2.z);
      const float hh = 1.f - dy, hw = 1.f - dx;   // (a weighted corner is valid and the location inside)
      if (fl & 0x100) unsafeAtomicAdd(gc + (int64_t)gp.x * p.C, m * (hh * hw) * v);
      if (fl & 0x200) unsafeAtomicAdd(gc + (int64_t)gp.y * p.C, m * (hh * dx) * v);
      if (fl & 0x400) unsafeAtomicAdd(gc + (int64_t)gp.z * p.C, m * (dy * hw) * v);
      if (fl & 0x800) unsafeAtomicAdd(gc + (int64_t)gp.w * p.C, m * (dy * dx) * v);
    }
  };

#pragma unroll 1
  for (int cc = 0; cc < ncc; ++cc) {
    const int c0 = cc * kOwnCH;
    const int og = (g * p.ICg + c0 + wc * 32) / p.cpog;   // one offset group per 32-channel block (the launcher checks)
#pragma unroll 1
    for (int tg = 0; tg < KK; tg += TG) {
      const int nt = min(TG, KK - tg);
      issue(c0, 0, tg, nt);   // (not held across the epilogue before: its registers there cost more than the exposed latency here)
#pragma unroll
      for (int u = 0; u < TG; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
      fetch_raw(tg, og);
#pragma unroll 1
      for (int ks = 0; ks < nks; ++ks) {
        __syncthreads();   // every wave has read the previous slab
        commit();
        __syncthreads();
        if (ks + 1 < nks) issue(c0, (ks + 1) * kOwnKB, tg, nt);
        if constexpr (k16) {
          static_assert(kOwnKB == 16, "one 32x32x16 step per slab");
          const uint4 a = *reinterpret_cast<const uint4*>(As16 + (wp * 32 + l31) * kBwPitch16 + 8 * kq);
#pragma unroll
          for (int u = 0; u < TG; ++u) {
            if (u < nt) {   // (uniform)
              const uint4 bb = *reinterpret_cast<const uint4*>(Bs16 + (u * kOwnCH + wc * 32 + l31) * kBwPitch16 + 8 * kq);
              acc[u] = bw_mfma16<T>(a, bb, acc[u]);
            }
          }
          continue;
        }
#pragma unroll
        for (int kk = 0; kk < kOwnKB; kk += 2) {
          const float a = As[(kk + kq) * kOwnAP + wp * 32 + l31];
#pragma unroll
          for (int u = 0; u < TG; ++u) {
            if (u < nt) {   // (uniform)
              const float bb = Bs[(kk + kq) * BP + u * kOwnCH + wc * 32 + l31];
              acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[u], 0, 0, 0);
            }
          }
        }
      }
      // one copy of the epilogue code for the taps of the group: the tap's block is moved into `cur`
#pragma unroll 1
      for (int u = 0; u < nt; ++u) {
        const int t = tg + u;
        f32x16 cur = acc[0];
#pragma unroll
        for (int w = 1; w < TG; ++w)
          if (u == w) cur = acc[w];   // (wave-uniform)
        build_table(t);
        if (u + 1 < nt) fetch_raw(t + 1, og);
        wave_sync();
        const LaneView lv = lane_view();
        // coordinate gradients first (all waves at once; batches inside the scatter phases of the other waves were measured:
        // a phase then lasts one global-load round trip instead of one wave's 16 read-add-write steps, 1.64 -> 1.71 ms), then
        // the four waves that hold the same channels scatter one after the other
        CoordState cs;
#pragma unroll 1
        for (int bi = 0; bi < 4; ++bi) coord_batch(lv, cs, bi, cur, c0);
        coord_finish(lv, cs, t, c0);
#pragma unroll 1
        for (int ph = 0; ph < 4; ++ph) {
          if (wp == ph) scatter(lv, cur);
          __syncthreads();
        }
        if (far_any) scatter_far(lv, cur, c0);
      }
    }
    // flush the window: one position (64 channels, 256 contiguous bytes of the channels-last sums) per wave and step
#pragma unroll 1
    for (int pos = wave; pos < wg.wsz; pos += NT / 64) {
      const float v = win[pos * kOwnCH + lane];
      win[pos * kOwnCH + lane] = 0.f;
      const int wy = pos / wg.ww, wx = pos - wy * wg.ww;
      const int iy = wy0 + wy, ix = wx0 + wx;
      if (v != 0.f && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
        unsafeAtomicAdd(git_b + ((int64_t)(iy * p.W + ix) * p.C + g * p.ICg + c0 + lane), v);
    }
  }
}

// ------------------------------------------------------------------ weight gradient on the matrix cores
// 8 waves: 4 along M (out channels, 64 each) x 2 along N (in channels, 64 each); K = output pixels in slabs of 16.
// LDS rows are [row][k] with a pitch of 17 floats: both operands arrive pixel-fastest (grad_out rows from memory, the sampled
// values per producer pixel) and are read as fragments with the row across the lanes — 17 l mod 64 is a permutation of the banks.
constexpr int kBwPitch = kBwBK + 1;

// CL: `input` is the channels-last copy [B, H*W, C] the owner form of the data kernel makes.  The im2col production is bound by
// the rate at which the texture path retires scattered loads: planar, a producer thread fetches 4 channels x 4
// corners with 16 dword loads; channels-last it is (pixel, channel quad) and fetches the same values with 4 16-byte loads — the
// four quads of a pixel that meet in a wave read one 64-byte line.  LDS row of channel c0 + 4 q + j = 32 j + q (the same
// expression as the planar rsub + 32 e, conflict-free for the same reason), so accumulator column n of block (wn, ni) is channel
// c0 + 4 (n & 31) + (wn NI + ni).
template <typename T, bool CL>
__global__ __launch_bounds__(512) void dcn_bwd_weight_mfma(const T* __restrict__ input, const T* __restrict__ offset,
                                                           const T* __restrict__ mask, const T* __restrict__ gout,
                                                           float* __restrict__ gw_ws, DcnParams p, int pix_per_wg, int ncc,
                                                           int nmc) {
  constexpr int WM = 4, WN = 2, MI = 2, NI = 2;
  constexpr int NT = 64 * WM * WN;      // 512
  constexpr int BM = 32 * MI * WM;      // 256 out channels
  constexpr int BN = 32 * NI * WN;      // 128 in channels
  constexpr int AV = BM * kBwBK / NT;   // 8 grad_out values per thread and slab
  constexpr int BV = BN * kBwBK / NT;   // 4 sampled values per thread and slab
  constexpr int RSUB = NT / kBwBK;      // 32 row subsets among the producers
  extern __shared__ __attribute__((aligned(16))) float dcn_bww_lds[];
  float(*As)[BM][kBwPitch] = reinterpret_cast<float(*)[BM][kBwPitch]>(dcn_bww_lds);
  float(*Bs)[BN][kBwPitch] = reinterpret_cast<float(*)[BN][kBwPitch]>(dcn_bww_lds + 2 * BM * kBwPitch);
  // 16-bit tensors: the same slabs as 16-bit [row][k] rows (the 16 pixels of a slab = ONE 32x32x16 step per accumulator block)
  constexpr bool k16 = sizeof(T) == 2;
  unsigned short(*As16)[BM][kBwPitch16] = reinterpret_cast<unsigned short(*)[BM][kBwPitch16]>(dcn_bww_lds);
  unsigned short(*Bs16)[BN][kBwPitch16] =
      reinterpret_cast<unsigned short(*)[BN][kBwPitch16]>(reinterpret_cast<unsigned short*>(dcn_bww_lds) + 2 * BM * kBwPitch16);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int kq = lane >> 5, l31 = lane & 31;
  const int g = blockIdx.z;
  const int KK = p.kh * p.kw;
  // blockIdx.y = (tap, channel chunk, out-channel chunk)
  const int mc = blockIdx.y % nmc, cc = (blockIdx.y / nmc) % ncc, tap = blockIdx.y / (nmc * ncc);
  const int o0 = mc * BM, c0 = cc * BN;
  const int64_t oplane = (int64_t)p.oh * p.ow, iplane = (int64_t)p.H * p.W;
  const int64_t npix = (int64_t)p.B * oplane;
  const int64_t n_begin = (int64_t)blockIdx.x * pix_per_wg, n_end = min(npix, n_begin + pix_per_wg);
  if (n_begin >= n_end) return;
  const int nslab = (int)((n_end - n_begin + kBwBK - 1) / kBwBK);

  // producer: pixel k of the slab, rows rsub + 32 e.  The producer pixel walks in steps of 16 through (image, y, x);
  // (pb, poy, pox) always describe pixel `pn` (clamped to the last pixel of the tensor when the range ends inside a slab).
  const int pk = tid % kBwBK, rsub = tid / kBwBK;
  int64_t pn = n_begin + pk;
  int pb, poy, pox;
  {
    const int64_t nn = min(pn, npix - 1);
    pox = (int)(nn % p.ow);
    poy = (int)((nn / p.ow) % p.oh);
    pb = (int)(nn / oplane);
  }
  auto advance = [&]() {
    pn += kBwBK;
    if (pn >= npix) return;   // (stays on a legal pixel; `pn` says that the slab position is past the end)
    pox += kBwBK;
    while (pox >= p.ow) {
      pox -= p.ow;
      if (++poy == p.oh) {
        poy = 0;
        ++pb;
      }
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // the offset group of this thread's first row: its raw (offset_h, offset_w, mask) are fetched one slab AHEAD (a dependent
  // load in front of the gathers of every slab otherwise); rows in other offset groups load theirs on the spot
  const int og_first = (g * p.ICg + min(c0 + (CL ? 4 * rsub : rsub), p.ICg - 1)) / p.cpog;
  TapRaw<T> raw = tap_raw_identity<T>();
  auto fetch_raw = [&]() { load_tap_raw_u<T>(raw, p, offset, mask, pb, og_first, tap, poy, pox); };   // ((pb, poy, pox) is always a legal pixel)

  // Every load of a slab is UNCONDITIONAL (rows clamped to a legal one, values kept in the tensor's type) and what must be
  // zero is zeroed at commit: a `cond ? load : 0` or a 16-bit conversion is a USE of the loaded value, and the wait it needs
  // lands in front of the MFMAs of the current slab — load latency and matrix work in series (measured: 0.69 ms).
  // TWO slabs of loads are in flight (register stages st0 / st1, the loop unrolled by two so that every stage index is static):
  // one 8-wave workgroup is resident per CU whatever the register count below 256, a slab is 32 MFMAs per wave (~1.7 us of
  // matrix pipe per SIMD) and a gather round trip under load is longer than that.
  struct Stage {
    T ga[AV];           // grad_out values of a coming slab
    bool ga_live;
    T xb[BV][4];        // its corner values
    float wgt[BV][4];   // their bilinear weights, zero for rows / pixels that do not exist
    float wmask[BV];    // the modulation mask of the row's offset group (multiplies the sum, as in the forward: sample_tap)
  };
  Stage st0, st1;
  auto issue = [&](Stage& st) {
    T(&ga)[AV] = st.ga;
    bool& ga_live = st.ga_live;
    T(&xb)[BV][4] = st.xb;
    float(&wgt)[BV][4] = st.wgt;
    float(&wmask)[BV] = st.wmask;
    const bool ok = pn < n_end;
    const int64_t pin = (int64_t)poy * p.ow + pox;
    ga_live = ok;
#pragma unroll
    for (int e = 0; e < AV; ++e) {
      const int o = min(o0 + rsub + e * RSUB, p.OCg - 1);
      ga[e] = gout[((int64_t)pb * p.OC + (int64_t)g * p.OCg + o) * oplane + pin];
    }
    Tap<float> t;
    {
      const int i = tap / p.kw, j = tap - i * p.kw;
      const float y = (float)(poy * p.sh - p.ph) + (float)(i * p.dh) + (float)ld(&raw.off_h);
      const float x = (float)(pox * p.sw - p.pw) + (float)(j * p.dw) + (float)ld(&raw.off_w);
      make_tap<float>(t, p.H, p.W, y, x, raw_mask<T>(p, raw));
    }
    if constexpr (CL) {
      static_assert(BV == 4, "a producer thread owns one channel quad");
      struct alignas(sizeof(T) * 4) Quad {
        T v[4];
      };
      const int c = c0 + 4 * rsub;
      const bool cok = ok && c < p.ICg;   // (ICg is a multiple of 4 here: whole quads)
      const int cl = min(c, p.ICg - 4);
      const T* px = input + (int64_t)pb * iplane * p.C + ((int64_t)g * p.ICg + cl);
      const Quad q1 = *reinterpret_cast<const Quad*>(px + (int64_t)t.o1 * p.C);
      const Quad q2 = *reinterpret_cast<const Quad*>(px + (int64_t)t.o2 * p.C);
      const Quad q3 = *reinterpret_cast<const Quad*>(px + (int64_t)t.o3 * p.C);
      const Quad q4 = *reinterpret_cast<const Quad*>(px + (int64_t)t.o4 * p.C);
#pragma unroll
      for (int e = 0; e < BV; ++e) {
        xb[e][0] = q1.v[e];
        xb[e][1] = q2.v[e];
        xb[e][2] = q3.v[e];
        xb[e][3] = q4.v[e];
      }
      wgt[0][0] = cok ? t.w1 : 0.f;   // (one pixel and one offset group per thread: one set of weights)
      wgt[0][1] = cok ? t.w2 : 0.f;
      wgt[0][2] = cok ? t.w3 : 0.f;
      wgt[0][3] = cok ? t.w4 : 0.f;
      wmask[0] = cok ? t.m : 0.f;
      advance();
      fetch_raw();
      return;
    }
    int og_cur = og_first;
#pragma unroll
    for (int e = 0; e < BV; ++e) {
      const int c = c0 + rsub + e * RSUB;
      const bool cok = ok && c < p.ICg;
      const int cl = min(c, p.ICg - 1);
      const int og = (g * p.ICg + cl) / p.cpog;
      if (og != og_cur) {
        load_tap<T, float>(t, p, offset, mask, pb, og, tap, poy, pox);
        og_cur = og;
      }
      const T* pl = input + ((int64_t)pb * p.C + (int64_t)g * p.ICg + cl) * iplane;
      xb[e][0] = pl[t.o1];
      xb[e][1] = pl[t.o2];
      xb[e][2] = pl[t.o3];
      xb[e][3] = pl[t.o4];
      wgt[e][0] = cok ? t.w1 : 0.f;
      wgt[e][1] = cok ? t.w2 : 0.f;
      wgt[e][2] = cok ? t.w3 : 0.f;
      wgt[e][3] = cok ? t.w4 : 0.f;
      wmask[e] = cok ? t.m : 0.f;
    }
    advance();
    fetch_raw();
  };
  auto commit = [&](int buf, const Stage& st) {
    const T(&ga)[AV] = st.ga;
    const bool ga_live = st.ga_live;
    const T(&xb)[BV][4] = st.xb;
    const float(&wgt)[BV][4] = st.wgt;
    const float(&wmask)[BV] = st.wmask;
#pragma unroll
    for (int e = 0; e < AV; ++e) {
      const bool a_ok = ga_live && o0 + rsub + e * RSUB < p.OCg;
      if constexpr (k16) As16[buf][rsub + e * RSUB][pk] = a_ok ? *reinterpret_cast<const unsigned short*>(&ga[e]) : (unsigned short)0;
      else As[buf][rsub + e * RSUB][pk] = a_ok ? (float)ld(&ga[e]) : 0.f;
    }
#pragma unroll
    for (int e = 0; e < BV; ++e) {
      // a row / pixel that does not exist contributes exactly zero (its corner values may be anything, even inf)
      const int w = CL ? 0 : e;
      const bool live = wmask[w] != 0.f || wgt[w][0] != 0.f || wgt[w][1] != 0.f || wgt[w][2] != 0.f || wgt[w][3] != 0.f;
      const float v = wmask[w] * (wgt[w][0] * (float)ld(&xb[e][0]) + wgt[w][1] * (float)ld(&xb[e][1]) +
                                  wgt[w][2] * (float)ld(&xb[e][2]) + wgt[w][3] * (float)ld(&xb[e][3]));
      // (16-bit: the sampled value rounded to the tensor type, as the reference's 16-bit `columns` holds it)
      if constexpr (k16) Bs16[buf][rsub + e * RSUB][pk] = live ? bw_bits16<T>(v) : (unsigned short)0;
      else Bs[buf][rsub + e * RSUB][pk] = live ? v : 0.f;
    }
  };

  auto contract = [&](int buf) {
    if constexpr (k16) {
      static_assert(kBwBK == 16, "one 32x32x16 step per slab");
      uint4 a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const uint4*>(&As16[buf][(wm * MI + mi) * 32 + l31][8 * kq]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const uint4*>(&Bs16[buf][(wn * NI + ni) * 32 + l31][8 * kq]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = bw_mfma16<T>(a[mi], b[ni], acc[mi][ni]);
      return;
    }
#pragma unroll
    for (int kk = 0; kk < kBwBK; kk += 2) {
      float a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = As[buf][(wm * MI + mi) * 32 + l31][kk + kq];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = Bs[buf][(wn * NI + ni) * 32 + l31][kk + kq];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
  };
  fetch_raw();
  issue(st0);
  if (nslab > 1) issue(st1);
#pragma unroll 1
  for (int s = 0; s < nslab; s += 2) {
    commit(0, st0);
    __syncthreads();
    if (s + 2 < nslab) issue(st0);
    contract(0);
    if (s + 1 < nslab) {
      commit(1, st1);
      __syncthreads();
      if (s + 3 < nslab) issue(st1);
      contract(1);
    }
  }

  // D[row][col]: col = lane & 31 = in channel (adjacent lanes, adjacent addresses), row = out channel
  float* dst = gw_ws + ((int64_t)g * KK + tap) * p.OCg * p.ICg;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int c = CL ? c0 + 4 * l31 + (wn * NI + ni) : c0 + (wn * NI + ni) * 32 + l31;
    if (c >= p.ICg) continue;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
        if (o < p.OCg) unsafeAtomicAdd(dst + (int64_t)o * p.ICg + c, acc[mi][ni][r]);
      }
  }
}

// ------------------------------------------------------------------ direct kernels (any dtype; depthwise, tiny channel counts, fp64)
// One thread per (image, offset group, tap, output pixel): it walks the channels of the offset group, forms
// gcol = sum_o W[o, c, tap] * grad_out[o, n] on the fly, scatters into grad_input and OWNS its grad_offset / grad_mask
// elements (plain stores).  GI = the type the sums are kept in (T for fp32 / fp64 tensors, float buffers for 16-bit ones).
template <typename T, typename GI>
__global__ __launch_bounds__(256) void dcn_bwd_data_direct(const T* __restrict__ input, const T* __restrict__ weight,
                                                           const T* __restrict__ offset, const T* __restrict__ mask,
                                                           const T* __restrict__ gout, GI* __restrict__ gi,
                                                           GI* __restrict__ goff, GI* __restrict__ gmask, DcnParams p,
                                                           int64_t total) {
  using A = typename Acc<T>::type;
  const int KK = p.kh * p.kw;
  const int64_t oplane = (int64_t)p.oh * p.ow, iplane = (int64_t)p.H * p.W;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int ox = (int)(idx % p.ow);
    const int oy = (int)((idx / p.ow) % p.oh);
    const int tap = (int)((idx / oplane) % KK);
    const int og = (int)((idx / (oplane * KK)) % p.ogroups);
    const int b = (int)(idx / (oplane * KK * p.ogroups));
    const int64_t pin = (int64_t)oy * p.ow + ox;
    BwdTap<A> t;
    load_bwd_tap<T, A>(t, p, offset, mask, b, og, tap, oy, ox);
    CoordSums<A> s{(A)0, (A)0, (A)0};
    for (int cc = 0; cc < p.cpog; ++cc) {
      const int c = og * p.cpog + cc;
      const int g = c / p.ICg, ic = c - g * p.ICg;
      const T* wp = weight + (((int64_t)g * p.OCg) * p.ICg + ic) * KK + tap;
      const T* gp = gout + ((int64_t)b * p.OC + (int64_t)g * p.OCg) * oplane + pin;
      A v = (A)0;
      for (int o = 0; o < p.OCg; ++o) v += ld(wp + (int64_t)o * p.ICg * KK) * ld(gp + (int64_t)o * oplane);
      const T* pl = input + ((int64_t)b * p.C + c) * iplane;
      const A x0 = ld(pl + t.o[0]), x1 = ld(pl + t.o[1]), x2 = ld(pl + t.o[2]), x3 = ld(pl + t.o[3]);
      GI* gpl = gi + ((int64_t)b * p.C + c) * iplane;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (t.bw[k] != (A)0) atomic_accum(gpl + t.o[k], t.m * t.bw[k] * v);
      accumulate_coord<A>(s, t, v, x0, x1, x2, x3, p.use_mask != 0);
    }
    GI* o = goff + ((int64_t)(b * p.ogroups + og) * 2 * KK + 2 * tap) * oplane + pin;
    o[0] = (GI)s.gy;
    o[oplane] = (GI)s.gx;
    if (p.use_mask) gmask[((int64_t)(b * p.ogroups + og) * KK + tap) * oplane + pin] = (GI)s.gm;
  }
}

// One workgroup per (in channel, tap, chunk of kDirOB out channels of the channel's weight group): every thread samples
// col[(c, tap), n] for its pixels once and multiplies it into the chunk's grad_out values; block reduction; plain stores.
constexpr int kDirOB = 8;
template <typename T>
__global__ __launch_bounds__(256) void dcn_bwd_weight_direct(const T* __restrict__ input, const T* __restrict__ offset,
                                                             const T* __restrict__ mask, const T* __restrict__ gout,
                                                             T* __restrict__ gw, DcnParams p) {
  using A = typename Acc<T>::type;
  __shared__ A part[4][kDirOB];
  const int KK = p.kh * p.kw;
  const int c = blockIdx.x, tap = blockIdx.y, oc0 = blockIdx.z * kDirOB;
  const int g = c / p.ICg, ic = c - g * p.ICg, og = c / p.cpog;
  const int64_t oplane = (int64_t)p.oh * p.ow, iplane = (int64_t)p.H * p.W;
  const int64_t npix = (int64_t)p.B * oplane;
  A acc[kDirOB];
#pragma unroll
  for (int j = 0; j < kDirOB; ++j) acc[j] = (A)0;
  for (int64_t n = threadIdx.x; n < npix; n += 256) {
    const int ox = (int)(n % p.ow);
    const int oy = (int)((n / p.ow) % p.oh);
    const int b = (int)(n / oplane);
    Tap<A> t;
    load_tap<T, A>(t, p, offset, mask, b, og, tap, oy, ox);
    const A val = sample_tap<T, A>(t, input + ((int64_t)b * p.C + c) * iplane);
    const T* gp = gout + ((int64_t)b * p.OC + (int64_t)g * p.OCg + oc0) * oplane + (int64_t)oy * p.ow + ox;
#pragma unroll
    for (int j = 0; j < kDirOB; ++j)
      if (oc0 + j < p.OCg) acc[j] += ld(gp + (int64_t)j * oplane) * val;
  }
#pragma unroll
  for (int j = 0; j < kDirOB; ++j) {
    A s = acc[j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][j] = s;
  }
  __syncthreads();
  if (threadIdx.x < kDirOB && oc0 + (int)threadIdx.x < p.OCg) {
    const int j = threadIdx.x;
    st(gw + (((int64_t)(g * p.OCg + oc0 + j)) * p.ICg + ic) * KK + tap, part[0][j] + part[1][j] + part[2][j] + part[3][j]);
  }
}

// ------------------------------------------------------------------ depthwise (groups = C = OC), channels-last
// The direct kernels above take 5.9 ms at config 4 with groups = 256: one thread per (pixel, tap) walks 256 channels and every
// one of its 250 M grad_input atomics lands in a different plane — a 64-byte granule each at the atomic unit.  Here lanes are
// CHANNELS: a wave owns (image, offset group, tap, run of pixels), reads the channels-last copies of the input and of grad_out
// (256 contiguous bytes per access), adds to channels-last grad_input sums (a wave atomic = 4 granules instead of 64), keeps the
// weight-gradient partials of its 64 x NPASS channels in registers over the run (one atomic per channel at the end), and owns
// its grad_offset / grad_mask elements: the channel sum is a wave reduction, stored plainly.
template <typename T, int NPASS>
__global__ __launch_bounds__(256) void dcn_bwd_dw_cl(const T* __restrict__ xt, const T* __restrict__ got, const T* __restrict__ weight,
                                                     const T* __restrict__ offset, const T* __restrict__ mask, float* git,
                                                     float* __restrict__ goff, float* __restrict__ gmask, float* gw, DcnParams p,
                                                     int pix_per_wave, int nchunk, int64_t nitems) {
  const int lane = threadIdx.x & 63;
  const int64_t item = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (item >= nitems) return;
  const int KK = p.kh * p.kw;
  const int chunk = (int)(item % nchunk);
  const int tap = (int)((item / nchunk) % KK);
  const int og = (int)((item / ((int64_t)nchunk * KK)) % p.ogroups);
  const int b = (int)(item / ((int64_t)nchunk * KK * p.ogroups));
  const int oplane = p.oh * p.ow, HW = p.H * p.W;
  const int n0 = chunk * pix_per_wave, n1 = min(n0 + pix_per_wave, oplane);
  const int cf = og * p.cpog + lane;   // this lane's channel of pass 0
  const int ti = tap / p.kw, tj = tap - ti * p.kw;

  float wv[NPASS], gwacc[NPASS];
#pragma unroll
  for (int q = 0; q < NPASS; ++q) {
    wv[q] = ld(weight + (int64_t)(cf + 64 * q) * KK + tap);
    gwacc[q] = 0.f;
  }
  const T* xb = xt + (int64_t)b * HW * p.C;
  float* gb = git + (int64_t)b * HW * p.C;
  TapRaw<T> raw = tap_raw_identity<T>();
  {
    const int oy = n0 / p.ow;
    load_tap_raw_u<T>(raw, p, offset, mask, b, og, tap, oy, n0 - oy * p.ow);
  }
  for (int n = n0; n < n1; ++n) {
    const int oy = n / p.ow, ox = n - oy * p.ow;
    BwdTap<float> tp;
    {
      const float y = (float)(oy * p.sh - p.ph) + (float)(ti * p.dh) + (float)ld(&raw.off_h);
      const float x = (float)(ox * p.sw - p.pw) + (float)(tj * p.dw) + (float)ld(&raw.off_w);
      make_bwd_tap<float>(tp, p.H, p.W, y, x, raw_mask<T>(p, raw));
    }
    {   // the next pixel's raw values travel under this pixel's work (the last pixel re-reads itself)
      const int nn = min(n + 1, n1 - 1), noy = nn / p.ow;
      load_tap_raw_u<T>(raw, p, offset, mask, b, og, tap, noy, nn - noy * p.ow);
    }
    const T* gop = got + ((int64_t)b * oplane + n) * p.C;
    CoordSums<float> s{0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const int c = cf + 64 * q;
      const float go = ld(gop + c);
      const float v = wv[q] * go;
      const float x0 = ld(xb + (int64_t)tp.o[0] * p.C + c), x1 = ld(xb + (int64_t)tp.o[1] * p.C + c);
      const float x2 = ld(xb + (int64_t)tp.o[2] * p.C + c), x3 = ld(xb + (int64_t)tp.o[3] * p.C + c);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (tp.bw[k] != 0.f) unsafeAtomicAdd(gb + (int64_t)tp.o[k] * p.C + c, tp.m * tp.bw[k] * v);   // (wave-uniform condition)
      accumulate_coord<float>(s, tp, v, x0, x1, x2, x3, p.use_mask != 0);
      gwacc[q] += go * (tp.m * (tp.bw[0] * x0 + tp.bw[1] * x1 + tp.bw[2] * x2 + tp.bw[3] * x3));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      s.gy += __shfl_xor(s.gy, d);
      s.gx += __shfl_xor(s.gx, d);
      s.gm += __shfl_xor(s.gm, d);
    }
    if (lane == 0) {
      float* o = goff + ((int64_t)(b * p.ogroups + og) * 2 * KK + 2 * tap) * oplane + n;
      o[0] = s.gy;
      o[oplane] = s.gx;
      if (p.use_mask) gmask[((int64_t)(b * p.ogroups + og) * KK + tap) * oplane + n] = s.gm;
    }
  }
#pragma unroll
  for (int q = 0; q < NPASS; ++q) unsafeAtomicAdd(gw + (int64_t)(cf + 64 * q) * KK + tap, gwacc[q]);
}

inline int bwd_dw_passes(const DcnParams& p, tvmi_dtype dt) {
  if (!(dt == TVMI_F32 || dt == TVMI_F16 || dt == TVMI_BF16)) return 0;
  if (p.ICg != 1 || p.OCg != 1 || p.C != p.OC || p.cpog % 64) return 0;
  const int n = p.cpog / 64;
  if (!(n == 1 || n == 2 || n == 4)) return 0;
  if ((int64_t)p.H * p.W * p.C >= (1ll << 31) || (int64_t)p.oh * p.ow * p.C >= (1ll << 31)) return 0;
  return n;
}

// the matrix-core kernels: fp32 / fp16 / bf16 tensors with a real contraction on both sides, one offset group per 32-channel
// accumulator block
inline bool bwd_mfma_shape(const DcnParams& p, tvmi_dtype dt) {
  if (!(dt == TVMI_F32 || dt == TVMI_F16 || dt == TVMI_BF16)) return false;
  if (p.OCg < 16 || p.ICg < 16) return false;
  return p.ogroups == 1 || (p.cpog % 32 == 0 && p.ICg % 32 == 0);
}

// the owner form of the data-gradient kernel: 3 x 3 taps, whole 64-channel chunks, the window in LDS
inline bool bwd_own_shape(const DcnParams& p, tvmi_dtype dt) {
  if (!bwd_mfma_shape(p, dt)) return false;
  if (p.kh != 3 || p.kw != 3 || p.ICg % kOwnCH) return false;   // 3 x 3 taps only (1 x 9 / 9 x 1 windows are another shape)
  if ((int64_t)p.H * p.W * p.C >= (1ll << 31)) return false;
  const WinGeom w = own_geom(p);
  return own_lds_bytes(w) <= (size_t)160 * 1024 && (int64_t)w.ntx * w.nty < (1ll << 31) && p.B <= 65535 && p.groups <= 65535;
}

struct BwdPlan {
  bool mfma, wide, own;     // wide: 16-bit tensors (sums live in fp32 buffers of the workspace); own: channels-last copies
  int dw;                   // depthwise channels-last route: 64-channel passes per offset group (0 = not taken)
  int OCg_pad, ICg_pad;
  size_t at_wtb, at_gw, at_gi, at_goff, at_gmask, at_xt, at_git, at_got, bytes;
};
inline BwdPlan bwd_plan(const DcnParams& p, tvmi_dtype dt) {
  BwdPlan q{};
  const bool shape = bwd_mfma_shape(p, dt);   // the workspace is sized by the shape alone (the option may change between query and call)
  q.mfma = shape && g_bwd_mfma.load(std::memory_order_relaxed);
  q.wide = dt == TVMI_F16 || dt == TVMI_BF16;
  q.OCg_pad = dcn_round_up(p.OCg, kBwBK);
  q.ICg_pad = dcn_round_up(p.ICg, 32);
  const int KK = p.kh * p.kw;
  size_t at = 0;
  if (shape) {
    q.at_wtb = at;
    at += dcn_align256((size_t)p.groups * KK * q.OCg_pad * q.ICg_pad * sizeof(float));
    q.at_gw = at;
    at += dcn_align256((size_t)p.OC * p.ICg * KK * sizeof(float));
  }
  if (q.wide) {
    q.at_gi = at;
    at += dcn_align256((size_t)p.B * p.C * p.H * p.W * sizeof(float));
    q.at_goff = at;
    at += dcn_align256((size_t)p.B * 2 * KK * p.ogroups * p.oh * p.ow * sizeof(float));
    q.at_gmask = at;
    at += dcn_align256((size_t)p.B * KK * p.ogroups * p.oh * p.ow * sizeof(float));
  }
  const bool own_shape = shape && bwd_own_shape(p, dt);
  q.own = own_shape && q.mfma && g_bwd_owner.load(std::memory_order_relaxed);
  if (own_shape) {
    const size_t n = (size_t)p.B * p.C * p.H * p.W;
    q.at_xt = at;
    at += dcn_align256(n * (dt == TVMI_F32 ? 4 : 2));
    q.at_git = at;
    at += dcn_align256(n * sizeof(float));
  }
  const int dw_passes = bwd_dw_passes(p, dt);
  q.dw = g_bwd_owner.load(std::memory_order_relaxed) ? dw_passes : 0;
  if (dw_passes) {   // (never together with the matrix-core shapes: ICg = 1)
    const size_t n = (size_t)p.B * p.C * p.H * p.W, no = (size_t)p.B * p.OC * p.oh * p.ow, esz = dt == TVMI_F32 ? 4 : 2;
    q.at_xt = at;
    at += dcn_align256(n * esz);
    q.at_got = at;
    at += dcn_align256(no * esz);
    q.at_git = at;
    at += dcn_align256(n * sizeof(float));
    q.at_gw = at;
    at += dcn_align256((size_t)p.OC * KK * sizeof(float));
  }
  q.bytes = at;
  return q;
}

constexpr size_t kMaxLdsBytes = 160 * 1024;   // per workgroup on gfx950

template <typename T>
int launch_bwd_data_mfma(const T* input, const float* wtb, const T* offset, const T* mask, const T* gout, float* gi, float* goff,
                         float* gmask, const DcnParams& p, int OCg_pad, int ICg_pad, hipStream_t s) {
  const WinGeom w = win_geom(p);
  const size_t lds = win_lds_bytes(w);
  const int64_t tiles = (int64_t)w.ntx * w.nty;
  const bool small = (int64_t)p.ICg * p.H * p.W < (1ll << 31);   // 32-bit element offsets inside an (image, weight group)
  if (g_bwd_window.load(std::memory_order_relaxed) && small && lds <= kMaxLdsBytes && tiles < (1ll << 31) && p.B <= 65535 && p.groups <= 65535) {
    auto kern = dcn_bwd_data_mfma_win<T>;
    static size_t attr_set[64] = {};  // the largest size set so far, per instantiation and device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || attr_set[dev] < lds) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return set_error((int)hipErrorInvalidValue, "deform_conv2d backward: cannot reserve the LDS window");
      if (dev >= 0 && dev < 64) attr_set[dev] = lds;
    }
    kern<<<dim3((unsigned)tiles, (unsigned)p.B, (unsigned)p.groups), dim3(512), lds, s>>>(input, wtb, offset, mask, gout, gi, goff,
                                                                                         gmask, p, OCg_pad, ICg_pad, w);
    return 0;
  }
  const int64_t npix = (int64_t)p.B * p.oh * p.ow;
  dcn_bwd_data_mfma<T><<<dim3((unsigned)ceil_div(npix, 64), 1, (unsigned)p.groups), dim3(512), 0, s>>>(
      input, wtb, offset, mask, gout, gi, goff, gmask, p, OCg_pad, ICg_pad);
  return 0;
}

// owner form: transposing pre-pass of the input, the kernel, transposing finish of the grad_input sums (git is zeroed by the caller)
template <typename T>
int launch_bwd_data_own(const T* input, T* xt, const float* wtb, const T* offset, const T* mask, const T* gout, float* git,
                        T* grad_input, float* goff, float* gmask, const DcnParams& p, int OCg_pad, int ICg_pad, hipStream_t s) {
  const WinGeom w = own_geom(p);
  const size_t lds = own_lds_bytes(w);
  auto kern = dcn_bwd_data_own<T, 9>;
  static size_t attr_set[64] = {};  // the largest size set so far, per instantiation and device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || attr_set[dev] < lds) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return set_error((int)hipErrorInvalidValue, "deform_conv2d backward: cannot reserve the LDS window");
    if (dev >= 0 && dev < 64) attr_set[dev] = lds;
  }
  const int HW = p.H * p.W;
  dcn_transpose_planes<T, T><<<dim3((unsigned)ceil_div(HW, 64), (unsigned)ceil_div(p.C, 64), (unsigned)p.B), dim3(256), 0, s>>>(input, xt, p.C, HW);
  kern<<<dim3((unsigned)(w.ntx * w.nty), (unsigned)p.B, (unsigned)p.groups), dim3(512), lds, s>>>(xt, wtb, offset, mask, gout, git, goff,
                                                                                                gmask, p, OCg_pad, ICg_pad, w);
  dcn_transpose_planes<float, T><<<dim3((unsigned)ceil_div(p.C, 64), (unsigned)ceil_div(HW, 64), (unsigned)p.B), dim3(256), 0, s>>>(git, grad_input, HW, p.C);
  return 0;
}

template <typename T, bool CL>
int launch_bwd_weight_mfma(const T* input, const T* offset, const T* mask, const T* gout, float* gw_ws, const DcnParams& p,
                           hipStream_t s) {
  constexpr int BM = 256, BN = 128;
  constexpr size_t lds = (size_t)2 * (BM + BN) * kBwPitch * sizeof(float);
  auto kern = dcn_bwd_weight_mfma<T, CL>;
  static bool attr_set[64] = {};  // per instantiation and device; racing threads set the same value
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return set_error((int)hipErrorInvalidValue, "deform_conv2d backward: cannot reserve the LDS slabs");
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const int KK = p.kh * p.kw;
  const int64_t npix = (int64_t)p.B * p.oh * p.ow;
  const int ncc = (int)ceil_div(p.ICg, BN), nmc = (int)ceil_div(p.OCg, BM);
  const int64_t per_pixel_range = (int64_t)KK * ncc * nmc * p.groups;
  // ONE workgroup per CU in total (144 - 175 VGPRs: one 8-wave workgroup is resident per CU, so the grid runs in rounds and
  // a grid a few workgroups above a multiple of the CU count pays a whole extra round: 774 workgroups = 4 rounds measured
  // 0.70 ms at config 4), pixel ranges of whole slabs, at least 8 slabs each
  static int cus[64] = {};
  if (dev >= 0 && dev < 64 && cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  const int64_t ncu = (dev >= 0 && dev < 64) ? cus[dev] : 256;
  // (the channels-last form squeezed to 128 VGPRs for two resident workgroups spills addresses inside the slab loop: 1.46 -> 1.59 ms)
  int64_t nchunks = std::max<int64_t>(1, std::min<int64_t>(ncu / per_pixel_range, ceil_div(npix, 8 * kBwBK)));
  const int64_t pix_per_wg = ceil_div(ceil_div(npix, nchunks), kBwBK) * kBwBK;
  nchunks = ceil_div(npix, pix_per_wg);
  if ((int64_t)KK * ncc * nmc > 65535 || p.groups > 65535)
    return set_error((int)hipErrorInvalidValue, "deform_conv2d backward: too many weight tiles");
  kern<<<dim3((unsigned)nchunks, (unsigned)(KK * ncc * nmc), (unsigned)p.groups), dim3(512), lds, s>>>(
      input, offset, mask, gout, gw_ws, p, (int)pix_per_wg, ncc, nmc);
  return 0;
}

}  // namespace

int set_dcn_bwd_option(const char* name, int64_t value) {
  if (std::strcmp(name, "dcn.bwd_mfma") == 0) {
    g_bwd_mfma.store(value != 0, std::memory_order_relaxed);
    return 0;
  }
  if (std::strcmp(name, "dcn.bwd_window") == 0) {
    g_bwd_window.store(value != 0, std::memory_order_relaxed);
    return 0;
  }
  if (std::strcmp(name, "dcn.bwd_owner") == 0) {
    g_bwd_owner.store(value != 0, std::memory_order_relaxed);
    return 0;
  }
  return -1;
}
int get_dcn_bwd_option(const char* name, int64_t* value) {
  if (std::strcmp(name, "dcn.bwd_mfma") == 0) {
    *value = g_bwd_mfma.load(std::memory_order_relaxed) ? 1 : 0;
    return 0;
  }
  if (std::strcmp(name, "dcn.bwd_window") == 0) {
    *value = g_bwd_window.load(std::memory_order_relaxed) ? 1 : 0;
    return 0;
  }
  if (std::strcmp(name, "dcn.bwd_owner") == 0) {
    *value = g_bwd_owner.load(std::memory_order_relaxed) ? 1 : 0;
    return 0;
  }
  return -1;
}
}  // namespace tvmi

using namespace tvmi;

extern "C" size_t tvmi_deform_conv2d_backward_workspace_bytes(tvmi_dtype dt, int64_t B, int64_t C, int64_t H, int64_t W,
                                                              int64_t OC, int64_t kh, int64_t kw, int64_t stride_h,
                                                              int64_t stride_w, int64_t pad_h, int64_t pad_w, int64_t dil_h,
                                                              int64_t dil_w, int64_t groups, int64_t offset_groups) {
  if (B <= 0 || C <= 0 || OC <= 0 || groups <= 0 || offset_groups <= 0 || kh <= 0 || kw <= 0) return 0;
  if (C % groups || OC % groups || C % offset_groups) return 0;
  // the SAME parameter block the call builds (stride / dilation decide whether the owner form of the data-gradient kernel
  // fits its window in LDS, i.e. whether its channels-last buffers are needed at all — ADVICE r04)
  DcnParams p;
  if (fill_params_common(p, B, C, H, W, OC, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, groups, offset_groups, 0)) return 0;
  if (p.oh <= 0 || p.ow <= 0) return 0;
  BwdPlan q = bwd_plan(p, dt);
  return q.bytes;
}

extern "C" int tvmi_deform_conv2d_backward(const void* grad_out, const void* input, const void* weight, const void* offset,
                                           const void* mask, void* grad_input, void* grad_weight, void* grad_offset,
                                           void* grad_mask, void* grad_bias, tvmi_dtype dt, int64_t B, int64_t C, int64_t H,
                                           int64_t W, int64_t OC, int64_t kh, int64_t kw, int64_t stride_h, int64_t stride_w,
                                           int64_t pad_h, int64_t pad_w, int64_t dil_h, int64_t dil_w, int64_t groups,
                                           int64_t offset_groups, int use_mask, void* workspace, size_t workspace_bytes,
                                           void* stream) {
  DcnParams p;
  if (int e = fill_params_common(p, B, C, H, W, OC, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, groups,
                                 offset_groups, use_mask))
    return e;
  TVMI_CHECK_ARG(dt == TVMI_F32 || dt == TVMI_F64 || dt == TVMI_F16 || dt == TVMI_BF16, "deform_conv2d backward: unsupported dtype");
  if (B == 0 || C == 0 || OC == 0) return 0;
  TVMI_CHECK_ARG(grad_out && input && weight && offset && grad_input && grad_weight && grad_offset && (!use_mask || (mask && grad_mask)),
                 "deform_conv2d backward: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const BwdPlan q = bwd_plan(p, dt);
  TVMI_CHECK_ARG(q.bytes == 0 || (workspace && workspace_bytes >= q.bytes), "deform_conv2d backward: workspace too small");
  const size_t esz = dt == TVMI_F64 ? 8 : (dt == TVMI_F32 ? 4 : 2);
  const int KK = p.kh * p.kw;
  const int64_t oplane = (int64_t)p.oh * p.ow;
  const int64_t n_gi = B * C * H * W, n_goff = B * 2 * KK * p.ogroups * oplane, n_gmask = B * KK * p.ogroups * oplane;
  char* ws = static_cast<char*>(workspace);
#define TVMI_HIP_OK(expr)                                                                         \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess) return set_error((int)e__, "tvmi_deform_conv2d_backward: memset");     \
  } while (0)

  // where the sums are formed: the outputs themselves, or fp32 buffers for 16-bit tensors
  void* gi_acc = q.wide ? (void*)(ws + q.at_gi) : grad_input;
  void* goff_acc = q.wide ? (void*)(ws + q.at_goff) : grad_offset;
  void* gmask_acc = q.wide ? (void*)(ws + q.at_gmask) : grad_mask;
  const size_t asz = q.wide ? 4 : esz;
  if (q.own || q.dw) TVMI_HIP_OK(hipMemsetAsync(ws + q.at_git, 0, (size_t)n_gi * sizeof(float), s));   // channels-last sums, transposed at the end
  else TVMI_HIP_OK(hipMemsetAsync(gi_acc, 0, (size_t)n_gi * asz, s));   // the scatter accumulates
  if (q.dw) TVMI_HIP_OK(hipMemsetAsync(q.wide ? (void*)(ws + q.at_gw) : grad_weight, 0, (size_t)p.OC * KK * sizeof(float), s));
  if (q.mfma) {                                                   // ... and so do the per-wave partial sums of the matrix-core route
    TVMI_HIP_OK(hipMemsetAsync(goff_acc, 0, (size_t)n_goff * asz, s));
    if (use_mask) TVMI_HIP_OK(hipMemsetAsync(gmask_acc, 0, (size_t)n_gmask * asz, s));
    TVMI_HIP_OK(hipMemsetAsync(ws + q.at_gw, 0, (size_t)p.OC * p.ICg * KK * sizeof(float), s));
  }
#undef TVMI_HIP_OK

  if (q.dw) {
    const int HW = p.H * p.W, OHW = p.oh * p.ow;
    const int ppw = 32;   // pixels per wave: B * offset groups * taps * ceil(oh ow / 32) waves
    const int nchunk = (int)ceil_div(OHW, ppw);
    const int64_t nitems = (int64_t)p.B * p.ogroups * KK * nchunk;
    TVMI_CHECK_ARG(ceil_div(nitems, 4) < (1ll << 31), "deform_conv2d backward: too many work items");
    float* gw_acc = q.wide ? reinterpret_cast<float*>(ws + q.at_gw) : static_cast<float*>(grad_weight);
#define TVMI_BWD_DW(scalar_t, NP)                                                                                       \
  dcn_bwd_dw_cl<scalar_t, NP><<<dim3((unsigned)ceil_div(nitems, 4)), dim3(256), 0, s>>>(                                 \
      (const scalar_t*)(ws + q.at_xt), (const scalar_t*)(ws + q.at_got), (const scalar_t*)weight, (const scalar_t*)offset, \
      (const scalar_t*)mask, (float*)(ws + q.at_git), (float*)goff_acc, (float*)gmask_acc, gw_acc, p, ppw, nchunk, nitems)
#define TVMI_BWD_DW_T(scalar_t)                                                                                         \
  do {                                                                                                                 \
    dcn_transpose_planes<scalar_t, scalar_t><<<dim3((unsigned)ceil_div(HW, 64), (unsigned)ceil_div(p.C, 64), (unsigned)p.B), dim3(256), 0, s>>>( \
        (const scalar_t*)input, (scalar_t*)(ws + q.at_xt), p.C, HW);                                                    \
    dcn_transpose_planes<scalar_t, scalar_t><<<dim3((unsigned)ceil_div(OHW, 64), (unsigned)ceil_div(p.OC, 64), (unsigned)p.B), dim3(256), 0, s>>>( \
        (const scalar_t*)grad_out, (scalar_t*)(ws + q.at_got), p.OC, OHW);                                              \
    if (q.dw == 1) TVMI_BWD_DW(scalar_t, 1);                                                                            \
    else if (q.dw == 2) TVMI_BWD_DW(scalar_t, 2);                                                                       \
    else TVMI_BWD_DW(scalar_t, 4);                                                                                      \
    dcn_transpose_planes<float, scalar_t><<<dim3((unsigned)ceil_div(p.C, 64), (unsigned)ceil_div(HW, 64), (unsigned)p.B), dim3(256), 0, s>>>( \
        (const float*)(ws + q.at_git), (scalar_t*)grad_input, HW, p.C);                                                 \
    if (q.wide)                                                                                                        \
      dcn_round_from_f32<scalar_t><<<dcn_grid1d((int64_t)p.OC * KK), dim3(256), 0, s>>>(gw_acc, (scalar_t*)grad_weight, (int64_t)p.OC * KK); \
  } while (0)
    if (dt == TVMI_F32) TVMI_BWD_DW_T(float);
    else if (dt == TVMI_F16) TVMI_BWD_DW_T(__half);
    else TVMI_BWD_DW_T(__hip_bfloat16);
#undef TVMI_BWD_DW_T
#undef TVMI_BWD_DW
  } else if (q.mfma) {
    float* wtb = reinterpret_cast<float*>(ws + q.at_wtb);
    float* gw_ws = reinterpret_cast<float*>(ws + q.at_gw);
    const int64_t wtotal = (int64_t)p.groups * KK * q.OCg_pad * q.ICg_pad;
    int st_ = 0;
#define TVMI_BWD_MFMA(scalar_t)                                                                                        \
  do {                                                                                                                 \
    dcn_weight_relayout_bwd<scalar_t><<<dcn_grid1d(wtotal), dim3(256), 0, s>>>((const scalar_t*)weight, wtb, p, q.OCg_pad, q.ICg_pad); \
    if (q.own)                                                                                                         \
      st_ = launch_bwd_data_own<scalar_t>((const scalar_t*)input, (scalar_t*)(ws + q.at_xt), wtb, (const scalar_t*)offset, \
                                          (const scalar_t*)mask, (const scalar_t*)grad_out, (float*)(ws + q.at_git),      \
                                          (scalar_t*)grad_input, (float*)goff_acc, (float*)gmask_acc, p, q.OCg_pad,       \
                                          q.ICg_pad, s);                                                                \
    else                                                                                                               \
      st_ = launch_bwd_data_mfma<scalar_t>((const scalar_t*)input, wtb, (const scalar_t*)offset, (const scalar_t*)mask, \
                                           (const scalar_t*)grad_out, (float*)gi_acc, (float*)goff_acc, (float*)gmask_acc, \
                                           p, q.OCg_pad, q.ICg_pad, s);                                                \
    if (st_ == 0 && q.own)   /* the channels-last copy of the input exists */                                           \
      st_ = launch_bwd_weight_mfma<scalar_t, true>((const scalar_t*)(ws + q.at_xt), (const scalar_t*)offset, (const scalar_t*)mask, \
                                                   (const scalar_t*)grad_out, gw_ws, p, s);                            \
    else if (st_ == 0)                                                                                                 \
      st_ = launch_bwd_weight_mfma<scalar_t, false>((const scalar_t*)input, (const scalar_t*)offset, (const scalar_t*)mask, \
                                                    (const scalar_t*)grad_out, gw_ws, p, s);                           \
    if (st_ == 0)                                                                                                      \
      dcn_bwd_weight_finish<scalar_t><<<dcn_grid1d((int64_t)p.OC * p.ICg * KK), dim3(256), 0, s>>>(gw_ws, (scalar_t*)grad_weight, p); \
  } while (0)
    if (dt == TVMI_F32) TVMI_BWD_MFMA(float);
    else if (dt == TVMI_F16) TVMI_BWD_MFMA(__half);
    else TVMI_BWD_MFMA(__hip_bfloat16);
#undef TVMI_BWD_MFMA
    if (st_) return st_;
  } else {
    const int64_t total = B * p.ogroups * KK * oplane;
    const dim3 wgrid((unsigned)C, (unsigned)KK, (unsigned)ceil_div(p.OCg, kDirOB));
    TVMI_CHECK_ARG(KK <= 65535 && ceil_div(p.OCg, kDirOB) <= 65535, "deform_conv2d backward: kernel / group size too large");
#define TVMI_BWD_DIRECT(scalar_t, acc_t)                                                                               \
  do {                                                                                                                 \
    dcn_bwd_data_direct<scalar_t, acc_t><<<dcn_grid1d(total), dim3(256), 0, s>>>(                                       \
        (const scalar_t*)input, (const scalar_t*)weight, (const scalar_t*)offset, (const scalar_t*)mask,                \
        (const scalar_t*)grad_out, (acc_t*)gi_acc, (acc_t*)goff_acc, (acc_t*)gmask_acc, p, total);                      \
    dcn_bwd_weight_direct<scalar_t><<<wgrid, dim3(256), 0, s>>>((const scalar_t*)input, (const scalar_t*)offset,        \
                                                                (const scalar_t*)mask, (const scalar_t*)grad_out,       \
                                                                (scalar_t*)grad_weight, p);                            \
  } while (0)
    if (dt == TVMI_F32) TVMI_BWD_DIRECT(float, float);
    else if (dt == TVMI_F64) TVMI_BWD_DIRECT(double, double);
    else if (dt == TVMI_F16) TVMI_BWD_DIRECT(__half, float);
    else TVMI_BWD_DIRECT(__hip_bfloat16, float);
#undef TVMI_BWD_DIRECT
  }
  if (q.wide) {
#define TVMI_BWD_ROUND(scalar_t)                                                                                        \
  do {                                                                                                                 \
    if (!q.own && !q.dw)                                                                                               \
      dcn_round_from_f32<scalar_t><<<dcn_grid1d(n_gi), dim3(256), 0, s>>>((const float*)gi_acc, (scalar_t*)grad_input, n_gi); \
    dcn_round_from_f32<scalar_t><<<dcn_grid1d(n_goff), dim3(256), 0, s>>>((const float*)goff_acc, (scalar_t*)grad_offset, n_goff); \
    if (use_mask)                                                                                                      \
      dcn_round_from_f32<scalar_t><<<dcn_grid1d(n_gmask), dim3(256), 0, s>>>((const float*)gmask_acc, (scalar_t*)grad_mask, n_gmask); \
  } while (0)
    if (dt == TVMI_F16) TVMI_BWD_ROUND(__half);
    else TVMI_BWD_ROUND(__hip_bfloat16);
#undef TVMI_BWD_ROUND
  }
  if (grad_bias) {
    TVMI_DISPATCH_FLOAT(dt, "deform_conv2d_backward",
                        dcn_bwd_bias<scalar_t><<<dim3((unsigned)OC), dim3(256), 0, s>>>((const scalar_t*)grad_out, (scalar_t*)grad_bias,
                                                                                        (int)B, (int)OC, oplane));
  }
  TVMI_RETURN_LAUNCH_STATUS("tvmi_deform_conv2d_backward");
}
