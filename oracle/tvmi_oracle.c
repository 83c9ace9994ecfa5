/* tvmi_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement, in plain C, of the reference algorithms on the hot path.  It is the
 * checker that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg compare the
 * HIP kernels against; nothing under vision_amd/ may import, link or call it.
 *
 * Parity of this restatement is PINNED: tests/test_oracle.py checks every function here
 * against (a) the reference's own CPU kernels compiled from /root/reference
 * (oracle/_ref, oracle/build_ref.py) on seeded inputs and (b) the committed golden vectors
 * under tests/golden/ that were generated from those kernels (oracle/gen_golden.py).
 *
 * Resize is pinned differently: its arithmetic is PyTorch core's (third-party to the
 * reference, torch 2.10.0), restated below from ATen/native/UpSample.h and
 * ATen/native/cuda/UpSample.cuh and checked against torch.nn.functional.interpolate on CPU.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define FN(x) x##_f32
#include "oracle_impl.inc"
#undef REAL
#undef FN

#define REAL double
#define FN(x) x##_f64
#include "oracle_impl.inc"
#undef REAL
#undef FN

/* ------------------------------------------------------------------ resize (fp32) ---- */
/* ATen/native/UpSample.h:259-287 */
static float rs_scale(int64_t in, int64_t out, int align, double scale_arg) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return scale_arg > 0. ? (float)(1.0 / scale_arg) : (float)in / (float)out;
}
/* UpSample.h:289-318 */
static float rs_src(float scale, int dst, int align, int cubic) {
  if (align) return scale * dst;
  const float s = scale * (dst + 0.5f) - 0.5f;
  return (!cubic && s < 0.f) ? 0.f : s;
}
static float rs_c1(float x, float A) { return ((A + 2) * x - (A + 3)) * x * x + 1; }
static float rs_c2(float x, float A) { return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A; }
static int rs_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* mode: 0 nearest, 1 nearest-exact, 2 bilinear, 3 bicubic (no antialias) */
void oracle_interpolate2d(const float* in, float* out, int NC, int IH, int IW, int OH, int OW, int mode, int align,
                          double scale_h, double scale_w) {
  const float sh = rs_scale(IH, OH, mode >= 2 ? align : 0, scale_h), sw = rs_scale(IW, OW, mode >= 2 ? align : 0, scale_w);
  for (int nc = 0; nc < NC; ++nc) {
    const float* p = in + (size_t)nc * IH * IW;
    float* o = out + (size_t)nc * OH * OW;
    for (int oy = 0; oy < OH; ++oy)
      for (int ox = 0; ox < OW; ++ox) {
        float v;
        if (mode <= 1) { /* UpSample.h:320-343 */
          const int iy = mode ? (int)floorf((oy + 0.5f) * sh) : (int)floorf(oy * sh);
          const int ix = mode ? (int)floorf((ox + 0.5f) * sw) : (int)floorf(ox * sw);
          v = p[(size_t)(iy < IH - 1 ? iy : IH - 1) * IW + (ix < IW - 1 ? ix : IW - 1)];
        } else if (mode == 2) { /* UpSample.h:442-476 */
          int y0, y1, x0, x1;
          float ly0, ly1, lx0, lx1;
          if (IH == OH) {
            y0 = y1 = oy;
            ly0 = 1.f;
            ly1 = 0.f;
          } else {
            const float r = rs_src(sh, oy, align, 0);
            y0 = (int)floorf(r) < IH - 1 ? (int)floorf(r) : IH - 1;
            ly1 = fminf(fmaxf(r - y0, 0.f), 1.f);
            y1 = y0 + (y0 < IH - 1 ? 1 : 0);
            ly0 = 1.f - ly1;
          }
          if (IW == OW) {
            x0 = x1 = ox;
            lx0 = 1.f;
            lx1 = 0.f;
          } else {
            const float r = rs_src(sw, ox, align, 0);
            x0 = (int)floorf(r) < IW - 1 ? (int)floorf(r) : IW - 1;
            lx1 = fminf(fmaxf(r - x0, 0.f), 1.f);
            x1 = x0 + (x0 < IW - 1 ? 1 : 0);
            lx0 = 1.f - lx1;
          }
          v = ly0 * (lx0 * p[(size_t)y0 * IW + x0] + lx1 * p[(size_t)y0 * IW + x1]) +
              ly1 * (lx0 * p[(size_t)y1 * IW + x0] + lx1 * p[(size_t)y1 * IW + x1]);
        } else { /* bicubic: UpSample.h:400-435, A = -0.75 */
          if (IH == OH && IW == OW) {
            v = p[(size_t)oy * IW + ox];
          } else {
            const float A = -0.75f;
            const float ry = rs_src(sh, oy, align, 1), rx = rs_src(sw, ox, align, 1);
            const int iy = (int)floorf(ry) < IH - 1 ? (int)floorf(ry) : IH - 1;
            const int ix = (int)floorf(rx) < IW - 1 ? (int)floorf(rx) : IW - 1;
            const float ty = fminf(fmaxf(ry - iy, 0.f), 1.f), tx = fminf(fmaxf(rx - ix, 0.f), 1.f);
            const float cy[4] = {rs_c2(ty + 1.f, A), rs_c1(ty, A), rs_c1(1.f - ty, A), rs_c2(1.f - ty + 1.f, A)};
            const float cx[4] = {rs_c2(tx + 1.f, A), rs_c1(tx, A), rs_c1(1.f - tx, A), rs_c2(1.f - tx + 1.f, A)};
            v = 0.f;
            for (int k = 0; k < 4; ++k) {
              const float* row = p + (size_t)rs_clampi(iy - 1 + k, 0, IH - 1) * IW;
              float r = 0.f;
              for (int l = 0; l < 4; ++l) r += row[rs_clampi(ix - 1 + l, 0, IW - 1)] * cx[l];
              v += r * cy[k];
            }
          }
        }
        o[(size_t)oy * OW + ox] = v;
      }
  }
}

/* Pillow-style anti-aliased resampling: ATen/native/cuda/UpSample.cuh:263-358.
 * mode: 0 bilinear (triangle, support 1), 1 bicubic (a = -0.5, support 2). */
static float aa_filter(float x, int mode) {
  if (x < 0.f) x = -x;
  if (mode == 0) return x < 1.f ? 1.f - x : 0.f;
  const float a = -0.5f;
  if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
  if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
  return 0.f;
}

static void aa_axis(int out, int in, float scale, int mode, int i, int* xmin, int* xsize, float* w, int maxw) {
  const int interp = mode == 0 ? 2 : 4;
  const float support = scale >= 1.f ? (interp * 0.5f) * scale : interp * 0.5f;
  const float center = scale * (i + 0.5f);
  int lo = (int)(center - support + 0.5f);
  if (lo < 0) lo = 0;
  int hi = (int)(center + support + 0.5f);
  if (hi > in) hi = in;
  int n = hi - lo;
  if (n < 0) n = 0;
  if (n > maxw) n = maxw;
  const float inv = scale >= 1.f ? 1.f / scale : 1.f;
  float total = 0.f;
  for (int j = 0; j < n; ++j) {
    w[j] = aa_filter((j + lo - center + 0.5f) * inv, mode);
    total += w[j];
  }
  if (total != 0.f)
    for (int j = 0; j < n; ++j) w[j] /= total;
  *xmin = lo;
  *xsize = n;
  (void)out;
}

void oracle_interpolate2d_aa(const float* in, float* out, int NC, int IH, int IW, int OH, int OW, int mode, int align,
                             double scale_h, double scale_w) {
  const float sh = rs_scale(IH, OH, align, scale_h), sw = rs_scale(IW, OW, align, scale_w);
  enum { MAXW = 4096 };
  float* wy = (float*)malloc(sizeof(float) * MAXW);
  float* wx = (float*)malloc(sizeof(float) * MAXW);
  for (int oy = 0; oy < OH; ++oy) {
    int ymin, ysize;
    aa_axis(OH, IH, sh, mode, oy, &ymin, &ysize, wy, MAXW);
    for (int ox = 0; ox < OW; ++ox) {
      int xmin, xsize;
      aa_axis(OW, IW, sw, mode, ox, &xmin, &xsize, wx, MAXW);
      for (int nc = 0; nc < NC; ++nc) {
        const float* p = in + (size_t)nc * IH * IW;
        float acc = 0.f;
        for (int j = 0; j < ysize; ++j) {
          float r = 0.f;
          for (int i = 0; i < xsize; ++i) r += p[(size_t)(ymin + j) * IW + xmin + i] * wx[i];
          acc += r * wy[j];
        }
        out[(size_t)nc * OH * OW + (size_t)oy * OW + ox] = acc;
      }
    }
  }
  free(wy);
  free(wx);
}
