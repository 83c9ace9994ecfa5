// Phase times of the LDS-window deform_conv2d forward kernel at BASELINE config 4 (2 x 256 x 100 x 136, 3 x 3, 256 -> 256, fp32):
// built with the kernel source included and -DTVMI_FW_TIMING, which makes wave 0 of a few workgroups accumulate the shader-clock
// cycles of its production phase (loop top -> barrier) and of its MFMA phase (barrier -> loop top).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DTVMI_FW_TIMING -Iinclude -Ivision_amd/csrc \
//         tools/probe/dcn_win_probe.hip build/obj/tvmi_core.o -o tools/probe/dcn_win_probe
#include "../../vision_amd/csrc/deform_conv2d.hip"

#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
  using namespace tvmi;
  const int B = argc > 1 ? atoi(argv[1]) : 2;
  const int C = 256, H = 100, W = 136, OC = 256;
  DcnParams p{};
  p.B = B; p.C = C; p.H = H; p.W = W; p.OC = OC; p.kh = 3; p.kw = 3; p.sh = p.sw = 1; p.ph = p.pw = 1; p.dh = p.dw = 1;
  p.groups = 1; p.ogroups = 1; p.oh = H; p.ow = W; p.ICg = C; p.OCg = OC; p.cpog = C; p.use_mask = 0;
  const size_t n_in = (size_t)B * C * H * W, n_off = (size_t)B * 18 * H * W, n_w = (size_t)9 * C * OC, n_out = (size_t)B * OC * H * W;
  std::vector<float> h_in(n_in), h_off(n_off), h_w(n_w), h_b(OC, 0.1f);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / (float)(1 << 24); };
  auto gauss = [&]() { float a = 0; for (int i = 0; i < 12; ++i) a += rnd(); return a - 6.f; };
  for (auto& v : h_in) v = gauss();
  for (auto& v : h_off) v = gauss();
  for (auto& v : h_w) v = 0.01f * gauss();
  float *d_in, *d_off, *d_w, *d_b, *d_out;
  hipMalloc(&d_in, n_in * 4); hipMalloc(&d_off, n_off * 4); hipMalloc(&d_w, n_w * 4); hipMalloc(&d_b, OC * 4); hipMalloc(&d_out, n_out * 4);
  hipMemcpy(d_in, h_in.data(), n_in * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_off, h_off.data(), n_off * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_w, h_w.data(), n_w * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_b, h_b.data(), OC * 4, hipMemcpyHostToDevice);
  const FwWinGeom fw = fw_win_geom(p);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) launch_f32_win(d_in, d_w, d_off, nullptr, d_b, d_out, p, C, OC, fw, B, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int it = 0; it < 10; ++it) launch_f32_win(d_in, d_w, d_off, nullptr, d_b, d_out, p, C, OC, fw, B, 0);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("B=%d  %.4f ms per launch  lds %zu B  err=%s\n", B, ms / 10, fw_win_lds_bytes(fw, 9), hipGetErrorString(hipGetLastError()));
#ifdef TVMI_FW_TIMING
  unsigned long long st[64];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(g_fw_stamp), sizeof(st));
  for (int k = 0; k < 4; ++k)
    printf("block %llu: rounds %llu  unhidden productions %.0f cyc  total %.0f cyc = %.0f cyc/round, of which %.0f at the barrier\n", st[k * 8 + 5],
           st[k * 8 + 0], (double)st[k * 8 + 1], (double)st[k * 8 + 4], (double)st[k * 8 + 4] / st[k * 8 + 0], (double)st[k * 8 + 2] / st[k * 8 + 0]);
#endif
  return 0;
}
