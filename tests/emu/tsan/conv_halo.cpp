// both kernels of csrc/conv_halo.hip: two tiles side by side, two channel groups, GroupNorm tables, residual, partials
#include EMU_SOURCE
#include "common.h"
int main(int argc, char** argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 1;
  const int n_img = 1, h = 16, w = 32, cin = 64, cout = 128, M = n_img * h * w;
  std::vector<float> x(M * cin), bias(cout), res((size_t)M * cout), out((size_t)M * cout), sc(n_img * cin), sh(n_img * cin);
  std::vector<uint16_t> ws((size_t)cout * 9 * cin * 2);
  std::vector<double> part((size_t)n_img * (h * w / 128) * 2 * cout);
  fill(x); fill(bias); fill(res); fill(sc, 0.1f); fill(sh, 0.1f); fill_f16(ws, 0.05f);
  for (auto& v : sc) v += 1.f;
  int ovf = 0;
  t2h_gemm_args g{};
  g.A = x.data(); g.B = reinterpret_cast<const float*>(ws.data()); g.C = out.data(); g.bias = bias.data(); g.residual = res.data();
  g.pro_scale = sc.data(); g.pro_shift = sh.data(); g.pro_ld = cin; g.pro_act = 1;
  g.M = M; g.N = cout; g.K = 9 * cin; g.lda = cin; g.ldc = cout; g.ldr = cout; g.a_mode = 1; g.alpha = 1.f;
  g.Hin = h; g.Win = w; g.Cin = cin; g.Hout = h; g.Wout = w; g.stride = 1; g.pad = 1; g.batch = 1;
  g.gn_part_out = part.data();
  t2h_conv_halo_force_variant(variant);
  const int rc = t2h_conv_halo_f32(&g, &ovf, nullptr);
  printf("rc %d\n", rc);
  return rc;
}
