// csrc/conv_split.hip: 3x3 'same' over split rows, both tile heights
#include EMU_SOURCE
#include "common.h"
int main(int argc, char** argv) {
  const int tile = argc > 1 ? atoi(argv[1]) : 0;
  const int n_img = 1, h = 16, w = 16, cin = 64, cout = 128, M = n_img * h * w;
  std::vector<float> bias(cout), res((size_t)M * cout), out((size_t)M * cout);
  std::vector<uint16_t> xs((size_t)M * cin * 2), ws((size_t)cout * 9 * cin * 2);
  std::vector<double> part((size_t)n_img * (h * w / 128) * 2 * cout);
  fill(bias); fill(res); fill_f16(xs, 1.f); fill_f16(ws, 0.05f);
  t2h_gemm_args g{};
  g.A = reinterpret_cast<const float*>(xs.data()); g.B = reinterpret_cast<const float*>(ws.data()); g.C = out.data();
  g.bias = bias.data(); g.residual = res.data();
  g.M = M; g.N = cout; g.K = 9 * cin; g.ldc = cout; g.ldr = cout; g.a_mode = 1; g.alpha = 1.f;
  g.Hin = h; g.Win = w; g.Cin = cin; g.Hout = h; g.Wout = w; g.stride = 1; g.pad = 1; g.batch = 1;
  g.gn_part_out = part.data();
  t2h_conv_split_force_tile(tile);
  const int rc = t2h_conv_split_f32(&g, nullptr);
  printf("rc %d\n", rc);
  return rc;
}
