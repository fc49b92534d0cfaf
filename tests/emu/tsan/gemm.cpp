// csrc/gemm.hip: implicit 3x3 convolution with the GroupNorm prologue (the tokenizer's / UNet's path), then a plain GEMM
#include EMU_SOURCE
#include "common.h"
int main(int, char**) {
  const int n_img = 1, h = 16, w = 16, cin = 32, cout = 64, M = n_img * h * w;
  std::vector<float> x(M * cin), wt((size_t)cout * 9 * cin), bias(cout), out((size_t)M * cout), sc(cin), sh(cin);
  fill(x); fill(wt, 0.1f); fill(bias); fill(sc, 0.1f); fill(sh, 0.1f);
  t2h_gemm_args g{};
  g.A = x.data(); g.B = wt.data(); g.C = out.data(); g.bias = bias.data();
  g.M = M; g.N = cout; g.K = 9 * cin; g.lda = cin; g.ldb = 9 * cin; g.ldc = cout; g.a_mode = 1; g.alpha = 1.f; g.batch = 1;
  g.Hin = h; g.Win = w; g.Cin = cin; g.Hout = h; g.Wout = w; g.stride = 1; g.pad = 1;
  g.pro_scale = sc.data(); g.pro_shift = sh.data(); g.pro_ld = cin; g.pro_act = 1;
  int rc = t2h_gemm_f32(&g, nullptr);
  t2h_gemm_args p{};
  std::vector<float> a(200 * 96), b(72 * 96), c(200 * 72);
  fill(a); fill(b);
  p.A = a.data(); p.B = b.data(); p.C = c.data(); p.M = 200; p.N = 72; p.K = 96; p.lda = 96; p.ldb = 96; p.ldc = 72; p.alpha = 1.f; p.batch = 1;
  rc |= t2h_gemm_f32(&p, nullptr);
  printf("rc %d\n", rc);
  return rc;
}
