// csrc/gemm_split.hip: argv = configuration id, x8 flag
#include EMU_SOURCE
#include "common.h"
int main(int argc, char** argv) {
  const int cfg = argc > 1 ? atoi(argv[1]) : 8, x8 = argc > 2 ? atoi(argv[2]) : 0;
  const int M = cfg == 9 ? 20 : 256, N = cfg == 10 ? 192 : 128, K = 256;
  std::vector<uint16_t> A((size_t)M * K * 2), B((size_t)N * K * 2);
  std::vector<float> bias(N), res((size_t)M * N), out((size_t)M * N);
  fill_f16(A, 1.f); fill_f16(B, 0.1f); fill(bias); fill(res);
  if (x8) {  // (the 8-bit planes of random fp16 patterns may decode to NaN: keep them finite)
    for (size_t t = 0; t < A.size() / 64; ++t) memset(reinterpret_cast<char*>(A.data()) + t * 128 + 64, 0x38, 64);
    for (size_t t = 0; t < B.size() / 64; ++t) memset(reinterpret_cast<char*>(B.data()) + t * 128 + 64, 0x30, 64);
  }
  t2h_gemm_split_args g{};
  g.A = A.data(); g.B = B.data(); g.C = out.data(); g.bias = bias.data(); g.residual = res.data();
  g.M = M; g.N = N; g.K = K; g.ldc = N; g.ldr = N; g.epi_act = 1; g.fmt = x8; g.lo_mul = x8 ? 1e-4f : 0.f;
  if (cfg != 9) t2h_gemm_split_force_config(cfg);
  const int rc = t2h_gemm_split_f32(&g, nullptr);
  printf("rc %d\n", rc);
  return rc;
}
