// csrc/attention.hip: argv = form (1 all keys, 2 key halves)
#include EMU_SOURCE
#include "common.h"
int main(int argc, char** argv) {
  const int form = argc > 1 ? atoi(argv[1]) : 2;
  const int B = 1, T = 256, H = 1, C = 64 * H;
  std::vector<uint16_t> qk((size_t)B * T * 2 * C * 2), vt((size_t)B * H * 2 * 64 * T), ys((size_t)B * T * C * 2);
  std::vector<float> y((size_t)B * T * C);
  fill_f16(qk, 1.f); fill_f16(vt, 1.f);
  int ovf = 0;
  t2h_mha_split_force_form(form);
  const int rc = t2h_mha_split_f32(qk.data(), 2 * C, vt.data(), y.data(), ys.data(), B, T, H, &ovf, nullptr);
  printf("rc %d\n", rc);
  return rc;
}
