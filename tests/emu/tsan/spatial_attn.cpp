// csrc/spatial_attn.hip: the flash-style AttnBlock attention (partial score tiles of four waves meet in LDS)
#include EMU_SOURCE
#include "common.h"
int main(int, char**) {
  const int n_img = 1, N = 96, C = 256;
  std::vector<float> qkv((size_t)n_img * N * 3 * C), out((size_t)n_img * N * C);
  fill(qkv, 0.5f);
  const int rc = t2h_spatial_attention_f32(qkv.data(), 3 * C, out.data(), C, n_img, N, C, 0.0625f, nullptr);
  printf("rc %d\n", rc);
  return rc;
}
