// csrc/vq.hip: codebook argmin (LDS-staged chunks + butterfly) and the texture-routed one
#include EMU_SOURCE
#include "common.h"
int main(int, char**) {
  const int n = 70, n_e = 512, d = 64;
  std::vector<float> z(n * d), book((size_t)n_e * d);
  std::vector<int64_t> idx(n);
  fill(z); fill(book);
  int rc = t2h_vq_l2_argmin_f32(z.data(), book.data(), idx.data(), n, n_e, d, nullptr);
  const int n2 = 12, nb = 2, ne2 = 64, d2 = 256;
  std::vector<float> z2(n2 * d2), books((size_t)nb * ne2 * d2);
  std::vector<int64_t> tex(n2), lists((size_t)nb * n2);
  fill(z2); fill(books);
  for (int i = 0; i < n2; ++i) tex[i] = i & 1;
  rc |= t2h_vq_argmin_tex_f32(z2.data(), books.data(), tex.data(), lists.data(), n2, nb, ne2, d2, 0, 0, nullptr);
  printf("rc %d\n", rc);
  return rc;
}
