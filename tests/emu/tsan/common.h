// Shared by the ThreadSanitizer drivers (tests/emu/tsan/*.cpp, built by tests/test_lds_races_tsan.py): deterministic data.
#pragma once
#include <random>
#include <vector>
static std::mt19937 g_rng(1);
static inline void fill(std::vector<float>& v, float scale = 1.f) {
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& x : v) x = scale * nd(g_rng);
}
static inline void fill_f16(std::vector<uint16_t>& v, float scale) {  // fp16 bit patterns (split-row / Vt planes)
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& x : v) {
    const _Float16 f = (_Float16)(scale * nd(g_rng));
    memcpy(&x, &f, 2);
  }
}
