// Shared helpers for the gfx950 kernels of libt2h_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/t2h_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void t2h_set_error(const char* fmt, ...);

#define T2H_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      t2h_set_error(__VA_ARGS__);              \
      return T2H_ERR_INVALID;                  \
    }                                          \
  } while (0)

#define T2H_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      t2h_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return T2H_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

static inline bool t2h_aligned16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

// ---- 64-lane wave reductions (wave = 64 on CDNA) -------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// exp(x) as exp2(x*log2e) on v_exp_f32 with the rounding error of the product
// (and of the log2e constant) folded back in: ~1-2 ulp like expf at 5 VALU ops
// instead of ocml's ~20.  x must be finite or the caller handles -inf.
__device__ __forceinline__ float fast_exp(float x) {
  const float L = 1.44269502162933349609375f;    // float(log2 e)
  const float Llo = 1.925963033500011e-8f;       // log2 e - L
  const float t = x * L;
  float e = fmaf(x, L, -t);
  e = fmaf(x, Llo, e);
  const float r = __builtin_amdgcn_exp2f(t);
  return fmaf(r, e * 0.693147180559945309417f, r);
}
