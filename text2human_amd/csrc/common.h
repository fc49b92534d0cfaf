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

// ---- split-row (3 x bf16 planes) helpers shared by the producers of
// t2h_gemm_split_f32 operands: x = p0 + p1 + p2, layout [rows][C/32][3][32] bf16
constexpr int T2H_SPLIT_TILE_B = 192;  // bytes per (row, 32-column tile)
typedef __bf16 t2h_bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void t2h_split3(float x, __bf16& p0, __bf16& p1, __bf16& p2) {
  p0 = (__bf16)x;
  const float r1 = x - (float)p0;
  p1 = (__bf16)r1;
  p2 = (__bf16)(r1 - (float)p1);
}

// writes 4 consecutive columns c0..c0+3 (c0 % 4 == 0) of `row` as split rows
__device__ __forceinline__ void t2h_store_split4(uint16_t* base, int64_t row, int C, int c0, f32x4 v) {
  __bf16 s[3][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) t2h_split3(v[e], s[0][e], s[1][e], s[2][e]);
  char* d = reinterpret_cast<char*>(base) + row * (int64_t)(C / 32) * T2H_SPLIT_TILE_B +
            (c0 >> 5) * T2H_SPLIT_TILE_B + (c0 & 31) * 2;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    t2h_bf16x4 w = {s[pl][0], s[pl][1], s[pl][2], s[pl][3]};
    *reinterpret_cast<t2h_bf16x4*>(d + pl * 64) = w;
  }
}
