// Shared helpers for the gfx950 kernels of libt2h_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/t2h_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void t2h_set_error(const char* fmt, ...);

// Sticky overflow flag of the split-precision producers: ONE int32 word in device memory that the CALLER
// owns and passes to every producer of split rows (t2h_gemm_split_args.overflow_flag, the overflow_flag
// argument of the LayerNorm / GroupNorm-apply / attention / split_rows entry points).  A producer sets it
// when a value that is about to be written as split rows does not fit fp16's range (|x| >= 65504 would turn
// into inf / NaN planes).  The library neither allocates nor reads it back (t2h_split_overflow_async only
// enqueues the copy).

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

// ---- 16-byte write-through store (global_store_dwordx4 ... sc1) for results that are written
// once and consumed by a LATER kernel (split rows, Vt planes).  A plain store leaves the line dirty
// in this XCD's L2, to be written back at the end of the kernel -- after the epilogue, with nothing
// to overlap -- although no other XCD can use that copy; sc1 sends the bytes out during the epilogue.
// Measured on the fc1 shape (33.5 MB of split rows): 35.7 -> 33.0 us per launch, q|k|v 29.4 -> 28.1
// (profiles/r02_store_flavour.log).  NOT for the fp32 residual stream, which proj / fc2 read and
// rewrite in place: there sc1 costs +1.5-2 us.
template <typename V>
__device__ __forceinline__ void t2h_store16_wt(void* ptr, const V& v) {
  static_assert(sizeof(V) == 16, "16-byte vector");
  // (s_nop 1: a store of more than 8 bytes reads its data registers up to two wait states after it
  // issues; the compiler's hazard recognizer does not see into the asm and may overwrite them in the
  // very next instruction -- it did, once code followed the split-row stores of the GEMM epilogue)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}

// ---- split-row helpers shared by the producers of t2h_gemm_split_f32 operands.
// An fp32 value is carried as two fp16 planes, x = h + l * 2^-11 with h = fp16(x) and
// l = fp16((x - h) * 2^11) (22 significant bits; the 2^11 keeps the residual out of
// fp16's subnormal range).  Layout [rows][C/32][2 planes][32] fp16 = 128 bytes -- one
// cache line -- per (row, 32-column tile).  Operands must satisfy |x| < 65504.
constexpr int T2H_SPLIT_TILE_B = 128;        // bytes per (row, 32-column tile)
constexpr int T2H_SPLIT_PLANE_B = 64;        // bytes per plane inside a tile
constexpr float T2H_SPLIT_LO_SCALE = 2048.0f;
constexpr float T2H_SPLIT_LO_INV = 1.0f / 2048.0f;
typedef _Float16 t2h_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 t2h_f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void t2h_split2(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)((x - (float)hi) * T2H_SPLIT_LO_SCALE);
}

// raises the sticky overflow flag if any of the 8 values leaves the split-row domain
// (|x| >= 65504, +-inf).  NaN inputs can only descend from an earlier, already flagged
// overflow (inf - inf) and are not looked for.
__device__ __forceinline__ void t2h_split_guard8(int* ovf, f32x4 va, f32x4 vb) {
  const float m = fmaxf(fmaxf(fmaxf(fabsf(va[0]), fabsf(va[1])), fmaxf(fabsf(va[2]), fabsf(va[3]))),
                        fmaxf(fmaxf(fabsf(vb[0]), fabsf(vb[1])), fmaxf(fabsf(vb[2]), fabsf(vb[3]))));
  if (m >= 65504.0f) atomicOr(ovf, 1);
}
__device__ __forceinline__ void t2h_split_guard4(int* ovf, f32x4 v) {
  const float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
  if (m >= 65504.0f) atomicOr(ovf, 1);
}

// writes 8 consecutive columns c0..c0+7 (c0 % 8 == 0) of `row` as split rows: one 16-byte
// store per plane
__device__ __forceinline__ void t2h_store_split8(uint16_t* base, int64_t row, int C, int c0, f32x4 va, f32x4 vb,
                                                 int* ovf) {
  t2h_split_guard8(ovf, va, vb);
  t2h_f16x8 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    _Float16 a, b;
    t2h_split2(va[e], a, b);
    h[e] = a;
    l[e] = b;
    t2h_split2(vb[e], a, b);
    h[4 + e] = a;
    l[4 + e] = b;
  }
  char* d = reinterpret_cast<char*>(base) + row * (int64_t)(C / 32) * T2H_SPLIT_TILE_B +
            (c0 >> 5) * T2H_SPLIT_TILE_B + (c0 & 31) * 2;
  t2h_store16_wt(d, h);
  t2h_store16_wt(d + T2H_SPLIT_PLANE_B, l);
}

// writes 4 consecutive columns c0..c0+3 (c0 % 4 == 0) of `row` as split rows
__device__ __forceinline__ void t2h_store_split4(uint16_t* base, int64_t row, int C, int c0, f32x4 v, int* ovf) {
  t2h_split_guard4(ovf, v);
  t2h_f16x4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    _Float16 a, b;
    t2h_split2(v[e], a, b);
    h[e] = a;
    l[e] = b;
  }
  char* d = reinterpret_cast<char*>(base) + row * (int64_t)(C / 32) * T2H_SPLIT_TILE_B +
            (c0 >> 5) * T2H_SPLIT_TILE_B + (c0 & 31) * 2;
  *reinterpret_cast<t2h_f16x4*>(d) = h;
  *reinterpret_cast<t2h_f16x4*>(d + T2H_SPLIT_PLANE_B) = l;
}
