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
// wave_sum_dpp (the LayerNorm kernels): inside a row of 16 lanes by DPP (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: every lane of the row ends
// with the row's value; a few cycles each), the four rows through v_readlane.  (Until round 5: six __shfl_xor =
// ds_bpermute_b32 round trips through the LDS crossbar per reduction -- a LayerNorm row does two, back to back,
// on its critical path.)  Every lane returns the same number; the summation order is fixed.
__device__ __forceinline__ float t2h_dpp_f(float v, int ctrl_sel) {
  const int x = __builtin_bit_cast(int, v);
  int r;
  switch (ctrl_sel) {
    case 0: r = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false); break;   // quad_perm [1,0,3,2]
    case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false); break;   // quad_perm [2,3,0,1]
    case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false); break;  // row_half_mirror
    default: r = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, false); break; // row_mirror
  }
  return __builtin_bit_cast(float, r);
}
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
// (all 64 lanes must be active)
__device__ __forceinline__ float wave_sum_dpp(float v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v += t2h_dpp_f(v, i);
  // (v_readlane_b32 moves BITS: the builtin's operand is an int)
  const int x = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48));
  return (r0 + r1) + (r2 + r3);
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
// rewrite in place: there sc1 costs +1.5-2 us.  Re-measured in round 6 on the whole B = 8 sampler with one kernel
// file at a time on plain stores (profiles/r06_ln_xcd_ab.log): LayerNorm output +1.4 %, attention output +-0, the
// GEMM epilogues' split rows / Vt planes +4.5 % -- write-through stays.
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

// ---- "x8" split rows: the operand format of the split GEMM whose CROSS terms run on the 8-bit matrix instructions
// (v_mfma_scale_f32_32x32x64_f8f6f4 at twice the fp16 rate; gemm_split.hip).  a.b = ah.bh + 2^-11 (ah.bl + al.bh): the
// hi.hi product needs the fp16 planes, the cross terms only need ~4 significant bits of each factor (their weight is
// 2^-11; tools/cross_term_emulation.py: the sampler's hidden state moves by 4e-5, tolerance 2e-4).  A (row, 32-wide K
// tile) is still ONE 128-byte line:
//     [ hi16: 32 x fp16 (64 B) | hi8: 32 x e4m3 of h * s (32 B) | lo8: 32 x e4m3 of l * s (32 B) ]
// with h = fp16(x), l = (x - h) * 2^11 as above and s a power of two fixed per tensor (weights: from the matrix's own
// maximum when it is packed; activations: from a calibration evaluation, engine.SamplerNet) so that the tensor's
// values sit in e4m3's normal range [2^-6, 448].  The consumer never sees s: it is folded into the one multiplier
// that merges the cross-term accumulator in the epilogue (t2h_gemm_split_args.lo_mul = 2^-11 / (s_A s_B)).  A value
// with |x| s >= 448 raises bit 1 of the overflow word (bit 0: the fp16 range): the caller re-runs on fp16 planes.
constexpr int T2H_X8_HI8_OFF = 64, T2H_X8_LO8_OFF = 96;
constexpr float T2H_E4M3_MAX = 448.0f;

__device__ __forceinline__ unsigned t2h_e4m3x4(float a, float b, float c, float d) {
  // (no clamp: a value beyond +-448 has raised the overflow word and the result is discarded)
  unsigned w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return w;
}

__device__ __forceinline__ void t2h_store8_wt(void* ptr, unsigned a, unsigned b) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 v = {a, b};
  asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 0" ::"v"(ptr), "v"(v) : "memory");
}

// writes 8 consecutive columns c0..c0+7 (c0 % 8 == 0) of `row` in the x8 format.  PAIR_XOR != 0: the lane whose id
// differs in bit log2(PAIR_XOR) holds the OTHER 8 columns of the same 16-column group of the same row (callers
// guarantee it, and that both lanes of a pair are active): the two swap their 8-bit halves so that the first lane
// writes the 16 bytes of hi8 and the second the 16 bytes of lo8 -- two 16-byte write-through stores per lane, as for
// the fp16-plane format -- instead of one 16-byte and two 8-byte stores.
template <int PAIR_XOR = 0>
__device__ __forceinline__ void t2h_store_x8_8(uint16_t* base, int64_t row, int C, int c0, f32x4 va, f32x4 vb, float s,
                                               int* ovf) {
  const float m = fmaxf(fmaxf(fmaxf(fabsf(va[0]), fabsf(va[1])), fmaxf(fabsf(va[2]), fabsf(va[3]))),
                        fmaxf(fmaxf(fabsf(vb[0]), fabsf(vb[1])), fmaxf(fabsf(vb[2]), fabsf(vb[3]))));
  // (the fp16 range is checked on its own: with a scale <= 2^-8 a value in [65504, 448 / s) would otherwise put inf
  // into the hi plane without raising any bit)
  if (m >= 65504.0f) atomicOr(ovf, 1);
  else if (m * s >= T2H_E4M3_MAX) atomicOr(ovf, 2);
  t2h_f16x8 h;
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  f32x2_ hs[4], ls[4];
  const float s_lo = T2H_SPLIT_LO_SCALE * s;
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const float x0 = e < 4 ? va[e] : vb[e - 4], x1 = e < 4 ? va[e + 1] : vb[e - 3];
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    h[e] = h0;
    h[e + 1] = h1;
    const f32x2_ hf = {(float)h0, (float)h1}, xf = {x0, x1};
    hs[e >> 1] = hf * s;              // v_pk_mul_f32
    ls[e >> 1] = (xf - hf) * s_lo;    // v_pk_add_f32, v_pk_mul_f32
  }
  unsigned h8a = t2h_e4m3x4(hs[0][0], hs[0][1], hs[1][0], hs[1][1]), h8b = t2h_e4m3x4(hs[2][0], hs[2][1], hs[3][0], hs[3][1]);
  unsigned l8a = t2h_e4m3x4(ls[0][0], ls[0][1], ls[1][0], ls[1][1]), l8b = t2h_e4m3x4(ls[2][0], ls[2][1], ls[3][0], ls[3][1]);
  char* d = reinterpret_cast<char*>(base) + row * (int64_t)(C / 32) * T2H_SPLIT_TILE_B + (c0 >> 5) * T2H_SPLIT_TILE_B;
  t2h_store16_wt(d + (c0 & 31) * 2, h);
  if constexpr (PAIR_XOR == 0) {
    t2h_store8_wt(d + T2H_X8_HI8_OFF + (c0 & 31), h8a, h8b);
    t2h_store8_wt(d + T2H_X8_LO8_OFF + (c0 & 31), l8a, l8b);
  } else {
    // first lane of the pair (columns c0 % 16 == 0) sends its lo8 and receives the partner's hi8; the second lane
    // sends its hi8 and receives the partner's lo8.  Branch-free: one select per dword, one store.
    const bool first = (c0 & 8) == 0;
    const unsigned send_a = first ? l8a : h8a, send_b = first ? l8b : h8b;
    constexpr int CTRL = PAIR_XOR == 1 ? 0xB1 : 0x4E;  // quad_perm [1,0,3,2] / [2,3,0,1]
    const unsigned recv_a = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send_a, CTRL, 0xf, 0xf, false);
    const unsigned recv_b = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send_b, CTRL, 0xf, 0xf, false);
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_ v = {first ? h8a : recv_a, first ? h8b : recv_b, first ? recv_a : l8a, first ? recv_b : l8b};
    t2h_store16_wt(d + (first ? T2H_X8_HI8_OFF + (c0 & 31) : T2H_X8_LO8_OFF + (c0 & 31) - 8), v);
  }
}

// the same for 4 consecutive columns (c0 % 4 == 0)
__device__ __forceinline__ void t2h_store_x8_4(uint16_t* base, int64_t row, int C, int c0, f32x4 v, float s, int* ovf) {
  const float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
  if (m >= 65504.0f) atomicOr(ovf, 1);
  else if (m * s >= T2H_E4M3_MAX) atomicOr(ovf, 2);
  t2h_f16x4 h;
  float hs[4], ls[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const _Float16 hi = (_Float16)v[e];
    h[e] = hi;
    hs[e] = (float)hi * s;
    ls[e] = (v[e] - (float)hi) * (T2H_SPLIT_LO_SCALE * s);
  }
  char* d = reinterpret_cast<char*>(base) + row * (int64_t)(C / 32) * T2H_SPLIT_TILE_B + (c0 >> 5) * T2H_SPLIT_TILE_B;
  *reinterpret_cast<t2h_f16x4*>(d + (c0 & 31) * 2) = h;
  *reinterpret_cast<unsigned*>(d + T2H_X8_HI8_OFF + (c0 & 31)) = t2h_e4m3x4(hs[0], hs[1], hs[2], hs[3]);
  *reinterpret_cast<unsigned*>(d + T2H_X8_LO8_OFF + (c0 & 31)) = t2h_e4m3x4(ls[0], ls[1], ls[2], ls[3]);
}
