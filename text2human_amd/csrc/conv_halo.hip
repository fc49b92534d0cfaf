// Halo-staged 3x3 convolution (NHWC fp32 activations, split-row weights) on the fp16 matrix cores
// with GroupNorm apply + swish + the fp16 split folded into the operand staging -- the decoders'
// large levels (ResnetBlock conv1 / conv2, Upsample, conv_in: models/archs/vqgan_arch.py:597-617,
// 529-534,1000-1033 of the reference) without the separate elementwise pass (t2h_gn_apply_split_f32:
// a full read + write of every normalised activation) and without re-reading every input pixel once
// per tap (t2h_conv_split_f32's implicit im2col: nine 128-byte lines per output pixel and K group
// through L2 -> registers -> LDS, 1.5-2.2x the algorithmic HBM bytes by the counters).
//
//   out[pixel][co] = bias[co] + sum over taps (dy, dx), channels c of
//                    a(in[pixel + (dy - 1, dx - 1)][c]) * w[co][tap][c]            (+ residual)
//   a(x) = swish(x * scale[img][c] + shift[img][c])   (PRO 2; PRO 0: a(x) = x), zero outside the image
//
//  * A workgroup owns a 16 x 16-pixel output tile (256 GEMM rows) x 128 output channels.  For one
//    group of 32 input channels it stages the 18 x 18-pixel halo ONCE: 324 pixels x 128 bytes of fp32
//    are read, a(.) is applied, the value is split into the two fp16 planes of gemm_split.hip's
//    arithmetic (x = h + l / 2048) and written to LDS as [halo row][halo pixel][2 planes][32] fp16, 144-byte
//    pixel stride, 2816-byte row stride (conflict-free ds_read_b128 fragment reads, see CH_HROW).  The
//    nine taps are then nine K tiles whose A fragments are the SAME LDS image read at a shifted
//    pixel: a(.) and the split are computed 1.27x per element (the halo overlap) instead of 9 x
//    Cout / 128 times (the first conv_split version, VALU-bound) or in a pass of their own.
//  * Weights stream one [128 rows][32 channels of one tap] tile per K tile, and the halo of the NEXT channel
//    group is requested, converted and written to the second halo buffer while the nine taps of the current
//    group are multiplied (three 8-channel pieces per thread).  HOW is what the two kernels of this file differ
//    in: kernel 1 (the first version, kept as the A/B reference) moves the weight tiles through registers and
//    converts a whole piece in one block; kernel 2 (the default) uses LDS-DMA for the weights, keeps two whole
//    fragment sets in flight and cuts the conversion into stages behind the matrix instructions -- see the
//    banners below and profiles/r05_conv_halo_phase_timing.log for the measurements that led there
//    (level-0 layer of the decoder, 8 x 512 x 256 pixels, 128 -> 128 channels: 1427 us for the two-kernel
//    path, 1008 us kernel 1, 892 us kernel 2).
//  * K order is [channel group][tap] (conv_split: [tap][channel group]): same products, a different
//    fp32 summation order.  Three partial products per k16 step (hi.hi, hi.lo, lo.hi), two fp32
//    accumulator sets merged in the epilogue -- conv_split.hip's arithmetic unchanged.
//  * Epilogue as conv_split.hip: bias, residual, fp32 rows out, per-(image, 128 pixels, channel) fp64
//    GroupNorm partials of the FINAL values (a 128-pixel chunk here is 8 rows x 16 pixels of a tile;
//    t2h_groupnorm_finalize_f32 only needs every pixel in exactly one chunk).
#include <type_traits>

#include "common.h"

namespace {

typedef t2h_f16x8 f16x8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int CH_T = 16;                          // output tile edge (pixels)
constexpr int CH_HW = CH_T + 2;                   // halo edge
constexpr int CH_HP = CH_HW * CH_HW;              // halo pixels (324)
constexpr int CH_BM = CH_T * CH_T, CH_BN = 128;   // GEMM tile
constexpr int CH_WM = 4, CH_WN = 2, CH_NT = 64 * CH_WM * CH_WN;
constexpr int CH_TM = 2, CH_TN = 2;               // 32 x 32 accumulator tiles of a 64 x 64 wave tile
constexpr int CH_ROW = 144;                       // LDS bytes per halo pixel (and per weight row of the register-staged kernel): 128 + 16
// bytes per halo ROW in LDS: 18 pixels x 144 padded to a multiple of 256 (64 banks x 4 B).  A fragment read covers two
// halo rows (16 + 16 pixels); ds_read_b128 is served in groups of 16 lanes that mix the two ({0-3, 12-15, 20-27}, ...):
// with the second row a multiple of 256 bytes further, its pixels fall on the banks of the pixels the group does not
// take from the first row -- conflict-free like 32 consecutive 144-byte rows (18 x 144 = 2592 left two 2-way conflicts
// in every group)
constexpr int CH_HROW = 2816;
static_assert(CH_HROW % 256 == 0 && CH_HROW >= CH_HW * CH_ROW, "halo row stride");
constexpr int CH_HALO_B = CH_HW * CH_HROW;        // 50688
constexpr int CH_PIECES = CH_HP * 4;              // 8-channel pieces of a halo (1296)
constexpr int CH_PJ = (CH_PIECES + CH_NT - 1) / CH_NT;  // per thread (3; the third only for 272 threads)
static_assert(CH_PJ == 3, "three pieces per thread and channel group");
// (the pieces beyond the halo -- third piece of threads 272..511 -- are written to a scratch area instead of being
// predicated off: the tap bodies stay free of divergent branches)
constexpr int CH_DUMMY_B = (CH_PJ * CH_NT - CH_PIECES) * 16 + 64 + 16;
constexpr int CH_O_LD = 64 + 4;                   // epilogue staging: floats per row of a wave tile
constexpr int CH_OW = 64 * CH_O_LD;
constexpr int CH_EPI_B = CH_OW * 4 * CH_WM * CH_WN;

struct ch_piece {
  f32x4 a, b;
};

// ---- explicit vector-memory requests of the LDS-DMA kernel: the compiler neither sees them nor waits for them; the
// kernel counts (s_waitcnt vmcnt(N), N = requests younger than the one needed: they return in order).
// 16 bytes per lane from gbase (scalar) + goff (per lane) into registers
__device__ __forceinline__ void ch_gload16(f32x4& dst, const char* gbase, unsigned goff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(goff), "s"(gbase) : "memory");
}
// 64 lanes x 16 bytes from gbase + goff straight into the 1 KiB of LDS at lds_dst (wave-uniform), lane-linear
__device__ __forceinline__ void ch_dma16(char* lds_dst, const char* gbase, unsigned goff) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(goff), "s"(gbase), "s"(dst) : "memory");
}
template <int N>
__device__ __forceinline__ void ch_wait_vm(f32x4& v) {  // ... and v is not read before
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N));
}
template <int N>
__device__ __forceinline__ void ch_wait_vm_all() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// an opaque copy of a register: what is derived from it is computed behind this point (not hoisted above a loop and
// carried through it in registers the loop needs)
__device__ __forceinline__ int ch_launder(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// ... and for a VALUE: the instructions that produce it are placed in front of this point (the IR-level passes sink pure
// arithmetic to its use, across __builtin_amdgcn_sched_barrier: a stage of the interleaved conversion ends by pinning
// what it produced)
template <typename T>
__device__ __forceinline__ void ch_pin(T& v) {
  asm volatile("" : "+v"(v));
}

// Phase stamps of a debug build (-DT2H_HALO_PROBE, tools/conv_halo_phase_timing.py; never in the product library):
// lane 0 of waves 0 and 4 of the first 64 workgroups stores s_memtime (shader clock) at five points of every tap of
// the LDS-DMA kernel's main loop, probe[((workgroup * 2 + wave / 4) * 40 + K tile) * 8 + i]; slot 36 holds entry,
// prologue done, main loop done, end.
#ifdef T2H_HALO_PROBE
#define HALO_PROBE_ARG , long long* const probe
#define HALO_STAMP(var)                    \
  __builtin_amdgcn_sched_barrier(0);       \
  var = __builtin_amdgcn_s_memtime();      \
  __builtin_amdgcn_sched_barrier(0)
#else
#define HALO_PROBE_ARG
#define HALO_STAMP(var)
#endif

// ---- geometry of a workgroup: output tile, column tile, this thread's halo pieces
struct ch_geom {
  int img, trem, tpi, y0, x0, n0, G;
};
__device__ __forceinline__ ch_geom ch_geometry(const t2h_gemm_args& p) {
  ch_geom q;
  const int tiles_x = p.Wout / CH_T;
  q.tpi = tiles_x * (p.Hout / CH_T);
  const int nbx = (p.N + CH_BN - 1) / CH_BN, nby = p.M / CH_BM;
  // XCD-aware tile mapping (see gemm.hip): consecutive tiles of an image stay on one XCD's L2
  const int total = nbx * nby, b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3, qq = total >> 3, r = total & 7;
  const int lin = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + slot;
  const int mt = lin / nbx;
  q.n0 = (lin - mt * nbx) * CH_BN;
  q.img = mt / q.tpi;
  q.trem = mt - q.img * q.tpi;
  const int ty = q.trem / tiles_x, tx = q.trem - ty * tiles_x;
  q.y0 = ty * CH_T;
  q.x0 = tx * CH_T;
  q.G = p.Cin / 32;  // channel groups; K tile (tap t, group g) of the packed weights is t * G + g
  return q;
}

// a(.) + split of NE (4 or 8) consecutive channels, two per instruction (v_pk_*_f32), written to LDS at d (hi plane)
// and d + 64 (lo plane).  a(x) = swish(x * sc + sh) (PRO 2) as x * rcp(1 + exp2(-x log2 e)): v_exp_f32 / v_rcp_f32
// directly (the elementwise pass and the exact-fp32 kernels use a corrected exp and an IEEE division: 25 instead of 9
// VALU instructions per element; the difference is a few ulp of the activation).  m = 0 outside the image (the
// reference pads the ACTIVATED tensor with zeros), 1 inside.
template <int PRO, int NE>
__device__ __forceinline__ void ch_put(const float (&v)[NE], const float (&sc)[NE], const float (&sh)[NE], float m,
                                       float& amax, char* d) {
  _Float16 h[NE], l[NE];
#pragma unroll
  for (int e = 0; e < NE; e += 2) {
    f32x2 x = {v[e], v[e + 1]};
    if (PRO) {
      const f32x2 s2 = {sc[e], sc[e + 1]}, b2 = {sh[e], sh[e + 1]};
      x = x * s2 + b2;
      if (PRO == 2) {
        const f32x2 tt = x * -1.44269504088896340736f;
        const f32x2 dd = f32x2{__builtin_amdgcn_exp2f(tt[0]), __builtin_amdgcn_exp2f(tt[1])} + 1.0f;
        const f32x2 rr = f32x2{__builtin_amdgcn_rcpf(dd[0]), __builtin_amdgcn_rcpf(dd[1])} * m;
        x = x * rr;
      } else {
        x = x * m;
      }
    } else {
      x = x * m;
    }
    amax = fmaxf(amax, fmaxf(fabsf(x[0]), fabsf(x[1])));
    const _Float16 h0 = (_Float16)x[0], h1 = (_Float16)x[1];
    const f32x2 hf = {(float)h0, (float)h1};
    const f32x2 lf = (x - hf) * T2H_SPLIT_LO_SCALE;
    h[e] = h0;
    h[e + 1] = h1;
    l[e] = (_Float16)lf[0];
    l[e + 1] = (_Float16)lf[1];
  }
  if constexpr (NE == 8) {
    f16x8 hv, lv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      hv[e] = h[e];
      lv[e] = l[e];
    }
    *reinterpret_cast<f16x8*>(d) = hv;
    *reinterpret_cast<f16x8*>(d + T2H_SPLIT_PLANE_B) = lv;
  } else {
    f16x4 hv, lv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hv[e] = h[e];
      lv[e] = l[e];
    }
    *reinterpret_cast<f16x4*>(d) = hv;
    *reinterpret_cast<f16x4*>(d + T2H_SPLIT_PLANE_B) = lv;
  }
}
template <int PRO>
__device__ __forceinline__ void ch_put_piece(const ch_piece& v, const f32x4 (&sc)[2], const f32x4 (&sh)[2], float m,
                                             float& amax, char* d) {
  float x[8], s[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    x[e] = e < 4 ? v.a[e] : v.b[e - 4];
    s[e] = PRO ? sc[e >> 2][e & 3] : 0.f;
    b[e] = PRO ? sh[e >> 2][e & 3] : 0.f;
  }
  ch_put<PRO, 8>(x, s, b, m, amax, d);
}
template <int PRO>
__device__ __forceinline__ void ch_put_half(const f32x4& v, const f32x4& sc, const f32x4& sh, float m, float& amax, char* d) {
  float x[4], s[4], b[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    x[e] = v[e];
    s[e] = PRO ? sc[e] : 0.f;
    b[e] = PRO ? sh[e] : 0.f;
  }
  ch_put<PRO, 4>(x, s, b, m, amax, d);
}

// ---- epilogue (conv_split.hip's): the accumulators (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) are
// merged (hi.hi + cross terms / 2048 + bias), transposed through the idle LDS so that every lane owns 8 consecutive
// channels of a pixel, the residual is added, fp32 rows go out; then the GroupNorm partials of the produced tensor:
// lane j sums channel j of its wave tile over the wave's 64 pixels in fp64, the two M-waves of a 128-pixel chunk are
// added through LDS in wave order, one plain store per (chunk, channel) -- no atomics, bit-reproducible.
// Chunk = 2 * (tile inside the image) + (M-wave / 2).  All waves must have left the main loop (LDS is reused).
// (Round 5 also built the epilogue WITHOUT the LDS pass -- 64 global_store_dword per lane straight from the accumulator
// layout, two full 128-byte lines each, sums from registers: 12.5 k instead of 14.5 k cycles per workgroup without a
// residual, but 1310 instead of 968 us per launch WITH one (64 four-byte residual loads per lane); the phase is bound
// by every CU's 128 KiB output burst meeting HBM at the same time, not by LDS: removed,
// profiles/r05_conv_halo_epilogue_ab.log.)
__device__ __forceinline__ void ch_epilogue(const t2h_gemm_args& p, const ch_geom& q, const f32x16 (&acc)[2][CH_TM][CH_TN],
                                            char* smem) {
  const int tid = ch_launder(threadIdx.x), lane = tid & 63, wave = tid >> 6;  // (nothing of this is computed before the main loop)
  const int l31 = lane & 31, hh = lane >> 5;
  const int wmi = wave / CH_WN, wni = wave % CH_WN;
  const int wn0 = wni * 64;
  float* const Ot = reinterpret_cast<float*>(smem) + wave * CH_OW;
#pragma unroll
  for (int ti = 0; ti < CH_TM; ++ti)
#pragma unroll
    for (int tj = 0; tj < CH_TN; ++tj) {
      const int col = q.n0 + wn0 + tj * 32 + l31;
      const float bv = (p.bias && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Ot[(ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * CH_O_LD + tj * 32 + l31] =
            fmaf(acc[1][ti][tj][r], T2H_SPLIT_LO_INV, acc[0][ti][tj][r]) + bv;
    }
  __syncthreads();
  constexpr int CPR = 64 / 8;         // 8-column chunks per staged row
  constexpr int NCH = 64 * CPR / 64;  // chunks per lane
  const int64_t img_row0 = (int64_t)q.img * p.Hout * p.Wout;
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int c = lane + 64 * it;
    const int rl = c / CPR, cc = (c - rl * CPR) * 8;
    // GEMM row 64 wmi + rl of the tile = pixel (wmi * 4 + rl / 16, rl % 16)
    const int64_t row = img_row0 + (int64_t)(q.y0 + wmi * 4 + (rl >> 4)) * p.Wout + q.x0 + (rl & 15);
    const int col = q.n0 + wn0 + cc;
    if (col >= p.N) continue;
    f32x4 va = *reinterpret_cast<const f32x4*>(Ot + rl * CH_O_LD + cc);
    f32x4 vb = *reinterpret_cast<const f32x4*>(Ot + rl * CH_O_LD + cc + 4);
    if (p.residual) {
      va += *reinterpret_cast<const f32x4*>(p.residual + row * p.ldr + col);
      vb += *reinterpret_cast<const f32x4*>(p.residual + row * p.ldr + col + 4);
    }
    *reinterpret_cast<f32x4*>(p.C + row * p.ldc + col) = va;
    *reinterpret_cast<f32x4*>(p.C + row * p.ldc + col + 4) = vb;
    if (p.gn_part_out && p.residual) {  // final values back into the staging tile for the channel sums below
      *reinterpret_cast<f32x4*>(Ot + rl * CH_O_LD + cc) = va;
      *reinterpret_cast<f32x4*>(Ot + rl * CH_O_LD + cc + 4) = vb;
    }
  }
  if (p.gn_part_out) {
    __syncthreads();
    double su = 0.0, sq = 0.0;
#pragma unroll 8
    for (int r = 0; r < 64; ++r) {
      const double v = (double)Ot[r * CH_O_LD + lane];
      su += v;
      sq = fma(v, v, sq);
    }
    __syncthreads();  // every wave is done reading its staging tile: the start of LDS becomes the table
    double* const red = reinterpret_cast<double*>(smem);  // [wave][64][2]
    red[(wave * 64 + lane) * 2] = su;
    red[(wave * 64 + lane) * 2 + 1] = sq;
    __syncthreads();
    constexpr int WPC = 128 / 64;  // M-waves per 128-pixel chunk (2)
    if (wmi % WPC == 0) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int k = 0; k < WPC; ++k) {
        a += red[(((wmi + k) * CH_WN + wni) * 64 + lane) * 2];
        b += red[(((wmi + k) * CH_WN + wni) * 64 + lane) * 2 + 1];
      }
      const int col = q.n0 + wni * 64 + lane;
      if (col < p.N) {
        const int chunks = q.tpi * (CH_BM / 128), chunk = q.trem * (CH_BM / 128) + wmi / WPC;
        double* dst = p.gn_part_out + (((int64_t)q.img * chunks + chunk) * 2) * p.N + col;
        dst[0] = a;
        dst[p.N] = b;
      }
    }
  }
}

constexpr int PA[3] = {1, 0, 0};  // the three partial products: plane of A, plane of B, accumulator set
constexpr int PB[3] = {0, 1, 0};
constexpr int PC[3] = {1, 1, 0};

// =====================================================================================================================
// Kernel 1 (variant 0): weight tiles through registers (global -> VGPR -> ds_write, two buffers), fragments read where
// the compiler places them, GroupNorm tables of a group in registers; the waves of a SIMD convert a whole piece in
// alternate taps before their matrix instructions.  The first version of this file: kept as the A/B reference
// (t2h_conv_halo_force_variant(0)).
constexpr int CH_BT_B = CH_BN * CH_ROW;           // 18432: padded weight tile
constexpr int CH_LOOP_B = 2 * CH_HALO_B + 2 * CH_BT_B;

template <int PRO>  // PRO 0: plain split; 2: GroupNorm tables + swish
__global__ __launch_bounds__(CH_NT, 2) void conv_halo_reg_kernel(const t2h_gemm_args p, int* ovf) {
  constexpr int TM = CH_TM, TN = CH_TN;
  constexpr int SMEM_B = CH_EPI_B > CH_LOOP_B + CH_DUMMY_B ? CH_EPI_B : CH_LOOP_B + CH_DUMMY_B;
  static_assert(SMEM_B <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) char smem[SMEM_B];
  char* const halo = smem;
  char* const btile = smem + 2 * CH_HALO_B;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wmi = wave / CH_WN, wni = wave % CH_WN;
  const int wn0 = wni * 64;
  const ch_geom q = ch_geometry(p);
  const int G = q.G;
  const int phase = __builtin_amdgcn_readfirstlane(wave) >> 2;  // wave-uniform: scalar branches
  const int c8 = tid & 3;                                        // this thread's 8-channel piece inside a group, every j

  // ---- this thread's halo pieces: source pixel (clamped; nearest-x2: >> ups), inside-the-image bit, LDS offset
  const float* a_src[CH_PJ];
  float a_m[CH_PJ];
  int a_dst[CH_PJ], a_sel[CH_PJ];  // LDS offset in halo buffer 0; what selecting buffer 1 adds (0 for the scratch slots)
#pragma unroll
  for (int j = 0; j < CH_PJ; ++j) {
    const int pi = tid + CH_NT * j;
    const int hp = min(pi >> 2, CH_HP - 1);
    const int hy = hp / CH_HW, hx = hp - hy * CH_HW;
    const int Y = q.y0 + hy - 1, X = q.x0 + hx - 1;  // in the convolution's input geometry (= output geometry)
    a_m[j] = ((unsigned)Y < (unsigned)p.Hout && (unsigned)X < (unsigned)p.Wout) ? 1.0f : 0.0f;
    const int sy = min(max(Y, 0), p.Hout - 1) >> p.ups, sx = min(max(X, 0), p.Wout - 1) >> p.ups;
    a_src[j] = p.A + ((int64_t)(q.img * p.Hin + sy) * p.Win + sx) * p.lda + c8 * 8;
    a_dst[j] = pi < CH_PIECES ? hy * CH_HROW + hx * CH_ROW + c8 * 16 : CH_LOOP_B + (pi - CH_PIECES) * 16;
    a_sel[j] = pi < CH_PIECES ? CH_HALO_B : 0;
  }
  const float* const t_scale = PRO ? p.pro_scale + (int64_t)q.img * p.pro_ld + c8 * 8 : nullptr;
  const float* const t_shift = PRO ? p.pro_shift + (int64_t)q.img * p.pro_ld + c8 * 8 : nullptr;
  float amax = 0.f;  // largest |value| this thread splits: one overflow check at the end

  const int pc = tid & 7;  // 16-byte piece of a weight row's K tile
  const int64_t brow_b = (int64_t)9 * G * T2H_SPLIT_TILE_B;
  const char* b_src[2];
  int b_dst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = min(q.n0 + (tid >> 3) + (CH_NT / 8) * i, p.N - 1);  // clamped: extra columns are never stored
    b_src[i] = reinterpret_cast<const char*>(p.B) + (int64_t)n * brow_b + pc * 16;
    b_dst[i] = ((tid >> 3) + (CH_NT / 8) * i) * CH_ROW + pc * 16;
  }

  auto load_piece = [&](int j, int g) {
    ch_piece v;
    const float* s = a_src[j] + g * 32;
    v.a = *reinterpret_cast<const f32x4*>(s);
    v.b = *reinterpret_cast<const f32x4*>(s + 4);
    return v;
  };
  auto put_piece = [&](int j, const ch_piece& v, const f32x4 (&sc)[2], const f32x4 (&sh)[2], int buf) {
    ch_put_piece<PRO>(v, sc, sh, a_m[j], amax, smem + a_dst[j] + buf * a_sel[j]);
  };
  auto load_tables = [&](int g, f32x4 (&sc)[2], f32x4 (&sh)[2]) {
    if (PRO) {
      sc[0] = *reinterpret_cast<const f32x4*>(t_scale + g * 32);
      sc[1] = *reinterpret_cast<const f32x4*>(t_scale + g * 32 + 4);
      sh[0] = *reinterpret_cast<const f32x4*>(t_shift + g * 32);
      sh[1] = *reinterpret_cast<const f32x4*>(t_shift + g * 32 + 4);
    }
  };

  f32x16 acc[2][TM][TN];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;

  // ---- prologue: halo of group 0, weight tile (tap 0, group 0)
  {
    f32x4 sc[2], sh[2];
    load_tables(0, sc, sh);
    ch_piece v[CH_PJ];
#pragma unroll
    for (int j = 0; j < CH_PJ; ++j) v[j] = load_piece(j, 0);
    u32x4 rb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const u32x4*>(b_src[i]);
#pragma unroll
    for (int j = 0; j < CH_PJ; ++j) put_piece(j, v[j], sc, sh, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(btile + b_dst[i]) = rb[i];
  }
  __syncthreads();

  // fragment addresses: A row (pixel) of lane = tile-local (wmi * 4 + ti * 2 + (l31 >> 4), l31 & 15); tap (dy, dx)
  // reads halo pixel (row + dy, col + dx)
  const int a_lane = (wmi * 4 + (l31 >> 4)) * CH_HROW + (l31 & 15) * CH_ROW + hh * 16;
  const int b_lane = (wn0 + l31) * CH_ROW + hh * 16;

  for (int g = 0; g < G; ++g) {
    const int gn = min(g + 1, G - 1);  // (last group: stages its own halo again into the idle buffer -- no branch)
    const char* const hcur = halo + (g & 1) * CH_HALO_B + a_lane;
    const int nbuf = (g + 1) & 1;
    f32x4 sc[2], sh[2];
    load_tables(gn, sc, sh);
    ch_piece pv[CH_PJ];
    auto tap = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      const int kt = g * 9 + t;
      // weight tile of the next K tile: (tap t + 1, g) or (tap 0, next group)
      const int k_next = t == 8 ? gn : (t + 1) * G + g;
      u32x4 rb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const u32x4*>(b_src[i] + (int64_t)k_next * T2H_SPLIT_TILE_B);
      // halo of the next group: the waves of a SIMD take turns (waves w and w + 4 share one: phase 0 = waves 0..3,
      // phase 1 = waves 4..7).  Piece j is requested at tap 2 j + phase and converted at the TOP of tap 2 j + 2 + phase,
      // before that wave's matrix instructions.
      if constexpr (t <= 5)
        if (phase == (t & 1)) pv[t / 2] = load_piece(t / 2, gn);
      // (the requests stay at the top of the tap: left alone, the scheduler sinks them to just before their use at
      // the end of the tap, a whole L2 round trip in front of the barrier)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (t >= 2 && t <= 7)
        if (phase == (t & 1)) put_piece((t - 2) / 2, pv[(t - 2) / 2], sc, sh, nbuf);

      const char* Ab = hcur + (t / 3) * CH_HROW + (t % 3) * CH_ROW;
      const char* Bb = btile + (kt & 1) * CH_BT_B + b_lane;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f16x8 af[TM][2], bfr[TN][2];
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
            af[ti][pl] = *reinterpret_cast<const f16x8*>(Ab + ti * 2 * CH_HROW + pl * 64 + u * 32);
#pragma unroll
        for (int tj = 0; tj < TN; ++tj)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
            bfr[tj][pl] = *reinterpret_cast<const f16x8*>(Bb + tj * 32 * CH_ROW + pl * 64 + u * 32);
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int ti = 0; ti < TM; ++ti)
#pragma unroll
            for (int tj = 0; tj < TN; ++tj)
              acc[PC[pr]][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ti][PA[pr]], bfr[tj][PB[pr]],
                                                                            acc[PC[pr]][ti][tj], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(btile + ((kt + 1) & 1) * CH_BT_B + b_dst[i]) = rb[i];
      __syncthreads();
    };
    tap(std::integral_constant<int, 0>{});
    tap(std::integral_constant<int, 1>{});
    tap(std::integral_constant<int, 2>{});
    tap(std::integral_constant<int, 3>{});
    tap(std::integral_constant<int, 4>{});
    tap(std::integral_constant<int, 5>{});
    tap(std::integral_constant<int, 6>{});
    tap(std::integral_constant<int, 7>{});
    tap(std::integral_constant<int, 8>{});
  }
  if (amax >= 65504.0f) atomicOr(ovf, 1);
  ch_epilogue(p, q, acc, smem);
}

// =====================================================================================================================
// Kernel 2 (variant 1, the default): what the phase stamps of kernel 1 asked for (tools/conv_halo_phase_timing.py,
// profiles/r05_conv_halo_*):
//  * weight tiles by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write, no compiler-placed
//    s_waitcnt vmcnt(0) in front of it), three buffers, requested two taps ahead; the image is unpadded (a request
//    fills 1 KiB lane-linearly) and XOR-swizzled on the request's source address and on the read address: piece c
//    of row r lives at piece c ^ ((r >> 1) & 7) (gemm_split.hip's image);
//  * two whole fragment sets (k16 steps 0 / 1 of a tap, 32 registers each): a set is requested twelve matrix
//    instructions before it is multiplied -- step 1 of tap t runs behind the barrier, in tap t + 1, while step 0 of
//    that tap is on its way from LDS;
//  * the halo of the next group is staged half a piece (4 channels) per wave and tap in taps 2 .. 7, the conversion cut
//    into stages of two to four independent instructions, one stage behind each matrix instruction of the tap
//    (issued as one block it is a serial chain of ~110 instructions at 11-12 cycles each -- longer than the partner
//    wave's 24 matrix instructions -- and sat on the critical path of the tap wherever it was put); the halo
//    requests are explicit (counted waits), issued at the END of a tap, two taps before the conversion; the
//    GroupNorm tables of the image live in LDS.
// piece j of thread tid (8 channels c8 of halo pixel hp): byte offset of the source pixel inside the image (clamped;
// nearest-x2: >> ups), LDS offset in halo buffer 0, what selecting buffer 1 adds (0 for the scratch slots), and the
// inside-the-image factor.  The LDS-DMA kernel RECOMPUTES this where it is needed (twenty scalar-ish instructions,
// twice per tap) instead of holding twelve registers through the main loop: with two fragment sets in flight there are
// none to spare (the first build spilled exactly these).
struct ch_pgeom {
  unsigned off;
  int dst, sel;
  float m;
};
__device__ __forceinline__ ch_pgeom ch_piece_geom(int j, int tid, const ch_geom& q, const t2h_gemm_args& p, int halo_base, int dummy_base) {
  ch_pgeom r;
  const int pi = tid + CH_NT * j, c8 = tid & 3;
  const int hp = min(pi >> 2, CH_HP - 1);
  const int hy = hp / CH_HW, hx = hp - hy * CH_HW;
  const int Y = q.y0 + hy - 1, X = q.x0 + hx - 1;  // in the convolution's input geometry (= output geometry)
  r.m = ((unsigned)Y < (unsigned)p.Hout && (unsigned)X < (unsigned)p.Wout) ? 1.0f : 0.0f;
  const int sy = min(max(Y, 0), p.Hout - 1) >> p.ups, sx = min(max(X, 0), p.Wout - 1) >> p.ups;
  r.off = ((unsigned)(sy * p.Win + sx) * (unsigned)p.lda + c8 * 8) * 4u;  // < 2 GiB per image (host-checked)
  r.dst = pi < CH_PIECES ? halo_base + hy * CH_HROW + hx * CH_ROW + c8 * 16 : dummy_base + (pi - CH_PIECES) * 16;
  r.sel = pi < CH_PIECES ? CH_HALO_B : 0;
  return r;
}

constexpr int CH_BD_B = CH_BN * 128;              // 16384: unpadded weight tile of the DMA image
constexpr int CH_DMA_LOOP_B = 2 * CH_HALO_B + 3 * CH_BD_B;
constexpr int CH_TAB_OFF = CH_DMA_LOOP_B + CH_DUMMY_B;
static_assert(CH_TAB_OFF % 16 == 0, "table alignment");

template <int PRO>
__global__ __launch_bounds__(CH_NT, 2) void conv_halo_dma_kernel(const t2h_gemm_args p, int* ovf HALO_PROBE_ARG) {
  constexpr int TM = CH_TM, TN = CH_TN;
  constexpr int TAB_B = PRO ? 2 * 512 * 4 : 0;  // scale[Cin] | shift[Cin] of this image (Cin <= 512, host-checked)
  constexpr int SMEM_B = CH_EPI_B > CH_TAB_OFF + TAB_B ? CH_EPI_B : CH_TAB_OFF + TAB_B;
  static_assert(SMEM_B <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_B];
  // weight tiles FIRST: their fragment reads address (row, swizzled piece) + buffer + 32-row block -- with the buffers
  // below 64 KiB the last two are the instruction's 16-bit immediate (behind the halos they were a dozen address
  // registers, hoisted out of the loop and spilled)
  char* const btile = smem;
  char* const halo = smem + 3 * CH_BD_B;

  [[maybe_unused]] unsigned long long pst[4] = {0, 0, 0, 0};
  HALO_STAMP(pst[0]);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wmi = wave / CH_WN, wni = wave % CH_WN;
  const int wn0 = wni * 64;
  const ch_geom q = ch_geometry(p);
  const int G = q.G, nk = 9 * G;
  const int uwave = __builtin_amdgcn_readfirstlane(wave);  // wave-uniform: scalar arithmetic
  const int c8 = tid & 3;

  const char* const a_img = reinterpret_cast<const char*>(p.A + (int64_t)q.img * p.Hin * p.Win * p.lda);  // scalar
  float amax = 0.f;

  // ---- weight tile requests of this wave: 8-row groups uwave and uwave + 8 of the 128-row tile image; lane -> row
  // 8 grp + lane / 8, physical piece lane % 8 = logical piece (lane % 8) ^ ((row >> 1) & 7)
  unsigned w_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave + 8 * i) * 8 + (lane >> 3), pcs = (lane & 7) ^ ((r >> 1) & 7);
    const int n = min(q.n0 + r, p.N - 1);  // clamped: extra columns are never stored
    w_off[i] = (unsigned)n * (unsigned)(nk * T2H_SPLIT_TILE_B) + pcs * 16;
  }
  auto dma_tile = [&](int kidx, int buf) {  // K tile kidx of the packed weights -> tile buffer buf
    const char* const base = reinterpret_cast<const char*>(p.B) + (int64_t)kidx * T2H_SPLIT_TILE_B;
#pragma unroll
    for (int i = 0; i < 2; ++i) ch_dma16(btile + buf * CH_BD_B + (uwave + 8 * i) * 1024, base, w_off[i]);
  };

  f32x16 acc[2][TM][TN];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;

  // ---- prologue: weight tiles of taps 0 and 1 (group 0), the image's tables, the halo of group 0
  {
    dma_tile(0, 0);
    dma_tile(G > 0 ? G : 0, 1);  // (tap 1, group 0) = K tile 1 * G + 0
    if constexpr (PRO != 0) {
      float* const tabw = reinterpret_cast<float*>(smem + CH_TAB_OFF);  // (scale, shift) per channel
      for (int i = tid * 2; i < p.Cin; i += CH_NT * 2) {
        const f32x2 a = *reinterpret_cast<const f32x2*>(p.pro_scale + (int64_t)q.img * p.pro_ld + i);
        const f32x2 b = *reinterpret_cast<const f32x2*>(p.pro_shift + (int64_t)q.img * p.pro_ld + i);
        *reinterpret_cast<f32x4*>(tabw + 2 * i) = f32x4{a[0], b[0], a[1], b[1]};
      }
    }
    f32x4 sc[2] = {}, sh[2] = {};
    if (PRO) {
      const float* ts = p.pro_scale + (int64_t)q.img * p.pro_ld + c8 * 8;
      const float* tb = p.pro_shift + (int64_t)q.img * p.pro_ld + c8 * 8;
      sc[0] = *reinterpret_cast<const f32x4*>(ts);
      sc[1] = *reinterpret_cast<const f32x4*>(ts + 4);
      sh[0] = *reinterpret_cast<const f32x4*>(tb);
      sh[1] = *reinterpret_cast<const f32x4*>(tb + 4);
    }
    ch_piece v[CH_PJ];
    ch_pgeom pg[CH_PJ];
#pragma unroll
    for (int j = 0; j < CH_PJ; ++j) {
      pg[j] = ch_piece_geom(j, tid, q, p, 3 * CH_BD_B, CH_DMA_LOOP_B);
      v[j].a = *reinterpret_cast<const f32x4*>(a_img + pg[j].off);
      v[j].b = *reinterpret_cast<const f32x4*>(a_img + pg[j].off + 16);
    }
#pragma unroll
    for (int j = 0; j < CH_PJ; ++j) ch_put_piece<PRO>(v[j], sc, sh, pg[j].m, amax, smem + pg[j].dst);
    ch_wait_vm_all<0>();  // the two weight tiles have landed
  }
  __syncthreads();
  HALO_STAMP(pst[1]);

  // fragment addresses.  A: row (pixel) of lane = tile-local (wmi * 4 + ti * 2 + (l31 >> 4), l31 & 15), tap (dy, dx)
  // reads halo pixel (row + dy, col + dx).  B: row wn0 + tj * 32 + l31 of the swizzled image, logical piece
  // plane * 4 + u * 2 + hh
  const int a_lane = (wmi * 4 + (l31 >> 4)) * CH_HROW + (l31 & 15) * CH_ROW + hh * 16;
  // (logical piece = hh | 2 u | 4 plane, swizzle = (row >> 1) & 7: the offset of (plane, u) is the offset of (0, 0) with
  // bits 5 (u) and 6 (plane) flipped -- one register instead of four)
  const int b_off0 = (wn0 + l31) * 128 + ((hh ^ ((l31 >> 1) & 7)) * 16);

  // fragment set F of k16 step u: A (ti, plane) in F[ti * 2 + pl], B (tj, plane) in F[4 + tj * 2 + pl]
  auto read_frags = [&](int u, const char* Ab, const char* Bt, f16x8 (&F)[8]) {
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        F[ti * 2 + pl] = *reinterpret_cast<const f16x8*>(Ab + ti * 2 * CH_HROW + pl * 64 + u * 32);
#pragma unroll
    for (int tj = 0; tj < TN; ++tj)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        F[4 + tj * 2 + pl] = *reinterpret_cast<const f16x8*>(Bt + tj * 32 * 128 + (b_off0 ^ (u * 32 + pl * 64)));
  };
  auto mfma12 = [&](const f16x8 (&F)[8]) {
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
      for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < TN; ++tj)
          acc[PC[pr]][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[ti * 2 + PA[pr]], F[4 + tj * 2 + PB[pr]],
                                                                        acc[PC[pr]][ti][tj], 0, 0, 0);
  };
  static_assert(TM == 2 && TN == 2, "fragment set layout");
  f16x8 R0[8], R1[8];
  f32x4 hv[2];                                           // halo requests in flight, by tap parity
  int hg[2] = {0, 0};                                    // their LDS destination | inside-the-image bit
  f32x4 tq = {0.f, 0.f, 0.f, 0.f};                       // (scale, shift) x 2 of the channel pair being converted
  f32x2 cx = {0.f, 0.f}, cw = {0.f, 0.f};                // conversion state of the pair: values, work
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  f16x2 chh = {};
  int ghx = 0, ghy = 0, gin = 0;                         // geometry of the request being prepared
  unsigned goff = 0;
  const float* const tab = reinterpret_cast<const float*>(smem + CH_TAB_OFF);

  for (int g = 0; g < G; ++g) {
    const int gn = min(g + 1, G - 1);  // (last group: stages its own halo again into the idle buffer -- no branch)
    const char* const hcur = halo + (g & 1) * CH_HALO_B + a_lane;
    const int nbuf = (g + 1) & 1;
    const char* const a_next = a_img + gn * 128;  // the next group's 32 channels (scalar)
    auto tap = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      [[maybe_unused]] const int kt = g * 9 + t;
      [[maybe_unused]] unsigned long long st[5] = {0, 0, 0, 0, 0};
      HALO_STAMP(st[0]);
      // Vector-memory queue of a wave, oldest first, when tap t begins:
      //   [hv(t - 2)] [tile t + 1: 2 requests, top of tap t - 1] [hv(t - 1)]        (hv(s) exists for s = 0 .. 5)
      // top of tap t: tile t + 2 (2 requests) -- always issued (past the last K tile: the last tile again, into a
      // buffer nobody reads) so that the counts below hold in every tap.
      {
        constexpr int t2 = (t + 2) % 9;
        const int g2 = t + 2 < 9 ? g : g + 1;
        dma_tile(g2 < G ? t2 * G + g2 : nk - 1, (t + 2) % 3);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- the conversion of half piece (j, h) = ((t - 2) / 2, (t - 2) % 2) (taps 2 .. 7) from hv(t - 2), and the
      // geometry of this tap's halo request (taps 0 .. 5), cut into STAGES of two to four independent instructions;
      // one stage follows each of the tap's 24 matrix instructions (cv_stage, fenced).  The phase stamps of the first
      // versions: a conversion issued as one block is a serial chain -- ~110 instructions at 11-12 cycles each, 1250
      // cycles for 4 channels (1400 for 8), next to a partner wave whose 24 matrix instructions take 1000 -- and it
      // sat on the tap's critical path wherever it was put (before, after, alternating between the SIMD's waves).
      // Here a stage's inputs were produced a matrix instruction earlier and its instructions issue in the shadow of
      // the wave's own matrix instruction.
      constexpr bool CV = t >= 2 && t <= 7;   // a half piece is converted in this tap
      constexpr bool RQ = t <= 5;             // a half piece is requested at the end of this tap
      constexpr int ch_ = CV ? (t - 2) % 2 : 0, rj = RQ ? t / 2 : 0;
      // the half piece is converted as two channel PAIRS, one behind the first twelve matrix instructions of the tap,
      // one behind the last twelve (registers: the state of one pair).  The tables of a pair are one 16-byte LDS read
      // (scale, shift, scale, shift: the prologue's layout), pair 0's issued here, pair 1's in stage 11.
      auto fetch_pair_tables = [&](int pp) {
        if constexpr (CV && PRO != 0) tq = *reinterpret_cast<const f32x4*>(tab + (gn * 32 + c8 * 8 + ch_ * 4 + pp * 2) * 2);
      };
      fetch_pair_tables(0);
      auto cv_stage = [&](auto sc_) {
        constexpr int S = decltype(sc_)::value;
        constexpr int pp = S / 12, L = S % 12;
        if constexpr (CV) {
          if constexpr (S == 0) {
            // hv(t - 2) has landed: younger than it are tile t + 1 (2), hv(t - 1) (t <= 6), tile t + 2 (2)
#ifndef T2H_HALO_DBG_NOWAIT  // (ablation switches of the debug build: timing only, wrong results)
            ch_wait_vm<(t <= 6 ? 5 : 4)>(hv[t & 1]);
#endif
          }
          if constexpr (L == 0) {
            cx = f32x2{hv[t & 1][2 * pp], hv[t & 1][2 * pp + 1]};
            ch_pin(cx);
          }
#ifndef T2H_HALO_DBG_NOMATH
          if constexpr (PRO != 0 && L == 1) {
            cx = cx * f32x2{tq[0], tq[2]} + f32x2{tq[1], tq[3]};
            ch_pin(cx);
          }
          if constexpr (PRO == 2) {
            if constexpr (L == 2) cw = cx * -1.44269504088896340736f;
            if constexpr (L == 3) cw = f32x2{__builtin_amdgcn_exp2f(cw[0]), __builtin_amdgcn_exp2f(cw[1])};
            if constexpr (L == 4) cw = cw + 1.0f;
            if constexpr (L == 5) cw = f32x2{__builtin_amdgcn_rcpf(cw[0]), __builtin_amdgcn_rcpf(cw[1])};
            if constexpr (L >= 2 && L <= 5) ch_pin(cw);
            if constexpr (L == 6) {
              cx = cx * (cw * (float)(hg[t & 1] & 1));
              ch_pin(cx);
            }
          } else {
            if constexpr (L == 6) {
              cx = cx * (float)(hg[t & 1] & 1);
              ch_pin(cx);
            }
          }
          if constexpr (L == 7) {
            amax = fmaxf(amax, fmaxf(fabsf(cx[0]), fabsf(cx[1])));
            chh = f16x2{(_Float16)cx[0], (_Float16)cx[1]};
            ch_pin(chh);
            ch_pin(amax);
          }
          if constexpr (L == 8) {
            cw = f32x2{(float)chh[0], (float)chh[1]};
            ch_pin(cw);
          }
          if constexpr (L == 9) {
            cw = (cx - cw) * T2H_SPLIT_LO_SCALE;
            ch_pin(cw);
          }
          if constexpr (L == 10) {
            const f16x2 l = {(_Float16)cw[0], (_Float16)cw[1]};
            char* const d = smem + (hg[t & 1] & ~1) + ch_ * 8 + pp * 4;  // (hg: LDS offset of the piece | inside bit)
            *reinterpret_cast<f16x2*>(d) = chh;
            *reinterpret_cast<f16x2*>(d + T2H_SPLIT_PLANE_B) = l;
          }
#else
          if constexpr (L == 10) *reinterpret_cast<f32x2*>(smem + (hg[t & 1] & ~1) + pp * 8) = cx;
#endif
          if constexpr (S == 11) fetch_pair_tables(1);
        }
        if constexpr (RQ) {
          // geometry of the piece requested at the end of this tap (piece rj of this thread = halo pixel
          // 16 wave + lane / 4 + 128 rj: the wave's first pixel is scalar arithmetic), a few instructions per stage.
          // (lane / tid laundered: all of this is loop-invariant, and hoisted out of the group loop it is carried in
          // registers the loop does not have -- the first build spilled it)
          if constexpr (S == 13) {
            const int hp0 = uwave * 16 + 128 * rj, hy0 = hp0 / CH_HW, hx0 = hp0 - hy0 * CH_HW;  // scalar
            const int hx = hx0 + (ch_launder(lane) >> 2);
            const bool wrap = hx >= CH_HW;
            ghx = wrap ? hx - CH_HW : hx;
            ghy = hy0 + (wrap ? 1 : 0);
            ch_pin(ghx);
            ch_pin(ghy);
          }
          if constexpr (S == 15) {
            const int Y = q.y0 + ghy - 1, X = q.x0 + ghx - 1;  // in the convolution's input geometry (= output geometry)
            gin = ((unsigned)Y < (unsigned)p.Hout && (unsigned)X < (unsigned)p.Wout) ? 1 : 0;
            ch_pin(gin);
          }
          if constexpr (S == 17) {
            const int Y = q.y0 + ghy - 1, X = q.x0 + ghx - 1;
            const int sy = min(max(Y, 0), p.Hout - 1) >> p.ups, sx = min(max(X, 0), p.Wout - 1) >> p.ups;
            goff = (unsigned)(sy * p.Win + sx);
            ch_pin(goff);
          }
          if constexpr (S == 19) {
            goff = (goff * (unsigned)p.lda + (ch_launder(tid) & 3) * 8) * 4u + (t % 2) * 16;
            ch_pin(goff);
          }
          if constexpr (S == 21) {
            const int c16 = (ch_launder(tid) & 3) * 16;
            int dst = 3 * CH_BD_B + nbuf * CH_HALO_B + ghy * CH_HROW + ghx * CH_ROW + c16;
            if constexpr (rj == CH_PJ - 1) {  // (only the last piece has threads beyond the halo: a select, not a branch)
              const int pi = ch_launder(tid) + CH_NT * rj;
              const int dead = CH_DMA_LOOP_B + (pi - CH_PIECES) * 16;
              dst = pi < CH_PIECES ? dst : dead;
            }
            gin = dst | gin;  // -> hg of the slot once the request is issued
            ch_pin(gin);
          }
        }
      };
      auto stages = [&](auto base_, auto n_) {  // (used where no matrix instruction runs: the very first tap)
        constexpr int base = decltype(base_)::value, n = decltype(n_)::value;
        [&]<int... I>(std::integer_sequence<int, I...>) { (cv_stage(std::integral_constant<int, base + I>{}), ...); }
        (std::make_integer_sequence<int, n>{});
      };
      // 12 matrix instructions on fragment set F, stage base + i behind the i-th
      auto mfma12s = [&](const f16x8 (&F)[8], auto base_) {
        constexpr int base = decltype(base_)::value;
        [&]<int... I>(std::integer_sequence<int, I...>) {
          ((acc[PC[I / 4]][(I % 4) / 2][I % 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                F[((I % 4) / 2) * 2 + PA[I / 4]], F[4 + (I % 2) * 2 + PB[I / 4]], acc[PC[I / 4]][(I % 4) / 2][I % 2], 0, 0, 0),
            cv_stage(std::integral_constant<int, base + I>{}), __builtin_amdgcn_sched_barrier(0)),
           ...);
        }
        (std::make_integer_sequence<int, 12>{});
      };
      HALO_STAMP(st[1]);
      const char* Ab = hcur + (t / 3) * CH_HROW + (t % 3) * CH_ROW;
      const char* Bt = btile + (t % 3) * CH_BD_B;  // (K tile 9 g + t lives in buffer (9 g + t) % 3 = t % 3)
      // (every phase fenced: left alone, the scheduler issues the twelve matrix instructions BEFORE the reads that
      // were meant to land behind them)
      read_frags(0, Ab, Bt, R0);
      __builtin_amdgcn_sched_barrier(0);
      if (t > 0 || g > 0) mfma12s(R1, std::integral_constant<int, 0>{});  // k16 step 1 of the previous tap
      else stages(std::integral_constant<int, 0>{}, std::integral_constant<int, 12>{});
      __builtin_amdgcn_sched_barrier(0);
      HALO_STAMP(st[2]);
      read_frags(1, Ab, Bt, R1);
      __builtin_amdgcn_sched_barrier(0);
      mfma12s(R0, std::integral_constant<int, 12>{});
      __builtin_amdgcn_sched_barrier(0);
      HALO_STAMP(st[3]);
      // tile t + 1 (this wave's 2 requests, top of tap t - 1) has landed: younger are hv(t - 1) (1 <= t <= 6) and tile
      // t + 2 (2).  Then the halo request of this tap: the oldest thing in flight when anything is waited for next.
      ch_wait_vm_all<(t >= 1 && t <= 6) ? 3 : 2>();
      if constexpr (RQ) {
        ch_gload16(hv[t & 1], a_next, goff);
        hg[t & 1] = gin;
      }
      HALO_STAMP(st[4]);
      __syncthreads();
#ifdef T2H_HALO_PROBE
      if (probe != nullptr && lane == 0 && (wave & 3) == 0 && blockIdx.x < 64) {
        long long* d = probe + (((int64_t)blockIdx.x * 2 + (wave >> 2)) * 40 + kt) * 8;
#pragma unroll
        for (int i = 0; i < 5; ++i) d[i] = (long long)st[i];
      }
#endif
    };
    tap(std::integral_constant<int, 0>{});
    tap(std::integral_constant<int, 1>{});
    tap(std::integral_constant<int, 2>{});
    tap(std::integral_constant<int, 3>{});
    tap(std::integral_constant<int, 4>{});
    tap(std::integral_constant<int, 5>{});
    tap(std::integral_constant<int, 6>{});
    tap(std::integral_constant<int, 7>{});
    tap(std::integral_constant<int, 8>{});
  }
  mfma12(R1);
  HALO_STAMP(pst[2]);
  ch_wait_vm_all<0>();  // the last two (redundant) weight tile requests must not land in what the epilogue keeps in LDS
  __syncthreads();
  if (amax >= 65504.0f) atomicOr(ovf, 1);
  ch_epilogue(p, q, acc, smem);
#ifdef T2H_HALO_PROBE
  HALO_STAMP(pst[3]);
  if (probe != nullptr && lane == 0 && (wave & 3) == 0 && blockIdx.x < 64) {
    long long* d = probe + (((int64_t)blockIdx.x * 2 + (wave >> 2)) * 40 + 36) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = (long long)pst[i];
  }
#endif
}

thread_local int g_halo_variant = 1;  // tuning / test hook of the calling thread
#ifdef T2H_HALO_PROBE
thread_local long long* g_halo_probe = nullptr;
#define HALO_PROBE_PASS , g_halo_probe
#else
#define HALO_PROBE_PASS
#endif

}  // namespace

extern "C" int t2h_conv_halo_force_variant(int v) {
  const int old = g_halo_variant;
  g_halo_variant = v == 0 ? 0 : 1;
  return old;
}

#ifdef T2H_HALO_PROBE
extern "C" int t2h_conv_halo_probe_next_launches(void* dev_int64_buf) {
  g_halo_probe = static_cast<long long*>(dev_int64_buf);
  return T2H_OK;
}
#endif

extern "C" int t2h_conv_halo_f32(const t2h_gemm_args* args, int32_t* overflow_flag, void* stream) {
  T2H_REQUIRE(args != nullptr, "t2h_conv_halo_f32: args is NULL");
  t2h_gemm_args a = *args;
  T2H_REQUIRE(a.A && a.B && a.C && overflow_flag, "t2h_conv_halo_f32: NULL operand / overflow word");
  T2H_REQUIRE(a.batch <= 1 && !a.b_trans && a.alpha == 1.0f && a.a_mode == 1,
              "t2h_conv_halo_f32: one 3x3 convolution in conv geometry");
  T2H_REQUIRE(a.M > 0 && a.N > 0 && a.Cin > 0 && a.Cin % 32 == 0 && a.K == 9 * a.Cin,
              "t2h_conv_halo_f32: bad shape M=%d N=%d K=%d Cin=%d", a.M, a.N, a.K, a.Cin);
  T2H_REQUIRE(a.stride == 1 && a.pad == 1 && (a.ups == 0 || a.ups == 1),
              "t2h_conv_halo_f32: 3x3 'same' / 3x3 after nearest-x2 only");
  T2H_REQUIRE(a.Hout == (a.Hin << a.ups) && a.Wout == (a.Win << a.ups) && a.M % (a.Hout * a.Wout) == 0,
              "t2h_conv_halo_f32: geometry");
  T2H_REQUIRE(a.Hout % CH_T == 0 && a.Wout % CH_T == 0,
              "t2h_conv_halo_f32: output height and width must be multiples of 16 (got %d x %d)", a.Hout, a.Wout);
  T2H_REQUIRE(a.N % 8 == 0 && a.lda % 4 == 0 && a.lda >= a.Cin && a.ldc % 4 == 0 && (!a.residual || a.ldr % 4 == 0) &&
                  t2h_aligned16(a.A) && t2h_aligned16(a.B) && t2h_aligned16(a.C) &&
                  (!a.residual || t2h_aligned16(a.residual)),
              "t2h_conv_halo_f32: N %% 8, leading dimensions %% 4, 16-byte aligned pointers");
  T2H_REQUIRE(a.epi_act == 0 && !a.res_pre, "t2h_conv_halo_f32: no epilogue activation");
  const bool pro = a.pro_scale != nullptr;
  T2H_REQUIRE(pro == (a.pro_shift != nullptr) && (!pro || (a.pro_act == 1 && a.pro_ld >= a.Cin && a.pro_ld % 4 == 0 &&
                                                           t2h_aligned16(a.pro_scale) && t2h_aligned16(a.pro_shift))),
              "t2h_conv_halo_f32: prologue = GroupNorm tables (scale and shift, 16-byte aligned rows) + swish, or none");
  const int nbx = (a.N + CH_BN - 1) / CH_BN;
  dim3 grid(nbx * (a.M / CH_BM)), block(CH_NT);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool dma = g_halo_variant != 0;
  T2H_REQUIRE(!dma || !pro || a.Cin <= 512, "t2h_conv_halo_f32: Cin <= 512 (the tables of an image live in LDS)");
  T2H_REQUIRE(!dma || ((int64_t)a.Hin * a.Win * a.lda * 4 < (int64_t(1) << 31) && (int64_t)a.N * a.K * 4 < (int64_t(1) << 31)),
              "t2h_conv_halo_f32: an image and the weights must span < 2 GiB each (32-bit byte offsets)");
  if (!dma) {
    if (pro) hipLaunchKernelGGL(conv_halo_reg_kernel<2>, grid, block, 0, s, a, overflow_flag);
    else hipLaunchKernelGGL(conv_halo_reg_kernel<0>, grid, block, 0, s, a, overflow_flag);
  } else {
    if (pro) hipLaunchKernelGGL(conv_halo_dma_kernel<2>, grid, block, 0, s, a, overflow_flag HALO_PROBE_PASS);
    else hipLaunchKernelGGL(conv_halo_dma_kernel<0>, grid, block, 0, s, a, overflow_flag HALO_PROBE_PASS);
  }
  T2H_CHECK_LAUNCH("t2h_conv_halo_f32");
  return T2H_OK;
}
