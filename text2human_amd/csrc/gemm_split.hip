// fp32-accurate GEMM on the 16-bit matrix cores ("2 x fp16 split", 3 products).
//
//   C[M,N] = epi(A[M,K] * B[N,K]^T + bias) + residual
//
// Every fp32 operand x is carried as two fp16 planes, x = h + l * 2^-11 with
// h = fp16(x), l = fp16((x - h) * 2^11): 22 significant bits, the scaling keeps the
// residual plane out of fp16's subnormal range (cf. Ootomo & Yokota's error-corrected
// tensor-core GEMM, the same idea as 3xTF32).  A product a*b is evaluated as
//   ah*bh  +  2^-11 (ah*bl + al*bh)                       (dropped: al*bl <= 2^-22)
// each an exact fp16 x fp16 product accumulated in fp32 by v_mfma_f32_32x32x16_f16,
// the 2^-11 terms in their own accumulator that is folded in once in the epilogue.
// The representation error (2^-22 per term, random sign) is an order of magnitude
// below the fp32 accumulation error of a K >= 512 dot product, so the result is as
// close to the fp64 product as the exact-fp32 kernel's (tests/test_gpu_split.py).
// Three MFMAs at 16x the fp32-MFMA rate = 5.3x the matrix throughput of
// v_mfma_f32_32x32x2_f32.
//
// Operand layout in HBM ("split rows"): [rows][K/32][2 planes][32 k] fp16, i.e. 128
// contiguous bytes (one cache line) per (row, 32-wide K tile): the producers
// (LayerNorm, GELU / attention epilogues, the weight repacker) write it directly, and
// a K tile of a row is staged global -> LDS as eight 16-byte pieces without any
// conversion.  LDS rows are padded to 144 B (9 slots of 16 B, odd) so the per-lane
// 16-byte fragment reads (ds_read_b128, 8 consecutive k of one plane) are conflict
// free.  Main loop = the same two-register-set, counted-vmcnt software pipeline as
// the fp32 kernel (gemm.hip): staged pieces are written to LDS and re-issued in the
// shadow of the MFMAs.
#include <hip/hip_ext.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef t2h_f16x8 f16x8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SP_TILE_B = T2H_SPLIT_TILE_B;  // 128 bytes per (row, K tile) in HBM
constexpr int SP_LDS_ROW = 144;              // bytes per row in LDS
constexpr int SP_PIECES = 8;                 // 16-byte pieces per (row, K tile)

// Ablation switches for tools/gemm_split_ablate.py (never defined in the product build):
// T2H_SDBG_NOGLOAD drops the global loads, _NOPUT the LDS stores, _NOFRAG the LDS
// fragment reads, _NOMMA the matrix instructions.
__device__ __forceinline__ void gload16_async(u32x4& dst, const char* ptr) {
#ifdef T2H_SDBG_NOGLOAD
  asm volatile("" : "=v"(dst) : "v"(ptr));
#else
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
#endif
}
// the same with an SGPR base and a 32-bit VGPR offset: half the address registers per issue and no
// 64-bit vector address arithmetic in the loop (measured on the 256x128 loop: -10 %)
__device__ __forceinline__ void gload16_async(u32x4& dst, unsigned voff, const char* sbase) {
#ifdef T2H_SDBG_NOGLOAD
  asm volatile("" : "=v"(dst) : "v"(voff), "s"(sbase));
#else
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
#endif
}
template <int N>
__device__ __forceinline__ void wait_vmcnt16(u32x4& v) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N));
}

// erf as x P(x^2) / Q(x^2) on [-4, 4] (the fp32 rational approximation used by Eigen /
// XLA): max abs error 4e-7 (about 3 ulp of 1.0) against 1 ulp for the library erff, at a
// third of its instruction count and without branches -- the exact-erf GELU epilogue of
// fc1 (nn.GELU(), transformer_arch.py:86) cost 10 of that launch's 52 us with erff.
__device__ __forceinline__ float erf_rational(float x) {
  x = fminf(fmaxf(x, -4.0f), 4.0f);
  const float x2 = x * x;
  float p = -2.72614225801306e-10f;
  p = fmaf(p, x2, 2.77068142495902e-08f);
  p = fmaf(p, x2, -2.10102402082508e-06f);
  p = fmaf(p, x2, -5.69250639462346e-05f);
  p = fmaf(p, x2, -7.34990630326855e-04f);
  p = fmaf(p, x2, -2.95459980854025e-03f);
  p = fmaf(p, x2, -1.60960333262415e-02f);
  float q = -1.45660718464996e-05f;
  q = fmaf(q, x2, -2.13374055278905e-04f);
  q = fmaf(q, x2, -1.68282697438203e-03f);
  q = fmaf(q, x2, -7.37332916720468e-03f);
  q = fmaf(q, x2, -1.42647390514189e-02f);
  return x * p * __builtin_amdgcn_rcpf(q);
}

__device__ __forceinline__ float gelu_erf_s(float v) {
  return 0.5f * v * (1.0f + erf_rational(v * 0.70710678118654752440f));
}

// the same on 4 values with the polynomial arithmetic on register pairs (v_pk_fma_f32)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf_2(f32x2 v) {
  f32x2 x = v * 0.70710678118654752440f;
  x[0] = fminf(fmaxf(x[0], -4.0f), 4.0f);
  x[1] = fminf(fmaxf(x[1], -4.0f), 4.0f);
  const f32x2 x2 = x * x;
  auto fma2 = [](f32x2 a, f32x2 b, float c) { return __builtin_elementwise_fma(a, b, f32x2{c, c}); };
  f32x2 p = {-2.72614225801306e-10f, -2.72614225801306e-10f};
  p = fma2(p, x2, 2.77068142495902e-08f);
  p = fma2(p, x2, -2.10102402082508e-06f);
  p = fma2(p, x2, -5.69250639462346e-05f);
  p = fma2(p, x2, -7.34990630326855e-04f);
  p = fma2(p, x2, -2.95459980854025e-03f);
  p = fma2(p, x2, -1.60960333262415e-02f);
  f32x2 q = {-1.45660718464996e-05f, -1.45660718464996e-05f};
  q = fma2(q, x2, -2.13374055278905e-04f);
  q = fma2(q, x2, -1.68282697438203e-03f);
  q = fma2(q, x2, -7.37332916720468e-03f);
  q = fma2(q, x2, -1.42647390514189e-02f);
  const f32x2 rq = {__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1])};
  const f32x2 e = x * p * rq;
  return (v * 0.5f) * (e + 1.0f);
}
__device__ __forceinline__ f32x4 gelu_erf_v(f32x4 v) {
  const f32x2 a = gelu_erf_2(f32x2{v[0], v[1]}), b = gelu_erf_2(f32x2{v[2], v[3]});
  return f32x4{a[0], a[1], b[0], b[1]};
}

// T2H_SDBG_HOTLOAD (timing tool only): every workgroup loads the operands of tile (0, 0) -- all L2 hits
#ifdef T2H_SDBG_HOTLOAD
#define LM0 0
#define LN0 0
#else
#define LM0 m0
#define LN0 n0
#endif

// Phase stamps (t2h_gemm_split_probe_next_launch; tools/gemm_phase_timing.py, bench.py): when the launch carries a
// probe buffer [workgroups][16] int64, thread 0 of every workgroup stores s_memrealtime (100 MHz, slots 0..7) and
// s_memtime (shader clock, slots 8..15) at entry (0), prologue done (1), main loop done (2), epilogue stores issued
// (3) -> phase lengths AND the clock the CU actually ran at in each phase (DVFS: the chip clocks to its power
// budget; the main loop of the B = 8 shapes runs at ~1.5 GHz, not 2.4).  probe == nullptr (every product launch):
// one scalar compare per mark.
#define G1_MARK(i)                                                                                      \
  do {                                                                                                  \
    if (probe != nullptr && threadIdx.x == 0) {                                                         \
      probe[(int64_t)blockIdx.x * 16 + (i)] = (long long)__builtin_amdgcn_s_memrealtime();              \
      probe[(int64_t)blockIdx.x * 16 + 8 + (i)] = (long long)__builtin_amdgcn_s_memtime();              \
    }                                                                                                   \
  } while (0)

// KS = 2: in-block K split.  Two wave groups of WARPS_M x WARPS_N waves each own the
// whole BM x BN tile, their own pair of LDS tile buffers and every second K tile (group
// g takes tiles 2s + g); the partial sums meet in LDS in the epilogue, group 0 first, so
// the result is deterministic.  It gives a tile that only fills the chip at one block
// per CU (N = 512 at M = 4096) two waves per SIMD without shrinking the wave tile.
//
// PP = 2 ("ping-pong", 8 waves, KS = 1): the two waves that share a SIMD (w and w + 4) run the
// SAME loop one phase apart -- [read the K tile's fragments from LDS, request a later tile by
// LDS-DMA] barrier [its matrix instructions] barrier -- so each SIMD's matrix pipe always has one
// wave in its matrix phase while the partner's LDS reads are in flight.  In the plain loop all
// eight waves leave the barrier together, read together and wait for the same fragments: its
// 256x128 main loop took 1.3 us per K step against 0.77 us of matrix instructions.  (The same
// schedule with register staging, PP = 1 of round 2, gained nothing and was removed:
// profiles/r02_gemm_pingpong_ablation_v*.log.)
//
// FMT = 1 ("x8" operands, common.h): the low planes of a K tile hold e4m3 copies of h and l instead of the fp16 l:
// [hi16 | hi8 | lo8].  The hi*hi product is unchanged; the TWO cross terms of a K tile and 32x32 block are ONE
// v_mfma_scale_f32_32x32x64_f8f6f4 (64 cycles instead of 4 x 32): along its K = 64 the A operand is [hi8 | lo8] and the
// B operand [lo8 | hi8] -- lane (row, h) supplies the 32 bytes of hi8 (h = 0) or lo8 (h = 1), lane (col, h) those of
// lo8 (h = 0) or hi8 (h = 1), so slot j of half h pairs (ah8[j], bl8[j]) resp. (al8[j], bh8[j]).  Same bytes moved,
// same fragment reads (two 16-byte reads per lane for the 8-bit operand, as for the fp16 low plane), a third fewer
// matrix-pipe cycles: fc1 29.6 -> 23.9 us, q|k|v 22.3 -> 19.4 (profiles/r05_cross_term_mx_timing.log).
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 x8_cross(const f16x8& a0, const f16x8& a1, const f16x8& b0, const f16x8& b1, const f32x16& c) {
  union { f16x8 h[2]; i32x8 v; } ua, ub;
  ua.h[0] = a0;
  ua.h[1] = a1;
  ub.h[0] = b0;
  ub.h[1] = b1;
  // formats 0 / 0 = e4m3 x e4m3; scales 0x7f = 2^0 (the tensors' power-of-two scales are folded into lo_mul)
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ua.v, ub.v, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int KS, int PP, int FMT = 0>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N * KS) void gemm_split_kernel(const t2h_gemm_split_args p, int* const ovf, long long* const probe) {
  constexpr int NT = 64 * WARPS_M * WARPS_N;  // threads per K group
  constexpr int NWG = WARPS_M * WARPS_N;      // waves per K group
  constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold a 32x32 MFMA tile");
  // Staging: the (BM + BN) * 8 16-byte pieces of a K tile (A rows first, then B rows)
  // are dealt round-robin to the NT threads, L per thread.  When NT does not divide the
  // piece count the last round wraps around: those threads re-load a piece another
  // thread also stages and store the identical bytes to the same LDS slot (benign).
  constexpr int PIECES = (BM + BN) * SP_PIECES;
  constexpr int L = (PIECES + NT - 1) / NT;
  constexpr int NMMA = 2 * 3 * TM * TN;  // MFMAs per wave and K tile
  constexpr int BUF_B = (BM + BN) * SP_LDS_ROW;  // bytes per LDS tile buffer: [A rows | B rows]
  constexpr int O_LD = WN + 4;                   // epilogue staging row (floats), odd # of 16-B slots
  constexpr int OT_LD = WM + 4;                  // transposed staging (value planes): floats per column
  constexpr int OW = WM * O_LD > WN * OT_LD ? WM * O_LD : WN * OT_LD;  // staging floats per wave
  constexpr int EPI_B = OW * 4 * NWG * KS;
  constexpr int TILE_IMG_B = (BM + BN) * SP_TILE_B;  // PP = 2: unpadded tile image (LDS-DMA), three buffers
  constexpr int MAIN_B = PP == 2 ? 3 * TILE_IMG_B : 2 * BUF_B * KS;
  constexpr int SMEM_B = MAIN_B > EPI_B ? MAIN_B : EPI_B;

  __shared__ __attribute__((aligned(16))) char smem[SMEM_B];

  G1_MARK(0);
  const int kg = KS == 1 ? 0 : (int)threadIdx.x / NT;  // K group
  const int tid = threadIdx.x - kg * NT;
  const int lane = tid & 63, wave = tid >> 6;  // wave index inside the group
  char* const gsm = smem + kg * (2 * BUF_B);   // this group's tile buffers
  const int l31 = lane & 31, hh = lane >> 5;
  const int wm0 = (wave / WARPS_N) * WM, wn0 = (wave % WARPS_N) * WN;
  // XCD-aware tile mapping (see gemm.hip)
  const int nbx = (p.N + BN - 1) / BN, nby = (p.M + BM - 1) / BM;
  // ksplit > 1 (ping-pong loop only; experiment of round 6, tools/fc2_split_k_bench.py): workgroup b + s * tiles
  // takes the s-th slice of K of tile b and writes its PARTIAL tile to C + s * M * ldc (bias / residual with slice 0)
  // (the integer division runs on the vector ALU: readfirstlane brings the workgroup-uniform result back to a scalar
  // register, which the LDS-DMA's base-address operand needs)
  const int ksid = (PP == 2 && p.ksplit > 1) ? __builtin_amdgcn_readfirstlane((int)blockIdx.x / (nbx * nby)) : 0;
  int m0, n0;
  {
    const int total = nbx * nby, b = (int)blockIdx.x - ksid * (nbx * nby);
    const int xcd = b & 7, slot = b >> 3, q = total >> 3, r = total & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    int mt = lin / nbx, nt = lin - mt * nbx;
    // One tile per CU and a wide tile grid (fc1 at M = 4096: 16 x 16 tiles, 32 per XCD): a 4-rows x 8-columns block per
    // XCD instead of 2 whole tile rows -- its L2 then holds 4 A panels + 8 W panels (4 MiB) instead of 2 + all 16
    // (5 MiB): 32 instead of 40 MiB fetched over the chip for 12.6 MiB of operands (XCD-private L2s: the floor).
    if (r == 0 && q <= 32 && nbx >= 16 && (nbx & 1) == 0 && (nby & 3) == 0 && q == (nby >> 2) * (nbx >> 1)) {
      const int hbx = nbx >> 1;
      const int sr = slot / hbx;
      mt = (xcd >> 1) * (nby >> 2) + sr;
      nt = (xcd & 1) * hbx + (slot - sr * hbx);
    }
    m0 = mt * BM;
    n0 = nt * BN;
  }
  const int nk_row = p.K / (32 * KS);  // K tiles per group over a whole row (host guarantees K % (32 KS) == 0)
  const int nk = (PP == 2 && p.ksplit > 1) ? nk_row >> (p.ksplit >> 1) : nk_row;  // ... of this workgroup's slice (ksplit 2 or 4: a shift, scalar)
  const int last = nk - 1;
  const int64_t row_b = (int64_t)nk_row * (SP_TILE_B * KS);  // bytes per split row
  // first byte of the slice inside a row (readfirstlane on the product: the form the compiler keeps in scalar registers)
  const int64_t k_base = (int64_t)__builtin_amdgcn_readfirstlane(ksid * nk) * SP_TILE_B;
  constexpr int K_STEP_B = SP_TILE_B * KS;               // bytes between a group's K tiles

  f32x16 acc[2][TM][TN];  // [0] ah*bh, [1] the 2^-11 terms ah*bl + al*bh
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;

  if constexpr (PP == 2) {
    // ---- ping-pong + LDS-DMA: global_load_lds_dwordx4 writes the K tiles straight into LDS (no
    // staging registers, no ds_write pass).  A DMA instruction fills 1 KiB = 8 rows x 128 B
    // lane-linearly, so the image is unpadded; the 16-byte fragment reads stay conflict free through
    // an XOR swizzle applied on the DMA's SOURCE address and on the read address: logical piece c of
    // row r lives at piece c ^ ((r >> 1) & 7).  Three tile buffers; group 1 runs one phase late
    // (phase 2t: group 0 reads tile t; 2t + 1: group 0 computes it, group 1 reads it; 2t + 2: group 1
    // computes it).  A wave reading tile j requests its 8-row groups of tile j + 2 into the buffer tile
    // j - 1 left (last read two / one phases ago).  Its pieces of tile j + 1 must have landed before
    // the barrier in front of phase 2j + 2: group 0 checks at the end of its matrix phase (1.5 steps
    // after the request), group 1 at the end of its read phase (1 step).
    static_assert(KS == 1 && NWG == 8 && (BM + BN) % 64 == 0, "ping-pong DMA: 8 waves, whole 8-row groups per wave");
    constexpr int NL = (BM + BN) / 64;  // DMA instructions per wave and K tile
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    auto pp_barrier = [] {
      __builtin_amdgcn_sched_barrier(0);
#ifndef T2H_SDBG_NOBAR
      __builtin_amdgcn_s_barrier();
#endif
      __builtin_amdgcn_sched_barrier(0);
    };
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    // SGPR base (advanced by one K tile per request round, scalar) + a 32-bit VGPR offset that never
    // changes: half the address registers a 64-bit VGPR address moves at every issue, no vector
    // address arithmetic in the loop (the host checks that the operands span < 2 GiB)
    unsigned goff[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int g = wave + 8 * i;  // 8-row group of the tile image
      const int r = g * 8 + (lane >> 3), pc = (lane & 7) ^ ((r >> 1) & 7);
      const bool isA = r < BM;
      const int grow = isA ? min(LM0 + r, p.M - 1) : min(LN0 + r - BM, p.N - 1);
      goff[i] = (unsigned)grow * (unsigned)row_b + pc * 16;
    }
    static_assert(BM % 64 == 0, "an 8-row group belongs to one operand");
    const unsigned wdst = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
    auto dma = [&](int kt, unsigned buf_off) {  // buf_off: byte offset of the tile buffer (scalar)
      // (no request past the last K tile: round 3 clamped kt to the last tile and re-fetched it twice per
      // workgroup -- 2 of 16 tiles at K = 512 -- into buffers nobody reads)
      if (kt > last) return;
      const int64_t k0 = (int64_t)kt * SP_TILE_B + k_base;
      const char* const baseA = reinterpret_cast<const char*>(p.A) + k0;
      const char* const baseB = reinterpret_cast<const char*>(p.B) + k0;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
#ifndef T2H_SDBG_NOGLOAD
        unsigned keep;
#ifdef T2H_SDBG_HALFDMA
        if (i & 1) continue;
#endif
        const char* const base = i < BM / 64 ? baseA : baseB;  // rows 64 i + 8 wave ..: one operand per i
        const unsigned dst_i = wdst + buf_off + i * 8192;
// nt: a CU reads each tile line once; measured -0.3 us on the first tiles and -0.2 us on the loop
// against the default policy, sc0 / sc1 no different (tools/gemm_dma_policy.sh)
#ifndef T2H_DMA_POLICY
#define T2H_DMA_POLICY " nt"
#endif
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" T2H_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(goff[i]), "s"(base), "s"(dst_i)
                     : "memory");
#endif
      }
    };
    // tile kt + 1 has landed: everything but the NL requests of tile kt + 2 -- if that tile exists
    auto wait_landed = [&](bool younger_in_flight) {
      __builtin_amdgcn_sched_barrier(0);
      if (younger_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    // fragment read offsets inside a tile image: row * 128 + ((plane * 4 + u * 2 + hh) ^ swizzle) * 16
    const int swz = (l31 >> 1) & 7;
    // (index pl * 2 + u.  FMT 1: "plane 1" = the 8-bit operand of the lane: pieces 4 + 2 h + u for A -- hi8 for
    // h = 0, lo8 for h = 1 --, 4 + 2 (1 - h) + u for B)
    int offA[4], offB[4];
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
      const int pcs = ((c2 * 2 + hh) ^ swz) * 16;
      offA[c2] = (wm0 + l31) * 128 + pcs;
      offB[c2] = (BM + wn0 + l31) * 128 + pcs;
      if (FMT == 1 && c2 >= 2) {
        offA[c2] = (wm0 + l31) * 128 + ((4 + 2 * hh + (c2 - 2)) ^ swz) * 16;
        offB[c2] = (BM + wn0 + l31) * 128 + ((4 + 2 * (1 - hh) + (c2 - 2)) ^ swz) * 16;
      }
    }
    constexpr int PA[3] = {1, 0, 0};
    constexpr int PB[3] = {0, 1, 0};
    constexpr int PC[3] = {1, 1, 0};
    dma(0, 0);
    dma(1, TILE_IMG_B);
    wait_landed(nk > 1);  // tile 0
    pp_barrier();
    G1_MARK(1);
    if (grp == 1) pp_barrier();  // phase 0 belongs to group 0
    unsigned b_cur = 0, b_nxt = TILE_IMG_B, b_free = 2 * TILE_IMG_B;  // buffers of tiles kt, kt + 1, kt + 2
    for (int kt = 0; kt < nk; ++kt) {
      const char* img = smem + b_cur;
      f16x8 af[2][TM][2], bfr[2][TN][2];
#ifdef T2H_SDBG_DMAFIRST
      dma(kt + 2, b_free);
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
#ifdef T2H_SDBG_NOFRAG
            asm volatile("" : "=v"(af[u][ti][pl]) : "v"(img));
#else
            af[u][ti][pl] = *reinterpret_cast<const f16x8*>(img + ti * 32 * 128 + offA[pl * 2 + u]);
#endif
#pragma unroll
        for (int tj = 0; tj < TN; ++tj)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
#ifdef T2H_SDBG_NOFRAG
            asm volatile("" : "=v"(bfr[u][tj][pl]) : "v"(img));
#else
            bfr[u][tj][pl] = *reinterpret_cast<const f16x8*>(img + tj * 32 * 128 + offB[pl * 2 + u]);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
#ifndef T2H_SDBG_DMAFIRST
      dma(kt + 2, b_free);
#endif
      if (grp == 1) wait_landed(kt + 2 < nk);  // tile kt + 1
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pp_barrier();
      // (round 3 tried issuing the requests for tile kt + 2 here instead, one after every fourth matrix
      // instruction: q|k|v 26.4 vs 26.9 us, fc1 30.7 vs 30.6, K = 2048 shapes 10 % slower -- not the
      // request issue cost either; removed.  profiles/r03_gemm_tile_and_dma_placement.log)
      if constexpr (FMT == 1) {
        // hi*hi of both k16 steps, then one 8-bit instruction per block for the two cross terms (interleaved so that
        // consecutive matrix instructions never share an accumulator)
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
          for (int tj = 0; tj < TN; ++tj) {
            acc[0][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][ti][0], bfr[0][tj][0], acc[0][ti][tj], 0, 0, 0);
            acc[1][ti][tj] = x8_cross(af[0][ti][1], af[1][ti][1], bfr[0][tj][1], bfr[1][tj][1], acc[1][ti][tj]);
          }
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
          for (int tj = 0; tj < TN; ++tj)
            acc[0][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][ti][0], bfr[1][tj][0], acc[0][ti][tj], 0, 0, 0);
      } else {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int ti = 0; ti < TM; ++ti)
#pragma unroll
            for (int tj = 0; tj < TN; ++tj) {
#ifdef T2H_SDBG_NOMMA
              asm volatile("" : "+v"(acc[PC[t]][ti][tj]) : "v"(af[u][ti][PA[t]]), "v"(bfr[u][tj][PB[t]]));
#else
              acc[PC[t]][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[u][ti][PA[t]], bfr[u][tj][PB[t]],
                                                                           acc[PC[t]][ti][tj], 0, 0, 0);
#endif
            }
      }
      if (grp == 0) wait_landed(kt + 2 < nk);  // tile kt + 1
      pp_barrier();
      const unsigned t = b_cur;
      b_cur = b_nxt;
      b_nxt = b_free;
      b_free = t;
    }
    if (grp == 0) pp_barrier();  // group 1's last matrix phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the tile buffers are reused by the epilogue
  } else {  // (scope: the staging registers and addresses are dead before the epilogue -- without it the
     // register allocator of ROCm 7.2 spilled 423 registers in the 256x128 instantiation)
  // ---- per-thread staging slots (fixed for the whole kernel).  Rows beyond M / N are
  // loaded from a clamped (valid) address and NOT masked: an output element depends only
  // on its own A row and B row, and rows / columns beyond the problem are never stored.
  static_assert(PIECES % NT == 0 && (BM * SP_PIECES) % NT == 0, "whole rounds of pieces, each of one operand");
  constexpr int LA = BM * SP_PIECES / NT;  // rounds i < LA stage A rows, the others B rows
  unsigned src[L];  // byte offset into the operand (the host checks it spans < 2 GiB)
  int dst[L];
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const int q = tid + NT * i;
    const int row = q / SP_PIECES, pc = q - row * SP_PIECES;  // row in [0, BM + BN)
    const bool isA = i < LA;
    const int grow = isA ? min(LM0 + row, p.M - 1) : min(LN0 + row - BM, p.N - 1);
    src[i] = (unsigned)grow * (unsigned)row_b + kg * SP_TILE_B + pc * 16;
    dst[i] = row * SP_LDS_ROW + pc * 16;
  }
  const char* const gA = reinterpret_cast<const char*>(p.A);
  const char* const gB = reinterpret_cast<const char*>(p.B);
  u32x4 rg[2][L];
#ifdef T2H_SDBG_NOPUT
  auto put = [&](int i, const u32x4& r, int buf) { asm volatile("" ::"v"(r), "v"(dst[i] + buf)); };
#else
  auto put = [&](int i, const u32x4& r, int buf) { *reinterpret_cast<u32x4*>(gsm + buf * BUF_B + dst[i]) = r; };
#endif
  auto issue = [&](auto setc, int kt) {
    constexpr int S = decltype(setc)::value;
    const int64_t k0 = (int64_t)min(kt, last) * K_STEP_B;
#pragma unroll
    for (int i = 0; i < L; ++i) gload16_async(rg[S][i], src[i], (i < LA ? gA : gB) + k0);
  };

  using set0 = std::integral_constant<int, 0>;
  using set1 = std::integral_constant<int, 1>;
  // prologue: tiles 0 and 1 are requested back to back (one memory round trip, not two);
  // tile 0 is complete once at most the L younger loads of tile 1 are outstanding
  issue(set0{}, 0);
  issue(set1{}, 1);
#pragma unroll
  for (int i = 0; i < L; ++i) {
    wait_vmcnt16<L>(rg[0][i]);
    put(i, rg[0][i], 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  issue(set0{}, 2);
  __syncthreads();
  G1_MARK(1);

  // partial products (A plane, B plane, accumulator): (l,h) (h,l) -> acc[1], (h,h) -> acc[0]
  constexpr int PA[3] = {1, 0, 0};
  constexpr int PB[3] = {0, 1, 0};
  constexpr int PC[3] = {1, 1, 0};

  constexpr int NMMA_F = FMT == 1 ? 3 * TM * TN : NMMA;  // matrix instructions per wave and K tile in this format
  auto step = [&](int kt, auto setc) {  // register set S holds tile kt+1
    constexpr int S = decltype(setc)::value;
    const int buf = kt & 1;
    const int64_t kn = (int64_t)min(kt + 3, last) * K_STEP_B;
    const char* Ab = gsm + buf * BUF_B + (wm0 + l31) * SP_LDS_ROW + hh * 16;
    const char* Bb = gsm + buf * BUF_B + (BM + wn0 + l31) * SP_LDS_ROW + hh * 16;
    int mma = 0;  // running MFMA count inside the tile (compile-time after unrolling)
    // staged pieces pinned behind the matrix instructions of the second half of the tile
    auto pinned = [&]() {
      ++mma;
#pragma unroll
      for (int q = 0; q < L; ++q) {
        // (more pieces than matrix instructions -- 64x64 tiles on x8 operands: 4 pieces, 3 instructions -- share slots)
        const int at0 = (2 * L <= NMMA_F) ? NMMA_F - 2 * (L - q) + 1 : (NMMA_F * (q + 1)) / L;
        const int at = at0 < 1 ? 1 : at0;
        if (at != mma) continue;
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt16<2 * L - 1>(rg[S][q]);
        put(q, rg[S][q], buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        gload16_async(rg[S][q], src[q], (q < LA ? gA : gB) + kn);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if constexpr (FMT == 1) {
      // x8 operands: hi16 pieces 2 u + h, 8-bit operand of the lane = pieces 4 + 2 h + j (A: hi8 | lo8) resp.
      // 4 + 2 (1 - h) + j (B: lo8 | hi8)
      f16x8 ah[2][TM], a8[2][TM], bh[2][TN], b8[2][TN];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int ti = 0; ti < TM; ++ti) {
          ah[j][ti] = *reinterpret_cast<const f16x8*>(Ab + ti * 32 * SP_LDS_ROW + j * 32);
          a8[j][ti] = *reinterpret_cast<const f16x8*>(Ab + ti * 32 * SP_LDS_ROW + 64 + hh * 16 + j * 16);
        }
#pragma unroll
        for (int tj = 0; tj < TN; ++tj) {
          bh[j][tj] = *reinterpret_cast<const f16x8*>(Bb + tj * 32 * SP_LDS_ROW + j * 32);
          b8[j][tj] = *reinterpret_cast<const f16x8*>(Bb + tj * 32 * SP_LDS_ROW + 96 - 48 * hh + j * 16);  // (Bb holds + 16 h)
        }
      }
#pragma unroll
      for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < TN; ++tj) {
          acc[0][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][ti], bh[0][tj], acc[0][ti][tj], 0, 0, 0);
          pinned();
          acc[1][ti][tj] = x8_cross(a8[0][ti], a8[1][ti], b8[0][tj], b8[1][tj], acc[1][ti][tj]);
          pinned();
        }
#pragma unroll
      for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < TN; ++tj) {
          acc[0][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][ti], bh[1][tj], acc[0][ti][tj], 0, 0, 0);
          pinned();
        }
    } else {
#pragma unroll
    for (int u = 0; u < 2; ++u) {  // two k16 steps per K tile
      f16x8 af[TM][2], bfr[TN][2];
#pragma unroll
      for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#ifdef T2H_SDBG_NOFRAG
          asm volatile("" : "=v"(af[ti][pl]) : "v"(Ab));
#else
          af[ti][pl] = *reinterpret_cast<const f16x8*>(Ab + ti * 32 * SP_LDS_ROW + pl * 64 + u * 32);
#endif
#pragma unroll
      for (int tj = 0; tj < TN; ++tj)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#ifdef T2H_SDBG_NOFRAG
          asm volatile("" : "=v"(bfr[tj][pl]) : "v"(Bb));
#else
          bfr[tj][pl] = *reinterpret_cast<const f16x8*>(Bb + tj * 32 * SP_LDS_ROW + pl * 64 + u * 32);
#endif
#pragma unroll
      for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
          for (int tj = 0; tj < TN; ++tj) {
#ifdef T2H_SDBG_NOMMA
            asm volatile("" : "+v"(acc[PC[t]][ti][tj]) : "v"(af[ti][PA[t]]), "v"(bfr[tj][PB[t]]));
#else
            acc[PC[t]][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ti][PA[t]], bfr[tj][PB[t]],
                                                                         acc[PC[t]][ti][tj], 0, 0, 0);
#endif
            pinned();
          }
      }
    }
    }
    __syncthreads();
  };
  // (no third, peeled copy of `step` for an odd tile count: the register allocator spilled 423
  // registers in it, and a kernel that needs scratch pays for it at every launch -- 45 instead of
  // 33 us on the q|k|v / fc1 shapes -- whether or not the spilling path runs)
  for (int kt = 0; kt < nk; kt += 2) {
    step(kt, set1{});
    if (kt + 1 < nk) step(kt + 1, set0{});
  }
#pragma unroll
  for (int S = 0; S < 2; ++S)
#pragma unroll
    for (int i = 0; i < L; ++i) wait_vmcnt16<0>(rg[S][i]);
  }
  G1_MARK(2);
  // ---- epilogue.  The accumulators (C/D layout: col = lane&31, row = (r&3) +
  // 8*(r>>2) + 4*(lane>>5)) are transposed through the (now idle) LDS so that every
  // lane owns 4 CONSECUTIVE columns of a row: residual loads and fp32 stores become
  // 16-byte accesses and a split-row store is three 8-byte writes instead of twelve
  // 2-byte ones.
  float* const Ot = reinterpret_cast<float*>(smem) + wave * OW;  // group 0's staged wave tile
  float* const Og = Ot + kg * (NWG * OW);                        // this group's
  // value of accumulator register r of MFMA tile (ti, tj) as it is staged: both partial
  // accumulators folded together, plus the bias (K group 0 only)
  const float lo_mul = FMT == 1 ? p.lo_mul : T2H_SPLIT_LO_INV;  // (x8: 2^-11 over the operands' 8-bit plane scales)
  auto fin = [&](int ti, int tj, int r, float bv) {
    return fmaf(acc[1][ti][tj][r], lo_mul, acc[0][ti][tj][r]) + bv;
  };
  // A tile whose columns straddle vt_col0 (BN = 192 does not divide the q|k / v boundary at 1024) takes
  // both forms of the epilogue, each skipping the other side's columns; tiles of the other shapes lie on
  // one side (the host checks vt_col0 % BN == 0) and keep the single pass.
  constexpr bool MIXED = BN % 128 != 0;
  const bool all_v = p.Vt != nullptr && n0 >= p.vt_col0;
  const bool some_v = MIXED ? (p.Vt != nullptr && n0 + BN > p.vt_col0) : all_v;
  if (some_v) {
    // Value heads of the q|k|v projection -> transposed planes Vt[B][H][2][hd][T].  The wave
    // tile is staged TRANSPOSED ([column][row]: a lane's 4 consecutive accumulator registers
    // are 4 consecutive rows = one 16-byte LDS write), then every lane takes one column and
    // the 8 keys {16j + 4h + 0..3, 16j + 8 + 4h + 0..3} that the attention kernel contracts
    // in one k16-step: they sit at the 8 consecutive positions 16j + 8h .. + 7 of the plane,
    // so the store is one 16-byte piece per plane and contiguous across lanes.
#pragma unroll
    for (int ti = 0; ti < TM; ++ti) {
#pragma unroll
      for (int tj = 0; tj < TN; ++tj) {
        const int col = n0 + wn0 + tj * 32 + l31;
        const float bv = (p.bias && col < p.N && kg == 0) ? p.bias[col] : 0.f;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f32x4 w4;
#pragma unroll
          for (int e = 0; e < 4; ++e) w4[e] = fin(ti, tj, 4 * g4 + e, bv);
          *reinterpret_cast<f32x4*>(Og + (tj * 32 + l31) * OT_LD + ti * 32 + 8 * g4 + 4 * hh) = w4;
        }
      }
    }
    __syncthreads();
    constexpr int RG = WM / 8;                   // 8-key chunks per staged column
    constexpr int NCV = RG * WN / 64 / KS;       // chunks per lane
    static_assert(NCV >= 1 && NCV * 64 * KS == RG * WN, "value-plane chunking");
    const int n_vh = (p.N - p.vt_col0) / p.vt_hd;
    const int row0 = m0 + wm0;                   // first row of the wave tile (multiple of 32)
    const int b = row0 / p.vt_T, key0 = row0 - b * p.vt_T;
#pragma unroll
    for (int it = 0; it < NCV; ++it) {
      const int c = lane + 64 * (it + kg * NCV);
      const int cl = c / RG, u = c - cl * RG;
      const int col = n0 + wn0 + cl;
      const int ra = 16 * (u >> 1) + 4 * (u & 1);  // rows ra..ra+3 and ra+8..ra+11
      if (row0 + ra >= p.M || col >= p.N || (MIXED && col < p.vt_col0)) continue;
      f32x4 va = *reinterpret_cast<const f32x4*>(Ot + cl * OT_LD + ra);
      f32x4 vb = *reinterpret_cast<const f32x4*>(Ot + cl * OT_LD + ra + 8);
      if (KS == 2) {
        va += *reinterpret_cast<const f32x4*>(Ot + NWG * OW + cl * OT_LD + ra);
        vb += *reinterpret_cast<const f32x4*>(Ot + NWG * OW + cl * OT_LD + ra + 8);
      }
      t2h_split_guard8(ovf, va, vb);
      t2h_f16x8 vh, vl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        _Float16 x0, x1;
        t2h_split2(va[e], x0, x1);
        vh[e] = x0;
        vl[e] = x1;
        t2h_split2(vb[e], x0, x1);
        vh[4 + e] = x0;
        vl[4 + e] = x1;
      }
      const int cc = col - p.vt_col0, head = cc / p.vt_hd, d = cc - head * p.vt_hd;
      uint16_t* dstp = p.Vt + ((((int64_t)b * n_vh + head) * 2) * p.vt_hd + d) * p.vt_T + key0 + 8 * u;
      t2h_store16_wt(dstp, vh);
      t2h_store16_wt(dstp + (int64_t)p.vt_hd * p.vt_T, vl);
    }
    if (!MIXED || all_v) return;
    __syncthreads();  // the staging area is rewritten row-major below
  }
#pragma unroll
  for (int ti = 0; ti < TM; ++ti) {
#pragma unroll
    for (int tj = 0; tj < TN; ++tj) {
      const int col = n0 + wn0 + tj * 32 + l31;
      const float bv = (p.bias && col < p.N && kg == 0 && ksid == 0) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Og[(ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * O_LD + tj * 32 + l31] = fin(ti, tj, r, bv);
    }
  }
  __syncthreads();
  constexpr int CPR = WN / 8;              // 8-column chunks per staged row
  constexpr int NCH = WM * CPR / 64 / KS;  // chunks per lane (the K groups share the stores)
  static_assert(NCH >= 1 && NCH * 64 * KS == WM * CPR, "epilogue chunking");
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int c = lane + 64 * (it + kg * NCH);
    const int rl = c / CPR, c8 = (c - rl * CPR) * 8;
    const int row = m0 + wm0 + rl, col = n0 + wn0 + c8;
    const bool valid = row < p.M && col < p.N && !(MIXED && some_v && col >= p.vt_col0);
    if (!valid) continue;
    f32x4 va = *reinterpret_cast<const f32x4*>(Ot + rl * O_LD + c8);
    f32x4 vb = *reinterpret_cast<const f32x4*>(Ot + rl * O_LD + c8 + 4);
    if (KS == 2) {
      va += *reinterpret_cast<const f32x4*>(Ot + NWG * OW + rl * O_LD + c8);
      vb += *reinterpret_cast<const f32x4*>(Ot + NWG * OW + rl * O_LD + c8 + 4);
    }
    if (p.epi_act == 1) {
      va = gelu_erf_v(va);
      vb = gelu_erf_v(vb);
    } else if (p.epi_act == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        va[e] = fmaxf(va[e], 0.f);
        vb[e] = fmaxf(vb[e], 0.f);
      }
    }
    if (p.residual && ksid == 0) {
      va += *reinterpret_cast<const f32x4*>(p.residual + (int64_t)row * p.ldr + col);
      vb += *reinterpret_cast<const f32x4*>(p.residual + (int64_t)row * p.ldr + col + 4);
    }
    if (p.C) {
      float* const Cs = p.C + (int64_t)ksid * p.M * p.ldc;  // (ksplit: slice s's partial tile)
      *reinterpret_cast<f32x4*>(Cs + (int64_t)row * p.ldc + col) = va;
      *reinterpret_cast<f32x4*>(Cs + (int64_t)row * p.ldc + col + 4) = vb;
    }
    if (p.C_split) {
      if (p.out_fmt == 1) t2h_store_x8_8<1>(p.C_split, row, p.N, col, va, vb, p.out_scale, ovf);  // (lanes 2j, 2j + 1: one 16-column group)
      else t2h_store_split8(p.C_split, row, p.N, col, va, vb, ovf);
    }
  }
  G1_MARK(3);
#ifdef T2H_GEMM_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G1_MARK(4);
#endif
}

// ---- a handful of rows (the last layer's tail on the changed rows of a sampling step, M ~ 16):
// the tiled kernel gives such a problem N / 64 workgroups that walk K alone (fc2: 8 workgroups x 64
// K steps, 20 us).  Here a workgroup owns a 16 x 16 output tile, its 8 waves split K between them
// (tiles w, w + 8, ..) and feed v_mfma_f32_16x16x32_f16 STRAIGHT from global memory -- a lane's
// fragment of a split row is one 16-byte piece -- and the partial sums meet in LDS, in wave order.
constexpr int SKINNY_WAVES = 8;
// FMT = 1 (x8 operands): v_mfma_scale_f32_16x16x128_f8f6f4 contracts 128 slots = 4 lane groups x 32 bytes: TWO K tiles
// per instruction -- lane group g supplies, for A, hi8 (g = 0) / lo8 (g = 1) of tile t and hi8 (2) / lo8 (3) of tile
// t + 1, for B lo8 / hi8 / lo8 / hi8 -- next to one v_mfma_f32_16x16x32_f16 per tile for hi*hi.  A wave walks the tile
// PAIRS w, w + 8, ..; a last unpaired tile (K / 32 odd) contributes zeros for its absent partner.
typedef int i32x8s __attribute__((ext_vector_type(8)));
template <int FMT>
__global__ __launch_bounds__(64 * SKINNY_WAVES) void gemm_split_skinny_kernel(const t2h_gemm_split_args p, int* const ovf) {
  __shared__ __attribute__((aligned(16))) float red[SKINNY_WAVES][16 * 16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, kg = lane >> 4;  // A row / B row (= output column) and 8-wide k group of the lane
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int nk = p.K / 32;
  const int64_t row_b = (int64_t)nk * SP_TILE_B;
  const char* ap = reinterpret_cast<const char*>(p.A) + (int64_t)min(m0 + r, p.M - 1) * row_b + kg * 16;
  const char* bp = reinterpret_cast<const char*>(p.B) + (int64_t)min(n0 + r, p.N - 1) * row_b + kg * 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_lo = acc;
  if constexpr (FMT == 1) {
    // 8-bit operand of the lane inside a tile: A group g -> plane (g & 1) (hi8, lo8), B -> plane 1 - (g & 1); the
    // tile of the pair: g >> 1
    const char* a8 = reinterpret_cast<const char*>(p.A) + (int64_t)min(m0 + r, p.M - 1) * row_b + 64 + (kg & 1) * 32;
    const char* b8 = reinterpret_cast<const char*>(p.B) + (int64_t)min(n0 + r, p.N - 1) * row_b + 64 + (1 - (kg & 1)) * 32;
    const int npair = (nk + 1) / 2;
#pragma unroll 2
    for (int t2 = wave; t2 < npair; t2 += SKINNY_WAVES) {
      const int t0 = 2 * t2, tg = t0 + (kg >> 1);
      const bool have = tg < nk;
      union { f16x8 h[2]; i32x8s v; } ua, ub;
      const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
      ua.h[0] = have ? *reinterpret_cast<const f16x8*>(a8 + (int64_t)tg * SP_TILE_B) : zero;
      ua.h[1] = have ? *reinterpret_cast<const f16x8*>(a8 + (int64_t)tg * SP_TILE_B + 16) : zero;
      ub.h[0] = have ? *reinterpret_cast<const f16x8*>(b8 + (int64_t)tg * SP_TILE_B) : zero;
      ub.h[1] = have ? *reinterpret_cast<const f16x8*>(b8 + (int64_t)tg * SP_TILE_B + 16) : zero;
      acc_lo = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ua.v, ub.v, acc_lo, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(ap + (int64_t)t0 * SP_TILE_B),
                                                   *reinterpret_cast<const f16x8*>(bp + (int64_t)t0 * SP_TILE_B), acc, 0, 0, 0);
      if (t0 + 1 < nk)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(ap + (int64_t)(t0 + 1) * SP_TILE_B),
                                                     *reinterpret_cast<const f16x8*>(bp + (int64_t)(t0 + 1) * SP_TILE_B), acc, 0, 0, 0);
    }
  } else {
#pragma unroll 4
  for (int t = wave; t < nk; t += SKINNY_WAVES) {
    const f16x8 ah = *reinterpret_cast<const f16x8*>(ap + (int64_t)t * SP_TILE_B);
    const f16x8 al = *reinterpret_cast<const f16x8*>(ap + (int64_t)t * SP_TILE_B + 64);
    const f16x8 bh = *reinterpret_cast<const f16x8*>(bp + (int64_t)t * SP_TILE_B);
    const f16x8 bl = *reinterpret_cast<const f16x8*>(bp + (int64_t)t * SP_TILE_B + 64);
    acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc_lo, 0, 0, 0);
    acc_lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc_lo, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
  }
  }
  const float lo_mul = FMT == 1 ? p.lo_mul : T2H_SPLIT_LO_INV;
  // accumulator register e of lane l: row 4 (l >> 4) + e, column l & 15
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wave][(4 * kg + e) * 16 + r] = fmaf(acc_lo[e], lo_mul, acc[e]);
  __syncthreads();
  if (wave != 0) return;
  const int rl = lane >> 2, c4 = (lane & 3) * 4;
  const int row = m0 + rl, col = n0 + c4;
  if (row >= p.M || col >= p.N) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][rl * 16 + c4]);
#pragma unroll
  for (int w = 1; w < SKINNY_WAVES; ++w) v += *reinterpret_cast<const f32x4*>(&red[w][rl * 16 + c4]);
  if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + col);
  if (p.epi_act == 1) v = gelu_erf_v(v);
  else if (p.epi_act == 2)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + (int64_t)row * p.ldr + col);
  if (p.C) *reinterpret_cast<f32x4*>(p.C + (int64_t)row * p.ldc + col) = v;
  if (p.C_split) {
    if (p.out_fmt == 1) t2h_store_x8_4(p.C_split, row, p.N, col, v, p.out_scale, ovf);
    else t2h_store_split4(p.C_split, row, p.N, col, v, ovf);
  }
}

// fp32 [rows, C] (ld) -> split rows; one thread per 4 consecutive columns
__global__ void split_rows_kernel(const float* __restrict__ x, int ldx, uint16_t* __restrict__ out, int64_t total,
                                  int C, int* ovf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over rows * C/4
  if (i >= total) return;
  const int q = C >> 2;
  const int64_t row = i / q;
  const int c0 = (int)(i - row * q) * 4;
  t2h_store_split4(out, row, C, c0, *reinterpret_cast<const f32x4*>(x + row * ldx + c0), ovf);
}

// fp32 [rows, C] (ld) -> x8 rows (common.h); one thread per 4 consecutive columns
__global__ void split_rows_x8_kernel(const float* __restrict__ x, int ldx, uint16_t* __restrict__ out, int64_t total,
                                     int C, float scale, int* ovf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int q = C >> 2;
  const int64_t row = i / q;
  const int c0 = (int)(i - row * q) * 4;
  t2h_store_x8_4(out, row, C, c0, *reinterpret_cast<const f32x4*>(x + row * ldx + c0), scale, ovf);
}

// t2h_gemm_split_time_next_launch: events that receive the START and the END of the next kernel this
// thread launches (hipExtLaunchKernelGGL: timestamps of the kernel itself, what rocprofv3's kernel
// trace reports) -- events recorded around a launch on the stream measure from the end of the
// PREVIOUS kernel and so include the dependent-launch boundary
thread_local hipEvent_t g_time_start = nullptr, g_time_stop = nullptr;
thread_local long long* g_probe = nullptr;  // t2h_gemm_split_probe_next_launch

template <typename K, typename... Args>
void launch_maybe_timed(K kernel, dim3 grid, dim3 block, hipStream_t s, Args... args) {
  if (g_time_start && g_time_stop) {
    hipExtLaunchKernelGGL(kernel, grid, block, 0, s, g_time_start, g_time_stop, 0, args...);
    g_time_start = g_time_stop = nullptr;
  } else {
    hipLaunchKernelGGL(kernel, grid, block, 0, s, args...);
  }
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int KS = 1, int PP = 0, int FMT = 0>
int launch_split(const t2h_gemm_split_args& a, hipStream_t s) {
  T2H_REQUIRE(a.K % (32 * KS) == 0, "t2h_gemm_split_f32: this tile config needs K %% %d == 0", 32 * KS);
  if (a.Vt)
    T2H_REQUIRE(a.vt_col0 % (BN % 128 ? 32 : BN) == 0, "t2h_gemm_split_f32: vt_col0=%d must be a multiple of %d for this tile",
                a.vt_col0, BN % 128 ? 32 : BN);
  T2H_REQUIRE((int64_t)(a.M > a.N ? a.M : a.N) * a.K * 4 < (int64_t(1) << 31),
              "t2h_gemm_split_f32: operands are addressed with 32-bit byte offsets (each must span < 2 GiB)");
  const int ks = a.ksplit > 1 ? a.ksplit : 1;
  if (ks > 1)
    T2H_REQUIRE(PP == 2 && a.K % (32 * ks) == 0 && a.C && !a.C_split && !a.Vt && a.epi_act == 0,
                "t2h_gemm_split_f32: ksplit=%d needs a ping-pong tile configuration (8, 10, 11), K %% %d == 0, an fp32 output "
                "of ksplit * M rows and no activation / split-row / Vt output", ks, 32 * ks);
  dim3 grid(((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM) * ks);
  int* ovf = a.overflow_flag;
  launch_maybe_timed(gemm_split_kernel<BM, BN, WARPS_M, WARPS_N, KS, PP, FMT>, grid, dim3(64 * WARPS_M * WARPS_N * KS), s, a,
                     ovf, g_probe);
  T2H_CHECK_LAUNCH("t2h_gemm_split_f32");
  return T2H_OK;
}

thread_local int g_force_split_cfg = -1;  // tuning / test hook of the calling thread

}  // namespace

extern "C" int t2h_gemm_split_probe_next_launch(void* dev_int64_buf) {
  g_probe = static_cast<long long*>(dev_int64_buf);
  return T2H_OK;
}

extern "C" int t2h_gemm_split_time_next_launch(void* start_event, void* stop_event) {
  g_time_start = static_cast<hipEvent_t>(start_event);
  g_time_stop = static_cast<hipEvent_t>(stop_event);
  return T2H_OK;
}

extern "C" int t2h_gemm_split_force_config(int cfg) {
  const int old = g_force_split_cfg;
  g_force_split_cfg = cfg;
  return old;
}

// tile configuration of the dispatcher for a (validated) problem
static int pick_split_cfg(const t2h_gemm_split_args& a) {
  int cfg = g_force_split_cfg;
  if (cfg < 0) {
    // measured on MI355X at M = 4096 (tools/gemm_split_bench.py, profiles/): the 4-wave
    // 128x64 tile wins wherever it gives every CU at least two tiles; when it gives at
    // most one (N = 512) the same tile with the in-block K split (two waves per SIMD)
    // is 5-8 % faster; 128x128 only pays once it still fills 2 x 256 CUs
    const int64_t tiles64 = (int64_t)((a.M + 127) / 128) * ((a.N + 63) / 64);
    const int64_t tiles128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128);
    const int64_t tiles256 = (int64_t)((a.M + 127) / 128) * ((a.N + 255) / 256);
    const int64_t tiles_big = (a.M % 256 == 0 && a.N % 128 == 0) ? (int64_t)(a.M / 256) * (a.N / 128) : 0;
    const int64_t cus = 256;
    if (a.M <= 64) cfg = 2;
    else if (tiles_big >= cus * 3 / 4) {
      // 256x128 tiles wherever they give (nearly) every CU one: q|k|v / fc1 at M = 4096 (192 / 256
      // tiles) and every sampler Linear but proj at M = 16384 (q|k|v 99 vs 104 / 126 us for the 128x64 /
      // 128x128 tiles, fc2 93 vs 111), on the ping-pong LDS-DMA loop (5-6 % faster than the same tile
      // on the register-staged loop, the twin round 2 kept for the A/B and round 3 removed)
      cfg = 8;
      // 128x192 tiles (the same loop, wave tile 32x96) where they need fewer tile-rounds of the chip's 256
      // CUs: q|k|v at M = 4096 is 192 tiles of 256x128 -- a quarter of the CUs idle for the whole launch --
      // or 256 tiles of 128x192, one per CU with 3/4 of the work each
      if (a.M % 128 == 0 && a.N % 192 == 0 && (a.Vt == nullptr || a.vt_col0 % 32 == 0)) {
        const int64_t t192 = (int64_t)(a.M / 128) * (a.N / 192);
        if (((t192 + cus - 1) / cus) * (128 * 192) < ((tiles_big + cus - 1) / cus) * (256 * 128)) cfg = 10;
      }
    } else if (tiles128 >= 4 * cus) cfg = 1;
    else if (tiles64 <= cus && a.K % 64 == 0 && a.K >= 256) cfg = 6;
    else cfg = 0;
  }
  const bool skinny_ok = a.N % 16 == 0 && !a.Vt && (a.bias == nullptr || t2h_aligned16(a.bias));
  if (g_force_split_cfg < 0 && a.M <= 64 && skinny_ok) cfg = 9;
  return cfg;
}

extern "C" int t2h_gemm_split_tile_config(const t2h_gemm_split_args* args) {
  T2H_REQUIRE(args != nullptr && args->M > 0 && args->N > 0 && args->K > 0, "t2h_gemm_split_tile_config: bad arguments");
  return pick_split_cfg(*args);
}

extern "C" int t2h_gemm_split_f32(const t2h_gemm_split_args* args, void* stream) {
  // the timing hook is for THIS call: a call that fails a check before it launches must not leave the events
  // armed for a later, unrelated launch
  struct Disarm {
    ~Disarm() {
      g_time_start = g_time_stop = nullptr;
      g_probe = nullptr;
    }
  } disarm_on_exit;
  T2H_REQUIRE(args != nullptr, "t2h_gemm_split_f32: args is NULL");
  const t2h_gemm_split_args a = *args;
  T2H_REQUIRE(a.A && a.B && (a.C || a.C_split || a.Vt), "t2h_gemm_split_f32: NULL operand");
  T2H_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 32 == 0, "t2h_gemm_split_f32: bad shape M=%d N=%d K=%d",
              a.M, a.N, a.K);
  T2H_REQUIRE(t2h_aligned16(a.A) && t2h_aligned16(a.B), "t2h_gemm_split_f32: operands must be 16-byte aligned");
  T2H_REQUIRE(a.N % 8 == 0 && (!a.C || (a.ldc % 4 == 0 && t2h_aligned16(a.C))) &&
                  (!a.residual || (a.ldr % 4 == 0 && t2h_aligned16(a.residual))),
              "t2h_gemm_split_f32: N must be a multiple of 8, ldc / ldr of 4, C / residual 16-byte aligned");
  if (a.C_split) T2H_REQUIRE(a.N % 32 == 0, "t2h_gemm_split_f32: split output needs N %% 32 == 0");
  T2H_REQUIRE(a.overflow_flag != nullptr || (!a.C_split && !a.Vt),
              "t2h_gemm_split_f32: overflow_flag is NULL (needed with C_split / Vt outputs)");
  if (a.Vt)
    T2H_REQUIRE(a.vt_hd > 0 && a.vt_T > 0 && a.vt_T % 128 == 0 && a.M % a.vt_T == 0 && a.vt_col0 >= 0 &&
                    a.vt_col0 < a.N && (a.N - a.vt_col0) % a.vt_hd == 0 && a.epi_act == 0 && !a.residual,
                "t2h_gemm_split_f32: bad Vt routing (col0=%d T=%d hd=%d M=%d N=%d)", a.vt_col0, a.vt_T, a.vt_hd,
                a.M, a.N);
  T2H_REQUIRE((a.fmt == 0 || a.fmt == 1) && (a.out_fmt == 0 || a.out_fmt == 1), "t2h_gemm_split_f32: fmt / out_fmt must be 0 or 1");
  T2H_REQUIRE(a.ksplit >= 0 && a.ksplit <= 4 && a.ksplit != 3, "t2h_gemm_split_f32: ksplit must be 0 / 1 (off), 2 or 4");
  if (a.ksplit > 1)
    T2H_REQUIRE(a.fmt == 1 && (pick_split_cfg(a) == 8 || pick_split_cfg(a) == 10 || pick_split_cfg(a) == 11),
                "t2h_gemm_split_f32: ksplit needs x8 operands on a ping-pong tile configuration (force 8, 10 or 11)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  t2h_gemm_split_args ax = a;
  if (ax.lo_mul == 0.f) ax.lo_mul = T2H_SPLIT_LO_INV;
  if (ax.out_scale == 0.f) ax.out_scale = 1.0f;
  const int cfg = pick_split_cfg(a);
  if (cfg == 9) {
    const bool skinny_ok = a.N % 16 == 0 && !a.Vt && (a.bias == nullptr || t2h_aligned16(a.bias));
    T2H_REQUIRE(skinny_ok, "t2h_gemm_split_f32: the few-rows kernel needs N %% 16 == 0 and no Vt routing");
    int* ovf = a.overflow_flag;
    if (a.fmt == 1)
      launch_maybe_timed(gemm_split_skinny_kernel<1>, dim3(a.N / 16, (a.M + 15) / 16), dim3(64 * SKINNY_WAVES), s, ax, ovf);
    else
      launch_maybe_timed(gemm_split_skinny_kernel<0>, dim3(a.N / 16, (a.M + 15) / 16), dim3(64 * SKINNY_WAVES), s, ax, ovf);
    T2H_CHECK_LAUNCH("t2h_gemm_split_f32");
    return T2H_OK;
  }
  if (a.fmt == 1) {  // x8 operands: the tile configurations the sampler's shapes take
    switch (cfg) {
      case 2: return launch_split<64, 64, 2, 2, 1, 0, 1>(ax, s);
      case 6: return launch_split<128, 64, 2, 2, 2, 0, 1>(ax, s);
      case 8: return launch_split<256, 128, 4, 2, 1, 2, 1>(ax, s);
      case 10: return launch_split<128, 192, 4, 2, 1, 2, 1>(ax, s);
      case 11: return launch_split<128, 128, 4, 2, 1, 2, 1>(ax, s);  // (round-6 split-K experiment; never picked automatically)
      case 0: return launch_split<128, 64, 2, 2, 1, 0, 1>(ax, s);
      default:
        t2h_set_error("t2h_gemm_split_f32: tile configuration %d is not built for x8 operands (0, 2, 6, 8, 9, 10, 11)", cfg);
        return T2H_ERR_INVALID;
    }
  }
  switch (cfg) {
    case 1: return launch_split<128, 128, 4, 2>(ax, s);  // 8 waves, wave tile 32x64
    case 2: return launch_split<64, 64, 2, 2>(ax, s);    // 4 waves, wave tile 32x32
    case 3: return launch_split<128, 64, 4, 2>(ax, s);   // 8 waves, wave tile 32x32
    case 5: return launch_split<128, 256, 4, 2>(ax, s);  // 8 waves, wave tile 32x128
    case 6: return launch_split<128, 64, 2, 2, 2>(ax, s);  // 2 K groups x 4 waves, wave tile 64x32
    case 8: return launch_split<256, 128, 4, 2, 1, 2>(ax, s);  // 8 waves, wave tile 64x64, ping-pong LDS-DMA loop
    case 10: return launch_split<128, 192, 4, 2, 1, 2>(ax, s);  // 8 waves, wave tile 32x96, the same loop
    default: return launch_split<128, 64, 2, 2>(ax, s);  // 4 waves, wave tile 64x32
  }
}

extern "C" int t2h_split_rows_x8_f32(const float* x, int32_t ldx, uint16_t* out, int64_t rows, int32_t C, float scale,
                                     int32_t* overflow_flag, void* stream) {
  T2H_REQUIRE(x && out && rows > 0 && C > 0 && C % 32 == 0 && ldx % 4 == 0 && scale > 0.f, "t2h_split_rows_x8_f32: bad arguments");
  T2H_REQUIRE(overflow_flag != nullptr, "t2h_split_rows_x8_f32: overflow_flag is NULL");
  const int64_t total = rows * (C / 4);
  hipLaunchKernelGGL(split_rows_x8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, ldx, out, total, C, scale, overflow_flag);
  T2H_CHECK_LAUNCH("t2h_split_rows_x8_f32");
  return T2H_OK;
}

extern "C" int t2h_split_rows_f32(const float* x, int32_t ldx, uint16_t* out, int64_t rows, int32_t C,
                                  int32_t* overflow_flag, void* stream) {
  T2H_REQUIRE(x && out && rows > 0 && C > 0 && C % 32 == 0 && ldx % 4 == 0, "t2h_split_rows_f32: bad arguments");
  const int64_t total = rows * (C / 4);
  int* ovf = overflow_flag;
  T2H_REQUIRE(ovf != nullptr, "t2h_split_rows_f32: overflow_flag is NULL");
  hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, ldx, out, total, C, ovf);
  T2H_CHECK_LAUNCH("t2h_split_rows_f32");
  return T2H_OK;
}
