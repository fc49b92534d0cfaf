// fp32 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   C[M,N] = epi(alpha * A'[M,K] * B[N,K]^T + bias) + residual
//
// One kernel family serves every dense contraction of the Text2Human sampling
// path (see include/t2h_hip.h).  Design (MI355X-first):
//   * v_mfma_f32_32x32x2_f32: exact-fp32 fma chains at the 157 TFLOP/s matrix
//     rate; a 256-thread workgroup = 4 waves (one per SIMD), each wave owns a
//     (BM/WARPS_M) x (BN/WARPS_N) tile of 32x32 accumulators.
//   * K is walked in tiles of 32.  The MFMA k-pair of step s is (s, 16+s)
//     inside the tile -- any fixed k permutation is legal as long as A and B
//     agree -- so that every lane reads its 16 k values as four 16-byte
//     ds_read_b128 (lane half h reads k = 16h .. 16h+15).
//   * LDS rows are padded to 36 floats (9 x 16-B slots, odd): the four 16-lane
//     groups of a ds_read_b128 hit 16 distinct slots -> conflict free.
//   * global -> register -> LDS staging, double-buffered LDS, ONE barrier per
//     K tile; next tile's global loads are issued before the MFMA block so HBM
//     / L2 latency hides under ~2-4k cycles of matrix work.
//   * conv mode builds the im2col operand on the fly from an NHWC image
//     (K = [tap][cin], Cin % 32 == 0 so a K tile never straddles a tap), with
//     nearest-x2 upsample / stride-2 asymmetric-pad variants folded into the
//     coordinate math and GroupNorm-apply + swish folded into the operand load
//     (per-(image,channel) scale/shift tables) -- zero padding is applied AFTER
//     the activation, as the reference pads the activated tensor.
#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = 36;

__device__ __forceinline__ float gelu_erf(float v) {
  return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}

__device__ __forceinline__ f32x4 prologue4(f32x4 v, const float* sc, const float* sh,
                                           int act) {
  const f32x4 s = *reinterpret_cast<const f32x4*>(sc);
  const f32x4 t = *reinterpret_cast<const f32x4*>(sh);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float x = fmaf(v[e], s[e], t[e]);
    if (act == 1) x = x / (1.0f + expf(-x));
    v[e] = x;
  }
  return v;
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int AMODE, bool PRO, bool BTRANS>
__global__ __launch_bounds__(256) void gemm_kernel(const t2h_gemm_args p) {
  static_assert(WARPS_M * WARPS_N == 4, "4 waves per workgroup");
  constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold a 32x32 MFMA tile");
  constexpr int A_F4 = BM / 32;  // float4 loads per thread for one A tile
  constexpr int B_F4 = BN / 32;

  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDS_LD];
  float* const As = smem;
  float* const Bs = smem + 2 * BM * LDS_LD;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wm0 = (wave / WARPS_N) * WM, wn0 = (wave % WARPS_N) * WN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int64_t zb = blockIdx.z;
  const float* __restrict__ Ag = p.A + zb * p.strideA;
  const float* __restrict__ Bg = p.B + zb * p.strideB;
  float* __restrict__ Cg = p.C + zb * p.strideC;

  const int col4 = tid & 7, row0 = tid >> 3;

  // ---- per-thread A row bookkeeping
  int64_t a_off[A_F4];  // plain: element offset of (row, col4*4); conv: image base pixel
  int a_y[A_F4], a_x[A_F4], a_img[A_F4];
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    const int m = m0 + row0 + 32 * i;
    const bool ok = m < p.M;
    if (AMODE == 0) {
      a_off[i] = ok ? (int64_t)m * p.lda + col4 * 4 : -1;
      a_img[i] = (PRO && ok) ? m / p.pro_rows : 0;
      a_y[i] = a_x[i] = 0;
    } else {
      const int hw = p.Hout * p.Wout;
      const int b = ok ? m / hw : 0;
      const int rem = ok ? m - b * hw : 0;
      const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
      a_img[i] = b;
      a_off[i] = (int64_t)b * p.Hin * p.Win;
      a_y[i] = ok ? oy * p.stride - p.pad : -(1 << 28);
      a_x[i] = ox * p.stride - p.pad;
    }
  }
  const int Hlim = (AMODE == 1) ? (p.Hin << p.ups) : 0;
  const int Wlim = (AMODE == 1) ? (p.Win << p.ups) : 0;

  f32x4 a_reg[A_F4], b_reg[B_F4];

  auto load_tiles = [&](int kt) {
    const int k0 = kt * BK;
    if (AMODE == 0) {
#pragma unroll
      for (int i = 0; i < A_F4; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (a_off[i] >= 0) {
          v = *reinterpret_cast<const f32x4*>(Ag + a_off[i] + k0);
          if (PRO) {
            const int64_t t = (int64_t)a_img[i] * p.pro_ld + k0 + col4 * 4;
            v = prologue4(v, p.pro_scale + t, p.pro_shift + t, p.pro_act);
          }
        }
        a_reg[i] = v;
      }
    } else {
      const int tap = k0 / p.Cin;
      const int c0 = k0 - tap * p.Cin;
      const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
      for (int i = 0; i < A_F4; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        int iy = a_y[i] + dy, ix = a_x[i] + dx;
        if ((unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim) {
          iy >>= p.ups;
          ix >>= p.ups;
          const int64_t pix = a_off[i] + (int64_t)iy * p.Win + ix;
          v = *reinterpret_cast<const f32x4*>(Ag + pix * p.lda + c0 + col4 * 4);
          if (PRO) {
            const int64_t t = (int64_t)a_img[i] * p.pro_ld + c0 + col4 * 4;
            v = prologue4(v, p.pro_scale + t, p.pro_shift + t, p.pro_act);
          }
        }
        a_reg[i] = v;
      }
    }
    if (!BTRANS) {
#pragma unroll
      for (int i = 0; i < B_F4; ++i) {
        const int n = n0 + row0 + 32 * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < p.N) v = *reinterpret_cast<const f32x4*>(Bg + (int64_t)n * p.ldb + k0 + col4 * 4);
        b_reg[i] = v;
      }
    } else {
      constexpr int NQ = BN / 4;        // float4 per k-row of the tile
      constexpr int RPP = 256 / NQ;     // k-rows per pass
      const int n4 = tid % NQ, kr = tid / NQ;
#pragma unroll
      for (int i = 0; i < B_F4; ++i) {
        const int k = k0 + kr + i * RPP;
        const int n = n0 + n4 * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < p.N) v = *reinterpret_cast<const f32x4*>(Bg + (int64_t)k * p.ldb + n);
        b_reg[i] = v;
      }
    }
  };

  auto store_tiles = [&](int buf) {
    float* Ad = As + buf * BM * LDS_LD;
    float* Bd = Bs + buf * BN * LDS_LD;
#pragma unroll
    for (int i = 0; i < A_F4; ++i)
      *reinterpret_cast<f32x4*>(Ad + (row0 + 32 * i) * LDS_LD + col4 * 4) = a_reg[i];
    if (!BTRANS) {
#pragma unroll
      for (int i = 0; i < B_F4; ++i)
        *reinterpret_cast<f32x4*>(Bd + (row0 + 32 * i) * LDS_LD + col4 * 4) = b_reg[i];
    } else {
      constexpr int NQ = BN / 4;
      constexpr int RPP = 256 / NQ;
      const int n4 = tid % NQ, kr = tid / NQ;
#pragma unroll
      for (int i = 0; i < B_F4; ++i) {
        const int kk = kr + i * RPP;           // logical k inside the tile
#pragma unroll
        for (int e = 0; e < 4; ++e) Bd[(n4 * 4 + e) * LDS_LD + kk] = b_reg[i][e];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);

    const float* Ab = As + buf * BM * LDS_LD + (wm0 + l31) * LDS_LD + 16 * hh;
    const float* Bb = Bs + buf * BN * LDS_LD + (wn0 + l31) * LDS_LD + 16 * hh;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int ti = 0; ti < TM; ++ti)
        af[ti] = *reinterpret_cast<const f32x4*>(Ab + ti * 32 * LDS_LD + 4 * j);
#pragma unroll
      for (int tj = 0; tj < TN; ++tj)
        bf[tj] = *reinterpret_cast<const f32x4*>(Bb + tj * 32 * LDS_LD + 4 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
          for (int tj = 0; tj < TN; ++tj)
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ti][e], bf[tj][e],
                                                                acc[ti][tj], 0, 0, 0);
    }

    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31,
  //      row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int ti = 0; ti < TM; ++ti) {
#pragma unroll
    for (int tj = 0; tj < TN; ++tj) {
      const int col = n0 + wn0 + tj * 32 + l31;
      if (col >= p.N) continue;
      const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (row >= p.M) continue;
        float v = acc[ti][tj][r] * p.alpha + bv;
        const float rv = p.residual ? p.residual[(int64_t)row * p.ldr + col] : 0.f;
        if (p.res_pre) v += rv;
        if (p.epi_act == 1) v = gelu_erf(v);
        else if (p.epi_act == 2) v = fmaxf(v, 0.f);
        if (!p.res_pre) v += rv;
        Cg[(int64_t)row * p.ldc + col] = v;
      }
    }
  }
}

template <int BM, int BN, int WARPS_M, int WARPS_N>
int launch_cfg(const t2h_gemm_args& a, hipStream_t s) {
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.batch);
  dim3 block(256);
  const bool pro = a.pro_scale != nullptr;
  if (a.a_mode == 0) {
    if (a.b_trans) {
      hipLaunchKernelGGL((gemm_kernel<BM, BN, WARPS_M, WARPS_N, 0, false, true>), grid, block, 0, s, a);
    } else if (pro) {
      hipLaunchKernelGGL((gemm_kernel<BM, BN, WARPS_M, WARPS_N, 0, true, false>), grid, block, 0, s, a);
    } else {
      hipLaunchKernelGGL((gemm_kernel<BM, BN, WARPS_M, WARPS_N, 0, false, false>), grid, block, 0, s, a);
    }
  } else {
    if (pro) {
      hipLaunchKernelGGL((gemm_kernel<BM, BN, WARPS_M, WARPS_N, 1, true, false>), grid, block, 0, s, a);
    } else {
      hipLaunchKernelGGL((gemm_kernel<BM, BN, WARPS_M, WARPS_N, 1, false, false>), grid, block, 0, s, a);
    }
  }
  T2H_CHECK_LAUNCH("t2h_gemm_f32");
  return T2H_OK;
}

}  // namespace

extern "C" int t2h_gemm_f32(const t2h_gemm_args* args, void* stream) {
  T2H_REQUIRE(args != nullptr, "t2h_gemm_f32: args is NULL");
  t2h_gemm_args a = *args;
  if (a.batch < 1) a.batch = 1;
  T2H_REQUIRE(a.A && a.B && a.C, "t2h_gemm_f32: NULL operand");
  T2H_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "t2h_gemm_f32: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  T2H_REQUIRE(a.K % BK == 0, "t2h_gemm_f32: K=%d must be a multiple of %d", a.K, BK);
  T2H_REQUIRE(a.lda % 4 == 0 && a.ldb % 4 == 0, "t2h_gemm_f32: lda/ldb must be multiples of 4");
  T2H_REQUIRE(t2h_aligned16(a.A) && t2h_aligned16(a.B), "t2h_gemm_f32: A/B must be 16-byte aligned");
  T2H_REQUIRE(a.strideA % 4 == 0 && a.strideB % 4 == 0, "t2h_gemm_f32: batch strides must be multiples of 4");
  if (a.pro_scale) {
    T2H_REQUIRE(a.pro_shift != nullptr, "t2h_gemm_f32: pro_shift missing");
    T2H_REQUIRE(!a.b_trans, "t2h_gemm_f32: prologue with b_trans unsupported");
    T2H_REQUIRE(a.pro_ld % 4 == 0 && t2h_aligned16(a.pro_scale) && t2h_aligned16(a.pro_shift),
                "t2h_gemm_f32: prologue tables must be 16-byte aligned rows");
    if (a.a_mode == 0) T2H_REQUIRE(a.pro_rows > 0, "t2h_gemm_f32: pro_rows must be > 0");
  }
  if (a.a_mode == 1) {
    T2H_REQUIRE(!a.b_trans, "t2h_gemm_f32: conv with b_trans unsupported");
    T2H_REQUIRE(a.Cin % BK == 0 && a.K == 9 * a.Cin, "t2h_gemm_f32: conv needs Cin %% 32 == 0 and K == 9*Cin (Cin=%d K=%d)", a.Cin, a.K);
    T2H_REQUIRE(a.Hin > 0 && a.Win > 0 && a.Hout > 0 && a.Wout > 0 && a.M % (a.Hout * a.Wout) == 0,
                "t2h_gemm_f32: bad conv geometry");
    T2H_REQUIRE(a.stride >= 1 && a.ups >= 0 && a.ups <= 1, "t2h_gemm_f32: bad stride/ups");
  } else {
    T2H_REQUIRE(a.a_mode == 0, "t2h_gemm_f32: unknown a_mode %d", a.a_mode);
    if (a.b_trans) T2H_REQUIRE(a.N % 4 == 0, "t2h_gemm_f32: b_trans needs N %% 4 == 0");
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t tiles128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.batch;
  if (a.M <= 64) return launch_cfg<64, 64, 2, 2>(a, s);
  if (a.N <= 32) return launch_cfg<128, 32, 4, 1>(a, s);
  if (a.N <= 64 || (a.N % 128 != 0 && a.N % 128 <= 64)) return launch_cfg<128, 64, 2, 2>(a, s);
  if (tiles128 >= 512) return launch_cfg<128, 128, 2, 2>(a, s);
  return launch_cfg<128, 64, 2, 2>(a, s);
}
