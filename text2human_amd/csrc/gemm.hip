// fp32 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   C[M,N] = epi(alpha * A'[M,K] * B[N,K]^T + bias) + residual
//
// One kernel family serves every dense contraction of the Text2Human sampling
// path (see include/t2h_hip.h).  Design (MI355X-first):
//   * v_mfma_f32_32x32x2_f32: exact-fp32 fma chains at the 157 TFLOP/s matrix
//     rate; a workgroup = 4 or 8 waves, each wave owns a (BM/WARPS_M) x
//     (BN/WARPS_N) tile of 32x32 accumulators.
//   * K is walked in tiles of BK (32 or 64).  The MFMA k-pair of step s is
//     (s, BK/2+s) inside the tile -- any fixed k permutation is legal as long
//     as A and B agree -- so that every lane reads its k values as 16-byte
//     ds_read_b128 (lane half h reads k = h*BK/2 .. h*BK/2+BK/2-1).
//   * LDS rows are padded to BK+4 floats (an odd number of 16-B slots): the
//     four 16-lane groups of a ds_read_b128 hit 16 distinct slots -> conflict
//     free.
//   * global -> register -> LDS staging, double-buffered LDS, ONE barrier per
//     K tile; the next tile's global loads are issued before the MFMA block so
//     HBM / L2 latency hides under the tile's matrix work (BK/2 x TM x TN MFMAs
//     of 64 cycles each per wave).
//   * conv mode builds the im2col operand on the fly from an NHWC image
//     (K = [tap][cin], Cin % BK == 0 so a K tile never straddles a tap), with
//     nearest-x2 upsample / stride-2 asymmetric-pad variants folded into the
//     coordinate math and GroupNorm-apply + swish folded into the operand
//     staging (per-(image,channel) scale/shift tables) -- zero padding is
//     applied AFTER the activation, as the reference pads the activated tensor.
#include <type_traits>

#include "common.h"

namespace {

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
    if (act == 1) x = x / (1.0f + fast_exp(fminf(-x, 87.0f)));
    v[e] = x;
  }
  return v;
}

#ifdef T2H_GEMM_TIMING
__device__ long long* g_timing_buf = nullptr;  // debug builds only (tools/gemm_timing.py)
#endif

// Inline-asm global load + counted wait: loads hipcc must NOT track, so that a
// register piece can stay in flight across two K tiles and be waited for with an
// exact `s_waitcnt vmcnt(N)` (hipcc only ever emits vmcnt(0) for loop-carried
// loads).  The wait names the destination "+v" so the compiler cannot consume
// the registers before the data has landed.
__device__ __forceinline__ void gload4_async(f32x4& dst, const float* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt(f32x4& v) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N));
}

// Register staging set of one K tile (global -> registers -> LDS).
template <int A_F4, int B_F4>
struct Stage {
  f32x4 a[A_F4];
  f32x4 b[B_F4];
  f32x4 sc, sh;    // PRO == 1: this thread's channel-quad scale / shift for the tile
  unsigned valid;  // PRO == 1: bit i set <=> a[i] is a real element (not zero padding)
};

// PRO: 0 none; 1 prologue with one image per workgroup (rows_per_image % BM == 0):
// raw values stay in registers and are transformed when they are written to LDS,
// so the global loads are never waited for early; 2 generic prologue (a workgroup
// may straddle images): transformed right after the load.
template <int BM, int BN, int BK, int WARPS_M, int WARPS_N, int AMODE, int PRO, bool BTRANS>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N) void gemm_kernel(const t2h_gemm_args p) {
  constexpr int NT = 64 * WARPS_M * WARPS_N;
  constexpr int LDS_LD = BK + 4;
  constexpr int KQ = BK / 4;  // float4 per tile row
  constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold a 32x32 MFMA tile");
  constexpr int RPP = NT / KQ;          // tile rows staged per pass
  constexpr int A_F4 = BM / RPP;        // float4 loads per thread for one A tile
  constexpr int B_F4 = BN / RPP;
  static_assert(A_F4 >= 1 && B_F4 >= 1 && BM % RPP == 0 && BN % RPP == 0, "bad staging shape");
  using StageT = Stage<A_F4, B_F4>;

  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDS_LD];
  float* const As = smem;
  float* const Bs = smem + 2 * BM * LDS_LD;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wm0 = (wave / WARPS_N) * WM, wn0 = (wave % WARPS_N) * WN;
  // XCD-aware tile mapping.  Workgroup b is dispatched to XCD b % 8 (8 XCDs,
  // private 4 MiB L2 each, 32 CUs).  Give every XCD a CONTIGUOUS range of the
  // row-major (m-tile, n-tile) sequence so that the workgroups that run together
  // on one XCD share their A rows / B columns through that XCD's L2 instead of
  // every XCD streaming the whole A matrix (bijective for any tile count).
  const int nbx = (p.N + BN - 1) / BN, nby = (p.M + BM - 1) / BM;
  int m0, n0;
  {
    const int total = nbx * nby, b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3, q = total >> 3, r = total & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int mt = lin / nbx;
    m0 = mt * BM;
    n0 = (lin - mt * nbx) * BN;
  }
  const int64_t zb = blockIdx.y;
  // split over K across workgroups (conv mode, small pixel counts: the deep levels of the index-prediction / parsing
  // UNets, where M = images x a handful of pixels gives 8-32 tiles that each stream megabytes of weights): slice
  // blockIdx.z of `ksplit` takes K tiles [kt_begin, kt_begin + nk) and stores its PARTIAL tile, unscaled, to
  // splitk_ws[slice][M][N]; splitk_reduce_kernel sums the slices in slice order and applies the epilogue
  const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
  const int nk_all = p.K / BK;
  const int kt_begin = (int)(((int64_t)blockIdx.z * nk_all) / ksplit);
  const int nk = (int)(((int64_t)(blockIdx.z + 1) * nk_all) / ksplit) - kt_begin;
  const float* __restrict__ Ag = p.A + zb * p.strideA;
  const float* __restrict__ Bg = p.B + zb * p.strideB;
  float* __restrict__ Cg = p.C + zb * p.strideC;

  const int col4 = tid % KQ, row0 = tid / KQ;

  // ---- per-thread A row bookkeeping
  int64_t a_off[A_F4];  // plain: element offset of (row, col4*4); conv: image base pixel
  int a_y[A_F4], a_x[A_F4], a_img[A_F4];
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    const int m = m0 + row0 + RPP * i;
    const bool ok = m < p.M;
    if (AMODE == 0) {
      a_off[i] = ok ? (int64_t)m * p.lda + col4 * 4 : -1;
      a_img[i] = (PRO == 2 && ok) ? m / p.pro_rows : 0;
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
  // PRO == 1: the whole workgroup lies in one image
  const int64_t img0_tbl =
      (PRO == 1) ? (int64_t)(m0 / (AMODE == 0 ? p.pro_rows : p.Hout * p.Wout)) * p.pro_ld : 0;

  StageT st;
  st.valid = 0u;

  // ---- staging "pieces": one float4 per piece, so that the main loop can issue
  // them one at a time in the shadow of individual MFMAs.
  struct TileK {  // wave-uniform description of K tile kt
    int k0, c0, dy, dx;
  };
  auto tile_k = [&](int kt) {  // kt: K tile of THIS workgroup's slice
    TileK t;
    t.k0 = (kt + kt_begin) * BK;
    t.c0 = t.dy = t.dx = 0;
    if (AMODE == 1) {
      const int tap = t.k0 / p.Cin;
      t.c0 = t.k0 - tap * p.Cin;
      t.dy = tap / 3;
      t.dx = tap - 3 * t.dy;
    }
    return t;
  };
  // Loads are UNCONDITIONAL from clamped (always valid) addresses; validity is
  // tracked in bit masks and applied when the piece is written to LDS, so the
  // main loop has no exec-masked branches around its global loads.
  unsigned b_ok = 0u;
#pragma unroll
  for (int i = 0; i < B_F4; ++i) {
    const bool ok = BTRANS ? (n0 + (tid % (BN / 4)) * 4 < p.N) : (n0 + row0 + RPP * i < p.N);
    if (ok) b_ok |= 1u << i;
  }
  auto load_tbl = [&](const TileK& t) {  // PRO == 1 scale/shift of this tile
    if (PRO == 1) {
      const int c = (AMODE == 0 ? t.k0 : t.c0) + col4 * 4;
      st.sc = *reinterpret_cast<const f32x4*>(p.pro_scale + img0_tbl + c);
      st.sh = *reinterpret_cast<const f32x4*>(p.pro_shift + img0_tbl + c);
    }
  };
  auto load_a = [&](int i, const TileK& t) {
    bool ok;
    const float* src;
    if (AMODE == 0) {
      ok = a_off[i] >= 0;
      src = Ag + (ok ? a_off[i] : (int64_t)(col4 * 4)) + t.k0;
    } else {
      const int iy = a_y[i] + t.dy, ix = a_x[i] + t.dx;
      ok = (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim;
      const int cy = min(max(iy, 0), Hlim - 1) >> p.ups, cx = min(max(ix, 0), Wlim - 1) >> p.ups;
      src = Ag + (a_off[i] + (int64_t)cy * p.Win + cx) * p.lda + t.c0 + col4 * 4;
    }
    f32x4 v = *reinterpret_cast<const f32x4*>(src);
    st.valid = (st.valid & ~(1u << i)) | ((ok ? 1u : 0u) << i);
    if (PRO == 2) {  // generic prologue: transform now (waits for the load)
      const int64_t o = (int64_t)a_img[i] * p.pro_ld + (AMODE == 0 ? t.k0 : t.c0) + col4 * 4;
      v = prologue4(v, p.pro_scale + o, p.pro_shift + o, p.pro_act);
    }
    st.a[i] = v;
  };
  constexpr int NQ = BN / 4;    // BTRANS: float4 per k-row of the tile
  constexpr int KPP = NT / NQ;  // BTRANS: k-rows per pass; BK / KPP == B_F4
  auto load_b = [&](int i, const TileK& t) {
    const bool ok = (b_ok >> i) & 1u;
    const float* src;
    if (!BTRANS) {
      src = Bg + (int64_t)(ok ? n0 + row0 + RPP * i : 0) * p.ldb + t.k0 + col4 * 4;
    } else {
      src = Bg + (int64_t)(t.k0 + tid / NQ + i * KPP) * p.ldb + (ok ? n0 + (tid % NQ) * 4 : 0);
    }
    st.b[i] = *reinterpret_cast<const f32x4*>(src);
  };
  auto store_a = [&](int i, int buf) {
    f32x4 v = st.a[i];
    const bool ok = (st.valid >> i) & 1u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = v[e];
      if (PRO == 1) {
        x = fmaf(x, st.sc[e], st.sh[e]);
        if (p.pro_act == 1) x = x / (1.0f + fast_exp(fminf(-x, 87.0f)));
      }
      v[e] = ok ? x : 0.f;
    }
    *reinterpret_cast<f32x4*>(As + buf * BM * LDS_LD + (row0 + RPP * i) * LDS_LD + col4 * 4) = v;
  };
  auto store_b = [&](int i, int buf) {
    float* Bd = Bs + buf * BN * LDS_LD;
    const bool ok = (b_ok >> i) & 1u;
    f32x4 v = st.b[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
    if (!BTRANS) {
      *reinterpret_cast<f32x4*>(Bd + (row0 + RPP * i) * LDS_LD + col4 * 4) = v;
    } else {
      const int kk = tid / NQ + i * KPP;  // logical k inside the tile
#pragma unroll
      for (int e = 0; e < 4; ++e) Bd[((tid % NQ) * 4 + e) * LDS_LD + kk] = v[e];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int last = nk - 1;

  // MFMA work of one K tile out of LDS[buf]; `aux(s)` is called after MFMA step s.
  auto mma_tile = [&](int buf, auto&& aux) {
    const float* Ab = As + buf * BM * LDS_LD + (wm0 + l31) * LDS_LD + (BK / 2) * hh;
    const float* Bb = Bs + buf * BN * LDS_LD + (wn0 + l31) * LDS_LD + (BK / 2) * hh;
    f32x4 af[2][TM], bf[2][TN];  // MFMA operand fragments, double-buffered over j
#pragma unroll
    for (int ti = 0; ti < TM; ++ti) af[0][ti] = *reinterpret_cast<const f32x4*>(Ab + ti * 32 * LDS_LD);
#pragma unroll
    for (int tj = 0; tj < TN; ++tj) bf[0][tj] = *reinterpret_cast<const f32x4*>(Bb + tj * 32 * LDS_LD);
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
      if (j + 1 < BK / 8) {  // next group's fragments fly under this group's MFMAs
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
          af[(j + 1) & 1][ti] = *reinterpret_cast<const f32x4*>(Ab + ti * 32 * LDS_LD + 4 * (j + 1));
#pragma unroll
        for (int tj = 0; tj < TN; ++tj)
          bf[(j + 1) & 1][tj] = *reinterpret_cast<const f32x4*>(Bb + tj * 32 * LDS_LD + 4 * (j + 1));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
          for (int tj = 0; tj < TN; ++tj)
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j & 1][ti][e], bf[j & 1][tj][e],
                                                                acc[ti][tj], 0, 0, 0);
        aux(4 * j + e);
      }
    }
  };

  constexpr bool DEEP = (AMODE == 0 && PRO == 0 && !BTRANS);
  if constexpr (DEEP) {
    // ---- plain GEMM: TWO register sets, every global load flies for two K tiles.
    // Slot q (B pieces, then A pieces) of the set holding tile kt+1 is waited for
    // with an exact vmcnt (all 2L-1 younger loads may still be in flight), written
    // to LDS[(kt+1)&1] and immediately re-issued for tile kt+3.
    constexpr int L = A_F4 + B_F4;
    constexpr int STEPS = BK / 2;
    f32x4 ra[2][A_F4], rb[2][B_F4];
    const float* a_src[A_F4];
    const float* b_src[B_F4];
    unsigned a_ok = 0u;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const bool ok = a_off[i] >= 0;
      a_src[i] = Ag + (ok ? a_off[i] : (int64_t)(col4 * 4));
      if (ok) a_ok |= 1u << i;
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const bool ok = (b_ok >> i) & 1u;
      b_src[i] = Bg + (int64_t)(ok ? n0 + row0 + RPP * i : 0) * p.ldb + col4 * 4;
    }
    auto put_a = [&](int i, const f32x4& r, int buf) {
      f32x4 v = r;
      const bool ok = (a_ok >> i) & 1u;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
      *reinterpret_cast<f32x4*>(As + buf * BM * LDS_LD + (row0 + RPP * i) * LDS_LD + col4 * 4) = v;
    };
    auto put_b = [&](int i, const f32x4& r, int buf) {
      f32x4 v = r;
      const bool ok = (b_ok >> i) & 1u;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
      *reinterpret_cast<f32x4*>(Bs + buf * BN * LDS_LD + (row0 + RPP * i) * LDS_LD + col4 * 4) = v;
    };
    auto issue = [&](auto setc, int kt) {  // all loads of tile kt into register set S
      constexpr int S = decltype(setc)::value;
      const int k0 = (min(kt, last) + kt_begin) * BK;
#pragma unroll
      for (int i = 0; i < B_F4; ++i) gload4_async(rb[S][i], b_src[i] + k0);
#pragma unroll
      for (int i = 0; i < A_F4; ++i) gload4_async(ra[S][i], a_src[i] + k0);
    };
    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;
    issue(set0{}, 0);
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      wait_vmcnt<0>(rb[0][i]);
      put_b(i, rb[0][i], 0);
    }
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      wait_vmcnt<0>(ra[0][i]);
      put_a(i, ra[0][i], 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    issue(set1{}, 1);
    issue(set0{}, 2);
    __syncthreads();

    auto step = [&](int kt, auto setc) {  // set S holds tile kt+1
      constexpr int S = decltype(setc)::value;
      const int buf = kt & 1;
      const int kn = (min(kt + 3, last) + kt_begin) * BK;
      mma_tile(buf, [&](int s) {
#pragma unroll
        for (int q = 0; q < L; ++q) {
          const int at = (L <= STEPS) ? STEPS - L + q : (q * STEPS) / L;
          if (at != s) continue;
          __builtin_amdgcn_sched_barrier(0);
          if (q < B_F4) {
            wait_vmcnt<2 * L - 1>(rb[S][q]);
            put_b(q, rb[S][q], buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            gload4_async(rb[S][q], b_src[q] + kn);
          } else {
            wait_vmcnt<2 * L - 1>(ra[S][q - B_F4]);
            put_a(q - B_F4, ra[S][q - B_F4], buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            gload4_async(ra[S][q - B_F4], a_src[q - B_F4] + kn);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      __syncthreads();
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      step(kt, set1{});
      step(kt + 1, set0{});
    }
    if (nk & 1) step(nk - 1, set1{});
    // drain the clamped tail loads before the registers can be reused
#pragma unroll
    for (int S = 0; S < 2; ++S) {
#pragma unroll
      for (int i = 0; i < B_F4; ++i) wait_vmcnt<0>(rb[S][i]);
#pragma unroll
      for (int i = 0; i < A_F4; ++i) wait_vmcnt<0>(ra[S][i]);
    }
  } else {
  // ---- prologue: tile 0 -> LDS[0]; tile 1 -> registers
  {
    const TileK t0 = tile_k(0);
    load_tbl(t0);
#pragma unroll
    for (int i = 0; i < A_F4; ++i) load_a(i, t0);
#pragma unroll
    for (int i = 0; i < B_F4; ++i) load_b(i, t0);
#pragma unroll
    for (int i = 0; i < A_F4; ++i) store_a(i, 0);
#pragma unroll
    for (int i = 0; i < B_F4; ++i) store_b(i, 0);
    const TileK t1 = tile_k(min(1, last));
    load_tbl(t1);
#pragma unroll
    for (int i = 0; i < A_F4; ++i) load_a(i, t1);
#pragma unroll
    for (int i = 0; i < B_F4; ++i) load_b(i, t1);
  }
  __syncthreads();

  // ---- main loop, software-pipelined and branch-free.  While tile kt is
  // multiplied out of LDS[kt&1], the registers (tile kt+1, loaded during the
  // previous iteration) are written to LDS[(kt+1)&1] and then refilled with
  // tile kt+2 -- one float4 "piece" after each MFMA step in the second part of
  // the step sequence, so the address math, ds_writes and global loads issue in
  // the shadow of the 64-cycle MFMAs instead of between tiles.  Tail tiles are
  // clamped (harmless duplicate loads / stores) so there is no divergent path.
  constexpr int STEPS = BK / 2;             // MFMA k-steps per tile
  constexpr int NPIECE = A_F4 + B_F4;       // float4 pieces per tile and direction
  constexpr int FIRST = (2 * NPIECE <= STEPS) ? STEPS - 2 * NPIECE : 0;
#ifdef T2H_GEMM_TIMING
  const long long tm_begin = clock64();
  long long tm_bar = 0;
#endif
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const TileK tn = tile_k(min(kt + 2, last));
    mma_tile(buf, [&](int s) {
        // auxiliary pieces pinned after MFMA step s: stores first, then loads
#pragma unroll
        for (int q = 0; q < 2 * NPIECE; ++q) {
          const int at = (2 * NPIECE <= STEPS) ? FIRST + q : (q * STEPS) / (2 * NPIECE);
          if (at != s) continue;
          __builtin_amdgcn_sched_barrier(0);
          // piece order: B pieces first (the PRO tables are reloaded with the
          // first A load, after every A store has consumed them), each register
          // piece is stored and IMMEDIATELY re-loaded with the tile after next,
          // which gives every global load ~a full tile period before its use.
          const int pc = q >> 1;  // piece index: [0,B_F4) = B, [B_F4,NPIECE) = A
#ifndef T2H_DBG_NOSTORE
          if ((q & 1) == 0) {
            if (pc < B_F4) store_b(pc, buf ^ 1);
            else store_a(pc - B_F4, buf ^ 1);
          }
#endif
#ifndef T2H_DBG_NOLOAD
          if ((q & 1) == 1) {
            if (pc < B_F4) load_b(pc, tn);
            else {
              load_a(pc - B_F4, tn);
              if (pc == NPIECE - 1) load_tbl(tn);  // after the last A store used the old tables
            }
          }
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
    });
#ifdef T2H_GEMM_TIMING
    const long long tb0 = clock64();
#endif
    __syncthreads();
#ifdef T2H_GEMM_TIMING
    tm_bar += clock64() - tb0;
#endif
  }
#ifdef T2H_GEMM_TIMING
  if (g_timing_buf && blockIdx.x == 8 && tid == 0) {
    g_timing_buf[0] = 0; g_timing_buf[1] = 0; g_timing_buf[2] = 0;
    g_timing_buf[3] = tm_bar; g_timing_buf[4] = clock64() - tm_begin; g_timing_buf[5] = nk;
  }
#endif
  }  // !DEEP

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31,
  //      row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (ksplit > 1) {  // partial tile of this K slice, as accumulated
    float* const W = p.splitk_ws + (int64_t)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
      for (int tj = 0; tj < TN; ++tj) {
        const int col = n0 + wn0 + tj * 32 + l31;
        if (col >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm0 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (row < p.M) W[(int64_t)row * p.N + col] = acc[ti][tj][r];
        }
      }
    return;
  }
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

// The slices of a K-split launch summed in slice order (fixed: bit-reproducible), then gemm_kernel's epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const t2h_gemm_args p, int ksplit) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t mn = (int64_t)p.M * p.N;
  if (i >= mn) return;
  const int row = (int)(i / p.N), col = (int)(i - (int64_t)row * p.N);
  float a = p.splitk_ws[i];
  for (int s = 1; s < ksplit; ++s) a += p.splitk_ws[s * mn + i];
  float v = a * p.alpha + (p.bias ? p.bias[col] : 0.f);
  const float rv = p.residual ? p.residual[(int64_t)row * p.ldr + col] : 0.f;
  if (p.res_pre) v += rv;
  if (p.epi_act == 1) v = gelu_erf(v);
  else if (p.epi_act == 2) v = fmaxf(v, 0.f);
  if (!p.res_pre) v += rv;
  p.C[(int64_t)row * p.ldc + col] = v;
}

// ---- launch: every (tile, mode) combination
template <int BM, int BN, int BK, int WARPS_M, int WARPS_N, bool FULL>
int launch_cfg(const t2h_gemm_args& a, hipStream_t s) {
  dim3 grid(((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM), a.batch, a.ksplit > 1 ? a.ksplit : 1);
  dim3 block(64 * WARPS_M * WARPS_N);
  const bool pro = a.pro_scale != nullptr;
  const int rows_per_img = a.a_mode == 0 ? a.pro_rows : a.Hout * a.Wout;
  const bool uniform = pro && rows_per_img % BM == 0;
#define T2H_LAUNCH(AM, PR, BT)                                                                    \
  hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, WARPS_M, WARPS_N, AM, PR, BT>), grid, block, 0, s, a)
  if (a.a_mode == 0) {
    if (a.b_trans) {
      if constexpr (FULL) T2H_LAUNCH(0, 0, true); else return T2H_ERR_UNSUPPORTED;
    } else if (uniform) {
      T2H_LAUNCH(0, 1, false);
    } else if (pro) {
      if constexpr (FULL) T2H_LAUNCH(0, 2, false); else return T2H_ERR_UNSUPPORTED;
    } else {
      T2H_LAUNCH(0, 0, false);
    }
  } else {
    if (uniform) {
      T2H_LAUNCH(1, 1, false);
    } else if (pro) {
      if constexpr (FULL) T2H_LAUNCH(1, 2, false); else return T2H_ERR_UNSUPPORTED;
    } else {
      T2H_LAUNCH(1, 0, false);
    }
  }
#undef T2H_LAUNCH
  if (a.ksplit > 1)
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((int64_t)a.M * a.N + 255) / 256)), dim3(256), 0, s, a, a.ksplit);
  T2H_CHECK_LAUNCH("t2h_gemm_f32");
  return T2H_OK;
}

// Tile configurations.  id -> (BM, BN, BK, waves).  0-3 support every mode
// (b_trans, generic prologue); 4+ are the high-throughput variants.
enum { CFG_64x64 = 0, CFG_128x32, CFG_128x64, CFG_128x128, CFG_128x64_K64, CFG_128x128_K64,
       CFG_128x64_W8, CFG_128x64_K64_W8, CFG_128x128_K64_W8, CFG_COUNT };

int cfg_bk(int cfg) {
  return (cfg == CFG_128x64_K64 || cfg == CFG_128x128_K64 || cfg == CFG_128x64_K64_W8 ||
          cfg == CFG_128x128_K64_W8) ? 64 : 32;
}

int launch_by_cfg(int cfg, const t2h_gemm_args& a, hipStream_t s) {
  switch (cfg) {
    case CFG_64x64: return launch_cfg<64, 64, 32, 2, 2, true>(a, s);
    case CFG_128x32: return launch_cfg<128, 32, 32, 4, 1, true>(a, s);
    case CFG_128x64: return launch_cfg<128, 64, 32, 2, 2, true>(a, s);
    case CFG_128x128: return launch_cfg<128, 128, 32, 2, 2, true>(a, s);
    case CFG_128x64_K64: return launch_cfg<128, 64, 64, 2, 2, false>(a, s);
    case CFG_128x128_K64: return launch_cfg<128, 128, 64, 2, 2, false>(a, s);
    case CFG_128x64_W8: return launch_cfg<128, 64, 32, 4, 2, false>(a, s);
    case CFG_128x64_K64_W8: return launch_cfg<128, 64, 64, 4, 2, false>(a, s);
    case CFG_128x128_K64_W8: return launch_cfg<128, 128, 64, 4, 2, false>(a, s);
    default: return T2H_ERR_INVALID;
  }
}

thread_local int g_force_cfg = -1;  // experiments / autotuning of the calling thread: t2h_gemm_force_config()

// Tile choice.  256 CUs, <= 2 resident 4-wave workgroups per CU (LDS).
int pick_cfg(const t2h_gemm_args& a) {
  const bool pro = a.pro_scale != nullptr;
  const int rows_per_img = a.a_mode == 0 ? a.pro_rows : a.Hout * a.Wout;
  const bool fast_ok = !a.b_trans && (!pro || rows_per_img % 128 == 0);
  if (g_force_cfg >= 0) {
    const bool k_ok = a.K % cfg_bk(g_force_cfg) == 0 && (a.a_mode == 0 || a.Cin % cfg_bk(g_force_cfg) == 0);
    if (k_ok && (g_force_cfg <= CFG_128x128 || fast_ok)) return g_force_cfg;
  }
  const int64_t tiles128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.batch;
  if (a.M <= 64) return CFG_64x64;
  if (a.N <= 32) return CFG_128x32;
  // plain GEMMs (transformer linears, 1x1 convs): 8 waves on a 128x64 tile keep two
  // waves per SIMD even when the grid is only one workgroup per CU (measured best
  // on every sampler shape at M = 4096, tools/gemm_bench.py)
  if (a.a_mode == 0 && fast_ok) return CFG_128x64_W8;
  if (a.N <= 64 || (a.N % 128 != 0 && a.N % 128 <= 64)) return CFG_128x64;
  if (tiles128 >= 512) return CFG_128x128;
  return CFG_128x64;
}

// K slices of a conv-mode launch, from (K, N, pixels per image) ONLY -- never from the number of images, so that an
// image's values do not depend on the batch (or decode chunk) it is computed in: as many slices as give a NOMINAL batch
// of 8 images (128 x 64 tiles) one workgroup per CU, at least four K tiles each, at most 32.
int pick_ksplit(const t2h_gemm_args& a) {
  if (a.a_mode != 1 || a.batch != 1 || !a.splitk_ws) return 1;
  const int64_t rpi = (int64_t)a.Hout * a.Wout;
  if (rpi > 512) return 1;
  const int nk = a.K / 32;
  const int64_t tiles8 = ((8 * rpi + 127) / 128) * ((a.N + 63) / 64);
  int ks = (int)((256 + tiles8 - 1) / tiles8);
  ks = ks < nk / 4 ? ks : nk / 4;
  ks = ks < 32 ? ks : 32;
  return ks > 1 ? ks : 1;
}

}  // namespace

#ifdef T2H_GEMM_TIMING
extern "C" int t2h_debug_set_timing_buffer(void* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_timing_buf), &dev_ptr, sizeof(void*));
}
#endif

extern "C" int t2h_gemm_force_config(int cfg) {
  const int old = g_force_cfg;
  g_force_cfg = (cfg >= 0 && cfg < CFG_COUNT) ? cfg : -1;
  return old;
}

extern "C" int t2h_gemm_f32(const t2h_gemm_args* args, void* stream) {
  T2H_REQUIRE(args != nullptr, "t2h_gemm_f32: args is NULL");
  t2h_gemm_args a = *args;
  if (a.batch < 1) a.batch = 1;
  T2H_REQUIRE(a.A && a.B && a.C, "t2h_gemm_f32: NULL operand");
  T2H_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "t2h_gemm_f32: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  T2H_REQUIRE(a.K % 32 == 0, "t2h_gemm_f32: K=%d must be a multiple of 32", a.K);
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
    T2H_REQUIRE(a.Cin % 32 == 0 && a.K == 9 * a.Cin, "t2h_gemm_f32: conv needs Cin %% 32 == 0 and K == 9*Cin (Cin=%d K=%d)", a.Cin, a.K);
    T2H_REQUIRE(a.Hin > 0 && a.Win > 0 && a.Hout > 0 && a.Wout > 0 && a.M % (a.Hout * a.Wout) == 0,
                "t2h_gemm_f32: bad conv geometry");
    T2H_REQUIRE(a.stride >= 1 && a.ups >= 0 && a.ups <= 1, "t2h_gemm_f32: bad stride/ups");
  } else {
    T2H_REQUIRE(a.a_mode == 0, "t2h_gemm_f32: unknown a_mode %d", a.a_mode);
    if (a.b_trans) T2H_REQUIRE(a.N % 4 == 0, "t2h_gemm_f32: b_trans needs N %% 4 == 0");
  }
  T2H_REQUIRE(a.ksplit >= 0 && a.ksplit <= 64, "t2h_gemm_f32: ksplit=%d (0 = automatic, <= 64)", a.ksplit);
  if (a.ksplit == 0) a.ksplit = pick_ksplit(a);
  if (a.ksplit > 1) {
    T2H_REQUIRE(a.batch == 1 && a.splitk_ws && a.ksplit <= a.K / 64,
                "t2h_gemm_f32: ksplit=%d needs batch 1, a workspace and at least 64 of K per slice (K=%d)", a.ksplit, a.K);
    if ((int64_t)a.ksplit * a.M * a.N > a.splitk_ws_floats) {
      T2H_REQUIRE(args->ksplit == 0, "t2h_gemm_f32: ksplit=%d needs a workspace of %lld floats (given %lld)", a.ksplit,
                  (long long)a.ksplit * a.M * a.N, (long long)a.splitk_ws_floats);
      // (automatic: the slice count must not depend on the batch, so a workspace that is too small for this batch is
      // an error of the caller's sizing, not a reason to compute differently)
      t2h_set_error("t2h_gemm_f32: the split-K workspace holds %lld floats, this launch needs %lld (ksplit %d x M %d x N %d)",
                    (long long)a.splitk_ws_floats, (long long)a.ksplit * a.M * a.N, a.ksplit, a.M, a.N);
      return T2H_ERR_INVALID;
    }
  }
  return launch_by_cfg(pick_cfg(a), a, static_cast<hipStream_t>(stream));
}

extern "C" int t2h_gemm_ksplit(const t2h_gemm_args* args) {
  if (!args) return -1;
  t2h_gemm_args a = *args;
  if (a.batch < 1) a.batch = 1;
  return a.ksplit > 0 ? a.ksplit : pick_ksplit(a);
}

extern "C" int t2h_gemm_tile_config(const t2h_gemm_args* args) {
  if (!args) return -1;
  t2h_gemm_args a = *args;
  if (a.batch < 1) a.batch = 1;
  return pick_cfg(a);
}
