// Implicit-GEMM convolution (3x3 / 1x1, NHWC fp32 activations) on the fp16 matrix cores at
// fp32-class accuracy: the split-precision arithmetic of gemm_split.hip (x = h + l / 2048, three
// partial products, fp32 accumulate) behind the operand staging of gemm.hip's conv mode.
//
//   C[M, Cout] = epi( A'[M, K] * W[Cout, K]^T + bias ) (+ residual),   K = taps * Cin, [tap][cin]
//
// Replaces, for the hierarchical VQGAN decode (DecoderRes + Decoder, models/archs/vqgan_arch.py:
// 1000-1033,1136-1151), the calls that t2h_gemm_f32 serves in conv mode: ResnetBlock conv1 / conv2
// (:597-617), Upsample (:529-534), conv_in, the AttnBlock 1x1 projections (:636-661) -- 563 GFLOP
// per image that ran at the fp32 matrix rate (157 TFLOP/s peak; 85-117 measured).
//
//  * BOTH operands are split rows (gemm_split.hip's format): weights [Cout][K/32][2][32] fp16, packed
//    once; activations [pixel][Cin/32][2][32] fp16, written by t2h_gn_apply_split_f32 (GroupNorm
//    apply + swish + split in ONE elementwise pass over the fp32 tensor) or t2h_split_rows_f32.  A K
//    tile (32 channels of one tap) of a pixel or of a weight row is one 128-byte line = eight
//    16-byte pieces that go global -> register -> LDS untouched; the only per-piece work is the
//    im2col address (tap shift, nearest-x2 upsample) and zeroing where the tap leaves the image.
//    (First version: fp32 activations with GroupNorm + swish + split applied while staging -- 9 taps
//    x Cout/128 column tiles times per element.  That kernel was VALU-bound: 207 VALU + 16
//    transcendental instructions per wave and K tile against 12 matrix instructions, 153 TFLOP/s
//    fp32-equivalent on the 128->128 @512x256 layer; profiles/r02_decode_kernel_stats.md.)
//  * Main loop: gemm_split.hip's two-register-set, counted-vmcnt pipeline (tiles kt+1 and kt+2 in
//    flight in registers while tile kt is multiplied; every staged piece is waited for with an
//    exact vmcnt, written and re-issued in the shadow of the matrix instructions).
//  * 128x128 tile, 8 waves (32x64 wave tiles), 74 KB of LDS.
//  * A workgroup must lie inside one image (pixels per image % 128 == 0: every decode shape).
#include <type_traits>

#include "common.h"

namespace {

typedef t2h_f16x8 f16x8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// tile: CS_BM (128 or 256) x 128, 8 waves as 4 (M) x 2 (N): 32x64 or 64x64 wave tiles
constexpr int CS_BN = 128, CS_WM = 4, CS_WN = 2, CS_NT = 64 * CS_WM * CS_WN;
constexpr int CS_LDS_ROW = 144;                     // bytes per tile row in LDS
constexpr int CS_LB = CS_BN * 8 / CS_NT;            // B pieces per thread and K tile (2)

__device__ __forceinline__ void cs_gload16(u32x4& dst, const void* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
template <int N>
__device__ __forceinline__ void cs_wait(u32x4& v) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N));
}

template <int CS_BM>
__global__ __launch_bounds__(CS_NT, 2) void conv_split_kernel(const t2h_gemm_args p) {
  constexpr int CS_LA = CS_BM * 8 / CS_NT;  // A pieces per thread and K tile (2 or 4)
  constexpr int CS_L = CS_LA + CS_LB;
  static_assert(CS_BM * 8 % CS_NT == 0 && CS_BN * 8 % CS_NT == 0 && CS_BM % 128 == 0, "tile / thread shape");
  constexpr int WM = CS_BM / CS_WM, WN = CS_BN / CS_WN;  // 32x64 or 64x64 wave tile
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int BUF_B = (CS_BM + CS_BN) * CS_LDS_ROW;
  constexpr int O_LD = WN + 4;
  constexpr int OW = WM * O_LD;
  constexpr int EPI_B = OW * 4 * CS_WM * CS_WN;
  constexpr int SMEM_B = 2 * BUF_B > EPI_B ? 2 * BUF_B : EPI_B;
  constexpr int NMMA = 2 * 3 * TM * TN;  // matrix instructions per wave and K tile
  __shared__ __attribute__((aligned(16))) char smem[SMEM_B];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wm0 = (wave / CS_WN) * WM, wn0 = (wave % CS_WN) * WN;
  const int nbx = (p.N + CS_BN - 1) / CS_BN, nby = (p.M + CS_BM - 1) / CS_BM;
  int m0, n0;
  {  // XCD-aware tile mapping (see gemm.hip)
    const int total = nbx * nby, b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3, q = total >> 3, r = total & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int mt = lin / nbx;
    m0 = mt * CS_BM;
    n0 = (lin - mt * nbx) * CS_BN;
  }
  const int nk = p.K / 32, last = nk - 1;
  const int taps = p.K / p.Cin;  // 9 (3x3) or 1 (1x1)
  const int hw = p.Hout * p.Wout;
  const int img = m0 / hw;       // the whole workgroup lies in this image (host-checked)
  const int Hlim = p.Hin << p.ups, Wlim = p.Win << p.ups;
  const int pc = tid & 7;        // this thread's 16-byte piece inside a K tile, A and B alike
  const int ctiles = p.Cin / 32; // K tiles per tap

  // ---- A rows of this thread (output pixels) and B rows (output channels)
  int a_y[CS_LA], a_x[CS_LA];
#pragma unroll
  for (int i = 0; i < CS_LA; ++i) {
    const int m = m0 + (tid >> 3) + (CS_NT / 8) * i;
    const int rem = m - img * hw;
    const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
    a_y[i] = m < p.M ? oy - p.pad : -(1 << 28);
    a_x[i] = ox - p.pad;
  }
  const int64_t arow_b = (int64_t)ctiles * T2H_SPLIT_TILE_B;  // bytes per pixel of the split-row activations
  const char* const a_img = reinterpret_cast<const char*>(p.A) + (int64_t)img * p.Hin * p.Win * arow_b + pc * 16;
  const int64_t brow_b = (int64_t)nk * T2H_SPLIT_TILE_B;
  const char* b_src[CS_LB];
#pragma unroll
  for (int i = 0; i < CS_LB; ++i) {
    const int n = min(n0 + (tid >> 3) + (CS_NT / 8) * i, p.N - 1);  // clamped: extra columns are never stored
    b_src[i] = reinterpret_cast<const char*>(p.B) + (int64_t)n * brow_b + pc * 16;
  }
  // register sets: L pieces + validity bits of the A pieces
  u32x4 rg[2][CS_L];
  unsigned valid[2] = {0u, 0u};

  auto issue_piece = [&](auto setc, int q, int kt) {
    constexpr int S = decltype(setc)::value;
    const int k = min(kt, last);
    if (q < CS_LA) {
      const int tap = k / ctiles, ct = k - tap * ctiles;  // wave-uniform
      const int dy = taps == 9 ? tap / 3 : 0, dx = taps == 9 ? tap - 3 * (tap / 3) : 0;
      const int iy = a_y[q] + dy, ix = a_x[q] + dx;
      const bool ok = (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim;
      const int cy = min(max(iy, 0), Hlim - 1) >> p.ups, cx = min(max(ix, 0), Wlim - 1) >> p.ups;
      cs_gload16(rg[S][q], a_img + ((int64_t)cy * p.Win + cx) * arow_b + ct * T2H_SPLIT_TILE_B);
      valid[S] = (valid[S] & ~(1u << q)) | ((ok ? 1u : 0u) << q);
    } else {
      cs_gload16(rg[S][q], b_src[q - CS_LA] + (int64_t)k * T2H_SPLIT_TILE_B);
    }
  };
  auto put_piece = [&](auto setc, int q, int buf) {
    constexpr int S = decltype(setc)::value;
    char* const base = smem + buf * BUF_B;
    if (q < CS_LA) {  // zero where the tap falls outside the image (the reference pads the activated tensor)
      const bool ok = (valid[S] >> q) & 1u;
      u32x4 v = rg[S][q];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0u;
      *reinterpret_cast<u32x4*>(base + ((tid >> 3) + (CS_NT / 8) * q) * CS_LDS_ROW + pc * 16) = v;
    } else {
      *reinterpret_cast<u32x4*>(base + (CS_BM + (tid >> 3) + (CS_NT / 8) * (q - CS_LA)) * CS_LDS_ROW + pc * 16) =
          rg[S][q];
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

  using set0 = std::integral_constant<int, 0>;
  using set1 = std::integral_constant<int, 1>;
  // prologue: tiles 0 and 1 requested back to back; tile 0 is complete once at most the L younger
  // loads of tile 1 are outstanding.
#pragma unroll
  for (int q = 0; q < CS_L; ++q) issue_piece(set0{}, q, 0);
#pragma unroll
  for (int q = 0; q < CS_L; ++q) issue_piece(set1{}, q, 1);
#pragma unroll
  for (int q = 0; q < CS_L; ++q) {
    cs_wait<CS_L>(rg[0][q]);
    put_piece(set0{}, q, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < CS_L; ++q) issue_piece(set0{}, q, 2);
  __syncthreads();

  constexpr int PA[3] = {1, 0, 0};
  constexpr int PB[3] = {0, 1, 0};
  constexpr int PC[3] = {1, 1, 0};
  auto step = [&](int kt, auto setc) {  // register set S holds tile kt+1
    const int buf = kt & 1;
    const char* Ab = smem + buf * BUF_B + (wm0 + l31) * CS_LDS_ROW + hh * 16;
    const char* Bb = smem + buf * BUF_B + (CS_BM + wn0 + l31) * CS_LDS_ROW + hh * 16;
    int mma = 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f16x8 af[TM][2], bfr[TN][2];
#pragma unroll
      for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          af[ti][pl] = *reinterpret_cast<const f16x8*>(Ab + ti * 32 * CS_LDS_ROW + pl * 64 + u * 32);
#pragma unroll
      for (int tj = 0; tj < TN; ++tj)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          bfr[tj][pl] = *reinterpret_cast<const f16x8*>(Bb + tj * 32 * CS_LDS_ROW + pl * 64 + u * 32);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
          for (int tj = 0; tj < TN; ++tj) {
            acc[PC[t]][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ti][PA[t]], bfr[tj][PB[t]],
                                                                         acc[PC[t]][ti][tj], 0, 0, 0);
            ++mma;
            // one staged piece behind each of the last L matrix instructions but one of the tile
#pragma unroll
            for (int q = 0; q < CS_L; ++q) {
              if (NMMA - 2 * (CS_L - q) + 1 != mma) continue;
              __builtin_amdgcn_sched_barrier(0);
              cs_wait<2 * CS_L - 1>(rg[decltype(setc)::value][q]);
              put_piece(setc, q, buf ^ 1);
              __builtin_amdgcn_sched_barrier(0);
              issue_piece(setc, q, kt + 3);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
    }
    __syncthreads();
  };
  static_assert(NMMA - 2 * CS_L + 1 >= 1, "not enough matrix instructions to pin the staged pieces behind");
  for (int kt = 0; kt < nk; kt += 2) {  // no peeled odd tail: a third copy of step() costs registers
    step(kt, set1{});
    if (kt + 1 < nk) step(kt + 1, set0{});
  }
#pragma unroll
  for (int S = 0; S < 2; ++S)
#pragma unroll
    for (int q = 0; q < CS_L; ++q) cs_wait<0>(rg[S][q]);

  // ---- epilogue: accumulators (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) are
  // transposed through the idle LDS so that every lane owns 8 consecutive columns of a row
  float* const Ot = reinterpret_cast<float*>(smem) + wave * OW;
#pragma unroll
  for (int ti = 0; ti < TM; ++ti)
#pragma unroll
    for (int tj = 0; tj < TN; ++tj) {
      const int col = n0 + wn0 + tj * 32 + l31;
      const float bv = (p.bias && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Ot[(ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * O_LD + tj * 32 + l31] =
            fmaf(acc[1][ti][tj][r], T2H_SPLIT_LO_INV, acc[0][ti][tj][r]) + bv;
    }
  __syncthreads();
  constexpr int CPR = WN / 8;         // 8-column chunks per staged row
  constexpr int NCH = WM * CPR / 64;  // chunks per lane
  static_assert(NCH >= 1 && NCH * 64 == WM * CPR, "epilogue chunking");
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int c = lane + 64 * it;
    const int rl = c / CPR, c8 = (c - rl * CPR) * 8;
    const int row = m0 + wm0 + rl, col = n0 + wn0 + c8;
    if (row >= p.M || col >= p.N) continue;
    f32x4 va = *reinterpret_cast<const f32x4*>(Ot + rl * O_LD + c8);
    f32x4 vb = *reinterpret_cast<const f32x4*>(Ot + rl * O_LD + c8 + 4);
    f32x4 ra = {0.f, 0.f, 0.f, 0.f}, rb = ra;
    if (p.residual) {
      ra = *reinterpret_cast<const f32x4*>(p.residual + (int64_t)row * p.ldr + col);
      rb = *reinterpret_cast<const f32x4*>(p.residual + (int64_t)row * p.ldr + col + 4);
    }
    if (p.res_pre) {
      va += ra;
      vb += rb;
    }
    if (p.epi_act == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        va[e] = fmaxf(va[e], 0.f);
        vb[e] = fmaxf(vb[e], 0.f);
      }
    }
    if (!p.res_pre) {
      va += ra;
      vb += rb;
    }
    *reinterpret_cast<f32x4*>(p.C + (int64_t)row * p.ldc + col) = va;
    *reinterpret_cast<f32x4*>(p.C + (int64_t)row * p.ldc + col + 4) = vb;
    if (p.gn_part_out) {  // final values back into the staging tile for the column sums below
      *reinterpret_cast<f32x4*>(Ot + rl * O_LD + c8) = va;
      *reinterpret_cast<f32x4*>(Ot + rl * O_LD + c8 + 4) = vb;
    }
  }
  // ---- GroupNorm partials of the tensor this kernel produces: per output channel, (sum, sum of
  // squares) of the FINAL values over every 128 rows, in fp64, fixed order: lane j sums column j of its
  // wave tile over its rows, the M-waves of a 128-row chunk are added through LDS, one plain store per
  // (chunk, channel) -- no atomics, bit-reproducible.  The following GroupNorm only reduces
  // rows/128 partials per channel instead of reading the tensor again (the gn_partial pass was 15 %
  // of the decode).
  if (p.gn_part_out) {
    __syncthreads();
    double su = 0.0, sq = 0.0;
#pragma unroll 8
    for (int r = 0; r < WM; ++r) {
      const double v = (double)Ot[r * O_LD + lane];
      su += v;
      sq = fma(v, v, sq);
    }
    __syncthreads();  // every wave is done reading its staging tile: reuse the start of LDS as the table
    double* const red = reinterpret_cast<double*>(smem);  // [wave][64][2]
    red[(wave * 64 + lane) * 2] = su;
    red[(wave * 64 + lane) * 2 + 1] = sq;
    __syncthreads();
    // partials are per 128 rows whatever the tile: 128 / WM consecutive M-waves per chunk; the first
    // M-wave of each chunk adds its group in fixed order and stores
    constexpr int WPC = 128 / WM;  // M-waves per 128-row chunk
    const int wmi0 = wave / CS_WN, wni = wave % CS_WN;
    if (wmi0 % WPC == 0) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int k = 0; k < WPC; ++k) {
        a += red[(((wmi0 + k) * CS_WN + wni) * 64 + lane) * 2];
        b += red[(((wmi0 + k) * CS_WN + wni) * 64 + lane) * 2 + 1];
      }
      const int col = n0 + wni * WN + lane;
      if (col < p.N) {
        const int chunks = hw / 128, chunk = (m0 - img * hw) / 128 + wmi0 / WPC;
        double* dst = p.gn_part_out + (((int64_t)img * chunks + chunk) * 2) * p.N + col;
        dst[0] = a;
        dst[p.N] = b;
      }
    }
  }
}

thread_local int g_force_bm = 0;  // tuning / test hook of the calling thread

}  // namespace

extern "C" int t2h_conv_split_force_tile(int rows) {
  const int old = g_force_bm;
  g_force_bm = (rows == 128 || rows == 256) ? rows : 0;
  return old;
}

extern "C" int t2h_conv_split_f32(const t2h_gemm_args* args, void* stream) {
  T2H_REQUIRE(args != nullptr, "t2h_conv_split_f32: args is NULL");
  t2h_gemm_args a = *args;
  T2H_REQUIRE(a.A && a.B && a.C, "t2h_conv_split_f32: NULL operand");
  T2H_REQUIRE(a.batch <= 1 && !a.b_trans && a.alpha == 1.0f, "t2h_conv_split_f32: plain single problem only");
  T2H_REQUIRE(a.a_mode == 1, "t2h_conv_split_f32: conv geometry required (a 1x1 convolution is K == Cin, pad 0)");
  T2H_REQUIRE(a.M > 0 && a.N > 0 && a.Cin > 0 && a.Cin % 32 == 0 && (a.K == a.Cin || a.K == 9 * a.Cin),
              "t2h_conv_split_f32: bad shape M=%d N=%d K=%d Cin=%d", a.M, a.N, a.K, a.Cin);
  T2H_REQUIRE(a.stride == 1 && (a.ups == 0 || a.ups == 1) && a.pad == (a.K == a.Cin ? 0 : 1),
              "t2h_conv_split_f32: stride-1 'same' / nearest-x2 / 1x1 convolutions only");
  T2H_REQUIRE(a.Hout == (a.Hin << a.ups) && a.Wout == (a.Win << a.ups) && a.M % (a.Hout * a.Wout) == 0,
              "t2h_conv_split_f32: geometry");
  T2H_REQUIRE((a.Hout * a.Wout) % 128 == 0, "t2h_conv_split_f32: pixels per image must be a multiple of 128");
  T2H_REQUIRE(a.N % 8 == 0 && a.ldc % 4 == 0 && (!a.residual || a.ldr % 4 == 0) &&
                  t2h_aligned16(a.A) && t2h_aligned16(a.B) && t2h_aligned16(a.C) &&
                  (!a.residual || t2h_aligned16(a.residual)),
              "t2h_conv_split_f32: N %% 8, leading dimensions %% 4, 16-byte aligned pointers");
  T2H_REQUIRE(a.epi_act == 0 || a.epi_act == 2, "t2h_conv_split_f32: epilogue activation none / ReLU");
  T2H_REQUIRE(a.pro_scale == nullptr && a.pro_shift == nullptr,
              "t2h_conv_split_f32: no prologue tables (apply GroupNorm with t2h_gn_apply_split_f32)");
  // 256-row tiles (64x64 wave tiles: 8 fragment reads per 12 matrix instructions instead of 6 per 6)
  // where they still give every CU a tile; t2h_conv_split_force_tile() overrides (tests, A/B)
  const int nbx = (a.N + CS_BN - 1) / CS_BN;
  const bool big = g_force_bm ? g_force_bm == 256
                              : ((a.Hout * a.Wout) % 256 == 0 && (int64_t)(a.M / 256) * nbx >= 256);
  T2H_REQUIRE(!big || (a.Hout * a.Wout) % 256 == 0, "t2h_conv_split_f32: 256-row tiles need pixels per image %% 256 == 0");
  dim3 block(CS_NT);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (big) hipLaunchKernelGGL(conv_split_kernel<256>, dim3(nbx * ((a.M + 255) / 256)), block, 0, s, a);
  else hipLaunchKernelGGL(conv_split_kernel<128>, dim3(nbx * ((a.M + 127) / 128)), block, 0, s, a);
  T2H_CHECK_LAUNCH("t2h_conv_split_f32");
  return T2H_OK;
}
