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
//    arithmetic (x = h + l / 2048) and written to LDS as [halo pixel][2 planes][32] fp16, 144-byte
//    pixel stride (16 consecutive pixels of a halo row hit 16 different 16-byte bank groups).  The
//    nine taps are then nine K tiles whose A fragments are the SAME LDS image read at a shifted
//    pixel: a(.) and the split are computed 1.27x per element (the halo overlap) instead of 9 x
//    Cout / 128 times (the first conv_split version, VALU-bound) or in a pass of their own.
//  * Weights stream as in conv_split.hip: one [128 rows][32 channels of one tap] tile per K tile,
//    global -> registers -> LDS, double buffered, requested one K tile ahead.
//  * The halo of the NEXT channel group is requested, converted and written to the second halo buffer
//    while the nine taps of the current group are multiplied: three 8-channel pieces per thread,
//    requested two taps before they are converted; the two waves of a SIMD convert in alternate taps, at the top of
//    the tap, so that one wave's conversion runs in the shadow of the other's matrix instructions.
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

constexpr int CH_T = 16;                          // output tile edge (pixels)
constexpr int CH_HW = CH_T + 2;                   // halo edge
constexpr int CH_HP = CH_HW * CH_HW;              // halo pixels (324)
constexpr int CH_BM = CH_T * CH_T, CH_BN = 128;   // GEMM tile
constexpr int CH_WM = 4, CH_WN = 2, CH_NT = 64 * CH_WM * CH_WN;
constexpr int CH_ROW = 144;                       // LDS bytes per halo pixel / weight row: 128 + 16
constexpr int CH_HALO_B = CH_HP * CH_ROW;         // 46656
constexpr int CH_BT_B = CH_BN * CH_ROW;           // 18432
constexpr int CH_LOOP_B = 2 * CH_HALO_B + 2 * CH_BT_B;
constexpr int CH_PIECES = CH_HP * 4;              // 8-channel pieces of a halo (1296)
constexpr int CH_PJ = (CH_PIECES + CH_NT - 1) / CH_NT;  // per thread (3; the third only for 272 threads)
static_assert(CH_PJ == 3, "taps 0 .. 5 request, 2 .. 7 convert");

struct ch_piece {
  f32x4 a, b;
};

template <int PRO>  // 0: plain split; 2: GroupNorm tables + swish
__global__ __launch_bounds__(CH_NT, 2) void conv_halo_kernel(const t2h_gemm_args p, int* ovf) {
  constexpr int WM = CH_BM / CH_WM, WN = CH_BN / CH_WN;  // 64 x 64 wave tile
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int O_LD = WN + 4;
  constexpr int OW = WM * O_LD;
  constexpr int EPI_B = OW * 4 * CH_WM * CH_WN;
  // (the pieces beyond the halo -- third piece of threads 272..511 -- are written to a scratch area behind the loop
  // buffers instead of being predicated off: the tap bodies stay free of branches)
  constexpr int DUMMY_B = (CH_PJ * CH_NT - CH_PIECES) * 16 + 64 + 16;
  constexpr int SMEM_B = EPI_B > CH_LOOP_B + DUMMY_B ? EPI_B : CH_LOOP_B + DUMMY_B;
  static_assert(SMEM_B <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) char smem[SMEM_B];
  char* const halo = smem;
  char* const btile = smem + 2 * CH_HALO_B;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wmi = wave / CH_WN, wni = wave % CH_WN;
  const int wm0 = wmi * WM, wn0 = wni * WN;
  const int tiles_x = p.Wout / CH_T, tpi = tiles_x * (p.Hout / CH_T);
  const int nbx = (p.N + CH_BN - 1) / CH_BN, nby = p.M / CH_BM;
  int mt, n0;
  {  // XCD-aware tile mapping (see gemm.hip): consecutive tiles of an image stay on one XCD's L2
    const int total = nbx * nby, b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3, q = total >> 3, r = total & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    mt = lin / nbx;
    n0 = (lin - mt * nbx) * CH_BN;
  }
  const int img = mt / tpi, trem = mt - img * tpi;
  const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
  const int y0 = ty * CH_T, x0 = tx * CH_T;  // the tile's first output pixel
  const int G = p.Cin / 32;                  // channel groups; K tile (tap t, group g) of the packed weights is t * G + g
  const int phase = __builtin_amdgcn_readfirstlane(wave) >> 2;  // wave-uniform: scalar branches
  const int c8 = tid & 3;                    // this thread's 8-channel piece inside a group, every j

  // ---- this thread's halo pieces: source pixel (clamped; nearest-x2: >> ups), inside-the-image bit, LDS offset
  const float* a_src[CH_PJ];
  bool a_in[CH_PJ];
  int a_dst[CH_PJ], a_sel[CH_PJ];  // LDS offset in halo buffer 0; what selecting buffer 1 adds (0 for the scratch slots)
#pragma unroll
  for (int j = 0; j < CH_PJ; ++j) {
    const int pi = tid + CH_NT * j;
    const int hp = min(pi >> 2, CH_HP - 1);
    const int hy = hp / CH_HW, hx = hp - hy * CH_HW;
    const int Y = y0 + hy - 1, X = x0 + hx - 1;  // in the convolution's input geometry (= output geometry)
    a_in[j] = (unsigned)Y < (unsigned)p.Hout && (unsigned)X < (unsigned)p.Wout;
    const int sy = min(max(Y, 0), p.Hout - 1) >> p.ups, sx = min(max(X, 0), p.Wout - 1) >> p.ups;
    a_src[j] = p.A + ((int64_t)(img * p.Hin + sy) * p.Win + sx) * p.lda + c8 * 8;
    a_dst[j] = pi < CH_PIECES ? hp * CH_ROW + c8 * 16 : CH_LOOP_B + (pi - CH_PIECES) * 16;
    a_sel[j] = pi < CH_PIECES ? CH_HALO_B : 0;
  }
  const float* const t_scale = PRO ? p.pro_scale + (int64_t)img * p.pro_ld + c8 * 8 : nullptr;
  const float* const t_shift = PRO ? p.pro_shift + (int64_t)img * p.pro_ld + c8 * 8 : nullptr;
  float amax = 0.f;  // largest |value| this thread splits: one overflow check at the end

  const int pc = tid & 7;  // 16-byte piece of a weight row's K tile
  const int64_t brow_b = (int64_t)9 * G * T2H_SPLIT_TILE_B;
  const char* b_src[2];
  int b_dst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = min(n0 + (tid >> 3) + (CH_NT / 8) * i, p.N - 1);  // clamped: extra columns are never stored
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
  // a(.) + split of one piece (8 channels) into halo buffer `buf`, two channels per instruction (v_pk_*_f32).
  // swish as x * rcp(1 + exp2(-x log2 e)): v_exp_f32 / v_rcp_f32 directly (the elementwise pass and the exact-fp32
  // kernels use a corrected exp and an IEEE division: 25 instead of 9 VALU instructions per element; the difference is
  // a few ulp of the activation).  Outside the image the value is multiplied by 0 (the reference pads the ACTIVATED
  // tensor with zeros).
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto put_piece = [&](int j, ch_piece v, const f32x4 (&sc)[2], const f32x4 (&sh)[2], int buf) {
    const float m = a_in[j] ? 1.0f : 0.0f;
    f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      f32x2 x = {e < 4 ? v.a[e] : v.b[e - 4], e < 4 ? v.a[e + 1] : v.b[e - 3]};
      if (PRO) {
        const f32x2 s2 = {sc[e >> 2][e & 3], sc[e >> 2][(e & 3) + 1]}, b2 = {sh[e >> 2][e & 3], sh[e >> 2][(e & 3) + 1]};
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
    char* const d = smem + a_dst[j] + buf * a_sel[j];
    *reinterpret_cast<f16x8*>(d) = h;
    *reinterpret_cast<f16x8*>(d + T2H_SPLIT_PLANE_B) = l;
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
  const int a_lane = ((wmi * 4 + (l31 >> 4)) * CH_HW + (l31 & 15)) * CH_ROW + hh * 16;
  const int b_lane = (wn0 + l31) * CH_ROW + hh * 16;
  constexpr int PA[3] = {1, 0, 0};
  constexpr int PB[3] = {0, 1, 0};
  constexpr int PC[3] = {1, 1, 0};

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
      // before that wave's matrix instructions, while the other wave of the SIMD -- which converts nothing in this
      // tap -- already issues its own: the conversion's VALU work runs in the shadow of the partner's matrix
      // instructions instead of leaving the matrix pipe idle (both waves run the same tap between two barriers).
      if constexpr (t <= 5)
        if (phase == (t & 1)) pv[t / 2] = load_piece(t / 2, gn);
      // (the requests stay at the top of the tap: left alone, the scheduler sinks them to just before their use at
      // the end of the tap, a whole L2 round trip in front of the barrier)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (t >= 2 && t <= 7)
        if (phase == (t & 1)) put_piece((t - 2) / 2, pv[(t - 2) / 2], sc, sh, nbuf);

      const char* Ab = hcur + ((t / 3) * CH_HW + (t % 3)) * CH_ROW;
      const char* Bb = btile + (kt & 1) * CH_BT_B + b_lane;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f16x8 af[TM][2], bfr[TN][2];
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
            af[ti][pl] = *reinterpret_cast<const f16x8*>(Ab + ti * 2 * CH_HW * CH_ROW + pl * 64 + u * 32);
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

  // ---- epilogue (conv_split.hip's): accumulators (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
  // transposed through the idle LDS so that every lane owns 8 consecutive channels of a pixel
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
  const int64_t img_row0 = (int64_t)img * p.Hout * p.Wout;
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int c = lane + 64 * it;
    const int rl = c / CPR, cc = (c - rl * CPR) * 8;
    // GEMM row wm0 + rl of the tile = pixel (wmi * 4 + rl / 16, rl % 16)
    const int64_t row = img_row0 + (int64_t)(y0 + wmi * 4 + (rl >> 4)) * p.Wout + x0 + (rl & 15);
    const int col = n0 + wn0 + cc;
    if (col >= p.N) continue;
    f32x4 va = *reinterpret_cast<const f32x4*>(Ot + rl * O_LD + cc);
    f32x4 vb = *reinterpret_cast<const f32x4*>(Ot + rl * O_LD + cc + 4);
    if (p.residual) {
      va += *reinterpret_cast<const f32x4*>(p.residual + row * p.ldr + col);
      vb += *reinterpret_cast<const f32x4*>(p.residual + row * p.ldr + col + 4);
    }
    *reinterpret_cast<f32x4*>(p.C + row * p.ldc + col) = va;
    *reinterpret_cast<f32x4*>(p.C + row * p.ldc + col + 4) = vb;
    if (p.gn_part_out) {  // final values back into the staging tile for the channel sums below
      *reinterpret_cast<f32x4*>(Ot + rl * O_LD + cc) = va;
      *reinterpret_cast<f32x4*>(Ot + rl * O_LD + cc + 4) = vb;
    }
  }
  // ---- GroupNorm partials of the produced tensor (conv_split.hip): lane j sums channel j of its wave tile over the
  // wave's 64 pixels in fp64, the two M-waves of a 128-pixel chunk are added through LDS in wave order, one plain
  // store per (chunk, channel): no atomics, bit-reproducible.  Chunk = 2 * (tile inside the image) + (wmi / 2).
  if (p.gn_part_out) {
    __syncthreads();
    double su = 0.0, sq = 0.0;
#pragma unroll 8
    for (int r = 0; r < WM; ++r) {
      const double v = (double)Ot[r * O_LD + lane];
      su += v;
      sq = fma(v, v, sq);
    }
    __syncthreads();  // every wave is done reading its staging tile: the start of LDS becomes the table
    double* const red = reinterpret_cast<double*>(smem);  // [wave][64][2]
    red[(wave * 64 + lane) * 2] = su;
    red[(wave * 64 + lane) * 2 + 1] = sq;
    __syncthreads();
    constexpr int WPC = 128 / WM;  // M-waves per 128-pixel chunk (2)
    if (wmi % WPC == 0) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int k = 0; k < WPC; ++k) {
        a += red[(((wmi + k) * CH_WN + wni) * 64 + lane) * 2];
        b += red[(((wmi + k) * CH_WN + wni) * 64 + lane) * 2 + 1];
      }
      const int col = n0 + wni * WN + lane;
      if (col < p.N) {
        const int chunks = tpi * (CH_BM / 128), chunk = trem * (CH_BM / 128) + wmi / WPC;
        double* dst = p.gn_part_out + (((int64_t)img * chunks + chunk) * 2) * p.N + col;
        dst[0] = a;
        dst[p.N] = b;
      }
    }
  }
}

}  // namespace

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
  if (pro) hipLaunchKernelGGL(conv_halo_kernel<2>, grid, block, 0, s, a, overflow_flag);
  else hipLaunchKernelGGL(conv_halo_kernel<0>, grid, block, 0, s, a, overflow_flag);
  T2H_CHECK_LAUNCH("t2h_conv_halo_f32");
  return T2H_OK;
}
