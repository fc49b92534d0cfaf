// LayerNorm, GroupNorm statistics -> prologue tables, row softmax.
#include "common.h"

namespace {

// One wave per row; each lane keeps its C/64 values in registers, two passes
// (mean, then centred variance) like torch's rowwise moments.
// SPLIT: the result is written as split rows (two fp16 planes) for the
// split-precision GEMM instead of fp32.
#ifndef T2H_LN_XCD
#define T2H_LN_XCD 1  // (0: rows handed out round-robin, the mapping until round 6 -- the A/B build of tools/build_ln_xcd.sh)
#endif
#ifndef T2H_LN_RPB
#define T2H_LN_RPB 8  // rows (= waves) per workgroup: with the input just written by the previous kernel,
                      // 6.3 us at 8 against 7.3 at 4, 6.6 at 2 / 16 (tools/ln_block_bench.py)
#endif
// SPLIT = 2: the x8 format (common.h: fp16 plane + two e4m3 planes scaled by `x8_scale`).
template <int VPL, int SPLIT = 0>  // float4 vectors per lane: C = 256 * VPL
__global__ __launch_bounds__(64 * T2H_LN_RPB) void layernorm_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        float* __restrict__ y, int rows,
                                                        float eps, int* ovf, float x8_scale = 1.0f) {
  constexpr int C = 256 * VPL;
  const int lane = threadIdx.x & 63;
  // Workgroup -> rows, XCD-aware (round 6).  Workgroup b runs on XCD b % 8; the GEMM that wrote x gives XCD i the i-th
  // eighth of its row blocks (gemm_split.hip / gemm.hip tile mapping).  With XCD i also normalising the i-th eighth of
  // the row blocks a [proj GEMM -> LayerNorm] chain takes 18.1-19.6 instead of 21.1-21.3 us per pair at M = 4096 (41.4
  // vs 43.3 at M = 16384) and the whole B = 8 sampler 1 % less over five interleaved pairs of runs, same bits
  // (profiles/r06_ln_xcd_ab.log).  NOT because the rows are found in that XCD's L2: FETCH_SIZE is the whole tensor under
  // either mapping (no line survives the kernel boundary), and the kernel's own duration barely moves (5.0 vs 5.4 us
  // under the counters) -- what shortens is the hand-over between the two kernels.  Bijective for any workgroup count;
  // a speed choice only.
  int blk = blockIdx.x;
#if T2H_LN_XCD
  {
    const int total = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3, q = total >> 3, r = total & 7;
    blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
#endif
  const int row = blk * T2H_LN_RPB + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * C;
  f32x4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i] = *reinterpret_cast<const f32x4*>(xr + i * 256 + lane * 4);
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wave_sum_dpp(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[i][e] - mean;
      q = fmaf(d, d, q);
    }
  const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) * (1.0f / C) + eps);
  float* yr = y + (int64_t)row * C;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + i * 256 + lane * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + i * 256 + lane * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
    if (!SPLIT) *reinterpret_cast<f32x4*>(yr + i * 256 + lane * 4) = o;
    else v[i] = o;
  }
  if (SPLIT) {
    // lanes 2j, 2j+1 hold columns 8j..8j+3 / 8j+4..8j+7 of every 256-column slab: they
    // swap vectors so that each writes 8 consecutive columns (one 16-byte store per
    // plane) -- the even lane slab i, the odd lane slab i+1
    static_assert(VPL % 2 == 0 || VPL == 1, "pairwise slab exchange");
    if (VPL == 1) {
      if (SPLIT == 2) t2h_store_x8_4(reinterpret_cast<uint16_t*>(y), row, C, lane * 4, v[0], x8_scale, ovf);
      else t2h_store_split4(reinterpret_cast<uint16_t*>(y), row, C, lane * 4, v[0], ovf);
    } else {
#pragma unroll
      for (int i = 0; i < VPL; i += 2) {
        const bool odd = lane & 1;
        f32x4 send, recv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          send[e] = odd ? v[i][e] : v[i + 1][e];
          recv[e] = __shfl_xor(send[e], 1, 64);
        }
        if (SPLIT == 2) {
          // (lanes l, l ^ 2 hold the two halves of a 16-column group of the same slab)
          if (!odd) t2h_store_x8_8<2>(reinterpret_cast<uint16_t*>(y), row, C, i * 256 + lane * 4, v[i], recv, x8_scale, ovf);
          else t2h_store_x8_8<2>(reinterpret_cast<uint16_t*>(y), row, C, (i + 1) * 256 + (lane - 1) * 4, recv, v[i + 1], x8_scale, ovf);
        } else if (!odd) t2h_store_split8(reinterpret_cast<uint16_t*>(y), row, C, i * 256 + lane * 4, v[i], recv, ovf);
        else t2h_store_split8(reinterpret_cast<uint16_t*>(y), row, C, (i + 1) * 256 + (lane - 1) * 4, recv, v[i + 1], ovf);
      }
    }
  }
}

// GroupNorm apply (+ swish) + split in one elementwise pass: fp32 NHWC rows -> split rows, the
// operand format of t2h_conv_split_f32.  One thread per 8 consecutive channels of a pixel.
__global__ __launch_bounds__(256) void gn_apply_split_kernel(const float* __restrict__ x, int ldx,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int tbl_ld,
                                                             int rows_per_img, int C, int act,
                                                             uint16_t* __restrict__ out, int64_t total, int* ovf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c8 = C >> 3;
  const int64_t row = i / c8;
  const int c0 = (int)(i - row * c8) * 8;
  const float* xp = x + row * ldx + c0;
  f32x4 va = *reinterpret_cast<const f32x4*>(xp), vb = *reinterpret_cast<const f32x4*>(xp + 4);
  if (scale) {
    const int64_t o = (row / rows_per_img) * tbl_ld + c0;
    const f32x4 sa = *reinterpret_cast<const f32x4*>(scale + o), sb = *reinterpret_cast<const f32x4*>(scale + o + 4);
    const f32x4 ta = *reinterpret_cast<const f32x4*>(shift + o), tb = *reinterpret_cast<const f32x4*>(shift + o + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      va[e] = fmaf(va[e], sa[e], ta[e]);
      vb[e] = fmaf(vb[e], sb[e], tb[e]);
      if (act == 1) {  // swish, same expression as the fp32 kernel's operand prologue (gemm.hip)
        va[e] = va[e] / (1.0f + fast_exp(fminf(-va[e], 87.0f)));
        vb[e] = vb[e] / (1.0f + fast_exp(fminf(-vb[e], 87.0f)));
      }
    }
  }
  t2h_store_split8(out, row, C, c0, va, vb, ovf);
}

// ---- GroupNorm statistics.  Grid (chunks, n_img).  Thread t owns the channel
// quad (t % (C/4)) and walks pixels with stride 256/(C/4); per-channel partial
// sums in fp64 (fp64 VALU is cheap on CDNA and removes the E[x^2]-E[x]^2
// cancellation worry), block-reduced per channel, written as partials.
constexpr int GN_PIX_PER_BLOCK = 1024;

__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, int ldx,
                                                         int HW, int C, int chunks,
                                                         double* __restrict__ part) {
  __shared__ double red[2][256][4];
  const int tid = threadIdx.x;
  const int q_per_pix = C >> 2;           // float4 per pixel
  const int pix_stride = 256 / q_per_pix;  // pixels covered per pass
  const int cq = tid % q_per_pix, p_lane = tid / q_per_pix;
  const int img = blockIdx.y, chunk = blockIdx.x;
  const int p_begin = chunk * GN_PIX_PER_BLOCK;
  const int p_end = min(HW, p_begin + GN_PIX_PER_BLOCK);
  double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  const float* xb = x + (int64_t)img * HW * ldx + cq * 4;
  for (int p = p_begin + p_lane; p < p_end; p += pix_stride) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (int64_t)p * ldx);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double d = (double)v[e];
      s[e] += d;
      ss[e] = fma(d, d, ss[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[0][tid][e] = s[e];
    red[1][tid][e] = ss[e];
  }
  __syncthreads();
  if (tid < q_per_pix) {
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    for (int r = 0; r < pix_stride; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] += red[0][r * q_per_pix + tid][e];
        b[e] += red[1][r * q_per_pix + tid][e];
      }
    double* out = part + (((int64_t)img * chunks + chunk) * 2) * C + tid * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      out[e] = a[e];
      out[C + e] = b[e];
    }
  }
}

// Grid (groups, n_img), 256 threads: a workgroup owns ONE GroupNorm group of one image -- its cpg = C /
// groups channels (4 .. 32) -- and spreads the `chunks` per-tile partials of those channels over
// 256 / cpg slices of consecutive chunks (a conv epilogue leaves up to 4096 of them per image at
// 1024x512; one workgroup per image walking them alone took 60-100 us per GroupNorm there).  Every slice
// sums in chunk order, the slices are combined in slice order: a fixed summation tree, in fp64.
constexpr int GNF_THREADS = 256;
__global__ __launch_bounds__(GNF_THREADS) void gn_finalize_kernel(const double* __restrict__ part, int chunks, int HW,
                                                                 int C, int groups, float eps,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta,
                                                                 float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ double cs[GNF_THREADS], css[GNF_THREADS];
  __shared__ float g_stat[2];
  const int g = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / groups;              // host: cpg <= 256
  const int slices = GNF_THREADS / cpg;    // whole slices; when cpg does not divide 256 (cpg = 3, 6, 12: ch = 96) the
                                           // threads beyond slices * cpg stay idle and add zeros.  Tested for those
                                           // counts (tests/test_gpu_kernels.py); the pass over the tensor itself,
                                           // t2h_groupnorm_tables_f32 below, still serves C / 4 | 256 only
  const int ch = tid % cpg, sl = tid / cpg;
  const int per = (chunks + slices - 1) / slices;
  const int k0 = sl < slices ? sl * per : chunks, k1 = min(chunks, k0 + per);
  const double* p = part + ((int64_t)img * chunks * 2) * C + g * cpg + ch;
  double s = 0, ss = 0;
  int k = k0;
  for (; k + 4 <= k1; k += 4) {  // 8 independent loads in flight; the additions stay in chunk order
    double a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = p[(int64_t)(k + j) * 2 * C];
      b[j] = p[(int64_t)(k + j) * 2 * C + C];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s += a[j];
      ss += b[j];
    }
  }
  for (; k < k1; ++k) {
    s += p[(int64_t)k * 2 * C];
    ss += p[(int64_t)k * 2 * C + C];
  }
  cs[tid] = s;
  css[tid] = ss;
  __syncthreads();
  if (tid == 0) {
    double a = 0, b = 0;
    for (int i = 0; i < GNF_THREADS; ++i) {  // slice-major, channel-minor: a fixed order
      a += cs[i];
      b += css[i];
    }
    const double n = (double)HW * cpg;
    const double mean = a / n;
    double var = b / n - mean * mean;
    if (var < 0) var = 0;
    g_stat[0] = (float)mean;
    g_stat[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  if (tid < cpg) {
    const int c = g * cpg + tid;
    const float sc = g_stat[1] * gamma[c];
    scale[(int64_t)img * C + c] = sc;
    shift[(int64_t)img * C + c] = fmaf(-g_stat[0], sc, beta[c]);
  }
}

// In-place softmax of each row; one wave per row, the row (n <= 64*MAXV) stays
// in registers between the max, exp-sum and normalise passes.
template <int MAXV>
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, int rows, int n,
                                                           int ld) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* xr = x + (int64_t)row * ld;
  float v[MAXV];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 64 + lane;
    v[i] = c < n ? xr[c] : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    v[i] = expf(v[i] - mx);
    s += v[i];
  }
  const float inv = 1.0f / wave_sum(s);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 64 + lane;
    if (c < n) xr[c] = v[i] * inv;
  }
}

}  // namespace

extern "C" int t2h_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y,
                                 int32_t rows, int32_t C, float eps, void* stream) {
  T2H_REQUIRE(x && gamma && beta && y, "t2h_layernorm_f32: NULL pointer");
  T2H_REQUIRE(rows > 0, "t2h_layernorm_f32: rows=%d", rows);
  T2H_REQUIRE(t2h_aligned16(x) && t2h_aligned16(y) && t2h_aligned16(gamma) && t2h_aligned16(beta),
              "t2h_layernorm_f32: 16-byte alignment");
  dim3 grid((rows + T2H_LN_RPB - 1) / T2H_LN_RPB), block(64 * T2H_LN_RPB);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (C == 512) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, x, gamma, beta, y, rows, eps, static_cast<int*>(nullptr), 1.0f);
  else if (C == 256) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, x, gamma, beta, y, rows, eps, static_cast<int*>(nullptr), 1.0f);
  else if (C == 1024) hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, x, gamma, beta, y, rows, eps, static_cast<int*>(nullptr), 1.0f);
  else {
    t2h_set_error("t2h_layernorm_f32: C=%d unsupported (256/512/1024)", C);
    return T2H_ERR_UNSUPPORTED;
  }
  T2H_CHECK_LAUNCH("t2h_layernorm_f32");
  return T2H_OK;
}

extern "C" int t2h_layernorm_split_f32(const float* x, const float* gamma, const float* beta,
                                       uint16_t* y_split, int32_t rows, int32_t C, float eps,
                                       int32_t* overflow_flag, void* stream) {
  T2H_REQUIRE(x && gamma && beta && y_split, "t2h_layernorm_split_f32: NULL pointer");
  T2H_REQUIRE(rows > 0, "t2h_layernorm_split_f32: rows=%d", rows);
  T2H_REQUIRE(t2h_aligned16(x) && t2h_aligned16(y_split) && t2h_aligned16(gamma) && t2h_aligned16(beta),
              "t2h_layernorm_split_f32: 16-byte alignment");
  dim3 grid((rows + T2H_LN_RPB - 1) / T2H_LN_RPB), block(64 * T2H_LN_RPB);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* y = reinterpret_cast<float*>(y_split);
  int* ovf = overflow_flag;
  T2H_REQUIRE(ovf != nullptr, "t2h_layernorm_split_f32: overflow_flag is NULL");
  if (C == 512) hipLaunchKernelGGL((layernorm_kernel<2, 1>), grid, block, 0, s, x, gamma, beta, y, rows, eps, ovf, 1.0f);
  else if (C == 256) hipLaunchKernelGGL((layernorm_kernel<1, 1>), grid, block, 0, s, x, gamma, beta, y, rows, eps, ovf, 1.0f);
  else if (C == 1024) hipLaunchKernelGGL((layernorm_kernel<4, 1>), grid, block, 0, s, x, gamma, beta, y, rows, eps, ovf, 1.0f);
  else {
    t2h_set_error("t2h_layernorm_split_f32: C=%d unsupported (256/512/1024)", C);
    return T2H_ERR_UNSUPPORTED;
  }
  T2H_CHECK_LAUNCH("t2h_layernorm_split_f32");
  return T2H_OK;
}

extern "C" int t2h_layernorm_x8_f32(const float* x, const float* gamma, const float* beta, uint16_t* y_x8, int32_t rows,
                                    int32_t C, float eps, float scale, int32_t* overflow_flag, void* stream) {
  T2H_REQUIRE(x && gamma && beta && y_x8 && overflow_flag, "t2h_layernorm_x8_f32: NULL pointer");
  T2H_REQUIRE(rows > 0 && scale > 0.f, "t2h_layernorm_x8_f32: rows=%d scale=%g", rows, (double)scale);
  T2H_REQUIRE(t2h_aligned16(x) && t2h_aligned16(y_x8) && t2h_aligned16(gamma) && t2h_aligned16(beta),
              "t2h_layernorm_x8_f32: 16-byte alignment");
  dim3 grid((rows + T2H_LN_RPB - 1) / T2H_LN_RPB), block(64 * T2H_LN_RPB);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* y = reinterpret_cast<float*>(y_x8);
  int* ovf = overflow_flag;
  if (C == 512) hipLaunchKernelGGL((layernorm_kernel<2, 2>), grid, block, 0, s, x, gamma, beta, y, rows, eps, ovf, scale);
  else if (C == 256) hipLaunchKernelGGL((layernorm_kernel<1, 2>), grid, block, 0, s, x, gamma, beta, y, rows, eps, ovf, scale);
  else if (C == 1024) hipLaunchKernelGGL((layernorm_kernel<4, 2>), grid, block, 0, s, x, gamma, beta, y, rows, eps, ovf, scale);
  else {
    t2h_set_error("t2h_layernorm_x8_f32: C=%d unsupported (256/512/1024)", C);
    return T2H_ERR_UNSUPPORTED;
  }
  T2H_CHECK_LAUNCH("t2h_layernorm_x8_f32");
  return T2H_OK;
}

static int gn_chunks(int HW) { return (HW + GN_PIX_PER_BLOCK - 1) / GN_PIX_PER_BLOCK; }

extern "C" int64_t t2h_groupnorm_workspace_bytes(int32_t n_img, int32_t HW, int32_t C) {
  return (int64_t)n_img * gn_chunks(HW) * 2 * C * (int64_t)sizeof(double);
}

extern "C" int t2h_groupnorm_tables_f32(const float* x, int32_t ldx, const float* gamma,
                                        const float* beta, float* scale, float* shift,
                                        int32_t n_img, int32_t HW, int32_t C, int32_t groups,
                                        float eps, void* workspace, void* stream) {
  T2H_REQUIRE(x && gamma && beta && scale && shift && workspace, "t2h_groupnorm_tables_f32: NULL pointer");
  T2H_REQUIRE(n_img > 0 && HW > 0, "t2h_groupnorm_tables_f32: empty problem");
  T2H_REQUIRE(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && C / 4 <= 256,
              "t2h_groupnorm_tables_f32: C=%d unsupported", C);
  T2H_REQUIRE(groups > 0 && groups <= 64 && C % groups == 0 && C / groups <= GNF_THREADS,
              "t2h_groupnorm_tables_f32: groups=%d (C / groups must be <= 256)", groups);
  T2H_REQUIRE(ldx % 4 == 0 && t2h_aligned16(x), "t2h_groupnorm_tables_f32: alignment");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int chunks = gn_chunks(HW);
  hipLaunchKernelGGL(gn_partial_kernel, dim3(chunks, n_img), dim3(256), 0, s, x, ldx, HW, C, chunks,
                     static_cast<double*>(workspace));
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, n_img), dim3(GNF_THREADS), 0, s,
                     static_cast<const double*>(workspace), chunks, HW, C, groups, eps, gamma, beta,
                     scale, shift);
  T2H_CHECK_LAUNCH("t2h_groupnorm_tables_f32");
  return T2H_OK;
}

extern "C" int t2h_groupnorm_finalize_f32(const double* part, int32_t chunks, const float* gamma, const float* beta,
                                          float* scale, float* shift, int32_t n_img, int32_t HW, int32_t C,
                                          int32_t groups, float eps, void* stream) {
  T2H_REQUIRE(part && gamma && beta && scale && shift, "t2h_groupnorm_finalize_f32: NULL pointer");
  T2H_REQUIRE(n_img > 0 && HW > 0 && chunks > 0 && C > 0 && C <= 1024 && groups > 0 && groups <= 64 && C % groups == 0 &&
                  C / groups <= GNF_THREADS,
              "t2h_groupnorm_finalize_f32: bad shape (C <= 1024, groups <= 64, C / groups <= 256)");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, n_img), dim3(GNF_THREADS), 0, static_cast<hipStream_t>(stream),
                     part, chunks, HW, C, groups, eps, gamma, beta, scale, shift);
  T2H_CHECK_LAUNCH("t2h_groupnorm_finalize_f32");
  return T2H_OK;
}

extern "C" int t2h_softmax_rows_f32(float* x, int32_t rows, int32_t n, int32_t ld, void* stream) {
  T2H_REQUIRE(x, "t2h_softmax_rows_f32: NULL pointer");
  T2H_REQUIRE(rows > 0 && n > 0 && ld >= n, "t2h_softmax_rows_f32: bad shape");
  dim3 grid((rows + 3) / 4), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n <= 512) hipLaunchKernelGGL(softmax_rows_kernel<8>, grid, block, 0, s, x, rows, n, ld);
  else if (n <= 2048) hipLaunchKernelGGL(softmax_rows_kernel<32>, grid, block, 0, s, x, rows, n, ld);
  else if (n <= 8192) hipLaunchKernelGGL(softmax_rows_kernel<128>, grid, block, 0, s, x, rows, n, ld);
  else {
    t2h_set_error("t2h_softmax_rows_f32: n=%d > 8192 unsupported", n);
    return T2H_ERR_UNSUPPORTED;
  }
  T2H_CHECK_LAUNCH("t2h_softmax_rows_f32");
  return T2H_OK;
}

extern "C" int t2h_gn_apply_split_f32(const float* x, int32_t ldx, const float* scale, const float* shift,
                                      int32_t tbl_ld, uint16_t* out_split, int64_t rows, int32_t rows_per_img,
                                      int32_t C, int32_t act, int32_t* overflow_flag, void* stream) {
  T2H_REQUIRE(x && out_split && rows > 0 && C > 0 && C % 32 == 0 && ldx % 4 == 0 && t2h_aligned16(x) &&
                  t2h_aligned16(out_split),
              "t2h_gn_apply_split_f32: bad arguments (C %% 32, ldx %% 4, 16-byte alignment)");
  T2H_REQUIRE((scale == nullptr) == (shift == nullptr) && (act == 0 || act == 1),
              "t2h_gn_apply_split_f32: scale / shift come together; act none / swish");
  if (scale)
    T2H_REQUIRE(rows_per_img > 0 && tbl_ld % 4 == 0 && t2h_aligned16(scale) && t2h_aligned16(shift),
                "t2h_gn_apply_split_f32: tables");
  int* ovf = overflow_flag;
  T2H_REQUIRE(ovf != nullptr, "t2h_gn_apply_split_f32: overflow_flag is NULL");
  const int64_t total = rows * (C / 8);
  hipLaunchKernelGGL(gn_apply_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, ldx, scale, shift, tbl_ld, rows_per_img > 0 ? rows_per_img : 1,
                     C, act, out_split, total, ovf);
  T2H_CHECK_LAUNCH("t2h_gn_apply_split_f32");
  return T2H_OK;
}
