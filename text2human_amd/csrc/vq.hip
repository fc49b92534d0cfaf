// Quantizer-side kernels: codebook L2-argmin (LDS-staged, wave-reduced),
// texture-routed codebook gathers (top: plain rows, bottom: 2x2 fold) and the
// texture-routed index-prediction head + argmax.
#include "common.h"

namespace {

// d(z, e_j) = sum z^2 + sum e_j^2 - 2 z.e_j, argmin_j with the first minimum
// winning -- the expanded form of VectorQuantizer.forward (vqgan_arch.py:88-92).
// Workgroup: 32 rows of z (8 per wave).  The codebook streams through LDS in
// chunks of 256 codes ([256][D+1] floats, padded so that lane j reading code j
// is bank-conflict free); each lane scores 4 codes per chunk per row, minima
// are combined with a 64-lane (value, index) butterfly.
template <int D>
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ z,
                                                        const float* __restrict__ cb,
                                                        int64_t* __restrict__ idx, int n, int n_e) {
  constexpr int ROWS = 32, CHUNK = 256, LD = D + 1;
  __shared__ float zs[ROWS][D];
  __shared__ float es[CHUNK * LD];
  __shared__ float ee[CHUNK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * ROWS;
  for (int i = tid; i < ROWS * D; i += 256) {
    const int r = i / D, k = i - r * D;
    zs[r][k] = (r0 + r < n) ? z[(int64_t)(r0 + r) * D + k] : 0.f;
  }
  float best[8];
  int best_j[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    best[r] = INFINITY;
    best_j[r] = 0x7fffffff;
  }
  for (int c0 = 0; c0 < n_e; c0 += CHUNK) {
    __syncthreads();
    {
      const int code = c0 + tid;
      float s = 0.f;
      for (int k = 0; k < D; ++k) {
        const float v = code < n_e ? cb[(int64_t)code * D + k] : 0.f;
        es[tid * LD + k] = v;
        s = fmaf(v, v, s);
      }
      ee[tid] = s;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float* zr = zs[wave * 8 + r];
      float zz = 0.f;
      for (int k = 0; k < D; ++k) zz = fmaf(zr[k], zr[k], zz);
#pragma unroll
      for (int c = 0; c < CHUNK / 64; ++c) {
        const int jl = c * 64 + lane;
        const float* er = es + jl * LD;
        float dot = 0.f;
        for (int k = 0; k < D; ++k) dot = fmaf(zr[k], er[k], dot);
        const float d = (zz + ee[jl]) - 2.0f * dot;
        const int j = c0 + jl;
        if (j < n_e && d < best[r]) {
          best[r] = d;
          best_j[r] = j;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float b = best[r];
    int bj = best_j[r];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(b, o, 64);
      const int oj = __shfl_xor(bj, o, 64);
      if (ob < b || (ob == b && oj < bj)) {
        b = ob;
        bj = oj;
      }
    }
    const int row = r0 + wave * 8 + r;
    if (lane == 0 && row < n) idx[row] = bj;
  }
}

// Texture-routed L2-argmin of the encode side: VectorQuantizerTexture.forward
// (vqgan_arch.py:232-268) and, with fold, VectorQuantizerSpatialTextureAware.forward
// (:392-443).  One wave per latent row; the row is quantised with the codebook of its own
// texture id (expanded distance sum z^2 + sum e^2 - 2 z.e, first minimum wins), all other
// heads get -1.  EPL float4 per lane: D = 256 * EPL.  With fold the row is the 2x2 patch
// (i, j) of an NHWC map in F.unfold layout [c, kh, kw]: lane element c*4 + (2 kh + kw).
template <int EPL>
__global__ __launch_bounds__(256) void vq_argmin_tex_kernel(const float* __restrict__ z,
                                                            const float* __restrict__ books,
                                                            const int64_t* __restrict__ tex,
                                                            int64_t* __restrict__ idx_lists, int n, int n_books,
                                                            int n_e, int fh, int fw) {
  constexpr int D = 256 * EPL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= n) return;
  const int t = (int)tex[row];
  for (int hd = lane; hd < n_books; hd += 64)
    if (hd != t) idx_lists[(int64_t)hd * n + row] = -1;
  if (t < 0 || t >= n_books) return;
  f32x4 zv[EPL];
  if (fh > 0) {
    const int C = D / 4;
    const int b = row / (fh * fw), rem = row - b * fh * fw;
    const int i = rem / fw, j = rem - i * fw;
#pragma unroll
    for (int m = 0; m < EPL; ++m) {
      const int c = m * 64 + lane;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        zv[m][e] = z[(((int64_t)b * 2 * fh + 2 * i + (e >> 1)) * 2 * fw + 2 * j + (e & 1)) * C + c];
    }
  } else {
#pragma unroll
    for (int m = 0; m < EPL; ++m) zv[m] = *reinterpret_cast<const f32x4*>(z + (int64_t)row * D + (m * 64 + lane) * 4);
  }
  float zz = 0.f;
#pragma unroll
  for (int m = 0; m < EPL; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) zz = fmaf(zv[m][e], zv[m][e], zz);
  zz = wave_sum(zz);
  const float* book = books + (int64_t)t * n_e * D;
  float best = INFINITY;
  int best_j = 0;
  for (int j0 = 0; j0 < n_e; j0 += 4) {
    float dot[4], ee[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* er = book + (int64_t)min(j0 + u, n_e - 1) * D;
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int m = 0; m < EPL; ++m) {
        const f32x4 ev = *reinterpret_cast<const f32x4*>(er + (m * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a = fmaf(zv[m][e], ev[e], a);
          q = fmaf(ev[e], ev[e], q);
        }
      }
      dot[u] = a;
      ee[u] = q;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float d = (zz + wave_sum(ee[u])) - 2.0f * wave_sum(dot[u]);
      if (j0 + u < n_e && d < best) {
        best = d;
        best_j = j0 + u;
      }
    }
  }
  if (lane == 0) idx_lists[(int64_t)t * n + row] = best_j;
}

__global__ void gather_tex_kernel(const int64_t* __restrict__ idx_lists, const int64_t* __restrict__ tex,
                                  const float* __restrict__ books, float* __restrict__ out, int n,
                                  int n_e, int e_dim) {
  const int row = blockIdx.x;
  const int t = (int)tex[row];
  const int64_t code = idx_lists[(int64_t)t * n + row];
  const float* src = books + ((int64_t)t * n_e + code) * e_dim;
  float* dst = out + (int64_t)row * e_dim;
  for (int i = threadIdx.x * 4; i < e_dim; i += blockDim.x * 4)
    *reinterpret_cast<f32x4*>(dst + i) = *reinterpret_cast<const f32x4*>(src + i);
}

// entry layout [c, kh, kw] (F.unfold / F.fold with k=2, s=2, vqgan_arch.py:324,479-484)
__global__ void gather_fold_kernel(const int64_t* __restrict__ idx_lists, const int64_t* __restrict__ tex,
                                   const float* __restrict__ books, float* __restrict__ out, int h,
                                   int w, int n, int n_e, int C) {
  const int row = blockIdx.x;  // b*h*w + i*w + j
  const int t = (int)tex[row];
  const int64_t code = idx_lists[(int64_t)t * n + row];
  const float* src = books + ((int64_t)t * n_e + code) * (C * 4);
  const int b = row / (h * w), rem = row - b * h * w;
  const int i = rem / w, j = rem - i * w;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + c * 4);  // (kh,kw) = 00,01,10,11
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int y = 2 * i + (e >> 1), x = 2 * j + (e & 1);
      out[(((int64_t)b * 2 * h + y) * 2 * w + x) * C + c] = v[e];
    }
  }
}

// One workgroup per token: 1x1 head of the token's own texture + argmax.
__global__ __launch_bounds__(256) void routed_head_argmax_kernel(
    const float* __restrict__ feat, int ldf, const float* __restrict__ w, const float* __restrict__ b,
    const int64_t* __restrict__ tex, int64_t* __restrict__ out_lists, int n, int n_heads, int Cf,
    int n_class) {
  extern __shared__ float fs[];  // Cf features + reduce slots
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = (int)tex[row];
  for (int k = tid; k < Cf; k += 256) fs[k] = feat[(int64_t)row * ldf + t * Cf + k];
  for (int hd = tid; hd < n_heads; hd += 256)
    if (hd != t) out_lists[(int64_t)hd * n + row] = -1;
  __syncthreads();
  float best = -INFINITY;
  int best_j = 0x7fffffff;
  for (int j = tid; j < n_class; j += 256) {
    const float* wr = w + ((int64_t)t * n_class + j) * Cf;
    float acc = 0.f;
    for (int k = 0; k < Cf; ++k) acc = fmaf(wr[k], fs[k], acc);
    acc += b[t * n_class + j];
    if (acc > best) {
      best = acc;
      best_j = j;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oj = __shfl_xor(best_j, o, 64);
    if (ob > best || (ob == best && oj < best_j)) {
      best = ob;
      best_j = oj;
    }
  }
  float* red = fs + Cf;
  int* redj = reinterpret_cast<int*>(red + 4);
  if (lane == 0) {
    red[wave] = best;
    redj[wave] = best_j;
  }
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < 4; ++k)
      if (red[k] > best || (red[k] == best && redj[k] < best_j)) {
        best = red[k];
        best_j = redj[k];
      }
    out_lists[(int64_t)t * n + row] = best_j;
  }
}

}  // namespace

extern "C" int t2h_vq_l2_argmin_f32(const float* z, const float* codebook, int64_t* idx, int32_t n,
                                    int32_t n_e, int32_t d, void* stream) {
  T2H_REQUIRE(z && codebook && idx, "t2h_vq_l2_argmin_f32: NULL pointer");
  T2H_REQUIRE(n_e > 0 && d > 0, "t2h_vq_l2_argmin_f32: bad codebook shape");
  if (n == 0) return T2H_OK;
  T2H_REQUIRE(n > 0, "t2h_vq_l2_argmin_f32: n=%d", n);
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid((n + 31) / 32), block(256);
  if (d == 32) hipLaunchKernelGGL(vq_argmin_kernel<32>, grid, block, 0, s, z, codebook, idx, n, n_e);
  else if (d == 64) hipLaunchKernelGGL(vq_argmin_kernel<64>, grid, block, 0, s, z, codebook, idx, n, n_e);
  else {
    t2h_set_error("t2h_vq_l2_argmin_f32: d=%d unsupported (32/64)", d);
    return T2H_ERR_UNSUPPORTED;
  }
  T2H_CHECK_LAUNCH("t2h_vq_l2_argmin_f32");
  return T2H_OK;
}

extern "C" int t2h_codebook_gather_tex_f32(const int64_t* idx_lists, const int64_t* tex,
                                           const float* books, float* out, int32_t n,
                                           int32_t n_books, int32_t n_e, int32_t e_dim, void* stream) {
  T2H_REQUIRE(idx_lists && tex && books && out, "t2h_codebook_gather_tex_f32: NULL pointer");
  T2H_REQUIRE(n > 0 && n_books > 0 && e_dim % 4 == 0, "t2h_codebook_gather_tex_f32: bad shape");
  hipLaunchKernelGGL(gather_tex_kernel, dim3(n), dim3(64), 0, static_cast<hipStream_t>(stream),
                     idx_lists, tex, books, out, n, n_e, e_dim);
  T2H_CHECK_LAUNCH("t2h_codebook_gather_tex_f32");
  return T2H_OK;
}

extern "C" int t2h_codebook_gather_fold_f32(const int64_t* idx_lists, const int64_t* tex,
                                            const float* books, float* out, int32_t B, int32_t h,
                                            int32_t w, int32_t n_books, int32_t n_e, int32_t C,
                                            void* stream) {
  T2H_REQUIRE(idx_lists && tex && books && out, "t2h_codebook_gather_fold_f32: NULL pointer");
  T2H_REQUIRE(B > 0 && h > 0 && w > 0 && C > 0 && n_books > 0, "t2h_codebook_gather_fold_f32: bad shape");
  const int n = B * h * w;
  hipLaunchKernelGGL(gather_fold_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream),
                     idx_lists, tex, books, out, h, w, n, n_e, C);
  T2H_CHECK_LAUNCH("t2h_codebook_gather_fold_f32");
  return T2H_OK;
}

extern "C" int t2h_routed_head_argmax(const float* feat, int32_t ldf, const float* w, const float* b,
                                      const int64_t* tex, int64_t* out_lists, int32_t n,
                                      int32_t n_heads, int32_t Cf, int32_t n_class, void* stream) {
  T2H_REQUIRE(feat && w && b && tex && out_lists, "t2h_routed_head_argmax: NULL pointer");
  T2H_REQUIRE(n > 0 && n_heads > 0 && Cf > 0 && n_class > 0, "t2h_routed_head_argmax: bad shape");
  const size_t lds = (size_t)(Cf + 8) * sizeof(float);
  hipLaunchKernelGGL(routed_head_argmax_kernel, dim3(n), dim3(256), lds,
                     static_cast<hipStream_t>(stream), feat, ldf, w, b, tex, out_lists, n, n_heads, Cf,
                     n_class);
  T2H_CHECK_LAUNCH("t2h_routed_head_argmax");
  return T2H_OK;
}

extern "C" int t2h_vq_argmin_tex_f32(const float* z, const float* books, const int64_t* tex, int64_t* idx_lists,
                                     int32_t n, int32_t n_books, int32_t n_e, int32_t d, int32_t fold_h,
                                     int32_t fold_w, void* stream) {
  T2H_REQUIRE(z && books && tex && idx_lists, "t2h_vq_argmin_tex_f32: NULL pointer");
  T2H_REQUIRE(n > 0 && n_books > 0 && n_e > 0, "t2h_vq_argmin_tex_f32: empty problem");
  T2H_REQUIRE((fold_h > 0) == (fold_w > 0) && (fold_h == 0 || n % (fold_h * fold_w) == 0),
              "t2h_vq_argmin_tex_f32: bad fold shape %d x %d for n=%d", fold_h, fold_w, n);
  T2H_REQUIRE(t2h_aligned16(z) && t2h_aligned16(books), "t2h_vq_argmin_tex_f32: 16-byte alignment");
  dim3 grid((n + 3) / 4), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d == 256)
    hipLaunchKernelGGL(vq_argmin_tex_kernel<1>, grid, block, 0, s, z, books, tex, idx_lists, n, n_books, n_e, fold_h, fold_w);
  else if (d == 1024)
    hipLaunchKernelGGL(vq_argmin_tex_kernel<4>, grid, block, 0, s, z, books, tex, idx_lists, n, n_books, n_e, fold_h, fold_w);
  else {
    t2h_set_error("t2h_vq_argmin_tex_f32: d=%d unsupported (256 / 1024)", d);
    return T2H_ERR_UNSUPPORTED;
  }
  T2H_CHECK_LAUNCH("t2h_vq_argmin_tex_f32");
  return T2H_OK;
}
