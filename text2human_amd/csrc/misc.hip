// Layout / pooling / resampling / epilogue kernels (all HBM-bound, coalesced
// along the NHWC channel dimension).
#include "common.h"

namespace {

__global__ void onehot_nhwc_kernel(const float* __restrict__ segm, float* __restrict__ out,
                                   int64_t n_pix, int n_cls, int Cpad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over n_pix * Cpad/4
  const int q = Cpad >> 2;
  if (i >= n_pix * q) return;
  const int64_t p = i / q;
  const int c0 = (int)(i - p * q) * 4;
  const int cls = (int)segm[p];
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (c0 + e == cls && cls < n_cls) ? 1.f : 0.f;
  *reinterpret_cast<f32x4*>(out + p * Cpad + c0) = v;
}

// [B,C,HW] -> [B,HW,C] via a 32x32 LDS tile (coalesced on both sides)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                    int HW, int ldy) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    tile[r][tx] = (c < C && p < HW) ? x[((int64_t)b * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    if (p < HW && c < C) y[((int64_t)b * HW + p) * ldy + c] = tile[tx][r];
  }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                    int C, int HW) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    tile[r][tx] = (c < C && p < HW) ? x[((int64_t)b * HW + p) * ldx + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    if (p < HW && c < C) y[((int64_t)b * C + c) * HW + p] = tile[tx][r];
  }
}

__global__ void maxpool2_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int H,
                                int W, int C, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*Ho*Wo*C/4
  if (i >= total) return;
  const int q = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const int c4 = (int)(i % q);
  int64_t p = i / q;
  const int ox = (int)(p % Wo);
  p /= Wo;
  const int oy = (int)(p % Ho);
  const int b = (int)(p / Ho);
  const float* s = x + (((int64_t)b * H + 2 * oy) * W + 2 * ox) * ldx + c4 * 4;
  const f32x4 a = *reinterpret_cast<const f32x4*>(s);
  const f32x4 bq = *reinterpret_cast<const f32x4*>(s + ldx);
  const f32x4 c = *reinterpret_cast<const f32x4*>(s + (int64_t)W * ldx);
  const f32x4 d = *reinterpret_cast<const f32x4*>(s + (int64_t)W * ldx + ldx);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaxf(a[e], bq[e]), fmaxf(c[e], d[e]));
  *reinterpret_cast<f32x4*>(y + i * 4) = o;
}

// torch upsample_bilinear2d, align_corners=False, scale_factor=2:
// src = max(0, 0.5*(dst+0.5)-0.5); i0=floor(src); i1=min(i0+1,in-1); l1=src-i0
__global__ void bilinear_up2_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                    int C, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*2H*2W*C/4
  if (i >= total) return;
  const int q = C >> 2, Ho = 2 * H, Wo = 2 * W;
  const int c4 = (int)(i % q);
  int64_t p = i / q;
  const int ox = (int)(p % Wo);
  p /= Wo;
  const int oy = (int)(p % Ho);
  const int b = (int)(p / Ho);
  const float sy = fmaxf(0.5f * ((float)oy + 0.5f) - 0.5f, 0.f);
  const float sx = fmaxf(0.5f * ((float)ox + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
  const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const float* base = x + (int64_t)b * H * W * C + c4 * 4;
  const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((int64_t)y0 * W + x0) * C);
  const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((int64_t)y0 * W + x1) * C);
  const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((int64_t)y1 * W + x0) * C);
  const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((int64_t)y1 * W + x1) * C);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    o[e] = ly0 * (lx0 * v00[e] + lx1 * v01[e]) + ly1 * (lx0 * v10[e] + lx1 * v11[e]);
  *reinterpret_cast<f32x4*>(y + i * 4) = o;
}

__global__ void argmax_rows_kernel(const float* __restrict__ x, int ld, int64_t* __restrict__ out,
                                   int64_t rows, int n) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* p = x + r * ld;
  float best = p[0];
  int bj = 0;
  for (int j = 1; j < n; ++j)
    if (p[j] > best) {
      best = p[j];
      bj = j;
    }
  out[r] = bj;
}

__global__ void image_epilogue_kernel(const float* __restrict__ dec, int ldd,
                                      float* __restrict__ img, uint8_t* __restrict__ u8, int HW,
                                      int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*HW
  if (i >= total) return;
  const int64_t b = i / HW, p = i - b * HW;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = (dec[i * ldd + c] + 1.0f) / 2.0f;
    v = fminf(fmaxf(v, 0.f), 1.f);
    if (img) img[(b * 3 + c) * HW + p] = v;
    if (u8) {
      const float s = fminf(fmaxf(__fadd_rn(__fmul_rn(v, 255.0f), 0.5f), 0.f), 255.f);
      u8[i * 3 + c] = (uint8_t)s;
    }
  }
}

// classes {1,4} -> upper, {3,5,21} -> lower, {2} -> outer; value attr+1; attr 17 = none
__global__ void texture_map_kernel(const int64_t* __restrict__ segm, const int64_t* __restrict__ upper,
                                   const int64_t* __restrict__ lower, const int64_t* __restrict__ outer,
                                   float* __restrict__ mask, int HW, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i / HW);
  const int64_t c = segm[i];
  float m = 0.f;
  if ((c == 1 || c == 4) && upper[b] != 17) m = (float)(upper[b] + 1);
  if ((c == 3 || c == 5 || c == 21) && lower[b] != 17) m = (float)(lower[b] + 1);
  if (c == 2 && outer[b] != 17) m = (float)(outer[b] + 1);
  mask[i] = m;
}

inline dim3 grid1d(int64_t total, int block = 256) { return dim3((unsigned)((total + block - 1) / block)); }

}  // namespace

extern "C" int t2h_onehot_nhwc_f32(const float* segm, float* out, int64_t n_pix, int32_t n_cls,
                                   int32_t Cpad, void* stream) {
  T2H_REQUIRE(segm && out, "t2h_onehot_nhwc_f32: NULL pointer");
  T2H_REQUIRE(n_pix > 0 && n_cls > 0 && Cpad >= n_cls && Cpad % 4 == 0, "t2h_onehot_nhwc_f32: bad shape");
  const int64_t total = n_pix * (Cpad / 4);
  hipLaunchKernelGGL(onehot_nhwc_kernel, grid1d(total), dim3(256), 0, static_cast<hipStream_t>(stream),
                     segm, out, n_pix, n_cls, Cpad);
  T2H_CHECK_LAUNCH("t2h_onehot_nhwc_f32");
  return T2H_OK;
}

extern "C" int t2h_nchw_to_nhwc_f32(const float* x, float* y, int32_t B, int32_t C, int32_t HW,
                                    int32_t ldy, void* stream) {
  T2H_REQUIRE(x && y && B > 0 && C > 0 && HW > 0 && ldy >= C, "t2h_nchw_to_nhwc_f32: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, C, HW, ldy);
  T2H_CHECK_LAUNCH("t2h_nchw_to_nhwc_f32");
  return T2H_OK;
}

extern "C" int t2h_nhwc_to_nchw_f32(const float* x, int32_t ldx, float* y, int32_t B, int32_t C,
                                    int32_t HW, void* stream) {
  T2H_REQUIRE(x && y && B > 0 && C > 0 && HW > 0 && ldx >= C, "t2h_nhwc_to_nchw_f32: bad arguments");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, ldx, y, C, HW);
  T2H_CHECK_LAUNCH("t2h_nhwc_to_nchw_f32");
  return T2H_OK;
}

extern "C" int t2h_maxpool2_nhwc_f32(const float* x, int32_t ldx, float* y, int32_t B, int32_t H,
                                     int32_t W, int32_t C, void* stream) {
  T2H_REQUIRE(x && y, "t2h_maxpool2_nhwc_f32: NULL pointer");
  T2H_REQUIRE(B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0 && ldx % 4 == 0,
              "t2h_maxpool2_nhwc_f32: bad shape");
  const int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2_kernel, grid1d(total), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     ldx, y, H, W, C, total);
  T2H_CHECK_LAUNCH("t2h_maxpool2_nhwc_f32");
  return T2H_OK;
}

extern "C" int t2h_bilinear_up2_nhwc_f32(const float* x, float* y, int32_t B, int32_t H, int32_t W,
                                         int32_t C, void* stream) {
  T2H_REQUIRE(x && y, "t2h_bilinear_up2_nhwc_f32: NULL pointer");
  T2H_REQUIRE(B > 0 && H > 0 && W > 0 && C % 4 == 0, "t2h_bilinear_up2_nhwc_f32: bad shape");
  const int64_t total = (int64_t)B * 2 * H * 2 * W * (C / 4);
  hipLaunchKernelGGL(bilinear_up2_kernel, grid1d(total), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, H, W, C, total);
  T2H_CHECK_LAUNCH("t2h_bilinear_up2_nhwc_f32");
  return T2H_OK;
}

extern "C" int t2h_argmax_rows_f32(const float* x, int32_t ld, int64_t* out, int64_t rows, int32_t n,
                                   void* stream) {
  T2H_REQUIRE(x && out && rows > 0 && n > 0 && ld >= n, "t2h_argmax_rows_f32: bad arguments");
  hipLaunchKernelGGL(argmax_rows_kernel, grid1d(rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x, ld, out, rows, n);
  T2H_CHECK_LAUNCH("t2h_argmax_rows_f32");
  return T2H_OK;
}

extern "C" int t2h_image_epilogue(const float* dec, int32_t ldd, float* img_nchw, uint8_t* img_u8,
                                  int32_t B, int32_t HW, void* stream) {
  T2H_REQUIRE(dec && (img_nchw || img_u8) && B > 0 && HW > 0 && ldd >= 3, "t2h_image_epilogue: bad arguments");
  const int64_t total = (int64_t)B * HW;
  hipLaunchKernelGGL(image_epilogue_kernel, grid1d(total), dim3(256), 0,
                     static_cast<hipStream_t>(stream), dec, ldd, img_nchw, img_u8, HW, total);
  T2H_CHECK_LAUNCH("t2h_image_epilogue");
  return T2H_OK;
}

extern "C" int t2h_texture_map(const int64_t* segm, const int64_t* upper, const int64_t* lower,
                               const int64_t* outer, float* mask, int32_t B, int32_t HW, void* stream) {
  T2H_REQUIRE(segm && upper && lower && outer && mask && B > 0 && HW > 0, "t2h_texture_map: bad arguments");
  const int64_t total = (int64_t)B * HW;
  hipLaunchKernelGGL(texture_map_kernel, grid1d(total), dim3(256), 0, static_cast<hipStream_t>(stream),
                     segm, upper, lower, outer, mask, HW, total);
  T2H_CHECK_LAUNCH("t2h_texture_map");
  return T2H_OK;
}

// ---------------------------------------------------------------- pose front-end glue
namespace {

// ShapeAttrEmbedding.forward (shape_attr_embedding_arch.py:23-35): per attribute
// one-hot -> Linear(cls,8) -> LeakyReLU(0.01) -> Linear(8,8); concat -> Linear(120,128)
// -> LeakyReLU -> Linear(128,128).  One workgroup (128 threads) per sample; the
// one-hot Linear is a row lookup in the transposed first-layer weights.
__global__ __launch_bounds__(128) void shape_attr_embed_kernel(
    const int64_t* __restrict__ attr, const int* __restrict__ cls_off, const float* __restrict__ w0t,
    const float* __restrict__ b0, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ f0, const float* __restrict__ fb0, const float* __restrict__ f1,
    const float* __restrict__ fb1, float* __restrict__ out, int n_attr, int dim, int out_dim) {
  __shared__ float h0[256], h1[256], h2[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const int cat = n_attr * dim;
  if (t < cat) {
    const int a = t / dim, j = t - a * dim;
    const int row = cls_off[a] + (int)attr[(int64_t)b * n_attr + a];
    const float v = w0t[(int64_t)row * dim + j] + b0[a * dim + j];
    h0[t] = v >= 0.f ? v : 0.01f * v;
  }
  __syncthreads();
  if (t < cat) {
    const int a = t / dim, j = t - a * dim;
    float acc = b1[a * dim + j];
    for (int k = 0; k < dim; ++k) acc = fmaf(w1[(a * dim + j) * dim + k], h0[a * dim + k], acc);
    h1[t] = acc;
  }
  __syncthreads();
  for (int o = t; o < out_dim; o += blockDim.x) {
    float acc = fb0[o];
    for (int k = 0; k < cat; ++k) acc = fmaf(f0[(int64_t)o * cat + k], h1[k], acc);
    h2[o] = acc >= 0.f ? acc : 0.01f * acc;
  }
  __syncthreads();
  for (int o = t; o < out_dim; o += blockDim.x) {
    float acc = fb1[o];
    for (int k = 0; k < out_dim; ++k) acc = fmaf(f1[(int64_t)o * out_dim + k], h2[k], acc);
    out[(int64_t)b * out_dim + o] = acc;
  }
}

// Per-pixel bias of spatially constant input channels of a 3x3 pad-1 conv:
// out[b, y, x, co] = sum over taps (dy,dx) whose source pixel lies inside the image
// of tapc[b, co, tap]  (the zero padding removes the taps that fall outside).
__global__ void tap_bias_map_kernel(const float* __restrict__ tapc, float* __restrict__ out, int H,
                                    int W, int Cout, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*H*W*Cout
  if (i >= total) return;
  const int co = (int)(i % Cout);
  int64_t p = i / Cout;
  const int x = (int)(p % W);
  p /= W;
  const int y = (int)(p % H);
  const int b = (int)(p / H);
  const float* t = tapc + ((int64_t)b * Cout + co) * 9;
  float acc = 0.f;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int sy = y + dy - 1, sx = x + dx - 1;
      if (sy >= 0 && sy < H && sx >= 0 && sx < W) acc += t[dy * 3 + dx];
    }
  out[i] = acc;
}

}  // namespace

extern "C" int t2h_shape_attr_embed_f32(const int64_t* attr, const int32_t* cls_off, const float* w0t,
                                        const float* b0, const float* w1, const float* b1,
                                        const float* f0, const float* fb0, const float* f1,
                                        const float* fb1, float* out, int32_t B, int32_t n_attr,
                                        int32_t dim, int32_t out_dim, void* stream) {
  T2H_REQUIRE(attr && cls_off && w0t && b0 && w1 && b1 && f0 && fb0 && f1 && fb1 && out,
              "t2h_shape_attr_embed_f32: NULL pointer");
  T2H_REQUIRE(B > 0 && n_attr > 0 && dim > 0 && n_attr * dim <= 128 && out_dim > 0 && out_dim <= 256,
              "t2h_shape_attr_embed_f32: unsupported sizes");
  hipLaunchKernelGGL(shape_attr_embed_kernel, dim3(B), dim3(128), 0, static_cast<hipStream_t>(stream),
                     attr, cls_off, w0t, b0, w1, b1, f0, fb0, f1, fb1, out, n_attr, dim, out_dim);
  T2H_CHECK_LAUNCH("t2h_shape_attr_embed_f32");
  return T2H_OK;
}

extern "C" int t2h_tap_bias_map_f32(const float* tapc, float* out, int32_t B, int32_t H, int32_t W,
                                    int32_t Cout, void* stream) {
  T2H_REQUIRE(tapc && out && B > 0 && H > 0 && W > 0 && Cout > 0, "t2h_tap_bias_map_f32: bad arguments");
  const int64_t total = (int64_t)B * H * W * Cout;
  hipLaunchKernelGGL(tap_bias_map_kernel, grid1d(total), dim3(256), 0, static_cast<hipStream_t>(stream),
                     tapc, out, H, W, Cout, total);
  T2H_CHECK_LAUNCH("t2h_tap_bias_map_f32");
  return T2H_OK;
}

// ---- max |x| of a tensor, as the BITS of the (non-negative) fp32 maximum: for non-negative floats the unsigned
// integer order is the numeric order, so the result is one atomicMax per wave into a caller-zeroed word -- order
// independent, hence deterministic.  Used once per checkpoint (engine.SamplerNet.calibrate_x8: the scales of the x8
// format's 8-bit planes), never on the sampling path.  NaN / inf propagate as a huge bit pattern (the caller checks).
__global__ void absmax_f32_kernel(const float* __restrict__ x, int ldx, int C, int64_t total, unsigned* __restrict__ out) {
  // (grid-stride: at most 256 workgroups, one atomic per wave -- a single word takes ~88 atomics per microsecond, and one
  // per wave of a 4 M-element tensor made this a 745 us kernel)
  unsigned b = 0u;  // (the maximum is taken on the bit patterns of |x|: monotonic for non-negative floats, NaN above all)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C;
    const unsigned t = __builtin_bit_cast(unsigned, fabsf(x[r * ldx + (i - r * C)]));
    b = t > b ? t : b;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned t = (unsigned)__shfl_xor((int)b, o, 64);
    b = t > b ? t : b;
  }
  if ((threadIdx.x & 63) == 0 && b != 0) atomicMax(out, b);
}

// the same over the fp16 hi plane of split rows / x8 rows [rows][C/32][128 bytes]: the first 64 bytes of a tile
__global__ void split_rows_absmax_kernel(const uint16_t* __restrict__ sp, int64_t total, unsigned* __restrict__ out) {
  unsigned b = 0u;  // element i of the hi planes: tile i / 32, k i % 32
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned t = __builtin_bit_cast(unsigned, fabsf((float)reinterpret_cast<const _Float16*>(sp)[(i >> 5) * 64 + (i & 31)]));
    b = t > b ? t : b;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned t = (unsigned)__shfl_xor((int)b, o, 64);
    b = t > b ? t : b;
  }
  if ((threadIdx.x & 63) == 0 && b != 0) atomicMax(out, b);
}

// dst[i] = src[rows[i]] for rows of row_bytes (multiple of 16) bytes: compacts the changed token rows
// (residual stream, attention output as split rows) for the last layer's row-wise tail
__global__ void gather_rows_kernel(const char* __restrict__ src, const int32_t* __restrict__ rows,
                                   char* __restrict__ dst, int pieces_per_row, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t r = i / pieces_per_row;
  const int pc = (int)(i - r * pieces_per_row);
  *reinterpret_cast<f32x4*>(dst + (r * pieces_per_row + pc) * 16) =
      *reinterpret_cast<const f32x4*>(src + ((int64_t)rows[r] * pieces_per_row + pc) * 16);
}

extern "C" int t2h_gather_rows(const void* src, const int32_t* rows, void* dst, int32_t n_rows, int32_t row_bytes,
                               void* stream) {
  T2H_REQUIRE(src && rows && dst && n_rows >= 0 && row_bytes > 0 && row_bytes % 16 == 0 && t2h_aligned16(src) &&
                  t2h_aligned16(dst),
              "t2h_gather_rows: bad arguments (row_bytes %% 16, 16-byte alignment)");
  if (n_rows == 0) return T2H_OK;
  const int64_t total = (int64_t)n_rows * (row_bytes / 16);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const char*>(src), rows, static_cast<char*>(dst),
                     row_bytes / 16, total);
  T2H_CHECK_LAUNCH("t2h_gather_rows");
  return T2H_OK;
}

extern "C" int t2h_absmax_f32(const float* x, int32_t ldx, int64_t rows, int32_t C, uint32_t* out_bits, void* stream) {
  T2H_REQUIRE(x && out_bits && rows > 0 && C > 0 && ldx >= C, "t2h_absmax_f32: bad arguments");
  const int64_t total = rows * C;
  dim3 grid = grid1d(total);
  grid.x = grid.x < 256u ? grid.x : 256u;
  hipLaunchKernelGGL(absmax_f32_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, C, total, out_bits);
  T2H_CHECK_LAUNCH("t2h_absmax_f32");
  return T2H_OK;
}

extern "C" int t2h_split_rows_absmax(const uint16_t* rows_split, int64_t rows, int32_t C, uint32_t* out_bits, void* stream) {
  T2H_REQUIRE(rows_split && out_bits && rows > 0 && C > 0 && C % 32 == 0, "t2h_split_rows_absmax: bad arguments");
  const int64_t total = rows * C;
  dim3 grid = grid1d(total);
  grid.x = grid.x < 256u ? grid.x : 256u;
  hipLaunchKernelGGL(split_rows_absmax_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), rows_split, total, out_bits);
  T2H_CHECK_LAUNCH("t2h_split_rows_absmax");
  return T2H_OK;
}
