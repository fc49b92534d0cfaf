// 3x3 'same' convolution with a handful of output channels -- the decoders' conv_out (128 -> 3,
// models/archs/vqgan_arch.py:997,1026-1033: conv_out(swish(norm_out(h)))) -- on the vector ALU, exact
// fp32, with the GroupNorm-apply + swish prologue.  On the matrix kernels a 3-column output occupies a
// 32- or 64-column tile: 1.35 ms for 8 images of 512x256 (1.65 ms per two images of 1024x512), 5-6 % of
// the decode for 0.3 % of its arithmetic.
//
// Workgroup = 256 threads = an 8 x 32 tile of output pixels of one image, one pixel per thread.  The
// input channels go through LDS in chunks of 32: the (8+2) x (32+2) halo tile is loaded with the
// prologue applied ONCE per input element (zero outside the image: the padding applies to the
// activated tensor), laid out [channel quad][pixel][4] so that the per-tap 16-byte reads of
// neighbouring pixels are conflict free; the weights are wave-uniform (scalar loads).
#include "common.h"

namespace {

constexpr int CS_TH = 8, CS_TW = 32, CS_HH = CS_TH + 2, CS_HW = CS_TW + 2, CS_HPIX = CS_HH * CS_HW, CS_CHUNK = 32;

template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_small_kernel(const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ w,      // [COUT][9][Cin]
                                                            const float* __restrict__ bias,   // [COUT] or NULL
                                                            const float* __restrict__ scale,  // [n_img][tbl_ld] or NULL
                                                            const float* __restrict__ shift, int tbl_ld, int act,
                                                            float* __restrict__ out, int ldo, int H, int W, int Cin) {
  __shared__ f32x4 tile[CS_CHUNK / 4][CS_HPIX];
  const int tid = threadIdx.x;
  const int tiles_x = (W + CS_TW - 1) / CS_TW, tiles_y = (H + CS_TH - 1) / CS_TH;
  const int img = blockIdx.x / (tiles_x * tiles_y), t = blockIdx.x - img * (tiles_x * tiles_y);
  const int y0 = (t / tiles_x) * CS_TH, x0 = (t % tiles_x) * CS_TW;
  const int ty = tid / CS_TW, tx = tid % CS_TW;
  const float* const xi = x + (int64_t)img * H * W * ldx;
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = bias ? bias[co] : 0.f;

  for (int c0 = 0; c0 < Cin; c0 += CS_CHUNK) {
    __syncthreads();  // the previous chunk's reads are done
    for (int i = tid; i < CS_HPIX * (CS_CHUNK / 4); i += 256) {
      const int pix = i / (CS_CHUNK / 4), q = i - pix * (CS_CHUNK / 4);
      const int py = y0 - 1 + pix / CS_HW, px = x0 - 1 + pix % CS_HW;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (py >= 0 && py < H && px >= 0 && px < W) {
        v = *reinterpret_cast<const f32x4*>(xi + ((int64_t)py * W + px) * ldx + c0 + 4 * q);
        if (scale) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + (int64_t)img * tbl_ld + c0 + 4 * q);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + (int64_t)img * tbl_ld + c0 + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a = fmaf(v[e], sc[e], sh[e]);
            if (act == 1) a = a / (1.0f + fast_exp(fminf(-a, 87.0f)));  // swish, as the matrix kernels' prologue
            v[e] = a;
          }
        }
      }
      tile[q][pix] = v;
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int pix = (ty + tap / 3) * CS_HW + tx + tap % 3;
#pragma unroll
      for (int q = 0; q < CS_CHUNK / 4; ++q) {
        const f32x4 a = tile[q][pix];
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          const float* wp = w + ((int64_t)co * 9 + tap) * Cin + c0 + 4 * q;  // wave-uniform: scalar loads
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[co] = fmaf(a[e], wp[e], acc[co]);
        }
      }
    }
  }
  const int oy = y0 + ty, ox = x0 + tx;
  if (oy < H && ox < W) {
    float* o = out + ((int64_t)img * H * W + (int64_t)oy * W + ox) * ldo;
#pragma unroll
    for (int co = 0; co < COUT; ++co) o[co] = acc[co];
  }
}

}  // namespace

extern "C" int t2h_conv3x3_small_f32(const float* x, int32_t ldx, const float* w, const float* bias, const float* scale,
                                     const float* shift, int32_t tbl_ld, int32_t act, float* out, int32_t ldo,
                                     int32_t n_img, int32_t H, int32_t W, int32_t Cin, int32_t Cout, void* stream) {
  T2H_REQUIRE(x && w && out, "t2h_conv3x3_small_f32: NULL pointer");
  T2H_REQUIRE(n_img > 0 && H > 0 && W > 0 && Cin > 0 && Cin % CS_CHUNK == 0 && Cout >= 1 && Cout <= 4,
              "t2h_conv3x3_small_f32: Cin=%d must be a multiple of 32, Cout=%d in 1..4", Cin, Cout);
  T2H_REQUIRE(ldx >= Cin && ldx % 4 == 0 && ldo >= Cout && t2h_aligned16(x) && (scale == nullptr) == (shift == nullptr) &&
                  (!scale || (tbl_ld >= Cin && tbl_ld % 4 == 0 && t2h_aligned16(scale) && t2h_aligned16(shift))) &&
                  (act == 0 || act == 1),
              "t2h_conv3x3_small_f32: bad strides / alignment / prologue");
  const int tiles = ((W + CS_TW - 1) / CS_TW) * ((H + CS_TH - 1) / CS_TH);
  const dim3 grid((unsigned)(n_img * tiles)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define T2H_CS_LAUNCH(N) \
  hipLaunchKernelGGL(conv3x3_small_kernel<N>, grid, block, 0, s, x, ldx, w, bias, scale, shift, tbl_ld, act, out, ldo, H, W, Cin)
  switch (Cout) {
    case 1: T2H_CS_LAUNCH(1); break;
    case 2: T2H_CS_LAUNCH(2); break;
    case 3: T2H_CS_LAUNCH(3); break;
    default: T2H_CS_LAUNCH(4); break;
  }
#undef T2H_CS_LAUNCH
  T2H_CHECK_LAUNCH("t2h_conv3x3_small_f32");
  return T2H_OK;
}
