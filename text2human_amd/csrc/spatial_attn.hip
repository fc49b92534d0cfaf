// Flash-style spatial self-attention of the VQGAN AttnBlock (models/archs/vqgan_arch.py:636-661):
//   out[i] = sum_j softmax_j( q_i . k_j * C^-0.5 ) v_j      over the N = H*W positions of one image,
// one head of width C (256 or 512), exact fp32 (v_mfma_f32_32x32x2_f32 = bitwise fp32 fma chains).
// The reference materialises the N x N score matrix (torch.bmm -> softmax -> torch.bmm: 16.8 MB per
// image at N = 2048, 268 MB at N = 8192); here it never leaves the registers (SURVEY.md 8(d) counts the
// block's HBM bytes this way).
//
// Workgroup = 4 waves = 32 queries of one image.  Both products are issued TRANSPOSED like the
// sampler's attention (attention.hip): S^T = K Q^T puts query = lane & 31 and 16 keys of the tile in a
// lane's 16 accumulator registers, which are exactly the B operand O^T += V^T P^T wants, so P never
// moves.  The head is 8x wider than the sampler's, so the CONTRACTION is what the waves share: wave w
// owns the channels [w C/4, (w+1) C/4) -- its quarter of every q.k dot product (the four partial score
// tiles meet in LDS, summed in wave order, so every wave holds the same bits and the softmax statistics
// are computed redundantly instead of being exchanged) and the same quarter of the output channels.
// K and V are read straight from global memory / L2 in fragment shape (a lane's 16-byte pieces); at two
// workgroups per CU the loads of one hide under the other's matrix instructions, and the exact-fp32
// matrix rate (1/16 of the 16-bit one) bounds the kernel, not the memory path.
#include "common.h"

namespace {

template <int C>
__global__ __launch_bounds__(256) void spatial_attn_kernel(const float* __restrict__ qkv, int ld,
                                                           float* __restrict__ out, int ldo, int N, float scale) {
  constexpr int DW = C / 4;   // channels of a wave
  constexpr int DH = DW / 2;  // ... of a lane half: MFMA step s contracts channels {s, DH + s} of the wave's range
  constexpr int DT = DW / 32; // output tiles of a wave; lane i of tile dt owns channel 4 i + dt (DT = 4) / 2 i + dt (DT = 2)
  static_assert(DT == 4 || DT == 2, "C = 512 or 256");
  __shared__ float sx[2][4][16][64];  // partial S^T tiles, double buffered: one barrier per key tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int nqt = N / 32;
  int img, q0;
  {  // XCD-aware mapping (workgroup id % 8 = XCD): the query tiles of one image run on one XCD and share
     // its K / V through that L2
    const int total = gridDim.x, id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3, q = total >> 3, r = total & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    img = lin / nqt;
    q0 = (lin - img * nqt) * 32;
  }
  const float* const base = qkv + (int64_t)img * N * ld;
  const int c0 = wave * DW;

  // Q fragment: lane (q = l31, h) holds Q[q][c0 + DH h + s], s = 0 .. DH - 1
  float qf[DH];
  {
    const float* qp = base + (int64_t)(q0 + l31) * ld + c0 + DH * hh;
#pragma unroll
    for (int j = 0; j < DH / 4; ++j) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(qp + 4 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) qf[4 * j + e] = v[e];
    }
  }
  f32x16 o_acc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int nkt = N / 32;
  for (int kt = 0; kt < nkt; ++kt) {
    // ---- this wave's quarter of S^T = K Q^T for the 32 keys of the tile
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    const float* kp = base + (int64_t)(kt * 32 + l31) * ld + C + c0 + DH * hh;
#pragma unroll
    for (int j = 0; j < DH / 4; ++j) {
      const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + 4 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[4 * j + e], st, 0, 0, 0);
    }
    // ---- the four quarters meet in LDS, summed in wave order
    float(*const sb)[16][64] = sx[kt & 1];
#pragma unroll
    for (int r = 0; r < 16; ++r) sb[wave][r][lane] = st[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = (((sb[0][r][lane] + sb[1][r][lane]) + sb[2][r][lane]) + sb[3][r][lane]) * scale;
    // ---- online softmax over this lane's 16 keys + the partner half's 16 (identical in every wave)
    float mx = st[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = (m_run == -INFINITY) ? 0.f : fast_exp(m_run - m_new);  // first tile: 0
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = fast_exp(st[r] - m_new);
      psum += st[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[dt][r] *= alpha;
    }
    // ---- O^T += V^T P^T for this wave's channels; step s contracts the keys {(s&3) + 8(s>>2) + 4h}
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int key = kt * 32 + (s & 3) + 8 * (s >> 2) + 4 * hh;
      const float* vp = base + (int64_t)key * ld + 2 * C + c0 + DT * l31;
      if constexpr (DT == 4) {
        const f32x4 vf = *reinterpret_cast<const f32x4*>(vp);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[dt], st[s], o_acc[dt], 0, 0, 0);
      } else {
        const float2 vf = *reinterpret_cast<const float2*>(vp);
        o_acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, st[s], o_acc[0], 0, 0, 0);
        o_acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, st[s], o_acc[1], 0, 0, 0);
      }
    }
  }
  // ---- normalise and store: accumulator register r of tile dt is channel c0 + DT i + dt of query l31,
  // i = (r & 3) + 8 (r >> 2) + 4 h: the DT tiles of one r are DT consecutive channels
  const float inv_l = 1.0f / l_run;
  float* const op = out + ((int64_t)img * N + q0 + l31) * ldo + c0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
    if constexpr (DT == 4) {
      *reinterpret_cast<f32x4*>(op + 4 * i) =
          f32x4{o_acc[0][r] * inv_l, o_acc[1][r] * inv_l, o_acc[2][r] * inv_l, o_acc[3][r] * inv_l};
    } else {
      *reinterpret_cast<float2*>(op + 2 * i) = make_float2(o_acc[0][r] * inv_l, o_acc[1][r] * inv_l);
    }
  }
}

}  // namespace

extern "C" int t2h_spatial_attention_f32(const float* qkv, int32_t ld, float* out, int32_t ldo, int32_t n_img,
                                         int32_t N, int32_t C, float scale, void* stream) {
  T2H_REQUIRE(qkv && out, "t2h_spatial_attention_f32: NULL pointer");
  T2H_REQUIRE(n_img > 0 && N > 0 && N % 32 == 0, "t2h_spatial_attention_f32: N=%d must be a positive multiple of 32", N);
  T2H_REQUIRE(C == 512 || C == 256, "t2h_spatial_attention_f32: C=%d unsupported (256, 512)", C);
  T2H_REQUIRE(ld >= 3 * C && ld % 4 == 0 && ldo >= C && ldo % 4 == 0 && t2h_aligned16(qkv) && t2h_aligned16(out),
              "t2h_spatial_attention_f32: rows must hold q|k|v (ld=%d) / the output (ldo=%d) at 16-byte alignment", ld, ldo);
  const dim3 grid((unsigned)(n_img * (N / 32))), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (C == 512) hipLaunchKernelGGL(spatial_attn_kernel<512>, grid, block, 0, s, qkv, ld, out, ldo, N, scale);
  else hipLaunchKernelGGL(spatial_attn_kernel<256>, grid, block, 0, s, qkv, ld, out, ldo, N, scale);
  T2H_CHECK_LAUNCH("t2h_spatial_attention_f32");
  return T2H_OK;
}
