// Non-causal multi-head self-attention of the index sampler, fp32 MFMA, flash
// style (no TxT matrix in HBM).  Replaces CausalSelfAttention.forward with
// causal=False (models/archs/transformer_arch.py:52-67): softmax(q k^T/sqrt(64)) v.
//
// Workgroup = 8 waves = 128 query rows of one (batch, head).  Wave w handles the
// 32 queries (w & 3) against key half (w >> 2): the two waves that share a SIMD
// work on different halves of the keys, so one wave's softmax VALU / LDS reads
// run in the shadow of the other wave's MFMAs (at batch 8 the whole chip only
// has 4 such 32-query units per CU, i.e. ONE wave per SIMD without the split).
// The two partial results (running max m, running sum l, unnormalised O) are
// merged through LDS at the end (flash-decoding style).
//
// Both matmuls are issued in TRANSPOSED form so that everything indexed by the
// query stays lane-local (lane&31 = query, lane>>5 = k-half of the MFMA):
//   S^T[key][q]  = sum_d K[key][d] * Q[q][d]      A = K tile (LDS), B = Q (regs)
//   O^T[d][q]   += sum_key V[key][d] * P[q][key]   A = V^T (LDS),  B = P (regs)
// The C/D layout of v_mfma_f32_32x32x2_f32 (col = lane&31, row = (r&3) +
// 8*(r>>2) + 4*(lane>>5)) puts, for S^T, query = lane&31 and 16 keys in the 16
// accumulator registers of each lane; the second MFMA wants B[k][j] from lane
// (j = query, k-half h) -- exactly those registers if MFMA step s contracts the
// key pair {(s&3)+8(s>>2)+4h, h=0,1}.  So P never leaves registers, the online
// softmax statistics (running max m, running sum l) are per-lane scalars, and
// rescaling the O^T accumulator is a plain per-lane multiply.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr int HD = 64;       // head dim
constexpr int QB = 128;      // queries per workgroup
constexpr int KT = 64;       // keys per LDS tile (per key half)
constexpr int K_LD = 68;     // K tile row stride (17 x 16B slots, odd -> conflict-free b128)
constexpr int V_LD = 64;
constexpr int O_LD = 68;
constexpr int KV_TILE = KT * K_LD + KT * V_LD;  // floats per (K,V) tile pair

__global__ __launch_bounds__(512) void mha_kernel(const float* __restrict__ qkv,
                                                  float* __restrict__ y, int T, int C,
                                                  int n_head, uint16_t* __restrict__ y_split, int* ovf) {
  // [2 key halves][K tile | V tile]; reused at the end: [0, 8192) partial O of
  // the second key half, [8192, 8192 + 4*32*O_LD) output transpose staging.
  __shared__ __attribute__((aligned(16))) float smem[2 * KV_TILE];
  static_assert(2 * KV_TILE >= 4 * 32 * 64 + 4 * 32 * O_LD, "LDS reuse layout");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int qw = wave & 3, kh = wave >> 2;
  // XCD-aware mapping (workgroup id % 8 = XCD): the T/128 query tiles of one
  // (batch, head) run on the same XCD and share its K/V through that L2.
  int qt, head, b;
  {
    const int nqt = T / QB, total = gridDim.x, id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3, q = total >> 3, r = total & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    qt = lin % nqt;
    const int hb = lin / nqt;
    head = hb % n_head;
    b = hb / n_head;
  }
  const int q0 = qt * QB + qw * 32;
  const int ld = 3 * C;
  const float* base = qkv + (int64_t)b * T * ld + head * HD;

  // Q fragment: lane (q = l31, h) holds Q[q][32h + s] * 1/8, s = 0..31
  float qf[32];
  {
    const float* qp = base + (int64_t)(q0 + l31) * ld + 32 * hh;
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(qp + 4 * s4);
#pragma unroll
      for (int e = 0; e < 4; ++e) qf[4 * s4 + e] = v[e] * 0.125f;
    }
  }

  f32x16 o_acc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // staging: per key half one K and one V tile of 64 keys x 64 floats = 1024
  // float4 each; threads [0,256) stage half 0, [256,512) half 1: 4 + 4 per thread
  const int s_half = tid >> 8, s_t = tid & 255;
  const int s_col4 = s_t & 15, s_row0 = s_t >> 4;  // 16 float4 per key row, 16 rows per pass
  float* const Ks_st = smem + s_half * KV_TILE;
  float* const Vs_st = Ks_st + KT * K_LD;
  const int half_keys = T / 2;
  f32x4 kreg[4], vreg[4];
  auto load_kv = [&](int it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = s_half * half_keys + it * KT + s_row0 + 16 * i;
      const float* p = base + (int64_t)key * ld + s_col4 * 4;
      kreg[i] = *reinterpret_cast<const f32x4*>(p + C);
      vreg[i] = *reinterpret_cast<const f32x4*>(p + 2 * C);
    }
  };
  auto store_kv = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = s_row0 + 16 * i;
      *reinterpret_cast<f32x4*>(Ks_st + r * K_LD + s_col4 * 4) = kreg[i];
      *reinterpret_cast<f32x4*>(Vs_st + r * V_LD + s_col4 * 4) = vreg[i];
    }
  };

  const float* const Ks = smem + kh * KV_TILE;
  const float* const Vs = Ks + KT * K_LD;
  const int nit = half_keys / KT;
  load_kv(0);
  for (int it = 0; it < nit; ++it) {
    __syncthreads();  // previous tiles fully consumed
    store_kv();
    __syncthreads();
    if (it + 1 < nit) load_kv(it + 1);

#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // two 32-key sub-tiles
      // ---- S^T = K Q^T
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
      const float* kp = Ks + (ks * 32 + l31) * K_LD + 32 * hh;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + 4 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[4 * j + e], st, 0, 0, 0);
      }
      // ---- online softmax over this lane's 16 keys + partner half's 16 keys
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
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[dt][r] *= alpha;
      // ---- O^T += V^T P^T ; step s contracts keys {(s&3)+8(s>>2)+4h}
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int key = ks * 32 + (s & 3) + 8 * (s >> 2) + 4 * hh;
        const float* vp = Vs + key * V_LD + l31;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32 * dt], st[s], o_acc[dt], 0, 0, 0);
      }
    }
  }

  // ---- merge the two key halves: waves 4-7 publish (m, l, O), waves 0-3 combine
  __syncthreads();
  float* const Ox = smem;                // [4 waves][32 regs][64 lanes]
  float* const Mx = smem + 4 * 32 * 64;  // borrowed from the staging area below:
  float* const Lx = Mx + 4 * 64;         // consumed before that area is written
  if (kh == 1) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) Ox[(qw * 32 + dt * 16 + r) * 64 + lane] = o_acc[dt][r];
    Mx[qw * 64 + lane] = m_run;
    Lx[qw * 64 + lane] = l_run;
  }
  __syncthreads();
  float inv_l = 0.f;
  if (kh == 0) {
    const float m2 = Mx[qw * 64 + lane], l2 = Lx[qw * 64 + lane];
    const float m = fmaxf(m_run, m2);
    const float a1 = expf(m_run - m), a2 = expf(m2 - m);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        o_acc[dt][r] = o_acc[dt][r] * a1 + Ox[(qw * 32 + dt * 16 + r) * 64 + lane] * a2;
    inv_l = 1.0f / (l_run * a1 + l2 * a2);
  }
  __syncthreads();  // Mx/Lx consumed before the staging area is overwritten
  // ---- normalise, transpose through LDS, coalesced row stores (waves 0-3)
  float* Os = smem + 4 * 32 * 64 + qw * 32 * O_LD;
  if (kh == 0) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        Os[l31 * O_LD + d] = o_acc[dt][r] * inv_l;
      }
  }
  __syncthreads();
  if (kh == 0) {
    // fp32 rows and / or split rows (operand of the split-precision proj GEMM)
    const int64_t grow = (int64_t)b * T + q0;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
      const f32x4 va = *reinterpret_cast<const f32x4*>(Os + row * O_LD + c8);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(Os + row * O_LD + c8 + 4);
      if (y) {
        *reinterpret_cast<f32x4*>(y + (grow + row) * C + head * HD + c8) = va;
        *reinterpret_cast<f32x4*>(y + (grow + row) * C + head * HD + c8 + 4) = vb;
      }
      if (y_split) t2h_store_split8(y_split, grow + row, C, head * HD + c8, va, vb, ovf);
    }
  }
}


// ---------------------------------------------------------------------------------
// The same attention with both products on the 16-bit matrix cores as three partial
// products of two-plane fp16 operands (see gemm_split.hip): q, k arrive as split rows
// (C_split of the q|k|v projection), v as the transposed planes Vt that projection's
// epilogue wrote, P is split in registers.  Structure identical to mha_kernel: 8 waves
// = 4 x 32 queries x 2 key halves, transposed products, P never leaves registers --
// with v_mfma_f32_32x32x16_f16 a lane's B operand of k16-step j is the 8 keys
// {16j + 4h + (e&3) + 8(e>>2)} = accumulator registers 8j .. 8j+7 of S^T, and Vt stores
// the keys of every 32-group in exactly that order, so the A operand is one 16-byte read.
typedef t2h_f16x8 f16x8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int J>
__device__ __forceinline__ void split8(const f32x16& x, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    _Float16 a, b;
    t2h_split2(x[8 * J + e], a, b);
    hi[e] = a;
    lo[e] = b;
  }
}

#ifndef T2H_MHA_PACKED
#define T2H_MHA_PACKED 0  // (round-6 A/B, tools/mha_packed_ab.py; see the softmax of mha_split_pipe_kernel)
#endif
#ifdef T2H_MHA_TIMING
__device__ long long* g_mha_timing = nullptr;  // debug builds only (tools/mha_split_ablate.py)
#define TM_NOW() clock64()
#else
#define TM_NOW() 0ll
#endif

// Barrier among the 4 waves of ONE key half (monotonic LDS counter).  The two key halves
// share nothing until the final merge; a block-wide barrier per K/V tile would keep the
// two waves of a SIMD (same queries, different key half) in lockstep -- both in their
// matrix phase, then both in their softmax phase -- whereas independent halves drift
// apart and run one wave's exp / split VALU work under the other's MFMAs.
__device__ __forceinline__ void half_barrier(int* ctr, int target, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS reads / writes are done
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}

// K / Vt tiles are staged by LDS-DMA: unpadded images
constexpr int DK_TILE = KT * 256;            // K tile: 64 keys x 256 B
constexpr int DV_TILE = 2 * HD * 128;        // Vt tile: 128 (plane, d) rows x 64 keys x 2 B
constexpr int DKV_TILE = DK_TILE + DV_TILE;  // per key half and buffer

// KH = 2 (the shape above): 4 x 32 queries x 2 key halves, merged through LDS at the end -- what fills the chip at
// B = 8 (256 workgroups).  KH = 1: 8 x 32 queries of one (batch, head), every wave over ALL keys -- the K / Vt tiles
// are shared by eight waves instead of four (half the L2 -> LDS bytes per query), twice the key tiles per wave behind
// one prologue, no (m, l, O) merge and no second barrier domain.  Needs >= 256 workgroups of 256 queries to fill the
// chip: B >= 16 (t2h_mha_split_f32 picks by the number of rounds of the 256 CUs each form takes).
template <int KH>
__global__ __launch_bounds__(512) void mha_split_pipe_kernel(const uint16_t* __restrict__ qk, int ld_cols,
                                                        const uint16_t* __restrict__ vt, float* __restrict__ y,
                                                        uint16_t* __restrict__ y_split, int T, int C, int n_head,
                                                        int* ovf, float y_x8_scale) {  // y_x8_scale > 0: y_split in the x8 format
  // two (K, Vt) tile pairs per key half (double buffer); reused at the end for the merge + output
  // transpose staging
  constexpr int NQW = 8 / KH;  // waves (x 32 queries) per key half
  constexpr int EPI_B = KH == 2 ? (4 * 32 * 64 + 4 * 32 * O_LD) * 4 : NQW * 32 * O_LD * 4;
  constexpr int SMEM_B = 2 * KH * DKV_TILE > EPI_B ? 2 * KH * DKV_TILE : EPI_B;
  __shared__ __attribute__((aligned(16))) char smem_raw[SMEM_B + 16];
  float* const smem = reinterpret_cast<float*>(smem_raw);
  int* const bar = reinterpret_cast<int*>(smem_raw + SMEM_B);  // one counter per key half

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform values in scalar registers
  const long long tm0 = TM_NOW();
#ifdef T2H_MHA_TIMING
  const long long rt0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  long long tm_stage = 0, tm_comp = 0;
  if (tid < 2) bar[tid] = 0;
  __syncthreads();
  const int l31 = lane & 31, hh = lane >> 5;
  const int qw = KH == 2 ? (wave & 3) : wave, kh = KH == 2 ? (wave >> 2) : 0;
  int qt, head, b;
  {
    const int nqt = T / (32 * NQW), total = gridDim.x, id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3, q = total >> 3, r = total & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    qt = lin % nqt;
    const int hb = lin / nqt;
    head = hb % n_head;
    b = hb / n_head;
  }
  const int q0 = qt * (32 * NQW) + qw * 32;
  const int64_t row_b = (int64_t)(ld_cols / 32) * T2H_SPLIT_TILE_B;  // bytes per split row
  const char* const qk_b = reinterpret_cast<const char*>(qk) + (int64_t)b * T * row_b;
  const int q_tile0 = 2 * head, k_tile0 = C / 32 + 2 * head;  // 32-column tiles of this head's q / k

  // Q fragments: k16-step kk covers d = 16 kk + 8 h .. + 7 of plane pl
  f16x8 qf[4][2];
  {
    const char* qp = qk_b + (int64_t)(q0 + l31) * row_b + q_tile0 * T2H_SPLIT_TILE_B + hh * 16;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        qf[kk][pl] = *reinterpret_cast<const f16x8*>(qp + (kk >> 1) * T2H_SPLIT_TILE_B + pl * 64 + (kk & 1) * 32);
  }

  f32x16 o_acc[2], o_lo[2];  // O^T = o_acc + 2^-11 o_lo (hi*hi and the cross products)
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[dt][r] = o_lo[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // ---- staging by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass).  Tile
  // images are unpadded -- K: 64 keys x 256 B, Vt: 128 (plane, d) rows x 128 B -- and the 16-byte
  // fragment reads stay conflict free through an XOR swizzle applied on the DMA's SOURCE piece and on
  // the read address: K piece p of key r lives at p ^ (r & 15), Vt piece p of row r at p ^ ((r >> 1) & 7).
  // A DMA instruction fills 1 KiB lane-linearly = 4 K rows or 8 Vt rows; wave w of the half issues
  // chunks 4w .. 4w + 3 of either tile (rows 16w + 4i + (lane >> 4) / 32w + 8i + (lane >> 3)).
  const int half_keys = T / KH;
  const int w4 = qw;                 // wave inside the key half: issues CPW of the 16 1-KiB chunks of either tile
  constexpr int CPW = 16 / NQW;      // 4 (four waves per half) or 2 (eight)
  const char* const vt_b = reinterpret_cast<const char*>(vt) + ((int64_t)(b * n_head + head) * 2 * HD) * T * 2;
  const unsigned lds_half = (unsigned)(uintptr_t)smem_raw + kh * (2 * DKV_TILE);  // this half's two buffers
  const char* const kbase = qk_b + (int64_t)(kh * half_keys) * row_b + k_tile0 * T2H_SPLIT_TILE_B;
  const char* const vbase = vt_b + (int64_t)(kh * half_keys) * 2;
  // (chunk c = CPW w4 + i: K rows 4 c + (lane >> 4), Vt rows 8 c + (lane >> 3))
  const unsigned k_rowoff = (unsigned)(4 * CPW * w4 + (lane >> 4)) * (unsigned)row_b;
  // source piece of chunk i: this ^ 64 i  (K piece p of key r lives at p ^ (r & 15), r & 15 = (4 CPW w4 & 15) + 4 i + (lane >> 4))
  const unsigned k_b16 = (unsigned)((lane & 15) ^ (lane >> 4) ^ ((4 * CPW * w4) & 15)) * 16;
  const unsigned v_rowoff = (unsigned)(8 * CPW * w4 + (lane >> 3)) * (unsigned)(T * 2);
  // source piece of chunk i: this ^ 64 (i & 1)  (Vt piece p of row r at p ^ ((r >> 1) & 7) = (4 CPW w4 + 4 i + (lane >> 4)) & 7)
  const unsigned v_b16 = (unsigned)((lane & 7) ^ (lane >> 4) ^ ((4 * CPW * w4) & 7)) * 16;
  auto dma16 = [&](unsigned voff, const char* sbase, unsigned lds_dst) {
    unsigned keep;
#ifndef T2H_MHA_DMA_POLICY
#define T2H_MHA_DMA_POLICY ""
#endif
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" T2H_MHA_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
  };
  auto dma_k = [&](int it, int buf) {
    const char* const sb = kbase + (int64_t)it * KT * row_b;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_half + buf * DKV_TILE + CPW * w4 * 1024);
#pragma unroll
    for (int i = 0; i < CPW; ++i) dma16(k_rowoff + (unsigned)(4 * i) * (unsigned)row_b + (k_b16 ^ (64u * i)), sb, dst + i * 1024);
  };
  auto dma_v = [&](int it, int buf) {
    const char* const sb = vbase + it * KT * 2;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_half + buf * DKV_TILE + DK_TILE + CPW * w4 * 1024);
#pragma unroll
    for (int i = 0; i < CPW; ++i) dma16(v_rowoff + (unsigned)(8 * i) * (unsigned)(T * 2) + (v_b16 ^ (64u * (i & 1))), sb, dst + i * 1024);
  };
  auto dma_landed = [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  const char* const Ks0 = smem_raw + kh * (2 * DKV_TILE);
  const int nit = half_keys / KT;
  // fragment addresses in the swizzled images
  const unsigned xk16 = (unsigned)(hh ^ (l31 & 15)) * 16, xv16 = (unsigned)(hh ^ ((l31 >> 1) & 7)) * 16;
  // S^T = K Q^T for BOTH 32-key sub-tiles of the tile at `Ks`.  Issue order (l,h)0 (l,h)1 (h,h)0
  // (h,l)0 (h,l)1 (h,h)1 per k16-step: consecutive matrix instructions never share an accumulator
  // (a dependent v_mfma waits for its predecessor's last pass).
  auto s_tile = [&](const char* Ks, f32x16 (&st)[2], f32x16 (&st_lo)[2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[ks][r] = st_lo[ks][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f16x8 kf[2][2];  // [sub-tile][plane]
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          kf[ks][pl] = *reinterpret_cast<const f16x8*>(
              Ks + (ks * 32 + l31) * 256 + ((unsigned)(((kk >> 1) * 8 + pl * 4 + (kk & 1) * 2) * 16) ^ xk16));
      st_lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0][1], qf[kk][0], st_lo[0], 0, 0, 0);
      st_lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1][1], qf[kk][0], st_lo[1], 0, 0, 0);
      st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0][0], qf[kk][0], st[0], 0, 0, 0);
      st_lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0][0], qf[kk][1], st_lo[0], 0, 0, 0);
      st_lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1][0], qf[kk][1], st_lo[1], 0, 0, 0);
      st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1][0], qf[kk][0], st[1], 0, 0, 0);
    }
  };

  int bar_n = 0;
  const long long tm1 = TM_NOW();
#ifdef T2H_MHA_TIMING
  const long long rt1 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  // the second-dispatched half of the workgroup loses every issue arbitration against its older SIMD
  // partner (measured: its loop took 37k cycles against 26k): static priority evens the two out
  if (KH == 2 && kh == 1) __builtin_amdgcn_s_setprio(1);
  // ---- software pipeline over the key tiles (two (K, Vt) buffers per half, one barrier per tile):
  // the matrix pipe forms S^T of tile j + 1 while the vector ALU does the softmax of tile j -- the two
  // are independent, so the compiler interleaves them instead of the wave waiting for its own matrix
  // results.  Iteration j requests K(j+2) into the buffer whose K(j) was last read by S^T(j), in
  // iteration j-1, and V(j+1) into the buffer whose V(j-1) was last read in iteration j-1.
  constexpr float SC = 0.125f * 1.44269504088896340736f;  // log2(e) / sqrt(d)
  f32x16 sc[2];  // S^T of the current tile (folded, unscaled); becomes the probabilities in place
  {
    f32x16 st[2], st_lo[2];
    dma_k(0, 0);
    dma_v(0, 0);
    if (nit > 1) dma_k(1, 1);
    dma_landed();
    half_barrier(bar + kh, bar_n += NQW, lane);
    s_tile(Ks0, st, st_lo);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[ks][r] = fmaf(st_lo[ks][r], T2H_SPLIT_LO_INV, st[ks][r]);
    half_barrier(bar + kh, bar_n += NQW, lane);  // K(0) consumed by every wave before K(2) replaces it
  }
  auto iteration = [&](int it, auto next_c) {
    constexpr bool NEXT = decltype(next_c)::value;  // also form S^T of tile it + 1 (all but the last)
    const long long tb = TM_NOW();
    const int buf = it & 1;
    const char* const Vs = Ks0 + buf * DKV_TILE + DK_TILE;
    if (it + 2 < nit) dma_k(it + 2, buf);
    if (it + 1 < nit) dma_v(it + 1, buf ^ 1);
    // Online softmax over the tile's 64 keys: this lane's 2 x 16 + the partner half's.  ONE rescale of
    // the running state per tile, skipped (wave-uniformly) when no lane's maximum moved: alpha would be
    // exp(0) = 1 exactly, so skipping changes no bit.  Base-2 domain: p = exp2(s c - M) with
    // c = log2(e) / sqrt(d) and M the running maximum of s c (one fma + v_exp_f32 per score; the
    // compensated exp of the exact-fp32 kernel costs six more instructions per score, and the vector
    // ALU, not the matrix pipe, is what this loop waits for).  M is the SAME rounded number in every
    // term of a row -- the rescale factor below is formed from the rounded values too -- so its
    // rounding cancels in the normalisation; what remains is the rounding of the argument, |arg| ulp
    // on a term of weight 2^arg: below the fp32 summation error.
    float mx = -INFINITY;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[ks][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * SC;
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new > m_run)) {
      const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);  // first tile: 0
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o_acc[dt][r] *= alpha;
          o_lo[dt][r] *= alpha;
        }
      m_run = m_new;
    }
    f32x16 sn[2], sn_lo[2];
    if constexpr (NEXT) s_tile(Ks0 + (buf ^ 1) * DKV_TILE, sn, sn_lo);
    float psum = 0.f;
#if T2H_MHA_PACKED
    // (round 6: the exponent arguments and the row sum on register PAIRS -- v_pk_fma_f32 / v_pk_add_f32: 16 + 16
    // instructions per sub-tile where the compiler had left 32 v_fma_f32 + a serial chain of 32 v_add_f32)
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ sc2 = {SC, SC}, nm2 = {-m_run, -m_run};
    f32x2_ ps2 = {0.f, 0.f};
#endif
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#if T2H_MHA_PACKED
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2_ a = {sc[ks][r], sc[ks][r + 1]};
        a = __builtin_elementwise_fma(a, sc2, nm2);
        a[0] = __builtin_amdgcn_exp2f(a[0]);
        a[1] = __builtin_amdgcn_exp2f(a[1]);
        sc[ks][r] = a[0];
        sc[ks][r + 1] = a[1];
        ps2 += a;
      }
#else
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sc[ks][r] = __builtin_amdgcn_exp2f(fmaf(sc[ks][r], SC, -m_run));
        psum += sc[ks][r];
      }
#endif
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f16x8 pf[2];
        if (j == 0) split8<0>(sc[ks], pf[0], pf[1]);
        else split8<1>(sc[ks], pf[0], pf[1]);
        f16x8 vf[2][2];  // [d half][plane]
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
            vf[dt][pl] = *reinterpret_cast<const f16x8*>(Vs + (pl * HD + dt * 32 + l31) * 128 +
                                                          ((unsigned)((ks * 4 + j * 2) * 16) ^ xv16));
        o_lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0][1], pf[0], o_lo[0], 0, 0, 0);
        o_lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[1][1], pf[0], o_lo[1], 0, 0, 0);
        o_acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0][0], pf[0], o_acc[0], 0, 0, 0);
        o_lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0][0], pf[1], o_lo[0], 0, 0, 0);
        o_lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[1][0], pf[1], o_lo[1], 0, 0, 0);
        o_acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[1][0], pf[0], o_acc[1], 0, 0, 0);
      }
    }
#if T2H_MHA_PACKED
    psum = ps2[0] + ps2[1];
#endif
    psum += __shfl_xor(psum, 32, 64);
    l_run += psum;
    if constexpr (NEXT) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[ks][r] = fmaf(sn_lo[ks][r], T2H_SPLIT_LO_INV, sn[ks][r]);
    }
    const long long tc = TM_NOW();
    tm_comp += tc - tb;
    if constexpr (NEXT) {
      dma_landed();
      half_barrier(bar + kh, bar_n += NQW, lane);  // tile it + 1 published, tile it released
    }
    tm_stage += TM_NOW() - tc;
  };
  for (int it = 0; it + 1 < nit; ++it) iteration(it, std::true_type{});
  iteration(nit - 1, std::false_type{});

#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[dt][r] = fmaf(o_lo[dt][r], T2H_SPLIT_LO_INV, o_acc[dt][r]);
  const long long tm2 = TM_NOW();
#ifdef T2H_MHA_TIMING
  const long long rt2 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  __syncthreads();
  if constexpr (KH == 2) {
  // ---- merge the two key halves: waves 4-7 publish (m, l, O), waves 0-3 combine
  float* const Ox = smem;                // [4 waves][32 regs][64 lanes]
  float* const Mx = smem + 4 * 32 * 64;  // borrowed from the staging area below:
  float* const Lx = Mx + 4 * 64;         // consumed before that area is written
  if (kh == 1) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) Ox[(qw * 32 + dt * 16 + r) * 64 + lane] = o_acc[dt][r];
    Mx[qw * 64 + lane] = m_run;
    Lx[qw * 64 + lane] = l_run;
  }
  __syncthreads();
  float inv_l = 0.f;
  if (kh == 0) {
    const float m2 = Mx[qw * 64 + lane], l2 = Lx[qw * 64 + lane];
    const float m = fmaxf(m_run, m2);
    const float a1 = exp2f(m_run - m), a2 = exp2f(m2 - m);  // (maxima are kept in the base-2 domain)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        o_acc[dt][r] = o_acc[dt][r] * a1 + Ox[(qw * 32 + dt * 16 + r) * 64 + lane] * a2;
    inv_l = 1.0f / (l_run * a1 + l2 * a2);
  }
  __syncthreads();  // Mx/Lx consumed before the staging area is overwritten
  float* Os = smem + 4 * 32 * 64 + qw * 32 * O_LD;
  if (kh == 0) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        Os[l31 * O_LD + d] = o_acc[dt][r] * inv_l;
      }
  }
  __syncthreads();
  {  // both key halves store: 16 of the 32 staged rows each
    const int64_t grow = (int64_t)b * T + q0;
#pragma unroll
    for (int it2 = 0; it2 < 2; ++it2) {
      const int it = it2 + 2 * kh;
      const int row = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
      const f32x4 va = *reinterpret_cast<const f32x4*>(Os + row * O_LD + c8);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(Os + row * O_LD + c8 + 4);
      if (y) {
        *reinterpret_cast<f32x4*>(y + (grow + row) * C + head * HD + c8) = va;
        *reinterpret_cast<f32x4*>(y + (grow + row) * C + head * HD + c8 + 4) = vb;
      }
      if (y_split) {
        if (y_x8_scale > 0.f) t2h_store_x8_8<1>(y_split, grow + row, C, head * HD + c8, va, vb, y_x8_scale, ovf);
        else t2h_store_split8(y_split, grow + row, C, head * HD + c8, va, vb, ovf);
      }
    }
  }
  } else {
    // ---- every wave has seen all keys: normalise, transpose its 32 x 64 tile through its own staging rows (the
    // tile buffers are idle: the barrier above), coalesced row stores
    const float inv_l = 1.0f / l_run;
    float* const Os = smem + qw * 32 * O_LD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        Os[l31 * O_LD + d] = o_acc[dt][r] * inv_l;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (wave-private staging: no barrier)
    const int64_t grow = (int64_t)b * T + q0;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
      const f32x4 va = *reinterpret_cast<const f32x4*>(Os + row * O_LD + c8);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(Os + row * O_LD + c8 + 4);
      if (y) {
        *reinterpret_cast<f32x4*>(y + (grow + row) * C + head * HD + c8) = va;
        *reinterpret_cast<f32x4*>(y + (grow + row) * C + head * HD + c8 + 4) = vb;
      }
      if (y_split) {
        if (y_x8_scale > 0.f) t2h_store_x8_8<1>(y_split, grow + row, C, head * HD + c8, va, vb, y_x8_scale, ovf);
        else t2h_store_split8(y_split, grow + row, C, head * HD + c8, va, vb, ovf);
      }
    }
  }
#ifdef T2H_MHA_TIMING
  if (g_mha_timing && blockIdx.x == 8 && lane == 0 && (wave == 0 || wave == 4)) {
    long long* o = g_mha_timing + (wave == 4 ? 8 : 0);
    const long long te = clock64();
    o[0] = te - tm0; o[1] = tm1 - tm0; o[2] = tm_stage; o[3] = tm_comp; o[4] = te - tm2; o[5] = tm2 - tm1;
    // s_memrealtime (100 MHz) spans of the same phases: total, prologue, loop -> the clock of each
    const long long rte = (long long)__builtin_amdgcn_s_memrealtime();
    o[6] = rte - rt0; o[7] = (rt1 - rt0) | ((rt2 - rt1) << 32);
  }
#endif
}

thread_local int g_force_mha_form = 0;  // tuning / test hook of the calling thread: 1 all keys, 2 key halves, 0 automatic

}  // namespace

extern "C" int t2h_mha_split_force_form(int form) {
  const int old = g_force_mha_form;
  g_force_mha_form = (form == 1 || form == 2) ? form : 0;
  return old;
}

extern "C" int t2h_mha_noncausal_f32(const float* qkv, float* y, int32_t B, int32_t T,
                                     int32_t n_head, void* stream) {
  T2H_REQUIRE(qkv && y, "t2h_mha_noncausal_f32: NULL pointer");
  T2H_REQUIRE(B > 0 && n_head > 0, "t2h_mha_noncausal_f32: empty problem");
  T2H_REQUIRE(T > 0 && T % QB == 0, "t2h_mha_noncausal_f32: T=%d must be a multiple of %d", T, QB);
  T2H_REQUIRE(t2h_aligned16(qkv) && t2h_aligned16(y), "t2h_mha_noncausal_f32: 16-byte alignment");
  const int C = n_head * HD;
  dim3 grid((T / QB) * n_head * B), block(512);
  hipLaunchKernelGGL(mha_kernel, grid, block, 0, static_cast<hipStream_t>(stream), qkv, y, T, C, n_head,
                     static_cast<uint16_t*>(nullptr), static_cast<int*>(nullptr));
  T2H_CHECK_LAUNCH("t2h_mha_noncausal_f32");
  return T2H_OK;
}

extern "C" int t2h_mha_noncausal_split_f32(const float* qkv, uint16_t* y_split, int32_t B, int32_t T,
                                           int32_t n_head, int32_t* overflow_flag, void* stream) {
  T2H_REQUIRE(qkv && y_split, "t2h_mha_noncausal_split_f32: NULL pointer");
  T2H_REQUIRE(B > 0 && n_head > 0, "t2h_mha_noncausal_split_f32: empty problem");
  T2H_REQUIRE(T > 0 && T % QB == 0, "t2h_mha_noncausal_split_f32: T=%d must be a multiple of %d", T, QB);
  T2H_REQUIRE(t2h_aligned16(qkv) && t2h_aligned16(y_split), "t2h_mha_noncausal_split_f32: 16-byte alignment");
  const int C = n_head * HD;
  dim3 grid((T / QB) * n_head * B), block(512);
  int* ovf = overflow_flag;
  T2H_REQUIRE(ovf != nullptr, "t2h_mha_noncausal_split_f32: overflow_flag is NULL");
  hipLaunchKernelGGL(mha_kernel, grid, block, 0, static_cast<hipStream_t>(stream), qkv,
                     static_cast<float*>(nullptr), T, C, n_head, y_split, ovf);
  T2H_CHECK_LAUNCH("t2h_mha_noncausal_split_f32");
  return T2H_OK;
}

static int mha_split_launch(const uint16_t* qk_split, int32_t ld_cols, const uint16_t* vt, float* y, uint16_t* y_split,
                            float y_x8_scale, int32_t B, int32_t T, int32_t n_head, int32_t* overflow_flag, void* stream) {
  T2H_REQUIRE(qk_split && vt && (y || y_split), "t2h_mha_split_f32: NULL pointer");
  T2H_REQUIRE(B > 0 && n_head > 0, "t2h_mha_split_f32: empty problem");
  T2H_REQUIRE(T > 0 && T % QB == 0, "t2h_mha_split_f32: T=%d must be a multiple of %d", T, QB);
  const int C = n_head * HD;
  T2H_REQUIRE(ld_cols % 32 == 0 && ld_cols >= 2 * C, "t2h_mha_split_f32: ld_cols=%d must hold q and k (%d columns)",
              ld_cols, 2 * C);
  T2H_REQUIRE(t2h_aligned16(qk_split) && t2h_aligned16(vt) && (!y || t2h_aligned16(y)) &&
                  (!y_split || t2h_aligned16(y_split)),
              "t2h_mha_split_f32: 16-byte alignment");
  dim3 block(512);
  int* ovf = overflow_flag;
  T2H_REQUIRE(ovf != nullptr || y_split == nullptr, "t2h_mha_split_f32: overflow_flag is NULL (needed with y_split)");
  // Which form: rounds of the 256 CUs (one workgroup per CU at a time) x the measured length of a workgroup -- a
  // 256-query workgroup over all keys takes 5 units where a 128-query one over the two key halves takes 3
  // (profiles/r05_mha_all_keys.log).  B = 8: 256 x 128-query workgroups, one round; B >= 16: the all-keys form.
  const int64_t wg2 = (int64_t)(T / QB) * n_head * B, wg1 = T % 256 == 0 ? (int64_t)(T / 256) * n_head * B : 0;
  int form = g_force_mha_form;
  if (form == 0) form = (wg1 > 0 && ((wg1 + 255) / 256) * 5 < ((wg2 + 255) / 256) * 3) ? 1 : 2;
  T2H_REQUIRE(form == 2 || wg1 > 0, "t2h_mha_split_f32: the all-keys form needs T %% 256 == 0");
  if (form == 1)
    hipLaunchKernelGGL(mha_split_pipe_kernel<1>, dim3((unsigned)wg1), block, 0, static_cast<hipStream_t>(stream), qk_split,
                       ld_cols, vt, y, y_split, T, C, n_head, ovf, y_x8_scale);
  else
    hipLaunchKernelGGL(mha_split_pipe_kernel<2>, dim3((unsigned)wg2), block, 0, static_cast<hipStream_t>(stream), qk_split,
                       ld_cols, vt, y, y_split, T, C, n_head, ovf, y_x8_scale);
  T2H_CHECK_LAUNCH("t2h_mha_split_f32");
  return T2H_OK;
}

extern "C" int t2h_mha_split_f32(const uint16_t* qk_split, int32_t ld_cols, const uint16_t* vt, float* y,
                                 uint16_t* y_split, int32_t B, int32_t T, int32_t n_head, int32_t* overflow_flag,
                                 void* stream) {
  return mha_split_launch(qk_split, ld_cols, vt, y, y_split, 0.f, B, T, n_head, overflow_flag, stream);
}

extern "C" int t2h_mha_split_x8_f32(const uint16_t* qk_split, int32_t ld_cols, const uint16_t* vt, uint16_t* y_x8,
                                    float y_scale, int32_t B, int32_t T, int32_t n_head, int32_t* overflow_flag,
                                    void* stream) {
  T2H_REQUIRE(y_x8 != nullptr && y_scale > 0.f, "t2h_mha_split_x8_f32: y_x8 is NULL or y_scale <= 0");
  return mha_split_launch(qk_split, ld_cols, vt, nullptr, y_x8, y_scale, B, T, n_head, overflow_flag, stream);
}

#ifdef T2H_MHA_TIMING
extern "C" int t2h_debug_set_mha_timing_buffer(void* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_mha_timing), &dev_ptr, sizeof(void*));
}
#endif
