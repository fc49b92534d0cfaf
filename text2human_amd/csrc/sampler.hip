// Index-sampler glue kernels: 4-way embedding sum, the unmasking schedule step
// and the texture-routed categorical sampling tail.
#include "common.h"

namespace {

// x[row] = ((tok_emb[idx] + pos_emb[t]) + segm_emb[segm]) + texture_emb[tex]
// -- same association order as models/archs/transformer_arch.py:266.
__global__ void embed_sum4_kernel(const int64_t* __restrict__ idx, const int64_t* __restrict__ segm,
                                  const int64_t* __restrict__ tex, const float* __restrict__ tok_emb,
                                  const float* __restrict__ pos_emb, const float* __restrict__ segm_emb,
                                  const float* __restrict__ tex_emb, float* __restrict__ x, int T,
                                  int C) {
  const int row = blockIdx.x;
  const int t = row % T;
  const float* a = tok_emb + idx[row] * C;
  const float* b = pos_emb + (int64_t)t * C;
  const float* c = segm_emb + segm[row] * C;
  const float* d = tex_emb + tex[row] * C;
  float* o = x + (int64_t)row * C;
  for (int i = threadIdx.x * 4; i < C; i += blockDim.x * 4) {
    const f32x4 va = *reinterpret_cast<const f32x4*>(a + i);
    const f32x4 vb = *reinterpret_cast<const f32x4*>(b + i);
    const f32x4 vc = *reinterpret_cast<const f32x4*>(c + i);
    const f32x4 vd = *reinterpret_cast<const f32x4*>(d + i);
    *reinterpret_cast<f32x4*>(o + i) = ((va + vb) + vc) + vd;
  }
}

// changes = rand < 1/t ; changes &= ~unmasked ; unmasked |= changes
// (models/sample_model.py:286-292).  `1 / t.float()` is an fp32 reciprocal.
// With changed_rows: the changed tokens are also appended to that list (in no particular
// order -- every row is sampled independently) and counted in head_count[n_heads].
__global__ void unmask_step_kernel(const float* __restrict__ rnd, float thresh,
                                   uint8_t* __restrict__ unmasked, uint8_t* __restrict__ changes,
                                   const int64_t* __restrict__ tex, int* __restrict__ head_count,
                                   int n, int* __restrict__ changed_rows, int n_heads) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool um = unmasked[i] != 0;
  const bool ch = (rnd[i] < thresh) && !um;
  changes[i] = ch ? 1 : 0;
  if (ch) {
    unmasked[i] = 1;
    atomicAdd(head_count + (int)tex[i], 1);
    if (changed_rows) changed_rows[atomicAdd(head_count + n_heads, 1)] = i;
  }
}

// One workgroup per token row; rows that are not (changed && of this head's
// texture) exit at once.  LN_f -> 512->n_class head (wave-cooperative dot
// products, coalesced weight rows) -> exponential-race argmax.
constexpr int SH_THREADS = 1024;

template <int C>
__device__ __forceinline__ void sample_row(float* lds, int row, const float* __restrict__ hidden,
                                           const float* __restrict__ g, const float* __restrict__ bta,
                                           const float* __restrict__ w, const float* __restrict__ expo, int head,
                                           float temp, int64_t* __restrict__ x_t,
                                           int64_t* __restrict__ out_idx, int n_class) {
  constexpr int VPL = C / 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = hidden + (int64_t)row * C;
  f32x4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i] = *reinterpret_cast<const f32x4*>(xr + i * 256 + lane * 4);
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wave_sum(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[i][e] - mean;
      q = fmaf(d, d, q);
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + 1e-5f);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const f32x4 gg = *reinterpret_cast<const f32x4*>(g + i * 256 + lane * 4);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bta + i * 256 + lane * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
  }
  // logits: NW waves, each takes 4 classes per iteration so that 8 independent
  // 1-KiB weight-row loads are in flight per wave (the loop is latency bound).
  constexpr int NW = SH_THREADS / 64;
  for (int j0 = wave * 4; j0 < n_class; j0 += NW * 4) {
    float acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = min(j0 + u, n_class - 1);
      const float* wr = w + (int64_t)j * C;
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const f32x4 ww = *reinterpret_cast<const f32x4*>(wr + i * 256 + lane * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) a = fmaf(ww[e], v[i][e], a);
      }
      acc[u] = a;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float r = wave_sum(acc[u]);
      if (lane == 0 && j0 + u < n_class) lds[j0 + u] = r / temp;  // (the reference divides: logits / temp, sample_model.py:303)
    }
  }
  __syncthreads();
  float* red = lds + n_class;
  float mx = -INFINITY;
  for (int j = tid; j < n_class; j += SH_THREADS) mx = fmaxf(mx, lds[j]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int k = 1; k < NW; ++k) mx = fmaxf(mx, red[k]);
  // argmax_j exp(l_j - max) / q_j  (first index wins ties)
  const float* er = expo + (int64_t)row * n_class;
  float best = -1.f;
  int best_j = 0x7fffffff;
  for (int j = tid; j < n_class; j += SH_THREADS) {
    const float sc = expf(lds[j] - mx) / er[j];
    if (sc > best) {
      best = sc;
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
  __syncthreads();
  int* redj = reinterpret_cast<int*>(red + NW);
  if (lane == 0) {
    red[wave] = best;
    redj[wave] = best_j;
  }
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < NW; ++k)
      if (red[k] > best || (red[k] == best && redj[k] < best_j)) {
        best = red[k];
        best_j = redj[k];
      }
    // all-NaN scores (only after a flagged split-precision overflow upstream): keep the token id
    // inside the embedding table so the run reaches the host-side overflow check instead of faulting
    if (best_j >= n_class) best_j = 0;
    x_t[row] = (int64_t)best_j + (int64_t)n_class * head;
    out_idx[row] = best_j;
  }
}

template <int C>
__global__ __launch_bounds__(SH_THREADS) void sample_head_kernel(
    const float* __restrict__ hidden, const float* __restrict__ g, const float* __restrict__ bta,
    const float* __restrict__ w, const float* __restrict__ expo, const uint8_t* __restrict__ changes,
    const int64_t* __restrict__ tex, int head, float temp, int64_t* __restrict__ x_t,
    int64_t* __restrict__ out_idx, int n_class) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // n_class logits + 2*NW reduce slots
  const int row = blockIdx.x;
  if (!changes[row] || (int)tex[row] != head) return;
  sample_row<C>(lds, row, hidden, g, bta, w, expo, head, temp, x_t, out_idx, n_class);
}

// ---- sampler training-time forward (models/transformer_model.py:212-274, forward only)
// q_sample: mask = u < t/T (fp32 division, like t.float() / num_timesteps); x_t = mask ? mask_id : x_0
__global__ void q_sample_kernel(const int64_t* __restrict__ x0, const float* __restrict__ u,
                                const int64_t* __restrict__ t, float num_timesteps, int64_t mask_id,
                                int64_t* __restrict__ x_t, uint8_t* __restrict__ mask, int T, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool m = u[i] < (float)t[i / T] / num_timesteps;
  mask[i] = m ? 1 : 0;
  x_t[i] = m ? mask_id : x0[i];
}

// Cross entropy of the masked tokens.  F.cross_entropy(..., ignore_index=-1) summed over the
// 18 heads only ever sees the head of the token's own texture (every other head's target is
// -1), so: one workgroup per token, rows that are unmasked or have target -1 contribute 0,
// the others LN_f -> head of their texture -> logsumexp(logits) - logits[target].
template <int C>
__global__ __launch_bounds__(SH_THREADS) void masked_ce_kernel(
    const float* __restrict__ hidden, const float* __restrict__ g, const float* __restrict__ bta,
    const float* __restrict__ w_heads, const int64_t* __restrict__ tex, const uint8_t* __restrict__ mask,
    const int64_t* __restrict__ gt_lists, float* __restrict__ ce, int n, int n_class, int n_heads) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = (int)tex[row];
  const int64_t target = (head >= 0 && head < n_heads) ? gt_lists[(int64_t)head * n + row] : -1;
  if (!mask[row] || target < 0) {
    if (tid == 0) ce[row] = 0.f;
    return;
  }
  constexpr int VPL = C / 256;
  const float* xr = hidden + (int64_t)row * C;
  f32x4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i] = *reinterpret_cast<const f32x4*>(xr + i * 256 + lane * 4);
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wave_sum(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[i][e] - mean;
      q = fmaf(d, d, q);
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + 1e-5f);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const f32x4 gg = *reinterpret_cast<const f32x4*>(g + i * 256 + lane * 4);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bta + i * 256 + lane * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
  }
  constexpr int NW = SH_THREADS / 64;
  const float* w = w_heads + (int64_t)head * n_class * C;
  for (int j = wave; j < n_class; j += NW) {
    const float* wr = w + (int64_t)j * C;
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const f32x4 ww = *reinterpret_cast<const f32x4*>(wr + i * 256 + lane * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) a = fmaf(ww[e], v[i][e], a);
    }
    a = wave_sum(a);
    if (lane == 0) lds[j] = a;
  }
  __syncthreads();
  float* red = lds + n_class;
  float mx = -INFINITY;
  for (int j = tid; j < n_class; j += SH_THREADS) mx = fmaxf(mx, lds[j]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int k = 1; k < NW; ++k) mx = fmaxf(mx, red[k]);
  float se = 0.f;
  for (int j = tid; j < n_class; j += SH_THREADS) se += expf(lds[j] - mx);
  se = wave_sum(se);
  __syncthreads();
  if (lane == 0) red[wave] = se;
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
    for (int k = 0; k < NW; ++k) tot += red[k];
    ce[row] = (mx + logf(tot)) - lds[target];
  }
}

// out[b] = sum_t x[b][t] in a fixed order (one workgroup per segment)
__global__ __launch_bounds__(256) void segment_sum_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          int T) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int i = tid; i < T; i += 256) s += x[(int64_t)b * T + i];
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) out[b] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- torch's `exponential_` draw, element by element.  The reference's Categorical.sample() draws
// a FULL [n, n_class] Exp(1) tensor per active head from the device Philox generator
// (models/sample_model.py:305-306 -> multinomial -> exponential_) although only the changed rows of
// that head use it.  ATen's kernel (distribution_elementwise_grid_stride_kernel, unroll 4, block 256,
// grid G) gives element e of the tensor the value
//     u = uniform( philox4x32_10(key = seed, counter = {offset/4 + it, subsequence = idx})[ii] ),
//     idx = e % (256 G), it = e / (256 G) / 4, ii = e / (256 G) % 4,
//     q = u >= 1 - 2^-24 ? 2^-24 : -log(u)
// (curand_init(seed, idx, offset) / curand_uniform4 = rocRAND's philox4x32_10 engine and
// uniform_distribution; transformation::exponential of ATen/core/TransformationHelper.h).  Computing
// it here for the (row, class) pairs that are actually used is bit-identical
// (tests/test_gpu_kernels.py: full-tensor equality with torch) and removes a 16 MB draw per active
// head and step.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t m0 = (uint64_t)0xD2511F53u * c[0], m1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t hi0 = (uint32_t)(m0 >> 32), lo0 = (uint32_t)m0, hi1 = (uint32_t)(m1 >> 32), lo1 = (uint32_t)m1;
  const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
  c[0] = n0;
  c[1] = lo1;
  c[2] = n2;
  c[3] = lo0;
}
// log(x) exactly as it is compiled into ATen's kernel: LLVM's f32 log lowering for targets with fast FMA
// (y = v_log_f32(x); r = y c; r + fma(y, cc, fma(y, c, -r)), c + cc = ln 2 to 49 bits).  Contraction
// must stay off: fusing the last add into fma(c, y, t) counts the rounding error of r twice and is off
// by one ulp for a third of the arguments (ocml's logf of ROCm 7.2 differs from ATen's build the same way).
__device__ __forceinline__ float aten_logf(float x) {
#pragma clang fp contract(off)
  const float y = __builtin_amdgcn_logf(x), c = 0x1.62e42ep-1f, cc = 0x1.efa39ep-25f;
  const float r = y * c;
  const float t = __builtin_fmaf(y, cc, __builtin_fmaf(y, c, -r));
  return r + t;
}
__device__ __forceinline__ float torch_exponential_at(uint64_t seed, uint64_t offset, uint32_t grid_threads, uint64_t e) {
  const uint64_t idx = e % grid_threads, m = e / grid_threads;
  const uint64_t ctr = offset / 4 + (m >> 2);
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)idx, (uint32_t)(idx >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const uint32_t v = c[m & 3];
  const float u = __builtin_fmaf((float)v, 2.3283064365386963e-10f, 2.3283064365386963e-10f);  // (0, 1]
  const float lg = u >= 1.0f - 5.9604644775390625e-8f ? -5.9604644775390625e-8f : aten_logf(u);
  return -lg;
}

__global__ void philox_exponential_kernel(uint64_t seed, uint64_t offset, uint32_t grid_threads, float* __restrict__ out,
                                          int64_t numel) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < numel) out[e] = torch_exponential_at(seed, offset, grid_threads, (uint64_t)e);
}

// ---- torch's `rand` draw, element by element: the same engine / element <-> (thread, counter)
// mapping as above (distribution_elementwise_grid_stride_kernel, unroll 4) with ATen's uniform
// transform (uniform_kernel of ATen/native/cuda/DistributionTemplates.h, from = 0, to = 1):
// value = u * 1 + 0 with u = curand_uniform in (0, 1], and the bounds reversed: value == 1 -> 0.
__device__ __forceinline__ float torch_uniform_at(uint64_t seed, uint64_t offset, uint32_t grid_threads, uint64_t e) {
  const uint64_t idx = e % grid_threads, m = e / grid_threads;
  const uint64_t ctr = offset / 4 + (m >> 2);
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)idx, (uint32_t)(idx >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const uint32_t v = c[m & 3];
  const float u = __builtin_fmaf((float)v, 2.3283064365386963e-10f, 2.3283064365386963e-10f);  // (0, 1]
  return u == 1.0f ? 0.0f : u;
}

__global__ void philox_uniform_kernel(uint64_t seed, uint64_t offset, uint32_t grid_threads, float* __restrict__ out,
                                      int64_t numel) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < numel) out[e] = torch_uniform_at(seed, offset, grid_threads, (uint64_t)e);
}

// ---- the WHOLE unmasking schedule of a sampling run in one launch.  Which token is unmasked at
// which step (models/sample_model.py:286-292) depends on the `rand` draws and on nothing the
// transformer computes; the generator offset of every draw depends on the schedule only through
// the number of heads that sample at a step (one full exponential_ draw per ACTIVE head, :301-306).
// So the `rand` draws are reproduced here, step after step, each at the offset the reference's
// generator would hold: off(t-1) = off(t) + rand_inc + popcount(active heads at t) * expo_inc.
// One workgroup (the steps are sequential and a step is n / 1024 Philox blocks per thread);
// outputs: step_of_row[i] = the step at which token row i changes (every row changes exactly once:
// at t = 1 the threshold is 1), head_mask[t] = bit h set iff head h samples at step t.
constexpr int SCHED_THREADS = 1024, SCHED_MAX_STEPS = 4096;
__global__ __launch_bounds__(SCHED_THREADS) void unmask_schedule_kernel(
    uint64_t seed, uint64_t offset, uint32_t rand_grid_threads, uint32_t rand_inc, uint32_t expo_inc,
    const int64_t* __restrict__ tex, int n, int steps, int32_t* __restrict__ step_of_row,
    uint32_t* __restrict__ head_mask) {
  __shared__ uint32_t mask_s[SCHED_MAX_STEPS + 1];  // one word per step: no reset, one barrier per step
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < n; e += SCHED_THREADS) step_of_row[e] = 0;
  for (int t = tid; t <= steps; t += SCHED_THREADS) mask_s[t] = 0;
  __syncthreads();
  uint64_t off = offset;
  for (int t = steps; t >= 1; --t) {
    const float thresh = 1.0f / (float)t;  // `1 / t.float()`: an fp32 reciprocal
    uint32_t m = 0;
    for (int e = tid; e < n; e += SCHED_THREADS) {
      if (step_of_row[e] != 0) continue;  // already unmasked (written by this same thread)
      if (torch_uniform_at(seed, off, rand_grid_threads, (uint64_t)e) < thresh) {
        step_of_row[e] = t;
        m |= 1u << (int)tex[e];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o, 64);
    if (lane == 0 && m) atomicOr(&mask_s[t], m);
    __syncthreads();
    off += (uint64_t)rand_inc + (uint64_t)__popc(mask_s[t]) * expo_inc;
  }
  for (int t = tid; t <= steps; t += SCHED_THREADS) head_mask[t] = mask_s[t];
}

// ---- round cursor of a pre-computed schedule (engine.sample_tokens, graph replay): copies round
// r = *round_ctr of the padded [rounds][maxr] tables into fixed staging buffers and advances the
// counter, so that every round of a sampling run is the SAME launch sequence with the same arguments
// (capturable once, replayed per round) while the rows / generator offsets it works on change.
__global__ void schedule_advance_kernel(const int32_t* __restrict__ rows_tbl, const int64_t* __restrict__ aux64_tbl,
                                        const int32_t* __restrict__ aux32_tbl, int32_t* __restrict__ round_ctr,
                                        int32_t* __restrict__ cur_rows, int64_t* __restrict__ cur_aux64,
                                        int32_t* __restrict__ cur_aux32, int maxr) {
  const int r = *round_ctr;
  for (int i = threadIdx.x; i < maxr; i += blockDim.x) {
    const int64_t j = (int64_t)r * maxr + i;
    cur_rows[i] = rows_tbl[j];
    if (aux64_tbl) cur_aux64[i] = aux64_tbl[j];
    if (aux32_tbl) cur_aux32[i] = aux32_tbl[j];
  }
  __syncthreads();  // every thread has read r
  if (threadIdx.x == 0) *round_ctr = r + 1;
}

// ---- two-launch form of the same tail (t2h_sample_heads with a logits workspace).  One workgroup
// per changed row streams 2 MB of head weights by itself (~50 us per step with ~16 rows on 16 CUs);
// here SL_SPLIT workgroups per row take n_class / SL_SPLIT classes each (LN_f recomputed per
// workgroup: 2 KB), then a second launch does max / exponential race over the row's logits.
// Every logit is the same per-lane fma chain + wave butterfly as in sample_row, so the results are
// bit-identical to the one-launch form.
constexpr int SL_SPLIT = 8, SL_THREADS = 256;

template <int C>
__global__ __launch_bounds__(SL_THREADS) void sample_logits_kernel(const t2h_sample_heads_args a, float* __restrict__ ws) {
  constexpr int VPL = C / 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slot = blockIdx.x / SL_SPLIT, part = blockIdx.x - slot * SL_SPLIT;
  const int row = a.rows[slot];
  const int head = (int)a.tex[row];
  const float* xr = a.hidden + (int64_t)(a.hidden_compact ? slot : row) * C;
  f32x4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i] = *reinterpret_cast<const f32x4*>(xr + i * 256 + lane * 4);
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wave_sum(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[i][e] - mean;
      q = fmaf(d, d, q);
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + 1e-5f);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const f32x4 gg = *reinterpret_cast<const f32x4*>(a.lnf_gamma + i * 256 + lane * 4);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(a.lnf_beta + i * 256 + lane * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
  }
  const float temp = a.temp;
  const float* w = a.w_heads + (int64_t)head * a.n_class * C;
  const int per = (a.n_class + SL_SPLIT - 1) / SL_SPLIT;
  const int j_end = min(a.n_class, (part + 1) * per);
  constexpr int NW = SL_THREADS / 64;
  for (int j0 = part * per + wave * 4; j0 < j_end; j0 += NW * 4) {
    float acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = min(j0 + u, a.n_class - 1);
      const float* wr = w + (int64_t)j * C;
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const f32x4 ww = *reinterpret_cast<const f32x4*>(wr + i * 256 + lane * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) t = fmaf(ww[e], v[i][e], t);
      }
      acc[u] = t;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float r = wave_sum(acc[u]);
      if (lane == 0 && j0 + u < j_end) ws[(int64_t)slot * a.n_class + j0 + u] = r / temp;  // (divides, like the reference)
    }
  }
}

__global__ __launch_bounds__(SH_THREADS) void sample_pick_kernel(const t2h_sample_heads_args a, const float* __restrict__ ws) {
  __shared__ float red[2 * (SH_THREADS / 64)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = SH_THREADS / 64;
  const int slot = blockIdx.x, row = a.rows[slot];
  const int head = (int)a.tex[row];
  const float* lg = ws + (int64_t)slot * a.n_class;
  float mx = -INFINITY;
  for (int j = tid; j < a.n_class; j += SH_THREADS) mx = fmaxf(mx, lg[j]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int k = 1; k < NW; ++k) mx = fmaxf(mx, red[k]);
  // noise of this row: explicit compact rows (expo_rows[expo_slot[slot]]), the head's explicit full
  // tensor, or computed -- at the row's own generator offset when the list mixes steps
  const float* er = a.expo_rows ? a.expo_rows + (int64_t)(a.expo_slot ? a.expo_slot[slot] : slot) * a.n_class
                    : a.philox_grid_threads ? nullptr
                                            : a.expo[head] + (int64_t)row * a.n_class;
  const uint64_t poff = a.row_philox_offset ? a.row_philox_offset[slot] : a.philox_offset[head];
  const uint64_t pseed = a.philox_seed_dev ? *a.philox_seed_dev : a.philox_seed;
  // the element of the reference's [n, n_class] draw this row owns: its row THERE (the host may have reordered the
  // samples of the batch, rng_rows) -- by default the row itself
  const int rng_row = a.rng_rows ? a.rng_rows[slot] : row;
  float best = -1.f;
  int best_j = 0x7fffffff;
  for (int j = tid; j < a.n_class; j += SH_THREADS) {
    const float q = er ? er[j]
                       : torch_exponential_at(pseed, poff, a.philox_grid_threads,
                                              (uint64_t)rng_row * a.n_class + j);
    const float sc = expf(lg[j] - mx) / q;
    if (sc > best) {
      best = sc;
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
  __syncthreads();
  int* redj = reinterpret_cast<int*>(red + NW);
  if (lane == 0) {
    red[wave] = best;
    redj[wave] = best_j;
  }
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < NW; ++k)
      if (red[k] > best || (red[k] == best && redj[k] < best_j)) {
        best = red[k];
        best_j = redj[k];
      }
    if (best_j >= a.n_class) best_j = 0;  // all-NaN scores: see sample_row
    a.x_t[row] = (int64_t)best_j + (int64_t)a.n_class * head;
    a.out_idx[(int64_t)head * a.n + row] = best_j;
  }
}

// All heads in one launch: one workgroup per CHANGED token (compact list from
// unmask_step), which picks the head / noise tensor of its own texture.
template <int C>
__global__ __launch_bounds__(SH_THREADS) void sample_heads_kernel(const t2h_sample_heads_args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int row = a.rows[blockIdx.x];
  const int head = (int)a.tex[row];
  const float* expo = a.expo[head];
  if (expo == nullptr) return;  // cannot happen: a head with changed tokens always drew its noise
  sample_row<C>(lds, row, a.hidden, a.lnf_gamma, a.lnf_beta, a.w_heads + (int64_t)head * a.n_class * C, expo, head,
                a.temp, a.x_t, a.out_idx + (int64_t)head * a.n, a.n_class);  // (full hidden only)
}

}  // namespace

extern "C" int t2h_embed_sum4_f32(const int64_t* idx, const int64_t* segm, const int64_t* tex,
                                  const float* tok_emb, const float* pos_emb, const float* segm_emb,
                                  const float* tex_emb, float* x, int32_t B, int32_t T, int32_t C,
                                  void* stream) {
  T2H_REQUIRE(idx && segm && tex && tok_emb && pos_emb && segm_emb && tex_emb && x,
              "t2h_embed_sum4_f32: NULL pointer");
  T2H_REQUIRE(B > 0 && T > 0 && C > 0 && C % 4 == 0, "t2h_embed_sum4_f32: bad shape");
  hipLaunchKernelGGL(embed_sum4_kernel, dim3(B * T), dim3(128), 0, static_cast<hipStream_t>(stream),
                     idx, segm, tex, tok_emb, pos_emb, segm_emb, tex_emb, x, T, C);
  T2H_CHECK_LAUNCH("t2h_embed_sum4_f32");
  return T2H_OK;
}

extern "C" int t2h_unmask_step(const float* rnd, int32_t t, uint8_t* unmasked, uint8_t* changes,
                               const int64_t* tex, int32_t* head_count, int32_t n, int32_t* changed_rows,
                               int32_t n_heads, void* stream) {
  T2H_REQUIRE(rnd && unmasked && changes && tex && head_count, "t2h_unmask_step: NULL pointer");
  T2H_REQUIRE(t >= 1 && n > 0 && n_heads >= 0, "t2h_unmask_step: t=%d n=%d", t, n);
  const float thresh = 1.0f / (float)t;
  hipLaunchKernelGGL(unmask_step_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), rnd, thresh, unmasked, changes, tex,
                     head_count, n, changed_rows, n_heads);
  T2H_CHECK_LAUNCH("t2h_unmask_step");
  return T2H_OK;
}

extern "C" int t2h_sample_head(const float* hidden, const float* lnf_gamma, const float* lnf_beta,
                               const float* w_head, const float* expo, const uint8_t* changes,
                               const int64_t* tex, int32_t head, float temp, int64_t* x_t,
                               int64_t* out_idx, int32_t n, int32_t C, int32_t n_class,
                               void* stream) {
  T2H_REQUIRE(hidden && lnf_gamma && lnf_beta && w_head && expo && changes && tex && x_t && out_idx,
              "t2h_sample_head: NULL pointer");
  T2H_REQUIRE(n > 0 && n_class > 0 && temp > 0.f, "t2h_sample_head: bad arguments");
  T2H_REQUIRE(C == 512, "t2h_sample_head: C=%d unsupported (512)", C);
  const size_t lds = (size_t)(n_class + 2 * (SH_THREADS / 64)) * sizeof(float);
  hipLaunchKernelGGL(sample_head_kernel<512>, dim3(n), dim3(SH_THREADS), lds,
                     static_cast<hipStream_t>(stream), hidden, lnf_gamma, lnf_beta, w_head, expo,
                     changes, tex, head, temp, x_t, out_idx, n_class);
  T2H_CHECK_LAUNCH("t2h_sample_head");
  return T2H_OK;
}

extern "C" int t2h_sample_heads(const t2h_sample_heads_args* args, void* stream) {
  T2H_REQUIRE(args != nullptr, "t2h_sample_heads: args is NULL");
  const t2h_sample_heads_args a = *args;
  T2H_REQUIRE(a.hidden && a.lnf_gamma && a.lnf_beta && a.w_heads && a.rows && a.tex && a.x_t && a.out_idx,
              "t2h_sample_heads: NULL pointer");
  T2H_REQUIRE(a.n > 0 && a.n_class > 0 && a.temp > 0.f && a.n_rows >= 0 && a.n_heads > 0 &&
                  a.n_heads <= T2H_MAX_HEADS,
              "t2h_sample_heads: bad arguments");
  T2H_REQUIRE(a.C == 512, "t2h_sample_heads: C=%d unsupported (512)", a.C);
  T2H_REQUIRE(!a.hidden_compact || a.logits_ws != nullptr, "t2h_sample_heads: compact hidden needs the two-launch form");
  T2H_REQUIRE(a.philox_grid_threads == 0 || a.logits_ws != nullptr,
              "t2h_sample_heads: the in-kernel exponential_ draw needs the two-launch form (logits_ws)");
  T2H_REQUIRE((a.row_philox_offset == nullptr || a.philox_grid_threads != 0) &&
                  ((a.expo_rows == nullptr && a.row_philox_offset == nullptr) || a.logits_ws != nullptr),
              "t2h_sample_heads: per-row noise (row_philox_offset / expo_rows) needs the two-launch form and, for "
              "offsets, philox_grid_threads");
  if (a.n_rows == 0) return T2H_OK;
  if (a.logits_ws) {  // two launches, SL_SPLIT workgroups per row stream the head weights
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sample_logits_kernel<512>, dim3(a.n_rows * SL_SPLIT), dim3(SL_THREADS), 0, s, a, a.logits_ws);
    hipLaunchKernelGGL(sample_pick_kernel, dim3(a.n_rows), dim3(SH_THREADS), 0, s, a, a.logits_ws);
    T2H_CHECK_LAUNCH("t2h_sample_heads");
    return T2H_OK;
  }
  const size_t lds = (size_t)(a.n_class + 2 * (SH_THREADS / 64)) * sizeof(float);
  hipLaunchKernelGGL(sample_heads_kernel<512>, dim3(a.n_rows), dim3(SH_THREADS), lds,
                     static_cast<hipStream_t>(stream), a);
  T2H_CHECK_LAUNCH("t2h_sample_heads");
  return T2H_OK;
}

extern "C" int t2h_q_sample(const int64_t* x0, const float* u, const int64_t* t, int32_t num_timesteps,
                            int64_t mask_id, int64_t* x_t, uint8_t* mask, int32_t B, int32_t T, void* stream) {
  T2H_REQUIRE(x0 && u && t && x_t && mask, "t2h_q_sample: NULL pointer");
  T2H_REQUIRE(B > 0 && T > 0 && num_timesteps > 0, "t2h_q_sample: bad arguments");
  const int n = B * T;
  hipLaunchKernelGGL(q_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), x0, u,
                     t, (float)num_timesteps, mask_id, x_t, mask, T, n);
  T2H_CHECK_LAUNCH("t2h_q_sample");
  return T2H_OK;
}

extern "C" int t2h_masked_ce_heads(const float* hidden, const float* lnf_gamma, const float* lnf_beta,
                                   const float* w_heads, const int64_t* tex, const uint8_t* mask,
                                   const int64_t* gt_lists, float* ce_rows, float* ce_samples, int32_t B, int32_t T,
                                   int32_t C, int32_t n_class, int32_t n_heads, void* stream) {
  T2H_REQUIRE(hidden && lnf_gamma && lnf_beta && w_heads && tex && mask && gt_lists && ce_rows && ce_samples,
              "t2h_masked_ce_heads: NULL pointer");
  T2H_REQUIRE(B > 0 && T > 0 && n_class > 0 && n_heads > 0, "t2h_masked_ce_heads: bad arguments");
  T2H_REQUIRE(C == 512, "t2h_masked_ce_heads: C=%d unsupported (512)", C);
  const size_t lds = (size_t)(n_class + 2 * (SH_THREADS / 64)) * sizeof(float);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(masked_ce_kernel<512>, dim3(B * T), dim3(SH_THREADS), lds, s, hidden, lnf_gamma, lnf_beta,
                     w_heads, tex, mask, gt_lists, ce_rows, B * T, n_class, n_heads);
  hipLaunchKernelGGL(segment_sum_kernel, dim3(B), dim3(256), 0, s, ce_rows, ce_samples, T);
  T2H_CHECK_LAUNCH("t2h_masked_ce_heads");
  return T2H_OK;
}

extern "C" int t2h_schedule_advance(const int32_t* rows_tbl, const int64_t* aux64_tbl, const int32_t* aux32_tbl,
                                    int32_t* round_ctr, int32_t* cur_rows, int64_t* cur_aux64, int32_t* cur_aux32,
                                    int32_t maxr, void* stream) {
  T2H_REQUIRE(rows_tbl && round_ctr && cur_rows && maxr > 0 && (!aux64_tbl || cur_aux64) && (!aux32_tbl || cur_aux32),
              "t2h_schedule_advance: bad arguments");
  hipLaunchKernelGGL(schedule_advance_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), rows_tbl,
                     aux64_tbl, aux32_tbl, round_ctr, cur_rows, cur_aux64, cur_aux32, maxr);
  T2H_CHECK_LAUNCH("t2h_schedule_advance");
  return T2H_OK;
}

extern "C" int t2h_philox_uniform_f32(uint64_t seed, uint64_t offset, uint32_t grid_threads, float* out,
                                      int64_t numel, void* stream) {
  T2H_REQUIRE(out && numel > 0 && grid_threads > 0 && offset % 4 == 0, "t2h_philox_uniform_f32: bad arguments");
  hipLaunchKernelGGL(philox_uniform_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), seed, offset, grid_threads, out, numel);
  T2H_CHECK_LAUNCH("t2h_philox_uniform_f32");
  return T2H_OK;
}

extern "C" int t2h_unmask_schedule(uint64_t seed, uint64_t offset, uint32_t rand_grid_threads, uint32_t rand_inc,
                                   uint32_t expo_inc, const int64_t* tex, int32_t n, int32_t steps, int32_t n_heads,
                                   int32_t* step_of_row, uint32_t* head_mask, void* stream) {
  T2H_REQUIRE(tex && step_of_row && head_mask, "t2h_unmask_schedule: NULL pointer");
  T2H_REQUIRE(n > 0 && steps >= 1 && steps <= SCHED_MAX_STEPS && n_heads > 0 && n_heads <= T2H_MAX_HEADS &&
                  rand_grid_threads > 0 && offset % 4 == 0 && rand_inc % 4 == 0 && expo_inc % 4 == 0,
              "t2h_unmask_schedule: bad arguments (n=%d steps=%d n_heads=%d)", n, steps, n_heads);
  hipLaunchKernelGGL(unmask_schedule_kernel, dim3(1), dim3(SCHED_THREADS), 0, static_cast<hipStream_t>(stream), seed,
                     offset, rand_grid_threads, rand_inc, expo_inc, tex, n, steps, step_of_row, head_mask);
  T2H_CHECK_LAUNCH("t2h_unmask_schedule");
  return T2H_OK;
}

extern "C" int t2h_philox_exponential_f32(uint64_t seed, uint64_t offset, uint32_t grid_threads, float* out,
                                          int64_t numel, void* stream) {
  T2H_REQUIRE(out && numel > 0 && grid_threads > 0 && offset % 4 == 0, "t2h_philox_exponential_f32: bad arguments");
  hipLaunchKernelGGL(philox_exponential_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), seed, offset, grid_threads, out, numel);
  T2H_CHECK_LAUNCH("t2h_philox_exponential_f32");
  return T2H_OK;
}
