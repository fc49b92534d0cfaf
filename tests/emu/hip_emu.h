// Host emulation of ONE HIP workgroup at a time, for running a kernel's SOURCE on the CPU (test infrastructure:
// tests/test_conv_halo_emulated.py; nothing in the product builds or loads this).
//
// A launch runs its workgroups one after another; the threads of a workgroup are real OS threads, __syncthreads() is
// a pthread barrier, __shared__ storage is a function-local static (one workgroup alive at a time), and the matrix
// instruction is a wave-collective: 64 threads deposit their operand registers, meet at the wave's barrier, and every
// lane computes the 16 accumulator elements it owns.  Operand / accumulator lane layout of
// v_mfma_f32_32x32x16_f16 as the product kernels use it (conv_split.hip, gemm_split.hip -- parity-tested on the GPU):
//   A: lane l holds row (l & 31), k = 8 (l >> 5) .. + 7;   B: lane l holds column (l & 31), the same k;
//   D: lane l holds column (l & 31), rows (r & 3) + 8 (r >> 2) + 4 (l >> 5), r = 0 .. 15.
// What it checks: index algebra, LDS layout, buffer rotation, borders, the epilogue's row mapping -- with truly
// concurrent threads.  What it cannot check: timing, bank conflicts, the hardware's instruction semantics beyond the
// ones restated here.
#pragma once
#include <math.h>
#include <cmath>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
static thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
struct float2 {
  float x, y;
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static thread_local int emu_tid;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__
using std::max;
using std::min;
typedef void* hipStream_t;
typedef int hipError_t;
constexpr int hipSuccess = 0;
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emulated"; }

static pthread_barrier_t emu_block_bar;
static inline void __syncthreads() { pthread_barrier_wait(&emu_block_bar); }
static inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline uint32_t atomicOr(uint32_t* p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline uint32_t atomicMax(uint32_t* p, uint32_t v) {
  uint32_t old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return old;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
// explicitly rounded single operations (no contraction into an fma)
static inline float __fmul_rn(float a, float b) {
  volatile float r = a * b;
  return r;
}
static inline float __fadd_rn(float a, float b) {
  volatile float r = a + b;
  return r;
}

// builtins of the product's helpers that the emulated kernels do not execute (common.h parses them)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_logf(x) log2f(x) /* v_log_f32 is a base-2 logarithm (the hardware's is an approximation) */
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)

typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 emu_f16x8 __attribute__((ext_vector_type(8)));
struct emu_wave {
  _Float16 a[64][8], b[64][8];
  uint64_t shfl[64];
  pthread_barrier_t bar;
};
static emu_wave emu_waves[16];

// ---- the vector-memory request queue of a thread (= of its wave: every lane issues the same sequence).  The kernels'
// explicit requests (inline-assembly global loads into registers, LDS-DMA) and counted waits (s_waitcnt vmcnt(N): at
// most N requests still in flight, and they return in order) go through here.  Immediate mode executes a request when
// it is issued -- the earliest it can land.  Deferred mode (emu_set_deferred(1)) executes it only when a wait forces it
// (or the thread ends) -- the LATEST the hardware may land it: data that is consumed before the wait that covers it is
// stale, a count that is one too permissive gives a wrong result instead of a lucky one.
struct emu_vm_op {
  void* dst;
  const void* src;
  int n;
};
static thread_local std::vector<emu_vm_op> emu_vm_q;
static int emu_vm_deferred = 0;
extern "C" int emu_set_deferred(int on) {
  const int old = emu_vm_deferred;
  emu_vm_deferred = on;
  return old;
}
static inline void emu_vm_issue(void* dst, const void* src, int n) {
  if (!emu_vm_deferred) memcpy(dst, src, n);
  else emu_vm_q.push_back(emu_vm_op{dst, src, n});
}
static inline void emu_vm_wait(int allow) {
  size_t done = 0;
  while (emu_vm_q.size() - done > (size_t)allow) {
    const emu_vm_op& op = emu_vm_q[done++];
    memcpy(op.dst, op.src, op.n);
  }
  emu_vm_q.erase(emu_vm_q.begin(), emu_vm_q.begin() + done);
}

// all lanes of the calling wave are here (what lockstep execution gives the hardware for free)
static inline void emu_wave_sync() { pthread_barrier_wait(&emu_waves[emu_tid >> 6].bar); }

// wave-collective lane exchange (every lane of the wave must call it)
template <typename T>
static inline T emu_shfl_xor(T v, int mask) {
  static_assert(sizeof(T) <= 8, "4- and 8-byte values");
  emu_wave& W = emu_waves[emu_tid >> 6];
  const int lane = emu_tid & 63;
  memcpy(&W.shfl[lane], &v, sizeof(T));
  pthread_barrier_wait(&W.bar);
  T r;
  memcpy(&r, &W.shfl[lane ^ mask], sizeof(T));
  pthread_barrier_wait(&W.bar);
  return r;
}
#define __shfl_xor(v, o, w) emu_shfl_xor(v, o)
// wave vote: true on every lane if the predicate holds on any lane
static inline int __any(int pred) {
  int v = pred ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) v |= emu_shfl_xor(v, o);
  return v;
}
// the DPP controls the product's wave reductions use (common.h: quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror,
// row_mirror, all lanes enabled) and v_readlane_b32, as lane exchanges
static inline int emu_lane_from(int v, int src) {
  emu_wave& W = emu_waves[emu_tid >> 6];
  const int lane = emu_tid & 63;
  memcpy(&W.shfl[lane], &v, sizeof v);
  pthread_barrier_wait(&W.bar);
  int r;
  memcpy(&r, &W.shfl[src], sizeof r);
  pthread_barrier_wait(&W.bar);
  return r;
}
static inline int emu_update_dpp(int x, int ctrl) {
  const int lane = emu_tid & 63;
  int src = lane;
  if (ctrl == 0xB1) src = lane ^ 1;
  else if (ctrl == 0x4E) src = lane ^ 2;
  else if (ctrl == 0x141) src = (lane & ~7) | (7 - (lane & 7));
  else if (ctrl == 0x140) src = (lane & ~15) | (15 - (lane & 15));
  else abort();
  return emu_lane_from(x, src);
}
#define __builtin_amdgcn_update_dpp(old, x, ctrl, rm, bm, bc) emu_update_dpp(x, ctrl)
#define __builtin_amdgcn_readlane(x, i) emu_lane_from(x, i)

static inline emu_f32x16 emu_mfma_f32_32x32x16_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x16 c) {
  const int lane = emu_tid & 63;
  emu_wave& W = emu_waves[emu_tid >> 6];
  for (int e = 0; e < 8; ++e) {
    W.a[lane][e] = a[e];
    W.b[lane][e] = b[e];
  }
  pthread_barrier_wait(&W.bar);
  const int j = lane & 31, hh = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += (float)W.a[i + 32 * (k >> 3)][k & 7] * (float)W.b[j + 32 * (k >> 3)][k & 7];
    c[r] += s;
  }
  pthread_barrier_wait(&W.bar);
  return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu_mfma_f32_32x32x16_f16(a, b, c)

// v_mfma_f32_32x32x2_f32 (gemm.hip, spatial_attn.hip): A lane l = row (l & 31), k = l >> 5; B lane l = column (l & 31),
// the same k; D as above; two fused multiply-adds per element, k = 0 first
static inline emu_f32x16 emu_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c) {
  const int lane = emu_tid & 63;
  emu_wave& W = emu_waves[emu_tid >> 6];
  float* fa = reinterpret_cast<float*>(&W.a[0][0]);
  float* fb = reinterpret_cast<float*>(&W.b[0][0]);
  fa[lane] = a;
  fb[lane] = b;
  pthread_barrier_wait(&W.bar);
  const int j = lane & 31, hh = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
    c[r] = fmaf(fa[i], fb[j], c[r]);
    c[r] = fmaf(fa[i + 32], fb[j + 32], c[r]);
  }
  pthread_barrier_wait(&W.bar);
  return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma_f32_32x32x2f32(a, b, c)

// ---- OCP e4m3 (e4m3fn: bias 7, no infinities, 0x7f / 0xff = NaN, max 448) as gfx950's 8-bit conversions and matrix
// instructions read it
static inline float emu_e4m3_decode(uint8_t b) {
  const int e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 15 && m == 7) v = NAN;
  else if (e == 0) v = ldexpf((float)m, -9);
  else v = ldexpf(1.0f + (float)m / 8.0f, e - 7);
  return (b & 0x80) ? -v : v;
}
static inline uint8_t emu_e4m3_encode(float x) {  // round to nearest even; beyond +-448: the largest finite value
  const uint8_t sgn = std::signbit(x) ? 0x80 : 0;
  const float a = fabsf(x);
  if (a != a) return sgn | 0x7f;
  if (a < ldexpf(1.0f, -6)) {  // subnormal range: multiples of 2^-9
    const int q = (int)nearbyintf(ldexpf(a, 9));
    return sgn | (uint8_t)(q >= 8 ? 0x08 : q);
  }
  int e;
  const float fr = frexpf(a, &e);  // a = fr 2^e, fr in [0.5, 1)
  int ex = e - 1, q = (int)nearbyintf((fr * 2.0f - 1.0f) * 8.0f);
  if (q == 8) {
    q = 0;
    ++ex;
  }
  if (ex > 8 || (ex == 8 && q == 7)) return sgn | 0x7e;
  return sgn | (uint8_t)(((ex + 7) << 3) | q);
}
// v_cvt_pk_fp8_f32: two conversions into the low (hi = false) or high half of the old dword
static inline unsigned emu_cvt_pk_fp8_f32(float a, float b, unsigned old, bool hi) {
  const unsigned two = (unsigned)emu_e4m3_encode(a) | ((unsigned)emu_e4m3_encode(b) << 8);
  return hi ? (old & 0x0000ffffu) | (two << 16) : (old & 0xffff0000u) | two;
}
#define __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, hi) emu_cvt_pk_fp8_f32(a, b, w, hi)

// v_mfma_f32_16x16x32_f16 (gemm_split.hip's few-rows kernel): A lane l = row (l & 15), k = 8 (l >> 4) .. + 7; B lane l =
// column (l & 15), the same k; D register e of lane l = row 4 (l >> 4) + e, column (l & 15)
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
static inline emu_f32x4 emu_mfma_f32_16x16x32_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x4 c) {
  const int lane = emu_tid & 63;
  emu_wave& W = emu_waves[emu_tid >> 6];
  for (int e = 0; e < 8; ++e) {
    W.a[lane][e] = a[e];
    W.b[lane][e] = b[e];
  }
  pthread_barrier_wait(&W.bar);
  const int j = lane & 15, g = lane >> 4;
  for (int e = 0; e < 4; ++e) {
    const int i = 4 * g + e;
    float s = 0.f;
    for (int k = 0; k < 32; ++k) s += (float)W.a[i + 16 * (k >> 3)][k & 7] * (float)W.b[j + 16 * (k >> 3)][k & 7];
    c[e] += s;
  }
  pthread_barrier_wait(&W.bar);
  return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) emu_mfma_f32_16x16x32_f16(a, b, c)

// v_mfma_scale_f32_32x32x64_f8f6f4 / _16x16x128_f8f6f4 with e4m3 x e4m3 operands and unit scales (the only form the
// product uses): a lane supplies 32 bytes = its row's (column's) 32 consecutive k of the lane's k group -- group
// l >> 5 of two (32x32x64), l >> 4 of four (16x16x128); accumulator layouts as the fp16 instructions of the same shape
typedef int emu_i32x8 __attribute__((ext_vector_type(8)));
struct emu_wave8 {
  uint8_t a[64][32], b[64][32];
};
static emu_wave8 emu_waves8[16];
static inline void emu_deposit8(emu_i32x8 a, emu_i32x8 b) {
  const int lane = emu_tid & 63;
  emu_wave8& W8 = emu_waves8[emu_tid >> 6];
  memcpy(W8.a[lane], &a, 32);
  memcpy(W8.b[lane], &b, 32);
}
static inline emu_f32x16 emu_mfma_scale_32x32x64(emu_i32x8 a, emu_i32x8 b, emu_f32x16 c) {
  const int lane = emu_tid & 63;
  emu_wave& W = emu_waves[emu_tid >> 6];
  emu_wave8& W8 = emu_waves8[emu_tid >> 6];
  emu_deposit8(a, b);
  pthread_barrier_wait(&W.bar);
  const int j = lane & 31, hh = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
    float s = 0.f;
    for (int k = 0; k < 64; ++k) s += emu_e4m3_decode(W8.a[i + 32 * (k >> 5)][k & 31]) * emu_e4m3_decode(W8.b[j + 32 * (k >> 5)][k & 31]);
    c[r] += s;
  }
  pthread_barrier_wait(&W.bar);
  return c;
}
static inline emu_f32x4 emu_mfma_scale_16x16x128(emu_i32x8 a, emu_i32x8 b, emu_f32x4 c) {
  const int lane = emu_tid & 63;
  emu_wave& W = emu_waves[emu_tid >> 6];
  emu_wave8& W8 = emu_waves8[emu_tid >> 6];
  emu_deposit8(a, b);
  pthread_barrier_wait(&W.bar);
  const int j = lane & 15, g = lane >> 4;
  for (int e = 0; e < 4; ++e) {
    const int i = 4 * g + e;
    float s = 0.f;
    for (int k = 0; k < 128; ++k) s += emu_e4m3_decode(W8.a[i + 16 * (k >> 5)][k & 31]) * emu_e4m3_decode(W8.b[j + 16 * (k >> 5)][k & 31]);
    c[e] += s;
  }
  pthread_barrier_wait(&W.bar);
  return c;
}
#define __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, fa, fb, x, sa, y, sb) emu_mfma_scale_32x32x64(a, b, c)
#define __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, fa, fb, x, sa, y, sb) emu_mfma_scale_16x16x128(a, b, c)
#define __builtin_amdgcn_s_barrier() pthread_barrier_wait(&emu_block_bar)
// (attention.hip's barrier among the waves of a key half: a monotonic LDS counter, polled)
#include <sched.h>
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST)
#define __hip_atomic_load(p, order, scope) __atomic_load_n(p, __ATOMIC_SEQ_CST)
#define __builtin_amdgcn_s_sleep(n) sched_yield()
#define __builtin_amdgcn_s_setprio(n) ((void)0)
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_s_memrealtime() 0ull

template <typename K, typename... A>
static void emu_launch(K kernel, dim3 grid, dim3 block, A... args) {
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned b = 0; b < grid.x; ++b) {
        pthread_barrier_init(&emu_block_bar, nullptr, block.x);
        for (unsigned w = 0; w < (block.x + 63) / 64; ++w)
          pthread_barrier_init(&emu_waves[w].bar, nullptr, std::min(64u, block.x - 64 * w));
        std::vector<std::thread> th;
        for (unsigned t = 0; t < block.x; ++t)
          th.emplace_back([=] {
            threadIdx = dim3(t);
            blockIdx = dim3(b, by, bz);
            blockDim = block;
            gridDim = grid;
            emu_tid = (int)t;
            kernel(args...);
            emu_vm_wait(0);  // (a wave's requests complete before it ends)
          });
        for (auto& x : th) x.join();
        pthread_barrier_destroy(&emu_block_bar);
        for (unsigned w = 0; w < (block.x + 63) / 64; ++w) pthread_barrier_destroy(&emu_waves[w].bar);
      }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(kernel, grid, block, __VA_ARGS__)
typedef void* hipEvent_t;
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ev0, ev1, flags, ...) emu_launch(kernel, grid, block, __VA_ARGS__)

static char emu_err[512];
#include <stdarg.h>
void t2h_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(emu_err, sizeof emu_err, fmt, ap);
  va_end(ap);
}
extern "C" const char* emu_last_error(void) { return emu_err; }
