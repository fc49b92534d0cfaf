// Standalone calibration: sustained v_mfma_f32_32x32x2_f32 rate and shader clock
// on the target (pure-register loop, no memory traffic), with random operands.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const float* in, float* out, int iters, long long* cyc) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = in[threadIdx.x], b = in[threadIdx.x + 256];
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(int blocks, int iters, const float* din, float* dout, long long* dcyc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, 256>>>(din, dout, 10, dcyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<NACC><<<blocks, 256>>>(din, dout, iters, dcyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long cyc;
  hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
  double nm = (double)blocks * 4 * iters * 8 * NACC;  // MFMAs (per wave x 4 waves)
  double tf = nm * 4096.0 / (ms * 1e-3) / 1e12;
  printf("NACC=%d blocks=%d: %.3f ms, %.1f TFLOP/s, cycles/MFMA(wave0)=%.1f, shader clock ~%.0f MHz\n", NACC,
         blocks, ms, tf, (double)cyc / (iters * 8.0 * NACC), (double)cyc / (ms * 1e-3) / 1e6);
}

int main() {
  float* hin = (float*)malloc(512 * 4);
  for (int i = 0; i < 512; ++i) hin[i] = (float)rand() / RAND_MAX - 0.5f;
  float *din, *dout;
  long long* dcyc;
  hipMalloc(&din, 512 * 4);
  hipMalloc(&dout, 4096 * 256 * 4);
  hipMalloc(&dcyc, 8);
  hipMemcpy(din, hin, 512 * 4, hipMemcpyHostToDevice);
  run<1>(256, 20000, din, dout, dcyc);
  run<2>(256, 10000, din, dout, dcyc);
  run<4>(256, 5000, din, dout, dcyc);
  run<2>(512, 10000, din, dout, dcyc);
  run<2>(1024, 10000, din, dout, dcyc);
  return 0;
}
