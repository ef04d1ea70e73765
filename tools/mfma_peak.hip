// tools/mfma_peak.hip -- what the fp32 MFMA pipe sustains with nothing else going on: 4 independent 32x32x2
// accumulator chains per wave (the GEMM kernel's shape), W waves per SIMD, no memory traffic; then the same with
// 1 / 2 chains per wave (the 64 x 64-tile kernels: ONE dependent chain per wave) at 1 / 2 / 4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o tools/mfma_peak tools/mfma_peak.hip && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void spin(float *out, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int i = 0; i < 4; i++)
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; i++)
    for (int r = 0; r < 16; r++) s += acc[i][r];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NCH>
__global__ __launch_bounds__(256) void spin_chains(float *out, int iters, float a0, float b0) {
  f32x16 acc[NCH];
  for (int i = 0; i < NCH; i++)
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int i = 0; i < NCH; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NCH; i++)
    for (int r = 0; r < 16; r++) s += acc[i][r];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NCH>
__global__ __launch_bounds__(256) void spin_chains16(float *out, int iters, float a0, float b0) {
  f32x4 acc[NCH];
  for (int i = 0; i < NCH; i++)
    for (int r = 0; r < 4; r++) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int i = 0; i < NCH; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NCH; i++)
    for (int r = 0; r < 4; r++) s += acc[i][r];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NCH> void run_chains16(float *out, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 8000;
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(spin_chains16<NCH>, dim3(256), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  // one wave per SIMD: ns per dependent step = time / (iters * 16 * NCH) when the chains serialise
  printf("16x16x4: %d chain(s) per wave, 1 wave per SIMD: %.3f ms  %.1f ns per MFMA issued  %.1f TFLOP/s\n", NCH, best,
         best * 1e6 / (iters * 16.0 * NCH), 1024.0 * iters * 16.0 * NCH * 2048.0 / best / 1e9);
}

template <int NCH> void run_chains(float *out, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 4000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int blocks = 256 * wps;
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(spin_chains<NCH>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    const double flop = (double)blocks * 4 * iters * 16.0 * NCH * 4096.0;
    printf("%d chain(s) per wave, %d wave(s) per SIMD: %.3f ms  %.1f TFLOP/s\n", NCH, wps, best, flop / best / 1e9);
  }
}

int main() {
  float *out;
  CK(hipMalloc(&out, 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 4000;
  for (int wps = 1; wps <= 2; wps++) {           // waves per SIMD: blocks of 4 waves, wps blocks per CU
    const int blocks = 256 * wps;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double flop = (double)blocks * 4 /*waves*/ * iters * 64.0 /*mfma*/ * 4096.0;
      printf("%d wave(s) per SIMD: %.3f ms  %.1f TFLOP/s (fp32 MFMA 32x32x2, %d blocks)\n", wps, ms, flop / ms / 1e9, blocks);
    }
  }
  run_chains16<1>(out, e0, e1);
  run_chains16<2>(out, e0, e1);
  run_chains16<4>(out, e0, e1);
  run_chains<1>(out, e0, e1);
  run_chains<2>(out, e0, e1);
  run_chains<4>(out, e0, e1);
  return 0;
}
