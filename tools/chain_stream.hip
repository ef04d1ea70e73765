// tools/chain_stream.hip -- what a chain wave's instruction stream costs per dependent 16x16x4 step: one wave per CU-SIMD
// runs ONE dependent MFMA chain and, per step, (V) nothing else, (L) two LDS fragment reads consumed 8 steps later,
// (LA) the same plus two VALU address ops, (L2) ONE ds_read2 fetching both fragments.  Cycles per step from the wall time.
//   hipcc -O3 --offload-arch=gfx950 -o tools/chain_stream tools/chain_stream.hip && tools/chain_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void chain(float *out, int iters, int zero) {
  __shared__ float T[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) T[i] = 1.0f + (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x;
  float fa[2][8], fb[2][8];
  for (int s = 0; s < 8; s++) { fa[0][s] = T[lane + s * 64]; fb[0][s] = T[4096 + lane + s * 64]; }
  int base = zero;  // 0 at run time: keeps the addresses in registers
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = q & 1;
#pragma unroll
      for (int s = 0; s < 8; s++) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c][s], fb[c][s], acc, 0, 0, 0);
        if constexpr (MODE == 1) {
          fa[c ^ 1][s] = T[base + lane + ((q * 8 + s) & 31) * 64];
          fb[c ^ 1][s] = T[base + 4096 + lane + ((q * 8 + s) & 31) * 64];
        } else if constexpr (MODE == 2) {
          const int o = (base | (it & 1)) + lane;   // a VALU op per read
          fa[c ^ 1][s] = T[o + ((q * 8 + s) & 31) * 64];
          const int o2 = (base | ((it >> 1) & 1)) + lane;
          fb[c ^ 1][s] = T[4096 + o2 + ((q * 8 + s) & 31) * 64];
        } else if constexpr (MODE == 4) {   // reads as in (L), plus ONE VALU write to a source register of the MFMA just issued
          asm volatile("v_or_b32 %0, %0, %1" : "+v"(fa[c][s]) : "v"(zero));
          fa[c ^ 1][s] = T[base + lane + ((q * 8 + s) & 31) * 64];
          fb[c ^ 1][s] = T[base + 4096 + lane + ((q * 8 + s) & 31) * 64];
        } else if constexpr (MODE == 5) {   // ... to a register the MFMA does not read
          asm volatile("v_or_b32 %0, %0, %1" : "+v"(base) : "v"(zero));
          fa[c ^ 1][s] = T[base + lane + ((q * 8 + s) & 31) * 64];
          fb[c ^ 1][s] = T[base + 4096 + lane + ((q * 8 + s) & 31) * 64];
        } else if constexpr (MODE == 3) {   // both fragments in one ds_read2st64 (A and B images 4096 words apart)
          const float *p = &T[base + lane + ((q * 8 + s) & 31) * 64];
          fa[c ^ 1][s] = p[0];
          fb[c ^ 1][s] = p[4096];
        } else {
          fa[c ^ 1][s] = fa[c][s];
          fb[c ^ 1][s] = fb[c][s];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
}

// The chain wave of the wave-specialised GEMM: the same stream plus ONE workgroup barrier per 32-step tile (after the third
// quarter), with EXTRA waves in the workgroup that do nothing but meet that barrier (the loader waves, idle).
template <int EXTRA>
__global__ __launch_bounds__(64 * (1 + EXTRA)) void chain_barrier(float *out, int iters, int zero) {
  __shared__ float T[8192];
  for (int i = threadIdx.x; i < 8192; i += 64 * (1 + EXTRA)) T[i] = 1.0f + (float)(i & 7) * 0.125f;
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) >= 64) {
    for (int it = 0; it < iters; it++) __syncthreads();
    return;
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x;
  float fa[2][8], fb[2][8];
  for (int s = 0; s < 8; s++) { fa[0][s] = T[lane + s * 64]; fb[0][s] = T[4096 + lane + s * 64]; }
  int base = zero;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = q & 1;
#pragma unroll
      for (int s = 0; s < 8; s++) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c][s], fb[c][s], acc, 0, 0, 0);
        fa[c ^ 1][s] = T[base + lane + ((q * 8 + s) & 31) * 64];
        fb[c ^ 1][s] = T[base + 4096 + lane + ((q * 8 + s) & 31) * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
      if (q == 2) {
#pragma unroll
        for (int s = 0; s < 8; s++) asm volatile("" : "+v"(fa[c ^ 1][s]), "+v"(fb[c ^ 1][s]));
        __syncthreads();
      }
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
}

// The chain wave (wave 0) and ONE busy partner wave: wave PARTNER of eight issues NV VALU instructions per 32-step tile and
// meets the barrier; the other waves leave at once.  PARTNER = 4 shares the chain's SIMD (waves go to the SIMDs in turn),
// PARTNER = 1 sits on the next one.
template <int PARTNER, int NV>
__global__ __launch_bounds__(512) void chain_partner(float *out, int iters, int zero) {
  __shared__ float T[8192];
  for (int i = threadIdx.x; i < 8192; i += 512) T[i] = 1.0f + (float)(i & 7) * 0.125f;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6;
  if (wave != 0 && wave != PARTNER) return;
  if (wave == PARTNER) {
    int x = threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int v = 0; v < NV; v++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(zero));
      __syncthreads();
    }
    if (x == 12345678) out[0] = 1.f;
    return;
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x;
  float fa[2][8], fb[2][8];
  for (int s = 0; s < 8; s++) { fa[0][s] = T[lane + s * 64]; fb[0][s] = T[4096 + lane + s * 64]; }
  int base = zero;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = q & 1;
#pragma unroll
      for (int s = 0; s < 8; s++) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c][s], fb[c][s], acc, 0, 0, 0);
        fa[c ^ 1][s] = T[base + lane + ((q * 8 + s) & 31) * 64];
        fb[c ^ 1][s] = T[base + 4096 + lane + ((q * 8 + s) & 31) * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
      if (q == 2) {
#pragma unroll
        for (int s = 0; s < 8; s++) asm volatile("" : "+v"(fa[c ^ 1][s]), "+v"(fb[c ^ 1][s]));
        __syncthreads();
      }
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
}

template <int PARTNER, int NV> void run_partner(float *out, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 20000;
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((chain_partner<PARTNER, NV>), dim3(144), dim3(512), 0, 0, out, iters, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  printf("chain + partner wave %d issuing %3d VALU instructions per 32 steps: %.3f ms  %.1f ns per step\n", PARTNER, NV, best,
         best * 1e6 / (iters * 32.0));
}

template <int EXTRA> void run_barrier(float *out, hipEvent_t e0, hipEvent_t e1, int wgs) {
  const int iters = 20000;
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(chain_barrier<EXTRA>, dim3(wgs), dim3(64 * (1 + EXTRA)), 0, 0, out, iters, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  printf("2 ds_read_b32 per step + a barrier per 32 steps, %d idle waves %4d workgroups: %.3f ms  %.1f ns per step\n", EXTRA, wgs,
         best, best * 1e6 / (iters * 32.0));
}

template <int MODE> void run(const char *what, float *out, hipEvent_t e0, hipEvent_t e1, int wgs) {
  const int iters = 20000;
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(chain<MODE>, dim3(wgs), dim3(64), 0, 0, out, iters, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  printf("%-46s %4d workgroups: %.3f ms  %.1f ns per step\n", what, wgs, best, best * 1e6 / (iters * 32.0));
}

int main() {
  float *out;
  CK(hipMalloc(&out, 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  run_partner<4, 0>(out, e0, e1);
  run_partner<4, 8>(out, e0, e1);
  run_partner<4, 32>(out, e0, e1);
  run_partner<4, 128>(out, e0, e1);
  run_partner<1, 32>(out, e0, e1);
  run_partner<1, 128>(out, e0, e1);
  run_barrier<0>(out, e0, e1, 144);
  run_barrier<1>(out, e0, e1, 144);
  run_barrier<4>(out, e0, e1, 144);
  run_barrier<8>(out, e0, e1, 144);
  for (int wgs : {144}) {
    run<0>("MFMA chain only", out, e0, e1, wgs);
    run<1>("+ 2 ds_read_b32 per step", out, e0, e1, wgs);
    run<2>("+ 2 ds_read_b32 + 2 VALU address ops per step", out, e0, e1, wgs);
    run<3>("+ 1 ds_read2st64_b32 per step", out, e0, e1, wgs);
    run<4>("+ 2 reads + a VALU write to the MFMA's source", out, e0, e1, wgs);
    run<5>("+ 2 reads + a VALU write to another register", out, e0, e1, wgs);
  }
  return 0;
}
