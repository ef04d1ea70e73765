// tools/valu_chain.hip -- premise check for a VALU form of the latency-bound chain GEMM (the conv weight update: few
// outputs, each a serial fmaf chain over N*OH*OW products).  A dependent v_mfma_f32_16x16x4_f32 retires 4 k per 32 cycles
// = 8 cycles per k (tools/chain_stream.hip: 13.5 ns per step).  What does a dependent VALU fma chain cost per k?
//   F    v_fmac_f32 acc, a[k], b[k]            operands in registers
//   P    v_pk_fma_f32                          two chains per lane
//   S    v_fmac_f32 acc, s[k], b[k]            a from an SGPR (wave-uniform row of A)
//   D    v_fmac_f32_dpp row_newbcast:n         a broadcast from lane n of each 16-lane row (4 rows x 16 columns per wave)
//   DL   D + the b operands re-read from LDS (4 ds_read_b128 per 16 k) + a (1 ds_read_b32 per 16 k)
// ns per k from the wall time of 64-lane workgroups, one per CU-SIMD slot used.
//   hipcc -O3 --offload-arch=gfx950 -o tools/valu_chain tools/valu_chain.hip && tools/valu_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void chain(float *out, const float *src, int iters, int zero) {
  __shared__ __attribute__((aligned(16))) float T[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) T[i] = 1.0f + (float)(i & 7) * 0.0001f;
  __syncthreads();
  const int lane = threadIdx.x;
  float a[16], b[16];
  for (int k = 0; k < 16; k++) { a[k] = T[lane + k * 64]; b[k] = T[1024 + lane + k * 64]; }
  float acc = 0.f;
  f32x2 acc2 = {0.f, 0.f};
  int base = zero;
  for (int it = 0; it < iters; it++) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 16; k++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a[k]), "v"(b[k]));
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 16; k += 2) {
        f32x2 av = {a[k], a[k + 1]}, bv = {b[k], b[k + 1]};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2) : "v"(av), "v"(bv));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2) : "v"(bv), "v"(av));
      }
    } else if constexpr (MODE == 2) {
      float s[16];
#pragma unroll
      for (int k = 0; k < 16; k++) s[k] = __builtin_amdgcn_readfirstlane(a[k]);
#pragma unroll
      for (int k = 0; k < 16; k++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "s"(s[k]), "v"(b[k]));
    } else if constexpr (MODE == 3 || MODE == 4) {
      if constexpr (MODE == 4) {   // next 16 k of b (64 B per lane) and of a (one word per lane) from LDS
        const f32x4 *p = reinterpret_cast<const f32x4 *>(&T[base + (lane & 15) * 16 + ((it & 7) << 8)]);
#pragma unroll
        for (int h = 0; h < 4; h++) {
          const f32x4 v = p[h];
          b[4 * h] = v[0]; b[4 * h + 1] = v[1]; b[4 * h + 2] = v[2]; b[4 * h + 3] = v[3];
        }
        a[0] = T[base + 2048 + lane + ((it & 7) << 6)];
      }
#define STEP(n) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a[0]), "v"(b[n]));
      STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
      STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
#undef STEP
    }
  }
  if (acc + acc2[0] + acc2[1] == 12345.678f) out[threadIdx.x] = acc;
}

template <int MODE> void run(const char *what, float *out, hipEvent_t e0, hipEvent_t e1, int wgs) {
  const int iters = 40000;
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(chain<MODE>, dim3(wgs), dim3(64), 0, 0, out, (const float *)out, iters, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  printf("%-64s %4d waves: %.3f ms  %.2f ns per k\n", what, wgs, best, best * 1e6 / (iters * 16.0));
}

int main() {
  float *out;
  CK(hipMalloc(&out, 1 << 20));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int wgs : {256, 576, 1024, 2048}) {
    run<0>("F  v_fmac_f32, operands in registers", out, e0, e1, wgs);
    run<1>("P  v_pk_fma_f32 (two chains per lane)", out, e0, e1, wgs);
    run<2>("S  v_fmac_f32 with an SGPR operand", out, e0, e1, wgs);
    run<3>("D  v_fmac_f32_dpp row_newbcast", out, e0, e1, wgs);
    run<4>("DL D + 4 ds_read_b128 + 1 ds_read_b32 per 16 k", out, e0, e1, wgs);
  }
  return 0;
}
