// tools/mfma_zero.hip -- signed zeros through the fp32 MFMAs: what do v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x2_f32
// return for acc + a*b when IEEE fmaf returns -0 (or a zero by cancellation / underflow)?  One wave, every lane the same
// operands, K = 4 products per case; prints the result bits beside the host's fmaf chain.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/mfma_zero tools/mfma_zero.hip && tools/mfma_zero
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Case { float a, b, c; };

__global__ void probe(const Case *cs, int n, uint32_t *out) {
  for (int i = 0; i < n; i++) {
    const Case c = cs[i];
    f32x4 acc4 = {c.c, c.c, c.c, c.c};
    acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.a, c.b, acc4, 0, 0, 0);   // 4 products a*b per output
    f32x16 acc16;
    for (int r = 0; r < 16; r++) acc16[r] = c.c;
    acc16 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.a, c.b, acc16, 0, 0, 0);  // 2 products
    acc16 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.a, c.b, acc16, 0, 0, 0);  // 2 more
    if (threadIdx.x == 0) {
      out[2 * i] = __float_as_uint(acc4[0]);
      out[2 * i + 1] = __float_as_uint(acc16[0]);
    }
  }
}

int main() {
  const float nz = -0.0f, tiny = 1e-30f;
  const Case cases[] = {
      {-1.f, 0.f, nz}, {1.f, nz, nz}, {nz, 0.f, nz}, {1.f, 0.f, nz}, {-1.f, nz, nz}, {nz, nz, nz},   // c = -0, zero products
      {-1.f, 0.f, 0.f}, {1.f, 0.f, 0.f},                                                             // c = +0
      {-1.f, 1.f, 4.f}, {1.f, 1.f, -4.f},                                                            // cancellation to zero
      {-tiny, tiny, nz}, {-tiny, tiny, 0.f}, {tiny, tiny, nz},                                       // products underflow
      {-1.f, 0.f, 1.f}, {-1.f, 0.f, -1.f},                                                           // nonzero acc
      {-1e-20f, 1e-20f, nz}, {-1e-20f, 1e-20f, 0.f},                                                 // denormal results
  };
  const int n = sizeof(cases) / sizeof(cases[0]);
  Case *d;
  uint32_t *o, h[2 * n];
  CK(hipMalloc(&d, sizeof(cases))); CK(hipMalloc(&o, sizeof(h)));
  CK(hipMemcpy(d, cases, sizeof(cases), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, n, o);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost));
  printf("%14s %14s %14s | %10s %10s %10s\n", "a", "b", "c", "fmaf x4", "16x16x4", "32x32x2 x2");
  for (int i = 0; i < n; i++) {
    float acc = cases[i].c;
    for (int k = 0; k < 4; k++) acc = fmaf(cases[i].a, cases[i].b, acc);
    uint32_t w;
    memcpy(&w, &acc, 4);
    printf("%14g %14g %14g | %08x   %08x%s   %08x%s\n", cases[i].a, cases[i].b, cases[i].c, w, h[2 * i],
           h[2 * i] == w ? "  " : " !", h[2 * i + 1], h[2 * i + 1] == w ? "  " : " !");
  }
  return 0;
}
