// tools/mfma_order.hip -- is v_mfma_f32_16x16x4_f32 a k-ascending fmaf chain, bit for bit?  (The 32x32x2 form is:
// tests/test_gpu_gemm.py pins the GEMM built on it against the reference's FMA build.)  A chain over K through
// 16x16x4 MFMAs is compared with the host's fmaf chain in k order, and with the other plausible orders.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/mfma_order tools/mfma_order.hip && tools/mfma_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// A: 16 x K row-major, B: K x 16 row-major, C: 16 x 16.  One wave.
__global__ void k16(const float *A, const float *B, float *C, int K) {
  const int l = threadIdx.x, i = l & 15, kq = l >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 4) {
    const float a = A[i * K + k0 + kq];
    const float b = B[(k0 + kq) * 16 + i];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; r++) C[(4 * kq + r) * 16 + i] = acc[r];
}

int main() {
  const int K = 256;
  std::vector<float> A(16 * K), B(K * 16), C(256), ref(256), ref_pair(256);
  uint32_t s = 12345u;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return (float)((int)(s % 2001u) - 1000) * 0.001f * (1.0f + (s >> 20) * 1e-3f); };
  for (auto &v : A) v = rnd();
  for (auto &v : B) v = rnd();
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 256 * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost));
  int bad_seq = 0, bad_tree = 0;
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) {
      float acc = 0.f;
      for (int k = 0; k < K; k++) acc = fmaf(A[i * K + k], B[k * 16 + j], acc);
      ref[i * 16 + j] = acc;
      // alternative: each instruction sums its four products first (unrounded pairs), then adds to acc
      float acc2 = 0.f;
      for (int k0 = 0; k0 < K; k0 += 4) {
        double t = 0.0;
        for (int k = k0; k < k0 + 4; k++) t += (double)A[i * K + k] * (double)B[k * 16 + j];
        acc2 = (float)((double)acc2 + t);
      }
      ref_pair[i * 16 + j] = acc2;
      if (memcmp(&acc, &C[i * 16 + j], 4)) bad_seq++;
      if (memcmp(&acc2, &C[i * 16 + j], 4)) bad_tree++;
    }
  printf("v_mfma_f32_16x16x4_f32, K = %d: %d / 256 outputs differ from the k-ascending fmaf chain; %d / 256 from a fused-4 sum\n", K, bad_seq, bad_tree);
  printf("%s\n", bad_seq == 0 ? "16x16x4 IS a k-ordered fmaf chain" : "16x16x4 is NOT a k-ordered fmaf chain");
  return 0;
}
