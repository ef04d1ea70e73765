// tools/ceilings.hip -- bare HBM streams for bench.py's `extras.ceilings` (measurement infrastructure, not
// part of the product): what the part gives a kernel that does nothing but move the bytes, timed in the
// same process and on the same buffers as the cast.  Geometry = the cast's: one 16-byte vector per lane,
// no loop, the grid is the buffer.
//   read  : nt loads, xor-folded so they stay alive, no stores
//   write : "sc1 nt" stores of a lane-dependent pattern, no loads
//   copy  : nt load -> sc1 nt store, 128- or 256-thread workgroups (the cast minus its arithmetic)
// Built by __graft_entry__.build() into tools/libfpemu_ceilings.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_stream(u32x4 *p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

__global__ __launch_bounds__(256) void k_read(const u32x4 *in, u32x4 *sink, uint64_t nvec) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= nvec) return;
  const u32x4 v = __builtin_nontemporal_load(&in[i]);
  if ((v[0] ^ v[1] ^ v[2] ^ v[3]) == 0x12345678u) sink[i & 63u] = v;  // practically never: keeps the load alive
}

__global__ __launch_bounds__(256) void k_write(u32x4 *out, uint64_t nvec) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= nvec) return;
  const u32x4 o = {(uint32_t)i, 1u, 2u, 3u};
  store_stream(out + i, o);
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_copy(const u32x4 *in, u32x4 *out, uint64_t nvec) {
  const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= nvec) return;
  store_stream(out + i, __builtin_nontemporal_load(&in[i]));
}

// Compute ceilings for the c-model GEMM legs, timed in the bench process like the streams above:
//   mfma : four independent v_mfma_f32_32x32x2_f32 accumulator chains per wave (the GEMM kernel's shape), no memory
//          traffic -- what the fp32 MFMA pipe sustains at the clock this process gets;
//   valu : sixteen independent packed chains per lane of a rounded product then a rounded sum (v_pk_mul_f32 +
//          v_pk_add_f32, built with -ffp-contract=off) -- the separate-rounding GEMM's instruction mix, nothing else.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_mfma(float *out, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int i = 0; i < 4; i++)
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  const float a = a0 + threadIdx.x, b = b0;
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

__global__ __launch_bounds__(256) void k_valu(float *out, int iters, float a0) {
  f32x2 acc[16];
  for (int i = 0; i < 16; i++) acc[i] = f32x2{1.0f + (float)i, 0.5f + (float)threadIdx.x};
  const f32x2 a = {a0, -a0};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const f32x2 p = a * acc[(i + 8) & 15];  // a product of its own (contraction is off), never loop-invariant
        acc[i] = acc[i] + p;
      }
    }
  }
  f32x2 s = {0.f, 0.f};
  for (int i = 0; i < 16; i++) s = s + acc[i];
  if (s[0] + s[1] == 12345.678f) out[threadIdx.x] = s[0];
}

static int done(void) { return (int)hipGetLastError(); }

extern "C" {
// flop of one launch: blocks * 4 waves * iters * 64 MFMAs * 4096
int ceil_mfma(void *out, int iters, int blocks, void *stream) {
  if (out == nullptr || iters <= 0 || blocks <= 0) return -1;
  hipLaunchKernelGGL(k_mfma, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float *)out, iters, 1.0f, 0.5f);
  return done();
}
// flop of one launch: blocks * 256 lanes * iters * 4 * 16 chains * 2 elements * 2
int ceil_valu(void *out, int iters, int blocks, void *stream) {
  if (out == nullptr || iters <= 0 || blocks <= 0) return -1;
  hipLaunchKernelGGL(k_valu, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float *)out, iters, 1e-9f);
  return done();
}
int ceil_read(const void *in, void *sink, uint64_t nvec, void *stream) {
  if (nvec == 0 || nvec > (1ull << 38)) return -1;
  hipLaunchKernelGGL(k_read, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4 *)in, (u32x4 *)sink, nvec);
  return done();
}
int ceil_write(void *out, uint64_t nvec, void *stream) {
  if (nvec == 0 || nvec > (1ull << 38)) return -1;
  hipLaunchKernelGGL(k_write, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (u32x4 *)out,
                     nvec);
  return done();
}
int ceil_copy(const void *in, void *out, uint64_t nvec, int block, void *stream) {
  if (nvec == 0 || nvec > (1ull << 38)) return -1;
  if (block == 128)
    hipLaunchKernelGGL(k_copy<128>, dim3((unsigned)((nvec + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
                       (const u32x4 *)in, (u32x4 *)out, nvec);
  else
    hipLaunchKernelGGL(k_copy<256>, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const u32x4 *)in, (u32x4 *)out, nvec);
  return done();
}
}
