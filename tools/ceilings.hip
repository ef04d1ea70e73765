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

static int done(void) { return (int)hipGetLastError(); }

extern "C" {
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
