// fpemu_hip.hip -- streaming cast + amax kernels for MI355X (gfx950) and the C ABI of
// include/fpemu_hip.h.  Written for CDNA4 only: 64-lane wavefronts, 16-byte (dwordx4)
// non-temporal global accesses, no LDS in the cast (nothing is reused), one launch per call,
// no host synchronisation, no allocation.
//
// Replaces, behind the same operator boundary, the reference's
//   fpemu_cuda_forward + *_Kernel   mpemu/pytquant/hip/fpemu_kernels.hip:254-1392
//   fpemu_common_function           mpemu/pytquant/cpp/fpemu_impl.cpp:1866-2009
// (one element per thread, 4-byte accesses, hipDeviceSynchronize per call, racy global RNG).
//
// Memory layout: buffers are flat and contiguous.  A "superblock" = U slots x 64 lanes x 16 B; lane L touches
// the U 16-byte vectors {slot*64 + L}, so every load/store instruction of a wave is one fully coalesced 1 KiB
// segment.  Deterministic modes: U = 1, one superblock per wave, no loop -- the grid IS the tensor (measured
// fastest on MI355X, tools/sweep.hip) -- in 128-thread workgroups; the data load is the first thing a wave
// does (kernarg preload puts its address in SGPRs at launch; a device-side scale is resolved after the load
// is in flight).  Stochastic modes: the noise layout's superblock has U = 4 (one Philox4x32 call per lane =
// 16 random bytes = its 16 elements); with fp32 storage a 256-thread workgroup owns one superblock, one slot
// per wave, and shares the Philox draw through LDS (cast_blocks_sr); with 16-bit storage a wave keeps all four
// slots.  Loads are non-temporal (read once); stores carry "sc1 nt" (streaming write-through): +2 % over nt
// stores, measured.  E5M2 / E4M3 / E3M4 under RTZ / RNE / STOCHASTIC quantise two elements per instruction
// (fpemu_device.hpp, quant_h2).  Elements past the last whole superblock (and whole buffers whose pointers are
// not 16-byte aligned) go through a 4-byte-per-lane path in the same launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <string.h>

#include <atomic>

#include "fpemu_common.hpp"

using namespace fpemu;

// fpemu_rows.hip: one launch of the per-row (per-channel / per-block) kernel
extern "C" int fpemu_rows_launch_(const void *in, void *out, uint64_t rows, uint64_t cols, int dtype,
                                  int fmt, int rmode, float flt_max, int policy, uint64_t seed,
                                  uint64_t offset, uint32_t *amax_out, int force_block, uint64_t col_base,
                                  void *stream);
// fpemu_rows.hip: row maxima only
extern "C" int fpemu_rows_amax_launch_(const void *in, uint64_t rows, uint64_t cols, int dtype, uint32_t *amax_bits,
                                       void *stream);

namespace {

constexpr int kDefaultMaxBlocks = 0x7FFFFFFF;  // cast: one superblock per wave, no loop
constexpr int kAmaxMaxBlocks = 256 * 16;       // amax: looped, one atomic per block
constexpr uint64_t kSmallVectors = 2048;       // <= this many 16-byte vectors: fused one-workgroup path (the
                                               // workgroup holds them in registers: fpemu_rows.hip, rows_reg_kernel)
constexpr uint64_t kKeepBytes = 160ull << 20;  // fused amax -> cast: tensors up to this size keep cache residency
constexpr uint64_t kPartialsBytes = 768ull << 20;  // above: zero + atomic amax + cast is as fast or faster (1 GiB: 0.489 vs 0.503 ms)
constexpr uint64_t kMaxPartials = 256;  // one per CU; measured best of 128 / 256 / 512 / 1024 / 2048
constexpr int kNoRowsKernel = -100;            // fpemu_rows_launch_: no kernel for this fmt/rmode/dtype

std::atomic<int> g_max_blocks{0};  // test hook only, see fpemu_test_set_max_blocks_
thread_local char g_err[256] = "";

int fail(int code, const char *msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

// Everything the cast kernel needs besides the four leading scalars.  The kernel takes the first
// fields as plain scalar arguments (see cast_kernel): together with in / out / nsb / n they are the 14
// dwords that gfx950 preloads into SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count, Makefile),
// so an unscaled cast issues its global load without waiting for a single scalar load.
struct CastParams {
  float scale;                // host-provided scale (ignored when amax_bits != nullptr)
  float rcp;                  // (float)(1.0/(double)scale), computed on the host
  const uint32_t *amax_bits;  // device word; non-null: derive scale on device
  uint32_t npart;             // > 0: `rest.partials` holds per-block maxima of a preceding amax_kernel<.., PARTIAL>
  int clamp;
  struct Rest {               // not preloaded: fetched by scalar loads where a path needs them
    float flt_max;
    uint64_t seed, offset;
    uint64_t elem_base;       // global index of element 0 of this buffer
    const uint32_t *partials; // npart > 0: every block reduces them itself (no atomics, no pre-zeroed word)
    uint32_t *amax_pub;       //            and block 0 publishes the maximum here
  } rest;
};

// Workgroup size of the flat cast kernels (the amax and rows kernels keep kBlock).  Measured on MI355X at
// 1 GiB (tools/sweep.hip against libraries built with each size): the one-vector-per-lane deterministic cast
// runs 6.74 / 6.84 / 6.79 TB/s with 256 / 128 / 64 threads -- smaller workgroups retire and are replaced at a
// finer grain, which keeps more requests in flight; the stochastic cast (four vectors per lane) prefers 256.
// 16-bit storage, deterministic: 128 as well since the packed quantiser (tools/storage_bench.py against libraries
// built with 64 / 128 / 256: bf16 at 1 GiB of traffic 6.72 / 6.84 / 6.71 TB/s with one vector per lane).
#ifndef FPEMU_CAST_BLOCK_DET
#define FPEMU_CAST_BLOCK_DET 128
#endif
#ifndef FPEMU_CAST_BLOCK_SR
#define FPEMU_CAST_BLOCK_SR 256
#endif
#ifndef FPEMU_CAST_BLOCK_16
#define FPEMU_CAST_BLOCK_16 128
#endif
#ifndef FPEMU_CAST_BLOCK_SR16
#define FPEMU_CAST_BLOCK_SR16 256   // stochastic, 16-bit storage: 6.31 / 6.26 / 6.19 TB/s with 256 / 128 / 64 (128 Mi bf16)
#endif
constexpr int cast_block(int rm, int dt) {
  return rm == STOCHASTIC ? (dt == FPEMU_F32 ? FPEMU_CAST_BLOCK_SR : FPEMU_CAST_BLOCK_SR16)
                          : (dt == FPEMU_F32 ? FPEMU_CAST_BLOCK_DET : FPEMU_CAST_BLOCK_16);
}

// Random bits for element e (0..V-1) of slot s from the lane's Philox results of this superblock.
template <int FMT, unsigned V, int NCALLS>
__device__ __forceinline__ uint32_t lane_noise(const Philox4 (&p)[NCALLS], unsigned slot, unsigned e) {
  constexpr unsigned NB = noise_bytes<FMT>();
  const unsigned pos = (slot * V + e) * NB;   // compile-time after unrolling
  const uint32_t w = philox_word(p[pos / 16u], (pos % 16u) / 4u) >> ((pos % 4u) * 8u);
  return w;  // consumers mask the bits they use
}

// The noise bytes of elements e and e + 1 (e even) of slot s, one per 16-bit half: the packed quantisers'
// operand.  Both bytes lie in one Philox word; v_perm_b32 spreads them (selector 0x0C = constant zero byte).
template <int FMT, unsigned V, int NCALLS>
__device__ __forceinline__ uint32_t lane_noise_pair(const Philox4 (&p)[NCALLS], unsigned slot, unsigned e) {
  static_assert(noise_bytes<FMT>() == 1, "packed noise is one byte per element");
  const unsigned pos = slot * V + e;          // compile-time after unrolling; even
  const uint32_t w = philox_word(p[pos / 16u], (pos % 16u) / 4u);
  return __builtin_amdgcn_perm(0u, w, (pos % 4u) == 0u ? 0x0C010C00u : 0x0C030C02u);
}

template <int FMT, unsigned V, int NCALLS>
struct LaneNoise {  // cast_vector's noise source in the flat cast: slot `slot` of the lane's Philox results
  const Philox4 (&p)[NCALLS];
  unsigned slot;
  __device__ __forceinline__ uint32_t one(unsigned e) const { return lane_noise<FMT, V, NCALLS>(p, slot, e); }
  __device__ __forceinline__ uint32_t pair(unsigned e) const {
    if constexpr (noise_bytes<FMT>() == 1) return lane_noise_pair<FMT, V, NCALLS>(p, slot, e);
    else return 0u;  // two-byte noise: formats without a packed quantiser never ask
  }
};

// 16-byte vectors per lane: 4 for stochastic modes (the noise layout's superblock); deterministic casts 1 (pure
// stream, measured fastest).  16-bit storage held two vectors per lane while its quantiser ran per element; with the
// packed quantiser the ALU no longer shows (E5M2_RTZ and E4M3_RNE run at the same rate) and the fp32 geometry wins:
// bf16, 1 GiB of traffic per launch, 6.43 (2 vectors, 256 threads) -> 6.84 TB/s (1 vector, 128 threads), the fp32
// cast's rate over the same bytes; 4 vectors 6.42, 2 vectors with 128 / 64 threads 6.45 / 6.58.
#ifndef FPEMU_CAST_U16
#define FPEMU_CAST_U16 1
#endif
constexpr int slots_per_wave(int rm, int dt) { return rm == STOCHASTIC ? 4 : (dt == FPEMU_F32 ? 1 : FPEMU_CAST_U16); }

// Scale sources of the streaming body.  resolve() runs once per thread AFTER the first superblock's
// loads have been issued, so whatever it waits for (a device word, the per-block maxima, a barrier)
// overlaps the HBM latency of the data instead of preceding it.
struct GivenScale {
  float scale, rcp;
  __device__ __forceinline__ void resolve(float &s, float &r) const { s = scale; r = rcp; }
};

template <int kCastBlock>
struct DeviceScale {  // cast_kernel: scale given, or derived from a device word / per-block maxima
  static constexpr int kCastWaves = kCastBlock / 64;
  float scale, rcp;
  const uint32_t *amax_bits;
  uint32_t npart;
  int clamp;
  const CastParams::Rest *rest;
  __device__ __forceinline__ void resolve(float &s, float &r) const {
    s = scale;
    r = rcp;
    if (npart != 0u) {  // uniform over the launch
      __shared__ uint32_t pmax[kCastWaves];
      uint32_t m = 0u;
      for (uint32_t i = threadIdx.x; i < npart; i += kCastBlock) {
        const uint32_t v = rest->partials[i];
        m = m > v ? m : v;
      }
      m = wave_max_u32(m);
      uint32_t mx = m;
      if constexpr (kCastWaves > 1) {
        if ((threadIdx.x & 63u) == 0u) pmax[threadIdx.x >> 6] = m;
        __syncthreads();
        mx = pmax[0];
#pragma unroll
        for (int w = 1; w < kCastWaves; w++) mx = mx > pmax[w] ? mx : pmax[w];
      }
      if (blockIdx.x == 0 && threadIdx.x == 0) *rest->amax_pub = mx;
      s = scale_from_amax(mx, rest->flt_max, clamp != 0);
      r = scale_reciprocal(s);
    } else if (amax_bits != nullptr) {
      // clamp == 2 (internal): the word is the scale itself, not an amax (fpemu_cast_device_scale)
      s = clamp == 2 ? u2f(*amax_bits) : scale_from_amax(*amax_bits, rest->flt_max, clamp != 0);
      r = scale_reciprocal(s);
    }
  }
};

// The streaming body: workgroup `blk` of `nblk` that share one buffer (cast_kernel: the whole grid;
// cast_multi_kernel: the tensor's share of the grid).
template <int FMT, int RM, int DT, int kCastBlock, class ScaleSrc>
__device__ __forceinline__ void cast_blocks(const void *in_, void *out_, uint64_t nsb, uint64_t n,
                                            const ScaleSrc &src, uint64_t seed, uint64_t offset, uint64_t elem_base,
                                            unsigned blk, unsigned nblk) {
  constexpr unsigned V = Storage<DT>::V;
  constexpr int U = slots_per_wave(RM, DT);
  constexpr uint64_t SB_ELEMS = (uint64_t)V * 64u * U;
  constexpr bool SR = (RM == STOCHASTIC);
  constexpr int NCALLS = SR ? (int)(4u * V * noise_bytes<FMT>() / 16u) : 1;
  constexpr int kCastWaves = kCastBlock / 64;

  const unsigned lane = threadIdx.x & 63u;
  const u32x4 *in = reinterpret_cast<const u32x4 *>(in_);
  u32x4 *out = reinterpret_cast<u32x4 *>(out_);

  // first (normally the only) superblock of this wave: issue its loads before anything else
  uint64_t sb = (uint64_t)blk * kCastWaves + (threadIdx.x >> 6);
  bool have = sb < nsb;
  u32x4 v[U];
  if (have) {
    const uint64_t v0 = sb * (64u * U) + lane;
#pragma unroll
    for (int s = 0; s < U; s++) v[s] = __builtin_nontemporal_load(&in[v0 + (uint64_t)s * 64u]);
  }

  float scale, rcp;
  src.resolve(scale, rcp);

  const uint64_t nwaves = (uint64_t)nblk * kCastWaves;
  const uint64_t sb_base = elem_base / SB_ELEMS;
  while (have) {
    const uint64_t v0 = sb * (64u * U) + lane;
    Philox4 p[NCALLS];
    if constexpr (SR) {
      const uint64_t c = ((sb_base + sb) * 64u + lane) << 2;
#pragma unroll
      for (int k = 0; k < NCALLS; k++) p[k] = philox4x32_10(seed, c | (uint64_t)k, offset);
    }

#pragma unroll
    for (int s = 0; s < U; s++) {
      const u32x4 o = cast_vector<FMT, RM, DT>(v[s], scale, rcp, LaneNoise<FMT, V, NCALLS>{p, (unsigned)s});
      store_stream(&out[v0 + (uint64_t)s * 64u], o);
    }

    sb += nwaves;  // only grids capped by the test hook come round again
    have = sb < nsb;
    if (have) {
      const uint64_t w0 = sb * (64u * U) + lane;
#pragma unroll
      for (int s = 0; s < U; s++) v[s] = __builtin_nontemporal_load(&in[w0 + (uint64_t)s * 64u]);
    }
  }

  // Tail (n % superblock) and unaligned buffers (launched with nsb == 0): one element per lane
  // per step, still coalesced, same element -> noise mapping.
  using elem_t = typename Storage<DT>::elem;
  const elem_t *ein = reinterpret_cast<const elem_t *>(in_);
  elem_t *eout = reinterpret_cast<elem_t *>(out_);
  const uint64_t tstride = (uint64_t)nblk * kCastBlock;
  for (uint64_t i = nsb * SB_ELEMS + (uint64_t)blk * kCastBlock + threadIdx.x; i < n; i += tstride) {
    uint32_t rnd = 0u;
    if constexpr (SR) rnd = rand_bits_elem<FMT, V>(seed, offset, elem_base + i);
    if constexpr (DT == FPEMU_F32) {
      eout[i] = cast_one<FMT, RM>(ein[i], scale, rcp, rnd);
    } else {
      eout[i] = (elem_t)store_elem16<DT>(cast_one<FMT, RM>(load_elem16<DT>(ein[i]), scale, rcp, rnd));
    }
  }
}

// Which casts take cast_blocks_sr (below): stochastic modes on fp32 storage.  With 16-bit storage the stochastic
// cast is ALU-bound either way (twice the elements and noise per byte) and measured no better by slot: fp16
// 6.2 -> 5.8 TB/s, bf16 6.3 -> 6.4.
constexpr bool sr_by_slot(int rm, int dt) { return rm == STOCHASTIC && dt == FPEMU_F32; }

// cast_vector's noise source in the stochastic flat cast: the W words of the lane's Philox results that cover
// one slot (byte range [slot * V * NB, (slot + 1) * V * NB) of the lane's draw).
template <int FMT, int W>
struct SlotNoise {
  uint32_t w[W];
  __device__ __forceinline__ uint32_t one(unsigned e) const {
    const unsigned pos = e * noise_bytes<FMT>();  // compile-time after unrolling
    return w[pos / 4u] >> ((pos % 4u) * 8u);      // consumers mask the bits they use
  }
  __device__ __forceinline__ uint32_t pair(unsigned e) const {  // e even, one noise byte per element
    return __builtin_amdgcn_perm(0u, w[e / 4u], (e % 4u) == 0u ? 0x0C010C00u : 0x0C030C02u);
  }
};

// Stochastic modes.  The noise layout ties one Philox call to the four slots of a lane (DESIGN.md "Stochastic
// rounding"), but four vectors per lane stream 3-5 % slower than one (tools/sweep.hip, U4 vs U1: a wave that
// alternates a load phase and a store phase keeps fewer requests in flight).  So a 256-thread workgroup owns
// ONE superblock and wave w streams slot w -- one vector per lane, as in the deterministic cast -- while the
// Philox results of the superblock's 64 lanes are drawn once (NCALLS waves draw one call each) and handed round
// through 1-4 KiB of LDS.  The draw and its barrier sit under the HBM latency of the loads issued before them.
template <int FMT, int RM, int DT, class ScaleSrc>
__device__ __forceinline__ void cast_blocks_sr(const void *in_, void *out_, uint64_t nsb, uint64_t n,
                                               const ScaleSrc &src, uint64_t seed, uint64_t offset,
                                               uint64_t elem_base, unsigned blk, unsigned nblk) {
  constexpr unsigned V = Storage<DT>::V;
  constexpr unsigned NB = noise_bytes<FMT>();
  constexpr uint64_t SB_ELEMS = (uint64_t)V * 64u * 4u;
  constexpr int NCALLS = (int)(4u * V * NB / 16u);  // Philox calls per lane and superblock: 1, 2 or 4
  constexpr int W = (int)(V * NB / 4u);             // words of the lane's draw per slot
  static_assert(cast_block(STOCHASTIC, DT) == 256 && sr_by_slot(RM, DT), "one wave per slot");
  __shared__ uint32_t draw[NCALLS * 4 * 64];        // [call][word][lane]

  const unsigned lane = threadIdx.x & 63u, slot = threadIdx.x >> 6;
  const u32x4 *in = reinterpret_cast<const u32x4 *>(in_);
  u32x4 *out = reinterpret_cast<u32x4 *>(out_);

  uint64_t sb = blk;  // uniform over the workgroup: the barriers below are reached by all four waves or none
  bool have = sb < nsb;
  u32x4 v;
  if (have) v = __builtin_nontemporal_load(&in[sb * 256u + slot * 64u + lane]);

  float scale, rcp;
  src.resolve(scale, rcp);

  const uint64_t sb_base = elem_base / SB_ELEMS;
  while (have) {
    if (slot < (unsigned)NCALLS) {  // wave-uniform
      const uint64_t c = (((sb_base + sb) * 64u + lane) << 2) | (uint64_t)slot;
      const Philox4 p = philox4x32_10(seed, c, offset);
      draw[(slot * 4u + 0u) * 64u + lane] = p.x;
      draw[(slot * 4u + 1u) * 64u + lane] = p.y;
      draw[(slot * 4u + 2u) * 64u + lane] = p.z;
      draw[(slot * 4u + 3u) * 64u + lane] = p.w;
    }
    __syncthreads();
    SlotNoise<FMT, W> noise;
#pragma unroll
    for (int i = 0; i < W; i++) noise.w[i] = draw[(slot * (unsigned)W + (unsigned)i) * 64u + lane];
    store_stream(&out[sb * 256u + slot * 64u + lane], cast_vector<FMT, RM, DT>(v, scale, rcp, noise));

    sb += nblk;  // only grids capped by the test hook come round again
    have = sb < nsb;
    if (have) {
      v = __builtin_nontemporal_load(&in[sb * 256u + slot * 64u + lane]);
      __syncthreads();  // everyone has read this round's draw before the next one is written
    }
  }

  using elem_t = typename Storage<DT>::elem;
  const elem_t *ein = reinterpret_cast<const elem_t *>(in_);
  elem_t *eout = reinterpret_cast<elem_t *>(out_);
  const uint64_t tstride = (uint64_t)nblk * 256u;
  for (uint64_t i = nsb * SB_ELEMS + (uint64_t)blk * 256u + threadIdx.x; i < n; i += tstride) {
    const uint32_t rnd = rand_bits_elem<FMT, V>(seed, offset, elem_base + i);
    if constexpr (DT == FPEMU_F32) {
      eout[i] = cast_one<FMT, RM>(ein[i], scale, rcp, rnd);
    } else {
      eout[i] = (elem_t)store_elem16<DT>(cast_one<FMT, RM>(load_elem16<DT>(ein[i]), scale, rcp, rnd));
    }
  }
}

// Workgroups a buffer of nsb superblocks takes on the 16-byte path.
inline uint64_t superblock_groups(uint64_t nsb, int rmode, int dtype) {
  if (sr_by_slot(rmode, dtype)) return nsb;  // one workgroup per superblock
  const uint64_t waves = (uint64_t)cast_block(rmode, dtype) / 64u;
  return (nsb + waves - 1) / waves;     // one wave per superblock
}

// The first ten arguments are 14 dwords: the part of the kernarg segment gfx950 preloads into SGPRs.
template <int FMT, int RM, int DT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void cast_kernel(const void *in_, void *out_, uint64_t nsb, uint64_t n,
                                                      float scale, float rcp, const uint32_t *amax_bits,
                                                      uint32_t npart, int clamp, CastParams::Rest rest) {
  const DeviceScale<BLOCK> src{scale, rcp, amax_bits, npart, clamp, &rest};
  if constexpr (sr_by_slot(RM, DT))
    cast_blocks_sr<FMT, RM, DT>(in_, out_, nsb, n, src, rest.seed, rest.offset, rest.elem_base, blockIdx.x, gridDim.x);
  else
    cast_blocks<FMT, RM, DT, BLOCK>(in_, out_, nsb, n, src, rest.seed, rest.offset, rest.elem_base, blockIdx.x,
                                    gridDim.x);
}

// float64 storage (FPEMU_F64): the reference's GPU dispatcher takes double tensors (hip/fpemu_kernels.hip:1070) and
// its kernels compute `float inval = in[gid] * scale` with both operands double -- ONE rounding of the double
// product to fp32 -- then the fp32 algorithm, and widen `h.f * scale_reciprocal` back (hip:429-431, 472).  The same
// here: p = (float)(x * (double)scale); cast_one(p, 1, rcp); widen.  One 16-byte vector (two elements) per lane, one
// shot; element i draws the noise of element elem_base + i of the fp32 layout, so a tensor gives the same random
// bits whichever of the two widths stores it.  Twice the bytes per element of the fp32 cast, the same stream.
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <int FMT, int RM>
__device__ __forceinline__ double cast_one_f64(double x, double dscale, float rcp, uint64_t seed, uint64_t offset,
                                               uint64_t idx) {
  uint32_t rnd = 0u;
  if constexpr (RM == STOCHASTIC) rnd = rand_bits_elem<FMT, 4>(seed, offset, idx);
  return (double)cast_one<FMT, RM>((float)(x * dscale), 1.0f, rcp, rnd);
}

// 128-thread workgroups and the sc1 nt store, as the fp32 deterministic cast (256 threads with a plain nt store: 6.60
// TB/s at 128 Mi elements against the fp32 cast's 6.85 over the same bytes).
constexpr int kCast64Block = FPEMU_CAST_BLOCK_DET;
template <int FMT, int RM>
__global__ __launch_bounds__(kCast64Block) void cast64_kernel(const double *in, double *out, uint64_t nvec, uint64_t n,
                                                        float scale, float rcp, uint64_t seed, uint64_t offset,
                                                        uint64_t elem_base) {
  const uint64_t gid = (uint64_t)blockIdx.x * kCast64Block + threadIdx.x;
  const double dscale = (double)scale;
  if (gid < nvec) {
    const f64x2 v = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(in) + gid);
    f64x2 o;
    o[0] = cast_one_f64<FMT, RM>(v[0], dscale, rcp, seed, offset, elem_base + 2u * gid);
    o[1] = cast_one_f64<FMT, RM>(v[1], dscale, rcp, seed, offset, elem_base + 2u * gid + 1u);
    store_stream(reinterpret_cast<u32x4 *>(out) + gid, __builtin_bit_cast(u32x4, o));
  }
  // the odd last element, or a whole buffer that is not 16-byte aligned (launched with nvec == 0)
  const uint64_t stride = (uint64_t)gridDim.x * kCast64Block;
  for (uint64_t e = nvec * 2u + gid; e < n; e += stride)
    out[e] = cast_one_f64<FMT, RM>(in[e], dscale, rcp, seed, offset, elem_base + e);
}

// Multi-tensor cast: one launch for up to kMultiTensors buffers of the same storage type and mode (the
// emulators re-quantise every master parameter after the optimizer step -- emulator_base.py
// update_fp16_master_params: 161 launches per ResNet-50 step, most of them a few microseconds of
// latency).  Tensor i is cast exactly as fpemu_cast(in_i, out_i, n_i, ..., seed, offset + i, 0): the
// same bits as the loop of calls it replaces.  The table travels in the kernel arguments; a workgroup
// finds its tensor by a scalar scan over the workgroup prefix sums.
constexpr int kMultiTensors = 32;
struct MultiTable {
  const void *in[kMultiTensors];
  void *out[kMultiTensors];
  uint64_t n[kMultiTensors];
  uint64_t nsb[kMultiTensors];        // whole superblocks on the 16-byte path (0: unaligned buffer)
  uint32_t blk_end[kMultiTensors];    // exclusive prefix end of the tensor's workgroups
  uint32_t count;
};

template <int FMT, int RM, int DT>
__global__ __launch_bounds__(cast_block(RM, DT)) void cast_multi_kernel(MultiTable tab, float scale, float rcp, uint64_t seed,
                                                            uint64_t offset) {
  unsigned t = 0;
  while (t + 1u < tab.count && blockIdx.x >= tab.blk_end[t]) t++;  // uniform: scalar loads and compares
  const unsigned first = t == 0u ? 0u : tab.blk_end[t - 1u];
  if constexpr (sr_by_slot(RM, DT))
    cast_blocks_sr<FMT, RM, DT>(tab.in[t], tab.out[t], tab.nsb[t], tab.n[t], GivenScale{scale, rcp}, seed, offset + t, 0,
                                blockIdx.x - first, tab.blk_end[t] - first);
  else
    cast_blocks<FMT, RM, DT, cast_block(RM, DT)>(tab.in[t], tab.out[t], tab.nsb[t], tab.n[t], GivenScale{scale, rcp},
                                                 seed, offset + t, 0, blockIdx.x - first, tab.blk_end[t] - first);
}

// ---------------------------------------------------------------------------------------------
// amax: unsigned max over |x| bit patterns; wave64 butterfly, one LDS word per wave, one
// device-scope atomic per block.
// ---------------------------------------------------------------------------------------------
constexpr int kAmaxVPT = 8;  // 16-byte vectors in flight per lane

// KEEP = false: read-once stream (non-temporal loads).  KEEP = true: default cache policy, for the fused
// amax -> cast of a tensor small enough to stay in the 256 MiB Infinity Cache: the cast's read of the
// same bytes right afterwards is then served from cache instead of HBM.
template <bool KEEP> __device__ __forceinline__ u32x4 amax_load(const u32x4 *p) {
  if constexpr (KEEP) return *p;
  else return __builtin_nontemporal_load(p);
}

// PARTIAL = true: instead of the atomic, block b stores its maximum to amax_bits[b] (the two-launch
// per-tensor path for mid-size tensors: no word to zero first, no atomics; cast_kernel reduces them).
template <int DT, bool KEEP, bool PARTIAL>
__global__ __launch_bounds__(kBlock) void amax_kernel(const void *__restrict__ in_, uint64_t nvec,
                                                      uint64_t n, uint32_t *amax_bits) {
  constexpr unsigned V = Storage<DT>::V;
  constexpr int U = kAmaxVPT;
  const u32x4 *in = reinterpret_cast<const u32x4 *>(in_);
  const unsigned lane = threadIdx.x & 63u;
  const uint64_t wave = (uint64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * kWavesPerBlock;
  const uint64_t nsb = nvec / (64u * U);
  uint32_t m = 0u;
  for (uint64_t sb = wave; sb < nsb; sb += nwaves) {  // one trip for every grid launch_amax picks
    const uint64_t v0 = sb * (64u * U) + lane;
    u32x4 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) v[k] = amax_load<KEEP>(&in[v0 + (uint64_t)k * 64u]);
#pragma unroll
    for (int k = 0; k < U; k++)
#pragma unroll
      for (int j = 0; j < 4; j++) m = absmax_word<DT>(m, v[k][j]);
  }
  // leftover vectors, then leftover elements / unaligned buffers (nvec == 0)
  const uint64_t tid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = nsb * (64u * U) + tid; i < nvec; i += stride) {
    const u32x4 v = amax_load<KEEP>(&in[i]);
#pragma unroll
    for (int j = 0; j < 4; j++) m = absmax_word<DT>(m, v[j]);
  }
  using elem_t = typename Storage<DT>::elem;
  const elem_t *ein = reinterpret_cast<const elem_t *>(in_);
  for (uint64_t e = nvec * V + tid; e < n; e += stride) {
    if constexpr (DT == FPEMU_F32) m = absmax_word<DT>(m, f2u(ein[e]));
    else m = absmax_word<DT>(m, (uint32_t)ein[e]);
  }

  m = wave_max_u32(m);
  __shared__ uint32_t part[kWavesPerBlock];
  if (lane == 0u) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t r = part[0];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; w++) r = r > part[w] ? r : part[w];
    // storage pattern -> fp32 pattern (order preserving; NaN stays above +Inf)
    if constexpr (DT == FPEMU_F16) r = f2u(h_to_f32(r));
    else if constexpr (DT == FPEMU_BF16) r = r << 16;
    if constexpr (PARTIAL) {
      amax_bits[blockIdx.x] = r;
      return;
    }
    // The running maximum only grows, so a block whose maximum does not beat the value already
    // published can skip its atomic: after the first wave of blocks almost all do, which keeps
    // tens of thousands of blocks off the single word.  (A stale read only costs a redundant atomic.)
    const uint32_t seen = __hip_atomic_load(amax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (r > seen) __hip_atomic_fetch_max(amax_bits, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}


// One-word reset ahead of an amax accumulation: a kernel launch costs the host ~3 us where
// hipMemsetAsync costs >10 us on this runtime, and the small-tensor regime is host-bound.
__global__ void zero_word_kernel(uint32_t *w) { *w = 0u; }

// Multi-tensor per-tensor quantisation (fpemu_quantize_per_tensor_multi): launch 1 -- every workgroup reduces its
// share of ITS tensor to one word of `partials` (indexed by workgroup: no atomics, nothing to zero); launch 2 --
// the multi-tensor cast, whose workgroups first reduce their tensor's words to the scale (L2 hits, under the HBM
// latency of their first loads), exactly as the two-launch single-tensor path does.
struct AmaxTable {
  const void *in[kMultiTensors];
  uint64_t n[kMultiTensors];
  uint64_t nvec[kMultiTensors];       // 16-byte vectors (0: unaligned buffer)
  uint32_t blk_end[kMultiTensors];    // exclusive prefix end of the tensor's workgroups = of its words in partials
  uint32_t count;
};

// ATOMIC = false: word per WORKGROUP (partials[blockIdx.x]).  ATOMIC = true: word per TENSOR (partials[t], zeroed by the
// caller), published with the conditional atomic of amax_kernel -- fpemu_amax_multi, whose words leave the device
// (an all-reduce over the process group) before any cast reads them.
template <int DT, bool ATOMIC>
__global__ __launch_bounds__(kBlock) void amax_multi_kernel(AmaxTable tab, uint32_t *partials) {
  constexpr unsigned V = Storage<DT>::V;
  unsigned t = 0;
  while (t + 1u < tab.count && blockIdx.x >= tab.blk_end[t]) t++;  // uniform
  const unsigned first = t == 0u ? 0u : tab.blk_end[t - 1u];
  const uint64_t tid = (uint64_t)(blockIdx.x - first) * kBlock + threadIdx.x;
  const uint64_t stride = (uint64_t)(tab.blk_end[t] - first) * kBlock;
  const u32x4 *in = reinterpret_cast<const u32x4 *>(tab.in[t]);
  const uint64_t nvec = tab.nvec[t], n = tab.n[t];
  uint32_t m = 0u;
#pragma unroll 4
  for (uint64_t i = tid; i < nvec; i += stride) {
    const u32x4 v = in[i];  // default cache policy: the cast re-reads these bytes next
#pragma unroll
    for (int j = 0; j < 4; j++) m = absmax_word<DT>(m, v[j]);
  }
  using elem_t = typename Storage<DT>::elem;
  const elem_t *ein = reinterpret_cast<const elem_t *>(tab.in[t]);
  for (uint64_t e = nvec * V + tid; e < n; e += stride) {
    if constexpr (DT == FPEMU_F32) m = absmax_word<DT>(m, f2u(ein[e]));
    else m = absmax_word<DT>(m, (uint32_t)ein[e]);
  }
  m = wave_max_u32(m);
  __shared__ uint32_t part[kWavesPerBlock];
  if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t r = part[0];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; w++) r = r > part[w] ? r : part[w];
    if constexpr (DT == FPEMU_F16) r = f2u(h_to_f32(r));
    else if constexpr (DT == FPEMU_BF16) r = r << 16;
    if constexpr (ATOMIC) {
      const uint32_t seen = __hip_atomic_load(partials + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (r > seen) __hip_atomic_fetch_max(partials + t, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      partials[blockIdx.x] = r;
    }
  }
}

__global__ __launch_bounds__(kBlock) void zero_words_kernel(uint32_t *w, uint64_t count) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < count) w[i] = 0u;
}

template <int kCastBlock>
struct PartialsScale {  // scale of one tensor of a multi-tensor launch from its workgroup maxima
  static constexpr int kCastWaves = kCastBlock / 64;
  const uint32_t *partials;
  uint32_t npart;
  float flt_max;
  int clamp;
  uint32_t *pub;  // the tensor's amax word; written by the tensor's first workgroup only (nullptr elsewhere)
  __device__ __forceinline__ void resolve(float &s, float &r) const {
    __shared__ uint32_t pmax[kCastWaves];
    uint32_t m = 0u;
    for (uint32_t i = threadIdx.x; i < npart; i += kCastBlock) {
      const uint32_t v = partials[i];
      m = m > v ? m : v;
    }
    m = wave_max_u32(m);
    uint32_t mx = m;
    if constexpr (kCastWaves > 1) {
      if ((threadIdx.x & 63u) == 0u) pmax[threadIdx.x >> 6] = m;
      __syncthreads();
      mx = pmax[0];
#pragma unroll
      for (int w = 1; w < kCastWaves; w++) mx = mx > pmax[w] ? mx : pmax[w];
    }
    if (pub != nullptr && threadIdx.x == 0) *pub = mx;
    s = scale_from_amax(mx, flt_max, clamp != 0);
    r = scale_reciprocal(s);
  }
};

struct MultiScaleArgs {
  uint32_t pend[kMultiTensors];  // exclusive prefix end of the tensor's words in partials
  const uint32_t *partials;
  uint32_t *amax_words;          // one per tensor of this launch
  float flt_max;
  int clamp;
};

template <int FMT, int RM, int DT>
__global__ __launch_bounds__(cast_block(RM, DT)) void cast_multi_scaled_kernel(MultiTable tab, MultiScaleArgs ms,
                                                                               uint64_t seed, uint64_t offset) {
  unsigned t = 0;
  while (t + 1u < tab.count && blockIdx.x >= tab.blk_end[t]) t++;  // uniform
  const unsigned first = t == 0u ? 0u : tab.blk_end[t - 1u];
  const unsigned pfirst = t == 0u ? 0u : ms.pend[t - 1u];
  const PartialsScale<cast_block(RM, DT)> src{ms.partials + pfirst, ms.pend[t] - pfirst, ms.flt_max, ms.clamp,
                                              (blockIdx.x == first && ms.amax_words != nullptr) ? ms.amax_words + t : nullptr};
  if constexpr (sr_by_slot(RM, DT))
    cast_blocks_sr<FMT, RM, DT>(tab.in[t], tab.out[t], tab.nsb[t], tab.n[t], src, seed, offset + t, 0, blockIdx.x - first,
                                tab.blk_end[t] - first);
  else
    cast_blocks<FMT, RM, DT, cast_block(RM, DT)>(tab.in[t], tab.out[t], tab.nsb[t], tab.n[t], src, seed, offset + t, 0,
                                                 blockIdx.x - first, tab.blk_end[t] - first);
}

// ---------------------------------------------------------------------------------------------
// Host side: dispatch tables and launch geometry.
// ---------------------------------------------------------------------------------------------
typedef void (*cast_fn)(const void *, void *, uint64_t, uint64_t, float, float, const uint32_t *, uint32_t, int,
                        CastParams::Rest);

// wide = true: the 256-thread build of a kernel whose default workgroup is smaller (the per-block-maxima
// path: every workgroup reduces the maxima first, which smaller workgroups repeat more often per byte)
template <int DT> cast_fn pick_cast(int fmt, int rm, bool wide) {
#define FPEMU_CASE(F, R) \
  if (fmt == F && rm == R) return wide ? cast_kernel<F, R, DT, kBlock> : cast_kernel<F, R, DT, cast_block(R, DT)>;
  FPEMU_CASE(E5M2, RTZ) FPEMU_CASE(E5M2, RNE) FPEMU_CASE(E5M2, STOCHASTIC) FPEMU_CASE(E5M2, RNAZ)
  FPEMU_CASE(E5M2, RNTZ) FPEMU_CASE(E5M2, RPINF) FPEMU_CASE(E5M2, RNINF)
  FPEMU_CASE(E5M2_DAZ, RNE) FPEMU_CASE(E5M2_DAZ, STOCHASTIC) FPEMU_CASE(E5M2_DAZ, RNAZ)
  FPEMU_CASE(E5M2_DAZ, RNTZ)
  FPEMU_CASE(E4M3, RTZ) FPEMU_CASE(E4M3, RNE) FPEMU_CASE(E4M3, STOCHASTIC)
  FPEMU_CASE(E3M4, RTZ) FPEMU_CASE(E3M4, RNE) FPEMU_CASE(E3M4, STOCHASTIC)
  FPEMU_CASE(BF16, RTZ) FPEMU_CASE(BF16, RNE) FPEMU_CASE(BF16, STOCHASTIC)
  FPEMU_CASE(FP16, RNE) FPEMU_CASE(FP16, STOCHASTIC)
  FPEMU_CASE(FP16_DAZ, RNE)
  FPEMU_CASE(E4M3_IEEE, RNE) FPEMU_CASE(E4M3_IEEE, STOCHASTIC)
  FPEMU_CASE(FP4, RNE)
#undef FPEMU_CASE
  return nullptr;
}

typedef void (*cast_multi_fn)(MultiTable, float, float, uint64_t, uint64_t);

template <int DT> cast_multi_fn pick_cast_multi(int fmt, int rm) {
#define FPEMU_CASE(F, R) \
  if (fmt == F && rm == R) return cast_multi_kernel<F, R, DT>;
  FPEMU_CASE(E5M2, RTZ) FPEMU_CASE(E5M2, RNE) FPEMU_CASE(E5M2, STOCHASTIC) FPEMU_CASE(E5M2, RNAZ)
  FPEMU_CASE(E5M2, RNTZ) FPEMU_CASE(E5M2, RPINF) FPEMU_CASE(E5M2, RNINF)
  FPEMU_CASE(E5M2_DAZ, RNE) FPEMU_CASE(E5M2_DAZ, STOCHASTIC) FPEMU_CASE(E5M2_DAZ, RNAZ)
  FPEMU_CASE(E5M2_DAZ, RNTZ)
  FPEMU_CASE(E4M3, RTZ) FPEMU_CASE(E4M3, RNE) FPEMU_CASE(E4M3, STOCHASTIC)
  FPEMU_CASE(E3M4, RTZ) FPEMU_CASE(E3M4, RNE) FPEMU_CASE(E3M4, STOCHASTIC)
  FPEMU_CASE(BF16, RTZ) FPEMU_CASE(BF16, RNE) FPEMU_CASE(BF16, STOCHASTIC)
  FPEMU_CASE(FP16, RNE) FPEMU_CASE(FP16, STOCHASTIC)
  FPEMU_CASE(FP16_DAZ, RNE)
  FPEMU_CASE(E4M3_IEEE, RNE) FPEMU_CASE(E4M3_IEEE, STOCHASTIC)
  FPEMU_CASE(FP4, RNE)
#undef FPEMU_CASE
  return nullptr;
}

typedef void (*cast_multi_scaled_fn)(MultiTable, MultiScaleArgs, uint64_t, uint64_t);

template <int DT> cast_multi_scaled_fn pick_cast_multi_scaled(int fmt, int rm) {
#define FPEMU_CASE(F, R) \
  if (fmt == F && rm == R) return cast_multi_scaled_kernel<F, R, DT>;
  FPEMU_CASE(E5M2, RTZ) FPEMU_CASE(E5M2, RNE) FPEMU_CASE(E5M2, STOCHASTIC) FPEMU_CASE(E5M2, RNAZ)
  FPEMU_CASE(E5M2, RNTZ) FPEMU_CASE(E5M2, RPINF) FPEMU_CASE(E5M2, RNINF)
  FPEMU_CASE(E5M2_DAZ, RNE) FPEMU_CASE(E5M2_DAZ, STOCHASTIC) FPEMU_CASE(E5M2_DAZ, RNAZ)
  FPEMU_CASE(E5M2_DAZ, RNTZ)
  FPEMU_CASE(E4M3, RTZ) FPEMU_CASE(E4M3, RNE) FPEMU_CASE(E4M3, STOCHASTIC)
  FPEMU_CASE(E3M4, RTZ) FPEMU_CASE(E3M4, RNE) FPEMU_CASE(E3M4, STOCHASTIC)
  FPEMU_CASE(E4M3_IEEE, RNE) FPEMU_CASE(E4M3_IEEE, STOCHASTIC)
  FPEMU_CASE(FP4, RNE)
#undef FPEMU_CASE
  return nullptr;  // BF16 / FP16 take no scale
}

typedef void (*cast64_fn)(const double *, double *, uint64_t, uint64_t, float, float, uint64_t, uint64_t, uint64_t);

cast64_fn pick_cast64(int fmt, int rm) {
#define FPEMU_CASE(F, R) \
  if (fmt == F && rm == R) return cast64_kernel<F, R>;
  FPEMU_CASE(E5M2, RTZ) FPEMU_CASE(E5M2, RNE) FPEMU_CASE(E5M2, STOCHASTIC) FPEMU_CASE(E5M2, RNAZ)
  FPEMU_CASE(E5M2, RNTZ) FPEMU_CASE(E5M2, RPINF) FPEMU_CASE(E5M2, RNINF)
  FPEMU_CASE(E5M2_DAZ, RNE) FPEMU_CASE(E5M2_DAZ, STOCHASTIC) FPEMU_CASE(E5M2_DAZ, RNAZ)
  FPEMU_CASE(E5M2_DAZ, RNTZ)
  FPEMU_CASE(E4M3, RTZ) FPEMU_CASE(E4M3, RNE) FPEMU_CASE(E4M3, STOCHASTIC)
  FPEMU_CASE(E3M4, RTZ) FPEMU_CASE(E3M4, RNE) FPEMU_CASE(E3M4, STOCHASTIC)
  FPEMU_CASE(BF16, RTZ) FPEMU_CASE(BF16, RNE) FPEMU_CASE(BF16, STOCHASTIC)
  FPEMU_CASE(FP16, RNE) FPEMU_CASE(FP16, STOCHASTIC)
  FPEMU_CASE(FP16_DAZ, RNE)
  FPEMU_CASE(E4M3_IEEE, RNE) FPEMU_CASE(E4M3_IEEE, STOCHASTIC)
  FPEMU_CASE(FP4, RNE)
#undef FPEMU_CASE
  return nullptr;
}

// float64 storage: plain casts with a host-side scale only (see cast64_kernel)
int launch_cast64(const void *in, void *out, uint64_t n, int fmt, int rmode, const CastParams &prm, void *stream);

cast_fn pick_cast_dt(int dtype, int fmt, int rm, bool wide = false) {
  switch (dtype) {
    case FPEMU_F32: return pick_cast<FPEMU_F32>(fmt, rm, wide);
    case FPEMU_F16: return pick_cast<FPEMU_F16>(fmt, rm, wide);
    case FPEMU_BF16: return pick_cast<FPEMU_BF16>(fmt, rm, wide);
    default: return nullptr;
  }
}


int max_blocks() {
  const int mb = g_max_blocks.load(std::memory_order_relaxed);
  return mb > 0 ? mb : kDefaultMaxBlocks;
}

int launch_cast(const void *in, void *out, uint64_t n, int dtype, int fmt, int rmode,
                const CastParams &prm, void *stream) {
  if (n == 0) return FPEMU_OK;
  if (in == nullptr || out == nullptr) return fail(FPEMU_EINVAL, "fpemu_cast: null buffer");
  if (dtype == FPEMU_F64) return launch_cast64(in, out, n, fmt, rmode, prm, stream);
  if (dtype < FPEMU_F32 || dtype > FPEMU_BF16) return fail(FPEMU_EINVAL, "fpemu_cast: bad dtype id");
  const bool wide = prm.npart != 0u;
  cast_fn fn = pick_cast_dt(dtype, fmt, rmode, wide);
  if (fn == nullptr) return fail(FPEMU_EINVAL, "fpemu_cast: format / rounding-mode pair not defined");
  const unsigned es = elem_size(dtype);
  if (((uintptr_t)in % es) || ((uintptr_t)out % es))
    return fail(FPEMU_EALIGN, "fpemu_cast: buffer not aligned to its element size");

  const uint64_t sb_elems = (uint64_t)vec_elems(dtype) * 64u * (unsigned)slots_per_wave(rmode, dtype);
  const bool vec_ok = (((uintptr_t)in | (uintptr_t)out) % 16u) == 0 &&
                      (rmode != STOCHASTIC || (prm.rest.elem_base % sb_elems) == 0);
  const uint64_t nsb = vec_ok ? n / sb_elems : 0;
  const uint64_t tail = n - nsb * sb_elems;
  const unsigned kCastBlock = wide ? (unsigned)kBlock : (unsigned)cast_block(rmode, dtype);
  uint64_t blocks = superblock_groups(nsb, rmode, dtype);
  if (wide && !sr_by_slot(rmode, dtype)) blocks = (nsb + kWavesPerBlock - 1) / kWavesPerBlock;
  const uint64_t tail_blocks = (tail + kCastBlock - 1) / kCastBlock;
  if (blocks < tail_blocks) blocks = tail_blocks;
  if (blocks > (uint64_t)max_blocks()) blocks = (uint64_t)max_blocks();
  if (blocks == 0) blocks = 1;

  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kCastBlock), 0, (hipStream_t)stream, in, out, nsb,
                     n, prm.scale, prm.rcp, prm.amax_bits, prm.npart, prm.clamp, prm.rest);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
  return FPEMU_OK;
}

int launch_cast64(const void *in, void *out, uint64_t n, int fmt, int rmode, const CastParams &prm, void *stream) {
  if (prm.amax_bits != nullptr || prm.npart != 0u)
    return fail(FPEMU_EINVAL, "float64 storage: only fpemu_cast / fpemu_forward (a scale given by the host) take it");
  const cast64_fn fn = pick_cast64(fmt, rmode);
  if (fn == nullptr) return fail(FPEMU_EINVAL, "fpemu_cast: format / rounding-mode pair not defined");
  if (((uintptr_t)in % 8u) || ((uintptr_t)out % 8u))
    return fail(FPEMU_EALIGN, "fpemu_cast: buffer not aligned to its element size");
  const bool vec_ok = (((uintptr_t)in | (uintptr_t)out) % 16u) == 0;
  const uint64_t nvec = vec_ok ? n / 2u : 0;
  const uint64_t tail = n - nvec * 2u;
  uint64_t blocks = (nvec + kCast64Block - 1) / kCast64Block;
  const uint64_t tail_blocks = (tail + kCast64Block - 1) / kCast64Block;
  if (blocks < tail_blocks) blocks = tail_blocks;
  if (blocks > 0x7FFFFFFFull) return fail(FPEMU_EINVAL, "fpemu_cast: float64 buffer too large for one launch");
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kCast64Block), 0, (hipStream_t)stream, (const double *)in, (double *)out,
                     nvec, n, prm.scale, prm.rcp, prm.rest.seed, prm.rest.offset, prm.rest.elem_base);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
  return FPEMU_OK;
}

// Scratch of the allocation-free entry point (fpemu_quantize_per_tensor_static): word 0 = the amax bit
// pattern, words 1.. = per-block maxima.  One copy per GPU, held until the process ends (1280 bytes).
constexpr uint64_t kStaticWords = 320;
constexpr int kStaticMaxDevices = 64;
static __device__ __align__(256) uint32_t scratch[kStaticWords];
static std::atomic<uint32_t *> g_scratch_ptr[kStaticMaxDevices];  // zero-initialised: not looked up yet

// The current device's copy of `scratch` (looked up on the first call on each device).
int static_scratch(uint32_t **ws) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
  if (dev < 0 || dev >= kStaticMaxDevices)
    return fail(FPEMU_EINVAL, "fpemu_quantize_per_tensor_static: device ordinal beyond the scratch table");
  uint32_t *p = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (p == nullptr) {
    void *sym = nullptr;
    e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));  // the address on the CURRENT device
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    if (sym == nullptr) return fail(FPEMU_EINVAL, "fpemu_quantize_per_tensor_static: no device address for the scratch");
    p = static_cast<uint32_t *>(sym);
    g_scratch_ptr[dev].store(p, std::memory_order_release);  // racing threads look up the same address
  }
  *ws = p;
  return FPEMU_OK;
}

}  // namespace

extern "C" {

int fpemu_abi_version(void) { return FPEMU_ABI_VERSION; }

// internal (not in the public header): lets the other translation units set the error text
void fpemu_set_error_(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }

const char *fpemu_last_error(void) { return g_err; }

// TEST-ONLY (not in the public header): cap on workgroups per cast launch, 0 = default.  The suite uses it to
// drive the kernels' grid-stride loops, which the default one-shot grids never enter; results do not depend on it.
void fpemu_test_set_max_blocks_(int mb) { g_max_blocks.store(mb, std::memory_order_relaxed); }

int fpemu_parse_mode(const char *mode, int *fmt, int *rmode) {
  static const struct { const char *name; int fmt, rmode; } tab[] = {
      {"E5M2_RTZ", E5M2, RTZ},
      {"E5M2_RNE", E5M2, RNE},
      {"E5M2_STOCHASTIC", E5M2, STOCHASTIC},
      {"E5M2_RNAZ", E5M2, RNAZ},
      {"E5M2_RNTZ", E5M2, RNTZ},
      {"E5M2_RPINF", E5M2, RPINF},
      {"E5M2_RNINF", E5M2, RNINF},
      {"E5M2_DAZ_RNE", E5M2_DAZ, RNE},
      {"E5M2_DAZ_STOCHASTIC", E5M2_DAZ, STOCHASTIC},
      {"E5M2_DAZ_RNAZ", E5M2_DAZ, RNAZ},
      {"E5M2_DAZ_RNTZ", E5M2_DAZ, RNTZ},
      {"FLOAT16_RNE", FP16, RNE},
      {"FLOAT16_STOCHASTIC", FP16, STOCHASTIC},
      {"FLOAT16_DAZ_RNE", FP16_DAZ, RNE},
      {"BFLOAT16_RNE", BF16, RNE},
      {"BFLOAT16_STOCHASTIC", BF16, STOCHASTIC},
      {"E4M3_RNE", E4M3, RNE},
      {"E4M3_STOCHASTIC", E4M3, STOCHASTIC},
      {"E3M4_RNE", E3M4, RNE},
      {"E3M4_STOCHASTIC", E3M4, STOCHASTIC},
      // RTZ extensions named by the north star; not defined by the reference
      {"E4M3_RTZ", E4M3, RTZ},
      {"E3M4_RTZ", E3M4, RTZ},
      {"BFLOAT16_RTZ", BF16, RTZ},
      {"E4M3_IEEE_RNE", E4M3_IEEE, RNE},
      {"E4M3_IEEE_STOCHASTIC", E4M3_IEEE, STOCHASTIC},
      {"FP4_NEAREST", FP4, RNE},
  };
  if (mode == nullptr || fmt == nullptr || rmode == nullptr)
    return fail(FPEMU_EINVAL, "fpemu_parse_mode: null argument");
  for (const auto &t : tab) {
    if (strcmp(mode, t.name) == 0) {
      *fmt = t.fmt;
      *rmode = t.rmode;
      return FPEMU_OK;
    }
  }
  return fail(FPEMU_EMODE, "fpemu: unknown mode string");
}

int fpemu_cast(const void *in, void *out, uint64_t n, int dtype, int fmt, int rmode, float scale,
               uint64_t seed, uint64_t offset, uint64_t elem_base, void *stream) {
  // the reference aborts the process on `scale > FLT_MAX` (cpp:772-776, hip:1048-1052); report it instead.  That test
  // is the whole check there: a NaN or -Inf scale runs and gives what the arithmetic gives, and so it does here.
  if (scale > 3.402823466e+38f) return fail(FPEMU_EINVAL, "fpemu_cast: invalid scale factor (> FLT_MAX)");
  CastParams prm;
  prm.scale = scale;
  prm.rcp = (float)(1.0 / (double)scale);  // cpp:714
  prm.amax_bits = nullptr;
  prm.rest.flt_max = 0.f;
  prm.clamp = 0;
  prm.rest.seed = seed;
  prm.rest.offset = offset;
  prm.rest.elem_base = elem_base;
  prm.rest.partials = nullptr;
  prm.npart = 0;
  prm.rest.amax_pub = nullptr;
  return launch_cast(in, out, n, dtype, fmt, rmode, prm, stream);
}

int fpemu_cast_multi(const fpemu_tensor_t *tensors, uint64_t count, int dtype, int fmt, int rmode, float scale,
                     uint64_t seed, uint64_t offset, void *stream) {
  if (count == 0) return FPEMU_OK;
  if (tensors == nullptr) return fail(FPEMU_EINVAL, "fpemu_cast_multi: null tensor table");
  if (scale > 3.402823466e+38f) return fail(FPEMU_EINVAL, "fpemu_cast_multi: invalid scale factor (> FLT_MAX)");
  if (dtype < FPEMU_F32 || dtype > FPEMU_BF16) return fail(FPEMU_EINVAL, "fpemu_cast_multi: bad dtype id");
  const cast_multi_fn fn = dtype == FPEMU_F32   ? pick_cast_multi<FPEMU_F32>(fmt, rmode)
                           : dtype == FPEMU_F16 ? pick_cast_multi<FPEMU_F16>(fmt, rmode)
                                                : pick_cast_multi<FPEMU_BF16>(fmt, rmode);
  if (fn == nullptr) return fail(FPEMU_EINVAL, "fpemu_cast_multi: format / rounding-mode pair not defined");
  const unsigned es = elem_size(dtype);
  for (uint64_t i = 0; i < count; i++) {  // validate everything before the first launch
    if (tensors[i].n != 0 && (tensors[i].in == nullptr || tensors[i].out == nullptr))
      return fail(FPEMU_EINVAL, "fpemu_cast_multi: null buffer");
    if (((uintptr_t)tensors[i].in % es) || ((uintptr_t)tensors[i].out % es))
      return fail(FPEMU_EALIGN, "fpemu_cast_multi: buffer not aligned to its element size");
  }
  const uint64_t sb_elems = (uint64_t)vec_elems(dtype) * 64u * (unsigned)slots_per_wave(rmode, dtype);
  const float rcp = (float)(1.0 / (double)scale);  // cpp:714
  const unsigned kCastBlock = (unsigned)cast_block(rmode, dtype);
  uint64_t i = 0;
  while (i < count) {
    MultiTable tab;
    memset(&tab, 0, sizeof(tab));
    uint64_t blocks = 0;
    unsigned k = 0;
    const uint64_t first = i;
    for (; i < count && k < (unsigned)kMultiTensors; i++, k++) {
      const fpemu_tensor_t &t = tensors[i];
      const bool vec_ok = (((uintptr_t)t.in | (uintptr_t)t.out) % 16u) == 0;
      const uint64_t nsb = vec_ok ? t.n / sb_elems : 0;
      const uint64_t tail = t.n - nsb * sb_elems;
      uint64_t b = superblock_groups(nsb, rmode, dtype);
      const uint64_t tb = (tail + kCastBlock - 1) / kCastBlock;
      if (b < tb) b = tb;
      if (b > (uint64_t)max_blocks()) b = (uint64_t)max_blocks();
      if (blocks + b > 0x7FFFFFFFull) {
        if (k == 0) return fail(FPEMU_EINVAL, "fpemu_cast_multi: tensor too large for one launch; use fpemu_cast");
        break;  // this tensor starts the next launch
      }
      blocks += b;  // an empty tensor owns no workgroups (and still consumes its offset)
      tab.in[k] = t.in;
      tab.out[k] = t.out;
      tab.n[k] = t.n;
      tab.nsb[k] = nsb;
      tab.blk_end[k] = (uint32_t)blocks;
    }
    tab.count = k;
    if (blocks != 0) {
      hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kCastBlock), 0, (hipStream_t)stream, tab, scale, rcp, seed,
                         offset + first);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
  }
  return FPEMU_OK;
}

int fpemu_quantize_per_tensor_multi(const fpemu_tensor_t *tensors, uint64_t count, int dtype, int fmt, int rmode,
                                    float flt_max, int clamp, uint32_t *ws, uint64_t ws_words, uint64_t seed,
                                    uint64_t offset, void *stream) {
  const char *who = "fpemu_quantize_per_tensor_multi";
  char msg[128];
  auto bad = [&](int code, const char *what) {
    snprintf(msg, sizeof(msg), "%s: %s", who, what);
    return fail(code, msg);
  };
  if (count == 0) return FPEMU_OK;
  if (tensors == nullptr) return bad(FPEMU_EINVAL, "null tensor table");
  if (dtype < FPEMU_F32 || dtype > FPEMU_BF16) return bad(FPEMU_EINVAL, "bad dtype id");
  const cast_multi_scaled_fn fn = dtype == FPEMU_F32   ? pick_cast_multi_scaled<FPEMU_F32>(fmt, rmode)
                                  : dtype == FPEMU_F16 ? pick_cast_multi_scaled<FPEMU_F16>(fmt, rmode)
                                                       : pick_cast_multi_scaled<FPEMU_BF16>(fmt, rmode);
  if (fn == nullptr) return bad(FPEMU_EINVAL, "format / rounding-mode pair takes no scale or is not defined");
  const uint64_t per_launch = count < (uint64_t)kMultiTensors ? count : (uint64_t)kMultiTensors;
  if (ws == nullptr || ws_words < count + per_launch) return bad(FPEMU_EINVAL, "workspace smaller than count + min(count, 32) words");
  uint64_t pcap = (ws_words - count) / per_launch;  // words of partials per tensor
  if (pcap > kMaxPartials) pcap = kMaxPartials;
  const unsigned es = elem_size(dtype), ve = vec_elems(dtype);
  for (uint64_t i = 0; i < count; i++) {  // validate everything before the first launch
    if (tensors[i].n != 0 && (tensors[i].in == nullptr || tensors[i].out == nullptr)) return bad(FPEMU_EINVAL, "null buffer");
    if (((uintptr_t)tensors[i].in % es) || ((uintptr_t)tensors[i].out % es))
      return bad(FPEMU_EALIGN, "buffer not aligned to its element size");
  }
  typedef void (*amax_multi_fn)(AmaxTable, uint32_t *);
  const amax_multi_fn afn = dtype == FPEMU_F32   ? (amax_multi_fn)amax_multi_kernel<FPEMU_F32, false>
                            : dtype == FPEMU_F16 ? (amax_multi_fn)amax_multi_kernel<FPEMU_F16, false>
                                                 : (amax_multi_fn)amax_multi_kernel<FPEMU_BF16, false>;
  const uint64_t sb_elems = (uint64_t)ve * 64u * (unsigned)slots_per_wave(rmode, dtype);
  const unsigned kCastBlock = (unsigned)cast_block(rmode, dtype);
  uint32_t *partials = ws + count;
  uint64_t i = 0;
  while (i < count) {
    MultiTable tab;
    AmaxTable at;
    MultiScaleArgs ms;
    memset(&tab, 0, sizeof(tab));
    memset(&at, 0, sizeof(at));
    memset(&ms, 0, sizeof(ms));
    uint64_t blocks = 0, ablocks = 0;
    unsigned k = 0;
    const uint64_t first = i;
    for (; i < count && k < (unsigned)kMultiTensors; i++, k++) {
      const fpemu_tensor_t &t = tensors[i];
      const bool vec_ok = (((uintptr_t)t.in | (uintptr_t)t.out) % 16u) == 0;
      const uint64_t nsb = vec_ok ? t.n / sb_elems : 0;
      const uint64_t tail = t.n - nsb * sb_elems;
      uint64_t b = superblock_groups(nsb, rmode, dtype);
      const uint64_t tb = (tail + kCastBlock - 1) / kCastBlock;
      if (b < tb) b = tb;
      if (b > (uint64_t)max_blocks()) b = (uint64_t)max_blocks();
      if (blocks + b > 0x7FFFFFFFull) {
        if (k == 0) return bad(FPEMU_EINVAL, "tensor too large for one launch; use fpemu_quantize_per_tensor_ws");
        break;  // this tensor starts the next launch
      }
      blocks += b;  // an empty tensor owns no workgroups (and still consumes its offset)
      tab.in[k] = t.in;
      tab.out[k] = t.out;
      tab.n[k] = t.n;
      tab.nsb[k] = nsb;
      tab.blk_end[k] = (uint32_t)blocks;
      // amax workgroups: at least four vectors (or 16 elements) per thread, at most pcap words
      const bool avec = ((uintptr_t)t.in % 16u) == 0;
      const uint64_t nvec = avec ? t.n / ve : 0;
      uint64_t ab = avec ? nvec / (4u * kBlock) : t.n / (16u * kBlock);
      if (ab > pcap) ab = pcap;
      if (ab < 1 && t.n != 0) ab = 1;
      ablocks += ab;
      at.in[k] = t.in;
      at.n[k] = t.n;
      at.nvec[k] = nvec;
      at.blk_end[k] = (uint32_t)ablocks;
      ms.pend[k] = (uint32_t)ablocks;
    }
    tab.count = k;
    at.count = k;
    if (blocks != 0) {
      hipLaunchKernelGGL(afn, dim3((unsigned)ablocks), dim3(kBlock), 0, (hipStream_t)stream, at, partials);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
      ms.partials = partials;
      ms.amax_words = ws + first;
      ms.flt_max = flt_max;
      ms.clamp = clamp ? 1 : 0;
      hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kCastBlock), 0, (hipStream_t)stream, tab, ms, seed, offset + first);
      e = hipGetLastError();
      if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
  }
  return FPEMU_OK;
}

int fpemu_amax_multi(const fpemu_tensor_t *tensors, uint64_t count, int dtype, uint32_t *amax_words, void *stream) {
  const char *who = "fpemu_amax_multi";
  char msg[128];
  auto bad = [&](int code, const char *what) {
    snprintf(msg, sizeof(msg), "%s: %s", who, what);
    return fail(code, msg);
  };
  if (count == 0) return FPEMU_OK;
  if (tensors == nullptr || amax_words == nullptr) return bad(FPEMU_EINVAL, "null argument");
  if (dtype < FPEMU_F32 || dtype > FPEMU_BF16) return bad(FPEMU_EINVAL, "bad dtype id");
  const unsigned es = elem_size(dtype), ve = vec_elems(dtype);
  for (uint64_t i = 0; i < count; i++) {
    if (tensors[i].n != 0 && tensors[i].in == nullptr) return bad(FPEMU_EINVAL, "null buffer");
    if ((uintptr_t)tensors[i].in % es) return bad(FPEMU_EALIGN, "buffer not aligned to its element size");
  }
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, amax_words, count);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
  typedef void (*amax_multi_fn)(AmaxTable, uint32_t *);
  const amax_multi_fn afn = dtype == FPEMU_F32   ? (amax_multi_fn)amax_multi_kernel<FPEMU_F32, true>
                            : dtype == FPEMU_F16 ? (amax_multi_fn)amax_multi_kernel<FPEMU_F16, true>
                                                 : (amax_multi_fn)amax_multi_kernel<FPEMU_BF16, true>;
  uint64_t i = 0;
  while (i < count) {
    AmaxTable at;
    memset(&at, 0, sizeof(at));
    uint64_t ablocks = 0;
    unsigned k = 0;
    const uint64_t first = i;
    for (; i < count && k < (unsigned)kMultiTensors; i++, k++) {
      const fpemu_tensor_t &t = tensors[i];
      // workgroups: at least four vectors (or 16 elements) per thread, at most kAmaxMaxBlocks per tensor
      const bool avec = ((uintptr_t)t.in % 16u) == 0;
      const uint64_t nvec = avec ? t.n / ve : 0;
      uint64_t ab = avec ? nvec / (4u * kBlock) : t.n / (16u * kBlock);
      if (ab > (uint64_t)kAmaxMaxBlocks) ab = (uint64_t)kAmaxMaxBlocks;
      if (ab < 1 && t.n != 0) ab = 1;
      ablocks += ab;
      at.in[k] = t.in;
      at.n[k] = t.n;
      at.nvec[k] = nvec;
      at.blk_end[k] = (uint32_t)ablocks;
    }
    at.count = k;
    if (ablocks != 0) {
      hipLaunchKernelGGL(afn, dim3((unsigned)ablocks), dim3(kBlock), 0, (hipStream_t)stream, at, amax_words + first);
      e = hipGetLastError();
      if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
  }
  return FPEMU_OK;
}

int fpemu_cast_multi_scaled(const fpemu_tensor_t *tensors, uint64_t count, int dtype, int fmt, int rmode,
                            const uint32_t *amax_words, float flt_max, int clamp, uint64_t seed, uint64_t offset,
                            void *stream) {
  const char *who = "fpemu_cast_multi_scaled";
  char msg[128];
  auto bad = [&](int code, const char *what) {
    snprintf(msg, sizeof(msg), "%s: %s", who, what);
    return fail(code, msg);
  };
  if (count == 0) return FPEMU_OK;
  if (tensors == nullptr || amax_words == nullptr) return bad(FPEMU_EINVAL, "null argument");
  if (dtype < FPEMU_F32 || dtype > FPEMU_BF16) return bad(FPEMU_EINVAL, "bad dtype id");
  const cast_multi_scaled_fn fn = dtype == FPEMU_F32   ? pick_cast_multi_scaled<FPEMU_F32>(fmt, rmode)
                                  : dtype == FPEMU_F16 ? pick_cast_multi_scaled<FPEMU_F16>(fmt, rmode)
                                                       : pick_cast_multi_scaled<FPEMU_BF16>(fmt, rmode);
  if (fn == nullptr) return bad(FPEMU_EINVAL, "format / rounding-mode pair takes no scale or is not defined");
  const unsigned es = elem_size(dtype);
  for (uint64_t i = 0; i < count; i++) {  // validate everything before the first launch
    if (tensors[i].n != 0 && (tensors[i].in == nullptr || tensors[i].out == nullptr)) return bad(FPEMU_EINVAL, "null buffer");
    if (((uintptr_t)tensors[i].in % es) || ((uintptr_t)tensors[i].out % es))
      return bad(FPEMU_EALIGN, "buffer not aligned to its element size");
  }
  const uint64_t sb_elems = (uint64_t)vec_elems(dtype) * 64u * (unsigned)slots_per_wave(rmode, dtype);
  const unsigned kCastBlock = (unsigned)cast_block(rmode, dtype);
  uint64_t i = 0;
  while (i < count) {
    MultiTable tab;
    MultiScaleArgs ms;
    memset(&tab, 0, sizeof(tab));
    memset(&ms, 0, sizeof(ms));
    uint64_t blocks = 0;
    unsigned k = 0;
    const uint64_t first = i;
    for (; i < count && k < (unsigned)kMultiTensors; i++, k++) {
      const fpemu_tensor_t &t = tensors[i];
      const bool vec_ok = (((uintptr_t)t.in | (uintptr_t)t.out) % 16u) == 0;
      const uint64_t nsb = vec_ok ? t.n / sb_elems : 0;
      const uint64_t tail = t.n - nsb * sb_elems;
      uint64_t b = superblock_groups(nsb, rmode, dtype);
      const uint64_t tb = (tail + kCastBlock - 1) / kCastBlock;
      if (b < tb) b = tb;
      if (b > (uint64_t)max_blocks()) b = (uint64_t)max_blocks();
      if (blocks + b > 0x7FFFFFFFull) {
        if (k == 0) return bad(FPEMU_EINVAL, "tensor too large for one launch; use fpemu_cast_scaled");
        break;
      }
      blocks += b;
      tab.in[k] = t.in;
      tab.out[k] = t.out;
      tab.n[k] = t.n;
      tab.nsb[k] = nsb;
      tab.blk_end[k] = (uint32_t)blocks;
      ms.pend[k] = k + 1u;  // one word per tensor: the word IS its maximum
    }
    tab.count = k;
    if (blocks != 0) {
      ms.partials = amax_words + first;
      ms.amax_words = nullptr;  // nothing to publish: the caller holds the words already
      ms.flt_max = flt_max;
      ms.clamp = clamp ? 1 : 0;
      hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kCastBlock), 0, (hipStream_t)stream, tab, ms, seed, offset + first);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
  }
  return FPEMU_OK;
}

int fpemu_forward(const void *in, void *out, int dtype, const char *mode, uint64_t size, float scale,
                  int block_norm, int block_size, uint64_t seed, uint64_t offset, void *stream) {
  int fmt = 0, rmode = 0;
  int rc = fpemu_parse_mode(mode, &fmt, &rmode);
  if (rc != FPEMU_OK) return rc;
  if (block_norm && block_size > 0 && (uint64_t)block_size != size && (size % (uint64_t)block_size) != 0)
    block_norm = 0;  // cpp:2020-2025
  // formats whose reference kernel has a block-norm branch; the others ignore the flag (cpp:1913-1957)
  if (block_norm && block_size > 0 && (fmt == E5M2 || fmt == E4M3 || fmt == E3M4 || fmt == E4M3_IEEE || fmt == FP4)) {
    if (scale > 3.402823466e+38f) return fail(FPEMU_EINVAL, "fpemu_forward: invalid scale factor (> FLT_MAX)");
    return fpemu_quantize_blocks(in, out, size, (uint64_t)block_size, dtype, fmt, rmode, seed, offset, stream);
  }
  return fpemu_cast(in, out, size, dtype, fmt, rmode, scale, seed, offset, 0, stream);
}

static int launch_amax(const void *in, uint64_t n, int dtype, uint32_t *amax_bits, bool keep, void *stream) {
  if (n == 0) return FPEMU_OK;
  if (in == nullptr || amax_bits == nullptr) return fail(FPEMU_EINVAL, "fpemu_amax: null argument");
  if (dtype < FPEMU_F32 || dtype > FPEMU_BF16) return fail(FPEMU_EINVAL, "fpemu_amax: bad dtype id");
  if ((uintptr_t)in % elem_size(dtype)) return fail(FPEMU_EALIGN, "fpemu_amax: buffer not aligned");
  const bool vec_ok = ((uintptr_t)in % 16u) == 0;
  const uint64_t nvec = vec_ok ? n / vec_elems(dtype) : 0;
  // one superblock (kAmaxVPT x 64 vectors) per wave; unaligned buffers: one element per lane-step
  const uint64_t nsb = nvec / (64u * kAmaxVPT);
  uint64_t blocks = (nsb + kWavesPerBlock - 1) / kWavesPerBlock;
  const uint64_t rest = n - nsb * 64u * kAmaxVPT * vec_elems(dtype);
  uint64_t rest_blocks = (rest + kBlock - 1) / kBlock;
  if (rest_blocks > (uint64_t)kAmaxMaxBlocks) rest_blocks = (uint64_t)kAmaxMaxBlocks;
  if (blocks < rest_blocks) blocks = rest_blocks;
  if (blocks > 0x7FFFFFFFull) blocks = 0x7FFFFFFFull;
  if (blocks == 0) blocks = 1;
  typedef void (*amax_fn)(const void *, uint64_t, uint64_t, uint32_t *);
  const amax_fn fn = keep ? (dtype == FPEMU_F32   ? (amax_fn)amax_kernel<FPEMU_F32, true, false>
                             : dtype == FPEMU_F16 ? (amax_fn)amax_kernel<FPEMU_F16, true, false>
                                                  : (amax_fn)amax_kernel<FPEMU_BF16, true, false>)
                          : (dtype == FPEMU_F32   ? (amax_fn)amax_kernel<FPEMU_F32, false, false>
                             : dtype == FPEMU_F16 ? (amax_fn)amax_kernel<FPEMU_F16, false, false>
                                                  : (amax_fn)amax_kernel<FPEMU_BF16, false, false>);
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, in, nvec, n,
                     amax_bits);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
  return FPEMU_OK;
}

int fpemu_amax(const void *in, uint64_t n, int dtype, uint32_t *amax_bits, void *stream) {
  return launch_amax(in, n, dtype, amax_bits, /*keep=*/false, stream);
}

int fpemu_cast_scaled(const void *in, void *out, uint64_t n, int dtype, int fmt, int rmode,
                      const uint32_t *amax_bits, float flt_max, int clamp, uint64_t seed,
                      uint64_t offset, uint64_t elem_base, void *stream) {
  if (amax_bits == nullptr) return fail(FPEMU_EINVAL, "fpemu_cast_scaled: null amax word");
  CastParams prm;
  prm.scale = 1.f;
  prm.rcp = 1.f;
  prm.amax_bits = amax_bits;
  prm.rest.flt_max = flt_max;
  prm.clamp = clamp ? 1 : 0;
  prm.rest.seed = seed;
  prm.rest.offset = offset;
  prm.rest.elem_base = elem_base;
  prm.rest.partials = nullptr;
  prm.npart = 0;
  prm.rest.amax_pub = nullptr;
  return launch_cast(in, out, n, dtype, fmt, rmode, prm, stream);
}

int fpemu_cast_device_scale(const void *in, void *out, uint64_t n, int dtype, int fmt, int rmode, const float *scale,
                            uint64_t seed, uint64_t offset, uint64_t elem_base, void *stream) {
  if (scale == nullptr) return fail(FPEMU_EINVAL, "fpemu_cast_device_scale: null scale");
  CastParams prm;
  prm.scale = 1.f;
  prm.rcp = 1.f;
  prm.amax_bits = reinterpret_cast<const uint32_t *>(scale);
  prm.rest.flt_max = 0.f;
  prm.clamp = 2;  // DeviceScale::resolve: the word is the scale
  prm.rest.seed = seed;
  prm.rest.offset = offset;
  prm.rest.elem_base = elem_base;
  prm.rest.partials = nullptr;
  prm.npart = 0;
  prm.rest.amax_pub = nullptr;
  return launch_cast(in, out, n, dtype, fmt, rmode, prm, stream);
}

int fpemu_quantize_per_tensor(const void *in, void *out, uint64_t n, int dtype, int fmt, int rmode,
                              float flt_max, int clamp, uint32_t *amax_ws, uint64_t seed,
                              uint64_t offset, void *stream) {
  if (amax_ws == nullptr) return fail(FPEMU_EINVAL, "fpemu_quantize_per_tensor: null workspace");
  if (n == 0) return FPEMU_OK;
  // Small tensors (the per-layer casts of configs 3/4): ONE launch -- a single workgroup reduces,
  // publishes the amax word and casts (the rows kernel with one row).  Larger ones are latency-bound in one
  // workgroup (17 us at 32 Ki elements, measured in a ResNet-50 step) and take the multi-launch paths.  Stochastic noise of row 0
  // at `offset` is exactly the flat layout's, so results do not depend on which path ran.
  if (dtype >= FPEMU_F32 && dtype <= FPEMU_BF16 && n <= kSmallVectors * vec_elems(dtype) && in != nullptr && out != nullptr &&
      ((uintptr_t)in % elem_size(dtype)) == 0 && ((uintptr_t)out % elem_size(dtype)) == 0) {
    const int rc1 = fpemu_rows_launch_(in, out, 1, n, dtype, fmt, rmode, flt_max, clamp ? 1 : 0, seed, offset,
                                       amax_ws, /*force_block=*/1, /*col_base=*/0, stream);
    if (rc1 != kNoRowsKernel) return rc1;  // formats without a scale fall through to the flat path
  }
  hipLaunchKernelGGL(zero_word_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, amax_ws);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
  // the cast re-reads what the amax pass just read: let a tensor that fits the Infinity Cache stay there
  const bool keep = dtype >= FPEMU_F32 && dtype <= FPEMU_BF16 && n * elem_size(dtype) <= kKeepBytes;
  int rc = launch_amax(in, n, dtype, amax_ws, keep, stream);
  if (rc != FPEMU_OK) return rc;
  return fpemu_cast_scaled(in, out, n, dtype, fmt, rmode, amax_ws, flt_max, clamp, seed, offset, 0, stream);
}

// word: receives the amax bit pattern; partials / max_partials: scratch for the per-block maxima (may be 0)
static int quantize_per_tensor_impl(const void *in, void *out, uint64_t n, int dtype, int fmt, int rmode,
                                    float flt_max, int clamp, uint32_t *word, uint32_t *partials,
                                    uint64_t max_partials, uint64_t seed, uint64_t offset, void *stream) {
  if (n == 0) return FPEMU_OK;
  const bool args_ok = in != nullptr && out != nullptr && dtype >= FPEMU_F32 && dtype <= FPEMU_BF16 &&
                       ((uintptr_t)in % elem_size(dtype)) == 0 && ((uintptr_t)out % elem_size(dtype)) == 0;
  // above the one-workgroup size: two launches -- per-block maxima into the scratch, then the cast, whose
  // blocks each reduce those maxima (L2 hits) -- instead of zero + atomic amax + cast
  if (args_ok && n > kSmallVectors * vec_elems(dtype) && n * elem_size(dtype) <= kPartialsBytes && max_partials >= 1 &&
      pick_cast_dt(dtype, fmt, rmode) != nullptr && fmt != BF16 && fmt != FP16 && fmt != FP16_DAZ) {
    const bool vec_ok = ((uintptr_t)in % 16u) == 0;
    const uint64_t nvec = vec_ok ? n / vec_elems(dtype) : 0;
    uint64_t nb = vec_ok ? nvec / 256u : n / 1024u;   // >= one vector (or four elements) per thread
    if (nb > kMaxPartials) nb = kMaxPartials;
    if (nb > max_partials) nb = max_partials;
    if (nb < 1) nb = 1;
    typedef void (*amax_fn)(const void *, uint64_t, uint64_t, uint32_t *);
    const amax_fn fn = dtype == FPEMU_F32   ? (amax_fn)amax_kernel<FPEMU_F32, true, true>
                       : dtype == FPEMU_F16 ? (amax_fn)amax_kernel<FPEMU_F16, true, true>
                                            : (amax_fn)amax_kernel<FPEMU_BF16, true, true>;
    hipLaunchKernelGGL(fn, dim3((unsigned)nb), dim3(kBlock), 0, (hipStream_t)stream, in, nvec, n, partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    CastParams prm;
    prm.scale = 1.f;
    prm.rcp = 1.f;
    prm.amax_bits = nullptr;
    prm.rest.flt_max = flt_max;
    prm.clamp = clamp;
    prm.rest.seed = seed;
    prm.rest.offset = offset;
    prm.rest.elem_base = 0;
    prm.rest.partials = partials;
    prm.npart = (uint32_t)nb;
    prm.rest.amax_pub = word;
    return launch_cast(in, out, n, dtype, fmt, rmode, prm, stream);
  }
  return fpemu_quantize_per_tensor(in, out, n, dtype, fmt, rmode, flt_max, clamp, word, seed, offset, stream);
}

int fpemu_quantize_per_tensor_ws(const void *in, void *out, uint64_t n, int dtype, int fmt, int rmode,
                                 float flt_max, int clamp, uint32_t *ws, uint64_t ws_words, uint64_t seed,
                                 uint64_t offset, void *stream) {
  if (ws == nullptr || ws_words < 1) return fail(FPEMU_EINVAL, "fpemu_quantize_per_tensor_ws: null workspace");
  return quantize_per_tensor_impl(in, out, n, dtype, fmt, rmode, flt_max, clamp, ws, ws + 1, ws_words - 1, seed, offset,
                                  stream);
}

// fpemu_quantize_per_tensor_static: fpemu_quantize_per_tensor_ws with the library's own scratch instead of a
// caller-provided workspace, for callers that cannot allocate (HIP graph capture, serving loops).  The scratch
// is a file-scope __device__ array: every GPU the library runs on holds its own copy at its own address, so
// the address is looked up once per device (hipGetSymbolAddress with that device current) and only that
// pointer -- never the array's name, which on the host is a stand-in, not a device address -- is handed to
// kernels.  Nothing here calls hipMalloc / hipFree or any allocating routine.
int fpemu_quantize_per_tensor_static(const void *in, void *out, uint64_t n, int dtype, int fmt, int rmode,
                                     float flt_max, int clamp, uint32_t *amax_out, uint64_t seed,
                                     uint64_t offset, void *stream) {
  if (n > FPEMU_STATIC_MAX_ELEMS)
    return fail(FPEMU_EINVAL, "fpemu_quantize_per_tensor_static: more than FPEMU_STATIC_MAX_ELEMS (2^36) elements");
  if (n == 0) return FPEMU_OK;
  uint32_t *ws = nullptr;
  const int rc = static_scratch(&ws);
  if (rc != FPEMU_OK) return rc;
  if (amax_out == nullptr)
    return fpemu_quantize_per_tensor_ws(in, out, n, dtype, fmt, rmode, flt_max, clamp, ws, kStaticWords, seed, offset,
                                        stream);
  // the caller wants the amax word in its own memory: same launches, word 0 redirected
  return quantize_per_tensor_impl(in, out, n, dtype, fmt, rmode, flt_max, clamp, amax_out, ws + 1, kStaticWords - 1,
                                  seed, offset, stream);
}

int fpemu_quantize_rows(const void *in, void *out, uint64_t rows, uint64_t cols, int dtype, int fmt,
                        int rmode, float flt_max, int clamp, uint64_t seed, uint64_t offset,
                        void *stream) {
  if (rows == 0 || cols == 0) return FPEMU_OK;
  if (in == nullptr || out == nullptr) return fail(FPEMU_EINVAL, "fpemu_quantize_rows: null buffer");
  if (dtype < FPEMU_F32 || dtype > FPEMU_BF16) return fail(FPEMU_EINVAL, "fpemu_quantize_rows: bad dtype id");
  if (rows > 0x7FFFFFFFull) return fail(FPEMU_EINVAL, "fpemu_quantize_rows: more than 2^31-1 rows");
  const unsigned es = elem_size(dtype);
  if (((uintptr_t)in % es) || ((uintptr_t)out % es))
    return fail(FPEMU_EALIGN, "fpemu_quantize_rows: buffer not aligned to its element size");
  const int rc = fpemu_rows_launch_(in, out, rows, cols, dtype, fmt, rmode, flt_max, clamp ? 1 : 0, seed, offset,
                                    nullptr, 0, 0, stream);
  if (rc == kNoRowsKernel)
    return fail(FPEMU_EINVAL, "fpemu_quantize_rows: format / rounding-mode pair takes no scale or is not defined");
  return rc;
}

int fpemu_cast_rows_scaled(const void *in, void *out, uint64_t rows, uint64_t cols, int dtype, int fmt, int rmode,
                           const float *scales, uint64_t seed, uint64_t offset, void *stream) {
  if (rows == 0 || cols == 0) return FPEMU_OK;
  if (in == nullptr || out == nullptr) return fail(FPEMU_EINVAL, "fpemu_cast_rows_scaled: null buffer");
  if (scales == nullptr) return fail(FPEMU_EINVAL, "fpemu_cast_rows_scaled: null scales");
  if (dtype < FPEMU_F32 || dtype > FPEMU_BF16) return fail(FPEMU_EINVAL, "fpemu_cast_rows_scaled: bad dtype id");
  if (rows > 0x7FFFFFFFull) return fail(FPEMU_EINVAL, "fpemu_cast_rows_scaled: more than 2^31-1 rows");
  const unsigned es = elem_size(dtype);
  if (((uintptr_t)in % es) || ((uintptr_t)out % es) || ((uintptr_t)scales % 4u))
    return fail(FPEMU_EALIGN, "fpemu_cast_rows_scaled: buffer not aligned to its element size");
  const int rc = fpemu_rows_launch_(in, out, rows, cols, dtype, fmt, rmode, 0.f, /*scale given*/ 3, seed, offset,
                                    const_cast<uint32_t *>(reinterpret_cast<const uint32_t *>(scales)), 0, 0, stream);
  if (rc == kNoRowsKernel)
    return fail(FPEMU_EINVAL, "fpemu_cast_rows_scaled: format / rounding-mode pair takes no scale or is not defined");
  return rc;
}

int fpemu_amax_rows(const void *in, uint64_t rows, uint64_t cols, int dtype, uint32_t *amax_bits, void *stream) {
  if (rows == 0) return FPEMU_OK;
  if (amax_bits == nullptr) return fail(FPEMU_EINVAL, "fpemu_amax_rows: null amax vector");
  if (dtype < FPEMU_F32 || dtype > FPEMU_BF16) return fail(FPEMU_EINVAL, "fpemu_amax_rows: bad dtype id");
  if (rows > 0x7FFFFFFFull) return fail(FPEMU_EINVAL, "fpemu_amax_rows: more than 2^31-1 rows");
  if (cols == 0) {  // empty rows: the maximum of nothing is the zero word
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((rows + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, amax_bits, rows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return FPEMU_OK;
  }
  if (in == nullptr) return fail(FPEMU_EINVAL, "fpemu_amax_rows: null buffer");
  if ((uintptr_t)in % elem_size(dtype)) return fail(FPEMU_EALIGN, "fpemu_amax_rows: buffer not aligned to its element size");
  return fpemu_rows_amax_launch_(in, rows, cols, dtype, amax_bits, stream);
}

int fpemu_cast_rows_amax(const void *in, void *out, uint64_t rows, uint64_t cols, int dtype, int fmt, int rmode,
                         const uint32_t *amax_bits, float flt_max, int clamp, uint64_t seed, uint64_t offset,
                         uint64_t col_base, void *stream) {
  if (rows == 0 || cols == 0) return FPEMU_OK;
  if (in == nullptr || out == nullptr) return fail(FPEMU_EINVAL, "fpemu_cast_rows_amax: null buffer");
  if (amax_bits == nullptr) return fail(FPEMU_EINVAL, "fpemu_cast_rows_amax: null amax vector");
  if (dtype < FPEMU_F32 || dtype > FPEMU_BF16) return fail(FPEMU_EINVAL, "fpemu_cast_rows_amax: bad dtype id");
  if (rows > 0x7FFFFFFFull) return fail(FPEMU_EINVAL, "fpemu_cast_rows_amax: more than 2^31-1 rows");
  const unsigned es = elem_size(dtype);
  if (((uintptr_t)in % es) || ((uintptr_t)out % es) || ((uintptr_t)amax_bits % 4u))
    return fail(FPEMU_EALIGN, "fpemu_cast_rows_amax: buffer not aligned to its element size");
  const int rc = fpemu_rows_launch_(in, out, rows, cols, dtype, fmt, rmode, flt_max, /*amax words given*/ clamp ? 5 : 4, seed,
                                    offset, const_cast<uint32_t *>(amax_bits), 0, col_base, stream);
  if (rc == kNoRowsKernel)
    return fail(FPEMU_EINVAL, "fpemu_cast_rows_amax: format / rounding-mode pair takes no scale or is not defined");
  return rc;
}

int fpemu_quantize_blocks(const void *in, void *out, uint64_t n, uint64_t block_size, int dtype, int fmt,
                          int rmode, uint64_t seed, uint64_t offset, void *stream) {
  if (n == 0) return FPEMU_OK;
  if (in == nullptr || out == nullptr) return fail(FPEMU_EINVAL, "fpemu_quantize_blocks: null buffer");
  if (dtype < FPEMU_F32 || dtype > FPEMU_BF16) return fail(FPEMU_EINVAL, "fpemu_quantize_blocks: bad dtype id");
  if (block_size == 0 || (n % block_size) != 0)
    return fail(FPEMU_EINVAL, "fpemu_quantize_blocks: size is not a multiple of block_size");
  if (!(fmt == E5M2 || fmt == E4M3 || fmt == E3M4 || fmt == E4M3_IEEE || fmt == FP4))
    return fail(FPEMU_EINVAL, "fpemu_quantize_blocks: the reference defines no block norm for this format");
  const uint64_t rows = n / block_size;
  if (rows > 0x7FFFFFFFull) return fail(FPEMU_EINVAL, "fpemu_quantize_blocks: more than 2^31-1 blocks");
  const unsigned es = elem_size(dtype);
  if (((uintptr_t)in % es) || ((uintptr_t)out % es))
    return fail(FPEMU_EALIGN, "fpemu_quantize_blocks: buffer not aligned to its element size");
  const int rc = fpemu_rows_launch_(in, out, rows, block_size, dtype, fmt, rmode, 0.f, /*block norm*/ 2, seed,
                                    offset, nullptr, 0, 0, stream);
  if (rc == kNoRowsKernel)
    return fail(FPEMU_EINVAL, "fpemu_quantize_blocks: format / rounding-mode pair not defined");
  return rc;
}

}  // extern "C"
