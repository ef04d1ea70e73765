// fpemu_common.hpp -- pieces shared by the kernel translation units (fpemu_hip.hip, fpemu_rows.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fpemu_hip.h"
#include "fpemu_device.hpp"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace fpemu {

constexpr int kBlock = 256;         // 4 wavefronts
constexpr int kWavesPerBlock = 4;

template <int DT> struct Storage;
template <> struct Storage<FPEMU_F32> { static constexpr unsigned V = 4; using elem = float; };
template <> struct Storage<FPEMU_F16> { static constexpr unsigned V = 8; using elem = uint16_t; };
template <> struct Storage<FPEMU_BF16> { static constexpr unsigned V = 8; using elem = uint16_t; };

__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  uint32_t u = f2u(f);
  const uint32_t rn = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
  const uint32_t nan = (u >> 16) | 0x0040u;
  return ((u & 0x7FFFFFFFu) > 0x7F800000u) ? nan : rn;
}

// Streaming 16-byte store with the "sc1 nt" policy bits (write-through, non-temporal): measured
// +2 % over a plain nt store on MI355X for this read-once / write-once stream (tools/sweep.hip);
// the builtin only offers "nt".  Vector store only; nothing in the launch reads it back.
// The trailing s_nop 1 is required: hipcc pads no hazards inside an asm statement, and a VALU
// write to the data registers right after a >64-bit store would overtake the store's read of them.
__device__ __forceinline__ void store_stream(u32x4 *p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

template <int DT> __device__ __forceinline__ float load_elem16(uint32_t b) {
  if constexpr (DT == FPEMU_F16) return h_to_f32(b);
  else return u2f(b << 16);
}
template <int DT> __device__ __forceinline__ uint32_t store_elem16(float f) {
  if constexpr (DT == FPEMU_F16) return f32_to_h(f);
  else return f32_to_bf16_bits(f);
}

// Two fp32 results -> one dword of 16-bit storage (RNE).  Written as vector converts so that gfx950's
// packed converts (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32) do the rounding: one instruction for the pair
// instead of the integer RNE sequence per element.  (NaN payloads are the hardware's; every other value
// is the same RNE.)
template <int DT> __device__ __forceinline__ uint32_t pack_pair16(float y0, float y1) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t y = {y0, y1};
  if constexpr (DT == FPEMU_F16) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(y, h2_t));
  } else {
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(y, b2_t));
  }
}


// One 16-byte vector of storage DT through the cast.  `noise` supplies the random bits of the vector's elements
// (used by stochastic modes only): one(e) for element e, pair(e) for elements e and e + 1 (e even), one per
// 16-bit half.  Formats with a packed quantiser (fpemu_device.hpp, has_packed_quant) go two elements at a time.
struct NoNoise {
  __device__ __forceinline__ uint32_t one(unsigned) const { return 0u; }
  __device__ __forceinline__ uint32_t pair(unsigned) const { return 0u; }
};

template <int FMT, int RM, int DT, class Noise>
__device__ __forceinline__ u32x4 cast_vector(const u32x4 v, float scale, float rcp, const Noise &noise) {
  constexpr bool SR = (RM == STOCHASTIC);
  u32x4 o;
  if constexpr (has_packed_quant<FMT, RM>() && DT == FPEMU_F32) {
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      const uint32_t rnd2 = SR ? noise.pair(j) : 0u;
      float y0, y1;
      cast_pair<FMT, RM>(u2f(v[j]), u2f(v[j + 1]), scale, rcp, rnd2, y0, y1);
      o[j] = f2u(y0);
      o[j + 1] = f2u(y1);
    }
  } else if constexpr (has_packed_quant<FMT, RM>()) {
#pragma unroll
    for (int j = 0; j < 4; j++) {  // 16-bit storage: each dword is a pair already
      const uint32_t rnd2 = SR ? noise.pair(2 * j) : 0u;
      float y0, y1;
      cast_pair<FMT, RM>(load_elem16<DT>(v[j] & 0xFFFFu), load_elem16<DT>(v[j] >> 16), scale, rcp, rnd2, y0, y1);
      o[j] = pack_pair16<DT>(y0, y1);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if constexpr (DT == FPEMU_F32) {
        o[j] = f2u(cast_one<FMT, RM>(u2f(v[j]), scale, rcp, SR ? noise.one(j) : 0u));
      } else {
        const float y0 = cast_one<FMT, RM>(load_elem16<DT>(v[j] & 0xFFFFu), scale, rcp, SR ? noise.one(2 * j) : 0u);
        const float y1 = cast_one<FMT, RM>(load_elem16<DT>(v[j] >> 16), scale, rcp, SR ? noise.one(2 * j + 1) : 0u);
        o[j] = pack_pair16<DT>(y0, y1);
      }
    }
  }
  return o;
}

// Noise of a vector whose first element has index `base` in the (seed, call) stream: the general form
// (rows / blocks, where a vector's position in its row decides the counter).
template <int FMT, unsigned V>
struct ElemNoise {
  uint64_t seed, call, base;
  __device__ __forceinline__ uint32_t one(unsigned e) const { return rand_bits_elem<FMT, V>(seed, call, base + e); }
  __device__ __forceinline__ uint32_t pair(unsigned e) const { return one(e) | (one(e + 1u) << 16); }
};

// The same noise for a WHOLE 16-byte vector whose first element has an index that is a multiple of V: its V elements are
// one slot of one lane of the layout, so their noise bytes are consecutive bytes of ONE Philox call -- one draw per
// vector instead of one per element (ElemNoise cannot share them: the compiler does not see that base + e stays in the
// slot).  Stochastic rows / block norm, fp32: 3.6-3.9 -> the deterministic kernels' rate (tools/rows_bench.py).
template <int FMT, unsigned V>
struct VecNoise {
  static constexpr unsigned NB = noise_bytes<FMT>();
  static constexpr unsigned W = V * NB / 4u;   // words of the call that cover the vector
  static_assert(V * NB <= 16u && (V * NB) % 4u == 0u, "a vector's noise lies inside one Philox call");
  uint32_t w[W];
  __device__ __forceinline__ VecNoise(uint64_t seed, uint64_t call, uint64_t base) {
    constexpr uint64_t SB = (uint64_t)V * 64u * 4u;
    const uint64_t sb = base / SB;
    const unsigned r = (unsigned)(base % SB);
    const unsigned slot = r / (V * 64u);
    const unsigned lane = (r / V) % 64u;
    const unsigned pos = slot * V * NB;          // byte of the lane's draw where the vector's noise starts
    const Philox4 p = philox4x32_10(seed, ((sb * 64u + lane) << 2) | (uint64_t)(pos / 16u), call);
#pragma unroll
    for (unsigned i = 0; i < W; i++) w[i] = philox_word(p, (pos % 16u) / 4u + i);
  }
  __device__ __forceinline__ uint32_t one(unsigned e) const {
    const unsigned pos = e * NB;                 // compile-time after unrolling
    const uint32_t v = w[pos / 4u] >> ((pos % 4u) * 8u);
    return NB == 2 ? (v & 0xFFFFu) : (v & 0xFFu);
  }
  __device__ __forceinline__ uint32_t pair(unsigned e) const {   // e even; one noise byte per element
    if constexpr (NB == 1) return __builtin_amdgcn_perm(0u, w[e / 4u], (e % 4u) == 0u ? 0x0C010C00u : 0x0C030C02u);
    else return one(e) | (one(e + 1u) << 16);
  }
};

// One step of an ASCENDING max butterfly: m <- max(m, m of the partner group of OFF lanes).  Before the step m is
// uniform over aligned groups of OFF lanes (steps 1, 2, ... OFF/2 have run), so any lane of the partner group will do:
// OFF 1 / 2 are quad permutes, OFF 4 / 8 the half-row / row mirrors (lane i <-> 7-i / 15-i lands in the partner
// group) -- DPP modifiers on the v_max itself, no LDS crossbar round trip.  OFF 16 / 32 cross DPP rows: ds_bpermute.
template <unsigned OFF> __device__ __forceinline__ uint32_t xor_max_step(uint32_t m) {
  uint32_t o;
  if constexpr (OFF == 1) o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
  else if constexpr (OFF == 2) o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  else if constexpr (OFF == 4) o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xF, 0xF, false);  // row_half_mirror
  else if constexpr (OFF == 8) o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xF, 0xF, false);  // row_mirror
  else o = (uint32_t)__shfl_xor((int)m, (int)OFF, 64);
  return m > o ? m : o;
}

// max over the aligned group of G lanes (G a power of two <= 64, uniform), result in every lane of the group
__device__ __forceinline__ uint32_t group_max_u32(uint32_t m, unsigned G) {
  if (1u < G) m = xor_max_step<1>(m);
  if (2u < G) m = xor_max_step<2>(m);
  if (4u < G) m = xor_max_step<4>(m);
  if (8u < G) m = xor_max_step<8>(m);
  if (16u < G) m = xor_max_step<16>(m);
  if (32u < G) m = xor_max_step<32>(m);
  return m;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t m) {
  m = xor_max_step<1>(m);
  m = xor_max_step<2>(m);
  m = xor_max_step<4>(m);
  m = xor_max_step<8>(m);
  m = xor_max_step<16>(m);
  return xor_max_step<32>(m);
}

template <int DT> __device__ __forceinline__ uint32_t absmax_word(uint32_t m, uint32_t w) {
  if constexpr (DT == FPEMU_F32) {
    const uint32_t a = w & 0x7FFFFFFFu;
    return m > a ? m : a;
  } else {
    const uint32_t a = w & 0x7FFFu, b = (w >> 16) & 0x7FFFu;
    const uint32_t c = a > b ? a : b;
    return m > c ? m : c;
  }
}

inline unsigned elem_size(int dtype) { return dtype == FPEMU_F32 ? 4u : dtype == FPEMU_F64 ? 8u : 2u; }
inline unsigned vec_elems(int dtype) { return dtype == FPEMU_F32 ? 4u : dtype == FPEMU_F64 ? 2u : 8u; }

}  // namespace fpemu

// thread-local error text of the library (fpemu_hip.hip); returns `code`
extern "C" void fpemu_set_error_(const char *msg);
