// fpemu_device.hpp -- per-element arithmetic of the fpemu cast for gfx950 (CDNA4).
//
// Everything here is a pure function on register values: the 16-bit quantisers work on the
// fp16 bit pattern held in a 32-bit VGPR, branch-free (v_cndmask selects), so a wavefront never
// diverges on data.  Semantics follow the reference's scalar routines
//   cvt_fp32_e5m2_scalar  mpemu/pytquant/cpp/fpemu_impl.cpp:691-763
//   cvt_fp32_e4m3_scalar  mpemu/pytquant/cpp/fpemu_impl.cpp:1038-1133
//   cvt_fp32_e3m4_scalar  mpemu/pytquant/cpp/fpemu_impl.cpp:1624-1718
//   E5M2_DAZ_Kernel       mpemu/pytquant/cpp/fpemu_impl.cpp:846-919
//   BFLOAT16 (RNE :207-232 canonical, SR :302-318), FLOAT16_Kernel :356-423
//   E4M3_IEEE: the AVX-512 routines :1213-1328 (canonical: the scalar one saturates from 15.0, :1364)
//   FP4_NEAREST: cvt_fp32_fp4_nearest_scalar :1797-1824
// restated as integer arithmetic on the whole 15-bit magnitude (exponent and mantissa are
// contiguous, so the reference's "mantissa += inc; mantissa + (exp << 10)" carry is a plain add).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fpemu {

enum Fmt : int { E5M2 = 0, E4M3 = 1, E3M4 = 2, BF16 = 3, FP16 = 4, E5M2_DAZ = 5, FP16_DAZ = 6,
                 E4M3_IEEE = 7, FP4 = 8 };
enum RMode : int { RTZ = 0, RNE = 1, STOCHASTIC = 2, RNAZ = 3, RNTZ = 4, RPINF = 5, RNINF = 6 };

__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// fp32 -> fp16 RNE (v_cvt_f16_f32; kernel mode keeps fp16/fp32 denormals, round-to-nearest-even)
__device__ __forceinline__ uint32_t f32_to_h(float p) {
  _Float16 h = (_Float16)p;
  return (uint32_t)__builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float h_to_f32(uint32_t h) {
  return (float)__builtin_bit_cast(_Float16, (uint16_t)h);
}

// Rounding increment (0/1) at bit K of `field`; neg = sign bit set, nz = magnitude non-zero.
template <int RM, int K>
__device__ __forceinline__ uint32_t round_inc(uint32_t field, bool neg, bool nz) {
  constexpr uint32_t LOWM = (1u << K) - 1u;
  constexpr uint32_t HALF = 1u << (K - 1);
  const uint32_t low = field & LOWM;
  if constexpr (RM == RNE) {
    // low > half, or tie with the kept LSB set  ==  carry out of (low + half-1 + lsb)
    return ((low + (HALF - 1u) + ((field >> K) & 1u)) >> K) & 1u;
  } else if constexpr (RM == RNAZ) {
    return low >= HALF;
  } else if constexpr (RM == RNTZ) {
    return low > HALF;
  } else if constexpr (RM == RPINF) {  // nearest-style, only for positive values (cpp:752)
    return (low >= HALF) && !neg && nz;
  } else if constexpr (RM == RNINF) {  // nearest-style, only for negative values (cpp:754)
    return (low >= HALF) && neg && nz;
  } else {
    return 0u;
  }
}

// E5M2 / E5M2_DAZ on the fp16 pattern h (16 significant bits).  rnd: >= 8 random bits.
template <int RM, bool DAZ>
__device__ __forceinline__ uint32_t quant_e5m2(uint32_t h, uint32_t rnd) {
  const uint32_t ef = h & 0x7C00u;
  const bool naninf = (ef == 0x7C00u);
  const bool denorm = (ef == 0u);
  const bool neg = (h & 0x8000u) != 0u;
  const bool nz = (h & 0x7FFFu) != 0u;
  uint32_t r;
  if constexpr (RM == STOCHASTIC) {
    const uint32_t sr = h + (rnd & 0xFFu);                         // fp16 normals: add noise (cpp:739)
    const uint32_t ne = h + (round_inc<RNE, 8>(h, neg, nz) << 8);  // fp16 denormals: RNE (cpp:741)
    r = denorm ? ne : sr;
  } else {
    r = h + (round_inc<RM, 8>(h, neg, nz) << 8);
  }
  if constexpr (DAZ) r = denorm ? 0u : r;  // cpp:911-914 (sign dropped as well)
  r = naninf ? h : r;                      // Inf stays Inf, quiet NaN stays NaN
  return r & 0xFF00u;
}

// E4M3 / E3M4 on the fp16 pattern.  Constants (cpp:1068-1094 / cpp:1654-1678):
//   K      dropped mantissa bits (7 / 6)
//   SAT    magnitude of the largest value as fp16 bits (0x5F00 = 448 / 0x4F80 = 30)
//   FLUSH  smallest magnitude kept (0x1800 = 2^-9 / 0x2400 = 2^-6); below: +-0
//   ENORM  fp16 exponent field of the smallest normal (9 = 2^-6 / 13 = 2^-2)
//   SUB_RNE  under STOCHASTIC the pseudo-subnormal range rounds RNE (E4M3_IEEE, cpp:1312-1315)
template <int RM, int K, uint32_t SAT, uint32_t FLUSH, int ENORM, bool SUB_RNE = false>
__device__ __forceinline__ uint32_t quant_e4m3_e3m4(uint32_t h, uint32_t rnd) {
  const uint32_t a = h & 0x7FFFu;
  const uint32_t sgn = h & 0x8000u;
  const bool neg = sgn != 0u;
  // pseudo-subnormals: clear the dshift = ENORM - expfield low mantissa bits (no re-rounding there)
  int d = ENORM - (int)(a >> 10);
  d = d < 0 ? 0 : d;
  const uint32_t m = a & (0xFFFFFFFFu << d);
  uint32_t r;
  if constexpr (RM == STOCHASTIC) {
    r = m + (rnd & ((1u << K) - 1u));  // every non-flushed value is an fp16 normal (cpp:1106)
    if constexpr (SUB_RNE) {
      const uint32_t ne = m + (round_inc<RNE, K>(m, neg, a != 0u) << K);
      r = ((int)(a >> 10) < ENORM) ? ne : r;
    }
  } else {
    r = m + (round_inc<RM, K>(m, neg, a != 0u) << K);
  }
  r &= ~((1u << K) - 1u);
  r = (a < FLUSH) ? 0u : r;   // below the pseudo-subnormal range (and all fp16 denormals)
  r = (a >= SAT) ? SAT : r;   // >= max, Inf and NaN saturate
  return r | sgn;
}

template <int FMT, int RM>
__device__ __forceinline__ uint32_t quant_h(uint32_t h, uint32_t rnd) {
  if constexpr (FMT == E5M2) return quant_e5m2<RM, false>(h, rnd);
  else if constexpr (FMT == E5M2_DAZ) return quant_e5m2<RM, true>(h, rnd);
  else if constexpr (FMT == E4M3) return quant_e4m3_e3m4<RM, 7, 0x5F00u, 0x1800u, 9>(h, rnd);
  else if constexpr (FMT == E3M4) return quant_e4m3_e3m4<RM, 6, 0x4F80u, 0x2400u, 13>(h, rnd);
  else if constexpr (FMT == E4M3_IEEE) return quant_e4m3_e3m4<RM, 7, 0x5B80u, 0x1800u, 9, true>(h, rnd);
  else return h;
}

// BFLOAT16 on the fp32 pattern.  rnd: >= 16 random bits.
template <int RM>
__device__ __forceinline__ float cast_bf16(float x, uint32_t rnd) {
  uint32_t u = f2u(x);
  const bool naninf = (u & 0x7F800000u) == 0x7F800000u;
  uint32_t r = u;
  if constexpr (RM == RNE) r = u + 0x7FFFu + ((u >> 16) & 1u);
  else if constexpr (RM == STOCHASTIC) r = u + (rnd & 0xFFFFu);
  r = naninf ? u : r;
  return u2f(r & 0xFFFF0000u);
}

// FLOAT16 (cpp:356-423).  rnd: >= 13 random bits.
template <int RM, bool DAZ>
__device__ __forceinline__ float cast_fp16(float x, uint32_t rnd) {
  uint32_t h = f32_to_h(x);
  if constexpr (RM == STOCHASTIC) {
    const uint32_t u = f2u(x);
    const uint32_t a = u & 0x7FFFFFFFu;
    // fp16-normal range [2^-14, 2^16): rebias, add 13 noise bits, truncate (cpp:413-417)
    const uint32_t t = ((a - 0x38000000u) + (rnd & 0x1FFFu)) >> 13;
    const uint32_t sgn = (u >> 16) & 0x8000u;
    uint32_t hs = sgn | t;
    hs = (a >= 0x47800000u) ? (sgn | 0x7C00u) : hs;  // exp >= 16: Inf (cpp:395-401)
    // NaN: keep sign, quiet bit, top payload bits (cpp:386-394)
    hs = (a > 0x7F800000u) ? (sgn | 0x7E00u | ((a >> 13) & 0x3FFu)) : hs;
    h = (a < 0x38800000u) ? h : hs;  // below 2^-14: the RNE convert (cpp:408-411)
  } else if constexpr (DAZ) {
    h = ((h & 0x7C00u) == 0u) ? 0u : h;  // cpp:374-377
  }
  return h_to_f32(h);
}

// FP4_NEAREST on the fp32 pattern (cpp:1797-1824): outputs are the even powers of two 2^0 .. 2^-12 and 0.
__device__ __forceinline__ float cast_fp4(float x, float scale, float rcp) {
  float p = x * scale;
  asm volatile("" : "+v"(p));
  uint32_t u = f2u(p);
  const uint32_t sgn = u & 0x80000000u;
  const int e0 = (int)((u >> 23) & 0xFFu) - 127;
  const bool odd0 = (e0 < 0) && (e0 & 1);
  // "see if round up works": x * 1.6 as a double product rounded back to fp32 (cpp:1811)
  const float up = (float)((double)p * 1.6);
  u = odd0 ? f2u(up) : u;
  u = (e0 > 0) ? (sgn | 0x3F800000u) : u;   // exponent > 0, Inf and NaN: +-1.0
  u &= 0xFF800000u;
  const int e1 = (int)((u >> 23) & 0xFFu) - 127;
  u = ((e1 < 0) && (e1 & 1)) ? (sgn | ((uint32_t)(e1 + 126) << 23)) : u;  // still odd: one exponent down
  u = (e1 < -12) ? 0u : u;                  // below 2^-12: +0
  float r = u2f(u) * rcp;
  asm volatile("" : "+v"(r));
  return r;
}

// One element, fp32 in -> fp32 out.  scale / rcp are ignored by BF16 and FP16 (the reference's
// BFLOAT16_Kernel / FLOAT16_Kernel take no scale, cpp:1933-1957).
template <int FMT, int RM>
__device__ __forceinline__ float cast_one(float x, float scale, float rcp, uint32_t rnd) {
  if constexpr (FMT == BF16) {
    return cast_bf16<RM>(x, rnd);
  } else if constexpr (FMT == FP16) {
    return cast_fp16<RM, false>(x, rnd);
  } else if constexpr (FMT == FP16_DAZ) {
    return cast_fp16<RM, true>(x, rnd);
  } else if constexpr (FMT == FP4) {
    return cast_fp4(x, scale, rcp);
  } else {
    float p = x * scale;
    // Keep the fp32 product a value of its own: two roundings (fp32, then fp16) are the
    // contract (cpp:718-720); without this hipcc folds mul+cvt into v_fma_mixlo_f16.
    asm volatile("" : "+v"(p));
    const uint32_t q = quant_h<FMT, RM>(f32_to_h(p), rnd);
    float r = h_to_f32(q) * rcp;
    // Same on the way out: with 16-bit storage hipcc would fold cvt+mul+cvt into
    // v_fma_mixlo_f16 d, rcp, q, 0 whose +0 addend turns a -0 product into +0.
    asm volatile("" : "+v"(r));
    return r;
  }
}

// ---------------------------------------------------------------------------------------------
// Packed form: TWO fp16 patterns in one VGPR, quantised by gfx950's packed 16-bit VALU ops (v_pk_add_u16,
// v_pk_lshlrev_b16, v_pk_min_u16, ...), so the integer part of the cast costs one instruction per PAIR of
// elements.  There is no packed compare / select: the selects of the scalar form become 0/1 gates
//     ge(a, c) = min(a -sat (c - 1), 1)        1 iff a >= c      (v_pk_sub_u16 clamp, v_pk_min_u16)
// multiplied in (v_pk_mul_lo_u16), caps (v_pk_min_u16 against a saturating difference) or masks; the
// saturation is an unsigned min (rounding is monotone and the largest value is representable, so
// round(a) >= SAT exactly when a >= SAT).  Same results as quant_h for
// every pattern (tests: all 65 536 fp16 patterns per mode through both forms).  Defined for the three FP8
// formats under RTZ / RNE / STOCHASTIC -- the modes of BASELINE config 2; everything else runs quant_h per half.
// ---------------------------------------------------------------------------------------------
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef int16_t i16x2 __attribute__((ext_vector_type(2)));

template <int FMT, int RM>
__host__ __device__ constexpr bool has_packed_quant() {
  return (FMT == E5M2 || FMT == E4M3 || FMT == E3M4) && (RM == RTZ || RM == RNE || RM == STOCHASTIC);
}

__device__ __forceinline__ u16x2 splat16(uint32_t c) { return u16x2{(uint16_t)c, (uint16_t)c}; }
// 1 iff a >= c (c >= 1), per half.  The min is an asm statement so that the gate stays a value of its own: left
// visible, LLVM rewrites min(x, 1) * y into per-half compares and selects (v_cmp_u16 + v_cndmask + v_perm: five or
// six instructions where the packed min and multiply are two).
__device__ __forceinline__ u16x2 ge_gate(u16x2 a, uint32_t c) {
  const uint32_t t = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(a, splat16(c - 1u)));
  uint32_t g;
  asm volatile("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(g) : "v"(t));
  return __builtin_bit_cast(u16x2, g);
}

// rnd: per half, >= 8 random bits in the low byte (ignored unless STOCHASTIC)
template <int RM>
__device__ __forceinline__ u16x2 quant2_e5m2(u16x2 h, u16x2 rnd) {
  if constexpr (RM == RTZ) {
    return h & splat16(0xFF00u);  // Inf / NaN: h & 0xFF00 as well
  } else {
    const u16x2 a = h & splat16(0x7FFFu);
    const u16x2 rne = ((h >> splat16(8u)) & splat16(1u)) + splat16(0x7Fu);  // + 0x7F + kept LSB, then truncate
    u16x2 inc = rne;
    if constexpr (RM == STOCHASTIC) {
      const u16x2 nrm = ge_gate(a, 0x0400u);  // fp16 normals take the noise, fp16 denormals RNE (cpp:739-742)
      inc = rne + nrm * ((rnd & splat16(0xFFu)) - rne);
    }
    // Inf / NaN take no increment, finite values the whole of it: inc is capped at 0x7C00 - a (saturating, so 0
    // from Inf up).  A cap that bites means a + inc would pass 0x7C00 by less than 0x100: the truncated result
    // is 0x7C00 either way.
    inc = __builtin_elementwise_min(inc, __builtin_elementwise_sub_sat(splat16(0x7C00u), a));
    return (h + inc) & splat16(0xFF00u);
  }
}

template <int RM, int K, uint32_t SAT, uint32_t FLUSH, int ENORM>
__device__ __forceinline__ u16x2 quant2_e4m3_e3m4(u16x2 h, u16x2 rnd) {
  constexpr uint32_t LOWM = (1u << K) - 1u;
  constexpr int DMAX = ENORM - (int)(FLUSH >> 10);  // largest pseudo-subnormal shift (3 / 4)
  static_assert((FLUSH & 0x3FFu) == 0u, "the flush bound is a power of two");
  const u16x2 a = h & splat16(0x7FFFu);
  // pseudo-subnormals: clear the d = ENORM - expfield low mantissa bits; below the flush bound (d > DMAX)
  // clear everything -- and a cleared magnitude rounds to 0 in every mode.  mask = -((1 << d) & (2^(DMAX+1) - 1)):
  // 0xFFFF << d for d <= DMAX, 0 above.
  const i16x2 e = __builtin_bit_cast(i16x2, a >> splat16(10u));
  const i16x2 d = __builtin_elementwise_max(i16x2{(int16_t)ENORM, (int16_t)ENORM} - e, i16x2{0, 0});
  const u16x2 bit = (splat16(1u) << __builtin_bit_cast(u16x2, d)) & splat16((2u << DMAX) - 1u);
  const u16x2 m = a & (splat16(0u) - bit);
  u16x2 r = m;
  if constexpr (RM == RNE) r = m + ((m >> splat16((uint32_t)K)) & splat16(1u)) + splat16((1u << (K - 1)) - 1u);
  else if constexpr (RM == STOCHASTIC) r = m + (rnd & splat16(LOWM));
  r &= splat16(~LOWM);
  r = __builtin_elementwise_min(r, splat16(SAT));           // >= max, Inf and NaN saturate
  return r | (h & splat16(0x8000u));
}

template <int FMT, int RM>
__device__ __forceinline__ u16x2 quant_h2(u16x2 h, u16x2 rnd) {
  if constexpr (FMT == E5M2) return quant2_e5m2<RM>(h, rnd);
  else if constexpr (FMT == E4M3) return quant2_e4m3_e3m4<RM, 7, 0x5F00u, 0x1800u, 9>(h, rnd);
  else return quant2_e4m3_e3m4<RM, 6, 0x4F80u, 0x2400u, 13>(h, rnd);
}

// f32( f16 half of q ) * rcp as ONE instruction: v_fma_mix_f32 reads the half straight out of the packed
// register (op_sel) and rounds the product once to fp32 -- the same value as v_cvt_f32_f16 (exact) followed by
// v_mul_f32.  The addend is -0.0 (held in an SGPR), which leaves the product's sign alone (x*y + -0 == x*y for
// every x*y, including both zeros); a +0 addend would turn a -0 product into +0.
template <int HI>
__device__ __forceinline__ float mix_mul(uint32_t q, float rcp) {
  float r;
  const float nz = -0.0f;
  if constexpr (HI) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(q), "v"(rcp), "s"(nz));
  else asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(q), "v"(rcp), "s"(nz));
  return r;
}

// Two elements, fp32 in -> fp32 out; rnd2: the two elements' noise, one per 16-bit half.
// Only for has_packed_quant<FMT, RM>().
template <int FMT, int RM>
__device__ __forceinline__ void cast_pair(float x0, float x1, float scale, float rcp, uint32_t rnd2, float &y0,
                                          float &y1) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
  float p0 = x0 * scale, p1 = x1 * scale;
  asm volatile("" : "+v"(p0), "+v"(p1));  // the fp32 products are values of their own (see cast_one)
  const h16x2 hh = __builtin_convertvector(f32x2{p0, p1}, h16x2);  // v_cvt_pk_f16_f32: RNE, denormals kept
  const u16x2 q = quant_h2<FMT, RM>(__builtin_bit_cast(u16x2, hh), __builtin_bit_cast(u16x2, rnd2));
  const uint32_t qq = __builtin_bit_cast(uint32_t, q);
  y0 = mix_mul<0>(qq, rcp);
  y1 = mix_mul<1>(qq, rcp);
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10, counter-based: replaces the shared mutable xoshiro state of the reference
// (cpp:41-88, hip/fpemu_kernels.hip:47-102), which races.  key = seed, counter = (ctr, offset).
// ---------------------------------------------------------------------------------------------
struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(uint64_t seed, uint64_t ctr, uint64_t offset) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32);
  uint32_t c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
#pragma unroll
  for (int r = 0; r < 10; r++) {
    // one 32x32->64 multiply per product (v_mad_u64_u32) instead of a mul_hi / mul_lo pair
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
    const uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += W0; k1 += W1;
  }
  return Philox4{c0, c1, c2, c3};
}

__device__ __forceinline__ uint32_t philox_word(const Philox4 &p, unsigned i) {
  return i == 0 ? p.x : i == 1 ? p.y : i == 2 ? p.z : p.w;
}

template <int FMT>
__host__ __device__ constexpr int noise_bytes() {
  return (FMT == BF16 || FMT == FP16 || FMT == FP16_DAZ) ? 2 : 1;
}

// Random bits of one element from its global index (layout in DESIGN.md "Stochastic rounding");
// the slow general form, used by tails and by unaligned buffers.  V = elements per 16-byte vector.
template <int FMT, unsigned V>
__device__ __forceinline__ uint32_t rand_bits_elem(uint64_t seed, uint64_t offset, uint64_t idx) {
  constexpr unsigned NB = noise_bytes<FMT>();
  constexpr uint64_t SB = (uint64_t)V * 64u * 4u;
  const uint64_t sb = idx / SB;
  const unsigned r = (unsigned)(idx % SB);
  const unsigned slot = r / (V * 64u);
  const unsigned lane = (r / V) % 64u;
  const unsigned e = r % V;
  const unsigned pos = (slot * V + e) * NB;
  const uint64_t ctr = ((sb * 64u + lane) << 2) | (pos / 16u);
  const Philox4 p = philox4x32_10(seed, ctr, offset);
  const uint32_t w = philox_word(p, (pos % 16u) / 4u) >> ((pos % 4u) * 8u);
  return NB == 2 ? (w & 0xFFFFu) : (w & 0xFFu);
}

// Per-tensor "max" scale on device, exactly as quantize_tensor (qutils.py:284-285):
// flt_max / vmax with a Python float on the left is vmax.reciprocal() * flt_max (two fp32
// roundings, IEEE division), then the 3.275e4 -> 6.55e4 clamp.
__device__ __forceinline__ float scale_from_amax(uint32_t amax_bits, float flt_max, bool clamp) {
  const float vmax = u2f(amax_bits);
  float r = 1.0f / vmax;  // correctly rounded (v_div_scale/fmas/fixup); fast-math is off
  asm volatile("" : "+v"(r));
  float s = r * flt_max;
  if (clamp && s > 3.275e4f) s = 6.55e4f;
  return s;
}

// Block-norm scale from the block's max |x| (cpp:794-799, :1166-1171, :1458-1463, :1751-1755, :1858).
template <int FMT>
__device__ __forceinline__ float blocknorm_scale(uint32_t amax_bits) {
  if constexpr (FMT == FP4) {
    return (float)(1.0 / (double)u2f(amax_bits));
  } else {
    float s = 2.0f * u2f(amax_bits & 0x7F800000u);
    if constexpr (FMT == E5M2) s = s / 16384.0f;
    else if constexpr (FMT == E4M3 || FMT == E4M3_IEEE) s = s / 8.0f;
    return s;
  }
}

// (float)(1.0 / (double)scale), cpp:714 / :1060 / :1646.
__device__ __forceinline__ float scale_reciprocal(float scale) {
  return (float)(1.0 / (double)scale);
}

}  // namespace fpemu
