/*
 * fpemu_oracle.c -- CPU restatement of the reference fpemu cast.  TEST INFRASTRUCTURE ONLY:
 * see fpemu_oracle.h.  Never linked, imported or called by the product path.
 *
 * Reference: /root/reference/mpemu/pytquant/cpp/fpemu_impl.cpp (line numbers cited inline).
 * Where the reference's AVX-512 and scalar routines disagree, the routine named in
 * SURVEY.md section 8(a) as canonical is the one restated (scalar for E5M2/E4M3/E3M4 and
 * every stochastic mode; the AVX-512 routine for BFLOAT16_RNE).
 *
 * Pinning: tests/test_oracle_golden.py checks every function here against fixtures produced
 * by the reference's own extension (tests/golden/make_golden.py, oracle/build_ref.py).
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off; no -ffast-math; OpenMP optional).
 */
#include "fpemu_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ------------------------------------------------------------------------------------------
 * fp32 <-> fp16.  The reference uses the F16C instructions (_cvtss_sh with
 * _MM_FROUND_TO_NEAREST_INT, cpp:117-119; _cvtsh_ss, cpp:121-123).  Restated in integer
 * arithmetic so the oracle does not depend on the host ISA; tests/golden pins it against
 * the reference on all fp16 boundaries and random fp32 patterns.
 * ---------------------------------------------------------------------------------------- */
uint16_t orc_f32_to_f16_rne(float f) {
  uint32_t x = f2u(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t a = x & 0x7FFFFFFFu;
  if (a >= 0x7F800000u) {
    if (a > 0x7F800000u) /* NaN: quieted, top 10 payload bits kept (vcvtps2ph behaviour) */
      return (uint16_t)(sign | 0x7E00u | ((a >> 13) & 0x03FFu));
    return (uint16_t)(sign | 0x7C00u);
  }
  if (a >= 0x477FF000u) /* >= 65520: rounds to infinity (65520 is the tie, even = inf) */
    return (uint16_t)(sign | 0x7C00u);
  if (a >= 0x38800000u) { /* fp16 normal range: rebias exponent by 112, RNE on 13 bits */
    uint32_t m = a - 0x38000000u;
    uint32_t r = m + 0x0FFFu + ((m >> 13) & 1u);
    return (uint16_t)(sign | (r >> 13));
  }
  uint32_t e = a >> 23;
  if (e < 102u) /* < 2^-25 (and 2^-25 itself is handled below as a tie to even = 0) */
    return (uint16_t)sign;
  uint32_t mant = (a & 0x007FFFFFu) | 0x00800000u;
  uint32_t shift = 126u - e; /* 14..24: result in units of 2^-24 */
  uint32_t q = mant >> shift;
  uint32_t rem = mant & ((1u << shift) - 1u);
  uint32_t half = 1u << (shift - 1u);
  if (rem > half || (rem == half && (q & 1u))) q++;
  return (uint16_t)(sign | q);
}

float orc_f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1Fu;
  uint32_t m = h & 0x03FFu;
  if (e == 0x1Fu) { /* a conversion instruction quiets a signalling NaN (vcvtph2ps, v_cvt_f32_f16) */
    if (m) m |= 0x0200u;
    return u2f(sign | 0x7F800000u | (m << 13));
  }
  if (e != 0) return u2f(sign | ((e + 112u) << 23) | (m << 13));
  if (m == 0) return u2f(sign);
  /* subnormal: normalise */
  uint32_t ee = 113u;
  while (!(m & 0x0400u)) { m <<= 1; ee--; }
  return u2f(sign | (ee << 23) | ((m & 0x03FFu) << 13));
}

/* ------------------------------------------------------------------------------------------
 * Rounding increment at bit k of `field` (the reference's masks, e.g. cpp:729-754 for E5M2:
 * grs_bitmask = 2^k-1, rne_tie = 3<<(k-1)).  `positive` / `negative` restate (h.f > 0) and
 * (h.f < 0) of cpp:752-754.  Note RPINF/RNINF are nearest-style in the reference, not
 * directed roundings.
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t round_inc(uint32_t field, int k, int rmode, int positive, int negative) {
  uint32_t low = field & ((1u << k) - 1u);
  uint32_t half = 1u << (k - 1);
  uint32_t tie = 3u << (k - 1);
  switch (rmode) {
    case ORC_RNE:   return (low > half) || ((field & tie) == tie);
    case ORC_RNAZ:  return low >= half;
    case ORC_RNTZ:  return low > half;
    case ORC_RPINF: return positive && (low >= half);
    case ORC_RNINF: return negative && (low >= half);
    default:        return 0; /* RTZ, and STOCHASTIC (handled by the caller) */
  }
}

/* E5M2: cvt_fp32_e5m2_scalar cpp:716-762; daz=1: E5M2_DAZ_Kernel cpp:874-918. */
static uint16_t q_e5m2(uint16_t h, int rmode, uint32_t rnd, int daz) {
  const int k = 8; /* lshift, cpp:694 */
  uint32_t u = h;
  uint32_t ef = u & 0x7C00u;
  int is_naninf = (ef == 0x7C00u);
  int is_normal = (ef <= 0x7800u) && (ef >= 0x0400u);
  int is_denorm = (ef == 0);
  int can_round = ((u & 0x7F00u) <= 0x7B00u);
  int positive = !(u & 0x8000u) && (u & 0x7FFFu) != 0 && !is_naninf;
  int negative = (u & 0x8000u) && (u & 0x7FFFu) != 0 && !is_naninf;
  if (daz) {
    if (!is_naninf && is_normal) {
      if (rmode == ORC_STOCHASTIC) u += can_round * (rnd & 0xFFu);                     /* cpp:897 */
      else u += (uint32_t)can_round * (round_inc(u, k, rmode, positive, negative) << k); /* cpp:900-909 */
    } else if (is_denorm) {
      u = 0; /* cpp:913: sign is dropped too */
    }
    return (uint16_t)(u & 0xFF00u);
  }
  if (!is_naninf) {
    if (rmode == ORC_STOCHASTIC) {
      /* cpp:739-742: noise on fp16 normals, RNE on fp16 denormals */
      if (is_normal) u += can_round * (rnd & 0xFFu);
      if (is_denorm) u += (uint32_t)can_round * (round_inc(u, k, ORC_RNE, 0, 0) << k);
    } else {
      u += (uint32_t)can_round * (round_inc(u, k, rmode, positive, negative) << k); /* cpp:745-754 */
    }
  }
  return (uint16_t)(u & 0xFF00u); /* cpp:758 */
}

/* E4M3 (cpp:1063-1132) and E3M4 (cpp:1649-1717) share one shape; constants per format:
 *            k  SAT     EMAX  SATM    EFLUSH  EMIN
 *   E4M3     7  0x5F00   8    0x0300   -9      -6
 *   E3M4     6  0x4F80   4    0x0380   -6      -2                                        */
static uint16_t q_e4m3_e3m4(uint16_t h, int rmode, uint32_t rnd, int k, uint32_t sat, int emax,
                            uint32_t satm, int eflush, int emin) {
  uint32_t u = h;
  int exp_h = (int)((u & 0x7C00u) >> 10) - 15;
  uint32_t sign_h = u & 0x8000u;
  uint32_t mant = u & 0x03FFu;
  uint32_t ef = u & 0x7C00u;
  int can_round = ((u & 0x7FFFu) < sat);
  int is_normal = (ef <= 0x7800u) && (ef >= 0x0400u);
  int is_denorm = (ef == 0);
  int is_naninf = (ef == 0x7C00u);
  int positive = !sign_h && (u & 0x7FFFu) != 0 && !is_naninf;
  int negative = sign_h && (u & 0x7FFFu) != 0 && !is_naninf;

  if (exp_h > emax || !can_round) { /* saturate: >= max, Inf and NaN all become +-max */
    mant = satm;
    exp_h = emax;
    can_round = 0;
  } else if (exp_h < eflush) { /* flush below the pseudo-subnormal range, sign kept */
    exp_h = -15;
    mant = 0;
  } else if (exp_h < emin) { /* pseudo-subnormal: drop dshift low bits, no re-rounding there */
    int dshift = emin - exp_h;
    mant = (mant >> dshift) << dshift;
  }
  if (!is_naninf) {
    if (rmode == ORC_STOCHASTIC) {
      if (is_normal) mant += can_round * (rnd & ((1u << k) - 1u));                /* cpp:1106 */
      if (is_denorm) mant += (uint32_t)can_round * (round_inc(mant, k, ORC_RNE, 0, 0) << k); /* cpp:1108 */
    } else {
      mant += (uint32_t)can_round * (round_inc(mant, k, rmode, positive, negative) << k);
    }
  }
  mant &= ~((1u << k) - 1u);                  /* cpp:1124 */
  mant += (uint32_t)(exp_h + 15) << 10;       /* '+': a mantissa carry bumps the exponent */
  mant |= sign_h;
  return (uint16_t)mant;
}

/* E4M3_IEEE: cvt_fp32_e4m3_ieee_rne_intrinsic cpp:1213-1265 and ..._stochastic_intrinsic cpp:1267-1328,
 * the AVX-512 routines, which are canonical here: the scalar routine tests can_round against 0x4B80
 * (15.0) instead of 0x5B80 (cpp:1364) and so saturates everything >= 15 to 240 -- a reference bug the
 * bulk (size/32*32) of every tensor never sees on an AVX-512 host.  Same shape as E4M3 with max 240
 * (0x5B80); NaN and Inf saturate; under STOCHASTIC the pseudo-subnormal range (< 2^-6) rounds RNE. */
static uint16_t q_e4m3_ieee(uint16_t h, int rmode, uint32_t rnd) {
  uint32_t a = h;
  uint32_t ef = a & 0x7C00u;
  int not_naninf = (ef != 0x7C00u);                 /* maska1 */
  uint32_t fix = (a & 0x0080u) ? 1u : 0u;           /* maska2: kept LSB, makes 0x3f+1 the tie-to-even add */
  int sat = ((a & 0x7FFFu) >= 0x5B80u);             /* maska3 */
  int flush = (ef < 0x1800u);                       /* maska4 */
  int sub = (ef < 0x2400u);                         /* maska5 */
  if (sub) {
    unsigned sh = 9u - (ef >> 10);                  /* cpp:1243-1248 */
    a = ((a >> sh) << sh) & 0xFFFFu;
  }
  uint32_t r = a;
  if (rmode == ORC_STOCHASTIC) {
    if (not_naninf && !sub) r = a + (rnd & 0x7Fu);  /* cpp:1312-1313: 7 random bits */
    if (sub) r = r + 0x3Fu + fix;                   /* cpp:1314-1315 */
  } else if (rmode == ORC_RNE) {
    if (not_naninf) r = a + 0x3Fu + fix;            /* cpp:1250-1251 */
  }
  r &= 0xFFFFu;
  if (sat) r = (r & 0x8000u) | 0x5B80u;             /* cpp:1253 */
  if (flush) r = r & 0x8000u;                       /* cpp:1254 */
  return (uint16_t)(r & 0xFF80u);                   /* cpp:1256 */
}

/* FP4_NEAREST: cvt_fp32_fp4_nearest_scalar cpp:1797-1824.  Works on the fp32 pattern: values are the
 * even powers of two 2^0, 2^-2 .. 2^-12 (and 0); odd negative exponents try x*1.6 (double multiply,
 * rounded back to float) and otherwise drop one exponent; anything with exponent > 0 (Inf / NaN too)
 * becomes +-1. */
static float fp4_cast(float x, float scale, float rcp) {
  volatile float p = scale * x;                     /* cpp:1805 */
  float f = p;
  uint32_t u = f2u(f);
  int exp_f = (int)((u & 0x7F800000u) >> 23) - 127;
  uint32_t sign_f = u & 0x80000000u;
  if (exp_f < 0 && (exp_f % 2)) {                   /* cpp:1811 */
    f = (float)((double)f * 1.6);
    u = f2u(f);
  }
  if (exp_f > 0) u = sign_f | (127u << 23);         /* cpp:1813 */
  u &= 0xFF800000u;
  exp_f = (int)((u & 0x7F800000u) >> 23) - 127;
  if (exp_f < 0 && (exp_f % 2)) u = sign_f | ((uint32_t)(exp_f + 126) << 23); /* cpp:1818 */
  if (exp_f < -12) u = 0;                           /* cpp:1821: +0, sign dropped */
  return u2f(u) * rcp;
}

uint16_t orc_quantize_h(uint16_t h, int fmt, int rmode, uint32_t rnd) {
  switch (fmt) {
    case ORC_FMT_E4M3_IEEE: return q_e4m3_ieee(h, rmode, rnd);
    case ORC_FMT_E5M2:     return q_e5m2(h, rmode, rnd, 0);
    case ORC_FMT_E5M2_DAZ: return q_e5m2(h, rmode, rnd, 1);
    case ORC_FMT_E4M3:     return q_e4m3_e3m4(h, rmode, rnd, 7, 0x5F00u, 8, 0x0300u, -9, -6);
    case ORC_FMT_E3M4:     return q_e4m3_e3m4(h, rmode, rnd, 6, 0x4F80u, 4, 0x0380u, -6, -2);
    default:               return h;
  }
}

/* BFLOAT16.  RNE: cvt_fp32_bf16_rne_intrinsic cpp:207-232 (canonical, == torch .to(bfloat16)
 * for every non-NaN).  STOCHASTIC: cvt_fp32_bf16_scalar cpp:302-318 (u += rand16, truncate)
 * with the reference GPU kernel's NaN/Inf guard (hip/fpemu_kernels.hip:291).  RTZ: truncate
 * (extension; the reference defines no BFLOAT16_RTZ -- "parity unpinned"). */
static float bf16_cast(float x, int rmode, uint32_t rnd) {
  uint32_t u = f2u(x);
  int naninf = ((u & 0x7F800000u) == 0x7F800000u);
  if (!naninf) {
    if (rmode == ORC_RNE) u += 0x7FFFu + ((u >> 16) & 1u);
    else if (rmode == ORC_STOCHASTIC) u += (rnd & 0xFFFFu);
  }
  return u2f(u & 0xFFFF0000u);
}

/* FLOAT16: FLOAT16_Kernel cpp:356-423.  RNE (+ optional DAZ, cpp:370-378); STOCHASTIC is the
 * 13-bit add on rebiased fp32 fields (cpp:379-420). */
static float fp16_cast(float x, int rmode, int no_denorm, uint32_t rnd) {
  uint16_t h;
  if (rmode == ORC_RNE) {
    h = orc_f32_to_f16_rne(x);
    int is_denorm = (((h & 0x7FFFu) >> 10) & 0x1Fu) == 0;
    if (is_denorm && no_denorm) h = 0; /* h.u *= !(is_denorm && no_denorm): sign dropped */
    return orc_f16_to_f32(h);
  }
  if (rmode == ORC_STOCHASTIC) {
    uint32_t fval = f2u(x);
    int exp_h = (int)((fval & 0x7F800000u) >> 23) - 127;
    uint32_t mantissa_h = fval & 0x007FFFFFu;
    uint32_t sign_h = fval & 0x80000000u;
    if (exp_h == 128) { /* Inf / NaN, cpp:386-394 */
      if (mantissa_h && ((mantissa_h & 0x400000u) == 0)) mantissa_h |= 0x400000u;
      mantissa_h |= (0x1Fu << 23);
      mantissa_h |= (sign_h >> 3);
      h = (uint16_t)(mantissa_h >> 13);
    } else if (exp_h >= 16) { /* saturate to Inf, cpp:395-401 */
      mantissa_h = (0x1Fu << 23) | (sign_h >> 3);
      h = (uint16_t)(mantissa_h >> 13);
    } else if (exp_h < -14) { /* fp16 denormal range: plain RNE convert, cpp:408-411 */
      h = orc_f32_to_f16_rne(x);
    } else { /* cpp:413-417 */
      mantissa_h |= ((uint32_t)(exp_h + 15) << 23);
      mantissa_h |= (sign_h >> 3);
      mantissa_h += (rnd & 0x1FFFu);
      h = (uint16_t)(mantissa_h >> 13);
    }
    return orc_f16_to_f32(h);
  }
  /* RTZ is not defined by the reference for FLOAT16; unreachable through orc_parse_mode. */
  return x;
}

float orc_scale_reciprocal(float scale) { return (float)(1.0 / (double)scale); }

float orc_cast_one(float x, int fmt, int rmode, float scale, float rcp, uint32_t rnd) {
  if (fmt == ORC_FMT_BF16) return bf16_cast(x, rmode, rnd);               /* no scale: cpp:1948-1957 */
  if (fmt == ORC_FMT_FP16) return fp16_cast(x, rmode, 0, rnd);
  if (fmt == ORC_FMT_FP16_DAZ) return fp16_cast(x, rmode, 1, rnd);
  if (fmt == ORC_FMT_FP4) return fp4_cast(x, scale, rcp);
  float p = scale * x;                                                    /* cpp:718 */
  uint16_t h = orc_f32_to_f16_rne(p);                                     /* cpp:720 */
  uint16_t q = orc_quantize_h(h, fmt, rmode, rnd);
  return orc_f16_to_f32(q) * rcp;                                         /* cpp:761 */
}

/* ------------------------------------------------------------------------------------------
 * Counter-based RNG (replaces the racy shared xoshiro state of cpp:41-88 / hip:47-102).
 * ---------------------------------------------------------------------------------------- */
void orc_philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                       uint32_t c3, uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static int noise_bytes(int fmt) {
  return (fmt == ORC_FMT_BF16 || fmt == ORC_FMT_FP16 || fmt == ORC_FMT_FP16_DAZ) ? 2 : 1;
}

/* Random word of element idx when the buffer is processed in 16-byte vectors of V elements
 * (V = 4 for fp32 storage, 8 for 16-bit storage).  A wavefront iteration covers a
 * "superblock" of S=4 slots x 64 lanes x V elements; lane L owns, per superblock, the
 * S*V elements {slot*64*V + L*V + e}.  Its noise is a byte stream of S*V*nb bytes cut from
 * consecutive Philox calls:  pos = (slot*V + e)*nb;  call = pos/16;
 * counter = (((superblock*64 + L) << 2) | call, offset);  key = seed. */
static uint32_t rand_bits_v(uint64_t seed, uint64_t offset, uint64_t idx, int fmt, unsigned V) {
  unsigned nb = (unsigned)noise_bytes(fmt);
  uint64_t sbsz = (uint64_t)V * 64u * 4u;
  uint64_t sb = idx / sbsz;
  unsigned r = (unsigned)(idx % sbsz);
  unsigned slot = r / (V * 64u);
  unsigned lane = (r / V) % 64u;
  unsigned e = r % V;
  unsigned pos = (slot * V + e) * nb;
  unsigned call = pos / 16u;
  uint64_t ctr = ((sb * 64u + lane) << 2) | call;
  uint32_t o[4];
  orc_philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32),
                    (uint32_t)offset, (uint32_t)(offset >> 32), o);
  uint32_t w = o[(pos % 16u) / 4u] >> ((pos % 4u) * 8u);
  return nb == 2 ? (w & 0xFFFFu) : (w & 0xFFu);
}

uint32_t orc_rand_bits(uint64_t seed, uint64_t offset, uint64_t idx, int fmt) {
  return rand_bits_v(seed, offset, idx, fmt, 4u);
}

static int valid_combo(int fmt, int rmode) {
  if (rmode < ORC_RTZ || rmode > ORC_RNINF) return 0;
  switch (fmt) {
    case ORC_FMT_E5M2: case ORC_FMT_E4M3: case ORC_FMT_E3M4: case ORC_FMT_E5M2_DAZ: return 1;
    case ORC_FMT_BF16: return rmode == ORC_RNE || rmode == ORC_STOCHASTIC || rmode == ORC_RTZ;
    case ORC_FMT_FP16: return rmode == ORC_RNE || rmode == ORC_STOCHASTIC;
    case ORC_FMT_FP16_DAZ: return rmode == ORC_RNE;
    case ORC_FMT_E4M3_IEEE: return rmode == ORC_RNE || rmode == ORC_STOCHASTIC;
    case ORC_FMT_FP4: return rmode == ORC_RNE; /* "NEAREST" */
    default: return 0;
  }
}

int orc_cast_f32(const float *in, float *out, uint64_t n, int fmt, int rmode, float scale,
                 uint64_t seed, uint64_t offset, uint64_t elem_base) {
  if (!valid_combo(fmt, rmode)) return -1;
  float rcp = orc_scale_reciprocal(scale);
  int sr = (rmode == ORC_STOCHASTIC);
  int64_t nn = (int64_t)n;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < nn; i++) {
    uint32_t rnd = sr ? rand_bits_v(seed, offset, elem_base + (uint64_t)i, fmt, 4u) : 0u;
    out[i] = orc_cast_one(in[i], fmt, rmode, scale, rcp, rnd);
  }
  return 0;
}

/* Block normalisation (E5M2_Kernel cpp:778-815, E4M3_Kernel cpp:1149-1190, E4M3_IEEE_Kernel
 * cpp:1441-1478, E3M4_Kernel cpp:1734-1770, FP4_Nearest_Kernel cpp:1840-1860): per block of
 * block_size elements, maxval = max |x| by a compare that never takes a NaN, p2 = maxval with the
 * mantissa cleared, scale = 2*p2/16384 (E5M2), 2*p2/8 (E4M3, E4M3_IEEE), 2*p2 (E3M4), 1/maxval (FP4),
 * then the ordinary cast of the block with that scale.  The caller guarantees n % block_size == 0
 * (fpemu_forward demotes block_norm otherwise, cpp:2020-2025).  Formats without a block-norm branch
 * in the reference (E5M2_DAZ, BFLOAT16, FLOAT16) are not accepted here.
 * Stochastic noise: block b draws as call offset + b, element index = position in the block. */
float orc_blocknorm_scale(float maxval, int fmt) {
  float p2 = u2f(f2u(maxval) & 0x7F800000u);
  volatile float s = 2.0f * p2;
  switch (fmt) {
    case ORC_FMT_E5M2: s = s / 16384.0f; break;
    case ORC_FMT_E4M3: case ORC_FMT_E4M3_IEEE: s = s / 8.0f; break;
    case ORC_FMT_E3M4: break;
    case ORC_FMT_FP4: s = (float)(1.0 / (double)maxval); break;
    default: break;
  }
  return s;
}

int orc_cast_blocknorm_f32(const float *in, float *out, uint64_t n, int fmt, int rmode,
                           uint64_t block_size, uint64_t seed, uint64_t offset) {
  if (!valid_combo(fmt, rmode)) return -1;
  if (fmt != ORC_FMT_E5M2 && fmt != ORC_FMT_E4M3 && fmt != ORC_FMT_E4M3_IEEE && fmt != ORC_FMT_E3M4 &&
      fmt != ORC_FMT_FP4)
    return -1;
  if (block_size == 0 || n % block_size) return -1;
  int sr = (rmode == ORC_STOCHASTIC);
  int64_t nb = (int64_t)(n / block_size);
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < nb; b++) {
    const float *bi = in + (uint64_t)b * block_size;
    float *bo = out + (uint64_t)b * block_size;
    float maxval = 0.0f;
    for (uint64_t i = 0; i < block_size; i++) {
      float a = fabsf(bi[i]);
      maxval = (maxval < a) ? a : maxval;
    }
    float scale = orc_blocknorm_scale(maxval, fmt);
    float rcp = orc_scale_reciprocal(scale);
    for (uint64_t i = 0; i < block_size; i++) {
      uint32_t rnd = sr ? rand_bits_v(seed, offset + (uint64_t)b, i, fmt, 4u) : 0u;
      bo[i] = orc_cast_one(bi[i], fmt, rmode, scale, rcp, rnd);
    }
  }
  return 0;
}

/* bf16 storage helpers: up-convert exact; down-convert RNE with NaN kept NaN. */
static inline float bf16_to_f32(uint16_t b) { return u2f((uint32_t)b << 16); }
static inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = f2u(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

int orc_cast_16(const uint16_t *in, uint16_t *out, uint64_t n, int is_bf16, int fmt, int rmode,
                float scale, uint64_t seed, uint64_t offset, uint64_t elem_base) {
  if (!valid_combo(fmt, rmode)) return -1;
  float rcp = orc_scale_reciprocal(scale);
  int sr = (rmode == ORC_STOCHASTIC);
  int64_t nn = (int64_t)n;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < nn; i++) {
    uint32_t rnd = sr ? rand_bits_v(seed, offset, elem_base + (uint64_t)i, fmt, 8u) : 0u;
    float x = is_bf16 ? bf16_to_f32(in[i]) : orc_f16_to_f32(in[i]);
    float y = orc_cast_one(x, fmt, rmode, scale, rcp, rnd);
    out[i] = is_bf16 ? f32_to_bf16_rne(y) : orc_f32_to_f16_rne(y);
  }
  return 0;
}

/* torch.max(abs(flatten(t))) (qutils.py:283): NaN propagates.  On non-negative floats the
 * unsigned order of the bit patterns is the float order and every NaN pattern sorts above
 * +Inf, so an unsigned max over (bits & 0x7FFFFFFF) restates it. */
float orc_amax_f32(const float *in, uint64_t n) {
  uint32_t m = 0;
  int64_t nn = (int64_t)n;
#pragma omp parallel for reduction(max : m) schedule(static)
  for (int64_t i = 0; i < nn; i++) {
    uint32_t a = f2u(in[i]) & 0x7FFFFFFFu;
    if (a > m) m = a;
  }
  if (m > 0x7F800000u) m = 0x7FC00000u;
  return u2f(m);
}

/* qutils.py:284-285.  `flt_max / vmax` with a Python float on the left and a 0-dim tensor on
 * the right is torch.Tensor.__rtruediv__ = vmax.reciprocal() * flt_max: two fp32 roundings.
 * The comparison promotes to double on the Python side; 3.275e4 is exactly representable so
 * a float compare is equivalent.  6.55e4 then reaches the C++ `float scale` argument as
 * (float)65500.0.  NaN compares false and stays NaN; vmax == 0 gives inf -> 6.55e4. */
float orc_scale_from_amax_noclamp(float vmax, float flt_max) {
  volatile float r = 1.0f / vmax;
  volatile float s = r * flt_max;
  return s;
}

float orc_scale_from_amax(float vmax, float flt_max) {
  float s = orc_scale_from_amax_noclamp(vmax, flt_max);
  if (s > 3.275e4f) s = 6.55e4f;
  return s;
}

void orc_build_lut(int fmt, int rmode, uint16_t *lut) {
  for (uint32_t h = 0; h < 65536u; h++) lut[h] = orc_quantize_h((uint16_t)h, fmt, rmode, 0);
}

int orc_parse_mode(const char *mode, int *fmt, int *rmode) {
  static const struct { const char *name; int fmt, rmode; } tab[] = {
    {"E5M2_RTZ", ORC_FMT_E5M2, ORC_RTZ},               /* cpp:1878 */
    {"E5M2_RNE", ORC_FMT_E5M2, ORC_RNE},               /* cpp:1883 */
    {"E5M2_STOCHASTIC", ORC_FMT_E5M2, ORC_STOCHASTIC}, /* cpp:1888 */
    {"E5M2_RNAZ", ORC_FMT_E5M2, ORC_RNAZ},             /* cpp:1893 */
    {"E5M2_RNTZ", ORC_FMT_E5M2, ORC_RNTZ},             /* cpp:1898 */
    {"E5M2_RPINF", ORC_FMT_E5M2, ORC_RPINF},           /* cpp:1903 */
    {"E5M2_RNINF", ORC_FMT_E5M2, ORC_RNINF},           /* cpp:1908 */
    {"E5M2_DAZ_RNE", ORC_FMT_E5M2_DAZ, ORC_RNE},       /* cpp:1913 */
    {"E5M2_DAZ_STOCHASTIC", ORC_FMT_E5M2_DAZ, ORC_STOCHASTIC},
    {"E5M2_DAZ_RNAZ", ORC_FMT_E5M2_DAZ, ORC_RNAZ},
    {"E5M2_DAZ_RNTZ", ORC_FMT_E5M2_DAZ, ORC_RNTZ},
    {"FLOAT16_RNE", ORC_FMT_FP16, ORC_RNE},            /* cpp:1933 */
    {"FLOAT16_STOCHASTIC", ORC_FMT_FP16, ORC_STOCHASTIC},
    {"FLOAT16_DAZ_RNE", ORC_FMT_FP16_DAZ, ORC_RNE},
    {"BFLOAT16_RNE", ORC_FMT_BF16, ORC_RNE},           /* cpp:1948 */
    {"BFLOAT16_STOCHASTIC", ORC_FMT_BF16, ORC_STOCHASTIC},
    {"E4M3_RNE", ORC_FMT_E4M3, ORC_RNE},               /* cpp:1970 */
    {"E4M3_STOCHASTIC", ORC_FMT_E4M3, ORC_STOCHASTIC},
    {"E3M4_RNE", ORC_FMT_E3M4, ORC_RNE},               /* cpp:1982 */
    {"E3M4_STOCHASTIC", ORC_FMT_E3M4, ORC_STOCHASTIC},
    /* extensions named by BASELINE.json config 2 (not defined by the reference) */
    {"E4M3_RTZ", ORC_FMT_E4M3, ORC_RTZ},
    {"E3M4_RTZ", ORC_FMT_E3M4, ORC_RTZ},
    {"BFLOAT16_RTZ", ORC_FMT_BF16, ORC_RTZ},
    {"E4M3_IEEE_RNE", ORC_FMT_E4M3_IEEE, ORC_RNE},     /* cpp:1958 */
    {"E4M3_IEEE_STOCHASTIC", ORC_FMT_E4M3_IEEE, ORC_STOCHASTIC},
    {"FP4_NEAREST", ORC_FMT_FP4, ORC_RNE},             /* cpp:1994 */
  };
  for (unsigned i = 0; i < sizeof(tab) / sizeof(tab[0]); i++) {
    if (strcmp(mode, tab[i].name) == 0) {
      *fmt = tab[i].fmt;
      *rmode = tab[i].rmode;
      return 0;
    }
  }
  return -1;
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * cmodel emulated GEMM: simple_sgemm_impl (mpemu/cmodel/simple/simple_gemm_impl.cpp:71-237) with
 * MMEngine_avx2_ps (simple_mm_engine.cpp:13-27), called as simple_gemm does (simple_gemm.cpp:45-67,
 * beta = 0).  Per output element: acc = +0; for k ascending the engine's add(acc, mul(a, b)); then
 * C += alpha*acc + beta*C (impl.cpp:226-231).  The reference's zero padding of K to a multiple of 64
 * adds +0 to a chain that is never -0: a no-op.
 * fused == 0: acc = fp32(acc + fp32(a*b)), separate vmulps / vaddps -- the sources compiled for a host
 *             without FMA (oracle/build_ref.py nofma=True);
 * fused != 0: acc = fmaf(a, b, acc) -- the sources compiled with the reference's own flags on an FMA
 *             host (g++ contracts the intrinsics pair into vfmadd231ps).  Both pinned by gemm.npz.
 * Row-major; op(A) = A[M][K] or, a_trans, A[K][M]; op(B) = B[K][N] or, b_trans, B[N][K].
 * ---------------------------------------------------------------------------------------- */
void orc_gemm_f32(float *C, const float *A, const float *B, int64_t M, int64_t N, int64_t K,
                  int64_t lda, int64_t ldb, int64_t ldc, float alpha, int a_trans, int b_trans,
                  int fused) {
  const float beta = 0.0f;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < M; i++) {
    for (int64_t j = 0; j < N; j++) {
      float acc = 0.0f;
      for (int64_t k = 0; k < K; k++) {
        const float a = a_trans ? A[k * lda + i] : A[i * lda + k];
        const float b = b_trans ? B[j * ldb + k] : B[k * ldb + j];
        if (fused) {
          acc = fmaf(a, b, acc);
        } else {
          volatile float p = a * b;
          acc = acc + p;
        }
      }
      volatile float t = alpha * acc;
      volatile float u = beta * C[i * ldc + j];
      volatile float w = t + u;
      C[i * ldc + j] = C[i * ldc + j] + w;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * cmodel direct convolution: simple_conv2d_impl_{fp,bp,wu} (mpemu/cmodel/simple/simple_conv2d_impl.cpp
 * :677-977) with gemm_kernel_conv_{fp,bp,wu} (:372-675) on MMEngine_avx2_ps / MMEngine_strideB_avx2_ps
 * (simple_mm_engine.cpp:13-44).  Square stride and padding, no dilation, no groups (simple.py:98-103).
 * What is restated is the accumulation chain of every result element as ONE thread executes the
 * reference (its OpenMP loops over ifm1 / ofm1 / oj race on shared results, so only the serial order is
 * defined).  The engine loads the running value, continues the chain over its k loop, stores it back:
 *   fp: out[n][k][oj][oi]  = bias[k]; for cb, kj, ki, c2 (c = cb*CB + c2): += w[k][c][kj][ki] * x_pad
 *   bp: gin[n][c][y][x]    = +0;      for ob, oj ascending (kj = y+pad - oj*s), ki ascending
 *                                     (oi = (x+pad - ki)/s), o2: += w[k][c][kj][ki] * gout[n][k][oj][oi]
 *   wu: gw[k][c][kj][ki]   = +0;      for n, oj, oi: += gout[n][k][oj][oi] * x_pad[n][c][oj*s+kj][oi*s+ki]
 * CB = 64-channel blocks (the whole C when C < 64).  Zero-padded taps ARE multiplied in fp and wu
 * (0 * w), as the reference does on its padded buffers; channel padding to 64 only adds +0 terms, which
 * matter for one case: a -0 bias becomes +0 when C % 64 != 0.  `fused` as in orc_gemm_f32.
 * The reference is only defined for C, K each < 64 or a multiple of 64 (its packing routines drop or
 * over-read channels otherwise, :143-160, :282-307); returns -1 for other shapes.
 * ---------------------------------------------------------------------------------------- */
static inline float mac(float acc, float a, float b, int fused) {
  if (fused) return fmaf(a, b, acc);
  volatile float p = a * b;
  return acc + p;
}

static int conv_shape_ok(int64_t C, int64_t K, int64_t stride, int64_t pad, int64_t H, int64_t W, int64_t R,
                         int64_t S) {
  if (stride < 1 || pad < 0 || H + 2 * pad < R || W + 2 * pad < S) return 0;
  if (C < 1 || K < 1 || (C > 64 && C % 64) || (K > 64 && K % 64)) return 0;
  return 1;
}

int orc_conv2d_fp(float *out, const float *in, const float *w, const float *bias, int64_t N, int64_t C,
                  int64_t H, int64_t W, int64_t K, int64_t R, int64_t S, int64_t stride, int64_t pad,
                  int fused) {
  if (!conv_shape_ok(C, K, stride, pad, H, W, R, S)) return -1;
  const int64_t OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
  const int64_t CB = C < 64 ? C : 64;
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t n = 0; n < N; n++)
    for (int64_t k = 0; k < K; k++)
      for (int64_t oj = 0; oj < OH; oj++)
        for (int64_t oi = 0; oi < OW; oi++) {
          float acc = bias ? bias[k] : 0.0f;
          if (C % 64) acc = acc + 0.0f; /* the +0 terms of the padded channels */
          for (int64_t cb = 0; cb < C / CB; cb++)
            for (int64_t kj = 0; kj < R; kj++)
              for (int64_t ki = 0; ki < S; ki++)
                for (int64_t c2 = 0; c2 < CB; c2++) {
                  const int64_t c = cb * CB + c2, y = oj * stride + kj - pad, x = oi * stride + ki - pad;
                  const float xv = (y >= 0 && y < H && x >= 0 && x < W) ? in[((n * C + c) * H + y) * W + x] : 0.0f;
                  acc = mac(acc, w[((k * C + c) * R + kj) * S + ki], xv, fused);
                }
          out[((n * K + k) * OH + oj) * OW + oi] = acc;
        }
  return 0;
}

int orc_conv2d_bp(float *gin, const float *gout, const float *w, int64_t N, int64_t C, int64_t H, int64_t W,
                  int64_t K, int64_t R, int64_t S, int64_t stride, int64_t pad, int fused) {
  if (!conv_shape_ok(C, K, stride, pad, H, W, R, S)) return -1;
  const int64_t OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
  const int64_t KB = K < 64 ? K : 64;
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t n = 0; n < N; n++)
    for (int64_t c = 0; c < C; c++)
      for (int64_t ih = 0; ih < H; ih++)
        for (int64_t iw = 0; iw < W; iw++) {
          const int64_t y = ih + pad, x = iw + pad;
          float acc = 0.0f;
          for (int64_t ob = 0; ob < K / KB; ob++)
            for (int64_t oj = 0; oj < OH; oj++) {
              const int64_t kj = y - oj * stride;
              if (kj < 0 || kj >= R) continue;
              for (int64_t ki = 0; ki < S; ki++) {
                const int64_t d = x - ki;
                if (d < 0 || d % stride || d / stride >= OW) continue;
                const int64_t oi = d / stride;
                for (int64_t o2 = 0; o2 < KB; o2++) {
                  const int64_t k = ob * KB + o2;
                  acc = mac(acc, w[((k * C + c) * R + kj) * S + ki], gout[((n * K + k) * OH + oj) * OW + oi], fused);
                }
              }
            }
          gin[((n * C + c) * H + ih) * W + iw] = acc;
        }
  return 0;
}

int orc_conv2d_wu(float *gw, const float *gout, const float *in, int64_t N, int64_t C, int64_t H, int64_t W,
                  int64_t K, int64_t R, int64_t S, int64_t stride, int64_t pad, int fused) {
  if (!conv_shape_ok(C, K, stride, pad, H, W, R, S)) return -1;
  const int64_t OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t k = 0; k < K; k++)
    for (int64_t c = 0; c < C; c++)
      for (int64_t kj = 0; kj < R; kj++)
        for (int64_t ki = 0; ki < S; ki++) {
          float acc = 0.0f;
          for (int64_t n = 0; n < N; n++)
            for (int64_t oj = 0; oj < OH; oj++)
              for (int64_t oi = 0; oi < OW; oi++) {
                const int64_t y = oj * stride + kj - pad, x = oi * stride + ki - pad;
                const float xv = (y >= 0 && y < H && x >= 0 && x < W) ? in[((n * C + c) * H + y) * W + x] : 0.0f;
                acc = mac(acc, gout[((n * K + k) * OH + oj) * OW + oi], xv, fused);
              }
          gw[((k * C + c) * R + kj) * S + ki] = acc;
        }
  return 0;
}
