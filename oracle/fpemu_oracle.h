/*
 * fpemu_oracle.h -- CPU restatement of the reference fpemu cast (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X HIP cast.  It restates, in plain C, the
 * algorithm of the reference's CPU extension
 *   /root/reference/mpemu/pytquant/cpp/fpemu_impl.cpp   (cited per function below)
 * and is pinned against that extension built from its own sources (oracle/build_ref.py ->
 * oracle/_ref/) through the committed fixtures in tests/golden/ (tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (fp8-emulation-toolkit_amd/) never links, imports or calls it.
 */
#ifndef FPEMU_ORACLE_H
#define FPEMU_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Format ids (same numbering as include/fpemu_hip.h). */
enum {
  ORC_FMT_E5M2 = 0,     /* cpp/fpemu_impl.cpp:691-763  cvt_fp32_e5m2_scalar */
  ORC_FMT_E4M3 = 1,     /* cpp/fpemu_impl.cpp:1038-1133 cvt_fp32_e4m3_scalar */
  ORC_FMT_E3M4 = 2,     /* cpp/fpemu_impl.cpp:1624-1718 cvt_fp32_e3m4_scalar */
  ORC_FMT_BF16 = 3,     /* cpp/fpemu_impl.cpp:207-232 (RNE, canonical), :284-329 (SR) */
  ORC_FMT_FP16 = 4,     /* cpp/fpemu_impl.cpp:356-423  FLOAT16_Kernel */
  ORC_FMT_E5M2_DAZ = 5, /* cpp/fpemu_impl.cpp:846-919  E5M2_DAZ_Kernel */
  ORC_FMT_FP16_DAZ = 6, /* FLOAT16_Kernel with no_denorm=1 (:374-377) */
  ORC_FMT_E4M3_IEEE = 7,/* cpp/fpemu_impl.cpp:1213-1328 (AVX-512 routines: canonical, see .c) */
  ORC_FMT_FP4 = 8       /* cpp/fpemu_impl.cpp:1797-1824 cvt_fp32_fp4_nearest_scalar */
};

/* Rounding-mode ids: the reference's own enum, cpp/fpemu_impl.cpp:15-23. */
enum {
  ORC_RTZ = 0,
  ORC_RNE = 1,
  ORC_STOCHASTIC = 2,
  ORC_RNAZ = 3,
  ORC_RNTZ = 4,
  ORC_RPINF = 5,
  ORC_RNINF = 6
};

/* fp32 -> fp16 RNE exactly as x86 F16C vcvtps2ph (cpp/fpemu_impl.cpp:117-119), in software. */
uint16_t orc_f32_to_f16_rne(float f);
/* fp16 -> fp32 (exact), cpp/fpemu_impl.cpp:121-123. */
float orc_f16_to_f32(uint16_t h);

/* 16-bit -> 16-bit quantiser on the fp16 pattern (SURVEY Appendix A "Q_fmt").
 * rnd supplies the random bits for ORC_STOCHASTIC (low 8/7/6 bits used). */
uint16_t orc_quantize_h(uint16_t h, int fmt, int rmode, uint32_t rnd);

/* One element, all formats.  `rcp` must be (float)(1.0/(double)scale) (cpp:714). */
float orc_cast_one(float x, int fmt, int rmode, float scale, float rcp, uint32_t rnd);

/* (float)(1.0 / (double)scale), cpp/fpemu_impl.cpp:714,1060,1646. */
float orc_scale_reciprocal(float scale);

/* Philox4x32-10 (Salmon et al., SC'11).  out[4] = philox(key={k0,k1}, ctr={c0..c3}). */
void orc_philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                       uint32_t c3, uint32_t out[4]);

/* The random word element `idx` (global element index) receives for a given (seed, offset):
 * the product's counter-based replacement for the reference's racy xoshiro state
 * (cpp/fpemu_impl.cpp:41-88).  Low bits are the rounding noise (8/7/6 for the FP8 formats,
 * 16 for BF16, 13 for FP16).  Layout documented in DESIGN.md "Stochastic rounding". */
uint32_t orc_rand_bits(uint64_t seed, uint64_t offset, uint64_t idx, int fmt);

/* Whole-buffer cast, fp32 in / fp32 out (may alias).  elem_base = global index of in[0]
 * (sharded runs), only used by ORC_STOCHASTIC.  OpenMP-parallel when built with -fopenmp.
 * Returns 0, or -1 on an unknown fmt/rmode combination. */
int orc_cast_f32(const float *in, float *out, uint64_t n, int fmt, int rmode, float scale,
                 uint64_t seed, uint64_t offset, uint64_t elem_base);

/* Block-normalised cast (block_norm = true of the pybind forward): per block of block_size elements
 * the scale is derived from the block's max |x| (formulas in the .c), then the ordinary cast.
 * n % block_size must be 0.  Returns 0 / -1. */
float orc_blocknorm_scale(float maxval, int fmt);
int orc_cast_blocknorm_f32(const float *in, float *out, uint64_t n, int fmt, int rmode,
                           uint64_t block_size, uint64_t seed, uint64_t offset);

/* 16-bit storage variants: up-convert (exact), canonical fp32 algorithm, RNE down-convert
 * (SURVEY section 8(a) "Non-fp32 inputs").  is_bf16 selects bf16 vs fp16 storage. */
int orc_cast_16(const uint16_t *in, uint16_t *out, uint64_t n, int is_bf16, int fmt, int rmode,
                float scale, uint64_t seed, uint64_t offset, uint64_t elem_base);

/* max |x| as torch.max(abs(x)) computes it (NaN propagates), qutils.py:283.  Returns the
 * fp32 bit pattern of the result in *bits as well (NaN payload is unspecified -> 0x7FC00000). */
float orc_amax_f32(const float *in, uint64_t n);

/* Per-tensor "max" scale of quantize_tensor (qutils.py:283-285):
 * scale = fp32(fp32(1/vmax) * flt_max); scale > 3.275e4 -> 6.55e4. */
float orc_scale_from_amax(float vmax, float flt_max);

/* Fine-grained variant (qutils.py:328-330): same product, no clamp. */
float orc_scale_from_amax_noclamp(float vmax, float flt_max);

/* Fill lut[65536] with orc_quantize_h(h, fmt, rmode, 0) for every fp16 pattern h. */
void orc_build_lut(int fmt, int rmode, uint16_t *lut);

/* Mode-string parser mirroring the dispatcher chain cpp/fpemu_impl.cpp:1878-2000.
 * Returns 0 and fills fmt/rmode, or -1 for a string the oracle does not implement. */
int orc_parse_mode(const char *mode, int *fmt, int *rmode);

/* cmodel emulated GEMM (reference simple_gemm_impl.cpp:71-237, simple_mm_engine.cpp:13-27): per
 * element a k-ascending fp32 chain; fused == 0: separate multiply and add (reference built without
 * FMA); fused != 0: fmaf chain (reference built with its own flags on an FMA host).
 * C accumulates: C += alpha*acc + 0*C. */
void orc_gemm_f32(float *C, const float *A, const float *B, int64_t M, int64_t N, int64_t K,
                  int64_t lda, int64_t ldb, int64_t ldc, float alpha, int a_trans, int b_trans,
                  int fused);

/* cmodel direct convolution, forward / backward-data / weight-update (reference
 * simple_conv2d_impl.cpp:677-977): NCHW / KCRS fp32, square stride and padding.  Chains documented in
 * the .c.  Return 0, or -1 for shapes the reference itself mishandles (C or K > 64 and not a multiple
 * of 64). */
int orc_conv2d_fp(float *out, const float *in, const float *w, const float *bias, int64_t N, int64_t C,
                  int64_t H, int64_t W, int64_t K, int64_t R, int64_t S, int64_t stride, int64_t pad,
                  int fused);
int orc_conv2d_bp(float *gin, const float *gout, const float *w, int64_t N, int64_t C, int64_t H, int64_t W,
                  int64_t K, int64_t R, int64_t S, int64_t stride, int64_t pad, int fused);
int orc_conv2d_wu(float *gw, const float *gout, const float *in, int64_t N, int64_t C, int64_t H, int64_t W,
                  int64_t K, int64_t R, int64_t S, int64_t stride, int64_t pad, int fused);

/* Number of OpenMP threads the library will use (1 when built without OpenMP). */
int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
