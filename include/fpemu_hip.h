/*
 * fpemu_hip.h -- C ABI of libfpemu_hip.so, the MI355X (gfx950) fpemu cast.
 *
 * Drop-in boundary for the reference's native cast entry point: the pybind11 functions
 * `forward` / `backward` of mpemu/pytquant/{cpp,hip}/fpemu_impl.cpp
 *     std::vector<Tensor> (Tensor input, std::string mode, int size, bool inplace,
 *                          float scale, bool block_norm, int block_size)
 * (reference: pytquant/hip/fpemu_impl.cpp:27-53, pytquant/cpp/fpemu_impl.cpp:2013-2050).
 * Here the same call is expressed with plain pointers: the caller owns every buffer, `out`
 * may equal `in` (the reference's inplace=true), and work is enqueued on the given HIP
 * stream without any host synchronisation or allocation.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - return value: 0 on success, a negative FPEMU_E* code for argument errors, or a positive
 *     hipError_t value if the HIP runtime refused the launch.  fpemu_last_error() returns a
 *     thread-local description of the last failure;
 *   - no function here allocates, frees or synchronises.
 */
#ifndef FPEMU_HIP_H
#define FPEMU_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPEMU_ABI_VERSION 10

/* Storage dtype of in/out buffers.  FPEMU_F64 is served by fpemu_cast / fpemu_forward (plain casts, no block norm)
 * only -- the reference's GPU dispatcher takes double tensors (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 * hip/fpemu_kernels.hip:1070), its CPU path does not (data_ptr<float>, cpp/fpemu_impl.cpp:1879).  Semantics, as its
 * kernels compute (hip/fpemu_kernels.hip:429-431, 472): p = (float)(x * (double)scale), one rounding; the fp32
 * algorithm on p with scale 1; out = (double)(q * rcp).  Noise is indexed by element exactly as for fp32 storage.
 * Every other entry point answers FPEMU_EINVAL for it (a per-tensor scale of a double tensor is computed by the
 * caller in double, as the reference's Python does, and handed to fpemu_cast). */
enum { FPEMU_F32 = 0, FPEMU_F16 = 1, FPEMU_BF16 = 2, FPEMU_F64 = 3 };

/* Emulated target format.  One id per reference kernel:
 *   E5M2      E5M2_Kernel        cpp/fpemu_impl.cpp:765   (scalar routine :691-763 is canonical)
 *   E4M3      E4M3_Kernel        cpp/fpemu_impl.cpp:1135  (:1038-1133)
 *   E3M4      E3M4_Kernel        cpp/fpemu_impl.cpp:1720  (:1624-1718)
 *   BF16      BFLOAT16_Kernel    cpp/fpemu_impl.cpp:331   (RNE :207-232, SR :284-329)
 *   FP16      FLOAT16_Kernel     cpp/fpemu_impl.cpp:356
 *   E5M2_DAZ  E5M2_DAZ_Kernel    cpp/fpemu_impl.cpp:846
 *   FP16_DAZ  FLOAT16_Kernel with no_denorm=1 (cpp:1943-1947)
 *   E4M3_IEEE E4M3_IEEE_Kernel   cpp/fpemu_impl.cpp:1428  (max 240; the AVX-512 routines :1213-1328 are
 *                                canonical: the scalar routine saturates from 15.0 on, :1364)
 *   FP4       FP4_Nearest_Kernel cpp/fpemu_impl.cpp:1827  (:1797-1824; rounding-mode id FPEMU_RNE) */
enum {
  FPEMU_FMT_E5M2 = 0,
  FPEMU_FMT_E4M3 = 1,
  FPEMU_FMT_E3M4 = 2,
  FPEMU_FMT_BF16 = 3,
  FPEMU_FMT_FP16 = 4,
  FPEMU_FMT_E5M2_DAZ = 5,
  FPEMU_FMT_FP16_DAZ = 6,
  FPEMU_FMT_E4M3_IEEE = 7,
  FPEMU_FMT_FP4 = 8
};

/* Rounding mode: the reference's enum ROUNDING_MODES, cpp/fpemu_impl.cpp:15-23. */
enum {
  FPEMU_RTZ = 0,
  FPEMU_RNE = 1,
  FPEMU_STOCHASTIC = 2,
  FPEMU_RNAZ = 3,
  FPEMU_RNTZ = 4,
  FPEMU_RPINF = 5,
  FPEMU_RNINF = 6
};

/* Error codes (negative). */
enum {
  FPEMU_OK = 0,
  FPEMU_EINVAL = -1,       /* null pointer, bad dtype/fmt/rmode id, scale > FLT_MAX (the reference's test, cpp:772-776) */
  FPEMU_EMODE = -2,        /* mode string not recognised (the reference silently no-ops) */
  FPEMU_EUNSUPPORTED = -3, /* recognised but not built (no such case at present) */
  FPEMU_EALIGN = -4        /* pointer not aligned to the element size */
};

int fpemu_abi_version(void);
const char *fpemu_last_error(void);

/* Mode-string dispatcher: replaces the if/else chain of fpemu_common_function
 * (cpp/fpemu_impl.cpp:1878-2000; hip/fpemu_kernels.hip:1069-1386).  Accepts every string the
 * reference dispatches plus the three RTZ extensions BASELINE.json names (E4M3_RTZ, E3M4_RTZ,
 * BFLOAT16_RTZ).  Returns FPEMU_EMODE for an unknown string. */
int fpemu_parse_mode(const char *mode, int *fmt, int *rmode);

/* The cast itself: out[i] = dequant(quant_fmt(in[i] * scale)) * (1/scale), i in [0, n).
 * Replaces E5M2_Kernel / E4M3_Kernel / E3M4_Kernel / BFLOAT16_Kernel / FLOAT16_Kernel /
 * E5M2_DAZ_Kernel (cpp/fpemu_impl.cpp:331,356,765,846,1135,1720 and their GPU twins
 * hip/fpemu_kernels.hip:254-995).  `out == in` is the in-place form.
 *   seed, offset : key and call counter of the Philox4x32-10 stream used by FPEMU_STOCHASTIC
 *                  (replaces the global xoshiro state cpp:41-88 / hip:47-102);
 *   elem_base    : global index of in[0] inside the logical tensor (sharded runs; 0 otherwise).
 * 64-bit n: the reference's `int size` overflows at 2^31 elements. */
int fpemu_cast(const void *in, void *out, uint64_t n, int dtype, int fmt, int rmode, float scale,
               uint64_t seed, uint64_t offset, uint64_t elem_base, void *stream);

/* The same cast over a list of buffers in as few launches as possible (32 buffers per launch): tensor i
 * is cast exactly as fpemu_cast(in_i, out_i, n_i, dtype, fmt, rmode, scale, seed, offset + i, 0, stream),
 * bit for bit -- what a loop of calls with a counter that advances by one per call produces.  Replaces
 * the reference's per-parameter loops of FPEmuOp calls after the optimizer step
 * (mpemu/emulator_base.py:update_fp16_master_params / update_e5m2_master_params; one cast launch + one
 * hipDeviceSynchronize per parameter there).  `tensors` is host memory, read before the call returns. */
typedef struct fpemu_tensor {
  const void *in;
  void *out;
  uint64_t n;
} fpemu_tensor_t;
int fpemu_cast_multi(const fpemu_tensor_t *tensors, uint64_t count, int dtype, int fmt, int rmode, float scale,
                     uint64_t seed, uint64_t offset, void *stream);

/* fpemu_quantize_per_tensor for a LIST of buffers of one storage type and mode in two launches per 32 tensors
 * (per-workgroup maxima of every tensor; then the multi-tensor cast, whose workgroups reduce their tensor's
 * maxima to its scale) instead of one to three launches per tensor.  For the callers that quantise many weights
 * in a row with one configuration: quantize_model_weights (reference qutils.py:478-487, a Python loop of
 * quantize_tensor over the model's modules) and the emulators' weight hooks.  Tensor i is quantised exactly as
 * fpemu_quantize_per_tensor(in_i, out_i, n_i, ..., seed, offset + i): the same bits as the loop it replaces.
 *   ws : count + 32 * 256 words for the full effect (at least count + min(count, 32)), contents need not be
 *        initialised; in stream order after the call ws[i] holds tensor i's amax bit pattern (empty tensors:
 *        not written).  One workspace per stream.
 * Formats without a scale (BF16 / FP16) are rejected: they are fpemu_cast_multi. */
int fpemu_quantize_per_tensor_multi(const fpemu_tensor_t *tensors, uint64_t count, int dtype, int fmt, int rmode,
                                    float flt_max, int clamp, uint32_t *ws, uint64_t ws_words, uint64_t seed,
                                    uint64_t offset, void *stream);

/* The two halves of fpemu_quantize_per_tensor_multi for tensors that are SHARDS of larger tensors (one process per
 * GPU): between them the caller all-reduces (MAX) the `count` words over its process group -- one collective for
 * the whole list, instead of one per weight as a loop of quantize_weights_of_a_module would issue
 * (reference qutils.py:470-475; the reference itself has no collective).
 *   fpemu_amax_multi        : amax_words[i] = bit pattern of max |tensor i| (the words are zeroed first, in stream
 *                             order; empty tensors stay 0), two launches + one per 32 tensors;
 *   fpemu_cast_multi_scaled : tensor i is cast exactly as fpemu_cast_scaled(in_i, out_i, n_i, ..., amax_words + i,
 *                             flt_max, clamp, seed, offset + i, elem_base 0), one launch per 32 tensors.
 * Without the exchange the pair equals fpemu_quantize_per_tensor_multi bit for bit. */
int fpemu_amax_multi(const fpemu_tensor_t *tensors, uint64_t count, int dtype, uint32_t *amax_words, void *stream);
int fpemu_cast_multi_scaled(const fpemu_tensor_t *tensors, uint64_t count, int dtype, int fmt, int rmode,
                            const uint32_t *amax_words, float flt_max, int clamp, uint64_t seed, uint64_t offset,
                            void *stream);

/* Mirror of the pybind `forward` signature (hip/fpemu_impl.cpp:27-37): mode string, size,
 * scale, block_norm, block_size.  inplace <=> out == in.  block_norm: as the reference
 * (cpp:2020-2025) demoted to false when block_size != size and size % block_size != 0; otherwise
 * fpemu_quantize_blocks for the formats that have the branch, ignored (plain cast) by the others
 * (E5M2_DAZ, BFLOAT16, FLOAT16: their reference kernels take no block_norm). */
int fpemu_forward(const void *in, void *out, int dtype, const char *mode, uint64_t size,
                  float scale, int block_norm, int block_size, uint64_t seed, uint64_t offset,
                  void *stream);

/* |x| max-reduction feeding the per-tensor scale (replaces torch.max(abs(flatten(t))),
 * qutils.py:283, and the never-launched absmax0 kernel hip/fpemu_kernels.hip:194-220).
 * *amax_bits = max(*amax_bits, max_i bits(|in[i]|)) as unsigned integers, which equals the
 * float maximum and makes any NaN win (torch.max propagates NaN).  The caller zeroes
 * *amax_bits (4 bytes, device) before the first call; several calls may accumulate into it. */
int fpemu_amax(const void *in, uint64_t n, int dtype, uint32_t *amax_bits, void *stream);

/* Cast with the scale derived ON DEVICE from an amax word (no host sync):
 *   scale = fp32(fp32(1/vmax) * flt_max);  if (clamp && scale > 3.275e4) scale = 6.55e4
 * exactly as quantize_tensor computes it (qutils.py:284-285; clamp=0 is the fine-grained
 * branch qutils.py:328-330).  amax_bits is read, not modified. */
int fpemu_cast_scaled(const void *in, void *out, uint64_t n, int dtype, int fmt, int rmode,
                      const uint32_t *amax_bits, float flt_max, int clamp, uint64_t seed,
                      uint64_t offset, uint64_t elem_base, void *stream);

/* Cast with the scale itself read ON DEVICE from *scale (one fp32 word, read in stream order; no host sync).
 * For the per-tensor "mean" policy of quantize_tensor (qutils.py:276-281, 287), whose statistic -- torch.mean --
 * and scale arithmetic stay torch ops on the device, and for any caller that already holds its scale in device
 * memory.  Identical to fpemu_cast(..., scale = *scale, ...) except that the scale is not checked: a zero,
 * infinite or NaN word gives what the arithmetic gives (the reference checks nothing either). */
int fpemu_cast_device_scale(const void *in, void *out, uint64_t n, int dtype, int fmt, int rmode, const float *scale,
                            uint64_t seed, uint64_t offset, uint64_t elem_base, void *stream);

/* quantize_tensor's per-tensor "max" branch in one call (qutils.py:282-287; clamp=0 gives the
 * fine-grained branch's unclamped scale, qutils.py:328-332): zero the workspace word, amax,
 * device-side scale, cast.  amax_ws: 4 bytes of device memory owned by the caller (holds the
 * amax bits afterwards). */
int fpemu_quantize_per_tensor(const void *in, void *out, uint64_t n, int dtype, int fmt, int rmode,
                              float flt_max, int clamp, uint32_t *amax_ws, uint64_t seed,
                              uint64_t offset, void *stream);

/* fpemu_quantize_per_tensor with a scratch workspace instead of the single amax word.  Above the
 * one-workgroup size the library then runs TWO launches -- up to 256 per-block maxima into ws[1..], and
 * the cast, whose blocks reduce those maxima themselves -- instead of three (zero the word, atomic amax,
 * cast): no atomics and no word to initialise; measured faster up to 512 MiB (11.6 -> 8.6 us at 3 MiB,
 * 53 -> 34 us at 64 MiB, 0.259 -> 0.246 ms at 512 MiB).  Tensors above 768 MiB take the single-word
 * sequence here as well (0.489 vs 0.503 ms at 1 GiB).  Results are identical.
 * ws: ws_words >= 1 device words (257 for the full effect), contents need not be initialised; in stream
 * order after the call ws[0] holds the amax bit pattern.  One workspace per stream (calls on different
 * streams must not share one). */
int fpemu_quantize_per_tensor_ws(const void *in, void *out, uint64_t n, int dtype, int fmt, int rmode,
                                 float flt_max, int clamp, uint32_t *ws, uint64_t ws_words, uint64_t seed,
                                 uint64_t offset, void *stream);

/* fpemu_quantize_per_tensor_ws WITHOUT a workspace argument and without any allocation.  No counterpart in
 * the reference: it is for callers that cannot allocate device memory or carry a workspace through their
 * interfaces -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  The
 * scratch (amax word + per-block maxima) is a 1280-byte __device__ array inside the library, one copy per
 * GPU, held until the process ends; its address is looked up once per device on the first call there
 * (the stream's device must be the current device, as for every entry point).  A device's FIRST call must therefore
 * be made outside stream capture (hipGetSymbolAddress is not a capturable call): warm the entry point up once per
 * device before capturing graphs that contain it.  Results are bit-identical to
 * fpemu_quantize_per_tensor_ws with a 257-word workspace.
 *   amax_out : optional caller-owned device word that receives the amax bit pattern (NULL: not reported);
 *   n        : at most FPEMU_STATIC_MAX_ELEMS, else FPEMU_EINVAL (nothing is enqueued).
 * Every call on a GPU shares that GPU's scratch: calls in one stream are ordered and safe; calls that
 * overlap on one GPU (different streams, or different host threads on different streams) RACE on it --
 * serialise them or use fpemu_quantize_per_tensor_ws with one workspace per stream.  Calls no hipMalloc,
 * hipFree or other allocating routine. */
#define FPEMU_STATIC_MAX_ELEMS (1ull << 36)
int fpemu_quantize_per_tensor_static(const void *in, void *out, uint64_t n, int dtype, int fmt, int rmode,
                                     float flt_max, int clamp, uint32_t *amax_out, uint64_t seed,
                                     uint64_t offset, void *stream);

/* Per-row amax -> scale -> cast of a contiguous [rows, cols] buffer in one launch: row r gets
 * scale_r = fp32(fp32(1/max|row r|) * flt_max) (+ clamp rule when clamp != 0).  Replaces the Python
 * loops of quantize_tensor's per-channel branch (qutils.py:289-311, clamp = 1, rows = shape[0]) and
 * fine-grained branch (qutils.py:313-337, clamp = 0, rows = out_channels x groups when the groups are
 * equal-sized).  Results equal a loop of fpemu_quantize_per_tensor over the rows with stochastic
 * offsets offset, offset+1, ...  Formats without a scale (BF16 / FP16) are rejected. */
int fpemu_quantize_rows(const void *in, void *out, uint64_t rows, uint64_t cols, int dtype, int fmt,
                        int rmode, float flt_max, int clamp, uint64_t seed, uint64_t offset,
                        void *stream);

/* Cast of a contiguous [rows, cols] buffer with one scale PER ROW read from device memory (scales[r], fp32), one
 * launch.  Replaces the loop of per-channel casts of quantize_tensor's "mean" policy (qutils.py:289-311 with
 * :294-301: per row a torch.mode, a host-synchronising scale computation and one cast launch): the caller computes
 * all rows' scales with torch ops on the device and passes the array.  Results equal a loop of
 * fpemu_cast(row r, scale = scales[r], offset = offset + r) over the rows.  Scales are not checked (see
 * fpemu_cast_device_scale).  Formats without a scale (BF16 / FP16) are rejected. */
int fpemu_cast_rows_scaled(const void *in, void *out, uint64_t rows, uint64_t cols, int dtype, int fmt, int rmode,
                           const float *scales, uint64_t seed, uint64_t offset, void *stream);

/* The two halves of fpemu_quantize_rows for a [rows, cols] tensor whose rows are spread over several GPUs (SURVEY
 * 8(e): "all-reduce(max) of the C-vector of amaxes"; the reference's per-channel loop, qutils.py:289-311, knows one
 * device only).  The buffer handed in is `rows` row SEGMENTS of `cols` elements each, contiguous:
 *   fpemu_amax_rows      : amax_bits[r] = bit pattern of max |segment r| (plain store; a NaN wins, as torch.max);
 *   -- the caller combines the vectors of all ranks with an integer MAX all-reduce --
 *   fpemu_cast_rows_amax : segment r is cast with scale_from_amax(amax_bits[r], flt_max, clamp) (clamp != 0: the
 *                          per-channel rule qutils.py:308; 0: the fine-grained one, :328-330).  Stochastic modes:
 *                          segment r draws as call `offset + r` and its element c as element `col_base + c` of that
 *                          call, i.e. exactly what fpemu_quantize_rows draws for column col_base + c of row r of the
 *                          whole tensor -- results do not depend on how the rows are cut.
 * Without the exchange (col_base 0) the pair equals fpemu_quantize_rows bit for bit.  Formats without a scale
 * (BF16 / FP16) are rejected by the cast. */
int fpemu_amax_rows(const void *in, uint64_t rows, uint64_t cols, int dtype, uint32_t *amax_bits, void *stream);
int fpemu_cast_rows_amax(const void *in, void *out, uint64_t rows, uint64_t cols, int dtype, int fmt, int rmode,
                         const uint32_t *amax_bits, float flt_max, int clamp, uint64_t seed, uint64_t offset,
                         uint64_t col_base, void *stream);

/* Block-normalised cast: the reference's forward(..., block_norm = true, block_size) for the formats
 * whose kernels have that branch (E5M2_Kernel cpp/fpemu_impl.cpp:778-815, E4M3_Kernel :1149-1190,
 * E4M3_IEEE_Kernel :1441-1478, E3M4_Kernel :1734-1770, FP4_Nearest_Kernel :1840-1860).  The buffer is
 * n / block_size consecutive blocks; per block: maxval = max |x| (a NaN is never taken), p2 = maxval
 * with its mantissa cleared, scale = 2*p2/16384 (E5M2) | 2*p2/8 (E4M3, E4M3_IEEE) | 2*p2 (E3M4) |
 * 1/maxval (FP4), then the ordinary cast of the block.  One launch for all blocks.
 * n % block_size must be 0 (fpemu_forward applies the reference's demotion rule, cpp:2020-2025, before
 * calling this).  Stochastic modes: block b draws as call `offset + b`, element index = position in
 * the block. */
int fpemu_quantize_blocks(const void *in, void *out, uint64_t n, uint64_t block_size, int dtype, int fmt,
                          int rmode, uint64_t seed, uint64_t offset, void *stream);

/* cmodel emulated GEMM ("next" row of the hot path; BASELINE config 5).  Replaces the reference's CPU
 * cmodel entry point  simple_gemm_dev.gemm(C, A, B, alpha, a_trans, b_trans)
 * (mpemu/cmodel/simple/simple_gemm.cpp:45-67 -> simple_sgemm_impl, simple_gemm_impl.cpp:71-237 ->
 * MMEngine_avx2_ps, simple_mm_engine.cpp:13-27).  Row-major fp32 matrices:
 *   C is M x N (ldc);  op(A) is M x K: A stored [M][K], or [K][M] when a_trans;  op(B) is K x N: B stored
 *   [K][N], or [N][K] when b_trans;  lda / ldb are the stored matrices' row strides in elements.
 *   C[i][j] = C[i][j] + (alpha * acc_ij + 0 * C[i][j]),   acc_ij = chain over k ascending from +0
 * variant 1 (FPEMU_GEMM_FMA, the drop-in default): acc = fmaf(a, b, acc) per step, on
 *            v_mfma_f32_32x32x2_f32.  This is the reference as its own setup.py builds it on an FMA-capable
 *            x86 host (-march=native, g++'s default -ffp-contract=fast contracts the engine's
 *            _mm256_add_ps(c, _mm256_mul_ps(a, b)), simple_mm_engine.cpp:13-27, into vfmadd231ps).
 * variant 0 (FPEMU_GEMM_SEPARATE): acc = fp32(acc + fp32(a*b)) per step, VALU, LDS-tiled: the same source
 *            compiled for a host without FMA (separate vmulps / vaddps).
 * Both are bit-identical to the corresponding reference build (tests/golden/gemm.npz holds both).
 * C accumulates into its existing contents exactly as the reference does: pass zeros for a plain product. */
int fpemu_gemm_f32(float *C, const float *A, const float *B, int64_t M, int64_t N, int64_t K,
                   int64_t lda, int64_t ldb, int64_t ldc, float alpha, int a_trans, int b_trans,
                   int variant, void *stream);

/* fpemu_gemm_f32 that also reports the maximum magnitude of what it wrote: *amax_bits = max(*amax_bits, bits(|C[i][j]|))
 * over every element of the final C, as unsigned integers (= the float maximum; a NaN wins, as torch.max) -- the
 * convention of fpemu_amax, so several calls may accumulate into one word that the caller zeroed first (the slices of
 * a stacked matmul, reference simple.py:52-63).  The reference quantises a hooked module's output right after the
 * patched matmul / conv (quantize_module_output, qutils.py:586-609: torch.max(abs(output)) then the cast); here the
 * epilogue has the values in registers, so that amax pass over the output disappears and the per-tensor
 * quantisation of the output is fpemu_cast_scaled(C, ..., amax_bits, ...) alone: 8 instead of 12 bytes per element.
 * The matrix result is bit-identical to fpemu_gemm_f32. */
int fpemu_gemm_f32_amax(float *C, const float *A, const float *B, int64_t M, int64_t N, int64_t K, int64_t lda,
                        int64_t ldb, int64_t ldc, float alpha, int a_trans, int b_trans, int variant,
                        uint32_t *amax_bits, void *stream);

/* Chain-mode (and batched) form of the cmodel GEMM, the building block of the cmodel convolution: the
 * accumulators START at C's current values and C = acc (what MMEngine_avx2_ps does inside the
 * reference's conv kernels: load C, continue the k-ascending chain, store; simple_mm_engine.cpp:18-24),
 * instead of simple_gemm's C += alpha*acc.  `batch` independent problems; operand b is at
 * X + b*stride_x elements (stride 0 = shared).  k_segments >= 1: the k loop runs that many times over
 * K into the same accumulators, A and B advancing by kseg_stride_a / _b elements each time -- ONE chain
 * of k_segments*K products (the conv weight update chains image after image,
 * simple_conv2d_impl.cpp:645-667).  variant as fpemu_gemm_f32. */
int fpemu_gemm_chain_f32(float *C, const float *A, const float *B, int64_t M, int64_t N, int64_t K,
                         int64_t lda, int64_t ldb, int64_t ldc, int a_trans, int b_trans, int variant,
                         int64_t batch, int64_t stride_a, int64_t stride_b, int64_t stride_c,
                         int64_t k_segments, int64_t kseg_stride_a, int64_t kseg_stride_b, void *stream);

/* cmodel direct convolution: replaces simple_conv2d_dev.conv2d_fp / conv2d_bp / conv2d_wu
 * (mpemu/cmodel/simple/simple_conv2d.cpp:44-121 -> simple_conv2d_impl.cpp:677-977).  NCHW activations,
 * KCRS weights, fp32, square stride and padding, no dilation, no groups (simple.py:98-103).  Every
 * result element is the reference's serial accumulation chain (csrc/fpemu_conv.hip states the three
 * orders); variant selects fmaf chain (1, = the reference as built on FMA hosts) or separate roundings
 * (0).  C and K must each be <= 64 or a multiple of 64.  bias may be NULL (zeros, simple.py:193-196).
 *   fp: out [N,K,OH,OW]  <- in [N,C,H,W], w [K,C,R,S], bias [K]
 *   bp: gin [N,C,H,W]    <- gout [N,K,OH,OW], w         (gin is overwritten, not accumulated)
 *   wu: gw  [K,C,R,S]    <- gout, in                     (gw is overwritten)
 * ws: caller-owned device workspace of at least fpemu_conv2d_workspace_bytes(kind, ...) bytes, 16-byte
 * aligned (packed weights + the gathered GEMM operand of a chunk of images); a smaller one still works
 * while it holds one image.  No allocation, no host sync. */
enum { FPEMU_CONV_FP = 0, FPEMU_CONV_BP = 1, FPEMU_CONV_WU = 2 };
uint64_t fpemu_conv2d_workspace_bytes(int kind, int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                                      int64_t R, int64_t S, int64_t stride, int64_t pad);
int fpemu_conv2d_fp_f32(float *out, const float *in, const float *w, const float *bias, int64_t N, int64_t C,
                        int64_t H, int64_t W, int64_t K, int64_t R, int64_t S, int64_t stride, int64_t pad,
                        int variant, void *ws, uint64_t ws_bytes, void *stream);
/* fpemu_conv2d_fp_f32 that also folds max |out| into *amax_bits (see fpemu_gemm_f32_amax: same convention, same use). */
int fpemu_conv2d_fp_f32_amax(float *out, const float *in, const float *w, const float *bias, int64_t N, int64_t C,
                             int64_t H, int64_t W, int64_t K, int64_t R, int64_t S, int64_t stride, int64_t pad,
                             int variant, void *ws, uint64_t ws_bytes, uint32_t *amax_bits, void *stream);
int fpemu_conv2d_bp_f32(float *gin, const float *gout, const float *w, int64_t N, int64_t C, int64_t H,
                        int64_t W, int64_t K, int64_t R, int64_t S, int64_t stride, int64_t pad, int variant,
                        void *ws, uint64_t ws_bytes, void *stream);
int fpemu_conv2d_wu_f32(float *gw, const float *gout, const float *in, int64_t N, int64_t C, int64_t H,
                        int64_t W, int64_t K, int64_t R, int64_t S, int64_t stride, int64_t pad, int variant,
                        void *ws, uint64_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
