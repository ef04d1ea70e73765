// fpemu_gemm.hip -- the cmodel "simple" emulated GEMM for MI355X (gfx950).
//
// Replaces the reference's CPU cmodel GEMM
//   simple_sgemm_impl   mpemu/cmodel/simple/simple_gemm_impl.cpp:71-237
//   MMEngine_avx2_ps    mpemu/cmodel/simple/simple_mm_engine.cpp:13-27
//   simple_gemm         mpemu/cmodel/simple/simple_gemm.cpp:45-67
// whose numerical contract is: every output element is an fp32 chain over k ASCENDING, starting from
// +0, of the engine's add(c, mul(a, b)); then  C = C + (alpha * acc + beta * C)  with beta = 0
// (impl.cpp:226-231).  No FP8 rounding happens inside: operands are pre-quantised by the cast kernels.
// How add(c, mul(a, b)) rounds depends on how the reference is compiled: with its own flags
// (-march=native, g++ default -ffp-contract=fast) on an FMA host it becomes vfmadd231ps, an fmaf chain;
// without FMA it is a rounded product then a rounded sum.  Both builds are pinned (tests/golden/gemm.npz).
//
// gemm_mfma_kernel   the fmaf chain, on v_mfma_f32_32x32x2_f32 (k-ordered, one rounding per step; bit-
//                    exact against the reference's FMA build).  128x128x32 tiles, 4 waves, each a 64x64
//                    quadrant as 2x2 MFMA tiles.  Roofline: fp32 MFMA, 157 TFLOP/s.
// gemm_mfma16_kernel the same chain on v_mfma_f32_16x16x4_f32 for latency-bound problems (few outputs, long k:
//                    the conv weight update): a quarter of the per-k latency on four times as many waves.
// gemm_seqk_kernel   separate roundings: LDS-tiled VALU kernel, v_mul_f32 + v_add_f32 per step
//                    (-ffp-contract=off), 8x8 outputs per lane, register prefetch of the next tile,
//                    XCD-aware tile order.  Bit-exact against the reference's no-FMA build.  Roofline:
//                    VALU at one multiply + one add per MAC = 78.6 TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/fpemu_hip.h"

extern "C" void fpemu_set_error_(const char *msg);  // fpemu_hip.hip: thread-local error text
extern "C" int fpemu_gemm_tile_edge_(int64_t M, int64_t N, int64_t batch, int variant);   // below, after the kernels

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// a 16-byte vector that is only 4-byte aligned in memory: gfx950 global loads take it as one dwordx4
// (the implicit-convolution operand: rows of the padded image shifted by the filter tap)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int BK = 16;  // VALU kernel: 128 x 128 x 16 tiles (64 x 64 x 16 for small problems)
#ifndef FPEMU_SEQK_T64_BELOW
#define FPEMU_SEQK_T64_BELOW 1024  // fewer 128 x 128 tiles than this (per launch): 64 x 64 tiles.  Measured, fraction of the
                                   // 78.6 TFLOP/s VALU peak with 128 / 64 tiles: 768x768x4096 0.077 / 0.205, 2048^3 0.54 / 0.61,
                                   // 3072x3072x2048 (576 tiles) 0.55 / 0.65, 4096^3 (1024 tiles) 0.72 / 0.67
#endif
#ifndef FPEMU_GEMM_MBK
#define FPEMU_GEMM_MBK 32
#endif
constexpr int MBK = FPEMU_GEMM_MBK;         // k-depth of the MFMA kernel's tiles (32: half the barriers of 16)
constexpr int NT = 256;
// LDS images are k-major, T[k][r], row stride ROWS + 4 floats (2-way instead of 4-way write conflicts;
// a 2-way ds_write_b32 conflict is free)
template <int ROWS> constexpr int ldt() { return ROWS + 4; }

// Operand view: element (r, k) of op(X) with r the M (or N) index.
//   KCONT = true : memory is [R][K] row-major (k contiguous)   -> A not transposed, B transposed
//   KCONT = false: memory is [K][R] row-major (r contiguous)   -> A transposed,     B not transposed
struct Operand {
  const float *p;
  int64_t ld;   // leading dimension (elements) of the stored matrix
  int R, K;     // extents of op(X): R rows (M or N), K depth
  int vec;      // 16-byte loads allowed (alignment + extents checked on the host)
  // What a load outside the matrix yields.  A tile's ragged k tail multiplies A's filler by B's: -0 for A and +0 for
  // B make every such product -0, the one value that leaves ANY accumulator as it is (acc + -0 = acc; with +0 * +0 a
  // chain sitting at -0 would turn +0).  Rows outside the matrix only feed results that are never stored.
  float pad;
};

// Stage one TBK x ROWS tile (rows r0.., depth k0..) into registers: TBK*ROWS/1024 float4 per thread.
// FULL: the whole tile is inside the matrix and 16-byte loads are allowed (checked on the host): no
// bounds tests in the loop.
template <bool KCONT, int TBK, int ROWS>
__device__ __forceinline__ void tile_coords(int h, int tid, int &r, int &k) {
  if constexpr (KCONT) {  // thread -> (row, 4 consecutive k): 4 threads per row, 64 rows per pass
    constexpr int RP = ROWS / 64;
    r = (h % RP) * 64 + (tid >> 2);
    k = (h / RP) * 16 + (tid & 3) * 4;
  } else {                // thread -> (k, 4 consecutive rows): ROWS/4 threads per k
    constexpr int TPR = ROWS / 4, KPP = NT / TPR;
    k = h * KPP + tid / TPR;
    r = (tid % TPR) * 4;
  }
}

template <bool KCONT, bool FULL, int TBK, int ROWS, bool UNALIGNED = false>
__device__ __forceinline__ void tile_load(const Operand &X, const float *base, int r0, int k0, int tid,
                                          f32x4 (&v)[TBK * ROWS / 1024]) {
#pragma unroll
  for (int h = 0; h < TBK * ROWS / 1024; h++) {
    int r, k;
    tile_coords<KCONT, TBK, ROWS>(h, tid, r, k);
    r += r0;
    k += k0;
    if constexpr (FULL) {
      const float *src = KCONT ? base + (int64_t)r * X.ld + k : base + (int64_t)k * X.ld + r;
      if constexpr (UNALIGNED) v[h] = *reinterpret_cast<const f32x4u *>(src);
      else v[h] = *reinterpret_cast<const f32x4 *>(src);
      continue;
    }
    f32x4 t = {X.pad, X.pad, X.pad, X.pad};
    if constexpr (KCONT) {
      if (r < X.R) {
        const float *src = base + (int64_t)r * X.ld + k;
        if (k + 3 < X.K) {  // four elements inside the matrix: one load, aligned or not (a 4-byte-aligned dwordx4 is
          t = *reinterpret_cast<const f32x4u *>(src);   // legal; rows of a matrix whose ld is not a multiple of 4 need it)
        } else {
#pragma unroll
          for (int c = 0; c < 4; c++)
            if (k + c < X.K) t[c] = src[c];
        }
      }
    } else {
      if (k < X.K) {
        const float *src = base + (int64_t)k * X.ld + r;
        if (r + 3 < X.R) {
          t = *reinterpret_cast<const f32x4u *>(src);
        } else {
#pragma unroll
          for (int c = 0; c < 4; c++)
            if (r + c < X.R) t[c] = src[c];
        }
      }
    }
    v[h] = t;
  }
}

// FULL tiles: the address of vector h is (per-thread pointer computed once) + (uniform offset of the tile), one
// 64-bit add per load.  Left to the generic form above, every iteration redoes the row * ld products in 64-bit
// VALU arithmetic (quarter-rate multiplies): ~300 issue cycles per tile ahead of the loads, in the MFMA stream.
template <bool KCONT, int TBK, int ROWS>
__device__ __forceinline__ void tile_ptrs(const Operand &X, int r0, int tid, const float *(&tp)[TBK * ROWS / 1024]) {
#pragma unroll
  for (int h = 0; h < TBK * ROWS / 1024; h++) {
    int r, k;
    tile_coords<KCONT, TBK, ROWS>(h, tid, r, k);
    r += r0;
    tp[h] = KCONT ? X.p + (int64_t)r * X.ld + k : X.p + (int64_t)k * X.ld + r;
  }
}
template <int NVEC, bool UNALIGNED = false>
__device__ __forceinline__ void tile_load_ptrs(const float *const (&tp)[NVEC], int64_t uoff, f32x4 (&v)[NVEC]) {
#pragma unroll
  for (int h = 0; h < NVEC; h++) {
    if constexpr (UNALIGNED) v[h] = *reinterpret_cast<const f32x4u *>(tp[h] + uoff);
    else v[h] = *reinterpret_cast<const f32x4 *>(tp[h] + uoff);
  }
}

// Registers -> LDS image T[k][r].
template <bool KCONT, int TBK, int ROWS>
__device__ __forceinline__ void tile_store(float *T, int tid, const f32x4 (&v)[TBK * ROWS / 1024]) {
  constexpr int LD = ldt<ROWS>();
#pragma unroll
  for (int h = 0; h < TBK * ROWS / 1024; h++) {
    int r, k;
    tile_coords<KCONT, TBK, ROWS>(h, tid, r, k);
    if constexpr (KCONT) {
#pragma unroll
      for (int c = 0; c < 4; c++) T[(k + c) * LD + r] = v[h][c];
    } else {
      *reinterpret_cast<f32x4 *>(&T[k * LD + r]) = v[h];
    }
  }
}

template <bool KCONT, int TBK, int ROWS>
__device__ __forceinline__ void tile_store_one(float *T, int tid, int h, const f32x4 v) {
  constexpr int LD = ldt<ROWS>();
  int r, k;
  tile_coords<KCONT, TBK, ROWS>(h, tid, r, k);
  if constexpr (KCONT) {
#pragma unroll
    for (int c = 0; c < 4; c++) T[(k + c) * LD + r] = v[c];
  } else {
    *reinterpret_cast<f32x4 *>(&T[k * LD + r]) = v;
  }
}

// XCD-aware tile order: blocks b, b+8, ... share an XCD (round-robin dispatch), so give each XCD a
// contiguous run of tiles, and walk tiles in groups of 8 tile-rows so a run shares A rows / B columns
// in that XCD's L2.  Bijective for any grid.
__device__ __forceinline__ int xcd_run_index(int nwg) {   // block -> its place in its XCD's contiguous run of nwg work items
  const int orig = blockIdx.x;
  const int xcd = orig % 8, q = nwg / 8, r = nwg % 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
}
__device__ __forceinline__ void tile_of_block(int tiles_m, int tiles_n, int &tm, int &tn) {
  const int nwg = tiles_m * tiles_n;
  const int wg = xcd_run_index(nwg);
  constexpr int GROUP = 8;
  const int per_group = GROUP * tiles_n;
  const int g = wg / per_group;
  const int first = g * GROUP;
  const int rows = (tiles_m - first) < GROUP ? (tiles_m - first) : GROUP;
  const int in_g = wg % per_group;
  tm = first + in_g % rows;
  tn = in_g / rows;
}

// C = C + (alpha * acc + beta * C), element by element as impl.cpp:226-231 writes it.
__device__ __forceinline__ float epilogue(float c_in, float acc, float alpha, float beta) {
  const float t = alpha * acc;
  const float u = beta * c_in;
  const float w = t + u;
  return c_in + w;
}

// Batch: blockIdx.y selects the problem; operands advance by their batch strides (0 = shared).
// k segments: the k loop runs nseg times over K, A and B advancing by ka / kb elements each time, all
// into the same accumulators -- one chain over nseg*K products (the conv weight update chains image
// after image, simple_conv2d_impl.cpp:645-667).
struct Batch {
  int64_t sa, sb, sc;
  int64_t ka, kb;
  int nseg;
  // CHAIN mode, where the chains start: 0 = at C's current values, 1 = at init[row], 2 = at init[row] + 0.0f (a -0
  // becomes +0), 3 = at +0.  (1-3: C is written only -- the convolution's bias / zero fill and its read-back are gone)
  int start;
  const float *init;
  // Columns >= nstore (when nonzero) are computed but not stored: B is padded with zero columns to whole tiles so that
  // every load is unchecked, while C keeps its own row length (the convolution's forward result: rows of OH*OW).
  int nstore;
  // Non-null: the kernel folds the bit patterns of |every value it stores to C| into this device word (unsigned max; a NaN
  // wins, as torch.max) -- the amax pass of a per-tensor quantisation of C that follows (quantize_module_output,
  // reference qutils.py:607-609) is then already done: the values are in registers here.  The caller zeroes the word.
  uint32_t *amax;
  // gemm_mfma16ws_kernel<..., WU = true> only (the convolution's weight update, stride 1): B is not a stored matrix but
  // the zero-padded input planes of an image.  Row r = (cb, kj, ki, c2) (c2 fastest, wu_cb channels per block) is plane
  // cb*wu_cb + c2 shifted by kj*wu_wp + ki; depth index p = oy*OW + ox (an output pixel) lies p + (p / OW) * wu_gap
  // floats into the shifted plane (wu_gap = Wp - OW; p / OW = umulhi(p, wu_magic), exact for p < 2^20, OW < 2^12).
  int wu_gap;
  uint32_t wu_magic;
  int wu_cb, wu_r, wu_s, wu_wp;
  int64_t wu_plane;
  // wu_st > 1: a strided weight update over the de-interleaved phase planes of fpemu_conv.hip (phase_planes): wu_plane is a
  // CHANNEL's stride (wu_st^2 phase planes of wu_ph floats, pitch wu_wp) and tap (kj, ki) is phase (kj % st, ki % st)
  // shifted by (kj / st) * wu_wp + ki / st.
  int wu_st;
  int64_t wu_ph;
};

__device__ __forceinline__ uint32_t abs_bits(float v) { return __float_as_uint(v) & 0x7FFFFFFFu; }
__device__ __forceinline__ uint32_t max_u32(uint32_t a, uint32_t b) { return a > b ? a : b; }

// One conditional atomic per wave: the running maximum only grows, so a wave whose maximum does not beat the value
// already published skips it (a stale read only costs a redundant atomic).  Reached by whole waves only.
__device__ __forceinline__ void publish_amax(uint32_t *word, uint32_t m) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = max_u32(m, (uint32_t)__shfl_xor((int)m, off, 64));
  if ((threadIdx.x & 63u) == 0u) {
    const uint32_t seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (m > seen) __hip_atomic_fetch_max(word, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ float chain_start(const Batch &bt, const float *C, int64_t idx, int row) {
  if (bt.start == 0) return C[idx];
  if (bt.start == 3) return 0.f;
  const float b = bt.init[row];
  return bt.start == 2 ? b + 0.0f : b;
}

// CHAIN = false: acc starts at +0 and C = C + (alpha*acc + beta*C)      (simple_gemm, impl.cpp:226-231)
// CHAIN = true : acc starts at C's current value and C = acc             (the conv kernels: MMEngine loads
//                C, continues the chain and stores it back, simple_mm_engine.cpp:18-24)
// T = tile edge: 128 (a lane owns 8 x 8 results: two 4-row and two 4-column groups 64 apart), or 64 (4 x 4) for problems
// with fewer 128 x 128 tiles than CUs -- a chain cannot be split along k, so only smaller tiles spread them.
template <bool A_KCONT, bool B_KCONT, bool CHAIN, bool FULL, int T>
__global__ __launch_bounds__(NT) void gemm_seqk_kernel(Operand A, Operand B, float *C, int64_t ldc,
                                                       int c_vec, float alpha, float beta,
                                                       int tiles_m, int tiles_n, Batch bt) {
  static_assert(T == 128 || T == 64, "tile edge");
  constexpr int H = T / 64;  // 4-row / 4-column groups per lane and dimension
  A.p += (int64_t)blockIdx.y * bt.sa;
  B.p += (int64_t)blockIdx.y * bt.sb;
  C += (int64_t)blockIdx.y * bt.sc;
  __shared__ __attribute__((aligned(16))) float As[2][BK * ldt<T>()];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * ldt<T>()];
  const int tid = threadIdx.x;
  int tm, tn;
  tile_of_block(tiles_m, tiles_n, tm, tn);
  const int m0 = tm * T, n0 = tn * T;
  const int ty = tid >> 4, tx = tid & 15;  // lane owns rows {ty*4.., 64+ty*4..} x cols {tx*4.., 64+tx*4..}
  const int K = A.K;

  f32x2 acc2[4 * H][2 * H];  // acc2[i][j] = outputs (i, 2j) and (i, 2j+1)
#pragma unroll
  for (int i = 0; i < 4 * H; i++)
#pragma unroll
    for (int j = 0; j < 4 * H; j++) {
      float v = 0.f;
      if constexpr (CHAIN) {
        const int r = m0 + (i >> 2) * 64 + ty * 4 + (i & 3), c = n0 + (j >> 2) * 64 + tx * 4 + (j & 3);
        if (r < A.R && c < B.R) v = chain_start(bt, C, (int64_t)r * ldc + c, r);
      }
      acc2[i][j >> 1][j & 1] = v;
    }

  constexpr int LDT = ldt<T>();
  f32x4 pa[BK * T / 1024], pb[BK * T / 1024];
  const float *abase = A.p, *bbase = B.p;
  tile_load<A_KCONT, FULL, BK, T>(A, abase, m0, 0, tid, pa);
  tile_load<B_KCONT, FULL, BK, T>(B, bbase, n0, 0, tid, pb);
  tile_store<A_KCONT, BK, T>(As[0], tid, pa);
  tile_store<B_KCONT, BK, T>(Bs[0], tid, pb);
  __syncthreads();

  int cur = 0;
  const int kt = (K + BK - 1) / BK, total = kt * bt.nseg;
  int nk0 = 0;  // k offset of the tile being prefetched
  for (int it = 0; it < total; it++) {
    const bool more = it + 1 < total;
    if (more) {
      nk0 += BK;
      if (nk0 >= K) {  // next k segment
        nk0 = 0;
        abase += bt.ka;
        bbase += bt.kb;
      }
      tile_load<A_KCONT, FULL, BK, T>(A, abase, m0, nk0, tid, pa);
      tile_load<B_KCONT, FULL, BK, T>(B, bbase, n0, nk0, tid, pb);
    }
    const float *as = As[cur], *bs = Bs[cur];
#pragma unroll 2
    for (int kk = 0; kk < BK; kk++) {
      float a[4 * H];
      f32x2 b2[2 * H];
#pragma unroll
      for (int h = 0; h < H; h++) {
        const f32x4 av = *reinterpret_cast<const f32x4 *>(&as[kk * LDT + h * 64 + ty * 4]);
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(&bs[kk * LDT + h * 64 + tx * 4]);
#pragma unroll
        for (int e = 0; e < 4; e++) a[4 * h + e] = av[e];
        b2[2 * h] = f32x2{bv[0], bv[1]};
        b2[2 * h + 1] = f32x2{bv[2], bv[3]};
      }
      // explicit 2-wide form so that every multiply and every add is a packed instruction
      // (v_pk_mul_f32 / v_pk_add_f32: two lanes of the chain per issue slot; left to the SLP
      // vectoriser half of the adds stayed scalar)
#pragma unroll
      for (int i = 0; i < 4 * H; i++) {
        const f32x2 ai = {a[i], a[i]};
#pragma unroll
        for (int j = 0; j < 2 * H; j++) {
          const f32x2 p = ai * b2[j];      // rounded products: values of their own (contraction is off)
          acc2[i][j] = acc2[i][j] + p;     // k ascending, one add per step
        }
      }
    }
    if (more) {
      tile_store<A_KCONT, BK, T>(As[cur ^ 1], tid, pa);
      tile_store<B_KCONT, BK, T>(Bs[cur ^ 1], tid, pb);
    }
    __syncthreads();
    cur ^= 1;
  }

  const int M = A.R, N = B.R;
  uint32_t mx = 0u;  // max |stored value| (Batch.amax)
  // CHAIN: the chains started from these addresses; opaque copies keep that arithmetic from staying alive across the k
  // loop (see gemm_mfma_body's epilogue: 136-140 VGPRs -> the non-chain kernel's 113-118, four waves per SIMD again)
  int64_t ldc_e = ldc;
  int ty_e = ty, tx_e = tx;
  if constexpr (CHAIN) asm volatile("" : "+s"(ldc_e), "+v"(ty_e), "+v"(tx_e));
#pragma unroll
  for (int ih = 0; ih < H; ih++)
#pragma unroll
    for (int ii = 0; ii < 4; ii++) {
      const int r = m0 + ih * 64 + ty_e * 4 + ii;
      if (r >= M) continue;
#pragma unroll
      for (int jh = 0; jh < H; jh++) {
        const int c = n0 + jh * 64 + tx_e * 4;
        float *dst = C + (int64_t)r * ldc_e + c;
        if constexpr (CHAIN) {
#pragma unroll
          for (int jj = 0; jj < 4; jj++)
            if (c + jj < N && (bt.nstore == 0 || c + jj < bt.nstore)) {
              const float o = acc2[ih * 4 + ii][(jh * 4 + jj) >> 1][jj & 1];
              dst[jj] = o;
              mx = max_u32(mx, abs_bits(o));
            }
        } else if (c_vec && c + 3 < N) {
          f32x4 ci = *reinterpret_cast<const f32x4 *>(dst), o;
#pragma unroll
          for (int jj = 0; jj < 4; jj++) {
            o[jj] = epilogue(ci[jj], acc2[ih * 4 + ii][(jh * 4 + jj) >> 1][jj & 1], alpha, beta);
            mx = max_u32(mx, abs_bits(o[jj]));
          }
          *reinterpret_cast<f32x4 *>(dst) = o;
        } else {
#pragma unroll
          for (int jj = 0; jj < 4; jj++)
            if (c + jj < N) {
              const float o = epilogue(dst[jj], acc2[ih * 4 + ii][(jh * 4 + jj) >> 1][jj & 1], alpha, beta);
              dst[jj] = o;
              mx = max_u32(mx, abs_bits(o));
            }
        }
      }
    }
  if (bt.amax != nullptr) publish_amax(bt.amax, mx);
}

// ---------------------------------------------------------------------------------------------
// MFMA variant.  4 waves per block in a 2 x 2 grid; wave w owns a (32 TW) x (32 TW) quadrant as TW x TW
// MFMA 32x32 tiles: TW = 2 -> 128 x 128 block tiles (large problems), TW = 1 -> 64 x 64 block tiles
// (problems that would not fill the chip with 128 x 128 tiles: a chain cannot be split along k, so small
// M x N needs small tiles to spread over CUs).  v_mfma_f32_32x32x2_f32: lane l supplies
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; D[reg]: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
// ---------------------------------------------------------------------------------------------
#ifndef FPEMU_GEMM_PTRS
#define FPEMU_GEMM_PTRS 1
#endif
#ifndef FPEMU_GEMM_AHEAD2
#define FPEMU_GEMM_AHEAD2 1
#endif
#ifndef FPEMU_GEMM_PIPE
#define FPEMU_GEMM_PIPE 1
#endif
#ifndef FPEMU_CONV_T128_MIN
#define FPEMU_CONV_T128_MIN 512   // implicit convolution: 128 x 128 tiles when at least this many exist
#endif
#ifndef FPEMU_GEMM_WAVES
#define FPEMU_GEMM_WAVES 2
#endif
// Implicit convolution (fpemu_conv.hip): the GEMM's N index is a position q = oy * Wp + ox' of the PADDED-width
// output grid of one image, so row (c, kj, ki) of the "gathered" operand is simply the padded image plane c
// shifted by kj * Wp + ki -- contiguous memory, nothing to materialise.  The k loop walks segments (channel block,
// kj, ki) of 64 channels each: B advances by 1 (next ki), Wp - (S-1) (next kj) or the channel-block step.
// Columns with ox' >= OW (or q >= nvalid) are garbage positions: computed and dropped.  Accumulators start at
// bias[row] (or +0) and C = acc, as in CHAIN mode.
#ifdef FPEMU_GEMM_TRACE  // tuning builds (tools/conv_trace.py): per-workgroup phase stamps (100 MHz) of the conv GEMM
__device__ unsigned long long *g_conv_phase = nullptr;
extern "C" int fpemu_test_conv_phase_(void *ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_phase), &ptr, sizeof(ptr)); }
#define CONV_STAMP(var) [[maybe_unused]] const unsigned long long var = CONV ? wall_clock64() : 0ull
#else
#define CONV_STAMP(var) do { } while (0)
#endif
struct ConvMap {
  const float *bias;  // nullptr: start at +0
  int S, R, Wp;       // filter taps per row / rows, padded row pitch
  int dk;             // +1: taps of a filter row walk right (forward); -1: left (backward-data: ki ascending = ox descending)
  int OW, P, nvalid;  // valid columns per output row, output pixels per channel (C's row pitch), OH * Wp
  // Strided forward convolution (st > 1): the padded input is de-interleaved into its st * st phase planes, channel c of
  // an image = [phase (py, px)][Hh][Wp] (Wp = the phase planes' pitch, PH = Hh * Wp floats each).  Tap (kj, ki) of output
  // (oy, ox) reads padded pixel (oy*st + kj, ox*st + ki) = phase (kj % st, ki % st) at (oy + kj / st, ox + ki / st): a
  // stride-1 walk again, the plane shifted by ((kj % st) * st + ki % st) * PH + (kj / st) * Wp + ki / st.
  int st;             // 0 / 1: stride 1 (the three pointer steps above)
  int64_t PH;
};

template <bool A_KCONT, bool B_KCONT, bool CHAIN, bool FULL, int TW, bool CONV>
__device__ __forceinline__ void gemm_mfma_body(Operand A, Operand B, float *C, int64_t ldc, float alpha, float beta,
                                               int tiles_m, int tiles_n, Batch bt, const ConvMap &cm) {
  constexpr int TB = 64 * TW;  // block tile edge
  constexpr int LDT = ldt<TB>();
  constexpr int NV = MBK * TB / 1024;
  A.p += (int64_t)blockIdx.y * bt.sa;
  B.p += (int64_t)blockIdx.y * bt.sb;
  C += (int64_t)blockIdx.y * bt.sc;
  __shared__ __attribute__((aligned(16))) float As[2][MBK * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[2][MBK * LDT];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int tm, tn;
  tile_of_block(tiles_m, tiles_n, tm, tn);
  const int m0 = tm * TB, n0 = tn * TB;
  const int wm = (w >> 1) * 32 * TW, wn = (w & 1) * 32 * TW;
  const int li = lane & 31, lk = lane >> 5;
  const int K = A.K;

  CONV_STAMP(ts0);
  f32x16 acc[TW][TW];
#pragma unroll
  for (int i = 0; i < TW; i++)
#pragma unroll
    for (int j = 0; j < TW; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        acc[i][j][r] = 0.f;
        if constexpr (CONV) {
          if (cm.bias != nullptr) acc[i][j][r] = cm.bias[m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk];
        } else if constexpr (CHAIN) {
          const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          const int col = n0 + wn + j * 32 + li;
          if (row < A.R && col < B.R) acc[i][j][r] = chain_start(bt, C, (int64_t)row * ldc + col, row);
        }
      }

  f32x4 pa[NV], pb[NV];
  const float *abase = A.p, *bbase = B.p;
  [[maybe_unused]] int wki = 0, wkj = 0;  // CONV: filter tap of the current k segment
  [[maybe_unused]] int wri = 0, wqi = 0, wrj = 0, wqj = 0;  // CONV, st > 1: ki % st, ki / st, kj % st, kj / st
  [[maybe_unused]] const float *cbase = B.p;                // CONV, st > 1: the current channel block's planes
  // next k segment: A by its segment stride; B by its segment stride, or (CONV) along the filter-tap walk
  auto next_segment = [&]() {
    abase += bt.ka;
    if constexpr (CONV) {
      if (cm.st > 1) {  // uniform: phase planes (scalar counters, no division)
        if (++wri == cm.st) { wri = 0; wqi++; }
        if (++wki == cm.S) {
          wki = 0; wri = 0; wqi = 0;
          if (++wrj == cm.st) { wrj = 0; wqj++; }
          if (++wkj == cm.R) {
            wkj = 0; wrj = 0; wqj = 0;
            cbase += bt.kb;
          }
        }
        bbase = cbase + (int64_t)(wrj * cm.st + wri) * cm.PH + (int64_t)wqj * cm.Wp + wqi;
        return;
      }
      int64_t inc = cm.dk;
      if (++wki == cm.S) {
        wki = 0;
        inc = (int64_t)cm.Wp - cm.dk * (cm.S - 1);
        if (++wkj == cm.R) {
          wkj = 0;
          inc = bt.kb - (int64_t)(cm.R - 1) * cm.Wp - cm.dk * (cm.S - 1);
        }
      }
      bbase += inc;
    } else {
      bbase += bt.kb;
    }
  };
  const float *tpa[NV], *tpb[NV];  // FULL: per-thread pointers of the vectors of tile (segment 0, k0 = 0)
  if constexpr (FULL) {
    tile_ptrs<A_KCONT, MBK, TB>(A, m0, tid, tpa);
    tile_ptrs<B_KCONT, MBK, TB>(B, n0, tid, tpb);
  }
  tile_load<A_KCONT, FULL, MBK, TB>(A, abase, m0, 0, tid, pa);
  tile_load<B_KCONT, FULL, MBK, TB, CONV>(B, bbase, n0, 0, tid, pb);
  tile_store<A_KCONT, MBK, TB>(As[0], tid, pa);
  tile_store<B_KCONT, MBK, TB>(Bs[0], tid, pb);
  __syncthreads();

  CONV_STAMP(ts1);  // first tile in LDS
  [[maybe_unused]] int cur = 0;
  const int kt = (K + MBK - 1) / MBK, total = kt * bt.nseg;
  int nk0 = 0;  // k offset of the tile being prefetched
#if FPEMU_GEMM_PIPE
  float fa[8][TW], fb[8][TW];
#pragma unroll
  for (int t = 0; t < 4; t++) {  // fragments of the first tile's first AHEAD steps
#pragma unroll
    for (int i = 0; i < TW; i++) fa[t][i] = As[0][lk * LDT + wm + li + t * 2 * LDT + i * 32];
#pragma unroll
    for (int j = 0; j < TW; j++) fb[t][j] = Bs[0][lk * LDT + wn + li + t * 2 * LDT + j * 32];
  }
#endif
#if FPEMU_GEMM_PIPE
  // Fragments are read AHEAD steps before the MFMAs that consume them, into a ring of RING register sets.
  // The iteration's one barrier sits after step NSTEP - AHEAD - 1: by then every wave has issued its last read
  // of this LDS buffer (the fragments of the last AHEAD steps are in registers) and has stored its share of
  // the next tile into the other buffer, so the last AHEAD steps already read the NEXT tile's first fragments
  // and the MFMA stream crosses the iteration boundary without a barrier or an exposed LDS round trip.
  // (Past the last tile those reads fetch unused values from the other buffer: no branch in the stream.)
  // The next tile's LDS stores go out one vector per step over the steps before the barrier (each waits only
  // for its own global load).  MORE is a compile-time flag -- the last iteration is peeled -- so there is no
  // branch inside the stream and the compiler's waits stay exact.
  constexpr int NSTEP = MBK / 2, AHEAD = 4, RING = 8, S0 = NSTEP - AHEAD - 2 * NV;
  static_assert(NSTEP % RING == 0 && AHEAD < RING && S0 >= 0, "ring slots are compile-time across iterations");
#if FPEMU_GEMM_AHEAD2
  // Tiles are fetched TWO ahead (register stage = tile parity): a global load gets a whole iteration plus the
  // steps up to its store (> 5000 cycles) to land, instead of the 1000-2800 of a one-tile distance.
  f32x4 qa[NV], qb[NV];  // stage 1 (stage 0 = pa / pb)
  int fetched = 1;       // tiles fetched so far (tile 0 came through pa / pb in the prologue)
  auto fetch = [&](f32x4 (&ra)[NV], f32x4 (&rb)[NV]) {
    nk0 += MBK;
    if (nk0 >= K) {  // next k segment
      nk0 = 0;
      next_segment();
    }
    if constexpr (FULL && FPEMU_GEMM_PTRS) {
      tile_load_ptrs<NV>(tpa, (abase - A.p) + (A_KCONT ? (int64_t)nk0 : (int64_t)nk0 * A.ld), ra);
      tile_load_ptrs<NV, CONV>(tpb, (bbase - B.p) + (B_KCONT ? (int64_t)nk0 : (int64_t)nk0 * B.ld), rb);
    } else {
      tile_load<A_KCONT, FULL, MBK, TB>(A, abase, m0, nk0, tid, ra);
      tile_load<B_KCONT, FULL, MBK, TB, CONV>(B, bbase, n0, nk0, tid, rb);
    }
    fetched++;
  };
  if (total > 1) fetch(qa, qb);  // tile 1 -> stage 1
  // tile t = it + PAR is multiplied from LDS buffer PAR; tile t+1 sits in stage PAR^1 and goes into LDS buffer
  // PAR^1 during this iteration; tile t+2 is fetched into stage PAR (whose tile t went into LDS an iteration ago)
  auto iter_body = [&](auto par_c, auto steady_c, bool store_rt, bool fetch_rt) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    const bool store_next = STEADY || store_rt, fetch_next = STEADY || fetch_rt;
    f32x4(&sa_)[NV] = PAR ? pa : qa;  // stage PAR^1: the tile to store
    f32x4(&sb_)[NV] = PAR ? pb : qb;
    if (fetch_next) {
      if constexpr (PAR) fetch(qa, qb);
      else fetch(pa, pb);
    }
    const float *as = As[PAR] + lk * LDT + wm + li, *bs = Bs[PAR] + lk * LDT + wn + li;
    const float *an = As[PAR ^ 1] + lk * LDT + wm + li, *bn = Bs[PAR ^ 1] + lk * LDT + wn + li;
#pragma unroll
    for (int st = 0; st < NSTEP; st++) {
      const int t = st + AHEAD, slot = t % RING;
      const float *sa = t < NSTEP ? as : an, *sb = t < NSTEP ? bs : bn;
#pragma unroll
      for (int i = 0; i < TW; i++) fa[slot][i] = sa[(t % NSTEP) * 2 * LDT + i * 32];
#pragma unroll
      for (int j = 0; j < TW; j++) fb[slot][j] = sb[(t % NSTEP) * 2 * LDT + j * 32];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TW; i++)
#pragma unroll
        for (int j = 0; j < TW; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st % RING][i], fb[st % RING][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (st >= S0 && st < S0 + 2 * NV && store_next) {
        const int idx = st - S0;
        if (idx & 1) tile_store_one<B_KCONT, MBK, TB>(Bs[PAR ^ 1], tid, idx >> 1, sb_[idx >> 1]);
        else tile_store_one<A_KCONT, MBK, TB>(As[PAR ^ 1], tid, idx >> 1, sa_[idx >> 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (st == NSTEP - AHEAD - 1) __syncthreads();
    }
  };
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  int it = 0;
  for (; it + 3 < total; it += 2) {
    iter_body(c0{}, std::true_type{}, true, true);
    iter_body(c1{}, std::true_type{}, true, true);
  }
  for (; it < total; it += 2) {
    iter_body(c0{}, std::false_type{}, it + 1 < total, fetched < total);
    if (it + 1 >= total) break;
    iter_body(c1{}, std::false_type{}, it + 2 < total, fetched < total);
  }
#else
  auto iter_body = [&](auto more_c) {
    constexpr bool MORE = decltype(more_c)::value;
    if constexpr (MORE) {
      nk0 += MBK;
      if (nk0 >= K) {  // next k segment
        nk0 = 0;
        next_segment();
      }
      if constexpr (FULL && FPEMU_GEMM_PTRS) {  // uniform offsets (scalar arithmetic), one 64-bit add per vector
        tile_load_ptrs<NV>(tpa, (abase - A.p) + (A_KCONT ? (int64_t)nk0 : (int64_t)nk0 * A.ld), pa);
        tile_load_ptrs<NV>(tpb, (bbase - B.p) + (B_KCONT ? (int64_t)nk0 : (int64_t)nk0 * B.ld), pb);
      } else {
        tile_load<A_KCONT, FULL, MBK, TB>(A, abase, m0, nk0, tid, pa);
        tile_load<B_KCONT, FULL, MBK, TB>(B, bbase, n0, nk0, tid, pb);
      }
    }
    const float *as = As[cur] + lk * LDT + wm + li, *bs = Bs[cur] + lk * LDT + wn + li;
    const float *an = As[cur ^ 1] + lk * LDT + wm + li, *bn = Bs[cur ^ 1] + lk * LDT + wn + li;
#pragma unroll
    for (int st = 0; st < NSTEP; st++) {
      const int t = st + AHEAD, slot = t % RING;
      const float *sa = t < NSTEP ? as : an, *sb = t < NSTEP ? bs : bn;
#pragma unroll
      for (int i = 0; i < TW; i++) fa[slot][i] = sa[(t % NSTEP) * 2 * LDT + i * 32];
#pragma unroll
      for (int j = 0; j < TW; j++) fb[slot][j] = sb[(t % NSTEP) * 2 * LDT + j * 32];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TW; i++)
#pragma unroll
        for (int j = 0; j < TW; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st % RING][i], fb[st % RING][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (MORE) {
        if (st >= S0 && st < S0 + 2 * NV) {
          const int idx = st - S0;
          if (idx & 1) tile_store_one<B_KCONT, MBK, TB>(Bs[cur ^ 1], tid, idx >> 1, pb[idx >> 1]);
          else tile_store_one<A_KCONT, MBK, TB>(As[cur ^ 1], tid, idx >> 1, pa[idx >> 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (st == NSTEP - AHEAD - 1) __syncthreads();
    }
    cur ^= 1;
  };
  for (int it = 0; it + 1 < total; it++) iter_body(std::true_type{});
  if (total > 0) iter_body(std::false_type{});
#endif
#else
  for (int it = 0; it < total; it++) {
    const bool more = it + 1 < total;
    if (more) {
      nk0 += MBK;
      if (nk0 >= K) {  // next k segment
        nk0 = 0;
        next_segment();
      }
      if constexpr (FULL && FPEMU_GEMM_PTRS) {  // uniform offsets (scalar arithmetic), one 64-bit add per vector
        tile_load_ptrs<NV>(tpa, (abase - A.p) + (A_KCONT ? (int64_t)nk0 : (int64_t)nk0 * A.ld), pa);
        tile_load_ptrs<NV>(tpb, (bbase - B.p) + (B_KCONT ? (int64_t)nk0 : (int64_t)nk0 * B.ld), pb);
      } else {
        tile_load<A_KCONT, FULL, MBK, TB>(A, abase, m0, nk0, tid, pa);
        tile_load<B_KCONT, FULL, MBK, TB>(B, bbase, n0, nk0, tid, pb);
      }
    }
    const float *as = As[cur] + lk * LDT + wm + li, *bs = Bs[cur] + lk * LDT + wn + li;
    // operand fragments one k-pair ahead of the MFMAs that consume them (register double buffer), so the
    // LDS latency of step s+1 runs under the MFMAs of step s
    float a[2][TW], b[2][TW];
#pragma unroll
    for (int i = 0; i < TW; i++) a[0][i] = as[i * 32];
#pragma unroll
    for (int j = 0; j < TW; j++) b[0][j] = bs[j * 32];
#pragma unroll
    for (int st = 0; st < MBK / 2; st++) {
      const int c = st & 1;
      if (st + 1 < MBK / 2) {
#pragma unroll
        for (int i = 0; i < TW; i++) a[c ^ 1][i] = as[(st + 1) * 2 * LDT + i * 32];
#pragma unroll
        for (int j = 0; j < TW; j++) b[c ^ 1][j] = bs[(st + 1) * 2 * LDT + j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the reads of step st+1 ahead of the MFMAs of step st
#pragma unroll
      for (int i = 0; i < TW; i++)
#pragma unroll
        for (int j = 0; j < TW; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // The prefetched tile goes into the OTHER LDS buffer, which nobody reads during this iteration
      // (every wave passed the barrier that ended the previous one), so its stores are issued in the
      // shadow of the MFMAs, three quarters of the way through, instead of after them.
      if (st == (3 * MBK) / 8 - 1 && more) {
        tile_store<A_KCONT, MBK, TB>(As[cur ^ 1], tid, pa);
        tile_store<B_KCONT, MBK, TB>(Bs[cur ^ 1], tid, pb);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    cur ^= 1;
  }

#endif

  CONV_STAMP(ts2);  // k loop done
  [[maybe_unused]] const int M = A.R, N = B.R;
  uint32_t mx = 0u;  // max |stored value| (Batch.amax)
  if constexpr (CONV) {
#pragma unroll
    for (int j = 0; j < TW; j++) {
      const int q = n0 + wn + j * 32 + li;          // position in the padded-width output grid
      const int oy = q / cm.Wp, ox = q - oy * cm.Wp;
      if (q < cm.nvalid && ox < cm.OW) {
        float *dst = C + (int64_t)oy * cm.OW + ox;
#pragma unroll
        for (int i = 0; i < TW; i++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            dst[(int64_t)row * cm.P] = acc[i][j][r];
            mx = max_u32(mx, abs_bits(acc[i][j][r]));
          }
      }
    }
#ifdef FPEMU_GEMM_TRACE
    if (threadIdx.x == 0 && g_conv_phase != nullptr) {
      const unsigned long long idx = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;
      g_conv_phase[4 * idx] = ts0;
      g_conv_phase[4 * idx + 1] = ts1;
      g_conv_phase[4 * idx + 2] = ts2;
      g_conv_phase[4 * idx + 3] = wall_clock64();
    }
#endif
  } else if constexpr (CHAIN) {
    // The chains started from these very addresses (chain_start above the k loop).  Left to itself the compiler keeps
    // that address arithmetic alive across the loop for reuse here, which is what pushed the 128 x 128 kernels past
    // 256 VGPRs (four dwords spilled before the loop and reloaded every iteration: 126 against 142 TFLOP/s at 8192^3).
    // Opaque copies of the leading dimension and the lane's coordinates make the epilogue compute its own.
    int64_t ldc_e = ldc;
    int li_e = li, lk_e = lk;
    asm volatile("" : "+s"(ldc_e), "+v"(li_e), "+v"(lk_e));
#pragma unroll
    for (int i = 0; i < TW; i++)
#pragma unroll
      for (int j = 0; j < TW; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk_e;
          const int col = n0 + wn + j * 32 + li_e;
          if ((FULL || (row < M && col < N)) && (bt.nstore == 0 || col < bt.nstore)) {
            C[(int64_t)row * ldc_e + col] = acc[i][j][r];
            mx = max_u32(mx, abs_bits(acc[i][j][r]));
          }
        }
  } else {
    // per 32x32 MFMA tile: its 16 reads of C first (independent), then the 16 read-modify-writes -- one
    // memory round trip per tile instead of 16, without holding all of C in registers
#pragma unroll
    for (int i = 0; i < TW; i++)
#pragma unroll
      for (int j = 0; j < TW; j++) {
        f32x16 cin;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          const int col = n0 + wn + j * 32 + li;
          cin[r] = (FULL || (row < M && col < N)) ? C[(int64_t)row * ldc + col] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          const int col = n0 + wn + j * 32 + li;
          if (FULL || (row < M && col < N)) {
            const float o = epilogue(cin[r], acc[i][j][r], alpha, beta);
            C[(int64_t)row * ldc + col] = o;
            mx = max_u32(mx, abs_bits(o));
          }
        }
      }
  }
  if (bt.amax != nullptr) publish_amax(bt.amax, mx);
}

template <bool A_KCONT, bool B_KCONT, bool CHAIN, bool FULL, int TW>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(FPEMU_GEMM_WAVES)))
void gemm_mfma_kernel(Operand A, Operand B, float *C, int64_t ldc, float alpha, float beta, int tiles_m, int tiles_n,
                      Batch bt) {
  gemm_mfma_body<A_KCONT, B_KCONT, CHAIN, FULL, TW, false>(A, B, C, ldc, alpha, beta, tiles_m, tiles_n, bt, ConvMap{});
}

// Up to kGroupMax chain GEMMs that share M, N, the leading dimensions of B and C and the batch, but not their depth, in
// ONE launch (blockIdx.z = problem): the phases of a strided backward-data convolution are four such problems of 17-46 us
// each, and a launch of a few hundred short workgroups spends ~8 us ramping up and draining.  Whole tiles only (the
// unchecked kernel), A row-major M x K_g (k contiguous), B row-major K_g x N, chains start at +0.
constexpr int kGroupMax = 4;
struct GroupTab {
  const float *a[kGroupMax];
  const float *b[kGroupMax];
  float *c[kGroupMax];
  int k[kGroupMax];
};
template <int TW>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(FPEMU_GEMM_WAVES)))
void gemm_mfma_group_kernel(Operand A, Operand B, int64_t ldc, int tiles_m, int tiles_n, Batch bt, GroupTab gt) {
  const int g = blockIdx.z;   // uniform: scalar loads from the kernel arguments
  const int k = gt.k[g];
  A.p = gt.a[g];
  A.K = k;
  A.ld = k;
  B.p = gt.b[g];
  B.K = k;
  bt.sb = (int64_t)k * B.ld;
  gemm_mfma_body<true, false, true, true, TW, false>(A, B, gt.c[g], ldc, 1.0f, 0.0f, tiles_m, tiles_n, bt, ConvMap{});
}

// The implicit convolution's form: A = packed weights [M][segments * 64] (k contiguous), B = padded image planes
// ([64 channels][plane], positions contiguous), whole tiles only.
#ifdef FPEMU_GEMM_TRACE  // tuning builds (tools/conv_trace.py): per-workgroup start / end time (100 MHz) and hardware id
__device__ unsigned long long *g_conv_trace = nullptr;
#endif
template <int TW>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(FPEMU_GEMM_WAVES)))
void gemm_mfma_conv_kernel(Operand A, Operand B, float *C, int tiles_m, int tiles_n, Batch bt, ConvMap cm) {
#ifdef FPEMU_GEMM_TRACE
  const unsigned long long t0 = wall_clock64();
#endif
  gemm_mfma_body<true, false, true, true, TW, true>(A, B, C, 0, 1.0f, 0.0f, tiles_m, tiles_n, bt, cm);
#ifdef FPEMU_GEMM_TRACE
  if (threadIdx.x == 0 && g_conv_trace != nullptr) {
    const unsigned long long idx = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    g_conv_trace[3 * idx] = t0;
    g_conv_trace[3 * idx + 1] = wall_clock64();
    g_conv_trace[3 * idx + 2] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
}
#ifdef FPEMU_GEMM_TRACE
extern "C" int fpemu_test_conv_trace_(void *ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_trace), &ptr, sizeof(ptr));
}
#endif

typedef void (*mfma_fn_t)(Operand, Operand, float *, int64_t, float, float, int, int, Batch);

// ---------------------------------------------------------------------------------------------
// Latency-bound chains: problems with few outputs and a long k (the conv weight update: K_out x C*R*S outputs,
// each a chain over N*OH*OW products) cannot fill the chip with 32x32 accumulator tiles, and every wave then runs
// ONE dependent MFMA chain: a 32x32x2 step retires 2 k per 64 cycles.  v_mfma_f32_16x16x4_f32 is a k-ascending
// fmaf chain as well (tools/mfma_order.hip: bit-identical to the host's chain) and retires 4 k per 32 cycles:
// a quarter of the latency per k, on four times as many waves.  32 x 32 block tiles, 4 waves in a 2 x 2 grid,
// one 16x16 accumulator tile per wave; 128-deep k tiles (32 dependent steps per barrier).
// Lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; D[reg]: row = 4*(l>>4) + reg, col = l&15.
// ---------------------------------------------------------------------------------------------
// LDS layout of a [SK][SB] tile: element (k, col) at k * SLD + (col ^ swz(k)), SLD = 48, swz(k) = 4 * ((k >> 2) & 7).
// Bank = address mod 32.  Fragment read of step s (lanes li < 16, lk < 4: k = 4s + lk, col = w + li): k * 48 = lk * 16
// (mod 32) and the XOR, the same for every lane of the instruction, permutes an aligned block of 16 columns -- the 32
// lanes of each half-wave hit 32 banks.  Transposing store of a k-contiguous operand (lanes: 4 rows x 8 groups of four
// k per half-wave; one of the four k per instruction): k * 48 = 16c (mod 32) for every group, and the XOR spreads the
// groups over the eight aligned blocks of four columns -- 32 banks again.  (With the padded layout SLD = SB + 4 these
// stores were 4-way and the reads 2-way conflicts, and the LDS pipe -- 1024 + 512 cycles per 1024-cycle tile --
// was what the chain waited for.)
constexpr int SB = 32, SK = 128, SLD = 48, SNV = SK * SB / (4 * NT);
__device__ __forceinline__ int small_swz(int k) { return ((k >> 2) & 7) << 2; }

template <bool KCONT>
__device__ __forceinline__ void small_coords(int h, int tid, int &r, int &k) {
  if constexpr (KCONT) {  // thread -> (row, 4 consecutive k): 8 threads per row, 32 k per pass
    r = tid >> 3;
    k = h * 32 + (tid & 7) * 4;
  } else {                // thread -> (k, 4 consecutive rows): 8 threads per k, 32 k per pass
    k = h * 32 + (tid >> 3);
    r = (tid & 7) * 4;
  }
}

// whole: (not FULL) this tile lies inside the operand and the operand is vector-aligned -- uniform over the
// workgroup, so the chain's steady state runs the unchecked load and only a segment's ragged last k tile (the conv
// weight update: OH*OW = 3136 = 24.5 tiles per image) takes the bounds-checked one.
template <bool KCONT, bool FULL>
__device__ __forceinline__ f32x4 small_load_one(const Operand &X, const float *base, int r0, int k0, int tid, int h,
                                                bool whole = false) {
  int r, k;
  small_coords<KCONT>(h, tid, r, k);
  r += r0;
  k += k0;
  const float *src = KCONT ? base + (int64_t)r * X.ld + k : base + (int64_t)k * X.ld + r;
  if (FULL || whole) return *reinterpret_cast<const f32x4 *>(src);
  f32x4 t = {X.pad, X.pad, X.pad, X.pad};
  const bool in_outer = KCONT ? (r < X.R) : (k < X.K);
  const int inner = KCONT ? k : r, inner_end = KCONT ? X.K : X.R;
  if (in_outer) {
    if (inner + 3 < inner_end) {
      t = *reinterpret_cast<const f32x4u *>(src);
    } else {
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (inner + c < inner_end) t[c] = src[c];
    }
  }
  return t;
}

template <bool KCONT>
__device__ __forceinline__ void small_store_one(float *T, int tid, int h, const f32x4 v) {
  int r, k;
  small_coords<KCONT>(h, tid, r, k);
  const int col = r ^ small_swz(k);  // KCONT: k is a multiple of 4, the four k share the swizzle; else r is, and stays one
  if constexpr (KCONT) {
#pragma unroll
    for (int c = 0; c < 4; c++) T[(k + c) * SLD + col] = v[c];
  } else {
    *reinterpret_cast<f32x4 *>(&T[k * SLD + col]) = v;
  }
}

// Where the next tile to fetch lies (k segments: the conv weight update chains image after image).
struct FetchCursor {
  const float *a, *b;
  int k0, idx, total, K;
  int64_t ka, kb;
  __device__ __forceinline__ bool more() const { return idx < total; }
  __device__ __forceinline__ void advance() {
    idx++;
    if (idx >= total) return;  // stay on the last tile: a fetch past the end re-reads it (and is never stored)
    k0 += SK;
    if (k0 >= K) {
      k0 = 0;
      a += ka;
      b += kb;
    }
  }
};

template <bool A_KCONT, bool B_KCONT, bool CHAIN, bool FULL>
__global__ __launch_bounds__(NT) void gemm_mfma16_kernel(Operand A, Operand B, float *C, int64_t ldc,
                                                         float alpha, float beta, int tiles_m, int tiles_n,
                                                         Batch bt) {
  A.p += (int64_t)blockIdx.y * bt.sa;
  B.p += (int64_t)blockIdx.y * bt.sb;
  C += (int64_t)blockIdx.y * bt.sc;
  __shared__ __attribute__((aligned(16))) float As[2][SK * SLD];
  __shared__ __attribute__((aligned(16))) float Bs[2][SK * SLD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int tm, tn;
  tile_of_block(tiles_m, tiles_n, tm, tn);
  const int m0 = tm * SB, n0 = tn * SB;
  const int wm = (w >> 1) * 16, wn = (w & 1) * 16;
  const int li = lane & 15, lk = lane >> 4;
  const int K = A.K, M = A.R, N = B.R;
  const int col = n0 + wn + li;

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if constexpr (CHAIN) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = m0 + wm + 4 * lk + r;
      if (row < M && col < N) acc[r] = chain_start(bt, C, (int64_t)row * ldc + col, row);
    }
  }
  // this workgroup's rows / columns are whole tiles of vector-aligned operands: only k can be ragged
  const bool mn_whole = !FULL && A.vec && B.vec && m0 + SB <= M && n0 + SB <= N;

  // A wave's MFMAs form ONE dependent chain (~40 cycles a step), and a wave issues in order: whatever else it
  // has to do -- LDS reads of the next fragments, the next tile's LDS stores, the global loads of the tile after
  // that -- is cut into slices and placed BETWEEN the chain's steps, where the issue slot would idle anyway.
  // Tiles are fetched two ahead of the one being multiplied (two register stages, stage = tile parity) and
  // stored into LDS a tile and a half later: a tile takes ~1300 cycles, a global load up to ~2000.
  const int kt = (K + SK - 1) / SK, total = kt * bt.nseg;
  FetchCursor cur{A.p, B.p, 0, 0, total, K, bt.ka, bt.kb};
  f32x4 ra[2][SNV], rb[2][SNV];
#pragma unroll
  for (int sgi = 0; sgi < 2; sgi++) {  // tiles 0 and 1
    if (cur.more()) {
#pragma unroll
      for (int h = 0; h < SNV; h++) {
        ra[sgi][h] = small_load_one<A_KCONT, FULL>(A, cur.a, m0, cur.k0, tid, h);
        rb[sgi][h] = small_load_one<B_KCONT, FULL>(B, cur.b, n0, cur.k0, tid, h);
      }
      cur.advance();
    }
  }
  if (total > 0) {
#pragma unroll
    for (int h = 0; h < SNV; h++) {
      small_store_one<A_KCONT>(As[0], tid, h, ra[0][h]);
      small_store_one<B_KCONT>(Bs[0], tid, h, rb[0][h]);
    }
  }
  __syncthreads();

  constexpr int Q = SK / 16;  // MFMA steps per quarter of a tile
  static_assert(Q == 2 * SNV, "one LDS-store slice and every other step one global-load slice per chain step");
  // fragment of step s: row 4s + lk, column (w + li) ^ 4 * (s & 7); s & 7 = the step within its quarter (Q = 8)
  static_assert(Q == 8, "the swizzle period is one quarter");
  int aoff[Q], boff[Q];
#pragma unroll
  for (int st = 0; st < Q; st++) {
    aoff[st] = lk * SLD + ((wm + li) ^ (st << 2));
    boff[st] = lk * SLD + ((wn + li) ^ (st << 2));
  }
  // Tile t = it + PAR, LDS buffer PAR, in four quarters of 8 chain steps; after each step one slice of other work:
  //   q0, q1  fragments of the next quarter; the global loads of tile t+2 into register stage PAR
  //   q2      fragments of q3; tile t+1 (fetched a tile and a half ago into stage PAR^1) into LDS buffer PAR^1
  //   -- the tile's ONE barrier: every wave holds its last fragments of buffer PAR and has stored its share of
  //      buffer PAR^1 --
  //   q3      the first fragments of tile t+1 from buffer PAR^1
  // so no LDS latency is exposed at the barrier, and the next tile's chain continues without a gap.
  // STEADY: the store and the fetch both happen (known at compile time: no branches inside the chain, so the
  // compiler's waits stay exact); the last tiles of a chain take the form with run-time flags.
  float fa[2][Q], fb[2][Q];
  if (total > 0) {
#pragma unroll
    for (int st = 0; st < Q; st++) {
      fa[0][st] = As[0][aoff[st] + st * 4 * SLD];
      fb[0][st] = Bs[0][boff[st] + st * 4 * SLD];
    }
  }
  auto tile_body = [&](auto par_c, auto steady_c, bool store_rt, bool fetch_rt) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    const bool store_next = STEADY || store_rt, fetch_next = STEADY || fetch_rt;
    const float *as = As[PAR], *bs = Bs[PAR];
    const float *an = As[PAR ^ 1], *bn = Bs[PAR ^ 1];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = q & 1;
#pragma unroll
      for (int st = 0; st < Q; st++) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c][st], fb[c][st], acc, 0, 0, 0);
        if (q + 1 < 4) {  // fragments of the next quarter
          fa[c ^ 1][st] = as[aoff[st] + ((q + 1) * Q + st) * 4 * SLD];
          fb[c ^ 1][st] = bs[boff[st] + ((q + 1) * Q + st) * 4 * SLD];
        } else if (store_next) {  // first fragments of the next tile (its buffer is complete: the barrier is behind us)
          fa[c ^ 1][st] = an[aoff[st] + st * 4 * SLD];
          fb[c ^ 1][st] = bn[boff[st] + st * 4 * SLD];
        }
        if (q == 2 && store_next) {  // one vector of the next tile into the other LDS buffer
          if (st & 1) small_store_one<B_KCONT>(Bs[PAR ^ 1], tid, st >> 1, rb[PAR ^ 1][st >> 1]);
          else small_store_one<A_KCONT>(As[PAR ^ 1], tid, st >> 1, ra[PAR ^ 1][st >> 1]);
        }
        if (q < 2 && fetch_next) {  // one vector of the tile after that into the register stage freed a tile ago
          const int s16 = q * Q + st;
          if ((s16 & 1) == 0) {
            const int h = s16 >> 2;
            const bool whole = mn_whole && cur.k0 + SK <= K;
            if (s16 & 2) rb[PAR][h] = small_load_one<B_KCONT, FULL>(B, cur.b, n0, cur.k0, tid, h, whole);
            else ra[PAR][h] = small_load_one<A_KCONT, FULL>(A, cur.a, m0, cur.k0, tid, h, whole);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (q == 2) {
        // the last quarter's fragments are in registers before the barrier (see gemm_mfma16ws_kernel)
#pragma unroll
        for (int st = 0; st < Q; st++) asm volatile("" : "+v"(fa[c ^ 1][st]), "+v"(fb[c ^ 1][st]));
        __syncthreads();
      }
    }
    if (fetch_next) cur.advance();
  };
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  int it = 0;
  for (; it + 3 < total; it += 2) {  // tiles it and it+1 both store tile +1 and fetch tile +2
    tile_body(c0{}, std::true_type{}, true, true);
    tile_body(c1{}, std::true_type{}, true, true);
  }
  for (; it < total; it += 2) {
    tile_body(c0{}, std::false_type{}, it + 1 < total, cur.more());
    if (it + 1 >= total) break;
    tile_body(c1{}, std::false_type{}, it + 2 < total, cur.more());
  }

  uint32_t mx = 0u;  // max |stored value| (Batch.amax)
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = m0 + wm + 4 * lk + r;
    if ((FULL || (row < M && col < N)) && (bt.nstore == 0 || col < bt.nstore)) {
      float *dst = C + (int64_t)row * ldc + col;
      const float o = CHAIN ? acc[r] : epilogue(*dst, acc[r], alpha, beta);
      *dst = o;
      mx = max_u32(mx, abs_bits(o));
    }
  }
  if (bt.amax != nullptr) publish_amax(bt.amax, mx);
}

// The same problem with the work SPLIT BY WAVE (FPEMU_GEMM_WS, the default): the chain waves own the 16x16 chains and
// issue nothing but their fragment reads and MFMAs; loader waves do the global loads and the transposing LDS stores,
// WSD tiles ahead.  In gemm_mfma16_kernel one wave per SIMD issues all of it in order and a chain step takes ~105
// cycles against the 32 of the bare chain (measured: conv weight update, 25 088 steps in 1.1 ms); the instruction
// streams are now disjoint.  One barrier per tile (after the third quarter: the chain waves then hold their last
// fragments of buffer PAR, the loader waves have finished buffer PAR^1), same fmaf chain, so the same bits.
// Two tile sizes (TS): 32 x 32 (four chain waves + 512 loader threads) and 16 x 16 (ONE chain wave + 256 loader
// threads) for problems with so few chains that 32 x 32 tiles leave most CUs idle: a CU sustains ~35 GB/s of loads
// that miss to HBM (its outstanding-miss budget over ~1 us), which is 0.93 us per 32 KiB tile against the chain's
// 0.43 us -- a quarter of the chains per CU halves the bytes per CU and tile.
#ifndef FPEMU_ABLATE_WS
#define FPEMU_ABLATE_WS 0  // timing experiments only: 1 loaders skip the global loads, 2 skip the LDS stores, 4 chain waves skip the MFMAs
#endif
#ifndef FPEMU_WS_SPREAD
#define FPEMU_WS_SPREAD 1
#endif
#ifndef FPEMU_WS_PRIO
#define FPEMU_WS_PRIO 0
#endif
#ifndef FPEMU_GEMM_WS_DEPTH
#define FPEMU_GEMM_WS_DEPTH 8     // register stages of the loader waves = tiles a load is issued ahead of its LDS store
#endif
constexpr int WSD = FPEMU_GEMM_WS_DEPTH;
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}
#ifdef FPEMU_WS_TRACE  // tuning builds: s_memtime stamps of workgroup 0's first chain wave and first loader wave, per tile
__device__ unsigned long long *g_ws_trace = nullptr;
extern "C" int fpemu_test_ws_trace_(void *ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ws_trace), &ptr, sizeof(ptr)); }
#define WS_STAMP(slot, it) do { if ((threadIdx.x & 255) == 0 && (it) < 256) ws_stamps[(it) * 8 + (slot)] = (unsigned)__builtin_readcyclecounter(); } while (0)
#else
#define WS_STAMP(slot, it) do { } while (0)
#endif

// Geometry of a TS x TS tile: NC chain waves, NL loader threads in groups of 256; a group covers the tile's
// [TS][SK] operand image in NPASS = TS * SK / 1024 passes of 256 vectors, group g taking passes g, g + G, ...
template <int TS> struct WsGeo {
  static_assert(TS == 32 || TS == 16, "tile side");
  static constexpr int NC = (TS / 16) * (TS / 16);
  static constexpr int NL = TS == 32 ? 512 : 256;
  static constexpr int G = NL / 256;
  static constexpr int NPASS = TS * SK / 1024;
  static constexpr int LNV = NPASS / G;       // vectors per operand, tile and loader thread (2 for both sizes)
  // 16 x 16: eight waves are launched and waves 4, 6 and 7 leave at once, so that (waves going to the SIMDs in turn)
  // the chain wave has SIMD 0 to itself and the loaders -- waves 1, 2, 3 and 5 -- sit on SIMDs 1, 2, 3 and 1: a loader
  // wave on the chain's SIMD takes issue slots between its dependent MFMAs.
  static constexpr bool SPREAD = TS == 16 && FPEMU_WS_SPREAD;
  static constexpr int NTHREADS = SPREAD ? 512 : NC * 64 + NL;
  static constexpr int WORDS = TS == 32 ? SK * 48 : SK * 16;  // one operand image
};
// thread t (0..255) of pass `pass` -> (row r, first k) of its vector: four consecutive k (KCONT) or four rows
template <bool KCONT, int TS> __device__ __forceinline__ void ws_coords(int pass, int t, int &r, int &k) {
  if constexpr (TS == 32) {
    small_coords<KCONT>(pass, t, r, k);
  } else if constexpr (KCONT) {  // 8 threads per row and 32 k; 16 rows x 64 k per pass
    r = (t >> 3) & 15;
    k = pass * 64 + (t >> 7) * 32 + (t & 7) * 4;
  } else {                        // 4 threads per k; 64 k per pass
    r = (t & 3) * 4;
    k = pass * 64 + (t >> 2);
  }
}
// LDS word of element (k, col).  TS = 32: see SLD above.  TS = 16: rows of 16 words, row k at k ^ ((k >> 2) & 1), column
// XORed with 4 * ((k >> 3) & 3): a fragment read (lanes li < 16, lk < 4 of step s: k = 4s + lk) has bank bit 4 =
// lk ^ s (mod 2) and the 16 columns permuted, a transposing store (4 rows x 8 k-groups per half-wave) gets bank bit 4
// from the k-group's low bit and bits 2-3 from its high bits: 32 distinct banks either way.
template <int TS> __device__ __forceinline__ int ws_word(int k, int col) {
  if constexpr (TS == 32) return k * SLD + (col ^ small_swz(k));
  else return ((k ^ ((k >> 2) & 1)) << 4) + (col ^ (((k >> 3) & 3) << 2));
}
template <bool KCONT, int TS>
__device__ __forceinline__ void ws_store_one(float *T, int t, int pass, const f32x4 v) {
  int r, k;
  ws_coords<KCONT, TS>(pass, t, r, k);
  if constexpr (KCONT) {
#pragma unroll
    for (int c = 0; c < 4; c++) T[ws_word<TS>(k + c, r)] = v[c];
  } else {
    *reinterpret_cast<f32x4 *>(&T[ws_word<TS>(k, r)]) = v;  // r and the XOR are multiples of four: one aligned vector
  }
}
template <bool KCONT, int TS>
__device__ __forceinline__ f32x4 ws_load_checked(const Operand &X, const float *base, int r0, int k0, int t, int pass) {
  int r, k;
  ws_coords<KCONT, TS>(pass, t, r, k);
  r += r0;
  k += k0;
  const float *src = KCONT ? base + (int64_t)r * X.ld + k : base + (int64_t)k * X.ld + r;
  f32x4 v = {X.pad, X.pad, X.pad, X.pad};
  const bool in_outer = KCONT ? (r < X.R) : (k < X.K);
  const int inner = KCONT ? k : r, inner_end = KCONT ? X.K : X.R;
  if (in_outer) {
    if (inner + 3 < inner_end) {
      v = *reinterpret_cast<const f32x4u *>(src);
    } else {
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (inner + c < inner_end) v[c] = src[c];
    }
  }
  return v;
}

// MODE 2: whole tiles of aligned operands (unchecked loads); 1: aligned operands, ragged edges (masked vector loads);
// 0: anything (bounds-checked scalar loads).
// WU: B is the implicit gathered operand of a stride-1 weight update (Batch.wu_*): the same chain over the same products
// as with the gathered matrix, read from the padded planes with 4-byte-aligned 16-byte loads (OW % 4 == 0: a vector of
// four pixels never leaves its image row).
template <bool A_KCONT, bool B_KCONT, bool CHAIN, int MODE, int TS, bool WU = false>
__global__ __launch_bounds__(WsGeo<TS>::NTHREADS) void gemm_mfma16ws_kernel(Operand A, Operand B, float *C, int64_t ldc,
                                                                            float alpha, float beta, int tiles_m,
                                                                            int tiles_n, Batch bt) {
  using Geo = WsGeo<TS>;
  constexpr int LNV = Geo::LNV, NCT = Geo::NC * 64;
  A.p += (int64_t)blockIdx.y * bt.sa;
  B.p += (int64_t)blockIdx.y * bt.sb;
  C += (int64_t)blockIdx.y * bt.sc;
  __shared__ __attribute__((aligned(16))) float As[2][Geo::WORDS];
  __shared__ __attribute__((aligned(16))) float Bs[2][Geo::WORDS];
#ifdef FPEMU_WS_TRACE
  __shared__ unsigned ws_stamps[256 * 8];  // stamps go to LDS (cheap, asynchronous) and are dumped once at the end
#endif
  int tm, tn;
  if (WU && bt.wu_cb % TS == 0) {
    // Weight update read from the planes: the R*S taps of one group of TS channels are the SAME planes, shifted.  An
    // XCD's run of tiles walks the taps fastest, then the tile rows, then the channel groups, so its L2 holds a few
    // groups of planes and a few row blocks of gout instead of every XCD fetching all of both (C64 56 x 56 at batch 32:
    // 401 MiB of L2 fills from the fabric for 49 MiB of operands in grid order).
    const int taps = bt.wu_r * bt.wu_s, tpg = bt.wu_cb / TS;       // tiles per (channel block, tap)
    const int wg = xcd_run_index(tiles_m * tiles_n);
    const int tap = wg % taps, rest = wg / taps;
    tm = rest % tiles_m;
    const int pg = rest / tiles_m;                                  // (channel block, group of TS channels)
    tn = ((pg / tpg) * taps + tap) * tpg + pg % tpg;
  } else {
    tile_of_block(tiles_m, tiles_n, tm, tn);
  }
  const int m0 = tm * TS, n0 = tn * TS;
  const int K = A.K, M = A.R, N = B.R;
  const int kt = (K + SK - 1) / SK, total = kt * bt.nseg;
  constexpr int Q = SK / 16;  // MFMA steps per quarter of a tile

  // wave-uniform by construction; readfirstlane makes it a scalar branch, so each barrier below is executed by
  // exactly the waves of its role
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6;
  if (Geo::SPREAD && (wave == 4 || wave >= 6)) return;   // before any barrier: a finished wave is not waited for
  if (wave >= Geo::NC) {
    // ---- loader waves: tile j travels through register stage j % WSD and LDS buffer j % 2 ----
    const int lt = Geo::SPREAD ? (((wave < 4 ? wave - 1 : 3) << 6) | (int)(threadIdx.x & 63)) : (int)threadIdx.x - NCT;
    const int t = lt & 255;   // position within a 256-thread pass of the tile
    const int g = lt >> 8;    // this wave's passes: g, g + G, ... (wave-uniform)
    if (total <= 0) {     // K = 0: nothing to load (the operands may be null); only the chain waves' first barrier
      __syncthreads();
      return;
    }
    FetchCursor cur{A.p, B.p, 0, 0, total, K, bt.ka, bt.kb};
    f32x4 ra[WSD][LNV], rb[WSD][LNV];
    // Aligned operands (MODE 1, 2): this thread's vectors of the tile at (segment 0, k0 = 0) as pointers, set up once;
    // a tile's addresses are pointer + one UNIFORM offset (scalar arithmetic, one 64-bit add per vector) instead of a
    // 64-bit multiply per vector per tile.  Rows past the operand's edge are clamped to its last row / last vector:
    // they only feed outputs that are never stored.  MODE 1 masks k past the segment's end (the one thing that
    // reaches valid outputs): the lane then reads the tile base and the zero is selected when the stage is STORED
    // (selecting at the load would make the wave wait for the data it has just asked for).
    [[maybe_unused]] const float *pa_[LNV], *pb_[LNV];
    [[maybe_unused]] int ka_[LNV], kb_[LNV];
    if constexpr (MODE != 0) {
#pragma unroll
      for (int h = 0; h < LNV; h++) {
        int r, k;
        ws_coords<A_KCONT, TS>(g + h * Geo::G, t, r, k);
        r += m0;
        if (A_KCONT) r = r < M ? r : M - 1;
        else r = r < M ? r : M - 4;
        pa_[h] = A_KCONT ? A.p + (int64_t)r * A.ld + k : A.p + (int64_t)k * A.ld + r;
        ka_[h] = k;
        ws_coords<B_KCONT, TS>(g + h * Geo::G, t, r, k);
        r += n0;
        if (B_KCONT) r = r < N ? r : N - 1;
        else r = r < N ? r : N - 4;
        pb_[h] = B_KCONT ? B.p + (int64_t)r * B.ld + k : B.p + (int64_t)k * B.ld + r;
        kb_[h] = k;
        if constexpr (WU) {   // the row's shifted plane; the pixel's offset is added per tile (wu_src)
          int rr = r;
          const int c2 = rr % bt.wu_cb; rr /= bt.wu_cb;
          const int ki = rr % bt.wu_s; rr /= bt.wu_s;
          const int kj = rr % bt.wu_r, cb = rr / bt.wu_r;
          if (bt.wu_st > 1)
            pb_[h] = B.p + (int64_t)(cb * bt.wu_cb + c2) * bt.wu_plane +
                     (int64_t)((kj % bt.wu_st) * bt.wu_st + ki % bt.wu_st) * bt.wu_ph + (kj / bt.wu_st) * bt.wu_wp + ki / bt.wu_st;
          else
            pb_[h] = B.p + (int64_t)(cb * bt.wu_cb + c2) * bt.wu_plane + kj * bt.wu_wp + ki;
        }
      }
    }
    [[maybe_unused]] auto wu_src = [&](const float *rowp, const float *seg, int p) -> const float * {
      const unsigned up = (unsigned)p;
      return rowp + (seg - B.p) + (up + __umulhi(up, bt.wu_magic) * (unsigned)bt.wu_gap);
    };
    // Fetches are UNCONDITIONAL: past the last tile the cursor stays where it is and the loads re-read that tile.
    // With a conditional fetch the compiler cannot count the loads in flight and waits for ALL of them (vmcnt(0))
    // before every LDS store -- the prefetch distance would be gone.
    [[maybe_unused]] unsigned outside[WSD];  // MODE 1: bit h: A's vector h lies past the segment's end; bit 8 + h: B's
    // MODE 1: the stage's tile reaches past its segment's end (uniform).  Only such a tile -- the last of a segment --
    // takes the masked loads and the selecting stores; both arms of either branch issue the same loads / stores, so the
    // counts of operations in flight survive the join.  (The masks cost ~25 VALU instructions per vector, issued on the
    // SIMD a chain wave shares: measured 21.8 ns per chain step with them in every tile, 15.9 without.)
    [[maybe_unused]] bool ragged[WSD];
    auto fetch = [&](f32x4 (&xa)[LNV], f32x4 (&xb)[LNV], unsigned &out, bool &rag) {
      if constexpr (MODE == 0) {
#pragma unroll
        for (int h = 0; h < LNV; h++) {
          xa[h] = ws_load_checked<A_KCONT, TS>(A, cur.a, m0, cur.k0, t, g + h * Geo::G);
          xb[h] = ws_load_checked<B_KCONT, TS>(B, cur.b, n0, cur.k0, t, g + h * Geo::G);
        }
      } else {
        const int64_t offa = (cur.a - A.p) + (A_KCONT ? (int64_t)cur.k0 : (int64_t)cur.k0 * A.ld);  // uniform
        const int64_t offb = (cur.b - B.p) + (B_KCONT ? (int64_t)cur.k0 : (int64_t)cur.k0 * B.ld);
#pragma unroll
        for (int h = 0; h < LNV; h++) {
          if (MODE == 2 || cur.k0 + SK <= K) {
            xa[h] = *reinterpret_cast<const f32x4 *>(pa_[h] + offa);
            if constexpr (WU) xb[h] = *reinterpret_cast<const f32x4u *>(wu_src(pb_[h], cur.b, cur.k0 + kb_[h]));
            else xb[h] = *reinterpret_cast<const f32x4 *>(pb_[h] + offb);
          } else {
            const bool ina = ka_[h] + cur.k0 < K, inb = kb_[h] + cur.k0 < K;
            xa[h] = *reinterpret_cast<const f32x4 *>(ina ? pa_[h] + offa : cur.a);
            if constexpr (WU) xb[h] = *reinterpret_cast<const f32x4u *>(wu_src(pb_[h], cur.b, inb ? cur.k0 + kb_[h] : 0));
            else xb[h] = *reinterpret_cast<const f32x4 *>(inb ? pb_[h] + offb : cur.b);
            if (h == 0) out = 0u;
            out |= (ina ? 0u : 1u << h) | (inb ? 0u : 256u << h);
          }
        }
      }
      if constexpr (MODE == 1) rag = !(cur.k0 + SK <= K);
      cur.advance();
    };
    auto store = [&](int buf, const f32x4 (&xa)[LNV], const f32x4 (&xb)[LNV], unsigned out, bool rag) {
      const f32x4 za = {A.pad, A.pad, A.pad, A.pad}, zb = {B.pad, B.pad, B.pad, B.pad};
      if (MODE == 1 && rag) {
#pragma unroll
        for (int h = 0; h < LNV; h++) {
          ws_store_one<A_KCONT, TS>(As[buf], t, g + h * Geo::G, ((out >> h) & 1u) ? za : xa[h]);
          ws_store_one<B_KCONT, TS>(Bs[buf], t, g + h * Geo::G, ((out >> (8 + h)) & 1u) ? zb : xb[h]);
        }
      } else {
#pragma unroll
        for (int h = 0; h < LNV; h++) {
          ws_store_one<A_KCONT, TS>(As[buf], t, g + h * Geo::G, xa[h]);
          ws_store_one<B_KCONT, TS>(Bs[buf], t, g + h * Geo::G, xb[h]);
        }
      }
    };
    static_for<WSD>([&](auto s_c) {  // tiles 0 .. WSD-1
      fetch(ra[decltype(s_c)::value], rb[decltype(s_c)::value], outside[decltype(s_c)::value], ragged[decltype(s_c)::value]);
    });
    store(0, ra[0], rb[0], outside[0], ragged[0]);
    fetch(ra[0], rb[0], outside[0], ragged[0]);  // tile WSD
    __syncthreads();      // (the chain waves' first barrier)
    // during the multiplication of tile `it`: tile it+1 into LDS buffer (it+1) % 2, tile it+1+WSD into the stage it frees
    // STEADY: every tile of the trip is followed by another one, so the store and the fetch are unconditional.  That is
    // what keeps the prefetch distance: with the `it + 1 < total` test around them in the steady state the compiler's
    // count of loads in flight joins a path that skipped its fetches at every step, and it waited vmcnt(3) / (1) / (0)
    // before each store -- for the loads issued one tile ago, whatever WSD (seen in the ISA; the loaders then paced the
    // chain at one memory round trip per tile).
    auto step = [&](auto s_c, auto steady_c, int it) {
      constexpr int S1 = (decltype(s_c)::value + 1) % WSD;
      constexpr bool STEADY = decltype(steady_c)::value;
      WS_STAMP(4, it);
      if (STEADY || it + 1 < total) {
#if !(FPEMU_ABLATE_WS & 2)
        store((it + 1) & 1, ra[S1], rb[S1], outside[S1], ragged[S1]);
#endif
        WS_STAMP(5, it);
#if !(FPEMU_ABLATE_WS & 1)
        fetch(ra[S1], rb[S1], outside[S1], ragged[S1]);
#endif
      }
      WS_STAMP(6, it);
      __syncthreads();
      WS_STAMP(7, it);
    };
    int it = 0;
    for (; it + WSD < total; it += WSD)   // tiles it .. it + WSD - 1, each with a successor
      static_for<WSD>([&](auto s_c) { step(s_c, std::true_type{}, it + decltype(s_c)::value); });
    {  // the last 1 .. WSD tiles
      bool done = false;  // uniform
      static_for<WSD>([&](auto s_c) {
        if (done) return;
        step(s_c, std::false_type{}, it + decltype(s_c)::value);
        done = it + decltype(s_c)::value + 1 >= total;
      });
    }
    return;
  }

  // ---- chain waves ----
#if FPEMU_WS_PRIO
  __builtin_amdgcn_s_setprio(3);  // ahead of the loader wave that shares the SIMD
#endif
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = TS == 32 ? (w >> 1) * 16 : 0, wn = TS == 32 ? (w & 1) * 16 : 0;
  const int li = lane & 15, lk = lane >> 4;
  const int col = n0 + wn + li;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if constexpr (CHAIN) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = m0 + wm + 4 * lk + r;
      if (row < M && col < N) acc[r] = chain_start(bt, C, (int64_t)row * ldc + col, row);
    }
  }
  __syncthreads();  // tile 0 is in LDS buffer 0
  static_assert(Q == 8, "the swizzles have a period of one quarter (8 steps)");
  // fragment of step s = 8q + st: element (4s + lk, w + li); its word is that of step st plus s - st whole steps
  constexpr int STEP = TS == 32 ? 4 * SLD : 4 * 16;
  int aoff[Q], boff[Q];
#pragma unroll
  for (int st = 0; st < Q; st++) {
    aoff[st] = ws_word<TS>(4 * st + lk, wm + li) - st * STEP;
    boff[st] = ws_word<TS>(4 * st + lk, wn + li) - st * STEP;
  }
  float fa[2][Q], fb[2][Q];
  if (total > 0) {
#pragma unroll
    for (int st = 0; st < Q; st++) {
      fa[0][st] = As[0][aoff[st] + st * STEP];
      fb[0][st] = Bs[0][boff[st] + st * STEP];
    }
  }
  // One tile from LDS buffer PAR.  PAR is a compile-time constant (the tile loop below runs two tiles per trip): every
  // fragment address is then one of the 2 * Q per-lane words plus an immediate, and the stream holds no VALU
  // instruction at all.  With the buffer picked at run time the compiler rebuilt 32 addresses per tile with v_or_b32 --
  // into registers the MFMA issued just before was still reading (measured: 52 cycles per dependent step with idle
  // loaders against 35 for the same stream without those).
  auto tile = [&](auto par_c, [[maybe_unused]] int it) {
    constexpr int PAR = decltype(par_c)::value;
    const float *as = As[PAR], *bs = Bs[PAR];
    const float *an = As[PAR ^ 1], *bn = Bs[PAR ^ 1];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = q & 1;
#pragma unroll
      for (int st = 0; st < Q; st++) {
#if !(FPEMU_ABLATE_WS & 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c][st], fb[c][st], acc, 0, 0, 0);
#endif
        if (q + 1 < 4) {  // fragments of the next quarter
          fa[c ^ 1][st] = as[aoff[st] + ((q + 1) * Q + st) * STEP];
          fb[c ^ 1][st] = bs[boff[st] + ((q + 1) * Q + st) * STEP];
        } else {  // first fragments of the next tile (its buffer is complete: the barrier is behind us).  Unconditional:
                  // after the last tile they are stale words nobody uses, and a branch here costs the compiler its
                  // count of LDS reads in flight (lgkmcnt(0) at the top of every tile)
          fa[c ^ 1][st] = an[aoff[st] + st * STEP];
          fb[c ^ 1][st] = bn[boff[st] + st * STEP];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (q == 0) WS_STAMP(0, it);
      if (q == 2) {
        // The loader waves overwrite this buffer right after the barrier, so the fragments of the last quarter must
        // BE in registers before it.  Without the pin the compiler sinks their ds_reads (and the quarter's first MFMA)
        // below the barrier towards their uses -- seen in the ISA, and as a rare wrong 16 x 16 tile on the GPU.
#pragma unroll
        for (int st = 0; st < Q; st++) asm volatile("" : "+v"(fa[c ^ 1][st]), "+v"(fb[c ^ 1][st]));
        WS_STAMP(1, it);
        __syncthreads();
        WS_STAMP(2, it);
      }
    }
  };
  int it = 0;
  for (; it + 2 <= total; it += 2) {
    tile(std::integral_constant<int, 0>{}, it);
    tile(std::integral_constant<int, 1>{}, it + 1);
  }
  if (it < total) tile(std::integral_constant<int, 0>{}, it);
  uint32_t mx = 0u;  // max |stored value| (Batch.amax)
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = m0 + wm + 4 * lk + r;
    if ((MODE == 2 || (row < M && col < N)) && (bt.nstore == 0 || col < bt.nstore)) {
      float *dst = C + (int64_t)row * ldc + col;
      const float o = CHAIN ? acc[r] : epilogue(*dst, acc[r], alpha, beta);
      *dst = o;
      mx = max_u32(mx, abs_bits(o));
    }
  }
  if (bt.amax != nullptr) publish_amax(bt.amax, mx);
#ifdef FPEMU_WS_TRACE
  if (g_ws_trace && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {  // loaders have long finished
    for (int i = threadIdx.x; i < 256 * 8; i += 64) g_ws_trace[i] = ws_stamps[i];
  }
#endif
}

#ifndef FPEMU_GEMM_WS
#define FPEMU_GEMM_WS 1
#endif
#ifndef FPEMU_GEMM_SMALL_MAX
#define FPEMU_GEMM_SMALL_MAX 320    // up to this many 32 x 32 tiles (per launch): the small-tile chain kernels
#endif
#ifndef FPEMU_GEMM_WS_TINY16
#define FPEMU_GEMM_WS_TINY16 256  // ... and at most one 16 x 16 tile per CU (measured 320x320x4096, 400 tiles: 34 ns a step
                                  // in two rounds of 16 x 16 tiles against 22 as 100 tiles of 32 x 32)
#endif
#ifndef FPEMU_GEMM_WS_TINY
#define FPEMU_GEMM_WS_TINY 128   // fewer 32 x 32 tiles than this (per launch): 16 x 16 tiles
#endif
template <bool CHAIN, int MODE, int TS> mfma_fn_t pick_mfma16ws(bool a_kcont, bool b_kcont) {
  return a_kcont ? (b_kcont ? gemm_mfma16ws_kernel<true, true, CHAIN, MODE, TS> : gemm_mfma16ws_kernel<true, false, CHAIN, MODE, TS>)
                 : (b_kcont ? gemm_mfma16ws_kernel<false, true, CHAIN, MODE, TS> : gemm_mfma16ws_kernel<false, false, CHAIN, MODE, TS>);
}
template <bool CHAIN, int TS> mfma_fn_t pick_mfma16ws_mode(int mode, bool a_kcont, bool b_kcont) {
  return mode == 2 ? pick_mfma16ws<CHAIN, 2, TS>(a_kcont, b_kcont)
       : mode == 1 ? pick_mfma16ws<CHAIN, 1, TS>(a_kcont, b_kcont) : pick_mfma16ws<CHAIN, 0, TS>(a_kcont, b_kcont);
}

template <bool CHAIN, bool FULL> mfma_fn_t pick_mfma16(bool a_kcont, bool b_kcont) {
  return a_kcont ? (b_kcont ? gemm_mfma16_kernel<true, true, CHAIN, FULL> : gemm_mfma16_kernel<true, false, CHAIN, FULL>)
                 : (b_kcont ? gemm_mfma16_kernel<false, true, CHAIN, FULL> : gemm_mfma16_kernel<false, false, CHAIN, FULL>);
}

typedef void (*seqk_fn)(Operand, Operand, float *, int64_t, int, float, float, int, int, Batch);
typedef void (*mfma_fn)(Operand, Operand, float *, int64_t, float, float, int, int, Batch);

bool aligned16(const void *p) { return ((uintptr_t)p % 16u) == 0; }

template <bool CHAIN, bool FULL, int T> seqk_fn pick_seqk(bool a_kcont, bool b_kcont) {
  return a_kcont ? (b_kcont ? gemm_seqk_kernel<true, true, CHAIN, FULL, T> : gemm_seqk_kernel<true, false, CHAIN, FULL, T>)
                 : (b_kcont ? gemm_seqk_kernel<false, true, CHAIN, FULL, T> : gemm_seqk_kernel<false, false, CHAIN, FULL, T>);
}
template <bool CHAIN, bool FULL, int TW> mfma_fn pick_mfma(bool a_kcont, bool b_kcont) {
  return a_kcont ? (b_kcont ? gemm_mfma_kernel<true, true, CHAIN, FULL, TW> : gemm_mfma_kernel<true, false, CHAIN, FULL, TW>)
                 : (b_kcont ? gemm_mfma_kernel<false, true, CHAIN, FULL, TW> : gemm_mfma_kernel<false, false, CHAIN, FULL, TW>);
}
template <int TW> mfma_fn pick_mfma_tw(bool chain, bool full, bool a_kcont, bool b_kcont) {
  return chain ? (full ? pick_mfma<true, true, TW>(a_kcont, b_kcont) : pick_mfma<true, false, TW>(a_kcont, b_kcont))
               : (full ? pick_mfma<false, true, TW>(a_kcont, b_kcont) : pick_mfma<false, false, TW>(a_kcont, b_kcont));
}

int launch_gemm_(const char *who, float *C, const float *A, const float *B, int64_t M, int64_t N, int64_t K,
                 int64_t lda, int64_t ldb, int64_t ldc, float alpha, int a_trans, int b_trans, int variant,
                 bool chain, int64_t batch, Batch bt, void *stream, bool split_ok) {
  char msg[160];
  auto fail = [&](int code, const char *what) {
    snprintf(msg, sizeof(msg), "%s: %s", who, what);
    fpemu_set_error_(msg);
    return code;
  };
  if (M <= 0 || N <= 0 || batch <= 0) return FPEMU_OK;
  if (C == nullptr || (K > 0 && (A == nullptr || B == nullptr))) return fail(FPEMU_EINVAL, "null matrix");
  if (M > 0x7FFFFFFF || N > 0x7FFFFFFF || K > 0x7FFFFFFF || K < 0 || batch > 65535)
    return fail(FPEMU_EINVAL, "dimension out of range");
  if (((uintptr_t)C % 4u) || ((uintptr_t)A % 4u) || ((uintptr_t)B % 4u))
    return fail(FPEMU_EALIGN, "matrix not aligned to 4 bytes");
  // op(A) is M x K: stored [M][K] (k contiguous) or, transposed, [K][M].  op(B) is K x N: stored
  // [K][N] (n contiguous) or, transposed, [N][K] (k contiguous).   (simple_gemm.cpp:52-65)
  Operand a, b;
  a.p = A; a.ld = lda; a.R = (int)M; a.K = (int)K;
  b.p = B; b.ld = ldb; b.R = (int)N; b.K = (int)K;
  const bool a_kcont = !a_trans, b_kcont = b_trans != 0;
  a.pad = -0.0f;
  b.pad = 0.0f;
  // base pointer, leading dimension and batch stride allow 16-byte loads; the extent along the contiguous axis decides
  const bool a_vec_base = aligned16(A) && (lda % 4 == 0) && (bt.sa % 4 == 0);
  const bool b_vec_base = aligned16(B) && (ldb % 4 == 0) && (bt.sb % 4 == 0);
  a.vec = a_vec_base && ((a_kcont ? K : M) % 4 == 0);
  b.vec = b_vec_base && ((b_kcont ? K : N) % 4 == 0);
  const int c_vec = aligned16(C) && (ldc % 4 == 0) && (bt.sc % 4 == 0) && (N % 4 == 0);
  if (bt.nseg < 1) return fail(FPEMU_EINVAL, "k_segments must be >= 1");
  if (bt.nseg > 1 && ((bt.ka % 4) || (bt.kb % 4))) a.vec = b.vec = 0;
  const int tb = fpemu_gemm_tile_edge_(M, N, batch, variant);   // 16 / 32: the small-tile chain kernels
  const bool tiny = tb == 16, small = tb <= 32;
  // Ragged M or N on the large-tile kernels: one ragged edge used to put EVERY tile on the bounds-checked loads
  // (4100 x 4090 x 4096: 0.56 of the MFMA peak against 0.88 whole).  Outputs are independent chains, so the problem is
  // cut along M and N instead: the interior of whole tiles (unchecked 16-byte loads), then the right strip and the
  // bottom strip as problems of their own (their tile size chosen for their own shape).  Same chains, same bits.
  // Plain GEMMs only (the chain / batched / segmented forms come from the convolution, which pads to whole tiles).
  static const bool no_split = getenv("FPEMU_GEMM_NOSPLIT") != nullptr;   // tuning aid (tools/gemm_bench.py)
  if (variant == 1 && !no_split && !chain && batch == 1 && bt.nseg == 1 && !small && split_ok && K > 0 && (M % tb || N % tb) && M >= tb && N >= tb &&
      a_vec_base && b_vec_base && K % (variant == 1 ? MBK : BK) == 0) {
    const int64_t Mi = (M / tb) * tb, Ni = (N / tb) * tb;
    int rc = launch_gemm_(who, C, A, B, Mi, Ni, K, lda, ldb, ldc, alpha, a_trans, b_trans, variant, chain, batch, bt, stream,
                          /*split_ok=*/false);
    if (rc != FPEMU_OK) return rc;
    if (Ni < N) {   // right strip: all M rows, columns Ni..N
      rc = launch_gemm_(who, C + Ni, A, b_kcont ? B + Ni * ldb : B + Ni, M, N - Ni, K, lda, ldb, ldc, alpha, a_trans,
                        b_trans, variant, chain, batch, bt, stream, false);
      if (rc != FPEMU_OK) return rc;
    }
    if (Mi < M) {   // bottom strip: rows Mi..M, columns 0..Ni
      rc = launch_gemm_(who, C + Mi * ldc, a_kcont ? A + Mi * lda : A + Mi, B, M - Mi, Ni, K, lda, ldb, ldc, alpha, a_trans,
                        b_trans, variant, chain, batch, bt, stream, false);
      if (rc != FPEMU_OK) return rc;
    }
    return FPEMU_OK;
  }
  const int tiles_m = (int)((M + tb - 1) / tb), tiles_n = (int)((N + tb - 1) / tb);
  const int64_t nwg = (int64_t)tiles_m * tiles_n;
  if (nwg > 0x7FFFFFFF) return fail(FPEMU_EINVAL, "too many tiles");
  const float beta = 0.0f;  // simple_gemm.cpp:58
  // every tile whole and 16-byte loads everywhere: the kernels drop their bounds tests
  const bool full = a.vec && b.vec && (M % tb == 0) && (N % tb == 0) && (K % (small ? SK : variant == 1 ? MBK : BK) == 0) && K > 0;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)nwg, (unsigned)batch);
  if (variant == 0) {
    seqk_fn fn;
    if (tb == 64)
      fn = chain ? (full ? pick_seqk<true, true, 64>(a_kcont, b_kcont) : pick_seqk<true, false, 64>(a_kcont, b_kcont))
                 : (full ? pick_seqk<false, true, 64>(a_kcont, b_kcont) : pick_seqk<false, false, 64>(a_kcont, b_kcont));
    else
      fn = chain ? (full ? pick_seqk<true, true, 128>(a_kcont, b_kcont) : pick_seqk<true, false, 128>(a_kcont, b_kcont))
                 : (full ? pick_seqk<false, true, 128>(a_kcont, b_kcont) : pick_seqk<false, false, 128>(a_kcont, b_kcont));
    hipLaunchKernelGGL(fn, grid, dim3(NT), 0, s, a, b, C, ldc, c_vec, alpha, beta, tiles_m, tiles_n, bt);
  } else if (small) {
    if (FPEMU_GEMM_WS) {
      const int mode = full ? 2 : (a.vec && b.vec) ? 1 : 0;
      if (tiny) {
        mfma_fn fn = chain ? pick_mfma16ws_mode<true, 16>(mode, a_kcont, b_kcont) : pick_mfma16ws_mode<false, 16>(mode, a_kcont, b_kcont);
        hipLaunchKernelGGL(fn, grid, dim3(WsGeo<16>::NTHREADS), 0, s, a, b, C, ldc, alpha, beta, tiles_m, tiles_n, bt);
      } else {
        mfma_fn fn = chain ? pick_mfma16ws_mode<true, 32>(mode, a_kcont, b_kcont) : pick_mfma16ws_mode<false, 32>(mode, a_kcont, b_kcont);
        hipLaunchKernelGGL(fn, grid, dim3(WsGeo<32>::NTHREADS), 0, s, a, b, C, ldc, alpha, beta, tiles_m, tiles_n, bt);
      }
    } else {
      mfma_fn fn = chain ? (full ? pick_mfma16<true, true>(a_kcont, b_kcont) : pick_mfma16<true, false>(a_kcont, b_kcont))
                         : (full ? pick_mfma16<false, true>(a_kcont, b_kcont) : pick_mfma16<false, false>(a_kcont, b_kcont));
      hipLaunchKernelGGL(fn, grid, dim3(NT), 0, s, a, b, C, ldc, alpha, beta, tiles_m, tiles_n, bt);
    }
  } else if (variant == 1) {
    mfma_fn fn = tb == 64 ? pick_mfma_tw<1>(chain, full, a_kcont, b_kcont) : pick_mfma_tw<2>(chain, full, a_kcont, b_kcont);
    hipLaunchKernelGGL(fn, grid, dim3(NT), 0, s, a, b, C, ldc, alpha, beta, tiles_m, tiles_n, bt);
  } else {
    return fail(FPEMU_EINVAL, "unknown variant");
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FPEMU_OK : fail((int)e, hipGetErrorString(e));
}

int launch_gemm(const char *who, float *C, const float *A, const float *B, int64_t M, int64_t N, int64_t K,
                int64_t lda, int64_t ldb, int64_t ldc, float alpha, int a_trans, int b_trans, int variant,
                bool chain, int64_t batch, Batch bt, void *stream) {
  return launch_gemm_(who, C, A, B, M, N, K, lda, ldb, ldc, alpha, a_trans, b_trans, variant, chain, batch, bt, stream,
                      /*split_ok=*/true);
}

}  // namespace

// The tile edge for an M x N (x batch) problem: 16 / 32 (the small-tile chain kernels, whose k tiles are 128 deep) or
// 64 / 128 (k tiles of 32; VALU kernel 16).  Also called by fpemu_conv.hip, which pads a ragged chain segment to whole
// k tiles only where that puts the GEMM on an unchecked-load kernel.
extern "C" int fpemu_gemm_tile_edge_(int64_t M, int64_t N, int64_t batch, int variant) {
  // Tile edge: 128 unless that leaves the chip mostly idle (fewer tiles than two per CU) -- a chain cannot
  // be split along k, so small M x N problems get 64 x 64 tiles (MFMA kernel only).
  const int64_t t128 = ((M + 127) / 128) * ((N + 127) / 128) * batch;
  const int64_t t64 = ((M + 63) / 64) * ((N + 63) / 64) * batch;
  // fewer 64 x 64 tiles than CUs: the chains are latency-bound -> 32 x 32 tiles of 16x16x4 MFMAs (gemm_mfma16_kernel)
  const int64_t t32 = ((M + 31) / 32) * ((N + 31) / 32) * batch;
  const bool small = variant == 1 && t64 < 256 && t32 <= FPEMU_GEMM_SMALL_MAX;  // up to ~1.25 waves per SIMD (measured: at 576 tiles
                                                               // the 64 x 64 kernel is the faster one again)
  // so few chains that even 32 x 32 tiles leave most CUs idle: 16 x 16 tiles, one chain per workgroup (a CU's load
  // rate, not the chain, bounds a 32 x 32 tile: see gemm_mfma16ws_kernel)
  const int64_t t16 = ((M + 15) / 16) * ((N + 15) / 16) * batch;
  const bool tiny = small && FPEMU_GEMM_WS && t32 < FPEMU_GEMM_WS_TINY && t16 <= FPEMU_GEMM_WS_TINY16;
  // (the VALU kernel: 64 x 64 tiles when 128 x 128 ones would leave CUs without a workgroup)
  return tiny ? 16 : small ? SB : (variant == 1 && (t128 < 512 || M <= 64 || N <= 64)) ? 64
               : (variant != 1 && t128 < FPEMU_SEQK_T64_BELOW) ? 64 : 128;
}


// Internal to the library (fpemu_conv.hip): one implicit-convolution launch over `batch` images.
//   out[b][m][oy][ox] = chain(bias[m]; for (cb, kj, ki, c2): Wm[m][((cb*R + kj)*S + ki)*64 + c2] *
//                                                    xp[b][cb*64 + c2][oy + kj][ox + (reverse ? S-1-ki : ki)])
// xp: zero-padded planes of Hp x Wp floats (plane = Hp * Wp), `cblocks` blocks of 64 channels per image, readable
// for tile + (R-1)*Wp + S floats past the last plane.  M % 64 == 0.  Returns 0 or a hipError_t.
// Tile edge fpemu_gemm_conv_launch_ will pick (callers size the operand's read-ahead slack by it).
extern "C" int fpemu_gemm_conv_tile_(int64_t M, int64_t nvalid, int64_t batch) {
  const int64_t t128 = (M / 128) * ((nvalid + 127) / 128) * batch;
  return (M % 128 == 0 && t128 >= FPEMU_CONV_T128_MIN) ? 128 : 64;
}

extern "C" int fpemu_gemm_conv_launch_(float *out, const float *Wm, const float *xp, const float *bias, int64_t batch,
                                       int64_t M, int64_t cblocks, int64_t R, int64_t S, int64_t plane, int64_t Wp,
                                       int64_t OH, int64_t OW, int reverse, uint32_t *amax, void *stream) {
  const int64_t nvalid = OH * Wp, crs = cblocks * 64 * R * S;
  const int tb = fpemu_gemm_conv_tile_(M, nvalid, batch);
  const int tiles_m = (int)(M / tb), tiles_n = (int)((nvalid + tb - 1) / tb);
  Operand a, b;
  a.p = Wm; a.ld = crs; a.R = (int)M; a.K = 64; a.vec = 1; a.pad = -0.0f;
  b.p = reverse ? xp + (S - 1) : xp; b.ld = plane; b.R = tiles_n * tb; b.K = 64; b.vec = 1; b.pad = 0.0f;
  Batch bt{0, cblocks * 64 * plane, M * OH * OW, 64, 64 * plane, (int)(cblocks * R * S)};
  bt.amax = amax;
  ConvMap cm{bias, (int)S, (int)R, (int)Wp, reverse ? -1 : 1, (int)OW, (int)(OH * OW), (int)nvalid, 1, 0};
  const dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)batch);
  if (tb == 128)
    hipLaunchKernelGGL(gemm_mfma_conv_kernel<2>, grid, dim3(NT), 0, (hipStream_t)stream, a, b, out, tiles_m, tiles_n, bt, cm);
  else
    hipLaunchKernelGGL(gemm_mfma_conv_kernel<1>, grid, dim3(NT), 0, (hipStream_t)stream, a, b, out, tiles_m, tiles_n, bt, cm);
  return (int)hipGetLastError();
}

// Internal to the library (fpemu_conv.hip): the strided forward convolution over de-interleaved phase planes (ConvMap.st).
//   out[b][m][oy][ox] = chain(bias[m]; for (cb, kj, ki, c2): Wm[m][((cb*R + kj)*S + ki)*64 + c2] *
//                             xp[b][cb*64 + c2][(kj % st)*st + ki % st][oy + kj / st][ox + ki / st])
// xp: per channel st*st phase planes of Hh x Wh floats (PH = Hh * Wh), readable for tile + (R/st + 1)*Wh + S floats past the
// last plane.  M % 64 == 0.  The position grid is OH rows of Wh columns, of which the first OW are results.
extern "C" int fpemu_gemm_conv_phase_launch_(float *out, const float *Wm, const float *xp, const float *bias, int64_t batch,
                                             int64_t M, int64_t cblocks, int64_t R, int64_t S, int64_t st, int64_t PH,
                                             int64_t Wh, int64_t OH, int64_t OW, uint32_t *amax, void *stream) {
  const int64_t nvalid = OH * Wh, crs = cblocks * 64 * R * S, chan = st * st * PH;
  const int tb = fpemu_gemm_conv_tile_(M, nvalid, batch);
  const int tiles_m = (int)(M / tb), tiles_n = (int)((nvalid + tb - 1) / tb);
  Operand a, b;
  a.p = Wm; a.ld = crs; a.R = (int)M; a.K = 64; a.vec = 1; a.pad = -0.0f;
  b.p = xp; b.ld = chan; b.R = tiles_n * tb; b.K = 64; b.vec = 1; b.pad = 0.0f;
  Batch bt{0, cblocks * 64 * chan, M * OH * OW, 64, 64 * chan, (int)(cblocks * R * S)};
  bt.amax = amax;
  ConvMap cm{bias, (int)S, (int)R, (int)Wh, 1, (int)OW, (int)(OH * OW), (int)nvalid, (int)st, PH};
  const dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)batch);
  if (tb == 128)
    hipLaunchKernelGGL(gemm_mfma_conv_kernel<2>, grid, dim3(NT), 0, (hipStream_t)stream, a, b, out, tiles_m, tiles_n, bt, cm);
  else
    hipLaunchKernelGGL(gemm_mfma_conv_kernel<1>, grid, dim3(NT), 0, (hipStream_t)stream, a, b, out, tiles_m, tiles_n, bt, cm);
  return (int)hipGetLastError();
}

extern "C" int fpemu_gemm_f32(float *C, const float *A, const float *B, int64_t M, int64_t N, int64_t K,
                              int64_t lda, int64_t ldb, int64_t ldc, float alpha, int a_trans,
                              int b_trans, int variant, void *stream) {
  return launch_gemm("fpemu_gemm_f32", C, A, B, M, N, K, lda, ldb, ldc, alpha, a_trans, b_trans, variant, false,
                     1, Batch{0, 0, 0, 0, 0, 1}, stream);
}

extern "C" int fpemu_gemm_f32_amax(float *C, const float *A, const float *B, int64_t M, int64_t N, int64_t K,
                                   int64_t lda, int64_t ldb, int64_t ldc, float alpha, int a_trans,
                                   int b_trans, int variant, uint32_t *amax_bits, void *stream) {
  if (amax_bits == nullptr || ((uintptr_t)amax_bits % 4u)) {
    fpemu_set_error_("fpemu_gemm_f32_amax: null or unaligned amax word");
    return amax_bits == nullptr ? FPEMU_EINVAL : FPEMU_EALIGN;
  }
  Batch bt{0, 0, 0, 0, 0, 1};
  bt.amax = amax_bits;
  return launch_gemm("fpemu_gemm_f32_amax", C, A, B, M, N, K, lda, ldb, ldc, alpha, a_trans, b_trans, variant, false,
                     1, bt, stream);
}

extern "C" int fpemu_gemm_chain_f32(float *C, const float *A, const float *B, int64_t M, int64_t N, int64_t K,
                                    int64_t lda, int64_t ldb, int64_t ldc, int a_trans, int b_trans,
                                    int variant, int64_t batch, int64_t stride_a, int64_t stride_b,
                                    int64_t stride_c, int64_t k_segments, int64_t kseg_stride_a,
                                    int64_t kseg_stride_b, void *stream) {
  if (k_segments < 1 || k_segments > 0x7FFFFFFF) {
    fpemu_set_error_("fpemu_gemm_chain_f32: k_segments out of range");
    return FPEMU_EINVAL;
  }
  return launch_gemm("fpemu_gemm_chain_f32", C, A, B, M, N, K, lda, ldb, ldc, 1.0f, a_trans, b_trans, variant,
                     true, batch, Batch{stride_a, stride_b, stride_c, kseg_stride_a, kseg_stride_b, (int)k_segments},
                     stream);
}

// fpemu_gemm_chain_f32 with the chains' starting values given instead of read from C (library-internal, used by
// fpemu_conv.hip's gathered forms): start 1 = init[row], 2 = init[row] + 0.0f, 3 = +0 (init unused).  C is only written,
// and only its first nstore columns when nstore != 0 (Batch.nstore).
// Internal to the library (fpemu_conv.hip): `count` (<= 4) chain GEMMs C_g[b] = chain from +0 of A_g (M x K_g, k
// contiguous, shared by the batch) times B_g[b] (K_g x N, ld = N, batch stride K_g * N), C_g[b] at stride_c, in one
// launch.  Returns FPEMU_OK, an error, or -100 when the problems do not qualify (ragged tiles, small-tile shapes, the
// VALU variant): the caller then launches them one by one.
extern "C" int fpemu_gemm_chain_group_launch_(int count, float *const *C, const float *const *A, const float *const *B,
                                              const int64_t *K, int64_t M, int64_t N, int64_t batch, int64_t stride_c,
                                              int variant, void *stream) {
  if (count < 2 || count > kGroupMax || variant != 1 || M <= 0 || N <= 0 || batch <= 0 || batch > 65535) return -100;
  const int tb = fpemu_gemm_tile_edge_(M, N, batch, variant);   // the edge the phases' pitch was padded to
  if (tb < 64 || M % tb || N % tb || N % 4 || stride_c % 4 || M > 0x7FFFFFFF || N > 0x7FFFFFFF) return -100;
  GroupTab gt;
  for (int g = 0; g < kGroupMax; g++) { gt.a[g] = nullptr; gt.b[g] = nullptr; gt.c[g] = nullptr; gt.k[g] = 0; }
  // deepest problem first: its workgroups are the long ones, the short ones fill in behind them
  int order[kGroupMax];
  for (int g = 0; g < count; g++) order[g] = g;
  for (int i = 1; i < count; i++)
    for (int j = i; j > 0 && K[order[j]] > K[order[j - 1]]; j--) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  for (int i = 0; i < count; i++) {
    const int g = order[i];
    if (K[g] <= 0 || K[g] % MBK || K[g] > 0x7FFFFFFF || !aligned16(A[g]) || !aligned16(B[g]) || !aligned16(C[g]) ||
        (K[g] * N) % 4)
      return -100;
    gt.a[i] = A[g]; gt.b[i] = B[g]; gt.c[i] = C[g]; gt.k[i] = (int)K[g];
  }
  const int tiles_m = (int)(M / tb), tiles_n = (int)(N / tb);
  if ((int64_t)tiles_m * tiles_n > 0x7FFFFFFF) return -100;
  Operand a, b;
  a.p = nullptr; a.ld = 0; a.R = (int)M; a.K = 0; a.vec = 1; a.pad = -0.0f;
  b.p = nullptr; b.ld = N; b.R = (int)N; b.K = 0; b.vec = 1; b.pad = 0.0f;
  Batch bt{0, 0, stride_c, 0, 0, 1, 3, nullptr, 0, nullptr};
  const dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)batch, (unsigned)count);
  if (tb == 64)
    hipLaunchKernelGGL(gemm_mfma_group_kernel<1>, grid, dim3(NT), 0, (hipStream_t)stream, a, b, N, tiles_m, tiles_n, bt, gt);
  else
    hipLaunchKernelGGL(gemm_mfma_group_kernel<2>, grid, dim3(NT), 0, (hipStream_t)stream, a, b, N, tiles_m, tiles_n, bt, gt);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fpemu_set_error_(hipGetErrorString(e));
    return (int)e;
  }
  return FPEMU_OK;
}

extern "C" int fpemu_gemm_chain_start_launch_(float *C, const float *A, const float *B, int64_t M, int64_t N, int64_t K,
                                              int64_t lda, int64_t ldb, int64_t ldc, int a_trans, int b_trans,
                                              int variant, int64_t batch, int64_t stride_a, int64_t stride_b,
                                              int64_t stride_c, int64_t k_segments, int64_t kseg_stride_a,
                                              int64_t kseg_stride_b, int start, const float *init, int64_t nstore,
                                              uint32_t *amax, void *stream) {
  if (k_segments < 1 || k_segments > 0x7FFFFFFF || start < 0 || start > 3 || ((start == 1 || start == 2) && !init) ||
      nstore < 0 || nstore > N || (nstore != 0 && start == 0)) {
    fpemu_set_error_("fpemu_gemm_chain_start_launch_: bad k_segments / start");
    return FPEMU_EINVAL;
  }
  return launch_gemm("fpemu_gemm_chain_f32", C, A, B, M, N, K, lda, ldb, ldc, 1.0f, a_trans, b_trans, variant, true, batch,
                     Batch{stride_a, stride_b, stride_c, kseg_stride_a, kseg_stride_b, (int)k_segments, start, init, (int)nstore,
                           amax},
                     stream);
}

// Internal to the library (fpemu_conv.hip): the weight update of a stride-1 convolution without its gathered operand.
//   Gm[k][r] = chain (from +0 when start == 3, from Gm's values when start == 0) over images n, pixels p ascending of
//              gout[n][k][p] * xp[n][plane(r)][(p / OW + kj(r)) * Wp + p % OW + ki(r)],   r = (cb, kj, ki, c2), c2 fastest
// -- the products and the order of fpemu_gemm_chain_start_launch_ over cols_fp's matrix (the padded planes hold the zeros
// cols_fp writes at padding taps), read in place by the small-tile chain kernel's loader waves.
// fpemu_gemm_wu_implicit_ok_: whether a problem qualifies (the small-tile kernels, vector loads of gout, OW % 4 == 0).
extern "C" int fpemu_gemm_wu_implicit_ok_(const float *gout, int64_t K, int64_t crs, int64_t P, int64_t OW, int64_t kseg_a,
                                          int variant) {
  if (!FPEMU_GEMM_WS || variant != 1 || K <= 0 || crs <= 0 || P <= 0 || OW <= 0) return 0;
  if (fpemu_gemm_tile_edge_(K, crs, 1, variant) > 32) return 0;
  if (!aligned16(gout) || (P % 4) || (kseg_a % 4) || (OW % 4)) return 0;
  if (P >= ((int64_t)1 << 20) || OW >= 4096 || K > 0x7FFFFFFF || crs > 0x7FFFFFFF) return 0;
  return 1;
}

extern "C" int fpemu_gemm_wu_implicit_launch_(float *Gm, const float *gout, const float *xp, int64_t K, int64_t crs,
                                              int64_t P, int64_t images, int64_t kseg_a, int64_t kseg_b, int start,
                                              int64_t CB, int64_t R, int64_t S, int64_t plane, int64_t Wp, int64_t OW,
                                              int64_t st, int64_t PH, void *stream) {
  // st > 1: `plane` is a channel's stride over its st * st phase planes of PH floats each (pitch Wp), see Batch.wu_st
  if (!fpemu_gemm_wu_implicit_ok_(gout, K, crs, P, OW, kseg_a, 1) || images < 1 || images > 0x7FFFFFFF ||
      (start != 0 && start != 3) || !Gm || !xp || CB < 1 || R < 1 || S < 1 || crs % (CB * R * S) || Wp < OW || st < 1 ||
      st > 4) {
    fpemu_set_error_("fpemu_gemm_wu_implicit_launch_: problem does not qualify");
    return FPEMU_EINVAL;
  }
  const int tb = fpemu_gemm_tile_edge_(K, crs, 1, 1);
  Operand a, b;
  a.p = gout; a.ld = P; a.R = (int)K; a.K = (int)P; a.vec = 1; a.pad = -0.0f;
  b.p = xp; b.ld = 0; b.R = (int)crs; b.K = (int)P; b.vec = 1; b.pad = 0.0f;
  Batch bt{0, 0, 0, kseg_a, kseg_b, (int)images, start, nullptr, 0, nullptr};
  bt.wu_gap = (int)(Wp - OW);
  bt.wu_magic = (uint32_t)((((uint64_t)1) << 32) / (uint64_t)OW + 1u);
  bt.wu_cb = (int)CB; bt.wu_r = (int)R; bt.wu_s = (int)S; bt.wu_wp = (int)Wp;
  bt.wu_plane = plane;
  bt.wu_st = (int)st;
  bt.wu_ph = PH;
  const int tiles_m = (int)((K + tb - 1) / tb), tiles_n = (int)((crs + tb - 1) / tb);
  const bool full = (K % tb == 0) && (crs % tb == 0) && (P % SK == 0);
  const dim3 grid((unsigned)(tiles_m * tiles_n), 1u);
  hipStream_t s = (hipStream_t)stream;
  if (tb == 16) {
    mfma_fn fn = full ? gemm_mfma16ws_kernel<true, true, true, 2, 16, true> : gemm_mfma16ws_kernel<true, true, true, 1, 16, true>;
    hipLaunchKernelGGL(fn, grid, dim3(WsGeo<16>::NTHREADS), 0, s, a, b, Gm, crs, 1.0f, 0.0f, tiles_m, tiles_n, bt);
  } else {
    mfma_fn fn = full ? gemm_mfma16ws_kernel<true, true, true, 2, 32, true> : gemm_mfma16ws_kernel<true, true, true, 1, 32, true>;
    hipLaunchKernelGGL(fn, grid, dim3(WsGeo<32>::NTHREADS), 0, s, a, b, Gm, crs, 1.0f, 0.0f, tiles_m, tiles_n, bt);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fpemu_set_error_(hipGetErrorString(e));
    return (int)e;
  }
  return FPEMU_OK;
}
