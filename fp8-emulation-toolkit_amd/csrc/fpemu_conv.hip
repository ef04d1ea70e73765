// fpemu_conv.hip -- the cmodel "simple" direct convolution (forward, backward-data, weight-update) for
// MI355X (gfx950).
//
// Replaces the reference's CPU cmodel convolution
//   simple_conv2d_impl_{fp,bp,wu}   mpemu/cmodel/simple/simple_conv2d_impl.cpp:677-977
//   gemm_kernel_conv_{fp,bp,wu}     mpemu/cmodel/simple/simple_conv2d_impl.cpp:372-675
//   MMEngine_(strideB_)avx2_ps      mpemu/cmodel/simple/simple_mm_engine.cpp:13-44
// whose numerical contract is the accumulation chain of each result element (serial execution order
// of the reference; its OpenMP loops race on shared results, so only that order is defined):
//   fp: out[n][k][oj][oi] = bias[k]; for cb, kj, ki, c2 : += w[k][cb*CB+c2][kj][ki] * x_padded
//   bp: gin[n][c][y][x]   = +0;      for ob, oj asc (kj = y+pad-oj*s), ki asc (oi = (x+pad-ki)/s), o2 :
//                                         += w[ob*KB+o2][c][kj][ki] * gout[n][ob*KB+o2][oj][oi]
//   wu: gw[k][c][kj][ki]  = +0;      for n, oj, oi : += gout[n][k][oj][oi] * x_padded[n][c][oj*s+kj][oi*s+ki]
// (CB / KB = 64-channel blocks, the whole dimension when it is < 64).  Each is a k-ordered chain, so
// each is ONE chain-mode GEMM (fpemu_gemm_chain_f32: accumulators start at C, fmaf chain on the fp32
// MFMA or separate roundings on the VALU, C = acc) over operands laid out in that order:
//   fp: out[n]  (K x P)      = chain(bias rows;  Wfp (K x CRS)    @ cols[n]  (CRS x P))     batched over n
//   bp: gin[n]  (C x HW)     = chain(0;          Wbp (C x KRS)    @ colsb[n] (KRS x HW))    batched over n
//   wu: Gw      (K x CRS)    = chain(Gw;         gout[n] (K x P)  @ cols[n]^T (P x CRS))    k segments n = 0, 1, ...
// cols / colsb are gathered by the kernels below (zero for padding taps: the reference multiplies its
// zero padding in as well; for bp taps that do not exist for a pixel contribute w * 0, which leaves a
// chain that started at +0 unchanged for every finite w -- with stride > 1 and more than 4 channels bp runs
// by phase of the stride over the existing taps only, see cols_bp_phase).  Everything here is HBM-bound
// gather work; the flops are in fpemu_gemm.hip.
//
// Reference defects NOT reproduced: with C < 64 (or K < 64) and N > 1 the reference's
// packing routines index images with C/64 = 0 and collapse the batch (:154, :242); here every image is
// convolved, which equals the reference run one image at a time.  C or K > 64 and not a multiple of
// 64 makes the reference drop channels / read out of bounds (:143-160, :282-307): rejected here.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/fpemu_hip.h"

extern "C" void fpemu_set_error_(const char *msg);
extern "C" int fpemu_gemm_tile_edge_(int64_t M, int64_t N, int64_t batch, int variant);
extern "C" int fpemu_gemm_chain_group_launch_(int count, float *const *C, const float *const *A, const float *const *B,
                                              const int64_t *K, int64_t M, int64_t N, int64_t batch, int64_t stride_c,
                                              int variant, void *stream);
extern "C" int fpemu_gemm_chain_start_launch_(float *C, const float *A, const float *B, int64_t M, int64_t N, int64_t K,
                                              int64_t lda, int64_t ldb, int64_t ldc, int a_trans, int b_trans,
                                              int variant, int64_t batch, int64_t stride_a, int64_t stride_b,
                                              int64_t stride_c, int64_t k_segments, int64_t kseg_stride_a,
                                              int64_t kseg_stride_b, int start, const float *init, int64_t nstore,
                                              uint32_t *amax, void *stream);

namespace {

constexpr int NT = 256;
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Geo {
  int64_t N, C, H, W, K, R, S, stride, pad, OH, OW, CB, KB;
  int64_t KP;  // forward, gathered form: the chain length C*R*S padded to whole k tiles of the GEMM (-0 weights times
               // +0 rows of cols: products of -0, which change no sum).  Only C < 64 leaves a ragged k tile.
};

__device__ __forceinline__ int64_t gtid() { return (int64_t)blockIdx.x * NT + threadIdx.x; }
__device__ __forceinline__ int64_t gstride() { return (int64_t)gridDim.x * NT; }

// Wfp[k][(cb, kj, ki, c2)] = w[k][cb*CB + c2][kj][ki], rows of KP floats (zeros behind the C*R*S weights)
// (first / step: this thread's share of a launch)
// (the index type I is uint32_t when the tensor has fewer than 2^31 elements: five 32-bit divisions per element instead
// of five 64-bit ones -- C512 3 x 3: 12 -> ~5 us)
template <class I>
__device__ __forceinline__ void pack_w_fp_body_t(const float *__restrict__ w, float *__restrict__ Wm, const Geo &g,
                                                 int64_t first, int64_t step) {
  const I C = (I)g.C, R = (I)g.R, S = (I)g.S, CB = (I)g.CB, KP = (I)g.KP;
  const I crs = C * R * S, total = (I)g.K * KP;
  for (I i = (I)first; i < total; i += (I)step) {
    const I k = i / KP;
    I kk = i - k * KP;
    if (kk >= crs) { Wm[i] = -0.0f; continue; }   // times the zero rows of cols: -0, which leaves any sum as it is
    const I c2 = kk % CB; kk /= CB;
    const I ki = kk % S; kk /= S;
    const I kj = kk % R;
    const I cb = kk / R;
    Wm[i] = w[((k * C + cb * CB + c2) * R + kj) * S + ki];
  }
}
__device__ __forceinline__ void pack_w_fp_body(const float *__restrict__ w, float *__restrict__ Wm, const Geo &g,
                                               int64_t first, int64_t step) {
  if (g.K * g.KP <= 0x7FFFFFFF && g.K * g.C * g.R * g.S <= 0x7FFFFFFF && step <= 0x7FFFFFFF)   // uniform
    pack_w_fp_body_t<uint32_t>(w, Wm, g, first, step);
  else
    pack_w_fp_body_t<int64_t>(w, Wm, g, first, step);
}
__global__ __launch_bounds__(NT) void pack_w_fp(const float *__restrict__ w, float *__restrict__ Wm, Geo g) {
  pack_w_fp_body(w, Wm, g, gtid(), gstride());
}

// gw[k][c][kj][ki] = Gw[k][(cb, kj, ki, c2)]
template <class I>
__device__ __forceinline__ void unpack_gw_t(const float *__restrict__ Gm, float *__restrict__ gw, const Geo &g) {
  const I C = (I)g.C, R = (I)g.R, S = (I)g.S, CB = (I)g.CB;
  const I crs = C * R * S, total = (I)g.K * crs;
  for (I i = (I)gtid(); i < total; i += (I)gstride()) {
    const I k = i / crs;
    I kk = i - k * crs;
    const I c2 = kk % CB; kk /= CB;
    const I ki = kk % S; kk /= S;
    const I kj = kk % R;
    const I cb = kk / R;
    gw[((k * C + cb * CB + c2) * R + kj) * S + ki] = Gm[i];
  }
}
__global__ __launch_bounds__(NT) void unpack_gw(const float *__restrict__ Gm, float *__restrict__ gw, Geo g) {
  if (g.K * g.C * g.R * g.S <= 0x7FFFFFFF) unpack_gw_t<uint32_t>(Gm, gw, g);   // uniform
  else unpack_gw_t<int64_t>(Gm, gw, g);
}

// Wbp[c][(ob, tj, ti, o2)] = w[ob*KB + o2][c][R-1-tj][ti]      (oj ascending = kj descending)
template <class I>
__device__ __forceinline__ void pack_w_bp_body_t(const float *__restrict__ w, float *__restrict__ Wm, const Geo &g,
                                                 int64_t first, int64_t step) {
  const I C = (I)g.C, R = (I)g.R, S = (I)g.S, KB = (I)g.KB;
  const I krs = (I)g.K * R * S, total = C * krs;
  for (I i = (I)first; i < total; i += (I)step) {
    const I c = i / krs;
    I kk = i - c * krs;
    const I o2 = kk % KB; kk /= KB;
    const I ti = kk % S; kk /= S;
    const I tj = kk % R;
    const I ob = kk / R;
    Wm[i] = w[(((ob * KB + o2) * C + c) * R + (R - 1 - tj)) * S + ti];
  }
}
__device__ __forceinline__ void pack_w_bp_body(const float *__restrict__ w, float *__restrict__ Wm, const Geo &g,
                                               int64_t first, int64_t step) {
  if (g.K * g.C * g.R * g.S <= 0x7FFFFFFF && step <= 0x7FFFFFFF) pack_w_bp_body_t<uint32_t>(w, Wm, g, first, step);   // uniform
  else pack_w_bp_body_t<int64_t>(w, Wm, g, first, step);
}
__global__ __launch_bounds__(NT) void pack_w_bp(const float *__restrict__ w, float *__restrict__ Wm, Geo g) {
  pack_w_bp_body(w, Wm, g, gtid(), gstride());
}

// cols[n][(cb, kj, ki, c2)][(oj, oi)] = x[n0 + n][cb*CB + c2][oj*s + kj - pad][oi*s + ki - pad], 0 outside.
// One (image, kk) row of the gathered matrix per blockIdx.y (decoded once, scalar), threads along the
// pixels: coalesced stores, row-local loads, 32-bit index arithmetic per element.
// ST = 1 / 2: the stride, known at compile time -- a vector whose four pixels lie inside one image row takes them from
// 4 / 7 contiguous floats (one / two wide 4-byte-aligned loads instead of four loads with a bounds test each).
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
// RB = rows per block trip: RB consecutive rows are RB channels of one filter tap (the channel index runs fastest), so
// they share every index and bounds computation and their loads are issued together -- a block that gathers one short
// row lives one memory round trip per few KiB (measured on the strided layers: 57 us for 122 MB with RB = 1).  RB = 4
// needs the channel block and the rows per image to be multiples of 4 (checked on the host).
// SHORT (vec4 only): rows of fewer than NT vectors (14 x 14 outputs: 49) get a lane group of 1 << lg lanes and a block
// takes NT >> lg row groups per trip, as cols_bp_phase.
template <bool vec4, int ST, int RB, bool SHORT = false>
__global__ __launch_bounds__(NT) void cols_fp(const float *__restrict__ x, float *__restrict__ cols, Geo g,
                                              int64_t nrows, int rows_per_img, int pitch, int lg) {
  // pitch: floats per gathered row, >= OH*OW (weight update: OH*OW padded to whole k tiles; the tail is +0)
  const int P = (int)(g.OH * g.OW), OW = (int)g.OW, H = (int)g.H, W = (int)g.W, st = (int)g.stride;
  const int crs = (int)(g.C * g.R * g.S);
  const int64_t HW = (int64_t)H * W;
  const int lpr = SHORT ? 1 << lg : NT;                                  // lanes per row group
  const int lane = SHORT ? (int)threadIdx.x & (lpr - 1) : (int)threadIdx.x;
  const int64_t rl = SHORT ? (int64_t)(threadIdx.x >> lg) : 0;
  const int64_t gpb = SHORT ? NT >> lg : 1;                              // row groups per block trip
  const int t0 = (int)blockIdx.x * lpr + lane, tstep = (int)gridDim.x * lpr;
  for (int64_t row = ((int64_t)blockIdx.y * gpb + rl) * RB; row < nrows; row += (int64_t)gridDim.y * gpb * RB) {
    int kk = (int)(row % rows_per_img);
    const int64_t n = row / rows_per_img;
    float *dst = cols + row * pitch;
    if (kk >= crs) {  // the zero rows that pad the chain to whole k tiles (forward: rows_per_img = KP)
      for (int p = t0; p < pitch; p += tstep)
#pragma unroll
        for (int u = 0; u < RB; u++) dst[(int64_t)u * pitch + p] = 0.f;
      continue;
    }
    const int c2 = kk % (int)g.CB; kk /= (int)g.CB;
    const int ki = kk % (int)g.S; kk /= (int)g.S;
    const int kj = kk % (int)g.R;
    const int cb = kk / (int)g.R;
    const float *src = x + (n * g.C + cb * g.CB + c2) * HW;   // rows u = 0 .. RB-1: planes c2 + u
    for (int p = P + t0; p < pitch; p += tstep)
#pragma unroll
      for (int u = 0; u < RB; u++) dst[(int64_t)u * pitch + p] = 0.f;
    const int y0 = kj - (int)g.pad, x0 = ki - (int)g.pad;
    if constexpr (vec4) {  // four pixels per lane, one 16-byte store (rows are whole vectors and 16-byte aligned)
      for (int p = t0 * 4; p < P; p += tstep * 4) {
        f32x4 o[RB];
        int oj = p / OW, oi = p - oj * OW;
        bool done = false;
        if constexpr (ST == 1 || ST == 2) {
          const int y = oj * ST + y0, xf = oi * ST + x0;
          if (oi + 3 < OW && (unsigned)y < (unsigned)H && xf >= 0 && xf + 3 * ST < W) {
            const float *q = src + y * W + xf;
            if constexpr (ST == 1) {
              f32x4u a[RB];
#pragma unroll
              for (int u = 0; u < RB; u++) a[u] = *reinterpret_cast<const f32x4u *>(q + u * HW);
#pragma unroll
              for (int u = 0; u < RB; u++) o[u] = f32x4{a[u][0], a[u][1], a[u][2], a[u][3]};
            } else {
              f32x4u a[RB];
              f32x3u b[RB];
#pragma unroll
              for (int u = 0; u < RB; u++) {
                a[u] = *reinterpret_cast<const f32x4u *>(q + u * HW);
                b[u] = *reinterpret_cast<const f32x3u *>(q + u * HW + 4);
              }
#pragma unroll
              for (int u = 0; u < RB; u++) o[u] = f32x4{a[u][0], a[u][2], b[u][0], b[u][2]};
            }
            done = true;
          }
        }
        if (!done) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int y = oj * st + y0, xx = oi * st + x0;
            const bool in = (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
#pragma unroll
            for (int u = 0; u < RB; u++) o[u][e] = in ? src[u * HW + y * W + xx] : 0.f;
            if (++oi == OW) { oi = 0; oj++; }
          }
        }
#pragma unroll
        for (int u = 0; u < RB; u++) *reinterpret_cast<f32x4 *>(dst + (int64_t)u * pitch + p) = o[u];
      }
      continue;
    }
    for (int p = t0; p < P; p += tstep) {
      const int oj = p / OW, oi = p - oj * OW;
      const int y = oj * st + y0, xx = oi * st + x0;
      const bool in = (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
#pragma unroll
      for (int u = 0; u < RB; u++) dst[(int64_t)u * pitch + p] = in ? src[u * HW + y * W + xx] : 0.f;
    }
  }
}

// Short rows that are not whole vectors (7 x 7 outputs: 49 pixels): cols_fp's scalar path gives a 256-thread block to
// one 196-byte row at a time (C512 7 x 7 at batch 32: 147 456 rows, 92 us for 29 MB -- 40 % of that layer's weight
// update).  Here a row gets 1 << LG lanes (the smallest power of two covering its pitch, up to 128) and a block takes
// NT >> LG consecutive rows per trip; the row's tap is decoded per thread (three 32-bit divisions per row, not per pixel).
template <int LG, int RB>   // RB consecutive rows (channels of one tap, as in cols_fp) per lane group
__global__ __launch_bounds__(NT) void cols_fp_short(const float *__restrict__ x, float *__restrict__ cols, Geo g,
                                                    int64_t nrows, int rows_per_img, int pitch, int) {
  constexpr int LPR = 1 << LG, RPB = (NT >> LG) * RB;
  const int P = (int)(g.OH * g.OW), OW = (int)g.OW, H = (int)g.H, W = (int)g.W, st = (int)g.stride;
  const int crs = (int)(g.C * g.R * g.S);
  const int HW = H * W;
  const int lane = (int)threadIdx.x & (LPR - 1), rl = ((int)threadIdx.x >> LG) * RB;
  for (int64_t row0 = (int64_t)blockIdx.x * RPB; row0 < nrows; row0 += (int64_t)gridDim.x * RPB) {
    int64_t n = row0 / rows_per_img;                   // uniform over the block
    int kk = (int)(row0 - n * rows_per_img) + rl;
    while (kk >= rows_per_img) { kk -= rows_per_img; n++; }
    const int64_t row = row0 + rl;
    if (row >= nrows) continue;                        // (RB > 1: nrows is a multiple of RB)
    float *dst = cols + row * pitch;
    const bool real = kk < crs;                        // else: zero rows that pad the chain to whole k tiles
    int k2 = real ? kk : 0;
    const int c2 = k2 % (int)g.CB; k2 /= (int)g.CB;
    const int ki = k2 % (int)g.S; k2 /= (int)g.S;
    const int kj = k2 % (int)g.R;
    const int cb = k2 / (int)g.R;
    const float *src = x + (n * g.C + cb * g.CB + c2) * (int64_t)HW;
    const int y0 = kj - (int)g.pad, x0 = ki - (int)g.pad;
    for (int p = lane; p < pitch; p += LPR) {
      float v[RB];
#pragma unroll
      for (int u = 0; u < RB; u++) v[u] = 0.f;
      if (real && p < P) {
        const int oj = p / OW, oi = p - oj * OW;
        const int y = oj * st + y0, xx = oi * st + x0;
        if ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W) {
#pragma unroll
          for (int u = 0; u < RB; u++) v[u] = src[u * HW + y * W + xx];
        }
      }
#pragma unroll
      for (int u = 0; u < RB; u++) dst[(int64_t)u * pitch + p] = v[u];
    }
  }
}

typedef void (*cols_fp_fn)(const float *, float *, Geo, int64_t, int, int, int);
cols_fp_fn pick_cols_fp(bool vec4, int64_t stride, bool rb4, bool shortrows) {
  if (!vec4) return cols_fp<false, 0, 1>;
  if (shortrows) {
    if (rb4) return stride == 1 ? cols_fp<true, 1, 4, true> : stride == 2 ? cols_fp<true, 2, 4, true> : cols_fp<true, 0, 4, true>;
    return stride == 1 ? cols_fp<true, 1, 1, true> : stride == 2 ? cols_fp<true, 2, 1, true> : cols_fp<true, 0, 1, true>;
  }
  if (rb4) return stride == 1 ? cols_fp<true, 1, 4> : stride == 2 ? cols_fp<true, 2, 4> : cols_fp<true, 0, 4>;
  return stride == 1 ? cols_fp<true, 1, 1> : stride == 2 ? cols_fp<true, 2, 1> : cols_fp<true, 0, 1>;
}

// colsb[n][(ob, tj, ti, o2)][(ih, iw)] = gout[n][ob*KB + o2][oj][oi] with kj = R-1-tj, ki = ti,
// oj = (ih + pad - kj)/s, oi = (iw + pad - ki)/s when those divide and are in range, else 0.
// Same decomposition: one gathered row per blockIdx.y, threads along the input pixels.
template <bool vec4>
__global__ __launch_bounds__(NT) void cols_bp(const float *__restrict__ go, float *__restrict__ cols, Geo g,
                                              int64_t nrows) {
  const int HW = (int)(g.H * g.W), W = (int)g.W, OH = (int)g.OH, OW = (int)g.OW, st = (int)g.stride;
  const int krs = (int)(g.K * g.R * g.S);
  for (int64_t row = blockIdx.y; row < nrows; row += gridDim.y) {
    int kk = (int)(row % krs);
    const int64_t n = row / krs;
    const int o2 = kk % (int)g.KB; kk /= (int)g.KB;
    const int ti = kk % (int)g.S; kk /= (int)g.S;
    const int tj = kk % (int)g.R;
    const int ob = kk / (int)g.R;
    const float *src = go + (n * g.K + ob * g.KB + o2) * (int64_t)OH * OW;
    float *dst = cols + row * HW;
    const int y0 = (int)g.pad - ((int)g.R - 1 - tj), x0 = (int)g.pad - ti;
    if constexpr (vec4) {  // four input pixels per lane, one 16-byte store
      for (int pix = (blockIdx.x * NT + threadIdx.x) * 4; pix < HW; pix += gridDim.x * NT * 4) {
        f32x4 o;
        int ih = pix / W, iw = pix - ih * W;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int dy = ih + y0, dx = iw + x0;
          float v = 0.f;
          if (dy >= 0 && dx >= 0) {
            const int oj = dy / st, oi = dx / st;
            if (oj * st == dy && oi * st == dx && oj < OH && oi < OW) v = src[oj * OW + oi];
          }
          o[e] = v;
          if (++iw == W) { iw = 0; ih++; }
        }
        *reinterpret_cast<f32x4 *>(dst + pix) = o;
      }
      continue;
    }
    for (int pix = blockIdx.x * NT + threadIdx.x; pix < HW; pix += gridDim.x * NT) {
      const int ih = pix / W, iw = pix - ih * W;
      const int dy = ih + y0, dx = iw + x0;
      float v = 0.f;
      if (dy >= 0 && dx >= 0) {
        const int oj = dy / st, oi = dx / st;
        if (oj * st == dy && oi * st == dx && oj < OH && oi < OW) v = src[oj * OW + oi];
      }
      dst[pix] = v;
    }
  }
}

// ---- Backward-data with stride > 1, by PHASE of the stride -------------------------------------------------------
// Input pixels (y, x) with y = yh * st + py, x = xh * st + px see only the taps kj = (py + pad) mod st, + st, ... and
// ki = (px + pad) mod st, + st, ...; for those oj = yh + (py + pad - kj) / st (exact) -- a stride-1 correlation of
// gout with a sub-filter.  The gathered form above multiplies the other taps in as zeros: st * st times the flops and
// a gathered operand st * st times the size (ResNet-50's 3 x 3 stride-2 layers: 0.48-0.56 ms at batch 32 against
// 0.1 ms for their stride-1 twins).  Here each phase is its own chain GEMM over its own taps in the reference's order
// (ob, oj ascending = kj descending, ki ascending, o2) -- exactly the products the reference forms, it skips taps that
// do not exist (simple_conv2d_impl.cpp:505-560) -- into a phase plane; one pass then interleaves the planes into gin.
struct Phase {
  int py, px;      // residues of y and x
  int kj0, ki0;    // smallest filter row / column of this phase
  int nj, ni;      // filter rows / columns of this phase (0: no tap -> the phase's pixels are +0)
  int Hh, Wh;      // pixels of this phase: y = yh * st + py < H, x = xh * st + px < W
  int pitch;       // floats per row of the phase's gathered matrix and phase plane (Hh * Wh rounded up to whole tiles)
};

// All phases of a call that have taps and pixels, with where their packed weights and gathered matrices live: ONE
// pack launch and ONE gather launch (blockIdx.y / .z = phase), ONE grouped launch for the phases' GEMMs where they share
// their shape (fpemu_gemm_chain_group_launch_), one interleave: thirteen launches per 3 x 3 stride-2 backward-data call
// became four.  (Folding the pack into the gather launch as an extra grid slice was measured: nothing at C128 / C256,
// 0.171 -> 0.184 ms at C512 14 x 14, where the weights are the larger job and the gather's grid is small.)
struct PhaseSet {
  Phase ph[16];
  int64_t woff[16];   // floats from the start of the packed-weight area
  int64_t goff[16];   // floats from the start of the chunk's gathered area
  int count;
};

// Wph[c][(ob, aj, ai, o2)] = w[ob*KB + o2][c][kj][ki], kj = kj0 + (nj - 1 - aj) * st (descending), ki = ki0 + ai * st
template <class I>
__device__ __forceinline__ void pack_w_bp_phase_body_t(const float *__restrict__ w, float *__restrict__ Wm, const Geo &g,
                                                       const Phase &ph, int64_t first, int64_t step) {
  const I C = (I)g.C, R = (I)g.R, S = (I)g.S, KB = (I)g.KB, nj = (I)ph.nj, ni = (I)ph.ni, st = (I)g.stride;
  const I krs = (I)g.K * nj * ni, total = C * krs;
  for (I i = (I)first; i < total; i += (I)step) {
    const I c = i / krs;
    I kk = i - c * krs;
    const I o2 = kk % KB; kk /= KB;
    const I ai = kk % ni; kk /= ni;
    const I aj = kk % nj;
    const I ob = kk / nj;
    const I kj = (I)ph.kj0 + (nj - 1 - aj) * st, ki = (I)ph.ki0 + ai * st;
    Wm[i] = w[(((ob * KB + o2) * C + c) * R + kj) * S + ki];
  }
}
__device__ __forceinline__ void pack_w_bp_phase_body(const float *__restrict__ w, float *__restrict__ Wm, const Geo &g,
                                                     const Phase &ph, int64_t first, int64_t step) {
  if (g.K * g.C * g.R * g.S <= 0x7FFFFFFF && step <= 0x7FFFFFFF) pack_w_bp_phase_body_t<uint32_t>(w, Wm, g, ph, first, step);
  else pack_w_bp_phase_body_t<int64_t>(w, Wm, g, ph, first, step);
}

__global__ __launch_bounds__(NT) void pack_w_bp_phase(const float *__restrict__ w, float *__restrict__ Wm_, Geo g, PhaseSet ps) {
  pack_w_bp_phase_body(w, Wm_ + ps.woff[blockIdx.y], g, ps.ph[blockIdx.y], gtid(), gstride());
}

// colsp[n][(ob, aj, ai, o2)][(yh, xh)] = gout[n][ob*KB + o2][yh + dj][xh + di] (0 outside), rows of ph.pitch floats
// with a +0 tail.  One gathered row per blockIdx.y, threads along the phase's pixels.
// SHORT: rows of fewer than NT vectors (a 14 x 14 / 7 x 7 phase: pitch 256 / 64) get a lane group of 1 << lg lanes and a
// block takes NT >> lg row groups per trip (the row's tap is then decoded per thread) -- one 256-thread block per
// 64-float row left 240 lanes idle: 53 us for the C512 14 x 14 stride-2 layer's gather.
template <int RB, bool SHORT>   // RB consecutive output channels o2 of one tap per lane group (see cols_fp); needs KB % RB == 0
__global__ __launch_bounds__(NT) void cols_bp_phase(const float *__restrict__ go, float *__restrict__ cols_, Geo g, PhaseSet ps,
                                                    int64_t nimg, int lg) {
  const Phase ph = ps.ph[blockIdx.z];   // uniform: scalar loads from the kernel arguments
  float *cols = cols_ + ps.goff[blockIdx.z];
  const int OH = (int)g.OH, OW = (int)g.OW, st = (int)g.stride, Wh = ph.Wh, HWp = ph.Hh * ph.Wh;
  const int krs = (int)g.K * ph.nj * ph.ni;
  const int64_t nrows = nimg * krs;
  const int64_t plane = (int64_t)OH * OW;
  const int lpr = SHORT ? 1 << lg : NT;                                  // lanes per row group
  const int lane = SHORT ? (int)threadIdx.x & (lpr - 1) : (int)threadIdx.x;
  const int64_t rl = SHORT ? (int64_t)(threadIdx.x >> lg) : 0;           // row group of this thread within the block trip
  const int64_t gpb = SHORT ? NT >> lg : 1;                              // row groups per block trip
  for (int64_t row = ((int64_t)blockIdx.y * gpb + rl) * RB; row < nrows; row += (int64_t)gridDim.y * gpb * RB) {
    int kk = (int)(row % krs);
    const int64_t n = row / krs;
    const int o2 = kk % (int)g.KB; kk /= (int)g.KB;
    const int ai = kk % ph.ni; kk /= ph.ni;
    const int aj = kk % ph.nj;
    const int ob = kk / ph.nj;
    const int kj = ph.kj0 + (ph.nj - 1 - aj) * st, ki = ph.ki0 + ai * st;
    const int dj = (ph.py + (int)g.pad - kj) / st, di = (ph.px + (int)g.pad - ki) / st;   // exact divisions
    const float *src = go + (n * g.K + ob * g.KB + o2) * plane;   // rows u = 0 .. RB-1: channels o2 + u
    float *dst = cols + row * ph.pitch;
    // four pixels per lane and one 16-byte store (the pitch is a multiple of 64); four pixels of one plane row that all
    // exist come from one 4-byte-aligned 16-byte load
    for (int pix = ((int)blockIdx.x * lpr + lane) * 4; pix < ph.pitch; pix += (int)gridDim.x * lpr * 4) {
      f32x4 o[RB];
#pragma unroll
      for (int u = 0; u < RB; u++) o[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (pix < HWp) {
        int yh = pix / Wh, xh = pix - yh * Wh;
        const int oj = yh + dj, oi = xh + di;
        if (xh + 3 < Wh && (unsigned)oj < (unsigned)OH && oi >= 0 && oi + 3 < OW) {
          f32x4u a[RB];
#pragma unroll
          for (int u = 0; u < RB; u++) a[u] = *reinterpret_cast<const f32x4u *>(src + u * plane + oj * OW + oi);
#pragma unroll
          for (int u = 0; u < RB; u++) o[u] = f32x4{a[u][0], a[u][1], a[u][2], a[u][3]};
        } else {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int qj = yh + dj, qi = xh + di;
            if (pix + e < HWp && (unsigned)qj < (unsigned)OH && (unsigned)qi < (unsigned)OW) {
#pragma unroll
              for (int u = 0; u < RB; u++) o[u][e] = src[u * plane + qj * OW + qi];
            }
            if (++xh == Wh) { xh = 0; yh++; }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < RB; u++) *reinterpret_cast<f32x4 *>(dst + (int64_t)u * ph.pitch + pix) = o[u];
    }
  }
}

// gin[n][c][y][x] = plane(y mod st, x mod st)[n][c][(y / st) * Wh + x / st]; +0 where the phase has no tap.
struct PhaseTable {
  const float *plane[16];   // by py * st + px (st <= 4); nullptr: no tap
  int pitch[16], Wh[16];
};
// A lane group of 1 << lg threads takes a row of gin (so the row's plane / y split is one 32-bit division per thread,
// not two 64-bit ones per element).  Q = 1: a thread takes the st pixels x = xh * st .. xh * st + st - 1: one coalesced
// read per phase plane, st consecutive floats written (one 8-byte store for stride 2 and an even width).  Q = 2
// (stride 2, W % 4 == 0, 16-byte aligned gin): two pixels of each of the row's two planes as 8-byte loads, one 16-byte
// store.
template <int ST, int Q>
__global__ __launch_bounds__(NT) void interleave_phases(float *__restrict__ gin, PhaseTable tb, int64_t rows, int H, int W,
                                                        int st_rt, int lg) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int st = ST ? ST : st_rt;
  const int64_t row = (int64_t)blockIdx.x * (NT >> lg) + (threadIdx.x >> lg);
  if (row >= rows) return;
  const int t = (int)threadIdx.x & ((1 << lg) - 1);
  int64_t pl;
  int y;
  if (rows <= 0x7FFFFFFF) {  // uniform
    const uint32_t q = (uint32_t)row / (uint32_t)H;
    pl = q;
    y = (int)((uint32_t)row - q * (uint32_t)H);
  } else {
    pl = row / H;
    y = (int)(row - pl * H);
  }
  const int py = y % st, yh = y / st;
  float *dst = gin + row * (int64_t)W;
  if constexpr (Q == 2) {
    const float *p0 = tb.plane[py * 2], *p1 = tb.plane[py * 2 + 1];
    if (p0 != nullptr) p0 += pl * tb.pitch[py * 2] + yh * tb.Wh[py * 2];
    if (p1 != nullptr) p1 += pl * tb.pitch[py * 2 + 1] + yh * tb.Wh[py * 2 + 1];
    for (int u = t; u < W / 4; u += 1 << lg) {
      f32x2 a = {0.f, 0.f}, b = {0.f, 0.f};
      if (p0 != nullptr) a = *reinterpret_cast<const f32x2 *>(p0 + 2 * u);
      if (p1 != nullptr) b = *reinterpret_cast<const f32x2 *>(p1 + 2 * u);
      *reinterpret_cast<f32x4 *>(dst + 4 * u) = f32x4{a[0], b[0], a[1], b[1]};
    }
  } else {
    const int Wq = (W + st - 1) / st;
    const bool pair = ST == 2 && (W & 1) == 0 && ((uintptr_t)gin & 7u) == 0;
    for (int xh = t; xh < Wq; xh += 1 << lg) {
      float v[ST ? ST : 4];
#pragma unroll
      for (int px = 0; px < (ST ? ST : 4); px++) {
        v[px] = 0.f;
        if (px < st && xh * st + px < W) {
          const int p = py * st + px;
          if (tb.plane[p] != nullptr) v[px] = tb.plane[p][pl * tb.pitch[p] + yh * tb.Wh[p] + xh];
        }
      }
      if (pair) {
        *reinterpret_cast<f32x2 *>(dst + xh * 2) = f32x2{v[0], v[1]};
      } else {
#pragma unroll
        for (int px = 0; px < (ST ? ST : 4); px++)
          if (px < st && xh * st + px < W) dst[xh * st + px] = v[px];
      }
    }
  }
}

// Implicit-GEMM path (stride 1, 64-channel blocks, fmaf-chain variant): instead of gathering R*S shifted copies
// of every plane, zero-pad the planes ONCE; the chain GEMM then reads row (c, kj, ki) of its operand as plane c
// shifted by kj*Wp + ki (fpemu_gemm.hip, ConvMap).  xp[n][c][y][x] = x[n][c][y - ph][x - pw], 0 outside.
// Four consecutive elements per thread, one 16-byte store.
// PACK 1 / 2: workgroups past the first pad_blocks pack the weights (pack_w_fp / pack_w_bp) in the same launch -- the
// GEMM that follows needs both, and a launch boundary costs these 100 us convolutions ~5 %.
struct PackJob {
  const float *w;
  float *Wm;
  Geo g;
};
template <int PACK>
__global__ __launch_bounds__(NT) void pad_planes(const float *__restrict__ x, float *__restrict__ xp, int64_t planes,
                                                 int H, int W, int Hp, int Wp, int ph, int pw, unsigned pad_blocks,
                                                 PackJob job) {
  if constexpr (PACK != 0) {
    if (blockIdx.x >= pad_blocks) {  // uniform
      const int64_t first = (int64_t)(blockIdx.x - pad_blocks) * NT + threadIdx.x, step = (int64_t)(gridDim.x - pad_blocks) * NT;
      if constexpr (PACK == 1) pack_w_fp_body(job.w, job.Wm, job.g, first, step);
      else pack_w_bp_body(job.w, job.Wm, job.g, first, step);
      return;
    }
  }
  const int64_t plane = (int64_t)Hp * Wp, total = planes * plane;
  for (int64_t i = ((int64_t)blockIdx.x * NT + threadIdx.x) * 4; i < total; i += (int64_t)pad_blocks * NT * 4) {
    f32x4 o;
    int64_t pl;
    int rem;
    if (total <= 0x7FFFFFFF) {  // uniform: one 32-bit division instead of a 64-bit one per vector
      const uint32_t q = (uint32_t)i / (uint32_t)plane;
      pl = q;
      rem = (int)((uint32_t)i - q * (uint32_t)plane);
    } else {
      pl = i / plane;
      rem = (int)(i - pl * plane);
    }
    int y = rem / Wp, xx = rem - y * Wp;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int sy = y - ph, sx = xx - pw;
      float v = 0.f;
      if (i + e < total && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W)
        v = x[(pl * H + sy) * (int64_t)W + sx];
      o[e] = v;
      if (++xx == Wp) {
        xx = 0;
        if (++y == Hp) { y = 0; pl++; }
      }
    }
    if (i + 3 < total) {
      *reinterpret_cast<f32x4 *>(xp + i) = o;
    } else {
      for (int e = 0; e < 4 && i + e < total; e++) xp[i + e] = o[e];
    }
  }
}

// Strided forward convolution, implicit form: the zero-padded planes de-interleaved by phase of the stride,
//   xp[plane][(py, px)][yh][xh] = x[plane][yh*st + py - ph][xh*st + px - pw], 0 outside   (Hh x Wh per phase)
// so that every filter tap reads one phase plane with a stride-1 shift (fpemu_gemm.hip, ConvMap.st).  Optionally packs the
// weights in the same launch (PACK, as pad_planes).  Four consecutive elements per thread, one 16-byte store.
template <int PACK>
__global__ __launch_bounds__(NT) void phase_planes(const float *__restrict__ x, float *__restrict__ xp, int64_t planes,
                                                   int H, int W, int Hh, int Wh, int st, int ph, int pw,
                                                   unsigned pad_blocks, PackJob job) {
  if constexpr (PACK != 0) {
    if (blockIdx.x >= pad_blocks) {  // uniform
      const int64_t first = (int64_t)(blockIdx.x - pad_blocks) * NT + threadIdx.x, step = (int64_t)(gridDim.x - pad_blocks) * NT;
      pack_w_fp_body(job.w, job.Wm, job.g, first, step);
      return;
    }
  }
  const int64_t PH = (int64_t)Hh * Wh, chan = PH * st * st, total = planes * chan;
  for (int64_t i = ((int64_t)blockIdx.x * NT + threadIdx.x) * 4; i < total; i += (int64_t)pad_blocks * NT * 4) {
    int64_t pl;
    int rem;
    if (total <= 0x7FFFFFFF) {  // uniform: 32-bit divisions
      const uint32_t q = (uint32_t)i / (uint32_t)chan;
      pl = q;
      rem = (int)((uint32_t)i - q * (uint32_t)chan);
    } else {
      pl = i / chan;
      rem = (int)(i - pl * chan);
    }
    int phase = rem / (int)PH;
    rem -= phase * (int)PH;
    int yh = rem / Wh, xh = rem - yh * Wh;
    int py = phase / st, px = phase - py * st;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int sy = yh * st + py - ph, sx = xh * st + px - pw;
      float v = 0.f;
      if (i + e < total && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W)
        v = x[(pl * H + sy) * (int64_t)W + sx];
      o[e] = v;
      if (++xh == Wh) {
        xh = 0;
        if (++yh == Hh) {
          yh = 0;
          if (++px == st) {
            px = 0;
            if (++py == st) { py = 0; pl++; }
          }
        }
      }
    }
    if (i + 3 < total) {
      *reinterpret_cast<f32x4 *>(xp + i) = o;
    } else {
      for (int e = 0; e < 4 && i + e < total; e++) xp[i + e] = o[e];
    }
  }
}

// phase_planes for stride 2, vectorised: a thread owns four consecutive columns of BOTH column phases of one row of one
// row phase -- eight consecutive input floats (two 4-byte-aligned 16-byte loads), even ones to phase px = 0, odd ones to
// px = 1, two 16-byte stores (4-byte aligned: the planes' pitch Wh is whatever the geometry gives).  The generic kernel
// reads with a stride of two floats and decodes four indices per element: 29 us for the 51 MB of ResNet-50's C128
// 56 x 56 layer at batch 32 against ~16 us here.
typedef float f32x4s __attribute__((ext_vector_type(4), aligned(4)));
template <int PACK>
__global__ __launch_bounds__(NT) void phase_planes_s2(const float *__restrict__ x, float *__restrict__ xp, int64_t planes,
                                                      int H, int W, int Hh, int Wh, int ph, int pw, unsigned pad_blocks,
                                                      PackJob job) {
  if constexpr (PACK != 0) {
    if (blockIdx.x >= pad_blocks) {  // uniform
      const int64_t first = (int64_t)(blockIdx.x - pad_blocks) * NT + threadIdx.x, step = (int64_t)(gridDim.x - pad_blocks) * NT;
      pack_w_fp_body(job.w, job.Wm, job.g, first, step);
      return;
    }
  }
  const int nq = (Wh + 3) / 4;                       // groups of four columns per row
  const int64_t PH = (int64_t)Hh * Wh;
  const int64_t per_plane = (int64_t)2 * Hh * nq, total = planes * per_plane;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)pad_blocks * NT) {
    int64_t pl;
    int rem;
    if (total <= 0x7FFFFFFF) {  // uniform: 32-bit divisions
      const uint32_t q = (uint32_t)i / (uint32_t)per_plane;
      pl = q;
      rem = (int)((uint32_t)i - q * (uint32_t)per_plane);
    } else {
      pl = i / per_plane;
      rem = (int)(i - pl * per_plane);
    }
    const int py = rem / (Hh * nq);
    rem -= py * (Hh * nq);
    const int yh = rem / nq, xq = rem - yh * nq;
    const int sy = yh * 2 + py - ph, sx0 = xq * 8 - pw;
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)sy < (unsigned)H) {
      const float *row = x + (pl * H + sy) * (int64_t)W;
      if (sx0 >= 0 && sx0 + 7 < W) {
        const f32x4u a = *reinterpret_cast<const f32x4u *>(row + sx0), b = *reinterpret_cast<const f32x4u *>(row + sx0 + 4);
        o0 = f32x4{a[0], a[2], b[0], b[2]};
        o1 = f32x4{a[1], a[3], b[1], b[3]};
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int s0 = sx0 + 2 * e, s1 = s0 + 1;
          if ((unsigned)s0 < (unsigned)W) o0[e] = row[s0];
          if ((unsigned)s1 < (unsigned)W) o1[e] = row[s1];
        }
      }
    }
    float *d0 = xp + (pl * 4 + py * 2) * PH + (int64_t)yh * Wh + xq * 4, *d1 = d0 + PH;
    if (xq * 4 + 3 < Wh) {
      *reinterpret_cast<f32x4s *>(d0) = o0;
      *reinterpret_cast<f32x4s *>(d1) = o1;
    } else {
      for (int e = 0; e < 4 && xq * 4 + e < Wh; e++) {
        d0[e] = o0[e];
        d1[e] = o1[e];
      }
    }
  }
}

// Backward-data for very few input channels (the network's first layer: C = 3, 7 x 7, stride 2).  As a GEMM this
// is a 3-row problem over a K*R*S x H*W gathered operand (20 GB for ResNet-50's first layer at batch 32); here a
// lane owns ONE input pixel and its <= 4 channel chains and walks the reference's order (ob, oj ascending,
// ki ascending, o2) directly, taking only the taps that exist (the reference does the same; a missing tap
// contributes nothing).  Lanes of a wave hold pixels x = xi * stride + phase of one row and one phase, so which
// (kj, ki) taps exist is uniform over the wave: the weights are scalar loads and gout is read coalesced.
// Wfew[(ob, kj, ki)][o2][c] = w[ob*KB + o2][c][kj][ki]: the U * C weights of a batch of chain steps are contiguous,
// so they arrive as a few wide scalar loads (in the tensor's own layout every one is a scalar load of its own, from
// a cache line of its own: 37 KB of weights walked by every wave through a 16 KB scalar cache).
__global__ __launch_bounds__(NT) void pack_w_few(const float *__restrict__ w, float *__restrict__ Wm, Geo g) {
  const int64_t total = g.K * g.C * g.R * g.S;
  for (int64_t i = gtid(); i < total; i += gstride()) {
    int64_t kk = i;
    const int64_t c = kk % g.C; kk /= g.C;
    const int64_t o2 = kk % g.KB; kk /= g.KB;
    const int64_t ki = kk % g.S; kk /= g.S;
    const int64_t kj = kk % g.R;
    const int64_t ob = kk / g.R;
    Wm[i] = w[(((ob * g.KB + o2) * g.C + c) * g.R + kj) * g.S + ki];
  }
}

struct FewGeo {  // 32-bit throughout: one image's tensors are far below 2^31 elements
  int st, pad, R, S, OH, OW, K, H, W, KB;
};

#ifndef FPEMU_FEW_ROWS
#define FPEMU_FEW_ROWS 1
#endif
#ifndef FPEMU_FEW_UNROLL
#define FPEMU_FEW_UNROLL 8
#endif
constexpr int kFewUnroll = FPEMU_FEW_UNROLL;  // output channels per batch of gout loads; measured 4 -> 0.402 ms, 8 -> 0.364, 16 -> 0.387
constexpr int kFewRows = FPEMU_FEW_ROWS;  // rows y, y + stride, ... per lane (they share the filter rows); measured on the first ResNet layer (batch 32): 1 -> 0.364 ms, 2 -> 0.46 (too few waves left to hide the load latency)

template <bool FUSED, int CC>
__global__ __launch_bounds__(NT) void bp_few_channels(const float *__restrict__ go, const float *__restrict__ w,
                                                      float *__restrict__ gin, FewGeo g) {
  constexpr int TR = kFewRows;
  const int st = g.st, pad = g.pad, R = g.R, S = g.S, OH = g.OH, OW = g.OW;
  // XCD-aware order.  Workgroups are dispatched round-robin over the 8 XCDs (x fastest, then y, then z), so in grid
  // order every XCD works on every image at once and its 4 MiB L2 thrashes on gout (3.2 MiB per image for the first
  // ResNet layer): measured 5.6 GB of HBM reads for 103 MB of gout.  Give XCD j a contiguous run of the grid instead
  // (whole images, one after the other): bijective for any grid.
  unsigned bx, by_, bz;
  {
    const uint64_t per_img = (uint64_t)gridDim.x * gridDim.y, total = per_img * gridDim.z;
    const uint64_t lin = blockIdx.x + (uint64_t)gridDim.x * (blockIdx.y + (uint64_t)gridDim.y * blockIdx.z);
    const uint64_t xcd = lin % 8u, q = total / 8u, r8 = total % 8u;
    const uint64_t v = (xcd < r8 ? xcd * (q + 1u) : r8 * (q + 1u) + (xcd - r8) * q) + lin / 8u;
    bz = (unsigned)(v / per_img);
    const unsigned in_img = (unsigned)(v - (uint64_t)bz * per_img);
    by_ = in_img / gridDim.x;
    bx = in_img - by_ * gridDim.x;
  }
  int by = (int)by_;
  const int phase = by % st;  by /= st;        // x = xi * st + phase
  const int r = by % st;                       // y = (yb * TR + t) * st + r
  const int yb = by / st;
  const int xi = bx * blockDim.x + threadIdx.x, x = xi * st + phase;   // blockDim.x: 64 / 128 / 256
  const int plane = OH * OW;
  float acc[TR][CC];
#pragma unroll
  for (int t = 0; t < TR; t++)
#pragma unroll
    for (int c = 0; c < CC; c++) acc[t][c] = 0.f;
  // filter rows of these pixel rows: kj = (r + pad) mod st, + st, ... ; output row oj = yb*TR + t + (r + pad - kj) / st.
  // The reference walks oj ascending = kj descending.
  const int res = (r + pad) % st;
  const int kjmax = R - 1 - ((R - 1 - res) % st + st) % st;
  const float *gon = go + (int64_t)bz * g.K * plane;
  for (int ob = 0; ob < g.K; ob += g.KB)
    for (int kj = kjmax; kj >= 0; kj -= st) {
      const int ojb = yb * TR + (r + pad - kj) / st;   // exact: r + pad - kj is a multiple of st (may be negative)
      bool okt[TR], any = false;
#pragma unroll
      for (int t = 0; t < TR; t++) {
        okt[t] = ojb + t >= 0 && ojb + t < OH;   // uniform
        any = any || okt[t];
      }
      if (!any) continue;                          // before the channel loop: a branch inside it serialises the loads
      for (int ki = 0; ki < S; ki++) {
        const int d = phase + pad - ki;
        const int dq = d / st;
        if (dq * st != d) continue;                // uniform: this tap does not exist for this phase
        const int oi = xi + dq;
        if (!(x < g.W && oi >= 0 && oi < OW)) continue;   // per lane, once per tap
        const float *wp = w + (((ob / g.KB) * R + kj) * S + ki) * (g.KB * CC);   // packed: [o2][c] of this tap
        const float *sp = gon + ((ob * OH + ojb) * OW + oi);
        int toff[TR];                              // a missing row reads row ojb's neighbour that exists; its sum is dropped
#pragma unroll
        for (int t = 0; t < TR; t++) toff[t] = (okt[t] ? t : (ojb < 0 ? -ojb : OH - 1 - ojb)) * OW;
        // U output channels at a time: their gout values are loaded one batch ahead of the batch whose U * CC chain
        // steps are running (2 * U loads in flight per lane; the weights are wide scalar loads that hit their cache)
        auto fetch = [&](auto &gv, int o2) {
          constexpr int U = sizeof(gv) / sizeof(gv[0]);
#pragma unroll
          for (int u = 0; u < U; u++)
#pragma unroll
            for (int t = 0; t < TR; t++) gv[u][t] = sp[(o2 + u) * plane + toff[t]];
        };
        auto steps = [&](const auto &gv, int o2) {
          constexpr int U = sizeof(gv) / sizeof(gv[0]);
          float wv[U][CC];
#pragma unroll
          for (int u = 0; u < U; u++)
#pragma unroll
            for (int c = 0; c < CC; c++) wv[u][c] = wp[(o2 + u) * CC + c];
#pragma unroll
          for (int u = 0; u < U; u++)
#pragma unroll
            for (int t = 0; t < TR; t++)
#pragma unroll
              for (int c = 0; c < CC; c++) {
                float a;
                if constexpr (FUSED) a = __builtin_fmaf(wv[u][c], gv[u][t], acc[t][c]);
                else a = acc[t][c] + wv[u][c] * gv[u][t];   // -ffp-contract=off: product and sum rounded separately
                acc[t][c] = okt[t] ? a : acc[t][c];
              }
        };
        constexpr int U = kFewUnroll;
        const int nb = g.KB / U;
        if (nb > 0) {
          float ga[U][TR], gb[U][TR];
          fetch(ga, 0);
          int b = 0;
          for (; b + 2 <= nb; b += 2) {
            fetch(gb, (b + 1) * U);
            steps(ga, b * U);
            fetch(ga, (b + 2 < nb ? b + 2 : nb - 1) * U);   // unconditional (the last one re-reads a batch): a
            steps(gb, (b + 1) * U);                         // load under a branch costs the counted waits
          }
          if (b < nb) steps(ga, b * U);
        }
        for (int o2 = nb * U; o2 < g.KB; o2++) {
          float g1[1][TR];
          fetch(g1, o2);
          steps(g1, o2);
        }
      }
    }
  if (x < g.W) {
#pragma unroll
    for (int t = 0; t < TR; t++) {
      const int y = (yb * TR + t) * st + r;
      if (y >= g.H) continue;
      float *dst = gin + ((int64_t)bz * CC * g.H + y) * g.W + x;
#pragma unroll
      for (int c = 0; c < CC; c++) dst[(int64_t)c * g.H * g.W] = acc[t][c];
    }
  }
}

typedef void (*bp_few_fn)(const float *, const float *, float *, FewGeo);
template <bool FUSED> bp_few_fn pick_bp_few(int C) {
  return C == 1 ? bp_few_channels<FUSED, 1> : C == 2 ? bp_few_channels<FUSED, 2>
       : C == 3 ? bp_few_channels<FUSED, 3> : bp_few_channels<FUSED, 4>;
}

// dst[r][0 .. pitch) = src[r][0 .. P), then -0 (times the +0 tail of the gathered rows: products of -0, which change no sum)
__global__ __launch_bounds__(NT) void pad_rows(const float *__restrict__ src, float *__restrict__ dst, int64_t rows, int P,
                                               int pitch) {
  const int64_t total = rows * pitch;
  for (int64_t i = gtid(); i < total; i += gstride()) {
    const int64_t r = i / pitch;
    const int p = (int)(i - r * pitch);
    dst[i] = p < P ? src[r * P + p] : -0.0f;
  }
}

__global__ __launch_bounds__(NT) void fill_zero(float *__restrict__ out, int64_t total) {
  for (int64_t i = gtid(); i < total; i += gstride()) out[i] = 0.f;
}

// grid for the gather kernels: x over the pixels of a row, y over (image, kk) rows
dim3 gather_grid(int64_t pixels, int64_t nrows) {
  // one block walks a whole row (its decode is amortised over the row) unless rows are very long
  int64_t gx = (pixels + 16 * NT - 1) / (16 * NT);
  if (gx > 64) gx = 64;
  if (gx < 1) gx = 1;
  int64_t gy = nrows < 65535 ? nrows : 65535;
  if (gy < 1) gy = 1;
  return dim3((unsigned)gx, (unsigned)gy);
}

unsigned blocks_for(int64_t total) {
  int64_t b = (total + NT - 1) / NT;
  if (b > 1048576) b = 1048576;
  return b < 1 ? 1u : (unsigned)b;
}

// The forward / weight-update gather of `nrows` rows (rows_per_img per image) of `pitch` floats.
void launch_cols_fp(const float *x, float *cols, const Geo &g, int64_t nrows, int64_t rows_per_img, int64_t pitch,
                    int64_t stride, hipStream_t s) {
  const int64_t P = g.OH * g.OW;
  const bool v4 = P % 4 == 0 && ((uintptr_t)cols % 16u) == 0, rb4 = v4 && g.CB % 4 == 0 && rows_per_img % 4 == 0;
  if (!v4 && pitch <= 128) {
    int lg = 3;
    while ((1 << lg) < pitch) lg++;
    // four channel-consecutive rows per lane group where the channel block and the rows per image allow (a group's
    // rows then share their tap and never straddle an image or the zero rows)
    const bool rb4s = g.CB % 4 == 0 && rows_per_img % 4 == 0 && (g.C * g.R * g.S) % 4 == 0 && nrows % 4 == 0;
    const int64_t rpb = (NT >> lg) * (rb4s ? 4 : 1);
    int64_t blocks = (nrows + rpb - 1) / rpb;
    if (blocks > 1048576) blocks = 1048576;
    if (blocks < 1) blocks = 1;
    cols_fp_fn fn = rb4s ? (lg == 3 ? cols_fp_short<3, 4> : lg == 4 ? cols_fp_short<4, 4> : lg == 5 ? cols_fp_short<5, 4>
                            : lg == 6 ? cols_fp_short<6, 4> : cols_fp_short<7, 4>)
                         : (lg == 3 ? cols_fp_short<3, 1> : lg == 4 ? cols_fp_short<4, 1> : lg == 5 ? cols_fp_short<5, 1>
                            : lg == 6 ? cols_fp_short<6, 1> : cols_fp_short<7, 1>);
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(NT), 0, s, x, cols, g, nrows, (int)rows_per_img, (int)pitch, 0);
    return;
  }
  // whole vectors but fewer than a block's worth per row (the tail beyond P is written by the same lanes)
  int lg = 4;
  const int64_t vecs = (pitch + 3) / 4, groups = nrows / (rb4 ? 4 : 1);
  while ((1 << lg) < vecs && lg < 8) lg++;
  const bool shortrows = v4 && lg < 8;
  const dim3 grid = shortrows ? gather_grid(P, (groups + (NT >> lg) - 1) / (NT >> lg)) : gather_grid(P, groups);
  hipLaunchKernelGGL(pick_cols_fp(v4, stride, rb4, shortrows), grid, dim3(NT), 0, s, x, cols, g, nrows, (int)rows_per_img,
                     (int)pitch, lg);
}

int fail(const char *who, int code, const char *what) {
  char msg[200];
  snprintf(msg, sizeof(msg), "%s: %s", who, what);
  fpemu_set_error_(msg);
  return code;
}

int make_geo(const char *who, Geo &g, int64_t N, int64_t C, int64_t H, int64_t W, int64_t K, int64_t R,
             int64_t S, int64_t stride, int64_t pad) {
  if (N < 0 || C < 1 || H < 1 || W < 1 || K < 1 || R < 1 || S < 1 || stride < 1 || pad < 0)
    return fail(who, FPEMU_EINVAL, "bad dimension");
  if (H + 2 * pad < R || W + 2 * pad < S) return fail(who, FPEMU_EINVAL, "filter larger than the padded input");
  if ((C > 64 && C % 64) || (K > 64 && K % 64))
    return fail(who, FPEMU_EINVAL,
                "C and K must each be <= 64 or a multiple of 64 (the reference's 64-channel packing is undefined otherwise)");
  g.N = N; g.C = C; g.H = H; g.W = W; g.K = K; g.R = R; g.S = S; g.stride = stride; g.pad = pad;
  g.OH = (H + 2 * pad - R) / stride + 1;
  g.OW = (W + 2 * pad - S) / stride + 1;
  g.CB = C < 64 ? C : 64;
  g.KB = K < 64 ? K : 64;
  g.KP = (C * R * S + 31) / 32 * 32;
  if (C * R * S > 0x7FFFFFFF || K * R * S > 0x7FFFFFFF || g.OH * g.OW > 0x7FFFFFFF || H * W > 0x7FFFFFFF)
    return fail(who, FPEMU_EINVAL, "dimension out of range");
  return FPEMU_OK;
}

// floats: packed weights (or the packed gradient) + the gathered operand of `chunk` images
constexpr int64_t kColsCapFloats = (int64_t)1 << 30;  // 4 GiB of gathered operand per chunk at most

// The phases of a strided backward-data convolution (see cols_bp_phase) and what they need per image: a plane and a
// gathered matrix per phase.
constexpr int kMaxPhaseStride = 4;
bool bp_by_phase(const Geo &g) { return g.stride > 1 && g.stride <= kMaxPhaseStride && g.C > 4; }
Phase make_phase(const Geo &g, int py, int px) {
  const int st = (int)g.stride, pad = (int)g.pad, R = (int)g.R, S = (int)g.S, H = (int)g.H, W = (int)g.W;
  Phase ph;
  ph.py = py; ph.px = px;
  ph.kj0 = (py + pad) % st; ph.ki0 = (px + pad) % st;
  ph.nj = ph.kj0 < R ? (R - 1 - ph.kj0) / st + 1 : 0;
  ph.ni = ph.ki0 < S ? (S - 1 - ph.ki0) / st + 1 : 0;
  ph.Hh = py < H ? (H - 1 - py) / st + 1 : 0;
  ph.Wh = px < W ? (W - 1 - px) / st + 1 : 0;
  ph.pitch = (ph.Hh * ph.Wh + 63) / 64 * 64;   // whole 64-column tiles: the GEMM drops its bounds tests
  return ph;
}
void bp_phase_floats(const Geo &g, int64_t &planes, int64_t &gathered) {   // per image
  planes = 0; gathered = 0;
  for (int py = 0; py < (int)g.stride; py++)
    for (int px = 0; px < (int)g.stride; px++) {
      const Phase ph = make_phase(g, py, px);
      if (ph.Hh * ph.Wh == 0 || ph.nj * ph.ni == 0) continue;
      planes += g.C * ph.pitch;
      gathered += g.K * ph.nj * ph.ni * ph.pitch;   // all phases' gathered matrices side by side (one gather launch)
    }
}

// Weight update: the chain of one image is OH*OW products long.  Where the GEMM runs on the large-tile kernels (k tiles
// of 32) and OH*OW is not a multiple of 32 (14 x 14, 7 x 7 planes), both operands are laid out with rows of Pp floats,
// padded with -0 (gout, copied) and +0 (the gathered rows): the extra products are -0 and change no sum, and the GEMM
// runs its unchecked-load kernel (measured, chain GEMM at batch 32: 256 x 2304 over 196 -> 224: 0.216 -> 0.137 ms,
// 512 x 4608 over 49 -> 64: 0.198 -> 0.107 ms).  The small-tile kernels mask a ragged tile cheaply: no padding there.
int64_t wu_pitch(const Geo &g, int variant) {
  const int64_t P = g.OH * g.OW;
  if (P % 32 == 0 || fpemu_gemm_tile_edge_(g.K, g.C * g.R * g.S, 1, variant) < 64) return P;
  return (P + 31) / 32 * 32;
}

// Forward, gathered form: the gathered rows are OH*OW pixels long; where the GEMM runs on the large-tile kernels they
// are padded with zero columns to whole tiles, so that every load is unchecked (the result keeps rows of OH*OW: the
// GEMM computes the extra columns and does not store them, Batch.nstore).
int64_t fp_pitch(const Geo &g, int variant) {
  const int64_t P = g.OH * g.OW;
  const int tile = fpemu_gemm_tile_edge_(g.K, P, g.N > 0 ? g.N : 1, variant);
  if (tile < 64 || P % tile == 0) return P;
  return (P + tile - 1) / tile * tile;
}

void ws_layout(int kind, const Geo &g, int variant, int64_t &w_floats, int64_t &per_img) {
  if (kind == FPEMU_CONV_BP) {
    w_floats = g.C * g.K * g.R * g.S;
    per_img = g.K * g.R * g.S * g.H * g.W;
    if (bp_by_phase(g)) {
      int64_t planes, gathered;
      bp_phase_floats(g, planes, gathered);
      per_img = planes + gathered;
    }
  } else if (kind == FPEMU_CONV_FP) {
    w_floats = g.K * g.KP;
    per_img = g.KP * fp_pitch(g, variant);
  } else {
    const int64_t P = g.OH * g.OW, Pp = wu_pitch(g, variant);
    w_floats = g.K * g.C * g.R * g.S;
    per_img = g.C * g.R * g.S * Pp + (Pp != P ? g.K * Pp : 0);   // gathered rows (+ the padded copy of gout)
  }
  w_floats = (w_floats + 63) & ~(int64_t)63;  // keep the gathered operand 256-byte aligned
}

int64_t chunk_images(const Geo &g, int64_t per_img, int64_t avail_floats) {
  int64_t c = avail_floats / (per_img > 0 ? per_img : 1);
  if (c > g.N) c = g.N;
  if (c > 65535) c = 65535;  // grid.y
  return c;
}

// Implicit path: floats of padded planes for `imgs` images plus the read-ahead slack of the last tile
int64_t implicit_floats(int64_t imgs, int64_t chans, int64_t Hp, int64_t Wp, int64_t R, int64_t S) {
  return imgs * chans * Hp * Wp + 128 + R * Wp + S + 4;
}

#define CHECK_LAUNCH(who)                                               \
  do {                                                                  \
    hipError_t e_ = hipGetLastError();                                  \
    if (e_ != hipSuccess) return fail(who, (int)e_, hipGetErrorString(e_)); \
  } while (0)

}  // namespace

extern "C" int fpemu_gemm_conv_launch_(float *out, const float *Wm, const float *xp, const float *bias, int64_t batch,
                                       int64_t M, int64_t cblocks, int64_t R, int64_t S, int64_t plane, int64_t Wp,
                                       int64_t OH, int64_t OW, int reverse, uint32_t *amax, void *stream);
extern "C" int fpemu_gemm_conv_tile_(int64_t M, int64_t nvalid, int64_t batch);
extern "C" int fpemu_gemm_conv_phase_launch_(float *out, const float *Wm, const float *xp, const float *bias, int64_t batch,
                                             int64_t M, int64_t cblocks, int64_t R, int64_t S, int64_t st, int64_t PH,
                                             int64_t Wh, int64_t OH, int64_t OW, uint32_t *amax, void *stream);
extern "C" int fpemu_gemm_wu_implicit_ok_(const float *gout, int64_t K, int64_t crs, int64_t P, int64_t OW, int64_t kseg_a,
                                          int variant);
extern "C" int fpemu_gemm_wu_implicit_launch_(float *Gm, const float *gout, const float *xp, int64_t K, int64_t crs,
                                              int64_t P, int64_t images, int64_t kseg_a, int64_t kseg_b, int start,
                                              int64_t CB, int64_t R, int64_t S, int64_t plane, int64_t Wp, int64_t OW,
                                              int64_t st, int64_t PH, void *stream);
#ifndef FPEMU_CONV_WU_IMPLICIT
#define FPEMU_CONV_WU_IMPLICIT 1
#endif

namespace {

// out[n][m][oy][ox] (OH x OW) = chain(bias[m]; (cb, kj, ki, c2) ascending: Wm[m][...] * src[n][cb*64+c2][oy+kj-ph][ox+ki'-pw]),
// ki' = ki, or S-1-ki when `reverse` (backward-data walks ki ascending = source column descending),
// for a stride-1 correlation of `chans` (multiple of 64) source channels of Hs x Ws with an R x S filter.
// Returns FPEMU_OK, an error, or 1 when the workspace cannot hold even one image (caller falls back).
int implicit_conv(const char *who, float *out, const float *src, const float *Wm, const float *bias, int64_t N,
                  int64_t chans, int64_t Hs, int64_t Ws, int64_t M, int64_t R, int64_t S, int64_t ph, int64_t pw,
                  int64_t OH, int64_t OW, int reverse, float *xp, int64_t xp_floats, hipStream_t s, int pack,
                  const PackJob &job, uint32_t *amax = nullptr) {
  // pack: 1 / 2 = the weights are not packed yet (forward / backward-data layout); done in the first launch here
  const int64_t Hp = Hs + 2 * ph, Wp = Ws + 2 * pw;
  const int64_t wtotal = pack == 1 ? job.g.K * job.g.C * job.g.R * job.g.S : pack == 2 ? job.g.C * job.g.K * job.g.R * job.g.S : 0;
  auto pack_alone = [&]() {
    if (pack == 1) hipLaunchKernelGGL(pack_w_fp, dim3(blocks_for(wtotal)), dim3(NT), 0, s, job.w, job.Wm, job.g);
    else if (pack == 2) hipLaunchKernelGGL(pack_w_bp, dim3(blocks_for(wtotal)), dim3(NT), 0, s, job.w, job.Wm, job.g);
  };
  // no padding (1 x 1 filters, ...) and whole tiles: the source planes ARE the operand, nothing is read past them
  if (ph == 0 && pw == 0 && R == 1 && S == 1 && N <= 65535 && (OH * Wp) % fpemu_gemm_conv_tile_(M, OH * Wp, N) == 0) {
    pack_alone();
    CHECK_LAUNCH(who);
    const int rc = fpemu_gemm_conv_launch_(out, Wm, src, bias, N, M, chans / 64, R, S, Hp * Wp, Wp, OH, OW, reverse, amax, (void *)s);
    if (rc != 0) return fail(who, rc, hipGetErrorString((hipError_t)rc));
    return FPEMU_OK;
  }
  int64_t chunk = N;
  while (chunk > 1 && implicit_floats(chunk, chans, Hp, Wp, R, S) > xp_floats) chunk = (chunk + 1) / 2;
  if (implicit_floats(chunk, chans, Hp, Wp, R, S) > xp_floats) return 1;
  if (chunk > 65535) chunk = 65535;
  for (int64_t n0 = 0; n0 < N; n0 += chunk) {
    const int64_t nb = (N - n0) < chunk ? (N - n0) : chunk;
    const unsigned pb = blocks_for((nb * chans * Hp * Wp + 3) / 4);
    const unsigned wb = (n0 == 0 && pack != 0) ? blocks_for(wtotal) : 0u;  // the weights ride on the first chunk's launch
    hipLaunchKernelGGL(wb == 0 ? pad_planes<0> : pack == 1 ? pad_planes<1> : pad_planes<2>, dim3(pb + wb), dim3(NT), 0, s,
                       src + n0 * chans * Hs * Ws, xp, nb * chans, (int)Hs, (int)Ws, (int)Hp, (int)Wp, (int)ph, (int)pw, pb,
                       job);
    CHECK_LAUNCH(who);
    const int rc = fpemu_gemm_conv_launch_(out + n0 * M * OH * OW, Wm, xp, bias, nb, M, chans / 64, R, S, Hp * Wp, Wp,
                                           OH, OW, reverse, amax, (void *)s);
    if (rc != 0) return fail(who, rc, hipGetErrorString((hipError_t)rc));
  }
  return FPEMU_OK;
}

// Forward convolution with stride 2..4 over phase planes (see phase_planes).  Returns FPEMU_OK, an error, or 1 when the
// workspace cannot hold one image's planes (nothing launched; the caller takes the gathered form).
int implicit_conv_strided(const char *who, float *out, const float *src, const float *Wm, const float *bias, int64_t N,
                          const Geo &g, float *xp, int64_t xp_floats, hipStream_t s, const PackJob &job, uint32_t *amax) {
  const int64_t st = g.stride, Hp = g.H + 2 * g.pad, Wp = g.W + 2 * g.pad;
  const int64_t Hh = (Hp + st - 1) / st, Wh = (Wp + st - 1) / st, PH = Hh * Wh, chan = st * st * PH;
  const int64_t slack = 128 + (g.R / st + 2) * Wh + g.S + 4;   // read-ahead of the last tile past the last plane
  int64_t chunk = N;
  while (chunk > 1 && chunk * g.C * chan + slack > xp_floats) chunk = (chunk + 1) / 2;
  if (chunk * g.C * chan + slack > xp_floats) return 1;
  if (chunk > 65535) chunk = 65535;
  const int64_t wtotal = g.K * g.C * g.R * g.S;
  for (int64_t n0 = 0; n0 < N; n0 += chunk) {
    const int64_t nb = (N - n0) < chunk ? (N - n0) : chunk;
    const unsigned wb = n0 == 0 ? blocks_for(wtotal) : 0u;   // the weights ride on the first chunk's launch
    if (st == 2) {
      const unsigned pb = blocks_for(nb * g.C * 2 * Hh * ((Wh + 3) / 4));
      hipLaunchKernelGGL(wb == 0 ? phase_planes_s2<0> : phase_planes_s2<1>, dim3(pb + wb), dim3(NT), 0, s,
                         src + n0 * g.C * g.H * g.W, xp, nb * g.C, (int)g.H, (int)g.W, (int)Hh, (int)Wh, (int)g.pad, (int)g.pad,
                         pb, job);
    } else {
      const unsigned pb = blocks_for((nb * g.C * chan + 3) / 4);
      hipLaunchKernelGGL(wb == 0 ? phase_planes<0> : phase_planes<1>, dim3(pb + wb), dim3(NT), 0, s,
                         src + n0 * g.C * g.H * g.W, xp, nb * g.C, (int)g.H, (int)g.W, (int)Hh, (int)Wh, (int)st, (int)g.pad,
                         (int)g.pad, pb, job);
    }
    CHECK_LAUNCH(who);
    const int rc = fpemu_gemm_conv_phase_launch_(out + n0 * g.K * g.OH * g.OW, Wm, xp, bias, nb, g.K, g.C / 64, g.R, g.S, st,
                                                 PH, Wh, g.OH, g.OW, amax, (void *)s);
    if (rc != 0) return fail(who, rc, hipGetErrorString((hipError_t)rc));
  }
  return FPEMU_OK;
}

#ifndef FPEMU_CONV_IMPLICIT
#define FPEMU_CONV_IMPLICIT 1
#endif
#ifndef FPEMU_CONV_IMPLICIT_STRIDED
#define FPEMU_CONV_IMPLICIT_STRIDED 1
#endif

}  // namespace

extern "C" {

uint64_t fpemu_conv2d_workspace_bytes(int kind, int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                                      int64_t R, int64_t S, int64_t stride, int64_t pad) {
  Geo g;
  if (kind < FPEMU_CONV_FP || kind > FPEMU_CONV_WU) return 0;
  if (make_geo("fpemu_conv2d_workspace_bytes", g, N, C, H, W, K, R, S, stride, pad) != FPEMU_OK) return 0;
  int64_t wf, per, per0;
  ws_layout(kind, g, 1, wf, per);
  ws_layout(kind, g, 0, wf, per0);   // (the call's variant is not known here: the larger of the two layouts)
  if (per0 > per) per = per0;
  int64_t imgs = per > 0 ? kColsCapFloats / per : N;
  if (imgs < 1) imgs = 1;
  if (imgs > N) imgs = N;
  // (+ the read-ahead slack of the implicit path's last tile, so a 1 x 1 convolution's planes fit whole)
  return (uint64_t)(wf + imgs * per + 128 + R * (W + 2 * pad + R) + S + 4) * 4u;
}

static int conv2d_fp_impl(const char *who, float *out, const float *in, const float *w, const float *bias, int64_t N,
                          int64_t C, int64_t H, int64_t W, int64_t K, int64_t R, int64_t S, int64_t stride, int64_t pad,
                          int variant, void *ws, uint64_t ws_bytes, uint32_t *amax, void *stream) {
  Geo g;
  int rc = make_geo(who, g, N, C, H, W, K, R, S, stride, pad);
  if (rc != FPEMU_OK) return rc;
  if (N == 0) return FPEMU_OK;
  if (!out || !in || !w || !ws) return fail(who, FPEMU_EINVAL, "null buffer");
  if ((uintptr_t)ws % 16u) return fail(who, FPEMU_EALIGN, "workspace not 16-byte aligned");
  int64_t wf, per;
  ws_layout(FPEMU_CONV_FP, g, variant, wf, per);
  const int64_t chunk = chunk_images(g, per, (int64_t)(ws_bytes / 4) - wf);
  if ((int64_t)(ws_bytes / 4) < wf) return fail(who, FPEMU_EINVAL, "workspace too small (fpemu_conv2d_workspace_bytes)");
  hipStream_t s = (hipStream_t)stream;
  float *Wm = (float *)ws, *cols = Wm + wf;
  const int64_t P = g.OH * g.OW;
  if (FPEMU_CONV_IMPLICIT && variant == 1 && stride == 1 && C % 64 == 0 && K % 64 == 0) {
    rc = implicit_conv(who, out, in, Wm, bias, N, C, H, W, K, R, S, pad, pad, g.OH, g.OW, 0, cols,
                       (int64_t)(ws_bytes / 4) - wf, s, 1, PackJob{w, Wm, g}, amax);
    if (rc != 1) return rc;  // 1: workspace too small for the padded planes (nothing launched) -> the gathered form below
  }
  // Strides 2..4 with a filter of more than one tap (ResNet-50's 3 x 3 stride-2 layers): phase planes instead of the
  // gathered operand -- one read + one write of the input where the gather writes R*S / stride^2 of it and the GEMM
  // reads that back.  (A 1 x 1 filter's gathered operand IS its one phase plane: nothing to gain.)
  static const bool no_strided = getenv("FPEMU_CONV_FP_GATHER") != nullptr;   // tuning aid (tools/conv_bench.py)
  if (FPEMU_CONV_IMPLICIT_STRIDED && !no_strided && variant == 1 && stride > 1 && stride <= 4 && R * S > 1 && C % 64 == 0 &&
      K % 64 == 0) {
    rc = implicit_conv_strided(who, out, in, Wm, bias, N, g, cols, (int64_t)(ws_bytes / 4) - wf, s, PackJob{w, Wm, g}, amax);
    if (rc != 1) return rc;
  }
  if (chunk < 1) return fail(who, FPEMU_EINVAL, "workspace too small (fpemu_conv2d_workspace_bytes)");
  const int64_t kp = g.KP;  // chain length with its zero tail
  const int64_t Pp = fp_pitch(g, variant);   // pixels per gathered row with their zero tail
  hipLaunchKernelGGL(pack_w_fp, dim3(blocks_for(K * kp)), dim3(NT), 0, s, w, Wm, g);
  CHECK_LAUNCH(who);
  // the chains start at bias[k] (+0 when channel padding exists in the reference: turns a -0 bias into +0), or at +0
  const int start = bias == nullptr ? 3 : C % 64 != 0 ? 2 : 1;
  for (int64_t n0 = 0; n0 < N; n0 += chunk) {
    const int64_t nb = (N - n0) < chunk ? (N - n0) : chunk;
    launch_cols_fp(in + n0 * C * H * W, cols, g, nb * kp, kp, Pp, stride, s);   // (kp is a multiple of 32)
    CHECK_LAUNCH(who);
    rc = fpemu_gemm_chain_start_launch_(out + n0 * K * P, Wm, cols, K, Pp, kp, kp, Pp, P, 0, 0, variant, nb, 0, kp * Pp,
                                        K * P, 1, 0, 0, start, bias, Pp != P ? P : 0, amax, stream);
    if (rc != FPEMU_OK) return rc;
  }
  return FPEMU_OK;
}

int fpemu_conv2d_fp_f32(float *out, const float *in, const float *w, const float *bias, int64_t N, int64_t C,
                        int64_t H, int64_t W, int64_t K, int64_t R, int64_t S, int64_t stride, int64_t pad,
                        int variant, void *ws, uint64_t ws_bytes, void *stream) {
  return conv2d_fp_impl("fpemu_conv2d_fp_f32", out, in, w, bias, N, C, H, W, K, R, S, stride, pad, variant, ws, ws_bytes,
                        nullptr, stream);
}

int fpemu_conv2d_fp_f32_amax(float *out, const float *in, const float *w, const float *bias, int64_t N, int64_t C,
                             int64_t H, int64_t W, int64_t K, int64_t R, int64_t S, int64_t stride, int64_t pad,
                             int variant, void *ws, uint64_t ws_bytes, uint32_t *amax_bits, void *stream) {
  const char *who = "fpemu_conv2d_fp_f32_amax";
  if (amax_bits == nullptr) return fail(who, FPEMU_EINVAL, "null amax word");
  if ((uintptr_t)amax_bits % 4u) return fail(who, FPEMU_EALIGN, "amax word not 4-byte aligned");
  return conv2d_fp_impl(who, out, in, w, bias, N, C, H, W, K, R, S, stride, pad, variant, ws, ws_bytes, amax_bits, stream);
}

int fpemu_conv2d_bp_f32(float *gin, const float *gout, const float *w, int64_t N, int64_t C, int64_t H,
                        int64_t W, int64_t K, int64_t R, int64_t S, int64_t stride, int64_t pad, int variant,
                        void *ws, uint64_t ws_bytes, void *stream) {
  const char *who = "fpemu_conv2d_bp_f32";
  Geo g;
  int rc = make_geo(who, g, N, C, H, W, K, R, S, stride, pad);
  if (rc != FPEMU_OK) return rc;
  if (N == 0) return FPEMU_OK;
  if (!gin || !gout || !w || !ws) return fail(who, FPEMU_EINVAL, "null buffer");
  if ((uintptr_t)ws % 16u) return fail(who, FPEMU_EALIGN, "workspace not 16-byte aligned");
  int64_t wf, per;
  ws_layout(FPEMU_CONV_BP, g, variant, wf, per);
  const int64_t chunk = chunk_images(g, per, (int64_t)(ws_bytes / 4) - wf);
  if ((int64_t)(ws_bytes / 4) < wf) return fail(who, FPEMU_EINVAL, "workspace too small (fpemu_conv2d_workspace_bytes)");
  hipStream_t s = (hipStream_t)stream;
  float *Wm = (float *)ws, *cols = Wm + wf;
  const int64_t HW = H * W, krs = K * R * S, P = g.OH * g.OW;
  if (C <= 4 && N <= 65535 && (H + stride) * stride <= 65535 && (variant == 0 || variant == 1) &&
      K * g.OH * g.OW <= 0x7FFFFFFF && C * H * W <= 0x7FFFFFFF && C * krs <= 0x7FFFFFFF) {
    const int64_t per_phase = (W + stride - 1) / stride;
    const unsigned bs = per_phase <= 64 ? 64u : per_phase <= 128 ? 128u : (unsigned)NT;  // few idle lanes per row
    const int64_t row_groups = ((H + stride - 1) / stride + kFewRows - 1) / kFewRows;   // per residue of y mod stride
    const dim3 grid((unsigned)((per_phase + bs - 1) / bs), (unsigned)(row_groups * stride * stride), (unsigned)N);
    const FewGeo fg{(int)stride, (int)pad, (int)R, (int)S, (int)g.OH, (int)g.OW, (int)K, (int)H, (int)W, (int)g.KB};
    hipLaunchKernelGGL(pack_w_few, dim3(blocks_for(C * krs)), dim3(NT), 0, s, w, Wm, g);
    hipLaunchKernelGGL(variant == 1 ? pick_bp_few<true>((int)C) : pick_bp_few<false>((int)C), grid, dim3(bs), 0, s, gout, Wm,
                       gin, fg);
    CHECK_LAUNCH(who);
    return FPEMU_OK;
  }
  // stride 1: backward-data is the correlation of gout, zero-padded by R-1-pad / S-1-pad, with the packed
  // (flipped) weights: (ob, tj, ti, o2) ascending = the reference's (ob, oj, ki, o2) order
  if (FPEMU_CONV_IMPLICIT && variant == 1 && stride == 1 && C % 64 == 0 && K % 64 == 0 && R - 1 - pad >= 0 &&
      S - 1 - pad >= 0) {
    rc = implicit_conv(who, gin, gout, Wm, nullptr, N, K, g.OH, g.OW, C, R, S, R - 1 - pad, S - 1 - pad, H, W, 1, cols,
                       (int64_t)(ws_bytes / 4) - wf, s, 2, PackJob{w, Wm, g});
    if (rc != 1) return rc;
  }
  if (chunk < 1) return fail(who, FPEMU_EINVAL, "workspace too small (fpemu_conv2d_workspace_bytes)");
  if (bp_by_phase(g)) {
    int64_t planes_per_img, gathered_per_img;
    bp_phase_floats(g, planes_per_img, gathered_per_img);
    const int st = (int)stride;
    PhaseSet ps;
    memset(&ps, 0, sizeof(ps));
    int slot_of[kMaxPhaseStride * kMaxPhaseStride];   // phase (py * st + px) -> its slot in ps, -1: no tap / no pixel
    int64_t wtot = 0, gper = 0, wmax = 0, pitch_max = 0, rows_max = 0;
    for (int p = 0; p < st * st; p++) {
      const Phase ph = make_phase(g, p / st, p % st);
      const int64_t n = C * K * ph.nj * ph.ni;
      slot_of[p] = -1;
      if (ph.Hh * ph.Wh == 0 || n == 0) continue;
      const int q = ps.count++;
      slot_of[p] = q;
      ps.ph[q] = ph;
      ps.woff[q] = wtot;    // the phases' weights side by side in the weight area (their taps partition the filter)
      ps.goff[q] = gper;    // per image here; scaled by the chunk's image count below
      wtot += n;
      gper += K * ph.nj * ph.ni * ph.pitch;
      wmax = n > wmax ? n : wmax;
      pitch_max = ph.pitch > pitch_max ? ph.pitch : pitch_max;
      rows_max = K * ph.nj * ph.ni > rows_max ? K * ph.nj * ph.ni : rows_max;
    }
    const bool rb4 = g.KB % 4 == 0;   // (then every phase's K * nj * ni is a multiple of 4)
    if (ps.count > 0) {
      hipLaunchKernelGGL(pack_w_bp_phase, dim3(blocks_for(wmax), (unsigned)ps.count), dim3(NT), 0, s, w, Wm, g, ps);
      CHECK_LAUNCH(who);
    }
    for (int64_t n0 = 0; n0 < N; n0 += chunk) {
      const int64_t nb = (N - n0) < chunk ? (N - n0) : chunk;
      float *plane = cols, *gath = cols + nb * planes_per_img;
      PhaseSet pc = ps;
      for (int q = 0; q < ps.count; q++) pc.goff[q] = ps.goff[q] * nb;
      if (ps.count > 0) {
        const int64_t vecs = pitch_max / 4, groups = nb * rows_max / (rb4 ? 4 : 1);
        int lg = 4;   // lanes per row group: the smallest power of two covering the longest phase row's vectors
        while ((1 << lg) < vecs && lg < 8) lg++;
        const bool shortrows = lg < 8;
        dim3 gg = shortrows ? gather_grid(vecs, (groups + (NT >> lg) - 1) / (NT >> lg)) : gather_grid(vecs, groups);
        gg.z = (unsigned)ps.count;
        void (*gk)(const float *, float *, Geo, PhaseSet, int64_t, int) =
            shortrows ? (rb4 ? cols_bp_phase<4, true> : cols_bp_phase<1, true>) : (rb4 ? cols_bp_phase<4, false> : cols_bp_phase<1, false>);
        hipLaunchKernelGGL(gk, gg, dim3(NT), 0, s, gout + n0 * K * P, gath, g, pc, nb, lg);
        CHECK_LAUNCH(who);
      }
      PhaseTable tb;
      for (int p = 0; p < 16; p++) { tb.plane[p] = nullptr; tb.pitch[p] = 0; tb.Wh[p] = 1; }
      // the phases' GEMMs: one grouped launch when they share their shape but for the depth (even H and W, stride 2),
      // else one launch each
      float *gc[4];
      const float *ga[4], *gb[4];
      int64_t gk[4];
      bool same = ps.count >= 2 && ps.count <= 4;
      {
        float *pl = plane;
        for (int p = 0, i = 0; p < st * st; p++) {
          const int q = slot_of[p];
          if (q < 0) continue;
          const Phase &ph = ps.ph[q];
          if (i < 4) { gc[i] = pl; ga[i] = Wm + ps.woff[q]; gb[i] = gath + pc.goff[q]; gk[i] = K * ph.nj * ph.ni; }
          same = same && ph.pitch == ps.ph[0].pitch;
          i++;
          pl += nb * C * ph.pitch;
        }
      }
      int grouped = -100;   // -100: not grouped (the problems do not qualify)
      if (same) {
        grouped = fpemu_gemm_chain_group_launch_(ps.count, gc, ga, gb, gk, C, ps.ph[0].pitch, nb, C * ps.ph[0].pitch, variant,
                                                 stream);
        if (grouped != FPEMU_OK && grouped != -100) return grouped;
      }
      for (int p = 0; p < st * st; p++) {
        const int q = slot_of[p];
        if (q < 0) continue;
        const Phase &ph = ps.ph[q];
        const int64_t krs_p = K * ph.nj * ph.ni;
        if (grouped != FPEMU_OK) {
          rc = fpemu_gemm_chain_start_launch_(plane, Wm + ps.woff[q], gath + pc.goff[q], C, ph.pitch, krs_p, krs_p, ph.pitch,
                                              ph.pitch, 0, 0, variant, nb, 0, krs_p * ph.pitch, C * ph.pitch, 1, 0, 0, 3,
                                              nullptr, 0, nullptr, stream);
          if (rc != FPEMU_OK) return rc;
        }
        tb.plane[p] = plane; tb.pitch[p] = ph.pitch; tb.Wh[p] = ph.Wh;
        plane += nb * C * ph.pitch;
      }
      {
        float *gdst = gin + n0 * C * HW;
        bool quad = st == 2 && W % 4 == 0 && ((uintptr_t)gdst % 16u) == 0;
        for (int p = 0; p < 4 && quad; p++)   // 8-byte loads: every plane row starts on an even element
          quad = tb.plane[p] == nullptr || (tb.pitch[p] % 2 == 0 && tb.Wh[p] % 2 == 0 && ((uintptr_t)tb.plane[p] % 8u) == 0);
        const int64_t units = quad ? W / 4 : (W + st - 1) / st, rows_g = nb * C * H;
        int lg = 0;
        while (lg < 8 && (1 << lg) < units) lg++;
        const int64_t nblk = (rows_g + (NT >> lg) - 1) / (NT >> lg);
        if (nblk > 0x7FFFFFFF) return fail(who, FPEMU_EINVAL, "more than 2^31-1 workgroups");
        void (*ik)(float *, PhaseTable, int64_t, int, int, int, int) = interleave_phases<0, 1>;
        if (quad) ik = interleave_phases<2, 2>;
        else if (st == 2) ik = interleave_phases<2, 1>;
        hipLaunchKernelGGL(ik, dim3((unsigned)nblk), dim3(NT), 0, s, gdst, tb, rows_g, (int)H, (int)W, st, lg);
        CHECK_LAUNCH(who);
      }
    }
    return FPEMU_OK;
  }
  hipLaunchKernelGGL(pack_w_bp, dim3(blocks_for(C * krs)), dim3(NT), 0, s, w, Wm, g);
  CHECK_LAUNCH(who);
  for (int64_t n0 = 0; n0 < N; n0 += chunk) {
    const int64_t nb = (N - n0) < chunk ? (N - n0) : chunk;
    hipLaunchKernelGGL((HW % 4 == 0 && ((uintptr_t)cols % 16u) == 0) ? cols_bp<true> : cols_bp<false>, gather_grid(HW, nb * krs),
                       dim3(NT), 0, s, gout + n0 * K * P, cols, g, nb * krs);
    CHECK_LAUNCH(who);
    rc = fpemu_gemm_chain_start_launch_(gin + n0 * C * HW, Wm, cols, C, HW, krs, krs, HW, HW, 0, 0, variant, nb, 0,
                                        krs * HW, C * HW, 1, 0, 0, 3, nullptr, 0, nullptr, stream);   // chains start at +0
    if (rc != FPEMU_OK) return rc;
  }
  return FPEMU_OK;
}

int fpemu_conv2d_wu_f32(float *gw, const float *gout, const float *in, int64_t N, int64_t C, int64_t H,
                        int64_t W, int64_t K, int64_t R, int64_t S, int64_t stride, int64_t pad, int variant,
                        void *ws, uint64_t ws_bytes, void *stream) {
  const char *who = "fpemu_conv2d_wu_f32";
  Geo g;
  int rc = make_geo(who, g, N, C, H, W, K, R, S, stride, pad);
  if (rc != FPEMU_OK) return rc;
  if (!gw || !ws || (N > 0 && (!gout || !in))) return fail(who, FPEMU_EINVAL, "null buffer");
  if ((uintptr_t)ws % 16u) return fail(who, FPEMU_EALIGN, "workspace not 16-byte aligned");
  int64_t wf, per;
  ws_layout(FPEMU_CONV_WU, g, variant, wf, per);
  const int64_t chunk = N == 0 ? 1 : chunk_images(g, per, (int64_t)(ws_bytes / 4) - wf);
  if (chunk < 1) return fail(who, FPEMU_EINVAL, "workspace too small (fpemu_conv2d_workspace_bytes)");
  hipStream_t s = (hipStream_t)stream;
  float *Gm = (float *)ws, *cols = Gm + wf;
  const int64_t P = g.OH * g.OW, crs = C * R * S, Pp = wu_pitch(g, variant);
  if (N == 0) {  // no image: the gradient is +0
    hipLaunchKernelGGL(fill_zero, dim3(blocks_for(K * crs)), dim3(NT), 0, s, Gm, K * crs);
    CHECK_LAUNCH(who);
  }
  // Stride 1 on the small-tile chain kernels (ResNet-50's 56 x 56 and 28 x 28 layers): no gathered operand.  cols_fp
  // writes R*S shifted copies of every plane (220 MiB for the 24.5 MiB input of the C64 56 x 56 layer, read back by the
  // GEMM: 62 of that call's 446 us); the GEMM's loader waves read the same values from the planes themselves -- zero-padded
  // once (pad_planes, as the implicit forward), or the input as it is when there is no padding (1 x 1 filters).
  const int64_t Hp = H + 2 * pad, Wp = W + 2 * pad;
  static const bool no_implicit = getenv("FPEMU_CONV_WU_GATHER") != nullptr;   // tuning aid (tools/conv_bench.py)
  // Stride 2..4 with more than one tap: the same, over the phase planes of the padded input (phase_planes, as the strided
  // forward): tap (kj, ki) is phase (kj % stride, ki % stride) shifted by (kj / stride, ki / stride).
  const int64_t Hh = (Hp + stride - 1) / stride, Wh = (Wp + stride - 1) / stride, PH = Hh * Wh, chan = stride * stride * PH;
  const bool phases = stride > 1;
  const bool implicit = FPEMU_CONV_WU_IMPLICIT && !no_implicit && N > 0 && Pp == P &&
                        (phases ? (stride <= 4 && R * S > 1 && C * chan <= per) : (pad == 0 || C * Hp * Wp <= per)) &&
                        fpemu_gemm_wu_implicit_ok_(gout, K, crs, P, g.OW, K * P, variant);
  for (int64_t n0 = 0; implicit && n0 < N; n0 += chunk) {
    const int64_t nb = (N - n0) < chunk ? (N - n0) : chunk;
    const float *xp = in + n0 * C * H * W;
    if (phases) {
      if (stride == 2) {
        const unsigned pb = blocks_for(nb * C * 2 * Hh * ((Wh + 3) / 4));
        hipLaunchKernelGGL(phase_planes_s2<0>, dim3(pb), dim3(NT), 0, s, xp, cols, nb * C, (int)H, (int)W, (int)Hh, (int)Wh, (int)pad,
                           (int)pad, pb, PackJob{});
      } else {
        const unsigned pb = blocks_for((nb * C * chan + 3) / 4);
        hipLaunchKernelGGL(phase_planes<0>, dim3(pb), dim3(NT), 0, s, xp, cols, nb * C, (int)H, (int)W, (int)Hh, (int)Wh, (int)stride,
                           (int)pad, (int)pad, pb, PackJob{});
      }
      CHECK_LAUNCH(who);
      rc = fpemu_gemm_wu_implicit_launch_(Gm, gout + n0 * K * P, cols, K, crs, P, nb, K * P, C * chan, n0 == 0 ? 3 : 0, g.CB, R, S,
                                          chan, Wh, g.OW, stride, PH, stream);
      if (rc != FPEMU_OK) return rc;
      continue;
    }
    if (pad != 0) {
      const unsigned pb = blocks_for((nb * C * Hp * Wp + 3) / 4);
      hipLaunchKernelGGL(pad_planes<0>, dim3(pb), dim3(NT), 0, s, xp, cols, nb * C, (int)H, (int)W, (int)Hp, (int)Wp, (int)pad,
                         (int)pad, pb, PackJob{});
      CHECK_LAUNCH(who);
      xp = cols;
    }
    rc = fpemu_gemm_wu_implicit_launch_(Gm, gout + n0 * K * P, xp, K, crs, P, nb, K * P, C * Hp * Wp, n0 == 0 ? 3 : 0, g.CB,
                                        R, S, Hp * Wp, Wp, g.OW, 1, 0, stream);
    if (rc != FPEMU_OK) return rc;
  }
  for (int64_t n0 = 0; !implicit && n0 < N; n0 += chunk) {
    const int64_t nb = (N - n0) < chunk ? (N - n0) : chunk;
    launch_cols_fp(in + n0 * C * H * W, cols, g, nb * crs, crs, Pp, stride, s);
    CHECK_LAUNCH(who);
    const float *ga = gout + n0 * K * P;
    if (Pp != P) {  // rows of gout with a tail of -0 up to whole k tiles (behind the chunk's gathered rows)
      float *gp = cols + nb * crs * Pp;
      hipLaunchKernelGGL(pad_rows, dim3(blocks_for(nb * K * Pp)), dim3(NT), 0, s, ga, gp, nb * K, (int)P, (int)Pp);
      CHECK_LAUNCH(who);
      ga = gp;
    }
    // one chain across the chunk's images in ONE launch: k segments = images (chunks follow in order)
    // (the first chunk's chains start at +0, the later ones where the chunk before left them)
    rc = fpemu_gemm_chain_start_launch_(Gm, ga, cols, K, crs, Pp, Pp, Pp, crs, 0, 1, variant, 1, 0, 0, 0, nb,
                                        K * Pp, crs * Pp, n0 == 0 ? 3 : 0, nullptr, 0, nullptr, stream);
    if (rc != FPEMU_OK) return rc;
  }
  hipLaunchKernelGGL(unpack_gw, dim3(blocks_for(K * crs)), dim3(NT), 0, s, Gm, gw, g);
  CHECK_LAUNCH(who);
  return FPEMU_OK;
}

}  // extern "C"
