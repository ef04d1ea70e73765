// fpemu_rows.hip -- per-row amax -> scale -> cast of a contiguous [rows, cols] buffer in ONE launch.
//
// Replaces three loops of the reference, all "reduce max |x| over a contiguous run, derive a scale,
// cast the run with it":
//   * the Python loop over channels (qutils.py:289-311: per row a torch.max, a host-synchronising scale
//     computation and one cast launch) -- policy kRowsClamp;
//   * the fine-grained loop over out_channels x groups (qutils.py:313-337) -- policy kRowsNoClamp;
//   * block normalisation inside the native kernels (block_norm = true: cpp/fpemu_impl.cpp:778-815,
//     :1149-1190, :1441-1478, :1734-1770, :1840-1860) -- policy kRowsBlockNorm, rows = blocks.
// Policy kRowsGiven skips the reduction: the row's scale is read from a device array (the "mean" policies of
// qutils.py:294-301, whose per-row statistic is torch.mode -- computed by torch, on the device, for all rows at
// once -- and any other caller that already holds one scale per row in device memory).
//
// Team per row (TEAM): a 256-thread block; one wave64, four rows per block (rows of <= kWaveRowCols
// elements: conv filters, block-norm blocks); or a single lane, 256 rows per block (rows of
// <= kLaneRowCols elements -- the reference's emulators call block norm with block_size = 1 on embedding
// tables, hybrid_emu.py:119: one "block" per element).  Pass 1 reduces |x| over the row
// (plain loads, so the row stays in L2 for pass 2), wave64 butterfly (+ one LDS word per wave for block
// teams), every thread derives the row's scale, pass 2 re-reads and casts.  VEC = rows are 16-byte
// aligned (cols % V == 0).  Stochastic noise: row r is "call offset + r", element index = column, exactly
// what a loop of per-row casts with consecutive offsets draws.
//
// Rows that are 16-byte aligned and short enough to sit in registers (<= 8 vectors per lane of a
// 256-thread team: 8192 fp32 / 16384 16-bit elements) take rows_reg_kernel instead: ONE read of the
// row (non-temporal), the maximum reduced over the row's lane group, the held vectors cast and streamed
// out -- a per-row quantisation at the flat cast's traffic.  The lane group is the smallest power of
// two covering the row's vectors (4 lanes for a 16-element fp32 block ... a wave ... the block), so a
// wave's load instruction still covers one contiguous run of rows.
#include "fpemu_common.hpp"

using namespace fpemu;

namespace {

// kRowsAmaxNoClamp / kRowsAmaxClamp: the reduction is skipped as for kRowsGiven, but aux[r] holds row r's AMAX WORD
// (fpemu_amax_rows, possibly all-reduced over a process group) and the scale is derived from it as usual.
enum { kRowsNoClamp = 0, kRowsClamp = 1, kRowsBlockNorm = 2, kRowsGiven = 3, kRowsAmaxNoClamp = 4, kRowsAmaxClamp = 5 };

// scale of a row whose reduction is skipped (policy >= kRowsGiven)
__device__ __forceinline__ float row_scale_given(const uint32_t *aux, uint64_t row, float flt_max, int policy) {
  return policy == kRowsGiven ? u2f(aux[row]) : scale_from_amax(aux[row], flt_max, policy == kRowsAmaxClamp);
}
constexpr uint64_t kWaveRowCols = 1024;
constexpr uint64_t kLaneRowCols = 16;

// |x| pattern for the running max.  Block norm compares floats with `maxval < fabs(x)`, which never
// takes a NaN (cpp:791-793); torch.max (the per-channel callers) propagates it.
template <int DT> __device__ __forceinline__ uint32_t drop_nan(uint32_t w) {
  if constexpr (DT == FPEMU_F32) {
    return ((w & 0x7FFFFFFFu) > 0x7F800000u) ? 0u : w;
  } else {
    constexpr uint32_t INF = DT == FPEMU_F16 ? 0x7C00u : 0x7F80u;
    const uint32_t a = ((w & 0x7FFFu) > INF) ? 0u : (w & 0xFFFFu);
    const uint32_t b = (((w >> 16) & 0x7FFFu) > INF) ? 0u : (w >> 16);
    return a | (b << 16);
  }
}

template <int FMT, int RM, int DT, bool VEC, int TEAM>
__global__ __launch_bounds__(kBlock) void rows_kernel(const void *in_, void *out_, uint64_t rows,
                                                      uint64_t cols, float flt_max, int policy,
                                                      uint64_t seed, uint64_t offset, uint32_t *aux,
                                                      uint64_t col_base) {
  // aux: kRowsGiven reads row r's scale (float bits) from aux[r], kRowsAmax* its amax word; otherwise, when not
  // null, aux[r] receives the amax word.  col_base: index of column 0 in the row's noise stream (a row SEGMENT).
  constexpr unsigned V = Storage<DT>::V;
  constexpr bool SR = (RM == STOCHASTIC);
  static_assert(TEAM == 1 || TEAM == 64 || TEAM == kBlock, "team = lane, wave or block");
  constexpr unsigned TS = (unsigned)TEAM;  // team size
  using elem_t = typename Storage<DT>::elem;
  const unsigned t = TEAM == 1 ? 0u : TEAM == 64 ? (threadIdx.x & 63u) : threadIdx.x;
  const uint64_t row = TEAM == 1    ? (uint64_t)blockIdx.x * kBlock + threadIdx.x
                       : TEAM == 64 ? (uint64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)
                                    : (uint64_t)blockIdx.x;
  if (TEAM != kBlock && row >= rows) return;  // only the block team has a barrier below
  const elem_t *ein = reinterpret_cast<const elem_t *>(in_) + row * cols;
  elem_t *eout = reinterpret_cast<elem_t *>(out_) + row * cols;
  const u32x4 *vin = reinterpret_cast<const u32x4 *>(ein);
  u32x4 *vout = reinterpret_cast<u32x4 *>(eout);
  const uint64_t nvec = (VEC && TEAM != 1) ? cols / V : 0;  // a lone lane walks its short row by element
  const bool bn = (policy == kRowsBlockNorm);
  const bool given = (policy >= kRowsGiven);  // uniform over the launch

  uint32_t m = 0u;
  for (uint64_t i = t; i < (given ? 0u : nvec); i += TS) {
    const u32x4 v = vin[i];
#pragma unroll
    for (int j = 0; j < 4; j++) m = absmax_word<DT>(m, bn ? drop_nan<DT>(v[j]) : v[j]);
  }
  // (a lone lane's loop is unit-stride: keep the loop vectoriser off it, it cannot select the result)
#pragma clang loop vectorize(disable) interleave(disable)
  for (uint64_t e = nvec * V + t; e < (given ? 0u : cols); e += TS) {
    uint32_t w;
    if constexpr (DT == FPEMU_F32) w = f2u(ein[e]);
    else w = (uint32_t)ein[e];
    m = absmax_word<DT>(m, bn ? drop_nan<DT>(w) : w);
  }
  uint32_t r = m;
  if constexpr (TEAM != 1) r = wave_max_u32(m);
  if constexpr (TEAM == kBlock) {
    __shared__ uint32_t part[kWavesPerBlock];
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = r;
    __syncthreads();
    r = part[0];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; w++) r = r > part[w] ? r : part[w];
  }
  if constexpr (DT == FPEMU_F16) r = f2u(h_to_f32(r));
  else if constexpr (DT == FPEMU_BF16) r = r << 16;
  if (aux != nullptr && !given && t == 0) aux[row] = r;
  const float scale = given ? row_scale_given(aux, row, flt_max, policy)
                      : bn  ? blocknorm_scale<FMT>(r) : scale_from_amax(r, flt_max, policy == kRowsClamp);
  const float rcp = scale_reciprocal(scale);
  const uint64_t roff = offset + row;

  const bool vec_noise = (col_base % V) == 0;   // uniform: every vector is one slot of one lane of the noise layout
  for (uint64_t i = t; i < nvec; i += TS) {
    u32x4 o;
    if constexpr (SR) {
      if (vec_noise) o = cast_vector<FMT, RM, DT>(vin[i], scale, rcp, VecNoise<FMT, V>(seed, roff, col_base + i * V));
      else o = cast_vector<FMT, RM, DT>(vin[i], scale, rcp, ElemNoise<FMT, V>{seed, roff, col_base + i * V});
    } else {
      o = cast_vector<FMT, RM, DT>(vin[i], scale, rcp, NoNoise{});
    }
    vout[i] = o;
  }
#pragma clang loop vectorize(disable) interleave(disable)
  for (uint64_t e = nvec * V + t; e < cols; e += TS) {
    uint32_t rnd = 0u;
    if constexpr (SR) rnd = rand_bits_elem<FMT, V>(seed, roff, col_base + e);
    if constexpr (DT == FPEMU_F32) eout[e] = cast_one<FMT, RM>(ein[e], scale, rcp, rnd);
    else eout[e] = (elem_t)store_elem16<DT>(cast_one<FMT, RM>(load_elem16<DT>(ein[e]), scale, rcp, rnd));
  }
}

// Register-resident rows: nv 16-byte vectors per row, a group of G = 1 << lg lanes per row (G <= 256),
// NV >= ceil(nv / G) vectors held per lane, kBlock / G rows per workgroup.  Straight-line code: slots past
// the row (or rows past the end) load a valid dummy address and are masked with selects, so the held
// vectors stay plain registers; no thread leaves early (the block-wide team has a barrier).
constexpr int kRegVecs = 8;
#ifndef FPEMU_ROWS_REG_VECS_F32
#define FPEMU_ROWS_REG_VECS_F32 8   // tuning: 16 lets 256-thread groups hold rows of up to 16384 fp32 elements
#endif
#ifndef FPEMU_ROWS_TIERS
#define FPEMU_ROWS_TIERS 3          // tuning: bit 0 = the 512-thread tier, bit 1 = the 1024-thread tier
#endif
constexpr int reg_vecs(int dtype) { return dtype == FPEMU_F32 ? FPEMU_ROWS_REG_VECS_F32 : kRegVecs; }

// BLOCK = kBlock: the general form.  BLOCK = kBigBlock (1024 threads, 16 waves = the whole CU's register file for
// one row): rows of up to kBigBlock * kBigRegVecs vectors (65536 fp32 / 131072 16-bit elements -- LLM-shaped
// weight rows of 11008 / 14336 / 16384 columns, block norm with blocks of 16 Ki - 64 Ki), G = BLOCK, one row per
// workgroup.  The two-pass rows_kernel read such rows twice, the second time mostly from the Infinity Cache:
// 3.3-3.7 TB/s at 8 B/element where this form streams them once.
constexpr int kBigBlock = 1024;
constexpr int kBigRegVecs = 16;

// 16-bit storage: the row maximum of |x| over a lane's four words as PACKED pairs (v_pk_max_*: one instruction per two
// elements).  keep_nan (torch.max semantics): unsigned maximum of the 15-bit patterns, a NaN pattern is the largest
// and wins.  Otherwise (block norm never takes a NaN, cpp:791-793): patterns are shifted up by 0x7FFF - INF so that
// every NaN pattern turns negative as int16 and a SIGNED maximum against the shifted zero ignores it.
template <int DT> __device__ __forceinline__ uint32_t absmax16_vector(const u32x4 &v, bool keep_nan) {
  constexpr uint32_t INF = DT == FPEMU_F16 ? 0x7C00u : 0x7F80u;
  constexpr uint32_t BIAS = 0x7FFFu - INF;
  uint32_t a[4];
#pragma unroll
  for (int j = 0; j < 4; j++) a[j] = v[j] & 0x7FFF7FFFu;
  if (keep_nan) {
    u16x2 m = __builtin_bit_cast(u16x2, a[0]);
#pragma unroll
    for (int j = 1; j < 4; j++) m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2, a[j]));
    const uint32_t w = __builtin_bit_cast(uint32_t, m);
    const uint32_t lo = w & 0xFFFFu, hi = w >> 16;
    return lo > hi ? lo : hi;
  }
  i16x2 m = i16x2{(int16_t)BIAS, (int16_t)BIAS};
#pragma unroll
  for (int j = 0; j < 4; j++)
    m = __builtin_elementwise_max(m, __builtin_bit_cast(i16x2, __builtin_bit_cast(u16x2, a[j]) + splat16(BIAS)));
  const uint32_t w = __builtin_bit_cast(uint32_t, m);
  const uint32_t lo = w & 0xFFFFu, hi = w >> 16;  // both >= BIAS, below 0x8000
  return (lo > hi ? lo : hi) - BIAS;
}

// R = lane groups (rows) per lane: 2 gives 16-bit storage with short rows a second independent vector in flight per
// lane (measured per lane-group size, see rows_per_lane below; the flat cast went back to one).  Row j of a lane
// is in the j-th run of BLOCK >> lg rows of the workgroup's R * (BLOCK >> lg), so a wave's load still covers one
// contiguous run of rows.
template <int FMT, int RM, int DT, int NV, int BLOCK, int R = 1>
__global__ __launch_bounds__(BLOCK) void rows_reg_kernel(const void *in_, void *out_, uint64_t rows,
                                                         uint32_t nv, uint32_t lg, float flt_max, int policy,
                                                         uint64_t seed, uint64_t offset, uint32_t *aux,
                                                         uint64_t col_base) {
  constexpr unsigned V = Storage<DT>::V;
  constexpr int kWaves = BLOCK / 64;
  const unsigned G = 1u << lg;
  const unsigned t = threadIdx.x & (G - 1u);
  const bool bn = (policy == kRowsBlockNorm);
  const bool given = (policy >= kRowsGiven);  // uniform over the launch: scalar branches

  uint64_t row[R];
  bool live[R];
  u32x4 v[R][NV];
#pragma unroll
  for (int q = 0; q < R; q++) {
    row[q] = ((((uint64_t)blockIdx.x * R + (unsigned)q) * BLOCK) >> lg) + (threadIdx.x >> lg);
    live[q] = row[q] < rows;
    const u32x4 *vin = reinterpret_cast<const u32x4 *>(in_) + (live[q] ? row[q] : 0) * nv;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const uint32_t i = (uint32_t)k * G + t;
      v[q][k] = __builtin_nontemporal_load(&vin[i < nv ? i : 0u]);
    }
  }
  float scale[R];
  if (given) {
#pragma unroll
    for (int q = 0; q < R; q++) scale[q] = row_scale_given(aux, live[q] ? row[q] : 0, flt_max, policy);
  } else {
    uint32_t m[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
      m[q] = 0u;
#pragma unroll
      for (int k = 0; k < NV; k++) {
        const bool ok = live[q] && (uint32_t)k * G + t < nv;
        uint32_t mk = 0u;
        if constexpr (DT == FPEMU_F32) {
#pragma unroll
          for (int j = 0; j < 4; j++) mk = absmax_word<DT>(mk, bn ? drop_nan<DT>(v[q][k][j]) : v[q][k][j]);
        } else {
          mk = absmax16_vector<DT>(v[q][k], !bn);
        }
        mk = ok ? mk : 0u;
        m[q] = m[q] > mk ? m[q] : mk;
      }
    }
    // butterfly over the row's lanes (G is uniform: scalar branches)
#pragma unroll
    for (int q = 0; q < R; q++) m[q] = group_max_u32(m[q], G < 64u ? G : 64u);
    if (G == (unsigned)BLOCK) {
      __shared__ uint32_t part[R][kWaves];
      if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
        for (int q = 0; q < R; q++) part[q][threadIdx.x >> 6] = m[q];
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < R; q++) {
        uint32_t r = part[q][0];
#pragma unroll
        for (int w = 1; w < kWaves; w++) r = r > part[q][w] ? r : part[q][w];
        m[q] = r;
      }
    }
#pragma unroll
    for (int q = 0; q < R; q++) {
      uint32_t r = m[q];
      if constexpr (DT == FPEMU_F16) r = f2u(h_to_f32(r));
      else if constexpr (DT == FPEMU_BF16) r = r << 16;
      if (aux != nullptr && t == 0u && live[q]) aux[row[q]] = r;
      scale[q] = bn ? blocknorm_scale<FMT>(r) : scale_from_amax(r, flt_max, policy == kRowsClamp);
    }
  }

  constexpr bool SR = (RM == STOCHASTIC);   // col_base % V == 0 here (the host sends other segments to rows_kernel): every
                                            // vector is one slot of one lane of the noise layout, one draw per vector
#pragma unroll
  for (int q = 0; q < R; q++) {
    const float rcp = scale_reciprocal(scale[q]);
    const uint64_t roff = offset + row[q];
    u32x4 *vout = reinterpret_cast<u32x4 *>(out_) + row[q] * nv;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const uint32_t i = (uint32_t)k * G + t;
      u32x4 o;
      if constexpr (SR) {
        o = cast_vector<FMT, RM, DT>(v[q][k], scale[q], rcp, VecNoise<FMT, V>(seed, roff, col_base + (uint64_t)i * V));
      } else {
        o = cast_vector<FMT, RM, DT>(v[q][k], scale[q], rcp, NoNoise{});
      }
      if (live[q] && i < nv) store_stream(&vout[i], o);
    }
  }
}

typedef void (*rows_reg_fn)(const void *, void *, uint64_t, uint32_t, uint32_t, float, int, uint64_t, uint64_t,
                            uint32_t *, uint64_t);

template <int DT, int NV, int BLOCK = kBlock, int RPL = 1> rows_reg_fn pick_rows_reg(int fmt, int rm) {
#define FPEMU_CASE(F, R) \
  if (fmt == F && rm == R) return rows_reg_kernel<F, R, DT, NV, BLOCK, RPL>;
  FPEMU_CASE(E5M2, RTZ) FPEMU_CASE(E5M2, RNE) FPEMU_CASE(E5M2, STOCHASTIC) FPEMU_CASE(E5M2, RNAZ)
  FPEMU_CASE(E5M2, RNTZ) FPEMU_CASE(E5M2, RPINF) FPEMU_CASE(E5M2, RNINF)
  FPEMU_CASE(E5M2_DAZ, RNE) FPEMU_CASE(E5M2_DAZ, STOCHASTIC) FPEMU_CASE(E5M2_DAZ, RNAZ)
  FPEMU_CASE(E5M2_DAZ, RNTZ)
  FPEMU_CASE(E4M3, RTZ) FPEMU_CASE(E4M3, RNE) FPEMU_CASE(E4M3, STOCHASTIC)
  FPEMU_CASE(E3M4, RTZ) FPEMU_CASE(E3M4, RNE) FPEMU_CASE(E3M4, STOCHASTIC)
  FPEMU_CASE(E4M3_IEEE, RNE) FPEMU_CASE(E4M3_IEEE, STOCHASTIC)
  FPEMU_CASE(FP4, RNE)
#undef FPEMU_CASE
  return nullptr;
}

#ifndef FPEMU_ROWS_PAIR16
#define FPEMU_ROWS_PAIR16 1   // tuning: 16-bit storage, one vector per lane, lane group within a wave: two rows per lane
#endif
// rocprofv3 kernel durations, bf16 E4M3_RNE block norm over 64 Mi elements (tools/gpu.sh rows-trace), two rows per lane
// vs one: blocks of 8 (one lane per row) 44.9 vs 42.7 us, 32: 44.3 vs 44.3, 128: 44.9 vs 45.8, 512: 45.0 vs 48.1 --
// the second vector in flight pays once the row's maximum crosses lanes, so groups of 8 lanes and more take it.
constexpr int rows_per_lane(int dtype, uint32_t per_lane, uint32_t lg) {
  return (FPEMU_ROWS_PAIR16 && dtype != FPEMU_F32 && per_lane <= 1 && lg >= 3 && lg <= 6) ? 2 : 1;
}

template <int DT> rows_reg_fn pick_rows_reg_nv(uint32_t per_lane, uint32_t lg, int fmt, int rm) {
  if (rows_per_lane(DT, per_lane, lg) == 2) return pick_rows_reg<DT, 1, kBlock, 2>(fmt, rm);
  return per_lane <= 1 ? pick_rows_reg<DT, 1>(fmt, rm)
       : per_lane <= 2 ? pick_rows_reg<DT, 2>(fmt, rm)
       : per_lane <= 4 ? pick_rows_reg<DT, 4>(fmt, rm)
       : (per_lane <= 8 || DT != FPEMU_F32 || FPEMU_ROWS_REG_VECS_F32 <= 8) ? pick_rows_reg<DT, kRegVecs>(fmt, rm)
                       : pick_rows_reg<DT, (DT == FPEMU_F32 ? FPEMU_ROWS_REG_VECS_F32 : kRegVecs)>(fmt, rm);
}

// 16-bit storage stops at 8 vectors per lane (65536 elements): with 16 the kernel spills (128 VGPRs at 16 waves)
constexpr int big_reg_vecs(int dtype) { return dtype == FPEMU_F32 ? kBigRegVecs : 8; }

template <int DT, int BLOCK> rows_reg_fn pick_rows_big_nv(uint32_t per_lane, int fmt, int rm) {
  if (per_lane <= 4) return pick_rows_reg<DT, 4, BLOCK>(fmt, rm);
  if (per_lane <= 8) return pick_rows_reg<DT, 8, BLOCK>(fmt, rm);
  if constexpr (DT == FPEMU_F32) return pick_rows_reg<DT, kBigRegVecs, BLOCK>(fmt, rm);
  else return nullptr;
}
template <int BLOCK> rows_reg_fn pick_rows_big(int dtype, uint32_t per_lane, int fmt, int rm) {
  return dtype == FPEMU_F32 ? pick_rows_big_nv<FPEMU_F32, BLOCK>(per_lane, fmt, rm)
       : dtype == FPEMU_F16 ? pick_rows_big_nv<FPEMU_F16, BLOCK>(per_lane, fmt, rm)
       : dtype == FPEMU_BF16 ? pick_rows_big_nv<FPEMU_BF16, BLOCK>(per_lane, fmt, rm) : nullptr;
}

// Blocks shorter than one 16-byte vector (the emulators' block_size = 1 on embedding tables,
// hybrid_emu.py:119; 2; 4 for 16-bit storage): one vector per lane as in the flat cast, every element with
// the scale of its own block.  bs divides V; the buffer is a whole number of vectors.
template <int FMT, int RM, int DT>
__global__ __launch_bounds__(kBlock) void rows_subvec_kernel(const void *in_, void *out_, uint64_t nvec,
                                                             uint32_t bs, float flt_max, int policy,
                                                             uint64_t seed, uint64_t offset, const uint32_t *aux) {
  constexpr unsigned V = Storage<DT>::V;
  constexpr bool SR = (RM == STOCHASTIC);
  const uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= nvec) return;
  const bool bn = (policy == kRowsBlockNorm);
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(in_) + idx);

  uint32_t raw[V], a[V];  // storage patterns and their |x| (NaN dropped for block norm)
#pragma unroll
  for (unsigned e = 0; e < V; e++) {
    if constexpr (DT == FPEMU_F32) {
      raw[e] = v[e];
      a[e] = (bn ? drop_nan<DT>(raw[e]) : raw[e]) & 0x7FFFFFFFu;
    } else {
      raw[e] = (v[e / 2u] >> (16u * (e & 1u))) & 0xFFFFu;
      a[e] = (bn ? (drop_nan<DT>(raw[e]) & 0xFFFFu) : raw[e]) & 0x7FFFu;
    }
  }
  uint32_t m[V];
#pragma unroll
  for (unsigned e = 0; e < V; e++) {
    uint32_t r = a[e];
    if (bs >= 2u) r = r > a[e ^ 1u] ? r : a[e ^ 1u];
    if constexpr (V == 8) {
      if (bs >= 4u) {
        const uint32_t p = a[e ^ 2u] > a[e ^ 3u] ? a[e ^ 2u] : a[e ^ 3u];
        r = r > p ? r : p;
      }
    }
    m[e] = r;
  }
  float y[V];
  const unsigned lgbs = (unsigned)__builtin_ctz(bs);  // bs divides V: a power of two
  Philox4 blk_noise = {0u, 0u, 0u, 0u};
#pragma unroll
  for (unsigned e = 0; e < V; e++) {
    uint32_t r = m[e];
    if constexpr (DT == FPEMU_F16) r = f2u(h_to_f32(r));
    else if constexpr (DT == FPEMU_BF16) r = r << 16;
    const uint64_t g = idx * V + e;  // global element
    const float scale = policy >= kRowsGiven ? row_scale_given(aux, g >> lgbs, flt_max, policy)
                        : bn                 ? blocknorm_scale<FMT>(r) : scale_from_amax(r, flt_max, policy == kRowsClamp);
    const float rcp = scale_reciprocal(scale);
    uint32_t rnd = 0u;
    if constexpr (SR) {
      // element c of a block this short is byte c * NB of the block's first Philox draw (rand_bits_elem with
      // idx = c < V: superblock 0, lane 0, slot 0): one draw per block, not per element
      constexpr unsigned NB = noise_bytes<FMT>();
      const unsigned c = e & (bs - 1u);
      if (c == 0u) blk_noise = philox4x32_10(seed, 0u, offset + (g >> lgbs));  // e, bs uniform: scalar branch
      const unsigned pos = c * NB;
      rnd = (philox_word(blk_noise, pos / 4u) >> ((pos % 4u) * 8u)) & (NB == 2 ? 0xFFFFu : 0xFFu);
    }
    float x;
    if constexpr (DT == FPEMU_F32) x = u2f(raw[e]);
    else x = load_elem16<DT>(raw[e]);
    y[e] = cast_one<FMT, RM>(x, scale, rcp, rnd);
  }
  u32x4 o;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if constexpr (DT == FPEMU_F32) o[j] = f2u(y[j]);
    else o[j] = pack_pair16<DT>(y[2 * j], y[2 * j + 1]);
  }
  store_stream(reinterpret_cast<u32x4 *>(out_) + idx, o);
}

typedef void (*rows_subvec_fn)(const void *, void *, uint64_t, uint32_t, float, int, uint64_t, uint64_t, const uint32_t *);

template <int DT> rows_subvec_fn pick_rows_subvec(int fmt, int rm) {
#define FPEMU_CASE(F, R) \
  if (fmt == F && rm == R) return rows_subvec_kernel<F, R, DT>;
  FPEMU_CASE(E5M2, RTZ) FPEMU_CASE(E5M2, RNE) FPEMU_CASE(E5M2, STOCHASTIC) FPEMU_CASE(E5M2, RNAZ)
  FPEMU_CASE(E5M2, RNTZ) FPEMU_CASE(E5M2, RPINF) FPEMU_CASE(E5M2, RNINF)
  FPEMU_CASE(E5M2_DAZ, RNE) FPEMU_CASE(E5M2_DAZ, STOCHASTIC) FPEMU_CASE(E5M2_DAZ, RNAZ)
  FPEMU_CASE(E5M2_DAZ, RNTZ)
  FPEMU_CASE(E4M3, RTZ) FPEMU_CASE(E4M3, RNE) FPEMU_CASE(E4M3, STOCHASTIC)
  FPEMU_CASE(E3M4, RTZ) FPEMU_CASE(E3M4, RNE) FPEMU_CASE(E3M4, STOCHASTIC)
  FPEMU_CASE(E4M3_IEEE, RNE) FPEMU_CASE(E4M3_IEEE, STOCHASTIC)
  FPEMU_CASE(FP4, RNE)
#undef FPEMU_CASE
  return nullptr;
}

typedef void (*rows_fn)(const void *, void *, uint64_t, uint64_t, float, int, uint64_t, uint64_t, uint32_t *, uint64_t);

template <int DT, bool VEC, int TEAM> rows_fn pick_rows(int fmt, int rm) {
#define FPEMU_CASE(F, R) \
  if (fmt == F && rm == R) return rows_kernel<F, R, DT, VEC, TEAM>;
  FPEMU_CASE(E5M2, RTZ) FPEMU_CASE(E5M2, RNE) FPEMU_CASE(E5M2, STOCHASTIC) FPEMU_CASE(E5M2, RNAZ)
  FPEMU_CASE(E5M2, RNTZ) FPEMU_CASE(E5M2, RPINF) FPEMU_CASE(E5M2, RNINF)
  FPEMU_CASE(E5M2_DAZ, RNE) FPEMU_CASE(E5M2_DAZ, STOCHASTIC) FPEMU_CASE(E5M2_DAZ, RNAZ)
  FPEMU_CASE(E5M2_DAZ, RNTZ)
  FPEMU_CASE(E4M3, RTZ) FPEMU_CASE(E4M3, RNE) FPEMU_CASE(E4M3, STOCHASTIC)
  FPEMU_CASE(E3M4, RTZ) FPEMU_CASE(E3M4, RNE) FPEMU_CASE(E3M4, STOCHASTIC)
  FPEMU_CASE(E4M3_IEEE, RNE) FPEMU_CASE(E4M3_IEEE, STOCHASTIC)
  FPEMU_CASE(FP4, RNE)
#undef FPEMU_CASE
  return nullptr;  // BF16 / FP16 take no scale: per-row scaling is meaningless for them
}

template <int DT> rows_fn pick_rows_dt(bool vec, int team, int fmt, int rm) {
  if (team == 1) return pick_rows<DT, false, 1>(fmt, rm);
  if (team == 64) return pick_rows<DT, false, 64>(fmt, rm);  // aligned short rows take rows_reg_kernel
  return vec ? pick_rows<DT, true, kBlock>(fmt, rm) : pick_rows<DT, false, kBlock>(fmt, rm);
}

// Row maxima only: amax_bits[r] = fp32 bit pattern of max |row r| (torch.max semantics: a NaN wins).  The first half of
// a per-channel quantisation whose rows are spread over several GPUs: every rank reduces its row SEGMENTS, the
// vector of words is all-reduced (MAX) over the process group, and the cast reads the words back (kRowsAmax*).
// Same teams as rows_kernel; read-once stream (non-temporal loads).
template <int DT, bool VEC, int TEAM>
__global__ __launch_bounds__(kBlock) void rows_amax_kernel(const void *in_, uint64_t rows, uint64_t cols,
                                                           uint32_t *amax_bits) {
  constexpr unsigned V = Storage<DT>::V;
  constexpr unsigned TS = (unsigned)TEAM;
  using elem_t = typename Storage<DT>::elem;
  const unsigned t = TEAM == 1 ? 0u : TEAM == 64 ? (threadIdx.x & 63u) : threadIdx.x;
  const uint64_t row = TEAM == 1    ? (uint64_t)blockIdx.x * kBlock + threadIdx.x
                       : TEAM == 64 ? (uint64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)
                                    : (uint64_t)blockIdx.x;
  if (TEAM != kBlock && row >= rows) return;  // only the block team has a barrier below
  const elem_t *ein = reinterpret_cast<const elem_t *>(in_) + row * cols;
  const u32x4 *vin = reinterpret_cast<const u32x4 *>(ein);
  const uint64_t nvec = (VEC && TEAM != 1) ? cols / V : 0;
  uint32_t m = 0u;
#pragma unroll 4
  for (uint64_t i = t; i < nvec; i += TS) {
    const u32x4 v = __builtin_nontemporal_load(&vin[i]);
#pragma unroll
    for (int j = 0; j < 4; j++) m = absmax_word<DT>(m, v[j]);
  }
#pragma clang loop vectorize(disable) interleave(disable)
  for (uint64_t e = nvec * V + t; e < cols; e += TS) {
    uint32_t w;
    if constexpr (DT == FPEMU_F32) w = f2u(ein[e]);
    else w = (uint32_t)ein[e];
    m = absmax_word<DT>(m, w);
  }
  uint32_t r = m;
  if constexpr (TEAM != 1) r = wave_max_u32(m);
  if constexpr (TEAM == kBlock) {
    __shared__ uint32_t part[kWavesPerBlock];
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = r;
    __syncthreads();
    r = part[0];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; w++) r = r > part[w] ? r : part[w];
  }
  if constexpr (DT == FPEMU_F16) r = f2u(h_to_f32(r));
  else if constexpr (DT == FPEMU_BF16) r = r << 16;
  if (t == 0) amax_bits[row] = r;
}

typedef void (*rows_amax_fn)(const void *, uint64_t, uint64_t, uint32_t *);

template <int DT> rows_amax_fn pick_rows_amax(bool vec, int team) {
  if (team == 1) return rows_amax_kernel<DT, false, 1>;
  if (team == 64) return vec ? rows_amax_kernel<DT, true, 64> : rows_amax_kernel<DT, false, 64>;
  return vec ? rows_amax_kernel<DT, true, kBlock> : rows_amax_kernel<DT, false, kBlock>;
}

}  // namespace

// Internal to the library (declared in fpemu_hip.hip): one launch of rows_amax_kernel; arguments validated by the caller.
extern "C" int fpemu_rows_amax_launch_(const void *in, uint64_t rows, uint64_t cols, int dtype, uint32_t *amax_bits,
                                       void *stream) {
  const int team = cols > kWaveRowCols ? kBlock : cols > kLaneRowCols ? 64 : 1;
  const bool vec = ((uintptr_t)in % 16u) == 0 && (cols % vec_elems(dtype)) == 0;
  rows_amax_fn fn = dtype == FPEMU_F32 ? pick_rows_amax<FPEMU_F32>(vec, team)
                  : dtype == FPEMU_F16 ? pick_rows_amax<FPEMU_F16>(vec, team)
                  : dtype == FPEMU_BF16 ? pick_rows_amax<FPEMU_BF16>(vec, team) : nullptr;
  if (fn == nullptr) {
    fpemu_set_error_("fpemu_amax_rows: bad dtype id");
    return FPEMU_EINVAL;
  }
  const uint64_t per_block = team == 1 ? (uint64_t)kBlock : team == 64 ? (uint64_t)kWavesPerBlock : 1u;
  const uint64_t blocks = (rows + per_block - 1) / per_block;
  if (blocks > 0x7FFFFFFFull) {
    fpemu_set_error_("fpemu_amax_rows: more than 2^31-1 workgroups");
    return FPEMU_EINVAL;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, in, rows, cols, amax_bits);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fpemu_set_error_(hipGetErrorString(e));
    return (int)e;
  }
  return FPEMU_OK;
}

// Internal to the library (declared in fpemu_hip.hip).  Arguments are validated by the callers except
// for what depends on the kernel choice.  force_block: always one block per row (the per-tensor small
// path: one row of up to 32 Ki elements).  amax_out: policy 3 (kRowsGiven) READS one float scale per row from it;
// policies 4 / 5 (kRowsAmax*) READ one amax word per row; the other policies write the row's amax word there when
// it is not null.  col_base: index of column 0 in each row's noise stream (row segments of a sharded tensor).
extern "C" int fpemu_rows_launch_(const void *in, void *out, uint64_t rows, uint64_t cols, int dtype,
                                  int fmt, int rmode, float flt_max, int policy, uint64_t seed,
                                  uint64_t offset, uint32_t *amax_out, int force_block, uint64_t col_base,
                                  void *stream) {
  const unsigned ve = vec_elems(dtype);
  // blocks shorter than a vector: the flat cast's geometry with a scale per element
  if (cols < ve && (ve % cols) == 0 && (amax_out == nullptr || policy >= kRowsGiven) && col_base == 0 &&
      (((uintptr_t)in | (uintptr_t)out) % 16u) == 0 && ((rows * cols) % ve) == 0) {
    rows_subvec_fn sf = dtype == FPEMU_F32 ? pick_rows_subvec<FPEMU_F32>(fmt, rmode)
                      : dtype == FPEMU_F16 ? pick_rows_subvec<FPEMU_F16>(fmt, rmode)
                      : dtype == FPEMU_BF16 ? pick_rows_subvec<FPEMU_BF16>(fmt, rmode) : nullptr;
    if (sf == nullptr) return -100;  // kNoRowsKernel of fpemu_hip.hip
    const uint64_t nvec = rows * cols / ve;
    const uint64_t nblk = (nvec + kBlock - 1) / kBlock;
    if (nblk > 0x7FFFFFFFull) {
      fpemu_set_error_("fpemu rows: more than 2^31-1 workgroups");
      return FPEMU_EINVAL;
    }
    hipLaunchKernelGGL(sf, dim3((unsigned)nblk), dim3(kBlock), 0, (hipStream_t)stream, in, out, nvec, (uint32_t)cols,
                       flt_max, policy, seed, offset, (const uint32_t *)amax_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      fpemu_set_error_(hipGetErrorString(e));
      return (int)e;
    }
    return FPEMU_OK;
  }
  // register-resident path: aligned rows of at most kBlock * kRegVecs vectors
  // (a row segment whose first column is not a multiple of the vector's elements straddles the noise layout's slots:
  // rows_kernel below keeps the per-element draw for it)
  const bool noise_aligned = rmode != STOCHASTIC || (col_base % ve) == 0;
  if (noise_aligned && (((uintptr_t)in | (uintptr_t)out) % 16u) == 0 && (cols % ve) == 0 && cols / ve <= (uint64_t)kBlock * (unsigned)reg_vecs(dtype)) {
    const uint32_t nv = (uint32_t)(cols / ve);
    // lane group: the smallest power of two covering the row's vectors, up to a wave; longer rows keep the
    // wave (<= 4 vectors per lane) or take the whole block; a lone row is spread over the block
    uint32_t lg = 0;
    // FPEMU_ROWS_WAVE_NV (tuning aid, tools/rows_quick.py): vectors per lane up to which a WAVE keeps a row (default 4)
    static const uint32_t wave_nv_env = [] {
      const char *e = getenv("FPEMU_ROWS_WAVE_NV");
      const int v = e ? atoi(e) : 4;
      return (uint32_t)(v < 1 ? 1 : v > 16 ? 16 : v);
    }();
    const uint32_t wave_nv = wave_nv_env < (uint32_t)reg_vecs(dtype) ? wave_nv_env : (uint32_t)reg_vecs(dtype);
    if (nv > 64u * wave_nv || ((force_block || rows == 1) && nv > 64u)) lg = 8;
    else if (nv > 64u) lg = 6;
    else while ((1u << lg) < nv) lg++;
    const uint32_t per_lane = (nv + (1u << lg) - 1u) >> lg;
    rows_reg_fn rf = dtype == FPEMU_F32 ? pick_rows_reg_nv<FPEMU_F32>(per_lane, lg, fmt, rmode)
                   : dtype == FPEMU_F16 ? pick_rows_reg_nv<FPEMU_F16>(per_lane, lg, fmt, rmode)
                   : dtype == FPEMU_BF16 ? pick_rows_reg_nv<FPEMU_BF16>(per_lane, lg, fmt, rmode) : nullptr;
    if (rf == nullptr) return -100;  // kNoRowsKernel of fpemu_hip.hip
    const uint64_t per_blk = ((uint64_t)kBlock >> lg) * (unsigned)rows_per_lane(dtype, per_lane, lg);
    const uint64_t nblk = (rows + per_blk - 1) / per_blk;
    if (nblk > 0x7FFFFFFFull) {
      fpemu_set_error_("fpemu rows: more than 2^31-1 workgroups");
      return FPEMU_EINVAL;
    }
    hipLaunchKernelGGL(rf, dim3((unsigned)nblk), dim3(kBlock), 0, (hipStream_t)stream, in, out, rows, nv, lg,
                       flt_max, policy, seed, offset, amax_out, col_base);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      fpemu_set_error_(hipGetErrorString(e));
      return (int)e;
    }
    return FPEMU_OK;
  }
  // longer aligned rows, up to kBigBlock * kBigRegVecs vectors: the 512- / 1024-thread register-resident forms
  if (noise_aligned && (((uintptr_t)in | (uintptr_t)out) % 16u) == 0 && (cols % ve) == 0 &&
      cols / ve <= (uint64_t)kBigBlock * (unsigned)big_reg_vecs(dtype)) {
    const uint32_t nv = (uint32_t)(cols / ve);
    // Measured on MI355X (tools/rows_quick.py, 64 Mi elements, TB/s at 8 | 4 B/element; 1024- vs 512-thread groups):
    // fp32 rows of 16384: 5.85 vs 5.75, of 32768: 5.43 vs 5.85; bf16 rows of 32768: 4.7 vs 5.25.  So fp32 takes
    // 512 threads only where 1024 would hold more than four vectors per lane; 16-bit storage wherever they fit.
    const bool mid = (FPEMU_ROWS_TIERS & 1) && nv <= 512u * (unsigned)big_reg_vecs(dtype) &&
                     (dtype != FPEMU_F32 || nv > 4u * (unsigned)kBigBlock || !(FPEMU_ROWS_TIERS & 2));
    if (mid || (FPEMU_ROWS_TIERS & 2)) {
      const uint32_t block = mid ? 512u : (uint32_t)kBigBlock;
      const uint32_t per_lane = (nv + block - 1u) / block;
      rows_reg_fn rf = mid ? pick_rows_big<512>(dtype, per_lane, fmt, rmode) : pick_rows_big<kBigBlock>(dtype, per_lane, fmt, rmode);
      if (rf == nullptr) return -100;  // kNoRowsKernel of fpemu_hip.hip
      if (rows > 0x7FFFFFFFull) {
        fpemu_set_error_("fpemu rows: more than 2^31-1 workgroups");
        return FPEMU_EINVAL;
      }
      hipLaunchKernelGGL(rf, dim3((unsigned)rows), dim3(block), 0, (hipStream_t)stream, in, out, rows, nv,
                         mid ? 9u : 10u, flt_max, policy, seed, offset, amax_out, col_base);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) {
        fpemu_set_error_(hipGetErrorString(e));
        return (int)e;
      }
      return FPEMU_OK;
    }
  }
  const int team = (force_block || rows == 1 || cols > kWaveRowCols) ? kBlock : cols > kLaneRowCols ? 64 : 1;
  // 16-byte accesses only when every lane of the team gets at least one vector; shorter rows go one
  // element per lane (still one coalesced segment per wave, and no idle lanes)
  const bool vec = (((uintptr_t)in | (uintptr_t)out) % 16u) == 0 && (cols % vec_elems(dtype)) == 0 &&
                   cols >= (uint64_t)team * vec_elems(dtype);
  rows_fn fn = dtype == FPEMU_F32 ? pick_rows_dt<FPEMU_F32>(vec, team, fmt, rmode)
             : dtype == FPEMU_F16 ? pick_rows_dt<FPEMU_F16>(vec, team, fmt, rmode)
             : dtype == FPEMU_BF16 ? pick_rows_dt<FPEMU_BF16>(vec, team, fmt, rmode) : nullptr;
  if (fn == nullptr) return -100;  // kNoRowsKernel of fpemu_hip.hip
  const uint64_t per_block = team == 1 ? (uint64_t)kBlock : team == 64 ? (uint64_t)kWavesPerBlock : 1u;
  const uint64_t blocks = (rows + per_block - 1) / per_block;
  if (blocks > 0x7FFFFFFFull) {
    fpemu_set_error_("fpemu rows: more than 2^31-1 workgroups");
    return FPEMU_EINVAL;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, in, out, rows, cols,
                     flt_max, policy, seed, offset, amax_out, col_base);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fpemu_set_error_(hipGetErrorString(e));
    return (int)e;
  }
  return FPEMU_OK;
}
