// Shared definitions of the gather-GEMM translation units (igemm.hip: the 128-row direct-fragment
// kernels, wgrad, dispatch and the C ABI; igemm_gen1.hip; igemm_bwdn.hip).
#pragma once
#include "common.h"

#include <stdlib.h>
#include <type_traits>
#include <utility>

// cache policy of the loads that stream a table once per launch (pair words, mask words, pair lists):
// 0 = default, 2 = non-temporal (A/B builds: -DSPX_AUX_TABLE=2)
#ifndef SPX_AUX_TABLE
#define SPX_AUX_TABLE 0
#endif
// the same for result rows / partial tiles that the launch itself never reads again: 2 = non-temporal
// (igemm_v4 epilogue: cfg 2 step 37.4 -> 33.6 us), 0 in the A/B build
// weight register sets of igemm_v4_kernel's step pipeline (A/B builds: -DSPX_WD=1)
#ifndef SPX_WD
#define SPX_WD 2
#endif
#ifndef SPX_AUX_OUT
#define SPX_AUX_OUT 2
#endif

namespace spx {

struct GemmParams {
  const void *A;          // [n_src, CIN] gathered operand (features / dout)
  const void *B;          // weights; element (k, n, c) at k*strideK + n*strideN + c
  void *out;              // [n_dst, COUT]
  const int32_t *pair;    // [kv, n_dst]: source row for (k, dst row) or -1
  const uint32_t *mask;   // [n_dst] or null
  const int32_t *argsort; // [n_dst] or null
  const void *bias;       // [COUT] or null
  long long strideK, strideN, strideD;   // strideD: stride of the reduction index (1 = contiguous)
  int n_src, n_dst, CIN, COUT, kv;
  int identity_k;         // offset whose pair is the identity, or -1
  int b_reverse;          // use weight slice kv-1-k for offset k (SubM dgrad)
  int tile_order;         // pair / mask are stored in tile order (row t of the tables belongs to
                          // destination row argsort[t]): rows sorted by mask keep coalesced table reads
  // kernel volumes 33 .. 128 run as groups of <= 32 offsets (one mask word each): `pair` and `mask`
  // point at the group's first table row / mask word, kbase is its first offset (weights are
  // addressed with kbase + k, kv stays the whole kernel volume), mask_words the row stride of `mask`;
  // partial sums travel between the launches of a layer in an fp32 [n_dst, COUT] scratch
  int kbase, mask_words;
  float *acc;             // fp32 scratch of a grouped layer, or null
  int acc_mode;           // bit 0: add acc to the result, bit 1: store fp32 into acc instead of `out`
  int act;
  float act_alpha;
  // int8 inference epilogue (igemm_v4_kernel<.., DT = 2, ..>): bias is fp32 here
  const float *scale;     // [COUT] per-channel multiplier of the i32 accumulator, or null
  const void *add;        // int8 [n_dst, COUT] residual input, or null
  float add_scale;
  int out_dtype;          // SPX_I8 / SPX_F16 / SPX_BF16 / SPX_F32
  int dbg;                // ablation builds only (-DSPX_ABLATE, tools/dense_probe.py)
  int xcd_rot;            // blocks of the launch ahead of this kernel body's first one, mod 8 (fused backward)
  int lpt;                // tables in tile order AND more tiles than resident workgroups: longest tiles first
  // rows layout of a SubM rulebook (spx_subm_layout, tile_order = SPX_ROWS_LAYOUT): the launch walks the rows in their
  // own order with `mask` = the blob's MAIN mask words (zero for the rows that moved to the appendix) and `pair` = the
  // caller's row-order table; `cls` points at the blob (class word, M), `argsort` at the appendix' row list, behind
  // which its mask words and pair table lie.  `mask_rows`: the caller's row-order mask words, for the paths that
  // take no layout (generic kernels).
  const int32_t *cls;
  const uint32_t *mask_rows;
  int app_rows;           // > 0: the HOST knows an upper bound of M (SPX_SPARSE_HINT): only that many appendix rows get
                          // workgroups instead of the n / 4 the class rule allows
  int app_budget;         // (device side) > 0: the appendix' rows are dealt to at most this many workgroups
  int dense_hint;         // the caller knows the neighbourhoods are dense (SPX_DENSE_HINT in tile_order): forward and
                          // dgrad take the weight-stationary kernel (igemm_ws.hip) where its shape limits allow
  // BatchNorm statistics of the rows the launch stores (spx_igemm_fwd_stats): workgroup b leaves {rows, mean, M2} of
  // every output channel at stats[field][channel][b] (bn_record_store below) -- the layout bn_partial_kernel writes
  // (norm.hip), so the normalisation layer behind the convolution starts at its merge step.  n_live: device row count of a static-shape tensor (rows
  // beyond it are padding: not counted), or null.  grid_out (host): receives the launch's workgroup count, 0 when the
  // kernel that was dispatched leaves no statistics.
  float *stats;
  const int32_t *n_live;
  int *grid_out;
};

// appendix workgroups of a fused backward launch, whose dgrad half shares the chip's 1024 workgroup slots with the
// wgrad ranges (wgrad_groups leaves that room); SubM rulebooks from kLayoutMinRows rows on carry a layout (the host
// rule: ops._LAYOUT_MIN_ROWS)
constexpr int kAppBudget = 64;
constexpr int kLayoutMinRows = 32768;
// appendix geometry (the same arithmetic in spx_subm_layout_mcap, rulebook.hip)
__host__ __device__ inline int layout_mcap(int n) { return ((n / 4 + 63) & ~63) + 256; }
// appendix workgroups that lead a launch of TM-row tiles: the class rule (4 M < n) bounds M by n / 4
__host__ __device__ inline int layout_app_tiles(int n, int tm) { return (n / 4 + tm - 1) / tm; }

// tile_order = SPX_ROWS_LAYOUT: the caller passed a layout blob as `argsort` (include/spconv_amd.h)
inline void apply_rows_layout(GemmParams &p, int tile_order) {
  p.cls = nullptr;
  p.mask_rows = nullptr;
  p.app_rows = 0;
  p.app_budget = 0;
  if (tile_order == SPX_ROWS_LAYOUT && p.argsort && p.pair && p.mask) {
    const int32_t *blob = p.argsort;
    const size_t npad = (static_cast<size_t>(p.n_dst) + 63) & ~static_cast<size_t>(63);
    p.cls = blob;
    p.mask_rows = p.mask;
    p.mask = reinterpret_cast<const uint32_t *>(blob + SPX_LAYOUT_HEADER);
    p.argsort = blob + SPX_LAYOUT_HEADER + npad;
    p.tile_order = 0;
  } else {
    p.tile_order = (tile_order == 1 && p.argsort) ? 1 : 0;
    if (tile_order == SPX_ROWS_LAYOUT) p.argsort = nullptr;       // (a blob without tables: row order)
  }
}

// back to the row-order tables (paths that do not read tables by tile position)
inline void drop_rows_layout(GemmParams &p) {
  if (!p.cls) return;
  p.mask = p.mask_rows;
  p.argsort = nullptr;
  p.tile_order = 0;
  p.cls = nullptr;
}

// DT: 0 = f16, 1 = bf16, 2 = int8 (i32 accumulate, quantised epilogue; forward only), 3 = fp32
// (v_mfma_f32_16x16x4_f32, exact fp32 at 1/16 of the 16-bit MFMA rate).  All
// addressing is in BYTES: a step contracts one 128-byte piece of the rows (64 16-bit or 128
// 8-bit reduction elements), a lane feeds 16 bytes per MFMA to either instruction family.
// Kernel arguments: the 16 dwords every wave needs before its first load are separate scalar
// arguments, which -amdgpu-kernarg-preload-count=16 (csrc/build.sh) turns into SGPRs
// preloaded at wave launch -- the two dependent kernarg fetches (~0.5 us each under load) at
// the head of every workgroup's critical path disappear.  The rest travels as a struct.
struct GemmRest {
  void *out;
  const void *bias;
  long long strideK, strideN, strideD;
  int COUT, act;
  float act_alpha;
  const float *scale;
  const void *add;
  float add_scale;
  int out_dtype;
  int dbg;
  float *acc;
  int acc_mode;
  int napp;               // rows layout: appendix workgroups at the head of the grid, or -1 = the n / 4 rule
  float *stats;           // per-workgroup BatchNorm statistics (GemmParams::stats), or null
  const int32_t *n_live;
};

// ---- BatchNorm statistics out of a gather-GEMM epilogue ------------------------------------------------------------
// A lane of the output-stationary kernels ends up with CPL consecutive channels (g * CPL + q, g = lane >> 4) of the rows
// lane & 15 of its m-blocks.  s1 / s2: the lane's sums of the (rounded) output values and their squares over its valid
// rows, nrows its valid rows (counted on g == 0 lanes only).  The 16 lanes of a group are reduced with DPP row
// rotations, the waves of the workgroup through LDS (`lds`: NWAVES * (2 * COUT + 1) floats, free at this point -- the
// caller has put a barrier behind its last read of the weight stages), and thread c < COUT writes channel c's
// {rows, mean, M2} as record `rec` of `nrec` (bn_record_store): bn_partial_kernel's record (norm.hip), merged by
// bn_finalize_kernel.
__device__ __forceinline__ float row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false));
  return x;
}

// ROUND: 0 = the values are stored as they are (fp32), 1 = rounded to f16, 2 = to bf16 (the statistics are those of the
// rounded rows).  One channel at a time -- sum over the lane's m-blocks, rotate, store -- so that the epilogue's
// register footprint stays what it was (arrays of 2 * CPL running sums cost igemm_v4_kernel<32, 2> half its waves).
// Deviations are taken around the WAVE's own mean of the channel (two sweeps over the accumulators, no extra barrier)
// and the waves are merged with Chan's update: no sum-of-squares cancellation however far a channel's mean is from
// zero -- through twelve normalisation layers in fp32 a plain sum(x^2) - sum(x)^2 / n per tile showed as 0.8 % in the
// first layer's weight gradient between two tilings of the same rows.
// Record layout: FIELD-major, then channel, then record -- stats[(f * COUT + c) * nrec + rec], f = 0 rows, 1 mean, 2 M2,
// nrec = the launch's workgroup count.  The merge (norm.hip bn_finalize_kernel: one block per channel) then reads every
// field of its channel as ONE contiguous run of nrec floats; record-major (one [3][COUT] block per workgroup, what this
// was) made each of its loads a 4-byte pick out of a different 64-byte sector: 9.1 us for 3125 records at 16 channels.
// The stores below are 3 scattered dwords per thread, COUT threads per workgroup: fire and forget.
template <int COUT>
__device__ __forceinline__ void bn_record_store(float *__restrict__ stats, int rec, int nrec, int c, float n, float m,
                                                float m2) {
  const size_t g = static_cast<size_t>(nrec);
  stats[static_cast<size_t>(c) * g + rec] = n;
  stats[(static_cast<size_t>(COUT) + c) * g + rec] = m;
  stats[(2 * static_cast<size_t>(COUT) + c) * g + rec] = m2;
}

template <int ROUND>
__device__ __forceinline__ float bn_rounded(float v) {
  if constexpr (ROUND == 1) v = static_cast<float>(static_cast<_Float16>(v));
  if constexpr (ROUND == 2) {
    uint32_t u = __builtin_bit_cast(uint32_t, v);
    u += 0x7fffu + ((u >> 16) & 1u);
    v = __builtin_bit_cast(float, u & 0xffff0000u);
  }
  return v;
}

template <int COUT, int CPL, int MB, int NWAVES, int ROUND, typename Acc>
__device__ __forceinline__ void wg_bn_stats(const Acc (&acc)[CPL / 4][MB], const int (&grow)[MB], int n_live,
                                            float *lds, float *__restrict__ stats, int rec, int nrec) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 15, lgrp = lane >> 4;
  constexpr int W = 2 * COUT + 1;
  bool ok[MB];
  int rows = 0;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    ok[mb] = grow[mb] >= 0 && grow[mb] < n_live;
    rows += ok[mb] ? 1 : 0;
  }
  const float cnt = row16_sum(static_cast<float>(rows));      // rows of this wave (the same in each of its lane groups)
  const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
#pragma unroll
  for (int nb = 0; nb < CPL / 4; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = 0.f;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) a += ok[mb] ? bn_rounded<ROUND>(acc[nb][mb][e]) : 0.f;
      const float mean = row16_sum(a) * inv;
      float b = 0.f;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const float d = bn_rounded<ROUND>(acc[nb][mb][e]) - mean;
        b += ok[mb] ? d * d : 0.f;
      }
      b = row16_sum(b);
      if (lrow == 0) {
        lds[wave * W + lgrp * CPL + nb * 4 + e] = mean;
        lds[wave * W + COUT + lgrp * CPL + nb * 4 + e] = b;
      }
    }
  if (lane == 0) lds[wave * W + 2 * COUT] = cnt;
  __syncthreads();
  if (tid < COUT) {
    float n = 0.f, m = 0.f, m2 = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) {
      const float nb = lds[w * W + 2 * COUT];
      if (nb > 0.f) {
        const float tot = n + nb, d = lds[w * W + tid] - m;
        m += d * (nb / tot);
        m2 += lds[w * W + COUT + tid] + d * d * (n * nb / tot);
        n = tot;
      }
    }
    bn_record_store<COUT>(stats, rec, nrec, tid, n, m, m2);
  }
}

// a workgroup that leaves without rows (appendix workgroups of a dense rulebook, ...): an empty record
template <int COUT>
__device__ __forceinline__ void wg_bn_stats_empty(float *__restrict__ stats, int rec, int nrec) {
  if (threadIdx.x < COUT) bn_record_store<COUT>(stats, rec, nrec, threadIdx.x, 0.f, 0.f, 0.f);
}

// balanced-segment weight gradient (igemm_bwd.h: wgrad_tr_body / wgrad_f32_body; plan: wgrad_plan2_kernel in igemm.hip)
struct Wgrad2Params {
  const void *feat;        // [n_in, C]
  const void *dout;        // [n_out, K]
  float *partial;          // [segment][tile][64*64]
  const int32_t *native;   // [2, kv, n_in]
  const int32_t *num;      // [kv]
  const int32_t *plan2;    // see wgrad_plan2_kernel
  int n_in, n_out, C, K, kv, subm, tiles_c, tiles_k, G;
  int xcd_order;           // ranges are handed out in the plan's XCD-aware order
};

// One translation unit per operand type (the template bodies live in igemm_v4.h / igemm_bwd.h; igemm.hip holds the f16
// instantiations, the generic kernels, the plans and the C ABI): a full rebuild compiles them side by side.
int dispatch_gather_gemm_bf16(const GemmParams &p, hipStream_t s);                                   // igemm_bf16.hip
int dispatch_bwd_bf16(const GemmParams &p, const Wgrad2Params &q, int n_wgrad_blocks, hipStream_t s);
int dispatch_gather_gemm_f32(const GemmParams &p, hipStream_t s);                                    // igemm_f32.hip
int dispatch_bwd_f32(const GemmParams &p, const Wgrad2Params &q, int n_wgrad_blocks, hipStream_t s);
int launch_gather_gemm_int8(const GemmParams &p, bool rows64, hipStream_t s);                        // igemm_i8.hip

// igemm_gen1.hip: first-generation gather-GEMM (tensors beyond 32-bit buffer offsets)
int launch_gather_gemm_gen1(const GemmParams &p, bool bf16, hipStream_t s);
// igemm_ws.hip: weight-stationary gather-GEMM for dense neighbourhoods (forward and dgrad)
bool ws_ok(const GemmParams &p, int dtype);
int launch_gather_gemm_ws(const GemmParams &p, int dtype, hipStream_t s);
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kTileM = 128;   // output rows per workgroup
constexpr int kCK = 64;       // reduction chunk staged per step (elements)
constexpr int kRowBytes = kCK * 2;

// Ablation switch of the measurement build (csrc/build_ablate.sh): which part of a step is left out
// (results are wrong): 1 = weights staged once, 2 = also no per-step barrier, 3 = no MFMAs, 4 = no
// gathered-row loads, 5 = no pair-word loads.  Compiles to nothing in the product build.
#ifdef SPX_ABLATE
#define SPX_ABL(p, v) (SPX_ABLATE == (v))     // compile-time: one library per variant, no branch in the loop
#else
#define SPX_ABL(p, v) false
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kOob = 0x80000000u;      // any offset >= this is out of range for our buffers
constexpr int kRsrcFlags = 0x00020000;      // raw buffer, 32-bit data format

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, static_cast<int>(bytes),
                                           kRsrcFlags);
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// ---- 16-bit <-> float helpers -------------------------------------------
template <bool BF16> __device__ __forceinline__ float to_float(uint16_t v);
template <> __device__ __forceinline__ float to_float<false>(uint16_t v) {
  return static_cast<float>(__builtin_bit_cast(_Float16, v));
}
template <> __device__ __forceinline__ float to_float<true>(uint16_t v) {
  return __builtin_bit_cast(float, static_cast<uint32_t>(v) << 16);
}
template <bool BF16> __device__ __forceinline__ uint16_t from_float(float f);
template <> __device__ __forceinline__ uint16_t from_float<false>(float f) {
  return __builtin_bit_cast(uint16_t, static_cast<_Float16>(f));
}
template <> __device__ __forceinline__ uint16_t from_float<true>(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40u);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even
  return static_cast<uint16_t>(u >> 16);
}

// two floats -> packed 16-bit pair, round to nearest even (v_cvt_pk_{f16,bf16}_f32 on gfx950)
template <bool BF16> __device__ __forceinline__ uint32_t pack2(float a, float b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  if constexpr (BF16) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
  } else {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, f16x2));
  }
}

__device__ __forceinline__ float apply_act(float v, int act, float alpha) {
  if (act == SPX_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == SPX_ACT_LEAKY_RELU) return v > 0.f ? v : v * alpha;
  if (act == SPX_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  return v;
}

template <bool BF16>
__device__ __forceinline__ f32x4 mfma16(const uint4 &a, const uint4 &b, f32x4 c) {
  if constexpr (BF16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
}

__device__ __forceinline__ uint32_t dword_of4(const uint4 &v, int i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// value-wise select (a ?: on the uint4 objects themselves would force them into memory)
__device__ __forceinline__ uint4 sel4(bool ok, const uint4 &v) {
  return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
}

// byte offset of 16-byte slot `slot` of row `row` in a [rows][ROWB bytes] LDS
// tile; the XOR keeps every ds_read_b128 lane group (rows r..r+15, two
// neighbouring slots) on 16 distinct 16-byte bank slots.
__device__ __forceinline__ int swz_off(int row, int slot, int row_bytes, int xmask = 7) {
  return row * row_bytes + ((slot ^ ((row >> 1) & xmask)) << 4);
}

// XCD-aware tile order: workgroup b runs on XCD b % 8 (observed), so give each
// XCD a contiguous range of tiles -> neighbouring tiles (which gather
// overlapping source rows) share one L2.  Bijective for any tile count.
__device__ __forceinline__ int xcd_tile(int bid, int ntiles) {
  const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, j = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + j;
}

// The same for workgroups that are `rot` blocks into a launch (the dgrad tiles of the fused backward
// come after the wgrad ranges): workgroup bid runs on XCD (bid + rot) % 8, and that XCD gets the
// tile range at ITS position, so that it works on the same eighth of the rows as the wgrad ranges
// the plan gave it (wgrad_plan2_kernel) and the two halves share the gradient rows in its L2.
__device__ __forceinline__ int xcd_tile_rot(int bid, int ntiles, int rot) {
  const int q = ntiles >> 3, r = ntiles & 7, cls = bid & 7, j = bid >> 3;   // class cls has q + (cls < r) tiles
  const int phys = (cls + rot) & 7;
  int base = 0;
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    const int c = (y - rot) & 7;                         // the class that runs on XCD y
    if (y < phys) base += q + (c < r ? 1 : 0);
  }
  return base + j;
}

// --------------------------------------------------------------------------
// gather-GEMM, 16-bit operands, fp32 accumulate.
//   out[d, :] = act(bias + sum_k A[pair[k][d], :] . B_k^T)
// --------------------------------------------------------------------------
// Step iterator over (offset k, reduction chunk): k runs over the set bits of the
// tile mask, chunk over ceil(CIN / 64).
struct StepIt {
  int k;           // -1 = end
  int chunk;
  uint32_t rest;   // offsets still to visit after k
  // narrow reduction rows (<= 32 bytes: 16 channels of a 16-bit type, or 8): up to three MORE offsets ride in the same
  // step -- the offset of group g in the 16 (8) reduction positions of lane group(s) g of the MFMA, next to k in group 0
  // (igemm_v4_body, PK); packed into ONE 64-bit word -- byte g - 1 = the offset of group g (up to seven more: eight
  // offsets per step when both 64-byte pieces of a step are packed), 0xff = none -- so that an iterator stays a handful
  // of scalars (three iterators are live in the step loop; separate fields pushed the kernels into scratch).
  unsigned long long kx;
};

// offset of group g (1 .. 7) of a step, or -1
__device__ __forceinline__ int step_k(const StepIt &it, int g) {
  const uint32_t v = static_cast<uint32_t>(it.kx >> (8 * (g - 1))) & 0xffu;
  return v == 0xffu ? -1 : static_cast<int>(v);
}

// T = offsets per step
template <int T = 1>
__device__ __forceinline__ void step_pack(StepIt &it) {
  it.kx = ~0ull;
  if constexpr (T > 1) {
#pragma unroll
    for (int g = 1; g < T; ++g) {
      if (it.k >= 0 && it.rest) {
        it.kx = (it.kx & ~(0xffull << (8 * (g - 1)))) |
                (static_cast<unsigned long long>(__builtin_ctz(it.rest)) << (8 * (g - 1)));
        it.rest &= it.rest - 1;
      }
    }
  }
}

template <int PK = 1>
__device__ __forceinline__ StepIt step_begin(uint32_t bits) {
  StepIt it;
  it.chunk = 0;
  it.k = bits ? __builtin_ctz(bits) : -1;
  it.rest = bits ? (bits & (bits - 1)) : 0u;
  step_pack<PK>(it);
  return it;
}

template <int PK = 1>
__device__ __forceinline__ StepIt step_next(StepIt it, int nchunk) {
  if (it.k < 0) return it;
  if (it.chunk + 1 < nchunk) {
    ++it.chunk;
    return it;
  }
  it.chunk = 0;
  it.k = it.rest ? __builtin_ctz(it.rest) : -1;
  it.rest = it.rest ? (it.rest & (it.rest - 1)) : 0u;
  step_pack<PK>(it);
  return it;
}

// one MFMA of a step: 16 bytes per lane and operand -- v_mfma_f32_16x16x32_{f16,bf16} (8
// elements) or v_mfma_i32_16x16x64_i8 (16 elements)
template <int DT, typename ACC>
__device__ __forceinline__ ACC mfma_step(const uint4 &a, const uint4 &b, ACC c) {
  if constexpr (DT == 2) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a),
                                                 __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
  } else if constexpr (DT == 3) {
    // fp32: four v_mfma_f32_16x16x4_f32 (exact fp32, one element of the 16-byte piece each; the
    // reduction index of element t of lane group g is 4 g + t on BOTH operands)
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int t = 0; t < 4; ++t) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[t], fb[t], c, 0, 0, 0);
    return c;
  } else {
    return mfma16<DT == 1>(a, b, c);
  }
}

// N consecutive dwords of one lane to / from a raw buffer, in the widest pieces
// AUX: cache policy bits of the store (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
template <int N, int AUX = 0>
__device__ __forceinline__ void store_dwords(const uint32_t (&d)[N], __amdgpu_buffer_rsrc_t r,
                                             uint32_t vo) {
  if constexpr (N == 1) {
    __builtin_amdgcn_raw_buffer_store_b32(d[0], r, vo, 0, AUX);
  } else if constexpr (N == 2) {
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{d[0], d[1]}, r, vo, 0, AUX);
  } else {
#pragma unroll
    for (int q = 0; q < N / 4; ++q)
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]},
                                             r, vo + q * 16, 0, AUX);
  }
}
template <int N>
__device__ __forceinline__ void load_dwords(uint32_t (&d)[N], __amdgpu_buffer_rsrc_t r, uint32_t vo) {
  if constexpr (N == 1) {
    d[0] = __builtin_amdgcn_raw_buffer_load_b32(r, vo, 0, 0);
  } else if constexpr (N == 2) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, vo, 0, 0);
    d[0] = v[0];
    d[1] = v[1];
  } else {
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, vo + q * 16, 0, 0);
      d[4 * q] = v[0];
      d[4 * q + 1] = v[1];
      d[4 * q + 2] = v[2];
      d[4 * q + 3] = v[3];
    }
  }
}

}  // namespace
}  // namespace spx
