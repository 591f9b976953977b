// The direct-fragment gather-GEMM (igemm_v4_kernel: forward and dgrad of every operand type), its launcher and the
// 16-bit dispatch rule -- template code shared by igemm.hip (f16), igemm_bf16.hip, igemm_f32.hip and igemm_i8.hip, each of
// which instantiates its own operand type (one 2400-line translation unit took 93 s to compile; split: see build.sh).
#pragma once
#include "igemm_defs.h"

namespace spx {
namespace {

// --------------------------------------------------------------------------
// gather-GEMM v4 ("direct fragments"): same contract as gather_gemm_mfma_kernel.
//  * the gathered operand never touches LDS: every lane loads the 16 bytes it
//    feeds to the MFMA (row = lane & 15 of an m-block, 8 reduction elements
//    selected by lane >> 4) with raw buffer loads; a missing pair (-1) turns into
//    an out-of-range offset, which the buffer unit answers with zeros -- no
//    selects, no wasted traffic;
//  * pair-table words and weight slices are fetched through buffer resources
//    with scalar (SGPR) offsets per step, so the per-step VALU address math is
//    one multiply-add and one min per row;
//  * weight slices go global -> registers -> LDS into a two-stage ring: one
//    __syncthreads() per step;
//  * register pipeline of depth two for the gathered rows (two named register
//    sets, statically indexed), depth three for the pair-table words.
// Limits (checked on the host, v3 handles the rest): n_src * CIN * 2 < 2^31 and
// n_dst * 4 < 2^31 (32-bit buffer offsets, bit 31 reserved for "out of range").
// --------------------------------------------------------------------------
// Optional per-workgroup timeline (debug builds only: -DSPX_TIMELINE, see tools/timeline.py):
// wave 0 of every workgroup stamps s_memtime at fixed points of igemm_v4_kernel into a global
// table, read back through spx_debug_timeline().
#ifdef SPX_TIMELINE
constexpr int kTlSlots = 8, kTlMaxWg = 8192;
__device__ unsigned long long g_timeline[kTlMaxWg * kTlSlots];
#define SPX_STAMP(i)                                                                     \
  do {                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x < kTlMaxWg)                                       \
      g_timeline[blockIdx.x * kTlSlots + (i)] = __builtin_amdgcn_s_memtime();            \
  } while (0)
#else
#define SPX_STAMP(i) do {} while (0)
#endif




template <int COUT, int MB, int DT, bool BT, int NKS = 2, int PK = 1>
__device__ __forceinline__ void igemm_v4_body(const GemmParams &p, int block);
__device__ __forceinline__ void unpack_gemm_args(GemmParams &p, const void *argA, const void *argB,
                                                 const uint32_t *arg_mask, const int32_t *arg_argsort,
                                                 const int32_t *arg_pair, int n_dst, int n_src, int CIN,
                                                 int kv, int identity_k, int b_reverse,
                                                 const GemmRest &rest);

// NKS = 1: rows of at most 64 bytes (16 / 32 16-bit channels): the second half of every 128-byte
// piece is empty, so its loads and MFMAs are not emitted at all -- the dense-scene kernels are
// bound by vector-memory INSTRUCTIONS (16 clocks each in the address unit, whatever the lanes
// fetch), and a dead load costs as much as a live one.
template <int COUT, int MB, int DT, bool BT, int NKS = 2, int PK = 1>
__global__ void __launch_bounds__(kThreads)
igemm_v4_kernel(const void *argA, const void *argB, const uint32_t *arg_mask,
                const int32_t *arg_argsort, const int32_t *arg_pair, int n_dst, int n_src, int CIN,
                int kv, int identity_k, int b_reverse, GemmRest rest) {
  GemmParams p;
  unpack_gemm_args(p, argA, argB, arg_mask, arg_argsort, arg_pair, n_dst, n_src, CIN, kv, identity_k,
                   b_reverse, rest);
  igemm_v4_body<COUT, MB, DT, BT, NKS, PK>(p, blockIdx.x);
}

__device__ __forceinline__ void unpack_gemm_args(GemmParams &p, const void *argA, const void *argB,
                                                 const uint32_t *arg_mask, const int32_t *arg_argsort,
                                                 const int32_t *arg_pair, int n_dst, int n_src, int CIN,
                                                 int kv, int identity_k, int b_reverse,
                                                 const GemmRest &rest) {
  p.A = argA;
  p.B = argB;
  p.mask = arg_mask;
  p.argsort = arg_argsort;
  p.pair = arg_pair;
  p.n_dst = n_dst;
  p.n_src = n_src;
  p.CIN = CIN;
  p.kv = kv;
  p.identity_k = identity_k;
  // the launch packs (reverse, tile order, mask stride, first offset of the group) into one preloaded SGPR
  p.b_reverse = b_reverse & 1;
  p.tile_order = (b_reverse >> 1) & 1;
  p.mask_words = ((b_reverse >> 2) & 3) + 1;
  p.kbase = (b_reverse >> 4) & 127;
  p.lpt = (b_reverse >> 11) & 1;
  p.acc = rest.acc;
  p.acc_mode = rest.acc_mode;
  p.out = rest.out;
  p.bias = rest.bias;
  p.strideK = rest.strideK;
  p.strideN = rest.strideN;
  p.strideD = rest.strideD;
  p.COUT = rest.COUT;
  p.act = rest.act;
  p.act_alpha = rest.act_alpha;
  p.scale = rest.scale;
  p.add = rest.add;
  p.add_scale = rest.add_scale;
  p.out_dtype = rest.out_dtype;
  p.dbg = rest.dbg;
  p.xcd_rot = 0;
  p.stats = rest.stats;
  p.n_live = rest.n_live;
  p.app_rows = rest.napp;      // (the body reads it as a workgroup count)
  p.app_budget = 0;
  // rows layout (bit 12): `mask` = the blob's main mask words, `argsort` = the appendix' row list (its mask words and
  // pair table lie behind it, the class word and M npad + 64 words ahead of it), `pair` = the row-order table
  p.cls = nullptr;
  if ((b_reverse >> 12) & 1) {
    const size_t npad = (static_cast<size_t>(n_dst) + 63) & ~static_cast<size_t>(63);
    p.cls = arg_argsort - npad - SPX_LAYOUT_HEADER;
  }
}

// int8 inference epilogue of a tile whose accumulators lie in acc[nb][mb] (shared by igemm_v4_body and the streaming
// main tiles of igemm_i8_sparse_kernel): lds_sb = [2][COUT] fp32 per-channel scale | bias
template <int COUT, int MB>
__device__ __forceinline__ void i8_epilogue(const GemmParams &p, i32x4 (&acc)[COUT / 16][MB], const int (&grow)[MB],
                                            int lgrp, const float *lds_sb) {
  constexpr int NB = COUT / 16, CPL = NB * 4;
  // int8 inference epilogue (reference numerics: test/test_all_algo.py:272-287):
  //   v = acc_i32 * scale[k] + bias[k] + add[o][k] * add_scale;  v = act(v)
  //   int8 out: clip(round_half_even(v), -128, 127);  f16 / f32 out: v
  const int oes = p.out_dtype == SPX_I8 ? 1 : (p.out_dtype == SPX_F32 ? 4 : 2);
  const __amdgpu_buffer_rsrc_t rO =
      make_rsrc(p.out, static_cast<uint32_t>(p.n_dst) * static_cast<uint32_t>(COUT * oes));
  const __amdgpu_buffer_rsrc_t rAdd =
      make_rsrc(p.add, p.add ? static_cast<uint32_t>(p.n_dst) * COUT : 0u);
  uint32_t rowoff[MB];
  uint32_t addw[MB][CPL / 4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    rowoff[mb] = grow[mb] < 0 ? kOob : static_cast<uint32_t>(grow[mb]) * COUT + lgrp * CPL;
    load_dwords<CPL / 4>(addw[mb], rAdd, rowoff[mb]);   // zeros when there is no residual input
  }
  // four channels (one output dword of an int8 row) at a time keeps the live set small
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const float4 sc4 = *reinterpret_cast<const float4 *>(lds_sb + lgrp * CPL + nb * 4);
    const float4 bv4 = *reinterpret_cast<const float4 *>(lds_sb + COUT + lgrp * CPL + nb * 4);
    const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, bv[4] = {bv4.x, bv4.y, bv4.z, bv4.w};
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      // (uniform conditions hoisted out of the per-value work: at 64 values per lane the epilogue of a
      // tile was ~5.5 us of vector ALU time, more than an identity-only tile's loads and MFMAs)
      float v[4];
#pragma unroll
      // every product and sum rounded on its own: the reference formula is numpy arithmetic,
      // ((acc * scale) + bias) + (add * add_scale), and a fused multiply-add lands on the other side of a
      // rounding tie for ~4 values in 10 million.  (HIP's __fmul_rn is a plain `*` that the compiler is
      // free to contract; the empty asm pins the rounded product in a register.)
      for (int e = 0; e < 4; ++e) {
        float prod = static_cast<float>(acc[nb][mb][e]) * sc[e];
        asm volatile("" : "+v"(prod));
        v[e] = prod + bv[e];
      }
      if (p.add) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int a8 = static_cast<int>(static_cast<int8_t>((addw[mb][nb] >> (e * 8)) & 0xff));
          float prod = static_cast<float>(a8) * p.add_scale;
          asm volatile("" : "+v"(prod));
          v[e] += prod;
        }
      }
      if (p.act == SPX_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      } else if (p.act != SPX_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act, p.act_alpha);
      }
      if (p.out_dtype == SPX_I8) {
        // round half to even, clamp, and pack the four low bytes: two v_cvt_pk_i16_i32 + one v_perm_b32
        int q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          q[e] = static_cast<int>(__builtin_amdgcn_fmed3f(__builtin_rintf(v[e]), -128.f, 127.f));
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        const uint32_t p01 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(q[0], q[1]));
        const uint32_t p23 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(q[2], q[3]));
        const uint32_t word = __builtin_amdgcn_perm(p23, p01, 0x06040200u);
        addw[mb][nb] = word;                            // reuse: the residual word is consumed
      } else if (p.out_dtype == SPX_F32) {
        uint32_t d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = __builtin_bit_cast(uint32_t, v[e]);
        store_dwords<4, SPX_AUX_OUT>(d, rO, rowoff[mb] == kOob ? kOob : (rowoff[mb] + nb * 4) * 4u);
      } else {
        uint32_t d[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
          d[q] = p.out_dtype == SPX_BF16 ? pack2<true>(v[2 * q], v[2 * q + 1])
                                         : pack2<false>(v[2 * q], v[2 * q + 1]);
        store_dwords<2, SPX_AUX_OUT>(d, rO, rowoff[mb] == kOob ? kOob : (rowoff[mb] + nb * 4) * 2u);
      }
    }
  }
  if (p.out_dtype == SPX_I8) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) store_dwords<CPL / 4, SPX_AUX_OUT>(addw[mb], rO, rowoff[mb]);
  }

}

// PK -- offset packing for narrow reduction rows of a 16-bit type.  A step feeds v_mfma_f32_16x16x32 64-byte pieces of
// a row (NKS of them); a row of 32 bytes (16 channels) fills half a piece, one of 16 bytes (8 channels: a 4-channel input
// layer padded) a quarter, and the lanes behind the row's end loaded out-of-range zeros and multiplied them.  With PK they
// gather the row of ANOTHER offset of the tile (every lane reads the pair word of its own offset through one resource
// over the whole table) and read that offset's weights (the stage row holds the slices side by side; forward layout and
// the transposing dgrad layout alike):
//   PK = 2 / 4, NKS = 1   2 / 4 offsets in the one piece of a step (rows <= 32 / 16 bytes)
//   PK = 8 / 16 / 32, NKS = 2   BOTH pieces of a step packed, 1 / 2 / 4 offsets each (rows <= 64 / 32 / 16 bytes):
//                               2 / 4 / 8 offsets per step, two pair words and two rows per lane and m-block
// A tile that meets all 27 offsets walks 1 + 13 / 7 / 4 steps instead of 27 (the identity step stays alone: it runs
// ahead of the mask exchange) -- the same load instructions per offset, a fraction of the steps, barriers and pair-word
// trips (~1 us of kernel time per step and launch on the 400 k-row level of config 4).  Sums of a row associate
// differently than with one offset per step: the same values to fp32 rounding, not bit for bit.
template <int COUT, int MB, int DT, bool BT, int NKS, int PK>
__device__ __forceinline__ void igemm_v4_body(const GemmParams &p, int block) {
  constexpr bool BF16 = DT == 1, I8 = DT == 2, F32 = DT == 3;
  constexpr bool XK = PK >= 8;                          // both pieces of a step packed
  constexpr int PP = XK ? PK / 8 : PK;                  // offsets per 64-byte piece
  constexpr int T = PP * (XK ? 2 : 1);                  // offsets per step
  static_assert(PK == 1 || (!I8 && !F32 && (XK ? NKS == 2 : NKS == 1)), "offset packing: 16-bit operands");
  static_assert(PP == 1 || PP == 2 || PP == 4, "offsets per piece");
  constexpr int ES = I8 ? 1 : (F32 ? 4 : 2);            // bytes per element
  static_assert(!(I8 && BT), "int8 is forward only");
  constexpr int NB = COUT / 16;
  constexpr int TM = 64 * MB;                           // rows per workgroup: 4 waves x MB x 16
  // 16-byte weight vectors staged per thread: [COUT][128 B] row-wise (forward), or the
  // transposing read of dgrad (pairs of reduction rows x 8 channels for 16-bit, one reduction
  // row x 4 channels for fp32)
  constexpr int BROWS = !BT ? (COUT + 31) / 32 : (F32 ? (COUT + 31) / 32 : 2 * ((COUT + 63) / 64));
  // one-element arrays captured by the lambdas below defeat SROA in hipcc 7.2 (the whole
  // parameter block then lives in scratch): keep every register array at >= 2 elements
  constexpr int BA = BROWS < 2 ? 2 : BROWS;
  constexpr int B_BYTES = COUT * kRowBytes;             // one staged weight slice [COUT][64]
  constexpr int CPL = NB * 4;                           // consecutive output channels per lane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t *lds_mask = reinterpret_cast<uint32_t *>(smem + 2 * B_BYTES);  // [4]
  // int8: per-channel scale and bias of the quantised epilogue, staged once per workgroup ([2][COUT] fp32
  // behind the mask words) -- read per lane from memory they were 64 dependent dword loads at the end of
  // every tile
  float *lds_sb = reinterpret_cast<float *>(smem + 2 * B_BYTES + 64);

  SPX_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = (p.n_dst + TM - 1) / TM;
  // Tables in tile order = rows sorted by mask word: the tiles at the END hold the rows with the most
  // offsets (the identity-only rows sort first), and a launch lasts as long as its slowest workgroup.
  // Those tiles are handed to the FIRST blocks (longest work first), one after the other to different
  // XCDs; contiguous per-XCD ranges buy nothing here -- a sorted tile gathers mostly its own rows.
  // (int8 config 5, two dispatch rounds: the 26 us tail tiles no longer start in the second round.)
  // Only when the launch has more tiles than resident workgroups (the host sets `lpt`): a single-round
  // launch keeps the XCD mapping, which lets the dgrad tiles and the wgrad ranges of one row eighth share
  // the gradient rows in one L2 (config 2 backward: 57 vs 67 MB of HBM traffic).
  // Rows layout (spx_subm_layout): the first `napp` workgroups of the launch are APPENDIX tiles -- the rows of a sparse
  // rulebook that have a neighbour, grouped by offset, with their own compact tables; they read {class, M} and leave at
  // once when there is nothing for them (a dense rulebook, or fewer rows than reserved).  They lead the grid because
  // they are the long tiles.  Every other workgroup is a MAIN tile: rows in their own order, masks from the blob -- on a
  // sparse rulebook those only carry the centre bit (a zero word = the row moved to the appendix: nothing stored), so a
  // main tile is ONE step with no row order and no pair word to fetch; on a dense one they are the rulebook's masks.
  const int napp = p.cls ? (p.app_rows >= 0 ? p.app_rows : layout_app_tiles(p.n_dst, TM)) : 0;
  const bool app = block < napp;                                   // (uniform)
  int app_m = 0;
  int tile = 0;
  int pos[MB];                                                     // position of this lane's rows in the tables the
                                                                   // workgroup walks (-1: no row)
  if (app) {
    typedef const int32_t __attribute__((address_space(4))) *cptr_t;
    const int cls = *(cptr_t)(p.cls);
    app_m = *(cptr_t)(p.cls + 1);
    if (!cls) {
      if constexpr (!I8 && !BT) if (p.stats) wg_bn_stats_empty<COUT>(p.stats, block, static_cast<int>(gridDim.x));
      return;
    }
    // The M rows of the appendix are dealt to the launch's napp appendix workgroups in whole 16-row blocks, h rows
    // each: the appendix tiles are the launch's critical path (a walk over every offset any of their rows has, after
    // the main tiles have long finished), and the workgroups reserved for it (n / 4 rows' worth) are there anyway --
    // config 2: 3 165 rows, 25 tiles of 128 rows walk 4.6 offsets on average and 10 at most, 99 tiles of 32 rows
    // 2.7 and 7 (forward 11.2 -> 9.5 us).  A full appendix (M = n / 4) keeps whole tiles.  The fused backward shares
    // the chip's 1024 workgroup slots with the wgrad ranges and deals to at most kAppBudget workgroups (app_budget).
    const int groups = p.app_budget > 0 ? min(napp, p.app_budget) : napp;
    const int h = min(TM, (((app_m + groups - 1) / groups) + 15) & ~15);
    if (block * h >= app_m) {
      if constexpr (!I8 && !BT) if (p.stats) wg_bn_stats_empty<COUT>(p.stats, block, static_cast<int>(gridDim.x));
      return;
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int blk = mb * (kThreads / 64) + (threadIdx.x >> 6);  // 16-row blocks go to the waves round-robin
      const int q = block * h + blk * 16 + (threadIdx.x & 15);
      pos[mb] = (blk * 16 < h && q < app_m) ? q : -1;
    }
    tile = block;
  } else {
    const int bid = block - napp;
    const int rot = (p.xcd_rot + napp) & 7;                        // workgroup bid runs on XCD (bid + rot) % 8
    tile = (p.tile_order && p.lpt) ? ntiles - 1 - bid : (rot ? xcd_tile_rot(bid, ntiles, rot) : xcd_tile(bid, ntiles));
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int t = tile * TM + ((threadIdx.x >> 6) * MB + mb) * 16 + (threadIdx.x & 15);
      pos[mb] = t < p.n_dst ? t : -1;
    }
  }
  const int mcap = layout_mcap(p.n_dst);
  const int32_t *order_app = p.argsort;                            // (layout launches only)
  const uint32_t *maskp = app ? reinterpret_cast<const uint32_t *>(order_app + mcap) : p.mask;
  const int32_t *pairp = app ? order_app + 2 * static_cast<size_t>(mcap) : p.pair;
  const int tbl_rows = app ? mcap : p.n_dst;                       // row stride of the pair table in use
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int slot = tid & 7, r0 = tid >> 3;
  // offset packing: 16-byte reduction slot (ks, lgrp) of a step belongs to offset group (ks * 4 + lgrp) / GL
  constexpr int GL = 4 / PP;                            // lane groups (16-byte reduction slots) per packed offset
  const int ga = lgrp / GL;                             // this lane's group inside a piece
  // Output-channel permutation: MFMA row (g = i >> 2, e = i & 3) of channel block nb carries
  // channel g * CPL + nb * 4 + e, so a lane ends up with CPL CONSECUTIVE channels of its voxel
  // row and stores them straight from registers (no LDS transpose in the epilogue).  The
  // weight stage in LDS is [channel][64 reduction elements], 16-byte slots XOR-swizzled with
  // (bit 1 of the channel, g): the 16 lanes of every ds_read_b128 group hit 16 distinct slots.
  auto swzB = [](int row, int sl) __attribute__((always_inline)) {
    const int x = ((row >> 1) & 1) | (((row / CPL) & 3) << 1);
    return row * kRowBytes + ((sl ^ x) << 4);
  };
  const uint32_t rowB = static_cast<uint32_t>(p.CIN) * ES;
  const int nchunk = (static_cast<int>(rowB) + kRowBytes - 1) / kRowBytes;
  const bool cfull = (rowB & (kRowBytes - 1)) == 0;

  const uint32_t a_bytes = static_cast<uint32_t>(p.n_src) * rowB;
  const uint32_t w_bytes = static_cast<uint32_t>(p.COUT) * p.kv * rowB;
  const uint32_t pair_bytes = static_cast<uint32_t>(tbl_rows) * 4u;

  // Rows of this lane.  Their numbers come from a list (the appendix' row list; an explicit mask argsort) or are the
  // positions themselves.  ONE load instruction for every kind of workgroup, through a zero-sized resource where there
  // is no list (nothing is fetched, the words come back at once): a load inside a branch makes the compiler's wait
  // counts inexact at the join, and the main tiles then waited for their mask words BEFORE requesting the centre
  // step's rows and weights (two dependent trips at the head of every tile of the launch instead of one).
  const int32_t *olist = (app || !p.cls) ? p.argsort : nullptr;
  const bool by_row = olist && !app && !p.tile_order;               // listed rows, tables in row order
  const __amdgpu_buffer_rsrc_t rO = make_rsrc(olist, olist ? pair_bytes : 0u);
  int glist[MB];
  uint32_t goff[MB];      // byte offset of the row's entry inside one pair-table row
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    goff[mb] = pos[mb] < 0 ? kOob : static_cast<uint32_t>(pos[mb]) * 4u;
    glist[mb] = static_cast<int>(__builtin_amdgcn_raw_buffer_load_b32(rO, goff[mb], 0, SPX_AUX_TABLE));
  }
  if (by_row) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) goff[mb] = pos[mb] < 0 ? kOob : static_cast<uint32_t>(glist[mb]) * 4u;
    asm volatile("" ::: "memory");                                  // (stays a branch: a select would wait for the list)
  }
  // the mask words head the longest dependency chain of the tile (mask -> pair words -> rows): requested before the
  // centre step's 24 KB of loads, not queued behind them
  const __amdgpu_buffer_rsrc_t rM = make_rsrc(maskp, maskp ? pair_bytes * p.mask_words : 0u);
  uint32_t mraw[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
    mraw[mb] = __builtin_amdgcn_raw_buffer_load_b32(rM, goff[mb] == kOob ? kOob : goff[mb] * p.mask_words, 0, SPX_AUX_TABLE);
  __builtin_amdgcn_sched_barrier(0);      // (the list words are waited for AFTER the mask words are requested)
  int grow[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) grow[mb] = pos[mb] < 0 ? -1 : (olist ? glist[mb] : pos[mb]);

  // per-thread constant offsets.  *_tail is the out-of-range bit to OR in for the last
  // reduction chunk when CIN is not a multiple of 64 (reduction elements >= CIN must read as
  // zero on BOTH operands).  Bitwise on purpose: a ?: between two arrays becomes a pointer
  // select that pins them (and the parameter block) in scratch.
  const int ctail = static_cast<int>(rowB) - (nchunk - 1) * kRowBytes;   // bytes in the last chunk
  constexpr int AK = NKS < 2 ? 2 : NKS;                 // (register arrays stay at >= 2 elements)
  uint32_t aoff[AK], aoff_tail[AK];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int c = T == 1 ? ks * 64 + lgrp * 16         // byte inside the 128-byte piece
                         : (lgrp % GL) * 16;            // ... inside the (<= 64-byte) row of this lane's offset
    aoff[ks] = static_cast<uint32_t>(c);
    aoff_tail[ks] = c < ctail ? 0u : kOob;
  }
  uint32_t boff[BA], boff_tail[BA];
  int bgrp[BA];           // offset packing: which of the step's offsets this thread's weight vector j belongs to
  if constexpr (!BT) {
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
      const int n = r0 + 32 * j;
      // PK: 16-byte slot `slot` of the stage row holds slot % GL of the row of offset slot / GL (slots 4 .. 7: nothing)
      const int sl = T == 1 ? slot : slot % GL;
      const uint32_t o = static_cast<uint32_t>(n) * static_cast<uint32_t>(p.strideN) * ES + sl * 16u;
      bgrp[j] = T == 1 ? 0 : slot / GL;
      boff[j] = (n < COUT && (T == 1 || XK || slot < 4)) ? o : kOob;
      boff_tail[j] = sl * 16 < ctail ? 0u : kOob;
    }
  } else {
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
      // reduction row inside the chunk, first of the 16 / ES channels of this vector
      const int d = F32 ? r0 : 2 * r0 + (j & 1);
      const int n = F32 ? j * 32 + slot * 4 : (j >> 1) * 64 + slot * 8;
      // PK: reduction position d of the stage = position d % (8 GL) of the row of offset d / (8 GL) (d >= 32: nothing)
      const int dl = T == 1 ? d : d % (8 * GL);
      const uint32_t o = (static_cast<uint32_t>(dl) * static_cast<uint32_t>(p.strideD) + n) * ES;
      bgrp[j] = T == 1 ? 0 : (d / (8 * GL)) & 7;
      boff[j] = (n < COUT && (T == 1 || XK || d < 32)) ? o : kOob;
      boff_tail[j] = dl * ES < ctail ? 0u : kOob;
    }
  }

  int idxr[2][MB][AK];    // pair words of a step ([.][.][1]: the second piece's offset, XK only)
  uint32_t identr[2] = {0u, 0u};   // wave-uniform: idxr[S] stands for the identity offset
  u32x4 areg[2][MB][AK];
  // WD = 2: two weight register sets -- a slice is requested THREE steps before its MFMAs (two before it
  // is written to the LDS stage) instead of two (one): on dense scenes a step was as long as the L2
  // round trip of its successor's weights.  16-bit operands up to 64 output channels (8 more
  // registers keep 4 waves per SIMD there); the wider and the int8 / fp32 variants keep one set.
  constexpr int WD = (!I8 && !F32 && COUT <= 64) ? SPX_WD : 1;
  u32x4 breg[2][BA];

  // Straight-line on purpose (no branch around a load): the compiler's s_waitcnt counts stay
  // exact only when every path issues the same loads.  A step that does not exist (k < 0)
  // reads through a zero-sized resource: every lane is out of range, nothing is fetched.
  // offset packing: the offset of step `it` that this lane's group gathers for (-1: none)
  auto lane_k = [&](const StepIt &it, int ks) __attribute__((always_inline)) {
    int km = it.k;
    if constexpr (T > 1) {
      const int g = (XK ? ks * PP : 0) + ga;
      const int kg = step_k(it, g < 1 ? 1 : g);       // (a per-lane shift of the packed word)
      km = g == 0 ? km : kg;
    }
    return km;
  };
  auto load_idx = [&](const StepIt &it, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    // the identity select happens where the words are consumed (load_a): selecting here would
    // make the loop-carried value depend on the load and park the wave on it at the loop end
    identr[S] = it.k == p.identity_k ? 0xffffffffu : 0u;
    if constexpr (T == 1) {
      const int k = it.k < 0 ? 0 : it.k;
      const __amdgpu_buffer_rsrc_t rP = make_rsrc(pairp + static_cast<size_t>(k) * tbl_rows,
                                                  (pairp && it.k >= 0) ? pair_bytes : 0u);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        idxr[S][mb][0] = SPX_ABL(p, 5) ? grow[mb] : static_cast<int>(__builtin_amdgcn_raw_buffer_load_b32(rP, goff[mb], 0, SPX_AUX_TABLE));
    } else {
      // every lane reads the pair word of ITS offset of the step (XK: of its two offsets): one resource over the whole
      // table, the table row in the lane's offset (an absent offset: out of range, the word comes back as 0 and is never
      // used -- load_a drops the lane's rows the same way)
      const __amdgpu_buffer_rsrc_t rP = make_rsrc(pairp, (pairp && it.k >= 0) ? pair_bytes * static_cast<uint32_t>(p.kv - p.kbase > 32 ? 32 : p.kv - p.kbase) : 0u);
#pragma unroll
      for (int ks = 0; ks < (XK ? 2 : 1); ++ks) {
        const int km = lane_k(it, ks);
        const uint32_t kb = km < 0 ? kOob : static_cast<uint32_t>(km) * pair_bytes;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          idxr[S][mb][ks] = static_cast<int>(__builtin_amdgcn_raw_buffer_load_b32(rP, min(kb + goff[mb], kOob) | ((kb | goff[mb]) & kOob), 0, SPX_AUX_TABLE));
      }
    }
  };
  auto load_a = [&](const StepIt &it, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const uint32_t tail = (!cfull && it.chunk == nchunk - 1) ? 0xffffffffu : 0u;
    const uint32_t so = static_cast<uint32_t>(it.chunk) * kRowBytes;
    const __amdgpu_buffer_rsrc_t r = make_rsrc(p.A, it.k >= 0 ? a_bytes : 0u);
    // offset packing: a lane whose group has no offset in this step contributes nothing (the identity step, an odd tail)
    uint32_t dead[AK];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) dead[ks] = (T > 1 && lane_k(it, ks) < 0) ? kOob : 0u;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const uint32_t idx = (static_cast<uint32_t>(grow[mb]) & identr[S]) |
                             (static_cast<uint32_t>(idxr[S][mb][XK ? ks : 0]) & ~identr[S]);
        const uint32_t rbase = (idx * rowB) | dead[ks];                    // -1 -> >= kOob
        const uint32_t lo = aoff[ks] | (aoff_tail[ks] & tail);
        const uint32_t vo = min(rbase + lo, kOob) | (lo & kOob);
        if (SPX_ABL(p, 4)) areg[S][mb][ks] = u32x4{vo, vo, vo, vo};
        else areg[S][mb][ks] = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
      }
    }
  };
  auto load_b = [&](const StepIt &it, auto WSET) __attribute__((always_inline)) {
    constexpr int WS = decltype(WSET)::value;
    const uint32_t tail = (!cfull && it.chunk == nchunk - 1) ? 0xffffffffu : 0u;
    const __amdgpu_buffer_rsrc_t r = make_rsrc(p.B, it.k >= 0 ? w_bytes : 0u);
    if constexpr (T == 1) {
      const int k = (it.k < 0 ? 0 : it.k) + p.kbase;
      const int kb = p.b_reverse ? p.kv - 1 - k : k;
      uint32_t so = static_cast<uint32_t>(kb) * static_cast<uint32_t>(p.strideK) * ES;
      if constexpr (!BT) so += static_cast<uint32_t>(it.chunk) * kRowBytes;
      else so += static_cast<uint32_t>(it.chunk) * (kRowBytes / ES) * static_cast<uint32_t>(p.strideD) * ES;
#pragma unroll
      for (int j = 0; j < BROWS; ++j)
        breg[WS][j] = __builtin_amdgcn_raw_buffer_load_b128(r, boff[j] | (boff_tail[j] & tail), so, 0);
    } else {
      // the weight slice of the offset this vector's stage position belongs to (rows are one chunk: it.chunk == 0);
      // no offset there: out of range -> zeros in the stage
#pragma unroll
      for (int j = 0; j < BROWS; ++j) {
        const int kg = step_k(it, bgrp[j] < 1 ? 1 : bgrp[j]);
        const int km = bgrp[j] == 0 ? it.k : kg;
        const int k = (km < 0 ? 0 : km) + p.kbase;
        const int kb = p.b_reverse ? p.kv - 1 - k : k;
        const uint32_t so = static_cast<uint32_t>(kb) * static_cast<uint32_t>(p.strideK) * ES;
        const uint32_t vo = boff[j] | (boff_tail[j] & tail) | (km < 0 ? kOob : 0u);
        breg[WS][j] = __builtin_amdgcn_raw_buffer_load_b128(r, min(vo + so, kOob) | (vo & kOob), 0, 0);
      }
    }
  };
  auto store_b = [&](char *ldsB, auto WSET) __attribute__((always_inline)) {
    constexpr int WS = decltype(WSET)::value;
    if constexpr (!BT) {
#pragma unroll
      for (int j = 0; j < BROWS; ++j) {
        const int n = r0 + 32 * j;
        if (COUT >= 32 * (j + 1) || n < COUT)    // compile-time true except for COUT == 16
          *reinterpret_cast<u32x4 *>(ldsB + swzB(n, slot)) = breg[WS][j];
      }
    } else if constexpr (F32) {
      // transpose: element (reduction row r0, channel n) lands in row n, byte column 4 * r0
#pragma unroll
      for (int j = 0; j < BROWS; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = j * 32 + slot * 4 + e;
          if (COUT >= 32 * (j + 1) || n < COUT)
            *reinterpret_cast<uint32_t *>(ldsB + swzB(n, r0 >> 2) + (r0 & 3) * 4) = breg[WS][j][e];
        }
      }
    } else {
      // transpose: the (d even, d odd) halves of channel n land in row n, reduction column 2*r0
#pragma unroll
      for (int jj = 0; jj < BROWS / 2; ++jj) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int n = jj * 64 + slot * 8 + e;
          const uint32_t ev = breg[WS][2 * jj][e >> 1], od = breg[WS][2 * jj + 1][e >> 1];
          const uint32_t v = (e & 1) ? __builtin_amdgcn_perm(od, ev, 0x07060302u)
                                     : __builtin_amdgcn_perm(od, ev, 0x05040100u);
          if (COUT >= 64 * (jj + 1) || n < COUT)
            *reinterpret_cast<uint32_t *>(ldsB + swzB(n, r0 >> 2) + (r0 & 3) * 4) = v;
        }
      }
    }
  };

  // ---- prologue ---------------------------------------------------------------------
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  const bool spec = p.identity_k >= 0;    // SubM: the identity offset exists for every row
  StepIt it0;
  it0.k = p.identity_k;
  it0.chunk = 0;
  it0.rest = 0;
  it0.kx = ~0ull;
  __builtin_amdgcn_sched_barrier(0);
  // identity step: start its loads before the mask words arrive.  Unconditional (a regular
  // conv has it0.k == -1 here and reads zero-sized resources) so that the wait for the mask
  // words below stays a counted one.
  load_b(it0, Set0{});
  identr[0] = 0xffffffffu;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) idxr[0][mb][0] = idxr[0][mb][1] = 0;
  load_a(it0, Set0{});
  __builtin_amdgcn_sched_barrier(0);
  SPX_STAMP(1);   // identity-step loads issued
  uint32_t wm = 0;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) wm |= mraw[mb];         // rows past the end read 0
  if (!maskp) wm = 0xffffffffu;
  wm |= __shfl_xor(wm, 1, 64);
  wm |= __shfl_xor(wm, 2, 64);
  wm |= __shfl_xor(wm, 4, 64);
  wm |= __shfl_xor(wm, 8, 64);
  const uint32_t wavemask =
      __builtin_amdgcn_readfirstlane(wm) | (spec ? (1u << p.identity_k) : 0u);
  if (lane == 0) lds_mask[wave] = wm;
  // SubM: the identity step's weights go into stage 0 BEFORE the barrier that publishes the
  // wave masks, so that one barrier serves both and the identity MFMAs can start as soon as
  // their rows have arrived (they do not depend on the mask exchange at all)
  if (spec) store_b(smem, Set0{});
  if constexpr (I8) {
    const float *bias_f = static_cast<const float *>(p.bias);
    for (int c = tid; c < 2 * COUT; c += kThreads)
      lds_sb[c] = c < COUT ? (p.scale ? p.scale[c] : 1.f) : (bias_f ? bias_f[c - COUT] : 0.f);
  }
  __syncthreads();
  SPX_STAMP(2);   // mask words arrived, tile mask exchanged
  // rows layout, main tile: a zero mask word marks a row that lives in the appendix -- its (centre-step) result is not
  // this tile's to store
  if (p.cls && !app) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      if (mraw[mb] == 0u) grow[mb] = -1;
  }
  uint32_t tilemask = lds_mask[0] | lds_mask[1] | lds_mask[2] | lds_mask[3];
  tilemask = __builtin_amdgcn_readfirstlane(tilemask);
  if (p.kv - p.kbase < 32) tilemask &= (1u << (p.kv - p.kbase)) - 1u;
  if (spec) {
    it0.rest = tilemask & ~(1u << p.identity_k);
  } else {
    it0 = step_begin<T>(tilemask);
    load_b(it0, Set0{});
    load_idx(it0, Set0{});
    load_a(it0, Set0{});
    store_b(smem, Set0{});
    __syncthreads();          // regular conv: the first step's weights could not be staged earlier
  }
  StepIt it1 = step_next<T>(it0, nchunk);
  StepIt it2 = step_next<T>(it1, nchunk);

  using acc_t = typename std::conditional<I8, i32x4, f32x4>::type;
  acc_t acc[NB][MB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = acc_t{0, 0, 0, 0};

  // MFMAs of step `it` on register set S / weight stage S.  None of this wave's rows uses
  // offset k (or the step does not exist): skipped.
  auto compute = [&](const StepIt &it, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    uint32_t stepbits = it.k >= 0 ? 1u << it.k : 0u;      // the offsets of this step (one, or up to PK)
    if constexpr (T > 1) {
#pragma unroll
      for (int g = 1; g < T; ++g) {
        const int kg = step_k(it, g);
        stepbits |= kg >= 0 ? 1u << kg : 0u;
      }
    }
    if (wavemask & stepbits) {
      const char *cur = smem + S * B_BYTES;
      const int ksteps = XK ? 2 : (min(kRowBytes, static_cast<int>(rowB) - it.chunk * kRowBytes) + 63) >> 6;  // 1 or 2
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        if (ks < ksteps) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const uint4 fa = *reinterpret_cast<const uint4 *>(
                cur + swzB((lrow >> 2) * CPL + nb * 4 + (lrow & 3), ks * 4 + lgrp));
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
              if (SPX_ABL(p, 3)) acc[nb][mb][0] += static_cast<decltype(acc[nb][mb][0] + 0)>(fa.x ^ areg[S][mb][ks][0]);
              else acc[nb][mb] = mfma_step<DT>(fa, __builtin_bit_cast(uint4, areg[S][mb][ks]), acc[nb][mb]);
            }
          }
        }
      }
    }
  };

  // ---- first step, peeled: no barrier (stage 0 is complete, stage 1 untouched), and its MFMAs
  // run before anything that waits for the pair words of the following steps.  Issue order as
  // in a loop step (weights, pair words, rows) so that the loop-header wait counts are the
  // steady-state ones.
  load_idx(it1, Set1{});
  __builtin_amdgcn_sched_barrier(0);
  using WSetA = std::integral_constant<int, (WD == 2 ? 1 : 0)>;   // set of the odd steps' weights
  load_b(it1, WSetA{});
  __builtin_amdgcn_sched_barrier(0);
  load_idx(it2, Set0{});        // idxr[0] was consumed by load_a(it0): reuse it for step 2
  __builtin_amdgcn_sched_barrier(0);
  compute(it0, Set0{});
  __builtin_amdgcn_sched_barrier(0);
  load_a(it1, Set1{});
  store_b(smem + B_BYTES, WSetA{});      // weights of step 1 -> stage 1 (published by step 1's barrier)
  {
    const StepIt it3 = step_next<T>(it2, nchunk);
    load_b(it2, Set0{});
    if constexpr (WD == 2) load_b(it3, Set1{});   // step 3's weights: in flight two steps before their LDS write
    __builtin_amdgcn_sched_barrier(0);
    load_idx(it3, Set1{});
    __builtin_amdgcn_sched_barrier(0);
    load_a(it2, Set0{});
    it0 = it1;
    it1 = it2;
    it2 = it3;
  }

  // ---- main loop: one step = one (offset, 128-byte reduction piece), two steps per trip -----
  // at step t (register set S = t & 1): areg[S] = gathered rows of step t, stage S of the
  // ring = weights of step t, breg = weights of step t+1, idxr[S] = pair words of step t+2.
  // Loads are unconditional (steps past the end read zero-sized resources).
  auto step = [&](auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    if (!SPX_ABL(p, 2)) __syncthreads();   // stage 1-S is free (read at step t-1), stage S is complete
    // WD = 1: breg[0] holds step t+1's weights, reloaded with step t+2's.  WD = 2: set (t+1) & 1 = 1 - S
    // holds step t+1's (requested at step t-2) and is reloaded with step t+3's; set S holds step t+2's.
    using WSetN = std::integral_constant<int, (WD == 2 ? 1 - S : 0)>;
    if (!SPX_ABL(p, 1) && !SPX_ABL(p, 2)) store_b(smem + (1 - S) * B_BYTES, WSetN{});
    compute(it0, SET);
    const StepIt it3 = step_next<T>(it2, nchunk);
    if (!SPX_ABL(p, 1) && !SPX_ABL(p, 2)) {
      if constexpr (WD == 2) load_b(it3, WSetN{});
      else load_b(it2, WSetN{});
    }
    __builtin_amdgcn_sched_barrier(0);   // weights first: they are the first thing step t+1 waits for
    load_idx(it3, std::integral_constant<int, 1 - S>{});
    __builtin_amdgcn_sched_barrier(0);
    load_a(it2, SET);
    it0 = it1;
    it1 = it2;
    it2 = it3;
  };
  SPX_STAMP(3);   // prologue + first step done
  while (it0.k >= 0) {
    step(Set1{});
    step(Set0{});   // may be a step past the end (no MFMAs, zero-sized loads): an early exit
                    // here would cost the exact wait counts of the whole loop
  }
  SPX_STAMP(4);   // main loop done

  // ---- epilogue: CPL consecutive channels per lane, stored straight from registers; rows past
  // the end have an out-of-range offset and are dropped by the buffer unit.
  if constexpr (!I8) {
    // bias/activation; fp32 -> 16 bit with packed converts, or fp32 as it is
    const bool plain = p.bias == nullptr && p.act == SPX_ACT_NONE;   // uniform: training path
    const __amdgpu_buffer_rsrc_t rO = make_rsrc(p.out, static_cast<uint32_t>(p.n_dst) * (COUT * ES));
    float bv[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) bv[q] = 0.f;
    if (p.bias) {
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        if constexpr (F32) bv[q] = static_cast<const float *>(p.bias)[lgrp * CPL + q];
        else bv[q] = to_float<BF16>(static_cast<const uint16_t *>(p.bias)[lgrp * CPL + q]);
      }
    }
    if (p.acc_mode) {
      // one group of a kernel volume > 32: partial sums come from / go to the fp32 scratch; bias and
      // activation apply with the last group only (acc_mode bit 1 clear)
      const __amdgpu_buffer_rsrc_t rS = make_rsrc(p.acc, static_cast<uint32_t>(p.n_dst) * (COUT * 4));
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const uint32_t so = grow[mb] < 0 ? kOob : static_cast<uint32_t>(grow[mb]) * (COUT * 4) + lgrp * (CPL * 4);
        uint32_t prev[CPL];
        if (p.acc_mode & 1) load_dwords<CPL>(prev, rS, so);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (p.acc_mode & 1) acc[nb][mb][e] += __builtin_bit_cast(float, prev[nb * 4 + e]);
            prev[nb * 4 + e] = __builtin_bit_cast(uint32_t, static_cast<float>(acc[nb][mb][e]));
          }
        if (p.acc_mode & 2) store_dwords<CPL>(prev, rS, so);
      }
      if (p.acc_mode & 2) return;
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      uint32_t d[F32 ? CPL : CPL / 2];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v0 = acc[nb][mb][2 * h], v1 = acc[nb][mb][2 * h + 1];
          if (!plain) {
            v0 = apply_act(v0 + bv[nb * 4 + 2 * h], p.act, p.act_alpha);
            v1 = apply_act(v1 + bv[nb * 4 + 2 * h + 1], p.act, p.act_alpha);
          }
          if constexpr (F32) {
            d[nb * 4 + 2 * h] = __builtin_bit_cast(uint32_t, v0);
            d[nb * 4 + 2 * h + 1] = __builtin_bit_cast(uint32_t, v1);
          } else {
            d[nb * 2 + h] = pack2<BF16>(v0, v1);
          }
        }
      }
      const uint32_t vo = grow[mb] < 0 ? kOob
                                       : static_cast<uint32_t>(grow[mb]) * (COUT * ES) + lgrp * (CPL * ES);
      // non-temporal stores: the rows are not read again by this launch, and lines left dirty in the L2 /
      // Infinity Cache are written back at the kernel boundary and push the next scene's inputs out
      // (cfg 2 step 37.4 -> 33.6 us; sc1 stores 41.9; neutral on the fixture and inside the backbone)
      if (p.dbg & 0x400) store_dwords<(F32 ? CPL : CPL / 2)>(d, rO, vo);          // (SPX_V4_DBG=1024: plain, A/B)
      else store_dwords<(F32 ? CPL : CPL / 2), 2>(d, rO, vo);
    }
    // BatchNorm statistics of the rows this workgroup stores (spx_igemm_fwd_stats; the host asks for them on plain
    // launches only: no bias, no activation) -- of the ROUNDED values, i.e. of what the normalisation layer will read
    if constexpr (!BT) {
      if (p.stats) {
        const int st_live = p.n_live ? *p.n_live : 0x7fffffff;
        __syncthreads();      // every wave is past its last read of the weight stages: the LDS is free
        wg_bn_stats<COUT, CPL, MB, kThreads / 64, (F32 ? 0 : (BF16 ? 2 : 1))>(
            acc, grow, st_live, reinterpret_cast<float *>(smem), p.stats, block, static_cast<int>(gridDim.x));
      }
    }
  } else {
    i8_epilogue<COUT, MB>(p, acc, grow, lgrp, lds_sb);
  }
  SPX_STAMP(6);   // stores issued
#ifdef SPX_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SPX_STAMP(7);   // stores retired
#endif
}

template <int COUT, int MB, int DT = 0>
constexpr size_t v4_smem_bytes() {
  // two weight stages + 4 mask words (+ int8: scale and bias of the epilogue)
  return 2 * static_cast<size_t>(COUT) * kRowBytes + 64 + (DT == 2 ? 2 * static_cast<size_t>(COUT) * 4 : 0);
}

int v4_flags(const GemmParams &p) {
  const int words = p.mask_words > 0 ? p.mask_words : 1;
  return p.b_reverse | (p.tile_order << 1) | ((words - 1) << 2) | (p.kbase << 4) | ((p.lpt ? 1 : 0) << 11) |
         ((p.cls ? 1 : 0) << 12);
}

bool v4_ok(const GemmParams &p, int es = 2, int out_es = 2) {
  const unsigned long long abytes = static_cast<unsigned long long>(p.n_src) * p.CIN * es;
  const unsigned long long pbytes = static_cast<unsigned long long>(p.n_dst) * 4ull;
  const unsigned long long wbytes = static_cast<unsigned long long>(p.COUT) * p.kv * p.CIN * es;
  const unsigned long long obytes = static_cast<unsigned long long>(p.n_dst) * p.COUT * out_es;
  return abytes < 0x7fff0000ull && pbytes < 0x7fff0000ull && wbytes < 0x7fff0000ull &&
         obytes < 0x7fff0000ull;
}

template <int COUT, int MB, int DT>
int launch_v4(const GemmParams &p, hipStream_t s);
GemmRest rest_of(const GemmParams &p);

// the PK code of a launch (igemm_v4_body) for reduction rows of a 16-bit type whose pair table fits one buffer resource
// (32 table rows): ONE piece per step packed -- 4 / 2 offsets for rows of <= 16 / 32 bytes -- in the forward / dgrad
// launches; BOTH pieces (32 / 16 / 8 for rows of <= 16 / 32 / 64 bytes) in the fused backward.  Measured on the levels of
// config 4 (profiles/r06_experiments.md section 8): these kernels are bound by vector-memory INSTRUCTIONS, and packing
// one piece halves them (the lanes behind a row's end issued dead loads); packing both pieces only halves the barriers
// -- level with one piece or slower forward (55.5 -> 55.6 us at 32 channels, 39.6 -> 43.4 at 16: three waves per SIMD
// less), 3-7 us faster in the fused backward (97.6 -> 90.7, 79.4 -> 76.4), whose occupancy the wgrad half sets anyway.
// SPX_PK = 0: no packing; 2: one piece everywhere; 3: both pieces everywhere (A/B runs, tests).
inline int v4_pack(const GemmParams &p, int DT, int es, bool fused_bwd = false) {
  const int mode = option_int("SPX_PK", 1);
  if ((DT != 0 && DT != 1) || !p.pair || mode == 0) return 1;
  if (static_cast<unsigned long long>(p.n_dst) * 4ull * 32ull >= 0x7fff0000ull) return 1;
  const int rb = p.CIN * es;
  const bool both = mode == 3 || (mode == 1 && fused_bwd);
  if (both) return rb <= 16 ? 32 : (rb <= 32 ? 16 : (rb <= 64 ? 8 : 1));
  return rb <= 16 ? 4 : (rb <= 32 ? 2 : 1);
}

template <int COUT, int MB, int DT>
int launch_v4(const GemmParams &p, hipStream_t s) {
  const int ntiles = div_up(p.n_dst, 64 * MB);
  // rows layout: appendix workgroups lead the grid -- as many as the class rule allows rows (n / 4), or as many as the
  // host says there are (app_rows, SPX_SPARSE_HINT)
  const int napp = p.cls ? (p.app_rows > 0 ? div_up(p.app_rows, 64 * MB) : layout_app_tiles(p.n_dst, 64 * MB)) : 0;
  GemmParams q = p;
  // more tiles than workgroups the chip holds at once (4 per CU up to 64 output channels, fewer beyond)
  q.lpt = p.tile_order && ntiles > ((DT == 2 || COUT > 64) ? 512 : 1024);
  GemmRest r = rest_of(p);
  r.napp = p.cls ? napp : -1;
  constexpr int es = DT == 2 ? 1 : (DT == 3 ? 4 : 2);
  const bool half = p.CIN * es <= 64;        // narrow rows: only the first 64 bytes of a piece exist
  const int pk = v4_pack(p, DT, es);         // ... <= 32 / 16 bytes: 2 / 4 offsets per step (igemm_v4_body, PK)
  count_launch(kFamV4);
#define SPX_LAUNCH_V4(BTV, NKSV, PKV)                                                                \
  hipLaunchKernelGGL((igemm_v4_kernel<COUT, MB, DT, BTV, NKSV, PKV>), dim3(napp + ntiles), dim3(kThreads),   \
                     (v4_smem_bytes<COUT, MB, DT>()), s, p.A, p.B, p.mask, p.argsort, p.pair, p.n_dst,  \
                     p.n_src, p.CIN, p.kv, p.identity_k, v4_flags(q), r)
  if (p.grid_out) *p.grid_out = (p.stats && DT != 2 && p.strideD == 1 && !p.acc_mode) ? napp + ntiles : 0;
  if (DT == 2 || p.strideD == 1) {
    if constexpr (DT == 0 || DT == 1) {
      if (pk == 32) SPX_LAUNCH_V4(false, 2, 32);
      else if (pk == 16) SPX_LAUNCH_V4(false, 2, 16);
      else if (pk == 8) SPX_LAUNCH_V4(false, 2, 8);
      else if (pk == 4) SPX_LAUNCH_V4(false, 1, 4);
      else if (pk == 2) SPX_LAUNCH_V4(false, 1, 2);
      else if (half) SPX_LAUNCH_V4(false, 1, 1);
      else SPX_LAUNCH_V4(false, 2, 1);
    } else {
      if (half) SPX_LAUNCH_V4(false, 1, 1);
      else SPX_LAUNCH_V4(false, 2, 1);
    }
  } else if constexpr (DT != 2) {
    if constexpr (DT == 0 || DT == 1) {
      if (pk == 32) SPX_LAUNCH_V4(true, 2, 32);
      else if (pk == 16) SPX_LAUNCH_V4(true, 2, 16);
      else if (pk == 8) SPX_LAUNCH_V4(true, 2, 8);
      else if (pk == 4) SPX_LAUNCH_V4(true, 1, 4);
      else if (pk == 2) SPX_LAUNCH_V4(true, 1, 2);
      else if (half) SPX_LAUNCH_V4(true, 1, 1);
      else SPX_LAUNCH_V4(true, 2, 1);
    } else {
      if (half) SPX_LAUNCH_V4(true, 1, 1);
      else SPX_LAUNCH_V4(true, 2, 1);
    }
  }
#undef SPX_LAUNCH_V4
  SPX_LAUNCH_CHECK();
  return 0;
}

inline GemmRest rest_of(const GemmParams &p) {
  GemmRest r{};
  r.out = p.out;
  r.bias = p.bias;
  r.strideK = p.strideK;
  r.strideN = p.strideN;
  r.strideD = p.strideD;
  r.COUT = p.COUT;
  r.act = p.act;
  r.act_alpha = p.act_alpha;
  r.scale = p.scale;
  r.add = p.add;
  r.add_scale = p.add_scale;
  r.out_dtype = p.out_dtype;
  constexpr int dbg = 0;
  r.dbg = dbg | p.dbg;
  r.acc = p.acc;
  r.acc_mode = p.acc_mode;
  r.napp = -1;
  r.stats = p.stats;
  r.n_live = p.n_live;
  return r;
}

// --------------------------------------------------------------------------------------------------------------
// int8, SPARSE class (rows layout, class word 1): appendix tiles and STREAMING main tiles in one launch.
// A main tile of a sparse rulebook is one step -- the centre pair: rows in their own order, no pair word -- and as a
// workgroup of igemm_v4_kernel it pays for that step with a 16 KB weight slice through global -> registers -> LDS, a
// barrier and a dispatch slot (3 125 tiles at BASELINE config 5: 51 MB of weight traffic for 51 MB of rows).  Here the
// main tiles are a LOOP: a workgroup stages the centre slice once and streams 64-row tiles behind it (next tile's rows
// and mask words in flight during the MFMAs and the quantised epilogue of the current one); the appendix tiles lead
// the grid and run igemm_v4_body unchanged.  Same integer arithmetic, same epilogue function: bit-identical.  The class
// word is read on the DEVICE as well: a dense rulebook behind a stale hint takes igemm_v4_body tile by tile.
template <int COUT>
__device__ __forceinline__ void i8_centre_stream(const GemmParams &p, int first, int stride) {
  constexpr int NB = COUT / 16, CPL = NB * 4, B_BYTES = COUT * kRowBytes, TM = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *lds_sb = reinterpret_cast<float *>(smem + 2 * B_BYTES + 64);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int slot = tid & 7, r0 = tid >> 3;
  auto swzB = [](int row, int sl) __attribute__((always_inline)) {
    const int x = ((row >> 1) & 1) | (((row / CPL) & 3) << 1);
    return row * kRowBytes + ((sl ^ x) << 4);
  };
  const uint32_t rowB = static_cast<uint32_t>(p.CIN);               // bytes of a row (int8)
  {
    // the centre slice -> stage 0 (pieces past the row's end: out of range -> zeros, as in igemm_v4_body)
    const uint32_t w_bytes = static_cast<uint32_t>(p.COUT) * p.kv * rowB;
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(p.B, w_bytes);
    const uint32_t so = static_cast<uint32_t>(p.identity_k) * static_cast<uint32_t>(p.strideK);
#pragma unroll
    for (int j = 0; j < (COUT + 31) / 32; ++j) {
      const int n = r0 + 32 * j;
      if (n < COUT) {
        const uint32_t o = static_cast<uint32_t>(n) * static_cast<uint32_t>(p.strideN) + slot * 16u;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rW, slot * 16u < rowB ? o : kOob, so, 0);
        *reinterpret_cast<u32x4 *>(smem + swzB(n, slot)) = v;
      }
    }
    const float *bias_f = static_cast<const float *>(p.bias);
    for (int c = tid; c < 2 * COUT; c += kThreads)
      lds_sb[c] = c < COUT ? (p.scale ? p.scale[c] : 1.f) : (bias_f ? bias_f[c - COUT] : 0.f);
  }
  __syncthreads();
  const int ntiles = (p.n_dst + TM - 1) / TM;
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A, static_cast<uint32_t>(p.n_src) * rowB);
  const __amdgpu_buffer_rsrc_t rM = make_rsrc(p.mask, static_cast<uint32_t>(p.n_dst) * 4u);
  const int nks = rowB > 64u ? 2 : 1;
  uint32_t aoff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const uint32_t c = static_cast<uint32_t>(ks * 64 + lgrp * 16);
    aoff[ks] = c < rowB ? c : kOob;
  }
  auto load_tile = [&](int t, u32x4 (&a)[2], uint32_t &m, int &row) __attribute__((always_inline)) {
    row = t * TM + wave * 16 + lrow;
    const bool ok = t < ntiles && row < p.n_dst;
    m = __builtin_amdgcn_raw_buffer_load_b32(rM, ok ? static_cast<uint32_t>(row) * 4u : kOob, 0, SPX_AUX_TABLE);
    const uint32_t base = ok ? static_cast<uint32_t>(row) * rowB : kOob;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      a[ks] = __builtin_amdgcn_raw_buffer_load_b128(rA, min(base + aoff[ks], kOob) | (aoff[ks] & kOob), 0, 0);
  };
  u32x4 a_cur[2], a_nxt[2];
  uint32_t m_cur, m_nxt;
  int row_cur, row_nxt;
  int t = first;
  load_tile(t, a_cur, m_cur, row_cur);
  while (t < ntiles) {
    const int tn = t + stride;
    load_tile(tn, a_nxt, m_nxt, row_nxt);                 // (past the end: nothing is fetched)
    i32x4 acc[NB][1];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb][0] = i32x4{0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks < nks) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const uint4 fa = *reinterpret_cast<const uint4 *>(
              smem + swzB((lrow >> 2) * CPL + nb * 4 + (lrow & 3), ks * 4 + lgrp));
          acc[nb][0] = mfma_step<2>(fa, __builtin_bit_cast(uint4, a_cur[ks]), acc[nb][0]);
        }
      }
    }
    // a zero mask word: the row lives in the appendix (its result is not this tile's to store); rows past the end
    const int grow[1] = {(row_cur < p.n_dst && m_cur != 0u) ? row_cur : -1};
    i8_epilogue<COUT, 1>(p, acc, grow, lgrp, lds_sb);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_cur[ks] = a_nxt[ks];
    m_cur = m_nxt;
    row_cur = row_nxt;
    t = tn;
  }
}

template <int COUT>
__global__ void __launch_bounds__(kThreads, 4)
igemm_i8_sparse_kernel(const void *argA, const void *argB, const uint32_t *arg_mask,
                       const int32_t *arg_argsort, const int32_t *arg_pair, int n_dst, int n_src, int CIN,
                       int kv, int identity_k, int b_reverse, GemmRest rest) {
  GemmParams p;
  unpack_gemm_args(p, argA, argB, arg_mask, arg_argsort, arg_pair, n_dst, n_src, CIN, kv, identity_k, b_reverse, rest);
  const int napp = p.app_rows;                             // appendix workgroups at the head of the grid (the launcher's)
  const int block = static_cast<int>(blockIdx.x), nmain = static_cast<int>(gridDim.x) - napp;
  if (block < napp) {
    igemm_v4_body<COUT, 1, 2, false, 2>(p, block);
    return;
  }
  typedef const int32_t __attribute__((address_space(4))) *cptr_t;
  if (*(cptr_t)(p.cls) == 1) {
    i8_centre_stream<COUT>(p, block - napp, nmain);
  } else {
    const int ntiles = (n_dst + 63) / 64;                  // a dense rulebook (the hint was wrong): the general body
    for (int b = block; b < napp + ntiles; b += nmain) {
      igemm_v4_body<COUT, 1, 2, false, 2>(p, b);
      __syncthreads();                                     // (the stages are reused by the next tile)
    }
  }
}

template <int COUT>
int launch_i8_sparse(const GemmParams &p, hipStream_t s) {
  const int ntiles = div_up(p.n_dst, 64);
  const int napp = p.app_rows > 0 ? div_up(p.app_rows, 64) : layout_app_tiles(p.n_dst, 64);
  const int resident = (COUT == 128 ? 4 : 5) * 256;        // workgroups the chip holds at once (113 / 83 registers per lane)
  const int want = option_int("SPX_I8_NMAIN", 0);           // (A/B runs)
  int nmain = want > 0 ? want : (resident - napp > 256 ? resident - napp : 256);
  if (nmain > ntiles) nmain = ntiles;
  GemmParams q = p;
  q.lpt = false;
  GemmRest r = rest_of(p);
  r.napp = napp;
  count_launch(kFamI8Stream);
  hipLaunchKernelGGL((igemm_i8_sparse_kernel<COUT>), dim3(napp + nmain), dim3(kThreads), (v4_smem_bytes<COUT, 1, 2>()), s,
                     p.A, p.B, p.mask, p.argsort, p.pair, p.n_dst, p.n_src, p.CIN, p.kv, p.identity_k, v4_flags(q), r);
  SPX_LAUNCH_CHECK();
  return 0;
}

// shapes the streaming form is instantiated for (igemm_i8.hip)
inline bool i8_sparse_ok(const GemmParams &p) {
  return p.cls && p.identity_k >= 0 && p.identity_k < p.kv && (p.COUT == 64 || p.COUT == 128) && p.CIN <= 128 &&
         p.kbase == 0 && p.mask_words <= 1 && !p.acc_mode && p.strideD == 1;
}

template <bool BF16>
int dispatch_gather_gemm(const GemmParams &p, hipStream_t s) {
  constexpr int mb_forced = 0;              // (tile height: rule below)
  // dense neighbourhoods (the caller's hint; SPX_WS = 1 / 0 forces / forbids): the weight-stationary kernel --
  // bit-identical results, so the choice never shows in an output
  const int wsv = option_int("SPX_WS", -1);
  if ((wsv > 0 || (wsv < 0 && p.dense_hint)) && ws_ok(p, BF16 ? SPX_BF16 : SPX_F16)) {
    GemmParams q = p;
    drop_rows_layout(q);
    return launch_gather_gemm_ws(q, BF16 ? SPX_BF16 : SPX_F16, s);
  }
  if (v4_ok(p)) {
    // 64-row tiles while the grid would otherwise leave CUs idle, 128-row tiles beyond -- except
    // for 128 output channels, whose 128-row variant holds 64 accumulator registers per lane and
    // drops to two waves per SIMD (measured at C = K = 128: 28 vs 37 us at 100 k uniform voxels,
    // 56 vs 75 us at 200 k, equal on dense scenes)
    // (threshold: at 50 k rows 128-row tiles already win at every width, sparse and dense)
    const int mb = mb_forced ? mb_forced : ((p.n_dst <= 32 * 1024 || p.COUT == 128) ? 1 : 2);
    switch (p.COUT) {
      case 16: return mb == 1 ? launch_v4<16, 1, BF16 ? 1 : 0>(p, s) : launch_v4<16, 2, BF16 ? 1 : 0>(p, s);
      case 32: return mb == 1 ? launch_v4<32, 1, BF16 ? 1 : 0>(p, s) : launch_v4<32, 2, BF16 ? 1 : 0>(p, s);
      case 64: return mb == 1 ? launch_v4<64, 1, BF16 ? 1 : 0>(p, s) : launch_v4<64, 2, BF16 ? 1 : 0>(p, s);
      case 128: return mb == 1 ? launch_v4<128, 1, BF16 ? 1 : 0>(p, s) : launch_v4<128, 2, BF16 ? 1 : 0>(p, s);
      case 256: return launch_v4<256, 1, BF16 ? 1 : 0>(p, s);
    }
  }
  if (p.cls) {                     // a rows layout is a hint: the first-generation kernel reads the tables by row
    GemmParams q = p;
    drop_rows_layout(q);
    return launch_gather_gemm_gen1(q, BF16, s);
  }
  if (p.tile_order) {
    set_error("tables in tile order need the direct-fragment kernel (tensor beyond 32-bit offsets?)");
    return -1;
  }
  return launch_gather_gemm_gen1(p, BF16, s);      // igemm_gen1.hip: tensors beyond 32-bit buffer offsets
}

}  // namespace
}  // namespace spx
