// Sparse convolution compute kernels for gfx950 (MI355X).
//
//  * gather_gemm_mfma: output-stationary implicit GEMM used for forward and
//    dgrad.  One 256-thread workgroup owns 128 output rows (4 waves x 32 rows)
//    -> every output row is written exactly once, no atomics.  For each kernel
//    offset k present in the tile (OR of the rows' rulebook masks) it gathers
//    the 128 source rows as full 128-byte lines into an XOR-swizzled LDS tile,
//    stages the [Cout x 64] weight slice next to it and issues
//    v_mfma_f32_16x16x32_{f16,bf16}.  Global loads of step t+1 are issued
//    before the MFMAs of step t (register-staged software pipeline).
//  * wgrad_mfma: per (offset, chunk-of-pairs) workgroups contract
//    dout^T (x) feat over the Native pair lists into 64x64 fp32 partials
//    (transposing, XOR-swizzled LDS stores), followed by a deterministic
//    second-stage reduction (no atomics, no split-K races).
//  * generic fp32-accumulate kernels for fp32 tensors and odd channel counts.
//
// Roofline note: at C=K=64 these kernels are HBM/L2-bandwidth bound (SURVEY.md
// section 8d): ~110 MB of compulsory traffic per fwd+bwd at 100k voxels versus
// 2.5 GFLOP, so the design spends its effort on coalesced 128-byte row
// gathers, mask-predicated rulebook reads and single-pass outputs, not on MFMA
// utilisation.
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




template <int COUT, int MB, int DT, bool BT, int NKS = 2>
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
template <int COUT, int MB, int DT, bool BT, int NKS = 2>
__global__ void __launch_bounds__(kThreads)
igemm_v4_kernel(const void *argA, const void *argB, const uint32_t *arg_mask,
                const int32_t *arg_argsort, const int32_t *arg_pair, int n_dst, int n_src, int CIN,
                int kv, int identity_k, int b_reverse, GemmRest rest) {
  GemmParams p;
  unpack_gemm_args(p, argA, argB, arg_mask, arg_argsort, arg_pair, n_dst, n_src, CIN, kv, identity_k,
                   b_reverse, rest);
  igemm_v4_body<COUT, MB, DT, BT, NKS>(p, blockIdx.x);
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

template <int COUT, int MB, int DT, bool BT, int NKS>
__device__ __forceinline__ void igemm_v4_body(const GemmParams &p, int block) {
  constexpr bool BF16 = DT == 1, I8 = DT == 2, F32 = DT == 3;
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
    if (!cls) return;
    // The M rows of the appendix are dealt to the launch's napp appendix workgroups in whole 16-row blocks, h rows
    // each: the appendix tiles are the launch's critical path (a walk over every offset any of their rows has, after
    // the main tiles have long finished), and the workgroups reserved for it (n / 4 rows' worth) are there anyway --
    // config 2: 3 165 rows, 25 tiles of 128 rows walk 4.6 offsets on average and 10 at most, 99 tiles of 32 rows
    // 2.7 and 7 (forward 11.2 -> 9.5 us).  A full appendix (M = n / 4) keeps whole tiles.  The fused backward shares
    // the chip's 1024 workgroup slots with the wgrad ranges and deals to at most kAppBudget workgroups (app_budget).
    const int groups = p.app_budget > 0 ? min(napp, p.app_budget) : napp;
    const int h = min(TM, (((app_m + groups - 1) / groups) + 15) & ~15);
    if (block * h >= app_m) return;
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
    const int c = ks * 64 + lgrp * 16;                  // byte inside the 128-byte piece
    aoff[ks] = static_cast<uint32_t>(c);
    aoff_tail[ks] = c < ctail ? 0u : kOob;
  }
  uint32_t boff[BA], boff_tail[BA];
  if constexpr (!BT) {
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
      const int n = r0 + 32 * j;
      const uint32_t o = static_cast<uint32_t>(n) * static_cast<uint32_t>(p.strideN) * ES + slot * 16u;
      boff[j] = n < COUT ? o : kOob;
      boff_tail[j] = slot * 16 < ctail ? 0u : kOob;
    }
  } else {
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
      // reduction row inside the chunk, first of the 16 / ES channels of this vector
      const int d = F32 ? r0 : 2 * r0 + (j & 1);
      const int n = F32 ? j * 32 + slot * 4 : (j >> 1) * 64 + slot * 8;
      const uint32_t o = (static_cast<uint32_t>(d) * static_cast<uint32_t>(p.strideD) + n) * ES;
      boff[j] = n < COUT ? o : kOob;
      boff_tail[j] = d * ES < ctail ? 0u : kOob;
    }
  }

  int idxr[2][MB];
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
  auto load_idx = [&](const StepIt &it, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const int k = it.k < 0 ? 0 : it.k;
    const __amdgpu_buffer_rsrc_t rP = make_rsrc(pairp + static_cast<size_t>(k) * tbl_rows,
                                                (pairp && it.k >= 0) ? pair_bytes : 0u);
    // the identity select happens where the words are consumed (load_a): selecting here would
    // make the loop-carried value depend on the load and park the wave on it at the loop end
    identr[S] = it.k == p.identity_k ? 0xffffffffu : 0u;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      idxr[S][mb] = SPX_ABL(p, 5) ? grow[mb] : static_cast<int>(__builtin_amdgcn_raw_buffer_load_b32(rP, goff[mb], 0, SPX_AUX_TABLE));
  };
  auto load_a = [&](const StepIt &it, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const uint32_t tail = (!cfull && it.chunk == nchunk - 1) ? 0xffffffffu : 0u;
    const uint32_t so = static_cast<uint32_t>(it.chunk) * kRowBytes;
    const __amdgpu_buffer_rsrc_t r = make_rsrc(p.A, it.k >= 0 ? a_bytes : 0u);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const uint32_t idx = (static_cast<uint32_t>(grow[mb]) & identr[S]) |
                           (static_cast<uint32_t>(idxr[S][mb]) & ~identr[S]);
      const uint32_t rbase = idx * rowB;                                   // -1 -> >= kOob
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
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
    const int k = (it.k < 0 ? 0 : it.k) + p.kbase;
    const int kb = p.b_reverse ? p.kv - 1 - k : k;
    uint32_t so = static_cast<uint32_t>(kb) * static_cast<uint32_t>(p.strideK) * ES;
    if constexpr (!BT) so += static_cast<uint32_t>(it.chunk) * kRowBytes;
    else so += static_cast<uint32_t>(it.chunk) * (kRowBytes / ES) * static_cast<uint32_t>(p.strideD) * ES;
    const __amdgpu_buffer_rsrc_t r = make_rsrc(p.B, it.k >= 0 ? w_bytes : 0u);
#pragma unroll
    for (int j = 0; j < BROWS; ++j)
      breg[WS][j] = __builtin_amdgcn_raw_buffer_load_b128(r, boff[j] | (boff_tail[j] & tail), so, 0);
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
  __builtin_amdgcn_sched_barrier(0);
  // identity step: start its loads before the mask words arrive.  Unconditional (a regular
  // conv has it0.k == -1 here and reads zero-sized resources) so that the wait for the mask
  // words below stays a counted one.
  load_b(it0, Set0{});
  identr[0] = 0xffffffffu;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) idxr[0][mb] = 0;
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
    it0 = step_begin(tilemask);
    load_b(it0, Set0{});
    load_idx(it0, Set0{});
    load_a(it0, Set0{});
    store_b(smem, Set0{});
    __syncthreads();          // regular conv: the first step's weights could not be staged earlier
  }
  StepIt it1 = step_next(it0, nchunk);
  StepIt it2 = step_next(it1, nchunk);

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
    if (it.k >= 0 && ((wavemask >> it.k) & 1u)) {
      const char *cur = smem + S * B_BYTES;
      const int ksteps = (min(kRowBytes, static_cast<int>(rowB) - it.chunk * kRowBytes) + 63) >> 6;  // 1 or 2
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
    const StepIt it3 = step_next(it2, nchunk);
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
    const StepIt it3 = step_next(it2, nchunk);
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
  } else {
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
#define SPX_LAUNCH_V4(BTV, NKSV)                                                                     \
  hipLaunchKernelGGL((igemm_v4_kernel<COUT, MB, DT, BTV, NKSV>), dim3(napp + ntiles), dim3(kThreads),   \
                     (v4_smem_bytes<COUT, MB, DT>()), s, p.A, p.B, p.mask, p.argsort, p.pair, p.n_dst,  \
                     p.n_src, p.CIN, p.kv, p.identity_k, v4_flags(q), r)
  if (DT == 2 || p.strideD == 1) {
    if (half) SPX_LAUNCH_V4(false, 1);
    else SPX_LAUNCH_V4(false, 2);
  } else if constexpr (DT != 2) {
    if (half) SPX_LAUNCH_V4(true, 1);
    else SPX_LAUNCH_V4(true, 2);
  }
#undef SPX_LAUNCH_V4
  SPX_LAUNCH_CHECK();
  return 0;
}

// --------------------------------------------------------------------------
// generic gather-GEMM: any dtype / channel count, fp32 accumulate.
// one thread per (dst row, out channel).
// --------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float load_f(const T *p);
template <> __device__ __forceinline__ float load_f<float>(const float *p) { return *p; }
struct h16 { uint16_t v; };
struct b16 { uint16_t v; };
template <> __device__ __forceinline__ float load_f<h16>(const h16 *p) { return to_float<false>(p->v); }
template <> __device__ __forceinline__ float load_f<b16>(const b16 *p) { return to_float<true>(p->v); }
template <typename T> __device__ __forceinline__ void store_f(T *p, float f);
template <> __device__ __forceinline__ void store_f<float>(float *p, float f) { *p = f; }
template <> __device__ __forceinline__ void store_f<h16>(h16 *p, float f) { p->v = from_float<false>(f); }
template <> __device__ __forceinline__ void store_f<b16>(b16 *p, float f) { p->v = from_float<true>(f); }

template <typename T>
__global__ void __launch_bounds__(kThreads)
gather_gemm_generic_kernel(GemmParams p) {
  const long long gid = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  const long long total = static_cast<long long>(p.n_dst) * p.COUT;
  if (gid >= total) return;
  const int d = static_cast<int>(gid / p.COUT), n = static_cast<int>(gid % p.COUT);
  const T *A = static_cast<const T *>(p.A);
  const T *B = static_cast<const T *>(p.B);
  const int words = (p.kv + 31) / 32;
  float acc = 0.f;
  for (int k = 0; k < p.kv; ++k) {
    if (p.mask && !((p.mask[static_cast<size_t>(d) * words + (k >> 5)] >> (k & 31)) & 1u)) continue;
    const int idx = (k == p.identity_k) ? d : p.pair[static_cast<size_t>(k) * p.n_dst + d];
    if (idx < 0) continue;
    const int kb = p.b_reverse ? p.kv - 1 - k : k;
    const T *a = A + static_cast<size_t>(idx) * p.CIN;
    const T *b = B + static_cast<size_t>(kb) * p.strideK + static_cast<size_t>(n) * p.strideN;
    for (int c = 0; c < p.CIN; ++c) acc = fmaf(load_f(a + c), load_f(b + c * p.strideD), acc);
  }
  if (p.bias) acc += load_f(static_cast<const T *>(p.bias) + n);
  acc = apply_act(acc, p.act, p.act_alpha);
  store_f(static_cast<T *>(p.out) + static_cast<size_t>(d) * p.COUT + n, acc);
}

// --------------------------------------------------------------------------
// wgrad, 16-bit operands: per (chunk of pairs, offset k, 64x64 dW tile) block.
//   partial[kk][c] = sum_{j in chunk} dout[out_j][kk0+kk] * feat[in_j][c0+c]
// LDS tiles are stored transposed ([channel][pair]) so that the MFMA
// fragments (8 consecutive pairs of one channel) are single ds_read_b128.
// --------------------------------------------------------------------------
constexpr int kWJ = 128;   // pairs staged per iteration
constexpr int kWT = 64;    // dW tile edge

struct WgradParams {
  const void *feat;        // [n_in, C]
  const void *dout;        // [n_out, K]
  float *partial;          // [item][tiles][64*64]
  const int32_t *native;   // [2, kv, n_in]
  const int32_t *num;      // [kv] device counts
  const int32_t *plan;     // see wgrad_plan_kernel
  int n_in, n_out, C, K, kv, subm, chunk, nchunks, tiles_c, tiles_k;
};

__device__ __forceinline__ int list_count(const int32_t *num, int kv, int subm, int n_in, int k) {
  int c;
  if (!subm) c = num[k];
  else if (k == kv / 2) c = n_in;
  else c = k < kv / 2 ? num[k] : num[kv - 1 - k];  // mirror rule, ops.py:962-968
  return c < n_in ? c : n_in;                         // convops.py:1592 clamp
}

// Work plan: the (offset, chunk) items that actually exist, so that the wgrad grid
// holds no empty workgroups.  Layout (int32):
//   [0] total items   [1 .. 1+kv) first item of offset k   [1+kv .. 1+2kv) chunks of offset k
//   [1+2kv + 2 i] = offset of item i,  [2+2kv + 2 i] = first pair of item i
__global__ void __launch_bounds__(kThreads)
wgrad_plan_kernel(const int32_t *__restrict__ num, int n_in, int kv, int subm, int chunk,
                  int32_t *__restrict__ plan) {
  __shared__ int first[129], nch[128];
  const int tid = threadIdx.x;
  if (tid < kv) {
    const int c = list_count(num, kv, subm, n_in, tid);
    nch[tid] = (c + chunk - 1) / chunk;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int k = 0; k < kv; ++k) {
      first[k] = run;
      run += nch[k];
    }
    first[kv] = run;
    plan[0] = run;
  }
  __syncthreads();
  if (tid < kv) {
    plan[1 + tid] = first[tid];
    plan[1 + kv + tid] = nch[tid];
  }
  int32_t *items = plan + 1 + 2 * kv;
  for (int k = 0; k < kv; ++k)
    for (int c = tid; c < nch[k]; c += kThreads) {
      items[2 * (first[k] + c)] = k;
      items[2 * (first[k] + c) + 1] = c * chunk;
    }
}

// element column of pair j in transposed row `ch`: XOR swizzle at 8-element
// granularity (conflict-free fragment reads, <=2-way conflicts on the stores)
__device__ __forceinline__ int wswz(int ch, int j) {
  const int h = (ch & 15) ^ (((ch >> 4) & 3) << 1);
  return j ^ ((h & 15) << 3);
}

__device__ __forceinline__ uint32_t dword_of(const uint4 &v, int i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

template <bool BF16>
__global__ void __launch_bounds__(kThreads)
wgrad_mfma_kernel(WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t *ldsD = reinterpret_cast<uint16_t *>(smem);                    // [64 kk][128 j]
  uint16_t *ldsF = reinterpret_cast<uint16_t *>(smem + kWT * kWJ * 2);    // [64 c ][128 j]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntile = p.tiles_k * p.tiles_c;
  const int total = p.plan[0] * ntile;
  const int32_t *items = p.plan + 1 + 2 * p.kv;
  const uint16_t *F = static_cast<const uint16_t *>(p.feat);
  const uint16_t *D = static_cast<const uint16_t *>(p.dout);
  const int slot = tid & 7;       // 8-channel group
  const int jp0 = tid >> 3;       // pair-of-rows index 0..31 (+32)
  const int wk = wave >> 1, wc = wave & 1;  // wave quadrant: kk [32*wk,+32), c [32*wc,+32)

  for (int work = blockIdx.x; work < total; work += gridDim.x) {
    const int item = work / ntile, tile = work - item * ntile;
    const int k = items[2 * item], begin = items[2 * item + 1];
    const int kk0 = (tile / p.tiles_c) * kWT, c0 = (tile % p.tiles_c) * kWT;
    const int cnt = list_count(p.num, p.kv, p.subm, p.n_in, k);
    const int end = min(cnt, begin + p.chunk);
    const bool identity = p.subm && k == p.kv / 2;
    const int32_t *in_list = p.native + static_cast<size_t>(k) * p.n_in;
    const int32_t *out_list = p.native + static_cast<size_t>(p.kv + k) * p.n_in;
    const bool d_ok = kk0 + slot * 8 < p.K, f_ok = c0 + slot * 8 < p.C;

    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 dv[2][2], fv[2][2];
    auto load_rows = [&](int base) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = base + 2 * (jp0 + 32 * q) + h;
          uint4 d = make_uint4(0, 0, 0, 0), f = make_uint4(0, 0, 0, 0);
          if (j < end) {
            const int oi = identity ? j : out_list[j];
            const int ii = identity ? j : in_list[j];
            if (d_ok) d = *reinterpret_cast<const uint4 *>(D + static_cast<size_t>(oi) * p.K + kk0 + slot * 8);
            if (f_ok) f = *reinterpret_cast<const uint4 *>(F + static_cast<size_t>(ii) * p.C + c0 + slot * 8);
          }
          dv[q][h] = d;
          fv[q][h] = f;
        }
    };
    load_rows(begin);
    for (int base = begin; base < end; base += kWJ) {
      __syncthreads();  // previous iteration's fragment reads are done
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int j = 2 * (jp0 + 32 * q);  // even -> a dword holds pairs (j, j+1)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ch = slot * 8 + e;
          const int sh = (e & 1) * 16;
          const uint32_t dd = ((dword_of(dv[q][0], e >> 1) >> sh) & 0xffffu) |
                              (((dword_of(dv[q][1], e >> 1) >> sh) & 0xffffu) << 16);
          const uint32_t ff = ((dword_of(fv[q][0], e >> 1) >> sh) & 0xffffu) |
                              (((dword_of(fv[q][1], e >> 1) >> sh) & 0xffffu) << 16);
          const int col = wswz(ch, j);
          *reinterpret_cast<uint32_t *>(ldsD + ch * kWJ + col) = dd;
          *reinterpret_cast<uint32_t *>(ldsF + ch * kWJ + col) = ff;
        }
      }
      __syncthreads();
      if (base + kWJ < end) load_rows(base + kWJ);   // in flight during the MFMAs
#pragma unroll
      for (int ks = 0; ks < kWJ / 32; ++ks) {
        const int j8 = (ks * 4 + (lane >> 4)) * 8;
        uint4 fa[2], fb[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int ch = wk * 32 + a * 16 + (lane & 15);
          fa[a] = *reinterpret_cast<const uint4 *>(ldsD + ch * kWJ + wswz(ch, j8));
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int ch = wc * 32 + b * 16 + (lane & 15);
          fb[b] = *reinterpret_cast<const uint4 *>(ldsF + ch * kWJ + wswz(ch, j8));
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = mfma16<BF16>(fa[a], fb[b], acc[a][b]);
      }
    }
    // D[i = kk][j = c]: lane holds c = lane & 15, kk = (lane >> 4) * 4 + reg
    float *dst = p.partial + static_cast<size_t>(work) * (kWT * kWT);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = wk * 32 + a * 16 + (lane >> 4) * 4 + e;
          const int c = wc * 32 + b * 16 + (lane & 15);
          if (SPX_AUX_OUT) __builtin_nontemporal_store(acc[a][b][e], &dst[kk * kWT + c]);
          else dst[kk * kWT + c] = acc[a][b][e];
        }
    __syncthreads();  // LDS is rewritten by the next work item
  }
}

// generic wgrad partial: thread per (kk, c) of the 64x64 tile, 16 elems/thread
template <typename T>
__global__ void __launch_bounds__(kThreads)
wgrad_generic_kernel(WgradParams p) {
  const int ntile = p.tiles_k * p.tiles_c;
  const int total = p.plan[0] * ntile;
  const int32_t *items = p.plan + 1 + 2 * p.kv;
  const T *F = static_cast<const T *>(p.feat);
  const T *D = static_cast<const T *>(p.dout);
  for (int work = blockIdx.x; work < total; work += gridDim.x) {
    const int item = work / ntile, tile = work - item * ntile;
    const int k = items[2 * item], begin = items[2 * item + 1];
    const int kk0 = (tile / p.tiles_c) * kWT, c0 = (tile % p.tiles_c) * kWT;
    const int cnt = list_count(p.num, p.kv, p.subm, p.n_in, k);
    const int end = min(cnt, begin + p.chunk);
    const bool identity = p.subm && k == p.kv / 2;
    const int32_t *in_list = p.native + static_cast<size_t>(k) * p.n_in;
    const int32_t *out_list = p.native + static_cast<size_t>(p.kv + k) * p.n_in;
    float *dst = p.partial + static_cast<size_t>(work) * (kWT * kWT);
    for (int e = threadIdx.x; e < kWT * kWT; e += kThreads) {
      const int kk = kk0 + e / kWT, c = c0 + e % kWT;
      float acc = 0.f;
      if (kk < p.K && c < p.C) {
        for (int j = begin; j < end; ++j) {
          const int oi = identity ? j : out_list[j];
          const int ii = identity ? j : in_list[j];
          acc = fmaf(load_f(D + static_cast<size_t>(oi) * p.K + kk),
                     load_f(F + static_cast<size_t>(ii) * p.C + c), acc);
        }
      }
      dst[e] = acc;
    }
  }
}

// dw[kk][k][c] = sum over the items of offset k (fixed order -> deterministic).
// 512 threads = 32 consecutive elements x 16 item groups; every thread keeps 8 loads in flight.
constexpr int kRedThreads = 512;
constexpr int kRedSplit = 16;
constexpr int kRedElems = kRedThreads / kRedSplit;  // 32

template <typename T>
__global__ void __launch_bounds__(kRedThreads)
wgrad_reduce_kernel(WgradParams p, T *__restrict__ dw) {
  __shared__ float red[kRedThreads];
  const int k = blockIdx.y;
  const int first = p.plan[1 + k], nch = p.plan[1 + p.kv + k];
  const int grp = threadIdx.x / kRedElems, el = threadIdx.x % kRedElems;
  const int ntile = p.tiles_k * p.tiles_c;
  const int e_global = blockIdx.x * kRedElems + el;  // element of [tiles][64*64]
  const int tile = e_global / (kWT * kWT), e = e_global % (kWT * kWT);
  const size_t stride = static_cast<size_t>(ntile) * (kWT * kWT);
  float acc = 0.f;
  if (tile < ntile) {
    const float *src = p.partial + (static_cast<size_t>(first) * ntile + tile) * (kWT * kWT) + e;
    int ch = grp;
    for (; ch + 7 * kRedSplit < nch; ch += 8 * kRedSplit) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[static_cast<size_t>(ch + u * kRedSplit) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; ch < nch; ch += kRedSplit) acc += src[static_cast<size_t>(ch) * stride];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (grp == 0 && tile < ntile) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < kRedSplit; ++g) s += red[g * kRedElems + el];
    const int kk = (tile / p.tiles_c) * kWT + e / kWT, c = (tile % p.tiles_c) * kWT + e % kWT;
    if (kk < p.K && c < p.C) store_f(dw + (static_cast<size_t>(kk) * p.kv + k) * p.C + c, s);
  }
}

// --------------------------------------------------------------------------
// wgrad v2 ("balanced segments + transpose reads"), 16-bit operands.
//  * work split: the concatenation of all pair lists is cut into G equal ranges, one per
//    workgroup; a range that crosses a list boundary becomes several segments.  Every
//    workgroup streams the same number of rows, so all CUs finish together (per-CU HBM
//    bandwidth is ~24 GB/s: an idle CU is lost bandwidth).
//  * rows go global -> registers -> LDS exactly as they lie in memory ([pair][channel],
//    ds_write_b128, 32-byte granules XOR-swizzled); the MFMA operands, which need 8
//    consecutive PAIRS of one channel per lane, come out of LDS through the hardware
//    transpose read ds_read_b64_tr_b16 (two per fragment) -- no shuffling VALU work.
//  * two LDS stages: one __syncthreads() per 128-pair chunk; the next chunk's rows are in
//    flight during the MFMAs, the pair-list words one chunk further ahead.
//  * per-segment fp32 partials + the deterministic second stage below (no atomics).
// --------------------------------------------------------------------------
constexpr int kW2MaxG = 1024;
constexpr int kW2J = 128;        // pairs per chunk
constexpr int kW2Rec = 8;        // ints per workgroup record in the plan
constexpr int kXcds = 8;         // MI355X: workgroup b of a launch runs on XCD b % 8

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

// plan2 layout (int32):
//   [0] number of segments   [1] pairs per workgroup
//   [8 + 8 w ..]             workgroup w: first segment, number of segments, then its first
//                            segment inline (offset k, first pair, end pair)      (G records)
//   [8 + 8 G ..]             first segment of offset k                           (kv + 1)
//   then 3 ints per segment: offset k, first pair, end pair (positions inside list k)
__host__ __device__ inline int plan2_wg(int w) { return 8 + kW2Rec * w; }
__host__ __device__ inline int plan2_kf(int G) { return 8 + kW2Rec * G; }
__host__ __device__ inline int plan2_seg(int G, int kv) { return 8 + kW2Rec * G + kv + 1; }
// work list of the second stage: [0] items, then (16-byte aligned) 4 ints per item: offset k |
// mode << 8, first element inside a 64x64 tile, first segment of k, segments of k -- everything a
// block needs comes with ONE scalar load.  At most kv * 256 items.
__host__ __device__ inline int plan2_red(int G, int kv) {
  return (plan2_seg(G, kv) + 3 * (G + kv) + 3) & ~3;
}

// exclusive prefix sum of one value per thread over the whole kW2MaxG-thread block (wave scans + the
// wave totals through LDS); returns the prefix, `total` = sum over the block.  One barrier.  The plan
// packs several small counters into one 64-bit value per scan.
template <typename T>
__device__ __forceinline__ T plan_scan(T v, T *wtot, T &total) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  T incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const T u = __shfl_up(incl, d, 64);
    if (lane >= d) incl += u;
  }
  if (lane == 63) wtot[wv] = incl;
  __syncthreads();
  T prefix = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < kW2MaxG / 64; ++w) {
    const T t = wtot[w];
    if (w < wv) prefix += t;
    total += t;
  }
  return prefix + incl - v;
}

// Segment cost padding of the plan (see wgrad_plan2_kernel): on unless switched off for an A/B run.
__device__ __forceinline__ bool subm_cost_pad(int n_in, int kv) {
  (void)n_in;
  (void)kv;
#ifdef SPX_WGRAD_NO_PAD
  return false;
#else
  return true;
#endif
}

// One block, three block-wide scans.  Everything is a prefix sum or a closed form of the kv list
// lengths: the pairs of all lists, laid end to end, are cut into G equal ranges (one per workgroup
// of the first stage); a segment is the part of one list inside one range.  A range finds its first
// list by bisection and walks on from there (1-2 lists at kv = 27), the number of ranges that touch
// list k is floor((end-1)/per) - floor(start/per) + 1.
//
// Which workgroup takes which range: workgroup b runs on XCD b % 8 (round-robin dispatch) and all
// ranges advance through their lists at about the same rate, so the ranges that start in the same
// eighth of their list -- the same eighth of the ROWS, every list being in row order -- go to the
// same XCD: its L2 then serves a row to the other offsets that use it, instead of every XCD pulling
// it out of the Infinity Cache once per offset.  rec[5] of workgroup b = its range.
__global__ void __launch_bounds__(kW2MaxG)
wgrad_plan2_kernel(const int32_t *__restrict__ num, int n_in, int kv, int subm, int G,
                   int32_t *__restrict__ plan) {
  typedef unsigned long long u64;
  __shared__ int start[130], lpad[130], kfirst[130], kcount[130], ritems[130];
  __shared__ int wtot_i[kW2MaxG / 64];
  __shared__ u64 wtot_a[kW2MaxG / 64], wtot_b[kW2MaxG / 64];
  const int tid = threadIdx.x;
  const int c = tid < kv ? list_count(num, kv, subm, n_in, tid) : 0;
  // The lists are laid end to end in COST units: a list of c pairs takes c + ov of them, the first ov being
  // the fixed price of a segment (pair words -> rows -> first MFMAs before anything overlaps, the partial
  // tile it writes).  Cutting by pairs alone put the 26 short lists of a sparse SubM rulebook (~120 pairs
  // each at BASELINE config 2) into a handful of ranges of 3-4 segments -- 3-4 dependent chains in a row,
  // the slowest workgroups of the launch; with the padding a short list fills most of a range by itself.
  const int ov = (c > 0 && subm_cost_pad(n_in, kv)) ? kW2J + kW2J / 2 : 0;
  int total;
  const int st = plan_scan<int>(c + ov, wtot_i, total);
  if (tid <= kv) start[tid] = st;               // start[kv] = total (threads >= kv add nothing)
  if (tid < kv) lpad[tid] = ov;
  const int per = total > 0 ? (total + G - 1) / G : 1;
  __syncthreads();
  // pairs of list k inside the cost interval [a, b): list positions [pa, pb)
  auto pairs_in = [&](int k, int a, int b, int &pa, int &pb) {
    const int p0 = start[k] + lpad[k], len = start[k + 1] - p0;
    pa = min(max(a - p0, 0), len);
    pb = min(max(b - p0, 0), len);
  };

  // ---- ranges: first list, number of segments, XCD
  int lo = 0, hi = 0, mine = 0, k0 = 0, x = 0;
  if (tid < G) {
    lo = min(total, tid * per);
    hi = min(total, lo + per);
    if (hi > lo) {
      int a = 0, b = kv;                        // smallest k with start[k + 1] > lo
      while (a < b) {
        const int m = (a + b) >> 1;
        if (start[m + 1] > lo) b = m;
        else a = m + 1;
      }
      k0 = a;
      for (int k = k0; k < kv && start[k] < hi; ++k) {
        int pa, pb;
        pairs_in(k, lo, hi, pa, pb);
        mine += pb > pa ? 1 : 0;
      }
      int pa, pb;
      pairs_in(k0, lo, hi, pa, pb);
      const int len = start[k0 + 1] - start[k0] - lpad[k0];
      x = min(kXcds - 1, static_cast<int>(static_cast<long long>(pa) * kXcds / (len > 0 ? len : 1)));
    }
  }
  // ---- per offset: segments (= ranges touching the list), second-stage items
  int kc = 0, nitems = 0;
  if (tid < kv) {
    if (c > 0) kc = (start[tid + 1] - 1) / per - (start[tid] + ov) / per + 1;   // ranges touching the PAIRS of the list
    // second-stage work list: block shape by segment count (see wgrad_reduce2_kernel)
    const int mode = kc >= 48 ? 0 : (kc >= 6 ? 1 : 2);
    nitems = (kWT * kWT) / (mode == 0 ? 16 : (mode == 1 ? 128 : 512));
  }
  // scan A: segments per range (11 bits: <= G + kv) | ranges of XCD 0..4 (10 bits each)
  // scan B: ranges of XCD 5..7 (10 bits each) | segments per offset (11 bits, bit 30) | items (bit 41)
  u64 va = 0, vb = 0;
  if (tid < G) {
    va = static_cast<u64>(mine);
    if (x < 5) va |= 1ull << (11 + 10 * x);
    else vb = 1ull << (10 * (x - 5));
  }
  vb |= (static_cast<u64>(kc) << 30) | (static_cast<u64>(nitems) << 41);
  u64 ta, tb;
  const u64 ea = plan_scan<u64>(va, wtot_a, ta);
  const u64 eb = plan_scan<u64>(vb, wtot_b, tb);
  const int seg0 = static_cast<int>(ea & 0x7ff), nseg_total = static_cast<int>(ta & 0x7ff);
  const int kf = static_cast<int>((eb >> 30) & 0x7ff), ri = static_cast<int>(eb >> 41);
  const int kc_total = static_cast<int>((tb >> 30) & 0x7ff), items_total = static_cast<int>(tb >> 41);
  if (tid == 0) {
    plan[0] = nseg_total;
    plan[1] = per;
    plan[plan2_red(G, kv)] = items_total;
  }
  if (tid <= kv) {
    kfirst[tid] = kf;
    kcount[tid] = kc;
    ritems[tid] = ri;
    plan[plan2_kf(G) + tid] = tid == kv ? kc_total : kf;
  }
  int32_t *seg = plan + plan2_seg(G, kv);
  if (tid < G) {
    int sg = seg0;
    int32_t *rec = plan + plan2_wg(tid);
    int r2 = 0, r3 = 0, r4 = 0;
    bool first = true;
    if (hi > lo) {
      for (int k = k0; k < kv && start[k] < hi; ++k) {
        int a, b;
        pairs_in(k, lo, hi, a, b);
        if (b > a) {
          seg[3 * sg] = k;
          seg[3 * sg + 1] = a;
          seg[3 * sg + 2] = b;
          if (first) {
            r2 = k;
            r3 = a;
            r4 = b;
            first = false;
          }
          ++sg;
        }
      }
    }
    rec[0] = seg0;
    rec[1] = mine;
    rec[2] = r2;
    rec[3] = r3;
    rec[4] = r4;
    // position of this range in (XCD, rank) order -> the workgroup at the same position in
    // (b % 8, b / 8) order; XCD y owns ceil((G - y) / 8) workgroups
    int pos = x < 5 ? static_cast<int>((ea >> (11 + 10 * x)) & 0x3ff) : static_cast<int>((eb >> (10 * (x - 5))) & 0x3ff);
#pragma unroll
    for (int y = 0; y < kXcds; ++y) {
      const int cy = y < 5 ? static_cast<int>((ta >> (11 + 10 * y)) & 0x3ff) : static_cast<int>((tb >> (10 * (y - 5))) & 0x3ff);
      if (y < x) pos += cy;
    }
    int y = 0, base = 0;
    for (; y < kXcds - 1; ++y) {
      const int ny = (G - y + kXcds - 1) / kXcds;
      if (pos < base + ny) break;
      base += ny;
    }
    const int b = y + kXcds * (pos - base);
    plan[plan2_wg(b < G ? b : tid) + 5] = tid;
  }
  __syncthreads();
  int32_t *rl = plan + plan2_red(G, kv);
  for (int k = 0; k < kv; ++k) {
    const int md = kcount[k] >= 48 ? 0 : (kcount[k] >= 6 ? 1 : 2);
    const int E = md == 0 ? 16 : (md == 1 ? 128 : 512), cnt_items = (kWT * kWT) / E;
    for (int q = tid; q < cnt_items; q += kW2MaxG) {
      int32_t *item = rl + 4 + 4 * (ritems[k] + q);
      item[0] = k | (md << 8);
      item[1] = q * E;
      item[2] = kfirst[k];
      item[3] = kcount[k];
    }
  }
}

// byte offset of 16-byte slot `sl` of pair row `row` in a [128 pairs][64 channels] stage; the
// 32-byte granule index is XORed with (bit 1, bit 3) of the row: the 32 lanes of one
// ds_read_b64_tr_b16 half (rows r..r+3 and r+8..r+11, one granule each) cover all 64 banks.
__device__ __forceinline__ int wtr_x(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }
__device__ __forceinline__ int wtr_slot(int row, int sl) {
  return row * 128 + ((((sl >> 1) ^ wtr_x(row)) << 5) | ((sl & 1) << 4));
}

typedef short s16x4 __attribute__((ext_vector_type(4)));

// 8 consecutive pairs (row0 .. row0+7 as seen by this lane group) of channel granule*16 + lrow
__device__ __forceinline__ uint4 wtr_frag(const char *stage, int row0, int lrow, int gran) {
  const int r = row0 + (lrow >> 2);
  const char *a0 = stage + r * 128 + ((gran ^ wtr_x(r)) << 5) + ((lrow & 3) << 3);
  const int r1 = r + 4;
  const char *a1 = stage + r1 * 128 + ((gran ^ wtr_x(r1)) << 5) + ((lrow & 3) << 3);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_s16x4 *)(__attribute__((address_space(3))) char *)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_s16x4 *)(__attribute__((address_space(3))) char *)a1);
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// STAGES = 2: one barrier per chunk (64 KB of LDS); STAGES = 1: two barriers per chunk, 32 KB
// (used when the kernel shares a launch with dgrad, see igemm_bwd_kernel)
// SL: 16-byte slots of a row that are loaded at all (8 = 64 channels; 4 / 2 when both C and K fit
// 32 / 16 channels: a thread then covers SL / 2 rows per operand instead of 4, with half / a quarter
// of the load instructions per chunk; 7-9 % at 0.3 M - 1.2 M voxels, standalone launch only)
template <bool BF16, int STAGES, int SL = 8>
__device__ __forceinline__ void wgrad_tr_body(const Wgrad2Params &p, int block) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE_B = kW2J * 128;                // one operand tile: 128 pairs x 128 bytes
  constexpr int RQ = SL / 2;                        // rows per thread and operand
  constexpr int RSTEP = 2 * kW2J / SL;              // distance between a thread's rows
  constexpr int RA = RQ < 2 ? 2 : RQ;               // (register arrays stay at >= 2 elements)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int wk = wave >> 1, wc = wave & 1;          // wave quadrant: kk [32*wk,+32), c [32*wc,+32)
  const int ntile = p.tiles_k * p.tiles_c;
  const int wb = block / ntile, tile = block - wb * ntile;
  const int w = p.xcd_order ? p.plan2[plan2_wg(wb) + 5] : wb;   // the range this workgroup takes (see the plan)
  const int kk0 = (tile / p.tiles_c) * kWT, c0 = (tile % p.tiles_c) * kWT;
  const int32_t *__restrict__ rec = p.plan2 + plan2_wg(w);     // uniform address: scalar loads
  const int32_t *__restrict__ segs = p.plan2 + plan2_seg(p.G, p.kv);
  const int seg_lo = rec[0], nseg = rec[1];
  const bool live = kk0 + wk * 32 < p.K && c0 + wc * 32 < p.C;   // wave-uniform
  SPX_STAMP(0);

  const uint32_t rowD = static_cast<uint32_t>(p.K) * 2u, rowF = static_cast<uint32_t>(p.C) * 2u;
  const __amdgpu_buffer_rsrc_t rD = make_rsrc(p.dout, static_cast<uint32_t>(p.n_out) * rowD);
  const __amdgpu_buffer_rsrc_t rF = make_rsrc(p.feat, static_cast<uint32_t>(p.n_in) * rowF);
  // both pair lists of an offset (in: native[0][k], out: native[1][k] = kv * n_in words further on) are
  // read through ONE resource, so that one load per wave fetches the words of a whole chunk
  const uint32_t list_bytes = (static_cast<uint32_t>(p.kv) + 1u) * static_cast<uint32_t>(p.n_in) * 4u;
  const uint32_t out_list = static_cast<uint32_t>(p.kv) * static_cast<uint32_t>(p.n_in) * 4u;

  // load role: 16-byte slot `slot` of rows r0 + RSTEP q (q < RQ) of both operand tiles
  const int slot = tid & (SL - 1), r0 = tid / SL;
  const uint32_t dcol = kk0 + slot * 8 < p.K ? static_cast<uint32_t>(kk0 + slot * 8) * 2u : kOob;
  const uint32_t fcol = c0 + slot * 8 < p.C ? static_cast<uint32_t>(c0 + slot * 8) * 2u : kOob;
  // pair-list words: the 64 / SL rows of a wave x RQ steps are 32 pairs per chunk; lane L < 32 fetches the
  // in-word of pair L of the wave, lane 32 + L its out-word (one buffer_load_dword per wave and chunk
  // instead of 2 RQ with every word fetched SL times); the 2 RQ words a thread needs come back through
  // ds_bpermute
  constexpr int W_ROWS = 64 / SL;
  const int wl_idx = lane & 31;
  const int wl_row = (64 * wave) / SL + (wl_idx % W_ROWS) + RSTEP * (wl_idx / W_ROWS);
  const uint32_t wl_list = (lane >> 5) ? out_list : 0u;
  const int rsub = lane / SL;                       // this thread's row among the wave's W_ROWS rows
  int lds_w[RA];
#pragma unroll
  for (int q = 0; q < RQ; ++q) lds_w[q] = wtr_slot(r0 + RSTEP * q, slot);
  if constexpr (SL < 8) {
    // slots nobody loads stay zero for the whole launch
    for (int o = tid * 16; o < STAGES * 2 * TILE_B; o += kThreads * 16)
      *reinterpret_cast<u32x4 *>(smem + o) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
  }

  for (int si = 0; si < nseg; ++si) {
    int k, begin, end;
    if (si == 0) {
      k = rec[2];
      begin = rec[3];
      end = rec[4];
    } else {
      k = segs[3 * (seg_lo + si)];
      begin = segs[3 * (seg_lo + si) + 1];
      end = segs[3 * (seg_lo + si) + 2];
    }
    const bool identity = p.subm && k == p.kv / 2;
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(p.native + static_cast<size_t>(k) * p.n_in, list_bytes);

    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // One chunk of rows in flight per workgroup, the pair-list words one chunk further ahead.  Deeper
    // pipelines (two row sets, two word sets, a single LDS stage with three workgroups per CU, 768 or
    // 1023 ranges) were all measured within 3 % of this on 0.3 M - 1.2 M voxel levels
    // (profiles/r02_dense_regime_experiments.md): at that size the loop is bound by the 128-byte
    // lines the gathers pull out of the Infinity Cache, once per offset, whatever the row width.
    uint32_t wd = 0;              // this lane's pair-list word of the chunk whose rows are fetched next
    u32x4 dv[RA], fv[RA];
    auto load_words = [&](int base) __attribute__((always_inline)) {
      const int j = base + wl_row;
      if (identity) {
        wd = static_cast<uint32_t>(j);
      } else {
        const uint32_t vo = j < end ? static_cast<uint32_t>(j) * 4u + wl_list : kOob;
        wd = __builtin_amdgcn_raw_buffer_load_b32(rW, vo, 0, SPX_AUX_TABLE);
      }
    };
    auto load_rows = [&](int base) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < RQ; ++q) {
        const uint32_t iq = static_cast<uint32_t>(
            __builtin_amdgcn_ds_bpermute((q * W_ROWS + rsub) * 4, static_cast<int>(wd)));
        const uint32_t oq = static_cast<uint32_t>(
            __builtin_amdgcn_ds_bpermute((32 + q * W_ROWS + rsub) * 4, static_cast<int>(wd)));
        const bool ok = base + r0 + RSTEP * q < end;      // rows past the segment read as zero
        dv[q] = __builtin_amdgcn_raw_buffer_load_b128(rD, ok ? (oq * rowD + dcol) | (dcol & kOob) : kOob, 0, 0);
        fv[q] = __builtin_amdgcn_raw_buffer_load_b128(rF, ok ? (iq * rowF + fcol) | (fcol & kOob) : kOob, 0, 0);
      }
    };
    load_words(begin);
    load_rows(begin);
    load_words(begin + kW2J);
    if (si == 0) SPX_STAMP(1);   // first rows issued
    int stage = 0;
    for (int base = begin; base < end; base += kW2J, stage ^= (STAGES - 1)) {
      char *sD = smem + stage * (2 * TILE_B);
      char *sF = sD + TILE_B;
      if (STAGES == 1) __syncthreads();   // the previous chunk's fragment reads are done
#pragma unroll
      for (int q = 0; q < RQ; ++q) {
        *reinterpret_cast<u32x4 *>(sD + lds_w[q]) = dv[q];
        *reinterpret_cast<u32x4 *>(sF + lds_w[q]) = fv[q];
      }
      __syncthreads();   // stage complete; the other stage was last read one iteration ago
      load_rows(base + kW2J);         // in flight during the MFMAs (out of range past the end)
      load_words(base + 2 * kW2J);
      if (live)                       // (waves whose 32 x 32 quadrant lies outside K x C idle)
#pragma unroll
      for (int ks = 0; ks < kW2J / 32; ++ks) {
        const int row0 = ks * 32 + lgrp * 8;
        uint4 fa[2], fb[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[a] = wtr_frag(sD, row0, lrow, wk * 2 + a);
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[b] = wtr_frag(sF, row0, lrow, wc * 2 + b);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = mfma16<BF16>(fa[a], fb[b], acc[a][b]);
      }
    }
    if (si == nseg - 1) SPX_STAMP(4);   // last chunk loop done
    // D[i = kk][j = c]: lane holds c = lane & 15, kk = (lane >> 4) * 4 + reg
    float *dst = p.partial + (static_cast<size_t>(seg_lo + si) * ntile + tile) * (kWT * kWT);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = wk * 32 + a * 16 + lgrp * 4 + e;
          const int c = wc * 32 + b * 16 + lrow;
          if (SPX_AUX_OUT) __builtin_nontemporal_store(acc[a][b][e], &dst[kk * kWT + c]);
          else dst[kk * kWT + c] = acc[a][b][e];
        }
    __syncthreads();  // both stages are rewritten by the next segment
  }
  SPX_STAMP(6);
#ifdef SPX_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SPX_STAMP(7);
#endif
}

template <bool BF16, int SL = 8>
__global__ void __launch_bounds__(kThreads)
wgrad_tr_kernel(Wgrad2Params p) {
  wgrad_tr_body<BF16, 2, SL>(p, blockIdx.x);
}

// fp32 wgrad on v_mfma_f32_16x16x4_f32 over the same balanced segments.  Each lane feeds ONE
// element per operand, so the tiles are read from LDS as they lie ([pair][channel], row stride
// 80 floats: the two rows a 32-lane ds_read_b32 group touches fall on disjoint bank halves) and
// no transposition is needed.  64-pair chunks, one LDS stage, next chunk's rows in flight
// during the MFMAs.
constexpr int kW3J = 64;          // pairs per chunk
constexpr int kW3Stride = 80;     // floats per LDS row

__device__ __forceinline__ void wgrad_f32_body(const Wgrad2Params &p, int block) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *sD = reinterpret_cast<float *>(smem);
  float *sF = sD + kW3J * kW3Stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int wk = wave >> 1, wc = wave & 1;
  const int ntile = p.tiles_k * p.tiles_c;
  const int w = block / ntile, tile = block - w * ntile;
  const int kk0 = (tile / p.tiles_c) * kWT, c0 = (tile % p.tiles_c) * kWT;
  const int32_t *__restrict__ rec = p.plan2 + plan2_wg(w);
  const int32_t *__restrict__ segs = p.plan2 + plan2_seg(p.G, p.kv);
  const int seg_lo = rec[0], nseg = rec[1];
  const uint32_t rowD = static_cast<uint32_t>(p.K) * 4u, rowF = static_cast<uint32_t>(p.C) * 4u;
  const __amdgpu_buffer_rsrc_t rD = make_rsrc(p.dout, static_cast<uint32_t>(p.n_out) * rowD);
  const __amdgpu_buffer_rsrc_t rF = make_rsrc(p.feat, static_cast<uint32_t>(p.n_in) * rowF);
  const uint32_t list_bytes = static_cast<uint32_t>(p.n_in) * 4u;
  // load role: 16-byte slot `slot` (4 channels) of rows r0 + 16 q (q = 0..3) of both tiles
  const int slot = tid & 15, r0 = tid >> 4;
  const uint32_t dcol = kk0 + slot * 4 < p.K ? static_cast<uint32_t>(kk0 + slot * 4) * 4u : kOob;
  const uint32_t fcol = c0 + slot * 4 < p.C ? static_cast<uint32_t>(c0 + slot * 4) * 4u : kOob;

  for (int si = 0; si < nseg; ++si) {
    int k, begin, end;
    if (si == 0) {
      k = rec[2];
      begin = rec[3];
      end = rec[4];
    } else {
      k = segs[3 * (seg_lo + si)];
      begin = segs[3 * (seg_lo + si) + 1];
      end = segs[3 * (seg_lo + si) + 2];
    }
    const bool identity = p.subm && k == p.kv / 2;
    const __amdgpu_buffer_rsrc_t rIn =
        make_rsrc(p.native + static_cast<size_t>(k) * p.n_in, list_bytes);
    const __amdgpu_buffer_rsrc_t rOut =
        make_rsrc(p.native + static_cast<size_t>(p.kv + k) * p.n_in, list_bytes);
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 dv[4], fv[4];
    auto load_rows = [&](int base) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = base + r0 + 16 * q;
        const bool ok = j < end;
        uint32_t ii = static_cast<uint32_t>(j), oi = static_cast<uint32_t>(j);
        if (!identity) {
          const uint32_t vo = ok ? static_cast<uint32_t>(j) * 4u : kOob;
          ii = __builtin_amdgcn_raw_buffer_load_b32(rIn, vo, 0, 0);
          oi = __builtin_amdgcn_raw_buffer_load_b32(rOut, vo, 0, 0);
        }
        dv[q] = __builtin_amdgcn_raw_buffer_load_b128(rD, ok ? (oi * rowD + dcol) | (dcol & kOob) : kOob, 0, 0);
        fv[q] = __builtin_amdgcn_raw_buffer_load_b128(rF, ok ? (ii * rowF + fcol) | (fcol & kOob) : kOob, 0, 0);
      }
    };
    load_rows(begin);
    for (int base = begin; base < end; base += kW3J) {
      __syncthreads();   // the previous chunk's fragment reads are done
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<u32x4 *>(sD + (r0 + 16 * q) * kW3Stride + slot * 4) = dv[q];
        *reinterpret_cast<u32x4 *>(sF + (r0 + 16 * q) * kW3Stride + slot * 4) = fv[q];
      }
      __syncthreads();
      load_rows(base + kW3J);          // in flight during the MFMAs (out of range past the end)
#pragma unroll 4
      for (int ks = 0; ks < kW3J / 4; ++ks) {
        const int row = (ks * 4 + lgrp) * kW3Stride;
        float fa[2], fb[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[a] = sD[row + wk * 32 + a * 16 + lrow];
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[b] = sF[row + wc * 32 + b * 16 + lrow];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
    }
    float *dst = p.partial + (static_cast<size_t>(seg_lo + si) * ntile + tile) * (kWT * kWT);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = wk * 32 + a * 16 + lgrp * 4 + e;
          const int c = wc * 32 + b * 16 + lrow;
          if (SPX_AUX_OUT) __builtin_nontemporal_store(acc[a][b][e], &dst[kk * kWT + c]);
          else dst[kk * kWT + c] = acc[a][b][e];
        }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(kThreads)
wgrad_f32_kernel(Wgrad2Params p) {
  wgrad_f32_body(p, blockIdx.x);
}

// Backward of one layer in ONE launch: the wgrad ranges (the longer, streaming workgroups)
// are dispatched first, the dgrad tiles after them.  The two halves only share read-only inputs; at ~100k voxels each of them
// is latency-bound with idle issue slots and idle HBM bandwidth, so running them side by side
// on the same CUs costs little more than the slower one -- and one kernel boundary (~1.7 us)
// disappears.  (Two HIP streams were tried first: the fork/join costs more than it buys.)
template <int COUT, int MB, int DT, int NKS = 2>
__global__ void __launch_bounds__(kThreads, COUT <= 64 ? 4 : 2)   // 4 waves/SIMD: 1024 resident workgroups
igemm_bwd_kernel(const void *argA, const void *argB, const uint32_t *arg_mask,
                 const int32_t *arg_argsort, const int32_t *arg_pair, int n_dst, int n_src, int CIN,
                 int kv, int identity_k, int b_reverse, GemmRest rest, int n_dgrad, Wgrad2Params wp) {
  // n_dgrad > 0: dgrad tiles first, then the wgrad ranges; n_dgrad < 0: the wgrad ranges
  // (-n_dgrad - 1 ... encoded as ~count) first
  const int nw = n_dgrad < 0 ? ~n_dgrad : 0;          // wgrad workgroups placed first
  const int b = static_cast<int>(blockIdx.x);
  const bool is_dgrad = n_dgrad < 0 ? b >= nw : b < n_dgrad;
  if (is_dgrad) {
    GemmParams p;
    unpack_gemm_args(p, argA, argB, arg_mask, arg_argsort, arg_pair, n_dst, n_src, CIN, kv,
                     identity_k, b_reverse, rest);
    p.xcd_rot = (rest.dbg & 0x100) ? 0 : (nw & 7);      // (SPX_V4_DBG=256: A/B switch)
    p.app_budget = kAppBudget;
    igemm_v4_body<COUT, MB, DT, true, NKS>(p, n_dgrad < 0 ? b - nw : b);
  } else {
    if constexpr (DT == 3) wgrad_f32_body(wp, n_dgrad < 0 ? b : b - n_dgrad);
    else wgrad_tr_body<DT == 1, 1>(wp, n_dgrad < 0 ? b : b - n_dgrad);
  }
}

// dw[kk][k][c] = sum over the segments of offset k (deterministic: fixed assignment of
// segments to threads, fixed summation order).  The work list (built with the plan) gives
// every offset a block shape that fits its segment count -- 16 elements x 128 segment groups
// for long lists, 128 x 4 or 512 x 1 for short ones -- so a SubM rulebook with one long and
// 26 short lists runs ~460 blocks of useful work instead of 27 x 128 mostly idle ones.
template <typename T>
__global__ void __launch_bounds__(kRedThreads)
wgrad_reduce2_kernel(Wgrad2Params p, T *__restrict__ dw) {
  __shared__ float red[kRedThreads];
  const int32_t *__restrict__ rl = p.plan2 + plan2_red(p.G, p.kv);
  const int nitems = rl[0];
  const int ntile = p.tiles_k * p.tiles_c;
  const size_t stride = static_cast<size_t>(ntile) * (kWT * kWT);
  const int tile = blockIdx.y;
  for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
    const int4 item = *reinterpret_cast<const int4 *>(rl + 4 + 4 * it);    // uniform: one s_load_dwordx4
    const int k = item.x & 0xff, mode = item.x >> 8, e0 = item.y, first = item.z, nseg = item.w;
    const float *base = p.partial + static_cast<size_t>(first) * stride + tile * (kWT * kWT) + e0;
    if (mode == 0) {
      // long list: 16 elements (4 lanes x float4) x 128 segment groups; with a few hundred
      // segments every thread has all of its loads in flight at once.  Wave-level butterfly
      // over the 16 groups of a wave, then 8 wave sums through LDS.
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const int c4 = lane & 3, grp = wave * 16 + (lane >> 2);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const float *src = base + c4 * 4;
      int ch = grp;
      for (; ch + 3 * 128 < nseg; ch += 4 * 128) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[u] = *reinterpret_cast<const float4 *>(src + static_cast<size_t>(ch + u * 128) * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
        }
      }
      for (; ch < nseg; ch += 128) {
        const float4 v = *reinterpret_cast<const float4 *>(src + static_cast<size_t>(ch) * stride);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
#pragma unroll
      for (int d = 4; d < 64; d <<= 1) {
        acc.x += __shfl_xor(acc.x, d, 64);
        acc.y += __shfl_xor(acc.y, d, 64);
        acc.z += __shfl_xor(acc.z, d, 64);
        acc.w += __shfl_xor(acc.w, d, 64);
      }
      if (lane < 4) reinterpret_cast<float4 *>(red)[wave * 4 + lane] = acc;
      __syncthreads();
      if (threadIdx.x < 16) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < kRedThreads / 64; ++w) sum += red[w * 16 + threadIdx.x];
        const int ee = e0 + threadIdx.x;
        const int kk = (tile / p.tiles_c) * kWT + ee / kWT, c = (tile % p.tiles_c) * kWT + ee % kWT;
        if (kk < p.K && c < p.C) store_f(dw + (static_cast<size_t>(kk) * p.kv + k) * p.C + c, sum);
      }
      __syncthreads();   // red[] is reused by the next item
      continue;
    }
    // short lists: 128 elements x 4 groups, or 512 x 1
    const int E = mode == 1 ? 128 : 512, S = kRedThreads / E;
    const int grp = threadIdx.x / E, el = threadIdx.x % E;
    float acc = 0.f;
    for (int ch = grp; ch < nseg; ch += S) acc += base[static_cast<size_t>(ch) * stride + el];
    if (S > 1) {
      red[threadIdx.x] = acc;
      __syncthreads();
      if (grp == 0) {
        acc = 0.f;
        for (int g = 0; g < S; ++g) acc += red[g * E + el];
      }
    }
    if (grp == 0) {
      const int ee = e0 + el;
      const int kk = (tile / p.tiles_c) * kWT + ee / kWT, c = (tile % p.tiles_c) * kWT + ee % kWT;
      if (kk < p.K && c < p.C) store_f(dw + (static_cast<size_t>(kk) * p.kv + k) * p.C + c, acc);
    }
    if (S > 1) __syncthreads();   // red[] is reused by the next item
  }
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
bias_act_kernel(T *__restrict__ out, const T *__restrict__ bias, long long total, int K, int act,
                float alpha) {
  const long long gid = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (gid >= total) return;
  float v = load_f(out + gid);
  if (bias) v += load_f(bias + gid % K);
  store_f(out + gid, apply_act(v, act, alpha));
}

int elem_bytes(int dtype) { return dtype == SPX_F32 ? 4 : (dtype == SPX_I8 ? 1 : 2); }

bool mfma_ok(int dtype, int cin, int cout, int kv, const uint32_t *mask) {
  if (dtype != SPX_F16 && dtype != SPX_BF16 && dtype != SPX_F32) return false;
  if (cin % (dtype == SPX_F32 ? 4 : 8) != 0) return false;     // 16-byte lane pieces
  if (kv > 128) return false;                                  // 33 .. 128: groups of 32 offsets
  (void)mask;
  return cout == 16 || cout == 32 || cout == 64 || cout == 128 || cout == 256;
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
    if (mb == 4 && p.COUT == 64) return launch_v4<64, 4, BF16 ? 1 : 0>(p, s);     // experiment: 256-row tiles
    if (mb == 4 && p.COUT == 32) return launch_v4<32, 4, BF16 ? 1 : 0>(p, s);
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

// fp32 tensors: the same kernel on v_mfma_f32_16x16x4_f32 (128-row tiles; no v3 fallback)
int dispatch_gather_gemm_f32(const GemmParams &p, hipStream_t s) {
  switch (p.COUT) {
    case 16: return launch_v4<16, 2, 3>(p, s);
    case 32: return launch_v4<32, 2, 3>(p, s);
    case 64: return launch_v4<64, 2, 3>(p, s);
    case 128: return launch_v4<128, 2, 3>(p, s);
    case 256: return launch_v4<256, 1, 3>(p, s);
  }
  return -1;
}

int run_gather_gemm_single(const GemmParams &p, int dtype, hipStream_t s);

// Kernel volumes 33 .. 128 (5x5x5, 4-d 3^4, ...): the reference covers them with multi-word masks
// (indices.py:1601-1618, ops.py:448,494-503); here the layer runs as ceil(kv / 32) launches of the same
// kernel, one mask word each, whose partial sums travel through an fp32 [n_dst, COUT] scratch -- the
// result is rounded once, like a single launch.
int run_gather_gemm(const GemmParams &p, int dtype, hipStream_t s) {
  if (p.n_dst == 0) return 0;
  if (p.kv <= 32 || !mfma_ok(dtype, p.CIN, p.COUT, p.kv, p.mask) || !p.pair) return run_gather_gemm_single(p, dtype, s);
  const int words = div_up(p.kv, 32);
  const bool fits = static_cast<unsigned long long>(p.n_dst) * 4ull * words < 0x7fff0000ull &&
                    static_cast<unsigned long long>(p.n_dst) * p.COUT * 4ull < 0x7fff0000ull &&
                    v4_ok(p, dtype == SPX_F32 ? 4 : 2, dtype == SPX_F32 ? 4 : 2) && !p.argsort;
  if (!fits || !p.acc) {
    GemmParams q = p;                 // no scratch / beyond 32-bit offsets: the generic kernel
    q.acc = nullptr;
    return run_gather_gemm_single(q, dtype, s);
  }
  for (int g = 0; g < words; ++g) {
    GemmParams q = p;
    q.kbase = 32 * g;
    q.pair = p.pair + static_cast<size_t>(32 * g) * p.n_dst;
    q.mask = p.mask ? p.mask + g : nullptr;
    q.mask_words = p.mask ? words : 1;
    q.identity_k = (p.identity_k >= 32 * g && p.identity_k < 32 * g + 32) ? p.identity_k - 32 * g : -1;
    q.acc_mode = (g > 0 ? 1 : 0) | (g < words - 1 ? 2 : 0);
    if (int rc = run_gather_gemm_single(q, dtype, s)) return rc;
  }
  return 0;
}

int run_gather_gemm_single(const GemmParams &p, int dtype, hipStream_t s) {
  if (p.n_dst == 0) return 0;
  constexpr int f32_mfma = 1;
  const bool grouped = p.acc_mode != 0;
  if (dtype == SPX_F32 && f32_mfma && mfma_ok(dtype, p.CIN, p.COUT, p.kv, p.mask) && v4_ok(p, 4, 4) &&
      (p.kv <= 32 || grouped))
    return dispatch_gather_gemm_f32(p, s);
  if (dtype != SPX_F32 && mfma_ok(dtype, p.CIN, p.COUT, p.kv, p.mask) && (p.kv <= 32 || grouped))
    return dtype == SPX_BF16 ? dispatch_gather_gemm<true>(p, s) : dispatch_gather_gemm<false>(p, s);
  if (p.cls) {                     // (as above: the generic kernel reads the tables by row)
    GemmParams q = p;
    drop_rows_layout(q);
    return run_gather_gemm_single(q, dtype, s);
  }
  if (p.tile_order) {
    set_error("tables in tile order are supported by the MFMA kernels only (channel counts / kernel volume)");
    return -1;
  }
  const long long total = static_cast<long long>(p.n_dst) * p.COUT;
  const dim3 grid(static_cast<unsigned>((total + kThreads - 1) / kThreads));
  if (dtype == SPX_F32)
    hipLaunchKernelGGL(gather_gemm_generic_kernel<float>, grid, dim3(kThreads), 0, s, p);
  else if (dtype == SPX_F16)
    hipLaunchKernelGGL(gather_gemm_generic_kernel<h16>, grid, dim3(kThreads), 0, s, p);
  else if (dtype == SPX_BF16)
    hipLaunchKernelGGL(gather_gemm_generic_kernel<b16>, grid, dim3(kThreads), 0, s, p);
  else {
    set_error("unsupported dtype %d", dtype);
    return -1;
  }
  SPX_LAUNCH_CHECK();
  return 0;
}

int wgrad_chunk(int n_in) {
  constexpr int forced = 0;
  if (forced > 0) return (forced + kWJ - 1) / kWJ * kWJ;
  // aim at >= ~512 workgroups for the dominant (centre) list, multiples of 128
  int c = (n_in / 512 + kWJ - 1) / kWJ * kWJ;
  if (c < kWJ) c = kWJ;
  if (c > 1024) c = 1024;
  return c;
}

size_t wgrad_plan_ints(int n_in, int kv) {
  const size_t nchunks = div_up(n_in > 0 ? n_in : 1, wgrad_chunk(n_in));
  return 1 + 2 * static_cast<size_t>(kv) + 2 * nchunks * kv;
}

// workgroups of the balanced wgrad: 1.5 per CU once there is enough work (more workgroups
// mean more partials for the second stage: 384 measured best at 100k voxels), never more
// ranges than twice the 128-pair chunks of the identity list
int wgrad_groups(int n_in, int subm) {
  constexpr int forced = 0;
  int g = forced > 0 ? forced : 384;
  if (forced <= 0) {
    // backward shares its launch with ceil(n / 128) dgrad tiles: when both halves fit the 1024
    // resident workgroup slots of the chip together there is no second dispatch round
    // (41.3 vs 44.7 us per step at 100 k voxels).  Beyond ~115 k voxels 384 stays: 128 ranges are
    // 7-13 % faster at C = 64 (the dgrad tiles alone fill the slots there), but the ranges become
    // chains of > 100 chunks and double the launch time of 16 / 32-channel layers, which is what
    // the large levels of a backbone are (igemm_bwd_kernel<32>: 158 -> 294 us at 450 k voxels);
    // the plan is built per rulebook, without knowing the layer widths that will use it
    // SubM rulebooks of that size carry a rows layout (spx_subm_layout): its appendix tiles lead the dgrad half of the
    // launch, kAppBudget of them at most (config 2: 50 tiles of 64 rows; step 25.7 -> 24.2-24.9 us with the room left)
    const int room = 1024 - div_up(n_in > 0 ? n_in : 1, 128) - ((subm && n_in >= kLayoutMinRows) ? kAppBudget : 0);
    if (room >= 128 && room < g) g = room;
  }
  const int chunks = div_up(n_in > 0 ? n_in : 1, 128);
  if (g > 2 * chunks) g = 2 * chunks;
  if (g > kW2MaxG - 1) g = kW2MaxG - 1;   // the plan kernel needs thread G for the end marker
  return g < 1 ? 1 : g;
}

// blocks of the second stage (block-stride over at most kv * 256 items)
int reduce2_blocks(int kv) {
  constexpr int cap = 512;
  return kv * 256 < cap ? kv * 256 : cap;
}

int wgrad_xcd_order() {
  constexpr int v = 1;
  return v;
}

// what the *_bytes functions size for: the larger of the two rules.  (They used to take wgrad_groups(n, 0) as "an
// upper bound of the SubM value", which the appendix budget broke for n in (106 496, 114 688]: 164 vs 384 ranges at
// n = 110 000 -- at small kernel volumes the plan kernel then wrote past the buffer; round-4 ADVICE.)
int wgrad_groups_max(int n_in) {
  const int a = wgrad_groups(n_in, 0), b = wgrad_groups(n_in, 1);
  return a > b ? a : b;
}

size_t wgrad_plan2_ints(int n_in, int kv) {
  const size_t G = wgrad_groups_max(n_in);
  return 8 + kW2Rec * G + kv + 1 + 3 * (G + kv) + 4 + 4 + 4 * static_cast<size_t>(kv) * 256 + 8;
}

GemmParams dgrad_params(const void *dout, const void *weight, void *din, const int32_t *pair,
                        const uint32_t *mask, const int32_t *argsort, int n_out, int n_in, int C,
                        int K, int kv, int subm) {
  GemmParams p{};
  p.A = dout;
  p.B = weight;                                   // KRSC read in place: (k, n=c, d=kk)
  p.out = din;
  p.pair = pair;
  p.mask = mask;
  p.argsort = argsort;
  p.bias = nullptr;
  p.strideK = C;
  p.strideN = 1;
  p.strideD = static_cast<long long>(kv) * C;
  p.n_src = n_out;
  p.n_dst = n_in;
  p.CIN = K;
  p.COUT = C;
  p.kv = kv;
  p.identity_k = subm ? kv / 2 : -1;
  p.b_reverse = subm ? 1 : 0;
  p.act = SPX_ACT_NONE;
  p.act_alpha = 0.f;
  return p;
}

GemmRest rest_of(const GemmParams &p) {
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
  return r;
}

// LDS of the fused backward launch: the dgrad weight ring or one wgrad stage pair
template <int COUT, int MB, int DT = 0>
constexpr size_t bwd_smem_bytes() {
  const size_t a = v4_smem_bytes<COUT, MB>();
  const size_t b = DT == 3 ? 2 * static_cast<size_t>(kW3J) * kW3Stride * sizeof(float)
                           : 2 * static_cast<size_t>(kW2J) * 128;
  return a > b ? a : b;
}

template <int COUT, int MB, int DT>
int launch_bwd(const GemmParams &p, const Wgrad2Params &q, int n_wgrad_blocks, hipStream_t s) {
  const int napp = p.cls ? layout_app_tiles(p.n_dst, 64 * MB) : 0;
  const int n_dgrad = div_up(p.n_dst, 64 * MB) + napp;
  GemmRest rr = rest_of(p);
  rr.napp = p.cls ? napp : -1;
  constexpr int wgrad_first = 1;           // (the longer chains are dispatched first: settled A/B)
  GemmParams pl = p;
  pl.lpt = p.tile_order && n_dgrad + n_wgrad_blocks > 1024;           // (see launch_v4)
  if (p.CIN * (DT == 3 ? 4 : 2) <= 64)     // dgrad's reduction rows (dout channels) fit half a piece
    hipLaunchKernelGGL((igemm_bwd_kernel<COUT, MB, DT, 1>), dim3(n_dgrad + n_wgrad_blocks), dim3(kThreads),
                       (bwd_smem_bytes<COUT, MB, DT>()), s, p.A, p.B, p.mask, p.argsort, p.pair, p.n_dst,
                       p.n_src, p.CIN, p.kv, p.identity_k, v4_flags(pl), rr,
                       wgrad_first ? ~n_wgrad_blocks : n_dgrad, q);
  else
    hipLaunchKernelGGL((igemm_bwd_kernel<COUT, MB, DT, 2>), dim3(n_dgrad + n_wgrad_blocks), dim3(kThreads),
                       (bwd_smem_bytes<COUT, MB, DT>()), s, p.A, p.B, p.mask, p.argsort, p.pair, p.n_dst,
                       p.n_src, p.CIN, p.kv, p.identity_k, v4_flags(pl), rr,
                       wgrad_first ? ~n_wgrad_blocks : n_dgrad, q);
  SPX_LAUNCH_CHECK();
  return 0;
}

template <int DT>
int dispatch_bwd(const GemmParams &p, const Wgrad2Params &q, int nwb, hipStream_t s) {
  switch (p.COUT) {
    case 16: return launch_bwd<16, 2, DT>(p, q, nwb, s);
    case 32: return launch_bwd<32, 2, DT>(p, q, nwb, s);
    case 64: return launch_bwd<64, 2, DT>(p, q, nwb, s);
    case 128: return launch_bwd<128, 2, DT>(p, q, nwb, s);
  }
  return -1;
}

}  // namespace
}  // namespace spx

using namespace spx;

extern "C" {

size_t spx_igemm_acc_bytes(int n_dst, int cout, int kv) {
  return kv > 32 ? align_up(static_cast<size_t>(n_dst > 0 ? n_dst : 1) * cout * sizeof(float), 256) : 0;
}

int spx_igemm_fwd(const void *feat, const void *weight, void *out, const int32_t *pair,
                  const uint32_t *mask, const int32_t *argsort, int tile_order, int n_in, int n_out, int C,
                  int K, int kv, int dtype, int identity_k, const void *bias, int act,
                  float act_alpha, void *ws, size_t ws_bytes, spx_stream_t stream) {
  SPX_CHECK(C > 0 && K > 0 && kv > 0 && n_in >= 0 && n_out >= 0, "bad sizes");
  if (n_out == 0) return 0;                                   // empty scene: nothing to write
  SPX_CHECK((feat || n_in == 0) && weight && out, "null tensor pointer");
  SPX_CHECK(pair || kv == 1, "pair table required");
  GemmParams p{};
  p.A = feat;
  p.B = weight;
  p.out = out;
  p.pair = pair;
  p.mask = mask;
  p.argsort = argsort;
  p.bias = bias;
  p.strideK = C;                                  // KRSC: W[n][k][c]
  p.strideN = static_cast<long long>(kv) * C;
  p.strideD = 1;
  p.n_src = n_in;
  p.n_dst = n_out;
  p.CIN = C;
  p.COUT = K;
  p.kv = kv;
  p.identity_k = identity_k;
  p.b_reverse = 0;
  p.dense_hint = (tile_order & SPX_DENSE_HINT) ? 1 : 0;
  apply_rows_layout(p, tile_order & ~SPX_DENSE_HINT);
  p.act = act & 0xff;
  if (act & SPX_OUT_CACHED) p.dbg = 0x400;       // plain result stores: the next launch reads the rows
  p.act_alpha = act_alpha;
  if (ws && ws_bytes >= spx_igemm_acc_bytes(n_out, K, kv) && kv > 32) p.acc = static_cast<float *>(ws);
  return run_gather_gemm(p, dtype, static_cast<hipStream_t>(stream));
}

int spx_igemm_fwd_int8(const void *feat, const void *weight, void *out, const int32_t *pair,
                       const uint32_t *mask, const int32_t *argsort, int n_in, int n_out, int C,
                       int K, int kv, int identity_k, const float *scale, const float *bias,
                       const void *add, float add_scale, int out_dtype, int act, float act_alpha,
                       spx_stream_t stream) {
  SPX_CHECK(C > 0 && K > 0 && kv > 0 && n_in >= 0 && n_out >= 0, "bad sizes");
  if (n_out == 0) return 0;                                   // empty scene: nothing to write
  SPX_CHECK((feat || n_in == 0) && weight && out, "null tensor pointer");
  SPX_CHECK(pair || kv == 1, "pair table required");
  // the reference has the same restriction (test/test_all_algo.py:376-377)
  SPX_CHECK(C % 16 == 0, "int8 needs in_channels %% 16 == 0, got %d", C);
  SPX_CHECK(K == 16 || K == 32 || K == 64 || K == 128 || K == 256,
            "int8 supports out_channels 16/32/64/128/256, got %d", K);
  SPX_CHECK(kv <= 32, "int8 supports kernel volumes up to 32, got %d", kv);
  SPX_CHECK(out_dtype == SPX_I8 || out_dtype == SPX_F16 || out_dtype == SPX_BF16 || out_dtype == SPX_F32,
            "bad output dtype %d", out_dtype);
  GemmParams p{};
  p.A = feat;
  p.B = weight;
  p.out = out;
  p.pair = pair;
  p.mask = mask;
  p.argsort = argsort;
  p.bias = bias;
  p.strideK = C;
  p.strideN = static_cast<long long>(kv) * C;
  p.strideD = 1;
  p.n_src = n_in;
  p.n_dst = n_out;
  p.CIN = C;
  p.COUT = K;
  p.kv = kv;
  p.identity_k = identity_k;
  p.b_reverse = 0;
  apply_rows_layout(p, (act & SPX_ROWS_LAYOUT_ACT) ? SPX_ROWS_LAYOUT : ((act & SPX_TILE_ORDER) ? 1 : 0));
  const bool hinted = p.cls && (act & SPX_SPARSE_HINT);   // the host has seen class word 1: a launch-shape hint
  if (hinted) p.app_rows = ((act >> 16) & 0xffff) * 64;   // ... and M (in units of 64 rows, 0 = not told)
  p.act = act & 0xff;
  p.act_alpha = act_alpha;
  p.scale = scale;
  p.add = add;
  p.add_scale = add_scale;
  p.out_dtype = out_dtype;
  if (n_out == 0) return 0;
  const int oes = out_dtype == SPX_I8 ? 1 : (out_dtype == SPX_F32 ? 4 : 2);
  SPX_CHECK(v4_ok(p, 1, oes), "tensor too large for 32-bit buffer offsets");
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (K) {
    case 16: return launch_v4<16, 2, 2>(p, s);
    case 32: return launch_v4<32, 2, 2>(p, s);
    case 64: return launch_v4<64, 2, 2>(p, s);
    case 128: {
      // tile height by density class: tables in tile order = a rulebook the host classified as SPARSE
      // (ops.sparse_neighbourhoods) -> 64-row tiles (70 instead of 165 registers per lane, five workgroups per CU
      // instead of three: 27.3 -> 25.7 us at BASELINE config 5); dense neighbourhoods keep 128 rows (LiDAR-like
      // 200 k: 104 vs 113 us, fixture 74 vs 83 us).  SPX_I8_MB = 1 / 2 forces one.
      constexpr int forced = 0;
      if (forced == 1 || (forced == 0 && (p.tile_order || hinted))) return launch_v4<128, 1, 2>(p, s);
      return launch_v4<128, 2, 2>(p, s);
    }
    case 256: return launch_v4<256, 1, 2>(p, s);
  }
  return -1;
}

size_t spx_igemm_dgrad_ws_bytes(int C, int K, int kv, int dtype) {
  (void)C; (void)K; (void)kv; (void)dtype;
  return 0;  // the weight transpose happens inside the kernel (LDS staging); kv > 32: spx_igemm_acc_bytes
}

int spx_igemm_dgrad(const void *dout, const void *weight, void *din, const int32_t *pair,
                    const uint32_t *mask, const int32_t *argsort, int tile_order, int n_out, int n_in, int C,
                    int K, int kv, int dtype, int subm, void *ws, size_t ws_bytes,
                    spx_stream_t stream) {
  if (n_in == 0) return 0;                                    // empty input: no gradient rows
  SPX_CHECK((dout || n_out == 0) && weight && din, "null tensor pointer");
  SPX_CHECK(pair || kv == 1, "pair table required");
  GemmParams p = dgrad_params(dout, weight, din, pair, mask, argsort, n_out, n_in, C, K, kv, subm);
  p.dense_hint = (tile_order & SPX_DENSE_HINT) ? 1 : 0;
  apply_rows_layout(p, tile_order & ~SPX_DENSE_HINT);
  if (ws && ws_bytes >= spx_igemm_acc_bytes(n_in, C, kv) && kv > 32) p.acc = static_cast<float *>(ws);
  return run_gather_gemm(p, dtype, static_cast<hipStream_t>(stream));
}

// the plan blob holds both forms: the item list of the generic kernels and, behind it, the
// balanced segment plan of the MFMA kernel
static size_t plan1_bytes(int n_in, int kv) {
  return align_up(wgrad_plan_ints(n_in, kv) * sizeof(int32_t), 256);
}

size_t spx_wgrad_plan_bytes(int n_in, int kv) {
  return plan1_bytes(n_in, kv) + align_up(wgrad_plan2_ints(n_in, kv) * sizeof(int32_t), 256);
}

int spx_wgrad_plan(const int32_t *num_per_loc, int n_in, int kv, int subm, int32_t *plan,
                   spx_stream_t stream) {
  SPX_CHECK(num_per_loc && plan, "null pointer");
  SPX_CHECK(kv >= 1 && kv <= 128, "kernel volume %d not supported by wgrad (max 128)", kv);
  // the first-generation item list (front of the buffer) is only read by the fallback kernels of
  // spx_igemm_wgrad, which build it themselves when they run: one launch less per rulebook
  int32_t *plan2 = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(plan) + plan1_bytes(n_in, kv));
  const int G = wgrad_groups(n_in, subm);
  SPX_CHECK(G <= wgrad_groups_max(n_in), "wgrad plan: %d ranges exceed the %d the plan buffer is sized for", G,
            wgrad_groups_max(n_in));
  hipLaunchKernelGGL(wgrad_plan2_kernel, dim3(1), dim3(kW2MaxG), 0, static_cast<hipStream_t>(stream),
                     num_per_loc, n_in, kv, subm, G, plan2);
  SPX_LAUNCH_CHECK();
  return 0;
}

size_t spx_igemm_wgrad_ws_bytes(int n_in, int C, int K, int kv) {
  const int chunk = wgrad_chunk(n_in);
  const size_t nchunks = div_up(n_in > 0 ? n_in : 1, chunk);
  const size_t tiles = static_cast<size_t>(div_up(C, kWT)) * div_up(K, kWT);
  size_t parts = nchunks * kv;                                   // item list (generic kernels)
  const size_t segs = static_cast<size_t>(wgrad_groups_max(n_in)) + kv;   // balanced segments
  if (segs > parts) parts = segs;
  return align_up(parts * tiles * kWT * kWT * sizeof(float), 256) + spx_wgrad_plan_bytes(n_in, kv);
}

int spx_igemm_wgrad(const void *feat, const void *dout, void *dw, const int32_t *pair_native,
                    const int32_t *num_per_loc, const int32_t *plan, int n_in, int n_out, int C,
                    int K, int kv, int dtype, int subm, void *ws, size_t ws_bytes,
                    spx_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  SPX_CHECK(dw && C > 0 && K > 0, "null tensor pointer");
  SPX_CHECK(kv >= 1 && kv <= 128, "kernel volume %d not supported by wgrad (max 128)", kv);
  if (n_in == 0 || n_out == 0) {                              // no pairs: the gradient is zero
    SPX_HIP(hipMemsetAsync(dw, 0, static_cast<size_t>(K) * kv * C * elem_bytes(dtype), s));
    return 0;
  }
  SPX_CHECK(feat && dout && ws, "null tensor pointer");
  SPX_CHECK(pair_native && num_per_loc, "Native pair lists and counts are required");
  SPX_CHECK(ws_bytes >= spx_igemm_wgrad_ws_bytes(n_in, C, K, kv), "workspace too small");
  WgradParams p{};
  p.feat = feat;
  p.dout = dout;
  p.partial = static_cast<float *>(ws);
  p.native = pair_native;
  p.num = num_per_loc;
  p.n_in = n_in;
  p.n_out = n_out;
  p.C = C;
  p.K = K;
  p.kv = kv;
  p.subm = subm;
  p.chunk = wgrad_chunk(n_in);
  p.nchunks = div_up(n_in > 0 ? n_in : 1, p.chunk);
  p.tiles_c = div_up(C, kWT);
  p.tiles_k = div_up(K, kWT);
  const int ntile = p.tiles_c * p.tiles_k;
  if (!plan) {  // caller did not cache a plan: build it behind the partials
    int32_t *own = reinterpret_cast<int32_t *>(static_cast<char *>(ws) + ws_bytes -
                                               spx_wgrad_plan_bytes(n_in, kv));
    if (spx_wgrad_plan(num_per_loc, n_in, kv, subm, own, stream)) return -2;
    plan = own;
  }
  p.plan = plan;
  const bool mfma = (dtype == SPX_F16 || dtype == SPX_BF16) && C % 8 == 0 && K % 8 == 0;
  constexpr int wgrad_version = 2;
  const bool small_offsets = static_cast<unsigned long long>(n_out) * K * 2ull < 0x7fff0000ull &&
                             static_cast<unsigned long long>(n_in) * C * 2ull < 0x7fff0000ull &&
                             static_cast<unsigned long long>(n_in) * 4ull * (kv + 1) < 0x7fff0000ull;   // both lists of an offset through one resource
  constexpr int f32_mfma = 1;
  const bool f32_path = dtype == SPX_F32 && f32_mfma && C % 4 == 0 && K % 4 == 0 &&
                        static_cast<unsigned long long>(n_out) * K * 4ull < 0x7fff0000ull &&
                        static_cast<unsigned long long>(n_in) * C * 4ull < 0x7fff0000ull &&
                        static_cast<unsigned long long>(n_in) * 4ull < 0x7fff0000ull;
  if (f32_path || (mfma && wgrad_version >= 2 && small_offsets)) {
    Wgrad2Params q{};
    q.feat = feat;
    q.dout = dout;
    q.partial = static_cast<float *>(ws);
    q.native = pair_native;
    q.num = num_per_loc;
    q.plan2 = reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(plan) + plan1_bytes(n_in, kv));
    q.n_in = n_in;
    q.n_out = n_out;
    q.C = C;
    q.K = K;
    q.kv = kv;
    q.subm = subm;
    q.tiles_c = p.tiles_c;
    q.tiles_k = p.tiles_k;
    q.G = wgrad_groups(n_in, subm);
    q.xcd_order = wgrad_xcd_order();
    const dim3 grid(static_cast<unsigned>(q.G) * ntile);
    const size_t lds = 2 * 2 * kW2J * 128;    // two stages x two operand tiles
    const int sl = (C <= 16 && K <= 16) ? 2 : ((C <= 32 && K <= 32) ? 4 : 8);   // live 16-byte slots per row
    if (dtype == SPX_F32)
      hipLaunchKernelGGL(wgrad_f32_kernel, grid, dim3(kThreads), 2 * kW3J * kW3Stride * sizeof(float), s, q);
    else if (dtype == SPX_F16) {
      if (sl == 2) hipLaunchKernelGGL((wgrad_tr_kernel<false, 2>), grid, dim3(kThreads), lds, s, q);
      else if (sl == 4) hipLaunchKernelGGL((wgrad_tr_kernel<false, 4>), grid, dim3(kThreads), lds, s, q);
      else hipLaunchKernelGGL((wgrad_tr_kernel<false, 8>), grid, dim3(kThreads), lds, s, q);
    } else {
      if (sl == 2) hipLaunchKernelGGL((wgrad_tr_kernel<true, 2>), grid, dim3(kThreads), lds, s, q);
      else if (sl == 4) hipLaunchKernelGGL((wgrad_tr_kernel<true, 4>), grid, dim3(kThreads), lds, s, q);
      else hipLaunchKernelGGL((wgrad_tr_kernel<true, 8>), grid, dim3(kThreads), lds, s, q);
    }
    const dim3 rgrid2(reduce2_blocks(kv), ntile);   // block-stride over the work list
    if (dtype == SPX_F32)
      hipLaunchKernelGGL(wgrad_reduce2_kernel<float>, rgrid2, dim3(kRedThreads), 0, s, q,
                         static_cast<float *>(dw));
    else if (dtype == SPX_F16)
      hipLaunchKernelGGL(wgrad_reduce2_kernel<h16>, rgrid2, dim3(kRedThreads), 0, s, q,
                         static_cast<h16 *>(dw));
    else
      hipLaunchKernelGGL(wgrad_reduce2_kernel<b16>, rgrid2, dim3(kRedThreads), 0, s, q,
                         static_cast<b16 *>(dw));
    SPX_LAUNCH_CHECK();
    return 0;
  }
  {
    // fallback kernels (odd channel counts, tensors beyond 32-bit offsets): their item list is
    // built here, behind the partials (spx_wgrad_plan does not write it)
    int32_t *plan1 = reinterpret_cast<int32_t *>(static_cast<char *>(ws) + ws_bytes -
                                                 spx_wgrad_plan_bytes(n_in, kv));
    hipLaunchKernelGGL(wgrad_plan_kernel, dim3(1), dim3(kThreads), 0, s, num_per_loc, n_in, kv, subm,
                       wgrad_chunk(n_in), plan1);
    p.plan = plan1;
  }
  {
    // upper bound of work items is nchunks * kv * ntile; the kernels loop over the real count
    const long long bound = static_cast<long long>(p.nchunks) * kv * ntile;
    constexpr int max_grid = 1024;
    const dim3 grid(static_cast<unsigned>(bound < max_grid ? bound : max_grid));
    const size_t lds = 2 * kWT * kWJ * 2;
    if (mfma && dtype == SPX_F16)
      hipLaunchKernelGGL(wgrad_mfma_kernel<false>, grid, dim3(kThreads), lds, s, p);
    else if (mfma)
      hipLaunchKernelGGL(wgrad_mfma_kernel<true>, grid, dim3(kThreads), lds, s, p);
    else if (dtype == SPX_F32)
      hipLaunchKernelGGL(wgrad_generic_kernel<float>, grid, dim3(kThreads), 0, s, p);
    else if (dtype == SPX_F16)
      hipLaunchKernelGGL(wgrad_generic_kernel<h16>, grid, dim3(kThreads), 0, s, p);
    else if (dtype == SPX_BF16)
      hipLaunchKernelGGL(wgrad_generic_kernel<b16>, grid, dim3(kThreads), 0, s, p);
    else
      SPX_CHECK(false, "unsupported dtype %d", dtype);
    SPX_LAUNCH_CHECK();
  }
  const dim3 rgrid(div_up(ntile * kWT * kWT, kRedElems), kv);
  if (dtype == SPX_F32)
    hipLaunchKernelGGL(wgrad_reduce_kernel<float>, rgrid, dim3(kRedThreads), 0, s, p,
                       static_cast<float *>(dw));
  else if (dtype == SPX_F16)
    hipLaunchKernelGGL(wgrad_reduce_kernel<h16>, rgrid, dim3(kRedThreads), 0, s, p,
                       static_cast<h16 *>(dw));
  else if (dtype == SPX_BF16)
    hipLaunchKernelGGL(wgrad_reduce_kernel<b16>, rgrid, dim3(kRedThreads), 0, s, p,
                       static_cast<b16 *>(dw));
  else
    SPX_CHECK(false, "unsupported dtype %d", dtype);
  SPX_LAUNCH_CHECK();
  return 0;
}

int spx_igemm_bwd(const void *feat, const void *dout, const void *weight, void *din, void *dw,
                  const int32_t *pair, const uint32_t *mask, const int32_t *argsort, int tile_order,
                  const int32_t *pair_native, const int32_t *num_per_loc, const int32_t *plan,
                  int n_in, int n_out, int C, int K, int kv, int dtype, int subm, void *ws,
                  size_t ws_bytes, spx_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_in == 0 || n_out == 0) {                              // empty scene: din empty / zero, dW zero
    SPX_CHECK(dw && C > 0 && K > 0 && kv > 0, "null tensor pointer");
    SPX_HIP(hipMemsetAsync(dw, 0, static_cast<size_t>(K) * kv * C * elem_bytes(dtype), s));
    if (n_in > 0) {
      SPX_CHECK(din, "null tensor pointer");
      SPX_HIP(hipMemsetAsync(din, 0, static_cast<size_t>(n_in) * C * elem_bytes(dtype), s));
    }
    return 0;
  }
  SPX_CHECK(feat && dout && weight && din && dw && ws, "null tensor pointer");
  SPX_CHECK(pair_native && num_per_loc, "Native pair lists and counts are required");
  SPX_CHECK(pair || kv == 1, "pair table required");
  SPX_CHECK(ws_bytes >= spx_igemm_wgrad_ws_bytes(n_in, C, K, kv), "workspace too small");
  constexpr int fuse = 1;                  // (dgrad + wgrad in one launch: settled A/B)
  GemmParams p = dgrad_params(dout, weight, din, pair, mask, argsort, n_out, n_in, C, K, kv, subm);
  p.dense_hint = (tile_order & SPX_DENSE_HINT) ? 1 : 0;
  apply_rows_layout(p, tile_order & ~SPX_DENSE_HINT);
  const bool small_offsets = static_cast<unsigned long long>(n_out) * K * 2ull < 0x7fff0000ull &&
                             static_cast<unsigned long long>(n_in) * C * 2ull < 0x7fff0000ull &&
                             static_cast<unsigned long long>(n_in) * 4ull * (kv + 1) < 0x7fff0000ull;   // both lists of an offset through one resource
  const int es = dtype == SPX_F32 ? 4 : 2, lanes = 16 / es;
  const bool offsets_fit = static_cast<unsigned long long>(n_out) * K * es < 0x7fff0000ull &&
                           static_cast<unsigned long long>(n_in) * C * es < 0x7fff0000ull && small_offsets;
  constexpr int f32_mfma = 1;
  const bool fusable = fuse && (dtype == SPX_F16 || dtype == SPX_BF16 || (dtype == SPX_F32 && f32_mfma)) &&
                       C % lanes == 0 && K % lanes == 0 && mfma_ok(dtype, p.CIN, p.COUT, kv, mask) && kv <= 32 &&
                       p.COUT <= 128 && v4_ok(p, es, es) && offsets_fit && n_in > 0 && n_out > 0;
  if (!fusable) {
    if (spx_igemm_dgrad(dout, weight, din, pair, mask, argsort, tile_order, n_out, n_in, C, K, kv, dtype, subm,
                        nullptr, 0, stream))
      return -2;
    return spx_igemm_wgrad(feat, dout, dw, pair_native, num_per_loc, plan, n_in, n_out, C, K, kv, dtype,
                           subm, ws, ws_bytes, stream);
  }
  if (!plan) {  // caller did not cache a plan: build it behind the partials
    int32_t *own = reinterpret_cast<int32_t *>(static_cast<char *>(ws) + ws_bytes -
                                               spx_wgrad_plan_bytes(n_in, kv));
    if (spx_wgrad_plan(num_per_loc, n_in, kv, subm, own, stream)) return -2;
    plan = own;
  }
  Wgrad2Params q{};
  q.feat = feat;
  q.dout = dout;
  q.partial = static_cast<float *>(ws);
  q.native = pair_native;
  q.num = num_per_loc;
  q.plan2 = reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(plan) + plan1_bytes(n_in, kv));
  q.n_in = n_in;
  q.n_out = n_out;
  q.C = C;
  q.K = K;
  q.kv = kv;
  q.subm = subm;
  q.tiles_c = div_up(C, kWT);
  q.tiles_k = div_up(K, kWT);
  q.G = wgrad_groups(n_in, subm);
  q.xcd_order = wgrad_xcd_order();
  const int ntile = q.tiles_c * q.tiles_k;
  const int rc = dtype == SPX_F32 ? dispatch_bwd<3>(p, q, q.G * ntile, s)
                                  : (dtype == SPX_BF16 ? dispatch_bwd<1>(p, q, q.G * ntile, s)
                                                       : dispatch_bwd<0>(p, q, q.G * ntile, s));
  if (rc) return rc;
  const dim3 rgrid2(reduce2_blocks(kv), ntile);   // block-stride over the work list
  if (dtype == SPX_F32)
    hipLaunchKernelGGL(wgrad_reduce2_kernel<float>, rgrid2, dim3(kRedThreads), 0, s, q, static_cast<float *>(dw));
  else if (dtype == SPX_F16)
    hipLaunchKernelGGL(wgrad_reduce2_kernel<h16>, rgrid2, dim3(kRedThreads), 0, s, q, static_cast<h16 *>(dw));
  else
    hipLaunchKernelGGL(wgrad_reduce2_kernel<b16>, rgrid2, dim3(kRedThreads), 0, s, q, static_cast<b16 *>(dw));
  SPX_LAUNCH_CHECK();
  return 0;
}

int spx_bias_act_inplace(void *out, const void *bias, int n, int K, int dtype, int act,
                         float act_alpha, spx_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long long total = static_cast<long long>(n) * K;
  if (total == 0) return 0;
  const dim3 grid(static_cast<unsigned>((total + kThreads - 1) / kThreads));
  if (dtype == SPX_F32)
    hipLaunchKernelGGL(bias_act_kernel<float>, grid, dim3(kThreads), 0, s, static_cast<float *>(out),
                       static_cast<const float *>(bias), total, K, act, act_alpha);
  else if (dtype == SPX_F16)
    hipLaunchKernelGGL(bias_act_kernel<h16>, grid, dim3(kThreads), 0, s, static_cast<h16 *>(out),
                       static_cast<const h16 *>(bias), total, K, act, act_alpha);
  else if (dtype == SPX_BF16)
    hipLaunchKernelGGL(bias_act_kernel<b16>, grid, dim3(kThreads), 0, s, static_cast<b16 *>(out),
                       static_cast<const b16 *>(bias), total, K, act, act_alpha);
  else
    SPX_CHECK(false, "unsupported dtype %d", dtype);
  SPX_LAUNCH_CHECK();
  return 0;
}

#ifdef SPX_TIMELINE
// debug builds only: copies the timeline table (8192 workgroups x 8 stamps, uint64) to host memory
int spx_debug_timeline(unsigned long long *dst_h) {
  SPX_HIP(hipDeviceSynchronize());
  SPX_HIP(hipMemcpyFromSymbol(dst_h, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * kTlMaxWg * kTlSlots));
  return 0;
}
#endif

}  // extern "C"
