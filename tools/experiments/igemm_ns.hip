// EXPERIMENT (round 4), NOT part of the library build: measured slower than igemm_v4_kernel, kept as the record of what
// was built (profiles/r04_experiments.md section 3).  To run it again: copy next to igemm.hip, add it to csrc/build.sh,
// declare ns_ok / launch_ns in igemm_defs.h and call launch_ns from dispatch_gather_gemm (git history: the commit that
// moved this file here shows the three hooks); tools/experiments/ns_probe.py is its A/B driver.
//
// Dense-neighbourhood gather-GEMM: MFMA rows are VALID PAIRS.
//
// igemm_v4_kernel (igemm.hip) walks the offsets of a 128-row tile one step at a time; a step gathers all 128 rows
// of the tile whether or not they have a pair at that offset (a missing pair reads zeros through the buffer unit, but
// its load instruction and its MFMA rows are issued all the same), stages an 8 KB weight slice in LDS and ends in a
// barrier.  On real point clouds a row has ~6 of 27 pairs (the reference's LiDAR fixture: 6.28): 77 % of the MFMA
// rows are zeros and the kernel is bound by instruction issue and by the chain of dependent latencies of a step
// (DESIGN.md section 6), at 0.15 of the HBM roofline.  The reference skips absent offsets per mask-sorted tile
// (csrc/sparse/convops.py:1363-1446, argsort all.py:935-991); on a dense tile every offset is present in SOME row,
// so that does not help there either.  This kernel compacts instead:
//
//  * prologue: the tile's 27 pair-table columns are read once (coalesced runs) and every offset's valid
//    (input row, output row) pairs are compacted into an LDS list with wave ballots + prefix counts -- the rulebook
//    stays the reference's dense table in memory, the compaction lives for one tile;
//  * N-split: wave w owns output channels [16 w, 16 w + 16) of ALL rows of the tile.  Its weight fragments (the
//    MFMA B operand: 64 x 16 per offset) come straight from the L2 into registers -- no weight stage in LDS, no
//    barrier anywhere in the main loop, waves drift freely;
//  * a unit of work is a block of 16 pairs of one offset: gathered rows are the MFMA A operand (one 16-byte load
//    per lane and 64-byte half row, only for pairs that exist), two chained v_mfma_f32_16x16x32 give the block's
//    16 x 16 partial results, which are ADDED into an fp32 output tile in LDS (read - add - write).  Every
//    element of that tile is only ever touched by ONE wave -- the owner of its channel slice -- and a wave's LDS
//    operations execute in program order, so the sum order is fixed: atomics-free ownership, bit-reproducible;
//  * groups of two blocks are software-pipelined: the loads of group i + 1 (pair-list entries from LDS, gathered
//    rows, weight fragments) are in flight while group i multiplies;
//  * epilogue: one barrier, the fp32 tile is rounded once (bias / activation as in v4) and stored as whole rows.
//
// Per (tile, offset) a workgroup issues ~24 vector-memory instructions and 16 MFMAs where v4 issues 32 and 64, no
// LDS weight traffic (v4: 40 KB per step) and no barrier.  Shapes: 16-bit operands, 64 output channels, rows of
// at most 128 bytes (C <= 64), kernel volume <= 32, reduction index contiguous in the weights (forward: KRSC as it
// lies; dgrad: a [kv][C][K] copy, like spx_igemm_bwd_rows).  Everything else takes v4.
#include "igemm_defs.h"

namespace spx {
namespace {

constexpr int kNsTM = 128;            // rows per tile
constexpr int kNsAccStride = 68;      // floats per row of the fp32 output tile (+ 4: consecutive rows start 4 banks apart)
constexpr int kNsMaxKv = 32;
constexpr int kNsGroup = 2;           // 16-pair blocks per pipelined group

struct NsParams {
  const void *A, *B;
  void *out;
  const int32_t *pair;
  const void *bias;
  long long strideK, strideN;         // weight element (k, n, d) at k * strideK + n * strideN + d
  int n_src, n_dst, CIN, kv, b_reverse, act;
  float act_alpha;
  int nt_store;
};

constexpr int kNsAccRows = kNsTM + 1;  // + one row that swallows the adds of a block's empty pair slots
constexpr int kNsStageBytes = 32 * kRowBytes;     // one group of gathered rows: 32 pairs x 128 bytes
constexpr size_t ns_smem_bytes() {
  return static_cast<size_t>(kNsAccRows) * kNsAccStride * 4 + static_cast<size_t>(kNsMaxKv) * kNsTM * 4 + kNsMaxKv * 4 + 64 +
         2 * kNsStageBytes;
}

struct NsIt {          // a group: offset k, first block blk0 (k < 0: end)
  int k, blk0;
  uint32_t rest;       // offsets still to visit after k
};

template <bool BF16, int NKS>
__global__ void __launch_bounds__(kThreads)
igemm_ns_kernel(NsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *acc = reinterpret_cast<float *>(smem);                                        // [kNsAccRows][kNsAccStride]
  uint32_t *plist = reinterpret_cast<uint32_t *>(smem + kNsAccRows * kNsAccStride * 4);  // [kv][kNsTM]: in_row | out_local << 24
  int *cnt = reinterpret_cast<int *>(plist + kNsMaxKv * kNsTM);                        // [kNsMaxKv]
  char *stage = reinterpret_cast<char *>(cnt + kNsMaxKv + 16);                         // [2][32 rows][128 B], 16-byte pieces swizzled
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = (p.n_dst + kNsTM - 1) / kNsTM;
  const int tile = xcd_tile(blockIdx.x, ntiles);
  const int row0 = tile * kNsTM;
  const int lp = lane & 15, lg = lane >> 4;

  // ---- prologue: zero the output tile, compact the pair lists (wave w: offsets w, w + 4, ...) ----------------
  {
    const __amdgpu_buffer_rsrc_t rP = make_rsrc(p.pair, static_cast<uint32_t>(p.n_dst) * static_cast<uint32_t>(p.kv) * 4u);
    int v0[kNsMaxKv / 4], v1[kNsMaxKv / 4];
    const int t0 = row0 + lane, t1 = row0 + 64 + lane;
#pragma unroll
    for (int j = 0; j < kNsMaxKv / 4; ++j) {
      const int k = wave + 4 * j;
      const uint32_t base = static_cast<uint32_t>(k) * static_cast<uint32_t>(p.n_dst);
      v0[j] = (k < p.kv && t0 < p.n_dst) ? static_cast<int>(__builtin_amdgcn_raw_buffer_load_b32(rP, (base + t0) * 4u, 0, 0)) : -1;
      v1[j] = (k < p.kv && t1 < p.n_dst) ? static_cast<int>(__builtin_amdgcn_raw_buffer_load_b32(rP, (base + t1) * 4u, 0, 0)) : -1;
    }
    float4 *z = reinterpret_cast<float4 *>(acc);
    for (int i = tid; i < kNsAccRows * kNsAccStride / 4; i += kThreads) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < kNsMaxKv / 4; ++j) {
      const int k = wave + 4 * j;
      if (k < p.kv) {                                                // (uniform per wave)
        const unsigned long long b0 = __ballot(v0[j] >= 0), b1 = __ballot(v1[j] >= 0);
        const int n0 = __popcll(b0);
        const int pos0 = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(b0 >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(b0), 0u));
        const int pos1 = n0 + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(b1 >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(b1), 0u));
        if (v0[j] >= 0) plist[k * kNsTM + pos0] = static_cast<uint32_t>(v0[j]) | (static_cast<uint32_t>(lane) << 24);
        if (v1[j] >= 0) plist[k * kNsTM + pos1] = static_cast<uint32_t>(v1[j]) | (static_cast<uint32_t>(64 + lane) << 24);
        if (lane == 0) cnt[k] = n0 + __popcll(b1);
      }
    }
  }
  __syncthreads();

  // ---- main loop ----------------------------------------------------------------------------------------------
  const uint32_t rowB = static_cast<uint32_t>(p.CIN) * 2u;
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A, static_cast<uint32_t>(p.n_src) * rowB);
  const uint32_t w_bytes = 64u * static_cast<uint32_t>(p.kv) * rowB;         // [64 channels][kv][CIN] in either order
  const __amdgpu_buffer_rsrc_t rW = make_rsrc(p.B, w_bytes);
  // offsets that have pairs in this tile (every lane reads one count)
  const int my_cnt = lane < p.kv ? cnt[lane] : 0;
  const uint32_t kbits = static_cast<uint32_t>(__ballot(my_cnt > 0));
  // per-lane constant parts: byte inside a row for (ks, lg); the channel this lane's B fragment belongs to
  uint32_t koff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const uint32_t c = ks * 64u + lg * 16u;
    koff[ks] = c < rowB ? c : kOob;
  }
  const uint32_t wlane = static_cast<uint32_t>((16 * wave + lp) * p.strideN) * 2u;

  auto first = [&]() __attribute__((always_inline)) {
    NsIt it;
    it.blk0 = 0;
    it.k = kbits ? __builtin_ctz(kbits) : -1;
    it.rest = kbits ? (kbits & (kbits - 1)) : 0u;
    return it;
  };
  auto next = [&](NsIt it) __attribute__((always_inline)) {
    if (it.k < 0) return it;
    const int nb = (cnt[it.k] + 15) >> 4;
    if (it.blk0 + kNsGroup < nb) {
      it.blk0 += kNsGroup;
      return it;
    }
    it.blk0 = 0;
    it.k = it.rest ? __builtin_ctz(it.rest) : -1;
    it.rest = it.rest ? (it.rest & (it.rest - 1)) : 0u;
    return it;
  };

  // A group = up to 32 pairs of one offset.  Its gathered rows are fetched ONCE per workgroup -- thread t brings the
  // 16-byte piece t & 7 of pair t >> 3 -- and staged in LDS (two stages, one barrier per group, pieces XOR-swizzled so
  // that both the 1 KB-per-wave writes and the MFMA fragment reads are conflict-free); every wave multiplies all 32
  // pairs with ITS 16 output channels, whose weight fragments it holds in registers.
  const int gj = tid >> 3, gq = tid & 7;                 // gather role: pair slot of the group, piece of the row
  const uint32_t gpiece = static_cast<uint32_t>(gq) * 16u < rowB ? static_cast<uint32_t>(gq) * 16u : kOob;
  const int st_w = gj * kRowBytes + ((gq ^ ((gj >> 1) & 7)) << 4);     // where this thread's piece lands in a stage
  u32x4 apiece[4], breg[4][2];     // rings of four: rows and weights are requested FOUR groups ahead of their MFMAs
  auto load_a = [&](const NsIt &it, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const int k = it.k < 0 ? 0 : it.k;
    const int n = it.k < 0 ? 0 : cnt[k];
    const int j = it.blk0 * 16 + gj;
    const uint32_t e = plist[k * kNsTM + (j & (kNsTM - 1))];
    const uint32_t rbase = j < n ? (e & 0xffffffu) * rowB : kOob;
    apiece[S] = __builtin_amdgcn_raw_buffer_load_b128(rA, min(rbase + gpiece, kOob), 0, 0);
  };
  auto load_w = [&](const NsIt &it, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const int k = it.k < 0 ? 0 : it.k;
    const int kb = p.b_reverse ? p.kv - 1 - k : k;
    const uint32_t wbase = static_cast<uint32_t>(kb * p.strideK) * 2u;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      breg[S][ks] = __builtin_amdgcn_raw_buffer_load_b128(rW, it.k < 0 ? kOob : min(wlane + koff[ks], kOob), wbase, 0);
  };
  auto stage_a = [&](auto SET) __attribute__((always_inline)) {        // apiece[S] -> LDS stage S & 1
    constexpr int S = decltype(SET)::value;
    *reinterpret_cast<u32x4 *>(stage + (S & 1) * kNsStageBytes + st_w) = apiece[S];
  };
  auto compute_group = [&](const NsIt &it, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    if (it.k < 0) return;
    const int n = cnt[it.k];
    const char *cur = stage + (S & 1) * kNsStageBytes;
#pragma unroll
    for (int b = 0; b < kNsGroup; ++b) {
      const int j0 = (it.blk0 + b) * 16;
      if (j0 < n) {                                                  // (uniform)
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        const int row = b * 16 + lp;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const uint4 fa = *reinterpret_cast<const uint4 *>(cur + row * kRowBytes + (((ks * 4 + lg) ^ ((row >> 1) & 7)) << 4));
          d = mfma16<BF16>(fa, __builtin_bit_cast(uint4, breg[S][ks]), d);
        }
        // D[pair lg * 4 + i][channel lp]: add into the output rows of those four pairs (an empty slot of the list's
        // last block adds into the spare row: no branch).  Plain read - add - write: this wave is the only writer of
        // its channel slice and a wave's LDS operations execute in program order.  (ds_add_f32 measured ~300 clocks
        // per instruction here: 295 us for the fixture forward.)
        const uint32_t *e4 = plist + it.k * kNsTM + j0 + lg * 4;
        float *dst[4];
        float curv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t orow = j0 + lg * 4 + i < n ? (e4[i] >> 24) : static_cast<uint32_t>(kNsTM);
          dst[i] = acc + orow * kNsAccStride + 16 * wave + lp;
          curv[i] = *dst[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *dst[i] = curv[i] + d[i];
      }
    }
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  using Set2 = std::integral_constant<int, 2>;
  using Set3 = std::integral_constant<int, 3>;
  // group i lives in ring slot i & 3 and LDS stage i & 1: its rows and weight fragments are requested four groups
  // ahead, its rows are written to their stage one group ahead (published by that group's barrier)
  NsIt g0 = first(), g1 = next(g0), g2 = next(g1), g3 = next(g2), g4 = next(g3);
  load_a(g0, Set0{});
  load_w(g0, Set0{});
  load_a(g1, Set1{});
  load_w(g1, Set1{});
  load_a(g2, Set2{});
  load_w(g2, Set2{});
  load_a(g3, Set3{});
  load_w(g3, Set3{});
  stage_a(Set0{});
  __syncthreads();
  auto one = [&](auto SLOT, auto NEXT) __attribute__((always_inline)) {
    stage_a(NEXT);                       // rows of group i + 1 -> the other stage
    compute_group(g0, SLOT);
    load_a(g4, SLOT);                    // group i + 4 takes over slot i & 3
    load_w(g4, SLOT);
    __syncthreads();
    g0 = g1;
    g1 = g2;
    g2 = g3;
    g3 = g4;
    g4 = next(g4);
  };
  while (g0.k >= 0) {
    one(Set0{}, Set1{});
    one(Set1{}, Set2{});
    one(Set2{}, Set3{});
    one(Set3{}, Set0{});
  }
  __syncthreads();

  // ---- epilogue: two threads per row, 32 channels each; rounded once ----------------------------------------------
  {
    const int r = tid >> 1, h = tid & 1;
    const int t = row0 + r;
    const float *src = acc + r * kNsAccStride + h * 32;
    const bool plain = p.bias == nullptr && p.act == SPX_ACT_NONE;
    uint32_t d[16];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(src + 4 * q);
      float x[4] = {v.x, v.y, v.z, v.w};
      if (!plain) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float bv = p.bias ? to_float<BF16>(static_cast<const uint16_t *>(p.bias)[h * 32 + 4 * q + e]) : 0.f;
          x[e] = apply_act(x[e] + bv, p.act, p.act_alpha);
        }
      }
      d[2 * q] = pack2<BF16>(x[0], x[1]);
      d[2 * q + 1] = pack2<BF16>(x[2], x[3]);
    }
    const __amdgpu_buffer_rsrc_t rO = make_rsrc(p.out, static_cast<uint32_t>(p.n_dst) * 128u);
    const uint32_t vo = t < p.n_dst ? static_cast<uint32_t>(t) * 128u + h * 64u : kOob;
    if (p.nt_store) store_dwords<16, 2>(d, rO, vo);
    else store_dwords<16>(d, rO, vo);
  }
}

}  // namespace

bool ns_ok(const GemmParams &p, int dtype) {
  return (dtype == SPX_F16 || dtype == SPX_BF16) && p.COUT == 64 && p.CIN % 8 == 0 && p.CIN <= 64 && p.kv <= kNsMaxKv &&
         p.strideD == 1 && p.pair && !p.argsort && !p.acc && p.acc_mode == 0 && p.n_src < (1 << 24) &&
         static_cast<unsigned long long>(p.n_src) * p.CIN * 2ull < 0x7fff0000ull &&
         static_cast<unsigned long long>(p.n_dst) * p.kv * 4ull < 0x7fff0000ull &&
         static_cast<unsigned long long>(p.n_dst) * 128ull < 0x7fff0000ull;
}

int launch_ns(const GemmParams &p, int dtype, hipStream_t s) {
  NsParams q{};
  q.A = p.A;
  q.B = p.B;
  q.out = p.out;
  q.pair = p.pair;
  q.bias = p.bias;
  q.strideK = p.strideK;
  q.strideN = p.strideN;
  q.n_src = p.n_src;
  q.n_dst = p.n_dst;
  q.CIN = p.CIN;
  q.kv = p.kv;
  q.b_reverse = p.b_reverse;
  q.act = p.act;
  q.act_alpha = p.act_alpha;
  q.nt_store = (p.dbg & 0x400) ? 0 : 1;
  const int ntiles = div_up(p.n_dst, kNsTM);
  const bool half = p.CIN * 2 <= 64;
#define SPX_LAUNCH_NS(BF, NKSV)                                                                                  \
  {                                                                                                              \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&igemm_ns_kernel<BF, NKSV>), \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize,                \
                                                       static_cast<int>(ns_smem_bytes()));                        \
    (void)attr;                                                                                                  \
    hipLaunchKernelGGL((igemm_ns_kernel<BF, NKSV>), dim3(ntiles), dim3(kThreads), ns_smem_bytes(), s, q);          \
  }
  if (dtype == SPX_BF16) {
    if (half) SPX_LAUNCH_NS(true, 1) else SPX_LAUNCH_NS(true, 2)
  } else {
    if (half) SPX_LAUNCH_NS(false, 1) else SPX_LAUNCH_NS(false, 2)
  }
#undef SPX_LAUNCH_NS
  SPX_LAUNCH_CHECK();
  return 0;
}

}  // namespace spx
