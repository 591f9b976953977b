// Gather-GEMM for SPARSE neighbourhoods (gfx950): one autonomous wave per 32 destination rows.
//
// On a uniform-random scene (BASELINE config 2: 1.03 pairs per voxel) a 128-row tile of igemm_v4_kernel
// holds the identity pair of every row plus ~4 other pairs in ~4 different offsets -- and walks all
// of them as workgroup-wide steps: 8 KB of weights through LDS and a barrier per step, every step
// one more dependent trip to memory (mask words -> pair words -> rows, weights one step ahead).  The
// launch is a chain of 5-6 round trips; its bytes would take a third of the time.
//
// Here a wave owns 32 rows and shares nothing:
//   * no LDS staging of weights, no barrier: the wave loads the MFMA fragments of a weight slice
//     straight from memory (8 x 16 bytes per lane; the 221 KB weight tensor lives in the L2) and only
//     for offsets ITS rows use (~1 besides the identity, not the tile's ~4);
//   * the wave's slice of the pair table ([kv][32] words, 3.4 KB) arrives by LDS-DMA together with
//     the mask words, the identity rows and the identity weights: ONE trip to memory; the second trip
//     fetches the rows and weight fragments of the (few) other offsets, two offsets in flight;
//   * results are stored straight from the accumulator registers (each lane holds consecutive
//     channels of its row, as in v4).
// Per-row arithmetic (identity offset first, ascending offsets after, v_mfma_f32_16x16x32 with the
// same lane <-> reduction-element assignment) is v4's: results are bit-identical.
//
// dgrad runs through the same kernel on the TRANSPOSED weights [C, kv, K] (spx_weight_transpose /
// the copy a training forward leaves behind): with the reduction index contiguous no transposing
// LDS pass is needed.  Dense scenes (every wave walks all 27 offsets, 8 KB of weights each) belong to
// igemm_v4_kernel; the dispatcher picks by the rulebook's pairs-per-row (spx_igemm_* `sparse` hint).
// Stands in for the reference's tuned choice among implicit-GEMM tile shapes per problem
// (spconv/csrc/sparse/convops.py:1150 tune_and_cache, :1311 get_tuned_algo).
#include "igemm_defs.h"

namespace spx {
namespace {

typedef __attribute__((address_space(3))) void lds_void_sp_t;
__device__ __forceinline__ void sp_dma16(__amdgpu_buffer_rsrc_t r, char *dst, uint32_t vo) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_sp_t *)dst, 16, vo, 0, 0, 0);
}

constexpr int kSpRows = 32;          // destination rows per wave (2 x 16-row MFMA blocks)

// per-wave timeline of the debug build (csrc/build_debug.sh, tools/timeline.py --kernel sp)
#ifdef SPX_TIMELINE
constexpr int kSpTlSlots = 8, kSpTlMax = 8192;
__device__ unsigned long long g_timeline_sp[kSpTlMax * kSpTlSlots];
#define SP_STAMP(i)                                                                          \
  do {                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < kSpTlMax)                                           \
      g_timeline_sp[blockIdx.x * kSpTlSlots + (i)] = __builtin_amdgcn_s_memtime();           \
  } while (0)
#else
#define SP_STAMP(i) do {} while (0)
#endif

}  // namespace

template <int COUT, int DT, int NKS>
__global__ void __launch_bounds__(64, 3)      // <= 168 registers: 12 waves per CU
igemm_sp_kernel(const void *argA, const void *argB, const uint32_t *arg_mask,
                const int32_t *arg_argsort, const int32_t *arg_pair, int n_dst, int n_src, int CIN,
                int kv, int identity_k, int flags, GemmRest rest) {
  constexpr bool BF16 = DT == 1;
  constexpr int ES = 2, MB = 2;
  constexpr int NB = COUT / 16, CPL = NB * 4;
  constexpr int AK = NKS < 2 ? 2 : NKS;                 // (register arrays stay at >= 2 elements)
  __shared__ __attribute__((aligned(16))) char sPair[32 * kSpRows * 4];      // [kv <= 32][32] int32

  SP_STAMP(0);
  const int lane = threadIdx.x & 63, lrow = lane & 15, lgrp = lane >> 4;
  const int ntiles = (n_dst + kSpRows - 1) / kSpRows;
  const int tile = xcd_tile(static_cast<int>(blockIdx.x), ntiles);
  const bool b_reverse = flags & 1;
  const bool spec = identity_k >= 0;
  const uint32_t idbit = spec ? (1u << identity_k) : 0u;
  const uint32_t kvmask = kv < 32 ? (1u << kv) - 1u : 0xffffffffu;
  const uint32_t rowB = static_cast<uint32_t>(CIN) * ES;
  const uint32_t a_bytes = static_cast<uint32_t>(n_src) * rowB;
  const uint32_t w_bytes = static_cast<uint32_t>(rest.COUT) * kv * rowB;
  const __amdgpu_buffer_rsrc_t rP = make_rsrc(arg_pair, static_cast<uint32_t>(kv) * n_dst * 4u);
  const __amdgpu_buffer_rsrc_t rM = make_rsrc(arg_mask, arg_mask ? static_cast<uint32_t>(n_dst) * 4u : 0u);

  // ---- trip 1, part 1: the wave's slice of the pair table, by LDS-DMA (oldest requests of the wave:
  // any later counted wait for a register load retires them as well)
  {
    const int kvp8 = (kv + 7) >> 3;                     // instructions: 8 columns of 128 bytes each
    for (int i = 0; i < kvp8; ++i) {
      const int q = i * 64 + lane, c = q >> 3, pc = q & 7;
      const uint32_t vo = c < kv ? (static_cast<uint32_t>(c) * static_cast<uint32_t>(n_dst) +
                                    static_cast<uint32_t>(tile) * kSpRows) * 4u + pc * 16u
                                 : kOob;
      sp_dma16(rP, sPair + i * 1024, vo);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const uint32_t tm = static_cast<uint32_t>(tile) * kSpRows + (lane & 31);
  const uint32_t mraw = __builtin_amdgcn_raw_buffer_load_b32(
      rM, (lane < 32 && tm < static_cast<uint32_t>(n_dst)) ? tm * 4u : kOob, 0, 0);

  uint32_t aoff[AK];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const uint32_t c = ks * 64 + lgrp * 16;
    aoff[ks] = c < rowB ? c : kOob;
  }
  // weight fragment (nb, ks) of this lane: 16 bytes of row chan(lrow, nb) of the slice, reduction
  // bytes [ks * 64 + lgrp * 16, +16).  MFMA row (g = i >> 2, e = i & 3) of channel block nb carries
  // channel g * CPL + nb * 4 + e (each lane ends up with CPL consecutive channels of its row).
  uint32_t woff[NB][AK];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int ch = (lrow >> 2) * CPL + nb * 4 + (lrow & 3);
      woff[nb][ks] = aoff[ks] == kOob ? kOob
                                      : static_cast<uint32_t>(ch) * static_cast<uint32_t>(rest.strideN) * ES + aoff[ks];
    }

  u32x4 areg[2][MB][AK];
  u32x4 wreg[2][NB][AK];
  int kq[2] = {-1, -1};
  f32x4 acc[NB][MB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = f32x4{0, 0, 0, 0};

  uint32_t trow[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) trow[mb] = static_cast<uint32_t>(tile) * kSpRows + mb * 16 + lrow;

  // rows `idx` (or the wave's own rows) and the weight slice of offset k into register set S
  auto load_set = [&](auto SET, int k, bool own_rows) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    kq[S] = k;
    const int kk = k < 0 ? 0 : k;
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(argA, k >= 0 ? a_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(argB, k >= 0 ? w_bytes : 0u);
    const int32_t *pb = reinterpret_cast<const int32_t *>(sPair) + kk * kSpRows + lrow;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      uint32_t idx = own_rows ? (trow[mb] < static_cast<uint32_t>(n_dst) ? trow[mb] : 0xffffffffu)
                              : static_cast<uint32_t>(pb[mb * 16]);
      const uint32_t rbase = idx * rowB;                                       // -1 -> >= kOob
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const uint32_t vo = min(rbase + aoff[ks], kOob) | (aoff[ks] & kOob);
        areg[S][mb][ks] = __builtin_amdgcn_raw_buffer_load_b128(rA, vo, 0, 0);
      }
    }
    const int kb = b_reverse ? kv - 1 - kk : kk;
    const uint32_t so = static_cast<uint32_t>(kb) * static_cast<uint32_t>(rest.strideK) * ES;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        wreg[S][nb][ks] = __builtin_amdgcn_raw_buffer_load_b128(rW, woff[nb][ks], so, 0);
  };
  auto compute = [&](auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    if (kq[S] >= 0) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            acc[nb][mb] = mfma16<BF16>(__builtin_bit_cast(uint4, wreg[S][nb][ks]),
                                       __builtin_bit_cast(uint4, areg[S][mb][ks]), acc[nb][mb]);
    }
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;

  // ---- trip 1, part 2: the identity step (its rows are the wave's own: no pair words needed)
  __builtin_amdgcn_sched_barrier(0);
  load_set(Set0{}, spec ? identity_k : -1, true);
  __builtin_amdgcn_sched_barrier(0);
  SP_STAMP(1);   // first trip issued

  // offsets some row of the wave uses (bit k of a mask word <=> pair[k][row] >= 0)
  uint32_t wm = mraw;
  wm |= __shfl_xor(wm, 1, 64);
  wm |= __shfl_xor(wm, 2, 64);
  wm |= __shfl_xor(wm, 4, 64);
  wm |= __shfl_xor(wm, 8, 64);
  wm |= __shfl_xor(wm, 16, 64);
  uint32_t bits = __builtin_amdgcn_readfirstlane(wm);
  if (!arg_mask) bits = 0xffffffffu;
  bits &= kvmask & ~idbit;
  SP_STAMP(2);   // mask words arrived
  auto next_k = [&]() __attribute__((always_inline)) {
    const int k = bits ? __builtin_ctz(bits) : -1;
    bits &= bits - 1;
    return k;
  };
  // ---- trip 2: rows + weight fragments of the next offsets, two in flight
  if (!spec) load_set(Set0{}, next_k(), false);
  __builtin_amdgcn_sched_barrier(0);
  load_set(Set1{}, next_k(), false);
  __builtin_amdgcn_sched_barrier(0);
  SP_STAMP(3);   // second trip issued
  while (kq[0] >= 0) {
    compute(Set0{});
    __builtin_amdgcn_sched_barrier(0);
    load_set(Set0{}, next_k(), false);
    __builtin_amdgcn_sched_barrier(0);
    compute(Set1{});
    __builtin_amdgcn_sched_barrier(0);
    load_set(Set1{}, next_k(), false);
    __builtin_amdgcn_sched_barrier(0);
  }

  SP_STAMP(4);   // MFMAs done
  // ---- epilogue: CPL consecutive channels per lane, straight from the accumulators
  const bool plain = rest.bias == nullptr && rest.act == SPX_ACT_NONE;
  const __amdgpu_buffer_rsrc_t rO = make_rsrc(rest.out, static_cast<uint32_t>(n_dst) * (COUT * ES));
  float bv[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) bv[q] = 0.f;
  if (rest.bias) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) bv[q] = to_float<BF16>(static_cast<const uint16_t *>(rest.bias)[lgrp * CPL + q]);
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    uint32_t d[CPL / 2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v0 = acc[nb][mb][2 * h], v1 = acc[nb][mb][2 * h + 1];
        if (!plain) {
          v0 = apply_act(v0 + bv[nb * 4 + 2 * h], rest.act, rest.act_alpha);
          v1 = apply_act(v1 + bv[nb * 4 + 2 * h + 1], rest.act, rest.act_alpha);
        }
        d[nb * 2 + h] = pack2<BF16>(v0, v1);
      }
    }
    const uint32_t vo = trow[mb] < static_cast<uint32_t>(n_dst) ? trow[mb] * (COUT * ES) + lgrp * (CPL * ES) : kOob;
    if (rest.dbg & 0x400) store_dwords<CPL / 2>(d, rO, vo);
    else store_dwords<CPL / 2, 2>(d, rO, vo);
  }
  SP_STAMP(6);   // stores issued
#ifdef SPX_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SP_STAMP(7);   // stores retired
#endif
}

namespace {

template <int COUT, int DT>
int launch_sp_shape(const GemmParams &p, const GemmRest &r, hipStream_t s) {
  const int ntiles = div_up(p.n_dst, kSpRows);
  const int flags = p.b_reverse & 1;
  if (p.CIN * 2 <= 64)
    hipLaunchKernelGGL((igemm_sp_kernel<COUT, DT, 1>), dim3(ntiles), dim3(64), 0, s, p.A, p.B, p.mask, p.argsort,
                       p.pair, p.n_dst, p.n_src, p.CIN, p.kv, p.identity_k, flags, r);
  else
    hipLaunchKernelGGL((igemm_sp_kernel<COUT, DT, 2>), dim3(ntiles), dim3(64), 0, s, p.A, p.B, p.mask, p.argsort,
                       p.pair, p.n_dst, p.n_src, p.CIN, p.kv, p.identity_k, flags, r);
  SPX_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// shapes the sparse-neighbourhood kernel covers: 16-bit operands, reduction contiguous (forward, or
// dgrad on transposed weights), gathered rows of at most 128 bytes, up to 64 output channels
bool sp_ok(const GemmParams &p, int dtype) {
  if (dtype != SPX_F16 && dtype != SPX_BF16) return false;
  if (!p.pair || p.argsort || p.tile_order || p.kv > 32 || p.kbase != 0 || p.mask_words > 1) return false;
  if (p.acc || p.acc_mode || p.strideD != 1) return false;
  if (p.COUT != 16 && p.COUT != 32 && p.COUT != 64) return false;
  if (p.CIN % 8 != 0 || p.CIN * 2 > 128) return false;
  const unsigned long long tbytes = static_cast<unsigned long long>(p.kv) * p.n_dst * 4ull;
  const unsigned long long abytes = static_cast<unsigned long long>(p.n_src) * p.CIN * 2ull;
  const unsigned long long obytes = static_cast<unsigned long long>(p.n_dst) * p.COUT * 2ull;
  const unsigned long long wbytes = static_cast<unsigned long long>(p.COUT) * p.kv * p.CIN * 2ull;
  return tbytes < 0x7fff0000ull && abytes < 0x7fff0000ull && obytes < 0x7fff0000ull && wbytes < 0x7fff0000ull;
}

int launch_sp(const GemmParams &p, const GemmRest &r, int dtype, hipStream_t s) {
  const bool bf = dtype == SPX_BF16;
  switch (p.COUT) {
    case 16: return bf ? launch_sp_shape<16, 1>(p, r, s) : launch_sp_shape<16, 0>(p, r, s);
    case 32: return bf ? launch_sp_shape<32, 1>(p, r, s) : launch_sp_shape<32, 0>(p, r, s);
    case 64: return bf ? launch_sp_shape<64, 1>(p, r, s) : launch_sp_shape<64, 0>(p, r, s);
  }
  return -1;
}

}  // namespace spx

#ifdef SPX_TIMELINE
extern "C" int spx_debug_timeline_sp(unsigned long long *dst_h) {
  SPX_HIP(hipDeviceSynchronize());
  SPX_HIP(hipMemcpyFromSymbol(dst_h, HIP_SYMBOL(spx::g_timeline_sp),
                              sizeof(unsigned long long) * spx::kSpTlMax * spx::kSpTlSlots));
  return 0;
}
#endif
