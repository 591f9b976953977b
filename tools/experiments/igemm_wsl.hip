// Weight-resident gather-GEMM for the NARROW layers of a backbone (gfx950): forward of a sparse convolution with 16 or
// 32 input channels (rows of 32 / 64 bytes) on dense neighbourhoods -- the first two levels of a SECOND-style voxel
// backbone (BASELINE config 4: 400 k voxels x 16 channels, 313 k x 32) and the strided layers between them.
//
// Same contract and the same arithmetic as igemm_v4_kernel (igemm_v4.h): output-stationary implicit GEMM over the
// [kv, n_dst] pair table, every output row written exactly once, v_mfma_f32_16x16x32_{f16,bf16} with the gathered rows
// fed straight from VGPRs (a missing pair is an out-of-range buffer offset -> zeros), per-row accumulation order
// "identity offset first, then ascending offsets", the same fragment layouts -- results are BIT-IDENTICAL to v4's.
// What differs is everything around the MFMAs, which is what bounds v4 at these widths (a step of a 128-row tile is
// two row loads, one or two MFMAs per wave -- and a weight slice through global -> registers -> LDS plus a workgroup
// barrier: ~1.2 us per step whatever it contains, 27 steps per tile on a dense scene):
//
//  * ALL kv weight slices are resident: kv x C_out x C_in x 2 bytes = 13.5 KB (16 -> 16), 27 KB (16 -> 32), 54 KB
//    (32 -> 32) or 108 KB (32 -> 64) of LDS, staged once per workgroup; after the one barrier behind the staging there
//    is no barrier, no weight load and no LDS write in the kernel;
//  * every WAVE is on its own: it takes 32 rows, walks the offsets THOSE rows have (the set bits of its own mask OR)
//    with its own register pipeline (gathered rows two steps ahead, pair words three), stores its rows and takes the
//    next 32 rows of its workgroup's share -- a workgroup is a set of 16 independent waves that share one weight image;
//  * the LDS image keeps every ds_read_b128 lane group on 16 distinct 16-byte bank slots (rows of 32 / 64 bytes: a
//    row permutation or an XOR of the slot with row bits, checked by enumeration -- tools/experiments/wsl_swizzle.py).
//
// Limits (the dispatcher keeps v4 otherwise): forward only (rows of the weight tensor contiguous), 16-bit operands,
// (C_in, C_out) in {(16,16), (16,32), (32,32), (32,64)}, kernel volume <= 32, 32-bit buffer offsets.  Reference kernels
// this stands in for: the mask-skipping implicit-GEMM forward of spconv/csrc/sparse/convops.py:1363-1446.
#include "igemm_defs.h"

namespace spx {
namespace {

struct WslArgs {
  const void *A;            // [n_src, CIN] gathered operand
  const void *B;            // weights, element (k, n, c) at k*strideK + n*strideN + c
  void *out;                // [n_dst, COUT]
  const int32_t *pair;      // [kv, n_dst]
  const uint32_t *mask;     // [n_dst] or null
  const int32_t *argsort;   // [n_dst] (tables in tile order) or null
  const void *bias;
  long long strideK, strideN;
  int n_dst, n_src, kv, identity_k, b_reverse, act, ngroups;
  float act_alpha;
};

struct WslStep {
  int j;           // position in the sequence (identity offset first, then ascending), -1 = end
  uint32_t rest;   // positions after j
};
__device__ __forceinline__ WslStep wsl_first(uint32_t bits) {
  WslStep s;
  s.j = bits ? __builtin_ctz(bits) : -1;
  s.rest = bits ? (bits & (bits - 1)) : 0u;
  return s;
}
__device__ __forceinline__ WslStep wsl_next(WslStep s) { return wsl_first(s.rest); }

// LDS image of one slice: weight row n (output channel) at row position wsl_row(n), its 16-byte slot s at physical
// slot s ^ wsl_swz(n).  Fragment reads take rows (lrow >> 2) * CPL + nb * 4 + (lrow & 3): conflict-free for the four
// shapes (enumerated over the ds_read_b128 lane groups of MI355X_MICROARCH.md, LDS).
template <int CIN, int COUT>
__device__ __forceinline__ int wsl_row(int n) {
  if constexpr (CIN == 16 && COUT == 32) return (n & 3) | ((n >> 3) << 2) | (((n >> 2) & 1) << 4);
  return n;
}
template <int CIN, int COUT>
__device__ __forceinline__ int wsl_swz(int n) {
  if constexpr (CIN == 32 && COUT == 32) return (0x78 >> (((n >> 3) & 3) * 2)) & 3;      // (0, 2, 3, 1)[(n >> 3) & 3]
  if constexpr (CIN == 32 && COUT == 64) return (n >> 4) & 2;                             // (0, 0, 2, 2)[(n >> 4) & 3]
  return 0;
}

template <int CIN, int COUT, bool BF16, int NW>
__global__ void __launch_bounds__(NW * 64, (COUT <= 32 ? 8 : 4))
igemm_wsl_kernel(WslArgs p) {
  constexpr int MB = 2, D = 2, NB = COUT / 16, CPL = NB * 4;
  constexpr int RB = CIN * 2, SPR = RB / 16;                // bytes / 16-byte slots per row (of both operands)
  constexpr int SLICE = COUT * RB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int kv = p.kv;
  const uint32_t kvbits = kv >= 32 ? 0xffffffffu : ((1u << kv) - 1u);
  const bool spec = p.identity_k >= 0;    // SubM: the identity offset exists for every row
  const int ik = spec ? p.identity_k : 0;
  auto koff = [&](int j) __attribute__((always_inline)) { return !spec ? j : (j == 0 ? ik : (j <= ik ? j - 1 : j)); };

  // ---- every slice -> LDS, once (slice of offset k at k * SLICE) --------------------------------------------------
  {
    const uint32_t w_bytes = static_cast<uint32_t>(COUT) * kv * RB;
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(p.B, w_bytes);
    const int npieces = kv * COUT * SPR;
    for (int q0 = tid; q0 < npieces; q0 += 4 * NW * 64) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + u * NW * 64;
        const int k = q / (COUT * SPR), n = (q / SPR) % COUT, s = q % SPR;
        const int kb = p.b_reverse ? kv - 1 - k : k;
        const uint32_t so = (static_cast<uint32_t>(kb) * static_cast<uint32_t>(p.strideK) +
                             static_cast<uint32_t>(n) * static_cast<uint32_t>(p.strideN)) * 2u + s * 16u;
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rW, q < npieces ? so : kOob, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + u * NW * 64;
        if (q < npieces) {
          const int k = q / (COUT * SPR), n = (q / SPR) % COUT, s = q % SPR;
          *reinterpret_cast<u32x4 *>(smem + k * SLICE + wsl_row<CIN, COUT>(n) * RB + ((s ^ wsl_swz<CIN, COUT>(n)) << 4)) = v[u];
        }
      }
    }
  }
  __syncthreads();

  // fragment address of this lane inside a slice (lanes beyond the row's slots -- 32-byte rows: lane groups 2, 3 --
  // re-read their partner's address and drop the value: their reduction elements do not exist)
  int foff[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = (lrow >> 2) * CPL + nb * 4 + (lrow & 3);
    foff[nb] = wsl_row<CIN, COUT>(n) * RB + ((((lgrp & (SPR - 1))) ^ wsl_swz<CIN, COUT>(n)) << 4);
  }
  const bool fvalid = SPR >= 4 || lgrp < SPR;

  const uint32_t tbl_bytes = static_cast<uint32_t>(p.n_dst) * 4u;
  const __amdgpu_buffer_rsrc_t rO = make_rsrc(p.argsort, p.argsort ? tbl_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rM = make_rsrc(p.mask, p.mask ? tbl_bytes : 0u);
  const uint32_t a_bytes = static_cast<uint32_t>(p.n_src) * RB;
  const uint32_t aoff = fvalid ? static_cast<uint32_t>(lgrp * 16) : kOob;
  const bool plain = p.bias == nullptr && p.act == SPX_ACT_NONE;
  const __amdgpu_buffer_rsrc_t rOut = make_rsrc(p.out, static_cast<uint32_t>(p.n_dst) * (COUT * 2));

  // 32-row groups of this wave: workgroup b (XCD b % 8) owns a contiguous share of the groups, dealt to its waves
  // round-robin
  const int nwg = static_cast<int>(gridDim.x);
  const int share = xcd_tile(static_cast<int>(blockIdx.x), nwg);
  const long long g0 = static_cast<long long>(p.ngroups) * share / nwg, g1 = static_cast<long long>(p.ngroups) * (share + 1) / nwg;
  for (int grp = static_cast<int>(g0) + wave; grp < static_cast<int>(g1); grp += NW) {
    // ---- rows of this lane ------------------------------------------------------------------------------------------
    int pos[MB], glist[MB], grow[MB];
    uint32_t goff[MB], mraw[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int t = grp * 32 + mb * 16 + lrow;
      pos[mb] = t < p.n_dst ? t : -1;
      goff[mb] = pos[mb] < 0 ? kOob : static_cast<uint32_t>(pos[mb]) * 4u;
      glist[mb] = static_cast<int>(__builtin_amdgcn_raw_buffer_load_b32(rO, goff[mb], 0, 0));
      mraw[mb] = __builtin_amdgcn_raw_buffer_load_b32(rM, goff[mb], 0, 0);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) grow[mb] = pos[mb] < 0 ? -1 : (p.argsort ? glist[mb] : pos[mb]);
    uint32_t wm = 0;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) wm |= mraw[mb];         // rows past the end read 0
    if (!p.mask) wm = 0xffffffffu;
    wm |= __shfl_xor(wm, 1, 64);
    wm |= __shfl_xor(wm, 2, 64);
    wm |= __shfl_xor(wm, 4, 64);
    wm |= __shfl_xor(wm, 8, 64);
    const uint32_t wavemask = (__builtin_amdgcn_readfirstlane(wm) | (spec ? (1u << ik) : 0u)) & kvbits;
    // the same set in sequence positions: bit ik -> position 0, the bits below it move up by one
    const uint32_t seqmask = !spec ? wavemask
                                   : (((wavemask >> ik) & 1u) | ((wavemask & ((1u << ik) - 1u)) << 1) |
                                      (wavemask & ~((2u << ik) - 1u)));

    // ---- gathered-operand pipeline --------------------------------------------------------------------------------
    int idxr[D][MB];
    uint32_t identr[D];       // wave-uniform: idxr[S] stands for the identity offset
    u32x4 areg[D][MB];
    WslStep it[D + 2];
    it[0] = wsl_first(seqmask);
#pragma unroll
    for (int j = 1; j < D + 2; ++j) it[j] = wsl_next(it[j - 1]);
    // Straight-line (no branch around a load): the compiler's counted waits stay exact.  A step that does not exist
    // (j < 0) reads through a zero-sized resource: nothing is fetched.
    auto load_idx = [&](const WslStep &s, auto SET) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value;
      const int k = s.j < 0 ? 0 : koff(s.j);
      const __amdgpu_buffer_rsrc_t rP =
          make_rsrc(p.pair + static_cast<size_t>(k) * p.n_dst, s.j >= 0 ? tbl_bytes : 0u);
      identr[S] = (spec && s.j == 0) ? 0xffffffffu : 0u;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        idxr[S][mb] = static_cast<int>(__builtin_amdgcn_raw_buffer_load_b32(rP, goff[mb], 0, 0));
    };
    auto load_a = [&](const WslStep &s, auto SET) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value;
      const __amdgpu_buffer_rsrc_t r = make_rsrc(p.A, s.j >= 0 ? a_bytes : 0u);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const uint32_t idx = (static_cast<uint32_t>(grow[mb]) & identr[S]) |
                             (static_cast<uint32_t>(idxr[S][mb]) & ~identr[S]);
        const uint32_t rbase = idx * RB;                           // -1 -> >= kOob
        areg[S][mb] = __builtin_amdgcn_raw_buffer_load_b128(r, min(rbase + aoff, kOob) | (aoff & kOob), 0, 0);
      }
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    load_idx(it[0], Set0{});
    load_idx(it[1], Set1{});
    __builtin_amdgcn_sched_barrier(0);
    load_a(it[0], Set0{});
    load_a(it[1], Set1{});
    __builtin_amdgcn_sched_barrier(0);
    load_idx(it[D], Set0{});            // pair words of step D -> set 0 (consumed above)

    f32x4 acc[NB][MB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](const WslStep &s, auto SET) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value;
      if (s.j >= 0) {
        const char *cur = smem + koff(s.j) * SLICE;
        uint4 fa[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          fa[nb] = *reinterpret_cast<const uint4 *>(cur + foff[nb]);
          if constexpr (SPR < 4) fa[nb] = sel4(fvalid, fa[nb]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            acc[nb][mb] = mfma16<BF16>(fa[nb], __builtin_bit_cast(uint4, areg[S][mb]), acc[nb][mb]);
      }
    };
    // one step: at step t (S = t % D) areg[S] = rows of step t, idxr[S] = pair words of step t + D (requested one
    // step ago), it[i] = step t + i
    auto step = [&](auto SET) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value;
      compute(it[0], SET);
      __builtin_amdgcn_sched_barrier(0);
      load_idx(it[D + 1], std::integral_constant<int, (S + 1) % D>{});
      __builtin_amdgcn_sched_barrier(0);
      load_a(it[D], SET);
#pragma unroll
      for (int i = 0; i < D + 1; ++i) it[i] = it[i + 1];
      it[D + 1] = wsl_next(it[D]);
    };
    while (it[0].j >= 0) {
      step(Set0{});
      step(Set1{});     // may be a step past the end (no MFMAs, zero-sized loads)
    }

    // ---- epilogue: CPL consecutive channels per lane (the channel permutation of igemm_v4_kernel) ------------------
    float bv[CPL];          // (read per group: a training pass has no bias, and eight registers held across the walk
                            // cost the second workgroup of a CU)
#pragma unroll
    for (int q = 0; q < CPL; ++q) bv[q] = 0.f;
    if (p.bias) {
#pragma unroll
      for (int q = 0; q < CPL; ++q) bv[q] = to_float<BF16>(static_cast<const uint16_t *>(p.bias)[lgrp * CPL + q]);
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
            v0 = apply_act(v0 + bv[nb * 4 + 2 * h], p.act, p.act_alpha);
            v1 = apply_act(v1 + bv[nb * 4 + 2 * h + 1], p.act, p.act_alpha);
          }
          d[nb * 2 + h] = pack2<BF16>(v0, v1);
        }
      }
      const uint32_t rb = grow[mb] < 0 ? kOob : static_cast<uint32_t>(grow[mb]) * (COUT * 2);
      store_dwords<CPL / 2, SPX_AUX_OUT>(d, rOut, rb == kOob ? kOob : rb + lgrp * (CPL * 2));
    }
  }
}

template <int CIN, int COUT, bool BF16>
int launch_wsl_one(const WslArgs &a, hipStream_t s) {
  constexpr int NW = 16;
  const size_t lds = static_cast<size_t>(a.kv) * COUT * CIN * 2;
  auto kern = igemm_wsl_kernel<CIN, COUT, BF16, NW>;
  static bool attr_done = false;
  if (!attr_done) {
    SPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                32 * COUT * CIN * 2));
    attr_done = true;
  }
  WslArgs q = a;
  q.ngroups = div_up(a.n_dst, 32);
  // one workgroup (16 waves) per CU and as many as fit its LDS; a wave takes at least two groups
  const int per_cu = (COUT <= 32 && lds <= 76 * 1024) ? 2 : 1;     // (<= 64 registers: eight waves per SIMD)
  int grid = div_up(q.ngroups, 2 * NW);
  if (grid > 256 * per_cu) grid = 256 * per_cu;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, q);
  SPX_LAUNCH_CHECK();
  return 0;
}

}  // namespace

bool wsl_ok(const GemmParams &p, int dtype) {
  if (dtype != SPX_F16 && dtype != SPX_BF16) return false;
  const bool shape = (p.CIN == 16 && (p.COUT == 16 || p.COUT == 32)) || (p.CIN == 32 && (p.COUT == 32 || p.COUT == 64));
  if (!shape || p.kv > 32 || p.kv < 1 || !p.pair) return false;
  if (p.acc_mode || p.kbase || p.mask_words > 1) return false;
  if (p.tile_order == 0 && p.argsort && !p.cls) return false;          // (listed rows over row-order tables: v4)
  if (p.strideD != 1) return false;                                    // forward: rows of W contiguous
  const unsigned long long abytes = static_cast<unsigned long long>(p.n_src) * p.CIN * 2;
  const unsigned long long obytes = static_cast<unsigned long long>(p.n_dst) * p.COUT * 2;
  const unsigned long long pbytes = static_cast<unsigned long long>(p.n_dst) * 4ull;
  const unsigned long long wbytes = static_cast<unsigned long long>(p.COUT) * p.kv * p.CIN * 2;
  return abytes < 0x7fff0000ull && obytes < 0x7fff0000ull && wbytes < 0x7fff0000ull && pbytes < 0x7fff0000ull;
}

int launch_gather_gemm_wsl(const GemmParams &p, int dtype, hipStream_t s) {
  WslArgs a{};
  a.A = p.A;
  a.B = p.B;
  a.out = p.out;
  a.pair = p.pair;
  a.mask = p.mask;
  a.argsort = p.tile_order == 1 ? p.argsort : nullptr;
  a.bias = p.bias;
  a.strideK = p.strideK;
  a.strideN = p.strideN;
  a.n_dst = p.n_dst;
  a.n_src = p.n_src;
  a.kv = p.kv;
  a.identity_k = p.identity_k;
  a.b_reverse = p.b_reverse;
  a.act = p.act;
  a.act_alpha = p.act_alpha;
  const bool bf = dtype == SPX_BF16;
#define SPX_WSL(CI, CO) (bf ? launch_wsl_one<CI, CO, true>(a, s) : launch_wsl_one<CI, CO, false>(a, s))
  if (p.CIN == 16 && p.COUT == 16) return SPX_WSL(16, 16);
  if (p.CIN == 16 && p.COUT == 32) return SPX_WSL(16, 32);
  if (p.CIN == 32 && p.COUT == 32) return SPX_WSL(32, 32);
  return SPX_WSL(32, 64);
#undef SPX_WSL
}

}  // namespace spx
