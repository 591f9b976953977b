// Gather-GEMM v5 for gfx950: persistent workgroups with a loader wave and LDS-DMA.
//
// Same contract and the same arithmetic as igemm_v4_kernel (igemm.hip) -- output-stationary
// implicit GEMM over the [kv, n_dst] pair table, every output row written exactly once, per-row
// accumulation order "identity offset first, then ascending offsets", v_mfma_f32_16x16x32_{f16,bf16}
// with the same lane <-> reduction-element assignment -- so forward results are bit-identical to v4.
// What differs is who moves what:
//
//  * A workgroup is 4 CONSUMER waves (32 destination rows each: a 128-row tile) plus 1 LOADER wave.
//    The loader never touches a VGPR operand: everything it fetches goes global -> LDS through
//    buffer_load_dwordx4 ... lds (LDS-DMA):
//      - the tile's whole pair table [kv][128] (13.8 KB at kv = 27) and its 128 mask words, ONE round
//        trip after the tile is known -- the consumers read their pair words from LDS, the dependent
//        chain "mask -> pair words -> rows" of v4 loses a trip to memory;
//      - the weight slice of every step into an R-stage ring, up to R - 1 steps ahead of the MFMAs,
//        as it lies in memory (the XOR swizzle of the LDS image is applied to the SOURCE address of
//        each lane).  dgrad reads the KRSC tensor in place as well: its [reduction][channel] image
//        is consumed through ds_read_b64_tr_b16, the perm + ds_write_b32 transpose of v4 is gone.
//    Consumers therefore issue nothing but the gathered rows (straight into MFMA operand
//    registers, NSET steps in flight) and their result stores; no weight staging, no register
//    pressure from it, no LDS bank conflicts from the transposing stores.
//  * Workgroups are PERSISTENT: 2 per CU, each walks the tiles of its XCD's contiguous range
//    (block b runs on XCD b % 8).  While the consumers work on tile j the loader already has tile
//    j + 1's tables and masks in flight, and the consumers fetch tile j + 1's identity rows (SubM:
//    97 % of all pairs of a uniform scene) before they finish tile j -- reads, MFMAs and result
//    stores of different tiles overlap inside one workgroup instead of marching in lock-step.
//  * One s_barrier per step hands a ring stage from the loader to the consumers; the loader waits
//    for its DMA with counted s_waitcnt vmcnt(N) (never 0 inside a tile), the consumers' row loads
//    stay in flight across the barriers.
//
// Limits (the dispatcher falls back to v4 otherwise): 16-bit operands, gathered rows of at most
// 128 bytes, at most 64 output channels, kernel volume <= 32, tables in row order (no argsort).
// Reference kernels this stands in for: the implicit-GEMM forward / input-gradient kernels the
// tuner picks in spconv/csrc/sparse/convops.py:1363-1446 (multi-stage smem pipeline, core.py:542).
#include "igemm_defs.h"

namespace spx {
namespace {

constexpr int k5Consumers = 4;
constexpr int k5Threads = (k5Consumers + 1) * 64;
constexpr int k5TM = 128;                       // rows per tile: 4 consumer waves x 2 x 16

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction only takes an immediate)
#define SPX_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vmcnt(int n) {
  switch (n) {
    SPX_VMCNT_CASE(0) SPX_VMCNT_CASE(1) SPX_VMCNT_CASE(2) SPX_VMCNT_CASE(3) SPX_VMCNT_CASE(4)
    SPX_VMCNT_CASE(5) SPX_VMCNT_CASE(6) SPX_VMCNT_CASE(7) SPX_VMCNT_CASE(8) SPX_VMCNT_CASE(9)
    SPX_VMCNT_CASE(10) SPX_VMCNT_CASE(11) SPX_VMCNT_CASE(12) SPX_VMCNT_CASE(13) SPX_VMCNT_CASE(14)
    SPX_VMCNT_CASE(15) SPX_VMCNT_CASE(16) SPX_VMCNT_CASE(17) SPX_VMCNT_CASE(18) SPX_VMCNT_CASE(19)
    SPX_VMCNT_CASE(20) SPX_VMCNT_CASE(21) SPX_VMCNT_CASE(22) SPX_VMCNT_CASE(23) SPX_VMCNT_CASE(24)
    SPX_VMCNT_CASE(25) SPX_VMCNT_CASE(26) SPX_VMCNT_CASE(27) SPX_VMCNT_CASE(28) SPX_VMCNT_CASE(29)
    SPX_VMCNT_CASE(30) SPX_VMCNT_CASE(31) SPX_VMCNT_CASE(32) SPX_VMCNT_CASE(33) SPX_VMCNT_CASE(34)
    SPX_VMCNT_CASE(35) SPX_VMCNT_CASE(36) SPX_VMCNT_CASE(37) SPX_VMCNT_CASE(38) SPX_VMCNT_CASE(39)
    SPX_VMCNT_CASE(40) SPX_VMCNT_CASE(41) SPX_VMCNT_CASE(42) SPX_VMCNT_CASE(43) SPX_VMCNT_CASE(44)
    SPX_VMCNT_CASE(45) SPX_VMCNT_CASE(46) SPX_VMCNT_CASE(47) SPX_VMCNT_CASE(48) SPX_VMCNT_CASE(49)
    SPX_VMCNT_CASE(50) SPX_VMCNT_CASE(51) SPX_VMCNT_CASE(52) SPX_VMCNT_CASE(53) SPX_VMCNT_CASE(54)
    SPX_VMCNT_CASE(55) SPX_VMCNT_CASE(56) SPX_VMCNT_CASE(57) SPX_VMCNT_CASE(58) SPX_VMCNT_CASE(59)
    SPX_VMCNT_CASE(60) SPX_VMCNT_CASE(61) SPX_VMCNT_CASE(62)
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
}
#undef SPX_VMCNT_CASE

// workgroup barrier that leaves vector-memory operations in flight (a __syncthreads() would drain
// the loader's LDS-DMA queue): LDS operations of this wave retired, then s_barrier
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef __attribute__((address_space(3))) void lds_void_t;

// LDS-DMA: every lane fetches 16 (4) bytes at byte offset vo + so of the buffer; the wave's 64 pieces
// land at dst + 16 (4) * lane.  (Plain functions, not inlined builtin calls inside the kernel template:
// hipcc 7.2's host pass silently drops a kernel template whose body names this builtin.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char *dst, uint32_t vo, uint32_t so) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)dst, 16, vo, so, 0, 0);
}
__device__ __forceinline__ void dma4(__amdgpu_buffer_rsrc_t r, char *dst, uint32_t vo, uint32_t so) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)dst, 4, vo, so, 0, 0);
}

// 32-byte granule swizzle of the [reduction row][channel] image dgrad reads with ds_read_b64_tr_b16
// (the layout of wgrad's operand stages: rows r..r+3 and r+8..r+11 of one granule column cover all
// 64 banks)
__device__ __forceinline__ int v5_trx(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

typedef short v5_s16x4 __attribute__((ext_vector_type(4)));

// 8 consecutive reduction rows (row0 .. row0 + 7) of channel gran * 16 + lrow
__device__ __forceinline__ uint4 v5_trfrag(const char *stage, int row0, int lrow, int gran) {
  const int r = row0 + (lrow >> 2);
  const char *a0 = stage + r * 128 + ((gran ^ v5_trx(r)) << 5) + ((lrow & 3) << 3);
  const int r1 = r + 4;
  const char *a1 = stage + r1 * 128 + ((gran ^ v5_trx(r1)) << 5) + ((lrow & 3) << 3);
  typedef __attribute__((address_space(3))) v5_s16x4 lds_s16x4;
  const v5_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_s16x4 *)(__attribute__((address_space(3))) char *)a0);
  const v5_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_s16x4 *)(__attribute__((address_space(3))) char *)a1);
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// Tiles of workgroup b (b % 8 = its XCD): XCD x owns the contiguous tile range [lo, lo + cnt), its
// G / 8 workgroups take every (G / 8)-th tile of it.
struct TileWalk {
  int first, stride, count;
};
__host__ __device__ inline TileWalk v5_tiles(int b, int G, int ntiles) {
  const int x = b & 7, l = b >> 3, gx = G >> 3;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int cnt = q + (x < r ? 1 : 0);
  const int lo = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  TileWalk w;
  w.first = lo + l;
  w.stride = gx;
  w.count = l < cnt ? (cnt - l + gx - 1) / gx : 0;
  return w;
}

template <int COUT, bool BT, int NKS, int R>
constexpr size_t v5_smem_bytes(int kv) {
  const size_t lrows = BT ? NKS * 32 : COUT;
  const size_t kvp = (kv + 1) & ~1;
  return R * lrows * 128 + 2 * kvp * k5TM * 4 + 2 * k5TM * 4 + 64 + static_cast<size_t>(k5TM) * COUT * 2;
}

}  // namespace

// (the kernel has external linkage: its address is taken for hipFuncSetAttribute)
// DT: 0 = f16, 1 = bf16.  BT = false: forward (weight rows [n][reduction] contiguous); BT = true: dgrad
// (the slice lies [reduction][n] in memory).  NKS: 64-byte halves of a gathered row that exist.
// NSET: register sets of gathered rows in flight per consumer wave; R: weight ring stages.
template <int COUT, int DT, bool BT, int NKS, int NSET, int R>
__global__ void __launch_bounds__(k5Threads, 3)
igemm_v5_kernel(const void *argA, const void *argB, const uint32_t *arg_mask,
                const int32_t *arg_argsort, const int32_t *arg_pair, int n_dst, int n_src, int CIN,
                int kv, int identity_k, int flags, GemmRest rest) {
  constexpr bool BF16 = DT == 1;
  constexpr int ES = 2, MB = 2, TM = k5TM;
  constexpr int NB = COUT / 16;
  constexpr int CPL = NB * 4;
  constexpr int LROWS = BT ? NKS * 32 : COUT;         // 128-byte rows of one staged weight slice
  constexpr int B_BYTES = LROWS * 128;
  constexpr int NWI = LROWS / 8;                       // LDS-DMA instructions per slice
  static_assert(R >= 2 && NSET >= 1 && NWI >= 1, "ring / register sets");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int kvp = (kv + 1) & ~1;
  char *sW = smem;
  char *sPair = smem + R * B_BYTES;                    // [2][kvp][TM] int32
  char *sMask = sPair + 2 * kvp * TM * 4;              // [2][TM] uint32
  uint32_t *sMeta = reinterpret_cast<uint32_t *>(sMask + 2 * TM * 4);   // [2] tile masks
  // result rows of a tile on their way out: [TM][COUT] 16-bit, 16-byte piece p of row r at piece
  // p ^ (r & (PPR - 1)).  The consumers only WRITE LDS; the loader wave stores the tile to memory as whole
  // lines.  (A consumer wave that issued the stores itself would carry vector-memory writes next to its
  // gathers, and the compiler then drains the wave's whole load queue -- s_waitcnt vmcnt(0) -- before the
  // first MFMA of every loop trip: loads and stores are not ordered among each other on one counter.)
  constexpr int OROWB = COUT * ES, PPR = OROWB / 16;
  char *sOut = smem + R * B_BYTES + 2 * kvp * TM * 4 + 2 * TM * 4 + 64;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = (n_dst + TM - 1) / TM;
  const TileWalk tw = v5_tiles(static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x), ntiles);
  const int nj = tw.count;
  if (nj == 0) return;
  const bool b_reverse = flags & 1;
  const bool spec = identity_k >= 0;
  const uint32_t idbit = spec ? (1u << identity_k) : 0u;
  const uint32_t kvmask = kv < 32 ? (1u << kv) - 1u : 0xffffffffu;
  const uint32_t rowB = static_cast<uint32_t>(CIN) * ES;

  const __amdgpu_buffer_rsrc_t rO = make_rsrc(rest.out, static_cast<uint32_t>(n_dst) * (COUT * ES));

  if (wave == k5Consumers) {
    // ======================= loader wave =======================
    const uint32_t w_bytes = static_cast<uint32_t>(rest.COUT) * kv * rowB;
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(argB, w_bytes);
    const __amdgpu_buffer_rsrc_t rP = make_rsrc(arg_pair, static_cast<uint32_t>(kv) * n_dst * 4u);
    const __amdgpu_buffer_rsrc_t rM = make_rsrc(arg_mask, arg_mask ? static_cast<uint32_t>(n_dst) * 4u : 0u);
    // source offset of the 16-byte piece lane L of DMA instruction j delivers: piece q = 64 j + L
    // of the image = row q / 8, PHYSICAL slot q % 8 -> the logical slot the swizzle maps there
    uint32_t wvo[NWI < 2 ? 2 : NWI];
#pragma unroll
    for (int j = 0; j < NWI; ++j) {
      const int q = j * 64 + lane, rr = q >> 3, ps = q & 7;
      if constexpr (!BT) {
        const int x = ((rr >> 1) & 1) | (((rr / CPL) & 3) << 1);
        const int ls = ps ^ x;
        const bool ok = rr < COUT && static_cast<uint32_t>(ls * 16) < rowB;
        wvo[j] = ok ? static_cast<uint32_t>(rr) * static_cast<uint32_t>(rest.strideN) * ES + ls * 16u : kOob;
      } else {
        const int ls = (((ps >> 1) ^ v5_trx(rr)) << 1) | (ps & 1);
        const bool ok = rr < CIN && ls * 16 < COUT * ES;
        wvo[j] = ok ? static_cast<uint32_t>(rr) * static_cast<uint32_t>(rest.strideD) * ES + ls * 16u : kOob;
      }
    }
    int seq = 0;                       // LDS-DMA instructions issued so far (all of this wave's VMEM)
    int seq_end[R];                    // seq after the last instruction of the slice in each stage
#pragma unroll
    for (int i = 0; i < R; ++i) seq_end[i] = 0;
    auto issue_w = [&](int k, int g) __attribute__((always_inline)) {
      const int st = g % R;
      const int kb = b_reverse ? kv - 1 - k : k;
      const uint32_t so = static_cast<uint32_t>(kb) * static_cast<uint32_t>(rest.strideK) * ES;
      char *dst = sW + st * B_BYTES;
#pragma unroll
      for (int j = 0; j < NWI; ++j)
        dma16(rW, dst + j * 1024, wvo[j], so);
      seq += NWI;
#pragma unroll
      for (int i = 0; i < R; ++i)
        if (i == st) seq_end[i] = seq;
    };
    auto seq_end_of = [&](int g) __attribute__((always_inline)) {
      const int st = g % R;
      int v = 0;
#pragma unroll
      for (int i = 0; i < R; ++i)
        if (i == st) v = seq_end[i];
      return v;
    };
    // tables of a tile: kvp / 2 instructions for the pair columns (two 512-byte columns each; lanes
    // 0..31 the even column, 32..63 the odd one) and two dword-sized ones for the 128 mask words.  A
    // 16-byte piece that straddles the END of the pair table (last column, last tile, n_dst % 4 != 0)
    // would depend on how the buffer unit range-checks a partially out-of-range access: that one tile
    // takes dword-sized pieces instead.  Rows past n_dst read the next column's head or zeros -- their
    // results are dropped by the epilogue.
    const bool pair16 = (flags >> 1) & 1;
    auto prefetch = [&](int tile, int buf) __attribute__((always_inline)) {
      char *dp = sPair + buf * (kvp * TM * 4);
      if (pair16 && !(tile == ntiles - 1 && (n_dst & 3))) {
        const uint32_t rbase = static_cast<uint32_t>(tile) * TM * 4u + (lane & 31) * 16u;
        for (int i = 0; i < (kvp >> 1); ++i) {
          const int c = 2 * i + (lane >> 5);
          const uint32_t vo = c < kv ? static_cast<uint32_t>(c) * static_cast<uint32_t>(n_dst) * 4u + rbase : kOob;
          dma16(rP, dp + i * 1024, vo, 0);
        }
        seq += kvp >> 1;
      } else {
        for (int c = 0; c < kv; ++c) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t t = static_cast<uint32_t>(tile) * TM + h * 64 + lane;
            const uint32_t vo = t < static_cast<uint32_t>(n_dst)
                                    ? (static_cast<uint32_t>(c) * static_cast<uint32_t>(n_dst) + t) * 4u : kOob;
            dma4(rP, dp + c * 512 + h * 256, vo, 0);
          }
        }
        seq += 2 * kv;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t t = static_cast<uint32_t>(tile) * TM + h * 64 + lane;
        const uint32_t mo = t < static_cast<uint32_t>(n_dst) ? t * 4u : kOob;
        dma4(rM, sMask + buf * (TM * 4) + h * 256, mo, 0);
      }
      seq += 2;
    };

    int gs = 0;          // global step number of the current tile's first step (ring stage = step % R)
    int issued = 0;      // global steps whose weight slice has been requested
    prefetch(tw.first, 0);
    if (spec) {
      issue_w(identity_k, 0);
      issued = 1;
    }
    // tile 0's tables, masks and identity weights (later tiles: waited for before the previous tile's
    // result stores are issued, see below)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int j = 0; j < nj; ++j) {
      const uint32_t *mw = reinterpret_cast<const uint32_t *>(sMask + (j & 1) * (TM * 4));
      uint32_t wm = mw[lane] | mw[lane + 64];
      wm |= __shfl_xor(wm, 1, 64);
      wm |= __shfl_xor(wm, 2, 64);
      wm |= __shfl_xor(wm, 4, 64);
      wm |= __shfl_xor(wm, 8, 64);
      wm |= __shfl_xor(wm, 16, 64);
      wm |= __shfl_xor(wm, 32, 64);
      uint32_t tilemask = __builtin_amdgcn_readfirstlane(wm);
      if (!arg_mask) tilemask = 0xffffffffu;
      tilemask = (tilemask & kvmask) | idbit;
      if (lane == 0) sMeta[j & 1] = tilemask;
      wg_barrier();                                        // META_j: tables + identity weights of tile j are in LDS
      int done = gs;                                       // steps below `done` have been computed
      uint32_t iss = tilemask & ~idbit;                    // offsets of this tile still to request
      const int nsteps = __builtin_popcount(iss) + (spec ? 1 : 0);
      const bool have_next = j + 1 < nj;
      bool pref_done = false, next_ident = false;
      auto topup = [&]() __attribute__((always_inline)) {
        while (iss && issued <= done + R - 1) {
          const int k = __builtin_ctz(iss);
          iss &= iss - 1;
          issue_w(k, issued);
          ++issued;
        }
        if (!pref_done && have_next) {
          prefetch(tw.first + (j + 1) * tw.stride, (j + 1) & 1);
          pref_done = true;
        }
        if (spec && have_next && !iss && !next_ident && issued <= done + R - 1) {
          issue_w(identity_k, issued);
          ++issued;
          next_ident = true;
        }
      };
      topup();
      for (int s = spec ? 1 : 0; s < nsteps; ++s) {
        const int g = gs + s;
        int n = seq - seq_end_of(g);                       // instructions younger than step g's slice
        wait_vmcnt(n > 63 ? 63 : n);
        wg_barrier();                                      // B_g: stage g % R is complete
        done = g;
        topup();
      }
      gs += nsteps;
      if (spec && have_next && !next_ident) {              // (the ring was full until the last barrier)
        issue_w(identity_k, issued);
        ++issued;
      }
      wg_barrier();                                        // E_j: the consumers have written the tile's rows to sOut
      // loads first, stores second: a counted wait cannot tell a younger store from an older load
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (tile j + 1's tables, masks, identity weights)
      {
        const int tile = tw.first + j * tw.stride;
        const uint32_t obase = static_cast<uint32_t>(tile) * (TM * OROWB);
        constexpr int NOI = TM * OROWB / 1024;             // 16-byte pieces per lane
        constexpr int OB = NOI < 4 ? NOI : 4;
#pragma unroll
        for (int i0 = 0; i0 < NOI; i0 += OB) {
          u32x4 v[OB < 2 ? 2 : OB];
#pragma unroll
          for (int i = 0; i < OB; ++i) {
            const int q = (i0 + i) * 64 + lane, r = q / PPR, pc = q % PPR;
            v[i] = *reinterpret_cast<const u32x4 *>(sOut + r * OROWB + ((pc ^ (r & (PPR - 1))) << 4));
          }
#pragma unroll
          for (int i = 0; i < OB; ++i) {
            const uint32_t vo = obase + static_cast<uint32_t>((i0 + i) * 64 + lane) * 16u;   // past the end: dropped
            if (rest.dbg & 0x400) __builtin_amdgcn_raw_buffer_store_b128(v[i], rO, vo, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(v[i], rO, vo, 0, 2);
          }
        }
      }
    }
    return;
  }

  // ======================= consumer waves =======================
  const int lrow = lane & 15, lgrp = lane >> 4;
  const uint32_t a_bytes = static_cast<uint32_t>(n_src) * rowB;
  constexpr int AK = NKS < 2 ? 2 : NKS;                 // (register arrays stay at >= 2 elements)
  uint32_t aoff[AK];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const uint32_t c = ks * 64 + lgrp * 16;
    aoff[ks] = c < rowB ? c : kOob;
  }
  // fwd: MFMA row (g = i >> 2, e = i & 3) of channel block nb carries channel g * CPL + nb * 4 + e
  // (each lane ends up with CPL consecutive channels of its row); the staged slice is
  // [channel][64 reduction elements], 16-byte slots XOR-swizzled with (bit 1 of the channel, g)
  auto swzB = [](int row, int sl) __attribute__((always_inline)) {
    const int x = ((row >> 1) & 1) | (((row / CPL) & 3) << 1);
    return row * 128 + ((sl ^ x) << 4);
  };

  using acc_t = f32x4;
  acc_t acc[NB][MB];
  u32x4 areg[NSET][MB][AK];
  u32x4 aid[MB][AK];                                    // rows of the identity step (this tile / the next one)
  int kq[NSET];                                         // offset whose rows sit in set S, -1 = none
  uint32_t vany = 0;                                    // bit S: some row of this wave has a pair in set S

  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = acc_t{0, 0, 0, 0};
  };
  // this wave's rows of tile `tile`: row t = tile * TM + wave * 32 + mb * 16 + lrow
  auto load_identity = [&](int tile, bool exists) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t r = make_rsrc(argA, exists ? a_bytes : 0u);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const uint32_t t = static_cast<uint32_t>(tile) * TM + wave * 32 + mb * 16 + lrow;
      const uint32_t rbase = t < static_cast<uint32_t>(n_dst) ? t * rowB : kOob;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const uint32_t vo = min(rbase + aoff[ks], kOob) | (aoff[ks] & kOob);
        aid[mb][ks] = __builtin_amdgcn_raw_buffer_load_b128(r, vo, 0, 0);
      }
    }
  };
  // MFMAs of one step: weight stage `cur`, gathered rows `a`
  auto mfma_block = [&](const char *cur, const u32x4 (&a)[MB][AK]) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        uint4 fa;
        if constexpr (!BT) {
          fa = *reinterpret_cast<const uint4 *>(cur + swzB((lrow >> 2) * CPL + nb * 4 + (lrow & 3), ks * 4 + lgrp));
        } else {
          fa = v5_trfrag(cur, ks * 32 + lgrp * 8, lrow, nb);
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          acc[nb][mb] = mfma16<BF16>(fa, __builtin_bit_cast(uint4, a[mb][ks]), acc[nb][mb]);
      }
    }
  };

  const bool plain = rest.bias == nullptr && rest.act == SPX_ACT_NONE;   // uniform: training path
  float bv[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) bv[q] = 0.f;
  if (rest.bias) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const int ch = BT ? (q >> 2) * 16 + lgrp * 4 + (q & 3) : lgrp * CPL + q;
      bv[q] = to_float<BF16>(static_cast<const uint16_t *>(rest.bias)[ch]);
    }
  }
  auto epilogue = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int r = wave * 32 + mb * 16 + lrow;            // row of the tile
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
      char *row = sOut + r * OROWB;
      if constexpr (!BT) {
        // CPL consecutive channels per lane: bytes [lgrp * CPL * 2, + CPL * 2) of the row
        if constexpr (CPL >= 8) {
#pragma unroll
          for (int q = 0; q < CPL / 8; ++q) {
            const int pc = lgrp * (CPL / 8) + q;
            *reinterpret_cast<u32x4 *>(row + ((pc ^ (r & (PPR - 1))) << 4)) =
                u32x4{d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
          }
        } else {
          // 16 output channels: 4 per lane = half a piece
          const int pc = lgrp >> 1;
          *reinterpret_cast<u32x2 *>(row + ((pc ^ (r & (PPR - 1))) << 4) + (lgrp & 1) * 8) = u32x2{d[0], d[1]};
        }
      } else {
        // natural channel order (the transpose read ties MFMA row i to channel nb * 16 + i): four
        // consecutive channels (8 bytes) per lane and channel block
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int pc = nb * 2 + (lgrp >> 1);
          *reinterpret_cast<u32x2 *>(row + ((pc ^ (r & (PPR - 1))) << 4) + (lgrp & 1) * 8) =
              u32x2{d[nb * 2], d[nb * 2 + 1]};
        }
      }
    }
  };

  int gs = 0;                                           // global step of the current tile's first step
  zero_acc();
  load_identity(tw.first, spec);                        // (regular conv: zero-sized resource, nothing fetched)
  for (int j = 0; j < nj; ++j) {
    const int tile = tw.first + j * tw.stride;
    wg_barrier();                                       // META_j
    const uint32_t tilemask = __builtin_amdgcn_readfirstlane(sMeta[j & 1]);
    uint32_t rest_bits = tilemask & ~idbit;             // offsets whose rows are still to be requested
    const int nsteps = __builtin_popcount(rest_bits) + (spec ? 1 : 0);
    // this lane's column of the tile's pair table: word (k, row) at k * TM + wave * 32 + mb * 16 + lrow
    const int32_t *pb = reinterpret_cast<const int32_t *>(sPair + (j & 1) * (kvp * TM * 4)) + wave * 32 + lrow;
    if (spec) {
      // identity step: its rows were requested a tile ago, its weights are in stage gs % R (META_j)
      mfma_block(sW + (gs % R) * B_BYTES, aid);
      __builtin_amdgcn_sched_barrier(0);
      load_identity(tile + tw.stride, j + 1 < nj);      // next tile's rows: in flight while this tile computes
    }
    auto load_step = [&](auto SET) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value;
      const int k = rest_bits ? __builtin_ctz(rest_bits) : -1;
      rest_bits &= rest_bits - 1;                       // (0 stays 0)
      kq[S] = k;
      const int kk = k < 0 ? 0 : k;
      const __amdgpu_buffer_rsrc_t r = make_rsrc(argA, k >= 0 ? a_bytes : 0u);
      int idx[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) idx[mb] = pb[kk * TM + mb * 16];
      bool any = false;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        any = any || idx[mb] >= 0;
        const uint32_t rbase = static_cast<uint32_t>(idx[mb]) * rowB;          // -1 -> >= kOob
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const uint32_t vo = min(rbase + aoff[ks], kOob) | (aoff[ks] & kOob);
          areg[S][mb][ks] = __builtin_amdgcn_raw_buffer_load_b128(r, vo, 0, 0);
        }
      }
      const bool wany = __builtin_amdgcn_ballot_w64(any) != 0ull && k >= 0;
      vany = (vany & ~(1u << S)) | (wany ? (1u << S) : 0u);
    };
    auto do_step = [&](auto SET, int g) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value;
      if (kq[S] >= 0) {
        wg_barrier();                                   // B_g: the loader's slice for this step has landed
        if ((vany >> S) & 1u) mfma_block(sW + (g % R) * B_BYTES, areg[S]);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_step(SET);
      __builtin_amdgcn_sched_barrier(0);
    };
    // rows of the first NSET steps (a tile of a uniform scene has fewer: everything is in flight at once)
    {
      // (issue order pinned: the compiler's counted waits at the loop head are only as good as the
      // worst order in which a path into the loop issued these loads)
      __builtin_amdgcn_sched_barrier(0);
      load_step(std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NSET > 1) load_step(std::integral_constant<int, 1>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NSET > 2) load_step(std::integral_constant<int, 2>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NSET > 3) load_step(std::integral_constant<int, 3>{});
      __builtin_amdgcn_sched_barrier(0);
    }
    int g = gs + (spec ? 1 : 0);
    while (kq[0] >= 0) {
      do_step(std::integral_constant<int, 0>{}, g);
      if constexpr (NSET > 1) do_step(std::integral_constant<int, 1>{}, g + 1);
      if constexpr (NSET > 2) do_step(std::integral_constant<int, 2>{}, g + 2);
      if constexpr (NSET > 3) do_step(std::integral_constant<int, 3>{}, g + 3);
      g += NSET;
    }
    gs += nsteps;
    epilogue();
    wg_barrier();                                       // E_j: the loader stores the tile
    zero_acc();
  }
}

namespace {

template <int COUT, int DT, bool BT, int NKS, int NSET, int R>
int launch_v5_one(const GemmParams &p, const GemmRest &r, int flags, int wgs, hipStream_t s) {
  const size_t smem = v5_smem_bytes<COUT, BT, NKS, R>(p.kv);
  static size_t attr_set = 0;                      // dynamic LDS beyond 64 KB needs the attribute
  if (smem > attr_set) {
    SPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&igemm_v5_kernel<COUT, DT, BT, NKS, NSET, R>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(smem)));
    attr_set = smem;
  }
  const int ntiles = div_up(p.n_dst, k5TM);
  int G = (ntiles + 7) & ~7;
  if (G > wgs) G = wgs;
  hipLaunchKernelGGL((igemm_v5_kernel<COUT, DT, BT, NKS, NSET, R>), dim3(G), dim3(k5Threads), smem, s, p.A, p.B, p.mask, p.argsort, p.pair, p.n_dst,
                     p.n_src, p.CIN, p.kv, p.identity_k, flags, r);
  SPX_LAUNCH_CHECK();
  return 0;
}

template <int COUT, int DT, int NSET, int R>
int launch_v5_shape(const GemmParams &p, const GemmRest &r, int flags, int wgs, hipStream_t s) {
  const bool half = p.CIN * 2 <= 64;
  if (p.strideD == 1) {
    return half ? launch_v5_one<COUT, DT, false, 1, NSET, R>(p, r, flags, wgs, s)
                : launch_v5_one<COUT, DT, false, 2, NSET, R>(p, r, flags, wgs, s);
  }
  return half ? launch_v5_one<COUT, DT, true, 1, NSET, R>(p, r, flags, wgs, s)
              : launch_v5_one<COUT, DT, true, 2, NSET, R>(p, r, flags, wgs, s);
}

template <int DT, int NSET, int R>
int launch_v5_cout(const GemmParams &p, const GemmRest &r, int flags, int wgs, hipStream_t s) {
  switch (p.COUT) {
    case 16: return launch_v5_shape<16, DT, NSET, R>(p, r, flags, wgs, s);
    case 32: return launch_v5_shape<32, DT, NSET, R>(p, r, flags, wgs, s);
    case 64: return launch_v5_shape<64, DT, NSET, R>(p, r, flags, wgs, s);
  }
  return -1;
}

}  // namespace

bool v5_ok(const GemmParams &p, int dtype) {
  if (dtype != SPX_F16 && dtype != SPX_BF16) return false;
  if (!p.pair || p.argsort || p.tile_order || p.kv > 32 || p.kbase != 0 || p.mask_words > 1) return false;
  if (p.acc || p.acc_mode) return false;
  if (p.COUT != 16 && p.COUT != 32 && p.COUT != 64) return false;
  if (p.CIN % 8 != 0 || p.CIN * 2 > 128) return false;
  const unsigned long long tbytes = static_cast<unsigned long long>(p.kv) * p.n_dst * 4ull;
  const unsigned long long abytes = static_cast<unsigned long long>(p.n_src) * p.CIN * 2ull;
  const unsigned long long obytes = static_cast<unsigned long long>(p.n_dst) * p.COUT * 2ull;
  const unsigned long long wbytes = static_cast<unsigned long long>(p.COUT) * p.kv * p.CIN * 2ull;
  return tbytes < 0x7fff0000ull && abytes < 0x7fff0000ull && obytes < 0x7fff0000ull && wbytes < 0x7fff0000ull;
}

int launch_v5(const GemmParams &p, const GemmRest &r, int dtype, hipStream_t s) {
  static const int wgs_env = env_int("SPX_V5_WGS", 512);        // persistent workgroups (2 per CU)
  static const int variant = env_int("SPX_V5_VARIANT", 0);       // A/B: 1 = two register sets, 3-stage ring
  int wgs = wgs_env < 8 ? 8 : (wgs_env & ~7);
  static const int pair16 = env_int("SPX_V5_PAIR16", 1);       // A/B: 0 = dword-sized table DMA
  const int flags = (p.b_reverse & 1) | (pair16 ? 2 : 0);
  if (variant == 1)
    return dtype == SPX_BF16 ? launch_v5_cout<1, 2, 3>(p, r, flags, wgs, s) : launch_v5_cout<0, 2, 3>(p, r, flags, wgs, s);
  return dtype == SPX_BF16 ? launch_v5_cout<1, 4, 4>(p, r, flags, wgs, s) : launch_v5_cout<0, 4, 4>(p, r, flags, wgs, s);
}

}  // namespace spx
