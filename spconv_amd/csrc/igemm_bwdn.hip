// Backward of a NARROW layer (C, K <= 32 channels) from ONE gather per pair.
//
// The pair-list backward (igemm.hip: igemm_bwd_kernel) gathers dout[pair] for dgrad and then, per pair
// again, feat[in] and dout[out] for the weight gradient: three rows per pair through the L2.  On the large
// levels of a backbone -- 300-400 k voxels, 5-15 pairs per voxel, 16 / 32 channels -- that gather traffic IS
// the kernel (4.6-5.0 TB/s against a 6.5 TB/s random-row ceiling).  By the symmetry
// of a rulebook
//     din[i]  = sum_r  dout[T[r][i]] . W_k(r)            (the dgrad tile walk, T = the dgrad table)
//     dW_k(r) = sum_i  dout[T[r][i]]^T (x) feat[i]       (the same gathered rows, the tile's own feat rows)
// so one walk over 128-row tiles of the INPUT rows feeds every gathered dout tile to both products:
//   * dgrad: the gathered rows are MFMA A fragments as they arrive (row = lane & 15, 16 bytes per lane, the
//     direct-fragment form of igemm_v4), the weight slice comes as B fragments straight from a [kv][C][K]
//     copy of the weights in the L2 (55 KB at C = K = 32);
//   * wgrad: the same registers are written to an LDS stage, the tile's feat rows sit in a second stage for
//     the whole tile, and both reach the MFMA through ds_read_b64_tr_b16 (reduction index = row);
//     wave w owns output tile w of the K x C matrix and keeps ONE accumulator tile per table row:
//     27 x 4 registers -- which is why this form exists for <= 32 channels only (27 x C x K fp32 is
//     110 KB at 32, 442 KB at 64: the pair-list form accumulates one offset at a time for that reason).
// Workgroups are persistent (two per CU) and stride over the tiles; their partial dW (kv x K x C fp32 each)
// are summed in a fixed order by bwdn_reduce_kernel: deterministic, no atomics.  No Native lists, no
// range plan.  Same arithmetic per element as the pair-list kernels up to the fp32 summation order.
#include "igemm_defs.h"

namespace spx {
namespace {

constexpr int kBnTile = 128;                    // input rows per tile
constexpr int kBnStage = kBnTile * 64;          // one LDS stage: [128 rows][64 bytes], see ctr_slot
constexpr int kBnMaxKv = 27;                    // accumulator tiles per wave (table rows)

// LDS stages hold [128 rows][32 channels] of 16-bit values, 64 bytes per row, unswizzled: a ds_read_b64_tr_b16
// touches rows r..r+3 of four 8-row groups, 32 bytes (one 16-channel granule) each -- 512 bytes over the eight
// 32-byte bank slots a 64-byte row stride offers, i.e. the two bank cycles 512 bytes need anyway; the fragment
// writes (16 rows x 64 bytes per m-block) are consecutive.
__device__ __forceinline__ int ctr_slot(int row, int sl) { return row * 64 + sl * 16; }
typedef short s16x4 __attribute__((ext_vector_type(4)));
// 8 consecutive rows (row0 .. row0 + 7 as seen by this lane group) of channel gran * 16 + lrow
__device__ __forceinline__ uint4 ctr_frag(const char *stage, int row0, int lrow, int gran) {
  const char *a0 = stage + (row0 + (lrow >> 2)) * 64 + (gran << 5) + ((lrow & 3) << 3);
  const char *a1 = a0 + 4 * 64;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(__attribute__((address_space(3))) char *)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(__attribute__((address_space(3))) char *)a1);
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// compile-time loop: body(std::integral_constant<int, I>) for I = 0 .. N-1 (accumulator tiles are indexed by
// the table row: the index must be a constant, whatever the unroller thinks of a 27-fold body)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&body) {
  if constexpr (I < N) {
    body(std::integral_constant<int, I>{});
    static_for<I + 1, N>(body);
  }
}

struct BwdnParams {
  const void *feat;        // [n_in, C]
  const void *dout;        // [n_out, K]
  const void *wt;          // [kv][C][K]: weights with the dout channel contiguous
  void *din;               // [n_in, C] or null
  float *partial;          // [G][kv][K][C] fp32
  const int32_t *table;    // [kv, n_in]: dout row of (table row r, input row i), or -1
  const uint32_t *mask;    // [n_in]: bit r <=> table[r][i] >= 0
  int n_in, n_out, kv, ntiles;
  int mirror;              // table row r pairs with weight slice kv - 1 - r (SubM)
};

// C, K in {16, 32}.  256 threads: wave w owns input rows [32 w, 32 w + 32) of a tile for dgrad and output
// tile w of the K x C weight gradient (k' block = w / (C / 16), c block = w % (C / 16)) for wgrad.
//
// Latency.  A step is ~0.3 us of issue work and two dependent trips to memory (table word -> dout row); with
// 108 accumulator registers only two workgroups fit a CU, so the trips are hidden by DEPTH instead of by
// occupancy: a wave keeps the dout rows of the next kRows steps and the table words of the step after those in
// flight (shift registers: the ring position of a step is not a compile-time constant, the table row -- the
// accumulator index -- is), and the look-ahead runs across tile boundaries: the masks of all tiles of the
// workgroup are reduced up front, the next tile's feat rows travel in registers during the current tile.
constexpr int kBnRows = 2;        // steps whose dout rows are in flight
constexpr int kBnMaxTiles = 64;   // tiles per persistent workgroup

// W8: 512-thread workgroups -- a wave owns 16 rows for dgrad, and the four weight-gradient tiles are shared by wave
// PAIRS that split the table rows by parity (wave w < 4: even rows, w >= 4: odd rows: in a pair step the two
// halves of the workgroup read one stage each), 14 accumulator tiles per wave instead of 27: ~128 registers per
// lane, four waves per SIMD instead of two.
template <bool BF16, int C, int K, bool W8>
__global__ void __launch_bounds__(W8 ? 512 : 256, W8 ? 4 : 2)
bwdn_kernel(BwdnParams p) {
  constexpr int NT = W8 ? 512 : 256;             // threads
  constexpr int MB = W8 ? 1 : 2;                 // 16-row m-blocks per wave
  constexpr int RPW = MB * 16;                   // dgrad rows per wave
  constexpr int NACC = W8 ? (kBnMaxKv + 1) / 2 : kBnMaxKv;
  constexpr int ROWS = (W8 && C == 32) ? 1 : kBnRows;      // ring depth (the 128-register form of C = 32 affords one)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *sF = smem;                               // the tile's feat rows
  char *sG = smem + kBnStage;                    // 2 x 2 stages of gathered dout rows (a pair of table rows per step)
  __shared__ uint32_t s_tmask[kBnMaxTiles + 1];
  constexpr int NBC = C / 16, NBK = K / 16;      // 16-channel blocks
  constexpr int SLK = K / 8, SLC = C / 8;        // 16-byte slots per row
  constexpr int FPT = (kBnTile * SLC + NT - 1) / NT;   // 16-byte pieces of the feat tile per thread (1 or 2)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lgrp = lane >> 4;
  const uint32_t rowD = K * 2u, rowF = C * 2u;
  const __amdgpu_buffer_rsrc_t rD = make_rsrc(p.dout, static_cast<uint32_t>(p.n_out) * rowD);
  const __amdgpu_buffer_rsrc_t rF = make_rsrc(p.feat, static_cast<uint32_t>(p.n_in) * rowF);
  const __amdgpu_buffer_rsrc_t rT = make_rsrc(p.table, static_cast<uint32_t>(p.kv) * static_cast<uint32_t>(p.n_in) * 4u);
  const __amdgpu_buffer_rsrc_t rM = make_rsrc(p.mask, static_cast<uint32_t>(p.n_in) * 4u);
  const __amdgpu_buffer_rsrc_t rW = make_rsrc(p.wt, static_cast<uint32_t>(p.kv) * C * K * 2u);
  const int wt = W8 ? (wave & 3) : wave, par = W8 ? (wave >> 2) : 0;
  const bool wg_live = wt < NBC * NBK;           // this wave owns (a parity class of) a weight-gradient tile
  const int kb = wt / NBC, cb = wt % NBC;
  // lanes whose 16-byte piece lies beyond a 16-channel row never load (their fragment half is zero)
  const uint32_t dcol = lgrp < SLK ? static_cast<uint32_t>(lgrp) * 16u : kOob;
  const uint32_t kvmask = p.kv >= 32 ? 0xffffffffu : ((1u << p.kv) - 1u);
  const int G = gridDim.x;
  const int my_tiles = blockIdx.x < p.ntiles ? (p.ntiles - 1 - blockIdx.x) / G + 1 : 0;     // tiles b, b + G, ...

  // ---- masks of all my tiles: thread t < 128 reads row t of every tile (loads in flight together), OR per tile
  for (int j0 = 0; j0 < my_tiles; j0 += 8) {
    uint32_t mw[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = (blockIdx.x + (j0 + q) * G) * kBnTile + (tid & 127);
      mw[q] = (j0 + q < my_tiles && row < p.n_in && tid < kBnTile)
                  ? __builtin_amdgcn_raw_buffer_load_b32(rM, static_cast<uint32_t>(row) * 4u, 0, 0) : 0u;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) mw[q] |= __shfl_xor(mw[q], d, 64);
    }
    if (tid == 0)
#pragma unroll
      for (int q = 0; q < 8; ++q) if (j0 + q < my_tiles) s_tmask[j0 + q] = 0u;
    __syncthreads();
    if (lane == 0 && wave < 2)
#pragma unroll
      for (int q = 0; q < 8; ++q) if (j0 + q < my_tiles) atomicOr(&s_tmask[j0 + q], mw[q] & kvmask);
    __syncthreads();
  }

  f32x4 acc_w[NACC];
#pragma unroll
  for (int r = 0; r < NACC; ++r) acc_w[r] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- look-ahead over the (tile, table row) items of this workgroup, in processing order (all wave-uniform)
  int la_j = -1;
  uint32_t la_bits = 0;
  auto la_next = [&](int &tile_of, int &r_of) __attribute__((always_inline)) {
    while (la_bits == 0u && la_j + 1 < my_tiles) {
      ++la_j;
      la_bits = __builtin_amdgcn_readfirstlane(s_tmask[la_j]);
    }
    if (la_bits == 0u) {
      tile_of = -1;
      r_of = 0;
      return;
    }
    tile_of = blockIdx.x + la_j * G;
    r_of = __builtin_ctz(la_bits);
    la_bits &= la_bits - 1u;
  };
  auto load_idx = [&](int tile_of, int r, int (&ix)[MB]) __attribute__((always_inline)) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int row = tile_of * kBnTile + wave * RPW + mb * 16 + lrow;
      const uint32_t off = (tile_of >= 0 && row < p.n_in) ? (static_cast<uint32_t>(r) * p.n_in + row) * 4u : kOob;
      ix[mb] = static_cast<int>(__builtin_amdgcn_raw_buffer_load_b32(rT, off, 0, SPX_AUX_TABLE));
    }
  };
  auto load_rows = [&](const int (&ix)[MB], uint4 (&g)[MB]) __attribute__((always_inline)) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      // (a table word that was never loaded -- past the end -- reads as 0: row 0, harmless and never used)
      const uint32_t off = ix[mb] >= 0 ? (static_cast<uint32_t>(ix[mb]) * rowD + dcol) | (dcol & kOob) : kOob;
      g[mb] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rD, off, 0, 0));
    }
  };
  // weight fragments of table row r: slice r, or kv - 1 - r with the mirrored weight order of SubM
  auto load_w = [&](int r, uint4 (&w)[NBC]) __attribute__((always_inline)) {
    const int kw = p.mirror ? p.kv - 1 - r : r;
#pragma unroll
    for (int nb = 0; nb < NBC; ++nb) {
      const uint32_t off = (static_cast<uint32_t>(kw) * C + nb * 16 + lrow) * rowD;
      w[nb] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rW, (off + dcol) | (dcol & kOob), 0, 0));
    }
  };
  auto load_feat = [&](int tile_of, uint4 (&f)[FPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < FPT; ++q) {
      const int pc = tid + q * NT, r = pc / SLC, sl = pc % SLC;
      const int row = tile_of * kBnTile + r;
      const uint32_t off = (tile_of >= 0 && row < p.n_in && pc < kBnTile * SLC) ? static_cast<uint32_t>(row) * rowF + sl * 16u : kOob;
      f[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rF, off, 0, 0));
    }
  };

  // pipeline: g[j] / w[j] = dout rows / weight fragments of the item j steps ahead (j < kBnRows),
  // ixn = table words of the item kBnRows steps ahead
  uint4 g[ROWS][MB], w[ROWS][NBC];
  int ixn[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) ixn[mb] = -1;
  int la_t_pending = -1, la_r_pending = 0;        // the item whose table words are in `ixn`
  {
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      int t_of, r_of, ix[MB];
      la_next(t_of, r_of);
      load_idx(t_of, r_of, ix);
      load_rows(ix, g[j]);                        // (prologue: waits for the words, once per workgroup)
      load_w(r_of, w[j]);
    }
    la_next(la_t_pending, la_r_pending);
    load_idx(la_t_pending, la_r_pending, ixn);
  }
  uint4 fpre[FPT];
  load_feat(my_tiles > 0 ? static_cast<int>(blockIdx.x) : -1, fpre);

  int stage = 0;
  for (int j = 0; j < my_tiles; ++j) {
    const int tile = blockIdx.x + j * G, base = tile * kBnTile;
    const uint32_t tmask = __builtin_amdgcn_readfirstlane(s_tmask[j]);
    __syncthreads();                             // the previous tile's stages are no longer read
#pragma unroll
    for (int q = 0; q < FPT; ++q) {
      const int pc = tid + q * NT;
      if (pc < kBnTile * SLC) *reinterpret_cast<uint4 *>(sF + ctr_slot(pc / SLC, pc % SLC)) = fpre[q];
    }
    load_feat(j + 1 < my_tiles ? tile + G : -1, fpre);      // the next tile's rows: in flight during this tile

    f32x4 acc_d[MB][NBC];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBC; ++nb) acc_d[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the tile's feat fragments of this wave's weight-gradient tile: the same for every table row, read once
    __syncthreads();
    // (the 512-thread form has no registers to spare for them: it reads them per step)
    uint4 ffr[W8 ? 1 : kBnTile / 32];
    if (wg_live && !W8)
#pragma unroll
      for (int ks = 0; ks < kBnTile / 32; ++ks) ffr[ks] = ctr_frag(sF, ks * 32 + lgrp * 8, lrow, cb);

    // Table rows go in PAIRS (2 q, 2 q + 1): both members' gathered tiles are staged before ONE barrier and both
    // weight-gradient products follow it -- a step is a chain LDS write -> barrier -> transposed read -> MFMA that
    // only two resident workgroups per CU can overlap, so the number of barriers is what the walk costs.
    static_for<0, (kBnMaxKv + 1) / 2>([&](auto qc) __attribute__((always_inline)) {
      constexpr int r0 = 2 * decltype(qc)::value, r1 = r0 + 1;
      const bool a0 = (tmask >> r0) & 1u, a1 = r1 < kBnMaxKv && ((tmask >> r1) & 1u);   // wave-uniform
      if (!a0 && !a1) return;
      char *sg0 = sG + (stage * 2) * kBnStage, *sg1 = sg0 + kBnStage;
      auto consume = [&](char *sg) __attribute__((always_inline)) {
        uint4 g_c[MB], w_c[NBC];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) g_c[mb] = g[0][mb];
#pragma unroll
        for (int nb = 0; nb < NBC; ++nb) w_c[nb] = w[0][nb];
        // shift the ring, refill its tail: rows of the item kBnRows ahead (its table words arrived a step ago),
        // table words of the item after that
#pragma unroll
        for (int q = 0; q + 1 < ROWS; ++q) {
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) g[q][mb] = g[q + 1][mb];
#pragma unroll
          for (int nb = 0; nb < NBC; ++nb) w[q][nb] = w[q + 1][nb];
        }
        load_rows(ixn, g[ROWS - 1]);
        load_w(la_r_pending, w[ROWS - 1]);
        la_next(la_t_pending, la_r_pending);
        load_idx(la_t_pending, la_r_pending, ixn);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          if (lgrp < SLK) *reinterpret_cast<uint4 *>(sg + ctr_slot(wave * RPW + mb * 16 + lrow, lgrp)) = g_c[mb];
        // dgrad: din[rows] += G . W_r   (one 32-deep MFMA step: the reduction index is the dout channel)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NBC; ++nb) acc_d[mb][nb] = mfma16<BF16>(g_c[mb], w_c[nb], acc_d[mb][nb]);
      };
      if (a0) consume(sg0);
      if (a1) consume(sg1);
      __syncthreads();                            // stages complete (the other two were last read one step ago)
      // wgrad: dW_r[k'][c] += G^T . F over the tile's 128 rows (four 32-row MFMA steps per table row)
      if (wg_live) {
        if constexpr (W8) {
          // the two halves of the workgroup take one member of the pair each (accumulator index = the pair)
          constexpr int qi = r0 / 2;
          if (par == 0 ? a0 : a1) {
            const char *sgp = par == 0 ? sg0 : sg1;
#pragma unroll
            for (int ks = 0; ks < kBnTile / 32; ++ks)
              acc_w[qi] = mfma16<BF16>(ctr_frag(sgp, ks * 32 + lgrp * 8, lrow, kb),
                                       ctr_frag(sF, ks * 32 + lgrp * 8, lrow, cb), acc_w[qi]);
          }
        } else {
          if (a0)
#pragma unroll
            for (int ks = 0; ks < kBnTile / 32; ++ks)
              acc_w[r0] = mfma16<BF16>(ctr_frag(sg0, ks * 32 + lgrp * 8, lrow, kb), ffr[ks], acc_w[r0]);
          if constexpr (r1 < kBnMaxKv) {
            if (a1)
#pragma unroll
              for (int ks = 0; ks < kBnTile / 32; ++ks)
                acc_w[r1] = mfma16<BF16>(ctr_frag(sg1, ks * 32 + lgrp * 8, lrow, kb), ffr[ks], acc_w[r1]);
          }
        }
      }
      stage ^= 1;
    });
    // ---- din rows of the tile: D layout (row = 4 (lane >> 4) + e, column = lane & 15) -> 2-byte stores
    if (p.din) {
      uint16_t *din = static_cast<uint16_t *>(p.din);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBC; ++nb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = base + wave * RPW + mb * 16 + lgrp * 4 + e;
            if (row < p.n_in) din[static_cast<size_t>(row) * C + nb * 16 + lrow] = from_float<BF16>(acc_d[mb][nb][e]);
          }
    }
  }
  // ---- partial weight gradient of this workgroup: [kv][K][C] fp32, D layout (k' = 4 (lane >> 4) + e, c = lane & 15)
  if (wg_live) {
    float *dst = p.partial + static_cast<size_t>(blockIdx.x) * p.kv * (K * C);
    static_for<0, NACC>([&](auto rc) __attribute__((always_inline)) {
      constexpr int a = decltype(rc)::value;
      const int r = W8 ? 2 * a + par : a;        // the table row accumulator `a` of this wave belongs to
      if (r < p.kv) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          dst[static_cast<size_t>(r) * (K * C) + (kb * 16 + lgrp * 4 + e) * C + cb * 16 + lrow] = acc_w[a][e];
      }
    });
  }
}

// dw[k'][kw(r)][c] = sum over the workgroups' partials, in a fixed order; kw(r) = kv - 1 - r when mirrored.
// A block owns 64 consecutive elements; its four waves each sum a quarter of the workgroups (independent loads,
// eight in flight), the quarters meet in LDS.
template <bool BF16>
__global__ void __launch_bounds__(kThreads)
bwdn_reduce_kernel(const float *__restrict__ partial, int G, int kv, int K, int C, int mirror,
                   uint16_t *__restrict__ dw) {
  __shared__ float part[kThreads / 64][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6, per = kv * K * C;
  const int e = blockIdx.x * 64 + lane;
  const int g0 = (G * q) / 4, g1 = (G * (q + 1)) / 4;
  float s = 0.f;
  if (e < per) {
    int g = g0;
    for (; g + 8 <= g1; g += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[static_cast<size_t>(g + u) * per + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; g < g1; ++g) s += partial[static_cast<size_t>(g) * per + e];
  }
  part[q][lane] = s;
  __syncthreads();
  if (q == 0 && e < per) {
    const float t = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    const int r = e / (K * C), kk = (e / C) % K, c = e % C;
    const int kw = mirror ? kv - 1 - r : r;
    dw[(static_cast<size_t>(kk) * kv + kw) * C + c] = from_float<BF16>(t);
  }
}

int bwdn_groups(int ntiles) {
  int g = 512;                                        // two persistent workgroups per CU
  if (g * kBnMaxTiles < ntiles) g = div_up(ntiles, kBnMaxTiles);     // (a workgroup keeps <= 64 tile masks in LDS)
  return ntiles < g ? ntiles : g;
}

}  // namespace
}  // namespace spx

using namespace spx;

extern "C" {

size_t spx_igemm_bwd_rows_ws_bytes(int n_in, int C, int K, int kv) {
  const int ntiles = div_up(n_in > 0 ? n_in : 1, kBnTile);
  return align_up(static_cast<size_t>(bwdn_groups(ntiles)) * kv * K * C * sizeof(float), 256) + 256;
}

int spx_igemm_bwd_rows(const void *feat, const void *dout, const void *weight_t, void *din, void *dw,
                       const int32_t *table, const uint32_t *mask, int n_in, int n_out, int C, int K, int kv,
                       int mirror, int dtype, void *ws, size_t ws_bytes, spx_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  SPX_CHECK(dtype == SPX_F16 || dtype == SPX_BF16, "16-bit features only");
  SPX_CHECK((C == 16 || C == 32) && (K == 16 || K == 32), "C, K must be 16 or 32 (got %d, %d)", C, K);
  SPX_CHECK(kv >= 1 && kv <= kBnMaxKv, "kernel volume must be <= %d (got %d)", kBnMaxKv, kv);
  SPX_CHECK(feat && dout && weight_t && dw && table && mask && ws, "null pointer");
  SPX_CHECK(ws_bytes >= spx_igemm_bwd_rows_ws_bytes(n_in, C, K, kv), "workspace too small");
  SPX_CHECK(static_cast<long long>(n_in) * kv * 4 < 2147483647LL && static_cast<long long>(n_out) * K * 2 < 2147483647LL &&
                static_cast<long long>(n_in) * C * 2 < 2147483647LL, "tensor too large for 32-bit buffer offsets");
  const size_t wbytes = static_cast<size_t>(kv) * K * C * 2;
  if (n_in == 0 || n_out == 0) {
    SPX_HIP(hipMemsetAsync(dw, 0, wbytes, s));
    return 0;
  }
  BwdnParams p;
  p.feat = feat;
  p.dout = dout;
  p.wt = weight_t;
  p.din = din;
  p.partial = static_cast<float *>(ws);
  p.table = table;
  p.mask = mask;
  p.n_in = n_in;
  p.n_out = n_out;
  p.kv = kv;
  p.ntiles = div_up(n_in, kBnTile);
  p.mirror = mirror;
  const int G = bwdn_groups(p.ntiles);
  const size_t smem = 5 * static_cast<size_t>(kBnStage);
  // measured on level 1 / 2 of the config-4 network: C = K = 16: 91 us with 8 waves, 99 with 4; C = K = 32: 115 vs 107
  // (the 8-wave form of C = 32 spills and affords a ring depth of one).  All four numbers sit near what the
  // gathered LINES cost -- a 64- (32-) byte row pulls a whole 128-byte line out of the L2 -- which is why neither
  // occupancy nor a deeper ring, fewer barriers or fewer LDS reads moved them.
  // 8 waves at C = K = 16 only; the 8-wave forms of the 32-channel shapes spilled (12-16 bytes of scratch per lane)
  // and are no longer instantiated: every instantiation of this kernel runs without scratch
  count_launch(kFamBwdRows);
#define SPX_BWDN(BF, CC, KK, W8)                                                                                \
  hipLaunchKernelGGL((bwdn_kernel<BF, CC, KK, W8>), dim3(G), dim3(W8 ? 512 : 256), smem, s, p)
  const bool bf = dtype == SPX_BF16;
  if (C == 16 && K == 16) { if (bf) SPX_BWDN(true, 16, 16, true); else SPX_BWDN(false, 16, 16, true); }
  else if (C == 16 && K == 32) { if (bf) SPX_BWDN(true, 16, 32, false); else SPX_BWDN(false, 16, 32, false); }
  else if (C == 32 && K == 16) { if (bf) SPX_BWDN(true, 32, 16, false); else SPX_BWDN(false, 32, 16, false); }
  else { if (bf) SPX_BWDN(true, 32, 32, false); else SPX_BWDN(false, 32, 32, false); }
#undef SPX_BWDN
  const int per = kv * K * C;
  if (bf) hipLaunchKernelGGL(bwdn_reduce_kernel<true>, dim3(div_up(per, 64)), dim3(kThreads), 0, s, p.partial, G,
                             kv, K, C, mirror, static_cast<uint16_t *>(dw));
  else hipLaunchKernelGGL(bwdn_reduce_kernel<false>, dim3(div_up(per, 64)), dim3(kThreads), 0, s, p.partial, G,
                          kv, K, C, mirror, static_cast<uint16_t *>(dw));
  SPX_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
