// One launch for the backward of a layer (igemm_bwd_kernel: dgrad tiles + balanced wgrad ranges) and the range bodies of
// the weight gradient -- template code shared by igemm.hip (f16), igemm_bf16.hip and igemm_f32.hip.
#pragma once
#include "igemm_v4.h"

namespace spx {
namespace {

constexpr int kWT = 64;    // dW tile edge

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

// Backward of one layer in ONE launch: the wgrad ranges (the longer, streaming workgroups)
// are dispatched first, the dgrad tiles after them.  The two halves only share read-only inputs; at ~100k voxels each of them
// is latency-bound with idle issue slots and idle HBM bandwidth, so running them side by side
// on the same CUs costs little more than the slower one -- and one kernel boundary (~1.7 us)
// disappears.  (Two HIP streams were tried first: the fork/join costs more than it buys.)
template <int COUT, int MB, int DT, int NKS = 2, int PK = 1>
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
    igemm_v4_body<COUT, MB, DT, true, NKS, PK>(p, n_dgrad < 0 ? b - nw : b);
  } else {
    if constexpr (DT == 3) wgrad_f32_body(wp, n_dgrad < 0 ? b : b - n_dgrad);
    else wgrad_tr_body<DT == 1, 1>(wp, n_dgrad < 0 ? b : b - n_dgrad);
  }
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
  count_launch(kFamBwdFused);
#define SPX_LAUNCH_BWD(NKSV, PKV)                                                                                  \
  hipLaunchKernelGGL((igemm_bwd_kernel<COUT, MB, DT, NKSV, PKV>), dim3(n_dgrad + n_wgrad_blocks), dim3(kThreads),  \
                     (bwd_smem_bytes<COUT, MB, DT>()), s, p.A, p.B, p.mask, p.argsort, p.pair, p.n_dst,           \
                     p.n_src, p.CIN, p.kv, p.identity_k, v4_flags(pl), rr,                                        \
                     wgrad_first ? ~n_wgrad_blocks : n_dgrad, q)
  const int pk = v4_pack(p, DT, DT == 3 ? 4 : 2, true);   // dgrad's reduction rows are the dout channels
  if constexpr (DT == 0 || DT == 1) {
    if (pk == 32) SPX_LAUNCH_BWD(2, 32);
    else if (pk == 16) SPX_LAUNCH_BWD(2, 16);
    else if (pk == 8) SPX_LAUNCH_BWD(2, 8);
    else if (pk == 4) SPX_LAUNCH_BWD(1, 4);
    else if (pk == 2) SPX_LAUNCH_BWD(1, 2);
    else if (p.CIN * 2 <= 64) SPX_LAUNCH_BWD(1, 1);     // ... fit half a piece
    else SPX_LAUNCH_BWD(2, 1);
  } else {
    if (p.CIN * (DT == 3 ? 4 : 2) <= 64) SPX_LAUNCH_BWD(1, 1);
    else SPX_LAUNCH_BWD(2, 1);
  }
#undef SPX_LAUNCH_BWD
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
