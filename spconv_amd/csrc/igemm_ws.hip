// Weight-stationary gather-GEMM for DENSE neighbourhoods (gfx950): forward and dgrad of a sparse convolution whose
// rows meet most of the kernel offsets (real LiDAR: 6-16 pairs per voxel; the rows-layout class word 0), C = K = 64.
//
// Same contract and the same arithmetic as igemm_v4_kernel (igemm.hip): output-stationary implicit GEMM over the
// [kv, n_dst] pair table, every output row written exactly once, no atomics, v_mfma_f32_16x16x32_{f16,bf16} with the
// gathered rows fed straight from VGPRs (a missing pair is an out-of-range buffer offset -> zeros), per-row
// accumulation order "identity offset first, then ascending offsets" -- results are BIT-IDENTICAL to v4's, so which
// of the two kernels a launch takes never shows in a result.  What differs is the weight side, which is what bounds
// v4 on dense data (one 8 KB slice per step through global -> registers -> LDS, a workgroup barrier per step, four
// waves in lock-step: 980 tiles x 221 KB = 217 MB of weight traffic per launch on the reference's LiDAR fixture):
//
//  * a workgroup is 16 waves x 32 rows = 512 rows, ONE per CU; the weights come G = 9 slices (72 KB) at a time -- a
//    PHASE -- into a double-buffered LDS stage through LDS-DMA (buffer_load_dwordx4 ... lds): the slices of phase
//    p + 1 travel global -> LDS without registers and without the VALU while the waves work through phase p, so a
//    phase boundary is ONE barrier and a launch moves 256 x 221 KB = 57 MB of weights;
//  * between two boundaries the waves run FREE: each wave walks the offsets ITS 32 rows have (the set bits of its own
//    mask OR), with its own register pipeline (gathered rows D steps ahead, pair words D + 1), no barrier, no weight
//    loads; all eight weight fragments of a step are read before its sixteen MFMAs (read by read a step was a chain of
//    eight LDS round trips: ~1300 clocks per wave and step whatever else the SIMD had to do);
//  * an LDS-DMA writes lane-linear (dest = M0 base + 16 * lane), so the XOR swizzle of the stage is applied to the
//    SOURCE address of each lane.  dgrad reads the KRSC tensor in place as well: its image is [reduction row][channel]
//    as it lies in memory and the fragments come out through ds_read_b64_tr_b16 (layout and read of wgrad's operand
//    stages, wtr_frag in igemm.hip) -- no transposing ds_write_b32 pass.  MFMA row i of channel block nb is then
//    channel nb * 16 + i (forward: the permuted order of igemm_v4_kernel), which only changes which 8 bytes of a row a
//    lane stores.
//
// The DMA instructions are inline asm (behind a pending LDS-DMA builtin the compiler orders every ds_read with
// vmcnt(0), which would drain the row pipeline at every fragment read), so the compiler does not count them:
//  * they are issued right behind a workgroup barrier, when the only loads of this wave still in flight are OLDER row /
//    pair-word loads: the compiler's counted waits for those are then too strict by the number of DMA instructions
//    (safe), and exact again for everything issued later;
//  * phase p + 1's slices are complete for a wave once it has waited for any row load issued after them, i.e. after
//    D + 1 steps of phase p; a wave with fewer steps drains its queue before the boundary barrier (it would wait there
//    anyway); behind the barrier every wave's share has landed (cdna_hip_programming.md 5.7 / 6: counted vmcnt, then
//    a barrier, then the ds_read).
//
// Measured on the reference's LiDAR fixture (125 562 voxels, 788 888 pairs, fp16; tools/ws_probe.py,
// profiles/r05_experiments.md): forward 42.4 -> 31.7 us, dgrad 40.0 -> 35.9 us.  A step is now MFMA-bound (four waves
// per SIMD share one matrix pipe: 16 MFMAs x 16 clocks x 4 = 1024 clocks, measured ~1170), 77 % of it on absent pairs.
//
// Limits (the dispatcher keeps v4 otherwise): 16-bit operands, C_in = C_out = 64, kernel volume <= 32, 32-bit buffer
// offsets.  Reference kernels this stands in for: the mask-skipping implicit-GEMM forward / input-gradient kernels of
// spconv/csrc/sparse/convops.py:1363-1446 (multi-stage smem pipeline, core.py:542).
#include "igemm_defs.h"

namespace spx {
namespace {

struct WsArgs {
  const void *A;            // [n_src, 64] gathered operand
  const void *B;            // weights, element (k, n, c) at k*strideK + n*strideN + c*strideD
  void *out;                // [n_dst, 64]
  const int32_t *pair;      // [kv, n_dst]
  const uint32_t *mask;     // [n_dst] or null
  const int32_t *argsort;   // [n_dst] (tables in tile order) or null
  const void *bias;
  long long strideK, strideN, strideD;
  int n_dst, n_src, kv, identity_k, b_reverse, act, ntiles;
  float act_alpha;
  float *stats;             // per-workgroup BatchNorm statistics of the stored rows (GemmParams::stats), or null
  const int32_t *n_live;
  unsigned long long *tl;   // per-wave timeline (tools/ws_probe.py, WS_TL=1) or null: 8 s_memtime stamps per wave
};

// workgroup barrier that leaves vector-memory loads in flight (__syncthreads() drains them)
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A wave's walk: positions of a SEQUENCE of the kernel offsets -- the identity offset first, then the others in
// ascending order (igemm_v4_kernel's accumulation order) -- that its rows use.
struct WsStep {
  int j;           // position in the sequence, -1 = end
  uint32_t rest;   // positions after j
};
__device__ __forceinline__ WsStep ws_first(uint32_t bits) {
  WsStep s;
  s.j = bits ? __builtin_ctz(bits) : -1;
  s.rest = bits ? (bits & (bits - 1)) : 0u;
  return s;
}
__device__ __forceinline__ WsStep ws_next(WsStep s) { return ws_first(s.rest); }

__device__ __forceinline__ int ws_trx(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

typedef short ws_s16x4 __attribute__((ext_vector_type(4)));
// 8 consecutive reduction rows (row0 .. row0 + 7 as seen by this lane group) of channel gran * 16 + lrow
__device__ __forceinline__ uint4 ws_trfrag(const char *stage, int row0, int lrow, int gran) {
  const int r = row0 + (lrow >> 2);
  const char *a0 = stage + r * 128 + ((gran ^ ws_trx(r)) << 5) + ((lrow & 3) << 3);
  const int r1 = r + 4;
  const char *a1 = stage + r1 * 128 + ((gran ^ ws_trx(r1)) << 5) + ((lrow & 3) << 3);
  typedef __attribute__((address_space(3))) ws_s16x4 lds_s16x4;
  const ws_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(__attribute__((address_space(3))) char *)a0);
  const ws_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(__attribute__((address_space(3))) char *)a1);
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// one LDS-DMA instruction: lane L fetches 16 bytes at byte offset vo of the buffer, the wave's 64 pieces land at LDS
// byte address lds_dst + 16 L.  M0 is written in the statement that reads it, and restored.
__device__ __forceinline__ void ws_dma16(u32x4 rsrc, uint32_t lds_dst, uint32_t vo) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(vo), "s"(lds_dst), "s"(rsrc)
               : "memory");
}

#define SPX_WS_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

template <int NW, int G, int D, bool BF16, bool BT>
__global__ void __launch_bounds__(NW * 64, NW / 4)
igemm_ws_kernel(WsArgs p) {
  static_assert(D >= 2 && D <= 3 && NW % 4 == 0, "shape");
  constexpr int COUT = 64, MB = 2, NB = COUT / 16, CPL = NB * 4, TM = NW * MB * 16, NKS = 2;
  constexpr int B_BYTES = 64 * kRowBytes;                   // one slice: [64 rows][128 B] in either direction
  constexpr int BUF_BYTES = G * B_BYTES;
  constexpr int NDMA = G * 8;                               // DMA instructions (1 KB each) per phase and workgroup
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int tile = xcd_tile(static_cast<int>(blockIdx.x), p.ntiles);
  const int kv = p.kv;
  const uint32_t kvbits = kv >= 32 ? 0xffffffffu : ((1u << kv) - 1u);
  const int nph = (kv + G - 1) / G;
  const bool spec = p.identity_k >= 0;    // SubM: the identity offset exists for every row
  const int ik = spec ? p.identity_k : 0;
  // offset at position j of the sequence
  auto koff = [&](int j) __attribute__((always_inline)) { return !spec ? j : (j == 0 ? ik : (j <= ik ? j - 1 : j)); };
  auto stamp = [&](int i) __attribute__((always_inline)) {
    if (p.tl && lane == 0) p.tl[(static_cast<size_t>(blockIdx.x) * NW + wave) * 8 + i] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  auto swzB = [](int row, int sl) __attribute__((always_inline)) {     // (swizzle of the stage: as igemm_v4_kernel)
    const int x = ((row >> 1) & 1) | (((row / CPL) & 3) << 1);
    return row * kRowBytes + ((sl ^ x) << 4);
  };

  // ---- weight phases through LDS-DMA ------------------------------------------------------------------------------
  const uint32_t w_bytes = static_cast<uint32_t>(COUT) * kv * kRowBytes;
  const unsigned long long wa = reinterpret_cast<unsigned long long>(p.B);
  const u32x4 rsW = {static_cast<uint32_t>(wa), static_cast<uint32_t>(wa >> 32) & 0xffffu, w_bytes,
                     static_cast<uint32_t>(kRsrcFlags)};
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) char *)smem));
  // source offset (inside one slice) of the piece lane L of DMA block b delivers: piece q = 64 b + L of the image = row
  // q / 8, PHYSICAL slot q % 8 -> the logical slot the swizzle maps there.  (Computed where it is used: an array
  // indexed by the wave number would live in scratch.)
  auto src_off = [&](int b) __attribute__((always_inline)) {
    const int q = b * 64 + lane, rr = q >> 3, ps = q & 7;
    if constexpr (!BT) {
      const int x = ((rr >> 1) & 1) | (((rr / CPL) & 3) << 1);
      return static_cast<uint32_t>(rr) * static_cast<uint32_t>(p.strideN) * 2u + ((ps ^ x) << 4);
    } else {
      const int ls = (((ps >> 1) ^ ws_trx(rr)) << 1) | (ps & 1);
      return static_cast<uint32_t>(rr) * static_cast<uint32_t>(p.strideD) * 2u + (ls << 4);
    }
  };
  // this wave's share of phase ph -> buffer buf: instructions i = wave, wave + NW, ... (slice i / 8, block i % 8)
  auto dma_phase = [&](int ph, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < (NDMA + NW - 1) / NW; ++u) {
      const int i = wave + u * NW;
      if (i < NDMA) {
        const int sl = i >> 3, b = i & 7;
        const int j = ph * G + sl;
        const int k = koff(j);
        const int kb = p.b_reverse ? kv - 1 - k : k;
        const uint32_t so = static_cast<uint32_t>(kb) * static_cast<uint32_t>(p.strideK) * 2u;
        ws_dma16(rsW, lds0 + buf * BUF_BYTES + sl * B_BYTES + b * 1024, j < kv ? src_off(b) + so : kOob);
      }
    }
  };
  dma_phase(0, 0);

  // ---- rows of this lane --------------------------------------------------------------------------------------------
  const uint32_t tbl_bytes = static_cast<uint32_t>(p.n_dst) * 4u;
  const __amdgpu_buffer_rsrc_t rO = make_rsrc(p.argsort, p.argsort ? tbl_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rM = make_rsrc(p.mask, p.mask ? tbl_bytes : 0u);
  int pos[MB], glist[MB], grow[MB];
  uint32_t goff[MB], mraw[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int t = tile * TM + (wave * MB + mb) * 16 + lrow;
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
  stamp(1);                 // mask words arrived

  // ---- gathered-operand pipeline ----------------------------------------------------------------------------------
  const uint32_t rowB = kRowBytes;
  const uint32_t a_bytes = static_cast<uint32_t>(p.n_src) * rowB;
  uint32_t aoff[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) aoff[ks] = static_cast<uint32_t>(ks * 64 + lgrp * 16);
  int idxr[D][MB];
  uint32_t identr[D];       // wave-uniform: idxr[S] stands for the identity offset
  u32x4 areg[D][MB][NKS];
  WsStep it[D + 2];
  it[0] = ws_first(seqmask);
#pragma unroll
  for (int j = 1; j < D + 2; ++j) it[j] = ws_next(it[j - 1]);
  // Straight-line (no branch around a load): the compiler's counted waits stay exact.  A step that does not exist
  // (j < 0) reads through a zero-sized resource: nothing is fetched.
  auto load_idx = [&](const WsStep &s, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const int k = s.j < 0 ? 0 : koff(s.j);
    const __amdgpu_buffer_rsrc_t rP =
        make_rsrc(p.pair + static_cast<size_t>(k) * p.n_dst, s.j >= 0 ? tbl_bytes : 0u);
    identr[S] = (spec && s.j == 0) ? 0xffffffffu : 0u;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      idxr[S][mb] = static_cast<int>(__builtin_amdgcn_raw_buffer_load_b32(rP, goff[mb], 0, 0));
  };
  auto load_a = [&](const WsStep &s, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const __amdgpu_buffer_rsrc_t r = make_rsrc(p.A, s.j >= 0 ? a_bytes : 0u);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const uint32_t idx = (static_cast<uint32_t>(grow[mb]) & identr[S]) |
                           (static_cast<uint32_t>(idxr[S][mb]) & ~identr[S]);
      const uint32_t rbase = idx * rowB;                         // -1 -> >= kOob
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        areg[S][mb][ks] = __builtin_amdgcn_raw_buffer_load_b128(r, min(rbase + aoff[ks], kOob), 0, 0);
    }
  };
  {
    // pair words of steps 0 .. D, rows of steps 0 .. D - 1 (nothing here depends on the LDS)
    auto pro_idx = [&](auto J) __attribute__((always_inline)) { load_idx(it[decltype(J)::value], J); };
    auto pro_a = [&](auto J) __attribute__((always_inline)) { load_a(it[decltype(J)::value], J); };
    pro_idx(std::integral_constant<int, 0>{});
    pro_idx(std::integral_constant<int, 1>{});
    if constexpr (D > 2) pro_idx(std::integral_constant<int, 2>{});
    __builtin_amdgcn_sched_barrier(0);
    pro_a(std::integral_constant<int, 0>{});
    pro_a(std::integral_constant<int, 1>{});
    if constexpr (D > 2) pro_a(std::integral_constant<int, 2>{});
    __builtin_amdgcn_sched_barrier(0);
    load_idx(it[D], std::integral_constant<int, 0>{});          // pair words of step D -> set 0 (consumed above)
  }
  // phase 0's slices have landed once at most the loads issued behind them are outstanding (2 MB list / mask words,
  // (D + 1) MB pair words, D MB NKS rows): the row pipeline stays in flight across the barrier
  __builtin_amdgcn_sched_barrier(0);
  static_assert(2 * MB + (D + 1) * MB + D * MB * NKS == (D == 2 ? 18 : 24), "count the prologue loads");
  if constexpr (D == 2) SPX_WS_VMCNT(18);
  else SPX_WS_VMCNT(24);
  ws_barrier();
  int ph = 0, nsteps = 0;
  if (nph > 1) dma_phase(1, 1);
  int phase_end = min(kv, G);
  stamp(2);                 // phase 0 ready
  int nadv = 0;

  // Boundary to the next phase, whose slices were requested a phase ago.  Every wave of the workgroup executes it
  // exactly nph - 1 times: from wherever its own walk crosses a boundary, or from the tail loop below.
  auto advance = [&]() __attribute__((always_inline)) {
    ++ph;
    if (ph < nph) {
      if (nadv == 0) stamp(3);   // this wave's steps of the first phase are done
      if (nsteps < D + 1) SPX_WS_VMCNT(0);         // (too few waits behind the DMA to know that it has landed)
      ws_barrier();              // every wave's share of phase ph has landed; nobody reads buffer (ph + 1) & 1 any more
      if (nadv == 0) stamp(4);   // ... and everybody else's
      if (ph + 1 < nph) dma_phase(ph + 1, (ph + 1) & 1);
      if (nadv == 0) stamp(5);
      ++nadv;
    }
    nsteps = 0;
    phase_end = ph < nph ? min(kv, (ph + 1) * G) : 0x7fffffff;
  };

  f32x4 acc[NB][MB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const WsStep &s, auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    if (s.j >= 0) {
      const char *cur = smem + (ph & 1) * BUF_BYTES + (s.j - ph * G) * B_BYTES;
      ++nsteps;
      uint4 fa[NKS][NB];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          if constexpr (!BT)
            fa[ks][nb] = *reinterpret_cast<const uint4 *>(cur + swzB((lrow >> 2) * CPL + nb * 4 + (lrow & 3), ks * 4 + lgrp));
          else
            fa[ks][nb] = ws_trfrag(cur, ks * 32 + lgrp * 8, lrow, nb);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            acc[nb][mb] = mfma16<BF16>(fa[ks][nb], __builtin_bit_cast(uint4, areg[S][mb][ks]), acc[nb][mb]);
    }
  };
  // one step: at step t (S = t % D) areg[S] = rows of step t, idxr[S] = pair words of step t + D (requested one step
  // ago), it[i] = step t + i
  auto step = [&](auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    while (it[0].j >= phase_end) advance();
    compute(it[0], SET);
    __builtin_amdgcn_sched_barrier(0);
    load_idx(it[D + 1], std::integral_constant<int, (S + 1) % D>{});
    __builtin_amdgcn_sched_barrier(0);
    load_a(it[D], SET);
#pragma unroll
    for (int i = 0; i < D + 1; ++i) it[i] = it[i + 1];
    it[D + 1] = ws_next(it[D]);
  };
  while (it[0].j >= 0) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});     // may be a step past the end (no MFMAs, zero-sized loads)
    if constexpr (D > 2) step(std::integral_constant<int, 2>{});
  }
  stamp(6);                            // this wave's walk is done
  while (ph < nph - 1) advance();      // boundaries this wave's rows do not reach: their barriers still count

  // ---- epilogue: straight from registers; rows past the end have an out-of-range offset ----------------------------
  const bool plain = p.bias == nullptr && p.act == SPX_ACT_NONE;
  const __amdgpu_buffer_rsrc_t rOut = make_rsrc(p.out, static_cast<uint32_t>(p.n_dst) * (COUT * 2));
  float bv[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) bv[q] = 0.f;
  if (p.bias) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const int ch = BT ? (q >> 2) * 16 + lgrp * 4 + (q & 3) : lgrp * CPL + q;
      bv[q] = to_float<BF16>(static_cast<const uint16_t *>(p.bias)[ch]);
    }
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
    if constexpr (!BT) {
      // CPL consecutive channels per lane (the channel permutation of igemm_v4_kernel)
      store_dwords<CPL / 2, SPX_AUX_OUT>(d, rOut, rb == kOob ? kOob : rb + lgrp * (CPL * 2));
    } else {
      // natural channel order (the transpose read ties MFMA row i to channel nb * 16 + i): 4 channels per lane and block
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const uint32_t dd[2] = {d[nb * 2], d[nb * 2 + 1]};
        store_dwords<2, SPX_AUX_OUT>(dd, rOut, rb == kOob ? kOob : rb + (nb * 16 + lgrp * 4) * 2);
      }
    }
  }
  stamp(7);                            // stores issued
  if constexpr (!BT) {
    if (p.stats) {                     // (as igemm_v4_body: statistics of the rounded rows this workgroup stores)
      const int st_live = p.n_live ? *p.n_live : 0x7fffffff;
      __syncthreads();                 // every wave has left its last phase: the weight buffers are free
      wg_bn_stats<COUT, CPL, MB, NW, (BF16 ? 2 : 1)>(acc, grow, st_live, reinterpret_cast<float *>(smem), p.stats,
                                                     static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
    }
  }
}

template <int NW, int G, int D, bool BF16, bool BT>
int launch_ws_one(const WsArgs &a, hipStream_t s, int *grid_out) {
  constexpr size_t lds = 2 * static_cast<size_t>(G) * 64 * kRowBytes;
  auto kern = igemm_ws_kernel<NW, G, D, BF16, BT>;
  static std::atomic<uint64_t> attr_done{0};      // one bit per device (common.h: ensure_dynamic_lds)
  SPX_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), static_cast<int>(lds), attr_done));
  WsArgs q = a;
  q.ntiles = div_up(a.n_dst, NW * 32);
  count_launch(kFamWs);
  hipLaunchKernelGGL(kern, dim3(q.ntiles), dim3(NW * 64), lds, s, q);
  if (grid_out) *grid_out = q.stats ? q.ntiles : 0;
  SPX_LAUNCH_CHECK();
  return 0;
}

unsigned long long *g_ws_timeline = nullptr;     // (set through spx_debug_ws_timeline)

}  // namespace

bool ws_ok(const GemmParams &p, int dtype) {
  if (dtype != SPX_F16 && dtype != SPX_BF16) return false;
  if (p.CIN != 64 || p.COUT != 64 || p.kv > 32 || p.kv < 1 || !p.pair) return false;
  if (p.acc_mode || p.kbase || p.mask_words > 1) return false;
  if (p.tile_order == 0 && p.argsort && !p.cls) return false;          // (listed rows over row-order tables: v4)
  const unsigned long long abytes = static_cast<unsigned long long>(p.n_src) * p.CIN * 2;
  const unsigned long long obytes = static_cast<unsigned long long>(p.n_dst) * p.COUT * 2;
  const unsigned long long pbytes = static_cast<unsigned long long>(p.n_dst) * 4ull;
  const unsigned long long wbytes = static_cast<unsigned long long>(p.COUT) * p.kv * p.CIN * 2;
  if (abytes >= 0x7fff0000ull || obytes >= 0x7fff0000ull || wbytes >= 0x7fff0000ull || pbytes >= 0x7fff0000ull)
    return false;
  if (p.strideD != 1 && p.strideN != 1) return false;                  // forward: rows of W contiguous; dgrad: columns
  return true;
}

int launch_gather_gemm_ws(const GemmParams &p, int dtype, hipStream_t s) {
  WsArgs a{};
  a.tl = g_ws_timeline;
  a.A = p.A;
  a.B = p.B;
  a.out = p.out;
  a.pair = p.pair;
  a.mask = p.mask;
  a.argsort = p.tile_order == 1 ? p.argsort : nullptr;
  a.bias = p.bias;
  a.strideK = p.strideK;
  a.strideN = p.strideN;
  a.strideD = p.strideD;
  a.n_dst = p.n_dst;
  a.n_src = p.n_src;
  a.kv = p.kv;
  a.identity_k = p.identity_k;
  a.b_reverse = p.b_reverse;
  a.act = p.act;
  a.act_alpha = p.act_alpha;
  const bool bt = p.strideD != 1;
  const bool bf = dtype == SPX_BF16;
  a.stats = bt ? nullptr : p.stats;
  a.n_live = p.n_live;
  if (bt) return bf ? launch_ws_one<16, 9, 2, true, true>(a, s, p.grid_out)
                    : launch_ws_one<16, 9, 2, false, true>(a, s, p.grid_out);
  return bf ? launch_ws_one<16, 9, 2, true, false>(a, s, p.grid_out)
            : launch_ws_one<16, 9, 2, false, false>(a, s, p.grid_out);
}

}  // namespace spx

extern "C" int spx_debug_ws_timeline(void *buf) {
  spx::g_ws_timeline = static_cast<unsigned long long *>(buf);
  return 0;
}
