// First-generation gather-GEMM (LDS-staged gathered rows, 128-row tiles): the path of tensors beyond
// 32-bit buffer offsets (> 2 GB), which the direct-fragment kernels of igemm.hip cannot address
// (bit 31 of an offset is their out-of-range flag).  Split from igemm.hip so that the default path
// compiles and reads alone (round-2 verdict); tests/test_gpu_conv.py::test_tensors_beyond_two_gigabytes
// keeps it honest.
#include "igemm_defs.h"

namespace spx {
namespace {

// BT = false: weight slice rows are [n][reduction] contiguous (forward, KRSC).
// BT = true : the slice is stored [reduction][n] (dgrad reads KRSC directly: reduction = K,
//             n = C); it is transposed on its way into LDS, so no re-laid-out copy of
//             the weights is ever written to memory.
template <int COUT, bool BF16, bool BT>
__global__ void __launch_bounds__(kThreads)
gather_gemm_mfma_kernel(GemmParams p) {
  constexpr int NB = COUT / 16;                       // 16-wide output-channel blocks
  // 16-byte weight vectors staged per thread
  constexpr int BROWS = BT ? 2 * ((COUT + 63) / 64) : (COUT + 31) / 32;
  constexpr int A_BYTES = kTileM * kRowBytes;         // 16 KiB
  constexpr int OUT_ROWB = COUT * 2;
  constexpr int OXM = (COUT / 8 - 1) < 7 ? (COUT / 8 - 1) : 7;  // swizzle stays inside the row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *ldsA = smem;
  char *ldsB = smem + A_BYTES;
  uint32_t *lds_mask = reinterpret_cast<uint32_t *>(smem + A_BYTES + COUT * kRowBytes);  // [4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = (p.n_dst + kTileM - 1) / kTileM;
  const int tile = xcd_tile(blockIdx.x, ntiles);
  const int slot = tid & 7;        // 16-byte slot of a 128-byte row
  const int r0 = tid >> 3;         // 0..31 (weight staging)
  const int rw = lane >> 3;        // 0..7  (row inside the wave's 32-row block)
  const uint16_t *A = static_cast<const uint16_t *>(p.A);
  const uint16_t *B = static_cast<const uint16_t *>(p.B);
  const int nchunk = (p.CIN + kCK - 1) / kCK;

  // Each wave stages the 32 rows it multiplies: tile rows 32*wave + rw + 8*j.
  int grow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = tile * kTileM + wave * 32 + rw + 8 * j;
    int g = -1;
    if (t < p.n_dst) g = p.argsort ? p.argsort[t] : t;
    grow[j] = g;
  }

  uint32_t rmask[4];
  int idxraw[4];                 // raw pair-table words of the step whose data is fetched next
  uint32_t aok = 0;              // bit j: row j of the data in flight is a real row
  uint4 areg[4], breg[BROWS];    // raw loaded vectors; invalid ones are zeroed at the LDS write

  // NOTE on structure: nothing below consumes a loaded value right after its load -- the
  // compiler puts s_waitcnt at the first use, so selects on fresh data would serialise
  // the prefetch.  Validity is applied one step later (aok / rmask), at the LDS write.

  // (1) index fetch for step `it`: straight-line, unconditional loads (an invalid row reads
  // entry 0) so that all four are in flight together.
  auto load_idx = [&](const StepIt &it) __attribute__((always_inline)) {
    if (it.k == p.identity_k) {
#pragma unroll
      for (int j = 0; j < 4; ++j) idxraw[j] = grow[j];
    } else {
      const int32_t *row = p.pair + static_cast<size_t>(it.k) * p.n_dst;
#pragma unroll
      for (int j = 0; j < 4; ++j) idxraw[j] = row[grow[j] < 0 ? 0 : grow[j]];
    }
  };

  // (2) data fetch for step `it` from the rows in idxraw + the weight slice.
  auto load_data = [&](const StepIt &it) __attribute__((always_inline)) {
    const int c0 = it.chunk * kCK;
    const bool cin_ok = c0 + slot * 8 < p.CIN;
    const int coff = cin_ok ? c0 + slot * 8 : 0;
    aok = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = ((rmask[j] >> it.k) & 1u) ? idxraw[j] : -1;
      aok |= (idx >= 0 ? 1u : 0u) << j;
      areg[j] = *reinterpret_cast<const uint4 *>(
          A + static_cast<size_t>(idx < 0 ? 0 : idx) * p.CIN + coff);
    }
    const int kb = p.b_reverse ? p.kv - 1 - it.k : it.k;
    const uint16_t *Bk = B + static_cast<size_t>(kb) * p.strideK;
    if constexpr (!BT) {
#pragma unroll
      for (int j = 0; j < BROWS; ++j) {
        const int n = r0 + 32 * j;
        breg[j] = *reinterpret_cast<const uint4 *>(
            Bk + static_cast<size_t>(n < COUT ? n : 0) * p.strideN + coff);
      }
    } else {
      // vector j: reduction row d = c0 + 2*r0 + (j & 1), n-block (j >> 1)*64 + slot*8 .. +8
#pragma unroll
      for (int j = 0; j < BROWS; ++j) {
        const int d = c0 + 2 * r0 + (j & 1);
        const int n = (j >> 1) * 64 + slot * 8;
        const bool ok = d < p.CIN && n < COUT;
        breg[j] = *reinterpret_cast<const uint4 *>(
            Bk + static_cast<size_t>(ok ? d : 0) * p.strideD + (ok ? n : 0));
      }
    }
  };

  // SubM: the identity offset exists for every valid row, so its data does not depend on the
  // mask words -> start it before the mask loads return.
  const bool spec = p.identity_k >= 0;
  StepIt it0;
  it0.k = p.identity_k;
  it0.chunk = 0;
  it0.rest = 0;
  if (spec) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rmask[j] = grow[j] >= 0 ? 0xffffffffu : 0u;
      idxraw[j] = grow[j];
    }
    load_data(it0);
  }
  uint32_t wm = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t m = p.mask ? p.mask[grow[j] < 0 ? 0 : grow[j]] : 0xffffffffu;
    rmask[j] = grow[j] >= 0 ? m : 0u;
    wm |= rmask[j];
  }
  // OR over the wave's rows: lanes with equal (lane >> 3) hold the same rows
  wm |= __shfl_xor(wm, 8, 64);
  wm |= __shfl_xor(wm, 16, 64);
  wm |= __shfl_xor(wm, 32, 64);
  const uint32_t wavemask = wm;
  if (lane == 0) lds_mask[wave] = wm;
  __syncthreads();
  uint32_t tilemask = lds_mask[0] | lds_mask[1] | lds_mask[2] | lds_mask[3];
  if (p.kv < 32) tilemask &= (1u << p.kv) - 1u;
  if (spec && ((tilemask >> p.identity_k) & 1u)) {
    it0.rest = tilemask & ~(1u << p.identity_k);
  } else {
    it0 = step_begin(tilemask);
    if (it0.k >= 0) {
      load_idx(it0);
      load_data(it0);
    }
  }
  StepIt it1 = step_next(it0, nchunk);
  if (it1.k >= 0) load_idx(it1);

  f32x4 acc[NB][2];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    acc[nb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[nb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  StepIt cur = it0, nxt = it1;
  while (cur.k >= 0) {
    __syncthreads();  // previous step's fragment reads are done
    const bool cur_cin_ok = cur.chunk * kCK + slot * 8 < p.CIN;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<uint4 *>(ldsA + swz_off(wave * 32 + rw + 8 * j, slot, kRowBytes)) =
          sel4(((aok >> j) & 1u) && cur_cin_ok, areg[j]);
    if constexpr (!BT) {
#pragma unroll
      for (int j = 0; j < BROWS; ++j) {
        const int n = r0 + 32 * j;
        if (n < COUT)
          *reinterpret_cast<uint4 *>(ldsB + swz_off(n, slot, kRowBytes)) = sel4(cur_cin_ok, breg[j]);
      }
    } else {
      // transpose: dword (d even, d odd) of channel n lands in row n, reduction column 2*r0
      const bool d_ok = cur.chunk * kCK + 2 * r0 < p.CIN;
#pragma unroll
      for (int jj = 0; jj < BROWS / 2; ++jj) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int n = jj * 64 + slot * 8 + e;
          const int sh = (e & 1) * 16;
          const uint32_t lo = (dword_of4(breg[2 * jj], e >> 1) >> sh) & 0xffffu;
          const uint32_t hi = (dword_of4(breg[2 * jj + 1], e >> 1) >> sh) & 0xffffu;
          if (n < COUT)
            *reinterpret_cast<uint32_t *>(ldsB + swz_off(n, r0 >> 2, kRowBytes) + (r0 & 3) * 4) =
                d_ok ? (lo | (hi << 16)) : 0u;
        }
      }
    }
    __syncthreads();
    // prefetch: data of the next step (its indices arrived during the previous step), then
    // the indices of the step after it
    const StepIt nn = step_next(nxt, nchunk);
    if (nxt.k >= 0) load_data(nxt);
    if (nn.k >= 0) load_idx(nn);
    if ((wavemask >> cur.k) & 1u) {        // none of this wave's 32 rows uses offset k: skip
      const int c0 = cur.chunk * kCK;
      const int ksteps = (min(kCK, p.CIN - c0) + 31) >> 5;  // 1 or 2
      for (int ks = 0; ks < ksteps; ++ks) {
        const int fslot = ks * 4 + (lane >> 4);
        uint4 fb[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
          fb[mb] = *reinterpret_cast<const uint4 *>(
              ldsA + swz_off(wave * 32 + mb * 16 + (lane & 15), fslot, kRowBytes));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const uint4 fa = *reinterpret_cast<const uint4 *>(
              ldsB + swz_off(nb * 16 + (lane & 15), fslot, kRowBytes));
          acc[nb][0] = mfma16<BF16>(fa, fb[0], acc[nb][0]);
          acc[nb][1] = mfma16<BF16>(fa, fb[1], acc[nb][1]);
        }
      }
    }
    cur = nxt;
    nxt = nn;
  }

  // ---- epilogue: bias/activation, fp32 -> 16 bit, transpose through LDS so
  // every output row leaves as full 16-byte-per-lane coalesced stores.
  __syncthreads();
  const uint16_t *bias = static_cast<const uint16_t *>(p.bias);
  const bool plain = bias == nullptr && p.act == SPX_ACT_NONE;   // uniform: training path
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int ch = nb * 16 + (lane >> 4) * 4;  // D row = channel, D col = voxel
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = to_float<BF16>(bias[ch + e]);
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int row = wave * 32 + mb * 16 + (lane & 15);
      uint16_t h[4];
      if (plain) {
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = from_float<BF16>(acc[nb][mb][e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          h[e] = from_float<BF16>(apply_act(acc[nb][mb][e] + bv[e], p.act, p.act_alpha));
      }
      uint2 pk;
      pk.x = static_cast<uint32_t>(h[0]) | (static_cast<uint32_t>(h[1]) << 16);
      pk.y = static_cast<uint32_t>(h[2]) | (static_cast<uint32_t>(h[3]) << 16);
      // 8-byte piece inside 16-byte slot (ch / 8)
      *reinterpret_cast<uint2 *>(smem + swz_off(row, ch >> 3, OUT_ROWB, OXM) + ((ch & 4) << 1)) = pk;
    }
  }
  __syncthreads();
  uint16_t *out = static_cast<uint16_t *>(p.out);
  constexpr int OSLOTS = COUT / 8;  // 16-byte slots per output row
  for (int s = tid; s < kTileM * OSLOTS; s += kThreads) {
    const int row = s / OSLOTS, sl = s % OSLOTS;
    const int t = tile * kTileM + row;
    if (t < p.n_dst) {
      const int g = p.argsort ? p.argsort[t] : t;
      *reinterpret_cast<uint4 *>(out + static_cast<size_t>(g) * COUT + sl * 8) =
          *reinterpret_cast<const uint4 *>(smem + swz_off(row, sl, OUT_ROWB, OXM));
    }
  }
}

template <int COUT>
constexpr size_t gemm_smem_bytes() {
  const size_t stage = kTileM * kRowBytes + COUT * kRowBytes + 32;
  const size_t outb = static_cast<size_t>(kTileM) * COUT * 2;
  return stage > outb ? stage : outb;
}

template <int COUT, bool BF16>
int launch_gather_gemm(const GemmParams &p, hipStream_t s) {
  const int ntiles = div_up(p.n_dst, kTileM);
  if (p.strideD == 1)
    hipLaunchKernelGGL((gather_gemm_mfma_kernel<COUT, BF16, false>), dim3(ntiles), dim3(kThreads),
                       gemm_smem_bytes<COUT>(), s, p);
  else
    hipLaunchKernelGGL((gather_gemm_mfma_kernel<COUT, BF16, true>), dim3(ntiles), dim3(kThreads),
                       gemm_smem_bytes<COUT>(), s, p);
  SPX_LAUNCH_CHECK();
  return 0;
}

template <bool BF16>
int launch_gen1_cout(const GemmParams &p, hipStream_t s) {
  switch (p.COUT) {
    case 16: return launch_gather_gemm<16, BF16>(p, s);
    case 32: return launch_gather_gemm<32, BF16>(p, s);
    case 64: return launch_gather_gemm<64, BF16>(p, s);
    case 128: return launch_gather_gemm<128, BF16>(p, s);
    case 256: return launch_gather_gemm<256, BF16>(p, s);
  }
  set_error("unsupported COUT %d for the MFMA path", p.COUT);
  return -1;
}

}  // namespace

int launch_gather_gemm_gen1(const GemmParams &p, bool bf16, hipStream_t s) {
  return bf16 ? launch_gen1_cout<true>(p, s) : launch_gen1_cout<false>(p, s);
}

}  // namespace spx
