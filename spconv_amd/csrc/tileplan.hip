// Tile plans for the dense-neighbourhood gather-GEMM (igemm.hip: igemm_halo_kernel).
//
// Real point clouds have ~6 pairs per voxel: gathering every pair's source row from global memory
// moves each feature row through a CU's vector-memory path six times.  A plan makes the reuse
// explicit (layout: common.h):
//   1. spatial order: a stable LSD radix sort of the destination rows by (batch, Morton code of the
//      two innermost coordinates / 2) -- the reference sorts its rows too (by mask,
//      all.py:935-991); neighbours in space become neighbours in the tile sequence;
//   2. per tile of 128 consecutive rows: the set of source rows its pairs reference is deduplicated
//      in an LDS hash (wave-parallel inserts, atomicCAS on LDS words), numbered, and written out as
//      the tile's halo list together with the (offset, row) -> halo slot table.
// Nothing here influences results: a plan only changes which workgroup computes a row and where
// the kernel finds its operands (the per-row arithmetic order is the offset order either way).
#include "common.h"

namespace spx {
namespace {

constexpr int kBlock = 256;
constexpr int kSortItems = 512;               // entries per block of a radix pass (>= 256 blocks at 128 k rows)
constexpr int kRadixBits = 8, kRadix = 1 << kRadixBits;

__device__ __forceinline__ uint32_t spread_bits(uint32_t v) {   // abcd -> 0a0b0c0d
  v &= 0xffffu;
  v = (v | (v << 8)) & 0x00ff00ffu;
  v = (v | (v << 4)) & 0x0f0f0f0fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

// key = batch : morton(y >> sh, x >> sh) for the two innermost coordinates (x only for 1-d).
// Rows with a batch index outside [0, batch) sort behind everything else.
__global__ void __launch_bounds__(kBlock)
plan_keys_kernel(const int32_t *__restrict__ indices, int n, int ndim, int batch, int sh, int mbits,
                 uint32_t *__restrict__ keys) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int32_t *row = indices + static_cast<size_t>(i) * (ndim + 1);
  const int b = row[0];
  const uint32_t x = static_cast<uint32_t>(row[ndim]) >> sh;
  const uint32_t y = ndim >= 2 ? static_cast<uint32_t>(row[ndim - 1]) >> sh : 0u;
  uint32_t m = (spread_bits(y) << 1) | spread_bits(x);
  uint32_t key = (b >= 0 && b < batch) ? ((static_cast<uint32_t>(b) << mbits) | m)
                                       : ((static_cast<uint32_t>(batch) << mbits));
  keys[i] = key;
}

__global__ void __launch_bounds__(kBlock)
plan_radix_count_kernel(const uint32_t *__restrict__ keys, int n, int shift, int nblk,
                        int32_t *__restrict__ hist /*[kRadix][nblk]*/) {
  __shared__ int lds_hist[kRadix];
  lds_hist[threadIdx.x] = 0;
  __syncthreads();
  const int begin = blockIdx.x * kSortItems;
#pragma unroll
  for (int it = 0; it < kSortItems / kBlock; ++it) {
    const int e = begin + it * kBlock + threadIdx.x;
    if (e < n) atomicAdd(&lds_hist[(keys[e] >> shift) & (kRadix - 1)], 1);
  }
  __syncthreads();
  hist[static_cast<size_t>(threadIdx.x) * nblk + blockIdx.x] = lds_hist[threadIdx.x];
}

// one block per digit: exclusive scan of the digit's per-block counts, digit total to totals[digit]
__global__ void __launch_bounds__(kBlock)
plan_radix_scan_kernel(const int32_t *__restrict__ hist, int32_t *__restrict__ off, int nblk,
                       int32_t *__restrict__ totals) {
  __shared__ int lds_wave[kBlock / 64];
  const int32_t *c = hist + static_cast<size_t>(blockIdx.x) * nblk;
  int32_t *o = off + static_cast<size_t>(blockIdx.x) * nblk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int carry = 0;
  for (int base = 0; base < nblk; base += kBlock) {
    const int idx = base + threadIdx.x;
    const int v = idx < nblk ? c[idx] : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int u = __shfl_up(incl, d, 64);
      if (lane >= d) incl += u;
    }
    __syncthreads();
    if (lane == 63) lds_wave[wave] = incl;
    __syncthreads();
    int prefix = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
      const int s = lds_wave[w];
      if (w < wave) prefix += s;
      total += s;
    }
    if (idx < nblk) o[idx] = carry + prefix + incl - v;
    carry += total;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// stable scatter of one pass; the base of a digit = (entries of smaller digits) + (entries of this
// digit in earlier blocks); inside a 256-entry group the rank among equal digits comes from a
// bitwise match over wave ballots
__global__ void __launch_bounds__(kBlock)
plan_radix_scatter_kernel(const uint32_t *__restrict__ keys_in, const int32_t *__restrict__ vals_in,
                          int n, int shift, int nblk, const int32_t *__restrict__ hist_off,
                          const int32_t *__restrict__ totals, uint32_t *__restrict__ keys_out,
                          int32_t *__restrict__ vals_out) {
  __shared__ int lds_base[kRadix];
  __shared__ int lds_cnt[kBlock / 64][kRadix];
  __shared__ int lds_wave[kBlock / 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  {   // exclusive scan of the 256 digit totals
    const int v = totals[threadIdx.x];
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int u = __shfl_up(incl, d, 64);
      if (lane >= d) incl += u;
    }
    if (lane == 63) lds_wave[wave] = incl;
    __syncthreads();
    int prefix = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w)
      if (w < wave) prefix += lds_wave[w];
    lds_base[threadIdx.x] = prefix + incl - v + hist_off[static_cast<size_t>(threadIdx.x) * nblk + blockIdx.x];
  }
  const int begin = blockIdx.x * kSortItems;
  for (int it = 0; it < kSortItems / kBlock; ++it) {
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) lds_cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const int e = begin + it * kBlock + threadIdx.x;
    const bool valid = e < n;
    const uint32_t key = valid ? keys_in[e] : 0u;
    const int val = valid ? (vals_in ? vals_in[e] : e) : 0;
    const int digit = valid ? static_cast<int>((key >> shift) & (kRadix - 1)) : -1;
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < kRadixBits; ++bit) {
      const unsigned long long bal = __ballot((digit >> bit) & 1);
      same &= ((digit >> bit) & 1) ? bal : ~bal;
    }
    const int rank_in_wave = __popcll(same & ((1ull << lane) - 1ull));
    if (valid && rank_in_wave == 0) lds_cnt[wave][digit] = __popcll(same);
    __syncthreads();
    if (valid) {
      int prior = 0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w)
        if (w < wave) prior += lds_cnt[w][digit];
      const int dst = lds_base[digit] + prior + rank_in_wave;
      keys_out[dst] = key;
      vals_out[dst] = val;
    }
    __syncthreads();
    {
      int sum = 0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) sum += lds_cnt[w][threadIdx.x];
      lds_base[threadIdx.x] += sum;
    }
    __syncthreads();
  }
}

// One workgroup per tile: halo list + (offset, row) -> slot table.
constexpr int kHashCap = 4096;                // >= 128 rows x 32 offsets
constexpr int kEntriesPerThread = (kTileRows * 32) / kBlock;   // 16

__global__ void __launch_bounds__(kBlock)
plan_halo_kernel(const int32_t *__restrict__ order_src /* sorted dst rows, n_dst */, int n_dst,
                 const int32_t *__restrict__ pair, int kv, int32_t *__restrict__ order_out,
                 int32_t *__restrict__ tile_info, int32_t *__restrict__ halo_rows,
                 uint16_t *__restrict__ plocal, int32_t *__restrict__ header) {
  __shared__ int lds_key[kHashCap];
  __shared__ uint16_t lds_dense[kHashCap];
  __shared__ int lds_wave[kBlock / 64];
  __shared__ unsigned int lds_kmask;
  const int tile = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < kHashCap; i += kBlock) lds_key[i] = -1;
  if (tid == 0) lds_kmask = 0u;
  const int r = tid & (kTileRows - 1), khalf = tid >> 7;        // row of the tile, offset parity
  const int pos = tile * kTileRows + r;
  const int g = pos < n_dst ? order_src[pos] : -1;
  if (khalf == 0) order_out[pos] = g;
  __syncthreads();
  int hslot[kEntriesPerThread];
  unsigned int kbits = 0;
#pragma unroll
  for (int i = 0; i < kEntriesPerThread; ++i) {
    const int k = 2 * i + khalf;
    int src = -1;
    if (k < kv && g >= 0) src = pair[static_cast<size_t>(k) * n_dst + g];
    int hs = -1;
    if (src >= 0) {
      kbits |= 1u << k;
      unsigned int h = (static_cast<unsigned int>(src) * 2654435761u) >> 20;     // 12 bits
      for (int probe = 0; probe < kHashCap; ++probe) {
        const int prev = atomicCAS(&lds_key[h], -1, src);
        if (prev == -1 || prev == src) {
          hs = static_cast<int>(h);
          break;
        }
        h = (h + 1) & (kHashCap - 1);
      }
    }
    hslot[i] = hs;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) kbits |= __shfl_xor(kbits, d, 64);
  if ((tid & 63) == 0 && kbits) atomicOr(&lds_kmask, kbits);
  __syncthreads();
  // number the occupied hash slots (16 consecutive slots per thread, block exclusive scan)
  int mine = 0;
#pragma unroll
  for (int j = 0; j < kHashCap / kBlock; ++j) mine += lds_key[tid * (kHashCap / kBlock) + j] >= 0 ? 1 : 0;
  int incl = mine;
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int u = __shfl_up(incl, d, 64);
    if (lane >= d) incl += u;
  }
  if (lane == 63) lds_wave[wave] = incl;
  __syncthreads();
  int prefix = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    if (w < wave) prefix += lds_wave[w];
    total += lds_wave[w];
  }
  int run = prefix + incl - mine;
#pragma unroll
  for (int j = 0; j < kHashCap / kBlock; ++j) {
    const int h = tid * (kHashCap / kBlock) + j;
    const int key = lds_key[h];
    if (key >= 0) {
      lds_dense[h] = static_cast<uint16_t>(run < kHaloMax ? run : kSpilled);
      if (run < kHaloMax) halo_rows[static_cast<size_t>(tile) * kHaloMax + run] = key;
      ++run;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kEntriesPerThread; ++i) {
    const int k = 2 * i + khalf;
    if (k < kv) {
      const uint16_t v = hslot[i] >= 0 ? lds_dense[hslot[i]] : kNoPair;
      plocal[(static_cast<size_t>(tile) * kv + k) * kTileRows + r] = v;
    }
  }
  if (tile == 0 && tid < kPlanHeader) {
    const int hv[5] = {kPlanMagic, n_dst, static_cast<int>(gridDim.x), kv, kHaloMax};
    header[tid] = tid < 5 ? hv[tid] : 0;
  }
  if (tid == 0) {
    int32_t *info = tile_info + static_cast<size_t>(tile) * 4;
    info[0] = static_cast<int32_t>(lds_kmask);
    info[1] = total < kHaloMax ? total : kHaloMax;
    info[2] = total > kHaloMax ? 1 : 0;
    info[3] = total;
  }
}

int bits_of(unsigned v) {
  int b = 0;
  while (v) {
    ++b;
    v >>= 1;
  }
  return b;
}

__global__ void __launch_bounds__(kBlock)
permute_tables_kernel(const int32_t *__restrict__ pair, const uint32_t *__restrict__ mask,
                      const int32_t *__restrict__ order, int kv, int n, int words,
                      int32_t *__restrict__ pair_t, uint32_t *__restrict__ mask_t) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= n) return;
  const int g = order[t];
  const int k = blockIdx.y;
  if (k < kv) {
    pair_t[static_cast<size_t>(k) * n + t] = pair[static_cast<size_t>(k) * n + g];
  } else {
    for (int w = 0; w < words; ++w) mask_t[static_cast<size_t>(t) * words + w] = mask[static_cast<size_t>(g) * words + w];
  }
}

}  // namespace

size_t radix_argsort_ws_bytes(int n_in) {
  const size_t n = n_in > 0 ? n_in : 1;
  const size_t nblk = (n + kSortItems - 1) / kSortItems;
  return 2 * align_up(n * 4, 256) + 2 * align_up(n * 4, 256) + 2 * align_up(kRadix * nblk * 4, 256) +
         align_up(kRadix * 4, 256) + 256;
}

// Stable LSD radix argsort of n 32-bit keys on their low `nbits` bits: order_out[t] = index of the
// t-th smallest key.  8-bit digits, three launches per pass (count, per-digit scan, scatter), all
// of them wide (>= n / 512 workgroups).  `keys` is not modified.
int radix_argsort(const uint32_t *keys, int n, int nbits, int32_t *order_out, void *ws, hipStream_t s) {
  if (n <= 0) return 0;
  const int nblk = div_up(n, kSortItems);
  Carver cv(ws);
  uint32_t *kA = cv.take<uint32_t>(n), *kB = cv.take<uint32_t>(n);
  int32_t *vA = cv.take<int32_t>(n), *vB = cv.take<int32_t>(n);
  (void)vB;
  int32_t *hist = cv.take<int32_t>(static_cast<size_t>(kRadix) * nblk);
  int32_t *hist_off = cv.take<int32_t>(static_cast<size_t>(kRadix) * nblk);
  int32_t *totals = cv.take<int32_t>(kRadix);
  const int passes = div_up(nbits > 0 ? nbits : 1, kRadixBits);
  const uint32_t *kin = keys;
  const int32_t *vin = nullptr;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = pass * kRadixBits;
    // value buffers alternate so that the LAST pass writes order_out
    int32_t *vout = ((passes - 1 - pass) & 1) ? vA : order_out;
    uint32_t *kout = (pass & 1) ? kA : kB;
    hipLaunchKernelGGL(plan_radix_count_kernel, dim3(nblk), dim3(kBlock), 0, s, kin, n, shift, nblk, hist);
    hipLaunchKernelGGL(plan_radix_scan_kernel, dim3(kRadix), dim3(kBlock), 0, s, hist, hist_off, nblk, totals);
    hipLaunchKernelGGL(plan_radix_scatter_kernel, dim3(nblk), dim3(kBlock), 0, s, kin, vin, n, shift, nblk,
                       hist_off, totals, kout, vout);
    kin = kout;
    vin = vout;
  }
  SPX_LAUNCH_CHECK();
  return 0;
}

}  // namespace spx

using namespace spx;

extern "C" {

size_t spx_tile_plan_bytes(int n_dst, int kv) { return plan_ints(n_dst > 0 ? n_dst : 1, kv) * sizeof(int32_t); }

size_t spx_tile_plan_ws_bytes(int n_dst) {
  const size_t n = n_dst > 0 ? n_dst : 1;
  return 2 * align_up(n * 4, 256) + align_up(radix_argsort_ws_bytes(n_dst), 256) + 256;
}

/* Copies of a pair table [kv, n] and its mask words [n, words] in TILE ORDER: row t of the copies
 * belongs to destination row order[t] (order = spx_mask_argsort's output).  With them the
 * gather-GEMM reads the tables of a sorted tile as contiguous 512-byte runs instead of 128
 * scattered words (spx_igemm_* with tile_order = 1). */
int spx_permute_tables(const int32_t *pair, const uint32_t *mask, const int32_t *order, int kv, int n,
                       int words, int32_t *pair_t, uint32_t *mask_t, spx_stream_t stream) {
  SPX_CHECK(pair && mask && order && pair_t && mask_t, "null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(permute_tables_kernel, dim3(div_up(n, kBlock), kv + 1), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), pair, mask, order, kv, n, words, pair_t, mask_t);
  SPX_LAUNCH_CHECK();
  return 0;
}

int spx_tile_plan_build(const int32_t *dst_indices, int n_dst, int ndim, int batch_size,
                        const int *dst_shape, const int32_t *pair, int kv, int32_t *plan, void *ws,
                        size_t ws_bytes, spx_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  SPX_CHECK(ndim >= 1 && ndim <= kMaxNdim, "ndim must be in [1,4], got %d", ndim);
  SPX_CHECK(kv >= 1 && kv <= 32, "tile plans cover kernel volumes up to 32, got %d", kv);
  SPX_CHECK(dst_indices && pair && plan && ws, "null pointer");
  SPX_CHECK(ws_bytes >= spx_tile_plan_ws_bytes(n_dst), "workspace too small");
  SPX_CHECK(n_dst > 0, "empty tensors need no plan");
  Carver cv(ws);
  uint32_t *keys = cv.take<uint32_t>(n_dst);
  int32_t *sorted = cv.take<int32_t>(n_dst);
  void *sort_ws = cv.take<char>(radix_argsort_ws_bytes(n_dst));
  // key width: batch bits on top of an even number of Morton bits; coarsen the cells (>> sh) until
  // the key fits 24 bits = three 8-bit passes (cells of 2 x 2 are fine enough: measured on the
  // reference's LiDAR fixture, halo 1.45 x the tile rows at sh = 1 vs 1.43 at sh = 0)
  const int ymax = ndim >= 2 ? dst_shape[ndim - 2] : 1, xmax = dst_shape[ndim - 1];
  const int bbits = bits_of(static_cast<unsigned>(batch_size));       // values 0 .. batch_size
  int sh = 1, mbits = 0;
  for (;; ++sh) {
    const int cb = bits_of(static_cast<unsigned>((xmax > ymax ? xmax : ymax) - 1) >> sh);
    mbits = 2 * cb;
    if (mbits + bbits <= 24 || cb == 0) break;
  }
  hipLaunchKernelGGL(plan_keys_kernel, dim3(div_up(n_dst, kBlock)), dim3(kBlock), 0, s, dst_indices, n_dst,
                     ndim, batch_size, sh, mbits, keys);
  if (radix_argsort(keys, n_dst, mbits + bbits, sorted, sort_ws, s)) return -2;
  const int32_t *vin = sorted;
  const int ntiles = div_up(n_dst, kTileRows);
  const PlanView v = plan_view(plan, n_dst, kv);
  hipLaunchKernelGGL(plan_halo_kernel, dim3(ntiles), dim3(kBlock), 0, s, vin, n_dst, pair, kv,
                     const_cast<int32_t *>(v.order), const_cast<int32_t *>(v.tile_info),
                     const_cast<int32_t *>(v.halo_rows), const_cast<uint16_t *>(v.plocal), plan);
  SPX_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
