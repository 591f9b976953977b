// Row orders of a rulebook: stable LSD radix argsort (spx_mask_argsort: the reference's
// sort_1d_by_key_allocator, all.py:935-991) and the copies of a pair table / its mask words in
// that order (spx_permute_tables) which let a sorted tile read its tables as contiguous runs.
// Nothing here influences results: a row order only changes which workgroup computes a row.
#include "common.h"

namespace spx {
namespace {

constexpr int kBlock = 256;
constexpr int kSortItems = 512;               // entries per block of a radix pass (>= 256 blocks at 128 k rows)

// Digits of BITS = 8 or 9 bits, R = 2^BITS bins, R threads per workgroup.  Round 6: 9-bit digits where they save a
// pass -- the mask words of a 3x3x3 kernel carry 27 bits: three passes (nine launches) instead of four (twelve); at
// 100-125 k keys a launch of this sort sits at its ~3 us floor, so the sort goes 37-39 -> 28-30 us.  (A one-launch-per-
// pass form -- tiles handing their digit counts to each other through {status, value} words, decoupled look-back -- was
// built and measured at 39-43 us: on this chip a cross-workgroup hand-over inside a launch goes through memory-side
// atomics at ~1 us per hop, which costs more than the kernel boundaries it removes; profiles/r06_experiments.md.)
template <int BITS>
__global__ void __launch_bounds__(1 << BITS)
radix_count_kernel(const uint32_t *__restrict__ keys, int n, int shift, int nblk,
                   int32_t *__restrict__ hist /*[R][nblk]*/) {
  constexpr int R = 1 << BITS, T = R;
  __shared__ int lds_hist[R];
  lds_hist[threadIdx.x] = 0;
  __syncthreads();
  const int begin = blockIdx.x * kSortItems;
#pragma unroll
  for (int it = 0; it < (kSortItems + T - 1) / T; ++it) {
    const int e = begin + it * T + threadIdx.x;
    if (e < n && it * T + static_cast<int>(threadIdx.x) < kSortItems)
      atomicAdd(&lds_hist[(keys[e] >> shift) & (R - 1)], 1);
  }
  __syncthreads();
  hist[static_cast<size_t>(threadIdx.x) * nblk + blockIdx.x] = lds_hist[threadIdx.x];
}

// one block per digit: exclusive scan of the digit's per-block counts, digit total to totals[digit]
__global__ void __launch_bounds__(kBlock)
radix_scan_kernel(const int32_t *__restrict__ hist, int32_t *__restrict__ off, int nblk,
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
// digit in earlier blocks); inside a T-entry group the rank among equal digits comes from a
// bitwise match over wave ballots
template <int BITS>
__global__ void __launch_bounds__(1 << BITS)
radix_scatter_kernel(const uint32_t *__restrict__ keys_in, const int32_t *__restrict__ vals_in,
                     int n, int shift, int nblk, const int32_t *__restrict__ hist_off,
                     const int32_t *__restrict__ totals, uint32_t *__restrict__ keys_out,
                     int32_t *__restrict__ vals_out) {
  constexpr int R = 1 << BITS, T = R, NW = T / 64;
  __shared__ int lds_base[R];
  __shared__ int lds_cnt[NW][R];
  __shared__ int lds_wave[NW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  {   // exclusive scan of the R digit totals
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
    for (int w = 0; w < NW; ++w)
      if (w < wave) prefix += lds_wave[w];
    lds_base[threadIdx.x] = prefix + incl - v + hist_off[static_cast<size_t>(threadIdx.x) * nblk + blockIdx.x];
  }
  const int begin = blockIdx.x * kSortItems;
  for (int it = 0; it < (kSortItems + T - 1) / T; ++it) {
#pragma unroll
    for (int w = 0; w < NW; ++w) lds_cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const int e = begin + it * T + threadIdx.x;
    const bool valid = e < n && it * T + static_cast<int>(threadIdx.x) < kSortItems;
    const uint32_t key = valid ? keys_in[e] : 0u;
    const int val = valid ? (vals_in ? vals_in[e] : e) : 0;
    const int digit = valid ? static_cast<int>((key >> shift) & (R - 1)) : -1;
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < BITS; ++bit) {
      const unsigned long long bal = __ballot((digit >> bit) & 1);
      same &= ((digit >> bit) & 1) ? bal : ~bal;
    }
    const int rank_in_wave = __popcll(same & ((1ull << lane) - 1ull));
    if (valid && rank_in_wave == 0) lds_cnt[wave][digit] = __popcll(same);
    __syncthreads();
    if (valid) {
      int prior = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if (w < wave) prior += lds_cnt[w][digit];
      const int dst = lds_base[digit] + prior + rank_in_wave;
      keys_out[dst] = key;
      vals_out[dst] = val;
    }
    __syncthreads();
    {
      int sum = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) sum += lds_cnt[w][threadIdx.x];
      lds_base[threadIdx.x] += sum;
    }
    __syncthreads();
  }
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
  return 2 * align_up(n * 4, 256) + 2 * align_up(n * 4, 256) + 2 * align_up(512 * nblk * 4, 256) +
         align_up(512 * 4, 256) + 256;
}

namespace {
template <int BITS>
void radix_pass(const uint32_t *kin, const int32_t *vin, int n, int shift, int nblk, int32_t *hist, int32_t *hist_off,
                int32_t *totals, uint32_t *kout, int32_t *vout, hipStream_t s) {
  constexpr int R = 1 << BITS;
  hipLaunchKernelGGL(radix_count_kernel<BITS>, dim3(nblk), dim3(R), 0, s, kin, n, shift, nblk, hist);
  hipLaunchKernelGGL(radix_scan_kernel, dim3(R), dim3(kBlock), 0, s, hist, hist_off, nblk, totals);
  hipLaunchKernelGGL(radix_scatter_kernel<BITS>, dim3(nblk), dim3(R), 0, s, kin, vin, n, shift, nblk, hist_off, totals,
                     kout, vout);
}
}  // namespace

// Stable LSD radix argsort of n 32-bit keys on their low `nbits` bits: order_out[t] = index of the
// t-th smallest key.  Three launches per pass (count, per-digit scan, scatter), all of them wide (>= n / 512
// workgroups); 9-bit digits when that saves a pass (nbits = 27: three passes), 8-bit otherwise.  `keys` is not modified.
int radix_argsort(const uint32_t *keys, int n, int nbits, int32_t *order_out, void *ws, hipStream_t s) {
  if (n <= 0) return 0;
  if (nbits < 1) nbits = 1;
  if (nbits > 32) nbits = 32;
  const int bits = (nbits + 8) / 9 < (nbits + 7) / 8 ? 9 : 8;
  const int nblk = div_up(n, kSortItems);
  Carver cv(ws);
  uint32_t *kA = cv.take<uint32_t>(n), *kB = cv.take<uint32_t>(n);
  int32_t *vA = cv.take<int32_t>(n), *vB = cv.take<int32_t>(n);
  (void)vB;
  int32_t *hist = cv.take<int32_t>(static_cast<size_t>(512) * nblk);
  int32_t *hist_off = cv.take<int32_t>(static_cast<size_t>(512) * nblk);
  int32_t *totals = cv.take<int32_t>(512);
  const int passes = div_up(nbits, bits);
  const uint32_t *kin = keys;
  const int32_t *vin = nullptr;
  for (int pass = 0; pass < passes; ++pass) {
    // value buffers alternate so that the LAST pass writes order_out
    int32_t *vout = ((passes - 1 - pass) & 1) ? vA : order_out;
    uint32_t *kout = (pass & 1) ? kA : kB;
    if (bits == 9) radix_pass<9>(kin, vin, n, pass * bits, nblk, hist, hist_off, totals, kout, vout, s);
    else radix_pass<8>(kin, vin, n, pass * bits, nblk, hist, hist_off, totals, kout, vout, s);
    kin = kout;
    vin = vout;
  }
  SPX_LAUNCH_CHECK();
  return 0;
}

}  // namespace spx

using namespace spx;

extern "C" {

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

}  // extern "C"
