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

// ---- rows of a level in coordinate-key order (spx_key_argsort) --------------------------------------------------------
// Keys of one level are UNIQUE (a voxeliser's output; spx_rankmap_from_sorted checks it), which buys a sort in four
// launches instead of thirteen: ONE stable radix pass (count | scan | scatter, above) on the UPPER key bits groups the
// rows into <= 511 buckets of 2^SH consecutive keys, and a workgroup per bucket then ranks its rows without comparing
// anything -- an occupancy bit per key in LDS (2^SH bits <= 128 KB), popcount prefixes per 8 words, rank = set bits
// below the row's own.  Dead rows (batch -1: static shapes) and rows out of range get the key of a bucket of their own
// behind every live one: the stable pass leaves them there in row order, and their positions are final.
// Cost model at 440 k rows / 2^29 keys: key + count 4.5 us, scan 4, scatter 5, buckets ~8 (512 workgroups, one per CU
// at a time); the thirteen-launch LSD form measured ~75 us in a captured config-4 step (profiles/r06_experiments.md).
struct KeyGeom {
  int ndim, batch;
  int dims[4];        // canonical 4-d extents, leading ones
  uint32_t past;      // key of a dead row: nb_live << sh
};

__device__ __forceinline__ uint32_t row_key(const int32_t *__restrict__ indices, int i, const KeyGeom &g) {
  int b, c[4];
  if (g.ndim == 3) {
    const int4 v = reinterpret_cast<const int4 *>(indices)[i];
    b = v.x; c[0] = 0; c[1] = v.y; c[2] = v.z; c[3] = v.w;
  } else {
    const int32_t *row = indices + static_cast<size_t>(i) * (g.ndim + 1);
    b = row[0];
    const int lead = 4 - g.ndim;
#pragma unroll
    for (int d = 0; d < 4; ++d) c[d] = (d < lead) ? 0 : row[1 + d - lead];
  }
  bool ok = b >= 0 && b < g.batch;
  unsigned long long v = static_cast<unsigned long long>(b);
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    ok = ok && c[d] >= 0 && c[d] < g.dims[d];
    v = v * static_cast<unsigned long long>(g.dims[d]) + static_cast<unsigned long long>(c[d]);
  }
  return ok ? static_cast<uint32_t>(v) : g.past;
}

// count pass of the first digit that also MAKES the keys (one read of the index rows)
__global__ void __launch_bounds__(512)
key_count_kernel(const int32_t *__restrict__ indices, int n, KeyGeom g, int shift, int nblk,
                 uint32_t *__restrict__ keys, int32_t *__restrict__ hist /*[512][nblk]*/, int32_t *__restrict__ violation,
                 uint4 *__restrict__ zero_fill, unsigned long long zero_units) {
  constexpr int R = 512;
  __shared__ int lds_hist[R];
  if (violation && blockIdx.x == 0 && threadIdx.x == 0) *violation = 0;
  // the level's rank map starts empty: its fill rides here (stores that nothing in this launch waits for), the bucket
  // pass then writes the occupied words only
  for (unsigned long long u = static_cast<unsigned long long>(blockIdx.x) * R + threadIdx.x; u < zero_units;
       u += static_cast<unsigned long long>(nblk) * R)
    zero_fill[u] = make_uint4(0u, 0u, 0u, 0u);
  lds_hist[threadIdx.x] = 0;
  __syncthreads();
  const int e = blockIdx.x * kSortItems + threadIdx.x;
  static_assert(kSortItems == R, "one entry per thread");
  if (e < n) {
    const uint32_t key = row_key(indices, e, g);
    keys[e] = key;
    atomicAdd(&lds_hist[(key >> shift) & (R - 1)], 1);
  }
  __syncthreads();
  hist[static_cast<size_t>(threadIdx.x) * nblk + blockIdx.x] = lds_hist[threadIdx.x];
}

constexpr int kBucketThreads = 1024;
constexpr int kDeadChunk = 4096;      // dead rows per workgroup (a copy)
constexpr int kPartRows = 2048;       // a bucket of more rows than this is ranked by several workgroups

__device__ __forceinline__ void store_index_row(int32_t *__restrict__ out, int pos, uint32_t key, const KeyGeom &g, bool dead) {
  int c[4], b;
  if (dead) {
    b = c[0] = c[1] = c[2] = c[3] = -1;
  } else {
    uint32_t v = key;
#pragma unroll
    for (int d = 3; d >= 0; --d) {
      const uint32_t q = v / static_cast<uint32_t>(g.dims[d]);
      c[d] = static_cast<int>(v - q * static_cast<uint32_t>(g.dims[d]));
      v = q;
    }
    b = static_cast<int>(v);
  }
  if (g.ndim == 3) {
    reinterpret_cast<int4 *>(out)[pos] = make_int4(b, c[1], c[2], c[3]);
  } else {
    int32_t *row = out + static_cast<size_t>(pos) * (g.ndim + 1);
    row[0] = b;
    const int lead = 4 - g.ndim;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      if (d >= lead) row[1 + d - lead] = c[d];
  }
}

struct RankMapOut {        // the level's rank map (csrc/rulebook.hip rank_of: {bits, rows before the word}, block offsets 0)
  uint2 *cells;            // null: not wanted.  Zero-filled by key_count_kernel: only occupied words are written
  unsigned long long words;
  int32_t *violation;
  // rows that travel with the sort (a level's features): rows_out[t] = rows_in[order[t]], row_bytes a multiple of 4
  const char *rows_in;
  char *rows_out;
  int row_bytes;
};

__device__ __forceinline__ void carry_row(const RankMapOut &rm, int dst, int src) {
  if (!rm.rows_out) return;
  const char *s = rm.rows_in + static_cast<size_t>(src) * rm.row_bytes;
  char *d = rm.rows_out + static_cast<size_t>(dst) * rm.row_bytes;
  if (rm.row_bytes == 8) {
    *reinterpret_cast<uint2 *>(d) = *reinterpret_cast<const uint2 *>(s);
  } else if ((rm.row_bytes & 15) == 0) {
    for (int o = 0; o < rm.row_bytes; o += 16) *reinterpret_cast<uint4 *>(d + o) = *reinterpret_cast<const uint4 *>(s + o);
  } else {
    for (int o = 0; o < rm.row_bytes; o += 4) *reinterpret_cast<uint32_t *>(d + o) = *reinterpret_cast<const uint32_t *>(s + o);
  }
}

// A JOB = (bucket b, part): the rows of bucket b -- a run of the bucket-sorted arrays -- to their ranks.  Every part
// workgroup of a bucket builds the bucket's whole occupancy map (LDS atomics are cheap, the keys come out of the L2) and
// ranks ITS share of the rows; a LiDAR scene puts 17 k of 400 k rows into one bucket of 2^20 keys, which one workgroup
// would walk for 40 us.  `totals` = the 512 digit counts of the ONE radix pass (bucket = digit): jobs are laid out by a
// scan over them in every workgroup.  null (several passes: key spaces beyond 2^29): one job per bucket, its run found
// by bisection on keys >> sh.  Bucket nb_live holds the dead rows, in place behind the stable pass: copied in chunks.
// With `rm.cells` every job also writes the occupied words of its share of the level's rank map (the first launch of
// the sort zero-filled it), so the map needs no pass of its own.
__global__ void __launch_bounds__(kBucketThreads)
key_bucket_kernel(const uint32_t *__restrict__ keys, const int32_t *__restrict__ vals, int n, KeyGeom g, int sh,
                  int nb_live, const int32_t *__restrict__ totals, int32_t *__restrict__ order,
                  int32_t *__restrict__ idx_out, int32_t *__restrict__ tmp, RankMapOut rm) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  __shared__ int s_job[5];                       // bucket, part, parts, first row, rows
  __shared__ int s_wave[kBucketThreads / 64], s_wave2[kBucketThreads / 64];
  __shared__ int s_extra;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = 1 << (sh - 5);                  // words of a bucket's occupancy map (sh >= 10: >= 32)
  const int NG = W >> 3;                        // groups of 8 words
  uint32_t *bm = lds_dyn;
  int *gp = reinterpret_cast<int *>(lds_dyn + W);
  // ---- this workgroup's job ---------------------------------------------------------------------------------------
  if (tid == 0) s_job[0] = -1;
  if (totals) {
    int c = 0, p = 0;
    if (tid <= nb_live) {
      c = totals[tid];
      p = tid < nb_live ? max(1, (c + kPartRows - 1) / kPartRows) : (c + kDeadChunk - 1) / kDeadChunk;
    }
    int ci = c, pi = p;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int uc = __shfl_up(ci, d, 64), up = __shfl_up(pi, d, 64);
      if (lane >= d) { ci += uc; pi += up; }
    }
    if (lane == 63) { s_wave[wave] = ci; s_wave2[wave] = pi; }
    __syncthreads();
    int cpre = 0, ppre = 0;
#pragma unroll
    for (int w = 0; w < kBucketThreads / 64; ++w)
      if (w < wave) { cpre += s_wave[w]; ppre += s_wave2[w]; }
    const int pbeg = ppre + pi - p;
    const int me = static_cast<int>(blockIdx.x);
    if (tid <= nb_live && me >= pbeg && me < pbeg + p) {
      s_job[0] = tid; s_job[1] = me - pbeg; s_job[2] = p; s_job[3] = cpre + ci - c; s_job[4] = c;
    }
  } else if (tid == 0) {
    auto lower = [&](uint32_t bucket) {
      int lo = 0, hi = n;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((keys[mid] >> sh) < bucket) lo = mid + 1; else hi = mid;
      }
      return lo;
    };
    const int b = min(static_cast<int>(blockIdx.x), nb_live);
    const int beg = lower(static_cast<uint32_t>(b));
    const int end = b < nb_live ? lower(static_cast<uint32_t>(b) + 1u) : n;
    const int part = b < nb_live ? 0 : static_cast<int>(blockIdx.x) - nb_live;
    if (b < nb_live || part * kDeadChunk < end - beg) {
      s_job[0] = b; s_job[1] = part; s_job[2] = 1; s_job[3] = beg; s_job[4] = end - beg;
    }
  }
  __syncthreads();
  const int b = s_job[0], part = s_job[1], parts = s_job[2], beg = s_job[3], m = s_job[4];
  if (b < 0) return;
  if (b >= nb_live) {                            // dead rows: already in place
    const int c0 = beg + part * kDeadChunk, c1 = min(c0 + kDeadChunk, beg + m);
    for (int i = c0 + tid; i < c1; i += kBucketThreads) {
      order[i] = vals[i];
      if (idx_out) store_index_row(idx_out, i, 0u, g, true);
      carry_row(rm, i, vals[i]);
    }
    return;
  }
  // ---- occupancy bits of the whole bucket ------------------------------------------------------------------------------
  for (int w = tid * 4; w < W; w += kBucketThreads * 4) *reinterpret_cast<uint4 *>(bm + w) = make_uint4(0u, 0u, 0u, 0u);
  if (tid == 0) s_extra = 0;
  __syncthreads();
  const uint32_t low = (1u << sh) - 1u;
  // (a bucket of at most 4 rows per thread -- nearly all of them -- keeps its rows in registers for the rank pass: one
  // trip to memory per job instead of two)
  const bool small = m <= 4 * kBucketThreads;
  uint32_t kr[4];
  int vr[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = u * kBucketThreads + tid;
    kr[u] = i < m ? keys[beg + i] : 0xffffffffu;
    vr[u] = (small && i < m) ? vals[beg + i] : 0;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (u * kBucketThreads + tid < m) atomicOr(&bm[(kr[u] & low) >> 5], 1u << (kr[u] & 31));
  for (int i0 = 4 * kBucketThreads; i0 < m; i0 += 4 * kBucketThreads) {
    uint32_t k[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * kBucketThreads + tid;
      k[u] = i < m ? keys[beg + i] : 0xffffffffu;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * kBucketThreads + tid < m) atomicOr(&bm[(k[u] & low) >> 5], 1u << (k[u] & 31));
  }
  __syncthreads();
  // ---- set bits before each group of 8 words: rounds of kBucketThreads groups, carry between rounds ---------------------
  int carry = 0;
  for (int g0 = 0; g0 < NG; g0 += kBucketThreads) {
    const int gi = g0 + tid;
    int cnt = 0;
    if (gi < NG) {
      const uint4 a = *reinterpret_cast<const uint4 *>(bm + gi * 8), c = *reinterpret_cast<const uint4 *>(bm + gi * 8 + 4);
      cnt = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(c.x) + __popc(c.y) + __popc(c.z) + __popc(c.w);
    }
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int u = __shfl_up(incl, d, 64);
      if (lane >= d) incl += u;
    }
    __syncthreads();                            // (s_wave of the previous round / of the job scan is consumed)
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int prefix = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBucketThreads / 64; ++w) {
      const int x = s_wave[w];
      if (w < wave) prefix += x;
      total += x;
    }
    if (gi < NG) gp[gi] = carry + prefix + incl - cnt;
    carry += total;
  }
  __syncthreads();
  const bool dup = carry != m;                  // a key twice: fewer bits than rows (uniform over the bucket's parts)
  // ---- the level's rank map: the occupied words of this part's share of the bucket ---------------------------------------
  if (rm.cells) {
    const unsigned long long w0 = static_cast<unsigned long long>(b) << (sh - 5);
    const int ga = static_cast<int>(static_cast<long long>(NG) * part / parts);
    const int gb = static_cast<int>(static_cast<long long>(NG) * (part + 1) / parts);
    for (int gi = ga + tid; gi < gb; gi += kBucketThreads) {
      const uint4 a = *reinterpret_cast<const uint4 *>(bm + gi * 8), c = *reinterpret_cast<const uint4 *>(bm + gi * 8 + 4);
      if ((a.x | a.y | a.z | a.w | c.x | c.y | c.z | c.w) == 0u) continue;
      const uint32_t bits[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
      uint32_t run = static_cast<uint32_t>(beg + gp[gi]);
      const unsigned long long wg = w0 + static_cast<unsigned long long>(gi) * 8ull;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (bits[u] && wg + u < rm.words) rm.cells[wg + u] = make_uint2(bits[u], run);
        run += __popc(bits[u]);
      }
    }
    if (dup && part == 0 && tid == 0 && rm.violation) atomicOr(rm.violation, 1);
  }
  if (dup && part != 0) return;                 // (the repair below is one workgroup's)
  // ---- ranks of this part's rows ----------------------------------------------------------------------------------------
  const int ia = dup ? 0 : static_cast<int>(static_cast<long long>(m) * part / parts);
  const int ib = dup ? m : static_cast<int>(static_cast<long long>(m) * (part + 1) / parts);
  auto place = [&](int i, uint32_t key, int val) __attribute__((always_inline)) {
    const uint32_t k = key & low;
    const int w = static_cast<int>(k >> 5);
    int r = gp[w >> 3] + __popc(bm[w] & ((1u << (k & 31)) - 1u));
    for (int u = w & ~7; u < w; ++u) r += __popc(bm[u]);
    if (!dup) {
      order[beg + r] = val;
      if (idx_out) store_index_row(idx_out, beg + r, key, g, false);
      carry_row(rm, beg + r, val);
    } else {
      tmp[beg + i] = r;
    }
  };
  if (small) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = u * kBucketThreads + tid;
      if (i >= ia && i < ib) place(i, kr[u], vr[u]);
    }
  } else {
    for (int i = ia + tid; i < ib; i += kBucketThreads) place(i, keys[beg + i], vals[beg + i]);
  }
  if (!dup) return;
  // A coordinate twice (the caller's contract is broken: `violation`, and spx_rankmap_from_sorted raises its flag on the
  // result as well): still a permutation -- the first row to clear a key's bit takes the key's rank, the others the
  // places behind the distinct keys of the bucket.
  __syncthreads();
  for (int i = tid; i < m; i += kBucketThreads) {
    const uint32_t key = keys[beg + i], k = key & low;
    const uint32_t bit = 1u << (k & 31);
    const uint32_t old = atomicAnd(&bm[k >> 5], ~bit);
    const int r = (old & bit) ? tmp[beg + i] : carry + atomicAdd(&s_extra, 1);
    order[beg + r] = vals[beg + i];
    if (idx_out) store_index_row(idx_out, beg + r, key, g, false);
    carry_row(rm, beg + r, vals[beg + i]);
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

size_t key_argsort_ws_bytes(int n_in) {
  const size_t n = n_in > 0 ? n_in : 1;
  const size_t nblk = (n + kSortItems - 1) / kSortItems;
  return 5 * align_up(n * 4, 256) + 2 * align_up(512 * nblk * 4, 256) + align_up(512 * 4, 256) + 256;
}

// sh (key bits ranked inside a bucket) and the number of live buckets for a key space of `cells` keys
static void key_split(unsigned long long cells, int *sh_out, int *nb_live_out) {
  int nbits = 1;
  while (nbits < 33 && (cells >> nbits) != 0) ++nbits;
  int sh = nbits - 9;
  if (sh < 10) sh = 10;
  if (sh > 20) sh = 20;
  *sh_out = sh;
  *nb_live_out = static_cast<int>((cells + (1ull << sh) - 1ull) >> sh);
}

int key_argsort(const int32_t *indices, int n, int ndim, int batch_size, const int *spatial_shape, int32_t *order,
                int32_t *indices_sorted, void *rankmap, int32_t *violation, const void *rows, void *rows_sorted,
                int row_bytes, void *ws, hipStream_t s) {
  KeyGeom g{};
  g.ndim = ndim;
  g.batch = batch_size;
  unsigned long long cells = static_cast<unsigned long long>(batch_size);
  for (int d = 0; d < 4; ++d) g.dims[d] = d < 4 - ndim ? 1 : spatial_shape[d - (4 - ndim)];
  for (int d = 0; d < ndim; ++d) cells *= static_cast<unsigned long long>(spatial_shape[d]);
  int sh, nb_live;
  key_split(cells, &sh, &nb_live);
  g.past = static_cast<uint32_t>(static_cast<unsigned long long>(nb_live) << sh);
  int ubits = 1;                                   // bucket numbers 0 .. nb_live
  while ((nb_live >> ubits) != 0) ++ubits;
  const int passes = div_up(ubits, 9);
  const int nblk = div_up(n, kSortItems);
  Carver cv(ws);
  uint32_t *k0 = cv.take<uint32_t>(n), *kA = cv.take<uint32_t>(n), *kB = cv.take<uint32_t>(n);
  int32_t *vA = cv.take<int32_t>(n), *vB = cv.take<int32_t>(n);
  int32_t *hist = cv.take<int32_t>(static_cast<size_t>(512) * nblk);
  int32_t *hist_off = cv.take<int32_t>(static_cast<size_t>(512) * nblk);
  int32_t *totals = cv.take<int32_t>(512);
  RankMapOut rm{};
  rm.violation = violation;
  rm.rows_in = static_cast<const char *>(rows);
  rm.rows_out = static_cast<char *>(rows_sorted);
  rm.row_bytes = row_bytes;
  unsigned long long zero_units = 0;               // 16-byte units of the rank map (cells, padding, block offsets)
  if (rankmap) {                                   // layout: csrc/rulebook.hip rank_bytes (cells, then one int per 2048 words)
    rm.words = (cells + 31ull) / 32ull;
    rm.cells = static_cast<uint2 *>(rankmap);
    zero_units = (align_up(rm.words * sizeof(uint2), 256) + align_up(((rm.words + 2047ull) / 2048ull) * sizeof(int32_t), 256)) / 16;
  }
  // pass 0 makes the keys while it counts (and clears the flag)
  hipLaunchKernelGGL(key_count_kernel, dim3(nblk), dim3(512), 0, s, indices, n, g, sh, nblk, k0, hist, violation,
                     static_cast<uint4 *>(rankmap), zero_units);
  hipLaunchKernelGGL(radix_scan_kernel, dim3(512), dim3(kBlock), 0, s, hist, hist_off, nblk, totals);
  hipLaunchKernelGGL(radix_scatter_kernel<9>, dim3(nblk), dim3(512), 0, s, k0, static_cast<const int32_t *>(nullptr), n,
                     sh, nblk, hist_off, totals, kA, vA);
  const uint32_t *kin = kA;
  const int32_t *vin = vA;
  for (int pass = 1; pass < passes; ++pass) {
    uint32_t *kout = (pass & 1) ? kB : kA;
    int32_t *vout = (pass & 1) ? vB : vA;
    radix_pass<9>(kin, vin, n, sh + 9 * pass, nblk, hist, hist_off, totals, kout, vout, s);
    kin = kout;
    vin = vout;
  }
  const size_t lds = (static_cast<size_t>(1) << (sh - 5)) * 4 + (static_cast<size_t>(1) << (sh - 8)) * 4;
  static std::atomic<uint64_t> attr_done{0};
  SPX_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(key_bucket_kernel), 144 * 1024, attr_done));
  const int grid = passes == 1 ? nb_live + n / kPartRows + n / kDeadChunk + 2 : nb_live + div_up(n, kDeadChunk);
  hipLaunchKernelGGL(key_bucket_kernel, dim3(grid), dim3(kBucketThreads), lds, s, kin, vin, n, g, sh, nb_live,
                     passes == 1 ? totals : static_cast<const int32_t *>(nullptr), order, indices_sorted,
                     reinterpret_cast<int32_t *>(k0), rm);
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

size_t spx_key_argsort_ws_bytes(int n) { return key_argsort_ws_bytes(n); }

int spx_key_argsort(const int32_t *indices, int n, int ndim, int batch_size, const int *spatial_shape, int32_t *order,
                    int32_t *indices_sorted, void *rankmap, size_t rankmap_bytes, int32_t *violation, const void *rows,
                    void *rows_sorted, int row_bytes, void *ws, size_t ws_bytes, spx_stream_t stream) {
  SPX_CHECK(!rows_sorted || (rows && row_bytes > 0 && row_bytes % 4 == 0), "rows to carry: rows, rows_sorted and a row size that is a multiple of 4");
  SPX_CHECK(ndim >= 1 && ndim <= 4, "ndim must be in [1,4], got %d", ndim);
  SPX_CHECK(n >= 0 && (n == 0 || (indices && order)), "indices and order are required");
  SPX_CHECK(batch_size >= 1, "batch_size must be >= 1, got %d", batch_size);
  unsigned long long cells = static_cast<unsigned long long>(batch_size);
  for (int d = 0; d < ndim; ++d) {
    SPX_CHECK(spatial_shape[d] >= 1, "spatial_shape[%d] must be >= 1", d);
    cells *= static_cast<unsigned long long>(spatial_shape[d]);
    SPX_CHECK(cells <= 0xffe00000ull, "key space of batch x grid exceeds 32 bits");
  }
  SPX_CHECK(ws && ws_bytes >= spx_key_argsort_ws_bytes(n), "workspace too small");
  if (rankmap) {
    const size_t need = spx_rankmap_bytes(ndim, batch_size, spatial_shape);
    SPX_CHECK(need > 0 && rankmap_bytes >= need, "rank map too small (%zu bytes needed)", need);
    SPX_CHECK(indices_sorted, "a rank map describes indices_sorted: pass it");
  }
  if (n == 0) {
    SPX_CHECK(!rankmap, "a rank map of no rows: use spx_rankmap_from_sorted");
    return 0;
  }
  return key_argsort(indices, n, ndim, batch_size, spatial_shape, order, indices_sorted, rankmap, violation, rows,
                     rows_sorted, row_bytes, ws, static_cast<hipStream_t>(stream));
}

}  // extern "C"
