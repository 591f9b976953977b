// Sparse convolution compute kernels for gfx950 (MI355X).
//
//  * gather_gemm_mfma: output-stationary implicit GEMM used for forward and
//    dgrad.  One 256-thread workgroup owns 128 output rows (4 waves x 32 rows)
//    -> every output row is written exactly once, no atomics.  For each kernel
//    offset k present in the tile (OR of the rows' rulebook masks) it gathers
//    the 128 source rows as full 128-byte lines into an XOR-swizzled LDS tile,
//    stages the [Cout x 64] weight slice next to it and issues
//    v_mfma_f32_16x16x32_{f16,bf16}.  Global loads of step t+1 are issued
//    before the MFMAs of step t (register-staged software pipeline).
//  * wgrad_mfma: per (offset, chunk-of-pairs) workgroups contract
//    dout^T (x) feat over the Native pair lists into 64x64 fp32 partials
//    (transposing, XOR-swizzled LDS stores), followed by a deterministic
//    second-stage reduction (no atomics, no split-K races).
//  * generic fp32-accumulate kernels for fp32 tensors and odd channel counts.
//
// Roofline note: at C=K=64 these kernels are HBM/L2-bandwidth bound (SURVEY.md
// section 8d): ~110 MB of compulsory traffic per fwd+bwd at 100k voxels versus
// 2.5 GFLOP, so the design spends its effort on coalesced 128-byte row
// gathers, mask-predicated rulebook reads and single-pass outputs, not on MFMA
// utilisation.
#include <cstring>
#include "igemm_bwd.h"

namespace spx {
namespace {





// --------------------------------------------------------------------------
// generic gather-GEMM: any dtype / channel count, fp32 accumulate.
// one thread per (dst row, out channel).
// --------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float load_f(const T *p);
template <> __device__ __forceinline__ float load_f<float>(const float *p) { return *p; }
struct h16 { uint16_t v; };
struct b16 { uint16_t v; };
template <> __device__ __forceinline__ float load_f<h16>(const h16 *p) { return to_float<false>(p->v); }
template <> __device__ __forceinline__ float load_f<b16>(const b16 *p) { return to_float<true>(p->v); }
template <typename T> __device__ __forceinline__ void store_f(T *p, float f);
template <> __device__ __forceinline__ void store_f<float>(float *p, float f) { *p = f; }
template <> __device__ __forceinline__ void store_f<h16>(h16 *p, float f) { p->v = from_float<false>(f); }
template <> __device__ __forceinline__ void store_f<b16>(b16 *p, float f) { p->v = from_float<true>(f); }

template <typename T>
__global__ void __launch_bounds__(kThreads)
gather_gemm_generic_kernel(GemmParams p) {
  const long long gid = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  const long long total = static_cast<long long>(p.n_dst) * p.COUT;
  if (gid >= total) return;
  const int d = static_cast<int>(gid / p.COUT), n = static_cast<int>(gid % p.COUT);
  const T *A = static_cast<const T *>(p.A);
  const T *B = static_cast<const T *>(p.B);
  const int words = (p.kv + 31) / 32;
  float acc = 0.f;
  for (int k = 0; k < p.kv; ++k) {
    if (p.mask && !((p.mask[static_cast<size_t>(d) * words + (k >> 5)] >> (k & 31)) & 1u)) continue;
    const int idx = (k == p.identity_k) ? d : p.pair[static_cast<size_t>(k) * p.n_dst + d];
    if (idx < 0) continue;
    const int kb = p.b_reverse ? p.kv - 1 - k : k;
    const T *a = A + static_cast<size_t>(idx) * p.CIN;
    const T *b = B + static_cast<size_t>(kb) * p.strideK + static_cast<size_t>(n) * p.strideN;
    for (int c = 0; c < p.CIN; ++c) acc = fmaf(load_f(a + c), load_f(b + c * p.strideD), acc);
  }
  if (p.bias) acc += load_f(static_cast<const T *>(p.bias) + n);
  acc = apply_act(acc, p.act, p.act_alpha);
  store_f(static_cast<T *>(p.out) + static_cast<size_t>(d) * p.COUT + n, acc);
}

// --------------------------------------------------------------------------
// wgrad, 16-bit operands: per (chunk of pairs, offset k, 64x64 dW tile) block.
//   partial[kk][c] = sum_{j in chunk} dout[out_j][kk0+kk] * feat[in_j][c0+c]
// LDS tiles are stored transposed ([channel][pair]) so that the MFMA
// fragments (8 consecutive pairs of one channel) are single ds_read_b128.
// --------------------------------------------------------------------------
constexpr int kWJ = 128;   // pairs staged per iteration

struct WgradParams {
  const void *feat;        // [n_in, C]
  const void *dout;        // [n_out, K]
  float *partial;          // [item][tiles][64*64]
  const int32_t *native;   // [2, kv, n_in]
  const int32_t *num;      // [kv] device counts
  const int32_t *plan;     // see wgrad_plan_kernel
  int n_in, n_out, C, K, kv, subm, chunk, nchunks, tiles_c, tiles_k;
};

__device__ __forceinline__ int list_count(const int32_t *num, int kv, int subm, int n_in, int k) {
  int c;
  if (!subm) c = num[k];
  else if (k == kv / 2) c = n_in;
  else c = k < kv / 2 ? num[k] : num[kv - 1 - k];  // mirror rule, ops.py:962-968
  return c < n_in ? c : n_in;                         // convops.py:1592 clamp
}

// Work plan: the (offset, chunk) items that actually exist, so that the wgrad grid
// holds no empty workgroups.  Layout (int32):
//   [0] total items   [1 .. 1+kv) first item of offset k   [1+kv .. 1+2kv) chunks of offset k
//   [1+2kv + 2 i] = offset of item i,  [2+2kv + 2 i] = first pair of item i
__global__ void __launch_bounds__(kThreads)
wgrad_plan_kernel(const int32_t *__restrict__ num, int n_in, int kv, int subm, int chunk,
                  int32_t *__restrict__ plan) {
  __shared__ int first[129], nch[128];
  const int tid = threadIdx.x;
  if (tid < kv) {
    const int c = list_count(num, kv, subm, n_in, tid);
    nch[tid] = (c + chunk - 1) / chunk;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int k = 0; k < kv; ++k) {
      first[k] = run;
      run += nch[k];
    }
    first[kv] = run;
    plan[0] = run;
  }
  __syncthreads();
  if (tid < kv) {
    plan[1 + tid] = first[tid];
    plan[1 + kv + tid] = nch[tid];
  }
  int32_t *items = plan + 1 + 2 * kv;
  for (int k = 0; k < kv; ++k)
    for (int c = tid; c < nch[k]; c += kThreads) {
      items[2 * (first[k] + c)] = k;
      items[2 * (first[k] + c) + 1] = c * chunk;
    }
}

// element column of pair j in transposed row `ch`: XOR swizzle at 8-element
// granularity (conflict-free fragment reads, <=2-way conflicts on the stores)
__device__ __forceinline__ int wswz(int ch, int j) {
  const int h = (ch & 15) ^ (((ch >> 4) & 3) << 1);
  return j ^ ((h & 15) << 3);
}

__device__ __forceinline__ uint32_t dword_of(const uint4 &v, int i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

template <bool BF16>
__global__ void __launch_bounds__(kThreads)
wgrad_mfma_kernel(WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t *ldsD = reinterpret_cast<uint16_t *>(smem);                    // [64 kk][128 j]
  uint16_t *ldsF = reinterpret_cast<uint16_t *>(smem + kWT * kWJ * 2);    // [64 c ][128 j]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntile = p.tiles_k * p.tiles_c;
  const int total = p.plan[0] * ntile;
  const int32_t *items = p.plan + 1 + 2 * p.kv;
  const uint16_t *F = static_cast<const uint16_t *>(p.feat);
  const uint16_t *D = static_cast<const uint16_t *>(p.dout);
  const int slot = tid & 7;       // 8-channel group
  const int jp0 = tid >> 3;       // pair-of-rows index 0..31 (+32)
  const int wk = wave >> 1, wc = wave & 1;  // wave quadrant: kk [32*wk,+32), c [32*wc,+32)

  for (int work = blockIdx.x; work < total; work += gridDim.x) {
    const int item = work / ntile, tile = work - item * ntile;
    const int k = items[2 * item], begin = items[2 * item + 1];
    const int kk0 = (tile / p.tiles_c) * kWT, c0 = (tile % p.tiles_c) * kWT;
    const int cnt = list_count(p.num, p.kv, p.subm, p.n_in, k);
    const int end = min(cnt, begin + p.chunk);
    const bool identity = p.subm && k == p.kv / 2;
    const int32_t *in_list = p.native + static_cast<size_t>(k) * p.n_in;
    const int32_t *out_list = p.native + static_cast<size_t>(p.kv + k) * p.n_in;
    const bool d_ok = kk0 + slot * 8 < p.K, f_ok = c0 + slot * 8 < p.C;

    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 dv[2][2], fv[2][2];
    auto load_rows = [&](int base) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = base + 2 * (jp0 + 32 * q) + h;
          uint4 d = make_uint4(0, 0, 0, 0), f = make_uint4(0, 0, 0, 0);
          if (j < end) {
            const int oi = identity ? j : out_list[j];
            const int ii = identity ? j : in_list[j];
            if (d_ok) d = *reinterpret_cast<const uint4 *>(D + static_cast<size_t>(oi) * p.K + kk0 + slot * 8);
            if (f_ok) f = *reinterpret_cast<const uint4 *>(F + static_cast<size_t>(ii) * p.C + c0 + slot * 8);
          }
          dv[q][h] = d;
          fv[q][h] = f;
        }
    };
    load_rows(begin);
    for (int base = begin; base < end; base += kWJ) {
      __syncthreads();  // previous iteration's fragment reads are done
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int j = 2 * (jp0 + 32 * q);  // even -> a dword holds pairs (j, j+1)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ch = slot * 8 + e;
          const int sh = (e & 1) * 16;
          const uint32_t dd = ((dword_of(dv[q][0], e >> 1) >> sh) & 0xffffu) |
                              (((dword_of(dv[q][1], e >> 1) >> sh) & 0xffffu) << 16);
          const uint32_t ff = ((dword_of(fv[q][0], e >> 1) >> sh) & 0xffffu) |
                              (((dword_of(fv[q][1], e >> 1) >> sh) & 0xffffu) << 16);
          const int col = wswz(ch, j);
          *reinterpret_cast<uint32_t *>(ldsD + ch * kWJ + col) = dd;
          *reinterpret_cast<uint32_t *>(ldsF + ch * kWJ + col) = ff;
        }
      }
      __syncthreads();
      if (base + kWJ < end) load_rows(base + kWJ);   // in flight during the MFMAs
#pragma unroll
      for (int ks = 0; ks < kWJ / 32; ++ks) {
        const int j8 = (ks * 4 + (lane >> 4)) * 8;
        uint4 fa[2], fb[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int ch = wk * 32 + a * 16 + (lane & 15);
          fa[a] = *reinterpret_cast<const uint4 *>(ldsD + ch * kWJ + wswz(ch, j8));
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int ch = wc * 32 + b * 16 + (lane & 15);
          fb[b] = *reinterpret_cast<const uint4 *>(ldsF + ch * kWJ + wswz(ch, j8));
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = mfma16<BF16>(fa[a], fb[b], acc[a][b]);
      }
    }
    // D[i = kk][j = c]: lane holds c = lane & 15, kk = (lane >> 4) * 4 + reg
    float *dst = p.partial + static_cast<size_t>(work) * (kWT * kWT);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = wk * 32 + a * 16 + (lane >> 4) * 4 + e;
          const int c = wc * 32 + b * 16 + (lane & 15);
          if (SPX_AUX_OUT) __builtin_nontemporal_store(acc[a][b][e], &dst[kk * kWT + c]);
          else dst[kk * kWT + c] = acc[a][b][e];
        }
    __syncthreads();  // LDS is rewritten by the next work item
  }
}

// generic wgrad partial: thread per (kk, c) of the 64x64 tile, 16 elems/thread
template <typename T>
__global__ void __launch_bounds__(kThreads)
wgrad_generic_kernel(WgradParams p) {
  const int ntile = p.tiles_k * p.tiles_c;
  const int total = p.plan[0] * ntile;
  const int32_t *items = p.plan + 1 + 2 * p.kv;
  const T *F = static_cast<const T *>(p.feat);
  const T *D = static_cast<const T *>(p.dout);
  for (int work = blockIdx.x; work < total; work += gridDim.x) {
    const int item = work / ntile, tile = work - item * ntile;
    const int k = items[2 * item], begin = items[2 * item + 1];
    const int kk0 = (tile / p.tiles_c) * kWT, c0 = (tile % p.tiles_c) * kWT;
    const int cnt = list_count(p.num, p.kv, p.subm, p.n_in, k);
    const int end = min(cnt, begin + p.chunk);
    const bool identity = p.subm && k == p.kv / 2;
    const int32_t *in_list = p.native + static_cast<size_t>(k) * p.n_in;
    const int32_t *out_list = p.native + static_cast<size_t>(p.kv + k) * p.n_in;
    float *dst = p.partial + static_cast<size_t>(work) * (kWT * kWT);
    for (int e = threadIdx.x; e < kWT * kWT; e += kThreads) {
      const int kk = kk0 + e / kWT, c = c0 + e % kWT;
      float acc = 0.f;
      if (kk < p.K && c < p.C) {
        for (int j = begin; j < end; ++j) {
          const int oi = identity ? j : out_list[j];
          const int ii = identity ? j : in_list[j];
          acc = fmaf(load_f(D + static_cast<size_t>(oi) * p.K + kk),
                     load_f(F + static_cast<size_t>(ii) * p.C + c), acc);
        }
      }
      dst[e] = acc;
    }
  }
}

// dw[kk][k][c] = sum over the items of offset k (fixed order -> deterministic).
// 512 threads = 32 consecutive elements x 16 item groups; every thread keeps 8 loads in flight.
constexpr int kRedThreads = 512;
constexpr int kRedSplit = 16;
constexpr int kRedElems = kRedThreads / kRedSplit;  // 32

template <typename T>
__global__ void __launch_bounds__(kRedThreads)
wgrad_reduce_kernel(WgradParams p, T *__restrict__ dw) {
  __shared__ float red[kRedThreads];
  const int k = blockIdx.y;
  const int first = p.plan[1 + k], nch = p.plan[1 + p.kv + k];
  const int grp = threadIdx.x / kRedElems, el = threadIdx.x % kRedElems;
  const int ntile = p.tiles_k * p.tiles_c;
  const int e_global = blockIdx.x * kRedElems + el;  // element of [tiles][64*64]
  const int tile = e_global / (kWT * kWT), e = e_global % (kWT * kWT);
  const size_t stride = static_cast<size_t>(ntile) * (kWT * kWT);
  float acc = 0.f;
  if (tile < ntile) {
    const float *src = p.partial + (static_cast<size_t>(first) * ntile + tile) * (kWT * kWT) + e;
    int ch = grp;
    for (; ch + 7 * kRedSplit < nch; ch += 8 * kRedSplit) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[static_cast<size_t>(ch + u * kRedSplit) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; ch < nch; ch += kRedSplit) acc += src[static_cast<size_t>(ch) * stride];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (grp == 0 && tile < ntile) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < kRedSplit; ++g) s += red[g * kRedElems + el];
    const int kk = (tile / p.tiles_c) * kWT + e / kWT, c = (tile % p.tiles_c) * kWT + e % kWT;
    if (kk < p.K && c < p.C) store_f(dw + (static_cast<size_t>(kk) * p.kv + k) * p.C + c, s);
  }
}


// exclusive prefix sum of one value per thread over the whole kW2MaxG-thread block (wave scans + the
// wave totals through LDS); returns the prefix, `total` = sum over the block.  One barrier.  The plan
// packs several small counters into one 64-bit value per scan.
template <typename T>
__device__ __forceinline__ T plan_scan(T v, T *wtot, T &total) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  T incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const T u = __shfl_up(incl, d, 64);
    if (lane >= d) incl += u;
  }
  if (lane == 63) wtot[wv] = incl;
  __syncthreads();
  T prefix = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < kW2MaxG / 64; ++w) {
    const T t = wtot[w];
    if (w < wv) prefix += t;
    total += t;
  }
  return prefix + incl - v;
}

// Segment cost padding of the plan (see wgrad_plan2_kernel): on unless switched off for an A/B run.
__device__ __forceinline__ bool subm_cost_pad(int n_in, int kv) {
  (void)n_in;
  (void)kv;
#ifdef SPX_WGRAD_NO_PAD
  return false;
#else
  return true;
#endif
}

// One block, three block-wide scans.  Everything is a prefix sum or a closed form of the kv list
// lengths: the pairs of all lists, laid end to end, are cut into G equal ranges (one per workgroup
// of the first stage); a segment is the part of one list inside one range.  A range finds its first
// list by bisection and walks on from there (1-2 lists at kv = 27), the number of ranges that touch
// list k is floor((end-1)/per) - floor(start/per) + 1.
//
// Which workgroup takes which range: workgroup b runs on XCD b % 8 (round-robin dispatch) and all
// ranges advance through their lists at about the same rate, so the ranges that start in the same
// eighth of their list -- the same eighth of the ROWS, every list being in row order -- go to the
// same XCD: its L2 then serves a row to the other offsets that use it, instead of every XCD pulling
// it out of the Infinity Cache once per offset.  rec[5] of workgroup b = its range.
__global__ void __launch_bounds__(kW2MaxG)
wgrad_plan2_kernel(const int32_t *__restrict__ num, int n_in, int kv, int subm, int G,
                   int32_t *__restrict__ plan) {
  typedef unsigned long long u64;
  __shared__ int start[130], lpad[130], kfirst[130], kcount[130], ritems[130];
  __shared__ int wtot_i[kW2MaxG / 64];
  __shared__ u64 wtot_a[kW2MaxG / 64], wtot_b[kW2MaxG / 64];
  const int tid = threadIdx.x;
  const int c = tid < kv ? list_count(num, kv, subm, n_in, tid) : 0;
  // The lists are laid end to end in COST units: a list of c pairs takes c + ov of them, the first ov being
  // the fixed price of a segment (pair words -> rows -> first MFMAs before anything overlaps, the partial
  // tile it writes).  Cutting by pairs alone put the 26 short lists of a sparse SubM rulebook (~120 pairs
  // each at BASELINE config 2) into a handful of ranges of 3-4 segments -- 3-4 dependent chains in a row,
  // the slowest workgroups of the launch; with the padding a short list fills most of a range by itself.
  const int ov = (c > 0 && subm_cost_pad(n_in, kv)) ? kW2J + kW2J / 2 : 0;
  int total;
  const int st = plan_scan<int>(c + ov, wtot_i, total);
  if (tid <= kv) start[tid] = st;               // start[kv] = total (threads >= kv add nothing)
  if (tid < kv) lpad[tid] = ov;
  const int per = total > 0 ? (total + G - 1) / G : 1;
  __syncthreads();
  // pairs of list k inside the cost interval [a, b): list positions [pa, pb)
  auto pairs_in = [&](int k, int a, int b, int &pa, int &pb) {
    const int p0 = start[k] + lpad[k], len = start[k + 1] - p0;
    pa = min(max(a - p0, 0), len);
    pb = min(max(b - p0, 0), len);
  };

  // ---- ranges: first list, number of segments, XCD
  int lo = 0, hi = 0, mine = 0, k0 = 0, x = 0;
  if (tid < G) {
    lo = min(total, tid * per);
    hi = min(total, lo + per);
    if (hi > lo) {
      int a = 0, b = kv;                        // smallest k with start[k + 1] > lo
      while (a < b) {
        const int m = (a + b) >> 1;
        if (start[m + 1] > lo) b = m;
        else a = m + 1;
      }
      k0 = a;
      for (int k = k0; k < kv && start[k] < hi; ++k) {
        int pa, pb;
        pairs_in(k, lo, hi, pa, pb);
        mine += pb > pa ? 1 : 0;
      }
      int pa, pb;
      pairs_in(k0, lo, hi, pa, pb);
      const int len = start[k0 + 1] - start[k0] - lpad[k0];
      x = min(kXcds - 1, static_cast<int>(static_cast<long long>(pa) * kXcds / (len > 0 ? len : 1)));
    }
  }
  // ---- per offset: segments (= ranges touching the list), second-stage items
  int kc = 0, nitems = 0;
  if (tid < kv) {
    if (c > 0) kc = (start[tid + 1] - 1) / per - (start[tid] + ov) / per + 1;   // ranges touching the PAIRS of the list
    // second-stage work list: block shape by segment count (see wgrad_reduce2_kernel)
    const int mode = kc >= 48 ? 0 : (kc >= 6 ? 1 : 2);
    nitems = (kWT * kWT) / (mode == 0 ? 16 : (mode == 1 ? 128 : 512));
  }
  // scan A: segments per range (11 bits: <= G + kv) | ranges of XCD 0..4 (10 bits each)
  // scan B: ranges of XCD 5..7 (10 bits each) | segments per offset (11 bits, bit 30) | items (bit 41)
  u64 va = 0, vb = 0;
  if (tid < G) {
    va = static_cast<u64>(mine);
    if (x < 5) va |= 1ull << (11 + 10 * x);
    else vb = 1ull << (10 * (x - 5));
  }
  vb |= (static_cast<u64>(kc) << 30) | (static_cast<u64>(nitems) << 41);
  u64 ta, tb;
  const u64 ea = plan_scan<u64>(va, wtot_a, ta);
  const u64 eb = plan_scan<u64>(vb, wtot_b, tb);
  const int seg0 = static_cast<int>(ea & 0x7ff), nseg_total = static_cast<int>(ta & 0x7ff);
  const int kf = static_cast<int>((eb >> 30) & 0x7ff), ri = static_cast<int>(eb >> 41);
  const int kc_total = static_cast<int>((tb >> 30) & 0x7ff), items_total = static_cast<int>(tb >> 41);
  if (tid == 0) {
    plan[0] = nseg_total;
    plan[1] = per;
    plan[plan2_red(G, kv)] = items_total;
  }
  if (tid <= kv) {
    kfirst[tid] = kf;
    kcount[tid] = kc;
    ritems[tid] = ri;
    plan[plan2_kf(G) + tid] = tid == kv ? kc_total : kf;
  }
  int32_t *seg = plan + plan2_seg(G, kv);
  if (tid < G) {
    int sg = seg0;
    int32_t *rec = plan + plan2_wg(tid);
    int r2 = 0, r3 = 0, r4 = 0;
    bool first = true;
    if (hi > lo) {
      for (int k = k0; k < kv && start[k] < hi; ++k) {
        int a, b;
        pairs_in(k, lo, hi, a, b);
        if (b > a) {
          seg[3 * sg] = k;
          seg[3 * sg + 1] = a;
          seg[3 * sg + 2] = b;
          if (first) {
            r2 = k;
            r3 = a;
            r4 = b;
            first = false;
          }
          ++sg;
        }
      }
    }
    rec[0] = seg0;
    rec[1] = mine;
    rec[2] = r2;
    rec[3] = r3;
    rec[4] = r4;
    // position of this range in (XCD, rank) order -> the workgroup at the same position in
    // (b % 8, b / 8) order; XCD y owns ceil((G - y) / 8) workgroups
    int pos = x < 5 ? static_cast<int>((ea >> (11 + 10 * x)) & 0x3ff) : static_cast<int>((eb >> (10 * (x - 5))) & 0x3ff);
#pragma unroll
    for (int y = 0; y < kXcds; ++y) {
      const int cy = y < 5 ? static_cast<int>((ta >> (11 + 10 * y)) & 0x3ff) : static_cast<int>((tb >> (10 * (y - 5))) & 0x3ff);
      if (y < x) pos += cy;
    }
    int y = 0, base = 0;
    for (; y < kXcds - 1; ++y) {
      const int ny = (G - y + kXcds - 1) / kXcds;
      if (pos < base + ny) break;
      base += ny;
    }
    const int b = y + kXcds * (pos - base);
    plan[plan2_wg(b < G ? b : tid) + 5] = tid;
  }
  __syncthreads();
  int32_t *rl = plan + plan2_red(G, kv);
  for (int k = 0; k < kv; ++k) {
    const int md = kcount[k] >= 48 ? 0 : (kcount[k] >= 6 ? 1 : 2);
    const int E = md == 0 ? 16 : (md == 1 ? 128 : 512), cnt_items = (kWT * kWT) / E;
    for (int q = tid; q < cnt_items; q += kW2MaxG) {
      int32_t *item = rl + 4 + 4 * (ritems[k] + q);
      item[0] = k | (md << 8);
      item[1] = q * E;
      item[2] = kfirst[k];
      item[3] = kcount[k];
    }
  }
}


template <bool BF16, int SL = 8>
__global__ void __launch_bounds__(kThreads)
wgrad_tr_kernel(Wgrad2Params p) {
  wgrad_tr_body<BF16, 2, SL>(p, blockIdx.x);
}


__global__ void __launch_bounds__(kThreads)
wgrad_f32_kernel(Wgrad2Params p) {
  wgrad_f32_body(p, blockIdx.x);
}


// dw[kk][k][c] = sum over the segments of offset k (deterministic: fixed assignment of
// segments to threads, fixed summation order).  The work list (built with the plan) gives
// every offset a block shape that fits its segment count -- 16 elements x 128 segment groups
// for long lists, 128 x 4 or 512 x 1 for short ones -- so a SubM rulebook with one long and
// 26 short lists runs ~460 blocks of useful work instead of 27 x 128 mostly idle ones.
// (one layer's second stage: the job of a workgroup column `tile`; shared by the one-layer launch and the batched one)
struct Wgrad2Job {
  const float *partial;    // [segment][tile][64*64]
  const int32_t *plan2;
  void *dw;
  int G, kv, tiles_k, tiles_c, K, C;
};

template <typename T>
__device__ __forceinline__ void wgrad_reduce2_body(const Wgrad2Job &p, T *__restrict__ dw, const int tile, float *red) {
  const int32_t *__restrict__ rl = p.plan2 + plan2_red(p.G, p.kv);
  const int nitems = rl[0];
  const int ntile = p.tiles_k * p.tiles_c;
  const size_t stride = static_cast<size_t>(ntile) * (kWT * kWT);
  for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
    const int4 item = *reinterpret_cast<const int4 *>(rl + 4 + 4 * it);    // uniform: one s_load_dwordx4
    const int k = item.x & 0xff, mode = item.x >> 8, e0 = item.y, first = item.z, nseg = item.w;
    const float *base = p.partial + static_cast<size_t>(first) * stride + tile * (kWT * kWT) + e0;
    if (mode == 0) {
      // long list: 16 elements (4 lanes x float4) x 128 segment groups; with a few hundred
      // segments every thread has all of its loads in flight at once.  Wave-level butterfly
      // over the 16 groups of a wave, then 8 wave sums through LDS.
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const int c4 = lane & 3, grp = wave * 16 + (lane >> 2);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const float *src = base + c4 * 4;
      int ch = grp;
      for (; ch + 3 * 128 < nseg; ch += 4 * 128) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[u] = *reinterpret_cast<const float4 *>(src + static_cast<size_t>(ch + u * 128) * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
        }
      }
      for (; ch < nseg; ch += 128) {
        const float4 v = *reinterpret_cast<const float4 *>(src + static_cast<size_t>(ch) * stride);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
#pragma unroll
      for (int d = 4; d < 64; d <<= 1) {
        acc.x += __shfl_xor(acc.x, d, 64);
        acc.y += __shfl_xor(acc.y, d, 64);
        acc.z += __shfl_xor(acc.z, d, 64);
        acc.w += __shfl_xor(acc.w, d, 64);
      }
      if (lane < 4) reinterpret_cast<float4 *>(red)[wave * 4 + lane] = acc;
      __syncthreads();
      if (threadIdx.x < 16) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < kRedThreads / 64; ++w) sum += red[w * 16 + threadIdx.x];
        const int ee = e0 + threadIdx.x;
        const int kk = (tile / p.tiles_c) * kWT + ee / kWT, c = (tile % p.tiles_c) * kWT + ee % kWT;
        if (kk < p.K && c < p.C) store_f(dw + (static_cast<size_t>(kk) * p.kv + k) * p.C + c, sum);
      }
      __syncthreads();   // red[] is reused by the next item
      continue;
    }
    // short lists: 128 elements x 4 groups, or 512 x 1
    const int E = mode == 1 ? 128 : 512, S = kRedThreads / E;
    const int grp = threadIdx.x / E, el = threadIdx.x % E;
    float acc = 0.f;
    for (int ch = grp; ch < nseg; ch += S) acc += base[static_cast<size_t>(ch) * stride + el];
    if (S > 1) {
      red[threadIdx.x] = acc;
      __syncthreads();
      if (grp == 0) {
        acc = 0.f;
        for (int g = 0; g < S; ++g) acc += red[g * E + el];
      }
    }
    if (grp == 0) {
      const int ee = e0 + el;
      const int kk = (tile / p.tiles_c) * kWT + ee / kWT, c = (tile % p.tiles_c) * kWT + ee % kWT;
      if (kk < p.K && c < p.C) store_f(dw + (static_cast<size_t>(kk) * p.kv + k) * p.C + c, acc);
    }
    if (S > 1) __syncthreads();   // red[] is reused by the next item
  }
}


template <typename T>
__global__ void __launch_bounds__(kRedThreads)
wgrad_reduce2_kernel(Wgrad2Params p, T *__restrict__ dw) {
  __shared__ float red[kRedThreads];
  const Wgrad2Job j{p.partial, p.plan2, dw, p.G, p.kv, p.tiles_k, p.tiles_c, p.K, p.C};
  wgrad_reduce2_body<T>(j, dw, static_cast<int>(blockIdx.y), red);
}

// The second stages of SEVERAL layers in one launch (spx_wgrad_stage2_batch): a backward pass of a network reduces
// every layer's partial tiles at its end instead of behind each layer -- a 4-9 us launch per layer leaves the chain.
// blockIdx.y = tile columns of job 0, then of job 1, ...; blockIdx.x strides over a job's work list as above.
constexpr int kStage2MaxJobs = 16;
struct Wgrad2Batch {
  int n;
  int ybase[kStage2MaxJobs + 1];
  Wgrad2Job job[kStage2MaxJobs];
};

template <typename T>
__global__ void __launch_bounds__(kRedThreads)
wgrad_reduce2_batch_kernel(Wgrad2Batch b) {
  __shared__ float red[kRedThreads];
  int j = 0;
  while (j + 1 < b.n && static_cast<int>(blockIdx.y) >= b.ybase[j + 1]) ++j;      // (uniform: scalar compares)
  const Wgrad2Job job = b.job[j];
  wgrad_reduce2_body<T>(job, static_cast<T *>(job.dw), static_cast<int>(blockIdx.y) - b.ybase[j], red);
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
bias_act_kernel(T *__restrict__ out, const T *__restrict__ bias, long long total, int K, int act,
                float alpha) {
  const long long gid = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (gid >= total) return;
  float v = load_f(out + gid);
  if (bias) v += load_f(bias + gid % K);
  store_f(out + gid, apply_act(v, act, alpha));
}

int elem_bytes(int dtype) { return dtype == SPX_F32 ? 4 : (dtype == SPX_I8 ? 1 : 2); }

bool mfma_ok(int dtype, int cin, int cout, int kv, const uint32_t *mask) {
  if (dtype != SPX_F16 && dtype != SPX_BF16 && dtype != SPX_F32) return false;
  if (cin % (dtype == SPX_F32 ? 4 : 8) != 0) return false;     // 16-byte lane pieces
  if (kv > 128) return false;                                  // 33 .. 128: groups of 32 offsets
  (void)mask;
  return cout == 16 || cout == 32 || cout == 64 || cout == 128 || cout == 256;
}



int run_gather_gemm_single(const GemmParams &p, int dtype, hipStream_t s);

// Kernel volumes 33 .. 128 (5x5x5, 4-d 3^4, ...): the reference covers them with multi-word masks
// (indices.py:1601-1618, ops.py:448,494-503); here the layer runs as ceil(kv / 32) launches of the same
// kernel, one mask word each, whose partial sums travel through an fp32 [n_dst, COUT] scratch -- the
// result is rounded once, like a single launch.
int run_gather_gemm(const GemmParams &p, int dtype, hipStream_t s) {
  if (p.n_dst == 0) return 0;
  if (p.kv <= 32 || !mfma_ok(dtype, p.CIN, p.COUT, p.kv, p.mask) || !p.pair) return run_gather_gemm_single(p, dtype, s);
  const int words = div_up(p.kv, 32);
  const bool fits = static_cast<unsigned long long>(p.n_dst) * 4ull * words < 0x7fff0000ull &&
                    static_cast<unsigned long long>(p.n_dst) * p.COUT * 4ull < 0x7fff0000ull &&
                    v4_ok(p, dtype == SPX_F32 ? 4 : 2, dtype == SPX_F32 ? 4 : 2) && !p.argsort;
  if (!fits || !p.acc) {
    GemmParams q = p;                 // no scratch / beyond 32-bit offsets: the generic kernel
    q.acc = nullptr;
    return run_gather_gemm_single(q, dtype, s);
  }
  for (int g = 0; g < words; ++g) {
    GemmParams q = p;
    q.kbase = 32 * g;
    q.pair = p.pair + static_cast<size_t>(32 * g) * p.n_dst;
    q.mask = p.mask ? p.mask + g : nullptr;
    q.mask_words = p.mask ? words : 1;
    q.identity_k = (p.identity_k >= 32 * g && p.identity_k < 32 * g + 32) ? p.identity_k - 32 * g : -1;
    q.acc_mode = (g > 0 ? 1 : 0) | (g < words - 1 ? 2 : 0);
    if (int rc = run_gather_gemm_single(q, dtype, s)) return rc;
  }
  return 0;
}

int run_gather_gemm_single(const GemmParams &p, int dtype, hipStream_t s) {
  if (p.n_dst == 0) return 0;
  constexpr int f32_mfma = 1;
  const bool grouped = p.acc_mode != 0;
  if (dtype == SPX_F32 && f32_mfma && mfma_ok(dtype, p.CIN, p.COUT, p.kv, p.mask) && v4_ok(p, 4, 4) &&
      (p.kv <= 32 || grouped))
    return dispatch_gather_gemm_f32(p, s);
  if (dtype != SPX_F32 && mfma_ok(dtype, p.CIN, p.COUT, p.kv, p.mask) && (p.kv <= 32 || grouped))
    return dtype == SPX_BF16 ? dispatch_gather_gemm_bf16(p, s) : dispatch_gather_gemm<false>(p, s);
  if (p.cls) {                     // (as above: the generic kernel reads the tables by row)
    GemmParams q = p;
    drop_rows_layout(q);
    return run_gather_gemm_single(q, dtype, s);
  }
  if (p.tile_order) {
    set_error("tables in tile order are supported by the MFMA kernels only (channel counts / kernel volume)");
    return -1;
  }
  const long long total = static_cast<long long>(p.n_dst) * p.COUT;
  const dim3 grid(static_cast<unsigned>((total + kThreads - 1) / kThreads));
  count_launch(kFamGeneric);
  if (dtype == SPX_F32)
    hipLaunchKernelGGL(gather_gemm_generic_kernel<float>, grid, dim3(kThreads), 0, s, p);
  else if (dtype == SPX_F16)
    hipLaunchKernelGGL(gather_gemm_generic_kernel<h16>, grid, dim3(kThreads), 0, s, p);
  else if (dtype == SPX_BF16)
    hipLaunchKernelGGL(gather_gemm_generic_kernel<b16>, grid, dim3(kThreads), 0, s, p);
  else {
    set_error("unsupported dtype %d", dtype);
    return -1;
  }
  SPX_LAUNCH_CHECK();
  return 0;
}

int wgrad_chunk(int n_in) {
  constexpr int forced = 0;
  if (forced > 0) return (forced + kWJ - 1) / kWJ * kWJ;
  // aim at >= ~512 workgroups for the dominant (centre) list, multiples of 128
  int c = (n_in / 512 + kWJ - 1) / kWJ * kWJ;
  if (c < kWJ) c = kWJ;
  if (c > 1024) c = 1024;
  return c;
}

size_t wgrad_plan_ints(int n_in, int kv) {
  const size_t nchunks = div_up(n_in > 0 ? n_in : 1, wgrad_chunk(n_in));
  return 1 + 2 * static_cast<size_t>(kv) + 2 * nchunks * kv;
}

// workgroups of the balanced wgrad: 1.5 per CU once there is enough work (more workgroups
// mean more partials for the second stage: 384 measured best at 100k voxels), never more
// ranges than twice the 128-pair chunks of the identity list
int wgrad_groups(int n_in, int subm) {
  constexpr int forced = 0;
  int g = forced > 0 ? forced : 384;
  if (forced <= 0) {
    // backward shares its launch with ceil(n / 128) dgrad tiles: when both halves fit the 1024
    // resident workgroup slots of the chip together there is no second dispatch round
    // (41.3 vs 44.7 us per step at 100 k voxels).  Beyond ~115 k voxels 384 stays: 128 ranges are
    // 7-13 % faster at C = 64 (the dgrad tiles alone fill the slots there), but the ranges become
    // chains of > 100 chunks and double the launch time of 16 / 32-channel layers, which is what
    // the large levels of a backbone are (igemm_bwd_kernel<32>: 158 -> 294 us at 450 k voxels);
    // the plan is built per rulebook, without knowing the layer widths that will use it
    // SubM rulebooks of that size carry a rows layout (spx_subm_layout): its appendix tiles lead the dgrad half of the
    // launch, kAppBudget of them at most (config 2: 50 tiles of 64 rows; step 25.7 -> 24.2-24.9 us with the room left)
    const int room = 1024 - div_up(n_in > 0 ? n_in : 1, 128) - ((subm && n_in >= kLayoutMinRows) ? kAppBudget : 0);
    if (room >= 128 && room < g) g = room;
  }
  const int chunks = div_up(n_in > 0 ? n_in : 1, 128);
  if (g > 2 * chunks) g = 2 * chunks;
  if (g > kW2MaxG - 1) g = kW2MaxG - 1;   // the plan kernel needs thread G for the end marker
  return g < 1 ? 1 : g;
}

// blocks of the second stage (block-stride over at most kv * 256 items)
int reduce2_blocks(int kv) {
  constexpr int cap = 512;
  return kv * 256 < cap ? kv * 256 : cap;
}

// host form of a deferred second stage (include/spconv_amd.h SPX_STAGE2_JOB_BYTES: opaque to the caller)
struct Stage2JobH {
  const float *partial;
  const int32_t *plan2;
  void *dw;
  int G, kv, tiles_k, tiles_c, K, C, dtype, valid;
};
static_assert(sizeof(Stage2JobH) <= SPX_STAGE2_JOB_BYTES, "job record");

// the second stage of one layer: launched here, or -- `defer` -- written down for spx_wgrad_stage2_batch
int launch_reduce2(const Wgrad2Params &q, void *dw, int dtype, int ntile, hipStream_t s, void *defer) {
  if (defer) {
    Stage2JobH j{q.partial, q.plan2, dw, q.G, q.kv, q.tiles_k, q.tiles_c, q.K, q.C, dtype, 1};
    memcpy(defer, &j, sizeof(j));
    return 0;
  }
  const dim3 rgrid2(reduce2_blocks(q.kv), ntile);   // block-stride over the work list
  count_launch(kFamStage2);
  if (dtype == SPX_F32)
    hipLaunchKernelGGL(wgrad_reduce2_kernel<float>, rgrid2, dim3(kRedThreads), 0, s, q, static_cast<float *>(dw));
  else if (dtype == SPX_F16)
    hipLaunchKernelGGL(wgrad_reduce2_kernel<h16>, rgrid2, dim3(kRedThreads), 0, s, q, static_cast<h16 *>(dw));
  else
    hipLaunchKernelGGL(wgrad_reduce2_kernel<b16>, rgrid2, dim3(kRedThreads), 0, s, q, static_cast<b16 *>(dw));
  SPX_LAUNCH_CHECK();
  return 0;
}

int wgrad_xcd_order() {
  constexpr int v = 1;
  return v;
}

// what the *_bytes functions size for: the larger of the two rules.  (They used to take wgrad_groups(n, 0) as "an
// upper bound of the SubM value", which the appendix budget broke for n in (106 496, 114 688]: 164 vs 384 ranges at
// n = 110 000 -- at small kernel volumes the plan kernel then wrote past the buffer; round-4 ADVICE.)
int wgrad_groups_max(int n_in) {
  const int a = wgrad_groups(n_in, 0), b = wgrad_groups(n_in, 1);
  return a > b ? a : b;
}

size_t wgrad_plan2_ints(int n_in, int kv) {
  const size_t G = wgrad_groups_max(n_in);
  return 8 + kW2Rec * G + kv + 1 + 3 * (G + kv) + 4 + 4 + 4 * static_cast<size_t>(kv) * 256 + 8;
}

GemmParams dgrad_params(const void *dout, const void *weight, void *din, const int32_t *pair,
                        const uint32_t *mask, const int32_t *argsort, int n_out, int n_in, int C,
                        int K, int kv, int subm) {
  GemmParams p{};
  p.A = dout;
  p.B = weight;                                   // KRSC read in place: (k, n=c, d=kk)
  p.out = din;
  p.pair = pair;
  p.mask = mask;
  p.argsort = argsort;
  p.bias = nullptr;
  p.strideK = C;
  p.strideN = 1;
  p.strideD = static_cast<long long>(kv) * C;
  p.n_src = n_out;
  p.n_dst = n_in;
  p.CIN = K;
  p.COUT = C;
  p.kv = kv;
  p.identity_k = subm ? kv / 2 : -1;
  p.b_reverse = subm ? 1 : 0;
  p.act = SPX_ACT_NONE;
  p.act_alpha = 0.f;
  return p;
}



}  // namespace
}  // namespace spx

using namespace spx;

namespace spx {
namespace {
// rows of `sw` 16-bit words copied into rows of `dw` >= sw words, the tail zero-filled: the channel padding of a layer
// whose width the MFMA kernels are not instantiated for (a backbone's 3-5 channel first layer) in ONE launch -- torch's
// pad is a fill and a copy
__global__ void __launch_bounds__(kThreads)
pad_rows_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, long long total, int sw, int dw) {
  const long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= total) return;
  const long long r = i / dw;
  const int c = static_cast<int>(i - r * dw);
  dst[i] = c < sw ? src[r * sw + c] : static_cast<uint16_t>(0);
}
}  // namespace
}  // namespace spx

extern "C" {

size_t spx_igemm_acc_bytes(int n_dst, int cout, int kv) {
  return kv > 32 ? align_up(static_cast<size_t>(n_dst > 0 ? n_dst : 1) * cout * sizeof(float), 256) : 0;
}

static int igemm_fwd_impl(const void *feat, const void *weight, void *out, const int32_t *pair,
                          const uint32_t *mask, const int32_t *argsort, int tile_order, int n_in, int n_out, int C,
                          int K, int kv, int dtype, int identity_k, const void *bias, int act,
                          float act_alpha, void *ws, size_t ws_bytes, spx_stream_t stream, float *stats,
                          const int32_t *n_live, int *slots_used_h) {
  if (slots_used_h) *slots_used_h = 0;
  SPX_CHECK(C > 0 && K > 0 && kv > 0 && n_in >= 0 && n_out >= 0, "bad sizes");
  if (n_out == 0) return 0;                                   // empty scene: nothing to write
  SPX_CHECK((feat || n_in == 0) && weight && out, "null tensor pointer");
  SPX_CHECK(pair || kv == 1, "pair table required");
  GemmParams p{};
  p.A = feat;
  p.B = weight;
  p.out = out;
  p.pair = pair;
  p.mask = mask;
  p.argsort = argsort;
  p.bias = bias;
  p.strideK = C;                                  // KRSC: W[n][k][c]
  p.strideN = static_cast<long long>(kv) * C;
  p.strideD = 1;
  p.n_src = n_in;
  p.n_dst = n_out;
  p.CIN = C;
  p.COUT = K;
  p.kv = kv;
  p.identity_k = identity_k;
  p.b_reverse = 0;
  p.dense_hint = (tile_order & SPX_DENSE_HINT) ? 1 : 0;
  apply_rows_layout(p, tile_order & ~SPX_DENSE_HINT);
  p.act = act & 0xff;
  if (act & SPX_OUT_CACHED) p.dbg = 0x400;       // plain result stores: the next launch reads the rows
  p.act_alpha = act_alpha;
  if (ws && ws_bytes >= spx_igemm_acc_bytes(n_out, K, kv) && kv > 32) p.acc = static_cast<float *>(ws);
  if (stats && kv <= 32 && !bias && p.act == SPX_ACT_NONE) {   // (the training-mode call: plain rows)
    p.stats = stats;
    p.n_live = n_live;
    p.grid_out = slots_used_h;
  }
  return run_gather_gemm(p, dtype, static_cast<hipStream_t>(stream));
}

int spx_igemm_fwd(const void *feat, const void *weight, void *out, const int32_t *pair,
                  const uint32_t *mask, const int32_t *argsort, int tile_order, int n_in, int n_out, int C,
                  int K, int kv, int dtype, int identity_k, const void *bias, int act,
                  float act_alpha, void *ws, size_t ws_bytes, spx_stream_t stream) {
  return igemm_fwd_impl(feat, weight, out, pair, mask, argsort, tile_order, n_in, n_out, C, K, kv, dtype, identity_k,
                        bias, act, act_alpha, ws, ws_bytes, stream, nullptr, nullptr, nullptr);
}

int spx_igemm_fwd_stats_slots(int n_out) {
  // upper bound of the workgroups of a forward launch over n_out rows: 64-row tiles + the appendix workgroups the
  // class rule allows (n / 4 rows) + slack
  return n_out <= 0 ? 0 : div_up(n_out, 64) + div_up(n_out, 256) + 8;
}

int spx_igemm_fwd_stats(const void *feat, const void *weight, void *out, const int32_t *pair,
                        const uint32_t *mask, const int32_t *argsort, int tile_order, int n_in, int n_out, int C,
                        int K, int kv, int dtype, int identity_k, const void *bias, int act,
                        float act_alpha, void *ws, size_t ws_bytes, float *stats, int stats_slots,
                        const int32_t *n_live, int *slots_used_h, spx_stream_t stream) {
  SPX_CHECK(slots_used_h, "slots_used_h is required");
  SPX_CHECK(!stats || stats_slots >= spx_igemm_fwd_stats_slots(n_out), "statistics buffer too small: %d slots < %d",
            stats_slots, spx_igemm_fwd_stats_slots(n_out));
  return igemm_fwd_impl(feat, weight, out, pair, mask, argsort, tile_order, n_in, n_out, C, K, kv, dtype, identity_k,
                        bias, act, act_alpha, ws, ws_bytes, stream, stats, n_live, slots_used_h);
}

int spx_igemm_fwd_int8(const void *feat, const void *weight, void *out, const int32_t *pair,
                       const uint32_t *mask, const int32_t *argsort, int n_in, int n_out, int C,
                       int K, int kv, int identity_k, const float *scale, const float *bias,
                       const void *add, float add_scale, int out_dtype, int act, float act_alpha,
                       spx_stream_t stream) {
  SPX_CHECK(C > 0 && K > 0 && kv > 0 && n_in >= 0 && n_out >= 0, "bad sizes");
  if (n_out == 0) return 0;                                   // empty scene: nothing to write
  SPX_CHECK((feat || n_in == 0) && weight && out, "null tensor pointer");
  SPX_CHECK(pair || kv == 1, "pair table required");
  // the reference has the same restriction (test/test_all_algo.py:376-377)
  SPX_CHECK(C % 16 == 0, "int8 needs in_channels %% 16 == 0, got %d", C);
  SPX_CHECK(K == 16 || K == 32 || K == 64 || K == 128 || K == 256,
            "int8 supports out_channels 16/32/64/128/256, got %d", K);
  SPX_CHECK(kv <= 32, "int8 supports kernel volumes up to 32, got %d", kv);
  SPX_CHECK(out_dtype == SPX_I8 || out_dtype == SPX_F16 || out_dtype == SPX_BF16 || out_dtype == SPX_F32,
            "bad output dtype %d", out_dtype);
  GemmParams p{};
  p.A = feat;
  p.B = weight;
  p.out = out;
  p.pair = pair;
  p.mask = mask;
  p.argsort = argsort;
  p.bias = bias;
  p.strideK = C;
  p.strideN = static_cast<long long>(kv) * C;
  p.strideD = 1;
  p.n_src = n_in;
  p.n_dst = n_out;
  p.CIN = C;
  p.COUT = K;
  p.kv = kv;
  p.identity_k = identity_k;
  p.b_reverse = 0;
  apply_rows_layout(p, (act & SPX_ROWS_LAYOUT_ACT) ? SPX_ROWS_LAYOUT : ((act & SPX_TILE_ORDER) ? 1 : 0));
  const bool hinted = p.cls && (act & SPX_SPARSE_HINT);   // the host has seen class word 1: a launch-shape hint
  if (hinted) p.app_rows = ((act >> 16) & 0xffff) * 64;   // ... and M (in units of 64 rows, 0 = not told)
  p.act = act & 0xff;
  p.act_alpha = act_alpha;
  p.scale = scale;
  p.add = add;
  p.add_scale = add_scale;
  p.out_dtype = out_dtype;
  if (n_out == 0) return 0;
  const int oes = out_dtype == SPX_I8 ? 1 : (out_dtype == SPX_F32 ? 4 : 2);
  SPX_CHECK(v4_ok(p, 1, oes), "tensor too large for 32-bit buffer offsets");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // tile height by density class (igemm_i8.hip): tables in tile order or the host's sparse hint -> 64-row tiles
  return launch_gather_gemm_int8(p, p.tile_order || hinted, s);
}

size_t spx_igemm_dgrad_ws_bytes(int C, int K, int kv, int dtype) {
  (void)C; (void)K; (void)kv; (void)dtype;
  return 0;  // the weight transpose happens inside the kernel (LDS staging); kv > 32: spx_igemm_acc_bytes
}

int spx_igemm_dgrad(const void *dout, const void *weight, void *din, const int32_t *pair,
                    const uint32_t *mask, const int32_t *argsort, int tile_order, int n_out, int n_in, int C,
                    int K, int kv, int dtype, int subm, void *ws, size_t ws_bytes,
                    spx_stream_t stream) {
  if (n_in == 0) return 0;                                    // empty input: no gradient rows
  SPX_CHECK((dout || n_out == 0) && weight && din, "null tensor pointer");
  SPX_CHECK(pair || kv == 1, "pair table required");
  GemmParams p = dgrad_params(dout, weight, din, pair, mask, argsort, n_out, n_in, C, K, kv, subm);
  p.dense_hint = (tile_order & SPX_DENSE_HINT) ? 1 : 0;
  apply_rows_layout(p, tile_order & ~SPX_DENSE_HINT);
  if (ws && ws_bytes >= spx_igemm_acc_bytes(n_in, C, kv) && kv > 32) p.acc = static_cast<float *>(ws);
  return run_gather_gemm(p, dtype, static_cast<hipStream_t>(stream));
}

// the plan blob holds both forms: the item list of the generic kernels and, behind it, the
// balanced segment plan of the MFMA kernel
static size_t plan1_bytes(int n_in, int kv) {
  return align_up(wgrad_plan_ints(n_in, kv) * sizeof(int32_t), 256);
}

size_t spx_wgrad_plan_bytes(int n_in, int kv) {
  return plan1_bytes(n_in, kv) + align_up(wgrad_plan2_ints(n_in, kv) * sizeof(int32_t), 256);
}

int spx_wgrad_plan(const int32_t *num_per_loc, int n_in, int kv, int subm, int32_t *plan,
                   spx_stream_t stream) {
  SPX_CHECK(num_per_loc && plan, "null pointer");
  SPX_CHECK(kv >= 1 && kv <= 128, "kernel volume %d not supported by wgrad (max 128)", kv);
  // the first-generation item list (front of the buffer) is only read by the fallback kernels of
  // spx_igemm_wgrad, which build it themselves when they run: one launch less per rulebook
  int32_t *plan2 = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(plan) + plan1_bytes(n_in, kv));
  const int G = wgrad_groups(n_in, subm);
  SPX_CHECK(G <= wgrad_groups_max(n_in), "wgrad plan: %d ranges exceed the %d the plan buffer is sized for", G,
            wgrad_groups_max(n_in));
  hipLaunchKernelGGL(wgrad_plan2_kernel, dim3(1), dim3(kW2MaxG), 0, static_cast<hipStream_t>(stream),
                     num_per_loc, n_in, kv, subm, G, plan2);
  SPX_LAUNCH_CHECK();
  return 0;
}

size_t spx_igemm_wgrad_ws_bytes(int n_in, int C, int K, int kv) {
  const int chunk = wgrad_chunk(n_in);
  const size_t nchunks = div_up(n_in > 0 ? n_in : 1, chunk);
  const size_t tiles = static_cast<size_t>(div_up(C, kWT)) * div_up(K, kWT);
  size_t parts = nchunks * kv;                                   // item list (generic kernels)
  const size_t segs = static_cast<size_t>(wgrad_groups_max(n_in)) + kv;   // balanced segments
  if (segs > parts) parts = segs;
  return align_up(parts * tiles * kWT * kWT * sizeof(float), 256) + spx_wgrad_plan_bytes(n_in, kv);
}

static int igemm_wgrad_impl(const void *feat, const void *dout, void *dw, const int32_t *pair_native,
                            const int32_t *num_per_loc, const int32_t *plan, int n_in, int n_out, int C,
                            int K, int kv, int dtype, int subm, void *ws, size_t ws_bytes,
                            spx_stream_t stream, void *stage2_job) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  SPX_CHECK(dw && C > 0 && K > 0, "null tensor pointer");
  SPX_CHECK(kv >= 1 && kv <= 128, "kernel volume %d not supported by wgrad (max 128)", kv);
  if (n_in == 0 || n_out == 0) {                              // no pairs: the gradient is zero
    SPX_HIP(hipMemsetAsync(dw, 0, static_cast<size_t>(K) * kv * C * elem_bytes(dtype), s));
    return 0;
  }
  SPX_CHECK(feat && dout && ws, "null tensor pointer");
  SPX_CHECK(pair_native && num_per_loc, "Native pair lists and counts are required");
  SPX_CHECK(ws_bytes >= spx_igemm_wgrad_ws_bytes(n_in, C, K, kv), "workspace too small");
  WgradParams p{};
  p.feat = feat;
  p.dout = dout;
  p.partial = static_cast<float *>(ws);
  p.native = pair_native;
  p.num = num_per_loc;
  p.n_in = n_in;
  p.n_out = n_out;
  p.C = C;
  p.K = K;
  p.kv = kv;
  p.subm = subm;
  p.chunk = wgrad_chunk(n_in);
  p.nchunks = div_up(n_in > 0 ? n_in : 1, p.chunk);
  p.tiles_c = div_up(C, kWT);
  p.tiles_k = div_up(K, kWT);
  const int ntile = p.tiles_c * p.tiles_k;
  if (!plan) {  // caller did not cache a plan: build it behind the partials
    int32_t *own = reinterpret_cast<int32_t *>(static_cast<char *>(ws) + ws_bytes -
                                               spx_wgrad_plan_bytes(n_in, kv));
    if (spx_wgrad_plan(num_per_loc, n_in, kv, subm, own, stream)) return -2;
    plan = own;
  }
  p.plan = plan;
  const bool mfma = (dtype == SPX_F16 || dtype == SPX_BF16) && C % 8 == 0 && K % 8 == 0;
  constexpr int wgrad_version = 2;
  const bool small_offsets = static_cast<unsigned long long>(n_out) * K * 2ull < 0x7fff0000ull &&
                             static_cast<unsigned long long>(n_in) * C * 2ull < 0x7fff0000ull &&
                             static_cast<unsigned long long>(n_in) * 4ull * (kv + 1) < 0x7fff0000ull;   // both lists of an offset through one resource
  constexpr int f32_mfma = 1;
  const bool f32_path = dtype == SPX_F32 && f32_mfma && C % 4 == 0 && K % 4 == 0 &&
                        static_cast<unsigned long long>(n_out) * K * 4ull < 0x7fff0000ull &&
                        static_cast<unsigned long long>(n_in) * C * 4ull < 0x7fff0000ull &&
                        static_cast<unsigned long long>(n_in) * 4ull < 0x7fff0000ull;
  if (f32_path || (mfma && wgrad_version >= 2 && small_offsets)) {
    Wgrad2Params q{};
    q.feat = feat;
    q.dout = dout;
    q.partial = static_cast<float *>(ws);
    q.native = pair_native;
    q.num = num_per_loc;
    q.plan2 = reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(plan) + plan1_bytes(n_in, kv));
    q.n_in = n_in;
    q.n_out = n_out;
    q.C = C;
    q.K = K;
    q.kv = kv;
    q.subm = subm;
    q.tiles_c = p.tiles_c;
    q.tiles_k = p.tiles_k;
    q.G = wgrad_groups(n_in, subm);
    q.xcd_order = wgrad_xcd_order();
    const dim3 grid(static_cast<unsigned>(q.G) * ntile);
    const size_t lds = 2 * 2 * kW2J * 128;    // two stages x two operand tiles
    const int sl = (C <= 16 && K <= 16) ? 2 : ((C <= 32 && K <= 32) ? 4 : 8);   // live 16-byte slots per row
    if (dtype == SPX_F32)
      hipLaunchKernelGGL(wgrad_f32_kernel, grid, dim3(kThreads), 2 * kW3J * kW3Stride * sizeof(float), s, q);
    else if (dtype == SPX_F16) {
      if (sl == 2) hipLaunchKernelGGL((wgrad_tr_kernel<false, 2>), grid, dim3(kThreads), lds, s, q);
      else if (sl == 4) hipLaunchKernelGGL((wgrad_tr_kernel<false, 4>), grid, dim3(kThreads), lds, s, q);
      else hipLaunchKernelGGL((wgrad_tr_kernel<false, 8>), grid, dim3(kThreads), lds, s, q);
    } else {
      if (sl == 2) hipLaunchKernelGGL((wgrad_tr_kernel<true, 2>), grid, dim3(kThreads), lds, s, q);
      else if (sl == 4) hipLaunchKernelGGL((wgrad_tr_kernel<true, 4>), grid, dim3(kThreads), lds, s, q);
      else hipLaunchKernelGGL((wgrad_tr_kernel<true, 8>), grid, dim3(kThreads), lds, s, q);
    }
    return launch_reduce2(q, dw, dtype, ntile, s, stage2_job);
  }
  {
    // fallback kernels (odd channel counts, tensors beyond 32-bit offsets): their item list is
    // built here, behind the partials (spx_wgrad_plan does not write it)
    int32_t *plan1 = reinterpret_cast<int32_t *>(static_cast<char *>(ws) + ws_bytes -
                                                 spx_wgrad_plan_bytes(n_in, kv));
    hipLaunchKernelGGL(wgrad_plan_kernel, dim3(1), dim3(kThreads), 0, s, num_per_loc, n_in, kv, subm,
                       wgrad_chunk(n_in), plan1);
    p.plan = plan1;
  }
  {
    // upper bound of work items is nchunks * kv * ntile; the kernels loop over the real count
    const long long bound = static_cast<long long>(p.nchunks) * kv * ntile;
    constexpr int max_grid = 1024;
    const dim3 grid(static_cast<unsigned>(bound < max_grid ? bound : max_grid));
    const size_t lds = 2 * kWT * kWJ * 2;
    if (mfma && dtype == SPX_F16)
      hipLaunchKernelGGL(wgrad_mfma_kernel<false>, grid, dim3(kThreads), lds, s, p);
    else if (mfma)
      hipLaunchKernelGGL(wgrad_mfma_kernel<true>, grid, dim3(kThreads), lds, s, p);
    else if (dtype == SPX_F32)
      hipLaunchKernelGGL(wgrad_generic_kernel<float>, grid, dim3(kThreads), 0, s, p);
    else if (dtype == SPX_F16)
      hipLaunchKernelGGL(wgrad_generic_kernel<h16>, grid, dim3(kThreads), 0, s, p);
    else if (dtype == SPX_BF16)
      hipLaunchKernelGGL(wgrad_generic_kernel<b16>, grid, dim3(kThreads), 0, s, p);
    else
      SPX_CHECK(false, "unsupported dtype %d", dtype);
    SPX_LAUNCH_CHECK();
  }
  const dim3 rgrid(div_up(ntile * kWT * kWT, kRedElems), kv);
  if (dtype == SPX_F32)
    hipLaunchKernelGGL(wgrad_reduce_kernel<float>, rgrid, dim3(kRedThreads), 0, s, p,
                       static_cast<float *>(dw));
  else if (dtype == SPX_F16)
    hipLaunchKernelGGL(wgrad_reduce_kernel<h16>, rgrid, dim3(kRedThreads), 0, s, p,
                       static_cast<h16 *>(dw));
  else if (dtype == SPX_BF16)
    hipLaunchKernelGGL(wgrad_reduce_kernel<b16>, rgrid, dim3(kRedThreads), 0, s, p,
                       static_cast<b16 *>(dw));
  else
    SPX_CHECK(false, "unsupported dtype %d", dtype);
  SPX_LAUNCH_CHECK();
  return 0;
}

static int igemm_bwd_impl(const void *feat, const void *dout, const void *weight, void *din, void *dw,
                          const int32_t *pair, const uint32_t *mask, const int32_t *argsort, int tile_order,
                          const int32_t *pair_native, const int32_t *num_per_loc, const int32_t *plan,
                          int n_in, int n_out, int C, int K, int kv, int dtype, int subm, void *ws,
                          size_t ws_bytes, spx_stream_t stream, void *stage2_job) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_in == 0 || n_out == 0) {                              // empty scene: din empty / zero, dW zero
    SPX_CHECK(dw && C > 0 && K > 0 && kv > 0, "null tensor pointer");
    SPX_HIP(hipMemsetAsync(dw, 0, static_cast<size_t>(K) * kv * C * elem_bytes(dtype), s));
    if (n_in > 0) {
      SPX_CHECK(din, "null tensor pointer");
      SPX_HIP(hipMemsetAsync(din, 0, static_cast<size_t>(n_in) * C * elem_bytes(dtype), s));
    }
    return 0;
  }
  SPX_CHECK(feat && dout && weight && din && dw && ws, "null tensor pointer");
  SPX_CHECK(pair_native && num_per_loc, "Native pair lists and counts are required");
  SPX_CHECK(pair || kv == 1, "pair table required");
  SPX_CHECK(ws_bytes >= spx_igemm_wgrad_ws_bytes(n_in, C, K, kv), "workspace too small");
  constexpr int fuse = 1;                  // (dgrad + wgrad in one launch: settled A/B)
  GemmParams p = dgrad_params(dout, weight, din, pair, mask, argsort, n_out, n_in, C, K, kv, subm);
  p.dense_hint = (tile_order & SPX_DENSE_HINT) ? 1 : 0;
  apply_rows_layout(p, tile_order & ~SPX_DENSE_HINT);
  const bool small_offsets = static_cast<unsigned long long>(n_out) * K * 2ull < 0x7fff0000ull &&
                             static_cast<unsigned long long>(n_in) * C * 2ull < 0x7fff0000ull &&
                             static_cast<unsigned long long>(n_in) * 4ull * (kv + 1) < 0x7fff0000ull;   // both lists of an offset through one resource
  const int es = dtype == SPX_F32 ? 4 : 2, lanes = 16 / es;
  const bool offsets_fit = static_cast<unsigned long long>(n_out) * K * es < 0x7fff0000ull &&
                           static_cast<unsigned long long>(n_in) * C * es < 0x7fff0000ull && small_offsets;
  constexpr int f32_mfma = 1;
  const bool fusable = fuse && (dtype == SPX_F16 || dtype == SPX_BF16 || (dtype == SPX_F32 && f32_mfma)) &&
                       C % lanes == 0 && K % lanes == 0 && mfma_ok(dtype, p.CIN, p.COUT, kv, mask) && kv <= 32 &&
                       p.COUT <= 128 && v4_ok(p, es, es) && offsets_fit && n_in > 0 && n_out > 0;
  if (!fusable) {
    if (spx_igemm_dgrad(dout, weight, din, pair, mask, argsort, tile_order, n_out, n_in, C, K, kv, dtype, subm,
                        nullptr, 0, stream))
      return -2;
    return igemm_wgrad_impl(feat, dout, dw, pair_native, num_per_loc, plan, n_in, n_out, C, K, kv, dtype,
                            subm, ws, ws_bytes, stream, stage2_job);
  }
  if (!plan) {  // caller did not cache a plan: build it behind the partials
    int32_t *own = reinterpret_cast<int32_t *>(static_cast<char *>(ws) + ws_bytes -
                                               spx_wgrad_plan_bytes(n_in, kv));
    if (spx_wgrad_plan(num_per_loc, n_in, kv, subm, own, stream)) return -2;
    plan = own;
  }
  Wgrad2Params q{};
  q.feat = feat;
  q.dout = dout;
  q.partial = static_cast<float *>(ws);
  q.native = pair_native;
  q.num = num_per_loc;
  q.plan2 = reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(plan) + plan1_bytes(n_in, kv));
  q.n_in = n_in;
  q.n_out = n_out;
  q.C = C;
  q.K = K;
  q.kv = kv;
  q.subm = subm;
  q.tiles_c = div_up(C, kWT);
  q.tiles_k = div_up(K, kWT);
  q.G = wgrad_groups(n_in, subm);
  q.xcd_order = wgrad_xcd_order();
  const int ntile = q.tiles_c * q.tiles_k;
  const int rc = dtype == SPX_F32 ? dispatch_bwd_f32(p, q, q.G * ntile, s)
                                  : (dtype == SPX_BF16 ? dispatch_bwd_bf16(p, q, q.G * ntile, s)
                                                       : dispatch_bwd<0>(p, q, q.G * ntile, s));
  if (rc) return rc;
  return launch_reduce2(q, dw, dtype, ntile, s, stage2_job);
}

int spx_igemm_wgrad(const void *feat, const void *dout, void *dw, const int32_t *pair_native,
                    const int32_t *num_per_loc, const int32_t *plan, int n_in, int n_out, int C,
                    int K, int kv, int dtype, int subm, void *ws, size_t ws_bytes,
                    spx_stream_t stream) {
  return igemm_wgrad_impl(feat, dout, dw, pair_native, num_per_loc, plan, n_in, n_out, C, K, kv, dtype, subm, ws,
                          ws_bytes, stream, nullptr);
}

int spx_igemm_bwd(const void *feat, const void *dout, const void *weight, void *din, void *dw,
                  const int32_t *pair, const uint32_t *mask, const int32_t *argsort, int tile_order,
                  const int32_t *pair_native, const int32_t *num_per_loc, const int32_t *plan,
                  int n_in, int n_out, int C, int K, int kv, int dtype, int subm, void *ws,
                  size_t ws_bytes, spx_stream_t stream) {
  return igemm_bwd_impl(feat, dout, weight, din, dw, pair, mask, argsort, tile_order, pair_native, num_per_loc, plan, n_in,
                        n_out, C, K, kv, dtype, subm, ws, ws_bytes, stream, nullptr);
}

int spx_igemm_wgrad_deferred(const void *feat, const void *dout, void *dw, const int32_t *pair_native,
                             const int32_t *num_per_loc, const int32_t *plan, int n_in, int n_out, int C,
                             int K, int kv, int dtype, int subm, void *ws, size_t ws_bytes,
                             spx_stream_t stream, void *stage2_job) {
  SPX_CHECK(stage2_job, "stage2_job is required");
  memset(stage2_job, 0, SPX_STAGE2_JOB_BYTES);
  return igemm_wgrad_impl(feat, dout, dw, pair_native, num_per_loc, plan, n_in, n_out, C, K, kv, dtype, subm, ws,
                          ws_bytes, stream, stage2_job);
}

int spx_igemm_bwd_deferred(const void *feat, const void *dout, const void *weight, void *din, void *dw,
                           const int32_t *pair, const uint32_t *mask, const int32_t *argsort, int tile_order,
                           const int32_t *pair_native, const int32_t *num_per_loc, const int32_t *plan,
                           int n_in, int n_out, int C, int K, int kv, int dtype, int subm, void *ws,
                           size_t ws_bytes, spx_stream_t stream, void *stage2_job) {
  SPX_CHECK(stage2_job, "stage2_job is required");
  memset(stage2_job, 0, SPX_STAGE2_JOB_BYTES);
  return igemm_bwd_impl(feat, dout, weight, din, dw, pair, mask, argsort, tile_order, pair_native, num_per_loc, plan, n_in,
                        n_out, C, K, kv, dtype, subm, ws, ws_bytes, stream, stage2_job);
}

int spx_stage2_job_retarget(void *stage2_job, void *dw) {
  SPX_CHECK(stage2_job && dw, "null pointer");
  Stage2JobH j;
  memcpy(&j, stage2_job, sizeof(j));
  if (j.valid) {
    j.dw = dw;
    memcpy(stage2_job, &j, sizeof(j));
  }
  return j.valid;
}

int spx_wgrad_stage2_batch(const void *jobs, int njobs, spx_stream_t stream) {
  SPX_CHECK(njobs >= 0 && (jobs || njobs == 0), "jobs required");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const char *base = static_cast<const char *>(jobs);
  for (int dtype : {SPX_F16, SPX_BF16, SPX_F32}) {
    Wgrad2Batch b{};
    int kvmax = 0;
    auto flush = [&]() -> int {
      if (b.n == 0) return 0;
      // ~2048 workgroups over all jobs -- one resident round: with a layer's own 512 per tile column a batch of ten
      // layers ran six rounds (34 us measured); a workgroup walks its few items one behind the other instead
      int gx = 2048 / (b.ybase[b.n] > 0 ? b.ybase[b.n] : 1);
      gx = gx < 32 ? 32 : gx;
      gx = gx > reduce2_blocks(kvmax) ? reduce2_blocks(kvmax) : gx;
      const dim3 grid(gx, b.ybase[b.n]);
      count_launch(kFamStage2Batch);
      if (dtype == SPX_F32) hipLaunchKernelGGL(wgrad_reduce2_batch_kernel<float>, grid, dim3(kRedThreads), 0, s, b);
      else if (dtype == SPX_F16) hipLaunchKernelGGL(wgrad_reduce2_batch_kernel<h16>, grid, dim3(kRedThreads), 0, s, b);
      else hipLaunchKernelGGL(wgrad_reduce2_batch_kernel<b16>, grid, dim3(kRedThreads), 0, s, b);
      SPX_LAUNCH_CHECK();
      b = Wgrad2Batch{};
      kvmax = 0;
      return 0;
    };
    for (int i = 0; i < njobs; ++i) {
      Stage2JobH j;
      memcpy(&j, base + static_cast<size_t>(i) * SPX_STAGE2_JOB_BYTES, sizeof(j));
      if (!j.valid || j.dtype != dtype) continue;
      SPX_CHECK(j.partial && j.plan2 && j.dw, "job %d: null pointer (was it filled by spx_igemm_*_deferred?)", i);
      b.job[b.n] = Wgrad2Job{j.partial, j.plan2, j.dw, j.G, j.kv, j.tiles_k, j.tiles_c, j.K, j.C};
      b.ybase[b.n + 1] = b.ybase[b.n] + j.tiles_k * j.tiles_c;
      kvmax = j.kv > kvmax ? j.kv : kvmax;
      if (++b.n == kStage2MaxJobs && flush()) return -1;
    }
    if (flush()) return -1;
  }
  return 0;
}

int spx_bias_act_inplace(void *out, const void *bias, int n, int K, int dtype, int act,
                         float act_alpha, spx_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long long total = static_cast<long long>(n) * K;
  if (total == 0) return 0;
  const dim3 grid(static_cast<unsigned>((total + kThreads - 1) / kThreads));
  if (dtype == SPX_F32)
    hipLaunchKernelGGL(bias_act_kernel<float>, grid, dim3(kThreads), 0, s, static_cast<float *>(out),
                       static_cast<const float *>(bias), total, K, act, act_alpha);
  else if (dtype == SPX_F16)
    hipLaunchKernelGGL(bias_act_kernel<h16>, grid, dim3(kThreads), 0, s, static_cast<h16 *>(out),
                       static_cast<const h16 *>(bias), total, K, act, act_alpha);
  else if (dtype == SPX_BF16)
    hipLaunchKernelGGL(bias_act_kernel<b16>, grid, dim3(kThreads), 0, s, static_cast<b16 *>(out),
                       static_cast<const b16 *>(bias), total, K, act, act_alpha);
  else
    SPX_CHECK(false, "unsupported dtype %d", dtype);
  SPX_LAUNCH_CHECK();
  return 0;
}

int spx_pad_rows(const void *src, void *dst, long long rows, int src_row_bytes, int dst_row_bytes,
                 spx_stream_t stream) {
  SPX_CHECK(src_row_bytes > 0 && dst_row_bytes >= src_row_bytes && src_row_bytes % 2 == 0 && dst_row_bytes % 2 == 0,
            "row sizes must be even, destination rows at least as long as source rows");
  if (rows <= 0) return 0;
  SPX_CHECK(src && dst, "null tensor pointer");
  const long long total = rows * (dst_row_bytes / 2);
  hipLaunchKernelGGL(pad_rows_kernel, dim3(static_cast<unsigned>((total + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), static_cast<const uint16_t *>(src), static_cast<uint16_t *>(dst),
                     total, src_row_bytes / 2, dst_row_bytes / 2);
  SPX_LAUNCH_CHECK();
  return 0;
}

#ifdef SPX_TIMELINE
// debug builds only: copies the timeline table (8192 workgroups x 8 stamps, uint64) to host memory
int spx_debug_timeline(unsigned long long *dst_h) {
  SPX_HIP(hipDeviceSynchronize());
  SPX_HIP(hipMemcpyFromSymbol(dst_h, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * kTlMaxWg * kTlSlots));
  return 0;
}
#endif

}  // extern "C"
