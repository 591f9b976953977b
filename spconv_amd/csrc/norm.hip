// Batch normalisation (+ fused ReLU) over the [N, C] feature matrix of a sparse tensor.
//
// A voxel backbone interleaves every sparse convolution with BatchNorm1d + ReLU on the features
// (the reference leaves them to torch: SparseSequential applies dense modules to `.features`,
// modules.py:131-145).  On MI355X torch's channels-last kernels need 55-60 us for a 400 k x 16 fp16
// matrix (13 MB: 0.23 TB/s) and were 35 % of the GPU time of BASELINE config 4; these are plain
// HBM-bound streaming kernels instead (16 bytes per lane, rows interleaved over the lanes of a wave so
// that every wave instruction covers whole 128-byte lines):
//   forward (training)  bn_partial -> bn_finalize -> bn_apply
//   forward (inference) bn_apply with the running statistics
//   backward            bn_bwd_partial -> bn_bwd_finalize -> bn_bwd_apply
// Statistics are accumulated in fp32 per block around a per-block shift (the block's first row) and
// merged with Chan's parallel update, so cancellation is bounded by the spread inside ~400 rows.
// Semantics are torch.nn.BatchNorm1d's: biased variance for normalisation, unbiased for the running
// estimate, `momentum`, `eps`, affine weight / bias (may be NULL).
#include "common.h"

namespace spx {
namespace {

constexpr int kT = 256;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int DT> struct Vec;       // 16 bytes of input -> VPL floats
template <> struct Vec<SPX_F32> {
  static constexpr int VPL = 4;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ void unpack(const u32x4 &v, float (&f)[4]) {
    const f32x4 t = __builtin_bit_cast(f32x4, v);
    f[0] = t.x;
    f[1] = t.y;
    f[2] = t.z;
    f[3] = t.w;
  }
  static __device__ __forceinline__ u32x4 pack(const float (&f)[4]) {
    return __builtin_bit_cast(u32x4, f32x4{f[0], f[1], f[2], f[3]});
  }
};
template <> struct Vec<SPX_F16> {
  static constexpr int VPL = 8;
  static __device__ __forceinline__ void unpack(const u32x4 &v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(v[i] & 0xffffu)));
      f[2 * i + 1] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(v[i] >> 16)));
    }
  }
  static __device__ __forceinline__ u32x4 pack(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint16_t lo = __builtin_bit_cast(uint16_t, static_cast<_Float16>(f[2 * i]));
      const uint16_t hi = __builtin_bit_cast(uint16_t, static_cast<_Float16>(f[2 * i + 1]));
      v[i] = static_cast<unsigned>(lo) | (static_cast<unsigned>(hi) << 16);
    }
    return v;
  }
};
template <> struct Vec<SPX_BF16> {
  static constexpr int VPL = 8;
  static __device__ __forceinline__ void unpack(const u32x4 &v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __builtin_bit_cast(float, v[i] << 16);
      f[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ unsigned rne(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
  }
  static __device__ __forceinline__ u32x4 pack(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = rne(f[2 * i]) | (rne(f[2 * i + 1]) << 16);
    return v;
  }
};

// Affine parameters and running estimates are [C] vectors in fp32 or in a 16-bit dtype (a model
// converted with .half() / .bfloat16()): read / written through a runtime dtype code.  They are
// touched once per block (staged into LDS) or once per channel, never per element.
__device__ __forceinline__ float ldp(const void *p, int dt, int i) {
  if (dt == SPX_F32) return static_cast<const float *>(p)[i];
  const uint16_t h = static_cast<const uint16_t *>(p)[i];
  if (dt == SPX_F16) return static_cast<float>(__builtin_bit_cast(_Float16, h));
  return __builtin_bit_cast(float, static_cast<unsigned>(h) << 16);
}
__device__ __forceinline__ void stp(void *p, int dt, int i, float v) {
  if (dt == SPX_F32) static_cast<float *>(p)[i] = v;
  else if (dt == SPX_F16) static_cast<uint16_t *>(p)[i] = __builtin_bit_cast(uint16_t, static_cast<_Float16>(v));
  else static_cast<uint16_t *>(p)[i] = static_cast<uint16_t>(Vec<SPX_BF16>::rne(v));
}

// Static-shape tensors (spconv_amd/pytorch/static.py): only the first *n_live rows are rows of the scene,
// the rest is padding ("dead rows").  Statistics run over the live rows; dead rows come out as zeros in
// both directions, so nothing downstream (a SubM layer's centre pair, a weight gradient) sees them.
__device__ __forceinline__ int live_rows(const int32_t *n_live, int n) {
  if (!n_live) return n;
  const int v = *n_live;
  return v < 0 ? 0 : (v < n ? v : n);
}

// A block owns rows [r0, r1); thread t reads the 16-byte piece (t % P) of rows r0 + t / P + i * (kT / P),
// P = pieces per row = C / VPL (a power of two <= 64 is not required: kT / P rows per sweep, threads past
// (kT / P) * P idle).
struct RowSplit {
  int r0, r1, piece, lane_row, rows_per_sweep;
  bool active;
};
__device__ __forceinline__ RowSplit row_split(int n, int P, int nblocks) {
  RowSplit s;
  const long long per = (static_cast<long long>(n) + nblocks - 1) / nblocks;
  s.r0 = static_cast<int>(min(static_cast<long long>(n), per * blockIdx.x));
  s.r1 = static_cast<int>(min(static_cast<long long>(n), per * (blockIdx.x + 1)));
  s.rows_per_sweep = kT / P;
  s.piece = threadIdx.x % P;
  s.lane_row = threadIdx.x / P;
  s.active = s.lane_row < s.rows_per_sweep;
  return s;
}

// Sums a[], b[] over the threads of the block that hold the same piece (threads piece + P * lane_row);
// on return threads 0 .. C-1 hold the totals of channel threadIdx.x in (ra, rb).  P dividing 64: the lanes of a 16-lane
// row that hold the same piece are summed with DPP row rotations (plain VALU: no trip through the LDS crossbar per
// step, which the ds_bpermute form of a wave butterfly pays 2 * VPL times per step), the 16 row totals of the block
// through LDS; other P: a short serial loop.
template <int D>
__device__ __forceinline__ float row_ror_add(float x) {
  return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + D, 0xf, 0xf, false));
}

template <int VPL>
__device__ __forceinline__ void piece_reduce(float (&a)[VPL], float (&b)[VPL], int P, int C, int rows_per_sweep,
                                             float (*lds)[kT][VPL + 1], float &ra, float &rb) {
  const int lane = threadIdx.x & 63;
  ra = rb = 0.f;
  if (64 % P == 0) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      if (P <= 8) { a[i] = row_ror_add<8>(a[i]); b[i] = row_ror_add<8>(b[i]); }
      if (P <= 4) { a[i] = row_ror_add<4>(a[i]); b[i] = row_ror_add<4>(b[i]); }
      if (P <= 2) { a[i] = row_ror_add<2>(a[i]); b[i] = row_ror_add<2>(b[i]); }
      if (P <= 1) { a[i] = row_ror_add<1>(a[i]); b[i] = row_ror_add<1>(b[i]); }
    }
    // lane j of a row now holds the row's total of piece (16 * row + j) % P, for j < min(P, 16)
    const int j = lane & 15, slot = threadIdx.x >> 4, per_row = P < 16 ? P : 16;
    if (j < per_row) {
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        lds[0][slot * 16 + j][i] = a[i];
        lds[1][slot * 16 + j][i] = b[i];
      }
    }
    __syncthreads();
    if (threadIdx.x < C) {
      const int piece = threadIdx.x / VPL, e = threadIdx.x % VPL;
      // the rows (of 16 lanes) that hold this piece: every one for P <= 16, every (P / 16)-th from piece / 16 on otherwise
      const int step = P <= 16 ? 1 : P / 16, first = P <= 16 ? 0 : piece / 16, jj = piece % 16;
      for (int r = first; r < kT / 16; r += step) {
        ra += lds[0][r * 16 + jj][e];
        rb += lds[1][r * 16 + jj][e];
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    lds[0][threadIdx.x][i] = a[i];
    lds[1][threadIdx.x][i] = b[i];
  }
  __syncthreads();
  if (threadIdx.x < C) {
    const int piece = threadIdx.x / VPL, e = threadIdx.x % VPL;
    for (int lr = 0; lr < rows_per_sweep; ++lr) {
      ra += lds[0][lr * P + piece][e];
      rb += lds[1][lr * P + piece][e];
    }
  }
}

// partial[0][c][b] = rows of block b, [1][c][b] = mean, [2][c][b] = M2 (sum of squared deviations): field-major, then
// channel, then block -- the merge reads each field of a channel as one contiguous run (igemm_defs.h bn_record_store)
template <int DT>
__global__ void __launch_bounds__(kT)
bn_partial_kernel(const u32x4 *__restrict__ x, int n, int C, float *__restrict__ partial,
                  const int32_t *__restrict__ n_live) {
  constexpr int VPL = Vec<DT>::VPL;
  __shared__ float lds[2][kT][VPL + 1];
  const int P = C / VPL;
  const RowSplit s = row_split(live_rows(n_live, n), P, gridDim.x);
  float shift[VPL], sum[VPL], sq[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) shift[i] = sum[i] = sq[i] = 0.f;
  if (s.active && s.r0 < s.r1) Vec<DT>::unpack(x[static_cast<size_t>(s.r0) * P + s.piece], shift);
  if (s.active) {
    // eight rows of loads in flight per thread (a block owns 64-390 rows: one trip; without this a thread's loads were
    // a chain of memory round trips); rows past the block's end are predicated, not branched around
    constexpr int U = 8;
    for (int r = s.r0 + s.lane_row; r < s.r1; r += U * s.rows_per_sweep) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ru = r + u * s.rows_per_sweep;
        v[u] = ru < s.r1 ? x[static_cast<size_t>(ru) * P + s.piece] : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (r + u * s.rows_per_sweep < s.r1) {
          float f[VPL];
          Vec<DT>::unpack(v[u], f);
#pragma unroll
          for (int i = 0; i < VPL; ++i) {
            const float d = f[i] - shift[i];
            sum[i] += d;
            sq[i] += d * d;
          }
        }
      }
    }
  }
  if (!s.active) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) sum[i] = sq[i] = 0.f;
  }
  float a, b;
  piece_reduce<VPL>(sum, sq, P, C, s.rows_per_sweep, lds, a, b);
  // threads 0 .. C-1: one channel each
  if (threadIdx.x < C) {
    const int piece = threadIdx.x / VPL, e = threadIdx.x % VPL;
    const float cnt = static_cast<float>(s.r1 - s.r0);
    // the shift of channel c is the block's first row, read again by this thread
    float sh = 0.f;
    if (s.r0 < s.r1) {
      float f[VPL];
      Vec<DT>::unpack(x[static_cast<size_t>(s.r0) * P + piece], f);
      sh = f[e];
    }
    const size_t G = gridDim.x, c = threadIdx.x;
    partial[c * G + blockIdx.x] = cnt;
    partial[(C + c) * G + blockIdx.x] = cnt > 0.f ? sh + a / cnt : 0.f;
    partial[(2 * static_cast<size_t>(C) + c) * G + blockIdx.x] = cnt > 0.f ? b - a * a / cnt : 0.f;
  }
}

// one block per channel: Chan merge of the G partials; mean / invstd out, running estimates updated
template <int T>
__global__ void __launch_bounds__(T)
bn_finalize_kernel(const float *__restrict__ partial, int G, int C, float eps, float momentum,
                   float *__restrict__ mean_out, float *__restrict__ invstd_out,
                   void *__restrict__ running_mean, void *__restrict__ running_var, int pdt,
                   long long *__restrict__ num_batches_tracked) {
  constexpr int kT = T;
  __shared__ float ln[kT], lm[kT], l2[kT];
  const int c = blockIdx.x;
  if (num_batches_tracked && c == 0 && threadIdx.x == 0) *num_batches_tracked += 1;   // (no launch of its own)
  float n = 0.f, m = 0.f, M2 = 0.f;
  // eight records per trip: their 24 loads leave together, the merges follow (the records a convolution's epilogue
  // leaves come one per workgroup of that launch -- ~3900 at 400 k rows: sixteen dependent trips per thread otherwise)
  constexpr int U = 8;
  for (int b0 = threadIdx.x; b0 < G; b0 += kT * U) {
    float nb[U], mb[U], Mb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = b0 + u * kT;
      const bool ok = b < G;
      const size_t at = static_cast<size_t>(c) * G + (ok ? b : 0), fs = static_cast<size_t>(C) * G;      // (field stride)
      nb[u] = ok ? partial[at] : 0.f;
      mb[u] = ok ? partial[at + fs] : 0.f;
      Mb[u] = ok ? partial[at + 2 * fs] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (nb[u] > 0.f) {
        const float tot = n + nb[u], d = mb[u] - m;
        m += d * (nb[u] / tot);
        M2 += Mb[u] + d * d * (n * nb[u] / tot);
        n = tot;
      }
    }
  }
  ln[threadIdx.x] = n;
  lm[threadIdx.x] = m;
  l2[threadIdx.x] = M2;
  __syncthreads();
  for (int stride = kT / 2; stride > 0; stride >>= 1) {
    if (threadIdx.x < stride) {
      const float na = ln[threadIdx.x], nb = ln[threadIdx.x + stride];
      if (nb > 0.f) {
        const float tot = na + nb, d = lm[threadIdx.x + stride] - lm[threadIdx.x];
        lm[threadIdx.x] += d * (nb / tot);
        l2[threadIdx.x] += l2[threadIdx.x + stride] + d * d * (na * nb / tot);
        ln[threadIdx.x] = tot;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float cnt = ln[0], mean = lm[0];
    const float var = cnt > 0.f ? l2[0] / cnt : 0.f;
    mean_out[c] = mean;
    invstd_out[c] = rsqrtf(var + eps);
    if (running_mean) stp(running_mean, pdt, c, (1.f - momentum) * ldp(running_mean, pdt, c) + momentum * mean);
    if (running_var) {
      const float unbiased = cnt > 1.f ? l2[0] / (cnt - 1.f) : var;
      stp(running_var, pdt, c, (1.f - momentum) * ldp(running_var, pdt, c) + momentum * unbiased);
    }
  }
}

// y = x * scale[c] + shift[c] (scale = invstd * w, shift = b - mean * scale), optional ReLU.  The two
// coefficient vectors are computed once per block into LDS.  stat_is_var: `stat2` holds a variance
// (inference with the running estimate, dtype pdt like `stat1`) instead of the saved fp32 1 / std.
template <int DT>
__global__ void __launch_bounds__(kT)
bn_apply_kernel(const u32x4 *__restrict__ x, u32x4 *__restrict__ y, long long pieces, int C,
                const void *__restrict__ stat1, const void *__restrict__ stat2,
                const void *__restrict__ weight, const void *__restrict__ bias, int pdt, float eps,
                int stat_is_var, int relu, const int32_t *__restrict__ n_live, int n) {
  constexpr int VPL = Vec<DT>::VPL;
  __shared__ __attribute__((aligned(16))) float l_sc[kT], l_sh[kT];
  // the first piece of this thread is asked for BEFORE the coefficients are put together: its latency runs under the
  // prologue's (parameter loads -> LDS -> barrier), not behind it
  long long i = static_cast<long long>(blockIdx.x) * kT + threadIdx.x;
  u32x4 v = i < pieces ? x[i] : u32x4{0u, 0u, 0u, 0u};
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    const float mean = stat_is_var ? ldp(stat1, pdt, c) : static_cast<const float *>(stat1)[c];
    const float is = stat_is_var ? rsqrtf(ldp(stat2, pdt, c) + eps) : static_cast<const float *>(stat2)[c];
    const float sc = is * (weight ? ldp(weight, pdt, c) : 1.f);
    l_sc[c] = sc;
    l_sh[c] = (bias ? ldp(bias, pdt, c) : 0.f) - mean * sc;
  }
  __syncthreads();
  const int P = C / VPL;
  const long long live = static_cast<long long>(live_rows(n_live, n)) * P;
  const long long stride = static_cast<long long>(gridDim.x) * kT;
  while (i < pieces) {
    const long long nxt = i + stride;
    const u32x4 vn = nxt < pieces ? x[nxt] : u32x4{0u, 0u, 0u, 0u};
    const int c0 = static_cast<int>(i % P) * VPL;
    float f[VPL];
    Vec<DT>::unpack(v, f);
#pragma unroll
    for (int e = 0; e < VPL; ++e) {
      float r = f[e] * l_sc[c0 + e] + l_sh[c0 + e];
      if (relu) r = r > 0.f ? r : 0.f;
      f[e] = i < live ? r : 0.f;
    }
    __builtin_nontemporal_store(Vec<DT>::pack(f), &y[i]);      // (not read again by this launch)
    v = vn;
    i = nxt;
  }
}

// partial[b][0][c] = sum dy, [1][c] = sum dy * xhat over the block's rows (dy masked by y > 0 with ReLU).  The first
// kBwdBatch rows of a thread (all of them up to 8 rows per thread: 1 M rows at 16 channels) are asked for before anything
// else; the per-channel coefficients come through LDS (threads < C load them once per block -- 4 * VPL loads per
// thread were as many requests as the rows themselves on a small level).
constexpr int kBwdBatch = 8;

template <int DT>
__global__ void __launch_bounds__(kT)
bn_bwd_partial_kernel(const u32x4 *__restrict__ x, const u32x4 *__restrict__ dy, int n, int C,
                      const float *__restrict__ mean, const float *__restrict__ invstd,
                      const void *__restrict__ weight, const void *__restrict__ bias, int pdt, int relu,
                      float *__restrict__ partial, const int32_t *__restrict__ n_live) {
  constexpr int VPL = Vec<DT>::VPL;
  __shared__ float lds[2][kT][VPL + 1];
  __shared__ __attribute__((aligned(16))) float l_a[kT], l_b[kT], l_w[kT], l_bias[kT];
  const int P = C / VPL;
  const RowSplit s = row_split(live_rows(n_live, n), P, gridDim.x);
  constexpr int U = kBwdBatch;
  u32x4 vx[U], vg[U];
  int r = s.r0 + s.lane_row;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int ru = r + u * s.rows_per_sweep;
    const bool ok = s.active && ru < s.r1;
    vx[u] = ok ? x[static_cast<size_t>(ru) * P + s.piece] : u32x4{0u, 0u, 0u, 0u};
    vg[u] = ok ? dy[static_cast<size_t>(ru) * P + s.piece] : u32x4{0u, 0u, 0u, 0u};
  }
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    const float is = invstd[c];
    l_a[c] = is;
    l_b[c] = -mean[c] * is;
    l_w[c] = weight ? ldp(weight, pdt, c) : 1.f;
    l_bias[c] = bias ? ldp(bias, pdt, c) : 0.f;
  }
  __syncthreads();
  float s1[VPL], s2[VPL], ca[VPL], cb[VPL], w[VPL], bb[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    s1[i] = s2[i] = 0.f;
    const int c = s.active ? s.piece * VPL + i : 0;
    ca[i] = l_a[c];
    cb[i] = l_b[c];
    w[i] = l_w[c];
    bb[i] = l_bias[c];
  }
  if (s.active) {
    while (r < s.r1) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (r + u * s.rows_per_sweep < s.r1) {
          float f[VPL], g[VPL];
          Vec<DT>::unpack(vx[u], f);
          Vec<DT>::unpack(vg[u], g);
#pragma unroll
          for (int i = 0; i < VPL; ++i) {
            const float xh = f[i] * ca[i] + cb[i];
            const float gg = (relu && xh * w[i] + bb[i] <= 0.f) ? 0.f : g[i];
            s1[i] += gg;
            s2[i] += gg * xh;
          }
        }
      }
      r += U * s.rows_per_sweep;
      if (r < s.r1) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ru = r + u * s.rows_per_sweep;
          const bool ok = ru < s.r1;
          vx[u] = ok ? x[static_cast<size_t>(ru) * P + s.piece] : u32x4{0u, 0u, 0u, 0u};
          vg[u] = ok ? dy[static_cast<size_t>(ru) * P + s.piece] : u32x4{0u, 0u, 0u, 0u};
        }
      }
    }
  }
  float a, b;
  piece_reduce<VPL>(s1, s2, P, C, s.rows_per_sweep, lds, a, b);
  if (threadIdx.x < C) {
    partial[static_cast<size_t>(blockIdx.x) * 2 * C + threadIdx.x] = a;
    partial[static_cast<size_t>(blockIdx.x) * 2 * C + C + threadIdx.x] = b;
  }
}

// sums[0][c] = sum dy (= dbias), sums[1][c] = sum dy * xhat (= dweight)
__global__ void __launch_bounds__(kT)
bn_bwd_finalize_kernel(const float *__restrict__ partial, int G, int C, float *__restrict__ sums,
                       void *__restrict__ dweight, void *__restrict__ dbias, int pdt) {
  __shared__ float la[kT], lb[kT];
  const int c = blockIdx.x;
  float a = 0.f, b = 0.f;
  for (int g = threadIdx.x; g < G; g += kT) {
    a += partial[static_cast<size_t>(g) * 2 * C + c];
    b += partial[static_cast<size_t>(g) * 2 * C + C + c];
  }
  la[threadIdx.x] = a;
  lb[threadIdx.x] = b;
  __syncthreads();
  for (int stride = kT / 2; stride > 0; stride >>= 1) {
    if (threadIdx.x < stride) {
      la[threadIdx.x] += la[threadIdx.x + stride];
      lb[threadIdx.x] += lb[threadIdx.x + stride];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[c] = la[0];                       // sum dy       (= dbias)
    sums[C + c] = lb[0];                   // sum dy * xhat (= dweight)
    if (dbias) stp(dbias, pdt, c, la[0]);
    if (dweight) stp(dweight, pdt, c, lb[0]);
  }
}

// training: dx = w * invstd * (dy - sum_dy / n - xhat * sum_dy_xhat / n);  inference statistics
// (use_batch_stats == 0): dx = w * invstd * dy.  Per-channel coefficients staged in LDS.
template <int DT>
__global__ void __launch_bounds__(kT)
bn_bwd_apply_kernel(const u32x4 *__restrict__ x, const u32x4 *__restrict__ dy, u32x4 *__restrict__ dx,
                    long long pieces, int n, int C, const float *__restrict__ mean,
                    const float *__restrict__ invstd, const void *__restrict__ weight,
                    const void *__restrict__ bias, int pdt, const float *__restrict__ sums, int relu,
                    int use_batch_stats, const int32_t *__restrict__ n_live) {
  constexpr int VPL = Vec<DT>::VPL;
  // xhat = x * l_a + l_b;  relu mask: xhat * l_w + l_bias <= 0;  dx = l_g * (dy' - l_c1 - xhat * l_c2)
  __shared__ __attribute__((aligned(16))) float l_a[kT], l_b[kT], l_w[kT], l_bias[kT], l_g[kT], l_c1[kT], l_c2[kT];
  const int n_eff = live_rows(n_live, n);
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    const float inv_n = (use_batch_stats && n_eff > 0) ? 1.f / static_cast<float>(n_eff) : 0.f;
    const float w = weight ? ldp(weight, pdt, c) : 1.f;
    l_a[c] = invstd[c];
    l_b[c] = -mean[c] * invstd[c];
    l_w[c] = w;
    l_bias[c] = bias ? ldp(bias, pdt, c) : 0.f;
    l_g[c] = w * invstd[c];
    l_c1[c] = sums[c] * inv_n;
    l_c2[c] = sums[C + c] * inv_n;
  }
  __syncthreads();
  const int P = C / VPL;
  const long long live = static_cast<long long>(n_eff) * P;
  for (long long i = static_cast<long long>(blockIdx.x) * kT + threadIdx.x; i < pieces;
       i += static_cast<long long>(gridDim.x) * kT) {
    const int c0 = static_cast<int>(i % P) * VPL;
    float f[VPL], g[VPL];
    Vec<DT>::unpack(x[i], f);
    Vec<DT>::unpack(dy[i], g);
#pragma unroll
    for (int e = 0; e < VPL; ++e) {
      const int c = c0 + e;
      const float xh = f[e] * l_a[c] + l_b[c];
      float gg = g[e];
      if (relu && xh * l_w[c] + l_bias[c] <= 0.f) gg = 0.f;
      f[e] = i < live ? l_g[c] * (gg - l_c1[c] - xh * l_c2[c]) : 0.f;
    }
    __builtin_nontemporal_store(Vec<DT>::pack(f), &dx[i]);
  }
}

int bn_blocks(int n) {
  // at most 1024 blocks (~390 rows each at 400 k rows), at least 64 rows per block: a 20 k-row level still puts a block
  // on every CU (53 blocks of 384 rows left four CUs in five idle and took 12 us for 5 MB)
  int g = div_up(n > 0 ? n : 1, 64);
  return g < 1 ? 1 : (g > 1024 ? 1024 : g);
}

bool bn_shape_ok(int C, int dtype) {
  const int vpl = dtype == SPX_F32 ? 4 : 8;
  return (dtype == SPX_F32 || dtype == SPX_F16 || dtype == SPX_BF16) && C > 0 && C % vpl == 0 && C <= kT &&
         C / vpl <= kT;
}

unsigned stream_grid(long long pieces) {
  long long b = (pieces + kT - 1) / kT;
  return static_cast<unsigned>(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace
}  // namespace spx

using namespace spx;

// measurement builds only (-DSPX_BN_PROBE, tools/bn_probe.py): which launches of a call are issued -- bit 0 = statistics
// pass, 1 = merge, 2 = apply -- so that each can be timed alone; the library always issues all three
#ifdef SPX_BN_PROBE
static int bn_phases() {
  static const int v = [] { const char *e = getenv("SPX_BN_PHASES"); return e ? atoi(e) : 7; }();
  return v;
}
#else
static constexpr int bn_phases() { return 7; }
#endif

#define SPX_BN_DISPATCH(dtype, CALL)                         \
  do {                                                       \
    if ((dtype) == SPX_F16) { CALL(SPX_F16); }               \
    else if ((dtype) == SPX_BF16) { CALL(SPX_BF16); }        \
    else { CALL(SPX_F32); }                                  \
  } while (0)

extern "C" {

size_t spx_batchnorm_ws_bytes(int n, int C) {
  // per-block partials (3 floats per channel) + the [2][C] sums of the backward pass
  return align_up((static_cast<size_t>(bn_blocks(n)) * 3 + 2) * (C > 0 ? C : 1) * sizeof(float), 256) + 256;
}

static int batchnorm_fwd_impl(const void *x, void *y, int n, int C, int dtype, const void *weight,
                              const void *bias, void *running_mean, void *running_var,
                              long long *num_batches_tracked, int param_dtype, int training, float momentum,
                              float eps, int relu, float *save_mean, float *save_invstd, void *ws,
                              size_t ws_bytes, const int32_t *n_live, spx_stream_t stream,
                              const float *ext_partial, int ext_G) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  SPX_CHECK(bn_shape_ok(C, dtype), "batchnorm: C = %d must be a multiple of %d (<= 256), dtype f16/bf16/f32", C,
            dtype == SPX_F32 ? 4 : 8);
  SPX_CHECK(param_dtype == SPX_F32 || param_dtype == SPX_F16 || param_dtype == SPX_BF16, "bad parameter dtype");
  if (n == 0) return 0;
  SPX_CHECK(x && y, "null tensor pointer");
  const int vpl = dtype == SPX_F32 ? 4 : 8;
  const long long pieces = static_cast<long long>(n) * (C / vpl);
  const u32x4 *xv = static_cast<const u32x4 *>(x);
  u32x4 *yv = static_cast<u32x4 *>(y);
  if (training) {
    SPX_CHECK(save_mean && save_invstd && (ext_partial || (ws && ws_bytes >= spx_batchnorm_ws_bytes(n, C))),
              "training needs save_mean / save_invstd and the workspace");
    // statistics records {rows, mean, M2} per channel: from the producing convolution's epilogue (spx_igemm_fwd_stats:
    // one per workgroup of that launch), or from a pass over the rows here
    const int G = ext_partial ? ext_G : bn_blocks(n);
    const float *partial = ext_partial ? ext_partial : static_cast<const float *>(ws);
    if (!ext_partial && (bn_phases() & 1)) {
#define SPX_BN_PARTIAL(D) \
  hipLaunchKernelGGL(bn_partial_kernel<D>, dim3(G), dim3(kT), 0, s, xv, n, C, static_cast<float *>(ws), n_live)
      SPX_BN_DISPATCH(dtype, SPX_BN_PARTIAL);
#undef SPX_BN_PARTIAL
    }
    if (bn_phases() & 2)
    hipLaunchKernelGGL(bn_finalize_kernel<kT>, dim3(C), dim3(kT), 0, s, partial, G, C, eps, momentum, save_mean,
                       save_invstd, running_mean, running_var, param_dtype, num_batches_tracked);
    if (!(bn_phases() & 4)) return 0;
#define SPX_BN_APPLY(D)                                                                                    \
  hipLaunchKernelGGL(bn_apply_kernel<D>, dim3(stream_grid(pieces)), dim3(kT), 0, s, xv, yv, pieces, C,     \
                     static_cast<const void *>(save_mean), static_cast<const void *>(save_invstd), weight,  \
                     bias, param_dtype, eps, 0, relu, n_live, n)
    SPX_BN_DISPATCH(dtype, SPX_BN_APPLY);
#undef SPX_BN_APPLY
  } else {
    SPX_CHECK(running_mean && running_var, "inference needs the running statistics");
#define SPX_BN_APPLY(D)                                                                                    \
  hipLaunchKernelGGL(bn_apply_kernel<D>, dim3(stream_grid(pieces)), dim3(kT), 0, s, xv, yv, pieces, C,     \
                     static_cast<const void *>(running_mean), static_cast<const void *>(running_var),      \
                     weight, bias, param_dtype, eps, 1, relu, n_live, n)
    SPX_BN_DISPATCH(dtype, SPX_BN_APPLY);
#undef SPX_BN_APPLY
  }
  SPX_LAUNCH_CHECK();
  return 0;
}

int spx_batchnorm_fwd(const void *x, void *y, int n, int C, int dtype, const void *weight,
                      const void *bias, void *running_mean, void *running_var,
                      long long *num_batches_tracked, int param_dtype, int training, float momentum,
                      float eps, int relu, float *save_mean, float *save_invstd, void *ws,
                      size_t ws_bytes, const int32_t *n_live, spx_stream_t stream) {
  return batchnorm_fwd_impl(x, y, n, C, dtype, weight, bias, running_mean, running_var, num_batches_tracked,
                            param_dtype, training, momentum, eps, relu, save_mean, save_invstd, ws, ws_bytes, n_live,
                            stream, nullptr, 0);
}

int spx_batchnorm_fwd_stats(const void *x, void *y, int n, int C, int dtype, const void *weight,
                            const void *bias, void *running_mean, void *running_var,
                            long long *num_batches_tracked, int param_dtype, float momentum, float eps, int relu,
                            float *save_mean, float *save_invstd, const float *stats, int stats_records,
                            const int32_t *n_live, spx_stream_t stream) {
  SPX_CHECK(stats && stats_records > 0, "statistics records required (spx_igemm_fwd_stats)");
  return batchnorm_fwd_impl(x, y, n, C, dtype, weight, bias, running_mean, running_var, num_batches_tracked,
                            param_dtype, 1, momentum, eps, relu, save_mean, save_invstd, nullptr, 0, n_live, stream,
                            stats, stats_records);
}

int spx_batchnorm_bwd(const void *x, const void *dy, void *dx, int n, int C, int dtype,
                      const void *weight, const void *bias, int param_dtype, const float *mean,
                      const float *invstd, int use_batch_stats, int relu, void *dweight, void *dbias,
                      void *ws, size_t ws_bytes, const int32_t *n_live, spx_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  SPX_CHECK(bn_shape_ok(C, dtype), "batchnorm: unsupported C = %d / dtype", C);
  SPX_CHECK(param_dtype == SPX_F32 || param_dtype == SPX_F16 || param_dtype == SPX_BF16, "bad parameter dtype");
  SPX_CHECK(mean && invstd, "null pointer");
  const size_t pbytes = param_dtype == SPX_F32 ? 4 : 2;
  if (n == 0) {
    if (dweight) SPX_HIP(hipMemsetAsync(dweight, 0, pbytes * C, s));
    if (dbias) SPX_HIP(hipMemsetAsync(dbias, 0, pbytes * C, s));
    return 0;
  }
  SPX_CHECK(x && dy && dx && ws && ws_bytes >= spx_batchnorm_ws_bytes(n, C), "null pointer / workspace too small");
  const int vpl = dtype == SPX_F32 ? 4 : 8;
  const long long pieces = static_cast<long long>(n) * (C / vpl);
  const int G = bn_blocks(n);
  float *partial = static_cast<float *>(ws);
  float *sums = partial + static_cast<size_t>(G) * 2 * C;      // [2][C] behind the partials
  const u32x4 *xv = static_cast<const u32x4 *>(x), *gv = static_cast<const u32x4 *>(dy);
#define SPX_BN_BP(D)                                                                                        \
  hipLaunchKernelGGL(bn_bwd_partial_kernel<D>, dim3(G), dim3(kT), 0, s, xv, gv, n, C, mean, invstd, weight, \
                     bias, param_dtype, relu, partial, n_live)
  if (bn_phases() & 1) SPX_BN_DISPATCH(dtype, SPX_BN_BP);
#undef SPX_BN_BP
  if (bn_phases() & 2)
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(kT), 0, s, partial, G, C, sums, dweight, dbias,
                     param_dtype);
#define SPX_BN_BA(D)                                                                                        \
  hipLaunchKernelGGL(bn_bwd_apply_kernel<D>, dim3(stream_grid(pieces)), dim3(kT), 0, s, xv, gv,             \
                     static_cast<u32x4 *>(dx), pieces, n, C, mean, invstd, weight, bias, param_dtype, sums,  \
                     relu, use_batch_stats, n_live)
  if (bn_phases() & 4) SPX_BN_DISPATCH(dtype, SPX_BN_BA);
#undef SPX_BN_BA
  SPX_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
