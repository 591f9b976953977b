// Sparse pooling over the rulebook tables.
//
// Replaces the reference's IndiceMaxPool implicit-GEMM kernels
// (spconv/csrc/sparse/maxpool.py:96-262: forward / backward max pool, forward / backward average
// pool).  Pure bandwidth kernels: one thread owns a 16-byte piece of one output row, walks the
// offsets present for that row (set bits of the rulebook mask word when there is one, every
// offset otherwise), gathers the source pieces as full-width loads and reduces in registers;
// every row is written exactly once, no atomics.
#include "common.h"

namespace spx {
namespace {

constexpr int kBlock = 256;

struct PoolParams {
  const void *src;        // gathered tensor: features (forward) / dout (backward)
  const void *feat;       // max-pool backward: input features [n_in, C]
  const void *out;        // max-pool backward: forward output [n_out, C]
  void *dst;              // [n_dst, C]
  int32_t *count_out;     // avg-pool forward: valid pairs per output row, or null
  const int32_t *count;   // avg-pool backward: the same counts
  int quirk_mul;          // avg-pool backward: multiply by the count like the reference kernel (SPCONV_AMD_REFERENCE_QUIRKS)
  const int32_t *pair;    // [kv, n_dst]
  const uint32_t *mask;   // [n_dst, ceil(kv / 32)] or null
  int n_dst, C, kv, init_zero;
};

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float to_f(float v) { return v; }
  static __device__ __forceinline__ float from_f(float v) { return v; }
  static __device__ __forceinline__ float lowest() { return -3.402823466e+38f; }
};
template <> struct Elem<_Float16> {
  static __device__ __forceinline__ float to_f(_Float16 v) { return static_cast<float>(v); }
  static __device__ __forceinline__ _Float16 from_f(float v) { return static_cast<_Float16>(v); }
  static __device__ __forceinline__ _Float16 lowest() { return static_cast<_Float16>(-65504.f); }
};
template <> struct Elem<__bf16> {
  static __device__ __forceinline__ float to_f(__bf16 v) { return static_cast<float>(v); }
  static __device__ __forceinline__ __bf16 from_f(float v) { return static_cast<__bf16>(v); }
  static __device__ __forceinline__ __bf16 lowest() {
    return __builtin_bit_cast(__bf16, static_cast<unsigned short>(0xff7f));
  }
};
template <> struct Elem<int8_t> {
  static __device__ __forceinline__ float to_f(int8_t v) { return static_cast<float>(v); }
  static __device__ __forceinline__ int8_t from_f(float v) { return static_cast<int8_t>(v); }
  static __device__ __forceinline__ int8_t lowest() { return -128; }
};

template <typename T, int V> struct Piece { T v[V]; };

// walks the offsets of row `r`: calls f(k, idx) for every valid pair
template <typename F>
__device__ __forceinline__ void for_each_pair(const PoolParams &p, int r, F f) {
  const int words = (p.kv + 31) >> 5;
  for (int w = 0; w < words; ++w) {
    uint32_t bits = p.mask ? p.mask[static_cast<size_t>(r) * words + w] : 0xffffffffu;
    if (w == words - 1 && (p.kv & 31)) bits &= (1u << (p.kv & 31)) - 1u;
    while (bits) {
      const int k = w * 32 + __builtin_ctz(bits);
      bits &= bits - 1;
      const int idx = p.pair[static_cast<size_t>(k) * p.n_dst + r];
      if (idx >= 0) f(k, idx);
    }
  }
}

enum PoolOp { kMaxFwd = 0, kMaxBwd = 1, kAvgFwd = 2, kAvgBwd = 3 };

// V elements (16 bytes when the row length allows it, 1 element otherwise) per thread
template <typename T, int V, int OP>
__global__ void __launch_bounds__(kBlock) pool_kernel(PoolParams p) {
  const int pieces = p.C / V;
  const long long gid = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x;
  if (gid >= static_cast<long long>(p.n_dst) * pieces) return;
  const int r = static_cast<int>(gid / pieces), c = static_cast<int>(gid % pieces) * V;
  typedef Piece<T, V> P;
  const T *src = static_cast<const T *>(p.src);
  T *dst = static_cast<T *>(p.dst);
  if (OP == kMaxFwd) {
    // maxpool.py:96-140: start from the lowest value, `in < in_temp` keeps the current value
    // for NaN.  init_zero reproduces the Native path, whose output starts as zeros
    // (pytorch/ops.py:1910, maxpool.py:36-60).
    P cur;
#pragma unroll
    for (int e = 0; e < V; ++e) cur.v[e] = p.init_zero ? Elem<T>::from_f(0.f) : Elem<T>::lowest();
    bool any = false;
    for_each_pair(p, r, [&](int, int idx) {
      const P in = *reinterpret_cast<const P *>(src + static_cast<size_t>(idx) * p.C + c);
      any = true;
#pragma unroll
      for (int e = 0; e < V; ++e)
        if (Elem<T>::to_f(cur.v[e]) < Elem<T>::to_f(in.v[e])) cur.v[e] = in.v[e];
    });
    // a row without a single pair is a DEAD row of a static-shape tensor (a real output has at least the pair that
    // created it): zeros, like every other padding row, not the lowest value
    if (!any) {
#pragma unroll
      for (int e = 0; e < V; ++e) cur.v[e] = Elem<T>::from_f(0.f);
    }
    *reinterpret_cast<P *>(dst + static_cast<size_t>(r) * p.C + c) = cur;
  } else if (OP == kMaxBwd) {
    // maxpool.py:142-209: din[i] = sum of dout[o] over the outputs whose maximum this input is
    const P in = *reinterpret_cast<const P *>(static_cast<const T *>(p.feat) + static_cast<size_t>(r) * p.C + c);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    for_each_pair(p, r, [&](int, int o) {
      const P out = *reinterpret_cast<const P *>(static_cast<const T *>(p.out) + static_cast<size_t>(o) * p.C + c);
      const P d = *reinterpret_cast<const P *>(src + static_cast<size_t>(o) * p.C + c);
#pragma unroll
      for (int e = 0; e < V; ++e)
        if (Elem<T>::to_f(in.v[e]) == Elem<T>::to_f(out.v[e])) acc[e] += Elem<T>::to_f(d.v[e]);
    });
    P res;
#pragma unroll
    for (int e = 0; e < V; ++e) res.v[e] = Elem<T>::from_f(acc[e]);
    *reinterpret_cast<P *>(dst + static_cast<size_t>(r) * p.C + c) = res;
  } else if (OP == kAvgFwd) {
    // maxpool.py:211-260: mean over the valid pairs, 0 for a row without pairs
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    int count = 0;
    for_each_pair(p, r, [&](int, int idx) {
      const P in = *reinterpret_cast<const P *>(src + static_cast<size_t>(idx) * p.C + c);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += Elem<T>::to_f(in.v[e]);
      ++count;
    });
    if (p.count_out && c == 0) p.count_out[r] = count;
    const float inv = count > 0 ? 1.f / static_cast<float>(count) : 0.f;
    P res;
#pragma unroll
    for (int e = 0; e < V; ++e) res.v[e] = Elem<T>::from_f(acc[e] * inv);
    *reinterpret_cast<P *>(dst + static_cast<size_t>(r) * p.C + c) = res;
  } else {
    // gradient of the mean: din[i] = sum_o dout[o] / count[o].  (The reference kernel,
    // maxpool.py:262-300, MULTIPLIES by count[o]; that is not the derivative of its own
    // forward -- DESIGN.md.  SPCONV_AMD_REFERENCE_QUIRKS=1 reproduces it: quirk_mul.)
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    for_each_pair(p, r, [&](int, int o) {
      const P d = *reinterpret_cast<const P *>(src + static_cast<size_t>(o) * p.C + c);
      const int cnt = p.count[o];
      const float inv = p.quirk_mul ? static_cast<float>(cnt) : (cnt > 0 ? 1.f / static_cast<float>(cnt) : 0.f);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += Elem<T>::to_f(d.v[e]) * inv;
    });
    P res;
#pragma unroll
    for (int e = 0; e < V; ++e) res.v[e] = Elem<T>::from_f(acc[e]);
    *reinterpret_cast<P *>(dst + static_cast<size_t>(r) * p.C + c) = res;
  }
}

template <typename T, int OP>
int launch_pool(const PoolParams &p, hipStream_t s) {
  constexpr int V = 16 / static_cast<int>(sizeof(T));
  if (p.n_dst == 0) return 0;
  if (p.C % V == 0) {
    const long long total = static_cast<long long>(p.n_dst) * (p.C / V);
    hipLaunchKernelGGL((pool_kernel<T, V, OP>), dim3(static_cast<unsigned>((total + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, p);
  } else {
    const long long total = static_cast<long long>(p.n_dst) * p.C;
    hipLaunchKernelGGL((pool_kernel<T, 1, OP>), dim3(static_cast<unsigned>((total + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, p);
  }
  SPX_LAUNCH_CHECK();
  return 0;
}

template <int OP>
int dispatch_pool(const PoolParams &p, int dtype, hipStream_t s) {
  switch (dtype) {
    case SPX_F32: return launch_pool<float, OP>(p, s);
    case SPX_F16: return launch_pool<_Float16, OP>(p, s);
    case SPX_BF16: return launch_pool<__bf16, OP>(p, s);
    case SPX_I8:
      if (OP == kMaxFwd) return launch_pool<int8_t, OP>(p, s);
      break;
  }
  set_error("unsupported dtype %d for this pooling op", dtype);
  return -1;
}

}  // namespace
}  // namespace spx

using namespace spx;

extern "C" {

int spx_maxpool_fwd(const void *feat, void *out, const int32_t *pair_fwd, const uint32_t *mask,
                    int n_out, int C, int kv, int dtype, int init_zero, spx_stream_t stream) {
  SPX_CHECK(feat && out && pair_fwd, "null pointer");
  SPX_CHECK(C > 0 && kv > 0 && n_out >= 0, "bad sizes");
  PoolParams p{};
  p.src = feat;
  p.dst = out;
  p.pair = pair_fwd;
  p.mask = mask;
  p.n_dst = n_out;
  p.C = C;
  p.kv = kv;
  p.init_zero = init_zero;
  return dispatch_pool<kMaxFwd>(p, dtype, static_cast<hipStream_t>(stream));
}

int spx_maxpool_bwd(const void *feat, const void *out, const void *dout, void *din,
                    const int32_t *pair_bwd, const uint32_t *mask_bwd, int n_in, int C, int kv,
                    int dtype, spx_stream_t stream) {
  SPX_CHECK(feat && out && dout && din && pair_bwd, "null pointer");
  SPX_CHECK(C > 0 && kv > 0 && n_in >= 0, "bad sizes");
  PoolParams p{};
  p.src = dout;
  p.feat = feat;
  p.out = out;
  p.dst = din;
  p.pair = pair_bwd;
  p.mask = mask_bwd;
  p.n_dst = n_in;
  p.C = C;
  p.kv = kv;
  return dispatch_pool<kMaxBwd>(p, dtype, static_cast<hipStream_t>(stream));
}

int spx_avgpool_fwd(const void *feat, void *out, int32_t *count_out, const int32_t *pair_fwd,
                    const uint32_t *mask, int n_out, int C, int kv, int dtype,
                    spx_stream_t stream) {
  SPX_CHECK(feat && out && pair_fwd, "null pointer");
  SPX_CHECK(C > 0 && kv > 0 && n_out >= 0, "bad sizes");
  PoolParams p{};
  p.src = feat;
  p.dst = out;
  p.count_out = count_out;
  p.pair = pair_fwd;
  p.mask = mask;
  p.n_dst = n_out;
  p.C = C;
  p.kv = kv;
  return dispatch_pool<kAvgFwd>(p, dtype, static_cast<hipStream_t>(stream));
}

int spx_avgpool_bwd(const void *dout, void *din, const int32_t *count, const int32_t *pair_bwd,
                    const uint32_t *mask_bwd, int n_in, int C, int kv, int dtype,
                    spx_stream_t stream) {
  SPX_CHECK(dout && din && count && pair_bwd, "null pointer");
  SPX_CHECK(C > 0 && kv > 0 && n_in >= 0, "bad sizes");
  PoolParams p{};
  p.src = dout;
  p.dst = din;
  p.count = count;
  p.quirk_mul = option_int("SPCONV_AMD_REFERENCE_QUIRKS", 0) ? 1 : 0;
  p.pair = pair_bwd;
  p.mask = mask_bwd;
  p.n_dst = n_in;
  p.C = C;
  p.kv = kv;
  return dispatch_pool<kAvgBwd>(p, dtype, static_cast<hipStream_t>(stream));
}

}  // extern "C"
