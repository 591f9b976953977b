// Shared host/device helpers for the gfx950 kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/spconv_amd.h"

namespace spx {

constexpr int kWave = 64;
constexpr int kMaxNdim = SPX_MAX_NDIM;

void set_error(const char *fmt, ...);

// Integer tuning knob from the environment (A/B measurements; defaults are the shipped choice).
inline int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v ? atoi(v) : dflt;
}

// The same, read at every call and overridable at run time through spx_set_option (switches a test or a
// benchmark flips inside one process; env_int values are usually cached in function-local statics).
int option_int(const char *name, int dflt);

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int div_up(int a, int b) { return (a + b - 1) / b; }

// Carves 256-byte aligned sub-buffers out of the caller's scratch.
struct Carver {
  char *base;
  size_t off = 0;
  explicit Carver(void *p) : base(static_cast<char *>(p)) {}
  template <typename T> T *take(size_t count) {
    T *r = reinterpret_cast<T *>(base + off);
    off += align_up(count * sizeof(T), 256);
    return r;
  }
};

#define SPX_CHECK(cond, ...)                 \
  do {                                       \
    if (!(cond)) {                           \
      ::spx::set_error(__VA_ARGS__);         \
      return -1;                             \
    }                                        \
  } while (0)

#define SPX_HIP(expr)                                                         \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) {                                                   \
      ::spx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                       __FILE__, __LINE__);                                   \
      return -2;                                                              \
    }                                                                         \
  } while (0)

#define SPX_LAUNCH_CHECK() SPX_HIP(hipGetLastError())

// A kernel that needs more than 64 KB of dynamic LDS raises hipFuncAttributeMaxDynamicSharedMemorySize first.  The
// attribute belongs to (function, DEVICE): `done` keeps one bit per device index, so a process that drives several
// GPUs sets it on each of them, from any thread (round-5 ADVICE: a plain static flag set it on the first device only).
inline hipError_t ensure_dynamic_lds(const void *fn, int bytes, std::atomic<uint64_t> &done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

// Canonical 4-d description of a 1..4-d problem: leading dims are padded with
// size-1 / ksize-1 entries so one kernel serves every ndim.  Column c of an
// index row (batch at column 0) maps to canonical spatial dim c-1+(4-ndim).
struct Geom {
  int ndim;           // user ndim
  int batch;
  int in_dims[4];
  int out_dims[4];
  int ksize[4], stride[4], padding[4], dilation[4];
  int kv;
};

inline Geom make_geom(int ndim, int batch, const int *in_dims, const int *out_dims,
                      const int *ksize, const int *stride, const int *padding,
                      const int *dilation) {
  Geom g;
  g.ndim = ndim;
  g.batch = batch;
  const int lead = 4 - ndim;
  g.kv = 1;
  for (int i = 0; i < 4; ++i) {
    const bool pad = i < lead;
    const int j = i - lead;
    g.in_dims[i] = pad ? 1 : in_dims[j];
    g.out_dims[i] = pad ? 1 : out_dims[j];
    g.ksize[i] = pad ? 1 : ksize[j];
    g.stride[i] = pad ? 1 : (stride ? stride[j] : 1);
    g.padding[i] = pad ? 0 : (padding ? padding[j] : 0);
    g.dilation[i] = pad ? 1 : (dilation ? dilation[j] : 1);
    g.kv *= g.ksize[i];
  }
  return g;
}

}  // namespace spx

// ---- launch counters (diagnostics: which kernel family a call dispatched; spx_launch_count) --------
namespace spx {
enum LaunchFamily { kFamV4 = 0, kFamWs, kFamBwdFused, kFamBwdRows, kFamI8Stream, kFamGeneric, kFamStage2, kFamStage2Batch, kFamCount };
extern std::atomic<long long> g_launches[kFamCount];
inline void count_launch(LaunchFamily f) { g_launches[f].fetch_add(1, std::memory_order_relaxed); }
}  // namespace spx

// ---- row orders (rowsort.hip) ---------------------------------------------------------------------
namespace spx {
// stable LSD radix argsort behind spx_mask_argsort (rowsort.hip)
size_t radix_argsort_ws_bytes(int n);
int radix_argsort(const uint32_t *keys, int n, int nbits, int32_t *order_out, void *ws, hipStream_t s);

}  // namespace spx
