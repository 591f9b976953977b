// Host-only pieces of the C ABI: error channel, version, output-shape math.
#include "common.h"

#include <string.h>

#include <mutex>
#include <string>

namespace spx {
namespace {
thread_local std::string g_error;
}

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
}

namespace {
struct Option {
  char name[48];
  int value;
};
Option g_options[32];
int g_noptions = 0;
std::mutex g_option_mutex;
}  // namespace

std::atomic<long long> g_launches[kFamCount];

int option_int(const char *name, int dflt) {
  {
    std::lock_guard<std::mutex> lock(g_option_mutex);
    for (int i = 0; i < g_noptions; ++i)
      if (strcmp(g_options[i].name, name) == 0) return g_options[i].value;
  }
  return env_int(name, dflt);
}
}  // namespace spx

extern "C" {

int spx_set_option(const char *name_h, int value) {
  SPX_CHECK(name_h && strlen(name_h) < sizeof(spx::Option::name), "bad option name");
  std::lock_guard<std::mutex> lock(spx::g_option_mutex);
  for (int i = 0; i < spx::g_noptions; ++i)
    if (strcmp(spx::g_options[i].name, name_h) == 0) {
      spx::g_options[i].value = value;
      return 0;
    }
  SPX_CHECK(spx::g_noptions < 32, "too many options");
  strcpy(spx::g_options[spx::g_noptions].name, name_h);
  spx::g_options[spx::g_noptions++].value = value;
  return 0;
}

const char *spx_last_error(void) { return spx::g_error.c_str(); }

long long spx_launch_count(const char *family_h) {
  static const char *names[spx::kFamCount] = {"igemm_v4", "igemm_ws", "igemm_bwd", "igemm_bwd_rows", "igemm_i8_stream",
                                              "generic", "wgrad_stage2", "wgrad_stage2_batch"};
  if (!family_h) return -1;
  for (int i = 0; i < spx::kFamCount; ++i)
    if (strcmp(names[i], family_h) == 0) return spx::g_launches[i].load(std::memory_order_relaxed);
  return -1;
}

int spx_version(void) { return 1000; }

// ops.get_conv_output_size / get_deconv_output_size (pytorch/ops.py:73-96)
int spx_conv_out_shape(int ndim, const int *in_shape, const int *ksize, const int *stride,
                       const int *padding, const int *dilation, const int *out_padding,
                       int transposed, int *out_shape) {
  SPX_CHECK(ndim >= 1 && ndim <= SPX_MAX_NDIM, "ndim must be in [1,4], got %d", ndim);
  for (int i = 0; i < ndim; ++i) {
    SPX_CHECK(stride[i] > 0, "stride must be positive");
    if (transposed) {
      SPX_CHECK(ksize[i] != -1, "deconv don't support kernel_size < 0");
      out_shape[i] = (in_shape[i] - 1) * stride[i] - 2 * padding[i] + ksize[i] +
                     (out_padding ? out_padding[i] : 0);
    } else if (ksize[i] == -1) {
      out_shape[i] = 1;
    } else {
      const int num = in_shape[i] + 2 * padding[i] - dilation[i] * (ksize[i] - 1) - 1;
      int q = num / stride[i];
      if (num % stride[i] != 0 && num < 0) --q;  // python floor division
      out_shape[i] = q + 1;
    }
  }
  return 0;
}

}  // extern "C"
