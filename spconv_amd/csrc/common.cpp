// Host-only pieces of the C ABI: error channel, version, output-shape math.
#include "common.h"

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
}  // namespace spx

extern "C" {

const char *spx_last_error(void) { return spx::g_error.c_str(); }

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
