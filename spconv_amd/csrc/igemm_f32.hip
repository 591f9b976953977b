// fp32 instantiations of the gather-GEMM (igemm_v4.h: v_mfma_f32_16x16x4_f32, exact fp32 at 1/16 of the 16-bit MFMA rate)
// and of the fused backward launch (igemm_bwd.h: dgrad tiles + wgrad_f32 ranges); BASELINE config 1.
#include "igemm_bwd.h"

namespace spx {

// fp32 tensors: 128-row tiles (256 output channels: 64 rows)
int dispatch_gather_gemm_f32(const GemmParams &p, hipStream_t s) {
  switch (p.COUT) {
    case 16: return launch_v4<16, 2, 3>(p, s);
    case 32: return launch_v4<32, 2, 3>(p, s);
    case 64: return launch_v4<64, 2, 3>(p, s);
    case 128: return launch_v4<128, 2, 3>(p, s);
    case 256: return launch_v4<256, 1, 3>(p, s);
  }
  return -1;
}

int dispatch_bwd_f32(const GemmParams &p, const Wgrad2Params &q, int n_wgrad_blocks, hipStream_t s) {
  return dispatch_bwd<3>(p, q, n_wgrad_blocks, s);
}

}  // namespace spx
