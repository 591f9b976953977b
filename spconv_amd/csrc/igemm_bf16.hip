// bf16 instantiations of the gather-GEMM (igemm_v4.h) and of the fused backward launch (igemm_bwd.h): their own
// translation unit so that a full rebuild compiles the operand types side by side (csrc/build.sh).
#include "igemm_bwd.h"

namespace spx {

int dispatch_gather_gemm_bf16(const GemmParams &p, hipStream_t s) { return dispatch_gather_gemm<true>(p, s); }

int dispatch_bwd_bf16(const GemmParams &p, const Wgrad2Params &q, int n_wgrad_blocks, hipStream_t s) {
  return dispatch_bwd<1>(p, q, n_wgrad_blocks, s);
}

}  // namespace spx
