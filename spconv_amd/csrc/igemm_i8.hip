// int8 instantiations of the gather-GEMM (igemm_v4.h: v_mfma_i32_16x16x64_i8, i32 accumulators, quantised epilogue;
// forward only) behind spx_igemm_fwd_int8 (reference: spconv/pytorch/quantization/quantized/conv.py:368-378).
#include "igemm_v4.h"

namespace spx {

// rows64: tile height by density class -- tables in tile order or the host's sparse hint = a rulebook classified as
// SPARSE -> 64-row tiles at 128 output channels (70 instead of 165 registers per lane, five workgroups per CU instead of
// three: 27.3 -> 25.7 us at BASELINE config 5); dense neighbourhoods keep 128 rows (LiDAR-like 200 k voxels: 104 vs
// 113 us, fixture 74 vs 83 us).
int launch_gather_gemm_int8(const GemmParams &p, bool rows64, hipStream_t s) {
  // the host has seen class word 1 of this rows layout (SPX_SPARSE_HINT: p.app_rows > 0): appendix tiles + streaming main
  // tiles in one launch (igemm_v4.h: igemm_i8_sparse_kernel); SPX_I8_STREAM = 0 keeps the tile-per-workgroup launch (A/B)
  if (rows64 && p.app_rows > 0 && i8_sparse_ok(p) && option_int("SPX_I8_STREAM", 1) != 0)
    return p.COUT == 128 ? launch_i8_sparse<128>(p, s) : launch_i8_sparse<64>(p, s);
  switch (p.COUT) {
    case 16: return launch_v4<16, 2, 2>(p, s);
    case 32: return launch_v4<32, 2, 2>(p, s);
    case 64: return launch_v4<64, 2, 2>(p, s);
    case 128: return rows64 ? launch_v4<128, 1, 2>(p, s) : launch_v4<128, 2, 2>(p, s);
    case 256: return launch_v4<256, 1, 2>(p, s);
  }
  return -1;
}

}  // namespace spx
