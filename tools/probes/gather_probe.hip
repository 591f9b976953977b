// Micro-benchmark: how fast can a CU gather random 128-byte rows with 16 bytes per lane, depending on
// WHICH lanes read the same row?  (build: hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip)
//   mode 0: MFMA operand layout of igemm_v4 -- lane l reads piece (l >> 4) of row (l & 15): the four lanes of
//           one row sit in four different quads;
//   mode 1: quad-coherent -- lane l reads piece (l & 3) of row (l >> 2): 64 contiguous bytes per quad;
//   mode 2: 8 lanes per row -- lane l reads piece (l & 7) of row (l >> 3): a full 128-byte line per 8 lanes.
// Every wave instruction moves 1 KB in all modes; rows are random (a permutation of n rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) gather_kernel(const u32x4 *__restrict__ feat, const int *__restrict__ idx,
                                                      int n_items, u32x4 *__restrict__ out, int iters) {
  // one item = 16 rows (modes 0, 1: half rows x 2 instrs; here: 64 bytes per row and instruction)
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * 256) >> 6;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = wave_global; it < n_items; it += nwaves) {
    int row, piece;
    if (MODE == 0) { row = idx[it * 16 + (lane & 15)]; piece = lane >> 4; }
    else if (MODE == 1) { row = idx[it * 16 + (lane >> 2)]; piece = lane & 3; }
    else { row = idx[it * 16 + (lane >> 3) + ((lane >> 3) & 0)]; piece = lane & 7; }
    if (MODE == 2) {
      // 8 rows x 128 B per instruction, two instructions per item
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r2 = idx[it * 16 + h * 8 + (lane >> 3)];
        const u32x4 v = feat[(size_t)r2 * 8 + piece];
        acc += v;
      }
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const u32x4 v = feat[(size_t)row * 8 + h * 4 + piece];
        acc += v;
      }
    }
  }
  if (acc.x == 0x12345678u) out[threadIdx.x] = acc;    // keep the loads alive
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 125562;          // rows of 128 B
  const int reps = argc > 2 ? atoi(argv[2]) : 6;            // gathers per row (pairs per voxel)
  std::vector<int> idx((size_t)n * reps);
  std::mt19937 rng(1);
  for (int r = 0; r < reps; ++r) {
    std::vector<int> p(n);
    for (int i = 0; i < n; ++i) p[i] = i;
    std::shuffle(p.begin(), p.end(), rng);
    std::copy(p.begin(), p.end(), idx.begin() + (size_t)r * n);
  }
  const int items = (int)(idx.size() / 16);
  u32x4 *feat, *out;
  int *d_idx;
  hipMalloc(&feat, (size_t)n * 128);
  hipMalloc(&out, 4096);
  hipMalloc(&d_idx, idx.size() * 4);
  hipMemset(feat, 1, (size_t)n * 128);
  hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int mode = 0; mode < 3; ++mode) {
    for (int grid : {768, 2048}) {
      float best = 1e9f;
      for (int t = 0; t < 6; ++t) {
        hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(gather_kernel<0>, dim3(grid), dim3(256), 0, 0, feat, d_idx, items, out, 1);
        if (mode == 1) hipLaunchKernelGGL(gather_kernel<1>, dim3(grid), dim3(256), 0, 0, feat, d_idx, items, out, 1);
        if (mode == 2) hipLaunchKernelGGL(gather_kernel<2>, dim3(grid), dim3(256), 0, 0, feat, d_idx, items, out, 1);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (t > 0 && ms < best) best = ms;
      }
      const double bytes = (double)items * 16 * 128;
      printf("mode %d grid %4d: %7.2f us  %6.1f GB/s  (%d rows x %d gathers)\n", mode, grid, best * 1e3,
             bytes / (best * 1e-3) / 1e9, n, reps);
    }
  }
  return 0;
}
