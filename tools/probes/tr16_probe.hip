// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  Every lane l points at its own 8-byte
// chunk holding the values 4*l + e (e = 0..3); the output shows which (lane, element) each
// result element came from.   hipcc --offload-arch=gfx950 tr16_probe.hip -o tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[256];
  const int l = threadIdx.x;
  for (int e = 0; e < 4; ++e) tile[l * 4 + e] = l * 4 + e;
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)&tile[l * 4]);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
  unsigned short* d; unsigned short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf(" (l%2d,e%d)", h[l * 4 + e] / 4, h[l * 4 + e] % 4);
    printf("\n");
  }
  return 0;
}
