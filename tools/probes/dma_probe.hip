// Probe (gfx950): what the buffer unit does with (a) a dwordx4 access that is only partly inside
// num_records, to VGPRs and as LDS-DMA, (b) LDS-DMA dwordx4 from a source address that is only
// 4-byte aligned.  Answers the two assumptions igemm5.hip's table prefetch avoids / relies on.
//   hipcc --offload-arch=gfx950 dma_probe.hip -o dma_probe && ./dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk(const void *p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, static_cast<int>(bytes), 0x00020000);
}
// out[0..3]: VGPR load of 16 bytes at byte offset `off` with num_records = nrec
// out[4..7]: the same through LDS-DMA; out[8 + 4 l ..]: LDS-DMA of lane l from offset off + 16 l
__global__ void k(const uint32_t *src, uint32_t nrec, uint32_t off, uint32_t *out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = mk(src, nrec);
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off + lane * 16, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)lds, 16, off + lane * 16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = 0; e < 4; ++e) {
    out[lane * 8 + e] = v[e];
    out[lane * 8 + 4 + e] = lds[lane * 4 + e];
  }
}
int main() {
  const int N = 1024;
  uint32_t h[N], *d, *o, ho[64 * 8];
  for (int i = 0; i < N; ++i) h[i] = 1000 + i;
  hipMalloc(&d, sizeof(h));
  hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  struct { uint32_t nrec, off; const char *what; } cases[] = {
      {4096, 0, "aligned, in range"},
      {4096, 4, "source 4-byte aligned only (offset 4)"},
      {4096, 36, "source 4-byte aligned only (offset 36)"},
      {24, 16, "lane 0 piece [16,32) with num_records = 24: dwords 0,1 inside, 2,3 outside"},
      {1000, 0, "num_records = 1000: lane 62 piece [992,1008) straddles"},
  };
  for (auto &c : cases) {
    hipMemset(o, 0, sizeof(ho));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c.nrec, c.off, o);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    printf("%s\n", c.what);
    for (int l : {0, 1, 62, 63}) {
      printf("  lane %2d expect %4u..: vgpr %u %u %u %u | lds %u %u %u %u\n", l, 1000 + (c.off + l * 16) / 4,
             ho[l * 8], ho[l * 8 + 1], ho[l * 8 + 2], ho[l * 8 + 3], ho[l * 8 + 4], ho[l * 8 + 5], ho[l * 8 + 6],
             ho[l * 8 + 7]);
    }
  }
  return 0;
}
