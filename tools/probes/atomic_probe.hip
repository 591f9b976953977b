// Probe (gfx950): what do device-scope atomics on a hash table cost?  N random slots of a table, one operation per
// thread, all CUs: (a) plain 8-byte loads, (b) 64-bit atomicCAS on EMPTY slots (an insert into an empty table),
// (c) atomicCAS where `dup` threads hit the same slot (the contended insert of a dense strided convolution),
// (d) the same with a look (plain load) before the atomic, (e) atomicOr without return (mask words).
// The rulebook builders' insert passes are made of (b)-(d); their probe passes of (a).
//   hipcc --offload-arch=gfx950 -O3 atomic_probe.hip -o atomic_probe && ./atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned long long u64;
__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
template <int MODE>
__global__ void k(u64 *table, uint32_t mask, int n, int dup, uint32_t *sink) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t key = static_cast<uint32_t>(i) / dup;             // dup threads share a key
  const uint32_t slot = mix(key) & mask;
  const u64 want = (static_cast<u64>(key) << 32) | static_cast<uint32_t>(i);
  uint32_t r = 0;
  if (MODE == 0) {
    r = static_cast<uint32_t>(table[slot]);
  } else if (MODE == 1) {
    const u64 prev = atomicCAS(&table[slot], ~0ull, want);
    if (prev != ~0ull && static_cast<uint32_t>(prev) > static_cast<uint32_t>(want)) atomicMin(&table[slot], want);
    r = static_cast<uint32_t>(prev);
  } else if (MODE == 2) {
    u64 cur = table[slot];
    if (cur == ~0ull) cur = atomicCAS(&table[slot], ~0ull, want);
    if (cur != ~0ull && static_cast<uint32_t>(cur) > static_cast<uint32_t>(want)) atomicMin(&table[slot], want);
    r = static_cast<uint32_t>(cur);
  } else {
    atomicOr(reinterpret_cast<uint32_t *>(table) + slot, 1u << (i & 31));
  }
  if (r == 0x12345678u) sink[0] = r;
}
int main() {
  const int cap_log = 23;                       // 8 M slots x 8 B = 64 MB (beyond the L2s), and 1 M slots = 8 MB
  u64 *table; uint32_t *sink;
  hipMalloc(&table, sizeof(u64) << cap_log); hipMalloc(&sink, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int cl : {20, 23}) for (int n : {400000, 1400000}) for (int dup : {1, 4}) for (int mode = 0; mode < 4; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemset(table, 0xff, sizeof(u64) << cl);
      hipDeviceSynchronize();
      hipEventRecord(a);
      const dim3 g((n + 255) / 256), t(256);
      const uint32_t m = (1u << cl) - 1u;
      if (mode == 0) hipLaunchKernelGGL(k<0>, g, t, 0, 0, table, m, n, dup, sink);
      if (mode == 1) hipLaunchKernelGGL(k<1>, g, t, 0, 0, table, m, n, dup, sink);
      if (mode == 2) hipLaunchKernelGGL(k<2>, g, t, 0, 0, table, m, n, dup, sink);
      if (mode == 3) hipLaunchKernelGGL(k<3>, g, t, 0, 0, table, m, n, dup, sink);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      best = ms < best ? ms : best;
    }
    const char *names[] = {"load 8 B", "CAS (+min)", "look, CAS (+min)", "atomicOr no return"};
    printf("table %3d MB  ops %7d  dup %d  %-20s %7.1f us  %6.1f G ops/s\n", (8 << cl) >> 20, n, dup, names[mode],
           best * 1e3f, n / (best * 1e-3f) * 1e-9f);
  }
  return 0;
}
