// Rulebook ("indice pair") builder for gfx950.
//
// Design (MI355X-first, not a translation of the reference kernels):
//  * one open-addressing hash table in global memory.  Whenever the key space (batch x grid
//    volume) fits 32 bits -- every configuration in BASELINE.json does -- a slot is ONE 64-bit
//    word {key32 : value32}: an insert is a single atomicCAS (plus an atomicMin only when a
//    duplicate key has to lower the value) and a probe is a single 8-byte load.  Larger key
//    spaces use separate int64 key / int32 value arrays.  At 2x load headroom the packed table
//    is 2 MB per 100k voxels and stays in the 4 MiB XCD-local L2;
//  * NO order-dependent atomics anywhere: duplicate keys are resolved with
//    atomicMin (smallest index wins == the CPU path's unordered_map::insert),
//    the dense tables are written by the thread that owns the row (coalesced
//    along the voxel axis), and every compaction / numbering step is a
//    count -> scan -> scatter pipeline built on wave64 ballot + mbcnt prefix
//    sums.  The result is therefore bit-identical to the reference CPU loops
//    (csrc/sparse/indices.py:1639-1778), including list order and the
//    first-seen numbering of regular-conv outputs;
//  * kernel boundaries are the only inter-workgroup synchronisation (XCD L2s
//    are not coherent inside a launch).
#include "common.h"

#include <algorithm>
#include <mutex>
#include <utility>

namespace spx {
namespace {

constexpr int kBlock = 256;
constexpr int kItems = 2048;  // entries per block in count/scatter passes (8 x 256)
typedef long long hkey_t;      // EMPTY == -1

struct Table {
  hkey_t *keys;    // wide: keys[cap].  packed: slots[cap], slot = (key32 << 32) | value32
  int32_t *vals;   // wide: vals[cap].  packed: the low halves of the slots (stride 2)
  uint32_t mask;   // capacity - 1 (capacity is a power of two)
  int packed;      // 1 when every key fits 32 bits
  int gbits;       // low key bits that pick the slot inside a group of 2^gbits slots (see hash_key)
  uint32_t max_probe;  // longest probe walk (slots - 1).  A table sized for the guaranteed bound is never more than
                       // half full and walks a handful of slots; a table sized for the outputs EXPECTED
                       // (table_shrink: static bound / last ratio) can fill up, and without a cap every insert and
                       // lookup of a key that no longer fits would walk all of it -- O(capacity) per candidate.
                       // Inserts and lookups share the cap, so a key that went in is found; one that did not fit
                       // raises the overflow flag of its pass (the count's read-back / the static form's counter).
};
constexpr uint32_t kMaxProbeShrunk = 2047;

constexpr unsigned long long kEmptySlot = ~0ull;

// Home slot of a key: the murmur3 finaliser of key >> gbits picks a group of 2^gbits slots, the low key
// bits the slot inside it (gbits = 3: 8 consecutive cells along the last spatial dimension share one
// 64-byte line of the table).  Measured and left at gbits = 0 everywhere: on the half-full SubM tables
// groups that are either empty or full turn every collision into a walk across a full group (fixture
// rulebook 78 -> 139 us); on the 4-8 % full regular-conv tables the lookups get 5-12 % faster
// (conv_count_first 8.5 -> 7.5, conv_assign 19.6 -> 18.0 us) but the inserts of neighbouring threads
// now contend for the same lines (conv_stage1 22.6 -> 30.6 us).  Results never depend on the slot.
__device__ __forceinline__ uint32_t hash_key(hkey_t k, int gbits) {
  // murmur3 fmix64
  unsigned long long x = static_cast<unsigned long long>(k) >> gbits;
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return (static_cast<uint32_t>(x) << gbits) | (static_cast<uint32_t>(k) & ((1u << gbits) - 1u));
}

__device__ __forceinline__ uint32_t hash_key32(uint32_t k, int gbits) {
  // murmur3 fmix32
  uint32_t x = k >> gbits;
  x ^= x >> 16;
  x *= 0x85ebca6bu;
  x ^= x >> 13;
  x *= 0xc2b2ae35u;
  x ^= x >> 16;
  return (x << gbits) | (k & ((1u << gbits) - 1u));
}

// Value stored in a slot returned by table_insert_min.
__device__ __forceinline__ int32_t table_val(const Table &t, int slot) {
  return t.vals[static_cast<size_t>(slot) << t.packed];
}

// Inserts key (if absent) and lowers its value to min(value, val). Returns the slot.
template <bool LOOK = false>
__device__ __forceinline__ int table_insert_min(const Table &t, hkey_t key, int32_t val) {
  if (t.packed) {
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(t.keys);
    const uint32_t k32 = static_cast<uint32_t>(key);
    const unsigned long long want =
        (static_cast<unsigned long long>(k32) << 32) | static_cast<uint32_t>(val);
    uint32_t slot = hash_key32(k32, t.gbits) & t.mask;
    for (uint32_t probe = 0; probe <= t.max_probe; ++probe) {  // bounded: the table is never full (or capped)
      // look before the atomic: a slot only ever goes empty -> key, and its value only decreases, so a
      // (possibly stale) plain read that shows our key with a value <= ours, or another key, is final --
      // several inputs reach the same output on dense scenes, and all but the winner leave here
      // (LOOK: regular-conv builders; SubM keys are distinct, there the read would only add latency)
      unsigned long long cur = LOOK ? slots[slot] : kEmptySlot;
      if (cur == kEmptySlot) {
        cur = atomicCAS(&slots[slot], kEmptySlot, want);
        if (cur == kEmptySlot) return static_cast<int>(slot);
      }
      if (static_cast<uint32_t>(cur >> 32) == k32) {
        if (static_cast<uint32_t>(cur) > static_cast<uint32_t>(val)) atomicMin(&slots[slot], want);
        return static_cast<int>(slot);
      }
      slot = (slot + (1u << t.gbits)) & t.mask;      // (stays in its in-line position, see hash_key)
    }
    return -1;
  }
  uint32_t slot = hash_key(key, t.gbits) & t.mask;
  for (uint32_t probe = 0; probe <= t.max_probe; ++probe) {
    unsigned long long prev = LOOK ? static_cast<unsigned long long>(t.keys[slot])     // (as above)
                                   : static_cast<unsigned long long>(-1LL);
    if (prev == static_cast<unsigned long long>(-1LL))
      prev = atomicCAS(reinterpret_cast<unsigned long long *>(&t.keys[slot]),
                       static_cast<unsigned long long>(-1LL), static_cast<unsigned long long>(key));
    if (prev == static_cast<unsigned long long>(-1LL) ||
        prev == static_cast<unsigned long long>(key)) {
      // values start as 0xFFFFFFFF (one memset with the keys): unsigned min
      if (!LOOK || static_cast<unsigned int>(t.vals[slot]) > static_cast<unsigned int>(val))
        atomicMin(reinterpret_cast<unsigned int *>(&t.vals[slot]), static_cast<unsigned int>(val));
      return static_cast<int>(slot);
    }
    slot = (slot + (1u << t.gbits)) & t.mask;
  }
  return -1;
}

// Home slot of a key.
__device__ __forceinline__ uint32_t table_home(const Table &t, hkey_t key) {
  return (t.packed ? hash_key32(static_cast<uint32_t>(key), t.gbits) : hash_key(key, t.gbits)) & t.mask;
}

// Value of key, or -1 when absent (values are row indices / positions, never negative): the walk from `slot`,
// `probe` slots into it.
__device__ __forceinline__ int32_t table_find_from(const Table &t, hkey_t key, uint32_t slot, uint32_t probe) {
  if (t.packed) {
    const unsigned long long *slots = reinterpret_cast<const unsigned long long *>(t.keys);
    const uint32_t k32 = static_cast<uint32_t>(key);
    for (; probe <= t.max_probe; ++probe) {
      const unsigned long long v = slots[slot];
      if (static_cast<uint32_t>(v >> 32) == k32 && v != kEmptySlot) return static_cast<int32_t>(v);
      if (v == kEmptySlot) return -1;
      slot = (slot + (1u << t.gbits)) & t.mask;
    }
    return -1;
  }
  for (; probe <= t.max_probe; ++probe) {
    const hkey_t k = t.keys[slot];
    if (k == key) return t.vals[slot];
    if (k == -1LL) return -1;
    slot = (slot + (1u << t.gbits)) & t.mask;
  }
  return -1;
}

__device__ __forceinline__ int32_t table_find(const Table &t, hkey_t key) {
  return table_find_from(t, key, table_home(t, key), 0u);
}

// Reads one index row (batch, coords...) into canonical 4-d form.
__device__ __forceinline__ void read_row(const int32_t *indices, int i, int ndim, int &b,
                                         int (&c)[4]) {
  if (ndim == 3) {
    const int4 v = reinterpret_cast<const int4 *>(indices)[i];
    b = v.x;
    c[0] = 0;
    c[1] = v.y;
    c[2] = v.z;
    c[3] = v.w;
  } else {
    const int32_t *row = indices + static_cast<size_t>(i) * (ndim + 1);
    b = row[0];
    const int lead = 4 - ndim;
#pragma unroll
    for (int d = 0; d < 4; ++d) c[d] = (d < lead) ? 0 : row[1 + d - lead];
  }
}

__device__ __forceinline__ hkey_t layout_key(int b, const int (&c)[4], const int (&dims)[4]) {
  hkey_t v = b;
#pragma unroll
  for (int d = 0; d < 4; ++d) v = v * dims[d] + c[d];
  return v;
}

__device__ __forceinline__ void decode_offset(int k, const int (&ksize)[4], int (&r)[4]) {
#pragma unroll
  for (int d = 3; d >= 0; --d) {
    r[d] = k % ksize[d];
    k /= ksize[d];
  }
}

__device__ __forceinline__ bool in_range(const int (&c)[4], const int (&dims)[4]) {
  bool ok = true;
#pragma unroll
  for (int d = 0; d < 4; ++d) ok = ok && c[d] >= 0 && c[d] < dims[d];
  return ok;
}

// ------------------------------------------------------------ range fills
// Every "memset" of a rulebook build in ONE launch: on a host-bound pipeline (a single scene per
// step) a rulebook is a dozen launches of ~6 us of host time each, and hipMemsetAsync costs a
// launch like any kernel.  Ranges are 4-byte aligned multiples of 4 bytes; the 16-byte aligned
// middle of each goes out as dwordx4 stores.
constexpr int kMaxFills = 8;
struct FillJobs {
  uint32_t *ptr[kMaxFills];
  unsigned long long words[kMaxFills];
  uint32_t value[kMaxFills];
  int n;
};

__global__ void __launch_bounds__(kBlock)
fill_ranges_kernel(FillJobs jobs) {
  const unsigned long long t = static_cast<unsigned long long>(blockIdx.x) * kBlock + threadIdx.x;
  const unsigned long long T = static_cast<unsigned long long>(gridDim.x) * kBlock;
  for (int j = 0; j < jobs.n; ++j) {
    uint32_t *p = jobs.ptr[j];
    const unsigned long long w = jobs.words[j];
    const uint32_t v = jobs.value[j];
    unsigned long long head = (4 - ((reinterpret_cast<uintptr_t>(p) >> 2) & 3)) & 3;
    if (head > w) head = w;
    const unsigned long long body = (w - head) >> 2, tail = (w - head) & 3;
    if (t < head) p[t] = v;
    uint4 *q = reinterpret_cast<uint4 *>(p + head);
    const uint4 vv = make_uint4(v, v, v, v);
    for (unsigned long long i = t; i < body; i += T) q[i] = vv;
    if (t < tail) p[head + 4 * body + t] = v;
  }
}

struct FillList {
  FillJobs jobs;
  FillList() { jobs.n = 0; }
  // adjacent ranges with the same value merge (tables carved from one buffer: one range)
  void add(void *ptr, size_t bytes, uint32_t value) {
    if (!ptr || bytes == 0) return;
    uint32_t *p = static_cast<uint32_t *>(ptr);
    for (int j = 0; j < jobs.n; ++j) {
      if (jobs.value[j] != value) continue;
      if (jobs.ptr[j] + jobs.words[j] == p) { jobs.words[j] += bytes / 4; return; }
      if (p + bytes / 4 == jobs.ptr[j]) { jobs.ptr[j] = p; jobs.words[j] += bytes / 4; return; }
    }
    jobs.ptr[jobs.n] = p;
    jobs.words[jobs.n] = bytes / 4;
    jobs.value[jobs.n] = value;
    ++jobs.n;
  }
  hipError_t launch(hipStream_t s) const {
    if (jobs.n == 0) return hipSuccess;
    unsigned long long most = 0;
    for (int j = 0; j < jobs.n; ++j) most = jobs.words[j] > most ? jobs.words[j] : most;
    // 16 words (four dwordx4) per thread of the longest range, at most 2048 workgroups
    unsigned long long blocks = (most + 16ull * kBlock - 1) / (16ull * kBlock);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(fill_ranges_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, s, jobs);
    return hipGetLastError();
  }
};

void table_fill(FillList &f, const Table &t);

// ---------------------------------------------------------------- SubM

__global__ void __launch_bounds__(kBlock)
subm_insert_kernel(const int32_t *__restrict__ indices, int n, Geom g, Table t,
                   int32_t *__restrict__ slot_of, uint32_t *__restrict__ mask_zero = nullptr,
                   int words = 0, int32_t *__restrict__ fill_fwd = nullptr, int32_t *__restrict__ fill_bwd = nullptr,
                   uint32_t *__restrict__ occupied = nullptr) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (mask_zero)      // the probe kernel ORs bits into the masks: clear them here (no fill launch)
    for (int w = 0; w < words; ++w) mask_zero[static_cast<size_t>(i) * words + w] = 0u;
  // the half of the tables that only receives scattered mirror entries starts as -1: written here, by the row's
  // own thread (coalesced along the rows), instead of by the fill launch -- whose job shrinks to the hash table
  if (fill_fwd)
    for (int k = 0; k < g.kv / 2; ++k) fill_fwd[static_cast<size_t>(k) * n + i] = -1;
  if (fill_bwd)
    for (int k = g.kv / 2 + 1; k < g.kv; ++k) fill_bwd[static_cast<size_t>(k) * n + i] = -1;
  int b, c[4];
  read_row(indices, i, g.ndim, b, c);
  // rows with a batch index outside [0, batch) ("deleted" points, docs/USAGE.md:150)
  // never match a neighbour query in the CPU path either; do not hash them.
  int slot = -1;
  if (b >= 0 && b < g.batch && in_range(c, g.in_dims))
    slot = table_insert_min(t, layout_key(b, c, g.in_dims), i);
  if (slot_of) slot_of[i] = slot;
  // one occupancy bit per slot for subm_probe5_kernel (set again by a duplicate coordinate: idempotent)
  if (occupied && slot >= 0) atomicOr(&occupied[slot >> 5], 1u << (slot & 31));
}

// One thread per (voxel, offset k < kv/2): every probe chain is independent and there are only
// kv/2 probes per voxel -- a hit at offset k from row o to row v is also the pair (kv-1-k) from
// v to o (the mirror symmetry the CPU path uses, indices.py:1685-1696), written as a scattered
// 4-byte store.  The tables are pre-filled with -1 and the masks with 0 (only hits write); the
// mask bits are OR-ed in with atomicOr (commutative: the result does not depend on the order).
// Duplicate coordinates: lookups only ever return the FIRST row of a coordinate
// (unordered_map::insert keeps it, indices.py:1672), so a later duplicate never appears as the
// found side; such a row keeps only its k > centre half, which it probes itself here.
__global__ void __launch_bounds__(kBlock)
subm_probe3_kernel(const int32_t *__restrict__ indices, int n, Geom g, Table t,
                   const int32_t *__restrict__ slot_of, int32_t *__restrict__ pair_fwd,
                   int32_t *__restrict__ pair_bwd, uint32_t *__restrict__ mask, int words,
                   int32_t *__restrict__ native) {
  const int o = blockIdx.x * kBlock + threadIdx.x;
  const int kv = g.kv, center = kv / 2;
  int k = blockIdx.y;                         // 0 .. kv/2 (the last one is the identity offset)
  if (o >= n) return;
  auto set = [&](int kk, int row, int val) __attribute__((always_inline)) {
    pair_fwd[static_cast<size_t>(kk) * n + row] = val;
    if (pair_bwd) pair_bwd[static_cast<size_t>(kv - 1 - kk) * n + row] = val;
    atomicOr(&mask[static_cast<size_t>(row) * words + (kk >> 5)], 1u << (kk & 31));
  };
  if (k == center) {
    set(center, o, o);
    if (native) {                            // identity lists of ConvAlgo.Native (indices.py:1678-1682)
      native[static_cast<size_t>(center) * n + o] = o;
      native[static_cast<size_t>(kv + center) * n + o] = o;
    }
    return;
  }
  const int self = slot_of[o];
  if (self < 0) return;
  const bool first = table_val(t, self) == o;
  if (!first) k = kv - 1 - k;                 // a duplicate row only owns its k > centre half
  int b, c[4], r[4], q[4];
  read_row(indices, o, g.ndim, b, c);
  decode_offset(k, g.ksize, r);
#pragma unroll
  for (int d = 0; d < 4; ++d) q[d] = c[d] - g.padding[d] + r[d] * g.dilation[d];
  if (!in_range(q, g.in_dims)) return;
  const int v = table_find(t, layout_key(b, q, g.in_dims));
  if (v < 0) return;
  set(k, o, v);
  if (first) set(kv - 1 - k, v, o);
}

// Third form of the probe pass: as subm_probe3_kernel, but every thread probes an offset ABOVE the
// centre (k' = kv-1-k), i.e. in the CPU loop's own orientation -- row o is the INPUT row i of list
// L = kv-1-k' and the hit is the output row (indices.py:1685-1696).  The direct entry
// pair_fwd[k'][o] then is the thread's own (coalesced, written whether hit or miss: that half of
// the table needs no -1 pre-fill), only the mirror entry pair_fwd[L][found] = o is scattered, and the
// hits of a block ARE the entries of list L that fall into the block's 256-voxel group: the block
// leaves their count for subm_lists_kernel (no count / scan launches).  A row that is not the first
// of its coordinate still owns its k' entries but writes no mirror entry (lookups return the first
// row only), so the first-row test is needed on hits only.
__global__ void __launch_bounds__(kBlock)
subm_probe4_kernel(const int32_t *__restrict__ indices, int n, Geom g, Table t,
                   const int32_t *__restrict__ slot_of, int32_t *__restrict__ pair_fwd,
                   int32_t *__restrict__ pair_bwd, uint32_t *__restrict__ mask, int words,
                   int32_t *__restrict__ groupcount, int ngroups, int mask_pass = 0) {
  // mask_pass: the masks come from a pass over the finished table instead of one atomicOr per entry
  __shared__ int lds_wave[kBlock / 64];
  const int o = blockIdx.x * kBlock + threadIdx.x;
  const int kv = g.kv, center = kv / 2;
  const int list = blockIdx.y;                // 0 .. kv/2 - 1, or kv/2 = the identity offset
  const int k = kv - 1 - list;                // probed offset (> centre), or the centre itself
  auto set = [&](int kk, int row, int val) __attribute__((always_inline)) {
    pair_fwd[static_cast<size_t>(kk) * n + row] = val;
    if (pair_bwd) pair_bwd[static_cast<size_t>(kv - 1 - kk) * n + row] = val;
  };
  if (list == center) {
    if (o < n) {
      set(center, o, o);
      if (!mask_pass) atomicOr(&mask[static_cast<size_t>(o) * words + (center >> 5)], 1u << (center & 31));
    }
    return;
  }
  int v = -1;
  int b = -1, c[4] = {0, 0, 0, 0};
  if (o < n) {
    read_row(indices, o, g.ndim, b, c);
    if (b >= 0 && b < g.batch && in_range(c, g.in_dims)) {
      int r[4], q[4];
      decode_offset(k, g.ksize, r);
#pragma unroll
      for (int d = 0; d < 4; ++d) q[d] = c[d] - g.padding[d] + r[d] * g.dilation[d];
      if (in_range(q, g.in_dims)) v = table_find(t, layout_key(b, q, g.in_dims));
    }
    set(k, o, v);                             // own entry, hit or miss
    if (v >= 0) {
      if (!mask_pass) atomicOr(&mask[static_cast<size_t>(o) * words + (k >> 5)], 1u << (k & 31));
      const int self = slot_of[o];
      if (self >= 0 && table_val(t, self) == o) {       // first row of its coordinate: mirror entry
        set(list, v, o);
        if (!mask_pass) atomicOr(&mask[static_cast<size_t>(v) * words + (list >> 5)], 1u << (list & 31));
      }
    }
  }
  if (groupcount) {
    const unsigned long long bal = __ballot(v >= 0);
    if ((threadIdx.x & 63) == 0) lds_wave[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
      int sum = 0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) sum += lds_wave[w];
      groupcount[static_cast<size_t>(list) * ngroups + blockIdx.x] = sum;
    }
  }
}

// Fifth form of the probe pass: an OCCUPANCY BIT per table slot, staged in LDS, answers most probes.
// A SubM probe asks for a neighbour that, on a sparse scene, is almost never there (config 2: 97 % misses), and
// with linear probing and no deletions a key whose HOME slot is empty was never inserted.  The insert kernel sets
// one bit per occupied slot (cap / 8 bytes: 64 KB at 100 k voxels and 4 N slots); a workgroup of 1024 threads copies
// the bits into LDS once, takes 256 voxels x all their upper-half offsets (thread = voxel x one of four offset
// groups), tests every home slot there, and only goes to the table -- random 8-byte reads, the operation the fourth
// form was rate-bound on (84 G/s device-wide, tools/probes/atomic_probe.hip) -- where the bit is set: the table's
// load factor (0.19-0.38) + the real hits.  The first table word of a thread's offsets is requested in one
// straight-line batch (an inactive lane reads slot 0); walks past the home slot are rare and serial.  Offsets are
// decoded once per workgroup (LDS).  Same entries, same masks, same group counts as subm_probe4_kernel.
constexpr int kP5Chunk = 4, kP5Groups = 4, kP5Threads = kBlock * kP5Groups;
__global__ void __launch_bounds__(kP5Threads, 8)   // two workgroups per CU (64 KB of bits each)
subm_probe5_kernel(const int32_t *__restrict__ indices, int n, Geom g, Table t,
                   const uint32_t *__restrict__ occupied, int fwords, const int32_t *__restrict__ slot_of,
                   int32_t *__restrict__ pair_fwd, int32_t *__restrict__ pair_bwd, uint32_t *__restrict__ mask,
                   int words, int32_t *__restrict__ groupcount, int ngroups, int mask_pass) {
  // [fwords] occupancy bits | [half] int4 coordinate steps | [half] key steps | [half][4] hit counts
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_occ[];
  const int tid = threadIdx.x;
  const int kv = g.kv, center = kv / 2, half = kv / 2;
  int4 *lds_delta = reinterpret_cast<int4 *>(lds_occ + fwords);
  hkey_t *lds_dkey = reinterpret_cast<hkey_t *>(lds_delta + half);
  int *lds_cnt = reinterpret_cast<int *>(lds_dkey + half);
  for (int i = tid; i < fwords / 4; i += kP5Threads)
    reinterpret_cast<uint4 *>(lds_occ)[i] = reinterpret_cast<const uint4 *>(occupied)[i];
  for (int l = tid; l < half; l += kP5Threads) {         // neighbour of list l: offset k = kv - 1 - l (> centre)
    int r[4], dq[4];
    decode_offset(kv - 1 - l, g.ksize, r);
    hkey_t dk = 0;                                       // the key is linear in the coordinates: key(c + dq) = key(c) + dk
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      dq[d] = r[d] * g.dilation[d] - g.padding[d];
      dk = dk * g.in_dims[d] + dq[d];
    }
    lds_delta[l] = make_int4(dq[0], dq[1], dq[2], dq[3]);
    lds_dkey[l] = dk;
  }
  const int o = blockIdx.x * kBlock + (tid & (kBlock - 1));
  const int grp = tid / kBlock;                          // lists grp, grp + 4, ... (uniform per wave)
  int b = -1, c[4] = {0, 0, 0, 0};
  bool valid = false;
  if (o < n) {
    read_row(indices, o, g.ndim, b, c);
    valid = b >= 0 && b < g.batch && in_range(c, g.in_dims);
  }
  __syncthreads();
  auto set = [&](int kk, int row, int val) __attribute__((always_inline)) {
    pair_fwd[static_cast<size_t>(kk) * n + row] = val;
    if (pair_bwd) pair_bwd[static_cast<size_t>(kv - 1 - kk) * n + row] = val;
  };
  const unsigned long long *slots = reinterpret_cast<const unsigned long long *>(t.keys);   // packed slots / wide keys
  const hkey_t key0 = layout_key(b, c, g.in_dims);
  int first = -1;                               // row o is the first of its coordinate: looked up at its first hit
  for (int l0 = grp; l0 < half; l0 += kP5Chunk * kP5Groups) {
    hkey_t key[kP5Chunk];
    uint32_t home[kP5Chunk];
    unsigned long long cur[kP5Chunk];
    bool act[kP5Chunk];
#pragma unroll
    for (int j = 0; j < kP5Chunk; ++j) {
      const int lj = l0 + j * kP5Groups;
      const int l = lj < half ? lj : half - 1;
      const int4 dq = lds_delta[l];
      const int q[4] = {c[0] + dq.x, c[1] + dq.y, c[2] + dq.z, c[3] + dq.w};
      key[j] = key0 + lds_dkey[l];
      home[j] = table_home(t, key[j]);
      bool a = valid && lj < half && in_range(q, g.in_dims);
      if (fwords) a = a && ((lds_occ[home[j] >> 5] >> (home[j] & 31)) & 1u);
      act[j] = a;
      cur[j] = slots[a ? home[j] : 0u];
    }
#pragma unroll
    for (int j = 0; j < kP5Chunk; ++j) {
      const int l = l0 + j * kP5Groups;
      const bool live = l < half;                 // (uniform per wave)
      const int k = kv - 1 - l;
      int v = -1;
      if (act[j]) {
        const bool hit = t.packed ? (static_cast<uint32_t>(cur[j] >> 32) == static_cast<uint32_t>(key[j]) &&
                                     cur[j] != kEmptySlot)
                                  : cur[j] == static_cast<unsigned long long>(key[j]);
        if (hit) v = t.packed ? static_cast<int32_t>(static_cast<uint32_t>(cur[j])) : t.vals[home[j]];
        else if (cur[j] != kEmptySlot)
          v = table_find_from(t, key[j], (home[j] + (1u << t.gbits)) & t.mask, 1u);
      }
      if (live && o < n) {
        set(k, o, v);                             // own entry, hit or miss
        if (v >= 0) {
          if (!mask_pass) atomicOr(&mask[static_cast<size_t>(o) * words + (k >> 5)], 1u << (k & 31));
          if (first < 0) {
            const int self = slot_of[o];
            first = (self >= 0 && table_val(t, self) == o) ? 1 : 0;
          }
          if (first) {                            // first row of its coordinate: mirror entry
            set(l, v, o);
            if (!mask_pass) atomicOr(&mask[static_cast<size_t>(v) * words + (l >> 5)], 1u << (l & 31));
          }
        }
      }
      if (groupcount && live) {
        const unsigned long long bal = __ballot(v >= 0);
        if ((tid & 63) == 0) lds_cnt[l * 4 + ((tid >> 6) & 3)] = __popcll(bal);
      }
    }
  }
  if (grp == 0 && o < n) {
    set(center, o, o);
    if (!mask_pass) atomicOr(&mask[static_cast<size_t>(o) * words + (center >> 5)], 1u << (center & 31));
  }
  if (groupcount) {                               // hits per (list, 256-voxel group), as the fourth form leaves them
    __syncthreads();
    for (int l = tid; l < half; l += kP5Threads)
      groupcount[static_cast<size_t>(l) * ngroups + blockIdx.x] =
          lds_cnt[l * 4] + lds_cnt[l * 4 + 1] + lds_cnt[l * 4 + 2] + lds_cnt[l * 4 + 3];
  }
}

// ------------------------------------------------- block-level primitives

// Exclusive rank of this thread among the threads of the block with pred set,
// plus the block total.  wave64 ballot + mbcnt; wave totals through LDS.
__device__ __forceinline__ int block_rank(bool pred, int &total, int *lds_wave /*[4]*/) {
  const unsigned long long bal = __ballot(pred);
  const int lane_rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(bal >> 32),
                            __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(bal), 0u));
  const int wave = threadIdx.x >> 6;
  __syncthreads();  // protect lds_wave reuse across calls
  if ((threadIdx.x & 63) == 0) lds_wave[wave] = __popcll(bal);
  __syncthreads();
  int prefix = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    const int c = lds_wave[w];
    if (w < wave) prefix += c;
    total += c;
  }
  return prefix + lane_rank;
}

// seq-wise exclusive scan of `cnt` (length len per sequence), one block per
// sequence; totals[seq] receives the sequence sum.
constexpr int kScanPer = 32;          // items per thread of the one-pass form
__global__ void __launch_bounds__(kBlock)
scan_kernel(const int32_t *__restrict__ cnt, int32_t *__restrict__ off, int len,
            int32_t *__restrict__ totals) {
  __shared__ int lds_wave[kBlock / 64];
  const int seq = blockIdx.x;
  const int32_t *c = cnt + static_cast<size_t>(seq) * len;
  int32_t *o = off + static_cast<size_t>(seq) * len;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (len <= kBlock * kScanPer) {
    // one pass: every thread owns `per` consecutive items (all loads in flight together), one block
    // scan of the thread sums -- the loop below pays a load -> barrier -> store round per 256 items
    const int per = (len + kBlock - 1) / kBlock, base = threadIdx.x * per;
    int v[kScanPer];
    int sum = 0;
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) {
      v[e] = (e < per && base + e < len) ? c[base + e] : 0;
      sum += v[e];
    }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int u = __shfl_up(incl, d, 64);
      if (lane >= d) incl += u;
    }
    if (lane == 63) lds_wave[wave] = incl;
    __syncthreads();
    int prefix = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
      const int x = lds_wave[w];
      if (w < wave) prefix += x;
      total += x;
    }
    int run = prefix + incl - sum;
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) {
      if (e < per && base + e < len) o[base + e] = run;
      run += v[e];
    }
    if (threadIdx.x == 0 && totals) totals[seq] = total;
    return;
  }
  int carry = 0;
  for (int base = 0; base < len; base += kBlock) {
    const int idx = base + threadIdx.x;
    const int v = idx < len ? c[idx] : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int u = __shfl_up(incl, d, 64);
      if (lane >= d) incl += u;
    }
    __syncthreads();
    if (lane == 63) lds_wave[wave] = incl;
    __syncthreads();
    int prefix = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
      const int s = lds_wave[w];
      if (w < wave) prefix += s;
      total += s;
    }
    if (idx < len) o[idx] = carry + prefix + incl - v;
    carry += total;
  }
  if (threadIdx.x == 0 && totals) totals[seq] = carry;
}

// ------------------------------------------- Native-list compaction (a4/a5)

// mode 0 (SubM): list k in [0, kv/2) is the set {(in=e, out=row[e])} with
//   row = pair_fwd[kv-1-k] (== pair_bwd[k]); the mirror list kv-1-k gets the
//   roles swapped (indices.py:1692-1696).
// mode 1 (conv): list k in [0, kv) from row = pair_bwd[k] (indices.py:1767-1768).
__device__ __forceinline__ const int32_t *list_row(const int32_t *table, int mode, int list,
                                                    int kv, int n) {
  const int row = mode == 0 ? kv - 1 - list : list;
  return table + static_cast<size_t>(row) * n;
}

__global__ void __launch_bounds__(kBlock)
compact_count_kernel(const int32_t *__restrict__ table, int mode, int kv, int n, int nblk,
                     int32_t *__restrict__ blockcount) {
  __shared__ int lds_wave[kBlock / 64];
  const int list = blockIdx.y, blk = blockIdx.x;
  const int32_t *row = list_row(table, mode, list, kv, n);
  const int begin = blk * kItems;
  int cnt = 0;
#pragma unroll
  for (int it = 0; it < kItems / kBlock; ++it) {
    const int e = begin + it * kBlock + threadIdx.x;
    const bool pred = e < n && row[e] >= 0;
    cnt += __popcll(__ballot(pred));
  }
  if ((threadIdx.x & 63) == 0) lds_wave[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < kBlock / 64; ++w) s += lds_wave[w];
    blockcount[static_cast<size_t>(list) * nblk + blk] = s;
  }
}

__global__ void __launch_bounds__(kBlock)
compact_scatter_kernel(const int32_t *__restrict__ table, int mode, int kv, int n, int nblk,
                       const int32_t *__restrict__ blockoff, int32_t *__restrict__ native) {
  __shared__ int lds_wave[kBlock / 64];
  const int list = blockIdx.y, blk = blockIdx.x;
  const int32_t *row = list_row(table, mode, list, kv, n);
  const int begin = blk * kItems;
  int running = blockoff[static_cast<size_t>(list) * nblk + blk];
  const size_t plane = static_cast<size_t>(kv) * n;  // native[1] offset
  int32_t *in_k = native + static_cast<size_t>(list) * n;
  int32_t *out_k = native + plane + static_cast<size_t>(list) * n;
  int32_t *in_m = native + static_cast<size_t>(kv - 1 - list) * n;
  int32_t *out_m = native + plane + static_cast<size_t>(kv - 1 - list) * n;
  for (int it = 0; it < kItems / kBlock; ++it) {
    const int e = begin + it * kBlock + threadIdx.x;
    const int v = e < n ? row[e] : -1;
    int total;
    const int rank = block_rank(v >= 0, total, lds_wave);
    if (v >= 0) {
      const int j = running + rank;
      in_k[j] = e;
      out_k[j] = v;
      if (mode == 0) {
        in_m[j] = v;
        out_m[j] = e;
      }
    }
    running += total;
  }
}

__global__ void __launch_bounds__(kBlock)
subm_center_list_kernel(int32_t *__restrict__ native, int kv, int n) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const size_t c = static_cast<size_t>(kv / 2) * n + i;
  native[c] = i;
  native[static_cast<size_t>(kv) * n + c] = i;
}


// ConvAlgo.Native lists of a SubM rulebook from the finished table, in the CPU loop's order
// (ascending input row inside a list, indices.py:1685-1696): blockIdx.y = list L < kv/2 (and its
// mirror kv-1-L with the roles swapped), or kv/2 = the identity list.  The block's output offset
// is the sum of the 256-voxel hit counts the probe kernel left (no separate scan launch); block 0
// of a list also writes num_per_loc[L].  Positions past a list's length are set to -1 here
// (ops.py:191-193 starts from a -1 filled tensor), so the caller's buffer needs no pre-fill.
__global__ void __launch_bounds__(kBlock)
subm_lists_kernel(const int32_t *__restrict__ pair_fwd, int kv, int n, int nblk256,
                  const int32_t *__restrict__ blockcount, int32_t *__restrict__ native,
                  int32_t *__restrict__ num_per_loc, int num_len, int conv = 0) {
  // conv != 0: regular / transposed convolution -- `pair_fwd` is then pair_bwd [kv, n_in], list k is
  // read off its row k (indices.py:1767-1768), there is no mirror list and no identity list
  __shared__ int lds_wave[kBlock / 64];
  __shared__ int lds_red[2][kBlock / 64];
  const int list = blockIdx.y, blk = blockIdx.x;
  const int begin = blk * kItems;
  const size_t plane = static_cast<size_t>(kv) * n;
  if (!conv && list == kv / 2) {               // identity lists (indices.py:1678-1682)
    // counts exist for k < kv/2 only (indices.py:1685,1692); the rest of num_per_loc reads 0
    if (blk == 0)
      for (int i = kv / 2 + threadIdx.x; i < num_len; i += kBlock) num_per_loc[i] = 0;
    if (!native) return;
    for (int it = 0; it < kItems / kBlock; ++it) {
      const int e = begin + it * kBlock + threadIdx.x;
      if (e < n) {
        native[static_cast<size_t>(list) * n + e] = e;
        native[plane + static_cast<size_t>(list) * n + e] = e;
      }
    }
    return;
  }
  // the block's table entries are requested first, ahead of the count prefix (two independent latencies)
  const int32_t *row = pair_fwd + static_cast<size_t>(conv ? list : kv - 1 - list) * n;
  int vals[kItems / kBlock];
#pragma unroll
  for (int it = 0; it < kItems / kBlock; ++it) {
    const int e = begin + it * kBlock + threadIdx.x;
    vals[it] = (native && e < n) ? row[e] : -1;
  }
  // prefix of the hit counts before this block's first 256-voxel group, and the list total
  const int32_t *cnt = blockcount + static_cast<size_t>(list) * nblk256;
  const int first_group = blk * (kItems / kBlock);
  int before = 0, all = 0;
  for (int i = threadIdx.x; i < nblk256; i += kBlock) {
    const int v = cnt[i];
    all += v;
    if (i < first_group) before += v;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    before += __shfl_xor(before, d, 64);
    all += __shfl_xor(all, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    lds_red[0][threadIdx.x >> 6] = before;
    lds_red[1][threadIdx.x >> 6] = all;
  }
  __syncthreads();
  before = all = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    before += lds_red[0][w];
    all += lds_red[1][w];
  }
  if (blk == 0 && threadIdx.x == 0 && num_per_loc) num_per_loc[list] = all;
  if (!native) return;
  int32_t *in_k = native + static_cast<size_t>(list) * n;
  int32_t *out_k = native + plane + static_cast<size_t>(list) * n;
  int32_t *in_m = native + static_cast<size_t>(kv - 1 - list) * n;
  int32_t *out_m = native + plane + static_cast<size_t>(kv - 1 - list) * n;
  int running = before;
#pragma unroll
  for (int it = 0; it < kItems / kBlock; ++it) {
    const int e = begin + it * kBlock + threadIdx.x;
    const int v = vals[it];
    int total;
    const int rank = block_rank(v >= 0, total, lds_wave);
    if (v >= 0) {
      const int j = running + rank;
      in_k[j] = e;
      out_k[j] = v;
      if (!conv) {
        in_m[j] = v;
        out_m[j] = e;
      }
    }
    running += total;
    if (e < n && e >= all) {                   // tail of the list: this block's own position range
      in_k[e] = -1;
      out_k[e] = -1;
      if (!conv) {
        in_m[e] = -1;
        out_m[e] = -1;
      }
    }
  }
}

// ------------------------------------------------ regular / transposed conv

// Output coordinate for (input row, offset k); false if the pair does not exist.
// Regular: query_npq (indices.py:174-203), transposed: query_nhw_out (:249-269).
__device__ __forceinline__ bool conv_out_coord(const Geom &g, const int (&c)[4],
                                               const int (&r)[4], int transposed,
                                               int (&q)[4]) {
  bool ok = true;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    if (transposed) {
      q[d] = c[d] * g.stride[d] - g.padding[d] + r[d] * g.dilation[d];
    } else {
      const int h = c[d] + g.padding[d] - r[d] * g.dilation[d];
      q[d] = h / g.stride[d];  // C++ truncation, like the reference
      ok = ok && (h % g.stride[d]) == 0;
    }
    ok = ok && q[d] >= 0 && q[d] < g.out_dims[d];
  }
  return ok;
}

// stage 1: hash every candidate output key with value = min first-seen position
// (k * n + i); remember the slot so later passes do not re-probe.
__global__ void __launch_bounds__(kBlock)
conv_stage1_kernel(const int32_t *__restrict__ indices, int n, Geom g, int transposed,
                   Table t, int32_t *__restrict__ slot_of, int32_t *__restrict__ overflow) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const int k = blockIdx.y;
  if (i >= n) return;
  int b, c[4], r[4], q[4];
  read_row(indices, i, g.ndim, b, c);
  decode_offset(k, g.ksize, r);
  const size_t pos = static_cast<size_t>(k) * n + i;
  int slot = -1;
  if (b >= 0 && b < g.batch && conv_out_coord(g, c, r, transposed, q)) {
    slot = table_insert_min<true>(t, layout_key(b, q, g.out_dims), static_cast<int32_t>(pos));
    if (slot < 0) *overflow = 1;          // table full: reported by spx_conv_rulebook_count
  }
  slot_of[pos] = slot;
}

__device__ __forceinline__ bool is_first_seen(const int32_t *slot_of, const Table &t,
                                              size_t pos, bool inb, int &slot) {
  slot = inb ? slot_of[pos] : -1;
  return slot >= 0 && table_val(t, slot) == static_cast<int32_t>(pos);
}

__global__ void __launch_bounds__(kBlock)
conv_count_first_kernel(const int32_t *__restrict__ slot_of, Table t,
                        int n, int nblk, int32_t *__restrict__ blockcount) {
  __shared__ int lds_wave[kBlock / 64];
  const int k = blockIdx.y, blk = blockIdx.x;
  const int begin = blk * kItems;
  int cnt = 0;
#pragma unroll
  for (int it = 0; it < kItems / kBlock; ++it) {
    const int e = begin + it * kBlock + threadIdx.x;
    int slot;
    const bool pred = is_first_seen(slot_of, t, static_cast<size_t>(k) * n + e, e < n, slot);
    cnt += __popcll(__ballot(pred));
  }
  if ((threadIdx.x & 63) == 0) lds_wave[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < kBlock / 64; ++w) s += lds_wave[w];
    blockcount[static_cast<size_t>(k) * nblk + blk] = s;
  }
}

// Numbers outputs in first-seen order and writes their coordinates.
__global__ void __launch_bounds__(kBlock)
conv_assign_kernel(const int32_t *__restrict__ indices, int n, Geom g, int transposed,
                   const int32_t *__restrict__ slot_of, Table t,
                   int nblk, const int32_t *__restrict__ blockoff,
                   int32_t *__restrict__ slot_out, int32_t *__restrict__ out_indices, int n_cap) {
  __shared__ int lds_wave[kBlock / 64];
  const int k = blockIdx.y, blk = blockIdx.x;
  const int begin = blk * kItems;
  int running = blockoff[static_cast<size_t>(k) * nblk + blk];
  int r[4];
  decode_offset(k, g.ksize, r);
  const int lead = 4 - g.ndim;
  // all of the block's loads first (8 independent slot -> table chains in flight): with the loads
  // inside the ranking loop every iteration paid two dependent memory latencies between barriers
  int slots[kItems / kBlock];
  bool firsts[kItems / kBlock];
#pragma unroll
  for (int it = 0; it < kItems / kBlock; ++it) {
    const int e = begin + it * kBlock + threadIdx.x;
    slots[it] = e < n ? slot_of[static_cast<size_t>(k) * n + e] : -1;
  }
#pragma unroll
  for (int it = 0; it < kItems / kBlock; ++it) {
    const int e = begin + it * kBlock + threadIdx.x;
    firsts[it] = slots[it] >= 0 &&
                 table_val(t, slots[it]) == static_cast<int32_t>(static_cast<size_t>(k) * n + e);
  }
#pragma unroll
  for (int it = 0; it < kItems / kBlock; ++it) {
    const int e = begin + it * kBlock + threadIdx.x;
    const int slot = slots[it];
    const bool first = firsts[it];
    int total;
    const int rank = block_rank(first, total, lds_wave);
    if (first) {
      // outputs beyond the caller's bound (num_out_act_bound, ops.py:263-266) are dropped: no
      // coordinates, no pairs
      const int oid = running + rank;
      slot_out[slot] = oid < n_cap ? oid : -1;
      if (oid < n_cap) {
        int b, c[4], q[4];
        read_row(indices, e, g.ndim, b, c);
        conv_out_coord(g, c, r, transposed, q);
        int32_t *dst = out_indices + static_cast<size_t>(oid) * (g.ndim + 1);
        dst[0] = b;
        for (int d = lead; d < 4; ++d) dst[1 + d - lead] = q[d];
      }
    }
    running += total;
  }
}

__global__ void __launch_bounds__(kBlock)
conv_stage2_kernel(const int32_t *__restrict__ slot_of, const int32_t *__restrict__ slot_out,
                   int n, int n_out, int32_t *__restrict__ pair_fwd,
                   int32_t *__restrict__ pair_bwd, int32_t *__restrict__ groupcount = nullptr) {
  // groupcount[k][block]: pairs of offset k among this block's 256 input rows = the entries of
  // ConvAlgo.Native list k that fall into the group (subm_lists_kernel turns them into offsets)
  __shared__ int lds_wave[kBlock / 64];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const int k = blockIdx.y;
  int oid = -1;
  if (i < n) {
    const size_t pos = static_cast<size_t>(k) * n + i;
    const int slot = slot_of[pos];
    if (slot >= 0) {
      oid = slot_out[slot];                          // -1: an output beyond the caller's bound
      if (oid >= 0) pair_fwd[static_cast<size_t>(k) * n_out + oid] = i;
    }
    pair_bwd[pos] = oid;
  }
  if (groupcount) {
    const unsigned long long bal = __ballot(oid >= 0);
    if ((threadIdx.x & 63) == 0) lds_wave[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
      int sum = 0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) sum += lds_wave[w];
      groupcount[static_cast<size_t>(k) * gridDim.x + blockIdx.x] = sum;
    }
  }
}


// ------------------------------------------ regular conv, third generation: compact candidates
// A strided convolution pairs an input with FEW of the kv offsets: along one axis only the r with
// (c + p - r d) % s == 0 reach an output, at most ceil(k gcd(d, s) / s) of them (conv_max_out), so
// k = 3 / s = 2 in 3-d has <= 8 candidates per input out of 27 (3.4 on average), k = 2 / s = 2 exactly
// one.  The second-generation passes above launch a thread per (offset, input) and stream kv x N
// arrays five times; 85 % of those threads divide, find no pair and write a -1.  Here one thread owns
// an input row: it derives the valid offsets of each axis as a bit set (no integer division: h / s by
// a float multiply, exact below 2^21), walks their product in ascending k -- the candidate index j --
// and every per-candidate array is [MJ, N] with MJ <= 8.  The first-seen numbering (k-major, then
// input-major, indices.py:1742-1771) comes from a BIT MAP of the first-seen candidates, one row per
// offset: the count of an (offset, 2048-input block) is a popcount, the rank of an entry a prefix
// popcount -- in input order by construction, no ballots, no barriers, every pass elementwise.
constexpr int kMaxCand = 8;       // candidates per input the compact passes are instantiated for
constexpr int kMaxKv3 = 64;       // offsets (bit-map rows held in LDS)

struct CandIter {
  uint32_t vm[4], v[4];
  bool live;
  // valid offsets per axis of input coordinate c (regular conv): bit r of vm[d]
  __device__ __forceinline__ void init(const Geom &g, const int (&c)[4], bool row_ok) {
    live = row_ok;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float inv = 1.0f / static_cast<float>(g.stride[d]);
      uint32_t m = 0;
      for (int r = 0; r < g.ksize[d]; ++r) {
        const int h = c[d] + g.padding[d] - r * g.dilation[d];
        const int q = __float2int_rn(static_cast<float>(h) * inv);
        if (q * g.stride[d] == h && q >= 0 && q < g.out_dims[d]) m |= 1u << r;
      }
      vm[d] = v[d] = m;
      live = live && m != 0;
    }
  }
  // current candidate: offset index k and output coordinate q
  __device__ __forceinline__ int offset(const Geom &g, const int (&c)[4], int (&q)[4]) const {
    int k = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int r = __builtin_ctz(v[d]);
      k = k * g.ksize[d] + r;
      const int h = c[d] + g.padding[d] - r * g.dilation[d];
      q[d] = __float2int_rn(static_cast<float>(h) * (1.0f / static_cast<float>(g.stride[d])));
    }
    return k;
  }
  __device__ __forceinline__ int offset(const Geom &g) const {
    int k = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) k = k * g.ksize[d] + __builtin_ctz(v[d]);
    return k;
  }
  __device__ __forceinline__ void next() {     // odometer over the bit sets, last axis fastest
#pragma unroll
    for (int d = 3; d >= 0; --d) {
      v[d] &= v[d] - 1;
      if (v[d]) return;
      v[d] = vm[d];
    }
    live = false;
  }
};

// pass 1: insert every candidate output key with value = min first-seen position (k * n + i);
// slot_c[j][i] = its slot (entries past an input's candidate count are never read)
template <int MJ>
__global__ void __launch_bounds__(kBlock)
conv3_insert_kernel(const int32_t *__restrict__ indices, int n, Geom g, Table t,
                    int32_t *__restrict__ slot_c, int32_t *__restrict__ overflow) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  int b, c[4];
  read_row(indices, i, g.ndim, b, c);
  CandIter it;
  it.init(g, c, b >= 0 && b < g.batch);
  // blockIdx.y: the thread's share of the row's candidates (j = y mod gridDim.y) -- a small scene
  // does not have enough rows to hide eight dependent atomic round trips per thread
  const int share = blockIdx.y, shares = gridDim.y - 1;
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    if (!it.live) break;
    if ((j & shares) == share) {
      int q[4];
      const int k = it.offset(g, c, q);
      const int slot = table_insert_min<true>(t, layout_key(b, q, g.out_dims), k * n + i);
      if (slot < 0) *overflow = 1;             // table full: reported by spx_conv_rulebook_count
      slot_c[static_cast<size_t>(j) * n + i] = slot;
    }
    it.next();
  }
}

// pass 2: bit map of the first-seen candidates, one row of nblk * 64 words per offset (bit i of row k:
// input i is the first to reach its output through k).  A workgroup collects the bits of its 256 rows
// in LDS -- 32 neighbouring inputs share a word, global atomics on it would serialise inside the wave --
// and ORs its non-zero words (8 per offset) into the map; also the per-input byte of first-seen
// candidate indices that lets pass 3 skip everything else
template <int MJ>
__global__ void __launch_bounds__(kBlock)
conv3_first_kernel(const int32_t *__restrict__ indices, int n, Geom g, Table t,
                   const int32_t *__restrict__ slot_c, int nblk, uint32_t *__restrict__ firstbits,
                   uint8_t *__restrict__ firstflags) {
  __shared__ uint32_t bm[kMaxKv3][kBlock / 32];
  const int i = blockIdx.x * kBlock + threadIdx.x, kv = g.kv;
  for (int w = threadIdx.x; w < kv * (kBlock / 32); w += kBlock) (&bm[0][0])[w] = 0;
  __syncthreads();
  const int share = blockIdx.y, shares = gridDim.y - 1;      // (as conv3_insert_kernel)
  if (i < n) {
    int b, c[4];
    read_row(indices, i, g.ndim, b, c);
    CandIter it;
    it.init(g, c, b >= 0 && b < g.batch);
    int kk[MJ], slot[MJ];
#pragma unroll
    for (int j = 0; j < MJ; ++j) {             // every load of the row first
      kk[j] = -1;
      slot[j] = -1;
      if (it.live) {
        if ((j & shares) == share) {
          kk[j] = it.offset(g);
          slot[j] = slot_c[static_cast<size_t>(j) * n + i];
        }
        it.next();
      }
    }
    int val[MJ];
#pragma unroll
    for (int j = 0; j < MJ; ++j) val[j] = slot[j] >= 0 ? table_val(t, slot[j]) : -1;
    uint32_t flags = 0;
#pragma unroll
    for (int j = 0; j < MJ; ++j)
      if (slot[j] >= 0 && val[j] == kk[j] * n + i) {
        atomicOr(&bm[kk[j]][threadIdx.x >> 5], 1u << (threadIdx.x & 31));
        flags |= 1u << j;
      }
    firstflags[static_cast<size_t>(share) * n + i] = static_cast<uint8_t>(flags);   // one plane per share
  }
  __syncthreads();
  const size_t rowwords = static_cast<size_t>(nblk) * (kItems / 32);
  for (int w = threadIdx.x; w < kv * (kBlock / 32); w += kBlock) {
    const uint32_t word = (&bm[0][0])[w];
    if (word) {
      uint32_t *dst = &firstbits[(w / (kBlock / 32)) * rowwords + blockIdx.x * (kBlock / 32) + (w % (kBlock / 32))];
      if (shares) atomicOr(dst, word); else *dst = word;       // one share: this workgroup owns the word
    }
  }
}

// pass 2b: one wave per (offset, 2048-input block): popcount of the block's 64 bit-map words
// (-> blockcount, for the scan) and their exclusive prefix inside the block (-> wordpre)
__global__ void __launch_bounds__(kBlock)
conv3_count_kernel(const uint32_t *__restrict__ firstbits, int rows, int32_t *__restrict__ wordpre,
                   int32_t *__restrict__ blockcount) {
  const int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const size_t at = static_cast<size_t>(row) * (kItems / 32) + lane;
  const int cnt = __popc(firstbits[at]);
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int u = __shfl_up(incl, d, 64);
    if (lane >= d) incl += u;
  }
  wordpre[at] = incl - cnt;
  if (lane == 63) blockcount[row] = incl;
}

// pass 3: number the first-seen candidates -- blockoff of (offset, block) + first-seen entries of the
// block before this word + set bits below this input's -- and write their coordinates;
// slot_out[slot] = output row (or -1 beyond the caller's bound)
template <int MJ>
__global__ void __launch_bounds__(kBlock)
conv3_assign_kernel(const int32_t *__restrict__ indices, int n, Geom g,
                    const int32_t *__restrict__ slot_c, int nblk,
                    const uint32_t *__restrict__ firstbits, const uint8_t *__restrict__ firstflags,
                    const int32_t *__restrict__ wordpre, const int32_t *__restrict__ blockoff,
                    int32_t *__restrict__ slot_out, int32_t *__restrict__ out_indices, int n_cap) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint32_t flags = firstflags[static_cast<size_t>(blockIdx.y) * n + i];     // (shares as in pass 2)
  if (!flags) return;                          // (most inputs are first for no offset)
  int b, c[4];
  read_row(indices, i, g.ndim, b, c);
  CandIter it;
  it.init(g, c, b >= 0 && b < g.batch);
  const int lead = 4 - g.ndim;
  const size_t rowwords = static_cast<size_t>(nblk) * (kItems / 32);
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    if (!it.live) break;
    if ((flags >> j) & 1u) {
      int q[4];
      const int k = it.offset(g, c, q);
      const size_t at = k * rowwords + (i >> 5);
      const int oid = blockoff[static_cast<size_t>(k) * nblk + i / kItems] + wordpre[at] +
                      __popc(firstbits[at] & ((1u << (i & 31)) - 1u));
      const int slot = slot_c[static_cast<size_t>(j) * n + i];
      // outputs beyond the caller's bound (num_out_act_bound, ops.py:263-266) are dropped
      slot_out[slot] = oid < n_cap ? oid : -1;
      if (oid < n_cap) {
        int32_t *dst = out_indices + static_cast<size_t>(oid) * (g.ndim + 1);
        dst[0] = b;
        for (int d = lead; d < 4; ++d) dst[1 + d - lead] = q[d];
      }
    }
    it.next();
  }
}

// pass 4: both tables, the input-side mask and the per-(offset, 256 rows) pair counts of the Native
// lists.  pair_bwd and mask_bwd are written whole (no -1 pre-fill, no second pass over pair_bwd).
template <int MJ>
__global__ void __launch_bounds__(kBlock)
conv3_pairs_kernel(const int32_t *__restrict__ indices, int n, Geom g,
                   const int32_t *__restrict__ slot_c, const int32_t *__restrict__ slot_out, int n_out,
                   int32_t *__restrict__ pair_fwd, int32_t *__restrict__ pair_bwd,
                   uint32_t *__restrict__ mask_bwd, int words, int32_t *__restrict__ groupcount) {
  __shared__ int lds_cnt[kMaxKv3];
  const int i = blockIdx.x * kBlock + threadIdx.x, kv = g.kv;
  if (groupcount) {
    if (threadIdx.x < kMaxKv3) lds_cnt[threadIdx.x] = 0;
    __syncthreads();
  }
  int kk[MJ], oid[MJ];
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    kk[j] = -1;
    oid[j] = -1;
  }
  if (i < n) {
    int b, c[4];
    read_row(indices, i, g.ndim, b, c);
    CandIter it;
    it.init(g, c, b >= 0 && b < g.batch);
    int slot[MJ];
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      slot[j] = -1;
      if (it.live) {
        kk[j] = it.offset(g);
        slot[j] = slot_c[static_cast<size_t>(j) * n + i];
        it.next();
      }
    }
#pragma unroll
    for (int j = 0; j < MJ; ++j)
      if (slot[j] >= 0) oid[j] = slot_out[slot[j]];    // -1: an output beyond the caller's bound
#pragma unroll
    for (int j = 0; j < MJ; ++j)
      if (oid[j] >= 0) pair_fwd[static_cast<size_t>(kk[j]) * n_out + oid[j]] = i;
  }
  uint32_t mword = 0;
  for (int k = 0; k < kv; ++k) {
    int val = -1;
#pragma unroll
    for (int j = 0; j < MJ; ++j) val = kk[j] == k ? oid[j] : val;
    if (i < n) pair_bwd[static_cast<size_t>(k) * n + i] = val;
    if (val >= 0) mword |= 1u << (k & 31);
    if (mask_bwd && i < n && ((k & 31) == 31 || k == kv - 1)) {
      mask_bwd[static_cast<size_t>(i) * words + (k >> 5)] = mword;
      mword = 0;
    }
    if (groupcount) {
      const unsigned long long bal = __ballot(val >= 0);
      if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&lds_cnt[k], __popcll(bal));
    }
  }
  if (groupcount) {
    __syncthreads();
    if (threadIdx.x < kv) groupcount[static_cast<size_t>(threadIdx.x) * gridDim.x + blockIdx.x] = lds_cnt[threadIdx.x];
  }
}

// ------------------------------------------ regular conv, fourth generation: outputs numbered by KEY RANK
// The passes above number the outputs in the CPU reference's first-seen order (indices.py:1742-1771), which takes a
// hash table, a first-seen bit map and a rank per (offset, input).  The reference's GPU path has no such order: its
// outputs come out of a sort + unique of the linear coordinate keys (all.py:1533-1552) or out of a hash table in slot
// order (indices.py:1380-1425).  This generation produces the SORTED order, without a sort and without a hash table:
//   * the RANK MAP of a level: one {occupancy bits, prefix} pair per 32 consecutive linear keys (batch-major, x
//     fastest), and behind the words the occupied cells before each block of 2048 words.
//     mark: one plain BYTE store per candidate into a byte-per-cell scratch map (idempotent: no atomics);
//     prefix: bytes -> bits, popcount scan of the words inside a block; scan_kernel: the blocks' offsets;
//     row of key = block offset + prefix + popcount(bits below): ONE 8-byte load, no probing, no first-seen resolution;
//   * pairs: per input, the rank of every candidate's key -> both pair tables, the input-side mask, list counts, and
//     the coordinates of the outputs it reaches (every input of an output stores the same values);
//   * the map stays with the level: a SubM layer behind the strided layer looks its neighbours up in it
//     (subm_rank_rows_kernel / subm_rank_probe_kernel below) -- no table fill, no insert, no slot walks -- and its
//     rows, being in key order, put x-neighbours in adjacent rows (what the gather-GEMMs of the level gain:
//     tools/order_probe.py).
// Memory: the map (batch x grid cells) / 4 bytes (47 M cells of a 21 x 800 x 704 x 4 level: 11.8 MB) + one byte per
// cell of scratch during the build; key spaces beyond 2^31 cells keep the hash builder.
constexpr int kRankWords = 2048;      // words (65536 cells) per prefix block
constexpr int kRankPer = kRankWords / kBlock;

template <int MJ>
__global__ void __launch_bounds__(kBlock)
conv4_mark_kernel(const int32_t *__restrict__ indices, int n, Geom g, uint8_t *__restrict__ occupied) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  int b, c[4];
  read_row(indices, i, g.ndim, b, c);
  CandIter it;
  it.init(g, c, b >= 0 && b < g.batch);
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    if (!it.live) break;
    int q[4];
    it.offset(g, c, q);
    // one BYTE per cell, plain stores: idempotent (every writer stores 1), nothing to wait for, no atomic -- a bit map
    // costs an agent-scope atomicOr per candidate (20-25 G/s device-wide against ~80 G/s for stores: 58 -> 20 us at
    // 400 k inputs); conv4_prefix_kernel packs the bytes into the words of the rank map
    occupied[static_cast<unsigned long long>(layout_key(b, q, g.out_dims))] = 1;
    it.next();
  }
}

// the byte map packed into the words of the rank map (cells[w].x), block-local exclusive prefix of their popcounts
// (cells[w].y), block total -> blockcount.  32 bytes (two 16-byte loads) per word; a byte is 0 or 1, so four of them
// become a nibble with one multiply: ((v * 0x01020408) >> 24) & 15.
__device__ __forceinline__ uint32_t pack_flags16(const uint4 &v) {
  auto nib = [](uint32_t d) __attribute__((always_inline)) { return ((d * 0x01020408u) >> 24) & 15u; };
  return nib(v.x) | (nib(v.y) << 4) | (nib(v.z) << 8) | (nib(v.w) << 12);
}

__global__ void __launch_bounds__(kBlock)
conv4_prefix_kernel(const uint4 *__restrict__ occupied, uint2 *__restrict__ cells, unsigned W,
                    int32_t *__restrict__ blockcount) {
  __shared__ int lds_wave[kBlock / 64];
  __shared__ __attribute__((aligned(16))) uint16_t lds_half[2 * kRankWords];
  const unsigned base = blockIdx.x * kRankWords + threadIdx.x * kRankPer;
  // The block's 64 KB of flag bytes in 16-byte pieces, lane-consecutive (a thread reading ITS eight words' 256 bytes put
  // every load instruction on 64 different lines: 27 us for the 59 MB of a 47 M-cell level); a piece becomes 16 bits,
  // the halves of a word meet in LDS
  {
    const size_t piece0 = static_cast<size_t>(blockIdx.x) * (2 * kRankWords);
    const size_t pieces = 2 * static_cast<size_t>(W);
    uint4 v[2 * kRankPer];
#pragma unroll
    for (int j = 0; j < 2 * kRankPer; ++j) {
      const size_t pc = piece0 + static_cast<size_t>(j) * kBlock + threadIdx.x;
      v[j] = pc < pieces ? occupied[pc] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int j = 0; j < 2 * kRankPer; ++j) lds_half[j * kBlock + threadIdx.x] = static_cast<uint16_t>(pack_flags16(v[j]));
  }
  __syncthreads();
  uint32_t bits[kRankPer];
  int cnt[kRankPer], sum = 0;
  {
    const uint4 *w4 = reinterpret_cast<const uint4 *>(lds_half) + threadIdx.x * (kRankPer / 4);
    static_assert(kRankPer == 8, "two 16-byte reads per thread");
    const uint4 a = w4[0], b = w4[1];
    const uint32_t w[kRankPer] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < kRankPer; ++e) {
      bits[e] = base + e < W ? w[e] : 0u;
      cnt[e] = __popc(bits[e]);
      sum += cnt[e];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int u = __shfl_up(incl, d, 64);
    if (lane >= d) incl += u;
  }
  if (lane == 63) lds_wave[wave] = incl;
  __syncthreads();
  int prefix = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    const int x = lds_wave[w];
    if (w < wave) prefix += x;
    total += x;
  }
  int run = prefix + incl - sum;
#pragma unroll
  for (int e = 0; e < kRankPer; ++e) {
    if (base + e < W) cells[base + e] = make_uint2(bits[e], static_cast<uint32_t>(run));
    run += cnt[e];
  }
  if (threadIdx.x == 0) blockcount[blockIdx.x] = total;
}

// row of a key: occupied cells before its 65536-cell block + before its word inside the block + below it in the word
__device__ __forceinline__ int rank_of(const uint2 *__restrict__ cells, const int32_t *__restrict__ blockoff,
                                       unsigned long long key) {
  const uint2 cell = cells[key >> 5];
  const uint32_t bit = 1u << (key & 31);
  return (cell.x & bit) ? blockoff[key >> 16] + static_cast<int>(cell.y) + __popc(cell.x & (bit - 1u)) : -1;
}

// The rank map of a level whose rows ALREADY are in ascending, unique key order (level 1 of a backbone when the data
// loader sorts its voxels: spconv_amd.pytorch.utils.sort_voxels_by_coordinate): row = rank, so the word of a key holds
// {bits of the level's rows that fall into it, index of the first of them} and every block offset is zero -- no marks,
// no prefix pass, no scan, no atomics.  The first row of a word writes it (it looks ahead over the <= 31 rows that can
// share the word).  Rows that break the contract (a key <= its predecessor's, a live row behind a dead one) raise
// `violation`; dead rows (batch -1: static shapes) must trail.
__global__ void __launch_bounds__(kBlock)
rankmap_from_sorted_kernel(const int32_t *__restrict__ indices, int n, Geom g, uint2 *__restrict__ cells,
                           int32_t *__restrict__ violation) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  auto key_of = [&](int row, bool &ok) __attribute__((always_inline)) {
    int b, c[4];
    read_row(indices, row, g.ndim, b, c);
    ok = b >= 0 && b < g.batch && in_range(c, g.in_dims);
    return ok ? static_cast<unsigned long long>(layout_key(b, c, g.in_dims)) : 0ull;
  };
  bool ok;
  const unsigned long long key = key_of(i, ok);
  if (!ok) return;
  bool first = true;
  if (i > 0) {
    bool pok;
    const unsigned long long prev = key_of(i - 1, pok);
    if (!pok || prev >= key) {
      if (violation) atomicOr(violation, 1);
    }
    first = !pok || (prev >> 5) != (key >> 5);
  }
  if (!first) return;
  uint32_t bits = 1u << (key & 31);
  for (int j = i + 1; j < n && j < i + 32; ++j) {
    bool jok;
    const unsigned long long kj = key_of(j, jok);
    if (!jok || (kj >> 5) != (key >> 5)) break;
    bits |= 1u << (kj & 31);
  }
  cells[key >> 5] = make_uint2(bits, static_cast<uint32_t>(i));
}

// both tables, the input-side mask and the pair counts of the Native lists (as conv3_pairs_kernel; the output row
// of a candidate is its key's rank)
template <int MJ>
__global__ void __launch_bounds__(kBlock)
conv4_pairs_kernel(const int32_t *__restrict__ indices, int n, Geom g, const uint2 *__restrict__ cells,
                   const int32_t *__restrict__ blockoff, int n_out, int32_t *__restrict__ out_indices,
                   int32_t *__restrict__ pair_fwd, int32_t *__restrict__ pair_bwd,
                   uint32_t *__restrict__ mask_bwd, int words, int32_t *__restrict__ groupcount,
                   int32_t *__restrict__ live_out) {
  __shared__ int lds_cnt[kMaxKv3];
  const int i = blockIdx.x * kBlock + threadIdx.x, kv = g.kv;
  // static-shape form: the number of live output rows (outputs found, at most the bound) for the layers behind
  if (live_out && i == 0) live_out[2] = live_out[0] < n_out ? live_out[0] : n_out;
  if (groupcount) {
    if (threadIdx.x < kMaxKv3) lds_cnt[threadIdx.x] = 0;
    __syncthreads();
  }
  int kk[MJ], oid[MJ];
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    kk[j] = -1;
    oid[j] = -1;
  }
  if (i < n) {
    int b, c[4];
    read_row(indices, i, g.ndim, b, c);
    CandIter it;
    it.init(g, c, b >= 0 && b < g.batch);
    unsigned long long key[MJ];
    int qx[MJ][4];
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      key[j] = 0;
      if (it.live) {
        kk[j] = it.offset(g, c, qx[j]);
        key[j] = static_cast<unsigned long long>(layout_key(b, qx[j], g.out_dims));
        it.next();
      }
    }
#pragma unroll
    for (int j = 0; j < MJ; ++j) {                 // (every map load of the row in flight together)
      if (kk[j] >= 0) {
        const int r = rank_of(cells, blockoff, key[j]);
        oid[j] = r < n_out ? r : -1;               // an output beyond the caller's bound
      }
    }
    const int lead = 4 - g.ndim;
#pragma unroll
    for (int j = 0; j < MJ; ++j)
      if (oid[j] >= 0) {
        pair_fwd[static_cast<size_t>(kk[j]) * n_out + oid[j]] = i;
        // the output's coordinates, by every input that reaches it (the same values: idempotent stores instead of a
        // pass over the whole map that decodes the set bits)
        int32_t *dst = out_indices + static_cast<size_t>(oid[j]) * (g.ndim + 1);
        dst[0] = b;
        for (int d = lead; d < 4; ++d) dst[1 + d - lead] = qx[j][d];
      }
  }
  uint32_t mword = 0;
  for (int k = 0; k < kv; ++k) {
    int val = -1;
#pragma unroll
    for (int j = 0; j < MJ; ++j) val = kk[j] == k ? oid[j] : val;
    if (i < n) pair_bwd[static_cast<size_t>(k) * n + i] = val;
    if (val >= 0) mword |= 1u << (k & 31);
    if (mask_bwd && i < n && ((k & 31) == 31 || k == kv - 1)) {
      mask_bwd[static_cast<size_t>(i) * words + (k >> 5)] = mword;
      mword = 0;
    }
    if (groupcount) {
      const unsigned long long bal = __ballot(val >= 0);
      if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&lds_cnt[k], __popcll(bal));
    }
  }
  if (groupcount) {
    __syncthreads();
    if (threadIdx.x < kv) groupcount[static_cast<size_t>(threadIdx.x) * gridDim.x + blockIdx.x] = lds_cnt[threadIdx.x];
  }
}

// SubM probe pass over the rank map of a level whose rows are in key order (row = rank of its key): as
// subm_probe4_kernel -- same entries, masks and group counts --, the neighbour looked up by rank_of instead of a
// hash walk; every row is the first (and only) row of its coordinate.
__global__ void __launch_bounds__(kBlock)
subm_rank_probe_kernel(const int32_t *__restrict__ indices, int n, Geom g, const uint2 *__restrict__ cells,
                       const int32_t *__restrict__ blockoff, int32_t *__restrict__ pair_fwd, int32_t *__restrict__ pair_bwd,
                       uint32_t *__restrict__ mask, int words, int32_t *__restrict__ groupcount, int ngroups,
                       int mask_pass) {
  __shared__ int lds_wave[kBlock / 64];
  const int o = blockIdx.x * kBlock + threadIdx.x;
  const int kv = g.kv, center = kv / 2;
  const int list = blockIdx.y;                // 0 .. kv/2 - 1, or kv/2 = the identity offset
  const int k = kv - 1 - list;                // probed offset (> centre), or the centre itself
  auto set = [&](int kk, int row, int val) __attribute__((always_inline)) {
    pair_fwd[static_cast<size_t>(kk) * n + row] = val;
    if (pair_bwd) pair_bwd[static_cast<size_t>(kv - 1 - kk) * n + row] = val;
  };
  if (list == center) {
    if (o < n) {
      set(center, o, o);
      if (!mask_pass) atomicOr(&mask[static_cast<size_t>(o) * words + (center >> 5)], 1u << (center & 31));
    }
    return;
  }
  int v = -1;
  if (o < n) {
    int b, c[4];
    read_row(indices, o, g.ndim, b, c);
    if (b >= 0 && b < g.batch && in_range(c, g.in_dims)) {
      int r[4], q[4];
      decode_offset(k, g.ksize, r);
#pragma unroll
      for (int d = 0; d < 4; ++d) q[d] = c[d] - g.padding[d] + r[d] * g.dilation[d];
      if (in_range(q, g.in_dims)) {
        v = rank_of(cells, blockoff, static_cast<unsigned long long>(layout_key(b, q, g.in_dims)));
        if (v >= n) v = -1;                   // (an output the producing layer's bound dropped)
      }
      set(k, o, v);                           // own entry, hit or miss
      if (v >= 0) {
        if (!mask_pass) atomicOr(&mask[static_cast<size_t>(o) * words + (k >> 5)], 1u << (k & 31));
        set(list, v, o);                      // mirror entry
        if (!mask_pass) atomicOr(&mask[static_cast<size_t>(v) * words + (list >> 5)], 1u << (list & 31));
      }
    } else {
      set(k, o, -1);                          // a dead row (static shapes): no neighbours, and nobody's neighbour
    }
  }
  if (groupcount) {
    const unsigned long long bal = __ballot(v >= 0);
    if ((threadIdx.x & 63) == 0) lds_wave[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
      int sum = 0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) sum += lds_wave[w];
      groupcount[static_cast<size_t>(list) * ngroups + blockIdx.x] = sum;
    }
  }
}

// SubM build over a rank map, ROW-OWNED: one thread per row looks up ALL its neighbours itself (a lookup is one 8-byte
// load, and in key order the three x-offsets of a (dz, dy) pair share a word) and writes its whole column of the
// table(s) and its mask word -- no mirror scatter, no atomicOr, no -1 pre-fill, one launch.  Same tables, masks and
// list counts as subm_rank_probe_kernel / subm_probe4_kernel (tests/test_gpu_sorted.py: torch.equal to the hash build).
constexpr int kRowsChunk = 9;
__global__ void __launch_bounds__(kBlock)
subm_rank_rows_kernel(const int32_t *__restrict__ indices, int n, Geom g, const uint2 *__restrict__ cells,
                      const int32_t *__restrict__ blockoff, int32_t *__restrict__ pair_fwd,
                      int32_t *__restrict__ pair_bwd, uint32_t *__restrict__ mask, int words,
                      int32_t *__restrict__ groupcount, int ngroups) {
  // [kv] coordinate steps | [kv] key steps | [kv] hits of the block per offset
  extern __shared__ __attribute__((aligned(16))) int4 lds_delta[];
  const int kv = g.kv, center = kv / 2;
  hkey_t *lds_dkey = reinterpret_cast<hkey_t *>(lds_delta + kv);
  int *lds_cnt = reinterpret_cast<int *>(lds_dkey + kv);
  for (int k = threadIdx.x; k < kv; k += kBlock) {
    int r[4], dq[4];
    decode_offset(k, g.ksize, r);
    hkey_t dk = 0;                                         // the key is linear in the coordinates
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      dq[d] = r[d] * g.dilation[d] - g.padding[d];
      dk = dk * g.in_dims[d] + dq[d];
    }
    lds_delta[k] = make_int4(dq[0], dq[1], dq[2], dq[3]);
    lds_dkey[k] = dk;
    lds_cnt[k] = 0;
  }
  const int o = blockIdx.x * kBlock + threadIdx.x;
  int b = -1, c[4] = {0, 0, 0, 0};
  bool valid = false;
  if (o < n) {
    read_row(indices, o, g.ndim, b, c);
    valid = b >= 0 && b < g.batch && in_range(c, g.in_dims);
  }
  __syncthreads();
  const hkey_t key0 = layout_key(b, c, g.in_dims);
  uint32_t mword = 0;
  for (int k0 = 0; k0 < kv; k0 += kRowsChunk) {
    uint2 cell[kRowsChunk];
    hkey_t key[kRowsChunk];
    bool act[kRowsChunk];
#pragma unroll
    for (int j = 0; j < kRowsChunk; ++j) {                 // the chunk's loads in one straight-line batch
      const int k = k0 + j < kv ? k0 + j : kv - 1;
      const int4 dq = lds_delta[k];
      const int q[4] = {c[0] + dq.x, c[1] + dq.y, c[2] + dq.z, c[3] + dq.w};
      key[j] = key0 + lds_dkey[k];
      act[j] = valid && k0 + j < kv && k != center && in_range(q, g.in_dims);
      cell[j] = cells[act[j] ? static_cast<unsigned long long>(key[j]) >> 5 : 0ull];
    }
#pragma unroll
    for (int j = 0; j < kRowsChunk; ++j) {
      const int k = k0 + j;
      const bool live = k < kv;                            // (uniform; no break: the loop must unroll -- cell[] in registers)
      int v = -1;
      if (act[j]) {
        const uint32_t bit = 1u << (static_cast<unsigned long long>(key[j]) & 31);
        if (cell[j].x & bit)
          v = blockoff[static_cast<unsigned long long>(key[j]) >> 16] + static_cast<int>(cell[j].y) +
              __popc(cell[j].x & (bit - 1u));
        if (v >= n) v = -1;                                // (an output the producing layer's bound dropped)
      }
      if (k == center && o < n) v = o;                     // (dead rows of a static level too: as the other forms)
      if (live && o < n) {
        pair_fwd[static_cast<size_t>(k) * n + o] = v;
        if (pair_bwd) pair_bwd[static_cast<size_t>(kv - 1 - k) * n + o] = v;
      }
      if (live && v >= 0) mword |= 1u << (k & 31);
      if (live && o < n && ((k & 31) == 31 || k == kv - 1)) {
        mask[static_cast<size_t>(o) * words + (k >> 5)] = mword;
        mword = 0;
      }
      if (groupcount && live && k > center) {              // list kv - 1 - k: the pairs found through offset k
        const unsigned long long bal = __ballot(v >= 0);
        if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&lds_cnt[k], __popcll(bal));
      }
    }
  }
  if (groupcount) {
    __syncthreads();
    for (int l = threadIdx.x; l < center; l += kBlock)
      groupcount[static_cast<size_t>(l) * ngroups + blockIdx.x] = lds_cnt[kv - 1 - l];
  }
}

// mask[row][w] bit k = (table[k][row] >= 0)  (indices.py:652-676)
__global__ void __launch_bounds__(kBlock)
mask_from_table_kernel(const int32_t *__restrict__ table, int kv, int n, int words,
                       uint32_t *__restrict__ mask) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  uint32_t mcur = 0;
  for (int k = 0; k < kv; ++k) {
    if (table[static_cast<size_t>(k) * n + i] >= 0) mcur |= 1u << (k & 31);
    if ((k & 31) == 31 || k == kv - 1) {
      mask[static_cast<size_t>(i) * words + (k >> 5)] = mcur;
      mcur = 0;
    }
  }
}

// Both masks of a regular-conv rulebook in one launch: rows [0, n_a) of table a, then rows of b.
__global__ void __launch_bounds__(kBlock)
mask_from_tables_kernel(const int32_t *__restrict__ ta, int n_a, uint32_t *__restrict__ ma,
                        const int32_t *__restrict__ tb, int n_b, uint32_t *__restrict__ mb,
                        int kv, int words) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  const int32_t *table = ta;
  uint32_t *mask = ma;
  int n = n_a;
  if (i >= n_a) {
    i -= n_a;
    table = tb;
    mask = mb;
    n = n_b;
  }
  if (i >= n) return;
  uint32_t mcur = 0;
  for (int k = 0; k < kv; ++k) {
    if (table[static_cast<size_t>(k) * n + i] >= 0) mcur |= 1u << (k & 31);
    if ((k & 31) == 31 || k == kv - 1) {
      mask[static_cast<size_t>(i) * words + (k >> 5)] = mcur;
      mcur = 0;
    }
  }
}

// ------------------------------------------------------- mask argsort (a9)
// Stable LSD radix sort of (mask word, row) with 8-bit digits built from the
// same count -> scan -> scatter primitives.  words == 1 only (kv <= 32).
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;

__global__ void __launch_bounds__(kBlock)
radix_count_kernel(const uint32_t *__restrict__ keys, int n, int shift, int nblk,
                   int32_t *__restrict__ hist /*[kRadix][nblk]*/) {
  __shared__ int lds_hist[kRadix];
  for (int d = threadIdx.x; d < kRadix; d += kBlock) lds_hist[d] = 0;
  __syncthreads();
  const int begin = blockIdx.x * kItems;
  for (int it = 0; it < kItems / kBlock; ++it) {
    const int e = begin + it * kBlock + threadIdx.x;
    if (e < n) atomicAdd(&lds_hist[(keys[e] >> shift) & (kRadix - 1)], 1);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < kRadix; d += kBlock)
    hist[static_cast<size_t>(d) * nblk + blockIdx.x] = lds_hist[d];
}

// Stable scatter: processes the block's entries in order, 256 at a time; the
// rank of an entry among equal digits inside the 256-entry tile comes from a
// per-digit match over wave ballots.
__global__ void __launch_bounds__(kBlock)
radix_scatter_kernel(const uint32_t *__restrict__ keys_in, const int32_t *__restrict__ vals_in,
                     int n, int shift, int nblk, const int32_t *__restrict__ hist_off,
                     uint32_t *__restrict__ keys_out, int32_t *__restrict__ vals_out) {
  __shared__ int lds_base[kRadix];                 // running output offset per digit
  __shared__ int lds_cnt[kBlock / 64][kRadix];     // per-wave digit counts of this tile
  for (int d = threadIdx.x; d < kRadix; d += kBlock)
    lds_base[d] = hist_off[static_cast<size_t>(d) * nblk + blockIdx.x];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int begin = blockIdx.x * kItems;
  for (int it = 0; it < kItems / kBlock; ++it) {
    for (int d = threadIdx.x; d < kRadix; d += kBlock)
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) lds_cnt[w][d] = 0;
    __syncthreads();
    const int e = begin + it * kBlock + threadIdx.x;
    const bool valid = e < n;
    const uint32_t key = valid ? keys_in[e] : 0u;
    const int val = (valid && vals_in) ? vals_in[e] : e;
    const int digit = valid ? static_cast<int>((key >> shift) & (kRadix - 1)) : -1;
    // lanes of this wave holding the same digit (bitwise match over 8 ballots)
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < kRadixBits; ++bit) {
      const unsigned long long bal = __ballot((digit >> bit) & 1);
      same &= ((digit >> bit) & 1) ? bal : ~bal;
    }
    const int rank_in_wave = __popcll(same & ((1ull << lane) - 1ull));
    if (valid && rank_in_wave == 0) lds_cnt[wave][digit] = __popcll(same);
    __syncthreads();
    if (valid) {
      int prior = 0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w)
        if (w < wave) prior += lds_cnt[w][digit];
      const int dst = lds_base[digit] + prior + rank_in_wave;
      keys_out[dst] = key;
      vals_out[dst] = val;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < kRadix; d += kBlock) {
      int s = 0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) s += lds_cnt[w][d];
      lds_base[d] += s;
    }
    __syncthreads();
  }
}

// Dense table from ConvAlgo.Native lists (for callers that only hold the lists):
// table[k][dst_j] = src_j for j < count(k).  count(): SubM mirror rule (ops.py:962-968).
__global__ void __launch_bounds__(kBlock)
native_to_table_kernel(const int32_t *__restrict__ native, const int32_t *__restrict__ num,
                       int kv, int n_in, int n_dst, int subm, int inverse,
                       int32_t *__restrict__ table) {
  const int k = blockIdx.y;
  const int j = blockIdx.x * kBlock + threadIdx.x;
  int cnt;
  if (!subm) cnt = num[k];
  else if (k == kv / 2) cnt = n_in;
  else cnt = k < kv / 2 ? num[k] : num[kv - 1 - k];
  if (cnt > n_in) cnt = n_in;  // convops.py:1592 clamp
  if (j >= cnt) return;
  const size_t plane = static_cast<size_t>(kv) * n_in;
  const size_t e = static_cast<size_t>(k) * n_in + j;
  const int in_idx = native[e], out_idx = native[plane + e];
  const int src = inverse ? out_idx : in_idx, dst = inverse ? in_idx : out_idx;
  table[static_cast<size_t>(k) * n_dst + dst] = src;
}

uint32_t table_capacity(size_t entries) {
  size_t cap = 256;
  while (cap < 2 * entries) cap <<= 1;
  return static_cast<uint32_t>(cap);
}

// True when every key of a (batch, dims[0..3]) layout is below 0xFFFFFFFF: the table then keeps
// key and value in one 64-bit slot.
bool keys_fit_u32(long long batch, const int *dims, int ndims) {
  unsigned long long v = batch > 0 ? static_cast<unsigned long long>(batch) : 1ull;
  for (int d = 0; d < ndims; ++d) {
    const unsigned long long e = dims[d] > 0 ? static_cast<unsigned long long>(dims[d]) : 1ull;
    if (v > 0xFFFFFFFFull / e) return false;
    v *= e;
  }
  return v <= 0xFFFFFFFFull;   // largest key is v - 1 <= 0xFFFFFFFE
}

// Places a table of `cap` slots at `mem` (room for the wide form: 12 bytes per slot).
void table_place(Table &t, hkey_t *keys, int32_t *vals, uint32_t cap, bool packed, int gbits = 0) {
  t.keys = keys;
  t.vals = packed ? reinterpret_cast<int32_t *>(keys) : vals;
  t.mask = cap - 1;
  t.packed = packed ? 1 : 0;
  t.gbits = gbits;
  t.max_probe = t.mask;
}

// The same storage as a smaller table (capacity a power of two below the placed one).
void table_shrink(Table &t, uint32_t cap) {
  if (!t.packed) t.vals = reinterpret_cast<int32_t *>(t.keys + cap);
  t.mask = cap - 1;
  t.max_probe = t.mask < kMaxProbeShrunk ? t.mask : kMaxProbeShrunk;
}

// The table's bytes as a 0xFF range of a FillList (see table_clear).
void table_fill(FillList &f, const Table &t) {
  const size_t cap = static_cast<size_t>(t.mask) + 1;
  const size_t bytes = t.packed ? cap * sizeof(unsigned long long)
                                : static_cast<size_t>(reinterpret_cast<char *>(t.vals + cap) -
                                                      reinterpret_cast<char *>(t.keys));
  f.add(t.keys, bytes, 0xFFFFFFFFu);
}

// Empties the table: every byte 0xFF (keys -1, values 0xFFFFFFFF, packed slots ~0).
hipError_t table_clear(const Table &t, hipStream_t s) {
  const size_t cap = static_cast<size_t>(t.mask) + 1;
  const size_t bytes = t.packed ? cap * sizeof(unsigned long long)
                                : static_cast<size_t>(reinterpret_cast<char *>(t.vals + cap) -
                                                      reinterpret_cast<char *>(t.keys));
  return hipMemsetAsync(t.keys, 0xFF, bytes, s);
}

int gcd_int(int a, int b) {
  while (b) {
    const int t = a % b;
    a = b;
    b = t;
  }
  return a < 0 ? -a : a;
}

size_t conv_max_out(int n_in, int ndim, const int *ksize, const int *stride, const int *dilation,
                    int transposed) {
  // Upper bound of distinct outputs.  The reference's SpconvOps.get_handcrafted_max_act_out
  // (all.py:1557-1578) uses N * prod(ceil(k/s)), which ignores dilation: along one axis an input
  // reaches the outputs o with o*s = c + p - r*d, and r*d mod s repeats with period s / gcd(d, s),
  // so up to ceil(k * gcd(d, s) / s) offsets r hit a multiple of s (k = 3, s = 2, d = 2: all 3,
  // not 2).  Transposed: kv * N (ops.py:569-570).
  size_t kv = 1, m = 1;
  for (int i = 0; i < ndim; ++i) {
    kv *= ksize[i];
    const int g = gcd_int(dilation ? dilation[i] : 1, stride[i]);
    size_t per = (static_cast<size_t>(ksize[i]) * g + stride[i] - 1) / stride[i];
    if (per > static_cast<size_t>(ksize[i])) per = ksize[i];
    m *= per;
  }
  if (transposed || m > kv) m = kv;
  return m * static_cast<size_t>(n_in);
}

struct ConvWs {
  Table t;
  int32_t *slot_out, *slot_of, *blockcount, *blockoff, *d_nout, *groupcount;
  uint32_t *firstbits;                 // compact passes: [kv, nblk, 64] first-seen bit map ...
  int32_t *wordpre;                    // ... first-seen entries of the block before each of its words
  uint8_t *firstflags;                 // ... and per input the candidate indices that are first-seen
  int nblk;
  size_t bytes;
};

// Candidates per input the compact passes (conv3_*) run with -- 1, 2, 4 or 8 -- or 0 when the
// problem takes the thread-per-(offset, input) passes: transposed convolution, stride 1 (every offset
// is a candidate), more than 8 candidates, kv > 64, coordinates beyond the float-exact range.
int conv3_cands(int ndim, const int *in_shape, const int *ksize, const int *stride, const int *padding,
                const int *dilation, int transposed) {
  if (option_int("SPX_CONV_V", 3) < 3 || transposed) return 0;       // (2: the passes above, for A/B runs)
  int kv = 1;
  for (int i = 0; i < ndim; ++i) {
    kv *= ksize[i];
    const long long reach = static_cast<long long>(ksize[i]) * (dilation ? dilation[i] : 1);
    if (ksize[i] > 32 || in_shape[i] + static_cast<long long>(padding[i]) >= (1 << 21) || reach >= (1 << 21)) return 0;
  }
  const size_t m = conv_max_out(1, ndim, ksize, stride, dilation, 0);
  if (kv > kMaxKv3 || m > kMaxCand || 2 * m > static_cast<size_t>(kv)) return 0;
  return m <= 1 ? 1 : (m <= 2 ? 2 : (m <= 4 ? 4 : 8));
}

// Shares per input row of the compact passes that are bound by dependent memory round trips (grid.y): a
// power of two <= the candidate count, enough for ~1.5 M threads.
int conv3_shares(int n_in, int mj) {
  int s = 1;
  while (s < mj && static_cast<long long>(n_in) * s < 1500000) s <<= 1;
  return s;
}

#define SPX_CONV3_LAUNCH(kernel, mj, ...)                                     \
  do {                                                                        \
    if ((mj) == 1) hipLaunchKernelGGL(kernel<1>, __VA_ARGS__);                \
    else if ((mj) == 2) hipLaunchKernelGGL(kernel<2>, __VA_ARGS__);           \
    else if ((mj) == 4) hipLaunchKernelGGL(kernel<4>, __VA_ARGS__);           \
    else hipLaunchKernelGGL(kernel<8>, __VA_ARGS__);                          \
  } while (0)

ConvWs carve_conv_ws(void *ws, int n_in, int ndim, const int *ksize, const int *stride,
                     const int *dilation, int transposed, bool packed = false) {
  int kv = 1;
  for (int i = 0; i < ndim; ++i) kv *= ksize[i];
  uint32_t cap = table_capacity(conv_max_out(n_in, ndim, ksize, stride, dilation, transposed));
  // tests only (spx_set_option): a table smaller than the bound, to exercise the overflow report
  const int test_cap = option_int("SPX_TEST_CONV_TABLE_CAP", 0);
  if (test_cap > 0 && static_cast<uint32_t>(test_cap) < cap) cap = table_capacity(static_cast<size_t>(test_cap) / 2);
  ConvWs w;
  w.nblk = div_up(n_in > 0 ? n_in : 1, kItems);
  Carver cv(ws);
  {
    hkey_t *keys = cv.take<hkey_t>(cap);
    table_place(w.t, keys, cv.take<int32_t>(cap), cap, packed);
  }
  w.slot_out = cv.take<int32_t>(cap);
  w.slot_of = cv.take<int32_t>(static_cast<size_t>(kv) * (n_in > 0 ? n_in : 1));
  w.blockcount = cv.take<int32_t>(static_cast<size_t>(kv) * w.nblk);
  w.blockoff = cv.take<int32_t>(static_cast<size_t>(kv) * w.nblk);
  w.d_nout = cv.take<int32_t>(2);      // [0] number of outputs, [1] hash-table overflow flag
  w.groupcount = cv.take<int32_t>(static_cast<size_t>(kv) * div_up(n_in > 0 ? n_in : 1, kBlock));
  w.firstbits = cv.take<uint32_t>(static_cast<size_t>(kv <= kMaxKv3 ? kv : 0) * w.nblk * (kItems / 32));
  w.wordpre = cv.take<int32_t>(static_cast<size_t>(kv <= kMaxKv3 ? kv : 0) * w.nblk * (kItems / 32));
  w.firstflags = cv.take<uint8_t>(static_cast<size_t>(kMaxCand) * (n_in > 0 ? n_in : 1));
  w.bytes = cv.off;
  return w;
}

int check_geom(int ndim, int n, int kv) {
  SPX_CHECK(ndim >= 1 && ndim <= kMaxNdim, "ndim must be in [1,4], got %d", ndim);
  SPX_CHECK(n >= 0, "negative voxel count %d", n);
  SPX_CHECK(kv >= 1 && static_cast<long long>(kv) * n < 2147483647LL,
            "kernel volume %d x %d voxels overflows int32 positions", kv, n);
  return 0;
}

// Launches the count -> scan -> scatter compaction that builds the Native lists.
int launch_native_lists(const int32_t *table, int mode, int kv, int n, int nlists, int nblk,
                        int32_t *blockcount, int32_t *blockoff, int32_t *native,
                        int32_t *num_per_loc, hipStream_t s) {
  if (n == 0 || nlists == 0) return 0;
  dim3 grid(nblk, nlists);
  hipLaunchKernelGGL(compact_count_kernel, grid, dim3(kBlock), 0, s, table, mode, kv, n, nblk,
                     blockcount);
  hipLaunchKernelGGL(scan_kernel, dim3(nlists), dim3(kBlock), 0, s, blockcount, blockoff, nblk,
                     num_per_loc);
  hipLaunchKernelGGL(compact_scatter_kernel, grid, dim3(kBlock), 0, s, table, mode, kv, n, nblk,
                     blockoff, native);
  SPX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------- point -> voxel
// Voxeliser.  Deterministic and identical to the reference's CPU loop
// (csrc/sparse/pointops.py Point2VoxelCPU::point_to_voxel, lines 135-172 of the class): voxels
// are numbered in first-seen point order, a voxel keeps its first max_points points in point
// order, voxels past max_voxels are dropped.  Same building blocks as the rulebook: hash with
// atomicMin (first point of a voxel), count -> scan -> assign (numbering), stable radix sort by
// voxel id (slot of a point inside its voxel), no order-dependent atomics.
struct P2VGeom {
  int ndim;
  float vsize[4], lo[4];
  int grid[4];
};

__device__ __forceinline__ bool p2v_coor(const float *__restrict__ pt, const P2VGeom &g, int (&c)[4]) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < g.ndim) {
      // zyx order: coordinate j comes from point column ndim-1-j (pointops.py:107,138)
      const float v = floorf((pt[g.ndim - 1 - j] - g.lo[j]) / g.vsize[j]);
      const int ci = static_cast<int>(v);
      ok = ok && !(v < 0.f) && v < static_cast<float>(g.grid[j]);
      c[j] = ci;
    } else {
      c[j] = 0;
    }
  }
  return ok;
}

__global__ void __launch_bounds__(kBlock)
p2v_insert_kernel(const float *__restrict__ pts, int n, int nfeat, P2VGeom g, Table t,
                  int32_t *__restrict__ slot_of) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  int c[4];
  int slot = -1;
  if (p2v_coor(pts + static_cast<size_t>(i) * nfeat, g, c)) {
    hkey_t key = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < g.ndim) key = key * g.grid[j] + c[j];
    slot = table_insert_min(t, key, i);
  }
  slot_of[i] = slot;
}

__global__ void __launch_bounds__(kBlock)
p2v_count_first_kernel(const int32_t *__restrict__ slot_of, Table t, int n,
                       int32_t *__restrict__ blockcount) {
  __shared__ int lds_wave[kBlock / 64];
  const int begin = blockIdx.x * kItems;
  int cnt = 0;
#pragma unroll
  for (int it = 0; it < kItems / kBlock; ++it) {
    const int e = begin + it * kBlock + threadIdx.x;
    const int slot = e < n ? slot_of[e] : -1;
    cnt += __popcll(__ballot(slot >= 0 && table_val(t, slot) == e));
  }
  if ((threadIdx.x & 63) == 0) lds_wave[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int sum = 0;
    for (int w = 0; w < kBlock / 64; ++w) sum += lds_wave[w];
    blockcount[blockIdx.x] = sum;
  }
}

__global__ void __launch_bounds__(kBlock)
p2v_assign_kernel(const float *__restrict__ pts, int n, int nfeat, P2VGeom g,
                  const int32_t *__restrict__ slot_of, Table t,
                  const int32_t *__restrict__ blockoff, int max_voxels,
                  int32_t *__restrict__ slot_vid, int32_t *__restrict__ indices) {
  __shared__ int lds_wave[kBlock / 64];
  const int begin = blockIdx.x * kItems;
  int running = blockoff[blockIdx.x];
  for (int it = 0; it < kItems / kBlock; ++it) {
    const int e = begin + it * kBlock + threadIdx.x;
    const int slot = e < n ? slot_of[e] : -1;
    const bool first = slot >= 0 && table_val(t, slot) == e;
    int total;
    const int rank = block_rank(first, total, lds_wave);
    if (first) {
      const int vid = running + rank;
      if (vid < max_voxels) {
        slot_vid[slot] = vid;
        int c[4];
        p2v_coor(pts + static_cast<size_t>(e) * nfeat, g, c);
        for (int j = 0; j < g.ndim; ++j) indices[static_cast<size_t>(vid) * g.ndim + j] = c[j];
      }
    }
    running += total;
  }
}

__global__ void __launch_bounds__(kBlock)
p2v_point_vid_kernel(const int32_t *__restrict__ slot_of, const int32_t *__restrict__ slot_vid, int n,
                     long long *__restrict__ pc_voxel_id, uint32_t *__restrict__ key32) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int slot = slot_of[i];
  const int vid = slot >= 0 ? slot_vid[slot] : -1;
  pc_voxel_id[i] = vid;
  key32[i] = vid < 0 ? 0xffffffffu : static_cast<uint32_t>(vid);
}

// sorted (voxel id, point) pairs -> slot of the point inside its voxel
__global__ void __launch_bounds__(kBlock)
p2v_segment_kernel(const uint32_t *__restrict__ keys, int n, int32_t *__restrict__ seg_start) {
  const int q = blockIdx.x * kBlock + threadIdx.x;
  if (q >= n) return;
  const uint32_t v = keys[q];
  if (v != 0xffffffffu && (q == 0 || keys[q - 1] != v)) seg_start[v] = q;
}

__global__ void __launch_bounds__(kBlock)
p2v_scatter_kernel(const float *__restrict__ pts, int nfeat, const uint32_t *__restrict__ keys,
                   const int32_t *__restrict__ order, int n, const int32_t *__restrict__ seg_start,
                   int max_points, float *__restrict__ voxels, int32_t *__restrict__ num_per_voxel) {
  const int q = blockIdx.x * kBlock + threadIdx.x;
  if (q >= n) return;
  const uint32_t v = keys[q];
  if (v == 0xffffffffu) return;
  const int rank = q - seg_start[v];
  if (rank < max_points) {
    const float *src = pts + static_cast<size_t>(order[q]) * nfeat;
    float *dst = voxels + (static_cast<size_t>(v) * max_points + rank) * nfeat;
    for (int k = 0; k < nfeat; ++k) dst[k] = src[k];
  }
  if (q == n - 1 || keys[q + 1] != v) num_per_voxel[v] = min(rank + 1, max_points);
}

// empty_mean: slots num..max_points-1 of a voxel receive the mean of its points
__global__ void __launch_bounds__(kBlock)
p2v_mean_kernel(float *__restrict__ voxels, const int32_t *__restrict__ num_per_voxel,
                const int32_t *__restrict__ n_voxels, int max_points, int nfeat) {
  const long long gid = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x;
  const int v = static_cast<int>(gid / nfeat), k = static_cast<int>(gid % nfeat);
  if (v >= *n_voxels) return;
  const int num = num_per_voxel[v];
  if (num <= 0 || num >= max_points) return;
  float *base = voxels + static_cast<size_t>(v) * max_points * nfeat + k;
  float sum = 0.f;
  for (int j = 0; j < num; ++j) sum += base[static_cast<size_t>(j) * nfeat];
  const float mean = sum / static_cast<float>(num);
  for (int j = num; j < max_points; ++j) base[static_cast<size_t>(j) * nfeat] = mean;
}

// The reference's CPU loop AS IT BEHAVES (pointops.py:663-686): `mean_value.clear()` leaves the accumulator's
// contents in place, so voxel v starts from the mean of voxel v - 1:  m_v = (m_{v-1} + sum_j x_j) / num_v, point by
// point in fp32.  A sequential recurrence over the voxels in their (first-seen) order: one thread per feature
// walks all of them.  Opt-in (empty_mean = 2: SPCONV_AMD_REFERENCE_QUIRKS=1), bit-identical to the reference's code
// executed (tests/golden/p2v_ref.npz); milliseconds, not microseconds.
__global__ void p2v_mean_carry_kernel(float *__restrict__ voxels, const int32_t *__restrict__ num_per_voxel,
                                      const int32_t *__restrict__ n_voxels, int max_points, int nfeat) {
  const int k = threadIdx.x;
  if (k >= nfeat) return;
  const int nv = *n_voxels;
  float carry = 0.f;
  for (int v = 0; v < nv; ++v) {
    const int num = num_per_voxel[v];
    if (num <= 0) continue;
    float *base = voxels + static_cast<size_t>(v) * max_points * nfeat + k;
    for (int j = 0; j < num; ++j) carry += base[static_cast<size_t>(j) * nfeat];
    carry /= static_cast<float>(num);
    for (int j = num; j < max_points; ++j) base[static_cast<size_t>(j) * nfeat] = carry;
  }
}

__global__ void p2v_clamp_count_kernel(const int32_t *total, int max_voxels, int32_t *n_voxels) {
  *n_voxels = *total < max_voxels ? *total : max_voxels;
}

struct P2VWs {
  Table t;
  int32_t *slot_of, *slot_vid, *blockcount, *blockoff, *total, *n_voxels, *seg_start, *order, *hist, *hist_off;
  uint32_t *key32, *kA, *kB;
  int32_t *vB;
  int nblk;
  size_t bytes;
};

P2VWs carve_p2v_ws(void *ws, int n, int max_voxels, bool packed = false) {
  const uint32_t cap = table_capacity(n > 0 ? n : 1);
  const size_t np = n > 0 ? n : 1;
  P2VWs w;
  w.nblk = div_up(static_cast<int>(np), kItems);
  Carver cv(ws);
  {
    hkey_t *keys = cv.take<hkey_t>(cap);
    table_place(w.t, keys, cv.take<int32_t>(cap), cap, packed);
  }
  w.slot_vid = cv.take<int32_t>(cap);
  w.slot_of = cv.take<int32_t>(np);
  w.blockcount = cv.take<int32_t>(w.nblk);
  w.blockoff = cv.take<int32_t>(w.nblk);
  w.total = cv.take<int32_t>(1);
  w.n_voxels = cv.take<int32_t>(1);
  w.seg_start = cv.take<int32_t>(max_voxels > 0 ? max_voxels : 1);
  w.order = cv.take<int32_t>(np);
  w.key32 = cv.take<uint32_t>(np);
  w.kA = cv.take<uint32_t>(np);
  w.kB = cv.take<uint32_t>(np);
  w.vB = cv.take<int32_t>(np);
  w.hist = cv.take<int32_t>(static_cast<size_t>(kRadix) * w.nblk);
  w.hist_off = cv.take<int32_t>(static_cast<size_t>(kRadix) * w.nblk);
  w.bytes = cv.off;
  return w;
}

// ------------------------------------------------------------- user hash table
// Fixed-size open-addressing table over caller-owned key / value arrays (SURVEY.md section 8f
// row 4; replaces spconv/csrc/hash/core.py HashTable as used by spconv/pytorch/hash.py).  Keys
// are 32- or 64-bit integers (all-ones = empty), values are opaque 4- or 8-byte items.
// assign_arange / items walk the table in SLOT order (count -> scan -> assign), so their result
// is a pure function of the set of keys -- the reference's GPU table numbers entries in atomic
// arrival order.
template <typename K> struct UKey;
template <> struct UKey<uint32_t> { static __device__ __forceinline__ uint32_t empty() { return 0xffffffffu; } };
template <> struct UKey<unsigned long long> {
  static __device__ __forceinline__ unsigned long long empty() { return ~0ull; }
};

template <typename K>
__device__ __forceinline__ uint32_t user_hash(K k) {
  return hash_key(static_cast<hkey_t>(k), 0);
}

template <typename K>
__device__ __forceinline__ int user_find(const K *keys, int cap, K key, bool insert) {
  uint32_t slot = user_hash(key) % static_cast<uint32_t>(cap);
  for (int probe = 0; probe < cap; ++probe) {
    if (insert) {
      const K prev = atomicCAS(const_cast<K *>(&keys[slot]), UKey<K>::empty(), key);
      if (prev == UKey<K>::empty() || prev == key) return static_cast<int>(slot);
    } else {
      const K cur = keys[slot];
      if (cur == key) return static_cast<int>(slot);
      if (cur == UKey<K>::empty()) return -1;
    }
    slot = slot + 1 == static_cast<uint32_t>(cap) ? 0u : slot + 1;
  }
  return -1;
}

// op 0: insert (values optional), 1: query, 2: insert only where the key exists
template <typename K, typename V>
__global__ void __launch_bounds__(kBlock)
user_hash_kernel(K *__restrict__ tkeys, V *__restrict__ tvals, int cap, const K *__restrict__ keys,
                 V *__restrict__ values, unsigned char *__restrict__ is_empty, int n, int op) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const K key = keys[i];
  if (op == 0) {
    const int slot = user_find(tkeys, cap, key, true);
    if (slot >= 0 && values) tvals[slot] = values[i];
  } else {
    const int slot = user_find(tkeys, cap, key, false);
    if (is_empty) is_empty[i] = slot < 0 ? 1 : 0;
    if (slot >= 0) {
      if (op == 1) values[i] = tvals[slot];
      else tvals[slot] = values[i];
    }
  }
}

template <typename K>
__global__ void __launch_bounds__(kBlock)
user_count_kernel(const K *__restrict__ tkeys, int cap, int32_t *__restrict__ blockcount) {
  __shared__ int lds_wave[kBlock / 64];
  const int begin = blockIdx.x * kItems;
  int cnt = 0;
#pragma unroll
  for (int it = 0; it < kItems / kBlock; ++it) {
    const int e = begin + it * kBlock + threadIdx.x;
    cnt += __popcll(__ballot(e < cap && tkeys[e] != UKey<K>::empty()));
  }
  if ((threadIdx.x & 63) == 0) lds_wave[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int sum = 0;
    for (int w = 0; w < kBlock / 64; ++w) sum += lds_wave[w];
    blockcount[blockIdx.x] = sum;
  }
}

// mode 0: tvals[slot] = rank (assign_arange); mode 1: (keys_out, vals_out)[rank] = entry (items)
template <typename K, typename V>
__global__ void __launch_bounds__(kBlock)
user_walk_kernel(const K *__restrict__ tkeys, V *__restrict__ tvals, int cap,
                 const int32_t *__restrict__ blockoff, int mode, K *__restrict__ keys_out,
                 V *__restrict__ vals_out, int max_out) {
  __shared__ int lds_wave[kBlock / 64];
  const int begin = blockIdx.x * kItems;
  int running = blockoff[blockIdx.x];
  for (int it = 0; it < kItems / kBlock; ++it) {
    const int e = begin + it * kBlock + threadIdx.x;
    const bool used = e < cap && tkeys[e] != UKey<K>::empty();
    int total;
    const int rank = block_rank(used, total, lds_wave);
    if (used) {
      const int r = running + rank;
      if (mode == 0) {
        tvals[e] = static_cast<V>(r);
      } else if (r < max_out) {
        keys_out[r] = tkeys[e];
        vals_out[r] = tvals[e];
      }
    }
    running += total;
  }
}

template <typename T>
__global__ void user_store_count_kernel(const int32_t *total, T *count_out) { *count_out = static_cast<T>(*total); }

template <typename K, typename V>
int user_hash_dispatch2(int what, void *tkeys, void *tvals, int cap, const void *keys, void *values,
                        unsigned char *is_empty, int n, void *keys_out, void *vals_out, int max_out,
                        void *count_out, void *ws, hipStream_t s) {
  K *tk = static_cast<K *>(tkeys);
  V *tv = static_cast<V *>(tvals);
  if (what <= 2) {
    if (n > 0)
      hipLaunchKernelGGL((user_hash_kernel<K, V>), dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, tk, tv, cap,
                         static_cast<const K *>(keys), static_cast<V *>(values), is_empty, n, what);
  } else {
    const int nblk = div_up(cap, kItems);
    Carver cv(ws);
    int32_t *blockcount = cv.take<int32_t>(nblk);
    int32_t *blockoff = cv.take<int32_t>(nblk);
    int32_t *total = cv.take<int32_t>(1);
    hipLaunchKernelGGL((user_count_kernel<K>), dim3(nblk), dim3(kBlock), 0, s, tk, cap, blockcount);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(kBlock), 0, s, blockcount, blockoff, nblk, total);
    hipLaunchKernelGGL((user_walk_kernel<K, V>), dim3(nblk), dim3(kBlock), 0, s, tk, tv, cap, blockoff,
                       what == 3 ? 0 : 1, static_cast<K *>(keys_out), static_cast<V *>(vals_out), max_out);
    if (count_out) hipLaunchKernelGGL((user_store_count_kernel<K>), dim3(1), dim3(1), 0, s, total, static_cast<K *>(count_out));
  }
  SPX_LAUNCH_CHECK();
  return 0;
}

}  // namespace
}  // namespace spx

using namespace spx;

namespace spx {
namespace {
// ---------------------------------------------------------------- SubM row layout
// The default row order of a SubM rulebook (include/spconv_amd.h, spx_subm_layout): the reference sorts the rows of
// every rulebook by mask (SPCONV_DO_SORT, constants.py:121; ops.py:763-785 -> all.py:935-991); here the finished
// masks are classified and, for a sparse rulebook, the rows WITH a neighbour move into a compact appendix -- bucket
// 1 + j = rows whose lowest neighbour offset is j, a stable counting partition -- while the row-order walk keeps the
// rows that only have their centre pair.  count -> scan (one block per bucket) -> scatter; ranks inside a block
// come from wave ballots, every position is a function of the masks alone (no order-dependent atomics).  Nothing is
// read back: class and count land in the blob, the gather-GEMM's appendix workgroups read them.
constexpr int kLayBuckets = 33;       // centre-only + lowest neighbour offset 0 .. 31
constexpr int kLayItems = 256;        // rows per block: one round (a 100 k-row rulebook must fill 256 CUs: 391 blocks)

__device__ __forceinline__ int lay_bucket(uint32_t m, int centre) {
  m &= ~(1u << centre);
  return m ? 1 + __builtin_ctz(m) : 0;
}

__global__ void __launch_bounds__(kBlock)
layout_count_kernel(const uint32_t *__restrict__ mask, int n, int kv, int nblk, int32_t *__restrict__ cnt) {
  __shared__ int h[kLayBuckets];
  if (threadIdx.x < kLayBuckets) h[threadIdx.x] = 0;
  __syncthreads();
  const int centre = kv / 2, begin = blockIdx.x * kLayItems;
#pragma unroll
  for (int it = 0; it < kLayItems / kBlock; ++it) {
    const int i = begin + it * kBlock + threadIdx.x;
    if (i < n) atomicAdd(&h[lay_bucket(mask[i], centre)], 1);      // (counts: the order of the adds is immaterial)
  }
  __syncthreads();
  if (threadIdx.x < kLayBuckets) cnt[static_cast<size_t>(threadIdx.x) * nblk + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(kBlock)
layout_scatter_kernel(const int32_t *__restrict__ pair, const uint32_t *__restrict__ mask, int n, int kv, int nblk,
                      const int32_t *__restrict__ off, const int32_t *__restrict__ totals,
                      int32_t *__restrict__ blob, int npad, int mcap) {
  __shared__ int base[kLayBuckets];              // first appendix position of (bucket, this block); bucket 0 unused
  __shared__ int wcnt[kBlock / 64][kLayBuckets]; // rows of a bucket per wave
  const int centre = kv / 2;
  const int heavy = n - totals[0];
  const int cls = (heavy > 0 && 4ll * heavy < n) ? 1 : 0;
  uint32_t *mask_main = reinterpret_cast<uint32_t *>(blob + SPX_LAYOUT_HEADER);
  int32_t *order = blob + SPX_LAYOUT_HEADER + npad;
  uint32_t *mask_app = reinterpret_cast<uint32_t *>(order + mcap);
  int32_t *pair_app = order + 2 * static_cast<size_t>(mcap);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    blob[0] = cls;
    blob[1] = heavy;
    blob[2] = n;
    blob[3] = kv;
    blob[4] = mcap;
  }
  const int i = blockIdx.x * kLayItems + threadIdx.x;
  const bool ok = i < n;
  const uint32_t m = ok ? mask[i] : 0u;
  const int b = ok ? lay_bucket(m, centre) : -1;
  // the row-order walk keeps every row of a dense rulebook, and the centre-only rows of a sparse one
  if (ok) mask_main[i] = (cls && b > 0) ? 0u : m;
  if (!cls) return;
  if (threadIdx.x < kLayBuckets) {
    int s = 0;
    for (int j = 1; j < static_cast<int>(threadIdx.x); ++j) s += totals[j];
    base[threadIdx.x] = s + off[static_cast<size_t>(threadIdx.x) * nblk + blockIdx.x];
  }
  for (int j = threadIdx.x; j < (kBlock / 64) * kLayBuckets; j += kBlock) (&wcnt[0][0])[j] = 0;
  __syncthreads();
  // rank of a row with a neighbour among the rows of its bucket in this wave: one ballot per DISTINCT bucket present
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool hv = ok && b > 0;
  int rank = 0;
  unsigned long long todo = __ballot(hv);
  while (todo) {
    const int leader = __builtin_ctzll(todo);
    const int lb = __builtin_amdgcn_readlane(b, leader);
    const unsigned long long same = __ballot(hv && b == lb);
    if (b == lb) rank = __popcll(same & ((1ull << lane) - 1ull));
    if (lane == leader) wcnt[wave][lb] = __popcll(same);
    todo &= ~same;
  }
  __syncthreads();
  if (hv) {
    int pos = base[b] + rank;
    for (int w = 0; w < wave; ++w) pos += wcnt[w][b];
    order[pos] = i;
    mask_app[pos] = m;
    for (int k = 0; k < kv; ++k) pair_app[static_cast<size_t>(k) * mcap + pos] = pair[static_cast<size_t>(k) * n + i];
  }
}
}  // namespace
}  // namespace spx

extern "C" {

size_t spx_subm_layout_mcap(int n) {
  const size_t nn = n > 0 ? n : 1;
  return ((nn / 4 + 63) & ~static_cast<size_t>(63)) + 256;
}

size_t spx_subm_layout_bytes(int n, int kv) {
  const size_t nn = n > 0 ? n : 1, npad = (nn + 63) & ~static_cast<size_t>(63), mcap = spx_subm_layout_mcap(n);
  return (SPX_LAYOUT_HEADER + npad + (2 + static_cast<size_t>(kv)) * mcap) * sizeof(int32_t);
}

size_t spx_subm_layout_ws_bytes(int n) {
  const size_t nblk = div_up(n > 0 ? n : 1, kLayItems);
  return 2 * align_up(static_cast<size_t>(kLayBuckets) * nblk * sizeof(int32_t), 256) + 256;
}

int spx_subm_layout(const int32_t *pair_fwd, const uint32_t *mask, int n, int kv, int32_t *layout, void *ws,
                    size_t ws_bytes, spx_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  SPX_CHECK(kv >= 1 && kv <= 32, "a rows layout needs a kernel volume <= 32 (one mask word), got %d", kv);
  if (n <= 0) return 0;
  SPX_CHECK(pair_fwd && mask && layout && ws, "null pointer");
  SPX_CHECK(ws_bytes >= spx_subm_layout_ws_bytes(n), "workspace too small");
  const int nblk = div_up(n, kLayItems);
  const int npad = (n + 63) & ~63;
  Carver cv(ws);
  int32_t *cnt = cv.take<int32_t>(static_cast<size_t>(kLayBuckets) * nblk);
  int32_t *off = cv.take<int32_t>(static_cast<size_t>(kLayBuckets) * nblk);
  int32_t *totals = cv.take<int32_t>(64);
  hipLaunchKernelGGL(layout_count_kernel, dim3(nblk), dim3(kBlock), 0, s, mask, n, kv, nblk, cnt);
  hipLaunchKernelGGL(scan_kernel, dim3(kLayBuckets), dim3(kBlock), 0, s, cnt, off, nblk, totals);
  hipLaunchKernelGGL(layout_scatter_kernel, dim3(nblk), dim3(kBlock), 0, s, pair_fwd, mask, n, kv, nblk, off,
                     totals, layout, npad, static_cast<int>(spx_subm_layout_mcap(n)));
  SPX_LAUNCH_CHECK();
  return 0;
}

size_t spx_subm_rulebook_ws_bytes(int n, int kv) {
  const uint32_t cap = table_capacity(n > 0 ? n : 1) << 1;      // (room for the larger of the two table sizes, below)
  const int nblk = div_up(n > 0 ? n : 1, kItems);
  size_t b = 0;
  b += align_up(cap * sizeof(hkey_t), 256) + align_up(cap * sizeof(int32_t), 256);
  b += 2 * align_up(static_cast<size_t>(kv) * nblk * sizeof(int32_t), 256);
  b += 256;  // scratch totals when num_per_loc is NULL
  b += align_up(static_cast<size_t>(n > 0 ? n : 1) * sizeof(int32_t), 256);   // hash slot of every row
  // second generation: hit counts per (list, 256-voxel group)
  b += align_up(static_cast<size_t>(kv / 2 + 1) * div_up(n > 0 ? n : 1, kBlock) * sizeof(int32_t), 256);
  b += align_up(cap / 8, 256);                // occupancy bit per table slot (subm_probe5_kernel)
  return b;
}

int spx_subm_rulebook(const int32_t *indices, int n, int ndim, int batch_size,
                      const int *spatial_shape, const int *ksize, const int *dilation,
                      int32_t *pair_fwd, int32_t *pair_bwd, uint32_t *mask,
                      int32_t *pair_native, int32_t *num_per_loc, void *ws, size_t ws_bytes,
                      spx_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  int padding[4], stride[4] = {1, 1, 1, 1}, kv = 1;
  SPX_CHECK(ndim >= 1 && ndim <= kMaxNdim, "ndim must be in [1,4], got %d", ndim);
  for (int i = 0; i < ndim; ++i) {
    SPX_CHECK(ksize[i] % 2 == 1, "subm only support odd ksize");  // indices.py:1650
    padding[i] = (ksize[i] / 2) * dilation[i];                    // indices.py:1652
    kv *= ksize[i];
  }
  if (check_geom(ndim, n, kv)) return -1;
  if (n == 0) {
    if (num_per_loc) SPX_HIP(hipMemsetAsync(num_per_loc, 0, sizeof(int32_t) * kv, s));
    return 0;
  }
  SPX_CHECK(pair_fwd && mask, "pair_fwd and mask are required");
  SPX_CHECK(ws_bytes >= spx_subm_rulebook_ws_bytes(n, kv), "workspace too small: %zu < %zu",
            ws_bytes, spx_subm_rulebook_ws_bytes(n, kv));
  const Geom g = make_geom(ndim, batch_size, spatial_shape, spatial_shape, ksize, stride,
                           padding, dilation);
  const int words = div_up(kv, 32);
  if (n == 0) return 0;

  // 4 N slots instead of 2 N while the table stays within 8 MB (two XCD L2s): at load 0.1-0.19 a lookup resolves in
  // ~1.2 probes instead of ~2 -- SubM tables 40.8 -> 36.9 us at 100 k uniform voxels, 67.1 -> 54.6 on the 125 k fixture;
  // beyond (400 k voxels: 16 MB) the extra lines cost more than the probes save (151 -> 159 us)
  uint32_t cap = table_capacity(n);
  if (cap <= (1u << 19)) cap <<= 1;
  const int nblk = div_up(n, kItems);
  Carver cv(ws);
  Table t;
  {
    hkey_t *keys = cv.take<hkey_t>(cap);
    table_place(t, keys, cv.take<int32_t>(cap), cap, keys_fit_u32(g.batch, g.in_dims, 4));
  }
  int32_t *blockcount = cv.take<int32_t>(static_cast<size_t>(kv) * nblk);
  int32_t *blockoff = cv.take<int32_t>(static_cast<size_t>(kv) * nblk);
  int32_t *scratch_totals = cv.take<int32_t>(64);
  int32_t *slot_of = cv.take<int32_t>(n);
  const int nblk256 = div_up(n, kBlock);
  int32_t *groupcount = cv.take<int32_t>(static_cast<size_t>(kv / 2 + 1) * nblk256);
  uint32_t *occupied = cv.take<uint32_t>(cap / 32);
  // probe pass, fifth form (subm_probe5_kernel): behind an occupancy bit per slot in LDS for tables up to 2^19 slots
  // (64 KB of bits, two workgroups per CU); larger tables take the same kernel without the bits
  // -- the north star's "LDS-staged open-address hashing".  Measured (profiles/r04_experiments.md 1f, r05 7): 35.4 vs
  // 36.9 us at 100 k uniform voxels, level at 100-125 k LiDAR-density voxels; without the bits (tables beyond 2^19 slots:
  // 128 KB of bits would leave one workgroup per CU) the fourth form is faster (150 vs 163 us at 400 k) and stays.
  const bool probe5 = option_int("SPX_SUBM_PROBE", 5) >= 5 && cap >= 1024u && cap <= (1u << 19) && kv <= 128;
  const bool occ_bits = probe5;
  const dim3 grid(div_up(n, kBlock));
  // second generation: 4 launches (table fill, insert, probe, lists), no table pre-fills; beyond
  // ~4 M voxels the lists kernel's in-block prefix over the group counts would dominate
  // third form: fills (table, lower half of pair_fwd [+ pair_bwd's upper half]) -> insert (+ mask
  // clear) -> probe4 (block-local list counts) -> lists: 4 launches, 11 MB of fills instead of 34
  if (kv > 1 && kv <= 128 && nblk256 <= 16384) {
    FillList fills;                    // the hash table only: the -1 halves of the tables ride in the insert kernel
    table_fill(fills, t);              // (pair_fwd rows k < centre; pair_bwd[kv-1-kk] mirrors pair_fwd[kk]: its rows above)
    if (occ_bits) fills.add(occupied, cap / 8, 0u);
    SPX_HIP(fills.launch(s));
    // masks from a pass over the finished table instead of one atomicOr per entry: the extra launch costs 5-10 us at
    // 100 k voxels, the saved atomics (20-25 G/s device-wide) win from ~250 k (400 k: 162 -> 151 us); -1 = by size
    const int mp_opt = option_int("SPX_SUBM_MASK_PASS", -1);
    const int mask_pass = mp_opt < 0 ? (n >= 250000 ? 1 : 0) : mp_opt;
    hipLaunchKernelGGL(subm_insert_kernel, grid, dim3(kBlock), 0, s, indices, n, g, t, slot_of,
                       mask_pass ? static_cast<uint32_t *>(nullptr) : mask, words, pair_fwd, pair_bwd,
                       occ_bits ? occupied : static_cast<uint32_t *>(nullptr));
    const bool lists = pair_native || num_per_loc;
    if (probe5) {
      const int fwords = occ_bits ? static_cast<int>(cap / 32) : 0;
      const size_t lds = static_cast<size_t>(fwords) * 4 + static_cast<size_t>(kv / 2) * (16 + 8 + 16);
      static std::atomic<uint64_t> attr_done{0};      // one bit per device (common.h: ensure_dynamic_lds)
      SPX_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(&subm_probe5_kernel), 160 * 1024, attr_done));
      hipLaunchKernelGGL(subm_probe5_kernel, dim3(nblk256), dim3(kP5Threads), lds, s, indices, n, g, t, occupied, fwords,
                         slot_of, pair_fwd, pair_bwd, mask, words, lists ? groupcount : nullptr, nblk256, mask_pass);
    } else {
      hipLaunchKernelGGL(subm_probe4_kernel, dim3(div_up(n, kBlock), kv / 2 + 1), dim3(kBlock), 0, s, indices, n,
                         g, t, slot_of, pair_fwd, pair_bwd, mask, words, lists ? groupcount : nullptr, nblk256, mask_pass);
    }
    if (mask_pass)
      hipLaunchKernelGGL(mask_from_table_kernel, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, pair_fwd, kv, n, words, mask);
    if (lists) {
      SPX_CHECK(!pair_native || num_per_loc || kv / 2 <= 64, "num_per_loc required for kv > 128");
      hipLaunchKernelGGL(subm_lists_kernel, dim3(nblk, kv / 2 + 1), dim3(kBlock), 0, s, pair_fwd, kv, n,
                         nblk256, groupcount, pair_native, num_per_loc ? num_per_loc : scratch_totals,
                         num_per_loc ? kv : 0);
    }
    SPX_LAUNCH_CHECK();
    return 0;
  }
  // every fill of this build in one launch: hash table, masks, counts, -1 tables (callers that
  // carve the tables out of one buffer get one contiguous range)
  FillList fills;
  table_fill(fills, t);
  fills.add(mask, sizeof(uint32_t) * static_cast<size_t>(n) * words, 0u);
  if (num_per_loc) fills.add(num_per_loc, sizeof(int32_t) * kv, 0u);
  {
    const size_t tb = sizeof(int32_t) * static_cast<size_t>(kv) * n;
    fills.add(pair_fwd, tb, 0xFFFFFFFFu);
    if (pair_bwd) fills.add(pair_bwd, tb, 0xFFFFFFFFu);
    if (pair_native) fills.add(pair_native, 2 * tb, 0xFFFFFFFFu);
  }
  SPX_HIP(fills.launch(s));
  hipLaunchKernelGGL(subm_insert_kernel, grid, dim3(kBlock), 0, s, indices, n, g, t, slot_of);
  hipLaunchKernelGGL(subm_probe3_kernel, dim3(div_up(n, kBlock), kv / 2 + 1), dim3(kBlock), 0, s, indices,
                     n, g, t, slot_of, pair_fwd, pair_bwd, mask, words, pair_native);
  SPX_LAUNCH_CHECK();
  if (pair_native) {
    // num_per_loc: counts only for k < kv/2 (indices.py:1685,1692)
    int32_t *totals = num_per_loc ? num_per_loc : scratch_totals;
    SPX_CHECK(num_per_loc || kv / 2 <= 64, "num_per_loc required for kv > 128");
    if (launch_native_lists(pair_fwd, 0, kv, n, kv / 2, nblk, blockcount, blockoff, pair_native,
                            totals, s))
      return -2;
  } else if (num_per_loc && kv / 2 > 0) {
    dim3 g2(nblk, kv / 2);
    hipLaunchKernelGGL(compact_count_kernel, g2, dim3(kBlock), 0, s, pair_fwd, 0, kv, n, nblk,
                       blockcount);
    hipLaunchKernelGGL(scan_kernel, dim3(kv / 2), dim3(kBlock), 0, s, blockcount, blockoff, nblk,
                       num_per_loc);
    SPX_LAUNCH_CHECK();
  }
  return 0;
}

size_t spx_conv_rulebook_ws_bytes(int n_in, int ndim, const int *ksize, const int *stride,
                                  const int *dilation, int transposed) {
  if (ndim < 1 || ndim > kMaxNdim) return 0;
  return carve_conv_ws(nullptr, n_in, ndim, ksize, stride, dilation, transposed).bytes + 256;
}

}  // extern "C"

namespace spx {
namespace {
// Table sizing of the compact passes.  The guaranteed bound (conv_max_out: 8 N for k = 3 / s = 2) gives
// a table that is 4 % full on a LiDAR scene -- 67 MB of slots for 313 k outputs at 400 k inputs, every
// probe its own HBM line, 20 us of fill.  The builder therefore sizes the table for the outputs it
// EXPECTS and keeps the bound as the fallback:
//   * static-shape form: the caller's output bound (more distinct candidates than 2x that: the
//     overflow flag the caller reads back with the count)
//   * two-call form: the outputs-per-input ratio the same geometry produced last time (+25 %), kept in
//     a small direct-mapped cache -- the role of the reference's per-problem tuner cache
//     (convops.py:1150,1283-1297); a table that overflows is seen in the count's read-back and the
//     pass is run again at the guaranteed size.  Results never depend on the size.
struct RatioEntry {
  unsigned long long key;
  int ratio_x64;                     // ceil(64 * n_out / n_in) of the last build, 0 = none yet
};
RatioEntry g_ratio[64];
std::mutex g_ratio_mutex;

unsigned long long geometry_key(int ndim, const int *in_shape, const int *ksize, const int *stride,
                                const int *padding, const int *dilation) {
  unsigned long long h = 1469598103934665603ull ^ static_cast<unsigned>(ndim);
  for (int i = 0; i < ndim; ++i)
    for (const int v : {in_shape[i], ksize[i], stride[i], padding[i], dilation ? dilation[i] : 1}) {
      h ^= static_cast<unsigned>(v);
      h *= 1099511628211ull;
    }
  return h | 1ull;
}

size_t expected_outputs(unsigned long long key, int n_in) {
  std::lock_guard<std::mutex> lock(g_ratio_mutex);
  const RatioEntry &e = g_ratio[key % 64];
  if (e.key != key || e.ratio_x64 <= 0) return 0;                       // unknown: the bound
  return static_cast<size_t>(n_in) * e.ratio_x64 / 64 * 5 / 4 + 4096;
}

void remember_outputs(unsigned long long key, int n_in, int n_out) {
  std::lock_guard<std::mutex> lock(g_ratio_mutex);
  RatioEntry &e = g_ratio[key % 64];
  e.key = key;
  e.ratio_x64 = static_cast<int>((static_cast<long long>(n_out) * 64 + n_in - 1) / (n_in > 0 ? n_in : 1)) + 1;
}

// expect_out: distinct outputs to size the hash table for (0 = the guaranteed bound).  *overflow_h (with
// n_out_h): the table filled up -- the caller decides whether a larger table exists.
int conv_count_impl(const int32_t *indices, int n_in, int ndim, int batch_size,
                    const int *in_shape, const int *out_shape, const int *ksize,
                    const int *stride, const int *padding, const int *dilation,
                    int transposed, void *ws, size_t ws_bytes, int *n_out_h, int *overflow_h,
                    size_t expect_out, spx_stream_t stream, const FillList *more = nullptr,
                    int32_t *nout_dev = nullptr) {
  // more: fills of the caller that ride in this pass's fill launch; nout_dev: where {count, overflow}
  // go instead of the workspace (static-shape form: no copy afterwards)
  hipStream_t s = static_cast<hipStream_t>(stream);
  SPX_CHECK(ndim >= 1 && ndim <= kMaxNdim, "ndim must be in [1,4], got %d", ndim);
  const Geom g = make_geom(ndim, batch_size, in_shape, out_shape, ksize, stride, padding, dilation);
  if (check_geom(ndim, n_in, g.kv)) return -1;
  for (int i = 0; i < ndim; ++i)
    SPX_CHECK(out_shape[i] > 0 && stride[i] > 0, "bad output shape / stride at dim %d", i);
  SPX_CHECK(ws_bytes >= spx_conv_rulebook_ws_bytes(n_in, ndim, ksize, stride, dilation, transposed),
            "workspace too small");
  if (n_out_h) *n_out_h = 0;
  if (n_in == 0) return 0;
  ConvWs w = carve_conv_ws(ws, n_in, ndim, ksize, stride, dilation, transposed,
                           keys_fit_u32(g.batch, g.out_dims, 4));
  const int mj = conv3_cands(ndim, in_shape, ksize, stride, padding, dilation, transposed);
  if (mj && expect_out > 0) {            // (the later passes of this form never touch the table)
    const uint32_t cap = table_capacity(expect_out);
    if (cap < w.t.mask + 1u) table_shrink(w.t, cap);
  }
  {
    FillList fills;                      // table and flags in one launch
    table_fill(fills, w.t);
    if (nout_dev) w.d_nout = nout_dev;
    fills.add(w.d_nout, 2 * sizeof(int32_t), 0u);
    if (more)
      for (int j = 0; j < more->jobs.n; ++j)
        fills.add(more->jobs.ptr[j], more->jobs.words[j] * 4, more->jobs.value[j]);
    if (mj) fills.add(w.firstbits, sizeof(uint32_t) * static_cast<size_t>(g.kv) * w.nblk * (kItems / 32), 0u);
    SPX_HIP(fills.launch(s));
  }
  if (mj) {
    const int shares = conv3_shares(n_in, mj);
    SPX_CONV3_LAUNCH(conv3_insert_kernel, mj, dim3(div_up(n_in, kBlock), shares), dim3(kBlock), 0, s, indices, n_in, g,
                     w.t, w.slot_of, w.d_nout + 1);
    SPX_CONV3_LAUNCH(conv3_first_kernel, mj, dim3(div_up(n_in, kBlock), shares), dim3(kBlock), 0, s, indices, n_in, g, w.t,
                     static_cast<const int32_t *>(w.slot_of), w.nblk, w.firstbits, w.firstflags);
    hipLaunchKernelGGL(conv3_count_kernel, dim3(div_up(g.kv * w.nblk, kBlock / 64)), dim3(kBlock), 0, s,
                       static_cast<const uint32_t *>(w.firstbits), g.kv * w.nblk, w.wordpre, w.blockcount);
  } else {
    const dim3 grid1(div_up(n_in, kBlock), g.kv);
    hipLaunchKernelGGL(conv_stage1_kernel, grid1, dim3(kBlock), 0, s, indices, n_in, g, transposed,
                       w.t, w.slot_of, w.d_nout + 1);
    const dim3 grid2(w.nblk, g.kv);
    hipLaunchKernelGGL(conv_count_first_kernel, grid2, dim3(kBlock), 0, s, w.slot_of, w.t, n_in,
                       w.nblk, w.blockcount);
  }
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(kBlock), 0, s, w.blockcount, w.blockoff,
                     g.kv * w.nblk, w.d_nout);
  SPX_LAUNCH_CHECK();
  if (!n_out_h) return 0;                // static-shape form: the count stays on the device
  int32_t host_n[2] = {0, 0};
  SPX_HIP(hipMemcpyAsync(host_n, w.d_nout, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  SPX_HIP(hipStreamSynchronize(s));
  *n_out_h = host_n[0];
  *overflow_h = host_n[1];
  return 0;
}
}  // namespace
}  // namespace spx

extern "C" {

int spx_conv_rulebook_count(const int32_t *indices, int n_in, int ndim, int batch_size,
                            const int *in_shape, const int *out_shape, const int *ksize,
                            const int *stride, const int *padding, const int *dilation,
                            int transposed, void *ws, size_t ws_bytes, int *n_out_h,
                            spx_stream_t stream) {
  using namespace spx;
  SPX_CHECK(ndim >= 1 && ndim <= kMaxNdim, "ndim must be in [1,4], got %d", ndim);
  int overflow = 0;
  if (!n_out_h)      // nothing read back: the guaranteed size
    return conv_count_impl(indices, n_in, ndim, batch_size, in_shape, out_shape, ksize, stride, padding,
                           dilation, transposed, ws, ws_bytes, nullptr, &overflow, 0, stream);
  const unsigned long long key = geometry_key(ndim, in_shape, ksize, stride, padding, dilation);
  const size_t expect = expected_outputs(key, n_in);
  int rc = conv_count_impl(indices, n_in, ndim, batch_size, in_shape, out_shape, ksize, stride, padding,
                           dilation, transposed, ws, ws_bytes, n_out_h, &overflow, expect, stream);
  if (rc) return rc;
  if (overflow && expect > 0) {          // the expectation was too small: once more, at the bound
    rc = conv_count_impl(indices, n_in, ndim, batch_size, in_shape, out_shape, ksize, stride, padding,
                         dilation, transposed, ws, ws_bytes, n_out_h, &overflow, 0, stream);
    if (rc) return rc;
  }
  SPX_CHECK(overflow == 0, "output hash table overflow: more distinct outputs than the bound %zu",
            conv_max_out(n_in, ndim, ksize, stride, dilation, transposed));
  if (n_in > 0) remember_outputs(key, n_in, *n_out_h);
  return 0;
}


}  // extern "C"

namespace spx {
namespace {
// prefilled: pair_fwd already holds -1 (the static-shape form puts that fill into the first launch)
int conv_fill_impl(const int32_t *indices, int n_in, int ndim, int batch_size,
                   const int *in_shape, const int *out_shape, const int *ksize,
                   const int *stride, const int *padding, const int *dilation,
                   int transposed, int n_out, int32_t *out_indices, int32_t *pair_fwd,
                   int32_t *pair_bwd, uint32_t *mask_fwd, uint32_t *mask_bwd,
                   int32_t *pair_native, int32_t *num_per_loc, void *ws,
                   size_t ws_bytes, spx_stream_t stream, bool prefilled) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  SPX_CHECK(ndim >= 1 && ndim <= kMaxNdim, "ndim must be in [1,4], got %d", ndim);
  const Geom g = make_geom(ndim, batch_size, in_shape, out_shape, ksize, stride, padding, dilation);
  if (check_geom(ndim, n_in, g.kv)) return -1;
  SPX_CHECK(ws_bytes >= spx_conv_rulebook_ws_bytes(n_in, ndim, ksize, stride, dilation, transposed),
            "workspace too small");
  SPX_CHECK(pair_fwd && pair_bwd && out_indices, "out_indices, pair_fwd and pair_bwd are required");
  const int kv = g.kv, words = div_up(kv, 32);
  const int ngroups = div_up(n_in > 0 ? n_in : 1, kBlock);
  const int version = 2;
  // second form: the Native lists come from subm_lists_kernel (conv mode) over the 256-row pair counts
  // stage 2 leaves behind -- no count / scan launches, no -1 pre-fill of the lists
  const bool v2 = version >= 2 && (pair_native || num_per_loc) && ngroups <= 16384 && kv <= 128;
  FillList fills;                                           // every fill of this call: one launch
  if (!v2) {
    if (num_per_loc) fills.add(num_per_loc, sizeof(int32_t) * kv, 0u);
    if (pair_native && n_in > 0)
      fills.add(pair_native, sizeof(int32_t) * 2 * static_cast<size_t>(kv) * n_in, 0xFFFFFFFFu);
  }
  if (n_in > 0 && n_out > 0 && !prefilled)
    fills.add(pair_fwd, sizeof(int32_t) * static_cast<size_t>(kv) * n_out, 0xFFFFFFFFu);
  if (v2 && n_in == 0 && num_per_loc) fills.add(num_per_loc, sizeof(int32_t) * kv, 0u);
  SPX_HIP(fills.launch(s));
  if (n_in == 0) return 0;
  ConvWs w = carve_conv_ws(ws, n_in, ndim, ksize, stride, dilation, transposed,
                           keys_fit_u32(g.batch, g.out_dims, 4));   // as spx_conv_rulebook_count
  const dim3 grid2(w.nblk, kv);
  const int mj = conv3_cands(ndim, in_shape, ksize, stride, padding, dilation, transposed);
  if (mj) {
    SPX_CONV3_LAUNCH(conv3_assign_kernel, mj, dim3(div_up(n_in, kBlock), conv3_shares(n_in, mj)), dim3(kBlock), 0, s, indices, n_in, g,
                     static_cast<const int32_t *>(w.slot_of), w.nblk, static_cast<const uint32_t *>(w.firstbits),
                     static_cast<const uint8_t *>(w.firstflags), static_cast<const int32_t *>(w.wordpre),
                     static_cast<const int32_t *>(w.blockoff), w.slot_out, out_indices, n_out);
    SPX_CONV3_LAUNCH(conv3_pairs_kernel, mj, dim3(div_up(n_in, kBlock)), dim3(kBlock), 0, s, indices, n_in, g,
                     static_cast<const int32_t *>(w.slot_of), static_cast<const int32_t *>(w.slot_out), n_out,
                     pair_fwd, pair_bwd, mask_bwd, words, v2 ? w.groupcount : nullptr);
  } else {
    hipLaunchKernelGGL(conv_assign_kernel, grid2, dim3(kBlock), 0, s, indices, n_in, g, transposed,
                       w.slot_of, w.t, w.nblk, w.blockoff, w.slot_out, out_indices, n_out);
    const dim3 grid1(div_up(n_in, kBlock), kv);
    hipLaunchKernelGGL(conv_stage2_kernel, grid1, dim3(kBlock), 0, s, w.slot_of, w.slot_out, n_in,
                       n_out, pair_fwd, pair_bwd, v2 ? w.groupcount : nullptr);
  }
  {
    const int na = mask_fwd ? n_out : 0, nb = (mask_bwd && !mj) ? n_in : 0;
    if (na + nb > 0)
      hipLaunchKernelGGL(mask_from_tables_kernel, dim3(div_up(na + nb, kBlock)), dim3(kBlock), 0, s, pair_fwd,
                         na, mask_fwd, pair_bwd, nb, mask_bwd, kv, words);
  }
  SPX_LAUNCH_CHECK();
  if (v2) {
    SPX_CHECK(!pair_native || num_per_loc, "num_per_loc is required with pair_native");
    hipLaunchKernelGGL(subm_lists_kernel, dim3(w.nblk, kv), dim3(kBlock), 0, s, pair_bwd, kv, n_in, ngroups,
                       w.groupcount, pair_native, num_per_loc, 0, 1);
    SPX_LAUNCH_CHECK();
    return 0;
  }
  if (pair_native) {
    SPX_CHECK(num_per_loc, "num_per_loc is required with pair_native");
    if (launch_native_lists(pair_bwd, 1, kv, n_in, kv, w.nblk, w.blockcount, w.blockoff,
                            pair_native, num_per_loc, s))
      return -2;
  } else if (num_per_loc) {
    hipLaunchKernelGGL(compact_count_kernel, grid2, dim3(kBlock), 0, s, pair_bwd, 1, kv, n_in,
                       w.nblk, w.blockcount);
    hipLaunchKernelGGL(scan_kernel, dim3(kv), dim3(kBlock), 0, s, w.blockcount, w.blockoff, w.nblk,
                       num_per_loc);
    SPX_LAUNCH_CHECK();
  }
  return 0;
}
}  // namespace
}  // namespace spx

extern "C" {

int spx_conv_rulebook_fill(const int32_t *indices, int n_in, int ndim, int batch_size,
                           const int *in_shape, const int *out_shape, const int *ksize,
                           const int *stride, const int *padding, const int *dilation,
                           int transposed, int n_out, int32_t *out_indices, int32_t *pair_fwd,
                           int32_t *pair_bwd, uint32_t *mask_fwd, uint32_t *mask_bwd,
                           int32_t *pair_native, int32_t *num_per_loc, void *ws,
                           size_t ws_bytes, spx_stream_t stream) {
  return spx::conv_fill_impl(indices, n_in, ndim, batch_size, in_shape, out_shape, ksize, stride, padding,
                             dilation, transposed, n_out, out_indices, pair_fwd, pair_bwd, mask_fwd, mask_bwd,
                             pair_native, num_per_loc, ws, ws_bytes, stream, false);
}

int spx_conv_rulebook_static(const int32_t *indices, int n_in, int ndim, int batch_size,
                             const int *in_shape, const int *out_shape, const int *ksize,
                             const int *stride, const int *padding, const int *dilation,
                             int transposed, int n_out_cap, int32_t *out_indices, int32_t *pair_fwd,
                             int32_t *pair_bwd, uint32_t *mask_fwd, uint32_t *mask_bwd,
                             int32_t *pair_native, int32_t *num_per_loc, int32_t *n_out_dev, void *ws,
                             size_t ws_bytes, spx_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  SPX_CHECK(n_out_cap > 0 && n_out_dev && out_indices, "n_out_cap > 0, n_out_dev and out_indices are required");
  SPX_CHECK(n_in > 0, "static-shape rulebook needs n_in > 0 (pad the input with batch = -1 rows)");
  // every launch below is stream-ordered and nothing is read back: the whole call can sit in a
  // hipGraph.  Rows past the number of distinct outputs keep out_indices = -1 (a dead row for the
  // next layer: subm_insert_kernel / conv_stage1_kernel skip batch < 0), pair_fwd = -1, mask = 0.
  // Both sizes are known up front, so the -1 fills of the outputs ride in the first pass's fill launch
  // and {distinct outputs found (may exceed the cap: the first n_out_cap survive), hash-table overflow
  // flag} are written straight into n_out_dev: two launches and a copy fewer than the two-call form.
  SPX_CHECK(pair_fwd && pair_bwd, "pair_fwd and pair_bwd are required");
  (void)s;
  int kv = 1;
  for (int i = 0; i < ndim; ++i) kv *= ksize[i];
  spx::FillList pre;
  pre.add(out_indices, sizeof(int32_t) * static_cast<size_t>(n_out_cap) * (ndim + 1), 0xFFFFFFFFu);
  pre.add(pair_fwd, sizeof(int32_t) * static_cast<size_t>(kv) * n_out_cap, 0xFFFFFFFFu);
  int unused = 0;
  int rc = spx::conv_count_impl(indices, n_in, ndim, batch_size, in_shape, out_shape, ksize, stride, padding,
                                dilation, transposed, ws, ws_bytes, nullptr, &unused,
                                static_cast<size_t>(n_out_cap), stream, &pre, n_out_dev);
  if (rc) return rc;
  return spx::conv_fill_impl(indices, n_in, ndim, batch_size, in_shape, out_shape, ksize, stride, padding,
                             dilation, transposed, n_out_cap, out_indices, pair_fwd, pair_bwd, mask_fwd, mask_bwd,
                             pair_native, num_per_loc, ws, ws_bytes, stream, true);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Sorted-order builds (conv4_* / subm_rank_probe_kernel above): the rank map is the caller's buffer -- it outlives
// the call, the SubM layers of the level read it.
namespace spx {
namespace {
// words of a level's rank map (0: the key space does not fit)
size_t rank_words(int ndim, int batch_size, const int *shape) {
  if (ndim < 1 || ndim > kMaxNdim || batch_size < 1) return 0;
  unsigned long long cells = static_cast<unsigned long long>(batch_size);
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] < 1) return 0;
    cells *= static_cast<unsigned long long>(shape[i]);
    if (cells > 0x7fffffe0ull) return 0;
  }
  return static_cast<size_t>((cells + 31) / 32);
}

// the caller's rank-map buffer: W {bits, prefix} words, then the occupied cells before each 2048-word block
size_t rank_cells_bytes(size_t W) { return align_up(W * sizeof(uint2), 256); }
size_t rank_blocks(size_t W) { return (W + kRankWords - 1) / kRankWords; }
size_t rank_bytes(size_t W) { return W ? rank_cells_bytes(W) + align_up(rank_blocks(W) * sizeof(int32_t), 256) : 0; }
int32_t *rank_blockoff(void *rankmap, size_t W) {
  return reinterpret_cast<int32_t *>(static_cast<char *>(rankmap) + rank_cells_bytes(W));
}

struct Conv4Ws {
  int32_t *blockcount, *d_nout, *groupcount;
  uint8_t *occupied;                   // one byte per cell (32 per word of the rank map), alive between mark and prefix
  int nblk;
  size_t bytes;
};
Conv4Ws carve_conv4_ws(void *ws, int n_in, int kv, size_t W) {
  Conv4Ws w;
  w.nblk = static_cast<int>((W + kRankWords - 1) / kRankWords);
  Carver cv(ws);
  w.occupied = cv.take<uint8_t>((W > 0 ? W : 1) * 32);
  w.blockcount = cv.take<int32_t>(w.nblk > 0 ? w.nblk : 1);
  w.d_nout = cv.take<int32_t>(2);
  w.groupcount = cv.take<int32_t>(static_cast<size_t>(kv) * div_up(n_in > 0 ? n_in : 1, kBlock));
  w.bytes = cv.off;
  return w;
}

int conv4_count_impl(const int32_t *indices, int n_in, int ndim, int batch_size, const int *in_shape,
                     const int *out_shape, const int *ksize, const int *stride, const int *padding,
                     const int *dilation, void *rankmap, size_t rankmap_bytes, void *ws, size_t ws_bytes,
                     int *n_out_h, hipStream_t s, const FillList *more, int32_t *nout_dev) {
  SPX_CHECK(ndim >= 1 && ndim <= kMaxNdim, "ndim must be in [1,4], got %d", ndim);
  const Geom g = make_geom(ndim, batch_size, in_shape, out_shape, ksize, stride, padding, dilation);
  if (check_geom(ndim, n_in, g.kv)) return -1;
  const int mj = conv3_cands(ndim, in_shape, ksize, stride, padding, dilation, 0);
  SPX_CHECK(mj > 0, "sorted-order build: this geometry takes the first-seen builder (spx_conv_sorted_ok)");
  const size_t W = rank_words(ndim, batch_size, out_shape);
  SPX_CHECK(W > 0 && rankmap && rankmap_bytes >= rank_bytes(W), "rank map missing or too small (%zu words)", W);
  Conv4Ws w = carve_conv4_ws(ws, n_in, g.kv, W);
  SPX_CHECK(ws && ws_bytes >= w.bytes, "workspace too small");
  if (n_out_h) *n_out_h = 0;
  uint2 *cells = static_cast<uint2 *>(rankmap);
  if (nout_dev) w.d_nout = nout_dev;
  {
    FillList fills;
    fills.add(w.occupied, W * 32, 0u);
    fills.add(w.d_nout, 2 * sizeof(int32_t), 0u);
    if (more)
      for (int j = 0; j < more->jobs.n; ++j)
        fills.add(more->jobs.ptr[j], more->jobs.words[j] * 4, more->jobs.value[j]);
    SPX_HIP(fills.launch(s));
  }
  if (n_in > 0) {
    SPX_CONV3_LAUNCH(conv4_mark_kernel, mj, dim3(div_up(n_in, kBlock)), dim3(kBlock), 0, s, indices, n_in, g, w.occupied);
    hipLaunchKernelGGL(conv4_prefix_kernel, dim3(w.nblk), dim3(kBlock), 0, s,
                       reinterpret_cast<const uint4 *>(w.occupied), cells, static_cast<unsigned>(W), w.blockcount);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(kBlock), 0, s, w.blockcount, rank_blockoff(rankmap, W), w.nblk, w.d_nout);
    SPX_LAUNCH_CHECK();
  }
  if (!n_out_h) return 0;                // static-shape form: the count stays on the device
  int32_t host_n[2] = {0, 0};
  SPX_HIP(hipMemcpyAsync(host_n, w.d_nout, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  SPX_HIP(hipStreamSynchronize(s));
  *n_out_h = host_n[0];
  return 0;
}

int conv4_fill_impl(const int32_t *indices, int n_in, int ndim, int batch_size, const int *in_shape,
                    const int *out_shape, const int *ksize, const int *stride, const int *padding,
                    const int *dilation, int n_out, int32_t *out_indices, int32_t *pair_fwd, int32_t *pair_bwd,
                    uint32_t *mask_fwd, uint32_t *mask_bwd, int32_t *pair_native, int32_t *num_per_loc,
                    void *rankmap, size_t rankmap_bytes, void *ws, size_t ws_bytes, hipStream_t s, bool prefilled,
                    int32_t *nout_dev = nullptr) {
  const Geom g = make_geom(ndim, batch_size, in_shape, out_shape, ksize, stride, padding, dilation);
  if (check_geom(ndim, n_in, g.kv)) return -1;
  const int mj = conv3_cands(ndim, in_shape, ksize, stride, padding, dilation, 0);
  SPX_CHECK(mj > 0, "sorted-order build: this geometry takes the first-seen builder (spx_conv_sorted_ok)");
  const size_t W = rank_words(ndim, batch_size, out_shape);
  SPX_CHECK(W > 0 && rankmap && rankmap_bytes >= rank_bytes(W), "rank map missing or too small (%zu words)", W);
  Conv4Ws w = carve_conv4_ws(ws, n_in, g.kv, W);
  SPX_CHECK(ws && ws_bytes >= w.bytes, "workspace too small");
  SPX_CHECK(pair_fwd && pair_bwd && out_indices, "out_indices, pair_fwd and pair_bwd are required");
  const int kv = g.kv, words = div_up(kv, 32);
  const int ngroups = div_up(n_in > 0 ? n_in : 1, kBlock);
  const bool lists = pair_native || num_per_loc;
  SPX_CHECK(!lists || (ngroups <= 16384 && kv <= 128), "Native lists of a sorted-order build: too many rows / offsets");
  SPX_CHECK(!pair_native || num_per_loc, "num_per_loc is required with pair_native");
  uint2 *cells = static_cast<uint2 *>(rankmap);
  {
    FillList fills;
    if (n_in > 0 && n_out > 0 && !prefilled)
      fills.add(pair_fwd, sizeof(int32_t) * static_cast<size_t>(kv) * n_out, 0xFFFFFFFFu);
    if (n_in == 0 && num_per_loc) fills.add(num_per_loc, sizeof(int32_t) * kv, 0u);
    SPX_HIP(fills.launch(s));
  }
  if (n_in == 0) return 0;
  SPX_CONV3_LAUNCH(conv4_pairs_kernel, mj, dim3(div_up(n_in, kBlock)), dim3(kBlock), 0, s, indices, n_in, g,
                   static_cast<const uint2 *>(cells), static_cast<const int32_t *>(rank_blockoff(rankmap, W)), n_out,
                   out_indices, pair_fwd, pair_bwd, mask_bwd, words, lists ? w.groupcount : nullptr, nout_dev);
  if (mask_fwd && n_out > 0)
    hipLaunchKernelGGL(mask_from_tables_kernel, dim3(div_up(n_out, kBlock)), dim3(kBlock), 0, s, pair_fwd, n_out, mask_fwd,
                       pair_bwd, 0, mask_bwd, kv, words);
  if (lists)
    hipLaunchKernelGGL(subm_lists_kernel, dim3(div_up(n_in, kItems), kv), dim3(kBlock), 0, s, pair_bwd, kv, n_in, ngroups,
                       w.groupcount, pair_native, num_per_loc, 0, 1);
  SPX_LAUNCH_CHECK();
  return 0;
}
}  // namespace
}  // namespace spx

extern "C" {

size_t spx_rankmap_bytes(int ndim, int batch_size, const int *shape) {
  return spx::rank_bytes(spx::rank_words(ndim, batch_size, shape));
}

int spx_conv_sorted_ok(int ndim, int batch_size, const int *in_shape, const int *out_shape, const int *ksize,
                       const int *stride, const int *padding, const int *dilation, int transposed) {
  if (ndim < 1 || ndim > spx::kMaxNdim || transposed) return 0;
  return spx::conv3_cands(ndim, in_shape, ksize, stride, padding, dilation, 0) > 0 &&
         spx::rank_words(ndim, batch_size, out_shape) > 0;
}

size_t spx_conv_rulebook_sorted_ws_bytes(int n_in, int ndim, int batch_size, const int *out_shape, const int *ksize) {
  if (ndim < 1 || ndim > spx::kMaxNdim) return 0;
  int kv = 1;
  for (int i = 0; i < ndim; ++i) kv *= ksize[i];
  return spx::carve_conv4_ws(nullptr, n_in, kv, spx::rank_words(ndim, batch_size, out_shape)).bytes + 256;
}

int spx_conv_rulebook_count_sorted(const int32_t *indices, int n_in, int ndim, int batch_size, const int *in_shape,
                                   const int *out_shape, const int *ksize, const int *stride, const int *padding,
                                   const int *dilation, void *rankmap, size_t rankmap_bytes, void *ws,
                                   size_t ws_bytes, int *n_out_h, spx_stream_t stream) {
  SPX_CHECK(n_out_h, "n_out_h is required");
  return spx::conv4_count_impl(indices, n_in, ndim, batch_size, in_shape, out_shape, ksize, stride, padding, dilation,
                               rankmap, rankmap_bytes, ws, ws_bytes, n_out_h, static_cast<hipStream_t>(stream),
                               nullptr, nullptr);
}

int spx_conv_rulebook_fill_sorted(const int32_t *indices, int n_in, int ndim, int batch_size, const int *in_shape,
                                  const int *out_shape, const int *ksize, const int *stride, const int *padding,
                                  const int *dilation, int n_out, int32_t *out_indices, int32_t *pair_fwd,
                                  int32_t *pair_bwd, uint32_t *mask_fwd, uint32_t *mask_bwd, int32_t *pair_native,
                                  int32_t *num_per_loc, void *rankmap, size_t rankmap_bytes, void *ws,
                                  size_t ws_bytes, spx_stream_t stream) {
  return spx::conv4_fill_impl(indices, n_in, ndim, batch_size, in_shape, out_shape, ksize, stride, padding, dilation,
                              n_out, out_indices, pair_fwd, pair_bwd, mask_fwd, mask_bwd, pair_native, num_per_loc,
                              rankmap, rankmap_bytes, ws, ws_bytes, static_cast<hipStream_t>(stream), false);
}

int spx_conv_rulebook_static_sorted(const int32_t *indices, int n_in, int ndim, int batch_size, const int *in_shape,
                                    const int *out_shape, const int *ksize, const int *stride, const int *padding,
                                    const int *dilation, int n_out_cap, int32_t *out_indices, int32_t *pair_fwd,
                                    int32_t *pair_bwd, uint32_t *mask_fwd, uint32_t *mask_bwd, int32_t *pair_native,
                                    int32_t *num_per_loc, int32_t *n_out_dev, void *rankmap, size_t rankmap_bytes,
                                    void *ws, size_t ws_bytes, spx_stream_t stream) {
  SPX_CHECK(n_out_cap > 0 && n_out_dev && out_indices, "n_out_cap > 0, n_out_dev and out_indices are required");
  SPX_CHECK(n_in > 0, "static-shape rulebook needs n_in > 0 (pad the input with batch = -1 rows)");
  SPX_CHECK(pair_fwd && pair_bwd, "pair_fwd and pair_bwd are required");
  int kv = 1;
  for (int i = 0; i < ndim; ++i) kv *= ksize[i];
  spx::FillList pre;         // (as spx_conv_rulebook_static: the -1 fills of the outputs ride in the first fill launch)
  pre.add(out_indices, sizeof(int32_t) * static_cast<size_t>(n_out_cap) * (ndim + 1), 0xFFFFFFFFu);
  pre.add(pair_fwd, sizeof(int32_t) * static_cast<size_t>(kv) * n_out_cap, 0xFFFFFFFFu);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = spx::conv4_count_impl(indices, n_in, ndim, batch_size, in_shape, out_shape, ksize, stride, padding,
                                 dilation, rankmap, rankmap_bytes, ws, ws_bytes, nullptr, s, &pre, n_out_dev);
  if (rc) return rc;
  return spx::conv4_fill_impl(indices, n_in, ndim, batch_size, in_shape, out_shape, ksize, stride, padding, dilation,
                              n_out_cap, out_indices, pair_fwd, pair_bwd, mask_fwd, mask_bwd, pair_native,
                              num_per_loc, rankmap, rankmap_bytes, ws, ws_bytes, s, true, n_out_dev);
}

int spx_rankmap_from_sorted(const int32_t *indices, int n, int ndim, int batch_size, const int *spatial_shape,
                            void *rankmap, size_t rankmap_bytes, int32_t *violation, spx_stream_t stream) {
  using namespace spx;
  hipStream_t s = static_cast<hipStream_t>(stream);
  SPX_CHECK(ndim >= 1 && ndim <= kMaxNdim, "ndim must be in [1,4], got %d", ndim);
  const size_t W = rank_words(ndim, batch_size, spatial_shape);
  SPX_CHECK(W > 0 && rankmap && rankmap_bytes >= rank_bytes(W), "rank map missing or too small (%zu words)", W);
  SPX_CHECK(n >= 0 && (indices || n == 0), "indices required");
  const int one[4] = {1, 1, 1, 1}, zero[4] = {0, 0, 0, 0};
  const Geom g = make_geom(ndim, batch_size, spatial_shape, spatial_shape, one, one, zero, one);
  {
    FillList fills;                    // every word empty, every block offset zero (row = rank: the prefixes are global)
    fills.add(rankmap, rank_bytes(W), 0u);
    if (violation) fills.add(violation, sizeof(int32_t), 0u);
    SPX_HIP(fills.launch(s));
  }
  if (n > 0)
    hipLaunchKernelGGL(rankmap_from_sorted_kernel, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, indices, n, g,
                       static_cast<uint2 *>(rankmap), violation);
  SPX_LAUNCH_CHECK();
  return 0;
}

size_t spx_subm_rulebook_ranked_ws_bytes(int n, int kv) {
  const int nblk256 = spx::div_up(n > 0 ? n : 1, spx::kBlock);
  return spx::align_up(static_cast<size_t>(kv / 2 + 1) * nblk256 * sizeof(int32_t), 256) + 512;
}

int spx_subm_rulebook_ranked(const int32_t *indices, int n, int ndim, int batch_size, const int *spatial_shape,
                             const int *ksize, const int *dilation, int32_t *pair_fwd, int32_t *pair_bwd,
                             uint32_t *mask, int32_t *pair_native, int32_t *num_per_loc, const void *rankmap,
                             size_t rankmap_bytes, void *ws, size_t ws_bytes, spx_stream_t stream) {
  using namespace spx;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int padding[4], stride[4] = {1, 1, 1, 1}, kv = 1;
  SPX_CHECK(ndim >= 1 && ndim <= kMaxNdim, "ndim must be in [1,4], got %d", ndim);
  for (int i = 0; i < ndim; ++i) {
    SPX_CHECK(ksize[i] % 2 == 1, "subm only support odd ksize");  // indices.py:1650
    padding[i] = (ksize[i] / 2) * dilation[i];                    // indices.py:1652
    kv *= ksize[i];
  }
  if (check_geom(ndim, n, kv)) return -1;
  if (n == 0) {
    if (num_per_loc) SPX_HIP(hipMemsetAsync(num_per_loc, 0, sizeof(int32_t) * kv, s));
    return 0;
  }
  SPX_CHECK(pair_fwd && mask, "pair_fwd and mask are required");
  const int nblk256 = div_up(n, kBlock);
  SPX_CHECK(kv > 1 && kv <= 128 && nblk256 <= 16384, "ranked SubM build: 1 < kernel volume <= 128, <= 4 M rows");
  const size_t W = rank_words(ndim, batch_size, spatial_shape);
  SPX_CHECK(W > 0 && rankmap && rankmap_bytes >= rank_bytes(W), "rank map missing or too small (%zu words)", W);
  SPX_CHECK(ws && ws_bytes >= spx_subm_rulebook_ranked_ws_bytes(n, kv), "workspace too small");
  const Geom g = make_geom(ndim, batch_size, spatial_shape, spatial_shape, ksize, stride, padding, dilation);
  const int words = div_up(kv, 32);
  Carver cv(ws);
  int32_t *scratch_totals = cv.take<int32_t>(64);
  int32_t *groupcount = cv.take<int32_t>(static_cast<size_t>(kv / 2 + 1) * nblk256);
  const bool lists = pair_native || num_per_loc;
  // row-owned form from ~200 k rows (28 vs 35-37 us at 313-326 k rows); below, one thread per row is too few threads to
  // hide its lookups (19 vs 14 us at 77 k) and the probe form stays.  SPX_SUBM_RANK_ROWS = 1 / 0 forces / forbids.
  const int rows_opt = option_int("SPX_SUBM_RANK_ROWS", -1);
  if (rows_opt > 0 || (rows_opt < 0 && n >= 196608)) {
    // every entry of the tables and the masks is written by its row's thread -- no fill launch
    const size_t lds = static_cast<size_t>(kv) * (sizeof(int4) + sizeof(hkey_t) + sizeof(int));
    hipLaunchKernelGGL(subm_rank_rows_kernel, dim3(nblk256), dim3(kBlock), lds, s, indices, n, g,
                       static_cast<const uint2 *>(rankmap),
                       static_cast<const int32_t *>(rank_blockoff(const_cast<void *>(rankmap), W)), pair_fwd, pair_bwd,
                       mask, words, lists ? groupcount : nullptr, nblk256);
    if (lists) {
      SPX_CHECK(!pair_native || num_per_loc || kv / 2 <= 64, "num_per_loc required for kv > 128");
      hipLaunchKernelGGL(subm_lists_kernel, dim3(div_up(n, kItems), kv / 2 + 1), dim3(kBlock), 0, s, pair_fwd, kv, n,
                         nblk256, groupcount, pair_native, num_per_loc ? num_per_loc : scratch_totals,
                         num_per_loc ? kv : 0);
    }
    SPX_LAUNCH_CHECK();
    return 0;
  }
  // probe form (SPX_SUBM_RANK_ROWS = 0, for A/B runs): thread per (row, upper offset), mirror entries scattered.
  // masks by atomicOr, at every size: rows in key order keep a wave's mask words in a few lines (measured 34.5 vs 45.0 us
  // with the table pass at 313 k rows, 36.9 vs 48.0 at 326 k; the hash build of shuffled rows switches at 250 k)
  const int mp_opt = option_int("SPX_SUBM_MASK_PASS", -1);
  const int mask_pass = mp_opt < 0 ? 0 : mp_opt;
  {
    // what subm_insert_kernel writes on the hash path: the halves of the tables that only receive scattered mirror
    // entries start as -1, the masks (atomicOr targets without the mask pass) as 0
    FillList fills;
    fills.add(pair_fwd, sizeof(int32_t) * static_cast<size_t>(kv / 2) * n, 0xFFFFFFFFu);
    if (pair_bwd)
      fills.add(pair_bwd + static_cast<size_t>(kv / 2 + 1) * n, sizeof(int32_t) * static_cast<size_t>(kv - kv / 2 - 1) * n,
                0xFFFFFFFFu);
    if (!mask_pass) fills.add(mask, sizeof(uint32_t) * static_cast<size_t>(n) * words, 0u);
    SPX_HIP(fills.launch(s));
  }
  hipLaunchKernelGGL(subm_rank_probe_kernel, dim3(nblk256, kv / 2 + 1), dim3(kBlock), 0, s, indices, n, g,
                     static_cast<const uint2 *>(rankmap),
                     static_cast<const int32_t *>(rank_blockoff(const_cast<void *>(rankmap), W)), pair_fwd, pair_bwd, mask, words,
                     lists ? groupcount : nullptr, nblk256, mask_pass);
  if (mask_pass)
    hipLaunchKernelGGL(mask_from_table_kernel, dim3(nblk256), dim3(kBlock), 0, s, pair_fwd, kv, n, words, mask);
  if (lists) {
    SPX_CHECK(!pair_native || num_per_loc || kv / 2 <= 64, "num_per_loc required for kv > 128");
    hipLaunchKernelGGL(subm_lists_kernel, dim3(div_up(n, kItems), kv / 2 + 1), dim3(kBlock), 0, s, pair_fwd, kv, n,
                       nblk256, groupcount, pair_native, num_per_loc ? num_per_loc : scratch_totals,
                       num_per_loc ? kv : 0);
  }
  SPX_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

extern "C" {

size_t spx_mask_argsort_ws_bytes(int n) { return radix_argsort_ws_bytes(n) + 256; }

int spx_mask_argsort(const uint32_t *mask, int n, int words, int32_t *argsort, void *ws,
                     size_t ws_bytes, spx_stream_t stream) {
  SPX_CHECK(words == 1, "mask_argsort supports kernel volume <= 32 (words == 1), got %d", words);
  SPX_CHECK(ws_bytes >= spx_mask_argsort_ws_bytes(n), "workspace too small");
  if (n == 0) return 0;
  // stable argsort of the mask words (all.py:935-991 sorts the same keys with thrust)
  return radix_argsort(mask, n, 32, argsort, ws, static_cast<hipStream_t>(stream));
}

int spx_mask_argsort_kv(const uint32_t *mask, int n, int kv, int32_t *argsort, void *ws, size_t ws_bytes,
                        spx_stream_t stream) {
  SPX_CHECK(kv >= 1 && kv <= 32, "mask_argsort supports kernel volume <= 32, got %d", kv);
  SPX_CHECK(ws_bytes >= spx_mask_argsort_ws_bytes(n), "workspace too small");
  if (n == 0) return 0;
  // the mask words of a kernel volume kv carry kv bits: 27 bits are three 9-bit passes instead of four 8-bit ones
  return radix_argsort(mask, n, kv, argsort, ws, static_cast<hipStream_t>(stream));
}

int spx_native_to_table(const int32_t *pair_native, const int32_t *num_per_loc, int n_in,
                        int n_dst, int kv, int subm, int inverse, int32_t *table,
                        uint32_t *mask, spx_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  SPX_CHECK(pair_native && num_per_loc && table, "null pointer");
  if (n_dst == 0) return 0;
  SPX_HIP(hipMemsetAsync(table, 0xFF, sizeof(int32_t) * static_cast<size_t>(kv) * n_dst, s));
  if (n_in > 0)
    hipLaunchKernelGGL(native_to_table_kernel, dim3(div_up(n_in, kBlock), kv), dim3(kBlock), 0, s,
                       pair_native, num_per_loc, kv, n_in, n_dst, subm, inverse, table);
  if (mask)
    hipLaunchKernelGGL(mask_from_table_kernel, dim3(div_up(n_dst, kBlock)), dim3(kBlock), 0, s,
                       table, kv, n_dst, div_up(kv, 32), mask);
  SPX_LAUNCH_CHECK();
  return 0;
}

size_t spx_table_to_native_ws_bytes(int n, int kv) {
  const int nblk = div_up(n > 0 ? n : 1, kItems);
  return 2 * align_up(static_cast<size_t>(kv) * nblk * sizeof(int32_t), 256) + 256;
}

int spx_table_to_native(const int32_t *table, int subm, int kv, int n, int32_t *pair_native,
                        int32_t *num_per_loc, void *ws, size_t ws_bytes, spx_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  SPX_CHECK(table && pair_native && num_per_loc && ws, "null pointer");
  SPX_CHECK(ws_bytes >= spx_table_to_native_ws_bytes(n, kv), "workspace too small");
  SPX_HIP(hipMemsetAsync(num_per_loc, 0, sizeof(int32_t) * kv, s));
  if (n == 0) return 0;
  SPX_HIP(hipMemsetAsync(pair_native, 0xFF, sizeof(int32_t) * 2 * static_cast<size_t>(kv) * n, s));
  const int nblk = div_up(n, kItems);
  Carver cv(ws);
  int32_t *blockcount = cv.take<int32_t>(static_cast<size_t>(kv) * nblk);
  int32_t *blockoff = cv.take<int32_t>(static_cast<size_t>(kv) * nblk);
  if (subm)
    hipLaunchKernelGGL(subm_center_list_kernel, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s,
                       pair_native, kv, n);
  return launch_native_lists(table, subm ? 0 : 1, kv, n, subm ? kv / 2 : kv, nblk, blockcount,
                             blockoff, pair_native, num_per_loc, s);
}

size_t spx_point2voxel_ws_bytes(int n_points, int max_voxels) {
  return carve_p2v_ws(nullptr, n_points, max_voxels).bytes + 256;
}

int spx_point2voxel(const float *points, int n, int nfeat, int ndim, const float *vsize,
                    const float *coors_range, const int *grid_size, int max_voxels, int max_points,
                    int empty_mean, int clear_voxels, float *voxels, int32_t *indices,
                    int32_t *num_per_voxel, long long *pc_voxel_id, int *n_voxels_h, void *ws,
                    size_t ws_bytes, spx_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  SPX_CHECK(ndim >= 1 && ndim <= kMaxNdim, "ndim must be in [1,4], got %d", ndim);
  SPX_CHECK(nfeat >= ndim, "points need at least %d columns, got %d", ndim, nfeat);
  SPX_CHECK(max_voxels > 0 && max_points > 0 && n >= 0, "bad sizes");
  SPX_CHECK(voxels && indices && num_per_voxel && (pc_voxel_id || n == 0) && n_voxels_h, "null pointer");
  SPX_CHECK(ws_bytes >= spx_point2voxel_ws_bytes(n, max_voxels), "workspace too small");
  *n_voxels_h = 0;
  SPX_HIP(hipMemsetAsync(num_per_voxel, 0, sizeof(int32_t) * max_voxels, s));
  if (clear_voxels)
    SPX_HIP(hipMemsetAsync(voxels, 0, sizeof(float) * static_cast<size_t>(max_voxels) * max_points * nfeat, s));
  if (n == 0) return 0;
  SPX_CHECK(points, "null pointer");
  P2VGeom g;
  g.ndim = ndim;
  for (int j = 0; j < 4; ++j) {
    g.vsize[j] = j < ndim ? vsize[j] : 1.f;
    g.lo[j] = j < ndim ? coors_range[j] : 0.f;
    g.grid[j] = j < ndim ? grid_size[j] : 1;
  }
  P2VWs w = carve_p2v_ws(ws, n, max_voxels, keys_fit_u32(1, g.grid, 4));
  const size_t cap = static_cast<size_t>(w.t.mask) + 1;
  SPX_HIP(table_clear(w.t, s));
  SPX_HIP(hipMemsetAsync(w.slot_vid, 0xFF, sizeof(int32_t) * cap, s));
  const dim3 gp(div_up(n, kBlock));
  hipLaunchKernelGGL(p2v_insert_kernel, gp, dim3(kBlock), 0, s, points, n, nfeat, g, w.t, w.slot_of);
  hipLaunchKernelGGL(p2v_count_first_kernel, dim3(w.nblk), dim3(kBlock), 0, s, w.slot_of, w.t, n,
                     w.blockcount);
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(kBlock), 0, s, w.blockcount, w.blockoff, w.nblk, w.total);
  hipLaunchKernelGGL(p2v_clamp_count_kernel, dim3(1), dim3(1), 0, s, w.total, max_voxels, w.n_voxels);
  hipLaunchKernelGGL(p2v_assign_kernel, dim3(w.nblk), dim3(kBlock), 0, s, points, n, nfeat, g, w.slot_of,
                     w.t, w.blockoff, max_voxels, w.slot_vid, indices);
  hipLaunchKernelGGL(p2v_point_vid_kernel, gp, dim3(kBlock), 0, s, w.slot_of, w.slot_vid, n, pc_voxel_id,
                     w.key32);
  // stable sort of the points by voxel id (4 x 8-bit LSD passes, as mask_argsort)
  const uint32_t *kin = w.key32;
  const int32_t *vin = nullptr;
  uint32_t *kout[4] = {w.kB, w.kA, w.kB, w.kA};
  int32_t *vout[4] = {w.vB, w.order, w.vB, w.order};
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(radix_count_kernel, dim3(w.nblk), dim3(kBlock), 0, s, kin, n, pass * kRadixBits,
                       w.nblk, w.hist);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(kBlock), 0, s, w.hist, w.hist_off, kRadix * w.nblk,
                       static_cast<int32_t *>(nullptr));
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(w.nblk), dim3(kBlock), 0, s, kin, vin, n,
                       pass * kRadixBits, w.nblk, w.hist_off, kout[pass], vout[pass]);
    kin = kout[pass];
    vin = vout[pass];
  }
  hipLaunchKernelGGL(p2v_segment_kernel, gp, dim3(kBlock), 0, s, w.kA, n, w.seg_start);
  hipLaunchKernelGGL(p2v_scatter_kernel, gp, dim3(kBlock), 0, s, points, nfeat, w.kA, w.order, n,
                     w.seg_start, max_points, voxels, num_per_voxel);
  if (empty_mean == 2) {
    SPX_CHECK(nfeat <= 1024, "reference-quirk mean fill: at most 1024 point features");
    hipLaunchKernelGGL(p2v_mean_carry_kernel, dim3(1), dim3(((nfeat + 63) / 64) * 64), 0, s, voxels, num_per_voxel,
                       w.n_voxels, max_points, nfeat);
  } else if (empty_mean) {
    const long long total = static_cast<long long>(max_voxels) * nfeat;
    hipLaunchKernelGGL(p2v_mean_kernel, dim3(static_cast<unsigned>((total + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, voxels, num_per_voxel, w.n_voxels, max_points, nfeat);
  }
  SPX_LAUNCH_CHECK();
  int32_t host_n = 0;
  SPX_HIP(hipMemcpyAsync(&host_n, w.n_voxels, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  SPX_HIP(hipStreamSynchronize(s));
  *n_voxels_h = host_n;
  return 0;
}

/* what: 0 insert, 1 query, 2 insert_exist_keys, 3 assign_arange, 4 items */
static int user_hash_call(int what, void *tkeys, void *tvals, int cap, int key_bytes, int val_bytes,
                          const void *keys, void *values, unsigned char *is_empty, int n, void *keys_out,
                          void *vals_out, int max_out, void *count_out, void *ws, size_t ws_bytes,
                          spx_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  SPX_CHECK(tkeys && tvals && cap > 0, "hash table storage required");
  SPX_CHECK((key_bytes == 4 || key_bytes == 8) && (val_bytes == 4 || val_bytes == 8),
            "keys and values must be 4 or 8 bytes wide");
  if (what >= 3) SPX_CHECK(ws && ws_bytes >= spx_hash_ws_bytes(cap), "workspace too small");
#define SPX_UH(KT, VT) \
  return user_hash_dispatch2<KT, VT>(what, tkeys, tvals, cap, keys, values, is_empty, n, keys_out, vals_out, \
                                     max_out, count_out, ws, s)
  if (key_bytes == 4 && val_bytes == 4) SPX_UH(uint32_t, uint32_t);
  if (key_bytes == 4 && val_bytes == 8) SPX_UH(uint32_t, unsigned long long);
  if (key_bytes == 8 && val_bytes == 4) SPX_UH(unsigned long long, uint32_t);
  SPX_UH(unsigned long long, unsigned long long);
#undef SPX_UH
}

size_t spx_hash_ws_bytes(int capacity) {
  const int nblk = div_up(capacity > 0 ? capacity : 1, kItems);
  return 2 * align_up(static_cast<size_t>(nblk) * sizeof(int32_t), 256) + 512;
}

int spx_hash_clear(void *table_keys, int capacity, int key_bytes, spx_stream_t stream) {
  SPX_CHECK(table_keys && capacity > 0 && (key_bytes == 4 || key_bytes == 8), "bad hash table");
  SPX_HIP(hipMemsetAsync(table_keys, 0xFF, static_cast<size_t>(capacity) * key_bytes,
                         static_cast<hipStream_t>(stream)));
  return 0;
}

int spx_hash_insert(void *table_keys, void *table_vals, int capacity, int key_bytes, int val_bytes,
                    const void *keys, const void *values, int n, spx_stream_t stream) {
  return user_hash_call(0, table_keys, table_vals, capacity, key_bytes, val_bytes, keys,
                        const_cast<void *>(values), nullptr, n, nullptr, nullptr, 0, nullptr, nullptr, 0, stream);
}

int spx_hash_query(void *table_keys, void *table_vals, int capacity, int key_bytes, int val_bytes,
                   const void *keys, void *values_out, unsigned char *is_empty, int n, spx_stream_t stream) {
  return user_hash_call(1, table_keys, table_vals, capacity, key_bytes, val_bytes, keys, values_out, is_empty,
                        n, nullptr, nullptr, 0, nullptr, nullptr, 0, stream);
}

int spx_hash_insert_exist(void *table_keys, void *table_vals, int capacity, int key_bytes, int val_bytes,
                          const void *keys, const void *values, unsigned char *is_empty, int n,
                          spx_stream_t stream) {
  return user_hash_call(2, table_keys, table_vals, capacity, key_bytes, val_bytes, keys,
                        const_cast<void *>(values), is_empty, n, nullptr, nullptr, 0, nullptr, nullptr, 0, stream);
}

int spx_hash_assign_arange(void *table_keys, void *table_vals, int capacity, int key_bytes, int val_bytes,
                           void *count_out, void *ws, size_t ws_bytes, spx_stream_t stream) {
  return user_hash_call(3, table_keys, table_vals, capacity, key_bytes, val_bytes, nullptr, nullptr, nullptr, 0,
                        nullptr, nullptr, 0, count_out, ws, ws_bytes, stream);
}

int spx_hash_items(void *table_keys, void *table_vals, int capacity, int key_bytes, int val_bytes,
                   void *keys_out, void *vals_out, int max_out, void *count_out, void *ws, size_t ws_bytes,
                   spx_stream_t stream) {
  return user_hash_call(4, table_keys, table_vals, capacity, key_bytes, val_bytes, nullptr, nullptr, nullptr, 0,
                        keys_out, vals_out, max_out, count_out, ws, ws_bytes, stream);
}

}  // extern "C"
