// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.
//
// A CPU restatement of the reference's ConvAlgo.Native CPU path for the
// rulebook ("indice pairs") and the gather / scatter-add helpers.  Only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// this library; the product (spconv_amd/) never does.
//
// The reference (traveller59/spconv v2.3.8) cannot be built with its own build system here: its
// C++ is assembled at import time by `pccm` and includes headers of the unvendored `cumm` package
// (SURVEY.md section 8c).  Every function below therefore restates the generated code
// operation-for-operation and cites the generator lines it follows.
//
// Parity status -- PINNED:
//  * rulebook (pairs, counts, output coordinates, their ORDER): pinned by executing the
//    reference's own CPU generators.  oracle/refbuild/render.py loads
//    /root/reference/spconv/csrc/sparse/indices.py where it lies, collects the reference's C++
//    text of ConvOutLocIter / SparseConvIndicesCPU and compiles it (against small stand-ins for
//    the cumm value types, refbuild/tv_shim.h) into oracle/_ref/libspconv_ref.so.  This
//    restatement equals that library bit-for-bit on every committed vector
//    (tests/golden/ref_*.npz, ref_digests.json incl. the real-LiDAR fixture and BASELINE
//    configs 1-3) and on a randomised 1-d..4-d sweep (tests/test_oracle.py);
//  * numerical results of the path: pinned by the reference's own test oracle (dense torch
//    conv3d, test/test_conv.py:247-357, re-run in tests/test_oracle.py) and its numpy per-offset
//    formula (test/test_all_algo.py:222-288).
//
// Build: see oracle/Makefile (g++ -O3 -shared -fPIC).

#include <cmath>
#include <cstdint>
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <limits>
#include <unordered_map>
#include <vector>

namespace {

constexpr int kMaxNdim = 4;

// Coordinate algebra of ConvOutLocIter (spconv/csrc/sparse/indices.py:77-269).
struct LocIter {
  int ndim;
  int batch;                 // problem_.N
  int in_dims[kMaxNdim];     // problem_.input_dims
  int out_dims[kMaxNdim];    // problem_.output_dims
  int ksize[kMaxNdim], stride[kMaxNdim], padding[kMaxNdim], dilation[kMaxNdim];
  int count[kMaxNdim];       // current filter offset (r0, r1, ...), last dim fastest

  // operator++ (indices.py:114-127): odometer, last spatial dim fastest.
  void next() {
    for (int i = ndim - 1; i >= 0; --i) {
      if (++count[i] < ksize[i]) return;
      count[i] = 0;
    }
  }
  // layout_npq (indices.py:84-90,108-109): row-major over (batch, out dims).
  template <typename K> K layout_npq(const int *c) const {
    K v = c[0];
    for (int i = 0; i < ndim; ++i) v = v * K(out_dims[i]) + K(c[i + 1]);
    return v;
  }
  // query_npq_no_stride (indices.py:205-225, nhw_to_npq<true> :141-155).
  bool query_npq_no_stride(const int *nhw, int *npq) const {
    npq[0] = nhw[0];
    bool ok = nhw[0] < batch && nhw[0] >= 0;
    for (int i = 0; i < ndim; ++i) {
      npq[i + 1] = nhw[i + 1] + padding[i] - count[i] * dilation[i];
      ok = ok && npq[i + 1] >= 0 && npq[i + 1] < out_dims[i];
    }
    return ok;
  }
  // query_npq (indices.py:174-203): C++ truncating '/' and '%'.
  bool query_npq(const int *nhw, int *npq) const {
    npq[0] = nhw[0];
    bool ok = nhw[0] < batch && nhw[0] >= 0;
    for (int i = 0; i < ndim; ++i) {
      int h = nhw[i + 1] + padding[i] - count[i] * dilation[i];
      npq[i + 1] = h / stride[i];
      ok = ok && npq[i + 1] >= 0 && npq[i + 1] < out_dims[i] && !(h % stride[i]);
    }
    return ok;
  }
  // query_nhw_out (indices.py:249-269, npq_to_nhw :157-172): transposed conv.
  bool query_nhw_out(const int *npq, int *nhw) const {
    nhw[0] = npq[0];
    bool ok = npq[0] < batch && npq[0] >= 0;
    for (int i = 0; i < ndim; ++i) {
      nhw[i + 1] = npq[i + 1] * stride[i] - padding[i] + count[i] * dilation[i];
      ok = ok && nhw[i + 1] >= 0 && nhw[i + 1] < out_dims[i];
    }
    return ok;
  }
};

LocIter make_iter(int ndim, int batch, const int *in_dims, const int *out_dims,
                  const int *ksize, const int *stride, const int *padding,
                  const int *dilation) {
  LocIter it{};
  it.ndim = ndim;
  it.batch = batch;
  for (int i = 0; i < ndim; ++i) {
    it.in_dims[i] = in_dims[i];
    it.out_dims[i] = out_dims[i];
    it.ksize[i] = ksize[i];
    it.stride[i] = stride[i];
    it.padding[i] = padding[i];
    it.dilation[i] = dilation[i];
    it.count[i] = 0;
  }
  return it;
}

// ConvProblem::check_npq_not_overflow (cumm, call sites indices.py:1658,1723;
// python twin ops.py:188-190): int32 keys iff batch * prod(out dims) < 2^31-1.
bool key_fits_int32(int ndim, int batch, const int *out_dims) {
  int64_t v = batch;
  for (int i = 0; i < ndim; ++i) v *= out_dims[i];
  return v < int64_t(std::numeric_limits<int32_t>::max());
}

// SparseConvIndicesCPU::generate_subm_conv_inds (indices.py:1639-1708).
template <typename K>
int subm_rulebook(LocIter it, const int32_t *indices, int n, int32_t *pairs,
                  int pair_size, int32_t *num_per_loc) {
  const int ndim = it.ndim;
  int kv = 1;
  for (int i = 0; i < ndim; ++i) kv *= it.ksize[i];
  const int pair_size_mul_kv = pair_size * kv;
  std::unordered_map<K, int32_t> hash;
  const int32_t *p = indices;
  for (int i = 0; i < n; ++i) {              // :1670-1674, first insert wins
    hash.insert({it.layout_npq<K>(p), i});
    p += ndim + 1;
  }
  for (int fo = 0; fo < kv / 2 + 1; ++fo) {  // :1675
    const int off = fo * pair_size;
    const int off_1 = (kv - 1 - fo) * pair_size;
    if (fo == kv / 2) {                      // :1678-1682 centre = identity
      for (int i = 0; i < n; ++i) {
        pairs[off + i] = i;
        pairs[pair_size_mul_kv + off + i] = i;
      }
    } else {
      p = indices;
      int32_t *cnt = num_per_loc + fo;
      for (int i = 0; i < n; ++i) {          // :1686-1700
        int npq[kMaxNdim + 1];
        if (it.query_npq_no_stride(p, npq)) {
          auto iter = hash.find(it.layout_npq<K>(npq));
          if (iter != hash.end()) {
            int old_num = cnt[0]++;
            pairs[off + old_num] = i;
            pairs[pair_size_mul_kv + off + old_num] = iter->second;
            pairs[off_1 + old_num] = iter->second;
            pairs[pair_size_mul_kv + off_1 + old_num] = i;
          }
        }
        p += ndim + 1;
      }
    }
    it.next();
  }
  return n;
}

// SparseConvIndicesCPU::generate_conv_inds (indices.py:1710-1778).
template <typename K>
int conv_rulebook(LocIter it, const int32_t *indices, int n, int transposed,
                  int32_t *pairs, int pair_size, int32_t *out_inds,
                  int32_t *num_per_loc) {
  const int ndim = it.ndim;
  int kv = 1;
  for (int i = 0; i < ndim; ++i) kv *= it.ksize[i];
  const int pair_size_mul_kv = pair_size * kv;
  std::unordered_map<K, int32_t> hash;
  int num_act = 0;
  int32_t *out_p = out_inds;
  for (int fo = 0; fo < kv; ++fo) {          // :1742 k-major
    const int off = fo * pair_size;
    const int32_t *p = indices;
    int32_t *cnt = num_per_loc + fo;
    for (int i = 0; i < n; ++i) {            // :1746 then input-major
      int npq[kMaxNdim + 1];
      bool valid = transposed ? it.query_nhw_out(p, npq) : it.query_npq(p, npq);
      if (valid) {
        K index = it.layout_npq<K>(npq);
        auto iter = hash.find(index);
        int32_t hashval;
        if (iter == hash.end()) {            // :1757-1763 first-seen numbering
          hashval = num_act++;
          hash.insert({index, hashval});
          for (int k = 0; k < ndim + 1; ++k) out_p[k] = npq[k];
          out_p += ndim + 1;
        } else {
          hashval = iter->second;
        }
        pairs[off + cnt[0]] = i;             // :1767-1768
        pairs[pair_size_mul_kv + off + cnt[0]++] = hashval;
      }
      p += ndim + 1;
    }
    it.next();
  }
  return num_act;
}

}  // namespace

extern "C" {

// get_conv_output_size / get_deconv_output_size (spconv/pytorch/ops.py:73-96).
// Python floor division: sizes here are >= 0 whenever the result is used.
void orc_conv_out_shape(int ndim, const int *in, const int *ksize, const int *stride,
                        const int *padding, const int *dilation, const int *out_padding,
                        int transposed, int *out) {
  for (int i = 0; i < ndim; ++i) {
    if (transposed) {
      out[i] = (in[i] - 1) * stride[i] - 2 * padding[i] + ksize[i] + out_padding[i];
    } else {
      int num = in[i] + 2 * padding[i] - dilation[i] * (ksize[i] - 1) - 1;
      int q = num / stride[i];
      if ((num % stride[i] != 0) && ((num < 0) != (stride[i] < 0))) --q;  // floor
      out[i] = (ksize[i] == -1) ? 1 : q + 1;
    }
  }
}

// Returns n (the number of outputs) or -1 on a bad argument (even ksize:
// "subm only support odd ksize", indices.py:1650).
// pairs is [2, kv, pair_size] pre-filled with -1, num_per_loc [kv] zeros
// (SpconvOps.get_indice_pairs, all.py:2071-2076).
int orc_subm_rulebook(const int32_t *indices, int n, int ndim, int batch,
                      const int *dims, const int *ksize, const int *dilation,
                      int32_t *pairs, int pair_size, int32_t *num_per_loc) {
  if (ndim < 1 || ndim > kMaxNdim) return -1;
  int stride[kMaxNdim], padding[kMaxNdim];
  for (int i = 0; i < ndim; ++i) {
    if (ksize[i] % 2 != 1) return -1;
    stride[i] = 1;
    padding[i] = (ksize[i] / 2) * dilation[i];   // indices.py:1651-1652
  }
  LocIter it = make_iter(ndim, batch, dims, dims, ksize, stride, padding, dilation);
  if (key_fits_int32(ndim, batch, dims))
    return subm_rulebook<int32_t>(it, indices, n, pairs, pair_size, num_per_loc);
  return subm_rulebook<int64_t>(it, indices, n, pairs, pair_size, num_per_loc);
}

// Returns num_act_out.  out_inds must hold kv*n rows (all.py:2121-2122).
int orc_conv_rulebook(const int32_t *indices, int n, int ndim, int batch,
                      const int *out_dims, const int *in_dims, const int *ksize,
                      const int *stride, const int *padding, const int *dilation,
                      int transposed, int32_t *pairs, int pair_size,
                      int32_t *out_inds, int32_t *num_per_loc) {
  if (ndim < 1 || ndim > kMaxNdim) return -1;
  LocIter it = make_iter(ndim, batch, in_dims, out_dims, ksize, stride, padding, dilation);
  if (key_fits_int32(ndim, batch, out_dims))
    return conv_rulebook<int32_t>(it, indices, n, transposed, pairs, pair_size,
                                  out_inds, num_per_loc);
  return conv_rulebook<int64_t>(it, indices, n, transposed, pairs, pair_size,
                                out_inds, num_per_loc);
}

// GatherCPU::gather (spconv/csrc/sparse/gather.py:30-53): row memcpy.
void orc_gather(void *out, const void *in, const int32_t *inds, int nhot,
                int channel, int elem_bytes) {
  const size_t row = size_t(channel) * elem_bytes;
  for (int i = 0; i < nhot; ++i)
    std::memcpy(static_cast<char *>(out) + i * row,
                static_cast<const char *>(in) + size_t(inds[i]) * row, row);
}

// GatherCPU::scatter_add (gather.py:55-86): out[inds[i]] += in[i], serial.
void orc_scatter_add_f32(float *out, const float *in, const int32_t *inds,
                         int nhot, int channel) {
  for (int i = 0; i < nhot; ++i) {
    const float *buf = in + size_t(i) * channel;
    float *o = out + size_t(inds[i]) * channel;
    for (int j = 0; j < channel; ++j) o[j] = o[j] + buf[j];
  }
}

void orc_scatter_add_f64(double *out, const double *in, const int32_t *inds,
                         int nhot, int channel) {
  for (int i = 0; i < nhot; ++i) {
    const double *buf = in + size_t(i) * channel;
    double *o = out + size_t(inds[i]) * channel;
    for (int j = 0; j < channel; ++j) o[j] = o[j] + buf[j];
  }
}

// Thread count of the faithful-omp variants below (bench.py's cpu_baseline leg).
void orc_set_omp_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}

// "faithful-omp" variants (what a source CUMM_CPU_ONLY_BUILD with -fopenmp
// does, gather.py:25-26,47-53): rows in parallel.  Used only by bench.py's
// cpu_baseline leg; compiled with -fopenmp.
void orc_gather_omp(void *out, const void *in, const int32_t *inds, int nhot,
                    int channel, int elem_bytes) {
  const size_t row = size_t(channel) * elem_bytes;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nhot; ++i)
    std::memcpy(static_cast<char *>(out) + i * row,
                static_cast<const char *>(in) + size_t(inds[i]) * row, row);
}

// Within one filter offset every output row appears at most once (distinct
// coordinates), so rows can be updated in parallel without a race.
void orc_scatter_add_f32_omp(float *out, const float *in, const int32_t *inds,
                             int nhot, int channel) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nhot; ++i) {
    const float *buf = in + size_t(i) * channel;
    float *o = out + size_t(inds[i]) * channel;
    for (int j = 0; j < channel; ++j) o[j] = o[j] + buf[j];
  }
}

}  // extern "C"

// Point2VoxelCPU::point_to_voxel (spconv/csrc/sparse/pointops.py, lines 97-196 of the class;
// zyx = true): sequential, first-seen voxel numbering, first max_points points per voxel.
// mean fill: empty_mean = 1 the arithmetic mean of the voxel's points (what the reference's loop means to do and
// the product's default), empty_mean = 2 the loop as it behaves (accumulator carried from voxel to voxel, below).
// Both forms are checked against the reference's own code, executed (oracle/_ref, tests/golden/p2v_ref.npz).
extern "C" int orc_point2voxel(const float *points, int n, int nfeat, int ndim, const float *vsize,
                               const float *coors_range, const int *grid_size, int max_voxels,
                               int max_points, int empty_mean, float *voxels, int32_t *indices,
                               int32_t *num_per_voxel, int64_t *pc_voxel_id) {
  std::unordered_map<int64_t, int> coor_to_voxelidx;   // stands in for the dense grid table
  int voxel_num = 0;
  for (int i = 0; i < n; ++i) {
    int coor[4];
    bool failed = false;
    for (int j = 0; j < ndim; ++j) {
      const float v = std::floor((points[static_cast<size_t>(i) * nfeat + (ndim - 1 - j)] - coors_range[j]) / vsize[j]);
      if (v < 0 || v >= static_cast<float>(grid_size[j])) {
        failed = true;
        break;
      }
      coor[j] = static_cast<int>(v);
    }
    if (failed) {
      pc_voxel_id[i] = -1;
      continue;
    }
    int64_t key = 0;
    for (int j = 0; j < ndim; ++j) key = key * grid_size[j] + coor[j];
    auto it = coor_to_voxelidx.find(key);
    int voxelidx;
    if (it == coor_to_voxelidx.end()) {
      voxelidx = voxel_num;
      if (voxel_num >= max_voxels) {
        pc_voxel_id[i] = -1;
        continue;
      }
      voxel_num += 1;
      coor_to_voxelidx.emplace(key, voxelidx);
      for (int k = 0; k < ndim; ++k) indices[static_cast<size_t>(voxelidx) * ndim + k] = coor[k];
    } else {
      voxelidx = it->second;
    }
    pc_voxel_id[i] = voxelidx;
    const int num = num_per_voxel[voxelidx];
    if (num < max_points) {
      for (int k = 0; k < nfeat; ++k)
        voxels[(static_cast<size_t>(voxelidx) * max_points + num) * nfeat + k] = points[static_cast<size_t>(i) * nfeat + k];
      num_per_voxel[voxelidx] += 1;
    }
  }
  if (empty_mean == 2) {
    // the reference's loop AS IT BEHAVES (pointops.py:663-686): `mean_value.clear()` does not zero the
    // accumulator (it only resets the vector's size; the elements keep their values and operator[] does not
    // check bounds), so voxel v starts from the MEAN of voxel v - 1: mean_v = (mean_{v-1} + sum_j x_j) / num_v,
    // accumulated point by point in fp32.  Opt-in (SPCONV_AMD_REFERENCE_QUIRKS=1 in the product).
    std::vector<float> carry(nfeat, 0.f);
    for (int v = 0; v < voxel_num; ++v) {
      const int num = num_per_voxel[v];
      if (num <= 0) continue;
      for (int j = 0; j < num; ++j)
        for (int k = 0; k < nfeat; ++k) carry[k] += voxels[(static_cast<size_t>(v) * max_points + j) * nfeat + k];
      for (int k = 0; k < nfeat; ++k) carry[k] /= num;
      for (int j = num; j < max_points; ++j)
        for (int k = 0; k < nfeat; ++k) voxels[(static_cast<size_t>(v) * max_points + j) * nfeat + k] = carry[k];
    }
  } else if (empty_mean) {
    for (int v = 0; v < voxel_num; ++v) {
      const int num = num_per_voxel[v];
      if (num <= 0) continue;
      for (int k = 0; k < nfeat; ++k) {
        float sum = 0.f;
        for (int j = 0; j < num; ++j) sum += voxels[(static_cast<size_t>(v) * max_points + j) * nfeat + k];
        const float mean = sum / static_cast<float>(num);
        for (int j = num; j < max_points; ++j) voxels[(static_cast<size_t>(v) * max_points + j) * nfeat + k] = mean;
      }
    }
  }
  return voxel_num;
}
