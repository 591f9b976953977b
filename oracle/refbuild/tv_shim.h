// Minimal stand-ins for the cumm / tensorview types the reference's CPU rulebook code is written
// against.  TEST INFRASTRUCTURE: used only to compile the reference's own function bodies
// (rendered by oracle/refbuild/render.py into oracle/_ref/ref_indices.cpp) into oracle/_ref/.
//
// cumm (PyPI `cumm`, pinned >=0.7.11,<0.8.0 by the reference's setup.py:42-44) is NOT under
// /root/reference, so its headers cannot be compiled from where they lie.  The few types below are
// restated from their documented behaviour -- all of it plain row-major index arithmetic:
//   tv::array<T, N>                 fixed-size array with element-wise helpers (`op<prod>`)
//   tv::Tensor                      a non-owning view: data pointer + shape (`dim`, `stride`, `data_ptr<T>`, `zero_`,
//                                   `slice_first_axis`, `tview<T, N>()` = row-major element accessor)
//   ConvProblem<ND>                 the convolution geometry record (cumm/conv/params.py)
//   TensorGeneric<ND, Index>        row-major layout: operator() = sum idx[i] * stride[i],
//                                   inverse() = successive div/mod (cumm/gemm/layout.py)
// The reference call sites that fix these semantics: indices.py:86-111 (layouts built with
// from_shape of {N, output dims} / {ksize}), :1667-1671 and :1741 (layout_npq applied to an index
// row / an offset array), :136 (layout_rs.inverse), :1660 (check_npq_not_overflow).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <tuple>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <initializer_list>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#define TV_HOST_DEVICE_INLINE inline

namespace tv {

namespace arrayops {
struct prod {};
}  // namespace arrayops

template <typename T, size_t N> struct array {
  T v_[N];
  array() : v_{} {}
  array(std::initializer_list<T> l) : v_{} {
    size_t i = 0;
    for (const T &x : l) {
      if (i < N) v_[i++] = x;
    }
  }
  T &operator[](size_t i) { return v_[i]; }
  const T &operator[](size_t i) const { return v_[i]; }
  T *data() { return v_; }
  const T *data() const { return v_; }
  template <typename Op> T op() const {            // only arrayops::prod is used by the reference
    T p = T(1);
    for (size_t i = 0; i < N; ++i) p *= v_[i];
    return p;
  }
};

// row-major element accessor of a view (tensorview's TensorView<T, N>::operator())
template <typename T, int N> struct TensorViewN {
  T *ptr;
  int64_t shape[N];
  template <typename... I> T &operator()(I... idx) const {
    static_assert(sizeof...(I) == N, "one index per axis");
    const int64_t a[N] = {static_cast<int64_t>(idx)...};
    int64_t off = 0;
    for (int k = 0; k < N; ++k) off = off * shape[k] + a[k];
    return ptr[off];
  }
};

class Tensor {
 public:
  Tensor() : ptr_(nullptr), itemsize_(4) {}
  Tensor(void *p, std::vector<int64_t> shape, int itemsize = 4)
      : ptr_(p), shape_(std::move(shape)), itemsize_(itemsize) {}
  int64_t dim(int i) const { return shape_[i]; }
  int ndim() const { return static_cast<int>(shape_.size()); }
  int64_t stride(int i) const {        // contiguous row-major views only
    int64_t s = 1;
    for (int k = ndim() - 1; k > i; --k) s *= shape_[k];
    return s;
  }
  int64_t size() const {
    int64_t s = 1;
    for (int64_t d : shape_) s *= d;
    return s;
  }
  int dtype() const { return 0; }      // (fp32 views only, see tv::dispatch below)
  int device() const { return -1; }
  template <typename T> T *data_ptr() { return static_cast<T *>(ptr_); }
  template <typename T> T *data_ptr() const { return static_cast<T *>(ptr_); }
  Tensor &zero_() {
    std::memset(ptr_, 0, static_cast<size_t>(size()) * itemsize_);
    return *this;
  }
  Tensor slice_first_axis(int64_t begin, int64_t end) const {
    std::vector<int64_t> sh = shape_;
    sh[0] = end - begin;
    return Tensor(static_cast<char *>(ptr_) + begin * stride(0) * itemsize_, sh, itemsize_);
  }
  template <typename T, int N> TensorViewN<T, N> tview() const {
    TensorViewN<T, N> v;
    v.ptr = static_cast<T *>(ptr_);
    for (int k = 0; k < N; ++k) v.shape[k] = shape_[k];
    return v;
  }

 private:
  void *ptr_;
  std::vector<int64_t> shape_;
  int itemsize_;
};

inline Tensor from_blob(void *p, std::initializer_list<int64_t> shape, int itemsize = 4) {
  return Tensor(p, std::vector<int64_t>(shape), itemsize);
}

// gather.py's element-type dispatch and 1-d loop helper: the oracle drives the reference's gather /
// scatter-add with fp32 rows only, so `dispatch` always takes the float branch; `kernel_1d` is the
// serial form (the published CPU wheel has no OpenMP, README.md:133): one call over [0, n), step 1.
struct half_t {};
struct bfloat16_t {};
template <typename... Ts, typename F> inline void dispatch(int /*dtype*/, F &&f) { f(float{}); }
template <typename F> inline void kernel_1d(int /*device*/, int64_t n, F &&f) { f(0, static_cast<int>(n), 1); }

template <typename... Ts> inline std::string ssprint(const Ts &...xs) {
  std::ostringstream ss;
  (void)std::initializer_list<int>{((ss << xs << ' '), 0)...};
  return ss.str();
}

}  // namespace tv

#define TV_DECLTYPE(x) std::decay_t<decltype(x)>
#define TV_IF_CONSTEXPR constexpr
#define TV_ASSERT_RT_ERR(cond, ...)                                                   \
  do {                                                                                \
    if (!(cond)) throw std::runtime_error(std::string(#cond " failed: ") + tv::ssprint(__VA_ARGS__)); \
  } while (0)

namespace tvshim {

template <int ND> struct ConvProblem {
  int N, C, K;
  tv::array<int, ND> input_dims, output_dims, ksize, padding, stride, dilation;
  ConvProblem(int n, int c, int k, tv::array<int, ND> in, tv::array<int, ND> out, tv::array<int, ND> ks,
              tv::array<int, ND> pad, tv::array<int, ND> st, tv::array<int, ND> dil)
      : N(n), C(c), K(k), input_dims(in), output_dims(out), ksize(ks), padding(pad), stride(st),
        dilation(dil) {}
  // true when N * prod(output_dims) is addressable with 32-bit keys
  bool check_npq_not_overflow() const {
    int64_t v = N;
    for (int i = 0; i < ND; ++i) v *= output_dims[i];
    return v >= 0 && v <= std::numeric_limits<int32_t>::max();
  }
};

template <int ND, typename Index> struct TensorGeneric {
  Index strides[ND];
  static TensorGeneric from_shape(const tv::array<int, ND> &shape) {
    TensorGeneric l;
    Index s = 1;
    for (int i = ND - 1; i >= 0; --i) {
      l.strides[i] = s;
      s *= static_cast<Index>(shape[i]);
    }
    return l;
  }
  Index operator()(const int *idx) const {
    Index r = 0;
    for (int i = 0; i < ND; ++i) r += static_cast<Index>(idx[i]) * strides[i];
    return r;
  }
  Index operator()(const tv::array<int, ND> &idx) const { return (*this)(idx.data()); }
  void inverse(Index index, tv::array<int, ND> &out) const {
    for (int i = 0; i < ND; ++i) {
      out[i] = static_cast<int>(index / strides[i]);
      index -= static_cast<Index>(out[i]) * strides[i];
    }
  }
};

}  // namespace tvshim
