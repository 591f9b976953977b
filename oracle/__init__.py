"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by spconv_amd/).

Python half of the CPU restatement of the reference's ``ConvAlgo.Native`` CPU
path (traveller59/spconv v2.3.8).  The rulebook loops and the gather /
scatter-add helpers live in ``oracle.cpp`` (C++, ``std::unordered_map`` like the
reference); the per-offset ``torch.mm`` driver loops live here because in the
reference they are Python too (``spconv/pytorch/ops.py:888-988,1164-1253``).

Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.

Parity status: numerics pinned by the reference's own test
oracle -- dense ``torch.nn.functional.conv3d`` (``test/test_conv.py:247-357``)
and its numpy per-offset formula (``test/test_all_algo.py:222-288``); rulebook
*order* is "parity unpinned" by the reference's tests and is defined by the
restated CPU loops.
"""
from __future__ import annotations

import ctypes
import functools
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_I32P = ctypes.POINTER(ctypes.c_int32)
_IP = ctypes.POINTER(ctypes.c_int)


def build(force: bool = False) -> str:
    """Compile oracle.cpp with g++ (see oracle/Makefile)."""
    src = os.path.join(_HERE, "oracle.cpp")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


@functools.lru_cache(maxsize=None)
def lib() -> ctypes.CDLL:
    build()
    L = ctypes.CDLL(_LIB_PATH)
    L.orc_conv_out_shape.argtypes = [ctypes.c_int] + [_IP] * 6 + [ctypes.c_int, _IP]
    L.orc_conv_out_shape.restype = None
    L.orc_subm_rulebook.argtypes = [_I32P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    _IP, _IP, _IP, _I32P, ctypes.c_int, _I32P]
    L.orc_subm_rulebook.restype = ctypes.c_int
    L.orc_conv_rulebook.argtypes = [_I32P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    _IP, _IP, _IP, _IP, _IP, _IP, ctypes.c_int,
                                    _I32P, ctypes.c_int, _I32P, _I32P]
    L.orc_conv_rulebook.restype = ctypes.c_int
    for name in ("orc_gather", "orc_gather_omp"):
        fn = getattr(L, name)
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, _I32P, ctypes.c_int,
                       ctypes.c_int, ctypes.c_int]
        fn.restype = None
    L.orc_set_omp_threads.argtypes = [ctypes.c_int]
    L.orc_set_omp_threads.restype = None
    for name in ("orc_scatter_add_f32", "orc_scatter_add_f64", "orc_scatter_add_f32_omp"):
        fn = getattr(L, name)
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, _I32P, ctypes.c_int, ctypes.c_int]
        fn.restype = None
    return L


def _ints(v: Sequence[int]):
    return (ctypes.c_int * len(v))(*[int(x) for x in v])


def _i32p(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_I32P)


# --------------------------------------------------------------------------
# output size (spconv/pytorch/ops.py:73-96)
# --------------------------------------------------------------------------
def conv_out_shape(in_shape, ksize, stride, padding, dilation, out_padding=None,
                   transposed: bool = False) -> List[int]:
    ndim = len(in_shape)
    if out_padding is None:
        out_padding = [0] * ndim
    out = (ctypes.c_int * ndim)()
    lib().orc_conv_out_shape(ndim, _ints(in_shape), _ints(ksize), _ints(stride),
                             _ints(padding), _ints(dilation), _ints(out_padding),
                             int(transposed), out)
    return [int(v) for v in out]


# --------------------------------------------------------------------------
# rulebook = ops.get_indice_pairs CPU branch (ops.py:171-326) which calls
# SparseConvIndicesCPU (csrc/sparse/indices.py:1639-1778)
# --------------------------------------------------------------------------
def get_indice_pairs(indices: np.ndarray, batch_size: int, spatial_shape,
                     ksize, stride, padding, dilation, out_padding=None,
                     subm: bool = False, transpose: bool = False
                     ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, List[int]]:
    """Returns (out_inds [N_out, ndim+1], pair [2, kv, N_in], num_per_loc [kv],
    out_spatial_shape).  ``pair`` is -1 filled, like ops.py:191-197."""
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    n, ndim = indices.shape[0], indices.shape[1] - 1
    kv = int(np.prod(ksize))
    if out_padding is None:
        out_padding = [0] * ndim
    if subm:
        out_shape = [int(v) for v in spatial_shape]
    else:
        out_shape = conv_out_shape(spatial_shape, ksize, stride, padding, dilation,
                                   out_padding, transpose)
    if any(x <= 0 for x in out_shape):
        raise ValueError(f"your out spatial shape {out_shape} reach zero!!! "
                         f"input shape: {list(spatial_shape)}")      # ops.py:182-185
    pair = np.full((2, kv, n), -1, dtype=np.int32)
    num = np.zeros((kv,), dtype=np.int32)
    if subm:
        r = lib().orc_subm_rulebook(_i32p(indices), n, ndim, batch_size,
                                    _ints(spatial_shape), _ints(ksize), _ints(dilation),
                                    _i32p(pair), n, _i32p(num))
        if r < 0:
            raise ValueError("subm only support odd ksize")
        return indices, pair, num, out_shape
    out_inds = np.empty((max(kv * n, 1), ndim + 1), dtype=np.int32)
    num_act = lib().orc_conv_rulebook(_i32p(indices), n, ndim, batch_size,
                                      _ints(out_shape), _ints(spatial_shape),
                                      _ints(ksize), _ints(stride), _ints(padding),
                                      _ints(dilation), int(transpose), _i32p(pair), n,
                                      _i32p(out_inds), _i32p(num))
    return out_inds[:num_act].copy(), pair, num, out_shape


# --------------------------------------------------------------------------
# canonical implicit-GEMM artefacts derived from the Native lists
# (layout semantics: indices.py:806-874 SubM, :599-676 regular conv)
# --------------------------------------------------------------------------
def native_counts(num_per_loc: np.ndarray, kv: int, subm: bool, n_in: int) -> List[int]:
    """Effective list length per offset (SubM mirror rule, ops.py:962-968)."""
    out = []
    for k in range(kv):
        if subm and k == kv // 2:
            out.append(n_in)
        elif subm and k > kv // 2:
            out.append(int(num_per_loc[kv - 1 - k]))
        else:
            out.append(int(num_per_loc[k]))
    return out


def dense_tables(pair: np.ndarray, num_per_loc: np.ndarray, n_in: int, n_out: int,
                 subm: bool):
    """(pair_fwd [kv,N_out], pair_bwd [kv,N_in], mask_fwd [N_out,W], mask_bwd [N_in,W])

    pair_fwd[k,o] = input index feeding output o through offset k (or -1);
    pair_bwd[k,i] = output index fed by input i through offset k (or -1);
    mask bit k set iff the entry is valid (uint32 words, W = ceil(kv/32))."""
    kv = pair.shape[1]
    words = (kv + 31) // 32
    fwd = np.full((kv, n_out), -1, dtype=np.int32)
    bwd = np.full((kv, n_in), -1, dtype=np.int32)
    mfwd = np.zeros((n_out, words), dtype=np.uint32)
    mbwd = np.zeros((n_in, words), dtype=np.uint32)
    counts = native_counts(num_per_loc, kv, subm, n_in)
    for k in range(kv):
        c = counts[k]
        i_idx = pair[0, k, :c]
        o_idx = pair[1, k, :c]
        # first entry wins on duplicates, as written order in the CPU lists
        fwd[k, o_idx[::-1]] = i_idx[::-1]
        bwd[k, i_idx[::-1]] = o_idx[::-1]
        bit = np.uint32(1 << (k % 32))
        np.bitwise_or.at(mfwd[:, k // 32], o_idx, bit)
        np.bitwise_or.at(mbwd[:, k // 32], i_idx, bit)
    return fwd, bwd, mfwd, mbwd


# --------------------------------------------------------------------------
# Native conv forward / backward, CPU branch
# (ops.indice_conv ops.py:811-988, ops.indice_conv_backward ops.py:1103-1253;
#  C++ twin convops.py:1540-1633,1775-1860 + cppcore.py:232-348)
# --------------------------------------------------------------------------
_REF_GATHER = False


def use_reference_gather(on: bool) -> None:
    """Route the row gather / scatter-add of the Native driver loops through the reference's OWN
    GatherCPU code (oracle/_ref, rendered from spconv/csrc/sparse/gather.py:30-86) instead of the
    restatement in oracle.cpp -- tests pin one against the other.  fp32 tensors, serial form only."""
    global _REF_GATHER
    _REF_GATHER = bool(on)


def _gather(buf: torch.Tensor, src: torch.Tensor, inds: np.ndarray, omp: bool):
    if _REF_GATHER and buf.dtype == torch.float32 and not omp:
        from oracle import ref
        ref.gather(buf.numpy(), src.numpy(), inds)
        return
    fn = lib().orc_gather_omp if omp else lib().orc_gather
    fn(buf.data_ptr(), src.data_ptr(), _i32p(inds), len(inds), src.shape[1],
       src.element_size())


def _scatter_add(dst: torch.Tensor, buf: torch.Tensor, inds: np.ndarray, omp: bool):
    if _REF_GATHER and dst.dtype == torch.float32 and not omp:
        from oracle import ref
        ref.scatter_add(dst.numpy(), buf.numpy(), inds)
        return
    if dst.dtype == torch.float32:
        fn = lib().orc_scatter_add_f32_omp if omp else lib().orc_scatter_add_f32
    elif dst.dtype == torch.float64:
        fn = lib().orc_scatter_add_f64
    else:
        raise TypeError(dst.dtype)
    fn(dst.data_ptr(), buf.data_ptr(), _i32p(inds), len(inds), dst.shape[1])


def indice_conv(features: torch.Tensor, filters: torch.Tensor, pair: np.ndarray,
                num_per_loc: np.ndarray, num_act_out: int, inverse: bool = False,
                subm: bool = False, omp: bool = False) -> torch.Tensor:
    """filters: KRSC [K, *ksize, C] (ALL_WEIGHT_IS_KRSC, ops.py:888-894)."""
    assert features.dtype in (torch.float32, torch.float64)
    features = features.contiguous()
    K = filters.shape[0]
    w = filters.reshape(K, -1, filters.shape[-1])
    kv = w.shape[1]
    center = kv // 2
    if subm:
        out = torch.mm(features, w[:, center].T)             # ops.py:908
    else:
        out = torch.zeros((num_act_out, K), dtype=features.dtype)
    if kv == 1 and subm:
        return out
    nums = [int(v) for v in num_per_loc]
    if subm and all(v == 0 for v in nums):
        return out
    maxnhot = max(nums)
    pair_in, pair_out = pair[int(inverse)], pair[int(not inverse)]
    inp_buf = torch.empty((maxnhot, features.shape[1]), dtype=features.dtype)
    out_buf = torch.empty((maxnhot, K), dtype=features.dtype)
    for i, nhot in enumerate(nums):                            # ops.py:962-986
        if subm and i == center:
            continue
        if subm and i > center:
            nhot = nums[kv - i - 1]
        if nhot <= 0:
            continue
        ii = np.ascontiguousarray(pair_in[i, :nhot])
        oi = np.ascontiguousarray(pair_out[i, :nhot])
        _gather(inp_buf, features, ii, omp)
        torch.mm(inp_buf[:nhot], w[:, i].T, out=out_buf[:nhot])
        _scatter_add(out, out_buf, oi, omp)
    return out


def indice_conv_backward(features: torch.Tensor, filters: torch.Tensor,
                         out_bp: torch.Tensor, pair: np.ndarray,
                         num_per_loc: np.ndarray, inverse: bool = False,
                         subm: bool = False, omp: bool = False
                         ) -> Tuple[torch.Tensor, torch.Tensor]:
    features = features.contiguous()
    out_bp = out_bp.contiguous()
    fshape = filters.shape
    K = fshape[0]
    w = filters.reshape(K, -1, fshape[-1]).contiguous()
    kv = w.shape[1]
    center = kv // 2
    dw = torch.zeros_like(w)
    if subm:
        torch.mm(out_bp.T, features, out=dw[:, center])      # ops.py:1185
        din = torch.mm(out_bp, w[:, center])                   # ops.py:1187
    else:
        din = torch.zeros_like(features)
    if kv == 1 and subm:
        return din, dw.reshape(fshape)
    nums = [int(v) for v in num_per_loc]
    if subm and all(v == 0 for v in nums):
        return din, dw.reshape(fshape)
    maxnhot = max(nums)
    pair_in, pair_out = pair[int(inverse)], pair[int(not inverse)]
    inp_buf = torch.empty((maxnhot, features.shape[1]), dtype=features.dtype)
    out_buf = torch.empty((maxnhot, K), dtype=out_bp.dtype)
    for i, nhot in enumerate(nums):                            # ops.py:1225-1252
        if subm and i == center:
            continue
        if subm and i > center:
            nhot = nums[kv - i - 1]
        if nhot <= 0:
            continue
        ii = np.ascontiguousarray(pair_in[i, :nhot])
        oi = np.ascontiguousarray(pair_out[i, :nhot])
        _gather(inp_buf, features, ii, omp)
        _gather(out_buf, out_bp, oi, omp)
        torch.mm(out_buf[:nhot].T, inp_buf[:nhot], out=dw[:, i])   # overwrite
        torch.mm(out_buf[:nhot], w[:, i], out=inp_buf[:nhot])
        _scatter_add(din, inp_buf, ii, omp)
    return din, dw.reshape(fshape)


# --------------------------------------------------------------------------
# the reference's own test oracle: dense conv on the scattered dense input
# (test/test_conv.py:83-109,286-357)
# --------------------------------------------------------------------------
def dense_conv_reference(features: torch.Tensor, indices: np.ndarray, batch_size: int,
                         spatial_shape, weight_krsc: torch.Tensor, stride, padding,
                         dilation, transposed: bool = False, out_padding=None):
    """Returns the dense NC[D]HW output of torch conv{1,2,3}d on the densified input."""
    import torch.nn.functional as F
    ndim = len(spatial_shape)
    C = features.shape[1]
    dense = torch.zeros((batch_size, C, *spatial_shape), dtype=features.dtype)
    idx = torch.from_numpy(indices.astype(np.int64))
    dense[(idx[:, 0], slice(None), *[idx[:, i + 1] for i in range(ndim)])] = features
    # KRSC -> K C R S (torch layout)
    perm = [0, ndim + 1] + list(range(1, ndim + 1))
    w = weight_krsc.permute(*perm).contiguous()
    conv = {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}[ndim]
    convt = {1: F.conv_transpose1d, 2: F.conv_transpose2d, 3: F.conv_transpose3d}[ndim]
    if transposed:
        # conv_transpose weight layout is [C_in, C_out, ...]
        wt = w.transpose(0, 1).contiguous()
        return convt(dense, wt, None, stride, padding,
                     out_padding if out_padding is not None else 0, 1, dilation)
    return conv(dense, w, None, stride, padding, dilation)


# --------------------------------------------------------------------------
# int8 inference reference (test/test_all_algo.py:222-288, numpy; the module semantics are
# spconv/pytorch/quantization/quantized/conv.py:368-378)
def int8_conv_ref(features_i8: np.ndarray, weight_i8: np.ndarray, pair: np.ndarray,
                  num_per_loc: np.ndarray, n_out: int, subm: bool, scale: np.ndarray,
                  bias: np.ndarray, add_i8: Optional[np.ndarray] = None, add_scale: float = 0.0,
                  relu: bool = False, out_dtype=np.int8) -> np.ndarray:
    """acc_i32 over the Native lists, then ``clip(round(act(acc*scale + bias + add*add_scale)))``.
    The integer accumulation is done in float64 BLAS (exact below 2**53) to stay fast."""
    K, C = weight_i8.shape[0], weight_i8.shape[-1]
    kv = pair.shape[1]
    w = weight_i8.reshape(K, kv, C).astype(np.float64)
    f = features_i8.astype(np.float64)
    acc = np.zeros((n_out, K), dtype=np.float64)
    for k in range(kv):
        if subm and k == kv // 2:
            acc += f @ w[:, k, :].T
            continue
        if subm and k > kv // 2:
            nhot = int(num_per_loc[kv - 1 - k])
        else:
            nhot = int(num_per_loc[k])
        if nhot == 0:
            continue
        i_inds, o_inds = pair[0][k][:nhot], pair[1][k][:nhot]
        np.add.at(acc, o_inds, f[i_inds] @ w[:, k, :].T)
    acc_i32 = acc.astype(np.int64).astype(np.int32)
    rescaled = acc_i32.astype(np.float32) * scale.astype(np.float32)
    rescaled = rescaled + bias.astype(np.float32)
    if add_i8 is not None:
        rescaled = rescaled + add_i8.astype(np.float32) * np.float32(add_scale)
    if relu:
        rescaled = np.maximum(rescaled, 0)
    if out_dtype == np.int8:
        return np.clip(np.round(rescaled), -128, 127).astype(np.int8)
    return rescaled.astype(out_dtype)


# --------------------------------------------------------------------------
# pooling over the Native lists (spconv/csrc/sparse/maxpool.py:96-300 GPU kernels,
# :620-700 CPU loops; drivers spconv/pytorch/ops.py:1899-2084)
def _pool_lists(pair: np.ndarray, num_per_loc: np.ndarray, subm: bool, n_in: int):
    kv = pair.shape[1]
    for k in range(kv):
        if subm and k == kv // 2:
            ar = np.arange(n_in, dtype=np.int64)
            yield ar, ar
            continue
        nhot = int(num_per_loc[kv - 1 - k] if (subm and k > kv // 2) else num_per_loc[k])
        if nhot:
            yield pair[0][k][:nhot].astype(np.int64), pair[1][k][:nhot].astype(np.int64)


def maxpool_ref(features: np.ndarray, pair: np.ndarray, num_per_loc: np.ndarray, n_out: int,
                subm: bool = False, init_zero: bool = False) -> np.ndarray:
    """out[o] = max over its pairs (maxpool.py:96-140); init_zero = the Native path's zero-filled
    output (ops.py:1910)."""
    f = features.astype(np.float64)
    lowest = 0.0 if init_zero else -np.inf
    out = np.full((n_out, f.shape[1]), lowest, dtype=np.float64)
    for i_inds, o_inds in _pool_lists(pair, num_per_loc, subm, f.shape[0]):
        np.maximum.at(out, o_inds, f[i_inds])
    return out.astype(features.dtype)


def maxpool_bwd_ref(features: np.ndarray, out: np.ndarray, dout: np.ndarray, pair: np.ndarray,
                    num_per_loc: np.ndarray, subm: bool = False) -> np.ndarray:
    """din[i] += dout[o] where in[i] == out[o] (maxpool.py:142-209)."""
    din = np.zeros(features.shape, dtype=np.float64)
    for i_inds, o_inds in _pool_lists(pair, num_per_loc, subm, features.shape[0]):
        hit = features[i_inds] == out[o_inds]
        np.add.at(din, i_inds, dout[o_inds].astype(np.float64) * hit)
    return din.astype(features.dtype)


def indice_maxpool_native(features: np.ndarray, pair: np.ndarray, num_per_loc: np.ndarray, n_out: int):
    """(out, din_fn) restating the reference's Native max pooling ON THE CPU operation for operation
    (pytorch/ops.py:1899-1975 over maxpool.py:620-700): fp32, zero-filled output, offsets in list order, pairs in
    list order, `in > out` replaces, and in the backward pass `in == out` adds dout -- sequential fp32 adds, so the
    result is bit-comparable with the reference's code executed (oracle.ref.indice_maxpool*)."""
    f = np.ascontiguousarray(features, dtype=np.float32)
    out = np.zeros((n_out, f.shape[1]), dtype=np.float32)
    lists = [(pair[0][k][:int(n)], pair[1][k][:int(n)]) for k, n in enumerate(np.asarray(num_per_loc).tolist()) if n > 0]
    for ii, oi in lists:
        for i, o in zip(ii.tolist(), oi.tolist()):
            np.maximum(out[o], f[i], out=out[o])

    def backward(dout: np.ndarray) -> np.ndarray:
        d = np.ascontiguousarray(dout, dtype=np.float32)
        din = np.zeros_like(f)
        for ii, oi in lists:
            for i, o in zip(ii.tolist(), oi.tolist()):
                hit = f[i] == out[o]
                din[i] = np.where(hit, din[i] + d[o], din[i])
        return din
    return out, backward


def avgpool_ref(features: np.ndarray, pair: np.ndarray, num_per_loc: np.ndarray, n_out: int,
                subm: bool = False):
    """(mean over the pairs, count) -- maxpool.py:211-260."""
    f = features.astype(np.float64)
    acc = np.zeros((n_out, f.shape[1]), dtype=np.float64)
    cnt = np.zeros((n_out,), dtype=np.int32)
    for i_inds, o_inds in _pool_lists(pair, num_per_loc, subm, f.shape[0]):
        np.add.at(acc, o_inds, f[i_inds])
        np.add.at(cnt, o_inds, 1)
    out = np.where(cnt[:, None] > 0, acc / np.maximum(cnt, 1)[:, None], 0.0)
    return out.astype(features.dtype), cnt


def avgpool_bwd_ref(dout: np.ndarray, count: np.ndarray, pair: np.ndarray, num_per_loc: np.ndarray,
                    n_in: int, subm: bool = False, reference_quirks: bool = False) -> np.ndarray:
    """Gradient of the mean: din[i] += dout[o] / count[o].  reference_quirks: din[i] += dout[o] * count[o], what
    the reference kernel computes (maxpool.py:262-300: it multiplies by the count -- not the derivative of its
    forward; the product does the same under SPCONV_AMD_REFERENCE_QUIRKS=1)."""
    din = np.zeros((n_in, dout.shape[1]), dtype=np.float64)
    inv = (count.astype(np.float64) if reference_quirks
           else np.where(count > 0, 1.0 / np.maximum(count, 1), 0.0))
    for i_inds, o_inds in _pool_lists(pair, num_per_loc, subm, n_in):
        np.add.at(din, i_inds, dout[o_inds].astype(np.float64) * inv[o_inds][:, None])
    return din.astype(dout.dtype)


# --------------------------------------------------------------------------
# voxeliser (spconv/pytorch/utils.py:23-160 over csrc/sparse/pointops.py Point2VoxelCPU)
def point2voxel(points: np.ndarray, vsize_zyx, coors_range_zyx, grid_size_zyx, max_voxels: int,
                max_points: int, empty_mean: bool = False, reference_quirks: bool = False):
    """-> (voxels [V, max_points, F], indices [V, ndim] zyx, num_per_voxel [V], pc_voxel_id [N]).
    reference_quirks: the mean fill exactly as the reference's CPU loop behaves (its accumulator is carried from
    voxel to voxel, pointops.py:663-686) instead of the arithmetic mean."""
    L = lib()
    L.orc_point2voxel.restype = ctypes.c_int
    pts = np.ascontiguousarray(points, dtype=np.float32)
    n, nfeat = pts.shape
    ndim = len(vsize_zyx)
    voxels = np.zeros((max_voxels, max_points, nfeat), dtype=np.float32)
    indices = np.zeros((max_voxels, ndim), dtype=np.int32)
    num = np.zeros((max_voxels,), dtype=np.int32)
    pid = np.zeros((n,), dtype=np.int64)
    fl = lambda v: (ctypes.c_float * len(v))(*[float(x) for x in v])
    nv = L.orc_point2voxel(pts.ctypes.data_as(ctypes.c_void_p), n, nfeat, ndim, fl(vsize_zyx),
                           fl(coors_range_zyx), _ints(grid_size_zyx), int(max_voxels), int(max_points),
                           (2 if reference_quirks else 1) if empty_mean else 0, voxels.ctypes.data_as(ctypes.c_void_p),
                           indices.ctypes.data_as(ctypes.c_void_p), num.ctypes.data_as(ctypes.c_void_p),
                           pid.ctypes.data_as(ctypes.c_void_p))
    return voxels[:nv], indices[:nv], num[:nv], pid



def set_omp_threads(n: int) -> None:
    """Threads used by the faithful-omp gather / scatter variants (``omp=True``)."""
    lib().orc_set_omp_threads(int(n))
