"""ctypes front of `oracle/_ref/libspconv_ref.so`: the reference's OWN CPU code -- the rulebook generators
(`SparseConvIndicesCPU.generate_subm_conv_inds` / `generate_conv_inds`, csrc/sparse/indices.py:
1639-1778, and `ConvOutLocIter`, :76-269), the row gather / scatter-add (gather.py:30-86), the voxeliser
(`Point2VoxelCPU`, pointops.py:493-766) and the max-pool loops (`IndiceMaxPoolCPU`, maxpool.py:590-703) --
rendered from the reference's source files where they lie and compiled by `make -C oracle ref` (see
refbuild/render.py for what is and is not reference text).

TEST INFRASTRUCTURE: pins the restatement in oracle.cpp and generates tests/golden/ref_*.npz.  The
library exists wherever it was built (`oracle/_ref/` travels to the GPU box with the snapshot; it is
git-ignored); `available()` says whether it does.  Buffers are prepared as the reference's Python
driver prepares them for its CPU branch (pytorch/ops.py:186-196,311-326): pairs filled with -1,
counts zeroed, `out_inds` sized kv * N and cut to the returned count.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "libspconv_ref.so")
REFERENCE = os.environ.get("SPCONV_REFERENCE", "/root/reference")
_lib: Optional[ctypes.CDLL] = None


def can_build() -> bool:
    return os.path.isfile(os.path.join(REFERENCE, "spconv", "csrc", "sparse", "indices.py"))


def build(force: bool = False) -> Optional[str]:
    """Renders + compiles the reference code when /root/reference is present; returns the path
    of the library, or None when it can neither be built nor found."""
    global _lib
    if can_build() and (force or not os.path.isfile(LIB)):
        if force:
            subprocess.run(["make", "-C", HERE, "clean-ref"], check=True, capture_output=True)
        subprocess.run(["make", "-C", HERE, "ref"], check=True, capture_output=True)
        _lib = None
    return LIB if os.path.isfile(LIB) else None


def available() -> bool:
    return os.path.isfile(LIB)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not available():
            raise FileNotFoundError(f"{LIB} not built (needs /root/reference: make -C oracle ref)")
        _lib = ctypes.CDLL(LIB)
        ip = ctypes.POINTER(ctypes.c_int)
        _lib.ref_generate_inds.restype = ctypes.c_int
        _lib.ref_generate_inds.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ip] * 6
        for fn in (_lib.ref_gather, _lib.ref_scatter_add):
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                           ctypes.c_int]
        vp, ci = ctypes.c_void_p, ctypes.c_int
        fp = ctypes.POINTER(ctypes.c_float)
        _lib.ref_point2voxel.restype = ci
        _lib.ref_point2voxel.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, fp, ip, ip, fp, ci, ci, ci]
        _lib.ref_maxpool_fwd.restype = ci
        _lib.ref_maxpool_fwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci]
        _lib.ref_maxpool_bwd.restype = ci
        _lib.ref_maxpool_bwd.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci]
        _lib.ref_global_pool_rearrange.restype = ci
        _lib.ref_global_pool_rearrange.argtypes = [vp, vp, vp, ci, ci, ci]
    return _lib


def point2voxel(points: np.ndarray, vsize_zyx, coors_range_zyx, grid_size_zyx, max_voxels: int, max_points: int,
                empty_mean: bool = False):
    """The reference's Point2VoxelCPU::point_to_voxel[_empty_mean]_static (pointops.py:598-700), executed, on
    buffers prepared like its constructor prepares them (pointops.py:568-578).  Same signature and return as
    oracle.point2voxel: (voxels [V, max_points, F], indices [V, 3] zyx, num_per_voxel [V], pc_voxel_id [N])."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    n, nfeat = pts.shape
    assert len(vsize_zyx) == 3, "the rendered class is the 3-d one"
    voxels = np.zeros((max_voxels, max_points, nfeat), dtype=np.float32)
    indices = np.zeros((max_voxels, 3), dtype=np.int32)
    num = np.zeros((max_voxels,), dtype=np.int32)
    grid = np.full(tuple(int(g) for g in grid_size_zyx), -1, dtype=np.int32)
    pid = np.zeros((n,), dtype=np.int64)
    stride, prod = [0, 0, 0], 1
    for i in (2, 1, 0):
        stride[i] = prod
        prod *= int(grid_size_zyx[i])
    fl = lambda v: (ctypes.c_float * len(v))(*[float(x) for x in v])
    nv = lib().ref_point2voxel(pts.ctypes.data, n, nfeat, voxels.ctypes.data, indices.ctypes.data, num.ctypes.data,
                               grid.ctypes.data, pid.ctypes.data, fl(vsize_zyx), _ints(grid_size_zyx), _ints(stride),
                               fl(coors_range_zyx), int(max_voxels), int(max_points), int(empty_mean))
    if nv < 0:
        raise RuntimeError(f"reference voxeliser failed (rc {nv})")
    return voxels[:nv], indices[:nv], num[:nv], pid


def indice_maxpool(features: np.ndarray, pair: np.ndarray, num_per_loc: np.ndarray, n_out: int) -> np.ndarray:
    """The reference's Native max pooling on the CPU, executed: the driver loop of pytorch/ops.py:1899-1934 (zero-filled
    output, one call per kernel offset with that offset's first `nhot` pairs) over IndiceMaxPoolCPU::forward
    (maxpool.py:620-655)."""
    f = np.ascontiguousarray(features, dtype=np.float32)
    out = np.zeros((n_out, f.shape[1]), dtype=np.float32)
    for k, nhot in enumerate(np.asarray(num_per_loc).tolist()):
        if nhot <= 0:
            continue
        ii = np.ascontiguousarray(pair[0][k][:nhot], dtype=np.int32)
        oi = np.ascontiguousarray(pair[1][k][:nhot], dtype=np.int32)
        lib().ref_maxpool_fwd(out.ctypes.data, f.ctypes.data, oi.ctypes.data, ii.ctypes.data, int(nhot), n_out,
                              f.shape[0], f.shape[1])
    return out


def indice_maxpool_backward(features: np.ndarray, out: np.ndarray, dout: np.ndarray, pair: np.ndarray,
                            num_per_loc: np.ndarray) -> np.ndarray:
    """pytorch/ops.py:1940-1975 over IndiceMaxPoolCPU::backward (maxpool.py:657-700), executed."""
    f = np.ascontiguousarray(features, dtype=np.float32)
    o = np.ascontiguousarray(out, dtype=np.float32)
    d = np.ascontiguousarray(dout, dtype=np.float32)
    din = np.zeros_like(f)
    for k, nhot in enumerate(np.asarray(num_per_loc).tolist()):
        if nhot <= 0:
            continue
        ii = np.ascontiguousarray(pair[0][k][:nhot], dtype=np.int32)
        oi = np.ascontiguousarray(pair[1][k][:nhot], dtype=np.int32)
        lib().ref_maxpool_bwd(o.ctypes.data, f.ctypes.data, d.ctypes.data, din.ctypes.data, oi.ctypes.data,
                              ii.ctypes.data, int(nhot), o.shape[0], f.shape[0], f.shape[1])
    return din


def global_pool_rearrange(coords: np.ndarray, batch_size: int):
    """IndiceMaxPoolCPU::global_pool_rearrange (maxpool.py:598-618), executed: (out_indices [batch, N] filled with
    -1 as pytorch/ops.py prepares it, counts [batch])."""
    c = np.ascontiguousarray(coords, dtype=np.int32)
    n = c.shape[0]
    out = np.full((batch_size, n), -1, dtype=np.int32)
    counts = np.zeros((batch_size,), dtype=np.int32)
    lib().ref_global_pool_rearrange(out.ctypes.data, c.ctypes.data, counts.ctypes.data, n, c.shape[1], batch_size)
    return out, counts


def gather(out: np.ndarray, src: np.ndarray, inds: np.ndarray) -> None:
    """The reference's GatherCPU::gather (gather.py:30-53), executed: out[i] = src[inds[i]] (fp32 rows)."""
    assert out.dtype == np.float32 and src.dtype == np.float32 and out.flags.c_contiguous and src.flags.c_contiguous
    inds = np.ascontiguousarray(inds, dtype=np.int32)
    lib().ref_gather(out.ctypes.data, src.ctypes.data, inds.ctypes.data, len(inds), src.shape[0], src.shape[1])


def scatter_add(out: np.ndarray, buf: np.ndarray, inds: np.ndarray) -> None:
    """The reference's GatherCPU::scatter_add (gather.py:55-86), executed: out[inds[i]] += buf[i]."""
    assert out.dtype == np.float32 and buf.dtype == np.float32 and out.flags.c_contiguous and buf.flags.c_contiguous
    inds = np.ascontiguousarray(inds, dtype=np.int32)
    lib().ref_scatter_add(out.ctypes.data, buf.ctypes.data, inds.ctypes.data, len(inds), out.shape[0],
                          out.shape[1])


def _ints(v: Sequence[int]):
    return (ctypes.c_int * len(v))(*[int(x) for x in v])


def get_indice_pairs(indices: np.ndarray, batch_size: int, spatial_shape, ksize, stride, padding,
                     dilation, out_padding=None, subm: bool = False, transpose: bool = False):
    """Same return shape as oracle.get_indice_pairs: (out_inds, pair [2, kv, N], num [kv], out_shape)."""
    import oracle
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    n, ndim = indices.shape[0], indices.shape[1] - 1
    kv = int(np.prod(ksize))
    if subm:
        out_shape = list(spatial_shape)
    else:
        out_shape = oracle.conv_out_shape(spatial_shape, ksize, stride, padding, dilation, out_padding,
                                          transpose)
    pair = np.full((2, kv, n), -1, dtype=np.int32)
    num = np.zeros((kv,), dtype=np.int32)
    out_inds = indices.copy() if subm else np.empty((max(kv * n, 1), ndim + 1), dtype=np.int32)
    rc = lib().ref_generate_inds(ndim, int(subm), int(transpose), indices.ctypes.data, n,
                                 pair.ctypes.data, n, out_inds.ctypes.data, out_inds.shape[0],
                                 num.ctypes.data, int(batch_size), _ints(spatial_shape), _ints(out_shape),
                                 _ints(ksize), _ints(stride if not subm else [1] * ndim),
                                 _ints(padding if not subm else [0] * ndim), _ints(dilation))
    if rc < 0:
        raise RuntimeError(f"reference generator failed (rc {rc})")
    return np.ascontiguousarray(out_inds[:rc]), pair, num, list(out_shape)
