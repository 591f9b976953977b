"""Point cloud -> voxel conversion (reference: ``spconv/pytorch/utils.py:23-200``).

``PointToVoxel`` keeps the reference's constructor and call signatures and its conventions:
parameters are given in XYZ order, the returned ``indices`` are ZYX (``docs/USAGE.md:222``),
voxels are numbered in first-seen point order and keep their first
``max_num_points_per_voxel`` points -- exactly what the reference's CPU generator produces
(``csrc/sparse/pointops.py``), here computed on the GPU by ``spx_point2voxel``."""
from __future__ import annotations

import ctypes
from typing import List, Union

import numpy as np
import torch

from spconv_amd import _lib, constants


def calc_point2voxel_meta_data(vsize_xyz: List[float], coors_range_xyz: List[float]):
    """(vsize, grid_size, grid_stride, coors_range) in ZYX order -- Point2VoxelCommon::calc_meta_data
    (csrc/sparse/pointops.py, all.py:1349-1386): float32 arithmetic, std::round."""
    ndim = len(vsize_xyz)
    assert len(coors_range_xyz) == 2 * ndim, "your params size not equal to ndim"
    vs = np.asarray(vsize_xyz, dtype=np.float32)[::-1].copy()
    lo = np.asarray(coors_range_xyz[:ndim], dtype=np.float32)[::-1].copy()
    hi = np.asarray(coors_range_xyz[ndim:], dtype=np.float32)[::-1].copy()
    q = (hi - lo) / vs                                   # float32 division, like the C++
    grid = np.where(q >= 0, np.floor(q + np.float32(0.5)), np.ceil(q - np.float32(0.5))).astype(np.int64)
    stride, prod = [0] * ndim, 1
    for i in range(ndim - 1, -1, -1):
        stride[i] = prod
        prod *= int(grid[i])
    return ([float(v) for v in vs], [int(v) for v in grid], stride,
            [float(v) for v in lo] + [float(v) for v in hi])


class PointToVoxel(object):
    """WARNING: you MUST construct PointToVoxel AFTER set device."""

    def __init__(self, vsize_xyz: List[float], coors_range_xyz: List[float], num_point_features: int,
                 max_num_voxels: int, max_num_points_per_voxel: int,
                 device: torch.device = torch.device("cuda:0"), key_order: bool = False):
        # key_order (not in the reference, pytorch/utils.py:23-160, whose voxels come out in point / hash-slot order):
        # voxels numbered by ascending (z, y, x) key instead of by their first point -- the order the first level of a
        # backbone wants (sort_voxels_by_coordinate below; DESIGN.md sections 3.15 / 3.17).  Scenes concatenated in
        # batch order stay in key order; pc_voxel_id follows the renumbering.
        self.key_order = bool(key_order)
        self.ndim = len(vsize_xyz)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise NotImplementedError("spconv_amd runs on MI355X only: construct PointToVoxel with a "
                                      "cuda device (there is no CPU path)")
        vsize, grid_size, grid_stride, coors_range = calc_point2voxel_meta_data(vsize_xyz, coors_range_xyz)
        self.num_point_features = num_point_features
        self.max_num_voxels = max_num_voxels
        self.max_num_points_per_voxel = max_num_points_per_voxel
        self.vsize = vsize
        self.grid_size = grid_size
        self.grid_stride = grid_stride
        self.coors_range = coors_range
        self.voxels = torch.zeros([max_num_voxels, max_num_points_per_voxel, num_point_features],
                                  dtype=torch.float32, device=self.device)
        self.indices = torch.zeros([max_num_voxels, self.ndim], dtype=torch.int32, device=self.device)
        self.num_per_voxel = torch.zeros([max_num_voxels], dtype=torch.int32, device=self.device)

    def __call__(self, pc: torch.Tensor, clear_voxels: bool = True, empty_mean: bool = False):
        """pc [N, 3+] -> (voxels, indices (zyx), num_per_voxel)."""
        res = self.generate_voxel_with_id(pc, clear_voxels, empty_mean)
        return res[0], res[1], res[2]

    def generate_voxel_with_id(self, pc: torch.Tensor, clear_voxels: bool = True,
                               empty_mean: bool = False):
        """-> (voxels, indices, num_per_voxel, pc_voxel_id [N] int64, -1 for dropped points)."""
        assert pc.device.type == self.device.type, "your pc device is wrong"
        assert pc.ndim == 2 and pc.shape[1] == self.num_point_features, \
            "your points num features doesn't equal to voxel."
        if pc.is_cuda and pc.device.index != torch.cuda.current_device():
            with torch.cuda.device(pc.device):         # DeviceGuard convention (ops._on_device)
                return self.generate_voxel_with_id(pc, clear_voxels, empty_mean)
        L = _lib.load()
        with torch.no_grad():
            pc = pc.contiguous().float()
            n = pc.shape[0]
            pc_voxel_id = torch.empty([n], dtype=torch.int64, device=pc.device)
            ws = torch.empty((max(int(L.spx_point2voxel_ws_bytes(n, self.max_num_voxels)), 16),),
                             dtype=torch.uint8, device=pc.device)
            nv = ctypes.c_int(0)
            f = lambda v: (ctypes.c_float * len(v))(*v)
            _lib.check(L.spx_point2voxel(
                pc.data_ptr(), n, self.num_point_features, self.ndim, f(self.vsize), f(self.coors_range),
                _lib.ints(self.grid_size), self.max_num_voxels, self.max_num_points_per_voxel,
                (2 if constants.REFERENCE_QUIRKS else 1) if empty_mean else 0, int(clear_voxels),
                self.voxels.data_ptr(), self.indices.data_ptr(),
                self.num_per_voxel.data_ptr(), pc_voxel_id.data_ptr(), ctypes.byref(nv), ws.data_ptr(),
                ws.numel(), torch.cuda.current_stream(pc.device).cuda_stream))
            num_voxels = int(nv.value)
            if self.key_order and num_voxels > 1:
                idx = self.indices[:num_voxels]
                key = idx[:, 0].to(torch.int64)
                for d in range(1, self.ndim):          # (indices and grid_size are both in zyx order)
                    key = key * int(self.grid_size[d]) + idx[:, d].to(torch.int64)
                order = torch.argsort(key)
                rank = torch.empty_like(order)
                rank[order] = torch.arange(num_voxels, device=order.device)
                pc_voxel_id = torch.where(pc_voxel_id >= 0, rank[pc_voxel_id.clamp_min(0)], pc_voxel_id)
                return (self.voxels[:num_voxels][order].contiguous(), idx[order].contiguous(),
                        self.num_per_voxel[:num_voxels][order].contiguous(), pc_voxel_id)
            return (self.voxels[:num_voxels].clone(), self.indices[:num_voxels].clone(),
                    self.num_per_voxel[:num_voxels].clone(), pc_voxel_id)


def gather_features_by_pc_voxel_id(seg_res_features: torch.Tensor, pc_voxel_id: torch.Tensor,
                                   invalid_value: Union[int, float] = 0):
    """Per-voxel results back to the points: row i of the result is the row of point i's voxel,
    `invalid_value` for points that fell outside the grid (pc_voxel_id == -1).  Same contract as
    the reference helper (spconv/pytorch/utils.py:163-176)."""
    ids = pc_voxel_id.to(seg_res_features.device)
    inside = ids >= 0
    rows = seg_res_features.index_select(0, ids.clamp_min(0))
    shape = [-1] + [1] * (seg_res_features.ndim - 1)
    fill = torch.full_like(rows, invalid_value)
    return torch.where(inside.view(shape), rows, fill)


def _row_keys(indices: torch.Tensor, spatial_shape) -> torch.Tensor:
    key = indices[:, 0].to(torch.int64)
    for d, s in enumerate(spatial_shape):
        key = key * int(s) + indices[:, 1 + d].to(torch.int64)
    return key


def sort_voxels_by_coordinate(indices: torch.Tensor, spatial_shape: List[int], *row_tensors: torch.Tensor,
                              batch_size: int = 0, rank_map: bool = True):
    """Rows in ascending coordinate-key order (batch-major, last axis fastest): ``(indices, *row_tensors, order)``.

    Not part of the reference's API.  The FIRST level of a backbone runs in the order the caller hands over; a
    voxeliser returns voxels in point or hash-slot order, i.e. shuffled in space.  Every gather of a level is cheaper when
    x-neighbours sit in adjacent rows (on the 4 x 100 k voxel level of BASELINE config 4 a SubM
    forward takes 40 instead of 49 us, its backward 84 instead of 96): a data loader that sorts once gives the first level
    what the layer modules give every level behind a strided layer.  ``order`` maps sorted rows to input rows
    (``x_sorted = x[order]``) for carrying labels along.

    ``rank_map=True`` with ``batch_size`` given (CUDA tensors): the sorted index tensor additionally carries the level's
    rank map (``ops.attach_rank_map``: row = rank, built by one pass over the rows), so the SubM layers of the first
    level build their rulebook without a hash table, like the levels behind a strided layer do.  Coordinates that occur
    twice are detected on the device (one synchronisation, here in the data loader) and leave the tensor untagged."""
    assert indices.dim() == 2 and indices.shape[1] == len(spatial_shape) + 1
    if (batch_size > 0 and indices.is_cuda and indices.dtype == torch.int32 and indices.is_contiguous()
            and indices.shape[0] > 0):
        # the library's own sort (spx_key_argsort: four launches, the rank map written by the same pass -- ~60 us for
        # 420 k rows where torch.argsort of the keys takes 160); a coordinate that occurs twice (one read of the
        # device-side verdict, here in the data loader) takes the general path below
        from spconv_amd.pytorch import ops
        flag = torch.zeros((1,), dtype=torch.int32, device=indices.device)
        res = ops.key_argsort(indices, int(batch_size), [int(v) for v in spatial_shape], rank_map=rank_map, violation=flag)
        if res is not None:
            order32, out = res
            tagged = getattr(out, "_spx_rankmap", None) is not None
            unique = bool(int(flag.item()) == 0) if tagged else bool((_row_keys(out, spatial_shape).diff() > 0).all())
            if unique:
                order = order32.long()
                return (out, *[t.index_select(0, order) for t in row_tensors], order)
    key = indices[:, 0].to(torch.int64)
    for d, s in enumerate(spatial_shape):
        key = key * int(s) + indices[:, 1 + d].to(torch.int64)
    order = torch.argsort(key)
    out = indices[order].contiguous()
    if rank_map and batch_size > 0 and out.is_cuda and out.dtype == torch.int32:
        from spconv_amd.pytorch import ops
        ops.attach_rank_map(out, int(batch_size), [int(v) for v in spatial_shape], check=True)
    return (out, *[t[order].contiguous() for t in row_tensors], order)
