"""SparseConvTensor and rulebook containers.

API parity target: ``spconv/pytorch/core.py:60-113,132-331`` of the reference
(constructor signature, attributes, ``replace_feature``, ``dense``,
``from_dense``, ``find_indice_pair``, ``shadow_copy``, ``select_by_index``,
arithmetic).  ``Rulebook`` is this implementation's single container for all
rulebook artefacts; ``IndiceData`` / ``ImplicitGemmIndiceData`` expose the
reference's attribute names on top of it.
"""
from __future__ import annotations

from enum import Enum
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.fx

from spconv_amd.constants import SPCONV_FX_TRACE_MODE


class ConvAlgo(Enum):
    """spconv/core.py:25-28."""
    Native = 0
    MaskImplicitGemm = 1
    MaskSplitImplicitGemm = 2


class Rulebook:
    """All artefacts of one rulebook, as produced by ``ops.build_rulebook``.

    pair_fwd  [kv, n_out] int32: input row feeding output row o through offset k, or -1
    pair_bwd  [kv, n_in ] int32: output row fed by input row i through offset k, or -1
                (SubM: None unless requested; pair_bwd[k] == pair_fwd[kv-1-k])
    mask_fwd  [n_out, W] int32 (bit k <=> pair_fwd[k][o] >= 0), mask_bwd [n_in, W]
    pair_native [2, kv, n_in] int32 + num_per_loc [kv]: ConvAlgo.Native lists, identical
                to the reference CPU path (csrc/sparse/indices.py:1639-1778)
    """

    def __init__(self, out_indices, pair_fwd, pair_bwd, mask_fwd, mask_bwd, pair_native,
                 num_per_loc, n_in: int, n_out: int, kv: int, subm: bool,
                 argsort_fwd=None, argsort_bwd=None):
        self.out_indices = out_indices
        self.pair_fwd = pair_fwd
        self.pair_bwd = pair_bwd
        self.mask_fwd = mask_fwd
        self.mask_bwd = mask_bwd
        self._pair_native = pair_native     # None: built on first use (see pair_native)
        self._num_per_loc = num_per_loc
        self.n_in = n_in
        self.n_out = n_out
        self.kv = kv
        self.subm = subm
        self.argsort_fwd = argsort_fwd
        self.argsort_bwd = argsort_bwd
        self.wgrad_plan = None          # built lazily by ops._plan_of
        self._native_swapped = None
        # geometry of the two row sets (set by ops.build_rulebook)
        self.in_indices = None
        self.in_shape = None
        self.out_shape = None
        self.batch_size = 1
        # rows layout of a SubM rulebook (ops.rows_layout: int32 blob of spx_subm_layout -- class word,
        # row order, mask words and pair table in tile order), None without one
        self.layout = None
        # copies of the tables in mask-sorted tile order (ops.sort_rulebook): [pair, mask] per
        # direction, None while unsorted; sort_decided: the automatic mode looked at this rulebook
        self.sorted_tables = {}
        self.sort_decided = False
        # density class (ops.sparse_neighbourhoods): None = not measured yet
        self.sparse_class = None
        self.heavy_rows = 0
        # asynchronous read of the class word (ops.poll_class): the pending request, and whose rulebook this is
        self._class_req = None
        self.pred_key = None
        # rank map of the OUTPUT level (a sorted-order strided build, ops._build_sorted), or None
        self.rankmap = None
        # static-shape build (ops.build_rulebook(static_num_out=...)): device int32 [2] =
        # {distinct outputs found, hash-table overflow}; rows >= the count are dead.  None otherwise.
        self.n_out_dev = None
        # live-row counts (device int32 [1] or None = all) of the input / output row sets
        self.in_n_live_dev = None
        self.out_n_live_dev = None

    def _ensure_native(self) -> None:
        """Inference builds only the dense tables; the ConvAlgo.Native lists (consumed by wgrad
        and by the Native-layout API) are derived from them when something asks."""
        if self._pair_native is None:
            from spconv_amd.pytorch import ops
            table = self.pair_fwd if self.subm else self.pair_bwd
            self._pair_native, self._num_per_loc = ops._native_from_table(table, self.subm)

    @property
    def has_native(self) -> bool:
        """The ConvAlgo.Native lists exist already (built with the rulebook or derived since)."""
        return self._pair_native is not None

    def native_handle(self):
        """(pair tensor, count tensor) to pass through the reference-shaped op signatures: the Native lists when
        they exist, else EMPTY placeholders [2, kv, 0] / [0] that carry the rulebook (ops.attach_rulebook) -- the
        ops resolve what they need from it and derive the lists only if a kernel asks for them."""
        from spconv_amd.pytorch import ops
        if self._pair_native is not None:
            return ops.attach_rulebook(self._pair_native, self), self._num_per_loc
        dev = self.pair_fwd.device
        return (ops.attach_rulebook(torch.empty((2, self.kv, 0), dtype=torch.int32, device=dev), self),
                torch.empty((0,), dtype=torch.int32, device=dev))

    @property
    def pair_native(self) -> torch.Tensor:
        self._ensure_native()
        return self._pair_native

    @property
    def num_per_loc(self) -> torch.Tensor:
        self._ensure_native()
        return self._num_per_loc

    def native_swapped(self) -> torch.Tensor:
        """Native lists with in/out roles exchanged (inverse convolution)."""
        if self._native_swapped is None:
            self._native_swapped = torch.stack([self.pair_native[1], self.pair_native[0]]).contiguous()
        return self._native_swapped


class IndiceData(object):
    """spconv/pytorch/core.py:60-78 (ConvAlgo.Native bookkeeping)."""

    def __init__(self, out_indices, indices, indice_pairs, indice_pair_num, spatial_shape,
                 out_spatial_shape, is_subm: bool, algo: ConvAlgo, ksize: List[int],
                 stride: List[int], dilation: List[int], padding: List[int],
                 voxel_num: Optional[Any] = None, rulebook: Optional[Rulebook] = None):
        self.out_indices = out_indices
        self.indices = indices
        self.indice_pairs = indice_pairs
        self.indice_pair_num = indice_pair_num
        self.spatial_shape = spatial_shape
        self.out_spatial_shape = out_spatial_shape
        self.is_subm = is_subm
        self.algo = algo
        self.ksize = ksize
        self.stride = stride
        self.dilation = dilation
        self.padding = padding
        self.voxel_num = voxel_num
        self.rulebook = rulebook


class ImplicitGemmIndiceData(object):
    """spconv/pytorch/core.py:81-112 (implicit-GEMM bookkeeping)."""

    def __init__(self, out_indices: torch.Tensor, indices: torch.Tensor, pair_fwd: torch.Tensor,
                 pair_bwd: torch.Tensor, pair_mask_fwd_splits: List[torch.Tensor],
                 pair_mask_bwd_splits: List[torch.Tensor],
                 mask_argsort_fwd_splits: List[torch.Tensor],
                 mask_argsort_bwd_splits: List[torch.Tensor], masks: List[np.ndarray],
                 spatial_shape, out_spatial_shape, is_subm: bool, algo: ConvAlgo,
                 ksize: List[int], stride: List[int], dilation: List[int], padding: List[int],
                 in_voxel_num: Optional[Any] = None, out_voxel_num: Optional[Any] = None,
                 rulebook: Optional[Rulebook] = None):
        self.out_indices = out_indices
        self.indices = indices
        self.pair_fwd = pair_fwd
        self.pair_bwd = pair_bwd
        self.pair_mask_fwd_splits = pair_mask_fwd_splits
        self.pair_mask_bwd_splits = pair_mask_bwd_splits
        self.mask_argsort_fwd_splits = mask_argsort_fwd_splits
        self.mask_argsort_bwd_splits = mask_argsort_bwd_splits
        self.masks = masks
        self.spatial_shape = spatial_shape
        self.out_spatial_shape = out_spatial_shape
        self.is_subm = is_subm
        self.algo = algo
        self.ksize = ksize
        self.stride = stride
        self.dilation = dilation
        self.padding = padding
        self.in_voxel_num = in_voxel_num
        self.out_voxel_num = out_voxel_num
        self.rulebook = rulebook


def scatter_nd(indices: torch.Tensor, updates: torch.Tensor, shape: Sequence[int]) -> torch.Tensor:
    """Dense tensor of ``shape`` with ``updates`` written at ``indices`` (no repeat-add)."""
    ret = torch.zeros(*shape, dtype=updates.dtype, device=updates.device)
    ndim = indices.shape[-1]
    flat = indices.view(-1, ndim)
    sel = tuple(flat[:, i] for i in range(ndim)) + (Ellipsis,)
    ret[sel] = updates.view(*(list(indices.shape[:-1]) + list(shape[ndim:])))
    return ret


def _under_fx_trace(*values) -> bool:
    return SPCONV_FX_TRACE_MODE or any(isinstance(v, torch.fx.Proxy) for v in values)


class SparseConvTensor(metaclass=torch.fx.ProxyableClassMeta):
    # ProxyableClassMeta (reference core.py:24-31,132): constructing the tensor from fx Proxies
    # while a model is traced records the construction in the graph
    def __init__(self, features: torch.Tensor, indices: torch.Tensor,
                 spatial_shape: Union[List[int], np.ndarray], batch_size: int,
                 grid: Optional[torch.Tensor] = None, voxel_num: Optional[torch.Tensor] = None,
                 indice_dict: Optional[dict] = None, benchmark: bool = False,
                 permanent_thrust_allocator: bool = False, enable_timer: bool = False,
                 force_algo: Optional[ConvAlgo] = None):
        """
        features: [num_points, num_features]; indices: int32 [num_points, ndim + 1] with the
        batch index in column 0; spatial_shape: ndim ints; batch_size > 0.  The remaining
        arguments exist for signature compatibility (reference core.py:133-144):
        ``grid``/``voxel_num`` are carried along, ``benchmark`` records per-layer wall time,
        ``permanent_thrust_allocator`` has no effect here; ``enable_timer`` attaches a
        ``spconv_amd.tools.CUDAKernelTimer`` (HIP events) that every sparse layer records into.
        """
        if not _under_fx_trace(features, indices, batch_size):
            ndim = indices.shape[1] - 1
            assert features.ndim == 2
            assert indices.ndim == 2
            assert len(spatial_shape) == ndim, "spatial shape must equal to ndim"
            assert indices.dtype == torch.int32, "only support int32"
            assert batch_size > 0
        self._features = features
        self.indices = indices
        self.spatial_shape = [int(v) for v in spatial_shape]
        self.batch_size = batch_size
        self.indice_dict: Dict[Any, Any] = {} if indice_dict is None else indice_dict
        self.grid = torch.Tensor() if grid is None else grid
        self.voxel_num = voxel_num
        self.benchmark = benchmark
        self.benchmark_record: Dict[str, Any] = {}
        self.thrust_allocator = None
        self._timer = None
        if enable_timer:
            from spconv_amd.tools import CUDAKernelTimer
            self._timer = CUDAKernelTimer(True)
        self.force_algo = force_algo
        self.int8_scale: Optional[np.ndarray] = None
        # static-shape tensors (spconv_amd/pytorch/static.py): device int32 [1], the number of leading rows
        # that are rows of the scene (the rest is padding with batch index -1); None = every row
        self.n_live_dev: Optional[torch.Tensor] = None

    def __repr__(self):
        return f"SparseConvTensor[shape={self._features.shape}]"

    @property
    def is_quantized(self):
        return self.features.dtype == torch.qint8

    def q_scale(self):
        if self.is_quantized:
            return self.features.q_scale()
        raise ValueError("sparse tensor must be quantized")

    def dequantize(self) -> "SparseConvTensor":
        return self.replace_feature(self.features.dequantize())

    def _like(self, features: torch.Tensor, indice_dict) -> "SparseConvTensor":
        if not isinstance(features, torch.fx.Proxy):
            # hot path (every layer makes one): skip the constructor's checks and the fx metaclass
            # call -- same coordinates, same bookkeeping, new features
            t = object.__new__(SparseConvTensor)
            t.__dict__.update(self.__dict__)
            t._features = features
            t.indice_dict = {} if indice_dict is None else indice_dict
            return t
        t = SparseConvTensor(features, self.indices, self.spatial_shape, self.batch_size,
                             self.grid, self.voxel_num, indice_dict, self.benchmark)
        t.benchmark_record = self.benchmark_record
        t.thrust_allocator = self.thrust_allocator
        t._timer = self._timer
        t.force_algo = self.force_algo
        t.int8_scale = self.int8_scale
        return t

    def replace_feature(self, feature: torch.Tensor) -> "SparseConvTensor":
        """``x = x.replace_feature(F.relu(x.features))`` instead of assigning ``x.features``."""
        return self._like(feature, self.indice_dict)

    def shadow_copy(self) -> "SparseConvTensor":
        """New tensor object sharing every member."""
        return self._like(self.features, self.indice_dict)

    def select_by_index(self, valid_indices: torch.Tensor) -> "SparseConvTensor":
        t = self._like(self.features[valid_indices], self.indice_dict)
        t.indices = self.indices[valid_indices]
        # cached rulebooks describe the old coordinate set
        t.indice_dict.clear()
        return t

    def minus(self):
        return self.replace_feature(-self.features)

    @property
    def features(self):
        return self._features

    @features.setter
    def features(self, val):
        raise ValueError("you can't set feature directly, use 'x = x.replace_feature("
                         "your_new_feature)' to generate new SparseConvTensor instead.")

    @classmethod
    def from_dense(cls, x: torch.Tensor) -> "SparseConvTensor":
        """x: channel-last dense tensor [N, *spatial, C]."""
        x_sp = x.to_sparse(x.ndim - 1)
        spatial_shape = x_sp.shape[1:-1]
        batch_size = x_sp.shape[0]
        indices = x_sp.indices().permute(1, 0).contiguous().int()
        return cls(x_sp.values(), indices, spatial_shape, batch_size)

    def dequantize(self):
        return self.replace_feature(self.features.dequantize())

    @property
    def spatial_size(self):
        return np.prod(self.spatial_shape)

    def find_indice_pair(self, key) -> Optional[Union[IndiceData, ImplicitGemmIndiceData]]:
        if key is None:
            return None
        return self.indice_dict.get(key, None)

    def dense(self, channels_first: bool = True) -> torch.Tensor:
        if self.n_live_dev is not None:
            # a static-shape tensor: padding rows carry batch index -1, which plain indexing would wrap around
            from spconv_amd.pytorch.static import dense_static
            return dense_static(self, channels_first)
        out_shape = [self.batch_size] + list(self.spatial_shape) + [self.features.shape[1]]
        res = scatter_nd(self.indices.to(self.features.device).long(), self.features, out_shape)
        if not channels_first:
            return res
        ndim = len(self.spatial_shape)
        perm = [0, ndim + 1] + list(range(1, ndim + 1))
        return res.permute(*perm).contiguous()

    @staticmethod
    def _other_features(other):
        assert isinstance(other, (SparseConvTensor, torch.Tensor))
        return other if isinstance(other, torch.Tensor) else other.features

    def __add__(self, other):
        return self.replace_feature(self.features + self._other_features(other))

    def __radd__(self, other):
        return self.replace_feature(self.features + self._other_features(other))

    def __iadd__(self, other):
        self._features += self._other_features(other)
        return self


def expand_nd(ndim: int, val: Union[int, List[int], Tuple[int, ...], np.ndarray]) -> List[int]:
    if isinstance(val, int):
        res = [val] * ndim
    else:
        res = list(val)
    assert len(res) == ndim
    return [int(v) for v in res]
