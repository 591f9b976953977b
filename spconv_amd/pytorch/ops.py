"""Op drivers: torch tensors in, C-ABI calls out.

Mirrors the public functions of ``spconv/pytorch/ops.py`` (names, argument
meaning, error behaviour) for the hot path:

* ``get_conv_output_size`` / ``get_deconv_output_size``   (ops.py:73-96)
* ``get_indice_pairs``                                     (ops.py:132-326)
* ``get_indice_pairs_implicit_gemm``                       (ops.py:329-808)
* ``indice_conv`` / ``indice_conv_backward``               (ops.py:811-1095,1103-1447)
* ``implicit_gemm`` / ``implicit_gemm_backward``           (ops.py:1450-1664,1667-1896)

All compute happens in the HIP library behind ``include/spconv_amd.h``; torch is
used for allocation (the reference's TorchAllocator role, cppcore.py:112-223)
and to read the current stream (cppcore.py:98-99).  CPU tensors are rejected:
this implementation has no CPU path.
"""
from __future__ import annotations

import collections
import contextlib
import ctypes
import functools
import os
import threading
import weakref
from typing import List, Optional, Tuple

import numpy as np
import torch

from spconv_amd import _lib
from spconv_amd.constants import SPCONV_DO_SORT
from spconv_amd.pytorch.core import ConvAlgo, Rulebook

INT32_MAX = 2147483647

_POINT_VANISH_MSG = """Your points vanished here, this usually because you provide
conv params that may ignore some input points. Example:
    spatial_shape=[8, 200, 200]
    ksize=3
    stride=2
    padding=[0, 1, 1]
    dilation=1
    Coordinates=[[0, 7, 153, 142]]
these params will cause ALL points in z == 7 dropped because of padding_z=0.
enlarge your spatial shape or change your conv param to make sure
every input point has a corresponding output point.
Your Conv Params:
    spatial_shape={}
    ksize={}
    stride={}
    padding={}
    dilation={}"""

_DTYPES = {torch.float32: _lib.DTYPE_F32, torch.float16: _lib.DTYPE_F16,
           torch.bfloat16: _lib.DTYPE_BF16}


class Activation:
    """tv.gemm.Activation values used by the reference's conv modules."""
    None_ = _lib.ACT_NONE
    ReLU = _lib.ACT_RELU
    Sigmoid = _lib.ACT_SIGMOID
    LeakyReLU = _lib.ACT_LEAKY_RELU


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    output_size = []
    for i in range(len(input_size)):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
        output_size.append(1 if kernel_size[i] == -1 else size)
    return output_size


def get_deconv_output_size(input_size, kernel_size, stride, padding, dilation, output_padding):
    output_size = []
    for i in range(len(input_size)):
        if kernel_size[i] == -1:
            raise ValueError("deconv don't support kernel_size < 0")
        output_size.append((input_size[i] - 1) * stride[i] - 2 * padding[i] + kernel_size[i]
                           + output_padding[i])
    return output_size


# ---------------------------------------------------------------- plumbing
def _require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise NotImplementedError(
            f"spconv_amd runs on MI355X only: {what} is on {t.device}. There is no CPU path "
            f"(move the SparseConvTensor to 'cuda').")


def _stream(t: torch.Tensor) -> int:
    """Raw hipStream_t of torch's current stream on t's device (the C call, not the Stream object:
    this runs for every kernel launch)."""
    return torch._C._cuda_getCurrentRawStream(t.device.index)


def _on_device(fn):
    """Runs a driver with the first tensor argument's device current (torch's DeviceGuard convention):
    kernels, fills and scratch allocations of a call all belong to the device the data lives on, also
    when the caller never called torch.cuda.set_device (model on cuda:1)."""
    @functools.wraps(fn)
    def guarded(*args, **kwargs):
        t = next((a for a in args if isinstance(a, torch.Tensor)), None)
        if t is not None and t.is_cuda and t.device.index != torch.cuda.current_device():
            with torch.cuda.device(t.device):
                return fn(*args, **kwargs)
        return fn(*args, **kwargs)
    return guarded


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# SPX_OUT_CACHED (include/spconv_amd.h): inside this context igemm_fwd asks for result rows that stay in
# the caches -- SparseSequential opens it around a convolution whose output a BatchNorm reads next
_OUT_CACHED = 0x100
_TILE_ORDER = 0x200           # SPX_TILE_ORDER: tables of the int8 forward are stored in tile order
_ROWS_LAYOUT_ACT = 0x400      # SPX_ROWS_LAYOUT_ACT: `argsort` of the int8 forward is a rows layout blob
_SPARSE_HINT = 0x800          # SPX_SPARSE_HINT: the host has seen class word 1 (launch-shape hint)
_out_policy = threading.local()


@contextlib.contextmanager
def output_stays_cached():
    prev = getattr(_out_policy, "cached", False)
    _out_policy.cached = True
    try:
        yield
    finally:
        _out_policy.cached = prev


# BatchNorm statistics out of the convolution's epilogue (include/spconv_amd.h: spx_igemm_fwd_stats).  SparseSequential
# opens `collect_bn_stats()` around a training-mode convolution that a plain BatchNorm1d follows; the first plain
# igemm_fwd inside it (no bias, no activation, kernel volume <= 32, MFMA shapes) leaves the per-workgroup {rows, mean, M2}
# records in the sink, and norm.batch_norm starts at its merge step.  Anything else leaves the sink empty and the
# normalisation layer runs its own statistics pass.  SPCONV_AMD_BN_EPILOGUE=0 switches it off.
BN_EPILOGUE = os.environ.get("SPCONV_AMD_BN_EPILOGUE", "1") != "0"
_stats_req = threading.local()


class StatsSink:
    __slots__ = ("n_live", "records", "count", "rows", "channels")

    def __init__(self):
        self.n_live, self.records, self.count, self.rows, self.channels = None, None, 0, 0, 0


@contextlib.contextmanager
def collect_bn_stats():
    sink = StatsSink()
    prev = getattr(_stats_req, "sink", None)
    _stats_req.sink = sink
    try:
        yield sink
    finally:
        _stats_req.sink = prev


def current_stats_sink() -> Optional[StatsSink]:
    return getattr(_stats_req, "sink", None)


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty((max(int(nbytes), 16),), dtype=torch.uint8, device=device)


def _dtype_code(t: torch.Tensor) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise NotImplementedError(f"unsupported feature dtype {t.dtype}") from None


def _kv(ksize) -> int:
    return int(functools.reduce(lambda a, b: a * b, ksize, 1))


# ---------------------------------------------------------------- rulebook
@_on_device
def build_rulebook(indices: torch.Tensor, batch_size: int, spatial_shape: List[int],
                   ksize: List[int], stride: List[int], padding: List[int],
                   dilation: List[int], out_padding: List[int], subm: bool = False,
                   transpose: bool = False, need_bwd_table: bool = False,
                   do_sort=False, need_native: bool = True,
                   num_out_act_bound: int = -1, static_num_out: int = 0,
                   pred_key=None, out_order: str = "first_seen") -> Tuple[Rulebook, List[int]]:
    """One call builds every artefact (dense tables, masks, Native lists).  need_native=False
    (inference) leaves the ConvAlgo.Native lists out -- three launches and two thirds of the
    fill traffic -- and the Rulebook derives them from the tables if they are asked for later.

    static_num_out > 0 (regular / transposed convolution, inference): the STATIC-SHAPE form -- every
    output tensor has static_num_out rows, nothing is read back from the device (the whole build can
    be captured in a graph), rows past the real output count are dead (out_indices -1, no pairs) and
    `rb.n_out_dev` holds {outputs found, table overflow} on the device.  Input rows with a negative
    batch index are dead rows in every mode.

    do_sort: False = rows stay in input order; "layout" (the modules' default; "auto" is an alias) =
    density-aware rows layout of a SubM rulebook, built on the device (rows_layout); True = the
    reference's explicit mask sort (sort_rulebook).

    out_order (regular convolution): "first_seen" = the CPU reference's numbering of the outputs (indices.py:1742-1771);
    "sorted" = ascending linear coordinate key, the order of the reference's GPU sort + unique path
    (all.py:1533-1552), built through the level's RANK MAP (include/spconv_amd.h, sorted-order levels) where the geometry
    allows it (k3 s2, k2 s2, ...; otherwise first seen).  The map travels with rb.out_indices (`_spx_rankmap`): a SubM
    build over exactly that tensor reads its neighbours from it instead of hashing the coordinates again."""
    _require_gpu(indices, "indices")
    assert indices.dtype == torch.int32 and indices.ndim == 2
    L = _lib.load()
    indices = indices.contiguous()
    dev = indices.device
    n_in, ndim = indices.shape[0], indices.shape[1] - 1
    kv = _kv(ksize)
    words = (kv + 31) // 32
    if subm:
        out_shape = list(spatial_shape)
    elif transpose:
        out_shape = get_deconv_output_size(spatial_shape, ksize, stride, padding, dilation, out_padding)
    else:
        out_shape = get_conv_output_size(spatial_shape, ksize, stride, padding, dilation)
    if any(x <= 0 for x in out_shape):
        raise ValueError(f"your out spatial shape {out_shape} reach zero!!! input shape: {spatial_shape}")
    stream = _stream(indices)
    i32 = dict(dtype=torch.int32, device=dev)
    if subm:
        # the -1 filled tables live in ONE buffer, so that the library needs a single fill
        tb = kv * n_in
        nat = 2 if need_native else 0
        buf = torch.empty(((1 + nat + (1 if need_bwd_table else 0)) * tb,), **i32)
        pair_fwd = buf[:tb].view(kv, n_in)
        native = buf[tb:3 * tb].view(2, kv, n_in) if need_native else None
        pair_bwd = buf[(1 + nat) * tb:].view(kv, n_in) if need_bwd_table else None
        mask = torch.empty((n_in, words), **i32)
        num = torch.empty((kv,), **i32) if need_native else None
        rm = _rankmap_of(indices, batch_size, spatial_shape, n_in, kv)
        if rm is not None:
            # rows in key order with their level's rank map (a sorted-order strided layer built them): no table
            ws = _ws(L.spx_subm_rulebook_ranked_ws_bytes(n_in, kv), dev)
            _lib.check(L.spx_subm_rulebook_ranked(indices.data_ptr(), n_in, ndim, batch_size,
                                                  _lib.ints(spatial_shape), _lib.ints(ksize),
                                                  _lib.ints(dilation), pair_fwd.data_ptr(), _ptr(pair_bwd),
                                                  mask.data_ptr(), _ptr(native), _ptr(num),
                                                  rm.data_ptr(), rm.numel() * 4, ws.data_ptr(), ws.numel(), stream))
        else:
            ws = _ws(L.spx_subm_rulebook_ws_bytes(n_in, kv), dev)
            _lib.check(L.spx_subm_rulebook(indices.data_ptr(), n_in, ndim, batch_size,
                                           _lib.ints(spatial_shape), _lib.ints(ksize),
                                           _lib.ints(dilation), pair_fwd.data_ptr(), _ptr(pair_bwd),
                                           mask.data_ptr(), _ptr(native), _ptr(num),
                                           ws.data_ptr(), ws.numel(), stream))
        rb = Rulebook(indices, pair_fwd, pair_bwd, mask, mask, native, num, n_in, n_in, kv, True)
    else:
        args = (_lib.ints(spatial_shape), _lib.ints(out_shape), _lib.ints(ksize), _lib.ints(stride),
                _lib.ints(padding), _lib.ints(dilation), int(transpose))
        if out_order not in ("first_seen", "sorted"):
            raise ValueError(f"out_order must be 'first_seen' or 'sorted', got {out_order!r}")
        if (out_order == "sorted" and n_in > 0 and L.spx_conv_sorted_ok(ndim, batch_size, *args)
                and _sorted_pays(n_in, batch_size, out_shape)):
            return _build_sorted(L, indices, batch_size, spatial_shape, out_shape, ksize, stride, padding, dilation,
                                 args[:-1], need_native, num_out_act_bound, static_num_out, pred_key, stream,
                                 do_sort)
        ws = _ws(L.spx_conv_rulebook_ws_bytes(n_in, ndim, _lib.ints(ksize), _lib.ints(stride),
                                              _lib.ints(dilation), int(transpose)), dev)
        if static_num_out > 0:
            n_out = int(static_num_out)
            out_indices = torch.empty((n_out, ndim + 1), **i32)
            pair_fwd = torch.empty((kv, n_out), **i32)
            pair_bwd = torch.empty((kv, n_in), **i32)
            mask_fwd = torch.empty((n_out, words), **i32)
            mask_bwd = torch.empty((n_in, words), **i32)
            n_out_dev = torch.empty((2,), **i32)
            native = torch.empty((2, kv, n_in), **i32) if need_native else None
            num = torch.empty((kv,), **i32) if need_native else None
            _lib.check(L.spx_conv_rulebook_static(indices.data_ptr(), n_in, ndim, batch_size, *args, n_out,
                                                  out_indices.data_ptr(), pair_fwd.data_ptr(),
                                                  pair_bwd.data_ptr(), mask_fwd.data_ptr(),
                                                  mask_bwd.data_ptr(), _ptr(native), _ptr(num),
                                                  n_out_dev.data_ptr(), ws.data_ptr(), ws.numel(), stream))
            rb = Rulebook(out_indices, pair_fwd, pair_bwd, mask_fwd, mask_bwd, native, num, n_in, n_out,
                          kv, False)
            rb.n_out_dev = n_out_dev
            rb.in_indices, rb.in_shape, rb.out_shape, rb.batch_size = (indices, list(spatial_shape),
                                                                       list(out_shape), batch_size)
            return rb, out_shape
        n_out_c = ctypes.c_int(0)
        _lib.check(L.spx_conv_rulebook_count(indices.data_ptr(), n_in, ndim, batch_size, *args,
                                             ws.data_ptr(), ws.numel(), ctypes.byref(n_out_c),
                                             stream))
        n_out = int(n_out_c.value)
        if n_out == 0:
            raise ValueError(_POINT_VANISH_MSG.format(spatial_shape, ksize, stride, padding, dilation))
        if 0 < num_out_act_bound < n_out:
            # the reference's bounded mode (ops.py:263-266, indices.py:460-499): at most `bound` outputs
            # exist; here the FIRST `bound` in the canonical (first-seen) order survive, pairs into the
            # others are dropped
            n_out = int(num_out_act_bound)
        out_indices = torch.empty((n_out, ndim + 1), **i32)
        pair_fwd = torch.empty((kv, n_out), **i32)
        pair_bwd = torch.empty((kv, n_in), **i32)
        mask_fwd = torch.empty((n_out, words), **i32)
        mask_bwd = torch.empty((n_in, words), **i32)
        native = torch.empty((2, kv, n_in), **i32) if need_native else None
        num = torch.empty((kv,), **i32) if need_native else None
        _lib.check(L.spx_conv_rulebook_fill(indices.data_ptr(), n_in, ndim, batch_size, *args,
                                            n_out, out_indices.data_ptr(), pair_fwd.data_ptr(),
                                            pair_bwd.data_ptr(), mask_fwd.data_ptr(),
                                            mask_bwd.data_ptr(), _ptr(native), _ptr(num),
                                            ws.data_ptr(), ws.numel(), stream))
        rb = Rulebook(out_indices, pair_fwd, pair_bwd, mask_fwd, mask_bwd, native, num, n_in,
                      n_out, kv, False)
    rb.in_indices, rb.in_shape, rb.out_shape, rb.batch_size = indices, list(spatial_shape), list(out_shape), batch_size
    rb.pred_key = pred_key        # whose rulebook this is, for the density-class prediction (poll_class)
    if do_sort in ("layout", "auto"):
        # the default row order: classified and regrouped on the device inside the build (SubM; nothing
        # read back).  Strided layers keep their row order (dense by construction: every output has
        # several inputs), small rulebooks are launch-bound either way.
        if subm and 1 < kv <= 32 and n_in >= _LAYOUT_MIN_ROWS:
            if (_LAYOUT_SKIP and torch.cuda.is_current_stream_capturing()
                    and bool(getattr(pred_key, _CLS0_ATTR, False))):
                # Inside a capture the class of this module's rulebooks is what the warm-up passes saw.  Class 0 (most
                # rows have a neighbour: every level of a LiDAR backbone) uses nothing of a layout but its class word:
                # the three launches that make it are left out, the gather-GEMMs walk the row-order tables as they do for
                # small rulebooks, with the density hint the prediction gives (a scene of the other class still computes
                # the same values, on the slower walk).
                rb.mask_fwd._spx_dense = _pred_get(pred_key)
            else:
                rows_layout(rb)
    elif do_sort and words == 1:
        sort_rulebook(rb)
    return rb, out_shape


# The rank-map builder's cost grows with the GRID (one byte per cell filled and scanned, cells / 4 bytes of map kept for
# the level's SubM layers), the hash builder's with the INPUT rows (~0.45 us per 1000).  Measured break-even is a few
# hundred cells per input row (config 4: 118 cells per row at level 1 -> 2, 20 at level 2 -> 3; 174 -> 118 us); beyond
# `_SORTED_MAX_CELLS_PER_ROW` cells per row, or 1 GiB of scratch, the first-seen hash build is taken instead (round-5
# ADVICE: a 500^3 x 2 grid with 50 k voxels would fill and scan 250 MB per build).  0 = no gate.
_SORTED_MAX_CELLS_PER_ROW = float(os.environ.get("SPCONV_AMD_SORTED_MAX_CELLS_PER_ROW", "512"))
_SORTED_MAX_SCRATCH = 1 << 30


def _sorted_pays(n_in: int, batch_size: int, out_shape) -> bool:
    cells = float(batch_size)
    for d in out_shape:
        cells *= float(d)
    if _SORTED_MAX_CELLS_PER_ROW <= 0:
        return True
    return cells <= _SORTED_MAX_CELLS_PER_ROW * max(n_in, 1) and cells * 1.25 <= _SORTED_MAX_SCRATCH


def _rankmap_of(indices: torch.Tensor, batch_size: int, spatial_shape, n: int, kv: int):
    """The rank map a sorted-order build attached to exactly this index tensor, if it describes this level -- and if
    the tensor has not been written since (its version counter and storage are the ones recorded with the map: an
    in-place edit of `out.indices` between the strided layer and the SubM layer behind it drops the map, round-5
    ADVICE)."""
    rm = getattr(indices, "_spx_rankmap", None)
    if rm is None or not (1 < kv <= 128) or n > (4 << 20):
        return None
    cells, bs, shape, rows = rm[:4]
    if bs != batch_size or shape != tuple(int(v) for v in spatial_shape) or rows != n or cells.device != indices.device:
        return None
    if len(rm) >= 6 and (rm[4] != indices._version or rm[5] != indices.data_ptr()):
        indices._spx_rankmap = None
        return None
    return cells


# Level 1 in key order.  The levels behind a strided layer are numbered by THIS library (ascending key: _build_sorted);
# the level a caller hands in is in the caller's order, and its SubM rulebook pays hash insert + probe (the largest
# rulebook build of a backbone: 160 us at 4 x 100 k voxels).  When the caller's rows ARE in ascending, unique key order
# (spconv_amd.pytorch.utils.sort_voxels_by_coordinate; a voxeliser that emits key order), `attach_rank_map` builds the
# level's rank map from them directly (row = rank: a fill + one pass, spx_rankmap_from_sorted) and leaves it on the index
# tensor exactly as a sorted-order strided build does: SubM layers over the tensor then take spx_subm_rulebook_ranked.
# The map costs grid cells / 4 bytes of fill: taken up to `_RANKMAP_MAX_CELLS_PER_ROW` cells per row and 1 GiB.
_RANKMAP_MAX_CELLS_PER_ROW = float(os.environ.get("SPCONV_AMD_RANKMAP_MAX_CELLS_PER_ROW", "4096"))


@_on_device
def attach_rank_map(indices: torch.Tensor, batch_size: int, spatial_shape, check: bool = True,
                    violation: Optional[torch.Tensor] = None) -> bool:
    """Declares `indices` [n, ndim + 1] (int32, batch index first) to be in ascending, unique coordinate-key order
    (batch-major, last axis fastest; rows with batch index -1 -- static shapes -- trail) and attaches the level's rank
    map to the tensor.  check=True reads the device-side verdict back (ONE synchronisation; a data loader's place): a
    tensor that breaks the contract is left untouched and False is returned.  check=False (inside a stream capture;
    the caller vouches for the order) never synchronises; `violation` (int32 [1] on the device) then receives the
    verdict for whoever wants to read it later (1 = the rows break the contract).  Returns whether a map was attached."""
    _require_gpu(indices, "indices")
    assert indices.dtype == torch.int32 and indices.ndim == 2 and indices.is_contiguous()
    L = _lib.load()
    n, ndim = indices.shape[0], indices.shape[1] - 1
    nbytes = int(L.spx_rankmap_bytes(ndim, batch_size, _lib.ints(spatial_shape)))
    cells = float(batch_size)
    for d in spatial_shape:
        cells *= float(d)
    if (nbytes == 0 or n == 0 or nbytes > _SORTED_MAX_SCRATCH
            or (_RANKMAP_MAX_CELLS_PER_ROW > 0 and cells > _RANKMAP_MAX_CELLS_PER_ROW * n)):
        return False
    i32 = dict(dtype=torch.int32, device=indices.device)
    cells_t = torch.empty((nbytes // 4,), **i32)
    bad = torch.empty((1,), **i32) if check else violation
    _lib.check(L.spx_rankmap_from_sorted(indices.data_ptr(), n, ndim, batch_size, _lib.ints(spatial_shape),
                                         cells_t.data_ptr(), nbytes, _ptr(bad), _stream(indices)))
    if check and int(bad.item()) != 0:
        return False
    indices._spx_rankmap = (cells_t, batch_size, tuple(int(v) for v in spatial_shape), n,
                            indices._version, indices.data_ptr())
    return True


@_on_device
def key_argsort(indices: torch.Tensor, batch_size: int, spatial_shape, want_indices: bool = True,
                rank_map: bool = False, violation: Optional[torch.Tensor] = None, rows: Optional[torch.Tensor] = None):
    """(order, indices[order]) with order[t] = the row of the t-th smallest coordinate key (batch-major, last axis
    fastest); dead rows (batch index -1) trail in their own order and come out as -1 in every column.  One C-ABI call
    (spx_key_argsort: four launches -- one stable radix pass on the upper key bits, then workgroups per bucket rank
    their rows through an occupancy map in LDS; the keys of a level are unique), nothing read back, so it can sit inside
    a stream capture -- the static runners sort their scene at the entry with it (static.py entry_sort).
    rank_map=True: the sorted index tensor leaves with the level's rank map attached, written by the same bucket pass
    (what attach_rank_map would build from it; same size gates -- beyond them the tensor stays untagged); `violation`
    (int32 [1] on the device) is raised when a coordinate occurs twice.  `rows` ([n, ...] contiguous, row size a
    multiple of 4 bytes: the level's features): a third result, rows[order], written by the same bucket pass.  Returns
    None when the key space of batch x grid does not fit 32 bits."""
    _require_gpu(indices, "indices")
    assert indices.dtype == torch.int32 and indices.ndim == 2 and indices.is_contiguous()
    cells = int(batch_size)
    for d in spatial_shape:
        cells *= int(d)
    if cells > 0xffe00000:
        return None
    L = _lib.load()
    n, ndim = indices.shape[0], indices.shape[1] - 1
    order = torch.empty((n,), dtype=torch.int32, device=indices.device)
    out = torch.empty_like(indices) if (want_indices or rank_map) else None
    cells_t = None
    if rank_map and n > 0:
        nbytes = int(L.spx_rankmap_bytes(ndim, batch_size, _lib.ints(spatial_shape)))
        if not (nbytes == 0 or nbytes > _SORTED_MAX_SCRATCH
                or (_RANKMAP_MAX_CELLS_PER_ROW > 0 and cells > _RANKMAP_MAX_CELLS_PER_ROW * n)):
            cells_t = torch.empty((nbytes // 4,), dtype=torch.int32, device=indices.device)
    ws = _ws(L.spx_key_argsort_ws_bytes(n), indices.device)
    rows_sorted, row_bytes = None, 0
    if rows is not None:
        assert rows.is_cuda and rows.is_contiguous() and rows.shape[0] == n
        row_bytes = (rows.numel() // max(n, 1)) * rows.element_size()
        assert row_bytes % 4 == 0, "rows carried by the sort: row size must be a multiple of 4 bytes"
        rows_sorted = torch.empty_like(rows)
    _lib.check(L.spx_key_argsort(indices.data_ptr(), n, ndim, int(batch_size), _lib.ints(spatial_shape),
                                 order.data_ptr(), _ptr(out), _ptr(cells_t), 0 if cells_t is None else cells_t.numel() * 4,
                                 _ptr(violation), _ptr(rows), _ptr(rows_sorted), row_bytes,
                                 ws.data_ptr(), ws.numel(), _stream(indices)))
    if cells_t is not None:
        out._spx_rankmap = (cells_t, batch_size, tuple(int(v) for v in spatial_shape), n, out._version, out.data_ptr())
    return (order, out) if rows is None else (order, out, rows_sorted)


def _build_sorted(L, indices, batch_size, spatial_shape, out_shape, ksize, stride, padding, dilation, args,
                  need_native, num_out_act_bound, static_num_out, pred_key, stream, do_sort=False):
    """Regular-convolution rulebook with the outputs in key order (spx_conv_rulebook_*_sorted)."""
    dev = indices.device
    n_in, ndim = indices.shape[0], indices.shape[1] - 1
    kv = _kv(ksize)
    words = (kv + 31) // 32
    i32 = dict(dtype=torch.int32, device=dev)
    cells = torch.empty((L.spx_rankmap_bytes(ndim, batch_size, _lib.ints(out_shape)) // 4,), **i32)
    ws = _ws(L.spx_conv_rulebook_sorted_ws_bytes(n_in, ndim, batch_size, _lib.ints(out_shape), _lib.ints(ksize)), dev)
    n_out_dev = None
    if static_num_out > 0:
        n_out = int(static_num_out)
        n_out_dev = torch.empty((4,), **i32)      # {found, 0, live rows = min(found, bound), -}
    else:
        n_out_c = ctypes.c_int(0)
        _lib.check(L.spx_conv_rulebook_count_sorted(indices.data_ptr(), n_in, ndim, batch_size, *args,
                                                    cells.data_ptr(), cells.numel() * 4, ws.data_ptr(), ws.numel(),
                                                    ctypes.byref(n_out_c), stream))
        n_out = int(n_out_c.value)
        if n_out == 0:
            raise ValueError(_POINT_VANISH_MSG.format(spatial_shape, ksize, stride, padding, dilation))
        if 0 < num_out_act_bound < n_out:
            n_out = int(num_out_act_bound)        # (the outputs with the smallest keys survive)
    out_indices = torch.empty((n_out, ndim + 1), **i32)
    pair_fwd = torch.empty((kv, n_out), **i32)
    pair_bwd = torch.empty((kv, n_in), **i32)
    mask_fwd = torch.empty((n_out, words), **i32)
    mask_bwd = torch.empty((n_in, words), **i32)
    native = torch.empty((2, kv, n_in), **i32) if need_native else None
    num = torch.empty((kv,), **i32) if need_native else None
    outs = (out_indices.data_ptr(), pair_fwd.data_ptr(), pair_bwd.data_ptr(), mask_fwd.data_ptr(),
            mask_bwd.data_ptr(), _ptr(native), _ptr(num))
    if static_num_out > 0:
        _lib.check(L.spx_conv_rulebook_static_sorted(indices.data_ptr(), n_in, ndim, batch_size, *args, n_out, *outs,
                                                     n_out_dev.data_ptr(), cells.data_ptr(), cells.numel() * 4,
                                                     ws.data_ptr(), ws.numel(), stream))
    else:
        _lib.check(L.spx_conv_rulebook_fill_sorted(indices.data_ptr(), n_in, ndim, batch_size, *args, n_out, *outs,
                                                   cells.data_ptr(), cells.numel() * 4, ws.data_ptr(), ws.numel(),
                                                   stream))
    rb = Rulebook(out_indices, pair_fwd, pair_bwd, mask_fwd, mask_bwd, native, num, n_in, n_out, kv, False)
    if n_out_dev is not None:
        rb.n_out_dev = n_out_dev[:2]
        rb.out_n_live_dev = n_out_dev[2:3]        # written by the build itself: no clamp launch behind it
    rb.in_indices, rb.in_shape, rb.out_shape, rb.batch_size = indices, list(spatial_shape), list(out_shape), batch_size
    rb.pred_key = pred_key
    rb.rankmap = cells
    out_indices._spx_rankmap = (cells, batch_size, tuple(int(v) for v in out_shape), n_out,
                                out_indices._version, out_indices.data_ptr())
    if do_sort is True and words == 1:
        sort_rulebook(rb)         # SPCONV_DO_SORT=1: "explicit mask sort of every rulebook" holds on this path too
    return rb, out_shape


# Density-aware row order = the DEFAULT of the layer modules (the reference sorts every rulebook by mask:
# SPCONV_DO_SORT = "1", constants.py:121, ops.py:346,550,763-785).  spx_subm_layout classifies the finished
# masks on the device and, for a SPARSE rulebook (fewer than a quarter of the rows have any neighbour),
# moves the rows with a neighbour into a compact appendix grouped by offset (a stable counting partition),
# which turns "every 128-row tile walks ~4 extra offsets" into "97 % of the rows run as a plain streaming
# GEMM on their centre pair -- no row order, no pair word to fetch -- and ~25 appendix tiles walk 2-3
# offsets each"; a dense (LiDAR) rulebook keeps everything in the row-order walk (regrouping loses 19 %
# there).  No sort, no read-back: class and count are words in the blob that the appendix workgroups of
# the gather-GEMM launch read, so the same launch serves both classes and the whole thing sits in a hipGraph.
_LAYOUT_MIN_ROWS = 32768
_ROWS_LAYOUT = 2              # SPX_ROWS_LAYOUT: `argsort` of a gather-GEMM call is a layout blob


@_on_device
def rows_layout(rb: Rulebook) -> None:
    """Builds rb.layout (int32 blob, include/spconv_amd.h: spx_subm_layout) for a SubM rulebook."""
    L = _lib.load()
    n, kv = rb.n_out, rb.kv
    dev = rb.pair_fwd.device
    blob = torch.empty((L.spx_subm_layout_bytes(n, kv) // 4,), dtype=torch.int32, device=dev)
    ws = _ws(L.spx_subm_layout_ws_bytes(n), dev)
    _lib.check(L.spx_subm_layout(rb.pair_fwd.data_ptr(), rb.mask_fwd.data_ptr(), n, kv, blob.data_ptr(),
                                 ws.data_ptr(), ws.numel(), _stream(rb.pair_fwd)))
    rb.layout = blob
    rb.sort_decided = True
    rb.sparse_class, rb.heavy_rows = None, 0      # (what was known belongs to the blob this one replaces)
    _request_class(rb)


# ---- the density class on the HOST, without a synchronisation ------------------------------------------------------
# A launch SHAPE cannot be picked on the device: the weight-stationary gather-GEMM of dense C = K = 64 layers
# (csrc/igemm_ws.hip, 512-row workgroups) and the int8 tile height are host choices.  The class word of a rows layout
# is therefore copied to pinned host memory asynchronously when the layout is built and looked at (never waited for)
# when a launch is prepared: until it has arrived -- the host usually runs ahead of the device on the first layer that
# uses a new rulebook -- the launch goes by what the same layer's previous rulebook turned out to be (an attribute of
# the owning module, `_pred_get`), and inside a stream capture, where nothing can be polled, by that prediction alone.
# Both kernels give bit-identical results, so neither a late answer nor a wrong prediction shows in an output.
_DENSE_HINT = 0x100           # SPX_DENSE_HINT (include/spconv_amd.h)
_WS_MIN_ROWS = 98304          # one 512-row workgroup per CU: below ~3/4 of 256 x 512 rows the 128-row tiles fill the chip better
_CLASS_SLOTS = 1024
_class_ring = {}              # device index -> [pinned int32 [_CLASS_SLOTS, 2], next slot, owner tokens]
_PRED_ATTR = "_spx_dense_pred"     # on the owning module (rb.pred_key): the rulebook built for it last time was dense
_CLS0_ATTR = "_spx_class0_pred"    # ... and its rows layout was of class 0 (no appendix)
_LAYOUT_SKIP = os.environ.get("SPCONV_AMD_LAYOUT_SKIP", "1") != "0"


def _pred_get(key) -> bool:
    """The prediction lives ON the module that owns the rulebooks (round-5 ADVICE: a dict keyed by id(module) outlives
    the module and can be answered by another object that reuses the id)."""
    return bool(getattr(key, _PRED_ATTR, False)) if key is not None else False


def _pred_set(key, dense: bool) -> None:
    if key is not None:
        try:
            object.__setattr__(key, _PRED_ATTR, bool(dense))
        except (AttributeError, TypeError):       # (a key that cannot carry attributes: no prediction)
            pass


class _ClassRequest:
    """One asynchronous read of a rows layout's class word (see above)."""
    __slots__ = ("ring", "slot", "token", "event", "key", "n", "rb", "done")

    def resolve(self) -> bool:
        """Non-blocking; True once the request is settled (answer taken, or lost to a wrapped ring)."""
        if self.done:
            return True
        if self.ring[2][self.slot] is not self.token:     # the ring wrapped before anybody looked: stay with the prediction
            self.done = True
            return True
        if not self.event.query():
            return False
        cls, heavy = (int(v) for v in self.ring[0][self.slot])
        self.done = True
        dense = (not cls) and self.n >= _WS_MIN_ROWS and 4 * heavy >= 3 * self.n
        _pred_set(self.key, dense)
        if self.key is not None and heavy > 0:      # (no row with a neighbour -- the empty scene a runner warms up on -- says nothing)
            try:
                object.__setattr__(self.key, _CLS0_ATTR, (not cls) and 4 * heavy >= self.n)
            except (AttributeError, TypeError):
                pass
        rb = self.rb()
        if rb is not None and rb.layout is not None:
            rb.sparse_class, rb.heavy_rows = bool(cls), heavy
            rb.layout._spx_dense = dense
            rb.layout._spx_heavy = heavy
        return True


_class_pending = collections.deque()


def _request_class(rb: Rulebook) -> None:
    blob = rb.layout
    rb._class_req = None
    if blob is None:
        return
    capturing = torch.cuda.is_current_stream_capturing()
    if not capturing:
        # answers to earlier requests (a layer that uses every rulebook once, right behind its build, never comes back
        # to ask: the next build does it for it)
        while _class_pending and _class_pending[0].resolve():
            _class_pending.popleft()
        if len(_class_pending) > 64:
            _class_pending.popleft()
    blob._spx_dense = _pred_get(getattr(rb, "pred_key", None))
    if capturing:
        return
    ring = _class_ring.get(blob.device.index)
    if ring is None:
        ring = [torch.zeros((_CLASS_SLOTS, 2), dtype=torch.int32).pin_memory(), 0, [None] * _CLASS_SLOTS]
        _class_ring[blob.device.index] = ring
    host, slot = ring[0], ring[1]
    ring[1] = (slot + 1) % _CLASS_SLOTS
    req = _ClassRequest()
    req.ring, req.slot, req.token, req.done = ring, slot, object(), False
    req.key, req.n, req.rb = getattr(rb, "pred_key", None), rb.n_out, weakref.ref(rb)
    ring[2][slot] = req.token
    host[slot].copy_(blob[:2], non_blocking=True)
    req.event = torch.cuda.Event()
    req.event.record()
    rb._class_req = req
    _class_pending.append(req)


def poll_class(rb: Optional[Rulebook]) -> None:
    """Non-blocking: takes the class word of rb's rows layout if its copy has arrived (see above)."""
    if rb is None or rb.layout is None or rb.sparse_class is not None:
        return
    req = getattr(rb, "_class_req", None)
    if req is None or torch.cuda.is_current_stream_capturing():
        return
    if req.resolve():
        rb._class_req = None


def _with_dense_hint(tile_order: int, argsort, mask=None) -> int:
    if tile_order == _ROWS_LAYOUT and getattr(argsort, "_spx_dense", False):
        return tile_order | _DENSE_HINT
    if tile_order == 0 and argsort is None and getattr(mask, "_spx_dense", False):
        return _DENSE_HINT               # (a rulebook whose layout was left out: build_rulebook)
    return tile_order


def layout_views(rb: Rulebook):
    """(header [5]: class, M, n, kv, mcap; main mask words [n]; appendix row list [mcap]; appendix mask words [mcap];
    appendix pair table [kv, mcap]) of rb.layout -- views for tests and tools; only the first M appendix entries are
    defined, and only for a class-1 rulebook."""
    n, kv, blob = rb.n_out, rb.kv, rb.layout
    npad = (n + 63) // 64 * 64
    mcap = int(_lib.load().spx_subm_layout_mcap(n))
    o = 64 + npad
    return (blob[:5], blob[64:64 + n], blob[o:o + mcap], blob[o + mcap:o + 2 * mcap],
            blob[o + 2 * mcap:o + (2 + kv) * mcap].view(kv, mcap))


def sparse_neighbourhoods(rb: Rulebook) -> bool:
    """True for a rulebook whose rows were regrouped (one device -> host read of the class word; cached
    on the rulebook).  A HOST-side view for tools and for choices that change a launch's shape (the
    int8 tile height); the kernels themselves read the class on the device."""
    cached = getattr(rb, "sparse_class", None)
    if cached is not None:
        return cached
    ok = False
    if rb.layout is not None:
        head = rb.layout[:2].tolist()
        ok, rb.heavy_rows = bool(head[0]), int(head[1])
        rb.layout._spx_dense = (not ok) and rb.n_out >= _WS_MIN_ROWS and 4 * rb.heavy_rows >= 3 * rb.n_out
        rb.layout._spx_heavy = rb.heavy_rows
        rb._class_req = None
    elif rb.subm and 1 < rb.kv <= 32 and rb.n_out >= _LAYOUT_MIN_ROWS and rb.mask_fwd is not None:
        centre = 1 << (rb.kv // 2)
        ok = float((rb.mask_fwd.view(-1) != centre).float().mean().item()) < 0.25
    rb.sparse_class = ok
    return ok


def sort_rulebook(rb: Rulebook) -> None:
    """The reference's explicit mask sort (SPCONV_DO_SORT=1): argsort of the mask words + copies of the
    tables in that order (both directions of a regular-conv rulebook).  The gather-GEMM then reads
    pair / mask by tile position."""
    dev = rb.pair_fwd.device
    if dev.type == "cuda" and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):                   # launches belong to the device the tables live on
            return sort_rulebook(rb)
    L = _lib.load()
    rb.sort_decided = True
    for which in (("fwd",) if rb.subm else ("fwd", "bwd")):
        pair, mask = (rb.pair_fwd, rb.mask_fwd) if which == "fwd" else (rb.pair_bwd, rb.mask_bwd)
        if pair is None or mask is None or mask.shape[1] != 1 or pair.shape[1] == 0:
            continue
        order = mask_argsort(mask, rb.kv)
        pair_t, mask_t = torch.empty_like(pair), torch.empty_like(mask)
        _lib.check(L.spx_permute_tables(pair.data_ptr(), mask.data_ptr(), order.data_ptr(), pair.shape[0],
                                        pair.shape[1], mask.shape[1], pair_t.data_ptr(), mask_t.data_ptr(),
                                        _stream(pair)))
        if which == "fwd":
            rb.argsort_fwd = order
        else:
            rb.argsort_bwd = order
        rb.sorted_tables[which] = (pair_t, mask_t)


def tables_of(rb: Rulebook, which: str, cout: int = 64):
    """(pair, mask, argsort, tile_order) the gather-GEMM should read for `which` ("fwd": pair_fwd
    over the output rows, "bwd": pair_bwd over the input rows).  tile_order 0: tables by row (argsort,
    if any, permutes the rows); 1: copies in tile order (explicit mask sort); 2: `argsort` is the rows
    layout blob and pair / mask the row-order tables (the device picks).  `cout`: output width of the
    GEMM (widths beyond the MFMA instantiations take the generic kernel, which reads the tables by row)."""
    if cout > _MFMA_COUT[-1] or rb.kv > 32:
        return ((rb.pair_fwd, rb.mask_fwd, None, 0) if which == "fwd"
                else (rb.pair_bwd, rb.mask_bwd, None, 0))
    if which == "fwd":
        pair, mask, order = rb.pair_fwd, rb.mask_fwd, rb.argsort_fwd
    else:
        pair, mask, order = rb.pair_bwd, rb.mask_bwd, rb.argsort_bwd
    st = rb.sorted_tables.get(which)
    if order is not None and st is not None:
        return st[0], st[1], order, 1
    if order is None and which == "fwd" and rb.subm and rb.layout is not None:
        poll_class(rb)
        return pair, mask, rb.layout, _ROWS_LAYOUT
    return pair, mask, order, 0


@_on_device
def mask_argsort(mask: torch.Tensor, kv: int = 0) -> torch.Tensor:
    """SpconvOps.sort_1d_by_key_allocator (all.py:935-991): stable argsort of the mask words.  kv (1..32): the kernel
    volume the words belong to -- only their low kv bits can be set, the sort runs fewer digit passes."""
    L = _lib.load()
    n, words = mask.shape
    out = torch.empty((n,), dtype=torch.int32, device=mask.device)
    ws = _ws(L.spx_mask_argsort_ws_bytes(n), mask.device)
    if 1 <= int(kv) <= 32 and words == 1:
        _lib.check(L.spx_mask_argsort_kv(mask.data_ptr(), n, int(kv), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                         _stream(mask)))
    else:
        _lib.check(L.spx_mask_argsort(mask.data_ptr(), n, words, out.data_ptr(), ws.data_ptr(),
                                      ws.numel(), _stream(mask)))
    return out


def attach_rulebook(t: torch.Tensor, rb: Rulebook) -> torch.Tensor:
    """Side channel for the reference-shaped op signatures below: a bare pair tensor
    remembers (weakly, to avoid a tensor<->rulebook cycle) the rulebook it belongs to."""
    t._spx_rulebook = weakref.ref(rb)
    return t


def rulebook_of(t: torch.Tensor) -> Optional[Rulebook]:
    ref = getattr(t, "_spx_rulebook", None)
    return ref() if ref is not None else None


_attach = attach_rulebook


def get_indice_pairs(indices: torch.Tensor, batch_size: int, spatial_shape: List[int],
                     algo: ConvAlgo, ksize: List[int], stride: List[int], padding: List[int],
                     dilation: List[int], out_padding: List[int], subm: bool = False,
                     transpose: bool = False, num_out_act_bound: int = -1):
    """Returns (out_inds, pair [2, kv, N_in], indice_num_per_loc [kv]) -- ConvAlgo.Native layout."""
    rb, _ = build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation,
                           out_padding, subm, transpose, num_out_act_bound=num_out_act_bound)
    return rb.out_indices, _attach(rb.pair_native, rb), rb.num_per_loc


def get_indice_pairs_implicit_gemm(indices: torch.Tensor, batch_size: int,
                                   spatial_shape: List[int], algo: ConvAlgo, ksize: List[int],
                                   stride: List[int], padding: List[int], dilation: List[int],
                                   out_padding: List[int], subm: bool = False,
                                   transpose: bool = False, is_train: bool = True, alloc=None,
                                   timer=None, num_out_act_bound: int = -1,
                                   direct_table: bool = True, do_sort: bool = SPCONV_DO_SORT):
    """Returns the reference's 9-tuple (ops.py:347-359):
    (out_inds, num_inds_per_loc, pair_fwd, pair_bwd, pair_mask_fwd_splits, pair_mask_bwd_splits,
     mask_argsort_fwd_splits, mask_argsort_bwd_splits, masks)."""
    if algo not in (ConvAlgo.MaskImplicitGemm, ConvAlgo.MaskSplitImplicitGemm):
        raise ValueError(f"get_indice_pairs_implicit_gemm builds the masked implicit-GEMM tables; algo {algo} "
                         f"(ConvAlgo.Native) goes through get_indice_pairs")
    kv = _kv(ksize)
    rb, _ = build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation,
                           out_padding, subm, transpose, need_bwd_table=subm and is_train,
                           do_sort=do_sort if kv <= 32 else False, num_out_act_bound=num_out_act_bound)
    masks = [np.array([0xffffffff], dtype=np.uint32)]
    arg_fwd = rb.argsort_fwd if rb.argsort_fwd is not None else torch.arange(
        rb.n_out, dtype=torch.int32, device=indices.device)
    pair_fwd = _attach(rb.pair_fwd, rb)
    if subm:
        pair_bwd = rb.pair_bwd if (is_train and rb.pair_bwd is not None) else torch.Tensor()
        return (rb.out_indices, rb.num_per_loc, pair_fwd, pair_bwd, [rb.mask_fwd], [], [arg_fwd],
                [], masks)
    arg_bwd = rb.argsort_bwd if rb.argsort_bwd is not None else torch.arange(
        rb.n_in, dtype=torch.int32, device=indices.device)
    return (rb.out_indices, rb.num_per_loc, pair_fwd, rb.pair_bwd, [rb.mask_fwd], [rb.mask_bwd],
            [arg_fwd], [arg_bwd], masks)


# ------------------------------------------------------------ conv primitives
def _check_feat(features: torch.Tensor, filters: torch.Tensor):
    _require_gpu(features, "features")
    if features.dtype != filters.dtype:
        raise TypeError(f"features ({features.dtype}) and filters ({filters.dtype}) must share a dtype")
    if features.dtype in (torch.int8, torch.qint8):
        raise NotImplementedError("int8 tensors go through implicit_gemm / igemm_fwd_int8 "
                                  "(inference only, like the reference)")


_OUT_CODES = {torch.int8: _lib.DTYPE_I8, torch.float16: _lib.DTYPE_F16,
              torch.bfloat16: _lib.DTYPE_BF16, torch.float32: _lib.DTYPE_F32}


def _int_repr(t: torch.Tensor) -> torch.Tensor:
    return t.int_repr() if t.is_quantized else t


@_on_device
def igemm_fwd_int8(features: torch.Tensor, filters: torch.Tensor, pair: torch.Tensor,
                   mask: Optional[torch.Tensor], argsort: Optional[torch.Tensor], n_out: int,
                   identity_k: int = -1, scale: Optional[torch.Tensor] = None,
                   bias: Optional[torch.Tensor] = None, add: Optional[torch.Tensor] = None,
                   add_scale: float = 0.0, out_dtype: torch.dtype = torch.int8,
                   act_type: int = Activation.None_, act_alpha: float = 0.0,
                   tile_order: int = 0, sparse_hint: bool = False, hint_rows: int = 0) -> torch.Tensor:
    """int8 inference forward (i32 accumulate on v_mfma_i32_16x16x64_i8):
    ``v = acc * scale[k] + bias[k] + add * add_scale; v = act(v)``; int8 output =
    ``clip(round_half_even(v), -128, 127)`` (reference numerics test/test_all_algo.py:272-287)."""
    _require_gpu(features, "features")
    L = _lib.load()
    features = _int_repr(features).contiguous()
    filters = _int_repr(filters).contiguous()
    if features.dtype != torch.int8 or filters.dtype != torch.int8:
        raise TypeError("igemm_fwd_int8 needs int8 features and filters")
    K0, C0 = filters.shape[0], filters.shape[-1]
    kv = filters.numel() // (K0 * C0)
    assert features.shape[1] == C0, "channel size mismatch"
    # the int8 MFMA kernel takes reduction lengths in whole 16-byte lane pieces and the output widths it
    # is instantiated for: zero-pad (a 4-channel first layer, 48 / 96-wide layers) and slice the result,
    # as igemm_fwd does for the float types
    C, K = -(-C0 // 16) * 16, (_round_cout(K0) or K0)
    features, filters = _pad_last(features, C), _pad_first(_pad_last(filters, C), K)
    out = torch.empty((n_out, K), dtype=out_dtype, device=features.device)
    f32 = lambda t: None if t is None else t.to(device=features.device, dtype=torch.float32).contiguous()
    scale, bias = f32(scale), f32(bias)
    if K != K0:
        scale = None if scale is None else torch.nn.functional.pad(scale.reshape(-1), (0, K - K0), value=1.0)
        bias = None if bias is None else torch.nn.functional.pad(bias.reshape(-1), (0, K - K0))
    if add is not None:
        add = _int_repr(add).contiguous()
        assert add.dtype == torch.int8 and tuple(add.shape) == (n_out, K0)
        add = _pad_last(add, K)
    if sparse_hint and int(tile_order) == _ROWS_LAYOUT:
        # The hint sizes the appendix part of the grid: rows past it would never be computed (round-4 ADVICE).  It is
        # therefore taken from what has been READ from this very blob (poll_class / sparse_neighbourhoods leave it on the
        # tensor), never from a caller's number alone: unknown -> no hint (the launch reserves n / 4 rows' worth).
        known = getattr(argsort, "_spx_heavy", None)
        if known is None:
            sparse_hint, hint_rows = False, 0
        else:
            hint_rows = max(int(hint_rows), int(known))
    _lib.check(L.spx_igemm_fwd_int8(features.data_ptr(), filters.data_ptr(), out.data_ptr(), _ptr(pair),
                                    _ptr(mask), _ptr(argsort), features.shape[0], n_out, C, K, kv,
                                    identity_k, _ptr(scale), _ptr(bias), _ptr(add), float(add_scale),
                                    _OUT_CODES[out_dtype],
                                    int(act_type) | (0 if argsort is None else
                                                     {0: 0, 1: _TILE_ORDER, 2: _ROWS_LAYOUT_ACT}[int(tile_order)])
                                    | ((_SPARSE_HINT | (min(65535, -(-int(hint_rows) // 64)) << 16)) if sparse_hint else 0),
                                    float(act_alpha), _stream(features)))
    return out if K == K0 else out[:, :K0].contiguous()


_MFMA_COUT = (16, 32, 64, 128, 256)


def _lane_mult(dtype: torch.dtype) -> int:
    """Channels per 16-byte MFMA lane piece: reduction lengths must be a multiple of this."""
    return 4 if dtype == torch.float32 else 8


def _round_cout(c: int) -> int:
    """Smallest output width the MFMA kernels are instantiated for (0: none, generic kernel)."""
    for v in _MFMA_COUT:
        if c <= v:
            return v
    return 0


def _pad_last(t: torch.Tensor, to: int) -> torch.Tensor:
    """Last dimension zero-padded to `to` entries: one launch of spx_pad_rows for contiguous CUDA tensors of 2- / 4-byte
    elements outside autograd (torch's pad is a fill + a copy: 4 of the ~190 launches of a backbone step went there)."""
    if t.shape[-1] == to:
        return t
    if (t.is_cuda and t.is_contiguous() and t.element_size() in (2, 4) and t.numel() > 0
            and not (torch.is_grad_enabled() and t.requires_grad)
            and not t.is_quantized):
        out = torch.empty(t.shape[:-1] + (to,), dtype=t.dtype, device=t.device)
        es = t.element_size()
        with torch.cuda.device(t.device):
            _lib.check(_lib.load().spx_pad_rows(t.data_ptr(), out.data_ptr(), t.numel() // t.shape[-1],
                                                t.shape[-1] * es, to * es, _stream(t)))
        return out
    return torch.nn.functional.pad(t, (0, to - t.shape[-1]))


def _pad_first(t: torch.Tensor, to: int) -> torch.Tensor:
    if t.shape[0] == to:
        return t
    pad = [0, 0] * (t.ndim - 1) + [0, to - t.shape[0]]
    return torch.nn.functional.pad(t, pad)


@_on_device
def igemm_fwd(features: torch.Tensor, filters: torch.Tensor, pair: torch.Tensor,
              mask: Optional[torch.Tensor], argsort: Optional[torch.Tensor], n_out: int,
              identity_k: int = -1, bias: Optional[torch.Tensor] = None,
              act_type: int = Activation.None_, act_alpha: float = 0.0,
              tile_order: int = 0) -> torch.Tensor:
    """out[o] = act(bias + sum_k feat[pair[k][o]] @ W[:, k, :].T); filters KRSC.  tile_order: see tables_of."""
    _check_feat(features, filters)
    L = _lib.load()
    K0, C0 = filters.shape[0], filters.shape[-1]
    assert features.shape[1] == C0, "channel size mismatch"
    kv = filters.numel() // (K0 * C0)
    # Shapes the MFMA kernels are not instantiated for (a backbone's first layer has 3-5 input
    # channels; widths like 48 or 96) are zero-padded to the next supported shape instead of
    # falling to the one-thread-per-output generic kernel (two orders of magnitude slower): the
    # extra columns cost a few bytes per row and contribute exact zeros.
    C = -(-C0 // _lane_mult(features.dtype)) * _lane_mult(features.dtype)
    K = _round_cout(K0) if kv <= 128 else K0
    if K and (C != C0 or K != K0):
        features = _pad_last(features, C)
        filters = _pad_first(_pad_last(filters, C), K)
        if bias is not None:
            bias = _pad_last(bias, K)
    else:
        C, K = C0, K0
    features = features.contiguous()
    filters = filters.contiguous()
    out = torch.empty((n_out, K), dtype=features.dtype, device=features.device)
    if bias is not None:
        bias = bias.to(features.dtype).contiguous()
    ws = _ws(L.spx_igemm_acc_bytes(n_out, K, kv), features.device) if kv > 32 else None   # fp32 partial sums
    sink = getattr(_stats_req, "sink", None)
    if (sink is not None and sink.records is None and bias is None and int(act_type) == Activation.None_
            and K == K0 and kv <= 32 and n_out > 0):
        slots = int(L.spx_igemm_fwd_stats_slots(n_out))
        # [field][channel][workgroup] over the `used` workgroups of the launch (igemm_defs.h bn_record_store): a flat buffer
        records = torch.empty((3 * K * slots,), dtype=torch.float32, device=features.device)
        used = ctypes.c_int(0)
        _lib.check(L.spx_igemm_fwd_stats(features.data_ptr(), filters.data_ptr(), out.data_ptr(),
                                         _ptr(pair), _ptr(mask), _ptr(argsort), _with_dense_hint(int(tile_order), argsort, mask),
                                         features.shape[0], n_out, C, K, kv, _dtype_code(features), identity_k, None,
                                         int(act_type) | (_OUT_CACHED if getattr(_out_policy, "cached", False) else 0),
                                         float(act_alpha), None, 0, records.data_ptr(), slots, _ptr(sink.n_live),
                                         ctypes.byref(used), _stream(features)))
        if used.value > 0:
            sink.records, sink.count, sink.rows, sink.channels = records, int(used.value), n_out, K
        return out
    _lib.check(L.spx_igemm_fwd(features.data_ptr(), filters.data_ptr(), out.data_ptr(),
                               _ptr(pair), _ptr(mask), _ptr(argsort), _with_dense_hint(int(tile_order), argsort, mask),
                               features.shape[0], n_out, C, K, kv, _dtype_code(features), identity_k, _ptr(bias),
                               int(act_type) | (_OUT_CACHED if getattr(_out_policy, "cached", False) else 0),
                               float(act_alpha), _ptr(ws), 0 if ws is None else ws.numel(),
                               _stream(features)))
    return out if K == K0 else out[:, :K0].contiguous()


@_on_device
def igemm_dgrad(out_bp: torch.Tensor, filters: torch.Tensor, pair: torch.Tensor,
                mask: Optional[torch.Tensor], argsort: Optional[torch.Tensor], n_in: int,
                subm: bool, tile_order: int = 0) -> torch.Tensor:
    """din[i] = sum_k dout[pair[k][i]] @ W[:, k, :] (SubM: pass the forward table, subm=True)."""
    _check_feat(out_bp, filters)
    L = _lib.load()
    K0, C0 = filters.shape[0], filters.shape[-1]
    kv = filters.numel() // (K0 * C0)
    # same padding rule as igemm_fwd: here K is the reduction length and C the output width
    K = -(-K0 // _lane_mult(out_bp.dtype)) * _lane_mult(out_bp.dtype)
    C = _round_cout(C0) if kv <= 128 else C0
    if C and (C != C0 or K != K0):
        out_bp = _pad_last(out_bp, K)
        filters = _pad_first(_pad_last(filters, C), K)
    else:
        C, K = C0, K0
    out_bp = out_bp.contiguous()
    filters = filters.contiguous()
    din = torch.empty((n_in, C), dtype=out_bp.dtype, device=out_bp.device)
    code = _dtype_code(out_bp)
    ws = _ws(max(L.spx_igemm_dgrad_ws_bytes(C, K, kv, code), L.spx_igemm_acc_bytes(n_in, C, kv)), out_bp.device)
    _lib.check(L.spx_igemm_dgrad(out_bp.data_ptr(), filters.data_ptr(), din.data_ptr(), _ptr(pair),
                                 _ptr(mask), _ptr(argsort), _with_dense_hint(int(tile_order), argsort, mask), out_bp.shape[0],
                                 n_in, C, K, kv, code,
                                 int(subm), ws.data_ptr(), ws.numel(), _stream(out_bp)))
    return din if C == C0 else din[:, :C0].contiguous()


@_on_device
def wgrad_plan(num_per_loc: torch.Tensor, n_in: int, kv: int, subm: bool) -> torch.Tensor:
    """Work plan of wgrad for one rulebook (built once, reused by every backward)."""
    L = _lib.load()
    plan = torch.empty((L.spx_wgrad_plan_bytes(n_in, kv) // 4,), dtype=torch.int32,
                       device=num_per_loc.device)
    _lib.check(L.spx_wgrad_plan(num_per_loc.data_ptr(), n_in, kv, int(subm), plan.data_ptr(),
                                _stream(num_per_loc)))
    return plan


_WGRAD_MAX_KV = 128     # offsets one spx_igemm_wgrad call / one plan covers


def _wgrad_in_groups(features, out_bp, filters_shape, native, num_per_loc, subm: bool, kv: int):
    """Kernel volumes beyond 128 (6x6x6, 7x7x7: the reference's Native path trains any volume,
    ops.py:962-1015): the offsets go through spx_igemm_wgrad 128 at a time, each group with its own
    slice of the lists and explicit per-list counts (SubM: the mirror rule of ops.py:962-968 resolved
    here, on the device)."""
    K0, C0 = filters_shape[0], filters_shape[-1]
    n_in = native.shape[2]
    counts = num_per_loc.to(torch.int32).clone()
    if subm:
        centre = kv // 2
        counts[centre] = n_in
        counts[centre + 1:] = num_per_loc[:kv - centre - 1].flip(0)
    dw = torch.empty((K0, kv, C0), dtype=features.dtype, device=features.device)
    for kbase in range(0, kv, _WGRAD_MAX_KV):
        g = min(_WGRAD_MAX_KV, kv - kbase)
        part = igemm_wgrad(features, out_bp, (K0, g, C0), native[:, kbase:kbase + g].contiguous(),
                           counts[kbase:kbase + g].contiguous(), False, None)
        dw[:, kbase:kbase + g] = part
    return dw.view(tuple(filters_shape))


@_on_device
def igemm_wgrad(features: torch.Tensor, out_bp: torch.Tensor, filters_shape, native: torch.Tensor,
                num_per_loc: torch.Tensor, subm: bool,
                plan: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW[:, k, :] = sum_j dout[native[1][k][j]].T (x) feat[native[0][k][j]]."""
    _require_gpu(features, "features")
    L = _lib.load()
    K0, C0 = filters_shape[0], filters_shape[-1]
    kv = int(np.prod(filters_shape)) // (K0 * C0)
    if kv > _WGRAD_MAX_KV:
        return _wgrad_in_groups(features, out_bp, filters_shape, native, num_per_loc, subm, kv)
    m = _lane_mult(features.dtype)
    C, K = -(-C0 // m) * m, -(-K0 // m) * m          # the MFMA wgrad needs whole lane pieces
    features = _pad_last(features, C).contiguous()
    out_bp = _pad_last(out_bp, K).contiguous()
    n_in = native.shape[2]
    shape = (K,) + tuple(filters_shape[1:-1]) + (C,)
    dw = torch.empty(shape, dtype=features.dtype, device=features.device)
    ws = _ws(L.spx_igemm_wgrad_ws_bytes(n_in, C, K, kv), features.device)
    _lib.check(L.spx_igemm_wgrad(features.data_ptr(), out_bp.data_ptr(), dw.data_ptr(),
                                 native.data_ptr(), num_per_loc.data_ptr(), _ptr(plan), n_in,
                                 out_bp.shape[0], C, K, kv, _dtype_code(features), int(subm),
                                 ws.data_ptr(), ws.numel(), _stream(features)))
    return dw if (C == C0 and K == K0) else dw[:K0, ..., :C0].contiguous()


# The weight gradient's second stage, deferred to the END of a backward pass (include/spconv_amd.h: spx_igemm_bwd_deferred
# / spx_wgrad_stage2_batch).  Nothing inside a pass waits for a layer's dW when the weight is a leaf whose .grad is empty
# (AccumulateGrad then just keeps the tensor it is handed): the reductions of the partial tiles -- a dependent 4-9 us
# launch per layer -- leave the chain and run as ONE launch per sixteen layers from the autograd engine's final callback.
# Only inside `deferred_wgrad()` -- the static training runner opens it around its backward call; eager passes never
# defer -- and not for weights that are not leaves (a cast in front: its backward READS dW), that already hold a gradient
# (accumulation reads it), that carry hooks, or whose shapes take a path without a second stage.  A weight that shows up a
# second time in a pass flushes what is pending first (the engine adds the two gradients).
# SPCONV_AMD_WGRAD_DEFER=0 turns it off.
_WGRAD_DEFER = os.environ.get("SPCONV_AMD_WGRAD_DEFER", "1") != "0"
_STAGE2_JOB_BYTES = 64     # SPX_STAGE2_JOB_BYTES
_defer_depth = [0]
_defer_passes = {}         # autograd graph-task id -> {"jobs": [...], "seen": {weight address}}
_defer_lock = threading.Lock()


class deferred_wgrad:
    """Context of a backward pass whose caller reads no weight gradient before the pass has ended."""

    def __enter__(self):
        _defer_depth[0] += 1
        return self

    def __exit__(self, *exc):
        _defer_depth[0] -= 1
        if _defer_depth[0] == 0 and _defer_passes:
            # (a pass that ended in an exception never reached its final callback: what it left pending is dropped)
            with _defer_lock:
                _defer_passes.clear()
        return False


def _flush_deferred(task_id: int) -> None:
    with _defer_lock:
        st = _defer_passes.pop(task_id, None)
    if not st or not st["jobs"]:
        return
    L = _lib.load()
    by_dev = {}
    for job, param, _keep in st["jobs"]:
        g = param.grad
        if g is not None:                   # (the engine kept the tensor it was handed, or made one of its own)
            if not (g.is_contiguous() and g.shape == param.shape and g.dtype == param.dtype):
                raise RuntimeError("deferred weight gradient: .grad of a weight has another layout than the weight")
            L.spx_stage2_job_retarget(job, g.data_ptr())
        by_dev.setdefault(param.device, []).append(job)
    for dev, jobs in by_dev.items():
        blob = b"".join(j.raw for j in jobs)
        with torch.cuda.device(dev):
            _lib.check(L.spx_wgrad_stage2_batch(blob, len(jobs), torch._C._cuda_getCurrentRawStream(dev.index)))


def _defer_state(filters: torch.Tensor):
    """The pending list of the backward pass this call belongs to, or None (see above)."""
    if not (_WGRAD_DEFER and _defer_depth[0] > 0 and filters.is_cuda):
        return None
    if not (filters.is_leaf and filters.requires_grad and filters.grad is None and filters.is_contiguous()):
        return None
    if filters._backward_hooks or getattr(filters, "_post_accumulate_grad_hooks", None):
        return None
    task = torch._C._current_graph_task_id()
    if task < 0:
        return None
    with _defer_lock:
        st = _defer_passes.get(task)
        if st is None:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(lambda t=task: _flush_deferred(t))
            except RuntimeError:            # not inside a backward pass
                return None
            st = _defer_passes[task] = {"jobs": [], "seen": set()}
        key = filters.data_ptr()
        again = key in st["seen"]
        st["seen"].add(key)
    if again:
        # a second use of this weight in the pass: every pending gradient is completed before the engine adds the two
        _flush_deferred(task)
        return None
    return st


@_on_device
def igemm_bwd(features: torch.Tensor, out_bp: torch.Tensor, filters: torch.Tensor,
              table: torch.Tensor, mask: Optional[torch.Tensor], argsort: Optional[torch.Tensor],
              native: torch.Tensor, num_per_loc: torch.Tensor, subm: bool,
              plan: Optional[torch.Tensor] = None, need_din: bool = True,
              tile_order: int = 0, dense_rows: bool = False, lists=None):
    """(din, dW) of one layer from one launch (+ the wgrad second stage).  need_din=False (the
    input does not require grad: a network's first layer) computes dW only and returns None.
    `table` / `mask` are row-order tables unless tile_order == 1 (see tables_of).  `native` may be None
    when `lists` is given: a callable returning (native, num_per_loc, plan), asked only if the pair-list
    weight gradient runs (the rows walk of narrow layers needs none of the three)."""
    _check_feat(out_bp, filters)
    K0, C0 = filters.shape[0], filters.shape[-1]
    m = _lane_mult(out_bp.dtype)
    kvf = filters.numel() // (K0 * C0)
    if ((_BWD_ROWS is True or (_BWD_ROWS == "auto" and dense_rows))
            and out_bp.dtype in (torch.float16, torch.bfloat16) and K0 in (16, 32) and C0 in (16, 32)
            and kvf <= 27 and table is not None and mask is not None and mask.shape[1] == 1
            and (argsort is None or tile_order == _ROWS_LAYOUT) and tile_order != 1):
        # (the rows walk reads the row-order tables: a rows layout does not concern it)
        return _igemm_bwd_rows(features, out_bp, filters, table, mask, subm, need_din, K0, C0, kvf)
    if native is None:
        native, num_per_loc, plan = lists()
    if not need_din or K0 % m or C0 not in _MFMA_COUT or kvf > 32:      # (kv > 32: dgrad in groups of 32 offsets)
        din = igemm_dgrad(out_bp, filters, table, mask, argsort, features.shape[0], subm,
                          tile_order=tile_order) if need_din else None
        return din, igemm_wgrad(features, out_bp, filters.shape, native, num_per_loc, subm, plan)
    L = _lib.load()
    features = features.contiguous()
    out_bp = out_bp.contiguous()
    st = _defer_state(filters)
    filters_leaf = filters
    filters = filters.contiguous()
    K, C = filters.shape[0], filters.shape[-1]
    kv = filters.numel() // (K * C)
    n_in = features.shape[0]
    din = torch.empty((n_in, C), dtype=out_bp.dtype, device=out_bp.device)
    dw = torch.empty_like(filters)
    ws = _ws(L.spx_igemm_wgrad_ws_bytes(native.shape[2], C, K, kv), features.device)
    if st is not None:
        job = ctypes.create_string_buffer(_STAGE2_JOB_BYTES)
        _lib.check(L.spx_igemm_bwd_deferred(features.data_ptr(), out_bp.data_ptr(), filters.data_ptr(),
                                            din.data_ptr(), dw.data_ptr(), _ptr(table), _ptr(mask), _ptr(argsort),
                                            int(tile_order), native.data_ptr(), num_per_loc.data_ptr(), _ptr(plan), n_in,
                                            out_bp.shape[0], C, K, kv, _dtype_code(out_bp), int(subm),
                                            ws.data_ptr(), ws.numel(), _stream(out_bp), job))
        with _defer_lock:       # (ws, plan and the lists live until the batch launch; dw is the engine's from here on)
            st["jobs"].append((job, filters_leaf, (ws, plan, native, num_per_loc)))
        return din, dw
    _lib.check(L.spx_igemm_bwd(features.data_ptr(), out_bp.data_ptr(), filters.data_ptr(),
                               din.data_ptr(), dw.data_ptr(), _ptr(table), _ptr(mask), _ptr(argsort),
                               int(tile_order), native.data_ptr(), num_per_loc.data_ptr(), _ptr(plan), n_in,
                               out_bp.shape[0], C, K, kv, _dtype_code(out_bp), int(subm),
                               ws.data_ptr(), ws.numel(), _stream(out_bp)))
    return din, dw


# Narrow layers (16 / 32 channels): the backward from one gather per pair (csrc/igemm_bwdn.hip).  Its tile
# walk visits every table row of a 128-row tile, so it pays where neighbourhoods are DENSE (level 2 of a LiDAR
# backbone, 15 pairs per voxel: 185 -> 114 us) and loses where they are not (level 1, 5 pairs per voxel: 98 ->
# 108 us; a uniform scene far more).  The pairs-per-voxel count lives on the device; what the host knows without
# a read-back is the OCCUPANCY of the grid the table's rows live in (voxels / cells), which grows with it:
# 0.1 % at level 1, 0.66 % at level 2, 1.8 % at level 3 of the config-4 network.
# SPCONV_AMD_BWD_ROWS: "auto" (default: occupancy >= _BWD_ROWS_OCC), "1" always, "0" never.
_BWD_ROWS = {"0": False, "1": True}.get(os.environ.get("SPCONV_AMD_BWD_ROWS", "auto"), "auto")
_BWD_ROWS_OCC = float(os.environ.get("SPCONV_AMD_BWD_ROWS_OCC", "0.003"))


def rows_backward_expected(dtype: torch.dtype, cin: int, cout: int, kv: int, n_rows: int, batch_size: int,
                           dims) -> bool:
    """Will the backward of a layer with these shapes take the rows walk (igemm_bwdn.hip), which reads the dense
    table and needs neither the Native lists nor the range plan?  The same rule as igemm_bwd / _dense_rows, usable
    BEFORE the rulebook exists: a module asks it to decide whether its build has to produce the lists at all (a
    layer that shares the rulebook and does need them derives them from the table on first use)."""
    if _BWD_ROWS is False or dtype not in (torch.float16, torch.bfloat16):
        return False
    if cin not in (16, 32) or cout not in (16, 32) or kv > 27:
        return False
    if _BWD_ROWS is True:
        return True
    cells = float(batch_size)
    for d in dims:
        cells *= float(d)
    return n_rows >= _BWD_ROWS_OCC * cells


def _dense_rows(rb: Optional[Rulebook], n_rows: int, which: str) -> bool:
    """Occupancy of the grid that the rows of table `which` ("fwd": output rows, "bwd": input rows) live in."""
    if rb is None:
        return False
    dims = rb.out_shape if which == "fwd" else rb.in_shape
    if not dims:
        return False
    cells = float(rb.batch_size)
    for d in dims:
        cells *= float(d)
    return n_rows >= _BWD_ROWS_OCC * cells


@_on_device
def _igemm_bwd_rows(features, out_bp, filters, table, mask, subm, need_din, K, C, kv):
    L = _lib.load()
    features, out_bp = features.contiguous(), out_bp.contiguous()
    n_in = features.shape[0]
    # the weights with the dout channel contiguous ([kv, C, K]: one small copy; SubM's mirrored slice order is the
    # kernel's business)
    wt = filters.reshape(K, kv, C).permute(1, 2, 0).contiguous()
    din = torch.empty((n_in, C), dtype=out_bp.dtype, device=out_bp.device) if need_din else None
    dw = torch.empty_like(filters)
    ws = _ws(L.spx_igemm_bwd_rows_ws_bytes(n_in, C, K, kv), features.device)
    _lib.check(L.spx_igemm_bwd_rows(features.data_ptr(), out_bp.data_ptr(), wt.data_ptr(), _ptr(din), dw.data_ptr(),
                                    table.data_ptr(), mask.data_ptr(), n_in, out_bp.shape[0], C, K, kv, int(subm),
                                    _dtype_code(out_bp), ws.data_ptr(), ws.numel(), _stream(out_bp)))
    return din, dw


def _plan_of(rb: Optional[Rulebook]) -> Optional[torch.Tensor]:
    if rb is None or rb.pair_native is None or rb.kv > _WGRAD_MAX_KV:
        return None
    if rb.wgrad_plan is None:
        rb.wgrad_plan = wgrad_plan(rb.num_per_loc, rb.n_in, rb.kv, rb.subm)
    return rb.wgrad_plan


@_on_device
def bias_act_inplace(out: torch.Tensor, bias: Optional[torch.Tensor], act_type: int,
                     act_alpha: float = 0.0) -> torch.Tensor:
    """InferenceOps.bias_add_act_inplace & friends (csrc/sparse/inference.py:26-146)."""
    L = _lib.load()
    if bias is not None:
        bias = bias.to(out.dtype).contiguous()
    _lib.check(L.spx_bias_act_inplace(out.data_ptr(), _ptr(bias), out.shape[0], out.shape[1],
                                      _dtype_code(out), int(act_type), float(act_alpha),
                                      _stream(out)))
    return out


def record_voxel_count_(buf: torch.Tensor, rb: Rulebook, rows: int) -> None:
    """max_num_voxels_during_training (reference conv.py:44,131-138): the running maximum of a strided layer's
    output count.  A static-shape build has `rows` = the frozen BOUND, not the count -- recording that would
    inflate every bound later derived from the buffer (freeze_bounds: recorded * margin) -- so there the count
    the build found on the device is folded in, without a read-back."""
    if rb.n_out_dev is not None:
        buf.copy_(torch.maximum(buf, rb.n_out_dev[:1].to(buf.dtype).reshape(buf.shape)))
    else:
        buf.clamp_(min=int(rows))


def maximum_value_int_(ten: torch.Tensor, value: int) -> torch.Tensor:
    """In-place ``ten = max(ten, value)`` for integer tensors (reference ops.py:2099-2105; used by the
    voxeliser to clamp counts)."""
    return ten.clamp_(min=int(value))


def fused_indice_conv(features, filters, bias, indice_pairs, indice_pair_num, num_activate_out,
                      inverse, subm):
    """Present in the reference's namespace and unimplemented there as well (ops.py:1098-1100)."""
    raise NotImplementedError


# ----------------------------------------- layout conversion for bare tensors
def _table_from_native(indice_pairs: torch.Tensor, indice_pair_num: torch.Tensor, n_dst: int,
                       subm: bool, inverse: bool):
    L = _lib.load()
    kv, n_in = indice_pairs.shape[1], indice_pairs.shape[2]
    table = torch.empty((kv, n_dst), dtype=torch.int32, device=indice_pairs.device)
    mask = torch.empty((n_dst, (kv + 31) // 32), dtype=torch.int32, device=indice_pairs.device)
    _lib.check(L.spx_native_to_table(indice_pairs.data_ptr(), indice_pair_num.data_ptr(), n_in,
                                     n_dst, kv, int(subm), int(inverse), table.data_ptr(),
                                     mask.data_ptr(), _stream(indice_pairs)))
    return table, mask


def _native_from_table(table: torch.Tensor, subm: bool):
    L = _lib.load()
    kv, n = table.shape
    native = torch.empty((2, kv, n), dtype=torch.int32, device=table.device)
    num = torch.empty((kv,), dtype=torch.int32, device=table.device)
    ws = _ws(L.spx_table_to_native_ws_bytes(n, kv), table.device)
    _lib.check(L.spx_table_to_native(table.data_ptr(), int(subm), kv, n, native.data_ptr(),
                                     num.data_ptr(), ws.data_ptr(), ws.numel(), _stream(table)))
    return native, num


# ------------------------------------------- reference-shaped public op API
def indice_conv(features: torch.Tensor, filters: torch.Tensor, indice_pairs: torch.Tensor,
                indice_pair_num: torch.Tensor, num_activate_out: int, inverse: bool = False,
                subm: bool = False, algo: ConvAlgo = ConvAlgo.Native, timer=None,
                bias: Optional[torch.Tensor] = None, act_alpha: float = 0.0,
                act_beta: float = 0.0, act_type: int = Activation.None_) -> torch.Tensor:
    """ConvAlgo.Native forward (ops.py:811-1095) on the Native lists [2, kv, N_in]."""
    _check_feat(features, filters)
    rb: Optional[Rulebook] = rulebook_of(indice_pairs)
    kv = indice_pairs.shape[1]
    argsort, tile_order = None, 0
    if rb is not None and not inverse:
        table, mask, argsort, tile_order = tables_of(rb, "fwd", filters.shape[0])
    elif rb is not None and inverse and rb.pair_bwd is not None:
        table, mask, argsort, tile_order = tables_of(rb, "bwd", filters.shape[0])
    else:
        table, mask = _table_from_native(indice_pairs, indice_pair_num, num_activate_out, subm, inverse)
    return igemm_fwd(features, filters, table, mask, argsort, num_activate_out,
                     kv // 2 if subm else -1, bias, act_type, act_alpha, tile_order=tile_order)


def indice_conv_backward(features: torch.Tensor, filters: torch.Tensor, out_bp: torch.Tensor,
                         indice_pairs: torch.Tensor, indice_pair_num: torch.Tensor,
                         inverse: bool = False, subm: bool = False,
                         algo: ConvAlgo = ConvAlgo.Native, timer=None, need_din: bool = True):
    """ConvAlgo.Native backward (ops.py:1103-1447): returns (din, dfilters)."""
    _check_feat(features, filters)
    rb: Optional[Rulebook] = rulebook_of(indice_pairs)
    n_in = features.shape[0]
    argsort, tile_order, which = None, 0, None
    if rb is not None:
        # dgrad gathers dout rows for every input row: SubM reads the forward table (mirrored
        # weights), a regular conv the table indexed by its input rows, an inverse conv the forward one
        which = "fwd" if (subm or inverse) else "bwd"
        table, mask, argsort, tile_order = tables_of(rb, which, filters.shape[-1])
    else:
        table, mask = _table_from_native(indice_pairs, indice_pair_num, n_in, subm, (not subm) and (not inverse))
    if rb is not None and not inverse and not rb.has_native:
        # the build left the lists out (conv.py: _needs_native_lists): they -- and the range plan -- are derived
        # only if the launch below turns out to need them
        return igemm_bwd(features, out_bp, filters, table, mask, argsort, None, None, subm, None, need_din,
                         tile_order=tile_order, dense_rows=_dense_rows(rb, n_in, which),
                         lists=lambda: (rb.pair_native, rb.num_per_loc, _plan_of(rb)))
    native = indice_pairs if (rb is None or indice_pairs.shape[2] > 0 or rb.n_in == 0) else rb.pair_native
    if rb is not None and indice_pair_num.numel() != rb.kv:
        indice_pair_num = rb.num_per_loc
    if inverse:
        native = rb.native_swapped() if rb is not None else torch.stack(
            [indice_pairs[1], indice_pairs[0]]).contiguous()
    plan = _plan_of(rb)
    if native.shape[2] == n_in or not need_din:
        return igemm_bwd(features, out_bp, filters, table, mask, argsort, native, indice_pair_num,
                         subm, plan, need_din, tile_order=tile_order,
                         dense_rows=_dense_rows(rb, n_in, which) if rb is not None else False)
    return (igemm_dgrad(out_bp, filters, table, mask, argsort, n_in, subm, tile_order=tile_order),
            igemm_wgrad(features, out_bp, filters.shape, native, indice_pair_num, subm, plan))


def implicit_gemm(features: torch.Tensor, filters: torch.Tensor, pair_fwd: torch.Tensor,
                  pair_mask_fwd_splits: List[torch.Tensor],
                  mask_argsort_fwd_splits: List[torch.Tensor], num_activate_out: int,
                  masks: List[np.ndarray], is_train: bool, is_subm: bool, timer=None,
                  fp32_accum: Optional[bool] = None, bias: Optional[torch.Tensor] = None,
                  act_alpha: float = 0.0, act_beta: float = 0.0,
                  act_type: int = Activation.None_, output_scale: float = 1.0,
                  scale: Optional[torch.Tensor] = None, output_add: Optional[torch.Tensor] = None,
                  output_add_scale: float = 0.0, output_dtype: Optional[torch.dtype] = None):
    """Masked implicit GEMM forward (ops.py:1450-1664).  Returns (out, mask_out, mask_width);
    the last two exist for signature parity (this wgrad does not consume tile masks)."""
    mask = pair_mask_fwd_splits[0] if pair_mask_fwd_splits else None
    rb: Optional[Rulebook] = rulebook_of(pair_fwd)
    argsort, tile_order = (rb.argsort_fwd if rb is not None else None), 0
    kv = pair_fwd.shape[0]
    if rb is not None and pair_fwd is rb.pair_fwd:
        pair_fwd, mask, argsort, tile_order = tables_of(rb, "fwd", filters.shape[0])
    if features.dtype in (torch.int8, torch.qint8):
        # int8 inference (ops.py:1540-1553,1631-1662): scale = per-channel multiplier, bias is
        # fp32 in output-quantised units, the residual input is scaled by add_scale / out_scale
        assert not is_train, "int8 is inference only"
        out_dt = torch.int8 if output_dtype in (None, torch.int8, torch.qint8) else output_dtype
        beta = output_add_scale / output_scale if output_add is not None else 0.0
        out = igemm_fwd_int8(features, filters, pair_fwd, mask, argsort, num_activate_out,
                             kv // 2 if is_subm else -1, scale, bias, output_add, beta, out_dt,
                             act_type, act_alpha, tile_order=tile_order,
                             sparse_hint=rb is not None and rb.sparse_class is True,
                             hint_rows=getattr(rb, "heavy_rows", 0) if rb is not None else 0)
        if out_dt == torch.int8 and features.is_quantized:
            out = torch._make_per_tensor_quantized_tensor(out, float(output_scale), 0)
        return out, None, -1
    if scale is not None or output_add is not None:
        raise NotImplementedError("scale / output_add belong to the int8 path")
    out = igemm_fwd(features, filters, pair_fwd, mask, argsort, num_activate_out,
                    kv // 2 if is_subm else -1, bias, act_type, act_alpha, tile_order=tile_order)
    return out, None, -1


def implicit_gemm_backward(features: torch.Tensor, filters: torch.Tensor, out_bp: torch.Tensor,
                           pair_fwd: torch.Tensor, pair_bwd: torch.Tensor,
                           pair_mask_fwd_splits: List[torch.Tensor],
                           pair_mask_bwd_splits: List[torch.Tensor],
                           mask_argsort_fwd_splits: List[torch.Tensor],
                           mask_argsort_bwd_splits: List[torch.Tensor],
                           mask_output_fwd: Optional[torch.Tensor], masks: List[np.ndarray],
                           mask_width: int, is_subm: bool, timer=None,
                           fp32_accum: Optional[bool] = None, need_din: bool = True):
    """Masked implicit GEMM backward (ops.py:1667-1896): returns (din, dfilters)."""
    rb: Optional[Rulebook] = rulebook_of(pair_fwd)
    n_in = features.shape[0]
    if rb is not None and rb.pair_native is not None:
        native, num = rb.pair_native, rb.num_per_loc
    else:
        native, num = _native_from_table(pair_fwd if is_subm else pair_bwd, is_subm)
    plan = _plan_of(rb)
    tile_order = 0
    if is_subm:
        table, mask = pair_fwd, pair_mask_fwd_splits[0]
        argsort = rb.argsort_fwd if rb is not None else None
        if rb is not None and pair_fwd is rb.pair_fwd:
            table, mask, argsort, tile_order = tables_of(rb, "fwd", filters.shape[-1])
    else:
        table, mask = pair_bwd, pair_mask_bwd_splits[0]
        argsort = rb.argsort_bwd if rb is not None else None
        if rb is not None and pair_bwd is rb.pair_bwd:
            table, mask, argsort, tile_order = tables_of(rb, "bwd", filters.shape[-1])
    if native.shape[2] == n_in or not need_din:
        return igemm_bwd(features, out_bp, filters, table, mask, argsort, native, num, is_subm, plan,
                         need_din, tile_order=tile_order,
                         dense_rows=_dense_rows(rb, n_in, "fwd" if is_subm else "bwd"))
    return (igemm_dgrad(out_bp, filters, table, mask, argsort, n_in, is_subm, tile_order=tile_order),
            igemm_wgrad(features, out_bp, filters.shape, native, num, is_subm, plan))


# ------------------------------------------------------------------ pooling
_POOL_CODES = dict(_DTYPES)
_POOL_CODES[torch.int8] = _lib.DTYPE_I8


def _pool_code(t: torch.Tensor) -> int:
    try:
        return _POOL_CODES[t.dtype]
    except KeyError:
        raise NotImplementedError(f"unsupported pooling dtype {t.dtype}") from None


def _mask_of(pair: torch.Tensor, fwd: bool) -> Optional[torch.Tensor]:
    rb = rulebook_of(pair)
    if rb is None:
        return None
    if fwd:
        return rb.mask_fwd
    # a SubM rulebook keeps ONE mask (bits of pair_fwd); its pair_bwd is the mirrored table, whose
    # bit positions differ -> walk every offset instead
    return None if rb.subm else rb.mask_bwd


@_on_device
def indice_maxpool_implicit_gemm(features: torch.Tensor, indice_pairs: torch.Tensor,
                                 num_activate_out: int, init_zero: bool = False) -> torch.Tensor:
    """out[o] = max over the valid pairs of column o of pair_fwd [kv, n_out] (ops.py:1976-1997)."""
    _require_gpu(features, "features")
    L = _lib.load()
    q = features.is_quantized
    feats = _int_repr(features).contiguous()
    out = torch.empty((num_activate_out, feats.shape[1]), dtype=feats.dtype, device=feats.device)
    _lib.check(L.spx_maxpool_fwd(feats.data_ptr(), out.data_ptr(), indice_pairs.data_ptr(),
                                 _ptr(_mask_of(indice_pairs, True)), num_activate_out, feats.shape[1],
                                 indice_pairs.shape[0], _pool_code(feats), int(init_zero), _stream(feats)))
    if q:
        out = torch._make_per_tensor_quantized_tensor(out, features.q_scale(), features.q_zero_point())
    return out


@_on_device
def indice_maxpool_implicit_gemm_backward(features: torch.Tensor, out_features: torch.Tensor,
                                          out_bp: torch.Tensor, indice_pairs: torch.Tensor) -> torch.Tensor:
    """din from pair_bwd [kv, n_in] (ops.py:2000-2023)."""
    L = _lib.load()
    features, out_features, out_bp = features.contiguous(), out_features.contiguous(), out_bp.contiguous()
    din = torch.empty_like(features)
    _lib.check(L.spx_maxpool_bwd(features.data_ptr(), out_features.data_ptr(), out_bp.data_ptr(),
                                 din.data_ptr(), indice_pairs.data_ptr(),
                                 _ptr(_mask_of(indice_pairs, False)), features.shape[0],
                                 features.shape[1], indice_pairs.shape[0], _pool_code(features),
                                 _stream(features)))
    return din


@_on_device
def indice_avgpool_implicit_gemm(features: torch.Tensor, indice_pairs: torch.Tensor,
                                 num_activate_out: int, calc_count: bool):
    """(mean over the valid pairs, count [n_out] or empty) -- ops.py:2026-2056."""
    _require_gpu(features, "features")
    L = _lib.load()
    features = features.contiguous()
    out = torch.empty((num_activate_out, features.shape[1]), dtype=features.dtype, device=features.device)
    count = (torch.empty((num_activate_out,), dtype=torch.int32, device=features.device)
             if calc_count else None)
    _lib.check(L.spx_avgpool_fwd(features.data_ptr(), out.data_ptr(), _ptr(count), indice_pairs.data_ptr(),
                                 _ptr(_mask_of(indice_pairs, True)), num_activate_out, features.shape[1],
                                 indice_pairs.shape[0], _pool_code(features), _stream(features)))
    return out, (count if count is not None else torch.Tensor())


@_on_device
def indice_avgpool_implicit_gemm_backward(out_bp: torch.Tensor, indice_pairs: torch.Tensor,
                                          count_out: torch.Tensor) -> torch.Tensor:
    """din[i] = sum_o dout[o] / count[o] over pair_bwd [kv, n_in] (cf. ops.py:2059-2084).  With
    SPCONV_AMD_REFERENCE_QUIRKS=1 the reference's own arithmetic (it MULTIPLIES by the count, maxpool.py:262-300): the
    Python constant is the single source of truth and is handed to the library with every call (round-4 ADVICE: the
    native side used to re-read the environment by itself)."""
    from spconv_amd import constants
    L = _lib.load()
    _lib.check(L.spx_set_option(b"SPCONV_AMD_REFERENCE_QUIRKS", int(bool(constants.REFERENCE_QUIRKS))))
    out_bp = out_bp.contiguous()
    n_in = indice_pairs.shape[1]
    din = torch.empty((n_in, out_bp.shape[1]), dtype=out_bp.dtype, device=out_bp.device)
    _lib.check(L.spx_avgpool_bwd(out_bp.data_ptr(), din.data_ptr(), count_out.data_ptr(),
                                 indice_pairs.data_ptr(), _ptr(_mask_of(indice_pairs, False)), n_in,
                                 out_bp.shape[1], indice_pairs.shape[0], _pool_code(out_bp),
                                 _stream(out_bp)))
    return din


def indice_maxpool(features: torch.Tensor, indice_pairs: torch.Tensor, indice_pair_num: torch.Tensor,
                   num_activate_out: int) -> torch.Tensor:
    """ConvAlgo.Native max pool on the lists [2, kv, N_in] (ops.py:1899-1935).  Its output starts
    as zeros in the reference, so the result is max(0, max over the pairs); kept."""
    rb = rulebook_of(indice_pairs)
    table = rb.pair_fwd if rb is not None else _table_from_native(
        indice_pairs, indice_pair_num, num_activate_out, False, False)[0]
    if rb is not None:
        attach_rulebook(table, rb)
    return indice_maxpool_implicit_gemm(features, table, num_activate_out, init_zero=True)


def indice_maxpool_backward(features, out_features, out_bp, indice_pairs, indice_pair_num):
    """ConvAlgo.Native max pool backward (ops.py:1938-1973)."""
    rb = rulebook_of(indice_pairs)
    n_in = features.shape[0]
    table = rb.pair_bwd if rb is not None and rb.pair_bwd is not None else _table_from_native(
        indice_pairs, indice_pair_num, n_in, False, True)[0]
    if rb is not None:
        attach_rulebook(table, rb)
    return indice_maxpool_implicit_gemm_backward(features, out_features, out_bp, table)


def global_pool_rearrange(indices: torch.Tensor, batch_size: int):
    """Row ids of every batch item, padded to [batch, N], and the counts (ops.py:2087-2100).  One
    stable sort by batch id + a scatter: no per-scene loop, no read-back."""
    n = indices.shape[0]
    dev = indices.device
    out = torch.zeros((batch_size, n), dtype=torch.int32, device=dev)
    if n == 0:
        return out, torch.zeros((batch_size,), dtype=torch.int32, device=dev)
    b = indices[:, 0].long()
    key = torch.where((b >= 0) & (b < batch_size), b, torch.full_like(b, batch_size))
    order = torch.argsort(key, stable=True)
    skey = key[order]
    counts = torch.bincount(key, minlength=batch_size + 1)
    starts = torch.cumsum(counts, 0) - counts
    col = torch.arange(n, device=dev) - starts[skey]
    keep = skey < batch_size
    out[skey[keep], col[keep]] = order[keep].int()
    return out, counts[:batch_size].int()
