"""Static shapes: one hipGraph for a whole sparse backbone -- inference or a training step --, rulebook builds included.

The reference's deployment path sizes every buffer before the first layer runs and bounds the number
of outputs of each strided layer (num_out_act_bound: spconv/pytorch/ops.py:263-266,644-645; the
pre-sized workspace of csrc/sparse/all.py:2030-2185; the per-layer maxima a trained model carries
in `max_num_voxels_during_training`, pytorch/conv.py:44,131-138).  With those bounds every tensor of
a forward pass has a static shape, so on this backend the WHOLE pass -- hash-table builds, pair
tables, gather-GEMMs -- can be recorded once and replayed per scene:

  * the input is padded to `max_voxels` rows; padding rows carry batch index -1 (dead rows: the
    rulebook kernels neither hash them nor pair them, csrc/rulebook.hip subm_insert_kernel /
    conv_stage1_kernel)
  * a strided layer builds its rulebook with spx_conv_rulebook_static (include/spconv_amd.h): room
    for `static_num_out` outputs, nothing read back; the rows past the real count come out dead
  * SubM layers need nothing special: n_out == n_in
  * the count each strided layer found stays on the device; `overflowed()` reads all of them in one
    synchronisation whenever the caller wants to know (a scene with more outputs than a bound keeps
    the first `bound` in the canonical order, like the reference's bounded mode)

Live rows of the result are bit-identical to the eager, unbounded forward pass of the same scene
(tests/test_gpu_static.py).  `StaticInference` is the eval-mode runner.

Training.  A strided layer keeps its frozen bound in training mode (the Native lists the weight-gradient
kernels read come out of the same sync-free build; dead rows are in no pair), and every SparseConvTensor
carries `n_live_dev`, the device-side number of leading live rows (SubM: the input's; strided layer:
min(count found, bound); inverse convolution: its partner's input).  The BatchNorm kernels
(spconv_amd/pytorch/norm.py) take their statistics over the live rows and write zeros into the padding
rows of y and dx, so nothing that mixes rows -- batch statistics, a SubM layer's centre pair (a dead row
pairs with itself), the weight gradients -- ever sees a dead row.  Conditions: the padding rows of the
input FEATURES are zero, and either the loss has zero gradient on dead rows or the network ends in a
normalisation layer.  bench.py (`static_training_steps`) captures the whole training step of BASELINE
configs 3 and 4 this way and checks it against the eager step before timing it.
"""
import os
from typing import Dict, List, Optional, Sequence

import torch

from spconv_amd.pytorch.conv import SparseConvolution
from spconv_amd.pytorch.core import SparseConvTensor
from spconv_amd.pytorch.pool import SparseMaxPool

__all__ = ["StaticInference", "StaticTrainingStep", "strided_layers", "freeze_bounds", "dense_static"]


def strided_layers(net: torch.nn.Module) -> Dict[str, torch.nn.Module]:
    """The layers that create a new set of voxels (their output count depends on the scene)."""
    out = {}
    for name, m in net.named_modules():
        if isinstance(m, SparseConvolution) and not m.subm and not m.inverse and not m.conv1x1:
            out[name] = m
        elif isinstance(m, SparseMaxPool) and not m.subm:
            out[name] = m
    return out


def freeze_bounds(net: torch.nn.Module, bounds: Optional[Dict[str, int]] = None,
                  margin: float = 1.25) -> Dict[str, int]:
    """Sets `static_num_out` on every strided layer: from `bounds` (layer name -> rows), else from the
    layer's recorded maximum (`record_voxel_count=True`: max_num_voxels_during_training) * margin.
    Returns the bounds used.  `freeze_bounds(net, {})` with margin 0 clears them."""
    used = {}
    for name, m in strided_layers(net).items():
        if bounds is not None and name in bounds:
            b = int(bounds[name])
        elif bounds is not None and margin == 0:
            b = 0
        else:
            rec = m.get_max_num_voxels()
            if rec is None or int(rec.item()) <= 0:
                raise ValueError(f"layer {name!r}: no bound given and no recorded voxel count (construct it with "
                                 f"record_voxel_count=True and run representative scenes first)")
            b = int(int(rec.item()) * margin) + 1
        m.static_num_out = b
        used[name] = b
    return used


def dense_static(t: SparseConvTensor, channels_first: bool = True) -> torch.Tensor:
    """SparseConvTensor.dense() for a tensor with dead rows (batch index -1): they land in a scratch
    batch slot that is cut off.  No data-dependent shape: can sit inside the captured graph."""
    idx = t.indices.long()
    B, nd = t.batch_size, len(t.spatial_shape)
    dead = idx[:, 0] < 0
    b = torch.where(dead, torch.full_like(idx[:, 0], B), idx[:, 0])
    buf = t.features.new_zeros((B + 1, *t.spatial_shape, t.features.shape[1]))
    buf[(b,) + tuple(idx[:, 1 + d].clamp(min=0) for d in range(nd))] = t.features
    res = buf[:B]
    if not channels_first:
        return res
    return res.permute(0, nd + 1, *range(1, nd + 1)).contiguous()


def _declare_key_order(runner) -> None:
    """`key_ordered_input=True`: the caller vouches that every scene it loads is in ascending, unique coordinate-key
    order (spconv_amd.pytorch.utils.sort_voxels_by_coordinate).  The level's rank map is then rebuilt from the static
    index buffer at the head of every pass -- inside the captured graph: a fill and one pass over the rows, no check,
    nothing read back -- and the first level's SubM layers build their rulebooks from it instead of a hash table
    (ops.attach_rank_map)."""
    if getattr(runner, "key_ordered_input", False):
        from spconv_amd.pytorch import ops
        ops.attach_rank_map(runner.indices, runner.batch_size, runner.spatial_shape, check=False,
                            violation=runner._order_flag)


class _GatherRows(torch.autograd.Function):
    """rows[order] with the gradient scattered back (order is a permutation: every row receives exactly one term)."""

    @staticmethod
    def forward(ctx, rows, order):
        ctx.save_for_backward(order)
        return rows.index_select(0, order)

    @staticmethod
    def backward(ctx, g):
        (order,) = ctx.saved_tensors
        return torch.empty_like(g).index_copy_(0, order.long(), g), None


class _EntrySort(torch.autograd.Function):
    """features[order] out of the sort itself (ops.key_argsort carries the rows: no gather launch); leaves the
    permutation and the sorted, rank-mapped index tensor on the runner.  Gradient: scattered back through the permutation."""

    @staticmethod
    def forward(ctx, feats, runner):
        from spconv_amd.pytorch import ops
        order, idx, rows = ops.key_argsort(runner.indices, runner.batch_size, runner.spatial_shape, rank_map=True,
                                           violation=runner._order_flag, rows=feats.detach())
        runner.order, runner._idx_sorted = order, idx
        ctx.save_for_backward(order)
        return rows

    @staticmethod
    def backward(ctx, g):
        (order,) = ctx.saved_tensors
        return torch.empty_like(g).index_copy_(0, order.long(), g.contiguous()), None


def _entry_sort_default(net: torch.nn.Module) -> bool:
    """entry_sort=None: on when the network's first sparse layer (definition order) is a submanifold convolution -- its
    rulebook, and that of every SubM layer of the first level, then comes from the rank map the sort leaves behind
    instead of a hash table, which is where the sort earns more than it costs (BASELINE config 4: 2.27 -> 2.10 ms; a
    network that opens with a strided layer -- config 3 -- only gains gather locality in that one layer and loses 80 us
    to the sort), and only with the default output order of the strided layers (SPCONV_AMD_CONV_ORDER=sorted).
    SPCONV_AMD_ENTRY_SORT=1 / 0 forces it."""
    env = os.environ.get("SPCONV_AMD_ENTRY_SORT", "auto")
    if env in ("0", "1"):
        return env == "1"
    from spconv_amd import constants
    if constants.CONV_OUTPUT_ORDER != "sorted":
        # first-seen numbering of a strided layer's outputs FOLLOWS the input's row order: with sorted input rows the
        # levels behind would be numbered differently from the eager pass of the same scene
        return False
    for m in net.modules():
        if isinstance(m, SparseConvolution):
            return bool(m.subm)
        if isinstance(m, SparseMaxPool):
            return False
    return False


def _entry(runner):
    """(features, indices) the network sees.  entry_sort (default for networks that open with a SubM layer, see
    _entry_sort_default; SPCONV_AMD_ENTRY_SORT=0 or entry_sort=False turns it off): the scene is sorted by coordinate
    key at the head of every pass, inside the captured graph (ops.key_argsort: four launches that also leave the level's
    rank map behind, nothing read back), so level 1 runs like the levels behind a strided layer do -- rulebooks from a
    rank map instead of a hash table, x-neighbours in adjacent rows for every gather.  What the caller sees: nothing --
    a result that lives on the input's rows goes back into the caller's order before the runner returns it (_exit), levels
    behind a strided layer are in key order with or without the sort; `runner.order[t]` = the input row behind sorted
    row t, for code that looks at tensors INSIDE the network.  The coordinates of a scene must be unique (a voxeliser's
    are): `runner.input_order_violation()` reads the device-side verdict."""
    if runner.key_ordered_input or not runner.entry_sort:
        _declare_key_order(runner)
        return runner.features, runner.indices
    cells = int(runner.batch_size)
    for d in runner.spatial_shape:
        cells *= int(d)
    feats = runner.features
    if cells > 0xffe00000:                            # (key space beyond 32 bits)
        runner.entry_sort = False
        return feats, runner.indices
    row_bytes = (feats.numel() // max(feats.shape[0], 1)) * feats.element_size()
    if row_bytes % 4 == 0 and feats.is_contiguous():
        return _EntrySort.apply(feats, runner), runner._idx_sorted     # (features ride in the sort's bucket pass)
    from spconv_amd.pytorch import ops
    runner.order, idx = ops.key_argsort(runner.indices, runner.batch_size, runner.spatial_shape, rank_map=True,
                                        violation=runner._order_flag)
    return _GatherRows.apply(feats, runner.order), idx


class _ScatterRows(torch.autograd.Function):
    """out[order[t]] = rows[t] (the inverse of _GatherRows)."""

    @staticmethod
    def forward(ctx, rows, order):
        ctx.save_for_backward(order)
        return torch.empty_like(rows).index_copy_(0, order.long(), rows)

    @staticmethod
    def backward(ctx, g):
        (order,) = ctx.saved_tensors
        return g.index_select(0, order), None


def _exit(runner, out, idx_seen):
    """A result that lives on the INPUT's rows (a SubM-only stack, a decoder that returns onto them through inverse
    convolutions) goes back into the caller's row order: entry_sort is then invisible in what the runner returns."""
    if (idx_seen is not runner.indices and isinstance(out, SparseConvTensor) and out.indices is idx_seen):
        back = out._like(_ScatterRows.apply(out.features, runner.order), {})
        back.indices = runner.indices
        return back
    return out


def _input_order_violation(runner) -> bool:
    """True when the last replay's scene broke the entry sort's / key_ordered_input's contract (a coordinate twice, a
    live row behind a dead one): its first-level rulebooks are then not the reference's.  One synchronisation."""
    return bool(int(runner._order_flag.item()) != 0)


class StaticInference:
    """`net` (eval mode, SparseConvTensor -> SparseConvTensor or tensor) captured for scenes of at
    most `max_voxels` voxels.

        runner = StaticInference(net, max_voxels=120_000, in_channels=4, spatial_shape=[41, 1600, 1408],
                                 batch_size=1, dtype=torch.float16, bounds={"conv2": 60_000, ...})
        out = runner(features, indices)         # replay; `out` lives in static buffers
        live = out.indices[:, 0] >= 0           # rows of this scene
        runner.overflowed()                     # one synchronisation: {layer: outputs found} over a bound
        runner.release_bounds()                 # when done: eager passes of `net` are unbounded again

    Constructing the runner freezes `static_num_out` on the network's strided layers: until `release_bounds()`
    every pass of `net` -- also an eager one outside the runner -- stays bounded and padded with dead rows.
    """

    def __init__(self, net: torch.nn.Module, max_voxels: int, in_channels: int,
                 spatial_shape: Sequence[int], batch_size: int, dtype: torch.dtype = torch.float16,
                 bounds: Optional[Dict[str, int]] = None, margin: float = 1.25,
                 device: Optional[torch.device] = None, warmup: int = 2, capture_error_mode: str = "global",
                 key_ordered_input: bool = False, entry_sort: Optional[bool] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("StaticInference needs the GPU (there is no CPU path)")
        self.key_ordered_input = bool(key_ordered_input)
        self.entry_sort = _entry_sort_default(net) if entry_sort is None else bool(entry_sort)
        self.order = None
        self.net = net.eval()
        self.device = torch.device(device if device is not None else "cuda")
        self.max_voxels = int(max_voxels)
        self.spatial_shape = list(spatial_shape)
        self.batch_size = int(batch_size)
        self.bounds = freeze_bounds(net, bounds, margin)
        self._layers = strided_layers(net)
        nd = len(self.spatial_shape)
        self.features = torch.zeros((self.max_voxels, in_channels), dtype=dtype, device=self.device)
        self.indices = torch.full((self.max_voxels, nd + 1), -1, dtype=torch.int32, device=self.device)
        self.n_live = torch.zeros((1,), dtype=torch.int32, device=self.device)   # rows of the current scene
        self._order_flag = torch.zeros((1,), dtype=torch.int32, device=self.device)
        self._live = 0
        self.graph = None
        self.out = None
        with torch.cuda.device(self.device):       # (capture and replay belong to the device the buffers live on)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(max(warmup, 1)):    # allocator / option caches warm, nothing captured yet
                    self._forward()
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            # capture_error_mode="thread_local": needed when another thread may touch HIP during the capture (the
            # watchdog of an initialised RCCL process group polls its events at any time)
            with torch.no_grad(), torch.cuda.graph(g, capture_error_mode=capture_error_mode):
                self.out = self._forward()
        self.graph = g
        # the device-side counters of the captured pass (static tensors of the graph's pool)
        self._counters = {name: m._static_n_out_dev for name, m in self._layers.items()
                          if getattr(m, "_static_n_out_dev", None) is not None}

    def _forward(self):
        feats, idx = _entry(self)
        x = SparseConvTensor(feats, idx, self.spatial_shape, self.batch_size)
        x.n_live_dev = self.n_live          # (normalisation layers write zeros into the padding rows)
        return _exit(self, self.net(x), idx)

    def load(self, features: torch.Tensor, indices: torch.Tensor) -> None:
        """Copies one scene into the static input buffers (stream-ordered, no synchronisation)."""
        n = features.shape[0]
        if n > self.max_voxels:
            raise ValueError(f"scene has {n} voxels, the graph was captured for at most {self.max_voxels}")
        assert indices.shape[0] == n and indices.dtype == torch.int32
        self.features[:n].copy_(features)
        self.indices[:n].copy_(indices)
        if n < self._live:                          # rows the previous scene used and this one does not
            self.features[n:self._live].zero_()
            self.indices[n:self._live].fill_(-1)
        self.n_live.fill_(n)
        self._live = n

    def __call__(self, features: torch.Tensor, indices: torch.Tensor):
        with torch.cuda.device(self.device):
            self.load(features, indices)
            self.graph.replay()
        return self.out

    def counts(self) -> Dict[str, List[int]]:
        """{layer: [outputs found, hash-table overflow flag]} of the last replay (synchronises)."""
        if not self._counters:
            return {}
        names = list(self._counters)
        host = torch.stack([self._counters[k] for k in names]).cpu().tolist()
        return dict(zip(names, host))

    def overflowed(self) -> Dict[str, int]:
        """Layers whose last replay found more outputs than their bound (or filled their hash table):
        {layer: outputs found}.  Empty = every live row is exact."""
        return {k: c for k, (c, ovf) in self.counts().items() if c > self.bounds[k] or ovf}

    input_order_violation = _input_order_violation

    def release_bounds(self) -> None:
        """Clears the frozen bounds of the network's strided layers (eager passes are unbounded again; the
        captured graph keeps working, its shapes are baked in)."""
        for m in self._layers.values():
            m.static_num_out = 0


class StaticTrainingStep:
    """One captured TRAINING step -- rulebook builds, forward, loss, backward -- for scenes of at most
    `max_voxels` voxels; the module docstring states the conditions (zero padding in the input features; a loss
    with zero gradient on dead rows, or a network that ends in a normalisation layer).

        step = StaticTrainingStep(net, max_voxels, in_channels, spatial_shape, batch_size, bounds=...,
                                  backward=lambda out: loss_of(out).backward())      # or out_grad=<[rows, K] tensor>
        out = step(features, indices)       # replay: parameter .grad tensors now hold this scene's gradients
        optimizer.step()                    # (outside the graph; the .grad tensors are static, do not set them to None)

    `backward(out)` runs inside the capture and must be free of host reads; `out.n_live_dev` is the device-side
    number of live output rows for a masked loss.  `input_grad=True` keeps the gradient of the input features
    (`step.features.grad`).  `example=(features, indices)`: the scene the warm-up passes run on (an empty scene
    without it).  The warm-up passes are real training-mode passes; the buffers they would move -- BatchNorm
    running estimates and batch counters, recorded voxel counts -- are put back before the capture, so building
    a runner leaves a (pretrained) model's state as it found it."""

    def __init__(self, net: torch.nn.Module, max_voxels: int, in_channels: int, spatial_shape: Sequence[int],
                 batch_size: int, dtype: torch.dtype = torch.float16, bounds: Optional[Dict[str, int]] = None,
                 margin: float = 1.25, backward=None, out_grad: Optional[torch.Tensor] = None,
                 input_grad: bool = False, device: Optional[torch.device] = None, warmup: int = 2,
                 example=None, capture_error_mode: str = "global", key_ordered_input: bool = False,
                 entry_sort: Optional[bool] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("StaticTrainingStep needs the GPU (there is no CPU path)")
        self.key_ordered_input = bool(key_ordered_input)
        self.entry_sort = _entry_sort_default(net) if entry_sort is None else bool(entry_sort)
        self.order = None
        if (backward is None) == (out_grad is None):
            raise ValueError("give exactly one of `backward` (callable on the output tensor) and `out_grad`")
        self.net = net.train()
        self.device = torch.device(device if device is not None else "cuda")
        self.max_voxels, self.spatial_shape, self.batch_size = int(max_voxels), list(spatial_shape), int(batch_size)
        self.bounds = freeze_bounds(net, bounds, margin)
        self._layers = strided_layers(net)
        self._backward = backward if backward is not None else (lambda out: out.features.backward(out_grad))
        ddp = torch.nn.parallel.DistributedDataParallel
        self._defer_wgrad = not any(isinstance(m, ddp) for m in net.modules())     # (its bucket hooks read gradients as they arrive)
        nd = len(self.spatial_shape)
        self.features = torch.zeros((self.max_voxels, in_channels), dtype=dtype,
                                    device=self.device).requires_grad_(input_grad)
        self.indices = torch.full((self.max_voxels, nd + 1), -1, dtype=torch.int32, device=self.device)
        self.n_live = torch.zeros((1,), dtype=torch.int32, device=self.device)
        self._order_flag = torch.zeros((1,), dtype=torch.int32, device=self.device)
        self._live = 0
        self.out = None
        with torch.cuda.device(self.device):
            if example is not None:
                self.load(*example)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                saved = [(b, b.detach().clone()) for b in net.buffers()]
                for _ in range(max(warmup, 1)):
                    self._compute()
                with torch.no_grad():
                    for b, v in saved:              # in place: the kernels hold these tensors' addresses
                        b.copy_(v)
                del saved
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode):      # (see StaticInference)
                self._compute()             # gradients land in tensors of the graph's pool: static from here on
        self._counters = {name: m._static_n_out_dev for name, m in self._layers.items()
                          if getattr(m, "_static_n_out_dev", None) is not None}

    def _compute(self):
        self.net.zero_grad(set_to_none=True)
        self.features.grad = None
        feats, idx = _entry(self)
        x = SparseConvTensor(feats, idx, self.spatial_shape, self.batch_size)
        x.n_live_dev = self.n_live
        self.out = _exit(self, self.net(x), idx)
        if self._defer_wgrad:
            # nothing reads a weight gradient before the pass has ended: the second stages of every layer's weight
            # gradient run as one launch behind it (ops.deferred_wgrad)
            from spconv_amd.pytorch import ops
            with ops.deferred_wgrad():
                self._backward(self.out)
        else:
            self._backward(self.out)

    def load(self, features: torch.Tensor, indices: torch.Tensor) -> None:
        n = features.shape[0]
        if n > self.max_voxels:
            raise ValueError(f"scene has {n} voxels, the graph was captured for at most {self.max_voxels}")
        assert indices.shape[0] == n and indices.dtype == torch.int32
        with torch.no_grad():
            self.features[:n].copy_(features)
            self.indices[:n].copy_(indices)
            if n < self._live:
                self.features[n:self._live].zero_()
                self.indices[n:self._live].fill_(-1)
            self.n_live.fill_(n)
        self._live = n

    def __call__(self, features: torch.Tensor, indices: torch.Tensor):
        with torch.cuda.device(self.device):
            self.load(features, indices)
            self.graph.replay()
        return self.out

    counts = StaticInference.counts
    input_order_violation = _input_order_violation
    overflowed = StaticInference.overflowed
    release_bounds = StaticInference.release_bounds
