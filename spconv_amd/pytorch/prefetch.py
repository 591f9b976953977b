"""Rulebooks of a layer chain, built AHEAD of the layers on a side stream.

The rulebook of a layer depends on the coordinates alone (reference: `get_indice_pairs*` take `indices`, never
`features`, spconv/pytorch/ops.py:132-326,329-808), so in a chain of layers every rulebook of the chain is known as soon
as the chain's input is: level L + 1's strided build needs level L's coordinates, nothing else.  The reference builds
them where they are first used, in line with the convolutions (conv.py:247-278); on MI355X a rulebook build is a string
of small, latency-bound launches (3-35 us each, a few hundred workgroups) that leave most of the chip idle, and in a
backbone they are a quarter of a training step and half of an inference pass.

`SparseSequential` therefore walks its children once per forward pass and, when nothing in the chain needs a device ->
host read (every strided layer carries a frozen output bound: spconv_amd.pytorch.static; SubM layers never read back),
builds all rulebooks -- tables, rows layouts, Native lists, weight-gradient range plans -- on ONE side stream that was
forked from the current stream, recording an event behind each build.  A layer then finds its rulebook already there
(`take`): it makes the current stream wait for that build's event and goes on with the gather-GEMM.  Inside a stream
capture this gives the graph two branches (rulebook chain | convolution + normalisation chain) that the graph executor
runs concurrently; outside a capture the two streams overlap through separate hardware queues.

What a layer receives is checked, not trusted: the rulebook is only taken when it was built from exactly the index
tensor (same object), batch size and spatial shape the layer is now called with; anything else (a module that changed
the coordinates in between, a different tensor) drops the prefetched rulebook and builds in line as before.  Results
are the same bit for bit either way: the same library calls with the same arguments, on another stream.

Measured (MI355X, BASELINE config 4, 4 x 100 k voxels): captured inference pass 1.096 -> 0.933 ms, captured training step
2.62 -> 2.50 ms.  The gain is what the overlap buys MINUS what a graph with two branches costs on this stack: a
single-branch graph is replayed from pre-built AQL packets, a forked one is dispatched node by node (probe:
tools/probes/graph_branch_probe.py -- 60 small + 60 streaming kernels: 625 us in one branch, 592 forked, 520 for the
streaming chain alone).  An EAGER pass is paced by the host (3.6 -> 3.9 ms with the extra stream bookkeeping), so:

Switch: SPCONV_AMD_PREFETCH = auto (default: inside a stream capture, for chains without a read-back) | 1 (always, eager
passes too; strided layers without a bound are built ahead with their read-back) | 0 (never).
"""
from __future__ import annotations

import os
import threading
from typing import List, Optional

import torch

_MODE = os.environ.get("SPCONV_AMD_PREFETCH", "auto")
_ATTR = "_spx_prefetched"
_side_streams = {}            # device index -> the side stream rulebook chains run on
_state = threading.local()    # .active: a chain is being served on this thread (nested containers do not start another)


def set_mode(mode: str) -> str:
    """'auto' | '1' | '0' (tests and tools; the environment variable sets the default).  Returns the previous mode."""
    global _MODE
    prev, _MODE = _MODE, str(mode)
    return prev


class _Prefetched:
    __slots__ = ("rb", "event", "indices", "batch_size", "spatial_shape")


def _rulebook_tensors(rb):
    for name in ("out_indices", "pair_fwd", "pair_bwd", "mask_fwd", "mask_bwd", "_pair_native", "_num_per_loc",
                 "layout", "rankmap", "n_out_dev", "out_n_live_dev", "wgrad_plan", "argsort_fwd", "argsort_bwd"):
        t = getattr(rb, name, None)
        if isinstance(t, torch.Tensor) and t.is_cuda:
            yield t
    for pair in getattr(rb, "sorted_tables", {}).values():
        for t in pair:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                yield t


def flatten(mods) -> list:
    """Children of a container in execution order, nested plain SparseSequential containers opened up."""
    from spconv_amd.pytorch.modules import SparseSequential
    out = []
    for m in mods:
        if type(m) is SparseSequential and not m._forward_hooks and not m._forward_pre_hooks:
            out.extend(flatten(m._modules.values()))
        else:
            out.append(m)
    return out


def _plan(mods, input) -> List:
    """[(module, builds?)] for the leading part of the chain whose coordinate flow is known: sparse convolutions and
    dense modules (which only touch `.features`).  Stops at anything else (pooling, ToDense, user blocks)."""
    from spconv_amd.pytorch.conv import SparseConvolution
    from spconv_amd.pytorch.modules import is_spconv_module
    from torch import nn
    keys = set(input.indice_dict.keys())
    todo = []
    for m in flatten(mods):
        if isinstance(m, SparseConvolution):
            if type(m).forward is not SparseConvolution.forward and not getattr(m, "_spx_prefetch_ok", False):
                break                    # (a subclass with its own forward: quantised modules, user layers)
            if m.conv1x1:
                continue                 # (keeps the coordinates, builds nothing)
            if m.inverse or m.transposed:
                break
            if m.indice_key is not None and m.indice_key in keys:
                if m.subm:
                    continue             # (reuses the rulebook of an earlier layer of the chain)
                break
            if not m.subm and _MODE != "1" and not int(getattr(m, "static_num_out", 0) or 0):
                break                    # (its build reads the output count back: stays in line)
            todo.append(m)
            if m.indice_key is not None:
                keys.add(m.indice_key)
        elif is_spconv_module(m):
            break
        elif not isinstance(m, nn.Module):
            break
    return todo


class Chain:
    """The prefetch of one container call: `finish()` joins the side stream and clears what no layer took."""

    def __init__(self, modules, side, main):
        self.modules, self.side, self.main = modules, side, main

    def finish(self) -> None:
        self.main.wait_stream(self.side)
        for m in self.modules:
            m.__dict__.pop(_ATTR, None)
        _state.active = False


def _side_stream(dev) -> "torch.cuda.Stream":
    side = _side_streams.get(dev.index)
    if side is None:
        side = _side_streams[dev.index] = torch.cuda.Stream(device=dev)
    return side


def start(mods, input) -> Optional[Chain]:
    """Builds the rulebooks of the chain `mods` (children of a SparseSequential) for `input` on the side stream."""
    if _MODE == "0" or getattr(_state, "active", False):
        return None
    feats = input.features
    if not isinstance(feats, torch.Tensor) or not isinstance(input.indices, torch.Tensor):
        return None
    if not feats.is_cuda or input.indices.shape[0] == 0 or input.benchmark or input._timer is not None:
        return None
    dev = feats.device
    side = _side_stream(dev)              # (created by the first eager pass: never inside a capture that follows warm-up)
    if _MODE != "1" and not torch.cuda.is_current_stream_capturing():
        return None                       # (an eager pass is paced by the host: a second stream buys nothing there)
    todo = _plan(mods, input)
    if len(todo) < 2:                     # (a single build is on the critical path anyway)
        return None
    with torch.cuda.device(dev):
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        _state.active = True
        chain = Chain(todo, side, main)
        try:
            from spconv_amd.pytorch import ops
            indices, shape, bs = input.indices, list(input.spatial_shape), input.batch_size
            n_live = getattr(input, "n_live_dev", None)
            late = []
            with torch.cuda.stream(side):
                for m in todo:
                    rb = m._build_rulebook(indices, bs, shape, feats.dtype, n_live)
                    if rb.has_native and torch.is_grad_enabled():
                        late.append(rb)
                    p = _Prefetched()
                    p.rb, p.indices, p.batch_size, p.spatial_shape = rb, indices, bs, list(shape)
                    p.event = torch.cuda.Event()
                    p.event.record(side)
                    for t in _rulebook_tensors(rb):
                        t.record_stream(main)    # (allocated on the side stream, read on the caller's)
                    m.__dict__[_ATTR] = p
                    if not m.subm:
                        indices, shape, n_live = rb.out_indices, list(rb.out_shape), rb.out_n_live_dev
                # the weight-gradient range plans -- read by the BACKWARD pass only -- behind the tables of every level:
                # off the backward's critical path, and no layer of the forward pass waits for them either (built right
                # behind each rulebook they delayed every later hand-over by ~10 us: config 4 2.397 -> 2.386 ms)
                for rb in late:
                    plan = ops._plan_of(rb)
                    if plan is not None:
                        plan.record_stream(main)
        except Exception:
            chain.finish()
            raise
    return chain


def take(module, indices, batch_size, spatial_shape):
    """The rulebook that was built ahead for `module`, if it was built from exactly these inputs; else None."""
    p = module.__dict__.pop(_ATTR, None)
    if p is None:
        return None
    if p.indices is not indices or p.batch_size != batch_size or p.spatial_shape != list(spatial_shape):
        return None
    torch.cuda.current_stream(indices.device).wait_event(p.event)
    return p.rb
