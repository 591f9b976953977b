"""Containers that route SparseConvTensor through sparse and dense modules.

API parity: ``spconv/pytorch/modules.py:50-185`` (SparseModule, SparseSequential,
SparseBatchNorm / SparseSyncBatchNorm / SparseReLU / SparseIdentity,
assign_name_for_sparse_modules).
"""
from __future__ import annotations

from collections import OrderedDict

from torch import nn

from spconv_amd.pytorch.core import SparseConvTensor


class SparseModule(nn.Module):
    """Marker base class: SparseSequential hands these a SparseConvTensor."""

    def __init__(self, name=None):
        super().__init__()
        self.name = name
        self._sparse_unique_name = ""


def is_spconv_module(module) -> bool:
    return isinstance(module, SparseModule)


def is_sparse_conv(module) -> bool:
    from spconv_amd.pytorch.conv import SparseConvolution
    return isinstance(module, SparseConvolution)


class SparseSequential(SparseModule):
    """Ordered container: sparse modules receive the SparseConvTensor, any other module is
    applied to its ``.features`` matrix.

    Construction follows ``nn.Sequential`` (positional modules numbered from "0", or one ordered
    mapping of names to modules) and additionally accepts named modules as keyword arguments,
    as the reference container does (modules.py:59-145)."""

    def __init__(self, *modules, **named):
        super().__init__()
        if len(modules) == 1 and isinstance(modules[0], OrderedDict):
            entries = list(modules[0].items())
        else:
            entries = [(str(i), m) for i, m in enumerate(modules)]
        for key, m in entries + list(named.items()):
            self._register(key, m)

    def _register(self, key, module):
        if key in self._modules:
            raise ValueError(f"a module named {key!r} is already part of this SparseSequential")
        self.add_module(key, module)

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, idx):
        n = len(self._modules)
        if isinstance(idx, slice):
            return SparseSequential(OrderedDict(list(self._modules.items())[idx]))
        if not -n <= idx < n:
            raise IndexError(f"index {idx} is out of range for a SparseSequential of {n} modules")
        return list(self._modules.values())[idx % n]

    def __iter__(self):
        return iter(self._modules.values())

    def add(self, module, name=None):
        """Appends `module` under `name` (default: its position)."""
        self._register(str(len(self._modules)) if name is None else name, module)

    def forward(self, input):
        from spconv_amd.pytorch import prefetch
        mods = list(self._modules.values())
        # the chain's rulebooks depend on the coordinates alone: built ahead on a side stream where no build reads
        # anything back (spconv_amd/pytorch/prefetch.py); the layers below pick them up, or build in line
        chain = prefetch.start(mods, input) if isinstance(input, SparseConvTensor) else None
        if chain is None:
            return self._run(mods, input)
        try:
            return self._run(mods, input)
        finally:
            chain.finish()

    def _run(self, mods, input):
        from spconv_amd.pytorch import norm, ops
        i = 0
        conv_stats = None       # BatchNorm statistics the convolution just before a normalisation layer left (or None)
        while i < len(mods):
            module = mods[i]
            i += 1
            if is_spconv_module(module):
                nxt = mods[i] if i < len(mods) else None
                conv_stats = None
                if isinstance(nxt, nn.modules.batchnorm._BatchNorm):
                    # a training-mode BatchNorm1d on the fused path behind a bias-free convolution: its statistics
                    # come out of the convolution's epilogue (ops.collect_bn_stats), it starts at the merge step
                    fuse_stats = (ops.BN_EPILOGUE and norm.ENABLED and type(nxt) is nn.BatchNorm1d
                                  and (nxt.training or nxt.running_mean is None) and is_sparse_conv(module)
                                  and getattr(module, "bias", None) is None and module.training
                                  and not module._forward_hooks and not nxt._forward_hooks
                                  and not nxt._forward_pre_hooks)
                    with ops.output_stays_cached():     # the normalisation reads the rows next (ops._OUT_CACHED)
                        if fuse_stats:
                            with ops.collect_bn_stats() as conv_stats:
                                input = module(input)
                        else:
                            input = module(input)
                else:
                    input = module(input)
            elif isinstance(input, SparseConvTensor):
                # dense layers see the [N, C] feature matrix; skipped for empty tensors
                if input.indices.shape[0] != 0:
                    if norm.supported(input.features, module):
                        # BatchNorm1d (+ the ReLU right behind it) in the streaming kernels of csrc/norm.hip
                        # (a ReLU with user hooks -- feature extractors, CAM tools, observers -- is called as a
                        # module so that they fire)
                        fuse = (i < len(mods) and type(mods[i]) is nn.ReLU and not mods[i]._forward_hooks
                                and not mods[i]._forward_pre_hooks and not mods[i]._backward_hooks
                                and not getattr(mods[i], "_backward_pre_hooks", None))
                        input = input.replace_feature(norm.batch_norm(input.features, module, relu=fuse,
                                                                      n_live=getattr(input, "n_live_dev", None),
                                                                      stats=conv_stats))
                        conv_stats = None
                        i += 1 if fuse else 0
                    else:
                        if (getattr(input, "n_live_dev", None) is not None and getattr(module, "training", False)
                                and isinstance(module, nn.modules.batchnorm._BatchNorm)):
                            raise RuntimeError("a static-shape tensor (padding rows) needs the batch statistics of "
                                               "csrc/norm.hip; this BatchNorm layer takes torch's path "
                                               "(norm.supported: dtype / channel count / hooks)")
                        input = input.replace_feature(module(input.features))
            else:
                input = module(input)
        return input


def assign_name_for_sparse_modules(module: nn.Module):
    for k, n in module.named_modules():
        if isinstance(n, SparseModule):
            n._sparse_unique_name = k


def _on_features(base):
    def forward(self, input):
        if isinstance(input, SparseConvTensor):
            return input.replace_feature(base.forward(self, input.features))
        return base.forward(self, input)
    return forward


class SparseBatchNorm(nn.BatchNorm1d):
    forward = _on_features(nn.BatchNorm1d)


class SparseSyncBatchNorm(nn.SyncBatchNorm):
    forward = _on_features(nn.SyncBatchNorm)


class SparseReLU(nn.ReLU):
    forward = _on_features(nn.ReLU)


class SparseIdentity(nn.Identity):
    forward = _on_features(nn.Identity)


class PrintTensorMeta(nn.Module):
    """Debug layer: prints min / max / mean of the (features of the) tensor passing through
    (reference modules.py:186-193)."""

    def forward(self, x):
        ft = x.features if isinstance(x, SparseConvTensor) else x
        print(ft.min(), ft.max(), ft.mean())
        return x


class PrintCurrentTime(nn.Module):
    """Debug layer: prints the wall time since construction, after draining the GPU queue
    (reference modules.py:195-208)."""

    def __init__(self) -> None:
        super().__init__()
        import time
        self._time = time
        self.first_time = time.time()

    def forward(self, x, msg="", reset: bool = False):
        import torch
        if reset:
            self.first_time = self._time.time()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        print(msg, self._time.time() - self.first_time)
        return x
