"""Containers that route SparseConvTensor through sparse and dense modules.

API parity: ``spconv/pytorch/modules.py:50-185`` (SparseModule, SparseSequential,
SparseBatchNorm / SparseSyncBatchNorm / SparseReLU / SparseIdentity,
assign_name_for_sparse_modules).
"""
from __future__ import annotations

import sys
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
    """nn.Sequential that applies dense modules to ``.features`` of a SparseConvTensor."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if sys.version_info < (3, 6):
                raise ValueError("kwargs only supported in py36+")
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        for module in self._modules.values():
            if is_spconv_module(module):
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                # dense layers see the [N, C] feature matrix; skipped for empty tensors
                if input.indices.shape[0] != 0:
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
