"""Quantize helpers that accept SparseConvTensor (reference ``quantization/core.py:10-37``)."""
from __future__ import annotations

import torch

from spconv_amd.pytorch.core import SparseConvTensor


def _q(t, scale, zero_point, dtype):
    if isinstance(t, SparseConvTensor):
        return t.replace_feature(torch.quantize_per_tensor(t.features, scale, zero_point, dtype))
    return torch.quantize_per_tensor(t, scale, zero_point, dtype)


def quantize_per_tensor(ten, scale, zero_point, dtype):
    """``torch.quantize_per_tensor`` for tensors, sparse tensors (features are quantised, the
    coordinate set is shared) and lists of either (then scale / zero_point are lists too)."""
    if isinstance(ten, (list, tuple)):
        return [_q(v, scale[i], zero_point[i], dtype) for i, v in enumerate(ten)]
    return _q(ten, scale, zero_point, dtype)


def quantized_add(x: torch.Tensor, y: torch.Tensor, scale, zero_point):
    """qint8 + qint8 -> qint8 with the output scale (symmetric, zero point 0):
    ``clip(round((x_i8 * s_x + y_i8 * s_y) / scale))``."""
    acc = (x.int_repr().to(torch.float32) * x.q_scale() + y.int_repr().to(torch.float32) * y.q_scale()) / scale
    q = torch.clip(torch.round(acc), -128, 127).to(torch.int8)
    return torch._make_per_tensor_quantized_tensor(q, float(scale), int(zero_point))
