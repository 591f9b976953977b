"""fx graph rewrites applied after ``convert_fx`` (reference ``quantization/graph.py:8-56``)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.fx
from torch import nn

from spconv_amd.pytorch.quantization.core import quantize_per_tensor, quantized_add
from spconv_amd.pytorch.quantization.intrinsic import quantized as snniq


def is_dequantize_node(node) -> bool:
    return isinstance(node, torch.fx.Node) and node.op == "call_method" and node.target == "dequantize"


def _module_of(node: torch.fx.Node, modules: Dict[str, nn.Module]) -> Optional[nn.Module]:
    return modules.get(str(node.target)) if node.op == "call_module" else None


def _finish(m: torch.fx.GraphModule) -> torch.fx.GraphModule:
    m.graph.eliminate_dead_code()
    m.recompile()
    m.graph.lint()
    return m


def remove_conv_add_dq(model: torch.fx.GraphModule) -> torch.fx.GraphModule:
    """The fused conv + add + relu kernel takes the residual in int8: drop the ``dequantize``
    feeding its second input."""
    modules = dict(model.named_modules(remove_duplicate=False))
    for n in model.graph.nodes:
        if type(_module_of(n, modules)) == snniq.SparseConvAddReLU and len(n.args) > 1 \
                and is_dequantize_node(n.args[1]):
            dq = n.args[1]
            n.replace_input_with(dq, dq.args[0])
    return _finish(model)


def transform_qdq(m: torch.fx.GraphModule) -> torch.fx.GraphModule:
    """``torch.quantize_per_tensor`` / ``quantized.add`` do not take SparseConvTensor: retarget
    them at the sparse-aware helpers."""
    for node in m.graph.nodes:
        if node.op == "call_function":
            if node.target == torch.quantize_per_tensor:
                node.target = quantize_per_tensor
            elif node.target == torch.ops.quantized.add:
                node.target = quantized_add
    return _finish(m)
