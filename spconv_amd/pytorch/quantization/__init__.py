"""int8 inference for the convolution hot path (reference: ``spconv/pytorch/quantization/``).

Only the statically quantised convolution module is provided
(``quantized.SparseConv``, reference ``quantization/quantized/conv.py:45-378``); the
torch.fx PTQ/QAT graph tooling of the reference is out of scope."""
from spconv_amd.pytorch.quantization import quantized  # noqa: F401
