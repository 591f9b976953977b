from spconv_amd.pytorch.quantization.quantized.conv import SparseConv  # noqa: F401
