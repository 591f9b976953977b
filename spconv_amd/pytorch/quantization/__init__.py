"""int8 for the convolution hot path (reference ``spconv/pytorch/quantization/``): the statically
quantised modules over the int8 MFMA kernel (``quantized``, ``intrinsic.quantized``), BatchNorm /
activation folding (``utils``), QAT modules (``intrinsic.qat``), sparse-aware observers and
qconfigs (``fake_q``), module-swap tables (``qmapping``) and the torch.ao fx backend description
(``backend_cfg``, ``graph``)."""
from spconv_amd.pytorch.quantization import quantized  # noqa: F401
from spconv_amd.pytorch.quantization.backend_cfg import (get_spconv_backend_config,  # noqa: F401
                                                         get_spconv_convert_custom_config,
                                                         get_spconv_prepare_custom_config,
                                                         prepare_spconv_torch_inference)
from spconv_amd.pytorch.quantization.core import quantize_per_tensor  # noqa: F401
from spconv_amd.pytorch.quantization.fake_q import (get_default_spconv_qconfig_mapping,  # noqa: F401
                                                    get_default_spconv_trt_ptq_qconfig,
                                                    get_default_spconv_trt_qat_qconfig)
from spconv_amd.pytorch.quantization.graph import remove_conv_add_dq, transform_qdq  # noqa: F401
from spconv_amd.pytorch.quantization.qmapping import (get_spconv_fmod_to_qat_mapping,  # noqa: F401
                                                      get_spconv_qat_to_static_mapping)
from spconv_amd.pytorch.quantization.quantized.conv import SparseConv  # noqa: F401
