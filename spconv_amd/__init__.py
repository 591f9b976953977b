"""spconv_amd -- MI355X (gfx950) native spatially-sparse convolution.

Drop-in for the hot path of traveller59/spconv: ``import spconv_amd.pytorch as
spconv`` gives SparseConvTensor / SubMConv3d / SparseConv3d / SparseSequential
with the reference's signatures; compute runs in hand-written HIP kernels
behind the C ABI of ``include/spconv_amd.h``.
"""
import sys

__version__ = "0.1.0"


def install_as_spconv() -> None:
    """Alias this package as ``spconv`` so ``import spconv.pytorch as spconv``
    in existing model code resolves to the MI355X implementation."""
    import spconv_amd
    import spconv_amd.pytorch as sp
    sys.modules.setdefault("spconv", spconv_amd)
    sys.modules.setdefault("spconv.pytorch", sp)
    for name in ("core", "conv", "functional", "ops", "modules", "pool", "hash", "utils", "tables",
                 "identity"):
        sys.modules.setdefault(f"spconv.pytorch.{name}", getattr(sp, name))
    # the quantization package tree (spconv.pytorch.quantization.intrinsic.qat, ...)
    import importlib
    import pkgutil
    import spconv_amd.pytorch.quantization as q
    sys.modules.setdefault("spconv.pytorch.quantization", q)
    for info in pkgutil.walk_packages(q.__path__, q.__name__ + "."):
        mod = importlib.import_module(info.name)
        sys.modules.setdefault(info.name.replace("spconv_amd.", "spconv.", 1), mod)
    for name in ("constants", "core", "tools", "debug_utils", "pytorch.spatial", "pytorch.constants"):
        sys.modules.setdefault(f"spconv.{name}", importlib.import_module(f"spconv_amd.{name}"))
