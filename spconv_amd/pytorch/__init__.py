"""``import spconv_amd.pytorch as spconv`` -- the reference's ``spconv.pytorch`` namespace
(``spconv/pytorch/__init__.py:1-41``) for the convolution hot path."""
from spconv_amd.pytorch import conv, core, functional, hash, identity, modules, ops, pool, tables, utils
from spconv_amd.pytorch.conv import (SparseConv1d, SparseConv2d, SparseConv3d, SparseConv4d,
                                     SparseConvolution, SparseConvTranspose1d,
                                     SparseConvTranspose2d, SparseConvTranspose3d,
                                     SparseConvTranspose4d, SparseInverseConv1d,
                                     SparseInverseConv2d, SparseInverseConv3d,
                                     SparseInverseConv4d, SubMConv1d, SubMConv2d, SubMConv3d,
                                     SubMConv4d)
from spconv_amd.pytorch.core import ConvAlgo, SparseConvTensor
from spconv_amd.pytorch.identity import Identity
from spconv_amd.pytorch.tables import AddTable, AddTableMisaligned, ConcatTable, JoinTable
from spconv_amd.pytorch.pool import (SparseAvgPool1d, SparseAvgPool2d, SparseAvgPool3d,
                                     SparseGlobalAvgPool, SparseGlobalMaxPool, SparseMaxPool1d,
                                     SparseMaxPool2d, SparseMaxPool3d, SparseMaxPool4d)
from spconv_amd.pytorch.modules import (SparseBatchNorm, SparseIdentity, SparseModule, SparseReLU,
                                        SparseSequential, SparseSyncBatchNorm,
                                        assign_name_for_sparse_modules)


def _autograd_on_the_calling_thread() -> str:
    """An EAGER training step of a sparse layer is a few launches of 5-15 us each; torch's autograd engine runs the
    backward of CUDA tensors on a per-device worker thread that the calling thread wakes every step, and on a many-core
    host that hand-over costs 70-130 us per step -- more than the kernels (a 100 k-voxel SubMConv3d step: 76 us with the
    backward on the calling thread, 150-210 us through the worker; profiles/r05_experiments.md section 8).  A process that
    drives ONE GPU has nothing to gain from the worker, so importing this package switches the engine to the calling
    thread there (`torch.autograd.set_multithreading_enabled(False)`); a process that sees several GPUs outside a launcher (no LOCAL_RANK) keeps torch's
    default (the workers run the devices' backward passes side by side) and is told once.  SPCONV_AMD_AUTOGRAD_THREADS =
    auto (default) | keep (never touch the engine) | single (always switch).  The reference needs no such switch: its
    layer step is one pybind call per direction with the tuner's kernel behind it (spconv/pytorch/cppcore.py:65-109)."""
    import os
    import warnings
    import torch
    mode = os.environ.get("SPCONV_AMD_AUTOGRAD_THREADS", "auto")
    if mode == "keep" or not hasattr(torch.autograd, "set_multithreading_enabled"):
        return "kept"
    try:
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        n_dev = 0
    # (a launcher's environment -- torchrun sets LOCAL_RANK -- means one process per GPU even though all are visible)
    if mode == "single" or n_dev == 1 or (n_dev > 1 and "LOCAL_RANK" in os.environ):
        torch.autograd.set_multithreading_enabled(False)
        return "calling thread"
    if n_dev > 1:
        warnings.warn("spconv_amd: this process sees %d GPUs, torch's multi-threaded autograd engine is left on; an EAGER "
                      "sparse-layer training step then pays a thread hand-over per backward (2-3x the kernels' time).  "
                      "One process per GPU, or torch.autograd.set_multithreading_enabled(False) / "
                      "SPCONV_AMD_AUTOGRAD_THREADS=single, removes it." % n_dev, stacklevel=3)
    return "kept"


AUTOGRAD_ENGINE = _autograd_on_the_calling_thread()


class ToDense(SparseModule):
    """convert SparseConvTensor to NCHW dense tensor."""

    def forward(self, x: SparseConvTensor):
        return x.dense()


class RemoveGrid(SparseModule):
    """remove pre-allocated grid buffer."""

    def forward(self, x: SparseConvTensor):
        x.grid = None
        return x
