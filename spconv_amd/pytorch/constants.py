"""``spconv.pytorch.constants`` (reference ``spconv/pytorch/constants.py:15-40``)."""
import re

import torch

PYTORCH_VERSION = [int(x) for x in re.match(r"(\d+)\.(\d+)\.(\d+)", torch.__version__).groups()]
TORCH_HAS_AMP = True          # torch >= 1.6: every version this package runs on


def is_amp_enabled() -> bool:
    return torch.is_autocast_enabled()
