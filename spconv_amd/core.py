"""``spconv.core`` as user code imports it: the algorithm enums (reference ``spconv/core.py:22-36``).
The rest of the reference's module is its CUTLASS-style kernel parameter tables (``core.py:38-1343``),
which have no counterpart here: there is one hand-written kernel per shape class and no tuner."""
from enum import Enum

from spconv_amd.pytorch.core import ConvAlgo  # noqa: F401


class AlgoHint(Enum):
    """Bit flags of the reference's tuner hints (accepted for signature parity, never consulted;
    the second member keeps the reference's spelling, ``core.py:33``)."""
    NoHint = 0b000
    Fowrard = 0b001
    BackwardInput = 0b010
    BackwardWeight = 0b100
