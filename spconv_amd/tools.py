"""Per-kernel-group timer and failure dump of the reference's auxiliary layer.

* ``CUDAKernelTimer`` -- API of spconv/tools.py:23-78 (namespace / record / get_all_pair_time /
  collect_by_name).  The reference wraps cumm's C++ event timer; here a record is a pair of HIP events
  on torch's current stream, durations [ms] are read when asked for (one synchronisation).  A
  ``SparseConvTensor(..., enable_timer=True)`` carries one through the network and every sparse layer
  records its rulebook build ("<layer>.gen_pairs") and its kernels ("<layer>.forward").
* ``save_debug_data`` -- spconv/debug_utils.py:20-32: when ``SPCONV_DEBUG_SAVE_PATH`` is set, the
  inputs of a failing rulebook build are pickled there (conv.py:289-297 does this on exceptions).
"""
from __future__ import annotations

import contextlib
import os
import pickle
from pathlib import Path
from typing import Dict, List, Tuple

import torch

SPCONV_DEBUG_SAVE_PATH = os.getenv("SPCONV_DEBUG_SAVE_PATH", "")


class CUDAKernelTimer:
    def __init__(self, enable: bool = True) -> None:
        self.enable = bool(enable) and torch.cuda.is_available()
        self._stack: List[str] = []
        self._pairs: List[Tuple[str, torch.cuda.Event, torch.cuda.Event]] = []

    @contextlib.contextmanager
    def _namespace(self, name: str):
        self._stack.append(name)
        try:
            yield
        finally:
            self._stack.pop()

    @contextlib.contextmanager
    def _record(self, name: str, stream: int = 0):
        self._stack.append(name)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        try:
            start.record()
            yield
            stop.record()
            self._pairs.append((".".join(self._stack), start, stop))
        finally:
            self._stack.pop()

    def namespace(self, name: str):
        return self._namespace(name) if self.enable else contextlib.nullcontext()

    def record(self, name: str, stream: int = 0):
        return self._record(name, stream) if self.enable else contextlib.nullcontext()

    def get_all_pair_time(self) -> Dict[str, float]:
        """name -> accumulated device time [ms] of every record made so far."""
        if not self.enable:
            return {}
        torch.cuda.synchronize()
        res: Dict[str, float] = {}
        for name, a, b in self._pairs:
            res[name] = res.get(name, 0.0) + a.elapsed_time(b)
        return res

    @staticmethod
    def collect_by_name(name: str, res: Dict[str, float]):
        return {k: v for k, v in res.items() if name in k.split(".")}


def save_debug_data(data) -> None:
    if not SPCONV_DEBUG_SAVE_PATH:
        return
    try:
        path = Path(SPCONV_DEBUG_SAVE_PATH)
        assert path.parent.exists(), "parent of SPCONV_DEBUG_SAVE_PATH must exist"
        with path.open("wb") as f:
            pickle.dump(data, f)
        print(f"spconv_amd saved debug data to {SPCONV_DEBUG_SAVE_PATH}: attach it to the issue together with the log")
    except Exception as e:  # noqa: BLE001 -- a failing dump must not hide the original error
        print(f"spconv_amd tried to save debug data to {SPCONV_DEBUG_SAVE_PATH} but failed with {e}")
