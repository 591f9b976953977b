"""Fixed-size GPU hash table for 32 / 64 bit keys and values (reference:
``spconv/pytorch/hash.py:29-170``).  Same constructor and methods; storage is two torch tensors
owned by this object, the operations run in ``spx_hash_*``.  ``assign_arange_`` / ``items``
enumerate the table in slot order, i.e. deterministically for a given key set."""
from __future__ import annotations

from typing import Optional

import contextlib
import functools

import torch

from spconv_amd import _lib


def _on_table_device(method):
    """Runs a method with the table's device current (torch's DeviceGuard convention): the native calls
    and their scratch allocations belong to the device the table lives on, also when the caller never
    called torch.cuda.set_device."""
    @functools.wraps(method)
    def guarded(self, *args, **kwargs):
        dev = self.keys_data.device
        ctx = (torch.cuda.device(dev) if dev.type == "cuda" and dev.index != torch.cuda.current_device()
               else contextlib.nullcontext())
        with ctx:
            return method(self, *args, **kwargs)
    return guarded

_ITEMSIZE = {torch.int32: 4, torch.int64: 8, torch.float32: 4, torch.float64: 8}


class HashTable:
    def __init__(self, device: torch.device, key_dtype: torch.dtype, value_dtype: torch.dtype,
                 max_size: int = -1) -> None:
        device = torch.device(device)
        if device.type != "cuda":
            raise NotImplementedError("spconv_amd runs on MI355X only: HashTable needs a cuda device")
        assert key_dtype in (torch.int32, torch.int64), "key must be int32/int64"
        assert value_dtype in _ITEMSIZE, "value must be a 32 or 64 bit type"
        assert max_size > 0, ("you must provide max_size for fixed-size cuda hash table, usually *2 "
                              "of num of keys")
        self.is_cpu = False
        self.key_dtype = key_dtype
        self.value_dtype = value_dtype
        self.key_itemsize = _ITEMSIZE[key_dtype]
        self.value_itemsize = _ITEMSIZE[value_dtype]
        self.keys_data = torch.empty([max_size], dtype=key_dtype, device=device)
        self.values_data = torch.empty([max_size], dtype=value_dtype, device=device)
        self._L = _lib.load()
        with torch.cuda.device(device):
            _lib.check(self._L.spx_hash_clear(self.keys_data.data_ptr(), max_size, self.key_itemsize,
                                              self._stream()))

    def _stream(self):
        return torch.cuda.current_stream(self.keys_data.device).cuda_stream

    def _args(self):
        return (self.keys_data.data_ptr(), self.values_data.data_ptr(), self.keys_data.shape[0],
                self.key_itemsize, self.value_itemsize)

    def _check_keys(self, keys: torch.Tensor) -> torch.Tensor:
        assert keys.dtype == self.key_dtype and keys.ndim == 1, "keys must be 1-d with the table's key dtype"
        return keys.contiguous()

    @_on_table_device
    def insert(self, keys: torch.Tensor, values: Optional[torch.Tensor] = None):
        """insert keys (and values; without values the stored value is undefined)"""
        keys = self._check_keys(keys)
        if values is not None:
            assert values.dtype == self.value_dtype and values.shape[0] == keys.shape[0]
            values = values.contiguous()
        _lib.check(self._L.spx_hash_insert(*self._args(), keys.data_ptr(),
                                           None if values is None else values.data_ptr(),
                                           keys.shape[0], self._stream()))

    @_on_table_device
    def query(self, keys: torch.Tensor, values: Optional[torch.Tensor] = None):
        """-> (values, bool tensor that is True where the key was NOT found)"""
        keys = self._check_keys(keys)
        if values is None:
            values = torch.empty([keys.shape[0]], dtype=self.value_dtype, device=keys.device)
        is_empty = torch.empty([keys.shape[0]], dtype=torch.uint8, device=keys.device)
        _lib.check(self._L.spx_hash_query(*self._args(), keys.data_ptr(), values.data_ptr(),
                                          is_empty.data_ptr(), keys.shape[0], self._stream()))
        return values, is_empty > 0

    @_on_table_device
    def insert_exist_keys(self, keys: torch.Tensor, values: torch.Tensor):
        """overwrite the values of keys that exist; -> uint8 tensor, 1 where the key was missing"""
        keys = self._check_keys(keys)
        assert values.dtype == self.value_dtype and values.shape[0] == keys.shape[0]
        is_empty = torch.empty([keys.shape[0]], dtype=torch.uint8, device=keys.device)
        _lib.check(self._L.spx_hash_insert_exist(*self._args(), keys.data_ptr(),
                                                 values.contiguous().data_ptr(), is_empty.data_ptr(),
                                                 keys.shape[0], self._stream()))
        return is_empty

    def _ws(self):
        return torch.empty((int(self._L.spx_hash_ws_bytes(self.keys_data.shape[0])),), dtype=torch.uint8,
                           device=self.keys_data.device)

    @_on_table_device
    def assign_arange_(self):
        """every key gets a distinct value in [0, count); -> count (1-element tensor)"""
        assert self.value_dtype in (torch.int32, torch.int64)
        count = torch.zeros([1], dtype=self.key_dtype, device=self.keys_data.device)
        ws = self._ws()
        _lib.check(self._L.spx_hash_assign_arange(*self._args(), count.data_ptr(), ws.data_ptr(),
                                                  ws.numel(), self._stream()))
        return count

    @_on_table_device
    def items(self, max_size: int = -1):
        """-> (keys, values, count): the first `count` entries are the table's content"""
        if max_size == -1:
            max_size = self.values_data.shape[0]
        dev = self.keys_data.device
        keys = torch.empty([max_size], dtype=self.key_dtype, device=dev)
        values = torch.empty([max_size], dtype=self.value_dtype, device=dev)
        count = torch.zeros([1], dtype=self.key_dtype, device=dev)
        ws = self._ws()
        _lib.check(self._L.spx_hash_items(*self._args(), keys.data_ptr(), values.data_ptr(), max_size,
                                          count.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
        return keys, values, count
