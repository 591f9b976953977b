"""ctypes binding of the C ABI declared in ``include/spconv_amd.h``.

This is the only place the Python layer touches native code.  It mirrors the
role of ``spconv/pytorch/cppcore.py`` in the reference (raw pointer + stream
hand-off, ``cppcore.py:65-109``) without any tensorview / pybind types.

The library is built in-tree by ``spconv_amd/csrc/build.sh`` (hipcc, gfx950).
There is no CPU fallback: if the shared object is missing the import of any
compute entry point raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPX_LIB selects another build of the same ABI (the -DSPX_TIMELINE debug build of tools/timeline.py)
LIB_PATH = os.environ.get("SPX_LIB") or os.path.join(_HERE, "lib", "libspconv_amd.so")

c_int_p = ctypes.POINTER(ctypes.c_int)
vp = ctypes.c_void_p

# Every exported symbol of include/spconv_amd.h with (restype, argtypes).
SIGNATURES = {
    "spx_last_error": (ctypes.c_char_p, []),
    "spx_version": (ctypes.c_int, []),
    "spx_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "spx_conv_out_shape": (ctypes.c_int, [ctypes.c_int] + [c_int_p] * 6 + [ctypes.c_int, c_int_p]),
    "spx_subm_rulebook_ws_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "spx_subm_rulebook": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_int_p,
                                         c_int_p, c_int_p, vp, vp, vp, vp, vp, vp,
                                         ctypes.c_size_t, vp]),
    "spx_conv_rulebook_ws_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, c_int_p, c_int_p,
                                                     c_int_p, ctypes.c_int]),
    "spx_conv_rulebook_count": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
                                + [c_int_p] * 6 + [ctypes.c_int, vp, ctypes.c_size_t, c_int_p, vp]),
    "spx_conv_rulebook_fill": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
                               + [c_int_p] * 6 + [ctypes.c_int, ctypes.c_int] + [vp] * 7
                               + [vp, ctypes.c_size_t, vp]),
    "spx_conv_rulebook_static": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
                                 + [c_int_p] * 6 + [ctypes.c_int, ctypes.c_int] + [vp] * 8
                                 + [vp, ctypes.c_size_t, vp]),
    "spx_rankmap_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, c_int_p]),
    "spx_conv_sorted_ok": (ctypes.c_int, [ctypes.c_int, ctypes.c_int] + [c_int_p] * 6 + [ctypes.c_int]),
    "spx_conv_rulebook_sorted_ws_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_int_p, c_int_p]),
    "spx_conv_rulebook_count_sorted": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
                                       + [c_int_p] * 6 + [vp, ctypes.c_size_t, vp, ctypes.c_size_t, c_int_p, vp]),
    "spx_conv_rulebook_fill_sorted": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
                                      + [c_int_p] * 6 + [ctypes.c_int] + [vp] * 7
                                      + [vp, ctypes.c_size_t, vp, ctypes.c_size_t, vp]),
    "spx_conv_rulebook_static_sorted": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
                                        + [c_int_p] * 6 + [ctypes.c_int] + [vp] * 8
                                        + [vp, ctypes.c_size_t, vp, ctypes.c_size_t, vp]),
    "spx_subm_rulebook_ranked_ws_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "spx_subm_rulebook_ranked": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_int_p,
                                                c_int_p, c_int_p, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp,
                                                ctypes.c_size_t, vp]),
    "spx_subm_layout_mcap": (ctypes.c_size_t, [ctypes.c_int]),
    "spx_subm_layout_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "spx_subm_layout_ws_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "spx_subm_layout": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_size_t, vp]),
    "spx_igemm_bwd_rows_ws_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "spx_igemm_bwd_rows": (ctypes.c_int, [vp] * 7 + [ctypes.c_int] * 7 + [vp, ctypes.c_size_t, vp]),
    "spx_batchnorm_ws_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "spx_batchnorm_fwd": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                         vp, vp, vp, ctypes.c_size_t, vp, vp]),
    "spx_batchnorm_fwd_stats": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp,
                                               ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, vp, vp,
                                               vp, ctypes.c_int, vp, vp]),
    "spx_batchnorm_bwd": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int,
                                         vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_size_t, vp, vp]),
    "spx_mask_argsort_ws_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "spx_mask_argsort": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_size_t, vp]),
    "spx_mask_argsort_kv": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_size_t, vp]),
    "spx_native_to_table": (ctypes.c_int, [vp, vp] + [ctypes.c_int] * 5 + [vp, vp, vp]),
    "spx_table_to_native_ws_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "spx_table_to_native": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp,
                                           ctypes.c_size_t, vp]),
    "spx_igemm_acc_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "spx_igemm_fwd": (ctypes.c_int, [vp] * 6 + [ctypes.c_int] * 8 + [vp, ctypes.c_int, ctypes.c_float, vp,
                                                                     ctypes.c_size_t, vp]),
    "spx_igemm_fwd_stats_slots": (ctypes.c_int, [ctypes.c_int]),
    "spx_igemm_fwd_stats": (ctypes.c_int, [vp] * 6 + [ctypes.c_int] * 8 + [vp, ctypes.c_int, ctypes.c_float, vp,
                                                                           ctypes.c_size_t, vp, ctypes.c_int, vp,
                                                                           c_int_p, vp]),
    "spx_igemm_fwd_int8": (ctypes.c_int, [vp] * 6 + [ctypes.c_int] * 6 + [vp, vp, vp, ctypes.c_float,
                                                                       ctypes.c_int, ctypes.c_int,
                                                                       ctypes.c_float, vp]),
    "spx_igemm_dgrad_ws_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "spx_igemm_dgrad": (ctypes.c_int, [vp] * 6 + [ctypes.c_int] * 8 + [vp, ctypes.c_size_t, vp]),
    "spx_igemm_wgrad_ws_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "spx_wgrad_plan_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "spx_wgrad_plan": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "spx_igemm_wgrad": (ctypes.c_int, [vp] * 6 + [ctypes.c_int] * 7 + [vp, ctypes.c_size_t, vp]),
    "spx_igemm_bwd": (ctypes.c_int, [vp] * 8 + [ctypes.c_int] + [vp] * 3 + [ctypes.c_int] * 7
                      + [vp, ctypes.c_size_t, vp]),
    "spx_igemm_wgrad_deferred": (ctypes.c_int, [vp] * 6 + [ctypes.c_int] * 7 + [vp, ctypes.c_size_t, vp, vp]),
    "spx_igemm_bwd_deferred": (ctypes.c_int, [vp] * 8 + [ctypes.c_int] + [vp] * 3 + [ctypes.c_int] * 7
                               + [vp, ctypes.c_size_t, vp, vp]),
    "spx_wgrad_stage2_batch": (ctypes.c_int, [vp, ctypes.c_int, vp]),
    "spx_stage2_job_retarget": (ctypes.c_int, [vp, vp]),
    "spx_permute_tables": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp]),
    "spx_maxpool_fwd": (ctypes.c_int, [vp] * 4 + [ctypes.c_int] * 5 + [vp]),
    "spx_maxpool_bwd": (ctypes.c_int, [vp] * 6 + [ctypes.c_int] * 4 + [vp]),
    "spx_avgpool_fwd": (ctypes.c_int, [vp] * 5 + [ctypes.c_int] * 4 + [vp]),
    "spx_avgpool_bwd": (ctypes.c_int, [vp] * 5 + [ctypes.c_int] * 4 + [vp]),
    "spx_point2voxel_ws_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "spx_point2voxel": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                       c_int_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       vp, vp, vp, vp, c_int_p, vp, ctypes.c_size_t, vp]),
    "spx_hash_ws_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "spx_hash_clear": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp]),
    "spx_hash_insert": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int, vp]),
    "spx_hash_query": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp]),
    "spx_hash_insert_exist": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp,
                                             ctypes.c_int, vp]),
    "spx_hash_assign_arange": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp,
                                              ctypes.c_size_t, vp]),
    "spx_hash_items": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int, vp,
                                      vp, ctypes.c_size_t, vp]),
    "spx_rankmap_from_sorted": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_int_p, vp,
                                               ctypes.c_size_t, vp, vp]),
    "spx_key_argsort_ws_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "spx_key_argsort": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_int_p, vp, vp, vp,
                                       ctypes.c_size_t, vp, vp, vp, ctypes.c_int, vp, ctypes.c_size_t, vp]),
    "spx_pad_rows": (ctypes.c_int, [vp, vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, vp]),
    "spx_launch_count": (ctypes.c_longlong, [ctypes.c_char_p]),
    "spx_bias_act_inplace": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_float, vp]),
}

DTYPE_F32, DTYPE_F16, DTYPE_BF16, DTYPE_I8 = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_LEAKY_RELU = 0, 1, 2, 3

_lib: Optional[ctypes.CDLL] = None


def linked_objects() -> list:
    """Object files the library is linked from: csrc/build.sh holds the one list of translation units."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    out = subprocess.check_output(["bash", script, "--list"], text=True)
    return [os.path.join(_HERE, "lib", line.strip()) for line in out.splitlines() if line.strip()]


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 (no GPU needed).  force=True deletes EVERY object the library is linked from
    (and the library) first -- the list is build.sh's own, so a translation unit added there cannot be forgotten here."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    subprocess.check_call(["bash", script] + (["--force"] if force else []))
    return LIB_PATH


def load() -> ctypes.CDLL:
    """Load libspconv_amd.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"spconv_amd native library not found at {LIB_PATH}. Build it with "
            f"`bash spconv_amd/csrc/build.sh` (hipcc --offload-arch=gfx950). "
            f"There is no CPU fallback.")
    # torch bundles its own libamdhip64 (same SONAME); importing it first makes
    # our library bind to the runtime that owns torch's streams and allocations.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        msg = load().spx_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"spconv_amd native error ({status}): {msg}")


def ints(values):
    return (ctypes.c_int * len(values))(*[int(v) for v in values])
