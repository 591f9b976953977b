"""CPU: the C-ABI library loads and exports every symbol include/spconv_amd.h declares;
host-only entry points behave (no kernel is launched in this file)."""
import ctypes
import os
import re
import subprocess

import pytest

from spconv_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "spconv_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spx_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_header_and_binding_agree():
    assert _declared() == sorted(_lib.SIGNATURES)


def test_every_declared_symbol_is_exported(lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in _declared() if s not in exported]
    assert not missing, missing
    for s in _declared():
        assert getattr(lib, s) is not None


def test_no_torch_or_oracle_dependency():
    """The boundary is a plain C ABI: the shared object links the HIP runtime only."""
    out = subprocess.check_output(["readelf", "-d", _lib.LIB_PATH], text=True)
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    assert any("amdhip64" in n for n in needed)
    assert not any(("torch" in n) or ("c10" in n) or ("oracle" in n) for n in needed), needed


def test_version_and_error_channel(lib):
    assert lib.spx_version() >= 1000
    out = (ctypes.c_int * 3)()
    rc = lib.spx_conv_out_shape(7, _lib.ints([1] * 3), _lib.ints([1] * 3), _lib.ints([1] * 3),
                                _lib.ints([0] * 3), _lib.ints([1] * 3), _lib.ints([0] * 3), 0, out)
    assert rc != 0 and b"ndim" in lib.spx_last_error()
    with pytest.raises(RuntimeError, match="ndim"):
        _lib.check(rc)


def test_conv_out_shape_matches_oracle(lib):
    import oracle
    cases = [([41, 1600, 1408], [3] * 3, [2] * 3, [1] * 3, [1] * 3, [0] * 3, 0),
             ([21, 800, 704], [3, 1, 1], [2, 1, 1], [0] * 3, [1] * 3, [0] * 3, 0),
             ([19, 18, 17], [3] * 3, [1] * 3, [0] * 3, [3] * 3, [0] * 3, 0),
             ([10, 9, 9], [3] * 3, [2] * 3, [1] * 3, [1] * 3, [1] * 3, 1),
             ([2, 2, 2], [3] * 3, [2] * 3, [0] * 3, [1] * 3, [0] * 3, 0)]
    for shape, k, s, p, d, op, tr in cases:
        out = (ctypes.c_int * 3)()
        assert lib.spx_conv_out_shape(3, _lib.ints(shape), _lib.ints(k), _lib.ints(s), _lib.ints(p),
                                      _lib.ints(d), _lib.ints(op), tr, out) == 0
        assert list(out) == oracle.conv_out_shape(shape, k, s, p, d, op, bool(tr))


def test_workspace_size_queries(lib):
    assert lib.spx_subm_rulebook_ws_bytes(100_000, 27) > 100_000 * 2 * 12
    assert lib.spx_conv_rulebook_ws_bytes(100_000, 3, _lib.ints([3] * 3), _lib.ints([2] * 3), _lib.ints([1] * 3), 0) > 0
    assert lib.spx_igemm_dgrad_ws_bytes(64, 64, 27, _lib.DTYPE_F16) == 0   # transpose is in-kernel
    assert lib.spx_wgrad_plan_bytes(100_000, 27) > 27 * 8
    assert lib.spx_igemm_wgrad_ws_bytes(100_000, 64, 64, 27) % 256 == 0
    assert lib.spx_mask_argsort_ws_bytes(100_000) > 0
    assert lib.spx_table_to_native_ws_bytes(100_000, 27) > 0


def test_sorted_order_host_queries(lib):
    """Rank-map sizing and the geometries that take the sorted-order builder (host-only entry points)."""
    I = _lib.ints
    # one {bits, prefix} pair (8 bytes) per 32 cells of batch x grid + one count per 2048 words, both 256-byte aligned
    al = lambda b: (b + 255) // 256 * 256
    W = (4 * 21 * 800 * 704 + 31) // 32
    assert lib.spx_rankmap_bytes(3, 4, I([21, 800, 704])) == al(W * 8) + al((W + 2047) // 2048 * 4)
    assert lib.spx_rankmap_bytes(2, 1, I([5, 7])) == al(2 * 8) + al(4)
    assert lib.spx_rankmap_bytes(3, 200, I([21, 800, 704])) == 0          # key space beyond 2^31 cells: hash builder
    assert lib.spx_rankmap_bytes(3, 0, I([8, 8, 8])) == 0 and lib.spx_rankmap_bytes(3, 1, I([8, 0, 8])) == 0
    ok = lambda shape, k, s, p, tr=0, bs=2: lib.spx_conv_sorted_ok(
        3, bs, I(shape), I([(shape[i] + 2 * p[i] - (k[i] - 1) - 1) // s[i] + 1 for i in range(3)]), I(k), I(s), I(p),
        I([1] * 3), tr)
    assert ok([41, 1600, 1408], [3] * 3, [2] * 3, [1] * 3) == 1              # k3 s2
    assert ok([41, 1600, 1408], [2] * 3, [2] * 3, [0] * 3) == 1              # k2 s2
    assert ok([41, 1600, 1408], [3] * 3, [3] * 3, [1] * 3) == 1
    assert ok([41, 1600, 1408], [3] * 3, [1] * 3, [1] * 3) == 0              # stride 1: every offset is a candidate
    assert ok([5, 200, 176], [3, 1, 1], [2, 1, 1], [0] * 3) == 0             # two of three offsets
    assert ok([41, 1600, 1408], [3] * 3, [2] * 3, [1] * 3, tr=1) == 0        # transposed
    assert ok([41, 1600, 1408], [3] * 3, [2] * 3, [1] * 3, bs=400) == 0      # key space too large
    ws = lib.spx_conv_rulebook_sorted_ws_bytes(400_000, 3, 4, I([21, 800, 704]), I([3] * 3))
    cells = 4 * 21 * 800 * 704
    assert cells < ws < cells + (1 << 20)      # one byte per cell between the mark and the prefix pass + counts: no table
    assert lib.spx_subm_rulebook_ranked_ws_bytes(313_000, 27) < lib.spx_subm_rulebook_ws_bytes(313_000, 27) // 8


def test_rank_map_follows_its_index_tensor_only():
    """ops._rankmap_of: the map a sorted-order build left on an index tensor is used for exactly that level."""
    import torch
    from spconv_amd.pytorch import ops
    ind = torch.zeros((10, 4), dtype=torch.int32)
    cells = torch.zeros((8,), dtype=torch.int32)
    assert ops._rankmap_of(ind, 2, [4, 4, 4], 10, 27) is None
    ind._spx_rankmap = (cells, 2, (4, 4, 4), 10)
    assert ops._rankmap_of(ind, 2, [4, 4, 4], 10, 27) is cells
    assert ops._rankmap_of(ind, 1, [4, 4, 4], 10, 27) is None          # other batch size
    assert ops._rankmap_of(ind, 2, [4, 4, 5], 10, 27) is None          # other grid
    assert ops._rankmap_of(ind, 2, [4, 4, 4], 9, 27) is None           # other row count
    assert ops._rankmap_of(ind, 2, [4, 4, 4], 10, 1) is None           # 1 x 1 x 1: nothing to look up
    assert ops._rankmap_of(ind.clone(), 2, [4, 4, 4], 10, 27) is None  # a copy does not carry it


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libspconv_amd.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_force_build_rebuilds_every_linked_object():
    """`_lib.build(force=True)` (what `__graft_entry__.build()` runs) leaves no object older than the call: the list of
    translation units is csrc/build.sh's own (`--list`), every HIP / C++ source of csrc/ is in it, and nothing else
    sits in lib/ (four of twelve objects survived a 'forced' build because a second list had gone stale)."""
    import time
    objs = _lib.linked_objects()
    units = {os.path.splitext(os.path.basename(o))[0] for o in objs}
    csrc = os.path.join(ROOT, "spconv_amd", "csrc")
    sources = {os.path.splitext(f)[0] for f in os.listdir(csrc) if f.endswith((".hip", ".cpp"))}
    assert units == sources, (sorted(units ^ sources))
    t0 = time.time() - 1.0
    _lib.build(force=True)
    stale = [o for o in objs + [_lib.LIB_PATH] if not os.path.exists(o) or os.path.getmtime(o) < t0]
    assert not stale, stale
    present = {f for f in os.listdir(os.path.join(ROOT, "spconv_amd", "lib")) if f.endswith(".o")}
    assert present == {os.path.basename(o) for o in objs}, present
