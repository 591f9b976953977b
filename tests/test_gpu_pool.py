"""GPU parity of sparse pooling against the numpy restatement of
the reference kernels (oracle.maxpool_ref etc., following spconv/csrc/sparse/maxpool.py) and,
at module level, against dense torch pooling where the two are comparable."""
import numpy as np
import pytest
import torch
from torch import nn

import oracle
from util import dense_scene, gpu_rulebook, oracle_rulebook, rel_err, scene, to_np

pytestmark = pytest.mark.gpu

POOL_CASES = [
    # shape, n, bs, C, ksize, stride, pad, dil, subm
    ([24, 24, 24], 2500, 2, 64, [2] * 3, [2] * 3, [0] * 3, [1] * 3, False),
    ([24, 24, 24], 2500, 1, 32, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False),
    ([24, 24, 24], 2000, 1, 20, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True),     # odd channel count
    ([30, 30], 600, 2, 16, [3, 3], [2, 2], [1, 1], [1, 1], False),
]


def _np_dtype(dtype):
    return {torch.float32: np.float32, torch.float16: np.float16}[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape,n,bs,C,ksize,stride,pad,dil,subm", POOL_CASES)
def test_max_and_avg_pool_vs_oracle(cuda, shape, n, bs, C, ksize, stride, pad, dil, subm, dtype):
    from spconv_amd.pytorch import ops
    rng = np.random.default_rng(2)
    idx = dense_scene(shape, n, bs, 2)
    ref = oracle_rulebook(idx, bs, shape, ksize, stride, pad, dil, subm)
    rb, _ = gpu_rulebook(idx, bs, shape, ksize, stride, pad, dil, subm, need_bwd_table=True)
    npd = _np_dtype(dtype)
    # few distinct values -> ties between inputs of one window are exercised by the backward
    f = (rng.integers(-8, 9, (idx.shape[0], C)) / 4.0).astype(npd)
    dout = rng.uniform(-1, 1, (ref["n_out"], C)).astype(npd)
    fg, dg = torch.from_numpy(f).to(cuda), torch.from_numpy(dout).to(cuda)
    pf = ops.attach_rulebook(rb.pair_fwd, rb)
    pb = ops.attach_rulebook(rb.pair_bwd, rb)
    # max
    out = ops.indice_maxpool_implicit_gemm(fg, pf, rb.n_out)
    want = oracle.maxpool_ref(f, ref["pair"], ref["num"], ref["n_out"], subm)
    np.testing.assert_array_equal(to_np(out), want)
    din = ops.indice_maxpool_implicit_gemm_backward(fg, out, dg, pb)
    want_din = oracle.maxpool_bwd_ref(f, want, dout, ref["pair"], ref["num"], subm)
    assert rel_err(to_np(din.float()), want_din.astype(np.float32)) < (1e-6 if dtype == torch.float32 else 2e-3)
    # Native flavour: zero-initialised output (reference quirk kept)
    out0 = ops.indice_maxpool(fg, ops.attach_rulebook(rb.pair_native, rb), rb.num_per_loc, rb.n_out)
    np.testing.assert_array_equal(to_np(out0), oracle.maxpool_ref(f, ref["pair"], ref["num"], ref["n_out"],
                                                                  subm, init_zero=True))
    # avg
    avg, cnt = ops.indice_avgpool_implicit_gemm(fg, pf, rb.n_out, True)
    want_avg, want_cnt = oracle.avgpool_ref(f, ref["pair"], ref["num"], ref["n_out"], subm)
    np.testing.assert_array_equal(to_np(cnt), want_cnt)
    tol = 1e-6 if dtype == torch.float32 else 2e-3
    assert rel_err(to_np(avg.float()), want_avg.astype(np.float32)) < tol
    dav = ops.indice_avgpool_implicit_gemm_backward(dg, pb, cnt)
    want_dav = oracle.avgpool_bwd_ref(dout, want_cnt, ref["pair"], ref["num"], ref["n_in"], subm)
    assert rel_err(to_np(dav.float()), want_dav.astype(np.float32)) < tol


def test_int8_maxpool(cuda):
    from spconv_amd.pytorch import ops
    shape, n, C = [20, 20, 20], 1500, 32
    idx = dense_scene(shape, n, 1, 4)
    ref = oracle_rulebook(idx, 1, shape, [2] * 3, [2] * 3, [0] * 3, [1] * 3, False)
    rb, _ = gpu_rulebook(idx, 1, shape, [2] * 3, [2] * 3, [0] * 3, [1] * 3, False)
    f = np.random.default_rng(4).integers(-128, 128, (idx.shape[0], C), dtype=np.int8)
    out = ops.indice_maxpool_implicit_gemm(torch.from_numpy(f).to(cuda), ops.attach_rulebook(rb.pair_fwd, rb), rb.n_out)
    np.testing.assert_array_equal(to_np(out), oracle.maxpool_ref(f, ref["pair"], ref["num"], ref["n_out"]))


def test_maxpool_module_matches_dense_and_trains(cuda):
    """SparseMaxPool3d(2, 2) on non-negative features equals dense max_pool3d at the active output
    sites (empty sites are 0 in the dense tensor); gradients flow to the arg-max inputs."""
    import spconv_amd.pytorch as spconv
    torch.manual_seed(0)
    shape, bs, C = [16, 16, 16], 2, 16
    idx = scene(shape, 900, bs, 6)
    f = torch.rand(idx.shape[0], C) + 0.1
    x = spconv.SparseConvTensor(f.to(cuda).requires_grad_(True), torch.from_numpy(idx).to(cuda), shape, bs)
    pool = spconv.SparseMaxPool3d(2, 2, indice_key="p")
    y = pool(x)
    dense_in = x.dense()                                         # [B, C, D, H, W]
    want = nn.functional.max_pool3d(dense_in, 2, 2)
    got = y.dense()
    assert y.spatial_shape == [8, 8, 8] and "p" in y.indice_dict
    assert torch.allclose(got, want, atol=0, rtol=0)
    y.features.sum().backward()
    g = x.features.grad
    assert g is not None and torch.all((g == 0) | (g >= 1))        # each arg-max input gets its outputs' ones
    assert abs(float(g.sum()) - y.features.numel()) < 1e-3


def test_avgpool_module_and_global_pools(cuda):
    import spconv_amd.pytorch as spconv
    shape, bs, C = [12, 12, 12], 2, 8
    idx = scene(shape, 500, bs, 8)
    f = torch.randn(idx.shape[0], C)
    x = spconv.SparseConvTensor(f.to(cuda).requires_grad_(True), torch.from_numpy(idx).to(cuda), shape, bs)
    y = spconv.SparseAvgPool3d(3, 2, 1)(x)
    y.features.square().sum().backward()
    assert torch.isfinite(x.features.grad).all() and y.features.shape[1] == C
    gm, ga = spconv.SparseGlobalMaxPool()(x), spconv.SparseGlobalAvgPool()(x)
    for b in range(bs):
        rows = torch.from_numpy(idx[:, 0] == b)
        assert torch.allclose(gm[b].cpu(), f[rows].max(dim=0)[0])
        assert torch.allclose(ga[b].cpu(), f[rows].mean(dim=0), atol=1e-6)


def _golden(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))


def test_native_maxpool_equals_the_reference_code_executed(cuda):
    """tests/golden/pool_ref.npz: what the reference's own CPU loops (IndiceMaxPoolCPU, maxpool.py:590-703, driven as
    pytorch/ops.py:1899-1975 drives them) returned -- forward, backward and global_pool_rearrange, bit for bit."""
    from spconv_amd.pytorch import ops
    d = _golden("pool_ref.npz")
    idx, shape = d["indices"], [int(v) for v in d["shape"]]
    rb, _ = gpu_rulebook(idx, 2, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False, need_bwd_table=True)
    np.testing.assert_array_equal(to_np(rb.pair_native), d["pair"])
    np.testing.assert_array_equal(to_np(rb.num_per_loc), d["num"])
    f, dout = torch.from_numpy(d["features"]).to(cuda), torch.from_numpy(d["dout"]).to(cuda)
    native = ops.attach_rulebook(rb.pair_native, rb)
    out = ops.indice_maxpool(f, native, rb.num_per_loc, rb.n_out)
    np.testing.assert_array_equal(to_np(out), d["out"])
    din = ops.indice_maxpool_backward(f, out, dout, native, rb.num_per_loc)
    np.testing.assert_array_equal(to_np(din), d["din"])
    gp_out, gp_cnt = ops.global_pool_rearrange(torch.from_numpy(d["gp_coords"]).to(cuda), 2)
    np.testing.assert_array_equal(to_np(gp_cnt), d["gp_counts"])
    for b in range(2):          # (the reference leaves the tail of a row uninitialised: compare the defined part)
        c = int(d["gp_counts"][b])
        np.testing.assert_array_equal(to_np(gp_out)[b, :c], d["gp_out"][b, :c])


def test_reference_quirks_switch_avgpool_backward(cuda):
    """SPCONV_AMD_REFERENCE_QUIRKS=1: the average-pool backward multiplies by the window count, as the reference
    kernel does (maxpool.py:262-300); the default divides."""
    from spconv_amd import constants
    from spconv_amd.pytorch import ops
    shape, bs, C = [24, 24, 24], 1, 16
    idx = dense_scene(shape, 2500, bs, 3)
    ref = oracle_rulebook(idx, bs, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False)
    rb, _ = gpu_rulebook(idx, bs, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False, need_bwd_table=True)
    rng = np.random.default_rng(9)
    f = (rng.integers(-8, 9, (idx.shape[0], C)) / 4.0).astype(np.float32)
    dout = (rng.integers(-16, 17, (ref["n_out"], C)) / 8.0).astype(np.float32)
    fg, dg = torch.from_numpy(f).to(cuda), torch.from_numpy(dout).to(cuda)
    _, cnt = ops.indice_avgpool_implicit_gemm(fg, ops.attach_rulebook(rb.pair_fwd, rb), rb.n_out, True)
    pb = ops.attach_rulebook(rb.pair_bwd, rb)
    # the Python constant is the single source of truth (it travels to the library with the call)
    saved = constants.REFERENCE_QUIRKS
    try:
        constants.REFERENCE_QUIRKS = True
        quirk = to_np(ops.indice_avgpool_implicit_gemm_backward(dg, pb, cnt))
    finally:
        constants.REFERENCE_QUIRKS = saved
    plain = to_np(ops.indice_avgpool_implicit_gemm_backward(dg, pb, cnt))
    want_q = oracle.avgpool_bwd_ref(dout, to_np(cnt), ref["pair"], ref["num"], ref["n_in"], False, reference_quirks=True)
    want_p = oracle.avgpool_bwd_ref(dout, to_np(cnt), ref["pair"], ref["num"], ref["n_in"], False)
    np.testing.assert_array_equal(quirk, want_q)                 # dyadic data: products and sums are exact
    assert rel_err(plain, want_p) < 1e-6 and not np.array_equal(quirk, plain)
