"""GPU: the persistent loader / consumer gather-GEMM (csrc/igemm5.hip, SPX_GEMM_V = 5) against the
128-row direct-fragment kernels (v4) it replaces -- same accumulation order and MFMA sequence per
output element, so forward and dgrad must agree BIT FOR BIT -- and against the CPU oracle
(ops.py:888-988,1164-1253 restated).  Shapes: every channel width v5 is instantiated for, row counts
that are not multiples of 4 / 128 (the dword-sized tail of the pair-table DMA), sparse (uniform) and
dense neighbourhoods (more steps than ring stages), regular strided conv (no identity offset, both
table directions)."""
import numpy as np
import pytest
import torch

import oracle
from util import dense_scene, gpu_rulebook, rel_err, scene

pytestmark = pytest.mark.gpu


def _set_version(v):
    from spconv_amd import _lib
    _lib.check(_lib.load().spx_set_option(b"SPX_GEMM_V", int(v)))


@pytest.fixture(autouse=True)
def _restore_version():
    yield
    _set_version(4)


def _both(fn, new=5):
    _set_version(4)
    a = fn()
    _set_version(new)
    b = fn()
    torch.cuda.synchronize()
    return a, b


def _tensors(rng, n_in, n_out, C, K, kv_shape, dtype, dev):
    f = torch.from_numpy(rng.uniform(-1, 1, (n_in, C)).astype(np.float32)).to(dev, dtype)
    w = torch.from_numpy(rng.uniform(-1, 1, (K, *kv_shape, C)).astype(np.float32)).to(dev, dtype)
    d = torch.from_numpy(rng.uniform(-0.2, 0.2, (n_out, K)).astype(np.float32)).to(dev, dtype)
    return f, w, d


SUBM_CASES = [
    # shape, voxels, dense, C, K
    ([40, 200, 200], 20_001, False, 64, 64),      # uniform: ~1 step per tile beyond the identity, n % 4 = 1
    ([40, 200, 200], 5_003, False, 32, 32),
    ([24, 24, 24], 3_000, True, 64, 64),          # dense: 27 steps per tile, ring wraps many times
    ([24, 24, 24], 3_001, True, 16, 64),
    ([24, 24, 24], 2_999, True, 64, 16),
    ([30, 30, 30], 9_000, True, 32, 64),
    ([64, 64, 64], 70_000, False, 64, 64),        # more tiles than persistent workgroups
    ([10, 10, 10], 77, True, 64, 64),             # one partial tile
]


@pytest.mark.parametrize("shape,n,dense,C,K", SUBM_CASES)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_sparse_kernel_fwd_bit_identical_to_v4(cuda, shape, n, dense, C, K, dtype):
    """SPX_GEMM_V = 6: one autonomous wave per 32 rows (csrc/igemm_sp.hip); forward-type GEMMs only."""
    from spconv_amd.pytorch import ops
    idx = dense_scene(shape, n, 1, 3) if dense else scene(shape, n, 1, 3)
    rb, _ = gpu_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    rng = np.random.default_rng(1)
    f, w, d = _tensors(rng, rb.n_in, rb.n_out, C, K, [3] * 3, dtype, cuda)
    o4, o6 = _both(lambda: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13), new=6)
    assert torch.equal(o4, o6), float((o4.float() - o6.float()).abs().max())


def test_sparse_kernel_regular_conv_and_epilogue(cuda):
    from spconv_amd.pytorch import ops
    shape = [30, 30, 30]
    idx = dense_scene(shape, 6000, 2, 7)
    rb, _ = gpu_rulebook(idx, 2, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False)
    rng = np.random.default_rng(3)
    f, w, d = _tensors(rng, rb.n_in, rb.n_out, 32, 64, [3] * 3, torch.float16, cuda)
    bias = torch.from_numpy(rng.uniform(-1, 1, 64).astype(np.float32)).to(cuda, torch.float16)
    o4, o6 = _both(lambda: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, -1), new=6)
    assert torch.equal(o4, o6)
    o4, o6 = _both(lambda: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, -1, bias,
                                         ops.Activation.ReLU, 0.0), new=6)
    assert torch.equal(o4, o6)


@pytest.mark.parametrize("shape,n,dense,C,K", SUBM_CASES)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_subm_fwd_dgrad_bit_identical_to_v4(cuda, shape, n, dense, C, K, dtype):
    from spconv_amd.pytorch import ops
    idx = dense_scene(shape, n, 1, 3) if dense else scene(shape, n, 1, 3)
    rb, _ = gpu_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    rng = np.random.default_rng(1)
    f, w, d = _tensors(rng, rb.n_in, rb.n_out, C, K, [3] * 3, dtype, cuda)
    o4, o5 = _both(lambda: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13))
    assert torch.equal(o4, o5), float((o4.float() - o5.float()).abs().max())
    g4, g5 = _both(lambda: ops.igemm_dgrad(d, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_in, True))
    assert torch.equal(g4, g5), float((g4.float() - g5.float()).abs().max())


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.2e-2)])
def test_v5_vs_oracle(cuda, dtype, tol):
    from spconv_amd.pytorch import ops
    shape, C, K = [20, 20, 20], 64, 64
    idx = dense_scene(shape, 2500, 1, 5)
    rb, _ = gpu_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    rng = np.random.default_rng(2)
    f, w, d = _tensors(rng, rb.n_in, rb.n_out, C, K, [3] * 3, dtype, cuda)
    _set_version(5)
    out = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13)
    din = ops.igemm_dgrad(d, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_in, True)
    torch.cuda.synchronize()
    _, pair, num, _ = oracle.get_indice_pairs(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, subm=True)
    f32, w32, d32 = f.float().cpu(), w.float().cpu(), d.float().cpu()
    out_ref = oracle.indice_conv(f32, w32, pair, num, rb.n_out, subm=True)
    din_ref, _ = oracle.indice_conv_backward(f32, w32, d32, pair, num, subm=True)
    assert rel_err(out.float().cpu().numpy(), out_ref.numpy()) <= tol
    assert rel_err(din.float().cpu().numpy(), din_ref.numpy()) <= tol


@pytest.mark.parametrize("C,K", [(16, 32), (32, 64), (64, 64)])
def test_regular_conv_both_directions(cuda, C, K):
    """stride-2 SparseConv3d tables: no identity offset, n_out != n_in, dgrad over pair_bwd."""
    from spconv_amd.pytorch import ops
    shape = [30, 30, 30]
    idx = dense_scene(shape, 6000, 2, 7)
    rb, _ = gpu_rulebook(idx, 2, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False)
    rng = np.random.default_rng(3)
    f, w, d = _tensors(rng, rb.n_in, rb.n_out, C, K, [3] * 3, torch.float16, cuda)
    o4, o5 = _both(lambda: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, -1))
    assert torch.equal(o4, o5)
    g4, g5 = _both(lambda: ops.igemm_dgrad(d, w, rb.pair_bwd, rb.mask_bwd, None, rb.n_in, False))
    assert torch.equal(g4, g5)


def test_bias_and_activation_epilogue(cuda):
    from spconv_amd.pytorch import ops
    shape = [24, 24, 24]
    idx = dense_scene(shape, 3000, 1, 9)
    rb, _ = gpu_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    rng = np.random.default_rng(4)
    f, w, _ = _tensors(rng, rb.n_in, rb.n_out, 32, 64, [3] * 3, torch.float16, cuda)
    bias = torch.from_numpy(rng.uniform(-1, 1, 64).astype(np.float32)).to(cuda, torch.float16)
    for act in (ops.Activation.ReLU, ops.Activation.LeakyReLU, ops.Activation.Sigmoid):
        o4, o5 = _both(lambda: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13, bias, act, 0.1))
        assert torch.equal(o4, o5)


def test_repeated_launches_are_stable(cuda):
    """The loader / consumer hand-off is a race if the counted waits are wrong: many launches over a
    dense scene (every tile wraps the ring ~7 times), every result compared with the first."""
    from spconv_amd.pytorch import ops
    shape = [28, 28, 28]
    idx = dense_scene(shape, 9000, 1, 11)
    rb, _ = gpu_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    rng = np.random.default_rng(5)
    f, w, d = _tensors(rng, rb.n_in, rb.n_out, 64, 64, [3] * 3, torch.float16, cuda)
    _set_version(4)
    ref = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13)
    gref = ops.igemm_dgrad(d, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_in, True)
    _set_version(5)
    outs = [ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13) for _ in range(50)]
    gins = [ops.igemm_dgrad(d, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_in, True) for _ in range(50)]
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for o in outs)
    assert all(torch.equal(g, gref) for g in gins)
