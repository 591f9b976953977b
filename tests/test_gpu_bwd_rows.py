"""GPU: the backward of a narrow layer from one gather per pair (csrc/igemm_bwdn.hip, spx_igemm_bwd_rows)
against the pair-list kernels it replaces for C, K in {16, 32} (spx_igemm_bwd) and against the CPU oracle
(ops.py:1164-1253 restated): input gradient and weight gradient, SubM (mirrored weight order) and strided
convolution (pair_bwd table), every channel combination, row counts that are not multiples of the 128-row tile,
dead tile rows, fp16 / bf16, dW-only calls, bit-reproducibility."""
import numpy as np
import pytest
import torch

import oracle
from util import dense_scene, gpu_rulebook, rel_err, scene

pytestmark = pytest.mark.gpu
TOL = {torch.float16: 2e-3, torch.bfloat16: 1.2e-2}


def _tensors(rng, n_in, n_out, C, K, dtype, dev):
    f = torch.from_numpy(rng.uniform(-1, 1, (n_in, C)).astype(np.float32)).to(dev, dtype)
    w = torch.from_numpy(rng.uniform(-1, 1, (K, 3, 3, 3, C)).astype(np.float32)).to(dev, dtype)
    d = torch.from_numpy(rng.uniform(-0.2, 0.2, (n_out, K)).astype(np.float32)).to(dev, dtype)
    return f, w, d


def _both(fn):
    from spconv_amd.pytorch import ops
    old = ops._BWD_ROWS
    try:
        ops._BWD_ROWS = False
        a = fn()
        ops._BWD_ROWS = True
        b = fn()
    finally:
        ops._BWD_ROWS = old
    torch.cuda.synchronize()
    return a, b


@pytest.mark.parametrize("C,K", [(16, 16), (32, 32), (16, 32), (32, 16)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,n,dense", [([40, 200, 200], 20_001, False), ([24, 24, 24], 3_001, True),
                                           ([10, 10, 10], 77, True)])
def test_subm_backward_rows_vs_pair_lists_and_oracle(cuda, C, K, dtype, shape, n, dense):
    from spconv_amd.pytorch import ops
    idx = dense_scene(shape, n, 1, 3) if dense else scene(shape, n, 1, 3)
    rb, _ = gpu_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    rng = np.random.default_rng(C + K)
    f, w, d = _tensors(rng, rb.n_in, rb.n_out, C, K, dtype, cuda)
    plan = ops._plan_of(rb)
    run = lambda: ops.igemm_bwd(f, d, w, rb.pair_fwd, rb.mask_fwd, None, rb.pair_native, rb.num_per_loc, True, plan)
    (din_a, dw_a), (din_b, dw_b) = _both(run)
    tol = TOL[dtype]
    assert rel_err(din_b.float().cpu().numpy(), din_a.float().cpu().numpy()) <= tol
    assert rel_err(dw_b.float().cpu().numpy(), dw_a.float().cpu().numpy()) <= tol
    _, pair, num, _ = oracle.get_indice_pairs(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, subm=True)
    din_ref, dw_ref = oracle.indice_conv_backward(f.float().cpu(), w.float().cpu(), d.float().cpu(), pair, num, subm=True)
    assert rel_err(din_b.float().cpu().numpy(), din_ref.numpy()) <= tol
    assert rel_err(dw_b.float().cpu().numpy(), dw_ref.numpy()) <= tol


@pytest.mark.parametrize("C,K", [(16, 32), (32, 32)])
def test_strided_conv_backward_rows(cuda, C, K):
    """regular convolution: the dgrad table is pair_bwd over the INPUT rows, no mirrored weight order, n_out != n_in"""
    from spconv_amd.pytorch import ops
    shape = [30, 30, 30]
    idx = dense_scene(shape, 6000, 2, 7)
    rb, _ = gpu_rulebook(idx, 2, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False)
    rng = np.random.default_rng(5)
    f, w, d = _tensors(rng, rb.n_in, rb.n_out, C, K, torch.float16, cuda)
    plan = ops._plan_of(rb)
    run = lambda: ops.igemm_bwd(f, d, w, rb.pair_bwd, rb.mask_bwd, None, rb.pair_native, rb.num_per_loc, False, plan)
    (din_a, dw_a), (din_b, dw_b) = _both(run)
    assert rel_err(din_b.float().cpu().numpy(), din_a.float().cpu().numpy()) <= 2e-3
    assert rel_err(dw_b.float().cpu().numpy(), dw_a.float().cpu().numpy()) <= 2e-3
    _, pair, num, _ = oracle.get_indice_pairs(idx, 2, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, subm=False)
    din_ref, dw_ref = oracle.indice_conv_backward(f.float().cpu(), w.float().cpu(), d.float().cpu(), pair, num, subm=False)
    assert rel_err(din_b.float().cpu().numpy(), din_ref.numpy()) <= 2e-3
    assert rel_err(dw_b.float().cpu().numpy(), dw_ref.numpy()) <= 2e-3


def test_dw_only_and_reproducible(cuda):
    from spconv_amd.pytorch import ops
    shape = [24, 24, 24]
    idx = dense_scene(shape, 3000, 1, 9)
    rb, _ = gpu_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    rng = np.random.default_rng(6)
    f, w, d = _tensors(rng, rb.n_in, rb.n_out, 32, 32, torch.float16, cuda)
    plan = ops._plan_of(rb)
    runs = [ops.igemm_bwd(f, d, w, rb.pair_fwd, rb.mask_fwd, None, rb.pair_native, rb.num_per_loc, True, plan)
            for _ in range(5)]
    none, dw_only = ops.igemm_bwd(f, d, w, rb.pair_fwd, rb.mask_fwd, None, rb.pair_native, rb.num_per_loc, True, plan,
                                  need_din=False)
    torch.cuda.synchronize()
    assert none is None and torch.equal(dw_only, runs[0][1])
    assert all(torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1]) for r in runs)


def test_module_training_step_takes_the_rows_kernel(cuda, monkeypatch):
    """SubMConv3d 32 -> 32 through autograd: the narrow-layer backward is the one that runs, gradients equal the
    pair-list kernels' within fp16 tolerance."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch import ops
    calls = []
    real = ops._igemm_bwd_rows
    monkeypatch.setattr(ops, "_igemm_bwd_rows", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    shape = [24, 24, 24]
    idx = torch.from_numpy(dense_scene(shape, 3000, 1, 4)).to(cuda)
    torch.manual_seed(0)
    net = spconv.SubMConv3d(32, 32, 3, bias=False).to(cuda).half().train()
    f0 = (torch.rand(idx.shape[0], 32, device=cuda) - 0.5).half()
    g = ((torch.rand(idx.shape[0], 32, device=cuda) - 0.5) * 0.2).half()

    def step():
        net.weight.grad = None
        f = f0.clone().requires_grad_(True)
        net(spconv.SparseConvTensor(f, idx, shape, 1)).features.backward(g)
        return f.grad.clone(), net.weight.grad.clone()
    (din_a, dw_a), (din_b, dw_b) = _both(step)
    assert calls, "the rows kernel did not run"
    assert rel_err(din_b.float().cpu().numpy(), din_a.float().cpu().numpy()) <= 2e-3
    assert rel_err(dw_b.float().cpu().numpy(), dw_a.float().cpu().numpy()) <= 2e-3


@pytest.mark.parametrize("ndim,ksize,stride,subm", [(2, 3, 1, True), (3, 2, 2, False), (2, 3, 2, False)])
def test_other_kernel_volumes(cuda, ndim, ksize, stride, subm):
    """kv = 9 (2-d SubM), kv = 8 (k2 s2 in 3-d: exactly one candidate per input), kv = 9 strided in 2-d"""
    from spconv_amd.pytorch import ops
    shape = [40] * ndim if ndim == 3 else [96, 96]
    rng = np.random.default_rng(ndim * 10 + ksize)
    vol = int(np.prod(shape))
    lin = rng.choice(vol, size=vol // 5, replace=False)
    coords = np.stack(np.unravel_index(lin, shape), -1).astype(np.int32)
    idx = np.concatenate([np.zeros((coords.shape[0], 1), np.int32), coords], 1)
    pad = [ksize // 2] * ndim if subm else ([1] * ndim if ksize == 3 else [0] * ndim)
    rb, _ = gpu_rulebook(idx, 1, shape, [ksize] * ndim, [stride] * ndim, pad, [1] * ndim, subm)
    C = K = 32
    f = torch.from_numpy(rng.uniform(-1, 1, (rb.n_in, C)).astype(np.float32)).to(cuda).half()
    w = torch.from_numpy(rng.uniform(-1, 1, (K, *([ksize] * ndim), C)).astype(np.float32)).to(cuda).half()
    d = torch.from_numpy(rng.uniform(-0.2, 0.2, (rb.n_out, K)).astype(np.float32)).to(cuda).half()
    table, mask = (rb.pair_fwd, rb.mask_fwd) if subm else (rb.pair_bwd, rb.mask_bwd)
    plan = ops._plan_of(rb)
    run = lambda: ops.igemm_bwd(f, d, w, table, mask, None, rb.pair_native, rb.num_per_loc, subm, plan)
    (din_a, dw_a), (din_b, dw_b) = _both(run)
    assert rel_err(din_b.float().cpu().numpy(), din_a.float().cpu().numpy()) <= 2e-3
    assert rel_err(dw_b.float().cpu().numpy(), dw_a.float().cpu().numpy()) <= 2e-3


def test_padded_rows_of_a_static_tensor(cuda):
    """dead rows (batch index -1, mask 0, zero features) in the table: no contribution, zero input gradient"""
    from spconv_amd.pytorch import ops
    shape = [24, 24, 24]
    idx = dense_scene(shape, 3000, 1, 2)
    n, n_static = idx.shape[0], idx.shape[0] + 700
    padded = np.concatenate([idx, np.full((n_static - n, 4), -1, np.int32)], 0)
    rb, _ = gpu_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    rbp, _ = gpu_rulebook(padded, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    rng = np.random.default_rng(8)
    f, w, d = _tensors(rng, n, n, 32, 32, torch.float16, cuda)
    fp = torch.zeros((n_static, 32), dtype=torch.float16, device=cuda)
    fp[:n] = f
    dp = torch.zeros((n_static, 32), dtype=torch.float16, device=cuda)
    dp[:n] = d
    old = ops._BWD_ROWS
    try:
        ops._BWD_ROWS = True
        din, dw = ops.igemm_bwd(f, d, w, rb.pair_fwd, rb.mask_fwd, None, rb.pair_native, rb.num_per_loc, True, ops._plan_of(rb))
        dinp, dwp = ops.igemm_bwd(fp, dp, w, rbp.pair_fwd, rbp.mask_fwd, None, rbp.pair_native, rbp.num_per_loc, True,
                                  ops._plan_of(rbp))
    finally:
        ops._BWD_ROWS = old
    torch.cuda.synchronize()
    assert torch.equal(dinp[:n], din) and not bool(dinp[n:].any())
    assert rel_err(dwp.float().cpu().numpy(), dw.float().cpu().numpy()) <= 1e-3      # (another tile partition: summation order)


def test_module_leaves_the_native_lists_out_when_the_rows_walk_runs(cuda):
    """A 32-channel SubM layer on a dense level takes the rows walk in its backward pass: its rulebook build leaves
    the ConvAlgo.Native lists (and the range plan) out, nothing derives them, and the gradients equal those of the
    pair-list backward over lists built the usual way."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch import ops
    shape = [16, 40, 40]
    idx = dense_scene(shape, 9000, 1, 4)
    n = idx.shape[0]
    assert n >= ops._BWD_ROWS_OCC * np.prod(shape)
    torch.manual_seed(2)
    net = spconv.SparseSequential(spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="r"),
                                  spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="r")).to(cuda).half()
    g = torch.Generator(device="cpu").manual_seed(5)
    f = (torch.rand((n, 32), generator=g) * 2 - 1).to(cuda).half()
    dout = ((torch.rand((n, 32), generator=g) * 2 - 1) * 0.2).to(cuda).half()

    def run():
        net.zero_grad(set_to_none=True)
        fe = f.clone().requires_grad_(True)
        y = net(spconv.SparseConvTensor(fe, torch.from_numpy(idx).to(cuda), shape, 1))
        y.features.backward(dout)
        torch.cuda.synchronize()
        return y.indice_dict["r"].rulebook, fe.grad.clone(), [p.grad.clone() for p in net.parameters()]
    rb, din, dws = run()
    assert not rb.has_native and rb.wgrad_plan is None, "the lists were built although nothing reads them"
    saved = ops._BWD_ROWS
    try:
        ops._BWD_ROWS = False                      # pair-list backward: lists built inside the rulebook build
        rb2, din2, dws2 = run()
    finally:
        ops._BWD_ROWS = saved
    assert rb2.has_native
    np.testing.assert_array_equal(rb.pair_fwd.cpu().numpy(), rb2.pair_fwd.cpu().numpy())
    assert float((din.float() - din2.float()).abs().max()) <= 2e-3 * float(din2.float().abs().max())
    for a, b in zip(dws, dws2):
        assert float((a.float() - b.float()).abs().max()) <= 3e-3 * float(b.float().abs().max())
    # asked for afterwards (reference-shaped API), the lists are derived from the table and equal the built ones
    np.testing.assert_array_equal(rb.pair_native.cpu().numpy(), rb2.pair_native.cpu().numpy())
    np.testing.assert_array_equal(rb.num_per_loc.cpu().numpy(), rb2.num_per_loc.cpu().numpy())
