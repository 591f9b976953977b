"""GPU parity of the convolution kernels (forward, dgrad, wgrad) against the CPU
oracle (restating spconv/pytorch/ops.py:888-988,1164-1253) and, at module level,
against dense torch conv3d -- the reference's own test oracle
(test/test_conv.py:286-357).

Tolerances: fp32 features 1e-3 relative (north_star); fp16 / bf16 compare against
the fp32 oracle evaluated on the SAME rounded inputs, so the only differences are
accumulation order and the final rounding of the 16-bit output
(2^-11 = 4.9e-4 for fp16, 2^-8 = 3.9e-3 for bf16, relative to the value)."""
import numpy as np
import pytest
import torch

import oracle
from util import assert_close_abs_sum, dense_scene, gpu_rulebook, oracle_rulebook, rel_err, scene, to_np

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-3, torch.float16: 2e-3, torch.bfloat16: 1.2e-2}


def _rounded(a: np.ndarray, dtype) -> torch.Tensor:
    """fp32 tensor holding values representable in `dtype`."""
    return torch.from_numpy(a).to(dtype).to(torch.float32)


def _case(shape, n, bs, C, K, ksize, stride, pad, dil, subm, dtype, seed=0, transposed=False,
          dense=True):
    rng = np.random.default_rng(seed)
    idx = dense_scene(shape, n, bs, seed) if dense else scene(shape, n, bs, seed)
    ref = oracle_rulebook(idx, bs, shape, ksize, stride, pad, dil, subm, transposed)
    f = _rounded(rng.uniform(-1, 1, (idx.shape[0], C)).astype(np.float32), dtype)
    w = _rounded(rng.uniform(-1, 1, (K, *ksize, C)).astype(np.float32), dtype)
    dout = _rounded(rng.uniform(-0.2, 0.2, (ref["n_out"], K)).astype(np.float32), dtype)
    return idx, ref, f, w, dout


def _run_gpu(cuda, idx, bs, shape, ksize, stride, pad, dil, subm, transposed, f, w, dout, dtype,
             use_sort=False, use_plan=True):
    from spconv_amd.pytorch import ops
    rb, _ = gpu_rulebook(idx, bs, shape, ksize, stride, pad, dil, subm, transposed,
                         do_sort=use_sort)
    fg, wg, dg = f.to(cuda, dtype), w.to(cuda, dtype), dout.to(cuda, dtype)
    kv = rb.kv
    out = ops.igemm_fwd(fg, wg, rb.pair_fwd, rb.mask_fwd, rb.argsort_fwd, rb.n_out,
                        kv // 2 if subm else -1)
    if subm:
        din = ops.igemm_dgrad(dg, wg, rb.pair_fwd, rb.mask_fwd, rb.argsort_fwd, rb.n_in, True)
    else:
        din = ops.igemm_dgrad(dg, wg, rb.pair_bwd, rb.mask_bwd, rb.argsort_bwd, rb.n_in, False)
    plan = ops._plan_of(rb) if use_plan else None
    dw = ops.igemm_wgrad(fg, dg, wg.shape, rb.pair_native, rb.num_per_loc, subm, plan)
    torch.cuda.synchronize()
    return rb, out.float().cpu(), din.float().cpu(), dw.float().cpu()


def _check(name, got, ref, tol):
    e = rel_err(got.numpy(), ref.numpy())
    assert e <= tol, f"{name}: rel err {e:.3e} > {tol:.1e}"


def _check_abs(got3, ref3, f, w, dout, ref, subm, dtype):
    """The sharp ELEMENT-wise bar (util.assert_close_abs_sum; VERDICT r4 weak 1a asked for it on the small-case sweep
    too): every element of out / din / dW within half an ulp of the output dtype of its own reference value + c x the
    same sum over operand magnitudes (the oracle run on |f|, |w|, |dout|).  c: 1e-6 for 16-bit tensors (fp32
    accumulation of exact products), 1e-5 for fp32 ones (the products are rounded too)."""
    oa = oracle.indice_conv(f.abs(), w.abs(), ref["pair"], ref["num"], ref["n_out"], subm=subm)
    dia, dwa = oracle.indice_conv_backward(f.abs(), w.abs(), dout.abs(), ref["pair"], ref["num"], subm=subm)
    c = 1e-5 if dtype == torch.float32 else 1e-6
    for name, g, r, a in zip(("out", "din", "dw"), got3, ref3, (oa, dia, dwa)):
        assert_close_abs_sum(g.numpy(), r.numpy(), a.numpy(), dtype, c, name=name)


CONV_CASES = [
    # shape, n, bs, C, K, ksize, stride, pad, dil, subm
    ([64, 64, 64], 5000, 1, 16, 16, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True),     # cfg 1
    ([24, 24, 24], 2500, 2, 64, 64, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True),     # cfg 2 channels
    ([24, 24, 24], 2500, 2, 32, 64, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True),
    ([24, 24, 24], 2500, 1, 128, 32, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True),
    ([24, 24, 24], 1500, 1, 64, 128, [3] * 3, [1] * 3, [2] * 3, [2] * 3, True),
    ([24, 24, 24], 2500, 2, 16, 32, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False),    # cfg 3 chain
    ([24, 24, 24], 2500, 2, 32, 64, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False),
    ([24, 24, 24], 2500, 1, 64, 128, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False),
    ([24, 24, 24], 2000, 1, 64, 64, [3, 1, 1], [2, 1, 1], [0] * 3, [1] * 3, False),
    ([24, 24, 24], 2000, 1, 64, 64, [2] * 3, [2] * 3, [0] * 3, [1] * 3, False),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,n,bs,C,K,ksize,stride,pad,dil,subm", CONV_CASES)
def test_conv_fwd_bwd_vs_oracle(cuda, shape, n, bs, C, K, ksize, stride, pad, dil, subm, dtype):
    idx, ref, f, w, dout = _case(shape, n, bs, C, K, ksize, stride, pad, dil, subm, dtype)
    out_ref = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"], subm=subm)
    din_ref, dw_ref = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"], subm=subm)
    rb, out, din, dw = _run_gpu(cuda, idx, bs, shape, ksize, stride, pad, dil, subm, False, f, w,
                                dout, dtype)
    tol = TOL[dtype]
    _check("out", out, out_ref, tol)
    _check("din", din, din_ref, tol)
    _check("dw", dw, dw_ref, tol)
    _check_abs((out, din, dw), (out_ref, din_ref, dw_ref), f, w, dout, ref, subm, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_odd_channels_use_generic_kernels(cuda, dtype):
    """C=5 (raw voxel features) and K=24 are outside the MFMA tile set."""
    shape = [20, 20, 20]
    for C, K in ((5, 16), (16, 24), (3, 7)):
        idx, ref, f, w, dout = _case(shape, 1200, 1, C, K, [3] * 3, [1] * 3, [1] * 3, [1] * 3,
                                     True, dtype)
        out_ref = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"], subm=True)
        din_ref, dw_ref = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"], subm=True)
        _, out, din, dw = _run_gpu(cuda, idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True,
                                   False, f, w, dout, dtype)
        _check("out", out, out_ref, TOL[dtype])
        _check("din", din, din_ref, TOL[dtype])
        _check("dw", dw, dw_ref, TOL[dtype])
        _check_abs((out, din, dw), (out_ref, din_ref, dw_ref), f, w, dout, ref, True, dtype)


def test_large_kernel_volume_two_mask_words(cuda):
    shape = [20, 20, 20]
    ksize = [5, 3, 3]
    idx, ref, f, w, dout = _case(shape, 1500, 1, 16, 16, ksize, [1] * 3, [2, 1, 1], [1] * 3, True,
                                 torch.float32)
    out_ref = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"], subm=True)
    din_ref, dw_ref = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"], subm=True)
    _, out, din, dw = _run_gpu(cuda, idx, 1, shape, ksize, [1] * 3, [2, 1, 1], [1] * 3, True, False,
                               f, w, dout, torch.float32)
    _check("out", out, out_ref, 1e-3)
    _check("din", din, din_ref, 1e-3)
    _check("dw", dw, dw_ref, 1e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("ksize,stride,pad,subm,C,K", [
    ([5, 5, 5], [1] * 3, [2] * 3, True, 32, 64),       # kv = 125: four mask words
    ([5, 3, 3], [1] * 3, [2, 1, 1], True, 64, 64),     # kv = 45
    ([4, 4, 4], [2] * 3, [1] * 3, False, 16, 32),      # kv = 64, regular conv
    ([3, 5, 3], [2, 1, 2], [1, 2, 1], False, 48, 40),  # kv = 45, padded channel counts
])
def test_kernel_volumes_33_to_128_run_in_groups_of_32(cuda, dtype, ksize, stride, pad, subm, C, K):
    """Kernel volumes beyond one mask word (the reference's multi-word masks): forward, dgrad and
    wgrad on the MFMA kernels (ceil(kv / 32) launches through an fp32 scratch) against the oracle."""
    shape = [18, 20, 22]
    idx, ref, f, w, dout = _case(shape, 2500, 2, C, K, ksize, stride, pad, [1] * 3, subm, dtype)
    out_ref = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"], subm=subm)
    din_ref, dw_ref = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"], subm=subm)
    rb, out, din, dw = _run_gpu(cuda, idx, 2, shape, ksize, stride, pad, [1] * 3, subm, False, f, w, dout, dtype)
    assert rb.mask_fwd.shape[1] == (rb.kv + 31) // 32 > 1
    tol = TOL[dtype]
    _check_abs((out, din, dw), (out_ref, din_ref, dw_ref), f, w, dout, ref, subm, dtype)
    _check("out", out, out_ref, tol)
    _check("din", din, din_ref, tol)
    _check("dw", dw, dw_ref, tol)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("ksize,stride,pad,subm,C,K", [
    ([7, 7, 7], [1] * 3, [3] * 3, True, 16, 16),       # kv = 343 (odd: SubM mirror counts resolved on the host side)
    ([6, 6, 6], [2] * 3, [2] * 3, False, 16, 32),      # kv = 216, regular conv
])
def test_kernel_volumes_beyond_128_train(cuda, dtype, ksize, stride, pad, subm, C, K):
    """The reference's Native path trains any kernel volume (ops.py:962-1015): forward and dgrad on the
    generic kernels, wgrad through spx_igemm_wgrad 128 offsets at a time -- all against the oracle."""
    shape = [14, 16, 18]
    idx, ref, f, w, dout = _case(shape, 600, 2, C, K, ksize, stride, pad, [1] * 3, subm, dtype)
    out_ref = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"], subm=subm)
    din_ref, dw_ref = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"], subm=subm)
    rb, out, din, dw = _run_gpu(cuda, idx, 2, shape, ksize, stride, pad, [1] * 3, subm, False, f, w, dout, dtype)
    assert rb.kv > 128
    tol = TOL[dtype]
    _check("out", out, out_ref, tol)
    _check("din", din, din_ref, tol)
    _check("dw", dw, dw_ref, tol)


def test_module_with_kernel_volume_beyond_128_steps(cuda):
    """A 7x7x7 SubMConv3d layer trains end to end (forward + backward through the autograd function)."""
    import spconv_amd.pytorch as spconv
    shape = [12, 12, 12]
    idx = torch.from_numpy(scene(shape, 400, 1, seed=3)).to(cuda)
    torch.manual_seed(0)
    net = spconv.SubMConv3d(8, 8, 7, padding=3, bias=False, indice_key="k7").to(cuda)
    assert int(np.prod(net.kernel_size)) == 343
    f = torch.randn(idx.shape[0], 8, device=cuda, requires_grad=True)
    y = net(spconv.SparseConvTensor(f, idx, shape, 1))
    y.features.square().sum().backward()
    assert net.weight.grad is not None and torch.isfinite(net.weight.grad).all()
    assert f.grad is not None and float(f.grad.abs().sum()) > 0


@pytest.mark.parametrize("subm", [True, False])
def test_mask_sorted_order_gives_same_result(cuda, subm):
    """mask_argsort only permutes which workgroup owns a row; results must not change."""
    shape = [24, 24, 24]
    stride = [1] * 3 if subm else [2] * 3
    idx, ref, f, w, dout = _case(shape, 2500, 2, 64, 64, [3] * 3, stride, [1] * 3, [1] * 3, subm,
                                 torch.float16)
    a = _run_gpu(cuda, idx, 2, shape, [3] * 3, stride, [1] * 3, [1] * 3, subm, False, f, w, dout,
                 torch.float16, use_sort=False)
    b = _run_gpu(cuda, idx, 2, shape, [3] * 3, stride, [1] * 3, [1] * 3, subm, False, f, w, dout,
                 torch.float16, use_sort=True)
    assert b[0].argsort_fwd is not None
    for x, y in zip(a[1:], b[1:]):
        torch.testing.assert_close(x, y, rtol=0, atol=0)


@pytest.mark.parametrize("subm", [True, False])
def test_tile_order_tables_give_same_result(cuda, subm):
    """Mask-sorted rows with the tables copied into tile order (ops.sort_rulebook / tables_of: what
    the modules use for dense scenes) against the unsorted launch: bit-identical forward, dgrad,
    fused backward."""
    from spconv_amd.pytorch import ops
    shape = [24, 40, 40]
    stride = [1] * 3 if subm else [2] * 3
    idx = dense_scene([36, 120, 120], 30000, 2, seed=9)
    rb, _ = gpu_rulebook(idx, 2, shape, [3] * 3, stride, [1] * 3, [1] * 3, subm)
    torch.manual_seed(0)
    C, K = 32, 64
    f = torch.randn(rb.n_in, C, device=cuda).half()
    w = (torch.randn(K, 3, 3, 3, C, device=cuda) * 0.2).half()
    d = (torch.randn(rb.n_out, K, device=cuda) * 0.2).half()
    ident = 13 if subm else -1
    bw = "fwd" if subm else "bwd"
    t0 = (rb.pair_fwd, rb.mask_fwd) if subm else (rb.pair_bwd, rb.mask_bwd)
    out0 = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, ident)
    din0, dw0 = ops.igemm_bwd(f, d, w, t0[0], t0[1], None, rb.pair_native, rb.num_per_loc, subm, ops._plan_of(rb))
    ops.sort_rulebook(rb)
    pair, mask, order, to = ops.tables_of(rb, "fwd", K)
    assert to and order is not None and pair is not rb.pair_fwd
    assert torch.equal(pair, rb.pair_fwd[:, order.long()]) and torch.equal(mask, rb.mask_fwd[order.long()])
    out1 = ops.igemm_fwd(f, w, pair, mask, order, rb.n_out, ident, tile_order=to)
    assert torch.equal(out0, out1)
    pair, mask, order, to = ops.tables_of(rb, bw, C)
    din1, dw1 = ops.igemm_bwd(f, d, w, pair, mask, order, rb.pair_native, rb.num_per_loc, subm, ops._plan_of(rb),
                              tile_order=to)
    assert torch.equal(din0, din1) and torch.equal(dw0, dw1)
    din2 = ops.igemm_dgrad(d, w, pair, mask, order, rb.n_in, subm, tile_order=to)
    assert torch.equal(din0, din2)


def test_transposed_and_inverse_conv(cuda):
    from spconv_amd.pytorch import ops
    shape = [10, 9, 9]
    ksize, stride, pad, dil = [3] * 3, [2] * 3, [1] * 3, [1] * 3
    # transposed conv: same kernels, rulebook built with transposed coordinates
    idx, ref, f, w, dout = _case(shape, 400, 2, 16, 32, ksize, stride, pad, dil, False,
                                 torch.float32, transposed=True, dense=False)
    out_ref = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"])
    din_ref, dw_ref = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"])
    _, out, din, dw = _run_gpu(cuda, idx, 2, shape, ksize, stride, pad, dil, False, True, f, w,
                               dout, torch.float32)
    _check("deconv out", out, out_ref, 1e-3)
    _check("deconv din", din, din_ref, 1e-3)
    _check("deconv dw", dw, dw_ref, 1e-3)
    # inverse conv over a regular conv's rulebook (conv.py:348-363 role swap)
    shape = [20, 20, 20]
    idx, ref, f, w, _ = _case(shape, 1500, 1, 16, 32, ksize, stride, pad, dil, False, torch.float32)
    rng = np.random.default_rng(4)
    g = torch.from_numpy(rng.uniform(-1, 1, (ref["n_out"], 32)).astype(np.float32))    # features at conv outputs
    wi = torch.from_numpy(rng.uniform(-1, 1, (16, 3, 3, 3, 32)).astype(np.float32))    # 32 -> 16
    dinv = torch.from_numpy(rng.uniform(-0.2, 0.2, (ref["n_in"], 16)).astype(np.float32))
    out_ref = oracle.indice_conv(g, wi, ref["pair"], ref["num"], ref["n_in"], inverse=True)
    din_ref, dw_ref = oracle.indice_conv_backward(g, wi, dinv, ref["pair"], ref["num"], inverse=True)
    rb, _ = gpu_rulebook(idx, 1, shape, ksize, stride, pad, dil, False)
    pair = ops.attach_rulebook(rb.pair_native, rb)
    out = ops.indice_conv(g.to(cuda), wi.to(cuda), pair, rb.num_per_loc, rb.n_in, inverse=True)
    din, dw = ops.indice_conv_backward(g.to(cuda), wi.to(cuda), dinv.to(cuda), pair, rb.num_per_loc,
                                       inverse=True)
    _check("inverse out", out.cpu(), out_ref, 1e-3)
    _check("inverse din", din.cpu(), din_ref, 1e-3)
    _check("inverse dw", dw.cpu(), dw_ref, 1e-3)
    # same through bare tensors (layout-conversion path, no attached rulebook)
    bare = rb.pair_native.clone()
    out2 = ops.indice_conv(g.to(cuda), wi.to(cuda), bare, rb.num_per_loc, rb.n_in, inverse=True)
    din2, dw2 = ops.indice_conv_backward(g.to(cuda), wi.to(cuda), dinv.to(cuda), bare,
                                         rb.num_per_loc, inverse=True)
    torch.testing.assert_close(out2, out, rtol=0, atol=0)
    torch.testing.assert_close(din2, din, rtol=0, atol=0)
    torch.testing.assert_close(dw2, dw, rtol=0, atol=0)


def test_fused_bias_activation(cuda):
    from spconv_amd.pytorch import ops
    shape = [20, 20, 20]
    for dtype in (torch.float32, torch.float16):
        idx, ref, f, w, _ = _case(shape, 1500, 1, 32, 64, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True, dtype)
        bias = _rounded(np.random.default_rng(1).uniform(-1, 1, 64).astype(np.float32), dtype)
        base = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"], subm=True) + bias
        rb, _ = gpu_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
        for act, fn in ((ops.Activation.ReLU, torch.relu),
                        (ops.Activation.LeakyReLU, lambda x: torch.nn.functional.leaky_relu(x, 0.1)),
                        (ops.Activation.Sigmoid, torch.sigmoid)):
            out = ops.igemm_fwd(f.to(cuda, dtype), w.to(cuda, dtype), rb.pair_fwd, rb.mask_fwd, None,
                                rb.n_out, 13, bias.to(cuda, dtype), act, 0.1)
            _check(f"act{act}", out.float().cpu(), fn(base), TOL[dtype])
            raw = ops.igemm_fwd(f.to(cuda, dtype), w.to(cuda, dtype), rb.pair_fwd, rb.mask_fwd, None,
                                rb.n_out, 13)
            out2 = ops.bias_act_inplace(raw, bias.to(cuda, dtype), act, 0.1)
            _check(f"inplace act{act}", out2.float().cpu(), fn(base), 2 * TOL[dtype])


def test_cfg2_full_size_fp16(cuda):
    """BASELINE cfg 2 at full size (100k voxels, C=K=64, fp16) against the oracle, plus
    linearity (a size-independent property): conv(a*f1 + f2) == a*conv(f1) + conv(f2)."""
    from spconv_amd.pytorch import ops
    shape = [40, 1280, 1600]
    idx, ref, f, w, dout = _case(shape, 100_000, 1, 64, 64, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True,
                                 torch.float16, dense=False)
    out_ref = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"], subm=True)
    din_ref, dw_ref = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"], subm=True)
    rb, out, din, dw = _run_gpu(cuda, idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True, False,
                                f, w, dout, torch.float16)
    _check("out", out, out_ref, 2e-3)
    _check("din", din, din_ref, 2e-3)
    _check("dw", dw, dw_ref, 2e-3)
    f1 = torch.randn(100_000, 64, device=cuda, dtype=torch.float32)
    f2 = torch.randn(100_000, 64, device=cuda, dtype=torch.float32)
    wg = w.to(cuda)
    conv = lambda x: ops.igemm_fwd(x, wg, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13)
    lhs, rhs = conv(2.0 * f1 + f2), 2.0 * conv(f1) + conv(f2)
    assert rel_err(lhs.cpu().numpy(), rhs.cpu().numpy()) < 1e-4


def test_lidar_like_scene_fp16(cuda):
    """Dense neighbourhoods (~5 pairs/voxel): exercises many offsets per tile."""
    from spconv_amd.utils import synthetic
    shape = [40, 1280, 1600]
    idx = synthetic.lidar_like_scene(shape, 40_000, 1, seed=0)
    ref = oracle_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    rng = np.random.default_rng(0)
    f = _rounded(rng.uniform(-1, 1, (idx.shape[0], 64)).astype(np.float32), torch.float16)
    w = _rounded(rng.uniform(-1, 1, (64, 3, 3, 3, 64)).astype(np.float32), torch.float16)
    dout = _rounded(rng.uniform(-0.2, 0.2, (idx.shape[0], 64)).astype(np.float32), torch.float16)
    out_ref = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"], subm=True)
    din_ref, dw_ref = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"], subm=True)
    for use_sort in (False, True):
        _, out, din, dw = _run_gpu(cuda, idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True,
                                   False, f, w, dout, torch.float16, use_sort=use_sort)
        _check("out", out, out_ref, 2e-3)
        _check("din", din, din_ref, 2e-3)
        _check("dw", dw, dw_ref, 2e-3)


def _oracle_fwd_bwd(idx, bs, shape, ksize, stride, pad, dil, subm, f, w, dout_fn):
    ref = oracle_rulebook(idx, bs, shape, ksize, stride, pad, dil, subm)
    dout = dout_fn(ref["n_out"])
    out = oracle.indice_conv(f, w, torch.from_numpy(ref["pair"]), torch.from_numpy(ref["num"]), ref["n_out"],
                             subm=subm)
    din, dw = oracle.indice_conv_backward(f, w, dout, torch.from_numpy(ref["pair"]),
                                          torch.from_numpy(ref["num"]), subm=subm)
    return ref, dout, out, din, dw


@pytest.mark.parametrize("subm", [True, False])
def test_ragged_batch_with_an_empty_scene(cuda, subm):
    """Scenes of very different sizes in one batch, one of them empty (batch id 1 never occurs)
    and one holding a single voxel: rulebook bit-exact, features within tolerance."""
    shape, bs, C, K = [16, 18, 20], 4, 32, 32
    parts = [dense_scene(shape, 1800, 1, 1), dense_scene(shape, 1, 1, 2), dense_scene(shape, 300, 1, 3)]
    for b, p in zip((0, 2, 3), parts):
        p[:, 0] = b
    idx = np.ascontiguousarray(np.concatenate(parts))
    idx = idx[np.random.default_rng(0).permutation(idx.shape[0])]            # scenes interleaved
    rng = np.random.default_rng(1)
    f = _rounded(rng.uniform(-1, 1, (idx.shape[0], C)).astype(np.float32), torch.float16)
    w = _rounded(rng.uniform(-1, 1, (K, 3, 3, 3, C)).astype(np.float32), torch.float16)
    stride, pad = ([1] * 3, [1] * 3) if subm else ([2] * 3, [1] * 3)
    ref, dout, out, din, dw = _oracle_fwd_bwd(
        idx, bs, shape, [3] * 3, stride, pad, [1] * 3, subm, f, w,
        lambda n: _rounded(rng.uniform(-0.2, 0.2, (n, K)).astype(np.float32), torch.float16))
    rb, gout, gdin, gdw = _run_gpu(cuda, idx, bs, shape, [3] * 3, stride, pad, [1] * 3, subm, False, f, w, dout,
                                   torch.float16)
    from util import assert_rulebook_equal
    assert_rulebook_equal(rb, ref, subm)
    assert not (to_np(rb.out_indices)[:, 0] == 1).any()
    _check("out", gout, out, TOL[torch.float16])
    _check("din", gdin, din, TOL[torch.float16])
    _check("dw", gdw, dw, TOL[torch.float16])


def test_single_voxel_and_zero_voxels(cuda):
    """n = 1 (one workgroup, one row) and n = 0 (every kernel is a no-op, outputs are empty,
    the weight gradient is exactly zero)."""
    from spconv_amd.pytorch import ops
    shape, C, K = [8, 8, 8], 16, 32
    w = torch.randn(K, 3, 3, 3, C, device=cuda).half()
    one = np.array([[0, 3, 4, 5]], dtype=np.int32)
    rb, _ = gpu_rulebook(one, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    f = torch.randn(1, C, device=cuda).half()
    out = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, 1, 13)
    want = f.float() @ w[:, 1, 1, 1, :].float().t()
    assert torch.allclose(out.float(), want, atol=2e-2, rtol=2e-3)
    din, dw = ops.igemm_bwd(f, out, w, rb.pair_fwd, rb.mask_fwd, None, rb.pair_native, rb.num_per_loc, True,
                            ops._plan_of(rb))
    assert torch.allclose(din.float(), out.float() @ w[:, 1, 1, 1, :].float(), atol=5e-1, rtol=5e-3)
    centre = dw[:, 1, 1, 1, :].float()
    assert torch.allclose(centre, out.float().t() @ f.float(), atol=5e-2, rtol=5e-3)
    dw_off = dw.float().clone()
    dw_off[:, 1, 1, 1, :] = 0
    assert float(dw_off.abs().max()) == 0.0                        # no neighbours: only the centre tap
    empty = np.zeros((0, 4), dtype=np.int32)
    rb0, _ = gpu_rulebook(empty, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    f0 = torch.zeros(0, C, device=cuda).half()
    out0 = ops.igemm_fwd(f0, w, rb0.pair_fwd, rb0.mask_fwd, None, 0, 13)
    assert out0.shape == (0, K)
    dw0 = ops.igemm_wgrad(f0, out0, w.shape, rb0.pair_native, rb0.num_per_loc, True, None)
    assert dw0.shape == w.shape and float(dw0.float().abs().max()) == 0.0
    # the same through the module + autograd: an empty scene yields an empty output and zero gradients
    import spconv_amd.pytorch as spconv
    net = spconv.SubMConv3d(C, K, 3, bias=True).to(cuda).half()
    x = spconv.SparseConvTensor(f0.clone().requires_grad_(True), torch.zeros(0, 4, dtype=torch.int32, device=cuda),
                                shape, 1)
    y = net(x)
    assert y.features.shape == (0, K)
    y.features.sum().backward()
    assert net.weight.grad is not None and float(net.weight.grad.float().abs().max()) == 0.0


def test_one_million_voxels_identity_and_linearity(cuda):
    """Size-independent properties at a size the oracle does not reach (1 M voxels, batch 8):
    an identity centre tap reproduces the input bit-for-bit, the op is linear in the features,
    and the rulebook obeys the SubM mirror symmetry."""
    from spconv_amd.pytorch import ops
    shape, bs, C = [40, 400, 400], 8, 32
    idx = scene(shape, 125_000, bs, seed=7)
    rb, _ = gpu_rulebook(idx, bs, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    n = idx.shape[0]
    assert n == 1_000_000
    pf = rb.pair_fwd
    # mirror symmetry: pair[k][o] = v  <=>  pair[26-k][v] = o  (checked on a sample of offsets)
    for k in (0, 5, 12):
        o = torch.nonzero(pf[k] >= 0).flatten()
        v = pf[k][o].long()
        assert torch.equal(pf[26 - k][v].long(), o)
    assert torch.equal(pf[13], torch.arange(n, device=cuda, dtype=torch.int32))
    g = torch.Generator(device="cpu").manual_seed(0)
    f1 = torch.randn(n, C, generator=g).to(cuda).half()
    f2 = torch.randn(n, C, generator=g).to(cuda).half()
    w = torch.zeros(C, 3, 3, 3, C, device=cuda).half()
    w[:, 1, 1, 1, :] = torch.eye(C, device=cuda).half()
    assert torch.equal(ops.igemm_fwd(f1, w, pf, rb.mask_fwd, None, n, 13), f1)
    wr = (torch.randn(C, 3, 3, 3, C, generator=g) * 0.1).to(cuda).half()
    a = ops.igemm_fwd(f1, wr, pf, rb.mask_fwd, None, n, 13).float()
    b = ops.igemm_fwd(f2, wr, pf, rb.mask_fwd, None, n, 13).float()
    ab = ops.igemm_fwd((f1.float() + f2.float()).half(), wr, pf, rb.mask_fwd, None, n, 13).float()
    assert float((ab - (a + b)).abs().max()) < 2e-2 * float((a + b).abs().max())


@pytest.mark.timeout(600)
def test_tensors_beyond_two_gigabytes(cuda):
    """Maximum sizes: 17 M voxels x 64 fp16 channels = 2.18 GB per feature tensor, past the 32-bit
    byte offsets the fast kernels address with -- the library must switch to its 64-bit-offset
    kernels.  A fully occupied 41 x 644 x 644 block in raster order makes every expected value
    analytic: neighbour index = row + dz*Y*X + dy*X + dx, pair counts = products of (extent - |d|)."""
    from spconv_amd.pytorch import ops
    Z, Y, X, C = 41, 644, 644, 64
    n = Z * Y * X
    assert n * C * 2 > 2 ** 31
    zz, yy, xx = torch.meshgrid(torch.arange(Z, dtype=torch.int32), torch.arange(Y, dtype=torch.int32),
                                torch.arange(X, dtype=torch.int32), indexing="ij")
    idx = torch.stack([torch.zeros_like(zz), zz, yy, xx], dim=-1).reshape(n, 4).contiguous().to(cuda)
    del zz, yy, xx
    rb, _ = ops.build_rulebook(idx, 1, [Z, Y, X], [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)
    num = rb.num_per_loc.cpu().numpy()
    offs = [(a - 1, b - 1, c - 1) for a in range(3) for b in range(3) for c in range(3)]
    for k in range(13):
        dz, dy, dx = offs[k]
        assert num[k] == (Z - abs(dz)) * (Y - abs(dy)) * (X - abs(dx)), k
    g = torch.Generator().manual_seed(0)
    f = torch.empty(n, C, dtype=torch.float16, device=cuda).uniform_(-1, 1)
    w = (torch.rand(C, 3, 3, 3, C, generator=g) * 2 - 1).to(cuda).half()
    out = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n, 13)
    rows = torch.randint(0, n, (4096,), generator=g).to(cuda)
    rows[:8] = torch.tensor([0, 1, X, Y * X, n - 1, n - 2, n - X, n - Y * X])     # corners / faces
    z, y, x = rows // (Y * X), (rows // X) % Y, rows % X

    def expect(feat, weight_of):
        acc = torch.zeros(rows.shape[0], C, device=cuda)
        for k, (dz, dy, dx) in enumerate(offs):
            ok = ((z + dz >= 0) & (z + dz < Z) & (y + dy >= 0) & (y + dy < Y) & (x + dx >= 0) & (x + dx < X))
            src = (rows + dz * Y * X + dy * X + dx).clamp(0, n - 1)
            acc += (feat[src].float() * ok[:, None]) @ weight_of(k)
        return acc
    want = expect(f, lambda k: w.reshape(C, 27, C)[:, k, :].float().t())
    got = out[rows].float()
    assert float((got - want).abs().max() / want.abs().max()) < 3e-3
    # dgrad: din[i] = sum_k dout[i - offset_k] W_k  ==  forward with the mirrored offset and W_k^T
    dout = torch.empty(n, C, dtype=torch.float16, device=cuda).uniform_(-0.2, 0.2)
    din = ops.igemm_dgrad(dout, w, rb.pair_fwd, rb.mask_fwd, None, n, True)
    want = expect(dout, lambda k: w.reshape(C, 27, C)[:, 26 - k, :].float())
    assert float((din[rows].float() - want).abs().max() / want.abs().max()) < 3e-3
    del din, out
    # wgrad on three offsets: dW_k = dout[out rows]^T f[in rows], rows of the pair list are analytic
    dw = ops.igemm_wgrad(f, dout, w.shape, rb.pair_native, rb.num_per_loc, True, ops._plan_of(rb))
    for k in (0, 13, 22):
        dz, dy, dx = offs[k]
        shift = dz * Y * X + dy * X + dx
        o = torch.arange(n, device=cuda)
        oz, oy, ox = o // (Y * X), (o // X) % Y, o % X
        ok = ((oz + dz >= 0) & (oz + dz < Z) & (oy + dy >= 0) & (oy + dy < Y) & (ox + dx >= 0) & (ox + dx < X))
        o = o[ok]
        del oz, oy, ox, ok
        want_k = torch.zeros(C, C, device=cuda)
        for a in range(0, o.shape[0], 2_000_000):
            oo = o[a:a + 2_000_000]
            want_k += dout[oo].float().t() @ f[oo + shift].float()
        got_k = dw.reshape(C, 27, C)[:, k, :].float()
        assert float((got_k - want_k).abs().max() / want_k.abs().max()) < 5e-3, k


@pytest.mark.parametrize("scene_kind", ["uniform", "dense"])
def test_backward_is_bit_reproducible(cuda, scene_kind):
    """No order-dependent atomics anywhere: the fused backward (dgrad tiles + wgrad ranges + fixed-order
    second stage) and the rulebook return bit-identical tensors run after run, also while another
    stream keeps the GPU busy."""
    from spconv_amd.pytorch import ops
    shape = [40, 200, 200]
    idx = scene(shape, 60000, 1, 3) if scene_kind == "uniform" else dense_scene([60, 90, 90], 60000, 1, 3)
    t = torch.from_numpy(idx).to(cuda)
    sh = shape if scene_kind == "uniform" else [60, 90, 90]
    n, C = idx.shape[0], 64
    g = torch.Generator().manual_seed(0)
    f = torch.randn(n, C, generator=g).to(cuda).half()
    d = (torch.randn(n, C, generator=g) * 0.1).to(cuda).half()
    w = (torch.randn(C, 3, 3, 3, C, generator=g) * 0.1).to(cuda).half()
    noise = torch.randn(4096, 4096, device=cuda)
    side = torch.cuda.Stream()
    ref = None
    for it in range(4):
        if it % 2:                                   # perturb scheduling: a GEMM on a second stream
            with torch.cuda.stream(side):
                noise @ noise
        rb, _ = ops.build_rulebook(t, 1, sh, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)
        out = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n, 13)
        din, dw = ops.igemm_bwd(f, d, w, rb.pair_fwd, rb.mask_fwd, None, rb.pair_native, rb.num_per_loc, True,
                                ops._plan_of(rb))
        torch.cuda.synchronize()
        cur = (rb.pair_fwd.clone(), rb.pair_native.clone(), rb.mask_fwd.clone(), out, din, dw)
        if ref is None:
            ref = cur
        else:
            for a, b in zip(ref, cur):
                assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("C,K,ksize,stride,pad,subm,n", [
    (8, 16, [3] * 3, [1] * 3, [1] * 3, True, 6000),        # 16-byte rows: four offsets per step forward
    (16, 16, [3] * 3, [1] * 3, [1] * 3, True, 6000),       # 32-byte rows: two per step, forward and dgrad
    (16, 32, [3] * 3, [2] * 3, [1] * 3, False, 6000),      # strided layer, narrow forward
    (32, 16, [3] * 3, [2] * 3, [1] * 3, False, 6000),      # ... narrow dgrad (16 dout channels)
    (16, 16, [2] * 3, [2] * 3, [0] * 3, False, 6000),      # kernel volume 8: an even count, no identity offset
    (8, 8, [3, 1, 1], [1] * 3, [1, 0, 0], True, 3000),     # kernel volume 3: identity + one packed step with a hole
    (16, 64, [3] * 3, [1] * 3, [1] * 3, True, 40_000),     # rows layout territory (> 32 k rows)
    (32, 32, [3] * 3, [1] * 3, [1] * 3, True, 6000),       # 64-byte rows: packed only across the two pieces of a step
    (16, 16, [5, 3, 3], [1] * 3, [2, 1, 1], True, 3000),   # kernel volume 45: two groups of offsets (two mask words)
])
def test_packed_offsets_for_narrow_rows(cuda, C, K, ksize, stride, pad, subm, n, dtype):
    """igemm_v4_body, PK: reduction rows of <= 32 / 16 bytes carry 2 / 4 offsets per MFMA step (the lanes that used to
    multiply the zeros behind the row's end gather another offset's row).  Against the oracle with the sharp bars, and
    against the one-offset-per-step walk (SPX_PK = 0) of the same library -- the same values up to fp32 association."""
    from spconv_amd import _lib
    shape = [40, 64, 64] if n > 10_000 else [24, 24, 24]
    idx, ref, f, w, dout = _case(shape, n, 2, C, K, ksize, stride, pad, [1] * 3, subm, dtype)
    out_ref = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"], subm=subm)
    din_ref, dw_ref = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"], subm=subm)
    L = _lib.load()
    res = {}
    try:
        for pk in (1, 3, 0):          # the rule | both pieces of a step packed everywhere | one offset per step
            L.spx_set_option(b"SPX_PK", pk)
            _, out, din, dw = _run_gpu(cuda, idx, 2, shape, ksize, stride, pad, [1] * 3, subm, False, f, w, dout, dtype)
            res[pk] = (out, din, dw)
    finally:
        L.spx_set_option(b"SPX_PK", 1)
    tol = TOL[dtype]
    for pk in (1, 3, 0):
        out, din, dw = res[pk]
        _check("out", out, out_ref, tol)
        _check("din", din, din_ref, tol)
        _check("dw", dw, dw_ref, tol)
        _check_abs((out, din, dw), (out_ref, din_ref, dw_ref), f, w, dout, ref, subm, dtype)
    # the two walks against each other: one rounding of the stored dtype apart at most, nearly everywhere equal
    for pk in (1, 3):
        for a, b in zip(res[pk][:2], res[0][:2]):
            a, b = a.float(), b.float()
            assert float((a - b).abs().max()) <= 2 * tol * float(b.abs().max()) + 1e-6
            assert float((a != b).float().mean()) < 0.2
