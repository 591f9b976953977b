"""BatchNorm1d (+ fused ReLU) over sparse-tensor features (csrc/norm.hip, spconv_amd/pytorch/norm.py)
against torch.nn.BatchNorm1d evaluated in fp32 on the same (rounded) inputs: output, input gradient,
affine gradients, running statistics, evaluation mode, and the SparseSequential hook."""
import copy

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _ref(bn32, x32, dy32, relu):
    x = x32.clone().requires_grad_(True)
    y = bn32(x)
    if relu:
        y = torch.relu(y)
    y.backward(dy32)
    return y.detach(), x.grad, bn32.weight.grad, bn32.bias.grad


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("n,C", [(100_003, 16), (40_000, 64), (7_777, 128), (5_000, 48), (300, 256), (1, 32)])
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm_training_matches_torch(cuda, dtype, tol, n, C, relu):
    from spconv_amd.pytorch import norm
    if dtype == torch.float32 and C % 4:
        pytest.skip("shape")
    torch.manual_seed(n + C)
    x = (torch.randn(n, C, device=cuda) * 1.7 + torch.linspace(-3, 3, C, device=cuda)).to(dtype)
    dy = torch.randn(n, C, device=cuda).to(dtype)
    bn = nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(cuda)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    ref = copy.deepcopy(bn).float()
    xg = x.clone().requires_grad_(True)
    if n == 1:
        # torch refuses one value per channel in training ("Expected more than 1 value per channel");
        # the fused path stands aside for such input so that the user sees torch's error (round-2 ADVICE)
        assert not norm.supported(x, bn)
        with pytest.raises(ValueError):
            ref(x.float())
        return
    assert norm.supported(x, bn)
    y = norm.batch_norm(xg, bn, relu=relu)
    y.backward(dy)
    y_ref, dx_ref, dw_ref, db_ref = _ref(ref, x.float(), dy.float(), relu)
    scale = lambda t: float(t.abs().max()) + 1e-12
    assert float((y.float() - y_ref).abs().max()) <= tol * scale(y_ref)
    assert float((xg.grad.float() - dx_ref).abs().max()) <= tol * scale(dx_ref)
    assert float((bn.weight.grad - dw_ref).abs().max()) <= max(tol, 1e-4) * scale(dw_ref)
    assert float((bn.bias.grad - db_ref).abs().max()) <= max(tol, 1e-4) * scale(db_ref)
    assert torch.allclose(bn.running_mean, ref.running_mean, rtol=1e-4, atol=1e-5)
    assert torch.allclose(bn.running_var, ref.running_var, rtol=1e-4, atol=1e-5)
    assert int(bn.num_batches_tracked) == 1


def test_batchnorm_eval_mode_and_momentum_none(cuda):
    from spconv_amd.pytorch import norm
    torch.manual_seed(0)
    C, n = 32, 9000
    x = torch.randn(n, C, device=cuda).half()
    bn = nn.BatchNorm1d(C, momentum=None).to(cuda)
    ref = copy.deepcopy(bn)
    for _ in range(3):                                   # cumulative moving average
        norm.batch_norm(x, bn)
        ref(x.float())
    assert int(bn.num_batches_tracked) == 3
    assert torch.allclose(bn.running_mean, ref.running_mean, atol=1e-4)
    assert torch.allclose(bn.running_var, ref.running_var, rtol=1e-3, atol=1e-4)
    bn.eval()
    ref.eval()
    xg = x.clone().requires_grad_(True)
    y = norm.batch_norm(xg, bn, relu=True)
    y.float().sum().backward()
    xr = x.float().requires_grad_(True)
    yr = torch.relu(ref(xr))
    yr.sum().backward()
    assert float((y.float() - yr).abs().max()) < 3e-3 * float(yr.abs().max())
    assert float((xg.grad.float() - xr.grad).abs().max()) < 3e-3 * float(xr.grad.abs().max())


def test_sparse_sequential_takes_the_fused_path(cuda, monkeypatch):
    """Conv -> BatchNorm1d -> ReLU in a SparseSequential: same result with the streaming kernels and
    with torch's BatchNorm (SPCONV_AMD_FUSED_BN=0), one launch set for BN + ReLU."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch import norm
    from spconv_amd.utils import synthetic
    shape = [16, 40, 40]
    idx = synthetic.lidar_like_scene(shape, 6000, 2, seed=1)
    torch.manual_seed(1)
    net = spconv.SparseSequential(spconv.SubMConv3d(8, 32, 3, bias=False, indice_key="a"), nn.BatchNorm1d(32),
                                  nn.ReLU(), spconv.SparseConv3d(32, 64, 3, 2, 1, bias=False), nn.BatchNorm1d(64),
                                  nn.ReLU()).to(cuda).half().train()
    ref = copy.deepcopy(net)
    f = torch.randn(idx.shape[0], 8, device=cuda).half()
    calls = []
    orig = norm.batch_norm
    monkeypatch.setattr(norm, "batch_norm", lambda *a, **k: (calls.append(k.get("relu")), orig(*a, **k))[1])
    y = net(spconv.SparseConvTensor(f, torch.from_numpy(idx).to(cuda), shape, 2))
    y.features.float().square().sum().backward()
    assert calls == [True, True]
    monkeypatch.setattr(norm, "ENABLED", False)
    yr = ref(spconv.SparseConvTensor(f, torch.from_numpy(idx).to(cuda), shape, 2))
    yr.features.float().square().sum().backward()
    assert float((y.features.float() - yr.features.float()).abs().max()) < 1e-2 * float(yr.features.float().abs().max())
    for p, q in zip(net.parameters(), ref.parameters()):
        assert float((p.grad.float() - q.grad.float()).abs().max()) <= 3e-2 * float(q.grad.float().abs().max()) + 1e-3


def test_half_model_parameters_are_read_in_place(cuda):
    """`.half()` converts BatchNorm's parameters and buffers too: the kernels read / update them in that
    dtype (no fp32 shadow copies), results match torch's fp16-parameter BatchNorm."""
    from spconv_amd.pytorch import norm
    torch.manual_seed(3)
    n, C = 20_000, 64
    x = torch.randn(n, C, device=cuda).half()
    bn = nn.BatchNorm1d(C).to(cuda).half()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    ref = copy.deepcopy(bn)
    dy = (torch.randn(n, C, device=cuda) * 0.01).half()       # (keeps the fp16 parameter gradients finite)
    xg = x.clone().requires_grad_(True)
    y = norm.batch_norm(xg, bn, relu=False)
    y.backward(dy)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy)
    assert bn.running_mean.dtype == torch.float16 and bn.weight.grad.dtype == torch.float16
    assert float((y.float() - yr.float()).abs().max()) < 4e-3 * float(yr.float().abs().max())
    assert float((xg.grad.float() - xr.grad.float()).abs().max()) < 6e-3 * float(xr.grad.float().abs().max())
    assert torch.allclose(bn.running_mean.float(), ref.running_mean.float(), atol=2e-3)
    assert torch.allclose(bn.running_var.float(), ref.running_var.float(), rtol=2e-3, atol=2e-3)
    assert torch.allclose(bn.weight.grad.float(), ref.weight.grad.float(), rtol=2e-2, atol=2e-2)
    assert torch.allclose(bn.bias.grad.float(), ref.bias.grad.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 2e-3)])
@pytest.mark.parametrize("n,live,C", [(50_000, 41_237, 32), (5_000, 1, 64), (9_000, 9_000, 16), (3_000, 0, 32)])
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm_over_the_live_rows_of_a_static_tensor(cuda, dtype, tol, n, live, C, relu):
    """n_live (device int32): the statistics, the affine gradients and dx are those of the first `live` rows
    alone; padding rows come out as zeros in both directions (spconv_amd/pytorch/static.py)."""
    from spconv_amd.pytorch import norm
    torch.manual_seed(n + live + C)
    x = (torch.randn(n, C, device=cuda) * 1.3 + torch.linspace(-2, 2, C, device=cuda)).to(dtype)
    x[live:] = 1000.0                              # junk in the padding must not reach the statistics
    dy = torch.randn(n, C, device=cuda).to(dtype)
    bn = nn.BatchNorm1d(C, eps=1e-3, momentum=0.1).to(cuda)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    ref = copy.deepcopy(bn)
    n_live = torch.tensor([live], dtype=torch.int32, device=cuda)
    xg = x.clone().requires_grad_(True)
    y = norm.batch_norm(xg, bn, relu=relu, n_live=n_live)
    y.backward(dy)
    assert not bool(y[live:].any()) and not bool(xg.grad[live:].any())
    if live <= 1:
        return                                     # (torch refuses one value per channel; the zeros are the claim)
    xr = x[:live].clone().requires_grad_(True)
    yr = norm.batch_norm(xr, ref, relu=relu)       # the same kernels on the live rows alone
    yr.backward(dy[:live])
    scale = float(yr.float().abs().max())
    assert float((y[:live].float() - yr.float()).abs().max()) <= tol * max(scale, 1.0)
    assert float((xg.grad[:live].float() - xr.grad.float()).abs().max()) <= tol * max(float(xr.grad.float().abs().max()), 1.0)
    for a, b in ((bn.weight.grad, ref.weight.grad), (bn.bias.grad, ref.bias.grad),
                 (bn.running_mean, ref.running_mean), (bn.running_var, ref.running_var)):
        assert float((a.float() - b.float()).abs().max()) <= 1e-3 * max(float(b.float().abs().max()), 1.0)


def _conv_bn(spconv, cin, cout, dev, dtype, subm=True):
    torch.manual_seed(cin * 1000 + cout)
    conv = (spconv.SubMConv3d(cin, cout, 3, bias=False, indice_key="k") if subm
            else spconv.SparseConv3d(cin, cout, 3, 2, 1, bias=False, indice_key="d"))
    net = spconv.SparseSequential(conv, nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01), nn.ReLU()).to(dev)
    with torch.no_grad():
        net[1].weight.uniform_(0.5, 1.5)
        net[1].bias.uniform_(-0.5, 0.5)
    return net.to(dtype).train()


@pytest.mark.parametrize("cin,cout,n_per,shape,subm,dtype,tol", [
    (16, 16, 6000, [24, 40, 40], True, torch.float16, 2e-3),       # igemm_v4_kernel<16>: dense-ish neighbourhoods
    (16, 32, 6000, [24, 40, 40], False, torch.float16, 2e-3),      # strided layer (its own output rows)
    (32, 32, 50_000, [40, 400, 400], True, torch.float16, 2e-3),   # 100 k uniform rows: rows layout, appendix workgroups
    (64, 64, 50_000, [40, 400, 400], True, torch.bfloat16, 1.6e-2),
    (64, 128, 4000, [24, 40, 40], True, torch.float16, 2e-3),      # 128 output channels (64-row tiles)
    (8, 16, 5000, [24, 40, 40], True, torch.float32, 5e-5),        # fp32 MFMA path
])
def test_conv_epilogue_statistics_feed_the_batchnorm(cuda, monkeypatch, cin, cout, n_per, shape, subm, dtype, tol):
    """BatchNorm statistics out of the producing convolution's epilogue (spx_igemm_fwd_stats ->
    spx_batchnorm_fwd_stats).  The reference hands `.features` to torch's BatchNorm1d (spconv/pytorch/modules.py:127-168),
    which is the oracle here: y, every gradient and the running estimates against nn.BatchNorm1d in fp32 over the
    convolution's (rounded) output, with the statistics taken (a) from the epilogue records and (b) -- the same numbers
    up to summation order -- from the normalisation layer's own pass (SPCONV_AMD_BN_EPILOGUE=0)."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch import norm, ops
    from util import scene
    bs = 2
    idx = torch.from_numpy(scene(shape, n_per, bs, seed=cin + cout)).to(cuda)
    n = idx.shape[0]
    f = (torch.randn(n, cin, device=cuda) * 0.7).to(dtype)
    net = _conv_bn(spconv, cin, cout, cuda, dtype, subm)
    used = []
    real = norm._BatchNormFn.apply

    def spy(*a):
        used.append(a[12] is not None)
        return real(*a)
    monkeypatch.setattr(norm._BatchNormFn, "apply", spy)

    def run(model, on):
        monkeypatch.setattr(ops, "BN_EPILOGUE", on)
        model.zero_grad(set_to_none=True)
        fe = f.clone().requires_grad_(True)
        y = model(spconv.SparseConvTensor(fe, idx, shape, bs))
        g = torch.ones_like(y.features) * 0.01 + (torch.arange(y.features.shape[1], device=cuda) % 3).to(dtype) * 0.01
        y.features.backward(g)
        torch.cuda.synchronize()
        return y.features.detach().float(), fe.grad.float(), g

    ref_net = copy.deepcopy(net)
    y_on, din_on, g = run(net, True)
    y_off, din_off, _ = run(ref_net, False)
    assert used == [True, False]
    # both against torch's BatchNorm1d in fp32 on the convolution's own output
    with torch.no_grad():
        conv_out = net[0](spconv.SparseConvTensor(f, idx, shape, bs)).features.float()
    bn32 = nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01).to(cuda).train()
    with torch.no_grad():
        bn32.weight.copy_(ref_net[1].weight.float())
        bn32.bias.copy_(ref_net[1].bias.float())
    y_ref = torch.relu(bn32(conv_out)).detach()
    scale = float(y_ref.abs().max())
    assert float((y_on - y_ref).abs().max()) <= tol * scale
    assert float((y_off - y_ref).abs().max()) <= tol * scale
    assert float((y_on - y_off).abs().max()) <= tol * scale
    assert float((din_on - din_off).abs().max()) <= 2 * tol * float(din_off.abs().max())
    for a, b in zip(net.parameters(), ref_net.parameters()):
        assert float((a.grad.float() - b.grad.float()).norm()) <= 2 * tol * float(b.grad.float().norm()) + 1e-7
    for name in ("running_mean", "running_var"):
        a, b = getattr(net[1], name).float(), getattr(bn32, name)
        assert torch.allclose(a, b, rtol=max(tol, 1e-3), atol=max(tol, 1e-3) * 0.1), name
    assert int(net[1].num_batches_tracked) == 1


def test_conv_epilogue_statistics_over_the_live_rows_of_a_static_tensor(cuda, monkeypatch):
    """Static shapes: padding rows (batch -1) behind the scene are dead rows of the convolution (zero output) and must
    not be counted -- the records' row counts follow `n_live`; a captured training step equals the eager one."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch import ops
    from spconv_amd.pytorch.static import StaticTrainingStep
    from util import scene
    shape, bs = [24, 40, 40], 2
    torch.manual_seed(2)
    net = spconv.SparseSequential(
        spconv.SubMConv3d(8, 16, 3, bias=False, indice_key="s"), nn.BatchNorm1d(16), nn.ReLU(),
        spconv.SparseConv3d(16, 32, 3, 2, 1, bias=False, indice_key="d"), nn.BatchNorm1d(32), nn.ReLU(),
        spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="s2"), nn.BatchNorm1d(32)).to(cuda).float().train()
    eager = copy.deepcopy(net)
    idx = torch.from_numpy(scene(shape, 3000, bs, seed=4)).to(cuda)
    n = idx.shape[0]
    f = torch.randn(n, 8, device=cuda)
    ye = eager(spconv.SparseConvTensor(f.clone().requires_grad_(True), idx, shape, bs))
    g = torch.randn(ye.features.shape[0] + 700, 32, device=cuda) * 0.1
    ye.features.backward(g[:ye.features.shape[0]])
    step = StaticTrainingStep(net, n + 900, 8, shape, bs, torch.float32, bounds={"3": ye.features.shape[0] + 700},
                              out_grad=g, example=(f, idx))
    out = step(f, idx)
    k = ye.features.shape[0]
    assert int(out.n_live_dev) == k
    assert float((out.features[:k] - ye.features).abs().max()) <= 2e-5 * float(ye.features.abs().max())
    assert bool((out.features[k:] == 0).all())
    for (name, a), b in zip(net.named_parameters(), eager.parameters()):
        assert float((a.grad - b.grad).norm()) <= 5e-4 * float(b.grad.norm()) + 1e-8, name
    for a, b in zip(net.buffers(), eager.buffers()):
        if a.dtype.is_floating_point:
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


def test_conv_epilogue_statistics_from_the_weight_stationary_kernel(cuda, monkeypatch):
    """Dense C = K = 64 layers take igemm_ws_kernel (512-row workgroups): its epilogue leaves the same records.  Forced
    with SPX_WS=1; the outputs of the two-launch normalisation equal the three-launch one within the rounding of y."""
    import spconv_amd.pytorch as spconv
    from spconv_amd import _lib
    from spconv_amd.pytorch import ops
    from util import scene
    L = _lib.load()
    shape, bs = [24, 64, 64], 1
    idx = torch.from_numpy(scene(shape, 40_001, bs, seed=3)).to(cuda)
    n = idx.shape[0]
    f = (torch.randn(n, 64, device=cuda) * 0.5).half()
    net = _conv_bn(spconv, 64, 64, cuda, torch.float16)
    ref = copy.deepcopy(net)
    before = L.spx_launch_count(b"igemm_ws")
    L.spx_set_option(b"SPX_WS", 1)
    try:
        monkeypatch.setattr(ops, "BN_EPILOGUE", True)
        y_on = net(spconv.SparseConvTensor(f, idx, shape, bs)).features
        monkeypatch.setattr(ops, "BN_EPILOGUE", False)
        y_off = ref(spconv.SparseConvTensor(f, idx, shape, bs)).features
        torch.cuda.synchronize()
    finally:
        L.spx_set_option(b"SPX_WS", -1)
    assert L.spx_launch_count(b"igemm_ws") - before == 2
    assert float((y_on.float() - y_off.float()).abs().max()) <= 2e-3 * float(y_off.float().abs().max())
    assert torch.allclose(net[1].running_var, ref[1].running_var, rtol=1e-3, atol=1e-4)
