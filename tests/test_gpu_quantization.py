"""GPU: quantization tooling end to end: BatchNorm folding and the QAT
conv+bn module against the float layers they replace, conversion of observed / QAT modules to the
int8 modules, and the torch.ao fx flow of the reference's MNIST PTQ / QAT examples on random data."""
import copy
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _scene(cuda, n=3000, C=16, shape=(20, 24, 28), bs=2, seed=0):
    import spconv_amd.pytorch as spconv
    from util import scene
    idx = scene(list(shape), n, bs, seed)
    g = torch.Generator().manual_seed(seed)
    f = torch.randn(idx.shape[0], C, generator=g)
    return spconv.SparseConvTensor(f.to(cuda), torch.from_numpy(idx).to(cuda), list(shape), bs)


def _trained_bn(K, cuda, seed=1):
    g = torch.Generator().manual_seed(seed)
    bn = nn.BatchNorm1d(K, eps=1e-3).to(cuda)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(K, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(K, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(K, generator=g) * 0.2)
        bn.running_var.copy_(torch.rand(K, generator=g) + 0.5)
    return bn


@pytest.mark.parametrize("subm", [True, False])
def test_bn_folding_matches_conv_then_bn(cuda, subm):
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch.quantization.utils import fuse_spconv_act_eval, fuse_spconv_bn_eval
    torch.manual_seed(0)
    x = _scene(cuda)
    conv = (spconv.SubMConv3d(16, 32, 3, bias=True) if subm else spconv.SparseConv3d(16, 32, 3, 2, 1, bias=False))
    conv = conv.to(cuda).eval()
    bn = _trained_bn(32, cuda).eval()
    want = torch.relu(bn(conv(x).features))
    fused = fuse_spconv_act_eval(fuse_spconv_bn_eval(conv, bn), nn.ReLU())
    got = fused(x).features
    assert torch.allclose(got, want, atol=2e-4, rtol=1e-4)


def test_qat_conv_bn_is_conv_then_bn_when_fake_quant_is_off(cuda):
    """intrinsic.qat.SparseConvBnReLU: scale weight by gamma/std, convolve, un-scale, BatchNorm.
    With the weight fake-quantizer disabled this must equal the float layers, in eval (running
    statistics) and in training (batch statistics, which it must also keep updating)."""
    import spconv_amd.pytorch as spconv
    import spconv_amd.pytorch.quantization as spconvq
    from spconv_amd.pytorch.quantization import intrinsic as snni
    from spconv_amd.pytorch.quantization.intrinsic import qat as snniqat
    torch.manual_seed(0)
    x = _scene(cuda)
    conv = spconv.SubMConv3d(16, 32, 3, bias=True, indice_key="k").to(cuda)
    bn = _trained_bn(32, cuda)
    fused = snni.SpconvBnReLUNd(copy.deepcopy(conv), copy.deepcopy(bn), nn.ReLU())
    fused.qconfig = spconvq.get_default_spconv_trt_qat_qconfig()
    qat = snniqat.SparseConvBnReLU.from_float(fused).to(cuda)
    qat.weight_fake_quant.disable_fake_quant()
    qat.weight_fake_quant.disable_observer()
    for training in (False, True):
        conv.train(training), bn.train(training), qat.train(training)
        ref_bn = copy.deepcopy(bn)
        want = torch.relu(ref_bn(conv(x).features))
        got = qat(x).features
        assert torch.allclose(got, want, atol=5e-4, rtol=1e-3), training
        if training:
            assert torch.allclose(qat.bn.running_mean, ref_bn.running_mean, atol=1e-5)
            got.square().mean().backward()
            assert qat.weight.grad is not None and qat.bn.weight.grad is not None and qat.bias.grad is not None
    # fake quantization on: weights snap to <= 255 levels per output channel, output stays close
    qat.eval()
    qat.weight_fake_quant.enable_observer()
    qat.weight_fake_quant.enable_fake_quant()
    stats = copy.deepcopy(qat.bn.state_dict())
    qat.freeze_bn_stats().train()(x)                # observe the weight once, statistics frozen
    qat.eval()
    assert torch.equal(qat.bn.running_mean, stats["running_mean"])
    with torch.no_grad():
        fq = qat(x).features
        bn.load_state_dict(stats)
        want = torch.relu(bn.eval()(conv.eval()(x).features))
    assert float((fq - want).abs().max() / want.abs().max()) < 0.05


def test_observed_module_converts_to_int8_conv_relu(cuda):
    """quantized.SparseConvReLU.from_float on an observed fused float module: per-channel weight
    scales from the weight observer, output scale from the activation observer; the int8 result
    dequantises to the float result within quantisation error."""
    import spconv_amd.pytorch as spconv
    import spconv_amd.pytorch.quantization as spconvq
    from spconv_amd.pytorch.quantization import intrinsic as snni
    from spconv_amd.pytorch.quantization.intrinsic import quantized as snniq
    torch.manual_seed(0)
    x = _scene(cuda, C=32)
    conv = spconv.SubMConv3d(32, 64, 3, bias=True).to(cuda).eval()
    fused = snni.SpconvReLUNd(conv, nn.ReLU()).eval()
    want = fused(x).features
    fused.qconfig = spconvq.get_default_spconv_trt_ptq_qconfig()
    fused.activation_post_process = fused.qconfig.activation().to(cuda)
    fused.activation_post_process(want)
    q = snniq.SparseConvReLU.from_float(fused)
    assert q.weight().dtype == torch.qint8 and q.weight().q_per_channel_scales().shape == (64,)
    in_scale = float(x.features.abs().max()) / 127
    xq = spconvq.quantize_per_tensor(x, in_scale, 0, torch.qint8)
    out = q(xq)
    assert out.features.dtype == torch.qint8 and abs(out.features.q_scale() - q.scale) < 1e-12
    got = out.features.dequantize()
    clipped = want.clamp(max=127 * q.scale)        # the histogram observer trades range for resolution
    assert float((got - clipped).abs().max() / want.abs().max()) < 0.05
    # state dict round trip of the quantised module
    q2 = snniq.SparseConvReLU(3, 32, 64, 3, subm=True)
    q2.load_state_dict(q.state_dict())
    assert q2.scale == q.scale and torch.equal(q2.weight().int_repr().cpu(), q.weight().int_repr().cpu())


@pytest.mark.filterwarnings("ignore")
def test_fx_ptq_and_qat_flow(cuda):
    """prepare_fx -> calibrate -> convert_fx -> transform_qdq -> remove_conv_add_dq (and the QAT
    variant) on a small residual 2-d net: every sparse layer ends as an int8 module and the int8
    network tracks the float / fake-quantised one."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fx_quant_demo
    res = fx_quant_demo.main()
    assert res["ptq_modules"] == ["SparseConvAddReLU", "SparseConvReLU"]
    assert res["ptq_rel_err"] < 0.2
    assert res["qat_rel_err_vs_fakequant"] < 0.12
    assert np.isfinite(res["qat_loss"])
