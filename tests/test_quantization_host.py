"""CPU: the quantization tooling around the int8 kernel -- folding
formulas, fused containers, swap tables, and the torch.ao fx backend description up to
``prepare_fx`` (no sparse kernel runs)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import spconv_amd.pytorch as spconv
import spconv_amd.pytorch.quantization as spconvq
from spconv_amd.pytorch.quantization import intrinsic as snni
from spconv_amd.pytorch.quantization.core import quantize_per_tensor, quantized_add
from spconv_amd.pytorch.quantization.intrinsic import qat as snniqat
from spconv_amd.pytorch.quantization.intrinsic import quantized as snniq
from spconv_amd.pytorch.quantization.utils import fuse_spconv_bn_eval, fuse_spconv_bn_weights


def test_bn_folding_formula_krsc():
    """w'[k] = w[k] * gamma[k] / sqrt(var[k] + eps), b' = (b - mean) * gamma / sqrt(var + eps) + beta
    (reference quantization/utils.py:5-24), on the KRSC layout (output channel = dim 0)."""
    g = torch.Generator().manual_seed(0)
    K, C = 6, 5
    w = torch.randn(K, 3, 3, 3, C, generator=g)
    b = torch.randn(K, generator=g)
    mean, var = torch.randn(K, generator=g), torch.rand(K, generator=g) + 0.5
    gamma, beta = torch.randn(K, generator=g), torch.randn(K, generator=g)
    fw, fb = fuse_spconv_bn_weights(w, b, mean, var, 1e-3, gamma, beta)
    gain = (gamma / torch.sqrt(var + 1e-3)).numpy()
    np.testing.assert_allclose(fw.detach().numpy(), w.numpy() * gain[:, None, None, None, None], rtol=1e-6)
    np.testing.assert_allclose(fb.detach().numpy(), (b - mean).numpy() * gain + beta.numpy(), rtol=1e-5, atol=1e-6)
    # no conv bias / no affine
    fw2, fb2 = fuse_spconv_bn_weights(w, None, mean, var, 1e-3, None, None)
    np.testing.assert_allclose(fb2.detach().numpy(), (-mean / torch.sqrt(var + 1e-3)).numpy(), rtol=1e-5, atol=1e-6)
    conv = spconv.SubMConv3d(C, K, 3).train()
    with pytest.raises(AssertionError, match="eval"):
        fuse_spconv_bn_eval(conv, nn.BatchNorm1d(K))


def test_fused_containers_and_swap_tables():
    conv, bn, relu = spconv.SubMConv3d(4, 8, 3), nn.BatchNorm1d(8), nn.ReLU()
    assert len(snni.SpconvBnReLUNd(conv, bn, relu)) == 3
    with pytest.raises(AssertionError, match="Incorrect types"):
        snni.SpconvReLUNd(conv, bn)
    with pytest.raises(AssertionError, match="Incorrect types"):
        snni.SpconvBnNd(nn.Linear(2, 2), bn)
    qat_map = spconvq.get_spconv_fmod_to_qat_mapping()
    static_map = spconvq.get_spconv_qat_to_static_mapping()
    assert qat_map[snni.SpconvBnReLUNd] is snniqat.SparseConvBnReLU
    assert qat_map[snni.SpconvBnAddReLUNd] is snniqat.SparseConvBnAddReLU
    assert static_map[spconv.SubMConv3d] is spconvq.SparseConv
    assert static_map[snniqat.SparseConvBnReLU] is snniq.SparseConvReLU
    assert static_map[snni.SpconvReLUNd] is snniq.SparseConvReLU


def test_qat_module_from_float_and_state_dict_versions():
    qconfig = spconvq.get_default_spconv_trt_qat_qconfig()
    conv, bn = spconv.SubMConv3d(4, 8, 3, bias=False, indice_key="k"), nn.BatchNorm1d(8, eps=1e-3, momentum=0.01)
    fused = snni.SpconvBnReLUNd(conv, bn, nn.ReLU())
    fused.qconfig = qconfig
    qat = snniqat.SparseConvBnReLU.from_float(fused)
    assert qat.weight is conv.weight and qat.bn.weight is bn.weight and qat.bn.eps == 1e-3
    assert qat.subm and qat.indice_key == "k" and qat.bias is None
    # a frozen BatchNorm stays in eval mode under .train()
    qat.freeze_bn_stats().train()
    assert qat.training and not qat.bn.training
    qat.update_bn_stats().train()
    assert qat.bn.training
    # version-1 checkpoints carried gamma / beta / running_* on the module itself
    sd = {k: v.clone() for k, v in qat.state_dict().items()}
    old = {k: v for k, v in sd.items() if not k.startswith("bn.")}
    old.update(gamma=sd["bn.weight"] * 2, beta=sd["bn.bias"], running_mean=sd["bn.running_mean"],
               running_var=sd["bn.running_var"], num_batches_tracked=sd["bn.num_batches_tracked"])
    meta = {"": {"version": 1}}
    import collections
    old = collections.OrderedDict(old)
    old._metadata = meta
    qat.load_state_dict(old)
    assert torch.equal(qat.bn.weight, sd["bn.weight"] * 2)
    # back to float: BatchNorm folded, ReLU kept beside the conv
    flt = qat.eval().to_float()
    assert isinstance(flt, snni.SpconvReLUNd) and flt[0].bias is not None
    with pytest.raises(AssertionError, match="qconfig"):
        snniqat.SparseConv.from_float(spconv.SubMConv3d(4, 8, 3))


def test_quantize_helpers_on_sparse_tensors():
    idx = torch.tensor([[0, 1, 1], [0, 2, 3]], dtype=torch.int32)
    x = spconv.SparseConvTensor(torch.tensor([[0.5, -1.0], [2.0, 0.26]]), idx, [4, 4], 1)
    q = quantize_per_tensor(x, 0.25, 0, torch.qint8)
    assert q.is_quantized and q.indices is x.indices and q.q_scale() == 0.25
    assert q.features.int_repr().tolist() == [[2, -4], [8, 1]]
    both = quantize_per_tensor([x, x.features], [0.25, 0.5], [0, 0], torch.qint8)
    assert both[0].is_quantized and both[1].q_scale() == 0.5
    a = torch.quantize_per_tensor(torch.tensor([1.0, -3.0, 100.0]), 0.5, 0, torch.qint8)
    b = torch.quantize_per_tensor(torch.tensor([0.25, 1.0, 30.0]), 0.25, 0, torch.qint8)
    s = quantized_add(a, b, 1.0, 0)
    assert s.q_scale() == 1.0 and s.int_repr().tolist() == [1, -2, 94]      # 63.5 + 30 = 93.5 -> 94 (even)


def _tiny_net():
    class Res(spconv.SparseModule):
        def __init__(self, c):
            super().__init__()
            self.c1 = spconv.SparseSequential(spconv.SubMConv2d(c, c, 3, bias=False, indice_key="r"),
                                              nn.BatchNorm1d(c), nn.ReLU())
            self.c2 = spconv.SparseSequential(spconv.SubMConv2d(c, c, 3, bias=False, indice_key="r"),
                                              nn.BatchNorm1d(c))
            self.relu = spconv.SparseReLU()

        def forward(self, x):
            return self.relu(self.c2(self.c1(x)) + x)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.net = spconv.SparseSequential(
                spconv.SubMConv2d(8, 16, 3, bias=False, indice_key="a"), nn.BatchNorm1d(16), nn.ReLU(),
                spconv.SparseConv2d(16, 16, 2, 2, bias=False), nn.BatchNorm1d(16), Res(16), spconv.ToDense())

        def forward(self, features, indices, batch_size: int):
            return torch.flatten(self.net(spconv.SparseConvTensor(features, indices, [28, 28], batch_size)), 1)
    return Net()


@pytest.mark.filterwarnings("ignore")
@pytest.mark.parametrize("is_qat", [False, True])
def test_prepare_fx_fuses_sparse_patterns(is_qat):
    """conv+bn+relu, conv+bn and relu(conv+bn + residual) are found by the backend config and
    replaced by the intrinsic (PTQ: bn folded) / QAT modules; observers land on every edge."""
    import torch.ao.quantization.quantize_fx as qfx
    m = _tiny_net()
    m.train(is_qat)
    prep = qfx.prepare_qat_fx if is_qat else qfx.prepare_fx
    g = prep(m, spconvq.get_default_spconv_qconfig_mapping(is_qat), (),
             backend_config=spconvq.get_spconv_backend_config(),
             prepare_custom_config=spconvq.get_spconv_prepare_custom_config())
    mods = dict(g.named_modules())
    if is_qat:
        want = {"net.0": snniqat.SparseConvBnReLU, "net.3": snniqat.SparseConvBn,
                "net.5.c1.0": snniqat.SparseConvBnReLU, "net.5.c2.0": snniqat.SparseConvBnAddReLU}
    else:
        want = {"net.0": snni.SpconvReLUNd, "net.3": spconv.SparseConv2d,
                "net.5.c1.0": snni.SpconvReLUNd, "net.5.c2.0": snni.SpconvAddReLUNd}
    for name, cls in want.items():
        assert type(mods[name]) is cls, (name, type(mods[name]))
    if not is_qat:
        assert mods["net.3"].bias is not None                     # BatchNorm folded into the conv
    res = [n for n in g.graph.nodes if n.op == "call_module" and n.target == "net.5.c2.0"][0]
    assert len(res.args) == 2                                     # (input, residual)
    n_obs = sum(1 for n in g.graph.nodes if n.op == "call_module" and "activation_post_process" in str(n.target))
    assert n_obs >= 6
