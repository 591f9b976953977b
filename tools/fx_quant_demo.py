#!/usr/bin/env python
"""End-to-end int8 post-training quantization (and a QAT round) of a small sparse 2-d network
through torch.ao fx graph mode with the spconv_amd backend config -- the flow of the reference's
example/mnist/mnist_ptq.py / mnist_qat.py on random data.  Prints the error of the int8 network
against the float one."""
import os
import sys

os.environ.setdefault("SPCONV_FX_TRACE_MODE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy
import warnings

import numpy as np
import torch
import torch.ao.quantization.quantize_fx as qfx
import torch.nn as nn
from torch.ao.quantization import DeQuantStub, QuantStub

import spconv_amd.pytorch as spconv
import spconv_amd.pytorch.quantization as spconvq

warnings.filterwarnings("ignore")


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
    def __init__(self, cin=16, c=32):
        super().__init__()
        self.net = spconv.SparseSequential(
            spconv.SubMConv2d(cin, c, 3, bias=False, indice_key="a"), nn.BatchNorm1d(c), nn.ReLU(),
            spconv.SparseConv2d(c, c, 2, 2, bias=False), nn.BatchNorm1d(c), nn.ReLU(),
            Res(c), spconv.ToDense())
        self.quant = QuantStub()
        self.dequant = DeQuantStub()

    def forward(self, features, indices, batch_size: int):
        x = spconv.SparseConvTensor(self.quant(features), indices, [28, 28], batch_size)
        return self.dequant(torch.flatten(self.net(x), 1))


def batch(seed, dev, bs=4, cin=16):
    rng = np.random.default_rng(seed)
    rows = []
    for b in range(bs):
        cells = rng.choice(28 * 28, 300, replace=False)
        rows.append(np.stack([np.full_like(cells, b), cells // 28, cells % 28], 1))
    idx = torch.from_numpy(np.concatenate(rows).astype(np.int32)).to(dev)
    g = torch.Generator().manual_seed(seed)
    return torch.randn(idx.shape[0], cin, generator=g).to(dev), idx, bs


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = Net().to(dev)
    model.train()
    with torch.no_grad():                       # give the BatchNorms non-trivial statistics
        for s in range(4):
            model(*batch(100 + s, dev))
    model.eval()
    ref = model(*batch(0, dev))
    backend = spconvq.get_spconv_backend_config()
    prep_cfg = spconvq.get_spconv_prepare_custom_config()
    spconvq.prepare_spconv_torch_inference(False)
    res = {}
    # ---- PTQ
    prepared = qfx.prepare_fx(copy.deepcopy(model), spconvq.get_default_spconv_qconfig_mapping(False), (),
                              backend_config=backend, prepare_custom_config=prep_cfg)
    with torch.no_grad():
        for s in range(8):
            prepared(*batch(s, dev))
    converted = qfx.convert_fx(prepared, backend_config=backend)
    converted = spconvq.remove_conv_add_dq(spconvq.transform_qdq(converted))
    kinds = sorted({type(m).__name__ for m in converted.modules()})
    with torch.no_grad():
        out = converted(*batch(0, dev))
    res["ptq_rel_err"] = float((out - ref).abs().max() / ref.abs().max())
    res["ptq_modules"] = [k for k in kinds if "Sparse" in k or "Quantized" in k]
    # ---- QAT
    qat_model = copy.deepcopy(model).train()
    prepared = qfx.prepare_qat_fx(qat_model, spconvq.get_default_spconv_qconfig_mapping(True), (),
                                  backend_config=backend, prepare_custom_config=prep_cfg)
    opt = torch.optim.SGD(prepared.parameters(), lr=1e-3)
    for s in range(6):
        opt.zero_grad()
        loss = prepared(*batch(s, dev)).square().mean()
        loss.backward()
        opt.step()
    res["qat_loss"] = float(loss)
    prepared.eval()
    with torch.no_grad():
        ref_q = prepared(*batch(0, dev))
    converted = qfx.convert_fx(prepared, backend_config=backend)
    converted = spconvq.remove_conv_add_dq(spconvq.transform_qdq(converted))
    with torch.no_grad():
        out = converted(*batch(0, dev))
    res["qat_rel_err_vs_fakequant"] = float((out - ref_q).abs().max() / ref_q.abs().max())
    print(res)
    return res


if __name__ == "__main__":
    main()
