#!/usr/bin/env python
"""How tight can the element-wise parity bound be?  For the full-size config 2 / 2b layers: the largest
(|got - ref| - u |ref|) / A over the elements, A = the same sum with every operand replaced by its magnitude
(oracle on |f|, |w|, |dout|), u = half an ulp of the output dtype."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from util import oracle_rulebook, scene
from test_gpu_fused_bwd import _fused_subm
from golden import lidar_scene

dev = torch.device("cuda:0")
res = {}
for tag, (idx, shape) in (("cfg2", (scene([40, 1280, 1600], 100_000, 1, 0), [40, 1280, 1600])), ("cfg2b", lidar_scene())):
    for dtype, u in ((torch.float16, 2.0 ** -11), (torch.bfloat16, 2.0 ** -8)):
        rb, (f, w, dout), got = _fused_subm(dev, idx, shape, 64, 64, dtype, seed=0)
        ref = oracle_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
        o = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"], subm=True)
        di, dw = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"], subm=True)
        oa = oracle.indice_conv(f.abs(), w.abs(), ref["pair"], ref["num"], ref["n_out"], subm=True)
        dia, dwa = oracle.indice_conv_backward(f.abs(), w.abs(), dout.abs(), ref["pair"], ref["num"], subm=True)
        for name, g, r, a in zip(("out", "din", "dw"), got, (o, di, dw), (oa, dia, dwa)):
            g = g.float().cpu().double().numpy(); r = r.double().numpy(); a = a.double().numpy()
            err = np.abs(g - r)
            excess = np.maximum(err - u * np.abs(r), 0) / np.maximum(a, 1e-30)
            res[f"{tag} {str(dtype)[6:]} {name}"] = dict(max_excess_over_abs_sum=float(excess.max()),
                                                         max_err_over_abs_sum=float((err / np.maximum(a, 1e-30)).max()),
                                                         A_over_rms=float(np.median(a) / np.sqrt((r * r).mean())))
print(json.dumps(res, indent=0))
