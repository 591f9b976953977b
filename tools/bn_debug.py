import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from spconv_amd.pytorch import norm
dev = torch.device("cuda:0")
for dtype in (torch.float32, torch.float16):
    for n, C in ((1000, 16), (300, 256)):
        torch.manual_seed(0)
        x = (torch.randn(n, C, device=dev) * 1.7 + torch.linspace(-3, 3, C, device=dev)).to(dtype)
        bn = nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(dev)
        ref = copy.deepcopy(bn)
        y = norm.batch_norm(x.clone().requires_grad_(True), bn)
        yr = ref(x.float())
        d = (y.float() - yr).abs()
        print(dtype, n, C, "max diff", float(d.max()), "per-channel", d.max(0)[0][:8].tolist(), "rows", d.max(1)[0][:6].tolist())
        print("  run mean ours", bn.running_mean[:4].tolist(), "ref", ref.running_mean[:4].tolist())
