import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import netbench
import spconv_amd.pytorch as spconv
from spconv_amd.utils import synthetic
dev = torch.device("cuda:0")
idx = torch.from_numpy(synthetic.lidar_like_scene(netbench.SHAPE, 100_000, 1, seed=0)).to(dev)
n = idx.shape[0]
net = netbench.backbone(4).to(dev).half().eval()
f4 = torch.randn(n, 4, device=dev).half()
def infer():
    with torch.no_grad():
        net(spconv.SparseConvTensor(f4, idx, netbench.SHAPE, 1))
for _ in range(5): infer()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): infer()
pr.disable(); torch.cuda.synchronize()
sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(45)
print("\n".join(l[:160] for l in sio.getvalue().splitlines()[4:60]))
