import sys, copy
sys.path.insert(0, "tests")
import numpy as np, torch
import spconv_amd.pytorch as spconv
from spconv_amd.pytorch import prefetch
from spconv_amd.pytorch.static import RulebookPipeline, strided_layers
from test_gpu_static import _backbone, _scene_tensors
cuda = torch.device("cuda:0")
shape, bs, C = [32, 40, 40], 2, 4
net = _backbone(spconv, C, cuda, torch.float16, False)
names = list(strided_layers(net))
eager = copy.deepcopy(net)
pipe = RulebookPipeline(net, max_voxels=12_000, in_channels=C, spatial_shape=shape, batch_size=bs,
                        dtype=torch.float16, bounds={names[0]: 13_000, names[1]: 1_700})
scenes = [_scene_tensors(shape, n, bs, C, seed, cuda, torch.float16)
          for n, seed in ((4500, 1), (2001, 2), (5999, 3), (388, 4), (3000, 5))]
mode = sys.argv[1]
want = []
with torch.no_grad():
    for f, idx in scenes:
        want.append(eager(spconv.SparseConvTensor(f, idx, shape, bs)))
pipe.submit(*scenes[0])
for k in range(len(scenes)):
    if k + 1 < len(scenes):
        pipe.submit(*scenes[k + 1])
    if mode == "sync":
        torch.cuda.synchronize()
    got = pipe.run()
    c = pipe.counts()
    n_live = want[k].indices.shape[0]
    print(k, c, n_live, torch.equal(got.indices[:n_live], want[k].indices), torch.equal(got.features[:n_live], want[k].features))
