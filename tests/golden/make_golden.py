"""Generates the committed golden fixtures.  Run here (where /root/reference exists):

    PYTHONPATH=. python tests/golden/make_golden.py

* lidar_scene.npz  -- the voxel COORDINATES of the reference's real-LiDAR test
  fixture /root/reference/test/data/test_spconv.pkl (125 562 voxels in
  [80,1600,1600], used by test/test_multi_impl.py:230-337), sorted and
  delta-encoded (uint32) so the file stays small.  Features are not stored (the
  reference's tests draw random features too).
* case_*.npz -- small seeded problems with the outputs of the CPU oracle
  (rulebook in the reference CPU order, forward, din, dW).  The oracle itself is
  pinned against dense torch conv (tests/test_oracle.py), mirroring
  test/test_conv.py:286-357; the reference cannot be imported here.
"""
import os
import pickle
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from spconv_amd.utils import synthetic  # noqa: E402

REF_PKL = "/root/reference/test/data/test_spconv.pkl"


def make_lidar():
    with open(REF_PKL, "rb") as f:
        _, coors, shape = pickle.load(f)
    assert coors.shape == (125562, 4) and list(shape) == [80, 1600, 1600]
    lin = np.sort(np.ravel_multi_index((coors[:, 1], coors[:, 2], coors[:, 3]), shape))
    delta = np.diff(lin, prepend=0).astype(np.uint32)
    np.savez_compressed(os.path.join(HERE, "lidar_scene.npz"), delta=delta,
                        shape=np.array(shape, dtype=np.int32))


CASES = {
    # name: (shape, n, bs, C, K, ksize, stride, pad, dil, subm, transposed)
    "subm_k3": ([16, 16, 16], 300, 2, 8, 8, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], True, False),
    "conv_k3s2": ([16, 16, 16], 300, 2, 8, 16, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], False, False),
    "conv_k2s2": ([16, 16, 16], 300, 1, 8, 8, [2, 2, 2], [2, 2, 2], [0, 0, 0], [1, 1, 1], False, False),
    "deconv_k3s2": ([8, 8, 8], 120, 1, 8, 8, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], False, True),
    "subm_k3d2_2d": ([40, 40], 300, 1, 8, 8, [3, 3], [1, 1], [2, 2], [2, 2], True, False),
}


def make_case(name, spec):
    shape, n, bs, C, K, ksize, stride, pad, dil, subm, transposed = spec
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
    idx = synthetic.uniform_scene(shape, n, bs, seed=sum(map(ord, name)))
    out_inds, pair, num, out_shape = oracle.get_indice_pairs(idx, bs, shape, ksize, stride, pad, dil,
                                                             None, subm, transposed)
    f = torch.from_numpy(rng.uniform(-1, 1, (idx.shape[0], C)).astype(np.float32))
    w = torch.from_numpy(rng.uniform(-1, 1, (K, *ksize, C)).astype(np.float32))
    dout = torch.from_numpy(rng.uniform(-0.2, 0.2, (out_inds.shape[0], K)).astype(np.float32))
    out = oracle.indice_conv(f, w, pair, num, out_inds.shape[0], subm=subm)
    din, dw = oracle.indice_conv_backward(f, w, dout, pair, num, subm=subm)
    np.savez_compressed(
        os.path.join(HERE, f"case_{name}.npz"), indices=idx, features=f.numpy(), weight=w.numpy(),
        dout=dout.numpy(), out_inds=out_inds, pair=pair, num=num, out=out.numpy(), din=din.numpy(),
        dw=dw.numpy(), shape=np.array(shape), bs=bs, ksize=np.array(ksize), stride=np.array(stride),
        pad=np.array(pad), dil=np.array(dil), subm=subm, transposed=transposed,
        out_shape=np.array(out_shape))


if __name__ == "__main__":
    make_lidar()
    for name, spec in CASES.items():
        make_case(name, spec)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
