"""Golden rulebooks produced by EXECUTING THE REFERENCE'S OWN CPU CODE.  Run here, where
/root/reference exists (the GPU box and CI only read the committed vectors):

    make -C oracle ref && PYTHONPATH=. python tests/golden/make_ref_golden.py

`oracle/_ref/libspconv_ref.so` is the reference's `SparseConvIndicesCPU.generate_subm_conv_inds` /
`generate_conv_inds` + `ConvOutLocIter` (csrc/sparse/indices.py:76-269,1620-1778) rendered from the
reference source where it lies (oracle/refbuild/render.py) -- so these vectors pin the RULEBOOK ORDER
(pair lists, counts, first-seen output numbering) to spconv itself, not to our restatement:

* ref_<case>.npz       small seeded problems: inputs + (out_inds, pair, num) as the reference wrote
                       them (1-d .. 4-d, SubM with duplicates and deleted rows, strided, dilated,
                       transposed, the (3,1,1)/(2,1,1) and pad (0,1,1) layers of VoxelBackBone8x);
* ref_digests.json     SHA-256 of (out_inds, pair, num) for the BASELINE-size problems whose tables
                       are too large to commit: config 1 / 2 (seeded uniform scenes) and the real-LiDAR
                       fixture (tests/golden/lidar_scene.npz) as SubM and as the stride-2 chain of
                       config 3 -- a checksum per artefact, compared by tests on CPU (oracle) and on
                       the GPU (HIP builder).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref  # noqa: E402
from spconv_amd.utils import synthetic  # noqa: E402

SMALL = {
    # name: (shape, n, bs, ksize, stride, pad, dil, subm, transposed, with_quirks)
    "subm_1d": ([200], 60, 2, [3], [1], [1], [1], True, False, False),
    "subm_2d_d2": ([40, 40], 300, 1, [3, 3], [1, 1], [2, 2], [2, 2], True, False, False),
    "subm_3d_dups_deleted": ([16, 16, 16], 300, 2, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True, False, True),
    "subm_3d_k533": ([12, 12, 12], 250, 1, [5, 3, 3], [1] * 3, [2, 1, 1], [1] * 3, True, False, False),
    "subm_4d": ([5, 6, 7, 8], 200, 2, [3] * 4, [1] * 4, [1] * 4, [1] * 4, True, False, False),
    "conv_3d_k3s2": ([16, 16, 16], 300, 2, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False, False, False),
    "conv_3d_k2s2": ([16, 16, 16], 300, 1, [2] * 3, [2] * 3, [0] * 3, [1] * 3, False, False, False),
    "conv_3d_k3s1d2": ([14, 13, 12], 250, 1, [3] * 3, [1] * 3, [0] * 3, [2] * 3, False, False, False),
    "conv_3d_k3s2d2p2": ([32, 32, 32], 60, 1, [3] * 3, [2] * 3, [2] * 3, [2] * 3, False, False, False),
    "conv_3d_k311s211": ([21, 16, 16], 400, 2, [3, 1, 1], [2, 1, 1], [0] * 3, [1] * 3, False, False, False),
    "conv_3d_p011": ([21, 16, 16], 400, 2, [3] * 3, [2] * 3, [0, 1, 1], [1] * 3, False, False, False),
    "conv_2d_k3s2": ([30, 25], 200, 2, [3, 3], [2, 2], [1, 1], [1, 1], False, False, False),
    "deconv_3d_k3s2": ([8, 8, 8], 120, 1, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False, True, False),
    "deconv_3d_k2s2": ([8, 8, 8], 120, 2, [2] * 3, [2] * 3, [0] * 3, [1] * 3, False, True, False),
}


def small_inputs(name, spec):
    shape, n, bs, ksize, stride, pad, dil, subm, transposed, quirks = spec
    idx = synthetic.uniform_scene(shape, n, bs, seed=sum(map(ord, name)))
    if quirks:      # duplicated coordinates (first wins) and rows outside [0, batch) ("deleted" points)
        extra = np.array([[-1, 1, 1, 1], [bs, 2, 2, 2]], dtype=np.int32)
        idx = np.ascontiguousarray(np.concatenate([idx, idx[5:12], extra, idx[:3]]))
    return idx


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def big_cases():
    """(name, indices, batch, shape, ksize, stride, pad, dil, subm) at BASELINE sizes."""
    from golden import lidar_scene
    out = []
    c1 = synthetic.uniform_scene([64, 64, 64], 5000, 1, seed=0)
    out.append(("cfg1_subm", c1, 1, [64, 64, 64], [3] * 3, [1] * 3, [1] * 3, [1] * 3, True))
    c2 = synthetic.uniform_scene([40, 1280, 1600], 100_000, 1, seed=0)
    out.append(("cfg2_subm", c2, 1, [40, 1280, 1600], [3] * 3, [1] * 3, [1] * 3, [1] * 3, True))
    idx, shape = lidar_scene()
    out.append(("fixture_subm", idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True))
    cur, cur_shape = idx, shape
    for level in range(3):                                  # config 3: k3 s2 p1 chain
        out.append((f"fixture_chain_l{level}", cur, 1, cur_shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False))
        cur, _, _, cur_shape = ref.get_indice_pairs(cur, 1, cur_shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3,
                                                    None, False, False)
    return out


def main():
    assert ref.build() is not None, "oracle/_ref could not be built (needs /root/reference)"
    for name, spec in SMALL.items():
        shape, n, bs, ksize, stride, pad, dil, subm, transposed, _ = spec
        idx = small_inputs(name, spec)
        out_inds, pair, num, out_shape = ref.get_indice_pairs(idx, bs, shape, ksize, stride, pad, dil, None,
                                                              subm, transposed)
        np.savez_compressed(os.path.join(HERE, f"ref_{name}.npz"), indices=idx, bs=bs, shape=np.array(shape),
                            ksize=np.array(ksize), stride=np.array(stride), pad=np.array(pad),
                            dil=np.array(dil), subm=subm, transposed=transposed, out_inds=out_inds,
                            pair=pair, num=num, out_shape=np.array(out_shape))
    digests = {}
    for name, idx, bs, shape, ksize, stride, pad, dil, subm in big_cases():
        out_inds, pair, num, out_shape = ref.get_indice_pairs(idx, bs, shape, ksize, stride, pad, dil, None,
                                                              subm, False)
        digests[name] = {"n_in": int(idx.shape[0]), "n_out": int(out_inds.shape[0]),
                         "pairs": int(num.sum()), "out_shape": [int(v) for v in out_shape],
                         "input": digest(idx), "out_inds": digest(out_inds), "pair": digest(pair),
                         "num": digest(num)}
    with open(os.path.join(HERE, "ref_digests.json"), "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)
    for f in sorted(os.listdir(HERE)):
        if f.startswith("ref_"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
