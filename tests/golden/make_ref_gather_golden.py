#!/usr/bin/env python
"""Golden vectors of the reference's OWN row gather / scatter-add (GatherCPU, spconv/csrc/sparse/gather.py:30-86),
executed here through oracle/_ref (rendered from the reference's source where it lies; needs /root/reference).

    python tests/golden/make_ref_gather_golden.py     ->  tests/golden/gather_ref.npz

Contents: a seeded source matrix, an index list WITH REPEATS (the scatter-add must accumulate in list
order), the gathered rows and the accumulator after the scatter-add, as the reference's code produced
them.  tests/test_oracle.py checks oracle.cpp's restatement against them bit for bit on any box."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref  # noqa: E402


def main():
    assert ref.build() is not None, "oracle/_ref needs /root/reference"
    rng = np.random.default_rng(86)
    n_src, n_dst, nhot, C = 211, 97, 400, 13
    src = rng.uniform(-1, 1, (n_src, C)).astype(np.float32)
    gi = rng.integers(0, n_src, nhot).astype(np.int32)
    so = rng.integers(0, n_dst, nhot).astype(np.int32)            # repeats: ~4 adds per destination row
    gathered = np.full((nhot, C), np.nan, dtype=np.float32)
    ref.gather(gathered, src, gi)
    acc = rng.uniform(-1, 1, (n_dst, C)).astype(np.float32)
    acc0 = acc.copy()
    ref.scatter_add(acc, gathered, so)
    np.savez_compressed(os.path.join(HERE, "gather_ref.npz"), src=src, gather_inds=gi, scatter_inds=so,
                        gathered=gathered, acc_before=acc0, acc_after=acc)
    print("wrote gather_ref.npz", os.path.getsize(os.path.join(HERE, "gather_ref.npz")))


if __name__ == "__main__":
    main()
