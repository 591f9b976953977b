"""Shared helpers for the parity tests: scene generation and oracle/GPU adapters."""
import numpy as np
import torch

import oracle
from spconv_amd.utils import synthetic


def scene(shape, n, bs=1, seed=0):
    return synthetic.uniform_scene(shape, n, bs, seed)


def dense_scene(shape, n, bs=1, seed=0):
    """Voxels concentrated in a sub-box so that neighbourhoods are well populated."""
    sub = [max(2, s // 3) for s in shape]
    idx = synthetic.uniform_scene(sub, min(n, int(np.prod(sub)) // 2), bs, seed)
    return idx


def oracle_rulebook(idx, bs, shape, ksize, stride, padding, dilation, subm, transpose=False,
                    out_padding=None):
    out_inds, pair, num, out_shape = oracle.get_indice_pairs(
        idx, bs, shape, ksize, stride, padding, dilation, out_padding, subm, transpose)
    n_in, n_out = idx.shape[0], out_inds.shape[0]
    fwd, bwd, mfwd, mbwd = oracle.dense_tables(pair, num, n_in, n_out, subm)
    return dict(out_inds=out_inds, pair=pair, num=num, out_shape=out_shape, fwd=fwd, bwd=bwd,
                mfwd=mfwd, mbwd=mbwd, n_in=n_in, n_out=n_out)


def gpu_rulebook(idx, bs, shape, ksize, stride, padding, dilation, subm, transpose=False,
                 out_padding=None, dev="cuda:0", **kw):
    from spconv_amd.pytorch import ops
    ndim = len(shape)
    t = torch.from_numpy(idx).to(dev)
    rb, out_shape = ops.build_rulebook(t, bs, list(shape), list(ksize), list(stride),
                                       list(padding), list(dilation),
                                       list(out_padding or [0] * ndim), subm, transpose, **kw)
    torch.cuda.synchronize()
    return rb, out_shape


def to_np(t):
    return None if t is None else t.detach().cpu().numpy()


def match_rows(got_idx, ref_idx, shape):
    """perm with got_idx == ref_idx[perm]: the two hold the same coordinates (asserted), possibly in another row order
    (a strided layer in sorted order against the oracle's first-seen numbering)."""
    got_idx, ref_idx = np.asarray(got_idx), np.asarray(ref_idx)
    assert got_idx.shape == ref_idx.shape, (got_idx.shape, ref_idx.shape)
    if np.array_equal(got_idx, ref_idx):
        return np.arange(got_idx.shape[0])
    key = lambda a: np.ravel_multi_index(tuple(a[:, 1 + d].astype(np.int64) for d in range(len(shape))), shape) \
        + a[:, 0].astype(np.int64) * int(np.prod(shape))
    kg, kr = key(got_idx), key(ref_idx)
    order = np.argsort(kr, kind="stable")
    pos = np.searchsorted(kr[order], kg)
    assert pos.max() < kr.shape[0] and np.array_equal(kr[order][pos], kg), "different coordinate sets"
    assert np.unique(kg).shape[0] == kg.shape[0]
    return order[pos]


def assert_rulebook_equal(rb, ref, subm, check_bwd=True):
    """Bit-exact comparison of every artefact against the oracle."""
    assert rb.n_out == ref["n_out"], (rb.n_out, ref["n_out"])
    np.testing.assert_array_equal(to_np(rb.out_indices), ref["out_inds"])
    np.testing.assert_array_equal(to_np(rb.num_per_loc), ref["num"])
    np.testing.assert_array_equal(to_np(rb.pair_native), ref["pair"])
    np.testing.assert_array_equal(to_np(rb.pair_fwd), ref["fwd"])
    np.testing.assert_array_equal(to_np(rb.mask_fwd).view(np.uint32), ref["mfwd"])
    if check_bwd and rb.pair_bwd is not None:
        np.testing.assert_array_equal(to_np(rb.pair_bwd), ref["bwd"])
    if not subm:
        np.testing.assert_array_equal(to_np(rb.mask_bwd).view(np.uint32), ref["mbwd"])


def rel_err(a, ref):
    """NORM-WISE relative error: max |a - ref| / max |ref|.  This is the bound a GEMM-shaped
    kernel can promise (an output element that is a cancelling sum of large terms has an error
    relative to the terms, not to itself); the element-wise statement north_star makes for fp32
    ("within 1e-3 rel") is checked by `assert_close_elementwise` below with an absolute floor."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-12))


HALF_ULP = {"float16": 2.0 ** -11, "bfloat16": 2.0 ** -8, "float32": 2.0 ** -24}


def assert_close_abs_sum(a, ref, abs_sum, dtype, c=1e-6, name=""):
    """Element-wise bound with no free floor: |a - ref| <= u |ref| + c A, where u is half an ulp of the
    OUTPUT dtype (the one rounding the kernel adds on top of its fp32 accumulator) and A is the same sum
    with every operand replaced by its magnitude (the oracle run on |f|, |w|, |dout|) -- the quantity
    fp32 accumulation error is relative to.  c: accumulation in an unknown order over n terms errs by about
    sqrt(n) 2^-24 A (n <= 27 * 64: 2.5e-6 A); measured on MI355X at the full-size configs 2 / 2b the
    excess over u |ref| is <= 2.2e-8 A for every element of out, din and dW in fp16 and bf16
    (tools/tol_probe.py), so c = 1e-6 leaves a factor 50 and still holds a cancelling element to ~1e-6 of
    its terms instead of to a fraction of the tensor's rms."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    A = np.asarray(abs_sum, dtype=np.float64)
    assert a.shape == ref.shape == A.shape, (a.shape, ref.shape, A.shape)
    if ref.size == 0:
        return
    u = HALF_ULP[str(dtype).replace("torch.", "")]
    bound = u * np.abs(ref) * (1 + 1e-6) + c * A + 1e-30
    bad = np.abs(a - ref) > bound
    if bad.any():
        i = np.unravel_index(np.argmax((np.abs(a - ref) - bound) * bad), a.shape)
        raise AssertionError(f"{name}: {int(bad.sum())} of {a.size} elements outside u|ref| + {c:g} A; worst at {i}: "
                             f"got {a[i]:.8g}, want {ref[i]:.8g}, A {A[i]:.6g}")


def assert_close_elementwise(a, ref, rtol, floor_frac=None, name=""):
    """Element-wise |a - ref| <= rtol * |ref| + floor, floor = floor_frac * rms(ref): every
    element within rtol of ITS OWN reference value, except that elements much smaller than the
    tensor's typical magnitude (cancelling sums) are held to an absolute floor instead.
    floor_frac defaults to rtol (i.e. the floor is the error allowed on a typical element)."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    if ref.size == 0:
        return
    rms = float(np.sqrt(np.mean(ref * ref)))
    floor = (rtol if floor_frac is None else floor_frac) * max(rms, 1e-30)
    bad = np.abs(a - ref) > rtol * np.abs(ref) + 4.0 * floor
    if bad.any():
        i = np.unravel_index(np.argmax(np.abs(a - ref) * bad), a.shape)
        raise AssertionError(f"{name}: {int(bad.sum())} of {a.size} elements outside rtol {rtol:g} "
                             f"(+ floor {4 * floor:.3g}); worst at {i}: got {a[i]:.6g}, want {ref[i]:.6g}")
