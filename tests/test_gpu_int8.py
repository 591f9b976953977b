"""GPU parity of the int8 inference path (BASELINE config 5)
against the numpy restatement of the reference's formula (oracle.int8_conv_ref, following
test/test_all_algo.py:222-288): integer accumulation is exact, the epilogue is evaluated in
fp32 in the same operation order, so int8 outputs must match BIT-EXACTLY and float outputs
to fp32 rounding (the reference's own tolerance for this path is 1e-4)."""
import numpy as np
import pytest
import torch

import oracle
from util import dense_scene, gpu_rulebook, oracle_rulebook, scene, to_np

pytestmark = pytest.mark.gpu


def _int8_case(shape, n, bs, C, K, ksize, stride, pad, dil, subm, seed, dense=True, wide=False):
    rng = np.random.default_rng(seed)
    idx = dense_scene(shape, n, bs, seed) if dense else scene(shape, n, bs, seed)
    ref = oracle_rulebook(idx, bs, shape, ksize, stride, pad, dil, subm)
    lo, hi = (-127, 128) if wide else (-8, 9)
    f = rng.integers(lo, hi, (idx.shape[0], C), dtype=np.int8)
    w = rng.integers(lo, hi, (K, *ksize, C), dtype=np.int8)
    # scale chosen so that the int8 range is actually exercised (incl. clipping at both ends)
    kvol = int(np.prod(ksize))
    mag = (hi * hi / 3.0) * np.sqrt(C * max(1.0, 0.25 * kvol))
    scale = (rng.uniform(0.5, 1.5, K) * 60.0 / mag).astype(np.float32)
    bias = rng.uniform(-5, 5, K).astype(np.float32)
    add = rng.integers(-127, 128, (ref["n_out"], K), dtype=np.int8)
    return idx, ref, f, w, scale, bias, add


INT8_CASES = [
    # shape, n, bs, C, K, ksize, stride, pad, dil, subm
    ([19, 18, 17], 1500, 1, 16, 16, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True),    # reference test shape
    ([24, 24, 24], 2500, 2, 64, 64, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True),
    ([24, 24, 24], 2500, 1, 128, 128, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True),  # cfg 5 channels
    ([24, 24, 24], 2000, 1, 32, 64, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False),
    ([24, 24, 24], 2000, 1, 48, 32, [2] * 3, [2] * 3, [0] * 3, [1] * 3, False),   # C not multiple of 128 B
    ([24, 24, 24], 1200, 1, 144, 256, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True),  # two reduction pieces
    # channel counts the kernel is not instantiated for: zero-padded by the host side (a 4-channel first
    # layer, 48 / 96-wide layers)
    ([24, 24, 24], 2500, 2, 4, 16, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True),
    ([24, 24, 24], 2000, 1, 40, 48, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False),
    ([24, 24, 24], 2000, 1, 64, 96, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True),
]


@pytest.mark.parametrize("with_add,relu", [(False, False), (True, True)])
@pytest.mark.parametrize("shape,n,bs,C,K,ksize,stride,pad,dil,subm", INT8_CASES)
def test_int8_forward_bit_exact(cuda, shape, n, bs, C, K, ksize, stride, pad, dil, subm, with_add, relu):
    from spconv_amd.pytorch import ops
    idx, ref, f, w, scale, bias, add = _int8_case(shape, n, bs, C, K, ksize, stride, pad, dil, subm, seed=3)
    rb, _ = gpu_rulebook(idx, bs, shape, ksize, stride, pad, dil, subm)
    kv = rb.kv
    add_scale = 0.37 if with_add else 0.0
    want = oracle.int8_conv_ref(f, w, ref["pair"], ref["num"], ref["n_out"], subm, scale, bias,
                                add if with_add else None, add_scale, relu)
    got = ops.igemm_fwd_int8(torch.from_numpy(f).to(cuda), torch.from_numpy(w).to(cuda), rb.pair_fwd,
                             rb.mask_fwd, None, rb.n_out, kv // 2 if subm else -1,
                             torch.from_numpy(scale), torch.from_numpy(bias),
                             torch.from_numpy(add).to(cuda) if with_add else None, add_scale,
                             torch.int8, ops.Activation.ReLU if relu else ops.Activation.None_)
    got = to_np(got)
    assert got.dtype == np.int8
    # make sure the case exercises rounding AND clipping
    assert np.abs(want.astype(np.int32)).max() == 127 or want.min() == -128
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("out_dtype,tol", [(torch.float32, 1e-6), (torch.float16, 1e-3)])
def test_int8_forward_float_output(cuda, out_dtype, tol):
    from spconv_amd.pytorch import ops
    shape, n, bs, C, K = [24, 24, 24], 2000, 1, 64, 64
    ks, st, pd, dl = [3] * 3, [1] * 3, [1] * 3, [1] * 3
    idx, ref, f, w, scale, bias, add = _int8_case(shape, n, bs, C, K, ks, st, pd, dl, True, seed=5)
    rb, _ = gpu_rulebook(idx, bs, shape, ks, st, pd, dl, True)
    want = oracle.int8_conv_ref(f, w, ref["pair"], ref["num"], ref["n_out"], True, scale, bias, add,
                                0.11, True, out_dtype=np.float32)
    got = ops.igemm_fwd_int8(torch.from_numpy(f).to(cuda), torch.from_numpy(w).to(cuda), rb.pair_fwd,
                             rb.mask_fwd, None, rb.n_out, rb.kv // 2, torch.from_numpy(scale),
                             torch.from_numpy(bias), torch.from_numpy(add).to(cuda), 0.11, out_dtype,
                             ops.Activation.ReLU)
    got = to_np(got.float())
    assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max())


def test_int8_mask_sorted_rows_and_argsort(cuda):
    from spconv_amd.pytorch import ops
    shape, n, bs, C, K = [24, 24, 24], 2500, 1, 32, 32
    ks, st, pd, dl = [3] * 3, [1] * 3, [1] * 3, [1] * 3
    idx, ref, f, w, scale, bias, add = _int8_case(shape, n, bs, C, K, ks, st, pd, dl, True, seed=7)
    rb, _ = gpu_rulebook(idx, bs, shape, ks, st, pd, dl, True, do_sort=True)
    want = oracle.int8_conv_ref(f, w, ref["pair"], ref["num"], ref["n_out"], True, scale, bias)
    got = ops.igemm_fwd_int8(torch.from_numpy(f).to(cuda), torch.from_numpy(w).to(cuda), rb.pair_fwd,
                             rb.mask_fwd, rb.argsort_fwd, rb.n_out, rb.kv // 2,
                             torch.from_numpy(scale), torch.from_numpy(bias))
    np.testing.assert_array_equal(to_np(got), want)
    # the same through the tables in tile order (SPX_TILE_ORDER: row t of pair / mask belongs to output
    # row argsort[t]), as ops.tables_of hands them out for a sorted rulebook
    pair_t, mask_t, order, to = ops.tables_of(rb, "fwd", K)
    assert to and order is rb.argsort_fwd and pair_t is not rb.pair_fwd
    got_t = ops.igemm_fwd_int8(torch.from_numpy(f).to(cuda), torch.from_numpy(w).to(cuda), pair_t, mask_t, order,
                               rb.n_out, rb.kv // 2, torch.from_numpy(scale), torch.from_numpy(bias),
                               tile_order=True)
    np.testing.assert_array_equal(to_np(got_t), want)


def test_quantized_module_matches_formula(cuda):
    """quantized.SparseConv (reference quantization/quantized/conv.py:368-378): qint8 in, qint8
    out, residual add + ReLU fused."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch.quantization import quantized as spq
    shape, n, bs, C, K = [20, 20, 20], 1500, 2, 32, 64
    rng = np.random.default_rng(11)
    idx = dense_scene(shape, n, bs, 11)
    ref = oracle_rulebook(idx, bs, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    conv = spconv.SubMConv3d(C, K, 3, bias=True, indice_key="q").to(cuda)
    conv.act_type = spconv.ops.Activation.ReLU
    in_scale, out_scale, add_q_scale = 0.02, 0.05, 0.04
    q = spq.SparseConv.from_float_conv(conv, out_scale).to(cuda)
    f_i8 = rng.integers(-127, 128, (idx.shape[0], C), dtype=np.int8)
    add_i8 = rng.integers(-127, 128, (idx.shape[0], K), dtype=np.int8)
    qf = torch._make_per_tensor_quantized_tensor(torch.from_numpy(f_i8).to(cuda), in_scale, 0)
    qa = torch._make_per_tensor_quantized_tensor(torch.from_numpy(add_i8).to(cuda), add_q_scale, 0)
    ind = torch.from_numpy(idx).to(cuda)
    x = spconv.SparseConvTensor(qf, ind, shape, bs)
    a = spconv.SparseConvTensor(qa, ind, shape, bs)
    y = q(x, a)
    assert y.features.dtype == torch.qint8 and abs(y.features.q_scale() - out_scale) < 1e-9
    w_i8 = to_np(q.weight().int_repr())
    ch_scale = (in_scale * to_np(q.weight().q_per_channel_scales().float())) / out_scale
    b = to_np(q.bias().float()) / out_scale
    want = oracle.int8_conv_ref(f_i8, w_i8, ref["pair"], ref["num"], ref["n_out"], True,
                                ch_scale.astype(np.float32), b.astype(np.float32), add_i8,
                                add_q_scale / out_scale, True)
    np.testing.assert_array_equal(to_np(y.features.int_repr()), want)


def test_cfg5_full_size_int8(cuda):
    """BASELINE config 5: 200k voxels in 1600x1280x40, C = K = 128, per-channel scale, ReLU,
    int8 out.  Full-size check against the oracle on the whole tensor (bit-exact)."""
    from spconv_amd.pytorch import ops
    shape, n, C, K = [40, 1280, 1600], 200_000, 128, 128
    ks, st, pd, dl = [3] * 3, [1] * 3, [1] * 3, [1] * 3
    rng = np.random.default_rng(0)
    idx = scene(shape, n, 1, 0)
    ref = oracle_rulebook(idx, 1, shape, ks, st, pd, dl, True)
    f = rng.integers(-127, 128, (n, C), dtype=np.int8)
    w = rng.integers(-127, 128, (K, *ks, C), dtype=np.int8)
    scale = (rng.uniform(0.5, 1.5, K) * 1e-2 * 0.2).astype(np.float32)
    bias = rng.uniform(-5, 5, K).astype(np.float32)
    rb, _ = gpu_rulebook(idx, 1, shape, ks, st, pd, dl, True)
    want = oracle.int8_conv_ref(f, w, ref["pair"], ref["num"], n, True, scale, bias, None, 0.0, True)
    got = ops.igemm_fwd_int8(torch.from_numpy(f).to(cuda), torch.from_numpy(w).to(cuda), rb.pair_fwd,
                             rb.mask_fwd, None, n, 13, torch.from_numpy(scale), torch.from_numpy(bias),
                             None, 0.0, torch.int8, ops.Activation.ReLU)
    got = to_np(got)
    assert 0 < (want == 127).mean() < 0.5 and (want == 0).mean() < 0.9
    np.testing.assert_array_equal(got, want)
