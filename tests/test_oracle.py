"""CPU: pins the oracle (test infrastructure) against
 (a) the reference's own test oracle -- dense torch conv on the scattered dense input
     (test/test_conv.py:83-109,286-357; shape/seeds :248-274),
 (b) the reference's numpy per-offset formula (test/test_all_algo.py:222-288),
 (c) the committed golden fixtures (regression pin, tests/golden/make_golden.py),
 (d) the pair statistics SURVEY.md section 0.5 measured on the reference's LiDAR fixture."""
import os

import numpy as np
import pytest
import torch

import oracle
from golden import case_names, lidar_scene, load_case
from util import scene

GRID = [  # (k, s, p, d) -- subset of the reference grid test_conv.py:255-262 (skips s>1 and d>1)
    (2, 1, 0, 1), (3, 1, 0, 1), (3, 1, 1, 1), (3, 1, 2, 1), (3, 2, 0, 1), (3, 2, 1, 1), (3, 3, 2, 1),
    (2, 2, 0, 1), (3, 1, 1, 2), (3, 1, 0, 3), (2, 3, 1, 1),
]


def _sample(dense, idx):
    i = torch.from_numpy(idx.astype(np.int64))
    return dense[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]]


@pytest.mark.parametrize("k,s,p,d", GRID)
def test_regular_conv_matches_dense_conv3d(k, s, p, d):
    rng = np.random.default_rng(484)
    shape, bs, C, K = [19, 18, 17], 2, 8, 12
    idx = scene(shape, 1500, bs, seed=484)
    f = torch.from_numpy(rng.uniform(-1, 1, (idx.shape[0], C)).astype(np.float32))
    w = torch.from_numpy(rng.uniform(-1, 1, (K, k, k, k, C)).astype(np.float32))
    oi, pair, num, oshape = oracle.get_indice_pairs(idx, bs, shape, [k] * 3, [s] * 3, [p] * 3, [d] * 3)
    fd, wd = f.clone().requires_grad_(True), w.clone().requires_grad_(True)
    dense = oracle.dense_conv_reference(fd, idx, bs, shape, wd, [s] * 3, [p] * 3, [d] * 3)
    assert list(dense.shape[2:]) == oshape
    out = oracle.indice_conv(f, w, pair, num, oi.shape[0])
    ref = _sample(dense, oi)
    assert (out - ref).abs().max() < 1e-4
    # every dense site that is not an active output must be exactly zero
    m = torch.ones(dense.shape[0], *dense.shape[2:], dtype=torch.bool)
    o = torch.from_numpy(oi.astype(np.int64))
    m[o[:, 0], o[:, 1], o[:, 2], o[:, 3]] = False
    rest = dense.detach().permute(0, 2, 3, 4, 1)[m]
    assert rest.numel() == 0 or rest.abs().max() == 0
    dout = torch.from_numpy(rng.uniform(-0.2, 0.2, ref.shape).astype(np.float32))
    ref.backward(dout)
    din, dw = oracle.indice_conv_backward(f, w, dout, pair, num)
    assert (din - fd.grad).abs().max() < 1e-4
    assert (dw - wd.grad).abs().max() < 1e-4


@pytest.mark.parametrize("k,d", [(3, 1), (3, 2), (5, 1)])
def test_subm_matches_dense_conv3d_at_active_sites(k, d):
    rng = np.random.default_rng(1)
    shape, bs, C, K = [19, 18, 17], 2, 8, 12
    idx = scene(shape, 1500, bs, seed=1)
    f = torch.from_numpy(rng.uniform(-1, 1, (idx.shape[0], C)).astype(np.float32))
    w = torch.from_numpy(rng.uniform(-1, 1, (K, k, k, k, C)).astype(np.float32))
    pad = [(k // 2) * d] * 3
    oi, pair, num, _ = oracle.get_indice_pairs(idx, bs, shape, [k] * 3, [1] * 3, pad, [d] * 3, subm=True)
    fd, wd = f.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = _sample(oracle.dense_conv_reference(fd, idx, bs, shape, wd, [1] * 3, pad, [d] * 3), idx)
    out = oracle.indice_conv(f, w, pair, num, idx.shape[0], subm=True)
    assert (out - ref).abs().max() < 1e-4
    dout = torch.from_numpy(rng.uniform(-0.2, 0.2, ref.shape).astype(np.float32))
    ref.backward(dout)
    din, dw = oracle.indice_conv_backward(f, w, dout, pair, num, subm=True)
    assert (din - fd.grad).abs().max() < 1e-4
    assert (dw - wd.grad).abs().max() < 1e-4


def test_transposed_conv_matches_dense_conv_transpose3d():
    rng = np.random.default_rng(3)
    shape, bs, C, K = [10, 9, 9], 2, 8, 12
    idx = scene(shape, 300, bs, seed=3)
    f = torch.from_numpy(rng.uniform(-1, 1, (idx.shape[0], C)).astype(np.float32))
    w = torch.from_numpy(rng.uniform(-1, 1, (K, 3, 3, 3, C)).astype(np.float32))
    oi, pair, num, oshape = oracle.get_indice_pairs(idx, bs, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3,
                                                    [0] * 3, False, True)
    dense = oracle.dense_conv_reference(f, idx, bs, shape, w, [2] * 3, [1] * 3, [1] * 3, True, [0] * 3)
    assert list(dense.shape[2:]) == oshape
    out = oracle.indice_conv(f, w, pair, num, oi.shape[0])
    assert (out - _sample(dense, oi)).abs().max() < 1e-4


def test_numpy_per_offset_reference_agrees():
    """test/test_all_algo.py:222-288 restated inline: out[o_inds] += inp[i_inds] @ W_k.T etc."""
    rng = np.random.default_rng(50005)
    shape, bs, C, K = [19, 18, 17], 1, 8, 12
    idx = scene(shape, 1500, bs, seed=50005)
    for subm in (True, False):
        stride = [1] * 3 if subm else [2] * 3
        oi, pair, num, _ = oracle.get_indice_pairs(idx, bs, shape, [3] * 3, stride, [1] * 3, [1] * 3,
                                                   subm=subm)
        f = rng.uniform(-1, 1, (idx.shape[0], C)).astype(np.float32)
        w = rng.uniform(-1, 1, (K, 3, 3, 3, C)).astype(np.float32)
        dout = rng.uniform(-1, 1, (oi.shape[0], K)).astype(np.float32)
        wk = w.reshape(K, 27, C)
        out = np.zeros((oi.shape[0], K), np.float32)
        din = np.zeros_like(f)
        dw = np.zeros((27, K, C), np.float32)
        counts = oracle.native_counts(num, 27, subm, idx.shape[0])
        for k in range(27):
            i_, o_ = pair[0, k, :counts[k]], pair[1, k, :counts[k]]
            out[o_] += f[i_] @ wk[:, k].T
            din[i_] += dout[o_] @ wk[:, k]
            dw[k] = dout[o_].T @ f[i_]
        got = oracle.indice_conv(torch.from_numpy(f), torch.from_numpy(w), pair, num, oi.shape[0], subm=subm)
        gdin, gdw = oracle.indice_conv_backward(torch.from_numpy(f), torch.from_numpy(w),
                                                torch.from_numpy(dout), pair, num, subm=subm)
        assert np.abs(got.numpy() - out).max() < 1e-4
        assert np.abs(gdin.numpy() - din).max() < 1e-4
        assert np.abs(gdw.numpy().reshape(K, 27, C).transpose(1, 0, 2) - dw).max() < 1e-3


@pytest.mark.parametrize("name", case_names())
def test_golden_fixture(name):
    g = load_case(name)
    oi, pair, num, oshape = oracle.get_indice_pairs(
        g["indices"], int(g["bs"]), list(g["shape"]), list(g["ksize"]), list(g["stride"]),
        list(g["pad"]), list(g["dil"]), None, bool(g["subm"]), bool(g["transposed"]))
    np.testing.assert_array_equal(oi, g["out_inds"])
    np.testing.assert_array_equal(pair, g["pair"])
    np.testing.assert_array_equal(num, g["num"])
    assert oshape == list(g["out_shape"])
    f, w, dout = (torch.from_numpy(g[k]) for k in ("features", "weight", "dout"))
    out = oracle.indice_conv(f, w, pair, num, oi.shape[0], subm=bool(g["subm"]))
    din, dw = oracle.indice_conv_backward(f, w, dout, pair, num, subm=bool(g["subm"]))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(din.numpy(), g["din"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dw.numpy(), g["dw"], rtol=1e-5, atol=1e-4)


def test_lidar_fixture_statistics():
    """the reference's real-LiDAR scene has 125 562 voxels and
    788 888 SubM pairs (6.28 per voxel); a stride-2 chain shrinks it 125k -> 137k -> 66k -> 26k."""
    idx, shape = lidar_scene()
    assert idx.shape == (125562, 4) and shape == [80, 1600, 1600]
    _, pair, num, _ = oracle.get_indice_pairs(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, subm=True)
    assert sum(oracle.native_counts(num, 27, True, idx.shape[0])) == 788888
    cur, cshape, sizes = idx, shape, []
    for _ in range(3):
        cur, _, _, cshape = oracle.get_indice_pairs(cur, 1, cshape, [3] * 3, [2] * 3, [1] * 3, [1] * 3)
        sizes.append(cur.shape[0])
    assert [round(s / 1000) for s in sizes] == [137, 66, 26]


def test_rulebook_conventions():
    """SubM count convention (indices.py:1685,1692): only k < kv/2 counted; centre identity;
    -1 fill beyond the counts; mirror lists."""
    idx = scene([12, 12, 12], 600, 1, seed=7)
    _, pair, num, _ = oracle.get_indice_pairs(idx, 1, [12] * 3, [3] * 3, [1] * 3, [1] * 3, [1] * 3, subm=True)
    assert np.all(num[13:] == 0) and num[:13].sum() > 0
    np.testing.assert_array_equal(pair[0, 13], np.arange(600))
    np.testing.assert_array_equal(pair[1, 13], np.arange(600))
    for k in range(13):
        c = num[k]
        assert np.all(pair[:, k, c:] == -1) and np.all(pair[:, k, :c] >= 0)
        np.testing.assert_array_equal(pair[0, k, :c], pair[1, 26 - k, :c])
        np.testing.assert_array_equal(pair[1, k, :c], pair[0, 26 - k, :c])
        assert np.all(np.diff(pair[0, k, :c]) > 0)


def test_output_shape_formula():
    assert oracle.conv_out_shape([41, 1600, 1408], [3] * 3, [2] * 3, [1] * 3, [1] * 3) == [21, 800, 704]
    assert oracle.conv_out_shape([21, 800, 704], [3, 1, 1], [2, 1, 1], [0] * 3, [1] * 3) == [10, 800, 704]
    assert oracle.conv_out_shape([5, 5], [3, 3], [2, 2], [1, 1], [1, 1], [1, 1], True) == [10, 10]


def test_int8_formula_matches_float_path():
    """a15: q = clip(round(relu(acc*s_k + b_k + add*s_add)), -128, 127)."""
    rng = np.random.default_rng(0)
    idx = scene([12, 12, 12], 500, 1, seed=0)
    _, pair, num, _ = oracle.get_indice_pairs(idx, 1, [12] * 3, [3] * 3, [1] * 3, [1] * 3, [1] * 3, subm=True)
    f = rng.integers(-127, 128, (500, 16)).astype(np.int8)
    w = rng.integers(-127, 128, (8, 3, 3, 3, 16)).astype(np.int8)
    scale = (rng.uniform(0.5, 1.5, 8) * 1e-3).astype(np.float32)
    bias = rng.uniform(-1, 1, 8).astype(np.float32)
    add = rng.integers(-127, 128, (500, 8)).astype(np.int8)
    q = oracle.int8_conv_ref(f, w, pair, num, 500, True, scale, bias, add, 0.01, relu=True)
    acc = oracle.indice_conv(torch.from_numpy(f.astype(np.float64)), torch.from_numpy(w.astype(np.float64)),
                             pair, num, 500, subm=True).numpy()
    r = np.maximum(acc.astype(np.float32) * scale + bias + add.astype(np.float32) * np.float32(0.01), 0)
    assert q.dtype == np.int8 and np.array_equal(q, np.clip(np.round(r), -128, 127).astype(np.int8))


def test_int8_reference_formula_small():
    """oracle.int8_conv_ref (float64 BLAS accumulation) against plain integer loops following
    test/test_all_algo.py:222-288 literally."""
    rng = np.random.default_rng(1)
    shape, n, C, K = [8, 8, 8], 120, 16, 16
    idx = scene(shape, n, 1, 1)
    out_inds, pair, num, _ = oracle.get_indice_pairs(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3,
                                                     None, True, False)
    f = rng.integers(-127, 128, (n, C), dtype=np.int8)
    w = rng.integers(-127, 128, (K, 3, 3, 3, C), dtype=np.int8)
    scale = rng.uniform(0.5, 1.5, K).astype(np.float32) * 1e-3
    bias = rng.uniform(-5, 5, K).astype(np.float32)
    add = rng.integers(-127, 128, (n, K), dtype=np.int8)
    got = oracle.int8_conv_ref(f, w, pair, num, n, True, scale, bias, add, 0.3, True)
    wr = w.reshape(K, 27, C)
    acc = np.zeros((n, K), dtype=np.int32)
    for k in range(27):
        nhot = n if k == 13 else int(num[k] if k < 13 else num[26 - k])
        for j in range(nhot):
            i, o = (j, j) if k == 13 else (pair[0][k][j], pair[1][k][j])
            acc[o] += wr[:, k, :].astype(np.int32) @ f[i].astype(np.int32)
    r = acc.astype(np.float32) * scale + bias + add.astype(np.float32) * np.float32(0.3)
    want = np.clip(np.round(np.maximum(r, 0)), -128, 127).astype(np.int8)
    np.testing.assert_array_equal(got, want)


def test_pool_oracle_matches_dense_torch_pooling():
    """oracle.maxpool_ref / avgpool_ref over the CPU rulebook vs dense torch pooling: with
    non-negative features an empty site (0 in the dense tensor) never wins a max, and a 2x2x2
    stride-2 average over the ACTIVE voxels is sum / count of the dense window."""
    rng = np.random.default_rng(5)
    shape, n, C = [8, 8, 8], 150, 4
    idx = scene(shape, n, 1, 5)
    out_inds, pair, num, out_shape = oracle.get_indice_pairs(idx, 1, shape, [2] * 3, [2] * 3, [0] * 3,
                                                             [1] * 3, None, False, False)
    f = (rng.uniform(0.1, 1.0, (n, C))).astype(np.float32)
    dense = np.zeros((1, C, *shape), dtype=np.float32)
    occ = np.zeros((1, 1, *shape), dtype=np.float32)
    dense[0, :, idx[:, 1], idx[:, 2], idx[:, 3]] = f
    occ[0, 0, idx[:, 1], idx[:, 2], idx[:, 3]] = 1
    got = oracle.maxpool_ref(f, pair, num, out_inds.shape[0])
    want = torch.nn.functional.max_pool3d(torch.from_numpy(dense), 2, 2).numpy()
    np.testing.assert_array_equal(got, want[0][:, out_inds[:, 1], out_inds[:, 2], out_inds[:, 3]].T)
    avg, cnt = oracle.avgpool_ref(f, pair, num, out_inds.shape[0])
    s = torch.nn.functional.avg_pool3d(torch.from_numpy(dense), 2, 2).numpy() * 8
    c = torch.nn.functional.avg_pool3d(torch.from_numpy(occ), 2, 2).numpy() * 8
    sel = (slice(None), out_inds[:, 1], out_inds[:, 2], out_inds[:, 3])
    np.testing.assert_array_equal(cnt, np.rint(c[0][sel][0]).astype(np.int32))
    np.testing.assert_allclose(avg, (s[0][sel] / c[0][sel]).T, rtol=1e-6)



# ---------------------------------------------------------------------------------------------
# The restatement against the REFERENCE'S OWN CPU CODE: committed vectors produced by executing
# spconv's generate_subm_conv_inds / generate_conv_inds (tests/golden/make_ref_golden.py), and --
# where oracle/_ref was built (this container, and the GPU box it travels to) -- the live library.
# ---------------------------------------------------------------------------------------------
from golden import digest, load_ref_case, ref_big_inputs, ref_case_names, ref_digests  # noqa: E402


@pytest.mark.parametrize("name", ref_case_names())
def test_oracle_rulebook_equals_reference_executed_vectors(name):
    c = load_ref_case(name)
    out_inds, pair, num, out_shape = oracle.get_indice_pairs(
        c["indices"], c["bs"], c["shape"], c["ksize"], c["stride"], c["pad"], c["dil"], None, c["subm"],
        c["transposed"])
    assert list(out_shape) == c["out_shape"]
    np.testing.assert_array_equal(out_inds, c["out_inds"])
    np.testing.assert_array_equal(num, c["num"])
    np.testing.assert_array_equal(pair, c["pair"])


def test_oracle_rulebook_equals_reference_digests_at_baseline_sizes():
    """config 1 / config 2 scenes and the real-LiDAR fixture (SubM and the stride-2 chain of config 3):
    SHA-256 of every artefact equals the digest of what the reference's code produced."""
    want = ref_digests()
    for name in ("cfg1_subm", "cfg2_subm", "fixture_subm"):
        idx, shape = ref_big_inputs(name)
        assert digest(idx) == want[name]["input"], f"{name}: input generator drifted"
        out_inds, pair, num, _ = oracle.get_indice_pairs(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3,
                                                         None, True, False)
        assert (digest(out_inds), digest(pair), digest(num)) == (
            want[name]["out_inds"], want[name]["pair"], want[name]["num"]), name
    cur, cur_shape = ref_big_inputs("fixture_chain_l0")
    for level in range(3):
        w = want[f"fixture_chain_l{level}"]
        assert digest(cur) == w["input"]
        out_inds, pair, num, out_shape = oracle.get_indice_pairs(cur, 1, cur_shape, [3] * 3, [2] * 3, [1] * 3,
                                                                 [1] * 3, None, False, False)
        assert out_inds.shape[0] == w["n_out"] and list(out_shape) == w["out_shape"]
        assert (digest(out_inds), digest(pair), digest(num)) == (w["out_inds"], w["pair"], w["num"]), level
        cur, cur_shape = out_inds, list(out_shape)


def test_oracle_rulebook_equals_live_reference_library():
    """Randomised sweep against oracle/_ref/libspconv_ref.so when it is present."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built here (needs /root/reference: make -C oracle ref)")
    from spconv_amd.utils import synthetic
    rng = np.random.default_rng(2024)
    for trial in range(40):
        nd = int(rng.integers(1, 5))
        shape = [int(v) for v in rng.integers(6, 20 if nd > 2 else 60, nd)]
        bs = int(rng.integers(1, 4))
        n = int(min(rng.integers(1, 800), np.prod(shape) // 2))
        subm = bool(rng.integers(0, 2))
        transposed = (not subm) and bool(rng.integers(0, 3) == 0)
        ksize = [int(v) for v in (rng.choice([1, 3, 5], nd) if subm else rng.integers(1, 4, nd))]
        stride = [1] * nd if subm else [int(v) for v in rng.integers(1, 4, nd)]
        dil = [int(v) for v in rng.integers(1, 3, nd)]
        pad = [int(v) for v in rng.integers(0, 3, nd)]
        idx = synthetic.uniform_scene(shape, n, bs, seed=trial)
        try:
            want = ref.get_indice_pairs(idx, bs, shape, ksize, stride, pad, dil, None, subm, transposed)
        except (ValueError, RuntimeError):
            continue
        got = oracle.get_indice_pairs(idx, bs, shape, ksize, stride, pad, dil, None, subm, transposed)
        for a, b in zip(got[:3], want[:3]):
            np.testing.assert_array_equal(a, b, err_msg=f"trial {trial}: {shape} k{ksize} s{stride} p{pad} d{dil}")


def test_gather_scatter_add_equal_reference_executed_vectors():
    """oracle.cpp's row gather / scatter-add against tests/golden/gather_ref.npz -- outputs of the
    reference's own GatherCPU code (gather.py:30-86) executed through oracle/_ref
    (tests/golden/make_ref_gather_golden.py): bit-exact, including the accumulation order of repeated
    destination rows."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gather_ref.npz"))
    src = torch.from_numpy(d["src"].copy())
    buf = torch.full(tuple(d["gathered"].shape), float("nan"))
    oracle._gather(buf, src, d["gather_inds"], False)
    np.testing.assert_array_equal(buf.numpy(), d["gathered"])
    acc = torch.from_numpy(d["acc_before"].copy())
    oracle._scatter_add(acc, buf, d["scatter_inds"], False)
    np.testing.assert_array_equal(acc.numpy(), d["acc_after"])
    # the row-parallel variant (bench.py's cpu_baseline leg) is only defined for DISTINCT destination rows --
    # what one filter offset produces -- so it is compared on the first occurrence of every destination
    inds = d["scatter_inds"]
    first = np.sort(np.unique(inds, return_index=True)[1])
    sub_buf, sub_inds = buf[torch.from_numpy(first)].contiguous(), np.ascontiguousarray(inds[first])
    want = torch.from_numpy(d["acc_before"].copy())
    oracle._scatter_add(want, sub_buf, sub_inds, False)
    acc2 = torch.from_numpy(d["acc_before"].copy())
    oracle._scatter_add(acc2, sub_buf, sub_inds, True)
    np.testing.assert_array_equal(acc2.numpy(), want.numpy())


def test_native_conv_loop_with_the_reference_gather_code():
    """The per-offset driver loop (ops.py:962-986, 1225-1252 restated) run twice: with oracle.cpp's
    gather / scatter-add and with the reference's own GatherCPU code switched in (oracle/_ref, when
    present).  Forward, input gradient and weight gradient must be identical bit for bit."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built here (needs /root/reference: make -C oracle ref)")
    from spconv_amd.utils import synthetic
    rng = np.random.default_rng(5)
    for subm, stride in ((True, 1), (False, 2)):
        shape, bs, C, K = [14, 15, 16], 2, 5, 7
        idx = synthetic.uniform_scene(shape, 700, bs, seed=3)
        out_inds, pair, num, _ = oracle.get_indice_pairs(idx, bs, shape, [3] * 3, [stride] * 3, [1] * 3, [1] * 3,
                                                         None, subm, False)
        f = torch.from_numpy(rng.uniform(-1, 1, (idx.shape[0], C)).astype(np.float32))
        w = torch.from_numpy(rng.uniform(-1, 1, (K, 3, 3, 3, C)).astype(np.float32))
        g = torch.from_numpy(rng.uniform(-1, 1, (out_inds.shape[0], K)).astype(np.float32))
        res = []
        for use_ref in (False, True):
            oracle.use_reference_gather(use_ref)
            try:
                y = oracle.indice_conv(f, w, pair, num, out_inds.shape[0], subm=subm)
                din, dw = oracle.indice_conv_backward(f, w, g, pair, num, subm=subm)
            finally:
                oracle.use_reference_gather(False)
            res.append((y, din, dw))
        for a, b in zip(*res):
            assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------------------
# the voxeliser and the pooling loops, pinned to the reference's own CPU code, executed
# (oracle/_ref: Point2VoxelCPU, pointops.py:493-766; IndiceMaxPoolCPU, maxpool.py:590-703; vectors made by
# tests/golden/make_ref_8f_golden.py)
def _npz(name):
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))


@pytest.mark.parametrize("tag", ["plain", "mean", "capped_mean"])
def test_point2voxel_restatement_equals_reference_executed_vectors(tag):
    d = _npz("p2v_ref.npz")
    mv, mp, mean = (int(v) for v in d[f"{tag}_args"])
    pts, vs, cr, grid = d["points"], d["vsize"], d["coors_range"], d["grid_size"]
    # the loop AS IT BEHAVES in the reference (mean accumulator carried from voxel to voxel): bit for bit
    v, i, c, pid = oracle.point2voxel(pts, vs, cr, grid, mv, mp, bool(mean), reference_quirks=True)
    np.testing.assert_array_equal(i, d[f"{tag}_indices"])
    np.testing.assert_array_equal(c, d[f"{tag}_num"])
    np.testing.assert_array_equal(pid, d[f"{tag}_pid"])
    np.testing.assert_array_equal(v, d[f"{tag}_voxels"])
    assert (pid == -1).any() and c.max() == mp                      # range check and the per-voxel cap are exercised
    # the default form (arithmetic mean): identical voxel set / numbering / stored points; only the FILL of the
    # empty slots differs, and only from the second voxel on (the first voxel starts from a zero accumulator)
    v2, i2, c2, pid2 = oracle.point2voxel(pts, vs, cr, grid, mv, mp, bool(mean))
    np.testing.assert_array_equal(i2, i)
    np.testing.assert_array_equal(c2, c)
    np.testing.assert_array_equal(pid2, pid)
    stored = np.arange(mp)[None, :] < c[:, None]
    np.testing.assert_array_equal(v2[stored], v[stored])
    if mean:
        np.testing.assert_array_equal(v2[0], v[0])
        want = np.stack([v[k, :c[k]].astype(np.float64).mean(0) for k in range(len(c))])
        for k in np.nonzero(c < mp)[0][:200]:
            np.testing.assert_allclose(v2[k, c[k]:], np.broadcast_to(want[k], v2[k, c[k]:].shape), rtol=1e-6, atol=1e-7)
        assert not np.array_equal(v2, v), "the carried accumulator must show in the reference's fill"
    else:
        np.testing.assert_array_equal(v2, v)


def test_maxpool_restatements_equal_reference_executed_vectors():
    d = _npz("pool_ref.npz")
    pair, num, n_out, f, dout = d["pair"], d["num"], int(d["n_out"]), d["features"], d["dout"]
    # the rulebook the vectors were made on is the restatement's own
    idx, shape = d["indices"], [int(v) for v in d["shape"]]
    _, pair2, num2, _ = oracle.get_indice_pairs(idx, 2, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, None, False, False)
    np.testing.assert_array_equal(pair2, pair)
    np.testing.assert_array_equal(num2, num)
    # operation-for-operation restatement of ops.py:1899-1975 over maxpool.py:620-700
    out, bwd = oracle.indice_maxpool_native(f, pair, num, n_out)
    np.testing.assert_array_equal(out, d["out"])
    np.testing.assert_array_equal(bwd(dout), d["din"])
    # the vectorised forms the GPU tests use (Native path: zero-filled output; dyadic data: sums exact in any order)
    np.testing.assert_array_equal(oracle.maxpool_ref(f, pair, num, n_out, init_zero=True), d["out"])
    np.testing.assert_array_equal(oracle.maxpool_bwd_ref(f, d["out"], dout, pair, num), d["din"])
    assert (d["din"] != 0).any() and (np.count_nonzero(d["din"], axis=0) > 0).all()


def test_global_pool_rearrange_reference_executed_vector():
    """IndiceMaxPoolCPU::global_pool_rearrange (maxpool.py:598-618): rows of scene b in their order, counts; rows with a
    negative batch index belong to no scene."""
    d = _npz("pool_ref.npz")
    coords, out, cnt = d["gp_coords"], d["gp_out"], d["gp_counts"]
    for b in range(2):
        rows = np.nonzero(coords[:, 0] == b)[0]
        assert cnt[b] == len(rows)
        np.testing.assert_array_equal(out[b, :cnt[b]], rows)
    assert cnt.sum() == (coords[:, 0] >= 0).sum() < coords.shape[0]


def test_section_8f_restatements_equal_the_live_reference_library():
    """Where oracle/_ref exists (this container; it travels to the GPU box): fresh seeds against the library."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(8)
    for seed in range(3):
        pts = rng.uniform((-1, -5, -3, 0), (9, 5, 3, 1), (3000, 4)).astype(np.float32)
        pts[:1500, :3] = pts[:1500, :3] * 0.03 + np.float32(2.0)
        vs, cr, grid = [0.2, 0.1, 0.1], [-2, -4, 0, 2, 4, 8], [20, 80, 80]
        for mean in (False, True):
            a = oracle.point2voxel(pts, vs, cr, grid, 900 + 2000 * seed, 4, mean, reference_quirks=True)
            b = ref.point2voxel(pts, vs, cr, grid, 900 + 2000 * seed, 4, mean)
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x, y)
    idx = scene([16, 16, 16], 1200, 2, 5)
    _, pair, num, _ = oracle.get_indice_pairs(idx, 2, [16] * 3, [2] * 3, [2] * 3, [0] * 3, [1] * 3, None, False, False)
    n_out = int(pair[1].max()) + 1
    f = (rng.integers(-64, 65, (idx.shape[0], 5)) / 64.0).astype(np.float32)
    dout = (rng.integers(-32, 33, (n_out, 5)) / 64.0).astype(np.float32)
    out, bwd = oracle.indice_maxpool_native(f, pair, num, n_out)
    np.testing.assert_array_equal(out, ref.indice_maxpool(f, pair, num, n_out))
    np.testing.assert_array_equal(bwd(dout), ref.indice_maxpool_backward(f, out, dout, pair, num))
