"""Parity of the path bench.py times: module forward + the FUSED backward launch
(`spx_igemm_bwd` -> `igemm_bwd_kernel` dgrad tiles + wgrad ranges + second stage) against the CPU
oracle at the BASELINE configurations' full sizes.

The fused launch plans its wgrad ranges from the scene size (`wgrad_groups`), so small cases do not
exercise the configuration the benchmark runs; these tests do:

* cfg 2   -- 100 000 uniform voxels in 40x1280x1600, C = K = 64, fp16 and bf16;
* cfg 2b  -- the reference's real-LiDAR fixture coordinates (125 562 voxels, 6.28 pairs/voxel;
             test/test_multi_impl.py:224-341 uses the same fixture with CPU-Native as oracle);
* cfg 3   -- backward of the stride-2 chain 16 -> 32 -> 64 -> 128 on the fixture;
* cfg 4   -- one training step of the SECOND-style backbone over 4 scenes: every conv layer's
             forward, input gradient and weight gradient against the oracle evaluated on that
             layer's actual (GPU-produced, fp16) inputs, so errors do not compound.

Tolerances: norm-wise (relative to the largest reference magnitude, util.rel_err) fp16 2e-3, bf16 1.2e-2; and
ELEMENT-wise with no free floor (util.assert_close_abs_sum): every element of out / din / dW within half an ulp
of the output dtype of its own reference value plus 1e-6 x the same sum taken over operand magnitudes (the
quantity fp32 accumulation error is relative to; measured excess <= 2.2e-8 of it, tools/tol_probe.py)."""
import numpy as np
import pytest
import torch

import oracle
from util import (assert_close_abs_sum, assert_close_elementwise, assert_rulebook_equal, gpu_rulebook, match_rows,
                  oracle_rulebook, rel_err, scene)

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 2e-3, torch.bfloat16: 1.2e-2, torch.float32: 1e-4}
K3, ONE = [3] * 3, [1] * 3


def _rounded(a, dtype):
    return torch.from_numpy(a).to(dtype).to(torch.float32)


def _check3(tag, got, ref, tol):
    for name, g, r in zip(("out", "din", "dw"), got, ref):
        e = rel_err(g.float().cpu().numpy(), r.numpy())
        assert e <= tol, f"{tag} {name}: rel err {e:.3e} > {tol:.1e}"
        assert_close_elementwise(g.float().cpu().numpy(), r.numpy(), tol, name=f"{tag} {name}")


def _check3_abs(tag, got, ref, operands, pair, num, n_out, dtype, c=1e-6):
    """The sharp element-wise statement (util.assert_close_abs_sum): every element of out / din / dW within
    half an ulp of the output dtype of its reference value + c x the same sum over operand magnitudes."""
    f, w, dout = operands
    oa = oracle.indice_conv(f.abs(), w.abs(), pair, num, n_out, subm=True)
    dia, dwa = oracle.indice_conv_backward(f.abs(), w.abs(), dout.abs(), pair, num, subm=True)
    for name, g, r, a in zip(("out", "din", "dw"), got, ref, (oa, dia, dwa)):
        assert_close_abs_sum(g.float().cpu().numpy(), r.numpy(), a.numpy(), dtype, c, name=f"{tag} {name}")


def _fused_subm(cuda, idx, shape, C, K, dtype, seed):
    """Module forward + fused backward of one SubMConv3d; returns (rulebook, (out, din, dw))."""
    import spconv_amd.pytorch as spconv
    rng = np.random.default_rng(seed)
    n = idx.shape[0]
    f = _rounded(rng.uniform(-1, 1, (n, C)).astype(np.float32), dtype)
    w = _rounded(rng.uniform(-1, 1, (K, 3, 3, 3, C)).astype(np.float32), dtype)
    dout = _rounded(rng.uniform(-0.2, 0.2, (n, K)).astype(np.float32), dtype)
    net = spconv.SubMConv3d(C, K, 3, bias=False, indice_key="t").to(cuda, dtype)
    with torch.no_grad():
        net.weight.copy_(w.to(cuda, dtype))
    net.train()
    feats = f.to(cuda, dtype).requires_grad_(True)
    x = spconv.SparseConvTensor(feats, torch.from_numpy(idx).to(cuda), shape, 1)
    y = net(x)
    y.features.backward(dout.to(cuda, dtype))
    torch.cuda.synchronize()
    rb = y.indice_dict["t"].rulebook
    return rb, (f, w, dout), (y.features.detach(), feats.grad, net.weight.grad)


@pytest.mark.parametrize("C,K,n", [(16, 16, 5000), (64, 32, 40_000)])
def test_cfg1_fp32_fused_bwd_vs_oracle(cuda, C, K, n):
    """BASELINE config 1 (fp32, 5 k voxels in 64^3, C = 16) and a larger fp32 layer: module forward +
    the fused fp32 backward (dgrad tiles + wgrad_f32 ranges in one launch); north_star asks 1e-3
    relative for fp32 features -- element-wise here, with the norm-wise bound at 1e-4."""
    shape = [64, 64, 64]
    idx = scene(shape, n, 1, 0)
    rb, (f, w, dout), got = _fused_subm(cuda, idx, shape, C, K, torch.float32, seed=5)
    ref = oracle_rulebook(idx, 1, shape, K3, ONE, ONE, ONE, True)
    assert_rulebook_equal(rb, ref, True)
    out_ref = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"], subm=True)
    din_ref, dw_ref = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"], subm=True)
    for name, g, r in zip(("out", "din", "dw"), got, (out_ref, din_ref, dw_ref)):
        assert rel_err(g.float().cpu().numpy(), r.numpy()) <= 1e-4, name
        assert_close_elementwise(g.float().cpu().numpy(), r.numpy(), 1e-3, floor_frac=1e-4, name=f"cfg1 {name}")
    # fp32: the products are rounded too (v_mfma_f32_16x16x4_f32), so the constant is sqrt(n) 2^-24 with some room
    _check3_abs("cfg1", got, (out_ref, din_ref, dw_ref), (f, w, dout), ref["pair"], ref["num"], ref["n_out"],
                torch.float32, c=1e-5)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_cfg2_full_size_fused_bwd_vs_oracle(cuda, dtype):
    shape = [40, 1280, 1600]
    idx = scene(shape, 100_000, 1, 0)
    rb, (f, w, dout), got = _fused_subm(cuda, idx, shape, 64, 64, dtype, seed=0)
    # what net(x) cached is what bench.py times: the rows layout built inside the rulebook build, regrouped here
    assert rb.layout is not None and int(rb.layout[0].item()) == 1 and rb.argsort_fwd is None
    ref = oracle_rulebook(idx, 1, shape, K3, ONE, ONE, ONE, True)
    assert_rulebook_equal(rb, ref, True)
    out_ref = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"], subm=True)
    din_ref, dw_ref = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"], subm=True)
    _check3(f"cfg2 {dtype}", got, (out_ref, din_ref, dw_ref), TOL[dtype])
    _check3_abs(f"cfg2 {dtype}", got, (out_ref, din_ref, dw_ref), (f, w, dout), ref["pair"], ref["num"], ref["n_out"], dtype)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_cfg2b_lidar_fixture_fused_bwd_vs_oracle(cuda, dtype):
    from golden import lidar_scene
    idx, shape = lidar_scene()
    rb, (f, w, dout), got = _fused_subm(cuda, idx, shape, 64, 64, dtype, seed=1)
    assert rb.layout is not None and int(rb.layout[0].item()) == 0         # dense: identity order
    ref = oracle_rulebook(idx, 1, shape, K3, ONE, ONE, ONE, True)
    assert_rulebook_equal(rb, ref, True)
    assert int(idx.shape[0] + 2 * ref["num"][:13].sum()) == 788_888
    out_ref = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"], subm=True)
    din_ref, dw_ref = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"], subm=True)
    _check3(f"cfg2b {dtype}", got, (out_ref, din_ref, dw_ref), TOL[dtype])
    _check3_abs(f"cfg2b {dtype}", got, (out_ref, din_ref, dw_ref), (f, w, dout), ref["pair"], ref["num"], ref["n_out"], dtype)


def test_cfg2_sorted_rows_fused_bwd_vs_oracle(cuda):
    """The same launch with mask-sorted row order (SPCONV_DO_SORT / bench.py --sort)."""
    from spconv_amd.pytorch import ops
    shape = [40, 1280, 1600]
    idx = scene(shape, 100_000, 1, 3)
    rng = np.random.default_rng(3)
    f = _rounded(rng.uniform(-1, 1, (idx.shape[0], 64)).astype(np.float32), torch.float16)
    w = _rounded(rng.uniform(-1, 1, (64, 3, 3, 3, 64)).astype(np.float32), torch.float16)
    dout = _rounded(rng.uniform(-0.2, 0.2, (idx.shape[0], 64)).astype(np.float32), torch.float16)
    ref = oracle_rulebook(idx, 1, shape, K3, ONE, ONE, ONE, True)
    rb, _ = gpu_rulebook(idx, 1, shape, K3, ONE, ONE, ONE, True, do_sort=True)
    fg, wg, dg = (t.to(cuda).half() for t in (f, w, dout))
    out = ops.igemm_fwd(fg, wg, rb.pair_fwd, rb.mask_fwd, rb.argsort_fwd, rb.n_out, 13)
    din, dw = ops.igemm_bwd(fg, dg, wg, rb.pair_fwd, rb.mask_fwd, rb.argsort_fwd, rb.pair_native,
                            rb.num_per_loc, True, ops._plan_of(rb))
    out_ref = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"], subm=True)
    din_ref, dw_ref = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"], subm=True)
    _check3("cfg2 sorted", (out, din, dw), (out_ref, din_ref, dw_ref), 2e-3)
    _check3_abs("cfg2 sorted", (out, din, dw), (out_ref, din_ref, dw_ref), (f, w, dout), ref["pair"], ref["num"],
                ref["n_out"], torch.float16)


class _Tap:
    """Records what every sparse conv layer of a network saw in one training step."""

    def __init__(self, net):
        from spconv_amd.utils.nets import conv_layers
        self.layers = conv_layers(net)
        self.rec = {}
        self.handles = [m.register_forward_hook(self._hook) for m in self.layers]

    def _hook(self, mod, args, out):
        x = args[0]
        fin, fout = x.features, out.features
        if fin.requires_grad:
            fin.retain_grad()
        fout.retain_grad()
        self.rec[id(mod)] = dict(idx=x.indices, shape=list(x.spatial_shape), bs=x.batch_size, fin=fin,
                                 fout=fout, out_idx=out.indices, out_shape=list(out.spatial_shape))

    def close(self):
        for h in self.handles:
            h.remove()


def _check_layers_vs_oracle(tap, tol, first_has_din=False, exact_order=True):
    """Every conv layer of the step: coordinates bit-exact, forward / dgrad / wgrad within tol of
    the oracle evaluated on the layer's own inputs."""
    for li, m in enumerate(tap.layers):
        r = tap.rec[id(m)]
        idx = r["idx"].cpu().numpy()
        ref = oracle_rulebook(idx, r["bs"], r["shape"], m.kernel_size, m.stride, m.padding, m.dilation,
                              m.subm)
        assert r["out_shape"] == list(ref["out_shape"])
        # rows of the layer's output in the oracle's (first-seen) numbering: the identity unless a strided layer
        # numbers its outputs by coordinate key (constants.CONV_OUTPUT_ORDER = "sorted")
        perm = match_rows(r["out_idx"].cpu().numpy(), ref["out_inds"], r["out_shape"])
        if exact_order:
            np.testing.assert_array_equal(r["out_idx"].cpu().numpy(), ref["out_inds"])
        f = r["fin"].detach().float().cpu()
        w = m.weight.detach().float().cpu()
        dout_got = r["fout"].grad.detach().float().cpu()
        assert torch.isfinite(dout_got).all() and float(dout_got.abs().max()) > 0, f"layer {li}: degenerate gradient"
        dout = torch.empty_like(dout_got)
        dout[torch.from_numpy(perm)] = dout_got
        out_ref = oracle.indice_conv(f, w, ref["pair"], ref["num"], ref["n_out"], subm=m.subm)[torch.from_numpy(perm)]
        din_ref, dw_ref = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"], subm=m.subm)
        tag = f"layer {li} ({m.in_channels}->{m.out_channels}, {'subm' if m.subm else 'conv'}, n={idx.shape[0]})"
        e = rel_err(r["fout"].detach().float().cpu().numpy(), out_ref.numpy())
        assert e <= tol, f"{tag} out: {e:.3e}"
        e = rel_err(m.weight.grad.float().cpu().numpy(), dw_ref.numpy())
        assert e <= tol, f"{tag} dw: {e:.3e}"
        if r["fin"].grad is not None and (li > 0 or first_has_din):
            e = rel_err(r["fin"].grad.float().cpu().numpy(), din_ref.numpy())
            assert e <= tol, f"{tag} din: {e:.3e}"


@pytest.mark.parametrize("order", ["first_seen", "sorted"])
def test_cfg3_chain_backward_on_lidar_fixture_vs_oracle(cuda, monkeypatch, order):
    """Backward of BASELINE config 3 (regular-conv rulebooks, fused dgrad + wgrad on pair_bwd and the
    Native lists) on the reference fixture; forward-only coverage lives in test_gpu_modules.py.  Both row orders of
    the strided layers' outputs: the CPU reference's (coordinates bit-exact, in order) and the sorted one (the same
    coordinate sets; features and gradients compared row by row under the permutation)."""
    import spconv_amd.pytorch as spconv
    from golden import lidar_scene
    from spconv_amd import constants
    from spconv_amd.utils.nets import downsample_chain
    monkeypatch.setattr(constants, "CONV_OUTPUT_ORDER", order)
    idx, shape = lidar_scene()
    torch.manual_seed(3)
    net = downsample_chain().to(cuda).half().train()
    tap = _Tap(net)
    f = torch.randn(idx.shape[0], 16).half().to(cuda).requires_grad_(True)
    y = net(spconv.SparseConvTensor(f, torch.from_numpy(idx).to(cuda), shape, 1))
    g = (torch.rand(y.features.shape, device=cuda) - 0.5).half() * 0.4
    y.features.backward(g)
    torch.cuda.synchronize()
    _check_layers_vs_oracle(tap, 2e-3, first_has_din=True, exact_order=order == "first_seen")
    tap.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("order", ["first_seen", "sorted"])
def test_cfg4_backbone_step_fused_bwd_vs_oracle(cuda, monkeypatch, order):
    """One training step of the SECOND-style backbone (BatchNorm + ReLU, fp16) over a batch of 4
    LiDAR-density scenes of 100 k voxels: all 12 sparse conv layers against the oracle, in both row orders of the
    strided layers' outputs (sorted: SubM rulebooks of levels 2-4 come from the rank maps)."""
    import spconv_amd.pytorch as spconv
    from spconv_amd import constants
    from spconv_amd.utils import synthetic
    monkeypatch.setattr(constants, "CONV_OUTPUT_ORDER", order)
    from spconv_amd.utils.nets import SECOND_SHAPE, second_backbone
    bs = 4
    idx = synthetic.lidar_like_scene(SECOND_SHAPE, 100_000, bs, seed=11)
    torch.manual_seed(4)
    net = second_backbone(4).to(cuda).half().train()
    tap = _Tap(net)
    f = torch.randn(idx.shape[0], 4).half().to(cuda)
    y = net(spconv.SparseConvTensor(f, torch.from_numpy(idx).to(cuda), SECOND_SHAPE, bs))
    # (a mean-of-squares loss would put 1/N-sized gradients below fp16's subnormal range)
    g = (torch.rand(y.features.shape, device=cuda) - 0.5).half() * 0.2
    y.features.backward(g)
    torch.cuda.synchronize()
    assert len(tap.layers) == 12
    _check_layers_vs_oracle(tap, 3e-3, exact_order=order == "first_seen")
    tap.close()


@pytest.mark.parametrize("ksize", [(3, 1, 1), (3, 3, 1)])
def test_wgrad_plan_fits_its_buffer_at_small_kernel_volumes(cuda, ksize):
    """Round-4 ADVICE (medium): for n in (106 496, 114 688] the SubM rule asks for 384 wgrad ranges while the plan
    buffer was sized with the non-SubM rule's 164 -- at kv = 3 / 9 the plan kernel then wrote ~10 KB past the buffer
    (into whatever the caching allocator handed out next).  A guard tensor allocated right behind the plan must stay
    untouched and the gradients must equal the oracle."""
    import spconv_amd.pytorch as spconv
    shape, n, C = [40, 1280, 1600], 110_000, 64
    idx = scene(shape, n, 1, 3)
    rng = np.random.default_rng(7)
    dtype = torch.float16
    f = _rounded(rng.uniform(-1, 1, (n, C)).astype(np.float32), dtype)
    w = _rounded(rng.uniform(-1, 1, (C, *ksize, C)).astype(np.float32), dtype)
    dout = _rounded(rng.uniform(-0.2, 0.2, (n, C)).astype(np.float32), dtype)
    net = spconv.SubMConv3d(C, C, ksize, bias=False, indice_key="t").to(cuda, dtype)
    with torch.no_grad():
        net.weight.copy_(w.to(cuda, dtype))
    feats = f.to(cuda, dtype).requires_grad_(True)
    x = spconv.SparseConvTensor(feats, torch.from_numpy(idx).to(cuda), shape, 1)
    y = net(x)
    # guards the allocator may place right behind the plan, which the backward derives on demand
    torch.cuda.synchronize()
    guards = [torch.full((4096,), 0x5A5A5A5A, dtype=torch.int32, device=cuda) for _ in range(8)]
    y.features.backward(dout.to(cuda, dtype))
    torch.cuda.synchronize()
    for g in guards:
        assert bool((g == 0x5A5A5A5A).all())
    ref = oracle_rulebook(idx, 1, shape, list(ksize), ONE, ONE, ONE, True)
    din_ref, dw_ref = oracle.indice_conv_backward(f, w, dout, ref["pair"], ref["num"], subm=True)
    assert rel_err(feats.grad.float().cpu().numpy(), din_ref.numpy()) <= TOL[dtype]
    assert rel_err(net.weight.grad.float().cpu().numpy(), dw_ref.numpy()) <= TOL[dtype]
    # the sizing rule itself: the byte count covers the larger of the two range counts
    from spconv_amd import _lib
    L = _lib.load()
    kv = int(np.prod(ksize))
    assert L.spx_wgrad_plan_bytes(n, kv) >= 4 * (8 + 8 * 384 + kv + 1 + 3 * (384 + kv))


def test_deferred_second_stage_is_bit_identical_and_one_launch(cuda):
    """spx_igemm_bwd_deferred + spx_wgrad_stage2_batch (ops.deferred_wgrad: the static training runner's backward): the
    reductions of every layer's partial weight-gradient tiles run as ONE launch when the pass ends, from the autograd
    engine's final callback.  Same kernel body, same summation order: every gradient bit-identical to the pass that
    reduces behind each layer (the reference returns din and dw from one blocking call, pytorch/ops.py:1667-1896) -- with a
    weight used twice in the pass (what is pending is flushed before the engine adds the two gradients), with a weight
    that already holds a gradient (not deferred), eagerly and inside a capture."""
    import spconv_amd.pytorch as spconv
    from spconv_amd import _lib
    from spconv_amd.pytorch import ops
    from util import scene
    L = _lib.load()
    shape, bs = [24, 48, 48], 2
    idx = torch.from_numpy(scene(shape, 9000, bs, seed=11)).to(cuda)
    n = idx.shape[0]
    torch.manual_seed(5)
    f = (torch.randn(n, 64, device=cuda) * 0.5).half()
    shared = spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="a")
    net = spconv.SparseSequential(spconv.SubMConv3d(64, 32, 3, bias=False, indice_key="a"),
                                  spconv.SubMConv3d(32, 64, 3, bias=False, indice_key="a"), shared,
                                  spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="a"), shared).to(cuda).half().train()
    g = ((torch.rand(n, 64, device=cuda) - 0.5) * 0.1).half()
    fin = f.clone().requires_grad_(True)

    def step(defer):
        net.zero_grad(set_to_none=True)
        fin.grad = None
        y = net(spconv.SparseConvTensor(fin, idx, shape, bs)).features
        if defer:
            with ops.deferred_wgrad():
                y.backward(g)
        else:
            y.backward(g)

    def counts():
        return [int(L.spx_launch_count(k)) for k in (b"wgrad_stage2", b"wgrad_stage2_batch", b"igemm_bwd")]

    step(False)
    torch.cuda.synchronize()
    want = [p.grad.clone() for p in net.parameters()] + [fin.grad.clone()]
    c0 = counts()
    step(False)
    c1 = counts()
    per_pass = c1[0] - c0[0]
    assert per_pass >= 5 and c1[1] == c0[1] and c1[2] - c0[2] >= 3      # a second stage behind every layer
    step(True)
    torch.cuda.synchronize()
    c2 = counts()
    got = [p.grad.clone() for p in net.parameters()] + [fin.grad.clone()]
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # deferred: the fused layers' second stages ran as batches (one at the shared weight's second use, one at the end)
    assert c2[0] - c1[0] < per_pass and 1 <= c2[1] - c1[1] <= 2, (c0, c1, c2)
    assert not ops._defer_passes
    # a weight that already holds a gradient accumulates: not deferred, still right
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    fin.grad = None
    with ops.deferred_wgrad():
        net(spconv.SparseConvTensor(fin, idx, shape, bs)).features.backward(g)
    torch.cuda.synchronize()
    for p, w in zip(net.parameters(), want):
        assert torch.equal(p.grad, w)
    # captured
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(True)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step(True)
    for p in net.parameters():
        p.grad.zero_()
    fin.grad.zero_()
    graph.replay()
    torch.cuda.synchronize()
    got = [p.grad.clone() for p in net.parameters()] + [fin.grad.clone()]
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert not ops._defer_passes
