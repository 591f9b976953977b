"""GPU: static-shape inference -- the bounded rulebook build that reads nothing back
(spx_conv_rulebook_static, include/spconv_amd.h; the reference's num_out_act_bound mode,
spconv/pytorch/ops.py:263-266,644-645 and csrc/sparse/all.py:2030-2185) and a whole backbone, rulebook
builds included, captured in one graph (spconv_amd/pytorch/static.py).  Bar: the live rows are
BIT-identical to the eager, unbounded path on the same scene; dead rows never leak into live ones."""
import copy

import numpy as np
import pytest
import torch

from util import gpu_rulebook, oracle_rulebook, scene, to_np

pytestmark = pytest.mark.gpu


def _padded(idx, n_static):
    pad = np.full((n_static - idx.shape[0], idx.shape[1]), -1, np.int32)
    return np.concatenate([idx, pad], 0)


@pytest.mark.parametrize("ksize,stride,padding,transpose", [([3] * 3, [2] * 3, [1] * 3, False),
                                                           ([2] * 3, [2] * 3, [0] * 3, False),
                                                           ([3, 1, 3], [2, 1, 2], [1, 0, 1], False),
                                                           ([3] * 3, [2] * 3, [1] * 3, True)])
def test_static_rulebook_equals_dynamic_and_oracle(cuda, ksize, stride, padding, transpose):
    shape, bs, n, n_static = [20, 22, 24], 2, 1500, 4096
    idx = scene(shape, n, bs, 3)
    n = idx.shape[0]
    ref = oracle_rulebook(idx, bs, shape, ksize, stride, padding, [1] * 3, False, transpose)
    n_out = ref["n_out"]
    cap = n_out + 777
    rb, _ = gpu_rulebook(_padded(idx, n_static), bs, shape, ksize, stride, padding, [1] * 3, False, transpose,
                         static_num_out=cap, need_native=False)
    assert rb.n_out == cap and rb.n_in == n_static
    assert to_np(rb.n_out_dev).tolist() == [n_out, 0]
    oi, pf, pb = to_np(rb.out_indices), to_np(rb.pair_fwd), to_np(rb.pair_bwd)
    np.testing.assert_array_equal(oi[:n_out], ref["out_inds"])
    assert (oi[n_out:] == -1).all()
    np.testing.assert_array_equal(pf[:, :n_out], ref["fwd"])
    assert (pf[:, n_out:] == -1).all()
    np.testing.assert_array_equal(pb[:, :n], ref["bwd"])
    assert (pb[:, n:] == -1).all()
    mf, mb = to_np(rb.mask_fwd).view(np.uint32), to_np(rb.mask_bwd).view(np.uint32)
    np.testing.assert_array_equal(mf[:n_out], ref["mfwd"])
    assert (mf[n_out:] == 0).all() and (mb[n:] == 0).all()


def test_static_rulebook_cap_below_count_keeps_first_rows(cuda):
    shape, bs = [24, 24, 24], 1
    idx = scene(shape, 4000, bs, 5)
    ref = oracle_rulebook(idx, bs, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False)
    n_out = ref["n_out"]
    cap = n_out // 2
    rb, _ = gpu_rulebook(idx, bs, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False, static_num_out=cap,
                         need_native=False)
    assert to_np(rb.n_out_dev).tolist() == [n_out, 0]          # the count found, not the cap
    np.testing.assert_array_equal(to_np(rb.out_indices), ref["out_inds"][:cap])
    np.testing.assert_array_equal(to_np(rb.pair_fwd), ref["fwd"][:, :cap])
    pb = to_np(rb.pair_bwd)
    np.testing.assert_array_equal(pb, np.where(ref["bwd"] < cap, ref["bwd"], -1))


def test_subm_rulebook_ignores_dead_rows(cuda):
    shape, bs, n, n_static = [30, 30, 30], 1, 5000, 6001
    idx = scene(shape, n, bs, 7)
    n = idx.shape[0]
    ref = oracle_rulebook(idx, bs, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    rb, _ = gpu_rulebook(_padded(idx, n_static), bs, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True,
                         need_native=False)
    pf = to_np(rb.pair_fwd)
    np.testing.assert_array_equal(pf[:, :n], ref["fwd"])
    dead = pf[:, n:]
    centre = 13
    assert (np.delete(dead, centre, 0) == -1).all()            # a dead row has no neighbour ...
    assert (pf[:, :n] < n).all()                               # ... and no live row points at one


def _backbone(spconv, C, dev, dtype, pool=False):
    from torch import nn
    torch.manual_seed(7)
    down2 = (spconv.SparseMaxPool3d(2, 2) if pool
             else spconv.SparseConv3d(32, 64, 3, 2, 1, bias=False, indice_key="d2"))
    net = spconv.SparseSequential(
        spconv.SubMConv3d(C, 16, 3, bias=False, indice_key="s0"), nn.BatchNorm1d(16), nn.ReLU(),
        spconv.SubMConv3d(16, 16, 3, bias=False, indice_key="s0"), nn.BatchNorm1d(16), nn.ReLU(),
        spconv.SparseConv3d(16, 32, 3, 2, 1, bias=False, indice_key="d1"), nn.BatchNorm1d(32), nn.ReLU(),
        spconv.SubMConv3d(32, 32, 3, bias=True, indice_key="s1"), nn.ReLU(),
        down2,
        spconv.SubMConv3d(32 if pool else 64, 64, 3, bias=False, indice_key="s2"), nn.BatchNorm1d(64), nn.ReLU(),
    ).to(dev)
    with torch.no_grad():                                      # non-trivial running statistics
        for m in net.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
    return net.to(dtype).eval()


def _scene_tensors(shape, n, bs, C, seed, dev, dtype):
    idx = scene(shape, n, bs, seed)
    rng = np.random.default_rng(seed)
    f = torch.from_numpy(rng.uniform(-1, 1, (idx.shape[0], C)).astype(np.float32)).to(dev, dtype)
    return f, torch.from_numpy(idx).to(dev)


@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_captured_backbone_is_bit_identical_to_eager(cuda, pool, dtype):
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch.static import StaticInference, dense_static, strided_layers
    shape, bs, C = [32, 40, 40], 2, 4
    net = _backbone(spconv, C, cuda, dtype, pool)
    assert len(strided_layers(net)) == 2
    names = list(strided_layers(net))
    eager = copy.deepcopy(net)                                  # (the runner freezes bounds on `net`)
    runner = StaticInference(net, max_voxels=12_000, in_channels=C, spatial_shape=shape, batch_size=bs,
                             dtype=dtype, bounds={names[0]: 13_000, names[1]: 1_700})
    for n, seed in ((4500, 1), (2001, 2), (5999, 3), (388, 4)):        # voxels per batch item: growing and shrinking scenes
        f, idx = _scene_tensors(shape, n, bs, C, seed, cuda, dtype)
        with torch.no_grad():
            want = eager(spconv.SparseConvTensor(f, idx, shape, bs))
        got = runner(f, idx)
        assert runner.overflowed() == {}, runner.counts()
        live = got.indices[:, 0] >= 0
        n_live = int(live.sum())
        assert n_live == want.indices.shape[0]
        assert bool(live[:n_live].all())                        # live rows first, in the canonical order
        assert torch.equal(got.indices[:n_live], want.indices)
        assert torch.equal(got.features[:n_live], want.features)
        if not pool:                                            # (a max pool leaves `lowest` in its dead rows)
            assert torch.isfinite(got.features.float()).all()
        assert torch.equal(dense_static(got), want.dense())
        assert torch.equal(got.dense(), want.dense())           # (.dense() of a static tensor takes the same path)
        counts = runner.counts()
        assert counts[names[1]][0] == n_live


def test_overflow_is_reported_and_first_rows_survive(cuda):
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch.static import StaticInference
    shape, bs, C = [24, 24, 24], 1, 4
    net = spconv.SparseSequential(spconv.SparseConv3d(C, 16, 3, 2, 1, bias=False)).to(cuda).half().eval()
    f, idx = _scene_tensors(shape, 5000, bs, C, 9, cuda, torch.float16)
    with torch.no_grad():
        want = net(spconv.SparseConvTensor(f, idx, shape, bs))
    n_out = want.indices.shape[0]
    cap = n_out - 100
    runner = StaticInference(net, 6000, C, shape, bs, torch.float16, bounds={"0": cap})
    got = runner(f, idx)
    assert runner.overflowed() == {"0": n_out}
    assert torch.equal(got.indices, want.indices[:cap])
    assert torch.equal(got.features, want.features[:cap])      # a surviving output keeps ALL its pairs


def test_bounds_from_recorded_voxel_counts(cuda):
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch.static import freeze_bounds, strided_layers
    shape, bs, C = [24, 24, 24], 1, 4
    net = spconv.SparseSequential(
        spconv.SparseConv3d(C, 16, 3, 2, 1, bias=False, record_voxel_count=True)).to(cuda).half()
    with pytest.raises(ValueError, match="no recorded voxel count"):
        freeze_bounds(net)
    sizes = []
    net.train()
    for seed in (1, 2, 3):
        f, idx = _scene_tensors(shape, 3000 + 500 * seed, bs, C, seed, cuda, torch.float16)
        sizes.append(net(spconv.SparseConvTensor(f, idx, shape, bs)).indices.shape[0])
    used = freeze_bounds(net, margin=1.5)
    assert used == {"0": int(max(sizes) * 1.5) + 1}
    assert list(strided_layers(net).values())[0].static_num_out == used["0"]
    assert freeze_bounds(net, {}, margin=0) == {"0": 0}
    f, idx = _scene_tensors(shape, 3500, bs, C, 1, cuda, torch.float16)
    assert net(spconv.SparseConvTensor(f, idx, shape, bs)).indices.shape[0] == sizes[0]


def test_static_training_step_matches_eager(cuda):
    """A strided layer with a frozen bound keeps it in training mode: the Native lists come out of the same
    sync-free build, forward / dgrad / wgrad over a padded input equal the eager step on the live rows."""
    import spconv_amd.pytorch as spconv
    shape, bs, C, K = [24, 24, 24], 1, 16, 32
    torch.manual_seed(3)
    net = spconv.SparseConv3d(C, K, 3, 2, 1, bias=False).to(cuda).half().train()
    f, idx = _scene_tensors(shape, 4000, bs, C, 11, cuda, torch.float16)
    fe = f.clone().requires_grad_(True)
    ye = net(spconv.SparseConvTensor(fe, idx, shape, bs))
    n_out = ye.features.shape[0]
    cap, n_static = n_out + 300, 4500
    g = ((torch.rand((cap, K), device=cuda) - 0.5) * 0.2).half()
    ye.features.backward(g[:n_out])
    dw_e, din_e = net.weight.grad.clone(), fe.grad.clone()
    net.weight.grad = None
    net.static_num_out = cap
    fs = torch.zeros((n_static, C), dtype=torch.float16, device=cuda)
    fs[:f.shape[0]] = f
    fs.requires_grad_(True)
    ids = torch.full((n_static, 4), -1, dtype=torch.int32, device=cuda)
    ids[:idx.shape[0]] = idx
    ys = net(spconv.SparseConvTensor(fs, ids, shape, bs))
    assert ys.features.shape[0] == cap
    assert torch.equal(ys.indices[:n_out], ye.indices) and bool((ys.indices[n_out:] == -1).all())
    assert torch.equal(ys.features[:n_out], ye.features)
    ys.features.backward(g)
    assert torch.equal(fs.grad[:f.shape[0]], din_e) and not bool(fs.grad[f.shape[0]:].any())
    rel = float((net.weight.grad.float() - dw_e.float()).norm() / dw_e.float().norm())
    assert rel < 1e-3, rel                      # (the weight-gradient ranges are cut by row count: order differs)


def test_static_training_step_with_subm_and_batchnorm(cuda):
    """SubM + BatchNorm + ReLU + strided layers in TRAINING mode over a padded input: the tensors carry the
    device-side live-row count, the normalisation takes its statistics over the live rows and zeroes the
    padding in both directions, so parameter gradients and live outputs equal the eager step's (up to
    the summation order of the statistics)."""
    import spconv_amd.pytorch as spconv
    shape, bs, C = [32, 40, 40], 2, 8
    net = _backbone(spconv, C, cuda, torch.float16).train()
    eager = copy.deepcopy(net)
    f, idx = _scene_tensors(shape, 4000, bs, C, 5, cuda, torch.float16)
    n = f.shape[0]
    fe = f.clone().requires_grad_(True)
    ye = eager(spconv.SparseConvTensor(fe, idx, shape, bs))
    n_out = ye.features.shape[0]
    names = list(__import__("spconv_amd.pytorch.static", fromlist=["x"]).strided_layers(net))
    mods = dict(net.named_modules())
    mods[names[0]].static_num_out, mods[names[1]].static_num_out = 13_000, n_out + 200
    g = ((torch.rand((n_out + 200, 64), device=cuda) - 0.5) * 0.2).half()
    ye.features.backward(g[:n_out])
    n_static = n + 1234
    fs = torch.zeros((n_static, C), dtype=torch.float16, device=cuda)
    fs[:n] = f
    fs.requires_grad_(True)
    ids = torch.full((n_static, 4), -1, dtype=torch.int32, device=cuda)
    ids[:n] = idx
    x = spconv.SparseConvTensor(fs, ids, shape, bs)
    x.n_live_dev = torch.tensor([n], dtype=torch.int32, device=cuda)
    ys = net(x)
    assert int(ys.n_live_dev) == n_out
    assert torch.equal(ys.indices[:n_out], ye.indices)
    assert not bool(ys.features[n_out:].any())                    # padding rows: zeros after the last BN + ReLU
    err = float((ys.features[:n_out].float() - ye.features.float()).abs().max())
    assert err <= 2e-2 * float(ye.features.float().abs().max()), err
    ys.features.backward(g)
    assert not bool(fs.grad[n:].any())
    for (name, pa), pb in zip(net.named_parameters(), eager.parameters()):
        rel = float((pa.grad.float() - pb.grad.float()).norm() / pb.grad.float().norm().clamp_min(1e-12))
        assert rel < 2e-2, (name, rel)
    rel = float((fs.grad[:n].float() - fe.grad.float()).norm() / fe.grad.float().norm())
    assert rel < 2e-2, rel


def test_static_training_step_fp32_is_a_bound_not_a_noise_floor(cuda):
    """The captured step against the eager step in fp32 (at fp16 the check through 12 BatchNorm
    layers is a noise-floor argument).  In fp32 only summation orders differ (statistics over padded blocks, weight-
    gradient ranges cut by a different row count): every parameter gradient within 2e-4, the output within 1e-5."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch.static import StaticTrainingStep, strided_layers
    shape, bs, C = [32, 40, 40], 2, 8
    net = _backbone(spconv, C, cuda, torch.float32).train()
    eager = copy.deepcopy(net)
    names = list(strided_layers(net))
    g = (torch.rand((1_700, 64), device=cuda) - 0.5) * 0.2
    f, idx = _scene_tensors(shape, 4000, bs, C, 1, cuda, torch.float32)
    step = StaticTrainingStep(net, 9_000, C, shape, bs, torch.float32, bounds={names[0]: 13_000, names[1]: 1_700},
                              out_grad=g, input_grad=True, example=(f, idx))
    out = step(f, idx)
    fe = f.clone().requires_grad_(True)
    ye = eager(spconv.SparseConvTensor(fe, idx, shape, bs))
    n_out = ye.features.shape[0]
    ye.features.backward(g[:n_out])
    assert int(out.n_live_dev) == n_out and torch.equal(out.indices[:n_out], ye.indices)
    err = float((out.features[:n_out] - ye.features).abs().max() / ye.features.abs().max())
    assert err < 1e-5, err
    for (name, pa), pb in zip(net.named_parameters(), eager.parameters()):
        rel = float((pa.grad - pb.grad).norm() / pb.grad.norm().clamp_min(1e-20))
        assert rel < 2e-4, (name, rel)
    rel = float((step.features.grad[:f.shape[0]] - fe.grad).norm() / fe.grad.norm())
    assert rel < 2e-4, rel


def test_config4_network_captured_step_equals_eager_in_fp32(cuda, monkeypatch):
    """The BASELINE config-4 network itself (spconv_amd.utils.nets.second_backbone: 12 sparse convolutions, 12
    BatchNorm1d + ReLU, the p = (0, 1, 1) and (3, 1, 1) / (2, 1, 1) layers) as ONE captured training step against the
    eager step -- in fp32, where the comparison is a bound and not a noise floor (bench.py accepts the
    fp16 step against a self-permutation floor).  Two references: (a) the same static-shape pass run eagerly -- every
    gradient BIT FOR BIT; (b) the eager UNBOUNDED step -- output 1e-5, gradients within the bound of one flipped ReLU
    mask (see the comment at (b): round 6 found the former 5e-4 bar to hold only while no pre-ReLU value of the ~1 M in
    the network sits within 1e-6 of zero, which depends on how the two passes tile their BatchNorm statistics)."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch import ops
    from spconv_amd.pytorch.static import StaticTrainingStep, strided_layers
    from spconv_amd.utils import nets
    # (a) below wants the SAME launches captured and eager: a capture leaves out the rows layout of a module whose
    # rulebooks were of class 0 so far (ops.build_rulebook; an eager pass keeps building it to keep learning the class),
    # which changes the tiling of the convolutions' statistics records and with it the last bit of every BatchNorm
    monkeypatch.setattr(ops, "_LAYOUT_SKIP", False)
    shape, bs, C = [41, 96, 88], 2, 4
    torch.manual_seed(3)
    net = nets.second_backbone(C).to(cuda).train()
    eager = copy.deepcopy(net)
    f, idx = _scene_tensors(shape, 9000, bs, C, 2, cuda, torch.float32)
    with torch.no_grad():                                  # the layer bounds: what this scene produces, + 10 %
        counts = {}
        probe = copy.deepcopy(net)
        hs = [m.register_forward_hook(lambda mod, inp, out, nm=nm: counts.__setitem__(nm, out.features.shape[0]))
              for nm, m in strided_layers(probe).items()]
        probe(spconv.SparseConvTensor(f, idx, shape, bs))
        for h in hs:
            h.remove()
    bounds = {nm: int(c * 1.1) + 64 for nm, c in counts.items()}
    assert len(bounds) == 4
    n_last = counts[list(strided_layers(net))[-1]]
    g = (torch.rand((bounds[list(strided_layers(net))[-1]], 128), device=cuda) - 0.5) * 0.2
    step = StaticTrainingStep(net, f.shape[0] + 500, C, shape, bs, torch.float32, bounds=bounds, out_grad=g,
                              input_grad=False, example=(f, idx))
    out = step(f, idx)
    assert step.overflowed() == {}
    ye = eager(spconv.SparseConvTensor(f, idx, shape, bs))
    n_out = ye.features.shape[0]
    assert n_out == n_last and int(out.n_live_dev) == n_out and torch.equal(out.indices[:n_out], ye.indices)
    ye.features.backward(g[:n_out])
    err = float((out.features[:n_out] - ye.features).abs().max() / ye.features.detach().abs().max())
    assert err < 1e-5, err
    # (a) the captured step against the SAME static-shape pass run eagerly (padded input, frozen bounds, live-row
    # counts on the device): the same launches on the same rows -- every gradient bit for bit
    twin = copy.deepcopy(net)                      # (carries the frozen bounds; its own .grad tensors)
    twin.zero_grad(set_to_none=True)
    # (the runner sorts its scene by coordinate key at the entry -- static.py entry_sort --: the twin gets those rows)
    assert step.entry_sort
    o = step.order.long()
    xs = spconv.SparseConvTensor(step.features.detach()[o].clone(), step.indices[o].clone(), shape, bs)
    xs.n_live_dev = step.n_live
    ys = twin(xs)
    ys.features.backward(g)
    torch.cuda.synchronize()
    assert torch.equal(ys.features, out.features)
    for (name, pa), pb in zip(net.named_parameters(), twin.parameters()):
        assert torch.equal(pa.grad, pb.grad), name
    # (b) against the eager UNBOUNDED step.  The two passes tile the rows differently (padding rows, bounded levels), so
    # their BatchNorm statistics differ in the last bit (merge order) and every activation by ~1e-6: outputs agree to
    # 1e-5 (above).  Gradients are smooth in that EXCEPT where a pre-ReLU value sits within 1e-6 of zero and its mask
    # flips: one flipped element of ~1 M moves the sums behind it -- sums of a zero-mean gradient through twelve
    # normalisation layers, small against their terms -- by ~1 % (measured: 1 flip of 84 480 elements at |y| = 1.8e-6
    # -> 0.8 % in 0.weight; without a flip 3e-6).  Hence the bound here is the flip bound, and (a) is the sharp one.
    for (name, pa), pb in zip(net.named_parameters(), eager.parameters()):
        rel = float((pa.grad - pb.grad).norm() / pb.grad.norm().clamp_min(1e-20))
        assert rel < 3e-2, (name, rel)


@pytest.mark.parametrize("pool", [False, True])
def test_static_training_step_runner(cuda, pool):
    """StaticTrainingStep: one graph, several scenes; parameter gradients of every replay against the eager step.
    pool: the second strided layer is a SparseMaxPool3d -- pooling layers take their frozen bound in training mode
    too (the reference's bounded mode covers them: pool.py:99-248 via ops.py:263-266)."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch.static import StaticTrainingStep, strided_layers
    shape, bs, C = [32, 40, 40], 2, 8
    net = _backbone(spconv, C, cuda, torch.float16, pool=pool).train()
    before = {k: v.clone() for k, v in net.state_dict().items()}
    eager = copy.deepcopy(net)
    names = list(strided_layers(net))
    g = ((torch.rand((1_700, 64), device=cuda) - 0.5) * 0.2).half()
    scenes = [_scene_tensors(shape, n, bs, C, seed, cuda, torch.float16) for n, seed in ((4000, 1), (1500, 2), (5500, 3))]
    step = StaticTrainingStep(net, 12_000, C, shape, bs, torch.float16, bounds={names[0]: 13_000, names[1]: 1_700},
                              out_grad=g, input_grad=True, example=scenes[0])
    # building the runner leaves the model as it found it: the warm-up passes are real training-mode passes, the
    # buffers they moved (BatchNorm running estimates, batch counters) are put back
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k]), k
    for f, idx in scenes:
        out = step(f, idx)
        assert step.overflowed() == {}
        eager.zero_grad(set_to_none=True)
        fe = f.clone().requires_grad_(True)
        ye = eager(spconv.SparseConvTensor(fe, idx, shape, bs))
        ye.features.backward(g[:ye.features.shape[0]])
        n_out = ye.features.shape[0]
        assert int(out.n_live_dev) == n_out and torch.equal(out.indices[:n_out], ye.indices)
        for (name, pa), pb in zip(net.named_parameters(), eager.parameters()):
            rel = float((pa.grad.float() - pb.grad.float()).norm() / pb.grad.float().norm().clamp_min(1e-12))
            assert rel < 3e-2, (name, rel)
        rel = float((step.features.grad[:f.shape[0]].float() - fe.grad.float()).norm() / fe.grad.float().norm())
        assert rel < 3e-2, rel
    step.release_bounds()
    assert all(m.static_num_out == 0 for m in strided_layers(net).values())


def test_fold_sequential_eval_matches_the_unfolded_network(cuda):
    """quantization.utils.fold_sequential_eval: [conv, BatchNorm1d, ReLU] runs become one convolution each; the
    folded network equals the original in eval mode up to the fp16 rounding of the rescaled weights."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch.conv import SparseConvolution
    from spconv_amd.pytorch.quantization.utils import fold_sequential_eval
    shape, bs, C = [32, 40, 40], 2, 8
    net = _backbone(spconv, C, cuda, torch.float16)
    folded = fold_sequential_eval(net)
    kids = list(folded.children())
    assert sum(isinstance(m, SparseConvolution) for m in kids) == 6 and not any(isinstance(m, torch.nn.BatchNorm1d) for m in kids)
    assert len(kids) == 7          # only the ReLU behind the layer that has no BatchNorm is left as a module
    f, idx = _scene_tensors(shape, 3000, bs, C, 6, cuda, torch.float16)
    with torch.no_grad():
        a = net(spconv.SparseConvTensor(f, idx, shape, bs))
        b = folded(spconv.SparseConvTensor(f, idx, shape, bs))
    assert torch.equal(a.indices, b.indices)
    err = float((a.features.float() - b.features.float()).abs().max() / a.features.float().abs().max())
    assert err < 2e-2, err


def test_captured_unet_with_inverse_convolution(cuda):
    """Encoder / decoder: SubM -> strided conv (indice_key) -> SubM -> inverse conv back onto the input rows.
    The inverse layer reuses its partner's static rulebook; its output has the INPUT's live rows."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch.static import StaticInference, strided_layers
    from torch import nn
    shape, bs, C = [24, 32, 32], 2, 8
    torch.manual_seed(11)
    net = spconv.SparseSequential(
        spconv.SubMConv3d(C, 16, 3, bias=False, indice_key="s0"), nn.BatchNorm1d(16), nn.ReLU(),
        spconv.SparseConv3d(16, 32, 3, 2, 1, bias=False, indice_key="down"), nn.BatchNorm1d(32), nn.ReLU(),
        spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="s1"), nn.ReLU(),
        spconv.SparseInverseConv3d(32, 16, 3, indice_key="down", bias=False), nn.BatchNorm1d(16), nn.ReLU(),
        spconv.SubMConv3d(16, 16, 3, bias=True, indice_key="s0"),
    ).to(cuda).half().eval()
    eager = copy.deepcopy(net)
    (name, _), = strided_layers(net).items()
    runner = StaticInference(net, 9_000, C, shape, bs, torch.float16, bounds={name: 9_000})
    for n, seed in ((3000, 1), (4400, 2), (900, 3)):
        f, idx = _scene_tensors(shape, n, bs, C, seed, cuda, torch.float16)
        with torch.no_grad():
            want = eager(spconv.SparseConvTensor(f, idx, shape, bs))
        got = runner(f, idx)
        assert runner.overflowed() == {}
        n_rows = f.shape[0]
        assert int(got.n_live_dev) == n_rows and want.features.shape[0] == n_rows
        assert torch.equal(got.indices[:n_rows], idx) and bool((got.indices[n_rows:, 0] < 0).all())
        assert torch.equal(got.features[:n_rows], want.features)
