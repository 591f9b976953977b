"""GPU: rulebooks of a layer chain built ahead of the layers on a side stream (spconv_amd/pytorch/prefetch.py).

The reference builds a rulebook where a layer first needs it (spconv/pytorch/conv.py:247-278); the rulebook depends on the
coordinates alone (ops.py:132-326 takes `indices`, never `features`), so the container may build the whole chain's
rulebooks ahead.  Bar: bit-identical outputs and gradients with and without the side stream; a prefetched rulebook is
only taken by a layer that is called with exactly the inputs it was built from."""
import copy

import numpy as np
import pytest
import torch

from util import scene

pytestmark = pytest.mark.gpu


def _net(spconv, dev, dtype=torch.float16):
    from torch import nn
    torch.manual_seed(11)
    return spconv.SparseSequential(
        spconv.SubMConv3d(8, 16, 3, bias=False, indice_key="s0"), nn.BatchNorm1d(16), nn.ReLU(),
        spconv.SubMConv3d(16, 16, 3, bias=False, indice_key="s0"), nn.BatchNorm1d(16), nn.ReLU(),
        spconv.SparseSequential(                                           # (a nested plain container is opened up)
            spconv.SparseConv3d(16, 32, 3, 2, 1, bias=False, indice_key="d1"), nn.BatchNorm1d(32), nn.ReLU()),
        spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="s1"), nn.ReLU(),
        spconv.SparseConv3d(32, 64, 3, 2, 1, bias=False, indice_key="d2"), nn.ReLU(),
        spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="s2"),
    ).to(dev).to(dtype)


def _data(dev, dtype, seed=3, n=5000):
    shape, bs = [32, 40, 40], 2
    idx = scene(shape, n, bs, seed)
    rng = np.random.default_rng(seed)
    f = torch.from_numpy(rng.uniform(-1, 1, (idx.shape[0], 8)).astype(np.float32)).to(dev, dtype)
    return f, torch.from_numpy(idx).to(dev), shape, bs


def _count_takes(monkeypatch):
    from spconv_amd.pytorch import prefetch
    hits = {"taken": 0, "asked": 0}
    real = prefetch.take

    def take(module, indices, batch_size, spatial_shape):
        rb = real(module, indices, batch_size, spatial_shape)
        hits["asked"] += 1
        hits["taken"] += rb is not None
        return rb
    monkeypatch.setattr(prefetch, "take", take)
    return hits


def test_plan_covers_the_chain_only_when_nothing_reads_back(cuda):
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch import prefetch
    net = _net(spconv, cuda)
    f, idx, shape, bs = _data(cuda, torch.float16)
    x = spconv.SparseConvTensor(f, idx, shape, bs)
    mods = list(net._modules.values())
    # strided layers without a bound read their output count back: only the first SubM build could go ahead
    assert [m.indice_key for m in prefetch._plan(mods, x)] == ["s0"]
    for m in net.modules():
        if isinstance(m, spconv.SparseConv3d):
            m.static_num_out = 20_000
    assert [m.indice_key for m in prefetch._plan(mods, x)] == ["s0", "d1", "s1", "d2", "s2"]
    prev = prefetch.set_mode("1")
    try:
        for m in net.modules():
            if isinstance(m, spconv.SparseConv3d):
                m.static_num_out = 0
        assert [m.indice_key for m in prefetch._plan(mods, x)] == ["s0", "d1", "s1", "d2", "s2"]
    finally:
        prefetch.set_mode(prev)


@pytest.mark.parametrize("train", [False, True])
def test_forced_prefetch_is_bit_identical_in_an_eager_pass(cuda, monkeypatch, train):
    """SPCONV_AMD_PREFETCH=1: every rulebook of the chain (strided ones with their read-back) comes from the side stream;
    outputs, input gradient and every weight gradient equal the in-line pass bit for bit."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch import prefetch
    net = _net(spconv, cuda).train(train)
    ref = copy.deepcopy(net)
    f, idx, shape, bs = _data(cuda, torch.float16)

    def run(model):
        fe = f.clone().requires_grad_(train)
        y = model(spconv.SparseConvTensor(fe, idx, shape, bs))
        if train:
            y.features.backward(torch.ones_like(y.features) * 0.01)
        torch.cuda.synchronize()
        return y, fe.grad

    prev = prefetch.set_mode("0")
    try:
        want, din_want = run(ref)
        prefetch.set_mode("1")
        hits = _count_takes(monkeypatch)
        got, din_got = run(net)
    finally:
        prefetch.set_mode(prev)
    assert hits == {"taken": 5, "asked": 5}
    assert torch.equal(got.indices, want.indices) and torch.equal(got.features, want.features)
    if train:
        assert torch.equal(din_got, din_want)
        for (name, pa), pb in zip(net.named_parameters(), ref.parameters()):
            assert torch.equal(pa.grad, pb.grad), name
    for m in net.modules():                                   # nothing left behind on the modules
        assert "_spx_prefetched" not in m.__dict__


def test_a_rulebook_built_for_other_inputs_is_not_taken(cuda, monkeypatch):
    """A module between two layers that hands on ANOTHER index tensor (here: a copy) invalidates what was built ahead
    for the layers behind it: they build in line, the result is the in-line result."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch import prefetch
    from torch import nn

    class Rewrap(nn.Module):                                  # dense-looking module that is really a sparse one
        def forward(self, feats):
            return feats

    class CopyIndices(spconv.SparseModule):
        def forward(self, x):
            y = x.shadow_copy()
            y.indices = x.indices.clone()
            y.indice_dict = {}
            return y

    torch.manual_seed(5)
    a = spconv.SubMConv3d(8, 16, 3, bias=False, indice_key="a").to(cuda).half()
    b = spconv.SubMConv3d(16, 16, 3, bias=False, indice_key="b").to(cuda).half()
    net = spconv.SparseSequential(a, Rewrap(), b)
    f, idx, shape, bs = _data(cuda, torch.float16)
    prev = prefetch.set_mode("1")
    try:
        hits = _count_takes(monkeypatch)
        with torch.no_grad():
            y1 = net(spconv.SparseConvTensor(f, idx, shape, bs))
        assert hits == {"taken": 2, "asked": 2}
        # the same two layers with the coordinates swapped for a copy in between: the plan stops at the sparse module
        net2 = spconv.SparseSequential(a, CopyIndices(), b)
        hits["taken"] = hits["asked"] = 0
        with torch.no_grad():
            y2 = net2(spconv.SparseConvTensor(f, idx, shape, bs))
        assert hits["taken"] == 0                             # (one build in the plan: nothing goes ahead)
        # and a stale hint is refused: built for `idx`, offered with a clone of it
        p = prefetch._Prefetched()
        p.rb, p.indices, p.batch_size, p.spatial_shape, p.event = object(), idx, bs, list(shape), torch.cuda.Event()
        b.__dict__["_spx_prefetched"] = p
        assert prefetch.take(b, idx.clone(), bs, shape) is None and "_spx_prefetched" not in b.__dict__
    finally:
        prefetch.set_mode(prev)
    assert torch.equal(y1.features, y2.features)


def test_captured_pass_takes_every_rulebook_from_the_side_stream(cuda, monkeypatch):
    """Default mode: inside a stream capture (StaticInference) the chain's rulebooks are built on the side branch of
    the graph; the replayed pass equals the eager, unbounded, in-line pass bit for bit on several scenes."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch.static import StaticInference, strided_layers
    net = _net(spconv, cuda).eval()
    eager = copy.deepcopy(net)
    names = list(strided_layers(net))
    hits = _count_takes(monkeypatch)
    runner = StaticInference(net, max_voxels=12_000, in_channels=8, spatial_shape=[32, 40, 40], batch_size=2,
                             dtype=torch.float16, bounds={names[0]: 13_000, names[1]: 2_500}, warmup=1)
    # one eager warm-up pass (in line: 5 asked, 0 taken) + the captured pass (5 of 5 taken)
    assert hits == {"taken": 5, "asked": 10}
    for n, seed in ((4500, 1), (2001, 2), (5999, 3)):
        f, idx, shape, bs = _data(cuda, torch.float16, seed, n)
        with torch.no_grad():
            want = eager(spconv.SparseConvTensor(f, idx, shape, bs))
        got = runner(f, idx)
        assert runner.overflowed() == {}
        k = want.indices.shape[0]
        assert torch.equal(got.indices[:k], want.indices) and torch.equal(got.features[:k], want.features)
    runner.release_bounds()
