"""GPU parity: the encoder's per-point stem (model5_b.py:447-448: mlp1 -> BatchNorm1d(num_points) -> relu -> mlp2 ->
BatchNorm1d(num_points) -> relu) as ONE HIP launch each way (csrc/stem.hip, ops.stem) against the same modules in torch
fp64: output, every parameter gradient, running statistics and batch counters; train and eval mode, ragged batch sizes."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _modules(N, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    lin1, lin2, bn1, bn2 = nn.Linear(3, 64), nn.Linear(64, 64), nn.BatchNorm1d(N), nn.BatchNorm1d(N)
    with torch.no_grad():
        lin1.weight.copy_(torch.randn(64, 3, generator=g) * 0.6)
        lin1.bias.copy_(torch.randn(64, generator=g) * 0.2)
        lin2.weight.copy_(torch.randn(64, 64, generator=g) * 0.15)
        lin2.bias.copy_(torch.randn(64, generator=g) * 0.2)
        for bn in (bn1, bn2):
            bn.weight.copy_(1 + 0.3 * torch.randn(N, generator=g))
            bn.bias.copy_(0.2 * torch.randn(N, generator=g))
            bn.running_mean.copy_(0.1 * torch.randn(N, generator=g))
            bn.running_var.copy_(1 + 0.2 * torch.rand(N, generator=g))
    return [m.to(dtype) for m in (lin1, bn1, lin2, bn2)]


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("B,N,training", [(32, 1024, True), (32, 1024, False), (5, 96, True), (64, 40, True), (1, 33, False),
                                          (9, 300, True)])
def test_stem_vs_torch(B, N, training):
    from puzzlenet_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7 * B + N)
    xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
    go = torch.randn(B, N, 64, generator=g)
    ref = _modules(N, N, torch.float64)
    mine = [m.to(dev) for m in _modules(N, N, torch.float32)]
    for m in ref + mine:
        m.train(training)
    assert ops.stem_supported(xyz.to(dev), *mine)
    y = ops.stem(xyz.to(dev), *mine)
    lin1, bn1, lin2, bn2 = ref
    yr = F.relu(bn2(lin2(F.relu(bn1(lin1(xyz.double()))))))
    assert y.shape == (B, N, 64) and y.dtype == torch.float32
    assert float((y.cpu().double() - yr).abs().max()) < 5e-5 * max(1.0, float(yr.abs().max()))
    (yr * go.double()).sum().backward()
    (y * go.to(dev)).sum().backward()
    names = ("lin1", "bn1", "lin2", "bn2")
    for name, a, b in zip(names, mine, ref):
        # a pre-activation within rounding of zero may gate differently in fp32: the norms absorb the odd element
        assert _rel(a.weight.grad, b.weight.grad) < 2e-4, (name, "weight", _rel(a.weight.grad, b.weight.grad))
        assert _rel(a.bias.grad, b.bias.grad) < 2e-4, (name, "bias", _rel(a.bias.grad, b.bias.grad))
    for a, b in ((mine[1], bn1), (mine[3], bn2)):
        assert _rel(a.running_mean, b.running_mean) < 1e-5 and _rel(a.running_var, b.running_var) < 1e-5
        assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == (1 if training else 0)


def test_stem_matches_the_unfused_ops():
    """Same modules through ops.linear + ops.bn_points_relu (the path the fused stem replaces): outputs and parameter gradients
    agree to fp32 rounding, and a second backward accumulates into .grad like any autograd function."""
    from puzzlenet_amd import ops
    dev = torch.device("cuda:0")
    B, N = 32, 1024
    g = torch.Generator().manual_seed(3)
    xyz = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(dev)
    go = torch.randn(B, N, 64, generator=g).to(dev)
    a = [m.to(dev).train() for m in _modules(N, 11, torch.float32)]
    b = [m.to(dev).train() for m in _modules(N, 11, torch.float32)]
    ya = ops.stem(xyz, *a)
    yb = ops.bn_points_relu(ops.linear(ops.bn_points_relu(ops.linear(xyz, b[0].weight, b[0].bias), b[1]), b[2].weight, b[2].bias), b[3])
    assert float((ya - yb).abs().max()) < 2e-5
    (ya * go).sum().backward()
    (yb * go).sum().backward()
    for name, ma, mb in zip(("lin1", "bn1", "lin2", "bn2"), a, b):
        # (both sides fp32 with their own summation orders; the fused side alone is held to 2e-4 of fp64 above)
        assert _rel(ma.weight.grad, mb.weight.grad) < 1e-3, (name, _rel(ma.weight.grad, mb.weight.grad))
        assert _rel(ma.bias.grad, mb.bias.grad) < 1e-3, (name, _rel(ma.bias.grad, mb.bias.grad))
    first = [m.weight.grad.clone() for m in a]
    (ops.stem(xyz, *a) * go).sum().backward()
    for m, f in zip(a, first):
        assert _rel(m.weight.grad, 2 * f) < 1e-3      # (the running statistics moved; the batch statistics did not)


def test_stem_two_names_add_their_gradients_in_the_launch():
    """ops.stem(two=True) returns the features under two names; gradients arriving through both are added inside the backward
    launch: the parameter gradients equal those of one name fed the sum, and one unused name is as good as none."""
    from puzzlenet_amd import ops
    dev = torch.device("cuda:0")
    B, N = 7, 130
    g = torch.Generator().manual_seed(5)
    xyz = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(dev)
    ga, gb = (torch.randn(B, N, 64, generator=g).to(dev) for _ in range(2))
    one = [m.to(dev).train() for m in _modules(N, 2, torch.float32)]
    two = [m.to(dev).train() for m in _modules(N, 2, torch.float32)]
    only = [m.to(dev).train() for m in _modules(N, 2, torch.float32)]
    y = ops.stem(xyz, *one)
    (y * (ga + gb)).sum().backward()
    a, b = ops.stem(xyz, *two, two=True)
    assert a.data_ptr() == b.data_ptr() and torch.equal(a, y)
    ((a * ga).sum() + (b * gb).sum()).backward()
    _, b2 = ops.stem(xyz, *only, two=True)
    (b2 * (ga + gb)).sum().backward()
    for m1, m2, m3 in zip(one, two, only):
        assert _rel(m2.weight.grad, m1.weight.grad) < 1e-5 and _rel(m2.bias.grad, m1.bias.grad) < 1e-5
        assert _rel(m3.weight.grad, m1.weight.grad) < 1e-5 and _rel(m3.bias.grad, m1.bias.grad) < 1e-5


def test_stem_refuses_coordinate_gradients():
    from puzzlenet_amd import ops
    dev = torch.device("cuda:0")
    xyz = torch.rand(4, 64, 3, device=dev, requires_grad=True)
    mods = [m.to(dev) for m in _modules(64, 1, torch.float32)]
    assert not ops.stem_supported(xyz, *mods)


def test_model_gradients_fused_stem_equals_unfused_on_the_same_values(golden_loss):
    """One training_step of the whole model with the one-launch stem, and again with the four launches it replaces whose
    output VALUES are replaced by the first run's (the gradient still flows through the four-launch backward): the gradient
    arriving at the per-point features and every stem parameter's gradient agree to fp32 rounding.  (Without the
    replacement the two runs differ in the last bits of the features, and on this batch a max-pool winner downstream flips:
    see test_training_step_full_gradients_vs_oracle.)"""
    import numpy as np
    from oracle import model_ref as mr
    from puzzlenet_amd import model5_b as mb, ops
    dev = torch.device("cuda:0")
    cfg = mr.Cfg(loss_mode=0)
    batch = [torch.from_numpy(np.ascontiguousarray(golden_loss[f"ts_batch{i}"])).to(dev) for i in range(8)]
    runs = []
    was, was_two = mb._STEM_FUSED, mb._STEM_TWO
    mb._STEM_TWO = False      # one name for the features: the hook below sees their whole gradient
    try:
        for fused in (True, False):
            mb._STEM_FUSED = fused
            ops.clear_grad_sinks()
            m = mb.TouchedRegraster(cfg)
            mr.fill_params(m)
            m.to(dev)
            cap = {}
            for name in ("Encoder", "Encoder2"):
                enc = getattr(m, name)

                def wrapped(xyz, orig=enc.local_features, name=name):
                    y = orig(xyz)
                    if runs:
                        y = y + (runs[0][name + ".xf"] - y).detach()
                    cap[name + ".xf"] = y.detach().clone()
                    y.register_hook(lambda g, name=name: cap.__setitem__(name + ".dxf", g.detach().clone()))
                    return y
                enc.local_features = wrapped
            torch.manual_seed(99)
            m.training_step(batch, 0)["loss"].backward()
            torch.cuda.synchronize()
            for n_, p in m.named_parameters():
                if p.grad is not None:
                    cap[n_] = p.grad.detach().clone()
            runs.append(cap)
    finally:
        mb._STEM_FUSED, mb._STEM_TWO = was, was_two
    a, b = runs
    for name in ("Encoder", "Encoder2"):
        assert float((a[name + ".xf"] - b[name + ".xf"]).abs().max()) < 1e-6
        assert _rel(a[name + ".dxf"], b[name + ".dxf"]) < 1e-5, (name, _rel(a[name + ".dxf"], b[name + ".dxf"]))
        for part in ("mlp1", "bn1", "mlp2", "bn2", "mlp3", "mlp5"):
            for leaf in ("weight", "bias"):
                k = f"{name}.{part}.{leaf}"
                assert _rel(a[k], b[k]) < 1e-4, (k, _rel(a[k], b[k]))
