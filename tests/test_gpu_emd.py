"""GPU parity for the EMD path (PyTorchEMD): HIP kernels vs the CPU restatement
of emd_kernel.cu, the reference's only known-answer test (test_emd_loss.py:8-25),
and structural invariants.  Tolerance: 1e-4 relative (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from oracle import point_ops as orc

pytestmark = pytest.mark.gpu
RTOL = 1e-4          # cost (the loss value): the bar BASELINE.json states
# The auction is ill-conditioned in the individual match entries: a point with two
# nearly equidistant partners splits its mass on rounding noise.  Two fp32 runs of
# the SAME algorithm (CPU restatement in fp32 vs fp64) already differ by ~2e-4 of
# max|match| and ~3e-4 of max|grad| on such points, while the cost agrees to 1e-6.
# Measured on MI355X, fused path vs three-call path (same algorithm, different
# instruction schedules) at n=m=2048: worst entry 3e-3 of max|grad|, 0.004 % of
# entries above 1e-3, L2 2.7e-4, cost 5e-6.  Entry-wise quantities are therefore
# held to 1e-2 of the max, 1e-3 in L2 and <0.1 % of entries off by more than 1e-3.
MATCH_TOL = 5e-3
GRAD_MAX_TOL = 1e-2
GRAD_L2_TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def _grad_close(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    frac = (np.abs(a - b) > 1e-3 * np.abs(b).max()).mean()
    return _rel(a, b) < GRAD_MAX_TOL and _l2(a, b) < GRAD_L2_TOL and frac < 1e-3


def test_known_answer_from_reference_test(dev):
    """PyTorchEMD/test_emd_loss.py:8-25 (commented-out KAT): 2-point clouds, the
    optimal assignment is the cross one, cost 0.30 + 0.41 = 0.71 per item;
    loss = d0/2 + 2 d1 + d2/3 and its autograd gradients."""
    from puzzlenet_amd.PyTorchEMD.emd import earth_mover_distance
    p1 = np.array([[[1.7, -0.1, 0.1], [0.1, 1.2, 0.3]]], dtype=np.float32).repeat(3, 0)
    p2 = np.array([[[0.3, 1.8, 0.2], [1.2, -0.2, 0.3]]], dtype=np.float32).repeat(3, 0)
    for materialize in (False, True):
        t1 = _t(p1, dev).requires_grad_(True)
        t2 = _t(p2, dev).requires_grad_(True)
        d = earth_mover_distance(t1, t2, transpose=False, materialize_match=materialize)
        np.testing.assert_allclose(d.detach().cpu().numpy(), [0.71] * 3, rtol=1e-4)
        loss = d[0] / 2 + d[1] * 2 + d[2] / 3
        np.testing.assert_allclose(float(loss.detach()), 0.71 * (0.5 + 2 + 1 / 3), rtol=1e-4)
        loss.backward()
        w = np.array([0.5, 2.0, 1 / 3], np.float32)[:, None, None]
        g1 = 2 * (p1 - p2[:, ::-1]) * w          # p1_0 <-> p2_1, p1_1 <-> p2_0
        g2 = 2 * (p2 - p1[:, ::-1]) * w
        np.testing.assert_allclose(t1.grad.cpu().numpy(), g1, rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(t2.grad.cpu().numpy(), g2, rtol=1e-3, atol=1e-4)


def test_hand_derived_capacity_vectors(dev):
    """tests/emd_vectors.py: clustered clouds whose matching can be written down by hand; they exercise the
    integer capacities multiL / multiR of emd_kernel.cu:29-35 (n = 6, m = 4 -> 1, not 1.5) and n != m on the
    three-call path (match itself), the fused path and the drop-in autograd function."""
    from puzzlenet_amd import emd_cuda, ops
    from puzzlenet_amd.PyTorchEMD.emd import earth_mover_distance
    from tests import emd_vectors as ev
    for case in ev.CASES:
        x1, x2, match, cost = case()
        t1, t2 = _t(x1, dev), _t(x2, dev)
        m = emd_cuda.approxmatch_forward(t1, t2)
        np.testing.assert_allclose(m.cpu().numpy()[0], match, atol=5e-5, err_msg=case.__name__)
        c3 = emd_cuda.matchcost_forward(t1, t2, m)
        np.testing.assert_allclose(c3.cpu().numpy(), [cost], rtol=RTOL, err_msg=case.__name__)
        w1, w2 = ev.gradients(x1, x2, match, 1.5)
        g1, g2 = emd_cuda.matchcost_backward(_t(np.array([1.5], np.float32), dev), t1, t2, m)
        np.testing.assert_allclose(g1.cpu().numpy(), w1, rtol=1e-4, atol=2e-4, err_msg=case.__name__)
        np.testing.assert_allclose(g2.cpu().numpy(), w2, rtol=1e-4, atol=2e-4, err_msg=case.__name__)
        for fn in (ops.emd_fused, lambda a, b: earth_mover_distance(a, b, transpose=False)):
            a, b = t1.clone().requires_grad_(True), t2.clone().requires_grad_(True)
            c = fn(a, b)
            np.testing.assert_allclose(c.detach().cpu().numpy(), [cost], rtol=RTOL, err_msg=case.__name__)
            (1.5 * c).sum().backward()
            np.testing.assert_allclose(a.grad.cpu().numpy(), w1, rtol=1e-4, atol=2e-4, err_msg=case.__name__)
            np.testing.assert_allclose(b.grad.cpu().numpy(), w2, rtol=1e-4, atol=2e-4, err_msg=case.__name__)


@pytest.mark.parametrize("B,n,m", [(3, 128, 128), (2, 64, 64), (2, 256, 128), (2, 100, 300), (1, 513, 200), (2, 1024, 1024)])
def test_three_call_path_vs_oracle(dev, B, n, m):
    from puzzlenet_amd import emd_cuda
    rng = np.random.default_rng(n * 3 + m)
    x1 = rng.random((B, n, 3), dtype=np.float32)
    x2 = rng.random((B, m, 3), dtype=np.float32)
    match = emd_cuda.approxmatch_forward(_t(x1, dev), _t(x2, dev))
    assert match.shape == (B, m, n)
    omatch = orc.emd_approxmatch(x1, x2)
    assert _rel(match.cpu().numpy(), omatch) < MATCH_TOL
    cost = emd_cuda.matchcost_forward(_t(x1, dev), _t(x2, dev), match)
    assert _rel(cost.cpu().numpy(), orc.emd_matchcost(x1, x2, omatch)) < RTOL
    # matchcost alone, on identical match input: pure summation, tight
    cost_o = emd_cuda.matchcost_forward(_t(x1, dev), _t(x2, dev), _t(omatch, dev))
    assert _rel(cost_o.cpu().numpy(), orc.emd_matchcost(x1, x2, omatch)) < 1e-5
    gc = rng.standard_normal(B).astype(np.float32)
    g1, g2 = emd_cuda.matchcost_backward(_t(gc, dev), _t(x1, dev), _t(x2, dev), _t(omatch, dev))
    o1, o2 = orc.emd_matchcost_grad(gc, x1, x2, omatch)
    assert _rel(g1.cpu().numpy(), o1) < RTOL and _rel(g2.cpu().numpy(), o2) < RTOL


@pytest.mark.parametrize("near", [False, True])
@pytest.mark.parametrize("B,n,m", [(3, 128, 128), (4, 64, 64), (2, 256, 128), (2, 100, 300), (2, 1024, 1024), (1, 2048, 2048),
                                   (1, 513, 200), (2, 300, 1000)])
def test_fused_path_vs_oracle(dev, B, n, m, near):
    """near=False: independent uniform clouds (what an untrained model produces), cost to 1e-4.
    near=True: xyz2 is a jittered permutation of xyz1 (a converged registration): the optimum is a
    near-permutation whose cost is a sum of ~1e-3-sized terms, and ONE near-tie that resolves the
    other way under different rounding moves the cost by up to ~3e-4.  That is a property of the
    ALGORITHM, not of the kernels: the same C restatement run in float32 and in float64 differs by as
    much on such pairs.  So the kernel's cost must lie within 1e-4 of the float32 restatement OR
    within twice the float32 <-> float64 spread of the restatement itself (per pair), and never
    further than 1e-3 from either."""
    from puzzlenet_amd import ops
    rng = np.random.default_rng(n * 5 + m)
    x1 = rng.random((B, n, 3), dtype=np.float32)
    if near:
        x2 = (x1[:, rng.permutation(n)[:m] if m <= n else rng.integers(0, n, m)] +
              0.02 * rng.standard_normal((B, m, 3))).astype(np.float32)
    else:
        x2 = rng.random((B, m, 3), dtype=np.float32)
    t1, t2 = _t(x1, dev).requires_grad_(True), _t(x2, dev).requires_grad_(True)
    cost = ops.emd_fused(t1, t2)
    ocost, omatch = orc.earth_mover_distance(x1, x2)
    got = cost.detach().cpu().numpy().astype(np.float64)
    if not near:
        assert _rel(got, ocost) < RTOL
    else:
        o64, _ = orc.earth_mover_distance(x1.astype(np.float64), x2.astype(np.float64))
        err32 = np.abs(got - ocost) / np.abs(o64)
        err64 = np.abs(got - o64) / np.abs(o64)
        spread = np.abs(ocost.astype(np.float64) - o64) / np.abs(o64)      # the restatement against itself
        ok = (err32 < RTOL) | (err64 < RTOL) | (np.minimum(err32, err64) <= 2 * spread)
        assert ok.all() and max(err32.max(), err64.max()) < 1e-3, (err32, err64, spread)
    gc = rng.standard_normal(B).astype(np.float32)
    (cost * _t(gc, dev)).sum().backward()
    o1, o2 = orc.emd_matchcost_grad(gc, x1, x2, omatch)
    assert _grad_close(t1.grad.cpu().numpy(), o1)
    assert _grad_close(t2.grad.cpu().numpy(), o2)


def test_fused_path_exhausted_points(dev):
    """The general path walks only the points of cloud 2 that still hold mass (emd.hip, emd_compact_kernel).  Clouds
    built so that the active list shrinks unevenly: cloud 2 = a few tight clusters (whole clusters run out of mass at
    the sharp levels) plus duplicates of single points, and one pair whose cloud 2 is ONE point repeated."""
    from puzzlenet_amd import ops
    rng = np.random.default_rng(77)
    B, n, m = 3, 640, 384
    x1 = rng.random((B, n, 3), dtype=np.float32)
    centres = rng.random((B, 6, 3), dtype=np.float32)
    x2 = (centres[:, rng.integers(0, 6, m)] + 0.003 * rng.standard_normal((B, m, 3))).astype(np.float32)
    x2[:, 100:140] = x2[:, 100:101]          # 40 copies of one point
    x2[2] = x2[2, :1]                        # a degenerate cloud
    t1, t2 = _t(x1, dev).requires_grad_(True), _t(x2, dev).requires_grad_(True)
    cost = ops.emd_fused(t1, t2)
    ocost, omatch = orc.earth_mover_distance(x1, x2)
    assert _rel(cost.detach().cpu().numpy(), ocost) < RTOL
    gc = np.ones(B, np.float32)
    cost.sum().backward()
    o1, o2 = orc.emd_matchcost_grad(gc, x1, x2, omatch)
    assert _grad_close(t1.grad.cpu().numpy(), o1)
    assert _grad_close(t2.grad.cpu().numpy(), o2)


def test_invariants_full_size(dev):
    """B=64, N=2048 (BASELINE configs[1]): too big for the scalar oracle; check
    what must hold for any correct auction."""
    from puzzlenet_amd import emd_cuda, ops
    g = torch.Generator().manual_seed(1)
    B, n = 64, 2048
    x1 = torch.rand(B, n, 3, generator=g).to(dev)
    x2 = torch.rand(B, n, 3, generator=g).to(dev)
    match = emd_cuda.approxmatch_forward(x1, x2)
    assert bool((match >= 0).all())
    assert float(match.sum(1).max()) <= 1 + 1e-4          # sum_l match[l,k] <= multiL = 1
    assert float(match.sum(2).max()) <= 1 + 1e-4          # sum_k match[l,k] <= multiR = 1
    assert float(match.sum((1, 2)).min()) > 0.99 * n      # (almost) all mass assigned after 10 levels
    cost3 = emd_cuda.matchcost_forward(x1, x2, match)
    costf = ops.emd_fused(x1, x2)
    assert _rel(costf.cpu().numpy(), cost3.cpu().numpy()) < RTOL
    # a cloud matched with a permutation of itself costs ~0
    perm = torch.randperm(n, generator=g).to(dev)
    assert float(ops.emd_fused(x1, x1[:, perm]).max()) < 1e-3 * float(costf.min())
    # fused gradients == the reference's three-call gradients
    ones = torch.ones(B, device=dev)
    g1, g2 = emd_cuda.matchcost_backward(ones, x1, x2, match)
    t1, t2 = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    ops.emd_fused(t1, t2).sum().backward()
    assert _grad_close(t1.grad.cpu().numpy(), g1.cpu().numpy())
    assert _grad_close(t2.grad.cpu().numpy(), g2.cpu().numpy())


def test_dropin_signature(dev):
    from puzzlenet_amd.PyTorchEMD.emd import earth_mover_distance
    a = torch.rand(2, 3, 50, device=dev)
    b = torch.rand(2, 3, 50, device=dev)
    c_t = earth_mover_distance(a, b)                                   # transpose=True default: (b,3,n)
    c_n = earth_mover_distance(a.transpose(1, 2), b.transpose(1, 2), transpose=False)
    assert torch.allclose(c_t, c_n)
    assert earth_mover_distance(a[0].t(), b[0].t(), transpose=False).shape == (1,)   # 2-D inputs are unsqueezed
    with pytest.raises(AssertionError):
        earth_mover_distance(a.cpu(), b.cpu())


@pytest.mark.parametrize("B,n,m", [(3, 96, 80), (2, 64, 64), (2, 100, 300), (1, 513, 200), (1, 700, 700)])
def test_double_instantiation_vs_oracle(dev, B, n, m):
    """emd_kernel.cu:187,273,391 instantiate the kernels for float AND double.  Double clouds run the double kernels
    (csrc/emd64.hip: the three calls, sequential sums in the reference's loop order) and are held to the oracle's f64
    instantiation far below fp32 resolution: match 1e-9 of its maximum, cost 1e-12 relative, gradients 1e-9 - which an
    fp32 computation cast to double cannot meet; the dtypes of cost and gradients stay double through autograd."""
    from puzzlenet_amd import emd_cuda
    from puzzlenet_amd.PyTorchEMD.emd import earth_mover_distance
    rng = np.random.default_rng(100 * n + m)
    x1 = rng.random((B, n, 3))
    x2 = rng.random((B, m, 3))
    omatch = orc.emd_approxmatch(x1, x2)
    ocost = orc.emd_matchcost(x1, x2, omatch)
    gc = rng.random(B) + 0.5
    o1, o2 = orc.emd_matchcost_grad(gc, x1, x2, omatch)
    t1, t2 = _t(x1, dev), _t(x2, dev)
    assert t1.dtype == torch.float64
    match = emd_cuda.approxmatch_forward(t1, t2)
    assert match.dtype == torch.float64 and match.shape == (B, m, n)
    assert _rel(match.cpu().numpy(), omatch) < 1e-9
    cost = emd_cuda.matchcost_forward(t1, t2, match)
    assert cost.dtype == torch.float64 and _rel(cost.cpu().numpy(), ocost) < 1e-12 + 1e-9
    g1, g2 = emd_cuda.matchcost_backward(_t(gc, dev), t1, t2, _t(omatch, dev))        # (on the oracle's match: the kernel alone)
    assert _rel(g1.cpu().numpy(), o1) < 1e-12 and _rel(g2.cpu().numpy(), o2) < 1e-12
    # through the drop-in: double in, double out, gradients of sum(gc * cost)
    a, b = t1.clone().requires_grad_(True), t2.clone().requires_grad_(True)
    c = earth_mover_distance(a, b, transpose=False)
    assert c.dtype == torch.float64 and c.shape == (B,)
    (c * _t(gc, dev)).sum().backward()
    assert a.grad.dtype == torch.float64 and b.grad.dtype == torch.float64
    assert _rel(c.detach().cpu().numpy(), ocost) < 1e-9
    assert _rel(a.grad.cpu().numpy(), o1) < 1e-8 and _rel(b.grad.cpu().numpy(), o2) < 1e-8
    # ... and it is NOT the fp32 computation cast back: that one is off by fp32 rounding
    c32 = earth_mover_distance(t1.float(), t2.float(), transpose=False).double()
    assert _rel(c32.cpu().numpy(), ocost) > 1e-9


def test_double_known_answer(dev):
    """The reference's only vector (test_emd_loss.py:8-25) in double: cost 0.71 per item."""
    from puzzlenet_amd.PyTorchEMD.emd import earth_mover_distance
    p1 = torch.tensor([[[1.7, -0.1, 0.1], [0.1, 1.2, 0.3]]], dtype=torch.float64, device=dev).repeat(3, 1, 1)
    p2 = torch.tensor([[[0.3, 1.8, 0.2], [1.2, -0.2, 0.3]]], dtype=torch.float64, device=dev).repeat(3, 1, 1)
    d = earth_mover_distance(p1, p2, transpose=False)
    assert d.dtype == torch.float64
    np.testing.assert_allclose(d.cpu().numpy(), [0.71] * 3, rtol=1e-6)


def test_small_calls_in_one_launch(dev):
    """ops.emd_fused_small_multi (pzn_emd_fused_small_multi_f32: the three small terms of the loss, model5_b.py:1012 and
    :1123-1125, as ONE launch of single-workgroup auctions) against the same calls made one by one: costs and both gradients
    bit-identical; shapes of the loss (64 x 64, 128 x 128 twice), a ragged one (n != m, different batch) and a single problem."""
    from puzzlenet_amd import ops
    g = torch.Generator().manual_seed(5)
    for shapes in ([(6, 64, 64), (6, 128, 128), (6, 128, 128)], [(3, 100, 50), (5, 17, 256), (2, 256, 256), (4, 1, 7)], [(2, 33, 33)]):
        pairs = [(torch.rand(B, n, 3, generator=g).to(dev).requires_grad_(True), torch.rand(B, m, 3, generator=g).to(dev).requires_grad_(True))
                 for B, n, m in shapes]
        ws = [torch.rand(B, generator=g).to(dev) for B, _, _ in shapes]
        costs = ops.emd_fused_small_multi(pairs)
        sum((c * w).sum() for c, w in zip(costs, ws)).backward()
        got = [(c.detach().clone(), a.grad.clone(), b.grad.clone()) for c, (a, b) in zip(costs, pairs)]
        for (a, b), w, (c1, ga, gb) in zip(pairs, ws, got):
            a.grad = b.grad = None
            c0 = ops.emd_fused(a, b)
            (c0 * w).sum().backward()
            assert torch.equal(c0, c1) and torch.equal(a.grad, ga) and torch.equal(b.grad, gb)
    big = [(torch.rand(1, 300, 3).to(dev), torch.rand(1, 300, 3).to(dev))]
    with pytest.raises(Exception):
        ops.emd_fused_small_multi(big)
