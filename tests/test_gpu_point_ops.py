"""GPU parity: the HIP kernels (through the C ABI, via puzzlenet_amd.ops /
the pointnet_util drop-in) against (1) the committed golden outputs of the
reference and (2) the CPU oracle on seeded inputs.  Indices bit-exact."""
import numpy as np
import pytest
import torch

from oracle import point_ops as orc

pytestmark = pytest.mark.gpu

SG_TAGS = ["a", "b", "c", "d", "e", "f"]
TIE_TAGS = {"b", "c", "d"}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from puzzlenet_amd import _lib
    assert _lib.load().pzn_device_check() == 0
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("tag", SG_TAGS)
def test_fps_golden(golden_point_ops, dev, tag):
    from puzzlenet_amd import ops
    G = golden_point_ops
    B, N, S, K, D = (int(v) for v in G[f"sg_{tag}_params"])
    want = G[f"sg_{tag}_fps_idx"]
    got = ops.farthest_point_sample(_t(G[f"sg_{tag}_xyz"], dev), S, _t(want[:, 0], dev))
    assert got.dtype == torch.int64
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("tag", SG_TAGS)
def test_knn_golden(golden_point_ops, dev, tag):
    from puzzlenet_amd import ops
    G = golden_point_ops
    B, N, S, K, D = (int(v) for v in G[f"sg_{tag}_params"])
    got = ops.knn(_t(G[f"sg_{tag}_xyz"], dev), _t(G[f"sg_{tag}_new_xyz"], dev), K).cpu().numpy()
    # the stable (distance, index) order == the oracle, always
    assert np.array_equal(got, orc.knn(G[f"sg_{tag}_xyz"], G[f"sg_{tag}_new_xyz"], K))
    if tag not in TIE_TAGS:
        assert np.array_equal(got, G[f"sg_{tag}_knn_idx"])          # == the reference itself


@pytest.mark.parametrize("tag", SG_TAGS)
def test_group_golden(golden_point_ops, dev, tag):
    from puzzlenet_amd import ops
    G = golden_point_ops
    B, N, S, K, D = (int(v) for v in G[f"sg_{tag}_params"])
    feat = _t(G[f"sg_{tag}_feat"], dev) if D else None
    out, gx = ops.group(_t(G[f"sg_{tag}_xyz"], dev), feat, _t(G[f"sg_{tag}_new_xyz"], dev),
                        _t(G[f"sg_{tag}_knn_idx"], dev), want_grouped_xyz=True)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), G[f"sg_{tag}_new_points"].view(np.uint32))
    assert np.array_equal(gx.cpu().numpy(), G[f"sg_{tag}_grouped_xyz"])


@pytest.mark.parametrize("tag", ["a", "e", "f"])
def test_sample_and_group_dropin_golden(golden_point_ops, dev, tag):
    """Full drop-in call, seeded exactly as the fixture generator seeded the reference."""
    import puzzlenet_amd.pointnet_util as pu
    G = golden_point_ops
    B, N, S, K, D = (int(v) for v in G[f"sg_{tag}_params"])
    feat = _t(G[f"sg_{tag}_feat"], dev) if D else None
    torch.manual_seed(100 + ord(tag))
    new_xyz, new_points, grouped_xyz, fps_idx = pu.sample_and_group(
        S, 0, K, _t(G[f"sg_{tag}_xyz"], dev), feat, returnfps=True, knn=True)
    assert np.array_equal(fps_idx.cpu().numpy(), G[f"sg_{tag}_fps_idx"])
    assert np.array_equal(new_xyz.cpu().numpy(), G[f"sg_{tag}_new_xyz"])
    assert np.array_equal(new_points.cpu().numpy().view(np.uint32), G[f"sg_{tag}_new_points"].view(np.uint32))
    assert np.array_equal(grouped_xyz.cpu().numpy(), G[f"sg_{tag}_grouped_xyz"])
    nx2, np2 = pu.sample_and_group(S, 0, K, _t(G[f"sg_{tag}_xyz"], dev), feat, knn=True)
    assert nx2.shape == new_xyz.shape and np2.shape == new_points.shape


@pytest.mark.parametrize("r", [0.1, 0.2, 0.37])
@pytest.mark.parametrize("ns", [8, 32])
def test_ball_query_golden(golden_point_ops, dev, r, ns):
    import puzzlenet_amd.pointnet_util as pu
    G = golden_point_ops
    got = pu.query_ball_point(r, ns, _t(G["ball_xyz"], dev), _t(G["ball_new_xyz"], dev))
    assert np.array_equal(got.cpu().numpy(), G[f"ball_r{r}_n{ns}"])


def test_ball_query_edge_and_dropin(golden_point_ops, dev):
    import puzzlenet_amd.pointnet_util as pu
    G = golden_point_ops
    r = float(G["ball_edge_radius"][0])
    got = pu.query_ball_point(r, 3, _t(G["ball_edge_xyz"], dev), _t(G["ball_edge_query"], dev))
    assert np.array_equal(got.cpu().numpy(), G["ball_edge_idx"])
    torch.manual_seed(7)
    o = pu.sample_and_group(32, 0.2, 16, _t(G["sgball_xyz"], dev), _t(G["sgball_feat"], dev), returnfps=True, knn=False)
    assert np.array_equal(o[3].cpu().numpy(), G["sgball_fps_idx"])
    assert np.array_equal(o[1].cpu().numpy().view(np.uint32), G["sgball_new_points"].view(np.uint32))
    assert np.array_equal(o[2].cpu().numpy(), G["sgball_grouped_xyz"])


def test_square_distance_golden(golden_point_ops, dev):
    import puzzlenet_amd.pointnet_util as pu
    G = golden_point_ops
    got = pu.square_distance(_t(G["sg_d_new_xyz"], dev), _t(G["sg_d_xyz"], dev))
    assert np.array_equal(got.cpu().numpy().view(np.uint32), G["sg_d_sqdist"].view(np.uint32))


def test_index_points_golden_and_grad(golden_point_ops, dev):
    import puzzlenet_amd.pointnet_util as pu
    G = golden_point_ops
    pts = _t(G["ip_points"], dev).requires_grad_(True)
    o2 = pu.index_points(pts, _t(G["ip_idx2"], dev))
    o3 = pu.index_points(pts, _t(G["ip_idx3"], dev))
    assert np.array_equal(o2.detach().cpu().numpy(), G["ip_out2"])
    assert np.array_equal(o3.detach().cpu().numpy(), G["ip_out3"])
    (o3 * _t(G["ip_w3"], dev)).sum().backward()
    np.testing.assert_allclose(pts.grad.cpu().numpy(), G["ip_grad3"], rtol=1e-5, atol=1e-6)


# ---- seeded inputs vs the oracle: sizes of every BASELINE config, edge cases ----

@pytest.mark.parametrize("B,N,S", [(4, 1024, 512), (3, 2048, 512), (2, 4096, 512), (1, 8192, 512),
                                   (5, 512, 256), (2, 100, 100), (1, 1, 1), (2, 65, 7), (1, 12000, 16),
                                   (1, 20000, 8)])
def test_fps_vs_oracle(dev, B, N, S):
    from puzzlenet_amd import ops
    rng = np.random.default_rng(N * 7 + S)
    xyz = rng.random((B, N, 3), dtype=np.float32)
    if N > 8:
        xyz[:, N // 2] = xyz[:, 1]                      # an exact duplicate: tie in the arg-max
    start = rng.integers(0, N, size=B)
    got = ops.farthest_point_sample(_t(xyz, dev), S, _t(start, dev)).cpu().numpy()
    want = orc.farthest_point_sample(xyz, S, start)
    assert np.array_equal(got, want)
    # the small-footprint launch of the data pipeline (no LDS image of the cloud): the same picks
    got_bg = ops.farthest_point_sample(_t(xyz, dev), S, _t(start, dev), background=True).cpu().numpy()
    assert np.array_equal(got_bg, want)
    # ... and with padding declared (rows >= counts[b] are copies of row 0, datapipe._compact): the picks of the real rows alone
    c = N // 2 + 3
    if S <= c:
        padded = xyz.copy()
        padded[:, c:] = padded[:, :1]
        st_c = start % c
        got_c = ops.farthest_point_sample(_t(padded, dev), S, _t(st_c, dev), background=True,
                                          counts=torch.full((B,), c, dtype=torch.int64, device=dev)).cpu().numpy()
        assert np.array_equal(got_c, orc.farthest_point_sample(np.ascontiguousarray(xyz[:, :c]), S, st_c))
        assert np.array_equal(got_c, ops.farthest_point_sample(_t(padded, dev), S, _t(st_c, dev)).cpu().numpy())
        # ... and with the caller's bound on the counts (the launch then holds only that many rows: fewer wavefronts)
        for bound in (c, c + 100):
            got_m = ops.farthest_point_sample(_t(padded, dev), S, _t(st_c, dev), background=True, max_count=bound,
                                              counts=torch.full((B,), c, dtype=torch.int64, device=dev)).cpu().numpy()
            assert np.array_equal(got_m, got_c)


@pytest.mark.parametrize("B,N,S,K", [(2, 1024, 512, 32), (2, 2048, 512, 32), (2, 512, 256, 32), (1, 4096, 512, 32),
                                     (1, 8192, 256, 32), (2, 100, 33, 7), (1, 64, 64, 64), (2, 33, 5, 33),
                                     (1, 300, 9, 1), (1, 5000, 40, 48), (1, 13000, 12, 32)])
def test_knn_vs_oracle(dev, B, N, S, K):
    from puzzlenet_amd import ops
    rng = np.random.default_rng(N + 13 * K)
    xyz = rng.random((B, N, 3), dtype=np.float32)
    if N >= 64:
        xyz[:, 5:25] = xyz[:, 40:60]                    # 20 duplicate points -> distance ties
    q = xyz[:, rng.permutation(N)[:S] if S <= N else rng.integers(0, N, S)].copy()
    got = ops.knn(_t(xyz, dev), _t(q, dev), K).cpu().numpy()
    assert np.array_equal(got, orc.knn(xyz, q, K))


def test_knn_adversarial_layouts(dev):
    """Inputs built to stress the candidate/merge path: all points identical,
    points sorted by decreasing distance, and near points packed on few lanes."""
    from puzzlenet_amd import ops
    N, K = 2048, 32
    rng = np.random.default_rng(5)
    same = np.tile(np.float32([0.3, 0.4, 0.5]), (1, N, 1))
    q = np.float32([[[0.3, 0.4, 0.5], [0.0, 0.0, 0.0]]])
    desc = np.zeros((1, N, 3), np.float32)
    desc[0, :, 0] = np.linspace(2.0, 0.01, N, dtype=np.float32)
    lanes = rng.random((1, N, 3), dtype=np.float32) + 5.0
    lanes[0, 3::64] = rng.random((N // 64, 3), dtype=np.float32) * 0.01     # lane 3 holds every near point
    lanes[0, 7::64] = rng.random((N // 64, 3), dtype=np.float32) * 0.01
    # 100 copies of one point next to the first query: 65..128 candidates pass the first bound (the reduce-after-
    # collect path of select32), and 40 + 40 copies of two points: ties straddling the 32nd slot
    dup = rng.random((1, N, 3), dtype=np.float32)
    dup[0, 300:400] = np.float32([0.31, 0.41, 0.51])
    dup2 = rng.random((1, N, 3), dtype=np.float32)
    dup2[0, 100:140] = np.float32([0.3, 0.4, 0.52])
    dup2[0, 1000:1040] = np.float32([0.3, 0.4, 0.48])
    for xyz in (same, desc, lanes, dup, dup2):
        got = ops.knn(_t(xyz, dev), _t(q, dev), K).cpu().numpy()
        assert np.array_equal(got, orc.knn(xyz, q, K))
        # the fused search + group launch on the same clouds (reference layout)
        feat = rng.standard_normal((1, N, 8)).astype(np.float32)
        new_points, gx, idx = ops.knn_group(_t(xyz, dev), _t(feat, dev), _t(q, dev), want_grouped_xyz=True)
        want_idx = orc.knn(xyz, q, K)
        assert np.array_equal(idx.cpu().numpy(), want_idx)
        assert np.array_equal(new_points.cpu().numpy().view(np.uint32), orc.group(xyz, feat, q, want_idx).view(np.uint32))


def test_group_backward_vs_oracle(dev):
    from puzzlenet_amd import ops
    rng = np.random.default_rng(11)
    B, N, S, K, D = 2, 256, 64, 32, 64
    xyz = rng.random((B, N, 3), dtype=np.float32)
    feat = rng.standard_normal((B, N, D)).astype(np.float32)
    new_xyz = xyz[:, :S].copy()
    idx = orc.knn(xyz, new_xyz, K)
    go = rng.standard_normal((B, S, K, 3 + D)).astype(np.float32)
    txyz, tfeat, tnew = (_t(a, dev).requires_grad_(True) for a in (xyz, feat, new_xyz))
    out = ops.group(txyz, tfeat, tnew, _t(idx, dev))
    assert np.array_equal(out.detach().cpu().numpy().view(np.uint32), orc.group(xyz, feat, new_xyz, idx).view(np.uint32))
    (out * _t(go, dev)).sum().backward()
    gxyz, gfeat, gnew = orc.group_grad(go, idx, N)
    np.testing.assert_allclose(tfeat.grad.cpu().numpy(), gfeat, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(txyz.grad.cpu().numpy(), gxyz, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tnew.grad.cpu().numpy(), gnew, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("K,D", [(32, 64), (32, 128), (16, 6), (7, 5), (5, 0)])
def test_group_shapes_vs_oracle(dev, K, D):
    from puzzlenet_amd import ops
    rng = np.random.default_rng(K * 100 + D)
    B, N, S = 3, 300, 50
    xyz = rng.random((B, N, 3), dtype=np.float32)
    feat = rng.standard_normal((B, N, D)).astype(np.float32) if D else None
    new_xyz = rng.random((B, S, 3), dtype=np.float32)
    idx = rng.integers(0, N, size=(B, S, K))
    out = ops.group(_t(xyz, dev), None if feat is None else _t(feat, dev), _t(new_xyz, dev), _t(idx, dev))
    assert np.array_equal(out.cpu().numpy().view(np.uint32), orc.group(xyz, feat, new_xyz, idx).view(np.uint32))


def test_full_size_properties(dev):
    """BASELINE configs[1] size (B=64, N=2048): properties that need no oracle run."""
    import puzzlenet_amd.pointnet_util as pu
    g = torch.Generator().manual_seed(3)
    xyz = torch.rand(64, 2048, 3, generator=g).to(dev)
    feat = torch.randn(64, 2048, 64, generator=g).to(dev)
    new_xyz, new_points, grouped_xyz, fps_idx = pu.sample_and_group(512, 0, 32, xyz, feat, returnfps=True, knn=True)
    # FPS: indices in range and distinct per cloud (random data has no duplicates)
    assert int(fps_idx.min()) >= 0 and int(fps_idx.max()) < 2048
    assert all(len(torch.unique(fps_idx[b])) == 512 for b in range(0, 64, 9))
    # kNN: slot 0 is the centroid itself, distances ascend (torch's own arithmetic: loose check)
    d = ((grouped_xyz - new_xyz[:, :, None]) ** 2).sum(-1)
    assert float(d[:, :, 0].abs().max()) == 0.0
    assert bool((d[:, :, 1:] >= d[:, :, :-1] - 1e-6).all())
    # exact check on two clouds with the kernel's own distances: the K-th distance is a
    # true threshold (exactly 32 points within it) and the slots ascend
    from puzzlenet_amd import ops
    full = pu.square_distance(new_xyz[:2], xyz[:2])
    knn_idx = ops.knn(xyz[:2], new_xyz[:2], 32)
    dk = torch.gather(full, 2, knn_idx)
    assert bool((dk[:, :, 1:] >= dk[:, :, :-1]).all())
    assert bool(((full <= dk[:, :, -1:]).sum(-1) == 32).all())
    assert torch.equal(knn_idx, torch.argsort(full, dim=-1, stable=True)[:, :, :32])
    # group: feature half equals a plain gather of the selected rows
    ref = torch.gather(feat[:2], 1, knn_idx.reshape(2, -1, 1).expand(-1, -1, 64)).reshape(2, 512, 32, 64)
    assert torch.equal(new_points[:2, :, :, 3:], ref)
    assert torch.equal(new_points[:, :, :, :3], grouped_xyz - new_xyz[:, :, None])


@pytest.mark.parametrize("B,N,S,D", [(3, 2048, 512, 64), (2, 512, 256, 128), (2, 200, 33, 8), (1, 4096, 64, 16),
                                     (2, 64, 10, 4), (1, 1000, 77, 12), (2, 130, 130, 20), (1, 3000, 19, 256),
                                     (1, 8192, 96, 64), (1, 6000, 50, 128)])
def test_knn_group_fused_vs_oracle(dev, B, N, S, D):
    """pzn_knn_group_f32 (search + group in one launch, REFERENCE layout [B,S,32,3+D]) == oracle kNN (bit-exact idx)
    + oracle group, incl. grouped_xyz; D / 4 a power of two or not, piece heights 32 / 16 / 8 / 4."""
    from puzzlenet_amd import ops
    rng = np.random.default_rng(N + D)
    xyz = rng.random((B, N, 3), dtype=np.float32)
    xyz[:, 5:15] = xyz[:, 40:50]                                    # duplicates: ties
    feat = rng.standard_normal((B, N, D)).astype(np.float32)
    q = xyz[:, rng.permutation(N)[:S]].copy()
    new_points, gx, idx = ops.knn_group(_t(xyz, dev), _t(feat, dev), _t(q, dev), want_grouped_xyz=True)
    want_idx = orc.knn(xyz, q, 32)
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    g = orc.group(xyz, feat, q, want_idx)                            # [B,S,32,3+D]
    assert np.array_equal(new_points.cpu().numpy().view(np.uint32), g.view(np.uint32))
    want_gx = np.take_along_axis(xyz, want_idx.reshape(B, -1, 1), axis=1).reshape(B, S, 32, 3)
    assert np.array_equal(gx.cpu().numpy(), want_gx)
    # the drop-in call takes this path and keeps its autograd (pointnet_util.py:123-132)
    import puzzlenet_amd.pointnet_util as pu
    tf = _t(feat, dev).requires_grad_(True)
    tx = _t(xyz, dev).requires_grad_(True)
    torch.manual_seed(3)
    new_xyz, npts = pu.sample_and_group(S, 0, 32, tx, tf, knn=True)
    torch.manual_seed(3)
    fps = pu.farthest_point_sample(tx.detach(), S)
    idx2 = ops.knn(tx.detach(), ops.index_points(tx.detach(), fps), 32)
    tf2, tx2 = _t(feat, dev).requires_grad_(True), _t(xyz, dev).requires_grad_(True)
    ref = ops.group(tx2, tf2, ops.index_points(tx2, fps), idx2)
    assert torch.equal(npts, ref)
    w = torch.randn(npts.shape, generator=torch.Generator().manual_seed(1)).to(dev)
    (npts * w).sum().backward()
    (ref * w).sum().backward()
    assert torch.allclose(tf.grad, tf2.grad, rtol=1e-5, atol=1e-5) and torch.allclose(tx.grad, tx2.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,L,C", [(3, 256, 1024), (2, 2048, 64), (1, 7, 4), (5, 33, 36)])
def test_max_over_points(B, L, C):
    """ops.max_over_points == torch.max(x, dim=1)[0] (model5_b.py:475, :741), gradient = one-hot scatter;
    exact (it is a selection); ties resolve to the lowest row."""
    from puzzlenet_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 1000 + L + C)
    x = torch.randn(B, L, C, generator=g)
    x[:, L // 2] = x[:, 0]                       # duplicate rows: ties between row 0 and row L//2
    xr = x.clone().requires_grad_(True)
    ref = torch.max(xr, dim=1)[0]
    go = torch.randn(B, C, generator=g)
    xd = x.to(dev).requires_grad_(True)
    out = ops.max_over_points(xd)
    assert torch.equal(out.cpu(), ref.detach())
    out.backward(go.to(dev))
    # reference gradient with the lowest arg-max row on ties
    idx = (x == ref.detach().unsqueeze(1)).float().argmax(dim=1)          # first row attaining the max
    want = torch.zeros_like(x).scatter_(1, idx.unsqueeze(1), go.unsqueeze(1))
    assert torch.equal(xd.grad.cpu(), want)


@pytest.mark.parametrize("B,N,C", [(8, 96, 64), (5, 33, 20), (64, 256, 64), (3, 7, 130)])
@pytest.mark.parametrize("training", [True, False])
def test_bn_points_relu_vs_torch(B, N, C, training):
    """relu(BatchNorm1d(num_points)(x)) as one HIP launch each way (csrc/bnpoints.hip) vs torch's BatchNorm + relu in
    fp64: output, input / weight / bias gradients, running statistics, batch counter; train and eval mode."""
    import torch.nn as nn
    import torch.nn.functional as F
    from puzzlenet_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 100 + N)
    x = torch.randn(B, N, C, generator=g) * 2 + 0.5
    go = torch.randn(B, N, C, generator=g)
    ref = nn.BatchNorm1d(N).double()
    with torch.no_grad():
        ref.weight.copy_(1 + 0.3 * torch.randn(N, generator=g))
        ref.bias.copy_(0.2 * torch.randn(N, generator=g))
        ref.running_mean.copy_(0.1 * torch.randn(N, generator=g))
        ref.running_var.copy_(1 + 0.2 * torch.rand(N, generator=g))
    mine = nn.BatchNorm1d(N)
    mine.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in ref.state_dict().items()})
    mine = mine.to(dev)
    ref.train(training), mine.train(training)
    xr = x.double().requires_grad_(True)
    xd = x.to(dev).requires_grad_(True)
    yr = F.relu(ref(xr))
    y = ops.bn_points_relu(xd, mine)
    # a pre-activation within rounding of zero may gate differently: take the gate from the device output
    gate = (y.detach().cpu() > 0)
    assert int((gate != (yr.detach() > 0)).sum()) <= 2
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 2e-5
    (torch.where(gate, ref(xr), torch.zeros_like(yr)) * go.double()).sum().backward()
    (y * go.to(dev)).sum().backward()
    def rel(a, b):
        return float((a.detach().cpu().double() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert rel(xd.grad, xr.grad) < 1e-4
    assert rel(mine.weight.grad, ref.weight.grad) < 1e-4
    assert rel(mine.bias.grad, ref.bias.grad) < 1e-4
    # ref(xr) ran twice (two momentum updates): the expected running statistics come from a fresh module updated once
    init = nn.BatchNorm1d(N).double()
    g2 = torch.Generator().manual_seed(B * 100 + N)
    _ = torch.randn(B, N, C, generator=g2), torch.randn(B, N, C, generator=g2)
    with torch.no_grad():
        init.weight.copy_(1 + 0.3 * torch.randn(N, generator=g2))
        init.bias.copy_(0.2 * torch.randn(N, generator=g2))
        init.running_mean.copy_(0.1 * torch.randn(N, generator=g2))
        init.running_var.copy_(1 + 0.2 * torch.rand(N, generator=g2))
    init.train(training)
    init(x.double())
    assert rel(mine.running_mean, init.running_mean) < 1e-5
    assert rel(mine.running_var, init.running_var) < 1e-5
    assert int(mine.num_batches_tracked) == int(init.num_batches_tracked)


@pytest.mark.parametrize("B,N,S", [(3, 97, 33), (2, 2048, 512), (5, 64, 64), (1, 8192, 16)])
def test_knn_inverse_lists(B, N, S):
    """pzn_knn_inverse_lists: the B*S*32 (row, point) pairs of a neighbour index sorted by point, per cloud — every row
    exactly once, under the point it gathered, `off` the exclusive prefix of the reference counts."""
    from puzzlenet_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N + S)
    idx = torch.randint(0, N, (B, S, 32), generator=g, dtype=torch.int64)
    idx[0, 0, :] = 0                      # a hub point: many rows under one point
    d = idx.to(dev)
    off = torch.empty(B * (N + 1), dtype=torch.int32, device=dev)
    rows = torch.empty(B * S * 32, dtype=torch.int32, device=dev)
    pts = torch.empty_like(rows)
    ops._call("pzn_knn_inverse_lists", ops._p(d), B, N, S, 32, ops._p(off), ops._p(rows), ops._p(pts), ops._stream())
    torch.cuda.synchronize()
    off, rows, pts = off.cpu().view(B, N + 1), rows.cpu().view(B, S * 32), pts.cpu().view(B, S * 32)
    flat = idx.view(B, -1)
    for b in range(B):
        counts = torch.bincount(flat[b], minlength=N)
        assert torch.equal(off[b, 1:].long(), torch.cumsum(counts, 0))
        assert int(off[b, 0]) == 0
        assert torch.equal(torch.sort(rows[b].long())[0], torch.arange(S * 32))      # a permutation of the rows
        assert torch.equal(flat[b][rows[b].long()], pts[b].long())                    # each row under its point
        assert bool((pts[b, 1:] >= pts[b, :-1]).all())                                # grouped by point, ascending
