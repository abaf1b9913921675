"""GPU training-pair construction (puzzlenet_amd/datapipe.py) against the reference's own functions
run on CPU (tests/golden/data.npz): same draws in, same 8-tuple out."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(G, dev):
    C = int(G["cases"])
    t = lambda k, dt=torch.float32: torch.stack([torch.from_numpy(np.asarray(G[f"c{c}_{k}"])) for c in range(C)]).to(dt).to(dev)
    return dict(raw=t("raw"), normal=t("normal", torch.float64), z=t("z", torch.float64).reshape(-1),
                s_up=t("s_up", torch.int64), s_down=t("s_down", torch.int64), twist=t("twist"))


def test_make_pairs_matches_reference_pipeline(golden_data):
    from puzzlenet_amd import datapipe
    G = golden_data
    dev = torch.device("cuda:0")
    b = _batch(G, dev)
    (down, moved, igt, up, downb, upb, down_mask, up_mask), ok = datapipe.make_pairs(
        b["raw"], b["normal"], b["z"], b["s_up"], b["s_down"], b["twist"], n=int(G["N"]))
    assert bool(ok.all())
    for c in range(int(G["cases"])):
        # FPS of both pieces: the reference's points in the reference's order, bit for bit
        assert np.array_equal(up[c].cpu().numpy(), G[f"c{c}_up"])
        assert np.array_equal(down[c].cpu().numpy(), G[f"c{c}_down"])
        # motion
        np.testing.assert_allclose(igt[c].cpu().numpy(), G[f"c{c}_igt"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(moved[c].cpu().numpy(), G[f"c{c}_mup"], rtol=1e-5, atol=1e-6)
        # boundary: 128 points per piece; a near-tie at the 128th distance may resolve differently (the reference's
        # chamfer is a bmm expansion on CPU), so at most two labels per piece may differ
        for mine, ref, pts, piece in ((down_mask, "fpc_idx", downb, down), (up_mask, "rpc_idx", upb, up)):
            m = mine[c].cpu().numpy()
            assert int(m.sum()) == 128
            assert int((m != G[f"c{c}_{ref}"]).sum()) <= 4      # one swapped pair = 2 differing labels
            sel = piece[c][mine[c] > 0].cpu().numpy()
            got = pts[c].cpu().numpy()
            assert {tuple(r) for r in got} == {tuple(r) for r in sel}     # the boundary points ARE the marked points


def test_rows_that_need_a_new_cut_are_flagged(golden_data):
    from puzzlenet_amd import datapipe
    G = golden_data
    dev = torch.device("cuda:0")
    b = _batch(G, dev)
    z = b["z"].clone()
    z[1] = 10.0                      # everything lands on one side of the plane
    _, ok = datapipe.make_pairs(b["raw"], b["normal"], z, b["s_up"], torch.zeros_like(b["s_down"]), b["twist"], n=int(G["N"]))
    assert ok.tolist() == [True, False, True, True]


def test_pairs_feed_a_training_step(golden_data):
    """The 8-tuple of datapipe.make_pairs is the batch contract of model5_b.training_step (dataset.py:98-105): one
    eager training step on it runs through every kernel and leaves finite loss and gradients."""
    from oracle import model_ref as mr
    from puzzlenet_amd import datapipe, engine
    from puzzlenet_amd import model5_b as mb
    G = golden_data
    dev = torch.device("cuda:0")
    b = _batch(G, dev)
    batch, ok = datapipe.make_pairs(b["raw"], b["normal"], b["z"], b["s_up"], b["s_down"], b["twist"], n=int(G["N"]))
    assert bool(ok.all())
    cfg = mr.Cfg(loss_mode=1, num_points=int(G["N"]))
    torch.manual_seed(0)
    model = mb.TouchedRegraster(cfg).to(dev)
    runner = engine.TrainStep(model, list(batch), cfg.lr, world=1)
    l0 = float(runner.step())
    l1 = float(runner.step())
    assert np.isfinite(l0) and np.isfinite(l1)
    assert bool(torch.isfinite(runner.grads.flat).all()) and float(runner.grads.flat.abs().max()) > 0
    runner.close()


def _check_item(out, G, k, c):
    down, moved, igt, up, downb, upb, down_mask, up_mask = out
    assert np.array_equal(up[c].cpu().numpy(), G[k + "up"]) and np.array_equal(down[c].cpu().numpy(), G[k + "down"])
    np.testing.assert_allclose(igt[c].cpu().numpy(), G[k + "igt"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(moved[c].cpu().numpy(), G[k + "mup"], rtol=1e-5, atol=1e-6)
    for mine, ref, pts, piece in ((down_mask, "fpc_idx", downb, down), (up_mask, "rpc_idx", upb, up)):
        m = mine[c].cpu().numpy()
        assert int(m.sum()) == 128
        assert int((m != G[k + ref]).sum()) <= 4          # (a near-tie at the 128th distance: one swapped pair)
        assert {tuple(r) for r in pts[c].cpu().numpy()} == {tuple(r) for r in piece[c][mine[c] > 0].cpu().numpy()}


def test_double_cut_pairs_match_reference(golden_data2):
    """datapipe.make_pairs_regions on the recipes of tests/golden/data2.npz (the reference's double-cut
    CADDataset.__getitem__ + MovedCADDataset2 run on CPU, one case per kind of pair): all cases as ONE batch."""
    from puzzlenet_amd import datapipe
    from tests.conftest import golden_cloud
    G = golden_data2
    dev = torch.device("cuda:0")
    seeds = G["seeds"].tolist()
    st = lambda name, dt: torch.stack([torch.from_numpy(np.asarray(G[f"s{s}_{name}"])) for s in seeds]).to(dt).to(dev)
    raw = torch.stack([torch.from_numpy(golden_cloud(s, G["M"])) for s in seeds]).to(dev)
    out, ok = datapipe.make_pairs_regions(raw, st("normal1", torch.float64), st("z1", torch.float64).reshape(-1),
                                          st("normal2", torch.float64), st("z2", torch.float64).reshape(-1),
                                          st("u_tab", torch.int64), st("d_tab", torch.int64), st("s_u", torch.int64),
                                          st("s_d", torch.int64), st("twist", torch.float32), n=int(G["N"]))
    assert bool(ok.all())
    for c, s in enumerate(seeds):
        _check_item(out, G, f"s{s}_", c)
    # the chamfer distance of the boundaries that decides the reference's half-vs-other branch (dataset.py:1250-1253)
    from puzzlenet_amd import ops
    c1, c2 = ops.chamfer(out[4], out[5])
    cd = (c1.mean(1) + c2.mean(1)).cpu().numpy()
    for c, s in enumerate(seeds):
        if str(G[f"s{s}_kind"]).startswith("half_vs_other"):
            np.testing.assert_allclose(cd[c], float(G[f"s{s}_cd"]), rtol=2e-2)
            assert cd[c] <= 0.015


def test_building_pairs_match_reference(golden_data2):
    """datapipe.building_pairs == MovedCADDataset2(BuildingDataset).__getitem__ (dataset.py:1370-1429, :92-105)."""
    from puzzlenet_amd import datapipe
    G = golden_data2
    dev = torch.device("cuda:0")
    twist = torch.stack([torch.from_numpy(G[f"b{i}_twist"]) for i in range(3)]).to(dev)
    out = datapipe.building_pairs(torch.from_numpy(G["b_fpcs"]).to(dev), torch.from_numpy(G["b_rpcs"]).to(dev), twist)
    for i in range(3):
        _check_item(out, G, f"b{i}_", i)


@pytest.mark.parametrize("kind", ["sphere", "cylinder", "cone"])
def test_make_pairs_solid_cuts(kind):
    """datapipe.make_pairs_solid (dataset.py:716-763 + the pair construction of CADDataset): the mesh's inside becomes
    `up`.  The mask on the GPU is the oracle's (oracle/solids.py: open3d 0.15.2's resolution-50 meshes restated, brute-force
    face-plane membership) point for point; then the contract of the pair construction: every sampled point is a point
    of the raw cloud on the right side, the FPS start point is first, 128 boundary points per piece, and the motion is
    the SE(3) exponential of the twist."""
    from puzzlenet_amd import datapipe, se3
    dev = torch.device("cuda:0")
    rng = np.random.default_rng({"sphere": 1, "cylinder": 2, "cone": 3}[kind])
    B, M, n = 3, 12000, 1024
    raw_np = (rng.random((B, M, 3)) * (1.6 if kind == "cone" else 1.0) - (0.8 if kind == "cone" else 0.0)).astype(np.float32)
    raw = torch.from_numpy(raw_np).to(dev)
    rot = torch.from_numpy(rng.random((B, 3))).to(dev)
    shift = torch.from_numpy(rng.random((B, 3)) / 3).to(dev)
    mask = datapipe.solid_cut_mask(raw, kind, rot, shift)
    from oracle import solids
    for b in range(B):
        want = solids.solid_cut_mask(raw_np[b].astype(np.float64), kind, rot[b].cpu().numpy(), shift[b].cpu().numpy())
        assert np.array_equal(mask[b].cpu().numpy(), want), (kind, b)
    n_up = mask.sum(1)
    assert bool(((n_up >= n) & (M - n_up >= n)).all()), n_up.tolist()
    s_up = torch.tensor([int(rng.integers(0, int(c))) for c in n_up.tolist()], device=dev)
    s_down = torch.tensor([int(rng.integers(0, M - int(c))) for c in n_up.tolist()], device=dev)
    tw = torch.from_numpy(rng.standard_normal((B, 6)).astype(np.float32))
    tw = (tw / tw.norm(dim=1, keepdim=True) * 0.8).to(dev)
    (down, moved, igt, up, downb, upb, down_mask, up_mask), ok = datapipe.make_pairs_solid(
        raw, kind, rot, shift, s_up, s_down, tw, n=n)
    assert bool(ok.all()) and up.shape == (B, n, 3) and down.shape == (B, n, 3)
    for b in range(B):
        rows = {tuple(r) for r in raw_np[b]}
        inside = {tuple(r) for r in raw_np[b][mask[b].cpu().numpy()]}
        u, d = up[b].cpu().numpy(), down[b].cpu().numpy()
        assert all(tuple(r) in inside for r in u)
        assert all(tuple(r) in rows and tuple(r) not in inside for r in d)
        assert tuple(u[0]) == tuple(raw_np[b][mask[b].cpu().numpy()][int(s_up[b])])      # FPS starts at the drawn point
        assert len({tuple(r) for r in u}) == n and len({tuple(r) for r in d}) == n
        assert int(up_mask[b].sum()) == 128 and int(down_mask[b].sum()) == 128
    g = se3.exp(tw)
    np.testing.assert_allclose(igt.cpu().numpy(), g.cpu().numpy(), rtol=1e-6, atol=1e-6)
    want = torch.einsum("bij,bnj->bni", g[:, :3, :3], up) + g[:, :3, 3].unsqueeze(1)
    np.testing.assert_allclose(moved.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_pair_feeder_builds_fresh_batches_beside_the_training_step():
    """datapipe.PairFeeder (the DataLoader's role, train.py:101-104): every next_batch() is a new valid 8-tuple cut from the
    resident raw clouds on a background stream (each piece on its side of the chosen plane, rigid motion, 128-point boundary
    masks), reproducible from the seed, and TrainStep.step(next_batch=...) trains on them without a host synchronisation."""
    from oracle import model_ref as mr
    from puzzlenet_amd import datapipe, engine
    from puzzlenet_amd import model5_b as mb
    dev = torch.device("cuda:0")
    B, M, N = 6, 5000, 1024
    rng = np.random.RandomState(3)
    u = rng.randn(B, M, 3).astype(np.float32)
    u /= np.linalg.norm(u, axis=2, keepdims=True)
    raw = u * (0.25 + 0.2 * rng.rand(B, 1, 3).astype(np.float32))

    def take(seed, count):
        f = datapipe.PairFeeder(raw, dev, n=N, seed=seed, candidates=32)
        out = [f.next_batch() for _ in range(count)]
        f.close()
        return out

    a, b = take(11, 3), take(11, 3)
    for x, y in zip(a, b):
        assert all(torch.equal(s, t) for s, t in zip(x, y))                      # same seed, same batches
    assert not torch.equal(a[0][0], a[1][0])                                       # a fresh cut every time
    for batch in a:
        assert bool(batch.ok.all())
        down, moved, igt, up, downb, upb, down_mask, up_mask = batch
        normal, z = batch.plane
        side = lambda p: (p.double() * normal.unsqueeze(1)).sum(-1) + z.reshape(-1, 1)
        assert bool((side(up) >= 0).all()) and bool((side(down) < 0).all())
        assert down.shape == up.shape == (B, N, 3) and float((igt[:, 3] - torch.tensor([0., 0., 0., 1.], device=dev)).abs().max()) == 0
        R = igt[:, :3, :3]
        assert float((R.transpose(1, 2) @ R - torch.eye(3, device=dev)).abs().max()) < 1e-5
        want = (R @ up.transpose(1, 2) + igt[:, :3, 3:]).transpose(1, 2)
        assert float((moved - want).abs().max()) < 1e-5
        assert bool((down_mask.sum(1) == 128).all()) and bool((up_mask.sum(1) == 128).all())
        # every sampled point is one of the raw cloud's
        rs = {tuple(r) for r in raw[0].tolist()}
        assert all(tuple(r) in rs for r in up[0].cpu().tolist()[:50] + down[0].cpu().tolist()[:50])
    feeder = datapipe.PairFeeder(raw, dev, n=N, seed=5)
    torch.manual_seed(0)
    model = mb.TouchedRegraster(mr.Cfg(loss_mode=1, num_points=N)).to(dev)
    runner = engine.TrainStep(model, feeder.next_batch(), 1e-3, world=1)
    losses = []
    for _ in range(4):
        losses.append(runner.step(next_batch=feeder.next_batch()))
    torch.cuda.synchronize()
    assert all(np.isfinite(float(l)) for l in losses) and len({round(float(l), 3) for l in losses}) > 1
    assert bool(torch.isfinite(runner.grads.flat).all())
    runner.close()
    feeder.close()


@pytest.mark.parametrize("B,M,K,n_min,cap", [(5, 5000, 8, 1024, 5000), (3, 777, 4, 100, 777), (2, 3000, 3, 1400, 2000), (1, 64, 2, 1, 64)])
def test_cut_compact_equals_the_tensor_form(B, M, K, n_min, cap):
    """ops.cut_compact (one launch: first valid of K candidate planes, stable partition, padding, start indices) against the
    tensor statement of the same thing: plane_cut_mask (float64, numpy's summation order) per candidate, the first valid one,
    datapipe._compact of both sides - pieces, counts, plane and ok bit for bit."""
    from puzzlenet_amd import datapipe, ops
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(B * 131 + M)
    raw = torch.from_numpy((rng.rand(B, M, 3).astype(np.float32) - 0.3)).to(dev)
    normals = torch.from_numpy(rng.rand(B, K, 3)).to(dev)
    zs = torch.from_numpy(rng.rand(B, K) / 3).to(dev)
    u = torch.from_numpy(rng.rand(B, 2)).to(dev)
    pieces, counts, start, plane, ok = ops.cut_compact(raw, normals, zs, u, n_min, cap)
    torch.cuda.synchronize()
    for b in range(B):
        pick, valid, best = None, False, (-1, 0)
        for k in range(K):
            m = datapipe.plane_cut_mask(raw[b:b + 1], normals[b:b + 1, k], zs[b:b + 1, k])
            up = int(m.sum())
            if min(up, M - up) > best[0]:
                best = (min(up, M - up), k)
            if up >= n_min and M - up >= n_min:
                pick, valid = k, True
                break
        pick = best[1] if pick is None else pick
        assert plane[b].tolist() == normals[b, pick].tolist() + [float(zs[b, pick])]
        m = datapipe.plane_cut_mask(raw[b:b + 1], normals[b:b + 1, pick], zs[b:b + 1, pick])
        n_up = int(m.sum())
        assert counts[b].item() == n_up and counts[B + b].item() == M - n_up
        assert bool(ok[b]) == (valid and n_up <= cap and M - n_up <= cap)
        for piece, mask, cnt in ((pieces[b], m, n_up), (pieces[B + b], ~m, M - n_up)):
            if 0 < cnt <= cap:
                want, _ = datapipe._compact(raw[b:b + 1], mask, cap)
                assert torch.equal(piece, want[0])
        for half, cnt in ((0, n_up), (1, M - n_up)):
            s = start[half * B + b].item()
            assert s == max(0, min(cnt - 1, int(np.floor(float(u[b, half]) * cnt))))
