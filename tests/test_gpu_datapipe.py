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
