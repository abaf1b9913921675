"""Data-pipeline fixtures against the CPU oracle: the reference's numpy FPS (dataset.py:1147-1163:
float64 `distance` array holding float32 values, first maximum) must be what the fp32 oracle FPS computes on the cut
pieces, draw for draw — that is what lets the GPU pipeline run `pzn_fps_f32` on them."""
import numpy as np

from oracle import point_ops as orc


def _pieces(G, c):
    raw = G[f"c{c}_raw"]
    dis = raw.astype(np.float64) @ G[f"c{c}_normal"].reshape(3, 1) + G[f"c{c}_z"]
    up = raw[(dis >= 0)[:, 0]]
    down = raw[(dis < 0)[:, 0]]
    return up, down


def test_cut_sizes_match_the_reference(golden_data):
    G = golden_data
    for c in range(int(G["cases"])):
        up, down = _pieces(G, c)
        assert up.shape[0] == int(G[f"c{c}_n_up"]) and down.shape[0] == int(G[f"c{c}_n_down"])


def test_oracle_fps_reproduces_dataset_fps(golden_data):
    G = golden_data
    n = int(G["N"])
    for c in range(int(G["cases"])):
        up, down = _pieces(G, c)
        for piece, start, want in ((up, G[f"c{c}_s_up"], G[f"c{c}_up"]), (down, G[f"c{c}_s_down"], G[f"c{c}_down"])):
            idx = orc.farthest_point_sample(piece[None], n, np.array([int(start)], np.int64))[0]
            assert np.array_equal(piece[idx], want)          # same points in the same selection order, bit for bit
        # padding a piece with copies of its first point (what the GPU pipeline does) changes nothing
        pad = np.concatenate([down, np.repeat(down[:1], 777, 0)], 0)
        idx = orc.farthest_point_sample(pad[None], n, np.array([int(G[f"c{c}_s_down"])], np.int64))[0]
        assert np.array_equal(pad[idx], G[f"c{c}_down"])


def test_boundary_fixture_is_consistent(golden_data):
    G = golden_data
    for c in range(int(G["cases"])):
        assert int(G[f"c{c}_fpc_idx"].sum()) == 128 and int(G[f"c{c}_rpc_idx"].sum()) == 128
        # the masks mark exactly the 128 smallest chamfer distances of each piece
        for cd, mask in ((G[f"c{c}_cd_over_down"], G[f"c{c}_fpc_idx"]), (G[f"c{c}_cd_over_up"], G[f"c{c}_rpc_idx"])):
            assert cd[mask > 0].max() <= np.sort(cd)[128] + 1e-12


def test_draws_follow_the_reference_order(golden_data):
    """datapipe.draws_like_reference consumes numpy's / torch's global generators exactly as the reference's
    plane_split -> fps(up) -> fps(down) -> RandomTransformSE3 sequence does (seeds of make_golden_data.py)."""
    import torch
    from puzzlenet_amd import datapipe
    G = golden_data
    for c in range(int(G["cases"])):
        np.random.seed(1000 + c)
        torch.manual_seed(7000 + c)
        d = datapipe.draws_like_reference(G[f"c{c}_raw"], n=int(G["N"]), mag=0.8)
        assert np.array_equal(d["normal"], G[f"c{c}_normal"]) and np.array_equal(d["z"], G[f"c{c}_z"])
        assert d["s_up"] == int(G[f"c{c}_s_up"]) and d["s_down"] == int(G[f"c{c}_s_down"])
        assert np.array_equal(d["twist"], G[f"c{c}_twist"])
