#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE on CPU.

Runs only in the build container (needs /root/reference); the fixtures it
writes are plain data (inputs + the reference's outputs) and are committed.
Nothing under tests/, bench.py or smoke() reads /root/reference at run time.

    python tests/golden/make_golden.py [point_ops] [model] [loss]

Inputs come from numpy's default_rng(seed) (PCG64: stable across numpy
versions) so they do not depend on the torch version.

Modules the image lacks (pytorch_lightning, open3d, torchvision, plyfile, and
the two modules the reference repo itself does not ship: pct,
pointtransformer_partseg) are registered as empty placeholders in sys.modules
*in this harness only* so that `import model5_b` succeeds; none of them is on
the predict5 path.  emd_cuda (CUDA-only) is likewise a placeholder: EMD is not
pinned by this script (see oracle/pzn_oracle.c header).
"""
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _ref_pointnet_util():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import pointnet_util as pu  # the reference's own file
    assert pu.__file__.startswith(REF), pu.__file__
    return pu


def _cloud(rng, B, N, dup=0):
    """U[0,1)^3 cloud; `dup` points are exact duplicates of earlier points
    (forces distance ties: exercises tie-breaking in FPS arg-max and kNN)."""
    xyz = rng.random((B, N, 3), dtype=np.float32)
    for b in range(B):
        if dup:
            src = rng.integers(0, N, size=dup)
            dst = rng.integers(0, N, size=dup)
            xyz[b, dst] = xyz[b, src]
    return xyz


def make_point_ops():
    pu = _ref_pointnet_util()
    rng = np.random.default_rng(20241008)
    G = {}

    # G1 FPS + G2 kNN + G4 sample_and_group, two cloud sizes, with duplicates.
    for tag, (B, N, S, K, D, dup) in {
        "a": (2, 1024, 512, 32, 8, 0),
        "b": (2, 2048, 512, 32, 4, 64),
        "c": (2, 512, 256, 32, 16, 16),     # sg2 shape
        "d": (3, 200, 33, 7, 5, 20),        # ragged: N not a multiple of 64, odd S/K
        "e": (1, 64, 64, 64, 0, 0),         # K == N, points=None
        "f": (2, 2048, 512, 32, 4, 0),      # headline size, tie-free
    }.items():
        xyz = _cloud(rng, B, N, dup)
        feat = rng.standard_normal((B, N, D)).astype(np.float32) if D else None
        txyz = torch.from_numpy(xyz)
        tfeat = torch.from_numpy(feat) if D else None
        torch.manual_seed(100 + ord(tag))
        new_xyz, new_points, grouped_xyz, fps_idx = pu.sample_and_group(
            S, 0, K, txyz, tfeat, returnfps=True, knn=True)
        d = pu.square_distance(new_xyz, txyz)
        knn_idx = d.argsort()[:, :, :K]
        G[f"sg_{tag}_xyz"] = xyz
        if D:
            G[f"sg_{tag}_feat"] = feat
        G[f"sg_{tag}_params"] = np.array([B, N, S, K, D], np.int64)
        G[f"sg_{tag}_fps_idx"] = fps_idx.numpy()
        G[f"sg_{tag}_knn_idx"] = knn_idx.numpy()
        G[f"sg_{tag}_new_xyz"] = new_xyz.numpy()
        G[f"sg_{tag}_new_points"] = new_points.numpy()
        G[f"sg_{tag}_grouped_xyz"] = grouped_xyz.numpy()
        if tag in ("d",):
            G[f"sg_{tag}_sqdist"] = d.numpy()

    # G3 ball query: radii incl. rows with < nsample hits and rows with 0 hits.
    B, N, S = 2, 300, 40
    xyz = _cloud(rng, B, N, 10)
    new_xyz = xyz[:, :S].copy()
    new_xyz[:, -4:] += 5.0  # four far-away queries per cloud: no hit at all
    for r in (0.1, 0.2, 0.37):
        for ns in (8, 32):
            idx = pu.query_ball_point(r, ns, torch.from_numpy(xyz), torch.from_numpy(new_xyz))
            G[f"ball_r{r}_n{ns}"] = idx.numpy()
    G["ball_xyz"] = xyz
    G["ball_new_xyz"] = new_xyz
    # boundary semantics of `sqrdists > radius ** 2` (fp32 vs fp64 compare):
    # a point whose squared distance is EXACTLY float32(r*r) where float32(r*r) > r*r.
    r = None
    for cand in np.linspace(0.11, 0.9, 400):
        if float(np.float32(cand * cand)) > cand * cand and math.sqrt(float(np.float32(cand * cand))) ** 2 > 0:
            s = np.float32(np.sqrt(np.float32(cand * cand)))
            if np.float32(s * s) == np.float32(cand * cand):
                r = float(cand)
                break
    assert r is not None
    s = np.float32(np.sqrt(np.float32(r * r)))
    exyz = np.zeros((1, 4, 3), np.float32)
    exyz[0, 1, 0] = s            # d == float32(r^2) exactly
    exyz[0, 2, 0] = s * 2
    equery = np.zeros((1, 1, 3), np.float32)
    eidx = pu.query_ball_point(r, 3, torch.from_numpy(exyz), torch.from_numpy(equery))
    G["ball_edge_radius"] = np.array([r], np.float64)
    G["ball_edge_xyz"] = exyz
    G["ball_edge_query"] = equery
    G["ball_edge_idx"] = eidx.numpy()
    # sample_and_group with knn=False (ball query inside)
    torch.manual_seed(7)
    bx = _cloud(rng, 2, 256, 0)
    bf = rng.standard_normal((2, 256, 6)).astype(np.float32)
    o = pu.sample_and_group(32, 0.2, 16, torch.from_numpy(bx), torch.from_numpy(bf), returnfps=True, knn=False)
    G["sgball_xyz"], G["sgball_feat"] = bx, bf
    G["sgball_new_xyz"], G["sgball_new_points"], G["sgball_grouped_xyz"], G["sgball_fps_idx"] = (t.numpy() for t in o)

    # G5 index_points rank-2 / rank-3 and its autograd gradient.
    B, N, C = 2, 50, 5
    pts = rng.standard_normal((B, N, C)).astype(np.float32)
    i2 = rng.integers(0, N, size=(B, 9))
    i3 = rng.integers(0, N, size=(B, 7, 4))
    tp = torch.from_numpy(pts).requires_grad_(True)
    o2 = pu.index_points(tp, torch.from_numpy(i2))
    o3 = pu.index_points(tp, torch.from_numpy(i3))
    w3 = rng.standard_normal(o3.shape).astype(np.float32)
    (o3 * torch.from_numpy(w3)).sum().backward()
    G["ip_points"], G["ip_idx2"], G["ip_idx3"] = pts, i2, i3
    G["ip_out2"], G["ip_out3"] = o2.detach().numpy(), o3.detach().numpy()
    G["ip_w3"], G["ip_grad3"] = w3, tp.grad.numpy()

    np.savez_compressed(os.path.join(OUT, "point_ops.npz"), **G)
    print("point_ops.npz:", len(G), "arrays")


if __name__ == "__main__":
    what = sys.argv[1:] or ["point_ops", "model", "loss"]
    if "point_ops" in what:
        make_point_ops()
    if "model" in what or "loss" in what:
        from make_golden_model import make_model, make_loss  # noqa: E402
        if "model" in what:
            make_model()
        if "loss" in what:
            make_loss()
