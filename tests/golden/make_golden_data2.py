#!/usr/bin/env python3
"""Data-pipeline golden fixtures, second set (SURVEY §8 row f2 remainder): runs the REFERENCE's own classes on CPU.

    python tests/golden/make_golden_data2.py      ->  tests/golden/data2.npz

  * `MovedCADDataset2(CADDataset(split_twice=True), RandomTransformSE3(0.8)).__getitem__` (dataset.py:92-105,
    :1203-1355): the double-cut item — which of the branches a sample takes depends on its random draws, so a range of
    seeds is run and one case per kind of pair is kept (single cut, half vs rest, half vs the other piece, the two
    halves, on the upper and on the lower piece).
  * `MovedCADDataset2(BuildingDataset)` (:1370-1429): two given clouds.
Clouds are regenerated from their seed (numpy Generator, stable bit streams), so only seeds, recipes and outputs are
stored.  For every kept case the product's host-side planner (puzzlenet_amd.datapipe.plan_double_cut_like_reference: pure
numpy / torch host code) is run on the same seeds with a CPU acceptance callback built from the REFERENCE's functions,
and the pieces its recipe describes are rebuilt here with the reference's own fps and checked against the reference's
item — the fixture therefore carries the recipe (planes, region tables, start indices, twist) next to the outputs.
Same harness rules as make_golden_model.py (sys.modules placeholders only, nothing of the reference is copied).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden_model as gm  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
N = 1024
M = 9000
SEEDS = range(40)


def cloud(seed):
    rng = np.random.default_rng(50_000 + seed)
    return (rng.random((M, 3), dtype=np.float32) - np.float32(0.5)).astype(np.float32)


def region_points(raw, rec, tab):
    s1 = (np.dot(raw, rec["normal1"].reshape(3, 1)) + rec["z1"] >= 0).reshape(-1)
    s2 = (np.dot(raw, rec["normal2"].reshape(3, 1)) + rec["z2"] >= 0).reshape(-1)
    code = 2 * s1.astype(np.int64) + s2.astype(np.int64)
    parts = [raw[((int(t) >> code) & 1) == 1] for t in tab if int(t)]
    return np.vstack(parts)


def fps_from(ds, ref, pts, start):
    """the reference's fps (dataset.py:1147-1163) with its randint draw replaced by `start`"""
    orig = np.random.randint
    np.random.randint = lambda *a, **k: start
    try:
        return ds.CADDataset.fps(ref, pts, N)
    finally:
        np.random.randint = orig


def main():
    gm.import_reference_model()
    import dataset as ds
    import se_math.transforms as tr
    from puzzlenet_amd import datapipe

    G = {"N": np.int64(N), "M": np.int64(M)}
    kept = {}
    for seed in SEEDS:
        pc = cloud(seed)
        inner = object.__new__(ds.CADDataset)
        inner.all, inner.split, inner.split_twice = [pc], ds.plane_split, True
        moved = ds.MovedCADDataset2(inner, tr.RandomTransformSE3(0.8))
        np.random.seed(3000 + seed)
        torch.manual_seed(9000 + seed)
        down, mup, igt, up, downb, upb, fpc_idx, rpc_idx = moved[0]
        # the product's planner on the same seeds; acceptance = the reference's own boundary + chamfer on the candidate
        ref = object.__new__(ds.CADDataset)

        def accept(rec):
            U = fps_from(ds, ref, region_points(pc, rec, rec["u_tab"]), rec["s_u"])
            D = fps_from(ds, ref, region_points(pc, rec, rec["d_tab"]), rec["s_d"])
            Ut, Dt = torch.from_numpy(U).to(torch.float32), torch.from_numpy(D).to(torch.float32)
            fb, rb, _, _ = ds.CADDataset.get_boundary(ref, Dt, Ut)
            c1, c2 = ds.CADDataset.chamfer_loss(ref, fb.unsqueeze(0), rb.unsqueeze(0))
            accept.cd = float(torch.mean(c1) + torch.mean(c2))
            return accept.cd
        accept.cd = float("nan")
        np.random.seed(3000 + seed)
        torch.manual_seed(9000 + seed)
        rec = datapipe.plan_double_cut_like_reference(pc, accept, n=N, mag=0.8)
        U = fps_from(ds, ref, region_points(pc, rec, rec["u_tab"]), rec["s_u"])
        D = fps_from(ds, ref, region_points(pc, rec, rec["d_tab"]), rec["s_d"])
        assert np.array_equal(U, up.numpy()) and np.array_equal(D, down.numpy()), (seed, rec["kind"])
        # the generators stand where the reference left them (the next sample would see the same draws)
        st_np, st_t = np.random.get_state()[1].copy(), torch.get_rng_state().clone()
        np.random.seed(3000 + seed)
        torch.manual_seed(9000 + seed)
        moved[0]
        assert np.array_equal(st_np, np.random.get_state()[1]) and torch.equal(st_t, torch.get_rng_state()), seed
        lower = int(rec["u_tab"][0]) in (datapipe.DOWN_UPPC, datapipe.DOWN_DOWNPC)
        kind = rec["kind"] + ("_lower" if lower else "")
        if rec["kind"] == "single" and not np.isnan(accept.cd):
            kind = "single_after_rejection"
        if not np.isnan(accept.cd) and abs(accept.cd - 0.015) < 2e-3:
            continue                         # (a decision this close to the threshold is not a fixture)
        print(f"seed {seed}: {kind} (cd {accept.cd:.4f})", flush=True)
        if kind in kept:
            continue
        kept[kind] = seed
        k = f"s{seed}_"
        G[k + "kind"] = np.array(kind)
        for name in ("normal1", "z1", "normal2", "z2", "u_tab", "d_tab", "twist"):
            G[k + name] = np.asarray(rec[name])
        G[k + "s_u"], G[k + "s_d"] = np.int64(rec["s_u"]), np.int64(rec["s_d"])
        G[k + "cd"] = np.float64(accept.cd)
        G[k + "down"], G[k + "mup"], G[k + "igt"], G[k + "up"] = down.numpy(), mup.numpy(), igt.numpy(), up.numpy()
        G[k + "downb"], G[k + "upb"] = downb.numpy(), upb.numpy()
        G[k + "fpc_idx"], G[k + "rpc_idx"] = fpc_idx.numpy(), rpc_idx.numpy()
    G["seeds"] = np.array(sorted(kept.values()), np.int64)
    print("kept", kept)

    # BuildingDataset: two given 1024-point clouds per item (buildings_{f,r}_train1024.npy in the reference)
    rng = np.random.default_rng(777)
    fpcs = rng.random((3, N, 3), dtype=np.float32)
    rpcs = (rng.random((3, N, 3), dtype=np.float32) * np.float32([1, 1, 0.4]) + np.float32([0, 0, 0.9])).astype(np.float32)
    b = object.__new__(ds.BuildingDataset)
    b.fpcs, b.rpcs = fpcs, rpcs
    movedb = ds.MovedCADDataset2(b, tr.RandomTransformSE3(0.8))
    G["b_fpcs"], G["b_rpcs"] = fpcs, rpcs
    for i in range(3):
        torch.manual_seed(400 + i)
        down, mup, igt, up, downb, upb, fpc_idx, rpc_idx = movedb[i]
        torch.manual_seed(400 + i)
        x = torch.randn(1, 6)
        x = x / x.norm(p=2, dim=1, keepdim=True) * 0.8
        k = f"b{i}_"
        G[k + "twist"] = x.reshape(6).numpy()
        G[k + "down"], G[k + "mup"], G[k + "igt"], G[k + "up"] = down.numpy(), mup.numpy(), igt.numpy(), up.numpy()
        G[k + "downb"], G[k + "upb"] = downb.numpy(), upb.numpy()
        G[k + "fpc_idx"], G[k + "rpc_idx"] = fpc_idx.numpy(), rpc_idx.numpy()
    np.savez_compressed(os.path.join(OUT, "data2.npz"), **G)
    print("wrote", os.path.join(OUT, "data2.npz"), os.path.getsize(os.path.join(OUT, "data2.npz")), "bytes")


if __name__ == "__main__":
    main()
