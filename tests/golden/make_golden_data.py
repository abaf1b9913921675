#!/usr/bin/env python3
"""Data-pipeline golden fixtures: runs the REFERENCE's own functions on CPU.

    python tests/golden/make_golden_data.py      ->  tests/golden/data.npz

The single-cut path of `CADDataset` (dataset.py:1165-1190: plane_split :761-775, fps :1147-1163, get_boundary
:1357-1367) followed by `MovedCADDataset2.__getitem__` (:98-105) with `RandomTransformSE3(0.8)`
(se_math/transforms.py:151-197), on seeded synthetic clouds.  The random draws the reference makes internally
(plane normal / offset, the two FPS start indices, the twist) are recovered by replaying its generators from the same
state and stored next to the outputs, so that the GPU pipeline can be driven with exactly those draws.
Same harness rules as make_golden_model.py (sys.modules placeholders only, nothing of the reference is copied).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden_model as gm  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
N = 1024          # the reference hard-wires 1024 in fps / get_boundary of this path
M = 6000          # raw points per cloud
CASES = 4


def main():
    gm.import_reference_model()          # installs the placeholders (open3d, plyfile, ...) and the sys.path entry
    import dataset as ds
    import se_math.transforms as tr
    rng = np.random.default_rng(4242)
    G = {"N": np.int64(N), "M": np.int64(M), "cases": np.int64(CASES)}
    ref = object.__new__(ds.CADDataset)   # methods only: no dataset file is read
    for c in range(CASES):
        pc = (rng.random((M, 3), dtype=np.float32) - np.float32(0.5)).astype(np.float32)
        np.random.seed(1000 + c)
        while True:
            st = np.random.get_state()
            up, down = ds.plane_split(pc)                                   # :761-775
            if up.shape[0] >= N and down.shape[0] >= N:
                break
        np.random.set_state(st)                                             # replay the draws of the accepted cut
        normal = np.random.rand(3, 1)
        z = np.random.rand(1) / 3
        st = np.random.get_state()
        up_n = ds.CADDataset.fps(ref, up, N)                                # :1147-1163
        np.random.set_state(st)
        s_up = np.random.randint(0, up.shape[0])
        st = np.random.get_state()
        down_n = ds.CADDataset.fps(ref, down, N)
        np.random.set_state(st)
        s_down = np.random.randint(0, down.shape[0])
        up_t = torch.from_numpy(up_n).to(torch.float32)
        down_t = torch.from_numpy(down_n).to(torch.float32)
        fpcb, rpcb, fpc_idx, rpc_idx = ds.CADDataset.get_boundary(ref, down_t, up_t)   # :1357-1367
        cd1, cd2 = ds.CADDataset.chamfer_loss(ref, down_t.unsqueeze(0), up_t.unsqueeze(0))
        T = tr.RandomTransformSE3(0.8)
        torch.manual_seed(7000 + c)
        mup = T(up_t)                                                       # dataset.py:99-100
        igt = T.igt
        x = T.get_x()
        k = f"c{c}_"
        G[k + "raw"] = pc
        G[k + "normal"], G[k + "z"] = normal.reshape(3), z.reshape(1)
        G[k + "n_up"], G[k + "n_down"] = np.int64(up.shape[0]), np.int64(down.shape[0])
        G[k + "s_up"], G[k + "s_down"] = np.int64(s_up), np.int64(s_down)
        G[k + "twist"] = x.numpy().reshape(6)
        G[k + "up"], G[k + "down"] = up_n.astype(np.float32), down_n.astype(np.float32)
        G[k + "fpcb"], G[k + "rpcb"] = fpcb.numpy(), rpcb.numpy()
        G[k + "fpc_idx"], G[k + "rpc_idx"] = fpc_idx.numpy(), rpc_idx.numpy()
        G[k + "cd_over_up"], G[k + "cd_over_down"] = cd1.numpy().reshape(-1), cd2.numpy().reshape(-1)
        G[k + "mup"], G[k + "igt"] = mup.numpy(), igt.numpy()
        print(f"case {c}: up {up.shape[0]} down {down.shape[0]} start {s_up}/{s_down}", flush=True)
    np.savez_compressed(os.path.join(OUT, "data.npz"), **G)
    print("wrote", os.path.join(OUT, "data.npz"))


if __name__ == "__main__":
    main()
