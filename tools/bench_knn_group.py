#!/usr/bin/env python3
"""Micro-benchmark of the kNN + group stage (SURVEY 8(d) bytes) on the bench clouds: every entry point that can serve
pointnet_util.sample_and_group(npoint, 0, 32, xyz, points, knn=True) after the FPS, per level.

    python tools/bench_knn_group.py [--batch 64] [--points 2048] [--reps 20]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def stage_bytes(N, S, K, D):
    return 12 * N + 4 * N * D + 12 * S + 8 * S * K + 4 * S * K * (D + 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    from puzzlenet_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B = args.batch
    g = torch.Generator().manual_seed(1)
    res = {}
    for (N, S, D) in ((args.points, 512, 64), (512, 256, 128)):
        xyz = torch.rand(B, N, 3, generator=g).to(dev)
        feat = torch.randn(B, N, D, generator=g).to(dev)
        fps = ops.farthest_point_sample(xyz, S, torch.zeros(B, dtype=torch.long, device=dev))
        new_xyz = ops.index_points(xyz, fps).contiguous()
        idx = torch.empty((B, S, 32), dtype=torch.int64, device=dev)
        out = torch.empty((B, S, 32, 3 + D), dtype=torch.float32, device=dev)
        p = ops._p
        st = ops._stream()
        calls = {
            "knn": lambda: _lib.call("pzn_knn_f32", p(xyz), p(new_xyz), B, N, S, 32, p(idx), st),
            "group": lambda: _lib.call("pzn_group_fwd_f32", p(xyz), p(feat), p(new_xyz), p(idx), B, N, S, 32, D, p(out),
                                       None, st),
        }
        if hasattr(lib, "pzn_knn_group_f32"):
            calls["knn_group"] = lambda: _lib.call("pzn_knn_group_f32", p(xyz), p(feat), p(new_xyz), B, N, S, D, p(idx),
                                                   p(out), None, st)
        nbytes = B * stage_bytes(N, S, 32, D)
        for name, fn in calls.items():
            try:
                fn()
            except Exception as e:      # noqa: BLE001
                res[f"{name}_N{N}"] = str(e)
                continue
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / args.reps
            res[f"{name}_N{N}"] = {"ms": round(ms, 4), "GBps_on_stage_bytes": round(nbytes / ms / 1e6, 1)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
