"""The encoder's per-point stem in isolation: one launch each way (csrc/stem.hip) against the 2 linear + 2 BatchNorm launches
each way it replaces, forward and forward+backward, at the bench shape.   python tools/stem_time.py [B [N]]"""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    from puzzlenet_amd import ops
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    xyz = torch.rand(B, N, 3, device=dev) * 2 - 1
    go = torch.randn(B, N, 64, device=dev)
    mods = [nn.Linear(3, 64), nn.BatchNorm1d(N), nn.Linear(64, 64), nn.BatchNorm1d(N)]
    mods = [m.to(dev).train() for m in mods]

    def fused():
        return ops.stem(xyz, *mods)

    def unfused():
        a = ops.bn_points_relu(ops.linear(xyz, mods[0].weight, mods[0].bias), mods[1])
        return ops.bn_points_relu(ops.linear(a, mods[2].weight, mods[2].bias), mods[3])

    def timed(fn, backward, iters=50):
        for _ in range(5):
            y = fn()
            if backward:
                y.backward(go)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            y = fn()
            if backward:
                y.backward(go)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    for name, fn in (("one launch each way", fused), ("2 linear + 2 BatchNorm", unfused)):
        f = timed(fn, False)
        fb = timed(fn, True)
        print(f"{name:26s} B={B} N={N}: forward {f:7.1f} us   forward+backward {fb:7.1f} us")
