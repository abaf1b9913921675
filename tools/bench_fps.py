"""FPS per round: the main form (LDS image) at the model's shapes, and the background form (datapipe: no image, padded pieces
with counts) on pieces of `real` points inside a 10 000-row buffer."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import ops

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for (B, N, S) in [(128, 2048, 512), (128, 512, 256), (128, 4096, 512), (64, 8192, 512)]:
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    st = torch.zeros(B, dtype=torch.long, device=dev)
    ms = timed(lambda: ops.farthest_point_sample(xyz, S, st))
    print(f"main       B {B:4d} N {N:5d} S {S:4d}: {1e3 * ms:8.1f} us  {1e3 * ms / S:.3f} us per round")
for (B, N, real, S) in [(128, 10000, 2048, 2048), (128, 10000, 4000, 2048), (128, 10000, 5000, 2048), (128, 10000, 7952, 2048), (1, 10000, 5000, 2048),
                        (128, 10000, 10000, 2048)]:
    xyz = torch.rand(B, N, 3, generator=g)
    xyz[:, real:] = xyz[:, :1]
    xyz = xyz.to(dev)
    st = torch.zeros(B, dtype=torch.long, device=dev)
    cnt = torch.full((B,), real, dtype=torch.long, device=dev)
    ms = timed(lambda: ops.farthest_point_sample(xyz, S, st, background=True, counts=cnt), reps=3)
    mb = timed(lambda: ops.farthest_point_sample(xyz, S, st, background=True, counts=cnt, max_count=7952), reps=3) if real <= 7952 else float("nan")
    print(f"background B {B:4d} N {N:5d} real {real:5d} S {S:4d}: {1e3 * ms:8.1f} us  {1e3 * ms / S:.3f} us per round;  counts bounded by 7952: {1e3 * mb / S:.3f}")
