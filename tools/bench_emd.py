"""Fused EMD entry point on the training shapes: launch time on independent uniform clouds and on a cloud against
its slightly perturbed copy (the two regimes of the loss terms)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, name, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    print('%-44s %8.3f ms' % (name, a.elapsed_time(b) / iters), flush=True)
g = torch.Generator().manual_seed(0)
for (B, n, m) in [(64, 2048, 2048), (64, 1024, 1024), (16, 4096, 4096), (64, 128, 128)]:
    x1 = torch.rand(B, n, 3, generator=g).to(dev); x2 = torch.rand(B, m, 3, generator=g).to(dev)
    timeit(lambda: ops.emd_fused(x1, x2), f'emd_fused {B}x{n}x{m} independent')
    if n == m:
        x3 = x1 + 0.01 * torch.randn(B, n, 3, generator=g).to(dev)
        timeit(lambda: ops.emd_fused(x1, x3), f'emd_fused {B}x{n}x{m} near copy')
