"""GPU training-pair construction (puzzlenet_amd/datapipe.py): pairs per second for raw clouds of M points cut, sampled
to N, labelled and moved — next to the training step's consumption rate, and to the reference-style numpy FPS of ONE
piece on one host core (the dominant cost of the reference's per-sample CPU pipeline)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import datapipe
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for (B, M, N) in [(64, 6000, 1024), (64, 10000, 2048), (64, 12000, 2048)]:
    raw = (torch.rand(B, M, 3, generator=g) - 0.5).to(dev)
    normal = torch.rand(B, 3, generator=g, dtype=torch.float64).to(dev)
    z = (torch.rand(B, generator=g, dtype=torch.float64) / 3 - 0.4).to(dev)      # cuts that leave both sides populated
    s = torch.zeros(B, dtype=torch.int64, device=dev)
    tw = torch.randn(B, 6, generator=g); tw = (tw / tw.norm(dim=1, keepdim=True) * 0.8).to(dev)
    for _ in range(2):
        out, ok = datapipe.make_pairs(raw, normal, z, s, s, tw, n=N)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); it = 5
    for _ in range(it):
        out, ok = datapipe.make_pairs(raw, normal, z, s, s, tw, n=N)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / it
    print('B=%d M=%d -> N=%d: %.2f ms per batch = %.0f pairs/s (valid cuts %d/%d)' % (B, M, N, dt * 1e3, B / dt, int(ok.sum()), B), flush=True)
# reference-style numpy FPS of one 5000-point piece to 2048 (dataset.py:1147-1163)
pts = (np.random.rand(5000, 3) - 0.5).astype(np.float32)
t0 = time.perf_counter()
distance = np.ones((5000,)) * 1e10; far = 0; cent = np.zeros(2048)
for i in range(2048):
    cent[i] = far
    d = np.sum((pts - pts[far]) ** 2, -1)
    m = d < distance; distance[m] = d[m]; far = np.argmax(distance, -1)
print('numpy FPS 5000 -> 2048 on one host core: %.1f ms per piece' % ((time.perf_counter() - t0) * 1e3))
