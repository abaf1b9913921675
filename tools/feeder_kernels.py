"""Per-kernel device time of datapipe.PairFeeder building batches alone (library timer): which launches make a batch's ~4 ms."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import datapipe, ops

dev = torch.device("cuda:0")
B, N, M = 64, 2048, int(os.environ.get("M", 10000))
rng = np.random.RandomState(0)
u = rng.randn(B, M, 3).astype(np.float32); u /= np.linalg.norm(u, axis=2, keepdims=True)
raw = (u * (0.25 + 0.2 * rng.rand(B, 1, 3).astype(np.float32))).astype(np.float32)
feeder = datapipe.PairFeeder(raw, dev, n=N, seed=0)
for _ in range(3):
    feeder.next_batch()
torch.cuda.synchronize()
nb = 5
ops.ktimer_start()
for _ in range(nb):
    b = feeder.next_batch()
torch.cuda.synchronize()
rows = ops.ktimer_stop()
print(f"{sum(ms for _, ms in rows.values()) / nb:.3f} ms of library kernels per batch")
for name, (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms / nb:8.3f} ms/batch {n / nb:5.1f} x {1e3 * ms / n:8.1f} us  {name[:100]}")
