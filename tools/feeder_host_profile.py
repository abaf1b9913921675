import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from puzzlenet_amd import datapipe
dev = torch.device("cuda:0")
B, N, M = 64, 2048, 10000
rng = np.random.RandomState(0)
u = rng.randn(B, M, 3).astype(np.float32); u /= np.linalg.norm(u, axis=2, keepdims=True)
raw = (u * (0.25 + 0.2 * rng.rand(B, 1, 3).astype(np.float32))).astype(np.float32)
feeder = datapipe.PairFeeder(raw, dev, n=N, seed=0)
for _ in range(5): feeder.next_batch()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(20): feeder.next_batch()
print("host ms per batch", (time.perf_counter()-t0)/20*1e3)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): feeder.next_batch()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
