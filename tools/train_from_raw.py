"""End to end on the GPU: resident raw clouds -> datapipe.PairFeeder (plane cut with re-draw, FPS of both pieces, boundary
labels, random motion on a background stream: the reference's loader processes, train.py:101-104 + dataset.py:1165-1190 +
:98-105) -> engine.TrainStep, a fresh batch every step, nothing synchronises.

    python tools/train_from_raw.py [--batch 64] [--points 2048] [--raw 10000] [--steps 20]
"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Cfg
from puzzlenet_amd import datapipe, engine, model5_b

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--points", type=int, default=2048)
ap.add_argument("--raw", type=int, default=10000)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, N, M = a.batch, a.points, a.raw
rng = np.random.RandomState(0)
u = rng.randn(B, M, 3).astype(np.float32)
u /= np.linalg.norm(u, axis=2, keepdims=True)
raw = (u * (0.25 + 0.2 * rng.rand(B, 1, 3).astype(np.float32))).astype(np.float32)      # ellipsoid shells, one per sample
cfg = Cfg(); cfg.num_points = N
torch.manual_seed(0)
model = model5_b.TouchedRegraster(cfg).to(dev)
feeder = datapipe.PairFeeder(raw, dev, n=N, seed=0)
runner = engine.TrainStep(model, feeder.next_batch(), cfg.lr, world=1)
nxt = feeder.next_batch()
losses, mem = [], []
for it in range(a.steps + 3):
    if it == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    losses.append(runner.step(next_batch=nxt))
    nxt = feeder.next_batch()
    if it % 50 == 0:
        mem.append(torch.cuda.memory_allocated() >> 20)      # (no synchronisation: the allocator's own count)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("loss first / last: %.4f / %.4f" % (float(losses[0]), float(losses[-1])))
ls = torch.stack([l.detach().float().reshape(()) for l in losses])
print("every loss finite:", bool(torch.isfinite(ls).all()), " MiB allocated every 50 steps:", mem)
print("%.2f ms per step of %d fresh pairs from %d-point raw clouds = %.0f pairs/s" % (1e3 * dt / a.steps, B, M, B * a.steps / dt))
runner.close(); feeder.close()
