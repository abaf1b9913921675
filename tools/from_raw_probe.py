"""The step fed from raw clouds (datapipe.PairFeeder) against the step on a resident batch, and where the difference goes:
host time of next_batch(), device time of a batch built alone, step time with the feeder beside it."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Cfg
from puzzlenet_amd import datapipe, engine, model5_b, synthetic

dev = torch.device("cuda:0")
B, N, M = 64, 2048, int(os.environ.get("M", 10000))
rng = np.random.RandomState(0)
u = rng.randn(B, M, 3).astype(np.float32); u /= np.linalg.norm(u, axis=2, keepdims=True)
raw = (u * (0.25 + 0.2 * rng.rand(B, 1, 3).astype(np.float32))).astype(np.float32)
cfg = Cfg(); cfg.num_points = N
torch.manual_seed(0)
model = model5_b.TouchedRegraster(cfg).to(dev)
feeder = datapipe.PairFeeder(raw, dev, n=N, seed=0)
for _ in range(3):
    b = feeder.next_batch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    b = feeder.next_batch()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"feeder alone: host {1e2 * (t1 - t0):.2f} ms per batch, device {1e2 * (t2 - t0):.2f} ms per batch, ok {float(b.ok.float().mean()):.3f}")

def run(fed, steps=20):
    r = engine.TrainStep(model, feeder.next_batch() if fed else synthetic.make_batch(B, N, dev, seed=1), cfg.lr, world=1)
    nxt = feeder.next_batch() if fed else None
    for _ in range(5):
        r.step(next_batch=nxt); nxt = feeder.next_batch() if fed else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        r.step(next_batch=nxt); nxt = feeder.next_batch() if fed else None
    te = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    r.close()
    return 1e3 * dt / steps, 1e3 * te / steps

for rep in range(2):
    a, ae = run(False); f, fe = run(True)
    print(f"resident {a:.2f} ms/step (host {ae:.2f})   from raw {f:.2f} ms/step (host {fe:.2f})   ratio {a / f:.3f}")
print("stream priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
