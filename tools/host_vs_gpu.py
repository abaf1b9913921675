"""Is the eager step host-bound?  Host time to enqueue K steps vs time until the GPU has finished them."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Cfg
from puzzlenet_amd import engine, model5_b, synthetic
dev = torch.device("cuda:0")
cfg = Cfg(); cfg.num_points = 2048
torch.manual_seed(0)
model = model5_b.TouchedRegraster(cfg).to(dev)
batch = synthetic.make_batch(64, 2048, dev, seed=1234)
r = engine.TrainStep(model, batch, cfg.lr, world=1, use_graph=False, warmup=2)
for _ in range(3):
    r.step()
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    r.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step, until GPU done %.2f ms/step (GPU tail after the last enqueue: %.2f ms)" % (
    (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3, (t2 - t1) * 1e3))
