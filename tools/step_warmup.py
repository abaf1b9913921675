"""ms per step in consecutive windows of 10 steps from process start: how long a fresh process / box takes to reach its
steady step time (what a default `python bench.py` run sees with its 5 warm-up steps)."""
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
r = engine.TrainStep(model, batch, cfg.lr, world=1)
out = []
for w in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        r.step()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) * 100)
print("ms per step, windows of 10 steps:", " ".join(f"{v:.2f}" for v in out))
