"""Per-(entry point, shape) device time of one training step (HIP events around every C-ABI call).

    python tools/stage_breakdown.py [--batch 64] [--points 2048] [--steps 2]
"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Cfg
from puzzlenet_amd import engine, model5_b, ops, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--points", type=int, default=2048)
ap.add_argument("--steps", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = Cfg(); cfg.num_points = a.points
torch.manual_seed(0)
model = model5_b.TouchedRegraster(cfg).to(dev)
model.two_streams = False
batch = synthetic.make_batch(a.batch, a.points, dev, seed=1234)
runner = engine.TrainStep(model, batch, cfg.lr, world=1, use_graph=False, warmup=2)
for _ in range(2):
    runner.step()
ops.KernelTimer.shapes = {}
ops.KernelTimer.start()
for _ in range(a.steps):
    runner.step()
ops.KernelTimer.stop()
rows = sorted(((sum(v) / a.steps, len(v) // a.steps, k) for k, v in ops.KernelTimer.shapes.items()), reverse=True)
tot = sum(r[0] for r in rows)
print("total C-ABI time %.3f ms/step" % tot)
for ms, n, k in rows:
    print("%8.3f ms  x%-3d %-28s %s" % (ms, n, k[0], k[1:]))
