"""GPU time of the sections of an eager training step (events on the main stream, which joins the side stream at the end
of each section): forward + losses, backward, all-reduce / Adam.  No kernel tracing: the host runs ahead as in bench.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Cfg  # noqa: E402
from puzzlenet_amd import engine, model5_b, synthetic  # noqa: E402

dev = torch.device("cuda:0")
cfg = Cfg()
cfg.num_points = 2048
torch.manual_seed(0)
model = model5_b.TouchedRegraster(cfg).to(dev)
batch = synthetic.make_batch(64, 2048, dev, seed=1234)
r = engine.TrainStep(model, batch, cfg.lr, world=1, use_graph=False, warmup=2)
marks = []
orig_ts = model.training_step
orig_heads = model._heads


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def ts(*a, **k):
    marks.append(("step start", ev()))
    out = orig_ts(*a, **k)
    marks.append(("forward + losses enqueued (main stream)", ev()))
    return out


def heads(*a, **k):
    marks.append(("encoder 1 forward done (main stream)", ev()))
    out = orig_heads(*a, **k)
    marks.append(("heads done, streams joined", ev()))
    return out


model.training_step = ts
model._heads = heads
for _ in range(3):
    r.step()
torch.cuda.synchronize()
K = 10
acc = {}
for _ in range(K):
    marks.clear()
    r.step()
    marks.append(("backward + Adam done", ev()))
    torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
tot = 0.0
for n, v in acc.items():
    print(f"{v / K:7.3f} ms  -> {n}")
    tot += v / K
print(f"{tot:7.3f} ms  total (step boundary to step boundary on the main stream)")
