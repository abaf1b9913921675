"""Does HIP-graph replay of the training step reproduce the eager step?  Same seeds, same batch, loss of the first steps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Cfg
from puzzlenet_amd import engine, model5_b, synthetic

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def run(use_graph, steps=6):
    cfg = Cfg(); cfg.num_points = 2048
    torch.manual_seed(0)
    model = model5_b.TouchedRegraster(cfg).to(dev)
    model.two_streams = False
    batch = synthetic.make_batch(B, 2048, dev, seed=1234)
    torch.manual_seed(1000)
    r = engine.TrainStep(model, batch, cfg.lr, world=1, use_graph=use_graph, warmup=2)
    out = []
    for _ in range(steps):
        out.append(float(r.step()))
    r.close()
    return out


e = run(False)
g = run(True)
print("eager", ["%.6f" % v for v in e])
print("graph", ["%.6f" % v for v in g])
