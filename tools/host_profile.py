"""Where the host's 8-9 ms per step go: cProfile of K eager steps (the GPU runs behind; nothing synchronises inside)."""
import cProfile
import os
import pstats
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
for _ in range(3):
    r.step()
torch.cuda.synchronize()
K = 10
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    r.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
