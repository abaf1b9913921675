"""Host time of a step, forward AND backward: cProfile with autograd's worker threads off (the backward's Python functions run
on the calling thread and are seen)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Cfg  # noqa: E402
from puzzlenet_amd import engine, model5_b, synthetic  # noqa: E402
dev = torch.device("cuda:0")
cfg = Cfg(); cfg.num_points = 2048
torch.manual_seed(0)
model = model5_b.TouchedRegraster(cfg).to(dev)
batch = synthetic.make_batch(64, 2048, dev, seed=1234)
r = engine.TrainStep(model, batch, cfg.lr, world=1)
torch.autograd.set_multithreading_enabled(False)
for _ in range(3): r.step()
torch.cuda.synchronize()
K = 10
pr = cProfile.Profile(); pr.enable()
for _ in range(K): r.step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(45)
