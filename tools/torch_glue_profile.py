"""Which torch ops (not C-ABI kernels) take device time in one training step: torch.profiler table."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Cfg
from puzzlenet_amd import engine, model5_b, synthetic
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
cfg = Cfg(); cfg.num_points = 2048
torch.manual_seed(0)
model = model5_b.TouchedRegraster(cfg).to(dev)
batch = synthetic.make_batch(64, 2048, dev, seed=1234)
runner = engine.TrainStep(model, batch, cfg.lr, world=1, use_graph=False, warmup=2)
for _ in range(2):
    runner.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for _ in range(2):
        runner.step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
