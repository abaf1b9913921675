"""List host-synchronising torch calls inside one training step (torch.cuda.set_sync_debug_mode)."""
import os, sys, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Cfg
from puzzlenet_amd import engine, model5_b, synthetic
dev = torch.device("cuda:0")
cfg = Cfg(); cfg.num_points = 2048
torch.manual_seed(0)
model = model5_b.TouchedRegraster(cfg).to(dev)
batch = synthetic.make_batch(16, 2048, dev, seed=1234)
r = engine.TrainStep(model, batch, cfg.lr, world=1, use_graph=False, warmup=2)
for _ in range(2):
    r.step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode(1)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    r.step()
torch.cuda.set_sync_debug_mode(0)
import traceback
for x in w:
    print(str(x.message)[:100], "|", x.filename.split("/")[-1], x.lineno)
print(len(w), "warnings")
