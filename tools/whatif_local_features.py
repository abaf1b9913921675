"""What the encoders' per-point MLP + BatchNorm (model5_b.py:447-448) costs the step: the step timed as it is and with
local_features() replaced by a leaf tensor holding the same values (everything downstream, including the gradients that
flow into it, unchanged).  Measurement only: an upper bound for fusing those four launches each way."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Cfg
from puzzlenet_amd import engine, model5_b, synthetic

dev = torch.device("cuda:0")


def run(free):
    cfg = Cfg(); cfg.num_points = 2048
    torch.manual_seed(0)
    model = model5_b.TouchedRegraster(cfg).to(dev)
    batch = synthetic.make_batch(64, 2048, dev, seed=1234)
    if free:
        for enc, cloud in ((model.Encoder, batch[0]), (model.Encoder2, batch[1])):
            with torch.no_grad():
                val = enc.local_features(cloud).detach()
            enc.local_features = (lambda v: (lambda xyz: v.clone().requires_grad_(True)))(val)
    r = engine.TrainStep(model, batch, cfg.lr, world=1)
    for _ in range(10):
        r.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        r.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 40 * 1e3


for rep in range(3):
    print("as it is: %.3f ms   local_features free (one 33.5 MB copy each): %.3f ms" % (run(False), run(True)))
