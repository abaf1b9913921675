import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import Cfg
from puzzlenet_amd import engine, model5_b, synthetic
dev = torch.device("cuda:0")
cfg = Cfg(); cfg.num_points = 2048
torch.manual_seed(0)
model = model5_b.TouchedRegraster(cfg).to(dev)
batch = synthetic.make_batch(64, 2048, dev, seed=1234)
torch.manual_seed(1000)
r = engine.TrainStep(model, batch, cfg.lr, world=1)
ts = []
for i in range(60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("synchronised per-step ms:", " ".join(f"{t:.2f}" for t in ts))
