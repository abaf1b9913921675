import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from bench import Cfg
from puzzlenet_amd import engine, model5_b, synthetic
dev = torch.device("cuda:0")
for two in (False, True):
    cfg = Cfg(); cfg.num_points = 2048
    torch.manual_seed(0)
    model = model5_b.TouchedRegraster(cfg).to(dev)
    model.two_streams = two
    batch = synthetic.make_batch(64, 2048, dev, seed=1234)
    r = engine.TrainStep(model, batch, cfg.lr, world=1, use_graph=False, warmup=2)
    for _ in range(3): r.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): r.step()
    torch.cuda.synchronize(); print("two_streams", two, "%.2f ms/step" % ((time.perf_counter() - t0) * 100))
