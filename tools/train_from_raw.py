"""End to end on the GPU: raw clouds -> datapipe.make_pairs (cut, FPS, boundary labels, random motion: the reference's
per-sample CPU pipeline, dataset.py:1165-1190 + :98-105) -> engine.TrainStep, a fresh batch every step.

    python tools/train_from_raw.py [--batch 64] [--points 2048] [--raw 10000] [--steps 20]
"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Cfg
from puzzlenet_amd import datapipe, engine, model5_b

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--points", type=int, default=2048)
ap.add_argument("--raw", type=int, default=10000)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, N, M = a.batch, a.points, a.raw
np.random.seed(0); torch.manual_seed(0)
# synthetic "objects": points on an ellipsoid shell around the origin, one shape per sample
u = np.random.randn(B, M, 3).astype(np.float32)
u /= np.linalg.norm(u, axis=2, keepdims=True)
raw_h = (u * (0.25 + 0.2 * np.random.rand(B, 1, 3).astype(np.float32))).astype(np.float32)
raw = torch.from_numpy(raw_h).to(dev)

def draw_batch():
    d = [datapipe.draws_like_reference(raw_h[i], n=N, mag=0.8) for i in range(B)]
    t = lambda k, dt: torch.from_numpy(np.stack([np.asarray(x[k]) for x in d])).to(dt).to(dev)
    return t("normal", torch.float64), t("z", torch.float64).reshape(-1), t("s_up", torch.int64), t("s_down", torch.int64), t("twist", torch.float32)

cfg = Cfg(); cfg.num_points = N
model = model5_b.TouchedRegraster(cfg).to(dev)
normal, z, su, sd, tw = draw_batch()
batch, ok = datapipe.make_pairs(raw, normal, z, su, sd, tw, n=N)
runner = engine.TrainStep(model, list(batch), cfg.lr, world=1, use_graph=False)
runner.step(); torch.cuda.synchronize()
t_pipe = t_step = 0.0
for it in range(a.steps):
    t0 = time.perf_counter()
    normal, z, su, sd, tw = draw_batch()                     # host: the reference's draws
    batch, ok = datapipe.make_pairs(raw, normal, z, su, sd, tw, n=N)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    runner.batch = list(batch)
    loss = runner.step()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    t_pipe += t1 - t0; t_step += t2 - t1
    if it % 5 == 0 or it == a.steps - 1:
        print("step %3d  loss %.4f  pairs built in %.1f ms, trained in %.1f ms" % (it, float(loss), (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
print("mean: build %.1f ms (host draws + GPU pipeline), train %.1f ms per %d pairs" % (t_pipe / a.steps * 1e3, t_step / a.steps * 1e3, B))
