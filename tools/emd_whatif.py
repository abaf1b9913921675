"""What the N x N EMD costs the step: the training step timed as it is and with the N x N term replaced by a free
stand-in (same graph shape: a custom node whose backward returns zeros).  Measurement only."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Cfg
from puzzlenet_amd import engine, model5_b, synthetic

dev = torch.device("cuda:0")


class _Free(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return torch.zeros(a.shape[0], device=a.device)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return torch.zeros_like(a), torch.zeros_like(b)


def run(free):
    cfg = Cfg(); cfg.num_points = 2048
    torch.manual_seed(0)
    model = model5_b.TouchedRegraster(cfg).to(dev)
    batch = synthetic.make_batch(64, 2048, dev, seed=1234)
    real = model5_b.earth_mover_distance
    if free:
        model5_b.earth_mover_distance = lambda a, b, transpose=True: _Free.apply(a, b) if a.shape[1] >= 1024 else real(a, b, transpose=transpose)
    try:
        r = engine.TrainStep(model, batch, cfg.lr, world=1)
        for _ in range(10):
            r.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            r.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 40 * 1e3
    finally:
        model5_b.earth_mover_distance = real


for rep in range(2):
    print("with the N x N EMD: %.3f ms   without: %.3f ms" % (run(False), run(True)))
