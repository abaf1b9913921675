"""Per-kernel device time of one training step from the library's own timer (pzn_ktimer_*): name, launches per step, us per
launch, ms per step - the encoders one after the other (exclusive times) or on two streams (PZN two_streams)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Cfg
from puzzlenet_amd import engine, model5_b, ops, synthetic

dev = torch.device("cuda:0")
B, N = int(os.environ.get("B", 64)), int(os.environ.get("N", 2048))
cfg = Cfg(); cfg.num_points = N
torch.manual_seed(0)
model = model5_b.TouchedRegraster(cfg).to(dev)
model.two_streams = os.environ.get("TWO", "0") == "1"
batch = synthetic.make_batch(B, N, dev, seed=1234)
r = engine.TrainStep(model, batch, cfg.lr, world=1)
for _ in range(int(os.environ.get("WARM", 3))):
    r.step()
torch.cuda.synchronize()
steps = 3
ops.ktimer_start()
for _ in range(steps):
    r.step()
rows = ops.ktimer_stop()
tot = sum(ms for _, ms in rows.values())
print(f"{len(rows)} kernels, {sum(n for n, _ in rows.values()) / steps:.0f} launches and {tot / steps:.3f} ms of kernel time per step")
for name, (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms / steps:8.3f} ms/step {n / steps:6.1f} x {1e3 * ms / n:8.1f} us  {name[:110]}")
