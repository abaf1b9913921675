"""Which torch ops (not C-ABI kernels) take device time in one training step: torch.profiler table grouped by
operator and input shape."""
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
model.two_streams = False
batch = synthetic.make_batch(64, 2048, dev, seed=1234)
runner = engine.TrainStep(model, batch, cfg.lr, world=1, use_graph=False, warmup=2)
for _ in range(2):
    runner.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2):
        runner.step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows) / 2e3
print("aten self device time per step: %.3f ms" % tot)
for e in rows[:45]:
    print("%8.3f ms  x%-4d %-28s %s" % (e.self_device_time_total / 2e3, e.count // 2, e.key, str(e.input_shapes)[:110]))

# the same ops by the innermost package frame that issued them (forward-side ops only: autograd's own ops have no stack)
if os.environ.get("PZN_GLUE_STACKS", "1") != "0":
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof2:
        runner.step()
        torch.cuda.synchronize()
    agg = {}
    for ev in prof2.events():
        if not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
            continue
        where = next((s for s in (ev.stack or []) if "puzzlenet_amd" in s or "bench.py" in s), "(autograd / no frame)")
        key = (ev.name, str(ev.input_shapes)[:70], where.split("puzzlenet_amd/")[-1][:60])
        a = agg.setdefault(key, [0.0, 0])
        a[0] += ev.self_device_time_total
        a[1] += 1
    print("\nby issuing frame (one step):")
    for (name, shp, where), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
        print("%7.1f us x%-3d %-22s %-70s %s" % (t, n, name, shp, where))
