"""Which lines of puzzlenet_amd launch the ATen kernels of a step (copy_, add, mean, fill_, index ...): torch.profiler on the CPU
side, grouped by (op, input shapes, innermost frame inside the package or - where the build records no Python stacks - the
enclosing profiler events: the autograd node whose backward ran the op).  The device cost of each group is in the
rocprofv3 kernel table; this names the call sites.   python tools/aten_sources.py [steps]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    from bench import Cfg
    from puzzlenet_amd import engine, model5_b, synthetic
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device("cuda:0")
    cfg = Cfg()
    cfg.num_points = 2048
    torch.manual_seed(0)
    model = model5_b.TouchedRegraster(cfg).to(dev)
    batch = synthetic.make_batch(64, 2048, dev, seed=1234)
    r = engine.TrainStep(model, batch, cfg.lr, world=1)
    for _ in range(3):
        r.step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
        for _ in range(K):
            r.step()
        torch.cuda.synchronize()
    want = ("aten::copy_", "aten::add", "aten::add_", "aten::mean", "aten::sum", "aten::fill_", "aten::zero_", "aten::index",
            "aten::cat", "aten::mul", "aten::sub", "aten::div", "aten::gather", "aten::scatter_", "aten::neg", "aten::clone",
            "aten::contiguous", "aten::index_put_", "aten::where", "aten::sqrt", "aten::pow", "aten::topk", "aten::sort")
    groups = collections.Counter()
    for ev in prof.events():
        if ev.name not in want:
            continue
        where = ""
        for fr in ev.stack or []:
            if "puzzlenet_amd" in fr or "bench.py" in fr:
                where = fr.strip()
                break
        if not where:      # no Python stack (the backward runs in the engine): name the enclosing profiler events instead
            chain, par = [], ev.cpu_parent
            while par is not None and len(chain) < 3:
                if not par.name.startswith("aten::"):
                    chain.append(par.name.replace("autograd::engine::evaluate_function: ", "bwd of "))
                par = par.cpu_parent
            where = " < ".join(chain) or "(top level)"
        shapes = str([tuple(s) for s in (ev.input_shapes or []) if s])[:70]
        groups[(ev.name, shapes, where[-110:])] += 1
    for (name, shapes, where), c in sorted(groups.items(), key=lambda t: (-t[1], t[0])):
        print(f"{c / K:6.1f}/step  {name:16s} {shapes:70s} {where}")
