#!/usr/bin/env python3
"""Is the training step bound by the host?  Time to ENQUEUE K steps (host returns from the last step()) against the time
until the device has finished them, and the host time of a step when the device has nothing to wait for.

    python tools/host_enqueue_time.py [steps]

enqueue ~ total: the host is the bottleneck (the device drains as soon as the last launch arrives); enqueue << total: the
device is, and the host runs ahead (unless a step synchronises inside)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from puzzlenet_amd import _lib, engine, model5_b, synthetic  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    cfg = bench.Cfg()
    torch.manual_seed(0)
    model = model5_b.TouchedRegraster(cfg).to(dev)
    batch = synthetic.make_batch(64, cfg.num_points, dev, seed=1234)
    torch.manual_seed(1000)
    runner = engine.TrainStep(model, batch, cfg.lr, world=1)
    for _ in range(5):
        runner.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per = []
    for _ in range(K):
        a = time.perf_counter()
        runner.step()
        per.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    per.sort()
    print(f"steps {K}: enqueue {1e3 * (t1 - t0) / K:.2f} ms/step, until the device is done {1e3 * (t2 - t0) / K:.2f} ms/step, "
          f"device tail after the last enqueue {1e3 * (t2 - t1):.2f} ms")
    print(f"host time of one step(): min {1e3 * per[0]:.2f}  median {1e3 * per[K // 2]:.2f}  max {1e3 * per[-1]:.2f} ms")
    # the same with a device that is never the bottleneck: synchronise before every step, time only the host part
    host = []
    for _ in range(10):
        torch.cuda.synchronize()
        a = time.perf_counter()
        runner.step()
        host.append(time.perf_counter() - a)
    host.sort()
    print(f"host time of step() on an idle device: min {1e3 * host[0]:.2f}  median {1e3 * host[5]:.2f} ms")
    runner.close()


if __name__ == "__main__":
    main()
