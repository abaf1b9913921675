"""Helper of tests/test_gpu_multirank.py::test_overlapped_all_reduce_on_the_real_model — one rank of a two-rank job
started by torch.distributed.run (gloo, both ranks on the one GPU of the test box).

Per step and rank: (a) the rank's own gradient with the early all-reduce switched off (plain forward + backward into the
flat bucket), (b) the same step through engine.TrainStep(world=2): marker hooks -> early piece on the communication
stream while the set-abstraction backward runs -> late piece -> Adam.  Dumps both buckets per step for the parent."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    out_dir, steps = sys.argv[1], int(sys.argv[2])
    from puzzlenet_amd import distributed as pdist
    rank, world, _ = pdist.init_from_env()                    # before any GPU call of this process
    assert world == 2
    from oracle import model_ref as mr                         # (parameter fill only: closed-form weights, same on both ranks)
    from puzzlenet_amd import engine, model5_b, synthetic
    dev = torch.device("cuda", torch.cuda.current_device())
    B, N = 2, 512
    cfg = mr.Cfg(num_points=N, loss_mode=1)
    model = model5_b.TouchedRegraster(cfg)
    mr.fill_params(model)
    model.to(dev)
    batch = synthetic.make_batch(B, N, dev, seed=1234 + rank)   # the rank's shard
    runner = engine.TrainStep(model, batch, cfg.lr, world=world, prefetch=False)
    rec = {"split": runner.grads.split, "n": runner.grads.flat.numel(), "local": [], "reduced": [], "early": [], "armed": []}
    for s in range(steps):
        # (a) local gradient, no collective: the gate is armed but opens into nothing
        gate_open, runner._gate.on_open = runner._gate.on_open, (lambda events: None)
        torch.manual_seed(100 + s)                              # FPS start indices (pointnet_util.py:65): same draw in (a) and (b)
        runner._fwd_bwd()
        torch.cuda.synchronize()
        rec["local"].append(runner.grads.flat.detach().cpu().clone())
        runner._gate.on_open = gate_open
        # (b) the real step
        torch.manual_seed(100 + s)
        runner.step()
        torch.cuda.synchronize()
        rec["reduced"].append(runner.grads.flat.detach().cpu().clone())
        rec["early"].append(bool(runner.grads._early_done))
        rec["armed"].append((runner._gate.armed, runner._gate.fired))
    runner.close()
    import torch.distributed as dist
    rec["backend"], rec["device"] = dist.get_backend(), torch.cuda.current_device()
    torch.save(rec, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
