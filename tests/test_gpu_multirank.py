"""GPU: the N > 1 launch path of bench.py end to end — two ranks started by torch.distributed.run exactly as the
driver does, sharing the one GPU of the test box over gloo (RCCL refuses two ranks on one device; PZN_DIST_BACKEND
only swaps the transport).  Guards against rank-0-only work joining collectives (the per-kernel pricing pass after
the timed loop once dead-locked every multi-GPU run)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_bench_completes():
    env = dict(os.environ, PZN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "4", "--points", "512", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["global_batch"] == 8 and "cpu_baseline" not in d
    assert d["roofline"]["achieved"] > 0                      # rank 0's single-rank pricing pass ran to the end


def _run_two_rank_step(tmp_path, backend):
    """tests/_two_rank_step.py on two ranks; backend None = what distributed.init_from_env picks on its own (RCCL)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PZN_DIST_BACKEND", None)
    if backend:
        env["PZN_DIST_BACKEND"] = backend
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_two_rank_step.py"), str(tmp_path), "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


def _check_two_rank_records(tmp_path):
    import torch
    recs = [torch.load(os.path.join(tmp_path, f"rank{k}.pt")) for k in (0, 1)]
    split, n = recs[0]["split"], recs[0]["n"]
    assert 0 < split < n and recs[1]["split"] == split

    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))

    for s in range(2):
        assert recs[0]["early"][s] and recs[1]["early"][s], "the early piece was not reduced early"
        assert recs[0]["armed"][s] == (4, 4) and recs[1]["armed"][s] == (4, 4), recs[0]["armed"]
        assert torch.equal(recs[0]["reduced"][s], recs[1]["reduced"][s]), "the ranks hold different buckets"
        want = (recs[0]["local"][s] + recs[1]["local"][s]) / 2
        got = recs[0]["reduced"][s]
        assert rel(recs[0]["local"][s], recs[1]["local"][s]) > 1e-2          # (the shards do differ)
        for name, sl in (("early", slice(0, split)), ("late", slice(split, n))):
            assert rel(got[sl], want[sl]) < 2e-3, (s, name, rel(got[sl], want[sl]))
            # the kernels are deterministic and the mean is the same two fp32 operations either way: in practice the
            # pieces agree to rounding of the few atomically accumulated entries
            assert rel(got[sl], want[sl]) < 1e-5, (s, name, rel(got[sl], want[sl]))
    return recs


def test_overlapped_all_reduce_on_the_real_model(tmp_path):
    """engine.TrainStep(world=2) on TouchedRegraster (two encoder streams + the communication stream): after
    all_reduce_mean() both ranks hold the same bucket, and it is the mean of the two ranks' own gradients on their shards
    (BatchNorm is rank-local, so this is the exact expectation) — early piece and late piece separately, over two
    optimiser steps; the marker gate armed four tensors (two attention-chain inputs, two boundary-head inputs), all four
    fired, and the early piece did go out early.  Two ranks over gloo on the ONE GPU of the test box."""
    _run_two_rank_step(tmp_path, "gloo")
    _check_two_rank_records(tmp_path)


def _device_count():
    import torch
    return torch.cuda.device_count()        # (does not initialise the GPU in this process: the ranks are children)


@pytest.mark.skipif(_device_count() < 2, reason="RCCL needs one device per rank: this box has a single GPU")
def test_overlapped_all_reduce_over_rccl(tmp_path):
    """The same value check over RCCL (backend nccl, no PZN_DIST_BACKEND), one device per rank — the first contact of
    distributed.FlatGradAllReduce's three streams with the real transport is a test, not a bench.  Arms itself on any
    box with two or more GPUs (the ranks are child processes started before this process touches a GPU); skipped on the
    single-GPU test boxes."""
    _run_two_rank_step(tmp_path, None)
    recs = _check_two_rank_records(tmp_path)
    assert recs[0]["backend"] == "nccl" and recs[0]["device"] != recs[1]["device"], (recs[0].get("backend"), recs[0].get("device"))
