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
