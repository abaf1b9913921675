"""CPU, world_size 2, gloo: the data-parallel plumbing of puzzlenet_amd.distributed / engine
(flat gradient bucket, one all-reduce per step, parameter broadcast, per-rank sharding) — the
N>1 path of bench.py minus the GPU kernels (which have no CPU fallback by design)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class TinyPairModel(nn.Module):
    """Stand-in with the same training_step contract as TouchedRegraster, pure torch, no BatchNorm
    (BatchNorm statistics are rank-local by design, see puzzlenet_amd/distributed.py)."""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(3, 16)
        self.b = nn.Linear(16, 6)
        self.unused = nn.Linear(4, 4)            # like fpc_decoder / rpc_decoder / dt: never gets a gradient

    def training_step(self, batch, _):
        x, y = batch
        out = self.b(torch.relu(self.a(x))).mean(dim=1)
        return {"loss": ((out - y) ** 2).mean()}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from puzzlenet_amd import distributed as pdist
    r, w, l = pdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                 # different init per rank on purpose ...
    model = TinyPairModel()
    pdist.broadcast_parameters(model)             # ... rank 0's parameters win
    ref = [torch.empty_like(p) for p in model.parameters()]
    for t, p in zip(ref, model.parameters()):
        t.copy_(p.data)
        dist.broadcast(t, 0)
        assert torch.equal(t, p.data)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 10, 3, generator=g), torch.randn(8, 6, generator=g)      # global batch of 8 "pairs"
    per = 8 // world
    shard = (X[rank * per:(rank + 1) * per], Y[rank * per:(rank + 1) * per])         # rank r gets pairs [r*B/W, (r+1)*B/W)
    grads = pdist.FlatGradAllReduce(model.parameters())
    assert grads.flat.numel() == sum((p.numel() + 3) // 4 * 4 for p in model.parameters())   # 16-byte aligned slots
    assert all(p.grad.data_ptr() >= grads.flat.data_ptr() for p in model.parameters())
    for it in range(2):                           # twice: zero_() must really reset the bucket
        grads.zero_()
        model.training_step(shard, 0)["loss"].backward()
        assert all(p.grad.data_ptr() >= grads.flat.data_ptr() for p in model.parameters())   # still views of the bucket
        grads.all_reduce_mean()
    torch.save({"flat": grads.flat.clone(), "state": model.state_dict()}, os.path.join(tmp, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["flat"], r1["flat"])                    # every rank holds the same averaged gradient
    # single process on the whole batch: mean-of-shard-means == full-batch mean for equal shards
    sys.path.insert(0, ROOT)
    from puzzlenet_amd import distributed as pdist
    model = TinyPairModel()
    model.load_state_dict(r0["state"])
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 10, 3, generator=g), torch.randn(8, 6, generator=g)
    grads = pdist.FlatGradAllReduce(model.parameters())
    grads.zero_()
    model.training_step((X, Y), 0)["loss"].backward()
    torch.testing.assert_close(r0["flat"], grads.flat, rtol=1e-5, atol=1e-7)
    n_unused = sum(p.numel() for p in model.unused.parameters())
    assert float(r0["flat"][-n_unused:].abs().max()) == 0.0       # never-used parameters stay at zero gradient


def test_init_from_env_single_process(monkeypatch):
    sys.path.insert(0, ROOT)
    from puzzlenet_amd import distributed as pdist
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert pdist.init_from_env() == (0, 1, 0)
    assert not dist.is_initialized()
    m = TinyPairModel()
    pdist.broadcast_parameters(m)                                  # no-op without a process group
    fg = pdist.FlatGradAllReduce(m.parameters())
    fg.all_reduce_mean()                                           # likewise


def _worker_two_piece(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from puzzlenet_amd import distributed as pdist
    pdist.init_from_env(backend="gloo")
    torch.manual_seed(5)
    out = {}
    for mode in ("one", "two"):
        torch.manual_seed(5)
        model = TinyPairModel()
        # "b." plays the layers whose gradient is complete last: laid out behind the others
        grads = pdist.FlatGradAllReduce(model.named_parameters(), late=(lambda n: n.startswith("b.")) if mode == "two" else None)
        if mode == "two":
            n_late = sum((p.numel() + 3) // 4 * 4 for n, p in model.named_parameters() if n.startswith("b."))
            assert grads.split == grads.flat.numel() - n_late
            assert [id(p) for p in grads.params[-2:]] == [id(model.b.weight), id(model.b.bias)]
        g = torch.Generator().manual_seed(7 + rank)
        X, Y = torch.randn(4, 10, 3, generator=g), torch.randn(4, 6, generator=g)
        for it in range(2):                       # twice: zero_() must re-arm the early piece
            grads.zero_()
            model.training_step((X, Y), 0)["loss"].backward()
            if mode == "two":
                grads.all_reduce_early()          # [0, split) while "the rest of the backward" would still run
                assert grads._early_done
            grads.all_reduce_mean()               # the late piece (mode two) or everything (mode one)
        out[mode] = {n: p.grad.clone() for n, p in model.named_parameters()}
    for n in out["one"]:
        assert torch.equal(out["one"][n], out["two"][n]), n       # same sums, element for element
    torch.save(out["two"], os.path.join(tmp, f"two{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_piece_allreduce_equals_single_bucket(tmp_path):
    """FlatGradAllReduce(late=...): the early piece reduced ahead of the late one gives, element for element, what one
    all-reduce of the whole bucket gives (gloo, world 2), also on the second step."""
    world, port = 2, _free_port()
    mp.spawn(_worker_two_piece, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = torch.load(tmp_path / "two0.pt"), torch.load(tmp_path / "two1.pt")
    for n in a:
        assert torch.equal(a[n], b[n])


def test_late_gradient_predicate_matches_model_parameters():
    """engine.late_gradient picks exactly the encoders' per-point / set-abstraction layers of TouchedRegraster."""
    sys.path.insert(0, ROOT)
    from puzzlenet_amd import engine
    assert engine.late_gradient("Encoder.mlp3.weight") and engine.late_gradient("Encoder2.bn1.bias")
    for n in ("Encoder.atten1.mlpq.weight", "Encoder2.out.bias", "tfMLP.0.weight", "MLPRpcb.2.bias", "dt"):
        assert not engine.late_gradient(n)
