"""Data parallelism for the pair batch: one process per GPU, RCCL over xGMI.

The reference has no distributed code at all.  Pairs are
independent, so each rank runs the whole step on its own 64 pairs and the only
exchange is the gradient all-reduce: 8,059,220 fp32 = 32.24 MB, ONE flat bucket,
one `all_reduce(SUM)` per step (ring time ~0.4 ms on 7x153 GB/s xGMI links vs a
step of tens of ms, so it is issued once after backward rather than bucketed and
overlapped).  BatchNorm statistics stay rank-local, exactly as a per-GPU run of
the reference would behave (model5_b.py:424,447: the BN channel axis is N).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*.
    Returns (rank, world, local_rank). world == 1 -> no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" IS RCCL on ROCm
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class FlatGradAllReduce:
    """All parameters' gradients live in one contiguous buffer (p.grad are views of it),
    so the per-step exchange is a single collective on 32 MB and optimizer / zero_grad
    touch one tensor."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        if self.flat.is_cuda:
            # let the weight-gradient kernels add straight into the bucket (ops._GRAD_SINKS)
            from . import ops
            ops.clear_grad_sinks()
            ops.register_grad_sinks(self.params)

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean(self):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())


def broadcast_parameters(module, src=0):
    """Rank `src`'s parameters and buffers to everyone (what DDP does at construction)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)
