"""Data parallelism for the pair batch: one process per GPU, RCCL over xGMI.

The reference has no distributed code at all.  Pairs are
independent, so each rank runs the whole step on its own 64 pairs and the only
exchange is the gradient all-reduce: 8,059,220 fp32 = 32.24 MB in ONE flat buffer,
reduced in two pieces: the "early" piece (heads, attention blocks, out projections:
31 MB, complete once the attention chains' backward has been enqueued) goes out on a
communication stream while the set-abstraction backward still runs; the "late" piece
(the encoders' per-point and shared MLPs: 1 MB) follows after the last backward node.
BatchNorm statistics stay rank-local, exactly as a per-GPU run of the reference would
behave (model5_b.py:424,447: the BN channel axis is N).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*.
    Returns (rank, world, local_rank). world == 1 -> no process group.  With a GPU the rank's device is selected
    BEFORE the process group exists and handed to it (`device_id`), so the RCCL communicator is bound to that device
    eagerly instead of to whatever device is current at the first collective."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = torch.cuda.is_available()
    if on_gpu:
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))   # (rehearsals with more ranks than GPUs share one)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            raise RuntimeError("WORLD_SIZE > 1 needs MASTER_PORT (torch.distributed.run sets it)")
        if backend is None:
            # "nccl" IS RCCL on ROCm.  PZN_DIST_BACKEND=gloo rehearses the multi-rank path where RCCL cannot run
            # (several ranks sharing one GPU, CPU-only hosts).
            backend = os.environ.get("PZN_DIST_BACKEND") or ("nccl" if on_gpu else "gloo")
        kw = {}
        if backend == "nccl" and on_gpu:
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        if os.environ.get("PZN_DIST_LAZY_INIT") == "1":
            kw = {}              # every rank takes the same path: the switch is an environment variable, not an exception
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        except TypeError:
            # this torch build does not know the `device_id` keyword: the lazy communicator binds to the current device
            # (selected above) at the first collective.  Only a TypeError - raised before any rendezvous, on every rank
            # alike - is retried: a RuntimeError from a half-built group on ONE rank must not lead that rank into a second
            # rendezvous on the same port while its peers are past the first (set PZN_DIST_LAZY_INIT=1 on all ranks instead).
            if not kw:
                raise
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        except RuntimeError as e:
            if kw:      # the eager (device-bound) RCCL communicator failed: say which switch selects the lazy one
                raise RuntimeError(f"{e}  [puzzlenet_amd: the process group was created with device_id={kw['device_id']} "
                                   "(eager RCCL communicator); set PZN_DIST_LAZY_INIT=1 on ALL ranks to create it lazily at "
                                   "the first collective instead]") from e
            raise
    return rank, world, local


class FlatGradAllReduce:
    """All parameters' gradients live in one contiguous buffer (p.grad are views of it),
    so the per-step exchange is a single collective on 32 MB and optimizer / zero_grad
    touch one tensor."""

    def __init__(self, params, late=None):
        """params: iterable of parameters, or of (name, parameter) pairs when `late` is given.
        late(name) -> True for parameters whose gradient is complete only at the very end of the backward: they are
        laid out behind all others, so that [0, split) can be reduced early (all_reduce_early) and [split, n) late."""
        items = list(params)
        if items and isinstance(items[0], tuple):
            named = [(n, p) for n, p in items if p.requires_grad]
        else:
            named = [("", p) for p in items if p.requires_grad]
        if late is not None:
            named = [np_ for np_ in named if not late(np_[0])] + [np_ for np_ in named if late(np_[0])]
        n_early = sum(1 for n_, _ in named if late is None or not late(n_))
        self.params = [p for _, p in named]
        # every tensor starts on a 16-byte boundary of the bucket (kernels read weights / biases with 16-byte loads)
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.split = self.offsets[n_early] if n_early < len(self.params) else n     # first element of the late piece
        self._early_done = False
        self._comm = None
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        for p, off in zip(self.params, self.offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)
        if self.flat.is_cuda:
            # let the weight-gradient kernels add straight into the bucket (ops._GRAD_SINKS)
            from . import ops
            ops.clear_grad_sinks()
            ops.register_grad_sinks(self.params)

    def zero_(self):
        self.flat.zero_()
        self._early_done = False

    def _active(self):
        return dist.is_initialized() and dist.get_world_size() > 1

    def _reduce(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.div_(dist.get_world_size())

    def all_reduce_early(self, events=()):
        """Reduce [0, split) now.  On the GPU the collective is enqueued on a communication stream behind `events`
        (recorded where the producers of these gradients were enqueued), so it overlaps with the rest of the backward;
        all_reduce_mean() later joins it."""
        if not self._active() or self.split == 0 or self._early_done:
            return
        early = self.flat[:self.split]
        if self.flat.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=self.flat.device)
            for ev in events:
                self._comm.wait_event(ev)
            with torch.cuda.stream(self._comm):
                self._reduce(early)
        else:
            self._reduce(early)
        self._early_done = True

    def all_reduce_mean(self):
        """After the whole backward: reduce what all_reduce_early did not, and order the caller's stream behind both."""
        if not self._active():
            return
        if self._early_done:
            if self.split < self.flat.numel():
                self._reduce(self.flat[self.split:])
            if self._comm is not None:
                torch.cuda.current_stream().wait_stream(self._comm)
        else:
            self._reduce(self.flat)


class FlatAdam:
    """torch.optim.Adam(lr) + StepLR(step_size, gamma) (model5_b.py:1453-1457) over flat buffers: the parameters
    are re-pointed into one contiguous buffer (like their gradients in FlatGradAllReduce), the two moments are
    flat too, and the whole update is ONE launch of pzn_adam_step_f32.  GPU only; no CPU fallback."""

    def __init__(self, grads, lr, betas=(0.9, 0.999), eps=1e-8, sched_step=50, sched_gamma=0.999):
        from . import _lib, ops
        if not grads.flat.is_cuda:
            raise _lib.PznError("FlatAdam needs parameters on the GPU (puzzlenet_amd has no CPU fallback)")
        self.grads = grads
        self.flat = torch.zeros_like(grads.flat)         # same layout as the gradient bucket (16-byte aligned tensors)
        with torch.no_grad():
            for p, off in zip(grads.params, grads.offsets):
                n = p.numel()
                self.flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + n].view_as(p)
        # the parameters moved: the weight-gradient sinks are keyed by parameter address
        ops.clear_grad_sinks()
        ops.register_grad_sinks(grads.params)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.lr0, self.betas, self.eps = float(lr), betas, float(eps)
        self.sched_step, self.sched_gamma = int(sched_step), float(sched_gamma)
        self.t = 0

    @property
    def lr(self):
        """StepLR stepped once per batch: lr0 * gamma ** (t // step_size) for the (t+1)-th update."""
        return self.lr0 * self.sched_gamma ** (self.t // self.sched_step)

    def step(self):
        from . import _lib
        lr = self.lr
        self.t += 1
        with torch.cuda.device(self.flat.device):
            _lib.call("pzn_adam_step_f32", self.flat.data_ptr(), self.grads.flat.data_ptr(), self.exp_avg.data_ptr(),
                      self.exp_avg_sq.data_ptr(), self.flat.numel(), lr, self.betas[0], self.betas[1], self.eps, self.t,
                      torch.cuda.current_stream().cuda_stream)


def broadcast_parameters(module, src=0):
    """Rank `src`'s parameters and buffers to everyone (what DDP does at construction)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)
