"""Training-step runner: predict5 + losses + backward + gradient all-reduce (N > 1) + Adam, eager.

The step is enqueued without a single host synchronisation (pinned asynchronous upload of the FPS start indices,
no `.item()`), so the host runs several steps ahead of the GPU and launch cost is hidden behind the kernels.  Two HIP
streams carry the two encoders (model5_b.TouchedRegraster.two_streams) and the N x N EMD; its loss term stays a separate
backward root (`defer_emd_loss`) so that the backward of the boundary terms does not queue behind the join.

A whole-step HIP graph was tried in round 1 and removed in round 2: it was slower than this path (15.2 vs 13.0 ms — the
fork / separate-root schedule exists only in eager mode) and it aborted once with an out-of-range index inside
torch's scatter kernel during replay.  The index came from `torch.topk(., 128)`, whose multi-block radix select zeroes
its counters with `hipMemsetAsync`; captured as memset NODES those ran out of stream order on this ROCm runtime
(tools/hip_graph_memset_repro.py shows the same defect in isolation), so the select read stale counters and produced
garbage indices.  DESIGN.md §7 keeps the record.
"""
import os

import torch

from . import _lib
from . import distributed as pdist


_LATE_LAYERS = ("mlp1.", "mlp2.", "mlp3.", "mlp4.", "mlp5.", "mlp6.", "bn1.", "bn2.")


def late_gradient(name):
    """Parameters of TouchedRegraster whose gradient is produced by the last nodes of the backward: the encoders'
    per-point MLP + BatchNorm (model5_b.py:447-448) and set-abstraction MLPs (:452-461)."""
    return name.startswith(("Encoder.", "Encoder2.")) and name.split(".", 1)[1].startswith(_LATE_LAYERS)


class _EarlyGate:
    """When may the early piece of the gradient bucket go out?  When the backward of EVERYTHING that writes into it has
    been enqueued.  The model marks tensors during the forward pass (`mark`): the input of each encoder's attention
    chain (behind it: the chain, the out projection and, through the global feature, the pose head) and an alias of
    each boundary head's per-point input (behind it: MLPLocalPre* and MLP*b, which hang off the per-point features and
    are NOT ancestors of the chain inputs).  Each marker's tensor hook fires when that tensor's gradient exists, i.e.
    after the backward nodes behind it have run (autograd's data dependency, not its queue order); the hook records an
    event on the stream it runs on.  When every marker armed in this step's forward has fired, `on_open(events)` is
    called once."""

    def __init__(self, on_open):
        self.on_open = on_open
        self.reset()

    def reset(self):
        self.armed, self.fired, self.events, self.opened = 0, 0, [], False

    def mark(self, t):
        self.armed += 1
        t.register_hook(self._fire)
        return t

    def _fire(self, grad):
        self.fired += 1
        if grad.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.events.append(ev)
        if self.fired == self.armed and not self.opened:
            self.opened = True
            self.on_open(self.events)
        return None


class TrainStep:
    def __init__(self, model, batch, lr, world=1, use_graph=False, warmup=0, prefetch=None, fps_generator=None):
        if use_graph:
            raise _lib.PznError("the HIP-graph step was removed (slower than eager and unsafe with torch.topk's memset "
                                "nodes on this runtime): see puzzlenet_amd/engine.py")
        self.model = model
        self.batch = batch
        self.world = world
        # "late" gradients = the encoders' per-point and set-abstraction layers (last nodes of the backward); everything
        # else is complete once both attention chains' AND both boundary heads' backward has been enqueued (_EarlyGate)
        # and is reduced early (N > 1)
        self.grads = pdist.FlatGradAllReduce(model.named_parameters(), late=late_gradient)
        self._gate = _EarlyGate(self.grads.all_reduce_early)
        self._saved_markers = None
        if world > 1 and hasattr(model, "Encoder") and hasattr(model, "Encoder2"):
            holders = (model, model.Encoder, model.Encoder2)
            self._saved_markers = [(h, getattr(h, "grad_marker", None)) for h in holders]
            for h in holders:
                h.grad_marker = self._gate.mark
        # Adam + StepLR(50, 0.999) over flat buffers: one launch per step (distributed.FlatAdam; model5_b.py:1453-1457)
        self.opt = pdist.FlatAdam(self.grads, lr, sched_step=50, sched_gamma=0.999)
        self.loss = None
        self.prefetch = os.environ.get("PZN_PLAN_PREFETCH", "1") != "0" if prefetch is None else bool(prefetch)
        self._plans_ahead = None
        self._saved_defer = getattr(model, "defer_emd_loss", False)
        model.defer_emd_loss = True
        # FPS start indices: the reference draws them from torch's global CPU generator (pointnet_util.py:65) and so does
        # this runner by default; the plan prefetch makes step k+1's draws before step k runs (and one unused set after
        # the last step unless step(last=True) says so), which any other consumer of the global generator between steps
        # would see as a shifted stream: hand in a private generator to keep the two apart
        self._saved_fps_generator = getattr(model, "fps_generator", None)
        if fps_generator is not None:
            model.fps_generator = fps_generator
        # one device, one stream of Python backward functions: autograd's per-device worker thread only adds a hand-off per
        # backward() and GIL traffic (host time of a step 5.3 -> 5.1 ms, tools/host_enqueue_time.py).  Scoped to the backward
        # calls of a step (_fwd_bwd): nothing process-wide is left behind by a runner that is dropped or raises.
        self._autograd_mt = False

    def _fwd_bwd(self):
        self.grads.zero_()
        self._gate.reset()
        out = self.model.training_step(self.batch, 0)
        cur = torch.cuda.current_stream()
        if "loss" in out:
            loss = out["loss"]
            with torch.autograd.set_multithreading_enabled(self._autograd_mt):
                loss.backward()
        else:
            # the N x N EMD term is still running on the side stream: both parts are backward roots, so the backward of
            # the boundary terms and heads starts without waiting for it
            terms = list(out["loss_terms"])
            with torch.autograd.set_multithreading_enabled(self._autograd_mt):
                torch.autograd.backward(terms)
            if out.get("join_stream") is not None:
                cur.wait_stream(out["join_stream"])
            loss = terms[0]
            for t in terms[1:]:
                loss = loss + t
            loss = loss.detach()
        # Encoder2's backward nodes ran on the side stream and, with gradient sinks, wrote the flat bucket from there
        # without an AccumulateGrad node on this stream: order the all-reduce / Adam after them explicitly.
        side = getattr(self.model, "_side_stream", None)
        if side is not None:
            cur.wait_stream(side)
        return loss

    def step(self, next_batch=None, last=False):
        """One training step on self.batch.  next_batch: the batch of the FOLLOWING step when it differs (a data loader);
        this runner then moves on to it.  last: no step follows (nothing is prefetched, no start indices are drawn for
        it).  With `prefetch` the coordinate-only part of the following step (FPS, centroid
        gathers, neighbour searches: model.prefetch_plans) is enqueued first, on a background stream, and runs beside
        this step's kernels instead of standing at the head of the next step; this step uses the plan the previous one
        left.  Work per step is unchanged (one sampling pass per step), only its place in the queue."""
        model = self.model
        self._adopt(self.batch)
        if next_batch is not None:
            self._adopt(next_batch, plans_only=True)
        if self.prefetch and hasattr(model, "prefetch_plans"):
            if self._plans_ahead is None:      # first step: its own plan first (the start indices keep their order of draw)
                self._plans_ahead = model.prefetch_plans(self.batch[0], self.batch[1])
            model.use_plans(self._plans_ahead)
            nb = self.batch if next_batch is None else next_batch
            self._plans_ahead = None if last else model.prefetch_plans(nb[0], nb[1])
        self.loss = self._fwd_bwd()
        if self.world > 1:          # (an extra single-rank runner inside a multi-rank job must not join collectives)
            self.grads.all_reduce_mean()
        self.opt.step()
        if next_batch is not None:
            self.batch = next_batch
        return self.loss

    def _adopt(self, batch, plans_only=False):
        """A batch produced on another stream (datapipe.PairBatch: `ready` event): its consumers' streams wait for it and the
        caching allocator is told who reads it.  plans_only: the batch of the FOLLOWING step, which only the plan-prefetch
        stream touches during this one."""
        ready = getattr(batch, "ready", None)
        if ready is None:
            return
        model = self.model
        ps = getattr(model, "_plan_stream", None)
        if plans_only:
            if ps is None and self.prefetch and hasattr(model, "prefetch_plans"):
                ps = model._plan_stream = torch.cuda.Stream()
            streams = [ps] if ps is not None else []
        else:
            if getattr(model, "two_streams", False) and hasattr(model, "side_stream"):
                model.side_stream()
            streams = [torch.cuda.current_stream()] + [s_ for s_ in (getattr(model, "_side_stream", None),) if s_ is not None]
            batch.ready = None          # (adopted: from here on ordinary stream order covers it)
        for s_ in streams:
            s_.wait_event(ready)
            for t in batch:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(s_)

    def close(self):
        """Give the model back as it was handed in (a later direct model.training_step() returns {'loss': ...} again)."""
        self.model.defer_emd_loss = self._saved_defer
        self.model.fps_generator = self._saved_fps_generator
        self._plans_ahead = None
        if hasattr(self.model, "use_plans"):
            self.model.use_plans(None)
        if self._saved_markers is not None:
            for h, m in self._saved_markers:
                h.grad_marker = m
            self._saved_markers = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
