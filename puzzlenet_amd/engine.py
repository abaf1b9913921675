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
import torch

from . import _lib
from . import distributed as pdist


class TrainStep:
    def __init__(self, model, batch, lr, world=1, use_graph=False, warmup=0):
        if use_graph:
            raise _lib.PznError("the HIP-graph step was removed (slower than eager and unsafe with torch.topk's memset "
                                "nodes on this runtime): see puzzlenet_amd/engine.py")
        self.model = model
        self.batch = batch
        self.world = world
        self.grads = pdist.FlatGradAllReduce(model.parameters())
        # Adam + StepLR(50, 0.999) over flat buffers: one launch per step (distributed.FlatAdam; model5_b.py:1453-1457)
        self.opt = pdist.FlatAdam(self.grads, lr, sched_step=50, sched_gamma=0.999)
        self.loss = None
        self._saved_defer = getattr(model, "defer_emd_loss", False)
        model.defer_emd_loss = True

    def _fwd_bwd(self):
        self.grads.zero_()
        out = self.model.training_step(self.batch, 0)
        cur = torch.cuda.current_stream()
        if "loss" in out:
            loss = out["loss"]
            loss.backward()
        else:
            # the N x N EMD term is still running on the side stream: both parts are backward roots, so the backward of
            # the boundary terms and heads starts without waiting for it
            terms = list(out["loss_terms"])
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

    def step(self):
        self.loss = self._fwd_bwd()
        if self.world > 1:          # (an extra single-rank runner inside a multi-rank job must not join collectives)
            self.grads.all_reduce_mean()
        self.opt.step()
        return self.loss

    def close(self):
        """Give the model back as it was handed in (a later direct model.training_step() returns {'loss': ...} again)."""
        self.model.defer_emd_loss = self._saved_defer

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
