"""Training-step runner: the whole step (predict5 + losses + backward [+ Adam]) as ONE HIP graph.

The reference leaves scheduling to PyTorch eager mode: ~1,000 kernel launches per step, each paid
for on the host.  On MI355X the step's kernels are short (tens of microseconds), so the host, not
the GPU, sets the pace.  Here the step is captured once into a HIP graph and replayed: launch cost
collapses to one graph launch, and the GPU runs the kernels back to back.

What makes the step capturable:
  * every op is enqueued on the current stream (the C ABI takes the stream explicitly);
  * no host<->device copies inside: FPS start indices come from StartIndexFeed buffers that are
    refilled from the CPU generator before each replay (same draws as the eager path);
  * gradients live in one flat buffer that is zeroed in place (FlatGradAllReduce);
  * Adam runs with capturable=True and a device-resident learning rate.
With more than one rank the gradient all-reduce (RCCL) and the optimizer run after the graph.
"""
import os

import torch

from . import distributed as pdist
from . import pointnet_util as pu


class TrainStep:
    def __init__(self, model, batch, lr, world=1, use_graph=True, warmup=3):
        self.model = model
        self.batch = batch
        self.world = world
        self.grads = pdist.FlatGradAllReduce(model.parameters())
        dev = self.grads.flat.device
        self.graph = None
        self.opt_in_graph = use_graph and world == 1
        lr_arg = torch.tensor(float(lr), device=dev) if use_graph else lr
        if use_graph:      # experimental path: torch's capturable Adam inside the graph
            self.opt = torch.optim.Adam(model.parameters(), lr=lr_arg, capturable=True)
            self.sched = torch.optim.lr_scheduler.StepLR(self.opt, 50, 0.999)   # model5_b.py:1453-1457
        else:              # Adam + StepLR(50, 0.999) over flat buffers: one launch per step (distributed.FlatAdam)
            self.opt = pdist.FlatAdam(self.grads, lr, sched_step=50, sched_gamma=0.999)
            self.sched = None
        self.loss = None
        self.feed = None
        model.defer_emd_loss = not use_graph
        if use_graph:
            if os.environ.get('PZN_GRAPH_STREAMS', '1') != '0' and getattr(model, 'two_streams', False):
                model.two_streams = 'graph'     # keep the encoder fork / join inside the captured graph
            self._capture(warmup)

    # -- one eager step (also the body that gets captured)
    def _fwd_bwd(self):
        self.grads.zero_()
        out = self.model.training_step(self.batch, 0)
        if "loss" in out:
            loss = out["loss"]
            loss.backward()
            return loss
        # the N x N EMD term is still running on the side stream: both parts are backward roots, so the backward of the
        # boundary terms and heads starts without waiting for it; the engine's end-of-backward sync joins the streams
        terms = list(out["loss_terms"])
        torch.autograd.backward(terms)
        if out.get("join_stream") is not None:
            torch.cuda.current_stream().wait_stream(out["join_stream"])
        loss = terms[0]
        for t in terms[1:]:
            loss = loss + t
        return loss.detach()

    def _capture(self, warmup):
        self.feed = pu.StartIndexFeed()
        pu.set_start_index_feed(self.feed)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                      # warm-up off the default stream
            for i in range(max(2, warmup)):
                if i == 0:
                    pass                                # first pass RECORDS the FPS call sequence (allocates the slots)
                else:
                    self.feed.refill()
                self._fwd_bwd()
                if i == 0:
                    self.feed.freeze()
                if self.world > 1:
                    self.grads.all_reduce_mean()
                self.opt.step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.feed.refill()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._fwd_bwd()
            if self.opt_in_graph:
                self.opt.step()

    def step(self):
        if self.graph is None:
            self.loss = self._fwd_bwd()
            if self.world > 1:          # (an extra single-rank runner inside a multi-rank job must not join collectives)
                self.grads.all_reduce_mean()
            self.opt.step()
        else:
            self.feed.refill()
            self.graph.replay()
            if not self.opt_in_graph:
                if self.world > 1:
                    self.grads.all_reduce_mean()
                self.opt.step()
        if self.sched is not None:
            self.sched.step()
        return self.loss

    def close(self):
        if self.feed is not None:
            pu.set_start_index_feed(None)
