"""Dump per-parameter gradients of one training step (B=8, N=2048) to an .npz — for A/B comparisons of kernel paths
selected by environment variables (PZN_WS_GEMM, PZN_DF_GEMM, PZN_GEMM_PRECISION ...)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Cfg
from puzzlenet_amd import model5_b, synthetic, distributed as pdist

out = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
cfg = Cfg(); cfg.num_points = 2048
torch.manual_seed(0)
model = model5_b.TouchedRegraster(cfg).to(dev)
model.two_streams = False
batch = synthetic.make_batch(B, 2048, dev, seed=1234)
grads = pdist.FlatGradAllReduce(model.parameters()) if os.environ.get("SINKS", "1") == "1" else None
torch.manual_seed(1000)
if grads is not None:
    grads.zero_()
loss = model.training_step(batch, 0)["loss"]
loss.backward()
torch.cuda.synchronize()
d = {"loss": np.array(float(loss))}
for n, p in model.named_parameters():
    d[n] = p.grad.detach().float().cpu().numpy()
np.savez(out, **d)
print("loss", float(loss))
