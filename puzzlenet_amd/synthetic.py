"""Synthetic pairs with the dataset's 8-tuple contract (dataset.py:97-105):
(fpc, mrpc, igt, rpc, fpcb, rpcb, fpc_idx, rpc_idx), standing in for the LFS-only
datasets.  Built on the GPU with the product's own ops.

  fpc, rpc ~ U[0,1)^3                      (data are normalised to the unit cube, README.md:43)
  twist x = randn(6); x <- 0.8 x / |x|     (se_math/transforms.py:161-168)
  igt = exp(x);  mrpc = igt . rpc          (transforms.py:178-187)
  fpcb / rpcb / fpc_idx / rpc_idx = the 128 mutually nearest points by chamfer distance
                                           (dataset.py:1357-1367 get_boundary)
"""
import torch

from . import ops, se3


def make_batch(B, N, device, seed=1234, n_boundary=128):
    g = torch.Generator().manual_seed(int(seed))
    fpc = torch.rand(B, N, 3, generator=g).to(device)
    rpc = torch.rand(B, N, 3, generator=g).to(device)
    x = torch.randn(B, 6, generator=g)
    x = (0.8 * x / x.norm(dim=1, keepdim=True)).to(device)
    igt = se3.exp(x)
    mrpc = se3.transform(igt, rpc.permute(0, 2, 1)).permute(0, 2, 1).contiguous()
    cd1, cd2 = ops.chamfer(fpc, rpc)                 # cd1: per rpc point, cd2: per fpc point
    r_top = torch.topk(-cd1, n_boundary, dim=1)[1]
    f_top = torch.topk(-cd2, n_boundary, dim=1)[1]
    rpcb = ops.index_points(rpc, r_top)
    fpcb = ops.index_points(fpc, f_top)
    fpc_idx = torch.zeros(B, N, device=device).scatter_(1, f_top, 1.0)
    rpc_idx = torch.zeros(B, N, device=device).scatter_(1, r_top, 1.0)
    return [fpc, mrpc, igt, rpc, fpcb, rpcb, fpc_idx, rpc_idx]
