"""Dense pieces of the encoder / heads / loss tail: the functional forms the model calls.

All run as hand-written gfx950 kernels behind the C ABI (include/pzn.h); results are fp32, the matrix-core operand path
is bf16x3 split precision by default (exact-fp32 MFMA on request, pzn_gemm_set_precision):
  linear          nn.Linear (+bias +ReLU) fwd / dgrad / wgrad       weight-stationary / direct-fragment / tile kernels
  shared_mlp_max  two shared-MLP layers + max over the K=32 axis    max-pool as a register epilogue
  attention       softmax(q k^T / sqrt(dk)) v, returns the map too  batched tile engine + wave-per-row softmax
  chamfer         min-both-ways of |a|^2+|b|^2-2ab without P[B,n,m] csrc/chamfer.hip
"""
from . import ops


def linear(x, weight, bias=None, relu=False):
    """nn.Linear (+ optional ReLU): y = act(x W^T + b)."""
    return ops.linear(x, weight, bias, relu)


def shared_mlp_max(x, w1, b1, w2, b2):
    """model5_b.py:452-454 / :459-461: relu(mlp_a) -> relu(mlp_b) -> max over the K axis.
    x [B,S,K=32,C0] -> [B,S,C2]."""
    return ops.shared_mlp_max(x, w1, b1, w2, b2)


def attention(q, k, v):
    """model5_b.py:67-75 scaled_dot_production -> (values, attention)."""
    return ops.attention(q, k, v)


def chamfer(a, b):
    """model5_b.py:1495-1505 chamfer_loss: fused kernel, P[B,n,m] never materialised."""
    return ops.chamfer(a, b)


def sa_mlp_max(xyz, feat, new_xyz, idx, w1, b1, w2, b2):
    """group (pointnet_util.py:123-132) + shared MLP + max over K (model5_b.py:452-454 / :459-461) on the
    model-internal padded layout; the [B,S,K,3+D] tensor of the drop-in sample_and_group is not built."""
    return ops.sa_mlp_max(xyz, feat, new_xyz, idx, w1, b1, w2, b2)
