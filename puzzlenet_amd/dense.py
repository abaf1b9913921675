"""Dense pieces of the encoder / heads / loss tail.

linear / shared_mlp_max / attention / chamfer are the functional forms the
model calls.  STATUS (round 1, first end-to-end slice): these four are still
composed from torch GPU ops (rocBLAS GEMMs, elementwise kernels); the
hand-written MFMA kernels replace them one by one behind the same signatures.
"""
import math

import torch
import torch.nn.functional as F


def linear(x, weight, bias=None, relu=False):
    """nn.Linear (+ optional ReLU): y = act(x W^T + b)."""
    y = F.linear(x, weight, bias)
    return F.relu(y) if relu else y


def shared_mlp_max(x, w1, b1, w2, b2):
    """model5_b.py:452-454 / :459-461: relu(mlp_a) -> relu(mlp_b) -> max over the K axis.
    x [B,S,K,C0] -> [B,S,C2]."""
    h = F.relu(F.linear(x, w1, b1))
    y = F.relu(F.linear(h, w2, b2))
    return torch.max(y, dim=-2)[0]


def attention(q, k, v):
    """model5_b.py:67-75 scaled_dot_production -> (values, attention)."""
    dk = q.size()[-1]
    logits = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
    attn = F.softmax(logits, dim=-1)
    return torch.matmul(attn, v), attn


def chamfer(a, b):
    """model5_b.py:1495-1505 chamfer_loss (expansion form, P materialised)."""
    x, y = a, b
    bs, numpoints, pc_dim = x.size()
    xx = torch.bmm(x, x.transpose(2, 1))
    yy = torch.bmm(y, y.transpose(2, 1))
    zz = torch.bmm(x, y.transpose(2, 1))
    diag_ind = torch.arange(0, numpoints, device=x.device)
    rx = xx[:, diag_ind, diag_ind].unsqueeze(1).expand_as(xx)
    ry = yy[:, diag_ind, diag_ind].unsqueeze(1).expand_as(yy)
    P = (rx.transpose(2, 1) + ry - 2 * zz)
    return torch.min(P, 1)[0], torch.min(P, 2)[0]
