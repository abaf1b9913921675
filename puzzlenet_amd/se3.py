"""SE(3) exponential and point transform as used by the training step
(reference: se_math/se3.py:57-80 `exp`, :110-120 `transform`; se_math/so3.py `mat`;
se_math/sinc.py:6-18, 96-108, 126-138 `sinc1/2/3` with their |t| < 0.01 Taylor branches).

Row (f1) of the scope table.  `exp` of GPU tensors is one HIP launch each way (csrc/se3.hip): as tensor ops
it was ~150 launches of 64-element kernels per step (2.3 ms of a 21 ms step).  The tensor-op form below
serves host tensors (fixtures, synthetic data on the CPU) and documents the formulas.
"""
import torch


def _sinc1(t):
    """sin(t)/t  (sinc.py:6-18)"""
    t2 = t * t
    small = 1 - t2 / 6 * (1 - t2 / 20 * (1 - t2 / 42))
    safe = torch.where(t.abs() < 0.01, torch.ones_like(t), t)
    return torch.where(t.abs() < 0.01, small, torch.sin(safe) / safe)


def _sinc2(t):
    """(1 - cos t)/t^2  (sinc.py:96-108)"""
    t2 = t * t
    small = 1 / 2 * (1 - t2 / 12 * (1 - t2 / 30 * (1 - t2 / 56)))
    safe = torch.where(t.abs() < 0.01, torch.ones_like(t), t)
    return torch.where(t.abs() < 0.01, small, (1 - torch.cos(safe)) / (safe * safe))


def _sinc3(t):
    """(t - sin t)/t^3  (sinc.py:126-138)"""
    t2 = t * t
    small = 1 / 6 * (1 - t2 / 20 * (1 - t2 / 42 * (1 - t2 / 72)))
    safe = torch.where(t.abs() < 0.01, torch.ones_like(t), t)
    return torch.where(t.abs() < 0.01, small, (safe - torch.sin(safe)) / (safe ** 3))


def _so3_mat(w):
    """[*,3] -> skew matrices [*,3,3]  (so3.py `mat`)"""
    w1, w2, w3 = w[:, 0], w[:, 1], w[:, 2]
    O = torch.zeros_like(w1)
    return torch.stack((torch.stack((O, -w3, w2), dim=1),
                        torch.stack((w3, O, -w1), dim=1),
                        torch.stack((-w2, w1, O), dim=1)), dim=1)


def exp(x):
    """twist [*,6] = (w, v) -> SE(3) [*,4,4]  (se3.py:57-80)"""
    x_ = x.reshape(-1, 6)
    if x_.is_cuda and x_.dtype == torch.float32:      # one HIP launch each way (csrc/se3.hip) instead of ~150 tensor ops
        from . import ops
        return ops.se3_exp(x_).view(*(x.size()[0:-1]), 4, 4)
    w, v = x_[:, 0:3], x_[:, 3:6]
    t = w.norm(p=2, dim=1).view(-1, 1, 1)
    W = _so3_mat(w)
    S = W.bmm(W)
    I = torch.eye(3, dtype=x.dtype, device=x.device)
    R = I + _sinc1(t) * W + _sinc2(t) * S          # Rodrigues
    V = I + _sinc2(t) * W + _sinc3(t) * S
    p = V.bmm(v.contiguous().view(-1, 3, 1))
    z = torch.zeros(x_.size(0), 1, 4, dtype=x.dtype, device=x.device)     # built on the device: no host copy
    z[:, :, 3] = 1
    g = torch.cat((torch.cat((R, p), dim=2), z), dim=1)
    return g.view(*(x.size()[0:-1]), 4, 4)


def transform_points(g, pts):
    """g [B,4,4], pts [B,N,3] -> R pts + p, i.e. transform(g, pts^T)^T without the two transposes: one HIP launch each
    way on the GPU (csrc/losstail.hip), the tensor form of `transform` otherwise."""
    if pts.is_cuda and pts.dtype == torch.float32 and g.dim() == 3 and pts.dim() == 3 and g.shape[0] == pts.shape[0]:
        from . import ops
        return ops.se3_transform_points(g.to(pts), pts)
    return transform(g, pts.transpose(-1, -2)).transpose(-1, -2)


def transform(g, a):
    """g [*,4,4], a [*,3,N] (or [*,3]) -> R a + p  (se3.py:110-120)"""
    if a.is_cuda and a.dtype == torch.float32 and g.dim() == 3 and a.dim() == 3 and a.shape[1] == 3 \
            and g.shape[0] == a.shape[0] and a.transpose(1, 2).is_contiguous():     # (a broadcast g takes the tensor form)
        # the caller's a is a [B,N,3] point tensor seen as [B,3,N] (model5_b.py:948-952): same kernel, no copies
        from . import ops
        return ops.se3_transform_points(g.to(a), a.transpose(1, 2)).transpose(1, 2)
    g_ = g.view(-1, 4, 4)
    R = g_[:, 0:3, 0:3].contiguous().view(*(g.size()[0:-2]), 3, 3)
    p = g_[:, 0:3, 3].contiguous().view(*(g.size()[0:-2]), 3)
    if len(g.size()) == len(a.size()):
        return R.matmul(a) + p.unsqueeze(-1)
    return R.matmul(a.unsqueeze(-1)).squeeze(-1) + p
