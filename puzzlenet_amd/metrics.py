"""Registration metrics of the eval path (reference `metrics.py`, used by
`model5_b.py:1426-1440` `compute_metrics`).  Host-side glue on B x 3 x 3 poses — the reference computes them
with numpy / scipy on the CPU as well — so tensors on any device are accepted and results come back as the
reference returns them: numpy arrays for the Euler-angle / translation errors, tensors for the isotropic ones.
"""
import math

import numpy as np
import torch
from scipy.spatial.transform import Rotation


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def inv_R_t(R, t):
    """Inverse of the rigid motion (R, t): (R^T, -R^T t).  metrics.py:7-10"""
    Rt = R.transpose(1, 2).contiguous()
    return Rt, -(Rt @ t.unsqueeze(-1)).squeeze(-1)


def anisotropic_R_error(r1, r2, seq="xyz", degrees=True):
    """Per-sample MSE / MAE between the Euler angles (extrinsic `seq`, degrees) of two rotation batches.
    metrics.py:12-35"""
    r1, r2 = _np(r1), _np(r2)
    assert r1.shape == r2.shape
    e1 = Rotation.from_matrix(r1).as_euler(seq, degrees=degrees)
    e2 = Rotation.from_matrix(r2).as_euler(seq, degrees=degrees)
    diff = e1 - e2
    return np.mean(diff ** 2, axis=-1), np.mean(np.abs(diff), axis=-1)


def anisotropic_t_error(t1, t2):
    """Per-sample MSE / MAE between two translation batches.  metrics.py:38-53"""
    t1, t2 = _np(t1), _np(t2)
    assert t1.shape == t2.shape
    diff = t1 - t2
    return np.mean(diff ** 2, axis=1), np.mean(np.abs(diff), axis=1)


def isotropic_R_error(r1, r2):
    """Geodesic angle (degrees) of r2^T r1.  metrics.py:56-72"""
    rel = r2.transpose(1, 2) @ r1
    tr = rel[:, 0, 0] + rel[:, 1, 1] + rel[:, 2, 2]
    return torch.acos(torch.clamp((tr - 1) / 2, -1, 1)) * (180.0 / math.pi)


def isotropic_t_error(t1, t2, R2):
    """|| R2^T t1 - R2^T t2 ||: translation error expressed in the frame of (R2, t2).  metrics.py:75-87"""
    R2i, t2i = inv_R_t(R2, t2)
    return torch.norm((R2i @ t1.unsqueeze(-1)).squeeze(-1) + t2i, dim=-1)
