"""Hand-derived multi-point EMD vectors (test data, shared by the CPU-oracle and GPU tests).

The reference's own test holds one 2-point known answer (PyTorchEMD/test_emd_loss.py:8-25), which never
reaches the capacity rule of the auction (emd_kernel.cu:29-35): with n >= m every point of cloud 1 offers
mass 1 and every point of cloud 2 takes `n / m` — an INTEGER division — and the other way round.  The clouds
below are clusters 10 units apart (exp(-4^j * 100) = 0 down to the last but one level, ~1e-11 there), so
the auction decides every cluster on its own and the answer can be written down:

  * a point of cloud 2 that is alone with its cloud-1 mates takes all they offer if its capacity allows;
  * mates at exactly the same distance (+-a along one axis: the squares are equal bit for bit) share a
    capacity-bound point equally, by symmetry of the recurrences.

Each case returns xyz1[1,n,3], xyz2[1,m,3], match[m,n] (emd_kernel.cu layout: match[l*n + k]) and the cost
sum(match * squared distance) (emd_kernel.cu:200-243).
"""
import numpy as np

_C = np.array([[0, 0, 0], [10, 0, 0], [0, 10, 0], [0, 0, 10]], np.float32)


def _finish(x1, x2, match):
    x1 = np.asarray(x1, np.float32)[None]
    x2 = np.asarray(x2, np.float32)[None]
    match = np.asarray(match, np.float64)
    d2 = ((x2[0][:, None, :].astype(np.float64) - x1[0][None, :, :]) ** 2).sum(-1)      # [m, n]
    return x1, x2, match, float((match * d2).sum())


def n4_m8():
    """n < m: multiL = m / n = 2, multiR = 1.  Every cloud-1 point serves its two mates in full."""
    a = np.array([0.5, 0.3, 0.7, 0.2], np.float32)
    b = np.array([0.25, 0.6, 0.1, 0.45], np.float32)
    z = np.zeros(4, np.float32)
    x2 = np.concatenate([_C + np.stack([a, z, z], 1), _C - np.stack([z, b, z], 1)], 0)
    match = np.concatenate([np.eye(4), np.eye(4)], 0)
    return _finish(_C.copy(), x2, match)


def n6_m4():
    """n > m with a remainder: multiR = 6 / 4 = 1 (integer), not 1.5.  Two clusters hold a symmetric pair of
    cloud-1 points that share ONE unit (0.5 each; a float division would hand out 0.75 each)."""
    a0, a1, c, d = 0.5, 0.25, 0.3, 0.7
    x1 = [_C[0] + [a0, 0, 0], _C[0] - [a0, 0, 0], _C[1] + [0, a1, 0], _C[1] - [0, a1, 0], _C[2] + [0, 0, c],
          _C[3] + [d, 0, 0]]
    match = np.zeros((4, 6))
    match[0, 0] = match[0, 1] = match[1, 2] = match[1, 3] = 0.5
    match[2, 4] = match[3, 5] = 1.0
    return _finish(x1, _C.copy(), match)


def n5_m2():
    """multiR = 5 / 2 = 2 (integer), not 2.5: three equidistant mates share two units (2/3 each), the two mates
    of the other point are served in full although they sit at different distances."""
    a, b1, b2 = 0.5, 0.25, 0.75
    c = _C[:2]
    x1 = [c[0] + [a, 0, 0], c[0] - [a, 0, 0], c[0] + [0, a, 0], c[1] + [0, b1, 0], c[1] - [0, 0, b2]]
    match = np.zeros((2, 5))
    match[0, :3] = 2.0 / 3.0
    match[1, 3:] = 1.0
    return _finish(x1, c.copy(), match)


def n4_m6():
    """n < m with a remainder: multiL = 6 / 4 = 1 (integer).  The mirror image of n6_m4."""
    x1, x2, match, cost = n6_m4()
    return x2, x1, match.T.copy(), cost


CASES = (n4_m8, n6_m4, n5_m2, n4_m6)


def gradients(x1, x2, match, grad_cost):
    """matchcostgrad1 / matchcostgrad2 (emd_kernel.cu:333-355, 286-327): d cost / d xyz for a constant match."""
    diff = x1[0][None, :, :].astype(np.float64) - x2[0][:, None, :]            # [m, n, 3] = x1_k - x2_l
    g1 = 2.0 * grad_cost * (match[:, :, None] * diff).sum(0)
    g2 = -2.0 * grad_cost * (match[:, :, None] * diff).sum(1)
    return g1[None], g2[None]
