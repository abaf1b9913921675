"""CPU: the EMD restatement (oracle/pzn_oracle.c) against the reference's only
numeric pin (PyTorchEMD/test_emd_loss.py:8-25) and against itself in fp64."""
import numpy as np

from oracle import point_ops as orc


def test_known_answer_0_71():
    p1 = np.array([[[1.7, -0.1, 0.1], [0.1, 1.2, 0.3]]], dtype=np.float32).repeat(3, 0)
    p2 = np.array([[[0.3, 1.8, 0.2], [1.2, -0.2, 0.3]]], dtype=np.float32).repeat(3, 0)
    cost, match = orc.earth_mover_distance(p1, p2)
    np.testing.assert_allclose(cost, [0.71] * 3, rtol=1e-5)
    # the matching is the cross assignment: match[l, k] ~ 1 for (l,k) in {(1,0),(0,1)}
    np.testing.assert_allclose(match[0], [[0, 1], [1, 0]], atol=1e-4)
    gc = np.array([0.5, 2.0, 1 / 3], np.float32)
    g1, g2 = orc.emd_matchcost_grad(gc, p1, p2, match)
    np.testing.assert_allclose(g1, 2 * (p1 - p2[:, ::-1]) * gc[:, None, None], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(g2, 2 * (p2 - p1[:, ::-1]) * gc[:, None, None], rtol=1e-3, atol=1e-4)


def test_fp32_agrees_with_fp64_and_invariants():
    rng = np.random.default_rng(0)
    for n, m in [(64, 64), (128, 64), (50, 150)]:
        x1 = rng.random((2, n, 3), dtype=np.float32)
        x2 = rng.random((2, m, 3), dtype=np.float32)
        c32, m32 = orc.earth_mover_distance(x1, x2)
        c64, m64 = orc.earth_mover_distance(x1.astype(np.float64), x2.astype(np.float64))
        np.testing.assert_allclose(c32, c64, rtol=2e-5)
        assert (m32 >= 0).all()
        multiL, multiR = (1, n // m) if n >= m else (m // n, 1)
        assert m32.sum(1).max() <= multiL * (1 + 1e-5)      # per xyz1 point
        assert m32.sum(2).max() <= multiR * (1 + 1e-5)      # per xyz2 point
    x = rng.random((1, 40, 3), dtype=np.float32)
    c, _ = orc.earth_mover_distance(x, x[:, rng.permutation(40)])
    assert c[0] < 1e-4


def test_hand_derived_capacity_vectors():
    """tests/emd_vectors.py: clustered clouds whose matching can be written down; they exercise the integer
    capacities multiL / multiR (emd_kernel.cu:29-35) and n != m, which the reference's 2-point vector does not."""
    from tests import emd_vectors as ev
    for case in ev.CASES:
        x1, x2, match, cost = case()
        c, m = orc.earth_mover_distance(x1, x2)
        assert m.shape == (1,) + match.shape
        np.testing.assert_allclose(m[0], match, atol=2e-5, err_msg=case.__name__)
        np.testing.assert_allclose(c[0], cost, rtol=1e-5, err_msg=case.__name__)
        g1, g2 = orc.emd_matchcost_grad(np.array([1.5], np.float32), x1, x2, m)
        w1, w2 = ev.gradients(x1, x2, match, 1.5)
        np.testing.assert_allclose(g1, w1, rtol=1e-4, atol=1e-4, err_msg=case.__name__)
        np.testing.assert_allclose(g2, w2, rtol=1e-4, atol=1e-4, err_msg=case.__name__)
