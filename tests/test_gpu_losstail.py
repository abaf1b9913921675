"""GPU parity of the loss-tail kernels (csrc/losstail.hip) against plain torch restatements of the
reference's lines evaluated on the CPU (fp64 where it is arithmetic, exact where it is a selection)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_se3_transform_points_vs_reference_formula():
    """se_math/se3.py:110-120 (R a + p on the transposed view), forward and both gradients."""
    from puzzlenet_amd import se3
    g = torch.Generator().manual_seed(1)
    tw = torch.randn(5, 6, generator=g)
    pts = torch.randn(5, 777, 3, generator=g)
    w = torch.randn(5, 777, 3, generator=g)
    twr = tw.double().requires_grad_(True)
    pr = pts.double().requires_grad_(True)
    Gr = se3.exp(twr)
    outr = se3.transform(Gr, pr.permute(0, 2, 1)).permute(0, 2, 1)
    (outr * w.double()).sum().backward()
    twd = tw.to(DEV).requires_grad_(True)
    pd = pts.to(DEV).requires_grad_(True)
    Gd = se3.exp(twd)
    out = se3.transform_points(Gd, pd)
    assert _rel(out, outr) < 1e-6
    (out * w.to(DEV)).sum().backward()
    assert _rel(pd.grad, pr.grad) < 1e-6 and _rel(twd.grad, twr.grad) < 1e-5
    # the reference's call shape (transform of the permuted view) takes the same kernel
    out2 = se3.transform(Gd.detach(), pd.detach().permute(0, 2, 1)).permute(0, 2, 1)
    assert torch.equal(out2, out.detach())


def test_comp_loss_vs_reference_formula():
    """model5_b.py:1512-1519: 16 * mse(g igt, I)."""
    from puzzlenet_amd import ops, se3
    gen = torch.Generator().manual_seed(2)
    g = se3.exp(torch.randn(64, 6, generator=gen))
    igt = se3.exp(torch.randn(64, 6, generator=gen))
    gr = g.double().requires_grad_(True)
    A = gr.matmul(igt.double())
    I = torch.eye(4, dtype=torch.float64).view(1, 4, 4).repeat(64, 1, 1)
    want = F.mse_loss(A, I, reduction="mean") * 16
    (3.0 * want).backward()
    gd = g.to(DEV).requires_grad_(True)
    got = ops.comp_loss(gd, igt.to(DEV))
    assert abs(float(got) - float(want)) <= 1e-6 * abs(float(want))
    (3.0 * got).backward()
    assert _rel(gd.grad, gr.grad) < 1e-5


@pytest.mark.parametrize("B,N", [(64, 2048), (3, 100), (1, 4096)])
def test_boundary_ce_vs_torch(B, N):
    """model5_b.py:1063-1064 F.cross_entropy([B,2,N], labels) and :1085-1090 softmax(.,1)[:,1,:]."""
    from puzzlenet_amd import ops
    gen = torch.Generator().manual_seed(N)
    logits = 3 * torch.randn(B, 2, N, generator=gen)
    labels = (torch.rand(B, N, generator=gen) < 0.07).float()
    lr = logits.double().requires_grad_(True)
    want = F.cross_entropy(lr, labels.long())
    (2.0 * want).backward()
    ld = logits.to(DEV).requires_grad_(True)
    loss, prob = ops.boundary_ce(ld, labels.to(DEV))
    assert abs(float(loss) - float(want)) <= 2e-6 * abs(float(want))
    assert _rel(prob, torch.softmax(logits.double(), dim=1)[:, 1, :]) < 1e-6
    (2.0 * loss).backward()
    assert _rel(ld.grad, lr.grad) < 1e-5


@pytest.mark.parametrize("B,N", [(64, 2048), (3, 77)])
def test_boundary_ce_on_the_permuted_head_output(B, N):
    """The heads produce [B,N,2]; the reference permutes to [B,2,N] (model5_b.py:751-754).  ops.boundary_ce reads that VIEW through
    its strides (no transposed copy) and returns the gradient in the same layout: same loss / probabilities / gradient as on a
    contiguous copy, bit for bit, and the gradient that reaches the head output is contiguous."""
    from puzzlenet_amd import ops
    gen = torch.Generator().manual_seed(B + N)
    head = (3 * torch.randn(B, N, 2, generator=gen)).to(DEV)
    labels = (torch.rand(B, N, generator=gen) < 0.07).float().to(DEV)
    a = head.clone().requires_grad_(True)
    la, pa = ops.boundary_ce(a.permute(0, 2, 1), labels)
    b = head.permute(0, 2, 1).contiguous().requires_grad_(True)
    lb, pb = ops.boundary_ce(b, labels)
    assert torch.equal(la, lb) and torch.equal(pa, pb)
    seen = {}
    def hook(g):
        seen["contig"] = g.is_contiguous()      # (a hook that returns a value replaces the gradient)
    a.register_hook(hook)
    (3.0 * la).backward()
    (3.0 * lb).backward()
    assert torch.equal(a.grad.permute(0, 2, 1), b.grad)
    assert seen["contig"]


@pytest.mark.parametrize("R,N,K", [(128, 2048, 128), (5, 100, 7), (3, 16384, 256), (2, 4096, 128), (4, 300, 300 - 44)])
def test_topk_rows_vs_torch(R, N, K):
    """model5_b.py:1089-1091 torch.topk(x, K, 1)[1]: random rows (no ties: the same indices in the same order), rows with
    many equal values (same values slot by slot, equal values by ascending index), negative values, constant rows."""
    from puzzlenet_amd import ops
    gen = torch.Generator().manual_seed(R * N + K)
    K = min(K, 256)
    x = torch.randn(R, N, generator=gen)
    got = ops.topk_rows(x.to(DEV), K).cpu()
    want = torch.topk(x, K, dim=1)[1]
    gvals, wvals = torch.gather(x, 1, got), torch.gather(x, 1, want)
    assert torch.equal(gvals, wvals)                           # the same values slot by slot
    # the same indices wherever the value is unique in its row (262 k float32 normals do hold a few exact duplicates,
    # whose order torch leaves unspecified)
    uniq = torch.ones_like(gvals, dtype=torch.bool)
    uniq[:, 1:] &= gvals[:, 1:] != gvals[:, :-1]
    uniq[:, :-1] &= gvals[:, :-1] != gvals[:, 1:]
    assert torch.equal(got[uniq], want[uniq]) and float(uniq.float().mean()) > 0.99
    assert all(len(set(r.tolist())) == K for r in got)         # no index twice
    # ties: quantised values; every value class must come out lowest index first
    q = torch.round(x * 4) / 4
    q[0] = -1.5
    if R > 1:
        q[1, : N // 2] = 2.0
    gi = ops.topk_rows(q.to(DEV), K).cpu()
    gv = torch.gather(q, 1, gi)
    wv = torch.topk(q, K, dim=1)[0]
    assert torch.equal(gv, wv)
    assert bool((gv[:, 1:] <= gv[:, :-1]).all())
    same = gv[:, 1:] == gv[:, :-1]
    assert bool((gi[:, 1:][same] > gi[:, :-1][same]).all())
    for r in range(R):      # inside the last value class the LOWEST indices are taken
        last = gv[r, -1]
        cls = torch.nonzero(q[r] == last).flatten()
        taken = gi[r][gv[r] == last]
        assert torch.equal(torch.sort(taken)[0], cls[: taken.numel()])


def test_avg4_and_colmean_argmax_vs_torch():
    """model5_b.py:468-469 (mean of the four attention maps) and :937-942 (attention.mean(dim=1) -> first top-k index)."""
    from puzzlenet_amd import ops
    gen = torch.Generator().manual_seed(5)
    maps = [torch.rand(6, 256, 256, generator=gen) for _ in range(4)]
    d = [m.to(DEV).requires_grad_(True) for m in maps]
    out = ops.avg4(*d)
    want = (((maps[0] + maps[1]) + maps[2]) + maps[3]) / 4
    assert torch.equal(out.detach().cpu(), want)
    w = torch.rand(6, 256, 256, generator=gen)
    (out * w.to(DEV)).sum().backward()
    assert torch.allclose(d[2].grad.cpu(), w / 4)
    mean, arg = ops.colmean_argmax(out.detach())
    wm = want.double().mean(dim=1)
    assert _rel(mean, wm) < 1e-6
    assert torch.equal(arg.cpu(), torch.topk(wm, 32)[1][:, 0])
    z = torch.zeros(2, 40, 100)
    z[0, :, 7] = 1.0
    z[0, :, 70] = 1.0           # tie: the lower column
    z[1, :, 99] = 0.5
    assert ops.colmean_argmax(z.to(DEV))[1].tolist() == [7, 99]
