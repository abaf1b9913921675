"""torch-CPU restatement of the reference's live model path — TEST INFRASTRUCTURE ONLY.

Follows the reference's own op sequence (Python-loop FPS, materialised
square_distance + full argsort, torch.gather, nn.Linear attention, bmm chamfer;
pointnet_util.py:22-136, model5_b.py:67-101, 411-478, 672-759, 912-1155,
1495-1519; se_math/se3.py:57-80,110-120).  EMD = the C restatement in
pzn_oracle.c.  Pinned against the reference itself by tests/golden/model.npz and
loss.npz (tests/test_oracle_model.py).  Used as: the checker in tests and in
smoke(), and the `cpu_baseline` (kind "port") leg of bench.py.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import contextlib
import time

from . import point_ops as orc

# bench.py's cpu_baseline sets this to a dict: host seconds per stage of the forward pass, accumulated over the steps
# (fps / knn / group / mlp / attention / heads / chamfer / emd / emd_backward / loss_tail)
STAGE_CLOCK = None


@contextlib.contextmanager
def _stage(name):
    if STAGE_CLOCK is None:
        yield
        return
    t0 = time.perf_counter()
    try:
        yield
    finally:
        STAGE_CLOCK[name] = STAGE_CLOCK.get(name, 0.0) + time.perf_counter() - t0


# ---- pointnet_util.py ------------------------------------------------------------------
def square_distance(src, dst):
    return torch.sum((src[:, :, None] - dst[:, None]) ** 2, dim=-1)              # :36


def index_points(points, idx):
    raw = idx.size()
    idx = idx.reshape(raw[0], -1)
    res = torch.gather(points, 1, idx[..., None].expand(-1, -1, points.size(-1)))  # :49
    return res.reshape(*raw, -1)


def farthest_point_sample(xyz, npoint):
    B, N, C = xyz.shape
    centroids = torch.zeros(B, npoint, dtype=torch.long)
    distance = torch.ones(B, N) * 1e10
    farthest = torch.randint(0, N, (B,), dtype=torch.long)                        # :65
    batch_indices = torch.arange(B, dtype=torch.long)
    for i in range(npoint):                                                       # :67-72
        centroids[:, i] = farthest
        centroid = xyz[batch_indices, farthest, :].view(B, 1, 3)
        dist = torch.sum((xyz - centroid) ** 2, -1)
        distance = torch.min(distance, dist)
        farthest = torch.max(distance, -1)[1]
    return centroids


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False, knn=False):
    B, N, C = xyz.shape
    S = npoint
    with _stage("fps"):
        fps_idx = farthest_point_sample(xyz, npoint)
        new_xyz = index_points(xyz, fps_idx)
    assert knn, "model5_b only uses knn=True (model5_b.py:449,456)"
    with _stage("knn"):
        dists = square_distance(new_xyz, xyz)
        idx = dists.argsort(stable=True)[:, :, :nsample]       # stable: ties -> ascending index
    with _stage("group"):
        grouped_xyz = index_points(xyz, idx)
        grouped_xyz_norm = grouped_xyz - new_xyz.view(B, S, 1, C)
        if points is not None:
            new_points = torch.cat([grouped_xyz_norm, index_points(points, idx)], dim=-1)
        else:
            new_points = grouped_xyz_norm
    if returnfps:
        return new_xyz, new_points, grouped_xyz, fps_idx
    return new_xyz, new_points


# ---- se_math ------------------------------------------------------------------------------
def _sinc(t, small, big):
    s = t.abs() < 0.01
    safe = torch.where(s, torch.ones_like(t), t)
    return torch.where(s, small(t * t), big(safe))


def se3_exp(x):
    x_ = x.reshape(-1, 6)
    w, v = x_[:, 0:3], x_[:, 3:6]
    t = w.norm(p=2, dim=1).view(-1, 1, 1)
    O = torch.zeros_like(w[:, 0])
    W = torch.stack((torch.stack((O, -w[:, 2], w[:, 1]), 1), torch.stack((w[:, 2], O, -w[:, 0]), 1),
                     torch.stack((-w[:, 1], w[:, 0], O), 1)), 1)
    S = W.bmm(W)
    I = torch.eye(3).to(w)
    s1 = _sinc(t, lambda t2: 1 - t2 / 6 * (1 - t2 / 20 * (1 - t2 / 42)), lambda u: torch.sin(u) / u)
    s2 = _sinc(t, lambda t2: 1 / 2 * (1 - t2 / 12 * (1 - t2 / 30 * (1 - t2 / 56))), lambda u: (1 - torch.cos(u)) / (u * u))
    s3 = _sinc(t, lambda t2: 1 / 6 * (1 - t2 / 20 * (1 - t2 / 42 * (1 - t2 / 72))), lambda u: (u - torch.sin(u)) / u ** 3)
    R = I + s1 * W + s2 * S
    V = I + s2 * W + s3 * S
    p = V.bmm(v.contiguous().view(-1, 3, 1))
    z = torch.tensor([0., 0., 0., 1.]).view(1, 1, 4).repeat(x_.size(0), 1, 1).to(x)
    return torch.cat((torch.cat((R, p), 2), z), 1).view(*x.size()[:-1], 4, 4)


def se3_transform(g, a):
    R = g[..., 0:3, 0:3]
    p = g[..., 0:3, 3]
    return R.matmul(a) + p.unsqueeze(-1)


# ---- model5_b.py ----------------------------------------------------------------------------
class LayerAttention(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.mlpq, self.mlpk = nn.Linear(d, d // 4), nn.Linear(d, d // 4)
        self.mlpv, self.out = nn.Linear(d, d), nn.Linear(d, d)

    def forward(self, x):
        q, k, v = self.mlpq(x), self.mlpk(x), self.mlpv(x)
        attn = F.softmax(torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(q.size(-1)), dim=-1)
        r = x - torch.matmul(attn, v)
        return x + F.relu(self.out(r)), attn


def _pool(x, dim, pins, key):
    """torch.max(x, dim)[0], or - with `pins` = {"winners": {key: int64 index tensor}, "flips": {}} - the entries of x AT the
    given winners (the checked implementation's arg-max) instead: the function downstream, and its gradient, is then the one
    of exactly those winners, whatever rounding did to a near-tie.  pins["flips"][key] = (entries whose own arg-max differs,
    the largest gap max - x[winner] among them relative to the largest |max|): the caller bounds both."""
    best, own = torch.max(x, dim)
    if pins is None or key not in pins["winners"]:
        if pins is not None:
            pins.setdefault("own", {})[key] = own
        return best
    w = pins["winners"][key].to(torch.long)
    got = torch.gather(x, dim, w.unsqueeze(dim)).squeeze(dim)
    diff = (own != w) & (best != got)         # (equal values: a tie, either index is an arg-max)
    gap = float(((best - got) * diff).detach().max() / best.detach().abs().max().clamp_min(1e-30)) if bool(diff.any()) else 0.0
    pins["flips"][key] = (int(diff.sum()), gap, own.numel())
    return got


class Encoder(nn.Module):
    pins, tag = None, ""      # tests: see _pool

    def __init__(self, num_points=1024):
        super().__init__()
        self.mlp1, self.mlp2 = nn.Linear(3, 64), nn.Linear(64, 64)
        self.mlp3, self.mlp4 = nn.Linear(67, 128), nn.Linear(128, 128)
        self.mlp5, self.mlp6 = nn.Linear(131, 256), nn.Linear(256, 256)
        self.bn1, self.bn2 = nn.BatchNorm1d(num_points), nn.BatchNorm1d(num_points)
        self.atten1, self.atten2, self.atten3, self.atten4 = (LayerAttention(256) for _ in range(4))
        self.out = nn.Linear(1280, 1024)

    def forward(self, xyz):
        with _stage("mlp"):
            xf = F.relu(self.bn1(self.mlp1(xyz)))
            xf = F.relu(self.bn2(self.mlp2(xf)))
        x, f1 = sample_and_group(512, 0, 32, xyz, xf, False, True)
        with _stage("mlp"):
            f1f = _pool(F.relu(self.mlp4(F.relu(self.mlp3(f1)))), 2, self.pins, self.tag + "sa1")
        x2, f2 = sample_and_group(256, 0, 32, x, f1f, False, True)
        with _stage("mlp"):
            f2f = _pool(F.relu(self.mlp6(F.relu(self.mlp5(f2)))), 2, self.pins, self.tag + "sa2")
        with _stage("attention"):
            a1, w1 = self.atten1(f2f)
            a2, w2 = self.atten2(a1)
            a3, w3 = self.atten3(a2)
            a4, w4 = self.atten4(a3)
            attention = (w1 + w2 + w3 + w4) / 4
        with _stage("mlp"):
            out = self.out(torch.cat([a1, a2, a3, a4, f2f], dim=-1))
            return _pool(out, 1, self.pins, self.tag + "gmax"), x2, attention, out, xf


class _Dec(nn.Module):
    def __init__(self):
        super().__init__()
        self.mlp1, self.mlp2, self.mlp3 = nn.Linear(512, 512), nn.Linear(512, 256), nn.Linear(256, 2)


def _seq(*d):
    L = []
    for i in range(len(d) - 1):
        L.append(nn.Linear(d[i], d[i + 1]))
        if i < len(d) - 2:
            L.append(nn.ReLU())
    return nn.Sequential(*L)


class _OracleEMD(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2):
        with _stage("emd"):
            a1, a2 = x1.detach().contiguous().numpy(), x2.detach().contiguous().numpy()
            match = orc.emd_approxmatch(a1, a2)
            ctx.save_for_backward(x1, x2, torch.from_numpy(match))
            return torch.from_numpy(orc.emd_matchcost(a1, a2, match))

    @staticmethod
    def backward(ctx, gc):
        x1, x2, match = ctx.saved_tensors
        with _stage("emd_backward"):
            g1, g2 = orc.emd_matchcost_grad(gc.contiguous().numpy(), x1.detach().contiguous().numpy(),
                                            x2.detach().contiguous().numpy(), match.numpy())
        return torch.from_numpy(g1), torch.from_numpy(g2)


def earth_mover_distance(x1, x2):
    return _OracleEMD.apply(x1, x2)


def chamfer_loss(x, y):
    with _stage("chamfer"):
        n = x.size(1)
        xx, yy, zz = torch.bmm(x, x.transpose(2, 1)), torch.bmm(y, y.transpose(2, 1)), torch.bmm(x, y.transpose(2, 1))
        d = torch.arange(0, n)
        rx = xx[:, d, d].unsqueeze(1).expand_as(xx)
        ry = yy[:, d, d].unsqueeze(1).expand_as(yy)
        P = rx.transpose(2, 1) + ry - 2 * zz
        return torch.min(P, 1)[0], torch.min(P, 2)[0]


def comp(g, igt):
    A = g.matmul(igt)
    I = torch.eye(4).to(A).view(1, 4, 4).repeat(A.size(0), 1, 1)
    return F.mse_loss(A, I, reduction='mean') * 16


class RefModel(nn.Module):
    """Same parameter names as the reference's TouchedRegraster (and the product's)."""

    def __init__(self, config):
        super().__init__()
        self.C = config
        n = int(getattr(config, "num_points", 1024))
        self.Encoder, self.Encoder2 = Encoder(n), Encoder(n)
        self.Encoder.tag, self.Encoder2.tag = "Encoder.", "Encoder2."
        self.pins = None
        self.fpc_decoder, self.rpc_decoder = _Dec(), _Dec()
        self.dt = nn.Parameter(torch.full((1, 6), 1.0e-2))
        self.tfMLP = _seq(2048, 1024, 512, 512, 256, 6)
        self.MLPLocalPreRpc, self.MLPLocalPreFpc = _seq(64, 64, 64, 64), _seq(64, 64, 64, 64)
        self.MLPRpcb, self.MLPFpcb = _seq(128, 64, 32, 2), _seq(128, 64, 32, 2)

    def pin_winners(self, winners):
        """Tests: force the max-pools to the given winners ({"Encoder.sa1": [B,512,128], "Encoder.sa2", "Encoder.gmax",
        "Encoder2.*", "heads.gmax"}: index tensors; see _pool).  -> the dict whose "flips" the next forward fills."""
        self.pins = None if winners is None else {"winners": dict(winners), "flips": {}}
        self.Encoder.pins = self.Encoder2.pins = self.pins
        return self.pins

    def predict5(self, batch, training=False):
        for m in (self.Encoder, self.Encoder2, self.tfMLP, self.fpc_decoder, self.rpc_decoder):
            m.train(training)
        fpc, mrpc = batch[0], batch[1]
        N = fpc.shape[1]
        ff = self.Encoder(fpc)
        fm = self.Encoder2(mrpc)
        with _stage("heads"):
            out = self.tfMLP(torch.cat([ff[0], fm[0]], dim=-1))
            lf = self.MLPLocalPreFpc(ff[4])
            lm = self.MLPLocalPreRpc(fm[4])
            gf = _pool(lm, 1, self.pins, "heads.gmax").unsqueeze(1).repeat(1, N, 1)       # model5_b.py:741 (the reference's bug)
            gm = gf
            de_fpcb = self.MLPFpcb(torch.cat([gm, lf], dim=-1)).permute(0, 2, 1)
            de_mrpcb = self.MLPRpcb(torch.cat([gf, lm], dim=-1)).permute(0, 2, 1)
        return out, [0], ff[1], ff[2], fm[1], fm[2], de_fpcb, de_mrpcb

    def training_step(self, batch, return_terms=False):
        with _stage("forward_total"):
            return self._training_step(batch, return_terms)

    def _training_step(self, batch, return_terms=False):
        fpc, mrpc, igt, rpc, fpcb, rpcb, fpc_idx, rpc_idx = batch
        C = self.C
        N = fpc.shape[1]
        out, _, x2, attention, mx2, mattention, de_fpcb, de_mrpcb = self.predict5(batch, training=True)
        x2att1 = x2[:, torch.topk(attention.mean(dim=1), 32)[1][:, 0]]
        x2att2 = mx2[:, torch.topk(mattention.mean(dim=1), 32)[1][:, 0]]
        mat = se3_exp(out)
        de_mrpc = se3_transform(mat, mrpc.permute(0, 2, 1)).permute(0, 2, 1)
        d1, d2 = chamfer_loss(rpc, de_mrpc)
        red = torch.sum if C.loss_sum else torch.mean
        loss_re = red(d1) + red(d2)
        loss_g = comp(mat, igt)
        c1, c2 = chamfer_loss(x2att1, x2att2)
        loss_emd = red(earth_mover_distance(de_mrpc, rpc))
        loss_cd2 = red(c1) + red(c2)
        emd2 = torch.sum(earth_mover_distance(x2att1, x2att2))
        loss = {0: loss_re + loss_g, 1: loss_re + loss_g + loss_emd, 2: loss_emd, 3: loss_emd + loss_g,
                4: loss_emd + loss_re, 5: loss_g, 6: loss_re}[C.loss_mode]
        if C.use_emd2:
            loss = loss + emd2
        if C.use_cd2:
            loss = loss + loss_cd2
        loss = loss + F.cross_entropy(de_fpcb, fpc_idx.squeeze().long()) + F.cross_entropy(de_mrpcb, rpc_idx.squeeze().long())
        fi = torch.topk(torch.softmax(de_fpcb, dim=1)[:, 1, :], 128, 1)[1]
        mi = torch.topk(torch.softmax(de_mrpcb, dim=1)[:, 1, :], 128, 1)[1]
        pf = torch.gather(fpc, 1, fi.unsqueeze(-1).repeat(1, 1, 3))
        pm = torch.gather(mrpc, 1, mi.unsqueeze(-1).repeat(1, 1, 3))
        a1, a2 = chamfer_loss(pf, fpcb)
        inv = se3_transform(se3_exp(out), pm.permute(0, 2, 1)).permute(0, 2, 1)
        b1, b2 = chamfer_loss(inv, rpcb)
        e1 = torch.mean(earth_mover_distance(pf, fpcb))
        e2 = torch.mean(earth_mover_distance(inv, rpcb))
        loss = loss + (b1.mean() + b2.mean()) + (a1.mean() + a2.mean())
        if C.use_emd3:
            loss = loss + e1 + e2
        if return_terms:
            return loss, {"train/loss_re": loss_re, "train/loss_g": loss_g, "train/loss_emd": loss_emd,
                          "train/cd2": loss_cd2, "train_emd2": emd2, "train/loss_fpcb": a1.mean() + a2.mean(),
                          "train/loss_rpcb": b1.mean() + b2.mean(), "train/loss_emd_fpcb": e1,
                          "train/loss_emc_mrpcb": e2, "fi": fi, "mi": mi, "out": out,
                          "de_fpcb": de_fpcb, "de_mrpcb": de_mrpcb}
        return loss


def fill_params(module):
    """Closed-form pseudo-random parameter fill (no RNG state, no state-dict file): parameters in
    sorted-name order; element i of the k-th tensor is u = frac(sin(12.9898 i + 78.233 (k+1)) * 43758.5453)
    in (-1, 1), evaluated in float64 (a 1-ulp libm difference moves u by ~4e-12).  Weights are scaled to
    uniform(-sqrt(3/fan_in), sqrt(3/fan_in)); BatchNorm weights 1 + 0.1u; biases 0.1u."""
    with torch.no_grad():
        for k, (name, p) in enumerate(sorted(module.named_parameters())):
            if name.endswith("dt"):
                continue
            i = np.arange(p.numel(), dtype=np.float64)
            u = torch.from_numpy(np.modf(np.sin(i * 12.9898 + (k + 1) * 78.233) * 43758.5453)[0])
            if p.dim() == 2:
                v = u * math.sqrt(3.0 / p.shape[1])
            elif "bn" in name and name.endswith("weight"):
                v = 1.0 + 0.1 * u
            else:
                v = 0.1 * u
            p.copy_(v.reshape(p.shape).to(p.dtype))


class Cfg:
    dataset = "cad"
    loss_mode = 1
    loss_sum = False
    use_emd2 = False
    use_cd2 = False
    use_emd3 = False
    pretrain_epochs = 0
    lr = 0.9e-3
    m = "oracle"
    output_path = "TRG"
    num_points = 1024

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)
