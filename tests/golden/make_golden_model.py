#!/usr/bin/env python3
"""Model-level golden fixtures: runs the REFERENCE model5_b.py on CPU.

See make_golden.py for the rules.  Harness-side placeholders (sys.modules
entries, never files in the repo) stand in for modules the image lacks and for
the two modules the reference repo does not ship.  The reference's EMD is
CUDA-only; for the training_step fixture `earth_mover_distance` is bound to the
C restatement in oracle/ (so that fixture pins the reference's GLUE around EMD,
not EMD itself — flagged `emd=oracle` in the fixture).
"""
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference_model():
    if "model5_b" in sys.modules and getattr(sys.modules["model5_b"], "__file__", "").startswith(REF):
        return sys.modules["model5_b"]
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if ROOT not in sys.path:
        sys.path.append(ROOT)

    class LightningModule(nn.Module):          # interface shell only
        current_epoch = 0
        global_step = 0
        logger = None

        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    pl = _placeholder("pytorch_lightning", LightningModule=LightningModule, seed_everything=lambda *a, **k: None)
    cb = _placeholder("pytorch_lightning.callbacks", ModelCheckpoint=object, early_stopping=types.ModuleType("es"))
    pl.callbacks = cb
    _placeholder("open3d")
    _placeholder("torchvision")
    _placeholder("plyfile", PlyData=object, PlyElement=object)
    _placeholder("pct")
    _placeholder("pointtransformer_partseg")
    _placeholder("emd_cuda")
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        mpl = _placeholder("matplotlib", use=lambda *a, **k: None, projections=None)
        _placeholder("matplotlib.pyplot")
        mpl.pyplot = sys.modules["matplotlib.pyplot"]
        _placeholder("mpl_toolkits")
        _placeholder("mpl_toolkits.mplot3d", Axes3D=object)
        _placeholder("pylab")
    import model5_b
    assert model5_b.__file__.startswith(REF)
    model5_b.math = math           # numpy-2 no longer re-exports `math` through pylab
    return model5_b


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
    m = "golden"
    output_path = "TRG"


def synth_batch(rng, B, N, nb=128):
    """8-tuple with the dataset contract (dataset.py:97-105): fpc, mrpc, igt, rpc, fpcb, rpcb, fpc_idx, rpc_idx."""
    import se_math.se3 as se3
    fpc = rng.random((B, N, 3), dtype=np.float32)
    rpc = rng.random((B, N, 3), dtype=np.float32)
    x = rng.standard_normal((B, 6))
    x = (0.8 * x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    igt = se3.exp(torch.from_numpy(x))
    mrpc = se3.transform(igt, torch.from_numpy(rpc).permute(0, 2, 1)).permute(0, 2, 1).contiguous()
    fi = np.zeros((B, N), np.float32)
    ri = np.zeros((B, N), np.float32)
    fpcb = np.zeros((B, nb, 3), np.float32)
    rpcb = np.zeros((B, nb, 3), np.float32)
    for b in range(B):
        a = rng.permutation(N)[:nb]
        c = rng.permutation(N)[:nb]
        fi[b, a] = 1
        ri[b, c] = 1
        fpcb[b] = fpc[b, np.sort(a)]
        rpcb[b] = rpc[b, np.sort(c)]
    return [torch.from_numpy(fpc), mrpc, igt, torch.from_numpy(rpc), torch.from_numpy(fpcb),
            torch.from_numpy(rpcb), torch.from_numpy(fi), torch.from_numpy(ri)]


def make_model():
    mb = import_reference_model()
    rng = np.random.default_rng(5150)
    G = {}
    torch.set_num_threads(8)

    # G6 layerAttention forward + input/param gradients
    att = mb.layerAttention(Cfg(), 256)
    fill_params(att)
    x = torch.from_numpy((0.5 * rng.standard_normal((1, 256, 256))).astype(np.float32)).requires_grad_(True)
    r, a = att(x)
    wr = torch.from_numpy(rng.standard_normal(r.shape).astype(np.float32))
    wa = torch.from_numpy(rng.standard_normal(a.shape).astype(np.float32))
    ((r * wr).sum() + (a * wa).sum()).backward()
    G["att_x"], G["att_r"], G["att_a"], G["att_wr"], G["att_wa"] = (t.detach().numpy() for t in (x, r, a, wr, wa))
    G["att_gx"] = x.grad.numpy()
    for n_, p in att.named_parameters():
        G["att_g_" + n_] = p.grad.numpy()

    # G7 encoder 5-tuple at N=1024 (reference default) and N=2048 (BASELINE size), train + eval
    for N in (1024, 2048):
        enc = mb.PCTransformer_nonsort(Cfg(), num_points=N)
        fill_params(enc)
        xyz = torch.from_numpy(rng.random((2, N, 3), dtype=np.float32))
        for mode in ("train", "eval"):
            enc.train(mode == "train")
            # fresh running stats each time so train-mode side effects do not leak into eval
            enc.bn1.reset_running_stats()
            enc.bn2.reset_running_stats()
            torch.manual_seed(1000 + N)
            f_global, x2, attention, out, x_feature = enc(xyz)
            tag = f"enc{N}_{mode}_"
            G[tag + "f_global"], G[tag + "x2"] = f_global.detach().numpy(), x2.detach().numpy()
            G[tag + "attention_s8"] = attention.detach().numpy()[:, ::8].copy()       # every 8th row
            G[tag + "x_feature_s8"] = x_feature.detach().numpy()[:, ::8].copy()
            G[tag + "out_max"] = out.detach().numpy().max(1)          # == f_global; full `out` is 2 MB per cloud
            G[tag + "out_sample"] = out.detach().numpy()[:, ::16, ::8].copy()
            if mode == "train":
                G[tag + "bn1_running_mean"] = enc.bn1.running_mean.numpy().copy()
                G[tag + "bn1_running_var"] = enc.bn1.running_var.numpy().copy()
        G[f"enc{N}_xyz"] = xyz.numpy()

    # G8 predict5 at config 1 (B=4, N=1024): eval and train
    model = mb.TouchedRegraster(Cfg())
    fill_params(model)
    G["n_params"] = np.array([sum(p.numel() for p in model.parameters())], np.int64)
    batch = synth_batch(rng, 4, 1024)
    for i, t in enumerate(batch):
        G[f"p5_batch{i}"] = t.numpy()
    for mode in ("eval", "train"):
        for e in (model.Encoder, model.Encoder2):
            e.bn1.reset_running_stats()
            e.bn2.reset_running_stats()
        torch.manual_seed(2024)
        with torch.no_grad():
            out = model.predict5(batch, 4, need=True, training=(mode == "train"))
        G[f"p5_{mode}_out"] = out[0].numpy()
        G[f"p5_{mode}_x2"], G[f"p5_{mode}_attention_s8"] = out[2].numpy(), out[3].numpy()[:, ::8].copy()
        G[f"p5_{mode}_mrpc_x2"], G[f"p5_{mode}_mrpc_attention_s8"] = out[4].numpy(), out[5].numpy()[:, ::8].copy()
        G[f"p5_{mode}_de_fpcb"], G[f"p5_{mode}_de_mrpcb"] = out[6].numpy(), out[7].numpy()

    np.savez_compressed(os.path.join(OUT, "model.npz"), **G)
    print("model.npz:", len(G), "arrays,", os.path.getsize(os.path.join(OUT, "model.npz")) >> 10, "KiB")


def make_loss():
    mb = import_reference_model()
    import se_math.se3 as se3
    from oracle import point_ops as orc
    rng = np.random.default_rng(777)
    G = {}

    # G9 se3.exp incl. the |w| < 0.01 Taylor branch, se3.transform
    tw = rng.standard_normal((6, 6)).astype(np.float32)
    tw[0, :3] *= 1e-3          # small-angle branch (sinc.py:12-16)
    tw[1, :3] = 0              # exactly zero rotation
    tw[2] *= 3.0
    g = se3.exp(torch.from_numpy(tw))
    pts = rng.random((6, 3, 50), dtype=np.float32)
    G["se3_twist"], G["se3_exp"] = tw, g.numpy()
    G["se3_pts"], G["se3_transform"] = pts, se3.transform(g, torch.from_numpy(pts)).numpy()
    t = torch.from_numpy(tw).requires_grad_(True)
    w = torch.from_numpy(rng.standard_normal((6, 4, 4)).astype(np.float32))
    (se3.exp(t) * w).sum().backward()
    G["se3_w"], G["se3_exp_grad"] = w.numpy(), t.grad.numpy()

    # G10 chamfer_loss and comp
    model = mb.TouchedRegraster(Cfg())
    a = torch.from_numpy(rng.random((3, 200, 3), dtype=np.float32)).requires_grad_(True)
    b = torch.from_numpy(rng.random((3, 200, 3), dtype=np.float32)).requires_grad_(True)
    d1, d2 = model.chamfer_loss(a, b)
    (d1.mean() + 2 * d2.mean()).backward()
    G["cd_a"], G["cd_b"], G["cd_d1"], G["cd_d2"] = (t_.detach().numpy() for t_ in (a, b, d1, d2))
    G["cd_ga"], G["cd_gb"] = a.grad.numpy(), b.grad.numpy()
    gg = se3.exp(torch.from_numpy(rng.standard_normal((5, 6)).astype(np.float32)))
    ig = se3.exp(torch.from_numpy(rng.standard_normal((5, 6)).astype(np.float32)))
    G["comp_g"], G["comp_igt"], G["comp_out"] = gg.numpy(), ig.numpy(), model.comp(gg, ig).numpy()

    # G11 whole training_step of the reference (its own glue: topk quirk, se3, chamfer x4,
    # comp, cross-entropy x2, top-128 boundary, loss-mode switch) with EMD = oracle restatement.
    class _OracleEMD(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x1, x2):
            a1, a2 = x1.detach().contiguous().numpy(), x2.detach().contiguous().numpy()
            match = orc.emd_approxmatch(a1, a2)
            ctx.save_for_backward(x1, x2, torch.from_numpy(match))
            return torch.from_numpy(orc.emd_matchcost(a1, a2, match))

        @staticmethod
        def backward(ctx, gc):
            x1, x2, match = ctx.saved_tensors
            g1, g2 = orc.emd_matchcost_grad(gc.contiguous().numpy(), x1.detach().contiguous().numpy(),
                                            x2.detach().contiguous().numpy(), match.numpy())
            return torch.from_numpy(g1), torch.from_numpy(g2)

    def oracle_emd(x1, x2, transpose=True):
        assert not transpose
        return _OracleEMD.apply(x1, x2)

    mb.earth_mover_distance = oracle_emd

    # The boundary branch picks the 128 most probable points with torch.topk (model5_b.py:1089-1091);
    # a pair of near-equal probabilities at rank 128/129 would make the fixture depend on rounding
    # noise.  Take the first batch seed whose rank-128 gap is comfortably above fp32 noise.
    def _gap(seed):
        cfg = Cfg()
        model = mb.TouchedRegraster(cfg)
        fill_params(model)
        b_ = synth_batch(np.random.default_rng(seed), 4, 1024)
        torch.manual_seed(99)
        with torch.no_grad():
            o = model.predict5(b_, 4, need=True, training=True)
        g_ = []
        for lg in (o[6], o[7]):
            s_ = torch.sort(torch.softmax(lg, dim=1)[:, 1, :], dim=1, descending=True)[0]
            g_.append((s_[:, 127] - s_[:, 128]).min().item())
        return min(g_)

    ts_seed = next(sd for sd in range(4242, 4342) if _gap(sd) > 1e-5)
    G["ts_seed"] = np.array([ts_seed], np.int64)
    print("training-step batch seed", ts_seed)
    from make_golden_model import synth_batch as _sb  # same function (module name when run via make_golden.py)
    for loss_mode, flags in ((0, {}), (1, dict(use_emd2=True, use_cd2=True, use_emd3=True))):
        cfg = Cfg()
        cfg.loss_mode = loss_mode
        for k_, v_ in flags.items():
            setattr(cfg, k_, v_)
        model = mb.TouchedRegraster(cfg)
        fill_params(model)
        model.configure_optimizers()                       # creates self.scheduler (read at model5_b.py:978)
        model.vis = lambda *a_, **k_: None                 # TensorBoard meshes (host-side logging only)
        model.vis_attention = lambda *a_, **k_: None
        batch = _sb(np.random.default_rng(ts_seed), 4, 1024)
        torch.manual_seed(99)
        model.zero_grad()
        loss = model.training_step(batch, 0)["loss"]
        loss.backward()
        tag = f"ts{loss_mode}_"
        if loss_mode == 0:
            for i, t_ in enumerate(batch):
                G[f"ts_batch{i}"] = t_.numpy()
        G[tag + "loss"] = np.array([loss.item()], np.float64)
        # gradients: a fingerprint per parameter tensor (L2 norm + 8 strided samples) keeps the fixture small
        names, norms, samples = [], [], []
        for n_, p in sorted(model.named_parameters()):
            gr = p.grad if p.grad is not None else torch.zeros_like(p)
            names.append(n_)
            norms.append(gr.norm().item())
            flat = gr.flatten()
            idx = torch.linspace(0, flat.numel() - 1, 8).long()
            samples.append(flat[idx].numpy())
        G[tag + "grad_names"] = np.array(names)
        G[tag + "grad_norms"] = np.array(norms, np.float64)
        G[tag + "grad_samples"] = np.stack(samples)
    G["ts_emd"] = np.array(["oracle"])
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **G)
    print("loss.npz:", len(G), "arrays,", os.path.getsize(os.path.join(OUT, "loss.npz")) >> 10, "KiB")


if __name__ == "__main__":
    what = sys.argv[1:] or ["model", "loss"]
    if "model" in what:
        make_model()
    if "loss" in what:
        make_loss()
