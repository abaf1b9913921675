"""Per-tensor gradient error of one training_step against the torch-CPU restatement (what
tests/test_gpu_model.py::test_training_step_full_gradients_vs_oracle asserts), worst tensors first.
  python tests/_grad_rows.py [loss_mode [two_streams]]
  JITTER=1e-7 [JSEED=k]  relative noise on Encoder2's per-point features (how sensitive is the gradient to their last bits?)
  XF_FROM_DEVICE=1       the oracle computes downstream of the DEVICE's per-point features
  PZN_STEM_FUSED=0       the four-launch stem
A diagnostic that lives in tests/ because it imports the oracle (only tests/, smoke() and bench.py's cpu_baseline may)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from oracle import model_ref as mr      # a diagnostic, like the tests: the checker, never the product
    from puzzlenet_amd import model5_b as mb, ops
    loss_mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    G = np.load(os.path.join(ROOT, "tests", "golden", "loss.npz"))
    dev = torch.device("cuda:0")
    flags = {} if loss_mode == 0 else dict(use_emd2=True, use_cd2=True, use_emd3=True)
    cfg = mr.Cfg(loss_mode=loss_mode, **flags)
    ops.clear_grad_sinks()
    mb._STEM_TWO = False      # the wrappers below take and return ONE tensor
    m = mb.TouchedRegraster(cfg)
    mr.fill_params(m)
    ref = mr.RefModel(cfg)
    ref.load_state_dict(m.state_dict(), strict=True)
    m.to(dev)
    if len(sys.argv) > 2:
        m.two_streams = bool(int(sys.argv[2]))
    jitter = float(os.environ.get("JITTER", "0"))      # relative noise on Encoder2's per-point features: how sensitive is the gradient?
    if jitter:
        enc = m.Encoder2
        orig = enc.local_features
        gen = torch.Generator(device=dev).manual_seed(int(os.environ.get("JSEED", "0")))
        enc.local_features = lambda xyz: (lambda y: y * (1 + jitter * torch.randn(y.shape, device=dev, generator=gen)))(orig(xyz))
    batch = [torch.from_numpy(np.ascontiguousarray(G[f"ts_batch{i}"])).to(dev) for i in range(8)]
    xf_dev = {}
    if os.environ.get("XF_FROM_DEVICE"):      # the oracle computes downstream of the DEVICE's per-point features (same gates, same arg-max)
        for name in ("Encoder", "Encoder2"):
            enc = getattr(m, name)
            enc.local_features = (lambda orig, name: lambda xyz: xf_dev.setdefault(name, orig(xyz)))(enc.local_features, name)
    torch.manual_seed(99)
    loss = m.training_step(batch, 0)["loss"]
    if xf_dev:
        import torch.nn.functional as F
        from oracle.model_ref import sample_and_group
        def fwd(self, xyz, name):
            xf = F.relu(self.bn2(self.mlp2(F.relu(self.bn1(self.mlp1(xyz))))))
            xf = xf + (xf_dev[name].detach().cpu() - xf.detach())
            x, f1 = sample_and_group(512, 0, 32, xyz, xf, False, True)
            f1f = torch.max(F.relu(self.mlp4(F.relu(self.mlp3(f1)))), dim=-2)[0]
            x2, f2 = sample_and_group(256, 0, 32, x, f1f, False, True)
            f2f = torch.max(F.relu(self.mlp6(F.relu(self.mlp5(f2)))), dim=-2)[0]
            a1, w1 = self.atten1(f2f); a2, w2 = self.atten2(a1); a3, w3 = self.atten3(a2); a4, w4 = self.atten4(a3)
            out = self.out(torch.cat([a1, a2, a3, a4, f2f], dim=-1))
            return torch.max(out, dim=1)[0], x2, (w1 + w2 + w3 + w4) / 4, out, xf
        for name in ("Encoder", "Encoder2"):
            e = getattr(ref, name)
            e.forward = (lambda e, name: lambda xyz: fwd(e, xyz, name))(e, name)
    torch.manual_seed(99)
    ref_loss = ref.training_step([t.cpu() for t in batch])
    ref_loss = ref_loss[0] if isinstance(ref_loss, tuple) else ref_loss
    loss.backward()
    ref_loss.backward()
    rp = dict(ref.named_parameters())
    rows = []
    for name, p in m.named_parameters():
        g = (p.grad if p.grad is not None else torch.zeros_like(p)).cpu().double()
        gr = rp[name].grad
        gr = (gr if gr is not None else torch.zeros_like(rp[name])).double()
        rows.append((float((g - gr).norm()), float(gr.norm()), name))
    total = sum(r * r for _, r, _ in rows) ** 0.5
    print("loss", loss.item(), ref_loss.item(), "total", total, "rel", sum(e * e for e, _, _ in rows) ** 0.5 / total)
    for e, r, name in sorted(rows, key=lambda t: -t[0] / (t[1] + 1e-6 * total))[:10]:
        print(f"{name:40s} err {e:.3e} norm {r:.3e} rel {e / (r + 1e-30):.3e}")
    rels = sorted(e / (r + 1e-6 * total / 1e-2) for e, r, _ in rows)      # e <= 1e-2 r + 1e-6 total  <=>  this <= 1e-2
    n = len(rels)
    print("tensors", n, "quantiles 50/80/90/95/100 %:", [f"{rels[min(n - 1, int(q * n))]:.2e}" for q in (0.5, 0.8, 0.9, 0.95, 1.0)])
