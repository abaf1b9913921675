"""Chained attention kernels (csrc/attnfused.hip) against the composed chain node and a float64 torch composition.

    python tools/check_attn_fused.py [B] [nprob]

Prints relative errors of outputs and gradients and the per-entry-point times of both paths.
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import ops  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def ref64(x, blocks, w, b):
    """model5_b.py:67-101, 462-475 in float64."""
    cur, maps, outs = x, [], []
    for wq, bq, wk, bk, wv, bv, wo, bo in blocks:
        q, k, v = cur @ wq.T + bq, cur @ wk.T + bk, cur @ wv.T + bv
        a = torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(q.shape[-1]), dim=-1)
        r = cur - a @ v
        cur = cur + torch.relu(r @ wo.T + bo)
        maps.append(a)
        outs.append(cur)
    y = torch.cat(outs + [x], dim=-1) @ w.T + b
    return y, sum(maps) / 4, y.max(dim=1)[0]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    nprob = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    use = sys.argv[3] if len(sys.argv) > 3 else "max"     # "max": loss on f_global (sparse backward, arg-max sensitive); "out": loss on out
    dev = torch.device("cuda:0")
    L, E, dk, Nout = 256, 256, 64, 1024
    g = torch.Generator().manual_seed(11)
    shapes = [(dk, E), (dk,), (dk, E), (dk,), (E, E), (E,), (E, E), (E,)]
    probs = []
    for _ in range(nprob):
        x0 = (0.5 * torch.randn(B, L, E, generator=g)).to(dev)
        blocks0 = [[(torch.randn(*s, generator=g) / (math.sqrt(E) if len(s) == 2 else 4)).to(dev) for s in shapes]
                   for _ in range(4)]
        w0 = (torch.randn(Nout, 5 * E, generator=g) / math.sqrt(5 * E)).to(dev)
        b0 = (0.1 * torch.randn(Nout, generator=g)).to(dev)
        wg = torch.randn(B, Nout, generator=g).to(dev) if use == "max" else (torch.randn(B, L, Nout, generator=g) / 16).to(dev)
        probs.append((x0, blocks0, w0, b0, wg))

    def leafs(dtype=torch.float32):
        out = []
        for x0, blocks0, w0, b0, wg in probs:
            x = x0.to(dtype).clone().requires_grad_(True)
            blocks = [[p.to(dtype).clone().requires_grad_(True) for p in blk] for blk in blocks0]
            w, b = w0.to(dtype).clone().requires_grad_(True), b0.to(dtype).clone().requires_grad_(True)
            out.append((x, blocks, w, b, wg.to(dtype)))
        return out

    def grads(lf):
        return [[x.grad] + [p.grad for blk in blocks for p in blk] + [w.grad, b.grad] for x, blocks, w, b, _ in lf]

    def run_fused():
        lf = leafs()
        assert ops.attention_chain_fused_supported(lf[0][0], dk, lf[0][2])
        res = ops.attention_chain_fused([l[0] for l in lf], [l[1] for l in lf], [l[2] for l in lf], [l[3] for l in lf])
        loss = sum((r[2 if use == 'max' else 0] * l[4]).sum() for r, l in zip(res, lf))
        loss.backward()
        return [(r[0].detach(), r[1].detach(), r[2].detach()) for r in res], grads(lf)

    def run_composed():
        lf = leafs()
        res = [ops.attention_chain_out(x, blocks, w, b) for x, blocks, w, b, _ in lf]
        loss = sum((r[2 if use == 'max' else 0] * l[4]).sum() for r, l in zip(res, lf))
        loss.backward()
        return [(r[0].detach(), r[1].detach(), r[2].detach()) for r in res], grads(lf)

    def run_ref():
        lf = leafs(torch.float64)
        res = [ref64(x, blocks, w, b) for x, blocks, w, b, _ in lf]
        loss = sum((r[2 if use == 'max' else 0] * l[4]).sum() for r, l in zip(res, lf))
        loss.backward()
        return [(r[0].detach(), r[1].detach(), r[2].detach()) for r in res], grads(lf)

    of, gf = run_fused()
    have_c = B * 256 >= 4096
    orf, gr = run_ref() if B <= 64 else (None, None)
    oc, gc = run_composed() if have_c else (orf, gr)
    torch.cuda.synchronize()
    names = ["dx"] + [f"blk{i}.{n}" for i in range(4) for n in ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo")] + ["w_out", "b_out"]
    worst = 0.0
    for p in range(nprob):
        print(f"problem {p}: fused vs composed  y {rel(of[p][0], oc[p][0]):.2e}  map {rel(of[p][1], oc[p][1]):.2e}  fg {rel(of[p][2], oc[p][2]):.2e}")
        if orf is not None:
            print(f"           fused vs fp64      y {rel(of[p][0], orf[p][0]):.2e}  map {rel(of[p][1], orf[p][1]):.2e}  fg {rel(of[p][2], orf[p][2]):.2e}")
            print(f"           composed vs fp64   y {rel(oc[p][0], orf[p][0]):.2e}  map {rel(oc[p][1], orf[p][1]):.2e}")
        for n, a, b in zip(names, gf[p], gc[p]):
            e = float((a - b).abs().max()) / max(float(b.abs().max()), 5e-2)
            worst = max(worst, e)
            extra = ""
            if gr is not None:
                r64 = gr[p][names.index(n)]
                extra = f"   vs fp64: fused {rel(a, r64):.2e} composed {rel(b, r64):.2e}"
            if e > 1e-4 or n in ("dx", "blk0.wq", "blk0.wo", "blk3.wv"):
                print(f"   grad {n:10s} max-abs-rel {e:.2e}  L2-rel {rel(a, b):.2e}{extra}")
    print("worst gradient deviation", f"{worst:.2e}")

    # timing
    for name, fn in (("fused", run_fused), ("composed", run_composed))[:2 if have_c else 1]:
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ops.KernelTimer.start()
        fn()
        rec = ops.KernelTimer.stop()
        tot = sum(v[1] for v in rec.values())
        print(f"--- {name}: {tot:.3f} ms in entry points")
        for k, (n, ms) in sorted(rec.items(), key=lambda kv: -kv[1][1]):
            print(f"    {k:36s} x{n:3d} {ms:8.3f} ms")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"    wall per fwd+bwd: {e0.elapsed_time(e1) / 5:.3f} ms")


if __name__ == "__main__":
    main()
