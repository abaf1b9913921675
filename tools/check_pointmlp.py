"""The fused per-point MLP chains (csrc/pointmlp.hip) against float64 torch and against the layer-by-layer launches,
with launch times on the heads' shape (M = 64 x 2048 rows)."""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import _lib, ops

dev = torch.device("cuda:0")
_p = lambda t: t.data_ptr()


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(B, N, C2, C3, per_cloud):
    M = B * N
    g = torch.Generator().manual_seed(C2 + C3 + B)
    x = torch.randn(M, 64, generator=g)
    ldw1 = 128 if per_cloud else 64
    W1f = torch.randn(64, ldw1, generator=g) / 8
    W1 = W1f[:, ldw1 - 64:]
    b1 = torch.randn(B, 64, generator=g) if per_cloud else 0.1 * torch.randn(64, generator=g)
    W2, b2 = torch.randn(C2, 64, generator=g) / 8, 0.1 * torch.randn(C2, generator=g)
    W3, b3 = torch.randn(C3, C2, generator=g) / C2 ** 0.5, 0.1 * torch.randn(C3, generator=g)
    xd = x.double()
    bias1 = b1.double().repeat_interleave(N, 0) if per_cloud else b1.double()
    h1r = F.relu(xd @ W1.double().t() + bias1)
    h2r = F.relu(h1r @ W2.double().t() + b2.double())
    yr = h2r @ W3.double().t() + b3.double()
    d = [t.to(dev).contiguous() for t in (x, W1f, b1, W2, b2, W3, b3)]
    h1, h2, y = (torch.empty(M, c, device=dev) for c in (64, C2, C3))
    st = torch.cuda.current_stream().cuda_stream
    w1p = _p(d[1]) + (ldw1 - 64) * 4

    def fused():
        _lib.check(_lib.load().pzn_point_mlp3_fwd_f32(_p(d[0]), M, N, w1p, ldw1, _p(d[2]), 1 if per_cloud else 0, _p(d[3]),
                                                      _p(d[4]), _p(d[5]), _p(d[6]), C2, C3, _p(h1), _p(h2), _p(y), st), "fwd")
    fused()
    torch.cuda.synchronize()
    print(f"B={B} N={N} 64->64->{C2}->{C3} per_cloud={per_cloud}: rel err h1 {rel(h1, h1r):.2e} h2 {rel(h2, h2r):.2e} y {rel(y, yr):.2e}",
          end="")
    if M >= 65536:
        w1c = d[1][:, ldw1 - 64:].contiguous()

        def composed():
            if per_cloud:
                a = ops.linear(d[0], w1c, None, False)
                _lib.check(_lib.load().pzn_cloud_bias_relu_f32(_p(a), _p(d[2]), B, N, 64, st), "cb")
            else:
                a = ops.linear(d[0], w1c, d[2], True)
            a = ops.linear(a, d[3], d[4], True)
            return ops.linear(a, d[5], d[6], False)
        with torch.no_grad():
            print(f" | fused {timeit(fused):.1f} us, layer by layer {timeit(composed):.1f} us", end="")
    print()


def run_bwd(B, N, C2, C3, per_cloud):
    """ops.point_mlp3 forward + backward against float64 autograd, and its time against the layer-by-layer path."""
    g = torch.Generator().manual_seed(7 + C2 + B)
    x = torch.randn(B, N, 64, generator=g)
    gl = torch.randn(B, 1, 64, generator=g) if per_cloud else None
    W1 = torch.randn(64, 128 if per_cloud else 64, generator=g) / 8
    b1 = 0.1 * torch.randn(64, generator=g)
    W2, b2 = torch.randn(C2, 64, generator=g) / 8, 0.1 * torch.randn(C2, generator=g)
    W3, b3 = torch.randn(C3, C2, generator=g) / C2 ** 0.5, 0.1 * torch.randn(C3, generator=g)
    go = torch.randn(B, N, C3, generator=g)
    leaves = [x, W1, b1, W2, b2, W3, b3] + ([gl] if per_cloud else [])
    ref = [t.double().requires_grad_(True) for t in leaves]
    xin = torch.cat([ref[7].expand(-1, N, -1), ref[0]], -1) if per_cloud else ref[0]
    h2r = F.relu(F.linear(F.relu(F.linear(xin, ref[1], ref[2])), ref[3], ref[4]))
    yr = F.linear(h2r, ref[5], ref[6])
    gate_h2 = h2r.detach() > 0
    (yr * go.double()).sum().backward()
    d = [t.to(dev).requires_grad_(True) for t in leaves]
    y = ops.point_mlp3(d[0], d[1], d[2], d[3], d[4], d[5], d[6], g=d[7] if per_cloud else None)
    (y * go.to(dev)).sum().backward()
    names = ["x", "W1", "b1", "W2", "b2", "W3", "b3", "g"]
    errs = " ".join(f"{n} {rel(a.grad, r.grad):.1e}" for n, a, r in zip(names, d, ref))
    print(f"bwd B={B} N={N} ->{C2}->{C3} per_cloud={per_cloud}: y {rel(y, yr):.1e} | grads {errs}", end="")
    if B * N >= 65536:
        god = go.to(dev)

        def fused():
            for t in d:
                t.grad = None
            (ops.point_mlp3(d[0], d[1], d[2], d[3], d[4], d[5], d[6], g=d[7] if per_cloud else None) * god).sum().backward()

        def composed():
            for t in d:
                t.grad = None
            a = ops.cat_global_linear_relu(d[0], d[7], d[1], d[2]) if per_cloud else ops.linear(d[0], d[1], d[2], True)
            a = ops.linear(a, d[3], d[4], True)
            (ops.linear(a, d[5], d[6], False) * god).sum().backward()
        print(f" | fwd+bwd fused {timeit(fused):.0f} us, layer by layer {timeit(composed):.0f} us", end="")
    print()


for cfg in [(2, 64, 64, 64, False), (3, 96, 32, 2, True), (5, 2048, 64, 64, False), (64, 2048, 64, 64, False), (64, 2048, 32, 2, True)]:
    run(*cfg)
for cfg in [(2, 64, 64, 64, False), (3, 96, 32, 2, True), (4, 2048, 32, 2, True), (3, 160, 64, 64, False), (64, 2048, 64, 64, False),
            (64, 2048, 32, 2, True)]:
    run_bwd(*cfg)
