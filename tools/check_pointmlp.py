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


for cfg in [(2, 64, 64, 64, False), (3, 96, 32, 2, True), (5, 2048, 64, 64, False), (64, 2048, 64, 64, False), (64, 2048, 32, 2, True)]:
    run(*cfg)
