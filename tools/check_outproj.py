"""The fused out projection + max over points (csrc/outproj.hip) against float64 torch and against the composed form.

    python tools/check_outproj.py [B]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import _lib, ops  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda:0")
    L, E, Nout = 256, 256, 1024
    M = B * L
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(M, E, generator=g).to(dev) for _ in range(5)]
    w = (torch.randn(Nout, 5 * E, generator=g) / 36).to(dev)
    b = (0.1 * torch.randn(Nout, generator=g)).to(dev)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(lib.pzn_outproj_maxpts_workspace_bytes(L, E, 5, Nout), dtype=torch.uint8, device=dev)
    out = torch.empty(M, Nout, device=dev)
    fmax = torch.empty(B, Nout, device=dev)
    arg = torch.empty(B, Nout, dtype=torch.int32, device=dev)

    def fused(o):
        _lib.call("pzn_outproj_maxpts_fwd_f32", ops._ptrs(xs), 5, w.data_ptr(), b.data_ptr(), B, L, E, Nout,
                  o.data_ptr() if o is not None else None, fmax.data_ptr(), arg.data_ptr(), ws.data_ptr(), st)

    y = torch.empty(M, Nout, device=dev)
    f2 = torch.empty(B, Nout, device=dev)
    a2 = torch.empty(B, Nout, dtype=torch.int32, device=dev)

    def composed():
        for i, xi in enumerate(xs):
            _lib.call("pzn_linear_slice_fwd_f32", xi.data_ptr(), w.data_ptr() + 4 * E * i, 5 * E, b.data_ptr(), M, E, Nout,
                      int(i > 0), y.data_ptr(), st)
        _lib.call("pzn_maxpool_points_fwd_f32", y.data_ptr(), B, L, Nout, f2.data_ptr(), a2.data_ptr(), st)

    def timeit(fn, n=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    t_full = timeit(lambda: fused(out))
    t_max = timeit(lambda: fused(None))
    t_comp = timeit(composed)
    fused(out)
    composed()
    torch.cuda.synchronize()
    nb = min(B, 4)
    X = torch.cat([x[: nb * L].double() for x in xs], dim=1)
    ref = X @ w.double().T + b.double()
    rel = float((out[: nb * L].double() - ref).norm() / ref.norm())
    fm, am = ref.view(nb, L, Nout).max(dim=1)
    fl = 2.0 * M * 5 * E * Nout
    print(f"B={B}: fused+out {t_full:.0f} us  fused max only {t_max:.0f} us ({fl / t_max / 1e6:.0f} TF/s)  composed {t_comp:.0f} us | "
          f"out vs fp64 {rel:.2e}  fmax vs fp64 {float((fmax[:nb].double() - fm).abs().max()):.2e}  "
          f"arg != fp64 {int((arg[:nb].long() != am).sum())}  arg != composed {int((arg != a2).sum())} of {arg.numel()}  "
          f"|fmax - composed| {float((fmax - f2).abs().max()):.2e}")


if __name__ == "__main__":
    main()
