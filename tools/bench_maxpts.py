#!/usr/bin/env python3
"""Sparse backward of the out projection + max over points (csrc/maxptsbwd.hip) on the model's shape, beside the dense
products it replaces:  python tools/bench_maxpts.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import _lib  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    B, L, E, nseg, Nout = 64, 256, 256, 5, 1024
    Kin = nseg * E
    g = torch.Generator().manual_seed(0)
    W = torch.randn(Nout, Kin, generator=g).to(dev)
    xs = [torch.randn(B * L, E, generator=g).to(dev) for _ in range(nseg)]
    y = torch.randn(B, L, Nout, generator=g).to(dev)
    dg = torch.randn(B, Nout, generator=g).to(dev)
    arg = y.argmax(dim=1).to(torch.int32).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    dx = torch.empty(B * L, Kin, device=dev)
    dW = torch.zeros(Nout, Kin, device=dev)
    db = torch.zeros(Nout, device=dev)
    segs = (ctypes.c_void_p * nseg)(*[x.data_ptr() for x in xs])
    ws = torch.empty(_lib.load().pzn_linear_maxpts_workspace_bytes(B, Nout) // 4, dtype=torch.int32, device=dev)
    for name, a in (("spread", arg), ("piled up", (arg % 3).contiguous()), ("one row", torch.zeros_like(arg))):
        t = timeit(lambda: _lib.call("pzn_linear_maxpts_dgrad_f32", dg.data_ptr(), a.data_ptr(), W.data_ptr(), B, L, Kin, Nout,
                                     ws.data_ptr(), dx.data_ptr(), st))
        print(f"sparse dgrad ({name:8s}) {t:8.1f} us   (dx write {dx.numel() * 4 / t / 1e6:.2f} TB/s)")
    t = timeit(lambda: _lib.call("pzn_linear_maxpts_wgrad_f32", dg.data_ptr(), arg.data_ptr(), segs, nseg, E, B, L, Nout,
                                 dW.data_ptr(), db.data_ptr(), st))
    print(f"sparse wgrad  {t:8.1f} us")
    dy = torch.randn(B * L, Nout, generator=g).to(dev)
    t = timeit(lambda: _lib.call("pzn_linear_dgrad_f32", dy.data_ptr(), None, W.data_ptr(), B * L, Kin, Nout, None, dx.data_ptr(), st))
    print(f"dense dgrad   {t:8.1f} us")


if __name__ == "__main__":
    main()
