#!/usr/bin/env python3
"""Launch time of the weight-stationary linear kernel against the row count (fixed cost vs throughput):
    python tools/bench_ws_scaling.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import _lib  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
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
    st = torch.cuda.current_stream().cuda_stream
    for K, N in ((64, 64), (128, 128), (256, 256)):
        W = torch.randn(N, K, device=dev) / 8
        b = torch.randn(N, device=dev)
        for M in (4096, 16384, 32768, 65536, 131072, 262144, 524288, 1048576):
            if M * (K + N) * 4 > 2e9:
                continue
            x = torch.randn(M, K, device=dev)
            y = torch.empty(M, N, device=dev)
            t = timeit(lambda: _lib.call("pzn_linear_fwd_f32", x.data_ptr(), W.data_ptr(), b.data_ptr(), M, K, N, 1, y.data_ptr(), st))
            mb = M * (K + N) * 4 / 1e6
            print(f"K={K:4d} N={N:4d} M={M:8d}: {t:7.1f} us   {mb / t:5.2f} TB/s   {2.0 * M * K * N / t / 1e6:6.1f} TFLOP/s")
    x = torch.randn(1 << 24, device=dev)
    y = torch.empty_like(x)
    for n in (1 << 20, 1 << 22, 1 << 24):
        t = timeit(lambda: torch.mul(x[:n], 2.0, out=y[:n]))
        print(f"elementwise y = 2x on {n * 4 / 1e6:6.1f} MB in + out: {t:6.1f} us   {n * 8 / t / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
