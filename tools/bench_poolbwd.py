"""Micro-benchmark of the max-pool backward entry points (sparse pass in poolbwd.hip) on the two encoder shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr() if t is not None else None


def timeit(fn, gbytes, name, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    print("%-40s %8.3f ms  %7.1f GB/s" % (name, ms, gbytes / ms * 1e3), flush=True)


for (M, K, N, tag) in [(1048576, 128, 128, "mlp4"), (524288, 256, 256, "mlp6")]:
    R = M // 32
    x = torch.randn(M, K, device=dev).relu_(); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    out = torch.empty(R, N, device=dev); arg = torch.empty(R, N, dtype=torch.int32, device=dev)
    lib.pzn_linear_maxpool_fwd_f32(P(x), P(w), P(b), R, K, N, P(out), P(arg), st())
    dout = torch.randn(R, N, device=dev); dx = torch.empty(M, K, device=dev)
    dW = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    gb = M * K * 4 / 1e9
    timeit(lambda: lib.pzn_linear_maxpool_dgrad_f32(P(dout), P(arg), P(out), P(w), R, K, N, P(x), P(dx), st()), 2 * gb, f"dgrad+mask {tag}")
    timeit(lambda: lib.pzn_linear_maxpool_dgrad_f32(P(dout), P(arg), P(out), P(w), R, K, N, None, P(dx), st()), gb, f"dgrad nomask {tag}")
    timeit(lambda: lib.pzn_linear_maxpool_wgrad_f32(P(dout), P(arg), P(out), P(x), R, K, N, P(dW), P(db), 0, st()), gb, f"wgrad {tag}")
