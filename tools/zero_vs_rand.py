import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import _lib
lib = _lib.load(); dev = torch.device('cuda:0')
st = lambda: torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr() if t is not None else None
def timeit(fn, flops, name, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    print('%-50s %8.3f ms  %6.1f TF/s' % (name, ms, flops / ms / 1e9), flush=True)
M, K, N = 524288, 256, 256
for kind in ('rand', 'zero', 'rand'):
    x = torch.randn(M, K, device=dev) if kind == 'rand' else torch.zeros(M, K, device=dev)
    w = torch.randn(N, K, device=dev) if kind == 'rand' else torch.zeros(N, K, device=dev)
    b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    R = M // 32; out = torch.empty(R, N, device=dev); arg = torch.empty(R, N, dtype=torch.int32, device=dev)
    timeit(lambda: lib.pzn_linear_maxpool_fwd_f32(P(x), P(w), P(b), R, K, N, P(out), P(arg), st()), 2 * M * K * N, f'maxpool fwd 524288x256x256 {kind}')
    timeit(lambda: lib.pzn_linear_fwd_f32(P(x), P(w), P(b), M, K, N, 1, P(y), st()), 2 * M * K * N, f'linear fwd 524288x256x256 {kind}')
    M2 = 16384
    x2 = torch.randn(M2, 1280, device=dev) if kind == 'rand' else torch.zeros(M2, 1280, device=dev)
    w2 = torch.randn(1024, 1280, device=dev) if kind == 'rand' else torch.zeros(1024, 1280, device=dev)
    y2 = torch.empty(M2, 1024, device=dev)
    timeit(lambda: lib.pzn_linear_fwd_f32(P(x2), P(w2), None, M2, 1280, 1024, 0, P(y2), st()), 2 * M2 * 1280 * 1024, f'wide fwd 16384x1280x1024 {kind}')
