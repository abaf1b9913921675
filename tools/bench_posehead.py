"""Few-row layers of the pose head (64 rows) and the narrow per-point heads (131072 rows): forward / input-gradient /
weight-gradient launch times."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
st = lambda: torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr() if t is not None else None
def timeit(fn, name, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    print('%-40s %8.1f us' % (name, a.elapsed_time(b) / iters * 1e3), flush=True)
for (M, K, N) in [(64, 2048, 1024), (64, 1024, 512), (64, 512, 512), (64, 512, 256), (64, 256, 6),
                  (131072, 64, 64), (131072, 128, 64), (131072, 64, 32), (131072, 3, 64), (131072, 32, 2)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); dy = torch.randn(M, N, device=dev); dx = torch.empty(M, K, device=dev)
    dW = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    timeit(lambda: lib.pzn_linear_fwd_f32(P(x), P(w), P(b), M, K, N, 1, P(y), st()), f'fwd   {M}x{K}x{N}')
    timeit(lambda: lib.pzn_linear_dgrad_f32(P(dy), P(y), P(w), M, K, N, None, P(dx), st()), f'dgrad {M}x{K}x{N}')
    timeit(lambda: lib.pzn_linear_wgrad_f32(P(dy), P(y), P(x), M, K, N, P(dW), P(db), 0, st()), f'wgrad {M}x{K}x{N}')
