"""Per-point first set-abstraction layer (csrc/sapoint.hip) on the encoder's two levels: gather forward, inverse
lists, list-sum backward — launch times and the HBM rate of the dominant stream (h write / dh read)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import ops
dev = torch.device('cuda:0')
P_, S_, C_ = ops._p, ops._stream, ops._call
def timeit(fn, name, nbytes, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    print('%-46s %8.3f ms  %6.2f TB/s' % (name, ms, nbytes / ms / 1e9), flush=True)
g = torch.Generator().manual_seed(0)
for (B, N, S, D, C1) in [(64, 2048, 512, 64, 128), (64, 512, 256, 128, 256)]:
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    new_xyz = xyz[:, :S].contiguous()
    idx = torch.empty(B, S, 32, dtype=torch.int64, device=dev)
    C_("pzn_knn_f32", P_(xyz), P_(new_xyz), B, N, S, 32, P_(idx), S_())
    Pm = torch.randn(B * N, C1, device=dev); W1 = torch.randn(C1, 3 + D, device=dev); b1 = torch.randn(C1, device=dev)
    h = torch.empty(B * S * 32, C1, device=dev); dh = torch.randn(B * S * 32, C1, device=dev)
    off = torch.empty(B * (N + 1), dtype=torch.int32, device=dev)
    rows = torch.empty(B * S * 32, dtype=torch.int32, device=dev); pts = torch.empty_like(rows)
    dP = torch.empty(B * N, C1, device=dev); dW1 = torch.zeros(C1, 3 + D, device=dev); db1 = torch.zeros(C1, device=dev)
    nb = 4 * B * S * 32 * C1
    timeit(lambda: C_("pzn_sa_point_l1_fwd_f32", P_(xyz), P_(new_xyz), P_(idx), P_(Pm), P_(W1), P_(b1), B, N, S, D, C1, P_(h), S_()),
           f'fwd  N={N} S={S} C1={C1}', nb)
    timeit(lambda: C_("pzn_knn_inverse_lists", P_(idx), B, N, S, 32, P_(off), P_(rows), P_(pts), S_()), f'inverse lists', 8 * B * S * 32)
    timeit(lambda: C_("pzn_sa_point_l1_bwd_f32", P_(dh), P_(xyz), P_(new_xyz), P_(rows), P_(pts), B, N, S, D, C1, P_(dP), P_(dW1), P_(db1), S_()),
           f'bwd  N={N} S={S} C1={C1}', nb)
