import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import ops
dev = torch.device("cuda:0")
for C2, C3, pc in ((64, 64, False), (32, 2, True)):
    g = torch.Generator().manual_seed(1)
    B, N = 64, 2048
    x = torch.randn(B, N, 64, generator=g).to(dev).requires_grad_(True)
    gl = torch.randn(B, 1, 64, generator=g).to(dev).requires_grad_(True) if pc else None
    W1 = (torch.randn(64, 128 if pc else 64, generator=g) / 8).to(dev).requires_grad_(True)
    b1 = torch.randn(64, generator=g).to(dev).requires_grad_(True)
    W2 = (torch.randn(C2, 64, generator=g) / 8).to(dev).requires_grad_(True); b2 = torch.randn(C2, generator=g).to(dev).requires_grad_(True)
    W3 = (torch.randn(C3, C2, generator=g) / 8).to(dev).requires_grad_(True); b3 = torch.randn(C3, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(B, N, C3, generator=g).to(dev)
    for _ in range(5):
        y = ops.point_mlp3(x, W1, b1, W2, b2, W3, b3, g=gl)
        y.backward(go)
    torch.cuda.synchronize()
