#!/usr/bin/env python3
"""One layerAttention block (model5_b.py:83-101) forward + backward on the model's shape [64, 256, 256], dk = 64:
    python tools/bench_attn.py [iters]
Under `rocprofv3 --kernel-trace --stats -- python3 tools/bench_attn.py` the per-kernel table shows what the 5 + 11
launches of a block cost each."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import ops  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    B, L, E, dk = 64, 256, 256, 64
    g = torch.Generator().manual_seed(0)
    x = (0.5 * torch.randn(B, L, E, generator=g)).to(dev).requires_grad_(True)
    shapes = [(dk, E), (dk,), (dk, E), (dk,), (E, E), (E,), (E, E), (E,)]
    ps = [(torch.randn(*s, generator=g) / (math.sqrt(E) if len(s) == 2 else 4)).to(dev).requires_grad_(True) for s in shapes]
    wy = torch.randn(B, L, E, generator=g).to(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for it in range(iters + 3):
        ev[0].record()
        y, a = ops.attention_block(x, *ps)
        ev[1].record()
        y.backward(wy)
        ev[2].record()
        torch.cuda.synchronize()
        if it >= 3:
            tf += ev[0].elapsed_time(ev[1])
            tb += ev[1].elapsed_time(ev[2])
        x.grad = None
        for p in ps:
            p.grad = None
    print(f"block forward {tf / iters * 1e3:.1f} us, backward {tb / iters * 1e3:.1f} us")


if __name__ == "__main__":
    main()
