"""One layerAttention block through the chained kernels, every intermediate against float64 torch.

    python tools/check_attn_fused_block.py [B]
"""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import _lib, ops  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device("cuda:0")
    L, E, dk = 256, 256, 64
    M = B * L
    g = torch.Generator().manual_seed(3)
    x = (0.5 * torch.randn(M, E, generator=g)).to(dev)
    wq, wk = [(torch.randn(dk, E, generator=g) / 16).to(dev) for _ in range(2)]
    wv, wo = [(torch.randn(E, E, generator=g) / 16).to(dev) for _ in range(2)]
    bq, bk = [(torch.randn(dk, generator=g) / 4).to(dev) for _ in range(2)]
    bv, bo = [(torch.randn(E, generator=g) / 4).to(dev) for _ in range(2)]
    dr = torch.randn(M, E, generator=g).to(dev)
    lib = _lib.load()
    P = ops._ptrs
    st = torch.cuda.current_stream().cuda_stream
    raw = lambda n: torch.zeros(n, dtype=torch.uint8, device=dev)
    mk = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    W = raw(lib.pzn_attn_fused_weight_bytes())
    _lib.call("pzn_attn_fused_prep_weights", wq.data_ptr(), wk.data_ptr(), wv.data_ptr(), wo.data_ptr(), W.data_ptr(), st)
    qkb, vb = lib.pzn_attn_fused_qk_image_bytes(B), lib.pzn_attn_fused_v_image_bytes(B)
    qrp, krp, vrp = raw(qkb), raw(qkb), raw(vb)
    _lib.call("pzn_attn_fused_proj", 1, P([x]), P([W]), P([bq]), P([bk]), P([bv]), B, P([qrp]), P([krp]), P([vrp]), st)
    r, t, lse, amap = mk(M, E), mk(M, E), mk(M), mk(B, L, L)
    mask = torch.zeros((M, 8), dtype=torch.int32, device=dev)
    _lib.call("pzn_attn_fused_fwd", 1, P([x]), P([qrp]), P([krp]), P([vrp]), P([W]), P([bo]), B, P([r]), P([t]), P([mask]),
              P([amap]), P([lse]), 0, 1.0, st)
    dz, u, dq, delta = mk(M, E), mk(M, E), mk(M, dk), mk(M)
    dqt = mk(M, dk)
    darp = raw(vb)
    _lib.call("pzn_attn_fused_bwd_q", 1, P([dr]), E, None, E, P([mask]), P([qrp]), P([krp]), P([vrp]), P([W]), B, P([dz]), P([u]),
              P([dq]), P([dqt]), P([darp]), P([delta]), st)
    dkk, dvv, dx = mk(M, dk), mk(M, E), mk(M, E)
    _lib.call("pzn_attn_fused_bwd_k", 1, P([qrp]), P([krp]), P([vrp]), P([darp]), P([W]), P([lse]),
              P([delta]), P([u]), P([dqt]), B, P([dkk]), P([dvv]), P([dx]), st)
    torch.cuda.synchronize()

    D = torch.float64
    X = x.to(D).view(B, L, E)
    q = X @ wq.to(D).T + bq.to(D)
    k = X @ wk.to(D).T + bk.to(D)
    v = X @ wv.to(D).T + bv.to(D)
    s = q @ k.transpose(1, 2) / 8
    Pm = torch.softmax(s, dim=-1)
    a = Pm @ v
    tt = X - a
    z = tt @ wo.to(D).T + bo.to(D)
    rr = X + torch.relu(z)
    DR = dr.to(D).view(B, L, E)
    DZ = DR * (z > 0)
    DT = DZ @ wo.to(D)
    DA = -DT
    DP = DA @ v.transpose(1, 2)
    dl = (Pm * DP).sum(-1)
    DS = Pm * (DP - dl[..., None]) / 8
    DQ = DS @ k
    DKr = DS.transpose(1, 2) @ q
    DVr = Pm.transpose(1, 2) @ DA
    U = DR + DT
    DX = U + DQ @ wq.to(D) + DKr @ wk.to(D) + DVr @ wv.to(D)
    LSE = torch.logsumexp(s, dim=-1)
    for name, got, want in (("r", r, rr), ("t", t, tt), ("map", amap, Pm), ("lse", lse, LSE), ("dz", dz, DZ), ("delta", delta, dl),
                            ("dq", dq, DQ),
                            ("u", ops.attention_tile_image_rows(u, E), U), ("dk", dkk, DKr), ("dv", dvv, DVr), ("dx", dx, DX)):
        print(f"{name:6s} rel {rel(got.reshape(-1), want.reshape(-1)):.3e}")
    print("dq tile image == dq rows:", bool(torch.equal(ops.attention_tile_image_rows(dqt, dk), dq)))


if __name__ == "__main__":
    main()
