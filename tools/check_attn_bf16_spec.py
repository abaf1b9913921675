"""The chained attention kernels in the bf16 mode against the float64 statement of their arithmetic (operands rounded to
bf16 where the default path splits them) and against the unrounded float64 block:  python tools/check_attn_bf16_spec.py [B]"""
import os
import sys

os.environ.setdefault("PZN_ATTN_PRECISION", "bf16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_gpu_dense as T  # noqa: E402


def main():
    from puzzlenet_amd import _lib, ops
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
    print("attention precision mode", lib.pzn_attn_get_precision())
    P = ops._ptrs
    st = torch.cuda.current_stream().cuda_stream
    raw = lambda n: torch.zeros(n, dtype=torch.uint8, device=dev)
    mk = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    W = raw(lib.pzn_attn_fused_weight_bytes())
    _lib.call("pzn_attn_fused_prep_weights", wq.data_ptr(), wk.data_ptr(), wv.data_ptr(), wo.data_ptr(), W.data_ptr(), st)
    qkb, vb = lib.pzn_attn_fused_qk_image_bytes(B), lib.pzn_attn_fused_v_image_bytes(B)
    qrp, krp, vrp = raw(qkb), raw(qkb), raw(vb)
    _lib.call("pzn_attn_fused_proj", 1, P([x]), P([W]), P([bq]), P([bk]), P([bv]), B, P([qrp]), P([krp]), P([vrp]), st)
    torch.cuda.synchronize()

    def decode(img, F):     # plane 0 of an Rp image -> [B, 256, F]
        v = img.view(B, F // 16, 3, 8, 64, 16)[:, :, 0].contiguous().view(torch.bfloat16).view(B, F // 16, 8, 2, 32, 8).double()
        out = torch.zeros(B, 256, F, dtype=torch.float64, device=img.device)
        for ks in range(F // 16):
            for h in range(2):
                for j in range(8):
                    out[:, :, 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3)] = v[:, ks, :, h, :, j].reshape(B, 256)
        return out
    D = torch.float64
    bf = lambda t_: t_.to(torch.float32).to(torch.bfloat16).to(D)
    Xb = bf(x.view(B, L, E))
    for nm, img, w_, b_, F in (("q", qrp, wq, bq, dk), ("k", krp, wk, bk, dk), ("v", vrp, wv, bv, E)):
        want = bf(Xb @ bf(w_).T + b_.to(D))
        print(f"image {nm}: rel L2 vs bf16(statement) {float((decode(img, F) - want).norm() / want.norm()):.3e}")
    r, t, lse, amap = mk(M, E), mk(M, E), mk(M), mk(B, L, L)
    mask = torch.zeros((M, 8), dtype=torch.int32, device=dev)
    _lib.call("pzn_attn_fused_fwd", 1, P([x]), P([qrp]), P([krp]), P([vrp]), P([W]), P([bo]), B, P([r]), P([t]), P([mask]),
              P([amap]), P([lse]), 0, 1.0, st)
    dz, u, dq, delta, dqt = mk(M, E), mk(M, E), mk(M, dk), mk(M), mk(M, dk)
    darp = raw(vb)
    _lib.call("pzn_attn_fused_bwd_q", 1, P([dr]), E, None, E, P([mask]), P([qrp]), P([krp]), P([vrp]), P([W]), B, P([dz]), P([u]),
              P([dq]), P([dqt]), P([darp]), P([delta]), st)
    dkk, dvv, dx = mk(M, dk), mk(M, E), mk(M, E)
    _lib.call("pzn_attn_fused_bwd_k", 1, P([qrp]), P([krp]), P([vrp]), P([darp]), P([W]), P([lse]),
              P([delta]), P([u]), P([dqt]), B, P([dkk]), P([dvv]), P([dx]), st)
    untile = ops.attention_tile_image_rows
    got = dict(r=r, t=t, map=amap, lse=lse, dz=dz, delta=delta, dq=dq, u=untile(u, E), dk=dkk, dv=dvv, dx=dx)
    args = (x.view(B, L, E), wq, bq, wk, bk, wv, bv, wo, bo, dr.view(B, L, E))
    spec, exact = T._attn_block_bf16_spec(*args), T._attn_block_ref64(*args)
    flips = ((dz != 0) != (spec["dz"].reshape(M, E) != 0))
    print(f"gate flips vs the statement: {int(flips.sum())} of {M * E}; |z| of the statement at the flips: max "
          f"{float(spec['z'].reshape(M, E)[flips].abs().max()) if int(flips.sum()) else 0:.2e}")
    for name, val in got.items():
        print(f"{name:6s} vs statement {T._l2(val, spec[name].reshape(val.shape)):.3e}   vs exact float64 {T._l2(val, exact[name].reshape(val.shape)):.3e}")


if __name__ == "__main__":
    main()
