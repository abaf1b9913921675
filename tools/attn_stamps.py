#!/usr/bin/env python3
"""Where the cycles of the chained attention kernels go: phase stamps (s_memtime) of wavefront 0 of two workgroups.

    python tools/attn_stamps.py build        # here (cross-compiles csrc/attnfused.hip with -DATTN_STAMPS into
                                             #  puzzlenet_amd/libpzn_stamps.so; the other objects are the product's)
    python tools/attn_stamps.py run [B ...]  # on the GPU box: launch times per kernel and B, stamps at the last B

s_memtime ticks at 100 MHz on gfx950 is NOT assumed: the tool reports ticks and the tick rate it measures against the
launch's event time."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "puzzlenet_amd")
STAMP_LIB = os.path.join(PKG, "libpzn_diag.so")


def build(extra=(), suffix=""):
    """extra: more -D switches (timing experiments of pzn_mfma.h: -DPZN_EXP_NODMA, -DPZN_EXP_NOSPLIT), suffix names the library"""
    from puzzlenet_amd import build as pb
    pb.build()
    os.makedirs(os.path.join(PKG, "_obj_stamps"), exist_ok=True)
    stamped = ("attnfused.hip",)
    objs = [os.path.join(pb.OBJ, s.replace(".hip", ".o")) for s, _ in pb.SOURCES if s not in stamped]
    for src in stamped:
        o = os.path.join(PKG, "_obj_stamps", src.replace(".hip", f"{suffix}.o"))
        flags = dict(pb.SOURCES)[src]
        subprocess.check_call([pb.hipcc()] + pb.COMMON + flags + ["-DATTN_STAMPS", *extra, "-c", os.path.join(pb.CSRC, src), "-o", o])
        objs.append(o)
    lib = STAMP_LIB.replace(".so", f"{suffix}.so")
    subprocess.check_call([pb.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


# stamp index -> what ran since the previous stamp that exists
LABELS = {
    0: {1: "load x", 2: "qkv loop (16 k-steps)", 3: "q, k images", 4: "v image"},
    1: {1: "q frags + S loop", 2: "softmax", 3: "map store", 4: "PV loop", 5: "x load, t = x - a", 6: "t store + Wo loop",
        7: "epilogue r store"},
    2: {1: "load dr(+dr2), gate bits", 2: "store dz", 3: "dt loop", 4: "step_sync", 5: "u = dr + dt store", 6: "da image",
        9: "DP loop", 10: "S loop", 11: "softmax + dS", 12: "dq loop", 13: "store dq"},
    3: {1: "row consts, first slabs", 2: "S loop", 3: "P = exp", 4: "dP loop", 5: "dS + dk loop", 6: "dv loop",
        7: "store dk dv, load dq u", 8: "dx loop (24)", 9: "store dx"},
}
NAMES = {0: "proj", 1: "fwd", 2: "bwd_q", 3: "bwd_k"}


def run(Bs):
    import torch
    from puzzlenet_amd import _lib, ops
    _lib.LIB_PATH = os.environ.get("PZN_STAMP_LIB", STAMP_LIB)
    lib = _lib.load()
    rd = lib.pzn_attn_fused_read_stamps
    rd.restype = ctypes.c_int
    rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    L, E, dk = 256, 256, 64
    P = ops._ptrs
    st = torch.cuda.current_stream().cuda_stream
    for B in Bs:
        M = B * L
        g = torch.Generator().manual_seed(3)
        x = (0.5 * torch.randn(M, E, generator=g)).to(dev)
        wq, wk = [(torch.randn(dk, E, generator=g) / 16).to(dev) for _ in range(2)]
        wv, wo = [(torch.randn(E, E, generator=g) / 16).to(dev) for _ in range(2)]
        bq, bk = [(torch.randn(dk, generator=g) / 4).to(dev) for _ in range(2)]
        bv, bo = [(torch.randn(E, generator=g) / 4).to(dev) for _ in range(2)]
        dr = torch.randn(M, E, generator=g).to(dev)
        dr2 = torch.randn(M, E, generator=g).to(dev)
        raw = lambda n: torch.zeros(n, dtype=torch.uint8, device=dev)
        mk = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        W = raw(lib.pzn_attn_fused_weight_bytes())
        _lib.call("pzn_attn_fused_prep_weights", wq.data_ptr(), wk.data_ptr(), wv.data_ptr(), wo.data_ptr(), W.data_ptr(), st)
        qkb, vb = lib.pzn_attn_fused_qk_image_bytes(B), lib.pzn_attn_fused_v_image_bytes(B)
        qrp, krp, vrp = raw(qkb), raw(qkb), raw(vb)
        r, t, lse, amap = mk(M, E), mk(M, E), mk(M), mk(B, L, L)
        mask = torch.zeros((M, 8), dtype=torch.int32, device=dev)
        dz, u, dq, delta = mk(M, E), mk(M, E), mk(M, dk), mk(M)
        dqt = mk(M, dk)
        darp = raw(vb)
        dkk, dvv, dx = mk(M, dk), mk(M, E), mk(M, E)
        calls = {
            0: lambda: _lib.call("pzn_attn_fused_proj", 1, P([x]), P([W]), P([bq]), P([bk]), P([bv]), B, P([qrp]), P([krp]), P([vrp]), st),
            1: lambda: _lib.call("pzn_attn_fused_fwd", 1, P([x]), P([qrp]), P([krp]), P([vrp]), P([W]), P([bo]), B, P([r]), P([t]),
                                 P([mask]), P([amap]), P([lse]), 1, 0.25, st),
            2: lambda: _lib.call("pzn_attn_fused_bwd_q", 1, P([dr]), E, P([dr2]), E, P([mask]), P([qrp]), P([krp]),
                                 P([vrp]), P([W]), B, P([dz]), P([u]), P([dq]), P([dqt]), P([darp]), P([delta]), st),
            3: lambda: _lib.call("pzn_attn_fused_bwd_k", 1, P([qrp]), P([krp]), P([vrp]), P([darp]), P([W]),
                                 P([lse]), P([delta]), P([u]), P([dqt]), B, P([dkk]), P([dvv]), P([dx]), st),
        }
        big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        times = {}
        for k in range(4):
            for _ in range(3):
                calls[k]()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                calls[k]()
            ev[1].record()
            torch.cuda.synchronize()
            warm = ev[0].elapsed_time(ev[1]) / 10 * 1e3
            cold = 0.0
            for _ in range(3):
                big.fill_(1)
                ev[0].record()
                calls[k]()
                ev[1].record()
                torch.cuda.synchronize()
                cold += ev[0].elapsed_time(ev[1]) / 3 * 1e3
            times[k] = (warm, cold)
        print(f"B={B:3d} ({2 * B} workgroups): " + "  ".join(f"{NAMES[k]} {times[k][0]:.1f} us (cold {times[k][1]:.1f})" for k in range(4)))
    # stamps at the last B (warm: three launches before)
    host = (ctypes.c_longlong * (4 * 2 * 64))()
    rd(host, 1)
    for k in range(4):
        calls[k]()
        torch.cuda.synchronize()
    rd(host, 0)
    for k in range(4):
        for wg in range(2):
            v = [host[(k * 2 + wg) * 64 + i] for i in range(64)]
            idx = [0] + sorted(LABELS[k])
            if v[idx[-1]] == 0:
                continue
            tot = v[idx[-1]] - v[0]
            print(f"{NAMES[k]} workgroup {'0' if wg == 0 else '77'}: total {tot} ticks ({times[k][0]:.1f} us launch)")
            for a_, b_ in zip(idx[:-1], idx[1:]):
                print(f"    {LABELS[k][b_]:34s} {v[b_] - v[a_]:8d}  {100.0 * (v[b_] - v[a_]) / tot:5.1f} %")
            if k == 1:
                print("    PV steps (sync, compute) k-steps 4..7:", " ".join(str(v[i + 1] - v[i]) for i in range(16, 23)))
                print("    Wo steps (sync, compute) k-steps 4..7:", " ".join(str(v[i + 1] - v[i]) for i in range(8, 15)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build(tuple(a for a in sys.argv[2:] if a.startswith("-D")), "".join(a for a in sys.argv[2:] if not a.startswith("-D")))
    else:
        run([int(a) for a in sys.argv[2:]] or [8, 16, 32, 64])
