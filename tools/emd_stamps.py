#!/usr/bin/env python3
"""When do the workgroups of the fused EMD's launches run, and where: per (launch, workgroup) start / end stamps
(s_memtime) with HW_ID / XCC_ID, from a diagnostic build of csrc/emd.hip (-DEMD_STAMPS).

    python tools/emd_stamps.py build              # here: puzzlenet_amd/libpzn_stamps.so
    python tools/emd_stamps.py run [B n regime]   # on the GPU box

Per launch: workgroups that did work, their duration (median / max), when the first and the last one started and ended
relative to the launch's first start, and the average / maximum number of working workgroups resident per CU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "puzzlenet_amd")
STAMP_LIB = os.path.join(PKG, "libpzn_diag.so")


def build():
    from puzzlenet_amd import build as pb
    pb.build()
    os.makedirs(os.path.join(PKG, "_obj_stamps"), exist_ok=True)
    src = "emd.hip"
    objs = [os.path.join(pb.OBJ, s.replace(".hip", ".o")) for s, _ in pb.SOURCES if s != src]
    o = os.path.join(PKG, "_obj_stamps", "emd_stamps.o")
    subprocess.check_call([pb.hipcc()] + pb.COMMON + dict(pb.SOURCES)[src] + ["-DEMD_STAMPS", "-c", os.path.join(pb.CSRC, src), "-o", o])
    subprocess.check_call([pb.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", STAMP_LIB] + objs + [o])
    print(STAMP_LIB)


def run(B, n, regime):
    import numpy as np
    import torch
    from puzzlenet_amd import _lib
    _lib.LIB_PATH = os.environ.get("PZN_STAMP_LIB", STAMP_LIB)
    from puzzlenet_amd import ops, se3
    from tools.emd_levels import launch_names
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    a = torch.rand(B, n, 3, generator=g).to(dev)
    if regime == "indep":
        b = torch.rand(B, n, 3, generator=g).to(dev)
    else:
        x = torch.randn(B, 6, generator=g)
        x = (0.8 if regime == "rigid" else 0.08) * x / x.norm(dim=1, keepdim=True)
        b = se3.transform(se3.exp(x.to(dev)), a.permute(0, 2, 1)).permute(0, 2, 1).contiguous()
    lib = _lib.load()
    nbytes = lib.pzn_emd_workspace_bytes(B, n, n)
    WGS, LIDS = 4096, 21
    sb = 8 * 8 * WGS * LIDS
    ws = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=dev)
    cost = torch.empty(B, device=dev)
    g1 = torch.empty(B, n, 3, device=dev)
    g2 = torch.empty(B, n, 3, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ws.zero_()
        torch.cuda.synchronize()
        ev0.record()
        _lib.call("pzn_emd_fused_f32", b.data_ptr(), a.data_ptr(), B, n, n, cost.data_ptr(), g1.data_ptr(), g2.data_ptr(),
                  ws.data_ptr(), st)
        ev1.record()
        torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)
    raw = ws.view(torch.uint8)[nbytes - sb: nbytes].cpu().numpy().view(np.uint64).reshape(LIDS, WGS, 8)
    names = launch_names()
    # s_memtime counts core clocks and is NOT synchronised between CUs: everything below is per CU, in kilo-cycles
    t0a, t1a = raw[..., 0].astype(np.int64), raw[..., 1].astype(np.int64)
    hwa, xa = raw[..., 5].astype(np.int64), raw[..., 6].astype(np.int64) & 0xF
    t1a = raw[..., 4].astype(np.int64)
    pha = raw[..., 0:5].astype(np.int64)
    oka = t0a > 0
    print(f"{regime} B={B} n={n}: call {ms * 1e3:.1f} us")
    print("%-8s %6s %9s %9s | per CU: %9s %9s %9s %9s" % ("launch", "wgs", "med kcyc", "max kcyc", "span med", "span max",
                                                          "avg conc", "max conc"))
    for lid in range(LIDS):
        ok = oka[lid]
        if not ok.any():
            continue
        t0, t1, hw, xcc = t0a[lid][ok], t1a[lid][ok], hwa[lid][ok], xa[lid][ok]
        dur = (t1 - t0) / 1e3
        work = dur > 0.25 * np.median(dur[dur >= np.percentile(dur, 60)])      # not the workgroups that leave at once
        cu = ((xcc << 8) | ((hw >> 8) & 0xFF))                                  # XCC | SE_ID, SH_ID, CU_ID of HW_ID
        spans, avg, mx, example = [], [], [], None
        for c in np.unique(cu[work]):
            sel = (cu == c) & work
            s0, s1 = t0[sel], t1[sel]
            base = t0[cu == c].min()
            span = s1.max() - base
            spans.append(span / 1e3)
            avg.append((s1 - s0).sum() / span)
            ev = sorted([(t, 1) for t in s0] + [(t, -1) for t in s1])
            cur = m = 0
            for _, d in ev:
                cur += d
                m = max(m, cur)
            mx.append(m)
            if example is None and len(s0) >= 8:
                o = np.argsort(s0)
                example = " ".join("%d-%d" % ((a_ - base) // 1000, (b_ - base) // 1000) for a_, b_ in zip(s0[o], s1[o]))
        print("%-8s %6d %9.1f %9.1f |         %9.1f %9.1f %9.2f %9d" % (
            names[lid], int(work.sum()), np.median(dur[work]), dur[work].max(), np.median(spans), max(spans),
            float(np.mean(avg)), int(max(mx))))
        ph = pha[lid][ok][work]
        if (ph[:, 1] > 0).all():
            d = np.diff(ph, axis=1) / 1e3
            print("         phases med kcyc: stage %.1f  walk %.1f  sums %.1f  epilogue %.1f" % tuple(np.median(d, axis=0)))
        if lid in (8, 9) and example:
            print("         one CU, start-end kcyc of its working workgroups:", example)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        a = sys.argv[2:]
        run(int(a[0]) if a else 64, int(a[1]) if len(a) > 1 else 2048, a[2] if len(a) > 2 else "rigid")
