#!/usr/bin/env python3
"""Where the cycles of the sparse pooled backward go (csrc/poolbwd.hip: pool_wgrad_kernel, pool_dgrad_kernel behind
pzn_sa_level_bwd_rm_f32): phase sums (s_memtime) of wavefront 0 of every workgroup, on the launches of a REAL training step
(B = 64, N = 2048: the arg-max rows and ReLU gates are the model's), from a diagnostic build (-DPOOL_STAMPS).

    python tools/pool_stamps.py build      # here: puzzlenet_amd/libpzn_stamps_pool.so
    python tools/pool_stamps.py run        # on the GPU box
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "puzzlenet_amd")
STAMP_LIB = os.path.join(PKG, "libpzn_diag.so")

WG_PHASES = ["loop", "rows landed + tile write", "barrier", "issue next loads", "hit loop", "atomics"]
DG_PHASES = ["(arg-max, gradient) vectors landed", "row mask + group setup", "gate rows requested", "hit loops",
             "gates landed + mask + stores", "W slice into LDS (once)"]


def build(extra=()):
    from puzzlenet_amd import build as pb
    pb.build()
    os.makedirs(os.path.join(PKG, "_obj_stamps"), exist_ok=True)
    src = "poolbwd.hip"
    objs = [os.path.join(pb.OBJ, s.replace(".hip", ".o")) for s, _ in pb.SOURCES if s != src]
    o = os.path.join(PKG, "_obj_stamps", "poolbwd_stamps.o")
    subprocess.check_call([pb.hipcc()] + pb.COMMON + dict(pb.SOURCES)[src] + ["-DPOOL_STAMPS", *extra, "-c",
                                                                            os.path.join(pb.CSRC, src), "-o", o])
    subprocess.check_call([pb.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", STAMP_LIB] + objs + [o])
    print(STAMP_LIB)


def run():
    import numpy as np
    import torch
    from puzzlenet_amd import _lib
    _lib.LIB_PATH = os.environ.get("PZN_STAMP_LIB", STAMP_LIB)
    from puzzlenet_amd import engine, model5_b, synthetic
    import bench
    lib = _lib.load()
    dev = torch.device("cuda:0")
    cfg = bench.Cfg()
    cfg.num_points = 2048
    torch.manual_seed(0)
    model = model5_b.TouchedRegraster(cfg).to(dev)
    batch = synthetic.make_batch(64, 2048, dev, seed=1234)
    torch.manual_seed(1000)
    runner = engine.TrainStep(model, batch, cfg.lr, world=1)
    for _ in range(5):
        runner.step()
    torch.cuda.synchronize()
    buf = torch.zeros(16 * 1024 * 8, dtype=torch.int64, device=dev)
    lib.pzn_pool_bwd_set_stamps.restype = None
    lib.pzn_pool_bwd_set_stamps.argtypes = [ctypes.c_void_p]
    lib.pzn_pool_bwd_stamp_log.restype = ctypes.c_int
    lib.pzn_pool_bwd_stamp_log.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    lib.pzn_pool_bwd_set_stamps(buf.data_ptr())
    runner.step()
    torch.cuda.synchronize()
    lib.pzn_pool_bwd_set_stamps(None)
    raw = buf.cpu().numpy().reshape(16, 1024, 8)
    out4 = (ctypes.c_int * 4)()
    n = lib.pzn_pool_bwd_stamp_log(0, out4)
    print(f"{n} launches stamped in one step (two streams); cycles are core clocks of wavefront 0, medians over the workgroups")
    for slot in range(min(n, 16)):
        lib.pzn_pool_bwd_stamp_log(slot, out4)
        kind, G, C1, C2 = list(out4)
        r = raw[slot]
        ok = r[:, 7] > 0
        r = r[ok].astype(np.float64)
        names = DG_PHASES if kind else WG_PHASES
        groups = np.median(r[:, 6])
        tot = np.median(r[:, 7])
        print("%s  G=%d C1=%d C2=%d: %d workgroups, %.0f groups per wavefront-0, total %.1f kcyc (max %.1f)" % (
            "pool_dgrad_kernel" if kind else "pool_wgrad_kernel", G, C1, C2, int(ok.sum()), groups, tot / 1e3, r[:, 7].max() / 1e3))
        for i, nm in enumerate(names):
            v = np.median(r[:, i])
            print("    %-44s %8.1f kcyc  %5.1f %%   %7.0f cycles per group" % (nm, v / 1e3, 100 * v / tot, v / max(groups, 1)))
    runner.close()


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run()
