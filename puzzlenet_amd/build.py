"""Ahead-of-time build of libpzn.so (hipcc, gfx950 only).

    python -m puzzlenet_amd.build [--force]

The shared object is written next to this file (puzzlenet_amd/libpzn.so) so it
travels with the repo snapshot to the GPU box; it is git-ignored.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libpzn.so")

# (source, extra flags).  The point-op kernels use explicitly rounded
# arithmetic (__f*_rn) where bit-exactness matters; -ffp-contract=off on those
# files is belt and braces.
SOURCES = [
    ("core.hip", []),
    ("fps.hip", ["-ffp-contract=off"]),
    ("knn.hip", ["-ffp-contract=off"]),
    ("group.hip", ["-ffp-contract=off"]),
    ("emd.hip", []),
    ("chamfer.hip", []),
    ("gemm.hip", []),
    ("poolbwd.hip", []),
    ("maxptsbwd.hip", []),
    ("wsgemm.hip", []),
    ("dfgemm.hip", []),
    ("optim.hip", ["-ffp-contract=off"]),
    ("se3.hip", []),
    ("sapoint.hip", []),
    ("bnpoints.hip", []),
    ("losstail.hip", []),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden",
          "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "pzn.h"))
    objs = []
    procs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc()] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
