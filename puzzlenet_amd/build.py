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
# NOSLP: no SLP vectorisation.  Under plain -O3 the compiler pairs adjacent scalar fp32 operations into v_pk_*_f32 with
# op_sel modifiers; that form (a) gave wrong sums in the set-abstraction prep kernel while workgroups of the matrix-core
# engine shared the CU (two-stream training step; tests/test_gpu_concurrency.py), and (b) costs extra cycles beside MFMAs
# (MI355X_MICROARCH.md, "packed f32 VALU").  Explicitly written packed math (knn.hip, emd.hip, chamfer.hip: float2
# vector types) is not touched by the flag.  wsgemm / dfgemm / poolbwd keep the default: their code has no op_sel
# packed forms (checked in the ISA) and the generated-row max-pool kernel measured 3-5 % slower without the pairing.
NOSLP = ["-fno-slp-vectorize"]
SOURCES = [
    ("core.hip", NOSLP),
    ("fps.hip", ["-ffp-contract=off"] + NOSLP),
    ("knn.hip", ["-ffp-contract=off"]),
    ("group.hip", ["-ffp-contract=off"] + NOSLP),
    ("emd.hip", NOSLP),
    ("emd64.hip", NOSLP),
    ("chamfer.hip", NOSLP),
    ("gemm.hip", NOSLP),
    ("attnfused.hip", NOSLP),
    ("attnchain.hip", NOSLP),
    ("salevel.hip", NOSLP),
    ("outproj.hip", NOSLP),
    ("pointmlp.hip", NOSLP),
    ("poolbwd.hip", []),
    ("maxptsbwd.hip", NOSLP),
    ("wsgemm.hip", []),
    ("dfgemm.hip", []),
    ("attnwgrad.hip", NOSLP),
    ("optim.hip", ["-ffp-contract=off"] + NOSLP),
    ("se3.hip", NOSLP),
    ("sapoint.hip", NOSLP),
    ("sapool.hip", NOSLP),
    ("sachain.hip", NOSLP),
    ("bnpoints.hip", NOSLP),
    ("stem.hip", NOSLP),
    ("losstail.hip", NOSLP),
    ("datapipe.hip", ["-ffp-contract=off"] + NOSLP),
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
