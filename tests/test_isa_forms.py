"""No kernel may contain the packed-fp32 forms that return wrong results beside the matrix-core engine.

tools/pk_probe.hip pins the fault down to `v_pk_fma_f32` / `v_pk_mul_f32` whose op_sel bit of
src1 is set (the LOW result reads the HIGH half of src1): wrong low results in lanes 48-63, only while a workgroup of
gemm_kernel shares the CU; every other form, and op_sel on src0 / src2, is clean.  The compiler produces that form when
it pairs scalar fp32 operations (SLP) of the shape "two weights x one coordinate"; build.py disables the pairing for the
files concerned.  This test compiles every source to gfx950 assembly with the build's own flags and scans it."""
import os
import re
import subprocess
import tempfile

from puzzlenet_amd import build

FORM = re.compile(r"v_pk_(?:fma|mul)_f32[^\n]*op_sel:\[[01],1")


def test_no_src1_op_sel_packed_multiplies():
    hipcc = build.hipcc()
    flags = [f for f in build.COMMON if f not in ("-fPIC", "-fvisibility=hidden")]
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for src, extra in build.SOURCES:
            out = os.path.join(tmp, src + ".s")
            cmd = [hipcc] + flags + extra + ["-S", "--cuda-device-only", "-o", out, os.path.join(build.CSRC, src)]
            procs.append((src, out, subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
        bad = {}
        for src, out, p in procs:
            assert p.wait() == 0, f"{src} did not compile to assembly"
            hits = FORM.findall(open(out).read())
            if hits:
                bad[src] = len(hits)
    assert not bad, f"packed multiplies with src1 op_sel (wrong beside gemm_kernel, see tools/pk_probe.hip): {bad}"
