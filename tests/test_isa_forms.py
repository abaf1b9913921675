"""No kernel may contain the packed-fp32 forms that return wrong results beside a matrix-core kernel.

tools/pk_probe.hip and tools/pk_probe2.hip pin the fault down to a PAIR of instructions on one CU:
a `v_pk_fma_f32` / `v_pk_mul_f32` / `v_pk_add_f32` whose op_sel bit of src1 is set while that of src0 is clear (the LOW
result reads the HIGH half of src1) returns a wrong low result in lanes 48-63 while another wavefront issues MFMAs whose
accumulators live in AGPRs (`v_mfma_* a[..]`: gemm_kernel, the generated-row kernel, the chained attention kernels).
MFMAs on VGPR accumulators, every other instruction class, and op_sel on src0 / src2 are clean.  The compiler produces
the form when it pairs scalar fp32 operations (SLP) of the shape "two weights x one coordinate".  This test compiles
EVERY source to gfx950 assembly with the build's own flags and fails on any packed fp32 instruction with the src1
op_sel bit, whatever src0 says."""
import os
import re
import subprocess
import tempfile

from puzzlenet_amd import build

FORM = re.compile(r"v_pk_(?:fma|mul|add)_f32[^\n]*op_sel:\[[01],1")


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
    assert not bad, f"packed fp32 instructions with src1 op_sel (wrong beside AGPR-accumulator MFMAs, see tools/pk_probe2.hip): {bad}"
