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


ASM_LOAD = re.compile(r"^global_load_dword(?:x[24])?\s+(v\[\d+:\d+\]|v\d+)")
VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def _vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def test_no_use_of_in_flight_asm_load_registers():
    """salevel.hip and outproj.hip issue their row loads as inline asm and covers them with the step's own `s_waitcnt vmcnt` (also asm):
    the compiler does not know the destination registers are written LATER.  Under register pressure it has spilled such
    a register right behind the load (the mixed-shape instantiations, not built: DESIGN.md section 4) - stale data, no
    diagnostic.  This test walks every instantiated streamed kernel in text order and fails when any instruction between
    an asm load and the next asm vmcnt wait touches one of the load's destination registers."""
    for source, kernel in (("salevel.hip", "sa_level_stream_kernel"), ("outproj.hip", "outproj_maxpts_kernel")):
        _check_asm_loads(source, kernel)


def _check_asm_loads(source, kernel):
    hipcc = build.hipcc()
    flags = [f for f in build.COMMON if f not in ("-fPIC", "-fvisibility=hidden")]
    extra = dict(build.SOURCES)[source]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, source + ".s")
        cmd = [hipcc] + flags + extra + ["-S", "--cuda-device-only", "-o", out, os.path.join(build.CSRC, source)]
        assert subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 0
        text = open(out).read()
    kernels = re.split(r"\n(?=_ZN\S*" + kernel + r"\S*:)", text)[1:]
    assert kernels, "no streamed kernel in the assembly"
    for k in kernels:
        name = k.split(":")[0]
        body = k[:k.index("s_endpgm")]
        in_asm, in_flight, bad, loads = False, set(), [], 0
        for line in body.split("\n"):
            t = line.strip()
            if "ASMSTART" in t:
                in_asm = True
                continue
            if "ASMEND" in t:
                in_asm = False
                continue
            if not t or t[0] in ";.":
                continue
            m = ASM_LOAD.match(t) if in_asm else None
            if m:
                in_flight |= _vregs(m.group(1))
                loads += 1
                continue
            if t.startswith("s_waitcnt vmcnt") and (in_asm or t.startswith("s_waitcnt vmcnt(0)")):
                in_flight.clear()
                continue
            hit = _vregs(t) & in_flight
            if hit:
                bad.append((t, sorted(hit)))
        assert loads > 0, name
        assert not bad, f"{name}: in-flight asm-load registers touched before their wait: {bad[:4]}"
        # the loads of the LAST step are never consumed: they must be drained before the final epilogue reuses their
        # registers (the text-order walk above cannot see that: the compiler places the epilogue in front of the loop)
        assert "pzn_drain" in body, f"{name}: no vmcnt(0) drain between the last asm loads and the epilogue"


ASM_DS = re.compile(r"^ds_read\w*\s+(v\[\d+:\d+\]|v\d+)")
LGKM = re.compile(r"^s_waitcnt\s+(?:vmcnt\(\d+\)\s+)?lgkmcnt\((\d+)\)")


def _lds_in_flight_violations(source):
    """Same walk for the asm LDS reads (pzn_mfma.h: RP_ISSUE / TR_ISSUE with counted lgkmcnt waits; LDS returns in order,
    so `lgkmcnt(N)` leaves the N youngest reads in flight)."""
    hipcc = build.hipcc()
    flags = [f for f in build.COMMON if f not in ("-fPIC", "-fvisibility=hidden")]
    extra = dict(build.SOURCES)[source]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, source + ".s")
        cmd = [hipcc] + flags + extra + ["-S", "--cuda-device-only", "-o", out, os.path.join(build.CSRC, source)]
        assert subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 0
        text = open(out).read()
    bad, reads = {}, 0
    for k in re.split(r"\n(?=_Z\S+:\s)", text)[1:]:
        name = k.split(":")[0]
        if "s_endpgm" not in k:
            continue
        in_asm, queue = False, []                       # queue: destination register sets of asm reads, oldest first
        for line in k[:k.index("s_endpgm")].split("\n"):
            t = line.strip()
            if "ASMSTART" in t:
                in_asm = True
                continue
            if "ASMEND" in t:
                in_asm = False
                continue
            if not t or t[0] in ";.":
                continue
            m = ASM_DS.match(t) if in_asm else None
            if m:
                queue.append(_vregs(m.group(1)))
                reads += 1
                continue
            w = LGKM.match(t)
            if w:
                n = int(w.group(1))
                queue = queue[len(queue) - n:] if n else []
                continue
            if t.startswith("s_waitcnt") and "lgkmcnt" not in t:
                continue
            flying = set().union(*queue) if queue else set()
            hit = _vregs(t) & flying
            if hit and not t.startswith("ds_read"):      # (a compiler-issued LDS read of its own does not touch them)
                bad.setdefault(name, []).append((t, sorted(hit)))
    return reads, bad


def test_no_use_of_in_flight_asm_lds_read_registers():
    for source in ("attnfused.hip", "salevel.hip", "outproj.hip"):
        reads, bad = _lds_in_flight_violations(source)
        assert reads > 0, source
        assert not bad, f"{source}: registers of asm LDS reads touched before their counted wait: " \
                        f"{ {k: v[:3] for k, v in bad.items()} }"


SPILL_FREE = {   # source -> kernels (substring of the mangled name) that must not spill a single register
    # (ILi3E: the three-plane instantiations, i.e. the default fp32-result path; ILi1E: the single-plane instantiations of
    #  the opt-in bf16 attention mode)
    "attnfused.hip": ("attn_proj_kernelILi3E", "attn_fwd_kernelILi3E", "attn_bwd_q_kernelILi3E", "attn_bwd_k_kernelILi3E",
                      "attn_proj_kernelILi1E", "attn_fwd_kernelILi1E", "attn_bwd_q_kernelILi1E", "attn_bwd_k_kernelILi1E"),
    "salevel.hip": ("sa_level_stream_kernel",),
    "outproj.hip": ("outproj_maxpts_kernel",),
    "pointmlp.hip": ("point_mlp3_fwd_kernel", "point_mlp3_bwd_kernel"),
    # the per-point stem (round 5): its two-workgroups-per-CU occupancy rests on 128 / 256 registers without scratch
    "stem.hip": ("stem_fwd_kernel", "stem_bwd_kernel"),
}

# register ceilings that an occupancy argument in the source rests on: kernel -> architected + accumulation registers
REGISTER_CAPS = {"stem.hip": {"stem_fwd_kernel": 128, "stem_bwd_kernel": 256}}


def test_matrix_core_kernels_do_not_spill():
    """The chained matrix-core kernels run on (nearly) their whole register budget (the attention kernels: 256 registers,
    two wavefronts per SIMD); a spilled register is a scratch access in the vmcnt stream their counted waits are written
    against.  Round 3's attention backward spilled 294 / 107 registers (996 / 368 bytes of scratch per lane): this test
    reads `.vgpr_spill_count` and `.private_segment_fixed_size` from the code object metadata of every such kernel."""
    hipcc = build.hipcc()
    flags = [f for f in build.COMMON if f not in ("-fPIC", "-fvisibility=hidden")]
    for source, kernels in SPILL_FREE.items():
        extra = dict(build.SOURCES)[source]
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, source + ".s")
            cmd = [hipcc] + flags + extra + ["-S", "--cuda-device-only", "-o", out, os.path.join(build.CSRC, source)]
            assert subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 0
            text = open(out).read()
        meta = text[text.index("amdhsa.kernels:"):]
        seen = set()
        for entry in re.split(r"\n  - ", meta)[1:]:
            m = re.search(r"\.name:\s+(\S+)", entry)
            which = [k for k in kernels if m and k in m.group(1)]
            if not which:
                continue
            name = m.group(1)
            seen.add(which[0])
            spills = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", entry).group(1))
            scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", entry).group(1))
            assert spills == 0 and scratch == 0, f"{name}: {spills} spilled registers, {scratch} bytes of scratch per lane"
            cap = REGISTER_CAPS.get(source, {}).get(which[0])
            if cap is not None:
                regs = int(re.search(r"\.vgpr_count:\s+(\d+)", entry).group(1)) + int(re.search(r"\.agpr_count:\s+(\d+)", entry).group(1))
                assert regs <= cap, f"{name}: {regs} registers, the occupancy it is launched for needs <= {cap}"
        assert seen == set(kernels), (source, seen)
