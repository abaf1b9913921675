"""CPU: libpzn.so builds (hipcc cross-compile, no GPU needed), loads, and exports
exactly the entry points include/pzn.h declares.  No compute call is made here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pzn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pzn_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    from puzzlenet_amd import build
    return build.build()


def test_header_declares_entry_points():
    names = _declared()
    assert "pzn_fps_f32" in names and "pzn_knn_f32" in names and "pzn_emd_fused_f32" in names
    assert len(names) >= 15


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_matches_header(lib_path):
    from puzzlenet_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.pzn_version() >= 100
    assert lib.pzn_strerror(-1).decode().startswith("invalid")
    assert lib.pzn_emd_workspace_bytes(2, 8, 8) > 0


def test_no_hidden_symbols_leak(lib_path):
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path]).decode()
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    ours = [s for s in exported if not s.startswith("_")]
    assert sorted(ours) == _declared()


def test_cpu_tensor_is_rejected_loudly():
    import torch
    from puzzlenet_amd import _lib, ops
    with pytest.raises(_lib.PznError):
        ops.knn(torch.zeros(1, 8, 3), torch.zeros(1, 2, 3), 2)


def test_product_does_not_import_oracle():
    """The product package must never route through the CPU oracle."""
    pkg = os.path.join(ROOT, "puzzlenet_amd")
    bad = []
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "pzn_oracle" in txt:
                    bad.append(f)
    assert not bad, bad


def test_integration_md_names_every_exported_symbol():
    """INTEGRATION.md is the index a maintainer binds from: every entry point include/pzn.h declares appears there (as its full
    name, inside a `pzn_x_{a,b}_f32` brace group, or as a ` / suffix` alternative of a named one)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "pzn.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    syms = set(re.findall(r"\b(pzn_[a-z0-9_]+)\s*\(", header))
    names = set(re.findall(r"pzn_[a-z0-9_]+", doc))
    for m in re.finditer(r"(pzn_[a-z0-9_]*)\{([^}]*)\}([a-z0-9_]*)", doc):
        names.update(m.group(1) + alt.strip() + m.group(3) for alt in m.group(2).split(","))
    for m in re.finditer(r"(pzn_[a-z0-9_]+)((?:\s*/\s*_?[a-z0-9_]+)+)", doc):
        toks = m.group(1).split("_")
        for alt in (q.strip().lstrip("_") for q in m.group(2).split("/") if q.strip()):
            names.update("_".join(toks[:k]) + "_" + alt for k in range(1, len(toks)))
    missing = sorted(syms - names)
    assert not missing, f"entry points of include/pzn.h that INTEGRATION.md does not name: {missing}"
