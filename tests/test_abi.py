"""CPU: the C-ABI library builds, loads and exports exactly the symbols include/givepose_hip.h declares."""
import os
import re
import subprocess

from givepose_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "givepose_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(gp_[a-z0-9_]+)\s*\(", src))


def test_library_builds_and_exports_header_symbols():
    build.build(verbose=False)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (gp_[a-z0-9_]+)", out))
    declared = _header_symbols()
    assert declared, "no declarations parsed"
    assert declared <= exported, f"declared but not exported: {declared - exported}"
    assert exported <= declared, f"exported but not declared: {exported - declared}"


def test_ctypes_prototypes_cover_header():
    assert set(_lib.PROTOTYPES) == _header_symbols()
    lib = _lib.load()
    assert lib.gp_version() >= 100


def test_code_object_targets_gfx950():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", _lib.LIB_PATH], capture_output=True, text=True)
    assert "gfx950" in out.stdout + out.stderr


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "givepose_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", s, flags=re.M), f
