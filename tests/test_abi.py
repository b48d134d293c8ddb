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


def _kernel_notes(tmp_path):
    """Extract the gfx950 code objects of the library (into tmp_path: llvm-objdump writes next to its input) and
    return {kernel name: {metadata key: int}} from their ELF notes."""
    import shutil
    lib = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", str(lib)], capture_output=True, text=True)
    assert "gfx950" in out.stdout + out.stderr
    kernels = {}
    for f in sorted(tmp_path.glob("lib.so.*gfx950")):
        notes = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", str(f)], text=True)
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            kernels[name] = {k: int(v) for k, v in re.findall(r"\.(private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|"
                                                              r"group_segment_fixed_size|vgpr_count):\s+(\d+)", blk)}
    return kernels


def test_code_object_targets_gfx950_and_no_kernel_uses_scratch(tmp_path):
    """Every kernel of the library must fit its registers: a kernel with a private (scratch) segment is banned from
    the product (DESIGN.md 6b -- spilled kernels on concurrent streams were the ingredient of the round-1 in-flight
    corruption, and compiler-issued scratch traffic sits inside hand-counted vmcnt windows)."""
    kernels = _kernel_notes(tmp_path)
    assert len(kernels) > 50, sorted(kernels)
    bad = {n: k for n, k in kernels.items() if k["private_segment_fixed_size"] or k["vgpr_spill_count"] or k["sgpr_spill_count"]}
    assert not bad, bad
    assert all(k["group_segment_fixed_size"] <= 160 * 1024 for k in kernels.values())


def test_no_packed_fp32_instructions_in_the_library(tmp_path):
    """MI355X: `v_pk_fma_f32 ... op_sel` results are corrupted while another wave of the same SIMD issues MFMAs + ds_read_b128
    (scripts/repro).  With several batches in flight any kernel can end up beside any other, so the shipped
    library carries no packed-fp32 VALU instruction at all (givepose_amd/build.py)."""
    _kernel_notes(tmp_path)      # extracts the code objects
    n = 0
    for f in sorted(tmp_path.glob("lib.so.*gfx950")):
        dis = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", str(f)], text=True)
        bad = sorted(set(re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b", dis)))
        assert not bad, bad
        n += dis.count("v_mfma_")
    assert n > 1000      # (the disassembly really is the library's: it is full of MFMAs)


def test_no_packed_fp32_in_inline_asm_or_builtins():
    """The compile flag that keeps packed fp32 ops out (-target-feature -packed-fp32-ops) does not reach hand-written asm
    strings, and the disassembly test above only sees what this build instantiated: no source may spell one (reproduced on
    ROCm 7.2 / gfx950, scripts/repro/pkfma_beside_mfma.hip; any v_pk_*_f32 form, mov included, is kept out)."""
    src = os.path.join(ROOT, "givepose_amd", "csrc")
    for f in sorted(os.listdir(src)):
        if f.endswith((".hip", ".hpp")):
            code = re.sub(r"//[^\n]*", "", open(os.path.join(src, f)).read())          # comments may name the instruction
            assert not re.search(r"v_pk_\w+_f32|__builtin_amdgcn_pk_\w*f32", code), f
    flags = open(os.path.join(ROOT, "givepose_amd", "build.py")).read()
    assert "-packed-fp32-ops" in flags


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "givepose_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", s, flags=re.M), f
