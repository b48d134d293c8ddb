"""Build libgivepose_hip.so (gfx950) in-tree with hipcc.  Run: python -m givepose_amd.build"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libgivepose_hip.so")
SOURCES = ["runtime.hip", "gemm.hip", "mlp.hip", "dcnv3.hip", "dcnv3_any.hip", "norm.hip", "misc.hip", "scalenet.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-value"]
# The library is built WITHOUT packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  On MI355X a
# wave's `v_pk_fma_f32 ... op_sel` results come out wrong (one 16-lane pass, low half of one packed register) while another
# wave on the same SIMD issues MFMAs + ds_read_b128 (scripts/repro/pkfma_beside_mfma.hip: 27-99 % of the victim
# launches, 0 with scalar FMAs).  Round 2 first removed the one aggressor kernel it knew and kept the packed ops (they were
# worth 2.3 % then); later in the round a new small-footprint MFMA kernel with a packed GELU corrupted ITSELF with batches in
# flight (8 of 150 stress repetitions, 0 once its GELU was plain VALU), and a low residual rate stayed on sensitive boxes
# (1 of 32 runs of the bs-64 stress test against 0 of 32 for this build).  With the GELU of the large GEMMs on plain FMAs by
# then, this build is no slower (10.81 against 10.75 k images/s, same box).  tests/test_abi.py checks the disassembly.
# GP_PACKED_FP32=1 re-enables them for A/B runs.  (The x86 host pass ignores the unknown target feature with a warning.)
if os.environ.get("GP_PACKED_FP32") != "1":
    FLAGS += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-Wno-unknown-warning-option"]
# Investigation builds (never the product library): GP_EXTRA_HIPCC_FLAGS="-DGP_WREG_STAMPS ..." adds compiler flags and
# GP_BUILD_TAG=<tag> writes libgivepose_hip_<tag>.so from its own object directory, so that the product .so next to it stays
# untouched; load it with GP_LIB_PATH (givepose_amd/_lib.py).  scripts/wreg_stamps.py, scripts/profile_r04.sh.
EXTRA = os.environ.get("GP_EXTRA_HIPCC_FLAGS", "").split()
TAG = os.environ.get("GP_BUILD_TAG", "")
if EXTRA and not TAG:
    raise RuntimeError("GP_EXTRA_HIPCC_FLAGS needs GP_BUILD_TAG: an investigation build must not overwrite the product library")
if TAG:
    OBJ = os.path.join(HERE, "csrc", "build", "tag_" + TAG)
    LIB = os.path.join(HERE, f"libgivepose_hip_{TAG}.so")
    FLAGS += EXTRA


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    deps = [os.path.join(CSRC, "common.hpp"), os.path.join(HERE, "..", "include", "givepose_hip.h")]
    objs, procs = [], []
    for src in SOURCES:
        s, o = os.path.join(CSRC, src), os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            if verbose:
                print("hipcc", src, flush=True)
            procs.append((src, subprocess.Popen([HIPCC, *FLAGS, "-c", s, "-o", o])))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or not os.path.exists(LIB) or any(_newer(o, LIB) for o in objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
        if verbose:
            print("linked", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
