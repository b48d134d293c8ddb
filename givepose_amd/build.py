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
# GP_NO_PACKED_FP32=1: build without packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  They
# are the victim half of the cross-wave corruption on MI355X (a wave's v_pk_fma_f32 ... op_sel beside another wave's dense
# MFMA stream; scripts/repro/pkfma_beside_mfma.hip: 0 failures once the victim uses scalar FMAs), and this
# switch removes every one of them from the library (checked: 0 in the disassembly) at a measured cost of 2.3 % of the
# step (8 257 against 8 064 us of kernels, 10.16 against 10.4 k images/s).  The default build keeps them: the one
# aggressor kernel is gone and the overlapped mode is clean in 4 650 stress slot-runs.  (The x86 host pass ignores the
# unknown target feature with a warning.)
FLAGS += os.environ.get("GP_EXTRA_HIPCC_FLAGS", "").split()   # investigation builds (e.g. -DGP_WREG_STAMPS), never the shipped library
if os.environ.get("GP_NO_PACKED_FP32") == "1":
    FLAGS += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-Wno-unknown-warning-option"]


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
