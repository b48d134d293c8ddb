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
