#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into per-kernel-class HBM traffic.

  python scripts/pmc_traffic.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> <out.json> [steps profiled] [commit]

(steps profiled = forward passes each rocprofv3 run executed, e.g. 4 for bench.py --steps 2 --warmup 1 --no-graph
--no-roofline --no-cpu-baseline --inflight 1: max(warmup, 2) + steps; gives hbm_bytes_per_step)

Correction (guides/MI355X_MICROARCH.md, HBM section): both counters are in KB; on gfx950 FETCH_SIZE tallies 128-B
requests at 64 B, i.e. reads exactly half of a wide coalesced stream -> doubled here; WRITE_SIZE is exact.
"""
import collections, csv, glob, json, sys

CLASS = (("gemm", ("gemm_big_kernel", "gemm_wreg_kernel", "gemm_wreg2_kernel", "splitk_reduce", "convnext_mlp_kernel", "conv3_pp_kernel")), ("dcnv3", ("dcnv3_",)),
         ("dwconv_ln", ("dwconv",)), ("norm", ("gn_", "layernorm")),
         ("elementwise", ("upsample", "col2im", "pointwise_k3", "mask_resize")),
         ("small", ("stem_", "xyz_out", "smallcin", "size_", "pose_tail")))


def load(d, counter):
    out = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            n = r["Kernel_Name"]
            for cls, keys in CLASS:
                if any(k in n for k in keys):
                    out[cls][0] += 1
                    out[cls][1] += float(r["Counter_Value"])
                    break
    return out


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
res = {}
for cls, _ in CLASS:
    if cls in fe and cls in wr and fe[cls][0]:
        n = fe[cls][0]
        rd = 2.0 * fe[cls][1] * 1024 / n          # gfx950 correction: x2
        wb = wr[cls][1] * 1024 / wr[cls][0]
        res[cls] = {"launches_profiled": n, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wb,
                    "hbm_bytes_per_launch": rd + wb, "hbm_bytes_per_step": (rd + wb) * n / steps}
commit = sys.argv[5] if len(sys.argv) > 5 else "unknown"
# optional 6th argument: scripts/pmc_calib.py's output -- FETCH_SIZE factors measured for this library's access shapes; the DCNv3
# gather reads 128-byte rows with 16 lanes x 8 B, for which the x2 of the wide-stream rule need not hold
calib = None
if len(sys.argv) > 6:
    try:
        calib = json.load(open(sys.argv[6]))
        f_rows = [v["factor_to_apply"] for k, v in calib.items() if "rows8" in k][0]
        if "dcnv3" in res:
            n = fe["dcnv3"][0]
            rd = f_rows * fe["dcnv3"][1] * 1024 / n
            wb = res["dcnv3"]["hbm_write_bytes_per_launch"]
            res["dcnv3_calibrated"] = {"fetch_factor": f_rows, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wb,
                                       "hbm_bytes_per_launch": rd + wb, "hbm_bytes_per_step": (rd + wb) * n / steps,
                                       "note": "FETCH_SIZE factor measured by scripts/pmc_calib.hip for 128-byte rows read by 16 lanes x 8 B"}
    except Exception as e:
        calib = {"error": repr(e)}
json.dump({"commit": commit, "batches_per_launch": int(sys.argv[7]) if len(sys.argv) > 7 else 1, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --no-graph --no-roofline --no-cpu-baseline --no-parity --no-h2d --no-check --no-serial --inflight 1 (scripts/profile_r0N.sh), bs=64 fp16; hbm_bytes_per_step = per batch of 64 crops",
           "correction": "FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B); counters in KB", "calibration": calib, "classes": res},
          open(sys.argv[3], "w"), indent=1)
print(json.dumps(res, indent=1))
