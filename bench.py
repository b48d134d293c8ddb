#!/usr/bin/env python3
"""bench.py -- images/s of PoseNet.forward on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the whole hot path (ConvNeXt-B trunk, SizeHead, NOCS head, DCNv3 MAPEncoder, IVFC head,
ConvPnPNet, pose decode) over one batch of 64 synthetic 256x256 crops per GPU, inputs resident in HBM, fp16
storage / fp32 accumulate, random-init (seeded) weights; for N > 1 each rank owns its own 64 crops (weak
scaling) and the step ends with the RCCL all-gather of the per-crop (R,t,s).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     -- dominant kernel (MFMA GEMM / implicit-GEMM conv): algorithmic FLOP per launch / average launch
                  duration, measured with hipEvents around every launch of a separate eager pass on the launch
                  stream (hipGraph replay of the timed region cannot carry per-kernel events); kernel_classes has
                  the same for every kernel class, incl. the DCNv3 gather against the HBM roofline.
  cpu_baseline -- the oracle (oracle/posenet_ref.py, fp32 PyTorch-CPU restatement of the reference) timed on this
                  box's host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_CROP = {"full": 67.517, "nodcn": 66.506, "resnet34": 36.81, "resnet34_nodcn": 35.80, "att": 66.43}
GATHER_MB_PER_CROP = 3.74
PEAK_F16_TFLOPS = 2500.0                                 # MI355X dense fp16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="crops per GPU")
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32"])
    ap.add_argument("--workload", default="full", choices=["full", "nodcn", "resnet34", "resnet34_nodcn", "att"],
                    help="full = reference wiring (ConvNeXt-B + DCNv3, BASELINE configs[2]); nodcn = use_dcn=''; "
                         "resnet34[_nodcn] = BASELINE configs[0-1] read literally (ResNet-34 trunk, not wired by the reference)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent batches in flight per GPU (PoseNet slots: own buffers / hipGraph / stream, shared "
                         "weights); 1 = strictly one step after the other")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    from givepose_amd import PoseNet, PoseNetConfig, _lib, synth
    from givepose_amd import dist as gd
    import torch.distributed as dist

    # GP_BENCH_REHEARSE=1: rehearsal of the N > 1 control flow on a ONE-GPU box (gloo backend, every rank on cuda:0);
    # the numbers mean nothing, the driver's real runs use RCCL with one rank per GPU
    rehearse = os.environ.get("GP_BENCH_REHEARSE") == "1"
    rank, local, world = gd.init_from_env(backend="gloo" if rehearse else None)
    if rehearse:
        local = 0
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B = args.batch
    dtype = torch.float16 if args.dtype == "f16" else torch.float32
    cfg = PoseNetConfig(use_dcn="" if args.workload.endswith("nodcn") else "dcnv3",
                        main_backbone="resnet34" if args.workload.startswith("resnet34") else "convnext",
                        nocsmap_encoder="att" if args.workload == "att" else "conv")   # att = BASELINE configs[3] in-repo analogue
    NF = 1 if args.no_graph else max(1, args.inflight)      # slots need the graph path's per-slot streams
    net = PoseNet(cfg, dtype=dtype, seed=0, use_graph=not args.no_graph, inflight=NF).to(dev)
    statics = [net.static_inputs(B, dev, slot=i) for i in range(NF)]
    static = statics[0]
    host = synth.synth_batch(B, seed=1000 + rank)
    for i, st in enumerate(statics):                        # every slot holds its own batch
        hb = host if i == 0 else synth.synth_batch(B, seed=1000 + rank + 100 * i)
        for k, v in hb.items():
            st[k].copy_(torch.from_numpy(v).reshape(st[k].shape))
    poses = [torch.empty(B, gd.POSE_WIDTH, device=dev) for _ in range(NF)]
    gathered = [torch.empty(world * B, gd.POSE_WIDTH, device=dev) for _ in range(NF)] if world > 1 else None
    counter = [0]

    def step():
        """One pass of the path over one batch; with NF > 1 consecutive steps use different slots and overlap on the
        device (nothing is skipped: every step runs the whole launch sequence on its own buffers)."""
        i = counter[0] % NF
        counter[0] += 1
        out = net.forward_device(statics[i], dev, slot=i, wait=NF == 1)
        if world > 1:
            with torch.cuda.stream(net.stream(i) if NF > 1 else torch.cuda.current_stream(dev)):
                gd.pack_poses(out["rot"], out["trans"], out["size"], out=poses[i])
                gd.all_gather_poses(poses[i], world, out=gathered[i])
        return out

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 2 * NF)):  # per slot: first call eager, graph capture on the second
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    line = {
        "metric": "images/sec PoseNet fwd, bs=64 256x256 fp16, 1/2/4/8 MI355X; % MFMA roofline",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": ("PoseNet.forward: " + ("ResNet-34" if args.workload.startswith("resnet34") else "ConvNeXt-B")
                                + " trunk + SizeHead + NOCS TopDownXyzHead + "
                                + ("plain-conv MAPEncoder (use_dcn='')" if args.workload.endswith("nodcn") else
                                   "MAPTransformerEncoer (64-token attention)" if args.workload == "att" else "DCNv3 MAPEncoder")
                                + " + IVFC TopDownXyzHead + ConvPnPNet + pose decode (BASELINE configs[2]; the reference wires "
                                  "ConvNeXt-B, not ResNet-34: SURVEY.md 0.2)"),
                   "batch_per_gpu": B, "global_batch": world * B, "img": "256x256", "parallelism": f"dp{world}",
                   "weights": "seeded random init (givepose_amd.synth, seed 0)", "hipgraph": not args.no_graph,
                   "batches_in_flight": NF,
                   "collective": "all_gather (B,15) fp32 per rank" if world > 1 else "none"},
        "path_roofline_frac_mfma": round(value / world * GFLOP_PER_CROP[args.workload] * 1e9 / (PEAK_F16_TFLOPS * 1e12), 4),
    }

    # the same K steps strictly one after the other (one batch in flight), for reference beside `value`
    if NF > 1:
        def step1():
            out = net.forward_device(statics[0], dev, slot=0, wait=True)
            if world > 1:
                gd.pack_poses(out["rot"], out["trans"], out["size"], out=poses[0])
                gd.all_gather_poses(poses[0], world, out=gathered[0])
        step1()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step1()
        fence()
        dt1 = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt1], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt1 = float(tt)
        line["one_batch_in_flight"] = {"value": round(world * B * args.steps / dt1, 2), "unit": "images/s",
                                       "ms_per_step": round(dt1 / args.steps * 1e3, 4)}

    # ---------------- roofline leg: per-launch hipEvents on the launch stream, eager pass
    if rank == 0 and not args.no_roofline:
        lib = _lib.load()
        net_e = net
        net_e.use_graph = False
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        reps = 3
        net_e.forward_device(static, dev)
        torch.cuda.synchronize(dev)
        _lib.check(lib.gp_timing_begin(stream), "gp_timing_begin")
        for _ in range(reps):
            net_e.forward_device(static, dev)
        _lib.check(lib.gp_timing_end(), "gp_timing_end")
        classes = {}
        for c, name in enumerate(_lib.KC_NAMES):
            n, ms, fl, by = ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            lib.gp_timing_report(c, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
            if n.value:
                classes[name] = {"launches_per_step": n.value // reps, "ms_per_step": round(ms.value / reps, 4),
                                 "avg_launch_us": round(ms.value / n.value * 1e3, 2),
                                 "tflops": round(fl.value / ms.value / 1e9, 2), "gbs": round(by.value / ms.value / 1e6, 1)}
        g = classes["gemm"]
        peak = PEAK_F16_TFLOPS if args.dtype == "f16" else 157.3
        line["roofline"] = {"kernel": "gemm_kernel<f16> (MFMA GEMM + implicit-GEMM conv, all launches of a step)",
                            "bound": "mfma", "achieved": g["tflops"], "peak": peak, "unit": "TFLOP/s",
                            "frac": round(g["tflops"] / peak, 4), "traffic": None,
                            "launches_per_step": g["launches_per_step"], "avg_launch_us": g["avg_launch_us"]}
        if "dcnv3" in classes:
            d = classes["dcnv3"]
            line["roofline_gather"] = {"kernel": "dcnv3_wave_kernel", "bound": "hbm", "achieved": d["gbs"], "peak": PEAK_HBM_GBS,
                                       "unit": "GB/s", "frac": round(d["gbs"] / PEAK_HBM_GBS, 4), "traffic": None}
        # HBM traffic of the same kernels from PMC counters (collected with rocprofv3 in separate passes and
        # committed under profiles/; bench.py itself cannot read PMCs): bytes per gp_gemm launch, FETCH_SIZE x2-corrected
        pmc = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("pmc_traffic.json")) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
        if pmc and args.batch == 64 and args.dtype == "f16" and args.workload == "full":
            t = json.load(open(os.path.join(ROOT, "profiles", pmc[-1])))["classes"]
            line["roofline"]["traffic"] = round(t["gemm"]["hbm_bytes_per_step"] / g["launches_per_step"])
            line["roofline"]["traffic_unit"] = "HBM bytes per launch (profiles/%s); algorithmic bytes per launch = %d" % (
                pmc[-1], round(g["gbs"] * 1e9 * g["avg_launch_us"] * 1e-6))
            if "dcnv3" in classes and "dcnv3" in t:
                line["roofline_gather"]["traffic"] = round(t["dcnv3"]["hbm_bytes_per_step"] / classes["dcnv3"]["launches_per_step"])
        line["kernel_classes"] = classes
        line["eager_ms_per_step_sum_of_kernels"] = round(sum(c["ms_per_step"] for c in classes.values()), 3)

    # ---------------- H->D inclusive rate (never `value`): the boundary hands over host tensors, so time
    # the same step with every input copied from pinned host memory first, copy and step serialised (no overlap)
    if rank == 0 and not args.no_roofline:
        net.use_graph = not args.no_graph
        pinned = {k: torch.from_numpy(v).reshape(static[k].shape).to(static[k].dtype).pin_memory() for k, v in host.items()}
        def step_h2d():
            for k, v in pinned.items():
                static[k].copy_(v, non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
            net.forward_device(static, dev)
        for _ in range(3):
            step_h2d()
        torch.cuda.synchronize(dev)
        n_h = max(5, min(args.steps, 20))
        t0 = time.perf_counter()
        for _ in range(n_h):
            step_h2d()
        torch.cuda.synchronize(dev)
        hdt = time.perf_counter() - t0
        line["h2d_inclusive"] = {"value": round(B * n_h / hdt, 2), "unit": "images/s (one rank)", "ms_per_step": round(hdt / n_h * 1e3, 4),
                                 "host_bytes_per_step": int(sum(v.numel() * v.element_size() for v in pinned.values())),
                                 "note": "pinned host -> HBM copy of all inputs, then the step; serialised"}
        # same, with the crops made on the device (givepose_amd.preprocess / gp_crop_rois): uint8 frames
        # (4 detections per 640x480 frame) + uint8 masks + boxes travel instead of fp32 crops
        import numpy as np
        from givepose_amd.preprocess import RoiCropper
        rng = np.random.default_rng(0)
        nf = (B + 3) // 4
        frames_h = torch.from_numpy(rng.integers(0, 256, (nf, 480, 640, 3), dtype=np.uint8)).pin_memory()
        masks_h = torch.from_numpy((rng.random((B, 480, 640)) > 0.5).astype(np.uint8)).pin_memory()
        frames_d, masks_d = torch.empty_like(frames_h, device=dev), torch.empty_like(masks_h, device=dev)
        y1, x1 = rng.integers(0, 200, B), rng.integers(0, 300, B)
        boxes = np.stack([y1, x1, y1 + rng.integers(60, 260, B), x1 + rng.integers(60, 320, B)], axis=1)
        fidx, midx = [i // 4 for i in range(B)], list(range(B))
        cropper = RoiCropper(480, 640, dev)
        def step_crop():
            frames_d.copy_(frames_h, non_blocking=True)
            masks_d.copy_(masks_h, non_blocking=True)
            cropper(frames_d, masks_d, fidx, midx, boxes, out=static)
            torch.cuda.current_stream(dev).synchronize()
            net.forward_device(static, dev)
        for _ in range(3):
            step_crop()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n_h):
            step_crop()
        torch.cuda.synchronize(dev)
        cdt2 = time.perf_counter() - t0
        line["h2d_inclusive_device_crop"] = {"value": round(B * n_h / cdt2, 2), "unit": "images/s (one rank)",
                                             "ms_per_step": round(cdt2 / n_h * 1e3, 4),
                                             "host_bytes_per_step": int(frames_h.numel() + masks_h.numel() + B * 120),
                                             "note": "uint8 frames + masks + boxes -> HBM, gp_crop_rois, then the step; serialised"}

    # ---------------- CPU baseline: oracle on the host cores, bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import posenet_ref as O
        P = O.load_params(synth.synth_state_dict(cfg, 0))
        nb = 2
        sample = {k: torch.from_numpy(v) for k, v in synth.synth_batch(nb, seed=1000).items()}
        torch.set_num_threads(min(os.cpu_count() or 1, 32))   # oversubscribing >32 threads slows the small ops down
        with torch.no_grad():
            O.posenet_forward_ref(P, sample, cfg)
            n_it, t0 = 0, time.perf_counter()
            while True:
                O.posenet_forward_ref(P, sample, cfg)
                n_it += 1
                if time.perf_counter() - t0 > 12.0 or n_it >= 20:
                    break
            cdt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(nb * n_it / cdt, 3), "unit": "images/s", "cores": torch.get_num_threads(),
                                "kind": "port", "sample": f"{n_it} x batch of {nb} crops, fp32 PyTorch-CPU oracle (oracle/posenet_ref.py)"}

    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
