"""Long form of tests/test_hip_posenet.py::test_grouped_launches_in_flight_bs128_stress: the bench default shape (2 launch sequences x
2 x 64 crops in flight, fp16) and the split-operand mode in flight, every written plan buffer compared bitwise with the slot's serial
run.   REPS=300 python scripts/stress_grouped.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import PoseNet, PoseNetConfig, synth
REPS = int(os.environ.get("REPS", 200))
dev = torch.device("cuda")
skip = ("h0", "h1", "e_in0", "e_in1", "e_in2")
CASES = (("fp16 2x(2x64)", 128, 2, dict(dtype=torch.float16, dcn_couple=64), REPS),
         ("split 3x64", 64, 3, dict(dtype=torch.float32, split_gemm=True), max(20, REPS // 5)))
if os.environ.get("SMALL"):     # SMALL=1: the small-batch path (small-M GEMM kernel, 2-pixel strip depth-wise kernel) with three forwards in flight
    CASES = (("fp16 3x1", 1, 3, dict(dtype=torch.float16), REPS), ("fp16 3x4", 4, 3, dict(dtype=torch.float16), REPS),
             ("fp16 3x(2x4)", 8, 3, dict(dtype=torch.float16, dcn_couple=4), REPS))
for name, B, NS, kw, reps in CASES:
    net = PoseNet(PoseNetConfig(), seed=0, use_graph=True, inflight=NS, **kw).cuda()
    d = [{k: torch.from_numpy(v).cuda() for k, v in synth.synth_batch(B, seed=81 + i).items()} for i in range(NS)]
    ref = []
    for i in range(NS):
        for _ in range(3):
            net.forward_device(d[i], slot=i)
        torch.cuda.synchronize()
        ref.append({k: v.clone() for k, v in net._plan(B, dev, i)["buf"].items() if k not in skip})
    bad_runs, t0 = 0, time.time()
    for rep in range(reps):
        for i in range(NS):
            net.forward_device(d[i], slot=i, wait=False)
        torch.cuda.synchronize()
        for i in range(NS):
            buf = net._plan(B, dev, i)["buf"]
            bad = [k for k, r in ref[i].items() if not torch.equal(buf[k], r)]
            if bad:
                bad_runs += 1
                print(name, "rep", rep, "slot", i, "DIFFERENT:", bad[:6], flush=True)
        if rep % 50 == 49:
            print(name, rep + 1, "repetitions,", bad_runs, "slot-runs with a differing buffer,", round(time.time() - t0), "s", flush=True)
    print(f"{name}: {bad_runs} of {reps * NS} overlapped slot-runs differ from their serial run", flush=True)
    del net
