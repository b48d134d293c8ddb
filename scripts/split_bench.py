"""Split-operand mode: images/s of PoseNet(split_gemm=True) against the fp32 MFMA mode, bs 64, hipGraph; per-class kernel times."""
import ctypes, json, sys, time
import torch
sys.path.insert(0, ".")
from givepose_amd import PoseNet, PoseNetConfig, _lib, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
cfg = PoseNetConfig()
host = synth.synth_batch(B, seed=1000)
res = {}
for name, kw in (("split", dict(dtype=torch.float32, split_gemm=True)), ("fp32", dict(dtype=torch.float32)), ("fp16", dict(dtype=torch.float16))):
    net = PoseNet(cfg, seed=0, use_graph=True, **kw).to(dev)
    st = net.static_inputs(B, dev)
    for k, v in host.items():
        st[k].copy_(torch.from_numpy(v).reshape(st[k].shape))
    for _ in range(3):
        o = net.forward_device(st, dev)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        o = net.forward_device(st, dev)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    res[name] = {"images_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3)}
    outs = {k: o[k].float().cpu() for k in ("rot", "trans", "size")}
    res[name]["_out"] = outs
    # per-class eager timing
    lib = _lib.load()
    net.use_graph = False
    net.forward_device(st, dev)
    torch.cuda.synchronize()
    _lib.check(lib.gp_timing_begin(ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "tb")
    net.forward_device(st, dev)
    _lib.check(lib.gp_timing_end(), "te")
    cl = {}
    for c, nm in enumerate(_lib.KC_NAMES):
        nn_, ms, fl, by = ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        lib.gp_timing_report(c, ctypes.byref(nn_), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
        if nn_.value:
            cl[nm] = {"n": nn_.value, "ms": round(ms.value, 3), "tflops": round(fl.value / ms.value / 1e9, 1)}
    res[name]["classes"] = cl
    top = []
    for r in range(12):
        lab = ctypes.create_string_buffer(160)
        c, nn_, ms, fl, by = ctypes.c_int(), ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        if lib.gp_timing_top(r, lab, 160, ctypes.byref(c), ctypes.byref(nn_), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)) != 0:
            break
        top.append((lab.value.decode(), nn_.value, round(ms.value / nn_.value * 1e3, 1), round(fl.value / ms.value / 1e9, 1)))
    res[name]["top"] = top
    del net
ref = res["fp32"]["_out"]
for name in res:
    res[name]["max_abs_vs_fp32_mode"] = {k: float((res[name]["_out"][k] - ref[k]).abs().max()) for k in ref}
    del res[name]["_out"]
print(json.dumps(res, indent=1))
