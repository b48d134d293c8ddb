"""fc1 epilogue ablation by cycles (clock-stamp build): variants 17 / 19 / 20 with GELU, RELU, none."""
import os, sys, time, torch
sys.path.insert(0, ".")
sys.argv = sys.argv[:1]
from givepose_amd import ops
import importlib.util
g = torch.Generator(device="cuda").manual_seed(0)
ops.CO_SCHEDULED = True
K, M, N = 512, 32768, 2048
x, w, b = torch.randn(M, K, device="cuda", generator=g).half(), (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half(), torch.randn(N, device="cuda", generator=g)
out = torch.empty(M, N, dtype=torch.half, device="cuda")
def run(name, launch):
    st = torch.zeros(256 * 4 + 64, dtype=torch.int64, device="cuda")
    t0 = time.time()
    while time.time() - t0 < 1.0:
        for _ in range(50):
            launch(None)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch(None)
    e1.record()
    launch(st)
    torch.cuda.synchronize()
    s = st[:1024].view(256, 4).cpu()
    cyc = (s[:, 2] - s[:, 0]).double().median().item()
    clk = ((s[:, 2] - s[:, 0]).double() / (s[:, 3] - s[:, 1]).double() * 0.1).median().item()
    print(f"{name:28s} {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us  main loop {cyc:8.0f} cycles = {cyc / 32:6.0f} per tile (matrix pipe: 2048)  clock {clk:.3f} GHz", flush=True)
for v in (17, 19, 20):
    for en, e in (("gelu", ops.EPI_GELU), ("relu", ops.EPI_RELU), ("none", ops.EPI_NONE)):
        if v == 20 and en != "gelu":
            continue
        run(f"v{v} {en}", lambda st, v=v, e=e: ops.gemm(x, w, out, bias=b, epilogue=e, variant=v, splitk=1, _stamps=st))
