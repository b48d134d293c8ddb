"""In-kernel s_memtime stamps of the weights-in-registers GEMM (variant 16): where a period goes.
Needs an investigation build of the library: GP_EXTRA_HIPCC_FLAGS=-DGP_WREG_STAMPS python -m givepose_amd.build --force
(each stamp costs ~200 cycles; the shipped library does not contain the stamped kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
dev = "cuda"
M, N, K = int(os.environ.get("M", 16384)), 2048, 512
x = torch.randn(M, K).half().to(dev); w = (torch.randn(N, K) * K ** -0.5).half().to(dev); b = torch.randn(N).to(dev)
out = torch.empty(M, N, dtype=torch.float16, device=dev)
st = torch.zeros(8 * 32 * 6, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.gemm(x, w, out, bias=b, epilogue=ops.EPI_GELU, variant=1616, splitk=1, _stamps=st)
torch.cuda.synchronize()
s = st.cpu().view(8, 32, 6)
t0 = s[:, 0, 0].min().item()
names = ["start", "dma issued", "mfma+gelu done", "stores+take", "vmcnt wait", "barrier"]
for wv in (0, 3, 4, 7):
    print(f"wave {wv}: first stamp +{s[wv, 0, 0].item() - t0}")
    for q in range(1, 16):
        r = s[wv, q]
        d = [(r[k + 1] - r[k]).item() for k in range(5)]
        print(f"  q={q:2d} start +{(r[0] - t0).item():6d}  dma {d[0]:5d}  mfma {d[1]:5d}  st {d[2]:5d}  wait {d[3]:5d}  barrier {d[4]:5d}   period {(s[wv, q + 1, 0] - r[0]).item() if q < 15 else 0}")
