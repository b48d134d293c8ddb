"""Round 5: where a K step of the ping-pong GEMM (variant 10) goes: s_memtime stamps of wave 0 (group 0) and wave 4 (group 1) of workgroup 0, steps 8-23.
Needs the investigation build: GP_EXTRA_HIPCC_FLAGS="-DGP_PP_STAMPS -DGP_CLOCK_STAMPS" GP_BUILD_TAG=pps python -m givepose_amd.build ; GP_LIB_PATH=.../libgivepose_hip_pps.so
(stamps 4 -> 5 replace the kernel's own lgkmcnt(0) wait: a stamp waits for lgkmcnt(0) itself; each stamp costs ~40 cycles, so the stamped steps are slower than the real ones)."""
import os, sys, torch
sys.path.insert(0, ".")
from givepose_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
ops.CO_SCHEDULED = True
M, N, K = 32768, 512, 2048
x, w = torch.randn(M, K, device="cuda", generator=g).half(), (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
out = torch.empty(M, N, dtype=torch.half, device="cuda")
res, gamma, b = torch.randn(M, N, device="cuda", generator=g).half(), torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
st = torch.zeros(4096 + 2 * 16 * 8 + 64, dtype=torch.int64, device="cuda")
for _ in range(50):
    ops.gemm(x, w, out, bias=b, epilogue=ops.EPI_SCALE_RES, gamma=gamma, residual=res, variant=10, splitk=1)
ops.gemm(x, w, out, bias=b, epilogue=ops.EPI_SCALE_RES, gamma=gamma, residual=res, variant=10, splitk=1, _stamps=st)
torch.cuda.synchronize()
s = st[4096:4096 + 256].view(2, 16, 8).cpu()
names = ["reads issued", "DMA issued", "vmcnt wait", "barrier 1", "lgkmcnt wait", "MFMAs issued", "barrier 2"]
for grp in (0, 1):
    print(f"wave {4 * grp} (group {grp}): cycles per segment, steps 8..23")
    for k in range(15):
        r = s[grp, k]
        d = [(r[i + 1] - r[i]).item() for i in range(7)]
        print(f"  step {k + 8:2d}: " + "  ".join(f"{n} {v:4d}" for n, v in zip(names, d)) + f"   | step total {(s[grp, k + 1, 0] - r[0]).item()}")
    d = (s[grp, 1:, :] - s[grp, :-1, :])[:, 0].float()
    seg = (s[grp, :, 1:] - s[grp, :, :-1]).float().median(0).values
    print("  median: " + "  ".join(f"{n} {int(v)}" for n, v in zip(names, seg.tolist())) + f"   | step {int(d.median())}")
print("offset of group 1's step start behind group 0's:", [(s[1, k, 0] - s[0, k, 0]).item() for k in range(4)])
