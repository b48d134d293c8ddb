"""Round 5 investigation (library built with -DGP_MLP_STAMPS, GP_MLP512_S32=1): per-wave cycle counts of the fused C = 512 MLP kernel on 32x32x16 MFMAs:
loop GEMM1 phases, loop GEMM2 phases (62 of each), main loop, whole kernel.  Matrix pipe: 32 MFMAs x 32 cycles = 1024 per phase."""
import sys, torch
sys.path.insert(0, ".")
from givepose_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
C, HD = 512, 2048
for CROPS in (64, 128, 256):
    M = 256 * CROPS
    x = torch.randn(M, C, device="cuda", generator=g).half()
    res = torch.randn(M, C, device="cuda", generator=g).half()
    w1 = (torch.randn(HD, C, device="cuda", generator=g) * C ** -0.5).half()
    w2 = (torch.randn(C, HD, device="cuda", generator=g) * HD ** -0.5).half()
    b1, b2, gamma = torch.randn(HD, device="cuda", generator=g), torch.randn(C, device="cuda", generator=g), torch.randn(C, device="cuda", generator=g) * 0.1
    w2p = ops.convnext_mlp_pack_w2(w2)
    o = res.clone()
    for _ in range(3):
        ops.convnext_mlp(x, w1, b1, w2p, b2, gamma, o, res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.convnext_mlp(x, w1, b1, w2p, b2, gamma, o, res)
    e1.record()
    torch.cuda.synchronize()
    st = o.view(M // 32, 32 * C)[:, :16].contiguous().view(torch.int64).double()     # (waves, 4)
    med = st.median(0).values
    us = e0.elapsed_time(e1) * 1e3
    print(f"{CROPS} crops: launch {us:.1f} us | per wave (median): GEMM1 phase {med[0].item() / 62:.0f} cycles, GEMM2 phase {med[1].item() / 62:.0f} (matrix pipe 1024 each), "
          f"main loop {med[2].item():.0f}, kernel {med[3].item():.0f} cycles => {med[3].item() / us / 1e3:.2f} GHz if the slowest wave were the median", flush=True)
