"""Round 6 (review item 4 of round 5): the fused ConvNeXt MLP at C = 128 on v_mfma_f32_32x32x16_f16 (gp_convnext_mlp with GP_MLP_S32) against the 16x16x32 kernel:
correctness against the fp32 formula, then interleaved medians at 64 / 128 crops (M = crops x 4096 rows), one box."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from givepose_amd import ops
C, HD = 128, 512
g = torch.Generator().manual_seed(5)
for crops in (64, 128):
    M = crops * 4096
    x = (torch.randn(M, C, generator=g)).half().cuda(); res = torch.randn(M, C, generator=g).half().cuda()
    w1 = (torch.randn(HD, C, generator=g) * C ** -0.5).half().cuda(); b1 = torch.randn(HD, generator=g).cuda()
    w2 = (torch.randn(C, HD, generator=g) * HD ** -0.5).half().cuda(); b2 = torch.randn(C, generator=g).cuda()
    gamma = torch.randn(C, generator=g).cuda()
    packs = {False: ops.convnext_mlp_pack_w2(w2), True: ops.convnext_mlp_pack_w2(w2, s32=True)}
    outs = {}
    for s32 in (False, True):
        out = res.clone()
        ops.convnext_mlp(x, w1, b1, packs[s32], b2, gamma, out, out, s32=s32)
        outs[s32] = out
    n = 8192
    hid = F.gelu(x[:n].float() @ w1.float().t() + b1).half().float()
    ref = res[:n].float() + gamma * (hid @ w2.float().t() + b2)
    e = {k: float((v[:n].float() - ref).abs().max() / ref.abs().max()) for k, v in outs.items()}
    d = float((outs[True].float() - outs[False].float()).abs().max())
    t = {False: [], True: []}
    for _ in range(7):
        for s32 in (False, True):
            out = res.clone()
            for _ in range(2): ops.convnext_mlp(x, w1, b1, packs[s32], b2, gamma, out, out, s32=s32)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.convnext_mlp(x, w1, b1, packs[s32], b2, gamma, out, out, s32=s32)
            e1.record(); torch.cuda.synchronize()
            t[s32].append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = 4.0 * M * C * HD
    print(f"C=128 {crops} crops (M={M}): 16x16x32 {statistics.median(t[False]):.1f} us ({fl / statistics.median(t[False]) / 1e6:.0f} TFLOP/s), 32x32x16 {statistics.median(t[True]):.1f} us "
          f"({fl / statistics.median(t[True]) / 1e6:.0f} TFLOP/s) | rel err vs fp32 formula {e[False]:.2e} / {e[True]:.2e} | max |s32 - s16| {d:.2e}", flush=True)
