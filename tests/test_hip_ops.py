"""GPU parity tests, op level: every C-ABI entry point of libgivepose_hip.so against a plain PyTorch fp32
CPU reference of the same op (DCNv3: the oracle restatement of the reference CUDA kernel + golden vectors).

Tolerances: fp32 storage path 2e-5 relative-to-scale (fp32 MFMA/accumulate, summation order differs);
fp16 storage path 4e-3 relative-to-scale (fp16 rounding of inputs/outputs, fp32 accumulate).
"""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.float16]
TOL = {torch.float32: 2e-5, torch.float16: 4e-3}


def ops():
    from givepose_amd import ops as o
    return o


def rel_err(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6))


def q(t, dt):
    """Round a fp32 CPU tensor to the storage dtype (so the reference sees the same inputs)."""
    return t.to(dt).float()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ----------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (200, 108, 256), (64, 2048, 1024), (4096, 256, 512), (37, 12, 64)])
def test_gemm_plain(dt, M, N, K):
    o = ops()
    x, w, b = q(rnd(M, K, seed=1), dt), q(rnd(N, K, seed=2, scale=K ** -0.5), dt), rnd(N, seed=3)
    ref = x @ w.t() + b
    out = torch.empty(M, N, dtype=dt, device="cuda")
    o.gemm(x.to("cuda", dt), w.to("cuda", dt), out, bias=b.cuda())
    assert rel_err(out, ref) < TOL[dt]
    out32 = torch.empty(M, N, dtype=torch.float32, device="cuda")
    o.gemm(x.to("cuda", dt), w.to("cuda", dt), out32, bias=b.cuda())
    assert rel_err(out32, ref) < 2e-5 * (4 if dt == torch.float16 else 1) + (0 if dt == torch.float32 else 1e-4)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("splitk", [1, 4, 7])
def test_gemm_splitk_and_epilogues(dt, splitk):
    o = ops()
    M, N, K = 64, 256, 2048
    x, w, b = q(rnd(M, K, seed=4), dt), q(rnd(N, K, seed=5, scale=K ** -0.5), dt), rnd(N, seed=6)
    res, gamma = q(rnd(M, N, seed=7), dt), rnd(N, seed=8)
    lin = x @ w.t() + b
    cases = {o.EPI_GELU: F.gelu(lin), o.EPI_RELU: F.relu(lin), o.EPI_LRELU: F.leaky_relu(lin, 0.1),
             o.EPI_SCALE_RES: res + gamma * lin}
    for epi, ref in cases.items():
        out = torch.empty(M, N, dtype=dt, device="cuda")
        kw = dict(gamma=gamma.cuda(), residual=res.to("cuda", dt)) if epi == o.EPI_SCALE_RES else {}
        o.gemm(x.to("cuda", dt), w.to("cuda", dt), out, bias=b.cuda(), epilogue=epi, splitk=splitk, **kw)
        assert rel_err(out, ref) < TOL[dt], epi


def test_gemm_auto_splitk_stage3_fc2_shape():
    """Round 6: the automatic split-K of long-K GEMMs on 33 .. 128 tiles (ConvNeXt stage-3 fc2 at 16 .. 32 crops): gamma-scaled residual epilogue IN PLACE
    (residual = out, as PoseNet calls it), the route the default takes, against fp32 and against the unsplit launch."""
    o = ops()
    dt = torch.float16
    M, N, K = 1536, 1024, 4096
    assert o.auto_splitk(M, N, K, 2) == 4 and o.auto_splitk(3072, N, K, 2) == 1 and o.auto_splitk(M, N, 2048, 2) == 1 and o.auto_splitk(M, N, K, 4) == 1
    x, w, b = q(rnd(M, K, seed=14), dt), q(rnd(N, K, seed=15, scale=K ** -0.5), dt), rnd(N, seed=16)
    res, gamma = q(rnd(M, N, seed=17), dt), rnd(N, seed=18)
    ref = res + gamma * (x @ w.t() + b)
    outs = []
    for sk in (None, 1):
        out = res.to("cuda", dt).clone()
        o.gemm(x.to("cuda", dt), w.to("cuda", dt), out, bias=b.cuda(), epilogue=o.EPI_SCALE_RES, gamma=gamma.cuda(), residual=out, splitk=sk)
        assert rel_err(out, ref) < TOL[dt], sk
        outs.append(out)
    assert float((outs[0].float() - outs[1].float()).abs().max()) < 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("dt", DT)
def test_gemm_strided_views(dt):
    """ldx (column slice of a wider matrix) and ldc (write into a concat buffer)."""
    o = ops()
    M, K, N = 96, 128, 64
    xw = q(rnd(M, 2 * K, seed=9), dt)
    w = q(rnd(N, K, seed=10, scale=K ** -0.5), dt)
    ref = xw[:, K:] @ w.t()
    big = torch.zeros(M, 3 * N, dtype=dt, device="cuda")
    xd = xw.to("cuda", dt)
    o.gemm(xd[:, K:], w.to("cuda", dt), big[:, N:2 * N], M=M, K=K, ldx=2 * K, ldc=3 * N)
    assert rel_err(big[:, N:2 * N], ref) < TOL[dt]
    assert float(big[:, :N].abs().max()) == 0 and float(big[:, 2 * N:].abs().max()) == 0


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("cfg", [dict(B=2, H=16, Cin=128, Cout=256, k=3, s=1, p=1), dict(B=3, H=16, Cin=64, Cout=128, k=3, s=2, p=1),
                                 dict(B=2, H=8, Cin=128, Cout=256, k=2, s=2, p=0), dict(B=1, H=8, Cin=64, Cout=64, k=1, s=1, p=0)])
def test_conv_implicit_gemm(dt, cfg):
    o = ops()
    B, H, Cin, Cout, k, s, p = (cfg[n] for n in ("B", "H", "Cin", "Cout", "k", "s", "p"))
    x = q(rnd(B, Cin, H, H, seed=11), dt)
    w = q(rnd(Cout, Cin, k, k, seed=12, scale=(Cin * k * k) ** -0.5), dt)
    b = rnd(Cout, seed=13)
    ref = F.conv2d(x, w, b, stride=s, padding=p).permute(0, 2, 3, 1)
    xp = x.permute(0, 2, 3, 1).contiguous().to("cuda", dt)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to("cuda", dt)
    out = o.conv2d_nhwc(xp, wp, k, k, s, p, bias=b.cuda())
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("variant", [2, 3, 4, 5, 7, 8, 9, 10, 11, 12])
def test_gemm_large_tile_variants(dt, variant):
    """256x128 / 256x256 LDS-DMA tiles: ragged M and N edges, every epilogue, ld strides, conv with padding."""
    o = ops()
    for (M, N, K) in [(256, 256, 128), (700, 388, 256), (1000, 512, 64 if dt == torch.float16 else 32), (130, 12, 512)]:
        x, w, b = q(rnd(M, K, seed=61), dt), q(rnd(N, K, seed=62, scale=K ** -0.5), dt), rnd(N, seed=63)
        res, gamma = q(rnd(M, N, seed=64), dt), rnd(N, seed=65)
        lin = x @ w.t() + b
        for epi, ref in ((o.EPI_NONE, lin), (o.EPI_GELU, F.gelu(lin)), (o.EPI_SCALE_RES, res + gamma * lin)):
            out = torch.zeros(M, N + 4, dtype=dt, device="cuda")
            kw = dict(gamma=gamma.cuda(), residual=res.to("cuda", dt)) if epi == o.EPI_SCALE_RES else {}
            o.gemm(x.to("cuda", dt), w.to("cuda", dt), out, bias=b.cuda(), epilogue=epi, variant=variant, ldc=N + 4, **kw)
            assert rel_err(out[:, :N], ref) < TOL[dt], (M, N, K, epi)
            assert float(out[:, N:].abs().max()) == 0.0
        out32 = torch.empty(M, N, dtype=torch.float32, device="cuda")
        o.gemm(x.to("cuda", dt), w.to("cuda", dt), out32, bias=b.cuda(), variant=variant)
        assert rel_err(out32, lin) < (2e-5 if dt == torch.float32 else 2e-4)
    for cfg in (dict(B=3, H=16, Cin=128, Cout=256, k=3, s=1, p=1), dict(B=5, H=16, Cin=64, Cout=128, k=3, s=2, p=1),
                dict(B=4, H=8, Cin=128, Cout=256, k=2, s=2, p=0)):
        B, H, Cin, Cout, k, s_, p_ = (cfg[n] for n in ("B", "H", "Cin", "Cout", "k", "s", "p"))
        x = q(rnd(B, Cin, H, H, seed=66), dt)
        w = q(rnd(Cout, Cin, k, k, seed=67, scale=(Cin * k * k) ** -0.5), dt)
        ref = F.conv2d(x, w, None, stride=s_, padding=p_).permute(0, 2, 3, 1)
        out = o.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to("cuda", dt),
                            w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to("cuda", dt), k, k, s_, p_, variant=variant)
        assert rel_err(out, ref) < TOL[dt], cfg


@pytest.mark.parametrize("variant", [10, 11, 12])
def test_gemm_pingpong_bitwise_vs_plain_schedule(variant):
    """Race screen for the ping-pong schedule (counted vmcnt, LDS ring re-use): every accumulator sees the same k
    order as in the two-stage kernel, so the fp16 outputs must be bit-identical, over repeated launches."""
    o = ops()
    dt = torch.float16
    for (M, N, K) in [(4096, 1024, 2048), (2048 + 128, 512 + 64, 64), (8192, 256, 192), (512, 512, 128)]:
        x = rnd(M, K, seed=71).to("cuda", dt)
        w = rnd(N, K, seed=72, scale=K ** -0.5).to("cuda", dt)
        # aligned shapes: bias + GELU (both sides run the lean epilogue); ragged shapes: no bias, no activation (the
        # edge tiles of the two tilings differ and the generic epilogue adds the bias after the sum, not before)
        aligned = M % 256 == 0 and N % 256 == 0
        b = rnd(N, seed=73).cuda() if aligned else None
        epi = o.EPI_GELU if aligned else o.EPI_NONE
        ref = torch.empty(M, N, dtype=dt, device="cuda")
        o.gemm(x, w, ref, bias=b, epilogue=epi, variant=4)
        for _ in range(5):
            out = torch.zeros(M, N, dtype=dt, device="cuda")
            o.gemm(x, w, out, bias=b, epilogue=epi, variant=variant)
            assert torch.equal(out, ref), (M, N, K)
    xc = rnd(8, 32, 32, 256, seed=74).to("cuda", dt)
    wc = rnd(256, 9 * 256, seed=75, scale=0.02).to("cuda", dt)
    ref = o.conv2d_nhwc(xc, wc, 3, 3, 1, 1, variant=4)
    for _ in range(3):
        assert torch.equal(o.conv2d_nhwc(xc, wc, 3, 3, 1, 1, variant=variant), ref)


def test_conv_window_wide_tile_bitwise_vs_square_tile():
    """The 3x3 LDS-window conv on 512-pixel x 128-channel tiles (variant 813; the default where it fits since round 4) against
    256 x 256 tiles (913): same K order per accumulator, so output and fused GroupNorm statistics must be bit-identical, over
    repeated launches -- at one tile per image half (W = 32: 16 rows), at 8 rows of a 64-wide image, with bias + GELU and without;
    shapes the wide tile does not fit (W = 16; an odd number of 256-pixel tiles) fall back to the square tile."""
    o = ops()
    dt = torch.float16
    for (B, R, epi, bias) in [(2, 64, o.EPI_NONE, False), (3, 32, o.EPI_GELU, True), (5, 64, o.EPI_GELU, True), (4, 32, o.EPI_RELU, False)]:
        x = rnd(B, R, R, 256, seed=141).to("cuda", dt)
        w = rnd(256, 9 * 256, seed=142, scale=0.02).to("cuda", dt)
        b = rnd(256, seed=143).cuda() if bias else None
        g_ref, g_out = torch.zeros(B * (R * R // 64) * 64, device="cuda"), torch.ones(B * (R * R // 64) * 64, device="cuda")
        ref = o.conv2d_nhwc(x, w, 3, 3, 1, 1, bias=b, epilogue=epi, variant=913, gn=(g_ref, 32, R * R))
        for _ in range(3):
            out = o.conv2d_nhwc(x, w, 3, 3, 1, 1, bias=b, epilogue=epi, variant=813, gn=(g_out, 32, R * R))
            assert torch.equal(out, ref) and torch.equal(g_out, g_ref), (B, R, epi)
        assert torch.equal(o.conv2d_nhwc(x, w, 3, 3, 1, 1, bias=b, epilogue=epi, variant=13), ref)       # the automatic choice between the two
    x16 = rnd(6, 16, 16, 256, seed=144).to("cuda", dt)      # W = 16: square tile only
    w = rnd(256, 9 * 256, seed=142, scale=0.02).to("cuda", dt)
    assert torch.equal(o.conv2d_nhwc(x16, w, 3, 3, 1, 1, variant=813), o.conv2d_nhwc(x16, w, 3, 3, 1, 1, variant=913))


@pytest.mark.parametrize("M,N,K", [(24, 256, 1024), (9, 64, 512), (40, 2048, 8192), (17, 1024, 256), (31, 96, 832)])
def test_gemm_small_m_kernel_with_a_partial_last_row_tile(M, N, K):
    """Round 6: variant 18 on a plain GEMM whose row count is no multiple of 16 -- ConvPnPNet's fc layers have M = the crop count, and a forward over
    9 .. 15, 17 .. 31 crops (or the 24-crop bucket of a multi-frame launch) went to a 128 x 128 tile kernel at 22 us a launch where this kernel takes 7.
    Rows past M re-read the last row and are never stored: sentinel rows behind the output stay untouched; forced and automatic route give the same
    bits; every epilogue of the PnP path."""
    o = ops()
    dt = torch.float16
    x, w, b = q(rnd(M, K, seed=171), dt), q(rnd(N, K, seed=172, scale=K ** -0.5), dt), rnd(N, seed=173)
    lin = x @ w.t() + b
    for epi, ref in ((o.EPI_NONE, lin), (o.EPI_LRELU, F.leaky_relu(lin, 0.1)), (o.EPI_GELU, F.gelu(lin))):
        outs = []
        for variant in (18, 0):
            buf = torch.full((M + 32, N), 7.0, dtype=dt, device="cuda")
            o.gemm(x.to("cuda", dt), w.to("cuda", dt), buf, bias=b.cuda(), epilogue=epi, variant=variant, M=M)
            assert rel_err(buf[:M], ref) < TOL[dt], (variant, epi, rel_err(buf[:M], ref))
            assert bool((buf[M:] == 7.0).all()), (variant, epi)
            outs.append(buf[:M].clone())
        assert torch.equal(outs[0], outs[1]), epi      # the automatic choice takes the latency kernel for these shapes


def test_gemm_small_m_latency_variant():
    """Variant 18 (few rows: the detections of one frame; four waves split K, fragments straight from global memory, fixed-order
    LDS reduction): every epilogue, ld strides, 16- and 32-row tiles, ragged K ranges (K / 32 not a multiple of 4), the conv forms
    of the heads and of the down-sampling layers, against torch fp32 -- and that it is what small fp16 launches get by default,
    bitwise reproducible, with split-K requests ignored."""
    o = ops()
    dt = torch.float16
    # (rows, columns, K, default route?): the last three exceed the operand-traffic budget of the automatic choice (32- and 64-row tiles)
    for (M, N, K, auto_route) in [(256, 2048, 512, True), (256, 512, 2048, True), (64, 1024, 4096, True), (512, 2048, 512, True), (16, 32, 64, True),
                                  (48, 96, 832, True), (32, 64, 2304, True), (1024, 512, 2048, True), (1024, 2048, 512, False), (2048, 256, 2048, False), (2112, 64, 576, False)]:
        x, w, b = q(rnd(M, K, seed=161), dt), q(rnd(N, K, seed=162, scale=K ** -0.5), dt), rnd(N, seed=163)
        res, gamma = q(rnd(M, N, seed=164), dt), rnd(N, seed=165)
        lin = x @ w.t() + b
        for epi, ref in ((o.EPI_NONE, lin), (o.EPI_GELU, F.gelu(lin)), (o.EPI_RELU, F.relu(lin)), (o.EPI_SCALE_RES, res + gamma * lin),
                         (o.EPI_RES_RELU, F.relu(res + lin))):
            out = torch.zeros(M, N + 8, dtype=dt, device="cuda")
            kw = dict(residual=res.to("cuda", dt)) if epi in (o.EPI_SCALE_RES, o.EPI_RES_RELU) else {}
            if epi == o.EPI_SCALE_RES:
                kw["gamma"] = gamma.cuda()
            o.gemm(x.to("cuda", dt), w.to("cuda", dt), out, bias=b.cuda(), epilogue=epi, variant=18, ldc=N + 8, **kw)
            assert rel_err(out[:, :N], ref) < TOL[dt], (M, N, K, epi)
            assert float(out[:, N:].abs().max()) == 0.0
            if auto_route:
                auto = torch.zeros(M, N + 8, dtype=dt, device="cuda")
                o.gemm(x.to("cuda", dt), w.to("cuda", dt), auto, bias=b.cuda(), epilogue=epi, ldc=N + 8, splitk=None, **kw)   # default route
                assert torch.equal(auto, out), (M, N, K, epi)
        nob = torch.empty(M, N, dtype=dt, device="cuda")
        o.gemm(x.to("cuda", dt), w.to("cuda", dt), nob, variant=18)
        assert rel_err(nob, x @ w.t()) < TOL[dt]
        for forced, rows in ((218, 16), (318, 32), (418, 64)):      # every tile height forced: the same sums in the same order
            if M % rows == 0:
                t = torch.empty(M, N, dtype=dt, device="cuda")
                o.gemm(x.to("cuda", dt), w.to("cuda", dt), t, variant=forced)
                assert torch.equal(t, nob), (M, N, K, forced)
    # X as a column slice of a wider matrix (ldx), in-place residual (fc2 of a ConvNeXt block: out is the residual)
    M, K, N = 256, 512, 512
    xw, w = q(rnd(M, 2 * K, seed=166), dt).cuda().half(), q(rnd(N, K, seed=167, scale=K ** -0.5), dt).cuda().half()
    y = q(rnd(M, N, seed=168), dt).cuda().half()
    ref = y.float().cpu() + (xw[:, K:].float().cpu() @ w.float().cpu().t())
    o.gemm(xw[:, K:], w, y, M=M, K=K, ldx=2 * K, epilogue=o.EPI_SCALE_RES, gamma=torch.ones(N, device="cuda"), residual=y, variant=18)
    assert rel_err(y, ref) < TOL[dt]
    for cfg in (dict(B=1, H=16, Cin=256, Cout=256, k=3, s=1, p=1), dict(B=2, H=32, Cin=64, Cout=64, k=3, s=1, p=1),
                dict(B=3, H=16, Cin=64, Cout=128, k=3, s=2, p=1), dict(B=1, H=16, Cin=256, Cout=512, k=2, s=2, p=0),
                dict(B=1, H=8, Cin=64, Cout=64, k=1, s=1, p=0)):
        B, H, Cin, Cout, k, s_, p_ = (cfg[n] for n in ("B", "H", "Cin", "Cout", "k", "s", "p"))
        x = q(rnd(B, Cin, H, H, seed=169), dt)
        w = q(rnd(Cout, Cin, k, k, seed=170, scale=(Cin * k * k) ** -0.5), dt)
        b = rnd(Cout, seed=171)
        ref = F.gelu(F.conv2d(x, w, b, stride=s_, padding=p_)).permute(0, 2, 3, 1)
        xp, wp = x.permute(0, 2, 3, 1).contiguous().to("cuda", dt), w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to("cuda", dt)
        out = o.conv2d_nhwc(xp, wp, k, k, s_, p_, bias=b.cuda(), epilogue=o.EPI_GELU, variant=18)
        assert rel_err(out, ref) < TOL[dt], cfg
        assert torch.equal(o.conv2d_nhwc(xp, wp, k, k, s_, p_, bias=b.cuda(), epilogue=o.EPI_GELU, variant=18), out)
    # fused GroupNorm statistics (64-, 32- and 16-row workgroup tiles = statistics chunks, gp_gemm_desc.gn_rows; 8 and 4 channels per group; plain GEMM and
    # conv): the statistics the kernel leaves in its chunks normalise its output exactly as a separate statistics pass over that output does
    for (Bc, HW, N, K, conv) in [(1, 256, 256, 256, None), (2, 64, 128, 512, None), (1, 1024, 256, 2304, 32), (3, 256, 128, 1152, 16)]:
        G = 32
        y64 = None
        for rows in (64, 32, 16):
            gn = lambda part_: (part_, G, HW) if rows == 64 else (part_, G, HW, rows)       # (the 3-tuple form = 64-row chunks)
            if conv:
                Cin = K // 9
                xi = q(rnd(Bc, conv, conv, Cin, seed=174), dt).to("cuda", dt)
                w = q(rnd(N, K, seed=175, scale=K ** -0.5), dt).to("cuda", dt)
                part = torch.zeros(1 << 16, device="cuda")
                y = o.conv2d_nhwc(xi, w, 3, 3, 1, 1, variant=18, gn=gn(part)).view(Bc, HW, N)
                y7 = o.conv2d_nhwc(xi, w, 3, 3, 1, 1, variant=7, gn=(torch.zeros(1 << 16, device="cuda"), G, HW)).view(Bc, HW, N)
            else:
                xi = q(rnd(Bc * HW, K, seed=174), dt).to("cuda", dt)
                w = q(rnd(N, K, seed=175, scale=K ** -0.5), dt).to("cuda", dt)
                part = torch.zeros(1 << 16, device="cuda")
                y = torch.empty(Bc, HW, N, dtype=dt, device="cuda")
                o.gemm(xi, w, y.view(-1, N), variant=18, gn=gn(part))
                y7 = torch.empty_like(y)
                o.gemm(xi, w, y7.view(-1, N), variant=7, gn=(torch.zeros(1 << 16, device="cuda"), G, HW))
            assert rel_err(y, y7.float().cpu()) < 2e-3, (Bc, HW, N, K, rows)
            if y64 is None:
                y64 = y.clone()
            else:
                assert torch.equal(y, y64), (Bc, HW, N, K, rows)        # the tile height changes which workgroup computes a value, not the value
            gw, gb = (1 + 0.1 * rnd(N, seed=176)).cuda(), (0.1 * rnd(N, seed=177)).cuda()
            fused = o.groupnorm(y, gw, gb, torch.empty_like(y), G, o.ACT_GELU, part, fused_stats=True, rows=rows)
            sep = o.groupnorm(y, gw, gb, torch.empty_like(y), G, o.ACT_GELU, torch.zeros(1 << 16, device="cuda"))
            # (the fused statistics are of the fp32 values before the fp16 store, the separate pass reads the stored fp16: not bitwise)
            assert rel_err(fused, sep.float().cpu()) < 2e-3, (Bc, HW, N, K, rows)
            ref = F.gelu(F.group_norm(y.float().cpu().permute(0, 2, 1), G, gw.cpu(), gb.cpu(), 1e-5)).permute(0, 2, 1)
            assert rel_err(fused, ref) < TOL[dt], (Bc, HW, N, K, rows)
            if conv and rows < 64:      # the other two consumers of fused statistics read the same chunks
                up = o.groupnorm_upsample2x(y.view(Bc, conv, conv, N), gw, gb, torch.empty(Bc, 2 * conv, 2 * conv, N, dtype=dt, device="cuda"), G, o.ACT_GELU, part, rows=rows)
                up_ref = o.upsample_bilinear2x(fused.view(Bc, conv, conv, N), torch.empty_like(up))
                assert torch.equal(up, up_ref), (Bc, HW, N, K, rows)
    # chunk rows the library asks for: 64 wherever the small-M kernel would not take the launch; a multiple of 16 that divides M otherwise
    assert o.gemm_gn_rows(64 * 4096, 256, 2304, 4096) == 64 and o.gemm_gn_rows(256, 256, 2304, 256) in (16, 32) and o.gemm_gn_rows(256, 100, 2304, 256) == 64
    # 16- / 32-row chunks exist in the small-M kernel only: a launch it cannot take fails instead of writing 64-row chunks the consumer would misread
    with pytest.raises(RuntimeError, match="gn_rows"):
        o.gemm(x32h := rnd(256, 128, seed=178).cuda().half(), rnd(64, 128, seed=179).cuda().half(), torch.empty(256, 64, dtype=torch.float32, device="cuda"),
               gn=(torch.zeros(1 << 12, device="cuda"), 16, 64, 16))
    # refused loudly where it does not apply: fp32 storage, fp32 output, N not a multiple of 32
    x32, w32 = rnd(64, 128, seed=172).cuda(), rnd(64, 128, seed=173).cuda()
    with pytest.raises(RuntimeError, match="variant 18"):
        o.gemm(x32, w32, torch.empty(64, 64, device="cuda"), variant=18)
    # fp16 operands with an fp32 output (the DCNv3 offset | mask projection, ops_dcnv3/modules/dcnv3.py:341-349): taken since round 5, activation epilogues only
    o32 = o.gemm(x32.half(), w32.half(), torch.empty(64, 64, device="cuda"), variant=18)
    assert rel_err(o32, x32.half().float().cpu() @ w32.half().float().cpu().t()) < TOL[dt]
    with pytest.raises(RuntimeError, match="variant 18"):
        o.gemm(x32.half(), w32[:48].half().contiguous(), torch.empty(64, 48, dtype=dt, device="cuda"), variant=18)


def test_gemm_row_vector_variant():
    """Variant 23 (round 5): M <= 8 rows -- ConvPnPNet's fc layers over the detections of one frame (network/conv_pnp_net.py:186-199) -- a wave per two output
    columns, weights straight into registers, v_dot2 into fp32.  Against torch on the fp16-rounded operands; chosen automatically for such shapes; replay-bitwise;
    refused where it does not apply."""
    import givepose_amd.ops as o
    dt = torch.float16
    for (M, N, K, ldx, epi) in [(1, 2048, 8192, 8192, o.EPI_LRELU), (4, 2048, 8192, 8192, o.EPI_LRELU), (3, 256, 1024, 2048, o.EPI_LRELU), (8, 256, 1024, 2048, o.EPI_NONE),
                                (5, 64, 512, 512, o.EPI_RELU), (2, 8, 4608, 4608, o.EPI_GELU)]:
        xfull = q(rnd(M, ldx, seed=181), dt)
        w = q(rnd(N, K, seed=182, scale=K ** -0.5), dt)
        b = rnd(N, seed=183)
        off = ldx - K                                    # (fc2z reads the second half of fc1's rows: conv_pnp_net.py:196)
        x = xfull[:, off:]
        v = x @ w.t() + b
        ref = {o.EPI_NONE: v, o.EPI_LRELU: F.leaky_relu(v, 0.1), o.EPI_RELU: F.relu(v), o.EPI_GELU: F.gelu(v)}[epi]
        xg, wg, bg = xfull.to("cuda", dt), w.to("cuda", dt), b.cuda()
        outs = []
        for variant in (23, 0, 23):
            out = torch.full((M, N), float("nan"), dtype=dt, device="cuda")
            o.gemm(xg[:, off:], wg, out, bias=bg, epilogue=epi, variant=variant, M=M, K=K, ldx=ldx)
            outs.append(out)
            assert rel_err(out, ref) < TOL[dt], (M, N, K, epi, variant)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (M, N, K)      # the automatic choice IS variant 23 here; replays agree bitwise
        out32 = torch.full((M, N), float("nan"), device="cuda")                                 # fp32 output (fc2 / fc2z feed the fp32 pose heads)
        o.gemm(xg[:, off:], wg, out32, bias=bg, epilogue=epi, variant=23, M=M, K=K, ldx=ldx)
        assert rel_err(out32, ref) < TOL[dt] and torch.equal(out32.half(), outs[0]), (M, N, K)
    x9, w9 = rnd(16, 512, seed=184).cuda().half(), rnd(64, 512, seed=185).cuda().half()
    with pytest.raises(RuntimeError, match="variant 23"):
        o.gemm(x9, w9, torch.empty(16, 64, dtype=dt, device="cuda"), variant=23)                 # 16 rows
    with pytest.raises(RuntimeError, match="variant 23"):
        o.gemm(x9[:4, :256].contiguous(), w9[:, :256].contiguous(), torch.empty(4, 64, dtype=dt, device="cuda"), variant=23)     # K = 256


@pytest.mark.parametrize("variant", [16, 17])
def test_gemm_weights_in_registers_variant(variant):
    """Variants 16 / 17 (K = 512, a 256-row slice of W resident in registers, X streamed in 32-row tiles, GELU in the MFMA
    shadow, counted vmcnt over DMA + stores; 17 = the round-4 form, the default of stage-2 fc1: output channels re-ordered for
    16-byte stores, DMA lead 2): bit-identical to the tile kernels where those run their lean epilogue (same k order per
    accumulator, same GELU roundings), over repeated launches; fp32 formula on the ragged cases (1, 3, 5 tiles per row group,
    groups of unequal length, padded grid items, ldc > N, no bias); refusals."""
    o = ops()
    dt, K = torch.float16, 512
    # tiles per row group T = 16, 2, 1, 1, 1, then 3 / 5 / 7 (prologue, steady state and drain of the counted vmcnt window)
    for (M, N, epi) in [(16384, 2048, o.EPI_GELU), (8192, 512, o.EPI_GELU), (4096, 256, o.EPI_NONE), (2048, 1024, o.EPI_LRELU), (1024, 256, o.EPI_RELU),
                        (24576, 256, o.EPI_GELU), (40960, 256, o.EPI_GELU), (57344, 256, o.EPI_NONE)]:
        x, w, b = rnd(M, K, seed=91).to("cuda", dt), rnd(N, K, seed=92, scale=K ** -0.5).to("cuda", dt), rnd(N, seed=93).cuda()
        ref = torch.empty(M, N, dtype=dt, device="cuda")
        o.gemm(x, w, ref, bias=b, epilogue=epi, variant=8, splitk=1)
        for _ in range(4):
            out = torch.full((M, N), 7.0, dtype=dt, device="cuda")
            o.gemm(x, w, out, bias=b, epilogue=epi, variant=variant, splitk=1)
            assert torch.equal(out, ref), (M, N, epi)
    for (M, N, epi, bias) in [(32, 256, o.EPI_GELU, True), (96, 256, o.EPI_NONE, False), (160, 512, o.EPI_GELU, True),
                              (4096 + 32, 768, o.EPI_LRELU, True), (33 * 32, 256, o.EPI_RELU, True)]:
        x, w = q(rnd(M, K, seed=94), dt), q(rnd(N, K, seed=95, scale=K ** -0.5), dt)
        b = rnd(N, seed=96) if bias else None
        lin = x @ w.t() + (b if bias else 0.0)
        ref = {o.EPI_NONE: lin, o.EPI_GELU: F.gelu(lin), o.EPI_RELU: F.relu(lin), o.EPI_LRELU: F.leaky_relu(lin, 0.1)}[epi]
        out = torch.zeros(M, N + 8, dtype=dt, device="cuda")
        o.gemm(x.to("cuda", dt), w.to("cuda", dt), out, bias=b.cuda() if bias else None, epilogue=epi, variant=variant, ldc=N + 8, splitk=1)
        assert rel_err(out[:, :N], ref) < TOL[dt], (M, N, epi)
        assert float(out[:, N:].abs().max()) == 0.0
    x, w = rnd(64, 512, seed=97).to("cuda", dt), rnd(256, 512, seed=98).to("cuda", dt)
    for bad in (dict(x=x[:48], w=w), dict(x=x, w=w[:192]), dict(x=rnd(64, 256, seed=97).to("cuda", dt), w=rnd(256, 256, seed=98).to("cuda", dt))):
        with pytest.raises(RuntimeError, match=f"variant {variant}"):
            o.gemm(bad["x"], bad["w"], torch.empty(bad["x"].shape[0], bad["w"].shape[0], dtype=dt, device="cuda"), variant=variant, splitk=1)
    if variant == 17:      # 16-byte stores: a row stride that is not a multiple of 8 halfs is refused (variant 0 then picks 16)
        with pytest.raises(RuntimeError, match="variant 17"):
            o.gemm(x, w, torch.zeros(64, 260, dtype=dt, device="cuda"), variant=17, ldc=260, splitk=1)


@pytest.mark.parametrize("variant", [19, 20, 21, 22])
def test_gemm_weights_in_registers_two_accumulator_sets(variant):
    """Variants 19-22 (round 5, gemm_wreg3_kernel: two accumulator sets, the first MFMA of every accumulator takes the bias as its C operand; 22 / 21 on 16x16x32 MFMAs,
    19 / 20 on 32x32x16; 21 -- the default of stage-2 fc1 -- and 20 apply the GELU on packed fp16).  22 is variant 17's arithmetic: bit-identical to it, every epilogue, even
    and odd tile counts per row group (T = 1, 2, 3, 5, 16: both accumulator sets, the peeled last tile), repeated launches.  19 differs by the K order inside an MFMA only
    (fp32 formula, TOL).  20 / 21: the packed GELU's absolute error (one fp16 rounding of a value in [0.25, 0.5), 2.2e-3 at most for |x| > 4) against the fp32 formula, equal to
    each other bit for bit only per MFMA shape; the other epilogues run the fp32 activation (= 19 / 22)."""
    o = ops()
    dt, K = torch.float16, 512
    for (M, N, epi) in [(16384, 2048, o.EPI_GELU), (8192, 512, o.EPI_GELU), (4096, 256, o.EPI_NONE), (2048, 1024, o.EPI_LRELU), (1024, 256, o.EPI_RELU),
                        (24576, 256, o.EPI_GELU), (40960, 256, o.EPI_GELU), (32, 256, o.EPI_GELU), (64, 256, o.EPI_GELU), (160, 512, o.EPI_GELU)]:
        x, w, b = rnd(M, K, seed=191).to("cuda", dt), rnd(N, K, seed=192, scale=K ** -0.5).to("cuda", dt), rnd(N, seed=193).cuda()
        ref17 = torch.empty(M, N, dtype=dt, device="cuda")
        o.gemm(x, w, ref17, bias=b, epilogue=epi, variant=17, splitk=1)
        lin = x.float() @ w.float().t() + b
        ref = {o.EPI_NONE: lin, o.EPI_GELU: F.gelu(lin), o.EPI_RELU: F.relu(lin), o.EPI_LRELU: F.leaky_relu(lin, 0.1)}[epi]
        outs = []
        for _ in range(3):
            out = torch.full((M, N), 7.0, dtype=dt, device="cuda")
            o.gemm(x, w, out, bias=b, epilogue=epi, variant=variant, splitk=1)
            outs.append(out)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (M, N, epi)
        packed = variant in (20, 21) and epi == o.EPI_GELU
        if variant == 22 or (variant == 21 and epi != o.EPI_GELU):
            assert torch.equal(outs[0], ref17), (M, N, epi)
        d = (outs[0].float() - ref).abs()
        assert float(d.max()) < (3.5e-3 if packed else 2.5e-3) * max(1.0, float(ref.abs().max()) / 4), (M, N, epi, float(d.max()))
        assert float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()) < (5e-4 if packed else 3e-4), (M, N, epi)      # measured 3.7e-4 / 2.1e-4 (fp16 storage: 2^-11 / sqrt 3)
    # ldc > N, no bias, ragged M (tiles of unequal groups), zero padding columns untouched
    for (M, N, epi, bias) in [(96, 256, o.EPI_NONE, False), (4096 + 32, 768, o.EPI_GELU, True), (33 * 32, 256, o.EPI_RELU, True)]:
        x, w = q(rnd(M, K, seed=194), dt), q(rnd(N, K, seed=195, scale=K ** -0.5), dt)
        b = rnd(N, seed=196) if bias else None
        lin = x @ w.t() + (b if bias else 0.0)
        ref = {o.EPI_NONE: lin, o.EPI_GELU: F.gelu(lin), o.EPI_RELU: F.relu(lin)}[epi]
        out = torch.zeros(M, N + 8, dtype=dt, device="cuda")
        o.gemm(x.to("cuda", dt), w.to("cuda", dt), out, bias=b.cuda() if bias else None, epilogue=epi, variant=variant, ldc=N + 8, splitk=1)
        assert rel_err(out[:, :N], ref) < TOL[dt], (M, N, epi)
        assert float(out[:, N:].abs().max()) == 0.0
    x, w = rnd(64, 512, seed=197).to("cuda", dt), rnd(256, 512, seed=198).to("cuda", dt)
    for bad in (dict(x=x[:48], w=w), dict(x=x, w=w[:192]), dict(x=rnd(64, 256, seed=197).to("cuda", dt), w=rnd(256, 256, seed=198).to("cuda", dt))):
        with pytest.raises(RuntimeError, match="variant 19-22"):
            o.gemm(bad["x"], bad["w"], torch.empty(bad["x"].shape[0], bad["w"].shape[0], dtype=dt, device="cuda"), variant=variant, splitk=1)


def test_gemm_prefetch_hint_changes_nothing():
    """gp_gemm_desc.prefetch is a hint: the workgroups touch the given bytes (any length, 16-byte pieces, at most four
    1-KB pieces per wave) and the result is bitwise the one without it -- every schedule family, conv and window conv."""
    o = ops()
    dt = torch.float16
    pf_big = torch.zeros(33 * 1024 * 1024 // 2 + 5, dtype=dt, device="cuda")     # 33 MB + 10 bytes
    for pf in (pf_big, pf_big[:500], pf_big[:8], pf_big[:2048 + 4], pf_big[: 2 * 1024 * 1024]):
        for (M, N, K, variant) in [(4096, 1024, 512, 7), (4096, 1024, 512, 10), (16384, 2048, 512, 16), (16384, 2048, 512, 17), (2048, 512, 256, 8),
                                   (1024, 512, 1024, 4), (1000, 260, 128, 5)]:
            x, w, b = rnd(M, K, seed=101).to("cuda", dt), rnd(N, K, seed=102, scale=K ** -0.5).to("cuda", dt), rnd(N, seed=103).cuda()
            ref = torch.empty(M, N, dtype=dt, device="cuda")
            out = torch.empty(M, N, dtype=dt, device="cuda")
            o.gemm(x, w, ref, bias=b, epilogue=o.EPI_GELU, variant=variant, splitk=1)
            o.gemm(x, w, out, bias=b, epilogue=o.EPI_GELU, variant=variant, splitk=1, prefetch=pf)
            assert torch.equal(out, ref), (M, N, K, variant, pf.numel())
        xc, wc = rnd(8, 32, 32, 256, seed=104).to("cuda", dt), rnd(256, 9 * 256, seed=105, scale=0.02).to("cuda", dt)
        for variant in (13, 7):
            assert torch.equal(o.conv2d_nhwc(xc, wc, 3, 3, 1, 1, variant=variant, prefetch=pf), o.conv2d_nhwc(xc, wc, 3, 3, 1, 1, variant=variant))
    xs, ws = rnd(64, 8192, seed=106).to("cuda", dt), rnd(2048, 8192, seed=107, scale=0.01).to("cuda", dt)   # split-K path
    ref, out = torch.empty(64, 2048, dtype=dt, device="cuda"), torch.empty(64, 2048, dtype=dt, device="cuda")
    o.gemm(xs, ws, ref, variant=4, splitk=8)
    o.gemm(xs, ws, out, variant=4, splitk=8, prefetch=pf_big)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("C", [128, 256, 512])
def test_convnext_mlp_fused(C):
    """Fused fc1 -> GELU -> fc2 -> gamma * . + shortcut against the fp32 formula (hidden rounded to fp16 like the
    two-GEMM path) and against the two-GEMM HIP path itself; in place over the residual.  C = 512: the round-5 one-wave-per-SIMD kernel
    (128 rows per workgroup: M = 640 = five workgroups, more than one XCD chunk; the library keeps it, PoseNet does not use it:
    profiles/r05_mlp512_ab.txt)."""
    o = ops()
    dt = torch.float16
    M, HD = (640 if C == 512 else 512), 4 * C
    x, res = q(rnd(M, C, seed=81), dt), q(rnd(M, C, seed=82), dt)
    w1, b1 = q(rnd(HD, C, seed=83, scale=C ** -0.5), dt), rnd(HD, seed=84)
    w2, b2 = q(rnd(C, HD, seed=85, scale=HD ** -0.5), dt), rnd(C, seed=86)
    gamma = rnd(C, seed=87)
    hid = q(F.gelu(x @ w1.t() + b1), dt)
    ref = res + gamma * (hid @ w2.t() + b2)
    dev = lambda t, d=None: t.to("cuda", d) if d else t.cuda()
    w2p = o.convnext_mlp_pack_w2(dev(w2, dt))
    out = dev(res, dt).clone()
    o.convnext_mlp(dev(x, dt), dev(w1, dt), dev(b1), w2p, dev(b2), dev(gamma), out, out)
    assert rel_err(out, ref) < 2e-3
    h2 = torch.empty(M, HD, dtype=dt, device="cuda")
    out2 = dev(res, dt).clone()
    o.gemm(dev(x, dt), dev(w1, dt), h2, bias=dev(b1), epilogue=o.EPI_GELU)
    o.gemm(h2, dev(w2, dt), out2, bias=dev(b2), epilogue=o.EPI_SCALE_RES, gamma=dev(gamma), residual=out2)
    assert rel_err(out, out2.float().cpu()) < 1e-3
    from givepose_amd._lib import GivePoseHipError
    with pytest.raises(GivePoseHipError):
        o.convnext_mlp(dev(x, dt)[:100], dev(w1, dt), dev(b1), w2p, dev(b2), dev(gamma), out[:100], out[:100])


def test_convnext_mlp_fused_s32_form():
    """Round 6: the C = 128 fused MLP on v_mfma_f32_32x32x16_f16 (GP_MLP_S32; its own W2 column order): against the fp32 formula and the 16x16x32 kernel.
    Measured slower (profiles/r06_mlp_s32_ab.txt), kept behind the flag; C = 256 is refused (it spilled)."""
    o = ops()
    dt = torch.float16
    C, M = 128, 768          # (three 4-wave workgroups of the s32 form, a multiple of the 16x16 kernel's 256 rows)
    HD = 4 * C
    x, res = q(rnd(M, C, seed=81), dt), q(rnd(M, C, seed=82), dt)
    w1, b1 = q(rnd(HD, C, seed=83, scale=C ** -0.5), dt), rnd(HD, seed=84)
    w2, b2 = q(rnd(C, HD, seed=85, scale=HD ** -0.5), dt), rnd(C, seed=86)
    gamma = rnd(C, seed=87)
    ref = res + gamma * (q(F.gelu(x @ w1.t() + b1), dt) @ w2.t() + b2)
    dev = lambda t, d=None: t.to("cuda", d) if d else t.cuda()
    outs = []
    for s32 in (False, True):
        out = dev(res, dt).clone()
        o.convnext_mlp(dev(x, dt), dev(w1, dt), dev(b1), o.convnext_mlp_pack_w2(dev(w2, dt), s32=s32), dev(b2), dev(gamma), out, out, s32=s32)
        assert rel_err(out, ref) < 2e-3, s32
        outs.append(out)
    assert float((outs[0].float() - outs[1].float()).abs().max()) < 1e-2 * float(ref.abs().max())
    from givepose_amd._lib import GivePoseHipError
    w256 = torch.zeros(256, 1024, dtype=dt, device="cuda")
    with pytest.raises(GivePoseHipError):
        o.convnext_mlp(torch.zeros(256, 256, dtype=dt, device="cuda"), w256.t().contiguous(), torch.zeros(1024, device="cuda"), w256, torch.zeros(256, device="cuda"),
                       torch.zeros(256, device="cuda"), torch.zeros(256, 256, dtype=dt, device="cuda"), torch.zeros(256, 256, dtype=dt, device="cuda"), s32=True)


def test_dwconv7_raw_stats_and_lnfold_gemm():
    """LayerNorm deferred to the GEMM epilogue: dw7x7 raw output + slab moments, then fc1 with GP_EPI_LNFOLD_GELU must
    reproduce gelu(fc1(LayerNorm(dwconv(x)))) (the ConvNeXt block front half)."""
    o = ops()
    dt = torch.float16
    B, H, C, N = 4, 16, 512, 512
    x = q(rnd(B, C, H, H, seed=91), dt)
    wdw, bdw = q(rnd(C, 1, 7, 7, seed=92, scale=0.15), dt), rnd(C, seed=93, scale=0.5)
    lw, lb = 1.0 + 0.3 * rnd(C, seed=94), 0.2 * rnd(C, seed=95)
    w1, b1 = rnd(N, C, seed=96, scale=C ** -0.5), rnd(N, seed=97)
    conv = F.conv2d(x, wdw, bdw, padding=3, groups=C).permute(0, 2, 3, 1)            # (B,H,W,C)
    hid_ref = F.gelu(F.layer_norm(conv, (C,), lw, lb, 1e-6).reshape(-1, C) @ w1.t() + b1)
    xd = x.permute(0, 2, 3, 1).contiguous().to("cuda", dt)
    wt = wdw.reshape(C, 49).t().contiguous().to("cuda", dt)
    y = torch.empty(B, H, H, C, dtype=dt, device="cuda")
    stats = torch.zeros(B * H * H, 2, C // 128, device="cuda")
    o.dwconv7_raw_stats(xd, wt, bdw.cuda(), y, stats)
    assert rel_err(y, conv) < 2e-3
    yr = y.float().cpu().reshape(-1, C // 128, 128)
    assert torch.allclose(stats[:, 0].cpu(), yr.sum(-1), rtol=1e-4, atol=1e-3)
    assert torch.allclose(stats[:, 1].cpu(), (yr * yr).sum(-1), rtol=1e-4, atol=1e-3)
    wg = (w1 * lw[None, :]).to("cuda", dt)
    cs = wg.float().sum(1).contiguous()
    cb = (w1 @ lb + b1).cuda()
    for variant in (0, 8, 10):
        hid = torch.empty(B * H * H, N, dtype=dt, device="cuda")
        o.gemm(y.view(-1, C), wg, hid, bias=cb, epilogue=o.EPI_LNFOLD_GELU, ln=(stats, cs, C // 128, 1e-6), variant=variant)
        assert rel_err(hid, hid_ref) < 4e-3, variant
    from givepose_amd._lib import GivePoseHipError
    with pytest.raises(GivePoseHipError):     # M not a multiple of 256
        o.gemm(y.view(-1, C)[:100], wg, hid[:100], bias=cb, epilogue=o.EPI_LNFOLD_GELU, ln=(stats, cs, C // 128, 1e-6))


@pytest.mark.parametrize("cfg", [dict(B=3, H=64, Cin=256), dict(B=5, H=32, Cin=256), dict(B=6, H=16, Cin=64), dict(B=2, H=32, Cin=128)])
def test_conv3x3_lds_window_kernel(cfg):
    """3x3 s1 p1 conv, Cout 256, X staged as an LDS window (variant 13; chunk-outer / tap-inner K order): against
    F.conv2d and against the tap-by-tap ping-pong kernel; image borders, several images per launch, fused GroupNorm
    statistics, every epilogue it supports; repeated launches are bit-identical (race screen)."""
    o = ops()
    dt = torch.float16
    B, H, Cin = cfg["B"], cfg["H"], cfg["Cin"]
    x = q(rnd(B, Cin, H, H, seed=101), dt)
    w = q(rnd(256, Cin, 3, 3, seed=102, scale=(Cin * 9) ** -0.5), dt)
    b = rnd(256, seed=103)
    lin = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1)
    xp = x.permute(0, 2, 3, 1).contiguous().to("cuda", dt)
    wp = w.permute(0, 2, 3, 1).reshape(256, -1).contiguous().to("cuda", dt)
    for epi, ref in ((o.EPI_NONE, lin), (o.EPI_GELU, F.gelu(lin)), (o.EPI_RELU, F.relu(lin))):
        out = o.conv2d_nhwc(xp, wp, 3, 3, 1, 1, bias=b.cuda(), epilogue=epi, variant=13)
        assert rel_err(out, ref) < TOL[dt], (cfg, epi)
        for _ in range(3):
            assert torch.equal(o.conv2d_nhwc(xp, wp, 3, 3, 1, 1, bias=b.cuda(), epilogue=epi, variant=13), out)
    # fused GroupNorm statistics: same numbers as the ping-pong kernel's epilogue (the statistics are those of the
    # rounded outputs, which may differ in the last fp16 bit between the two K orders)
    G, hw = 32, H * H
    pa = torch.zeros(B * (hw // 64) * G * 2, device="cuda")
    pb = torch.zeros_like(pa)
    ya = o.conv2d_nhwc(xp, wp, 3, 3, 1, 1, variant=13, gn=(pa, G, hw))
    yb = o.conv2d_nhwc(xp, wp, 3, 3, 1, 1, variant=10, gn=(pb, G, hw))
    assert rel_err(ya, yb.float().cpu()) < 2e-3
    assert torch.allclose(pa, pb, rtol=2e-3, atol=5e-2)
    from givepose_amd._lib import GivePoseHipError
    with pytest.raises(GivePoseHipError):     # Cout != 256
        o.conv2d_nhwc(xp, wp[:128].contiguous(), 3, 3, 1, 1, variant=13)


def test_gemm_rejects_bad_shapes():
    o = ops()
    from givepose_amd._lib import GivePoseHipError
    x = torch.zeros(8, 100, device="cuda")
    w = torch.zeros(8, 100, device="cuda")
    with pytest.raises(GivePoseHipError):
        o.gemm(x, w, torch.empty(8, 8, device="cuda"))      # K not a multiple of 32


# ----------------------------------------------------------------------------------------------- DCNv3
@pytest.mark.parametrize("name", ["dcnv3_s1", "dcnv3_s2_B1", "dcnv3_s2_B4", "dcnv3_s2_B5"])
def test_dcnv3_golden_fp32(golden, name):
    """Drop-in op (reference pybind signature) against vectors produced by the reference itself."""
    o = ops()
    z = golden(name)
    K, s, p, d, G, D, rc = (int(v) for v in z["params"])
    cu = lambda a: torch.from_numpy(a).cuda()
    out = o.dcnv3_forward(cu(z["input"]), cu(z["offset"]), cu(z["mask"]), K, K, s, s, p, p, d, d, G, D, float(z["offset_scale"]), 256, rc)
    assert tuple(out.shape) == z["expected"].shape
    assert np.abs(out.cpu().numpy() - z["expected"]).max() < 5e-6


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("rc", [0, 1])
def test_dcnv3_vs_oracle(dt, rc):
    from oracle.posenet_ref import dcnv3_forward_ref
    o = ops()
    N, H, G, D, K = 3, 12, 4, 64, 3
    P = K * K - rc
    x = q(rnd(N, H, H, G * D, seed=20), dt)
    off = q(rnd(N, H, H, G * P * 2, seed=21, scale=2.0), dt)
    m = torch.softmax(rnd(N, H, H, G, P, seed=22), -1).reshape(N, H, H, G * P)
    m = q(m, dt)
    for stride in (1, 2):
        ref = dcnv3_forward_ref(x, off, m, K, stride, 1, 1, G, D, 1.0, rc)
        out = o.dcnv3_forward(x.to("cuda", dt), off.to("cuda", dt), m.to("cuda", dt), K, K, stride, stride, 1, 1, 1, 1, G, D, 1.0, 256, rc)
        assert rel_err(out, ref) < (1e-5 if dt == torch.float32 else 2e-3)


def test_dcnv3_fused_softmax_and_strided_om():
    """mask logits + fp32 offset/mask rows of a single (rows,108) GEMM output, as PoseNet uses the op."""
    from oracle.posenet_ref import dcnv3_forward_ref
    o = ops()
    N, H, G, D, K, P = 4, 16, 4, 64, 3, 9
    Ho = H // 2
    x = rnd(N, H, H, 256, seed=23)
    om = rnd(N * Ho * Ho, 108, seed=24, scale=1.5)
    off, logits = om[:, :72], om[:, 72:]
    mask = torch.softmax(logits.reshape(-1, G, P), -1).reshape(-1, G * P)
    ref = dcnv3_forward_ref(x, off.contiguous(), mask.contiguous(), K, 2, 1, 1, G, D, 1.0, 0)
    for dt in DT:
        out = torch.empty(N, Ho, Ho, 256, dtype=dt, device="cuda")
        omd = om.cuda()
        o.dcnv3_forward_into(x.to("cuda", dt), omd, omd[:, 72:], out, K, 2, 1, 1, G, D, 1.0, off_ld=108, mask_ld=108, mask_is_logits=True)
        refq = dcnv3_forward_ref(q(x, dt), off.contiguous(), mask.contiguous(), K, 2, 1, 1, G, D, 1.0, 0)
        assert rel_err(out, refq) < (1e-5 if dt == torch.float32 else 2e-3)
    assert ref.shape == (N, Ho, Ho, 256)


@pytest.mark.parametrize("N,H,off_scale,logits,om32", [(5, 16, 1.5, True, True), (3, 32, 4.0, True, True), (2, 64, 0.0, True, True), (4, 16, 12.0, False, False), (1, 8, 2.0, False, True)])
def test_dcnv3_sixteen_bytes_per_lane_kernel_is_bitwise_the_eight_byte_one(N, H, off_scale, logits, om32):
    """dcnv3_wave8_kernel (round 6: 8 fp16 channels per lane, a 16-lane row = one pixel x TWO groups) against dcnv3_wave_kernel (4 channels per lane,
    GP_DCN_WAVE8=0 -- the switch is read per call): same per-channel arithmetic in the same order, so the outputs are the same bits; against the oracle as
    well.  Offsets from 0 (every tap on the grid) to 12 pixels (most taps outside the map: zero corners, clamped addresses); masks as logits (softmax in
    the kernel) and as weights; fp32 and fp16 offset / mask rows; 8 x 8 .. 64 x 64 maps (stride 2: output 4 x 4 .. 32 x 32)."""
    from oracle.posenet_ref import dcnv3_forward_ref
    o = ops()
    G, D, K, P = 4, 64, 3, 9
    Ho = H // 2
    x = q(rnd(N, H, H, 256, seed=230), torch.float16)
    om = rnd(N * Ho * Ho, 108, seed=231, scale=1.0)
    om[:, :72] *= off_scale
    if not logits:
        om[:, 72:] = torch.softmax(om[:, 72:].reshape(-1, G, P), -1).reshape(-1, G * P)
    omd = om.cuda() if om32 else q(om, torch.float16).cuda().half()
    outs = {}
    for arm, env in (("0", {"GP_DCN_WAVE8": "0"}), ("dpp", {"GP_DCN_LDSBC": "1"}), ("1", {})):     # 8 bytes per lane | 16 bytes, LDS records (measurement arm) | 16 bytes, DPP broadcasts (the default)
        os.environ.update(env)
        try:
            out = torch.full((N, Ho, Ho, 256), float("nan"), dtype=torch.float16, device="cuda")
            o.dcnv3_forward_into(x.cuda().half(), omd, omd[:, 72:], out, K, 2, 1, 1, G, D, 1.0, off_ld=108, mask_ld=108, mask_is_logits=logits)
            outs[arm] = out
        finally:
            for k in env:
                os.environ.pop(k, None)
    assert torch.equal(outs["0"], outs["1"]), float((outs["0"].float() - outs["1"].float()).abs().max())
    assert torch.equal(outs["dpp"], outs["1"]), float((outs["dpp"].float() - outs["1"].float()).abs().max())
    omr = omd.float().cpu()
    mask = torch.softmax(omr[:, 72:].reshape(-1, G, P), -1).reshape(-1, G * P) if logits else omr[:, 72:]
    ref = dcnv3_forward_ref(x, omr[:, :72].contiguous(), mask.contiguous(), K, 2, 1, 1, G, D, 1.0, 0)
    assert rel_err(outs["1"], ref) < 2e-3


def test_dcnv3_generic_geometry_and_errors():
    from oracle.posenet_ref import dcnv3_forward_ref
    from givepose_amd._lib import GivePoseHipError
    o = ops()
    N, H, G, D, K = 2, 9, 3, 8, 5       # odd sizes, G*D != 256 -> generic kernel
    x, off = rnd(N, H, H, G * D, seed=25), rnd(N, H, H, G * K * K * 2, seed=26, scale=1.5)
    m = torch.rand(N, H, H, G * K * K, generator=torch.Generator().manual_seed(27))
    ref = dcnv3_forward_ref(x, off, m, K, 1, 2, 1, G, D, 1.5, 0)
    out = o.dcnv3_forward(x.cuda(), off.cuda(), m.cuda(), K, K, 1, 1, 2, 2, 1, 1, G, D, 1.5, 256, 0)
    assert rel_err(out, ref) < 1e-5
    with pytest.raises(RuntimeError):      # reference: AT_ASSERTM contiguous (dcnv3_cuda.cu:29-31)
        o.dcnv3_forward(x.cuda().permute(0, 2, 1, 3), off.cuda(), m.cuda(), K, K, 1, 1, 2, 2, 1, 1, G, D, 1.5, 256, 0)
    with pytest.raises(RuntimeError):      # reference: CUDA tensors only (dcnv3_cuda.cu:32-34)
        o.dcnv3_forward(x, off.cuda(), m.cuda(), K, K, 1, 1, 2, 2, 1, 1, G, D, 1.5, 256, 0)
    with pytest.raises(GivePoseHipError):  # reference: batch % im2col_step (dcnv3_cuda.cu:46-49)
        x3 = torch.zeros(3, 4, 4, 8, device="cuda")
        o.dcnv3_forward(x3, torch.zeros(3, 4, 4, 18, device="cuda"), torch.zeros(3, 4, 4, 9, device="cuda"), 3, 3, 1, 1, 1, 1, 1, 1, 1, 8, 1.0, 2, 0)


# ----------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("C,H,KS", [(128, 16, 7), (256, 32, 7), (512, 16, 7), (512, 8, 7), (1024, 8, 7), (256, 16, 3)])
def test_dwconv_ln(dt, C, H, KS):
    o = ops()
    B = 2
    x = q(rnd(B, C, H, H, seed=30), dt)
    w, b = q(rnd(C, 1, KS, KS, seed=31, scale=1.0 / KS), dt), rnd(C, seed=32, scale=0.1)
    lw, lb = 1 + 0.1 * rnd(C, seed=33), 0.1 * rnd(C, seed=34)
    y = F.conv2d(x, w, b, padding=KS // 2, groups=C).permute(0, 2, 3, 1)
    ref = F.layer_norm(y, (C,), lw, lb, 1e-6)
    act = o.ACT_GELU if KS == 3 else o.ACT_NONE
    if KS == 3:
        ref = F.gelu(ref)
    xd = x.permute(0, 2, 3, 1).contiguous().to("cuda", dt)
    out = torch.zeros(B, H, H, C, dtype=dt, device="cuda")
    o.dwconv_ln(xd, w.reshape(C, KS * KS).t().contiguous().to("cuda", dt), b.cuda(), lw.cuda(), lb.cuda(), out, KS, act=act)
    assert rel_err(out, ref) < TOL[dt]
    if KS == 7 and H % 8 == 0:    # act code 104 forces the LDS-tiled VALU kernel (picked by itself only for >= 192 tiles)
        out.zero_()
        o.dwconv_ln(xd, w.reshape(C, KS * KS).t().contiguous().to("cuda", dt), b.cuda(), lw.cuda(), lb.cuda(), out, KS, act=104)
        assert rel_err(out, ref) < TOL[dt]
    # prefix mode (DCNv3 consumes only the first quarter of the full-resolution grid)
    n = B * H * H // 4
    out2 = torch.full((B * H * H, C), 7.0, dtype=dt, device="cuda")
    o.dwconv_ln(xd, w.reshape(C, KS * KS).t().contiguous().to("cuda", dt), b.cuda(), lw.cuda(), lb.cuda(), out2, KS, act=act, n_pixels=n)
    assert rel_err(out2[:n], ref.reshape(-1, C)[:n]) < TOL[dt]
    assert float((out2[n:] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("C,H,KS,B,offset", [(256, 32, 7, 3, 0.0), (512, 16, 7, 13, 0.0), (512, 16, 7, 128, 0.0), (128, 64, 7, 1, 0.0),
                                             (256, 64, 3, 2, 0.0), (1024, 8, 7, 32, 0.0), (512, 16, 7, 13, 5.0), (256, 32, 7, 3, 5.0)])
def test_dwconv_ln_fp16_kernels_by_grid_size(C, H, KS, B, offset):
    """test_dwconv_ln runs B = 2, where gp_dwconv_ln's routing (dw_mfma_min_wgs: 33 workgroups at C = 256, 52 at C = 512) sends C = 256 / 512
    to the 2-pixel strip kernel: the round-4 MFMA tiling (two output rows x eight channels per MFMA, row-parity swizzle, one-pass
    LayerNorm) had no operator-level test for NSLAB = 2 / 4 (round-4 advice).  Here: batches that reach dwconv7_ln_mfma_kernel<2>, <4, NBUF 2>
    (52 ... 511 workgroups) and <4, NBUF 1> (>= 512), the fp16 8-pixel strip kernel with >= 128 workgroups (dw3x3 at 64 x 64, C = 1024), and
    -- offset 5 -- channel vectors whose mean is ~50 x their standard deviation: the MFMA kernel's one-pass variance E[x^2] - mean^2 (fp32) loses
    (mean / std)^2 of relative precision; at 50 x that is 2.5e3 x 1e-7 -- fine -- at 500 x (offset 50 here: measured 4.7e-2 relative error of the output)
    it is not: the price of the single statistics round, bounded here at the ratio a trained ConvNeXt can plausibly produce."""
    o = ops()
    dt = torch.float16
    x = q(rnd(B, C, H, H, seed=130), dt)
    w = q(rnd(C, 1, KS, KS, seed=131, scale=(0.02 if offset else 1.0) / KS), dt)
    b = rnd(C, seed=132, scale=0.1) + offset
    lw, lb = 1 + 0.1 * rnd(C, seed=133), 0.1 * rnd(C, seed=134)
    y = F.conv2d(x, w, b, padding=KS // 2, groups=C).permute(0, 2, 3, 1)
    if offset:
        sd, mu = y.std(-1), y.mean(-1).abs()
        assert float((mu / sd).median()) > 30      # the case really is |mean| >> std
    ref = F.layer_norm(y, (C,), lw, lb, 1e-6)
    act = o.ACT_GELU if KS == 3 else o.ACT_NONE
    if KS == 3:
        ref = F.gelu(ref)
    xd = x.permute(0, 2, 3, 1).contiguous().to("cuda", dt)
    out = torch.zeros(B, H, H, C, dtype=dt, device="cuda")
    o.dwconv_ln(xd, w.reshape(C, KS * KS).t().contiguous().to("cuda", dt), b.cuda(), lw.cuda(), lb.cuda(), out, KS, act=act)
    assert rel_err(out, ref) < (2 * TOL[dt] if offset else TOL[dt]), rel_err(out, ref)
    out2 = torch.zeros_like(out)      # bitwise repeatable
    o.dwconv_ln(xd, w.reshape(C, KS * KS).t().contiguous().to("cuda", dt), b.cuda(), lw.cuda(), lb.cuda(), out2, KS, act=act)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("B,H,W,gelu", [(8, 16, 16, True), (4, 32, 32, True), (6, 32, 32, True), (1, 64, 64, True), (4, 32, 16, True), (16, 64, 64, True), (3, 16, 48, True)])
def test_dwconv3_ln_tile_kernel_is_bitwise_the_strip_kernel(B, H, W, gelu):
    """dwconv3_ln_tile_kernel (round 6: the DCNv3 prefix kernel dw3x3 -> LN -> GELU with its 18 x 6 halo window in LDS; act code 120 + act forces it) against the
    strip kernel on the same prefix (the first quarter of the flat pixel rows, ops_dcnv3/modules/dcnv3.py:318-356 + SURVEY 0.3): same arithmetic in the same
    order, so the same bits; and against the fp32 reference.  Prefixes that are whole images (B % 4 == 0), that end inside an image (B = 6: 1.5 images; B = 1 and
    3: a quarter / three quarters of one -- the halo row below the prefix is real data), a non-square map, W = 48 (three column tiles).  GELU only: without an
    activation the launch stays on the strip kernel (forcing the tile kernel is refused)."""
    o = ops()
    C, dt = 256, torch.float16
    x = q(rnd(B, C, H, W, seed=240), dt)
    w, b = q(rnd(C, 1, 3, 3, seed=241, scale=1.0 / 3), dt), rnd(C, seed=242, scale=0.1)
    lw, lb = 1 + 0.1 * rnd(C, seed=243), 0.1 * rnd(C, seed=244)
    n = B * H * W // 4
    ref = F.layer_norm(F.conv2d(x, w, b, padding=1, groups=C).permute(0, 2, 3, 1), (C,), lw, lb, 1e-6)
    if gelu:
        ref = F.gelu(ref)
    ref = ref.reshape(-1, C)[:n]
    xd = x.permute(0, 2, 3, 1).contiguous().to("cuda", dt)
    wd = w.reshape(C, 9).t().contiguous().to("cuda", dt)
    act = o.ACT_GELU if gelu else o.ACT_NONE
    tile = torch.full((n, C), float("nan"), dtype=dt, device="cuda")
    o.dwconv_ln(xd, wd, b.cuda(), lw.cuda(), lb.cuda(), tile, 3, act=120 + act, n_pixels=n)       # 16 x 4 tiles
    assert rel_err(tile, ref) < TOL[dt], rel_err(tile, ref)
    tile2 = torch.full((n, C), float("nan"), dtype=dt, device="cuda")
    o.dwconv_ln(xd, wd, b.cuda(), lw.cuda(), lb.cuda(), tile2, 3, act=125 + act, n_pixels=n)      # 16 x 2 tiles (the routed form)
    assert torch.equal(tile, tile2)
    from givepose_amd._lib import GivePoseHipError
    with pytest.raises(GivePoseHipError):
        o.dwconv_ln(xd, wd, b.cuda(), lw.cuda(), lb.cuda(), torch.empty_like(tile), 3, act=120 + o.ACT_NONE, n_pixels=n)
    if n // 64 < 256:                      # below 256 tiles the routing itself takes the strip kernel
        strip = torch.full((n, C), float("nan"), dtype=dt, device="cuda")
        o.dwconv_ln(xd, wd, b.cuda(), lw.cuda(), lb.cuda(), strip, 3, act=act, n_pixels=n)
        assert torch.equal(tile, strip), float((tile.float() - strip.float()).abs().max())


@pytest.mark.parametrize("C,H,B,offset,W", [(512, 16, 3, 0.0, 16), (512, 16, 128, 0.0, 16), (256, 16, 5, 0.0, 16), (128, 16, 2, 0.0, 16), (512, 8, 2, 0.0, 16),
                                            (512, 32, 2, 0.0, 16), (256, 24, 1, 0.0, 16), (512, 16, 7, 5.0, 16),
                                            (128, 64, 2, 0.0, 64), (256, 32, 3, 0.0, 32), (128, 24, 1, 0.0, 48), (256, 8, 2, 0.0, 32), (128, 64, 1, 5.0, 64),
                                            (1024, 8, 2, 0.0, 8), (1024, 8, 6, 5.0, 8), (1024, 8, 64, 0.0, 8), (1024, 16, 2, 0.0, 8)])
def test_dwconv_ln_tall_tiles(C, H, B, offset, W):
    """dwconv7_ln_tall_kernel (round 6: 16 x 8 tiles of 16-pixel-wide maps, zero pixels in LDS instead of a halo, three-stage slab ring):
    forced with act code 110 at grids the routing would not send it (it takes over from 130 half-image workgroups), against the fp32
    reference and against the 16 x 4 / strip kernels on the same input; H = 8 (one tile, zero rows above AND below), H = 16 (the
    product shape), H = 24 / 32 (interior tiles with 14 real halo rows: the five-instruction DMA plan); offset 5: |mean| ~ 50 std."""
    # W > 16: the column-halo form (stages 0 / 1 of the trunk): edge tiles on every side, interior tiles (64 x 64), a non-square map, a one-tile-high map
    # W = 8, C = 1024 (stage 3; act code 113): two images side by side in one tile -- neither may see the other's columns; H = 16: interior-row tiles
    o = ops()
    force = 113 if W == 8 else 110
    dt = torch.float16
    x = q(rnd(B, C, H, W, seed=140), dt)
    w = q(rnd(C, 1, 7, 7, seed=141, scale=(0.02 if offset else 1.0) / 7), dt)
    b = rnd(C, seed=142, scale=0.1) + offset
    lw, lb = 1 + 0.1 * rnd(C, seed=143), 0.1 * rnd(C, seed=144)
    ref = F.layer_norm(F.conv2d(x, w, b, padding=3, groups=C).permute(0, 2, 3, 1), (C,), lw, lb, 1e-6)
    xd = x.permute(0, 2, 3, 1).contiguous().to("cuda", dt)
    wd = w.reshape(C, 49).t().contiguous().to("cuda", dt)
    out = torch.zeros(B, H, W, C, dtype=dt, device="cuda")
    o.dwconv_ln(xd, wd, b.cuda(), lw.cuda(), lb.cuda(), out, 7, act=force)
    assert rel_err(out, ref) < (2 * TOL[dt] if offset else TOL[dt]), rel_err(out, ref)
    old = torch.zeros_like(out)
    o.dwconv_ln(xd, wd, b.cuda(), lw.cuda(), lb.cuda(), old, 7)       # below 130 workgroups the routing takes the older kernels
    if B * (H // 8) * (W // 16) < 130:
        assert float((out.float() - old.float()).abs().max()) <= 4e-3 * max(1.0, float(ref.abs().max()))
    out2 = torch.zeros_like(out)      # bitwise repeatable
    o.dwconv_ln(xd, wd, b.cuda(), lw.cuda(), lb.cuda(), out2, 7, act=force)
    nz = (out != out2).nonzero()
    assert nz.shape[0] == 0, (nz.shape[0], nz[:8].tolist(), float((out.float() - out2.float()).abs().max()), float((out2.float() - ref.cuda()).abs().max()), float((out.float() - ref.cuda()).abs().max()))
    if W == 8:                        # the pair form on quarter-image tiles (TH = 2; act code 114): same arithmetic per pixel, same bits
        out2r = torch.zeros_like(out)
        o.dwconv_ln(xd, wd, b.cuda(), lw.cuda(), lb.cuda(), out2r, 7, act=114)
        assert torch.equal(out, out2r), float((out.float() - out2r.float()).abs().max())
    if W == 16:                       # the quarter-image form (TH = 4; act code 112): what a launch of 33 .. 64 crops at stage 2 takes
        out4 = torch.zeros_like(out)
        o.dwconv_ln(xd, wd, b.cuda(), lw.cuda(), lb.cuda(), out4, 7, act=112)
        assert rel_err(out4, ref) < (2 * TOL[dt] if offset else TOL[dt]), rel_err(out4, ref)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("C", [128, 256, 512])
def test_layernorm(dt, C):
    o = ops()
    x = q(rnd(300, C, seed=35) * 2 + 0.5, dt)
    lw, lb = 1 + 0.1 * rnd(C, seed=36), 0.1 * rnd(C, seed=37)
    out = torch.empty(300, C, dtype=dt, device="cuda")
    o.layernorm(x.to("cuda", dt), lw.cuda(), lb.cuda(), out)
    assert rel_err(out, F.layer_norm(x, (C,), lw, lb, 1e-6)) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("C,HW,act", [(256, 1024, "gelu"), (256, 256, "relu"), (128, 64, "relu"), (256, 64, "gelu")])
def test_groupnorm(dt, C, HW, act):
    o = ops()
    B = 3
    x = q(rnd(B, C, HW, seed=38) * 1.5 + 0.3, dt)
    gw, gb = 1 + 0.1 * rnd(C, seed=39), 0.1 * rnd(C, seed=40)
    ref = F.group_norm(x, 32, gw, gb, 1e-5)
    ref = (F.gelu(ref) if act == "gelu" else F.relu(ref)).permute(0, 2, 1)
    xd = x.permute(0, 2, 1).contiguous().to("cuda", dt)
    partial = torch.empty(B * o.groupnorm_chunks(B, HW) * 64, device="cuda")
    wide = torch.zeros(B, HW, 2 * C, dtype=dt, device="cuda")
    o.groupnorm(xd, gw.cuda(), gb.cuda(), wide[:, :, C:], 32, o.ACT_GELU if act == "gelu" else o.ACT_RELU, partial, ldy=2 * C)
    assert rel_err(wide[:, :, C:], ref) < TOL[dt]
    assert float(wide[:, :, :C].abs().max()) == 0.0
    o.groupnorm(xd, gw.cuda(), gb.cuda(), xd, 32, o.ACT_GELU if act == "gelu" else o.ACT_RELU, partial)   # in place
    assert rel_err(xd, ref) < TOL[dt]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("cfg", [dict(B=3, H=16, Cin=128, Cout=256, s=1), dict(B=2, H=16, Cin=64, Cout=128, s=2), dict(B=1, H=32, Cin=64, Cout=256, s=1)])
def test_conv_with_fused_groupnorm_stats(dt, cfg):
    """GroupNorm statistics produced by the conv epilogue (64-row chunks) == statistics pass, end result vs torch."""
    o = ops()
    B, H, Cin, Cout, s_ = (cfg[n] for n in ("B", "H", "Cin", "Cout", "s"))
    x = q(rnd(B, Cin, H, H, seed=70), dt)
    w = q(rnd(Cout, Cin, 3, 3, seed=71, scale=(Cin * 9) ** -0.5), dt)
    gw, gb = 1 + 0.1 * rnd(Cout, seed=72), 0.1 * rnd(Cout, seed=73)
    conv = F.conv2d(x, w, None, stride=s_, padding=1)
    ref = F.gelu(F.group_norm(q(conv, dt) if dt == torch.float16 else conv, 32, gw, gb, 1e-5)).permute(0, 2, 3, 1)
    Ho = conv.shape[-1]
    partial = torch.zeros(B * max(Ho * Ho // 64, 4) * 64, device="cuda")
    out = o.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to("cuda", dt), w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to("cuda", dt),
                        3, 3, s_, 1, gn=(partial, 32, Ho * Ho))
    ov = out.view(B, Ho * Ho, Cout)
    o.groupnorm(ov, gw.cuda(), gb.cuda(), ov, 32, o.ACT_GELU, partial, fused_stats=True)
    assert rel_err(out, ref) < (TOL[dt] if dt == torch.float32 else 6e-3)


# ----------------------------------------------------------------------------------------------- small ops
@pytest.mark.parametrize("dt", DT)
def test_stem(dt):
    o = ops()
    B = 2
    img = rnd(B, 3, 64, 128 if dt == torch.float32 else 256, seed=41)
    w, b = rnd(128, 3, 4, 4, seed=42, scale=48 ** -0.5), rnd(128, seed=43, scale=0.1)
    lw, lb = 1 + 0.1 * rnd(128, seed=44), 0.1 * rnd(128, seed=45)
    ref = F.layer_norm(F.conv2d(img, w, b, stride=4).permute(0, 2, 3, 1), (128,), lw, lb, 1e-6)
    out = torch.empty(B, 16, img.shape[-1] // 4, 128, dtype=dt, device="cuda")
    o.convnext_stem(img.cuda(), w.reshape(128, 48).t().contiguous().cuda(), b.cuda(), lw.cuda(), lb.cuda(), out)
    assert rel_err(out, ref) < TOL[dt]


def test_stem_mfma_forms_fp32_accurate():
    """fp16-output stem on MFMA (fp16 hi + lo split of image and taps): four rows per workgroup (H/4 % 4 == 0) and one row
    per workgroup (H/4 % 4 != 0, or -- since round 5 -- fewer than 256 four-row workgroups: the first three cases; the last one is the four-row form); against the fp64 formula
    the only error left is the fp16 rounding of the output."""
    o = ops()
    for (B, H) in ((2, 64), (1, 72), (3, 8), (16, 256)):
        img = rnd(B, 3, H, 256, seed=141)
        w, b = rnd(128, 3, 4, 4, seed=142, scale=48 ** -0.5), rnd(128, seed=143, scale=0.1)
        lw, lb = 1 + 0.1 * rnd(128, seed=144), 0.1 * rnd(128, seed=145)
        ref = F.layer_norm(F.conv2d(img.double(), w.double(), b.double(), stride=4).permute(0, 2, 3, 1), (128,), lw.double(), lb.double(), 1e-6)
        out = torch.empty(B, H // 4, 64, 128, dtype=torch.float16, device="cuda")
        o.convnext_stem(img.cuda(), w.reshape(128, 48).t().contiguous().cuda(), b.cuda(), lw.cuda(), lb.cuda(), out)
        d = (out.cpu().double() - ref).abs()
        assert float(d.max()) <= 2.0 ** -10 * max(1.0, float(ref.abs().max())) and float(d.mean()) < 2.5e-4, (B, H, float(d.max()), float(d.mean()))


@pytest.mark.parametrize("HW", [4096, 1024, 1600])
def test_groupnorm_apply_xyz(HW):
    """GroupNorm apply + GELU + the 1x1 out layer in one pass from the conv epilogue's 64-row statistics (fp16 C = 256: the MFMA
    form with hi/lo split operands; fp32: the VALU form): both write the NCHW and the (rows, 4) maps, fp32-accurate."""
    o = ops()
    from givepose_amd._lib import ACT_GELU
    B, C, G = 3, 256, 32
    for dt in (torch.float16, torch.float32):
        x = q(rnd(B, HW, C, seed=151), dt)
        gw, gb = 1 + 0.1 * rnd(C, seed=152), 0.1 * rnd(C, seed=153)
        ow, ob = rnd(3, C, seed=154, scale=C ** -0.5), rnd(3, seed=155, scale=0.1)
        xf = x.view(B, HW // 64, 64, G, C // G)
        part = torch.stack([xf.sum((2, 4)), (xf * xf).sum((2, 4))], -1).contiguous().view(-1).cuda()
        y = F.gelu(F.group_norm(x.double().permute(0, 2, 1), G, gw.double(), gb.double(), 1e-5).permute(0, 2, 1))
        ref = y @ ow.double().t() + ob.double()                                                    # (B, HW, 3)
        nchw, nhwc4 = torch.empty(B, 3, HW, device="cuda"), torch.full((B * HW, 4), 7.0, device="cuda")
        o.groupnorm_apply_xyz(x.to("cuda", dt), gw.cuda(), gb.cuda(), ow.cuda(), ob.cuda(), nchw, nhwc4, G, ACT_GELU, part)
        d = (nchw.cpu().double().permute(0, 2, 1) - ref).abs()
        assert float(d.max()) < 2e-4 and float(d.mean()) < 3e-5, (dt, HW, float(d.max()), float(d.mean()))
        assert torch.equal(nhwc4.view(B, HW, 4)[..., :3].permute(0, 2, 1), nchw) and float(nhwc4[:, 3].abs().max()) == 0.0
        # GP_ACT_PACKED16 (round 5; what PoseNet's fp16 mode asks for): affine + GELU on packed fp16 -- one fp16 rounding per activation instead of fp32 accuracy
        # (measured 2.7e-4 mean / 1.6e-3 max at 64 x 64); the flag is ignored for fp32 storage (same bits as without it)
        n2, h2 = torch.empty_like(nchw), torch.full_like(nhwc4, 7.0)
        o.groupnorm_apply_xyz(x.to("cuda", dt), gw.cuda(), gb.cuda(), ow.cuda(), ob.cuda(), n2, h2, G, ACT_GELU, part, packed16=True)
        d2 = (n2.cpu().double().permute(0, 2, 1) - ref).abs()
        if dt == torch.float16:
            assert float(d2.max()) < 4e-3 and float(d2.mean()) < 6e-4 and not torch.equal(n2, nchw), (HW, float(d2.max()), float(d2.mean()))
        else:
            assert torch.equal(n2, nchw)
        assert torch.equal(h2.view(B, HW, 4)[..., :3].permute(0, 2, 1), n2) and float(h2[:, 3].abs().max()) == 0.0


@pytest.mark.parametrize("ratio", [5.0, 20.0, 50.0])
def test_groupnorm_apply_xyz_packed16_at_large_group_means(ratio):
    """GP_ACT_PACKED16 rounds the per-channel scale and SHIFT (b - mean * s) to fp16 before the v_pk_fma, so its absolute error grows like
    2^-11 * |group mean| / std of the normalised value (round-5 advice: the synthetic weights of the end-to-end tests have group means near
    zero).  Here the groups' |mean| is `ratio` x their std: the packed form against the fp32-accurate form of the same kernel, bounded by that
    model (x 256^(1/2) channels x |out-layer weight| through the 1x1 layer) -- and the fp32-accurate form itself against float64."""
    o = ops()
    from givepose_amd._lib import ACT_GELU
    B, C, G, HW = 2, 256, 32, 1024
    dt = torch.float16
    x = q(rnd(B, HW, C, seed=161) + ratio, dt)          # std 1 around `ratio`
    gw, gb = 1 + 0.1 * rnd(C, seed=162), 0.1 * rnd(C, seed=163)
    ow, ob = rnd(3, C, seed=164, scale=C ** -0.5), rnd(3, seed=165, scale=0.1)
    xf = x.view(B, HW // 64, 64, G, C // G)
    mu, sd = xf.mean((2, 4)), xf.std((2, 4))
    assert float((mu.abs() / sd).min()) > 0.8 * ratio
    part = torch.stack([xf.sum((2, 4)), (xf * xf).sum((2, 4))], -1).contiguous().view(-1).cuda()
    y = F.gelu(F.group_norm(x.double().permute(0, 2, 1), G, gw.double(), gb.double(), 1e-5).permute(0, 2, 1))
    ref = y @ ow.double().t() + ob.double()
    outs = []
    for packed in (False, True):
        nchw, nhwc4 = torch.empty(B, 3, HW, device="cuda"), torch.full((B * HW, 4), 7.0, device="cuda")
        o.groupnorm_apply_xyz(x.to("cuda", dt), gw.cuda(), gb.cuda(), ow.cuda(), ob.cuda(), nchw, nhwc4, G, ACT_GELU, part, packed16=packed)
        outs.append(nchw.cpu().double().permute(0, 2, 1))
    e32, e16 = float((outs[0] - ref).abs().max()), float((outs[1] - ref).abs().max())
    # fp32-accurate form: the statistics E[x^2] - mean^2 lose (mean / std)^2 of fp32's precision: 2.5e3 x 6e-8 at ratio 50
    assert e32 < 2e-4 + 3e-7 * ratio * ratio, (ratio, e32)
    # packed form: per activation ~2^-11 x (1 + ratio) from the rounded shift; 256 of them with random signs through |w| ~ 1/16
    bound = 4e-3 + 2.0 ** -11 * (1 + ratio) * 1.5
    print(f"|mean| / std = {ratio}: fp32-accurate form {e32:.2e}, packed fp16 form {e16:.2e} (bound {bound:.2e})")
    assert e16 < bound, (ratio, e16, bound)


@pytest.mark.parametrize("dt", DT)
def test_upsample_and_col2im(dt):
    o = ops()
    B, H, C = 2, 8, 256
    x = q(rnd(B, C, H, H, seed=46), dt)
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    out = torch.empty(B, 2 * H, 2 * H, C, dtype=dt, device="cuda")
    o.upsample_bilinear2x(x.permute(0, 2, 3, 1).contiguous().to("cuda", dt), out)
    assert rel_err(out, ref) < TOL[dt]
    # ConvTranspose2d(k3,s2,p1,op1) = GEMM + col2im
    Cin = 128
    xi = q(rnd(B, Cin, H, H, seed=47), dt)
    wt = q(rnd(Cin, C, 3, 3, seed=48, scale=(Cin * 9 / 4) ** -0.5), dt)
    ref = F.conv_transpose2d(xi, wt, None, stride=2, padding=1, output_padding=1).permute(0, 2, 3, 1)
    cols = torch.empty(B * H * H, 9 * C, dtype=torch.float32, device="cuda")
    o.gemm(xi.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to("cuda", dt), wt.permute(2, 3, 1, 0).reshape(9 * C, Cin).contiguous().to("cuda", dt), cols)
    out = torch.empty(B, 2 * H, 2 * H, C, dtype=dt, device="cuda")
    o.deconv_col2im(cols, out, B, H, H, C)
    assert rel_err(out, ref) < TOL[dt]
    if dt == torch.float16:      # round 5 (PoseNetConfig.deconv_cols_f16): fp16 column matrix, summed in fp32 by the col2im (GP_COLS_F16)
        cols16 = torch.empty(B * H * H, 9 * C, dtype=dt, device="cuda")
        o.gemm(xi.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to("cuda", dt), wt.permute(2, 3, 1, 0).reshape(9 * C, Cin).contiguous().to("cuda", dt), cols16)
        out16 = torch.full_like(out, 7.0)
        o.deconv_col2im(cols16, out16, B, H, H, C)
        assert rel_err(out16, ref) < TOL[dt]
        assert rel_err(out16, out.float().cpu()) < 2e-3


@pytest.mark.parametrize("B,H,W,C", [(3, 16, 16, 256), (2, 32, 32, 256), (2, 8, 24, 128), (1, 5, 7, 64)])
def test_groupnorm_upsample2x_bitwise_vs_two_passes(B, H, W, C):
    """gp_groupnorm_upsample2x (TopDownXyzHead's GN + GELU + Upsample, xyz_head.py:250-264) against the two kernels it replaces
    (bit for bit: same statistics, same fp16 rounding of the normalised tensor, same blend) and against torch in fp32."""
    o = ops()
    G = 32
    x = (rnd(B, H, W, C, seed=140) * 1.7 + 0.3).to("cuda", torch.float16)
    gw, gb = (1 + 0.2 * rnd(C, seed=141)).cuda(), (0.1 * rnd(C, seed=142)).cuda()
    partial = torch.zeros(1 << 16, device="cuda")
    chunks = H * W // 64 if H * W % 64 == 0 else 0
    lib = __import__("givepose_amd._lib", fromlist=["x"]).load()
    P, st = (lambda t: ctypes.c_void_p(t.data_ptr())), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.gp_groupnorm_stats(P(x), P(partial), B, H * W, C, G, 1, st) == 0
    if chunks:      # the product path: statistics in 64-row chunks, as the producing conv leaves them
        part64 = torch.zeros(1 << 16, device="cuda")
        xf = x.float().view(B, chunks, 64, G, C // G)
        part64[:B * chunks * G * 2] = torch.stack([xf.sum((2, 4)), (xf * xf).sum((2, 4))], -1).reshape(-1)
        partial = part64
    mid = torch.empty_like(x)
    assert lib.gp_groupnorm_apply(P(x), P(partial), P(gw), P(gb), P(mid), B, H * W, C, G, 1e-5, 1, C, chunks, 1, st) == 0
    two = o.upsample_bilinear2x(mid, torch.empty(B, 2 * H, 2 * W, C, dtype=torch.float16, device="cuda"))
    one = torch.full((B, 2 * H, 2 * W, C), float("nan"), dtype=torch.float16, device="cuda")
    assert lib.gp_groupnorm_upsample2x(P(x), P(partial), P(gw), P(gb), P(one), B, H, W, C, G, 1e-5, 1, chunks, 1, st) == 0, lib.gp_last_error()
    assert torch.equal(one, two), float((one.float() - two.float()).abs().max())
    xr = x.float().cpu().permute(0, 3, 1, 2)
    ref = F.interpolate(F.gelu(F.group_norm(xr, G, gw.cpu(), gb.cpu(), 1e-5)), scale_factor=2, mode="bilinear", align_corners=True)
    assert rel_err(one, ref.permute(0, 2, 3, 1)) < TOL[torch.float16]
    # fp32 storage is refused loudly (the fp32 modes keep the two passes)
    assert lib.gp_groupnorm_upsample2x(P(x), P(partial), P(gw), P(gb), P(one), B, H, W, C, G, 1e-5, 1, chunks, 0, st) != 0


@pytest.mark.parametrize("dt", DT)
def test_xyz_out_pointwise_smallcin(dt):
    o = ops()
    B, R, C = 2, 16, 256
    x = q(rnd(B, R * R, C, seed=49), dt)
    w, b = rnd(3, C, seed=50, scale=C ** -0.5), rnd(3, seed=51)
    ref = x @ w.t() + b                                           # (B, HW, 3)
    nchw, nhwc4 = torch.empty(B, 3, R, R, device="cuda"), torch.empty(B * R * R, 4, device="cuda")
    o.xyz_out_layer(x.to("cuda", dt), w.cuda(), b.cuda(), nchw, nhwc4)
    assert rel_err(nchw.reshape(B, 3, -1).permute(0, 2, 1), ref) < 2e-5
    assert rel_err(nhwc4.reshape(B, -1, 4)[..., :3], ref) < 2e-5 and float(nhwc4[:, 3].abs().max()) == 0
    xyz = nhwc4.cpu()
    w3, b3 = rnd(256, 3, seed=52), rnd(256, seed=53)
    out = torch.empty(B * R * R, 256, dtype=dt, device="cuda")
    o.pointwise_k3(nhwc4, w3.cuda(), b3.cuda(), out)
    assert rel_err(out, xyz[:, :3] @ w3.t() + b3) < TOL[dt]
    # tiny-Cin 3x3 s2 convs
    coord = rnd(B, 2, R, R, seed=54)
    xin = torch.cat([xyz[:, :3].reshape(B, R, R, 3).permute(0, 3, 1, 2), coord], 1)
    w5 = rnd(128, 5, 3, 3, seed=55, scale=45 ** -0.5)
    out5 = torch.empty(B, R // 2, R // 2, 128, dtype=dt, device="cuda")
    o.pnp_conv1(nhwc4, coord.cuda(), w5.reshape(128, 45).t().contiguous().cuda(), out5, B, R)
    assert rel_err(out5, F.conv2d(xin, w5, None, stride=2, padding=1).permute(0, 2, 3, 1)) < TOL[dt]
    w3c = rnd(256, 3, 3, 3, seed=56, scale=27 ** -0.5)
    out3 = torch.empty(B, R // 2, R // 2, 256, dtype=dt, device="cuda")
    o.xyz_conv3x3_s2(nhwc4, w3c.reshape(256, 27).t().contiguous().cuda(), out3, B, R)
    assert rel_err(out3, F.conv2d(xin[:, :3], w3c, None, stride=2, padding=1).permute(0, 2, 3, 1)) < TOL[dt]


def test_smallcin_conv_mfma_path():
    """R = 64 (the PoseNet map size) takes the MFMA form of the tiny-Cin 3x3 s2 convs in fp16: fp32 inputs and taps split into
    fp16 hi + lo inside the kernel, so it must be as close to the fp32 formula as the VALU form (fp16 output rounding only)."""
    o = ops()
    B, R = 3, 64
    xyz, coord = rnd(B, R, R, 3, seed=57), rnd(B, 2, R, R, seed=58)
    nhwc4 = torch.cat([xyz, torch.zeros(B, R, R, 1)], -1).reshape(B * R * R, 4).cuda()
    xin = torch.cat([xyz.permute(0, 3, 1, 2), coord], 1)
    for (cin, cout, seed) in ((5, 128, 59), (3, 256, 60), (3, 128, 61)):
        w = rnd(cout, cin, 3, 3, seed=seed, scale=(9 * cin) ** -0.5)
        ref = F.conv2d(xin[:, :cin].double(), w.double(), None, stride=2, padding=1).permute(0, 2, 3, 1)
        out = torch.empty(B, R // 2, R // 2, cout, dtype=torch.float16, device="cuda")
        wp = w.reshape(cout, 9 * cin).t().contiguous().cuda()
        if cin == 5:
            o.pnp_conv1(nhwc4, coord.cuda(), wp, out, B, R)
        else:
            o.xyz_conv3x3_s2(nhwc4, wp, out, B, R)
        d = (out.cpu().double() - ref).abs()
        assert float(d.max()) <= 2.0 ** -10 * max(1.0, float(ref.abs().max())) and float(d.mean()) < 2e-4, (cin, cout, float(d.max()), float(d.mean()))


def test_size_head_golden(golden):
    """vs the reference SizeHead output (BN folded on the host) + mean-size residual."""
    from givepose_amd import synth
    o = ops()
    z = golden("size_head")
    sd = {k: torch.from_numpy(synth.synth_tensor("size_head." + k, s, 0)) for k, s in
          (("conv1.weight", (128, 1024, 1)), ("conv1.bias", (128,)), ("conv2.weight", (3, 128, 1)), ("conv2.bias", (3,)),
           ("bn1.weight", (128,)), ("bn1.bias", (128,)), ("bn1.running_mean", (128,)), ("bn1.running_var", (128,)))}
    sc = sd["bn1.weight"] / torch.sqrt(sd["bn1.running_var"] + 1e-5)
    w1 = (sd["conv1.weight"].squeeze(-1) * sc[:, None]).contiguous()
    b1 = (sd["conv1.bias"] - sd["bn1.running_mean"]) * sc + sd["bn1.bias"]
    x = torch.from_numpy(z["x"])
    B = x.shape[0]
    ms = torch.tensor([[0.1, 0.2, 0.3]] * B)
    out = torch.empty(B, 3, device="cuda")
    o.size_head(x.permute(0, 2, 3, 1).reshape(B, 64, 1024).contiguous().cuda(), w1.cuda(), b1.cuda(),
                sd["conv2.weight"].squeeze(-1).contiguous().cuda(), sd["conv2.bias"].cuda(), ms.cuda(), out, torch.empty(B * (128 + 1024), device="cuda"))
    ref = torch.from_numpy(z["expected"]) + ms / ms.norm(dim=1, keepdim=True)
    assert float((out.cpu() - ref).abs().max()) < 2e-5


@pytest.mark.parametrize("ds", ["CAMERA_Real", "wild6d"])
def test_pose_tail_golden(golden, ds):
    """fc_r/fc_t/fc_z + rot6d + allo->ego decode on device vs the reference's numpy/transforms3d path."""
    o = ops()
    z = golden("pose_decode_" + ds)
    B = z["d6"].shape[0]
    # feed identity-like heads so the tail's fc outputs equal the golden d6 / pred_t
    h = torch.zeros(B, 256)
    hz = torch.zeros(B, 256)
    h[:, :6] = torch.from_numpy(z["d6"])
    h[:, 6:8] = torch.from_numpy(z["pred_t"][:, :2])
    hz[:, 0] = torch.from_numpy(z["pred_t"][:, 2])
    wr, wt_, wz = torch.zeros(6, 256), torch.zeros(2, 256), torch.zeros(1, 256)
    for i in range(6):
        wr[i, i] = 1
    wt_[0, 6] = wt_[1, 7] = 1
    wz[0, 0] = 1
    W = {"fc_r.w": wr.cuda(), "fc_r.b": torch.zeros(6).cuda(), "fc_t.w": wt_.cuda(), "fc_t.b": torch.zeros(2).cuda(),
         "fc_z.w": wz.cuda(), "fc_z.b": torch.zeros(1).cuda()}
    outs = {k: torch.empty(B, n, device="cuda") for k, n in (("rot6d", 6), ("pred_t", 3), ("rot_allo", 9), ("rot_ego", 9), ("trans", 3))}
    cu = lambda k: torch.from_numpy(z[k]).cuda()
    o.pose_tail(h.cuda(), hz.cuda(), 256, W, cu("cam_K"), cu("bbox_center"), cu("resize_ratio"), cu("roi_wh"), ds == "wild6d", True, outs, B)
    assert np.abs(outs["rot_allo"].cpu().numpy().reshape(B, 3, 3) - z["rot_allo"]).max() < 2e-6
    assert np.abs(outs["rot_ego"].cpu().numpy().reshape(B, 3, 3) - z["rot"]).max() < 2e-6
    assert np.abs(outs["trans"].cpu().numpy() - z["trans"]).max() < 1e-5 * max(1.0, np.abs(z["trans"]).max())


def test_mask_resize_bit_exact():
    o = ops()
    m = (torch.rand(3, 1, 256, 256, generator=torch.Generator().manual_seed(60)) > 0.5).float()
    out = torch.empty(3, 1, 64, 64, device="cuda")
    o.mask_resize_nearest(m.cuda(), out)
    assert torch.equal(out.cpu(), m[..., ::4, ::4])
