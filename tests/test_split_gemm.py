"""GPU parity of the split-operand mode (gp_gemm_desc.split_shift, PoseNet(split_gemm=True)): dense contractions on the
fp16 matrix pipe with hi + 2^-11 lo' operand planes (three MFMAs per product, fp32 accumulate).

Op level: against float64 products of the SAME fp32 inputs -- tolerance 3e-6 relative to scale (an fp32 GEMM with a
different summation order passes 2e-5 in tests/test_hip_ops.py; plain fp16 operands sit at 4e-3).
Whole path: against the reference golden vectors and the CPU oracle at north_star's 1e-4 on R / t / s.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 3e-6


def ops():
    from givepose_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def rel64(got, ref):
    got, ref = got.detach().double().cpu(), ref.double()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12))


def test_split_planes_reconstruct():
    o = ops()
    x = torch.cat([rnd(64, 256, seed=1), rnd(64, 256, seed=2, scale=1e-3), rnd(64, 256, seed=3, scale=30.0)], 0)
    wide = torch.zeros(192, 320)
    wide[:, 32:288] = x
    xd = wide.cuda()
    planes = o.split_planes(xd[:, 32:], 192, 256, 320).clone()[: 2 * 192 * 256].view(2, 192, 256).cpu().double()
    back = planes[0] + planes[1] * 2.0 ** -o.SPLIT_SHIFT
    # 2^-21 relative wherever the hi plane is a normal fp16 number; below that (|x| < 6e-5) the planes bottom out at the
    # low plane's subnormal spacing: 2^-24 / 2^11 absolute
    assert bool(((back - x.double()).abs() <= 2.0 ** -21 * x.double().abs() + 2.0 ** -34).all())
    assert torch.equal(planes[0].float(), x.half().float())


@pytest.mark.parametrize("variant", [0, 4, 7, 8, 10])
def test_split_gemm_variants_and_epilogues(variant):
    o = ops()
    for (M, N, K) in [(256, 256, 128), (700, 388, 256), (1000, 512, 64), (130, 12, 512)]:
        x, w, b = rnd(M, K, seed=61), rnd(N, K, seed=62, scale=K ** -0.5), rnd(N, seed=63)
        res, gamma = rnd(M, N, seed=64), rnd(N, seed=65)
        lin = x.double() @ w.double().t() + b.double()
        sw = o.split_weights(w, "cuda")
        for epi, ref in ((o.EPI_NONE, lin), (o.EPI_GELU, F.gelu(lin)), (o.EPI_RELU, F.relu(lin)), (o.EPI_SCALE_RES, res.double() + gamma.double() * lin)):
            out = torch.zeros(M, N + 4, dtype=torch.float32, device="cuda")
            kw = dict(gamma=gamma.cuda(), residual=res.cuda()) if epi == o.EPI_SCALE_RES else {}
            o.gemm(x.cuda(), sw, out, bias=b.cuda(), epilogue=epi, variant=variant, ldc=N + 4, **kw)
            assert rel64(out[:, :N], ref) < TOL, (M, N, K, epi, rel64(out[:, :N], ref))
            assert float(out[:, N:].abs().max()) == 0.0


def test_split_gemm_small_magnitudes():
    """|x| ~ 1e-3 (the reference's std = 1e-3 head initialisation): the scaled low plane keeps 2^-22 relative."""
    o = ops()
    M, N, K = 512, 256, 512
    x, w = rnd(M, K, seed=5, scale=1e-3), rnd(N, K, seed=6, scale=1e-3)
    ref = x.double() @ w.double().t()
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    o.gemm(x.cuda(), o.split_weights(w, "cuda"), out)
    assert rel64(out, ref) < TOL


@pytest.mark.parametrize("splitk", [1, 4, 7, 16])
def test_split_gemm_splitk(splitk):
    """K ranges of the concatenated 3-segment loop: ranges that end inside the cross segments scale their own slab."""
    o = ops()
    M, N, K = 64, 256, 2048
    x, w, b = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=K ** -0.5), rnd(N, seed=6)
    ref = F.leaky_relu(x.double() @ w.double().t() + b.double(), 0.1)
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    o.gemm(x.cuda(), o.split_weights(w, "cuda"), out, bias=b.cuda(), epilogue=o.EPI_LRELU, splitk=splitk)
    assert rel64(out, ref) < TOL


def test_split_gemm_strided_x_and_auto_splitk():
    o = ops()
    B = 64
    fc1 = rnd(B, 2048, seed=9)
    w = rnd(256, 1024, seed=10, scale=1024 ** -0.5)
    ref = fc1[:, 1024:].double() @ w.double().t()
    out = torch.empty(B, 256, dtype=torch.float32, device="cuda")
    o.gemm(fc1.cuda()[:, 1024:], o.split_weights(w, "cuda"), out, M=B, K=1024, ldx=2048)
    assert rel64(out, ref) < TOL
    x, w2 = rnd(B, 8192, seed=11), rnd(2048, 8192, seed=12, scale=8192 ** -0.5)     # the PnP fc1 shape: automatic split-K
    out2 = torch.empty(B, 2048, dtype=torch.float32, device="cuda")
    o.gemm(x.cuda(), o.split_weights(w2, "cuda"), out2)
    assert rel64(out2, x.double() @ w2.double().t()) < TOL


@pytest.mark.parametrize("cfg", [dict(B=2, H=16, Cin=128, Cout=256, k=3, s=1, p=1), dict(B=3, H=16, Cin=64, Cout=128, k=3, s=2, p=1),
                                 dict(B=2, H=8, Cin=128, Cout=256, k=2, s=2, p=0), dict(B=4, H=32, Cin=256, Cout=256, k=3, s=1, p=1)])
def test_split_conv_with_groupnorm_stats(cfg):
    o = ops()
    B, H, Cin, Cout, k, s, p = (cfg[n] for n in ("B", "H", "Cin", "Cout", "k", "s", "p"))
    x = rnd(B, Cin, H, H, seed=11)
    w = rnd(Cout, Cin, k, k, seed=12, scale=(Cin * k * k) ** -0.5)
    ref = F.conv2d(x.double(), w.double(), None, stride=s, padding=p).permute(0, 2, 3, 1)
    xp = x.permute(0, 2, 3, 1).contiguous().cuda()
    sw = o.split_weights(w.permute(0, 2, 3, 1).reshape(Cout, -1), "cuda")
    Ho = ref.shape[1]
    out = torch.empty(B, Ho, Ho, Cout, dtype=torch.float32, device="cuda")
    hw = Ho * Ho
    gn = None
    if hw % 64 == 0:
        part = torch.zeros(B * (hw // 64) * 32 * 2, dtype=torch.float32, device="cuda")
        gn = (part, 32, hw)
    o.conv2d_nhwc(xp, sw, k, k, s, p, out=out, gn=gn)
    assert rel64(out, ref) < TOL
    if gn is not None:
        st = part.view(B, hw // 64, 32, 2).sum(1).cpu().double()
        r = ref.reshape(B, hw, 32, Cout // 32)
        assert float((st[..., 0] - r.sum((1, 3))).abs().max()) < 1e-3 * float(r.abs().sum((1, 3)).max())
        assert float((st[..., 1] - (r * r).sum((1, 3))).abs().max()) < 1e-5 * float((r * r).sum((1, 3)).max())


# ------------------------------------------------------------------------------------------------ whole path
def _batch(B, seed):
    from givepose_amd import synth
    return {k: torch.from_numpy(v) for k, v in synth.synth_batch(B, seed=seed).items()}


@pytest.fixture(scope="module")
def net_split():
    from givepose_amd import PoseNet, PoseNetConfig
    return PoseNet(PoseNetConfig(), dtype=torch.float32, seed=0, split_gemm=True).cuda()


@pytest.mark.parametrize("B", [1, 4, 5])
def test_split_mode_matches_reference_golden(golden, net_split, B):
    z = golden(f"posenet_e2e_B{B}")
    out = net_split(_batch(B, int(z["batch_seed"])), "cuda")
    assert np.array_equal(out["mask"].cpu().numpy(), z["out_mask"])
    err = {k: float(np.abs(out[k].cpu().numpy() - z["out_" + k]).max()) for k in ("rot", "trans", "size", "nocs_coor", "ivfc_coor")}
    print("split", B, err)
    assert err["nocs_coor"] < 2e-4 and err["ivfc_coor"] < 2e-4
    assert err["rot"] < 1e-4 and err["trans"] < 1e-4 and err["size"] < 1e-4


def test_split_mode_runs_split_kernels(net_split):
    from test_hip_posenet import _launch_labels
    labels = _launch_labels(net_split, _batch(2, 3))
    gemms = {l: n for l, n in labels.items() if l.startswith(("gemm ", "conv"))}
    assert gemms and all("split3" in l for l in gemms), gemms


@pytest.mark.parametrize("use_dcn", ["dcnv3", ""])
def test_split_mode_bs64_matches_oracle(use_dcn):
    """The bench shape, hipGraph replay: north_star's 1e-4 on R / t / s against the CPU oracle."""
    from givepose_amd import PoseNet, PoseNetConfig, synth
    from oracle import posenet_ref as O
    cfg = PoseNetConfig(use_dcn=use_dcn)
    data = _batch(64, 640)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = O.posenet_forward_ref(O.load_params(synth.synth_state_dict(cfg, 0)), data, cfg)
    net = PoseNet(cfg, dtype=torch.float32, seed=0, use_graph=True, split_gemm=True).cuda()
    for _ in range(3):
        out = net(data, "cuda")
    err = {k: float((out[k].cpu() - ref[k]).abs().max()) for k in ("rot", "trans", "size", "nocs_coor", "ivfc_coor")}
    print("split bs64", repr(use_dcn), err)
    assert torch.equal(out["mask"].cpu(), ref["mask"])
    assert err["rot"] < 1e-4 and err["trans"] < 1e-4 and err["size"] < 1e-4
    assert err["nocs_coor"] < 2e-4 and err["ivfc_coor"] < 2e-4


@pytest.mark.parametrize("variant", [0, 7, 8, 10])
def test_split_gemm_out_planes_feed_the_next_gemm(variant):
    """fc1 -> GELU -> planes -> fc2 (+ layer scale, shortcut): the hidden tensor never exists in fp32."""
    o = ops()
    M, C = 1024, 128
    x, w1, b1 = rnd(M, C, seed=1), rnd(4 * C, C, seed=2, scale=C ** -0.5), rnd(4 * C, seed=3)
    w2, b2, gamma, res = rnd(C, 4 * C, seed=4, scale=(4 * C) ** -0.5), rnd(C, seed=5), rnd(C, seed=6), rnd(M, C, seed=7)
    hid = F.gelu(x.double() @ w1.double().t() + b1.double())
    ref = res.double() + gamma.double() * (hid @ w2.double().t() + b2.double())
    h = torch.empty(M, 4 * C, dtype=torch.float32, device="cuda")
    o.gemm(x.cuda(), o.split_weights(w1, "cuda"), h, bias=b1.cuda(), epilogue=o.EPI_GELU, variant=variant, out_planes=True)
    planes = h.view(torch.float16).reshape(2, M, 4 * C).cpu().double()
    assert rel64(planes[0] + planes[1] * 2.0 ** -o.SPLIT_SHIFT, hid) < TOL
    out = res.clone().cuda()
    o.gemm(h, o.split_weights(w2, "cuda"), out, bias=b2.cuda(), epilogue=o.EPI_SCALE_RES, gamma=gamma.cuda(), residual=out,
           variant=variant, x_planes=True)
    assert rel64(out, ref) < TOL


@pytest.mark.parametrize("B,H,Cin", [(2, 64, 256), (3, 32, 256), (5, 16, 256), (1, 64, 128)])
def test_split_window_conv(B, H, Cin):
    """The 3x3 LDS-window kernel (variant 13) in the split-operand mode: three passes over the (chunk, tap) loop, fp32 output,
    fused GroupNorm statistics -- against float64 and against the tap-by-tap ping-pong kernel (variant 10)."""
    o = ops()
    Cout = 256
    x = rnd(B, Cin, H, H, seed=21)
    w = rnd(Cout, Cin, 3, 3, seed=22, scale=(Cin * 9) ** -0.5)
    ref = F.conv2d(x.double(), w.double(), None, padding=1).permute(0, 2, 3, 1)
    xp = x.permute(0, 2, 3, 1).contiguous().cuda()
    sw = o.split_weights(w.permute(0, 2, 3, 1).reshape(Cout, -1), "cuda")
    hw = H * H
    outs = {}
    for v in (13, 10):
        out = torch.empty(B, H, H, Cout, dtype=torch.float32, device="cuda")
        part = torch.zeros(B * (hw // 64) * 32 * 2, dtype=torch.float32, device="cuda")
        o.conv2d_nhwc(xp, sw, 3, 3, 1, 1, out=out, gn=(part, 32, hw), variant=v, epilogue=o.EPI_GELU if v == 13 and Cin == 128 else o.EPI_NONE)
        if not (v == 13 and Cin == 128):
            assert rel64(out, ref) < TOL, v
            st = part.view(B, hw // 64, 32, 2).sum(1).cpu().double()
            r = ref.reshape(B, hw, 32, Cout // 32)
            assert float((st[..., 1] - (r * r).sum((1, 3))).abs().max()) < 1e-5 * float((r * r).sum((1, 3)).max())
        else:
            assert rel64(out, F.gelu(ref)) < TOL
        outs[v] = out


def test_producers_write_planes_like_gp_split_planes():
    """dwconv_ln / layernorm / groupnorm_apply / upsample with out_planes: bit for bit the planes gp_split_planes makes of
    their fp32 output (the split-operand GEMM that follows reads them with x_planes=True, no split pass in between)."""
    o = ops()
    dev = "cuda"
    B, H, C = 2, 16, 256

    def planes_of(t32):
        return o.split_planes(t32.reshape(-1, t32.shape[-1]), t32.numel() // t32.shape[-1], t32.shape[-1], t32.shape[-1]).clone()[: 2 * t32.numel()]

    x = rnd(B, H, H, C, seed=1).to(dev)
    # dw7x7 + LN
    wt, bias, lw, lb = rnd(49, C, seed=2).to(dev), rnd(C, seed=3).to(dev), rnd(C, seed=4).to(dev), rnd(C, seed=5).to(dev)
    ref = o.dwconv_ln(x, wt, bias, lw, lb, torch.empty_like(x), 7)
    got = o.dwconv_ln(x, wt, bias, lw, lb, torch.empty_like(x), 7, out_planes=True)
    assert torch.equal(got.view(torch.float16).reshape(-1), planes_of(ref))
    # row LayerNorm
    ref = o.layernorm(x, lw, lb, torch.empty_like(x))
    got = o.layernorm(x, lw, lb, torch.empty_like(x), out_planes=True)
    assert torch.equal(got.view(torch.float16).reshape(-1), planes_of(ref))
    # GroupNorm + GELU
    part = torch.zeros(B * 64 * 32 * 2, device=dev)
    xv = x.view(B, H * H, C)
    ref = o.groupnorm(xv, lw, lb, torch.empty_like(xv), 32, o.ACT_GELU, part)
    got = o.groupnorm(xv, lw, lb, torch.empty_like(xv), 32, o.ACT_GELU, part, out_planes=True)
    assert torch.equal(got.view(torch.float16).reshape(-1), planes_of(ref))
    with pytest.raises(RuntimeError):
        o.groupnorm(xv, lw, lb, xv, 32, o.ACT_GELU, part, out_planes=True)        # in place is impossible: the planes overlap unread input
    # bilinear x2
    ref = o.upsample_bilinear2x(x, torch.empty(B, 2 * H, 2 * H, C, device=dev))
    got = o.upsample_bilinear2x(x, torch.empty(B, 2 * H, 2 * H, C, device=dev), out_planes=True)
    assert torch.equal(got.view(torch.float16).reshape(-1), planes_of(ref))
