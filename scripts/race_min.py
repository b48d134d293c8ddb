"""Minimal two-stream reproducer for the batches-in-flight corruption.

Stream A loops an AGGRESSOR launch, stream B loops a VICTIM launch into rotating output buffers; after every round the
victim outputs are compared bitwise with a reference computed alone.  Nothing is shared between the two streams.

  AGG = v1 | v1dbgNN | v1small | v7 | v7big | v4 | v8 | v10 | v13 | v16 | mlp | gnxyz | stem | torchmm | none
  VIC = k3 | gnapply | gn | deconv | torchfma            (the scalar-FMA victim of r02_race6.log was a build switch of the investigation;
                                                 scripts/repro/pkfma_beside_mfma.hip -DNOPK is its stand-alone form)
The v1* aggressors are the round-1 register-staged `gemm_kernel`, which no longer exists in the library: run them
against a build of the round-1 sources (git show 35dfc42:givepose_amd/csrc/<file> for the six .hip files + common.hpp,
hipcc -O3 --offload-arch=gfx950 -fPIC -shared) with GP_LIB_PATH=<that .so>; scripts/repro/pkfma_beside_mfma.hip is the
self-contained form.  Results of round 2: profiles/r02_race3...6.log.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from givepose_amd import ops
from givepose_amd._lib import EPI_LRELU

AGG = os.environ.get("AGG", "v1"); VIC = os.environ.get("VIC", "k3")
ROUNDS = int(os.environ.get("ROUNDS", 150)); NV = int(os.environ.get("NV", 6)); NA = int(os.environ.get("NA", 12))
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g)

# aggressor operands: PnP fc1 (M=64, N=2048, K=8192) fp16
ax, aw, ab = rnd(64, 8192).half().to(dev), (rnd(2048, 8192) * 0.01).half().to(dev), rnd(2048).to(dev)
aout = torch.empty(64, 2048, dtype=torch.float16, device=dev)
bx, bw = rnd(4096, 1024).half().to(dev), (rnd(256, 1024) * 0.03).half().to(dev)
bout = torch.empty(4096, 256, dtype=torch.float16, device=dev)


if AGG in ("v10", "v8", "v7big", "v4"):
    cx, cw, cb = rnd(4096, 1024).half().to(dev), (rnd(1024, 1024) * 0.03).half().to(dev), rnd(1024).to(dev)
    cout = torch.empty(4096, 1024, dtype=torch.float16, device=dev)
if AGG == "v16":     # weights-in-registers GEMM (K = 512): stage-2 fc1 shape
    ex, ew, eb = rnd(16384, 512).half().to(dev), (rnd(2048, 512) * 0.04).half().to(dev), rnd(2048).to(dev)
    eout = torch.empty(16384, 2048, dtype=torch.float16, device=dev)
if AGG == "gnxyz":   # GroupNorm + GELU + out layer of the xyz heads (MFMA form unless GP_GNXYZ_MFMA=0)
    from givepose_amd._lib import ACT_GELU
    qx = rnd(64, 4096, 256).half().to(dev)
    qxf = qx.float().view(64, 64, 64, 32, 8)
    qpart = torch.stack([qxf.sum((2, 4)), (qxf * qxf).sum((2, 4))], -1).contiguous().view(-1)
    qgw, qgb, qow, qob = rnd(256).to(dev), rnd(256).to(dev), (rnd(3, 256) * 0.06).to(dev), rnd(3).to(dev)
    qn, q4 = torch.empty(64, 3, 64, 64, device=dev), torch.empty(64 * 4096, 4, device=dev)
if AGG == "stem":
    simg, sw_, sb_ = rnd(64, 3, 256, 256).to(dev), (rnd(48, 128) * 0.14).to(dev), rnd(128).to(dev)
    slw, slb, sout = rnd(128).to(dev), rnd(128).to(dev), torch.empty(64, 64, 64, 128, dtype=torch.float16, device=dev)
if AGG == "v13":
    dx, dw = rnd(8, 32, 32, 256).half().to(dev), (rnd(256, 2304) * 0.02).half().to(dev)
    dout = torch.empty(8, 32, 32, 256, dtype=torch.float16, device=dev)
if AGG == "mlp":
    mx, mres = rnd(16384, 128).half().to(dev), rnd(16384, 128).half().to(dev)
    mw1, mb1 = (rnd(512, 128) * 0.05).half().to(dev), rnd(512).to(dev)
    mw2p, mb2, mg = ops.convnext_mlp_pack_w2((rnd(128, 512) * 0.05).half().to(dev)), rnd(128).to(dev), rnd(128).to(dev)
    mout = torch.empty_like(mx)


def aggressor():
    if AGG == "v1":
        ops.gemm(ax, aw, aout, bias=ab, epilogue=EPI_LRELU, variant=1, splitk=1)
    elif AGG == "v7":
        ops.gemm(ax, aw, aout, bias=ab, epilogue=EPI_LRELU, variant=7, splitk=1)
    elif AGG == "v1small":
        ops.gemm(bx, bw, bout, variant=1, splitk=1)
    elif AGG.startswith("v1dbg"):
        ops.gemm(ax, aw, aout, bias=ab, epilogue=EPI_LRELU, variant=1 + 100 * int(AGG[5:]), splitk=1)
    elif AGG in ("v10", "v8", "v7big", "v4"):      # large-tile kernels on a trunk-like shape
        ops.gemm(cx, cw, cout, bias=cb, epilogue=EPI_LRELU, variant={"v10": 10, "v8": 8, "v7big": 7, "v4": 4}[AGG], splitk=1)
    elif AGG == "v16":
        ops.gemm(ex, ew, eout, bias=eb, epilogue=1, variant=16, splitk=1)
    elif AGG == "gnxyz":
        ops.groupnorm_apply_xyz(qx, qgw, qgb, qow, qob, qn, q4, 32, ACT_GELU, qpart)
    elif AGG == "stem":
        ops.convnext_stem(simg, sw_, sb_, slw, slb, sout)
    elif AGG == "v13":
        ops.conv2d_nhwc(dx, dw, 3, 3, 1, 1, out=dout, variant=13)
    elif AGG == "mlp":
        ops.convnext_mlp(mx, mw1, mb1, mw2p, mb2, mg, mres, mout)
    elif AGG == "torchmm":
        torch.mm(ax, aw.t(), out=aout)


# victim operands
R = 262144
xyz4 = rnd(R, 4).to(dev)
kw_, kb_ = rnd(256, 3).to(dev), rnd(256).to(dev)
gx = rnd(64, 1024, 256).half().to(dev)
outs = [torch.empty(R, 256, dtype=torch.float16, device=dev) for _ in range(NV)]
gouts = [torch.empty(64 * ops.groupnorm_chunks(64, 1024) * 32 * 2, dtype=torch.float32, device=dev) for _ in range(NV)]


gnw, gnb = rnd(256).to(dev), rnd(256).to(dev)
gpart = torch.empty(64 * 64 * 32 * 2, dtype=torch.float32, device=dev)
gao = [torch.empty_like(gx) for _ in range(NV)] if VIC == "gnapply" else None


if VIC == "deconv":    # the xyz heads' deconv-as-GEMM (128x128 tile, fp32 output: generic epilogue)
    vx, vw = rnd(4096, 1024).half().to(dev), (rnd(2304, 1024) * 0.03).half().to(dev)
    vouts = [torch.empty(4096, 2304, device=dev) for _ in range(NV)]


def victim(i):
    if VIC == "deconv":
        ops.gemm(vx, vw, vouts[i], variant=7, splitk=1)
        return vouts[i]
    if VIC == "k3":
        ops.pointwise_k3(xyz4, kw_, kb_, outs[i])
        return outs[i]
    if VIC == "torchfma":
        torch.addcmul(xyz4[:, 1:2], xyz4[:, 0:1], kw_[:, 0].view(1, -1), out=None)   # warm
        o = (xyz4[:, 0:1] * kw_[:, 0].view(1, -1) + xyz4[:, 1:2] * kw_[:, 1].view(1, -1) + kb_.view(1, -1)).half()
        outs[i].copy_(o)
        return outs[i]
    if VIC == "gnapply":      # GroupNorm statistics + apply + GELU (packed-fp32 polynomial) into a rotating output
        return ops.groupnorm(gx, gnw, gnb, gao[i], 32, 1, gpart)
    if VIC == "gn":
        from givepose_amd import _lib
        import ctypes
        _lib.check(_lib.load().gp_groupnorm_stats(ctypes.c_void_p(gx.data_ptr()), ctypes.c_void_p(gouts[i].data_ptr()), 64, 1024, 256, 32,
                                                  ops.dtype_code(gx.dtype), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "stats")
        return gouts[i]


ref = victim(0).clone()
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
bits = lambda t: t.view(torch.int16 if t.dtype == torch.float16 else torch.int32)
bad = 0
aref, abad = None, 0
for r in range(ROUNDS):
    with torch.cuda.stream(sa):
        for _ in range(NA):
            aggressor()
    with torch.cuda.stream(sb):
        res = [victim(i) for i in range(NV)]
    torch.cuda.synchronize()
    if AGG.startswith("v1") and AGG != "v1small":      # is the aggressor's own result stable?
        if aref is None:
            aref = aout.clone()
        elif not torch.equal(bits(aout), bits(aref)):
            abad += 1
    for i, o in enumerate(res):
        if not torch.equal(bits(o), bits(ref)):
            bad += 1
            if bad <= 4:
                d = torch.nonzero(bits(o).reshape(-1) != bits(ref).reshape(-1)).flatten()
                C = o.shape[-1] if o.dim() > 1 else 64
                print(f"round {r} launch {i}: {d.numel()} elems differ; rows {sorted(set((d // C).tolist()))[:8]} cols {sorted(set((d % C).tolist()))[:20]}")
print(f"AGG={AGG} VIC={VIC} NOPK={os.environ.get('GP_K3_NOPK', '0')}: {bad} corrupted victim launches of {ROUNDS * NV}; aggressor output changed in {abad} rounds")
