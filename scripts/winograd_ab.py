"""Winograd F(2x2, 3x3) against the direct LDS-window conv (gp_gemm variant 13) for the 256 -> 256 head convs of TopDownXyzHead
(network/xyz_head.py:241-316) at 64 x 64 and 32 x 32 -- the experiment north_star's "im2col / Winograd tiles" and the round-3 review asked for.

Non-fused form, which is what can be built from the library's kernels: Y = A^T [ (G g G^T) . (B^T d B) ] A as
   V[p] = (B^T d B)[p]  (16 positions p, one row per 4 x 4 input patch = 2 x 2 output pixels, C columns)      -- torch here, NOT timed as ours
   M[p] = V[p] U[p]^T, U[p] = (G g G^T)[p] (Cout x Cin)                                                          -- 16 x gp_gemm, TIMED
   Y    = A^T M A                                                                                                 -- torch here, not timed
The 16 GEMMs carry 16/36 of the direct conv's MACs.  If THEY alone are not >= 1.3 x faster than the direct conv, the form is rejected
whatever the transforms cost (each of them moves 4 x the activation bytes on top: V is 16 values per 4 output pixels).
Errors: against torch's fp32 conv of the fp32 inputs, for the direct conv and the Winograd form in fp16 and in the split-operand mode."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from givepose_amd import ops

dev = "cuda"
torch.manual_seed(0)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32, device=dev)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32, device=dev)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32, device=dev)


def timed(f, n=20, rounds=5):
    ts = []
    for _ in range(rounds):
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)


def patches(xp, R):
    """xp (B, R+2, R+2, C) zero-padded -> (B, R/2, R/2, 4, 4, C): the 4 x 4 input patch of every 2 x 2 output tile."""
    B, _, _, C = xp.shape
    return xp.unfold(1, 4, 2).unfold(2, 4, 2).permute(0, 1, 2, 4, 5, 3)       # (B, T, T, 4, 4, C)


for (Bc, R) in ((64, 64), (128, 64), (128, 32)):
    C = 256
    x32 = torch.randn(Bc, R, R, C, device=dev)
    w32 = torch.randn(C, C, 3, 3, device=dev) * (9 * C) ** -0.5
    ref = F.conv2d(x32.permute(0, 3, 1, 2), w32, padding=1).permute(0, 2, 3, 1).contiguous()        # fp32 oracle of the op
    scale = float(ref.abs().max())
    T = R // 2
    M = Bc * T * T
    flops_direct = 2.0 * Bc * R * R * C * 9 * C
    print(f"=== {Bc} crops, {R} x {R}, 256 -> 256   (direct conv: {flops_direct / 1e9:.0f} GFLOP; Winograd GEMMs: {flops_direct * 16 / 36 / 1e9:.0f} GFLOP)")
    # ---------------- fp16
    x16 = x32.half()
    wp16 = w32.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous().half()
    out16 = torch.empty(Bc, R, R, C, dtype=torch.float16, device=dev)
    f_direct = lambda: ops.conv2d_nhwc(x16, wp16, 3, 3, 1, 1, out=out16)
    t_direct = timed(f_direct)
    e_direct = float((out16.float() - ref).abs().max()) / scale
    U32 = torch.einsum("ij,ocjk,lk->iloc", G, w32, G).reshape(16, C, C).contiguous()              # (pos, Cout, Cin)
    U16 = U32.half()
    xp = F.pad(x32, (0, 0, 1, 1, 1, 1))
    d = patches(xp, R)                                                                            # (B, T, T, 4, 4, C) fp32 view
    V32 = torch.einsum("ij,btujkc,lk->ilbtuc", BT, d, BT).reshape(16, M, C).contiguous()          # transform in fp32 (a kernel would do the same adds)
    V16 = V32.half()
    for out_dt, name in ((torch.float32, "fp32 M"), (torch.float16, "fp16 M")):
        Mo = torch.empty(16, M, C, dtype=out_dt, device=dev)
        def f_wino():
            for p in range(16):
                ops.gemm(V16[p], U16[p], Mo[p], splitk=1)
        t_w = timed(f_wino, n=5)
        Y = torch.einsum("ij,jlbtuc,kl->btiukc", AT, Mo.float().reshape(4, 4, Bc, T, T, C), AT).reshape(Bc, R, R, C)
        e_w = float((Y - ref).abs().max()) / scale
        print(f"  fp16  direct window conv (v13) {t_direct:7.1f} us  err {e_direct:.2e} | Winograd 16 GEMMs ({name}) {t_w:7.1f} us  err {e_w:.2e}"
              f" | GEMMs alone {t_direct / t_w:.2f} x the direct conv's speed (needed >= 1.3 incl. transforms)")
    # what the two transform passes would cost at best: bytes moved / 6.3 TB/s (V written + read: 16/4 x the activations; M likewise)
    act = Bc * R * R * C * 2
    print(f"  transforms at the HBM roofline: input (read {act / 1e6:.0f} MB, write {4 * act / 1e6:.0f} MB) + output (read {4 * act / 1e6:.0f}-{8 * act / 1e6:.0f} MB, write {act / 1e6:.0f} MB)"
          f" >= {(10 * act) / 6.3e6:.0f} us on top")
    # ---------------- split-operand mode (fp32 activations, 3 MFMAs per product)
    if Bc == 64:
        ws = ops.split_weights(w32.permute(0, 2, 3, 1).reshape(C, 9 * C), dev)
        outs = torch.empty(Bc, R, R, C, dtype=torch.float32, device=dev)
        f_ds = lambda: ops.conv2d_nhwc(x32, ws, 3, 3, 1, 1, out=outs)
        t_ds = timed(f_ds, n=5)
        e_ds = float((outs - ref).abs().max()) / scale
        Us = [ops.split_weights(U32[p], dev) for p in range(16)]
        Ms = torch.empty(16, M, C, dtype=torch.float32, device=dev)
        def f_ws():
            for p in range(16):
                ops.gemm(V32[p], Us[p], Ms[p], splitk=1)
        t_ws = timed(f_ws, n=3)
        Ys = torch.einsum("ij,jlbtuc,kl->btiukc", AT, Ms.reshape(4, 4, Bc, T, T, C), AT).reshape(Bc, R, R, C)
        e_ws = float((Ys - ref).abs().max()) / scale
        print(f"  split direct window conv        {t_ds:7.1f} us  err {e_ds:.2e} | Winograd 16 split GEMMs (incl. their X split passes) {t_ws:7.1f} us  err {e_ws:.2e}"
              f" | {t_ds / t_ws:.2f} x")
    del x32, w32, ref, V32, V16, d, xp
    torch.cuda.empty_cache()
