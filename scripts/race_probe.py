"""Batches-in-flight determinism probe: run NS slots overlapped, compare every plan buffer bitwise with
its own serial result and, for every failing (rep, slot), list the differing buffers IN LAUNCH ORDER with the shape of
the difference (rows / images / channel ranges), so that the first wrong producer can be read off.

env: MODE = "" | nosplit (round-1 library via GP_LIB_PATH: also v1nosplit) ; B, NS, REPS ; EVENTS = failing events to describe (default 6)
"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from givepose_amd import PoseNet, PoseNetConfig, ops, synth

B = int(os.environ.get("B", 64)); NS = int(os.environ.get("NS", 3)); REPS = int(os.environ.get("REPS", 40))
EVENTS = int(os.environ.get("EVENTS", 6))
mode = os.environ.get("MODE", "")
dev = torch.device("cuda")
_orig = ops.auto_splitk
if mode == "v1nosplit":
    _gemm = ops.gemm
    def gemm(x, w, out, *a, **k):
        M = k.get("M") or x.shape[0]
        if k.get("conv") is None and k.get("gn") is None and _orig(M, w.shape[0], w.shape[1], 2) > 1:
            k["variant"] = 1
        return _gemm(x, w, out, *a, **k)
    ops.gemm = gemm
net = PoseNet(PoseNetConfig(), dtype=torch.float16, seed=0, use_graph=True, inflight=NS).cuda()
if mode == "nosplit":
    ops.AUTO_SPLITK = False
d = [{k: torch.from_numpy(v).cuda() for k, v in synth.synth_batch(B, seed=21 + i).items()} for i in range(NS)]
# buffers in the order their LAST writer runs (approximately the launch order)
ORDER = ["x0", "t0", "x1", "t1", "x2", "t2", "h2", "x3", "t3", "h3", "size", "cols", "nocs_nchw", "nocs_nhwc4", "e_proj0", "e_x10",
         "e_om0", "e_g0", "e_o0", "e_proj1", "e_x11", "e_om1", "e_g1", "e_o1", "e_proj2", "e_x12", "e_om2", "e_g2", "e_o2",
         "feat_cat", "ya16", "yb16", "ya32", "yb32", "ya64", "yb64", "ivfc_nchw", "ivfc_nhwc4", "p0", "p1", "p2", "fc1", "hh", "hz",
         "rot6d", "pred_t", "rot_ego", "trans"]
ref = []
for i in range(NS):
    for _ in range(3):
        net.forward_device(d[i], slot=i)
    torch.cuda.synchronize()
    ref.append({k: net._plan(B, dev, i)["buf"][k].clone() for k in ORDER})
cnt = collections.Counter()
events = 0
for rep in range(REPS):
    for i in range(NS):
        net.forward_device(d[i], slot=i, wait=False)
    torch.cuda.synchronize()
    for i in range(NS):
        buf = net._plan(B, dev, i)["buf"]
        bits = lambda t: t.view(torch.int16 if t.dtype == torch.float16 else torch.int32)
        bad = [k for k in ORDER if not torch.equal(bits(buf[k]), bits(ref[i][k]))]
        for k in bad:
            cnt[k] += 1
        if bad and events < EVENTS:
            events += 1
            print(f"--- rep {rep} slot {i}: differing (launch order): {bad}")
            for k in bad[:4]:
                t, r = buf[k], ref[i][k]
                a2, r2 = t.float().reshape(t.shape[0], -1), r.float().reshape(t.shape[0], -1)
                dm = (a2 != r2)
                rows = torch.nonzero(dm.any(1)).flatten()
                C = t.shape[-1]
                cols = torch.nonzero((t.float().reshape(-1, C) != r.float().reshape(-1, C)).any(0)).flatten()
                prow = torch.nonzero((t.float().reshape(-1, C) != r.float().reshape(-1, C)).any(1)).flatten()
                print(f"   {k} {tuple(t.shape)}: {int(dm.sum())} elems; dim0 idx {rows.tolist()[:12]}; "
                      f"rows(-1,C) {int(prow.min())}..{int(prow.max())} n={prow.numel()}; cols {int(cols.min())}..{int(cols.max())} n={cols.numel()}; "
                      f"max|d| {float((a2 - r2).abs().max()):.5f}")
                if k == bad[0]:
                    fl = torch.nonzero(t.reshape(-1).float() != r.reshape(-1).float()).flatten()[:32]
                    print("      flat idx / got / ref:", [(int(i), float(t.reshape(-1)[i]), float(r.reshape(-1)[i])) for i in fl])
print(f"MODE={mode!r} reps={REPS} slots={NS}: buffers that ever differed:", sorted(cnt.items(), key=lambda kv: ORDER.index(kv[0])))
