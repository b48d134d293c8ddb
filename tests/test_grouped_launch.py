"""GPU parity of GROUPED launches (PoseNet(dcn_couple=64): G batches of 64 crops in one launch sequence -- the launch shape
bench.py times) and of the BASELINE configs in their literal wording at bs = 64.

What is checked, and against what:
  * a grouped launch against SEPARATE forwards of the same batches: numerically equivalent, not bitwise (split-K factors, tile
    choices and the GroupNorm chunking follow the row count: other summation orders) -- max over ALL crops of |dR|, |dt|, |ds|
    and of both coordinate maps bounded at the level of two equivalent builds of the mode;
  * every batch of a grouped launch (group 0 AND group 1: an indexing slip in the per-group DCNv3 slices would only show in
    group >= 1) against the CPU oracle of that batch alone: 1e-4 on R / t / s in the fp32 and split-operand modes, the fp16
    distribution bounds of tests/test_hip_posenet.py::test_fp16_bs64_close_to_oracle otherwise;
  * at 2 x 64 crops (the benched launch shape) the same per 64-crop batch, with a second batch seed (641) beside the seed
    every other bs-64 test uses (640), so that a badly conditioned crop shows here and not on the driver's run;
  * BASELINE configs[1] literally: ResNet-34 trunk, DCNv3 off, bs = 64 (and with DCNv3: configs[2] on that trunk).
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from givepose_amd.rot_cond import rot_error_bound  # noqa: E402

pytestmark = pytest.mark.gpu

MODES = {"f32": dict(dtype=torch.float32), "split": dict(dtype=torch.float32, split_gemm=True), "f16": dict(dtype=torch.float16)}
KEYS = ("rot", "trans", "size", "nocs_coor", "ivfc_coor")


def _batch(B, seed):
    from givepose_amd import synth
    return {k: torch.from_numpy(v) for k, v in synth.synth_batch(B, seed=seed).items()}


def _cat(batches):
    return {k: torch.cat([b[k] for b in batches], 0) for k in batches[0]}


def _oracle(cfg, data, f64=False, inter=False):
    from givepose_amd import synth
    from oracle import posenet_ref as O
    torch.set_num_threads(min(16, torch.get_num_threads()))
    P = O.load_params(synth.synth_state_dict(cfg, 0))
    if f64:
        P = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
        data = {k: (v.double() if v.is_floating_point() else v) for k, v in data.items()}
    with torch.no_grad():
        return O.posenet_forward_ref(P, data, cfg, return_intermediates=inter)


def _labels(net, data):
    """Launch labels of one eager forward with the row count stripped: which schedule ran where."""
    import ctypes
    import re
    from givepose_amd import _lib
    lib = _lib.load()
    net.forward_device(data)
    torch.cuda.synchronize()
    _lib.check(lib.gp_timing_begin(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gp_timing_begin")
    net.forward_device(data)
    _lib.check(lib.gp_timing_end(), "gp_timing_end")
    out = set()
    for r in range(500):
        lab = ctypes.create_string_buffer(160)
        c, n, ms, fl, by = ctypes.c_int(), ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        if lib.gp_timing_top(r, lab, 160, ctypes.byref(c), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)) != 0:
            break
        s = lab.value.decode()
        if s.startswith(("gemm", "conv")):
            out.add(re.sub(r" M\d+", "", s))
    return out


def _errs(out, ref, sl=slice(None)):
    return {k: float((out[k][sl].float().cpu() - ref[k].float()).abs().max()) for k in KEYS}


@pytest.mark.parametrize("mode", ["f32", "split", "f16"])
def test_grouped_launch_equals_separate_batches(mode):
    """2 x 8 crops as one launch sequence (dcn_couple = 8) against two forwards of 8 crops, and each group against the oracle of
    its batch.  8 crops per group: crops 4..7 of a group read the offset rows of crop 1 OF THEIR GROUP, so a
    group that read another group's prefix, or group 0's, fails the oracle comparison of group 1."""
    from givepose_amd import PoseNet, PoseNetConfig
    cfg = PoseNetConfig()
    b0, b1 = _batch(8, 71), _batch(8, 72)
    both = _cat([b0, b1])
    grouped = PoseNet(cfg, seed=0, dcn_couple=8, **MODES[mode]).cuda()
    alone = PoseNet(cfg, seed=0, **MODES[mode]).cuda()
    keep = KEYS + ("rot6d", "rot_allo")
    og = {k: v.clone() for k, v in grouped.forward_device(both).items() if k in keep}
    oa = [{k: v.clone() for k, v in alone.forward_device(b).items() if k in keep} for b in (b0, b1)]
    same_schedules = _labels(grouped, both) == _labels(alone, b0)
    for g in range(2):
        sl = slice(8 * g, 8 * g + 8)
        d = {k: float((og[k][sl].float() - oa[g][k].float()).abs().max()) for k in KEYS}
        print(f"grouped vs alone [{mode}] group {g}: same schedules {same_schedules}", d)
        # Not required bitwise: even where every GEMM runs the same schedule (the labels say so), split-K factors and the GroupNorm
        # chunking depend on the row count, i.e. another summation order (measured: 2e-5 on R in the fp32 mode).  Bounded like
        # two numerically equivalent builds of the mode -- over ALL crops, R, t, s and both coordinate maps.
        bit = all(torch.equal(og[k][sl], oa[g][k]) for k in KEYS)
        print(f"   bitwise: {bit}")
        if mode == "f16":
            # two numerically equivalent schedules of the fp16 mode: the rot6d logits (what the network computes), t, s and the coordinate maps
            # tightly; every crop's allocentric |dR| within what its own logit difference and conditioning explain (givepose_amd/rot_cond.py);
            # the maximum of the egocentric |dR| over the crops is the worst-conditioned crop's and gets no ceiling of its own
            lg = float((og["rot6d"][sl].float() - oa[g]["rot6d"].float()).abs().max() / oa[g]["rot6d"].float().abs().max())
            bnd = rot_error_bound(oa[g]["rot6d"].float().cpu(), og["rot6d"][sl].float().cpu(), max_logit_err=1.5e-2 * float(oa[g]["rot6d"].float().abs().max()))
            per_u = (og["rot_allo"][sl].float().cpu().reshape(8, -1) - oa[g]["rot_allo"].float().cpu().reshape(8, -1)).abs().max(1).values.double()
            print(f"   rot6d logits rel {lg:.2e}; allocentric |dR| / bound max {float(torch.nan_to_num(per_u / bnd, nan=1e9, posinf=1e9).max()):.3f}")
            assert lg < 1.5e-2 and bool((per_u <= bnd).all()), (g, lg, per_u, bnd)
            # t / s: ONE oracle tolerance, as bench.py's self-check (two schedules that are each within 3e-2 of the oracle; measured between them: s up to 2.1e-2 on the
            # ResNet-34 variant, 1.6e-2 here once the 16-crop launch took the 16 x 8-tile depth-wise kernel and the 8-crop launch the 16 x 4 one -- round 6)
            assert d["trans"] < 3e-2 and d["size"] < 3e-2 and d["nocs_coor"] < 2e-2 and d["ivfc_coor"] < 2e-2, (g, d)
        else:
            assert all(v < 5e-5 for v in d.values()), (g, d)
        ref = _oracle(cfg, (b0, b1)[g])
        e = _errs(og, ref, sl)
        print(f"grouped vs oracle [{mode}] group {g}", e)
        if mode == "f16":
            assert e["rot"] < 3e-2 and e["trans"] < 3e-2 and e["size"] < 3e-2 and e["nocs_coor"] < 2e-2 and e["ivfc_coor"] < 2e-2, (g, e)
        else:
            assert e["rot"] < 1e-4 and e["trans"] < 1e-4 and e["size"] < 1e-4 and e["nocs_coor"] < 2e-4 and e["ivfc_coor"] < 2e-4, (g, e)
    # the coupling really is per group: group 1 alone is NOT what the 16 crops give as one coupled batch of 16
    coupled16 = {k: v.clone() for k, v in alone.forward_device(both).items() if k in ("rot",)}
    assert not torch.equal(coupled16["rot"][8:], og["rot"][8:])


@pytest.fixture(scope="module")
def oracle_2x64():
    """fp32 (and float64) oracle poses of the two 64-crop batches a 128-crop grouped launch holds: seeds 640 and 641."""
    from givepose_amd.config import PoseNetConfig
    cache = {}

    def get():
        if not cache:
            cfg = PoseNetConfig()
            bs = [_batch(64, 640), _batch(64, 641)]
            cache["b"] = bs
            cache["ref"] = [_oracle(cfg, b, inter=True) for b in bs]
            cache["ref64"] = [_oracle(cfg, b, f64=True) for b in bs]
        return cache
    return get


@pytest.mark.parametrize("mode", ["f16", "split"])
def test_grouped_launch_bs128_matches_oracle_per_batch(oracle_2x64, mode):
    """The launch shape `value` is timed on -- PoseNet(dcn_couple=64) over 2 x 64 crops, hipGraph replay -- against the oracle PER
    64-crop batch: both groups, two batch seeds.  split-operand mode: north_star's 1e-4 on R / t / s; fp16: the distribution
    bounds of the single-batch test."""
    from givepose_amd import PoseNet, PoseNetConfig
    c = oracle_2x64()
    net = PoseNet(PoseNetConfig(), seed=0, use_graph=True, dcn_couple=64, **MODES[mode]).cuda()
    data = _cat(c["b"])
    for _ in range(3):
        out = net(data, "cuda")
    dev = net.forward_device(data)
    assert torch.equal(out["mask"].cpu(), torch.cat([r["mask"] for r in c["ref"]], 0))
    for g in range(2):
        sl = slice(64 * g, 64 * g + 64)
        ref, ref64 = c["ref"][g], c["ref64"][g]
        e = _errs(out, ref, sl)
        e64 = _errs(out, ref64, sl)
        noise = {k: float((ref[k].double() - ref64[k]).abs().max()) for k in ("rot", "trans", "size")}
        per = (out["rot"][sl].cpu() - ref["rot"]).abs().reshape(64, -1).max(1).values.sort().values
        print(f"bs128 grouped [{mode}] batch {g} (seed {640 + g}): vs fp32 oracle {e}; vs float64 oracle rot {e64['rot']:.2e} trans {e64['trans']:.2e} "
              f"size {e64['size']:.2e}; fp32 oracle vs its float64 self {noise}; per-crop |dR| median {float(per[32]):.2e} p90 {float(per[57]):.2e}")
        if mode == "split":
            # against the float64 oracle the mode has the whole 1e-4 to itself; against the fp32 CPU oracle (what north_star names)
            # the bar is shared with that oracle's own rounding on its worst-conditioned crop (measured beside it above)
            assert e64["rot"] < 1e-4 and e64["trans"] < 1e-4 and e64["size"] < 1e-4, (g, e64)
            assert e["rot"] < 1e-4 + noise["rot"] and e["trans"] < 1e-4 and e["size"] < 1e-4, (g, e, noise)
            assert float(per[32]) < 2e-5 and float(per[57]) < 5e-5, (g, float(per[32]), float(per[57]))
            assert e["nocs_coor"] < 2e-4 and e["ivfc_coor"] < 2e-4, (g, e)
        else:
            # The distribution of |dR| over the crops, the rot6d logits (what the network computes, before the 6-D -> R
            # normalisation) relative to their scale, and EVERY crop's |dR| against what its own logit error and conditioning
            # explain (givepose_amd/rot_cond.py).  The plain maximum of |dR| is the worst-conditioned crop of the batch and moves with
            # the batch seed and with every rounding-level change of any kernel (seed 641: 4.3e-2 -> 9.4e-2 when the bilinear
            # blend began to round once instead of twice): it gets no ceiling of its own.
            r6d = dev["rot6d"][sl].float().cpu()
            r6 = float((r6d - ref["rot6d"]).abs().max() / ref["rot6d"].abs().max())
            bound = rot_error_bound(ref["rot6d"], r6d, max_logit_err=1.5e-2 * float(ref["rot6d"].abs().max()))   # inf only where the reference's logits alone excuse the crop
            assert int(torch.isinf(bound).sum()) <= 2, (g, int(torch.isinf(bound).sum()))
            per_u = (dev["rot_allo"][sl].float().cpu().reshape(64, -1) - ref["rot_allo"].reshape(64, -1)).abs().max(1).values.double()   # allocentric: the map the bound is for
            worst = int(per_u.argmax())
            print(f"   rot6d logits rel {r6:.2e}; worst crop {worst}: |dR| {float(per_u[worst]):.3e}, explained up to {float(bound[worst]):.3e}")
            assert r6 < 1.5e-2, (g, r6)
            assert float(per[32]) < 8e-3 and float(per[57]) < 2e-2 and float(per[62]) < 5e-2, (g, e)
            assert bool((per_u <= bound).all()), (g, (per_u / bound).max())
            assert e["size"] < 3e-2 and e["trans"] < 3e-2 * max(1.0, float(ref["trans"].abs().max())), (g, e)
            assert e["nocs_coor"] < 2e-2 and e["ivfc_coor"] < 2e-2, (g, e)


@pytest.mark.parametrize("mode", ["f32", "split"])
def test_single_batch_bs64_second_seed_meets_1e_4(oracle_2x64, mode):
    """The 1e-4 assert of the parity modes at bs = 64 on a second batch (seed 641; tests/test_hip_posenet.py and
    tests/test_split_gemm.py use 640): R / t / s against the fp32 CPU oracle and against its float64 run."""
    from givepose_amd import PoseNet, PoseNetConfig
    c = oracle_2x64()
    net = PoseNet(PoseNetConfig(), seed=0, **MODES[mode]).cuda()
    out = net(c["b"][1], "cuda")
    e, e64 = _errs(out, c["ref"][1]), _errs(out, c["ref64"][1])
    noise = float((c["ref"][1]["rot"].double() - c["ref64"][1]["rot"]).abs().max())
    print(f"bs64 seed 641 [{mode}] vs fp32 oracle {e} vs float64 {e64} (fp32 oracle vs float64: rot {noise:.2e})")
    assert e64["rot"] < 1e-4 and e64["trans"] < 1e-4 and e64["size"] < 1e-4, e64
    assert e["rot"] < 1e-4 + noise and e["trans"] < 1e-4 and e["size"] < 1e-4, (e, noise)


@pytest.mark.parametrize("seed", [642, 645])
def test_split_mode_bs64_more_seeds_meet_1e_4_against_float64(seed):
    """Round-4 review: the split-operand mode meets north_star's 1e-4 against the fp32 CPU oracle with 7 % of margin on the bench batch, and
    the asserts above compare with `1e-4 + (fp32 oracle - float64 oracle)` because that oracle's own rounding error on its worst-conditioned
    crop is of the size of the bar.  Here: two MORE batch seeds at bs = 64 (four until round 6: 643 and 644 dropped for the suite's time budget, ~37 s of float64
    oracle each) against the oracle run in FLOAT64, the plain < 1e-4 on R / t / s
    with no noise term -- so that the margin against the fp32 oracle cannot hide a seed at 1.2e-4."""
    from givepose_amd import PoseNet, PoseNetConfig
    cfg = PoseNetConfig()
    b = _batch(64, seed)
    ref64 = _oracle(cfg, b, f64=True)
    net = PoseNet(cfg, seed=0, **MODES["split"]).cuda()
    out = net(b, "cuda")
    e64 = {k: float((out[k].double().cpu() - ref64[k].double()).abs().max()) for k in ("rot", "trans", "size")}
    per = (out["rot"].double().cpu() - ref64["rot"].double()).abs().reshape(64, -1).max(1).values.sort().values
    print(f"split bs64 seed {seed} vs float64 oracle: {e64}; per-crop |dR| median {float(per[32]):.2e} p90 {float(per[57]):.2e}")
    assert e64["rot"] < 1e-4 and e64["trans"] < 1e-4 and e64["size"] < 1e-4, (seed, e64)


@pytest.mark.parametrize("use_dcn", ["", "dcnv3"])
@pytest.mark.parametrize("mode", ["f32", "f16"])
def test_resnet34_bs64_matches_oracle(use_dcn, mode):
    """BASELINE configs[1] in its literal wording -- ResNet-34 trunk, DCNv3 disabled (plain 3x3 convs), bs = 64 -- and configs[2]
    on the same trunk (use_dcn = 'dcnv3'); hipGraph replay as bench.py --workload resnet34[_nodcn] runs it."""
    from givepose_amd import PoseNet, PoseNetConfig
    cfg = PoseNetConfig(main_backbone="resnet34", use_dcn=use_dcn)
    data = _batch(64, 3464)
    ref = _oracle(cfg, data)
    net = PoseNet(cfg, seed=0, use_graph=True, **MODES[mode]).cuda()
    for _ in range(3):
        out = net(data, "cuda")
    e = _errs(out, ref)
    per = (out["rot"].cpu() - ref["rot"]).abs().reshape(64, -1).max(1).values.sort().values
    print(f"resnet34 use_dcn={use_dcn!r} bs64 [{mode}]", e, "per-crop |dR| median %.2e p90 %.2e" % (float(per[32]), float(per[57])))
    assert torch.equal(out["mask"].cpu(), ref["mask"])
    if mode == "f32":
        assert e["rot"] < 1e-4 and e["trans"] < 1e-4 and e["size"] < 1e-4 and e["nocs_coor"] < 2e-4 and e["ivfc_coor"] < 2e-4, e
    else:
        assert float(per[32]) < 8e-3 and float(per[57]) < 2.5e-2 and e["rot"] < 8e-2, e
        assert e["size"] < 3e-2 and e["trans"] < 3e-2 * max(1.0, float(ref["trans"].abs().max())) and e["nocs_coor"] < 2e-2 and e["ivfc_coor"] < 2e-2, e
