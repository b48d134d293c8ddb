"""GPU parity of RAGGED multi-frame launches: PoseNet.forward_device(data, groups=[b0, b1, ...]) runs the detections of several frames --
each frame one `forward` of the reference (evaluation/evaluate.py:89-114: B = the detections of ONE frame) -- in ONE launch sequence; the
one thing that couples the crops of a forward, the DCNv3 stride-2 offset / mask prefix, stays per frame through a device
table (gp_dwconv_ln_groups).  Every frame of the set is compared with the CPU oracle run ON THAT FRAME ALONE."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from givepose_amd.rot_cond import rot_error_bound  # noqa: E402

pytestmark = pytest.mark.gpu

MODES = {"f32": dict(dtype=torch.float32), "split": dict(dtype=torch.float32, split_gemm=True), "f16": dict(dtype=torch.float16)}
KEYS = ("rot", "trans", "size", "nocs_coor", "ivfc_coor")
SIZES = (1, 3, 4, 7, 2, 5)          # 22 crops -> bucket 24: two one-crop padding batches ride along


def _batch(B, seed):
    from givepose_amd import synth
    return {k: torch.from_numpy(v) for k, v in synth.synth_batch(B, seed=seed).items()}


def _cat(batches):
    return {k: torch.cat([b[k] for b in batches], 0) for k in batches[0]}


@pytest.fixture(scope="module")
def frames_and_oracle():
    from givepose_amd import synth
    from givepose_amd.config import PoseNetConfig
    from oracle import posenet_ref as O
    cfg = PoseNetConfig()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    frames = [_batch(b, 900 + i) for i, b in enumerate(SIZES)]
    P = O.load_params(synth.synth_state_dict(cfg, 0))
    with torch.no_grad():
        refs = [O.posenet_forward_ref(P, f, cfg, return_intermediates=True) for f in frames]
    return cfg, frames, refs


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("mode", ["f32", "split", "f16"])
def test_ragged_frames_match_the_oracle_of_each_frame(frames_and_oracle, mode, graph):
    from givepose_amd import PoseNet
    cfg, frames, refs = frames_and_oracle
    net = PoseNet(cfg, seed=0, use_graph=graph, **MODES[mode]).cuda()
    data = _cat(frames)
    keep = KEYS + ("rot6d", "rot_allo", "mask")
    for rep in range(3 if graph else 1):          # graph: eager warm-up, capture, replay
        out = {k: v.clone() for k, v in net.forward_device(data, groups=SIZES).items() if k in keep}
    torch.cuda.synchronize()
    assert out["rot"].shape[0] == sum(SIZES)
    i = 0
    for f, (b, ref) in enumerate(zip(SIZES, refs)):
        sl = slice(i, i + b)
        i += b
        e = {k: float((out[k][sl].float().cpu() - ref[k].float()).abs().max()) for k in KEYS}
        print(f"ragged [{mode}{' graph' if graph else ''}] frame {f} ({b} crops) vs the oracle of that frame", e)
        assert torch.equal(out["mask"][sl].cpu(), ref["mask"])
        if mode == "f16":
            # the fp16 mode's bounds (tests/test_hip_posenet.py): logits, t, s and the maps tightly, every crop's allocentric |dR| within what its
            # own logit error and conditioning explain
            r6 = ref["rot6d"].float()
            lg = float((out["rot6d"][sl].float().cpu() - r6).abs().max() / r6.abs().max())
            bnd = rot_error_bound(r6, out["rot6d"][sl].float().cpu(), max_logit_err=1.5e-2 * float(r6.abs().max()))
            per_u = (out["rot_allo"][sl].float().cpu().reshape(b, -1) - ref["rot_allo"].float().reshape(b, -1)).abs().max(1).values.double()
            assert lg < 1.5e-2 and bool((per_u <= bnd).all()), (f, lg, per_u, bnd)
            assert e["trans"] < 3e-2 and e["size"] < 3e-2 and e["nocs_coor"] < 2e-2 and e["ivfc_coor"] < 2e-2, (f, e)
        else:
            assert e["rot"] < 1e-4 and e["trans"] < 1e-4 and e["size"] < 1e-4 and e["nocs_coor"] < 2e-4 and e["ivfc_coor"] < 2e-4, (f, e)
    # the coupling really is per frame: the same 22 crops as ONE coupled batch give other poses from the second frame on
    plain = net.forward_device(data)["rot"].clone()
    assert torch.equal(plain[:1], out["rot"][:1]) or mode == "f16" or float((plain[:1] - out["rot"][:1]).abs().max()) < 5e-5
    assert not torch.equal(plain[4:], out["rot"][4:])


def test_ragged_equals_single_frame_forwards_and_launch_count_is_flat():
    """(i) a ragged launch against separate forwards of its frames on the same library (numerically equivalent, not bitwise: tile choices and
    GroupNorm chunking follow the row count); (ii) the launch count of a ragged forward does not depend on how many frames it holds."""
    import ctypes
    from givepose_amd import PoseNet, PoseNetConfig, _lib
    cfg = PoseNetConfig()
    frames = [_batch(b, 950 + i) for i, b in enumerate(SIZES)]
    data = _cat(frames)
    net = PoseNet(cfg, seed=0, dtype=torch.float32).cuda()
    og = {k: v.clone() for k, v in net.forward_device(data, groups=SIZES).items() if k in KEYS}
    i = 0
    for f, fr in zip(SIZES, frames):
        oa = net.forward_device(fr)
        d = {k: float((og[k][i:i + f].float() - oa[k].float()).abs().max()) for k in KEYS}
        assert all(v < 5e-5 for v in d.values()), (f, d)
        i += f
    lib = _lib.load()

    def launches(groups):
        net.forward_device(data, groups=groups)
        torch.cuda.synchronize()
        _lib.check(lib.gp_timing_begin(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gp_timing_begin")
        net.forward_device(data, groups=groups)
        _lib.check(lib.gp_timing_end(), "gp_timing_end")
        tot = 0
        for r in range(2000):
            lab = ctypes.create_string_buffer(160)
            c, n, ms, fl, by = ctypes.c_int(), ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            if lib.gp_timing_top(r, lab, 160, ctypes.byref(c), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)) != 0:
                break
            tot += n.value
        return tot
    n6, n2, n22 = launches(SIZES), launches((10, 12)), launches((1,) * 22)
    print("launches per ragged forward of 22 crops: 6 frames", n6, "2 frames", n2, "22 frames", n22)
    assert n6 == n2 == n22


def test_two_ragged_launch_sequences_in_flight_equal_one_at_a_time():
    """Two ragged frame sets on two slots of one PoseNet, launched back to back with wait=False (the `frames_grouped` leg's
    `crops_per_s_two_launch_sequences_in_flight`: each slot owns its buffers, its group table, its hipGraph and its stream): the poses of
    both are bitwise what the same sets give one at a time (same net, a device synchronise between the launches), replay after replay."""
    from givepose_amd import PoseNet, PoseNetConfig
    cfg = PoseNetConfig()
    sets = [((2, 5, 1, 4), 930), ((3, 3, 6), 940)]             # 12 crops each -> bucket 16 (different group tables, same plan shape)
    data = [_cat([_batch(b, seed + i) for i, b in enumerate(sizes)]) for sizes, seed in sets]
    data = [{k: v.cuda() for k, v in d.items()} for d in data]
    two = PoseNet(cfg, seed=0, use_graph=True, inflight=2, dtype=torch.float16).cuda()
    alone = []                                                # one at a time on the SAME net (a net built for overlap picks other GEMM schedules than inflight = 1: fp16 roundings)
    for s_, ((sizes, _), d) in enumerate(zip(sets, data)):
        for _ in range(3):
            o = two.forward_device(d, slot=s_, wait=True, groups=sizes)
            torch.cuda.synchronize()
        alone.append({k: o[k].clone() for k in ("rot", "trans", "size")})
    for rep in range(5):                                      # eager, capture, then replays with both sequences in flight
        outs = [two.forward_device(d, slot=s, wait=False, groups=sizes) for s, ((sizes, _), d) in enumerate(zip(sets, data))]
        torch.cuda.synchronize()
        for s in range(2):
            for k in ("rot", "trans", "size"):
                assert torch.equal(outs[s][k], alone[s][k]), (rep, s, k)


def test_plan_cache_is_bounded():
    """Plans (buffers + hipGraph per crop count) are evicted least-recently-used beyond PoseNet.max_plans; an evicted size still works."""
    from givepose_amd import PoseNet, PoseNetConfig
    net = PoseNet(PoseNetConfig(), seed=0, dtype=torch.float16, use_graph=True).cuda()
    net.max_plans = 3
    first = None
    for B in (1, 2, 3, 4, 5, 1):
        d = _batch(B, 77)
        for _ in range(3):
            r = net.forward_device(d)["rot"].clone()
        if B == 1:
            if first is None:
                first = r
            else:
                assert torch.equal(first, r)       # rebuilt after eviction: same result
        assert len(net._plans) <= 3
    assert [k[0] for k in net._plans] == [4, 5, 1]


def test_groups_argument_checks_and_uncoupled_configs():
    """groups must be positive sizes summing to the crop count; where nothing couples the crops of a forward (use_dcn = '' / the attention encoder) a grouped
    forward IS the plain forward: same plan, same bits."""
    from givepose_amd import PoseNet, PoseNetConfig
    data = _batch(6, 77)
    net = PoseNet(PoseNetConfig(), seed=0, dtype=torch.float16).cuda()
    for bad in ([], [3, 2], [6, 0], [7, -1], [2, 2, 3]):
        with pytest.raises(ValueError):
            net.forward_device(data, groups=bad)
    for kw in (dict(use_dcn=""), dict(nocsmap_encoder="att")):
        n2 = PoseNet(PoseNetConfig(**kw), seed=0, dtype=torch.float16).cuda()
        a = n2.forward_device(data, groups=[1, 2, 3])["rot"].clone()
        b = n2.forward_device(data)["rot"].clone()
        assert torch.equal(a, b) and all(not k[2] for k in n2._plans), kw
    # a ragged total beyond the largest bucket pads to a multiple of 64
    assert PoseNet.ragged_bucket(129) == 192 and PoseNet.ragged_bucket(8) == 8 and PoseNet.ragged_bucket(9) == 16
