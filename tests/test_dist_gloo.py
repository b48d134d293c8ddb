"""CPU, world_size 2 and 8 over gloo: shard bounds + all-gather of per-crop poses reproduce global crop order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from givepose_amd import dist as gd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, _, w = gd.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(0)
    full = {"rot": torch.randn(n_items, 3, 3, generator=g), "trans": torch.randn(n_items, 3, generator=g),
            "size": torch.randn(n_items, 3, generator=g)}
    mine = gd.shard_batch(full, r, w)
    local = gd.pack_poses(mine["rot"], mine["trans"], mine["size"])
    allp = gd.all_gather_poses(local, w)
    R, t, s = gd.unpack_poses(allp)
    ok = torch.equal(R, full["rot"]) and torch.equal(t, full["trans"]) and torch.equal(s, full["size"])
    q.put((rank, ok, tuple(allp.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 7])   # equal shards -> fused gather; ragged -> padded list gather
def test_all_gather_poses_world2(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in ps]
    [p.join(60) for p in ps]
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (n_items, 15) for _, _, shape in res)


def test_all_gather_poses_world8_configs4_rank_count():
    """BASELINE configs[4] is 8 ranks x 64 crops: the exact rank count of the driver's scaling run, control flow only (gloo on the CPU -- eight
    processes on one GPU would exceed the box's process guard, and the builder's boxes have one GPU).  512 crops sharded 8 ways, every rank's
    gathered poses in global crop order; plus a ragged total (509) through the padded gather."""
    ctx = mp.get_context("spawn")
    for n_items in (512, 509):
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_worker, args=(r, 8, port, n_items, q)) for r in range(8)]
        [p.start() for p in ps]
        res = [q.get(timeout=240) for _ in ps]
        [p.join(60) for p in ps]
        assert sorted(r for r, _, _ in res) == list(range(8))
        assert all(ok for _, ok, _ in res), res
        assert all(shape == (n_items, 15) for _, _, shape in res)


def test_shard_bounds_cover_and_order():
    for n in (0, 1, 5, 64, 513):
        for w in (1, 2, 3, 8):
            b = [gd.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_checkpoint_key_report(tmp_path):
    """load_checkpoint(report_path=...): the full rename / unknown / missing report is written BEFORE a strict load fails, so that the
    first user with the released weights can send back one file that settles the timm key layout."""
    import json
    import torch
    from givepose_amd import PoseNet
    from givepose_amd.checkpoint import load_checkpoint
    net = PoseNet()
    sd = net.state_dict()
    some = list(sd)[:6]
    ck = {("module." + k): sd[k] for k in some}                       # a DataParallel checkpoint: renamed
    ck["backbone.not_a_layer.weight"] = torch.zeros(3, 5)             # ... with a key that matches nothing
    rep = tmp_path / "keys.json"
    import pytest
    with pytest.raises(KeyError, match="match no tensor"):
        load_checkpoint(net, ck, verbose=False, report_path=str(rep))
    doc = json.load(open(rep))
    assert len(doc["renamed"]) == 6 and doc["renamed"][0]["checkpoint"].startswith("module.")
    assert doc["unknown"] == [{"checkpoint": "backbone.not_a_layer.weight", "shape": [3, 5]}]
    assert len(doc["missing"]) == len(sd) - 6 and doc["shape_mismatch"] == []
    out = load_checkpoint(net, {("module." + k): sd[k] for k in some}, verbose=False, report_path=str(rep))
    assert len(out["renamed"]) == 6 and json.load(open(rep))["unknown"] == []


def test_expected_checkpoint_key_list_is_current(tmp_path):
    """tests/golden/expected_checkpoint_keys.txt (what a user with the released weights diffs their checkpoint against,
    `python -m givepose_amd.checkpoint <ckpt>`) is exactly what PoseNet registers; the diff reports renames / unknown / missing keys."""
    import torch
    from givepose_amd import checkpoint as C
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = C._read_expected(os.path.join(root, C.EXPECTED_KEYS_FILE))
    assert exp == C.expected_keys()
    sd = {k: torch.zeros(s) for k, s in exp[:40]}
    sd["module.backbone.stem.0.weight"] = sd.pop("backbone.stem_0.weight")      # DataParallel prefix + un-flattened timm name
    sd["not.a.key"] = torch.zeros(2)
    d = C.diff_keys(sd, exp)
    assert d["renamed"] == {"module.backbone.stem.0.weight": "backbone.stem_0.weight"} and [k for k, _ in d["unknown"]] == ["not.a.key"]
    assert len(d["missing"]) == len(exp) - 40 and not d["shape_mismatch"]
    f = tmp_path / "ck.pth"
    torch.save({"state_dict": sd}, f)
    assert C.main([str(f)]) == 1
