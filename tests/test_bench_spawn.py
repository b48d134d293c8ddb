"""`python bench.py --gpus 2` with no torchrun environment starts its own rank processes (before anything touches the GPU) and
fails when a rank fails.  GPU: the two-rank control flow on the one GPU of the box (GP_BENCH_REHEARSE=1: gloo, both ranks on
cuda:0 -- the numbers mean nothing, the driver's real runs use RCCL with one rank per GPU).  CPU: the launcher alone."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spawn_ranks_propagates_failure(tmp_path, monkeypatch):
    """The launcher part alone (no GPU): children that exit non-zero make the parent exit non-zero and stop the others."""
    sys.path.insert(0, ROOT)
    import bench
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\nr = int(os.environ['RANK'])\nassert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "time.sleep(30 if r == 2 else 0.2)\nsys.exit(7 if r == 1 else 0)\n")
    monkeypatch.setattr(bench, "__file__", str(script))
    monkeypatch.setattr(bench.os.path, "abspath", lambda p: str(script) if p == str(script) else os.path.normpath(p))
    assert bench.spawn_ranks(3, []) == 7          # returns long before rank 2's 30 s: it was terminated


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("ranks,extra", [(2, ["--no-roofline"]), (4, ["--no-serial"])])
def test_bench_ranks_without_torchrun(ranks, extra):
    """2 ranks (with the serial legs), and 4 ranks WITH the roofline leg (which every rank of an N > 1 run executes, so that none waits in the final barrier
    while rank 0 measures): the 4- / 8-rank control flow of the driver's scaling run, rehearsed on gloo with every rank on cuda:0
    (at most 6 processes may use the card: 4 ranks + this one)."""
    import multiprocessing as mp
    from givepose_amd.runner import run_cli
    ctx = mp.get_context("forkserver")        # started in conftest.pytest_configure before any GPU call: bench.py is exec'ed from a clean process
    q = ctx.Queue()
    argv = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "6", "--warmup", "2", "--batch", "8", "--inflight", "2",
            "--no-cpu-baseline", "--no-parity"] + extra
    p = ctx.Process(target=run_cli, args=(argv, {"GP_BENCH_REHEARSE": "1", "WORLD_SIZE": None, "RANK": None, "LOCAL_RANK": None}, q, 700))
    p.start()
    rc, out, err = q.get(timeout=800)
    p.join(30)
    assert rc == 0, err
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == ranks and line["config"]["global_batch"] == 8 * ranks and line["value"] > 0
    assert line["n_ranks_seen"] == ranks and line["timed_regions"] == 5 and line["value_min"] <= line["value"] <= line["value_max"]
    oc = line["overlap_check"]
    assert oc["slots"] == 2 and oc["ranks"] == ranks and oc["batches_per_launch"] == 2 and oc["poses_bitwise_equal_to_serial_replay"] is True
    assert line["config"]["batches_in_flight"] == 4
    assert list(line)[-1] == "summary" and line["summary"]["value"] == line["value"] and line["summary"]["n_ranks_seen"] == ranks
    if "--no-serial" not in extra:      # (the serial legs are the 2-rank case's; the 4-rank case runs the roofline leg on every rank instead: 100 -> ~50 s)
        assert line["summary"]["one_batch_in_flight_bs64_serial"] > 0 and line["summary"]["one_launch_in_flight"] > 0
    if "--no-roofline" not in extra:
        assert line["roofline"]["frac"] > 0 and "not the N-rank step" in line["roofline"]["measured_on"]


def test_bench_helpers_on_cpu():
    """err_stats (the measured vs_reference object) and the subprocess-free commit id."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    ref = torch.zeros(10, 15)
    got = ref.clone()
    got[3, 4] = 2e-4            # one crop's R off by 2e-4
    got[:, 9] = 5e-6            # every t by 5e-6
    e = bench.err_stats(got, ref)
    assert abs(e["rot"] - 2e-4) < 1e-9 and e["rot_median_over_crops"] == 0.0 and abs(e["trans"] - 5e-6) < 1e-9 and e["size"] == 0.0
    assert e["meets_1e-4"] is False and bench.err_stats(ref + 5e-5, ref)["meets_1e-4"] is True
    c = bench.git_head()
    assert c is None or (len(c) == 7 and all(ch in "0123456789abcdef" for ch in c))
