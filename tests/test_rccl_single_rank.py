"""GPU, RCCL itself: the N > 1 step path of givepose_amd.runner.ShardRunner on a ONE-rank "nccl" (= RCCL) process group.

A one-GPU box cannot run two RCCL ranks (RCCL refuses two ranks per device; tests/test_multirank_gpu.py therefore uses gloo),
but it can make the driver's 2/4/8-GPU run not be the first RCCL call this code ever makes: communicator creation with a bound
device, `all_gather_into_tensor` of the packed poses on the rank's ONE communication stream (event-ordered behind the slots'
hipGraphs), `dist.barrier(device_ids=...)`, teardown.  The rank is a fresh process forked from the fork server that never
touched the GPU.  The multi-GPU behaviour of RCCL over xGMI stays unmeasured here."""
import multiprocessing as mp
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_one_rank_rccl_gather_path_equals_serial():
    from givepose_amd.runner import rank_selfcheck
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    B, NF = 8, 2
    p = ctx.Process(target=rank_selfcheck, args=(0, 1, _free_port(), q, B, 6, NF, "nccl", 0))
    p.start()
    try:
        rank, status, gathered, own = q.get(timeout=480)
    finally:
        p.join(60)
        if p.is_alive():
            p.kill()
    assert status == "ok", status
    for slot in range(NF):
        assert gathered[slot].shape == (B, 15)
        assert np.array_equal(gathered[slot], own[slot]), (slot, float(np.abs(gathered[slot] - own[slot]).max()))
    assert not np.array_equal(own[0], own[1])


@pytest.mark.timeout(900)
def test_bench_single_rank_on_rccl():
    """`GP_BENCH_FORCE_COLLECTIVE=1 python bench.py --gpus 1`: bench.py's own N > 1 control flow (process group, ShardRunner with the
    gather inside the step, barrier fences, all-reduced overlap verdict) on a one-rank RCCL communicator."""
    import json
    import os
    import sys
    from givepose_amd.runner import run_cli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    argv = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "4", "--batch", "8",
            "--no-cpu-baseline", "--no-parity", "--no-roofline", "--no-h2d"]
    p = ctx.Process(target=run_cli, args=(argv, {"GP_BENCH_FORCE_COLLECTIVE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                                                 "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"}, q, 800))
    p.start()
    try:
        rc, out, err = q.get(timeout=850)
    finally:
        p.join(30)
        if p.is_alive():
            p.kill()
    assert rc == 0, err
    line = json.loads(out.strip().splitlines()[-1])
    assert line["value"] and line["n_gpus"] == 1
    assert line["config"]["collective"].startswith("all_gather") and "RCCL" in line["config"]["collective"]
    assert line["overlap_check"]["poses_bitwise_equal_to_serial_replay"] is True
