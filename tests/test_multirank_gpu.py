"""GPU, two ranks: the real N > 1 step path (givepose_amd.runner.ShardRunner -- the class bench.py runs: slots in flight,
ONE communication stream per rank, all-gather of the per-crop (R, t, s)) in two fresh rank processes sharing the one
GPU of the box (gloo backend; RCCL refuses two ranks per device -- the driver's 2/4/8-GPU runs use RCCL).  Every rank's
gathered poses must equal, bitwise and in global crop order, the poses both ranks computed serially on their own."""
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
def test_two_ranks_gathered_poses_equal_serial():
    from givepose_amd.runner import rank_selfcheck
    ctx = mp.get_context("forkserver")        # server started in conftest.pytest_configure, before any GPU call
    q = ctx.Queue()
    port = _free_port()
    world, B, NF = 2, 8, 2
    ps = [ctx.Process(target=rank_selfcheck, args=(r, world, port, q, B, 6, NF)) for r in range(world)]
    [p.start() for p in ps]
    res = {}
    try:
        for _ in ps:
            rank, status, gathered, own = q.get(timeout=480)
            assert status == "ok", (rank, status)
            res[rank] = (gathered, own)
    finally:
        [p.join(60) for p in ps]
        [p.kill() for p in ps if p.is_alive()]
    for slot in range(NF):
        expect = np.concatenate([res[r][1][slot] for r in range(world)], 0)       # rank-major = global crop order
        assert expect.shape == (world * B, 15)
        for r in range(world):
            assert np.array_equal(res[r][0][slot], expect), (slot, r, float(np.abs(res[r][0][slot] - expect).max()))
    assert not np.array_equal(res[0][1][0], res[1][1][0])     # the ranks really held different crops
