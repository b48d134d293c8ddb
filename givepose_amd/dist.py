"""Data-parallel sharding of crops over the GPUs of one node + all-gather of per-crop poses.

The reference has no distributed code.  Crops are independent units except for the
DCNv3 stride-2 offset coupling, which stays inside one per-GPU batch, so the global batch is cut into
contiguous shards (rank r owns [r*per, (r+1)*per)), every rank runs the same PoseNet replica, and the only
exchange is ONE all-gather of (R 9, t 3, s 3) = 15 fp32 per crop over RCCL/xGMI (backend "nccl" on ROCm;
"gloo" in the CPU tests).  No reduction exists on this path.
"""
import os

import torch
import torch.distributed as dist

POSE_WIDTH = 15  # 9 (R row-major) + 3 (t) + 3 (s)


def init_from_env(backend=None, force=False):
    """(rank, local_rank, world) from the torchrun env; initialises the default group when world > 1 -- or, with `force`, also
    for world == 1: a ONE-rank RCCL communicator, so that a one-GPU box can exercise communicator creation, the collective and
    the barrier of the N > 1 path (tests/test_rccl_single_rank.py, GP_BENCH_FORCE_COLLECTIVE=1 in bench.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def shard_bounds(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of rank; sizes differ by at most one, earlier ranks get the extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(data: dict, rank: int, world: int):
    """Slice every (B, ...) tensor of an eval ``data`` dict to this rank's contiguous shard."""
    B = next(iter(data.values())).shape[0]
    lo, hi = shard_bounds(B, rank, world)
    return {k: v[lo:hi] for k, v in data.items()}


def pack_poses(rot, trans, size, out=None):
    """(B,3,3),(B,3),(B,3) -> (B,15) fp32 on rot's device."""
    B = rot.shape[0]
    if out is None:
        out = torch.empty(B, POSE_WIDTH, dtype=rot.dtype if rot.dtype == torch.float64 else torch.float32, device=rot.device)
    if rot.is_cuda and out.dtype == torch.float32 and all(x.dtype == torch.float32 and x.is_contiguous() for x in (rot, trans, size, out)):
        import ctypes
        from . import _lib       # one library kernel on the current stream (no PyTorch kernels beside the slots' MFMA kernels)
        _lib.check(_lib.load().gp_pack_poses(rot.data_ptr(), trans.data_ptr(), size.data_ptr(), out.data_ptr(), B,
                                            ctypes.c_void_p(torch.cuda.current_stream(rot.device).cuda_stream)), "gp_pack_poses")
        return out
    out[:, :9] = rot.reshape(B, 9)
    out[:, 9:12] = trans
    out[:, 12:15] = size
    return out


def unpack_poses(p):
    return p[:, :9].reshape(-1, 3, 3), p[:, 9:12], p[:, 12:15]


def all_gather_poses(local, world: int, out=None, group=None, force=False):
    """local (per, 15) -> (world*per, 15), rank-major = global crop order.  Equal shard sizes use the fused
    all_gather_into_tensor; ragged shards fall back to the list form.  force: run the collective for world == 1 too (a
    one-rank communicator: init_from_env(force=True))."""
    if world == 1 and not force:
        return local
    sizes = [None] * world
    if out is not None or _equal_shards(local, world, group):
        if out is None:
            out = torch.empty(world * local.shape[0], local.shape[1], dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    n = torch.tensor([local.shape[0]], device=local.device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    mx = int(max(int(v) for v in ns))
    pad = torch.zeros(mx, local.shape[1], dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: int(k)] for p, k in zip(parts, ns)], 0)


def _equal_shards(local, world, group):
    n = torch.tensor([local.shape[0], -local.shape[0]], device=local.device)
    dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    return int(n[0]) == -int(n[1])
