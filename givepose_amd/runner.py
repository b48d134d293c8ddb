"""One rank of the data-parallel inference path: `inflight` independent batches rotate over PoseNet slots and, when
world > 1, every step ends with the all-gather of the per-crop (R, t, s) -- issued on ONE communication stream per
rank whatever slot produced the poses, so that all ranks enqueue their collectives in the same order on one stream of
one communicator (bench.py and tests/test_multirank_gpu.py run exactly this class).

The reference has no distributed code; the shard layout is givepose_amd/dist.py.
"""
import torch

from . import dist as gd


class ShardRunner:
    """h2d: None = the inputs stay resident in HBM (what bench.py's `value` measures); "crops" = every step first brings its
    fp32 crops from pinned host memory; "frames" = uint8 frames + masks + boxes travel and `gp_crop_rois` makes the crops on the
    device (givepose_amd.preprocess).  Either way the transfer of a step runs on a COPY stream into a per-slot staging
    buffer -- it depends only on the staging buffer's previous consumer, not on the slot's previous forward -- and the
    slot's stream turns it into the model's static inputs (one device copy / the crop kernel) right in front of its
    hipGraph, so the PCIe traffic of step n + 1 overlaps the kernels of step n."""

    def __init__(self, net, batch, device, world=1, inflight=None, h2d=None, frame_hw=(480, 640), force_collective=False):
        """force_collective: run the pack + all-gather of the N > 1 step for world == 1 as well (a one-rank RCCL communicator
        made by dist.init_from_env(force=True)): the only way a one-GPU box can execute the collective path on RCCL itself."""
        self.net, self.B, self.dev, self.world = net, batch, torch.device(device), world
        self.collective = world > 1 or bool(force_collective)
        self.NF = max(1, net.inflight if inflight is None else inflight)
        if self.NF > 1 and not net.use_graph:
            raise ValueError("batches in flight need the hipGraph path (per-slot streams)")
        if h2d not in (None, "crops", "frames"):
            raise ValueError("h2d: None | 'crops' | 'frames'")
        self.h2d = h2d
        self.statics = [net.static_inputs(batch, self.dev, slot=i) for i in range(self.NF)]
        self.poses = [torch.empty(batch, gd.POSE_WIDTH, device=self.dev) for _ in range(self.NF)]
        self.gathered = [torch.empty(world * batch, gd.POSE_WIDTH, device=self.dev) for _ in range(self.NF)] if self.collective else None
        self.comm = torch.cuda.Stream(device=self.dev) if self.collective else None
        self.packed = [None] * self.NF      # event: the previous poses of slot i have been packed (its outputs may be overwritten)
        self.count = 0
        self.last = None
        if h2d:
            self.copy = torch.cuda.Stream(device=self.dev)
            self.pinned = [None] * self.NF      # slot -> {name: pinned host tensor}
            self.staging = [None] * self.NF     # slot -> {name: device tensor the copy stream fills}
            self.staged_free = [None] * self.NF  # event: the slot stream has consumed the staging buffers
            self.host_bytes = 0
            if h2d == "frames":
                from .preprocess import RoiCropper
                self.cropper = RoiCropper(frame_hw[0], frame_hw[1], self.dev)
                self.boxes = [None] * self.NF

    def load(self, slot, host_batch):
        """Fill slot `slot`'s device-resident inputs from a dict of numpy arrays / tensors (and, with h2d = "crops", keep a
        pinned host copy that every step transfers again)."""
        for k, v in host_batch.items():
            t = torch.as_tensor(v)
            self.statics[slot][k].copy_(t.reshape(self.statics[slot][k].shape))
        if self.h2d == "crops":
            # pinned in the static input's own dtype: the per-step transfer is a plain copy into a buffer of that dtype
            self.pinned[slot] = {k: torch.as_tensor(v).reshape(self.statics[slot][k].shape).to(self.statics[slot][k].dtype).contiguous().pin_memory()
                                 for k, v in host_batch.items()}
            self.staging[slot] = {k: torch.empty_like(self.statics[slot][k]) for k in host_batch}
            self.host_bytes = sum(v.numel() * v.element_size() for v in self.pinned[slot].values())

    def load_frames(self, slot, frames_u8, masks_u8, frame_idx, mask_idx, boxes, scalars):
        """h2d = "frames": the uint8 frames (F,H,W,3) / masks (B,H,W) and detection boxes of slot `slot`; `scalars` = the inputs
        the crop kernel does not produce (cam_K, mean_size), kept resident."""
        assert self.h2d == "frames"
        for k, v in scalars.items():
            self.statics[slot][k].copy_(torch.as_tensor(v).reshape(self.statics[slot][k].shape))
        self.pinned[slot] = {"frames": torch.as_tensor(frames_u8).contiguous().pin_memory(), "masks": torch.as_tensor(masks_u8).contiguous().pin_memory()}
        self.staging[slot] = {k: torch.empty_like(v, device=self.dev) for k, v in self.pinned[slot].items()}
        self.boxes[slot] = (list(frame_idx), list(mask_idx), boxes)
        self.host_bytes = sum(v.numel() * v.element_size() for v in self.pinned[slot].values()) + len(frame_idx) * 120

    def _bring_inputs(self, i, cur):
        """Copy stream: host -> staging (waits only for the staging buffer's last consumer); slot stream: staging -> static inputs."""
        # The staging buffers of slot i are free once the slot stream has consumed them (three steps ago).  The HOST waits for
        # that event -- it also keeps the host from running more than NF steps ahead -- rather than the copy stream: a
        # device-side wait of the copy queue on a slot queue cost 8 % of the step rate (scripts/h2d_ablate.py: 9 420 against
        # 10 300 images/s), a host wait costs nothing because the slot's next launches are queued long before they can run.
        if self.staged_free[i] is not None:
            self.staged_free[i].synchronize()
        with torch.cuda.stream(self.copy):
            for k, v in self.pinned[i].items():
                self.staging[i][k].copy_(v, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.copy)
        s = self.net.ensure_stream(i, self.dev) if self.net.use_graph else cur
        s.wait_event(ready)
        with torch.cuda.stream(s):
            if self.h2d == "frames":
                fi, mi, boxes = self.boxes[i]
                self.cropper(self.staging[i]["frames"], self.staging[i]["masks"], fi, mi, boxes, out=self.statics[i])
            else:
                for k, v in self.staging[i].items():
                    self.statics[i][k].copy_(v, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(s)
            self.staged_free[i] = ev

    def step(self):
        """One pass of the whole path over one batch; consecutive steps use consecutive slots and overlap on the device
        (nothing is skipped: every step replays the full launch sequence on its own buffers)."""
        i = self.count % self.NF
        if self.h2d and (self.pinned[i] is None or self.staging[i] is None):
            raise RuntimeError(f"ShardRunner(h2d={self.h2d!r}): slot {i} has no host inputs -- call load() / load_frames() for every one of "
                               f"the {self.NF} slots before step()")
        self.count += 1
        cur = torch.cuda.current_stream(self.dev)
        if self.packed[i] is not None:
            cur.wait_event(self.packed[i])          # forward_device orders the slot stream after `cur`
        if self.h2d:
            self._bring_inputs(i, cur)
        out = self.net.forward_device(self.statics[i], self.dev, slot=i, wait=self.NF == 1)
        if self.collective:
            done = torch.cuda.Event()
            done.record(self.net.stream(i) if self.net.use_graph else cur)
            self.comm.wait_event(done)
            with torch.cuda.stream(self.comm):
                gd.pack_poses(out["rot"], out["trans"], out["size"], out=self.poses[i])
                ev = torch.cuda.Event()
                ev.record(self.comm)
                self.packed[i] = ev
                gd.all_gather_poses(self.poses[i], self.world, out=self.gathered[i], force=True)
        self.last = (i, out)
        return out

    def result(self, slot=None):
        """(world*B, 15) gathered poses of `slot` (default: the last step's), valid after a device synchronise."""
        i = self.last[0] if slot is None else slot
        if self.collective:
            return self.gathered[i]
        o = self.net._plan(self.B, self.dev, i)["buf"]
        return gd.pack_poses(o["rot_ego"].view(self.B, 3, 3), o["trans"], o["size"])


def rank_selfcheck(rank, world, port, queue, batch=8, steps=6, inflight=2, backend="gloo", device_index=0):
    """Entry point of ONE rank process of the N > 1 self-check (tests/test_multirank_gpu.py starts `world` of them from a
    fork server that never touched the GPU): runs `steps` steps of the real step path (ShardRunner: slots in flight, one
    comm stream, all-gather of the poses) and reports, per slot, the gathered (world*B, 15) poses of its last use next
    to this rank's own poses computed strictly serially (eager, slot 0) on a separate PoseNet.  backend "gloo" lets every rank share
    one GPU (RCCL refuses two ranks per device); the driver's real runs use "nccl" = RCCL with one rank per GPU.
    world == 1 with backend "nccl": the same step path on a ONE-rank RCCL communicator (communicator creation,
    all_gather_into_tensor on the comm stream, barrier) -- what a one-GPU box can execute of RCCL (tests/test_rccl_single_rank.py)."""
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(device_index if backend == "nccl" else rank))
    # this pool's host driver only supports dmabuf IPC: without it RCCL's buffer exchange fails with `hipIpcGetMemHandle: invalid
    # argument` (the image exports the variable; kept for environments built by hand)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        import torch.distributed as dist
        from . import PoseNet, PoseNetConfig, synth
        gd.init_from_env(backend=backend, force=world == 1)
        torch.cuda.set_device(device_index)
        dev = torch.device("cuda", device_index)
        barrier = (lambda: dist.barrier(device_ids=[device_index])) if backend == "nccl" else dist.barrier
        cfg = PoseNetConfig()
        net = PoseNet(cfg, dtype=torch.float16, seed=0, use_graph=True, inflight=inflight).to(dev)
        run = ShardRunner(net, batch, dev, world, force_collective=world == 1)
        batches = [synth.synth_batch(batch, seed=500 + 10 * rank + i) for i in range(run.NF)]
        for i, b in enumerate(batches):
            run.load(i, b)
        for _ in range(steps):
            run.step()
        torch.cuda.synchronize(dev)
        barrier()
        gathered = [run.result(i).cpu().numpy() for i in range(run.NF)]
        serial = PoseNet(cfg, dtype=torch.float16, seed=0, use_graph=False, inflight=inflight).to(dev)   # same tile choices, run strictly serially
        own = []
        for b in batches:
            o = serial.forward_device({k: torch.from_numpy(v) for k, v in b.items()}, dev)
            own.append(gd.pack_poses(o["rot"], o["trans"], o["size"]).cpu().numpy())
        queue.put((rank, "ok", gathered, own))
        barrier()
        dist.destroy_process_group()
    except Exception as e:   # the parent must never wait for a dead rank
        import traceback
        queue.put((rank, "error: " + repr(e) + "\n" + traceback.format_exc(), None, None))


def run_cli(argv, env_updates, queue, timeout=900):
    """Run a command line in a child of THIS process and report (returncode, stdout, stderr tail) through `queue`.  For tests
    that must start a program from a process that never touched the GPU (a GPU-initialised process must not fork + exec on
    this pool): the caller starts this function through the multiprocessing fork server."""
    import os
    import subprocess
    env = dict(os.environ)
    for k, v in env_updates.items():
        if v is None:
            env.pop(k, None)
        else:
            env[k] = v
    try:
        r = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=timeout)
        queue.put((r.returncode, r.stdout, r.stderr[-4000:]))
    except Exception as e:
        queue.put((-1, "", repr(e)))
