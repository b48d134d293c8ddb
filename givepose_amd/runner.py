"""One rank of the data-parallel inference path: `inflight` independent batches rotate over PoseNet slots and, when
world > 1, every step ends with the all-gather of the per-crop (R, t, s) -- issued on ONE communication stream per
rank whatever slot produced the poses, so that all ranks enqueue their collectives in the same order on one stream of
one communicator (bench.py and tests/test_multirank_gpu.py run exactly this class).

The reference has no distributed code; the shard layout is givepose_amd/dist.py.
"""
import torch

from . import dist as gd


class ShardRunner:
    def __init__(self, net, batch, device, world=1, inflight=None):
        self.net, self.B, self.dev, self.world = net, batch, torch.device(device), world
        self.NF = max(1, net.inflight if inflight is None else inflight)
        if self.NF > 1 and not net.use_graph:
            raise ValueError("batches in flight need the hipGraph path (per-slot streams)")
        self.statics = [net.static_inputs(batch, self.dev, slot=i) for i in range(self.NF)]
        self.poses = [torch.empty(batch, gd.POSE_WIDTH, device=self.dev) for _ in range(self.NF)]
        self.gathered = [torch.empty(world * batch, gd.POSE_WIDTH, device=self.dev) for _ in range(self.NF)] if world > 1 else None
        self.comm = torch.cuda.Stream(device=self.dev) if world > 1 else None
        self.packed = [None] * self.NF      # event: the previous poses of slot i have been packed (its outputs may be overwritten)
        self.count = 0
        self.last = None

    def load(self, slot, host_batch):
        """Fill slot `slot`'s device-resident inputs from a dict of numpy arrays / tensors."""
        for k, v in host_batch.items():
            t = torch.as_tensor(v)
            self.statics[slot][k].copy_(t.reshape(self.statics[slot][k].shape))

    def step(self):
        """One pass of the whole path over one batch; consecutive steps use consecutive slots and overlap on the device
        (nothing is skipped: every step replays the full launch sequence on its own buffers)."""
        i = self.count % self.NF
        self.count += 1
        cur = torch.cuda.current_stream(self.dev)
        if self.packed[i] is not None:
            cur.wait_event(self.packed[i])          # forward_device orders the slot stream after `cur`
        out = self.net.forward_device(self.statics[i], self.dev, slot=i, wait=self.NF == 1)
        if self.world > 1:
            done = torch.cuda.Event()
            done.record(self.net.stream(i) if self.net.use_graph else cur)
            self.comm.wait_event(done)
            with torch.cuda.stream(self.comm):
                gd.pack_poses(out["rot"], out["trans"], out["size"], out=self.poses[i])
                ev = torch.cuda.Event()
                ev.record(self.comm)
                self.packed[i] = ev
                gd.all_gather_poses(self.poses[i], self.world, out=self.gathered[i])
        self.last = (i, out)
        return out

    def result(self, slot=None):
        """(world*B, 15) gathered poses of `slot` (default: the last step's), valid after a device synchronise."""
        i = self.last[0] if slot is None else slot
        if self.world > 1:
            return self.gathered[i]
        o = self.net._plan(self.B, self.dev, i)["buf"]
        return gd.pack_poses(o["rot_ego"].view(self.B, 3, 3), o["trans"], o["size"])
