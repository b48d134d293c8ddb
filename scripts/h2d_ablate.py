"""Where does the pipelined H->D path lose its 12 %?  ShardRunner(h2d="crops") with parts of _bring_inputs removed."""
import sys, time, torch
sys.path.insert(0, ".")
from givepose_amd import PoseNet, PoseNetConfig, synth
from givepose_amd.runner import ShardRunner

dev = torch.device("cuda", 0)
B, NF = 64, 3
net = PoseNet(PoseNetConfig(), dtype=torch.float16, seed=0, use_graph=True, inflight=NF).to(dev)
batches = [synth.synth_batch(B, seed=1000 + 100 * i) for i in range(NF)]


def rate(run, steps=150):
    for _ in range(3 * NF):
        run.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run.step()
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


def make(h2d):
    r = ShardRunner(net, B, dev, 1, inflight=NF, h2d=h2d)
    for i in range(NF):
        r.load(i, batches[i])
    return r


print("resident        ", round(rate(make(None))))
r = make("crops")
print("full h2d        ", round(rate(r)))


def only_h2d(self, i, cur):          # copies to staging, nothing depends on them
    with torch.cuda.stream(self.copy):
        for k, v in self.pinned[i].items():
            self.staging[i][k].copy_(v, non_blocking=True)
def only_h2d_img(self, i, cur):
    with torch.cuda.stream(self.copy):
        self.staging[i]["roi_img"].copy_(self.pinned[i]["roi_img"], non_blocking=True)
def only_d2d(self, i, cur):
    s = self.net.ensure_stream(i, self.dev)
    with torch.cuda.stream(s):
        for k, v in self.staging[i].items():
            self.statics[i][k].copy_(v, non_blocking=True)
def h2d_dep_no_d2d(self, i, cur):    # H2D + the event dependency, no device copy
    with torch.cuda.stream(self.copy):
        for k, v in self.pinned[i].items():
            self.staging[i][k].copy_(v, non_blocking=True)
        ready = torch.cuda.Event(); ready.record(self.copy)
    self.net.ensure_stream(i, self.dev).wait_event(ready)
for name, fn in (("only H2D (no deps)", only_h2d), ("only H2D of roi_img", only_h2d_img), ("only D2D", only_d2d), ("H2D + dependency, no D2D", h2d_dep_no_d2d)):
    ShardRunner._bring_inputs = fn
    print(f"{name:28s}", round(rate(make("crops"))))


def full_no_reverse_wait(self, i, cur):      # (timing only: the copy stream does not wait for the staging buffer's consumer)
    with torch.cuda.stream(self.copy):
        for k, v in self.pinned[i].items():
            self.staging[i][k].copy_(v, non_blocking=True)
        ready = torch.cuda.Event(); ready.record(self.copy)
    s = self.net.ensure_stream(i, self.dev)
    s.wait_event(ready)
    with torch.cuda.stream(s):
        for k, v in self.staging[i].items():
            self.statics[i][k].copy_(v, non_blocking=True)
def full_big_only(self, i, cur):             # only the three large tensors travel (img, mask, coord): 5 tiny copies fewer each way
    big = ("roi_img", "roi_mask", "roi_coord_2d")
    with torch.cuda.stream(self.copy):
        if self.staged_free[i] is not None:
            self.copy.wait_event(self.staged_free[i])
        for k in big:
            self.staging[i][k].copy_(self.pinned[i][k], non_blocking=True)
        ready = torch.cuda.Event(); ready.record(self.copy)
    s = self.net.ensure_stream(i, self.dev)
    s.wait_event(ready)
    with torch.cuda.stream(s):
        for k in big:
            self.statics[i][k].copy_(self.staging[i][k], non_blocking=True)
        ev = torch.cuda.Event(); ev.record(s); self.staged_free[i] = ev
def full_direct(self, i, cur):               # no staging: H2D straight into the static inputs once the slot's previous forward is done
    s = self.net.ensure_stream(i, self.dev)
    done = torch.cuda.Event(); done.record(s)
    with torch.cuda.stream(self.copy):
        self.copy.wait_event(done)
        for k, v in self.pinned[i].items():
            self.statics[i][k].copy_(v, non_blocking=True)
        ready = torch.cuda.Event(); ready.record(self.copy)
    s.wait_event(ready)
for name, fn in (("full, no reverse wait", full_no_reverse_wait), ("full, 3 large tensors only", full_big_only), ("direct, no staging", full_direct)):
    ShardRunner._bring_inputs = fn
    print(f"{name:28s}", round(rate(make("crops"))))
