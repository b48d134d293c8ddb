import torch, sys
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (16, 64, 256, 1024):
    x = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    y = torch.empty_like(x)
    us = t(lambda: x.zero_())
    uc = t(lambda: y.copy_(x))
    ur = t(lambda: x.view(torch.float16).sum())
    print(f"{mb:5d} MB: fill {us:7.1f} us {mb*1.048576/us*1e3:6.0f} GB/s | copy {uc:7.1f} us {2*mb*1.048576/uc*1e3:6.0f} GB/s (r+w) | read-sum {ur:7.1f} us {mb*1.048576/ur*1e3:6.0f} GB/s")
