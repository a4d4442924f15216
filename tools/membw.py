#!/usr/bin/env python3
"""HBM ceilings on this box: fill (write only), copy (read+write), sum (read only); GB/s"""
import torch, time
dev = torch.device("cuda", 0)
n = 1 << 29   # 2 GiB of fp32
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
tf = t(lambda: a.fill_(1.0)); print("fill  %.0f GB/s written" % (4 * n / tf / 1e9))
tc = t(lambda: b.copy_(a)); print("copy  %.0f GB/s read + %.0f GB/s written" % (4 * n / tc / 1e9, 4 * n / tc / 1e9))
ts = t(lambda: a.sum()); print("sum   %.0f GB/s read" % (4 * n / ts / 1e9))
m = 1 << 26
c = torch.empty(m, dtype=torch.float32, device=dev)
tf = t(lambda: c.fill_(1.0)); print("fill 256 MiB %.0f GB/s" % (4 * m / tf / 1e9))
