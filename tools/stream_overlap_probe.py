#!/usr/bin/env python
"""What a fork/join between two HIP streams costs on this box (needs a GPU): a ~20-us kernel A and a ~8-us kernel B per
iteration, (i) both on one stream, (ii) A and B on two streams with an event each way, B overlapping A, (iii) the same
without any dependency (upper bound of the overlap).  Decides whether the halo exchange of a brick step could hide
behind interior pair work (docs/history/round4.md, "what comes next" 1)."""
import time

import torch

dev = torch.device("cuda:0")
a = torch.randn(48 * 1024 * 1024 // 4, device=dev)   # ~20 us of streaming at ~5 TB/s (read + write)
b = torch.randn(16 * 1024 * 1024 // 4, device=dev)   # ~8 us
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
N = 2000


def run(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        if mode == "serial":
            with torch.cuda.stream(s1):
                a.mul_(1.0001)
                b.mul_(1.0001)
        elif mode == "forkjoin":
            with torch.cuda.stream(s1):
                e1 = s1.record_event()
                a.mul_(1.0001)
            with torch.cuda.stream(s2):
                s2.wait_event(e1)
                b.mul_(1.0001)
                e2 = s2.record_event()
            s1.wait_event(e2)
        else:
            with torch.cuda.stream(s1):
                a.mul_(1.0001)
            with torch.cuda.stream(s2):
                b.mul_(1.0001)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6


for mode in ("serial", "forkjoin", "free", "serial", "forkjoin", "free"):
    print(f"{mode:9s} {run(mode):7.2f} us per iteration")
