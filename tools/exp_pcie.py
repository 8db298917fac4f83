#!/usr/bin/env python3
"""GPU experiment: host-to-device rates of pageable, registered (hipHostRegister) and pinned host memory, and what registering costs."""
import time

import numpy as np
import torch

n = 500 * 1024 * 1024 // 4
a = np.random.rand(n).astype(np.float32)
d = torch.empty(n, dtype=torch.float32, device="cuda")
t = torch.from_numpy(a)


def rate(src, reps=5):
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        d.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return a.nbytes / best / 1e9


print("pageable  : %.1f GB/s" % rate(t))
rt = torch.cuda.cudart()
t0 = time.perf_counter()
rc = rt.cudaHostRegister(t.data_ptr(), t.numel() * 4, 0)
t1 = time.perf_counter()
print("hipHostRegister of 500 MB: rc %s, %.1f ms" % (rc, (t1 - t0) * 1e3))
print("registered: %.1f GB/s" % rate(t))
t0 = time.perf_counter()
rt.cudaHostUnregister(t.data_ptr())
print("unregister: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
p = torch.empty(n, dtype=torch.float32).pin_memory()
p.copy_(t)
print("pinned    : %.1f GB/s" % rate(p))
h = torch.empty(n, dtype=torch.float32)
torch.cuda.synchronize()
t0 = time.perf_counter()
h.copy_(d)
torch.cuda.synchronize()
print("D2H pageable: %.1f GB/s" % (a.nbytes / (time.perf_counter() - t0) / 1e9))
