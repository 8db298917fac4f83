#!/usr/bin/env python3
"""GPU experiment: row bands INSIDE one GPU -- K concurrent band pipelines (the multi-GPU driver with K ranks on one device, in-process
transport) against the single pipeline.  Times the resident-data part of a frame (stats.compute_ms) next to bcd_hip_denoise."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bcd_amd.core as core
import bcd_amd.hip as bh
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
col, ns, hist, cov = core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01)
prm = bh.default_params(b=6, w=1, m=1.0, random_order=1, seed=1234)
ctx = bh.Context(0)
d = [torch.from_numpy(a).cuda() for a in (col, ns, hist, cov)]
for _ in range(3):
    ref = ctx.denoise(*d, 3, prm)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    ref = ctx.denoise(*d, 3, prm)
torch.cuda.synchronize()
print("single pipeline: %.2f ms" % ((time.perf_counter() - t0) * 100))
ref = ref.cpu().numpy()
for K in (1, 2, 3, 4, 6):
    md = bh.MultiDenoiser([0] * K)
    try:
        ms = []
        for i in range(5):
            out = md.denoise_host(col, ns, hist, cov, 3, prm)
            ms.append(md.stats().compute_ms)
        err = float(np.max(np.abs(out - ref)) / np.max(np.abs(ref)))
        print("%d band(s) on one GPU: %.2f ms (best of 5 after warm-up; rel. Linf vs single %.1e)" % (K, min(ms[1:]), err))
    finally:
        md.close()
