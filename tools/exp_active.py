#!/usr/bin/env python3
"""experiment: time the marking launches (active_init + active_step batches) on the bench frame"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bcd_amd.core as core
import bcd_amd.hip as bh

W, H = 1280, 720
col, ns, hist, cov = core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01)
ctx = bh.Context(0)
d = [torch.from_numpy(a).cuda() for a in (col, ns, hist, cov)]
mask, cnt = ctx.similarity_masks(d[2], d[1], 1, 6, 1.0)
torch.cuda.synchronize()
for rep in range(3):
    state = ctx.active_init(cnt, 1, 0, H, 1.0, 99, 0)
    torch.cuda.synchronize()
    ts = []
    first = True
    while True:
        t0 = time.perf_counter()
        und = ctx.active_step(mask, cnt, state, 1, 6, 0, H, 1, 99, 0, first); n = 2
        ts.append(((time.perf_counter() - t0) * 1e6, und, n))
        first = False
        if und == 0:
            break
    print("iters", os.environ.get("BCD_ACTIVE_ITERS", "6"), " ".join("%.0fus(und %d, %d launches)" % t for t in ts))
