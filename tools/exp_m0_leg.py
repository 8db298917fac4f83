#!/usr/bin/env python3
"""the -m 0 leg of bench.py in isolation: textured frames first (like bench.py), then 1 warm + N timed -m 0 frames"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bcd_amd.core as core
import bcd_amd.hip as bh
W, H, S = 1920, 1080, 3
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = bh.Context(0, stream)
prm = bh.default_params(b=6, w=1, m=1.0, random_order=1, seed=1234)
prm0 = bh.default_params(b=6, w=1, m=0.0, random_order=1, seed=1234)
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
def run(frame, p, reps, tag):
    d = [torch.from_numpy(a).cuda() for a in frame]
    ctx.denoise(*d, S, p, out); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); ctx.denoise(*d, S, p, out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(tag, " ".join("%.2f" % t for t in ts), flush=True)
head = core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01)
if "--head-only" not in sys.argv:
    run(head, prm, 3, "headline")
    run(core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01, pattern=1), prm, 3, "textured")
run(head, prm0, 6, "m0")
