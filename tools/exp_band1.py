#!/usr/bin/env python3
"""GPU experiment: the band driver with ONE rank (bcd_hip_multi_rank_* as bench.py --gpus N uses it) -- overhead of the driver itself"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bcd_amd.core as core
import bcd_amd.hip as bh
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
col, ns, hist, cov = core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01)
prm = bh.default_params(b=6, w=1, m=1.0, random_order=1, seed=1234)
rd = bh.RankDenoiser(0, 1, 0, None)
rd.configure(W, H, 60, 3, prm)
rd.upload(col, ns, hist, cov)
for _ in range(3):
    rd.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    rd.step()
torch.cuda.synchronize()
print("band driver, one rank, %dx%d: %.2f ms per step" % (W, H, (time.perf_counter() - t0) * 1e3 / n))
rd.close()
