#!/usr/bin/env python3
"""PCIe-inclusive rate: bcd_hip_denoise_host (pageable host buffers in, host buffer out) vs the device-resident call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bcd_amd.core as core
import bcd_amd.hip as bh
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1280, 720)
col, ns, hist, cov = core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01)
ctx = bh.Context(0)
prm = bh.default_params()
for _ in range(2):
    ctx.denoise_host(col, ns, hist, cov, 3, prm)
t0 = time.perf_counter(); n = 5
for _ in range(n):
    out = ctx.denoise_host(col, ns, hist, cov, 3, prm)
dt = (time.perf_counter() - t0) / n
mb = (col.nbytes + ns.nbytes + hist.nbytes + cov.nbytes + out.nbytes) / 1e6
print("%dx%d host buffers: %.2f ms/frame = %.1f Mpix/s (%.0f MB over PCIe per frame)" % (W, H, dt * 1e3, W * H / 1e6 / dt, mb))
