#!/usr/bin/env python3
"""GPU box: call-to-call latency of the blocking bcd_hip_denoise on resident inputs -- N calls per frame size, every call timed on the host:
median, percentiles and every call slower than 1.5 x the median (a slow call in bench.py's frame_720p leg, 23 ms once, prompted it).
usage: python tools/exp_soak.py [calls_per_size]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bcd_amd.core as core  # noqa: E402
import bcd_amd.hip as bh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ctx = bh.Context(0)
prm = bh.default_params(b=6, w=1, m=1.0, random_order=1, seed=1234)
for (W, H) in ((1280, 720), (1920, 1080), (1280, 720), (3840, 2160)):
    d = [torch.from_numpy(a).cuda() for a in core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01)]
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    calls = n if W < 3000 else max(20, n // 8)
    ts = []
    for i in range(calls):
        t = time.perf_counter()
        ctx.denoise(*d, 3, prm, out)
        ts.append((time.perf_counter() - t) * 1e3)
    a = np.array(ts[3:])
    med = float(np.median(a))
    slow = [(i + 3, round(v, 2)) for i, v in enumerate(a) if v > 1.5 * med]
    print("%dx%d: %d calls, first three %s ms, then median %.3f  p90 %.3f  p99 %.3f  max %.3f ms; calls above 1.5 x median: %s"
          % (W, H, calls, [round(v, 2) for v in ts[:3]], med, np.percentile(a, 90), np.percentile(a, 99), a.max(), slow[:20]), flush=True)
    del d, out
ctx.close()
