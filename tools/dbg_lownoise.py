#!/usr/bin/env python3
"""Debug probe (GPU): per-scale counters of the headline frame and of the low-noise frame at 1080p, through two fresh contexts in
both orders, plus the raw |S| statistics of scale 0 from the mask kernel -- to explain why bench.py's `low_noise` leg showed the
headline's scale-0 counters."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bcd_amd.core as core
import bcd_amd.hip as bh

W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1920x1080").split("x"))
prm = bh.default_params(b=6, w=1, m=1.0, random_order=1, seed=1234)
frames = {"sigma0.35+spikes": core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01), "sigma0.10": core.synthetic_scene(W, H, 32, 1234, 0.10, 0.0)}
for name, f in frames.items():
    print(name, "hist sum", float(f[2].sum()), "hist nonzero frac", float((f[2] > 0).mean()), "col std", float(f[0].std()))
for order in (list(frames), list(frames)[::-1]):
    ctx = bh.Context(0)
    for name in order:
        d = [torch.from_numpy(a).cuda() for a in frames[name]]
        ctx.denoise(*d, 3, prm)
        ctx.denoise(*d, 3, prm)
        print(name, [(ctx.stats(s).processed, ctx.stats(s).fallback, ctx.stats(s).similar_total, ctx.stats(s).borderline_pairs) for s in range(3)])
        mask, cnt = ctx.similarity_masks(d[2], d[1], 1, 6, 1.0)
        c = cnt[1:-1, 1:-1]
        print("   scale 0 |S|: mean %.3f  <28: %d  >=28: %d  hist of |S|//16: %s" % (float(c.float().mean()), int((c < 28).sum()), int((c >= 28).sum()),
              torch.bincount((c // 16).flatten(), minlength=11).tolist()))
    ctx.close()
