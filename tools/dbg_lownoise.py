import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bcd_amd.core as core, bcd_amd.hip as bh
ctx = bh.Context(0)
prm = bh.default_params(b=6, w=1, m=1.0, random_order=1, seed=1234)
W, H = 640, 360
for sigma, spk in ((0.35, 0.01), (0.10, 0.0), (0.35, 0.01)):
    fr = core.synthetic_scene(W, H, 32, 1234, sigma, spk)
    d = [torch.from_numpy(a).cuda() for a in fr]
    out = ctx.denoise(*d, 3, prm)
    torch.cuda.synchronize()
    print(sigma, "hist mean nz", float((fr[2] > 0).sum(-1).mean()), [(ctx.stats(s).processed, ctx.stats(s).fallback, ctx.stats(s).similar_total, ctx.stats(s).borderline_pairs) for s in range(3)])
