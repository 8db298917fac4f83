#!/usr/bin/env python3
"""GPU experiment: the mask stage (pair-distance planes, forward masks, verification, symmetric masks) of one scale, for per-kernel timing under
rocprofv3 --kernel-trace.  usage: exp_masks.py WxH b spp sigma spikes [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H = (int(v) for v in sys.argv[1].split("x"))
    b, spp, sigma, spikes = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5])
    reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
    ctx = bh.Context(0)
    col, ns, hist, cov = core.synthetic_scene(W, H, spp, 1234, sigma, spikes)
    d_hist, d_ns = torch.from_numpy(hist).cuda(), torch.from_numpy(ns).cuda()
    for _ in range(reps + 1):
        m, c = ctx.similarity_masks(d_hist, d_ns, 1, b, 1.0)
    torch.cuda.synchronize()
    print("masks checksum", int(c.sum().item()), int(m.to(torch.int64).sum().item()))


if __name__ == "__main__":
    main()
