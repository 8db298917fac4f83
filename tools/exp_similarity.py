#!/usr/bin/env python3
"""GPU experiment: the similar-patch selection stage alone (pair-distance planes + masks) with the fast path
(approximate planes + exact verification at the threshold) and with the exact kernels, on frames of the benchmark
generator.  Prints per size: stage time, pair-distance kernel time (HIP events), borderline pairs, mask equality."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    ctx = bh.Context(0)
    quick = "--quick" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    sizes = [(1280, 720), (1920, 1080), (640, 360), (320, 180)] if not args else [tuple(int(v) for v in a.split("x")) for a in args]
    if quick:
        # fast path only, one size: for sweeps over BCD_CS_MODE and for counter passes
        W, H = sizes[0] if args else (1280, 720)
        for sigma, spikes in ((0.35, 0.01), (0.10, 0.0)):
            col, ns, hist, cov = core.synthetic_scene(W, H, 32, 1234, sigma, spikes)
            d_hist, d_ns = torch.from_numpy(hist).cuda(), torch.from_numpy(ns).cuda()
            for _ in range(2):
                ctx.similarity_masks(d_hist, d_ns, 1, 6, 1.0)
            torch.cuda.synchronize()
            ctx.reset_kernel_time()
            for _ in range(5):
                ctx.similarity_masks(d_hist, d_ns, 1, 6, 1.0)
            kms, kn = ctx.kernel_time()
            print("mode %s sigma %.2f %dx%d pairdist %.3f ms" % (os.environ.get("BCD_CS_MODE", "default"), sigma, W, H, kms / max(1, kn)), flush=True)
        return
    for sigma, spikes in ((0.35, 0.01), (0.10, 0.0)):
        for (W, H) in sizes:
            col, ns, hist, cov = core.synthetic_scene(W, H, 32, 1234, sigma, spikes)
            d_hist, d_ns = torch.from_numpy(hist).cuda(), torch.from_numpy(ns).cuda()
            res = {}
            for fast in (1, 0):
                ctx.set_fast_similarity(fast)
                for _ in range(2):
                    m, c = ctx.similarity_masks(d_hist, d_ns, 1, 6, 1.0)
                torch.cuda.synchronize()
                ctx.reset_kernel_time()
                t0 = time.perf_counter()
                n = 5
                for _ in range(n):
                    m, c = ctx.similarity_masks(d_hist, d_ns, 1, 6, 1.0)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n * 1e3
                kms, kn = ctx.kernel_time()
                res[fast] = (m.cpu().numpy(), c.cpu().numpy(), dt, kms / max(1, kn))
            same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
            rel, mism, flags = ctx.selftest_approx_distance(d_hist, d_ns, 6)
            print("sigma %.2f %4dx%-4d  fast: stage %.3f ms pairdist %.3f ms | exact: stage %.3f ms pairdist %.3f ms | masks equal %s | "
                  "max rel dev %.3g (delta %.3g) count mismatches %d flags %d" %
                  (sigma, W, H, res[1][2], res[1][3], res[0][2], res[0][3], same, rel, 2.0 ** -14, mism, flags), flush=True)
    ctx.set_fast_similarity(1)


if __name__ == "__main__":
    main()
