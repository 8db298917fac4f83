#!/usr/bin/env python3
"""GPU experiment (round 5): the own-list pair-distance kernel (k_similarity_nz.hip) against the exact planes and against the dense
approximate kernel (k_pairdist_rw), per pyramid level of the bench generator's frames.
usage: tools/exp_nz.py [--quick] [WxH ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def downscale_sum(h):
    H, W, D = h.shape
    return np.ascontiguousarray(h[0:H - H % 2:2, 0:W - W % 2:2] + h[1:H:2, 0:W - W % 2:2] + h[0:H - H % 2:2, 1:W:2] + h[1:H:2, 1:W:2])


def main():
    import torch
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    ctx = bh.Context(0)
    quick = "--quick" in sys.argv
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if not a.startswith("--")] or [(1920, 1080)]
    frames = [("noisy", 32, 0.35, 0.01, 0), ("clean", 32, 0.10, 0.0, 0)]
    if not quick:
        frames += [("textured", 32, 0.35, 0.01, 1), ("8spp", 8, 0.15, 0.0, 0), ("24spp", 24, 0.35, 0.01, 0)]
    for (W, H) in sizes:
        for name, spp, sigma, spikes, pattern in frames:
            col, ns, hist, cov = core.synthetic_scene(W, H, spp, 1234, sigma, spikes, pattern=pattern)
            for scale in range(1 if quick else 3):
                d_hist, d_ns = torch.from_numpy(hist).cuda(), torch.from_numpy(ns).cuda()
                for variant in (0, 1, 2, 3):
                    rel, mism, flags, ms_nz, ms_pl, ms_dense = ctx.selftest_nz_distance(d_hist, d_ns, 6, 1.0, variant, 3)
                    print("%-8s %4dx%-4d scale %d variant %d: max rel dev %.3g count mismatches %d flags %d | own-list %.3f ms (plane-major %.3f) dense %.3f ms -> x%.2f"
                          % (name, hist.shape[1], hist.shape[0], scale, variant, rel, mism, flags, ms_nz, ms_pl, ms_dense, ms_dense / ms_nz), flush=True)
                    p = ctx.nz_prof
                    if p[4] > 0:
                        print("         cycles per workgroup: staging %.1f%% S-pass %.1f%% items %.1f%% | wavefront busy in the item phase %.1f%% | %.1f busy cycles per slot and wavefront, %d slots | counting launch %.3f ms, %.2f G cycles per CU-second"
                              % (100.0 * p[0] / p[4], 100.0 * p[1] / p[4], 100.0 * (p[3] / 16.0) / p[4], 100.0 * p[2] / max(1, p[3]), p[2] / max(1, p[5]), p[5], p[6] / 1e3,
                                 p[4] / 256.0 / max(1, p[6]) / 1e3), flush=True)
                        print("         dense kernel with the general formula on this input: %.3f ms" % (p[7] / 1e3), flush=True)
                del d_hist, d_ns
                hist, ns = downscale_sum(hist), downscale_sum(ns)
    # mixed sample counts (general formula): drop samples per pixel like the parity test does
    W, H = sizes[0]
    rng = np.random.default_rng(5)
    col, ns, hist, cov = core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01)
    keep = rng.choice(np.array([0.5, 0.75, 1.0], np.float32), size=(H, W, 1))
    hist2, ns2 = np.ascontiguousarray(hist * keep), np.ascontiguousarray(ns * keep)
    d_hist, d_ns = torch.from_numpy(hist2).cuda(), torch.from_numpy(ns2).cuda()
    for variant in (0, 1, 2, 3):
        rel, mism, flags, ms_nz, ms_pl, ms_dense = ctx.selftest_nz_distance(d_hist, d_ns, 6, 1.0, variant, 3)
        print("mixed-n  %4dx%-4d variant %d: max rel dev %.3g count mismatches %d flags %d | own-list %.3f ms (plane-major %.3f) dense, general formula %.3f ms"
              % (W, H, variant, rel, mism, flags, ms_nz, ms_pl, ctx.nz_prof[7] / 1e3), flush=True)


if __name__ == "__main__":
    main()
