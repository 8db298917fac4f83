#!/usr/bin/env python3
"""GPU box: the streaming stages of the path (pyramid reducers, interpolate, merge, spike filter, per-pixel covariances, samples accumulator) on seeded random
geometries against the CPU oracle -- which is pinned bit for bit to the reference's own compiled units for exactly these stages (tests/golden/ref_*.npz) -- so
every comparison is BIT FOR BIT (accumulator histograms: float round-off of the device powf, as in the GPU test).  Test infrastructure: the oracle is the checker.
usage: python tools/fuzz_streaming.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol  # noqa: E402
import bcd_amd.hip as bh  # noqa: E402


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    if a.shape != b.shape:
        return False
    ua, ub = a.view(np.uint32), b.view(np.uint32)
    return bool(np.all((ua == ub) | (np.isnan(a) & np.isnan(b))))


def run_cases(ctx, n_cases, seed, say=print):
    import torch
    dev = lambda *arrs: [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrs]
    rng = np.random.default_rng(seed)
    o = ol.oracle_ops()
    bad = 0
    for case in range(n_cases):
        W, H = int(rng.integers(4, 260)), int(rng.integers(4, 180))
        spp = int(rng.choice([1, 2, 3, 5, 8, 16]))
        nbins = int(rng.choice([4, 7, 12, 20, 20, 40]))
        sigma, spike = float(rng.choice([0.05, 0.35, 0.8])), float(rng.choice([0.0, 0.02, 0.1]))
        factor = float(rng.choice([1.0, 2.0, 3.5]))
        weighted = bool(rng.random() < 0.5)
        samples, _ = ol.synth_samples(W, H, spp, int(rng.integers(1, 1 << 20)), sigma, spike)
        if weighted:
            samples[:: int(rng.integers(2, 6)), 5] = float(rng.choice([0.25, 0.5, 3.0]))
        ns, col, cov, hist = o["accumulate"](samples, W, H, nbins)
        fails = []
        s4 = samples.reshape(H, W, spp, 6)
        d_s, d_w = dev(s4[..., 2:5], s4[..., 5])
        g = [t.cpu().numpy() for t in ctx.accumulate_samples(d_s, d_w if weighted else None, nbins)]
        if not (bits_equal(g[0], ns) and bits_equal(g[1], col) and bits_equal(g[2], cov)):
            fails.append("accumulator n / mean / cov")
        if not np.max(np.abs(g[3] - hist)) < 2e-5 * max(1.0, float(np.max(hist))):
            fails.append("accumulator histogram")
        d_col, d_ns, d_hist, d_cov = dev(col, ns, hist, cov)
        want = np.empty_like(cov)
        ol.oracle().bcdo_pixel_cov_from_sample_cov(ol._fp(cov), ol._fp(ns), W, H, ol._fp(want))
        if not bits_equal(ctx.pixel_cov(d_cov, d_ns).cpu().numpy(), want):
            fails.append("pixel covariances")
        if W >= 2 and H >= 2:
            for name, got, want in (("downscale_sum(hist)", ctx.downscale_sum(d_hist), o["dsum"](hist)), ("downscale_sum(n)", ctx.downscale_sum(d_ns), o["dsum"](ns)),
                                    ("downscale_avg(colour)", ctx.downscale_avg(d_col), o["davg"](col)), ("downscale_avg(hist)", ctx.downscale_avg(d_hist), o["davg"](hist)),
                                    ("downscale_cov", ctx.downscale_cov(d_cov, d_ns), o["dcov"](cov, ns))):
                if not bits_equal(got.cpu().numpy(), want):
                    fails.append(name)
            lo = o["davg"](col)
            (d_lo,) = dev(lo)
            if not bits_equal(ctx.interpolate(d_lo, H, W).cpu().numpy(), o["interp"](lo, H, W)):
                fails.append("interpolate")
            if not bits_equal(ctx.merge(d_col, d_lo).cpu().numpy(), o["merge"](col, lo)):
                fails.append("merge")
        if W >= 3 and H >= 3:
            want = o["spike"](col, ns, hist, cov, factor)
            got = ctx.spike_filter(d_col, d_ns, d_hist, d_cov, factor)
            if not all(bits_equal(a.cpu().numpy(), b) for a, b in zip(got, want)):
                fails.append("spike filter")
        bad += 1 if fails else 0
        say("%4d: %3dx%-3d spp=%-2d bins=%-2d%s sigma=%.2f spikes=%.2f factor=%.1f  %s" % (case, W, H, spp, nbins, " weighted" if weighted else "", sigma, spike, factor,
                                                                                          "bit-exact" if not fails else "DIFFER: " + ", ".join(fails) + "   <-- MISMATCH"))
    return bad


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    ctx = bh.Context(0)
    t0 = time.time()
    bad = run_cases(ctx, n_cases, seed, say=lambda s: print(s, flush=True))
    print("%d cases, %d mismatches, %.0f s" % (n_cases, bad, time.time() - t0))
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
