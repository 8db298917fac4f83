#!/usr/bin/env python3
"""Generates the committed golden fixtures (run in the AUTHORING container only; needs /root/reference).

  ref_*.npz     expected outputs computed by the REFERENCE's own compiled translation units (oracle/_ref, built by
                `make -C oracle ref` from /root/reference/src/core/{SamplesAccumulator,SpikeRemovalFilter,Utils,
                CovarianceMatrix,MultiscaleDenoiser}.cpp): samples accumulator, pyramid reducers, interpolate, merge,
                spike filter, histogram/sample-count packing.  These pin the oracle (and, on the GPU box where
                /root/reference does not exist, the HIP kernels) to the reference bit for bit.
  core_patch_trace.npz  (`make_golden.py trace`) SURVEY 8c fixture F3: every intermediate of the two Bayesian steps (noise mean, mean, C,
                C - N, clamped, + N, inverse, Step-1 estimates, Step-2 mean / covariance / inverse / estimates) for three full-estimate
                pixels (smallest, median and largest similar set) and one fallback pixel of the core_regression frame, by the ORACLE.
  core_*.npz    regression vectors of the Eigen-dependent core (distances, similar sets, processed sets, denoised
                frames) computed by the ORACLE itself: the reference core cannot be built here (no Eigen), so these
                are "parity unpinned" snapshots that guard against drift, not reference outputs.
Inputs are seeded (numpy default_rng) and stored in the fixtures, so the files are self-contained data.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402


def main():
    ref = ol.ref_ops()
    assert ref is not None, "oracle/_ref/libbcd_ref.so missing: run `make -C oracle ref` (needs /root/reference)"
    o = ol.oracle_ops()
    # ---- reference-generated fixtures --------------------------------------------------------------------
    W, H, spp = 23, 17, 6  # odd sizes exercise the clamping in downscale*/interpolate
    samples, _ = ol.synth_samples(W, H, spp, seed=42, sigma=0.4, spike_prob=0.05)
    # a few weighted samples pin the weight handling and the bias correction
    samples[::7, 5] = 0.5
    samples[::11, 5] = 2.0
    ns, mean, cov, hist = ref["accumulate"](samples, W, H)
    np.savez_compressed(os.path.join(HERE, "ref_accumulator.npz"), samples=samples, W=W, H=H, ns=ns, mean=mean, cov=cov, hist=hist)
    lo = ref["davg"](mean)
    np.savez_compressed(os.path.join(HERE, "ref_pyramid.npz"), ns=ns, mean=mean, cov=cov, hist=hist,
                        dsum_hist=ref["dsum"](hist), dsum_ns=ref["dsum"](ns), davg_mean=lo, dcov=ref["dcov"](cov, ns),
                        interp=ref["interp"](lo, H, W), merge=ref["merge"](mean, ref["davg"](ref["interp"](lo, H, W))))
    sp = ref["spike"](mean, ns, hist, cov, 2.0)
    assert (sp[0] != mean).any()
    np.savez_compressed(os.path.join(HERE, "ref_spike.npz"), ns=ns, mean=mean, cov=cov, hist=hist, factor=2.0,
                        o_mean=sp[0], o_ns=sp[1], o_hist=sp[2], o_cov=sp[3])
    # the oracle must agree with the reference on all of the above before anything else is trusted
    for k, (a, b) in {"acc": (o["accumulate"](samples, W, H), (ns, mean, cov, hist)), "spike": (o["spike"](mean, ns, hist, cov, 2.0), sp)}.items():
        for x, y in zip(a, b):
            assert np.array_equal(x, y, equal_nan=True), k
    # ---- oracle-generated regression vectors of the core ---------------------------------------------------
    W, H = 40, 28
    col, ns, hist, cov, _ = ol.synth_inputs(W, H, 8, 7, 0.2, 0.01)
    pts = [(1, 1), (14, 20), (26, 38), (5, 33)]
    dist = np.stack([ol.window_distances(ns, hist, 1, 6, l, c) for (l, c) in pts])
    mask, cnt = ol.similarity_masks(ns, hist, 1, 6, 1.0)
    out_m0 = ol.denoise_mono(col, ns, hist, cov, ol.params(m=0.0, threads=1))
    out_m1, (proc, fb, nsim) = ol.denoise_mono(col, ns, hist, cov, ol.params(m=1.0), want_diag=True)
    out_ms = ol.denoise_multiscale(col, ns, hist, cov, 2, ol.params(m=1.0))
    np.savez_compressed(os.path.join(HERE, "core_regression.npz"), col=col, ns=ns, hist=hist, cov=cov, pts=np.array(pts), dist=dist,
                        mask=mask, cnt=cnt, out_m0=out_m0, out_m1=out_m1, processed=proc, fallback=fb, nsim=nsim, out_ms2=out_ms)
    # low-sample-count frame: NaN distances (0/0) and empty similar sets
    # (left half: one sample of weight 0.4 per pixel -> every b1+b2 <= 1 -> 0/0; right half: 3 spp)
    sa, _ = ol.synth_samples(24, 16, 3, seed=3, sigma=0.5, spike_prob=0.0)
    left = sa[:, 1] < 12
    keep = ~left | (np.arange(sa.shape[0]) % 3 == 0)
    sa = sa[keep]
    sa[sa[:, 1] < 12, 5] = 0.4
    ns, col, cov, hist = o["accumulate"](sa, 24, 16)
    cov = np.nan_to_num(cov, nan=0.0, posinf=0.0, neginf=0.0)  # single-sample pixels have an undefined (0/0) covariance
    dist = np.stack([ol.window_distances(ns, hist, 1, 6, l, c) for (l, c) in [(1, 1), (8, 12), (8, 20)]])
    assert np.isnan(dist).any() and np.isfinite(dist[2]).any()
    mask, cnt = ol.similarity_masks(ns, hist, 1, 6, 1.0)
    np.savez_compressed(os.path.join(HERE, "core_lowspp.npz"), col=col, ns=ns, hist=hist, cov=cov, dist=dist, mask=mask, cnt=cnt,
                        out_m1=ol.denoise_mono(col, ns, hist, cov, ol.params(m=1.0)))
    print("fixtures written to", HERE)


def trace_pixels(nsim):
    """three full-estimate pixels (smallest / median / largest similar set) + one fallback pixel of the regression frame"""
    H, W = nsim.shape
    strong = [(int(nsim[l, c]), l, c) for l in range(1, H - 1) for c in range(1, W - 1) if nsim[l, c] >= 28]
    strong.sort()
    weak = [(l, c) for l in range(1, H - 1) for c in range(1, W - 1) if 0 < nsim[l, c] < 28]
    picks = [strong[0][1:], strong[len(strong) // 2][1:], strong[-1][1:], weak[len(weak) // 2]]
    return [(int(l), int(c)) for (l, c) in picks]


def make_trace():
    f = np.load(os.path.join(HERE, "core_regression.npz"))
    col, ns, hist, cov = f["col"], f["ns"], f["hist"], f["cov"]
    _, cnt = ol.similarity_masks(ns, hist, 1, 6, 1.0)
    pts = trace_pixels(cnt)
    out = {"pts": np.array(pts, np.int32)}
    for i, (l, c) in enumerate(pts):
        t = ol.patch_trace(col, ns, hist, cov, ol.params(m=0.0), l, c)
        for k, v in t.items():
            out["p%d_%s" % (i, k)] = v
    np.savez_compressed(os.path.join(HERE, "core_patch_trace.npz"), **out)
    print("trace fixture:", pts, [int(out["p%d_members" % i].size) for i in range(len(pts))])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "trace":
        make_trace()
    else:
        main()
