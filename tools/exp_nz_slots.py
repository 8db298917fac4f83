#!/usr/bin/env python3
"""Host-side slot count of the sparse ("own non-zero bins") form of the pair-distance sum, per lane mapping.

The reference's bin rule (src/core/DenoisingUnit.cpp:379-383) skips a bin when b1 + b2 <= 1.  For a bin with b1 == 0 the term is
(n1/n2) b2 if b2 > 1, else nothing, so with NZ(x) = {k : b1_k > 0}, S(y) = sum_k [b2_k > 1] b2_k and C1(y) = #{k : b2_k > 1}:
    T(x,y) = sum_{k in NZ(x)} ([b1+b2>1] t_k - (n1/n2) [b2>1] b2_k) + (n1/n2) S(y)
    C(x,y) = sum_{k in NZ(x)} ([b1+b2>1] - [b2>1]) + C1(y)
This script counts, on the bench generator's frames (no GPU needed), the loop lengths ("slots") each lane mapping would issue per pixel pair:
  dense            the shipped kernel: 60 slots, lanes = 64 pixels of a line (useful / issued fractions as k_pairdist_rw counts them)
  own-list, lanes = displacements of ONE own pixel: slots = |NZ(x)| (wave-uniform bin index, no max-over-lanes loss)
  own-list, lanes = 3 own pixels x 21 displacements (the 21 displacements that do not fit the first 64 lanes): slots = |NZ(x1) u NZ(x2) u NZ(x3)|
  own-list, lanes = 64 pixels of a line (each lane its own list): slots = max over the 64 lanes of |NZ|
and the fraction of those slots whose term is live (b1 + b2 > 1).
usage: tools/exp_nz_slots.py [--lines 48] [--scales 3]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcd_amd.core as core  # noqa: E402


def downscale_sum(h):
    H, W, D = h.shape
    return h[0:H - H % 2:2, 0:W - W % 2:2] + h[1:H:2, 0:W - W % 2:2] + h[0:H - H % 2:2, 1:W:2] + h[1:H:2, 1:W:2]


def analyse(hist, b, rng, npix=600):
    H, W, D = hist.shape
    nz = hist > 0
    cnt = nz.sum(-1)
    # union of three horizontally adjacent pixels
    u3 = (nz[:, 0:W - 2] | nz[:, 1:W - 1] | nz[:, 2:W]).sum(-1)
    u2 = (nz[:, 0:W - 1] | nz[:, 1:W]).sum(-1)
    # lanes = 64 pixels of a line: the wave's loop bound is the longest list
    wcols = (W // 64) * 64
    mx64 = cnt[:, :wcols].reshape(H, -1, 64).max(-1)
    out = {"nz_mean": cnt.mean(), "nz_p95": np.percentile(cnt, 95), "nz_max": cnt.max(), "union3": u3.mean(), "union2": u2.mean(), "max64": mx64.mean()}
    # live fractions on a sample of own pixels x forward displacements
    ys = rng.integers(0, H - b, npix)
    xs = rng.integers(b, W - b, npix)
    slots = live = dense_useful = b1zero_live = 0
    issued = 0
    for r, c in zip(ys, xs):
        h1 = hist[r, c]
        m = h1 > 0
        for dl in range(0, b + 1):
            for dc in range(-b, b + 1):
                if dl == 0 and dc <= 0:
                    continue
                h2 = hist[r + dl, c + dc]
                s = h1 + h2
                lv = s > 1.0
                slots += int(m.sum())
                live += int((lv & m).sum())
                dense_useful += int(lv.sum())
                b1zero_live += int((lv & ~m).sum())
    npairs = npix * (b + b * (2 * b + 1))
    out.update({"slots_per_pair": slots / npairs, "live_frac": live / max(1, slots), "dense_useful_frac": dense_useful / (npairs * D),
                "live_terms_per_pair": dense_useful / npairs, "b1zero_live_per_pair": b1zero_live / npairs})
    # issued fraction of the dense kernel: a bin is issued when any of the 64 pixels of a line segment needs it (sampled)
    iss = tot = 0
    for _ in range(40):
        r = int(rng.integers(0, H - b))
        c0 = int(rng.integers(b, max(b + 1, W - 64 - b)))
        dl = int(rng.integers(0, b + 1))
        dc = int(rng.integers(-b, b + 1))
        if dl == 0 and dc <= 0:
            dc = 1
        s = hist[r, c0:c0 + 64] + hist[r + dl, c0 + dc:c0 + dc + 64]
        iss += int((s > 1.0).any(0).sum())
        tot += D
    out["dense_issued_frac"] = iss / tot
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=48)
    ap.add_argument("--scales", type=int, default=3)
    ap.add_argument("--b", type=int, default=6)
    a = ap.parse_args()
    rng = np.random.default_rng(7)
    frames = [("noisy 1080p 32spp s=0.35 spikes 1%", dict(W=1920, H=1080, spp=32, sigma=0.35, spike_prob=0.01, pattern=0)),
              ("clean 1080p 32spp s=0.10", dict(W=1920, H=1080, spp=32, sigma=0.10, spike_prob=0.0, pattern=0)),
              ("textured 1080p 32spp s=0.35 spikes 1%", dict(W=1920, H=1080, spp=32, sigma=0.35, spike_prob=0.01, pattern=1)),
              ("4K 8spp s=0.15", dict(W=3840, H=2160, spp=8, sigma=0.15, spike_prob=0.0, pattern=0)),
              ("1080p 24spp s=0.35 spikes 1%", dict(W=1920, H=1080, spp=24, sigma=0.35, spike_prob=0.01, pattern=0))]
    # instruction model (wave64 VALU issue slots, v_rcp_f32 = 3): dense kernel 2 per slot for the test + 8 per ISSUED bin;
    # own-list: 14 per slot (address, s, d, rcp x3, d^2, cmp, select, fma, count, [b2>1] indicator, its count, its sum)
    for name, p in frames:
        nl = a.lines * (1 << (a.scales - 1))
        first = (p["H"] // 2 // 16) * 16
        _, _, hist, _ = core.synthetic_scene(p["W"], p["H"], p["spp"], 1234, p["sigma"], p["spike_prob"], first_line=first, nb_lines=nl, pattern=p["pattern"])
        print("== %s (lines %d..%d)" % (name, first, first + nl))
        h = hist
        for s in range(a.scales):
            r = analyse(h, a.b, rng)
            per_px_own = r["nz_mean"] + r["union3"] / 3.0
            dense_instr = 60 * 2 + 60 * r["dense_issued_frac"] * 8
            print("  scale %d: |NZ| mean %.1f p95 %.0f max %d | union2 %.1f union3 %.1f | max over 64 lanes %.1f | slots/pair %.1f live %.1f%% | dense: useful %.1f%% issued %.1f%% "
                  "live terms/pair %.1f (b1=0 live %.2f)" % (s, r["nz_mean"], r["nz_p95"], r["nz_max"], r["union2"], r["union3"], r["max64"], r["slots_per_pair"], 100 * r["live_frac"],
                                                           100 * r["dense_useful_frac"], 100 * r["dense_issued_frac"], r["live_terms_per_pair"], r["b1zero_live_per_pair"]))
            dense_px = 85 * dense_instr / 64.0
            own_px = per_px_own * 14
            pix_px = 85 * r["max64"] * 15 / 64.0
            print("           wave instructions per own pixel: dense %.0f | own-list, lanes = displacements (1 px x 64 + 3 px x 21): (%.1f + %.1f/3) x 14 = %.0f (x%.2f) | "
                  "own lists, lanes = pixels: %.0f (x%.2f)" % (dense_px, r["nz_mean"], r["union3"], own_px, dense_px / own_px, pix_px, dense_px / pix_px))
            h = downscale_sum(h)


if __name__ == "__main__":
    main()
