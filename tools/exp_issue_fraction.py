#!/usr/bin/env python3
"""CPU model (NumPy, no GPU): which share of the (pixel pair, bin) slots does a 64-lane wavefront of k_pairdist_rw ISSUE under other lane layouts?
A bin is evaluated by a lane when b1 + b2 > 1 (DenoisingUnit.cpp:379) and issued by the wavefront when any of its 64 lanes needs it.  The shipped layout is
64 consecutive pixels x 1 displacement; the alternatives give a wavefront px consecutive pixels x 64 / px consecutive displacements of the half plane.
Reproduces the device counters of the headline frame (25.5 % evaluated, 50.2 % issued, 72.2 % of the groups of four entered: roofline.valu in the bench line)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcd_amd.core as core  # noqa: E402

W, H, b = 256, 128, 6
col, ns, hist, cov = core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01)
hist = hist.reshape(H, W, 60)
deltas = [(0, dc) for dc in range(0, b + 1)] + [(dl, dc) for dl in range(1, b + 1) for dc in range(-b, b + 1)]
rs, cs = slice(8, 8 + 64), slice(16, 16 + 128)


def live(dl, dc):
    return (hist[rs][:, cs] + hist[rs.start + dl:rs.stop + dl][:, cs.start + dc:cs.stop + dc]) > 1.0


L = np.stack([live(dl, dc) for dl, dc in deltas])   # [85, 64, 128, 60]
print("evaluated (per lane): %.3f of the slots" % L.mean())
for px, nd_per in [(64, 1), (32, 2), (16, 4), (8, 8), (4, 16), (2, 32), (1, 64)]:
    tot = cnt = g_tot = g_cnt = 0
    for d0 in range(0, len(deltas), nd_per):
        blk = L[d0:d0 + nd_per]
        n = blk.shape[0]
        x = blk.reshape(n, 64, 128 // px, px, 60).any(axis=(0, 3))
        tot += x.sum() * n * px
        cnt += x.size * n * px
        g = blk.reshape(n, 64, 128 // px, px, 15, 4).any(axis=(0, 3, 5))
        g_tot += g.sum()
        g_cnt += g.size
    print("%2d pixels x %2d displacements per wavefront: bins issued %.3f, groups of four entered %.3f" % (px, nd_per, tot / cnt, g_tot / g_cnt))
