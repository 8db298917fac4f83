#!/usr/bin/env python3
"""GPU check at full size: the shipping estimate kernels against the comparison kernels (BCD_HIP_FINISH_LDS=1, BCD_HIP_JACOBI_PAIRS=1: read once per
process, so each variant runs in a child) on the same 1080p frame, -m 0 (1.6 M full estimates) and -m 1: relative L-inf of the differences."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import sys; sys.path.insert(0, %r)
import numpy as np, torch, bcd_amd.core as core, bcd_amd.hip as bh
W, H = 1920, 1080
ctx = bh.Context(0)
for tag, m, pat in (("m0", 0.0, 0), ("m1", 1.0, 0), ("tex", 1.0, 1)):
    col, ns, hist, cov = core.synthetic_scene(W, H, 32, 99, 0.35, 0.01, pattern=pat)
    d = [torch.from_numpy(a).cuda() for a in (col, ns, hist, cov)]
    out = ctx.denoise(*d, 3, bh.default_params(m=m, random_order=1, seed=11)).cpu().numpy()
    np.save(sys.argv[1] + "_" + tag + ".npy", out)
    print(tag, [ctx.stats(s).spectral_inverses for s in range(3)])
""" % ROOT


def run(name, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", CODE, "/tmp/ab_" + name], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    print(name, out.stdout.strip().replace("\n", " | "))


run("ship", {})
run("lds", {"BCD_HIP_FINISH_LDS": "1"})
run("pairs", {"BCD_HIP_JACOBI_PAIRS": "1"})
for tag in ("m0", "m1", "tex"):
    a = np.load("/tmp/ab_ship_%s.npy" % tag)
    for other in ("lds", "pairs"):
        b = np.load("/tmp/ab_%s_%s.npy" % (other, tag))
        print("%-4s ship vs %-5s: rel L-inf %.3e, finite %s" % (tag, other, np.max(np.abs(a - b)) / np.max(np.abs(b)), bool(np.isfinite(a).all() and np.isfinite(b).all())))
