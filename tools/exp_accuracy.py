import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bcd_amd.core as core, bcd_amd.hip as bh, oracle_lib as ol
ctx = bh.Context(0)
def orders(W, H, seed, S):
    out = []
    for s in range(S):
        out.append(bh.visit_order(W >> s, H >> s, 1, 1, bh.scale_seed(seed, s)))
    return out
def run(name, W, H, spp, gseed, sigma, spikes, seed, S=3, pattern=0, m=1.0):
    col, ns, hist, cov = core.synthetic_scene(W, H, spp, gseed, sigma, spikes, pattern=pattern)
    prm = bh.default_params(m=m, random_order=1, seed=seed)
    got = ctx.denoise(*[torch.from_numpy(a).cuda() for a in (col, ns, hist, cov)], S, prm).cpu().numpy()
    thr = min(128, os.cpu_count() or 1)
    want = ol.denoise_multiscale(col, ns, hist, cov, S, ol.params(m=m, skip_seed=seed, threads=thr), orders=orders(W, H, seed, S) if m > 0 else None)
    ok = np.isfinite(want)
    e = float(np.max(np.abs(np.where(ok, got, 0) - np.where(ok, want, 0))) / np.max(np.abs(np.where(ok, want, 0))))
    print("%s: rel Linf vs oracle %.2e (full estimates scale0: %d)" % (name, e, ctx.stats(0).processed - ctx.stats(0).fallback), flush=True)
if "--only4k" not in sys.argv: run("quarter-hd noisy", 480, 270, 32, 1234, 0.35, 0.01, 1234)
if "--only4k" not in sys.argv: run("textured 640x360", 640, 360, 32, 1234, 0.35, 0.0, 3, pattern=1)
if "--only4k" not in sys.argv: run("720p m0", 1280, 720, 32, 1234, 0.35, 0.01, 1234, m=0.0)
run("4k config3 (8 spp)", 3840, 2160, 8, 3, 0.15, 0.0, 17)
