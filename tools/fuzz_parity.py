#!/usr/bin/env python3
"""GPU box: seeded random configurations of the hot path against the CPU oracle (test infrastructure: the oracle is the checker).
Frame sizes (ragged, narrower than a tile, odd), scales, search radius, threshold, sample counts (1 .. 48, sometimes mixed per pixel), noise level, -m 0 / -m 1,
-r 0 / -r 1.  Checks per case: similarity masks and |S| of the finest scale bit for bit, the denoised frame's finite pattern and relative L-inf < 1e-4.
usage: python tools/fuzz_parity.py [n_cases] [seed] [--big] [--only=i,j,..] [--strict] [--dump=dir] [--bands] [--wide] [--rccl] [--nan]   -> one line per case, a summary, exit code 1 on any mismatch
(--big: frames up to 700 x 400; --wide: also host buffers / spike prefilter / fractional -m / other depths, patch radii and -e; --nan: a few non-finite input values, checked: the HIP result's non-finite set is a subset of the oracle's; --rccl: also one rank of the band driver with RCCL in loopback against the single-GPU frame; --bands: also the row-band driver with 2 .. 4 virtual ranks against the single-GPU frame; --only: evaluate these cases of the sequence; --strict: bcd_hip_set_strict_eigensolver)"""
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol  # noqa: E402
import bcd_amd.hip as bh  # noqa: E402


def cases(n_cases, seed, only=None, big=False):
    """the seeded sequence of configurations: dicts with the parameters and the four input images (built only for the cases asked for)"""
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        S = int(rng.choice([1, 2, 3]))
        b = int(rng.choice([1, 2, 3, 4, 6, 6, 6, 9, 12]))
        W = int(rng.integers(max(12, 4 << (S - 1)), 700 if big else 150))
        H = int(rng.integers(max(12, 4 << (S - 1)), 400 if big else 110))
        spp = int(rng.choice([1, 2, 3, 4, 8, 16, 24, 32, 48]))
        sigma = float(rng.choice([0.05, 0.15, 0.35, 0.6]))
        spike = float(rng.choice([0.0, 0.01, 0.05]))
        tau = float(rng.choice([0.5, 1.0, 1.0, 1.0, 2.0]))
        m = float(rng.choice([0.0, 1.0, 1.0]))
        ro = int(rng.choice([0, 1]))
        seed_c = int(rng.integers(1, 1 << 20))
        mixed = bool(rng.random() < 0.25) and spp >= 4
        keep = rng.choice(np.array([0.25, 0.5, 0.75, 1.0], np.float32), size=(H, W, 1)).astype(np.float32) if mixed else None
        if only is not None and case not in only:
            continue
        col, ns, hist, cov, _ = ol.synth_inputs(W, H, spp, seed_c, sigma, spike)
        if mixed:  # thinned per pixel: what an adaptive sampler's early exit leaves
            ns2 = np.ascontiguousarray(np.maximum(1.0, np.rint(ns * keep[..., 0] if ns.ndim == 2 else ns * keep)).astype(np.float32))
            hist = np.ascontiguousarray(hist * (ns2 / ns).reshape(H, W, 1)).astype(np.float32)
            ns = ns2
        yield dict(case=case, S=S, b=b, W=W, H=H, spp=spp, sigma=sigma, spike=spike, tau=tau, m=m, ro=ro, seed=seed_c, mixed=mixed, col=col, ns=ns, hist=hist, cov=cov)


def visiting_orders(c, w=1):
    if c["m"] == 0.0:
        return None
    orders, w_, h_ = [], c["W"], c["H"]
    for s in range(c["S"]):
        orders.append(bh.visit_order(w_, h_, w, c["ro"], bh.scale_seed(c["seed"], s)))
        w_, h_ = w_ // 2, h_ // 2
    return orders


def run_cases(ctx, n_cases, seed, say=print, only=None, big=False, dump=None, bands=False, wide=False, rccl=False, nan=False):
    """n_cases seeded random configurations through `ctx`; returns (mismatches, refused, worst relative L-inf)"""
    import torch
    bad, refused, worst = 0, 0, 0.0
    rng_b = np.random.default_rng(seed + 1)   # (a stream of its own: the sequence of cases does not depend on --bands)
    for c in cases(n_cases, seed, only, big):
        case, S, b, W, H, spp, sigma, spike, tau, m, ro, mixed = (c[k] for k in ("case", "S", "b", "W", "H", "spp", "sigma", "spike", "tau", "m", "ro", "mixed"))
        col, ns, hist, cov = c["col"], c["ns"], c["hist"], c["cov"]
        if nan:  # a few non-finite input values (renderers produce them): the finite pattern of the result must be the oracle's
            nr = np.random.default_rng([seed, 3, case])
            col, cov = col.copy(), cov.copy()
            for _ in range(int(nr.integers(1, 4))):
                l_, c_ = int(nr.integers(0, H)), int(nr.integers(0, W))
                which = int(nr.integers(0, 3))
                if which == 0:
                    col[l_, c_, int(nr.integers(0, 3))] = np.float32(nr.choice([np.nan, np.inf, -np.inf]))
                elif which == 1:
                    cov[l_, c_, int(nr.integers(0, 6))] = np.float32(nr.choice([np.nan, np.inf]))
                else:
                    col[l_, c_, :] = np.nan
        # --wide: the other entry points and parameters, drawn from a stream of their own (the sequence of frames stays the plain one)
        path, w, min_eig, skip_seed, nbins, note = 0, 1, 1e-8, 0, 20, ""
        if wide:
            wr = np.random.default_rng([seed, 2, case])
            path = int(wr.choice([0, 1, 2]))                       # resident buffers / host buffers (streamed upload) / host buffers + spike prefilter
            if wr.random() < 0.25:
                m = c["m"] = float(wr.choice([0.3, 0.7]))          # fractional -m: per-pixel hash of the seed, mirrored by the oracle
                skip_seed = c["seed"]
            if wr.random() < 0.25:
                nbins = int(wr.choice([4, 7, 8, 12, 40]))          # D = 12, 21, 24, 36, 120
                samples, _ = ol.synth_samples(W, H, spp, c["seed"], sigma, spike)
                ns, col, cov, hist = ol.oracle_ops()["accumulate"](samples, W, H, nbins)
                mixed = False
            if wr.random() < 0.15:
                w = int(wr.choice([0, 2]))                         # generic mask / marking / estimate kernels
                b = max(b, w + 1)
                if w == 2 and (min(W, H) >> (S - 1)) < 8:
                    S = c["S"] = 1
            if wr.random() < 0.2:
                min_eig = 1e-3                                     # inverses through the spectral branch
            note = " path=%d w=%d D=%d e=%g" % (path, w, 3 * nbins, min_eig)
        tag = "%3d: %3dx%-3d S=%d b=%-2d spp=%-2d%s sigma=%.2f spikes=%.2f tau=%.1f -m %g -r %d%s" % (case, W, H, S, b, spp, "*" if mixed else " ", sigma, spike, tau, m, ro, note)
        prm = bh.default_params(m=m, random_order=ro, seed=c["seed"], b=b, tau=tau, w=w, min_eig=min_eig)
        col, ns, hist, cov = (np.ascontiguousarray(a, np.float32) for a in (col, ns, hist, cov))
        ocol, ons, ohist, ocov = col, ns, hist, cov                # what the oracle denoises (the filtered frame when the prefilter is on)
        try:
            if path == 0:
                d = [torch.from_numpy(a).cuda() for a in (col, ns, hist, cov)]
                got = ctx.denoise(*d, S, prm).cpu().numpy()
            elif path == 1:
                got = ctx.denoise_host(col, ns, hist, cov, S, prm)
            else:
                got = ctx.denoise_host(col, ns, hist, cov, S, prm, spike_factor=2.0)
                ocol, ons, ohist, ocov = ol.oracle_ops()["spike"](col, ns, hist, cov, 2.0)
        except bh.BcdHipError as e:
            refused += 1
            say(tag + "  refused: %s" % str(e)[:90])
            continue
        d_hist, d_ns = torch.from_numpy(np.ascontiguousarray(ohist)).cuda(), torch.from_numpy(np.ascontiguousarray(ons)).cuda()
        mask, cnt = ctx.similarity_masks(d_hist, d_ns, w, b, tau)
        wmask, wcnt = ol.similarity_masks(ons, ohist, w, b, tau)
        masks_ok = np.array_equal(mask.cpu().numpy().view(np.uint32), wmask) and np.array_equal(cnt.cpu().numpy(), wcnt)
        if dump:
            np.save(os.path.join(dump, "fuzz_%d_%d.npy" % (seed, case)), got)
        orders = visiting_orders(c, w)
        op = ol.params(tau=tau, w=w, b=b, m=m, min_eig=min_eig, skip_seed=skip_seed)
        try:
            want = (ol.denoise_multiscale(ocol, ons, ohist, ocov, S, op, orders=orders) if S > 1 else
                    ol.denoise_mono(ocol, ons, ohist, ocov, op, order=orders[0] if orders else None))
        except AssertionError as e:   # (the oracle declines a geometry the engine accepted: reported, counted as a mismatch)
            bad += 1
            say(tag + "  oracle declined (rc %s)   <-- MISMATCH" % e)
            continue
        ok = np.isfinite(want)
        fin_ok = np.array_equal(np.isfinite(got), ok)
        if nan and not fin_ok:
            # Non-finite INPUT values cannot be pinned to the reference (what it does with a NaN matrix is what Eigen's solver leaves in its eigenvectors; its CLI
            # zeroes non-finite values of the OUTPUT, src/cli/main.cpp:389-470); what is checked: the HIP path never yields a non-finite value where the oracle's is finite, and agrees wherever both are finite
            hip_bad, ora_bad = ~np.isfinite(got), ~ok
            fin_ok = not bool((hip_bad & ~ora_bad).any())
            tag += "  [non-finite: oracle %d values, HIP %d, HIP only %d]" % (int(ora_bad.sum()), int(hip_bad.sum()), int((hip_bad & ~ora_bad).sum()))
            ok = ok & np.isfinite(got)
        scale = float(np.max(np.abs(np.where(ok, want, 0)))) or 1.0
        with np.errstate(invalid="ignore"):
            err = float(np.max(np.abs(np.where(ok, got, 0) - np.where(ok, want, 0))) / scale) if ok.any() else 0.0
        worst = max(worst, err)
        good = masks_ok and fin_ok and err < 1e-4
        band_note = ""
        if bands and path != 2:  # the row-band driver with 2 .. 4 (--big: 2 .. 8) virtual ranks on this device (in-process transport) against the single-GPU frame
            ranks = int(rng_b.integers(2, 9 if big else 5))
            md = bh.MultiDenoiser([0] * ranks)
            try:
                gb = md.denoise_host(col, ns, hist, cov, S, prm)
                eb = float(np.max(np.abs(np.where(ok, gb, 0) - np.where(ok, got, 0))) / scale) if ok.any() else 0.0
                band_ok = np.array_equal(np.isfinite(gb), np.isfinite(got)) and eb < 1e-5
                good = good and band_ok
                band_note = "  %d bands vs one GPU %.1e" % (ranks, eb)
            except bh.BcdHipError as e:   # (a frame too short for that many bands of that pyramid is declined, not mis-denoised)
                band_note = "  %d bands declined: %s" % (ranks, str(e)[:60])
            finally:
                md.close()
        if rccl and path != 2:  # rank 0 of 1 in loopback: every exchange / all-reduce of the band protocol on real RCCL communicators, the rank its own neighbour
            try:
                rd = bh.RankDenoiser(0, 1, 0, bh.multi_unique_ids(S + 1))
                try:
                    rd.set_loopback(True)
                    rd.configure(W, H, hist.shape[2], S, prm)
                    rd.upload(col, ns, hist, cov)
                    rd.step()
                    gr = rd.download()
                    rd.step()                      # (steady state: the same communicators, a second frame)
                    gr2 = rd.download()
                    tr = rd.stats().transport
                finally:
                    rd.close()
                er = max(float(np.max(np.abs(np.where(ok, g_, 0) - np.where(ok, got, 0))) / scale) if ok.any() else 0.0 for g_ in (gr, gr2))
                good = good and tr == 1 and er < 1e-5
                band_note += "  RCCL loopback vs one GPU %.1e" % er
            except bh.BcdHipError as e:
                band_note += "  RCCL loopback declined: %s" % str(e)[:60]
        bad += 0 if good else 1
        say(tag + "  masks %s  finite %s  rel Linf %.2e%s%s" % ("==" if masks_ok else "DIFFER", "==" if fin_ok else "DIFFER", err, band_note, "" if good else "   <-- MISMATCH"))
    return bad, refused, worst


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_cases = int(argv[0]) if len(argv) > 0 else 60
    seed = int(argv[1]) if len(argv) > 1 else 2026
    only, dump = None, None
    for a in sys.argv[1:]:
        if a.startswith("--only="):
            only = set(int(x) for x in a[7:].split(","))
        if a.startswith("--dump="):
            dump = a[7:]
            os.makedirs(dump, exist_ok=True)
    ctx = bh.Context(0)
    if "--strict" in sys.argv:
        bh.set_strict_eigensolver(True)    # the eigensolver's plain stopping rule (1e-12) instead of 2e-9 + first-order correction
    t0 = time.time()
    bad, refused, worst = run_cases(ctx, n_cases, seed, say=lambda s: print(s, flush=True), only=only, big="--big" in sys.argv, dump=dump, bands="--bands" in sys.argv, wide="--wide" in sys.argv, rccl="--rccl" in sys.argv, nan="--nan" in sys.argv)
    print("%d cases, %d refused, %d mismatches, worst rel Linf %.2e, %.0f s" % (n_cases, refused, bad, worst, time.time() - t0))
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
