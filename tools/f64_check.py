#!/usr/bin/env python3
"""CPU: where does a single-scale frame of tools/fuzz_parity.py sit against FLOAT64 arithmetic?  The two Bayesian steps restated in NumPy float64
(the restatement of tests/test_oracle_golden.py, written from /root/reference/src/core/DenoisingUnit.cpp:400-481,483-693; numpy.linalg.eigh for the spectral
steps) on the oracle's own similar sets and processed set, compared with (a) the float32 oracle and (b) the HIP result dumped by
`tools/fuzz_parity.py ... --dump=dir` on the GPU box.  Answers: is a 1e-4 difference between the HIP path and the oracle an error of one of them, or the
float32 conditioning of the frame?   usage: python tools/f64_check.py n_cases seed --only=i,j [--dump=dir]   (single-scale cases only)"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, os.path.join(R, "tools"))
import oracle_lib as ol  # noqa: E402
import fuzz_parity as fz  # noqa: E402


def f64_frame(col, ns, cov, mask, processed, b, min_eig=1e-8):
    col, ns, cov = col.astype(np.float64), ns.astype(np.float64).reshape(col.shape[0], col.shape[1], 1), cov.astype(np.float64)
    H, W, _ = col.shape
    side = 2 * b + 1
    pixcov = cov * (1.0 / ns)
    offs = [(a, d) for a in (-1, 0, 1) for d in (-1, 0, 1)]

    def block(v6):
        xx, yy, zz, yz, xz, xy = v6
        return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])

    def spectral(M, fn):
        lam, V = np.linalg.eigh(M)
        return (V * fn(lam)) @ V.T

    acc, cnt = np.zeros((H, W, 3)), np.zeros((H, W))
    worst_cond = 0.0
    for l in range(1, H - 1):
        for c in range(1, W - 1):
            if not processed[l, c]:
                continue
            bits = np.unpackbits(mask[l, c].view(np.uint8), bitorder="little")[:side * side]
            members = [(l + k // side - b, c + k % side - b) for k in np.nonzero(bits)[0]]
            n = len(members)
            X = np.stack([np.concatenate([col[ql + a, qc + d] for (a, d) in offs]) for (ql, qc) in members])
            if n < 28:
                est = X.mean(axis=0)
                for o, (a, d) in enumerate(offs):
                    acc[l + a, c + d] += est[3 * o:3 * o + 3]
                    cnt[l + a, c + d] += 1
                continue
            N = np.zeros((27, 27))
            for (ql, qc) in members:
                for o, (a, d) in enumerate(offs):
                    N[3 * o:3 * o + 3, 3 * o:3 * o + 3] += block(pixcov[ql + a, qc + d])
            N /= n
            m1 = X.mean(axis=0)
            Xc = X - m1
            C = Xc.T @ Xc / (n - 1)
            C1 = spectral(C - N, lambda lam: np.maximum(0.0, lam)) + N
            lam1 = np.linalg.eigvalsh(C1)
            worst_cond = max(worst_cond, float(lam1[-1] / max(min_eig, lam1[0])))
            I1 = spectral(C1, lambda lam: 1.0 / np.maximum(min_eig, lam))
            X1 = X - (N @ (I1 @ Xc.T)).T
            m2 = X1.mean(axis=0)
            X1c = X1 - m2
            C2 = X1c.T @ X1c / (n - 1) + N
            I2 = spectral(C2, lambda lam: 1.0 / np.maximum(min_eig, lam))
            X2 = X - (N @ (I2 @ (X - m2).T)).T
            for (ql, qc), est in zip(members, X2):
                for o, (a, d) in enumerate(offs):
                    acc[ql + a, qc + d] += est[3 * o:3 * o + 3]
                    cnt[ql + a, qc + d] += 1
    with np.errstate(invalid="ignore", divide="ignore"):
        return acc / cnt[..., None], worst_cond


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_cases, seed = int(argv[0]), int(argv[1])
    only, dump = None, None
    for a in sys.argv[1:]:
        if a.startswith("--only="):
            only = set(int(x) for x in a[7:].split(","))
        if a.startswith("--dump="):
            dump = a[7:]
    for c in fz.cases(n_cases, seed, only):
        if c["S"] != 1:
            print("%d: S = %d, skipped (single-scale cases only)" % (c["case"], c["S"]))
            continue
        orders = fz.visiting_orders(c)
        op = ol.params(tau=c["tau"], b=c["b"], m=c["m"])
        want, (processed, fallback, nsim) = ol.denoise_mono(c["col"], c["ns"], c["hist"], c["cov"], op, order=orders[0] if orders else None, want_diag=True)
        mask, _ = ol.similarity_masks(c["ns"], c["hist"], 1, c["b"], c["tau"])
        ref, cond = f64_frame(c["col"], c["ns"], c["cov"], mask, processed, c["b"])
        ok = np.isfinite(want) & np.isfinite(ref)
        scale = np.max(np.abs(ref[ok]))
        e_oracle = np.max(np.abs(np.where(ok, want - ref, 0))) / scale
        line = "%4d: %dx%d b=%d spp=%d sigma=%.2f  processed %d (full %d)  worst cond(C1) %.1e   oracle(f32) vs f64 %.2e" % (
            c["case"], c["W"], c["H"], c["b"], c["spp"], c["sigma"], int(processed.sum()), int((processed & (fallback == 0)).sum()), cond, e_oracle)
        f = os.path.join(dump, "fuzz_%d_%d.npy" % (seed, c["case"])) if dump else None
        if f and os.path.exists(f):
            got = np.load(f)
            line += "   HIP vs f64 %.2e   HIP vs oracle %.2e" % (np.max(np.abs(np.where(ok, got - ref, 0))) / scale, np.max(np.abs(np.where(ok, got - want, 0))) / np.max(np.abs(want[ok])))
        print(line, flush=True)


if __name__ == "__main__":
    main()
