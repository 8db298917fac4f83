"""CPU tests: the oracle against the committed golden fixtures.

ref_*.npz were computed by the reference's own compiled translation units (tests/golden/make_golden.py); the
oracle must reproduce them bit for bit -- that is the pin.  core_*.npz are oracle-generated regression vectors of
the Eigen-dependent core ("parity unpinned": the reference core is unbuildable without Eigen)."""
import os

import numpy as np
import pytest

import oracle_lib as ol

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


def test_accumulator_matches_reference_bits():
    f = load("ref_accumulator.npz")
    got = ol.oracle_ops()["accumulate"](f["samples"], int(f["W"]), int(f["H"]))
    for g, k in zip(got, ("ns", "mean", "cov", "hist")):
        assert same(g, f[k]), k


def test_pyramid_merge_match_reference_bits():
    f = load("ref_pyramid.npz")
    o = ol.oracle_ops()
    H, W, _ = f["mean"].shape
    assert same(o["dsum"](f["hist"]), f["dsum_hist"])
    assert same(o["dsum"](f["ns"]), f["dsum_ns"])
    assert same(o["davg"](f["mean"]), f["davg_mean"])
    assert same(o["dcov"](f["cov"], f["ns"]), f["dcov"])
    assert same(o["interp"](f["davg_mean"], H, W), f["interp"])
    assert same(o["merge"](f["mean"], o["davg"](f["interp"])), f["merge"])


def test_spike_filter_matches_reference_bits():
    f = load("ref_spike.npz")
    got = ol.oracle_ops()["spike"](f["mean"], f["ns"], f["hist"], f["cov"], float(f["factor"]))
    for g, k in zip(got, ("o_mean", "o_ns", "o_hist", "o_cov")):
        assert same(g, f[k]), k


@pytest.mark.skipif(not ol.ref_available(), reason="oracle/_ref not built (needs /root/reference; authoring container only)")
def test_oracle_equals_live_reference_units():
    """fresh seeded inputs through the compiled reference units and the oracle, bit for bit"""
    o, r = ol.oracle_ops(), ol.ref_ops()
    for seed, (W, H) in enumerate([(31, 22), (16, 16), (3, 3), (50, 9)]):
        samples, _ = ol.synth_samples(W, H, 5, seed=seed, sigma=0.5, spike_prob=0.1)
        a, b = o["accumulate"](samples, W, H), r["accumulate"](samples, W, H)
        assert all(same(x, y) for x, y in zip(a, b))
        ns, mean, cov, hist = a
        if W >= 4 and H >= 4:
            assert same(o["dsum"](hist), r["dsum"](hist)) and same(o["dcov"](cov, ns), r["dcov"](cov, ns))
            lo = o["davg"](mean)
            assert same(lo, r["davg"](mean)) and same(o["interp"](lo, H, W), r["interp"](lo, H, W))
            assert same(o["merge"](mean, lo), r["merge"](mean, lo))
        assert all(same(x, y) for x, y in zip(o["spike"](mean, ns, hist, cov, 1.5), r["spike"](mean, ns, hist, cov, 1.5)))


def test_core_regression_vectors():
    f = load("core_regression.npz")
    col, ns, hist, cov = f["col"], f["ns"], f["hist"], f["cov"]
    for (l, c), want in zip(f["pts"], f["dist"]):
        assert same(ol.window_distances(ns, hist, 1, 6, int(l), int(c)), want)
    mask, cnt = ol.similarity_masks(ns, hist, 1, 6, 1.0)
    assert np.array_equal(mask, f["mask"]) and np.array_equal(cnt, f["cnt"])
    out, (proc, fb, nsim) = ol.denoise_mono(col, ns, hist, cov, ol.params(m=1.0), want_diag=True)
    assert np.array_equal(proc, f["processed"]) and np.array_equal(fb, f["fallback"]) and np.array_equal(nsim, f["nsim"])
    assert np.allclose(out, f["out_m1"], rtol=0, atol=1e-6)
    assert np.allclose(ol.denoise_mono(col, ns, hist, cov, ol.params(m=0.0, threads=1)), f["out_m0"], rtol=0, atol=1e-6)
    assert np.allclose(ol.denoise_multiscale(col, ns, hist, cov, 2, ol.params(m=1.0)), f["out_ms2"], rtol=0, atol=1e-6)


def test_low_sample_count_nan_semantics():
    """1 spp: every bin pair has b1+b2 <= 1 somewhere -> 0/0 distances, empty similar sets, NaN outputs that the
    CLI later zeroes (src/cli/main.cpp:389-420)"""
    f = load("core_lowspp.npz")
    assert np.isnan(f["dist"]).any()
    mask, cnt = ol.similarity_masks(f["ns"], f["hist"], 1, 6, 1.0)
    assert np.array_equal(mask, f["mask"]) and np.array_equal(cnt, f["cnt"])
    out = ol.denoise_mono(f["col"], f["ns"], f["hist"], f["cov"], ol.params(m=1.0))
    assert np.array_equal(np.isnan(out), np.isnan(f["out_m1"]))
    z = out.copy()
    ol.oracle().bcdo_zero_bad_values(ol._fp(z), z.size)
    assert np.isfinite(z).all() and (z >= 0).all()


def test_distance_is_bitwise_symmetric():
    col, ns, hist, cov, _ = ol.synth_inputs(20, 14, 4, 11, 0.4, 0.05)
    import ctypes as C
    lib = ol.oracle()
    rng = np.random.default_rng(0)
    for _ in range(200):
        pl, ql = rng.integers(1, 13, 2)
        pc, qc = rng.integers(1, 19, 2)
        a = lib.bcdo_patch_distance(ol._fp(hist), ol._fp(ns), 20, 14, 60, 1, int(pl), int(pc), int(ql), int(qc))
        b = lib.bcdo_patch_distance(ol._fp(hist), ol._fp(ns), 20, 14, 60, 1, int(ql), int(qc), int(pl), int(pc))
        assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32) or (np.isnan(a) and np.isnan(b))


def test_eigensolver_against_lapack():
    rng = np.random.default_rng(3)
    for n in (3, 27, 28, 75):
        M = rng.standard_normal((n, n + 5)).astype(np.float32)
        A = M @ M.T - 2.0 * np.eye(n, dtype=np.float32)
        ev, V = ol.sym_eig(A)
        ref = np.linalg.eigvalsh(A.astype(np.float64))
        assert np.all(np.diff(ev) >= 0)
        assert np.max(np.abs(ev - ref)) < 2e-5 * np.max(np.abs(ref))
        assert np.max(np.abs(V @ np.diag(ev) @ V.T - A)) < 5e-5 * np.max(np.abs(A))
        assert np.max(np.abs(V.T @ V - np.eye(n))) < 1e-5
    # only the lower triangle is read (Eigen's SelfAdjointEigenSolver contract)
    A2 = A.copy()
    A2[np.triu_indices(n, 1)] = 123.0
    assert np.array_equal(ol.sym_eig(A2)[0], ev)


def test_denoiser_reduces_error_and_orders_matter():
    col, ns, hist, cov, base = ol.synth_inputs(48, 32, 16, 5, 0.3, 0.0)
    rmse = lambda a: float(np.sqrt(np.mean((a - base) ** 2)))
    o0 = ol.denoise_mono(col, ns, hist, cov, ol.params(m=0.0, threads=2))
    o1 = ol.denoise_mono(col, ns, hist, cov, ol.params(m=1.0))
    assert rmse(o0) < 0.7 * rmse(col) and rmse(o1) < 0.7 * rmse(col)
    # -m 0 is order/thread independent up to summation order (1.5e-6)
    o0b = ol.denoise_mono(col, ns, hist, cov, ol.params(m=0.0, threads=1))
    assert np.max(np.abs(o0 - o0b)) / np.max(np.abs(o0b)) < 1e-5


# ---- a10, second reading -------------------------------------------------------------------------------------------------------------
# selectSimilarPatches / histogramPatchDistance / pixelSummedHistogramDistance transcribed from /root/reference/src/core/DenoisingUnit.cpp:196-219,
# 336-358, 360-386 and the window clip of include/bcd/core/DeepImage.hpp:181-196 -- from the reference text, not from oracle/bcd_oracle.c -- in
# np.float32 arithmetic (every NumPy float32 operation is one correctly rounded IEEE operation, never fused: the x86-64 baseline build of the
# reference).  Two forms: a scalar one that IS the reference's loop nest (slow: used on a few main pixels), and an array one that runs the same
# per-element operation sequence for every pixel pair of a displacement at once (bins still one after the other, the nine patch pixels still added
# in row-major order starting from 0.f).

def _np32_patch_distance_scalar(ns, hist, w, p, q):
    """histogramPatchDistance(p, q), statement by statement (:336-358 calling :360-386)"""
    f32 = np.float32
    summed = f32(0.0)                                                    # float summedDistance = 0;
    total = 0                                                            # int totalNbOfNonBoth0Bins = 0;
    with np.errstate(all="ignore"):
        for a in range(-w, w + 1):                                       # PixelPatch iteration: row-major (DeepImage.hpp window iterators)
            for d in range(-w, w + 1):
                h1, h2 = hist[p[0] + a, p[1] + d], hist[q[0] + a, q[1] + d]
                n1, n2 = f32(ns[p[0] + a, p[1] + d, 0]), f32(ns[q[0] + a, q[1] + d, 0])
                nb = 0                                                   # i_rNbOfNonBoth0Bins = 0;
                acc = f32(0.0)                                           # float sum = 0.f;
                for k in range(hist.shape[2]):
                    b1, b2 = f32(h1[k]), f32(h2[k])
                    if b1 + b2 <= f32(1.0):                              # :379 (the "TEMPORARY" criterion is the one compiled)
                        continue
                    nb += 1
                    diff = n2 * b1 - n1 * b2                             # :382
                    acc = acc + diff * diff / (n1 * n2 * (b1 + b2))      # :383
                summed = summed + acc                                    # :354
                total += nb                                              # :355
        return summed / f32(total)                                       # :357 (int -> float, 0 / 0 -> NaN)


def _np32_window_distances(ns, hist, w, b):
    """(H, W, (2b+1)^2) float32: histogramPatchDistance of every main pixel to every pixel of its clipped search window (+inf outside it), the
    array form of the transcription above"""
    H, W, D = hist.shape
    side = 2 * b + 1
    out = np.full((H, W, side * side), np.inf, np.float32)
    n = ns[..., 0]
    with np.errstate(all="ignore"):
        for dl in range(-b, b + 1):
            for dc in range(-b, b + 1):
                # pixel pairs (x, x + (dl, dc)) inside the image
                r0, r1, c0, c1 = max(0, -dl), min(H, H - dl), max(0, -dc), min(W, W - dc)
                if r0 >= r1 or c0 >= c1:
                    continue
                h1, h2 = hist[r0:r1, c0:c1], hist[r0 + dl:r1 + dl, c0 + dc:c1 + dc]
                n1, n2 = n[r0:r1, c0:c1], n[r0 + dl:r1 + dl, c0 + dc:c1 + dc]
                acc = np.zeros(n1.shape, np.float32)
                nb = np.zeros(n1.shape, np.int32)
                for k in range(D):                                       # pixelSummedHistogramDistance, bins in order
                    b1, b2 = h1[..., k], h2[..., k]
                    both = b1 + b2
                    use = ~(both <= np.float32(1.0))                     # "if (b1 + b2 <= 1.f) continue;"
                    diff = n2 * b1 - n1 * b2
                    term = diff * diff / (n1 * n2 * both)
                    acc = np.where(use, acc + term, acc)
                    nb += use
                t = np.full((H, W), np.nan, np.float32)
                cn = np.zeros((H, W), np.int32)
                t[r0:r1, c0:c1], cn[r0:r1, c0:c1] = acc, nb
                # main pixels p whose partner q = p + (dl, dc) lies inside the window clipped to [w, dim - 1 - w] (DeepImage.hpp:181-196)
                pl0, pl1, pc0, pc1 = max(w, w - dl), min(H - 1 - w, H - 1 - w - dl), max(w, w - dc), min(W - 1 - w, W - 1 - w - dc)
                if pl0 > pl1 or pc0 > pc1:
                    continue
                summed = np.zeros((pl1 - pl0 + 1, pc1 - pc0 + 1), np.float32)
                total = np.zeros(summed.shape, np.int32)
                for a in range(-w, w + 1):                               # histogramPatchDistance: patch pixels row-major
                    for d in range(-w, w + 1):
                        summed = summed + t[pl0 + a:pl1 + a + 1, pc0 + d:pc1 + d + 1]
                        total += cn[pl0 + a:pl1 + a + 1, pc0 + d:pc1 + d + 1]
                out[pl0:pl1 + 1, pc0:pc1 + 1, (dl + b) * side + (dc + b)] = summed / total.astype(np.float32)
    return out


def _same_or_both_nan(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def _np32_similar_sets(ns, hist, w, b, tau):
    """selectSimilarPatches (:196-219) for every main pixel from the transcription: (bit masks in the fixture's layout, counts)"""
    dist = _np32_window_distances(ns, hist, w, b)
    H, W, K = dist.shape
    sim = dist <= np.float32(tau)                                        # :209 (NaN and the +inf outside the window: not similar)
    sim[:w], sim[H - w:], sim[:, :w], sim[:, W - w:] = False, False, False, False
    words = (K + 31) // 32
    bits = np.zeros((H, W, words * 32), np.uint8)
    bits[..., :K] = sim
    mask = np.packbits(bits, axis=-1, bitorder="little").view(np.uint32).reshape(H, W, words)
    return mask, sim.sum(axis=-1).astype(np.int32), dist


@pytest.mark.parametrize("fixture,pts", [("core_regression.npz", None), ("core_lowspp.npz", [(1, 1), (7, 11), (14, 22)])])
def test_float32_numpy_transcription_of_the_patch_distance_equals_the_oracle_bit_for_bit(fixture, pts):
    """a10 gets a second reading: the NumPy-float32 transcription of DenoisingUnit.cpp:336-386 above against the C oracle --
    every distance of every main pixel's window bit for bit (NaNs at the same places), the scalar loop nest on the fixture's pixels, and the
    similar sets / counts of selectSimilarPatches against the fixture's masks.  40 x 28 at 8 spp and the 24 x 16 frame at 1 spp (0 / 0 -> NaN)."""
    f = load(fixture)
    ns, hist = f["ns"], f["hist"]
    H, W, _ = hist.shape
    w, b, side = 1, 6, 13
    mask, cnt, dist = _np32_similar_sets(ns, hist, w, b, 1.0)
    for l in range(w, H - w):
        for c in range(w, W - w):
            assert _same_or_both_nan(dist[l, c], ol.window_distances(ns, hist, w, b, l, c)), (l, c)
    assert np.array_equal(mask, f["mask"]) and np.array_equal(cnt, f["cnt"])
    pts = [tuple(int(v) for v in p) for p in (f["pts"] if pts is None else pts)]
    for i, (l, c) in enumerate(pts):                                     # the literal loop nest, on a few main pixels
        if "pts" in f.files:
            assert _same_or_both_nan(dist[l, c], f["dist"][i])
        for k in range(side * side):
            ql, qc = l + k // side - b, c + k % side - b
            if w <= ql <= H - 1 - w and w <= qc <= W - 1 - w:
                assert _same_or_both_nan(_np32_patch_distance_scalar(ns, hist, w, (l, c), (ql, qc)), dist[l, c, k]), (l, c, k)
            else:
                assert np.isinf(dist[l, c, k])
    if fixture == "core_lowspp.npz":
        assert np.isnan(dist[np.isfinite(dist) | np.isnan(dist)]).any()   # the NaN semantics are exercised


def test_float64_numpy_restatement_of_the_bayesian_steps_agrees_with_the_oracle():
    """Independent cross-check of the unpinned core (a11-a16): a float64 NumPy restatement written from the reference source
    (/root/reference/src/core/DenoisingUnit.cpp:400-481 and 483-693, Denoiser.cpp:357-373,434-470) -- not from oracle/bcd_oracle.c --
    with numpy.linalg.eigh in place of Eigen's solver, run on every main pixel of the 40 x 28 regression frame (-m 0) with the similar
    sets of the float32 transcription of a10 above (its own reading end to end), must reproduce the oracle's fp32 image.  It does not pin the oracle to the reference (nothing can without
    Eigen) but removes "single author, single reading" of the two Bayesian steps as a failure mode."""
    f = load("core_regression.npz")
    # (round 6) the similar sets come from the independent float32 transcription of the patch distance above, not from the oracle's masks
    mask, _, _ = _np32_similar_sets(f["ns"], f["hist"], 1, 6, 1.0)
    col, ns, cov, want = f["col"].astype(np.float64), f["ns"].astype(np.float64), f["cov"].astype(np.float64), f["out_m0"]
    H, W, _ = col.shape
    b, w, min_eig = 6, 1, 1e-8
    side = 2 * b + 1
    pixcov = cov * (1.0 / ns)                                            # Denoiser.cpp:357-373
    offs = [(ol_, oc) for ol_ in (-1, 0, 1) for oc in (-1, 0, 1)]        # patch pixels, row-major (DeepImage.hpp window iteration)

    def block(v6):                                                       # CovarianceMatrix.h:18-27: xx,yy,zz,yz,xz,xy
        xx, yy, zz, yz, xz, xy = v6
        return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])

    def patch_vec(img, l, c):                                            # :483-498: pixel-major, RGB
        return np.concatenate([img[l + a, c + d] for (a, d) in offs])

    def spectral(M, fn):                                                 # :578-630
        lam, V = np.linalg.eigh(M)
        return (V * fn(lam)) @ V.T

    acc = np.zeros((H, W, 3))
    cnt = np.zeros((H, W))
    for l in range(w, H - w):
        for c in range(w, W - w):
            bits = np.unpackbits(mask[l, c].view(np.uint8), bitorder="little")[:side * side]
            members = [(l + k // side - b, c + k % side - b) for k in np.nonzero(bits)[0]]   # window order (:196-219)
            n = len(members)
            X = np.stack([patch_vec(col, ql, qc) for (ql, qc) in members])
            if n < 3 * 9 + 1:                                            # :182 -> denoiseOnlyMainPatch (:455-481)
                est = X.mean(axis=0)
                for o, (a, d) in enumerate(offs):
                    acc[l + a, c + d] += est[3 * o:3 * o + 3]
                    cnt[l + a, c + d] += 1
                continue
            N = np.zeros((27, 27))                                       # computeNoiseCovPatchesMean (:400-419)
            for (ql, qc) in members:
                for o, (a, d) in enumerate(offs):
                    N[3 * o:3 * o + 3, 3 * o:3 * o + 3] += block(pixcov[ql + a, qc + d])
            N /= n
            # Step 1 (:421-436)
            m1 = X.mean(axis=0)
            Xc = X - m1
            C = Xc.T @ Xc / (n - 1)
            C1 = spectral(C - N, lambda lam: np.maximum(0.0, lam)) + N
            I1 = spectral(C1, lambda lam: 1.0 / np.maximum(min_eig, lam))
            X1 = X - (N @ (I1 @ Xc.T)).T                                 # finalDenoisingMatrixMultiplication (:656-670)
            # Step 2 (:438-453): no clamp; the noisy patches are centred on the mean of the Step-1 estimates
            m2 = X1.mean(axis=0)
            X1c = X1 - m2
            C2 = X1c.T @ X1c / (n - 1) + N
            I2 = spectral(C2, lambda lam: 1.0 / np.maximum(min_eig, lam))
            X2 = X - (N @ (I2 @ (X - m2).T)).T
            for (ql, qc), est in zip(members, X2):                       # aggregateOutputPatches (:672-693)
                for o, (a, d) in enumerate(offs):
                    acc[ql + a, qc + d] += est[3 * o:3 * o + 3]
                    cnt[ql + a, qc + d] += 1
    got = acc / cnt[..., None]                                           # finalAggregation (Denoiser.cpp:458-469)
    assert (f["fallback"] > 0).any() and (f["nsim"] >= 28).any()         # both paths are exercised on this frame
    err = np.max(np.abs(got - want)) / np.max(np.abs(want))
    assert err < 1e-5, err


def _np64_stages(col, ns, cov, members, W, min_eig=1e-8):
    """float64 restatement of denoiseSelectedPatches written from /root/reference/src/core/DenoisingUnit.cpp:388-453,483-670 (not from
    the oracle): every intermediate, numpy.linalg.eigh in place of Eigen's solver"""
    col, ns, cov = col.astype(np.float64), ns.astype(np.float64), cov.astype(np.float64)
    pixcov = cov * (1.0 / ns)                                            # Denoiser.cpp:357-373
    offs = [(a, d) for a in (-1, 0, 1) for d in (-1, 0, 1)]
    pos = [(int(m) // W, int(m) % W) for m in members]
    X = np.stack([np.concatenate([col[l + a, c + d] for (a, d) in offs]) for (l, c) in pos])
    n = len(pos)
    st = {"x": X, "mean1": X.mean(axis=0)}
    if n < 28:
        return st
    noise6 = np.zeros((9, 6))
    for (l, c) in pos:
        for o, (a, d) in enumerate(offs):
            noise6[o] += pixcov[l + a, c + d]
    noise6 /= n                                                          # :400-419
    N = np.zeros((27, 27))
    for o in range(9):
        xx, yy, zz, yz, xz, xy = noise6[o]                               # CovarianceMatrix.h:18-27
        N[3 * o:3 * o + 3, 3 * o:3 * o + 3] = [[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]]

    def spectral(M, fn):
        lam, V = np.linalg.eigh(M)
        return (V * fn(lam)) @ V.T

    Xc = X - st["mean1"]
    C = Xc.T @ Xc / (n - 1)                                              # :522-536
    clamped = spectral(C - N, lambda lam: np.maximum(0.0, lam))          # :606-630
    inv1 = spectral(clamped + N, lambda lam: 1.0 / np.maximum(min_eig, lam))   # :578-604
    X1 = X - (N @ (inv1 @ Xc.T)).T                                       # :656-670
    m2 = X1.mean(axis=0)
    X1c = X1 - m2
    C2 = X1c.T @ X1c / (n - 1)
    inv2 = spectral(C2 + N, lambda lam: 1.0 / np.maximum(min_eig, lam))
    X2 = X - (N @ (inv2 @ (X - m2).T)).T                                 # :449-450: the NOISY patches centred on the Step-2 mean
    st.update(noise=noise6, cov1=C, cov1_minus_noise=C - N, clamped=clamped, clamped_plus_noise=clamped + N, inverse1=inv1, step1=X1,
              mean2=m2, cov2=C2, inverse2=inv2, step2=X2)
    return st


# stage -> tolerance of the fp32 oracle against the float64 restatement, relative to the largest magnitude of the stage: about 5x the
# measured deviation (noise 2.3e-7, mean1 2.0e-7, cov1 2.0e-7, C - N 2.3e-7, clamped 1.1e-6, inverse1 1.1e-5, step1 1.2e-6, mean2 2.8e-7,
# cov2 1.1e-6, inverse2 1.4e-5, step2 2.0e-6).  The inverses carry the conditioning of (clamped + N).
_STAGE_TOL = dict(noise=1e-6, x=0.0, mean1=1e-6, cov1=1e-6, cov1_minus_noise=1e-6, clamped=5e-6, clamped_plus_noise=5e-6, inverse1=6e-5,
                  step1=6e-6, mean2=1.5e-6, cov2=5e-6, inverse2=7e-5, step2=1e-5)


def test_one_patch_trace_fixture_and_stagewise_float64_cross_check():
    """SURVEY 8c fixture F3.  (1) the oracle reproduces its committed one-patch traces (regression: a change of any stage of
    oracle/bcd_oracle.c shows up AT that stage); (2) every stage agrees with the float64 NumPy restatement above, so a slip in one
    stage of either reading is localised instead of being diluted in a whole-frame norm.  Three full-estimate pixels (|S| = 28, 46,
    91) and one fallback pixel of the 40 x 28 regression frame."""
    f, t = load("core_regression.npz"), load("core_patch_trace.npz")
    col, ns, hist, cov = f["col"], f["ns"], f["hist"], f["cov"]
    H, W, _ = col.shape
    sizes = []
    for i, (l, c) in enumerate(t["pts"]):
        got = ol.patch_trace(col, ns, hist, cov, ol.params(m=0.0), int(l), int(c))
        members = t["p%d_members" % i]
        assert np.array_equal(got["members"], members)
        assert np.array_equal(members, np.nonzero(np.unpackbits(f["mask"][l, c].view(np.uint8), bitorder="little")[:169])[0] // 13 * W
                              + np.nonzero(np.unpackbits(f["mask"][l, c].view(np.uint8), bitorder="little")[:169])[0] % 13
                              + (l - 6) * W + (c - 6))                     # window order of the similar set (:196-219)
        sizes.append(members.size)
        ref64 = _np64_stages(col, ns, cov, members, W)
        for k in ol.TRACE_FIELDS:
            want = t["p%d_%s" % (i, k)]
            assert np.allclose(got[k], want, rtol=0, atol=1e-6 * max(1e-30, float(np.max(np.abs(want))))), (i, k)   # regression
            if k in ref64:
                scale = float(np.max(np.abs(ref64[k])))
                err = float(np.max(np.abs(got[k].astype(np.float64) - ref64[k]))) / scale
                assert err <= _STAGE_TOL[k], (i, k, err)
            else:
                assert members.size < 28 and not got[k].any()            # fallback pixel: the Bayesian stages are not run
    assert sorted(sizes) == [7, 28, 46, 91]
    # the traces are the frame: aggregating the Step-2 estimates / fallback means of EVERY main pixel gives out_m0 (checked on one pixel's
    # contributions here: the final image test above covers the sum)
    l, c = [int(v) for v in t["pts"][0]]
    assert np.allclose(t["p0_step2"][list(t["p0_members"]).index(l * W + c)][12:15], f["out_m0"][l, c], atol=0.2)


def test_phased_ordered_visit_equals_the_serial_one():
    """bcdo_denoise_mono with an order / a skip probability and nb_threads > 1 runs the same visit in three phases (similar sets in parallel, the
    decisions of the visit sequentially, the estimates in parallel): same processed set, same marks, results equal to the one-thread loop up to the
    summation order of the per-thread accumulators -- this is what lets the GPU tests compare the 1080p bench frame with its marking order"""
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H, S = 96, 72, 2
    col, ns, hist, cov = core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01)
    orders, w_, h_ = [], W, H
    for s_ in range(S):
        orders.append(bh.visit_order(w_, h_, 1, 1, bh.scale_seed(5, s_)))
        w_, h_ = w_ // 2, h_ // 2
    for m in (1.0, 0.5):
        serial = ol.denoise_multiscale(col, ns, hist, cov, S, ol.params(m=m, skip_seed=7), orders=orders)
        phased = ol.denoise_multiscale(col, ns, hist, cov, S, ol.params(m=m, skip_seed=7, threads=4), orders=orders)
        assert np.max(np.abs(serial - phased)) / np.max(np.abs(serial)) < 5e-6
    serial = ol.denoise_mono(col, ns, hist, cov, ol.params(m=1.0))          # scanline order
    phased = ol.denoise_mono(col, ns, hist, cov, ol.params(m=1.0, threads=4))
    assert np.max(np.abs(serial - phased)) / np.max(np.abs(serial)) < 5e-6


def test_first_order_correction_of_the_positive_part_after_an_early_stop():
    """the mathematics behind the early-stopped eigensolver (k_bayes27.hip, pos_part_lds): with A = V (D + E) V^T, E the residual a Jacobi solver leaves
    off the diagonal, the positive part V max(0, L) V^T of clampNegativeEigenValues (DenoisingUnit.cpp:606-630) is V (max(0, D) + E o Phi) V^T up to
    second order, Phi the divided differences of max(0, .) at the diagonal -- checked in float64 on the clamp inputs C - N of the one-patch trace
    fixture, at the solver's stopping rule off^2 <= 2e-9 diag^2: the corrected form is within 1e-6 |A| of the exact positive part and never worse than
    the plain one; where no two estimates of opposite sign lie closer together than the residual is large it is better by orders of magnitude (the
    pairs it cannot help are why the rule is not looser: k_bayes27.hip)"""
    f = np.load(os.path.join(G, "core_patch_trace.npz"))
    mats = [f[k].astype(np.float64) for k in f.files if k.endswith("cov1_minus_noise")]
    assert len(mats) >= 3

    def sweep(A, V):
        n = A.shape[0]
        for p in range(n - 1):
            for q in range(p + 1, n):
                if A[p, q] == 0.0:
                    continue
                th = (A[q, q] - A[p, p]) / (2.0 * A[p, q])
                t = (1.0 if th >= 0 else -1.0) / (abs(th) + np.sqrt(th * th + 1.0))
                c = 1.0 / np.sqrt(t * t + 1.0)
                s = t * c
                J = np.eye(n)
                J[p, p] = J[q, q] = c
                J[p, q], J[q, p] = s, -s
                A, V = J.T @ A @ J, V @ J
        return A, V

    checked, gains = 0, []
    for A0 in mats:
        if not np.any(A0):
            continue                                     # (the fallback pixel of the fixture has no clamp input)
        w, v = np.linalg.eigh(A0)
        exact = (v * np.maximum(w, 0.0)) @ v.T
        A, V = A0.copy(), np.eye(A0.shape[0])
        for _ in range(8):
            A, V = sweep(A, V)
            d = np.diag(A)
            off = np.sqrt(max(0.0, (A * A).sum() - (d * d).sum())) / np.linalg.norm(d)
            if off * off <= 2e-9:
                break
        assert 1e-12 < off                               # an early stop, not a converged solve
        plain = (V * np.maximum(d, 0.0)) @ V.T
        di, dj = d[:, None], d[None, :]
        hi, lo = np.maximum(di, dj), np.minimum(di, dj)
        with np.errstate(divide="ignore", invalid="ignore"):
            phi = np.where(lo > 0, 1.0, np.where(hi <= 0, 0.0, hi / (hi - lo)))
        M = A * phi
        np.fill_diagonal(M, np.maximum(d, 0.0))
        corrected = V @ M @ V.T
        nrm = np.linalg.norm(A0, 2)
        e_plain, e_corr = np.abs(plain - exact).max() / nrm, np.abs(corrected - exact).max() / nrm
        assert e_corr < 1e-6 and e_corr <= e_plain * 1.0001, (off, e_plain, e_corr)
        gains.append(e_plain / max(e_corr, 1e-300))
        checked += 1
    assert checked >= 3 and max(gains) > 20.0
