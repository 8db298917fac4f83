"""CPU tests of libbcdcore (the C++ mirror of the reference's include/bcd API): SamplesAccumulator and Utils against
the oracle / reference fixtures, synthetic scenes, and IDenoiser input validation (no GPU needed: validation runs
before any device work)."""
import os

import numpy as np

import bcd_amd.core as core
import oracle_lib as ol

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


def test_samples_accumulator_matches_reference_fixture():
    f = np.load(os.path.join(G, "ref_accumulator.npz"))
    got = core.accumulate(f["samples"], int(f["W"]), int(f["H"]))
    for g, k in zip(got, ("ns", "mean", "cov", "hist")):
        assert same(g, f[k]), k


def test_samples_accumulator_other_histogram_parameters():
    samples, _ = ol.synth_samples(12, 9, 7, seed=2, sigma=0.6, spike_prob=0.2)
    for nbins, gamma, maxval in [(20, 2.2, 2.5), (10, 1.0, 1.0), (32, 3.0, 0.0)]:
        a = core.accumulate(samples, 12, 9, nbins, gamma, maxval)
        b = ol.oracle_ops()["accumulate"](samples, 12, 9, nbins, gamma, maxval)
        assert all(same(x, y) for x, y in zip(a, b))
        assert np.allclose(a[3].sum(-1), 3 * a[0][..., 0])       # each sample adds weight 1 to each channel's histogram


def test_histogram_packing_roundtrip():
    f = np.load(os.path.join(G, "ref_pyramid.npz"))
    m = core.merge_hist_ns(f["hist"], f["ns"])
    assert m.shape[-1] == 61 and same(m[..., 60:], f["ns"]) and same(m[..., :60], f["hist"])
    h, n = core.split_hist_ns(m)
    assert same(h, f["hist"]) and same(n, f["ns"])


def test_synthetic_scene_is_seeded_and_band_consistent():
    a = core.synthetic_scene(40, 30, 4, seed=5)
    b = core.synthetic_scene(40, 30, 4, seed=5)
    c = core.synthetic_scene(40, 30, 4, seed=6)
    assert all(same(x, y) for x, y in zip(a, b)) and not same(a[0], c[0])
    band = core.synthetic_scene(40, 30, 4, seed=5, first_line=8, nb_lines=13)
    assert all(same(x[8:21], y) for x, y in zip(a, band))
    col, ns, hist, cov = a
    assert (ns == 4).all() and np.isfinite(cov).all() and (hist >= 0).all()


def test_denoiser_rejects_bad_inputs_like_the_reference():
    col, ns, hist, cov = core.synthetic_scene(16, 12, 2)
    assert core.denoise(None, ns, hist, cov)[0] is False             # nullptr input (Denoiser.cpp:266-293)
    assert core.denoise(col, ns, hist, cov, hist_width_override=8)[0] is False  # size mismatch (:321-346)
    assert core.denoise(col[:0], ns[:0], hist[:0], cov[:0])[0] is False  # empty (:294-320)
