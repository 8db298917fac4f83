"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Integer / discrete results (masks, counts, processed sets) and order-exact float stages must be bit-exact;
the denoised colours must be within 1e-4 relative L-infinity (BASELINE.json north_star)."""
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

TOL = 1e-4  # relative L-inf on in-memory fp32 buffers (north_star)
# The full-size frames have a budget of their own (round 5): they measure 1e-6 ... 1e-5 against the oracle (9.8e-6 on the 4K 8-spp frame since the
# eigensolver stops at 2e-9 with a first-order correction), so a further loosening of a solver or an accumulation order is a red test here long
# before it reaches the north-star bar.
TOL_FULL_SIZE = 3e-5


def rel_linf(a, b):
    v = float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
    if os.environ.get("BCD_TEST_REPORT"):  # (a log of every compared pair: how far below the bar the suite runs)
        with open(os.environ["BCD_TEST_REPORT"], "a") as f:
            f.write("%s %.3e\n" % (os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], v))
    return v


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.uint32)
    b = np.ascontiguousarray(b, np.float32).view(np.uint32)
    nan = np.isnan(a.view(np.float32)) & np.isnan(b.view(np.float32))
    return bool(np.all((a == b) | nan))


_cache = {}


def inputs(W, H, spp=16, sigma=0.35, spike=0.01, seed=1234):
    key = (W, H, spp, sigma, spike, seed)
    if key not in _cache:
        _cache[key] = ol.synth_inputs(W, H, spp, seed, sigma, spike)
    return _cache[key]


def dev(*arrs):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrs]


def test_pixel_cov_bitexact(hipctx):
    col, ns, hist, cov, _ = inputs(70, 37)
    d_cov, d_ns = dev(cov, ns)
    got = hipctx.pixel_cov(d_cov, d_ns).cpu().numpy()
    want = np.empty_like(cov)
    ol.oracle().bcdo_pixel_cov_from_sample_cov(ol._fp(cov), ol._fp(ns), 70, 37, ol._fp(want))
    assert bits_equal(got, want)


@pytest.mark.parametrize("W,H,spp,b", [(70, 37, 16, 6), (130, 21, 2, 6), (66, 50, 8, 3), (41, 33, 4, 12)])
def test_window_distances_bitexact(hipctx, W, H, spp, b):
    col, ns, hist, cov, _ = inputs(W, H, spp)
    d_hist, d_ns = dev(hist, ns)
    rng = np.random.default_rng(7)
    pts = [(1, 1), (H - 2, W - 2), (1, W - 2), (H // 2, W // 2)] + [(int(rng.integers(1, H - 1)), int(rng.integers(1, W - 1))) for _ in range(6)]
    for (l, c) in pts:
        got = hipctx.window_distances(d_hist, d_ns, 1, b, l, c)
        want = ol.window_distances(ns, hist, 1, b, l, c)
        assert bits_equal(got, want), (l, c)


@pytest.mark.parametrize("W,H,spp,sigma,b,tau", [(70, 37, 16, 0.35, 6, 1.0), (129, 21, 2, 0.35, 6, 1.0), (64, 48, 32, 0.10, 6, 1.0),
                                                 (45, 31, 8, 0.2, 12, 0.8), (9, 7, 8, 0.2, 6, 1.5), (3, 3, 8, 0.2, 6, 1.0)])
def test_similarity_masks_bitexact(hipctx, W, H, spp, sigma, b, tau):
    col, ns, hist, cov, _ = inputs(W, H, spp, sigma)
    d_hist, d_ns = dev(hist, ns)
    mask, cnt = hipctx.similarity_masks(d_hist, d_ns, 1, b, tau)
    wmask, wcnt = ol.similarity_masks(ns, hist, 1, b, tau)
    assert np.array_equal(mask.cpu().numpy().view(np.uint32), wmask)
    assert np.array_equal(cnt.cpu().numpy(), wcnt)


def _orders(W, H, w, random_order, seed, nscales):
    import bcd_amd.hip as bh
    out = []
    for s in range(nscales):
        out.append(bh.visit_order(W, H, w, random_order, bh.scale_seed(seed, s)))
        W, H = W // 2, H // 2
    return out


@pytest.mark.parametrize("random_order", [0, 1])
@pytest.mark.parametrize("sigma", [0.35, 0.08])
def test_processed_set_matches_reference_order(hipctx, random_order, sigma):
    """the parallel fixed point reproduces the sequential marking strategy exactly (same visiting order)"""
    import bcd_amd.hip as bh
    W, H = 72, 50
    col, ns, hist, cov, _ = inputs(W, H, 16, sigma, 0.0)
    d_hist, d_ns = dev(hist, ns)
    mask, cnt = hipctx.similarity_masks(d_hist, d_ns, 1, 6, 1.0)
    state, rounds = hipctx.active_set(mask, cnt, 1, 6, 1.0, random_order, 77)
    order = bh.visit_order(W, H, 1, random_order, 77)
    _, (proc, fb, nsim) = ol.denoise_mono(col, ns, hist, cov, ol.params(m=1.0), order=order, want_diag=True)
    st = state.cpu().numpy()
    assert np.array_equal(st == 1, proc == 1)
    assert rounds > 0


@pytest.mark.parametrize("m,random_order,sigma,spp", [(0.0, 0, 0.35, 16), (0.0, 0, 0.08, 32), (1.0, 0, 0.08, 32), (1.0, 1, 0.08, 32),
                                                      (1.0, 1, 0.35, 16), (1.0, 1, 0.35, 2)])
def test_mono_parity(hipctx, m, random_order, sigma, spp):
    import bcd_amd.hip as bh
    W, H = 80, 56
    col, ns, hist, cov, _ = inputs(W, H, spp, sigma)
    prm = bh.default_params(m=m, random_order=random_order, seed=5)
    d = dev(col, ns, hist, cov)
    got = hipctx.denoise(*d, 1, prm).cpu().numpy()
    order = _orders(W, H, 1, random_order, 5, 1)[0] if m != 0.0 else None
    want = ol.denoise_mono(col, ns, hist, cov, ol.params(m=m), order=order)
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), ok)
    assert rel_linf(np.where(ok, got, 0), np.where(ok, want, 0)) < TOL


@pytest.mark.parametrize("W,H,S", [(72, 66, 1), (90, 131, 2)])
def test_strip_visiting_order_against_the_oracle(hipctx, W, H, S):
    """pixel order 2 = the list the reference builds for -r 0 with several threads (reorderPixelSetJumpNextStrip, Denoiser.cpp:381-414: the
    even strips of 2b lines, then the odd ones): the marking fixed point follows it, per scale with that scale's geometry"""
    import bcd_amd.hip as bh
    col, ns, hist, cov, _ = inputs(W, H, 16, 0.15)
    prm = bh.default_params(m=1.0, random_order=2, seed=5)
    got = hipctx.denoise(*dev(col, ns, hist, cov), S, prm).cpu().numpy()
    orders, w_, h_ = [], W, H
    for _s in range(S):
        orders.append(bh.visit_order(w_, h_, 1, 2, bh.strip_order_seed(w_, h_, 1, 6)))
        w_, h_ = w_ // 2, h_ // 2
    assert not np.array_equal(orders[0], bh.visit_order(W, H, 1, 0, 0))      # (not the scanline order)
    want = ol.denoise_multiscale(col, ns, hist, cov, S, ol.params(m=1.0), orders=orders) if S > 1 else ol.denoise_mono(col, ns, hist, cov, ol.params(m=1.0), order=orders[0])
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), ok)
    assert rel_linf(np.where(ok, got, 0), np.where(ok, want, 0)) < TOL
    md = bh.MultiDenoiser([0, 0])
    try:
        with pytest.raises(bh.BcdHipError, match="strip"):                    # (row bands: refused, not silently reordered)
            md.denoise_host(col, ns, hist, cov, S, prm)
    finally:
        md.close()


@pytest.mark.parametrize("m,random_order,W,H", [(1.0, 1, 96, 64), (0.0, 0, 61, 45), (1.0, 0, 97, 65)])
def test_multiscale_parity(hipctx, m, random_order, W, H):
    import bcd_amd.hip as bh
    col, ns, hist, cov, _ = inputs(W, H, 16, 0.15)
    prm = bh.default_params(m=m, random_order=random_order, seed=11)
    d = dev(col, ns, hist, cov)
    got = hipctx.denoise(*d, 3, prm).cpu().numpy()
    orders = _orders(W, H, 1, random_order, 11, 3) if m != 0.0 else None
    want = ol.denoise_multiscale(col, ns, hist, cov, 3, ol.params(m=m), orders=orders)
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), ok)
    assert rel_linf(np.where(ok, got, 0), np.where(ok, want, 0)) < TOL


def test_pyramid_and_merge_bitexact(hipctx):
    W, H = 97, 65  # odd sizes pin the clamping
    col, ns, hist, cov, _ = inputs(W, H, 8)
    o = ol.oracle_ops()
    d_col, d_ns, d_hist, d_cov = dev(col, ns, hist, cov)
    assert bits_equal(hipctx.downscale_sum(d_hist).cpu().numpy(), o["dsum"](hist))
    assert bits_equal(hipctx.downscale_sum(d_ns).cpu().numpy(), o["dsum"](ns))
    assert bits_equal(hipctx.downscale_avg(d_col).cpu().numpy(), o["davg"](col))
    assert bits_equal(hipctx.downscale_avg(d_hist).cpu().numpy(), o["davg"](hist))  # (depth 60: the four-bins-per-thread kernel, average mode)
    assert bits_equal(hipctx.downscale_cov(d_cov, d_ns).cpu().numpy(), o["dcov"](cov, ns))
    lo = o["davg"](col)
    (d_lo,) = dev(lo)
    assert bits_equal(hipctx.interpolate(d_lo, H, W).cpu().numpy(), o["interp"](lo, H, W))
    assert bits_equal(hipctx.merge(d_col, d_lo).cpu().numpy(), o["merge"](col, lo))


def test_spike_filter_bitexact(hipctx):
    W, H = 75, 41
    col, ns, hist, cov, _ = inputs(W, H, 8, 0.35, 0.05)
    want = ol.oracle_ops()["spike"](col, ns, hist, cov, 2.0)
    got = hipctx.spike_filter(*dev(col, ns, hist, cov), 2.0)
    assert (want[0] != col).any()
    for g, w_ in zip(got, want):
        assert bits_equal(g.cpu().numpy(), w_)


def test_host_entry_point_and_errors(hipctx):
    import ctypes as C
    import bcd_amd.hip as bh
    W, H = 40, 30
    col, ns, hist, cov, _ = inputs(W, H, 8, 0.15)
    prm = bh.default_params(m=0.0)
    got = hipctx.denoise_host(col, ns, hist, cov, 1, prm)
    want = ol.denoise_mono(col, ns, hist, cov, ol.params(m=0.0))
    assert rel_linf(got, want) < TOL
    # null / empty inputs are refused with a status code, like Denoiser::inputsOutputsAreOk returning false
    rc = bh.lib().bcd_hip_denoise(hipctx.h, None, None, None, None, W, H, 60, 1, C.byref(prm), None)
    assert rc == -1
    d = dev(col, ns, hist, cov)
    import torch
    out = torch.empty((H, W, 3), device="cuda")
    rc = bh.lib().bcd_hip_denoise(hipctx.h, bh._dp(d[0]), bh._dp(d[1]), bh._dp(d[2]), bh._dp(d[3]), 0, H, 60, 1, C.byref(prm), bh._dp(out))
    assert rc == -1


@pytest.mark.parametrize("W,H,S,world,m", [(96, 80, 3, 2, 0.0), (70, 66, 2, 3, 0.0), (96, 80, 3, 2, 1.0)])
def test_band_path_virtual_ranks(hipctx, W, H, S, world, m):
    """the multi-GPU row-band path (per-band pyramid, accumulator/output halo exchange, edge merges) run as virtual
    ranks on one GPU: -m 0 must reproduce the single-GPU frame; -m 1 marks per band (a different, valid order)"""
    import torch
    import bcd_amd.hip as bh
    from bcd_amd.tiling import BandGeometry, HipEngine, run_virtual
    col, ns, hist, cov, _ = inputs(W, H, 16, 0.15)
    prm = bh.default_params(m=m, random_order=1, seed=9, b=3 if S == 3 else 6)
    full = hipctx.denoise(*dev(col, ns, hist, cov), S, prm).cpu().numpy()
    g = BandGeometry(W, H, S, prm.search_radius, 1, world)
    ins = []
    for r in range(world):
        l0, l1 = g.input_lines(r)
        ins.append(dev(col[l0:l1], ns[l0:l1], hist[l0:l1], cov[l0:l1]))
    outs = run_virtual(HipEngine(hipctx, reuse_buffers=False), g, ins, prm, prm.order_seed)
    torch.cuda.synchronize()
    got = np.concatenate([o.cpu().numpy() for o in outs], 0)
    assert got.shape == full.shape
    if m == 0.0:
        assert rel_linf(got, full) < 1e-5
    else:
        assert np.isfinite(got).all()
        assert np.sqrt(np.mean((got - full) ** 2)) < 0.05 * np.sqrt(np.mean(full ** 2))


# ---- committed fixtures: reference-generated (ref_*) and oracle regression (core_*) ---------------------------
import os as _os
_G = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")


def test_hip_kernels_match_reference_fixtures(hipctx):
    """pyramid / interpolate / merge / spike kernels against outputs of the REFERENCE's own compiled units"""
    f = np.load(_os.path.join(_G, "ref_pyramid.npz"))
    H, W, _ = f["mean"].shape
    d_ns, d_mean, d_cov, d_hist = dev(f["ns"], f["mean"], f["cov"], f["hist"])
    assert bits_equal(hipctx.downscale_sum(d_hist).cpu().numpy(), f["dsum_hist"])
    assert bits_equal(hipctx.downscale_sum(d_ns).cpu().numpy(), f["dsum_ns"])
    assert bits_equal(hipctx.downscale_avg(d_mean).cpu().numpy(), f["davg_mean"])
    assert bits_equal(hipctx.downscale_cov(d_cov, d_ns).cpu().numpy(), f["dcov"])
    (d_lo,) = dev(f["davg_mean"])
    d_up = hipctx.interpolate(d_lo, H, W)
    assert bits_equal(d_up.cpu().numpy(), f["interp"])
    assert bits_equal(hipctx.merge(d_mean, hipctx.downscale_avg(d_up)).cpu().numpy(), f["merge"])
    s = np.load(_os.path.join(_G, "ref_spike.npz"))
    got = hipctx.spike_filter(*dev(s["mean"], s["ns"], s["hist"], s["cov"]), float(s["factor"]))
    for g, k in zip(got, ("o_mean", "o_ns", "o_hist", "o_cov")):
        assert bits_equal(g.cpu().numpy(), s[k]), k


def test_low_sample_count_nan_semantics_on_gpu(hipctx):
    import bcd_amd.hip as bh
    f = np.load(_os.path.join(_G, "core_lowspp.npz"))
    d_col, d_ns, d_hist, d_cov = dev(f["col"], f["ns"], f["hist"], f["cov"])
    for (l, c), want in zip([(1, 1), (8, 12), (8, 20)], f["dist"]):
        assert bits_equal(hipctx.window_distances(d_hist, d_ns, 1, 6, l, c), want)
    mask, cnt = hipctx.similarity_masks(d_hist, d_ns, 1, 6, 1.0)
    assert np.array_equal(mask.cpu().numpy().view(np.uint32), f["mask"]) and np.array_equal(cnt.cpu().numpy(), f["cnt"])
    out = hipctx.denoise(d_col, d_ns, d_hist, d_cov, 1, bh.default_params(m=1.0, random_order=0))
    got = out.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(f["out_m1"]))
    z = hipctx.zero_bad_values(out).cpu().numpy()
    assert np.isfinite(z).all() and (z >= 0).all()


def test_core_regression_fixture_on_gpu(hipctx):
    import bcd_amd.hip as bh
    f = np.load(_os.path.join(_G, "core_regression.npz"))
    d_col, d_ns, d_hist, d_cov = dev(f["col"], f["ns"], f["hist"], f["cov"])
    mask, cnt = hipctx.similarity_masks(d_hist, d_ns, 1, 6, 1.0)
    assert np.array_equal(mask.cpu().numpy().view(np.uint32), f["mask"]) and np.array_equal(cnt.cpu().numpy(), f["cnt"])
    state, _ = hipctx.active_set(mask, cnt, 1, 6, 1.0, 0, 0)
    assert np.array_equal(state.cpu().numpy() == 1, f["processed"] == 1)
    out = hipctx.denoise(d_col, d_ns, d_hist, d_cov, 1, bh.default_params(m=1.0, random_order=0)).cpu().numpy()
    assert rel_linf(out, f["out_m1"]) < TOL
    out0 = hipctx.denoise(d_col, d_ns, d_hist, d_cov, 1, bh.default_params(m=0.0)).cpu().numpy()
    assert rel_linf(out0, f["out_m0"]) < TOL


def test_large_min_eigenvalue_takes_the_spectral_path(hipctx):
    """-e large: eigenvalues below the floor exist, the guarded sweep inverse must decline and the Jacobi
    spectral inverse reproduce the reference's V diag(1/max(e, lambda)) V^T"""
    import bcd_amd.hip as bh
    W, H = 64, 48
    col, ns, hist, cov, _ = inputs(W, H, 32, 0.08, 0.0)
    for e in (1e-3, 3e-2):
        got = hipctx.denoise(*dev(col, ns, hist, cov), 1, bh.default_params(m=0.0, min_eig=e)).cpu().numpy()
        want = ol.denoise_mono(col, ns, hist, cov, ol.params(m=0.0, min_eig=e))
        assert rel_linf(got, want) < TOL
        # the register-resident finish kernel hands such items to the LDS kernel (k_bayes27.hip: redo list): the path this test is about
        st = hipctx.stats(0)
        assert 0 < st.spectral_inverses <= st.processed - st.fallback
    # ... and with the reference's default floor no item of this frame needs it
    hipctx.denoise(*dev(col, ns, hist, cov), 1, bh.default_params(m=0.0))
    assert hipctx.stats(0).spectral_inverses == 0


# ---- BASELINE.json sizes: size-independent properties ----------------------------------------------------------
def _shift(t, dl, dc, fill):
    """t[l + dl, c + dc] at (l, c), `fill` outside"""
    import torch
    H, W = t.shape
    out = torch.full_like(t, fill)
    l0, l1 = max(0, -dl), min(H, H - dl)
    c0, c1 = max(0, -dc), min(W, W - dc)
    out[l0:l1, c0:c1] = t[l0 + dl:l1 + dl, c0 + dc:c1 + dc]
    return out


@pytest.mark.parametrize("random_order", [1, 0])
def test_720p_mask_symmetry_and_greedy_validity(hipctx, random_order):
    """at the benchmark size: (a) similarity masks are symmetric (d is bitwise symmetric), counts are popcounts;
    (b) the processed set IS the sequential greedy of the reference for the given order: a pixel is skipped iff an
    earlier-visited processed pixel with >= 28 similar patches contains it"""
    import torch
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H, b = 1280, 720, 6
    col, ns, hist, cov = core.synthetic_scene(W, H, 8, 77, 0.2, 0.01)
    d_hist, d_ns = dev(hist, ns)
    mask, cnt = hipctx.similarity_masks(d_hist, d_ns, 1, b, 1.0)
    state, rounds = hipctx.active_set(mask, cnt, 1, b, 1.0, random_order, 4242)
    side = 2 * b + 1
    order = torch.from_numpy(bh.visit_order(W, H, 1, random_order, 4242).astype(np.int64)).cuda()
    rank = torch.full((H * W,), 1 << 40, dtype=torch.int64, device="cuda")
    rank[order] = torch.arange(order.numel(), device="cuda")
    rank = rank.view(H, W)
    proc = state == 1
    strong_in = proc & (cnt >= 28)
    marked = torch.zeros((H, W), dtype=torch.bool, device="cuda")
    pop = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    m = mask.view(H, W, -1)
    for k in range(side * side):
        dl, dc = k // side - b, k % side - b
        bit = ((m[:, :, k // 32] >> (k % 32)) & 1).bool()
        pop += bit.int()
        kk = (b - dl) * side + (b - dc)
        rbit = ((m[:, :, kk // 32] >> (kk % 32)) & 1).bool()
        assert torch.equal(bit, _shift(rbit, dl, dc, False)), (dl, dc)         # (a) symmetry
        q_strong_in = _shift(strong_in, dl, dc, False)
        q_rank = _shift(rank, dl, dc, 1 << 41)
        marked |= bit & q_strong_in & (q_rank < rank)
    assert torch.equal(pop, cnt)
    main = state != 0
    assert int(main.sum()) == (W - 2) * (H - 2)
    assert torch.equal(proc, main & ~marked)                                     # (b) greedy validity
    assert rounds >= 1


def test_1080p_m1_default_run_is_sane(hipctx):
    """BASELINE config 2 shape (1920x1080, 3-scale, defaults): finite output, error vs the noise-free base reduced,
    and identical results for two runs with the same seed (the marking fixed point has a unique solution)"""
    import torch
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H = 1920, 1080
    col, ns, hist, cov = core.synthetic_scene(W, H, 8, 1, 0.15, 0.0)
    d = dev(col, ns, hist, cov)
    prm = bh.default_params(seed=99)
    hipctx.denoise(*d, 3, prm)
    st = [hipctx.stats(s) for s in range(3)]
    a = hipctx.denoise(*d, 3, prm).cpu().numpy()
    st2 = [hipctx.stats(s) for s in range(3)]
    assert np.isfinite(a).all()
    assert [(s.processed, s.fallback, s.similar_total) for s in st] == [(s.processed, s.fallback, s.similar_total) for s in st2]
    l, c = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    base = np.stack([0.2 + 0.6 * c / W, 0.5 + 0.4 * np.sin(12.0 * l / H), np.where(((l // 16 + c // 16) % 2) > 0, 0.8, 0.15)], -1)
    rmse = lambda x: float(np.sqrt(np.mean((x - base) ** 2)))
    assert rmse(a) < 0.6 * rmse(col)


def test_bcd_cli_end_to_end(hipctx, tmp_path):
    """the reference's front-end contract: three EXR inputs (hist file carries nSamples as last channel), flags, half RGBA
    output with negative/NaN/Inf zeroed; result == the engine's in-memory result rounded to half"""
    import subprocess
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H = 72, 56
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 21, 0.15, 0.02)
    stem = str(tmp_path / "frame")
    core.write_exr(stem + ".exr", col, False)
    core.write_exr(stem + "_hist.exr", core.merge_hist_ns(hist, ns), True)
    core.write_exr(stem + "_cov.exr", cov, True)
    exe = _os.path.join(_os.path.dirname(core.LIB_PATH), "bcd_cli")
    # colours go through half precision on disk, like the reference's own pipeline (raw2bcd writes half RGBA)
    col_h = core.read_exr(stem + ".exr", False)
    for p_flag in (0, 1):
        out_path = str(tmp_path / ("out%d.exr" % p_flag))
        r = subprocess.run([exe, "-i", stem + ".exr", "-o", out_path, "-p", str(p_flag), "-s", "2", "-b", "4", "-m", "0", "--seed", "5"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "assuming '" + stem + "_hist.exr'" in r.stdout            # -h / -c inferred from -i (main.cpp:344-370)
        got = core.read_exr(out_path, False)
        d = dev(col_h, ns, hist, cov)
        if p_flag:
            d = hipctx.spike_filter(*d, 2.0)
        want = hipctx.denoise(*d, 2, bh.default_params(b=4, m=0.0, seed=5))
        want = hipctx.zero_bad_values(want).cpu().numpy()
        assert np.max(np.abs(got - want.astype(np.float16).astype(np.float32))) <= 2e-3 * np.max(want)


def test_scale_free_division_is_ieee_exact(hipctx):
    """the 8-operation division of k_pairdist == hipcc's correctly rounded a / b on 2e9 operand pairs of its guarded range"""
    assert hipctx.selftest_division(2_000_000_000, seed=7) == 0
    assert hipctx.selftest_division(500_000_000, seed=12345) == 0


def test_out_of_range_inputs_take_the_exact_fallback(hipctx):
    """histogram bins / sample counts outside the guarded range switch the workgroup to the compiler's division: still bit-exact"""
    W, H = 70, 21
    col, ns, hist, cov, _ = inputs(W, H, 4, 0.3)
    hist = hist.copy(); ns = ns.copy()
    hist[5, 7, :] *= 4.0e6      # bins > 2^20
    ns[5, 7] *= 4.0e6
    ns[12, 40] = 3.0e-4         # n < 2^-10
    hist[12, 40, :] *= 1e-4
    d_hist, d_ns = dev(hist, ns)
    mask, cnt = hipctx.similarity_masks(d_hist, d_ns, 1, 6, 1.0)
    wmask, wcnt = ol.similarity_masks(ns, hist, 1, 6, 1.0)
    assert np.array_equal(mask.cpu().numpy().view(np.uint32), wmask) and np.array_equal(cnt.cpu().numpy(), wcnt)
    for (l, c) in [(5, 7), (6, 8), (12, 40), (11, 41)]:
        assert bits_equal(hipctx.window_distances(d_hist, d_ns, 1, 6, l, c), ol.window_distances(ns, hist, 1, 6, l, c))


@pytest.mark.parametrize("m,nscales", [(0.5, 1), (0.25, 2), (0.9, 1)])
def test_fractional_skip_probability(hipctx, m, nscales):
    """0 < -m < 1: marked pixels are skipped with probability m.  The reference draws unseeded rand(); the build uses a
    per-pixel hash, mirrored by the oracle, so the whole marking logic (non-skippable pixels still mark others) is checked"""
    import bcd_amd.hip as bh
    W, H = 72, 52
    col, ns, hist, cov, _ = inputs(W, H, 32, 0.08, 0.0)
    prm = bh.default_params(m=m, random_order=1, seed=21)
    got = hipctx.denoise(*dev(col, ns, hist, cov), nscales, prm).cpu().numpy()
    orders = _orders(W, H, 1, 1, 21, nscales)
    op = ol.params(m=m, skip_seed=21)
    want = ol.denoise_multiscale(col, ns, hist, cov, nscales, op, orders=orders) if nscales > 1 else ol.denoise_mono(col, ns, hist, cov, op, order=orders[0])
    assert rel_linf(got, want) < TOL
    st = hipctx.stats(0)
    full = ol.denoise_mono(col, ns, hist, cov, ol.params(m=1.0), order=orders[0], want_diag=True)[1][0].sum()
    assert st.processed > full          # fewer pixels are skipped than with -m 1


def test_mono_parity_large_search_window(hipctx):
    """-b 12 (BASELINE config 5): 625-bit masks, 313 displacement planes"""
    import bcd_amd.hip as bh
    W, H = 70, 44
    col, ns, hist, cov, _ = inputs(W, H, 16, 0.12, 0.0)
    prm = bh.default_params(m=1.0, random_order=1, seed=2, b=12)
    got = hipctx.denoise(*dev(col, ns, hist, cov), 1, prm).cpu().numpy()
    want = ol.denoise_mono(col, ns, hist, cov, ol.params(m=1.0, b=12), order=_orders(W, H, 1, 1, 2, 1)[0])
    assert rel_linf(got, want) < TOL
    got0 = hipctx.denoise(*dev(col, ns, hist, cov), 1, bh.default_params(m=0.0, b=12)).cpu().numpy()
    assert rel_linf(got0, ol.denoise_mono(col, ns, hist, cov, ol.params(m=0.0, b=12))) < TOL


def test_concurrent_and_serial_scales_agree(hipctx):
    """the per-scale streams/threads of a multiscale run change scheduling only"""
    import bcd_amd.hip as bh
    W, H = 120, 88
    col, ns, hist, cov, _ = inputs(W, H, 16, 0.12, 0.0)
    d = dev(col, ns, hist, cov)
    prm = bh.default_params(seed=4)
    a = hipctx.denoise(*d, 3, prm).cpu().numpy()
    sa = [(hipctx.stats(s).processed, hipctx.stats(s).fallback) for s in range(3)]
    hipctx.set_concurrent_scales(False)
    try:
        b_ = hipctx.denoise(*d, 3, prm).cpu().numpy()
        sb = [(hipctx.stats(s).processed, hipctx.stats(s).fallback) for s in range(3)]
    finally:
        hipctx.set_concurrent_scales(True)
    assert sa == sb
    assert rel_linf(a, b_) < 1e-5


def test_bcd_cli_preset_file(hipctx, tmp_path):
    """-a <file.bcd.json>: inputs and parameters come from the preset, later flags override it"""
    import json
    import subprocess
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H = 64, 48
    col, ns, hist, cov = core.synthetic_scene(W, H, 8, 3, 0.2, 0.0)
    (tmp_path / "in").mkdir()
    core.write_exr(str(tmp_path / "in" / "f.exr"), col, False)
    core.write_exr(str(tmp_path / "in" / "h.exr"), core.merge_hist_ns(hist, ns), True)
    core.write_exr(str(tmp_path / "in" / "c.exr"), cov, True)
    (tmp_path / "p.bcd.json").write_text(json.dumps({"inputColorFile": "in/f.exr", "inputHistoFile": "in/h.exr", "inputCovarFile": "in/c.exr",
                                                       "nbOfScales": 1, "searchWindowRadius": 3, "markedPixelsSkippingProbability": 0.0,
                                                       "performSpikeRemovalPrefiltering": False, "histoDistanceThreshold": 1.25}))
    exe = _os.path.join(_os.path.dirname(core.LIB_PATH), "bcd_cli")
    out_path = str(tmp_path / "o.exr")
    r = subprocess.run([exe, "-a", str(tmp_path / "p.bcd.json"), "-o", out_path, "-b", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = core.read_exr(out_path, False)
    col_h = core.read_exr(str(tmp_path / "in" / "f.exr"), False)
    want = hipctx.denoise(*dev(col_h, ns, hist, cov), 1, bh.default_params(b=4, m=0.0, tau=1.25))
    want = hipctx.zero_bad_values(want).cpu().numpy()
    assert np.max(np.abs(got - want.astype(np.float16).astype(np.float32))) <= 2e-3 * np.max(want)


@pytest.mark.parametrize("weighted,nbins", [(False, 20), (True, 20), (True, 12)])
def test_gpu_samples_accumulator(hipctx, weighted, nbins):
    """device SamplesAccumulator: nSamples / mean / covariance bit-identical to the host class (same order and operations),
    histograms to float round-off (device powf)"""
    import torch
    W, H, spp = 37, 22, 9
    samples, _ = ol.synth_samples(W, H, spp, seed=4, sigma=0.5, spike_prob=0.1)
    if weighted:
        samples[::3, 5] = 0.25
        samples[::7, 5] = 3.0
    want = ol.oracle_ops()["accumulate"](samples, W, H, nbins)
    s4 = samples.reshape(H, W, spp, 6)
    d_s = torch.from_numpy(np.ascontiguousarray(s4[..., 2:5])).cuda()
    d_w = torch.from_numpy(np.ascontiguousarray(s4[..., 5])).cuda() if weighted else None
    ns, mean, cov, hist = [t.cpu().numpy() for t in hipctx.accumulate_samples(d_s, d_w, nbins)]
    assert bits_equal(ns, want[0]) and bits_equal(mean, want[1]) and bits_equal(cov, want[2])
    assert np.max(np.abs(hist - want[3])) < 2e-5 * max(1.0, float(np.max(want[3])))
    assert np.allclose(hist.sum(-1), 3 * ns[..., 0], rtol=1e-5)


@pytest.mark.parametrize("W,H,S,world,random_order,b", [(96, 80, 3, 2, 1, 3), (80, 90, 2, 3, 1, 6), (96, 64, 1, 2, 0, 6)])
def test_band_path_exact_marking_equals_single_gpu(hipctx, W, H, S, world, random_order, b):
    """-m 1 over bands with exact_marking: global keys + boundary state exchange => the single-GPU frame"""
    import torch
    import bcd_amd.hip as bh
    from bcd_amd.tiling import BandGeometry, HipEngine, run_virtual
    col, ns, hist, cov, _ = inputs(W, H, 32, 0.08, 0.0)
    prm = bh.default_params(m=1.0, random_order=random_order, seed=9, b=b)
    full = hipctx.denoise(*dev(col, ns, hist, cov), S, prm).cpu().numpy()
    stats_full = [(hipctx.stats(s).processed, hipctx.stats(s).fallback) for s in range(S)]
    g = BandGeometry(W, H, S, prm.search_radius, 1, world)
    ins = []
    for r in range(world):
        l0, l1 = g.input_lines(r)
        ins.append(dev(col[l0:l1], ns[l0:l1], hist[l0:l1], cov[l0:l1]))
    outs = run_virtual(HipEngine(hipctx, reuse_buffers=False), g, ins, prm, prm.order_seed, exact_marking=True)
    torch.cuda.synchronize()
    got = np.concatenate([o.cpu().numpy() for o in outs], 0)
    assert stats_full[0][0] > 0
    assert rel_linf(got, full) < 1e-5


def _inputs_bins(W, H, spp, nbins, sigma=0.2, seed=5):
    samples, _ = ol.synth_samples(W, H, spp, seed=seed, sigma=sigma, spike_prob=0.0)
    ns, mean, cov, hist = ol.oracle_ops()["accumulate"](samples, W, H, nbins)
    return mean, ns, hist, cov


@pytest.mark.parametrize("nbins", [10, 8, 4, 40, 12])
def test_other_histogram_depths(hipctx, nbins):
    """D = 3 x bins other than 60: templated kernels for D in {12, 24, 36, 120}, the generic pair-distance kernel otherwise"""
    import bcd_amd.hip as bh
    W, H = 70, 40
    col, ns, hist, cov = _inputs_bins(W, H, 16, nbins)
    d_col, d_ns, d_hist, d_cov = dev(col, ns, hist, cov)
    mask, cnt = hipctx.similarity_masks(d_hist, d_ns, 1, 6, 1.0)
    wmask, wcnt = ol.similarity_masks(ns, hist, 1, 6, 1.0)
    assert np.array_equal(mask.cpu().numpy().view(np.uint32), wmask) and np.array_equal(cnt.cpu().numpy(), wcnt)
    got = hipctx.denoise(d_col, d_ns, d_hist, d_cov, 1, bh.default_params(m=0.0)).cpu().numpy()
    assert rel_linf(got, ol.denoise_mono(col, ns, hist, cov, ol.params(m=0.0))) < TOL


@pytest.mark.parametrize("nbins", [12, 8])
def test_ratio_form_other_histogram_depths_and_mixed_counts(hipctx, nbins):
    """the RATIO form of the distance kernel (general sample counts) has instantiations for D = 36 and 24 too: mixed sample counts at those depths go
    through it (masks and counts against the oracle), and its planes agree with the exact ones"""
    W, H = 90, 50
    rng = np.random.default_rng(nbins)
    samples, _ = ol.synth_samples(W, H, 16, seed=7, sigma=0.3, spike_prob=0.01)
    keep = rng.random(samples.shape[0]) < 0.7
    keep[::16] = True
    ns, mean, cov, hist = ol.oracle_ops()["accumulate"](np.ascontiguousarray(samples[keep]), W, H, nbins)
    assert hist.shape[-1] == 3 * nbins and len(np.unique(ns)) > 3
    d_hist, d_ns = dev(hist, ns)
    rel, count_mismatches, flags = hipctx.selftest_approx_distance(d_hist, d_ns, 6)
    assert count_mismatches == 0 and flags == 3 and rel < 2.0 ** -10 / 1.9     # 3: the RATIO form ran, no flag above it
    mask, cnt = hipctx.similarity_masks(d_hist, d_ns, 1, 6, 1.0)
    wmask, wcnt = ol.similarity_masks(ns, hist, 1, 6, 1.0)
    assert np.array_equal(mask.cpu().numpy().view(np.uint32), wmask) and np.array_equal(cnt.cpu().numpy(), wcnt)


@pytest.mark.parametrize("w,b,m", [(2, 3, 0.0), (2, 3, 1.0), (0, 4, 1.0), (2, 6, 1.0)])
def test_other_patch_radii(hipctx, w, b, m):
    """-w != 1 runs the generic mask / marking / Bayes kernels (K = 3(2w+1)^2 = 75 or 3)"""
    import bcd_amd.hip as bh
    W, H = 60, 44
    col, ns, hist, cov, _ = inputs(W, H, 32, 0.08, 0.0)
    prm = bh.default_params(m=m, random_order=1, seed=6, w=w, b=b)
    got = hipctx.denoise(*dev(col, ns, hist, cov), 1, prm).cpu().numpy()
    order = bh.visit_order(W, H, w, 1, 6) if m else None
    want = ol.denoise_mono(col, ns, hist, cov, ol.params(m=m, w=w, b=b), order=order)
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), ok)
    assert rel_linf(np.where(ok, got, 0), np.where(ok, want, 0)) < TOL


def test_unsupported_geometry_is_refused(hipctx):
    import bcd_amd.hip as bh
    col, ns, hist, cov, _ = inputs(40, 30, 8, 0.2, 0.0)
    with pytest.raises(bh.BcdHipError, match="LDS|supported"):
        hipctx.denoise(*dev(col, ns, hist, cov), 1, bh.default_params(w=3, b=6))      # three 148 x 149 matrices do not fit the LDS
    with pytest.raises(bh.BcdHipError):
        hipctx.denoise(*dev(col, ns, hist, cov), 6, bh.default_params())               # too many scales for 40 x 30



@pytest.mark.gpu
@pytest.mark.parametrize("up,down", [(True, True), (True, False), (False, True), (False, False)])
def test_finalize_band_equals_halo_adds_then_finalize(hipctx, up, down):
    """the fused tail of the band path: one launch == add the received halos, then finalise (bit for bit)"""
    import torch
    rows, W, halo = 19, 37, 7
    g = torch.Generator(device="cpu").manual_seed(3)
    s = torch.rand((rows, W, 3), generator=g).cuda()
    c = torch.randint(0, 5, (rows, W), generator=g, dtype=torch.int32).cuda()
    us, ds = torch.rand((halo, W, 3), generator=g).cuda(), torch.rand((halo, W, 3), generator=g).cuda()
    uc, dc = (torch.randint(0, 3, (halo, W), generator=g, dtype=torch.int32).cuda() for _ in range(2))
    s2, c2 = s.clone(), c.clone()
    if up:
        s2[:halo] += us; c2[:halo] += uc
    if down:
        s2[rows - halo:] += ds; c2[rows - halo:] += dc
    want = hipctx.finalize(s2, c2).cpu().numpy()
    out = torch.empty_like(s)
    hipctx.finalize_band(s, c, halo, (us, uc) if up else None, (ds, dc) if down else None, out)
    assert bits_equal(out.cpu().numpy(), want)     # count 0 -> inf / nan like the reference, same bits


@pytest.mark.gpu
def test_720p_three_scale_frame_against_the_oracle(hipctx):
    """BASELINE config[1] at its full size: 1280 x 720, 3 scales, b = 6, w = 1, with -m 0 (order-free: the run the 1e-4 bar is
    defined on); the oracle runs the reference's loop on the host cores of the box (about 20 s on 64 threads)"""
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H = 1280, 720
    col, ns, hist, cov = core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01)
    got = hipctx.denoise(*dev(col, ns, hist, cov), 3, bh.default_params(m=0.0)).cpu().numpy()
    threads = min(64, _os.cpu_count() or 1)
    want = ol.denoise_multiscale(col, ns, hist, cov, 3, ol.params(m=0.0, threads=threads))
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), ok)
    assert rel_linf(np.where(ok, got, 0), np.where(ok, want, 0)) < TOL_FULL_SIZE


@pytest.mark.gpu
def test_1080p_headline_frame_against_the_oracle(hipctx):
    """BASELINE config[2] at its FULL size on the bench's own frame (1920 x 1080, 32 spp, sigma 0.35, spikes, 3 scales, b = 6, w = 1): -m 0
    (order-free: every main pixel estimated, 1.6 M of them through the full Bayesian estimate -- the run the 1e-4 bar is defined on) against
    the oracle on the host cores of the box (about half a minute on 128 threads)"""
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H = 1920, 1080
    col, ns, hist, cov = core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01)
    got = hipctx.denoise(*dev(col, ns, hist, cov), 3, bh.default_params(m=0.0)).cpu().numpy()
    st = hipctx.stats(0)
    assert st.processed == st.main_pixels and st.processed - st.fallback > 1000000 and st.spectral_inverses == 0
    threads = min(128, _os.cpu_count() or 1)
    want = ol.denoise_multiscale(col, ns, hist, cov, 3, ol.params(m=0.0, threads=threads))
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), ok)
    assert rel_linf(np.where(ok, got, 0), np.where(ok, want, 0)) < TOL_FULL_SIZE


@pytest.mark.gpu
def test_1080p_bench_workload_against_the_oracle(hipctx):
    """the bench's headline workload ITSELF -- 1920 x 1080, 3 scales, b = 6, -m 1 -r 1 with the bench's seed, on the bench's frame -- against the oracle
    visiting the pixels in the same explicit orders (its three-phase ordered visit on the host cores of the box: the decisions of the visit stay
    sequential): marking decisions, fallback and full estimates, pyramid and merges of the measured configuration in one comparison"""
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H, S = 1920, 1080, 3
    col, ns, hist, cov = core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01)
    prm = bh.default_params(m=1.0, random_order=1, seed=1234)
    got = hipctx.denoise(*dev(col, ns, hist, cov), S, prm).cpu().numpy()
    st = hipctx.stats(0)
    assert (st.processed, st.fallback, st.similar_total) == (513339, 481684, 7370779)   # (what bench.py prints for this frame)
    threads = min(128, _os.cpu_count() or 1)
    want = ol.denoise_multiscale(col, ns, hist, cov, S, ol.params(m=1.0, skip_seed=1234, threads=threads), orders=_orders(W, H, 1, 1, 1234, S))
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), ok)
    assert rel_linf(np.where(ok, got, 0), np.where(ok, want, 0)) < TOL_FULL_SIZE


def _nonuniform_full_size_check(hipctx, W, H, frame, seed, expect_paths):
    """a frame whose sample counts are not one power of two, at a timed leg's size, 3 scales, -m 1 -r 1, against the oracle's ordered visit: the RATIO
    form of the distance kernel serves the scales (similarity_path 2; a scale whose a-posteriori error check fails would take the reference's operations,
    path 1 -- which scales do is a property of the frame and pinned per test), and the frame is the oracle's"""
    import bcd_amd.hip as bh
    S = 3
    col, ns, hist, cov = frame
    prm = bh.default_params(m=1.0, random_order=1, seed=seed)
    fresh = bh.Context(0)                                            # (a workspace with no memory of declined sizes: the decline path itself runs)
    try:
        got = fresh.denoise(*dev(col, ns, hist, cov), S, prm).cpu().numpy()
        paths = [fresh.stats(s).similarity_path for s in range(S)]
        full = fresh.stats(0).processed - fresh.stats(0).fallback
    finally:
        fresh.close()
    assert paths == expect_paths, paths
    assert full > 1000                                               # the full Bayesian estimate is exercised on the finest scale
    threads = min(128, _os.cpu_count() or 1)
    want = ol.denoise_multiscale(col, ns, hist, cov, S, ol.params(m=1.0, skip_seed=seed, threads=threads), orders=_orders(W, H, 1, 1, seed, S))
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), ok)
    assert rel_linf(np.where(ok, got, 0), np.where(ok, want, 0)) < TOL_FULL_SIZE
    return paths


@pytest.mark.gpu
def test_1080p_24spp_frame_against_the_oracle(hipctx):
    """bench.py's leg `nonuniform_uniform_24spp` at its full size (the general-sample-count path became production and was only checked at
    96 x 72): 1920 x 1080, a uniform 24 samples per pixel -- not a power of two, so the count products do not drop out"""
    import bcd_amd.core as core
    W, H = 1920, 1080
    _nonuniform_full_size_check(hipctx, W, H, core.synthetic_scene(W, H, 24, 1234, 0.35, 0.01), 1234, [2, 2, 2])


@pytest.mark.gpu
def test_1080p_mixed_sample_counts_frame_against_the_oracle(hipctx):
    """bench.py's leg `nonuniform_mixed_16_24_32_48` at its full size: per-pixel counts drawn from {16, 24, 32, 48} (48-spp statistics thinned per pixel, as
    bench.py builds them)"""
    import bcd_amd.core as core
    W, H = 1920, 1080
    col48, ns48, hist48, cov48 = core.synthetic_scene(W, H, 48, 1234, 0.35, 0.01)
    keep = np.random.default_rng(5).choice(np.array([1.0 / 3.0, 0.5, 2.0 / 3.0, 1.0], np.float32), size=(H, W, 1)).astype(np.float32)
    ns_mix = np.ascontiguousarray(np.rint(ns48 * keep).astype(np.float32))
    hist_mix = np.ascontiguousarray(hist48 * (ns_mix / ns48))
    assert sorted(np.unique(ns_mix)) == [16.0, 24.0, 32.0, 48.0]
    _nonuniform_full_size_check(hipctx, W, H, (col48, ns_mix, hist_mix, cov48), 1234, [2, 2, 2])


@pytest.mark.gpu
@pytest.mark.skipif((_os.cpu_count() or 1) < 64 and not _os.environ.get("BCD_TEST_SLOW"), reason="the 4K oracle run needs >= 64 host cores (80 s on 128 threads of the GPU box); BCD_TEST_SLOW=1 forces it")
def test_4k_config3_frame_against_the_oracle(hipctx):
    """BASELINE configs[3]'s frame at its FULL size (3840 x 2160, 3 scales, b = 6, -m 1 -r 1) on one GPU against the oracle's ordered visit on the
    host cores (opt-in: the oracle needs about two minutes); the 8-row-band form of the same frame is compared with this single-GPU result by
    test_4k_config3_properties_and_eight_row_bands"""
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H, S = 3840, 2160, 3
    col, ns, hist, cov = core.synthetic_scene(W, H, 8, 3, 0.15, 0.0)
    prm = bh.default_params(seed=17)
    got = hipctx.denoise(*dev(col, ns, hist, cov), S, prm).cpu().numpy()
    threads = min(128, _os.cpu_count() or 1)
    want = ol.denoise_multiscale(col, ns, hist, cov, S, ol.params(m=1.0, skip_seed=17, threads=threads), orders=_orders(W, H, 1, 1, 17, S))
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), ok)
    assert rel_linf(np.where(ok, got, 0), np.where(ok, want, 0)) < TOL_FULL_SIZE


@pytest.mark.gpu
@pytest.mark.skipif((_os.cpu_count() or 1) < 64 and not _os.environ.get("BCD_TEST_SLOW"), reason="the 4K b = 12 oracle run needs >= 64 host cores (220 s on 128 threads of the GPU box); BCD_TEST_SLOW=1 forces it")
def test_4k_config4_frame_against_the_oracle(hipctx):
    """BASELINE configs[4]'s chain at its FULL size (3840 x 2160, b = 12, -p 1 --p-factor 2, -r 1 -m 1, 3 scales) on one GPU against the oracle
    (its spike filter -- pinned to the reference's compiled unit -- then its ordered visit on the host cores; opt-in)"""
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H, S, b = 3840, 2160, 3, 12
    col, ns, hist, cov = core.synthetic_scene(W, H, 8, 3, 0.25, 0.01)
    prm = bh.default_params(seed=23, b=b, random_order=1)
    got = hipctx.denoise_host(col, ns, hist, cov, S, prm, spike_factor=2.0)
    fc, fn, fh, fv = ol.oracle_ops()["spike"](col, ns, hist, cov, 2.0)
    threads = min(128, _os.cpu_count() or 1)
    want = ol.denoise_multiscale(fc, fn, fh, fv, S, ol.params(b=b, m=1.0, skip_seed=23, threads=threads), orders=_orders(W, H, 1, 1, 23, S))
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), ok)
    assert rel_linf(np.where(ok, got, 0), np.where(ok, want, 0)) < TOL_FULL_SIZE


@pytest.mark.gpu
def test_quarter_hd_three_scale_marking_run_against_the_oracle(hipctx):
    """the bench workload (noisy frame, 3 scales, b = 6, -m 1 -r 1) at 480 x 270 against the oracle visiting the pixels in the SAME
    explicit order (one thread, ~15 s): marking decisions, fallback and full estimates, pyramid and merges in one comparison"""
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H = 480, 270
    col, ns, hist, cov = core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01)
    prm = bh.default_params(m=1.0, random_order=1, seed=1234)
    got = hipctx.denoise(*dev(col, ns, hist, cov), 3, prm).cpu().numpy()
    frac = [hipctx.stats(s).processed / max(1, hipctx.stats(s).main_pixels) for s in range(3)]
    orders = _orders(W, H, 1, 1, 1234, 3)
    want = ol.denoise_multiscale(col, ns, hist, cov, 3, ol.params(m=1.0), orders=orders)
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), ok)
    assert rel_linf(np.where(ok, got, 0), np.where(ok, want, 0)) < TOL
    assert 0.05 < frac[0] < 0.8   # marking did skip pixels on the finest scale


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["uniform_pow2", "uniform_12", "mixed", "one_pixel_differs"])
def test_sample_count_paths_of_the_distance_kernel_bitexact(hipctx, kind):
    """k_pairdist drops the sample-count products when every pixel has the same power-of-two count (exact identity); any other
    input -- a uniform count that is no power of two, mixed counts, a single odd pixel -- takes the general formula.  Both bit-exact."""
    W, H, b = 70, 37, 6
    rng = np.random.default_rng(11)
    spp = {"uniform_pow2": 16, "uniform_12": 12}.get(kind, 16)
    samples, _ = ol.synth_samples(W, H, spp, seed=5, sigma=0.3, spike_prob=0.01)
    if kind == "mixed":            # drop a random subset of the samples: 8..16 samples per pixel
        keep = rng.random(samples.shape[0]) < 0.75
        keep[::spp] = True
        samples = np.ascontiguousarray(samples[keep])
    elif kind == "one_pixel_differs":
        samples = np.ascontiguousarray(samples[1:])      # pixel (0, 0) has one sample less
    ns, mean, cov, hist = ol.oracle_ops()["accumulate"](samples, W, H)
    assert (len(np.unique(ns)) == 1) == kind.startswith("uniform")
    d_hist, d_ns = dev(hist, ns)
    mask, cnt = hipctx.similarity_masks(d_hist, d_ns, 1, b, 1.0)
    wmask, wcnt = ol.similarity_masks(ns, hist, 1, b, 1.0)
    assert np.array_equal(mask.cpu().numpy().view(np.uint32), wmask)
    assert np.array_equal(cnt.cpu().numpy(), wcnt)
    for (l, c) in [(1, 1), (H // 2, W // 2), (2, 3)]:
        assert bits_equal(hipctx.window_distances(d_hist, d_ns, 1, b, l, c), ol.window_distances(ns, hist, 1, b, l, c))
    # every entry of the 85 T / C planes: production variant == compiler division + general formula, bit for bit
    variant, mismatches = hipctx.selftest_distance_kernels(d_hist, d_ns, b)
    assert variant == (2 if kind == "uniform_pow2" else 1) and mismatches == 0


@pytest.mark.gpu
@pytest.mark.parametrize("sigma,spikes", [(0.35, 0.01), (0.08, 0.0)])
def test_distance_planes_of_a_benchmark_frame_bitexact(hipctx, sigma, spikes):
    """all 85 T / C planes of a 640 x 360 frame of the benchmark generator (32 spp: uniform power-of-two counts, wave-uniform bin
    skipping at work): production kernel == compiler division + general formula, bit for bit"""
    import bcd_amd.core as core
    col, ns, hist, cov = core.synthetic_scene(640, 360, 32, 1234, sigma, spikes)
    variant, mismatches = hipctx.selftest_distance_kernels(*dev(hist, ns), 6)
    assert variant == 2 and mismatches == 0


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,b", [(1024, 400, 6), (1000, 402, 3)])
def test_similarity_masks_bitexact_on_a_large_scale(hipctx, W, H, b):
    """scales of >= 400 k pixels take the four-columns-per-lane forward-mask kernel: masks and counts against the oracle (host threads).  Second case:
    a height that is no multiple of the kernel's four-line strips, a width that is no multiple of its 248 columns, and a radius whose 25 forward planes
    are one short word with an odd count (the kernel evaluates two planes per trip and drops the extra bit; other radii than 6 / 12 also take the
    generic symmetric-mask kernel on the word-plane layout)"""
    import bcd_amd.core as core
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 99, 0.3, 0.01)
    # the plane entries of pairs that leave the image are never written: leave fp32 planes of another frame in the workspace first (read as
    # binary16 they are negative numbers and NaNs), so that a kernel that lets them reach a border pixel's decision fails here (round 6)
    hipctx.set_fast_similarity(0)
    try:
        _, ns2, hist2, _ = core.synthetic_scene(1100, 420, 16, 7, 0.5, 0.02)
        hipctx.similarity_masks(*dev(hist2, ns2), 1, b, 1.0)
    finally:
        hipctx.set_fast_similarity(1)
    mask, cnt = hipctx.similarity_masks(*dev(hist, ns), 1, b, 1.0)
    wmask, wcnt = ol.similarity_masks(ns, hist, 1, b, 1.0, threads=min(64, _os.cpu_count() or 1))
    assert np.array_equal(mask.cpu().numpy().view(np.uint32), wmask)
    assert np.array_equal(cnt.cpu().numpy(), wcnt)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["bench_noisy", "bench_clean", "mixed_counts", "b12", "ragged"])
def test_fast_similarity_path_equals_exact_kernels(hipctx, kind):
    """approximate distance planes + exact re-evaluation of the pairs within tau (1 +- 2^-10) (k_similarity_fast.hip) give the
    masks of the exact kernels bit for bit; the approximate patch distances stay far inside that band; bin counts are exact"""
    import bcd_amd.core as core
    b, tau = 6, 1.0
    if kind == "bench_noisy":
        col, ns, hist, cov = core.synthetic_scene(640, 360, 32, 1234, 0.35, 0.01)
    elif kind == "bench_clean":
        col, ns, hist, cov = core.synthetic_scene(640, 360, 32, 1234, 0.10, 0.0)
    elif kind == "mixed_counts":
        rng = np.random.default_rng(3)
        samples, _ = ol.synth_samples(200, 64, 16, seed=5, sigma=0.3, spike_prob=0.01)
        keep = rng.random(samples.shape[0]) < 0.75
        keep[::16] = True
        ns, mean, cov, hist = ol.oracle_ops()["accumulate"](np.ascontiguousarray(samples[keep]), 200, 64)
    elif kind == "b12":
        col, ns, hist, cov = core.synthetic_scene(200, 90, 16, 7, 0.2, 0.01)
        b, tau = 12, 0.9
    else:
        col, ns, hist, cov = core.synthetic_scene(131, 37, 8, 11, 0.3, 0.01)
    d_hist, d_ns = dev(hist, ns)
    try:
        hipctx.set_fast_similarity(True)
        m1, c1 = hipctx.similarity_masks(d_hist, d_ns, 1, b, tau)
        hipctx.set_fast_similarity(False)
        m0, c0 = hipctx.similarity_masks(d_hist, d_ns, 1, b, tau)
    finally:
        hipctx.set_fast_similarity(True)
    assert np.array_equal(m1.cpu().numpy(), m0.cpu().numpy())
    assert np.array_equal(c1.cpu().numpy(), c0.cpu().numpy())
    rel, count_mismatches, flags = hipctx.selftest_approx_distance(d_hist, d_ns, b)
    assert count_mismatches == 0 and (flags >> 4) == 0
    # the approximate T plane is stored in binary16: worst case 5.0e-4 (2^-11 per entry + the fp32 round-off), measured ~2.4e-4;
    # the exactly re-evaluated band is +-2^-10
    assert rel < 2.0 ** -10 / 1.9, rel


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["uniform_24", "uniform_100", "mixed_counts", "mixed_wide_b12", "ragged", "one_tile"])
def test_ratio_form_of_the_distance_kernel_counts_exact_and_distances_inside_the_band(hipctx, kind):
    """k_pairdist_rw<D, false, false, RATIO> (production for every frame whose sample counts are not one power of two: similarity_path 2): one fma per bin
    with rho = n1 / n2, the count ratio applied once per pair -- bin counts identical to the exact planes', patch distances inside the verified band, its
    absolute-error verdict clear on these inputs; uniform counts that are no power of two (rho == 1: the uniform kernel's arithmetic), counts mixed 1 : 3,
    the large search window, ragged sizes"""
    import bcd_amd.core as core
    b = 6
    if kind == "uniform_24":
        col, ns, hist, cov = core.synthetic_scene(320, 200, 24, 1234, 0.35, 0.01)
    elif kind == "uniform_100":
        col, ns, hist, cov = core.synthetic_scene(200, 120, 100, 5, 0.25, 0.0)
    elif kind in ("mixed_counts", "mixed_wide_b12"):
        rng = np.random.default_rng(3)
        W, H = (200, 64) if kind == "mixed_counts" else (120, 70)
        col, ns, hist, cov = core.synthetic_scene(W, H, 48, 5, 0.3, 0.01)
        keep = rng.choice(np.array([1.0 / 3.0, 0.5, 2.0 / 3.0, 1.0], np.float32), size=(H, W, 1)).astype(np.float32)
        ns2 = np.ascontiguousarray(np.rint(ns * keep).astype(np.float32))
        hist, ns = np.ascontiguousarray(hist * (ns2 / ns)), ns2
        b = 6 if kind == "mixed_counts" else 12
    elif kind == "ragged":
        col, ns, hist, cov = core.synthetic_scene(131, 37, 12, 11, 0.3, 0.01)
    else:
        col, ns, hist, cov = core.synthetic_scene(19, 9, 12, 2, 0.3, 0.01)
    d_hist, d_ns = dev(hist, ns)
    rel, count_mismatches, flags = hipctx.selftest_approx_distance(d_hist, d_ns, b)
    assert count_mismatches == 0 and flags == 3, (count_mismatches, flags)
    assert rel < 2.0 ** -10 / 1.9, rel
    mask, cnt = hipctx.similarity_masks(d_hist, d_ns, 1, b, 1.0)
    assert hipctx.stats(0).similarity_path in (0, 1, 2)                  # (stage call: the path is reported by the frame tests)
    wmask, wcnt = ol.similarity_masks(ns, hist, 1, b, 1.0)
    assert np.array_equal(mask.cpu().numpy().view(np.uint32), wmask) and np.array_equal(cnt.cpu().numpy(), wcnt)


@pytest.mark.gpu
@pytest.mark.parametrize("tau", [2.0 ** -6, 0.01, 64.0, 100.0])
def test_similarity_thresholds_at_and_beyond_the_range_of_the_approximate_planes(hipctx, tau):
    """the binary16 planes of the fast path serve thresholds in [2^-6, 64] only (subnormals below, +inf above); outside, the exact
    kernels run -- either way the masks are the exact path's masks"""
    import bcd_amd.core as core
    col, ns, hist, cov = core.synthetic_scene(96, 48, 16, 3, 0.05 if tau < 0.1 else 0.6, 0.02)
    d_hist, d_ns = dev(hist, ns)
    try:
        hipctx.set_fast_similarity(True)
        m1, c1 = hipctx.similarity_masks(d_hist, d_ns, 1, 6, tau)
        hipctx.set_fast_similarity(False)
        m0, c0 = hipctx.similarity_masks(d_hist, d_ns, 1, 6, tau)
    finally:
        hipctx.set_fast_similarity(True)
    assert np.array_equal(m1.cpu().numpy(), m0.cpu().numpy())
    assert np.array_equal(c1.cpu().numpy(), c0.cpu().numpy())
    assert 0 < int(c1.sum()) < c1.numel() * 169   # a threshold that actually separates pairs on this frame


@pytest.mark.gpu
def test_fast_similarity_path_is_the_default_and_reports_its_borderline_pairs(hipctx):
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    col, ns, hist, cov = core.synthetic_scene(320, 200, 32, 1234, 0.35, 0.01)
    prm = bh.default_params(b=6, w=1, m=1.0, random_order=1, seed=1234)
    fresh = bh.Context(0)  # (a workspace that remembers frames with mixed sample counts would start with the general formula: path 2)
    out = fresh.denoise(*dev(col, ns, hist, cov), 1, prm)
    st = fresh.stats(0)
    assert st.similarity_path == 1 and 0 <= st.borderline_pairs < 320 * 200
    assert np.isfinite(out.cpu().numpy()[1:-1, 1:-1]).all()


def _drop_samples(W, H, spp, seed, keep_frac, sigma=0.3):
    """a frame whose pixels carry different sample counts (adaptive sampling): a random subset of a uniform sample stream"""
    rng = np.random.default_rng(seed)
    samples, _ = ol.synth_samples(W, H, spp, seed=seed, sigma=sigma, spike_prob=0.01)
    keep = rng.random(samples.shape[0]) < keep_frac
    keep[::spp] = True
    ns, mean, cov, hist = ol.oracle_ops()["accumulate"](np.ascontiguousarray(samples[keep]), W, H)
    return mean, ns, hist, cov


@pytest.mark.gpu
@pytest.mark.parametrize("m,S", [(0.0, 1), (1.0, 2)])
def test_frames_with_mixed_sample_counts_take_the_ratio_form_and_match_the_oracle(hipctx, m, S):
    """general sample counts (src/core/DenoisingUnit.cpp:371-383 takes any n1, n2): the production path is the RATIO form of the
    distance kernel (similarity_path 2); the frame is the oracle's"""
    import bcd_amd.hip as bh
    W, H = 96, 72
    col, ns, hist, cov = _drop_samples(W, H, 16, 9, 0.7)
    assert len(np.unique(ns)) > 3
    prm = bh.default_params(m=m, random_order=0, seed=5)
    got = hipctx.denoise(*dev(col, ns, hist, cov), S, prm).cpu().numpy()
    assert hipctx.stats(0).similarity_path == 2
    if S == 1:
        want = ol.denoise_mono(col, ns, hist, cov, ol.params(m=m), order=_orders(W, H, 1, 0, 5, 1)[0] if m != 0.0 else None)
    else:
        want = ol.denoise_multiscale(col, ns, hist, cov, S, ol.params(m=m), orders=_orders(W, H, 1, 0, 5, S) if m != 0.0 else None)
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), ok)
    assert rel_linf(np.where(ok, got, 0), np.where(ok, want, 0)) < TOL


@pytest.mark.gpu
def test_ratio_form_declines_when_its_error_bound_fails_and_the_reference_operations_take_over(hipctx):
    """counts spread 1 : 64 over a frame with hundreds of samples per pixel: the RATIO form's a-posteriori check (10 u kappa G <= 2^-12 tau) fails
    (flag value 4), the pass is repeated with the reference's operations, and the masks are the exact kernels' either way"""
    W, H = 64, 40
    col, ns, hist, cov = _drop_samples(W, H, 16, 4, 0.8)
    scale = np.where(np.random.default_rng(1).random((H, W, 1)) < 0.5, 640.0, 10.0).astype(np.float32)   # 40 ... 10 000 samples per pixel
    hist, ns = np.ascontiguousarray(hist * scale), np.ascontiguousarray(ns * scale)
    d_hist, d_ns = dev(hist, ns)
    rel, count_mismatches, flags = hipctx.selftest_approx_distance(d_hist, d_ns, 6)
    assert count_mismatches == 0 and (flags & 15) == 3 and ((flags >> 4) & 4) != 0, flags
    import bcd_amd.hip as bh
    fresh = bh.Context(0)                       # (a workspace that has not met such a frame yet)
    try:
        m1, c1 = fresh.similarity_masks(d_hist, d_ns, 1, 6, 1.0)
        wmask, wcnt = ol.similarity_masks(ns, hist, 1, 6, 1.0)
        assert np.array_equal(m1.cpu().numpy().view(np.uint32), wmask) and np.array_equal(c1.cpu().numpy(), wcnt)
        m2, c2 = fresh.similarity_masks(d_hist, d_ns, 1, 6, 1.0)      # second call: the workspace goes to the reference's operations directly
        assert np.array_equal(m2.cpu().numpy().view(np.uint32), wmask)
    finally:
        fresh.close()


@pytest.mark.gpu
def test_low_sample_frame_stays_inside_its_budget_and_strict_eigensolver_tightens_it(hipctx):
    """an ill-conditioned low-sample frame at reduced size (8 spp, -m 0: every pixel takes the full estimate): the production stopping rule of the
    eigensolver (2e-9 + first-order correction) stays within 2e-5 of the oracle, the strict rule (bcd_hip_set_strict_eigensolver) is not worse"""
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H = 160, 120
    col, ns, hist, cov = core.synthetic_scene(W, H, 8, 77, 0.15, 0.0)
    prm = bh.default_params(m=0.0)
    want = ol.denoise_mono(col, ns, hist, cov, ol.params(m=0.0))
    ok = np.isfinite(want)
    d = dev(col, ns, hist, cov)
    got = hipctx.denoise(*d, 1, prm).cpu().numpy()
    e_prod = rel_linf(np.where(ok, got, 0), np.where(ok, want, 0))
    try:
        bh.set_strict_eigensolver(True)
        got_s = hipctx.denoise(*d, 1, prm).cpu().numpy()
    finally:
        bh.set_strict_eigensolver(False)
    e_strict = rel_linf(np.where(ok, got_s, 0), np.where(ok, want, 0))
    assert e_prod < 2e-5 and e_strict < 2e-5 and e_strict <= e_prod * 1.5 + 1e-7, (e_prod, e_strict)


@pytest.mark.gpu
def test_batched_eigensolver_against_lapack(hipctx):
    """bcd_hip_eig27_batch (the Jacobi solver of the Bayesian steps on its own, two matrices per wavefront): V diag(lambda) V^T
    reconstructs the input, V is orthogonal, the eigenvalues are LAPACK's -- for an odd number of matrices (half-empty last pair),
    an already diagonal matrix, a rank-one matrix and matrices with equal diagonal elements"""
    import torch
    rng = np.random.default_rng(4)
    n = 257
    A = np.zeros((n, 28, 28), np.float32)
    for i in range(n):
        X = rng.standard_normal((int(rng.integers(28, 120)), 27)) * (0.02 + rng.random(27))
        N = np.diag(rng.random(27) * 0.3)
        A[i, :27, :27] = (np.cov(X.T) - N).astype(np.float32)
    A[3, :27, :27] = np.diag(np.linspace(-1, 2, 27)).astype(np.float32)
    v = rng.standard_normal(27).astype(np.float32)
    A[5, :27, :27] = np.outer(v, v)
    # equal diagonal elements (theta = 0: 45-degree rotations, where the two lanes of a pair must still agree on the sign) and exact ties
    A[7, :27, :27] = 0.25 * np.eye(27) + 0.01 * np.ones((27, 27))
    A[9, :27, :27] = np.kron(np.eye(9), np.array([[1.0, 0.5, 0.5], [0.5, 1.0, 0.5], [0.5, 0.5, 1.0]]))
    A = (A + A.transpose(0, 2, 1)) / 2
    eig, V, _ = hipctx.eig27_batch(torch.from_numpy(A).cuda())
    eig, V = eig.cpu().numpy().astype(np.float64), V.cpu().numpy().astype(np.float64)
    for i in range(n):
        a = A[i, :27, :27].astype(np.float64)
        nrm = np.linalg.norm(a)
        Vi = V[i, :27, :27]
        assert np.linalg.norm((Vi * eig[i, :27]) @ Vi.T - a) / nrm < 1e-5, i
        assert np.linalg.norm(Vi.T @ Vi - np.eye(27)) < 1e-5, i
        assert np.max(np.abs(np.sort(eig[i, :27]) - np.linalg.eigvalsh(a))) / nrm < 1e-5, i


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["BCD_HIP_JACOBI_PAIRS", "BCD_HIP_FINISH_LDS"])
def test_comparison_kernels_stay_correct(switch):
    """the kernels kept for A/B comparisons behind environment switches (read once per process: run in a child) -- the round-per-LDS-trip
    eigensolver and the finish kernel through LDS matrices -- still reproduce the oracle"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, torch, oracle_lib as ol, bcd_amd.core as core, bcd_amd.hip as bh\n"
            "col, ns, hist, cov = core.synthetic_scene(72, 56, 32, 5, 0.08, 0.0)\n"
            "ctx = bh.Context(0)\n"
            "got = ctx.denoise(*[torch.from_numpy(a).cuda() for a in (col, ns, hist, cov)], 1, bh.default_params(m=0.0)).cpu().numpy()\n"
            "want = ol.denoise_mono(col, ns, hist, cov, ol.params(m=0.0))\n"
            "print('ERR %%.3e' %% (np.max(np.abs(got - want)) / np.max(np.abs(want))))\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                          os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env[switch] = "1"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    err = float([l for l in out.stdout.splitlines() if l.startswith("ERR")][-1].split()[1])
    assert err < TOL, err


@pytest.mark.gpu
@pytest.mark.parametrize("switch,value,expect_path", [("BCD_HIP_SERIAL_SCALES", "1", 1), ("BCD_HIP_EXACT_SIMILARITY", "1", 0), ("BCD_HIP_STRICT_EIGEN", "1", 1),
                                                      ("BCD_HIP_STREAM_UPLOADS", "0", 1), ("BCD_HIP_SPARSE_UPLOAD", "0", 1), ("BCD_HIP_UPLOAD_THREADS", "1", 1),
                                                      ("BCD_HIP_UPLOAD_SIMD", "scalar", 1)])
def test_every_environment_switch_of_the_engine_reproduces_the_oracle(switch, value, expect_path):
    """Every environment variable libbcd_hip.so reads at context creation (they are read once per process: run in a child) selects code that is
    checked here against the oracle, on a 3-scale host-buffer frame tall enough for the streamed upload: scales one after the other, the exact
    distance kernels, the fully converged eigensolver, whole-frame uploads, plain (unpacked) histogram uploads, one packing thread, the scalar
    packer.  (The comparison kernels have their own tests: BCD_HIP_JACOBI_PAIRS, BCD_HIP_FINISH_LDS, BCD_HIP_PREPARE_GATHER; the band driver's
    variables are set by the multi-rank tests and the RCCL canary: BCD_HIP_MULTI_ORDERED, _TIMEOUT_S, _ABORT_WAIT_MS, _VERBOSE.)"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, oracle_lib as ol, bcd_amd.core as core, bcd_amd.hip as bh\n"
            "W, H, S = 80, 272, 3\n"
            "col, ns, hist, cov = core.synthetic_scene(W, H, 16, 11, 0.12, 0.005)\n"
            "ctx = bh.Context(0)\n"
            "got = ctx.denoise_host(col, ns, hist, cov, S, bh.default_params(m=1.0, random_order=1, seed=4))\n"
            "orders = [bh.visit_order(W >> s, H >> s, 1, 1, bh.scale_seed(4, s)) for s in range(S)]\n"
            "want = ol.denoise_multiscale(col, ns, hist, cov, S, ol.params(m=1.0), orders=orders)\n"
            "raw, sent = ctx.last_upload_bytes()\n"
            "print('ERR %%.3e PATH %%d SENT %%d RAW %%d' %% (np.max(np.abs(got - want)) / np.max(np.abs(want)), ctx.stats(0).similarity_path, sent, raw))\n") % (
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env[switch] = value
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("ERR")][-1].split()
    assert float(line[1]) < TOL, line
    assert int(line[3]) == expect_path, line                                      # the exact kernels report similarity_path 0
    if switch in ("BCD_HIP_SPARSE_UPLOAD", "BCD_HIP_STREAM_UPLOADS"):
        assert int(line[5]) == int(line[7]) or switch == "BCD_HIP_STREAM_UPLOADS", line    # plain copies: every histogram byte travelled


@pytest.mark.gpu
@pytest.mark.parametrize("gather", ["0", "1"])
def test_large_window_prepare_kernels_agree_with_the_oracle(gather):
    """-b 12 (BASELINE configs[4]'s window): the prepare kernel with the colour window in LDS (round 5, default) and the gather kernel it replaced
    (BCD_HIP_PREPARE_GATHER=1, read once per process: run in a child) both reproduce the oracle, on a frame whose border pixels have windows
    that leave the image (the window kernel reads stale member slots as the centre pixel there) and with |S| up to the full 625"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, torch, oracle_lib as ol, bcd_amd.core as core, bcd_amd.hip as bh\n"
            "col, ns, hist, cov = core.synthetic_scene(60, 44, 32, 5, 0.08, 0.0)\n"
            "ctx = bh.Context(0)\n"
            "got = ctx.denoise(*[torch.from_numpy(a).cuda() for a in (col, ns, hist, cov)], 1, bh.default_params(m=0.0, b=12)).cpu().numpy()\n"
            "want = ol.denoise_mono(col, ns, hist, cov, ol.params(m=0.0, b=12))\n"
            "st = ctx.stats(0)\n"
            "print('ERR %%.3e FULL %%d' %% (np.max(np.abs(got - want)) / np.max(np.abs(want)), st.processed - st.fallback))\n") % (
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["BCD_HIP_PREPARE_GATHER"] = gather
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("ERR")][-1].split()
    assert float(line[1]) < TOL and int(line[3]) > 200, line     # (392 of the 2 436 processed pixels take the full estimate)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,S,ranks,m,random_order,b", [(256, 288, 3, 2, 1.0, 1, 6), (256, 288, 3, 4, 1.0, 1, 6), (256, 288, 3, 4, 0.0, 0, 6),
                                                          (96, 80, 3, 2, 1.0, 0, 3), (70, 66, 2, 3, 0.5, 1, 6), (64, 48, 1, 1, 1.0, 1, 6)])
def test_native_multi_rank_driver_equals_single_gpu(hipctx, W, H, S, ranks, m, random_order, b):
    """bcd_hip_multi_denoise_host (C++ driver of the row-band partition: per-band pyramid, frame-ordered marking with boundary
    state exchange, accumulator / output halo exchange, merges at the band edges) with several ranks on ONE device (in-process
    transport; distinct devices use RCCL) reproduces the single-GPU frame -- BASELINE configs[3]'s geometry (3 scales, b = 6) at a
    reduced size, for -m 1 (both orders), a fractional -m and -m 0"""
    import bcd_amd.hip as bh
    import bcd_amd.core as core
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 21, 0.12, 0.005)
    prm = bh.default_params(m=m, random_order=random_order, seed=9, b=b)
    want = hipctx.denoise_host(col, ns, hist, cov, S, prm)
    md = bh.MultiDenoiser([0] * ranks)
    try:
        got = md.denoise_host(col, ns, hist, cov, S, prm)
        st = md.stats()
        again = md.denoise_host(col, ns, hist, cov, S, prm)   # buffers / contexts are reused
    finally:
        md.close()
    assert st.n_ranks == ranks and st.transport == 0 and st.compute_ms > 0
    assert rel_linf(got, want) < 1e-5
    assert rel_linf(again, want) < 1e-5
    # ... and the ORACLE's frame: the band path is checked against the CPU restatement of the reference, not only against the
    # single-GPU HIP path (same explicit visiting order per scale; -m 0 is order-free)
    oracle_frame = _oracle_frame(("scene", W, H, 16, 21, 0.12, 0.005), (col, ns, hist, cov), S, b, m, random_order, 9)
    assert rel_linf(got, oracle_frame) < TOL


_oracle_frames = {}


def _oracle_frame(key, frame, S, b, m, random_order, seed, min_eig=1e-8):
    """oracle result of one configuration (cached: several rank counts share it)"""
    col, ns, hist, cov = frame
    H, W, _ = hist.shape
    k = (key, S, b, m, random_order, seed, min_eig)
    if k not in _oracle_frames:
        op = ol.params(b=b, m=m, min_eig=min_eig, skip_seed=seed)
        orders = _orders(W, H, 1, random_order, seed, S) if m != 0.0 else None
        _oracle_frames[k] = (ol.denoise_multiscale(col, ns, hist, cov, S, op, orders=orders) if S > 1 else
                             ol.denoise_mono(col, ns, hist, cov, op, order=None if orders is None else orders[0]))
    return _oracle_frames[k]


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,m", [(2, 1.0), (4, 1.0), (3, 0.0)])
def test_native_multi_rank_driver_ordered_communication(hipctx, monkeypatch, ranks, m):
    """the issue-order rule of the RCCL transport (CommGate in bcd_multi.hip: a scale starts communicating when the coarser
    scales of its rank are through) exercised on the in-process transport: same frame, no deadlock"""
    import bcd_amd.hip as bh
    import bcd_amd.core as core
    monkeypatch.setenv("BCD_HIP_MULTI_ORDERED", "1")
    W, H, S = 256, 288, 3
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 21, 0.12, 0.005)
    prm = bh.default_params(m=m, random_order=1, seed=9, b=6)
    want = hipctx.denoise_host(col, ns, hist, cov, S, prm)
    md = bh.MultiDenoiser([0] * ranks)
    try:
        got = md.denoise_host(col, ns, hist, cov, S, prm)
        again = md.denoise_host(col, ns, hist, cov, S, prm)
    finally:
        md.close()
    assert rel_linf(got, want) < 1e-5
    assert rel_linf(again, want) < 1e-5


def _check_phase_major_order(trace, S, W, halo):
    """the issue order of bcd_multi.hip's CommGate (round 6):  P1(S-1) .. P1(0)  R(S-1) .. R(0)  P2(S-1) .. P2(0)  merges.  P1 of a scale = the two
    operations of its first marking batch (|S| of the boundary lines, the all-reduced count: the initial states of the neighbours' boundary lines are
    computed locally), R = whatever its marking needed beyond them (further batches -- boundary states + all-reduce each; normally nothing), P2 = its accumulator exchange (sums + counts: halo x W x 16 bytes), merges = channel S"""
    chans = [ch for ch, _, _, _ in trace]
    acc = lambda i: trace[i][1] == 0 and max(trace[i][2], trace[i][3]) == halo * (W >> trace[i][0]) * 16   # sums (12 bytes per pixel) + counts (4) in one operation
    p2 = {c: [i for i in range(len(trace)) if chans[i] == c][-1:] for c in range(S)}            # the last operation of a scale's channel
    assert all(len(v) == 1 and acc(v[0]) for v in p2.values())
    mark = {c: [i for i in range(len(trace)) if chans[i] == c and i not in p2[c]] for c in range(S)}
    p1 = {c: v[:2] for c, v in mark.items()}
    rr = {c: v[2:] for c, v in mark.items()}
    for c in range(S):
        assert not p1[c] or [trace[i][1] for i in p1[c]] == [0, 1]                              # exchange, all-reduce
        assert [trace[i][1] for i in rr[c]] == [0, 1] * (len(rr[c]) // 2)                       # boundary states + all-reduce per further batch
        assert not rr[c] or trace[rr[c][-1]][1] == 1                                            # a marking always ends with an all-reduce
    flat = lambda d: [i for v in d.values() for i in v]
    for c in range(S - 1):
        assert not p1[c] or not p1[c + 1] or max(p1[c + 1]) < min(p1[c])                       # first batches: coarser scales first
        assert not rr[c] or not rr[c + 1] or max(rr[c + 1]) < min(rr[c])                       # further batches: coarser scales first
        assert max(p2[c + 1]) < min(p2[c])                                                      # accumulators: coarser scales first
    assert max(flat(p1) or [-1]) < min(flat(rr) or [len(trace)])                                # every first batch before any further batch
    assert max(flat(p1) + flat(rr) or [-1]) < min(flat(p2))                                     # every marking operation before any accumulator exchange
    merges = [i for i in range(len(trace)) if chans[i] == S]
    assert merges and min(merges) > max(flat(p2))                                                # the merges' exchanges come last


@pytest.mark.gpu
@pytest.mark.parametrize("ordered", [True, False])
@pytest.mark.parametrize("m", [1.0, 0.0])
def test_native_multi_rank_driver_communication_protocol(hipctx, monkeypatch, ordered, m):
    """what decides whether the RCCL transport can block, checked on one GPU from the driver's communication trace: every rank
    enqueues the same sequence of (channel, operation) -- globally with the issue-order rule of the RCCL transport (coarse scales
    first, the merges' exchanges last), per channel without it -- and neighbouring ranks agree on the size of every message"""
    import bcd_amd.hip as bh
    import bcd_amd.core as core
    monkeypatch.setenv("BCD_HIP_MULTI_ORDERED", "1" if ordered else "0")
    W, H, S, ranks = 256, 288, 3, 4
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 21, 0.12, 0.005)
    prm = bh.default_params(m=m, random_order=1, seed=9, b=6)
    md = bh.MultiDenoiser([0] * ranks)
    try:
        md.set_comm_trace(True)
        md.denoise_host(col, ns, hist, cov, S, prm)
        traces = [md.comm_trace(r) for r in range(ranks)]
    finally:
        md.close()
    assert all(len(t) == len(traces[0]) and len(t) > 0 for t in traces)
    per_channel = lambda t, c: [(k, up, dn) for ch, k, up, dn in t if ch == c]
    for c in range(S + 1):
        ops = [per_channel(t, c) for t in traces]
        assert all([k for k, _, _ in o] == [k for k, _, _ in ops[0]] for o in ops)          # same operations on the channel
        for i in range(len(ops[0])):
            assert ops[0][i][1] == 0 and ops[-1][i][2] == 0                                     # nothing beyond the outer ranks
            for r in range(ranks - 1):
                assert ops[r][i][2] == ops[r + 1][i][1] and (ops[r][i][0] == 1 or ops[r][i][2] > 0)   # message sizes agree
    if ordered:
        seq = [[(ch, k) for ch, k, _, _ in t] for t in traces]
        assert all(q == seq[0] for q in seq)                                                    # one global sequence
        _check_phase_major_order(traces[1], S, W, 6 + 1)                                        # (an interior rank: both neighbours)
    if m > 0:
        assert any(k == 1 for _, k, _, _ in traces[0])                                          # the marking all-reduce was there


@pytest.mark.gpu
@pytest.mark.parametrize("m", [1.0, 0.0])
def test_native_multi_rank_driver_recomputes_masks_when_one_band_leaves_the_guarded_range(hipctx, m):
    """one pixel of ONE band has histogram bins / a sample count outside the range the production distance kernels guard: that
    rank's verdict arrives with its first marking batch, the all-reduce tells every rank, the marking restarts on exact masks --
    and the frame is still the single-GPU frame (which takes the same fallback on its own)"""
    import bcd_amd.hip as bh
    import bcd_amd.core as core
    W, H, S, ranks = 128, 160, 2, 3
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 4, 0.2, 0.0)
    hist = hist.copy(); ns = ns.copy()
    hist[H - 20, 33, :] *= 4.0e6    # in the last band only
    ns[H - 20, 33] *= 4.0e6
    prm = bh.default_params(m=m, random_order=1, seed=3, b=6)
    want = hipctx.denoise_host(col, ns, hist, cov, S, prm)
    assert hipctx.stats(0).similarity_path != 1      # the single-GPU run itself fell back to the exact kernels on scale 0
    md = bh.MultiDenoiser([0] * ranks)
    try:
        md.set_comm_trace(True)
        got = md.denoise_host(col, ns, hist, cov, S, prm)
        traces = [md.comm_trace(r) for r in range(ranks)]
    finally:
        md.close()
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), ok)
    assert rel_linf(np.where(ok, got, 0), np.where(ok, want, 0)) < 1e-5
    for c in range(S + 1):   # every rank restarted alike: the same operations on every channel
        assert all([k for ch, k, _, _ in t if ch == c] == [k for ch, k, _, _ in traces[0] if ch == c] for t in traces)


@pytest.mark.gpu
def test_native_multi_rank_driver_large_window_prefilter_random_order(hipctx):
    """BASELINE configs[4] through the band path at a reduced size: b = 12, spike prefilter (-p 1) and random order (-r 1), 3 scales"""
    import bcd_amd.hip as bh
    import bcd_amd.core as core
    W, H, S = 160, 224, 3
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 5, 0.3, 0.02)
    col, ns, hist, cov = core.spike_filter(col, ns, hist, cov, 2.0)
    prm = bh.default_params(m=1.0, random_order=1, seed=77, b=12)
    want = hipctx.denoise_host(col, ns, hist, cov, S, prm)
    md = bh.MultiDenoiser([0, 0])
    try:
        got = md.denoise_host(col, ns, hist, cov, S, prm)
    finally:
        md.close()
    assert rel_linf(got, want) < 1e-5


@pytest.mark.gpu
def test_native_multi_rank_driver_reports_errors(hipctx):
    import bcd_amd.hip as bh
    import bcd_amd.core as core
    col, ns, hist, cov = core.synthetic_scene(64, 40, 8, 5, 0.3, 0.0)
    md = bh.MultiDenoiser([0] * 8)
    try:
        with pytest.raises(bh.BcdHipError, match="band"):
            md.denoise_host(col, ns, hist, cov, 3, bh.default_params())   # 40 lines cannot feed 8 bands of a 3-scale pyramid
    finally:
        md.close()
    with pytest.raises(bh.BcdHipError):
        bh.MultiDenoiser([0, 99])


@pytest.mark.gpu
def test_progress_callback_is_monotone_and_fires_inside_the_loop(hipctx):
    """IDenoiser::setProgressCallback (Denoiser.cpp:181-192 of the reference fires it inside the loop): here every scale reports
    twice, weighted by its pixels -- more than the two end values, never decreasing, ending at 1"""
    import bcd_amd.core as core
    col, ns, hist, cov = core.synthetic_scene(96, 80, 16, 3, 0.15, 0.0)
    seen = []
    hipctx.set_progress_callback(seen.append)
    try:
        import bcd_amd.hip as bh
        hipctx.denoise_host(col, ns, hist, cov, 3, bh.default_params(m=1.0))
    finally:
        hipctx.set_progress_callback(None)
    assert len(seen) == 6 and all(b >= a for a, b in zip(seen, seen[1:])) and 0 < seen[0] < 1 and abs(seen[-1] - 1.0) < 1e-6
    # through the C++ classes: 0, the engine's values, 1
    ok, out, monotone = core.denoise(col, ns, hist, cov, nscales=3, m=1.0)
    assert ok and monotone and core.lib().bcdcore_last_progress_values() > 4


@pytest.mark.gpu
def test_device_resident_prefilter_and_cleanup_chain(hipctx):
    """bcd_hip_denoise_host_ex: spike prefilter on the uploaded copies + denoise + bad-value clean-up == the three separate steps;
    the caller's images are not modified"""
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    col, ns, hist, cov = core.synthetic_scene(80, 60, 16, 21, 0.2, 0.03)
    keep = [a.copy() for a in (col, ns, hist, cov)]
    prm = bh.default_params(b=4, m=0.0, seed=5)
    got = hipctx.denoise_host(col, ns, hist, cov, 2, prm, spike_factor=2.0, zero_bad_values=True)
    assert all(np.array_equal(a, b) for a, b in zip(keep, (col, ns, hist, cov)))
    d = hipctx.spike_filter(*dev(col, ns, hist, cov), 2.0)
    want = hipctx.zero_bad_values(hipctx.denoise(*d, 2, prm)).cpu().numpy()
    assert rel_linf(got, want) < 1e-5   # (the aggregation's float atomics are order-dependent at ~1e-7)


@pytest.mark.gpu
def test_bcd_cli_devices_list(hipctx, tmp_path):
    """bcd_cli --devices a,b: the frame goes through the native multi-rank driver (two ranks on device 0 here) and must give the
    single-device file"""
    import subprocess
    import bcd_amd.core as core
    W, H = 96, 72
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 21, 0.15, 0.02)
    stem = str(tmp_path / "frame")
    core.write_exr(stem + ".exr", col, False)
    core.write_exr(stem + "_hist.exr", core.merge_hist_ns(hist, ns), True)
    core.write_exr(stem + "_cov.exr", cov, True)
    exe = _os.path.join(_os.path.dirname(core.LIB_PATH), "bcd_cli")
    outs = []
    for devs in ("0", "0,0"):
        out_path = str(tmp_path / ("out_%s.exr" % devs.replace(",", "_")))
        r = subprocess.run([exe, "-i", stem + ".exr", "-o", out_path, "-p", "1", "-s", "2", "-b", "4", "--seed", "5", "--devices", devs],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(core.read_exr(out_path, False))
    assert np.max(np.abs(outs[0] - outs[1])) <= 2e-3 * np.max(outs[0])   # both files are half precision
    r = subprocess.run([exe, "-i", stem + ".exr", "-o", str(tmp_path / "x.exr"), "--devices", "3-1"], capture_output=True, text=True)
    assert r.returncode == 1 and "--devices" in r.stdout


@pytest.mark.gpu
def test_one_process_per_gpu_rank_api_with_one_rank(hipctx):
    """bcd_hip_multi_rank_* (what bench.py --gpus N drives, one process per GPU): with a single rank the band is the frame; the
    resident-input / step / download cycle must give the single-GPU result, twice"""
    import bcd_amd.hip as bh
    import bcd_amd.core as core
    W, H, S = 128, 96, 3
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 21, 0.12, 0.005)
    prm = bh.default_params(m=1.0, random_order=1, seed=9)
    want = hipctx.denoise_host(col, ns, hist, cov, S, prm)
    rd = bh.RankDenoiser(0, 1, 0, None)
    try:
        l0, nl, o0, no = rd.configure(W, H, 60, S, prm)
        assert (l0, nl, o0, no) == (0, H, 0, H)
        rd.upload(col, ns, hist, cov)
        rd.step()
        a = rd.download()
        rd.step()
        b_ = rd.download()
    finally:
        rd.close()
    assert rel_linf(a, want) < 1e-5 and rel_linf(b_, want) < 1e-5
    ids = bh.multi_unique_ids(2)
    assert len(ids) == 2 * bh.MULTI_ID_BYTES and ids[:128] != ids[128:]


@pytest.mark.gpu
def test_cpu_request_is_declined_with_a_note_and_served_by_the_device(hipctx, capfd):
    """m_useCuda = false through the C++ API (bcd::Denoiser / MultiscaleDenoiser): a note on cout, the device result (identical to m_useCuda = true)"""
    import bcd_amd.core as core
    col, ns, hist, cov = core.synthetic_scene(48, 40, 8, 3, 0.2, 0.0)
    ok1, out1, _ = core.denoise(col, ns, hist, cov, 2, use_cuda=True, random_order=False)
    ok0, out0, _ = core.denoise(col, ns, hist, cov, 2, use_cuda=False, random_order=False)
    assert ok1 and ok0 and rel_linf(out0, out1) < 1e-6      # (two device runs: the accumulation order of the float atomics differs)
    assert "running on the HIP device" in capfd.readouterr().out


@pytest.mark.gpu
def test_bcd_cli_baseline_config0_plumbing(hipctx, tmp_path):
    """BASELINE configs[0] as far as this build goes: `bcd_cli -s 1 -b 6 -w 1 --ncores 1 -r 0` on a 128 x 96 scene (the reference
    bundles none: data/inputs holds a .gitignore only).  There is no CPU path: `--use-cuda 0` is declined with a note and served by the HIP device
    (under BCD_STRICT_CPU_REQUEST=1: refused, exit code 2, before any file is read); the RESULT is the CPU path's -- the oracle's single-thread
    scanline run on the same (half-precision) colours"""
    import subprocess
    import bcd_amd.core as core
    W, H = 128, 96
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 21, 0.12, 0.0)
    stem = str(tmp_path / "scene")
    core.write_exr(stem + ".exr", col, False)
    core.write_exr(stem + "_hist.exr", core.merge_hist_ns(hist, ns), True)
    core.write_exr(stem + "_cov.exr", cov, True)
    exe = _os.path.join(_os.path.dirname(core.LIB_PATH), "bcd_cli")
    out_path = str(tmp_path / "out.exr")
    flags = [exe, "-i", stem + ".exr", "-h", stem + "_hist.exr", "-c", stem + "_cov.exr", "-o", out_path, "-s", "1", "-b", "6", "-w", "1",
             "-r", "0", "-p", "0", "-m", "1", "--ncores", "1"]
    r = subprocess.run(flags + ["--use-cuda", "0"], capture_output=True, text=True, env=dict(_os.environ, BCD_STRICT_CPU_REQUEST="1"))
    assert r.returncode == 2 and "does not have" in r.stderr and not _os.path.exists(out_path)   # strict mode: the CPU request is refused, loudly
    r = subprocess.run(flags + ["--use-cuda", "0"], capture_output=True, text=True)
    assert r.returncode == 0 and "running on the HIP device" in r.stdout, r.stdout + r.stderr     # default: declined with a note, served by the device
    got_cpu_request = core.read_exr(out_path, False)
    r = subprocess.run(flags + ["--use-cuda", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = core.read_exr(out_path, False)
    assert np.max(np.abs(got - got_cpu_request)) <= 1e-3 * np.max(got)     # (two device runs, written as half)
    col_h = core.read_exr(stem + ".exr", False)                           # colours as the CLI saw them (half on disk)
    want = ol.denoise_mono(col_h, ns, hist, cov, ol.params(b=6, m=1.0, threads=1))   # 1 thread, -r 0: plain scanline order
    want = np.where(np.isfinite(want) & (want >= 0), want, 0.0).astype(np.float32)
    assert np.max(np.abs(got - want.astype(np.float16).astype(np.float32))) <= 2e-3 * np.max(want)
    # --ncores 4 -r 0: the reference reorders the pixel list strip-wise (Denoiser.cpp:375-414); so does the engine, and the marking follows it
    import bcd_amd.hip as bh
    r = subprocess.run(flags[:-1] + ["4", "--use-cuda", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got4 = core.read_exr(out_path, False)
    strips = bh.visit_order(W, H, 1, 2, bh.strip_order_seed(W, H, 1, 6))
    want4 = ol.denoise_mono(col_h, ns, hist, cov, ol.params(b=6, m=1.0, threads=1), order=strips)
    want4 = np.where(np.isfinite(want4) & (want4 >= 0), want4, 0.0).astype(np.float32)
    assert np.max(np.abs(got4 - want4.astype(np.float16).astype(np.float32))) <= 2e-3 * np.max(want4)
    assert np.max(np.abs(want4 - want)) > 1e-3 * np.max(want)             # (the two orders do give different images)


@pytest.mark.gpu
@pytest.mark.parametrize("nb_of_cores,expect_strips", [(0, True), (1, False), (4, True)])
def test_reused_multiscale_denoiser_visits_in_the_same_order_twice(hipctx, nb_of_cores, expect_strips):
    """ONE bcd::MultiscaleDenoiser, denoise() twice with -r 0: the core count written back after the first call (Denoiser.cpp:121) is the one
    that decided its visiting order (:375-380), so the second call gives the same frame (to the round-off of the accumulators' atomics) -- and m_nbOfCores = 0 means all cores (strips on a
    multi-core host, as in the reference), 1 the scanline order"""
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H, S = 96, 80, 3
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 21, 0.12, 0.0)
    ok, o1, o2, after = core.denoise_reuse(col, ns, hist, cov, nscales=S, b=6, nb_of_cores=nb_of_cores)
    assert ok and after[0] == after[1] and (after[0] == nb_of_cores or nb_of_cores == 0)
    assert rel_linf(o1, o2) < 2e-6                                   # (the accumulators' float atomics: ~1e-7; another ORDER is > 1e-4, below)
    if (_os.cpu_count() or 1) > 1:
        prm = bh.default_params(m=1.0, random_order=2 if expect_strips else 0, seed=bh.strip_order_seed(W, H, 1, 6) if expect_strips else 0)
        # (per-scale strip seeds are derived inside the engine; compare against the engine called with that order directly)
        other = bh.default_params(m=1.0, random_order=0 if expect_strips else 2)
        a = hipctx.denoise_host(col, ns, hist, cov, S, prm)
        b_ = hipctx.denoise_host(col, ns, hist, cov, S, other)
        assert rel_linf(o1, a) < 1e-6 and rel_linf(o1, b_) > 1e-4     # the order asked for, not the other one


# ---- BASELINE configs[3] and configs[4]: against the ORACLE at reduced size, by properties at full size ------------------------------
@pytest.mark.gpu
def test_config4_chain_single_gpu_against_the_oracle(hipctx):
    """BASELINE configs[4]'s chain on one GPU against the oracle: spike prefilter (-p 1 --p-factor 2; the ORACLE's filter, which is
    pinned to the reference's compiled unit) -> 3 scales, b = 12, random order (-r 1), -m 1 -- and the same with -m 0.  The HIP side
    runs its own prefilter kernel inside bcd_hip_denoise_host_ex (what bcd_cli -p 1 uses)"""
    import bcd_amd.hip as bh
    import bcd_amd.core as core
    W, H, S, b = 176, 232, 3, 12
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 5, 0.3, 0.02)
    fc, fn, fh, fv = ol.oracle_ops()["spike"](col, ns, hist, cov, 2.0)
    assert (fc != col).any()                                              # the filter does something on this frame
    for m in (1.0, 0.0):
        prm = bh.default_params(m=m, random_order=1, seed=77, b=b)
        got = hipctx.denoise_host(col, ns, hist, cov, S, prm, spike_factor=2.0)
        want = _oracle_frame(("config4", W, H), (fc, fn, fh, fv), S, b, m, 1, 77)
        ok = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), ok)
        assert rel_linf(np.where(ok, got, 0), np.where(ok, want, 0)) < TOL
    assert hipctx.stats(0).similarity_path == 1                          # b = 12 went through the production similarity kernels


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_config4_chain_row_bands_against_the_oracle(hipctx, ranks):
    """the same chain through the row-band driver (prefilter on host copies by the C++ class, bands with b + w = 13 halo lines)
    against the oracle"""
    import bcd_amd.hip as bh
    import bcd_amd.core as core
    W, H, S, b = 176, 232, 3, 12
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 5, 0.3, 0.02)
    fc, fn, fh, fv = ol.oracle_ops()["spike"](col, ns, hist, cov, 2.0)
    prm = bh.default_params(m=1.0, random_order=1, seed=77, b=b)
    md = bh.MultiDenoiser([0] * ranks)
    try:
        got = md.denoise_host(fc, fn, fh, fv, S, prm)
    finally:
        md.close()
    want = _oracle_frame(("config4", W, H), (fc, fn, fh, fv), S, b, 1.0, 1, 77)
    assert rel_linf(got, want) < TOL
    # through bcd::MultiscaleDenoiser::setDevices + setSpikePrefilter (what bcd_cli --devices a,b -p 1 does)
    ok, out, monotone = core.denoise(col, ns, hist, cov, nscales=S, b=b, m=1.0, seed=77, devices=[0] * ranks, prefilter_factor=2.0)
    assert ok and monotone and rel_linf(out, want) < TOL
    # m_nbOfCores = 0 in, the thread count the reference would have run with out (Denoiser.cpp:113-121: OpenMP's default) -- the number that
    # also decides the -r 0 visiting order, so that it reproduces itself on a reused object
    assert core.last_nb_of_cores() >= 1
    assert core.lib().bcdcore_last_progress_values() > 4                  # the multi-device path reports progress inside the loop


def _mask_properties(hipctx, hist, ns, b, random_order, seed):
    """(a) symmetric masks, counts = popcounts; (b) the processed set is the sequential greedy set of the visiting order"""
    import torch
    import bcd_amd.hip as bh
    H, W, _ = hist.shape
    mask, cnt = hipctx.similarity_masks(hist, ns, 1, b, 1.0)
    state, rounds = hipctx.active_set(mask, cnt, 1, b, 1.0, random_order, seed)
    side = 2 * b + 1
    order = torch.from_numpy(bh.visit_order(W, H, 1, random_order, seed).astype(np.int64)).cuda()
    rank = torch.full((H * W,), 1 << 40, dtype=torch.int64, device="cuda")
    rank[order] = torch.arange(order.numel(), device="cuda")
    rank = rank.view(H, W)
    proc = state == 1
    strong_in = proc & (cnt >= 28)
    marked = torch.zeros((H, W), dtype=torch.bool, device="cuda")
    pop = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    m = mask.view(H, W, -1)
    for k in range(side * side):
        dl, dc = k // side - b, k % side - b
        bit = ((m[:, :, k // 32] >> (k % 32)) & 1).bool()
        pop += bit.int()
        kk = (b - dl) * side + (b - dc)
        rbit = ((m[:, :, kk // 32] >> (kk % 32)) & 1).bool()
        assert torch.equal(bit, _shift(rbit, dl, dc, False)), (dl, dc)
        marked |= bit & _shift(strong_in, dl, dc, False) & (_shift(rank, dl, dc, 1 << 41) < rank)
    assert torch.equal(pop, cnt)
    main = state != 0
    assert int(main.sum()) == (W - 2) * (H - 2)
    assert torch.equal(proc, main & ~marked)
    return rounds


@pytest.mark.gpu
def test_4k_config3_properties_and_eight_row_bands(hipctx):
    """BASELINE configs[3] at its FULL size (3840 x 2160, 3 scales, b = 6): finite deterministic output that reduces the error, masks
    symmetric and the processed set the sequential greedy set at scale 0, and the frame cut into EIGHT row bands (the 8-GPU
    partition: 270 owned lines + 2 x 7 * 2^s halo lines per scale and rank, here as 8 ranks on one device) equal to the single-GPU
    frame to 1e-5"""
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H, S = 3840, 2160, 3
    col, ns, hist, cov = core.synthetic_scene(W, H, 8, 3, 0.15, 0.0)
    prm = bh.default_params(seed=17)
    d = dev(col, ns, hist, cov)
    a = hipctx.denoise(*d, S, prm).cpu().numpy()
    st = [(hipctx.stats(s).processed, hipctx.stats(s).fallback, hipctx.stats(s).similar_total) for s in range(S)]
    assert np.isfinite(a).all()
    assert np.array_equal(a, np.where(np.isfinite(a), a, 0)) and st[0][0] > 0 and st[0][0] > st[0][1] > 0
    l, c = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    base = np.stack([0.2 + 0.6 * c / W, 0.5 + 0.4 * np.sin(12.0 * l / H), np.where(((l // 16 + c // 16) % 2) > 0, 0.8, 0.15)], -1)
    rmse = lambda x: float(np.sqrt(np.mean((x - base) ** 2)))
    assert rmse(a) < 0.6 * rmse(col)
    assert _mask_properties(hipctx, d[2], d[1], 6, 1, bh.scale_seed(17, 0)) >= 1
    del d
    md = bh.MultiDenoiser([0] * 8)
    try:
        got = md.denoise_host(col, ns, hist, cov, S, prm)
    finally:
        md.close()
    assert rel_linf(got, a) < 1e-5


@pytest.mark.gpu
def test_4k_config4_large_window_prefilter_eight_row_bands(hipctx):
    """BASELINE configs[4] at its FULL size (3840 x 2160, b = 12, -p 1 --p-factor 2, -r 1, 3 scales): finite output that reduces the
    error, and eight row bands (13 * 2^s halo lines) equal to the single-GPU frame to 1e-5"""
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H, S, b = 3840, 2160, 3, 12
    col, ns, hist, cov = core.synthetic_scene(W, H, 8, 3, 0.25, 0.01)
    prm = bh.default_params(seed=23, b=b, random_order=1)
    a = hipctx.denoise_host(col, ns, hist, cov, S, prm, spike_factor=2.0)
    assert np.isfinite(a).all()
    l, c = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    base = np.stack([0.2 + 0.6 * c / W, 0.5 + 0.4 * np.sin(12.0 * l / H), np.where(((l // 16 + c // 16) % 2) > 0, 0.8, 0.15)], -1)
    rmse = lambda x: float(np.sqrt(np.mean((x - base) ** 2)))
    assert rmse(a) < 0.6 * rmse(col)
    ok, out, _ = core.denoise(col, ns, hist, cov, nscales=S, b=b, m=1.0, seed=23, devices=[0] * 8, prefilter_factor=2.0)
    assert ok and rel_linf(out, a) < 1e-5


def _run_band_bench(extra, nproc, port):
    """bench.py through its row-band branch in a child process; returns the parsed JSON line"""
    import json
    import subprocess
    import sys
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    env = dict(_os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    launcher = [sys.executable]
    if nproc > 1:
        launcher += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(port)]
    else:
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run(launcher + [_os.path.join(root, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1", "--watchdog", "300"] + extra,
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert out and r.stdout.rstrip().endswith(out[-1]), "the JSON line must be the last line of stdout"
    return json.loads(out[-1]), out[-1]


def _check_band_bench_line(line, raw, world, with_4k):
    """the keys a reader (and the driver) takes from the N > 1 line; ONE place, used by the one-GPU and the two-GPU test alike"""
    assert line["n_gpus"] == world and line["scaling"] == "strong" and line["metric"].startswith("Mpixels/sec")
    assert line["config"]["parallelism"] == "rowband%d-exactmark-native" % world
    assert line["band_check"]["rel_linf_vs_single_gpu"] < 1e-5
    assert line["value"] > 0 and abs(line["value"] - 1920 * 1080 / 1e3 / line["ms_per_step"]) < 0.01 * line["value"]
    assert line["rccl"]["one_copy"] is True
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["peak"] == 8000.0
    tail = raw[-1500:]
    assert '"band_check"' in tail and '"legs"' in tail and '"rccl"' in tail
    if with_4k:
        assert line["scaling_frame"] == "3840x2160"
        assert line["value_4k"] == line["legs"]["frame_4k"][0] > 0 and line["legs"]["frame_4k_b12_prefilter"][0] > 0
        assert '"frame_4k"' in tail and '"frame_4k_b12_prefilter"' in tail and '"value_4k"' in tail
    else:
        assert "value_4k" not in line


@pytest.mark.gpu
@pytest.mark.parametrize("with_4k", [False, True])
def test_bench_row_band_branch_on_one_gpu(with_4k):
    """`bench.py --band-path` with one rank: the branch every N > 1 run takes (native band driver over real RCCL communicators, the reduced-size
    equality check against a single-GPU run, the 4K legs over the same bands, the assembly of the JSON line) on the one GPU there is -- so that a
    change to the line cannot break the two-GPU test below unnoticed (it did once: VERDICT r5)"""
    line, raw = _run_band_bench(["--band-path", "--no-predict", "--no-cpu-baseline"] + ([] if with_4k else ["--no-extras"]), 1, 29534 + int(with_4k))
    _check_band_bench_line(line, raw, 1, with_4k)


@pytest.mark.gpu
def test_two_gpu_bench_exercises_the_rccl_transport():
    """first box with two GPUs: `bench.py --gpus 2` under torch.distributed.run drives bcd_hip_multi_rank_* over RCCL (ncclCommInitRank,
    grouped send / recv with the neighbour, the marking all-reduce) and asserts on rank 0 that the gathered frame equals the single-GPU
    frame.  Skipped on one-GPU boxes (every other multi-rank test uses the in-process transport there; the line's assembly is covered by
    test_bench_row_band_branch_on_one_gpu with the same assertions)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    line, raw = _run_band_bench(["--no-extras"], 2, 29533)
    _check_band_bench_line(line, raw, 2, False)


_RCCL_CANARY = {}


def _rccl_canary():
    """the transport self-test once in a child process with a time limit: a communication kernel that never returns must cost this test,
    not the session (the in-process calls below only run after the child came back)"""
    if "rc" not in _RCCL_CANARY:
        import subprocess
        import sys
        root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
        # (the band driver's diagnostic switches ride along: which RCCL it resolved on stderr, a short abort wait for the simulated failure)
        env = dict(_os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BCD_HIP_MULTI_VERBOSE="1", BCD_HIP_MULTI_ABORT_WAIT_MS="200")
        code = "import bcd_amd.hip as bh; rc, msg = bh.selftest_transport(0); print(msg); raise SystemExit(rc)"
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=root)
            _RCCL_CANARY["rc"], _RCCL_CANARY["msg"] = r.returncode, (r.stdout + r.stderr)[-1500:]
        except subprocess.TimeoutExpired:
            _RCCL_CANARY["rc"], _RCCL_CANARY["msg"] = -9, "the transport self-test did not return within 300 s"
    return _RCCL_CANARY["rc"], _RCCL_CANARY["msg"]


@pytest.mark.gpu
def test_rccl_transport_selftest_on_one_gpu():
    """the band driver's RCCL usage on ONE device, through its own exchange() / allreduce() / fail() / prepare(): ncclCommInitRank with n = 1 from
    real unique ids on two channels, a grouped self send / recv of two halo-sized buffers (7 lines of a 4K frame's accumulators) with the data
    checked, the int64 all-reduce, a simulated failure (ncclCommAbort; the aborted communicators refuse further use, consumed ids cannot rebuild
    them), bcd_hip_multi_rank_renew_ids with fresh ids, and a second exchange on the rebuilt communicators"""
    import bcd_amd.hip as bh
    rc, msg = _rccl_canary()
    assert rc == 0, msg
    assert "communicators from rccl version" in msg, msg      # BCD_HIP_MULTI_VERBOSE=1 in the child: which library the driver resolved
    rc, msg = bh.selftest_transport(0, 13 * 3840 * 16)          # in this process too (b = 12 halo size): librccl is mapped and used here
    assert rc == 0 and msg.startswith("ok"), msg


@pytest.mark.gpu
@pytest.mark.parametrize("m", [1.0, 0.0])
def test_whole_frame_through_the_rccl_transport_in_loopback(hipctx, m):
    """bcd_hip_multi_create_rank(0 of 1) with real ids in loopback mode: every exchange and all-reduce of the band protocol of a 3-scale frame
    is enqueued on real RCCL communicators (one per scale + the merges'), on the scales' streams between the compute kernels and under the
    issue-order gate -- with the sizes a band inside a larger world would use -- and the frame is still the single-GPU frame"""
    import bcd_amd.hip as bh
    import bcd_amd.core as core
    rc, msg = _rccl_canary()
    assert rc == 0, msg
    W, H, S, b = 256, 288, 3, 6
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 21, 0.12, 0.005)
    prm = bh.default_params(m=m, random_order=1, seed=9, b=b)
    want = hipctx.denoise_host(col, ns, hist, cov, S, prm)
    rd = bh.RankDenoiser(0, 1, 0, bh.multi_unique_ids(S + 1))
    try:
        rd.set_loopback(True)
        rd.set_comm_trace(True)
        first, lines, own0, owned = rd.configure(W, H, hist.shape[2], S, prm)
        assert (first, lines, own0, owned) == (0, H, 0, H)
        rd.upload(col, ns, hist, cov)
        rd.step()
        got = rd.download()
        trace = rd.comm_trace()
        rd.step()                                               # steady state: same communicators, second frame
        again = rd.download()
        st = rd.stats()
    finally:
        rd.close()
    assert st.transport == 1 and st.frames == 2
    assert rel_linf(got, want) < 1e-5 and rel_linf(again, want) < 1e-5
    chans = [ch for ch, _, _, _ in trace]
    assert set(chans) == set(range(S + 1))                      # every scale's communicator and the merges' one carried traffic
    halo = b + 1
    for s in range(S):
        sizes = [(up, dn) for ch, k, up, dn in trace if ch == s and k == 0]
        assert (halo * (W >> s) * 16,) * 2 in sizes                                                 # accumulator halos of the scale (sums + counts), both neighbours
    assert any(k == 1 for _, k, _, _ in trace) == (m > 0)       # the marking all-reduce
    _check_phase_major_order(trace, S, W, halo)                 # issue order: marking coarse to fine, accumulators coarse to fine, merges


@pytest.mark.gpu
def test_two_frames_in_flight_equal_the_blocking_calls(hipctx):
    """bcd_hip_denoise_begin / _wait: two contexts with a (different) frame in flight each, several rounds; every result is the blocking call's, a second
    _begin on a busy context is refused, a bad argument is reported by _begin itself"""
    import torch
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H, S = 320, 200, 3
    prm = bh.default_params(m=1.0, random_order=1, seed=5)
    frames = [dev(*core.synthetic_scene(W, H, 16, 3 + i, 0.12 + 0.1 * i, 0.005)) for i in range(2)]
    want = [hipctx.denoise(*f, S, prm).clone() for f in frames]
    other = bh.Context(0)
    try:
        outs = [torch.empty_like(want[0]) for _ in range(2)]
        ctxs = [hipctx, other]
        for rnd in range(3):
            for i in (0, 1):
                ctxs[i].denoise_begin(*frames[(i + rnd) % 2], S, prm, outs[i])
            with pytest.raises(bh.BcdHipError):
                ctxs[0].denoise_begin(*frames[0], S, prm, outs[0])          # one frame per context
            for i in (0, 1):
                ctxs[i].denoise_wait()
                assert rel_linf(outs[i].cpu().numpy(), want[(i + rnd) % 2].cpu().numpy()) < 1e-5
        with pytest.raises(bh.BcdHipError):
            other.denoise_begin(*frames[0], 0, prm, outs[1])                # no scales: refused at once, nothing in flight
        other.denoise_wait()                                                # (returns immediately)
    finally:
        other.close()


@pytest.mark.gpu
def test_release_engines_gives_the_memory_back(hipctx):
    """bcd::releaseEngines(): the cached engine contexts of libbcdcore (grow-only workspaces) are destroyed and rebuilt on demand"""
    import torch
    import bcd_amd.core as core
    col, ns, hist, cov = core.synthetic_scene(640, 360, 8, 3, 0.2, 0.0)
    ok, a, _ = core.denoise(col, ns, hist, cov, nscales=2, seed=3)
    assert ok
    torch.cuda.synchronize()
    used = lambda: (lambda f, t: t - f)(*torch.cuda.mem_get_info())
    before = used()
    core.release_engines()
    after = used()
    assert before - after > 640 * 360 * 85 * 3          # at least the distance planes of scale 0 came back
    ok, b_, _ = core.denoise(col, ns, hist, cov, nscales=2, seed=3)
    assert ok and rel_linf(b_, a) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("spike_factor", [0.0, 2.0])
def test_streamed_host_upload_equals_the_resident_path(hipctx, spike_factor):
    """bcd_hip_denoise_host_ex on frames of >= 256 lines uploads in row chunks and computes the finest scale's distance planes (and
    the prefilter) for the lines that have arrived: the result must be the resident-input result -- same kernels, another launch
    partition -- with and without the prefilter, and also when the uniform-sample-count guess taken from the first pixel is wrong
    (the kernel's own check sends the scale to the exact kernels)"""
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H = 200, 300
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 9, 0.2, 0.01)
    prm = bh.default_params(m=1.0, random_order=1, seed=5)
    for variant in ("uniform", "one pixel differs"):
        if variant != "uniform":
            ns = ns.copy(); hist = hist.copy()
            ns[211, 77] = 15.0                       # (the first pixel still says 16)
            hist[211, 77] *= 15.0 / 16.0
        got = hipctx.denoise_host(col, ns, hist, cov, 3, prm, spike_factor=spike_factor)
        path = hipctx.stats(0).similarity_path
        d = dev(col, ns, hist, cov)
        if spike_factor > 0:
            d = hipctx.spike_filter(*d, spike_factor)
        want = hipctx.denoise(*d, 3, prm).cpu().numpy()
        assert rel_linf(got, want) < 1e-5
        # both paths stay on the approximate-planes kernels: a single odd pixel that the host's sample (streamed path) or the speculative
        # launch on the first pixel's count (resident path) misses is caught by the kernel's per-pixel check, and the pass is repeated with
        # general sample counts -- since round 6 by the RATIO form of the same kernel (similarity_path 2)
        # (with the prefilter the odd pixel may itself be replaced by a neighbour, which makes the counts uniform again)
        both = (path, hipctx.stats(0).similarity_path)
        assert both == (1, 1) if (variant == "uniform" or spike_factor > 0 and both == (1, 1)) else both == (2, 2)


@pytest.mark.gpu
def test_sparse_histogram_upload_is_lossless_and_falls_back_on_dense_images(hipctx, monkeypatch):
    """bcd_hip_denoise_host on frames of >= 256 lines sends the histogram image without its zeros (bit-pattern test: -0.0f and denormals travel as
    values), packed by host threads piece by piece and rebuilt by a kernel: the result must be bit-identical to the plain-copy path's inputs --
    checked through the frame (same kernels afterwards) and through the byte counters -- and an image that is mostly non-zero is copied as it is"""
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H, S = 320, 288, 3
    col, ns, hist, cov = core.synthetic_scene(W, H, 16, 11, 0.2, 0.01)
    hist = hist.copy()
    hist[5, 7, 3] = np.float32(-0.0)                      # (bit pattern 0x80000000: a value for the packer, zero for the arithmetic)
    hist[100, 200, 59] = np.float32(1e-41)                # a denormal
    prm = bh.default_params(m=1.0, random_order=1, seed=5)
    got = hipctx.denoise_host(col, ns, hist, cov, S, prm)
    raw, sent = hipctx.last_upload_bytes()
    assert raw == hist.size * 4 and sent < 0.6 * raw      # 16 spp over 60 bins: most values are zero
    want = hipctx.denoise(*dev(col, ns, hist, cov), S, prm).cpu().numpy()
    assert rel_linf(got, want) < 1e-6                     # (the accumulators' atomics: ~1e-7)
    monkeypatch.setenv("BCD_HIP_SPARSE_UPLOAD", "0")
    plain = bh.Context(0)
    try:
        ref = plain.denoise_host(col, ns, hist, cov, S, prm)
        assert plain.last_upload_bytes() == (raw, raw)
    finally:
        plain.close()
    assert rel_linf(got, ref) < 1e-6
    # a dense image (every bin of every pixel occupied): recognised on the first piece, copied as it is; same result as the resident path
    dense = (hist + np.float32(0.25)).astype(np.float32)
    got_d = hipctx.denoise_host(col, ns, dense, cov, S, prm)
    assert hipctx.last_upload_bytes() == (raw, raw)
    want_d = hipctx.denoise(*dev(col, ns, dense, cov), S, prm).cpu().numpy()
    ok = np.isfinite(want_d)
    assert np.array_equal(np.isfinite(got_d), ok) and rel_linf(np.where(ok, got_d, 0), np.where(ok, want_d, 0)) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("sigma,spp", [(0.35, 32), (0.10, 32), (0.35, 8)])
def test_textured_frame_masks_and_parity(hipctx, sigma, spp):
    """the band-limited texture scene (SyntheticScene pattern 1): distances spread continuously across the threshold, so the approximate
    planes leave thousands of borderline pairs to the exact re-evaluation (wave-aggregated list appends, k_verify_pairs) -- masks and
    counts must still be the oracle's bit for bit, and the denoised frame within tolerance"""
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    W, H = 200, 120
    col, ns, hist, cov = core.synthetic_scene(W, H, spp, 1234, sigma, 0.0, pattern=1)
    d = dev(col, ns, hist, cov)
    mask, cnt = hipctx.similarity_masks(d[2], d[1], 1, 6, 1.0)
    wmask, wcnt = ol.similarity_masks(ns, hist, 1, 6, 1.0)
    assert np.array_equal(mask.cpu().numpy().view(np.uint32), wmask) and np.array_equal(cnt.cpu().numpy(), wcnt)
    prm = bh.default_params(m=1.0, random_order=1, seed=3)
    got = hipctx.denoise(*d, 1, prm).cpu().numpy()
    st = hipctx.stats(0)
    assert st.similarity_path == 1 and st.borderline_pairs > 100          # the threshold band was populated, and decided exactly
    want = ol.denoise_mono(col, ns, hist, cov, ol.params(m=1.0), order=_orders(W, H, 1, 1, 3, 1)[0])
    assert rel_linf(got, want) < TOL


@pytest.mark.gpu
def test_multi_rank_frame_timeout_fails_the_frame_and_the_handle_recovers(hipctx):
    """a frame of several ranks that does not finish within the handle's limit (BCD_HIP_MULTI_TIMEOUT_S, bcd_hip_multi_set_frame_timeout)
    is failed by the watchdog thread -- host barriers released; on the RCCL transport the communicators are aborted -- instead of
    hanging; the next frame on the same handle starts from clean barriers and gates and gives the right result"""
    import bcd_amd.hip as bh
    import bcd_amd.core as core
    W, H, S = 1280, 720, 3
    frame = core.synthetic_scene(W, H, 8, 3, 0.25, 0.01)
    prm = bh.default_params(m=1.0, random_order=1, seed=9)
    md = bh.MultiDenoiser([0] * 4)
    try:
        md.denoise_host(*frame, S, prm)                 # (allocations done)
        md.set_frame_timeout(1)                         # one millisecond: no frame of this size makes it
        with pytest.raises(bh.BcdHipError, match="timed out"):
            md.denoise_host(*frame, S, prm)
        md.set_frame_timeout(600000)
        got = md.denoise_host(*frame, S, prm)           # same handle, after the failure
    finally:
        md.close()
    want = hipctx.denoise_host(*frame, S, prm)
    assert rel_linf(got, want) < 1e-5


@pytest.mark.gpu
def test_seeded_random_configurations_against_the_oracle(hipctx):
    """forty seeded random configurations (tools/fuzz_parity.py: ragged / narrow / odd frame sizes, 1-3 scales, b = 1 .. 12, tau 0.5 .. 2, 1 .. 48 samples per
    pixel and per-pixel mixtures, -m 0 / 1, -r 0 / 1): masks and |S| of the finest scale bit for bit, the frame's finite pattern, relative L-inf < 1e-4.
    (2 000 cases of the same generator ran clean in round 6; the worst were 2-3-spp frames with sigma 0.05 at 4e-5: ill-conditioned inverses.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    lines = []
    bad, refused, worst = fz.run_cases(hipctx, 40, 2026, say=lines.append)
    assert bad == 0 and refused == 0 and worst < TOL, "\n".join(l for l in lines if "MISMATCH" in l or "refused" in l)
    # ... and thirty of the --wide sequence: host buffers (streamed upload), host buffers + spike prefilter, fractional -m, D = 12 .. 120, w = 0 / 2, -e 1e-3
    # (600 cases of it ran clean in round 6, worst 3.1e-5), ten of them also through the row-band driver on 2-4 virtual ranks
    bad, refused, worst = fz.run_cases(hipctx, 30, 31337, say=lines.append, wide=True)
    assert bad == 0 and refused == 0 and worst < TOL, "\n".join(l for l in lines if "MISMATCH" in l or "refused" in l)
    bad, refused, worst = fz.run_cases(hipctx, 10, 4242, say=lines.append, bands=True)
    assert bad == 0 and refused == 0, "\n".join(l for l in lines if "MISMATCH" in l or "refused" in l)


@pytest.mark.gpu
def test_seeded_random_geometries_of_the_streaming_stages_bitexact(hipctx):
    """tools/fuzz_streaming.py: 150 seeded random geometries (4 .. 260 x 4 .. 180 pixels, 1-16 samples, 4-40 bins, weighted samples, filter factors) of the pyramid
    reducers, interpolate, merge, spike filter, per-pixel covariances and the samples accumulator against the oracle -- pinned to the reference's compiled units
    for exactly these stages -- bit for bit (accumulator histograms: device powf round-off).  (5 000 cases ran clean in round 6.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_streaming", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_streaming.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    lines = []
    bad = fz.run_cases(hipctx, 150, 7, say=lines.append)
    assert bad == 0, "\n".join(l for l in lines if "MISMATCH" in l)
