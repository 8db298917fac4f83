"""CPU tests of the multi-GPU row-band path: partition bookkeeping, and the full orchestration (per-band pyramid,
accumulator / output halo exchange, merge at band edges) executed with the oracle as the per-band engine --
in-process ("virtual ranks") and as two real processes over gloo."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle_lib as ol
from oracle_engine import OracleEngine

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bcd_amd.tiling import BandGeometry, run_distributed, run_virtual  # noqa: E402


class Prm:
    hist_dist_threshold = 1.0
    patch_radius = 1
    search_radius = 3
    min_eigen_value = 1e-8
    use_random_pixel_order = 0
    marked_skip_probability = 0.0
    order_seed = 5


@pytest.mark.parametrize("W,H,S,b,world", [(64, 64, 3, 6, 2), (3840, 2160, 3, 6, 8), (1280, 720, 3, 6, 8), (101, 67, 2, 3, 3), (50, 40, 1, 6, 2)])
def test_geometry_invariants(W, H, S, b, world):
    g = BandGeometry(W, H, S, b, 1, world)
    for s in range(S):
        covered = []
        for r in range(world):
            sb = g.scale_bands(r)[s]
            assert 0 <= sb.loc0 <= sb.own0 < sb.own1 <= sb.loc1 <= (H >> s)
            assert sb.loc0 % 2 == 0
            if r > 0:
                assert sb.own0 - sb.loc0 >= g.halo      # input halo above
            if r < world - 1:
                assert sb.loc1 - sb.own1 >= g.halo      # input halo below
            if s + 1 < S:
                nxt = g.scale_bands(r)[s + 1]
                assert sb.loc0 <= 2 * nxt.loc0 and 2 * nxt.loc1 <= sb.loc1   # the coarser level is buildable locally
                assert sb.own0 % 2 == 0
            covered.append((sb.own0, sb.own1))
        assert covered[0][0] == 0 and covered[-1][1] == (H >> s)
        assert all(covered[i][1] == covered[i + 1][0] for i in range(world - 1))   # a partition of the lines


def test_geometry_rejects_thin_bands():
    with pytest.raises(ValueError):
        BandGeometry(64, 40, 3, 6, 1, 8)


def _inputs(W, H, spp=6):
    col, ns, hist, cov, _ = ol.synth_inputs(W, H, spp, 99, 0.2, 0.0)
    return col, ns, hist, cov


def _slice(arrs, g, r):
    l0, l1 = g.input_lines(r)
    return [torch.from_numpy(np.ascontiguousarray(a[l0:l1])) for a in arrs]


@pytest.mark.parametrize("W,H,S,world", [(40, 48, 2, 2), (37, 58, 3, 2), (36, 49, 2, 3)])
def test_virtual_ranks_match_full_frame_m0(W, H, S, world):
    """-m 0 is order-free: the band decomposition must reproduce the full-frame multiscale result"""
    arrs = _inputs(W, H)
    prm = Prm()
    g = BandGeometry(W, H, S, prm.search_radius, 1, world)
    outs = run_virtual(OracleEngine(None), g, [_slice(arrs, g, r) for r in range(world)], prm, 5)
    got = np.concatenate([o.numpy() for o in outs], 0)
    want = ol.denoise_multiscale(*arrs, S, ol.params(b=prm.search_radius, m=0.0, threads=1))
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 2e-6


def _gloo_worker(rank, world, port, W, H, S, q, exact=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    arrs = _inputs(W, H)
    prm = Prm()
    if exact:
        prm.marked_skip_probability = 1.0
        prm.use_random_pixel_order = 1
    g = BandGeometry(W, H, S, prm.search_radius, 1, world)
    out = run_distributed(OracleEngine(None), g, rank, dist, _slice(arrs, g, rank), prm, 5, exact_marking=exact)
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_gloo_match_full_frame():
    import torch.multiprocessing as mp
    W, H, S, world = 36, 40, 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, W, H, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got = np.concatenate([res[r] for r in range(world)], 0)
    want = ol.denoise_multiscale(*_inputs(W, H), S, ol.params(b=Prm.search_radius, m=0.0, threads=1))
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 2e-6


def _visit_order_global(W, H, w, random_order, seed):
    """main pixels of the full frame sorted by the build's visiting key (python restatement of bcd_hip_visit_order)"""
    eng = OracleEngine(None)
    idx = np.array([l * W + c for l in range(w, H - w) for c in range(w, W - w)], np.uint64)
    return idx[np.argsort(eng._keys(idx, random_order, seed), kind="stable")].astype(np.int32)


@pytest.mark.parametrize("W,H,S,world,random_order", [(30, 40, 1, 2, 1), (30, 40, 1, 2, 0), (28, 44, 2, 3, 1)])
def test_exact_marking_across_bands_matches_full_frame_m1(W, H, S, world, random_order):
    """-m 1 with the exact band program: boundary states are exchanged between marking rounds and keys are global, so the
    processed set -- and the image -- are those of the whole frame visited in the same order"""
    arrs = _inputs(W, H, spp=8)
    prm = Prm()
    prm.marked_skip_probability = 1.0
    prm.use_random_pixel_order = random_order
    g = BandGeometry(W, H, S, prm.search_radius, 1, world)
    outs = run_virtual(OracleEngine(None), g, [_slice(arrs, g, r) for r in range(world)], prm, 5, exact_marking=True)
    got = np.concatenate([o.numpy() for o in outs], 0)
    orders, w_, h_ = [], W, H
    for s in range(S):
        orders.append(_visit_order_global(w_, h_, 1, random_order, 5 + s))
        w_, h_ = w_ // 2, h_ // 2
    want = ol.denoise_multiscale(*arrs, S, ol.params(b=prm.search_radius, m=1.0), orders=orders)
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 2e-6
    # the per-band (default) program is a different, valid greedy: it must NOT be required to match
    Prm.marked_skip_probability = 0.0


def test_python_keys_match_the_library_order():
    import bcd_amd.hip as bh
    for ro in (0, 1):
        assert np.array_equal(_visit_order_global(23, 17, 1, ro, 77), bh.visit_order(23, 17, 1, ro, 77))


def test_two_processes_gloo_exact_marking():
    """the exact -m 1 band program over real processes: point-to-point state exchanges + the all-reduce of undecided counts"""
    import torch.multiprocessing as mp
    W, H, S, world = 30, 36, 1, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, W, H, S, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got = np.concatenate([res[r] for r in range(world)], 0)
    want = ol.denoise_mono(*_inputs(W, H), ol.params(b=Prm.search_radius, m=1.0), order=_visit_order_global(W, H, 1, 1, 5))
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 2e-6


@pytest.mark.parametrize("m,random_order", [(0.0, 0), (1.0, 1)])
def test_three_scales_b6_bands_match_full_frame(m, random_order):
    """BASELINE configs[3]'s geometry (3 scales, b = 6: bands aligned to 4 lines, 7 halo lines per scale) at a reduced size, two
    virtual ranks, oracle engine: -m 0 and the exact -m 1 program against the full frame in the same visiting order"""
    W, H, S, world = 40, 64, 3, 2
    arrs = _inputs(W, H, spp=8)
    prm = Prm()
    prm.search_radius = 6
    prm.marked_skip_probability = m
    prm.use_random_pixel_order = random_order
    g = BandGeometry(W, H, S, prm.search_radius, 1, world)
    outs = run_virtual(OracleEngine(None), g, [_slice(arrs, g, r) for r in range(world)], prm, 5, exact_marking=m > 0)
    got = np.concatenate([o.numpy() for o in outs], 0)
    orders, w_, h_ = None, W, H
    if m > 0:
        orders = []
        for s in range(S):
            orders.append(_visit_order_global(w_, h_, 1, random_order, 5 + s))
            w_, h_ = w_ // 2, h_ // 2
    want = ol.denoise_multiscale(*arrs, S, ol.params(b=6, m=m, threads=1), orders=orders)
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 2e-6


def _gloo_worker_b6(rank, world, port, W, H, S, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    arrs = _inputs(W, H, spp=8)
    prm = Prm()
    prm.search_radius = 6
    prm.marked_skip_probability = 1.0
    prm.use_random_pixel_order = 1
    g = BandGeometry(W, H, S, prm.search_radius, 1, world)
    out = run_distributed(OracleEngine(None), g, rank, dist, _slice(arrs, g, rank), prm, 5, exact_marking=True)
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_gloo_three_scales_b6_exact_marking():
    """the same over two real processes (gloo): isend / irecv of |S|, states, accumulator and output lines + the all-reduce"""
    import torch.multiprocessing as mp
    W, H, S, world = 40, 64, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker_b6, args=(r, world, port, W, H, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got = np.concatenate([res[r] for r in range(world)], 0)
    orders, w_, h_ = [], W, H
    for s in range(S):
        orders.append(_visit_order_global(w_, h_, 1, 1, 5 + s))
        w_, h_ = w_ // 2, h_ // 2
    want = ol.denoise_multiscale(*_inputs(W, H, spp=8), S, ol.params(b=6, m=1.0), orders=orders)
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 2e-6
