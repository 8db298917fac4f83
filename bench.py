#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the denoising hot path on MI355X (BASELINE.json metric).

A "step" is one full denoise of one synthetic 1920x1080 frame (BASELINE.json configs[2]: 3-scale, b=6, w=1, tau=1, m=1,
seeded random order): the pyramid build, and per scale the pair-distance / mask kernels, the marking fixed point, the
Bayesian patch kernel, finalisation and merge.  Inputs are resident in HBM before the timed region.
After the timed region (N = 1, untimed, skipped by --no-extras): the pair-distance kernel with the scales serialised
(`roofline.isolated_*`), the host-buffer call a drop-in user makes (`end_to_end`: PCIe both ways inside), the low-noise variant of
the frame (`low_noise`), a textured frame whose similar sets depend on the noise (`textured`), the same frame with -m 0 (`m0`), two frames in flight on two engine
contexts (`pipelined_2frames`), frames with general sample counts (`nonuniform_*`), the 3840x2160
frame of BASELINE.json configs[3] (`frame_4k`; at N > 1 the same frame over the same row bands, measured before the timed
region: the per-N points of the 4K strong-scaling curve) and of configs[4] (`frame_4k_b12_prefilter`, N = 1), and the CPU oracle on all host cores and on one core (`cpu_baseline`).
N > 1 (launched by torch.distributed.run, one rank per GPU): the SAME frame is split into horizontal bands of
main pixels (strong scaling); every rank owns a band plus (b+w)*2^(S-1) halo lines of input, rebuilds the pyramid
for its band, and exchanges marking states, accumulator and output halo lines with its two neighbours over RCCL
(native driver, bcd_hip_multi_rank_*; the marking follows the visiting order of the whole frame, so the gathered
frame IS the single-GPU frame -- checked on rank 0 at a reduced size before the timed region).

One JSON line on rank 0 (contract in the task statement), with the extra objects `roofline` (pair-distance
kernel, HIP-event timed inside this process) and `cpu_baseline` (oracle on host cores, bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_READ_BYTES_PER_PIXEL = 280  # (D+1+3+6)*4 bytes read per pixel of a scale, D = 60


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scales", type=int, default=3)
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--sigma", type=float, default=0.35, help="synthetic noise level (0.35 + 1%% spikes = SURVEY probe)")
    ap.add_argument("--spikes", type=float, default=0.01)
    ap.add_argument("--pattern", type=int, default=0, help="synthetic scene: 0 = ramps + 16-pixel checker (the SURVEY probe scene, headline), 1 = band-limited texture")
    ap.add_argument("--search-radius", type=int, default=6)
    ap.add_argument("--skip-prob", type=float, default=1.0, help="-m of bcd_cli")
    ap.add_argument("--random-order", type=int, default=1, help="-r of bcd_cli")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed legs after the timed region (isolated kernel timing, low-noise frame, -m 0)")
    ap.add_argument("--no-4k", action="store_true", help="skip the untimed 3840x2160 leg (frame_4k in the JSON line)")
    ap.add_argument("--python-bands", action="store_true", help="N > 1: the torch.distributed orchestration of bcd_amd/tiling.py instead of the native driver (test harness)")
    ap.add_argument("--band-marking", action="store_true", help="with --python-bands: every band marks on its own (a valid order, NOT the single-GPU frame)")
    ap.add_argument("--band-path", action="store_true", help="use the multi-GPU row-band code path even with one rank (debug)")
    ap.add_argument("--check-size", default="640x576", help="N > 1: frame size of the equality check against a single-GPU run on rank 0")
    ap.add_argument("--watchdog", type=int, default=900, help="N > 1: seconds after which a run that has not reached its JSON line dumps every thread's stack and exits (a blocked collective would otherwise hang the launcher)")
    ap.add_argument("--no-predict", action="store_true", help="skip the `predicted_8gpu` leg (one rank's band of the 4K frame through the band driver with RCCL in loopback, in a child process)")
    ap.add_argument("--predict-band-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-sample", default=None, help="frame size of the CPU-baseline sample on all host cores (default: the headline frame itself on hosts with >= 64 cores, 960x540 below)")
    ap.add_argument("--cpu-sample-1core", default="320x240", help="frame size of the one-core CPU-baseline sample")
    return ap.parse_args()


def cpu_baseline(args):
    """oracle (own C port of the reference CPU/OpenMP path) on the host cores, bounded samples of the same workload:
    same generator, same flags, reference-style OpenMP scheduling (racy marks, strip order); all cores and --ncores 1."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    import bcd_amd.core as core
    cores = os.cpu_count() or 1

    def run(size, threads, reps):
        w, h = [int(v) for v in size.split("x")]
        col, ns, hist, cov = core.synthetic_scene(w, h, args.spp, 1234, args.sigma, args.spikes)
        prm = ol.params(tau=1.0, w=1, b=args.search_radius, m=args.skip_prob, threads=threads)
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            ol.denoise_multiscale(col, ns, hist, cov, args.scales, prm, racy=threads > 1)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return w, h, best

    # the headline frame itself when the host can do it inside the bounded sample (~10 s per run on the GPU box's 256 cores)
    size = args.cpu_sample or ("%dx%d" % (args.width, args.height) if cores >= 64 else "960x540")
    reps = 3
    w, h, best = run(size, cores, reps)
    # the reference's schedule(dynamic, (W - 2) * 2b) hands out strips of 2b lines: at most ceil((H - 2) / 2b) threads have work
    strips = [max(1, -(-((h >> s) - 2) // (2 * args.search_radius))) for s in range(args.scales)]
    w1, h1, best1 = run(args.cpu_sample_1core, 1, 1)
    return {"value": round(w * h / 1e6 / best, 5), "unit": "Mpix/s", "cores": cores, "kind": "port",
            "effective_parallelism_per_scale": [min(cores, n) for n in strips],
            "sample": "%dx%d synthetic frame (same generator/flags), %d-scale, OpenMP dynamic strips of 2b lines like the reference "
                      "(Denoiser.cpp:149-205), best of %d, %.2f s%s" % (w, h, args.scales, reps, best, " -- the headline frame" if (w, h) == (args.width, args.height) else ""),
            "one_core": {"value": round(w1 * h1 / 1e6 / best1, 5), "unit": "Mpix/s", "cores": 1,
                         "sample": "%dx%d synthetic frame, --ncores 1 (sequential visiting order), %.2f s" % (w1, h1, best1)}}


def per_scale_stats(ctx, S):
    scales = []
    for s in range(S):
        st = ctx.stats(s)
        scales.append({"scale": s, "w": st.width, "h": st.height, "processed": st.processed, "fallback": st.fallback,
                       "full_estimates": st.processed - st.fallback, "similar_total": st.similar_total,
                       "processed_frac": round(st.processed / max(1, st.main_pixels), 4),
                       "fallback_frac": round(st.fallback / max(1, st.processed), 4),
                       "mean_similar": round(st.similar_total / max(1, st.processed), 2), "rounds": st.active_rounds,
                       "borderline_pairs": st.borderline_pairs if st.similarity_path >= 1 else None, "similarity_path": st.similarity_path, "cu_share_pct": st.cu_share, "spectral_inverses": st.spectral_inverses})
    return scales


def source_hash(rel):
    """first 16 hex digits of the sha256 of a source file: profiles taken offline carry the hash of the kernel source they were taken on"""
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


PAIRDIST_SOURCE = "bcd_amd/csrc/k_similarity_fast.hip"
VALU_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: peak FP32 (vector), spec


def predict_band_child(args):
    """One rank's share of the 4K frame on this one GPU, through the band driver: world = 8 cuts 3840x2160 (S = 3, boundaries aligned to 4 lines)
    into bands of 268 / 272 owned lines; rank 3 owns lines [808, 1080) and computes, per scale, its band plus 7 halo lines on either side.  The
    stand-in is the frame made of lines [800, 1088) of the 4K frame (288 / 144 / 72 lines per scale against 286 / 150 / 82 in the band), run as rank 0
    of 1 in LOOPBACK: every exchange and all-reduce of the band protocol is enqueued on real RCCL communicators with the band's message sizes
    (the rank is its own neighbour; what it receives is discarded).  Waiting for real neighbours is NOT in it."""
    import torch
    import bcd_amd.core as core
    import bcd_amd.hip as bh
    S, b = args.scales, args.search_radius
    w4, h4, l0, nl = 3840, 2160, 800, 288
    prm = bh.default_params(b=b, w=1, m=args.skip_prob, random_order=args.random_order, seed=1234)
    rd = bh.RankDenoiser(0, 1, 0, bh.multi_unique_ids(S + 1))
    rd.set_loopback(True)
    rd.configure(w4, nl, 60, S, prm)
    rd.upload(*core.synthetic_scene(w4, h4, args.spp, 1234, args.sigma, args.spikes, l0, nl))
    rd.step()
    rd.step()
    rd.set_comm_trace(True)
    rd.step()
    trace = rd.comm_trace()
    rd.set_comm_trace(False)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        rd.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    rd.close()
    res = {"band_ms": round(ms, 4), "steps": reps,
           "exchanges_per_frame": sum(1 for _, k, _, _ in trace if k == 0),
           "marking_allreduces_per_frame": sum(1 for _, k, _, _ in trace if k == 1),
           "bytes_per_neighbour": int(sum(up for _, k, up, _ in trace if k == 0)),
           "largest_message_bytes": int(max([up for _, k, up, _ in trace if k == 0] or [0]))}
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    print("PREDICT " + json.dumps(res), flush=True)


def predicted_8gpu(args, frame_ms):
    """runs predict_band_child in a child process with a time limit (a communication kernel that never returns must not cost the bench line)"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--predict-band-child", "--scales", str(args.scales), "--search-radius", str(args.search_radius),
           "--spp", str(args.spp), "--sigma", str(args.sigma), "--spikes", str(args.spikes), "--skip-prob", str(args.skip_prob), "--random-order", str(args.random_order)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("PREDICT ")]
        if r.returncode != 0 or not line:
            return {"error": "child failed (rc %d): %s" % (r.returncode, (r.stdout + r.stderr)[-400:])}
        res = json.loads(line[-1][len("PREDICT "):])
    except subprocess.TimeoutExpired:
        return {"error": "child did not finish within 240 s"}
    res["frame_ms"] = None if frame_ms is None else round(frame_ms, 4)
    res["speedup_before_waiting"] = None if frame_ms is None else round(frame_ms / res["band_ms"], 2)
    res["workload"] = ("one of 8 row bands of the 3840x2160 frame (BASELINE configs[3]): lines [800, 1088) as a frame of its own through the band driver as rank 0 of 1 in "
                       "loopback -- every exchange / all-reduce of the band protocol on real RCCL communicators with the band's message sizes, the rank being its own "
                       "neighbour; frame_ms = the whole frame on this GPU (frame_4k).  A PREDICTION from one GPU: waiting for real neighbours and xGMI transfers are not in it")
    return res


def main():
    args = parse()
    # multi-process GPU work on these hosts needs dmabuf IPC (the host driver does not support the legacy IPC mode: without it RCCL's
    # hipIpcGetMemHandle fails); the variable is read when the HSA runtime starts, i.e. before torch is imported
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.predict_band_child:
        return predict_band_child(args)
    import torch
    import bcd_amd.core as core
    import bcd_amd.hip as bh

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.band_path:
        import faulthandler
        if args.watchdog > 0:
            faulthandler.dump_traceback_later(args.watchdog, exit=True)
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29511"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    W, H, S, b, w = args.width, args.height, args.scales, args.search_radius, 1
    prm = bh.default_params(b=b, w=w, m=args.skip_prob, random_order=args.random_order, seed=1234)
    # one side stream for everything: engine kernels, torch glue and (N > 1) the RCCL point-to-point ops are stream-ordered
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = bh.Context(local_rank, stream)

    extras = {}   # untimed legs reported next to the headline
    parallelism = "single"
    if world == 1 and not args.band_path:
        col, ns, hist, cov = core.synthetic_scene(W, H, args.spp, 1234, args.sigma, args.spikes, pattern=args.pattern)
        d_in = [torch.from_numpy(a).cuda() for a in (col, ns, hist, cov)]
        out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")

        def step():
            ctx.denoise(*d_in, S, prm, out)
    else:
        def python_bands(exact):
            """the same decomposition over torch.distributed (bcd_amd/tiling.py)"""
            from bcd_amd.tiling import BandDenoiser
            band = BandDenoiser(ctx, dist, rank, world, W, H, 60, S, prm, exact_marking=exact)
            g0, g1 = band.input_lines()
            band.upload(*core.synthetic_scene(W, H, args.spp, 1234, args.sigma, args.spikes, g0, g1 - g0))
            return band.step, "rowband%d-%s-torchdist" % (world, "exactmark" if exact else "bandmark")

        def native_bands():
            """native driver, one process per GPU: RCCL unique ids from rank 0, one communicator per scale + one for the merges"""
            ids = [bh.multi_unique_ids(S + 1) if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(ids, src=0)
            rd = bh.RankDenoiser(rank, world, local_rank, ids[0])

            def load(w_, h_):
                l0, nl, _, _ = rd.configure(w_, h_, 60, S, prm)
                rd.upload(*core.synthetic_scene(w_, h_, args.spp, 1234, args.sigma, args.spikes, l0, nl))
            # ---- the band path must reproduce the single-GPU frame: checked at a reduced size, outside the timed region
            cw, ch = [int(v) for v in args.check_size.split("x")]
            load(cw, ch)
            rd.step()
            parts = [None] * world
            if world > 1:
                dist.all_gather_object(parts, rd.download())
            else:
                parts = [rd.download()]
            check = None
            if rank == 0:
                frame = core.synthetic_scene(cw, ch, args.spp, 1234, args.sigma, args.spikes)
                want = ctx.denoise(*[torch.from_numpy(a).cuda() for a in frame], S, prm).cpu().numpy()
                got = np.concatenate(parts, 0)
                err = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
                assert got.shape == want.shape and err < 1e-5, "band path differs from the single-GPU frame: %g" % err
                check = {"size": "%dx%d" % (cw, ch), "rel_linf_vs_single_gpu": err}
            # BASELINE configs[3]: the 4K frame row-banded over the same ranks (the strong-scaling target of north_star is quoted on it);
            # same calls as the timed region below, before it
            if not (args.no_extras or args.no_4k):
                w4, h4 = 3840, 2160
                load(w4, h4)
                rd.step()   # two untimed steps: the first one at a new size grows the workspaces, the second one settles the
                rd.step()   # size of the first marking batch
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t1 = time.perf_counter()
                for _ in range(3):
                    rd.step()
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t4 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
                if world > 1:
                    dist.all_reduce(t4, op=dist.ReduceOp.MAX)
                ms4 = float(t4.item()) * 1e3 / 3
                extras["frame_4k"] = {"value": round(w4 * h4 / 1e6 / (ms4 * 1e-3), 3), "unit": "Mpix/s", "ms_per_step": round(ms4, 4), "steps": 3,
                                      "workload": "3840x2160 frame of the same generator and flags (BASELINE configs[3]) over the same %d row bands, inputs resident" % world}
                # BASELINE configs[4] over the same row bands: -b 12 -r 1 with the spike prefilter.  The prefilter (a 3 x 3
                # per-pixel kernel: 0.8 ms of the 46 ms frame at N = 1, where it is inside the timed step) runs once per rank on its band's input
                # lines before the upload -- with one extra line on either interior side, dropped afterwards, so that every kept line sees the
                # neighbourhood it has in the whole frame -- and is NOT in the timed region here.
                prm12 = bh.default_params(b=12, w=w, m=args.skip_prob, random_order=1, seed=1234)
                l0, nl, _, _ = rd.configure(w4, h4, 60, S, prm12)
                g0, g1 = max(0, l0 - 1), min(h4, l0 + nl + 1)
                part = [torch.from_numpy(a).cuda() for a in core.synthetic_scene(w4, h4, args.spp, 1234, args.sigma, args.spikes, g0, g1 - g0)]
                filt = ctx.spike_filter(*part, 2.0)
                rd.upload(*[np.ascontiguousarray(t.cpu().numpy()[l0 - g0:l0 - g0 + nl]) for t in filt])
                del part, filt
                rd.step()
                rd.step()
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t1 = time.perf_counter()
                for _ in range(2):
                    rd.step()
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t12 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
                if world > 1:
                    dist.all_reduce(t12, op=dist.ReduceOp.MAX)
                ms12 = float(t12.item()) * 1e3 / 2
                extras["frame_4k_b12_prefilter"] = {"value": round(w4 * h4 / 1e6 / (ms12 * 1e-3), 3), "unit": "Mpix/s", "ms_per_step": round(ms12, 4), "steps": 2,
                                                    "workload": "3840x2160, -b 12 -r 1, prefiltered inputs (BASELINE configs[4]) over the same %d row bands; the prefilter itself is outside the timed region at N > 1" % world}
            load(W, H)
            rd.step()   # (workspaces and the size of the first marking batch settle in two steps at a new frame size; untimed,
            rd.step()   # before the warm-up steps the caller asked for)
            return rd.step, "rowband%d-exactmark-native" % world, check

        band_check, native_error = None, None
        if args.python_bands:
            step, parallelism = python_bands(not args.band_marking)
        else:
            try:
                step, parallelism, band_check = native_bands()
            except Exception as e:   # reported in the JSON line, never silent; every rank must take the same decision
                native_error = "%s: %s" % (type(e).__name__, e)
            failed = torch.tensor([1 if native_error else 0], device="cuda")
            if world > 1:
                dist.all_reduce(failed, op=dist.ReduceOp.MAX)
            if int(failed.item()):
                # no silent substitution: the native driver is what --gpus N measures; the torch.distributed band program is a test
                # harness and has to be asked for (--python-bands)
                raise SystemExit("bench.py --gpus %d: the native multi-GPU driver failed (%s); rerun with --python-bands to time the "
                                 "torch.distributed test harness instead" % (world, native_error or "on another rank"))
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    kt_warm = ctx.kernel_time()  # (ms, launches) of the pair-distance kernel so far: the event pool is never reset, legs are differences
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_step = elapsed * 1e3 / args.steps

    # ---- roofline of the dominant kernel (pair-distance planes), HIP events on the engine's streams over the timed region
    kt_timed = ctx.kernel_time()
    pd_ms, pd_launches = kt_timed[0] - kt_warm[0], kt_timed[1] - kt_warm[1]
    single = world == 1 and not args.band_path
    # (per-scale counters and kernel timing come from the single-GPU engine context; the band drivers keep their contexts inside)
    scales = per_scale_stats(ctx, S) if single else [{"scale": s_, "w": W >> s_, "h": H >> s_, "borderline_pairs": None} for s_ in range(S)]
    # the three scales run concurrently on separate streams, so a launch's event-to-event time includes the kernels it
    # overlaps with; the same kernel timed in isolation (scales one after the other, three extra untimed steps):
    iso_ms = None
    frame4k_ms = None
    valu = None
    if single and not args.no_extras:
        # ---- the arithmetic of the distance kernel on the headline frame's finest scale (the bound that binds: SURVEY 8(d) "honest second bound")
        try:
            lane_bins, wave_bins, wave_groups, k_ms = ctx.selftest_bin_work(d_in[2], d_in[1], b, 3)
            nd = (b + 1) + b * (2 * b + 1)
            slots = W * H * nd * 60
            flop = 2 * slots + 6 * lane_bins
            valu = {"scale": 0, "bin_slots": slots, "evaluated_bins": lane_bins, "evaluated_frac": round(lane_bins / slots, 4),
                    "wave_issued_bins_x64": wave_bins * 64, "wave_issued_frac": round(wave_bins * 64 / slots, 4), "wave_groups_entered": wave_groups,
                    "flop": flop, "kernel_ms": round(k_ms, 4), "achieved_tflops": round(flop / (k_ms * 1e-3) / 1e12, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(flop / (k_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4),
                    "issued_tflops": round((2 * slots + 6 * 64 * wave_bins) / (k_ms * 1e-3) / 1e12, 3)}
        except Exception as e:   # (reported, never fatal for the headline)
            valu = {"error": "%s: %s" % (type(e).__name__, e)}
    if single and not args.no_extras:
        ctx.set_concurrent_scales(False)
        step()
        k0 = ctx.kernel_time()
        step()
        step()
        k1 = ctx.kernel_time()
        iso_ms = (k1[0] - k0[0]) / max(1, k1[1] - k0[1])
        ctx.set_concurrent_scales(True)

        def leg(frame, prm_leg, reps):
            d = [torch.from_numpy(a).cuda() for a in frame]
            ctx.denoise(*d, S, prm_leg, out)   # two untimed calls: a new kind of frame grows workspaces, settles the marking batch and the list-length
            ctx.denoise(*d, S, prm_leg, out)   # guess, and (general sample counts) finds out whether a scale declines the RATIO form of the distance kernel
            torch.cuda.synchronize()
            each = []
            t1 = time.perf_counter()
            for _ in range(reps):
                t2 = time.perf_counter()
                ctx.denoise(*d, S, prm_leg, out)   # (a blocking call: every stream of the context is synchronised when it returns)
                each.append(round((time.perf_counter() - t2) * 1e3, 3))
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t1) * 1e3 / reps
            return {"value": round(W * H / 1e6 / (ms * 1e-3), 3), "unit": "Mpix/s", "ms_per_step": round(ms, 4), "steps": reps, "ms_each": each,
                    "per_scale": per_scale_stats(ctx, S)}
        # ---- what a drop-in caller gets: bcd_hip_denoise_host on HOST buffers (pageable, like a DeepImage's std::vector): upload of the
        # four planes, the same denoise, download of the result.  Never `value`; PCIe moves (D + 10) * 4 + 12 bytes per pixel.
        h_frame = (col, ns, hist, cov)
        ctx.denoise_host(*h_frame, S, prm)
        t1 = time.perf_counter()
        for _ in range(3):
            ctx.denoise_host(*h_frame, S, prm)
        ms_e2e = (time.perf_counter() - t1) * 1e3 / 3
        hist_raw, hist_sent = ctx.last_upload_bytes()
        pcie_bytes = W * H * ((1 + 3 + 6) * 4 + 12) + hist_sent
        extras["end_to_end"] = {"value": round(W * H / 1e6 / (ms_e2e * 1e-3), 3), "unit": "Mpix/s", "ms_per_frame": round(ms_e2e, 3), "steps": 3,
                                "pcie_bytes_per_frame": pcie_bytes, "pcie_gbs_if_all_of_the_time_were_transfer": round(pcie_bytes / (ms_e2e * 1e-3) / 1e9, 1),
                                "histogram_bytes": hist_raw, "histogram_bytes_sent": hist_sent,
                                "upload": "histogram image packed on the host (one bit per value + the non-zero values, lossless) while the previous piece travels, rebuilt in HBM by a kernel" if hist_sent < hist_raw else "plain copies",
                                "workload": "the headline frame through bcd_hip_denoise_host (pageable host buffers in, host buffer out; engine context and device "
                                            "staging buffers persistent): what bcd::Denoiser::denoise() / bcd_cli callers see, minus file IO"}
        # on the default frame most processed pixels take the fallback path; the low-noise variant (sigma 0.10, no
        # spikes), the textured frame and -m 0 (every main pixel processed) exercise the full Bayesian estimate
        head = [(sc["processed"], sc["fallback"], sc["similar_total"]) for sc in scales]
        low = leg(core.synthetic_scene(W, H, args.spp, 1234, 0.10, 0.0), prm, 3)
        low_c = [(sc["processed"], sc["fallback"], sc["similar_total"]) for sc in low["per_scale"]]
        # The frames differ (checked on the coarse scales' counters).  Scale 0 may legitimately show the headline's counters: at 32 spp
        # the similar sets of the checker scene are decided by the checker geometry -- pairs inside a 16-pixel cell are similar at both
        # noise levels, pairs across a cell edge are not (0 borderline pairs on both frames; tools/dbg_lownoise.py prints the identical
        # |S| histograms) -- so what the low-noise leg changes is scales 1..2 and the estimates themselves, not scale 0's processed set.
        assert low_c[1:] != head[1:] or S == 1, "the low-noise leg reported the headline frame's counters on every scale"
        extras["low_noise"] = dict(low, workload="same frame generator with sigma 0.10, no spikes",
                                   scale0_counters_equal_headline=bool(low_c[0] == head[0]),
                                   note="scale 0's similar sets follow the 16-pixel checker geometry at both noise levels (DESIGN.md 8); the legs "
                                        "whose similar sets depend on the noise are `textured` and `textured_low_noise`")
        tex = leg(core.synthetic_scene(W, H, args.spp, 1234, args.sigma, args.spikes, pattern=1), prm, 3)
        tex_low = leg(core.synthetic_scene(W, H, args.spp, 1234, 0.10, 0.0, pattern=1), prm, 3)
        assert [sc["similar_total"] for sc in tex["per_scale"]] != [sc["similar_total"] for sc in tex_low["per_scale"]]
        extras["textured"] = dict(tex, workload="band-limited texture with oblique soft edges (SyntheticScene pattern 1), same sigma / spikes / flags: "
                                                "similar sets depend on the noise, most processed pixels take the full estimate")
        extras["textured_low_noise"] = dict(tex_low, workload="the textured frame with sigma 0.10, no spikes")
        extras["m0"] = dict(leg((col, ns, hist, cov), bh.default_params(b=b, w=w, m=0.0, random_order=args.random_order, seed=1234), 2),
                            workload="the default frame with -m 0 (no marking: every main pixel is processed)")
        # Two frames in flight (never `value`): sequences and AOV passes are independent frames, and the distance kernels of one frame
        # fill the chip under the latency-bound tail of another.  Two engine contexts, each with the headline frame in flight on its own worker thread
        # (bcd_hip_denoise_begin / _wait); a context starts its next frame as soon as its last one is complete, so the two drift out of phase.
        ctx_b = bh.Context(local_rank)
        out_b = torch.empty_like(out)
        ctx_b.denoise(*d_in, S, prm, out_b)          # (grows the second context's workspaces; also the reference result)
        ctx_b.denoise(*d_in, S, prm, out_b)
        ref2 = out_b.clone()
        torch.cuda.synchronize()
        pairs = 8
        t1 = time.perf_counter()
        ctx.denoise_begin(*d_in, S, prm, out)
        ctx_b.denoise_begin(*d_in, S, prm, out_b)
        for _ in range(pairs - 1):
            ctx.denoise_wait()
            ctx.denoise_begin(*d_in, S, prm, out)
            ctx_b.denoise_wait()
            ctx_b.denoise_begin(*d_in, S, prm, out_b)
        ctx.denoise_wait()
        ctx_b.denoise_wait()
        torch.cuda.synchronize()
        ms_pipe = (time.perf_counter() - t1) * 1e3 / (2 * pairs)
        ok_fin = torch.isfinite(ref2)
        dev_pipe = max(float((torch.where(ok_fin, out - ref2, torch.zeros_like(out)).abs().max() / ref2[ok_fin].abs().max()).item()),
                       float((torch.where(ok_fin, out_b - ref2, torch.zeros_like(out)).abs().max() / ref2[ok_fin].abs().max()).item()))
        assert dev_pipe < 1e-5, "frames in flight differ from the blocking call: %g" % dev_pipe
        extras["pipelined_2frames"] = {"value": round(W * H / 1e6 / (ms_pipe * 1e-3), 3), "unit": "Mpix/s", "ms_per_step": round(ms_pipe, 4), "steps": 2 * pairs,
                                       "rel_linf_vs_blocking_call": dev_pipe,
                                       "workload": "the headline frame, two engine contexts with one frame in flight each (bcd_hip_denoise_begin / _wait): frames per "
                                                   "second of a sequence, not the latency of a frame -- never `value`"}
        ctx_b.close()
        del out_b, ref2
        # General sample counts (src/core/DenoisingUnit.cpp:371-383 takes any n1, n2): the headline frame at a uniform 24 spp (not
        # a power of two: the count products do not drop out) and with per-pixel counts drawn from {16, 24, 32, 48} (48-spp statistics thinned per
        # pixel: histogram and count scaled by 1/3, 1/2, 2/3 or 1 -- what an adaptive sampler's early exit leaves).  Both take the RATIO form of the distance
        # kernel (similarity_path 2 in per_scale).
        f24 = core.synthetic_scene(W, H, 24, 1234, args.sigma, args.spikes)
        extras["nonuniform_counts"] = {"uniform_24spp": leg(f24, prm, 3)}
        col48, ns48, hist48, cov48 = core.synthetic_scene(W, H, 48, 1234, args.sigma, args.spikes)
        keep = np.random.default_rng(5).choice(np.array([1.0 / 3.0, 0.5, 2.0 / 3.0, 1.0], np.float32), size=(H, W, 1)).astype(np.float32)
        ns_mix = np.ascontiguousarray(np.rint(ns48 * keep).astype(np.float32))
        hist_mix = np.ascontiguousarray(hist48 * (ns_mix / ns48))
        extras["nonuniform_counts"]["mixed_16_24_32_48"] = leg((col48, ns_mix, hist_mix, cov48), prm, 3)
        del col48, ns48, hist48, cov48, hist_mix, ns_mix, f24
        # BASELINE configs[1]'s frame (1280x720, 3-scale defaults): the small end of north_star's "720p-4K" range, untimed leg
        if (W, H) != (1280, 720):
            w7, h7 = 1280, 720
            d7 = [torch.from_numpy(a).cuda() for a in core.synthetic_scene(w7, h7, args.spp, 1234, args.sigma, args.spikes)]
            out7 = torch.empty((h7, w7, 3), dtype=torch.float32, device="cuda")
            for _ in range(3):   # (a new geometry: workspaces, the marking batch and the coarse scales' CU share settle over the first calls)
                ctx.denoise(*d7, S, prm, out7)
            torch.cuda.synchronize()
            each7 = []
            for _ in range(7):
                t2 = time.perf_counter()
                ctx.denoise(*d7, S, prm, out7)   # (a blocking call)
                each7.append(round((time.perf_counter() - t2) * 1e3, 3))
            ms7 = sorted(each7)[len(each7) // 2]   # the median call: one slow call of seven (seen once: 23 ms) is reported in ms_each, not averaged in
            extras["frame_720p"] = {"value": round(w7 * h7 / 1e6 / (ms7 * 1e-3), 3), "unit": "Mpix/s", "ms_per_step": round(ms7, 4), "steps": 7, "ms_each": each7,
                                    "workload": "1280x720 frame of the same generator and flags (BASELINE configs[1]), inputs resident; median of seven blocking calls"}
            del d7, out7
        # BASELINE configs[3]'s frame on this one GPU: the N = 1 point of the 4K strong-scaling curve (north_star), untimed leg
        if not args.no_4k:
            w4, h4 = 3840, 2160
            d4 = [torch.from_numpy(a).cuda() for a in core.synthetic_scene(w4, h4, args.spp, 1234, args.sigma, args.spikes)]
            out4 = torch.empty((h4, w4, 3), dtype=torch.float32, device="cuda")
            for _ in range(3):   # (a new geometry: workspaces, the marking batch and the coarse scales' CU share settle over the first calls -- tools/exp_soak.py)
                ctx.denoise(*d4, S, prm, out4)
            torch.cuda.synchronize()
            each4 = []
            t1 = time.perf_counter()
            for _ in range(5):
                t2 = time.perf_counter()
                ctx.denoise(*d4, S, prm, out4)   # (a blocking call)
                each4.append(round((time.perf_counter() - t2) * 1e3, 3))
            torch.cuda.synchronize()
            ms4 = (time.perf_counter() - t1) * 1e3 / 5
            frame4k_ms = ms4
            extras["frame_4k"] = {"value": round(w4 * h4 / 1e6 / (ms4 * 1e-3), 3), "unit": "Mpix/s", "ms_per_step": round(ms4, 4), "steps": 5, "ms_each": each4,
                                  "workload": "3840x2160 frame of the same generator and flags (BASELINE configs[3]), inputs resident"}
            # BASELINE configs[4] on this one GPU: large search window, spike prefilter (a step of its own, on the resident copies:
            # src/cli/main.cpp:428-441), random order
            prm12 = bh.default_params(b=12, w=w, m=args.skip_prob, random_order=1, seed=1234)
            ctx.denoise(*ctx.spike_filter(*d4, 2.0), S, prm12, out4)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                ctx.denoise(*ctx.spike_filter(*d4, 2.0), S, prm12, out4)
            torch.cuda.synchronize()
            ms12 = (time.perf_counter() - t1) * 1e3 / 2
            extras["frame_4k_b12_prefilter"] = {"value": round(w4 * h4 / 1e6 / (ms12 * 1e-3), 3), "unit": "Mpix/s", "ms_per_step": round(ms12, 4), "steps": 2,
                                                "workload": "the 3840x2160 frame with -b 12 -p 1 --p-factor 2 -r 1 (BASELINE configs[4]); the prefilter kernel is inside the timed step"}
            del d4, out4
    all_ms, all_launches = ctx.kernel_time()  # every launch of this process, warm-up and untimed legs included
    algo_bytes_per_step = ALGO_READ_BYTES_PER_PIXEL * sum(sc["w"] * sc["h"] for sc in scales)
    achieved = (algo_bytes_per_step * args.steps / (pd_ms * 1e-3)) / 1e9 if pd_ms > 0 else 0.0
    fast = all(sc["borderline_pairs"] is not None for sc in scales) or not single
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "traffic_source": None,
                "kernel": "k_pairdist_rw<60> (approximate planes; borderline pairs re-evaluated exactly by k_verify_pairs)" if fast else "k_pairdist<60>",
                "launches": pd_launches,
                "avg_launch_ms": round(pd_ms / max(1, pd_launches), 4),
                "whole_process": {"launches": all_launches, "avg_launch_ms": round(all_ms / max(1, all_launches), 4),
                                  },
                "algorithmic_bytes_per_launch_avg": int(algo_bytes_per_step / S),
                "isolated_avg_launch_ms": None if iso_ms is None else round(iso_ms, 4),
                "isolated_frac": None if not iso_ms else round((algo_bytes_per_step / S) / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
    roofline["valu"] = valu
    # numbers taken offline (rocprofv3 --pmc passes) are only quoted while the kernel source they were taken on is the one that runs
    src_hash = source_hash(PAIRDIST_SOURCE)
    roofline["kernel_source_sha256_16"] = src_hash
    traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(traffic_file):
        try:
            tr = json.load(open(traffic_file))
            key = "%dx%d_s%d" % (W, H, S)
            if key in tr:
                taken_on = tr[key].get("kernel_source_sha256_16")
                if taken_on == src_hash:
                    roofline["traffic"] = tr[key]["hbm_bytes_per_launch_avg"]
                    roofline["traffic_source"] = "profiled offline: tools/pmc_traffic.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload), profiles/pmc_traffic.json"
                else:
                    roofline["traffic_source"] = "null: profiles/pmc_traffic.json was taken on kernel source %s, this run is %s -- rerun tools/pmc_traffic.sh" % (taken_on, src_hash)
        except Exception:
            pass

    # what actually bounds the kernel: the vector pipe, from the committed SQ counter table of the same kernel (rocprofv3 --pmc; offline)
    try:
        import glob
        import re
        tabs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq_counters.txt")))
        txt = open(tabs[-1]).read()
        m = re.search(r"k_pairdist_rw.*?VALU pipe busy ([0-9.]+) %", txt, re.S)
        mh = re.search(r"k_similarity_fast\.hip sha256_16=([0-9a-f]+)", txt)
        if m and (not mh or mh.group(1) != src_hash):
            roofline["vector_pipe_busy_profiled"] = {"frac": None, "source": "null: %s was taken on kernel source %s, this run is %s -- rerun tools/refresh_profiles.sh" %
                                                     ("profiles/" + os.path.basename(tabs[-1]), mh.group(1) if mh else "unknown", src_hash)}
        elif m:
            roofline["vector_pipe_busy_profiled"] = {"frac": round(float(m.group(1)) / 100.0, 3), "source": "profiles/" + os.path.basename(tabs[-1]) +
                                                     " (SQ_ACTIVE_INST_VALU / (8 x SQ_BUSY_CYCLES) per shader engine, 1280x720 scale 0 alone)"}
    except Exception:
        pass

    if rank == 0:
        # The line ends with what a reader of its last 2 000 characters must find: `roofline` (compact), `cpu_baseline` (compact) and `legs` (bare
        # [Mpix/s, ms per frame] pairs of every untimed leg).  Everything wordy -- per-scale counters of the legs, the distance kernel's counters, the
        # band prediction's message sizes -- sits in `details` before them; what the fields mean is written down in DESIGN.md 8 ("The bench line").
        details = {"roofline": roofline}
        legs = {}

        def pair(o):
            return [o.get("value"), o.get("ms_per_step", o.get("ms_per_frame"))]
        for name, o in extras.items():
            if name == "nonuniform_counts":
                for sub, oo in o.items():
                    legs["nonuniform_" + sub] = pair(oo)
            else:
                legs[name] = pair(o)
        details["legs"] = extras
        rccl = bh.rccl_info()
        if world > 1:
            # the band driver must talk to the RCCL copy this process already holds (torch is imported first: same SONAME, one mapping -- DESIGN.md 7)
            assert rccl["one_copy"] and rccl["native_is_mapped"], "two RCCL copies in one process: %r" % (rccl,)
        res = {
            "metric": "Mpixels/sec denoised (3-scale, b=6, w=1)", "value": round(W * H / 1e6 / (ms_step * 1e-3), 3), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%dx%d synthetic frame (%s%d spp, sigma %.2f, spikes %.2f), %d-scale, b=%d w=1 d=1 e=1e-8, -m %g -r %d (seeded), no prefilter"
                                   % (W, H, "texture pattern, " if args.pattern else "", args.spp, args.sigma, args.spikes, S, b, args.skip_prob, args.random_order),
                       "parallelism": parallelism, "per_scale": scales},
            "details": details,
        }
        if not single:
            roofline.update({"achieved": None, "frac": None, "launches": None, "avg_launch_ms": None})
            roofline.pop("whole_process", None)
        if single and not (args.no_extras or args.no_predict):
            pred = predicted_8gpu(args, frame4k_ms)
            details["predicted_8gpu"] = pred
            legs["predicted_8gpu"] = [pred.get("speedup_before_waiting"), pred.get("band_ms")] if "error" not in pred else ["error", None]
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args)
            details["cpu_baseline"] = cpu
        v = roofline.get("valu") or {}
        busy = roofline.get("vector_pipe_busy_profiled") or {}
        res["rccl"] = {"native": rccl["native"], "one_copy": rccl["one_copy"]}
        res["roofline"] = {"bound": "hbm", "achieved": roofline["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": roofline["frac"], "traffic": roofline["traffic"],
                           "kernel": "k_pairdist_rw<60>" if fast else "k_pairdist<60>", "avg_launch_ms": roofline["avg_launch_ms"],
                           "isolated_avg_launch_ms": roofline["isolated_avg_launch_ms"], "isolated_frac": roofline["isolated_frac"],
                           "valu": {"achieved": v.get("achieved_tflops"), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": v.get("frac"), "pipe_busy": busy.get("frac")},
                           "kernel_source_sha256_16": roofline["kernel_source_sha256_16"]}
        if cpu is not None:
            res["cpu_baseline"] = {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                                   "sample": "%s, %d-scale, best of 3 on all cores" % (cpu["sample"].split(" synthetic")[0], S), "one_core": cpu["one_core"]["value"]}
        # north_star quotes its strong-scaling target on the 4K frame (BASELINE configs[3]): the same frame over the same ranks, readable without
        # knowing the leg names (`value` stays the 1080p frame BASELINE's metric is quoted on, at every N)
        if "frame_4k" in extras:
            res["scaling_frame"] = "3840x2160"
            res["value_4k"] = extras["frame_4k"]["value"]
            res["ms_per_step_4k"] = extras["frame_4k"]["ms_per_step"]
        if not single:
            res["band_check"] = band_check   # rank 0's equality check of the gathered band frame against a single-GPU run, before the timed region
        res["legs"] = legs
        # native libraries (RCCL's version banner) write to the C stdio buffer: flush it first so that the JSON line is the last line
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(res), flush=True)
    if dist is not None:
        if args.watchdog > 0:
            faulthandler.cancel_dump_traceback_later()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
