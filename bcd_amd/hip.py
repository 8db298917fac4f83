"""ctypes binding of include/bcd_hip.h.  Fails loudly when libbcd_hip.so is missing: there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BCD_HIP_LIB") or os.path.join(_HERE, "lib", "libbcd_hip.so")   # (BCD_HIP_LIB: tools load instrumented builds)

_F = C.POINTER(C.c_float)
_VP = C.c_void_p


class Params(C.Structure):
    """bcd_hip_params (mirrors bcd::DenoiserParameters, include/bcd/core/IDenoiser.h:20-44 of the reference)"""
    _fields_ = [("hist_dist_threshold", C.c_float), ("patch_radius", C.c_int32), ("search_radius", C.c_int32),
                ("min_eigen_value", C.c_float), ("use_random_pixel_order", C.c_int32),
                ("marked_skip_probability", C.c_float), ("order_seed", C.c_uint32)]


class BandJob(C.Structure):
    _fields_ = [("d_colors", C.c_void_p), ("d_nsamples", C.c_void_p), ("d_histograms", C.c_void_p), ("d_covariances", C.c_void_p),
                ("W", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("main_row_begin", C.c_int32), ("main_row_end", C.c_int32),
                ("order_seed", C.c_uint32), ("d_sum", C.c_void_p), ("d_count", C.c_void_p)]


class HostOptions(C.Structure):
    _fields_ = [("spike_factor", C.c_float), ("zero_bad_values", C.c_int32)]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_float, C.c_void_p)


class ScaleStats(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("main_pixels", C.c_int64), ("processed", C.c_int64),
                ("fallback", C.c_int64), ("similar_total", C.c_int64), ("active_rounds", C.c_int32),
                ("ms_similarity", C.c_float), ("ms_active", C.c_float), ("ms_bayes", C.c_float), ("ms_total", C.c_float),
                ("similarity_path", C.c_int32), ("borderline_pairs", C.c_int32), ("cu_share", C.c_int32), ("spectral_inverses", C.c_int32)]


# every symbol include/bcd_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "bcd_hip_ctx_create", "bcd_hip_ctx_destroy", "bcd_hip_last_error", "bcd_hip_device_count", "bcd_hip_default_params",
    "bcd_hip_set_profiling", "bcd_hip_set_concurrent_scales", "bcd_hip_set_fast_similarity", "bcd_hip_set_strict_eigensolver", "bcd_hip_set_cu_share", "bcd_hip_get_stats", "bcd_hip_kernel_time", "bcd_hip_reset_kernel_time",
    "bcd_hip_denoise", "bcd_hip_denoise_begin", "bcd_hip_denoise_wait", "bcd_hip_denoise_band", "bcd_hip_denoise_bands", "bcd_hip_denoise_host", "bcd_hip_denoise_host_ex", "bcd_hip_last_upload_bytes", "bcd_hip_selftest_pack32", "bcd_hip_set_progress_callback",
    "bcd_hip_multi_create", "bcd_hip_multi_destroy", "bcd_hip_multi_last_error", "bcd_hip_multi_get_stats", "bcd_hip_multi_set_progress_callback", "bcd_hip_multi_set_frame_timeout", "bcd_hip_multi_set_comm_trace", "bcd_hip_multi_get_comm_trace", "bcd_hip_multi_denoise_host",
    "bcd_hip_multi_unique_id", "bcd_hip_multi_rccl_info", "bcd_hip_multi_create_rank", "bcd_hip_multi_rank_configure", "bcd_hip_multi_rank_upload", "bcd_hip_multi_rank_step",
    "bcd_hip_multi_rank_download", "bcd_hip_multi_rank_renew_ids", "bcd_hip_multi_set_loopback", "bcd_hip_multi_selftest_transport",
    "bcd_hip_scale_begin", "bcd_hip_pixel_cov", "bcd_hip_similarity_masks", "bcd_hip_similarity_masks_deferred", "bcd_hip_similarity_masks_verdict", "bcd_hip_similarity_masks_exact", "bcd_hip_window_distances", "bcd_hip_active_set", "bcd_hip_active_init", "bcd_hip_active_step", "bcd_hip_active_step_enqueue", "bcd_hip_active_step_collect",
    "bcd_hip_bayes_accumulate", "bcd_hip_bayes_accumulate_rows", "bcd_hip_finalize", "bcd_hip_finalize_band", "bcd_hip_downscale_sum", "bcd_hip_downscale_avg",
    "bcd_hip_downscale_cov", "bcd_hip_interpolate", "bcd_hip_merge", "bcd_hip_spike_filter", "bcd_hip_accumulate_samples", "bcd_hip_zero_bad_values",
    "bcd_hip_visit_order", "bcd_hip_scale_seed", "bcd_hip_strip_order_seed", "bcd_hip_selftest_division", "bcd_hip_selftest_distance_kernels", "bcd_hip_selftest_approx_distance", "bcd_hip_selftest_bin_work", "bcd_hip_eig27_batch",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libbcd_hip.so is not built (%s): run `python -m bcd_amd.build`; "
                               "there is no CPU fallback" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.bcd_hip_last_error.restype = C.c_char_p
        _lib.bcd_hip_last_error.argtypes = [_VP]
        _lib.bcd_hip_scale_seed.restype = C.c_uint32
        _lib.bcd_hip_strip_order_seed.restype = C.c_uint32
        _lib.bcd_hip_ctx_create.argtypes = [C.POINTER(_VP), C.c_int, _VP]
        _lib.bcd_hip_ctx_destroy.argtypes = [_VP]
        _lib.bcd_hip_ctx_destroy.restype = None
    return _lib


def default_params(**kw):
    p = Params()
    lib().bcd_hip_default_params(C.byref(p))
    names = {"tau": "hist_dist_threshold", "w": "patch_radius", "b": "search_radius", "min_eig": "min_eigen_value",
             "random_order": "use_random_pixel_order", "m": "marked_skip_probability", "seed": "order_seed"}
    for k, v in kw.items():
        setattr(p, names.get(k, k), v)
    return p


class BcdHipError(RuntimeError):
    pass


def _dp(t):
    """device pointer of a contiguous torch tensor"""
    assert t.is_cuda and t.is_contiguous(), "expected a contiguous device tensor"
    return C.c_void_p(t.data_ptr())


class Context:
    """bcd_hip_ctx bound to a torch device/stream."""

    def __init__(self, device=0, stream=None):
        import torch
        self.torch = torch
        self.device = device
        h = _VP()
        st = None
        if stream is not None:
            st = C.c_void_p(stream.cuda_stream)
        rc = lib().bcd_hip_ctx_create(C.byref(h), int(device), st)
        if rc != 0:
            raise BcdHipError("bcd_hip_ctx_create failed: rc=%d" % rc)
        self.h = h

    def close(self):
        if self.h:
            lib().bcd_hip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise BcdHipError("rc=%d: %s" % (rc, lib().bcd_hip_last_error(self.h).decode()))

    # ---- whole path
    def denoise(self, col, ns, hist, cov, nscales, prm, out=None):
        torch = self.torch
        H, W, D = hist.shape
        if out is None:
            out = torch.empty((H, W, 3), dtype=torch.float32, device=hist.device)
        self._chk(lib().bcd_hip_denoise(self.h, _dp(col), _dp(ns), _dp(hist), _dp(cov), W, H, D, nscales, C.byref(prm), _dp(out)))
        return out

    def denoise_begin(self, col, ns, hist, cov, nscales, prm, out):
        """bcd_hip_denoise_begin: returns at once; the tensors must stay alive and untouched until denoise_wait()"""
        H, W, D = hist.shape
        self._chk(lib().bcd_hip_denoise_begin(self.h, _dp(col), _dp(ns), _dp(hist), _dp(cov), W, H, D, nscales, C.byref(prm), _dp(out)))
        self._inflight = (col, ns, hist, cov, out, prm)   # (kept alive until denoise_wait; a refused call leaves the frame in flight alone)

    def denoise_wait(self):
        self._chk(lib().bcd_hip_denoise_wait(self.h))
        out = self._inflight[4] if getattr(self, "_inflight", None) else None
        self._inflight = None
        return out

    def denoise_band(self, col, ns, hist, cov, row_begin, row_end, prm, seed, sum_, cnt):
        H, W, D = hist.shape
        self._chk(lib().bcd_hip_denoise_band(self.h, _dp(col), _dp(ns), _dp(hist), _dp(cov), W, H, D, row_begin, row_end,
                                             C.byref(prm), C.c_uint32(seed), _dp(sum_), _dp(cnt)))

    def denoise_bands(self, jobs, prm):
        """jobs: list of (col, ns, hist, cov, row_begin, row_end, seed, sum, cnt) device tensors; run concurrently"""
        arr = (BandJob * len(jobs))()
        for i, (col, ns, hist, cov, r0, r1, seed, s, c) in enumerate(jobs):
            H, W, D = hist.shape
            arr[i] = BandJob(_dp(col).value, _dp(ns).value, _dp(hist).value, _dp(cov).value, W, H, D, r0, r1, seed, _dp(s).value, _dp(c).value)
        self._chk(lib().bcd_hip_denoise_bands(self.h, arr, len(jobs), C.byref(prm)))

    def denoise_host(self, col, ns, hist, cov, nscales, prm, spike_factor=0.0, zero_bad_values=False):
        import numpy as np
        H, W, D = hist.shape
        out = np.empty((H, W, 3), np.float32)
        f = lambda a: a.ctypes.data_as(_F)
        opt = HostOptions(spike_factor, 1 if zero_bad_values else 0)
        self._chk(lib().bcd_hip_denoise_host_ex(self.h, f(col), f(ns), f(hist), f(cov), W, H, D, nscales, C.byref(prm), C.byref(opt), f(out)))
        return out

    def last_upload_bytes(self):
        """(bytes of the histogram image of the last denoise_host call, bytes of it that crossed PCIe)"""
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(lib().bcd_hip_last_upload_bytes(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_progress_callback(self, fn):
        """fn(progress) or None; the ctypes thunk is kept alive on the context"""
        self._progress = PROGRESS_FN(lambda v, user: fn(v)) if fn else C.cast(None, PROGRESS_FN)
        self._chk(lib().bcd_hip_set_progress_callback(self.h, self._progress, None))

    # ---- stages
    def pixel_cov(self, cov, ns):
        H, W, _ = cov.shape
        out = self.torch.empty_like(cov)
        self._chk(lib().bcd_hip_pixel_cov(self.h, _dp(cov), _dp(ns), W, H, _dp(out)))
        return out

    def similarity_masks(self, hist, ns, w, b, tau):
        torch = self.torch
        H, W, D = hist.shape
        words = ((2 * b + 1) ** 2 + 31) // 32
        mask = torch.zeros((H, W, words), dtype=torch.int32, device=hist.device)
        cnt = torch.zeros((H, W), dtype=torch.int32, device=hist.device)
        self._chk(lib().bcd_hip_similarity_masks(self.h, _dp(hist), _dp(ns), W, H, D, w, b, C.c_float(tau), _dp(mask), _dp(cnt)))
        return mask, cnt

    def window_distances(self, hist, ns, w, b, line, col):
        import numpy as np
        H, W, D = hist.shape
        out = np.empty(((2 * b + 1) ** 2,), np.float32)
        self._chk(lib().bcd_hip_window_distances(self.h, _dp(hist), _dp(ns), W, H, D, w, b, line, col, out.ctypes.data_as(_F)))
        return out

    def active_set(self, mask, cnt, w, b, m, random_order, seed, row_begin=0, row_end=None):
        torch = self.torch
        H, W, _ = mask.shape
        state = torch.zeros((H, W), dtype=torch.uint8, device=mask.device)
        rounds = C.c_int32(0)
        self._chk(lib().bcd_hip_active_set(self.h, _dp(mask), _dp(cnt), W, H, w, b, row_begin, H if row_end is None else row_end,
                                           C.c_float(m), int(random_order), C.c_uint32(seed), _dp(state), C.byref(rounds)))
        return state, rounds.value

    def active_init(self, cnt, w, row_begin, row_end, m, seed, row_offset, state=None):
        H, W = cnt.shape
        if state is None:
            state = self.torch.zeros((H, W), dtype=self.torch.uint8, device=cnt.device)
        self._chk(lib().bcd_hip_active_init(self.h, _dp(cnt), W, H, w, row_begin, row_end, C.c_float(m), C.c_uint32(seed), row_offset, _dp(state)))
        return state

    def active_step(self, mask, cnt, state, w, b, row_begin, row_end, random_order, seed, row_offset, first_pass):
        H, W = cnt.shape
        u = C.c_int32(0)
        self._chk(lib().bcd_hip_active_step(self.h, _dp(mask), _dp(cnt), W, H, w, b, row_begin, row_end, int(random_order), C.c_uint32(seed),
                                            row_offset, 1 if first_pass else 0, _dp(state), C.byref(u)))
        return u.value

    def bayes_accumulate(self, col, pixcov, mask, nsim, state, w, b, min_eig):
        torch = self.torch
        H, W, _ = col.shape
        s = torch.zeros((H, W, 3), dtype=torch.float32, device=col.device)
        c = torch.zeros((H, W), dtype=torch.int32, device=col.device)
        self._chk(lib().bcd_hip_bayes_accumulate(self.h, _dp(col), _dp(pixcov), _dp(mask), _dp(nsim), _dp(state), W, H, w, b,
                                                 C.c_float(min_eig), _dp(s), _dp(c)))
        return s, c

    def finalize(self, s, c):
        out = self.torch.empty_like(s)
        self._chk(lib().bcd_hip_finalize(self.h, _dp(s), _dp(c), C.c_int64(c.numel()), _dp(out)))
        return out

    def finalize_band(self, s, c, halo, up, down, out):
        """out (rows x W x 3 view) = finalisation of the accumulator rows s / c with the neighbours' halos (pairs or None) added"""
        rows, W, _ = s.shape
        z = C.c_void_p(0)
        self._chk(lib().bcd_hip_finalize_band(self.h, _dp(s), _dp(c), W, rows, halo, _dp(up[0]) if up else z, _dp(up[1]) if up else z,
                                              _dp(down[0]) if down else z, _dp(down[1]) if down else z, _dp(out)))
        return out

    def downscale_sum(self, a):
        H, W, D = a.shape
        o = self.torch.empty((H // 2, W // 2, D), dtype=a.dtype, device=a.device)
        self._chk(lib().bcd_hip_downscale_sum(self.h, _dp(a), W, H, D, _dp(o)))
        return o

    def downscale_avg(self, a):
        H, W, D = a.shape
        o = self.torch.empty((H // 2, W // 2, D), dtype=a.dtype, device=a.device)
        self._chk(lib().bcd_hip_downscale_avg(self.h, _dp(a), W, H, D, _dp(o)))
        return o

    def downscale_cov(self, cov, ns):
        H, W, D = cov.shape
        o = self.torch.empty((H // 2, W // 2, D), dtype=cov.dtype, device=cov.device)
        self._chk(lib().bcd_hip_downscale_cov(self.h, _dp(cov), _dp(ns), W, H, _dp(o)))
        return o

    def interpolate(self, lo, H, W):
        h, w, D = lo.shape
        o = self.torch.empty((H, W, D), dtype=lo.dtype, device=lo.device)
        self._chk(lib().bcd_hip_interpolate(self.h, _dp(lo), w, h, D, _dp(o), W, H))
        return o

    def merge(self, hi, lo):
        H, W, D = hi.shape
        o = hi.clone()
        self._chk(lib().bcd_hip_merge(self.h, _dp(o), W, H, _dp(lo), D))
        return o

    def merge_(self, hi, lo):
        """in place on hi (a contiguous H x W x D tensor or row-slice view)"""
        H, W, D = hi.shape
        self._chk(lib().bcd_hip_merge(self.h, _dp(hi), W, H, _dp(lo), D))
        return hi

    def spike_filter(self, col, ns, hist, cov, factor):
        H, W, D = hist.shape
        o = [self.torch.empty_like(t) for t in (col, ns, hist, cov)]
        self._chk(lib().bcd_hip_spike_filter(self.h, _dp(col), _dp(ns), _dp(hist), _dp(cov), W, H, D, C.c_float(factor),
                                             _dp(o[0]), _dp(o[1]), _dp(o[2]), _dp(o[3])))
        return o

    def accumulate_samples(self, samples, weights=None, nbins=20, gamma=2.2, maxval=2.5):
        """samples: (H, W, spp, 3) device tensor; weights: (H, W, spp) or None"""
        torch = self.torch
        H, W, spp, _ = samples.shape
        mk = lambda d: torch.empty((H, W, d), dtype=torch.float32, device=samples.device)
        ns, mean, cov, hist = mk(1), mk(3), mk(6), mk(3 * nbins)
        self._chk(lib().bcd_hip_accumulate_samples(self.h, _dp(samples), _dp(weights) if weights is not None else None, W, H, spp, nbins,
                                                   C.c_float(gamma), C.c_float(maxval), _dp(ns), _dp(mean), _dp(cov), _dp(hist)))
        return ns, mean, cov, hist

    def zero_bad_values(self, img):
        self._chk(lib().bcd_hip_zero_bad_values(self.h, _dp(img), C.c_int64(img.numel())))
        return img

    # ---- stats / timing
    def set_profiling(self, on):
        self._chk(lib().bcd_hip_set_profiling(self.h, 1 if on else 0))

    def set_concurrent_scales(self, on):
        self._chk(lib().bcd_hip_set_concurrent_scales(self.h, 1 if on else 0))

    def stats(self, scale):
        s = ScaleStats()
        self._chk(lib().bcd_hip_get_stats(self.h, scale, C.byref(s)))
        return s

    def kernel_time(self):
        ms, n = C.c_float(0), C.c_int32(0)
        self._chk(lib().bcd_hip_kernel_time(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def reset_kernel_time(self):
        self._chk(lib().bcd_hip_reset_kernel_time(self.h))

    def selftest_distance_kernels(self, hist, ns, b):
        """(variant, mismatching plane entries) of the production distance kernel against the exact general one on these inputs"""
        H, W, D = hist.shape
        v, n = C.c_int(0), C.c_int64(0)
        self._chk(lib().bcd_hip_selftest_distance_kernels(self.h, _dp(hist), _dp(ns), W, H, D, b, C.byref(v), C.byref(n)))
        return v.value, n.value

    def set_fast_similarity(self, on):
        self._chk(lib().bcd_hip_set_fast_similarity(self.h, 1 if on else 0))

    def set_cu_share(self, percent):
        self._chk(lib().bcd_hip_set_cu_share(self.h, int(percent)))

    def selftest_approx_distance(self, hist, ns, b):
        """(max relative deviation of a patch distance, pairs with different bin counts, flags) of the approximate distance planes
        against the exact ones on these inputs"""
        H, W, D = hist.shape
        r, n, f = C.c_float(0), C.c_int64(0), C.c_int(0)
        self._chk(lib().bcd_hip_selftest_approx_distance(self.h, _dp(hist), _dp(ns), W, H, D, b, C.byref(r), C.byref(n), C.byref(f)))
        return r.value, n.value, f.value

    def selftest_bin_work(self, hist, ns, b, reps=3):
        """-> (lane_bins, wave_bins, wave_groups, production kernel ms): the arithmetic of the distance kernel on this frame (counting instantiation)"""
        H, W, D = hist.shape
        a, b_, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        ms = C.c_float(0)
        self._chk(lib().bcd_hip_selftest_bin_work(self.h, _dp(hist), _dp(ns), W, H, D, int(b), int(reps), C.byref(a), C.byref(b_), C.byref(c), C.byref(ms)))
        return a.value, b_.value, c.value, ms.value

    def eig27_batch(self, A):
        """A: (n, 28, 28) symmetric device tensor (row / column 27 zero) -> (eigenvalues (n, 28), eigenvectors (n, 28, 28), kernel ms)"""
        torch = self.torch
        n = A.shape[0]
        eig = torch.zeros((n, 28), dtype=torch.float32, device=A.device)
        V = torch.zeros((n, 28, 28), dtype=torch.float32, device=A.device)
        ms = C.c_float(0)
        self._chk(lib().bcd_hip_eig27_batch(self.h, _dp(A), n, _dp(eig), _dp(V), C.byref(ms)))
        return eig, V, ms.value

    def selftest_division(self, samples, seed=1):
        n = C.c_int64(-1)
        self._chk(lib().bcd_hip_selftest_division(self.h, C.c_uint32(seed), C.c_int64(samples), C.byref(n)))
        return n.value

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)


class MultiStats(C.Structure):
    _fields_ = [("n_ranks", C.c_int32), ("transport", C.c_int32), ("frames", C.c_int64), ("marking_rounds", C.c_int32 * 8), ("compute_ms", C.c_float)]


class MultiDenoiser:
    """bcd_hip_multi: one frame over several GPUs (or several virtual ranks on one GPU); host buffers in and out"""

    def __init__(self, devices):
        h = _VP()
        arr = (C.c_int * len(devices))(*devices)
        lib().bcd_hip_multi_last_error.restype = C.c_char_p
        lib().bcd_hip_multi_last_error.argtypes = [_VP]
        lib().bcd_hip_multi_destroy.argtypes = [_VP]
        lib().bcd_hip_multi_destroy.restype = None
        rc = lib().bcd_hip_multi_create(C.byref(h), arr, len(devices))
        if rc != 0:
            raise BcdHipError("bcd_hip_multi_create failed: rc=%d" % rc)
        self.h = h

    def close(self):
        if self.h:
            lib().bcd_hip_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def denoise_host(self, col, ns, hist, cov, nscales, prm):
        import numpy as np
        H, W, D = hist.shape
        out = np.empty((H, W, 3), np.float32)
        f = lambda a: a.ctypes.data_as(_F)
        rc = lib().bcd_hip_multi_denoise_host(self.h, f(col), f(ns), f(hist), f(cov), W, H, D, nscales, C.byref(prm), f(out))
        if rc != 0:
            raise BcdHipError("rc=%d: %s" % (rc, lib().bcd_hip_multi_last_error(self.h).decode()))
        return out

    def stats(self):
        s = MultiStats()
        lib().bcd_hip_multi_get_stats(self.h, C.byref(s))
        return s

    def set_comm_trace(self, on):
        lib().bcd_hip_multi_set_comm_trace(self.h, 1 if on else 0)

    def set_frame_timeout(self, milliseconds):
        lib().bcd_hip_multi_set_frame_timeout(self.h, int(milliseconds))

    def comm_trace(self, rank):
        """[(channel, kind, bytes_up, bytes_down), ...] of the last frame, in the order `rank` enqueued its operations"""
        n = lib().bcd_hip_multi_get_comm_trace(self.h, rank, None, 0)
        buf = (C.c_int64 * max(1, n))()
        n = lib().bcd_hip_multi_get_comm_trace(self.h, rank, buf, n)
        v = list(buf[:n])
        return [tuple(v[i:i + 4]) for i in range(0, n, 4)]


MULTI_ID_BYTES = 128


def set_strict_eigensolver(on):
    """process-wide: the fully converged stopping rule of the estimate chain's eigensolver (bcd_hip_set_strict_eigensolver)"""
    lib().bcd_hip_set_strict_eigensolver(1 if on else 0)


def rccl_info():
    """{"native": what libbcd_hip.so's band driver is linked against at run time, "mapped": every librccl copy mapped into this process}"""
    buf = C.create_string_buffer(512)
    lib().bcd_hip_multi_rccl_info(buf, 512)
    mapped = []
    try:
        for ln in open("/proc/self/maps"):
            f = ln.split()
            if len(f) >= 6 and "librccl" in f[5] and f[5] not in mapped:
                mapped.append(f[5])
    except OSError:
        pass
    native = buf.value.decode()
    path = native.split(" from ", 1)[1] if " from " in native else "?"
    return {"native": native, "mapped": mapped, "one_copy": len(mapped) <= 1, "native_is_mapped": os.path.realpath(path) in [os.path.realpath(m) for m in mapped]}


def multi_unique_ids(n):
    """n RCCL unique ids (bytes), to be created by rank 0 and handed to every process (bcd_hip_multi_create_rank)"""
    out = b""
    for _ in range(n):
        buf = C.create_string_buffer(MULTI_ID_BYTES)
        rc = lib().bcd_hip_multi_unique_id(buf)
        if rc != 0:
            raise BcdHipError("bcd_hip_multi_unique_id rc=%d" % rc)
        out += buf.raw
    return out


class RankDenoiser:
    """one rank of the row-band partition in a one-process-per-GPU job (bcd_hip_multi_rank_*): the band's inputs and result stay in HBM"""

    def __init__(self, rank, world, device, ids):
        h = _VP()
        lib().bcd_hip_multi_last_error.restype = C.c_char_p
        lib().bcd_hip_multi_last_error.argtypes = [_VP]
        lib().bcd_hip_multi_destroy.argtypes = [_VP]
        lib().bcd_hip_multi_destroy.restype = None
        rc = lib().bcd_hip_multi_create_rank(C.byref(h), rank, world, device, ids, len(ids) // MULTI_ID_BYTES if ids else 0)
        if rc != 0:
            raise BcdHipError("bcd_hip_multi_create_rank failed: rc=%d" % rc)
        self.h, self.rank, self.world = h, rank, world

    def _chk(self, rc):
        if rc != 0:
            raise BcdHipError("rc=%d: %s" % (rc, lib().bcd_hip_multi_last_error(self.h).decode()))

    def configure(self, W, H, D, nscales, prm):
        """-> (first input line, input lines, first owned line, owned lines) of this rank's band"""
        v = [C.c_int(0) for _ in range(4)]
        self._chk(lib().bcd_hip_multi_rank_configure(self.h, W, H, D, nscales, C.byref(prm), *[C.byref(x) for x in v]))
        self.W, self.owned = W, (v[2].value, v[3].value)
        return tuple(x.value for x in v)

    def upload(self, col, ns, hist, cov):
        f = lambda a: a.ctypes.data_as(_F)
        self._chk(lib().bcd_hip_multi_rank_upload(self.h, f(col), f(ns), f(hist), f(cov)))

    def step(self):
        self._chk(lib().bcd_hip_multi_rank_step(self.h))

    def set_loopback(self, on=True):
        """one rank of one, exchanging with itself over real RCCL communicators (one-GPU test of the transport code)"""
        self._chk(lib().bcd_hip_multi_set_loopback(self.h, 1 if on else 0))

    def renew_ids(self, ids):
        self._chk(lib().bcd_hip_multi_rank_renew_ids(self.h, ids, len(ids) // MULTI_ID_BYTES))

    def stats(self):
        s = MultiStats()
        lib().bcd_hip_multi_get_stats(self.h, C.byref(s))
        return s

    def set_comm_trace(self, on):
        lib().bcd_hip_multi_set_comm_trace(self.h, 1 if on else 0)

    def comm_trace(self):
        n = lib().bcd_hip_multi_get_comm_trace(self.h, self.rank, None, 0)
        buf = (C.c_int64 * max(1, n))()
        n = lib().bcd_hip_multi_get_comm_trace(self.h, self.rank, buf, n)
        v = list(buf[:n])
        return [tuple(v[i:i + 4]) for i in range(0, n, 4)]

    def download(self):
        import numpy as np
        out = np.empty((self.owned[1], self.W, 3), np.float32)
        self._chk(lib().bcd_hip_multi_rank_download(self.h, out.ctypes.data_as(_F)))
        return out

    def close(self):
        if self.h:
            lib().bcd_hip_multi_destroy(self.h)
            self.h = None


def selftest_transport(device=0, halo_bytes=7 * 3840 * 16):
    """bcd_hip_multi_selftest_transport: (rc, report line)"""
    buf = C.create_string_buffer(512)
    rc = lib().bcd_hip_multi_selftest_transport(int(device), C.c_longlong(int(halo_bytes)), buf, 512)
    return rc, buf.value.decode()


def visit_order(W, H, w, random_order, seed):
    import numpy as np
    out = np.empty(((W - 2 * w) * (H - 2 * w),), np.int32)
    rc = lib().bcd_hip_visit_order(W, H, w, int(random_order), C.c_uint32(seed), out.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise BcdHipError("bcd_hip_visit_order rc=%d" % rc)
    return out


def scale_seed(seed0, scale):
    return int(lib().bcd_hip_scale_seed(C.c_uint32(seed0), int(scale)))


def strip_order_seed(W, H, w, b):
    """the `seed` argument of visit_order / the marking entry points for pixel order 2 (the reference's multi-thread -r 0 list: even
    strips of 2b lines, then the odd ones): it carries the frame geometry"""
    return int(lib().bcd_hip_strip_order_seed(int(W), int(H), int(w), int(b)))
