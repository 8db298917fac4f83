"""ctypes bindings for the CPU oracle (oracle/libbcd_oracle.so) and, when present, the compiled
reference translation units (oracle/_ref/libbcd_ref.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_FP = C.POINTER(C.c_float)


class BcdoParams(C.Structure):
    _fields_ = [("hist_dist_threshold", C.c_float), ("patch_radius", C.c_int), ("search_radius", C.c_int),
                ("min_eigen_value", C.c_float), ("skip_probability", C.c_float), ("nb_threads", C.c_int), ("skip_seed", C.c_uint32)]


class BcdoDiag(C.Structure):
    _fields_ = [("processed", C.POINTER(C.c_uint8)), ("fallback", C.POINTER(C.c_uint8)),
                ("nb_similar", C.POINTER(C.c_int32))]


def _fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_FP)


def build_oracle():
    so = os.path.join(ORACLE_DIR, "libbcd_oracle.so")
    src = os.path.join(ORACLE_DIR, "bcd_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libbcd_oracle.so"], stdout=subprocess.DEVNULL)
    return so


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        _oracle = C.CDLL(build_oracle())
        _oracle.bcdo_patch_distance.restype = C.c_float
    return _oracle


REF_SO = os.path.join(ORACLE_DIR, "_ref", "libbcd_ref.so")


def ref_available():
    """is the compiled reference present?  (a file test: nothing is loaded -- test modules call this at import time, and the GPU
    box must not map the compiled reference units just because a test file was collected)"""
    return os.path.exists(REF_SO)


def ref():
    """compiled reference TUs; None when oracle/_ref is absent (e.g. never built).  Loaded on first CALL only; used by the CPU
    tests that pin the oracle and by tests/golden/make_golden.py, never by a GPU test."""
    global _ref
    if _ref is None and os.path.exists(REF_SO):
        _ref = C.CDLL(REF_SO)
    return _ref


def params(tau=1.0, w=1, b=6, min_eig=1e-8, m=1.0, threads=0, skip_seed=0):
    return BcdoParams(tau, w, b, min_eig, m, threads, skip_seed)


# ---------------------------------------------------------------------------------------------
def denoise_mono(col, ns, hist, cov, prm, order=None, want_diag=False):
    H, W, D = hist.shape
    out = np.empty((H, W, 3), np.float32)
    diag = None
    arrs = None
    if want_diag:
        arrs = (np.zeros((H, W), np.uint8), np.zeros((H, W), np.uint8), np.zeros((H, W), np.int32))
        diag = BcdoDiag(arrs[0].ctypes.data_as(C.POINTER(C.c_uint8)), arrs[1].ctypes.data_as(C.POINTER(C.c_uint8)),
                        arrs[2].ctypes.data_as(C.POINTER(C.c_int32)))
    op, on = None, 0
    if order is not None:
        order = np.ascontiguousarray(order, np.int32)
        op, on = order.ctypes.data_as(C.POINTER(C.c_int32)), order.size
    rc = oracle().bcdo_denoise_mono(_fp(col), _fp(ns), _fp(hist), _fp(cov), W, H, D, C.byref(prm), op,
                                    C.c_int64(on), _fp(out), C.byref(diag) if diag else None)
    assert rc == 0, rc
    return (out, arrs) if want_diag else out


def denoise_multiscale(col, ns, hist, cov, nscales, prm, orders=None, racy=False):
    H, W, D = hist.shape
    out = np.empty((H, W, 3), np.float32)
    po, pn = None, None
    if orders is not None:
        keep = [np.ascontiguousarray(o, np.int32) for o in orders]
        po = (C.POINTER(C.c_int32) * nscales)(*[o.ctypes.data_as(C.POINTER(C.c_int32)) for o in keep])
        pn = (C.c_int64 * nscales)(*[o.size for o in keep])
    rc = oracle().bcdo_denoise_multiscale(_fp(col), _fp(ns), _fp(hist), _fp(cov), W, H, D, nscales, C.byref(prm),
                                          po, pn, _fp(out), 1 if racy else 0)
    assert rc == 0, rc
    return out


TRACE_FIELDS = ("noise", "x", "mean1", "cov1", "cov1_minus_noise", "clamped", "clamped_plus_noise", "inverse1", "step1", "mean2", "cov2",
                "inverse2", "step2")


class BcdoPatchTrace(C.Structure):
    _fields_ = [("members", C.POINTER(C.c_int32))] + [(k, _FP) for k in TRACE_FIELDS]


def patch_trace(col, ns, hist, cov, prm, pl, pc):
    """every intermediate of the two Bayesian steps for the main pixel (pl, pc): dict of arrays (SURVEY 8c fixture F3)"""
    H, W, D = hist.shape
    K = 3 * (2 * prm.patch_radius + 1) ** 2
    P = K // 3
    cap = (2 * prm.search_radius + 1) ** 2
    shapes = dict(noise=(P, 6), x=(cap, K), mean1=(K,), cov1=(K, K), cov1_minus_noise=(K, K), clamped=(K, K), clamped_plus_noise=(K, K),
                  inverse1=(K, K), step1=(cap, K), mean2=(K,), cov2=(K, K), inverse2=(K, K), step2=(cap, K))
    arrs = {k: np.zeros(shapes[k], np.float32) for k in TRACE_FIELDS}
    members = np.zeros(cap, np.int32)
    t = BcdoPatchTrace(members.ctypes.data_as(C.POINTER(C.c_int32)), *[_fp(arrs[k]) for k in TRACE_FIELDS])
    n = oracle().bcdo_patch_trace(_fp(col), _fp(ns), _fp(hist), _fp(cov), W, H, D, C.byref(prm), int(pl), int(pc), C.byref(t))
    assert n >= 0, n
    out = {k: (v[:n] if shapes[k][0] == cap else v) for k, v in arrs.items()}
    out["members"] = members[:n]
    return out


def similarity_masks(ns, hist, w, b, tau, threads=0):
    H, W, D = hist.shape
    side = 2 * b + 1
    words = (side * side + 31) // 32
    mask = np.zeros((H, W, words), np.uint32)
    cnt = np.zeros((H, W), np.int32)
    oracle().bcdo_similarity_masks(_fp(hist), _fp(ns), W, H, D, w, b, C.c_float(tau),
                                   mask.ctypes.data_as(C.POINTER(C.c_uint32)),
                                   cnt.ctypes.data_as(C.POINTER(C.c_int32)), threads)
    return mask, cnt


def window_distances(ns, hist, w, b, pl, pc):
    H, W, D = hist.shape
    out = np.empty(((2 * b + 1) ** 2,), np.float32)
    oracle().bcdo_window_distances(_fp(hist), _fp(ns), W, H, D, w, b, pl, pc, _fp(out))
    return out


def sym_eig(A):
    n = A.shape[0]
    A = np.ascontiguousarray(A, np.float32)
    ev = np.empty(n, np.float32)
    V = np.empty((n, n), np.float32)
    oracle().bcdo_sym_eig(n, _fp(A), _fp(ev), _fp(V))
    return ev, V


def _pyr(lib, prefix):
    def dsum(a):
        H, W, D = a.shape
        o = np.empty((H // 2, W // 2, D), np.float32)
        getattr(lib, prefix + "downscale_sum")(_fp(a), W, H, D, _fp(o))
        return o

    def davg(a):
        H, W, D = a.shape
        o = np.empty((H // 2, W // 2, D), np.float32)
        getattr(lib, prefix + "downscale_avg")(_fp(a), W, H, D, _fp(o))
        return o

    def dcov(cov, ns):
        H, W, D = cov.shape
        o = np.empty((H // 2, W // 2, D), np.float32)
        getattr(lib, prefix + "downscale_cov")(_fp(cov), _fp(ns), W, H, D, _fp(o))
        return o

    def interp(lo, H, W):
        h, w, D = lo.shape
        o = np.empty((H, W, D), np.float32)
        getattr(lib, prefix + "interpolate")(_fp(lo), w, h, D, _fp(o), W, H)
        return o

    def merge(hi, lo):
        H, W, D = hi.shape
        o = hi.copy()
        getattr(lib, prefix + "merge")(_fp(o), W, H, _fp(lo), D)
        return o

    def spike(col, ns, hist, cov, factor):
        H, W, D = hist.shape
        c, n, h, v = col.copy(), ns.copy(), hist.copy(), cov.copy()
        getattr(lib, prefix + "spike_filter")(_fp(c), _fp(n), _fp(h), _fp(v), W, H, D, C.c_float(factor))
        return c, n, h, v

    def accumulate(samples, W, H, nbins=20, gamma=2.2, maxval=2.5):
        samples = np.ascontiguousarray(samples, np.float32)
        ns = np.empty((H, W, 1), np.float32)
        mean = np.empty((H, W, 3), np.float32)
        cov = np.empty((H, W, 6), np.float32)
        hist = np.empty((H, W, 3 * nbins), np.float32)
        getattr(lib, prefix + "accumulate")(_fp(samples), C.c_int64(samples.shape[0]), W, H, nbins,
                                            C.c_float(gamma), C.c_float(maxval), _fp(ns), _fp(mean), _fp(cov), _fp(hist))
        return ns, mean, cov, hist

    return dict(dsum=dsum, davg=davg, dcov=dcov, interp=interp, merge=merge, spike=spike, accumulate=accumulate)


def oracle_ops():
    return _pyr(oracle(), "bcdo_")


def ref_ops():
    return _pyr(ref(), "bcdref_") if ref() is not None else None


# ---------------------------------------------------------------------------------------------
def synth_samples(W, H, spp, seed=1234, sigma=0.35, spike_prob=0.01):
    """seeded per-sample radiance model: smooth ramps + checker, multiplicative noise,
    rare spikes.  Returns n x 6 float32 (line, col, r, g, b, weight) and the noise-free base image."""
    rng = np.random.default_rng(seed)
    l, c = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    checker = ((l // 16 + c // 16) % 2).astype(np.float32)
    base = np.stack([0.2 + 0.6 * c / W, 0.5 + 0.4 * np.sin(12.0 * l / H), np.where(checker > 0, 0.8, 0.15)], -1).astype(np.float32)
    noise = rng.standard_normal((H, W, spp, 3)).astype(np.float32)
    s = base[:, :, None, :] * (1.0 + sigma * noise)
    if spike_prob > 0:
        sp = rng.random((H, W, spp, 1)) < spike_prob
        s = s + sp * 4.0 * rng.random((H, W, spp, 3))
    s = np.maximum(s, 0).astype(np.float32)
    ll = np.broadcast_to(l[:, :, None], (H, W, spp)).astype(np.float32)
    cc = np.broadcast_to(c[:, :, None], (H, W, spp)).astype(np.float32)
    samples = np.concatenate([ll[..., None], cc[..., None], s, np.ones((H, W, spp, 1), np.float32)], -1)
    return np.ascontiguousarray(samples.reshape(-1, 6)), base


def synth_inputs(W, H, spp=32, seed=1234, sigma=0.35, spike_prob=0.01):
    samples, base = synth_samples(W, H, spp, seed, sigma, spike_prob)
    ns, mean, cov, hist = oracle_ops()["accumulate"](samples, W, H)
    return mean, ns, hist, cov, base
