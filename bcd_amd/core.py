"""ctypes binding of the plain-C exports of libbcdcore.so (bcd_amd/host/capi.cpp): synthetic scenes, the
SamplesAccumulator, and the C++ bcd::Denoiser / bcd::MultiscaleDenoiser classes.  Plumbing for bench.py and tests."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libbcdcore.so")
_F = C.POINTER(C.c_float)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libbcdcore.so is not built (%s): run `python -m bcd_amd.build`" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_F)


def synthetic_scene(W, H, spp=32, seed=1234, sigma=0.35, spike_prob=0.01, first_line=0, nb_lines=None, pattern=0):
    """(colors, nsamples, histograms, covariances) of lines [first_line, first_line+nb_lines) of a W x H frame;
    pattern 0 = ramps + 16-pixel checker (SURVEY 8d probe scene), 1 = band-limited texture with oblique soft edges"""
    n = H - first_line if nb_lines is None else nb_lines
    ns = np.empty((n, W, 1), np.float32)
    mean = np.empty((n, W, 3), np.float32)
    cov = np.empty((n, W, 6), np.float32)
    hist = np.empty((n, W, 60), np.float32)
    rc = lib().bcdcore_synthetic_scene_ex(W, H, spp, C.c_uint(seed), C.c_float(sigma), C.c_float(spike_prob), int(pattern), first_line, n,
                                          _fp(ns), _fp(mean), _fp(cov), _fp(hist))
    if rc != 0:
        raise ValueError("bcdcore_synthetic_scene rc=%d" % rc)
    return mean, ns, hist, cov


def accumulate(samples, W, H, nbins=20, gamma=2.2, maxval=2.5):
    samples = np.ascontiguousarray(samples, np.float32)
    ns = np.empty((H, W, 1), np.float32)
    mean = np.empty((H, W, 3), np.float32)
    cov = np.empty((H, W, 6), np.float32)
    hist = np.empty((H, W, 3 * nbins), np.float32)
    lib().bcdcore_accumulate(_fp(samples), C.c_longlong(samples.shape[0]), W, H, nbins, C.c_float(gamma), C.c_float(maxval),
                             _fp(ns), _fp(mean), _fp(cov), _fp(hist))
    return ns, mean, cov, hist


def denoise(col, ns, hist, cov, nscales=1, tau=1.0, w=1, b=6, min_eig=1e-8, random_order=True, m=1.0, seed=1234, hist_width_override=0,
            use_cuda=True, devices=None, prefilter_factor=0.0):
    """bcd::Denoiser / bcd::MultiscaleDenoiser via IDenoiser; returns (ok, out, progress_monotone).
    use_cuda -> DenoiserParameters::m_useCuda, devices -> setDevices, prefilter_factor -> setSpikePrefilter"""
    H, W, D = hist.shape
    out = np.zeros((H, W, 3), np.float32)
    p = lambda a: None if a is None else _fp(a)
    devs = (C.c_int * len(devices))(*devices) if devices else None
    rc = lib().bcdcore_denoise_ex(p(col), p(ns), p(hist), p(cov), W, H, D, nscales, C.c_float(tau), w, b, C.c_float(min_eig),
                                  1 if random_order else 0, C.c_float(m), C.c_uint(seed), _fp(out), int(hist_width_override),
                                  1 if use_cuda else 0, devs, len(devices) if devices else 0, C.c_float(prefilter_factor))
    return rc != 0, out, rc == 1


def denoise_reuse(col, ns, hist, cov, nscales=3, b=6, nb_of_cores=0):
    """one IDenoiser object, denoise() twice with -r 0: (ok, first output, second output, (m_nbOfCores after call 1, after call 2))"""
    H, W, D = hist.shape
    o1, o2 = np.zeros((H, W, 3), np.float32), np.zeros((H, W, 3), np.float32)
    after = (C.c_int * 2)()
    rc = lib().bcdcore_denoise_reuse(_fp(col), _fp(ns), _fp(hist), _fp(cov), W, H, D, nscales, b, nb_of_cores, _fp(o1), _fp(o2), after)
    return rc != 0, o1, o2, (after[0], after[1])


def last_nb_of_cores():
    """DenoiserParameters::m_nbOfCores after the last denoise() (the reference writes the actual thread count back)"""
    return lib().bcdcore_last_nb_of_cores()


def release_engines():
    """bcd::releaseEngines(): destroy the cached engine contexts (device workspaces) of libbcdcore"""
    lib().bcdcore_release_engines()


def accumulate_threadsafe(samples, W, H, threads=4, nbins=20, gamma=2.2, maxval=2.5):
    """SamplesAccumulatorThreadSafe::addSampleThreadSafely from `threads` OpenMP threads"""
    samples = np.ascontiguousarray(samples, np.float32)
    ns = np.empty((H, W, 1), np.float32)
    mean = np.empty((H, W, 3), np.float32)
    cov = np.empty((H, W, 6), np.float32)
    hist = np.empty((H, W, 3 * nbins), np.float32)
    lib().bcdcore_accumulate_threadsafe(_fp(samples), C.c_longlong(samples.shape[0]), W, H, nbins, C.c_float(gamma), C.c_float(maxval), threads,
                                        _fp(ns), _fp(mean), _fp(cov), _fp(hist))
    return ns, mean, cov, hist


def spike_filter(col, ns, hist, cov, factor=2.0):
    H, W, D = hist.shape
    c, n, h, v = col.copy(), ns.copy(), hist.copy(), cov.copy()
    lib().bcdcore_spike_filter(_fp(c), _fp(n), _fp(h), _fp(v), W, H, D, C.c_float(factor))
    return c, n, h, v


def spike_filter_host(col, ns, hist, cov, factor=2.0):
    """SpikeRemovalFilter::filterOnHost (the loops filter() runs when no HIP device is usable)"""
    H, W, D = hist.shape
    c, n, h, v = col.copy(), ns.copy(), hist.copy(), cov.copy()
    lib().bcdcore_spike_filter_host(_fp(c), _fp(n), _fp(h), _fp(v), W, H, D, C.c_float(factor))
    return c, n, h, v


def merge_hist_ns(hist, ns):
    H, W, D = hist.shape
    out = np.empty((H, W, D + 1), np.float32)
    lib().bcdcore_merge_hist_ns(_fp(hist), _fp(ns), W, H, D, _fp(out))
    return out


def split_hist_ns(merged):
    H, W, D1 = merged.shape
    hist = np.empty((H, W, D1 - 1), np.float32)
    ns = np.empty((H, W, 1), np.float32)
    rc = lib().bcdcore_split_hist_ns(_fp(merged), W, H, D1, _fp(hist), _fp(ns))
    return (hist, ns) if rc == 0 else None


def write_exr(path, img, multi_channels):
    H, W, D = img.shape
    rc = lib().bcdcore_write_exr(path.encode(), _fp(np.ascontiguousarray(img, np.float32)), W, H, D, 1 if multi_channels else 0)
    if rc != 0:
        lib().bcdcore_exr_last_error.restype = C.c_char_p
        raise IOError(lib().bcdcore_exr_last_error().decode())


def read_exr(path, multi_channels):
    W, H, D = C.c_int(), C.c_int(), C.c_int()
    lib().bcdcore_exr_last_error.restype = C.c_char_p
    if lib().bcdcore_read_exr(path.encode(), 1 if multi_channels else 0, C.byref(W), C.byref(H), C.byref(D), None, C.c_longlong(0)) != 0:
        raise IOError(lib().bcdcore_exr_last_error().decode())
    out = np.empty((H.value, W.value, D.value), np.float32)
    rc = lib().bcdcore_read_exr(path.encode(), 1 if multi_channels else 0, C.byref(W), C.byref(H), C.byref(D), _fp(out), C.c_longlong(out.size))
    if rc != 0:
        raise IOError("read_exr rc=%d" % rc)
    return out
