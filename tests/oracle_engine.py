"""CPU stand-in for bcd_amd.tiling.HipEngine backed by the oracle -- lets the band orchestration (partition, halo
exchange, per-band pyramid, merge at band edges) run on CPU tensors, e.g. under gloo.  TEST INFRASTRUCTURE."""
import ctypes as C

import numpy as np
import torch

import oracle_lib as ol


def _np(t):
    return np.ascontiguousarray(t.numpy(), np.float32)


class OracleEngine:
    torch = torch

    def __init__(self, visit_order):
        self.visit_order = visit_order  # callable (W, H, w, random, seed) -> order, or None for scanline

    def scale_seed(self, seed0, s):
        return seed0 + s

    def downscale_avg(self, t):
        return torch.from_numpy(ol.oracle_ops()["davg"](_np(t)))

    def downscale_sum(self, t):
        return torch.from_numpy(ol.oracle_ops()["dsum"](_np(t)))

    def downscale_cov(self, cov, ns):
        return torch.from_numpy(ol.oracle_ops()["dcov"](_np(cov), _np(ns)))

    def accumulate_band(self, col, ns, hist, cov, row0, row1, prm, seed, scale):
        H, W, D = hist.shape
        s = np.empty((H, W, 3), np.float32)
        c = np.empty((H, W), np.int32)
        op = ol.params(prm.hist_dist_threshold, prm.patch_radius, prm.search_radius, prm.min_eigen_value, prm.marked_skip_probability)
        order, on = None, 0
        if prm.marked_skip_probability != 0.0 and self.visit_order is not None:
            o = np.ascontiguousarray(self.visit_order(W, H, prm.patch_radius, prm.use_random_pixel_order, seed), np.int32)
            self._keep = o
            order, on = o.ctypes.data_as(C.POINTER(C.c_int32)), o.size
        rc = ol.oracle().bcdo_accumulate_band(ol._fp(_np(col)), ol._fp(_np(ns)), ol._fp(_np(hist)), ol._fp(_np(cov)), W, H, D,
                                              C.byref(op), row0, row1, order, C.c_int64(on), ol._fp(s),
                                              c.ctypes.data_as(C.POINTER(C.c_int32)))
        assert rc == 0
        return torch.from_numpy(s), torch.from_numpy(c)

    def accumulate_bands(self, jobs, prm):
        return [self.accumulate_band(col, ns, hist, cov, r0, r1, prm, seed, scale) for (col, ns, hist, cov, r0, r1, seed, scale) in jobs]

    def zeros_like_rows(self, t, rows):
        return torch.zeros((rows,) + tuple(t.shape[1:]), dtype=t.dtype)

    def finalize(self, s, c):
        out = np.empty(tuple(s.shape), np.float32)
        with np.errstate(all="ignore"):
            ol.oracle().bcdo_finalize(ol._fp(_np(s)), np.ascontiguousarray(c.numpy(), np.int32).ctypes.data_as(C.POINTER(C.c_int32)),
                                      C.c_int64(c.numel()), ol._fp(out))
        return torch.from_numpy(out)

    def merge(self, hi, lo):
        return torch.from_numpy(ol.oracle_ops()["merge"](_np(hi), _np(lo)))
