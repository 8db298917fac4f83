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

    # ---- stage-level calls used by the exact-marking band program (numpy restatement of the marking fixed point) ----
    @staticmethod
    def _mix32(x):
        x = np.asarray(x, np.uint64) & 0xFFFFFFFF
        x ^= x >> 16; x = (x * 0x85ebca6b) & 0xFFFFFFFF
        x ^= x >> 13; x = (x * 0xc2b2ae35) & 0xFFFFFFFF
        x ^= x >> 16
        return x

    def _keys(self, idx, random_order, seed):
        """64-bit visiting key of global pixel indices (bcd_common.h: bcd_order_key)"""
        idx = np.asarray(idx, np.uint64)
        hi = self._mix32(idx ^ self._mix32((seed + 0x9E3779B9) & 0xFFFFFFFF)) if random_order else np.zeros_like(idx)
        return (hi << np.uint64(32)) | idx

    def similarity(self, hist, ns, w, b, tau):
        mask, cnt = ol.similarity_masks(_np(ns), _np(hist), w, b, tau, threads=1)
        return torch.from_numpy(mask.view(np.int32)), torch.from_numpy(cnt)

    def active_init(self, nsim, w, row0, row1, m, seed, row_offset):
        H, W = nsim.shape
        st = np.zeros((H, W), np.uint8)
        main = np.zeros((H, W), bool)
        main[max(w, row0):min(H - w, row1), w:W - w] = True
        if m <= 0:
            st[main] = 1
        elif m >= 1:
            st[main] = 3
        else:
            l, c = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
            gidx = ((l + row_offset) * W + c).astype(np.uint64)
            u = (self._mix32((gidx * 0x9E3779B1 + self._mix32(seed ^ 0x51ed270b)) & 0xFFFFFFFF) >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)
            st[main] = np.where(u[main] < np.float32(m), 3, 1)
        return torch.from_numpy(st)

    def active_step(self, mask, nsim, state, w, b, row0, row1, random_order, seed, row_offset, first):
        """one Jacobi-style round over the band's undecided pixels (in place on `state`); returns how many remain undecided"""
        m = mask.numpy().view(np.uint32)
        ns_, st = nsim.numpy(), state.numpy()
        H, W = st.shape
        side, K1 = 2 * b + 1, 3 * (2 * w + 1) ** 2 + 1
        l, c = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        keys = self._keys((l + row_offset) * W + c, random_order, seed)
        new = st.copy()
        for r in range(max(row0, 0), min(row1, H)):
            for cc in range(W):
                if st[r, cc] != 3:
                    continue
                any_in, wait = False, False
                for k in range(side * side):
                    if not (m[r, cc, k >> 5] >> np.uint32(k & 31)) & 1:
                        continue
                    qr, qc = r + k // side - b, cc + k % side - b
                    if (qr, qc) == (r, cc) or ns_[qr, qc] < K1 or keys[qr, qc] > keys[r, cc]:
                        continue
                    if st[qr, qc] == 1:
                        any_in = True
                        break
                    if st[qr, qc] == 3:
                        wait = True
                if any_in:
                    new[r, cc] = 2
                elif not wait:
                    new[r, cc] = 1
        st[...] = new
        return int((st[max(row0, 0):min(row1, H)] == 3).sum())

    def bayes(self, col, cov, ns, hist, mask, nsim, state, prm):
        H, W, D = hist.shape
        pix = np.ascontiguousarray(np.flatnonzero(state.numpy().reshape(-1) == 1), np.int32)
        s = np.empty((H, W, 3), np.float32)
        c = np.empty((H, W), np.int32)
        op = ol.params(prm.hist_dist_threshold, prm.patch_radius, prm.search_radius, prm.min_eigen_value, 0.0)
        rc = ol.oracle().bcdo_accumulate_pixels(ol._fp(_np(col)), ol._fp(_np(ns)), ol._fp(_np(hist)), ol._fp(_np(cov)), W, H, D, C.byref(op),
                                                pix.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(pix.size), ol._fp(s),
                                                c.ctypes.data_as(C.POINTER(C.c_int32)))
        assert rc == 0
        return torch.from_numpy(s), torch.from_numpy(c)

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
