"""Row-band partition of ONE frame over N ranks (one process per GPU), the multi-GPU form of the hot path.

The reference is single-process; this is the build's own decomposition:
  * the frame is cut into N horizontal bands of main pixels whose boundaries are multiples of 2^(S-1) lines, so
    every pyramid level of a band is built from exactly the 2x2 blocks the full frame would use
    (MultiscaleDenoiser.cpp:256-266 of the reference);
  * a rank holds, per scale, its band plus (b+w) halo lines of INPUT on each interior side (more where the
    coarser level needs them) -- no input exchange at run time;
  * per scale there are three neighbour exchanges (the only communication, point-to-point, <= 1 MB):
      1. accumulator halos: the (b+w) lines of sum(3)+count(1) written outside the owned band,
      2. two lines of the finalised (unmerged) output, needed by `hi - up(down(hi))` at the band edge,
      3. one line of the merged output, needed by `up(lo)` of the next finer scale.
    No collective is involved; with torch.distributed these are batched isend/irecv over RCCL (xGMI).
  * `-m 1` marking runs per band (each rank's fixed point sees only its own pixels): a valid greedy order, but not
    the single-GPU image; `-m 0` is order-free and matches the single-GPU result to fp32 round-off.

The orchestration is engine-agnostic (torch tensors in, engine does the math): `HipEngine` drives libbcd_hip.so;
the CPU tests plug an oracle-backed engine in (tests/) and run it with gloo, world_size 2.
"""
from dataclasses import dataclass


@dataclass
class ScaleBand:
    W: int
    H: int        # full-frame height at this scale
    own0: int     # owned lines [own0, own1) (global, this scale)
    own1: int
    loc0: int     # locally held lines [loc0, loc1)
    loc1: int


class BandGeometry:
    """pure integer bookkeeping of the partition (unit-tested on CPU)"""

    def __init__(self, W, H, nscales, search_radius, patch_radius, world):
        self.W, self.H, self.S, self.b, self.w, self.world = W, H, nscales, search_radius, patch_radius, world
        self.halo = search_radius + patch_radius
        self.align = 1 << (nscales - 1)
        units = H // self.align
        if units < world:
            raise ValueError("frame too small: %d lines cannot be split into %d bands aligned to %d" % (H, world, self.align))
        self.bounds = [(units * r // world) * self.align for r in range(world)] + [H]
        self.Ws = [W >> s for s in range(nscales)]
        self.Hs = [H >> s for s in range(nscales)]
        coarsest = nscales - 1
        for r in range(world):
            o0, o1 = self.owned(r, coarsest)
            if world > 1 and o1 - o0 < max(self.halo, 2):
                raise ValueError("band %d owns only %d lines at the coarsest scale (< halo %d): use fewer ranks" % (r, o1 - o0, self.halo))

    def owned(self, rank, s):
        o0 = self.bounds[rank] >> s
        o1 = self.Hs[s] if rank == self.world - 1 else self.bounds[rank + 1] >> s
        return o0, o1

    def scale_bands(self, rank):
        """per scale: owned and locally-held line ranges; level s+1 is the 2x2 reduction of local lines
        [2*loc0(s+1), 2*loc1(s+1)) of level s"""
        out = [None] * self.S
        for s in range(self.S - 1, -1, -1):
            o0, o1 = self.owned(rank, s)
            l0 = max(0, o0 - self.halo) if rank > 0 else 0
            l1 = min(self.Hs[s], o1 + self.halo) if rank < self.world - 1 else self.Hs[s]
            if s < self.S - 1:
                nxt = out[s + 1]
                l0 = min(l0, 2 * nxt.loc0)
                l1 = max(l1, min(self.Hs[s], 2 * nxt.loc1))
                if rank == self.world - 1:
                    l1 = self.Hs[s]
            # merges work on [own0-2, own1+2): keep the local start even so that local line l/2 maps to the coarser level
            if l0 % 2:
                l0 -= 1
            out[s] = ScaleBand(self.Ws[s], self.Hs[s], o0, o1, l0, l1)
        return out

    def input_lines(self, rank):
        sb = self.scale_bands(rank)[0]
        return sb.loc0, sb.loc1


# ---------------------------------------------------------------------------------------------------------
def band_program(eng, geom, rank, col, ns, hist, cov, prm, seed0):
    """generator: yields (tag, send_up, send_down) at each neighbour exchange and receives (from_up, from_down);
    each item is a list of tensors or None at the frame border.  Returns the owned lines of the denoised frame."""
    S, halo, world = geom.S, geom.halo, geom.world
    bands = geom.scale_bands(rank)
    up, down = rank > 0, rank < world - 1
    # ---- local pyramid (MultiscaleDenoiser.cpp:41-53)
    cols, nss, hists, covs = [col], [ns], [hist], [cov]
    for s in range(1, S):
        prev, cur = bands[s - 1], bands[s]
        a, b_ = 2 * cur.loc0 - prev.loc0, 2 * cur.loc1 - prev.loc0
        cols.append(eng.downscale_avg(cols[s - 1][a:b_]))
        nss.append(eng.downscale_sum(nss[s - 1][a:b_]))
        hists.append(eng.downscale_sum(hists[s - 1][a:b_]))
        covs.append(eng.downscale_cov(covs[s - 1][a:b_], nss[s - 1][a:b_]))
    outs = [None] * S
    for s in range(S - 1, -1, -1):
        sb = bands[s]
        o0, o1 = sb.own0 - sb.loc0, sb.own1 - sb.loc0           # owned lines, local indices
        sum_, cnt = eng.accumulate_band(cols[s], nss[s], hists[s], covs[s], o0, o1, prm, eng.scale_seed(seed0, s), s)
        # 1. accumulator halos
        send_up = [sum_[o0 - halo:o0].contiguous(), cnt[o0 - halo:o0].contiguous()] if up else None
        send_down = [sum_[o1:o1 + halo].contiguous(), cnt[o1:o1 + halo].contiguous()] if down else None
        got_up, got_down = yield ("acc%d" % s, send_up, send_down)
        if up:
            sum_[o0:o0 + halo] += got_up[0]
            cnt[o0:o0 + halo] += got_up[1]
        if down:
            sum_[o1 - halo:o1] += got_down[0]
            cnt[o1 - halo:o1] += got_down[1]
        out = eng.finalize(sum_, cnt)                             # valid on owned lines
        if s < S - 1:
            # 2. two lines of the unmerged output each side, then merge on [own0-2, own1+2)
            send_up = [out[o0:o0 + 2].contiguous()] if up else None
            send_down = [out[o1 - 2:o1].contiguous()] if down else None
            got_up, got_down = yield ("out%d" % s, send_up, send_down)
            if up:
                out[o0 - 2:o0] = got_up[0]
            if down:
                out[o1:o1 + 2] = got_down[0]
            m0 = o0 - 2 if up else o0
            m1 = o1 + 2 if down else o1
            nb = bands[s + 1]
            lo = outs[s + 1]
            g_lo0 = (sb.loc0 + m0) // 2 - nb.loc0                 # local line of the coarser level under local line m0
            out[m0:m1] = eng.merge(out[m0:m1], lo[g_lo0:g_lo0 + (m1 - m0) // 2])
        if s > 0:
            # 3. one line of the merged output each side, for up(lo) of the next finer scale
            send_up = [out[o0:o0 + 1].contiguous()] if up else None
            send_down = [out[o1 - 1:o1].contiguous()] if down else None
            got_up, got_down = yield ("mrg%d" % s, send_up, send_down)
            if up:
                out[o0 - 1:o0] = got_up[0]
            if down:
                out[o1:o1 + 1] = got_down[0]
        outs[s] = out
    sb = bands[0]
    return outs[0][sb.own0 - sb.loc0:sb.own1 - sb.loc0]


def run_virtual(eng, geom, inputs_per_rank, prm, seed0):
    """all bands in ONE process, exchanges routed in memory (single-GPU / CPU check of the band path).
    inputs_per_rank[r] = (col, ns, hist, cov) of rank r's local lines.  Returns the owned outputs per rank."""
    def snap(m):  # messages are views into live buffers: copy them like a real send would
        return None if m is None else (m[0],) + tuple(None if x is None else [t.clone() for t in x] for x in m[1:])

    progs = [band_program(eng, geom, r, *inputs_per_rank[r], prm, seed0) for r in range(geom.world)]
    msgs = [snap(next(p)) for p in progs]
    results = [None] * geom.world
    while any(m is not None for m in msgs):
        tags = {m[0] for m in msgs if m is not None}
        assert len(tags) == 1, tags
        nxt = []
        for r, p in enumerate(progs):
            from_up = msgs[r - 1][2] if r > 0 else None
            from_down = msgs[r + 1][1] if r < geom.world - 1 else None
            try:
                nxt.append(snap(p.send((from_up, from_down))))
            except StopIteration as e:
                results[r] = e.value
                nxt.append(None)
        msgs = nxt
    return results


def run_distributed(eng, geom, rank, dist, inputs, prm, seed0, device=None):
    """one band per rank; exchanges are batched isend/irecv with the two neighbours (RCCL on GPUs, gloo on CPU)"""
    import torch
    prog = band_program(eng, geom, rank, *inputs, prm, seed0)
    world = geom.world
    try:
        msg = next(prog)
        while True:
            _, send_up, send_down = msg
            ops, got_up, got_down = [], None, None
            if rank > 0:
                got_up = [torch.empty_like(t) for t in send_up]
                for t in send_up:
                    ops.append(dist.P2POp(dist.isend, t, rank - 1))
                for t in got_up:
                    ops.append(dist.P2POp(dist.irecv, t, rank - 1))
            if rank < world - 1:
                got_down = [torch.empty_like(t) for t in send_down]
                for t in send_down:
                    ops.append(dist.P2POp(dist.isend, t, rank + 1))
                for t in got_down:
                    ops.append(dist.P2POp(dist.irecv, t, rank + 1))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            msg = prog.send((got_up, got_down))
    except StopIteration as e:
        return e.value


# ---------------------------------------------------------------------------------------------------------
class HipEngine:
    """the math of a band on the MI355X engine (libbcd_hip.so through bcd_amd.hip)"""

    def __init__(self, ctx, reuse_buffers=True):
        import bcd_amd.hip as bh
        self.bh = bh
        self.ctx = ctx
        self.torch = ctx.torch
        self._acc = {}
        self.reuse_buffers = reuse_buffers  # False when several virtual ranks share one engine

    def scale_seed(self, seed0, s):
        return self.bh.scale_seed(seed0, s)

    def downscale_avg(self, t):
        return self.ctx.downscale_avg(t.contiguous())

    def downscale_sum(self, t):
        return self.ctx.downscale_sum(t.contiguous())

    def downscale_cov(self, cov, ns):
        return self.ctx.downscale_cov(cov.contiguous(), ns.contiguous())

    def accumulate_band(self, col, ns, hist, cov, row0, row1, prm, seed, scale):
        H, W, _ = hist.shape
        key = (scale, H, W)
        if key not in self._acc or not self.reuse_buffers:
            self._acc[key] = (self.torch.empty((H, W, 3), dtype=self.torch.float32, device=hist.device),
                              self.torch.empty((H, W), dtype=self.torch.int32, device=hist.device))
        s, c = self._acc[key]
        self.ctx.denoise_band(col, ns, hist, cov, row0, row1, prm, seed, s, c)
        return s, c

    def finalize(self, s, c):
        return self.ctx.finalize(s, c)

    def merge(self, hi, lo):
        return self.ctx.merge(hi.contiguous(), lo.contiguous())


class BandDenoiser:
    """bench.py / library front-end of one rank's band"""

    def __init__(self, ctx, dist, rank, world, W, H, D, nscales, prm):
        self.ctx, self.dist, self.rank, self.world = ctx, dist, rank, world
        self.geom = BandGeometry(W, H, nscales, prm.search_radius, prm.patch_radius, world)
        self.eng = HipEngine(ctx)
        self.prm = prm
        self.inputs = None
        self.out = None
        self.shared_stream = True

    def input_lines(self):
        return self.geom.input_lines(self.rank)

    def owned_lines(self):
        return self.geom.owned(self.rank, 0)

    def upload(self, col, ns, hist, cov):
        torch = self.ctx.torch
        self.inputs = [torch.from_numpy(a).cuda() for a in (col, ns, hist, cov)]
        torch.cuda.synchronize()

    def step(self):
        # the context must be bound to torch's CURRENT stream (bench.py does that): engine kernels, torch slicing/adds
        # and the RCCL point-to-point ops are then ordered by the stream itself.  Otherwise fence explicitly.
        eng = self.eng if self.shared_stream else _SyncedEngine(self.eng)
        self.out = run_distributed(eng, self.geom, self.rank, self.dist, self.inputs, self.prm, self.prm.order_seed)
        return self.out


class _SyncedEngine:
    """wraps an engine so that every call is complete before torch (and RCCL) touch its results"""

    def __init__(self, eng):
        self._e = eng

    def scale_seed(self, a, b):
        return self._e.scale_seed(a, b)

    def __getattr__(self, name):
        f = getattr(self._e, name)

        def call(*a, **k):
            self._e.torch.cuda.current_stream().synchronize()
            r = f(*a, **k)
            self._e.ctx.synchronize()
            return r
        return call
