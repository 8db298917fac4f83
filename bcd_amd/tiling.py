"""Row-band partition of ONE frame over N ranks (one process per GPU), the multi-GPU form of the hot path.

The reference is single-process; this is the build's own decomposition:
  * the frame is cut into N horizontal bands of main pixels whose boundaries are multiples of 2^(S-1) lines, so
    every pyramid level of a band is built from exactly the 2x2 blocks the full frame would use
    (MultiscaleDenoiser.cpp:256-266 of the reference);
  * a rank holds, per scale, its band plus (b+w) halo lines of INPUT on each interior side (more where the
    coarser level needs them) -- no input exchange at run time;
  * the scales' bands go through the denoiser concurrently, then S neighbour exchanges per frame (the only communication,
    point-to-point, <= 1 MB each):
      1. accumulator halos of ALL scales: the (b+w) lines of sum(3)+count(1) written outside the owned band,
      2. two lines of every finalised (unmerged) finer output, needed by `hi - up(down(hi))` at the band edge, and one line
         of the coarsest output,
      3. per intermediate scale, one line of the merged output, needed by `up(lo)` of the next finer scale.
    No collective is involved; with torch.distributed these are batched isend/irecv over RCCL (xGMI).
  * `-m 1` marking: by default per band (each rank's fixed point sees only its own pixels): a valid greedy order, but not
    the single-GPU image; `exact_marking=True` (band_program_exact) follows the visiting order of the whole frame by
    exchanging boundary states between marking launches -- the single-GPU image, at the price of a few more exchanges and
    one small all-reduce per launch batch.  `-m 0` is order-free and matches the single-GPU result either way.

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
    # ---- A. every scale's band through the denoiser (independent of each other: the engine may run them concurrently);
    # a scale only needs its owned lines +- (b+w), the rest of the local band exists to build the coarser pyramid levels
    jobs, spans = [], []
    for s in range(S):
        sb = bands[s]
        o0, o1 = sb.own0 - sb.loc0, sb.own1 - sb.loc0           # owned lines, local indices
        a0 = o0 - halo if up else 0
        a1 = o1 + halo if down else sb.loc1 - sb.loc0
        spans.append((o0, o1, a0, a1))
        jobs.append((cols[s][a0:a1], nss[s][a0:a1], hists[s][a0:a1], covs[s][a0:a1], o0 - a0, o1 - a0, eng.scale_seed(seed0, s), s))
    accs = eng.accumulate_bands(jobs, prm)                       # [(sum, count)] indexed like [a0, a1)
    # ---- B. one exchange for the accumulator halos of all scales
    send_up = [t[:halo].contiguous() for acc in accs for t in acc] if up else None
    send_down = [t[-halo:].contiguous() for acc in accs for t in acc] if down else None
    got_up, got_down = yield ("acc", send_up, send_down)
    result = yield from _finish_bands(eng, geom, rank, bands, spans, accs, got_up, got_down)
    return result


def _finish_bands(eng, geom, rank, bands, spans, accs, got_up, got_down):
    """common tail: add the received accumulator halos, finalise, exchange output halos, merge coarse to fine"""
    S, halo, world = geom.S, geom.halo, geom.world
    up, down = rank > 0, rank < world - 1
    outs = [None] * S
    fused = hasattr(eng, "finalize_band")
    for s in range(S):
        sb = bands[s]
        o0, o1, a0, a1 = spans[s]
        sum_, cnt = accs[s]
        if fused:
            # one launch: halo add + finalisation of the owned lines, written in place into the (persistent) local output
            out = eng.out_buffer(s, sb.loc1 - sb.loc0, sum_)
            eng.finalize_band(sum_[o0 - a0:o1 - a0], cnt[o0 - a0:o1 - a0], halo,
                              (got_up[2 * s], got_up[2 * s + 1]) if up else None,
                              (got_down[2 * s], got_down[2 * s + 1]) if down else None, out[o0:o1])
            outs[s] = out
            continue
        if up:
            sum_[o0 - a0:o0 - a0 + halo] += got_up[2 * s]
            cnt[o0 - a0:o0 - a0 + halo] += got_up[2 * s + 1]
        if down:
            sum_[o1 - a0 - halo:o1 - a0] += got_down[2 * s]
            cnt[o1 - a0 - halo:o1 - a0] += got_down[2 * s + 1]
        fin = eng.finalize(sum_, cnt)                             # valid on owned lines
        out = eng.zeros_like_rows(fin, sb.loc1 - sb.loc0)
        out[o0:o1] = fin[o0 - a0:o1 - a0]
        outs[s] = out
    # ---- C. one exchange: two lines of every unmerged finer output (for hi - up(down(hi)) at the band edge) and one line of
    # the coarsest output (for up(lo) of the scale above it)
    def edge(s, n, top):
        o0, o1 = spans[s][0], spans[s][1]
        return outs[s][o0:o0 + n].contiguous() if top else outs[s][o1 - n:o1].contiguous()
    lines = [2] * (S - 1) + [1]
    if S > 1:
        send_up = [edge(s, lines[s], True) for s in range(S)] if up else None
        send_down = [edge(s, lines[s], False) for s in range(S)] if down else None
        # the received lines land directly in the outputs (contiguous row views)
        recv_up = [outs[s][spans[s][0] - lines[s]:spans[s][0]] for s in range(S)] if up else None
        recv_down = [outs[s][spans[s][1]:spans[s][1] + lines[s]] for s in range(S)] if down else None
        yield ("out", send_up, send_down, recv_up, recv_down)
    # ---- D. merges coarse to fine; between two merges one line of the freshly merged output travels
    for s in range(S - 2, -1, -1):
        sb, nb = bands[s], bands[s + 1]
        o0, o1 = spans[s][0], spans[s][1]
        m0 = o0 - 2 if up else o0
        m1 = o1 + 2 if down else o1
        g_lo0 = (sb.loc0 + m0) // 2 - nb.loc0                     # local line of the coarser level under local line m0
        if fused:
            eng.merge_(outs[s][m0:m1], outs[s + 1][g_lo0:g_lo0 + (m1 - m0) // 2])
        else:
            outs[s][m0:m1] = eng.merge(outs[s][m0:m1], outs[s + 1][g_lo0:g_lo0 + (m1 - m0) // 2])
        if s > 0:
            send_up = [outs[s][o0:o0 + 1].contiguous()] if up else None
            send_down = [outs[s][o1 - 1:o1].contiguous()] if down else None
            yield ("mrg%d" % s, send_up, send_down, [outs[s][o0 - 1:o0]] if up else None, [outs[s][o1:o1 + 1]] if down else None)
    sb = bands[0]
    return outs[0][sb.own0 - sb.loc0:sb.own1 - sb.loc0]


def band_program_exact(eng, geom, rank, col, ns, hist, cov, prm, seed0):
    """like band_program, but the marking strategy (-m > 0) follows the visiting order of the WHOLE frame: keys are functions of
    the global pixel index, bands exchange the strong flags and, after every batch of marking launches, the states of their b
    boundary lines, and a global sum of undecided pixels ends the iteration.  The processed set -- hence the image -- is the
    single-GPU one.  Extra messages: ("sum", value) is answered with the sum over all ranks.  Scales run one after the other."""
    S, halo, world, b, w = geom.S, geom.halo, geom.world, geom.b, geom.w
    bands = geom.scale_bands(rank)
    up, down = rank > 0, rank < world - 1
    cols, nss, hists, covs = [col], [ns], [hist], [cov]
    for s in range(1, S):
        prev, cur = bands[s - 1], bands[s]
        a, b_ = 2 * cur.loc0 - prev.loc0, 2 * cur.loc1 - prev.loc0
        cols.append(eng.downscale_avg(cols[s - 1][a:b_]))
        nss.append(eng.downscale_sum(nss[s - 1][a:b_]))
        hists.append(eng.downscale_sum(hists[s - 1][a:b_]))
        covs.append(eng.downscale_cov(covs[s - 1][a:b_], nss[s - 1][a:b_]))
    accs, spans = [], []
    for s in range(S):
        sb = bands[s]
        o0, o1 = sb.own0 - sb.loc0, sb.own1 - sb.loc0
        a0 = o0 - halo if up else 0
        a1 = o1 + halo if down else sb.loc1 - sb.loc0
        spans.append((o0, o1, a0, a1))
        r0, r1 = o0 - a0, o1 - a0                                   # owned lines inside the sub-band [a0, a1)
        row_offset = sb.loc0 + a0                                   # global line of sub-band line 0
        seed = eng.scale_seed(seed0, s)
        c_, n_, h_, v_ = cols[s][a0:a1], nss[s][a0:a1], hists[s][a0:a1], covs[s][a0:a1]
        mask, nsim = eng.similarity(h_, n_, w, b, prm.hist_dist_threshold)
        # |S| of the b boundary lines comes from their owner (locally their windows are cut by the band edge)
        got_up, got_down = yield ("nsim%d" % s, [nsim[r0:r0 + b].contiguous()] if up else None, [nsim[r1 - b:r1].contiguous()] if down else None)
        if up:
            nsim[r0 - b:r0] = got_up[0]
        if down:
            nsim[r1:r1 + b] = got_down[0]
        state = eng.active_init(nsim, w, r0, r1, prm.marked_skip_probability, seed, row_offset)
        if prm.marked_skip_probability > 0:
            first = prm.marked_skip_probability >= 1.0
            before = None
            while True:   # every batch decides at least the earliest undecided pixel of the frame: ends after finitely many batches
                got_up, got_down = yield ("st%d" % s, [state[r0:r0 + b].contiguous()] if up else None, [state[r1 - b:r1].contiguous()] if down else None)
                if up:
                    state[r0 - b:r0] = got_up[0]
                if down:
                    state[r1:r1 + b] = got_down[0]
                left = eng.active_step(mask, nsim, state, w, b, r0, r1, prm.use_random_pixel_order, seed, row_offset, first)
                first = False
                total = yield ("sum", left)
                if total == 0:
                    break
                if before is not None and total >= before:
                    raise RuntimeError("marking fixed point made no progress")
                before = total
        state[:r0] = 0                                              # halo lines are processed by their owner
        state[r1:] = 0
        accs.append(eng.bayes(c_, v_, n_, h_, mask, nsim, state, prm))
    # ---- from here on identical to band_program: accumulator halos, finalisation, output halos, merges
    send_up = [t[:halo].contiguous() for acc in accs for t in acc] if up else None
    send_down = [t[-halo:].contiguous() for acc in accs for t in acc] if down else None
    got_up, got_down = yield ("acc", send_up, send_down)
    result = yield from _finish_bands(eng, geom, rank, bands, spans, accs, got_up, got_down)
    return result


def run_virtual(eng, geom, inputs_per_rank, prm, seed0, exact_marking=False):
    """all bands in ONE process, exchanges routed in memory (single-GPU / CPU check of the band path).
    inputs_per_rank[r] = (col, ns, hist, cov) of rank r's local lines.  Returns the owned outputs per rank."""
    def snap(m):  # messages are views into live buffers: copy them like a real send would
        if m is None or m[0] == "sum":
            return m
        return (m[0],) + tuple(None if x is None else [t.clone() for t in x] for x in m[1:3]) + tuple(m[3:])

    program = band_program_exact if exact_marking else band_program
    progs = [program(eng, geom, r, *inputs_per_rank[r], prm, seed0) for r in range(geom.world)]
    msgs = [snap(next(p)) for p in progs]
    results = [None] * geom.world
    while any(m is not None for m in msgs):
        tags = {m[0] for m in msgs if m is not None}
        assert len(tags) == 1, tags
        nxt = []
        total = sum(m[1] for m in msgs) if "sum" in tags else None
        for r, p in enumerate(progs):
            from_up = msgs[r - 1][2] if r > 0 and total is None else None
            from_down = msgs[r + 1][1] if r < geom.world - 1 and total is None else None
            if total is None and len(msgs[r]) == 5:        # receive buffers given: deliver in place
                for dst, src in ((msgs[r][3], from_up), (msgs[r][4], from_down)):
                    if dst is not None:
                        for d, s_ in zip(dst, src):
                            d.copy_(s_)
                from_up = from_down = None
            try:
                nxt.append(snap(p.send(total if total is not None else (from_up, from_down))))
            except StopIteration as e:
                results[r] = e.value
                nxt.append(None)
        msgs = nxt
    return results


def run_distributed(eng, geom, rank, dist, inputs, prm, seed0, device=None, exact_marking=False):
    """one band per rank; exchanges are batched isend/irecv with the two neighbours (RCCL on GPUs, gloo on CPU)"""
    import torch
    prog = (band_program_exact if exact_marking else band_program)(eng, geom, rank, *inputs, prm, seed0)
    world = geom.world
    try:
        msg = next(prog)
        while True:
            if msg[0] == "sum":
                tsum = torch.tensor([int(msg[1])], dtype=torch.int64, device=device if device is not None else "cpu")
                dist.all_reduce(tsum)
                msg = prog.send(int(tsum.item()))
                continue
            send_up, send_down = msg[1], msg[2]
            in_place = len(msg) == 5
            ops, got_up, got_down = [], None, None
            if rank > 0:
                got_up = msg[3] if in_place else [torch.empty_like(t) for t in send_up]
                for t in send_up:
                    ops.append(dist.P2POp(dist.isend, t, rank - 1))
                for t in got_up:
                    ops.append(dist.P2POp(dist.irecv, t, rank - 1))
            if rank < world - 1:
                got_down = msg[4] if in_place else [torch.empty_like(t) for t in send_down]
                for t in send_down:
                    ops.append(dist.P2POp(dist.isend, t, rank + 1))
                for t in got_down:
                    ops.append(dist.P2POp(dist.irecv, t, rank + 1))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            msg = prog.send(None if in_place else (got_up, got_down))
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
        col, ns, hist, cov = col.contiguous(), ns.contiguous(), hist.contiguous(), cov.contiguous()
        H, W, _ = hist.shape
        key = (scale, H, W)
        if key not in self._acc or not self.reuse_buffers:
            self._acc[key] = (self.torch.empty((H, W, 3), dtype=self.torch.float32, device=hist.device),
                              self.torch.empty((H, W), dtype=self.torch.int32, device=hist.device))
        s, c = self._acc[key]
        self.ctx.denoise_band(col, ns, hist, cov, row0, row1, prm, seed, s, c)
        return s, c

    def accumulate_bands(self, jobs, prm):
        """jobs: (col, ns, hist, cov, row0, row1, seed, scale); all bands through the engine concurrently"""
        torch, cj, out = self.torch, [], []
        for (col, ns, hist, cov, r0, r1, seed, scale) in jobs:
            col, ns, hist, cov = col.contiguous(), ns.contiguous(), hist.contiguous(), cov.contiguous()
            H, W, _ = hist.shape
            key = (scale, H, W)
            if key not in self._acc or not self.reuse_buffers:
                self._acc[key] = (torch.empty((H, W, 3), dtype=torch.float32, device=hist.device),
                                  torch.empty((H, W), dtype=torch.int32, device=hist.device))
            s, c = self._acc[key]
            cj.append((col, ns, hist, cov, r0, r1, seed, s, c))
            out.append((s, c))
        self.ctx.denoise_bands(cj, prm)
        return out

    def zeros_like_rows(self, t, rows):
        return self.torch.zeros((rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)

    # ---- stage-level calls used by the exact-marking band program
    def similarity(self, hist, ns, w, b, tau):
        return self.ctx.similarity_masks(hist.contiguous(), ns.contiguous(), w, b, tau)

    def active_init(self, nsim, w, row0, row1, m, seed, row_offset):
        return self.ctx.active_init(nsim, w, row0, row1, m, seed, row_offset)

    def active_step(self, mask, nsim, state, w, b, row0, row1, random_order, seed, row_offset, first):
        return self.ctx.active_step(mask, nsim, state, w, b, row0, row1, random_order, seed, row_offset, first)

    def bayes(self, col, cov, ns, hist, mask, nsim, state, prm):
        pixcov = self.ctx.pixel_cov(cov.contiguous(), ns.contiguous())
        return self.ctx.bayes_accumulate(col.contiguous(), pixcov, mask, nsim, state, prm.patch_radius, prm.search_radius, prm.min_eigen_value)

    def finalize(self, s, c):
        return self.ctx.finalize(s, c)

    def merge(self, hi, lo):
        return self.ctx.merge(hi.contiguous(), lo.contiguous())

    # ---- fused tail of a band (fewer, larger launches: at 8 ranks the tail of small torch ops was 45 % of a 720p step)
    def out_buffer(self, scale, rows, like):
        """persistent rows x W x 3 output of a scale's local band (lines outside the owned range are only ever read where a
        received line was written first)"""
        key = ("out", scale, rows, like.shape[1])
        if key not in self._acc or not self.reuse_buffers:
            self._acc[key] = self.torch.empty((rows, like.shape[1], 3), dtype=self.torch.float32, device=like.device)
        return self._acc[key]

    def finalize_band(self, s, c, halo, up, down, out):
        return self.ctx.finalize_band(s, c, halo, up, down, out)

    def merge_(self, hi, lo):
        return self.ctx.merge_(hi, lo.contiguous())


class BandDenoiser:
    """bench.py / library front-end of one rank's band"""

    def __init__(self, ctx, dist, rank, world, W, H, D, nscales, prm, exact_marking=False):
        self.ctx, self.dist, self.rank, self.world = ctx, dist, rank, world
        self.exact_marking = exact_marking
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
        self.out = run_distributed(eng, self.geom, self.rank, self.dist, self.inputs, self.prm, self.prm.order_seed,
                                   device=self.inputs[0].device, exact_marking=self.exact_marking)
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
