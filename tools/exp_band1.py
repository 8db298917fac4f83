#!/usr/bin/env python3
"""one interior rank of an N-way split, exchanges answered locally (see exp_band.py); for rocprof timelines: python tools/exp_band1.py W H N"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bcd_amd.core as core
import bcd_amd.hip as bh
from bcd_amd.tiling import BandGeometry, HipEngine, band_program
W, H, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
S, b, w = 3, 6, 1
prm = bh.default_params(b=b, w=w, m=1.0, random_order=1, seed=1234)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = bh.Context(0, stream)
eng = HipEngine(ctx)
geom = BandGeometry(W, H, S, b, w, world)
rank = world // 2
g0, g1 = geom.input_lines(rank)
col, ns, hist, cov = core.synthetic_scene(W, H, 32, 1234, 0.35, 0.01, g0, g1 - g0)
inputs = [torch.from_numpy(a).cuda() for a in (col, ns, hist, cov)]
def step():
    prog = band_program(eng, geom, rank, *inputs, prm, prm.order_seed)
    try:
        msg = next(prog)
        while True:
            up, down = msg[1], msg[2]
            if len(msg) == 5:   # receive buffers given: deliver in place
                for dst, src in ((msg[3], up), (msg[4], down)):
                    if dst is not None:
                        for d, t in zip(dst, src):
                            d.copy_(t)
                msg = prog.send(None)
            else:
                msg = prog.send((None if up is None else [t.clone() for t in up], None if down is None else [t.clone() for t in down]))
    except StopIteration as e:
        return e.value
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print("ms/step %.3f" % ((time.perf_counter() - t0) * 100))
