#!/usr/bin/env python3
"""CPU (NumPy float32): could clampNegativeEigenValues be a matrix-sign iteration on the matrix core instead of an eigensolve?  A+ = (A + A sign(A)) / 2;
sign(A) by the cubic Newton-Schulz step X <- 1.5 X - 0.5 X^3 from X0 = A / ||A||_F, all products in float32 (what v_mfma_f32_32x32x2_f32 computes).
Matrices: the generator of tools/exp_eig.py (sample covariances of 30-90 points in 27-D minus a block-diagonal noise estimate).  Error of A+ against
float64 eigh, relative to ||A||_2, after k iterations; the Jacobi path's budget is ~1e-6.  (counted, not built.)"""
import numpy as np


def matrices(n, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        m = int(rng.integers(30, 90))
        X = rng.standard_normal((m, 27)) * (0.05 + 0.3 * rng.random(27))
        C = np.cov(X.T)
        N = np.zeros((27, 27))
        for o in range(9):
            B = rng.standard_normal((3, 3)) * 0.05
            N[3 * o:3 * o + 3, 3 * o:3 * o + 3] = B @ B.T
        out.append((C - N))
    return out


def main():
    mats = matrices(200)
    ks = [8, 12, 16, 20, 24, 28, 32, 40]
    worst = {k: 0.0 for k in ks}
    for A64 in mats:
        lam, V = np.linalg.eigh(A64)
        ref = (V * np.maximum(lam, 0.0)) @ V.T
        scale = np.max(np.abs(lam))
        A = A64.astype(np.float32)
        X = A / np.float32(np.linalg.norm(A))          # ||A||_F >= ||A||_2: eigenvalues of X0 in [-1, 1]
        for k in range(1, max(ks) + 1):
            X2 = X @ X
            X = np.float32(1.5) * X - np.float32(0.5) * (X2 @ X)
            X = np.float32(0.5) * (X + X.T)             # (the kernel's symmetric operand trick assumes symmetry: what re-symmetrising every step costs is one add)
            if k in worst:
                Ap = np.float32(0.5) * (A + A @ X)
                Ap = np.float32(0.5) * (Ap + Ap.T)
                worst[k] = max(worst[k], float(np.max(np.abs(Ap.astype(np.float64) - ref)) / scale))
    for k in ks:
        print("k = %2d iterations (%2d products): worst |A+ - ref| / ||A||_2 over %d matrices = %.2e" % (k, 2 * k + 1, len(mats), worst[k]))


if __name__ == "__main__":
    main()
