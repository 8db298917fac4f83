// k_bayes27.hip -- Bayesian patch estimate for the default patch radius w = 1 (K = 27), one wavefront per
// processed pixel, 18 KB of LDS per wavefront (8 wavefronts per CU).
//
// Same mathematics as DenoisingUnit::denoiseSelectedPatches (src/core/DenoisingUnit.cpp:388-453) and
// aggregateOutputPatches (:672-693), reorganised for the GPU:
//   * the similar patches are streamed through a 32-member LDS chunk three times (mean, covariance,
//     final estimate) instead of being held as n x 27 clouds; sums keep the reference's sequential order;
//   * Step 2's covariance of the Step-1 estimates (:441-443) is obtained without touching the members
//     again: the Step-1 estimate is affine, xhat = x - G (x - m), G = N Cinv1, so its empirical mean is m
//     and its empirical covariance is F C F^T with F = I - G  (exact identity, fp32 round-off apart);
//   * clampNegativeEigenValues (:606-630) is a parallel two-sided Jacobi eigendecomposition in LDS;
//   * inverseSymmetricMatrix (:578-604) = V diag(1/max(minEig, lambda)) V^T equals the plain inverse
//     whenever lambda_min >= minEig.  The inverse is computed with the symmetric sweep operator and
//     accepted only if every pivot is positive and ||M^-1||_F * minEig <= 1 (which proves
//     lambda_min(M) >= minEig); otherwise the spectral form is evaluated with the Jacobi solver.
#include "bcd_common.h"
#include <cstdio>
#include <cstdlib>

__device__ long long bcd_dbg_cycles[16];

namespace {

#define DBG_T(i) do { if (DBG && blockIdx.x == 100 && lane == 0) bcd_dbg_cycles[i] = __builtin_readcyclecounter(); } while (0)

constexpr int K = 27, KP = 28, LD = 29, P = 9, MSZ = KP * LD, CHUNK = 32;

struct Geom27 {
    int W, H, b, side, words, maxS;
};

__device__ inline float wsum(float v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ inline int noise_idx(int i, int j)
{
    // 3x3 symmetric block from xx,yy,zz,yz,xz,xy
    return i == j ? i : (i + j == 1 ? 5 : (i + j == 2 ? 4 : 3));
}

// ---- parallel (round-robin) two-sided Jacobi; A -> ~diag(lambda), V <- eigenvectors (columns) ----------
__device__ void jacobi27(float *A, float *V, float *rc, float *rs, int *rp, int *rq, int lane)
{
    for (int e = lane; e < K * K; e += 64) {
        int r = e / K, c = e - r * K;
        V[r * LD + c] = (r == c) ? 1.f : 0.f;
    }
    __syncthreads();
    constexpr int N1 = KP - 1, NP = KP / 2;
    for (int sweep = 0; sweep < 12; ++sweep) {
        float off = 0.f, dg = 0.f;
        for (int e = lane; e < K * K; e += 64) {
            int r = e / K, c = e - r * K;
            float v = A[r * LD + c];
            if (r == c) dg = fmaf(v, v, dg); else off = fmaf(v, v, off);
        }
        off = wsum(off);
        dg = wsum(dg);
        if (off <= 1e-13f * dg) break;
        for (int round = 0; round < N1; ++round) {
            if (lane < NP) {
                int a = (lane == 0) ? N1 : (round + lane) % N1;
                int bb = (lane == 0) ? round : (round - lane + N1) % N1;
                int p = min(a, bb), q = max(a, bb);
                float c = 1.f, s = 0.f;
                if (q < K) {
                    float apq = A[p * LD + q];
                    if (apq != 0.f) {
                        float app = A[p * LD + p], aqq = A[q * LD + q];
                        float theta = (aqq - app) / (2.f * apq);
                        float t = 1.f / (fabsf(theta) + sqrtf(fmaf(theta, theta, 1.f)));
                        t = theta < 0.f ? -t : t;
                        c = 1.f / sqrtf(fmaf(t, t, 1.f));
                        s = t * c;
                    }
                } else { p = 0; q = 0; }
                rc[lane] = c; rs[lane] = s; rp[lane] = p; rq[lane] = q;
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < (NP * K + 63) / 64; ++it) { // A <- A J, V <- V J
                int t = lane + it * 64;
                if (t < NP * K) {
                    int k = t / K, row = t - k * K;
                    float c = rc[k], s = rs[k];
                    int p = rp[k], q = rq[k];
                    float ap = A[row * LD + p], aq = A[row * LD + q];
                    float vp = V[row * LD + p], vq = V[row * LD + q];
                    if (s != 0.f) {
                        A[row * LD + p] = fmaf(c, ap, -s * aq);
                        A[row * LD + q] = fmaf(s, ap, c * aq);
                        V[row * LD + p] = fmaf(c, vp, -s * vq);
                        V[row * LD + q] = fmaf(s, vp, c * vq);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < (NP * K + 63) / 64; ++it) { // A <- J^T A
                int t = lane + it * 64;
                if (t < NP * K) {
                    int k = t / K, col = t - k * K;
                    float c = rc[k], s = rs[k];
                    int p = rp[k], q = rq[k];
                    float ap = A[p * LD + col], aq = A[q * LD + col];
                    if (s != 0.f) {
                        A[p * LD + col] = fmaf(c, ap, -s * aq);
                        A[q * LD + col] = fmaf(s, ap, c * aq);
                    }
                }
            }
            __syncthreads();
        }
    }
}

// out = V f(lambda) V^T ; f = max(0,.) (clamp) or 1/max(minEig,.) (inverse)
__device__ void rebuild27(float *out, const float *A, const float *V, float *fl, int lane, bool inverse, float min_eig)
{
    if (lane < K) {
        float lam = A[lane * LD + lane];
        fl[lane] = inverse ? 1.f / fmaxf(min_eig, lam) : fmaxf(0.f, lam);
    }
    __syncthreads();
    for (int e = lane; e < K * K; e += 64) {
        int r = e / K, c = e - r * K;
        float s = 0.f;
#pragma unroll 9
        for (int k = 0; k < K; ++k) s = fmaf(V[r * LD + k], fl[k] * V[c * LD + k], s);
        out[r * LD + c] = s;
    }
    __syncthreads();
}

__device__ void add_noise27(float *M, const float *noise, int lane, float sign)
{
    for (int t = lane; t < P * 9; t += 64) {
        int blk = t / 9, e = t - blk * 9, i = e / 3, j = e - i * 3;
        M[(3 * blk + i) * LD + 3 * blk + j] += sign * noise[blk * 6 + noise_idx(i, j)];
    }
    __syncthreads();
}

// in-place inverse of the symmetric positive definite M by the sweep operator.  Returns false (wave-uniform) if a
// pivot is not positive or the bound ||M^-1||_F * min_eig <= 1 fails (then lambda_min >= min_eig is not proven).
__device__ bool sweep_inverse27(float *M, int lane, float min_eig)
{
    constexpr int NE = (K * K + 63) / 64;
    bool ok = true;
    for (int k = 0; k < K; ++k) {
        float d = M[k * LD + k];
        if (!(d > 0.f)) { ok = false; break; } // uniform: every lane reads the same element
        float inv_d = 1.f / d;
        float nv[NE];
#pragma unroll
        for (int it = 0; it < NE; ++it) {
            int e = lane + it * 64;
            float v = 0.f;
            if (e < K * K) {
                int r = e / K, c = e - r * K;
                float mrk = M[r * LD + k], mkc = M[k * LD + c], mrc = M[r * LD + c];
                if (r == k && c == k) v = -inv_d;
                else if (r == k) v = mkc * inv_d;
                else if (c == k) v = mrk * inv_d;
                else v = fmaf(-mrk * inv_d, mkc, mrc);
            }
            nv[it] = v;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NE; ++it) {
            int e = lane + it * 64;
            if (e < K * K) { int r = e / K, c = e - r * K; M[r * LD + c] = nv[it]; }
        }
        __syncthreads();
    }
    if (!ok) return false;
    float fro = 0.f;
    for (int e = lane; e < K * K; e += 64) {
        int r = e / K, c = e - r * K;
        float v = -M[r * LD + c];
        M[r * LD + c] = v;
        fro = fmaf(v, v, fro);
    }
    fro = wsum(fro);
    __syncthreads();
    return sqrtf(fro) * min_eig <= 1.f && isfinite(fro);
}

// M <- inverseSymmetricMatrix(M) (DenoisingUnit.cpp:578-604), in place.  scratch: A (backup / Jacobi), V
__device__ void inverse27(float *M, float *A, float *V, float *fl, float *rc, float *rs, int *rp, int *rq, int lane, float min_eig)
{
    // backup: lower triangle mirrored, like Eigen's SelfAdjointEigenSolver reads it
    for (int e = lane; e < K * K; e += 64) { int r = e / K, c = e - r * K; A[r * LD + c] = M[(r >= c ? r : c) * LD + (r >= c ? c : r)]; }
    __syncthreads();
    if (sweep_inverse27(M, lane, min_eig)) return;
    jacobi27(A, V, rc, rs, rp, rq, lane);
    rebuild27(M, A, V, fl, lane, true, min_eig);
}

// out[3o+i][c] = delta*(r==c) - sign * sum_j N_o[i][j] * in[3o+j][c]   (block-diagonal noise covariance times a dense matrix)
__device__ void noise_times27(float *out, const float *noise, const float *in, int lane, bool identity_minus)
{
    for (int e = lane; e < K * K; e += 64) {
        int r = e / K, c = e - r * K, o = r / 3, i = r - 3 * o;
        const float *n6 = noise + o * 6;
        float g = n6[noise_idx(i, 0)] * in[(3 * o) * LD + c] + n6[noise_idx(i, 1)] * in[(3 * o + 1) * LD + c] +
                  n6[noise_idx(i, 2)] * in[(3 * o + 2) * LD + c];
        out[r * LD + c] = identity_minus ? ((r == c ? 1.f : 0.f) - g) : g;
    }
    __syncthreads();
}

// out = X * Y (TRANS_Y == false) or X * Y^T (TRANS_Y == true); out must not alias X or Y
template <bool TRANS_Y>
__device__ void matmul27(float *out, const float *X, const float *Y, int lane)
{
    for (int e = lane; e < K * K; e += 64) {
        int r = e / K, c = e - r * K;
        float s = 0.f;
#pragma unroll 9
        for (int k = 0; k < K; ++k) s = fmaf(X[r * LD + k], TRANS_Y ? Y[c * LD + k] : Y[k * LD + c], s);
        out[r * LD + c] = s;
    }
    __syncthreads();
}

__device__ int decode_members27(const uint32_t *mask, int p, const Geom27 &g, int *mem, int lane)
{
    int r = p / g.W, c = p - r * g.W;
    uint32_t m = (lane < g.words) ? mask[(size_t)p * g.words + lane] : 0u;
    int cntw = __popc(m), pre = cntw;
    for (int off = 1; off < 32; off <<= 1) {
        int v = __shfl_up(pre, off);
        if (lane >= off) pre += v;
    }
    int total = __shfl(pre, g.words - 1);
    int pos = pre - cntw;
    while (m) {
        int bit = __ffs(m) - 1;
        m &= m - 1;
        int k = lane * 32 + bit;
        int dl = k / g.side - g.b, dc = k % g.side - g.b;
        mem[pos++] = (r + dl) * g.W + (c + dc);
    }
    __syncthreads();
    return total;
}

// stage members [i0, i0+cn) of the similar set into the LDS chunk (pickColorPatchesFromColorImage :483-498)
__device__ inline void stage_chunk(float *chunk, const float *__restrict__ colors, const int *mem, int i0, int cn, int W, int lane)
{
    for (int t = lane; t < cn * K; t += 64) {
        int i = t / K, k = t - i * K, o = k / 3, ch = k - o * 3;
        int offp = (o / 3 - 1) * W + (o % 3 - 1);
        chunk[t] = colors[(size_t)(mem[i0 + i] + offp) * 3 + ch];
    }
    __syncthreads();
}

template <bool DBG>
__global__ __launch_bounds__(64) void k_bayes27(const float *__restrict__ colors, const float *__restrict__ pixcov,
                                                const uint32_t *__restrict__ mask, const int32_t *__restrict__ list,
                                                Geom27 g, float min_eig, float *sum, int32_t *cnt)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    float *A = lds, *V = A + MSZ, *Cm = V + MSZ, *Bm = Cm + MSZ;
    float *chunk = Bm + MSZ;
    float *noise = chunk + CHUNK * K;
    float *mean = noise + P * 6;
    float *fl = mean + K + 1;
    float *rc = fl + KP, *rs = rc + KP / 2;
    int *rp = reinterpret_cast<int *>(rs + KP / 2), *rq = rp + KP / 2;
    int *mem = rq + KP / 2;

    DBG_T(0);
    const int p = list[blockIdx.x];
    const int n = decode_members27(mask, p, g, mem, lane);
    DBG_T(1);
    const float n_inv = 1.f / (float)n;
    const int W = g.W;

    // computeNoiseCovPatchesMean (:400-419)
    if (lane < P * 6) {
        int o = lane / 6, j = lane - o * 6;
        int offp = (o / 3 - 1) * W + (o % 3 - 1);
        float acc = 0.f;
        for (int i = 0; i < n; ++i) acc += pixcov[(size_t)(mem[i] + offp) * 6 + j];
        noise[lane] = acc * n_inv;
    }
    __syncthreads();
    DBG_T(2);
    // empiricalMean (:500-509), members in order
    {
        float acc = 0.f;
        for (int i0 = 0; i0 < n; i0 += CHUNK) {
            int cn = min(CHUNK, n - i0);
            stage_chunk(chunk, colors, mem, i0, cn, W, lane);
            if (lane < K)
                for (int i = 0; i < cn; ++i) acc += chunk[i * K + lane];
            __syncthreads();
        }
        if (lane < K) mean[lane] = acc * n_inv;
        __syncthreads();
    }
    DBG_T(3);
    // centerPointCloud + empiricalCovarianceMatrix (:511-536): 378 lower-triangle entries, 6 per lane
    {
        constexpr int NE = (K * (K + 1) / 2 + 63) / 64;
        int er[NE], ec[NE];
        float acc[NE], mr[NE], mc[NE];
#pragma unroll
        for (int it = 0; it < NE; ++it) {
            int e = lane + it * 64, r = 0;
            if (e >= K * (K + 1) / 2) e = 0;
            while (e > r) { e -= r + 1; ++r; }
            er[it] = r; ec[it] = e; acc[it] = 0.f; mr[it] = mean[r]; mc[it] = mean[e];
        }
        for (int i0 = 0; i0 < n; i0 += CHUNK) {
            int cn = min(CHUNK, n - i0);
            stage_chunk(chunk, colors, mem, i0, cn, W, lane);
            for (int i = 0; i < cn; ++i) {
#pragma unroll
                for (int it = 0; it < NE; ++it) acc[it] += (chunk[i * K + er[it]] - mr[it]) * (chunk[i * K + ec[it]] - mc[it]);
            }
            __syncthreads();
        }
        const float inv = 1.f / (float)(n - 1);
#pragma unroll
        for (int it = 0; it < NE; ++it)
            if (lane + it * 64 < K * (K + 1) / 2) {
                float v = acc[it] * inv;
                A[er[it] * LD + ec[it]] = v; A[ec[it] * LD + er[it]] = v;
                Cm[er[it] * LD + ec[it]] = v; Cm[ec[it] * LD + er[it]] = v;
            }
        __syncthreads();
    }

    // ---- Step 1 (:421-436): M1 = clamp(C - N) + N ; Cinv1 = inverse(M1)
    DBG_T(4);
    add_noise27(A, noise, lane, -1.f);
    jacobi27(A, V, rc, rs, rp, rq, lane);
    DBG_T(5);
    rebuild27(Bm, A, V, fl, lane, false, 0.f);
    add_noise27(Bm, noise, lane, +1.f);
    DBG_T(6);
    inverse27(Bm, A, V, fl, rc, rs, rp, rq, lane, min_eig);
    DBG_T(7);
    // ---- Step 2 (:438-453): the Step-1 estimates are xhat = x - G (x - m) with G = N Cinv1, hence their empirical
    // mean is m and their empirical covariance is F C F^T, F = I - G
    noise_times27(V, noise, Bm, lane, true);       // V  = F
    matmul27<false>(A, V, Cm, lane);               // A  = F C
    matmul27<true>(Bm, A, V, lane);                // Bm = F C F^T
    for (int e = lane; e < K * K; e += 64) {       // exact symmetry (lower triangle wins)
        int r = e / K, c = e - r * K;
        if (r < c) Cm[r * LD + c] = Bm[c * LD + r];
    }
    __syncthreads();
    for (int e = lane; e < K * K; e += 64) { int r = e / K, c = e - r * K; if (r < c) Bm[r * LD + c] = Cm[r * LD + c]; }
    __syncthreads();
    add_noise27(Bm, noise, lane, +1.f);
    DBG_T(8);
    inverse27(Bm, A, V, fl, rc, rs, rp, rq, lane, min_eig);
    noise_times27(Cm, noise, Bm, lane, false);
    DBG_T(9);     // Cm = G2 = N Cinv2

    // ---- finalDenoisingMatrixMultiplication (:656-670) on the noisy patches centred on m, aggregateOutputPatches (:672-693)
    for (int i0 = 0; i0 < n; i0 += CHUNK) {
        int cn = min(CHUNK, n - i0);
        stage_chunk(chunk, colors, mem, i0, cn, W, lane);
        for (int u = lane; u < cn * P; u += 64) {
            int i = u / P, o = u - i * P;
            const float *x = chunk + i * K;
            const float *g0 = Cm + (3 * o) * LD, *g1 = g0 + LD, *g2 = g1 + LD;
            float y0 = 0.f, y1 = 0.f, y2 = 0.f;
#pragma unroll 9
            for (int c = 0; c < K; ++c) {
                float xc = x[c] - mean[c];
                y0 = fmaf(g0[c], xc, y0); y1 = fmaf(g1[c], xc, y1); y2 = fmaf(g2[c], xc, y2);
            }
            int q = mem[i0 + i] + (o / 3 - 1) * W + (o % 3 - 1);
            unsafeAtomicAdd(sum + (size_t)q * 3 + 0, x[3 * o] - y0);
            unsafeAtomicAdd(sum + (size_t)q * 3 + 1, x[3 * o + 1] - y1);
            unsafeAtomicAdd(sum + (size_t)q * 3 + 2, x[3 * o + 2] - y2);
            atomicAdd(cnt + q, 1);
        }
        __syncthreads();
    }
    DBG_T(10);
    if (DBG && blockIdx.x == 100 && lane == 0) bcd_dbg_cycles[11] = n;
}

} // namespace

size_t bcd_bayes27_lds_bytes(int b)
{
    int side = 2 * b + 1;
    return (size_t)(4 * MSZ + CHUNK * K + P * 6 + (K + 1) + KP + 4 * (KP / 2) + side * side) * sizeof(float);
}

hipError_t bcd_launch_bayes27(const float *colors, const float *pixcov, const uint32_t *mask, const int32_t *list, int nlist,
                              int W, int H, int b, float min_eig, float *sum, int32_t *cnt, hipStream_t st)
{
    if (nlist <= 0) return hipSuccess;
    Geom27 g;
    g.W = W; g.H = H; g.b = b; g.side = 2 * b + 1; g.words = (g.side * g.side + 31) / 32; g.maxS = g.side * g.side;
    if (g.words > 32) return hipErrorInvalidValue;
    if (getenv("BCD_DBG_BAYES")) {
        hipLaunchKernelGGL(k_bayes27<true>, dim3(nlist), dim3(64), bcd_bayes27_lds_bytes(b), st, colors, pixcov, mask, list, g, min_eig, sum, cnt);
        long long h[16];
        hipStreamSynchronize(st);
        hipMemcpyFromSymbol(h, HIP_SYMBOL(bcd_dbg_cycles), sizeof(h));
        fprintf(stderr, "bayes27 dbg n=%lld: decode %lld noise %lld mean %lld cov %lld jacobi %lld rebuild %lld inv1 %lld step2mm %lld inv2 %lld final %lld total %lld\n", h[11], h[1]-h[0], h[2]-h[1], h[3]-h[2], h[4]-h[3], h[5]-h[4], h[6]-h[5], h[7]-h[6], h[8]-h[7], h[9]-h[8], h[10]-h[9], h[10]-h[0]);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_bayes27<false>, dim3(nlist), dim3(64), bcd_bayes27_lds_bytes(b), st, colors, pixcov, mask, list, g, min_eig, sum, cnt);
    return hipGetLastError();
}
