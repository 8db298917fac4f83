// k_bayes.hip -- collaborative Bayesian denoising of the selected patches, one wavefront per processed pixel.
//
// Replaces DenoisingUnit::denoiseSelectedPatches (src/core/DenoisingUnit.cpp:388-453: computeNoiseCovPatchesMean,
// Step1, Step2), denoiseOnlyMainPatch (:455-481) and aggregateOutputPatches (:672-693).
// Everything lives in LDS: the n x K colour patches of the similar set (K = 3(2w+1)^2 = 27), the step-1
// estimates, and the K x K matrices.  The three SelfAdjointEigenSolver calls (:589,617) are a parallel
// (round-robin ordered) two-sided Jacobi iteration on the wavefront -- V f(Lambda) V^T is invariant to
// the eigenbasis, so any accurate symmetric eigensolver reproduces the reference within fp32 round-off.
// Sums that the reference accumulates sequentially (means, covariances, noise mean) are accumulated in
// the same order here.  Compiled with -ffp-contract=off; fmaf() is used explicitly where contraction is wanted.
#include "bcd_common.h"
#include <algorithm>

namespace {

struct BayesGeom {
    int W, H, w, b, side, words, P, K, KP, LD, maxS;
};

__device__ inline float wave_sum(float v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ---- parallel Jacobi on an LDS-resident symmetric matrix ------------------------------------------------
// A (K x K, leading dimension LD) is overwritten by (nearly) diag(lambda); V receives the eigenvectors in columns.
__device__ void jacobi_eig(float *A, float *V, float *rc, float *rs, int *rp, int *rq, int K, int KP, int LD, int lane)
{
    for (int e = lane; e < K * K; e += 64) {
        int r = e / K, c = e - r * K;
        V[r * LD + c] = (r == c) ? 1.f : 0.f;
    }
    __syncthreads();
    const int N1 = KP - 1, npairs = KP / 2;
    for (int sweep = 0; sweep < 14; ++sweep) {
        float off = 0.f, dg = 0.f;
        for (int e = lane; e < K * K; e += 64) {
            int r = e / K, c = e - r * K;
            float v = A[r * LD + c];
            if (r == c) dg = fmaf(v, v, dg); else off = fmaf(v, v, off);
        }
        off = wave_sum(off);
        dg = wave_sum(dg);
        if (off <= 1e-13f * dg || off == 0.f) break;
        for (int round = 0; round < N1; ++round) {
            if (lane < npairs) {
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
            // column rotations: A <- A J, V <- V J
            for (int t = lane; t < npairs * K; t += 64) {
                int k = t / K, row = t - k * K;
                float c = rc[k], s = rs[k];
                if (s != 0.f) {
                    int p = rp[k], q = rq[k];
                    float ap = A[row * LD + p], aq = A[row * LD + q];
                    A[row * LD + p] = fmaf(c, ap, -s * aq);
                    A[row * LD + q] = fmaf(s, ap, c * aq);
                    float vp = V[row * LD + p], vq = V[row * LD + q];
                    V[row * LD + p] = fmaf(c, vp, -s * vq);
                    V[row * LD + q] = fmaf(s, vp, c * vq);
                }
            }
            __syncthreads();
            // row rotations: A <- J^T A
            for (int t = lane; t < npairs * K; t += 64) {
                int k = t / K, col = t - k * K;
                float c = rc[k], s = rs[k];
                if (s != 0.f) {
                    int p = rp[k], q = rq[k];
                    float ap = A[p * LD + col], aq = A[q * LD + col];
                    A[p * LD + col] = fmaf(c, ap, -s * aq);
                    A[q * LD + col] = fmaf(s, ap, c * aq);
                }
            }
            __syncthreads();
        }
    }
}

// out = V f(lambda) V^T  (clampNegativeEigenValues :606-630 / inverseSymmetricMatrix :578-604)
__device__ void spectral_rebuild(float *out, const float *A, const float *V, float *fl, int K, int LD, int lane,
                                 bool inverse, float min_eig)
{
    for (int k = lane; k < K; k += 64) {
        float lam = A[k * LD + k];
        fl[k] = inverse ? 1.f / fmaxf(min_eig, lam) : fmaxf(0.f, lam);
    }
    __syncthreads();
    for (int e = lane; e < K * K; e += 64) {
        int r = e / K, c = e - r * K;
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += V[r * LD + k] * (fl[k] * V[c * LD + k]);
        out[r * LD + c] = s;
    }
    __syncthreads();
}

// +/- the block-diagonal noise covariance (add/substractCovMatPatch*Matrix :538-576)
__device__ void add_noise_blocks(float *M, const float *noise, int P, int LD, int lane, float sign)
{
    for (int t = lane; t < P * 9; t += 64) {
        int blk = t / 9, e = t - blk * 9, i = e / 3, j = e - i * 3;
        // xx,yy,zz,yz,xz,xy
        const int idx[3][3] = { { 0, 5, 4 }, { 5, 1, 3 }, { 4, 3, 2 } };
        M[(3 * blk + i) * LD + 3 * blk + j] += sign * noise[blk * 6 + idx[i][j]];
    }
    __syncthreads();
}

// empiricalMean (:500-509) of an n x K cloud
__device__ void cloud_mean(float *mean, const float *cloud, int n, int K, int lane)
{
    const float inv = 1.f / (float)n;
    for (int k = lane; k < K; k += 64) {
        float acc = 0.f;
        for (int i = 0; i < n; ++i) acc += cloud[i * K + k];
        mean[k] = acc * inv;
    }
    __syncthreads();
}

// centerPointCloud + empiricalCovarianceMatrix (:511-536) -> full symmetric A
__device__ void cloud_cov(float *A, const float *cloud, const float *mean, int n, int K, int LD, int lane)
{
    const float inv = 1.f / (float)(n - 1);
    const int nE = K * (K + 1) / 2;
    for (int e = lane; e < nE; e += 64) {
        int r = 0, rem = e;
        while (rem > r) { rem -= r + 1; ++r; }
        int c = rem;
        float mr = mean[r], mc = mean[c], acc = 0.f;
        for (int i = 0; i < n; ++i) acc += (cloud[i * K + r] - mr) * (cloud[i * K + c] - mc);
        acc *= inv;
        A[r * LD + c] = acc;
        A[c * LD + r] = acc;
    }
    __syncthreads();
}

// finalDenoisingMatrixMultiplication (:656-670): est = x - N (Cinv (x - mean)), one (member, patch pixel) unit per lane
template <bool SCATTER>
__device__ void apply_estimate(const float *X, const float *Cinv, const float *mean, const float *noise, const int *mem,
                               int n, const BayesGeom &g, int lane, float *Xd, float *sum, int32_t *cnt)
{
    const int K = g.K, P = g.P, LD = g.LD, pw = 2 * g.w + 1;
    for (int u = lane; u < n * P; u += 64) {
        int i = u / P, o = u - i * P;
        const float *x = X + i * K;
        float y0 = 0.f, y1 = 0.f, y2 = 0.f;
        const float *r0 = Cinv + (3 * o) * LD, *r1 = r0 + LD, *r2 = r1 + LD;
        for (int c = 0; c < K; ++c) {
            float xc = x[c] - mean[c];
            y0 += r0[c] * xc; y1 += r1[c] * xc; y2 += r2[c] * xc;
        }
        y0 *= -1.f; y1 *= -1.f; y2 *= -1.f;
        const float *n6 = noise + o * 6;
        float e0 = n6[0] * y0 + n6[5] * y1 + n6[4] * y2;
        float e1 = n6[5] * y0 + n6[1] * y1 + n6[3] * y2;
        float e2 = n6[4] * y0 + n6[3] * y1 + n6[2] * y2;
        e0 += x[3 * o]; e1 += x[3 * o + 1]; e2 += x[3 * o + 2];
        if (SCATTER) {
            int q = mem[i] + (o / pw - g.w) * g.W + (o % pw - g.w);
            unsafeAtomicAdd(sum + (size_t)q * 3 + 0, e0);
            unsafeAtomicAdd(sum + (size_t)q * 3 + 1, e1);
            unsafeAtomicAdd(sum + (size_t)q * 3 + 2, e2);
            atomicAdd(cnt + q, 1);
        } else {
            Xd[i * K + 3 * o] = e0; Xd[i * K + 3 * o + 1] = e1; Xd[i * K + 3 * o + 2] = e2;
        }
    }
    __syncthreads();
}

// decode the similarity mask of pixel p into the ordered member list (window row-major order,
// the order of m_similarPatchesCenters, DenoisingUnit.cpp:206-211); returns |S|
__device__ int decode_members(const uint32_t *mask, int p, const BayesGeom &g, int *mem, int lane)
{
    int r = p / g.W, c = p - r * g.W;
    const float inv_side = 1.f / (float)g.side;
    uint32_t m = (lane < g.words) ? mask[(size_t)p * g.words + lane] : 0u;
    int cntw = __popc(m), pre = cntw;
    for (int off = 1; off < 32; off <<= 1) {
        int v = __shfl_up(pre, off);
        if ((lane & 63) >= off) pre += v;
    }
    int total = __shfl(pre, 31 < g.words - 1 ? 31 : g.words - 1);
    int pos = pre - cntw;
    while (m) {
        int bit = __ffs(m) - 1;
        m &= m - 1;
        int k = lane * 32 + bit;
        // k / side without an integer division: (k + 0.5) / side is at least 0.5 / side away from an integer, k < 2^10
        int kl = (int)(((float)k + 0.5f) * inv_side), dl = kl - g.b, dc = k - kl * g.side - g.b;
        mem[pos++] = (r + dl) * g.W + (c + dc);
    }
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(64) void k_bayes_strong_generic(const float *__restrict__ colors, const float *__restrict__ pixcov,
                                                     const uint32_t *__restrict__ mask, const int32_t *__restrict__ list,
                                                     BayesGeom g, const int32_t *__restrict__ d_nlist, float min_eig, float *sum, int32_t *cnt,
                                                     float *gscratch)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const int K = g.K, P = g.P, KP = g.KP, LD = g.LD, pw = 2 * g.w + 1;
    int *mem = reinterpret_cast<int *>(lds);
    // the two n x K patch clouds live in LDS when they fit; for large patches (w = 2: K = 75) they go to a per-workgroup slice
    // of a global scratch buffer and the grid strides over the list
    float *X = gscratch ? gscratch + (size_t)blockIdx.x * 2 * g.maxS * K : lds + g.maxS;
    float *Xd = X + g.maxS * K;
    float *A = gscratch ? lds + g.maxS : Xd + g.maxS * K;
    float *V = A + KP * LD;
    float *Bm = V + KP * LD;
    float *noise = Bm + KP * LD;
    float *mean = noise + P * 6;
    float *fl = mean + K;
    float *rc = fl + K;
    float *rs = rc + KP / 2;
    int *rp = reinterpret_cast<int *>(rs + KP / 2);
    int *rq = rp + KP / 2;

  const int nlist = *d_nlist; // list length in device memory: no host round trip before the launch
  for (int item = blockIdx.x; item < nlist; item += gridDim.x) {
    __syncthreads();
    const int p = list[item];
    const int n = decode_members(mask, p, g, mem, lane);
    const float n_inv = 1.f / (float)n;

    // computeNoiseCovPatchesMean (:400-419)
    for (int v = lane; v < P * 6; v += 64) {
        int o = v / 6, j = v - o * 6;
        int offp = (o / pw - g.w) * g.W + (o % pw - g.w);
        float acc = 0.f;
        for (int i = 0; i < n; ++i) acc += pixcov[(size_t)(mem[i] + offp) * 6 + j];
        noise[v] = acc * n_inv;
    }
    // pickColorPatchesFromColorImage (:483-498)
    for (int t = lane; t < n * K; t += 64) {
        int i = t / K, k = t - i * K, o = k / 3, ch = k - o * 3;
        int offp = (o / pw - g.w) * g.W + (o % pw - g.w);
        X[t] = colors[(size_t)(mem[i] + offp) * 3 + ch];
    }
    __syncthreads();

    // ---- Step 1 (:421-436)
    cloud_mean(mean, X, n, K, lane);
    cloud_cov(A, X, mean, n, K, LD, lane);
    add_noise_blocks(A, noise, P, LD, lane, -1.f);
    jacobi_eig(A, V, rc, rs, rp, rq, K, KP, LD, lane);
    spectral_rebuild(Bm, A, V, fl, K, LD, lane, false, 0.f);      // clamp negative eigenvalues
    add_noise_blocks(Bm, noise, P, LD, lane, +1.f);
    for (int e = lane; e < K * K; e += 64) { int r = e / K, c = e - r * K; A[r * LD + c] = Bm[(r >= c ? r : c) * LD + (r >= c ? c : r)]; }
    __syncthreads();
    jacobi_eig(A, V, rc, rs, rp, rq, K, KP, LD, lane);
    spectral_rebuild(Bm, A, V, fl, K, LD, lane, true, min_eig);   // inverse
    apply_estimate<false>(X, Bm, mean, noise, mem, n, g, lane, Xd, nullptr, nullptr);

    // ---- Step 2 (:438-453)
    cloud_mean(mean, Xd, n, K, lane);
    cloud_cov(A, Xd, mean, n, K, LD, lane);
    add_noise_blocks(A, noise, P, LD, lane, +1.f);
    jacobi_eig(A, V, rc, rs, rp, rq, K, KP, LD, lane);
    spectral_rebuild(Bm, A, V, fl, K, LD, lane, true, min_eig);
    // aggregateOutputPatches (:672-693)
    apply_estimate<true>(X, Bm, mean, noise, mem, n, g, lane, nullptr, sum, cnt);
  }
}

// denoiseOnlyMainPatch (:455-481): average of the similar colour patches added to the main patch only.
// One lane per (fallback pixel, patch pixel): a wavefront takes 64 / P pixels at once (7 for the 3 x 3 patch), every lane walks
// the similar-set bitmask of its pixel in window order (the reference's member order, so the three channel sums are the
// reference's sequential sums) and adds its patch pixel of every member; no LDS, no barrier.  Four members are decoded per step
// so that their loads are in flight together.  |S| = 0 gives 0 * inf = NaN like the reference (assert compiled out, :212-213).
__global__ __launch_bounds__(64) void k_bayes_weak(const float *__restrict__ colors, const uint32_t *__restrict__ mask,
                                                   const int32_t *__restrict__ list, const int32_t *__restrict__ d_nlist, BayesGeom g,
                                                   float *sum, int32_t *cnt)
{
    const int lane = threadIdx.x, pw = 2 * g.w + 1, P = g.P;
    const int per_wave = P <= 64 ? 64 / P : 1;       // pixels per wavefront
    const int passes = P <= 64 ? 1 : (P + 63) / 64;  // patches larger than a wavefront: several patch pixels per lane
    const int slot = P <= 64 ? lane / P : 0;
    const int nlist = *d_nlist;
    const float inv_side = 1.f / (float)g.side;
    for (int base = blockIdx.x * per_wave; base < nlist; base += gridDim.x * per_wave) {
        const int item = base + slot;
        for (int pass = 0; pass < passes; ++pass) {
            const int o = (P <= 64 ? lane - slot * P : lane) + 64 * pass;
            if (slot >= per_wave || item >= nlist || o >= P) continue;
            const int p = list[item];
            const int offp = (o / pw - g.w) * g.W + (o % pw - g.w);
            const float *src = colors + (size_t)(p + offp) * 3;  // colour of patch pixel o of the member at window offset 0
            const uint32_t *mw = mask + (size_t)p * g.words;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            int n = 0;
            for (int wd = 0; wd < g.words; ++wd) {
                uint32_t m = mw[wd];
                while (m) {
                    int rel[4];
                    int got = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        rel[u] = 0;
                        if (m) {
                            const int k = wd * 32 + __ffs(m) - 1;
                            m &= m - 1;
                            // k / side without an integer division: (k + 0.5) / side is at least 0.5 / side away from an integer, k < 2^10
                            const int kl = (int)(((float)k + 0.5f) * inv_side), kc = k - kl * g.side;
                            rel[u] = ((kl - g.b) * g.W + (kc - g.b)) * 3;
                            got = u + 1;
                        }
                    }
                    float v[4][3];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        v[u][0] = src[rel[u]]; v[u][1] = src[rel[u] + 1]; v[u][2] = src[rel[u] + 2]; // (idle slots re-read the main patch)
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (u < got) { a0 += v[u][0]; a1 += v[u][1]; a2 += v[u][2]; }
                    n += got;
                }
            }
            const float n_inv = 1.f / (float)n;
            float *dst = sum + (size_t)(p + offp) * 3;
            unsafeAtomicAdd(dst, n_inv * a0);
            unsafeAtomicAdd(dst + 1, n_inv * a1);
            unsafeAtomicAdd(dst + 2, n_inv * a2);
            atomicAdd(cnt + p + offp, 1);
        }
    }
}

// The same for the default 3 x 3 patch, one lane per (fallback pixel, patch ROW): a patch row is nine consecutive floats (three
// pixels, RGB), so a lane loads 36 contiguous bytes per member instead of 12, and a wavefront carries 21 pixels instead of 7 -- a third
// of the load instructions and of the bitmask walks per pixel (the r2 kernel issued 10 % of the vector pipe's slots and waited on its
// gathers 55 % of the time: r3 counters).  Sums per component stay in window order (the reference's sequential sums, :462-474).
__global__ __launch_bounds__(64) void k_bayes_weak_w1(const float *__restrict__ colors, const uint32_t *__restrict__ mask,
                                                      const int32_t *__restrict__ list, const int32_t *__restrict__ d_nlist, BayesGeom g,
                                                      float *sum, int32_t *cnt)
{
    const int lane = threadIdx.x;
    constexpr int PER_WAVE = 21;
    const int slot = lane / 3, prow = lane - slot * 3;
    const int nlist = *d_nlist;
    const float inv_side = 1.f / (float)g.side;
    for (int base = blockIdx.x * PER_WAVE; base < nlist; base += gridDim.x * PER_WAVE) {
        const int item = base + slot;
        if (slot >= PER_WAVE || item >= nlist) continue;
        const int p = list[item];
        const int offp = (prow - 1) * g.W - 1;                     // first pixel of this lane's patch row, relative to a member
        const float *src = colors + ((long long)p + offp) * 3;
        const uint32_t *mw = mask + (size_t)p * g.words;
        float a[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) a[e] = 0.f;
        int n = 0;
        for (int wd = 0; wd < g.words; ++wd) {
            uint32_t m = mw[wd];
            while (m) {
                int rel[4];
                int got = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    rel[u] = 0;
                    if (m) {
                        const int k = wd * 32 + __ffs(m) - 1;
                        m &= m - 1;
                        const int kl = (int)(((float)k + 0.5f) * inv_side), kc = k - kl * g.side; // k / side, see k_bayes_weak
                        rel[u] = ((kl - g.b) * g.W + (kc - g.b)) * 3;
                        got = u + 1;
                    }
                }
                float v[4][9];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int e = 0; e < 9; ++e) v[u][e] = src[rel[u] + e];   // (idle slots re-read the main patch row)
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u < got) {
#pragma unroll
                        for (int e = 0; e < 9; ++e) a[e] += v[u][e];
                    }
                n += got;
            }
        }
        const float n_inv = 1.f / (float)n;
        float *dst = sum + ((long long)p + offp) * 3;
#pragma unroll
        for (int e = 0; e < 9; ++e) unsafeAtomicAdd(dst + e, n_inv * a[e]);
#pragma unroll
        for (int e = 0; e < 3; ++e) atomicAdd(cnt + p + offp + e, 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// denoiseOnlyMainPatch for the 3 x 3 patch, TILED: one workgroup per 16 x 16 pixel tile stages the colours of the tile plus the
// (b + 1)-pixel frame every member patch can reach -- (16 + 2 b + 2)^2 pixels, 10.8 KB for b = 6 -- in LDS with coalesced row
// loads, finds the tile's fallback pixels itself (state == IN and |S| < 28: no list needed), and averages their similar patches
// out of LDS.  Fallback pixels cluster (on the bench frames they are the pixels next to an edge), so neighbouring fallback pixels
// read the same members: the list kernels above gather every member patch from global memory once per fallback pixel -- 730 MB of
// 36-byte segments at 1080p, bound by the L1 tag rate (r3: 0.41 ms at 1080p scale 0, all wave slots of the chip taken meanwhile,
// which is what the full-estimate kernels started beside it were really waiting for).  The estimates are summed into an 18 x 18
// LDS window and flushed with one global atomic per touched value.
// One lane per (fallback pixel, patch row), sums in window order.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int WT = 16; // tile edge
__global__ __launch_bounds__(256) void k_bayes_weak_tile(const float *__restrict__ colors, const uint32_t *__restrict__ mask,
                                                         const uint8_t *__restrict__ state, const int32_t *__restrict__ nsim, int min_strong,
                                                         BayesGeom g, float *sum, int32_t *cnt, int row_begin, int row_end /* only pixels of these lines */,
                                                         const long long *__restrict__ skip_if /* optional: the launch does nothing when this word is not zero */)
{
    if (skip_if && *skip_if != 0) return; // (launched behind a marking batch whose outcome the host has not seen yet, like k_active_lists)
    extern __shared__ float lds[];
    const int b1 = g.b + 1, TW = WT + 2 * b1, row3 = TW * 3;
    float *win = lds;                                         // TW x TW x 3 colours
    float *accS = win + ((TW * row3 + 3) & ~3);               // (WT + 2)^2 x 3 sums
    int *accC = reinterpret_cast<int *>(accS + (WT + 2) * (WT + 2) * 3);
    uint16_t *wlist = reinterpret_cast<uint16_t *>(accC + (WT + 2) * (WT + 2)); // local indices of the tile's fallback pixels
    __shared__ int n_weak;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx0 = blockIdx.x * WT, ty0 = blockIdx.y * WT;
    const int W = g.W, H = g.H;
    // ---- fallback pixels of the tile
    const int lx = tid & (WT - 1), ly = tid >> 4, gx = tx0 + lx, gy = ty0 + ly;
    const bool inside = gx < W && gy < H;
    const long long pg = (long long)gy * W + gx;
    const bool weak = inside && gy >= row_begin && gy < row_end && state[pg] == BCD_ST_IN && nsim[pg] < min_strong;
    if (tid == 0) n_weak = 0;
    __syncthreads();
    {
        const unsigned long long bal = __ballot(weak);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&n_weak, __popcll(bal));
        base = __shfl(base, 0);
        if (weak) wlist[base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)tid;
    }
    __syncthreads();
    const int nw = n_weak;
    if (nw == 0) return;                                      // (uniform)
    // ---- colour window (cells outside the image are never part of a member's patch: clamped addresses, no branches) and accumulators
    for (int e = tid; e < TW * row3; e += 256) {
        const int wy = e / row3, r = e - wy * row3, px = r / 3, ch = r - px * 3;
        const int yy = min(max(ty0 - b1 + wy, 0), H - 1), xx = min(max(tx0 - b1 + px, 0), W - 1);
        win[e] = colors[((long long)yy * W + xx) * 3 + ch];
    }
    for (int e = tid; e < (WT + 2) * (WT + 2) * 4; e += 256) accS[e] = 0.f; // (sums and counts are contiguous)
    __syncthreads();
    // ---- one lane per (fallback pixel, patch row).  Wavefront w takes the pairs whose row of the 18 x 18 window, ply + prow, is w mod 4:
    // no two wavefronts ever touch the same accumulator row, so the sums need no atomics -- ds_add_f32 is served one lane at a time on
    // gfx950 (193 cycles of the CU's LDS pipe per wavefront instruction against 2.5 for a read and 4.7 for a write, tools/ubench/
    // lds_rate.hip; with nine of them per pass they were most of this kernel).  Inside a wavefront the LDS instructions are served in
    // program order; two lanes of ONE instruction must not meet, which they would on the same window row with columns less than 3 apart:
    // the adds are issued in nine turns, by (plx % 3, prow) -- the same turn and the same window row mean the same pixel row, so two such
    // pixels are at least 3 columns apart.
    uint16_t *pairs = wlist + 256 + wave * 192;              // (pixel | prow << 8) of this wavefront: at most 3/4 of 256 pixels
    int npairs = 0;
    for (int i0 = 0; i0 < nw; i0 += 64) {
        const int i = i0 + lane;
        int code = 0;
        bool take = false;
        if (i < nw) {
            const int lp = wlist[i], t = (wave - (lp >> 4)) & 3;
            take = t < 3;
            code = lp | (t << 8);
        }
        const unsigned long long bal = __ballot(take);
        if (take) pairs[npairs + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)code;
        npairs += __popcll(bal);
    }
    __syncthreads();
    const float inv_side = 1.f / (float)g.side;
    const int cell0 = (g.b & 15) * ((TW & 63) + 1);             // window offset (-b, -b) in cells
    for (int base = 0; base < npairs; base += 64) {
        const bool live = base + lane < npairs;
        const int code = pairs[live ? base + lane : 0];
        const int lp = code & 255, prow = code >> 8, plx = lp & (WT - 1), ply = lp >> 4;
        float a[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) a[e] = 0.f;
        if (live) {
            const long long p = (long long)(ty0 + ply) * W + tx0 + plx;
            // first float of this lane's patch row for the member at window offset (0, 0), i.e. the pixel itself
            const float *src = win + ((ply + b1 + prow - 1) * TW + plx + b1 - 1) * 3;
            const uint32_t *mw = mask + (size_t)p * g.words;
            int n = 0;
            for (int w0 = 0; w0 < g.words; w0 += 6) {
                uint32_t mreg[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) mreg[u] = (w0 + u < g.words) ? mw[w0 + u] : 0u; // the words of the pass travel together
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    uint32_t m = mreg[u];
                    while (m) {
                        const int k = (w0 + u) * 32 + __ffs(m) - 1;
                        m &= m - 1;
                        // k / side, see k_bayes_weak; the window cell as ONE multiply: kl (TW - side) + k - (b, b).  (The masks change nothing -- kl < 25,
                        // side <= 25, TW <= 42 -- they let the compiler see small factors and fold the two products; 32-bit integer multiplies are
                        // not slow on gfx950, tools/ubench/int_rate.hip: 4.6-5.4 cycles per wavefront instruction against 3.6 for v_add_f32.)
                        const int kl = (int)(((float)k + 0.5f) * inv_side) & 31, kc = k - kl * (g.side & 31);
                        const int cell = kl * (TW & 63) + kc - cell0;
                        const float *q = src + cell * 3;
#pragma unroll
                        for (int e = 0; e < 9; ++e) a[e] += q[e];
                        ++n;
                    }
                }
            }
            const float n_inv = 1.f / (float)n;
#pragma unroll
            for (int e = 0; e < 9; ++e) a[e] *= n_inv;
        }
        float *dS = accS + ((ply + prow) * (WT + 2) + plx) * 3;        // patch row prow of the pixel in the 18 x 18 window
        int *dC = accC + (ply + prow) * (WT + 2) + plx;
        const int turn = (plx % 3) * 3 + prow;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (live && turn == t) {
#pragma unroll
                for (int e = 0; e < 9; ++e) dS[e] += a[e];
            }
            asm volatile("" ::: "memory"); // (program order of the LDS accesses: the compiler only reasons about one lane)
        }
        if (live) {
#pragma unroll
            for (int e = 0; e < 3; ++e) atomicAdd(dC + e, 1);
        }
    }
    __syncthreads();
    // ---- flush: one global atomic per touched value
    for (int e = tid; e < (WT + 2) * (WT + 2); e += 256) {
        const int c = accC[e];
        if (c != 0) {
            const int wy = e / (WT + 2), wx = e - wy * (WT + 2);
            const long long q = (long long)(ty0 - 1 + wy) * W + (tx0 - 1 + wx);
            unsafeAtomicAdd(sum + q * 3, accS[e * 3]);
            unsafeAtomicAdd(sum + q * 3 + 1, accS[e * 3 + 1]);
            unsafeAtomicAdd(sum + q * 3 + 2, accS[e * 3 + 2]);
            atomicAdd(cnt + q, c);
        }
    }
}

BayesGeom make_geom(int W, int H, int w, int b)
{
    BayesGeom g;
    g.W = W; g.H = H; g.w = w; g.b = b;
    g.side = 2 * b + 1;
    g.words = (g.side * g.side + 31) / 32;
    g.P = (2 * w + 1) * (2 * w + 1);
    g.K = 3 * g.P;
    g.KP = g.K + (g.K & 1);
    g.LD = g.KP + 1;
    g.maxS = g.side * g.side;
    return g;
}

} // namespace

size_t bcd_bayes27_lds_bytes(int b);

// LDS of the generic kernel; with_clouds = false: patch clouds in global scratch
static size_t generic_lds_bytes(int w, int b, bool with_clouds)
{
    BayesGeom g = make_geom(0, 0, w, b);
    size_t f = (size_t)g.maxS + (with_clouds ? 2 * (size_t)g.maxS * g.K : 0) + 3 * (size_t)g.KP * g.LD + g.P * 6 + 2 * g.K + 4 * (g.KP / 2);
    return f * sizeof(float);
}

// smallest LDS footprint with which (w, b) can run (check_params)
size_t bcd_bayes_lds_bytes(int w, int b)
{
    if (w == 1) return bcd_bayes27_lds_bytes(b);
    size_t full = generic_lds_bytes(w, b, true);
    return full <= 160 * 1024 ? full : generic_lds_bytes(w, b, false);
}

// bytes of global scratch one workgroup needs when the clouds do not fit the LDS (0 otherwise)
size_t bcd_bayes_scratch_bytes_per_block(int w, int b)
{
    if (w == 1 || generic_lds_bytes(w, b, true) <= 160 * 1024) return 0;
    BayesGeom g = make_geom(0, 0, w, b);
    return 2 * (size_t)g.maxS * g.K * sizeof(float);
}

hipError_t bcd_launch_bayes_strong(const float *colors, const float *pixcov, const uint32_t *mask, const int32_t *list,
                                   const int32_t *d_nlist, int *d_work, int blocks, int W, int H, int w, int b, float min_eig, float *sum,
                                   int32_t *cnt, float *gscratch, size_t gscratch_bytes, hipStream_t st)
{
    if (blocks <= 0) return hipSuccess;
    if (w == 1) return hipErrorInvalidValue; // the 3 x 3 patch has its own launcher (bcd_launch_bayes27)
    BayesGeom g = make_geom(W, H, w, b);
    if (g.words > 32) return hipErrorInvalidValue;
    const size_t per_block = bcd_bayes_scratch_bytes_per_block(w, b);
    const size_t lds = generic_lds_bytes(w, b, per_block == 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (per_block) {
        if (!gscratch || gscratch_bytes < per_block) return hipErrorInvalidValue;
        blocks = (int)std::min<size_t>((size_t)blocks, gscratch_bytes / per_block);
    }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bayes_strong_generic), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_bayes_strong_generic, dim3(blocks), dim3(64), lds, st, colors, pixcov, mask, list, g, d_nlist, min_eig, sum, cnt,
                       per_block ? gscratch : nullptr);
    return hipGetLastError();
}

// the tiled fallback kernel (3 x 3 patches): needs no list -- it reads the marking states and |S| itself
hipError_t bcd_launch_bayes_weak_tiles(const float *colors, const uint32_t *mask, const uint8_t *state, const int32_t *nsim, int min_strong,
                                       int W, int H, int b, float *sum, int32_t *cnt, hipStream_t st, int row_begin, int row_end, const long long *skip_if)
{
    BayesGeom g = make_geom(W, H, 1, b);
    if (g.words > 32) return hipErrorInvalidValue;
    const int TW = WT + 2 * (b + 1);
    const size_t lds = (size_t)((TW * TW * 3 + 3) & ~3) * 4 + (size_t)(WT + 2) * (WT + 2) * 16 + (256 + 4 * 192) * sizeof(uint16_t);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_bayes_weak_tile, dim3((W + WT - 1) / WT, (H + WT - 1) / WT), dim3(256), lds, st, colors, mask, state, nsim, min_strong, g, sum, cnt, row_begin, row_end, skip_if);
    return hipGetLastError();
}

hipError_t bcd_launch_bayes_weak(const float *colors, const uint32_t *mask, const int32_t *list, const int32_t *d_nlist, int blocks,
                                 int W, int H, int w, int b, float *sum, int32_t *cnt, hipStream_t st)
{
    if (blocks <= 0) return hipSuccess;
    BayesGeom g = make_geom(W, H, w, b);
    if (g.words > 32) return hipErrorInvalidValue;
    if (w == 1) hipLaunchKernelGGL(k_bayes_weak_w1, dim3(blocks), dim3(64), 0, st, colors, mask, list, d_nlist, g, sum, cnt);
    else hipLaunchKernelGGL(k_bayes_weak, dim3(blocks), dim3(64), 0, st, colors, mask, list, d_nlist, g, sum, cnt);
    return hipGetLastError();
}
