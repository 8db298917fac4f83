/*
 * bcd_oracle.c -- CPU ORACLE for the BCD hot path.  TEST INFRASTRUCTURE ONLY.
 * See bcd_oracle.h for the pin status ("parity unpinned" for the Eigen-dependent core).
 *
 * Every function cites the reference lines (relative to /root/reference) it restates.
 * Compile: gcc -std=c99 -O2 -fopenmp -ffp-contract=off -fPIC -shared  (no -ffast-math, no -march=native)
 */
#include "bcd_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define IDX(l, c, W, D) (((size_t)(l) * (size_t)(W) + (size_t)(c)) * (size_t)(D))

static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

/* ------------------------------------------------------------------------------------------
 * a6  Denoiser::computePixelCovFromSampleCov   (src/core/Denoiser.cpp:357-373)
 *     cov_pixel[k] = cov_sample[k] * (1.f / n)   -- reciprocal first, then multiply.
 * ---------------------------------------------------------------------------------------- */
void bcdo_pixel_cov_from_sample_cov(const float *cov, const float *nsamp, int W, int H, float *out)
{
    for (size_t p = 0; p < (size_t)W * H; ++p) {
        float inv = 1.f / nsamp[p];
        for (int k = 0; k < 6; ++k) out[p * 6 + k] = cov[p * 6 + k] * inv;
    }
}

/* ------------------------------------------------------------------------------------------
 * a10 DenoisingUnit::pixelSummedHistogramDistance  (src/core/DenoisingUnit.cpp:360-386)
 *     sequential over bins; bins with b1+b2 <= 1 are skipped (":379  TEMPORARY" criterion);
 *     diff = n2*b1 - n1*b2;  sum += diff*diff / (n1*n2*(b1+b2)).
 * ---------------------------------------------------------------------------------------- */
static float pixel_summed_hist_distance(int *cnt, const float *h1, const float *h2, float n1, float n2, int D)
{
    int c = 0;
    float sum = 0.f;
    for (int k = 0; k < D; ++k) {
        float b1 = h1[k], b2 = h2[k];
        if (b1 + b2 <= 1.f) continue;
        ++c;
        float diff = n2 * b1 - n1 * b2;
        sum += diff * diff / (n1 * n2 * (b1 + b2));
    }
    *cnt = c;
    return sum;
}

/* a10 DenoisingUnit::histogramPatchDistance  (src/core/DenoisingUnit.cpp:336-358)
 *     patch pixels row-major; summed = ((0+s0)+s1)+...; result = summed / (float)(int total). */
float bcdo_patch_distance(const float *hist, const float *nsamp, int W, int H, int D, int w,
                          int pl, int pc, int ql, int qc)
{
    (void)H;
    float summed = 0.f;
    int total = 0;
    for (int dl = -w; dl <= w; ++dl)
        for (int dc = -w; dc <= w; ++dc) {
            int cnt;
            size_t i1 = (size_t)(pl + dl) * W + (pc + dc);
            size_t i2 = (size_t)(ql + dl) * W + (qc + dc);
            summed += pixel_summed_hist_distance(&cnt, hist + i1 * D, hist + i2 * D, nsamp[i1], nsamp[i2], D);
            total += cnt;
        }
    return summed / total; /* int -> float conversion, 0/0 = NaN */
}

/* search window of a main pixel: PixelWindow(width,height,center,radius=b,border=w)
 * (src/core/DenoisingUnit.cpp:200-203, include/bcd/core/DeepImage.hpp:181-196): clipped, not shifted. */
static void window_bounds(int W, int H, int w, int b, int pl, int pc, int *l0, int *l1, int *c0, int *c1)
{
    *l0 = imax(w, pl - b);
    *c0 = imax(w, pc - b);
    *l1 = imin(H - 1 - w, pl + b);
    *c1 = imin(W - 1 - w, pc + b);
}

void bcdo_window_distances(const float *hist, const float *nsamp, int W, int H, int D, int w, int b,
                           int pl, int pc, float *out)
{
    int side = 2 * b + 1;
    for (int k = 0; k < side * side; ++k) out[k] = INFINITY;
    int l0, l1, c0, c1;
    window_bounds(W, H, w, b, pl, pc, &l0, &l1, &c0, &c1);
    for (int l = l0; l <= l1; ++l)
        for (int c = c0; c <= c1; ++c)
            out[(l - pl + b) * side + (c - pc + b)] = bcdo_patch_distance(hist, nsamp, W, H, D, w, pl, pc, l, c);
}

int bcdo_similarity_masks(const float *hist, const float *nsamp, int W, int H, int D, int w, int b,
                          float tau, uint32_t *mask, int32_t *count, int nb_threads)
{
    int side = 2 * b + 1;
    int words = (side * side + 31) / 32;
    memset(mask, 0, (size_t)W * H * words * sizeof(uint32_t));
    if (count) memset(count, 0, (size_t)W * H * sizeof(int32_t));
#ifdef _OPENMP
    if (nb_threads > 0) omp_set_num_threads(nb_threads);
#endif
#pragma omp parallel for schedule(dynamic, 4)
    for (int pl = w; pl <= H - 1 - w; ++pl)
        for (int pc = w; pc <= W - 1 - w; ++pc) {
            int l0, l1, c0, c1, n = 0;
            window_bounds(W, H, w, b, pl, pc, &l0, &l1, &c0, &c1);
            uint32_t *m = mask + ((size_t)pl * W + pc) * words;
            for (int l = l0; l <= l1; ++l)
                for (int c = c0; c <= c1; ++c)
                    if (bcdo_patch_distance(hist, nsamp, W, H, D, w, pl, pc, l, c) <= tau) {
                        int k = (l - pl + b) * side + (c - pc + b);
                        m[k >> 5] |= 1u << (k & 31);
                        ++n;
                    }
            if (count) count[(size_t)pl * W + pc] = n;
        }
    return words;
}

/* ------------------------------------------------------------------------------------------
 * Eigen::SelfAdjointEigenSolver<MatrixXf>::compute  (third-party, absent: libigl/eigen, unpinned;
 * call sites src/core/DenoisingUnit.cpp:589-591,617-619).  Published algorithm: Householder
 * reduction to tridiagonal form, then the implicit symmetric QR iteration with Wilkinson shifts
 * (Golub & Van Loan, Matrix Computations, Alg. 8.3.1-8.3.3); only the LOWER triangle of the input
 * is read; eigenvalues are returned in ascending order with matching eigenvector columns.
 * Scalar type float, like MatrixXf.
 * ---------------------------------------------------------------------------------------- */
void bcdo_sym_eig(int n, const float *A, float *evals, float *evecs)
{
    float *M = (float *)malloc(sizeof(float) * n * n);
    float *Q = evecs;
    float *d = evals;
    float *e = (float *)calloc((size_t)n + 1, sizeof(float));
    float *v = (float *)malloc(sizeof(float) * n);
    float *p = (float *)malloc(sizeof(float) * n);

    for (int r = 0; r < n; ++r)
        for (int c = 0; c <= r; ++c) M[r * n + c] = M[c * n + r] = A[r * n + c];
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) Q[r * n + c] = (r == c) ? 1.f : 0.f;

    /* Householder tridiagonalisation: M <- Hk M Hk, Q <- Q Hk, Hk = I - beta v v^T */
    for (int k = 0; k + 2 < n; ++k) {
        float tail2 = 0.f;
        for (int i = k + 2; i < n; ++i) tail2 += M[i * n + k] * M[i * n + k];
        if (tail2 == 0.f) continue;
        float x0 = M[(k + 1) * n + k];
        float norm = sqrtf(x0 * x0 + tail2);
        float alpha = (x0 >= 0.f) ? -norm : norm;
        for (int i = 0; i < n; ++i) v[i] = 0.f;
        v[k + 1] = x0 - alpha;
        for (int i = k + 2; i < n; ++i) v[i] = M[i * n + k];
        float vtv = v[k + 1] * v[k + 1] + tail2;
        float beta = 2.f / vtv;
        /* p = beta * M v ; K = (beta/2) p.v ; w = p - K v ; M -= v w^T + w v^T */
        for (int i = 0; i < n; ++i) {
            float s = 0.f;
            for (int j = k + 1; j < n; ++j) s += M[i * n + j] * v[j];
            p[i] = beta * s;
        }
        float pv = 0.f;
        for (int j = k + 1; j < n; ++j) pv += p[j] * v[j];
        float K = 0.5f * beta * pv;
        for (int i = 0; i < n; ++i) p[i] -= K * v[i];
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) M[i * n + j] -= v[i] * p[j] + p[i] * v[j];
        /* exact structure of the reduced column */
        M[(k + 1) * n + k] = M[k * n + (k + 1)] = alpha;
        for (int i = k + 2; i < n; ++i) M[i * n + k] = M[k * n + i] = 0.f;
        for (int i = 0; i < n; ++i) {
            float s = 0.f;
            for (int j = k + 1; j < n; ++j) s += Q[i * n + j] * v[j];
            s *= beta;
            for (int j = k + 1; j < n; ++j) Q[i * n + j] -= s * v[j];
        }
    }
    for (int i = 0; i < n; ++i) d[i] = M[i * n + i];
    for (int i = 0; i + 1 < n; ++i) e[i] = M[(i + 1) * n + i];

    /* implicit QR iteration with Wilkinson shift on the unreduced trailing block [lo..hi] */
    const float eps = 1.1920929e-07f;
    int hi = n - 1, iter = 0;
    const int max_iter = 30 * n;
    while (hi > 0 && iter < max_iter) {
        for (int i = 0; i < hi; ++i)
            if (fabsf(e[i]) <= eps * (fabsf(d[i]) + fabsf(d[i + 1]))) e[i] = 0.f;
        while (hi > 0 && e[hi - 1] == 0.f) --hi;
        if (hi == 0) break;
        int lo = hi - 1;
        while (lo > 0 && e[lo - 1] != 0.f) --lo;
        ++iter;
        float dd = 0.5f * (d[hi - 1] - d[hi]);
        float ee = e[hi - 1];
        float mu = d[hi];
        if (dd == 0.f) mu -= fabsf(ee);
        else {
            float h = hypotf(dd, ee);
            mu -= ee * ee / (dd + (dd > 0.f ? h : -h));
        }
        float x = d[lo] - mu, z = e[lo];
        for (int k = lo; k < hi; ++k) {
            float r = hypotf(x, z);
            float c = 1.f, s = 0.f;
            if (r != 0.f) { c = x / r; s = -z / r; }
            if (k > lo) e[k - 1] = r;
            float a = d[k], bb = e[k], g = d[k + 1];
            d[k]     = c * c * a - 2.f * c * s * bb + s * s * g;
            d[k + 1] = s * s * a + 2.f * c * s * bb + c * c * g;
            e[k]     = c * s * (a - g) + (c * c - s * s) * bb;
            if (k < hi - 1) {
                z = -s * e[k + 1];
                e[k + 1] = c * e[k + 1];
                x = e[k];
            }
            for (int i = 0; i < n; ++i) {
                float q0 = Q[i * n + k], q1 = Q[i * n + k + 1];
                Q[i * n + k]     = c * q0 - s * q1;
                Q[i * n + k + 1] = s * q0 + c * q1;
            }
        }
    }
    /* ascending sort (selection), permuting eigenvector columns */
    for (int i = 0; i < n - 1; ++i) {
        int m = i;
        for (int j = i + 1; j < n; ++j) if (d[j] < d[m]) m = j;
        if (m != i) {
            float t = d[i]; d[i] = d[m]; d[m] = t;
            for (int r = 0; r < n; ++r) { t = Q[r * n + i]; Q[r * n + i] = Q[r * n + m]; Q[r * n + m] = t; }
        }
    }
    free(M); free(e); free(v); free(p);
}

/* ------------------------------------------------------------------------------------------
 * DenoisingUnit working set (src/core/DenoisingUnit.cpp:83-151)
 * ---------------------------------------------------------------------------------------- */
typedef struct Unit {
    int W, H, D, w, b, P, K, maxS;
    float tau, min_eig;
    const float *colors, *nsamp, *hist, *pixcov;
    float *sum;      /* W*H*3 */
    int32_t *cnt;    /* W*H   */
    uint8_t *marked; /* W*H (shared) */
    int *sl, *sc;    /* similar patch centres */
    int nS;
    float nS_inv;
    float *noise;    /* P*6 : mean noise cov blocks */
    float *X, *Xc, *Xd; /* maxS*K : noisy, centred, denoised patches */
    float *mean;     /* K */
    float *C, *Cl, *Ci, *tmpM, *ev, *evec; /* K*K ... */
    float *tmpv;     /* K */
} Unit;

static void unit_init(Unit *u, int W, int H, int D, const BcdoParams *prm,
                      const float *colors, const float *nsamp, const float *hist, const float *pixcov,
                      float *sum, int32_t *cnt, uint8_t *marked)
{
    u->W = W; u->H = H; u->D = D; u->w = prm->patch_radius; u->b = prm->search_radius;
    u->P = (2 * u->w + 1) * (2 * u->w + 1);
    u->K = 3 * u->P;
    u->maxS = (2 * u->b + 1) * (2 * u->b + 1);
    u->tau = prm->hist_dist_threshold; u->min_eig = prm->min_eigen_value;
    u->colors = colors; u->nsamp = nsamp; u->hist = hist; u->pixcov = pixcov;
    u->sum = sum; u->cnt = cnt; u->marked = marked;
    u->sl = (int *)malloc(sizeof(int) * u->maxS);
    u->sc = (int *)malloc(sizeof(int) * u->maxS);
    u->noise = (float *)malloc(sizeof(float) * u->P * 6);
    size_t sk = (size_t)u->maxS * u->K, kk = (size_t)u->K * u->K;
    u->X = (float *)malloc(sizeof(float) * sk);
    u->Xc = (float *)malloc(sizeof(float) * sk);
    u->Xd = (float *)malloc(sizeof(float) * sk);
    u->mean = (float *)malloc(sizeof(float) * u->K);
    u->C = (float *)malloc(sizeof(float) * kk);
    u->Cl = (float *)malloc(sizeof(float) * kk);
    u->Ci = (float *)malloc(sizeof(float) * kk);
    u->tmpM = (float *)malloc(sizeof(float) * kk);
    u->evec = (float *)malloc(sizeof(float) * kk);
    u->ev = (float *)malloc(sizeof(float) * u->K);
    u->tmpv = (float *)malloc(sizeof(float) * u->K);
}

static void unit_free(Unit *u)
{
    free(u->sl); free(u->sc); free(u->noise); free(u->X); free(u->Xc); free(u->Xd); free(u->mean);
    free(u->C); free(u->Cl); free(u->Ci); free(u->tmpM); free(u->evec); free(u->ev); free(u->tmpv);
}

/* a10 selectSimilarPatches (src/core/DenoisingUnit.cpp:196-219) */
static void select_similar(Unit *u, int pl, int pc)
{
    int l0, l1, c0, c1;
    window_bounds(u->W, u->H, u->w, u->b, pl, pc, &l0, &l1, &c0, &c1);
    u->nS = 0;
    for (int l = l0; l <= l1; ++l)
        for (int c = c0; c <= c1; ++c)
            if (bcdo_patch_distance(u->hist, u->nsamp, u->W, u->H, u->D, u->w, pl, pc, l, c) <= u->tau) {
                u->sl[u->nS] = l; u->sc[u->nS] = c; ++u->nS;
            }
    u->nS_inv = 1.f / u->nS; /* 1/0 = inf when nS == 0 (assert compiled out, :212-213) */
}

/* a14 denoiseOnlyMainPatch (src/core/DenoisingUnit.cpp:455-481): no marking */
static void denoise_only_main_patch(Unit *u, int pl, int pc)
{
    int K = u->K, w = u->w, W = u->W;
    for (int k = 0; k < K; ++k) u->mean[k] = 0.f;
    for (int i = 0; i < u->nS; ++i) {
        int k = 0;
        for (int dl = -w; dl <= w; ++dl)
            for (int dc = -w; dc <= w; ++dc) {
                const float *px = u->colors + IDX(u->sl[i] + dl, u->sc[i] + dc, W, 3);
                u->mean[k++] += px[0]; u->mean[k++] += px[1]; u->mean[k++] += px[2];
            }
    }
    int k = 0;
    for (int dl = -w; dl <= w; ++dl)
        for (int dc = -w; dc <= w; ++dc) {
            size_t pi = (size_t)(pl + dl) * W + (pc + dc);
            u->sum[pi * 3 + 0] += u->nS_inv * u->mean[k++];
            u->sum[pi * 3 + 1] += u->nS_inv * u->mean[k++];
            u->sum[pi * 3 + 2] += u->nS_inv * u->mean[k++];
            ++u->cnt[pi];
        }
}

/* a11 computeNoiseCovPatchesMean (src/core/DenoisingUnit.cpp:400-419) */
static void noise_cov_patches_mean(Unit *u)
{
    int w = u->w, W = u->W, P6 = u->P * 6;
    for (int k = 0; k < P6; ++k) u->noise[k] = 0.f;
    for (int i = 0; i < u->nS; ++i) {
        int k = 0;
        for (int dl = -w; dl <= w; ++dl)
            for (int dc = -w; dc <= w; ++dc) {
                const float *pc = u->pixcov + IDX(u->sl[i] + dl, u->sc[i] + dc, W, 6);
                for (int j = 0; j < 6; ++j) u->noise[k++] += pc[j];
            }
    }
    for (int k = 0; k < P6; ++k) u->noise[k] *= u->nS_inv;
}

/* pickColorPatchesFromColorImage (:483-498) */
static void pick_color_patches(Unit *u)
{
    int K = u->K, w = u->w, W = u->W;
    for (int i = 0; i < u->nS; ++i) {
        float *x = u->X + (size_t)i * K;
        int k = 0;
        for (int dl = -w; dl <= w; ++dl)
            for (int dc = -w; dc <= w; ++dc) {
                const float *px = u->colors + IDX(u->sl[i] + dl, u->sc[i] + dc, W, 3);
                x[k++] = px[0]; x[k++] = px[1]; x[k++] = px[2];
            }
    }
}

/* empiricalMean (:500-509) */
static void empirical_mean(float *mean, const float *cloud, int n, int K)
{
    for (int k = 0; k < K; ++k) mean[k] = 0.f;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < K; ++k) mean[k] += cloud[(size_t)i * K + k];
    float inv = 1.f / n;
    for (int k = 0; k < K; ++k) mean[k] *= inv;
}

/* centerPointCloud (:511-520) */
static void center_cloud(float *out, const float *mean, const float *cloud, int n, int K)
{
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < K; ++k) out[(size_t)i * K + k] = cloud[(size_t)i * K + k] - mean[k];
}

/* empiricalCovarianceMatrix (:522-536):  C(r,c) = sum_i x_i(r) x_i(c), then *= 1/(n-1) */
static void empirical_cov(float *C, const float *centred, int n, int K)
{
    for (int k = 0; k < K * K; ++k) C[k] = 0.f;
    for (int i = 0; i < n; ++i) {
        const float *x = centred + (size_t)i * K;
        for (int c = 0; c < K; ++c)
            for (int r = 0; r < K; ++r) C[r * K + c] += x[r] * x[c];
    }
    float inv = 1.f / (n - 1);
    for (int k = 0; k < K * K; ++k) C[k] *= inv;
}

/* add / substractCovMatPatch{To,From}Matrix (:538-576); block order xx,yy,zz,yz,xz,xy
 * (include/bcd/core/CovarianceMatrix.h:18-27) */
static void add_noise_blocks(float *M, const float *noise, int P, int K, float sign)
{
    for (int b = 0; b < P; ++b) {
        const float *n6 = noise + b * 6;
        int x = 3 * b, y = x + 1, z = x + 2;
        M[x * K + x] += sign * n6[0];
        M[y * K + y] += sign * n6[1];
        M[z * K + z] += sign * n6[2];
        M[y * K + z] += sign * n6[3]; M[z * K + y] += sign * n6[3];
        M[x * K + z] += sign * n6[4]; M[z * K + x] += sign * n6[4];
        M[x * K + y] += sign * n6[5]; M[y * K + x] += sign * n6[5];
    }
}

/* clampNegativeEigenValues (:606-630) and inverseSymmetricMatrix (:578-604):
 * out = V * (f(lambda) V^T), f = max(0,.) or 1/max(minEig,.) */
static void spectral_map(Unit *u, float *out, const float *in, int inverse)
{
    int K = u->K;
    bcdo_sym_eig(K, in, u->ev, u->evec);
    for (int r = 0; r < K; ++r) {
        float diag = inverse ? 1.f / fmaxf(u->min_eig, u->ev[r]) : fmaxf(0.f, u->ev[r]);
        for (int c = 0; c < K; ++c) u->tmpM[r * K + c] = diag * u->evec[c * K + r];
    }
    for (int r = 0; r < K; ++r)
        for (int c = 0; c < K; ++c) {
            float s = 0.f;
            for (int k = 0; k < K; ++k) s += u->evec[r * K + k] * u->tmpM[k * K + c];
            out[r * K + c] = s;
        }
}

/* finalDenoisingMatrixMultiplication (:656-670) + multiplyCovMatPatchByVector (:633-654) */
static void final_multiplication(Unit *u, float *out, const float *noisy, const float *inv, const float *centred)
{
    int K = u->K, P = u->P;
    for (int i = 0; i < u->nS; ++i) {
        const float *xc = centred + (size_t)i * K;
        for (int r = 0; r < K; ++r) {
            float s = 0.f;
            for (int c = 0; c < K; ++c) s += inv[r * K + c] * xc[c];
            u->tmpv[r] = s * -1.f;
        }
        float *o = out + (size_t)i * K;
        for (int b = 0; b < P; ++b) {
            const float *n6 = u->noise + b * 6;
            float vx = u->tmpv[3 * b], vy = u->tmpv[3 * b + 1], vz = u->tmpv[3 * b + 2];
            o[3 * b]     = n6[0] * vx + n6[5] * vy + n6[4] * vz;
            o[3 * b + 1] = n6[5] * vx + n6[1] * vy + n6[3] * vz;
            o[3 * b + 2] = n6[4] * vx + n6[3] * vy + n6[2] * vz;
        }
        for (int k = 0; k < K; ++k) o[k] += noisy[(size_t)i * K + k];
    }
}

/* a12 denoiseSelectedPatchesStep1 (:421-436) */
static void step1(Unit *u)
{
    int K = u->K, n = u->nS;
    pick_color_patches(u);
    empirical_mean(u->mean, u->X, n, K);
    center_cloud(u->Xc, u->mean, u->X, n, K);
    empirical_cov(u->C, u->Xc, n, K);
    add_noise_blocks(u->C, u->noise, u->P, K, -1.f);
    spectral_map(u, u->Cl, u->C, 0);
    add_noise_blocks(u->Cl, u->noise, u->P, K, +1.f);
    spectral_map(u, u->Ci, u->Cl, 1);
    final_multiplication(u, u->Xd, u->X, u->Ci, u->Xc);
}

/* a13 denoiseSelectedPatchesStep2 (:438-453): covariance from step-1 output, no clamp */
static void step2(Unit *u)
{
    int K = u->K, n = u->nS;
    empirical_mean(u->mean, u->Xd, n, K);
    center_cloud(u->Xc, u->mean, u->Xd, n, K);
    empirical_cov(u->C, u->Xc, n, K);
    memcpy(u->Cl, u->C, sizeof(float) * K * K);
    add_noise_blocks(u->Cl, u->noise, u->P, K, +1.f);
    spectral_map(u, u->Ci, u->Cl, 1);
    center_cloud(u->Xc, u->mean, u->X, n, K);
    final_multiplication(u, u->Xd, u->X, u->Ci, u->Xc);
}

/* a15 aggregateOutputPatches (:672-693) */
static void aggregate(Unit *u)
{
    int K = u->K, w = u->w, W = u->W;
    for (int i = 0; i < u->nS; ++i) {
        const float *x = u->Xd + (size_t)i * K;
        int k = 0;
        for (int dl = -w; dl <= w; ++dl)
            for (int dc = -w; dc <= w; ++dc) {
                size_t pi = (size_t)(u->sl[i] + dl) * W + (u->sc[i] + dc);
                u->sum[pi * 3 + 0] += x[k++];
                u->sum[pi * 3 + 1] += x[k++];
                u->sum[pi * 3 + 2] += x[k++];
                ++u->cnt[pi];
            }
        u->marked[(size_t)u->sl[i] * W + u->sc[i]] = 1;
    }
}

/* per-pixel uniform in [0,1) standing in for `rand() / RAND_MAX` of DenoisingUnit.cpp:168 (build-defined, see bcd_common.h) */
static uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
static float unit_hash(uint32_t idx, uint32_t seed)
{
    return (float)(mix32(idx * 0x9E3779B1u + mix32(seed ^ 0x51ed270bu)) >> 8) * (1.0f / 16777216.0f);
}
static uint32_t g_skip_seed = 0; /* set by the entry points (serial paths only) */

/* a9 denoisePatchAndSimilarPatches (:157-194): a marked pixel is skipped with probability skip_prob */
static void denoise_patch_and_similar(Unit *u, int pl, int pc, float skip_prob, const BcdoDiag *diag)
{
    size_t pi = (size_t)pl * u->W + pc;
    if (skip_prob != 0.f && u->marked[pi])
        if (skip_prob == 1.f || unit_hash((uint32_t)pi, g_skip_seed) < skip_prob) return;
    select_similar(u, pl, pc);
    if (diag && diag->processed) diag->processed[pi] = 1;
    if (diag && diag->nb_similar) diag->nb_similar[pi] = u->nS;
    if (u->nS < u->K + 1) {
        denoise_only_main_patch(u, pl, pc);
        if (diag && diag->fallback) diag->fallback[pi] = 1;
        return;
    }
    noise_cov_patches_mean(u);
    step1(u);
    step2(u);
    aggregate(u);
}

static int check_inputs(const float *a, const float *b, const float *c, const float *d, int W, int H, int D, const BcdoParams *prm);

/* ---- one-patch trace (fixture F3): every intermediate of denoiseSelectedPatches (:388-453) for ONE main pixel, so
 * that a regression of a single stage is localised instead of showing up only in a whole-frame norm.  Same calls as step1 / step2
 * above, with copies taken between them.  Returns |S| (>= 0), or -1 for bad arguments; when |S| < 3P + 1 (the fallback case) only
 * members, x and mean1 (= the fallback estimate of :455-481) are filled. */
int bcdo_patch_trace(const float *colors, const float *nsamp, const float *hist, const float *cov, int W, int H, int D,
                     const BcdoParams *prm, int pl, int pc, BcdoPatchTrace *t)
{
    if (check_inputs(colors, nsamp, hist, cov, W, H, D, prm) || !t) return -1;
    int w = prm->patch_radius;
    if (pl < w || pl > H - 1 - w || pc < w || pc > W - 1 - w) return -1;
    size_t npix = (size_t)W * H;
    float *pixcov = (float *)malloc(sizeof(float) * npix * 6);
    bcdo_pixel_cov_from_sample_cov(cov, nsamp, W, H, pixcov);
    float *sum = (float *)calloc(npix * 3, sizeof(float));
    int32_t *cnt = (int32_t *)calloc(npix, sizeof(int32_t));
    uint8_t *marked = (uint8_t *)calloc(npix, 1);
    Unit u;
    unit_init(&u, W, H, D, prm, colors, nsamp, hist, pixcov, sum, cnt, marked);
    select_similar(&u, pl, pc);
    const int K = u.K, n = u.nS;
    const size_t kk = sizeof(float) * K * K, nk = sizeof(float) * (size_t)n * K;
    if (t->members) for (int i = 0; i < n; ++i) t->members[i] = u.sl[i] * W + u.sc[i];
    if (n < K + 1) {
        /* denoiseOnlyMainPatch (:455-481): the estimate is the plain mean of the similar patches */
        pick_color_patches(&u);
        if (t->x && n > 0) memcpy(t->x, u.X, nk);
        if (t->mean1 && n > 0) empirical_mean(t->mean1, u.X, n, K);
    } else {
        noise_cov_patches_mean(&u);
        if (t->noise) memcpy(t->noise, u.noise, sizeof(float) * u.P * 6);
        /* Step 1 (:421-436) */
        pick_color_patches(&u);
        if (t->x) memcpy(t->x, u.X, nk);
        empirical_mean(u.mean, u.X, n, K);
        if (t->mean1) memcpy(t->mean1, u.mean, sizeof(float) * K);
        center_cloud(u.Xc, u.mean, u.X, n, K);
        empirical_cov(u.C, u.Xc, n, K);
        if (t->cov1) memcpy(t->cov1, u.C, kk);
        add_noise_blocks(u.C, u.noise, u.P, K, -1.f);
        if (t->cov1_minus_noise) memcpy(t->cov1_minus_noise, u.C, kk);
        spectral_map(&u, u.Cl, u.C, 0);
        if (t->clamped) memcpy(t->clamped, u.Cl, kk);
        add_noise_blocks(u.Cl, u.noise, u.P, K, +1.f);
        if (t->clamped_plus_noise) memcpy(t->clamped_plus_noise, u.Cl, kk);
        spectral_map(&u, u.Ci, u.Cl, 1);
        if (t->inverse1) memcpy(t->inverse1, u.Ci, kk);
        final_multiplication(&u, u.Xd, u.X, u.Ci, u.Xc);
        if (t->step1) memcpy(t->step1, u.Xd, nk);
        /* Step 2 (:438-453) */
        empirical_mean(u.mean, u.Xd, n, K);
        if (t->mean2) memcpy(t->mean2, u.mean, sizeof(float) * K);
        center_cloud(u.Xc, u.mean, u.Xd, n, K);
        empirical_cov(u.C, u.Xc, n, K);
        if (t->cov2) memcpy(t->cov2, u.C, kk);
        memcpy(u.Cl, u.C, kk);
        add_noise_blocks(u.Cl, u.noise, u.P, K, +1.f);
        spectral_map(&u, u.Ci, u.Cl, 1);
        if (t->inverse2) memcpy(t->inverse2, u.Ci, kk);
        center_cloud(u.Xc, u.mean, u.X, n, K);
        final_multiplication(&u, u.Xd, u.X, u.Ci, u.Xc);
        if (t->step2) memcpy(t->step2, u.Xd, nk);
    }
    unit_free(&u);
    free(pixcov); free(sum); free(cnt); free(marked);
    return n;
}

/* a16 finalAggregation tail (src/core/Denoiser.cpp:458-469) */
static void final_divide(const float *sum, const int32_t *cnt, size_t npix, float *out)
{
    for (size_t p = 0; p < npix; ++p) {
        float inv = 1.f / cnt[p];
        out[p * 3 + 0] = inv * sum[p * 3 + 0];
        out[p * 3 + 1] = inv * sum[p * 3 + 1];
        out[p * 3 + 2] = inv * sum[p * 3 + 2];
    }
}

static int check_inputs(const float *a, const float *b, const float *c, const float *d, int W, int H, int D, const BcdoParams *prm)
{
    if (!a || !b || !c || !d || !prm) return 1;           /* Denoiser.cpp:266-293 */
    if (W <= 0 || H <= 0 || D <= 0) return 2;             /* :294-320 (empty images) */
    if (W < 2 * prm->patch_radius + 1 || H < 2 * prm->patch_radius + 1) return 3;
    return 0;
}

int bcdo_denoise_mono(const float *colors, const float *nsamp, const float *hist, const float *cov,
                      int W, int H, int D, const BcdoParams *prm,
                      const int32_t *order, int64_t n_order,
                      float *out, const BcdoDiag *diag)
{
    int rc = check_inputs(colors, nsamp, hist, cov, W, H, D, prm);
    if (rc) return rc;
    size_t npix = (size_t)W * H;
    int w = prm->patch_radius;
    float *pixcov = (float *)malloc(sizeof(float) * npix * 6);
    bcdo_pixel_cov_from_sample_cov(cov, nsamp, W, H, pixcov);
    uint8_t *marked = (uint8_t *)calloc(npix, 1);
    if (diag) {
        if (diag->processed) memset(diag->processed, 0, npix);
        if (diag->fallback) memset(diag->fallback, 0, npix);
        if (diag->nb_similar) for (size_t p = 0; p < npix; ++p) diag->nb_similar[p] = -1;
    }
    int Wm = W - 2 * w, Hm = H - 2 * w;
    int64_t nmain = (int64_t)Wm * Hm;
    int serial = (prm->skip_probability != 0.f) || order != NULL;
    /* An ordered visit with nb_threads > 1 asked for explicitly: the SAME visit in three phases (below) -- what is decided one pixel
     * after the other stays sequential, the similar sets and the estimates are computed on all threads.  For frames too large for the
     * one-thread loop (the 1080p bench frame with its marking order). */
    int phased = serial && prm->nb_threads > 1;
    int nthreads = 1;
#ifdef _OPENMP
    if (!serial) {
        nthreads = prm->nb_threads > 0 ? prm->nb_threads : omp_get_max_threads();
    }
    if (phased) nthreads = prm->nb_threads;
#endif
    float **sums = (float **)malloc(sizeof(float *) * nthreads);
    int32_t **cnts = (int32_t **)malloc(sizeof(int32_t *) * nthreads);
    for (int t = 0; t < nthreads; ++t) {
        sums[t] = (float *)calloc(npix * 3, sizeof(float));
        cnts[t] = (int32_t *)calloc(npix, sizeof(int32_t));
    }
    if (phased) {
        /* phase 1 (parallel): S(p) of every main pixel, as bit masks -- selectSimilarPatches (:196-219) does not depend on the visit */
        int b = prm->search_radius, side = 2 * b + 1, words = (side * side + 31) / 32, K = 3 * (2 * w + 1) * (2 * w + 1);
        uint32_t *mask = (uint32_t *)malloc(sizeof(uint32_t) * npix * words);
        int32_t *count = (int32_t *)malloc(sizeof(int32_t) * npix);
        bcdo_similarity_masks(hist, nsamp, W, H, D, w, b, prm->hist_dist_threshold, mask, count, nthreads);
        /* phase 2 (sequential): the visit of denoisePatchAndSimilarPatches (:157-194) reduced to its decisions -- a pixel is processed unless
         * it was marked (and the skip draw says so); a pixel processed through the full estimate (|S| >= 3P + 1) marks its similar set (:690),
         * one processed through denoiseOnlyMainPatch marks nobody */
        int64_t n = order ? n_order : nmain, nt = 0;
        int32_t *todo = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
        for (int64_t i = 0; i < n; ++i) {
            int pl, pc;
            if (order) { pl = order[i] / W; pc = order[i] % W; }
            else { pl = w + (int)(i / Wm); pc = w + (int)(i % Wm); }
            size_t pi = (size_t)pl * W + pc;
            if (prm->skip_probability != 0.f && marked[pi])
                if (prm->skip_probability == 1.f || unit_hash((uint32_t)pi, prm->skip_seed) < prm->skip_probability) continue;
            todo[nt++] = (int32_t)pi;
            if (count[pi] >= K + 1) {
                const uint32_t *m = mask + pi * words;
                for (int k = 0; k < side * side; ++k)
                    if (m[k >> 5] >> (k & 31) & 1u) marked[(size_t)(pl + k / side - b) * W + (pc + k % side - b)] = 1;
            }
        }
        free(mask); free(count);
        /* phase 3 (parallel): the processed pixels' estimates with the per-pixel code of the serial visit (marks no longer matter: skip
         * probability 0), per-thread accumulators like Denoiser.cpp:149-159 */
#pragma omp parallel num_threads(nthreads)
        {
            int t = 0;
#ifdef _OPENMP
            t = omp_get_thread_num();
#endif
            Unit u;
            unit_init(&u, W, H, D, prm, colors, nsamp, hist, pixcov, sums[t], cnts[t], marked);
#pragma omp for schedule(dynamic, 16)
            for (int64_t i = 0; i < nt; ++i)
                denoise_patch_and_similar(&u, todo[i] / W, todo[i] % W, 0.f, diag);
            unit_free(&u);
        }
        free(todo);
        for (int t = 1; t < nthreads; ++t) {
            for (size_t k = 0; k < npix * 3; ++k) sums[0][k] += sums[t][k];
            for (size_t k = 0; k < npix; ++k) cnts[0][k] += cnts[t][k];
        }
    } else if (serial) {
        Unit u;
        g_skip_seed = prm->skip_seed;
        unit_init(&u, W, H, D, prm, colors, nsamp, hist, pixcov, sums[0], cnts[0], marked);
        int64_t n = order ? n_order : nmain;
        for (int64_t i = 0; i < n; ++i) {
            int pl, pc;
            if (order) { pl = order[i] / W; pc = order[i] % W; }
            else { pl = w + (int)(i / Wm); pc = w + (int)(i % Wm); }
            denoise_patch_and_similar(&u, pl, pc, prm->skip_probability, diag);
        }
        unit_free(&u);
    } else {
        /* m = 0: order-free; per-thread accumulators like Denoiser.cpp:149-159 */
#pragma omp parallel num_threads(nthreads)
        {
            int t = 0;
#ifdef _OPENMP
            t = omp_get_thread_num();
#endif
            Unit u;
            unit_init(&u, W, H, D, prm, colors, nsamp, hist, pixcov, sums[t], cnts[t], marked);
#pragma omp for schedule(dynamic, 64)
            for (int64_t i = 0; i < nmain; ++i)
                denoise_patch_and_similar(&u, w + (int)(i / Wm), w + (int)(i % Wm), 0.f, diag);
            unit_free(&u);
        }
        for (int t = 1; t < nthreads; ++t) { /* Denoiser.cpp:438-457 */
            for (size_t k = 0; k < npix * 3; ++k) sums[0][k] += sums[t][k];
            for (size_t k = 0; k < npix; ++k) cnts[0][k] += cnts[t][k];
        }
    }
    final_divide(sums[0], cnts[0], npix, out);
    for (int t = 0; t < nthreads; ++t) { free(sums[t]); free(cnts[t]); }
    free(sums); free(cnts); free(marked); free(pixcov);
    return 0;
}


/* Band form used by the multi-GPU tests: main pixels on lines [row_begin,row_end) only, raw accumulators out
 * (sum W*H*3, cnt W*H, zeroed here).  Serial; order == NULL -> scanline.  Same per-pixel code as above. */
int bcdo_accumulate_band(const float *colors, const float *nsamp, const float *hist, const float *cov,
                         int W, int H, int D, const BcdoParams *prm, int row_begin, int row_end,
                         const int32_t *order, int64_t n_order, float *sum, int32_t *cnt)
{
    int rc = check_inputs(colors, nsamp, hist, cov, W, H, D, prm);
    if (rc) return rc;
    size_t npix = (size_t)W * H;
    int w = prm->patch_radius;
    float *pixcov = (float *)malloc(sizeof(float) * npix * 6);
    bcdo_pixel_cov_from_sample_cov(cov, nsamp, W, H, pixcov);
    uint8_t *marked = (uint8_t *)calloc(npix, 1);
    memset(sum, 0, sizeof(float) * npix * 3);
    memset(cnt, 0, sizeof(int32_t) * npix);
    Unit u;
    g_skip_seed = prm->skip_seed;
    unit_init(&u, W, H, D, prm, colors, nsamp, hist, pixcov, sum, cnt, marked);
    if (order) {
        for (int64_t i = 0; i < n_order; ++i) {
            int pl = order[i] / W, pc = order[i] % W;
            if (pl >= row_begin && pl < row_end) denoise_patch_and_similar(&u, pl, pc, prm->skip_probability, NULL);
        }
    } else {
        for (int pl = imax(w, row_begin); pl <= imin(H - 1 - w, row_end - 1); ++pl)
            for (int pc = w; pc <= W - 1 - w; ++pc) denoise_patch_and_similar(&u, pl, pc, prm->skip_probability, NULL);
    }
    unit_free(&u);
    free(marked); free(pixcov);
    return 0;
}

/* processes exactly the listed main pixels (no marking logic at all): the per-band tail of the exact multi-GPU marking
 * tests, where the processed set has been decided globally beforehand.  sum / cnt are zeroed here. */
int bcdo_accumulate_pixels(const float *colors, const float *nsamp, const float *hist, const float *cov,
                           int W, int H, int D, const BcdoParams *prm, const int32_t *pixels, int64_t n, float *sum, int32_t *cnt)
{
    int rc = check_inputs(colors, nsamp, hist, cov, W, H, D, prm);
    if (rc) return rc;
    size_t npix = (size_t)W * H;
    float *pixcov = (float *)malloc(sizeof(float) * npix * 6);
    bcdo_pixel_cov_from_sample_cov(cov, nsamp, W, H, pixcov);
    uint8_t *marked = (uint8_t *)calloc(npix, 1);
    memset(sum, 0, sizeof(float) * npix * 3);
    memset(cnt, 0, sizeof(int32_t) * npix);
    Unit u;
    unit_init(&u, W, H, D, prm, colors, nsamp, hist, pixcov, sum, cnt, marked);
    for (int64_t i = 0; i < n; ++i) denoise_patch_and_similar(&u, pixels[i] / W, pixels[i] % W, 0.f, NULL);
    unit_free(&u);
    free(marked); free(pixcov);
    return 0;
}

void bcdo_finalize(const float *sum, const int32_t *cnt, int64_t npix, float *out) { final_divide(sum, cnt, (size_t)npix, out); }

/* Reference-style parallel m=1 (racy, NOT reproducible): strip reorder (Denoiser.cpp:382-414),
 * schedule(dynamic, chunk) (:164-172), shared mark image (:161-162).  Timing baseline only. */
int bcdo_denoise_mono_omp_racy(const float *colors, const float *nsamp, const float *hist, const float *cov,
                               int W, int H, int D, const BcdoParams *prm, float *out)
{
    int rc = check_inputs(colors, nsamp, hist, cov, W, H, D, prm);
    if (rc) return rc;
    size_t npix = (size_t)W * H;
    int w = prm->patch_radius;
    float *pixcov = (float *)malloc(sizeof(float) * npix * 6);
    bcdo_pixel_cov_from_sample_cov(cov, nsamp, W, H, pixcov);
    uint8_t *marked = (uint8_t *)calloc(npix, 1);
    int Wm = W - 2 * w, Hm = H - 2 * w;
    int64_t nmain = (int64_t)Wm * Hm;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = prm->nb_threads > 0 ? prm->nb_threads : omp_get_max_threads();
#endif
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * nmain);
    int64_t chunk = (int64_t)Wm * 2 * prm->search_radius;
    {   /* reorderPixelSetJumpNextChunk: even chunks then odd chunks; a trailing partial chunk keeps
           its scanline content in place (it is never copied over, :398-413) */
        int64_t nfull = nmain / chunk, o = 0;
        for (int64_t i = 0; i < nmain; ++i) {
            int pl = w + (int)(i / Wm), pc = w + (int)(i % Wm);
            order[i] = pl * W + pc;
        }
        if (nthreads > 1 && nfull > 0) {
            int32_t *copy = (int32_t *)malloc(sizeof(int32_t) * nmain);
            memcpy(copy, order, sizeof(int32_t) * nmain);
            for (int start = 0; start < 2; ++start)
                for (int64_t ch = start; ch < nfull; ch += 2) {
                    memcpy(order + o, copy + ch * chunk, sizeof(int32_t) * chunk);
                    o += chunk;
                }
            free(copy);
        }
    }
    float **sums = (float **)malloc(sizeof(float *) * nthreads);
    int32_t **cnts = (int32_t **)malloc(sizeof(int32_t *) * nthreads);
    for (int t = 0; t < nthreads; ++t) {
        sums[t] = (float *)calloc(npix * 3, sizeof(float));
        cnts[t] = (int32_t *)calloc(npix, sizeof(int32_t));
    }
    float skip = prm->skip_probability;
#pragma omp parallel num_threads(nthreads)
    {
        int t = 0;
#ifdef _OPENMP
        t = omp_get_thread_num();
#endif
        Unit u;
        unit_init(&u, W, H, D, prm, colors, nsamp, hist, pixcov, sums[t], cnts[t], marked);
#pragma omp for schedule(dynamic, chunk)
        for (int64_t i = 0; i < nmain; ++i)
            denoise_patch_and_similar(&u, order[i] / W, order[i] % W, skip, NULL);
        unit_free(&u);
    }
    for (int t = 1; t < nthreads; ++t) {
        for (size_t k = 0; k < npix * 3; ++k) sums[0][k] += sums[t][k];
        for (size_t k = 0; k < npix; ++k) cnts[0][k] += cnts[t][k];
    }
    final_divide(sums[0], cnts[0], npix, out);
    for (int t = 0; t < nthreads; ++t) { free(sums[t]); free(cnts[t]); }
    free(sums); free(cnts); free(marked); free(pixcov); free(order);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a18 pyramid reducers (src/core/MultiscaleDenoiser.cpp:243-334).
 * 2x2 block: p1=(2l,2c), p2=p1+(1,0) [next line], p3=p1+(0,1) [next column], p4=p1+(1,1),
 * each clamped to the image; evaluation order ((p1+p2)+p3)+p4.
 * ---------------------------------------------------------------------------------------- */
static void block_positions(int W, int H, int l, int c, size_t pos[4])
{
    int l1 = imin(2 * l + 1, H - 1), c1 = imin(2 * c + 1, W - 1);
    pos[0] = (size_t)(2 * l) * W + 2 * c;
    pos[1] = (size_t)l1 * W + 2 * c;
    pos[2] = (size_t)(2 * l) * W + c1;
    pos[3] = (size_t)l1 * W + c1;
}

void bcdo_downscale_sum(const float *in, int W, int H, int D, float *out)
{
    int w2 = W / 2, h2 = H / 2;
    for (int l = 0; l < h2; ++l)
        for (int c = 0; c < w2; ++c) {
            size_t p[4];
            block_positions(W, H, l, c, p);
            float *o = out + IDX(l, c, w2, D);
            for (int z = 0; z < D; ++z)
                o[z] = in[p[0] * D + z] + in[p[1] * D + z] + in[p[2] * D + z] + in[p[3] * D + z];
        }
}

void bcdo_downscale_avg(const float *in, int W, int H, int D, float *out)
{
    int w2 = W / 2, h2 = H / 2;
    for (int l = 0; l < h2; ++l)
        for (int c = 0; c < w2; ++c) {
            size_t p[4];
            block_positions(W, H, l, c, p);
            float *o = out + IDX(l, c, w2, D);
            for (int z = 0; z < D; ++z)
                o[z] = 0.25f * (in[p[0] * D + z] + in[p[1] * D + z] + in[p[2] * D + z] + in[p[3] * D + z]);
        }
}

/* downscaleSampleCovarianceSum (:297-334): w_i = (1/16) * nSum / n_i */
void bcdo_downscale_cov(const float *cov, const float *nsamp, int W, int H, int D, float *out)
{
    int w2 = W / 2, h2 = H / 2;
    const float sq = (1.f / 4.f) * (1.f / 4.f);
    for (int l = 0; l < h2; ++l)
        for (int c = 0; c < w2; ++c) {
            size_t p[4];
            block_positions(W, H, l, c, p);
            float n1 = nsamp[p[0]], n2 = nsamp[p[1]], n3 = nsamp[p[2]], n4 = nsamp[p[3]];
            float ns = n1 + n2 + n3 + n4;
            float w1 = sq * ns / n1, w2_ = sq * ns / n2, w3 = sq * ns / n3, w4 = sq * ns / n4;
            float *o = out + IDX(l, c, w2, D);
            for (int z = 0; z < D; ++z)
                o[z] = w1 * cov[p[0] * D + z] + w2_ * cov[p[1] * D + z] + w3 * cov[p[2] * D + z] + w4 * cov[p[3] * D + z];
        }
}

static int clamp_pos(int v, int maxp1) { return v <= 0 ? 0 : (v >= maxp1 ? maxp1 - 1 : v); }

/* a19 interpolate (:473-512): weights 9/16, 3/16 (two adjacent, summed first), 1/16 */
void bcdo_interpolate(const float *lo, int w, int h, int D, float *hi, int W, int H)
{
    const float wm = 9.f / 16.f, wa = 3.f / 16.f, wd = 1.f / 16.f;
    for (int ul = 0; ul < H; ++ul)
        for (int uc = 0; uc < W; ++uc) {
            int l = ul / 2, c = uc / 2;
            int al = clamp_pos(l + ((ul % 2) * 2 - 1), h);
            int ac = clamp_pos(c + ((uc % 2) * 2 - 1), w);
            int lc = imax(0, imin(l, h - 1)), cc = imax(0, imin(c, w - 1)); /* i_rImage.clamp() */
            const float *p1 = lo + IDX(lc, cc, w, D), *p2 = lo + IDX(lc, ac, w, D);
            const float *p3 = lo + IDX(al, cc, w, D), *p4 = lo + IDX(al, ac, w, D);
            float *o = hi + IDX(ul, uc, W, D);
            for (int z = 0; z < D; ++z) o[z] = wm * p1[z] + wa * (p2[z] + p3[z]) + wd * p4[z];
        }
}

/* a19 mergeOutputs (:453-466): hi -= up(down(hi)); hi += up(lo) */
void bcdo_merge(float *hi, int W, int H, const float *lo, int D)
{
    int w2 = W / 2, h2 = H / 2;
    float *tmp_lo = (float *)malloc(sizeof(float) * (size_t)w2 * h2 * D);
    float *tmp_hi = (float *)malloc(sizeof(float) * (size_t)W * H * D);
    bcdo_downscale_avg(hi, W, H, D, tmp_lo);          /* downscale() :514-539 == downscaleAverage */
    bcdo_interpolate(tmp_lo, w2, h2, D, tmp_hi, W, H);
    for (size_t k = 0; k < (size_t)W * H * D; ++k) hi[k] -= tmp_hi[k];
    bcdo_interpolate(lo, w2, h2, D, tmp_hi, W, H);
    for (size_t k = 0; k < (size_t)W * H * D; ++k) hi[k] += tmp_hi[k];
    free(tmp_lo); free(tmp_hi);
}

/* a17 MultiscaleDenoiser::denoise (src/core/MultiscaleDenoiser.cpp:31-136) */
int bcdo_denoise_multiscale(const float *colors, const float *nsamp, const float *hist, const float *cov,
                            int W, int H, int D, int nb_scales, const BcdoParams *prm,
                            const int32_t *const *orders, const int64_t *n_orders,
                            float *out, int racy_omp)
{
    int rc = check_inputs(colors, nsamp, hist, cov, W, H, D, prm);
    if (rc) return rc;
    if (nb_scales < 1 || nb_scales > 16) return 4;
    const float *col[16], *ns[16], *hs[16], *cv[16];
    float *own[16][4];
    float *outs[16];
    int ws[16], hs_[16];
    col[0] = colors; ns[0] = nsamp; hs[0] = hist; cv[0] = cov; ws[0] = W; hs_[0] = H; outs[0] = out;
    for (int s = 1; s < nb_scales; ++s) {
        int pw = ws[s - 1], ph = hs_[s - 1], w2 = pw / 2, h2 = ph / 2;
        ws[s] = w2; hs_[s] = h2;
        size_t np = (size_t)w2 * h2;
        own[s][0] = (float *)malloc(sizeof(float) * np * 3);
        own[s][1] = (float *)malloc(sizeof(float) * np);
        own[s][2] = (float *)malloc(sizeof(float) * np * D);
        own[s][3] = (float *)malloc(sizeof(float) * np * 6);
        outs[s] = (float *)malloc(sizeof(float) * np * 3);
        bcdo_downscale_avg(col[s - 1], pw, ph, 3, own[s][0]);
        bcdo_downscale_sum(ns[s - 1], pw, ph, 1, own[s][1]);
        bcdo_downscale_sum(hs[s - 1], pw, ph, D, own[s][2]);
        bcdo_downscale_cov(cv[s - 1], ns[s - 1], pw, ph, 6, own[s][3]);
        col[s] = own[s][0]; ns[s] = own[s][1]; hs[s] = own[s][2]; cv[s] = own[s][3];
    }
    for (int s = nb_scales - 1; s >= 0; --s) {
        BcdoParams ps = *prm;
        ps.skip_seed = prm->skip_seed + (uint32_t)s;
        if (racy_omp)
            rc = bcdo_denoise_mono_omp_racy(col[s], ns[s], hs[s], cv[s], ws[s], hs_[s], D, prm, outs[s]);
        else
            rc = bcdo_denoise_mono(col[s], ns[s], hs[s], cv[s], ws[s], hs_[s], D, &ps,
                                   orders ? orders[s] : NULL, n_orders ? n_orders[s] : 0, outs[s], NULL);
        if (rc) break;
        if (s < nb_scales - 1) bcdo_merge(outs[s], ws[s], hs_[s], outs[s + 1], 3);
    }
    for (int s = 1; s < nb_scales; ++s) {
        for (int k = 0; k < 4; ++k) free(own[s][k]);
        free(outs[s]);
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * a20 SpikeRemovalFilter::filter (src/core/SpikeRemovalFilter.cpp:18-116), float abs.
 * ---------------------------------------------------------------------------------------- */
static void avg_std(float *avg, float *sd, const float *v, int n)
{
    float total = 0.f;
    for (int i = 0; i < n; ++i) total += v[i];
    *avg = total / n;
    total = 0;
    for (int i = 0; i < n; ++i) total += (v[i] - *avg) * (v[i] - *avg);
    *sd = sqrtf(total / (n - 1));
}

static int median3d_index(const float *r, const float *g, const float *b, int n)
{
    int best = 0;
    float bestd = -1.f;
    for (int m = 0; m < n; ++m) {
        float tot = 0.f;
        for (int i = 0; i < n; ++i)
            tot += fabsf(r[i] - r[m]) + fabsf(g[i] - g[m]) + fabsf(b[i] - b[m]);
        if (bestd < 0 || tot < bestd) { bestd = tot; best = m; }
    }
    return best;
}

void bcdo_spike_filter(float *colors, float *nsamp, float *hist, float *cov, int W, int H, int D, float factor)
{
    size_t np = (size_t)W * H;
    float *c0 = (float *)malloc(sizeof(float) * np * 3), *n0 = (float *)malloc(sizeof(float) * np);
    float *h0 = (float *)malloc(sizeof(float) * np * D), *v0 = (float *)malloc(sizeof(float) * np * 6);
    memcpy(c0, colors, sizeof(float) * np * 3); memcpy(n0, nsamp, sizeof(float) * np);
    memcpy(h0, hist, sizeof(float) * np * D);   memcpy(v0, cov, sizeof(float) * np * 6);
    for (int l = 0; l < H; ++l)
        for (int c = 0; c < W; ++c) {
            int cl = l < 1 ? 1 : (l > H - 2 ? H - 2 : l);
            int cc = c < 1 ? 1 : (c > W - 2 ? W - 2 : c);
            float r[9], g[9], b[9], avg[3], sd[3];
            int k = 0;
            for (int nl = cl - 1; nl <= cl + 1; ++nl)
                for (int nc = cc - 1; nc <= cc + 1; ++nc) {
                    const float *px = c0 + IDX(nl, nc, W, 3);
                    r[k] = px[0]; g[k] = px[1]; b[k] = px[2]; ++k;
                }
            avg_std(&avg[0], &sd[0], r, 9); avg_std(&avg[1], &sd[1], g, 9); avg_std(&avg[2], &sd[2], b, 9);
            const float *me = c0 + IDX(l, c, W, 3);
            if (fabsf(me[0] - avg[0]) > factor * sd[0] || fabsf(me[1] - avg[1]) > factor * sd[1] ||
                fabsf(me[2] - avg[2]) > factor * sd[2]) {
                int m = median3d_index(r, g, b, 9);
                size_t src = (size_t)(cl - 1 + m / 3) * W + (cc - 1 + m % 3), dst = (size_t)l * W + c;
                memcpy(colors + dst * 3, c0 + src * 3, sizeof(float) * 3);
                nsamp[dst] = n0[src];
                memcpy(hist + dst * D, h0 + src * D, sizeof(float) * D);
                memcpy(cov + dst * 6, v0 + src * 6, sizeof(float) * 6);
            }
        }
    free(c0); free(n0); free(h0); free(v0);
}

/* ------------------------------------------------------------------------------------------
 * SamplesAccumulator::addSample + computeSampleStatistics (src/core/SamplesAccumulator.cpp:44-141)
 * ---------------------------------------------------------------------------------------- */
void bcdo_accumulate(const float *samples, int64_t n, int W, int H, int nbins, float gamma, float maxval,
                     float *nsamp, float *mean, float *cov, float *hist)
{
    size_t np = (size_t)W * H;
    int D = 3 * nbins;
    float *sqw = (float *)calloc(np, sizeof(float));
    memset(nsamp, 0, sizeof(float) * np); memset(mean, 0, sizeof(float) * np * 3);
    memset(cov, 0, sizeof(float) * np * 6); memset(hist, 0, sizeof(float) * np * D);
    const float sat = 2.f;
    for (int64_t i = 0; i < n; ++i) {
        const float *s = samples + i * 6;
        int l = (int)s[0], c = (int)s[1];
        float R = s[2], G = s[3], B = s[4], wgt = s[5];
        size_t p = (size_t)l * W + c;
        nsamp[p] += wgt;
        sqw[p] += wgt * wgt;
        mean[p * 3 + 0] += wgt * R; mean[p * 3 + 1] += wgt * G; mean[p * 3 + 2] += wgt * B;
        cov[p * 6 + 0] += wgt * R * R; cov[p * 6 + 1] += wgt * G * G; cov[p * 6 + 2] += wgt * B * B;
        cov[p * 6 + 3] += wgt * G * B; cov[p * 6 + 4] += wgt * R * B; cov[p * 6 + 5] += wgt * R * G;
        const float smp[3] = { R, G, B };
        for (int ch = 0; ch < 3; ++ch) {
            float v = smp[ch];
            v = (v > 0 ? v : 0);
            if (gamma > 1) v = powf(v, 1.f / gamma);
            if (maxval > 0) v = v / maxval;
            v = v > sat ? sat : v;
            float fi = v * (nbins - 2);
            int fl = (int)fi, ce;
            float cw, fw;
            if (fl < nbins - 2) { ce = fl + 1; cw = fi - fl; fw = 1.0f - cw; }
            else { fl = nbins - 2; ce = fl + 1; cw = (v - 1.0f) / (sat - 1.f); fw = 1.0f - cw; }
            hist[p * D + ch * nbins + fl] += wgt * fw;
            hist[p * D + ch * nbins + ce] += wgt * cw;
        }
    }
    for (size_t p = 0; p < np; ++p) { /* computeSampleStatistics :109-141 */
        float ws = nsamp[p], sq = sqw[p], inv = 1.f / ws, m[3], cv[6];
        for (int i = 0; i < 3; ++i) { m[i] = inv * mean[p * 3 + i]; mean[p * 3 + i] = m[i]; }
        for (int i = 0; i < 6; ++i) cv[i] = cov[p * 6 + i] * inv;
        cv[0] -= m[0] * m[0]; cv[1] -= m[1] * m[1]; cv[2] -= m[2] * m[2];
        cv[3] -= m[1] * m[2]; cv[4] -= m[0] * m[2]; cv[5] -= m[0] * m[1];
        float bias = 1.f / (1 - sq / (ws * ws));
        for (int i = 0; i < 6; ++i) cov[p * 6 + i] = cv[i] * bias;
    }
    free(sqw);
}

/* checkAndPutToZeroNegativeInfNaNValues (src/cli/main.cpp:389-420) */
void bcdo_zero_bad_values(float *img, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        float v = img[i];
        if (v < 0 || isnan(v) || isinf(v)) img[i] = 0.f;
    }
}
