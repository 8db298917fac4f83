// k_bayes27.hip -- Bayesian patch estimate for the default patch radius w = 1 (K = 27): persistent wavefronts (one
// workgroup = one wavefront, 13.2 KB of LDS and 148 VGPRs: 12 per CU), one processed pixel at a time.
//
// Same mathematics as DenoisingUnit::denoiseSelectedPatches (src/core/DenoisingUnit.cpp:388-453) and
// aggregateOutputPatches (:672-693), reorganised for the GPU:
//   * the noise mean and the colour mean are summed straight from global memory; the similar patches are then streamed
//     through a 29-member LDS chunk twice (covariance, final estimate) instead of being held as n x 27 clouds; sums keep
//     the reference's sequential member order;
//   * the covariance, the spectral rebuild, the Step-2 products and the final estimate run on the f32 matrix core
//     (v_mfma_f32_32x32x2_f32: exact f32, a chain of fma over k);
//   * Step 2's covariance of the Step-1 estimates (:441-443) is obtained without touching the members
//     again: the Step-1 estimate is affine, xhat = x - G (x - m), G = N Cinv1, so its empirical mean is m
//     and its empirical covariance is F C F^T with F = I - G  (exact identity, fp32 round-off apart);
//   * clampNegativeEigenValues (:606-630) is a parallel two-sided Jacobi eigendecomposition, rows in registers;
//   * inverseSymmetricMatrix (:578-604) = V diag(1/max(minEig, lambda)) V^T equals the plain inverse
//     whenever lambda_min >= minEig.  The inverse is computed with the symmetric sweep operator and
//     accepted only if every pivot is positive and ||M^-1||_F * minEig <= 1 (which proves
//     lambda_min(M) >= minEig); otherwise the spectral form is evaluated with the Jacobi solver.
#include "bcd_common.h"
#include <cstdio>
#include <cstdlib>

// per-phase cycle counters of one wavefront, filled only by the DBG instantiation (BCD_DBG_BAYES=1: printed after the launch)
__device__ long long bcd_dbg_cycles[24];
__device__ int bcd_dbg_item = 100; // which work item the DBG instantiation stamps (BCD_DBG_ITEM)

namespace {

#define DBG_T(i) do { if (DBG && item == bcd_dbg_item && lane == 0) bcd_dbg_cycles[i] = __builtin_readcyclecounter(); } while (0)

constexpr int K = 27, KP = 28, LD = 29, P = 9, MSZ = KP * KP, CHUNK = MSZ / K; // matrix buffer: 784 floats; 29 members per staging chunk

typedef float v16f __attribute__((ext_vector_type(16)));

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

// ---- parallel two-sided Jacobi, Brent-Luk ordering, rows held in registers ------------------------------
// Slots 0..27 (27 = zero padding) are paired (2i, 2i+1).  One round: lanes i < 14 compute the rotation of pair i from
// LDS; lanes 0..27 load ONE row of A each (7 x ds_read_b128), lanes 32..58 keep their row of V in registers; every lane applies
// the 14 column-pair rotations on registers with static indices, lane pairs (2i, 2i+1) exchange their rows with a DPP
// quad_perm (row rotation of A), and the rows of A are written back in place at their Brent-Luk permuted
// position: slot s moves to sigma(s), 0->0, 1->2, 2i->2i+2, 26->27, 2i+1->2i-1.  After 27 rounds every pair of slots
// has met once (one sweep).  Eigenvalues = diagonal of A (in slot order), eigenvectors = columns of V (same order):
// V f(diag) V^T needs no bookkeeping of the permutation.  Matrices here use a leading dimension of 28 floats.
constexpr int JLD = 28;
static_assert(MSZ >= KP * JLD && MSZ >= (K - 1) * LD + K, "a matrix buffer holds 28 x 28 (Jacobi layout) or 27 rows of stride 29");

__device__ inline float dpp_xor1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
}

__device__ inline int sigma_slot(int s)
{
    return s == 0 ? 0 : (s == 1 ? 2 : (s == 26 ? 27 : ((s & 1) ? s - 2 : s + 2)));
}

// A0 holds the symmetric input (JLD layout, row/col 27 zero).  The kernel is bound by LDS bandwidth (measured: 21 KB per
// wavefront and round with everything staged through LDS), so only what must change lanes goes through LDS:
//   * rows of A (lanes 0..27) are stored at their Brent-Luk permuted slot and reloaded -- in place: the workgroup is a single
//     wavefront, every row is in registers before the first permuted row is written back;
//   * rows of V (lanes 32..58) never move between lanes and stay in registers for the whole solve;
//   * the 14 rotations (c, s) are broadcast with v_readlane (wave-uniform SGPR operands of the column rotations).
template <bool DBG>
__device__ void jacobi27(float *A0, float *V0, float *cs, int lane, int item)
{
    const bool isA = lane < KP, isV = lane >= 32 && lane < 32 + K;
    const int vrow = lane - 32;
    const float *src = A0 + (isA ? lane : 0) * JLD;
    float *dst = A0 + (isA ? sigma_slot(lane) : 0) * JLD;
    float row[JLD];
#pragma unroll
    for (int k = 0; k < JLD; ++k) row[k] = (isV && k == vrow) ? 1.f : 0.f; // V = identity
    for (int sweep = 0; sweep < 12; ++sweep) {
        float off = 0.f, dg = 0.f;
        for (int e = lane; e < KP * JLD; e += 64) {
            int r = e / JLD, c = e - r * JLD;
            float v = A0[e];
            if (r == c) dg = fmaf(v, v, dg); else off = fmaf(v, v, off);
        }
        off = wsum(off);
        dg = wsum(dg);
        if (DBG && item == bcd_dbg_item && lane == 0) bcd_dbg_cycles[12 + sweep] = (long long)(1e18f * off / dg);
        if (off <= 1e-13f * dg) break;
        for (int round = 0; round < KP - 1; ++round) {
            // lanes 0..13: rotation of the slot pair (2 lane, 2 lane + 1); the other lanes compute on a harmless copy of pair 0
            float c = 1.f, s = 0.f;
            {
                const int p = lane < KP / 2 ? 2 * lane : 0, q = p + 1;
                const float apq = A0[p * JLD + q], app = A0[p * JLD + p], aqq = A0[q * JLD + q];
                if (apq != 0.f) {
                    // 1-ulp hardware reciprocal / sqrt / rsqrt: a rotation only has to be orthogonal to working
                    // precision (c^2 + s^2 = 1 +- 1e-7), not the exact minimiser
                    float theta = (aqq - app) * __builtin_amdgcn_rcpf(2.f * apq);
                    float t = __builtin_amdgcn_rcpf(fabsf(theta) + __builtin_amdgcn_sqrtf(fmaf(theta, theta, 1.f)));
                    t = theta < 0.f ? -t : t;
                    c = __builtin_amdgcn_rsqf(fmaf(t, t, 1.f));
                    s = t * c;
                    if (!(fabsf(theta) < 1e18f)) { c = 1.f; s = 0.f; } // theta^2 overflows: the rotation is the identity to fp32
                }
            }
            if (lane < KP / 2) reinterpret_cast<float2 *>(cs)[lane] = make_float2(c, s);
            if (isA) {
#pragma unroll
                for (int q4 = 0; q4 < JLD / 4; ++q4) {
                    float4 v = reinterpret_cast<const float4 *>(src)[q4];
                    row[4 * q4] = v.x; row[4 * q4 + 1] = v.y; row[4 * q4 + 2] = v.z; row[4 * q4 + 3] = v.w;
                }
            }
            __syncthreads();
            float rot[JLD];
#pragma unroll
            for (int q4 = 0; q4 < JLD / 4; ++q4) {
                float4 w4 = reinterpret_cast<const float4 *>(cs)[q4];
                rot[4 * q4] = w4.x; rot[4 * q4 + 1] = w4.y; rot[4 * q4 + 2] = w4.z; rot[4 * q4 + 3] = w4.w;
            }
            // column rotations: (x, y) <- (c x - s y, s x + c y) for every slot pair
#pragma unroll
            for (int j = 0; j < KP / 2; ++j) {
                const float cj = rot[2 * j], sj = rot[2 * j + 1];
                const float x = row[2 * j], y = row[2 * j + 1];
                row[2 * j] = fmaf(cj, x, -sj * y);
                row[2 * j + 1] = fmaf(sj, x, cj * y);
            }
            // row rotations of A: rows (2i, 2i+1) live in lanes (2i, 2i+1) and take the rotation of lane i; the V lanes apply
            // the identity (1, 0), exact because every entry of V is finite
            const float pc = __shfl(c, lane >> 1), ps = __shfl(s, lane >> 1);
            const float mc = isA ? pc : 1.f, ms = isA ? ((lane & 1) ? ps : -ps) : 0.f;
            float out[JLD];
#pragma unroll
            for (int k = 0; k < JLD; ++k)
                out[sigma_slot(k)] = fmaf(mc, row[k], dpp_xor1(row[k]) * ms); // + Brent-Luk column move (register renaming)
#pragma unroll
            for (int k = 0; k < JLD; ++k) row[k] = out[k];
            if (isA) {
#pragma unroll
                for (int q4 = 0; q4 < JLD / 4; ++q4)
                    reinterpret_cast<float4 *>(dst)[q4] = make_float4(out[4 * q4], out[4 * q4 + 1], out[4 * q4 + 2], out[4 * q4 + 3]);
            }
            __syncthreads();
        }
    }
    // eigenvectors: the V lanes publish their rows (columns are back in slot order after every complete sweep)
    if (isV) {
#pragma unroll
        for (int q4 = 0; q4 < JLD / 4; ++q4)
            reinterpret_cast<float4 *>(V0 + vrow * JLD)[q4] = make_float4(row[4 * q4], row[4 * q4 + 1], row[4 * q4 + 2], row[4 * q4 + 3]);
    }
    __syncthreads();
}

// out (LD layout) = V f(lambda) V^T ; f = max(0,.) (clamp) or 1/max(minEig,.) (inverse); A, V in the Jacobi (JLD) layout
__device__ void rebuild27(float *out, const float *A, const float *V, float *fl, int lane, bool inverse, float min_eig);
// JLD-layout copy of the lower triangle of M (LD layout), mirrored, padding row/column zeroed (what Eigen's
// SelfAdjointEigenSolver reads)
__device__ void to_jacobi_layout(float *J, const float *M, int lane)
{
    for (int e = lane; e < KP * JLD; e += 64) {
        int r = e / JLD, c = e - r * JLD;
        J[e] = (r < K && c < K) ? M[(r >= c ? r : c) * LD + (r >= c ? c : r)] : 0.f;
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

// in-place inverse of the symmetric positive definite M (LD layout) by the sweep operator, rows held in registers:
// lane r owns row r; at step k the pivot row is broadcast with v_readlane (k is a compile-time constant), so the 27
// steps need no LDS traffic and no barrier.  Returns false (wave-uniform) if a pivot is not positive or the bound
// ||M^-1||_F * min_eig <= 1 fails (then lambda_min >= min_eig is not proven); M is then unspecified.
__device__ inline float bcast_lane(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }

__device__ bool sweep_inverse27(float *M, int lane, float min_eig)
{
    const int r = lane < K ? lane : 0;
    float m[K];
#pragma unroll
    for (int c = 0; c < K; ++c) m[c] = M[r * LD + c];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float d = bcast_lane(m[k], k);
        ok = ok && (d > 0.f);
        const float inv_d = 1.f / d;
        const bool pivot_row = (lane == k);
        const float f = m[k] * inv_d; // a_rk / d
#pragma unroll
        for (int c = 0; c < K; ++c) {
            if (c == k) continue;
            const float pkc = bcast_lane(m[c], k); // a_kc (old)
            m[c] = pivot_row ? pkc * inv_d : fmaf(-f, pkc, m[c]);
        }
        m[k] = pivot_row ? -inv_d : f;
    }
    float fro = 0.f;
#pragma unroll
    for (int c = 0; c < K; ++c) {
        m[c] = -m[c];
        fro = fmaf(m[c], m[c], fro);
    }
    if (lane >= K) fro = 0.f;
    fro = wsum(fro);
    __syncthreads();
    if (lane < K) {
#pragma unroll
        for (int c = 0; c < K; ++c) M[r * LD + c] = m[c];
    }
    __syncthreads();
    return ok && isfinite(fro) && sqrtf(fro) * min_eig <= 1.f;
}

// compact in-place two-sided Jacobi on LD-layout matrices (round-robin pairs, everything through LDS): only used by the rare
// spectral fallback of inverse27 (LD layout in, LD layout out, no conversion)
__device__ void jacobi27_inplace(float *A, float *V, float *prm /* 4 * KP/2 floats */, int lane)
{
    float *rc = prm, *rs = prm + KP / 2;
    int *rp = reinterpret_cast<int *>(prm + KP), *rq = rp + KP / 2;
    for (int e = lane; e < K * K; e += 64) { int r = e / K, c = e - r * K; V[r * LD + c] = (r == c) ? 1.f : 0.f; }
    __syncthreads();
    constexpr int N1 = KP - 1, NP = KP / 2;
    for (int sweep = 0; sweep < 14; ++sweep) {
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
                int a = (lane == 0) ? N1 : (round + lane) % N1, bb = (lane == 0) ? round : (round - lane + N1) % N1;
                int p = min(a, bb), q = max(a, bb);
                float c = 1.f, s = 0.f;
                if (q < K) {
                    float apq = A[p * LD + q];
                    if (apq != 0.f) {
                        float theta = (A[q * LD + q] - A[p * LD + p]) / (2.f * apq);
                        float t = 1.f / (fabsf(theta) + sqrtf(fmaf(theta, theta, 1.f)));
                        t = theta < 0.f ? -t : t;
                        c = 1.f / sqrtf(fmaf(t, t, 1.f));
                        s = t * c;
                        if (!(fabsf(theta) < 1e18f)) { c = 1.f; s = 0.f; }
                    }
                } else { p = 0; q = 0; }
                rc[lane] = c; rs[lane] = s; rp[lane] = p; rq[lane] = q;
            }
            __syncthreads();
            for (int t = lane; t < NP * K; t += 64) { // A <- A J, V <- V J
                int k = t / K, row = t - k * K;
                float c = rc[k], s = rs[k];
                if (s != 0.f) {
                    int p = rp[k], q = rq[k];
                    float ap = A[row * LD + p], aq = A[row * LD + q], vp = V[row * LD + p], vq = V[row * LD + q];
                    A[row * LD + p] = fmaf(c, ap, -s * aq); A[row * LD + q] = fmaf(s, ap, c * aq);
                    V[row * LD + p] = fmaf(c, vp, -s * vq); V[row * LD + q] = fmaf(s, vp, c * vq);
                }
            }
            __syncthreads();
            for (int t = lane; t < NP * K; t += 64) { // A <- J^T A
                int k = t / K, col = t - k * K;
                float c = rc[k], s = rs[k];
                if (s != 0.f) {
                    int p = rp[k], q = rq[k];
                    float ap = A[p * LD + col], aq = A[q * LD + col];
                    A[p * LD + col] = fmaf(c, ap, -s * aq); A[q * LD + col] = fmaf(s, ap, c * aq);
                }
            }
            __syncthreads();
        }
    }
}

// M <- inverseSymmetricMatrix(M) (DenoisingUnit.cpp:578-604), in place.  scratch: S0, S1 (matrix sized), fl, prm (2*KP floats)
__device__ void inverse27(float *M, float *S0, float *S1, float *fl, float *prm, int lane, float min_eig)
{
    // backup for the spectral path, taken before the sweep destroys M: lower triangle mirrored, like Eigen's solver reads it
    for (int e = lane; e < K * K; e += 64) { int r = e / K, c = e - r * K; S0[r * LD + c] = M[(r >= c ? r : c) * LD + (r >= c ? c : r)]; }
    __syncthreads();
    if (sweep_inverse27(M, lane, min_eig)) return;
    jacobi27_inplace(S0, S1, prm, lane);
    if (lane < K) fl[lane] = 1.f / fmaxf(min_eig, S0[lane * LD + lane]);
    __syncthreads();
    for (int e = lane; e < K * K; e += 64) {
        int r = e / K, c = e - r * K;
        float s = 0.f;
        for (int k = 0; k < K; ++k) s = fmaf(S1[r * LD + k], fl[k] * S1[c * LD + k], s);
        M[r * LD + c] = s;
    }
    __syncthreads();
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

// 27 x 27 products on the f32 matrix core (v_mfma_f32_32x32x2_f32: exact f32, a chain of fma over k):
//   out[i][j] = sum_k X[i][k] (* scale[k]) * Y[k][j]      (TRANS_Y: Y[j][k]),   k < kdim <= 28,
// lane l feeds A[i = l & 31][k = 2s + (l >> 5)] and B[k][j = l & 31]; rows / columns 27..31 are zero padding.
// All operands are in registers before the first store (single wavefront), so `out` may alias X or Y.
template <bool TRANS_Y, bool SCALE>
__device__ __attribute__((noinline)) void mfma27(float *out, int ldo, const float *X, int ldx, const float *Y, int ldy, const float *scale, int kdim, int lane)
{
    const int idx = lane & 31, kh = lane >> 5;
    v16f acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll 7
    for (int k0 = 0; k0 < kdim; k0 += 2) {
        const int k = k0 + kh;
        const bool ok = idx < K && k < kdim;
        float a = ok ? X[idx * ldx + k] : 0.f;
        if (SCALE) a *= ok ? scale[k] : 0.f;
        const float b = ok ? (TRANS_Y ? Y[idx * ldy + k] : Y[k * ldy + idx]) : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * kh; // C/D layout: column = lane & 31
        if (r < K && idx < K) out[r * ldo + idx] = acc[e];
    }
    __syncthreads();
}

__device__ void rebuild27(float *out, const float *A, const float *V, float *fl, int lane, bool inverse, float min_eig)
{
    if (lane < KP) {
        float lam = A[lane * JLD + lane];
        fl[lane] = inverse ? 1.f / fmaxf(min_eig, lam) : fmaxf(0.f, lam);
    }
    __syncthreads();
    mfma27<true, true>(out, LD, V, JLD, V, JLD, fl, KP, lane);
}

// members are kept as 16-bit window codes ((dl + b) << 8 | (dc + b)): half the LDS of pixel indices, decoded with member_pixel()
__device__ inline int member_pixel(uint16_t code, int p, int W, int b) { return p + ((int)(code >> 8) - b) * W + ((int)(code & 255) - b); }

__device__ int decode_members27(const uint32_t *mask, int p, const Geom27 &g, uint16_t *mem, int lane)
{
    const float inv_side = 1.f / (float)g.side;
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
        // k / side without an integer division: (k + 0.5) / side is at least 0.5 / side away from an integer, k < 2^10
        int kl = (int)(((float)k + 0.5f) * inv_side), kc = k - kl * g.side;
        mem[pos++] = (uint16_t)((kl << 8) | kc);
    }
    __syncthreads();
    return total;
}

// stage members [i0, i0+cn) of the similar set into the LDS chunk (pickColorPatchesFromColorImage :483-498)
// staging of members [i0, i0+cn) of the similar set (pickColorPatchesFromColorImage :483-498) in two halves, so that the global
// loads of the next chunk are in flight while the current one is being consumed: stage_load -> registers, stage_store -> LDS
constexpr int STAGE_REGS = (CHUNK * K + 63) / 64;
__device__ inline void stage_load(float (&pre)[STAGE_REGS], const float *__restrict__ colors, const uint16_t *mem, int p, int b, int i0,
                                  int cn, int W, int lane)
{
#pragma unroll
    for (int j = 0; j < STAGE_REGS; ++j) {
        const int t = min(lane + 64 * j, cn * K - 1); // idle slots repeat the last element
        const int i = t / K, k = t - i * K, o = k / 3, ch = k - o * 3;
        const int offp = (o / 3 - 1) * W + (o % 3 - 1);
        pre[j] = colors[(size_t)(member_pixel(mem[i0 + i], p, W, b) + offp) * 3 + ch];
    }
}
__device__ inline void stage_store(float *chunk, const float (&pre)[STAGE_REGS], int cn, int lane)
{
#pragma unroll
    for (int j = 0; j < STAGE_REGS; ++j) {
        const int t = lane + 64 * j;
        if (t < cn * K) chunk[t] = pre[j];
    }
    __syncthreads();
}

// empirical covariance of the member patches on the matrix core (see the call site)
__device__ __attribute__((noinline)) void covariance27(float *A, float *Cm, float *chunk, const float *mean, const float *__restrict__ colors,
                                                       const uint16_t *mem, int p, int b, int n, int W, int lane)
{
    {
        const int mi = lane & 31, mk = lane >> 5;
        const float my_mean = mi < K ? mean[mi] : 0.f;
        v16f acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        float pre[STAGE_REGS];
        stage_load(pre, colors, mem, p, b, 0, min(CHUNK, n), W, lane);
        for (int i0 = 0; i0 < n; i0 += CHUNK) {
            int cn = min(CHUNK, n - i0);
            stage_store(chunk, pre, cn, lane);
            if (i0 + CHUNK < n) stage_load(pre, colors, mem, p, b, i0 + CHUNK, min(CHUNK, n - i0 - CHUNK), W, lane);
            for (int s2 = 0; s2 < cn; s2 += 2) {
                const int m = s2 + mk;
                const float a = (m < cn && mi < K) ? chunk[m * K + mi] - my_mean : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc, 0, 0, 0);
            }
            __syncthreads();
        }
        const float inv = 1.f / (float)(n - 1);
        // C/D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * mk;
            if (r < K && mi < K) {
                const float v = acc[e] * inv;
                A[r * LD + mi] = v;
                Cm[r * LD + mi] = v;
            }
        }
        __syncthreads();
    }

}

// y = G2 (x - m) for one staged chunk on the matrix core, and its aggregation (see the call site)
__device__ __attribute__((noinline)) void final_chunk27(const float *Cm, const float *chunk, const float *mean, const uint16_t *mem, int i0, int cn, int p,
                                                        int W, int b, bool in_lds, int AW, int b1, float *accS, int *accC, float *sum,
                                                        int32_t *cnt, int lane)
{
        // D[r][j] = sum_k G2[r][k] * xc_j[k] (fma chain over k), lane l
        // feeds A = G2[l & 31][k] and B = xc of member l & 31, k = 2s + (l >> 5); it gets back components
        // r = (e & 3) + 8 (e >> 2) + 4 (l >> 5) of member l & 31
        {
            const int mj = lane & 31, kh = lane >> 5;
            v16f y;
#pragma unroll
            for (int e = 0; e < 16; ++e) y[e] = 0.f;
#pragma unroll 7
            for (int k0 = 0; k0 < K; k0 += 2) {
                const int k = k0 + kh;
                const bool ok = k < K;
                const float a = (ok && mj < K) ? Cm[mj * LD + k] : 0.f;
                const float bq = (ok && mj < cn) ? chunk[mj * K + k] - mean[k] : 0.f;
                y = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, y, 0, 0, 0);
            }
            if (mj < cn) {
                const float *x = chunk + mj * K;
                const uint16_t code = mem[i0 + mj];
                const int dy = (int)(code >> 8) - b, dx = (int)(code & 255) - b, q0 = p + dy * W + dx;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = (e & 3) + 8 * (e >> 2) + 4 * kh;
                    if (r < K) {
                        const int o = r / 3, ch = r - 3 * o, oy = o / 3 - 1, ox = o - 3 * (o / 3) - 1;
                        const float v = x[r] - y[e];
                        if (in_lds) {
                            const int wq = (dy + b1 + oy) * AW + dx + b1 + ox;
                            unsafeAtomicAdd(accS + wq * 3 + ch, v);
                            if (ch == 0) atomicAdd(accC + wq, 1);
                        } else {
                            const int q = q0 + oy * W + ox;
                            unsafeAtomicAdd(sum + (size_t)q * 3 + ch, v);
                            if (ch == 0) atomicAdd(cnt + q, 1);
                        }
                    }
                }
            }
        }
}

// the output pass over all chunks, next chunk's loads in flight during the product and the aggregation of the current one
__device__ __attribute__((noinline)) void final_pass27(const float *Cm, float *chunk, const float *mean, const float *__restrict__ colors,
                                                       const uint16_t *mem, int n, int p, int W, int b, bool in_lds, int AW, int b1,
                                                       float *accS, int *accC, float *sum, int32_t *cnt, int lane)
{
    float pre[STAGE_REGS];
    stage_load(pre, colors, mem, p, b, 0, min(CHUNK, n), W, lane);
    for (int i0 = 0; i0 < n; i0 += CHUNK) {
        int cn = min(CHUNK, n - i0);
        stage_store(chunk, pre, cn, lane);
        if (i0 + CHUNK < n) stage_load(pre, colors, mem, p, b, i0 + CHUNK, min(CHUNK, n - i0 - CHUNK), W, lane);
        final_chunk27(Cm, chunk, mean, mem, i0, cn, p, W, b, in_lds, AW, b1, accS, accC, sum, cnt, lane);
        __syncthreads();
    }
}

template <bool DBG>
__global__ __launch_bounds__(64) void k_bayes27(const float *__restrict__ colors, const float *__restrict__ pixcov,
                                                const uint32_t *__restrict__ mask, const int32_t *__restrict__ list,
                                                const int32_t *__restrict__ d_nlist, int *work, Geom27 g, float min_eig, float *sum,
                                                int32_t *cnt)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    float *Cm = lds, *A = Cm + MSZ, *V = A + MSZ, *Bm = V + MSZ; // A, V contiguous: reused as the aggregation window
    float *chunk = Bm;                         // member-staging chunk: Bm is free while the clouds are streamed (mean/covariance, output)
    float *cs = Bm + MSZ;                      // 2 x 28 floats (rotation parameters), read as float4: offsets are multiples of 16 bytes
    float *noise = cs + 2 * KP;
    float *mean = noise + P * 6;
    float *fl = mean + K + 1;
    uint16_t *mem = reinterpret_cast<uint16_t *>(fl + KP);

    // persistent wavefronts: the list length lives in device memory (no host round trip between the marking and this
    // launch), items are handed out through an atomic counter
    const int nlist = *d_nlist;
  for (;;) {
    int item = 0;
    if (lane == 0) item = atomicAdd(work, 1);
    item = __builtin_amdgcn_readfirstlane(item);
    if (item >= nlist) break;
    DBG_T(0);
    const int p = list[item];
    const int n = decode_members27(mask, p, g, mem, lane);
    DBG_T(1);
    const float n_inv = 1.f / (float)n;
    const int W = g.W;

    // computeNoiseCovPatchesMean (:400-419) and empiricalMean (:500-509) in one sweep over the members, straight from global
    // memory: lanes 0..53 own one noise component, lanes 0..26 also one colour component.  The loads of 8 members are issued
    // back to back (independent addresses), the sums stay in member order.
    {
        const bool do_noise = lane < P * 6, do_mean = lane < K;
        // (idle lanes repeat the loads of the last owner: no branches around the loads)
        int noff, coff;
        { const int l = min(lane, P * 6 - 1), o = l / 6, j = l - o * 6; noff = ((o / 3 - 1) * W + (o % 3 - 1)) * 6 + j; }
        { const int l = min(lane, K - 1), o = l / 3, ch = l - o * 3; coff = ((o / 3 - 1) * W + (o % 3 - 1)) * 3 + ch; }
        float accn = 0.f, accm = 0.f;
        for (int i = 0; i < n; i += 8) {
            float vn[8], vm[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = member_pixel(mem[min(i + u, n - 1)], p, W, g.b);
                vn[u] = pixcov[(long long)q * 6 + noff];
                vm[u] = colors[(long long)q * 3 + coff];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i + u < n) { accn += vn[u]; accm += vm[u]; }
        }
        if (do_noise) noise[lane] = accn * n_inv;
        if (do_mean) mean[lane] = accm * n_inv;
        __syncthreads();
    }
    DBG_T(2);
    DBG_T(3);
    // centerPointCloud + empiricalCovarianceMatrix (:511-536) on the matrix core: C = Xc^T Xc, two members per
    // v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: a chain of fma in member order, i.e. the reference's sequential sum with the
    // product fused; A and B operands are the same centred value, so the result is bitwise symmetric).
    // Operand layout: lane l holds element i = l & 31 of member 2s + (l >> 5); rows/columns 27..31 are zero.
    covariance27(A, Cm, chunk, mean, colors, mem, p, g.b, n, W, lane);

    // ---- Step 1 (:421-436): M1 = clamp(C - N) + N ; Cinv1 = inverse(M1)
    DBG_T(4);
    add_noise27(A, noise, lane, -1.f);
    to_jacobi_layout(Bm, A, lane);
    {
        jacobi27<DBG>(Bm, V, cs, lane, item);
        DBG_T(5);
        rebuild27(Bm, Bm, V, fl, lane, false, 0.f); // reads the eigenvalues (diagonal) before it overwrites Bm: M1 lives in Bm
    }
    add_noise27(Bm, noise, lane, +1.f);
    DBG_T(6);
    inverse27(Bm, A, V, fl, cs, lane, min_eig);
    DBG_T(7);
    // ---- Step 2 (:438-453): the Step-1 estimates are xhat = x - G (x - m) with G = N Cinv1, hence their empirical
    // mean is m and their empirical covariance is F C F^T, F = I - G
    noise_times27(V, noise, Bm, lane, true);       // V  = F
    mfma27<false, false>(A, LD, V, LD, Cm, LD, nullptr, K, lane);  // A  = F C
    mfma27<true, false>(Bm, LD, A, LD, V, LD, nullptr, K, lane);   // Bm = F C F^T
    for (int e = lane; e < K * K; e += 64) {       // exact symmetry (lower triangle wins)
        int r = e / K, c = e - r * K;
        if (r < c) Cm[r * LD + c] = Bm[c * LD + r];
    }
    __syncthreads();
    for (int e = lane; e < K * K; e += 64) { int r = e / K, c = e - r * K; if (r < c) Bm[r * LD + c] = Cm[r * LD + c]; }
    __syncthreads();
    add_noise27(Bm, noise, lane, +1.f);
    DBG_T(8);
    inverse27(Bm, A, V, fl, cs, lane, min_eig);
    noise_times27(Cm, noise, Bm, lane, false);
    DBG_T(9);     // Cm = G2 = N Cinv2

    // ---- finalDenoisingMatrixMultiplication (:656-670) on the noisy patches centred on m, aggregateOutputPatches (:672-693).
    // Every patch of every member lies in the (side+2)^2 window around p: the contributions are first summed there in LDS
    // (A and V are dead by now) and flushed with one global atomic per touched value, rows contiguous; windows that do not
    // fit (b > 8) go straight to global atomics.
    const int AW = g.side + 2, b1 = g.b + 1;
    const bool in_lds = AW * AW * 4 <= 2 * MSZ; // A and V; Bm holds the chunk
    float *accS = A;
    int *accC = reinterpret_cast<int *>(A + AW * AW * 3);
    if (in_lds) {
        for (int e = lane; e < AW * AW * 4; e += 64) A[e] = 0.f;
        __syncthreads();
    }
    final_pass27(Cm, chunk, mean, colors, mem, n, p, W, g.b, in_lds, AW, b1, accS, accC, sum, cnt, lane);
    if (in_lds) {
        const int row3 = AW * 3;
        const long long base = (long long)p - (long long)b1 * W - b1; // window origin; untouched cells may lie outside the image
        for (int e = lane; e < AW * row3; e += 64) {
            int wy = e / row3, r = e - wy * row3, wx = r / 3;
            if (accC[wy * AW + wx] != 0) unsafeAtomicAdd(sum + (base + (long long)wy * W) * 3 + r, accS[e]);
        }
        for (int e = lane; e < AW * AW; e += 64) {
            int wy = e / AW, wx = e - wy * AW, c = accC[e];
            if (c != 0) atomicAdd(cnt + (base + (long long)wy * W + wx), c);
        }
    }
    DBG_T(10);
    if (DBG && item == bcd_dbg_item && lane == 0) bcd_dbg_cycles[11] = n;
    __syncthreads(); // the next item reuses the LDS
  }
}

} // namespace

size_t bcd_bayes27_lds_bytes(int b)
{
    int side = 2 * b + 1;
    return (size_t)(4 * MSZ + 2 * KP + P * 6 + (K + 1) + KP) * sizeof(float) + (((size_t)side * side * sizeof(uint16_t) + 15) & ~(size_t)15);
}

// workgroups (= wavefronts) of k_bayes27 one CU holds: LDS-bound, at most 3 per SIMD (148 VGPRs)
int bcd_bayes27_blocks_per_cu(int b)
{
    size_t n = (size_t)160 * 1024 / bcd_bayes27_lds_bytes(b);
    return (int)(n > 12 ? 12 : n);
}

hipError_t bcd_launch_bayes27(const float *colors, const float *pixcov, const uint32_t *mask, const int32_t *list, const int32_t *d_nlist,
                              int *d_work, int blocks, int W, int H, int b, float min_eig, float *sum, int32_t *cnt, hipStream_t st)
{
    if (blocks <= 0) return hipSuccess;
    Geom27 g;
    g.W = W; g.H = H; g.b = b; g.side = 2 * b + 1; g.words = (g.side * g.side + 31) / 32; g.maxS = g.side * g.side;
    if (g.words > 32) return hipErrorInvalidValue;
    if (getenv("BCD_DBG_BAYES")) {
        if (const char *e = getenv("BCD_DBG_ITEM")) { int v = atoi(e); (void)hipMemcpyToSymbol(HIP_SYMBOL(bcd_dbg_item), &v, sizeof(v)); }
        hipLaunchKernelGGL(k_bayes27<true>, dim3(blocks), dim3(64), bcd_bayes27_lds_bytes(b), st, colors, pixcov, mask, list, d_nlist, d_work, g, min_eig, sum, cnt);
        long long h[24];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(bcd_dbg_cycles), sizeof(h));
        fprintf(stderr, "sweeps off2/dg2 x1e18:"); for (int i = 12; i < 24; ++i) fprintf(stderr, " %lld", h[i]); fprintf(stderr, "\n");
        fprintf(stderr, "bayes27 dbg n=%lld: decode %lld noise %lld mean %lld cov %lld jacobi %lld rebuild %lld inv1 %lld step2mm %lld inv2 %lld final %lld total %lld\n", h[11], h[1]-h[0], h[2]-h[1], h[3]-h[2], h[4]-h[3], h[5]-h[4], h[6]-h[5], h[7]-h[6], h[8]-h[7], h[9]-h[8], h[10]-h[9], h[10]-h[0]);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_bayes27<false>, dim3(blocks), dim3(64), bcd_bayes27_lds_bytes(b), st, colors, pixcov, mask, list, d_nlist, d_work, g, min_eig, sum, cnt);
    return hipGetLastError();
}
