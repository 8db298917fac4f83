// k_bayes27.hip -- full Bayesian estimate of a processed pixel for the default patch radius w = 1 (K = 27): three kernels with a
// per-pixel record in HBM between them (persistent single-wavefront workgroups, items from work queues):
//   k_bayes27w<1> / k_bayes27<1>  PREPARE  members of S(p), noise mean, colour mean, covariance C on the f32 matrix core;
//                                          record: A = C - N (28 x 28, zero padded), C, N, m
//   k_jacobi27_quads              SOLVE    eigen-decomposition of every A (clampNegativeEigenValues), two matrices per wavefront, the
//                                          rows through LDS once per three rounds (k_jacobi27_batch: once per round, kept for comparison)
//   k_finish27w                   FINISH   clamp, inverses, Step 2, final estimates, aggregation -- the 27 x 27 algebra in registers on the
//                                          matrix core (default search radius); items whose inverse needs the spectral branch go to a redo list
//   k_bayes27w<2> / k_bayes27<2>  FINISH   the same through LDS matrices: the redo list / the other search radii
// The `w` kernels serve the default search radius b = 6: the 15 x 15 pixel window around p is staged in LDS once and every member
// access is an LDS read; other radii take the gather kernels (members fetched from global memory).
//
// Same mathematics as DenoisingUnit::denoiseSelectedPatches (src/core/DenoisingUnit.cpp:388-453) and
// aggregateOutputPatches (:672-693), reorganised for the GPU:
//   * sums over the members keep the reference's sequential member order;
//   * the covariance, the spectral rebuild, the Step-2 products and the final estimate run on the f32 matrix core
//     (v_mfma_f32_32x32x2_f32: exact f32, a chain of fma over k);
//   * Step 2's covariance of the Step-1 estimates (:441-443) is obtained without touching the members
//     again: the Step-1 estimate is affine, xhat = x - G (x - m), G = N Cinv1, so its empirical mean is m
//     and its empirical covariance is F C F^T with F = I - G  (exact identity, fp32 round-off apart);
//   * clampNegativeEigenValues (:606-630) is a parallel two-sided Jacobi eigendecomposition, rows in registers;
//   * inverseSymmetricMatrix (:578-604) = V diag(1/max(minEig, lambda)) V^T equals the plain inverse
//     whenever lambda_min >= minEig.  The inverse is computed with the symmetric sweep operator and
//     accepted only if every pivot is positive and ||M^-1||_F * minEig <= 1 (which proves
//     lambda_min(M) >= minEig); otherwise the spectral form is evaluated with a compact Jacobi solver;
//   * sums into LDS windows are plain read-add-write, never ds_add_f32 (193 cycles per wavefront instruction on gfx950).
#include "bcd_common.h"
#include <atomic>
#include <cstdio>
#include <algorithm>
#include <cstdlib>

namespace {


constexpr int K = 27, KP = 28, LD = 29, P = 9, MSZ = KP * KP, CHUNK = MSZ / K; // matrix buffer: 784 floats; 29 members per staging chunk

typedef float v16f __attribute__((ext_vector_type(16)));

struct Geom27 {
    int W, H, b, side, words, maxS;
};

// The out-of-line phases below receive their LDS buffers as plain pointers; without this hint the compiler addresses them with
// FLAT instructions (aperture check, both memory counters, a fraction of the ds_read / ds_write rate).
#define LDS_POINTER(p) __builtin_assume(__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void *)(p)))

// divided difference of f = max(0, .) at two eigenvalue estimates (see pos_part_lds)
__device__ inline float pos_phi(float di, float dj)
{
    const float hi = fmaxf(di, dj), lo = fminf(di, dj);
    if (lo > 0.f) return 1.f;
    if (hi <= 0.f) return 0.f;
    return hi * __builtin_amdgcn_rcpf(hi - lo); // (hi > 0 >= lo: the denominator is >= hi)
}

// what the eigensolvers leave in the record next to the eigenvalue estimates and V: row r of E o Phi -- the residual off-diagonal part of
// V^T A V times the divided differences of max(0, .) at the estimates, zero on the diagonal -- which the finish kernels turn into the
// first-order correction V (E o Phi) V^T of the positive part (pos_part_lds).  row: the lane's row in LDS, diag: all 28 estimates in LDS.
__device__ inline void jacobi_write_correction(float *out_row, const float *row, const float *diag, int r)
{
    const float d_r = diag[r];
#pragma unroll
    for (int q4 = 0; q4 < 7; ++q4) {
        const float4 e4 = reinterpret_cast<const float4 *>(row)[q4], d4 = reinterpret_cast<const float4 *>(diag)[q4];
        float4 o;
        o.x = 4 * q4 == r ? 0.f : e4.x * pos_phi(d_r, d4.x);
        o.y = 4 * q4 + 1 == r ? 0.f : e4.y * pos_phi(d_r, d4.y);
        o.z = 4 * q4 + 2 == r ? 0.f : e4.z * pos_phi(d_r, d4.z);
        o.w = 4 * q4 + 3 == r ? 0.f : e4.w * pos_phi(d_r, d4.w);
        reinterpret_cast<float4 *>(out_row)[q4] = o;
    }
}

__device__ inline float wsum(float v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Work distribution of the persistent kernels.  One counter serves 50-90 same-address atomics per microsecond: with a single counter per
// kernel the 36 000 grabs of a 1080p scale were 0.4 ms (finish kernel with its body skipped) to 0.7 ms (prepare kernel) on their own.
// Items are dealt round-robin to BCD_WORK_QUEUES counters in cache lines of their own (item = queue + 8 k); a wavefront starts on
// the queue of its XCD (workgroups go round-robin over the 8 XCDs) and moves on to the next queue when one is empty, so nothing is
// left behind and the tail stays balanced.
struct WorkCursor { int q, tried; };
__device__ inline WorkCursor work_begin() { return WorkCursor{ (int)(blockIdx.x % BCD_WORK_QUEUES), 0 }; }
__device__ inline int work_next(int *work, int nb_items, int lane, WorkCursor &c)
{
    while (c.tried < BCD_WORK_QUEUES) {
        int k = 0;
        if (lane == 0) k = atomicAdd(work + c.q * BCD_WORK_STRIDE, 1);
        k = __builtin_amdgcn_readfirstlane(k);
        const int item = c.q + BCD_WORK_QUEUES * k;
        if (item < nb_items) return item;
        c.q = (c.q + 1) % BCD_WORK_QUEUES;
        ++c.tried;
    }
    return -1;
}

// Round 6: the host launches the estimate kernels of a chunk BEFORE it knows how many items the list holds (the count is still on its way back:
// bayes() in bcd_api.hip); `d_n` then points at the list's length on the device and `nb_items` is the capacity the records were sized for.
__device__ inline int items_on_device(const int *d_n, int first_item, int nb_items)
{
    return d_n ? max(0, min(nb_items, *d_n - first_item)) : nb_items;
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

// x[k] += ms * (x[k] of lane ^ 1) for the 28 elements of a row, in place: 28 v_fmac_f32_dpp.  One asm block: the leading s_nop covers
// the two wait states the hardware wants between a VALU write of a VGPR and a DPP read of it (the compiler does not look inside an
// asm statement for that hazard); inside the block every instruction touches its own register only.
__device__ inline void rowrot_dpp(float (&x)[28], float ms)
{
#define BCD_F(i) "v_fmac_f32_dpp %" #i ", %" #i ", %28 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm volatile("s_nop 1\n\t"
                 BCD_F(0) BCD_F(1) BCD_F(2) BCD_F(3) BCD_F(4) BCD_F(5) BCD_F(6) BCD_F(7) BCD_F(8) BCD_F(9) BCD_F(10) BCD_F(11) BCD_F(12) BCD_F(13)
                 BCD_F(14) BCD_F(15) BCD_F(16) BCD_F(17) BCD_F(18) BCD_F(19) BCD_F(20) BCD_F(21) BCD_F(22) BCD_F(23) BCD_F(24) BCD_F(25) BCD_F(26) BCD_F(27)
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]),
                   "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17]), "+v"(x[18]), "+v"(x[19]),
                   "+v"(x[20]), "+v"(x[21]), "+v"(x[22]), "+v"(x[23]), "+v"(x[24]), "+v"(x[25]), "+v"(x[26]), "+v"(x[27])
                 : "v"(ms));
#undef BCD_F
}

__device__ inline int sigma_slot(int s)
{
    return s == 0 ? 0 : (s == 1 ? 2 : (s == 26 ? 27 : ((s & 1) ? s - 2 : s + 2)));
}

// ---------------------------------------------------------------------------------------------------------------------
// Batched eigensolver: the same Brent-Luk two-sided Jacobi, two matrices per wavefront and nothing else in the kernel.
// Matrix h of a pair lives in lanes 32 h .. 32 h + 27 (rows of A, moved through LDS every round) and, in a second register
// set of the same lanes, the rows of V (never move).  The LDS traffic and the row / column rotations of A are shared by the
// two matrices -- per matrix and round ~100 VALU and ~10 LDS instructions instead of ~140 and ~21 -- and with 6.5 KB of LDS
// per wavefront the residency is bound by registers only (3 wavefronts per SIMD = 24 matrices per CU instead of 12).
//   in:  A  [n][28 x 28] (JLD layout, row / column 27 zero)      out: eig [n][28] (slot order), V [n][28 x 28] (rows 0..26)
// ---------------------------------------------------------------------------------------------------------------------
// one wavefront per workgroup: no s_barrier is emitted for it, only the ordering of the LDS accesses
__device__ inline void wave_sync() { __syncthreads(); }

__device__ inline float half_sum(float v)
{
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// LDS placement of row r of a matrix of the batched solver (bank facts: MI355X_MICROARCH.md, LDS).  ds_read_b128 serves the lane groups
// {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} of a half, banks (a / 4) mod 64; ds_write_b128 serves groups of 8 consecutive lanes, banks
// (a / 4) mod 32.  With the rows at a stride of 28 dwords the LOADS of a round (lane r <- row r) are conflict free, but (a) the idle lanes
// 28..31 used to re-read row 0, which shares its banks with row 16 of their group, and (b) the STORES go to the Brent-Luk permuted slots:
// lanes 0..7 write the rows {0, 2, 4, 1, 6, 3, 8, 5}, and rows 0 and 8 start on the same bank mod 32 -- together SQ_LDS_BANK_CONFLICT =
// 17 % of the LDS pipe's cycles (r3 counters).  Row 8 therefore lives behind the matrix at a start bank of its own (found by exhaustive
// search over the load and store groups: every group conflict free), and the idle lanes re-read row 16.
constexpr int JROW8 = 28 * 28 + 20;        // start of row 8: (JROW8 / 4) mod 16 = 9
constexpr int JMAT = JROW8 + 28;           // floats per matrix in LDS (832)
template <bool PAD> __device__ inline int jrow_t(int r) { return (PAD && r == 8) ? JROW8 : r * 28; }

template <bool PAD, bool FUSE>
__global__ __launch_bounds__(64, 3) void k_jacobi27_batch(const float *Ain, int n, int *work,
                                                          float *__restrict__ eig, float *__restrict__ Vout,
                                                          float conv2 /* stop at off^2 <= conv2 diag^2 */, float *Aout /* optional (may be Ain): the matrix as the sweeps left it, V^T A V */,
                                                          const int *d_n, int first_item)
{
    n = items_on_device(d_n, first_item, n);
    auto jrow = [](int r) { return jrow_t<PAD>(r); };
    __shared__ float4 lds4[(2 * JMAT + 4 * KP) / 4];
    float *Abuf = reinterpret_cast<float *>(lds4);
    float *cs = Abuf + 2 * JMAT;      // [matrix][28]: per slot pair (-beta, alpha) of the current round
    float *dv = cs + 2 * KP;          // [matrix][28]: scale of every slot (see below)
    const int lane = threadIdx.x, h = lane >> 5, r = lane & 31;
    float *Ah = Abuf + h * JMAT;
    float *cs_h = cs + h * KP, *dv_h = dv + h * KP;
    const bool isRow = r < KP;
    const float *src = Ah + jrow(isRow ? r : (PAD ? 16 : 0)); // (idle lanes: the address of a lane of their own ds_read_b128 group)
    float *dst = Ah + jrow(isRow ? sigma_slot(r) : 0);
    WorkCursor cursor = work_begin();
    for (;;) {
        const int pair = work_next(work, (n + 1) / 2, lane, cursor);
        if (pair < 0) break;
        const int first = 2 * pair;
        const int item = first + h;
        const bool live = item < n;
        // matrix -> LDS (an absent second matrix is the identity: converged from the start)
        {
            const float4 *g = reinterpret_cast<const float4 *>(Ain + (size_t)(live ? item : first) * (KP * JLD));
            for (int e = r; e < KP * JLD / 4; e += 32) {
                float4 v = g[e];
                const int e0 = 4 * e, rr = e0 / JLD, c0 = e0 - rr * JLD;
                if (!live) v = make_float4(rr == c0, rr == c0 + 1, rr == c0 + 2, rr == c0 + 3);
                *reinterpret_cast<float4 *>(Ah + jrow(rr) + c0) = v;
            }
            if (isRow) dv_h[r] = 1.f;
        }
        wave_sync();
        // Scaled ("fast") rotations: the matrices are kept as A = D A~ D and V = V~ D with a diagonal D (one scale per slot).
        // J = [[c, s], [-s, c]] = [[1, t], [-t, 1]] diag(c, c), so rotating the slot pair (p, q) is
        //     x' = x - beta y,  y' = y + alpha x   (beta = t d_q / d_p, alpha = t d_p / d_q)   and   d' = c d
        // -- two fused multiply-adds per element pair instead of four operations; D is folded back into A~ and V~ at the start of
        // every sweep (within a sweep a scale cannot fall below 2^(-27/2)).
        float vrow[JLD];
#pragma unroll
        for (int k = 0; k < JLD; ++k) vrow[k] = (k == r) ? 1.f : 0.f; // V = identity
        for (int sweep = 0; sweep < 12; ++sweep) {
            // fold the scales: A~ <- D A~ D, V~ <- V~ D, D <- I; off / diagonal norms of the true matrix on the way
            float off = 0.f, dg = 0.f;
            {
                float dk[JLD];
#pragma unroll
                for (int q4 = 0; q4 < JLD / 4; ++q4) {
                    float4 w4 = reinterpret_cast<const float4 *>(dv_h)[q4];
                    dk[4 * q4] = w4.x; dk[4 * q4 + 1] = w4.y; dk[4 * q4 + 2] = w4.z; dk[4 * q4 + 3] = w4.w;
                }
                const float dr = dv_h[isRow ? r : 0];
                float row[JLD];
#pragma unroll
                for (int q4 = 0; q4 < JLD / 4; ++q4) {
                    float4 v = reinterpret_cast<const float4 *>(src)[q4];
                    row[4 * q4] = v.x; row[4 * q4 + 1] = v.y; row[4 * q4 + 2] = v.z; row[4 * q4 + 3] = v.w;
                }
#pragma unroll
                for (int k = 0; k < JLD; ++k) {
                    row[k] *= dr * dk[k];
                    vrow[k] *= dk[k];
                    if (isRow) { if (k == r) dg = fmaf(row[k], row[k], dg); else off = fmaf(row[k], row[k], off); }
                }
                wave_sync(); // every lane has read the scales and its row
                if (isRow) {
#pragma unroll
                    for (int q4 = 0; q4 < JLD / 4; ++q4)
                        reinterpret_cast<float4 *>(Ah + jrow(r))[q4] = make_float4(row[4 * q4], row[4 * q4 + 1], row[4 * q4 + 2], row[4 * q4 + 3]);
                    dv_h[r] = 1.f;
                }
                wave_sync();
            }
            off = half_sum(off);
            dg = half_sum(dg);
            // off^2 <= 1e-12 diag^2: the positive part V max(0, lambda) V^T is Lipschitz in the matrix, so what is left off the diagonal moves it
            // by <= 1e-6 |A|.  (1e-10 saved 3 % of the solver and was fine on the 32-spp frames, but the 8-spp 4K frame of configs[3] then
            // differed from the oracle by 1.9e-5 -- its inverses are worse conditioned -- against 2.9e-6 with this threshold.)  A converged
            // matrix is frozen -- identity rotations -- while its partner finishes: its result does not depend on who shares the wavefront
            const bool settled = off <= conv2 * dg;
            if (__builtin_amdgcn_ballot_w64(!settled) == 0) break; // both matrices converged
            // fully unrolled: the Brent-Luk column move of the rows of V~ (a 27-cycle of the register names) costs no instruction
#pragma unroll
            for (int round = 0; round < KP - 1; ++round) {
                // lanes 0..13 of each half: rotation of the slot pair (2 r, 2 r + 1); the other lanes compute on a harmless copy of pair 0
                float mbeta = 0.f, alpha = 0.f;
                {
                    const int p = r < KP / 2 ? 2 * r : 0, q = p + 1;
                    const float2 d2 = reinterpret_cast<const float2 *>(dv_h)[p >> 1];
                    const float apq_s = Ah[jrow(p) + q], app_s = Ah[jrow(p) + p], aqq_s = Ah[jrow(q) + q];
                    float dp = d2.x, dq = d2.y;
                    if (apq_s != 0.f && !settled) {
                        const float apq = dp * dq * apq_s, app = dp * dp * app_s, aqq = dq * dq * aqq_s;
                        // 1-ulp hardware reciprocal / sqrt / rsqrt: a rotation only has to be orthogonal to working
                        // precision (c^2 + s^2 = 1 +- 1e-7), not the exact minimiser
                        const float theta = (aqq - app) * __builtin_amdgcn_rcpf(2.f * apq);
                        if (fabsf(theta) < 1e18f) { // (otherwise theta^2 overflows: the rotation is the identity to fp32)
                            float t = __builtin_amdgcn_rcpf(fabsf(theta) + __builtin_amdgcn_sqrtf(fmaf(theta, theta, 1.f)));
                            t = theta < 0.f ? -t : t;
                            const float c = __builtin_amdgcn_rsqf(fmaf(t, t, 1.f));
                            const float ratio = dq * __builtin_amdgcn_rcpf(dp);
                            mbeta = -t * ratio;
                            alpha = t * __builtin_amdgcn_rcpf(ratio);
                            dp *= c;
                            dq *= c;
                        }
                    }
                    float row[JLD];
#pragma unroll
                    for (int q4 = 0; q4 < JLD / 4; ++q4) {
                        float4 v = reinterpret_cast<const float4 *>(src)[q4];
                        row[4 * q4] = v.x; row[4 * q4 + 1] = v.y; row[4 * q4 + 2] = v.z; row[4 * q4 + 3] = v.w;
                    }
                    if (r < KP / 2) {
                        reinterpret_cast<float2 *>(cs_h)[r] = make_float2(mbeta, alpha);
                        dv_h[sigma_slot(p)] = dp; // the scales move with their slots
                        dv_h[sigma_slot(q)] = dq;
                    }
                    wave_sync();
                    float rot[JLD];
#pragma unroll
                    for (int q4 = 0; q4 < JLD / 4; ++q4) {
                        float4 w4 = reinterpret_cast<const float4 *>(cs_h)[q4];
                        rot[4 * q4] = w4.x; rot[4 * q4 + 1] = w4.y; rot[4 * q4 + 2] = w4.z; rot[4 * q4 + 3] = w4.w;
                    }
                    const float2 pcs = reinterpret_cast<const float2 *>(cs_h)[isRow ? (r >> 1) : 0];
                    // column rotations of A~
#pragma unroll
                    for (int j = 0; j < KP / 2; ++j) {
                        const float x = row[2 * j], y = row[2 * j + 1];
                        row[2 * j] = fmaf(rot[2 * j], y, x);
                        row[2 * j + 1] = fmaf(rot[2 * j + 1], x, y);
                    }
                    // row rotations of A~: rows (2i, 2i+1) live in lanes (2i, 2i+1) of their half and take the rotation of pair i
                    {
                        const float ms = (r & 1) ? pcs.y : pcs.x;
                        // row[k] += ms * row[k] of the partner lane: ONE v_fmac_f32 with the lane exchange as its DPP modifier (the
                        // compiler emits v_mov_b32_dpp + v_fma_f32; VOP3 has no DPP form on gfx9)
                        float out[JLD];
                        if (FUSE) {
                            rowrot_dpp(row, ms);
#pragma unroll
                            for (int k = 0; k < JLD; ++k) out[sigma_slot(k)] = row[k]; // Brent-Luk column move (register renaming)
                        } else {
#pragma unroll
                            for (int k = 0; k < JLD; ++k) out[sigma_slot(k)] = fmaf(ms, dpp_xor1(row[k]), row[k]);
                        }
                        if (isRow) {
#pragma unroll
                            for (int q4 = 0; q4 < JLD / 4; ++q4)
                                reinterpret_cast<float4 *>(dst)[q4] = make_float4(out[4 * q4], out[4 * q4 + 1], out[4 * q4 + 2], out[4 * q4 + 3]);
                        }
                    }
                    // the same column rotations (and column move) on the rows of V~
                    {
                        float out[JLD];
#pragma unroll
                        for (int j = 0; j < KP / 2; ++j) {
                            const float x = vrow[2 * j], y = vrow[2 * j + 1];
                            out[sigma_slot(2 * j)] = fmaf(rot[2 * j], y, x);
                            out[sigma_slot(2 * j + 1)] = fmaf(rot[2 * j + 1], x, y);
                        }
#pragma unroll
                        for (int k = 0; k < JLD; ++k) vrow[k] = out[k];
                    }
                }
                wave_sync(); // the rows of A~ are back in LDS before the next round reads its pivots
            }
        }
        if (live) {
            if (isRow) eig[(size_t)item * KP + r] = Ah[jrow(r) + r];
        }
        if (Aout) { // (wave-uniform)
            if (isRow) cs_h[r] = Ah[jrow(r) + r];
            wave_sync();
            if (isRow && live) jacobi_write_correction(Aout + (size_t)item * (KP * JLD) + r * JLD, Ah + jrow(r), cs_h, r);
        }
        if (live) {
            if (r < K) {
                float4 *o = reinterpret_cast<float4 *>(Vout + (size_t)item * (KP * JLD) + r * JLD);
#pragma unroll
                for (int q4 = 0; q4 < JLD / 4; ++q4) o[q4] = make_float4(vrow[4 * q4], vrow[4 * q4 + 1], vrow[4 * q4 + 2], vrow[4 * q4 + 3]);
            }
        }
        wave_sync(); // the next pair reuses the LDS
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_jacobi27_quads: the same two-sided Jacobi (two matrices per wavefront, rows in registers, scaled rotations, V~ accumulated in
// registers), with the rows moved through LDS once per THREE rounds instead of once per round.
// k_jacobi27_batch is bound by the LDS pipe, not by arithmetic: per wavefront and round ~97 cycles for the 7 ds_write_b128 of the rows
// (a 16-byte store costs 13.9 cycles per wavefront instruction, three times a load: tools/ubench/lds_rate.hip) + 64 for the row and
// rotation-table loads + ~25 for pivots and scales = ~186 cycles of a pipe that 12 wavefronts share, against ~130 CU-cycles of vector
// work (r3: removing six of the seven row stores made the kernel 22 % faster).
// Here a sweep is 9 "super-rounds".  In each the 28 slots form 7 quads of neighbouring lanes and ALL SIX pairs of a quad are rotated --
// three rounds, partners at lane distance 1, 2, 3 (xor), every exchange a DPP quad_perm, every column pair a compile-time register
// pair -- before the rows go back to LDS at their next positions.  The nine partitions into quads are the parallel classes of a
// resolvable 2-(28,4,1) design (tools/jacobi_schedule.py: every pair of slots shares a quad exactly once per sweep; two lane
// permutations, JSX and JSY, applied x x y x x y x x y, bring every slot home after the ninth).  The rotations of the second and third
// round depend on the first's results, but only on the 4 x 4 diagonal block of the quad: each lane carries its row of that block along
// (4 values, in xor-relative order: w[y] = A~[me][me ^ y], so that every index is a compile-time constant in every lane) and the three
// rounds' angles come out of it before the 28-wide rows are touched.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int JSX[28] = { 1, 6, 12, 8, 10, 24, 0, 17, 19, 11, 20, 14, 23, 5, 9, 26, 15, 25, 22, 3, 4, 18, 21, 2, 13, 7, 16, 27 };
constexpr int JSY[28] = { 7, 17, 1, 9, 18, 8, 25, 23, 11, 13, 22, 5, 6, 3, 24, 20, 10, 2, 26, 14, 21, 15, 16, 0, 19, 12, 4, 27 };
// start of row r of a matrix in LDS, floats: no bank conflict in the loads (lane l reads row l) nor in the stores (lane l writes row
// JSX[l] or JSY[l]) of ds_*_b128 -- found and checked by tools/jacobi_schedule.py
constexpr int JPLACE[28] = { 196, 252, 308, 28, 364, 564, 392, 84, 112, 168, 280, 708, 596, 140, 748, 336, 476, 652, 0, 508, 624, 776, 536, 56, 224, 420, 448, 680 };
constexpr int JIDLE_ROW = 15;                       // what the idle lanes 28..31 of a half read
constexpr int JQ_MAT = 808;                         // floats per matrix (804 used)
constexpr int JQ_HALF = JQ_MAT + 32 + 3 * 32;       // + scales + the three rounds' rotation tables

// sum over the 32 lanes of a half, without address registers (ds_swizzle, xor mode)
__device__ inline float half_sum_swz(float v)
{
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (16 << 10) | 0x1f));
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (8 << 10) | 0x1f));
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (4 << 10) | 0x1f));
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (2 << 10) | 0x1f));
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (1 << 10) | 0x1f));
    return v;
}

template <int X> __device__ inline float dpp_xor(float v)   // v of lane ^ X inside the quad
{
    constexpr int ctrl = X == 1 ? 0xB1 /* [1,0,3,2] */ : (X == 2 ? 0x4E /* [2,3,0,1] */ : 0x1B /* [3,2,1,0] */);
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true));
}

// x[k] += ms * (x[k] of lane ^ X): rowrot_dpp for the three partners of a quad
#define BCD_ROWROT(NAME, PERM)                                                                                                              \
    __device__ inline void NAME(float (&x)[28], float ms)                                                                                   \
    {                                                                                                                                       \
        asm volatile("s_nop 1\n\t" BCD_G(0, PERM) BCD_G(1, PERM) BCD_G(2, PERM) BCD_G(3, PERM) BCD_G(4, PERM) BCD_G(5, PERM) BCD_G(6, PERM)  \
                     BCD_G(7, PERM) BCD_G(8, PERM) BCD_G(9, PERM) BCD_G(10, PERM) BCD_G(11, PERM) BCD_G(12, PERM) BCD_G(13, PERM)            \
                     BCD_G(14, PERM) BCD_G(15, PERM) BCD_G(16, PERM) BCD_G(17, PERM) BCD_G(18, PERM) BCD_G(19, PERM) BCD_G(20, PERM)         \
                     BCD_G(21, PERM) BCD_G(22, PERM) BCD_G(23, PERM) BCD_G(24, PERM) BCD_G(25, PERM) BCD_G(26, PERM) BCD_G(27, PERM)         \
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),          \
                       "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]), "+v"(x[16]), "+v"(x[17]),  \
                       "+v"(x[18]), "+v"(x[19]), "+v"(x[20]), "+v"(x[21]), "+v"(x[22]), "+v"(x[23]), "+v"(x[24]), "+v"(x[25]), "+v"(x[26]), \
                       "+v"(x[27])                                                                                                          \
                     : "v"(ms));                                                                                                            \
    }
#define BCD_G(i, PERM) "v_fmac_f32_dpp %" #i ", %" #i ", %28 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
BCD_ROWROT(rowrot_x1, "[1,0,3,2]")
BCD_ROWROT(rowrot_x2, "[2,3,0,1]")
BCD_ROWROT(rowrot_x3, "[3,2,1,0]")
#undef BCD_G
#undef BCD_ROWROT
template <int X> __device__ inline void rowrot_quad(float (&x)[28], float ms)
{
    if (X == 1) rowrot_x1(x, ms); else if (X == 2) rowrot_x2(x, ms); else rowrot_x3(x, ms);
}

// one round of a super-round in the quad's 4 x 4 block: my coefficient of the rotation of (me, me ^ X), the block row and scale updated.
// Written from the lane's own point of view, the two lanes of a pair run the SAME instructions on mirrored operands: with
// delta = a_partner - a_me (true values), theta = delta / (2 a_pq), t = sign(theta) / (|theta| + sqrt(theta^2 + 1)), the lower lane's theta is
// the pair's and the upper lane's its exact negative, so that  mine = -t (d_partner / d_me)  is -beta in the lower lane (x' = x - beta y) and
// +alpha in the upper one (y' = y + alpha x), and d' = c d in both: no lower / upper selects.  (Both lanes must see the same off-diagonal
// element: the lower lane's copy -- the two copies of a symmetric element agree only to rounding, and two slightly different angles cost
// 1.7e-4 of orthogonality.)
template <int X>
__device__ inline float quad_round(float (&w)[4], float &dme, bool lo, bool frozen)
{
    const float pdiag = dpp_xor<X>(w[0]), pd = dpp_xor<X>(dme), poff = dpp_xor<X>(w[X]);
    const float apq_s = lo ? w[X] : poff;
    float mine = 0.f, c = 1.f;
    if (apq_s != 0.f && !frozen) {
        const float apq = (dme * pd) * apq_s, a_me = (dme * dme) * w[0], a_pt = (pd * pd) * pdiag;
        // 1-ulp hardware reciprocal / sqrt / rsqrt: a rotation only has to be orthogonal to working precision, not the exact minimiser
        const float theta = (a_pt - a_me) * __builtin_amdgcn_rcpf(2.f * apq);
        if (fabsf(theta) < 1e18f) { // (otherwise theta^2 overflows: the rotation is the identity to fp32)
            float t = __builtin_amdgcn_rcpf(fabsf(theta) + __builtin_amdgcn_sqrtf(fmaf(theta, theta, 1.f)));
            t = (lo ? theta < 0.f : !(theta > 0.f)) ? -t : t;   // (theta == 0, equal diagonal elements: the upper lane still takes the opposite sign)
            c = __builtin_amdgcn_rsqf(fmaf(t, t, 1.f));
            mine = -t * (pd * __builtin_amdgcn_rcpf(dme));
        }
    }
    dme *= c;
    // the block row: column rotations (the coefficient of column me ^ y is the one lane me ^ y holds), then the row rotation
    const float rc1 = dpp_xor<1>(mine), rc2 = dpp_xor<2>(mine), rc3 = dpp_xor<3>(mine);
    const float rc[4] = { mine, rc1, rc2, rc3 };
    float nw[4];
#pragma unroll
    for (int y = 0; y < 4; ++y) nw[y] = fmaf(rc[y], w[y ^ X], w[y]);
#pragma unroll
    for (int y = 0; y < 4; ++y) w[y] = fmaf(mine, dpp_xor<X>(nw[y ^ X]), nw[y]);
    return mine;
}

// one super-round (see k_jacobi27_quads); YSTEP: the rows then move by JSY, otherwise by JSX
template <bool YSTEP>
__device__ __forceinline__ void jacobi_super_round(float (&vrow)[JLD], float *src, const float *wsrc, float *dst, float *dv_h, float *rot_h, int r,
                                                   int me, int snext, bool isRow, bool settled)
{
        float row[JLD];
#pragma unroll
        for (int q4 = 0; q4 < JLD / 4; ++q4) {
            float4 v = reinterpret_cast<const float4 *>(src)[q4];
            row[4 * q4] = v.x; row[4 * q4 + 1] = v.y; row[4 * q4 + 2] = v.z; row[4 * q4 + 3] = v.w;
        }
        // my row of the quad's diagonal block, xor-relative, and my scale
        float w[4] = { wsrc[me], wsrc[me ^ 1], wsrc[me ^ 2], wsrc[me ^ 3] };
        float dme = dv_h[isRow ? r : 0];
        const float m1 = quad_round<1>(w, dme, (me & 1) == 0, settled);
        const float m2 = quad_round<2>(w, dme, (me & 2) == 0, settled);
        const float m3 = quad_round<3>(w, dme, me == 0 || me == 1, settled);   // pairs (0,3), (1,2): the lower lane is 0 resp. 1
        if (isRow) {
            rot_h[r] = m1; rot_h[32 + r] = m2; rot_h[64 + r] = m3;
            dv_h[snext] = dme; // the scales move with their slots
        }
        wave_sync();
        // the three rounds on the rows of A~ (columns in registers, rows by DPP) and on the rows of V~ (columns)
#define BCD_ROUND(X, MINE)                                                                                                            \
        {                                                                                                                      \
            float rot[JLD];                                                                                                    \
            _Pragma("unroll") for (int q4 = 0; q4 < JLD / 4; ++q4) {                                                           \
                float4 w4 = reinterpret_cast<const float4 *>(rot_h + 32 * (X - 1))[q4];                                        \
                rot[4 * q4] = w4.x; rot[4 * q4 + 1] = w4.y; rot[4 * q4 + 2] = w4.z; rot[4 * q4 + 3] = w4.w;                    \
            }                                                                                                                  \
            _Pragma("unroll") for (int k = 0; k < JLD; ++k)                                                                    \
                if ((k & 3) < ((k ^ X) & 3)) {                                                                                 \
                    const int q = k ^ X;                                                                                       \
                    const float xv = vrow[k], yv = vrow[q];                                                                    \
                    vrow[k] = fmaf(rot[k], yv, xv); vrow[q] = fmaf(rot[q], xv, yv);                                            \
                }                                                                                                              \
            /* (pinned here: the optimiser otherwise sinks all of a sweep's updates of V~ to the end of the sweep and keeps the    */ \
            /* 27 rotation tables in scratch memory until then)                                                                 */ \
            _Pragma("unroll") for (int k = 0; k < JLD; ++k) asm volatile("" : "+v"(vrow[k]));                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                                 \
            _Pragma("unroll") for (int k = 0; k < JLD; ++k)                                                                    \
                if ((k & 3) < ((k ^ X) & 3)) {                                                                                 \
                    const int q = k ^ X;                                                                                       \
                    const float xa = row[k], ya = row[q];                                                                      \
                    row[k] = fmaf(rot[k], ya, xa); row[q] = fmaf(rot[q], xa, ya);                                              \
                }                                                                                                              \
            rowrot_quad<X>(row, MINE);                                                                                         \
        }
        BCD_ROUND(1, m1)
        __builtin_amdgcn_sched_barrier(0);
        BCD_ROUND(2, m2)
        __builtin_amdgcn_sched_barrier(0);
        BCD_ROUND(3, m3)
        __builtin_amdgcn_sched_barrier(0);
#undef BCD_ROUND
        // back to LDS at the next positions; the same move of the columns is a renaming of registers
        {
            float out[JLD], vout[JLD];
#pragma unroll
            for (int k = 0; k < JLD; ++k) { out[YSTEP ? JSY[k] : JSX[k]] = row[k]; vout[YSTEP ? JSY[k] : JSX[k]] = vrow[k]; }
#pragma unroll
            for (int k = 0; k < JLD; ++k) vrow[k] = vout[k];
            if (isRow) {
#pragma unroll
                for (int q4 = 0; q4 < JLD / 4; ++q4)
                    reinterpret_cast<float4 *>(dst)[q4] = make_float4(out[4 * q4], out[4 * q4 + 1], out[4 * q4 + 2], out[4 * q4 + 3]);
            }
        }
        wave_sync(); // the rows of A~ are back in LDS before the next super-round reads them
        __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(64, 3) void k_jacobi27_quads(const float *Ain, int n, int *work, float *__restrict__ eig,
                                                          float *__restrict__ Vout, float conv2, float *Aout, const int *d_n, int first_item)
{
    n = items_on_device(d_n, first_item, n);
    __shared__ float4 lds4[(2 * JQ_HALF + 32) / 4];
    float *lds = reinterpret_cast<float *>(lds4);
    int *place_tab = reinterpret_cast<int *>(lds + 2 * JQ_HALF);
    const int lane = threadIdx.x, h = lane >> 5, r = lane & 31;
    const bool isRow = r < KP;
    if (lane < KP) place_tab[lane] = JPLACE[lane];
    float *Ah = lds + h * JQ_HALF, *dv_h = Ah + JQ_MAT, *rot_h = dv_h + 32;
    const int rr_ = isRow ? r : JIDLE_ROW;
    const int sx = isRow ? JSX[rr_] : 0, sy = isRow ? JSY[rr_] : 0;
    float *src = Ah + JPLACE[rr_];                  // (idle lanes: a row of their own ds_read_b128 bank group)
    float *dst_x = Ah + JPLACE[sx], *dst_y = Ah + JPLACE[sy];
    const float *wsrc = src + (isRow ? (r & ~3) : 0);   // the quad's columns of my row
    const int me = r & 3;
    wave_sync();
    WorkCursor cursor = work_begin();
    for (;;) {
        const int pair = work_next(work, (n + 1) / 2, lane, cursor);
        if (pair < 0) break;
        const int first = 2 * pair;
        const int item = first + h;
        const bool live = item < n;
        // matrix -> LDS (an absent second matrix is the identity: converged from the start)
        {
            const float4 *gsrc = reinterpret_cast<const float4 *>(Ain + (size_t)(live ? item : first) * (KP * JLD));
            for (int e = r; e < KP * JLD / 4; e += 32) {
                float4 v = gsrc[e];
                const int e0 = 4 * e, rr = e0 / JLD, c0 = e0 - rr * JLD;
                if (!live) v = make_float4(rr == c0, rr == c0 + 1, rr == c0 + 2, rr == c0 + 3);
                *reinterpret_cast<float4 *>(Ah + place_tab[rr] + c0) = v;
            }
            if (isRow) dv_h[r] = 1.f;
        }
        wave_sync();
        float vrow[JLD];
#pragma unroll
        for (int k = 0; k < JLD; ++k) vrow[k] = (k == r) ? 1.f : 0.f; // V = identity
        for (int sweep = 0; sweep < 12; ++sweep) {
            // fold the scales: A~ <- D A~ D, V~ <- V~ D, D <- I; off / diagonal norms of the true matrix on the way (as k_jacobi27_batch)
            float off = 0.f, dg = 0.f;
            {
                float dk[JLD];
#pragma unroll
                for (int q4 = 0; q4 < JLD / 4; ++q4) {
                    float4 w4 = reinterpret_cast<const float4 *>(dv_h)[q4];
                    dk[4 * q4] = w4.x; dk[4 * q4 + 1] = w4.y; dk[4 * q4 + 2] = w4.z; dk[4 * q4 + 3] = w4.w;
                }
                const float dr = dv_h[isRow ? r : 0];
                float row[JLD];
#pragma unroll
                for (int q4 = 0; q4 < JLD / 4; ++q4) {
                    float4 v = reinterpret_cast<const float4 *>(src)[q4];
                    row[4 * q4] = v.x; row[4 * q4 + 1] = v.y; row[4 * q4 + 2] = v.z; row[4 * q4 + 3] = v.w;
                }
#pragma unroll
                for (int k = 0; k < JLD; ++k) {
                    row[k] *= dr * dk[k];
                    vrow[k] *= dk[k];
                    if (isRow) { if (k == r) dg = fmaf(row[k], row[k], dg); else off = fmaf(row[k], row[k], off); }
                }
                wave_sync(); // every lane has read the scales and its row
                if (isRow) {
#pragma unroll
                    for (int q4 = 0; q4 < JLD / 4; ++q4)
                        reinterpret_cast<float4 *>(src)[q4] = make_float4(row[4 * q4], row[4 * q4 + 1], row[4 * q4 + 2], row[4 * q4 + 3]);
                    dv_h[r] = 1.f;
                }
                wave_sync();
            }
            off = half_sum_swz(off);
            dg = half_sum_swz(dg);
            // (threshold and freezing of a converged matrix: as k_jacobi27_batch)
            const bool settled = off <= conv2 * dg;
            if (__builtin_amdgcn_ballot_w64(!settled) == 0) break; // both matrices converged
            // x x y x x y x x y: one sweep, every slot home again
            jacobi_super_round<false>(vrow, src, wsrc, dst_x, dv_h, rot_h, r, me, sx, isRow, settled);
            jacobi_super_round<false>(vrow, src, wsrc, dst_x, dv_h, rot_h, r, me, sx, isRow, settled);
            jacobi_super_round<true>(vrow, src, wsrc, dst_y, dv_h, rot_h, r, me, sy, isRow, settled);
            jacobi_super_round<false>(vrow, src, wsrc, dst_x, dv_h, rot_h, r, me, sx, isRow, settled);
            jacobi_super_round<false>(vrow, src, wsrc, dst_x, dv_h, rot_h, r, me, sx, isRow, settled);
            jacobi_super_round<true>(vrow, src, wsrc, dst_y, dv_h, rot_h, r, me, sy, isRow, settled);
            jacobi_super_round<false>(vrow, src, wsrc, dst_x, dv_h, rot_h, r, me, sx, isRow, settled);
            jacobi_super_round<false>(vrow, src, wsrc, dst_x, dv_h, rot_h, r, me, sx, isRow, settled);
            jacobi_super_round<true>(vrow, src, wsrc, dst_y, dv_h, rot_h, r, me, sy, isRow, settled);
        }
        if (live) {
            if (isRow) eig[(size_t)item * KP + r] = src[r];
        }
        if (Aout) { // (wave-uniform)
            if (isRow) rot_h[r] = src[r];
            wave_sync();
            if (isRow && live) jacobi_write_correction(Aout + (size_t)item * (KP * JLD) + r * JLD, src, rot_h, r);
        }
        if (live) {
            if (r < K) {
                float4 *o = reinterpret_cast<float4 *>(Vout + (size_t)item * (KP * JLD) + r * JLD);
#pragma unroll
                for (int q4 = 0; q4 < JLD / 4; ++q4) o[q4] = make_float4(vrow[4 * q4], vrow[4 * q4 + 1], vrow[4 * q4 + 2], vrow[4 * q4 + 3]);
            }
        }
        wave_sync(); // the next pair reuses the LDS
    }
}

__device__ void add_noise27(float *M, const float *noise, int lane, float sign)
{
    for (int t = lane; t < P * 9; t += 64) {
        int blk = t / 9, e = t - blk * 9, i = e / 3, j = e - i * 3;
        M[(3 * blk + i) * LD + 3 * blk + j] += sign * noise[blk * 6 + noise_idx(i, j)];
    }
    __syncthreads();
}

// in-place inverse of the symmetric positive definite M (LD layout) by the sweep operator on the matrix core.  Sweeping pivot k,
//     N_kk = -1 / d,   N_rk = N_kr = m_rk / d,   N_rc = m_rc - m_rk m_kc / d      (d = m_kk; after all 27 pivots N = -M^-1),
// is ONE rank-1 update N = M - u u^T / d - 2 e_k e_k^T with u = (column k of M) - e_k, i.e. one v_mfma_f32_32x32x2_f32 (exact f32 fma)
// on a matrix that stays in the accumulator registers for all 27 steps.  In the C/D layout lane (j, h) holds column j, rows
// (e & 3) + 8 (e >> 2) + 4 h; the matrix is symmetric, so the lanes of half h_k = (k >> 2) & 1 already hold u_j in register e_k: they
// feed it as the B operand of k-slot h_k and, times -1 / d, as the A operand; the other half's k-slot carries the -2 e_k e_k^T term.  No
// lane exchange, no LDS, ~12 instructions per pivot (the pivot itself comes through v_readlane; 1 / d = hardware reciprocal + one Newton step: the pivots of
// a Gauss-Jordan sweep need no correctly rounded quotient).  The previous form -- lane r owns row r, the pivot row broadcast element
// by element with v_readlane + v_fma -- took 56 instructions per pivot on 27 of the 64 lanes, 3 000 of the ~6 800 vector instructions
// of a full estimate's finish (r3 counters).
// Returns false (wave-uniform) if a pivot is not positive or the bound ||M^-1||_F * min_eig <= 1 fails (then lambda_min >= min_eig
// is not proven); M is then unspecified.

__device__ inline float bcast_lane(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }

// the 27 sweeps on a matrix in the accumulator layout; acc <- -(M^-1); returns the wave-uniform verdict (see above)
__device__ inline bool sweep_regs(v16f &acc, int idx, int h, float min_eig)
{
    // (the accumulator is only ever written by the matrix core: patching one of its elements from the vector unit makes the compiler move
    // all 16 registers out of and back into the accumulator file around every step.  The -2 e_k e_k^T term therefore rides in the k-slot
    // of the OTHER half, which would feed zeros: A = -2 delta_ik, B = delta_jk.)
    const float half0 = h == 0 ? 1.f : 0.f, half1 = 1.f - half0;
    const float sgn0 = h == 0 ? -1.f : 1.f, sgn1 = -sgn0; // feeding half: u = m - e_k; other half: + e_k
    bool ok = true;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int hk = (k >> 2) & 1, ek = (k & 3) + 4 * (k >> 3);
        const float v = acc[ek];
        const float d = bcast_lane(v, k + 32 * hk);
        ok = ok && (d > 0.f);
        float inv_d = __builtin_amdgcn_rcpf(d);
        inv_d = fmaf(fmaf(-d, inv_d, 1.f), inv_d, inv_d);
        const float hm = hk ? half1 : half0;
        const float ek_signed = (idx == k) ? (hk ? sgn1 : sgn0) : 0.f;
        const float bop = fmaf(v, hm, ek_signed);                  // feeding half: u_j = m_kj - delta_jk; other half: delta_jk
        const float aop = bop * ((h == hk) ? -inv_d : -2.f);       // feeding half: -u_i / d;               other half: -2 delta_ik
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, bop, acc, 0, 0, 0);
    }
    float fro = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) fro = fmaf(acc[e], acc[e], fro);
    fro = wsum(fro);
    return ok && isfinite(fro) && sqrtf(fro) * min_eig <= 1.f;
}

__device__ bool sweep_inverse27(float *M, int lane, float min_eig)
{
    LDS_POINTER(M);
    const int idx = lane & 31, h = lane >> 5;
    const float *col = M + (idx < K ? idx : 0) + 4 * h * LD;
    v16f acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r0 = (e & 3) + 8 * (e >> 2); // row of half 0; half 1: + 4
        const bool inside = idx < K && r0 + 4 * h < K;
        acc[e] = inside ? col[r0 * LD] : 0.f;
    }
    const bool ok = sweep_regs(acc, idx, h, min_eig);
    __syncthreads(); // (every lane has read M)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r0 = (e & 3) + 8 * (e >> 2);
        if (idx < K && r0 + 4 * h < K) M[(r0 + 4 * h) * LD + idx] = -acc[e];
    }
    __syncthreads();
    return ok;
}

// compact in-place two-sided Jacobi on LD-layout matrices (round-robin pairs, everything through LDS): only used by the rare
// spectral fallback of inverse27 (LD layout in, LD layout out, no conversion)
__device__ void jacobi27_inplace(float *A, float *V, float *prm /* 4 * KP/2 floats */, int lane)
{
    LDS_POINTER(A); LDS_POINTER(V); LDS_POINTER(prm);
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
    LDS_POINTER(M); LDS_POINTER(S0); LDS_POINTER(S1); LDS_POINTER(fl); LDS_POINTER(prm);
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
// (one element per lane and pass.  A column-per-lane form with compile-time block indices -- a third of the instructions, 27 lanes of
// each half busy -- was measured in r3: the finish kernel went from 0.79 to 0.86 ms; the phase waits on LDS latency, not on issue slots)
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
    LDS_POINTER(out); LDS_POINTER(X); LDS_POINTER(Y);
    if (SCALE) LDS_POINTER(scale);
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

// ---- clampNegativeEigenValues (:606-630) from an eigensolver that stopped EARLY (round 4) ------------------------------------------------
// The solver leaves A_k = V^T A V with a residual off-diagonal part E (off^2 <= 2e-9 diag^2 instead of 1e-12: five sweeps instead of six to
// seven for most matrices).  The positive part of A = V A_k V^T is V f(A_k) V^T, and f(D + E) = f(D) + E o Phi + O(|E|^2) with the divided differences
//     Phi_ij = (f(d_i) - f(d_j)) / (d_i - d_j),   f = max(0, .):   1 if both eigenvalues are positive, 0 if neither is, d+ / (d+ - d-) across zero
// (Daleckii-Krein; Phi in [0, 1], no small denominators: f is linear on either side of zero, so a pair on one side is exact whatever its gap).
// What the correction cannot absorb is a residual between two estimates of opposite sign that are closer to each other than the residual is
// large (error ~ |e_ij|, like the plain form): frames with ill-conditioned inverses feel the threshold -- BASELINE configs[3]'s 8-spp 4K frame is
// 9.8e-6 from the oracle at 2e-9 (1.6e-5 at 1e-8; 2.9e-6 with the plain rule at 1e-12; 2.9e-4 at 1e-6 corrected); the 32-spp frames stay at 4e-7 ... 9e-7.
// A per-pair stopping rule (every opposite-sign pair decoupled, loose bound on the rest) was built and measured out: in fp32 it needs the
// sixth sweep as often as the plain rule does, and its test costs 5 % of a sweep.
constexpr float JACOBI_CONV2_CORRECTED = 2e-9f; // off / diag <= 4.5e-5
constexpr bool FOLD_MAIN_TERM = true;             // k_finish27w: V (diag f + E o Phi) V^T as two products instead of three (round 5)
constexpr float JACOBI_CONV2_STRICT = 1e-12f;   // the round-3 rule: one sweep more, the correction then has nothing left to do


// Bm (LD layout) <- V f(D) V^T + V (E o Phi) V^T; Mb: 28 x 28 scratch (JLD layout), eigs: 28 floats of scratch, V: LDS copy of the record's V (JLD
// layout), recAk / recEig: the solver's output in the record (global memory)
__device__ __attribute__((noinline)) void pos_part_lds(float *Bm, float *Mb, float *eigs, const float *V, const float *__restrict__ recAk,
                                                       const float *__restrict__ recEig, int lane)
{
    LDS_POINTER(Bm); LDS_POINTER(Mb); LDS_POINTER(eigs); LDS_POINTER(V);
    if (lane < KP) eigs[lane] = recEig[lane];
    __syncthreads();
    for (int e = lane; e < KP * JLD; e += 64) Mb[e] = recAk[e]; // the correction matrix E o Phi (zero diagonal), as the eigensolver wrote it
    __syncthreads();
    if (lane < KP) eigs[lane] = fmaxf(0.f, eigs[lane]);
    __syncthreads();
    // the main term as ever: a sum of positive multiples of v v^T, bitwise symmetric; the correction (~1e-4 of it) through two plain products
    mfma27<true, true>(Bm, LD, V, JLD, V, JLD, eigs, KP, lane);          // Bm = V max(0, d) V^T
    mfma27<true, false>(Mb, LD, Mb, JLD, V, JLD, nullptr, KP, lane);     // Mb = (E o Phi) V^T        (all operands are in registers before the first store)
    mfma27<false, false>(Mb, LD, V, JLD, Mb, LD, nullptr, K, lane);      // Mb = V (E o Phi) V^T
    for (int e = lane; e < K * K; e += 64) { const int r = e / K, c = e - r * K; Bm[r * LD + c] += Mb[r * LD + c]; }
    __syncthreads();
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
// (the matrix leaves the accumulator registers for the per-pixel records: C in the LD layout, and C - N -- the matrix whose
// negative eigenvalues are clamped, Step 1 (:421-436) -- in the eigensolver's 28 x 28 layout, padding row / column zero)
__device__ __attribute__((noinline)) void covariance27(float *__restrict__ recA, float *__restrict__ recC, float *chunk, const float *mean,
                                                       const float *noise, const float *__restrict__ colors, const uint16_t *mem, int p, int b, int n,
                                                       int W, int lane)
{
    LDS_POINTER(chunk); LDS_POINTER(mean); LDS_POINTER(noise); LDS_POINTER(mem);
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
        const int mo = mi / 3, mj = mi - 3 * mo;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = (e & 3) + 8 * (e >> 2) + 4 * mk;
            const float v = acc[e] * inv; // rows / columns 27..31 of the product are exactly zero (zero operands)
            if (r < K && mi < LD) recC[r * LD + mi] = v;
            if (r < KP && mi < KP) {
                const int ro = r / 3, ri = r - 3 * ro;
                // the block-diagonal noise covariance: 3 x 3 block of pixel ro (symmetric storage xx,yy,zz,yz,xz,xy)
                const float nv = (ro == mo && r < K && mi < K) ? noise[ro * 6 + noise_idx(ri, mj)] : 0.f;
                recA[r * JLD + mi] = v - nv;
            }
        }
    }

}

// y = G2 (x - m) for one staged chunk on the matrix core, and its aggregation (see the call site)
__device__ __attribute__((noinline)) void final_chunk27(const float *Cm, const float *chunk, const float *mean, const uint16_t *mem, int i0, int cn, int p,
                                                        int W, int b, bool in_lds, int AW, int b1, float *accS, int *accC, float *sum,
                                                        int32_t *cnt, int lane)
{
    LDS_POINTER(Cm); LDS_POINTER(chunk); LDS_POINTER(mean); LDS_POINTER(mem); LDS_POINTER(accS); LDS_POINTER(accC);
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
                            accS[wq * 3 + ch] += v;
                            asm volatile("" ::: "memory"); // (not ds_add_f32: see k_bayes27w; the 64 addresses of one instruction are distinct -- members are whole
                                                    // pixels apart, the two halves' components are of different channels)
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
    LDS_POINTER(Cm); LDS_POINTER(chunk); LDS_POINTER(mean); LDS_POINTER(mem); LDS_POINTER(accS); LDS_POINTER(accC);
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

// Per processed pixel the estimate is three kernels; between them a pixel's state lives in a record in HBM (~9.9 KB):
//   PHASE 1  k_bayes27<1>  members, noise / colour means, covariance C; writes  A = C - N (28 x 28, zero-padded), C, N, m
//   (k_jacobi27_batch: eigen-decomposition of every A, two per wavefront)
//   PHASE 2  k_bayes27<2>  reads C, N, m, eigenvalues and eigenvectors; clamp, inverses, Step 2, final estimate, aggregation.
// Splitting costs ~20 KB of HBM traffic per pixel (nothing next to the ~300 us a pixel takes) and buys a solver that is not
// tied to the LDS footprint and register pressure of the other phases (12 -> 24 matrices per CU, shared rotations).
struct Records27 {
    float *A, *V, *C, *aux, *eig; // [items][784], [items][784], [items][784], [items][AUX27], [items][28]
};
constexpr int AUX27 = 96; // noise 54, mean 27, padding

template <int PHASE>
__global__ __launch_bounds__(64) void k_bayes27(const float *__restrict__ colors, const float *__restrict__ pixcov,
                                                const uint32_t *__restrict__ mask, const int32_t *__restrict__ list,
                                                int first_item, int nb_items, int *work, Geom27 g, float min_eig, Records27 rec, float *sum,
                                                int32_t *cnt, const int *redo = nullptr /* PHASE 2, optional: [0] count, [1..] the items to process */,
                                                const int *d_n = nullptr)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    nb_items = items_on_device(d_n, first_item, nb_items);
    if (PHASE == 2 && redo) nb_items = redo[0];
    // PHASE 1 needs one matrix buffer only (the member chunk; its matrices go from the accumulator registers to the records) and
    // runs 20 wavefronts per CU instead of 12, which is what its member gathers want (half of its wave cycles wait for memory;
    // 28 per CU were measured: no faster, and 4 KB instead of 7 KB per wavefront leave LDS to the kernels running beside it)
    float *Cm = lds, *A = Cm + MSZ, *V = A + MSZ, *Bm = PHASE == 1 ? lds : V + MSZ; // A, V contiguous: reused as the aggregation window
    float *chunk = Bm;                         // member-staging chunk: Bm is free while the clouds are streamed (mean/covariance, output)
    float *cs = Bm + MSZ;                      // 2 x 28 floats (scratch of the spectral inverse), read as float4: offsets are multiples of 16 bytes
    float *noise = cs + 2 * KP;
    float *mean = noise + P * 6;
    float *fl = mean + K + 1;
    uint16_t *mem = reinterpret_cast<uint16_t *>(fl + KP);

    // persistent wavefronts: items are handed out through the work queues
  WorkCursor cursor = work_begin();
  for (;;) {
    int slot = work_next(work, nb_items, lane, cursor);
    if (slot < 0) break;
    if (PHASE == 2 && redo) slot = __builtin_amdgcn_readfirstlane(redo[1 + slot]);
    const int p = list[first_item + slot];
    const int n = decode_members27(mask, p, g, mem, lane);
    const int W = g.W;
    float *recA = rec.A + (size_t)slot * MSZ, *recC = rec.C + (size_t)slot * MSZ, *recX = rec.aux + (size_t)slot * AUX27;

  if (PHASE == 1) {
    const float n_inv = 1.f / (float)n;
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
    // centerPointCloud + empiricalCovarianceMatrix (:511-536) on the matrix core: C = Xc^T Xc, two members per
    // v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: a chain of fma in member order, i.e. the reference's sequential sum with the
    // product fused; A and B operands are the same centred value, so the result is bitwise symmetric).
    // Operand layout: lane l holds element i = l & 31 of member 2s + (l >> 5); rows/columns 27..31 are zero.
    covariance27(recA, recC, chunk, mean, noise, colors, mem, p, g.b, n, W, lane);
    if (lane < P * 6) recX[lane] = noise[lane];
    if (lane < K) recX[P * 6 + lane] = mean[lane];
    __syncthreads(); // the next item reuses the LDS
  } else {
    // ---- state of PHASE 1 and the eigen-decomposition
    {
        const float *recV = rec.V + (size_t)slot * MSZ;
        for (int e = lane; e < MSZ / 4; e += 64) {
            reinterpret_cast<float4 *>(Cm)[e] = reinterpret_cast<const float4 *>(recC)[e];
            reinterpret_cast<float4 *>(V)[e] = reinterpret_cast<const float4 *>(recV)[e];
        }
        if (lane < P * 6) noise[lane] = recX[lane];
        if (lane < K) mean[lane] = recX[P * 6 + lane];
        if (lane < KP) fl[lane] = fmaxf(0.f, rec.eig[(size_t)slot * KP + lane]); // clampNegativeEigenValues (:606-630)
        __syncthreads();
    }
    // ---- Step 1 (:421-436), second half: M1 = clamp(C - N) + N ; Cinv1 = inverse(M1)
    pos_part_lds(Bm, A, fl, V, recA, rec.eig + (size_t)slot * KP, lane); // V (max(0, lambda) + residual correction) V^T: clampNegativeEigenValues (:606-630)
    add_noise27(Bm, noise, lane, +1.f);
    inverse27(Bm, A, V, fl, cs, lane, min_eig);
    // ---- Step 2 (:438-453): the Step-1 estimates are xhat = x - G (x - m) with G = N Cinv1, hence their empirical
    // mean is m and their empirical covariance is F C F^T, F = I - G
    noise_times27(V, noise, Bm, lane, true);       // V  = F
    mfma27<false, false>(A, LD, V, LD, Cm, LD, nullptr, K, lane);  // A  = F C
    mfma27<true, false>(Bm, LD, A, LD, V, LD, nullptr, K, lane);   // Bm = F C F^T
    for (int e = lane; e < K * K; e += 64) {       // exact symmetry (lower triangle wins; reads and writes touch different elements)
        int r = e / K, c = e - r * K;
        if (r < c) Bm[r * LD + c] = Bm[c * LD + r];
    }
    __syncthreads();
    add_noise27(Bm, noise, lane, +1.f);
    inverse27(Bm, A, V, fl, cs, lane, min_eig);
    noise_times27(Cm, noise, Bm, lane, false);     // Cm = G2 = N Cinv2

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
    __syncthreads(); // the next item reuses the LDS
  }
  }
}


// =====================================================================================================================
// Windowed variant for the default search radius (b = 6): every patch of every member of S(p) lies inside the 15 x 15 pixel
// window around p.  The kernels stage that window ONCE per pixel -- colours (2.7 KB; PHASE 1 also the per-pixel covariances,
// 5.4 KB) with 33 / 11 fully coalesced loads (window rows are contiguous in the image) -- and every per-member access becomes
// an LDS read: the r2 kernels gathered 108 scattered floats per member from global memory in PHASE 1 (half of its wave cycles
// waited for them; r3 counters) and 27 per member in the output pass of PHASE 2.  Members are kept as WINDOW PIXEL INDICES
// (kl + 1) * 15 + (kc + 1), which index the colour / covariance windows and the aggregation window alike.
// =====================================================================================================================
constexpr int WB = 6, WSIDE = 2 * WB + 1, WAW = WSIDE + 2, WPIX = WAW * WAW;   // 13, 15, 225
constexpr int WMEM = ((WSIDE * WSIDE + 7) / 8) * 8;                             // member list, padded to whole 16-byte reads
// (round 4) the same window for any search radius B -- the register-resident finish kernel also serves b = 12 (27 x 27 pixels): SIDE search
// positions per line, AW window pixels per line, G = floats between the end of one patch line and the start of the next in the window
constexpr int WIN_SLICE = 11;   // loads per lane and slice when a window is staged
template <int B> struct WinT {
    static constexpr int SIDE = 2 * B + 1, AW = SIDE + 2, PIX = AW * AW, G = AW * 3 - 9;
    static constexpr int MEM = ((SIDE * SIDE + 7) / 8) * 8;                       // member codes (uint16), padded to whole 16-byte reads
    static constexpr int NWORDS = (SIDE * SIDE + 31) / 32, NBITS = (SIDE * SIDE + 63) / 64; // mask words of a pixel; window positions per lane
    static constexpr int CSLICES = (PIX * 3 + 64 * WIN_SLICE - 1) / (64 * WIN_SLICE);   // slices of the colour window
    // LDS layout of k_finish27w (floats): colour window | sums | counts | noise | mean | member codes
    static constexpr int F2_ACCS = (PIX * 3 + 3) / 4 * 4, F2_ACCC = 2 * F2_ACCS, F2_NOISE = F2_ACCC + (PIX + 3) / 4 * 4;
    static constexpr int F2_MEAN = F2_NOISE + 56, F2_MEM = F2_MEAN + 32;
    static constexpr size_t F2_BYTES = (size_t)F2_MEM * sizeof(float) + MEM * sizeof(uint16_t);
};
static_assert(WinT<WB>::AW == WAW && WinT<WB>::PIX == WPIX && WinT<WB>::MEM == WMEM && WinT<WB>::G == 36 && WinT<WB>::CSLICES == 1, "b = 6 is the geometry the r3 kernels were written for");
// LDS layout of PHASE 1 (floats): colour window | covariance window | noise | mean | members (16-byte aligned)
// (b = 12: no covariance window -- 17.5 KB more per wavefront left 5 wavefronts per CU and a kernel slower than the gather form, measured; the
// per-pixel covariances of the members are gathered from memory there and the area only holds the 28 x 29 result tile)
template <int B> struct W1L {
    static constexpr bool COV_WINDOW = B == WB;
    static constexpr int NWIN = (WinT<B>::PIX * 3 + 3) / 4 * 4, NOISE = NWIN + ((COV_WINDOW ? WinT<B>::PIX * 6 : 28 * 29) + 3) / 4 * 4, MEM = NOISE + 56 + 28;
    static constexpr int NSLICES = (WinT<B>::PIX * 6 + 64 * WIN_SLICE - 1) / (64 * WIN_SLICE);   // slices of the covariance window
    static constexpr int MEM_SLOTS = WinT<B>::MEM + 16;  // (the covariance loop reads two groups of 8 ahead)
    static constexpr size_t BYTES = (size_t)MEM * sizeof(float) + MEM_SLOTS * sizeof(uint16_t);
};
constexpr int W1_MEM = W1L<WB>::MEM;
constexpr int W2_MEM = 4 * 784 + 56 + 56 + 28 + 28;                              // PHASE 2: four matrix buffers | cs | noise | mean | fl | members
static_assert(W1_MEM % 4 == 0 && W2_MEM % 4 == 0 && 28 * 29 <= WPIX * 6, "aligned member lists; the covariance tile fits the covariance window");
static_assert(W1L<12>::MEM % 4 == 0 && W1L<12>::BYTES <= 14 * 1024, "b = 12: a 27 x 27 pixel colour window, eleven wavefronts per CU");

// similar set of p in window order -> mem[i] = window pixel index of member i; returns |S|.  The mask words of p are wave-uniform
// (scalar loads); lane l looks at the bits l, l + 64, l + 128 and places its members by prefix population counts.
// (SCALE: the codes are stored multiplied by it -- the prepare kernel keeps float offsets into the colour window, 3 per pixel)
template <int B = WB, int SCALE = 1>
__device__ inline int decode_members_win(const uint32_t *__restrict__ mask, int p, int words, uint16_t *mem, int lane)
{
    using G_ = WinT<B>;
    constexpr int NW = 2 * G_::NBITS; // (an even number of words: lane l looks at bit l & 31 of word 2 j + (l >> 5))
    uint32_t wd[NW];
    int pre[NW + 1];
    pre[0] = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        wd[i] = i < words ? mask[(size_t)p * words + i] : 0u;
        pre[i + 1] = pre[i] + __popc(wd[i]);
    }
    const int hi = lane >> 5, bit = lane & 31;
    const uint32_t below = (1u << bit) - 1u;
#pragma unroll
    for (int j = 0; j < G_::NBITS; ++j) {
        const uint32_t w = hi ? wd[2 * j + 1] : wd[2 * j];
        const int base = hi ? pre[2 * j + 1] : pre[2 * j];
        if ((w >> bit) & 1u) {
            const int k = lane + 64 * j;
            const int kl = k / G_::SIDE, kc = k - kl * G_::SIDE;          // (division by a constant: a multiply and a shift)
            mem[base + __popc(w & below)] = (uint16_t)(((kl + 1) * G_::AW + kc + 1) * SCALE);
        }
    }
    __syncthreads();
    return pre[NW];
}

// one 15 x 15 window of an interleaved image with D floats per pixel -> LDS (cells outside the image: unspecified; no member patch touches them),
// in slices of WIN_SLICE loads per lane: win_issue puts the loads of slice `first` in flight, win_commit stores them
template <int D, int B = WB>
__device__ inline void win_issue(float (&v)[WIN_SLICE], const float *__restrict__ img, int first, int pr, int pc, int W, int H, int lane)
{
    constexpr int AW = WinT<B>::AW, ROW = AW * D, N = AW * ROW;
    const int row0 = pr - (B + 1), col0 = pc - (B + 1);
    // (round 5) a window inside the image -- all but the pixels within b + 1 of the frame's edge -- needs no clamps and no pixel / channel split:
    // element e of the window is float wy * (W * D) + r of the image, counted from the window's first float (5 vector instructions per load
    // instead of ~22; pr and pc are wave-uniform, the branch is a scalar one)
    if (row0 >= 0 && col0 >= 0 && row0 + AW <= H && col0 + AW <= W) {
        const __attribute__((address_space(1))) float *origin = (const __attribute__((address_space(1))) float *)img + (row0 * W + col0) * D;
        const int line = W * D;
#pragma unroll
        for (int u = 0; u < WIN_SLICE; ++u) {
            const int e = min(lane + 64 * (first + u), N - 1);
            const int wy = e / ROW, r = e - wy * ROW;
            v[u] = origin[wy * line + r];
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < WIN_SLICE; ++u) {
        // (cells outside the image are never part of a member's patch: their content is irrelevant, so the address is clamped
        // into the image instead of branching around the load)
        const int e = min(lane + 64 * (first + u), N - 1);
        const int wy = e / ROW, r = e - wy * ROW, px = r / D, ch = r - px * D;
        const int gy = min(max(row0 + wy, 0), H - 1), gx = min(max(col0 + px, 0), W - 1);
        // (32-bit index: DeepImage indices are ints, checked by the host; the image is in the global address space -- inside an out-of-line step
        // the pointer would otherwise be a flat one)
        v[u] = ((const __attribute__((address_space(1))) float *)img)[(gy * W + gx) * D + ch];
    }
}
template <int D, int B = WB>
__device__ inline void win_commit(float *win, const float (&v)[WIN_SLICE], int first, int lane)
{
    constexpr int N = WinT<B>::PIX * D;
#pragma unroll
    for (int u = 0; u < WIN_SLICE; ++u) {
        const int e = lane + 64 * (first + u);
        if (e < N) win[e] = v[u];
    }
}
static_assert(WIN_SLICE * 64 >= WPIX * 3 && 2 * WIN_SLICE * 64 >= WPIX * 6, "one slice holds the colour window, two the covariance window");

// offset (floats, D per pixel) of component k of a patch vector relative to the member's centre pixel in a window image
template <int D, int B = WB> __device__ inline int patch_off(int o) { return ((o / 3 - 1) * WinT<B>::AW + (o % 3 - 1)) * D; }

// ---- PHASE 1 of the windowed kernel, in out-of-line steps (each gets the register file to itself) ----
// the colour window travels while the members are decoded, then the covariance window in two slices; returns |S|
// (round 5: any search radius B -- b = 12 stages 4 + 7 slices, 121 loads per lane in flight; with 28 KB of LDS per wavefront at most two share a SIMD,
// so the registers are there)
template <int B>
__device__ __attribute__((noinline)) int win_stage_phase1(float *cwin, float *nwin, uint16_t *mem, const float *__restrict__ colors,
                                                          const float *__restrict__ pixcov, const uint32_t *__restrict__ mask, int p, int pr, int pc,
                                                          int W, int H, int words, int lane)
{
    LDS_POINTER(cwin); LDS_POINTER(nwin); LDS_POINTER(mem);
    // (arguments of an out-of-line function travel in vector registers: the wave-uniform ones are made scalar again)
    pr = __builtin_amdgcn_readfirstlane(pr); pc = __builtin_amdgcn_readfirstlane(pc);
    W = __builtin_amdgcn_readfirstlane(W); H = __builtin_amdgcn_readfirstlane(H);
    // all 33 loads of the lane are in flight at once (one exposed round trip to memory per item instead of three)
    constexpr int CS = WinT<B>::CSLICES, NS = W1L<B>::COV_WINDOW ? W1L<B>::NSLICES : 0;
    float wv[CS][WIN_SLICE], wn[NS > 0 ? NS : 1][WIN_SLICE];
#pragma unroll
    for (int sl = 0; sl < CS; ++sl) win_issue<3, B>(wv[sl], colors, sl * WIN_SLICE, pr, pc, W, H, lane);
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) win_issue<6, B>(wn[sl], pixcov, sl * WIN_SLICE, pr, pc, W, H, lane);
    const int n = decode_members_win<B, 3>(mask, p, words, mem, lane);
#pragma unroll
    for (int sl = 0; sl < CS; ++sl) win_commit<3, B>(cwin, wv[sl], sl * WIN_SLICE, lane);
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) win_commit<6, B>(nwin, wn[sl], sl * WIN_SLICE, lane);
    __syncthreads();
    return n;
}

// computeNoiseCovPatchesMean (:400-419) and empiricalMean (:500-509): lanes 0..53 own one noise component, lanes 0..26 also one
// colour component; sums in member order.
// (round 5) The member codes are the same for every lane: they are moved to scalar registers (v_readfirstlane), so that unpacking them and
// the window offsets they stand for are scalar-unit work and a member costs the vector unit an address add, a read and an add per sum; whole
// groups of 8 need no "is this slot a member" selects, only the last, partial group does.  (Before: ~75 vector-unit cycles per member --
// unpacking, offsets, a compare and two selects per slot -- in a kernel that is bound by instruction issue.)
template <int B>
__device__ __attribute__((noinline)) void win_noise_mean(float *noise, float *mean, const float *cwin, const float *nwin, const uint16_t *mem, int n_, int lane,
                                                         const float *__restrict__ pixcov = nullptr, int win_origin = 0, int W = 0)
{
    LDS_POINTER(noise); LDS_POINTER(mean); LDS_POINTER(cwin); LDS_POINTER(nwin); LDS_POINTER(mem);
    const int n = __builtin_amdgcn_readfirstlane(n_);
    W = __builtin_amdgcn_readfirstlane(W);
    win_origin = __builtin_amdgcn_readfirstlane(win_origin);
    // (arguments of an out-of-line function travel in vector registers: the image pointer is made scalar again)
    const unsigned long long pixcov_bits = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)((unsigned long long)pixcov >> 32)) << 32)
                                           | (uint32_t)__builtin_amdgcn_readfirstlane((int)(unsigned long long)pixcov);
    const float n_inv = 1.f / (float)n;
    const int ln = min(lane, P * 6 - 1), lc = min(lane, K - 1);
    const float *nlane = nwin + (patch_off<6, B>(ln / 6) + ln % 6), *clane = cwin + (patch_off<3, B>(lc / 3) + lc % 3);
    // without a covariance window: the lane's component of the 3 x 3 patch whose TOP-LEFT pixel is pixel 0 (a non-negative byte offset)
    const uint32_t glane = (uint32_t)((((ln / 6) / 3) * W + (ln / 6) % 3) * 6 + ln % 6) * 4u;
    constexpr int AW = WinT<B>::AW, CENTRE = (B + 1) * AW + B + 1;
    float accn = 0.f, accm = 0.f;
    const int nfull = n & ~7;
    if (!W1L<B>::COV_WINDOW) {
        // the per-pixel covariances of the members come from memory, NG members per round trip, sums in member order.  Slots past |S| hold stale
        // codes of earlier items, which may lie outside the image for this one: they read the centre pixel instead.
        constexpr int NG = 32;
        for (int i = 0; i < n; i += NG) {
            float vn[NG];
#pragma unroll
            for (int j = 0; j < NG / 8; ++j)
                if (i + 8 * j < n) {
                    const uint4 c8 = *reinterpret_cast<const uint4 *>(mem + i + 8 * j);
                    const uint32_t cw[4] = { (uint32_t)__builtin_amdgcn_readfirstlane((int)c8.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)c8.y),
                                             (uint32_t)__builtin_amdgcn_readfirstlane((int)c8.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)c8.w) };
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int wp = (int)((cw[u >> 1] >> (16 * (u & 1))) & 0xffffu) / 3;
                        const int wq = (i + 8 * j + u < n) ? wp : CENTRE, wy = wq / AW, wx = wq - wy * AW;
                        // (a scalar base in the global address space + the lane's 32-bit offset: global_load_dword v, v, s[..])
                        typedef const __attribute__((address_space(1))) char *GlobalBytes;
                        typedef const __attribute__((address_space(1))) float *GlobalFloat;
                        GlobalBytes patch = (GlobalBytes)pixcov_bits + (long long)(win_origin + (wy - 1) * W + wx - 1) * 24;
                        vn[8 * j + u] = *(GlobalFloat)(patch + glane);
                    }
                }
#pragma unroll
            for (int j = 0; j < NG / 8; ++j)
                if (i + 8 * j < n) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) accn += (i + 8 * j + u < n) ? vn[8 * j + u] : 0.f;
                }
        }
    }
    for (int i = 0; i < nfull; i += 8) {
        const uint4 c8 = *reinterpret_cast<const uint4 *>(mem + i);
        const uint32_t cw[4] = { (uint32_t)__builtin_amdgcn_readfirstlane((int)c8.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)c8.y),
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)c8.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)c8.w) };
        float vn[8], vm[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int w3 = (int)((cw[u >> 1] >> (16 * (u & 1))) & 0xffffu); // 3 x the window pixel
            if (W1L<B>::COV_WINDOW) vn[u] = nlane[w3 * 2];
            vm[u] = clane[w3];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (W1L<B>::COV_WINDOW) accn += vn[u];
            accm += vm[u];
        }
    }
    if (nfull < n) { // the last, partial group (slots past |S| hold codes inside the window: read, not added; x + 0.f == x)
        const int i = nfull;
        const uint4 c8 = *reinterpret_cast<const uint4 *>(mem + i);
        const uint32_t cw[4] = { (uint32_t)__builtin_amdgcn_readfirstlane((int)c8.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)c8.y),
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)c8.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)c8.w) };
        float vn[8], vm[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int w3 = (int)((cw[u >> 1] >> (16 * (u & 1))) & 0xffffu); // 3 x the window pixel
            if (W1L<B>::COV_WINDOW) vn[u] = nlane[w3 * 2];
            vm[u] = clane[w3];
        }
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            if (W1L<B>::COV_WINDOW) accn += (i + u < n) ? vn[u] : 0.f;
            accm += (i + u < n) ? vm[u] : 0.f;
        }
    }
    __syncthreads(); // (every lane is done with the previous pixel's means)
    if (lane < P * 6) noise[lane] = accn * n_inv;
    if (lane < K) mean[lane] = accm * n_inv;
    __syncthreads();
}

// centerPointCloud + empiricalCovarianceMatrix (:511-536) on the matrix core, operands straight from the colour window: lane l feeds
// element i = l & 31 of member 2 s + (l >> 5) (rows / columns 27..31 zero); a chain of fma in member order, i.e. the reference's
// sequential sum with the product fused; A and B operands are the same centred value, so the result is bitwise symmetric.
// The product (x 1 / (n - 1)) leaves the accumulators for a 28 x 29 tile in LDS (row / column 27 exactly zero).
// (round 5: |S| in a scalar register -- the loop's tests are scalar branches, whole groups of 8 members run without them -- and the operands of
// the next group are read while the matrix core works on the current one; before, every product waited for its own LDS read)
template <int B>
__device__ __attribute__((noinline)) void win_covariance(float *tile, const float *cwin, const float *mean, const uint16_t *mem, int n_, int lane)
{
    LDS_POINTER(tile); LDS_POINTER(cwin); LDS_POINTER(mean); LDS_POINTER(mem);
    const int n = __builtin_amdgcn_readfirstlane(n_);
    const int mi = lane & 31, mk = lane >> 5, mic = min(mi, K - 1);
    const bool real = mi < K;
    const float my_mean = mean[mic];
    const float *alane = cwin + (patch_off<3, B>(mic / 3) + mic % 3);
    v16f acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // operands of the group of 8 members at s8: member s8 + 2 t + mk for t = 0..3 (slots past |S| hold codes inside the window: read, not used)
    auto fetch = [&](float (&av)[4], const uint4 c8) {
        const uint32_t cw[4] = { c8.x, c8.y, c8.z, c8.w };
#pragma unroll
        for (int t = 0; t < 4; ++t) av[t] = alane[(cw[t] >> (16 * mk)) & 0xffffu];   // (the codes are float offsets: 3 x the window pixel)
    };
    float av[4], nx[4];
    fetch(av, *reinterpret_cast<const uint4 *>(mem));
    uint4 codes = *reinterpret_cast<const uint4 *>(mem + 8);
    int s8 = 0;
    for (; s8 + 8 <= n; s8 += 8) {          // whole groups (the list has 16 slots of slack past its last group: W1L::MEM_SLOTS)
        fetch(nx, codes);                                               // operands one group ahead,
        codes = *reinterpret_cast<const uint4 *>(mem + s8 + 16);        // codes two groups ahead
        __builtin_amdgcn_sched_barrier(0);                              // (the reads stay in front of the products)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float a = real ? av[t] - my_mean : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) av[t] = nx[t];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {           // the last, partial group
        if (s8 + 2 * t < n) {                                           // (scalar)
            const float a = (s8 + 2 * t + mk < n && real) ? av[t] - my_mean : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc, 0, 0, 0);
        }
    }
    const float inv = 1.f / (float)(n - 1);
    // C/D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * mk;
        if (r < KP && mi < LD) tile[r * LD + mi] = acc[e] * inv;
    }
    __syncthreads();
}

// the records of PHASE 1 from the covariance tile: C (LD layout), A = C - N in the eigensolver's 28 x 28 layout (the matrix whose negative
// eigenvalues are clamped, Step 1 (:421-436); padding row / column zero), noise and mean -- whole 256-byte lines per store instruction
// Round 6: the records leave as 16-byte stores.  C (the 28 x 29 tile as it lies) is a straight copy; A = C - N is the tile with its 81 block-diagonal
// entries changed IN the tile (one wavefront: its LDS instructions execute in order) and then copied row by row into the 28-column record.  The loop this
// replaces decided per element whether it lies on the block diagonal (two divisions by three, a table look-up and a select for each of 784 entries: ~360
// vector instructions per item, as many as the member loops).
__device__ __attribute__((noinline)) void win_write_records(float *__restrict__ recA, float *__restrict__ recC, float *__restrict__ recX,
                                                            float *tile, const float *noise, const float *mean, int lane)
{
    LDS_POINTER(tile); LDS_POINTER(noise); LDS_POINTER(mean);
    static_assert((K * LD + 3) / 4 <= KP * LD / 4 && MSZ % 4 == 0, "whole 16-byte groups inside the tile and the record");
    for (int q = lane; q < (K * LD + 3) / 4; q += 64) reinterpret_cast<float4 *>(recC)[q] = reinterpret_cast<const float4 *>(tile)[q];
    // the block-diagonal noise covariance: 3 x 3 block of patch pixel ro (symmetric storage xx,yy,zz,yz,xz,xy)
    for (int t = lane; t < P * 9; t += 64) {
        const int ro = t / 9, ij = t - 9 * ro, i = ij / 3, j = ij - 3 * i;
        float *e = tile + (3 * ro + i) * LD + 3 * ro + j;
        *e -= noise[ro * 6 + noise_idx(i, j)];
    }
    for (int e = lane; e < KP * (JLD / 4); e += 64) {
        const int r = e / (JLD / 4), q = e - r * (JLD / 4);
        const float *src = tile + r * LD + 4 * q; // (rows of the tile are 29 floats apart: four 4-byte reads)
        reinterpret_cast<float4 *>(recA)[e] = make_float4(src[0], src[1], src[2], src[3]);
    }
    if (lane < P * 6) recX[lane] = noise[lane];
    if (lane < K) recX[P * 6 + lane] = mean[lane];
}

template <int PHASE, int B = WB>
__global__ __launch_bounds__(64, PHASE == 1 ? 5 : 3) void k_bayes27w(const float *__restrict__ colors, const float *__restrict__ pixcov,
                                                 const uint32_t *__restrict__ mask, const int32_t *__restrict__ list,
                                                 int first_item, int nb_items, int *work, Geom27 g, float min_eig, Records27 rec, float *sum,
                                                 int32_t *cnt, const int *redo /* PHASE 2, optional: [0] count, [1..] the items to process */,
                                                 const int *d_n = nullptr)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    nb_items = items_on_device(d_n, first_item, nb_items);
    if (PHASE == 2 && redo) nb_items = redo[0];
    // (PHASE 1 opens the chunk: it clears the redo list's counter for the finish kernel two launches later -- this was a fill between the kernels)
    if (PHASE == 1 && redo && blockIdx.x == 0 && lane == 0) const_cast<int *>(redo)[0] = 0;
    // PHASE 1: colour window | covariance window | noise | mean | members            (8.8 KB: 18 wavefronts per CU)
    // PHASE 2: Cm | A | V | Bm (matrix scratch, then the colour window) | cs | noise | mean | fl | members   (as the gather kernel)
    float *Cm = lds, *A = Cm + MSZ, *V = A + MSZ, *Bm = V + MSZ;
    float *cwin = PHASE == 1 ? lds : Bm;
    static_assert(PHASE == 1 || B == WB, "the LDS finish (PHASE 2) exists for the default search radius only");
    float *nwin = cwin + W1L<B>::NWIN;             // PHASE 1 only (later the 28 x 29 covariance tile)
    float *cs = Bm + MSZ;                          // PHASE 2 only
    float *noise = PHASE == 1 ? lds + W1L<B>::NOISE : cs + 2 * KP;
    float *mean = noise + 56;                      // (54 noise components; offsets stay multiples of 16 bytes)
    float *fl = mean + KP;                         // PHASE 2 only
    uint16_t *mem = reinterpret_cast<uint16_t *>(PHASE == 1 ? lds + W1L<B>::MEM : fl + KP);
    static_assert(WPIX * 3 <= MSZ, "the colour window fits a matrix buffer");
    for (int e = lane; e < (PHASE == 1 ? W1L<B>::MEM_SLOTS : WinT<B>::MEM); e += 64) mem[e] = (uint16_t)((WinT<B>::AW + 1) * (PHASE == 1 ? 3 : 1)); // (PHASE 1 keeps codes x 3; entries past |S| are read, never used: keep them inside the window)
    if (lane == 0) mean[K] = 0.f;
    __syncthreads();
    const int W = g.W, H = g.H;

  WorkCursor cursor = work_begin();
  for (;;) {
    int slot = work_next(work, nb_items, lane, cursor);
    if (slot < 0) break;
    if (PHASE == 2 && redo) slot = __builtin_amdgcn_readfirstlane(redo[1 + slot]);
    const int p = __builtin_amdgcn_readfirstlane(list[first_item + slot]);
    const int pr = p / W, pc = p - pr * W;
    float *recA = rec.A + (size_t)slot * MSZ, *recC = rec.C + (size_t)slot * MSZ, *recX = rec.aux + (size_t)slot * AUX27;

  if (PHASE == 1) {
    const int n = win_stage_phase1<B>(cwin, nwin, mem, colors, pixcov, mask, p, pr, pc, W, H, g.words, lane);
    win_noise_mean<B>(noise, mean, cwin, nwin, mem, n, lane, pixcov, (pr - (B + 1)) * W + pc - (B + 1), W);
    win_covariance<B>(nwin /* tile: the covariance window (if there is one) is dead by then */, cwin, mean, mem, n, lane);
    win_write_records(recA, recC, recX, nwin, noise, mean, lane);
    __syncthreads(); // the next item reuses the LDS
  } else {
    const int n = decode_members_win<WB>(mask, p, g.words, mem, lane);
    {
        const float *recV = rec.V + (size_t)slot * MSZ;
        for (int e = lane; e < MSZ / 4; e += 64) {
            reinterpret_cast<float4 *>(Cm)[e] = reinterpret_cast<const float4 *>(recC)[e];
            reinterpret_cast<float4 *>(V)[e] = reinterpret_cast<const float4 *>(recV)[e];
        }
        if (lane < P * 6) noise[lane] = recX[lane];
        if (lane < K) mean[lane] = recX[P * 6 + lane];
        if (lane < KP) fl[lane] = fmaxf(0.f, rec.eig[(size_t)slot * KP + lane]); // clampNegativeEigenValues (:606-630)
        __syncthreads();
    }
    // ---- Step 1 (:421-436), second half, and Step 2 (:438-453): as in k_bayes27<2>
    pos_part_lds(Bm, A, fl, V, recA, rec.eig + (size_t)slot * KP, lane); // V (max(0, lambda) + residual correction) V^T: clampNegativeEigenValues (:606-630)
    add_noise27(Bm, noise, lane, +1.f);
    inverse27(Bm, A, V, fl, cs, lane, min_eig);
    noise_times27(V, noise, Bm, lane, true);       // V  = F = I - N Cinv1
    mfma27<false, false>(A, LD, V, LD, Cm, LD, nullptr, K, lane);  // A  = F C
    mfma27<true, false>(Bm, LD, A, LD, V, LD, nullptr, K, lane);   // Bm = F C F^T
    for (int e = lane; e < K * K; e += 64) {       // exact symmetry (lower triangle wins; reads and writes touch different elements)
        int r = e / K, c = e - r * K;
        if (r < c) Bm[r * LD + c] = Bm[c * LD + r];
    }
    __syncthreads();
    add_noise27(Bm, noise, lane, +1.f);
    inverse27(Bm, A, V, fl, cs, lane, min_eig);
    // finalDenoisingMatrixMultiplication (:656-670) as xhat = m + F2 (x - m), F2 = I - N Cinv2 (the same affine map as x - G2 (x - m))
    noise_times27(Cm, noise, Bm, lane, true);      // Cm = F2
    // Bm is free: the colour window of p; A / V are dead: the aggregation window (sums 15 x 15 x 3, then counts 15 x 15)
    {
        float wv[WIN_SLICE];
        win_issue<3>(wv, colors, 0, pr, pc, W, H, lane);
        win_commit<3>(cwin, wv, 0, lane);
    }
    float *accS = A;
    int *accC = reinterpret_cast<int *>(A + WPIX * 3);
    for (int e = lane; e < WPIX * 4; e += 64) A[e] = 0.f;
    __syncthreads();
    // ---- output pass, 32 members per matrix-core product: D[r][j] = c[r] + sum_k F2[r][k] x_j[k] with c = m - F2 m (the same affine map).
    // The sum over k may run in any order as long as both operands agree: the lanes of half h take k = 14 h + s at step s, so that a lane
    // feeds A = F2[l & 31][14 h .. 14 h + 13] -- 14 registers, loaded once per item -- and B = 14 neighbouring components of the patch of
    // member l & 31 (read two at a time); it gets back components r = (e & 3) + 8 (e >> 2) + 4 h of that member, on top of c (the product's
    // initial accumulator: one extra product per item, with B = m for every column).
    {
        const int mj = lane & 31, kh = lane >> 5;
        float fa[14];
        {
            const float *frow = Cm + min(mj, K - 1) * LD + 14 * kh;   // (rows 27..31 of the A operand only reach rows 27..31 of D: unused)
#pragma unroll
            for (int s_ = 0; s_ < 14; ++s_) fa[s_] = frow[s_];
            if (kh) fa[13] = 0.f;                                   // k = 27: padding
        }
        v16f c0;
#pragma unroll
        for (int e = 0; e < 16; ++e) c0[e] = 0.f;
        {
            const float *mk_ = mean + 14 * kh;
#pragma unroll
            for (int s_ = 0; s_ < 14; ++s_) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s_], mk_[s_ < 13 ? s_ : 12 + (1 - kh)], c0, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * kh;
                c0[e] = mean[min(r, K - 1)] - c0[e];
            }
        }
        for (int i0 = 0; i0 < n; i0 += 32) {
            const bool valid = i0 + mj < n;
            const int wp = mem[valid ? i0 + mj : 0];
            const int base = (wp - WAW - 1) * 3;                // top-left pixel of the member's patch in the window, floats
            // component k of a patch vector sits at k + 36 (k / 9) floats from the patch's top-left pixel: k = 14 h + s is at
            //   s < 4: s + 50 h,    4 <= s < 9: s + 86 h,    9 <= s: s + 36 + 50 h
            const float *pa = cwin + base + 50 * kh, *pb = cwin + base + 86 * kh;
            v16f y = c0;
#pragma unroll
            for (int s_ = 0; s_ < 14; ++s_) {
                float bq = s_ < 4 ? pa[s_] : (s_ < 9 ? pb[s_] : pa[s_ + 36]);
                if (s_ == 13) bq = kh ? 0.f : bq;                  // k = 27: padding (the cell read in its place is a window cell, any content)
                y = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s_], bq, y, 0, 0, 0);
            }
            if (valid) {
                // component r = r0 + 4 h lands r + 36 (r / 9) floats into the patch: r0 + 36 (r0 / 9) plus 4 h -- or 40 h where r0 + 4 is on the
                // next patch line -- i.e. a compile-time offset from one of two per-lane bases: no address arithmetic per component
                float *d4 = accS + base + 4 * kh, *d40 = accS + base + 40 * kh;
#pragma unroll
                // Plain read-add-write, not ds_add_f32: the LDS float atomic is served one lane at a time on gfx950 (193 cycles of the CU's
                // LDS pipe per wavefront instruction, measured: tools/ubench/lds_rate.hip; a read is 2.5 and a write 4.7), and there is
                // nobody to be atomic against -- one wavefront per workgroup, and within one instruction the 64 addresses are distinct:
                // two members' patches are whole pixels (multiples of 3 floats) apart, the two halves' components 4 or 40 floats.
                for (int e = 0; e < 12; ++e) {                     // r0 <= 19: both halves inside the 27 components
                    const int r0 = (e & 3) + 8 * (e >> 2);
                    float *q = ((r0 + 4) / 9 != r0 / 9 ? d40 : d4) + r0 + 36 * (r0 / 9);
                    *q = *q + y[e];
                    // (the NEXT component of another lane may be this address: program order must be kept -- the compiler only reasons about
                    // one lane, where the addresses differ; the hardware serves a wavefront's LDS instructions in order)
                    asm volatile("" ::: "memory");
                }
                // the 9 pixels of the patch, counted once each: pixel q is (q % 3) + 15 (q / 3) cells from the top-left one; half 0 takes
                // q = 0..4, half 1 q = 5..8
                int *c17 = accC + (wp - WAW - 1) + 17 * kh, *c29 = accC + (wp - WAW - 1) + 29 * kh;
                atomicAdd(c17, 1);
                atomicAdd(c29 + 1, 1);
                atomicAdd(c29 + 2, 1);
                atomicAdd(c17 + 15, 1);
                if (!kh) {
#pragma unroll
                    for (int e = 12; e < 15; ++e) { float *q = d4 + 24 + (e - 12) + 72; *q = *q + y[e]; asm volatile("" ::: "memory"); } // r0 = 24..26 (half 1 would be 28..30)
                    atomicAdd(c17 + 16, 1);
                }
            }
        }
        __syncthreads();
    }
    // aggregateOutputPatches (:672-693): one global atomic per touched value of the window, rows contiguous
    {
        constexpr int row3 = WAW * 3;
        const long long base = (long long)p - (long long)(WB + 1) * W - (WB + 1); // window origin; untouched cells may lie outside the image
        for (int e = lane; e < WAW * row3; e += 64) {
            int wy = e / row3, r = e - wy * row3, wx = r / 3;
            if (accC[wy * WAW + wx] != 0) unsafeAtomicAdd(sum + (base + (long long)wy * W) * 3 + r, accS[e]);
        }
        for (int e = lane; e < WPIX; e += 64) {
            int wy = e / WAW, wx = e - wy * WAW, c = accC[e];
            if (c != 0) atomicAdd(cnt + (base + (long long)wy * W + wx), c);
        }
    }
    __syncthreads(); // the next item reuses the LDS
  }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_finish27w: FINISH of the windowed path with the 27 x 27 algebra in registers.  Every matrix of Steps 1 and 2 after the eigensolver
// is symmetric or enters a product next to a symmetric one, and a symmetric matrix in the accumulator layout of v_mfma_f32_32x32x2_f32
// (lane (j, h) holds M[r][j], r = (e & 3) + 8 (e >> 2) + 4 h, e = 0..15) is ALSO a valid A operand and a valid B operand of the next
// product when the sum over k runs in the order k = r(s, h), s = 0..15: lane (i, h) feeds A[i][k] = M[k][i], lane (j, h) feeds
// B[k][j] = M[k][j] -- the same register.  A product's result D comes back "by columns" (lane j holds column j), which is what a B
// operand wants and, read as D^T, what an A operand wants; so the chain is arranged to need only transposes it can have for free:
//     P    = (V max(0, lambda)) V^T + N         operands straight from the record in HBM (rows of V, 14 consecutive floats per lane)
//     -C1  = sweep(P)                           27 rank-1 products (sweep_regs)
//     F^T  = I - C1 N                           A = C1 (symmetric), B = N (block diagonal, per-lane constants of the item)
//     G^T  = C F^T                              A = C  (symmetric, from the record), B = F^T
//     H    = F G^T + N  ( = F C F^T + N )       A = F (= F^T read as an A operand), B = G^T
//     -C2  = sweep(H)
//     F2^T = I - C2 N                           -> the A operand of the output pass, xhat = (m - F2 m) + F2 x
// A sweep that fails its checks (a pivot not positive, lambda_min >= min_eig not proven) is the rare spectral branch of
// inverseSymmetricMatrix: the item is put on a list and left to k_bayes27w<2>, which is launched on that list afterwards.
// No matrix goes through LDS (the previous form -- k_bayes27<2>'s, kept for the other search radii -- spent 0.2 of its 0.5 ms per 32 768
// pixels in LDS round trips of these stages), and LDS holds only the windows: 7.4 KB per wavefront instead of 17.
// ---------------------------------------------------------------------------------------------------------------------
// (LDS layout: WinT<B>::F2_*.)  B = 6: 15 x 15 window, 7.4 KB per wavefront; B = 12 (round 4): 27 x 27, 22 KB -- seven wavefronts per CU.
template <int B>
__global__ __launch_bounds__(64, 3) void k_finish27w(const float *__restrict__ colors, const uint32_t *__restrict__ mask,
                                                     const int32_t *__restrict__ list, int first_item, int nb_items, int *work, Geom27 g,
                                                     float min_eig, Records27 rec, float *sum, int32_t *cnt, int *redo /* [0] count, [1..] items */,
                                                     int *redo_total /* statistics: items handed over, all launches of the scale */, const int *d_n)
{
    nb_items = items_on_device(d_n, first_item, nb_items);
    using G_ = WinT<B>;
    constexpr int AW = G_::AW, PIX = G_::PIX, GAP = G_::G; // window pixels per line / in all; floats from the end of a patch line to the start of the next
    extern __shared__ float lds[];
    const int lane0 = threadIdx.x;
    float *cwin = lds, *accS = lds + G_::F2_ACCS;
    int *accC = reinterpret_cast<int *>(lds + G_::F2_ACCC);
    float *noise = lds + G_::F2_NOISE, *mean = lds + G_::F2_MEAN;
    uint16_t *mem = reinterpret_cast<uint16_t *>(lds + G_::F2_MEM);
    for (int e = lane0; e < G_::MEM; e += 64) mem[e] = (uint16_t)(AW + 1); // (entries past |S| are read, never used: keep them inside the window)
    if (lane0 < 32) mean[lane0] = 0.f;                                   // (components 27..31: zero padding)
    __syncthreads();
    const int W = g.W, H = g.H;

    WorkCursor cursor = work_begin();
    for (;;) {
        const int slot = work_next(work, nb_items, lane0, cursor);
        if (slot < 0) break;
        // per-lane constants of the accumulator layout, derived afresh in every item from a lane number the compiler cannot see through:
        // hoisted out of the loop, the dozens of lane-dependent offsets and masks of an item stay live across all of it -- they were what
        // got spilled, and reloaded one by one with a wait each
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int idx = lane & 31, h = lane >> 5;
        const bool col_ok = idx < K;
        const int blk = min(idx, K - 1) / 3, jj = min(idx, K - 1) - 3 * blk;
        const int nz_off0 = blk * 6 + noise_idx(0, jj), nz_off1 = blk * 6 + noise_idx(1, jj), nz_off2 = blk * 6 + noise_idx(2, jj);
        const int u_blk = col_ok ? 3 * blk - 4 * h : -100;   // row r(e, h) is line t of the column's 3 x 3 noise block  <=>  r0(e) - u_blk == t
        const int u_eye = col_ok ? idx - 4 * h : -100;       // r(e, h) == idx  <=>  r0(e) == u_eye
        // ---- the item's reads from HBM are requested ahead of their use: the record's V rows, eigenvalues, noise and mean and the colour
        // window here; C just before the first sweep, whose 27 dependent products cover its latency.  (The stages below are fenced with
        // sched_barrier: left alone, the scheduler hoists every load of the item to the top and needs 264 registers -- 64 of them spilled
        // at three wavefronts per SIMD; in stage order the peak is about 110.)
        const float *recV = rec.V + (size_t)slot * MSZ, *recC = rec.C + (size_t)slot * MSZ, *recX = rec.aux + (size_t)slot * AUX27;
        // row `idx` of V and of the correction matrix E o Phi the eigensolver left in place of its input, and the eigenvalue estimates, at the
        // columns r(s, h) = (s & 3) + 8 (s >> 2) + 4 h of the accumulator layout: four 16-byte loads each
        float vB[16], aK[16], dK[16];
        {
            const float *recAk = rec.A + (size_t)slot * MSZ;
            const float4 *rowV = reinterpret_cast<const float4 *>(recV + min(idx, K - 1) * JLD + 4 * h);
            const float4 *rowA = reinterpret_cast<const float4 *>(recAk + min(idx, K - 1) * JLD + 4 * h);
            const float4 *rowE = reinterpret_cast<const float4 *>(rec.eig + (size_t)slot * KP + 4 * h);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float4 tv = make_float4(0.f, 0.f, 0.f, 0.f), ta = tv, te = tv;
                if (t < 3 || h == 0) { tv = rowV[2 * t]; ta = rowA[2 * t]; te = rowE[2 * t]; } // (half 1, t = 3: columns 28..31, padding)
                vB[4 * t] = tv.x; vB[4 * t + 1] = tv.y; vB[4 * t + 2] = tv.z; vB[4 * t + 3] = tv.w;
                aK[4 * t] = ta.x; aK[4 * t + 1] = ta.y; aK[4 * t + 2] = ta.z; aK[4 * t + 3] = ta.w;
                dK[4 * t] = te.x; dK[4 * t + 1] = te.y; dK[4 * t + 2] = te.z; dK[4 * t + 3] = te.w;
            }
        }
        const float aux_n = recX[min(lane, P * 6 - 1)], aux_m = recX[P * 6 + min(lane, K - 1)];
        const int p = __builtin_amdgcn_readfirstlane(list[first_item + slot]);
        const int pr = p / W, pc = p - pr * W;
        float wv[G_::CSLICES][WIN_SLICE];
#pragma unroll
        for (int c = 0; c < G_::CSLICES; ++c) win_issue<3, B>(wv[c], colors, c * WIN_SLICE, pr, pc, W, H, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (lane < P * 6) noise[lane] = aux_n;
        if (lane < K) mean[lane] = aux_m;
        const int n = decode_members_win<B>(mask, p, g.words, mem, lane);   // (ends with a barrier: noise and mean are visible too)
#pragma unroll
        for (int c = 0; c < G_::CSLICES; ++c) win_commit<3, B>(cwin, wv[c], c * WIN_SLICE, lane);   // (nothing else uses the LDS windows during the algebra; the loads return with the record's)
        __builtin_amdgcn_sched_barrier(0);

        // N in the accumulator layout: element (r(e, h), idx) of the block-diagonal noise covariance
        float Nop[16];
        {
            const float nz0 = noise[nz_off0], nz1 = noise[nz_off1], nz2 = noise[nz_off2];   // (columns 27..31: u_blk keeps them out)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int t = (e & 3) + 8 * (e >> 2) - u_blk;
                Nop[e] = t == 0 ? nz0 : (t == 1 ? nz1 : (t == 2 ? nz2 : 0.f));
            }
        }
        // ---- Step 1 (:421-436), second half.  clampNegativeEigenValues (:606-630) from the early-stopped solver (pos_part_lds above has the
        // mathematics): P = V f(D) V^T + N, then + V (E o Phi) V^T as two chained products -- U = (E o Phi) V^T (A operand: row idx of the correction
        // matrix at the columns r(s, h), built here from A_k and the eigenvalue estimates; B operand: row idx of V at the same columns), then V U
        // (A operand: the same registers of V; B operand: U as it lies in the accumulators).  Step 15 is column 27 / 31: zero padding everywhere.
        v16f acc;
        {
#pragma unroll
            for (int e = 0; e < 16; ++e) {      // (columns 27..31 of the operands: padding)
                aK[e] = col_ok ? aK[e] : 0.f;     // row idx of the correction matrix E o Phi at the columns r(e, h), as the eigensolver wrote it
                vB[e] = col_ok ? vB[e] : 0.f;
            }
            v16f U;
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[e] = Nop[e]; U[e] = 0.f; }
            if (FOLD_MAIN_TERM) {
                // (round 5) main term and correction in ONE pair of products: M = diag(f(D)) + E o Phi (the correction matrix has a zero diagonal, so f(d_idx)
                // simply takes the diagonal place of row idx), U = M V^T, P = N + V U -- 30 matrix-core steps instead of 45.  P is then symmetric up to
                // round-off only (the separate main term was bitwise symmetric); the sweep inverse and the acceptance test below do not rely on it
                // (measured: the same deviation from the oracle as the three-product form, .notes of round 4)
#pragma unroll
                for (int e = 0; e < 16; ++e) aK[e] += ((e & 3) + 8 * (e >> 2) + 4 * h == idx) ? fmaxf(0.f, dK[e]) : 0.f;
#pragma unroll
                for (int s_ = 0; s_ < 15; ++s_) U = __builtin_amdgcn_mfma_f32_32x32x2f32(aK[s_], vB[s_], U, 0, 0, 0);
            } else {
                // the main term (positive multiples of v v^T on top of N, bitwise symmetric) and U = (E o Phi) V^T: two independent chains,
                // interleaved so that each covers the other's latency
#pragma unroll
                for (int s_ = 0; s_ < 15; ++s_) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vB[s_] * fmaxf(0.f, dK[s_]), vB[s_], acc, 0, 0, 0);
                    U = __builtin_amdgcn_mfma_f32_32x32x2f32(aK[s_], vB[s_], U, 0, 0, 0);
                }
            }
#pragma unroll
            for (int s_ = 0; s_ < 15; ++s_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vB[s_], U[s_], acc, 0, 0, 0);   // + V U
        }
        __builtin_amdgcn_sched_barrier(0);
        float cS[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) cS[e] = recC[min((e & 3) + 8 * (e >> 2) + 4 * h, K - 1) * LD + min(idx, K - 1)];   // (clamped addresses: no branches)
        __builtin_amdgcn_sched_barrier(0);
        if (!sweep_regs(acc, idx, h, min_eig)) { // rare: the spectral branch of inverseSymmetricMatrix (:578-604) -- the item is left to k_bayes27w<2>
            if (lane == 0) { redo[1 + atomicAdd(redo, 1)] = slot; atomicAdd(redo_total, 1); }
            continue;
        }
        __builtin_amdgcn_sched_barrier(0);
        // (the products below run over s = 0..14: step 15 is k = 27 in half 0 and k = 31 in half 1, zero padding in every operand)
        v16f FT;
#pragma unroll
        for (int e = 0; e < 16; ++e) FT[e] = ((e & 3) + 8 * (e >> 2) == u_eye) ? 1.f : 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 15; ++s_) FT = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[s_], Nop[s_], FT, 0, 0, 0);         // F^T = I - C1 N  (acc = -C1)
        __builtin_amdgcn_sched_barrier(0);
        // ---- Step 2 (:438-453)
        v16f GT;
#pragma unroll
        for (int e = 0; e < 16; ++e) GT[e] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 15; ++s_) {
            const bool inside = col_ok && (s_ & 3) + 8 * (s_ >> 2) + 4 * h < K;
            GT = __builtin_amdgcn_mfma_f32_32x32x2f32(inside ? cS[s_] : 0.f, FT[s_], GT, 0, 0, 0);                       // G^T = C F^T
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = Nop[e];
#pragma unroll
        for (int s_ = 0; s_ < 15; ++s_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(FT[s_], GT[s_], acc, 0, 0, 0);          // F C F^T + N
        __builtin_amdgcn_sched_barrier(0);
        if (!sweep_regs(acc, idx, h, min_eig)) {
            if (lane == 0) { redo[1 + atomicAdd(redo, 1)] = slot; atomicAdd(redo_total, 1); }
            continue;
        }
        __builtin_amdgcn_sched_barrier(0);
        // finalDenoisingMatrixMultiplication (:656-670) as xhat = (m - F2 m) + F2 x, F2 = I - N C2 (the same affine map as x - G2 (x - m))
#pragma unroll
        for (int e = 0; e < 16; ++e) FT[e] = ((e & 3) + 8 * (e >> 2) == u_eye) ? 1.f : 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 15; ++s_) FT = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[s_], Nop[s_], FT, 0, 0, 0);         // F2^T = I - C2 N: lane (i, h) holds F2[i][r(s, h)]
        __builtin_amdgcn_sched_barrier(0);
        v16f c0;
        {
            float mreg[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) { mreg[e] = mean[(e & 3) + 8 * (e >> 2) + 4 * h]; c0[e] = 0.f; }
#pragma unroll
            for (int s_ = 0; s_ < 15; ++s_) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(FT[s_], mreg[s_], c0, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 16; ++e) c0[e] = mreg[e] - c0[e];                                                          // m - F2 m
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- the zeroed aggregation window
        for (int e = lane; e < G_::F2_NOISE - G_::F2_ACCS; e += 64) accS[e] = 0.f;   // (sums and counts are contiguous)
        __syncthreads();
        // ---- output pass, 32 members per product: lane (j, h) feeds B = x_j[r(s, h)] and gets back components r(e, h) of member j
        for (int i0 = 0; i0 < n; i0 += 32) {
            const bool valid = i0 + idx < n;
            const int wp = mem[valid ? i0 + idx : 0];
            const int base = (wp - AW - 1) * 3;                 // top-left pixel of the member's patch in the window, floats
            // component r = r0 + 4 h is r + GAP (r / 9) floats into the patch (GAP = 36 for the 15-pixel window): r0 + GAP (r0 / 9) plus 4 h -- or
            // (4 + GAP) h where r0 + 4 is on the next patch line -- i.e. a compile-time offset from one of two per-lane bases, for the operand
            // reads and for the sums alike
            const float *x4 = cwin + base + 4 * h, *x40 = cwin + base + (4 + GAP) * h;
            v16f y = c0;
#pragma unroll
            for (int s_ = 0; s_ < 15; ++s_) {
                const int r0 = (s_ & 3) + 8 * (s_ >> 2);
                float bq;
                if (r0 <= 19) bq = ((r0 + 4) / 9 != r0 / 9 ? x40 : x4)[r0 + GAP * (r0 / 9)];
                else bq = (h ? cwin + base : x4 + r0 + 2 * GAP)[0];   // r0 = 24..26: half 1 is padding (its A operand is zero: any finite cell)
                y = __builtin_amdgcn_mfma_f32_32x32x2f32(FT[s_], bq, y, 0, 0, 0);
            }
            if (valid) {
                float *d4 = accS + base + 4 * h, *d40 = accS + base + (4 + GAP) * h;
                // Plain read-add-write, not ds_add_f32: the LDS float atomic is served one lane at a time on gfx950 (193 cycles of the CU's
                // LDS pipe per wavefront instruction, measured: tools/ubench/lds_rate.hip; a read is 2.5 and a write 4.7), and there is
                // nobody to be atomic against -- one wavefront per workgroup, and within one instruction the 64 addresses are distinct:
                // two members' patches are whole pixels (multiples of 3 floats) apart, the two halves' components 4 or 4 + GAP floats (GAP is a
                // multiple of 3: neither is).
#pragma unroll
                for (int e = 0; e < 12; ++e) {                     // r0 <= 19: both halves inside the 27 components
                    const int r0 = (e & 3) + 8 * (e >> 2);
                    float *q = ((r0 + 4) / 9 != r0 / 9 ? d40 : d4) + r0 + GAP * (r0 / 9);
                    *q = *q + y[e];
                    // (the NEXT component of another lane may be this address: program order must be kept -- the compiler only reasons about
                    // one lane, where the addresses differ; the hardware serves a wavefront's LDS instructions in order)
                    asm volatile("" ::: "memory");
                }
                // the 9 pixels of the patch, counted once each: pixel q is (q % 3) + AW (q / 3) cells from the top-left one; half 0 takes
                // q = 0..4 (cells 0, 1, 2, AW, AW + 1), half 1 q = 5..8 (AW + 2, 2 AW, 2 AW + 1, 2 AW + 2)
                int *c17 = accC + (wp - AW - 1) + (AW + 2) * h, *c29 = accC + (wp - AW - 1) + (2 * AW - 1) * h;
                atomicAdd(c17, 1);
                atomicAdd(c29 + 1, 1);
                atomicAdd(c29 + 2, 1);
                atomicAdd(c17 + AW, 1);
                if (!h) {
#pragma unroll
                    for (int e = 12; e < 15; ++e) { float *q = d4 + 24 + (e - 12) + 2 * GAP; *q = *q + y[e]; asm volatile("" ::: "memory"); } // r0 = 24..26
                    atomicAdd(c17 + AW + 1, 1);
                }
            }
        }
        __syncthreads();
        // aggregateOutputPatches (:672-693): one global atomic per touched value of the window, rows contiguous
        {
            constexpr int row3 = AW * 3;
            const long long base = (long long)p - (long long)(B + 1) * W - (B + 1); // window origin; untouched cells may lie outside the image
            for (int e = lane; e < AW * row3; e += 64) {
                int wy = e / row3, r = e - wy * row3, wx = r / 3;
                if (accC[wy * AW + wx] != 0) unsafeAtomicAdd(sum + (base + (long long)wy * W) * 3 + r, accS[e]);
            }
            for (int e = lane; e < PIX; e += 64) {
                int wy = e / AW, wx = e - wy * AW, c = accC[e];
                if (c != 0) atomicAdd(cnt + (base + (long long)wy * W + wx), c);
            }
        }
        __syncthreads(); // the next item reuses the LDS
    }
}

} // namespace

// LDS of the widest phase (PHASE 2: four matrix buffers)
size_t bcd_bayes27_lds_bytes(int b)
{
    int side = 2 * b + 1;
    return (size_t)(4 * MSZ + 2 * KP + P * 6 + (K + 1) + KP) * sizeof(float) + (((size_t)side * side * sizeof(uint16_t) + 15) & ~(size_t)15);
}

// bytes of HBM one processed pixel needs between the phases (A, V, C, noise + mean, eigenvalues, its entry of the redo list)
size_t bcd_bayes27_record_bytes() { return (size_t)(3 * MSZ + AUX27 + KP) * sizeof(float) + 2 * sizeof(int); } // (+ the redo list: 1 + items ints)

hipError_t bcd_launch_jacobi27_batch(const float *A, int n, int *d_work, int blocks, float *eig, float *V, hipStream_t st, float conv2 = 1e-12f, float *Aout = nullptr,
                                     const int *d_n = nullptr, int first_item = 0);

// Stopping rule of the estimate chain's eigensolver: 2e-9 + first-order correction (production), or -- strict mode, BCD_HIP_STRICT_EIGEN=1 or
// bcd_hip_set_strict_eigensolver -- the fully converged 1e-12 (the looser rule spends accuracy on ill-conditioned low-spp frames, 9.8e-6
// instead of 2.9e-6 on the 4K 8-spp frame; a caller who wants the margin back can have it for ~1.5 % of the step)
static std::atomic<int> g_strict_eigen{ -1 };
void bcd_bayes27_set_strict_eigensolver(int on) { g_strict_eigen.store(on ? 1 : 0); }
static float jacobi_conv2()
{
    int v = g_strict_eigen.load();
    if (v < 0) { const char *e = getenv("BCD_HIP_STRICT_EIGEN"); v = (e && e[0] == '1') ? 1 : 0; g_strict_eigen.store(v); }
    return v ? JACOBI_CONV2_STRICT : JACOBI_CONV2_CORRECTED;
}

// Full estimate of items [first_item, first_item + nb_items) of `list`: three launches; `records` holds nb_items records
// (bcd_bayes27_record_bytes() each), d_work BCD_WORK_INTS zeroed ints (the work queues of the three kernels).
hipError_t bcd_launch_bayes27(const float *colors, const float *pixcov, const uint32_t *mask, const int32_t *list, int first_item, int nb_items,
                              int *d_work, int num_cus, int W, int H, int b, float min_eig, float *records, float *sum,
                              int32_t *cnt, int *d_spectral /* += items whose inverse took the spectral branch (windowed path) */, hipStream_t st,
                              int defer_redo /* != 0: the caller reads *d_spectral after its next synchronisation and calls bcd_launch_bayes27_redo if it is > 0 */,
                              const int *d_nb_items /* optional: the list's length on the device -- the chunk is then min(nb_items, *d_nb_items - first_item) items,
                                                       nb_items being the capacity of `records` (grids are sized for it; wavefronts without an item leave at once) */)
{
    const int *dn = d_nb_items;
    if (nb_items <= 0) return hipSuccess;
    Geom27 g;
    g.W = W; g.H = H; g.b = b; g.side = 2 * b + 1; g.words = (g.side * g.side + 31) / 32; g.maxS = g.side * g.side;
    if (g.words > 32) return hipErrorInvalidValue;
    Records27 rec;
    rec.A = records;
    rec.V = rec.A + (size_t)nb_items * MSZ;
    rec.C = rec.V + (size_t)nb_items * MSZ;
    rec.aux = rec.C + (size_t)nb_items * MSZ;
    rec.eig = rec.aux + (size_t)nb_items * AUX27;
    const size_t lds2 = bcd_bayes27_lds_bytes(b), lds1 = lds2 - 3 * MSZ * sizeof(float);
    const int per_cu1 = (int)std::min<size_t>(20, (size_t)160 * 1024 / lds1), per_cu2 = (int)std::min<size_t>(12, (size_t)160 * 1024 / lds2);
    if (b == WB) {
        // default search radius: the windowed kernels (members read from LDS windows)
        const size_t wl1 = W1L<WB>::BYTES;
        const size_t wl2 = (size_t)W2_MEM * sizeof(float) + WMEM * sizeof(uint16_t);
        const int w_cu1 = (int)std::min<size_t>(20, (size_t)160 * 1024 / wl1), w_cu2 = (int)std::min<size_t>(12, (size_t)160 * 1024 / wl2);
        static const bool lds_algebra = [] { const char *e = getenv("BCD_HIP_FINISH_LDS"); return e && e[0] == '1'; }();
        int *redo = reinterpret_cast<int *>(rec.eig + (size_t)nb_items * KP); // (the register-resident finish; cleared by the prepare kernel)
        hipLaunchKernelGGL(k_bayes27w<1>, dim3(std::min(nb_items, num_cus * w_cu1)), dim3(64), wl1, st, colors, pixcov, mask, list, first_item, nb_items,
                           d_work, g, min_eig, rec, sum, cnt, lds_algebra ? nullptr : redo, dn);
        { hipError_t e = bcd_launch_jacobi27_batch(rec.A, nb_items, d_work + BCD_WORK_QUEUES * BCD_WORK_STRIDE, std::min((nb_items + 1) / 2, num_cus * 12), rec.eig, rec.V, st, jacobi_conv2(), rec.A, dn, first_item); if (e != hipSuccess) return e; }
        if (lds_algebra)
            hipLaunchKernelGGL(k_bayes27w<2>, dim3(std::min(nb_items, num_cus * w_cu2)), dim3(64), wl2, st, colors, pixcov, mask, list, first_item, nb_items,
                               d_work + 2 * BCD_WORK_QUEUES * BCD_WORK_STRIDE, g, min_eig, rec, sum, cnt, nullptr, dn);
        else {
            // the register-resident finish; the items whose sweep inverse fails its checks (rare) come back on a list for the LDS kernel
            const size_t wl3 = WinT<WB>::F2_BYTES;
            hipLaunchKernelGGL(k_finish27w<WB>, dim3(std::min(nb_items, num_cus * 12)), dim3(64), wl3, st, colors, mask, list, first_item, nb_items,
                               d_work + 2 * BCD_WORK_QUEUES * BCD_WORK_STRIDE, g, min_eig, rec, sum, cnt, redo, d_spectral, dn);
            // (normally an empty list: a small grid, so that its 17 KB workgroups do not queue for LDS behind the kernels of the other scales --
            // a full-size launch that only reads "0 items" was seen waiting 0.7 ms for room.  A long list is still processed, by fewer wavefronts.)
            // Round 4: a caller that looks at the counter anyway does not launch it at all on an empty list -- beside the persistent kernels of the other
            // scales even the small launch sat 170-470 us on a coarse scale's stream waiting for LDS, with nothing to do.
            if (!defer_redo)
                hipLaunchKernelGGL(k_bayes27w<2>, dim3(std::min(nb_items, num_cus * 2)), dim3(64), wl2, st, colors, pixcov, mask, list, first_item, nb_items,
                                   d_work + 3 * BCD_WORK_QUEUES * BCD_WORK_STRIDE, g, min_eig, rec, sum, cnt, redo, dn);
        }
        return hipGetLastError();
    }
    constexpr int WB2 = 12; // the large search window of BASELINE configs[4]: register-resident finish on a 27 x 27 pixel window (round 4)
    static const bool finish_lds_b12 = [] { const char *e = getenv("BCD_HIP_FINISH_LDS"); return e && e[0] == '1'; }();
    const bool finish_regs = b == WB2 && !finish_lds_b12;
    int *redo = reinterpret_cast<int *>(rec.eig + (size_t)nb_items * KP);
    // (round 5) b = 12 prepares from LDS windows too: the gather kernel read 105 KB per item (325 members x 81 floats, scattered) where the
    // 27 x 27 window is 26 KB of whole rows; BCD_HIP_PREPARE_GATHER=1 keeps the gather form (A/B; the records are bit-identical)
    static const bool prepare_gather = [] { const char *e = getenv("BCD_HIP_PREPARE_GATHER"); return e && e[0] == '1'; }();
    const bool prepare_win = finish_regs && !prepare_gather;
    if (prepare_win) {
        const size_t wl1 = W1L<WB2>::BYTES;
        const int w_cu1 = (int)std::min<size_t>(20, (size_t)160 * 1024 / wl1);
        hipLaunchKernelGGL((k_bayes27w<1, WB2>), dim3(std::min(nb_items, num_cus * w_cu1)), dim3(64), wl1, st, colors, pixcov, mask, list, first_item, nb_items,
                           d_work, g, min_eig, rec, sum, cnt, (const int *)redo, dn);
    } else
        hipLaunchKernelGGL(k_bayes27<1>, dim3(std::min(nb_items, num_cus * per_cu1)), dim3(64), lds1, st, colors, pixcov, mask, list, first_item, nb_items,
                           d_work, g, min_eig, rec, sum, cnt, (const int *)nullptr, dn);
    { hipError_t e = bcd_launch_jacobi27_batch(rec.A, nb_items, d_work + BCD_WORK_QUEUES * BCD_WORK_STRIDE, std::min((nb_items + 1) / 2, num_cus * 12), rec.eig, rec.V, st, jacobi_conv2(), rec.A, dn, first_item); if (e != hipSuccess) return e; }
    if (finish_regs) {
        // the gather prepare kernel does not know the redo list: its counter is cleared here (one fill per chunk)
        if (!prepare_win) { hipError_t e = hipMemsetAsync(redo, 0, sizeof(int), st); if (e != hipSuccess) return e; }
        const size_t wl3 = WinT<WB2>::F2_BYTES;
        static std::atomic<int> granted[64];
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) dev = -1;
        if (wl3 > 64 * 1024 && (dev < 0 || dev >= 64 || granted[dev].load() == 0)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_finish27w<WB2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl3);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) granted[dev].store(1);
        }
        const int per_cu3 = (int)std::min<size_t>(12, (size_t)160 * 1024 / wl3);
        hipLaunchKernelGGL(k_finish27w<WB2>, dim3(std::min(nb_items, num_cus * per_cu3)), dim3(64), wl3, st, colors, mask, list, first_item, nb_items,
                           d_work + 2 * BCD_WORK_QUEUES * BCD_WORK_STRIDE, g, min_eig, rec, sum, cnt, redo, d_spectral, dn);
        if (!defer_redo)
            hipLaunchKernelGGL(k_bayes27<2>, dim3(std::min(nb_items, num_cus * 2)), dim3(64), lds2, st, colors, pixcov, mask, list, first_item, nb_items,
                               d_work + 3 * BCD_WORK_QUEUES * BCD_WORK_STRIDE, g, min_eig, rec, sum, cnt, (const int *)redo, dn);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_bayes27<2>, dim3(std::min(nb_items, num_cus * per_cu2)), dim3(64), lds2, st, colors, pixcov, mask, list, first_item, nb_items,
                       d_work + 2 * BCD_WORK_QUEUES * BCD_WORK_STRIDE, g, min_eig, rec, sum, cnt, (const int *)nullptr, dn);
    return hipGetLastError();
}

// the deferred redo pass of bcd_launch_bayes27(..., defer_redo = 1): same arguments (the records still hold the chunk)
hipError_t bcd_launch_bayes27_redo(const float *colors, const float *pixcov, const uint32_t *mask, const int32_t *list, int first_item, int nb_items,
                                   int *d_work, int num_cus, int W, int H, int b, float min_eig, float *records, float *sum, int32_t *cnt, hipStream_t st)
{
    if (nb_items <= 0 || (b != WB && b != 12)) return hipSuccess;
    Geom27 g;
    g.W = W; g.H = H; g.b = b; g.side = 2 * b + 1; g.words = (g.side * g.side + 31) / 32; g.maxS = g.side * g.side;
    Records27 rec;
    rec.A = records;
    rec.V = rec.A + (size_t)nb_items * MSZ;
    rec.C = rec.V + (size_t)nb_items * MSZ;
    rec.aux = rec.C + (size_t)nb_items * MSZ;
    rec.eig = rec.aux + (size_t)nb_items * AUX27;
    int *redo = reinterpret_cast<int *>(rec.eig + (size_t)nb_items * KP);
    if (b != WB) { // the large window: the gather kernel walks the list
        hipLaunchKernelGGL(k_bayes27<2>, dim3(std::min(nb_items, num_cus * 2)), dim3(64), bcd_bayes27_lds_bytes(b), st, colors, pixcov, mask, list, first_item, nb_items,
                           d_work + 3 * BCD_WORK_QUEUES * BCD_WORK_STRIDE, g, min_eig, rec, sum, cnt, (const int *)redo);
        return hipGetLastError();
    }
    const size_t wl2 = (size_t)W2_MEM * sizeof(float) + WMEM * sizeof(uint16_t);
    hipLaunchKernelGGL(k_bayes27w<2>, dim3(std::min(nb_items, num_cus * 2)), dim3(64), wl2, st, colors, pixcov, mask, list, first_item, nb_items,
                       d_work + 3 * BCD_WORK_QUEUES * BCD_WORK_STRIDE, g, min_eig, rec, sum, cnt, redo);
    return hipGetLastError();
}

// eigen-decomposition of n symmetric 27 x 27 matrices (28 x 28 zero-padded, row-major): persistent wavefronts, two matrices each
hipError_t bcd_launch_jacobi27_batch(const float *A, int n, int *d_work, int blocks, float *eig, float *V, hipStream_t st, float conv2, float *Aout,
                                     const int *d_n /* optional: the number of matrices is min(n, *d_n - first_item), read on the device */, int first_item)
{
    if (blocks <= 0) return hipSuccess;
    // <padded row placement, DPP-fused row rotation>: measured on 65 536 matrices 31.3 ns per matrix without either, 31.1 with the
    // placement alone (bank conflicts 17 % -> 0.4 % of the LDS cycles), 29.8 with the fused rotation alone (-17 % vector instructions),
    // 29.4 with both
    static const bool pairs = [] { const char *e = getenv("BCD_HIP_JACOBI_PAIRS"); return e && e[0] == '1'; }();
    // (the comparison kernel keeps the plain rule: its residual is then far below what the correction of the finish kernels would notice)
    if (pairs) hipLaunchKernelGGL((k_jacobi27_batch<true, true>), dim3(blocks), dim3(64), 0, st, A, n, d_work, eig, V, 1e-12f, Aout, d_n, first_item);
    else hipLaunchKernelGGL(k_jacobi27_quads, dim3(blocks), dim3(64), 0, st, A, n, d_work, eig, V, conv2, Aout, d_n, first_item);
    return hipGetLastError();
}
