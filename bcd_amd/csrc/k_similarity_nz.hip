// k_similarity_nz.hip -- pair-distance planes over the OWN pixel's non-zero bins (round 5).
//
// The reference's bin rule (src/core/DenoisingUnit.cpp:379-383) skips a bin when b1 + b2 <= 1.  For a bin with b1 == 0 the term is
// therefore (n1 b2)^2 / (n1 n2 b2) = (n1/n2) b2 if b2 > 1 and nothing otherwise -- a closed form that needs no pass over the bin.  With
//     NZ(x) = { k : b1_k > 0 },   S(y) = sum_k [b2_k > 1] b2_k,   C1(y) = #{ k : b2_k > 1 }     (S, C1: one float + one int per pixel)
// the pair sums of k_similarity_fast.hip become
//     T(x,y) = (n2/n1) sum_{k in NZ(x), b1+b2>1} (b1 - (n1/n2) b2)^2 / (b1 + b2)  +  (n1/n2) ( S(y) - sum_{k in NZ(x), b2>1} b2_k )
//     C(x,y) = #{ k in NZ(x) : b1+b2 > 1, b2 <= 1 } + C1(y)                  (b2 > 1 implies b1 + b2 > 1)
// so the loop runs over the own pixel's list only: 18.8 bins instead of 60 on the noisy bench frame, 9.3 on a clean one, 78 % of
// them live (tools/exp_nz_slots.py).  The list must be wave-uniform to pay, hence the lane mapping:
//   A item: ONE own pixel, lane l = displacement 13 dl + dc = l + 1 (the first 64 of the 84 half-plane displacements of b = 6); the
//           bin index and b1 are scalars (b1 by a scalar load), the neighbour bin is one ds_read_b32 at pixel stride 61 (conflict-free:
//           the bank is 29 (13 dl + dc) mod 32 when the line stride is 13 * 61 mod 32).
//   B item: THREE horizontally adjacent own pixels x the remaining 20 displacements (lanes 0..59) + their self pairs (lanes 60..62);
//           the loop runs over the union of the three lists (24.8 bins), b1 is a second ds_read_b32.
// A workgroup (16 wavefronts, one per CU: 157 KB of LDS) stages the histograms of a 24 x 11 tile of own pixels and of the 6 lines
// below / 6 columns either side once, derives S and C1 of every staged pixel, and its wavefronts then draw items from an LDS counter
// (B items first: they are the longer ones).  No rolling window and no barrier between items.  One workgroup per tile (a persistent form
// with the next tile prefetched into registers exists behind BCD_HIP_NZ_PERSIST=1: no faster alone, and it shuts every other kernel out).
//
// T is still the approximate plane of k_similarity_fast.hip (rcp + fma, any summation order, binary16 store), consumed by the same
// mask / verify kernels; C is exact.  The subtraction S - A2 adds an ABSOLUTE error to that file's relative bound.  With u = 2^-24,
// L = |list|, t = d^2 / s:  s' = s (1 + e), |e| <= u;  d' = RN(b1 - rho' b2) with rho' = rho (1 + 3u) (rcp + product; rho' == 1 exactly
// for equal counts), so |d' - d| <= u |d| + 3u rho b2;  the term RN(d'^2) rcp(s') entering the fma carries (1 + 4u);  all terms of A1,
// A2 and S are >= 0 and each is a sequential sum of <= L, <= L and C1 terms.  Together
//     |T' - T*| <= u [ (L + 10) q A1 + (C1 + 4) rho S + (L + 4) rho A2 + T ]  +  [n1 != n2] 6u max(1, rho) B(y),   B(y) = sum_k b2_k
// (the last term bounds q * 6u rho max(1, rho) sum_live b2: b2 |d| / s <= max(1, rho) b2).  The kernel CHECKS, per pair, that this stays
// below the part of the band the relative errors leave free,
//     bound <= NZ_ABS_MARGIN * tau * C        (2^-12: the band 2^-10 = 9.8e-4 minus binary16 4.9e-4 minus the fp32 relative part 2e-5 is > 2^-12)
// and a pair that fails raises range_flag bit 2: the host repeats the scale with the dense kernel.  On the bench frames the bound is
// ~1e-4 against a limit of ~4e-3 at scale 0; coarse scales of noisy frames (512 samples per pixel, S ~ 1500) come close to the limit --
// those have long lists and belong to the dense kernel anyway.  C == 0 implies C1 = 0 and no live bin: S = A1 = A2 = 0, T == 0 exactly.
#include "bcd_common.h"
#include <hip/hip_fp16.h>
#include <atomic>
#include <algorithm>
#include <cstdlib>

namespace {

constexpr int NZ_B = 6, NZ_SIDE = 2 * NZ_B + 1;
constexpr int NZ_TC = 24, NZ_TR = 11;                      // own pixels of a tile (columns a multiple of 3: B items)
constexpr int NZ_NCS = NZ_TC + 2 * NZ_B, NZ_NLS = NZ_TR + NZ_B; // staged columns / lines
constexpr int NZ_THREADS = 1024;
constexpr int NZ_NB = (NZ_TC / 3) * NZ_TR, NZ_NA = NZ_TC * NZ_TR, NZ_ITEMS = NZ_NA + NZ_NB;
constexpr float NZ_BIN_MAX = 1048576.f, NZ_N_MIN = 0.0009765625f, NZ_N_MAX = 65536.f; // the guarded range of k_similarity.hip
constexpr float NZ_ABS_MARGIN = 0.000244140625f; // 2^-12

constexpr int nz_row_stride(int ps)
{
    int rs = NZ_NCS * ps;
    while (((rs - NZ_SIDE * ps) & 31) != 0) ++rs; // line stride = 13 pixel strides (mod 32 banks): lane l of an A item hits bank 29 (l + 1)
    return rs;
}

template <int D> struct NzLayout {
    static_assert(D % 4 == 0, "whole float4 groups");
    static constexpr int PS = D + 1;               // odd pixel stride (dwords)
    static constexpr int RS = nz_row_stride(PS);
    static constexpr int Q = D / 4;
    static constexpr int NPIX = NZ_NLS * NZ_NCS;
    static constexpr int RING = NZ_NLS * RS;       // dwords
    static constexpr int LDS_DWORDS = RING + 4 * NPIX + 4;
    static constexpr int NLOAD = (NPIX * Q + NZ_THREADS - 1) / NZ_THREADS; // float4 per thread of the staged tile
};

__device__ inline int nz_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// One (pixel pair) lane of an item: the loop over the wave-uniform bin list `m`.  b1 comes (OWN = 0) from the own pixel's histogram in
// global memory by a scalar load, (OWN = 1) per lane from LDS (B items: three own pixels), (OWN = 2) from lane k of `own_v` (v_readlane).
// The next bin's operands travel while the current one is evaluated; two copies of the body alternate the registers.
template <int OWN, int BODY, bool RHO1>
__device__ inline void nz_pair_loop(unsigned long long m, const float *__restrict__ own_g, const float *own_l, float own_v, const float *nb, float rho,
                                    float &A1_, float &A2_, int &Cm_)
{
    float A1 = 0.f, A2 = 0.f;
    int Cm = 0;
    auto own = [&](int k) __attribute__((always_inline)) {
        if (OWN == 0) return own_g[k];
        if (OWN == 1) return own_l[k];
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, own_v), k));
    };
    auto eval = [&](float b1, float b2) __attribute__((always_inline)) {
        const float s = b1 + b2;
        if (BODY == 0) {
            if (s > 1.f) { // DenoisingUnit.cpp:379
                asm volatile(""); // keep the branch (exec mask): no if-conversion into selects
                const float d = RHO1 ? b1 - b2 : fmaf(-rho, b2, b1);
                A1 = fmaf(d * d, __builtin_amdgcn_rcpf(s), A1);
                const bool big = b2 > 1.f; // the closed form counted this bin: take it back
                A2 += big ? b2 : 0.f;
                Cm += big ? 0 : 1;
            }
        } else { // straight-line: every lane evaluates, selects keep the dead ones out (s <= 1: r = 0; b2 > 1 implies s > 1)
            const bool live = s > 1.f, big = b2 > 1.f;
            const float r = live ? __builtin_amdgcn_rcpf(s) : 0.f;
            const float d = RHO1 ? b1 - b2 : fmaf(-rho, b2, b1);
            A1 = fmaf(d * d, r, A1);
            A2 += big ? b2 : 0.f;
            Cm += (live && !big) ? 1 : 0;
        }
    };
    if (m != 0ull) {
        int k = __builtin_ctzll(m);
        m &= m - 1;
        float b1 = own(k), b2 = nb[k];
#pragma unroll 2
        while (m != 0ull) {
            k = __builtin_ctzll(m);
            m &= m - 1;
            const float b1n = own(k), b2n = nb[k];
            eval(b1, b2);
            b1 = b1n; b2 = b2n;
        }
        eval(b1, b2);
    }
    A1_ = A1; A2_ = A2; Cm_ = Cm;
}

template <int D, bool READLANE, int BODY, bool PROF, bool PERSIST>
__global__ __launch_bounds__(NZ_THREADS, 4) void k_pairdist_nz(const float *__restrict__ hist, const float *__restrict__ ns, int W, int H,
                                                               __half *__restrict__ T, uint8_t *__restrict__ Cn, long long t_ps, long long t_ds,
                                                               int *range_flag, float tau, int tiles_x, int ntiles, unsigned long long *prof)
{
    using L = NzLayout<D>;
    constexpr int PS = L::PS, RS = L::RS, Q = L::Q, NPIX = L::NPIX;
    extern __shared__ float nz_lds[];
    float *ring = nz_lds;
    float *s_n = ring + L::RING;
    float *s_S = s_n + NPIX;
    float *s_B = s_S + NPIX;
    int *s_C1 = reinterpret_cast<int *>(s_B + NPIX);
    int *s_counter = s_C1 + NPIX;

    const int tid = threadIdx.x, lane = tid & 63;
    // (PERSIST only) the tiles are dealt in row-major order, a contiguous share per XCD (workgroup w runs on XCD w & 7), and the workgroups of
    // an XCD take consecutive tiles of that share
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (int)gridDim.x >> 3; // (gridDim.x is a multiple of 8)
    const int tq = ntiles / 8, trem = ntiles - 8 * tq;
    const int share_begin = xcd * tq + min(xcd, trem), share_end = share_begin + tq + (xcd < trem ? 1 : 0);

    // per-thread share of a tile's staged histograms: NLOAD 16-byte groups (LDS offsets do not depend on the tile)
    float4 v[L::NLOAD];
    float nv = 1.f;
    bool nv_in = false; // the thread's staged pixel lies inside the image (an explicit flag: ANY count value of an in-image pixel goes through the range check)
    auto prefetch = [&](int tile) __attribute__((always_inline)) {
        const int c0 = (tile % tiles_x) * NZ_TC, r0 = (tile / tiles_x) * NZ_TR;
#pragma unroll
        for (int u = 0; u < L::NLOAD; ++u) {
            const int j = tid + u * NZ_THREADS;
            const int px = j / Q, q = j - px * Q, line = px / NZ_NCS, col = px - line * NZ_NCS;
            const int gr = r0 + line, gc = c0 - NZ_B + col;
            const bool in = j < NPIX * Q && gr < H && gc >= 0 && gc < W;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in) v[u] = reinterpret_cast<const float4 *>(hist)[((size_t)gr * W + gc) * Q + q];
        }
        nv = 1.f;
        nv_in = false;
        if (tid < NPIX) {
            const int line = tid / NZ_NCS, col = tid - line * NZ_NCS, gr = r0 + line, gc = c0 - NZ_B + col;
            if (gr < H && gc >= 0 && gc < W) { nv = ns[(size_t)gr * W + gc]; nv_in = true; }
        }
    };
    long long pc[5] = { 0, 0, 0, 0, 0 }, slots = 0;
    bool imprecise = false, tile_uni = false;
    // PERSIST: one workgroup per CU walks its share of the tiles, the next tile's histograms in flight during the current one (127 registers, every
    // wavefront slot of the chip taken for the whole launch: nothing else runs beside it).  !PERSIST (the production form, round 5): one workgroup per
    // tile, no prefetch (68 registers) -- the same speed alone (staging is 4 % either way once several workgroups queue per CU), and the kernels of the
    // other scales get CUs as tiles retire: measured at 24 spp, the step went from 5.6 to the figure in DESIGN 3.
    // (!PERSIST: tiles in launch order -- a contiguous share per XCD, tried, was 5 % slower: 2.03 against 1.93 ms at 1080p)
    int tile = PERSIST ? share_begin + slot : (int)blockIdx.x;
    const int tile_end = PERSIST ? share_end : (tile < ntiles ? tile + 1 : tile), tile_step = PERSIST ? per_xcd : 1;
    if (tile < tile_end) prefetch(tile);
    for (; tile < tile_end; tile += tile_step) {
    const int c0 = (tile % tiles_x) * NZ_TC, r0 = (tile / tiles_x) * NZ_TR;
    const long long t0 = PROF ? (long long)__builtin_readcyclecounter() : 0;
    // ---- the prefetched tile -> LDS: lines r0 .. r0 + TR + 5, columns c0 - 6 .. c0 + TC + 5 (zeros outside the image) ----
    {
        bool bad = false;
#pragma unroll
        for (int u = 0; u < L::NLOAD; ++u) {
            const int j = tid + u * NZ_THREADS;
            const int px = j / Q, q = j - px * Q, line = px / NZ_NCS, col = px - line * NZ_NCS;
            if (j < NPIX * Q) {
                float *p = ring + line * RS + col * PS + 4 * q;
                p[0] = v[u].x; p[1] = v[u].y; p[2] = v[u].z; p[3] = v[u].w;
            }
            const float lo = fminf(fminf(v[u].x, v[u].y), fminf(v[u].z, v[u].w)), hi = fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w));
            // (NaN: fminf / fmaxf drop it, so test the values themselves too)
            bad = bad || !(lo >= 0.f && hi <= NZ_BIN_MAX) || !(v[u].x == v[u].x && v[u].y == v[u].y && v[u].z == v[u].z && v[u].w == v[u].w);
        }
        const bool in_image = nv_in; // (a NaN or negative count is inside, and bad)
        if (tid < NPIX) { bad = bad || (in_image && !(nv >= NZ_N_MIN && nv <= NZ_N_MAX)); s_n[tid] = in_image ? nv : 1.f; }
        if (tid == 0) *s_counter = 0;
        if (__syncthreads_or(bad) && tid == 0) atomicOr(range_flag, 1);
        // one sample count on the whole staged tile (pixels outside the image have no pairs that are kept): rho = q = 1, no loads
        tile_uni = __syncthreads_and(tid >= NPIX || !in_image || nv == s_n[NZ_B]) != 0; // (s_n[B]: the tile's first own pixel, always inside)
    }
    const long long t1 = PROF ? (long long)__builtin_readcyclecounter() : 0;
    // ---- S(y), C1(y) of every staged pixel ----
    if (tid < NPIX) {
        const int line = tid / NZ_NCS, col = tid - line * NZ_NCS;
        const float *p = ring + line * RS + col * PS;
        float S = 0.f, Bs = 0.f;
        int C1 = 0;
#pragma unroll 12
        for (int k = 0; k < D; ++k) {
            const float v = p[k];
            Bs += v;
            if (v > 1.f) { S += v; ++C1; }
        }
        s_S[tid] = S;
        s_B[tid] = Bs;
        s_C1[tid] = C1;
    }
    __syncthreads();
    // the next tile's histograms travel while this one is evaluated
    if (PERSIST && tile + per_xcd < share_end) prefetch(tile + per_xcd);
    const long long t2 = PROF ? (long long)__builtin_readcyclecounter() : 0;
    // ---- per-lane displacement tables ----
    // A: lane l <-> 13 dl + dc = l + 1
    const int ipA = lane + 1, dlA = (ipA + NZ_B) / NZ_SIDE, dcA = ipA - NZ_SIDE * dlA;
    const int offA = dlA * RS + dcA * PS, spA = dlA * NZ_NCS + dcA;
    // B: lanes 0..59 = own pixel j = l / 20, displacement 65 + l % 20; lanes 60..62 = self pair of own pixel l - 60; lane 63 idle
    const int jB = lane < 60 ? lane / 20 : min(lane - 60, 2);
    const int ipB = lane < 60 ? 65 + (lane - 20 * jB) : 0;
    const int dlB = (ipB + NZ_B) / NZ_SIDE, dcB = ipB - NZ_SIDE * dlB; // (ipB == 0: dl = dc = 0)
    const int ownB = jB * PS, offB = ownB + dlB * RS + dcB * PS, spB = jB + dlB * NZ_NCS + dcB;
    const float margin = NZ_ABS_MARGIN * tau * 16777216.f; // (the bound is formed in units of u = 2^-24)
    // where the lane's entry goes: element = base + pixel * mult (plane elements fit 32 bits: checked by the launcher).  t_ds != 0: the general form
    // (displacement i at i * t_ds + pixel * t_ps).  t_ds == 0: the SPLIT pixel-major layout the production path uses (round 6) -- two records per pixel, each
    // aligned to its size: record A = 64 entries (displacements 0..63) at pixel * 64, record B = 32 entries at W H 64 + pixel * 32 holding displacements
    // 65..84 at positions 0..19 and displacement 64 at position 20.  An A item's store is then one aligned 128-byte line (T) / 64 bytes (counts), and the
    // mask kernel's two passes read disjoint lines (with the 85-entry records of round 5 both passes fetched nearly every line of the planes).
    const bool split = t_ds == 0;
    const unsigned int npix_all = (unsigned int)W * (unsigned int)H;
    const unsigned int baseA = split ? (lane < 63 ? (unsigned int)ipA : npix_all * 64u + 20u) : (unsigned int)ipA * (unsigned int)t_ds;
    const unsigned int multA = split ? (lane < 63 ? 64u : 32u) : (unsigned int)t_ps;
    const unsigned int baseB = split ? (lane < 60 ? npix_all * 64u + (unsigned int)(ipB - 65) : 0u) : (unsigned int)ipB * (unsigned int)t_ds;
    const unsigned int multB = split ? (lane < 60 ? 32u : 64u) : (unsigned int)t_ps;

    auto grab = [&]() __attribute__((always_inline)) {
        int v = 0;
        if (lane == 0) v = atomicAdd(s_counter, 1);
        return nz_uniform(v);
    };
    int it = grab();
    while (it < NZ_ITEMS) {
        const int next = grab();
        if (it >= NZ_NB) {
            // ---- A item: one own pixel, displacements 1..64 ----
            const int a = it - NZ_NB, orow = a / NZ_TC, ocol = a - orow * NZ_TC;
            const int gr = r0 + orow, gc = c0 + ocol;
            if (gr < H && gc < W) {
                const int o = orow * RS + (ocol + NZ_B) * PS, spo = orow * NZ_NCS + ocol + NZ_B;
                const float hv = lane < D ? ring[o + lane] : 0.f;
                const float Sy = s_S[spo + spA];
                const int C1y = s_C1[spo + spA];
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hv > 0.f);
                const unsigned int pix = (unsigned int)(gr * W + gc);
                float A1, A2, rho = 1.f, qq = 1.f, extra = 0.f;
                int Cm;
                if (PROF) slots += __builtin_popcountll(m);
                if (tile_uni)
                    nz_pair_loop<READLANE ? 2 : 0, BODY, true>(m, hist + (size_t)pix * D, nullptr, hv, ring + o + offA, 1.f, A1, A2, Cm);
                else {
                    const float n1 = s_n[spo], n2 = s_n[spo + spA];
                    if (n1 != n2) { rho = n1 * __builtin_amdgcn_rcpf(n2); qq = n2 * __builtin_amdgcn_rcpf(n1); extra = 6.f * fmaxf(1.f, rho) * s_B[spo + spA]; }
                    nz_pair_loop<READLANE ? 2 : 0, BODY, false>(m, hist + (size_t)pix * D, nullptr, hv, ring + o + offA, rho, A1, A2, Cm);
                }
                const int C = Cm + C1y, Ln = __builtin_popcountll(m);
                const float Tv = fmaf(qq, A1, rho * (Sy - A2));
                const int nc = gc + dcA, nr = gr + dlA;
                if (nc >= 0 && nc < W && nr < H) {
                    // (the bound of the header with the larger of its two factors on every term; the count-ratio term only where a bin was live)
                    const float bound = fmaf((float)(max(Ln, C1y) + 10), fmaf(qq, A1, rho * (Sy + A2)), Tv + ((Cm > 0 || A2 > 0.f) ? extra : 0.f));
                    imprecise = imprecise || bound > margin * (float)C;
                    const unsigned int e = baseA + pix * multA;
                    T[e] = __float2half_rn(fmaxf(Tv, 0.f));
                    Cn[e] = (uint8_t)C;
                }
            }
        } else {
            // ---- B item: three own pixels, displacements 65..84 and the self pairs ----
            const int orow = it / (NZ_TC / 3), ocol = 3 * (it - orow * (NZ_TC / 3));
            const int gr = r0 + orow, gc0 = c0 + ocol;
            if (gr < H && gc0 < W) {
                const int o = orow * RS + (ocol + NZ_B) * PS, spo = orow * NZ_NCS + ocol + NZ_B;
                const float h0 = lane < D ? ring[o + lane] : 0.f, h1 = lane < D ? ring[o + PS + lane] : 0.f, h2 = lane < D ? ring[o + 2 * PS + lane] : 0.f;
                const float Sy = s_S[spo + spB];
                const int C1y = s_C1[spo + spB];
                const unsigned long long m = __builtin_amdgcn_ballot_w64(h0 > 0.f || h1 > 0.f || h2 > 0.f);
                const int gc = gc0 + jB;
                const unsigned int pix = (unsigned int)(gr * W + gc);
                float A1, A2, rho = 1.f, qq = 1.f, extra = 0.f;
                int Cm;
                if (PROF) slots += __builtin_popcountll(m);
                if (tile_uni)
                    nz_pair_loop<1, BODY, true>(m, nullptr, ring + o + ownB, 0.f, ring + o + offB, 1.f, A1, A2, Cm);
                else {
                    const float n1 = s_n[spo + jB], n2 = s_n[spo + spB];
                    if (n1 != n2) { rho = n1 * __builtin_amdgcn_rcpf(n2); qq = n2 * __builtin_amdgcn_rcpf(n1); extra = 6.f * fmaxf(1.f, rho) * s_B[spo + spB]; }
                    nz_pair_loop<1, BODY, false>(m, nullptr, ring + o + ownB, 0.f, ring + o + offB, rho, A1, A2, Cm);
                }
                const int C = Cm + C1y, Ln = __builtin_popcountll(m);
                const float Tv = fmaf(qq, A1, rho * (Sy - A2));
                const int nc = gc + dcB, nr = gr + dlB;
                if (lane < 63 && gc < W && nc >= 0 && nc < W && nr < H) {
                    // (the bound of the header with the larger of its two factors on every term; the count-ratio term only where a bin was live)
                    const float bound = fmaf((float)(max(Ln, C1y) + 10), fmaf(qq, A1, rho * (Sy + A2)), Tv + ((Cm > 0 || A2 > 0.f) ? extra : 0.f));
                    imprecise = imprecise || bound > margin * (float)C;
                    const unsigned int e = baseB + pix * multB;
                    T[e] = __float2half_rn(fmaxf(Tv, 0.f));
                    Cn[e] = (uint8_t)C;
                }
            }
        }
        it = next;
    }
    if (PROF) { // (measurement only) cycles: staging, S pass, wavefronts inside the item loop, item phase of the workgroup x wavefronts; bin slots
        const long long t3 = (long long)__builtin_readcyclecounter();
        __syncthreads();
        const long long t4 = (long long)__builtin_readcyclecounter();
        pc[0] += t1 - t0; pc[1] += t2 - t1; pc[2] += t3 - t2; pc[3] += (t4 - t2) * (NZ_THREADS / 64); pc[4] += t4 - t0;
    } else
        __syncthreads(); // every wavefront is done with the tile before the next one overwrites it
    }
    if (__builtin_amdgcn_ballot_w64(imprecise) != 0ull && lane == 0) atomicOr(range_flag, 4);
    if (PROF) {
        if (lane == 0) { atomicAdd(prof + 2, (unsigned long long)pc[2]); atomicAdd(prof + 5, (unsigned long long)slots); }
        if (tid == 0) {
            atomicAdd(prof + 0, (unsigned long long)pc[0]); atomicAdd(prof + 1, (unsigned long long)pc[1]);
            atomicAdd(prof + 3, (unsigned long long)pc[3]); atomicAdd(prof + 4, (unsigned long long)pc[4]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Forward similarity bits from the SPLIT PIXEL-major planes k_pairdist_nz writes with (t_ps, t_ds) = (0, 0): per pixel a record A of 64 entries
// (displacements 0..63, at pixel * 64) and a record B of 32 entries (at W H 64 + pixel * 32: displacements 65..84 at positions 0..19, displacement 64 at
// position 20).  Lanes = displacements, so the 3 x 3 patch sum of a displacement stays in one lane: a wavefront walks along a
// strip of FWD_PM_RB lines, loads per column the FWD_PM_RB + 2 plane lines of its displacement (one aligned 128-byte line + 64 bytes per pixel in pass 0),
// keeps the last columns in registers, and decides a column of FWD_PM_RB pixels per step -- any summation order: these are the
// approximate planes, pairs inside tau (1 +- delta) go to the borderline list exactly like in k_fwd_masks_w1 (k_similarity.hip).
//   CHUNK 0: record A of one strip (words 0 and 1 of the pixel's three forward words come straight from the ballot);
//   CHUNK 1: record B of three strips (lanes 21 j .. 21 j + 20 = strip 3 blockIdx.x + j; word 2).
// ---------------------------------------------------------------------------------------------------
constexpr int FWD_PM_RB = 4;

// Round 6: a workgroup is FWD_PM_WV wavefronts on vertically adjacent strips of lines, marching along the same columns: the two plane lines a strip shares
// with the next one are fetched by both within the same few hundred cycles and come out of the CU's L1 the second time (as single-wavefront workgroups the
// two readers sat on different XCDs and both went to HBM: 6 lines read per 4 decided), and the column strips are 32 wide (4 warm-up columns per strip).
constexpr int FWD_PM_WV = 4;
#define FWD_PM_REC (CHUNK == 0 ? 64u : 32u)
template <int CHUNK>
__global__ __launch_bounds__(64 * FWD_PM_WV) void k_fwd_masks_pm( /* FWD_PM_REC: entries per pixel of the record this pass reads */const __half *__restrict__ T, const uint8_t *__restrict__ Cn, int W, int H, float tau_lo,
                                                     BcdBorderline bl, uint32_t *__restrict__ fwd, int strip_cols, int nstrips)
{
    const int lane = threadIdx.x & 63;
    const int sub = CHUNK == 0 ? 0 : lane / 21;
    const int pos = CHUNK == 0 ? lane : lane - 21 * sub;                     // position inside the pixel's record A (64 entries) / record B (32 entries)
    const int idx = CHUNK == 0 ? lane : (pos < 20 ? 65 + pos : 64);          // the displacement it holds (k_pairdist_nz, split layout)
    const int strip = CHUNK == 0 ? (int)blockIdx.x : 3 * (int)blockIdx.x + sub;
    const bool lane_on = (CHUNK == 0 || lane < 63) && strip < nstrips;
    const int dl = (idx + NZ_B) / NZ_SIDE, dc = idx - NZ_SIDE * dl; // (idx 0: the self pair)
    const int rb = (blockIdx.y * FWD_PM_WV + (threadIdx.x >> 6)) * FWD_PM_RB;
    if (rb >= H) return; // (no barriers in this kernel)
    const int cs = min(strip, nstrips - 1) * strip_cols, ce = min(W, cs + strip_cols);
    const int ncols = strip_cols; // (uniform loop bound; columns beyond a strip's end are gated by c < ce)
    unsigned int line_off[FWD_PM_RB + 2];
#pragma unroll
    for (int i = 0; i < FWD_PM_RB + 2; ++i) line_off[i] = (CHUNK == 0 ? 0u : (unsigned int)(W * H) * 64u) + (unsigned int)(min(max(rb - 1 + i, 0), H - 1) * W) * FWD_PM_REC + pos;
    // The plane values of four consecutive columns live in registers: a step decides column c from columns c - 1 | c | c + 1, column c + 2 is in flight, and
    // the loads of column c + 3 are issued into the registers of column c - 1 as soon as its sums are taken.  The four register sets ROTATE BY NAME (the
    // loop body is four steps, round 6): until then every step ended with `t0 = t1; t1 = t2; ... t3 = t4`, and a move out of a register a load is still
    // writing waits for that load -- each step waited for the loads it had just issued, 1.5 us of exposed memory latency per column.
    float tA[FWD_PM_RB + 2], tB[FWD_PM_RB + 2], tC[FWD_PM_RB + 2], tD[FWD_PM_RB + 2];
    int nA[FWD_PM_RB + 2], nB[FWD_PM_RB + 2], nC[FWD_PM_RB + 2], nD[FWD_PM_RB + 2];
    auto load = [&](int c, float (&t)[FWD_PM_RB + 2], int (&n)[FWD_PM_RB + 2]) __attribute__((always_inline)) {
        const unsigned int co = (unsigned int)min(max(c, 0), W - 1) * FWD_PM_REC;
#pragma unroll
        for (int i = 0; i < FWD_PM_RB + 2; ++i) { t[i] = __half2float(T[line_off[i] + co]); n[i] = Cn[line_off[i] + co]; }
    };
    // one step: decide column c from (p | q | r) = columns (c - 1 | c | c + 1); afterwards p holds column c + 3
    auto step = [&](int c, float (&tp)[FWD_PM_RB + 2], int (&np)[FWD_PM_RB + 2], const float (&tq)[FWD_PM_RB + 2], const int (&nq)[FWD_PM_RB + 2],
                    const float (&tr)[FWD_PM_RB + 2], const int (&nr)[FWD_PM_RB + 2]) __attribute__((always_inline)) {
        float h[FWD_PM_RB + 2];
        int hn[FWD_PM_RB + 2];
#pragma unroll
        for (int i = 0; i < FWD_PM_RB + 2; ++i) { h[i] = (tp[i] + tq[i]) + tr[i]; hn[i] = np[i] + nq[i] + nr[i]; }
        load(c + 3, tp, np);
        const int qc = c + dc;
        const bool cols_ok = lane_on && c < ce && c >= 1 && c <= W - 2 && qc >= 1 && qc <= W - 2;
#pragma unroll
        for (int i = 0; i < FWD_PM_RB; ++i) {
            const int r = rb + i;
            const float sT = (h[i] + h[i + 1]) + h[i + 2];
            const int n = hn[i] + hn[i + 1] + hn[i + 2];
            const float fn = (float)n;
            const bool ok = cols_ok && r >= 1 && r <= H - 2 && r + dl <= H - 2 && n > 0;
            const bool sure = ok && sT <= tau_lo * fn; // (no division: see k_fwd_masks_w1; 0 / 0 in the reference is never similar)
            const bool border = ok && !sure && sT <= bl.tau_hi * fn;
            const unsigned long long ms = __builtin_amdgcn_ballot_w64(sure), mb = __builtin_amdgcn_ballot_w64(border);
            if (r < H) {
                if (CHUNK == 0) {
                    if (lane < 2 && c < ce) fwd[(size_t)(r * W + c) * 3 + lane] = lane ? (uint32_t)(ms >> 32) : (uint32_t)ms;
                } else {
                    // word 2 of the pixel: bit 0 = displacement 64 (position 20 of the record), bits 1..20 = displacements 65..84 (positions 0..19)
                    const uint32_t bits = (uint32_t)(ms >> (21 * sub)) & 0x1fffffu;
                    if (lane < 63 && lane == 21 * sub && strip < nstrips && c < ce) fwd[(size_t)(r * W + c) * 3 + 2] = ((bits & 0xfffffu) << 1) | (bits >> 20);
                }
            }
            if (mb != 0ull) { // (rare) append the borderline pairs of this line: one atomic per wavefront
                const int total = __builtin_popcountll(mb);
                int base = 0;
                if (lane == 0) base = atomicAdd(bl.counter, total);
                base = __builtin_amdgcn_readfirstlane(base);
                if (border) {
                    const int slot = base + __builtin_popcountll(mb & ((1ull << lane) - 1ull));
                    if (slot < bl.capacity) bl.list[slot] = make_uint2((uint32_t)(r * W + c), (uint32_t)idx);
                }
            }
        }
    };
    load(cs - 1, tA, nA);
    load(cs, tB, nB);
    load(cs + 1, tC, nC);
    load(cs + 2, tD, nD);
    for (int c = cs; c < cs + ncols; c += 4) { // (ncols is a multiple of four; columns beyond a strip's end are gated by c < ce)
        step(c, tA, nA, tB, nB, tC, nC);
        step(c + 1, tB, nB, tC, nC, tD, nD);
        step(c + 2, tC, nC, tD, nD, tA, nA);
        step(c + 3, tD, nD, tA, nA, tB, nB);
    }
}

} // namespace

int bcd_pairdist_nz_supported(int D, int b) { return b == NZ_B && (D == 60 || D == 36 || D == 24); }

// T / Cn element (pixel, displacement index i) lives at i * t_ds + pixel * t_ps: (1, W*H) is the plane-major layout of k_pairdist_rw,
// (stride >= 85, 1) a pixel-major one; (0, 0) selects the split pixel-major layout k_fwd_masks_pm reads (96 entries per pixel: see the kernel).  variant: bit 0 = own bins by v_readlane instead of scalar loads, bit 1 = straight-line bin body
hipError_t bcd_launch_pairdist_nz(const float *hist, const float *ns, int W, int H, int D, int b, void *T, uint8_t *Cn, long long t_ps, long long t_ds,
                                  int *d_range_flag, float tau, int variant, hipStream_t st, unsigned long long *prof = nullptr)
{
    if (!bcd_pairdist_nz_supported(D, b) || (long long)W * H * 96 >= (1ll << 31)) return hipErrorInvalidValue;
    const int tx = (W + NZ_TC - 1) / NZ_TC, ty = (H + NZ_TR - 1) / NZ_TR;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    // one persistent workgroup per CU (157 KB of LDS each), a multiple of the 8 XCDs
    static std::atomic<int> cus_of[64];
    int cus = (dev >= 0 && dev < 64) ? cus_of[dev].load() : 0;
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev < 0 ? 0 : dev) != hipSuccess || cus <= 0) cus = 256;
        if (dev >= 0 && dev < 64) cus_of[dev].store(cus);
    }
    const int nwg = std::max(8, std::min((cus / 8) * 8, ((tx * ty + 7) / 8) * 8));
    const bool persist = (variant & 4) != 0; // (variant bit 2: the persistent form, for A/B measurements)
    variant &= 3;
#define BCD_NZ_LAUNCH3(DD, RL, BD, PF, PS)                                                                                             \
    {                                                                                                                      \
        const size_t lds = (size_t)NzLayout<DD>::LDS_DWORDS * 4;                                                           \
        static std::atomic<int> granted[64];                                                                               \
        if (dev < 0 || dev >= 64 || granted[dev].load() == 0) {                                                            \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pairdist_nz<DD, RL, BD, PF, PS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                                 \
            if (dev >= 0 && dev < 64) granted[dev].store(1);                                                               \
        }                                                                                                                  \
        hipLaunchKernelGGL((k_pairdist_nz<DD, RL, BD, PF, PS>), dim3(PS ? nwg : tx * ty), dim3(NZ_THREADS), lds, st, hist, ns, W, H, static_cast<__half *>(T), Cn, t_ps, t_ds, d_range_flag, tau, tx, tx * ty, prof); \
        return hipGetLastError();                                                                                          \
    }
#define BCD_NZ_LAUNCH2(DD, RL, BD, PF) { if (persist) BCD_NZ_LAUNCH3(DD, RL, BD, PF, true) else BCD_NZ_LAUNCH3(DD, RL, BD, PF, false) }
#define BCD_NZ_LAUNCH1(DD, RL, BD) { if (prof) BCD_NZ_LAUNCH2(DD, RL, BD, true) else BCD_NZ_LAUNCH2(DD, RL, BD, false) }
#define BCD_NZ_LAUNCH(DD) case DD: if (variant == 1) BCD_NZ_LAUNCH1(DD, true, 0) else if (variant == 2) BCD_NZ_LAUNCH1(DD, false, 1) else if (variant == 3) BCD_NZ_LAUNCH1(DD, true, 1) else BCD_NZ_LAUNCH1(DD, false, 0)
    switch (D) {
    BCD_NZ_LAUNCH(60)
    BCD_NZ_LAUNCH(36)
    BCD_NZ_LAUNCH(24)
    default: break;
    }
#undef BCD_NZ_LAUNCH
#undef BCD_NZ_LAUNCH1
#undef BCD_NZ_LAUNCH2
#undef BCD_NZ_LAUNCH3
    return hipErrorInvalidValue;
}

// forward bits from the split pixel-major planes of bcd_launch_pairdist_nz (t_ps = t_ds = 0); ap: the borderline list (tau_hi is set here)
hipError_t bcd_launch_fwd_masks_pm(const void *T, const uint8_t *Cn, int W, int H, float tau, uint32_t *fwd, const BcdBorderline *ap, hipStream_t st)
{
    if (!ap) return hipErrorInvalidValue;
    BcdBorderline bl = *ap;
    bl.tau_hi = tau * (1.f + BCD_APPROX_DELTA);
    const float tau_lo = tau * (1.f - BCD_APPROX_DELTA);
    const int strip_cols = 32, nstrips = (W + strip_cols - 1) / strip_cols, nrb = ((H + FWD_PM_RB - 1) / FWD_PM_RB + FWD_PM_WV - 1) / FWD_PM_WV;
    hipLaunchKernelGGL(k_fwd_masks_pm<0>, dim3(nstrips, nrb), dim3(64 * FWD_PM_WV), 0, st, static_cast<const __half *>(T), Cn, W, H, tau_lo, bl, fwd, strip_cols, nstrips);
    hipLaunchKernelGGL(k_fwd_masks_pm<1>, dim3((nstrips + 2) / 3, nrb), dim3(64 * FWD_PM_WV), 0, st, static_cast<const __half *>(T), Cn, W, H, tau_lo, bl, fwd, strip_cols, nstrips);
    return hipGetLastError();
}
