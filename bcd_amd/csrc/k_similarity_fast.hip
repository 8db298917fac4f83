// k_similarity_fast.hip -- similar-patch selection, fast path: APPROXIMATE pair-distance planes with a rigorous error
// bound, decided exactly only at the threshold.
//
// The planes T_delta(x), C_delta(x) of k_similarity.hip are consumed by exactly one test, d(p, p+delta) <= tau
// (src/core/DenoisingUnit.cpp:209).  Here T is evaluated with one v_rcp_f32 + fma per bin (error bound below) and in any
// summation order, and stored in binary16; C (integer bin counts, the `b1 + b2 <= 1` skip test of DenoisingUnit.cpp:379) stays exact.  The mask
// kernel then decides every pair whose approximate distance lies outside tau (1 +- BCD_APPROX_DELTA) and appends the few
// borderline pairs to a list; k_verify_pairs re-evaluates those with the reference's exact operation sequence
// (sequential bins, IEEE division) and sets their bits.  The masks are therefore bit-identical to the exact path.
//
// Error bound (fp32, u = 2^-24; inputs inside the guarded range of k_similarity.hip, so nothing overflows or is
// subnormal on an evaluated bin).  Both paths compute the SAME diff and den with the same operations: den = RN(RN(n1 n2)
// RN(b1+b2)), diff = RN(RN(n2 b1) - RN(n1 b2)) (uniform power-of-two counts: RN(b1-b2), RN(b1+b2), see k_similarity.hip).
// With t* = RN(diff^2) / den (exact quotient):  reference term RN(t*) = t* (1 + e), |e| <= u;  here the product
// RN(diff^2) * rcp(den) enters a fused multiply-add unrounded, and v_rcp_f32 is accurate to 1 ulp: t' = t* (1 + e'),
// |e'| <= 2u.  All terms are >= 0, so every addition adds at most u of relative error to what it sums.  A term of the
// reference's patch distance goes through <= (D - 1) + 8 additions and one division: d_ref = d* (1 + a), |a| <= (D + 9) u;
// here through <= D + 8 fma / additions and one division: d' = d* (1 + a'), |a'| <= (D + 11) u.  Hence
//     |d' / d_ref - 1| <= (2 D + 20) u + O(u^2)  =  8.3e-6 for D = 60
// in the worst case (every round-off pointing the same way).  The plane itself is then stored in binary16 (round to nearest: one
// more factor 1 +- 2^-11 per entry, and entries are >= 0, so the same bound carries to their sums): together |d' / d_ref - 1| <
// 5.0e-4, and BCD_APPROX_DELTA = 2^-10 = 9.8e-4 is twice that.  The plane costs 3 bytes per entry instead of 5 (count byte
// included) and the mask kernel, which streams it, is memory bound.  Thresholds outside [2^-6, 64] go to the exact kernels
// (binary16 subnormals below, overflow to +inf above: bcd_common.h).  Measured maximum over whole frames: 2.4e-4
// (bcd_hip_selftest_approx_distance; the GPU tests assert < delta / 2), a few thousand re-evaluated pairs per 1080p scale.
//
// Cost model that shaped the kernel (measured on MI355X): a SIMD issues about one instruction of ANY kind per 2-3 cycles
// -- VALU, scalar, branch and LDS instructions all compete for it -- so the aim is the smallest instruction count per
// (pixel, displacement, bin), not the cheapest arithmetic alone:
//   * a divergent `if (b1 + b2 > 1)` around the 5 arithmetic instructions of a bin (execz branch) under a wave-uniform
//     test of its group of 4 bins: 49 % of the bins are evaluated on the noisy bench frame, 20 % on a clean one,
//     ~7.5 instructions per bin on average (flat, select-based code: 9-10; the exact kernel: ~20);
//   * 128 VGPRs -> 4 wavefronts per SIMD (the exact kernel: 2), neighbour histograms read two float4 ahead of their use.
#include "bcd_common.h"
#include <hip/hip_fp16.h>
#include <atomic>
#include <cstdlib>


namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

// sixteen zero bytes in global memory: the source of LDS-DMA lanes whose pixel lies outside the image
__device__ const float4 g_rw_zero16 = { 0.f, 0.f, 0.f, 0.f };

constexpr int RW_TW = 64, RW_TH = 4;   // tile
constexpr int RW_CW = 12;              // widest span of column displacements served by one staged window
constexpr int RW_NCOLS = RW_TW + RW_CW;
constexpr int RW_ND = RW_CW + 1;       // displacements per window

constexpr float RW_BIN_MAX = 1048576.f;   // the guarded range of k_similarity.hip (PD_BIN_MAX, PD_N_MIN, PD_N_MAX)
constexpr float RW_N_MIN = 0.0009765625f;
constexpr float RW_N_MAX = 65536.f;

// ---------------------------------------------------------------------------------------------------
// k_pairdist_rw: rolling window, displacements split between two wavefront sets.
//   workgroup = 4 x 64 pixel tile, 8 wavefronts: wavefront (line ty, half hv) owns the 64 pixels of tile line ty (whole
//   histogram in VGPRs) and evaluates the displacements j = hv, hv + 2, ... of the current displacement line.  The neighbour
//   histograms live in a ring of four image lines in LDS (line g in slot g & 3; 64 + 12 columns): going from displacement line
//   dl to dl + 1 replaces ONE line (row0 + dl by row0 + dl + 4) -- 10 staged lines per tile instead of 28 -- and that line
//   is fetched into registers while dl is being evaluated.  No exchange between wavefronts: every wavefront stores whole
//   256-byte lines of the T plane itself.  Search windows wider than 13 columns are covered by several passes over dl, one
//   per chunk of <= 13 column displacements.
// ---------------------------------------------------------------------------------------------------
constexpr int RW_THREADS = 512;

template <int D> struct RwLayout {
    static_assert(D % 4 == 0, "whole float4 groups");
    static constexpr int ROW = ((RW_NCOLS * D + 63) / 64) * 64; // ring line stride (dwords): a multiple of the 64 banks
    static constexpr int LDS_DWORDS = 4 * ROW + 4 * RW_NCOLS;    // + sample counts
    static constexpr int NPRE = (RW_NCOLS * (D / 4) + RW_THREADS - 1) / RW_THREADS; // float4 per thread of one prefetched line
};

// COUNT (measurement only, bcd_hip_selftest_bin_work): the same kernel also counts the bins it evaluates -- per lane (b1 + b2 > 1: the
// reference's own count, DenoisingUnit.cpp:379-381) and per wavefront instruction stream (a bin is issued when ANY of the 64 pairs needs it)
// -- into work_count[0..1]; the production instantiation carries no trace of it.
//
// RATIO (round 6; UNI = false): general sample counts at the cost of uniform ones.  With rho = n1 / n2 the reference's term is
//     (n2 b1 - n1 b2)^2 / (n1 n2 (b1 + b2))  =  (n2 / n1) (b1 - rho b2)^2 / (b1 + b2)
// in real arithmetic: ONE fma per bin instead of two multiplies, a subtraction and a product in the denominator, rho by one reciprocal per pixel pair
// (rho = 1 exactly for equal counts: a frame with a uniform count that is no power of two -- 24 spp -- then runs the arithmetic of the UNI kernel).
// The planes stay approximate planes: what changes is that the kernel and the reference no longer share `diff` and `den`, which adds ABSOLUTE
// errors to the relative bound of the header (u = 2^-24, s = b1 + b2, B = the mass of a histogram, r = max(rho, 1 / rho)):
//   reference vs real arithmetic:  diff = RN(RN(n2 b1) - RN(n1 b2)) is off by <= u (n2 b1 + n1 b2) + u |diff|, so a term by <= 2u r s (+ relative 7u);
//   here vs real arithmetic:       d' = RN(b1 - rho' b2), rho' = rho (1 + 3u): |d' - d| <= u |d| + 3u rho b2, a term by <= 6u max(1, rho) b2 (+ relative);
// summed over the evaluated bins of a pixel pair: |T' - T_ref| <= u r (2 B1 + 8 B2) + the relative part <= 10 u r max(B1, B2).  The kernel records
//   kappa = max over the pixels of B / n   and   G = max over the pairs with C > 0 of  r max(n1, n2) / C
// (so that 10 u r max(B1, B2) <= 10 u kappa G C for every pair), and k_ratio_verdict raises flag bit 2 (value 4) when
//     10 u kappa G  >  2^-12 tau          (what the relative errors leave of the verified band: 2^-10 - 2^-11 [binary16] - 2e-5 [fp32] = 4.7e-4 > 2^-12)
// -- the host then repeats the scale with the reference's operations (RATIO = false).  Pairs with C = 0 have T = 0 exactly either way.
template <int D, bool UNI, bool COUNT = false, bool RATIO = false>
__global__ __launch_bounds__(RW_THREADS, (UNI || RATIO) ? 4 : 2) void k_pairdist_rw(const float *__restrict__ hist, const float *__restrict__ ns, int W, int H,
                                                              int b, __half *__restrict__ T, uint8_t *__restrict__ Cn, int *range_flag,
                                                              float uni_n, int tile_row0 /* first tile row of this launch */,
                                                              unsigned long long *work_count = nullptr, unsigned int *ratio_stats = nullptr /* RATIO: 8 lines of (kappa, G) */)
{
    static_assert(!(UNI && RATIO), "RATIO is a form of the general kernel");
    using L = RwLayout<D>;
    constexpr int Q = D / 4, NPRE = L::NPRE;
    extern __shared__ float4 lds4[];
    float *ring = reinterpret_cast<float *>(lds4);
    float *ring_n = ring + 4 * L::ROW;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // (wave-uniform: loop control on the scalar unit)
    unsigned int cnt_lane_bins = 0, cnt_wave_bins = 0, cnt_wave_groups = 0; // (COUNT only; wave-uniform)
    const int ty = wave & 3, hv = wave >> 2; // the two halves of a tile line land on the same SIMD
    int tile;
    {
        const int nt = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = id & 7, k = id >> 3, q = nt >> 3, rem = nt & 7;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
    }
    const int col0 = (tile % gridDim.x) * RW_TW, row0 = (tile / gridDim.x + tile_row0) * RW_TH; // row0 % 4 == 0: line g of the image lives in slot g & 3
    const int c = col0 + lane, r = row0 + ty;
    const bool inside = (c < W) && (r < H);
    const size_t plane = (size_t)W * H;
    const size_t pix = (size_t)r * W + c;

    // UNI with uni_n < 0: the caller did not look -- the count of the first pixel is taken as the uniform count here, and checked like a
    // given one: it must be a power of two in [1, 65536] (then the count products drop out exactly) and every pixel must carry it
    float un = uni_n;
    if (UNI && uni_n < 0.f) {
        un = ns[0];
        const bool pow2 = (__float_as_uint(un) & 0x7fffffu) == 0u && un >= 1.f && un <= 65536.f;
        if (!pow2) { if (threadIdx.x == 0) range_flag[2] = 1; return; } // (the same for every workgroup: nothing of this launch is used)
    }
    // (round 4) The four lines that open a pass over the displacement lines go straight from global memory into their ring slots
    // (global_load_lds_dwordx4: lane l of a wavefront lands at base + 16 l, which IS the ring's layout; no staging registers), all four
    // at once -- and for the first pass before anything else, so that the own histogram, the range check and the sample-count check run
    // in the shadow of that fetch.  Before: own histogram, checks, then four fetch -> store round trips one after the other, during which
    // the workgroup's eight wavefronts did no arithmetic.  Lanes outside the image copy sixteen zero bytes.
    auto pass_geometry = [&](int cbeg, int &cend, int &nc, int &wcols, int &dl0) __attribute__((always_inline)) {
        cend = min(b, cbeg + RW_CW); nc = cend - cbeg + 1; wcols = RW_TW + (cend - cbeg);
        dl0 = cend >= 0 ? 0 : 1; // displacement line 0 only holds dc >= 0
    };
    auto open_pass = [&](int cbeg, int wcols, int dl0) __attribute__((always_inline)) {
        for (int g = row0 + dl0; g < row0 + dl0 + 4; ++g) {
            float *slot = ring + (g & 3) * L::ROW;
#pragma unroll
            for (int u = 0; u < NPRE; ++u) {
                const int i = threadIdx.x + u * RW_THREADS;
                const int lc = i / Q, q = i - lc * Q, gc = col0 + cbeg + lc;
                const bool in_image = g < H && gc >= 0 && gc < W;
                const float4 *src = in_image ? reinterpret_cast<const float4 *>(hist) + ((size_t)g * W + gc) * Q + q : &g_rw_zero16;
                if (i < wcols * Q) // (whole wavefronts beyond the line's end issue nothing)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(slot + 4 * (wave * 64 + u * RW_THREADS)), 16, 0, 0);
            }
            if (!UNI) {
                const int lc = threadIdx.x, gc = col0 + cbeg + lc;
                if (lc < wcols) ring_n[(g & 3) * RW_NCOLS + lc] = (g < H && gc >= 0 && gc < W) ? ns[(size_t)g * W + gc] : 1.f;
            }
        }
    };
    {
        int cend, nc, wcols, dl0;
        pass_geometry(-b, cend, nc, wcols, dl0);
        open_pass(-b, wcols, dl0);
    }
    float h1[D];
    float n1 = UNI ? un : 1.f;
    {
        const float4 *src = reinterpret_cast<const float4 *>(hist + (inside ? pix * D : 0));
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            float4 v = inside ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            h1[4 * q] = v.x; h1[4 * q + 1] = v.y; h1[4 * q + 2] = v.z; h1[4 * q + 3] = v.w;
        }
        bool other_count = false;
        if (hv == 0) { // every pixel of the image is the own pixel of exactly one (line, half 0) wavefront: this covers the whole input
            bool bad = false;
#pragma unroll
            for (int k = 0; k < D; ++k) bad = bad || !(h1[k] >= 0.f && h1[k] <= RW_BIN_MAX);
            if (inside) {
                const float nv = ns[pix];
                bad = bad || !(nv >= RW_N_MIN && nv <= RW_N_MAX);
                other_count = UNI && nv != un;
            }
            if (bad) atomicOr(range_flag, 1);
        }
        // a pixel with another sample count: the launch is void (the host repeats the pass with the general formula).  One plain store per
        // workgroup into a flag word of its own -- on a frame without a uniform count every pixel would otherwise queue up on one atomic --
        // and the workgroup leaves at once
        if (UNI && __syncthreads_or(other_count)) { if (threadIdx.x == 0) range_flag[2] = 1; return; }
        if (!UNI && inside) n1 = ns[pix];
        if (RATIO && hv == 0) { // kappa: the mass of the own histogram per unit of its sample count, the largest of the wavefront to its XCD's line
            float mass = 0.f;
#pragma unroll
            for (int k = 0; k < D; ++k) mass += h1[k];
            float kap = inside ? mass * __builtin_amdgcn_rcpf(n1) * 1.00001f : 0.f; // (59 additions, a reciprocal, a product: rounded up)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) kap = fmaxf(kap, __shfl_xor(kap, off));
            unsigned int *line = ratio_stats + 16 * (blockIdx.x & 7);
            if (lane == 0 && __float_as_uint(kap) > *reinterpret_cast<volatile unsigned int *>(line)) atomicMax(line, __float_as_uint(kap)); // (non-negative floats order like their bits)
        }
    }
    float gmax = 0.f; // RATIO: largest r max(n1, n2) / C over this lane's pairs

    // one image line (columns col0 + cbeg ...) -> registers / -> its ring slot
    auto fetch_line = [&](float4 (&pre)[NPRE], float &pre_n, int g, int cbeg, int wcols) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int i = threadIdx.x + u * RW_THREADS;
            const int lc = i / Q, q = i - lc * Q, gc = col0 + cbeg + lc;
            pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lc < wcols && g < H && gc >= 0 && gc < W) pre[u] = reinterpret_cast<const float4 *>(hist)[((size_t)g * W + gc) * Q + q];
        }
        if (!UNI) {
            const int lc = threadIdx.x, gc = col0 + cbeg + lc;
            pre_n = (lc < wcols && g < H && gc >= 0 && gc < W) ? ns[(size_t)g * W + gc] : 1.f;
        }
    };
    auto store_line = [&](const float4 (&pre)[NPRE], float pre_n, int g, int wcols) __attribute__((always_inline)) {
        float *dst = ring + (g & 3) * L::ROW;
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int i = threadIdx.x + u * RW_THREADS;
            if (i < wcols * Q) *reinterpret_cast<float4 *>(dst + 4 * i) = pre[u];
        }
        if (!UNI && (int)threadIdx.x < wcols) ring_n[(g & 3) * RW_NCOLS + threadIdx.x] = pre_n;
    };

    for (int cbeg = -b; cbeg <= b; cbeg += RW_ND) {
        int cend, nc, wcols, dl0;
        pass_geometry(cbeg, cend, nc, wcols, dl0);
        float4 pre[NPRE];
        float pre_n = 1.f;
        if (cbeg != -b) { // (the first pass's lines are on their way since the top of the kernel)
            __syncthreads(); // the previous pass is done with the ring
            open_pass(cbeg, wcols, dl0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): the LDS-DMA of this wavefront has landed (hipcc does not count it: counted here)
        __syncthreads();
        for (int dl = dl0; dl <= b; ++dl) {
            // the line that enters the ring for dl + 1 travels while dl is evaluated
            if (dl < b) fetch_line(pre, pre_n, row0 + dl + 4, cbeg, wcols);
            const float *nrow = ring + ((ty + dl) & 3) * L::ROW;
            const float *nrow_n = ring_n + ((ty + dl) & 3) * RW_NCOLS;
            const int nr = r + dl;
            // first displacement of this wavefront on this line (dl == 0 only holds dc >= 0)
            const int j0 = (dl == 0 && cbeg < 0) ? -cbeg : 0;
            int j = j0 + ((j0 ^ hv) & 1); // smallest j >= j0 of this wavefront's parity
            // neighbour histograms are read PF float4 ahead of their use, across displacements: the LDS latency hides behind the
            // arithmetic of the previous bins
            constexpr int PF = 2;
            float4 pf[PF];
            {
                const float *nb0 = nrow + (lane + min(j, nc - 1)) * D;
#pragma unroll
                for (int u = 0; u < PF; ++u) pf[u] = *reinterpret_cast<const float4 *>(nb0 + 4 * u);
            }
            for (; j < nc; j += 2) {
                const int dc = cbeg + j;
                const float *nb = nrow + (lane + j) * D;
                const float *nb_next = nrow + (lane + min(j + 2, nc - 1)) * D; // (a harmless re-read after the last displacement)
                float n2 = 1.f, n12 = 1.f, rho = 1.f;
                if (!UNI) { n2 = nrow_n[lane + j]; n12 = n1 * n2; }
                if (RATIO) rho = n1 == n2 ? 1.f : n1 * __builtin_amdgcn_rcpf(n2);
                // (sum, number of bins counted) as one packed pair: the bin's term and its count go in with a single v_pk_fma_f32
                v2f acc2 = { 0.f, 0.f };
                v2f pd = { 0.f, 1.f }, pr = { 0.f, 1.f };
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const float4 v = pf[q % PF];
                    pf[q % PF] = (q + PF < Q) ? *reinterpret_cast<const float4 *>(nb + 4 * (q + PF)) : *reinterpret_cast<const float4 *>(nb_next + 4 * (q + PF - Q));
                    const float b2[4] = { v.x, v.y, v.z, v.w };
                    // one bin: DenoisingUnit.cpp:379-383 with the division replaced by rcp + fma (error bound in the header)
                    auto term = [&](int e) __attribute__((always_inline)) {
                        const float b1 = h1[4 * q + e];
                        if (UNI) { const float d = b1 - b2[e]; return d * d; }
                        if (RATIO) { const float d = fmaf(-rho, b2[e], b1); return d * d; }
                        const float d = n2 * b1 - n1 * b2[e];
                        return d * d;
                    };
                    auto den = [&](float s_) __attribute__((always_inline)) { return (UNI || RATIO) ? s_ : n12 * s_; };
                    // a wave-uniform branch per group of 4 bins (most groups are empty for all 64 pixels of a line segment), then a
                    // divergent branch (execz) per bin: DenoisingUnit.cpp:379 decides exactly which bins count
                    // (the group test is spelled as four lane masks OR-ed on the scalar unit: written as a per-lane `any`, the compiler
                    // packs the four compare results into a bit field first -- 12 extra vector instructions per group)
                    float sg[4];
                    uint64_t live = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sg[e] = h1[4 * q + e] + b2[e];
                        const uint64_t lm = __builtin_amdgcn_ballot_w64(sg[e] > 1.f);
                        live |= lm;
                        if (COUNT) { cnt_lane_bins += (unsigned int)__builtin_popcountll(lm); cnt_wave_bins += lm != 0 ? 1u : 0u; } // (wave-uniform: outside the divergent bodies)
                    }
                    if (live != 0) {
                        if (COUNT) ++cnt_wave_groups;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (sg[e] > 1.f) {
                                asm volatile(""); // keep the branch: no if-conversion into selects
                                pd.x = term(e);
                                pr.x = __builtin_amdgcn_rcpf(den(sg[e]));
                                acc2 = __builtin_elementwise_fma(pd, pr, acc2);
                            }
                    }
                }
                if (Q % PF != 0) { // the ring of read-ahead registers restarts at slot 0 for the next displacement
                    float4 t[PF];
#pragma unroll
                    for (int u = 0; u < PF; ++u) t[u] = pf[(u + Q) % PF];
#pragma unroll
                    for (int u = 0; u < PF; ++u) pf[u] = t[u];
                }
                const int nc_ = c + dc;
                if (inside && nc_ >= 0 && nc_ < W && nr < H) {
                    const size_t o = (size_t)bcd_delta_index(dl, dc, b) * plane + pix;
                    float tv = acc2.x;
                    if (RATIO) { // (from rho alone: the neighbour's count is not kept across the bins)
                        const float q = rho == 1.f ? 1.f : __builtin_amdgcn_rcpf(rho);     // n2 / n1
                        tv *= q;
                        if (acc2.y > 0.f) gmax = fmaxf(gmax, fmaxf(rho, q) * (n1 * fmaxf(1.f, q)) * __builtin_amdgcn_rcpf(acc2.y)); // r max(n1, n2) / C
                    }
                    T[o] = __float2half_rn(tv); // binary16 plane (a sum beyond 65504 becomes +inf: far above any admitted tau)
                    Cn[o] = (uint8_t)(int)acc2.y;
                }
            }
            if (dl < b) {
                __syncthreads(); // line row0 + dl is not needed any more (nor is any other slot being replaced)
                store_line(pre, pre_n, row0 + dl + 4, wcols);
                __syncthreads();
            }
        }
    }
    if (RATIO) {
        gmax *= 1.00001f; // (rho carries 3u, then three reciprocals and three products: rounded up)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off));
        unsigned int *line = ratio_stats + 16 * (blockIdx.x & 7) + 1;
        if (lane == 0 && __float_as_uint(gmax) > *reinterpret_cast<volatile unsigned int *>(line)) atomicMax(line, __float_as_uint(gmax));
    }
    if (COUNT && lane == 0 && work_count) {
        atomicAdd(work_count, (unsigned long long)cnt_lane_bins);
        atomicAdd(work_count + 1, (unsigned long long)cnt_wave_bins);
        atomicAdd(work_count + 2, (unsigned long long)cnt_wave_groups);
    }
}

// the a-posteriori check of the RATIO form (see k_pairdist_rw): one wavefront
__global__ void k_ratio_verdict(const unsigned int *__restrict__ stats, float tau, int *range_flag)
{
    const int lane = threadIdx.x;
    float kap = lane < 8 ? __uint_as_float(stats[16 * lane]) : 0.f, g = lane < 8 ? __uint_as_float(stats[16 * lane + 1]) : 0.f;
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) { kap = fmaxf(kap, __shfl_xor(kap, off)); g = fmaxf(g, __shfl_xor(g, off)); }
    // 10 u kappa G <= 2^-12 tau, u = 2^-24  <=>  10 kappa G <= 4096 tau; NaN (a NaN count or bin: also caught by the range flag) fails
    if (lane == 0 && !(10.f * kap * g <= 4096.f * tau)) atomicOr(range_flag, 4);
}

// ---------------------------------------------------------------------------------------------------
// Exact re-evaluation of the borderline pairs: entry = (pixel index, displacement index); nine lanes evaluate the nine pixel
// pairs of the patch with the reference's operation sequence (DenoisingUnit.cpp:360-386: sequential bins, IEEE division),
// one lane adds them in patch order (:336-358) and sets the forward bit when d <= tau.
// A pixel's 240-byte histogram row read as fifteen separate 16-byte loads of ONE lane (the form of rounds 2-4) keeps nothing in the L1 -- 63 rows per
// instruction, every 64-byte line fetched up to four times -- and at 2e5 .. 1e6 borderline pairs (textured frames, b = 12) the kernel was bound by that
// traffic (0.5 .. 1.9 ms).  Here the 126 rows of a wavefront's 63 pixel pairs are loaded in pieces of five 16-byte groups by FIVE CONSECUTIVE lanes each
// (80 contiguous bytes), parked in LDS, and every pair lane then reads its own two rows back.
// ---------------------------------------------------------------------------------------------------
constexpr int VP_PIECE = 5;                       // 16-byte groups per piece (20 bins)
constexpr int VP_ROWS = 126;                      // rows per wavefront: 63 pixel pairs x 2
constexpr int VP_STRIDE = VP_PIECE * 4 + 1;       // floats per parked row piece (odd: conflict-free column reads)
__global__ __launch_bounds__(64) void k_verify_pairs_lds(const float *__restrict__ hist, const float *__restrict__ ns, int W, int H, int Q /* D / 4 */, int b,
                                                         float tau, const uint2 *__restrict__ list, const int *__restrict__ d_count, int capacity,
                                                         int fwords, uint32_t *__restrict__ fwd)
{
    __shared__ float s_piece[VP_ROWS * VP_STRIDE];
    __shared__ unsigned int s_row[VP_ROWS + 2];   // pixel index of every row (x of pair 0, y of pair 0, x of pair 1, ...)
    const int lane = threadIdx.x, slot = lane / 9, o = lane - slot * 9; // 7 entries per wavefront, lane 63 idle
    const int n = min(*d_count, capacity);
    const int side = 2 * b + 1;
    for (int base = blockIdx.x * 7; base < n; base += gridDim.x * 7) {
        const int e = base + slot;
        const bool live = slot < 7 && e < n;
        uint2 ent = make_uint2(0u, 0u);
        unsigned int x = 0u, y = 0u;
        if (live) {
            ent = list[e];
            const int p = (int)ent.x, didx = (int)ent.y;
            int dl = 0, dc = didx;
            if (didx > b) { const int t = didx - (b + 1); dl = 1 + t / side; dc = t - (dl - 1) * side - b; }
            const int pr = p / W, pc = p - pr * W;
            const int xr = pr + o / 3 - 1, xc = pc + o % 3 - 1; // patch pixel o (row-major), and its partner
            x = (unsigned int)(xr * W + xc); y = (unsigned int)((xr + dl) * W + (xc + dc));
        }
        __syncthreads(); // (the previous trip's readers are done with the tables)
        if (lane < 63) { s_row[2 * lane] = x; s_row[2 * lane + 1] = y; } // (dead pairs: pixel 0 -- loaded, never used)
        __syncthreads();
        const float n1 = live ? ns[x] : 1.f, n2 = live ? ns[y] : 1.f, n12 = n1 * n2;
        float sum = 0.f;
        int cnt = 0;
        // one bin of DenoisingUnit.cpp:379-383, in the reference's order
        auto bin = [&](float b1, float b2) __attribute__((always_inline)) {
            const float s = b1 + b2;
            if (s <= 1.f) return;
            ++cnt;
            const float diff = n2 * b1 - n1 * b2;
            sum += diff * diff / (n12 * s);
        };
        for (int q0 = 0; q0 < Q; q0 += VP_PIECE) {
            const int np = min(VP_PIECE, Q - q0); // groups of this piece
            // VP_ROWS rows x VP_PIECE groups, a run of consecutive lanes per row
            float4 v[(VP_ROWS * VP_PIECE + 63) / 64];
#pragma unroll
            for (int u = 0; u < (VP_ROWS * VP_PIECE + 63) / 64; ++u) {
                const int i = lane + 64 * u, row = i / VP_PIECE, g = i - row * VP_PIECE; // (a short last piece leaves the lanes g >= np idle)
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < VP_ROWS && g < np) v[u] = reinterpret_cast<const float4 *>(hist)[(size_t)s_row[row] * Q + q0 + g];
            }
            __syncthreads(); // (the previous piece has been consumed)
#pragma unroll
            for (int u = 0; u < (VP_ROWS * VP_PIECE + 63) / 64; ++u) {
                const int i = lane + 64 * u, row = i / VP_PIECE, g = i - row * VP_PIECE;
                if (row < VP_ROWS) { float *d = s_piece + row * VP_STRIDE + 4 * g; d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w; }
            }
            __syncthreads();
            if (live) {
                const float *r1 = s_piece + (2 * lane) * VP_STRIDE, *r2 = r1 + VP_STRIDE;
                for (int k = 0; k < 4 * np; ++k) bin(r1[k], r2[k]);
            }
        }
        // patch order: ((((t0 + t1) + t2) + ...) + t8), counts as integers
        float tot = 0.f;
        int ctot = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int srcl = min(slot * 9 + i, 63);
            const float ti = __shfl(sum, srcl);
            const int ci = __shfl(cnt, srcl);
            tot = (i == 0) ? ti : tot + ti; // the reference starts from 0.f: 0 + t0 == t0 bit for bit (t0 >= +0)
            ctot += ci;
        }
        if (live && o == 0) {
            const float d = tot / (float)ctot; // 0/0 = NaN -> not similar
            if (d <= tau) atomicOr(fwd + (size_t)(ent.y >> 5) * W * H + ent.x, 1u << (ent.y & 31)); // (word-plane layout: word * pixels + pixel)
        }
    }
}

// self-test: largest relative deviation between the patch distances from two sets of planes (approximate vs exact), over all
// pairs of main pixels; bits of (max rel, as uint) via atomicMax -- valid because the values are non-negative floats
__global__ __launch_bounds__(256) void k_max_rel_dev(const __half *__restrict__ Ta, const float *__restrict__ Tb, const uint8_t *__restrict__ Ca,
                                                     const uint8_t *__restrict__ Cb, int W, int H, int b, unsigned int *__restrict__ out /* [0] max rel bits, [1] count mismatches */)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6), didx = blockIdx.z;
    const int side = 2 * b + 1;
    int dl = 0, dc = didx;
    if (didx > b) { const int t = didx - (b + 1); dl = 1 + t / side; dc = t - (dl - 1) * side - b; }
    if (c < 1 || c > W - 2 || r < 1 || r > H - 2 || c + dc < 1 || c + dc > W - 2 || r + dl > H - 2) return;
    const size_t plane = (size_t)W * H, base = (size_t)didx * plane;
    float sa = 0.f, sb = 0.f;
    int na = 0, nb = 0;
    for (int ol = -1; ol <= 1; ++ol)
        for (int oc = -1; oc <= 1; ++oc) {
            const size_t i = base + (size_t)(r + ol) * W + (c + oc);
            sa += __half2float(Ta[i]); sb += Tb[i]; na += Ca[i]; nb += Cb[i];
        }
    if (na != nb) { atomicAdd(out + 1, 1u); return; }
    if (nb == 0) return;
    if (isinf(sa) && sb > 65504.f) return; // a binary16 entry overflowed: such a pair is far above any admitted threshold
    const float rel = sb > 0.f ? fabsf(sa - sb) / sb : (sa == 0.f ? 0.f : 1.f);
    atomicMax(out, __float_as_uint(rel));
}

} // namespace

// 1 if the fast (approximate + verify) path has a kernel for this histogram depth
int bcd_pairdist_rw_supported(int D) { return D == 60 || D == 36 || D == 24; }

// tile rows [tile_row_begin, tile_row_end) of the frame (a tile row = RW_TH image lines; end < 0: to the last one).  A launch reads the
// histogram lines of its tile rows and the b lines below them: callers that stream the frame in (bcd_hip_denoise_host_ex) launch the tile rows
// whose lines have arrived.
int bcd_pairdist_rw_tile_lines() { return RW_TH; }

static hipError_t pairdist_rw_rows(const float *hist, const float *ns, int W, int H, int D, int b, void *T /* binary16 planes */, uint8_t *Cn, int *d_range_flag,
                                   float uni_n, int tile_row_begin, int tile_row_end, hipStream_t st, unsigned long long *work_count)
{
    const int tile_rows = (H + RW_TH - 1) / RW_TH;
    if (tile_row_end < 0 || tile_row_end > tile_rows) tile_row_end = tile_rows;
    if (tile_row_begin < 0 || tile_row_begin >= tile_row_end) return tile_row_begin == tile_row_end ? hipSuccess : hipErrorInvalidValue;
    dim3 grid((W + RW_TW - 1) / RW_TW, tile_row_end - tile_row_begin), block(RW_THREADS);
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
#define BCD_RW_LAUNCH(DD, UU)                                                                                        \
    {                                                                                                                \
        const size_t lds = (size_t)RwLayout<DD>::LDS_DWORDS * 4;                                                     \
        static std::atomic<int> granted[64]; /* per instantiation and device */                                     \
        if (lds > 64 * 1024 && (dev < 0 || dev >= 64 || granted[dev].load() == 0)) {                                 \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pairdist_rw<DD, UU>),               \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
            if (e != hipSuccess) return e;                                                                           \
            if (dev >= 0 && dev < 64) granted[dev].store(1);                                                         \
        }                                                                                                            \
        if (work_count) { /* the counting instantiation (measurement only) */                                       \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pairdist_rw<DD, UU, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                           \
            hipLaunchKernelGGL((k_pairdist_rw<DD, UU, true>), grid, block, lds, st, hist, ns, W, H, b, static_cast<__half *>(T), Cn, d_range_flag, uni_n, tile_row_begin, work_count); \
            return hipGetLastError();                                                                                \
        }                                                                                                            \
        hipLaunchKernelGGL((k_pairdist_rw<DD, UU>), grid, block, lds, st, hist, ns, W, H, b, static_cast<__half *>(T), Cn, d_range_flag, uni_n, tile_row_begin, (unsigned long long *)nullptr); \
        return hipGetLastError();                                                                                    \
    }
#define BCD_RW_DEPTH(DD) case DD: if (uni_n != 0.f) BCD_RW_LAUNCH(DD, true) else BCD_RW_LAUNCH(DD, false)
    switch (D) {
    BCD_RW_DEPTH(60)
    BCD_RW_DEPTH(36)
    BCD_RW_DEPTH(24)
    default: break;
    }
#undef BCD_RW_DEPTH
#undef BCD_RW_LAUNCH
    return hipErrorInvalidValue;
}

// general sample counts by the RATIO form (whole frame): clears `stats` (128 words), runs the kernel and its verdict (flag value 4 in d_range_flag[0]
// when the form's absolute-error check fails: the caller repeats the pass with bcd_launch_pairdist_rw(..., uni_n = 0))
hipError_t bcd_launch_pairdist_rw_ratio(const float *hist, const float *ns, int W, int H, int D, int b, void *T /* binary16 planes */, uint8_t *Cn, int *d_range_flag,
                                        float tau, unsigned int *stats, hipStream_t st)
{
    if (!stats) return hipErrorInvalidValue;
    dim3 grid((W + RW_TW - 1) / RW_TW, (H + RW_TH - 1) / RW_TH), block(RW_THREADS);
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    hipError_t e = hipMemsetAsync(stats, 0, 128 * sizeof(unsigned int), st);
    if (e != hipSuccess) return e;
#define BCD_RW_RATIO(DD)                                                                                             \
    case DD: {                                                                                                       \
        const size_t lds = (size_t)RwLayout<DD>::LDS_DWORDS * 4;                                                     \
        static std::atomic<int> granted[64];                                                                         \
        if (lds > 64 * 1024 && (dev < 0 || dev >= 64 || granted[dev].load() == 0)) {                                 \
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pairdist_rw<DD, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                           \
            if (dev >= 0 && dev < 64) granted[dev].store(1);                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL((k_pairdist_rw<DD, false, false, true>), grid, block, lds, st, hist, ns, W, H, b, static_cast<__half *>(T), Cn, d_range_flag, 0.f, 0, \
                           (unsigned long long *)nullptr, stats);                                                    \
    } break;
    switch (D) {
    BCD_RW_RATIO(60)
    BCD_RW_RATIO(36)
    BCD_RW_RATIO(24)
    default: return hipErrorInvalidValue;
    }
#undef BCD_RW_RATIO
    hipLaunchKernelGGL(k_ratio_verdict, dim3(1), dim3(64), 0, st, stats, tau, d_range_flag);
    return hipGetLastError();
}

hipError_t bcd_launch_pairdist_rw_rows(const float *hist, const float *ns, int W, int H, int D, int b, void *T /* binary16 planes */, uint8_t *Cn, int *d_range_flag,
                                       float uni_n, int tile_row_begin, int tile_row_end, hipStream_t st)
{
    return pairdist_rw_rows(hist, ns, W, H, D, b, T, Cn, d_range_flag, uni_n, tile_row_begin, tile_row_end, st, nullptr);
}

// measurement: the whole frame with the counting instantiation; work_count[0] += evaluated (pair, bin) terms, [1] += bins issued by a wavefront, [2] += groups of four entered
hipError_t bcd_launch_pairdist_rw_counting(const float *hist, const float *ns, int W, int H, int D, int b, void *T, uint8_t *Cn, int *d_range_flag, float uni_n,
                                           unsigned long long *work_count, hipStream_t st)
{
    return pairdist_rw_rows(hist, ns, W, H, D, b, T, Cn, d_range_flag, uni_n, 0, -1, st, work_count);
}

hipError_t bcd_launch_pairdist_rw(const float *hist, const float *ns, int W, int H, int D, int b, void *T /* binary16 planes */, uint8_t *Cn, int *d_range_flag,
                                  float uni_n, hipStream_t st)
{
    return bcd_launch_pairdist_rw_rows(hist, ns, W, H, D, b, T, Cn, d_range_flag, uni_n, 0, -1, st);
}

hipError_t bcd_launch_verify_pairs(const float *hist, const float *ns, int W, int H, int D, int b, float tau, const void *list, const int *d_count,
                                   int capacity, uint32_t *fwd, hipStream_t st)
{
    const int fwords = (bcd_delta_count(b) + 31) / 32;
    // (the number of pairs is on the device: a fixed grid of 8 wavefronts per CU; with nothing listed every wavefront leaves at once)
    if ((D & 3) != 0) return hipErrorInvalidValue; // (every depth bcd_pairdist_rw_supported admits is a multiple of four)
    // grid (round 6, measured on 2.7 M pairs of a 3840x2160 b = 12 scale): 512 wavefronts 3.0 ms (latency: five dependent round trips per trip of 63
    // pixel pairs), 1024 1.66, 1536 .. 2048 1.20, 3072 .. 7168 1.41, 4096 1.52 (14 workgroups fit a CU's LDS: the last 512 ran as a second round) --
    // more wavefronts in flight only spread the rows being fetched over more of the list, and the L2 keeps fewer of the shared ones
    hipLaunchKernelGGL(k_verify_pairs_lds, dim3(2048), dim3(64), 0, st, hist, ns, W, H, D >> 2, b, tau, (const uint2 *)list, d_count, capacity, fwords, fwd);
    return hipGetLastError();
}

hipError_t bcd_launch_max_rel_dev(const float *Ta, const float *Tb, const uint8_t *Ca, const uint8_t *Cb, int W, int H, int b, unsigned int *out,
                                  hipStream_t st)
{
    hipLaunchKernelGGL(k_max_rel_dev, dim3((W + 63) / 64, (H + 3) / 4, bcd_delta_count(b)), dim3(256), 0, st, reinterpret_cast<const __half *>(Ta), Tb, Ca, Cb, W, H, b, out);
    return hipGetLastError();
}
