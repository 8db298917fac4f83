// k_similarity_fast.hip -- similar-patch selection, fast path: APPROXIMATE pair-distance planes with a rigorous error
// bound, decided exactly only at the threshold.
//
// The planes T_delta(x), C_delta(x) of k_similarity.hip are consumed by exactly one test, d(p, p+delta) <= tau
// (src/core/DenoisingUnit.cpp:209).  Here T is evaluated with one v_rcp_f32 + fma per bin (error bound below) and in any
// summation order; C (integer bin counts, the `b1 + b2 <= 1` skip test of DenoisingUnit.cpp:379) stays exact.  The mask
// kernel then decides every pair whose approximate distance lies outside tau (1 +- BCD_APPROX_DELTA) and appends the few
// borderline pairs to a list; k_verify_pairs re-evaluates those with the reference's exact operation sequence
// (sequential bins, IEEE division) and sets their bits.  The masks are therefore bit-identical to the exact path.
//
// Error bound (fp32, u = 2^-24; inputs inside the guarded range of k_similarity.hip, so nothing overflows or is
// subnormal on an evaluated bin):  reference term  t = RN(RN(diff^2) / den),   here  t' = RN-fused(RN(diff^2) * rcp(den)),
// with the SAME diff and den (computed by the same operations as the reference: den = RN(RN(n1 n2) RN(b1+b2)), diff =
// RN(RN(n2 b1) - RN(n1 b2)); for uniform power-of-two counts RN(b1-b2) and RN(b1+b2), see k_similarity.hip).  v_rcp_f32 is
// accurate to 1 ulp (|rel| <= 2u), so |t'/t - 1| <= 3u + O(u^2) before the accumulation; all terms are >= 0, hence any
// summation order of the <= 9 D terms of a patch distance carries a relative error <= (9 D - 1) u in either path, the
// final division one more u.  For D = 60:  |d'/d - 1| <= (3 + 2 * 540 + 2) u ~ 6.5e-5 in the WORST case over all
// round-offs pointing the same way -- BCD_APPROX_DELTA = 2^-13 = 1.22e-4 is almost twice that; the measured maximum over
// whole frames is ~3e-7 (bcd_hip_selftest_approx_distance reports it; the GPU tests assert it stays below delta / 16).
// Depths above 120 bins do not use this path.
//
// Layout of the kernel (gfx950, SIMD-32, 512 VGPRs per lane and SIMD, 160 KB LDS per CU):
//   workgroup = 4 x 64 pixel tile, 12 wavefronts = 4 pixel patches of 16 x 4  x  3 thirds of the histogram (the colour
//   channels when D = 3 x nbOfBins): a wavefront keeps D/3 = 20 bins of its own pixels in VGPRs (~75 VGPRs: 6 wavefronts
//   per SIMD instead of the 2 of the exact kernel), all 12 share one staged window of neighbour histograms in LDS (73 KB:
//   two workgroups per CU = 24 wavefronts per CU).  With that occupancy LDS reads, branches and scalar work issue in the
//   shadow of the VALU, and a divergent `if (b1 + b2 > 1)` around the 5 arithmetic instructions of a bin (execz branch)
//   skips a bin as soon as it is empty for the 64 pixels of the wavefront: 49 % of the bins are evaluated on the noisy
//   bench frame, 20 % on a clean one.  The partial sums of the three thirds meet in LDS once per staged window (13
//   displacements), where the 12 wavefronts also re-map them to full 256-byte plane lines for the stores.
#include "bcd_common.h"
#include <atomic>

namespace {

constexpr int CS_TW = 64, CS_TH = 4;   // tile
constexpr int CS_CW = 12;              // widest span of column displacements served by one staged window
constexpr int CS_NCOLS = CS_TW + CS_CW;
constexpr int CS_ND = CS_CW + 1;       // displacements per window
constexpr int CS_WAVES = 12, CS_THREADS = CS_WAVES * 64;

constexpr float CS_BIN_MAX = 1048576.f;   // the guarded range of k_similarity.hip (PD_BIN_MAX, PD_N_MIN, PD_N_MAX)
constexpr float CS_N_MIN = 0.0009765625f;
constexpr float CS_N_MAX = 65536.f;

template <int D> struct CsLayout {
    static_assert(D % 12 == 0, "three thirds of whole float4 groups");
    static constexpr int NB = D / 3;                              // bins per wavefront
    static constexpr int ROW = ((CS_NCOLS * D + 63) / 64) * 64;   // window line stride (dwords): a multiple of the 64 banks
    static constexpr int WIN = CS_TH * ROW;                       // window (dwords)
    static constexpr int EXT = 3 * CS_ND * 256;                   // exchange area, T partials (aliases the window)
    static constexpr int EXC = 3 * 4 * 256;                       //                C partials, four 8-bit counts per dword
    static constexpr int BUF = WIN > EXT + EXC ? WIN : EXT + EXC; // window and exchange area share one buffer
    static constexpr int LDS_DWORDS = BUF + CS_TH * CS_NCOLS;     // + sample counts of the window
};

template <int D, bool UNI>
__global__ __launch_bounds__(CS_THREADS) void k_pairdist_cs(const float *__restrict__ hist, const float *__restrict__ ns, int W, int H,
                                                              int b, float *__restrict__ T, uint8_t *__restrict__ Cn, int *range_flag,
                                                              float uni_n)
{
    using L = CsLayout<D>;
    constexpr int NB = L::NB, Q = NB / 4;
    extern __shared__ float4 lds4[];
    float *win = reinterpret_cast<float *>(lds4);
    float *win_n = win + L::BUF;
    float *exT = win;
    uint32_t *exC = reinterpret_cast<uint32_t *>(win + L::EXT);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ch = wave >> 2, patch = wave & 3; // the four patches of a third land on the four SIMDs
    const int tx = (patch << 4) | (lane & 15), ty = lane >> 4;
    // XCD-aware tile order (see k_pairdist): each XCD gets one contiguous band of tiles, so the re-reads of a neighbour line by
    // the 7 displacement rows of vertically adjacent tiles hit one L2
    int tile;
    {
        const int nt = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = id & 7, k = id >> 3, q = nt >> 3, rem = nt & 7;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
    }
    const int col0 = (tile % gridDim.x) * CS_TW, row0 = (tile / gridDim.x) * CS_TH;
    const int c = col0 + tx, r = row0 + ty;
    const bool inside = (c < W) && (r < H);
    const size_t plane = (size_t)W * H;
    const size_t pix = (size_t)r * W + c;

    float h1[NB];
    float n1 = UNI ? uni_n : 1.f;
    {
        const float4 *src = reinterpret_cast<const float4 *>(hist + (inside ? pix * D : 0) + ch * NB);
        bool bad = false;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            float4 v = inside ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            h1[4 * q] = v.x; h1[4 * q + 1] = v.y; h1[4 * q + 2] = v.z; h1[4 * q + 3] = v.w;
        }
        // every pixel of the image is the own pixel of exactly one tile: checking the own values covers the whole input
#pragma unroll
        for (int k = 0; k < NB; ++k) bad = bad || !(h1[k] >= 0.f && h1[k] <= CS_BIN_MAX);
        if (inside) {
            const float nv = ns[pix];
            bad = bad || !(nv >= CS_N_MIN && nv <= CS_N_MAX);
            if (UNI) { if (nv != uni_n) atomicOr(range_flag, 2); }
            else n1 = nv;
        }
        if (bad) atomicOr(range_flag, 1);
    }

    int didx = 0;
    for (int dl = 0; dl <= b; ++dl)
      for (int cbeg = (dl == 0) ? 0 : -b; cbeg <= b; cbeg += CS_ND) {
        const int cend = min(b, cbeg + CS_CW), nc = cend - cbeg + 1;
        __syncthreads(); // the combine step of the previous window has read the exchange area
        // ---- stage lines row0+dl .. row0+dl+3, columns col0+cbeg .. col0+63+cend (whole pixels: D/4 float4 each)
        {
            const int wcols = CS_TW + (cend - cbeg);
            const int npx = CS_TH * wcols;
            for (int i = threadIdx.x; i < npx * (D / 4); i += CS_THREADS) {
                const int p = i / (D / 4), q = i - p * (D / 4);
                const int lr = p / wcols, lc = p - lr * wcols;
                const int gr = row0 + dl + lr, gc = col0 + cbeg + lc;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gr < H && gc >= 0 && gc < W) v = reinterpret_cast<const float4 *>(hist)[((size_t)gr * W + gc) * (D / 4) + q];
                *reinterpret_cast<float4 *>(win + lr * L::ROW + lc * D + 4 * q) = v;
            }
            if (!UNI)
                for (int i = threadIdx.x; i < npx; i += CS_THREADS) {
                    const int lr = i / wcols, lc = i - lr * wcols;
                    const int gr = row0 + dl + lr, gc = col0 + cbeg + lc;
                    win_n[lr * CS_NCOLS + lc] = (gr < H && gc >= 0 && gc < W) ? ns[(size_t)gr * W + gc] : 1.f;
                }
        }
        __syncthreads();

        float Tp[CS_ND];
        uint32_t Cp[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
        for (int j = 0; j < CS_ND; ++j) {
            Tp[j] = 0.f;
            if (j < nc) {
                const float *nb = win + ty * L::ROW + (tx + j) * D + ch * NB;
                float n2 = 1.f, n12 = 1.f;
                if (!UNI) { n2 = win_n[ty * CS_NCOLS + tx + j]; n12 = n1 * n2; }
                float sum = 0.f;
                uint32_t cnt = 0;
                float4 nbv[Q]; // the whole third of the neighbour's histogram first: Q ds_read_b128 in flight
#pragma unroll
                for (int q = 0; q < Q; ++q) nbv[q] = *reinterpret_cast<const float4 *>(nb + 4 * q);
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const float b2[4] = { nbv[q].x, nbv[q].y, nbv[q].z, nbv[q].w };
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float b1 = h1[4 * q + e];
                        const float s = b1 + b2[e];
                        if (s > 1.f) { // DenoisingUnit.cpp:379 (exact); a bin that is empty for the whole wavefront is skipped (execz)
                            asm volatile(""); // keep the branch: no if-conversion into selects
                            if (UNI) {
                                const float d = b1 - b2[e];
                                sum = fmaf(d * d, __builtin_amdgcn_rcpf(s), sum);
                            } else {
                                const float d = n2 * b1 - n1 * b2[e];
                                sum = fmaf(d * d, __builtin_amdgcn_rcpf(n12 * s), sum);
                            }
                            ++cnt;
                        }
                    }
                }
                Tp[j] = sum;
                Cp[j >> 2] |= cnt << (8 * (j & 3));
            }
        }
        __syncthreads(); // every wavefront is done with the window: it becomes the exchange area
        {
            const int slot = (patch << 6) | lane;
#pragma unroll
            for (int j = 0; j < CS_ND; ++j)
                if (j < nc) exT[(ch * CS_ND + j) * 256 + slot] = Tp[j];
#pragma unroll
            for (int q = 0; q < 4; ++q) exC[(ch * 4 + q) * 256 + slot] = Cp[q];
        }
        __syncthreads();
        // ---- combine the three thirds; a wavefront stores whole tile lines (64 consecutive pixels of a plane)
        for (int job = wave; job < nc * CS_TH; job += CS_WAVES) {
            const int j = job >> 2, line = job & 3;
            const int slot = ((lane >> 4) << 6) | (line << 4) | (lane & 15); // pixel (line, column = lane) of the tile
            const float t = (exT[(0 * CS_ND + j) * 256 + slot] + exT[(1 * CS_ND + j) * 256 + slot]) + exT[(2 * CS_ND + j) * 256 + slot];
            const uint32_t sh = 8 * (j & 3);
            const uint32_t cn = ((exC[(0 * 4 + (j >> 2)) * 256 + slot] >> sh) & 255u) + ((exC[(1 * 4 + (j >> 2)) * 256 + slot] >> sh) & 255u) +
                                ((exC[(2 * 4 + (j >> 2)) * 256 + slot] >> sh) & 255u);
            const int oc = col0 + lane, orow = row0 + line;
            const int qc = oc + cbeg + j, qr = orow + dl;
            if (oc < W && orow < H && qc >= 0 && qc < W && qr < H) {
                const size_t o = (size_t)(didx + j) * plane + (size_t)orow * W + oc;
                T[o] = t;
                Cn[o] = (uint8_t)cn;
            }
        }
        didx += nc;
      }
}

// ---------------------------------------------------------------------------------------------------
// Exact re-evaluation of the borderline pairs: entry = (pixel index, displacement index); nine lanes evaluate the nine pixel
// pairs of the patch with the reference's operation sequence (DenoisingUnit.cpp:360-386: sequential bins, IEEE division),
// one lane adds them in patch order (:336-358) and sets the forward bit when d <= tau.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_verify_pairs(const float *__restrict__ hist, const float *__restrict__ ns, int W, int H, int D, int b,
                                                     float tau, const uint2 *__restrict__ list, const int *__restrict__ d_count, int capacity,
                                                     int fwords, uint32_t *__restrict__ fwd)
{
    const int lane = threadIdx.x, slot = lane / 9, o = lane - slot * 9; // 7 entries per wavefront, lane 63 idle
    const int n = min(*d_count, capacity);
    const int side = 2 * b + 1;
    for (int base = blockIdx.x * 7; base < n; base += gridDim.x * 7) {
        const int e = base + slot;
        const bool live = slot < 7 && e < n;
        float sum = 0.f;
        int cnt = 0;
        uint2 ent = make_uint2(0u, 0u);
        if (live) {
            ent = list[e];
            const int p = (int)ent.x, didx = (int)ent.y;
            int dl = 0, dc = didx;
            if (didx > b) { const int t = didx - (b + 1); dl = 1 + t / side; dc = t - (dl - 1) * side - b; }
            const int pr = p / W, pc = p - pr * W;
            const int xr = pr + o / 3 - 1, xc = pc + o % 3 - 1; // patch pixel o (row-major), and its partner
            const size_t x = (size_t)xr * W + xc, y = (size_t)(xr + dl) * W + (xc + dc);
            const float *h1 = hist + x * D, *h2 = hist + y * D;
            const float n1 = ns[x], n2 = ns[y], n12 = n1 * n2;
            for (int k = 0; k < D; ++k) {
                const float b1 = h1[k], b2 = h2[k], s = b1 + b2;
                if (s <= 1.f) continue;
                ++cnt;
                const float diff = n2 * b1 - n1 * b2;
                sum += diff * diff / (n12 * s);
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
            if (d <= tau) atomicOr(fwd + (size_t)ent.x * fwords + (ent.y >> 5), 1u << (ent.y & 31));
        }
    }
}

// self-test: largest relative deviation between the patch distances from two sets of planes (approximate vs exact), over all
// pairs of main pixels; bits of (max rel, as uint) via atomicMax -- valid because the values are non-negative floats
__global__ __launch_bounds__(256) void k_max_rel_dev(const float *__restrict__ Ta, const float *__restrict__ Tb, const uint8_t *__restrict__ Ca,
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
            sa += Ta[i]; sb += Tb[i]; na += Ca[i]; nb += Cb[i];
        }
    if (na != nb) { atomicAdd(out + 1, 1u); return; }
    if (nb == 0) return;
    const float rel = sb > 0.f ? fabsf(sa - sb) / sb : (sa == 0.f ? 0.f : 1.f);
    atomicMax(out, __float_as_uint(rel));
}

} // namespace

size_t bcd_pairdist_cs_lds_bytes(int D)
{
    switch (D) {
    case 60: return (size_t)CsLayout<60>::LDS_DWORDS * 4;
    case 36: return (size_t)CsLayout<36>::LDS_DWORDS * 4;
    case 24: return (size_t)CsLayout<24>::LDS_DWORDS * 4;
    default: return 0;
    }
}

// 1 if the fast (approximate + verify) path has a kernel for this histogram depth
int bcd_pairdist_cs_supported(int D) { return D == 60 || D == 36 || D == 24; }

hipError_t bcd_launch_pairdist_cs(const float *hist, const float *ns, int W, int H, int D, int b, float *T, uint8_t *Cn, int *d_range_flag,
                                  float uni_n, hipStream_t st)
{
    dim3 grid((W + CS_TW - 1) / CS_TW, (H + CS_TH - 1) / CS_TH), block(CS_THREADS);
    const size_t lds = bcd_pairdist_cs_lds_bytes(D);
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
#define BCD_CS_LAUNCH(DD, UU)                                                                                        \
    {                                                                                                                \
        static std::atomic<int> granted[64];                                                                         \
        if (lds > 64 * 1024 && (dev < 0 || dev >= 64 || granted[dev].load() == 0)) {                                 \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pairdist_cs<DD, UU>),               \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
            if (e != hipSuccess) return e;                                                                           \
            if (dev >= 0 && dev < 64) granted[dev].store(1);                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL((k_pairdist_cs<DD, UU>), grid, block, lds, st, hist, ns, W, H, b, T, Cn, d_range_flag, uni_n); \
        return hipGetLastError();                                                                                    \
    }
    switch (D) {
    case 60: if (uni_n > 0.f) BCD_CS_LAUNCH(60, true) else BCD_CS_LAUNCH(60, false)
    case 36: if (uni_n > 0.f) BCD_CS_LAUNCH(36, true) else BCD_CS_LAUNCH(36, false)
    case 24: if (uni_n > 0.f) BCD_CS_LAUNCH(24, true) else BCD_CS_LAUNCH(24, false)
    default: break;
    }
#undef BCD_CS_LAUNCH
    return hipErrorInvalidValue;
}

hipError_t bcd_launch_verify_pairs(const float *hist, const float *ns, int W, int H, int D, int b, float tau, const void *list, const int *d_count,
                                   int capacity, uint32_t *fwd, hipStream_t st)
{
    const int fwords = (bcd_delta_count(b) + 31) / 32;
    hipLaunchKernelGGL(k_verify_pairs, dim3(2048), dim3(64), 0, st, hist, ns, W, H, D, b, tau, (const uint2 *)list, d_count, capacity, fwords, fwd);
    return hipGetLastError();
}

hipError_t bcd_launch_max_rel_dev(const float *Ta, const float *Tb, const uint8_t *Ca, const uint8_t *Cb, int W, int H, int b, unsigned int *out,
                                  hipStream_t st)
{
    hipLaunchKernelGGL(k_max_rel_dev, dim3((W + 63) / 64, (H + 3) / 4, bcd_delta_count(b)), dim3(256), 0, st, Ta, Tb, Ca, Cb, W, H, b, out);
    return hipGetLastError();
}
