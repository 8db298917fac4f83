// k_similarity.hip -- chi-square histogram patch distances and similar-patch bitmasks for a whole scale.
//
// Replaces, for every main pixel at once, DenoisingUnit::selectSimilarPatches ->
// histogramPatchDistance -> pixelSummedHistogramDistance (src/core/DenoisingUnit.cpp:196-219,336-386),
// and the per-main-pixel CUDA offload (src/core/CudaHistogramDistance.cu:72-239).
//
// Decomposition (exact, no reassociation of the reference's sums):
//   d(p, p+delta) = ( ((T(p+o0) + T(p+o1)) + ...) ) / (float)(sum_o C(p+o)),   o over the patch, row-major
//   T_delta(x) = sequential-in-bin chi-square sum between pixels x and x+delta, C_delta(x) its bin count.
// T/C are bitwise symmetric (T_delta(x) == T_-delta(x+delta)), so only the half plane of displacements
// is evaluated: kernel 1 writes the T/C planes, kernel 2 box-sums them into the 169-bit masks.
// This file is compiled with -ffp-contract=off: the reference is built without FMA contraction and the
// membership test d <= tau is discrete.
#include "bcd_common.h"
#include <hip/hip_fp16.h>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int PD_TW = 64; // tile width  (one wavefront per tile row)
constexpr int PD_TH = 4;
constexpr int PD_CW = 12; // widest span of column displacements served by one staged window  // tile height (4 wavefronts per workgroup)

// LDS pixel stride (floats): D/4 odd keeps ds_read_b128 at a per-lane stride of D*4 bytes conflict-free
// IEEE-754 correctly rounded a / b without the range scaling of the compiler's sequence.  hipcc lowers an fp32 division to
//   v_div_scale x2, v_rcp, fma, fma, mul, fma, fma, fma, v_div_fmas, v_div_fixup;
// v_div_scale / v_div_fixup only act when an operand or the quotient is zero, subnormal, huge or non-finite (CDNA4 ISA,
// V_DIV_SCALE_F32).  For operands inside the guarded range below they are the identity and v_div_fmas is a plain fma, so the
// eight operations here produce bit-identical results (self-test: bcd_hip_selftest_division).  a == 0 gives +0 like a / b.
constexpr float PD_BIN_MAX = 1048576.f;   // 2^20   histogram bins must be in [0, 2^20]
constexpr float PD_N_MIN = 0.0009765625f; // 2^-10  sample counts must be in [2^-10, 2^16]
constexpr float PD_N_MAX = 65536.f;       //        => 1 < s <= 2^21, den in (2^-20, 2^53], num in {0} U [2^-70, 2^74], quotient >= 2^-123

template <bool FAST>
__device__ inline float pd_div(float a, float b)
{
    if (!FAST) return a / b;
    float y = __builtin_amdgcn_rcpf(b);
    float e = fmaf(-b, y, 1.f);
    y = fmaf(e, y, y);
    float q = a * y;
    float r = fmaf(-b, q, a);
    q = fmaf(r, y, q);
    r = fmaf(-b, q, a);
    return fmaf(r, y, q);
}

typedef float v2f __attribute__((ext_vector_type(2)));

// two divisions at once on packed operands (same eight operations per component)
template <bool FAST>
__device__ inline v2f pd_div2(v2f a, v2f b)
{
    if (!FAST) return v2f{ a.x / b.x, a.y / b.y };
    v2f y = { __builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y) };
    const v2f one = { 1.f, 1.f };
    v2f e = __builtin_elementwise_fma(-b, y, one);
    y = __builtin_elementwise_fma(e, y, y);
    v2f q = a * y;
    v2f r = __builtin_elementwise_fma(-b, q, a);
    q = __builtin_elementwise_fma(r, y, q);
    r = __builtin_elementwise_fma(-b, q, a);
    return __builtin_elementwise_fma(r, y, q);
}

__device__ inline bool pd_bin_bad(float v) { return !(v >= 0.f && v <= PD_BIN_MAX); }
__device__ inline bool pd_n_bad(float v) { return !(v >= PD_N_MIN && v <= PD_N_MAX); }

template <int D> struct PdStride { static constexpr int value = ((D / 4) % 2 == 1) ? D : D + 4; };

// ---------------------------------------------------------------------------------------------------
// kernel 1: pair-distance planes.  One thread per pixel x of a 4x64 tile; own histogram in VGPRs,
// neighbour rows staged through LDS once per displacement row dl and re-used for the 2b+1 (or b+1)
// displacements of that row.
// ---------------------------------------------------------------------------------------------------
// FAST = true: scale-free division (pd_div<true>); any staged value outside its guarded range raises *range_flag and the
// host re-runs the scale with FAST = false (the compiler's division), so results are exact either way.
// UNI = true (with FAST): every pixel has the same sample count uni_n = 2^k (the usual case: fixed samples per pixel).  Then
// n2 b1 - n1 b2 = 2^k RN(b1 - b2) and n1 n2 (b1 + b2) = 4^k RN(b1 + b2) exactly (scaling by a power of two commutes with rounding;
// on the evaluated bins b1 + b2 > 1, so nothing is subnormal), the 4^k cancels in the correctly rounded quotient, and the term is
// RN(RN(RN(b1 - b2)^2) / RN(b1 + b2)) bit for bit: three multiplications per bin less.  Every staged count is compared with uni_n;
// a mismatch raises bit 1 of *range_flag and the host re-runs the scale with the general kernel.
template <int D, bool FAST, bool UNI>
__global__ __launch_bounds__(256) void k_pairdist(const float *__restrict__ hist, const float *__restrict__ ns,
                                                   int W, int H, int b,
                                                   float *__restrict__ T, uint8_t *__restrict__ Cn, int *range_flag, float uni_n)
{
    constexpr int DS = PdStride<D>::value;
    constexpr int Q = D / 4;
    extern __shared__ float4 lds4[];
    // the displacements of a line are evaluated in chunks of at most PD_CW + 1 columns, each with its own staged column window of
    // 64 + PD_CW pixels: the LDS tile (73 KB -> two workgroups per CU) does not grow with the search radius
    const int ncols = PD_TW + min(2 * b, PD_CW);
    float *lds_n = reinterpret_cast<float *>(lds4 + PD_TH * ncols * (DS / 4));

    // a wavefront covers a compact 16 x 4 pixel patch of the 64 x 4 tile (not a 64 x 1 line): neighbouring pixels in 2-D share
    // their occupied histogram bins, which makes the wave-uniform bin skipping below bite more often
    const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
    const int tx = (wave_ << 4) | (lane_ & 15), ty = lane_ >> 4;
    // XCD-aware tile order: workgroup id i runs on XCD i % 8 (each XCD has its own 4 MiB L2); give every XCD one
    // contiguous horizontal band of tiles in row-major order, so the tiles that re-read the same neighbour rows
    // (vertical neighbours, for the 7 displacement rows) hit the same L2 instead of eight different ones.
    int tile;
    {
        const int nt = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = id & 7, k = id >> 3, q = nt >> 3, rem = nt & 7;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
    }
    const int col0 = (tile % gridDim.x) * PD_TW, row0 = (tile / gridDim.x) * PD_TH;
    const int c = col0 + tx, r = row0 + ty;
    const bool inside = (c < W) && (r < H);
    const size_t plane = (size_t)W * H;
    const size_t pix = (size_t)r * W + c;

    float h1[D];
    float n1 = 1.f;
    {
        const float4 *src = reinterpret_cast<const float4 *>(hist) + (inside ? pix * Q : 0);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            float4 v = inside ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            h1[4 * q] = v.x; h1[4 * q + 1] = v.y; h1[4 * q + 2] = v.z; h1[4 * q + 3] = v.w;
        }
        if (inside) n1 = ns[pix];
    }
    bool own_bad = false;
    if (FAST) {
        own_bad = pd_n_bad(n1);
#pragma unroll
        for (int k = 0; k < D; ++k) own_bad = own_bad || pd_bin_bad(h1[k]);
    }

    int didx = 0;
    for (int dl = 0; dl <= b; ++dl)
      for (int cbeg = (dl == 0) ? 0 : -b; cbeg <= b; cbeg += PD_CW + 1) {
        const int cend = min(b, cbeg + PD_CW);
        __syncthreads();
        // stage rows row0+dl .. row0+dl+3, columns col0+cbeg .. col0+63+cend
        const int npix = PD_TH * ncols;
        bool stage_bad = own_bad;
        bool uni_bad = UNI && inside && n1 != uni_n;
        for (int i = threadIdx.x; i < npix * Q; i += 256) {
            int p = i / Q, q = i - p * Q;
            int lr = p / ncols, lc = p - lr * ncols;
            int gr = row0 + dl + lr, gc = col0 + cbeg + lc;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < H && gc >= 0 && gc < W) v = reinterpret_cast<const float4 *>(hist)[((size_t)gr * W + gc) * Q + q];
            lds4[p * (DS / 4) + q] = v;
            if (FAST) stage_bad = stage_bad || pd_bin_bad(v.x) || pd_bin_bad(v.y) || pd_bin_bad(v.z) || pd_bin_bad(v.w);
        }
        for (int i = threadIdx.x; i < npix; i += 256) {
            int lr = i / ncols, lc = i - lr * ncols;
            int gr = row0 + dl + lr, gc = col0 + cbeg + lc;
            float nv = (gr < H && gc >= 0 && gc < W) ? ns[(size_t)gr * W + gc] : (UNI ? uni_n : 1.f);
            lds_n[i] = nv;
            if (FAST) stage_bad = stage_bad || pd_n_bad(nv);
            if (UNI) uni_bad = uni_bad || nv != uni_n;
        }
        if (FAST && stage_bad) atomicOr(range_flag, 1); // some value is outside the range where pd_div<true> is proven exact
        if (UNI && uni_bad) atomicOr(range_flag, 2);    // a sample count differs from uni_n
        __syncthreads();

        const int dc0 = cbeg;
        // neighbour histograms are double-buffered in registers: the 15 ds_read_b128 of displacement dc+1 are in
        // flight while displacement dc is evaluated
        float4 bufA[Q], bufB[Q];
        auto fetch = [&](float4 *buf, int dc) __attribute__((always_inline)) {
            const float4 *nb = lds4 + (ty * ncols + tx + dc - cbeg) * (DS / 4);
#pragma unroll
            for (int q = 0; q < Q; ++q) buf[q] = nb[q];
        };
        auto evaluate = [&](const float4 *buf, int dc, int di) __attribute__((always_inline)) {
            const float n2 = lds_n[ty * ncols + tx + dc - cbeg];
            const float n12 = n1 * n2;
            const v2f n1v = { n1, n1 }, n2v = { n2, n2 }, n12v = { n12, n12 };
            float sum = 0.f;
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                // bins 4q..4q+3 as two packed pairs: every operation below is IEEE per component (v_pk_*_f32), i.e. the same
                // roundings as the scalar code, two bins per instruction
                const v2f b2lo = { buf[q].x, buf[q].y }, b2hi = { buf[q].z, buf[q].w };
                const v2f b1lo = { h1[4 * q], h1[4 * q + 1] }, b1hi = { h1[4 * q + 2], h1[4 * q + 3] };
                const v2f slo = b1lo + b2lo, shi = b1hi + b2hi;
                const bool u0 = slo.x > 1.f, u1 = slo.y > 1.f, u2 = shi.x > 1.f, u3 = shi.y > 1.f; // DenoisingUnit.cpp:379
                // wave-uniform skip of a group of 4 bins that is empty for all 64 pixels of the wavefront (exact: skipped
                // bins contribute nothing); inside an active group the divisions are independent
                if (__builtin_amdgcn_ballot_w64(u0 || u1 || u2 || u3) != 0) {
                    v2f tlo, thi;
                    if (UNI) {
                        const v2f dlo = b1lo - b2lo, dhi = b1hi - b2hi;
                        tlo = pd_div2<FAST>(dlo * dlo, slo); thi = pd_div2<FAST>(dhi * dhi, shi);
                    } else {
                        const v2f dlo = n2v * b1lo - n1v * b2lo, dhi = n2v * b1hi - n1v * b2hi;
                        tlo = pd_div2<FAST>(dlo * dlo, n12v * slo); thi = pd_div2<FAST>(dhi * dhi, n12v * shi);
                    }
                    sum = u0 ? sum + tlo.x : sum; // bins in order: the reference's sequential sum
                    sum = u1 ? sum + tlo.y : sum;
                    sum = u2 ? sum + thi.x : sum;
                    sum = u3 ? sum + thi.y : sum;
                    cnt += (u0 ? 1 : 0) + (u1 ? 1 : 0) + (u2 ? 1 : 0) + (u3 ? 1 : 0);
                }
            }
            const int nc = c + dc, nr = r + dl;
            if (inside && nc >= 0 && nc < W && nr < H) {
                T[(size_t)di * plane + pix] = sum;
                Cn[(size_t)di * plane + pix] = (uint8_t)cnt;
            }
        };
        fetch(bufA, dc0);
        for (int dc = dc0; dc <= cend; dc += 2) {
            if (dc + 1 <= cend) fetch(bufB, dc + 1);
            evaluate(bufA, dc, didx++);
            if (dc + 1 <= cend) {
                if (dc + 2 <= cend) fetch(bufA, dc + 2);
                evaluate(bufB, dc + 1, didx++);
            }
        }
      }
}

// generic-depth variant (any D <= 255): both histograms read from global memory, no staging.
__global__ __launch_bounds__(256) void k_pairdist_generic(const float *__restrict__ hist, const float *__restrict__ ns,
                                                           int W, int H, int D, int b,
                                                           float *__restrict__ T, uint8_t *__restrict__ Cn)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= W || r >= H) return;
    const size_t plane = (size_t)W * H, pix = (size_t)r * W + c;
    const float *h1 = hist + pix * D;
    const float n1 = ns[pix];
    int didx = 0;
    for (int dl = 0; dl <= b; ++dl)
        for (int dc = (dl == 0 ? 0 : -b); dc <= b; ++dc, ++didx) {
            int nc = c + dc, nr = r + dl;
            if (nc < 0 || nc >= W || nr >= H) continue;
            size_t q = (size_t)nr * W + nc;
            const float *h2 = hist + q * D;
            float n2 = ns[q], n12 = n1 * n2, sum = 0.f;
            int cnt = 0;
            for (int k = 0; k < D; ++k) {
                float b1 = h1[k], b2 = h2[k], s = b1 + b2;
                if (s <= 1.f) continue;
                ++cnt;
                float diff = n2 * b1 - n1 * b2;
                sum += diff * diff / (n12 * s);
            }
            T[(size_t)didx * plane + pix] = sum;
            Cn[(size_t)didx * plane + pix] = (uint8_t)cnt;
        }
}

// ---------------------------------------------------------------------------------------------------
// kernel 2: similarity masks.  Thread = (pixel, mask word).  For each window offset the 9 (or (2w+1)^2)
// T values of the canonical (half-plane) displacement are added in the reference's patch order, the
// integer counts likewise, then one IEEE division and the <= tau test.
// The window is PixelWindow(center, radius b, border w) (DenoisingUnit.cpp:200-203): clipped.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_masks(const float *__restrict__ T, const uint8_t *__restrict__ Cn,
                                               int W, int H, int w, int b, float tau, int words,
                                               uint32_t *__restrict__ mask)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int r = blockIdx.y;
    const int word = blockIdx.z * blockDim.y + threadIdx.y;
    if (c >= W || word >= words) return;
    const size_t plane = (size_t)W * H, pix = (size_t)r * W + c;
    const int side = 2 * b + 1, nbits = side * side;
    uint32_t m = 0;
    const bool is_main = (r >= w && r <= H - 1 - w && c >= w && c <= W - 1 - w);
    if (is_main) {
        for (int bit = 0; bit < 32; ++bit) {
            int k = word * 32 + bit;
            if (k >= nbits) break;
            int dl = k / side - b, dc = k % side - b;
            int qr = r + dl, qc = c + dc;
            if (qr < w || qr > H - 1 - w || qc < w || qc > W - 1 - w) continue;
            // canonical displacement and base pixel
            int br = r, bc = c, cl = dl, cc = dc;
            if (dl < 0 || (dl == 0 && dc < 0)) { br = qr; bc = qc; cl = -dl; cc = -dc; }
            const size_t base = (size_t)bcd_delta_index(cl, cc, b) * plane;
            float s = 0.f;
            int n = 0;
            for (int ol = -w; ol <= w; ++ol)
                for (int oc = -w; oc <= w; ++oc) {
                    size_t idx = base + (size_t)(br + ol) * W + (bc + oc);
                    s += T[idx];
                    n += Cn[idx];
                }
            float d = s / (float)n; // 0/0 = NaN -> not similar
            if (d <= tau) m |= 1u << bit;
        }
    }
    mask[pix * words + word] = m;
}

// ---------------------------------------------------------------------------------------------------
// kernel 2 (w = 1): forward similarity bits.  One wavefront covers 62 consecutive pixels of a line (lanes 0 and 63
// only feed their neighbours); per displacement plane every lane loads T and C of its own column on the three patch
// lines and obtains the left / right columns from the adjacent lanes, so a 3x3 box sum costs 3 loads instead of 9.
// The nine values are added in the reference's patch order (row-major), the counts as integers, then one IEEE
// division and the <= tau test (DenoisingUnit.cpp:336-358, 209).  Output: ceil(ndelta/32) word PLANES (fwd[word][pixel], round 6), bit i =
// similar(p, p + delta_i) for the half-plane displacements; 0 when p or p + delta is not a main pixel.
// ---------------------------------------------------------------------------------------------------
__device__ inline float lane_up(float v)   { return __shfl_up(v, 1); }   // value of lane - 1
__device__ inline float lane_down(float v) { return __shfl_down(v, 1); } // value of lane + 1
__device__ inline int lane_up(int v)       { return __shfl_up(v, 1); }
__device__ inline int lane_down(int v)     { return __shfl_down(v, 1); }

// One wavefront: 62 columns x FWD_RB lines of one 32-displacement word.  Per displacement the FWD_RB + 2 lines of the T / C
// planes are loaded once and shared by the lines of the strip (1.5 loads per output instead of 3); the nine T values are
// summed in the reference's order (patch line by line, left to right), the counts are integers.
constexpr int FWD_RB = 4;
// APPROX: the planes come from k_pairdist_rw (k_similarity_fast.hip); tau is then tau (1 - delta), pairs up to bl.tau_hi are
// appended to the borderline list (their bit is decided by k_verify_pairs)
// The borderline pairs of a wavefront are collected as bit words (one bit per displacement, like the forward words) and appended at
// the end of the kernel with ONE atomic on the list counter per wavefront: a counter serves ~50 same-address atomics per microsecond,
// and a textured 1080p frame has 2e5 borderline pairs at scale 0 (one atomic per pair: 2.2 ms in this kernel, r3 profile).
__device__ inline int wave_exclusive_sum(int v, int lane, int *total)
{
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    *total = __shfl(incl, 63);
    return incl - v;
}
template <int NW>
__device__ inline void borderline_flush(const BcdBorderline &bl, const uint32_t (&bword)[NW], const uint32_t (&pix)[NW], int word_index, int lane)
{
    int mine = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) mine += __popc(bword[i]);
    if (__ballot(mine != 0) == 0) return; // (the usual case: nothing near the threshold)
    int total = 0;
    const int before = wave_exclusive_sum(mine, lane, &total);
    int base = 0;
    if (lane == 0) base = atomicAdd(bl.counter, total);
    base = __shfl(base, 0);
    int slot = base + before;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint32_t m = bword[i];
        while (m) {
            const int bit = __ffs(m) - 1;
            m &= m - 1;
            if (slot < bl.capacity) bl.list[slot] = make_uint2(pix[i], (uint32_t)(32 * word_index + bit));
            ++slot;
        }
    }
}

// plane element type: the approximate planes hold T in binary16 (bcd_common.h), the exact ones in fp32
template <bool APPROX> struct PlaneT { typedef float type; };
template <> struct PlaneT<true> { typedef __half type; };
__device__ inline float plane_value(float v) { return v; }
__device__ inline float plane_value(__half v) { return __half2float(v); }

template <bool APPROX>
__global__ __launch_bounds__(64) void k_fwd_masks_w1(const typename PlaneT<APPROX>::type *__restrict__ T, const uint8_t *__restrict__ Cn,
                                                     int W, int H, int b, float tau, int fwords, int nd,
                                                     uint32_t *__restrict__ fwd, BcdBorderline bl, int rb0 /* first block of FWD_RB lines */, int row_end /* lines written: < row_end */)
{
    const int lane = threadIdx.x;
    const int c = blockIdx.x * 62 + lane - 1; // lanes 1..62 produce output
    const int rb = (blockIdx.y + rb0) * FWD_RB;
    const int wi = blockIdx.z;
    const bool writer = lane >= 1 && lane <= 62 && c < W;
    const bool c_main = c >= 1 && c <= W - 2;
    const size_t plane = (size_t)W * H;
    const int cc = min(max(c, 0), W - 1), side = 2 * b + 1;
    // lines rb-1 .. rb+FWD_RB clamped for the loads (values of non-main pixels are never used)
    size_t off[FWD_RB + 2];
#pragma unroll
    for (int i = 0; i < FWD_RB + 2; ++i) off[i] = (size_t)min(max(rb - 1 + i, 0), H - 1) * W + cc;
    uint32_t word[FWD_RB], bword[FWD_RB];
#pragma unroll
    for (int i = 0; i < FWD_RB; ++i) { word[i] = 0; bword[i] = 0; }
    const int d_end = min(nd, 32 * wi + 32);
    // (the displacement of the first plane by one division, the following ones by stepping: dc + 1, wrapping into the next displacement line)
    int dl = 0, dc = 32 * wi;
    if (dc > b) { int e = dc - (b + 1); dl = 1 + e / side; dc = e - (dl - 1) * side - b; }
    for (int didx = 32 * wi; didx < d_end; ++didx, dc = (dc == b ? -b : dc + 1), dl += (dc == -b ? 1 : 0)) {
        const typename PlaneT<APPROX>::type *Tp = T + (size_t)didx * plane;
        const uint8_t *Cp = Cn + (size_t)didx * plane;
        float tc[FWD_RB + 2], tl[FWD_RB + 2], tr[FWD_RB + 2];
        int nh[FWD_RB + 2];
#pragma unroll
        for (int i = 0; i < FWD_RB + 2; ++i) { tc[i] = plane_value(Tp[off[i]]); nh[i] = Cp[off[i]]; }
#pragma unroll
        for (int i = 0; i < FWD_RB + 2; ++i) {
            tl[i] = lane_up(tc[i]); tr[i] = lane_down(tc[i]);
            nh[i] = lane_up(nh[i]) + nh[i] + lane_down(nh[i]);
        }
        const int qc = c + dc;
        const bool cols_ok = c_main && qc >= 1 && qc <= W - 2;
#pragma unroll
        for (int i = 0; i < FWD_RB; ++i) {
            float s = tl[i];
            s += tc[i]; s += tr[i];
            s += tl[i + 1]; s += tc[i + 1]; s += tr[i + 1];
            s += tl[i + 2]; s += tc[i + 2]; s += tr[i + 2];
            const int n = nh[i] + nh[i + 1] + nh[i + 2];
            const int r = rb + i;
            if (cols_ok && r >= 1 && r <= H - 2 && r + dl <= H - 2) {
                if (APPROX) {
                    // no division on the approximate planes: s <= tau' n differs from s / n <= tau' by an ulp, which the band around tau
                    // absorbs (its half-width is 2^-10, the planes' error 5e-4); n == 0 (0 / 0 in the reference) is never similar
                    const float fn = (float)n;
                    if (n > 0 && s <= tau * fn) word[i] |= 1u << (didx & 31);
                    else if (n > 0 && writer && s <= bl.tau_hi * fn) bword[i] |= 1u << (didx & 31);
                } else {
                    const float d = s / (float)n; // 0/0 = NaN -> not similar
                    if (d <= tau) word[i] |= 1u << (didx & 31);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < FWD_RB; ++i)
        if (writer && rb + i < row_end) fwd[(size_t)wi * plane + (size_t)(rb + i) * W + c] = word[i];
    if (APPROX) {
        uint32_t pix[FWD_RB];
#pragma unroll
        for (int i = 0; i < FWD_RB; ++i) { pix[i] = (uint32_t)((rb + i) * W + c); if (rb + i >= row_end) bword[i] = 0; } // (lines of a later launch)
        borderline_flush<FWD_RB>(bl, bword, pix, wi, lane);
    }
}

// The same with four columns per lane (image widths that are multiples of 4), exact fp32 planes: 16-byte plane loads instead of 4-byte ones.  A
// wavefront covers 256 columns, of which the 248 of lanes 1..62 are produced (the outer lanes only supply the halo).  (The approximate planes
// have their own kernel below.)
__global__ __launch_bounds__(64) void k_fwd_masks_w1v4(const float *__restrict__ T, const uint8_t *__restrict__ Cn,
                                                       int W, int H, int b, float tau, int fwords, int nd,
                                                       uint32_t *__restrict__ fwd, int rb0, int row_end)
{
    const int lane = threadIdx.x;
    const int c = blockIdx.x * 248 - 4 + 4 * lane; // first of the lane's four columns
    const int rb = (blockIdx.y + rb0) * FWD_RB;
    const int wi = blockIdx.z;
    const bool writer = lane >= 1 && lane <= 62 && c < W;
    const size_t plane = (size_t)W * H;
    const int cc = min(max(c, 0), W - 4), side = 2 * b + 1;
    size_t off[FWD_RB + 2];
#pragma unroll
    for (int i = 0; i < FWD_RB + 2; ++i) off[i] = (size_t)min(max(rb - 1 + i, 0), H - 1) * W + cc;
    uint32_t word[FWD_RB][4];
#pragma unroll
    for (int i = 0; i < FWD_RB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) word[i][j] = 0;
    const int d_end = min(nd, 32 * wi + 32);
    // (the displacement of the first plane by one division, the following ones by stepping: dc + 1, wrapping into the next displacement line)
    int dl = 0, dc = 32 * wi;
    if (dc > b) { int e = dc - (b + 1); dl = 1 + e / side; dc = e - (dl - 1) * side - b; }
    for (int didx = 32 * wi; didx < d_end; ++didx, dc = (dc == b ? -b : dc + 1), dl += (dc == -b ? 1 : 0)) {
        const float *Tp = T + (size_t)didx * plane;
        const uint8_t *Cp = Cn + (size_t)didx * plane;
        float t[FWD_RB + 2][6]; // left neighbour, the lane's four columns, right neighbour
        int nh[FWD_RB + 2][4];  // horizontal 3-sums of the counts
#pragma unroll
        for (int i = 0; i < FWD_RB + 2; ++i) {
            const float4 v = *reinterpret_cast<const float4 *>(Tp + off[i]);
            const uint32_t cw = *reinterpret_cast<const uint32_t *>(Cp + off[i]);
            t[i][1] = v.x; t[i][2] = v.y; t[i][3] = v.z; t[i][4] = v.w;
            const int n0 = cw & 255, n1 = (cw >> 8) & 255, n2 = (cw >> 16) & 255, n3 = cw >> 24;
            t[i][0] = lane_up(v.w); t[i][5] = lane_down(v.x);
            const int nl = lane_up(n3), nr = lane_down(n0);
            nh[i][0] = nl + n0 + n1; nh[i][1] = n0 + n1 + n2; nh[i][2] = n1 + n2 + n3; nh[i][3] = n2 + n3 + nr;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cj = c + j, qc = cj + dc;
            const bool cols_ok = cj >= 1 && cj <= W - 2 && qc >= 1 && qc <= W - 2;
#pragma unroll
            for (int i = 0; i < FWD_RB; ++i) {
                float s = t[i][j];
                s += t[i][j + 1]; s += t[i][j + 2];
                s += t[i + 1][j]; s += t[i + 1][j + 1]; s += t[i + 1][j + 2];
                s += t[i + 2][j]; s += t[i + 2][j + 1]; s += t[i + 2][j + 2];
                const int n = nh[i][j] + nh[i + 1][j] + nh[i + 2][j];
                const int r = rb + i;
                if (cols_ok && r >= 1 && r <= H - 2 && r + dl <= H - 2) {
                    const float d = s / (float)n; // 0/0 = NaN -> not similar
                    if (d <= tau) word[i][j] |= 1u << (didx & 31);
                }
            }
        }
    }
    if (writer) {
#pragma unroll
        for (int i = 0; i < FWD_RB; ++i)
            if (rb + i < row_end) {
#pragma unroll
                for (int j = 0; j < 4; ++j) fwd[(size_t)wi * plane + (size_t)(rb + i) * W + c + j] = word[i][j];
            }
    }
}

// Four columns per lane on the APPROXIMATE planes (binary16 T, k_similarity_fast.hip), without a branch in the displacement loop (round 6).  The
// form it replaced (the kernel above with binary16 loads and the borderline test) spent ~58 instructions per output -- each of its 16 outputs per
// plane behind three nested exec-mask branches -- and stored its words one by one into a pixel-major buffer, 64 scattered 4-byte writes per store
// instruction: those partial-line writes alone were 1.2 of its 3.0 ms on the 312 planes of a 3840x2160 scale (measured by leaving them out).  The
// forward words now live in word planes (fwd[word][pixel]: a lane's four columns are one 16-byte store, a wavefront's line one kilobyte) and an
// output costs 7 vector instructions: 3.0 -> 1.86 ms there (4.2 TB/s of plane reads), 0.19 -> 0.13 ms at 1920x1080, b = 6.
//   * the decision is the SIGN of y = tau' n - s, one fma, shifted into a 32-plane accumulator with one v_alignbit (bit = 1: not similar); two
//     accumulators per output, lo (tau' = tau (1 - delta)) and hi (tau (1 + delta)); the borderline word is hi & ~lo after the loop.  The fma
//     compares s with the unrounded product where the kernel above compares with RN(tau' n): at most an ulp of the threshold, 6e-8 of the
//     band's 2^-10, like the missing division;
//   * n == 0 (never similar, DenoisingUnit.cpp:351-357: 0 / 0; then s == 0 too) without a test: the sum is formed as -s - FLT_MIN, which is -s
//     for s > 0 and -FLT_MIN for s == 0, so y < 0 for n == 0 and y > 0 for identical histograms with n > 0;
//   * image-border validity (main pixels only, p + delta a main pixel) only in wavefronts that touch the border (template EDGE): an invalid
//     output gets y = -1 (its plane entries were never written and may hold anything, NaN included);
//   * the 3-sums of the counts on packed bytes (3 x 60 fits a byte), their vertical sums on packed 16-bit fields;
//   * lane +-1 values by DPP wavefront shifts instead of ds_bpermute;
//   * buffer loads: the plane's base in a scalar resource, 32-bit per-lane offsets that never change -- no vector address arithmetic;
//   * two planes per trip with two register sets: the six lines of plane d + 1 are requested before plane d is evaluated (a word with an
//     odd number of planes evaluates its last plane twice and drops the extra bit).
__device__ inline uint32_t wave_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true); } // value of lane - 1
__device__ inline uint32_t wave_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true); } // value of lane + 1
__device__ inline float wave_prev(float v) { return __uint_as_float(wave_prev(__float_as_uint(v))); }
__device__ inline float wave_next(float v) { return __uint_as_float(wave_next(__float_as_uint(v))); }

typedef unsigned int fwd_u32x2 __attribute__((ext_vector_type(2)));
struct FwdRaw { fwd_u32x2 t[FWD_RB + 2]; uint32_t c[FWD_RB + 2]; };
constexpr float FWD_NEG_TINY = -1.17549435e-38f; // -FLT_MIN

template <bool EDGE>
__device__ inline void fwd_v4a_loop(const __half *__restrict__ T, const uint8_t *__restrict__ Cn, size_t plane, const uint32_t (&off)[FWD_RB + 2],
                                    int W, int H, int b, int c, int rb, float tau_lo, float tau_hi, int d_begin, int trips,
                                    int d_last, uint32_t (&alo)[FWD_RB * 4], uint32_t (&ahi)[FWD_RB * 4])
{
    const int side = 2 * b + 1;
    auto load = [&](int didx, FwdRaw &x) __attribute__((always_inline)) {
        // (raw buffers: byte offsets, no stride; the range check is not used -- every offset lies inside the plane)
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half *>(T + (size_t)didx * plane), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(Cn + (size_t)didx * plane), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < FWD_RB + 2; ++i) {
            x.t[i] = __builtin_amdgcn_raw_buffer_load_b64(rt, 2u * off[i], 0, 0);
            x.c[i] = __builtin_amdgcn_raw_buffer_load_b32(rc, off[i], 0, 0);
        }
    };
    auto eval = [&](const FwdRaw &x, int dl, int dc) __attribute__((always_inline)) {
        float hs[FWD_RB + 2][4];
        uint32_t ev[FWD_RB + 2], od[FWD_RB + 2];
#pragma unroll
        for (int i = 0; i < FWD_RB + 2; ++i) {
            const uint32_t t01 = x.t[i].x, t23 = x.t[i].y;
            const __half2 h01 = *reinterpret_cast<const __half2 *>(&t01), h23 = *reinterpret_cast<const __half2 *>(&t23);
            const float2 f01 = __half22float2(h01), f23 = __half22float2(h23);
            const float tl = wave_prev(f23.y), tr = wave_next(f01.x);
            hs[i][0] = (tl + f01.x) + f01.y; hs[i][1] = (f01.x + f01.y) + f23.x; hs[i][2] = (f01.y + f23.x) + f23.y; hs[i][3] = (f23.x + f23.y) + tr;
            const uint32_t cw = x.c[i], lw = wave_prev(cw), rw = wave_next(cw);
            // bytes of cw = counts of the lane's columns 0..3; the same shifted by one column either way; 3 x 60 < 256: no carry between bytes
            const uint32_t h3 = cw + __builtin_amdgcn_alignbyte(cw, lw, 3) + __builtin_amdgcn_alignbyte(rw, cw, 1);
            ev[i] = h3 & 0x00ff00ffu;         // columns 0 and 2 as 16-bit fields
            od[i] = (h3 >> 8) & 0x00ff00ffu;  // columns 1 and 3
        }
        // (EDGE) columns: c + j and c + j + dc main pixels <=> c + j in [clo, clo + cspan]; lines: r and r + dl main pixels
        const int clo = max(1, 1 - dc), cspan = min(W - 2, W - 2 - dc) - clo;
#pragma unroll
        for (int i = 0; i < FWD_RB; ++i) {
            const uint32_t e3 = ev[i] + ev[i + 1] + ev[i + 2], o3 = od[i] + od[i + 1] + od[i + 2];
            const int r = rb + i;
            const bool row_ok = r >= 1 && r <= H - 2 && r + dl <= H - 2; // (wave-uniform; EDGE only)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t n = (j & 1) ? ((j & 2) ? (o3 >> 16) : (o3 & 0xffffu)) : ((j & 2) ? (e3 >> 16) : (e3 & 0xffffu));
                const float fn = (float)n;
                const float ms = (FWD_NEG_TINY - (hs[i][j] + hs[i + 1][j])) - hs[i + 2][j]; // -s, or -FLT_MIN when s == 0
                float ylo = fmaf(tau_lo, fn, ms), yhi = fmaf(tau_hi, fn, ms);
                if (EDGE) { // (plane entries of pairs that leave the image are never written: whatever they hold must not reach the sign)
                    const bool valid = row_ok && cspan >= 0 && (uint32_t)(c + j - clo) <= (uint32_t)cspan;
                    ylo = valid ? ylo : -1.f; yhi = valid ? yhi : -1.f;
                }
                alo[i * 4 + j] = __builtin_amdgcn_alignbit(alo[i * 4 + j], __float_as_uint(ylo), 31);
                ahi[i * 4 + j] = __builtin_amdgcn_alignbit(ahi[i * 4 + j], __float_as_uint(yhi), 31);
            }
        }
    };
    int dl = 0, dc = d_begin;
    if (dc > b) { int e = dc - (b + 1); dl = 1 + e / side; dc = e - (dl - 1) * side - b; }
    FwdRaw A, B;
    load(d_begin, A);
    // (the loop-carried set must not look like a plain load to the optimiser: it folds a phi of loads into a load of a phi of addresses, i.e. it
    // moves the request for plane d + 2 behind the evaluation of plane d + 1)
#pragma unroll
    for (int i = 0; i < FWD_RB + 2; ++i) asm volatile("" : "+v"(A.t[i]), "+v"(A.c[i]));
    int didx = d_begin;
    for (int t = 0; t < trips; ++t) {
        load(min(didx + 1, d_last), B);
        __builtin_amdgcn_sched_barrier(0); // (the scheduler would move the requests down to the end of the evaluation that is to hide them)
        eval(A, dl, dc);
        dc = (dc == b ? -b : dc + 1); dl += (dc == -b ? 1 : 0);
        load(min(didx + 2, d_last), A);
        __builtin_amdgcn_sched_barrier(0);
        eval(B, dl, dc);
        dc = (dc == b ? -b : dc + 1); dl += (dc == -b ? 1 : 0);
        didx += 2;
    }
}

__global__ __launch_bounds__(64) void k_fwd_masks_w1v4a(const __half *__restrict__ T, const uint8_t *__restrict__ Cn, int W, int H, int b, float tau_lo, int fwords,
                                                        int nd, uint32_t *__restrict__ fwd, BcdBorderline bl, int rb0, int row_end)
{
    const int lane = threadIdx.x;
    const int xb = blockIdx.x;
    const int rb = (blockIdx.y + rb0) * FWD_RB;
    const int c = xb * 248 - 4 + 4 * lane; // first of the lane's four columns
    const int wi = blockIdx.z;
    const bool writer = lane >= 1 && lane <= 62 && c < W;
    const size_t plane = (size_t)W * H;
    const int cc = min(max(c, 0), W - 4);
    uint32_t off[FWD_RB + 2]; // (element offsets inside a plane: 32 bits -- the caller admits frames below 2^30 pixels)
#pragma unroll
    for (int i = 0; i < FWD_RB + 2; ++i) off[i] = (uint32_t)(min(max(rb - 1 + i, 0), H - 1) * W + cc);
    uint32_t alo[FWD_RB * 4], ahi[FWD_RB * 4];
#pragma unroll
    for (int i = 0; i < FWD_RB * 4; ++i) { alo[i] = ~0u; ahi[i] = ~0u; }
    const int d_begin = 32 * wi, k = min(nd, d_begin + 32) - d_begin, trips = (k + 1) >> 1; // 1 <= k <= 32 planes, evaluated two per trip
    // does this wavefront produce a pixel whose window leaves the main pixels?  (columns blockIdx.x * 248 .. + 247, lines rb .. rb + 3)
    const int cfirst = xb * 248, clast = min(cfirst + 247, W - 1);
    const bool edge = cfirst - b < 1 || clast + b > W - 2 || rb < 1 || rb + FWD_RB - 1 + b > H - 2;
    if (edge) fwd_v4a_loop<true>(T, Cn, plane, off, W, H, b, c, rb, tau_lo, bl.tau_hi, d_begin, trips, d_begin + k - 1, alo, ahi);
    else fwd_v4a_loop<false>(T, Cn, plane, off, W, H, b, c, rb, tau_lo, bl.tau_hi, d_begin, trips, d_begin + k - 1, alo, ahi);
    // accumulator: plane d_begin + i at bit 2 trips - 1 - i, 1 = not similar  ->  word: plane i at bit i, 1 = similar, nothing beyond the k planes
    const int sh = 32 - 2 * trips;                          // (0 .. 30)
    const uint32_t keep = k == 32 ? ~0u : ((1u << k) - 1u);
    uint32_t bword[FWD_RB * 4], pix[FWD_RB * 4];
#pragma unroll
    for (int i = 0; i < FWD_RB; ++i) {
        const bool mine = writer && rb + i < row_end; // (lines of a later launch are that launch's)
        uint32_t wl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t wlo = (__brev(~alo[i * 4 + j]) >> sh) & keep, whi = (__brev(~ahi[i * 4 + j]) >> sh) & keep;
            wl[j] = wlo;
            pix[i * 4 + j] = (uint32_t)((rb + i) * W + c + j);
            bword[i * 4 + j] = mine ? (whi & ~wlo) : 0u;
        }
        // word-plane layout: the lane's four columns are 16 contiguous bytes (W % 4 == 0, c % 4 == 0), a wavefront's line one kilobyte
        if (mine) *reinterpret_cast<uint4 *>(fwd + (size_t)wi * plane + (size_t)(rb + i) * W + c) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
    }
    borderline_flush<FWD_RB * 4>(bl, bword, pix, wi, lane);
}

// kernel 3: full (2b+1)^2-bit masks and |S| from the forward bits: bit(p, -delta) = bit(p - delta, +delta)
__global__ __launch_bounds__(256) void k_sym_masks(const uint32_t *__restrict__ fwd, int W, int H, int b, int fwords, int words,
                                                   uint32_t *__restrict__ mask, int32_t *__restrict__ count)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= W || r >= H) return;
    const int side = 2 * b + 1;
    const size_t pix = (size_t)r * W + c;
    uint32_t out[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) out[j] = 0;
    int n = 0;
    for (int dl = -b; dl <= b; ++dl)
        for (int dc = -b; dc <= b; ++dc) {
            const bool forward = dl > 0 || (dl == 0 && dc >= 0);
            const int br = forward ? r : r + dl, bc = forward ? c : c + dc; // pixel holding the forward bit
            if (br < 0 || bc < 0 || bc >= W) continue;
            const int di = bcd_delta_index(forward ? dl : -dl, forward ? dc : -dc, b);
            const uint32_t wv = fwd[(size_t)(di >> 5) * W * H + (size_t)br * W + bc];
            if ((wv >> (di & 31)) & 1u) {
                const int k = (dl + b) * side + (dc + b);
                out[k >> 5] |= 1u << (k & 31);
                ++n;
            }
        }
    for (int j = 0; j < words; ++j) mask[pix * words + j] = out[j];
    count[pix] = n;
}

// kernel 3, compile-time search radius (round 6: rolling form).  In mask order the forward half-plane is the pixel's own bits moved up by
// KC = ((2B+1)^2-1)/2 (bit KC + i <- forward bit i, a funnel shift of the words), and the backward half is bit KC - i <- forward bit i of p - delta_i.
// A workgroup walks a 64-column strip downwards, four lines per step, with the forward words of the last B + 4 lines in an LDS ring of B + 8 slots
// (line g in slot g mod (B + 8); a slot = FW segments of 64 + 2B words, one per word plane): a step's four new lines are the only loads (1.4 cells per
// pixel incl. the column halo, x (rows + B) / rows for the lines above the chunk), and they travel by LDS-DMA (global_load_lds_dword: lane l lands
// at base + 4 l, which is the ring's layout) into the four free slots while the step's pixels are being assembled.  The tile form it replaced
// (k_sym_masks_t: (4 + B) x (64 + 2B) cells staged per 4 x 64 tile, load -> barrier -> compute) waited on its global loads: 1.46 -> 0.41 ms for the
// 3840x2160 scale at B = 12, 38 -> 26 us for the 1920x1080 scale at B = 6.
__device__ const uint32_t g_sym_zero = 0u; // source of the LDS-DMA lanes whose cell lies outside the image
template <int B>
__global__ __launch_bounds__(256) void k_sym_masks_roll(const uint32_t *__restrict__ fwd, int W, int H, int chunk_rows /* multiple of 4 */,
                                                        uint32_t *__restrict__ mask, int32_t *__restrict__ count)
{
    constexpr int side = 2 * B + 1, ND = (B + 1) + B * side, FW = (ND + 31) / 32, WORDS = (side * side + 31) / 32, KC = ND - 1;
    constexpr int TC = 64 + 2 * B, LW = TC * FW, NSLOT = B + 8, NDMA = (LW + 255) / 256;
    extern __shared__ uint32_t s_ring[]; // [NSLOT][FW][TC]
    const int64_t npix = (int64_t)W * H;
    const int lx = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = blockIdx.x * 64, r_begin = blockIdx.y * chunk_rows, r_end = min(H, r_begin + chunk_rows);
    // line g (any g >= -B; lines outside the image are zeros) on its way into its slot
    auto stage = [&](int g) __attribute__((always_inline)) {
        uint32_t *slot = s_ring + ((g + NSLOT) % NSLOT) * LW;
        const bool row_in = g >= 0 && g < H;
        const int64_t line0 = (int64_t)g * W + (c0 - B); // pixel index of the segment's first cell (may lie before the line's start: those cells are not read)
#pragma unroll
        for (int u = 0; u < NDMA; ++u) {
            const int i = threadIdx.x + u * 256;
            const int j = i / TC, cell = i - j * TC, gc = c0 - B + cell; // (word plane j, cell)
            const uint32_t *src = (row_in && gc >= 0 && gc < W) ? fwd + ((int64_t)j * npix + line0 + cell) : &g_sym_zero;
            if (i < LW)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(slot + wave * 64 + u * 256), 4, 0, 0);
        }
    };
    for (int g = r_begin - B; g < r_begin + 4; ++g) stage(g);
    const int c = c0 + lx;
    for (int r0 = r_begin; r0 < r_end; r0 += 4) {
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): this wavefront's LDS-DMA (and its stores) are done -- hipcc does not count the DMA
        __syncthreads();                    // ... and everybody's: the lines r0 - B .. r0 + 3 are in the ring, the step before has read what it needed
        if (r0 + 4 < r_end)
            for (int g = r0 + 4; g < r0 + 8; ++g) stage(g); // (slots of the lines r0 - B - 4 .. r0 - B - 1)
        const int r = r0 + wave;
        if (c < W && r < H) {
            uint32_t out[WORDS];
#pragma unroll
            for (int j = 0; j < WORDS; ++j) out[j] = 0;
            const int s_own = r % NSLOT; // (wave-uniform)
            const uint32_t *own = s_ring + s_own * LW + lx + B;
#pragma unroll
            for (int j = 0; j < FW; ++j) {
                const uint32_t v = own[j * TC];
                const int pos = KC + 32 * j, wd = pos >> 5, sh = pos & 31;
                out[wd] |= v << sh;
                if (sh != 0 && wd + 1 < WORDS) out[wd + 1] |= v >> (32 - sh);
            }
#pragma unroll
            for (int dl = 0; dl <= B; ++dl) {
                int s = s_own - dl;
                if (s < 0) s += NSLOT;
                const uint32_t *line = s_ring + s * LW + lx + B;
#pragma unroll
                for (int dc = -B; dc <= B; ++dc) {
                    if (dl == 0 && dc <= 0) continue;
                    const int di = dl == 0 ? dc : (B + 1) + (dl - 1) * side + (dc + B);
                    const uint32_t wv = line[(di >> 5) * TC - dc];
                    const int k = KC - di;
                    out[k >> 5] |= ((wv >> (di & 31)) & 1u) << (k & 31);
                }
            }
            const size_t pix = (size_t)r * W + c;
            int n = 0;
#pragma unroll
            for (int j = 0; j < WORDS; ++j) { mask[pix * WORDS + j] = out[j]; n += __popc(out[j]); }
            count[pix] = n;
        }
    }
}

__global__ void k_mask_count(const uint32_t *__restrict__ mask, int64_t npix, int words, int32_t *__restrict__ count)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    int n = 0;
    for (int j = 0; j < words; ++j) n += __popc(mask[p * words + j]);
    count[p] = n;
}

// debug / parity: raw distances of one main pixel to its window
__global__ void k_window_distances(const float *__restrict__ T, const uint8_t *__restrict__ Cn,
                                   int W, int H, int w, int b, int r, int c, float *__restrict__ out)
{
    const int side = 2 * b + 1;
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= side * side) return;
    const size_t plane = (size_t)W * H;
    int dl = k / side - b, dc = k % side - b;
    int qr = r + dl, qc = c + dc;
    float d = INFINITY;
    if (!(qr < w || qr > H - 1 - w || qc < w || qc > W - 1 - w)) {
        int br = r, bc = c, cl = dl, cc = dc;
        if (dl < 0 || (dl == 0 && dc < 0)) { br = qr; bc = qc; cl = -dl; cc = -dc; }
        const size_t base = (size_t)bcd_delta_index(cl, cc, b) * plane;
        float s = 0.f;
        int n = 0;
        for (int ol = -w; ol <= w; ++ol)
            for (int oc = -w; oc <= w; ++oc) {
                size_t idx = base + (size_t)(br + ol) * W + (bc + oc);
                s += T[idx];
                n += Cn[idx];
            }
        d = s / (float)n;
    }
    out[k] = d;
}

// self-test of pd_div<true> against the compiler's IEEE division on hashed operands drawn from the guarded range
__global__ void k_selftest_div(uint32_t seed, int per_thread, unsigned long long *mismatches)
{
    uint32_t st = bcd_mix32(seed ^ (blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B1u);
    unsigned bad = 0;
    for (int i = 0; i < per_thread; ++i) {
        st = bcd_mix32(st + 0x6D2B79F5u); uint32_t ma = st;
        st = bcd_mix32(st + 0x6D2B79F5u); uint32_t mb = st;
        st = bcd_mix32(st + 0x6D2B79F5u); uint32_t ex = st;
        // random mantissas (every 16th: all-ones / all-zeros patterns), exponents spanning the guarded range
        uint32_t fa = (ma & 0x7fffffu), fb = (mb & 0x7fffffu);
        if ((ex & 0xf00000u) == 0) { fa = (ex & 1) ? 0x7fffffu : 0u; }
        if ((ex & 0x0f0000u) == 0) { fb = (ex & 2) ? 0x7fffffu : 0x7ffffeu; }
        int ea = 127 - 70 + (int)((ex & 0xffu) % 145u);        // 2^-70 .. 2^74
        int eb = 127 - 20 + (int)(((ex >> 8) & 0xffu) % 74u);  // 2^-20 .. 2^53
        float a = __int_as_float((uint32_t)ea << 23 | fa), b = __int_as_float((uint32_t)eb << 23 | fb);
        if ((ex >> 28) == 0) a = 0.f;
        float q1 = pd_div<true>(a, b), q2 = a / b;
        if (__float_as_int(q1) != __float_as_int(q2)) ++bad;
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

} // namespace

hipError_t bcd_launch_selftest_div(uint32_t seed, int blocks, int per_thread, unsigned long long *d_mismatches, hipStream_t st)
{
    hipLaunchKernelGGL(k_selftest_div, dim3(blocks), dim3(256), 0, st, seed, per_thread, d_mismatches);
    return hipGetLastError();
}

// ---- launchers (called from bcd_api.hip) --------------------------------------------------------------
size_t bcd_pairdist_lds_bytes(int D, int b)
{
    int DS = ((D / 4) % 2 == 1) ? D : D + 4;
    return (size_t)PD_TH * (PD_TW + (2 * b < PD_CW ? 2 * b : PD_CW)) * (DS + 1) * sizeof(float);
}

// are all sample counts bitwise equal to the first one?  (out[0] |= 1 if not; the host looks at ns[0] itself)
static __global__ void k_uniform_n(const float *__restrict__ ns, int64_t npix, int *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool diff = i < npix && __float_as_uint(ns[i]) != __float_as_uint(ns[0]);
    if (__builtin_amdgcn_ballot_w64(diff) != 0 && (threadIdx.x & 63) == 0) atomicOr(out, 1);
}

// self-test: number of entries where two sets of T / C planes differ bitwise
static __global__ void k_compare_planes(const float *__restrict__ Ta, const uint8_t *__restrict__ Ca, const float *__restrict__ Tb,
                                        const uint8_t *__restrict__ Cb, int64_t n, unsigned long long *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool diff = i < n && (__float_as_uint(Ta[i]) != __float_as_uint(Tb[i]) || Ca[i] != Cb[i]);
    unsigned long long bal = __builtin_amdgcn_ballot_w64(diff);
    if (bal != 0 && (threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)__popcll(bal));
}

hipError_t bcd_launch_compare_planes(const float *Ta, const uint8_t *Ca, const float *Tb, const uint8_t *Cb, int64_t n, unsigned long long *out,
                                     hipStream_t st)
{
    hipLaunchKernelGGL(k_compare_planes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Ta, Ca, Tb, Cb, n, out);
    return hipGetLastError();
}

hipError_t bcd_launch_uniform_n(const float *ns, int64_t npix, int *d_out, hipStream_t st)
{
    hipLaunchKernelGGL(k_uniform_n, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, ns, npix, d_out);
    return hipGetLastError();
}

// fast != 0: scale-free division + range flag (d_range_flag must be zeroed by the caller); fast == 0: compiler division
// uni_n > 0 (with fast): all sample counts equal this power of two (verified by the kernel, bit 1 of the flag)
hipError_t bcd_launch_pairdist(const float *hist, const float *ns, int W, int H, int D, int b,
                               float *T, uint8_t *Cn, int fast, int *d_range_flag, float uni_n, hipStream_t st)
{
    dim3 grid((W + PD_TW - 1) / PD_TW, (H + PD_TH - 1) / PD_TH), block(256);
    size_t lds = bcd_pairdist_lds_bytes(D, b);
    bool tiled = (D % 4 == 0) && lds <= 160 * 1024;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
#define BCD_PD_LAUNCH(DD, FF, UU)                                                                                    \
    {                                                                                                                \
        static std::atomic<size_t> granted[64]; /* per instantiation and device: make the attribute call once per size */ \
        if (lds > 64 * 1024 && (dev < 0 || dev >= 64 || granted[dev].load() < lds)) {                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pairdist<DD, FF, UU>),              \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
            if (e != hipSuccess) return e;                                                                           \
            if (dev >= 0 && dev < 64) granted[dev].store(lds);                                                       \
        }                                                                                                            \
        hipLaunchKernelGGL((k_pairdist<DD, FF, UU>), grid, block, lds, st, hist, ns, W, H, b, T, Cn, d_range_flag, uni_n); \
        return hipGetLastError();                                                                                    \
    }
#define BCD_PD_CASE(DD)                                                                                              \
    case DD:                                                                                                         \
        if (fast && uni_n > 0.f) BCD_PD_LAUNCH(DD, true, true)                                                      \
        else if (fast) BCD_PD_LAUNCH(DD, true, false) else BCD_PD_LAUNCH(DD, false, false)
    if (tiled) switch (D) {
        BCD_PD_CASE(60)
        BCD_PD_CASE(120)
        BCD_PD_CASE(36)
        BCD_PD_CASE(24)
        BCD_PD_CASE(12)
        default: break;
    }
#undef BCD_PD_CASE
#undef BCD_PD_LAUNCH
    hipLaunchKernelGGL(k_pairdist_generic, grid, block, 0, st, hist, ns, W, H, D, b, T, Cn);
    return hipGetLastError();
}

hipError_t bcd_launch_verify_pairs(const float *, const float *, int, int, int, int, float, const void *, const int *, int, uint32_t *, hipStream_t);

// forward bits of the lines [row_begin, row_end) (row_begin a multiple of FWD_RB; the planes of the lines row_begin - 1 .. row_end must be complete).
// ap != nullptr: T / Cn are the approximate planes of k_pairdist_rw (T in binary16, passed as an untyped buffer); pairs within
// tau (1 +- BCD_APPROX_DELTA) are appended to the borderline list
hipError_t bcd_launch_fwd_masks_rows(const float *T, const uint8_t *Cn, int W, int H, int b, float tau, uint32_t *fwd_scratch, hipStream_t st,
                                     const BcdBorderline *ap, int row_begin, int row_end)
{
    if (row_begin % FWD_RB != 0 || row_begin < 0 || row_end > H) return hipErrorInvalidValue;
    if (row_begin >= row_end) return hipSuccess;
    const int fwords = (bcd_delta_count(b) + 31) / 32;
    BcdBorderline bl = { tau, nullptr, nullptr, 0 };
    float tau_k = tau;
    if (ap) { bl = *ap; bl.tau_hi = tau * (1.f + BCD_APPROX_DELTA); tau_k = tau * (1.f - BCD_APPROX_DELTA); }
    // the wide kernel needs enough lines to fill the chip (few, fat wavefronts); small scales keep the narrow one
    const bool wide = W % 4 == 0 && (int64_t)W * H >= 400000;
    const int rb0 = row_begin / FWD_RB, nrb = (row_end - row_begin + FWD_RB - 1) / FWD_RB;
    const dim3 gw((W + 247) / 248, nrb, fwords), gn((W + 61) / 62, nrb, fwords);
    if (wide && ap)
        hipLaunchKernelGGL(k_fwd_masks_w1v4a, gw, dim3(64), 0, st, reinterpret_cast<const __half *>(T), Cn, W, H, b, tau_k, fwords, bcd_delta_count(b), fwd_scratch, bl, rb0, row_end);
    else if (wide)
        hipLaunchKernelGGL(k_fwd_masks_w1v4, gw, dim3(64), 0, st, T, Cn, W, H, b, tau_k, fwords, bcd_delta_count(b), fwd_scratch, rb0, row_end);
    else if (ap)
        hipLaunchKernelGGL(k_fwd_masks_w1<true>, gn, dim3(64), 0, st, reinterpret_cast<const __half *>(T), Cn, W, H, b, tau_k, fwords, bcd_delta_count(b), fwd_scratch, bl, rb0, row_end);
    else
        hipLaunchKernelGGL(k_fwd_masks_w1<false>, gn, dim3(64), 0, st, T, Cn, W, H, b, tau_k, fwords, bcd_delta_count(b), fwd_scratch, bl, rb0, row_end);
    return hipGetLastError();
}

// the rest of the w = 1 mask stage: exact re-evaluation of the listed borderline pairs (ap != nullptr), then the symmetric masks and |S|
hipError_t bcd_launch_masks_finish(int W, int H, int b, float tau, uint32_t *mask, int32_t *count, uint32_t *fwd_scratch, hipStream_t st,
                                   const BcdBorderline *ap, const float *hist, const float *ns, int D)
{
    const int side = 2 * b + 1, words = (side * side + 31) / 32, fwords = (bcd_delta_count(b) + 31) / 32;
    if (ap) {
        hipError_t e = bcd_launch_verify_pairs(hist, ns, W, H, D, b, tau, ap->list, ap->counter, ap->capacity, fwd_scratch, st);
        if (e != hipSuccess) return e;
    }
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    if (b == 12 || b == 6) {
        // strips of 64 columns, cut into chunks of lines so that the launch has ~2 000 workgroups; a chunk re-reads the b lines above it
        const int strips = (W + 63) / 64;
        int chunk = ((H * strips + 2047) / 2048 + 3) / 4 * 4;
        chunk = std::min(std::max(chunk, 16), 128);
        const size_t lds = (size_t)(b + 8) * (64 + 2 * b) * fwords * sizeof(uint32_t); // b = 12: 70 400 bytes, b = 6: 12 768
        const dim3 rgrid(strips, (H + chunk - 1) / chunk);
        if (b == 12) {
            static std::atomic<int> granted[64];
            int dev = -1;
            if (hipGetDevice(&dev) != hipSuccess) dev = -1;
            if (dev < 0 || dev >= 64 || granted[dev].load() == 0) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sym_masks_roll<12>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
                if (dev >= 0 && dev < 64) granted[dev].store(1);
            }
            hipLaunchKernelGGL(k_sym_masks_roll<12>, rgrid, dim3(256), lds, st, fwd_scratch, W, H, chunk, mask, count);
        } else
            hipLaunchKernelGGL(k_sym_masks_roll<6>, rgrid, dim3(256), lds, st, fwd_scratch, W, H, chunk, mask, count);
    } else
        hipLaunchKernelGGL(k_sym_masks, grid, dim3(256), 0, st, fwd_scratch, W, H, b, fwords, words, mask, count);
    return hipGetLastError();
}

// ap != nullptr: T / Cn are the approximate planes of k_pairdist_rw (T in binary16, passed as an untyped buffer); pairs within tau (1 +- BCD_APPROX_DELTA) are listed and
// re-evaluated exactly from (hist, ns) before the symmetric masks are completed (w = 1 only)
hipError_t bcd_launch_masks(const float *T, const uint8_t *Cn, int W, int H, int w, int b, float tau,
                            uint32_t *mask, int32_t *count, uint32_t *fwd_scratch, hipStream_t st,
                            const BcdBorderline *ap, const float *hist, const float *ns, int D)
{
    const int side = 2 * b + 1, words = (side * side + 31) / 32;
    int64_t npix = (int64_t)W * H;
    if (w == 1 && fwd_scratch) {
        hipError_t e = bcd_launch_fwd_masks_rows(T, Cn, W, H, b, tau, fwd_scratch, st, ap, 0, H);
        if (e != hipSuccess) return e;
        return bcd_launch_masks_finish(W, H, b, tau, mask, count, fwd_scratch, st, ap, hist, ns, D);
    }
    if (ap) return hipErrorInvalidValue;
    dim3 block(64, 4);
    dim3 grid((W + 63) / 64, H, (words + 3) / 4);
    hipLaunchKernelGGL(k_masks, grid, block, 0, st, T, Cn, W, H, w, b, tau, words, mask);
    hipLaunchKernelGGL(k_mask_count, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, mask, npix, words, count);
    return hipGetLastError();
}

hipError_t bcd_launch_window_distances(const float *T, const uint8_t *Cn, int W, int H, int w, int b, int r, int c,
                                       float *out, hipStream_t st)
{
    int n = (2 * b + 1) * (2 * b + 1);
    hipLaunchKernelGGL(k_window_distances, dim3((n + 63) / 64), dim3(64), 0, st, T, Cn, W, H, w, b, r, c, out);
    return hipGetLastError();
}
