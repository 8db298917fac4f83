// k_pointwise.hip -- per-pixel pre/post passes, multiscale pyramid + merge, spike prefilter.
// All arithmetic is written in the reference's evaluation order and compiled with -ffp-contract=off,
// so these kernels are bit-exact against the CPU path.
#include "bcd_common.h"

namespace {

// Denoiser::computePixelCovFromSampleCov (src/core/Denoiser.cpp:357-373): cov * (1.f / n)
__global__ void k_pixel_cov(const float *__restrict__ cov, const float *__restrict__ ns, int64_t npix, float *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * 6) return;
    float inv = 1.f / ns[i / 6];
    out[i] = cov[i] * inv;
}

// the same, and the accumulators of the scale cleared in the same pass (sum: 3 floats per pixel, count: 1 int): one launch on the scale's side
// stream instead of the kernel and two fills (round 4)
__global__ void k_pixel_cov_clear(const float *__restrict__ cov, const float *__restrict__ ns, int64_t npix, float *__restrict__ out,
                                  float *__restrict__ sum, int32_t *__restrict__ cnt)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * 6) return;
    float inv = 1.f / ns[i / 6];
    out[i] = cov[i] * inv;
    if (i < npix * 3) sum[i] = 0.f;
    else if (i < npix * 4) cnt[i - npix * 3] = 0;
}

// The same, TWO pixels per thread (round 6): twelve covariance values as three 16-byte loads / stores, one division per pixel instead of one per value
// (the quotient is the same correctly rounded 1.f / n, every product the same single multiplication: bit-identical), no 64-bit index arithmetic.  The
// per-value form above ran at 1.1 TB/s on the 3840x2160 scale -- 50 M threads, each with a 64-bit division by six and an IEEE division.
__global__ __launch_bounds__(256) void k_pixel_cov_clear2(const float *__restrict__ cov, const float *__restrict__ ns, uint32_t npairs, float *__restrict__ out,
                                                          float *__restrict__ sum, int32_t *__restrict__ cnt)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= npairs) return;
    const float2 n = reinterpret_cast<const float2 *>(ns)[t];
    const float i0 = 1.f / n.x, i1 = 1.f / n.y;
    const float4 *src = reinterpret_cast<const float4 *>(cov) + 3 * (size_t)t;
    const float4 a = src[0], b = src[1], c = src[2]; // pixel 0: a.xyzw b.xy, pixel 1: b.zw c.xyzw
    float4 *dst = reinterpret_cast<float4 *>(out) + 3 * (size_t)t;
    dst[0] = make_float4(a.x * i0, a.y * i0, a.z * i0, a.w * i0);
    dst[1] = make_float4(b.x * i0, b.y * i0, b.z * i1, b.w * i1);
    dst[2] = make_float4(c.x * i1, c.y * i1, c.z * i1, c.w * i1);
    float2 *s2 = reinterpret_cast<float2 *>(sum) + 3 * (size_t)t;
    s2[0] = make_float2(0.f, 0.f); s2[1] = make_float2(0.f, 0.f); s2[2] = make_float2(0.f, 0.f);
    reinterpret_cast<int2 *>(cnt)[t] = make_int2(0, 0);
}

// every counter, flag and work queue a scale's chain starts from, in one launch (round 4; before: one fill per buffer, on the critical stream):
// a: the 64 control words (keep0 / keep1: words that belong to launches already made -- the flags of distance planes computed ahead),
// b: the sub-counter lines of the marking launches, c: the work queues of the estimate kernels
__global__ void k_scale_begin(int *__restrict__ a, int na, int keep0, int keep1, int *__restrict__ b, int nb, int *__restrict__ c, int nc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) { if (i != keep0 && i != keep1) a[i] = 0; }
    else if (i < na + nb) b[i - na] = 0;
    else if (i < na + nb + nc) c[i - na - nb] = 0;
}

// Denoiser::finalAggregation tail (src/core/Denoiser.cpp:458-469): (1.f / count) * sum
__global__ void k_finalize(const float *__restrict__ sum, const int32_t *__restrict__ cnt, int64_t npix, float *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * 3) return;
    float inv = 1.f / (float)cnt[i / 3];
    out[i] = inv * sum[i];
}

// the same, one thread per pixel: one division instead of three, no 64-bit index arithmetic (round 6; 4K: 0.42 -> 0.1 ms)
__global__ __launch_bounds__(256) void k_finalize_px(const float *__restrict__ sum, const int32_t *__restrict__ cnt, uint32_t npix, float *__restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= npix) return;
    const float inv = 1.f / (float)cnt[i];
    const float *s3 = sum + 3 * (size_t)i;
    float *o3 = out + 3 * (size_t)i;
    o3[0] = inv * s3[0]; o3[1] = inv * s3[1]; o3[2] = inv * s3[2];
}

// finalisation of a band (multi-GPU row-band path): out = (1 / (count + halo counts)) * (sum + halo sums) on `rows` lines; the
// first / last `halo` lines of the range also receive the accumulator halos of the neighbouring bands (nullptr at a frame border).
// Same operations as "add the received halos, then k_finalize".
__global__ void k_finalize_band(const float *__restrict__ sum, const int32_t *__restrict__ cnt, int W, int rows, int halo,
                                const float *__restrict__ up_sum, const int32_t *__restrict__ up_cnt,
                                const float *__restrict__ dn_sum, const int32_t *__restrict__ dn_cnt, float *__restrict__ out)
{
    const int64_t npix = (int64_t)W * rows;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * 3) return;
    const int64_t pix = i / 3;
    const int r = (int)(pix / W);
    float s = sum[i];
    int c = cnt[pix];
    if (up_sum && r < halo) { s += up_sum[i]; c += up_cnt[pix]; }
    if (dn_sum && r >= rows - halo) {
        const int64_t off = (int64_t)(rows - halo) * W;
        s += dn_sum[i - off * 3];
        c += dn_cnt[pix - off];
    }
    out[i] = (1.f / (float)c) * s;
}

// checkAndPutToZeroNegativeInfNaNValues (src/cli/main.cpp:389-420)
__global__ void k_zero_bad(float *__restrict__ img, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = img[i];
    if (v < 0.f || isnan(v) || isinf(v)) img[i] = 0.f;
}

// 2x2 block positions of MultiscaleDenoiser::downscale* (src/core/MultiscaleDenoiser.cpp:256-266):
// p1=(2l,2c), p2 = next line, p3 = next column, p4 = both; clamped to the image.
__device__ inline void block_pos(int W, int H, int l, int c, size_t p[4])
{
    int l1 = min(2 * l + 1, H - 1), c1 = min(2 * c + 1, W - 1);
    p[0] = (size_t)(2 * l) * W + 2 * c;
    p[1] = (size_t)l1 * W + 2 * c;
    p[2] = (size_t)(2 * l) * W + c1;
    p[3] = (size_t)l1 * W + c1;
}

// mode 0: downscaleSum (:243-268)   mode 1: downscaleAverage / downscale (:270-295,514-539)
template <int MODE>
__global__ void k_downscale(const float *__restrict__ in, int W, int H, int D, float *__restrict__ out)
{
    const int w2 = W / 2, h2 = H / 2;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)w2 * h2 * D) return;
    int z = (int)(i % D);
    int64_t pq = i / D;
    int c = (int)(pq % w2), l = (int)(pq / w2);
    size_t p[4];
    block_pos(W, H, l, c, p);
    float v = in[p[0] * D + z] + in[p[1] * D + z] + in[p[2] * D + z] + in[p[3] * D + z];
    out[i] = MODE == 0 ? v : 0.25f * v;
}

// downscaleSampleCovarianceSum (:297-334): w_i = (1/16) * nSum / n_i
__global__ void k_downscale_cov(const float *__restrict__ cov, const float *__restrict__ ns, int W, int H, float *__restrict__ out)
{
    const int w2 = W / 2, h2 = H / 2;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)w2 * h2 * 6) return;
    int z = (int)(i % 6);
    int64_t pq = i / 6;
    int c = (int)(pq % w2), l = (int)(pq / w2);
    size_t p[4];
    block_pos(W, H, l, c, p);
    const float sq = (1.f / 4.f) * (1.f / 4.f);
    float n1 = ns[p[0]], n2 = ns[p[1]], n3 = ns[p[2]], n4 = ns[p[3]];
    float nsum = n1 + n2 + n3 + n4;
    float w1 = sq * nsum / n1, w2_ = sq * nsum / n2, w3 = sq * nsum / n3, w4 = sq * nsum / n4;
    out[i] = w1 * cov[p[0] * 6 + z] + w2_ * cov[p[1] * 6 + z] + w3 * cov[p[2] * 6 + z] + w4 * cov[p[3] * 6 + z];
}
// the same, one thread per output pixel: four divisions per pixel instead of twenty-four, no 64-bit index arithmetic (round 6)
__global__ __launch_bounds__(256) void k_downscale_cov_px(const float *__restrict__ cov, const float *__restrict__ ns, int W, int H, float *__restrict__ out, uint32_t npix2)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= npix2) return;
    const uint32_t w2 = (uint32_t)W / 2u;
    const int l = (int)(i / w2), c = (int)(i - (uint32_t)l * w2);
    size_t p[4];
    block_pos(W, H, l, c, p);
    const float sq = (1.f / 4.f) * (1.f / 4.f);
    const float n1 = ns[p[0]], n2 = ns[p[1]], n3 = ns[p[2]], n4 = ns[p[3]];
    const float nsum = n1 + n2 + n3 + n4;
    const float w1 = sq * nsum / n1, w2_ = sq * nsum / n2, w3 = sq * nsum / n3, w4 = sq * nsum / n4;
#pragma unroll
    for (int z = 0; z < 6; ++z)
        out[(size_t)i * 6 + z] = w1 * cov[p[0] * 6 + z] + w2_ * cov[p[1] * 6 + z] + w3 * cov[p[2] * 6 + z] + w4 * cov[p[3] * 6 + z];
}

__device__ inline int clamp_pos(int v, int maxp1) { return v <= 0 ? 0 : (v >= maxp1 ? maxp1 - 1 : v); }

// The same for depths that are multiples of four (the histograms: D = 60), one thread per (output pixel, group of four bins): 16-byte loads and
// stores, 32-bit index arithmetic (round 6).  The per-value kernel above spends two 64-bit divisions per float and read the 2 GB of a 3840x2160
// histogram image at 2.9 TB/s.  Same four additions per value, in the same order.
template <int MODE>
__global__ __launch_bounds__(256) void k_downscale4(const float *__restrict__ in, int W, int H, int Q /* D / 4 */, float *__restrict__ out, uint32_t n /* w2 h2 Q */)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t w2 = (uint32_t)W / 2u;
    const uint32_t pq = i / (uint32_t)Q, q = i - pq * (uint32_t)Q;
    const int l = (int)(pq / w2), c = (int)(pq - (uint32_t)l * w2);
    size_t p[4];
    block_pos(W, H, l, c, p);
    const float4 *src = reinterpret_cast<const float4 *>(in);
    const float4 a = src[p[0] * Q + q], b = src[p[1] * Q + q], d = src[p[2] * Q + q], e = src[p[3] * Q + q];
    float4 v = make_float4(a.x + b.x + d.x + e.x, a.y + b.y + d.y + e.y, a.z + b.z + d.z + e.z, a.w + b.w + d.w + e.w);
    if (MODE != 0) v = make_float4(0.25f * v.x, 0.25f * v.y, 0.25f * v.z, 0.25f * v.w);
    reinterpret_cast<float4 *>(out)[i] = v;
}
// One thread per output PIXEL for shallow images (colours, sample counts: D <= 8), every value by the same four additions (round 6): the per-value
// kernel pays two 64-bit divisions for each float.
template <int MODE>
__global__ __launch_bounds__(256) void k_downscale_px(const float *__restrict__ in, int W, int H, int D, float *__restrict__ out, uint32_t npix2)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= npix2) return;
    const uint32_t w2 = (uint32_t)W / 2u;
    const int l = (int)(i / w2), c = (int)(i - (uint32_t)l * w2);
    size_t p[4];
    block_pos(W, H, l, c, p);
    for (int z = 0; z < D; ++z) {
        const float v = in[p[0] * D + z] + in[p[1] * D + z] + in[p[2] * D + z] + in[p[3] * D + z];
        out[(size_t)i * D + z] = MODE == 0 ? v : 0.25f * v;
    }
}

// interpolate (:473-512): 9/16 main, 3/16 x (two adjacent, summed first), 1/16 diagonal.
// MODE 0: hi = up(lo)     MODE 1: hi -= up(lo)     MODE 2: hi += up(lo)
template <int MODE>
__global__ void k_interpolate(const float *__restrict__ lo, int w, int h, int D, float *__restrict__ hi, int W, int H)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)W * H * D) return;
    int z = (int)(i % D);
    int64_t pq = i / D;
    int uc = (int)(pq % W), ul = (int)(pq / W);
    int l = ul / 2, c = uc / 2;
    int al = clamp_pos(l + ((ul % 2) * 2 - 1), h);
    int ac = clamp_pos(c + ((uc % 2) * 2 - 1), w);
    int lc = max(0, min(l, h - 1)), cc = max(0, min(c, w - 1));
    const float wm = 9.f / 16.f, wa = 3.f / 16.f, wd = 1.f / 16.f;
    float v = wm * lo[((size_t)lc * w + cc) * D + z] +
              wa * (lo[((size_t)lc * w + ac) * D + z] + lo[((size_t)al * w + cc) * D + z]) +
              wd * lo[((size_t)al * w + ac) * D + z];
    if (MODE == 0) hi[i] = v;
    else if (MODE == 1) hi[i] -= v;
    else hi[i] += v;
}

// interpolation weights' four source pixels of output pixel (ul, uc): main, column neighbour, line neighbour, diagonal (indices into the w x h image)
__device__ inline void interp_pos(int ul, int uc, int w, int h, uint32_t (&q)[4])
{
    const int l = ul / 2, c = uc / 2;
    const int al = clamp_pos(l + ((ul % 2) * 2 - 1), h), ac = clamp_pos(c + ((uc % 2) * 2 - 1), w);
    const int lc = max(0, min(l, h - 1)), cc = max(0, min(c, w - 1));
    q[0] = (uint32_t)(lc * w + cc); q[1] = (uint32_t)(lc * w + ac); q[2] = (uint32_t)(al * w + cc); q[3] = (uint32_t)(al * w + ac);
}
__device__ inline float interp_value(const float *__restrict__ lo, const uint32_t (&q)[4], int D, int z)
{
    const float wm = 9.f / 16.f, wa = 3.f / 16.f, wd = 1.f / 16.f;
    return wm * lo[(size_t)q[0] * D + z] + wa * (lo[(size_t)q[1] * D + z] + lo[(size_t)q[2] * D + z]) + wd * lo[(size_t)q[3] * D + z];
}
// one thread per output pixel (shallow images), same expression per value (round 6)
template <int MODE>
__global__ __launch_bounds__(256) void k_interpolate_px(const float *__restrict__ lo, int w, int h, int D, float *__restrict__ hi, int W, uint32_t npix)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= npix) return;
    const int ul = (int)(i / (uint32_t)W), uc = (int)(i - (uint32_t)ul * (uint32_t)W);
    uint32_t q[4];
    interp_pos(ul, uc, w, h, q);
    for (int z = 0; z < D; ++z) {
        const float v = interp_value(lo, q, D, z);
        float *dst = hi + (size_t)i * D + z;
        if (MODE == 0) *dst = v;
        else if (MODE == 1) *dst -= v;
        else *dst += v;
    }
}
// mergeOutputs' two interpolations in one pass over the fine image (MultiscaleDenoiser.cpp:453-466): hi = (hi - up(a)) + up(b) -- the same two roundings
// per value as "hi -= up(a)" followed by "hi += up(b)", one read-modify-write of the fine image instead of two (round 6)
__global__ __launch_bounds__(256) void k_merge_px(const float *__restrict__ a, const float *__restrict__ b, int w, int h, int D, float *__restrict__ hi, int W, uint32_t npix)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= npix) return;
    const int ul = (int)(i / (uint32_t)W), uc = (int)(i - (uint32_t)ul * (uint32_t)W);
    uint32_t q[4];
    interp_pos(ul, uc, w, h, q);
    for (int z = 0; z < D; ++z) {
        float *dst = hi + (size_t)i * D + z;
        float r = *dst;
        r -= interp_value(a, q, D, z);
        r += interp_value(b, q, D, z);
        *dst = r;
    }
}

// SpikeRemovalFilter::filter (src/core/SpikeRemovalFilter.cpp:18-116), float-abs semantics; out of place.
__global__ void k_spike(const float *__restrict__ col, const float *__restrict__ ns, const float *__restrict__ hist,
                        const float *__restrict__ cov, int W, int H, int D, float factor,
                        float *__restrict__ ocol, float *__restrict__ ons, float *__restrict__ ohist, float *__restrict__ ocov, int row_begin, int vec16 /* hist / ohist are 16-byte aligned */)
{
    // (round 4) the decision is per pixel, the copies are per workgroup: the 64 source indices go through LDS and the 64 lanes then move the 64
    // pixels' values as contiguous runs (16 bytes per lane for the histograms).  One lane copying the 60 bins of its own pixel was 4 bytes per lane
    // at a stride of 240: 5 ms for a 4K frame, now bound by the 2 x 2.4 GB it moves.
    __shared__ unsigned int s_src[64];
    const int c0 = blockIdx.x * 64, c = c0 + threadIdx.x, l = row_begin + blockIdx.y;
    const bool in_row = c < W;
    const int ncols = min(64, W - c0);
    size_t dst = (size_t)l * W + min(c, W - 1), src = dst;
    if (in_row) {
    int cl = l < 1 ? 1 : (l > H - 2 ? H - 2 : l);
    int cc = c < 1 ? 1 : (c > W - 2 ? W - 2 : c);
    float v[3][9];
    int k = 0;
    for (int nl = cl - 1; nl <= cl + 1; ++nl)
        for (int nc = cc - 1; nc <= cc + 1; ++nc, ++k) {
            const float *px = col + ((size_t)nl * W + nc) * 3;
            v[0][k] = px[0]; v[1][k] = px[1]; v[2][k] = px[2];
        }
    const float *me = col + ((size_t)l * W + c) * 3;
    bool spike = false;
    for (int ch = 0; ch < 3; ++ch) {
        float total = 0.f;
        for (int i = 0; i < 9; ++i) total += v[ch][i];
        float avg = total / 9;
        total = 0;
        for (int i = 0; i < 9; ++i) total += (v[ch][i] - avg) * (v[ch][i] - avg);
        float sd = sqrtf(total / 8);
        spike = spike || (fabsf(me[ch] - avg) > factor * sd);
    }
    if (spike) {
        int best = 0;
        float bestd = -1.f;
        for (int m = 0; m < 9; ++m) {
            float tot = 0.f;
            for (int i = 0; i < 9; ++i)
                tot += fabsf(v[0][i] - v[0][m]) + fabsf(v[1][i] - v[1][m]) + fabsf(v[2][i] - v[2][m]);
            if (bestd < 0 || tot < bestd) { bestd = tot; best = m; }
        }
        src = (size_t)(cl - 1 + best / 3) * W + (cc - 1 + best % 3);
    }
    }
    s_src[threadIdx.x] = (unsigned int)src; // (pixel indices fit 31 bits: checked by the host entry points)
    __syncthreads();
    const size_t row_base = (size_t)l * W + c0; // first destination pixel of the workgroup
    for (int e = threadIdx.x; e < ncols * 3; e += 64) { const int px = e / 3; ocol[row_base * 3 + e] = col[(size_t)s_src[px] * 3 + (e - px * 3)]; }
    if (in_row) ons[dst] = ns[src];
    for (int e = threadIdx.x; e < ncols * 6; e += 64) { const int px = e / 6; ocov[row_base * 6 + e] = cov[(size_t)s_src[px] * 6 + (e - px * 6)]; }
    if ((D & 3) == 0 && vec16) {
        const int Q = D >> 2;
        const float4 *h4 = reinterpret_cast<const float4 *>(hist);
        float4 *o4 = reinterpret_cast<float4 *>(ohist);
        for (int e = threadIdx.x; e < ncols * Q; e += 64) { const int px = e / Q; o4[row_base * Q + e] = h4[(size_t)s_src[px] * Q + (e - px * Q)]; }
    } else
        for (int e = threadIdx.x; e < ncols * D; e += 64) { const int px = e / D; ohist[row_base * D + e] = hist[(size_t)s_src[px] * D + (e - px * D)]; }
}

// SamplesAccumulator::addSample + computeSampleStatistics on the device (src/core/SamplesAccumulator.cpp:44-141):
// one thread per pixel walks its samples in order (same operations and order as the host class, so nSamples, mean and
// covariance are bit-identical; the histogram goes through the device powf and agrees to float round-off).
// The per-thread histogram lives in LDS, bin-major ([bin][thread]) to stay bank-conflict free.
__global__ __launch_bounds__(64) void k_accumulate_samples(const float *__restrict__ samples, const float *__restrict__ weights,
                                                           int64_t npix, int spp, int nbins, float gamma, float maxval,
                                                           float *__restrict__ ons, float *__restrict__ omean, float *__restrict__ ocov,
                                                           float *__restrict__ ohist)
{
    extern __shared__ float lds_h[];
    const int t = threadIdx.x, D = 3 * nbins;
    const int64_t p = (int64_t)blockIdx.x * 64 + t;
    for (int k = 0; k < D; ++k) lds_h[k * 64 + t] = 0.f;
    if (p >= npix) return;
    const float sat = 2.f;
    float wsum = 0.f, w2sum = 0.f, m[3] = { 0.f, 0.f, 0.f }, c[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    const float *sp = samples + p * spp * 3;
    for (int i = 0; i < spp; ++i) {
        const float R = sp[3 * i], G = sp[3 * i + 1], B = sp[3 * i + 2];
        const float w = weights ? weights[p * spp + i] : 1.f;
        wsum += w;
        w2sum += w * w;
        m[0] += w * R; m[1] += w * G; m[2] += w * B;
        c[0] += w * R * R; c[1] += w * G * G; c[2] += w * B * B;
        c[3] += w * G * B; c[4] += w * R * B; c[5] += w * R * G;
        const float rgb[3] = { R, G, B };
        for (int ch = 0; ch < 3; ++ch) {
            float v = rgb[ch] > 0 ? rgb[ch] : 0;
            if (gamma > 1) v = powf(v, 1.f / gamma);
            if (maxval > 0) v = v / maxval;
            v = v > sat ? sat : v;
            const float fi = v * (nbins - 2);
            int lo = (int)fi;
            float hw;
            if (lo < nbins - 2) hw = fi - lo;
            else { lo = nbins - 2; hw = (v - 1.0f) / (sat - 1.f); }
            const float lw = 1.0f - hw;
            lds_h[(ch * nbins + lo) * 64 + t] += w * lw;
            lds_h[(ch * nbins + lo + 1) * 64 + t] += w * hw;
        }
    }
    const float inv = 1.f / wsum;
    float mean[3];
    for (int i = 0; i < 3; ++i) { mean[i] = inv * m[i]; omean[p * 3 + i] = mean[i]; }
    float cv[6];
    for (int i = 0; i < 6; ++i) cv[i] = c[i] * inv;
    cv[0] -= mean[0] * mean[0]; cv[1] -= mean[1] * mean[1]; cv[2] -= mean[2] * mean[2];
    cv[3] -= mean[1] * mean[2]; cv[4] -= mean[0] * mean[2]; cv[5] -= mean[0] * mean[1];
    const float bias = 1.f / (1 - w2sum / (wsum * wsum));
    for (int i = 0; i < 6; ++i) ocov[p * 6 + i] = cv[i] * bias;
    ons[p] = wsum;
    for (int k = 0; k < D; ++k) ohist[p * D + k] = lds_h[k * 64 + t];
}

inline unsigned nblk(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

} // namespace

hipError_t bcd_launch_pixel_cov(const float *cov, const float *ns, int64_t npix, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_pixel_cov, dim3(nblk(npix * 6, 256)), dim3(256), 0, st, cov, ns, npix, out);
    return hipGetLastError();
}
hipError_t bcd_launch_pixel_cov_clear(const float *cov, const float *ns, int64_t npix, float *out, float *sum, int32_t *cnt, hipStream_t st)
{
    const int64_t npairs = npix / 2;
    if (npairs > 0 && npairs < (int64_t)1 << 32) {
        hipLaunchKernelGGL(k_pixel_cov_clear2, dim3(nblk(npairs, 256)), dim3(256), 0, st, cov, ns, (uint32_t)npairs, out, sum, cnt);
        const int64_t done = 2 * npairs; // (an odd pixel count: the last pixel by the per-value kernel)
        if (done < npix)
            hipLaunchKernelGGL(k_pixel_cov_clear, dim3(1), dim3(256), 0, st, cov + done * 6, ns + done, npix - done, out + done * 6, sum + done * 3, cnt + done);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_pixel_cov_clear, dim3(nblk(npix * 6, 256)), dim3(256), 0, st, cov, ns, npix, out, sum, cnt);
    return hipGetLastError();
}
hipError_t bcd_launch_scale_begin(int *a, int na, int keep0, int keep1, int *b, int nb, int *c, int nc, hipStream_t st)
{
    hipLaunchKernelGGL(k_scale_begin, dim3(nblk((int64_t)na + nb + nc, 256)), dim3(256), 0, st, a, na, keep0, keep1, b, nb, c, nc);
    return hipGetLastError();
}
hipError_t bcd_launch_finalize(const float *sum, const int32_t *cnt, int64_t npix, float *out, hipStream_t st)
{
    if (npix < (int64_t)1 << 32) hipLaunchKernelGGL(k_finalize_px, dim3(nblk(npix, 256)), dim3(256), 0, st, sum, cnt, (uint32_t)npix, out);
    else hipLaunchKernelGGL(k_finalize, dim3(nblk(npix * 3, 256)), dim3(256), 0, st, sum, cnt, npix, out);
    return hipGetLastError();
}
hipError_t bcd_launch_finalize_band(const float *sum, const int32_t *cnt, int W, int rows, int halo, const float *up_sum, const int32_t *up_cnt,
                                    const float *dn_sum, const int32_t *dn_cnt, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_finalize_band, dim3(nblk((int64_t)W * rows * 3, 256)), dim3(256), 0, st, sum, cnt, W, rows, halo, up_sum, up_cnt, dn_sum,
                       dn_cnt, out);
    return hipGetLastError();
}
hipError_t bcd_launch_zero_bad(float *img, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_zero_bad, dim3(nblk(n, 256)), dim3(256), 0, st, img, n);
    return hipGetLastError();
}
hipError_t bcd_launch_downscale(int mode, const float *in, int W, int H, int D, float *out, hipStream_t st)
{
    int64_t n = (int64_t)(W / 2) * (H / 2) * D;
    if (n <= 0) return hipSuccess;
    if (D % 4 == 0 && n / 4 < (int64_t)1 << 31) { // (histograms: four bins per thread)
        const uint32_t n4 = (uint32_t)(n / 4);
        if (mode == 0) hipLaunchKernelGGL(k_downscale4<0>, dim3(nblk(n4, 256)), dim3(256), 0, st, in, W, H, D / 4, out, n4);
        else hipLaunchKernelGGL(k_downscale4<1>, dim3(nblk(n4, 256)), dim3(256), 0, st, in, W, H, D / 4, out, n4);
        return hipGetLastError();
    }
    if (D <= 8 && (int64_t)(W / 2) * (H / 2) < (int64_t)1 << 31) { // shallow images: one thread per pixel
        const uint32_t npix2 = (uint32_t)(W / 2) * (uint32_t)(H / 2);
        if (mode == 0) hipLaunchKernelGGL(k_downscale_px<0>, dim3(nblk(npix2, 256)), dim3(256), 0, st, in, W, H, D, out, npix2);
        else hipLaunchKernelGGL(k_downscale_px<1>, dim3(nblk(npix2, 256)), dim3(256), 0, st, in, W, H, D, out, npix2);
        return hipGetLastError();
    }
    if (mode == 0) hipLaunchKernelGGL(k_downscale<0>, dim3(nblk(n, 256)), dim3(256), 0, st, in, W, H, D, out);
    else hipLaunchKernelGGL(k_downscale<1>, dim3(nblk(n, 256)), dim3(256), 0, st, in, W, H, D, out);
    return hipGetLastError();
}
hipError_t bcd_launch_downscale_cov(const float *cov, const float *ns, int W, int H, float *out, hipStream_t st)
{
    int64_t n = (int64_t)(W / 2) * (H / 2) * 6;
    if (n <= 0) return hipSuccess;
    if (n / 6 < (int64_t)1 << 31) {
        const uint32_t npix2 = (uint32_t)(n / 6);
        hipLaunchKernelGGL(k_downscale_cov_px, dim3(nblk(npix2, 256)), dim3(256), 0, st, cov, ns, W, H, out, npix2);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_downscale_cov, dim3(nblk(n, 256)), dim3(256), 0, st, cov, ns, W, H, out);
    return hipGetLastError();
}
// hi = (hi - up(a)) + up(b): the two interpolations of mergeOutputs in one pass (a, b: w x h x D)
hipError_t bcd_launch_merge_interpolate(const float *a, const float *b, int w, int h, int D, float *hi, int W, int H, hipStream_t st)
{
    if (D <= 8 && (int64_t)W * H < (int64_t)1 << 31) {
        const uint32_t npix = (uint32_t)W * (uint32_t)H;
        hipLaunchKernelGGL(k_merge_px, dim3(nblk(npix, 256)), dim3(256), 0, st, a, b, w, h, D, hi, W, npix);
        return hipGetLastError();
    }
    const int64_t n = (int64_t)W * H * D;
    hipLaunchKernelGGL(k_interpolate<1>, dim3(nblk(n, 256)), dim3(256), 0, st, a, w, h, D, hi, W, H);
    hipLaunchKernelGGL(k_interpolate<2>, dim3(nblk(n, 256)), dim3(256), 0, st, b, w, h, D, hi, W, H);
    return hipGetLastError();
}
hipError_t bcd_launch_interpolate(int mode, const float *lo, int w, int h, int D, float *hi, int W, int H, hipStream_t st)
{
    int64_t n = (int64_t)W * H * D;
    if (D <= 8 && (int64_t)W * H < (int64_t)1 << 31) { // shallow images: one thread per pixel
        const uint32_t npix = (uint32_t)W * (uint32_t)H;
        if (mode == 0) hipLaunchKernelGGL(k_interpolate_px<0>, dim3(nblk(npix, 256)), dim3(256), 0, st, lo, w, h, D, hi, W, npix);
        else if (mode == 1) hipLaunchKernelGGL(k_interpolate_px<1>, dim3(nblk(npix, 256)), dim3(256), 0, st, lo, w, h, D, hi, W, npix);
        else hipLaunchKernelGGL(k_interpolate_px<2>, dim3(nblk(npix, 256)), dim3(256), 0, st, lo, w, h, D, hi, W, npix);
        return hipGetLastError();
    }
    if (mode == 0) hipLaunchKernelGGL(k_interpolate<0>, dim3(nblk(n, 256)), dim3(256), 0, st, lo, w, h, D, hi, W, H);
    else if (mode == 1) hipLaunchKernelGGL(k_interpolate<1>, dim3(nblk(n, 256)), dim3(256), 0, st, lo, w, h, D, hi, W, H);
    else hipLaunchKernelGGL(k_interpolate<2>, dim3(nblk(n, 256)), dim3(256), 0, st, lo, w, h, D, hi, W, H);
    return hipGetLastError();
}
// lines [row_begin, row_end) of the filtered images (a line reads its own and the two adjacent input lines, clamped inward)
hipError_t bcd_launch_spike_rows(const float *col, const float *ns, const float *hist, const float *cov, int W, int H, int D,
                                 float factor, float *ocol, float *ons, float *ohist, float *ocov, int row_begin, int row_end, hipStream_t st)
{
    if (row_end <= row_begin) return hipSuccess;
    // (the public entry point takes any device pointers -- a view into a larger buffer may be 4-byte aligned only: those take the scalar copies)
    const int vec16 = (((uintptr_t)hist | (uintptr_t)ohist) & 15) == 0 ? 1 : 0;
    hipLaunchKernelGGL(k_spike, dim3((W + 63) / 64, row_end - row_begin), dim3(64), 0, st, col, ns, hist, cov, W, H, D, factor, ocol, ons, ohist, ocov, row_begin, vec16);
    return hipGetLastError();
}

hipError_t bcd_launch_spike(const float *col, const float *ns, const float *hist, const float *cov, int W, int H, int D,
                            float factor, float *ocol, float *ons, float *ohist, float *ocov, hipStream_t st)
{
    return bcd_launch_spike_rows(col, ns, hist, cov, W, H, D, factor, ocol, ons, ohist, ocov, 0, H, st);
}

hipError_t bcd_launch_accumulate_samples(const float *samples, const float *weights, int64_t npix, int spp, int nbins, float gamma,
                                         float maxval, float *ons, float *omean, float *ocov, float *ohist, hipStream_t st)
{
    size_t lds = (size_t)3 * nbins * 64 * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_accumulate_samples), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_accumulate_samples, dim3(nblk(npix, 64)), dim3(64), lds, st, samples, weights, npix, spp, nbins, gamma, maxval, ons, omean, ocov, ohist);
    return hipGetLastError();
}
