// micro-benchmark: compaction of the processed pixels into strong / weak lists (k_active_lists of k_active.hip) -- variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>

constexpr int LPT = 8;
// V0: the shipped kernel (1024 threads x 8 pixels, thread 0 scans 128 LDS entries, 3 atomics per workgroup)
__global__ __launch_bounds__(1024) void k_v0(const uint8_t *__restrict__ state, const int32_t *__restrict__ nsim, int64_t npix, int min_strong,
                                             int32_t *__restrict__ strong_list, int32_t *__restrict__ weak_list, int32_t *__restrict__ counters)
{
    __shared__ int ws[LPT][16], ww[LPT][16], base[2];
    __shared__ long long wt[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * (1024 * LPT) + threadIdx.x;
    unsigned long long bs[LPT], bw[LPT];
    long long tot = 0;
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
        const int64_t p = p0 + u * 1024;
        const bool in = p < npix && state[p] == 1;
        const int n = in ? nsim[p] : 0;
        bs[u] = __ballot(in && n >= min_strong);
        bw[u] = __ballot(in && n < min_strong);
        tot += n;
    }
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off);
    if (lane == 0) {
#pragma unroll
        for (int u = 0; u < LPT; ++u) { ws[u][wave] = __popcll(bs[u]); ww[u][wave] = __popcll(bw[u]); }
        wt[wave] = tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0, w = 0;
        long long t = 0;
        for (int u = 0; u < LPT; ++u)
            for (int i = 0; i < 16; ++i) { int a = ws[u][i], bq = ww[u][i]; ws[u][i] = s; ww[u][i] = w; s += a; w += bq; }
        for (int i = 0; i < 16; ++i) t += wt[i];
        base[0] = s ? atomicAdd(&counters[0], s) : 0;
        base[1] = w ? atomicAdd(&counters[1], w) : 0;
        if (t) atomicAdd(reinterpret_cast<unsigned long long *>(&counters[2]), (unsigned long long)t);
    }
    __syncthreads();
    const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
        const int64_t p = p0 + u * 1024;
        if ((bs[u] >> lane) & 1ull) strong_list[base[0] + ws[u][wave] + __popcll(bs[u] & lower)] = (int32_t)p;
        if ((bw[u] >> lane) & 1ull) weak_list[base[1] + ww[u][wave] + __popcll(bw[u] & lower)] = (int32_t)p;
    }
}

// V1: 256 threads, 16 pixels per thread as 4 x (uchar4 state + int4 |S|) = 4096 pixels per workgroup; per-wavefront counts combined by the
// first wavefront with shuffles; the weak list is optional (WEAK = false: only counted)
template <bool WEAK>
__global__ __launch_bounds__(256) void k_v1(const uint8_t *__restrict__ state, const int32_t *__restrict__ nsim, int64_t npix, int min_strong,
                                            int32_t *__restrict__ strong_list, int32_t *__restrict__ weak_list, int32_t *__restrict__ counters)
{
    constexpr int G = 4; // groups of 4 pixels per thread
    __shared__ int s_cnt[2][4], s_base[2];
    __shared__ long long s_tot[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q0 = (int64_t)blockIdx.x * (256 * G) + threadIdx.x; // index of a group of 4 pixels
    const int64_t nq = npix >> 2; // (npix % 4 == 0 assumed here; the tail is handled by the caller in the real kernel)
    uint32_t sbits = 0, wbits = 0; // 16 flags each
    long long tot = 0;
    uchar4 st[G];
    int4 ns[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t q = q0 + g * 256;
        st[g] = q < nq ? reinterpret_cast<const uchar4 *>(state)[q] : make_uchar4(0, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t q = q0 + g * 256;
        const bool any = st[g].x == 1 || st[g].y == 1 || st[g].z == 1 || st[g].w == 1;
        ns[g] = (q < nq && any) ? reinterpret_cast<const int4 *>(nsim)[q] : make_int4(0, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const uint8_t s4[4] = { st[g].x, st[g].y, st[g].z, st[g].w };
        const int n4[4] = { ns[g].x, ns[g].y, ns[g].z, ns[g].w };
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool in = s4[e] == 1;
            const int n = in ? n4[e] : 0;
            if (in && n >= min_strong) sbits |= 1u << (4 * g + e);
            if (in && n < min_strong) wbits |= 1u << (4 * g + e);
            tot += n;
        }
    }
    int cs = __popc(sbits), cw = __popc(wbits);
    // exclusive scans inside the wavefront
    int is = cs, iw = cw;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int a = __shfl_up(is, off), b = __shfl_up(iw, off);
        if (lane >= off) { is += a; iw += b; }
    }
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off);
    if (lane == 63) { s_cnt[0][wave] = is; s_cnt[1][wave] = iw; }
    if (lane == 0) s_tot[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int s = s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3], w = s_cnt[1][0] + s_cnt[1][1] + s_cnt[1][2] + s_cnt[1][3];
        const long long t = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
        s_base[0] = s ? atomicAdd(&counters[0], s) : 0;
        s_base[1] = w ? atomicAdd(&counters[1], w) : 0;
        if (t) atomicAdd(reinterpret_cast<unsigned long long *>(&counters[2]), (unsigned long long)t);
    }
    __syncthreads();
    int bs = s_base[0] + is - cs, bw = s_base[1] + iw - cw;
    for (int i = 0; i < wave; ++i) { bs += s_cnt[0][i]; bw += s_cnt[1][i]; }
    while (sbits) {
        const int bit = __ffs(sbits) - 1;
        sbits &= sbits - 1;
        strong_list[bs++] = (int32_t)(4 * (q0 + (bit >> 2) * 256) + (bit & 3));
    }
    if (WEAK)
        while (wbits) {
            const int bit = __ffs(wbits) - 1;
            wbits &= wbits - 1;
            weak_list[bw++] = (int32_t)(4 * (q0 + (bit >> 2) * 256) + (bit & 3));
        }
}

int main()
{
    const int W = 1920, H = 1080;
    const int64_t npix = (int64_t)W * H;
    std::vector<uint8_t> st(npix);
    std::vector<int32_t> ns(npix);
    srand(3);
    for (int64_t i = 0; i < npix; ++i) { int r = rand() % 1000; st[i] = r < 248 ? 1 : 2; ns[i] = (rand() % 1000 < 62) ? 28 + rand() % 100 : rand() % 28; }
    uint8_t *d_st; int32_t *d_ns, *d_s, *d_w, *d_c;
    hipMalloc(&d_st, npix); hipMalloc(&d_ns, npix * 4); hipMalloc(&d_s, npix * 4); hipMalloc(&d_w, npix * 4); hipMalloc(&d_c, 64);
    hipMemcpy(d_st, st.data(), npix, hipMemcpyHostToDevice); hipMemcpy(d_ns, ns.data(), npix * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 3; ++variant) {
        float best = 1e9f;
        int32_t h[4] = { 0, 0, 0, 0 };
        for (int r = 0; r < 6; ++r) {
            hipMemset(d_c, 0, 64);
            hipEventRecord(e0, 0);
            if (variant == 0) hipLaunchKernelGGL(k_v0, dim3((unsigned)((npix + 8191) / 8192)), dim3(1024), 0, 0, d_st, d_ns, npix, 28, d_s, d_w, d_c);
            else if (variant == 1) hipLaunchKernelGGL(k_v1<true>, dim3((unsigned)((npix / 4 + 1023) / 1024)), dim3(256), 0, 0, d_st, d_ns, npix, 28, d_s, d_w, d_c);
            else hipLaunchKernelGGL(k_v1<false>, dim3((unsigned)((npix / 4 + 1023) / 1024)), dim3(256), 0, 0, d_st, d_ns, npix, 28, d_s, d_w, d_c);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r > 0 && ms < best) best = ms;
        }
        hipMemcpy(h, d_c, 16, hipMemcpyDeviceToHost);
        printf("variant %d: %.1f us  strong %d weak %d\n", variant, best * 1e3f, h[0], h[1]);
    }
    return 0;
}
