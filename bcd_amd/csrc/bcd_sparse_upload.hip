// bcd_sparse_upload.hip -- host-buffer entry points: the histogram image crosses PCIe without its zeros.
//
// A drop-in caller hands over the reference's DeepImage buffers in host memory; 82 % of the bytes of a frame are the
// histogram image (D = 60 floats per pixel), and a histogram of 8 .. 64 samples is mostly zeros (69 % on the 32-spp bench frame, 85 % at
// 8 spp).  The link moves 56 GB/s, so the upload is 10.7 of the 14.9 ms a 1080p frame takes through bcd_hip_denoise_host.
// Here a few host threads pack each piece of the image into (one bit per value: "is not +0.0f") + (the values that are not, in order) while
// the previous piece travels, and a kernel rebuilds the fp32 image in HBM: lossless -- the test is on the bit pattern, so -0.0f, NaNs and
// denormals travel as values -- and transparent to everything downstream, which sees the same interleaved buffer as after a plain copy.
// Dense images (long sample counts) are recognised on their first piece and copied as they are.
#include "bcd_common.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(_M_X64))
#include <immintrin.h>
#define BCD_SPARSE_X86 1
#endif

namespace {

constexpr int SP_BLOCK = 2048;              // floats per unpack wavefront: 64 mask words
constexpr size_t SP_PIECE = (size_t)12 << 20; // floats per piece (48 MB of input): what is packed while the previous piece travels
constexpr int SP_BUFFERS = 3;

// rebuilds floats [blk * 2048, ...) of dst: lane l owns 32 consecutive values, its set bits take the next values of the packed stream.
// The block's packed values go to LDS with coalesced 16-byte loads first (they start at any offset of the stream), the scatter reads LDS.
__global__ __launch_bounds__(64) void k_sparse_unpack(const uint32_t *__restrict__ masks, const uint32_t *__restrict__ voff, const uint32_t *__restrict__ vcnt,
                                                      const float *__restrict__ vals, float *__restrict__ dst, size_t n)
{
    __shared__ float4 s_vals4[SP_BLOCK / 4 + 1];
    float *s_vals = reinterpret_cast<float *>(s_vals4);
    const int lane = threadIdx.x;
    const size_t blk = blockIdx.x;
    const uint32_t m = masks[blk * 64 + lane];
    const uint32_t v0 = voff[blk], nv = vcnt[blk];
    // the values of the block start at vals + v0 (any alignment): whole 16-byte groups from the aligned address below it
    const uint32_t a0 = v0 & ~3u, shift = v0 - a0, ngroups = (shift + nv + 3) / 4;
    const float4 *src4 = reinterpret_cast<const float4 *>(vals + a0);
    for (uint32_t g = lane; g < ngroups; g += 64) s_vals4[g] = src4[g];
    const int c = __popc(m);
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    __syncthreads();
    const float *p = s_vals + shift + (incl - c);
    const size_t base = blk * SP_BLOCK + (size_t)lane * 32;
    int k = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = 0.f; if ((m >> (4 * j + e)) & 1u) v[e] = p[k++]; }
        const size_t i = base + 4 * j;
        if (i + 3 < n) *reinterpret_cast<float4 *>(dst + i) = make_float4(v[0], v[1], v[2], v[3]); // (dst + i is 16-byte aligned: callers pass n0 % 4 == 0)
        else
            for (int e = 0; e < 4; ++e) if (i + e < n) dst[i + e] = v[e];
    }
}

struct Piece {
    const float *src = nullptr;
    size_t n = 0, nblk = 0;
    // pinned: meta = [masks: 64 words per block | voff: 1 | vcnt: 1] and the packed values, ONE contiguous stream (sub-blocks in the order the
    // threads finished them: voff says where a block's values start); device copies of both
    uint32_t *h_meta = nullptr, *h_vals = nullptr, *d_meta = nullptr, *d_vals = nullptr;
    size_t cap_floats = 0, cap_blk = 0;              // capacity the buffers were sized for
    int segs = 0;                                    // packing tasks of a job (sub-block ranges)
    size_t per_task = 0;                             // blocks per task
    hipEvent_t ev = nullptr; bool ev_pending = false;
    // the packing job on this buffer: tasks are claimed from `next`, `remaining` counts the ones not finished (a worker that is late
    // for a job only ever touches the counters of the buffer it was handed, never those of a newer job on another buffer)
    std::atomic<int> next{ 0 }, remaining{ 0 };
    std::atomic<size_t> cursor{ 0 };                 // values appended to h_vals so far
    int drainers = 0;                                // pool threads inside drain() on this buffer (guarded by the uploader's mutex): submit() waits for 0
                                                     // before it re-arms the counters, so a late worker can never claim a task of a job that is being set up
    uint32_t *masks() const { return h_meta; }                  // (laid out for the piece at hand: nblk blocks)
    uint32_t *voff() const { return h_meta + nblk * 64; }
    uint32_t *vcnt() const { return h_meta + nblk * 65; }
};

// ---- packing one 32-value word group: mask bits + the values that count, appended at out[cnt...]; three forms, chosen once per process.
// All of them may write up to 32 values past the cursor (whatever follows is overwritten by the next group or never sent).
typedef size_t (*pack32_fn)(const uint32_t *in, uint32_t *out, size_t cnt, uint32_t *bits);

size_t pack32_scalar(const uint32_t *in, uint32_t *out, size_t cnt, uint32_t *bits_out)
{
    uint32_t bits = 0;
    for (int i = 0; i < 32; ++i) { // branch-free: every value is stored, the cursor only moves past the ones that count
        const uint32_t v = in[i];
        const uint32_t nz = v != 0u;
        out[cnt] = v;
        cnt += nz;
        bits |= nz << i;
    }
    *bits_out = bits;
    return cnt;
}

#ifdef BCD_SPARSE_X86
// AVX-512: compress in registers (the memory form of vpcompressd is microcoded on some cores), one full-width store per 16 values
__attribute__((target("avx512f,avx512vl,popcnt"))) size_t pack32_avx512(const uint32_t *in, uint32_t *out, size_t cnt, uint32_t *bits_out)
{
    const __m512i a = _mm512_loadu_si512(in), b = _mm512_loadu_si512(in + 16);
    const __mmask16 ka = _mm512_test_epi32_mask(a, a), kb = _mm512_test_epi32_mask(b, b);
    _mm512_storeu_si512(out + cnt, _mm512_maskz_compress_epi32(ka, a));
    cnt += (size_t)_mm_popcnt_u32(ka);
    _mm512_storeu_si512(out + cnt, _mm512_maskz_compress_epi32(kb, b));
    cnt += (size_t)_mm_popcnt_u32(kb);
    *bits_out = (uint32_t)ka | ((uint32_t)kb << 16);
    return cnt;
}

// AVX2: 8 values at a time, the compaction is a lane permutation looked up by the 8-bit mask
alignas(32) uint32_t g_perm8[256][8];
void init_perm8()
{
    for (int m = 0; m < 256; ++m) {
        int k = 0;
        for (int i = 0; i < 8; ++i) if (m >> i & 1) g_perm8[m][k++] = (uint32_t)i;
        for (; k < 8; ++k) g_perm8[m][k] = 0;
    }
}
__attribute__((target("avx2,popcnt"))) size_t pack32_avx2(const uint32_t *in, uint32_t *out, size_t cnt, uint32_t *bits_out)
{
    uint32_t bits = 0;
    const __m256i zero = _mm256_setzero_si256();
    for (int g = 0; g < 4; ++g) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(in + 8 * g));
        const uint32_t m = 0xffu & ~(uint32_t)_mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpeq_epi32(v, zero)));
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(out + cnt), _mm256_permutevar8x32_epi32(v, _mm256_load_si256(reinterpret_cast<const __m256i *>(g_perm8[m]))));
        cnt += (size_t)_mm_popcnt_u32(m);
        bits |= m << (8 * g);
    }
    *bits_out = bits;
    return cnt;
}
#endif

pack32_fn choose_pack32()
{
    const char *force = getenv("BCD_HIP_UPLOAD_SIMD"); // "scalar" / "avx2" / "avx512": tests and timing; default: the widest the host has
#ifdef BCD_SPARSE_X86
    __builtin_cpu_init();
    init_perm8();
    const bool has512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("popcnt");
    const bool has2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt");
    if (force && !strcmp(force, "scalar")) return pack32_scalar;
    if (force && !strcmp(force, "avx2") && has2) return pack32_avx2;
    if (has512 && !(force && !strcmp(force, "avx2"))) return pack32_avx512;
    if (has2) return pack32_avx2;
#endif
    (void)force;
    return pack32_scalar;
}
const pack32_fn g_pack32 = choose_pack32();

} // namespace

// tools / tests: which packer this process uses (0 scalar, 2 AVX2, 5 AVX-512), and one group packed with it
extern "C" int bcd_hip_selftest_pack32(const uint32_t *in32, uint32_t *out64, uint32_t *bits, int *count)
{
    *count = (int)g_pack32(in32, out64, 0, bits);
#ifdef BCD_SPARSE_X86
    return g_pack32 == pack32_avx512 ? 5 : (g_pack32 == pack32_avx2 ? 2 : 0);
#else
    return 0;
#endif
}

struct BcdSparseUploader {
    std::vector<std::thread> pool;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    bool quit = false;
    unsigned long long epoch = 0;   // packing job number
    Piece *job = nullptr;
    Piece buf[SP_BUFFERS];
    int cur = 0;                    // buffer the next submit() packs into
    bool dense = false;             // this frame's image turned out dense: plain copies from here on
    long long raw_bytes = 0, sent_bytes = 0; // of the last frame

    // one task = SP_TASK consecutive blocks: packed into a scratch buffer of the thread (stays in its cache), then appended to the piece's
    // contiguous value stream at a position reserved from the shared cursor
    static constexpr size_t SP_TASK = 16; // blocks: 128 KB of input
    static void pack_task(Piece &pc, int t)
    {
        static thread_local std::vector<uint32_t> scratch;
        if (scratch.size() < SP_TASK * SP_BLOCK + 64) scratch.resize(SP_TASK * SP_BLOCK + 64);
        const size_t b0 = std::min(pc.nblk, SP_TASK * (size_t)t), b1 = std::min(pc.nblk, b0 + SP_TASK);
        const uint32_t *in = reinterpret_cast<const uint32_t *>(pc.src);
        uint32_t *out = scratch.data(), *masks = pc.masks(), *voff = pc.voff(), *vcnt = pc.vcnt();
        size_t cnt = 0;
        for (size_t blk = b0; blk < b1; ++blk) {
            voff[blk] = (uint32_t)cnt; // (relative to the task's place in the stream: made absolute below)
            const size_t cnt0 = cnt, base = blk * SP_BLOCK;
            for (int w = 0; w < 64; ++w) {
                const size_t i0 = base + (size_t)w * 32;
                uint32_t bits = 0;
                if (i0 + 32 <= pc.n) cnt = g_pack32(in + i0, out, cnt, &bits);
                else
                    for (int i = 0; i < 32 && i0 + i < pc.n; ++i) {
                        const uint32_t v = in[i0 + i];
                        if (v != 0u) { out[cnt++] = v; bits |= 1u << i; }
                    }
                masks[blk * 64 + w] = bits;
            }
            vcnt[blk] = (uint32_t)(cnt - cnt0);
        }
        const size_t at = pc.cursor.fetch_add(cnt);
        memcpy(pc.h_vals + at, out, cnt * 4);
        for (size_t blk = b0; blk < b1; ++blk) voff[blk] += (uint32_t)at;
    }

    void worker()
    {
        unsigned long long seen = 0;
        for (;;) {
            Piece *pc;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return quit || epoch != seen; });
                if (quit) return;
                seen = epoch;
                pc = job;
                ++pc->drainers; // (counted under the lock that submit() holds while it resets the job fields: the two never overlap)
            }
            drain(pc);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pc->drainers == 0) cv_done.notify_all();
            }
        }
    }
    void drain(Piece *pc)
    {
        for (;;) {
            const int s = pc->next.fetch_add(1);
            if (s >= pc->segs) break;
            pack_task(*pc, s);
            if (pc->remaining.fetch_sub(1) == 1) { std::lock_guard<std::mutex> lk(mu); cv_done.notify_all(); }
        }
    }

    BcdSparseUploader()
    {
        int t = (int)std::thread::hardware_concurrency();
        if (const char *e = getenv("BCD_HIP_UPLOAD_THREADS")) t = atoi(e) + 1;
        else t = std::min(16, std::max(2, t / 2));
        for (int i = 0; i + 1 < t; ++i) pool.emplace_back([this] { worker(); }); // (the caller is the t-th packer)
    }
    ~BcdSparseUploader()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_work.notify_all();
        for (auto &th : pool) th.join();
        for (Piece &p : buf) release(p);
    }
    static void release(Piece &p)
    {
        if (p.ev) (void)hipEventDestroy(p.ev);
        if (p.h_meta) (void)hipHostFree(p.h_meta);
        if (p.h_vals) (void)hipHostFree(p.h_vals);
        if (p.d_meta) (void)hipFree(p.d_meta);
        if (p.d_vals) (void)hipFree(p.d_vals);
        p.ev = nullptr; p.ev_pending = false;
        p.h_meta = p.h_vals = p.d_meta = p.d_vals = nullptr;
        p.cap_floats = p.cap_blk = 0;
    }
    hipError_t size_for(Piece &p, size_t n)
    {
        if (p.cap_floats >= n) return hipSuccess;
        release(p);
        const size_t nblk = (n + SP_BLOCK - 1) / SP_BLOCK;
        hipError_t e;
        if ((e = hipHostMalloc((void **)&p.h_meta, nblk * 66 * 4)) != hipSuccess) return e;
        if ((e = hipHostMalloc((void **)&p.h_vals, (nblk * SP_BLOCK + 64) * 4)) != hipSuccess) return e;
        if ((e = hipMalloc((void **)&p.d_meta, nblk * 66 * 4)) != hipSuccess) return e;
        if ((e = hipMalloc((void **)&p.d_vals, (nblk * SP_BLOCK + 64) * 4)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&p.ev, hipEventDisableTiming)) != hipSuccess) return e;
        p.cap_floats = n; p.cap_blk = nblk;
        return hipSuccess;
    }

    // start packing `n` floats at `src` (asynchronously; at most one packing job at a time)
    hipError_t submit(const float *src, size_t n)
    {
        Piece &p = buf[cur];
        if (p.ev_pending) { hipError_t e = hipEventSynchronize(p.ev); if (e != hipSuccess) return e; p.ev_pending = false; } // its last upload has left the staging
        hipError_t e = size_for(p, std::max(n, SP_PIECE));
        if (e != hipSuccess) return e;
        {
            // A pool thread that was handed this buffer for an EARLIER job may still be on its way into (or out of) drain(): it would claim from
            // `next` while the fields below change (round-4 ADVICE: a stale index below the new `segs` packs a task twice, `remaining` hits
            // zero early, the cursor moves twice).  Wait until nobody is inside, then re-arm the job under the same lock the workers take to enter.
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return p.drainers == 0; });
            p.src = src; p.n = n; p.nblk = (n + SP_BLOCK - 1) / SP_BLOCK;
            p.segs = (int)((p.nblk + SP_TASK - 1) / SP_TASK);
            p.cursor.store(0);
            p.remaining.store(p.segs);
            p.next.store(0);
            job = &p;
            ++epoch;
        }
        cv_work.notify_all();
        return hipSuccess;
    }
    // the caller packs along, waits for the job, then enqueues the transfers and the unpack kernel of the piece on `st`
    hipError_t flush(float *dst, hipStream_t st)
    {
        Piece &p = buf[cur];
        drain(&p);
        { std::unique_lock<std::mutex> lk(mu); cv_done.wait(lk, [&] { return p.remaining.load() == 0; }); }
        cur = (cur + 1) % SP_BUFFERS;
        const size_t nonzero = p.cursor.load();
        raw_bytes += (long long)p.n * 4;
        if (nonzero * 10 > p.n * 6) { // more than 60 % of the values count: the packed form saves too little -- this piece and the rest of the frame travel as they are
            dense = true;
            sent_bytes += (long long)p.n * 4;
            return hipMemcpyAsync(dst, p.src, p.n * 4, hipMemcpyHostToDevice, st);
        }
        // two transfers per piece, both from pinned memory by the copy engines (a kernel reading the pinned buffers itself reached 38 GB/s
        // of the link's 56: 64-byte read requests; one transfer per packing thread cost more in calls than the packing saved)
        hipError_t e;
        if ((e = hipMemcpyAsync(p.d_meta, p.h_meta, p.nblk * 66 * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
        if (nonzero && (e = hipMemcpyAsync(p.d_vals, p.h_vals, nonzero * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
        sent_bytes += (long long)(p.nblk * 66 * 4) + (long long)nonzero * 4;
        hipLaunchKernelGGL(k_sparse_unpack, dim3((unsigned)p.nblk), dim3(64), 0, st, p.d_meta, p.d_meta + p.nblk * 64, p.d_meta + p.nblk * 65,
                           reinterpret_cast<const float *>(p.d_vals), dst, p.n);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = hipEventRecord(p.ev, st)) != hipSuccess) return e;
        p.ev_pending = true;
        return hipSuccess;
    }
};

BcdSparseUploader *bcd_sparse_create() { return new (std::nothrow) BcdSparseUploader(); }
void bcd_sparse_destroy(BcdSparseUploader *u) { delete u; }
void bcd_sparse_frame_begin(BcdSparseUploader *u) { u->dense = false; u->raw_bytes = 0; u->sent_bytes = 0; }
void bcd_sparse_frame_bytes(const BcdSparseUploader *u, long long *raw, long long *sent) { *raw = u->raw_bytes; *sent = u->sent_bytes; }

// `n` floats from host memory `src` to device memory `dst` (both 16-byte aligned) on `st`, in pieces: piece k + 1 is packed while piece k
// travels.  Returns when everything is enqueued; the host buffer must stay valid until `st` has passed these operations (dense images are
// copied from it directly, like a plain hipMemcpyAsync from pageable memory).
hipError_t bcd_sparse_upload(BcdSparseUploader *u, float *dst, const float *src, size_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    // (the unpack kernel stores 16-byte groups: a destination that is not 16-byte aligned travels as a plain copy)
    if (((uintptr_t)dst & 15) != 0) { u->raw_bytes += (long long)n * 4; u->sent_bytes += (long long)n * 4; return hipMemcpyAsync(dst, src, n * 4, hipMemcpyHostToDevice, st); }
    size_t done = 0;
    hipError_t e;
    bool packing = false; // a job on buf[cur] is in flight (its threads read the caller's buffer: never return while it runs)
    auto settle = [&](hipError_t rc) {
        if (packing) { Piece &p = u->buf[u->cur]; u->drain(&p); std::unique_lock<std::mutex> lk(u->mu); u->cv_done.wait(lk, [&] { return p.remaining.load() == 0; }); }
        return rc;
    };
    if (!u->dense) {
        if ((e = u->submit(src, std::min(n, SP_PIECE))) != hipSuccess) return e;
        packing = true;
    }
    while (done < n) {
        const size_t len = std::min(n - done, SP_PIECE);
        if (u->dense) { // (found out on an earlier piece)
            u->raw_bytes += (long long)(n - done) * 4; u->sent_bytes += (long long)(n - done) * 4;
            return settle(hipMemcpyAsync(dst + done, src + done, (n - done) * 4, hipMemcpyHostToDevice, st));
        }
        // this piece is being packed into buf[cur]; finish it and send it, then start on the next one -- whose packing overlaps this piece's transfer
        packing = false; // (flush waits for the job itself)
        if ((e = u->flush(dst + done, st)) != hipSuccess) return e;
        done += len;
        if (done < n && !u->dense) {
            if ((e = u->submit(src + done, std::min(n - done, SP_PIECE))) != hipSuccess) return e;
            packing = true;
        }
    }
    return settle(hipSuccess);
}
