// bcd_api.hip -- implementation of the C ABI declared in include/bcd_hip.h: context, workspace,
// the per-scale driver (Denoiser::denoise, src/core/Denoiser.cpp:84-212) and the multiscale driver
// (MultiscaleDenoiser::denoise, src/core/MultiscaleDenoiser.cpp:31-136) on device-resident images.
#include "../../include/bcd_hip.h"
#include "bcd_common.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <utility>
#include <vector>

// launchers implemented in the k_*.hip files
size_t bcd_pairdist_lds_bytes(int D, int b);
hipError_t bcd_launch_pairdist(const float *, const float *, int, int, int, int, float *, uint8_t *, int, int *, float, hipStream_t);
hipError_t bcd_launch_uniform_n(const float *, int64_t, int *, hipStream_t);
hipError_t bcd_launch_compare_planes(const float *, const uint8_t *, const float *, const uint8_t *, int64_t, unsigned long long *, hipStream_t);
hipError_t bcd_launch_selftest_div(uint32_t, int, int, unsigned long long *, hipStream_t);
hipError_t bcd_launch_masks(const float *, const uint8_t *, int, int, int, int, float, uint32_t *, int32_t *, uint32_t *, hipStream_t,
                            const BcdBorderline *, const float *, const float *, int);
hipError_t bcd_launch_masks_finish(int, int, int, float, uint32_t *, int32_t *, uint32_t *, hipStream_t, const BcdBorderline *, const float *, const float *, int);
int bcd_pairdist_rw_supported(int D);
hipError_t bcd_launch_pairdist_rw(const float *, const float *, int, int, int, int, void * /* binary16 T planes */, uint8_t *, int *, float, hipStream_t);
hipError_t bcd_launch_pairdist_rw_rows(const float *, const float *, int, int, int, int, void *, uint8_t *, int *, float, int, int, hipStream_t);
hipError_t bcd_launch_pairdist_rw_ratio(const float *, const float *, int, int, int, int, void *, uint8_t *, int *, float, unsigned int *, hipStream_t);
int bcd_pairdist_rw_tile_lines();
struct BcdSparseUploader;
BcdSparseUploader *bcd_sparse_create();
void bcd_sparse_destroy(BcdSparseUploader *);
void bcd_sparse_frame_begin(BcdSparseUploader *);
void bcd_sparse_frame_bytes(const BcdSparseUploader *, long long *, long long *);
hipError_t bcd_sparse_upload(BcdSparseUploader *, float *, const float *, size_t, hipStream_t);
void bcd_bayes27_set_strict_eigensolver(int on);
hipError_t bcd_launch_pairdist_rw_counting(const float *, const float *, int, int, int, int, void *, uint8_t *, int *, float, unsigned long long *, hipStream_t);
hipError_t bcd_launch_spike_rows(const float *, const float *, const float *, const float *, int, int, int, float, float *, float *, float *, float *, int, int,
                                 hipStream_t);
hipError_t bcd_launch_max_rel_dev(const float *, const float *, const uint8_t *, const uint8_t *, int, int, int, unsigned int *, hipStream_t);
hipError_t bcd_launch_window_distances(const float *, const uint8_t *, int, int, int, int, int, int, float *, hipStream_t);
hipError_t bcd_launch_pixel_cov(const float *, const float *, int64_t, float *, hipStream_t);
hipError_t bcd_launch_pixel_cov_clear(const float *, const float *, int64_t, float *, float *, int32_t *, hipStream_t);
hipError_t bcd_launch_scale_begin(int *, int, int, int, int *, int, int *, int, hipStream_t);
hipError_t bcd_launch_finalize_band(const float *, const int32_t *, int, int, int, const float *, const int32_t *, const float *, const int32_t *, float *,
                                    hipStream_t);
hipError_t bcd_launch_finalize(const float *, const int32_t *, int64_t, float *, hipStream_t);
hipError_t bcd_launch_zero_bad(float *, int64_t, hipStream_t);
hipError_t bcd_launch_downscale(int, const float *, int, int, int, float *, hipStream_t);
hipError_t bcd_launch_downscale_cov(const float *, const float *, int, int, float *, hipStream_t);
hipError_t bcd_launch_interpolate(int, const float *, int, int, int, float *, int, int, hipStream_t);
hipError_t bcd_launch_merge_interpolate(const float *, const float *, int, int, int, float *, int, int, hipStream_t);
hipError_t bcd_launch_spike(const float *, const float *, const float *, const float *, int, int, int, float, float *, float *,
                            float *, float *, hipStream_t);
hipError_t bcd_launch_accumulate_samples(const float *, const float *, int64_t, int, int, float, float, float *, float *, float *, float *, hipStream_t);
hipError_t bcd_launch_active_init(const int32_t *, int, int, int, int, int, float, uint32_t, int, uint8_t *, hipStream_t);
hipError_t bcd_launch_active_round(const uint32_t *, const int32_t *, uint8_t *, int, int, int, int, int, uint32_t, int, int, int, int *, hipStream_t);
hipError_t bcd_launch_mark_deps(const uint32_t *, const int32_t *, uint8_t *, uint32_t *, int, int, int, int, int, uint32_t, int, int, int, int *, hipStream_t);
hipError_t bcd_launch_mark_round(const uint32_t *, uint8_t *, int, int, int, int, int, int, int *, hipStream_t);
hipError_t bcd_launch_sum_counter_lines(const int *, int, int *, hipStream_t, long long * = nullptr, const int * = nullptr, int = 0);
hipError_t bcd_launch_active_lists(const uint8_t *, const int32_t *, int64_t, int64_t, int, int32_t *, int32_t *, int32_t *, hipStream_t, const long long *);
hipError_t bcd_launch_jacobi27_batch(const float *, int, int *, int, float *, float *, hipStream_t, float = 1e-12f, float * = nullptr, const int * = nullptr, int = 0);
size_t bcd_bayes_lds_bytes(int w, int b);
size_t bcd_bayes_scratch_bytes_per_block(int w, int b);
size_t bcd_bayes27_record_bytes();
hipError_t bcd_launch_bayes27(const float *, const float *, const uint32_t *, const int32_t *, int, int, int *, int, int, int, int, float, float *, float *,
                              int32_t *, int *, hipStream_t, int, const int *d_nb_items);
hipError_t bcd_launch_bayes27_redo(const float *, const float *, const uint32_t *, const int32_t *, int, int, int *, int, int, int, int, float, float *, float *,
                                   int32_t *, hipStream_t);
hipError_t bcd_launch_bayes_strong(const float *, const float *, const uint32_t *, const int32_t *, const int32_t *, int *, int, int, int, int,
                                   int, float, float *, int32_t *, float *, size_t, hipStream_t);
hipError_t bcd_launch_bayes_weak(const float *, const uint32_t *, const int32_t *, const int32_t *, int, int, int, int, int, float *,
                                 int32_t *, hipStream_t);
hipError_t bcd_launch_bayes_weak_tiles(const float *, const uint32_t *, const uint8_t *, const int32_t *, int, int, int, int, float *, int32_t *, hipStream_t, int, int, const long long *);

namespace {

constexpr int MAX_SCALES = 16;
constexpr int ROUND_BATCH = 16;
constexpr int MAX_EVENT_PAIRS = 4096;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

} // namespace

// everything one scale's pipeline needs: a multiscale run drives one Work per scale concurrently (own stream, own host
// thread), because the scales are independent until the merge and the coarse ones cannot fill 256 CUs on their own
struct Work {
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    DevBuf T, Cn, mask, fwd, nsim, state, strong, weak, counters, cnt_lines, work_q, pixcov, sum, cnt, gscratch, dep, tmp_lo, border, ratio_stats; // grow-only
    int border_capacity = 0;       // entries of `border` offered to the last fast similarity pass (0: the exact kernels ran)
    int rounds_hint = 0;           // marking launches the last problem needed
    int last_batch = 0;            // launches of the batch active_step_enqueue left in flight
    bool dep_ready = false;        // dependency lists of the current marking problem are in `dep` (reset by active_init)
    const void *dep_mask = nullptr, *dep_state = nullptr; // ... extracted for these buffers (another problem on the same context rebuilds them)
    int32_t *h_counters = nullptr; // pinned
    bool initialised = false;      // set once every stream / event / pinned buffer below exists
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool; // pair-distance kernel timing
    int ev_used = 0;
    hipEvent_t ev_stage[4] = { nullptr, nullptr, nullptr, nullptr };
    hipEvent_t ev_done = nullptr;
    hipEvent_t ev_built = nullptr; // this scale's pyramid level is complete
    hipStream_t aux = nullptr;     // side stream: the fallback-pixel kernel runs beside the (latency-bound) full estimate kernel
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_pixcov = nullptr; // the per-pixel covariances (side stream, beside the distance kernel) are complete
    hipEvent_t ev_counts = nullptr; // the list lengths of bayes() are on the host
    // (round 6) full-estimate items of the last frame of this geometry on this workspace: the next frame's first chunk of estimate kernels is launched for
    // that many (+ 1/8) BEFORE the host knows the new count
    int strong_hint = 0, strong_hint_W = 0, strong_hint_H = 0;
    // approximate distance planes computed ahead of similarity() by a caller that streams the frame in (bcd_hip_denoise_host_ex): valid for
    // exactly this problem; similarity() consumes the note
    struct { bool ready = false; const float *hist = nullptr, *ns = nullptr; int W = 0, H = 0, D = 0, b = 0; float tau = 0.f, uni_n = 0.f; } planes;
    // uniform-sample-count speculation of the approximate distance kernel (similarity()): did the last frames on this workspace fail it?
    bool nonuniform = false;
    bool speculated = false;       // the current pass launched the uniform kernel on the first pixel's count, unchecked by the host
    // the RATIO form of the distance kernel raised its absolute-error flag on frames of these sizes on this workspace: the reference's operations serve them (a small
    // set, oldest replaced: serialised scales share one workspace, a caller may alternate frame sizes)
    struct { int W = 0, H = 0; } ratio_declined[4];
    int ratio_declined_next = 0;
    bool ratio_is_declined(int W, int H) const { for (const auto &k : ratio_declined) if (k.W == W && k.H == H) return true; return false; }
    bool ratio_used = false;          // the current pass ran the RATIO form of the distance kernel (general sample counts)
    int ratio_W = 0, ratio_H = 0;        // ... on a frame of this size
    // (round 4) k_scale_begin cleared these at the head of the scale's stream: the first user takes them as they are, a repeated use (second
    // similarity attempt, second marking batch, second chunk of a long list) clears its own as before
    bool clean_flags = false, clean_lines = false, clean_dc = false, clean_wq = false;
    // the redo kernel of the register-resident finish (k_bayes27w<2> over the -- normally empty -- list of items whose sweep inverse failed its
    // checks) is only launched when the list's counter, read with the scale's last synchronisation, says so
    struct { bool pending = false; int first = 0, n = 0, cus = 0; } redo;
};

struct bcd_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    bool profiling = false;
    bool concurrent_scales = true;
    bool fast_similarity = true; // approximate distance planes + exact re-evaluation at the threshold (k_similarity_fast.hip)
    int num_cus = 256;
    int cu_share_pct = 100;  // bcd_hip_set_cu_share
    // share of the CU slots the coarse scales' persistent estimate kernels take inside bcd_hip_denoise (bayes()); adjusted from call to
    // call on the same geometry so that the coarse scales end shortly before the finest one (see bcd_hip_denoise)
    int coarse_share = 25;
    int64_t share_key = 0;          // geometry the current value was tuned on
    std::mutex err_mutex;
    std::string err;
    bcd_hip_scale_stats stats[MAX_SCALES];
    Work main;               // bound to `stream`
    Work extra[MAX_SCALES];  // lazily created streams for scales 1.. of a multiscale run
    DevBuf tmp_lo;
    DevBuf pyr[MAX_SCALES][5]; // colours, nsamples, hist, cov, out
    DevBuf host_stage[9];      // host-buffer entry points: device copies of the four inputs, the output, the prefiltered inputs (grow-only)
    hipEvent_t ev_pyramid = nullptr;
    hipStream_t upload_stream = nullptr;        // host-buffer entry points: uploads run beside the kernels of the lines that have arrived
    hipStream_t upload_stream2 = nullptr;       // ... colours and covariances beside the histogram pieces (helper thread)
    hipEvent_t ev_upload2 = nullptr;
    std::vector<hipEvent_t> ev_upload;
    bool stream_uploads = true;                 // BCD_HIP_STREAM_UPLOADS=0: upload everything, then compute
    // (round 4) the histogram image crosses PCIe without its zeros (bcd_sparse_upload.hip); BCD_HIP_SPARSE_UPLOAD=0: plain copies
    bool sparse_uploads = true;
    BcdSparseUploader *sparse = nullptr;
    long long upload_raw_bytes = 0, upload_sent_bytes = 0; // histogram image of the last host-buffer frame: as it is / as it travelled
    // progress reporting (IDenoiser::setProgressCallback; Denoiser.cpp:181-192 of the reference): every scale adds its share when
    // its marking is done and when its estimate is done; calls are serialised and monotone
    bcd_hip_progress_fn progress_fn = nullptr;
    void *progress_user = nullptr;
    std::mutex progress_mutex;
    double progress_done = 0.0, progress_total = 0.0;
    // bcd_hip_denoise_begin / _wait (round 6): one frame of this context in flight on a worker thread of its own, so that a caller can keep a second
    // context busy meanwhile (frames of a sequence, AOV passes: the distance kernels of one frame fill the chip under the latency-bound tail of another)
    struct Async {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        bool has_job = false, in_flight = false, quit = false;
        int rc = 0;
        const float *col = nullptr, *ns = nullptr, *hist = nullptr, *cov = nullptr;
        float *out = nullptr;
        int W = 0, H = 0, D = 0, S = 0;
        bcd_hip_params prm;
    } async;
};

namespace {

void set_err(bcd_hip_ctx *ctx, const std::string &msg)
{
    if (!ctx) return;
    std::lock_guard<std::mutex> lock(ctx->err_mutex);
    ctx->err = msg;
}

// every entry point that allocates or launches runs on the context's device and leaves the caller's current device as it was
struct DeviceGuard {
    int prev = -1, dev;
    bool ok = true;
    explicit DeviceGuard(const bcd_hip_ctx *ctx) : dev(ctx ? ctx->device : -1)
    {
        if (dev < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (dev >= 0 && prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};
#define DEVICE_GUARD(ctx)                                                                                             \
    DeviceGuard guard__(ctx);                                                                                         \
    if (!guard__.ok) { set_err((ctx), "hipSetDevice failed"); return BCD_HIP_EDEVICE; }

#define HIPCHK(ctx, expr)                                                                                             \
    do {                                                                                                              \
        hipError_t e__ = (expr);                                                                                      \
        if (e__ != hipSuccess) {                                                                                      \
            set_err((ctx), std::string(#expr) + ": " + hipGetErrorString(e__));                                       \
            return BCD_HIP_EDEVICE;                                                                                   \
        }                                                                                                             \
    } while (0)

#define RCCHK(expr)                                                                                                   \
    do {                                                                                                              \
        int rc__ = (expr);                                                                                            \
        if (rc__ != BCD_HIP_OK) return rc__;                                                                          \
    } while (0)

int ensure(bcd_hip_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (b.bytes >= bytes && b.p) return BCD_HIP_OK;
    if (b.p) { HIPCHK(ctx, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    size_t want = bytes + bytes / 16 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) { set_err(ctx, "hipMalloc failed: " + std::string(hipGetErrorString(e))); b.p = nullptr; return BCD_HIP_ENOMEM; }
    b.bytes = want;
    return BCD_HIP_OK;
}

// `share` of the frame's work (pixels of a scale, weighted) is done
void progress_add(bcd_hip_ctx *ctx, double share)
{
    if (!ctx->progress_fn || !(ctx->progress_total > 0.0)) return;
    std::lock_guard<std::mutex> lock(ctx->progress_mutex);
    ctx->progress_done = std::min(ctx->progress_total, ctx->progress_done + share);
    ctx->progress_fn((float)(ctx->progress_done / ctx->progress_total), ctx->progress_user);
}

int bad(bcd_hip_ctx *ctx, const char *msg)
{
    set_err(ctx, msg);
    return BCD_HIP_EINVAL;
}

int check_params(bcd_hip_ctx *ctx, int W, int H, int D, const bcd_hip_params *prm)
{
    if (!prm) return bad(ctx, "null parameters");
    if (W <= 0 || H <= 0 || D <= 0) return bad(ctx, "empty input image");            // Denoiser.cpp:294-320
    if (prm->patch_radius < 0 || prm->search_radius < 0) return bad(ctx, "negative radius");
    if (W < 2 * prm->patch_radius + 1 || H < 2 * prm->patch_radius + 1) return bad(ctx, "image smaller than a patch");
    if (D > 255) { set_err(ctx, "histogram depth > 255 is not supported"); return BCD_HIP_EUNSUPPORTED; }
    int side = 2 * prm->search_radius + 1;
    if ((side * side + 31) / 32 > 32) { set_err(ctx, "search radius > 15 is not supported"); return BCD_HIP_EUNSUPPORTED; }
    if (bcd_bayes_lds_bytes(prm->patch_radius, prm->search_radius) > 160 * 1024) {
        set_err(ctx, "patch/search radius combination exceeds the 160 KiB LDS working set");
        return BCD_HIP_EUNSUPPORTED;
    }
    if ((int64_t)W * H >= (1ll << 31) / (D > 6 ? D : 6)) return bad(ctx, "image too large for 32-bit DeepImage indices");
    if (prm->use_random_pixel_order < 0 || prm->use_random_pixel_order > 2) return bad(ctx, "pixel order must be 0 (scanline), 1 (seeded random) or 2 (strips)");
    if (prm->use_random_pixel_order == 2 && (W > 8191 || H > 8191 || prm->patch_radius > 3 || prm->search_radius < 1))
        return bad(ctx, "the strip order supports frames up to 8191 x 8191, patch radius <= 3, search radius >= 1");
    return BCD_HIP_OK;
}

// bytes of the count planes of a scale.  ONE place: the host-buffer entry point computes planes ahead of similarity(), and a larger request there
// would free them (round 6: it happened when one of the two grew, found by the environment-switch test on a fresh context)
size_t count_plane_bytes(size_t npix, int nd) { return npix * (size_t)nd; }

// did the last similarity() pass on this workspace leave the range flag raised or overflow its borderline list?  (valid after the
// stream has been synchronised; the caller then repeats the pass with exact_mode = 1)
// a user of the workspace's counters / flags / work queues / sub-counter lines outside the scale chain (self-tests, the eigensolver entry point): whatever
// k_scale_begin left clean is not clean any more
void touch(Work &wk) { wk.clean_flags = wk.clean_lines = wk.clean_dc = wk.clean_wq = false; }

bool similarity_needs_redo(const Work &wk)
{
    return wk.h_counters[40] != 0 || (wk.border_capacity > 0 && (wk.h_counters[42] != 0 || wk.h_counters[43] > wk.border_capacity));
}

// ... and with which kernels: 0 = no redo; 3 = the approximate kernels again with the general (non-uniform) formula -- the only complaint
// was that the sample counts are not one power of two (flag bit 1 of k_pairdist_rw); 1 = the exact kernels.  Also keeps the workspace's
// memory of whether its frames have uniform counts (valid after the stream has been synchronised).
int similarity_redo_mode(Work &wk)
{
    const int flag = wk.h_counters[40];
    const bool fast = wk.border_capacity > 0;
    const bool other_count = fast && wk.h_counters[42] != 0; // a pixel carries another sample count than the uniform kernel was launched for
    const bool overflow = fast && wk.h_counters[43] > wk.border_capacity;
    if (fast && (wk.speculated || other_count)) wk.nonuniform = other_count;
    if (flag == 0 && !other_count && !overflow) return 0;
    if (fast && wk.ratio_used && (flag & 4) != 0) { // the RATIO form's absolute-error check: the reference's operations serve this frame size on this workspace from now on
        if (!wk.ratio_is_declined(wk.ratio_W, wk.ratio_H)) { wk.ratio_declined[wk.ratio_declined_next] = { wk.ratio_W, wk.ratio_H }; wk.ratio_declined_next = (wk.ratio_declined_next + 1) & 3; }
        if ((flag & ~4) == 0) return 3;
    }
    return (flag == 0 && other_count) ? 3 : 1; // (a void launch has no meaningful list count: other_count alone decides)
}

// can the approximate-planes path serve this problem?  (w = 1, a supported depth, a threshold binary16 can decide: bcd_common.h)
bool fast_similarity_applies(const bcd_hip_ctx *ctx, int D, int w, float tau)
{
    return ctx->fast_similarity && w == 1 && bcd_pairdist_rw_supported(D) && tau >= BCD_APPROX_TAU_MIN && tau <= BCD_APPROX_TAU_MAX;
}

// exact_mode: 0 = production kernels, flags checked here (one stream synchronisation); 1 = exact kernels with the compiler's division;
// 2 = production kernels, flags copied to wk.h_counters[40] / [43] but NOT checked: the caller validates after its own
// synchronisation with similarity_needs_redo() / similarity_redo_mode(); 3 = like 2 with the general (non-uniform) formula forced.
// Production kernels: w = 1 and a supported depth -> approximate planes (k_pairdist_rw) + exact verification of the borderline
// pairs; otherwise the exact planes with the scale-free division (k_pairdist<FAST>).
int similarity(bcd_hip_ctx *ctx, Work &wk, const float *d_hist, const float *d_ns, int W, int H, int D, int w, int b, float tau,
               uint32_t *d_mask, int32_t *d_count, int exact_mode = 0)
{
    const size_t npix = (size_t)W * H;
    const int nd = bcd_delta_count(b);
    RCCHK(ensure(ctx, wk.T, npix * nd * sizeof(float)));
    RCCHK(ensure(ctx, wk.Cn, count_plane_bytes(npix, nd)));
    RCCHK(ensure(ctx, wk.fwd, npix * ((nd + 31) / 32) * sizeof(uint32_t)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (wk.ev_used < MAX_EVENT_PAIRS) {
        if (wk.ev_used == (int)wk.ev_pool.size()) {
            hipEvent_t a, c;
            HIPCHK(ctx, hipEventCreate(&a));
            HIPCHK(ctx, hipEventCreate(&c));
            wk.ev_pool.emplace_back(a, c);
        }
        e0 = wk.ev_pool[wk.ev_used].first;
        e1 = wk.ev_pool[wk.ev_used].second;
        ++wk.ev_used;
    }
    RCCHK(ensure(ctx, wk.counters, 64 * sizeof(int32_t)));
    int *d_flag = (int *)wk.counters.p + 40; // [0] range / count flag, [1] uniform-count scan, [3] borderline pairs
    // planes of exactly this problem already computed by the caller (its launches raised the flags in d_flag[0] themselves)?
    const bool pre = wk.planes.ready && exact_mode != 1 && wk.planes.hist == d_hist && wk.planes.ns == d_ns && wk.planes.W == W && wk.planes.H == H &&
                     wk.planes.D == D && wk.planes.b == b && wk.planes.tau == tau && w == 1;
    const bool planes_kept = wk.planes.ready; // (k_scale_begin kept words [0] and [2] for the launches that made them)
    wk.planes.ready = false;
    if (wk.clean_flags) { // cleared by k_scale_begin, which kept the words of planes computed ahead ...
        wk.clean_flags = false;
        if (planes_kept && !pre) { // ... of ANOTHER problem (serialised scales: the coarsest scale runs first on this workspace): their flags are not this pass's
            HIPCHK(ctx, hipMemsetAsync(d_flag, 0, sizeof(int), wk.stream));
            HIPCHK(ctx, hipMemsetAsync(d_flag + 2, 0, sizeof(int), wk.stream));
        }
    } else if (pre) { // (flags [0] and [2] belong to the launches that made the planes)
        HIPCHK(ctx, hipMemsetAsync(d_flag + 1, 0, sizeof(int), wk.stream));
        HIPCHK(ctx, hipMemsetAsync(d_flag + 3, 0, sizeof(int), wk.stream));
    } else HIPCHK(ctx, hipMemsetAsync(d_flag, 0, 4 * sizeof(int), wk.stream));
    wk.h_counters[40] = 0;
    wk.h_counters[42] = 0;
    wk.h_counters[43] = 0;
    wk.border_capacity = 0;
    wk.ratio_used = false;
    wk.ratio_W = W; wk.ratio_H = H;
    // fixed samples per pixel, a power of two (the usual case): the distance kernel drops the sample-count products (exactly,
    // see k_pairdist).  One small reduction and one host round trip at the head of the chain (~30 us).
    float uni_n = pre ? wk.planes.uni_n : 0.f;
    const bool fast_path = exact_mode != 1 && fast_similarity_applies(ctx, D, w, tau);
    wk.speculated = false;
    auto scan_uniform_count = [&]() -> int { // one small reduction and one host round trip (~30 us): uni_n = the frame's power-of-two count, or 0
        HIPCHK(ctx, bcd_launch_uniform_n(d_ns, (int64_t)npix, d_flag + 1, wk.stream));
        HIPCHK(ctx, hipMemcpyAsync(wk.h_counters + 41, d_flag + 1, sizeof(int), hipMemcpyDeviceToHost, wk.stream));
        HIPCHK(ctx, hipMemcpyAsync(wk.h_counters + 42, d_ns, sizeof(float), hipMemcpyDeviceToHost, wk.stream));
        HIPCHK(ctx, hipStreamSynchronize(wk.stream));
        float n0;
        memcpy(&n0, wk.h_counters + 42, sizeof(n0));
        wk.h_counters[42] = 0;
        int e = 0;
        if (wk.h_counters[41] == 0 && n0 >= 1.f && n0 <= 65536.f && frexpf(n0, &e) == 0.5f) uni_n = n0;
        return BCD_HIP_OK;
    };
    if (fast_path && !pre) {
        // No scan and no round trip at the head of the chain: the approximate kernel takes the first pixel's count as THE count and checks
        // every pixel against it itself (flag bit 1 -> the pass is repeated with the general formula).  A workspace whose LAST frame was not
        // uniform (adaptive sampling) looks first (round 5: it used to take the general formula blindly and look again every 32nd pass, which
        // cost a uniform frame that followed frames with mixed counts the general formula for up to 31 passes).
        if (exact_mode != 3 && !wk.nonuniform) { uni_n = -1.f; wk.speculated = true; }
        else if (exact_mode != 3) {
            RCCHK(scan_uniform_count());
            wk.nonuniform = uni_n == 0.f;
        }
    } else if (exact_mode != 1 && !pre)
        RCCHK(scan_uniform_count());
    // the approximate path keeps its T plane in binary16: thresholds it cannot decide safely take the exact kernels (bcd_common.h)
    const bool fast = fast_path;
    if (fast) {
        const int capacity = (int)std::min<size_t>(std::max<size_t>(npix, 1u << 16), 1u << 28);
        RCCHK(ensure(ctx, wk.border, (size_t)capacity * sizeof(uint2)));
        wk.border_capacity = capacity;
        BcdBorderline bl = { 0.f, (uint2 *)wk.border.p, d_flag + 3, capacity };
        // General sample counts (adaptive sampling, 24 spp, ...; src/core/DenoisingUnit.cpp:371-383 handles any n1, n2): the RATIO form of the dense kernel
        // (round 6, k_similarity_fast.hip) evaluates them at the cost of uniform ones -- 1.83 ms at 1080p against 1.73 for the uniform kernel, 3.0 ms for the
        // reference's operations and 1.97 + 0.07 ms for the own-list kernel of round 5, which it replaced -- and checks afterwards that the absolute errors it
        // adds stay inside the verified band (flag bit 2: the pass is then repeated with the reference's operations, and the workspace remembers the size).
        const bool use_ratio = !pre && uni_n == 0.f && !wk.ratio_is_declined(W, H);
        wk.ratio_used = use_ratio;
        if (pre) { if (e0) --wk.ev_used; } // (nothing to time: the planes are there)
        else if (use_ratio) {
            RCCHK(ensure(ctx, wk.ratio_stats, 128 * sizeof(unsigned int)));
            if (e0) HIPCHK(ctx, hipEventRecord(e0, wk.stream));
            HIPCHK(ctx, bcd_launch_pairdist_rw_ratio(d_hist, d_ns, W, H, D, b, wk.T.p, (uint8_t *)wk.Cn.p, d_flag, tau, (unsigned int *)wk.ratio_stats.p, wk.stream));
            if (e1) HIPCHK(ctx, hipEventRecord(e1, wk.stream));
        } else {
            if (e0) HIPCHK(ctx, hipEventRecord(e0, wk.stream));
            HIPCHK(ctx, bcd_launch_pairdist_rw(d_hist, d_ns, W, H, D, b, wk.T.p, (uint8_t *)wk.Cn.p, d_flag, uni_n, wk.stream));
            if (e1) HIPCHK(ctx, hipEventRecord(e1, wk.stream));
        }
        HIPCHK(ctx, hipMemcpyAsync(wk.h_counters + 40, d_flag, sizeof(int), hipMemcpyDeviceToHost, wk.stream));
        HIPCHK(ctx, hipMemcpyAsync(wk.h_counters + 42, d_flag + 2, sizeof(int), hipMemcpyDeviceToHost, wk.stream)); // "another sample count" (plain-store flag)
        HIPCHK(ctx, bcd_launch_masks((const float *)wk.T.p, (const uint8_t *)wk.Cn.p, W, H, w, b, tau, d_mask, d_count, (uint32_t *)wk.fwd.p, wk.stream,
                                     &bl, d_hist, d_ns, D));
        HIPCHK(ctx, hipMemcpyAsync(wk.h_counters + 43, d_flag + 3, sizeof(int), hipMemcpyDeviceToHost, wk.stream));
        if (exact_mode == 0) {
            HIPCHK(ctx, hipStreamSynchronize(wk.stream));
            const int redo = similarity_redo_mode(wk);
            if (redo == 3) {
                RCCHK(similarity(ctx, wk, d_hist, d_ns, W, H, D, w, b, tau, d_mask, d_count, 3));
                HIPCHK(ctx, hipStreamSynchronize(wk.stream));
                if (similarity_needs_redo(wk) && similarity_redo_mode(wk) == 3) { // (the RATIO form declined: once more with the reference's operations)
                    RCCHK(similarity(ctx, wk, d_hist, d_ns, W, H, D, w, b, tau, d_mask, d_count, 3));
                    HIPCHK(ctx, hipStreamSynchronize(wk.stream));
                }
                if (similarity_needs_redo(wk)) return similarity(ctx, wk, d_hist, d_ns, W, H, D, w, b, tau, d_mask, d_count, 1);
            } else if (redo == 1)
                return similarity(ctx, wk, d_hist, d_ns, W, H, D, w, b, tau, d_mask, d_count, 1);
        }
        return BCD_HIP_OK;
    }
    if (e0) HIPCHK(ctx, hipEventRecord(e0, wk.stream));
    HIPCHK(ctx, bcd_launch_pairdist(d_hist, d_ns, W, H, D, b, (float *)wk.T.p, (uint8_t *)wk.Cn.p, exact_mode == 1 ? 0 : 1, d_flag, uni_n, wk.stream));
    if (e1) HIPCHK(ctx, hipEventRecord(e1, wk.stream));
    // the fast kernel flags inputs outside the range where its division is proven exact: redo with the compiler's division
    if (exact_mode != 1) HIPCHK(ctx, hipMemcpyAsync(wk.h_counters + 40, d_flag, sizeof(int), hipMemcpyDeviceToHost, wk.stream));
    if (exact_mode == 0) {
        HIPCHK(ctx, hipStreamSynchronize(wk.stream));
        if (wk.h_counters[40] != 0)
            HIPCHK(ctx, bcd_launch_pairdist(d_hist, d_ns, W, H, D, b, (float *)wk.T.p, (uint8_t *)wk.Cn.p, 0, d_flag, 0.f, wk.stream));
    }
    HIPCHK(ctx, bcd_launch_masks((const float *)wk.T.p, (const uint8_t *)wk.Cn.p, W, H, w, b, tau, d_mask, d_count, (uint32_t *)wk.fwd.p, wk.stream,
                                 nullptr, nullptr, nullptr, 0));
    return BCD_HIP_OK;
}

// one batch of marking launches on lines [row_begin, row_end); *undecided_out = pixels of those lines still undecided
// Two halves (round 6): active_step_enqueue launches the batch and the copy of its counters WITHOUT waiting -- with `d_total` it also leaves the rank's
// contribution to the all-reduced count on the device (k_sum_counter_lines; `with_verdict`: + 2^40 when this workspace's last similarity pass has to be
// repeated), so that the band driver's all-reduce follows in stream order and one synchronisation serves both; active_step_collect reads the counters
// after the caller's synchronisation.
int active_step_enqueue(bcd_hip_ctx *ctx, Work &wk, const uint32_t *d_mask, const int32_t *d_nsim, int W, int H, int w, int b, int row_begin,
                        int row_end, int random_order, uint32_t seed, int row_offset, uint8_t *d_state, long long *d_total, bool with_verdict)
{
    const int K = 3 * (2 * w + 1) * (2 * w + 1);
    RCCHK(ensure(ctx, wk.counters, 64 * sizeof(int32_t)));
    int *d_cnt = (int *)wk.counters.p;
    // every launch counts the pixels it leaves undecided into its own set of BCD_CNT_LINES sub-counters (one per cache line: a single
    // counter is a serial resource, k_active.hip); one small kernel folds them into d_cnt[launch] for the host
    constexpr size_t LINE_INTS = (size_t)BCD_CNT_LINES * BCD_CNT_STRIDE;
    RCCHK(ensure(ctx, wk.cnt_lines, ROUND_BATCH * LINE_INTS * sizeof(int)));
    int *d_lines = (int *)wk.cnt_lines.p;
    if (wk.clean_lines) wk.clean_lines = false; // (k_scale_begin)
    else {
        HIPCHK(ctx, hipMemsetAsync(d_lines, 0, ROUND_BATCH * LINE_INTS * sizeof(int), wk.stream));
        HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, ROUND_BATCH * sizeof(int), wk.stream));
    }
    // b = 6 / 12: dependency lists extracted once per marking problem, then cheap tile rounds (in-tile chains are resolved
    // inside a launch: few launches for a random order, more for the long chains of the scanline order);
    // other radii: the generic one-level-per-launch kernel
    const bool listed = (b == 6 || b == 12);
    int batch = 4;
    if (listed) {
        const int side = 2 * b + 1, words = (side * side + 31) / 32;
        const int iters = 8; // in-tile iterations per launch
        int i = 0;
        batch = random_order == 1 ? 3 : ROUND_BATCH;
        if (wk.dep_ready && (wk.dep_mask != (const void *)d_mask || wk.dep_state != (const void *)d_state)) wk.dep_ready = false;
        if (!wk.dep_ready) {
            RCCHK(ensure(ctx, wk.dep, (size_t)W * H * words * sizeof(uint32_t)));
            HIPCHK(ctx, bcd_launch_mark_deps(d_mask, d_nsim, d_state, (uint32_t *)wk.dep.p, W, H, b, K + 1, random_order, seed,
                                             row_begin, row_end, row_offset, d_lines + LINE_INTS * i++, wk.stream));
            wk.dep_ready = true;
            wk.dep_mask = d_mask;
            wk.dep_state = d_state;
            // first batch: what the previous marking problem of this workspace needed, plus one (frames of a sequence and
            // the bands of a frame behave alike), so that the usual case costs a single host round trip
            if (random_order == 1) batch = std::min(ROUND_BATCH, std::max(5, wk.rounds_hint + 1));
        }
        for (; i < batch; ++i)
            HIPCHK(ctx, bcd_launch_mark_round((const uint32_t *)wk.dep.p, d_state, W, H, b, row_begin, row_end, iters, d_lines + LINE_INTS * i, wk.stream));
    } else {
        for (int i = 0; i < batch; ++i)
            HIPCHK(ctx, bcd_launch_active_round(d_mask, d_nsim, d_state, W, H, b, K + 1, random_order, seed, row_begin, row_end, row_offset,
                                                d_lines + LINE_INTS * i, wk.stream));
    }
    HIPCHK(ctx, bcd_launch_sum_counter_lines(d_lines, batch, d_cnt, wk.stream, d_total, with_verdict ? (const int *)wk.counters.p + 40 : nullptr, wk.border_capacity));
    HIPCHK(ctx, hipMemcpyAsync(wk.h_counters, d_cnt, ROUND_BATCH * sizeof(int), hipMemcpyDeviceToHost, wk.stream));
    wk.last_batch = batch;
    return BCD_HIP_OK;
}

void active_step_collect(Work &wk, int *undecided_out, int *launches_out)
{
    const int batch = wk.last_batch;
    int n = batch;
    for (int i = 0; i < batch; ++i)
        if (wk.h_counters[i] == 0) { n = i + 1; break; }
    *undecided_out = wk.h_counters[n - 1];
    if (launches_out) *launches_out = n;
}

// one batch of marking launches on lines [row_begin, row_end); *undecided_out = pixels of those lines still undecided
int active_step(bcd_hip_ctx *ctx, Work &wk, const uint32_t *d_mask, const int32_t *d_nsim, int W, int H, int w, int b, int row_begin,
                int row_end, int random_order, uint32_t seed, int row_offset, bool first_pass, uint8_t *d_state, int *undecided_out,
                int *launches_out)
{
    (void)first_pass; // a hint of the C ABI ("every pixel is still undecided"); the dependency-list kernels do not need it
    RCCHK(active_step_enqueue(ctx, wk, d_mask, d_nsim, W, H, w, b, row_begin, row_end, random_order, seed, row_offset, d_state, nullptr, false));
    HIPCHK(ctx, hipStreamSynchronize(wk.stream));
    active_step_collect(wk, undecided_out, launches_out);
    return BCD_HIP_OK;
}

int active_set(bcd_hip_ctx *ctx, Work &wk, const uint32_t *d_mask, const int32_t *d_nsim, int W, int H, int w, int b, int row_begin,
               int row_end, float skip_prob, int random_order, uint32_t seed, uint8_t *d_state, int32_t *rounds_out)
{
    HIPCHK(ctx, bcd_launch_active_init(d_nsim, W, H, w, row_begin, row_end, skip_prob, seed, 0, d_state, wk.stream));
    wk.dep_ready = false;
    int rounds = 0;
    if (skip_prob > 0.f) {
        // every launch decides at least the earliest undecided pixel of the visiting order, so the iteration ends after at most
        // (number of main pixels) launches; stop only when a whole batch makes no progress (which would be an engine bug)
        int undecided = 1, before = INT_MAX;
        while (undecided != 0) {
            int n = 0;
            // (strip order: the key needs the frame's geometry, which travels in the seed argument; the skip draws keep the caller's seed)
            const uint32_t key_seed = random_order == 2 ? bcd_strip_order_seed(W, H, w, b) : seed;
            RCCHK(active_step(ctx, wk, d_mask, d_nsim, W, H, w, b, row_begin, row_end, random_order, key_seed, 0, rounds == 0 && skip_prob >= 1.f,
                              d_state, &undecided, &n));
            rounds += n;
            if (undecided != 0 && undecided >= before) { set_err(ctx, "marking fixed point made no progress"); return BCD_HIP_EDEVICE; }
            before = undecided;
        }
        wk.rounds_hint = rounds;
    }
    if (rounds_out) *rounds_out = rounds;
    return BCD_HIP_OK;
}

// lists of processed pixels + the estimate kernels.  The list lengths and the sum of |S| are copied to wk.h_counters[16..20):
// read them with bayes_counts() after the stream has been synchronised.
// w = 1: the full estimate is three kernels with a per-pixel record in HBM between them (k_bayes27.hip); the host reads the
// number of full-estimate pixels (one short round trip, the fallback kernel is already running on its side stream) to size the
// record buffer and to cut very long lists (-m 0) into chunks.  Other patch radii: one persistent kernel, no round trip.
// Only processed pixels of lines [row_begin, row_end) are listed / estimated (a row band's owned lines: the states of its halo lines belong to
// the neighbours).
// Speculative use (round 6; w = 1 only): `d_skip` points at a device word that the work enqueued ahead of this call leaves at zero when this call is
// wanted -- the all-reduced count of undecided pixels of the marking batch that precedes it in the stream -- and `h_skip` at the host copy of that word
// (copied on the same stream before the call).  The list and fallback kernels do nothing when the word is not zero, the estimate kernels then find empty
// lists, and *skipped says so once the host has seen the word: the caller goes on marking and calls again.  The host waits for ONE event in here (list
// lengths + that word), with the first chunk of estimate kernels already enqueued behind it.
int bayes(bcd_hip_ctx *ctx, Work &wk, const float *d_colors, const float *d_pixcov, const uint32_t *d_mask, const int32_t *d_nsim,
          const uint8_t *d_state, int W, int H, int w, int b, float min_eig, float *d_sum, int32_t *d_count, bool defer_redo = false,
          int row_begin = 0, int row_end = INT_MAX, const long long *d_skip = nullptr, const long long *h_skip = nullptr, bool *skipped = nullptr)
{
    const int64_t npix = (int64_t)W * H;
    row_begin = std::max(0, row_begin); row_end = std::min(H, row_end);
    if (skipped) *skipped = false;
    if (d_skip && (w != 1 || !h_skip || !skipped)) return bad(ctx, "speculative estimate: patch radius 1 and a host copy of the word are required");
    const int K = 3 * (2 * w + 1) * (2 * w + 1);
    wk.redo.pending = false;
    RCCHK(ensure(ctx, wk.strong, npix * sizeof(int32_t)));
    RCCHK(ensure(ctx, wk.weak, npix * sizeof(int32_t)));
    RCCHK(ensure(ctx, wk.counters, 64 * sizeof(int32_t)));
    int32_t *d_c = (int32_t *)wk.counters.p + 16; // [0] strong, [1] weak, [2..3] sum |S|, [4..6] work counters of the generic estimate kernel, [7] spectral inverses
    const bool weak_tiles = w == 1; // (other patch radii: the list kernel below)
    wk.h_counters[23] = 0;
    // the two paths only meet in the atomic accumulators: the fallback pixels run on a side stream.  The tiled fallback kernel needs no
    // list (it reads states and |S| itself), so it starts at once -- beside the list compaction and the host round trip for the number of
    // full estimates, during which this scale would otherwise leave the chip idle -- and is out of the way when the prepare kernel arrives
    auto fork_weak_tiles = [&]() -> int {
        HIPCHK(ctx, hipEventRecord(wk.ev_fork, wk.stream));
        HIPCHK(ctx, hipStreamWaitEvent(wk.aux, wk.ev_fork, 0));
        HIPCHK(ctx, bcd_launch_bayes_weak_tiles(d_colors, d_mask, d_state, d_nsim, K + 1, W, H, b, d_sum, d_count, wk.aux, row_begin, row_end, d_skip));
        HIPCHK(ctx, hipEventRecord(wk.ev_join, wk.aux));
        return BCD_HIP_OK;
    };
    if (wk.clean_dc) wk.clean_dc = false; // (k_scale_begin)
    else HIPCHK(ctx, hipMemsetAsync(d_c, 0, 8 * sizeof(int32_t), wk.stream));
    HIPCHK(ctx, bcd_launch_active_lists(d_state, d_nsim, (int64_t)row_begin * W, (int64_t)row_end * W, K + 1, (int32_t *)wk.strong.p, (int32_t *)wk.weak.p, d_c, wk.stream, d_skip));
    HIPCHK(ctx, hipMemcpyAsync(wk.h_counters + 16, d_c, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, wk.stream));
    // (round 5) the list compaction is 253 workgroups of 1024 threads: 18 us alone, 150 us when the fallback kernel's 8 160 tiles were launched
    // first and every CU had to drain before one of them fitted.  The fallback kernel starts BEHIND it (it still overlaps the host round trip).
    if (weak_tiles) RCCHK(fork_weak_tiles());
    const int64_t cap = std::max<int64_t>(1, npix);
    // The estimate kernels are persistent (a wavefront per CU slot, items from a counter), so whatever they occupy stays
    // occupied until they end.  In a multiscale call the finest scale is the critical path and the coarse scales have slack: they
    // take a quarter of the slots, which leaves LDS and wave slots on every CU to the finest scale's short kernels (masks,
    // marking, lists) running beside them -- measured 1080p: 306 -> 315 Mpix/s (100 % -> 25 %; 12 %: 320, 6 %: 250).  The share is
    // ctx->coarse_share: 25 on a new geometry, then steered by bcd_hip_denoise.
    int cus = std::max(1, ctx->num_cus * ctx->cu_share_pct / 100);
    if (&wk != &ctx->main) cus = std::max(1, cus * ctx->coarse_share / 100);
    const int weak_blocks = (int)std::min<int64_t>(cap, (int64_t)cus * 32);
    if (!weak_tiles) { // the list kernel (other patch radii): many cheap items beside the full estimate's few long ones
        HIPCHK(ctx, hipEventRecord(wk.ev_fork, wk.stream));
        HIPCHK(ctx, hipStreamWaitEvent(wk.aux, wk.ev_fork, 0));
        HIPCHK(ctx, bcd_launch_bayes_weak(d_colors, d_mask, (const int32_t *)wk.weak.p, d_c + 1, weak_blocks, W, H, w, b, d_sum, d_count, wk.aux));
        HIPCHK(ctx, hipEventRecord(wk.ev_join, wk.aux));
    }
    if (w == 1) {
        const size_t rec = bcd_bayes27_record_bytes();
        // (round 6: 2^18 instead of 2^17 -- half as many drains of the three persistent kernels on a -m 0 frame: 60.9 -> 60.2-60.5 ms at 1080p; sizes aligned to
        // the eigensolver's 6 144 matrices per round of the grid made no difference)
        const int chunk_max = 1 << 18; // 262144 pixels = 2.6 GB of records
        RCCHK(ensure(ctx, wk.work_q, BCD_WORK_INTS * sizeof(int32_t)));
        auto launch_chunk = [&](int first, int n, bool defer, const int *d_n) -> int {
            if (wk.clean_wq) wk.clean_wq = false; // (k_scale_begin)
            else HIPCHK(ctx, hipMemsetAsync(wk.work_q.p, 0, BCD_WORK_INTS * sizeof(int32_t), wk.stream)); // the work queues of the three kernels
            HIPCHK(ctx, bcd_launch_bayes27(d_colors, d_pixcov, d_mask, (const int32_t *)wk.strong.p, first, n, (int *)wk.work_q.p, cus, W, H, b, min_eig,
                                           (float *)wk.gscratch.p, d_sum, d_count, d_c + 7, wk.stream, defer ? 1 : 0, d_n));
            return BCD_HIP_OK;
        };
        // Round 6: the first chunk does not wait for the host.  The list's length is on its way back (the copy above), the previous frame of this
        // geometry on this workspace said how many items to expect: the three kernels are launched for that many + 1/8 with the length read on the
        // DEVICE (records sized for the capacity; wavefronts without an item leave at once), the host waits for the COPY only -- while the prepare
        // kernel is already running -- and launches further chunks for whatever the capacity did not cover.  A first frame, a workspace whose last
        // frame had (next to) no full estimates, and lists beyond one chunk take the synchronous path below.
        int ahead = 0;
        HIPCHK(ctx, hipEventRecord(wk.ev_counts, wk.stream));
        if (wk.strong_hint_W == W && wk.strong_hint_H == H && wk.strong_hint >= 512 && wk.strong_hint + wk.strong_hint / 8 + 1024 <= chunk_max) {
            ahead = wk.strong_hint + wk.strong_hint / 8 + 1024;
            RCCHK(ensure(ctx, wk.gscratch, rec * (size_t)ahead));
            RCCHK(launch_chunk(0, ahead, defer_redo, d_c));
        }
        HIPCHK(ctx, hipEventSynchronize(wk.ev_counts));
        if (h_skip && *h_skip != 0) { // the speculation failed: nothing was listed, the kernels enqueued above found nothing to do
            *skipped = true;
            HIPCHK(ctx, hipStreamWaitEvent(wk.stream, wk.ev_join, 0));
            return BCD_HIP_OK;
        }
        const int n_strong = wk.h_counters[16];
        wk.strong_hint = n_strong; wk.strong_hint_W = W; wk.strong_hint_H = H;
        if (ahead > 0 && n_strong <= ahead) {
            if (defer_redo) { wk.redo.pending = true; wk.redo.first = 0; wk.redo.n = ahead; wk.redo.cus = cus; } // (the counter is h_counters[23], below)
        } else {
            if (ahead > 0 && defer_redo) // the first chunk's records are about to be reused: its redo list (normally empty) is walked now
                HIPCHK(ctx, bcd_launch_bayes27_redo(d_colors, d_pixcov, d_mask, (const int32_t *)wk.strong.p, 0, ahead, (int *)wk.work_q.p, cus, W, H, b, min_eig,
                                                    (float *)wk.gscratch.p, d_sum, d_count, wk.stream));
            const int remaining = n_strong - ahead;
            if (remaining > 0 && ahead == 0) RCCHK(ensure(ctx, wk.gscratch, rec * (size_t)std::min(remaining, chunk_max)));
            const int chunk = ahead > 0 ? ahead : chunk_max; // (the records of a first chunk launched ahead are laid out for `ahead` items)
            for (int first = ahead; first < n_strong; first += chunk) {
                const int n = std::min(chunk, n_strong - first);
                // one chunk (the usual case) and a caller that looks at the redo counter after its last synchronisation: the redo kernel waits for that
                const bool defer = defer_redo && ahead == 0 && n_strong <= chunk_max;
                RCCHK(launch_chunk(first, n, defer, nullptr));
                if (defer) { wk.redo.pending = true; wk.redo.first = first; wk.redo.n = n; wk.redo.cus = cus; }
            }
            if (ahead > 0 && remaining > 0) {
                // the guess was too small (a frame unlike its predecessor: -m 0 after -m 1): this frame has gone through in chunks of the guessed size;
                // the records grow to what the next frame of its kind needs NOW, in the frame that met the change (the growth waits for the chunks)
                HIPCHK(ctx, hipStreamSynchronize(wk.stream));
                RCCHK(ensure(ctx, wk.gscratch, rec * (size_t)std::min(n_strong, chunk_max)));
            }
        }
        HIPCHK(ctx, hipMemcpyAsync(wk.h_counters + 23, d_c + 7, sizeof(int32_t), hipMemcpyDeviceToHost, wk.stream)); // read after the scale's last synchronisation
    } else {
        const size_t per_block = bcd_bayes_scratch_bytes_per_block(w, b);
        const int strong_blocks = (int)std::min<int64_t>(cap, 1024); // generic kernel: 1024 scratch slices
        if (per_block) RCCHK(ensure(ctx, wk.gscratch, per_block * (size_t)strong_blocks));
        HIPCHK(ctx, bcd_launch_bayes_strong(d_colors, d_pixcov, d_mask, (const int32_t *)wk.strong.p, d_c, d_c + 4, strong_blocks, W, H, w, b, min_eig,
                                            d_sum, d_count, (float *)wk.gscratch.p, wk.gscratch.bytes, wk.stream));
    }
    HIPCHK(ctx, hipStreamWaitEvent(wk.stream, wk.ev_join, 0));
    return BCD_HIP_OK;
}

void bayes_counts(const Work &wk, int64_t *n_strong, int64_t *n_weak, int64_t *sim_total)
{
    *n_strong = wk.h_counters[16];
    *n_weak = wk.h_counters[17];
    memcpy(sim_total, wk.h_counters + 18, sizeof(*sim_total));
}

float stage_ms(Work &wk, int a, int b)
{
    float ms = 0.f;
    hipEventElapsedTime(&ms, wk.ev_stage[a], wk.ev_stage[b]);
    return ms;
}

// one scale: accumulators only (d_sum / d_count are zeroed here)
int mono_accumulate(bcd_hip_ctx *ctx, Work &wk, const float *d_colors, const float *d_ns, const float *d_hist, const float *d_cov,
                    int W, int H, int D, int row_begin, int row_end, const bcd_hip_params *prm, uint32_t seed, int scale,
                    float *d_sum, int32_t *d_count, float *d_out = nullptr /* finalised image, optional */)
{
    const int w = prm->patch_radius, b = prm->search_radius;
    const size_t npix = (size_t)W * H;
    const int side = 2 * b + 1, words = (side * side + 31) / 32;
    RCCHK(ensure(ctx, wk.pixcov, npix * 6 * sizeof(float)));
    RCCHK(ensure(ctx, wk.mask, npix * words * sizeof(uint32_t)));
    RCCHK(ensure(ctx, wk.nsim, npix * sizeof(int32_t)));
    RCCHK(ensure(ctx, wk.state, npix));
    bcd_hip_scale_stats &st = ctx->stats[scale < MAX_SCALES ? scale : MAX_SCALES - 1];
    memset(&st, 0, sizeof(st));
    st.width = W; st.height = H;
    st.main_pixels = (int64_t)std::max(0, W - 2 * w) * std::max(0, std::min(row_end, H - w) - std::max(row_begin, w));
    const bool prof = ctx->profiling;
    // Host round trips of a scale: one per batch of marking launches (normally a single batch; it also brings the range flag of
    // the fast division back), and one at the end for the counters.  The estimate kernels take their list lengths from device
    // memory, the finalisation is enqueued before the last synchronisation.
    if (prof) HIPCHK(ctx, hipEventRecord(wk.ev_stage[0], wk.stream));
    // The per-pixel covariances (only the estimate stage reads them) and the clearing of the accumulators go to the side stream: the scale's
    // own stream starts with the distance kernel, they run beside it instead of ahead of it / between marking and estimate
    HIPCHK(ctx, hipEventRecord(wk.ev_fork, wk.stream)); // (the inputs are ready at this point of the scale's stream)
    HIPCHK(ctx, hipStreamWaitEvent(wk.aux, wk.ev_fork, 0));
    HIPCHK(ctx, bcd_launch_pixel_cov_clear(d_cov, d_ns, (int64_t)npix, (float *)wk.pixcov.p, d_sum, d_count, wk.aux)); // (+ the accumulators cleared: one launch)
    HIPCHK(ctx, hipEventRecord(wk.ev_pixcov, wk.aux));
    // every counter, flag and work queue of the chain in one launch at the head of the scale's stream (round 4: they were ~7 fills between
    // the kernels of the critical path); flags raised by distance planes computed ahead of this call are kept
    {
        constexpr size_t LINE_INTS = (size_t)BCD_CNT_LINES * BCD_CNT_STRIDE;
        RCCHK(ensure(ctx, wk.counters, 64 * sizeof(int32_t)));
        RCCHK(ensure(ctx, wk.cnt_lines, ROUND_BATCH * LINE_INTS * sizeof(int)));
        RCCHK(ensure(ctx, wk.work_q, BCD_WORK_INTS * sizeof(int32_t)));
        const bool ahead = wk.planes.ready;
        HIPCHK(ctx, bcd_launch_scale_begin((int *)wk.counters.p, 64, ahead ? 40 : -1, ahead ? 42 : -1, (int *)wk.cnt_lines.p, (int)(ROUND_BATCH * LINE_INTS),
                                           (int *)wk.work_q.p, BCD_WORK_INTS, wk.stream));
        wk.clean_flags = wk.clean_lines = wk.clean_dc = wk.clean_wq = true;
    }
    const bool marking = prm->marked_skip_probability > 0.f;
    // Round 6: with marking and 3 x 3 patches the estimate is enqueued BEHIND every marking batch, valid only if that batch decided the last pixel and the
    // masks passed their checks (one device word says so: k_sum_counter_lines); the host waits once per batch -- for that word and the list lengths together,
    // inside bayes() -- where it used to wait for the batch, then for the lists.  A batch that leaves pixels undecided (rare: the batch is sized from the
    // previous frame) costs the empty launches of one skipped estimate.
    const bool speculate = marking && w == 1;
    const long long REDO = 1ll << 40;
    bool estimated = false;
    for (int attempt = 0, mode = 2; attempt < 4; ++attempt) { // production kernels; if they complain: general sample counts (RATIO form, then the reference's operations), then exact kernels
        RCCHK(similarity(ctx, wk, d_hist, d_ns, W, H, D, w, b, prm->hist_dist_threshold, (uint32_t *)wk.mask.p, (int32_t *)wk.nsim.p, mode));
        if (prof && attempt == 0) HIPCHK(ctx, hipEventRecord(wk.ev_stage[1], wk.stream));
        if (!speculate) {
            RCCHK(active_set(ctx, wk, (const uint32_t *)wk.mask.p, (const int32_t *)wk.nsim.p, W, H, w, b, row_begin, row_end,
                             prm->marked_skip_probability, prm->use_random_pixel_order, seed, (uint8_t *)wk.state.p, &st.active_rounds));
            if (mode == 1) break;
            if (!marking) HIPCHK(ctx, hipStreamSynchronize(wk.stream)); // no marking batch brought the flag back
        } else {
            HIPCHK(ctx, bcd_launch_active_init((const int32_t *)wk.nsim.p, W, H, w, row_begin, row_end, prm->marked_skip_probability, seed, 0, (uint8_t *)wk.state.p, wk.stream));
            wk.dep_ready = false;
            if (attempt == 0) HIPCHK(ctx, hipStreamWaitEvent(wk.stream, wk.ev_pixcov, 0)); // covariances computed, accumulators cleared (long done)
            long long *d_total = reinterpret_cast<long long *>((int *)wk.counters.p + 48), *h_total = reinterpret_cast<long long *>(wk.h_counters + 48);
            const uint32_t key_seed = prm->use_random_pixel_order == 2 ? bcd_strip_order_seed(W, H, w, b) : seed; // (as active_set)
            int rounds = 0;
            long long before = -1;
            for (;;) {
                RCCHK(active_step_enqueue(ctx, wk, (const uint32_t *)wk.mask.p, (const int32_t *)wk.nsim.p, W, H, w, b, row_begin, row_end, prm->use_random_pixel_order,
                                          key_seed, 0, (uint8_t *)wk.state.p, d_total, mode != 1));
                HIPCHK(ctx, hipMemcpyAsync(h_total, d_total, sizeof(long long), hipMemcpyDeviceToHost, wk.stream));
                bool skipped = false;
                wk.clean_flags = wk.clean_lines = false; // (consumed by the similarity pass and the batch)
                RCCHK(bayes(ctx, wk, d_colors, (const float *)wk.pixcov.p, (const uint32_t *)wk.mask.p, (const int32_t *)wk.nsim.p, (const uint8_t *)wk.state.p, W, H, w, b,
                            prm->min_eigen_value, d_sum, d_count, true, 0, H, d_total, h_total, &skipped));
                int undecided = 0, n = 0;
                active_step_collect(wk, &undecided, &n); // (bayes() waited for an event behind the batch's counters)
                rounds += n;
                if (!skipped) { estimated = true; (void)similarity_redo_mode(wk); break; } // (0 by construction; keeps the workspace's memory of uniform sample counts)
                if (*h_total >= REDO) break; // the masks did not pass: again with the next kind of kernels
                if (before >= 0 && *h_total >= before) { set_err(ctx, "marking fixed point made no progress"); return BCD_HIP_EDEVICE; }
                before = *h_total;
            }
            wk.rounds_hint = rounds;
            st.active_rounds = rounds;
            if (estimated) break;
        }
        const int redo = similarity_redo_mode(wk);
        if (redo == 0) break; // inputs inside the guarded range, uniform-count guess right, borderline list not overflowed
        mode = (redo == 3 && (mode == 2 || wk.ratio_used)) ? 3 : 1; // (wk.ratio_used: the RATIO form declined and the workspace has noted it -- the reference's operations are next)
    }
    progress_add(ctx, 0.5 * (double)npix); // similar patches selected, processed set known
    if (prof) HIPCHK(ctx, hipEventRecord(wk.ev_stage[2], wk.stream));
    if (!estimated) {
        HIPCHK(ctx, hipStreamWaitEvent(wk.stream, wk.ev_pixcov, 0)); // covariances computed, accumulators cleared (long done)
        wk.clean_flags = wk.clean_lines = false; // (consumed, or never used by this configuration)
        RCCHK(bayes(ctx, wk, d_colors, (const float *)wk.pixcov.p, (const uint32_t *)wk.mask.p, (const int32_t *)wk.nsim.p,
                    (const uint8_t *)wk.state.p, W, H, w, b, prm->min_eigen_value, d_sum, d_count, true));
    }
    wk.clean_dc = wk.clean_wq = false;
    if (prof) HIPCHK(ctx, hipEventRecord(wk.ev_stage[3], wk.stream));
    if (d_out) HIPCHK(ctx, bcd_launch_finalize(d_sum, d_count, (int64_t)npix, d_out, wk.stream));
    HIPCHK(ctx, hipStreamSynchronize(wk.stream));
    if (wk.redo.pending && wk.h_counters[23] > 0) {
        // items whose sweep inverse failed its checks (large -e, ill-conditioned frames): their list goes through the LDS kernel now
        // (spectral inverse), and the finalisation is repeated on the completed accumulators
        HIPCHK(ctx, bcd_launch_bayes27_redo(d_colors, (const float *)wk.pixcov.p, (const uint32_t *)wk.mask.p, (const int32_t *)wk.strong.p, wk.redo.first, wk.redo.n,
                                            (int *)wk.work_q.p, wk.redo.cus, W, H, b, prm->min_eigen_value, (float *)wk.gscratch.p, d_sum, d_count, wk.stream));
        if (d_out) HIPCHK(ctx, bcd_launch_finalize(d_sum, d_count, (int64_t)npix, d_out, wk.stream));
        HIPCHK(ctx, hipStreamSynchronize(wk.stream));
    }
    wk.redo.pending = false;
    progress_add(ctx, 0.5 * (double)npix);
    int64_t ns = 0, nw = 0, tot = 0;
    bayes_counts(wk, &ns, &nw, &tot);
    st.processed = ns + nw; st.fallback = nw; st.similar_total = tot;
    st.similarity_path = wk.border_capacity > 0 ? (wk.ratio_used ? 2 : 1) : 0;
    st.borderline_pairs = wk.border_capacity > 0 ? wk.h_counters[43] : 0;
    st.cu_share = ctx->cu_share_pct * (&wk != &ctx->main ? ctx->coarse_share : 100) / 100;
    st.spectral_inverses = wk.h_counters[23];
    if (prof) {
        st.ms_similarity = stage_ms(wk, 0, 1);
        st.ms_active = stage_ms(wk, 1, 2);
        st.ms_bayes = stage_ms(wk, 2, 3);
        st.ms_total = stage_ms(wk, 0, 3);
    }
    return BCD_HIP_OK;
}

// mergeOutputs (MultiscaleDenoiser.cpp:453-466) on a workspace's stream: hi -= up(down(hi)); hi += up(lo)
int merge_on(bcd_hip_ctx *ctx, Work &wk, float *d_hi, int W, int H, const float *d_lo, int D)
{
    const int w2 = W / 2, h2 = H / 2;
    RCCHK(ensure(ctx, wk.tmp_lo, (size_t)w2 * h2 * D * sizeof(float)));
    HIPCHK(ctx, bcd_launch_downscale(1, d_hi, W, H, D, (float *)wk.tmp_lo.p, wk.stream));
    HIPCHK(ctx, bcd_launch_merge_interpolate((const float *)wk.tmp_lo.p, d_lo, w2, h2, D, d_hi, W, H, wk.stream)); // (round 6: the two interpolations in one pass)
    return BCD_HIP_OK;
}

// one pyramid level from the finer one (MultiscaleDenoiser.cpp:41-53) on `st`
int build_level(bcd_hip_ctx *ctx, const float *col, const float *ns, const float *hs, const float *cv, int W, int H, int D,
                DevBuf (&lvl)[5], hipStream_t st)
{
    HIPCHK(ctx, bcd_launch_downscale(1, col, W, H, 3, (float *)lvl[0].p, st));
    HIPCHK(ctx, bcd_launch_downscale(0, ns, W, H, 1, (float *)lvl[1].p, st));
    HIPCHK(ctx, bcd_launch_downscale(0, hs, W, H, D, (float *)lvl[2].p, st));
    HIPCHK(ctx, bcd_launch_downscale_cov(cv, ns, W, H, (float *)lvl[3].p, st));
    return BCD_HIP_OK;
}

int mono(bcd_hip_ctx *ctx, Work &wk, const float *d_colors, const float *d_ns, const float *d_hist, const float *d_cov, int W, int H, int D,
         const bcd_hip_params *prm, uint32_t seed, int scale, float *d_out)
{
    const size_t npix = (size_t)W * H;
    RCCHK(ensure(ctx, wk.sum, npix * 3 * sizeof(float)));
    RCCHK(ensure(ctx, wk.cnt, npix * sizeof(int32_t)));
    return mono_accumulate(ctx, wk, d_colors, d_ns, d_hist, d_cov, W, H, D, 0, H, prm, seed, scale, (float *)wk.sum.p, (int32_t *)wk.cnt.p, d_out);
}


int work_init(bcd_hip_ctx *ctx, Work &w, hipStream_t stream)
{
    if (w.initialised) return BCD_HIP_OK;
    if (w.h_counters || w.aux || w.ev_done) return bad(ctx, "workspace left half-initialised by an earlier failure"); // never run on null handles
    if (stream) w.stream = stream;
    else {
        HIPCHK(ctx, hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
        w.owns_stream = true;
    }
    HIPCHK(ctx, hipHostMalloc((void **)&w.h_counters, 64 * sizeof(int32_t), hipHostMallocDefault));
    for (int i = 0; i < 4; ++i) HIPCHK(ctx, hipEventCreate(&w.ev_stage[i]));
    HIPCHK(ctx, hipEventCreateWithFlags(&w.ev_done, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&w.ev_built, hipEventDisableTiming));
    HIPCHK(ctx, hipStreamCreateWithFlags(&w.aux, hipStreamNonBlocking));
    HIPCHK(ctx, hipEventCreateWithFlags(&w.ev_fork, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&w.ev_join, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&w.ev_pixcov, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&w.ev_counts, hipEventDisableTiming));
    w.initialised = true;
    return BCD_HIP_OK;
}

void work_destroy(Work &w)
{
    DevBuf *bufs[] = { &w.T, &w.Cn, &w.mask, &w.fwd, &w.nsim, &w.state, &w.strong, &w.weak, &w.counters, &w.cnt_lines, &w.work_q, &w.pixcov, &w.sum, &w.cnt, &w.gscratch, &w.dep, &w.tmp_lo, &w.border, &w.ratio_stats };
    for (DevBuf *b : bufs) if (b->p) (void)hipFree(b->p);
    if (w.h_counters) (void)hipHostFree(w.h_counters);
    for (auto &pr : w.ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (int i = 0; i < 4; ++i) if (w.ev_stage[i]) (void)hipEventDestroy(w.ev_stage[i]);
    if (w.ev_done) (void)hipEventDestroy(w.ev_done);
    if (w.ev_built) (void)hipEventDestroy(w.ev_built);
    if (w.ev_fork) (void)hipEventDestroy(w.ev_fork);
    if (w.ev_join) (void)hipEventDestroy(w.ev_join);
    if (w.ev_pixcov) (void)hipEventDestroy(w.ev_pixcov);
    if (w.ev_counts) (void)hipEventDestroy(w.ev_counts);
    if (w.aux) (void)hipStreamDestroy(w.aux);
    if (w.owns_stream && w.stream) (void)hipStreamDestroy(w.stream);
}

} // namespace

// =====================================================================================================
extern "C" {

int bcd_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void bcd_hip_default_params(bcd_hip_params *p)
{
    if (!p) return;
    p->hist_dist_threshold = 1.f;   // IDenoiser.h:23-31
    p->patch_radius = 1;
    p->search_radius = 6;
    p->min_eigen_value = 1.e-8f;
    p->use_random_pixel_order = 1;
    p->marked_skip_probability = 1.f;
    p->order_seed = 1234u;
}

int bcd_hip_ctx_create(bcd_hip_ctx **out, int device, void *hip_stream)
{
    if (!out) return BCD_HIP_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0 || device < 0 || device >= n) return BCD_HIP_EDEVICE;
    bcd_hip_ctx *ctx = new (std::nothrow) bcd_hip_ctx();
    if (!ctx) return BCD_HIP_ENOMEM;
    ctx->device = device;
    DeviceGuard guard(ctx);
    if (!guard.ok) { delete ctx; return BCD_HIP_EDEVICE; }
    memset(ctx->stats, 0, sizeof(ctx->stats));
    if (hip_stream) ctx->stream = (hipStream_t)hip_stream;
    else {
        if (hipStreamCreate(&ctx->stream) != hipSuccess) { delete ctx; return BCD_HIP_EDEVICE; }
        ctx->owns_stream = true;
    }
    if (work_init(ctx, ctx->main, ctx->stream) != BCD_HIP_OK || hipEventCreateWithFlags(&ctx->ev_pyramid, hipEventDisableTiming) != hipSuccess) {
        bcd_hip_ctx_destroy(ctx);
        return BCD_HIP_EDEVICE;
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && cus > 0) ctx->num_cus = cus;
    }
    const char *env = getenv("BCD_HIP_SERIAL_SCALES");
    ctx->concurrent_scales = !(env && env[0] == '1');
    env = getenv("BCD_HIP_EXACT_SIMILARITY");
    ctx->fast_similarity = !(env && env[0] == '1');
    env = getenv("BCD_HIP_STREAM_UPLOADS");
    ctx->stream_uploads = !(env && env[0] == '0');
    env = getenv("BCD_HIP_SPARSE_UPLOAD");
    ctx->sparse_uploads = !(env && env[0] == '0');
    *out = ctx;
    return BCD_HIP_OK;
}

void bcd_hip_ctx_destroy(bcd_hip_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->async.th.joinable()) { // (a frame still in flight is finished first)
        { std::lock_guard<std::mutex> lk(ctx->async.mu); ctx->async.quit = true; }
        ctx->async.cv.notify_all();
        ctx->async.th.join();
    }
    DeviceGuard guard(ctx);
    (void)hipDeviceSynchronize();
    work_destroy(ctx->main);
    for (int s = 0; s < MAX_SCALES; ++s) work_destroy(ctx->extra[s]);
    if (ctx->tmp_lo.p) (void)hipFree(ctx->tmp_lo.p);
    for (DevBuf &hb : ctx->host_stage) if (hb.p) (void)hipFree(hb.p);
    for (int s = 0; s < MAX_SCALES; ++s)
        for (int k = 0; k < 5; ++k) if (ctx->pyr[s][k].p) (void)hipFree(ctx->pyr[s][k].p);
    if (ctx->ev_pyramid) (void)hipEventDestroy(ctx->ev_pyramid);
    for (hipEvent_t ev : ctx->ev_upload) (void)hipEventDestroy(ev);
    if (ctx->upload_stream) (void)hipStreamDestroy(ctx->upload_stream);
    if (ctx->upload_stream2) (void)hipStreamDestroy(ctx->upload_stream2);
    if (ctx->ev_upload2) (void)hipEventDestroy(ctx->ev_upload2);
    if (ctx->sparse) bcd_sparse_destroy(ctx->sparse);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *bcd_hip_last_error(const bcd_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int bcd_hip_set_profiling(bcd_hip_ctx *ctx, int enabled)
{
    if (!ctx) return BCD_HIP_EINVAL;
    ctx->profiling = enabled != 0;
    return BCD_HIP_OK;
}

int bcd_hip_set_concurrent_scales(bcd_hip_ctx *ctx, int enabled)
{
    if (!ctx) return BCD_HIP_EINVAL;
    ctx->concurrent_scales = enabled != 0;
    return BCD_HIP_OK;
}

int bcd_hip_set_strict_eigensolver(int enabled)
{
    bcd_bayes27_set_strict_eigensolver(enabled);
    return BCD_HIP_OK;
}

int bcd_hip_set_fast_similarity(bcd_hip_ctx *ctx, int enabled)
{
    if (!ctx) return BCD_HIP_EINVAL;
    ctx->fast_similarity = enabled != 0;
    return BCD_HIP_OK;
}

int bcd_hip_set_cu_share(bcd_hip_ctx *ctx, int percent)
{
    if (!ctx || percent < 1 || percent > 100) return BCD_HIP_EINVAL;
    ctx->cu_share_pct = percent;
    return BCD_HIP_OK;
}

int bcd_hip_get_stats(const bcd_hip_ctx *ctx, int scale, bcd_hip_scale_stats *out)
{
    if (!ctx || !out || scale < 0 || scale >= MAX_SCALES) return BCD_HIP_EINVAL;
    *out = ctx->stats[scale];
    return BCD_HIP_OK;
}

int bcd_hip_kernel_time(const bcd_hip_ctx *cctx, float *ms_pairdist, int32_t *launches)
{
    bcd_hip_ctx *ctx = const_cast<bcd_hip_ctx *>(cctx);
    if (!ctx) return BCD_HIP_EINVAL;
    DEVICE_GUARD(ctx);
    HIPCHK(ctx, hipDeviceSynchronize());
    float tot = 0.f;
    int count = 0;
    Work *works[MAX_SCALES + 1];
    works[0] = &ctx->main;
    for (int s = 0; s < MAX_SCALES; ++s) works[s + 1] = &ctx->extra[s];
    for (Work *w : works)
        for (int i = 0; i < w->ev_used; ++i) {
            float ms = 0.f;
            HIPCHK(ctx, hipEventElapsedTime(&ms, w->ev_pool[i].first, w->ev_pool[i].second));
            tot += ms;
            ++count;
        }
    if (ms_pairdist) *ms_pairdist = tot;
    if (launches) *launches = count;
    return BCD_HIP_OK;
}

int bcd_hip_reset_kernel_time(bcd_hip_ctx *ctx)
{
    if (!ctx) return BCD_HIP_EINVAL;
    DEVICE_GUARD(ctx);
    HIPCHK(ctx, hipDeviceSynchronize());
    ctx->main.ev_used = 0;
    for (int s = 0; s < MAX_SCALES; ++s) ctx->extra[s].ev_used = 0;
    return BCD_HIP_OK;
}

int bcd_hip_denoise(bcd_hip_ctx *ctx, const float *d_colors, const float *d_ns, const float *d_hist, const float *d_cov,
                    int W, int H, int D, int nb_scales, const bcd_hip_params *prm, float *d_out)
{
    if (!ctx) return BCD_HIP_EINVAL;
    if (!d_colors || !d_ns || !d_hist || !d_cov || !d_out) return bad(ctx, "null image pointer"); // Denoiser.cpp:266-293
    RCCHK(check_params(ctx, W, H, D, prm));
    if (nb_scales < 1 || nb_scales > MAX_SCALES) return bad(ctx, "bad number of scales");
    DEVICE_GUARD(ctx);
    {
        std::lock_guard<std::mutex> lock(ctx->progress_mutex);
        ctx->progress_done = 0.0;
        ctx->progress_total = 0.0;
        for (int s = 0; s < nb_scales; ++s) ctx->progress_total += (double)(W >> s) * (double)(H >> s);
    }
    if (nb_scales == 1) return mono(ctx, ctx->main, d_colors, d_ns, d_hist, d_cov, W, H, D, prm, bcd_hip_scale_seed(prm->order_seed, 0), 0, d_out);

    // ---- pyramids (MultiscaleDenoiser.cpp:41-53): level s has dims of level s-1 // 2
    const float *col[MAX_SCALES], *ns[MAX_SCALES], *hs[MAX_SCALES], *cv[MAX_SCALES];
    float *out[MAX_SCALES];
    int ws[MAX_SCALES], hh[MAX_SCALES];
    col[0] = d_colors; ns[0] = d_ns; hs[0] = d_hist; cv[0] = d_cov; out[0] = d_out; ws[0] = W; hh[0] = H;
    for (int s = 1; s < nb_scales; ++s) {
        ws[s] = ws[s - 1] / 2; hh[s] = hh[s - 1] / 2;
        if (ws[s] < 2 * prm->patch_radius + 1 || hh[s] < 2 * prm->patch_radius + 1) return bad(ctx, "too many scales for this image size");
        size_t np = (size_t)ws[s] * hh[s];
        RCCHK(ensure(ctx, ctx->pyr[s][0], np * 3 * sizeof(float)));
        RCCHK(ensure(ctx, ctx->pyr[s][1], np * sizeof(float)));
        RCCHK(ensure(ctx, ctx->pyr[s][2], np * D * sizeof(float)));
        RCCHK(ensure(ctx, ctx->pyr[s][3], np * 6 * sizeof(float)));
        RCCHK(ensure(ctx, ctx->pyr[s][4], np * 3 * sizeof(float)));
        col[s] = (float *)ctx->pyr[s][0].p; ns[s] = (float *)ctx->pyr[s][1].p; hs[s] = (float *)ctx->pyr[s][2].p;
        cv[s] = (float *)ctx->pyr[s][3].p; out[s] = (float *)ctx->pyr[s][4].p;
    }
    // ---- the scales are independent until the merges (MultiscaleDenoiser.cpp:79-134 runs them coarse to fine, but each
    // Denoiser only reads its own pyramid level): one stream + host thread + workspace per scale.  Scale s > 0 builds its
    // own pyramid level on its stream (from level s-1, once that is complete), so that the finest scale -- the critical
    // path -- starts at once; after its own chain scale s merges the (already merged) scale s+1 into its output.
    if (ctx->concurrent_scales) {
        // The coarse scales' share of the CU slots (bayes()) follows the previous call on the same geometry: they should be through when
        // the finest scale is at 80 - 92 % of its chain -- earlier means their persistent kernels took more room than they needed next to
        // the finest scale's short kernels, later means they have become the critical path.  Small steps down, larger ones up.
        const int64_t key = ((int64_t)W << 40) ^ ((int64_t)H << 20) ^ ((int64_t)nb_scales << 12) ^ ((int64_t)prm->search_radius << 4) ^ (prm->marked_skip_probability > 0.f);
        if (key != ctx->share_key) { ctx->coarse_share = 25; ctx->share_key = key; }
        const auto t_start = std::chrono::steady_clock::now();
        double t_done[MAX_SCALES] = { 0 };
        HIPCHK(ctx, hipEventRecord(ctx->ev_pyramid, ctx->stream)); // the caller's inputs are ready
        int rcs[MAX_SCALES];
        std::thread threads[MAX_SCALES];
        std::atomic<int> built[MAX_SCALES], done[MAX_SCALES]; // 0 = pending, 1 = event recorded, -1 = failed
        for (int s = 0; s < MAX_SCALES; ++s) { built[s].store(0); done[s].store(0); }
        for (int s = 1; s < nb_scales; ++s) RCCHK(work_init(ctx, ctx->extra[s], nullptr));
        auto await = [](std::atomic<int> &f) { int v; while ((v = f.load()) == 0) std::this_thread::yield(); return v; };
        for (int s = nb_scales - 1; s >= 0; --s) {
            Work *w = s == 0 ? &ctx->main : &ctx->extra[s];
            rcs[s] = BCD_HIP_OK;
            auto job = [&, s, w]() {
                int rc = BCD_HIP_OK;
                bool built_set = s == 0, done_set = s == 0;
                do {
                    if (hipSetDevice(ctx->device) != hipSuccess) { rc = BCD_HIP_EDEVICE; break; }
                    if (s != 0) {
                        if (hipStreamWaitEvent(w->stream, ctx->ev_pyramid, 0) != hipSuccess) { rc = BCD_HIP_EDEVICE; break; }
                        if (s >= 2) {
                            if (await(built[s - 1]) < 0) { rc = BCD_HIP_EDEVICE; break; }
                            if (hipStreamWaitEvent(w->stream, ctx->extra[s - 1].ev_built, 0) != hipSuccess) { rc = BCD_HIP_EDEVICE; break; }
                        }
                        rc = build_level(ctx, col[s - 1], ns[s - 1], hs[s - 1], cv[s - 1], ws[s - 1], hh[s - 1], D, ctx->pyr[s], w->stream);
                        if (rc != BCD_HIP_OK) break;
                        if (hipEventRecord(w->ev_built, w->stream) != hipSuccess) { rc = BCD_HIP_EDEVICE; break; }
                        built[s].store(1); built_set = true;
                    }
                    rc = mono(ctx, *w, col[s], ns[s], hs[s], cv[s], ws[s], hh[s], D, prm, bcd_hip_scale_seed(prm->order_seed, s), s, out[s]);
                    if (rc != BCD_HIP_OK) break;
                    t_done[s] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); // (mono ends with a stream synchronisation)
                    if (s < nb_scales - 1) {
                        if (await(done[s + 1]) < 0) { rc = BCD_HIP_EDEVICE; break; }
                        if (hipStreamWaitEvent(w->stream, ctx->extra[s + 1].ev_done, 0) != hipSuccess) { rc = BCD_HIP_EDEVICE; break; }
                        rc = merge_on(ctx, *w, out[s], ws[s], hh[s], out[s + 1], 3);
                        if (rc != BCD_HIP_OK) break;
                    }
                    if (s != 0) {
                        if (hipEventRecord(w->ev_done, w->stream) != hipSuccess) { rc = BCD_HIP_EDEVICE; break; }
                        done[s].store(1); done_set = true;
                    }
                } while (false);
                if (!built_set) built[s].store(-1); // never leave a waiter spinning
                if (!done_set) done[s].store(-1);
                rcs[s] = rc;
            };
            if (s == 0) job(); else threads[s] = std::thread(job);
        }
        for (int s = 1; s < nb_scales; ++s) threads[s].join();
        for (int s = 0; s < nb_scales; ++s) RCCHK(rcs[s]);
        if (nb_scales > 1 && t_done[0] > 0.0) {
            double last = 0.0;
            for (int s = 1; s < nb_scales; ++s) last = std::max(last, t_done[s]);
            const double frac = last / t_done[0];
            if (frac < 0.80) ctx->coarse_share = std::max(8, ctx->coarse_share - 2);
            else if (frac > 0.92) ctx->coarse_share = std::min(60, ctx->coarse_share + 6);
        }
        return BCD_HIP_OK;
    }
    for (int s = 1; s < nb_scales; ++s)
        RCCHK(build_level(ctx, col[s - 1], ns[s - 1], hs[s - 1], cv[s - 1], ws[s - 1], hh[s - 1], D, ctx->pyr[s], ctx->stream));
    // ---- coarse to fine, one after the other
    for (int s = nb_scales - 1; s >= 0; --s) {
        RCCHK(mono(ctx, ctx->main, col[s], ns[s], hs[s], cv[s], ws[s], hh[s], D, prm, bcd_hip_scale_seed(prm->order_seed, s), s, out[s]));
        if (s < nb_scales - 1) RCCHK(bcd_hip_merge(ctx, out[s], ws[s], hh[s], out[s + 1], 3));
    }
    return BCD_HIP_OK;
}

int bcd_hip_denoise_begin(bcd_hip_ctx *ctx, const float *d_colors, const float *d_ns, const float *d_hist, const float *d_cov,
                          int W, int H, int D, int nb_scales, const bcd_hip_params *prm, float *d_out)
{
    if (!ctx) return BCD_HIP_EINVAL;
    if (!d_colors || !d_ns || !d_hist || !d_cov || !d_out) return bad(ctx, "null image pointer");
    RCCHK(check_params(ctx, W, H, D, prm));
    if (nb_scales < 1 || nb_scales > MAX_SCALES) return bad(ctx, "bad number of scales");
    bcd_hip_ctx::Async &a = ctx->async;
    std::unique_lock<std::mutex> lk(a.mu);
    if (a.in_flight) return bad(ctx, "a frame of this context is already in flight: call bcd_hip_denoise_wait first");
    a.col = d_colors; a.ns = d_ns; a.hist = d_hist; a.cov = d_cov; a.out = d_out; a.W = W; a.H = H; a.D = D; a.S = nb_scales; a.prm = *prm;
    a.has_job = true; a.in_flight = true; a.rc = BCD_HIP_OK;
    if (!a.th.joinable())
        a.th = std::thread([ctx]() {
            bcd_hip_ctx::Async &j = ctx->async;
            std::unique_lock<std::mutex> l(j.mu);
            for (;;) {
                j.cv.wait(l, [&]() { return j.has_job || j.quit; });
                if (!j.has_job) return; // (quit)
                j.has_job = false;
                l.unlock();
                const int rc = bcd_hip_denoise(ctx, j.col, j.ns, j.hist, j.cov, j.W, j.H, j.D, j.S, &j.prm, j.out); // (synchronises the context's streams)
                l.lock();
                j.rc = rc;
                j.in_flight = false;
                j.cv.notify_all();
            }
        });
    lk.unlock();
    a.cv.notify_all();
    return BCD_HIP_OK;
}

int bcd_hip_denoise_wait(bcd_hip_ctx *ctx)
{
    if (!ctx) return BCD_HIP_EINVAL;
    bcd_hip_ctx::Async &a = ctx->async;
    std::unique_lock<std::mutex> lk(a.mu);
    a.cv.wait(lk, [&]() { return !a.in_flight; });
    return a.rc;
}

int bcd_hip_denoise_band(bcd_hip_ctx *ctx, const float *d_colors, const float *d_ns, const float *d_hist, const float *d_cov,
                         int W, int H, int D, int main_row_begin, int main_row_end, const bcd_hip_params *prm, uint32_t order_seed,
                         float *d_sum, int32_t *d_count)
{
    if (!ctx) return BCD_HIP_EINVAL;
    if (!d_colors || !d_ns || !d_hist || !d_cov || !d_sum || !d_count) return bad(ctx, "null image pointer");
    RCCHK(check_params(ctx, W, H, D, prm));
    if (main_row_begin < 0 || main_row_end > H || main_row_begin > main_row_end) return bad(ctx, "bad main row range");
    DEVICE_GUARD(ctx);
    return mono_accumulate(ctx, ctx->main, d_colors, d_ns, d_hist, d_cov, W, H, D, main_row_begin, main_row_end, prm, order_seed, 0, d_sum, d_count);
}

int bcd_hip_denoise_bands(bcd_hip_ctx *ctx, const bcd_hip_band_job *jobs, int njobs, const bcd_hip_params *prm)
{
    if (!ctx) return BCD_HIP_EINVAL;
    if (!jobs || njobs < 1 || njobs > MAX_SCALES) return bad(ctx, "bad job list");
    for (int i = 0; i < njobs; ++i) {
        const bcd_hip_band_job &j = jobs[i];
        if (!j.d_colors || !j.d_nsamples || !j.d_histograms || !j.d_covariances || !j.d_sum || !j.d_count) return bad(ctx, "null image pointer");
        RCCHK(check_params(ctx, j.W, j.H, j.D, prm));
        if (j.main_row_begin < 0 || j.main_row_end > j.H || j.main_row_begin > j.main_row_end) return bad(ctx, "bad main row range");
    }
    DEVICE_GUARD(ctx);
    if (!ctx->concurrent_scales || njobs == 1) {
        for (int i = 0; i < njobs; ++i) {
            const bcd_hip_band_job &j = jobs[i];
            RCCHK(mono_accumulate(ctx, ctx->main, j.d_colors, j.d_nsamples, j.d_histograms, j.d_covariances, j.W, j.H, j.D, j.main_row_begin,
                                  j.main_row_end, prm, j.order_seed, i, j.d_sum, j.d_count));
        }
        return BCD_HIP_OK;
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_pyramid, ctx->stream)); // inputs were produced on the context's stream
    int rcs[MAX_SCALES];
    std::thread threads[MAX_SCALES];
    for (int i = 1; i < njobs; ++i) RCCHK(work_init(ctx, ctx->extra[i], nullptr));
    for (int i = njobs - 1; i >= 0; --i) {
        Work *w = i == 0 ? &ctx->main : &ctx->extra[i];
        const bcd_hip_band_job j = jobs[i];
        rcs[i] = BCD_HIP_OK;
        auto job = [=, &rcs]() {
            if (hipSetDevice(ctx->device) != hipSuccess) { rcs[i] = BCD_HIP_EDEVICE; return; }
            if (i != 0 && hipStreamWaitEvent(w->stream, ctx->ev_pyramid, 0) != hipSuccess) { rcs[i] = BCD_HIP_EDEVICE; return; }
            rcs[i] = mono_accumulate(ctx, *w, j.d_colors, j.d_nsamples, j.d_histograms, j.d_covariances, j.W, j.H, j.D, j.main_row_begin,
                                     j.main_row_end, prm, j.order_seed, i, j.d_sum, j.d_count);
            if (rcs[i] == BCD_HIP_OK && i != 0 && hipEventRecord(w->ev_done, w->stream) != hipSuccess) rcs[i] = BCD_HIP_EDEVICE;
        };
        if (i == 0) job(); else threads[i] = std::thread(job);
    }
    for (int i = 1; i < njobs; ++i) threads[i].join();
    for (int i = 0; i < njobs; ++i) RCCHK(rcs[i]);
    for (int i = 1; i < njobs; ++i) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->extra[i].ev_done, 0));
    return BCD_HIP_OK;
}

int bcd_hip_denoise_host_ex(bcd_hip_ctx *ctx, const float *h_colors, const float *h_ns, const float *h_hist, const float *h_cov,
                            int W, int H, int D, int nb_scales, const bcd_hip_params *prm, const bcd_hip_host_options *opt, float *h_out)
{
    if (!ctx) return BCD_HIP_EINVAL;
    if (!h_colors || !h_ns || !h_hist || !h_cov || !h_out) return bad(ctx, "null image pointer");
    RCCHK(check_params(ctx, W, H, D, prm));
    DEVICE_GUARD(ctx);
    const size_t np = (size_t)W * H;
    const size_t sz[5] = { np * 3, np, np * D, np * 6, np * 3 };
    const float *src[4] = { h_colors, h_ns, h_hist, h_cov };
    const bool prefilter = opt && opt->spike_factor > 0.f;
    if (prefilter && (W < 3 || H < 3)) return bad(ctx, "image smaller than 3x3");
    // device copies live in the context (grow-only): a sequence of frames pays for the allocations once
    float *d[9];
    for (int i = 0; i < 5; ++i) { RCCHK(ensure(ctx, ctx->host_stage[i], sz[i] * sizeof(float))); d[i] = (float *)ctx->host_stage[i].p; }
    for (int i = 0; i < 4; ++i) {
        d[5 + i] = d[i];
        if (prefilter) { RCCHK(ensure(ctx, ctx->host_stage[5 + i], sz[i] * sizeof(float))); d[5 + i] = (float *)ctx->host_stage[5 + i].p; }
    }
    // The frame arrives in row chunks on an upload stream; the lines that have arrived are prefiltered (SpikeRemovalFilter::filter,
    // src/cli/main.cpp:428-441, on the device copies: no second trip over PCIe) and the finest scale's approximate distance planes -- the
    // largest single kernel of the frame, and a function of the histograms alone -- are computed for them while the next chunk travels.
    // Everything else needs the whole frame (pyramid, the marking order) and follows the last chunk.
    const int b = prm->search_radius, tile = bcd_pairdist_rw_tile_lines();
    const bool stream_in = ctx->stream_uploads && fast_similarity_applies(ctx, D, prm->patch_radius, prm->hist_dist_threshold) && H >= 256;
    if (!stream_in) {
        for (int i = 0; i < 4; ++i) HIPCHK(ctx, hipMemcpyAsync(d[i], src[i], sz[i] * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        if (prefilter) HIPCHK(ctx, bcd_launch_spike(d[0], d[1], d[2], d[3], W, H, D, opt->spike_factor, d[5], d[6], d[7], d[8], ctx->stream));
    } else {
        Work &wk = ctx->main;
        if (!ctx->upload_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
        const int nd = bcd_delta_count(b);
        RCCHK(ensure(ctx, wk.T, np * nd * sizeof(float)));
        RCCHK(ensure(ctx, wk.Cn, count_plane_bytes(np, nd))); // (the size similarity() will ask for: a larger request there would REALLOCATE the planes computed here)
        RCCHK(ensure(ctx, wk.counters, 64 * sizeof(int32_t)));
        int *d_flag = (int *)wk.counters.p + 40;
        HIPCHK(ctx, hipMemsetAsync(d_flag, 0, 4 * sizeof(int), ctx->stream));
        // uniform power-of-two sample count: taken from the first pixel; the distance kernel checks every pixel against it and raises the
        // flag that sends the scale to the exact kernels if the guess was wrong (k_pairdist_rw, range_flag bit 1)
        // (a strided sample of 1024 pixels settles the usual non-uniform case -- adaptive sampling -- on the host at no cost)
        float uni_n = 0.f;
        {
            int e = 0;
            const float n0 = h_ns[0];
            if (n0 >= 1.f && n0 <= 65536.f && frexpf(n0, &e) == 0.5f) uni_n = n0;
            const size_t stride = std::max<size_t>(1, np / 1024);
            for (size_t i = 0; i < np && uni_n > 0.f; i += stride)
                if (h_ns[i] != n0) uni_n = 0.f;
        }
        const int chunk = std::max(64, ((H + 7) / 8 + tile - 1) / tile * tile); // ~8 chunks, whole tile rows
        const int tile_rows = (H + tile - 1) / tile;
        int filtered = 0, tiles_done = 0, k = 0;
        // the upload stream must not overwrite device copies an earlier frame's kernels may still read
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        // colours, sample counts and covariances first, whole (83 MB at 1080p; the prefilter and the distance kernel need them with the
        // first histogram lines), then the histograms -- 87 % of the bytes -- in row chunks
        // Without the prefilter only the sample counts are needed with the first histogram lines (the distance kernel); colours and covariances
        // are first read by the pyramid and the estimate stage.  Their (pageable, host-blocking) copies then run on a helper thread and a
        // stream of their own beside the histogram pieces, whose pace is set by the host-side packing and leaves the link half idle (round 4).
        std::thread side_copy;
        hipError_t side_rc = hipSuccess;
        struct SideJoin { std::thread &t; ~SideJoin() { if (t.joinable()) t.join(); } } side_join{ side_copy };
        const bool side = !prefilter && ctx->sparse_uploads && (D & 3) == 0;
        if (side) {
            if (!ctx->upload_stream2) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->upload_stream2, hipStreamNonBlocking));
            if (!ctx->ev_upload2) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_upload2, hipEventDisableTiming));
            HIPCHK(ctx, hipMemcpyAsync(d[1], src[1], sz[1] * sizeof(float), hipMemcpyHostToDevice, ctx->upload_stream));
            const int dev = ctx->device;
            hipStream_t s2 = ctx->upload_stream2;
            hipEvent_t e2 = ctx->ev_upload2;
            float *dc = d[0], *dv = d[3];
            const float *hc = src[0], *hv = src[3];
            const size_t nc = sz[0] * sizeof(float), nv = sz[3] * sizeof(float);
            side_copy = std::thread([=, &side_rc]() {
                hipError_t e = hipSetDevice(dev);
                if (e == hipSuccess) e = hipMemcpyAsync(dc, hc, nc, hipMemcpyHostToDevice, s2);
                if (e == hipSuccess) e = hipMemcpyAsync(dv, hv, nv, hipMemcpyHostToDevice, s2);
                if (e == hipSuccess) e = hipEventRecord(e2, s2);
                side_rc = e;
            });
        } else
            for (int i : { 0, 1, 3 }) HIPCHK(ctx, hipMemcpyAsync(d[i], src[i], sz[i] * sizeof(float), hipMemcpyHostToDevice, ctx->upload_stream));
        const bool sparse = ctx->sparse_uploads && (D & 3) == 0;
        if (sparse) {
            if (!ctx->sparse && !(ctx->sparse = bcd_sparse_create())) { set_err(ctx, "out of host memory"); return BCD_HIP_ENOMEM; }
            bcd_sparse_frame_begin(ctx->sparse);
        }
        ctx->upload_raw_bytes = ctx->upload_sent_bytes = (long long)sz[2] * 4;
        for (int r0 = 0; r0 < H; r0 += chunk, ++k) {
            const int r1 = std::min(H, r0 + chunk);
            {
                const size_t off = (size_t)r0 * W * D, n = (size_t)(r1 - r0) * W * D;
                if (sparse) HIPCHK(ctx, bcd_sparse_upload(ctx->sparse, d[2] + off, h_hist + off, n, ctx->upload_stream)); // (off % 4 == 0: D % 4 == 0 on this path)
                else HIPCHK(ctx, hipMemcpyAsync(d[2] + off, h_hist + off, n * sizeof(float), hipMemcpyHostToDevice, ctx->upload_stream));
            }
            if ((int)ctx->ev_upload.size() <= k) {
                hipEvent_t ev;
                HIPCHK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                ctx->ev_upload.push_back(ev);
            }
            HIPCHK(ctx, hipEventRecord(ctx->ev_upload[k], ctx->upload_stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_upload[k], 0));
            int avail = r1;
            if (prefilter) { // a filtered line reads its own and the two adjacent input lines (clamped inward at the frame border)
                const int upto = r1 == H ? H : std::max(0, r1 - 1);
                HIPCHK(ctx, bcd_launch_spike_rows(d[0], d[1], d[2], d[3], W, H, D, opt->spike_factor, d[5], d[6], d[7], d[8], filtered, upto, ctx->stream));
                filtered = std::max(filtered, upto);
                avail = filtered;
            }
            // a tile row reads its own lines and the b lines below them
            const int t_end = avail == H ? tile_rows : std::max(0, (avail - b) / tile);
            if (t_end > tiles_done) {
                HIPCHK(ctx, bcd_launch_pairdist_rw_rows(d[7], d[6], W, H, D, b, wk.T.p, (uint8_t *)wk.Cn.p, d_flag, uni_n, tiles_done, t_end, ctx->stream));
                tiles_done = t_end;
            }
        }
        if (sparse) bcd_sparse_frame_bytes(ctx->sparse, &ctx->upload_raw_bytes, &ctx->upload_sent_bytes);
        if (side) { // colours and covariances have been enqueued by now (the helper thread is joined), the frame's kernels wait for their arrival
            side_copy.join();
            HIPCHK(ctx, side_rc);
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_upload2, 0));
        }
        wk.planes.ready = true; wk.planes.hist = d[7]; wk.planes.ns = d[6]; wk.planes.W = W; wk.planes.H = H; wk.planes.D = D; wk.planes.b = b;
        wk.planes.tau = prm->hist_dist_threshold; wk.planes.uni_n = uni_n;
    }
    {
        const int rc = bcd_hip_denoise(ctx, d[5], d[6], d[7], d[8], W, H, D, nb_scales, prm, d[4]);
        ctx->main.planes.ready = false; // (consumed by the finest scale's similarity stage; never left behind by a call that failed earlier)
        if (rc != BCD_HIP_OK) return rc;
    }
    // checkAndPutToZeroNegativeInfNaNValues (src/cli/main.cpp:389-420, 470)
    if (opt && opt->zero_bad_values) HIPCHK(ctx, bcd_launch_zero_bad(d[4], (int64_t)np * 3, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h_out, d[4], sz[4] * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return BCD_HIP_OK;
}

int bcd_hip_last_upload_bytes(const bcd_hip_ctx *ctx, int64_t *hist_bytes, int64_t *hist_bytes_sent)
{
    if (!ctx || !hist_bytes || !hist_bytes_sent) return BCD_HIP_EINVAL;
    *hist_bytes = ctx->upload_raw_bytes;
    *hist_bytes_sent = ctx->upload_sent_bytes;
    return BCD_HIP_OK;
}

int bcd_hip_denoise_host(bcd_hip_ctx *ctx, const float *h_colors, const float *h_ns, const float *h_hist, const float *h_cov,
                         int W, int H, int D, int nb_scales, const bcd_hip_params *prm, float *h_out)
{
    return bcd_hip_denoise_host_ex(ctx, h_colors, h_ns, h_hist, h_cov, W, H, D, nb_scales, prm, nullptr, h_out);
}

int bcd_hip_set_progress_callback(bcd_hip_ctx *ctx, bcd_hip_progress_fn fn, void *user)
{
    if (!ctx) return BCD_HIP_EINVAL;
    std::lock_guard<std::mutex> lock(ctx->progress_mutex);
    ctx->progress_fn = fn;
    ctx->progress_user = user;
    return BCD_HIP_OK;
}

// ---- stages -------------------------------------------------------------------------------------------
int bcd_hip_pixel_cov(bcd_hip_ctx *ctx, const float *d_cov, const float *d_ns, int W, int H, float *d_out)
{
    if (!ctx || !d_cov || !d_ns || !d_out || W <= 0 || H <= 0) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    HIPCHK(ctx, bcd_launch_pixel_cov(d_cov, d_ns, (int64_t)W * H, d_out, ctx->stream));
    return BCD_HIP_OK;
}

int bcd_hip_scale_begin(bcd_hip_ctx *ctx, const float *d_cov, const float *d_ns, int W, int H, float *d_pixcov, float *d_sum, int32_t *d_count)
{
    if (!ctx || !d_cov || !d_ns || !d_pixcov || !d_sum || !d_count || W <= 0 || H <= 0) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    Work &wk = ctx->main;
    constexpr size_t LINE_INTS = (size_t)BCD_CNT_LINES * BCD_CNT_STRIDE;
    RCCHK(ensure(ctx, wk.counters, 64 * sizeof(int32_t)));
    RCCHK(ensure(ctx, wk.cnt_lines, ROUND_BATCH * LINE_INTS * sizeof(int)));
    RCCHK(ensure(ctx, wk.work_q, BCD_WORK_INTS * sizeof(int32_t)));
    HIPCHK(ctx, bcd_launch_pixel_cov_clear(d_cov, d_ns, (int64_t)W * H, d_pixcov, d_sum, d_count, wk.stream));
    HIPCHK(ctx, bcd_launch_scale_begin((int *)wk.counters.p, 64, -1, -1, (int *)wk.cnt_lines.p, (int)(ROUND_BATCH * LINE_INTS), (int *)wk.work_q.p, BCD_WORK_INTS, wk.stream));
    // ("clean" = zero because nobody has used it since: every user of these buffers takes the note and clears it)
    wk.clean_flags = wk.clean_lines = wk.clean_dc = wk.clean_wq = true;
    wk.planes.ready = false;
    return BCD_HIP_OK;
}

int bcd_hip_similarity_masks(bcd_hip_ctx *ctx, const float *d_hist, const float *d_ns, int W, int H, int D, int w, int b, float tau,
                             uint32_t *d_mask, int32_t *d_count)
{
    if (!ctx || !d_hist || !d_ns || !d_mask || !d_count) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    bcd_hip_params p; bcd_hip_default_params(&p); p.patch_radius = w; p.search_radius = b;
    RCCHK(check_params(ctx, W, H, D, &p));
    return similarity(ctx, ctx->main, d_hist, d_ns, W, H, D, w, b, tau, d_mask, d_count);
}

int bcd_hip_similarity_masks_deferred(bcd_hip_ctx *ctx, const float *d_hist, const float *d_ns, int W, int H, int D, int w, int b, float tau,
                                      uint32_t *d_mask, int32_t *d_count)
{
    if (!ctx || !d_hist || !d_ns || !d_mask || !d_count) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    bcd_hip_params p; bcd_hip_default_params(&p); p.patch_radius = w; p.search_radius = b;
    RCCHK(check_params(ctx, W, H, D, &p));
    return similarity(ctx, ctx->main, d_hist, d_ns, W, H, D, w, b, tau, d_mask, d_count, 2);
}

int bcd_hip_similarity_masks_verdict(bcd_hip_ctx *ctx, int *redo)
{
    if (!ctx || !redo) return BCD_HIP_EINVAL;
    *redo = similarity_redo_mode(ctx->main) != 0 ? 1 : 0; // (also keeps the workspace's memory of non-uniform sample counts)
    return BCD_HIP_OK;
}

int bcd_hip_similarity_masks_exact(bcd_hip_ctx *ctx, const float *d_hist, const float *d_ns, int W, int H, int D, int w, int b, float tau,
                                   uint32_t *d_mask, int32_t *d_count)
{
    if (!ctx || !d_hist || !d_ns || !d_mask || !d_count) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    bcd_hip_params p; bcd_hip_default_params(&p); p.patch_radius = w; p.search_radius = b;
    RCCHK(check_params(ctx, W, H, D, &p));
    return similarity(ctx, ctx->main, d_hist, d_ns, W, H, D, w, b, tau, d_mask, d_count, 1);
}

int bcd_hip_window_distances(bcd_hip_ctx *ctx, const float *d_hist, const float *d_ns, int W, int H, int D, int w, int b,
                             int line, int col, float *h_out)
{
    if (!ctx || !d_hist || !d_ns || !h_out) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    bcd_hip_params p; bcd_hip_default_params(&p); p.patch_radius = w; p.search_radius = b;
    RCCHK(check_params(ctx, W, H, D, &p));
    if (line < w || line > H - 1 - w || col < w || col > W - 1 - w) return bad(ctx, "not a main pixel");
    const size_t npix = (size_t)W * H;
    const int nd = bcd_delta_count(b), n = (2 * b + 1) * (2 * b + 1);
    RCCHK(ensure(ctx, ctx->main.T, npix * nd * sizeof(float)));
    RCCHK(ensure(ctx, ctx->main.Cn, npix * nd));
    RCCHK(ensure(ctx, ctx->tmp_lo, n * sizeof(float)));
    HIPCHK(ctx, bcd_launch_pairdist(d_hist, d_ns, W, H, D, b, (float *)ctx->main.T.p, (uint8_t *)ctx->main.Cn.p, 0, nullptr, 0.f, ctx->stream));
    HIPCHK(ctx, bcd_launch_window_distances((const float *)ctx->main.T.p, (const uint8_t *)ctx->main.Cn.p, W, H, w, b, line, col, (float *)ctx->tmp_lo.p, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h_out, ctx->tmp_lo.p, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return BCD_HIP_OK;
}

int bcd_hip_active_set(bcd_hip_ctx *ctx, const uint32_t *d_mask, const int32_t *d_count, int W, int H, int w, int b,
                       int main_row_begin, int main_row_end, float skip_probability, int random_order, uint32_t seed,
                       uint8_t *d_state, int32_t *rounds)
{
    if (!ctx || !d_mask || !d_count || !d_state) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    return active_set(ctx, ctx->main, d_mask, d_count, W, H, w, b, main_row_begin, main_row_end, skip_probability, random_order, seed, d_state, rounds);
}

int bcd_hip_active_init(bcd_hip_ctx *ctx, const int32_t *d_count, int W, int H, int w, int main_row_begin, int main_row_end,
                        float skip_probability, uint32_t seed, int row_offset, uint8_t *d_state)
{
    if (!ctx || !d_count || !d_state || W <= 0 || H <= 0) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    HIPCHK(ctx, bcd_launch_active_init(d_count, W, H, w, main_row_begin, main_row_end, skip_probability, seed, row_offset, d_state, ctx->stream));
    ctx->main.dep_ready = false;
    return BCD_HIP_OK;
}

int bcd_hip_active_step(bcd_hip_ctx *ctx, const uint32_t *d_mask, const int32_t *d_count, int W, int H, int w, int b, int main_row_begin,
                        int main_row_end, int random_order, uint32_t seed, int row_offset, int first_pass, uint8_t *d_state, int32_t *undecided)
{
    if (!ctx || !d_mask || !d_count || !d_state || !undecided) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    int u = 0;
    // the dependency lists extracted by the first step after bcd_hip_active_init stay valid for the following steps of the same
    // marking problem on the same buffers (masks and counts do not change; a pixel that is still undecided keeps its list)
    RCCHK(active_step(ctx, ctx->main, d_mask, d_count, W, H, w, b, main_row_begin, main_row_end, random_order, seed, row_offset, first_pass != 0,
                      d_state, &u, nullptr));
    *undecided = u;
    return BCD_HIP_OK;
}

int bcd_hip_active_step_enqueue(bcd_hip_ctx *ctx, const uint32_t *d_mask, const int32_t *d_count, int W, int H, int w, int b, int main_row_begin,
                                int main_row_end, int random_order, uint32_t seed, int row_offset, uint8_t *d_state, int64_t *d_total, int with_verdict)
{
    if (!ctx || !d_mask || !d_count || !d_state) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    static_assert(sizeof(long long) == sizeof(int64_t), "64-bit counters");
    return active_step_enqueue(ctx, ctx->main, d_mask, d_count, W, H, w, b, main_row_begin, main_row_end, random_order, seed, row_offset, d_state,
                               reinterpret_cast<long long *>(d_total), with_verdict != 0);
}

int bcd_hip_active_step_collect(bcd_hip_ctx *ctx, int32_t *undecided, int32_t *launches)
{
    if (!ctx || !undecided) return BCD_HIP_EINVAL;
    int u = 0, n = 0;
    active_step_collect(ctx->main, &u, &n);
    *undecided = u;
    if (launches) *launches = n;
    return BCD_HIP_OK;
}

int bcd_hip_bayes_accumulate(bcd_hip_ctx *ctx, const float *d_colors, const float *d_pixcov, const uint32_t *d_mask,
                             const int32_t *d_nsim, const uint8_t *d_state, int W, int H, int w, int b, float min_eig,
                             float *d_sum, int32_t *d_count)
{
    if (!ctx || !d_colors || !d_pixcov || !d_mask || !d_nsim || !d_state || !d_sum || !d_count) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    return bayes(ctx, ctx->main, d_colors, d_pixcov, d_mask, d_nsim, d_state, W, H, w, b, min_eig, d_sum, d_count);
}

int bcd_hip_bayes_accumulate_rows(bcd_hip_ctx *ctx, const float *d_colors, const float *d_pixcov, const uint32_t *d_mask,
                                  const int32_t *d_nsim, const uint8_t *d_state, int W, int H, int w, int b, float min_eig,
                                  float *d_sum, int32_t *d_count, int row_begin, int row_end, const int64_t *d_skip_if, const int64_t *h_skip_if, int *skipped)
{
    if (!ctx || !d_colors || !d_pixcov || !d_mask || !d_nsim || !d_state || !d_sum || !d_count) return bad(ctx, "bad argument");
    if (d_skip_if && (!h_skip_if || !skipped)) return bad(ctx, "a speculative estimate needs the host copy of its word and a place for the verdict");
    DEVICE_GUARD(ctx);
    bool sk = false;
    const int rc = bayes(ctx, ctx->main, d_colors, d_pixcov, d_mask, d_nsim, d_state, W, H, w, b, min_eig, d_sum, d_count, false, row_begin, row_end,
                         reinterpret_cast<const long long *>(d_skip_if), reinterpret_cast<const long long *>(h_skip_if), d_skip_if ? &sk : nullptr);
    if (skipped) *skipped = sk ? 1 : 0;
    return rc;
}

int bcd_hip_finalize(bcd_hip_ctx *ctx, const float *d_sum, const int32_t *d_count, int64_t npix, float *d_out)
{
    if (!ctx || !d_sum || !d_count || !d_out || npix <= 0) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    HIPCHK(ctx, bcd_launch_finalize(d_sum, d_count, npix, d_out, ctx->stream));
    return BCD_HIP_OK;
}

int bcd_hip_finalize_band(bcd_hip_ctx *ctx, const float *d_sum, const int32_t *d_count, int W, int rows, int halo, const float *d_up_sum,
                          const int32_t *d_up_count, const float *d_down_sum, const int32_t *d_down_count, float *d_out)
{
    if (!ctx || !d_sum || !d_count || !d_out || W <= 0 || rows <= 0 || halo < 0) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    if ((d_up_sum == nullptr) != (d_up_count == nullptr) || (d_down_sum == nullptr) != (d_down_count == nullptr)) return bad(ctx, "halo sum without count");
    if ((d_up_sum || d_down_sum) && halo > rows) return bad(ctx, "halo larger than the band");
    HIPCHK(ctx, bcd_launch_finalize_band(d_sum, d_count, W, rows, halo, d_up_sum, d_up_count, d_down_sum, d_down_count, d_out, ctx->stream));
    return BCD_HIP_OK;
}

int bcd_hip_downscale_sum(bcd_hip_ctx *ctx, const float *d_in, int W, int H, int D, float *d_out)
{
    if (!ctx || !d_in || !d_out || W < 2 || H < 2 || D <= 0) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    HIPCHK(ctx, bcd_launch_downscale(0, d_in, W, H, D, d_out, ctx->stream));
    return BCD_HIP_OK;
}

int bcd_hip_downscale_avg(bcd_hip_ctx *ctx, const float *d_in, int W, int H, int D, float *d_out)
{
    if (!ctx || !d_in || !d_out || W < 2 || H < 2 || D <= 0) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    HIPCHK(ctx, bcd_launch_downscale(1, d_in, W, H, D, d_out, ctx->stream));
    return BCD_HIP_OK;
}

int bcd_hip_downscale_cov(bcd_hip_ctx *ctx, const float *d_cov, const float *d_ns, int W, int H, float *d_out)
{
    if (!ctx || !d_cov || !d_ns || !d_out || W < 2 || H < 2) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    HIPCHK(ctx, bcd_launch_downscale_cov(d_cov, d_ns, W, H, d_out, ctx->stream));
    return BCD_HIP_OK;
}

int bcd_hip_interpolate(bcd_hip_ctx *ctx, const float *d_lo, int w, int h, int D, float *d_hi, int W, int H)
{
    if (!ctx || !d_lo || !d_hi || w != W / 2 || h != H / 2 || w <= 0 || h <= 0 || D <= 0) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    HIPCHK(ctx, bcd_launch_interpolate(0, d_lo, w, h, D, d_hi, W, H, ctx->stream));
    return BCD_HIP_OK;
}

int bcd_hip_merge(bcd_hip_ctx *ctx, float *d_hi, int W, int H, const float *d_lo, int D)
{
    if (!ctx || !d_hi || !d_lo || W < 2 || H < 2 || D <= 0) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    return merge_on(ctx, ctx->main, d_hi, W, H, d_lo, D);
}

int bcd_hip_spike_filter(bcd_hip_ctx *ctx, const float *d_col, const float *d_ns, const float *d_hist, const float *d_cov, int W, int H,
                         int D, float factor, float *o_col, float *o_ns, float *o_hist, float *o_cov)
{
    if (!ctx || !d_col || !d_ns || !d_hist || !d_cov || !o_col || !o_ns || !o_hist || !o_cov) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    if (W < 3 || H < 3 || D <= 0) return bad(ctx, "image smaller than 3x3");
    HIPCHK(ctx, bcd_launch_spike(d_col, d_ns, d_hist, d_cov, W, H, D, factor, o_col, o_ns, o_hist, o_cov, ctx->stream));
    return BCD_HIP_OK;
}

int bcd_hip_accumulate_samples(bcd_hip_ctx *ctx, const float *d_samples, const float *d_weights, int W, int H, int spp, int nb_bins,
                               float gamma, float max_value, float *d_nsamples, float *d_mean, float *d_cov, float *d_hist)
{
    if (!ctx || !d_samples || !d_nsamples || !d_mean || !d_cov || !d_hist) return bad(ctx, "null pointer");
    DEVICE_GUARD(ctx);
    if (W <= 0 || H <= 0 || spp <= 0 || nb_bins < 3) return bad(ctx, "bad size");
    if ((size_t)3 * nb_bins * 64 * sizeof(float) > 160 * 1024) { set_err(ctx, "more than 213 bins per channel are not supported"); return BCD_HIP_EUNSUPPORTED; }
    HIPCHK(ctx, bcd_launch_accumulate_samples(d_samples, d_weights, (int64_t)W * H, spp, nb_bins, gamma, max_value, d_nsamples, d_mean, d_cov, d_hist, ctx->stream));
    return BCD_HIP_OK;
}

int bcd_hip_zero_bad_values(bcd_hip_ctx *ctx, float *d_img, int64_t n)
{
    if (!ctx || !d_img || n <= 0) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    HIPCHK(ctx, bcd_launch_zero_bad(d_img, n, ctx->stream));
    return BCD_HIP_OK;
}

int bcd_hip_selftest_distance_kernels(bcd_hip_ctx *ctx, const float *d_hist, const float *d_ns, int W, int H, int D, int search_radius,
                                      int *variant, int64_t *mismatches)
{
    if (!ctx || !d_hist || !d_ns || !mismatches || W <= 0 || H <= 0 || D <= 0 || search_radius < 1) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    touch(ctx->main);
    Work &wk = ctx->main;
    const size_t npix = (size_t)W * H;
    const int nd = bcd_delta_count(search_radius);
    RCCHK(ensure(ctx, wk.T, npix * nd * sizeof(float)));
    RCCHK(ensure(ctx, wk.Cn, count_plane_bytes(npix, nd)));
    RCCHK(ensure(ctx, wk.counters, 64 * sizeof(int32_t)));
    float *T2 = nullptr;
    uint8_t *C2 = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&T2, npix * nd * sizeof(float)));
    if (hipMalloc((void **)&C2, npix * nd) != hipSuccess) { (void)hipFree(T2); set_err(ctx, "hipMalloc"); return BCD_HIP_EDEVICE; }
    int rc = BCD_HIP_OK;
    do {
        int *d_flag = (int *)wk.counters.p + 40;
        unsigned long long *d_cnt = reinterpret_cast<unsigned long long *>((int32_t *)wk.counters.p + 32);
        if (hipMemsetAsync(d_flag, 0, 2 * sizeof(int), wk.stream) != hipSuccess || hipMemsetAsync(d_cnt, 0, sizeof(*d_cnt), wk.stream) != hipSuccess ||
            bcd_launch_uniform_n(d_ns, (int64_t)npix, d_flag + 1, wk.stream) != hipSuccess ||
            hipMemcpyAsync(wk.h_counters + 41, d_flag + 1, sizeof(int), hipMemcpyDeviceToHost, wk.stream) != hipSuccess ||
            hipMemcpyAsync(wk.h_counters + 42, d_ns, sizeof(float), hipMemcpyDeviceToHost, wk.stream) != hipSuccess ||
            hipStreamSynchronize(wk.stream) != hipSuccess) { rc = BCD_HIP_EDEVICE; break; }
        float n0, uni_n = 0.f;
        memcpy(&n0, wk.h_counters + 42, sizeof(n0));
        int e = 0;
        if (wk.h_counters[41] == 0 && n0 >= 1.f && n0 <= 65536.f && frexpf(n0, &e) == 0.5f) uni_n = n0;
        // (entries whose neighbour lies outside the image are never written: clear both sets first)
        if (hipMemsetAsync(wk.T.p, 0, npix * nd * sizeof(float), wk.stream) != hipSuccess || hipMemsetAsync(wk.Cn.p, 0, npix * nd, wk.stream) != hipSuccess ||
            hipMemsetAsync(T2, 0, npix * nd * sizeof(float), wk.stream) != hipSuccess || hipMemsetAsync(C2, 0, npix * nd, wk.stream) != hipSuccess) {
            rc = BCD_HIP_EDEVICE; break;
        }
        // production choice (fast division, uniform-count formula when it applies) against the compiler's division + general formula
        if (bcd_launch_pairdist(d_hist, d_ns, W, H, D, search_radius, (float *)wk.T.p, (uint8_t *)wk.Cn.p, 1, d_flag, uni_n, wk.stream) != hipSuccess ||
            bcd_launch_pairdist(d_hist, d_ns, W, H, D, search_radius, T2, C2, 0, d_flag, 0.f, wk.stream) != hipSuccess ||
            bcd_launch_compare_planes((const float *)wk.T.p, (const uint8_t *)wk.Cn.p, T2, C2, (int64_t)(npix * nd), d_cnt, wk.stream) != hipSuccess) {
            rc = BCD_HIP_EDEVICE; break;
        }
        unsigned long long h = 0;
        int flag = 0;
        if (hipMemcpyAsync(&h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, wk.stream) != hipSuccess ||
            hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, wk.stream) != hipSuccess ||
            hipStreamSynchronize(wk.stream) != hipSuccess) { rc = BCD_HIP_EDEVICE; break; }
        *mismatches = (int64_t)h;
        if (variant) *variant = (uni_n > 0.f ? 2 : 1) | (flag << 4); // 1 = fast division, 2 = + uniform counts; bits 4.. = range / count flags raised
    } while (false);
    (void)hipFree(T2);
    (void)hipFree(C2);
    if (rc != BCD_HIP_OK) set_err(ctx, "distance kernel self-test failed to run");
    return rc;
}

// Measurement (bench.py `roofline.valu`): what the production distance kernel computes on this frame -- the (pixel pair, bin) terms it
// evaluates (exactly the reference's count of bins with b1 + b2 > 1 over the half plane), the bins a wavefront issues because one of its 64
// pairs needs them, the groups of four bins it enters -- from a counting instantiation of the kernel, and the duration of the PRODUCTION
// instantiation on the same input (HIP events, best of `reps`).
int bcd_hip_selftest_bin_work(bcd_hip_ctx *ctx, const float *d_hist, const float *d_ns, int W, int H, int D, int search_radius, int reps,
                              int64_t *lane_bins, int64_t *wave_bins, int64_t *wave_groups, float *kernel_ms)
{
    if (!ctx || !d_hist || !d_ns || !lane_bins || !wave_bins || !wave_groups || !kernel_ms || W <= 0 || H <= 0 || search_radius < 1 || reps < 1) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    touch(ctx->main);
    if (!bcd_pairdist_rw_supported(D)) { set_err(ctx, "no approximate kernel for this histogram depth"); return BCD_HIP_EUNSUPPORTED; }
    Work &wk = ctx->main;
    const size_t npix = (size_t)W * H;
    const int nd = bcd_delta_count(search_radius);
    RCCHK(ensure(ctx, wk.T, npix * nd * sizeof(float)));
    RCCHK(ensure(ctx, wk.Cn, count_plane_bytes(npix, nd)));
    RCCHK(ensure(ctx, wk.counters, 64 * sizeof(int32_t)));
    int *d_flag = (int *)wk.counters.p + 40;
    unsigned long long *d_work = reinterpret_cast<unsigned long long *>((int32_t *)wk.counters.p + 48); // (8-byte aligned: words 48..53)
    HIPCHK(ctx, hipMemsetAsync(d_flag, 0, 4 * sizeof(int), wk.stream));
    HIPCHK(ctx, hipMemsetAsync(d_work, 0, 3 * sizeof(unsigned long long), wk.stream));
    // the uniform kernel on the first pixel's count if every pixel carries it (the kernel checks), else the general formula -- like similarity()
    float uni_n = -1.f;
    HIPCHK(ctx, bcd_launch_pairdist_rw_counting(d_hist, d_ns, W, H, D, search_radius, wk.T.p, (uint8_t *)wk.Cn.p, d_flag, uni_n, d_work, wk.stream));
    int flags[4] = { 0, 0, 0, 0 };
    HIPCHK(ctx, hipMemcpyAsync(flags, d_flag, sizeof(flags), hipMemcpyDeviceToHost, wk.stream));
    HIPCHK(ctx, hipStreamSynchronize(wk.stream));
    if (flags[2] != 0) { // not one power-of-two count: count again with the general formula
        uni_n = 0.f;
        HIPCHK(ctx, hipMemsetAsync(d_flag, 0, 4 * sizeof(int), wk.stream));
        HIPCHK(ctx, hipMemsetAsync(d_work, 0, 3 * sizeof(unsigned long long), wk.stream));
        HIPCHK(ctx, bcd_launch_pairdist_rw_counting(d_hist, d_ns, W, H, D, search_radius, wk.T.p, (uint8_t *)wk.Cn.p, d_flag, uni_n, d_work, wk.stream));
    }
    unsigned long long h[3] = { 0, 0, 0 };
    HIPCHK(ctx, hipMemcpyAsync(h, d_work, sizeof(h), hipMemcpyDeviceToHost, wk.stream));
    hipEvent_t e0, e1;
    HIPCHK(ctx, hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); set_err(ctx, "hipEventCreate"); return BCD_HIP_EDEVICE; }
    float best = -1.f;
    int rc = BCD_HIP_OK;
    for (int r = 0; r < reps + 1 && rc == BCD_HIP_OK; ++r) { // (the first one warms up)
        if (hipEventRecord(e0, wk.stream) != hipSuccess ||
            bcd_launch_pairdist_rw(d_hist, d_ns, W, H, D, search_radius, wk.T.p, (uint8_t *)wk.Cn.p, d_flag, uni_n, wk.stream) != hipSuccess ||
            hipEventRecord(e1, wk.stream) != hipSuccess || hipStreamSynchronize(wk.stream) != hipSuccess) { rc = BCD_HIP_EDEVICE; break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && (best < 0.f || ms < best)) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != BCD_HIP_OK) { set_err(ctx, "bin-work self-test failed to run"); return rc; }
    *lane_bins = (int64_t)h[0]; *wave_bins = (int64_t)h[1]; *wave_groups = (int64_t)h[2]; *kernel_ms = best;
    return BCD_HIP_OK;
}

int bcd_hip_selftest_approx_distance(bcd_hip_ctx *ctx, const float *d_hist, const float *d_ns, int W, int H, int D, int search_radius,
                                     float *max_rel_dev, int64_t *count_mismatches, int *flags)
{
    if (!ctx || !d_hist || !d_ns || !max_rel_dev || !count_mismatches || W <= 0 || H <= 0 || search_radius < 1) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    touch(ctx->main);
    if (!bcd_pairdist_rw_supported(D)) { set_err(ctx, "no approximate kernel for this histogram depth"); return BCD_HIP_EUNSUPPORTED; }
    Work &wk = ctx->main;
    const size_t npix = (size_t)W * H;
    const int nd = bcd_delta_count(search_radius);
    RCCHK(ensure(ctx, wk.T, npix * nd * sizeof(float)));
    RCCHK(ensure(ctx, wk.Cn, count_plane_bytes(npix, nd)));
    RCCHK(ensure(ctx, wk.counters, 64 * sizeof(int32_t)));
    float *T2 = nullptr;
    uint8_t *C2 = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&T2, npix * nd * sizeof(float)));
    if (hipMalloc((void **)&C2, npix * nd) != hipSuccess) { (void)hipFree(T2); set_err(ctx, "hipMalloc"); return BCD_HIP_EDEVICE; }
    int rc = BCD_HIP_OK;
    do {
        int *d_flag = (int *)wk.counters.p + 40;
        unsigned int *d_res = reinterpret_cast<unsigned int *>((int32_t *)wk.counters.p + 32);
        if (hipMemsetAsync(d_flag, 0, 4 * sizeof(int), wk.stream) != hipSuccess || hipMemsetAsync(d_res, 0, 2 * sizeof(unsigned int), wk.stream) != hipSuccess ||
            bcd_launch_uniform_n(d_ns, (int64_t)npix, d_flag + 1, wk.stream) != hipSuccess ||
            hipMemcpyAsync(wk.h_counters + 41, d_flag + 1, sizeof(int), hipMemcpyDeviceToHost, wk.stream) != hipSuccess ||
            hipMemcpyAsync(wk.h_counters + 42, d_ns, sizeof(float), hipMemcpyDeviceToHost, wk.stream) != hipSuccess ||
            hipStreamSynchronize(wk.stream) != hipSuccess) { rc = BCD_HIP_EDEVICE; break; }
        float n0, uni_n = 0.f;
        memcpy(&n0, wk.h_counters + 42, sizeof(n0));
        int e = 0;
        if (wk.h_counters[41] == 0 && n0 >= 1.f && n0 <= 65536.f && frexpf(n0, &e) == 0.5f) uni_n = n0;
        if (hipMemsetAsync(wk.T.p, 0, npix * nd * sizeof(float), wk.stream) != hipSuccess || hipMemsetAsync(wk.Cn.p, 0, npix * nd, wk.stream) != hipSuccess ||
            hipMemsetAsync(T2, 0, npix * nd * sizeof(float), wk.stream) != hipSuccess || hipMemsetAsync(C2, 0, npix * nd, wk.stream) != hipSuccess) {
            rc = BCD_HIP_EDEVICE; break;
        }
        // approximate planes (production variant: the uniform kernel, or -- general sample counts -- the RATIO form with its verdict in flag bit 2) against the
        // exact planes (compiler's division, general formula)
        if (uni_n == 0.f && ensure(ctx, wk.ratio_stats, 128 * sizeof(unsigned int)) != BCD_HIP_OK) { rc = BCD_HIP_ENOMEM; break; }
        if ((uni_n != 0.f ? bcd_launch_pairdist_rw(d_hist, d_ns, W, H, D, search_radius, wk.T.p, (uint8_t *)wk.Cn.p, d_flag, uni_n, wk.stream)
                          : bcd_launch_pairdist_rw_ratio(d_hist, d_ns, W, H, D, search_radius, wk.T.p, (uint8_t *)wk.Cn.p, d_flag, 1.f, (unsigned int *)wk.ratio_stats.p, wk.stream)) != hipSuccess ||
            bcd_launch_pairdist(d_hist, d_ns, W, H, D, search_radius, T2, C2, 0, d_flag, 0.f, wk.stream) != hipSuccess ||
            bcd_launch_max_rel_dev((const float *)wk.T.p, T2, (const uint8_t *)wk.Cn.p, C2, W, H, search_radius, d_res, wk.stream) != hipSuccess) {
            rc = BCD_HIP_EDEVICE; break;
        }
        unsigned int h[2] = { 0u, 0u };
        int flag = 0;
        if (hipMemcpyAsync(h, d_res, sizeof(h), hipMemcpyDeviceToHost, wk.stream) != hipSuccess ||
            hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, wk.stream) != hipSuccess ||
            hipStreamSynchronize(wk.stream) != hipSuccess) { rc = BCD_HIP_EDEVICE; break; }
        memcpy(max_rel_dev, &h[0], sizeof(float));
        *count_mismatches = (int64_t)h[1];
        if (flags) *flags = (uni_n > 0.f ? 2 : 3) | (flag << 4); // low nibble: 2 = uniform kernel, 3 = RATIO form; above: the kernels' flag word (4: the RATIO form declined)
    } while (false);
    (void)hipFree(T2);
    (void)hipFree(C2);
    if (rc != BCD_HIP_OK) set_err(ctx, "approximate-distance self-test failed to run");
    return rc;
}

int bcd_hip_eig27_batch(bcd_hip_ctx *ctx, const float *d_A, int n, float *d_eig, float *d_V, float *ms)
{
    if (!ctx || !d_A || !d_eig || !d_V || n <= 0) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    touch(ctx->main);
    Work &wk = ctx->main;
    RCCHK(ensure(ctx, wk.counters, 64 * sizeof(int32_t)));
    RCCHK(ensure(ctx, wk.work_q, BCD_WORK_INTS * sizeof(int32_t)));
    int32_t *d_c = (int32_t *)wk.work_q.p; // work queues
    HIPCHK(ctx, hipMemsetAsync(d_c, 0, BCD_WORK_INTS * sizeof(int32_t), wk.stream));
    HIPCHK(ctx, hipEventRecord(wk.ev_stage[0], wk.stream));
    HIPCHK(ctx, bcd_launch_jacobi27_batch(d_A, n, d_c, std::min(ctx->num_cus * 12, (n + 1) / 2), d_eig, d_V, wk.stream));
    HIPCHK(ctx, hipEventRecord(wk.ev_stage[1], wk.stream));
    HIPCHK(ctx, hipStreamSynchronize(wk.stream));
    if (ms) *ms = stage_ms(wk, 0, 1);
    return BCD_HIP_OK;
}

int bcd_hip_selftest_division(bcd_hip_ctx *ctx, uint32_t seed, int64_t samples, int64_t *mismatches)
{
    if (!ctx || !mismatches || samples <= 0) return bad(ctx, "bad argument");
    DEVICE_GUARD(ctx);
    touch(ctx->main);
    RCCHK(ensure(ctx, ctx->main.counters, 64 * sizeof(int32_t)));
    unsigned long long *d = reinterpret_cast<unsigned long long *>((int32_t *)ctx->main.counters.p + 32);
    HIPCHK(ctx, hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream));
    const int per_thread = 1024, blocks = (int)std::min<int64_t>(1 << 20, (samples + 256ll * per_thread - 1) / (256ll * per_thread));
    HIPCHK(ctx, bcd_launch_selftest_div(seed, blocks, per_thread, d, ctx->stream));
    unsigned long long h = 0;
    HIPCHK(ctx, hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *mismatches = (int64_t)h;
    return BCD_HIP_OK;
}

// ---- host utilities -------------------------------------------------------------------------------------
uint32_t bcd_hip_scale_seed(uint32_t seed0, int scale) { return seed0 + (uint32_t)scale; }

uint32_t bcd_hip_strip_order_seed(int W, int H, int patch_radius, int search_radius) { return bcd_strip_order_seed(W, H, patch_radius, search_radius); }

int bcd_hip_visit_order(int W, int H, int w, int random_order, uint32_t seed, int32_t *h_order)
{
    if (!h_order || W < 2 * w + 1 || H < 2 * w + 1 || w < 0) return BCD_HIP_EINVAL;
    std::vector<uint64_t> keys;
    keys.reserve((size_t)(W - 2 * w) * (H - 2 * w));
    for (int l = w; l <= H - 1 - w; ++l)
        for (int c = w; c <= W - 1 - w; ++c) keys.push_back(bcd_order_key((uint32_t)(l * W + c), random_order, seed));
    std::sort(keys.begin(), keys.end());
    for (size_t i = 0; i < keys.size(); ++i) h_order[i] = (int32_t)(keys[i] & 0xffffffffu);
    return BCD_HIP_OK;
}

} // extern "C"
