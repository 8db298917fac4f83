/*
 * bcd_hip.h -- C ABI of the MI355X (gfx950) denoising engine: libbcd_hip.so
 *
 * This is the drop-in boundary UNDER the reference's C++ API.  The reference has no C ABI
 * (SURVEY.md 8b); the entry points below are exactly the calls that the library's own
 * bcd::Denoiser::denoise() / bcd::MultiscaleDenoiser::denoise() (bcd_amd/host/) make, i.e. what
 * replaces, in the reference:
 *     Denoiser::denoise()                 src/core/Denoiser.cpp:84-212
 *     MultiscaleDenoiser::denoise()       src/core/MultiscaleDenoiser.cpp:31-136
 *     DenoisingUnit::*                    src/core/DenoisingUnit.cpp:157-693
 *     CudaHistogramDistance (per-pixel CUDA offload, not reproduced)  src/core/CudaHistogramDistance.cu:164-239
 *
 * Conventions
 *   - every image is the reference's interleaved DeepImage layout, fp32:
 *         index = (line * W + col) * depth + d      (include/bcd/core/DeepImage.hpp:385-396)
 *     colours depth 3, nbOfSamples depth 1, histograms depth D (= 3 x bins), covariances depth 6 in
 *     the order xx,yy,zz,yz,xz,xy (include/bcd/core/CovarianceMatrix.h:18-27).
 *   - pointers named d_* are DEVICE pointers (HBM), h_* are host pointers.
 *   - every function returns 0 on success, a negative BCD_HIP_E* code otherwise, and never calls
 *     exit() (unlike HANDLE_ERROR, include/bcd/core/CudaUtils.h:18-30).
 *   - work is enqueued on the context's stream; calls that return host-visible results synchronise it.
 */
#ifndef BCD_HIP_H
#define BCD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BCD_HIP_OK 0
#define BCD_HIP_EINVAL (-1)   /* null / empty / mismatched inputs (Denoiser.cpp:266-347 returns false) */
#define BCD_HIP_EDEVICE (-2)  /* HIP runtime error, message in bcd_hip_last_error()                     */
#define BCD_HIP_ENOMEM (-3)
#define BCD_HIP_EUNSUPPORTED (-4)

typedef struct bcd_hip_ctx bcd_hip_ctx;

/* IDenoiser::setProgressCallback (include/bcd/core/IDenoiser.h:91, fired from src/core/Denoiser.cpp:181-192): called from the
 * engine's host threads, serialised, with monotone values in (0, 1] -- every scale reports when its processed set is known and
 * when its estimate is complete, weighted by its share of the pixels */
typedef void (*bcd_hip_progress_fn)(float progress, void *user);

/* mirrors bcd::DenoiserParameters (include/bcd/core/IDenoiser.h:20-44) */
typedef struct bcd_hip_params {
    float    hist_dist_threshold;     /* m_histogramDistanceThreshold          default 1     */
    int32_t  patch_radius;            /* m_patchRadius                         default 1     */
    int32_t  search_radius;           /* m_searchWindowRadius                  default 6     */
    float    min_eigen_value;         /* m_minEigenValue                       default 1e-8  */
    int32_t  use_random_pixel_order;  /* m_useRandomPixelOrder                 default 1     (0 scanline, 1 seeded shuffle, 2 the reference's
                                         multi-thread -r 0 list: even strips of 2b lines, then the odd ones, Denoiser.cpp:381-414; single GPU only) */
    float    marked_skip_probability; /* m_markedPixelsSkippingProbability     default 1     */
    uint32_t order_seed;              /* seed of the visiting order; the reference seeds its
                                         shuffle from the wall clock (Denoiser.cpp:418)      */
} bcd_hip_params;

/* per-scale counters filled by the denoise calls (cf. COMPUTE_DENOISING_STATS,
 * include/bcd/core/DenoisingUnit.h:35-65) */
typedef struct bcd_hip_scale_stats {
    int32_t width, height;
    int64_t main_pixels;      /* (W-2w)(H-2w)                                    */
    int64_t processed;        /* main pixels not skipped                          */
    int64_t fallback;         /* processed through the <3P+1 similar-patch path   */
    int64_t similar_total;    /* sum of |S| over processed pixels                 */
    int32_t active_rounds;    /* fixed-point rounds of the marking strategy       */
    float   ms_similarity;    /* GPU time of the distance + mask kernels (events) */
    float   ms_active;
    float   ms_bayes;
    float   ms_total;
    int32_t similarity_path;  /* 1 = approximate planes + exact verification at the threshold, 2 = the same with the RATIO form of the distance
                               * kernel (general sample counts: any counts that are not one power of two), 0 = exact planes */
    int32_t borderline_pairs; /* pairs re-evaluated exactly (similarity_path >= 1)  */
    int32_t cu_share;         /* share (%) of the CU slots this scale's persistent estimate kernels took (100: all)  */
    int32_t spectral_inverses; /* full estimates (3x3 patches, default search radius) whose matrix inverse failed the sweep's checks and took
                                  the spectral branch of inverseSymmetricMatrix in the LDS kernel (normally 0)        */
} bcd_hip_scale_stats;

/* ---- context ------------------------------------------------------------------------------ */
int  bcd_hip_ctx_create(bcd_hip_ctx **ctx, int device, void *hip_stream /* hipStream_t or NULL */);
void bcd_hip_ctx_destroy(bcd_hip_ctx *ctx);
const char *bcd_hip_last_error(const bcd_hip_ctx *ctx);
int  bcd_hip_device_count(void);
void bcd_hip_default_params(bcd_hip_params *p);
/* enable per-stage event timing into the stats (adds stream synchronisations) */
int  bcd_hip_set_profiling(bcd_hip_ctx *ctx, int enabled);
/* multiscale runs drive the (independent) scales concurrently, one HIP stream + host thread each (default on; also
 * disabled by BCD_HIP_SERIAL_SCALES=1).  Results are identical either way. */
int  bcd_hip_set_concurrent_scales(bcd_hip_ctx *ctx, int enabled);
/* similar-patch selection through approximate pair-distance planes (binary16 T plane, exact bin counts) with an exact
 * re-evaluation of every pair within tau (1 +- 2^-10) (BCD_APPROX_DELTA; the approximate distance is within 5e-4 of the exact one).
 * Default on for w = 1, D in {24, 36, 60} and tau in [2^-6, 64] -- other settings take the exact kernels; also disabled by
 * BCD_HIP_EXACT_SIMILARITY=1.  The masks are bit-identical either way; 0 forces the exact kernels. */
int  bcd_hip_set_fast_similarity(bcd_hip_ctx *ctx, int enabled);
/* process-wide: 1 = the eigensolver of the Bayesian steps (Eigen::SelfAdjointEigenSolver of DenoisingUnit.cpp:589,617) runs to off^2 <= 1e-12 diag^2
 * instead of the production rule (2e-9 + first-order correction of the positive part): ~1.5 % of a step for a 3x smaller deviation on
 * ill-conditioned low-sample frames (4K at 8 spp: 2.9e-6 instead of 9.8e-6 from the CPU path).  Default: the environment's BCD_HIP_STRICT_EIGEN (0). */
int  bcd_hip_set_strict_eigensolver(int enabled);
/* share (1..100 %, default 100) of the device's CU slots the persistent estimate kernels of this context occupy.  A caller that
 * runs several contexts on one device at once lowers it for the contexts that have slack, so that the short kernels of the one on
 * the critical path find room beside them (bcd_hip_denoise does this itself for its coarse scales; the multi-GPU driver uses
 * it for its per-scale contexts).  Results do not depend on it. */
int  bcd_hip_set_cu_share(bcd_hip_ctx *ctx, int percent);
int  bcd_hip_set_progress_callback(bcd_hip_ctx *ctx, bcd_hip_progress_fn fn, void *user);
int  bcd_hip_get_stats(const bcd_hip_ctx *ctx, int scale, bcd_hip_scale_stats *out);
/* duration (ms, HIP events on the context's stream) and launch count of the pair-distance kernel
 * accumulated since the last reset -- the dominant kernel measured by bench.py's roofline */
int  bcd_hip_kernel_time(const bcd_hip_ctx *ctx, float *ms_pairdist, int32_t *launches);
int  bcd_hip_reset_kernel_time(bcd_hip_ctx *ctx);

/* ---- whole path, device-resident inputs (what bench.py times) ----------------------------------
 * replaces Denoiser::denoise() (nb_scales == 1) / MultiscaleDenoiser::denoise() (nb_scales > 1).
 * d_out: W*H*3 floats. */
int bcd_hip_denoise(bcd_hip_ctx *ctx, const float *d_colors, const float *d_nsamples,
                    const float *d_histograms, const float *d_covariances,
                    int W, int H, int D, int nb_scales, const bcd_hip_params *prm, float *d_out);

/* The same call in two halves (round 6): _begin hands the frame to a worker thread of the context and returns at once, _wait returns when the frame is
 * complete (d_out valid, every stream of the context synchronised) with the status bcd_hip_denoise would have returned.  One frame per context at a
 * time; inputs, output and the context must stay untouched until _wait.  For callers with independent frames (a sequence, AOV passes): two contexts
 * with a frame in flight each keep the chip busier than one blocking call after the other -- the results are those of the blocking call. */
int bcd_hip_denoise_begin(bcd_hip_ctx *ctx, const float *d_colors, const float *d_nsamples,
                          const float *d_histograms, const float *d_covariances,
                          int W, int H, int D, int nb_scales, const bcd_hip_params *prm, float *d_out);
int bcd_hip_denoise_wait(bcd_hip_ctx *ctx);

/* row-block variant for multi-GPU tiling: the images are a horizontal band of a larger frame;
 * only main pixels on local lines [main_row_begin, main_row_end) are processed, and instead of the
 * finalised colours the raw accumulators are returned (d_sum W*H*3 floats, d_count W*H int32), so
 * that neighbouring bands can exchange and add their halo lines before bcd_hip_finalize(). */
int bcd_hip_denoise_band(bcd_hip_ctx *ctx, const float *d_colors, const float *d_nsamples,
                         const float *d_histograms, const float *d_covariances,
                         int W, int H, int D, int main_row_begin, int main_row_end,
                         const bcd_hip_params *prm, uint32_t order_seed, float *d_sum, int32_t *d_count);

/* several bands at once -- the per-scale bands of one rank in the multi-GPU path -- run concurrently (job i on its own
 * stream / host thread / workspace, stats slot i), like the scales of bcd_hip_denoise() */
typedef struct bcd_hip_band_job {
    const float *d_colors, *d_nsamples, *d_histograms, *d_covariances;
    int32_t W, H, D, main_row_begin, main_row_end;
    uint32_t order_seed;
    float *d_sum;
    int32_t *d_count;
} bcd_hip_band_job;
int bcd_hip_denoise_bands(bcd_hip_ctx *ctx, const bcd_hip_band_job *jobs, int njobs, const bcd_hip_params *prm);

/* ---- one frame over several GPUs of a node (what bcd::Denoiser::setDevices / bcd_cli --devices use) ------------------
 * Row-band partition: rank r (one host thread + device devices[r]) owns a band of main pixels aligned to 2^(S-1) lines, holds
 * (b + w) halo lines of input per scale, rebuilds the pyramid for its band, and exchanges with its two neighbours only: |S| and
 * marking states of b boundary lines between marking batches (the visiting order is the whole frame's, so the result is the
 * single-GPU frame for every -m / -r setting), (b + w) accumulator halo lines, and 2 + 1 output lines for the merges.
 * Transport: RCCL point-to-point over xGMI (ncclSend / ncclRecv, one communicator per scale) when the ranks are distinct devices;
 * several ranks on ONE device (tests, debugging) exchange through device copies.  Host buffers in, host buffer out. */
typedef struct bcd_hip_multi bcd_hip_multi;
typedef struct bcd_hip_multi_stats {
    int32_t n_ranks;
    int32_t transport;          /* 1 = RCCL, 0 = in-process copies (ranks share a device) */
    int64_t frames;
    int32_t marking_rounds[8];  /* exchange + marking batches per scale of the last frame */
    float   compute_ms;         /* bcd_hip_multi_denoise_host: last frame between "every rank has its inputs" and "every rank has its band" */
} bcd_hip_multi_stats;
int  bcd_hip_multi_create(bcd_hip_multi **m, const int *devices, int n_ranks);
void bcd_hip_multi_destroy(bcd_hip_multi *m);
const char *bcd_hip_multi_last_error(const bcd_hip_multi *m);
int  bcd_hip_multi_get_stats(const bcd_hip_multi *m, bcd_hip_multi_stats *out);
/* IDenoiser::setProgressCallback for the multi-device path: every (rank, scale) reports its owned pixels when its processed set
 * is known and when its estimate is complete; calls are serialised and monotone, the last value is 1 */
int  bcd_hip_multi_set_progress_callback(bcd_hip_multi *m, bcd_hip_progress_fn fn, void *user);
int  bcd_hip_multi_set_frame_timeout(bcd_hip_multi *m, int milliseconds);
/* Failures.  A call that returns an error leaves the handle usable for the next frame: barriers, gates and the error state are
 * reset on entry.  On the RCCL transport the first failure of a frame aborts the local communicators (ncclCommAbort -- peers blocked
 * in a send / receive / all-reduce are released instead of waiting for ever), and a frame that has not finished after
 * BCD_HIP_MULTI_TIMEOUT_S seconds (default 600; 0 = never; bcd_hip_multi_set_frame_timeout sets it in milliseconds) is failed the
 * same way by the handle's watchdog thread, which is what ends a frame whose peer process died.  bcd_hip_multi_create handles rebuild their communicators on the next call; a bcd_hip_multi_create_rank handle (one process
 * per GPU) needs fresh unique ids from all processes: bcd_hip_multi_rank_renew_ids, or destroy and create it again. */
/* Communication trace of the last frame (debugging / tests): per rank, in the order the rank ENQUEUED them, four values per
 * operation: channel (scale, or nb_scales for the merges), kind (0 = neighbour exchange, 1 = all-reduce), bytes exchanged with the
 * rank above, bytes exchanged with the rank below.  All ranks must show the same (channel, kind) sequence and neighbours the same
 * sizes -- the conditions under which the RCCL transport cannot block.  get returns the number of values (4 per operation). */
int  bcd_hip_multi_set_comm_trace(bcd_hip_multi *m, int enabled);
int  bcd_hip_multi_get_comm_trace(bcd_hip_multi *m, int rank, int64_t *out, int capacity);
int  bcd_hip_multi_denoise_host(bcd_hip_multi *m, const float *h_colors, const float *h_nsamples, const float *h_histograms,
                                const float *h_covariances, int W, int H, int D, int nb_scales, const bcd_hip_params *prm, float *h_out);
/* The same partition with ONE PROCESS PER GPU (MPI-style launchers; what bench.py --gpus N uses): every process creates the handle
 * of its own rank from unique ids all processes share (rank 0 calls bcd_hip_multi_unique_id once per channel -- nb_scales + 1 of
 * them -- and distributes the bytes by whatever means the launcher offers), configures the frame, uploads the lines
 * [first_input_line, +nb_input_lines) of the four inputs once, and calls bcd_hip_multi_rank_step per frame: inputs and result of
 * the band stay in HBM.  bcd_hip_multi_rank_download copies the owned lines [first_owned_line, +nb_owned_lines) of the result. */
#define BCD_HIP_MULTI_ID_BYTES 128
int  bcd_hip_multi_unique_id(char *out /* BCD_HIP_MULTI_ID_BYTES */);
/* which RCCL library this build talks to: "rccl version <code> from <path of the shared object ncclCommInitRank resolved to>" (a process that also
 * maps a second copy -- an ML framework's bundled one -- can tell which of the two serves the band driver; BCD_HIP_MULTI_VERBOSE=1 prints the same at communicator creation) */
int  bcd_hip_multi_rccl_info(char *out, int capacity);
int  bcd_hip_multi_create_rank(bcd_hip_multi **m, int rank, int n_ranks, int device, const char *ids, int n_ids);
int  bcd_hip_multi_rank_configure(bcd_hip_multi *m, int W, int H, int D, int nb_scales, const bcd_hip_params *prm, int *first_input_line,
                                  int *nb_input_lines, int *first_owned_line, int *nb_owned_lines);
int  bcd_hip_multi_rank_upload(bcd_hip_multi *m, const float *h_colors, const float *h_nsamples, const float *h_histograms, const float *h_covariances);
int  bcd_hip_multi_rank_step(bcd_hip_multi *m);
int  bcd_hip_multi_rank_download(bcd_hip_multi *m, float *h_out_owned);
/* After a failure (or a timeout) the communicators of a one-rank handle are gone and their unique ids are consumed: instead of
 * destroying the handle, every process may hand in nb_scales + 1 FRESH ids (shared like the first set) and go on with the next frame;
 * contexts, streams and the resident band stay. */
int  bcd_hip_multi_rank_renew_ids(bcd_hip_multi *m, const char *ids, int n_ids);
/* Loopback (tests on a one-GPU box; a handle made by bcd_hip_multi_create_rank(rank 0 of 1) with ids): the rank is its own neighbour on
 * both sides, so that a frame enqueues every exchange and all-reduce of the band protocol on real RCCL communicators (ncclCommInitRank
 * with n = 1, grouped self send / recv) in the order and with the sizes a band inside a larger world would use; received data goes to
 * scratch and the result is the single-GPU frame. */
int  bcd_hip_multi_set_loopback(bcd_hip_multi *m, int enabled);
/* One device, real RCCL, through the driver's own transport code: communicators of a world of one from real unique ids, a grouped
 * self send / recv of two halo_bytes buffers and the int64 all-reduce on two channels (data checked), a simulated failure
 * (ncclCommAbort; the aborted communicators refuse further use; consumed ids cannot rebuild them), renewal from fresh ids, a second
 * exchange.  0 = all of it worked; `report` gets one line either way. */
int  bcd_hip_multi_selftest_transport(int device, long long halo_bytes, char *report, int report_capacity);

/* ---- whole path, host buffers (what bcd::Denoiser / bcd_cli call): H2D + denoise + D2H -------- */
int bcd_hip_denoise_host(bcd_hip_ctx *ctx, const float *h_colors, const float *h_nsamples,
                         const float *h_histograms, const float *h_covariances,
                         int W, int H, int D, int nb_scales, const bcd_hip_params *prm, float *h_out);
/* the same with the steps either side of the path kept on the device (one upload, one download): the spike prefilter of
 * bcd_cli -p 1 (SpikeRemovalFilter::filter, src/cli/main.cpp:428-441) on the uploaded copies -- the host images are NOT modified --
 * and the clean-up of the result (checkAndPutToZeroNegativeInfNaNValues, :389-420).  The device copies stay in the context. */
typedef struct bcd_hip_host_options {
    float   spike_factor;     /* > 0: prefilter with this standard-deviation factor (--p-factor); <= 0: off */
    int32_t zero_bad_values;  /* != 0: negative / infinite / NaN output values become 0 */
} bcd_hip_host_options;
int bcd_hip_denoise_host_ex(bcd_hip_ctx *ctx, const float *h_colors, const float *h_nsamples,
                            const float *h_histograms, const float *h_covariances,
                            int W, int H, int D, int nb_scales, const bcd_hip_params *prm, const bcd_hip_host_options *opt, float *h_out);

/* The histogram image of the last bcd_hip_denoise_host(_ex) call: its size, and the bytes that crossed the link.  On frames of >= 256 lines the
 * image travels without its zeros -- host threads pack every piece into one bit per value ("is not +0.0f", a test on the bit pattern: lossless)
 * plus the remaining values while the previous piece travels, a kernel rebuilds the fp32 image in HBM; an image with more than 60 % of
 * non-zero values is copied as it is.  BCD_HIP_SPARSE_UPLOAD=0 turns it off, BCD_HIP_UPLOAD_THREADS=<n> sets the packing threads (default:
 * half the host's hardware threads, at most 16). */
int bcd_hip_last_upload_bytes(const bcd_hip_ctx *ctx, int64_t *hist_bytes, int64_t *hist_bytes_sent);
/* host-side self-test of the packer (no device needed): packs the 32 values at in32 with the form this process uses -- returned: 0 scalar,
 * 2 AVX2, 5 AVX-512 (BCD_HIP_UPLOAD_SIMD=scalar|avx2|avx512 forces one the host has) -- into out64 (>= 64 values of room), sets the mask bits
 * and the number of values kept */
int bcd_hip_selftest_pack32(const uint32_t *in32, uint32_t *out64, uint32_t *bits, int *count);

/* ---- stages (device pointers) -- exposed for parity tests and multi-GPU composition ------------ */
/* The head of a scale's chain in two launches (the band driver's form of what bcd_hip_denoise does per scale): the per-pixel covariances
 * (bcd_hip_pixel_cov) with the accumulators d_sum (3 floats per pixel) / d_count cleared in the same pass, and every counter, flag and work queue
 * the following stage calls on this context start from (similarity masks, marking steps, the estimate) cleared in one launch instead of a fill each */
int bcd_hip_scale_begin(bcd_hip_ctx *ctx, const float *d_covariances, const float *d_nsamples, int W, int H, float *d_pixcov, float *d_sum, int32_t *d_count);
/* Denoiser::computePixelCovFromSampleCov   src/core/Denoiser.cpp:357-373 */
int bcd_hip_pixel_cov(bcd_hip_ctx *ctx, const float *d_cov, const float *d_nsamples, int W, int H, float *d_out);
/* DenoisingUnit::selectSimilarPatches for every main pixel   src/core/DenoisingUnit.cpp:196-219,336-386
 * d_mask: W*H*words uint32, bit (dl+b)*(2b+1)+(dc+b); d_count: W*H int32 = |S|.  words = ceil((2b+1)^2/32) */
int bcd_hip_similarity_masks(bcd_hip_ctx *ctx, const float *d_histograms, const float *d_nsamples,
                             int W, int H, int D, int patch_radius, int search_radius, float threshold,
                             uint32_t *d_mask, int32_t *d_count);
/* The same in three steps, for callers that have a host round trip of their own coming (the multi-GPU driver: its first marking
 * batch): _deferred enqueues the production kernels and copies their validity flags to the host WITHOUT waiting; after the caller's
 * next synchronisation of the context's stream, _verdict says whether the masks have to be recomputed with the exact kernels
 * (inputs outside the guarded range, borderline list overflowed) -- then call _exact. */
int bcd_hip_similarity_masks_deferred(bcd_hip_ctx *ctx, const float *d_histograms, const float *d_nsamples,
                                      int W, int H, int D, int patch_radius, int search_radius, float threshold,
                                      uint32_t *d_mask, int32_t *d_count);
int bcd_hip_similarity_masks_verdict(bcd_hip_ctx *ctx, int *redo);
int bcd_hip_similarity_masks_exact(bcd_hip_ctx *ctx, const float *d_histograms, const float *d_nsamples,
                                   int W, int H, int D, int patch_radius, int search_radius, float threshold,
                                   uint32_t *d_mask, int32_t *d_count);
/* raw patch distances of one main pixel to its window (debug / parity): (2b+1)^2 floats, +inf outside */
int bcd_hip_window_distances(bcd_hip_ctx *ctx, const float *d_histograms, const float *d_nsamples,
                             int W, int H, int D, int patch_radius, int search_radius,
                             int line, int col, float *h_out);
/* the marking strategy (DenoisingUnit.cpp:164-173,690) as a parallel fixed point.
 * d_state: W*H uint8: 0 = not a main pixel / outside band, 1 = processed, 2 = skipped. */
int bcd_hip_active_set(bcd_hip_ctx *ctx, const uint32_t *d_mask, const int32_t *d_count,
                       int W, int H, int patch_radius, int search_radius,
                       int main_row_begin, int main_row_end,
                       float skip_probability, int random_order, uint32_t seed,
                       uint8_t *d_state, int32_t *rounds);
/* the two halves of bcd_hip_active_set, for the multi-GPU band path: between two steps neighbouring bands exchange the states
 * of their boundary lines.  Only lines [main_row_begin, main_row_end) are initialised / decided; row_offset = line of the full
 * frame under local line 0 (keys and hashes are functions of the global pixel index).  bcd_hip_active_step keeps the dependency
 * lists it extracts on the first call after bcd_hip_active_init (same masks and counts until the next init); first_pass is a hint ("everything still undecided")
 * that implementations may ignore. */
int bcd_hip_active_init(bcd_hip_ctx *ctx, const int32_t *d_count, int W, int H, int patch_radius, int main_row_begin, int main_row_end,
                        float skip_probability, uint32_t seed, int row_offset, uint8_t *d_state);
int bcd_hip_active_step(bcd_hip_ctx *ctx, const uint32_t *d_mask, const int32_t *d_count, int W, int H, int patch_radius, int search_radius,
                        int main_row_begin, int main_row_end, int random_order, uint32_t seed, int row_offset, int first_pass,
                        uint8_t *d_state, int32_t *undecided);
/* bcd_hip_active_step in two halves, for a caller whose own reduction follows in stream order (the multi-GPU driver's all-reduce; round 6): _enqueue
 * launches the batch without waiting and, if d_total is not null, leaves on the DEVICE the rank's contribution to the all-reduced count of undecided
 * pixels -- the count after the batch, + 2^40 when with_verdict != 0 and the masks of the last bcd_hip_similarity_masks_deferred on this context are
 * not valid (the test bcd_hip_similarity_masks_verdict makes on the host); after the caller's synchronisation of the context's stream _collect returns
 * the local count and the number of launches the batch needed. */
int bcd_hip_active_step_enqueue(bcd_hip_ctx *ctx, const uint32_t *d_mask, const int32_t *d_count, int W, int H, int patch_radius, int search_radius,
                                int main_row_begin, int main_row_end, int random_order, uint32_t seed, int row_offset,
                                uint8_t *d_state, int64_t *d_total, int with_verdict);
int bcd_hip_active_step_collect(bcd_hip_ctx *ctx, int32_t *undecided, int32_t *launches);
/* denoiseSelectedPatches / denoiseOnlyMainPatch + aggregateOutputPatches for every processed pixel
 * (DenoisingUnit.cpp:388-481,672-693); d_sum / d_count are accumulated into (zero them first). */
int bcd_hip_bayes_accumulate(bcd_hip_ctx *ctx, const float *d_colors, const float *d_pixel_cov,
                             const uint32_t *d_mask, const int32_t *d_nsim, const uint8_t *d_state,
                             int W, int H, int patch_radius, int search_radius, float min_eigen_value,
                             float *d_sum, int32_t *d_count);
/* The same for the processed pixels of lines [row_begin, row_end) only (a row band's owned lines), optionally SPECULATIVE (round 6; patch radius 1):
 * d_skip_if points at a device word that the work already enqueued on the context's stream leaves at zero when this estimate is wanted (the band
 * driver: the all-reduced count of undecided pixels of the marking batch just enqueued), h_skip_if at the host copy of that word, copied on the same
 * stream before this call.  The call enqueues the lists and the estimate kernels -- which do nothing when the word is not zero -- waits for ONE event
 * (list lengths + the word) and reports *skipped = 1 if the word was not zero: the caller continues its marking and calls again. */
int bcd_hip_bayes_accumulate_rows(bcd_hip_ctx *ctx, const float *d_colors, const float *d_pixel_cov,
                                  const uint32_t *d_mask, const int32_t *d_nsim, const uint8_t *d_state,
                                  int W, int H, int patch_radius, int search_radius, float min_eigen_value,
                                  float *d_sum, int32_t *d_count, int row_begin, int row_end,
                                  const int64_t *d_skip_if, const int64_t *h_skip_if, int *skipped);
/* Denoiser::finalAggregation   src/core/Denoiser.cpp:458-469 */
int bcd_hip_finalize(bcd_hip_ctx *ctx, const float *d_sum, const int32_t *d_count, int64_t npix, float *d_out);
/* the same on `rows` lines of a row band (multi-GPU path), with the accumulator halos received from the neighbouring bands added
 * to the first / last `halo` lines first (nullptr pair at a frame border): one launch instead of four adds and a finalisation */
int bcd_hip_finalize_band(bcd_hip_ctx *ctx, const float *d_sum, const int32_t *d_count, int W, int rows, int halo,
                          const float *d_up_sum, const int32_t *d_up_count, const float *d_down_sum, const int32_t *d_down_count,
                          float *d_out);
/* MultiscaleDenoiser pyramid + merge   src/core/MultiscaleDenoiser.cpp:243-334,453-548 */
int bcd_hip_downscale_sum(bcd_hip_ctx *ctx, const float *d_in, int W, int H, int D, float *d_out);
int bcd_hip_downscale_avg(bcd_hip_ctx *ctx, const float *d_in, int W, int H, int D, float *d_out);
int bcd_hip_downscale_cov(bcd_hip_ctx *ctx, const float *d_cov, const float *d_nsamples, int W, int H, float *d_out);
int bcd_hip_interpolate(bcd_hip_ctx *ctx, const float *d_lo, int w, int h, int D, float *d_hi, int W, int H);
/* d_hi (W x H x D) <- d_hi - up(down(d_hi)) + up(d_lo) */
int bcd_hip_merge(bcd_hip_ctx *ctx, float *d_hi, int W, int H, const float *d_lo, int D);
/* SpikeRemovalFilter::filter   src/core/SpikeRemovalFilter.cpp:18-116 (out of place) */
int bcd_hip_spike_filter(bcd_hip_ctx *ctx, const float *d_colors, const float *d_nsamples,
                         const float *d_histograms, const float *d_covariances, int W, int H, int D, float factor,
                         float *d_colors_out, float *d_nsamples_out, float *d_histograms_out, float *d_covariances_out);
/* SamplesAccumulator::addSample + getSamplesStatistics for a whole frame   src/core/SamplesAccumulator.cpp:44-141
 * (lets a GPU renderer keep the statistics in HBM).  d_samples: W*H*spp*3 floats, the spp samples of a pixel contiguous
 * and in accumulation order; d_weights: W*H*spp floats or NULL (all 1).  Outputs in DeepImage layout, hist depth 3*nb_bins. */
int bcd_hip_accumulate_samples(bcd_hip_ctx *ctx, const float *d_samples, const float *d_weights, int W, int H, int spp, int nb_bins,
                               float gamma, float max_value, float *d_nsamples, float *d_mean, float *d_cov, float *d_hist);
/* checkAndPutToZeroNegativeInfNaNValues   src/cli/main.cpp:389-420 */
int bcd_hip_zero_bad_values(bcd_hip_ctx *ctx, float *d_img, int64_t n);

/* self-test: the scale-free division used by the pair-distance kernel against the compiler's IEEE division on
 * `samples` hashed operand pairs drawn from its guarded range; *mismatches must come back 0 */
int bcd_hip_selftest_division(bcd_hip_ctx *ctx, uint32_t seed, int64_t samples, int64_t *mismatches);
/* self-test of the pair-distance kernels on given inputs: the production variant (fast division; the uniform power-of-two sample-count
 * formula when it applies) against the compiler's division with the general formula; *mismatches = entries of the T / C planes that
 * differ bitwise (0 expected unless a range / count flag was raised: *variant bits 4..); *variant & 15: 1 = fast, 2 = fast + uniform */
int bcd_hip_selftest_distance_kernels(bcd_hip_ctx *ctx, const float *d_hist, const float *d_nsamples, int W, int H, int D, int search_radius,
                                      int *variant, int64_t *mismatches);

/* Measurement (bench.py `roofline.valu`): the arithmetic the production distance kernel performs on this frame, from a counting instantiation
 * of the same kernel -- lane_bins: (pixel pair, bin) terms evaluated = the reference's own count of bins with b1 + b2 > 1 over the half-plane
 * displacements (src/core/DenoisingUnit.cpp:379-381); wave_bins: bins a wavefront issues because at least one of its 64 pairs needs them;
 * wave_groups: groups of four bins entered -- and kernel_ms: the production instantiation on the same input (HIP events, best of reps). */
int bcd_hip_selftest_bin_work(bcd_hip_ctx *ctx, const float *d_hist, const float *d_nsamples, int W, int H, int D, int search_radius, int reps,
                              int64_t *lane_bins, int64_t *wave_bins, int64_t *wave_groups, float *kernel_ms);

/* self-test of the approximate pair-distance kernel (k_pairdist_rw) on given inputs: *max_rel_dev = largest relative deviation of a
 * patch distance d(p, p + delta) computed from the approximate planes from the one computed from the exact planes, over all pairs of
 * main pixels (bound 5e-4, measured 2.4e-4; must stay below 2^-10 = 9.8e-4, BCD_APPROX_DELTA, the half-width of the band that is
 * re-evaluated exactly); *count_mismatches = pairs whose
 * integer bin counts differ (must be 0); *flags: low nibble 2 = the uniform kernel ran, 3 = the RATIO form (general sample counts: the production kernel for
 * them since round 6), the kernels' flag word above it (bit 2, value 4 << 4: the RATIO form's absolute-error check declined) */
int bcd_hip_selftest_approx_distance(bcd_hip_ctx *ctx, const float *d_hist, const float *d_nsamples, int W, int H, int D, int search_radius,
                                     float *max_rel_dev, int64_t *count_mismatches, int *flags);

/* the eigensolver of the Bayesian steps on its own (Eigen::SelfAdjointEigenSolver of DenoisingUnit.cpp:589,617 for 27 x 27 matrices):
 * d_A = n symmetric matrices, 28 x 28 floats each, row-major, row / column 27 zero; d_eig[n][28] = eigenvalues (unordered, entry 27 = 0),
 * d_V[n][28][28] = eigenvectors in columns, same order (rows 0..26 written); *ms = kernel time (may be NULL).  Parity / timing aid. */
int bcd_hip_eig27_batch(bcd_hip_ctx *ctx, const float *d_A, int n, float *d_eig, float *d_V, float *ms);

/* ---- host utilities (no device work) ----------------------------------------------------------- */
/* the visiting order implied by (random_order, seed): main-pixel linear indices line*W+col in
 * visiting order, written to h_order[(W-2w)*(H-2w)].  random_order == 0 is the reference's
 * single-thread scanline order (Denoiser.cpp:136-146). */
int bcd_hip_visit_order(int W, int H, int patch_radius, int random_order, uint32_t seed, int32_t *h_order);
/* order 2 (strips): the `seed` argument of bcd_hip_visit_order / bcd_hip_active_set carries the frame geometry instead of a seed */
uint32_t bcd_hip_strip_order_seed(int W, int H, int patch_radius, int search_radius);
/* seed used for scale s of a multiscale run started with seed0 */
uint32_t bcd_hip_scale_seed(uint32_t seed0, int scale);

#ifdef __cplusplus
}
#endif
#endif
