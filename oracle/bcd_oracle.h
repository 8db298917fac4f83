/*
 * bcd_oracle.h -- CPU ORACLE for the BCD hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference algorithm (superboubek/bcd v1.1) used as the
 * checker for the HIP path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product (bcd_amd/)
 * never links, imports or calls it.
 *
 * PARITY PIN STATUS
 *   - pyramid reducers, merge/interpolate, spike filter, samples accumulator:
 *     PINNED bit-exactly against the compiled reference translation units
 *     (oracle/_ref, built by oracle/Makefile from /root/reference sources).
 *   - Denoiser / DenoisingUnit core (distances, selection, Bayesian steps,
 *     aggregation): PARITY UNPINNED.  The reference core needs Eigen
 *     (include/bcd/core/DenoisingUnit.h:21-22), an un-vendored, un-pinned
 *     submodule (.gitmodules:10-12, libigl/eigen) absent from this image, so it
 *     is unbuildable here and the reference ships no tests or golden vectors.
 *     The restatement follows the reference line by line (citations on each
 *     function); Eigen's SelfAdjointEigenSolver is restated as the published
 *     algorithm it implements (Householder tridiagonalisation + implicit
 *     symmetric QL/QR with shifts, eigenvalues ascending).
 *
 * All images are interleaved row-major "DeepImage" buffers:
 *     index = (line * W + col) * D + d       (include/bcd/core/DeepImage.hpp:385-396)
 * All arithmetic is IEEE fp32, no FMA contraction (compile with -ffp-contract=off).
 */
#ifndef BCD_ORACLE_H
#define BCD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct BcdoParams {
    float hist_dist_threshold; /* m_histogramDistanceThreshold (IDenoiser.h:36)  default 1   */
    int   patch_radius;        /* m_patchRadius               (IDenoiser.h:37)  default 1   */
    int   search_radius;       /* m_searchWindowRadius        (IDenoiser.h:38)  default 6   */
    float min_eigen_value;     /* m_minEigenValue             (IDenoiser.h:39)  default 1e-8 */
    float skip_probability;    /* m_markedPixelsSkippingProbability (:41); only 0 and 1 are
                                  deterministic in the reference (rand() otherwise)          */
    int   nb_threads;          /* OpenMP threads for the m=0 path (order-free); m=1 is serial */
    uint32_t skip_seed;        /* 0 < skip_probability < 1 only: the reference draws unseeded rand() (DenoisingUnit.cpp:168);
                                  the build replaces it by a per-pixel hash of (pixel index, seed) -- mirrored here so that
                                  the marking logic can be checked; scale s of a multiscale run uses skip_seed + s       */
} BcdoParams;

/* per-pixel diagnostics (all optional, may be NULL); W*H entries each */
typedef struct BcdoDiag {
    uint8_t *processed;   /* 1 if the main pixel was not skipped                              */
    uint8_t *fallback;    /* 1 if processed through denoiseOnlyMainPatch (|S| < 3P+1)         */
    int32_t *nb_similar;  /* |S| for processed pixels, -1 otherwise                           */
} BcdoDiag;

/* ---- a6: pre-pass  (src/core/Denoiser.cpp:357-373) ---- */
void bcdo_pixel_cov_from_sample_cov(const float *cov, const float *nsamp, int W, int H, float *out);

/* ---- a10: distances (src/core/DenoisingUnit.cpp:336-386) ---- */
float bcdo_patch_distance(const float *hist, const float *nsamp, int W, int H, int D, int w,
                          int pl, int pc, int ql, int qc);
/* distances of main pixel (pl,pc) to its clipped window, row-major, written to out[(2b+1)^2]
 * at slot (dl+b)*(2b+1)+(dc+b); slots outside the clipped window are set to +inf. */
void bcdo_window_distances(const float *hist, const float *nsamp, int W, int H, int D, int w, int b,
                           int pl, int pc, float *out);
/* similarity bitmask for every pixel: bit k=(dl+b)*(2b+1)+(dc+b) of mask[p*words+k/32];
 * non-main (border) pixels get an all-zero mask.  Returns words per pixel. */
int bcdo_similarity_masks(const float *hist, const float *nsamp, int W, int H, int D, int w, int b,
                          float tau, uint32_t *mask, int32_t *count, int nb_threads);

/* ---- Eigen::SelfAdjointEigenSolver restatement (lower triangle read, ascending) ---- */
void bcdo_sym_eig(int n, const float *A, float *evals, float *evecs /* column j = vector j, row-major [r*n+j] */);

/* ---- a9..a16: monoscale denoiser (src/core/Denoiser.cpp:84-212, DenoisingUnit.cpp:157-693)
 * order: visiting order of main pixels as linear indices line*W+col (NULL = scanline,
 * the reference's 1-thread -r 0 order, Denoiser.cpp:136-146).  Returns 0 on success.
 * An ordered / marking visit is a one-thread loop; with prm->nb_threads > 1 the SAME visit runs in three phases -- the similar sets of
 * all pixels in parallel (selectSimilarPatches does not depend on the visit), the decisions of denoisePatchAndSimilarPatches one pixel
 * after the other (processed unless marked; a full estimate marks its similar set), the processed pixels' estimates in parallel with the
 * loop's per-pixel code -- equal to the loop up to the summation order of the per-thread accumulators (full-size frames). ---- */
int bcdo_denoise_mono(const float *colors, const float *nsamp, const float *hist, const float *cov,
                      int W, int H, int D, const BcdoParams *prm,
                      const int32_t *order, int64_t n_order,
                      float *out, const BcdoDiag *diag);

/* ---- one-patch trace (SURVEY.md 8c fixture F3): the intermediates of denoiseSelectedPatches (DenoisingUnit.cpp:388-453) for one
 * main pixel.  K = 3 (2w+1)^2; every pointer is optional (NULL = not wanted).  Matrices K x K row-major, patch arrays |S| x K in
 * window order (:196-219), vectors pixel-major RGB (:483-498). ---- */
typedef struct BcdoPatchTrace {
    int32_t *members;          /* |S| linear indices line*W+col, capacity (2b+1)^2                               */
    float *noise;              /* (2w+1)^2 x 6   computeNoiseCovPatchesMean (:400-419), order xx,yy,zz,yz,xz,xy  */
    float *x;                  /* |S| x K        noisy patches (:483-498)                                        */
    float *mean1;              /* K              empiricalMean of x (:500-509); the whole estimate when |S| < K+1 */
    float *cov1;               /* K x K          empiricalCovarianceMatrix (:522-536)                            */
    float *cov1_minus_noise;   /* K x K          substractCovMatPatchFromMatrix (:558-576)                       */
    float *clamped;            /* K x K          clampNegativeEigenValues (:606-630)                             */
    float *clamped_plus_noise; /* K x K          addCovMatPatchToMatrix (:538-556)                               */
    float *inverse1;           /* K x K          inverseSymmetricMatrix (:578-604)                               */
    float *step1;              /* |S| x K        finalDenoisingMatrixMultiplication (:656-670)                   */
    float *mean2;              /* K              mean of the Step-1 estimates (:440)                             */
    float *cov2;               /* K x K          their covariance (:441-442)                                     */
    float *inverse2;           /* K x K          inverse of cov2 + noise (:445-446)                              */
    float *step2;              /* |S| x K        final estimates (:449-450)                                      */
} BcdoPatchTrace;
int bcdo_patch_trace(const float *colors, const float *nsamp, const float *hist, const float *cov, int W, int H, int D,
                     const BcdoParams *prm, int pl, int pc, BcdoPatchTrace *trace);

/* band form for the multi-GPU tests: main pixels on lines [row_begin,row_end), raw accumulators (zeroed here) */
int bcdo_accumulate_band(const float *colors, const float *nsamp, const float *hist, const float *cov,
                         int W, int H, int D, const BcdoParams *prm, int row_begin, int row_end,
                         const int32_t *order, int64_t n_order, float *sum, int32_t *cnt);
int bcdo_accumulate_pixels(const float *colors, const float *nsamp, const float *hist, const float *cov,
                           int W, int H, int D, const BcdoParams *prm, const int32_t *pixels, int64_t n, float *sum, int32_t *cnt);
void bcdo_finalize(const float *sum, const int32_t *cnt, int64_t npix, float *out);

/* reference-style racy OpenMP m=1 run (shared mark image, strip order, dynamic schedule;
 * Denoiser.cpp:164-205,375-414) -- for cpu_baseline timing only, NOT reproducible. */
int bcdo_denoise_mono_omp_racy(const float *colors, const float *nsamp, const float *hist, const float *cov,
                               int W, int H, int D, const BcdoParams *prm, float *out);

/* ---- a18/a19: pyramid + merge (src/core/MultiscaleDenoiser.cpp:243-334,453-548) ---- */
void bcdo_downscale_sum(const float *in, int W, int H, int D, float *out);
void bcdo_downscale_avg(const float *in, int W, int H, int D, float *out);
void bcdo_downscale_cov(const float *cov, const float *nsamp, int W, int H, int D, float *out);
void bcdo_interpolate(const float *lo, int w, int h, int D, float *hi, int W, int H);
void bcdo_merge(float *hi, int W, int H, const float *lo, int D);

/* ---- a17: multiscale (src/core/MultiscaleDenoiser.cpp:31-136).
 * orders[s] / n_orders[s] per scale (NULL => scanline at every scale). ---- */
int bcdo_denoise_multiscale(const float *colors, const float *nsamp, const float *hist, const float *cov,
                            int W, int H, int D, int nb_scales, const BcdoParams *prm,
                            const int32_t *const *orders, const int64_t *n_orders,
                            float *out, int racy_omp);

/* ---- a20: spike removal prefilter (src/core/SpikeRemovalFilter.cpp:18-116), in place,
 * float-abs semantics (MSVC / -include math.h). ---- */
void bcdo_spike_filter(float *colors, float *nsamp, float *hist, float *cov, int W, int H, int D, float factor);

/* ---- SamplesAccumulator (src/core/SamplesAccumulator.cpp:44-141) ----
 * samples: n x 6 floats (line, col, r, g, b, weight); outputs must be zero-initialised by the caller?
 * no: the function zeroes them.  hist depth = 3*nbins. */
void bcdo_accumulate(const float *samples, int64_t n, int W, int H, int nbins, float gamma, float maxval,
                     float *nsamp, float *mean, float *cov, float *hist);

/* ---- CLI clean-up (src/cli/main.cpp:389-420) ---- */
void bcdo_zero_bad_values(float *img, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
