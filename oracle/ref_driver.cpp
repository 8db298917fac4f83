// ref_driver.cpp -- C entry points onto the REFERENCE's own compiled translation units
// (TEST INFRASTRUCTURE).  This file contains no algorithm: it only marshals raw float buffers
// into the reference's DeepImage objects and calls the reference functions named below.
// Built only in the authoring container (needs /root/reference); output oracle/_ref/libbcd_ref.so.
#include "DeepImage.h"
#include "MultiscaleDenoiser.h"
#include "SamplesAccumulator.h"
#include "SpikeRemovalFilter.h"
#include "Utils.h"

#include <cstring>
#include <memory>

#define API extern "C" __attribute__((visibility("default")))

using namespace bcd;

static Deepimf wrap(const float* p, int W, int H, int D)
{
	Deepimf im(W, H, D);
	im.copyDataFrom(p);
	return im;
}

// MultiscaleDenoiser::downscaleSum (src/core/MultiscaleDenoiser.cpp:243-268)
API void bcdref_downscale_sum(const float* in, int W, int H, int D, float* out)
{
	std::unique_ptr<Deepimf> r = MultiscaleDenoiser::downscaleSum(wrap(in, W, H, D));
	r->copyDataTo(out);
}
// MultiscaleDenoiser::downscaleAverage (:270-295)
API void bcdref_downscale_avg(const float* in, int W, int H, int D, float* out)
{
	std::unique_ptr<Deepimf> r = MultiscaleDenoiser::downscaleAverage(wrap(in, W, H, D));
	r->copyDataTo(out);
}
// MultiscaleDenoiser::downscaleSampleCovarianceSum (:297-334)
API void bcdref_downscale_cov(const float* cov, const float* ns, int W, int H, int D, float* out)
{
	std::unique_ptr<Deepimf> r = MultiscaleDenoiser::downscaleSampleCovarianceSum(wrap(cov, W, H, D), wrap(ns, W, H, 1));
	r->copyDataTo(out);
}
// MultiscaleDenoiser::interpolate (:473-512)
API void bcdref_interpolate(const float* lo, int w, int h, int D, float* hi, int W, int H)
{
	Deepimf out(W, H, D);
	MultiscaleDenoiser::interpolate(out, wrap(lo, w, h, D));
	out.copyDataTo(hi);
}
// MultiscaleDenoiser::mergeOutputs (:453-466), in place on hi
API void bcdref_merge(float* hi, int W, int H, const float* lo, int D)
{
	Deepimf hiIm = wrap(hi, W, H, D), tmpHi(W, H, D), tmpLo(W / 2, H / 2, D);
	MultiscaleDenoiser::mergeOutputs(hiIm, tmpHi, tmpLo, wrap(lo, W / 2, H / 2, D), hiIm);
	hiIm.copyDataTo(hi);
}
// SpikeRemovalFilter::filter (src/core/SpikeRemovalFilter.cpp:18-75), in place
API void bcdref_spike_filter(float* col, float* ns, float* hist, float* cov, int W, int H, int D, float factor)
{
	Deepimf c = wrap(col, W, H, 3), n = wrap(ns, W, H, 1), h = wrap(hist, W, H, D), v = wrap(cov, W, H, 6);
	SpikeRemovalFilter::filter(c, n, h, v, factor);
	c.copyDataTo(col); n.copyDataTo(ns); h.copyDataTo(hist); v.copyDataTo(cov);
}
// SamplesAccumulator (src/core/SamplesAccumulator.cpp:29-153); samples: n x (line,col,r,g,b,weight)
API void bcdref_accumulate(const float* s, long long n, int W, int H, int nbins, float gamma, float maxval,
		float* ns, float* mean, float* cov, float* hist)
{
	HistogramParameters hp;
	hp.m_nbOfBins = nbins; hp.m_gamma = gamma; hp.m_maxValue = maxval;
	SamplesAccumulator acc(W, H, hp);
	for(long long i = 0; i < n; ++i, s += 6)
		acc.addSample(int(s[0]), int(s[1]), s[2], s[3], s[4], s[5]);
	SamplesStatisticsImages st = acc.getSamplesStatistics();
	st.m_nbOfSamplesImage.copyDataTo(ns);
	st.m_meanImage.copyDataTo(mean);
	st.m_covarImage.copyDataTo(cov);
	st.m_histoImage.copyDataTo(hist);
}
// Utils::mergeHistogramAndNbOfSamples / separateNbOfSamplesFromHistogram (src/core/Utils.cpp:21-77)
API void bcdref_merge_hist_ns(const float* hist, const float* ns, int W, int H, int D, float* out)
{
	Deepimf r = Utils::mergeHistogramAndNbOfSamples(wrap(hist, W, H, D), wrap(ns, W, H, 1));
	r.copyDataTo(out);
}
API void bcdref_split_hist_ns(const float* in, int W, int H, int Dp1, float* hist, float* ns)
{
	Deepimf h, n;
	Utils::separateNbOfSamplesFromHistogram(h, n, wrap(in, W, H, Dp1));
	h.copyDataTo(hist); n.copyDataTo(ns);
}
