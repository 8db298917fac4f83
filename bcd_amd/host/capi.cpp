// capi.cpp -- plain-C exports of libbcdcore for the Python plumbing (bench.py, tests): synthetic scenes,
// the SamplesAccumulator, Utils packing, and the bcd::Denoiser / bcd::MultiscaleDenoiser classes themselves.
#include "Denoiser.h"
#include "MultiscaleDenoiser.h"
#include "SamplesAccumulator.h"
#include "SpikeRemovalFilter.h"
#include "SyntheticScene.h"
#include "Utils.h"

#include <cstdio>
#include <memory>

using namespace bcd;

extern "C" {

// W x i_nbOfLines band of a W x H synthetic frame; outputs: ns [n], mean [n*3], cov [n*6], hist [n*60]
int bcdcore_synthetic_scene_ex(int W, int H, int spp, unsigned seed, float sigma, float spikeProbability, int pattern, int firstLine, int nbOfLines,
		float* o_ns, float* o_mean, float* o_cov, float* o_hist);

int bcdcore_synthetic_scene(int W, int H, int spp, unsigned seed, float sigma, float spikeProbability, int firstLine, int nbOfLines,
		float* o_ns, float* o_mean, float* o_cov, float* o_hist)
{
	return bcdcore_synthetic_scene_ex(W, H, spp, seed, sigma, spikeProbability, 0, firstLine, nbOfLines, o_ns, o_mean, o_cov, o_hist);
}

int bcdcore_synthetic_scene_ex(int W, int H, int spp, unsigned seed, float sigma, float spikeProbability, int pattern, int firstLine, int nbOfLines,
		float* o_ns, float* o_mean, float* o_cov, float* o_hist)
{
	if(W <= 0 || H <= 0 || spp <= 0 || firstLine < 0 || firstLine + nbOfLines > H) return -1;
	SyntheticSceneParameters p;
	p.m_width = W; p.m_height = H; p.m_samplesPerPixel = spp; p.m_seed = seed; p.m_noiseSigma = sigma; p.m_spikeProbability = spikeProbability;
	p.m_pattern = pattern;
	SamplesStatisticsImages st = generateSyntheticScene(p, firstLine, nbOfLines);
	st.m_nbOfSamplesImage.copyDataTo(o_ns);
	st.m_meanImage.copyDataTo(o_mean);
	st.m_covarImage.copyDataTo(o_cov);
	st.m_histoImage.copyDataTo(o_hist);
	return 0;
}

// samples: n x (line, col, r, g, b, weight)
int bcdcore_accumulate(const float* s, long long n, int W, int H, int nbins, float gamma, float maxval, float* ns, float* mean, float* cov, float* hist)
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
	return 0;
}

// the same through SamplesAccumulatorThreadSafe::addSampleThreadSafely from `threads` OpenMP threads that share the sample list
// round-robin (so that threads do meet on the same pixel)
int bcdcore_accumulate_threadsafe(const float* s, long long n, int W, int H, int nbins, float gamma, float maxval, int threads, float* ns, float* mean,
		float* cov, float* hist)
{
	HistogramParameters hp;
	hp.m_nbOfBins = nbins; hp.m_gamma = gamma; hp.m_maxValue = maxval;
	SamplesAccumulatorThreadSafe acc(W, H, hp);
#pragma omp parallel for schedule(static, 1) num_threads(threads)
	for(long long i = 0; i < n; ++i)
	{
		const float* p = s + 6 * i;
		acc.addSampleThreadSafely(int(p[0]), int(p[1]), p[2], p[3], p[4], p[5]);
	}
	SamplesStatisticsImages st = acc.getSamplesStatistics();
	st.m_nbOfSamplesImage.copyDataTo(ns);
	st.m_meanImage.copyDataTo(mean);
	st.m_covarImage.copyDataTo(cov);
	st.m_histoImage.copyDataTo(hist);
	return 0;
}

void bcdcore_release_engines() { releaseEngines(); }

// runs bcd::Denoiser (nscales == 1) or bcd::MultiscaleDenoiser through the IDenoiser interface; returns denoise()'s bool.
// Null pointers are forwarded as null images to exercise the validation path.
static int g_lastProgressValues = 0;
/// number of distinct progress values the last bcdcore_denoise reported
int bcdcore_last_progress_values() { return g_lastProgressValues; }

static int g_lastNbOfCores = 0;
/// m_nbOfCores as the last bcdcore_denoise* left it in the denoiser's parameters (Denoiser.cpp:121 writes it back)
int bcdcore_last_nb_of_cores() { return g_lastNbOfCores; }

int bcdcore_denoise_ex(const float* col, const float* ns, const float* hist, const float* cov, int W, int H, int D, int nscales,
		float tau, int w, int b, float minEig, int randomOrder, float skipProbability, unsigned seed, float* out, int histWidthOverride,
		int useCuda, const int* devices, int nbOfDevices, float prefilterFactor);

int bcdcore_denoise(const float* col, const float* ns, const float* hist, const float* cov, int W, int H, int D, int nscales,
		float tau, int w, int b, float minEig, int randomOrder, float skipProbability, unsigned seed, float* out, int histWidthOverride)
{
	return bcdcore_denoise_ex(col, ns, hist, cov, W, H, D, nscales, tau, w, b, minEig, randomOrder, skipProbability, seed, out, histWidthOverride, 1, nullptr, 0, 0.f);
}

/// + DenoiserParameters::m_useCuda, HipEngineSettings::setDevices / setSpikePrefilter
int bcdcore_denoise_ex(const float* col, const float* ns, const float* hist, const float* cov, int W, int H, int D, int nscales,
		float tau, int w, int b, float minEig, int randomOrder, float skipProbability, unsigned seed, float* out, int histWidthOverride,
		int useCuda, const int* devices, int nbOfDevices, float prefilterFactor)
{
	Deepimf cImg, nImg, hImg, vImg, oImg(W > 0 ? W : 0, H > 0 ? H : 0, 3);
	DenoiserInputs in;
	if(col) { cImg.resize(W, H, 3); cImg.copyDataFrom(col); in.m_pColors = &cImg; }
	if(ns) { nImg.resize(W, H, 1); nImg.copyDataFrom(ns); in.m_pNbOfSamples = &nImg; }
	if(hist && histWidthOverride > 0) { hImg.resize(histWidthOverride, H, D); hImg.fill(0.f); in.m_pHistograms = &hImg; } // deliberately mismatched size
	else if(hist) { hImg.resize(W, H, D); hImg.copyDataFrom(hist); in.m_pHistograms = &hImg; }
	if(cov) { vImg.resize(W, H, 6); vImg.copyDataFrom(cov); in.m_pSampleCovariances = &vImg; }
	DenoiserOutputs o;
	o.m_pDenoisedColors = &oImg;
	DenoiserParameters p;
	p.m_histogramDistanceThreshold = tau; p.m_patchRadius = w; p.m_searchWindowRadius = b; p.m_minEigenValue = minEig;
	p.m_useRandomPixelOrder = randomOrder != 0; p.m_markedPixelsSkippingProbability = skipProbability;
	p.m_useCuda = useCuda != 0;
	std::unique_ptr<IDenoiser> d;
	HipEngineSettings* pSettings = nullptr;
	if(nscales > 1) { MultiscaleDenoiser* m = new MultiscaleDenoiser(nscales); pSettings = m; d.reset(m); }
	else { Denoiser* m = new Denoiser(); pSettings = m; d.reset(m); }
	pSettings->setOrderSeed(seed);
	if(devices && nbOfDevices > 0)
		pSettings->setDevices(std::vector<int>(devices, devices + nbOfDevices));
	pSettings->setSpikePrefilter(prefilterFactor);
	IDenoiser* pDenoiser = d.get();
	pDenoiser->setInputs(in);
	pDenoiser->setOutputs(o);
	pDenoiser->setParameters(p);
	float last = -1.f;
	bool monotone = true;
	int distinct = 0;
	pDenoiser->setProgressCallback([&](float f) { if(f < last) monotone = false; if(f != last) ++distinct; last = f; });
	const bool ok = pDenoiser->denoise();
	if(ok && out) oImg.copyDataTo(out);
	g_lastProgressValues = distinct;
	g_lastNbOfCores = pDenoiser->getParameters().m_nbOfCores;
	return ok ? (monotone ? 1 : 2) : 0;
}

/// ONE MultiscaleDenoiser (or Denoiser), denoise() called twice with -r 0 and the given m_nbOfCores: the written-back core count must not
/// change what the second call does (returns 0 on failure, else 1; nbOfCoresAfter[2] = the field after each call)
int bcdcore_denoise_reuse(const float* col, const float* ns, const float* hist, const float* cov, int W, int H, int D, int nscales, int b,
		int nbOfCores, float* out1, float* out2, int* nbOfCoresAfter)
{
	Deepimf cImg(W, H, 3), nImg(W, H, 1), hImg(W, H, D), vImg(W, H, 6), oImg(W, H, 3);
	cImg.copyDataFrom(col); nImg.copyDataFrom(ns); hImg.copyDataFrom(hist); vImg.copyDataFrom(cov);
	DenoiserInputs in;
	in.m_pColors = &cImg; in.m_pNbOfSamples = &nImg; in.m_pHistograms = &hImg; in.m_pSampleCovariances = &vImg;
	DenoiserOutputs o;
	o.m_pDenoisedColors = &oImg;
	DenoiserParameters p;
	p.m_searchWindowRadius = b;
	p.m_useRandomPixelOrder = false;
	p.m_nbOfCores = nbOfCores;
	std::unique_ptr<IDenoiser> d;
	if(nscales > 1) d.reset(new MultiscaleDenoiser(nscales));
	else d.reset(new Denoiser());
	d->setInputs(in);
	d->setOutputs(o);
	d->setParameters(p);
	float* outs[2] = { out1, out2 };
	for(int i = 0; i < 2; ++i)
	{
		oImg = cImg; // (the multiscale path wants the output pre-sized like the input)
		if(!d->denoise())
			return 0;
		oImg.copyDataTo(outs[i]);
		nbOfCoresAfter[i] = d->getParameters().m_nbOfCores;
	}
	return 1;
}

int bcdcore_spike_filter(float* col, float* ns, float* hist, float* cov, int W, int H, int D, float factor)
{
	Deepimf c(W, H, 3), n(W, H, 1), h(W, H, D), v(W, H, 6);
	c.copyDataFrom(col); n.copyDataFrom(ns); h.copyDataFrom(hist); v.copyDataFrom(cov);
	SpikeRemovalFilter::filter(c, n, h, v, factor);
	c.copyDataTo(col); n.copyDataTo(ns); h.copyDataTo(hist); v.copyDataTo(cov);
	return 0;
}

// the host loops of the prefilter on their own (what filter() runs without a usable device): tests pin them against the reference's fixture on any box
int bcdcore_spike_filter_host(float* col, float* ns, float* hist, float* cov, int W, int H, int D, float factor)
{
	Deepimf c(W, H, 3), n(W, H, 1), h(W, H, D), v(W, H, 6);
	c.copyDataFrom(col); n.copyDataFrom(ns); h.copyDataFrom(hist); v.copyDataFrom(cov);
	SpikeRemovalFilter::filterOnHost(c, n, h, v, factor);
	c.copyDataTo(col); n.copyDataTo(ns); h.copyDataTo(hist); v.copyDataTo(cov);
	return 0;
}

int bcdcore_merge_hist_ns(const float* hist, const float* ns, int W, int H, int D, float* out)
{
	Deepimf h(W, H, D), n(W, H, 1);
	h.copyDataFrom(hist); n.copyDataFrom(ns);
	Utils::mergeHistogramAndNbOfSamples(h, n).copyDataTo(out);
	return 0;
}

int bcdcore_split_hist_ns(const float* in, int W, int H, int Dp1, float* hist, float* ns)
{
	Deepimf m(W, H, Dp1), h, n;
	m.copyDataFrom(in);
	if(!Utils::separateNbOfSamplesFromHistogram(h, n, m)) return -1;
	h.copyDataTo(hist); n.copyDataTo(ns);
	return 0;
}

} // extern "C"

// ---- EXR (ImageIO) ------------------------------------------------------------------------------------------
#include "ImageIO.h"

extern "C" {

int bcdcore_write_exr(const char* path, const float* data, int W, int H, int D, int multiChannels)
{
	Deepimf img(W, H, D);
	img.copyDataFrom(data);
	return (multiChannels ? ImageIO::writeMultiChannelsEXR(img, path) : ImageIO::writeEXR(img, path)) ? 0 : -1;
}

// two-call protocol: out == nullptr -> only the dimensions are returned
int bcdcore_read_exr(const char* path, int multiChannels, int* W, int* H, int* D, float* out, long long capacity)
{
	Deepimf img;
	if(!(multiChannels ? ImageIO::loadMultiChannelsEXR(img, path) : ImageIO::loadEXR(img, path))) return -1;
	*W = img.getWidth(); *H = img.getHeight(); *D = img.getDepth();
	if(out)
	{
		if(capacity < (long long)img.getSize()) return -2;
		img.copyDataTo(out);
	}
	return 0;
}

const char* bcdcore_exr_last_error() { return ImageIO::lastError().c_str(); }

} // extern "C"

// ---- .bcd.json presets (ParametersIO) ----------------------------------------------------------------------------
#include "ParametersIO.h"

extern "C" {

// round trip helper for the tests: loads `in` over the defaults, writes everything to `out`, returns a few fields
int bcdcore_presets_roundtrip(const char* in, const char* out, int* nbOfScales, float* tau, int* b, int* randomOrder, float* m, float* minEig,
		int* spike, float* spikeFactor, char* colorPath, int colorPathCapacity)
{
	PipelineParameters p;
	if(!ParametersIO::load(p, in)) return -1;
	if(out && !ParametersIO::write(p, out)) return -2;
	const DenoiserParameters& d = p.m_denoiserParameters.m_monoscaleParameters;
	*nbOfScales = p.m_denoiserParameters.m_nbOfScales; *tau = d.m_histogramDistanceThreshold; *b = d.m_searchWindowRadius;
	*randomOrder = d.m_useRandomPixelOrder ? 1 : 0; *m = d.m_markedPixelsSkippingProbability; *minEig = d.m_minEigenValue;
	*spike = p.m_prefilteringParameters.m_performSpikeRemoval ? 1 : 0; *spikeFactor = p.m_prefilteringParameters.m_spikeRemovalThresholdStDevFactor;
	snprintf(colorPath, size_t(colorPathCapacity), "%s", p.m_inputFileNames.m_colors.c_str());
	return 0;
}

} // extern "C"
