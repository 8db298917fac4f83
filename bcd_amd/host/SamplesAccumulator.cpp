// SamplesAccumulator.cpp -- weighted first/second moments and 3 x nbOfBins soft histograms per pixel
// (behaviour of the reference's src/core/SamplesAccumulator.cpp:44-153; host code, runs inside the renderer).
#include "SamplesAccumulator.h"
#include "CovarianceMatrix.h"

#include <cassert>
#include <cmath>
#include <utility>

namespace bcd
{

	SamplesStatisticsImages::SamplesStatisticsImages(int i_width, int i_height, int i_nbOfBins) :
			m_nbOfSamplesImage(i_width, i_height, 1),
			m_meanImage(i_width, i_height, 3),
			m_covarImage(i_width, i_height, 6),
			m_histoImage(i_width, i_height, 3 * i_nbOfBins)
	{
	}

	SamplesAccumulator::SamplesAccumulator(int i_width, int i_height, const HistogramParameters& i_rHistogramParameters) :
			m_width(i_width), m_height(i_height),
			m_histogramParameters(i_rHistogramParameters),
			m_samplesStatisticsImages(i_width, i_height, i_rHistogramParameters.m_nbOfBins),
			m_squaredWeightSumsImage(i_width, i_height, 1),
			m_isValid(true)
	{
		m_samplesStatisticsImages.m_nbOfSamplesImage.fill(0.f);
		m_samplesStatisticsImages.m_meanImage.fill(0.f);
		m_samplesStatisticsImages.m_covarImage.fill(0.f);
		m_samplesStatisticsImages.m_histoImage.fill(0.f);
		m_squaredWeightSumsImage.fill(0.f);
	}

	void SamplesAccumulator::addSample(int i_line, int i_column, float i_sampleR, float i_sampleG, float i_sampleB, float i_weight)
	{
		assert(m_isValid);
		const int nbOfBins = m_histogramParameters.m_nbOfBins;
		const float rgb[3] = { i_sampleR, i_sampleG, i_sampleB };
		const float saturation = 2.f; // normalised values above 1 are spread over the last two bins up to this level

		m_samplesStatisticsImages.m_nbOfSamplesImage.get(i_line, i_column, 0) += i_weight;
		m_squaredWeightSumsImage.get(i_line, i_column, 0) += i_weight * i_weight;

		float* pSum = &m_samplesStatisticsImages.m_meanImage.get(i_line, i_column, 0);
		pSum[0] += i_weight * i_sampleR;
		pSum[1] += i_weight * i_sampleG;
		pSum[2] += i_weight * i_sampleB;

		float* pCov = &m_samplesStatisticsImages.m_covarImage.get(i_line, i_column, 0);
		pCov[int(ESymMatData::e_xx)] += i_weight * i_sampleR * i_sampleR;
		pCov[int(ESymMatData::e_yy)] += i_weight * i_sampleG * i_sampleG;
		pCov[int(ESymMatData::e_zz)] += i_weight * i_sampleB * i_sampleB;
		pCov[int(ESymMatData::e_yz)] += i_weight * i_sampleG * i_sampleB;
		pCov[int(ESymMatData::e_xz)] += i_weight * i_sampleR * i_sampleB;
		pCov[int(ESymMatData::e_xy)] += i_weight * i_sampleR * i_sampleG;

		float* pHisto = &m_samplesStatisticsImages.m_histoImage.get(i_line, i_column, 0);
		for(int channel = 0; channel < 3; ++channel)
		{
			float v = rgb[channel] > 0 ? rgb[channel] : 0;
			if(m_histogramParameters.m_gamma > 1)
				v = std::pow(v, 1.f / m_histogramParameters.m_gamma);
			if(m_histogramParameters.m_maxValue > 0)
				v = v / m_histogramParameters.m_maxValue;
			if(v > saturation)
				v = saturation;

			const float binPosition = v * (nbOfBins - 2);
			int lowBin = int(binPosition);
			float highWeight;
			if(lowBin < nbOfBins - 2)
				highWeight = binPosition - lowBin; // linear split between two regular bins
			else
			{	// v >= 1: split between the last two bins according to the saturation level
				lowBin = nbOfBins - 2;
				highWeight = (v - 1.0f) / (saturation - 1.f);
			}
			const float lowWeight = 1.0f - highWeight;
			pHisto[channel * nbOfBins + lowBin] += i_weight * lowWeight;
			pHisto[channel * nbOfBins + lowBin + 1] += i_weight * highWeight;
		}
	}

	void SamplesAccumulator::computeSampleStatistics(SamplesStatisticsImages& io_sampleStats) const
	{
		for(int line = 0; line < m_height; ++line)
			for(int column = 0; column < m_width; ++column)
			{
				const float weightSum = io_sampleStats.m_nbOfSamplesImage.get(line, column, 0);
				const float squaredWeightSum = m_squaredWeightSumsImage.get(line, column, 0);
				const float invWeightSum = 1.f / weightSum;

				float* pMean = &io_sampleStats.m_meanImage.get(line, column, 0);
				float mean[3];
				for(int i = 0; i < 3; ++i)
					pMean[i] = mean[i] = invWeightSum * pMean[i];

				float* pCov = &io_sampleStats.m_covarImage.get(line, column, 0);
				float cov[6];
				for(int i = 0; i < 6; ++i)
					cov[i] = pCov[i] * invWeightSum;
				cov[int(ESymMatData::e_xx)] -= mean[0] * mean[0];
				cov[int(ESymMatData::e_yy)] -= mean[1] * mean[1];
				cov[int(ESymMatData::e_zz)] -= mean[2] * mean[2];
				cov[int(ESymMatData::e_yz)] -= mean[1] * mean[2];
				cov[int(ESymMatData::e_xz)] -= mean[0] * mean[2];
				cov[int(ESymMatData::e_xy)] -= mean[0] * mean[1];
				// unbiased estimate for weighted samples
				const float biasCorrectionFactor = 1.f / (1 - squaredWeightSum / (weightSum * weightSum));
				for(int i = 0; i < 6; ++i)
					pCov[i] = cov[i] * biasCorrectionFactor;
			}
	}

	SamplesStatisticsImages SamplesAccumulator::getSamplesStatistics() const
	{
		SamplesStatisticsImages stats(m_samplesStatisticsImages);
		computeSampleStatistics(stats);
		return stats;
	}

	SamplesStatisticsImages SamplesAccumulator::extractSamplesStatistics()
	{
		computeSampleStatistics(m_samplesStatisticsImages);
		m_isValid = false;
		return std::move(m_samplesStatisticsImages);
	}

	SamplesAccumulatorThreadSafe::SamplesAccumulatorThreadSafe(int i_width, int i_height, const HistogramParameters& i_rHistogramParameters) :
			SamplesAccumulator(i_width, i_height, i_rHistogramParameters),
			m_lockWidth(i_width),
			m_pixelLocks(new std::atomic_flag[size_t(i_width) * size_t(i_height)])
	{
		for(size_t i = 0, n = size_t(i_width) * size_t(i_height); i < n; ++i)
			m_pixelLocks[i].clear();
	}

	void SamplesAccumulatorThreadSafe::addSampleThreadSafely(int i_line, int i_column, float i_sampleR, float i_sampleG, float i_sampleB, float i_weight)
	{
		std::atomic_flag& rLock = m_pixelLocks[size_t(i_line) * size_t(m_lockWidth) + size_t(i_column)];
		while(rLock.test_and_set(std::memory_order_acquire))
		{	// another thread is inside addSample for this very pixel: a few dozen instructions, spin
		}
		addSample(i_line, i_column, i_sampleR, i_sampleG, i_sampleB, i_weight);
		rLock.clear(std::memory_order_release);
	}

} // namespace bcd
