// SamplesAccumulator.h -- renderer-side statistics producer (mean, covariance, soft histograms), host code.
// API of the reference's include/bcd/core/SamplesAccumulator.h:19-80.
#ifndef SAMPLES_ACCUMULATOR_H
#define SAMPLES_ACCUMULATOR_H

#include "DeepImage.h"

#include <atomic>
#include <memory>

namespace bcd
{

	struct HistogramParameters
	{
		HistogramParameters() : m_nbOfBins(20), m_gamma(2.2f), m_maxValue(2.5f) {}

		int m_nbOfBins;
		float m_gamma; ///< bins grow exponentially: values are raised to 1/gamma before binning
		float m_maxValue; ///< value mapped to the last regular bin
	};

	struct SamplesStatisticsImages
	{
		SamplesStatisticsImages() = default;
		SamplesStatisticsImages(int i_width, int i_height, int i_nbOfBins);

		DeepImage<float> m_nbOfSamplesImage;
		DeepImage<float> m_meanImage;
		DeepImage<float> m_covarImage;
		DeepImage<float> m_histoImage;
	};

	class SamplesAccumulator
	{
	public:
		SamplesAccumulator(int i_width, int i_height, const HistogramParameters& i_rHistogramParameters);

		void addSample(int i_line, int i_column, float i_sampleR, float i_sampleG, float i_sampleB, float i_weight = 1.f);

		/// copy of the statistics accumulated so far
		SamplesStatisticsImages getSamplesStatistics() const;

		/// moves the statistics out; the accumulator must not be used afterwards
		SamplesStatisticsImages extractSamplesStatistics();

	private:
		void computeSampleStatistics(SamplesStatisticsImages& io_sampleStats) const;

	private:
		int m_width;
		int m_height;
		HistogramParameters m_histogramParameters;
		SamplesStatisticsImages m_samplesStatisticsImages;
		DeepImage<float> m_squaredWeightSumsImage;
		bool m_isValid;
	};

	/// Accumulator that several render threads may feed at once (declared by the reference, include/bcd/core/SamplesAccumulator.h:82-97,
	/// whose constructor is never defined and whose addSampleThreadSafely takes no lock, src/core/SamplesAccumulator.cpp:156-165).
	/// Here: one spin flag per pixel, held for the duration of one addSample; samples of different pixels never contend.  The
	/// statistics of a pixel depend on the order its samples arrive in only through fp32 rounding.
	class SamplesAccumulatorThreadSafe : public SamplesAccumulator
	{
	public:
		SamplesAccumulatorThreadSafe(int i_width, int i_height, const HistogramParameters& i_rHistogramParameters);

		void addSampleThreadSafely(int i_line, int i_column, float i_sampleR, float i_sampleG, float i_sampleB, float i_weight = 1.f);

	private:
		int m_lockWidth;
		std::unique_ptr< std::atomic_flag[] > m_pixelLocks;
	};

} // namespace bcd

#endif // SAMPLES_ACCUMULATOR_H
