// ParametersIO.h -- .bcd.json pipeline presets; API of the reference's include/bcd/io/ParametersIO.h:20-80.
// The reference implements it on nlohmann::json and only its GUI uses it (bcd_cli advertises "-a <file>" in its usage,
// src/cli/main.cpp:107, without parsing it); here a small flat-object JSON reader/writer backs it and bcd_cli honours -a.
// File layout (src/io/ParametersIO.cpp:60-124 of the reference): ONE flat object with the keys
//   inputColorFile, inputHistoFile, inputCovarFile        (paths relative to the folder of the .bcd.json file)
//   performSpikeRemovalPrefiltering, spikeRemovalThresholdStDevFactor
//   nbOfScales, histoDistanceThreshold, useCuda, nbOfCores, patchRadius, searchWindowRadius, randomPixelOrder,
//   markedPixelsSkippingProbability, minEigenValue
// The declarations below keep the names and member layout of the reference's include/bcd/io/ParametersIO.h (BCD -- Bayesian Collaborative Denoising for
// Monte-Carlo Rendering, M. Boughida and T. Boubekeur, Computer Graphics Forum (Proc. EGSR 2017) 36(4); BSD-style licence, see the
// reference's LICENSE.txt) so that callers written against it compile unchanged; the implementation behind them is this build's own.
#ifndef PARAMETERS_IO_H
#define PARAMETERS_IO_H

#include "IDenoiser.h"

#include <string>

namespace bcd
{

	struct InputFileNames
	{
		std::string m_colors;
		std::string m_histograms;
		std::string m_covariances;
	};

	struct PrefilteringParameters
	{
		PrefilteringParameters() : m_performSpikeRemoval(true), m_spikeRemovalThresholdStDevFactor(1.5f) {}
		bool m_performSpikeRemoval;
		float m_spikeRemovalThresholdStDevFactor;
	};

	struct MultiscaleDenoiserParameters
	{
		MultiscaleDenoiserParameters() : m_nbOfScales(3), m_monoscaleParameters() {}
		int m_nbOfScales;
		DenoiserParameters m_monoscaleParameters;
	};

	struct PipelineParameters
	{
		InputFileNames m_inputFileNames;
		PrefilteringParameters m_prefilteringParameters;
		MultiscaleDenoiserParameters m_denoiserParameters;
	};

	struct PipelineParametersSelector
	{
		PipelineParametersSelector() : m_inputFileNames(true), m_prefilteringParameters(true), m_denoiserParameters(true) {}
		bool m_inputFileNames;
		bool m_prefilteringParameters;
		bool m_denoiserParameters;
	};

	class ParametersIO
	{
	private:
		ParametersIO() {}

	public:
		static const std::string& getPipelineParametersFileExtension()
		{
			static const std::string extension = "bcd.json";
			return extension;
		}

		/// keys absent from the file leave the corresponding members untouched
		static bool load(PipelineParameters& o_rParams, const std::string& i_rFilePath,
				PipelineParametersSelector i_selector = PipelineParametersSelector());

		static bool write(const PipelineParameters& i_rParams, const std::string& i_rFilePath,
				PipelineParametersSelector i_selector = PipelineParametersSelector());
	};

}

#endif // PARAMETERS_IO_H
