// bcd_cli.cpp -- command-line front-end with the reference's flags, defaults and file conventions
// (src/cli/main.cpp:122-504 of the reference): -o -i -h -c -d -b -w -r -p --p-factor -m -s --ncores --use-cuda -e.
// Real defaults are -r 1 and -p 1 (main.cpp:52-53) although the reference's README says 0.  Missing -h / -c are
// inferred as <input>_hist.exr / <input>_cov.exr (:344-370).  Extra flags of this build: --seed <n> (visiting
// order), --device <n>.  --ncores only selects the visiting order (n > 1 with -r 0: the reference's strip list; the loop runs on the HIP device); --use-cuda 0 (a request for the CPU path this
// build does not have) is declined with a note and served by the device; under BCD_STRICT_CPU_REQUEST=1 it is refused with an error before any file is read.
// -a <file.bcd.json> (advertised but never parsed by the reference, main.cpp:107) loads a preset; later flags override it.
#include "Chronometer.h"
#include "DeepImage.h"
#include "Denoiser.h"
#include "ImageIO.h"
#include "MultiscaleDenoiser.h"
#include "ParametersIO.h"
#include "SpikeRemovalFilter.h"
#include "Utils.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>
#include <memory>
#include <string>

using namespace std;
using namespace bcd;

namespace
{

	struct ProgramArguments
	{
		string m_denoisedOutputFilePath;
		Deepimf m_colorImage, m_nbOfSamplesImage, m_histogramImage, m_covarianceImage;
		float m_histogramPatchDistanceThreshold = 1.f;
		int m_patchRadius = 1;
		int m_searchWindowRadius = 6;
		float m_minEigenValue = 1.e-8f;
		bool m_useRandomPixelOrder = true;
		bool m_prefilterSpikes = true;
		float m_prefilterThresholdStDevFactor = 2.f;
		float m_markedPixelsSkippingProbability = 1.f;
		int m_nbOfScales = 3;
		int m_nbOfCores = 0;
		bool m_useCuda = true;
		unsigned m_orderSeed = 1234u;
		std::vector<int> m_devices = std::vector<int>(1, 0);
	};

	const char* g_pProgramPath = "bcd_cli";

	void printUsage()
	{
		ProgramArguments d;
		cout << "Bayesian Collaborative Denoising (MI355X / HIP build)" << endl << endl;
		cout << "Usage: " << g_pProgramPath << " <arguments list>" << endl;
		cout << "Only EXR images are supported." << endl << endl;
		cout << "Required arguments list:" << endl;
		cout << "    -o <output>          The file path to the output image" << endl;
		cout << "    -i <input>           The file path to the input image" << endl;
		cout << "    -h <hist>            The file path to the input histograms buffer (default: <input>_hist.exr)" << endl;
		cout << "    -c <cov>             The file path to the input covariance matrices buffer (default: <input>_cov.exr)" << endl;
		cout << "Optional arguments list:" << endl;
		cout << "    -a <file>            The file path to the .bcd.json file containing arguments for the program" << endl;
		cout << "    -d <float>           Histogram patch distance threshold (default: " << d.m_histogramPatchDistanceThreshold << ")" << endl;
		cout << "    -b <int>             Radius of search windows (default: " << d.m_searchWindowRadius << ")" << endl;
		cout << "    -w <int>             Radius of patches (default: " << d.m_patchRadius << ")" << endl;
		cout << "    -r <0/1>             1 for random pixel order (in case of grid artifacts) (default: " << (d.m_useRandomPixelOrder ? 1 : 0) << ")" << endl;
		cout << "    -p <0/1>             1 for a spike removal prefiltering (default: " << (d.m_prefilterSpikes ? 1 : 0) << ")" << endl;
		cout << "    --p-factor <float>   Standard deviation factor of the spike threshold (default: " << d.m_prefilterThresholdStDevFactor << ")" << endl;
		cout << "    -m <float in [0,1]>  Probability of skipping marked centers of denoised patches (default: " << d.m_markedPixelsSkippingProbability << ")" << endl;
		cout << "    -s <int>             Number of Scales for Multi-Scaling (default: " << d.m_nbOfScales << ")" << endl;
		cout << "    --ncores <n>         the loop runs on the HIP device; n > 1 with -r 0 selects the reference's strip visiting order" << endl;
		cout << "    --use-cuda <0/1>     1 (default): run on the HIP device; 0 asks for the CPU path, which this build does not have: declined with a note, the device runs (BCD_STRICT_CPU_REQUEST=1: refused)" << endl;
		cout << "    -e <float>           Minimum eigen value for matrix inversion (default: " << d.m_minEigenValue << ")" << endl;
		cout << "    --seed <int>         Seed of the random pixel order (default: " << d.m_orderSeed << ")" << endl;
		cout << "    --device <int>       HIP device index (default: 0)" << endl;
		cout << "    --devices <list>     several HIP devices, e.g. 0-7 or 0,2,4: the frame is split into row bands (RCCL halo exchange)" << endl;
	}

	bool badValue(const char* flag, const char* what)
	{
		cout << "ERROR in program arguments: " << what << " after " << flag << endl;
		return false;
	}

	bool parseProgramArguments(int argc, const char** argv, ProgramArguments& a)
	{
		bool missingColor = true, missingHist = true, missingCov = true, missingOutput = true;
		string inputColorFilePath;
		for(int i = 1; i < argc; ++i)
		{
			const string flag = argv[i];
			if(flag == "--help") { printUsage(); return false; }
			if(i + 1 >= argc) { cout << "ERROR in program arguments: expecting a value after " << flag << endl; return false; }
			const char* value = argv[++i];
			if(flag == "-o") { a.m_denoisedOutputFilePath = value; missingOutput = false; }
			else if(flag == "-i")
			{
				inputColorFilePath = value;
				if(!ImageIO::loadEXR(a.m_colorImage, value)) { cout << "ERROR in program arguments: couldn't load input color image file '" << value << "'" << endl; return false; }
				missingColor = false;
			}
			else if(flag == "-h")
			{
				Deepimf histAndNbOfSamplesImage;
				if(!ImageIO::loadMultiChannelsEXR(histAndNbOfSamplesImage, value)) { cout << "ERROR in program arguments: couldn't load input histogram image file '" << value << "'" << endl; return false; }
				Utils::separateNbOfSamplesFromHistogram(a.m_histogramImage, a.m_nbOfSamplesImage, histAndNbOfSamplesImage);
				missingHist = false;
			}
			else if(flag == "-c")
			{
				if(!ImageIO::loadMultiChannelsEXR(a.m_covarianceImage, value)) { cout << "ERROR in program arguments: couldn't load input covariance matrix image file '" << value << "'" << endl; return false; }
				missingCov = false;
			}
			else if(flag == "-a")
			{
				PipelineParameters preset;
				preset.m_prefilteringParameters.m_performSpikeRemoval = a.m_prefilterSpikes;
				preset.m_prefilteringParameters.m_spikeRemovalThresholdStDevFactor = a.m_prefilterThresholdStDevFactor;
				preset.m_denoiserParameters.m_nbOfScales = a.m_nbOfScales;
				DenoiserParameters& p = preset.m_denoiserParameters.m_monoscaleParameters;
				p.m_histogramDistanceThreshold = a.m_histogramPatchDistanceThreshold; p.m_patchRadius = a.m_patchRadius;
				p.m_searchWindowRadius = a.m_searchWindowRadius; p.m_minEigenValue = a.m_minEigenValue;
				p.m_useRandomPixelOrder = a.m_useRandomPixelOrder; p.m_markedPixelsSkippingProbability = a.m_markedPixelsSkippingProbability;
				if(!ParametersIO::load(preset, value)) { cout << "ERROR in program arguments: couldn't load the parameters file '" << value << "'" << endl; return false; }
				a.m_prefilterSpikes = preset.m_prefilteringParameters.m_performSpikeRemoval;
				a.m_prefilterThresholdStDevFactor = preset.m_prefilteringParameters.m_spikeRemovalThresholdStDevFactor;
				a.m_nbOfScales = preset.m_denoiserParameters.m_nbOfScales;
				a.m_histogramPatchDistanceThreshold = p.m_histogramDistanceThreshold; a.m_patchRadius = p.m_patchRadius;
				a.m_searchWindowRadius = p.m_searchWindowRadius; a.m_minEigenValue = p.m_minEigenValue;
				a.m_useRandomPixelOrder = p.m_useRandomPixelOrder; a.m_markedPixelsSkippingProbability = p.m_markedPixelsSkippingProbability;
				const InputFileNames& names = preset.m_inputFileNames;
				if(!names.m_colors.empty() && missingColor)
				{
					inputColorFilePath = names.m_colors;
					if(!ImageIO::loadEXR(a.m_colorImage, names.m_colors.c_str())) { cout << "ERROR in program arguments: couldn't load input color image file '" << names.m_colors << "'" << endl; return false; }
					missingColor = false;
				}
				if(!names.m_histograms.empty() && missingHist)
				{
					Deepimf histAndNbOfSamplesImage;
					if(!ImageIO::loadMultiChannelsEXR(histAndNbOfSamplesImage, names.m_histograms.c_str())) { cout << "ERROR in program arguments: couldn't load input histogram image file '" << names.m_histograms << "'" << endl; return false; }
					Utils::separateNbOfSamplesFromHistogram(a.m_histogramImage, a.m_nbOfSamplesImage, histAndNbOfSamplesImage);
					missingHist = false;
				}
				if(!names.m_covariances.empty() && missingCov)
				{
					if(!ImageIO::loadMultiChannelsEXR(a.m_covarianceImage, names.m_covariances.c_str())) { cout << "ERROR in program arguments: couldn't load input covariance matrix image file '" << names.m_covariances << "'" << endl; return false; }
					missingCov = false;
				}
			}
			else if(flag == "-d") { a.m_histogramPatchDistanceThreshold = float(atof(value)); if(a.m_histogramPatchDistanceThreshold <= 0.f) return badValue("-d", "expecting a positive floating number"); }
			else if(flag == "-b") { a.m_searchWindowRadius = atoi(value); if(a.m_searchWindowRadius < 0) return badValue("-b", "expecting a non-negative integer"); }
			else if(flag == "-w") { a.m_patchRadius = atoi(value); if(a.m_patchRadius < 0) return badValue("-w", "expecting a non-negative integer"); }
			else if(flag == "-e") { a.m_minEigenValue = float(atof(value)); if(a.m_minEigenValue <= 0.f) return badValue("-e", "expecting a positive floating number"); }
			else if(flag == "-r") { const int v = atoi(value); if(v != 0 && v != 1) return badValue("-r", "expecting 0 or 1"); a.m_useRandomPixelOrder = v == 1; }
			else if(flag == "-p") { const int v = atoi(value); if(v != 0 && v != 1) return badValue("-p", "expecting 0 or 1"); a.m_prefilterSpikes = v == 1; }
			else if(flag == "--p-factor") { a.m_prefilterThresholdStDevFactor = float(atof(value)); if(a.m_prefilterThresholdStDevFactor <= 0.f) return badValue("--p-factor", "expecting a positive floating number"); }
			else if(flag == "-m") { a.m_markedPixelsSkippingProbability = float(atof(value)); if(a.m_markedPixelsSkippingProbability < 0.f || a.m_markedPixelsSkippingProbability > 1.f) return badValue("-m", "expecting a floating number between 0 and 1"); }
			else if(flag == "-s") { a.m_nbOfScales = atoi(value); if(a.m_nbOfScales <= 0) return badValue("-s", "expecting a positive integer"); }
			else if(flag == "--ncores") { a.m_nbOfCores = atoi(value); }
			else if(flag == "--use-cuda") { a.m_useCuda = atoi(value) == 1; }
			else if(flag == "--seed") { a.m_orderSeed = unsigned(strtoul(value, nullptr, 10)); }
			else if(flag == "--device") { a.m_devices.assign(1, atoi(value)); }
			else if(flag == "--devices")
			{	// "0-7", "0,1,2", "0-3,6"
				a.m_devices.clear();
				const string list(value);
				size_t pos = 0;
				while(pos < list.size())
				{
					size_t end = list.find(',', pos);
					if(end == string::npos) end = list.size();
					const string item = list.substr(pos, end - pos);
					const size_t dash = item.find('-');
					const int first = atoi(item.substr(0, dash).c_str()), last = dash == string::npos ? first : atoi(item.substr(dash + 1).c_str());
					if(item.empty() || first < 0 || last < first) return badValue("--devices", "expecting a list like 0-7 or 0,2,4");
					for(int d = first; d <= last; ++d) a.m_devices.push_back(d);
					pos = end + 1;
				}
				if(a.m_devices.empty()) return badValue("--devices", "expecting a list like 0-7 or 0,2,4");
			}
			else { cout << "ERROR in program arguments: unknown argument " << flag << endl << endl; printUsage(); return false; }
		}
		if(!missingColor && inputColorFilePath.length() > 4)
		{
			const string stem = inputColorFilePath.substr(0, inputColorFilePath.length() - 4); // drops ".exr"
			if(missingHist)
			{
				const string path = stem + "_hist.exr";
				cout << "Warning: input histogram file not provided by -h argument: assuming '" << path << "'" << endl;
				Deepimf histAndNbOfSamplesImage;
				if(!ImageIO::loadMultiChannelsEXR(histAndNbOfSamplesImage, path.c_str())) { cout << "ERROR in program arguments: couldn't load input histogram image file '" << path << "'" << endl; return false; }
				Utils::separateNbOfSamplesFromHistogram(a.m_histogramImage, a.m_nbOfSamplesImage, histAndNbOfSamplesImage);
				missingHist = false;
			}
			if(missingCov)
			{
				const string path = stem + "_cov.exr";
				cout << "Warning: input covariance file not provided by -c argument: assuming '" << path << "'" << endl;
				if(!ImageIO::loadMultiChannelsEXR(a.m_covarianceImage, path.c_str())) { cout << "ERROR in program arguments: couldn't load input covariance matrix image file '" << path << "'" << endl; return false; }
				missingCov = false;
			}
		}
		if(missingColor || missingHist || missingCov || missingOutput)
		{
			cout << "ERROR: Missing required program argument(s):";
			if(missingColor) cout << " -i";
			if(missingHist) cout << " -h";
			if(missingCov) cout << " -c";
			if(missingOutput) cout << " -o";
			cout << endl << endl;
			printUsage();
			return false;
		}
		return true;
	}

	// negative, infinite and NaN values are put to zero before writing (src/cli/main.cpp:389-420)
	void checkAndPutToZeroNegativeInfNaNValues(DeepImage<float>& io_rImage)
	{
		float* p = io_rImage.getDataPtr();
		for(int i = 0, n = io_rImage.getSize(); i < n; ++i)
			if(p[i] < 0 || std::isnan(p[i]) || std::isinf(p[i]))
				p[i] = 0.f;
	}

	int launchBayesianCollaborativeDenoising(int argc, const char** argv)
	{
		// --use-cuda 0 asks for the reference's CPU/OpenMP loop, which this build does not have (one product path: the HIP device).  The library
		// declines the request with a note and runs on the device (Denoiser.cpp); under BCD_STRICT_CPU_REQUEST=1 it refuses instead, and that is
		// said here, before the input files are read, not after minutes of EXR decoding
		{
			const char* pStrict = getenv("BCD_STRICT_CPU_REQUEST");
			for(int i = 1; i + 1 < argc; ++i)
				if(string(argv[i]) == "--use-cuda" && atoi(argv[i + 1]) != 1 && pStrict != nullptr && pStrict[0] == '1')
				{
					cerr << "bcd_cli: --use-cuda 0 requests the CPU/OpenMP path, which this build does not have, and BCD_STRICT_CPU_REQUEST=1 forbids "
							"answering it with the HIP device; nothing was read or written" << endl;
					return 2;
				}
		}
		ProgramArguments args;
		if(!parseProgramArguments(argc, argv, args))
			return 1;
		if(args.m_colorImage.getDepth() == 1)
		{	// grey colour file: the denoiser works on 3 channels
			Deepimf rgb(args.m_colorImage.getWidth(), args.m_colorImage.getHeight(), 3);
			for(int i = 0, n = args.m_colorImage.getSize(); i < n; ++i)
				rgb.get(3 * i) = rgb.get(3 * i + 1) = rgb.get(3 * i + 2) = args.m_colorImage.get(i);
			args.m_colorImage = std::move(rgb);
		}
		// -p 1 (src/cli/main.cpp:428-441): on one device the prefilter runs on the uploaded copies, right before the denoiser (one trip
		// over PCIe for everything); with several devices it runs first, as in the reference
		const bool prefilterOnDevice = args.m_prefilterSpikes && args.m_devices.size() == 1;
		if(args.m_prefilterSpikes && !prefilterOnDevice)
			SpikeRemovalFilter::filter(args.m_colorImage, args.m_nbOfSamplesImage, args.m_histogramImage, args.m_covarianceImage, args.m_prefilterThresholdStDevFactor);

		DenoiserInputs inputs;
		inputs.m_pColors = &args.m_colorImage;
		inputs.m_pNbOfSamples = &args.m_nbOfSamplesImage;
		inputs.m_pHistograms = &args.m_histogramImage;
		inputs.m_pSampleCovariances = &args.m_covarianceImage;
		Deepimf outputDenoisedColorImage(args.m_colorImage);
		DenoiserOutputs outputs;
		outputs.m_pDenoisedColors = &outputDenoisedColorImage;
		DenoiserParameters parameters;
		parameters.m_histogramDistanceThreshold = args.m_histogramPatchDistanceThreshold;
		parameters.m_patchRadius = args.m_patchRadius;
		parameters.m_searchWindowRadius = args.m_searchWindowRadius;
		parameters.m_minEigenValue = args.m_minEigenValue;
		parameters.m_useRandomPixelOrder = args.m_useRandomPixelOrder;
		parameters.m_markedPixelsSkippingProbability = args.m_markedPixelsSkippingProbability;
		parameters.m_nbOfCores = args.m_nbOfCores;
		parameters.m_useCuda = args.m_useCuda;

		unique_ptr<IDenoiser> uDenoiser;
		HipEngineSettings* pSettings = nullptr;
		if(args.m_nbOfScales > 1) { MultiscaleDenoiser* p = new MultiscaleDenoiser(args.m_nbOfScales); uDenoiser.reset(p); pSettings = p; }
		else { Denoiser* p = new Denoiser(); uDenoiser.reset(p); pSettings = p; }
		pSettings->setOrderSeed(args.m_orderSeed);
		pSettings->setDevices(args.m_devices);
		if(prefilterOnDevice)
			pSettings->setSpikePrefilter(args.m_prefilterThresholdStDevFactor);
		pSettings->setZeroBadOutputValues(true); // checkAndPutToZeroNegativeInfNaNValues (src/cli/main.cpp:470) before the download
		uDenoiser->setInputs(inputs);
		uDenoiser->setOutputs(outputs);
		uDenoiser->setParameters(parameters);
		if(!uDenoiser->denoise())
			return 2;

		checkAndPutToZeroNegativeInfNaNValues(outputDenoisedColorImage); // (already clean: kept as the reference's last line of defence)
		if(!ImageIO::writeEXR(outputDenoisedColorImage, args.m_denoisedOutputFilePath.c_str()))
		{
			cerr << "Couldn't write " << args.m_denoisedOutputFilePath << ": " << ImageIO::lastError() << endl;
			return 3;
		}
		cout << "Written denoised output in file " << args.m_denoisedOutputFilePath << endl;
		return 0;
	}

}

int main(int argc, const char** argv)
{
	Chronometer programTotalTime;
	programTotalTime.start();
	g_pProgramPath = argv[0];
	const int rc = launchBayesianCollaborativeDenoising(argc, argv);
	programTotalTime.stop();
	cout << "Program total time: ";
	programTotalTime.printElapsedTime();
	cout << endl;
	return rc;
}
