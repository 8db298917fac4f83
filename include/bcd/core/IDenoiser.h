// IDenoiser.h -- the drop-in boundary of libbcdcore (MI355X build).
// Same class / member names, defaults and ownership rules as the reference's include/bcd/core/IDenoiser.h:20-97
// (callers: src/cli/main.cpp:436-468, src/gui/GuiWindow.cpp:345-376 of the reference).
#ifndef I_DENOISER_H
#define I_DENOISER_H

#include <functional>
#include <memory>

namespace bcd
{

	template<class T> class DeepImage;

	/// Algorithm parameters; copied by value into the denoiser (setParameters)
	class DenoiserParameters
	{
	public:
		DenoiserParameters() :
				m_histogramDistanceThreshold(1.f),
				m_patchRadius(1),
				m_searchWindowRadius(6),
				m_minEigenValue(1.e-8f),
				m_useRandomPixelOrder(true),
				m_markedPixelsSkippingProbability(1.f),
				m_nbOfCores(0),
				m_useCuda(true)
		{
		}

	public:
		float m_histogramDistanceThreshold; ///< similar iff chi-square patch distance <= threshold
		int m_patchRadius; ///< patches are (2r+1)^2 pixels
		int m_searchWindowRadius; ///< similar patches are searched in a (2r+1)^2 window
		float m_minEigenValue; ///< eigenvalue floor used when inverting covariance matrices
		bool m_useRandomPixelOrder; ///< visit main pixels in a (seeded, reproducible) pseudo-random order
		float m_markedPixelsSkippingProbability; ///< 1: skip centres of already denoised patches; 0: process every pixel
		int m_nbOfCores; ///< <= 0: OpenMP's default; the thread count the reference would run with is written back by denoise() and, without m_useRandomPixelOrder, selects the visiting order (> 1: strip list, 1: scanline) -- the loop itself runs on the HIP device
		bool m_useCuda; ///< a request, as in the reference: true (default) = the HIP device; false = the CPU/OpenMP path, which this build does not have -- declined with a note on cout and served by the device (refused with `false` under BCD_STRICT_CPU_REQUEST=1)
	};

	/// Non-owning pointers to the four input images (must outlive denoise())
	class DenoiserInputs
	{
	public:
		DenoiserInputs() : m_pColors(nullptr), m_pNbOfSamples(nullptr), m_pHistograms(nullptr), m_pSampleCovariances(nullptr) {}

	public:
		const DeepImage<float>* m_pColors; ///< W x H x 3 mean colours
		const DeepImage<float>* m_pNbOfSamples; ///< W x H x 1
		const DeepImage<float>* m_pHistograms; ///< W x H x (3 x bins)
		const DeepImage<float>* m_pSampleCovariances; ///< W x H x 6 (xx,yy,zz,yz,xz,xy)
	};

	class DenoiserOutputs
	{
	public:
		DenoiserOutputs() : m_pDenoisedColors(nullptr) {}

	public:
		DeepImage<float>* m_pDenoisedColors; ///< resized to W x H x 3 and overwritten
	};

	/// Interface of the monoscale and multiscale denoisers
	class IDenoiser
	{
	public:
		IDenoiser() : m_progressCallback([](float){}) {}
		virtual ~IDenoiser() {}

	public:
		virtual bool denoise() = 0; ///< blocking; false on invalid inputs or device failure (message on cerr)

	public:
		const DenoiserInputs& getInputs() const { return m_inputs; }
		void setInputs(const DenoiserInputs& i_rInputs) { m_inputs = i_rInputs; }
		const DenoiserOutputs& getOutputs() const { return m_outputs; }
		void setOutputs(const DenoiserOutputs& i_rOutputs) { m_outputs = i_rOutputs; }
		const DenoiserParameters& getParameters() const { return m_parameters; }
		void setParameters(const DenoiserParameters& i_rParameters) { m_parameters = i_rParameters; }

		void setProgressCallback(std::function<void(float)> i_progressCallback) { m_progressCallback = i_progressCallback; }

	protected:
		DenoiserParameters m_parameters;
		DenoiserInputs m_inputs;
		DenoiserOutputs m_outputs;
		std::function<void(float)> m_progressCallback;
	};

} // namespace bcd

#endif // I_DENOISER_H
