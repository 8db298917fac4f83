// Denoiser.cpp -- bcd::Denoiser / bcd::MultiscaleDenoiser of the MI355X build: input validation with the
// reference's messages and return values (src/core/Denoiser.cpp:238-348), then one call into the C ABI
// (include/bcd_hip.h) which runs the whole loop on the device.  No CPU fallback: a missing device is an error.
#include "Denoiser.h"
#include "MultiscaleDenoiser.h"

#include "bcd_hip.h"

#include <iostream>

using namespace std;

namespace bcd
{

	bool Denoiser::inputsOutputsAreOk()
	{
		const DeepImage<float>* images[4] = { m_inputs.m_pColors, m_inputs.m_pNbOfSamples, m_inputs.m_pHistograms, m_inputs.m_pSampleCovariances };
		const char* names[4] = { "color", "number of samples", "histogram", "covariance" };
		bool ok = true;
		for(int i = 0; i < 4; ++i)
			if(!images[i])
			{
				ok = false;
				cerr << "Aborting denoising: nullptr for input " << names[i] << " image" << endl;
			}
		if(!ok)
			return false;
		if(!m_outputs.m_pDenoisedColors)
		{
			cerr << "Aborting denoising: nullptr for output image" << endl;
			return false;
		}
		for(int i = 0; i < 4; ++i)
			if(images[i]->isEmpty())
			{
				ok = false;
				cerr << "Aborting denoising: input " << names[i] << " image is empty" << endl;
			}
		if(!ok)
			return false;
		const int w = images[0]->getWidth(), h = images[0]->getHeight();
		for(int i = 1; i < 4; ++i)
			if(images[i]->getWidth() != w || images[i]->getHeight() != h)
			{
				ok = false;
				cerr << "Aborting denoising: input " << names[i] << " image is " << images[i]->getWidth() << "x" << images[i]->getHeight()
						<< " but input color image is " << w << "x" << h << endl;
			}
		if(!ok)
			return false;
		if(images[0]->getDepth() != 3 || images[1]->getDepth() != 1 || images[3]->getDepth() != 6)
		{
			cerr << "Aborting denoising: expected depths 3 / 1 / 6 for color / number of samples / covariance images" << endl;
			return false;
		}
		return true;
	}

	bool Denoiser::denoiseWithNbOfScales(int i_nbOfScales)
	{
		if(!inputsOutputsAreOk())
			return false;
		m_width = m_inputs.m_pColors->getWidth();
		m_height = m_inputs.m_pColors->getHeight();
		m_nbOfPixels = m_width * m_height;

		bcd_hip_ctx* pCtx = nullptr;
		int rc = bcd_hip_ctx_create(&pCtx, m_device, nullptr);
		if(rc != BCD_HIP_OK)
		{
			cerr << "Aborting denoising: no usable HIP device " << m_device << " (this build has no CPU path)" << endl;
			return false;
		}
		bcd_hip_params prm;
		bcd_hip_default_params(&prm);
		prm.hist_dist_threshold = m_parameters.m_histogramDistanceThreshold;
		prm.patch_radius = m_parameters.m_patchRadius;
		prm.search_radius = m_parameters.m_searchWindowRadius;
		prm.min_eigen_value = m_parameters.m_minEigenValue;
		prm.use_random_pixel_order = m_parameters.m_useRandomPixelOrder ? 1 : 0;
		prm.marked_skip_probability = m_parameters.m_markedPixelsSkippingProbability;
		prm.order_seed = m_orderSeed;

		m_progressCallback(0.f);
		Deepimf result(m_width, m_height, 3); // inputs may alias the output image (the CLI pre-copies colours into it)
		rc = bcd_hip_denoise_host(pCtx,
				m_inputs.m_pColors->getDataPtr(), m_inputs.m_pNbOfSamples->getDataPtr(),
				m_inputs.m_pHistograms->getDataPtr(), m_inputs.m_pSampleCovariances->getDataPtr(),
				m_width, m_height, m_inputs.m_pHistograms->getDepth(), i_nbOfScales, &prm, result.getDataPtr());
		if(rc != BCD_HIP_OK)
			cerr << "Aborting denoising: " << bcd_hip_last_error(pCtx) << endl;
		bcd_hip_ctx_destroy(pCtx);
		if(rc != BCD_HIP_OK)
			return false;
		*m_outputs.m_pDenoisedColors = std::move(result); // resized to W x H x 3 and overwritten (Denoiser.cpp:207-208)
		m_progressCallback(1.f);
		return true;
	}

	bool Denoiser::denoise()
	{
		return denoiseWithNbOfScales(1);
	}

	bool MultiscaleDenoiser::denoise()
	{
		if(m_nbOfScales < 1)
		{
			cerr << "Aborting denoising: number of scales must be >= 1" << endl;
			return false;
		}
		return denoiseWithNbOfScales(m_nbOfScales);
	}

} // namespace bcd
