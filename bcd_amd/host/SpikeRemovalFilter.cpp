// SpikeRemovalFilter.cpp -- host wrapper: uploads the four images, runs the HIP kernel (bcd_hip_spike_filter,
// behaviour of the reference's src/core/SpikeRemovalFilter.cpp:18-116 with float-abs semantics), downloads in place.
#include "SpikeRemovalFilter.h"
#include "DeepImage.h"

#include "bcd_hip.h"

#include <hip/hip_runtime_api.h>
#include <iostream>

namespace bcd
{

	void SpikeRemovalFilter::filter(
			DeepImage<float>& io_rInputColorImage,
			DeepImage<float>& io_rInputNbOfSamplesImage,
			DeepImage<float>& io_rInputHistogramImage,
			DeepImage<float>& io_rInputCovImage,
			float i_thresholdStDevFactor)
	{
		// (the reference's signature has no way to report a failure; filterOnDevice has, and has already said why on cerr)
		(void)filterOnDevice(0, io_rInputColorImage, io_rInputNbOfSamplesImage, io_rInputHistogramImage, io_rInputCovImage, i_thresholdStDevFactor);
	}

	bool SpikeRemovalFilter::filterOnDevice(
			int i_device,
			DeepImage<float>& io_rInputColorImage,
			DeepImage<float>& io_rInputNbOfSamplesImage,
			DeepImage<float>& io_rInputHistogramImage,
			DeepImage<float>& io_rInputCovImage,
			float i_thresholdStDevFactor)
	{
		const int w = io_rInputColorImage.getWidth(), h = io_rInputColorImage.getHeight(), d = io_rInputHistogramImage.getDepth();
		bcd_hip_ctx* pCtx = nullptr;
		if(bcd_hip_ctx_create(&pCtx, i_device, nullptr) != BCD_HIP_OK)
		{
			std::cerr << "SpikeRemovalFilter: NOT APPLIED -- no usable HIP device " << i_device << " (this build has no CPU path); images left untouched" << std::endl;
			return false;
		}
		int previousDevice = -1;
		if(hipGetDevice(&previousDevice) != hipSuccess)
			previousDevice = -1;
		(void)hipSetDevice(i_device);
		DeepImage<float>* images[4] = { &io_rInputColorImage, &io_rInputNbOfSamplesImage, &io_rInputHistogramImage, &io_rInputCovImage };
		float* in[4] = { nullptr, nullptr, nullptr, nullptr };
		float* out[4] = { nullptr, nullptr, nullptr, nullptr };
		bool ok = true;
		for(int i = 0; i < 4 && ok; ++i)
		{
			const size_t bytes = sizeof(float) * size_t(images[i]->getSize());
			ok = hipMalloc(reinterpret_cast<void**>(&in[i]), bytes) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&out[i]), bytes) == hipSuccess
					&& hipMemcpy(in[i], images[i]->getDataPtr(), bytes, hipMemcpyHostToDevice) == hipSuccess;
		}
		if(ok)
			ok = bcd_hip_spike_filter(pCtx, in[0], in[1], in[2], in[3], w, h, d, i_thresholdStDevFactor, out[0], out[1], out[2], out[3]) == BCD_HIP_OK;
		bcd_hip_ctx_destroy(pCtx); // synchronises the context's stream
		for(int i = 0; i < 4 && ok; ++i)
			ok = hipMemcpy(images[i]->getDataPtr(), out[i], sizeof(float) * size_t(images[i]->getSize()), hipMemcpyDeviceToHost) == hipSuccess;
		for(int i = 0; i < 4; ++i)
		{
			if(in[i]) (void)hipFree(in[i]);
			if(out[i]) (void)hipFree(out[i]);
		}
		if(previousDevice >= 0)
			(void)hipSetDevice(previousDevice);
		if(!ok)
			std::cerr << "SpikeRemovalFilter: device error, images may be partially filtered" << std::endl;
		return ok;
	}

} // namespace bcd
