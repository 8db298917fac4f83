// SpikeRemovalFilter.cpp -- the outlier prefilter of bcd_cli -p (behaviour of the reference's src/core/SpikeRemovalFilter.cpp:18-116 with
// float-abs semantics).  filter() uploads the four images, runs the HIP kernel (bcd_hip_spike_filter) and downloads in place; when no HIP device
// is usable it runs filterOnHost(), the same decision and copies as plain loops (the reference's filter is a host function: callers must not get
// their images back untouched).  The host path is pinned bit for bit by tests/golden/ref_spike.npz (outputs of the reference's own translation unit).
#include "SpikeRemovalFilter.h"
#include "DeepImage.h"

#include "bcd_hip.h"

#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <cmath>
#include <iostream>
#include <vector>

namespace bcd
{

	void SpikeRemovalFilter::filter(
			DeepImage<float>& io_rInputColorImage,
			DeepImage<float>& io_rInputNbOfSamplesImage,
			DeepImage<float>& io_rInputHistogramImage,
			DeepImage<float>& io_rInputCovImage,
			float i_thresholdStDevFactor)
	{
		// (the reference's signature has no way to report a failure: without a usable device the host loops do the work)
		if(bcd_hip_device_count() > 0
				&& filterOnDevice(0, io_rInputColorImage, io_rInputNbOfSamplesImage, io_rInputHistogramImage, io_rInputCovImage, i_thresholdStDevFactor, true))
			return;
		filterOnHost(io_rInputColorImage, io_rInputNbOfSamplesImage, io_rInputHistogramImage, io_rInputCovImage, i_thresholdStDevFactor);
	}

	void SpikeRemovalFilter::filterOnHost(
			DeepImage<float>& io_rInputColorImage,
			DeepImage<float>& io_rInputNbOfSamplesImage,
			DeepImage<float>& io_rInputHistogramImage,
			DeepImage<float>& io_rInputCovImage,
			float i_thresholdStDevFactor)
	{
		// Same steps as k_spike (bcd_amd/csrc/k_pointwise.hip): every pixel is decided on the UNFILTERED colours (the images are copied first), a
		// spike takes all four images' values from one neighbour.
		const int w = io_rInputColorImage.getWidth(), h = io_rInputColorImage.getHeight();
		if(w < 3 || h < 3)
			return; // (no 3x3 neighbourhood fits)
		const int depth[4] = { io_rInputColorImage.getDepth(), io_rInputNbOfSamplesImage.getDepth(), io_rInputHistogramImage.getDepth(), io_rInputCovImage.getDepth() };
		DeepImage<float>* images[4] = { &io_rInputColorImage, &io_rInputNbOfSamplesImage, &io_rInputHistogramImage, &io_rInputCovImage };
		std::vector<float> before[4];
		for(int i = 0; i < 4; ++i)
			before[i].assign(images[i]->getDataPtr(), images[i]->getDataPtr() + images[i]->getSize());
		const float* colours = before[0].data();
#pragma omp parallel for schedule(static)
		for(int line = 0; line < h; ++line)
			for(int col = 0; col < w; ++col)
			{
				// the neighbourhood is the 3x3 block centred on the pixel, moved inward at the image border
				const int centreLine = line < 1 ? 1 : (line > h - 2 ? h - 2 : line);
				const int centreCol = col < 1 ? 1 : (col > w - 2 ? w - 2 : col);
				float v[3][9];
				int k = 0;
				for(int nl = centreLine - 1; nl <= centreLine + 1; ++nl)
					for(int nc = centreCol - 1; nc <= centreCol + 1; ++nc, ++k)
						for(int ch = 0; ch < 3; ++ch)
							v[ch][k] = colours[(size_t(nl) * w + nc) * 3 + ch];
				const float* me = colours + (size_t(line) * w + col) * 3;
				bool spike = false;
				for(int ch = 0; ch < 3; ++ch)
				{
					float total = 0.f;
					for(int i = 0; i < 9; ++i)
						total += v[ch][i];
					const float average = total / 9;
					total = 0.f;
					for(int i = 0; i < 9; ++i)
						total += (v[ch][i] - average) * (v[ch][i] - average);
					const float standardDeviation = std::sqrt(total / 8); // (sample deviation: n - 1)
					spike = spike || (std::fabs(me[ch] - average) > i_thresholdStDevFactor * standardDeviation);
				}
				if(!spike)
					continue;
				// the neighbour whose summed L1 colour distance to the nine is smallest (the first one on a tie)
				int best = 0;
				float bestDistance = -1.f;
				for(int m = 0; m < 9; ++m)
				{
					float distance = 0.f;
					for(int i = 0; i < 9; ++i)
						distance += std::fabs(v[0][i] - v[0][m]) + std::fabs(v[1][i] - v[1][m]) + std::fabs(v[2][i] - v[2][m]);
					if(bestDistance < 0.f || distance < bestDistance)
					{
						bestDistance = distance;
						best = m;
					}
				}
				const size_t source = size_t(centreLine - 1 + best / 3) * w + (centreCol - 1 + best % 3), target = size_t(line) * w + col;
				for(int i = 0; i < 4; ++i)
					for(int d = 0; d < depth[i]; ++d)
						images[i]->getDataPtr()[target * depth[i] + d] = before[i][source * depth[i] + d];
			}
	}

	bool SpikeRemovalFilter::filterOnDevice(
			int i_device,
			DeepImage<float>& io_rInputColorImage,
			DeepImage<float>& io_rInputNbOfSamplesImage,
			DeepImage<float>& io_rInputHistogramImage,
			DeepImage<float>& io_rInputCovImage,
			float i_thresholdStDevFactor,
			bool i_quiet)
	{
		const int w = io_rInputColorImage.getWidth(), h = io_rInputColorImage.getHeight(), d = io_rInputHistogramImage.getDepth();
		bcd_hip_ctx* pCtx = nullptr;
		if(bcd_hip_ctx_create(&pCtx, i_device, nullptr) != BCD_HIP_OK)
		{
			if(!i_quiet)
				std::cerr << "SpikeRemovalFilter: no usable HIP device " << i_device << "; images left untouched (filter() or filterOnHost() run the host loops)" << std::endl;
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
		// All or nothing: the four results are downloaded into temporaries and only committed once every copy has succeeded, so a failure at
		// any point leaves the caller's images exactly as they came in (filter() relies on this when it falls back to the host loops: they must
		// decide spikes on unfiltered colours and copy from unfiltered images).
		std::vector<float> filtered[4];
		for(int i = 0; i < 4 && ok; ++i)
		{
			filtered[i].resize(size_t(images[i]->getSize()));
			ok = hipMemcpy(filtered[i].data(), out[i], sizeof(float) * filtered[i].size(), hipMemcpyDeviceToHost) == hipSuccess;
		}
		if(ok)
			for(int i = 0; i < 4; ++i)
				std::copy(filtered[i].begin(), filtered[i].end(), images[i]->getDataPtr());
		for(int i = 0; i < 4; ++i)
		{
			if(in[i]) (void)hipFree(in[i]);
			if(out[i]) (void)hipFree(out[i]);
		}
		if(previousDevice >= 0)
			(void)hipSetDevice(previousDevice);
		if(!ok)
			std::cerr << "SpikeRemovalFilter: device error, images left untouched" << std::endl;
		return ok;
	}

} // namespace bcd
