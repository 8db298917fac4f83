// Denoiser.h -- monoscale Bayesian collaborative denoiser, MI355X build.
// Public surface of the reference's include/bcd/core/Denoiser.h:25-44; the per-thread accumulator getters
// that only DenoisingUnit used (:36-44) do not exist here: the whole loop runs on the device behind
// include/bcd_hip.h.
#ifndef DENOISER_H
#define DENOISER_H

#include "IDenoiser.h"
#include "DeepImage.h"

#include <cstdint>

namespace bcd
{

	class Denoiser : public IDenoiser
	{
	public:
		Denoiser() : IDenoiser(), m_width(0), m_height(0), m_nbOfPixels(0), m_orderSeed(1234u), m_device(0) {}
		virtual ~Denoiser() {}

	public:
		virtual bool denoise();

		/// null / empty / size-mismatch checks of the reference (src/core/Denoiser.cpp:238-348); prints to cerr
		bool inputsOutputsAreOk();

		int getImagesWidth() const { return m_width; }
		int getImagesHeight() const { return m_height; }

		/// extensions of this build: seed of the visiting order (the reference uses the wall clock) and HIP device index
		void setOrderSeed(uint32_t i_seed) { m_orderSeed = i_seed; }
		void setDevice(int i_device) { m_device = i_device; }

	protected:
		bool denoiseWithNbOfScales(int i_nbOfScales);

	private:
		int m_width;
		int m_height;
		int m_nbOfPixels;
		uint32_t m_orderSeed;
		int m_device;
	};

} // namespace bcd

#endif // DENOISER_H
