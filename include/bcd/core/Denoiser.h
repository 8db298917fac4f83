// Denoiser.h -- monoscale Bayesian collaborative denoiser, MI355X build.
// Public surface of the reference's include/bcd/core/Denoiser.h:25-44; the per-thread accumulator getters
// that only DenoisingUnit used (:36-44) do not exist here: the whole loop runs on the device behind
// include/bcd_hip.h.
#ifndef DENOISER_H
#define DENOISER_H

#include "IDenoiser.h"
#include "DeepImage.h"

#include <cstdint>
#include <vector>

namespace bcd
{

	/// Settings of this build that the reference's API has no place for.  Both denoiser classes expose them through the same
	/// setters; defaults reproduce the reference's behaviour on device 0.
	class HipEngineSettings
	{
	public:
		HipEngineSettings() : m_orderSeed(1234u), m_devices(1, 0), m_prefilterThresholdStDevFactor(0.f), m_zeroBadOutputValues(false) {}

		/// seed of the visiting order of -r 1 (the reference seeds its shuffle from the wall clock, src/core/Denoiser.cpp:418)
		void setOrderSeed(uint32_t i_seed) { m_orderSeed = i_seed; }
		/// HIP device; several devices split the frame into row bands (bcd_hip_multi_*, RCCL over xGMI between neighbours)
		void setDevice(int i_device) { m_devices.assign(1, i_device); }
		void setDevices(const std::vector<int>& i_rDevices) { if(!i_rDevices.empty()) m_devices = i_rDevices; }
		const std::vector<int>& getDevices() const { return m_devices; }
		/// > 0: run SpikeRemovalFilter::filter on the uploaded copies of the inputs before denoising (what bcd_cli -p 1 does on
		/// the host, src/cli/main.cpp:428-441); the caller's images are not modified.  One device only.
		void setSpikePrefilter(float i_thresholdStDevFactor) { m_prefilterThresholdStDevFactor = i_thresholdStDevFactor; }
		/// put negative / infinite / NaN output values to zero on the device (src/cli/main.cpp:389-420)
		void setZeroBadOutputValues(bool i_enabled) { m_zeroBadOutputValues = i_enabled; }

	protected:
		uint32_t m_orderSeed;
		std::vector<int> m_devices;
		float m_prefilterThresholdStDevFactor;
		bool m_zeroBadOutputValues;
	};

	/// The engine contexts behind denoise() (device workspaces, pyramids, staging buffers: grow-only, sized by the largest frame seen,
	/// one set per device / device list) live until the process ends so that a sequence of frames pays for them once.  A long-lived
	/// host application calls this to give the device memory back; the next denoise() builds what it needs again.  Thread safe; waits
	/// for calls in flight.
	void releaseEngines();

	class Denoiser : public IDenoiser, public HipEngineSettings
	{
	public:
		Denoiser() : IDenoiser(), m_width(0), m_height(0), m_nbOfPixels(0) {}
		virtual ~Denoiser() {}

	public:
		virtual bool denoise();

		/// null / empty / size-mismatch checks of the reference (src/core/Denoiser.cpp:238-348); prints to cerr
		bool inputsOutputsAreOk();

		int getImagesWidth() const { return m_width; }
		int getImagesHeight() const { return m_height; }

		/// the whole path for 1..n scales (MultiscaleDenoiser forwards here: the pyramid, the scales and the merges are one call
		/// into the engine)
		bool denoiseWithNbOfScales(int i_nbOfScales);

	private:
		int m_width;
		int m_height;
		int m_nbOfPixels;
	};

} // namespace bcd

#endif // DENOISER_H
