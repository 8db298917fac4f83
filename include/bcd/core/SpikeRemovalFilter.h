// SpikeRemovalFilter.h -- optional outlier prefilter (bcd_cli -p); API of the reference's
// include/bcd/core/SpikeRemovalFilter.h.  Runs as a HIP kernel (bcd_hip_spike_filter); without a usable device, as host loops.
#ifndef SPIKE_REMOVAL_FILTER_H
#define SPIKE_REMOVAL_FILTER_H

namespace bcd
{

	template<typename T> class DeepImage;

	class SpikeRemovalFilter
	{
	public:
		/// replaces, in place, every pixel whose colour deviates from its 3x3 neighbourhood mean by more than
		/// i_thresholdStDevFactor standard deviations (any channel) by the neighbour of median colour
		static void filter(
				DeepImage<float>& io_rInputColorImage,
				DeepImage<float>& io_rInputNbOfSamplesImage,
				DeepImage<float>& io_rInputHistogramImage,
				DeepImage<float>& io_rInputCovImage,
				float i_thresholdStDevFactor = 2.f);

		/// the same on a chosen HIP device, with a result: false = no usable device or a device error (the images are then
		/// UNCHANGED -- results are committed only after all four came back -- and a message is on cerr unless i_quiet).  filter() is this on device 0 with the reference's void
		/// signature, falling back to filterOnHost() on the untouched images when that returns false
		static bool filterOnDevice(
				int i_device,
				DeepImage<float>& io_rInputColorImage,
				DeepImage<float>& io_rInputNbOfSamplesImage,
				DeepImage<float>& io_rInputHistogramImage,
				DeepImage<float>& io_rInputCovImage,
				float i_thresholdStDevFactor = 2.f,
				bool i_quiet = false);

		/// the same as host loops (OpenMP over the lines): what filter() runs when no HIP device is usable
		static void filterOnHost(
				DeepImage<float>& io_rInputColorImage,
				DeepImage<float>& io_rInputNbOfSamplesImage,
				DeepImage<float>& io_rInputHistogramImage,
				DeepImage<float>& io_rInputCovImage,
				float i_thresholdStDevFactor = 2.f);
	};

} // namespace bcd

#endif // SPIKE_REMOVAL_FILTER_H
