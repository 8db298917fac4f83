// MultiscaleDenoiser.h -- multiscale driver (pyramid, per-scale denoising, merge), MI355X build.
// Public surface of the reference's include/bcd/core/MultiscaleDenoiser.h:23-31: an IDenoiser, like the reference's.
#ifndef MULTISCALE_DENOISER_H
#define MULTISCALE_DENOISER_H

#include "Denoiser.h"

namespace bcd
{

	class MultiscaleDenoiser : public IDenoiser, public HipEngineSettings
	{
	public:
		MultiscaleDenoiser(int i_nbOfScales) : IDenoiser(), m_nbOfScales(i_nbOfScales) {}
		virtual ~MultiscaleDenoiser() {}

	public:
		virtual bool denoise();

	private:
		int m_nbOfScales;
	};

} // namespace bcd

#endif // MULTISCALE_DENOISER_H
