// SyntheticScene.h -- seeded synthetic Monte-Carlo frames pushed through SamplesAccumulator (NOT in the reference:
// the reference ships no scene, data/inputs/.gitignore).  Used by bench.py and the CLI's --synthetic mode so that
// mean / covariance / histograms are mutually consistent.
#ifndef SYNTHETIC_SCENE_H
#define SYNTHETIC_SCENE_H

#include "SamplesAccumulator.h"

#include <cstdint>

namespace bcd
{

	struct SyntheticSceneParameters
	{
		SyntheticSceneParameters() : m_width(320), m_height(240), m_samplesPerPixel(32), m_seed(1234u), m_noiseSigma(0.35f), m_spikeProbability(0.01f), m_pattern(0) {}

		int m_width, m_height;
		int m_samplesPerPixel;
		uint32_t m_seed;
		float m_noiseSigma; ///< multiplicative gaussian noise on the base radiance
		float m_spikeProbability; ///< probability of an additive outlier per sample
		int m_pattern; ///< 0: ramps + 16-pixel checker (the SURVEY.md 8d probe scene); 1: band-limited texture with oblique soft edges
	};

	/// radiance model: smooth ramps + 16-pixel checker, sample = base * (1 + sigma * N(0,1)) [+ 4 U(0,1) spike], clamped >= 0.
	/// On the checker scene the similar sets of a 32-spp frame are decided by the checker geometry alone (pairs inside a cell are
	/// similar, pairs across a cell edge are not, whatever the noise level); pattern 1 is a texture in PIXEL units -- gradients of
	/// 0.002 .. 0.05 per pixel, soft oblique edges -- on which the similar sets do depend on the noise, like on a render.
	/// Counter-based RNG keyed on (seed, global line, column, sample): any band [i_firstLine, i_firstLine + i_nbOfLines) of the
	/// full frame can be generated independently and is identical to the same lines of a full-frame generation.
	SamplesStatisticsImages generateSyntheticScene(const SyntheticSceneParameters& i_rParams, int i_firstLine = 0, int i_nbOfLines = -1);

} // namespace bcd

#endif // SYNTHETIC_SCENE_H
