// SyntheticScene.cpp -- seeded synthetic frames for benchmarks and tests (see SyntheticScene.h).
#include "SyntheticScene.h"

#include <cmath>

namespace bcd
{

	namespace
	{
		inline uint64_t splitmix(uint64_t x)
		{
			x += 0x9E3779B97F4A7C15ull;
			x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
			x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
			return x ^ (x >> 31);
		}
		inline float unit(uint64_t bits) { return float((bits >> 40) + 1) * (1.0f / 16777217.0f); } // (0,1)
	}

	SamplesStatisticsImages generateSyntheticScene(const SyntheticSceneParameters& i_rParams, int i_firstLine, int i_nbOfLines)
	{
		const int W = i_rParams.m_width, H = i_rParams.m_height;
		if(i_nbOfLines < 0)
			i_nbOfLines = H - i_firstLine;
		SamplesAccumulator accumulator(W, i_nbOfLines, HistogramParameters());
		const float twoPi = 6.28318530717958647692f;
#pragma omp parallel for schedule(static)
		for(int localLine = 0; localLine < i_nbOfLines; ++localLine)
		{
			const int line = i_firstLine + localLine;
			for(int column = 0; column < W; ++column)
			{
				float base[3];
				if(i_rParams.m_pattern == 1)
				{
					const float x = float(column), y = float(line);
					// soft oblique edges: a smooth step across lines of slope ~ 0.37, every ~90 pixels
					const float u = std::sin(0.0698f * (x * 0.9397f + y * 0.3420f));
					const float edge = 0.5f + 0.5f * std::tanh(6.f * u);
					base[0] = 0.45f + 0.25f * std::sin(0.11f * x + 0.05f * y) * std::cos(0.07f * y - 0.02f * x) + 0.12f * std::sin(0.31f * x - 0.23f * y);
					base[1] = 0.50f + 0.30f * std::sin(0.045f * (x + y)) * std::sin(0.013f * (x - 2.f * y)) + 0.10f * edge;
					base[2] = 0.20f + 0.55f * edge + 0.08f * std::cos(0.19f * x + 0.27f * y);
				}
				else
				{
					const bool checker = ((line / 16 + column / 16) % 2) != 0;
					base[0] = 0.2f + 0.6f * float(column) / float(W);
					base[1] = 0.5f + 0.4f * std::sin(12.f * float(line) / float(H));
					base[2] = checker ? 0.8f : 0.15f;
				}
				for(int s = 0; s < i_rParams.m_samplesPerPixel; ++s)
				{
					uint64_t state = splitmix((uint64_t(i_rParams.m_seed) << 32) ^ (uint64_t(line) * uint64_t(W) + uint64_t(column)));
					state = splitmix(state ^ (uint64_t(s) * 0xD6E8FEB86659FD93ull));
					float rgb[3];
					for(int ch = 0; ch < 3; ch += 2)
					{	// Box-Muller, two normals per pair of uniforms
						const uint64_t a = splitmix(state + 2 * ch), b = splitmix(state + 2 * ch + 1);
						const float radius = std::sqrt(-2.f * std::log(unit(a))), angle = twoPi * unit(b);
						rgb[ch] = radius * std::cos(angle);
						if(ch + 1 < 3)
							rgb[ch + 1] = radius * std::sin(angle);
					}
					const uint64_t spikeBits = splitmix(state + 11);
					const bool spike = unit(spikeBits) < i_rParams.m_spikeProbability;
					for(int ch = 0; ch < 3; ++ch)
					{
						float v = base[ch] * (1.f + i_rParams.m_noiseSigma * rgb[ch]);
						if(spike)
							v += 4.f * unit(splitmix(state + 12 + ch));
						rgb[ch] = v > 0.f ? v : 0.f;
					}
					accumulator.addSample(localLine, column, rgb[0], rgb[1], rgb[2], 1.f);
				}
			}
		}
		return accumulator.extractSamplesStatistics();
	}

} // namespace bcd
