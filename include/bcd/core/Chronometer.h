// Chronometer.h -- wall-clock stopwatch used by the CLI (API subset of the reference's include/bcd/core/Chronometer.h).
#ifndef CHRONOMETER_H
#define CHRONOMETER_H

#include <chrono>
#include <iostream>
#include <string>

namespace bcd
{

	class Chronometer
	{
	public:
		Chronometer() : m_elapsed(0.f), m_running(false) {}
		void reset() { m_elapsed = 0.f; m_running = false; }
		void start() { m_start = std::chrono::high_resolution_clock::now(); m_running = true; }
		void stop() { if(m_running) { m_elapsed += std::chrono::duration<float>(std::chrono::high_resolution_clock::now() - m_start).count(); m_running = false; } }
		float getElapsedTime() const
		{
			return m_running ? m_elapsed + std::chrono::duration<float>(std::chrono::high_resolution_clock::now() - m_start).count() : m_elapsed;
		}
		static std::string getStringFromTime(float i_seconds)
		{
			int h = int(i_seconds / 3600.f), m = int(i_seconds / 60.f) % 60;
			float s = i_seconds - 3600.f * h - 60.f * m;
			return std::to_string(h) + " h " + std::to_string(m) + " min " + std::to_string(s) + " s";
		}
		void printElapsedTime() const { std::cout << getStringFromTime(getElapsedTime()); }

	private:
		std::chrono::high_resolution_clock::time_point m_start;
		float m_elapsed;
		bool m_running;
	};

} // namespace bcd

#endif // CHRONOMETER_H
