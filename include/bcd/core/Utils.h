// Utils.h -- helpers around the "histogram + number of samples as last channel" file layout and paths;
// API of the reference's include/bcd/core/Utils.h.
#ifndef UTILS_H
#define UTILS_H

#include <string>

namespace bcd
{

	template<typename T> class DeepImage;

	class Utils
	{
	public:
		/// splits a W x H x (D+1) image into the W x H x D histogram and the W x H x 1 sample count (last channel)
		static bool separateNbOfSamplesFromHistogram(
				DeepImage<float>& o_rHistoImage,
				DeepImage<float>& o_rNbOfSamplesImage,
				const DeepImage<float>& i_rHistoAndNbOfSamplesImage);

		static DeepImage<float> mergeHistogramAndNbOfSamples(
				const DeepImage<float>& i_rHistoImage,
				const DeepImage<float>& i_rNbOfSamplesImage);

		static std::string extractFolderPath(const std::string& i_rFilePath);
		static std::string getRelativePathFromFolder(const std::string& i_rFileAbsolutePath, const std::string& i_rFolderAbsolutePath);
	};

} // namespace bcd

#endif // UTILS_H
