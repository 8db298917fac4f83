// Utils.cpp -- histogram/sample-count channel packing used by the EXR layout, and path helpers
// (behaviour of the reference's src/core/Utils.cpp:21-115).
#include "Utils.h"
#include "DeepImage.h"

#include <cassert>
#include <cstring>

using namespace std;

namespace bcd
{

	bool Utils::separateNbOfSamplesFromHistogram(Deepimf& o_rHistoImage, Deepimf& o_rNbOfSamplesImage, const Deepimf& i_rHistoAndNbOfSamplesImage)
	{
		const int w = i_rHistoAndNbOfSamplesImage.getWidth(), h = i_rHistoAndNbOfSamplesImage.getHeight();
		const int d = i_rHistoAndNbOfSamplesImage.getDepth() - 1;
		if(d < 1)
			return false;
		o_rHistoImage.resize(w, h, d);
		o_rNbOfSamplesImage.resize(w, h, 1);
		const float* pSrc = i_rHistoAndNbOfSamplesImage.getDataPtr();
		float* pHisto = o_rHistoImage.getDataPtr();
		float* pNb = o_rNbOfSamplesImage.getDataPtr();
		for(size_t pixel = 0, n = size_t(w) * h; pixel < n; ++pixel, pSrc += d + 1, pHisto += d)
		{
			memcpy(pHisto, pSrc, d * sizeof(float));
			pNb[pixel] = pSrc[d];
		}
		return true;
	}

	Deepimf Utils::mergeHistogramAndNbOfSamples(const Deepimf& i_rHistoImage, const Deepimf& i_rNbOfSamplesImage)
	{
		const int w = i_rHistoImage.getWidth(), h = i_rHistoImage.getHeight(), d = i_rHistoImage.getDepth();
		assert(i_rNbOfSamplesImage.getWidth() == w && i_rNbOfSamplesImage.getHeight() == h && i_rNbOfSamplesImage.getDepth() == 1);
		Deepimf merged(w, h, d + 1);
		const float* pHisto = i_rHistoImage.getDataPtr();
		const float* pNb = i_rNbOfSamplesImage.getDataPtr();
		float* pDst = merged.getDataPtr();
		for(size_t pixel = 0, n = size_t(w) * h; pixel < n; ++pixel, pHisto += d, pDst += d + 1)
		{
			memcpy(pDst, pHisto, d * sizeof(float));
			pDst[d] = pNb[pixel];
		}
		return merged;
	}

	string Utils::extractFolderPath(const string& i_rFilePath)
	{
		const size_t pos = i_rFilePath.find_last_of("/\\");
		return pos == string::npos ? string() : i_rFilePath.substr(0, pos + 1);
	}

	string Utils::getRelativePathFromFolder(const string& i_rFileAbsolutePath, const string& i_rFolderAbsolutePath)
	{
		// drop the common leading folders, then climb out of what is left of the folder path
		size_t common = 0;
		for(size_t i = 0; i < i_rFileAbsolutePath.size() && i < i_rFolderAbsolutePath.size() && i_rFileAbsolutePath[i] == i_rFolderAbsolutePath[i]; ++i)
			if(i_rFileAbsolutePath[i] == '/' || i_rFileAbsolutePath[i] == '\\')
				common = i + 1;
		string relative;
		for(size_t i = common; i < i_rFolderAbsolutePath.size(); ++i)
			if(i_rFolderAbsolutePath[i] == '/' || i_rFolderAbsolutePath[i] == '\\')
				relative += "../";
		return relative + i_rFileAbsolutePath.substr(common);
	}

} // namespace bcd
