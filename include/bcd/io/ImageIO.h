// ImageIO.h -- EXR <-> DeepImage, API of the reference's include/bcd/io/ImageIO.h:22-38.
// OpenEXR is not available on the target boxes, so the codec underneath is this build's own minimal scanline
// reader/writer (bcd_amd/host/ImageIO.cpp): single-part scanline files, channels HALF / FLOAT / UINT without
// subsampling, compression NONE / RLE / ZIPS / ZIP / PIZ on reading (ZIP on writing).  (PXR24 / B44 / DWA, tiled, deep and multi-part
// files are reported as unsupported.)  File conventions of the reference (src/io/exr/io_exr.cpp):
//   colours     : channels R, G, B (read as float whatever their storage type); written as HALF A,B,G,R with A = 1
//   histograms  : FLOAT channels Bin_0000 ... Bin_{D}, the LAST one being the number of samples (src/core/Utils.cpp:21-45)
//   covariances : FLOAT channels Bin_0000 ... Bin_0005 = xx, yy, zz, yz, xz, xy
// The declarations below keep the names and member layout of the reference's include/bcd/io/ImageIO.h (BCD -- Bayesian Collaborative Denoising for
// Monte-Carlo Rendering, M. Boughida and T. Boubekeur, Computer Graphics Forum (Proc. EGSR 2017) 36(4); BSD-style licence, see the
// reference's LICENSE.txt) so that callers written against it compile unchanged; the implementation behind them is this build's own.
#ifndef IMAGE_IO_H
#define IMAGE_IO_H

#include <string>
#include <vector>

namespace bcd
{

	template<typename T> class DeepImage;

	class ImageIO
	{
	private:
		ImageIO() {}

	public:
		/// reads R, G, B (missing channels are 0); the image has depth 1 when R == G == B everywhere, 3 otherwise
		static bool loadEXR(DeepImage<float>& o_rImage, const char* i_pFilePath);
		/// reads every channel, in the file's (alphabetical) channel order
		static bool loadMultiChannelsEXR(DeepImage<float>& o_rImage, const char* i_pFilePath);

		/// writes a 1- or 3-channel image as HALF RGBA (a 1-channel image is replicated)
		static bool writeEXR(const DeepImage<float>& i_rImage, const char* i_pFilePath);
		/// writes FLOAT channels Bin_0000 ... Bin_{depth-1}
		static bool writeMultiChannelsEXR(const DeepImage<float>& i_rImage, const char* i_pFilePath);

		/// message of the last failure on this thread
		static const std::string& lastError();
	};

}

#endif // IMAGE_IO_H
