// ImageIO.cpp -- minimal OpenEXR scanline codec (own implementation over zlib) behind the reference's ImageIO API.
// Format notes follow the published OpenEXR file layout: magic 20000630, version 2, attribute list, line offset
// table, chunks of 1 (NONE/RLE/ZIPS) or 16 (ZIP) scanlines, channels stored alphabetically, scanline-planar.
#include "ImageIO.h"
#include "DeepImage.h"

#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>

using namespace std;

namespace bcd
{

	namespace
	{
		thread_local string g_lastError;

		bool fail(const string& msg)
		{
			g_lastError = msg;
			return false;
		}

		enum PixelType { e_uint = 0, e_half = 1, e_float = 2 };
		enum Compression { e_none = 0, e_rle = 1, e_zips = 2, e_zip = 3 };

		struct Channel
		{
			string name;
			int type;
		};

		float halfToFloat(uint16_t h)
		{
			const uint32_t sign = uint32_t(h & 0x8000u) << 16;
			uint32_t exponent = (h >> 10) & 0x1f, mantissa = h & 0x3ffu, bits;
			if(exponent == 0)
			{
				if(mantissa == 0)
					bits = sign;
				else
				{	// subnormal half -> normal float
					exponent = 127 - 15 + 1;
					while(!(mantissa & 0x400u)) { mantissa <<= 1; --exponent; }
					bits = sign | (exponent << 23) | ((mantissa & 0x3ffu) << 13);
				}
			}
			else if(exponent == 31)
				bits = sign | 0x7f800000u | (mantissa << 13);
			else
				bits = sign | ((exponent + 127 - 15) << 23) | (mantissa << 13);
			float f;
			memcpy(&f, &bits, 4);
			return f;
		}

		uint16_t floatToHalf(float f)
		{	// round to nearest even
			uint32_t bits;
			memcpy(&bits, &f, 4);
			const uint16_t sign = uint16_t((bits >> 16) & 0x8000u);
			const uint32_t absBits = bits & 0x7fffffffu;
			if(absBits >= 0x7f800000u) // inf / nan
				return uint16_t(sign | 0x7c00u | (absBits > 0x7f800000u ? 0x200u : 0u));
			if(absBits >= 0x477ff000u) // rounds to >= 65520 -> inf
				return uint16_t(sign | 0x7c00u);
			if(absBits < 0x33000001u) // < half of the smallest subnormal
				return sign;
			int exponent = int(absBits >> 23) - 127 + 15;
			uint32_t mantissa = (absBits & 0x7fffffu) | 0x800000u;
			int shift = 13;
			if(exponent <= 0) { shift += 1 - exponent; exponent = 0; }
			uint32_t halfMantissa = mantissa >> shift;
			const uint32_t remainder = mantissa & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
			if(remainder > halfway || (remainder == halfway && (halfMantissa & 1u)))
				++halfMantissa;
			// the implicit bit (0x400) carries into the exponent field naturally
			uint32_t result = exponent == 0 ? halfMantissa : (uint32_t(exponent - 1) << 10) + halfMantissa;
			return uint16_t(sign | result);
		}

		// ---- byte predictor + even/odd reordering shared by RLE and ZIP -----------------------------------
		void undoPredictorAndReorder(vector<unsigned char>& io_rData, vector<unsigned char>& tmp)
		{
			const size_t n = io_rData.size();
			for(size_t i = 1; i < n; ++i)
				io_rData[i] = (unsigned char)(io_rData[i - 1] + io_rData[i] - 128);
			tmp.resize(n);
			const size_t half = (n + 1) / 2;
			for(size_t i = 0; i < n; ++i)
				tmp[i] = (i & 1) ? io_rData[half + i / 2] : io_rData[i / 2];
			io_rData.swap(tmp);
		}

		void reorderAndPredict(const unsigned char* i_pRaw, size_t n, vector<unsigned char>& o_rOut)
		{
			o_rOut.resize(n);
			const size_t half = (n + 1) / 2;
			for(size_t i = 0; i < n; ++i)
				o_rOut[(i & 1) ? half + i / 2 : i / 2] = i_pRaw[i];
			unsigned char previous = n ? o_rOut[0] : 0;
			for(size_t i = 1; i < n; ++i)
			{
				const unsigned char current = o_rOut[i];
				o_rOut[i] = (unsigned char)(current - previous + 128);
				previous = current;
			}
		}

		bool rleDecode(const unsigned char* in, size_t inSize, vector<unsigned char>& out, size_t expected)
		{
			out.clear();
			out.reserve(expected);
			size_t i = 0;
			while(i < inSize)
			{
				const int count = (signed char)in[i++];
				if(count < 0)
				{
					const size_t n = size_t(-count);
					if(i + n > inSize) return false;
					out.insert(out.end(), in + i, in + i + n);
					i += n;
				}
				else
				{
					if(i >= inSize) return false;
					out.insert(out.end(), size_t(count) + 1, in[i++]);
				}
			}
			return out.size() == expected;
		}

		struct Reader
		{
			const vector<unsigned char>& d;
			size_t pos;
			bool ok;
			explicit Reader(const vector<unsigned char>& data) : d(data), pos(0), ok(true) {}
			template<typename T> T get()
			{
				T v = T();
				if(pos + sizeof(T) > d.size()) { ok = false; return v; }
				memcpy(&v, &d[pos], sizeof(T));
				pos += sizeof(T);
				return v;
			}
			string str()
			{
				string s;
				while(pos < d.size() && d[pos]) s.push_back(char(d[pos++]));
				if(pos >= d.size()) ok = false;
				++pos;
				return s;
			}
		};

		/// planes[c] = W*H floats of channel c (channels in file order)
		bool readExr(const char* path, int& W, int& H, vector<Channel>& channels, vector< vector<float> >& planes)
		{
			FILE* f = fopen(path, "rb");
			if(!f) return fail(string("cannot open '") + path + "'");
			vector<unsigned char> data;
			fseek(f, 0, SEEK_END);
			const long size = ftell(f);
			fseek(f, 0, SEEK_SET);
			data.resize(size > 0 ? size_t(size) : 0);
			const size_t got = data.empty() ? 0 : fread(data.data(), 1, data.size(), f);
			fclose(f);
			if(got != data.size() || data.size() < 8) return fail(string("cannot read '") + path + "'");

			Reader r(data);
			if(r.get<uint32_t>() != 20000630u) return fail(string("'") + path + "' is not an OpenEXR file");
			const uint32_t version = r.get<uint32_t>();
			if((version & 0xff) != 2 || (version & 0x1a00)) return fail("unsupported EXR flavour (tiled, deep or multi-part)");
			int compression = -1, minX = 0, minY = 0, maxX = -1, maxY = -1;
			channels.clear();
			while(r.ok)
			{
				const string name = r.str();
				if(name.empty()) break;
				const string type = r.str();
				const int32_t attrSize = r.get<int32_t>();
				const size_t next = r.pos + size_t(attrSize);
				if(!r.ok || attrSize < 0 || next > data.size()) return fail("corrupt EXR header");
				if(name == "channels")
				{
					while(r.ok && r.pos < next)
					{
						Channel c;
						c.name = r.str();
						if(c.name.empty()) break;
						c.type = r.get<int32_t>();
						r.get<uint32_t>(); // pLinear + reserved
						const int xs = r.get<int32_t>(), ys = r.get<int32_t>();
						if(xs != 1 || ys != 1) return fail("subsampled EXR channels are not supported");
						if(c.type < 0 || c.type > 2) return fail("unknown EXR pixel type");
						channels.push_back(c);
					}
				}
				else if(name == "compression")
					compression = r.get<unsigned char>();
				else if(name == "dataWindow")
				{
					minX = r.get<int32_t>(); minY = r.get<int32_t>(); maxX = r.get<int32_t>(); maxY = r.get<int32_t>();
				}
				r.pos = next;
			}
			if(!r.ok || channels.empty() || maxX < minX || maxY < minY) return fail("incomplete EXR header");
			if(compression < e_none || compression > e_zip)
				return fail("unsupported EXR compression (only NONE, RLE, ZIPS and ZIP are implemented; re-save PIZ/B44/DWA files)");
			W = maxX - minX + 1;
			H = maxY - minY + 1;
			const int linesPerBlock = compression == e_zip ? 16 : 1;
			const int nbOfBlocks = (H + linesPerBlock - 1) / linesPerBlock;
			size_t bytesPerLine = 0;
			for(const Channel& c : channels) bytesPerLine += size_t(c.type == e_half ? 2 : 4) * W;
			vector<uint64_t> offsets(nbOfBlocks);
			for(int b = 0; b < nbOfBlocks; ++b) offsets[b] = r.get<uint64_t>();
			if(!r.ok) return fail("truncated EXR offset table");

			planes.assign(channels.size(), vector<float>(size_t(W) * H, 0.f));
			vector<unsigned char> raw, tmp;
			for(int b = 0; b < nbOfBlocks; ++b)
			{
				if(offsets[b] + 8 > data.size()) return fail("EXR chunk offset out of range");
				r.pos = size_t(offsets[b]);
				const int y = r.get<int32_t>() - minY;
				const int32_t chunkSize = r.get<int32_t>();
				if(!r.ok || chunkSize < 0 || r.pos + size_t(chunkSize) > data.size() || y < 0 || y >= H) return fail("corrupt EXR chunk");
				const int lines = min(linesPerBlock, H - y);
				const size_t expected = bytesPerLine * lines;
				const unsigned char* src = &data[r.pos];
				if(size_t(chunkSize) == expected || compression == e_none)
				{
					if(size_t(chunkSize) != expected) return fail("EXR chunk has an unexpected size");
					raw.assign(src, src + expected);
				}
				else if(compression == e_rle)
				{
					if(!rleDecode(src, size_t(chunkSize), raw, expected)) return fail("corrupt RLE data in EXR chunk");
					undoPredictorAndReorder(raw, tmp);
				}
				else
				{
					raw.resize(expected);
					uLongf outSize = uLongf(expected);
					if(uncompress(raw.data(), &outSize, src, uLong(chunkSize)) != Z_OK || outSize != expected) return fail("corrupt ZIP data in EXR chunk");
					undoPredictorAndReorder(raw, tmp);
				}
				const unsigned char* p = raw.data();
				for(int l = 0; l < lines; ++l)
					for(size_t c = 0; c < channels.size(); ++c)
					{
						float* dst = &planes[c][size_t(y + l) * W];
						if(channels[c].type == e_half)
							for(int x = 0; x < W; ++x, p += 2) { uint16_t h; memcpy(&h, p, 2); dst[x] = halfToFloat(h); }
						else if(channels[c].type == e_float)
							for(int x = 0; x < W; ++x, p += 4) memcpy(&dst[x], p, 4);
						else
							for(int x = 0; x < W; ++x, p += 4) { uint32_t u; memcpy(&u, p, 4); dst[x] = float(u); }
					}
			}
			return true;
		}

		void putString(vector<unsigned char>& o, const string& s) { o.insert(o.end(), s.begin(), s.end()); o.push_back(0); }
		template<typename T> void put(vector<unsigned char>& o, T v) { const unsigned char* p = reinterpret_cast<const unsigned char*>(&v); o.insert(o.end(), p, p + sizeof(T)); }
		void putAttribute(vector<unsigned char>& o, const string& name, const string& type, const vector<unsigned char>& value)
		{
			putString(o, name); putString(o, type); put<int32_t>(o, int32_t(value.size()));
			o.insert(o.end(), value.begin(), value.end());
		}

		/// interleaved source: value of channel c at pixel i is i_pPixels[i * stride + offsets[c]] (offset < 0: constant 1)
		bool writeExr(const char* path, int W, int H, const vector<string>& names, int pixelType, const float* i_pPixels, int stride, const vector<int>& channelOffsets)
		{
			vector<size_t> order(names.size());
			for(size_t i = 0; i < order.size(); ++i) order[i] = i;
			sort(order.begin(), order.end(), [&](size_t a, size_t b) { return names[a] < names[b]; }); // channels are stored alphabetically

			vector<unsigned char> out;
			put<uint32_t>(out, 20000630u);
			put<uint32_t>(out, 2u);
			vector<unsigned char> v;
			for(size_t k : order)
			{
				putString(v, names[k]); put<int32_t>(v, pixelType); put<uint32_t>(v, 0u); put<int32_t>(v, 1); put<int32_t>(v, 1);
			}
			v.push_back(0);
			putAttribute(out, "channels", "chlist", v);
			v.assign(1, (unsigned char)e_zip);
			putAttribute(out, "compression", "compression", v);
			v.clear(); put<int32_t>(v, 0); put<int32_t>(v, 0); put<int32_t>(v, W - 1); put<int32_t>(v, H - 1);
			putAttribute(out, "dataWindow", "box2i", v);
			putAttribute(out, "displayWindow", "box2i", v);
			v.assign(1, 0);
			putAttribute(out, "lineOrder", "lineOrder", v);
			v.clear(); put<float>(v, 1.f);
			putAttribute(out, "pixelAspectRatio", "float", v);
			v.clear(); put<float>(v, 0.f); put<float>(v, 0.f);
			putAttribute(out, "screenWindowCenter", "v2f", v);
			v.clear(); put<float>(v, 1.f);
			putAttribute(out, "screenWindowWidth", "float", v);
			out.push_back(0);

			const int linesPerBlock = 16, nbOfBlocks = (H + linesPerBlock - 1) / linesPerBlock;
			const size_t tablePos = out.size();
			out.resize(out.size() + size_t(nbOfBlocks) * 8);
			const size_t bytesPerPixel = pixelType == e_half ? 2 : 4, bytesPerLine = bytesPerPixel * names.size() * W;
			vector<unsigned char> raw, shuffled, packed;
			for(int b = 0; b < nbOfBlocks; ++b)
			{
				const int y = b * linesPerBlock, lines = min(linesPerBlock, H - y);
				raw.resize(bytesPerLine * lines);
				unsigned char* p = raw.data();
				for(int l = 0; l < lines; ++l)
					for(size_t k : order)
						for(int x = 0; x < W; ++x, p += bytesPerPixel)
						{
							const float value = channelOffsets[k] < 0 ? 1.f : i_pPixels[(size_t(y + l) * W + x) * stride + channelOffsets[k]];
							if(pixelType == e_half) { const uint16_t h = floatToHalf(value); memcpy(p, &h, 2); }
							else memcpy(p, &value, 4);
						}
				reorderAndPredict(raw.data(), raw.size(), shuffled);
				uLongf packedSize = compressBound(uLong(shuffled.size()));
				packed.resize(packedSize);
				if(compress2(packed.data(), &packedSize, shuffled.data(), uLong(shuffled.size()), Z_DEFAULT_COMPRESSION) != Z_OK) return fail("zlib compression failed");
				const uint64_t offset = out.size();
				memcpy(&out[tablePos + size_t(b) * 8], &offset, 8);
				put<int32_t>(out, y);
				if(packedSize < raw.size())
				{
					put<int32_t>(out, int32_t(packedSize));
					out.insert(out.end(), packed.begin(), packed.begin() + packedSize);
				}
				else
				{	// incompressible block: stored raw
					put<int32_t>(out, int32_t(raw.size()));
					out.insert(out.end(), raw.begin(), raw.end());
				}
			}
			FILE* f = fopen(path, "wb");
			if(!f) return fail(string("cannot create '") + path + "'");
			const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
			fclose(f);
			return ok ? true : fail(string("cannot write '") + path + "'");
		}
	}

	const string& ImageIO::lastError() { return g_lastError; }

	bool ImageIO::loadEXR(DeepImage<float>& o_rImage, const char* i_pFilePath)
	{
		cout << "Loading " << i_pFilePath << endl;
		int w = 0, h = 0;
		vector<Channel> channels;
		vector< vector<float> > planes;
		if(!readExr(i_pFilePath, w, h, channels, planes))
		{
			cerr << "error :: '" << i_pFilePath << "' not found  or not a correct exr image (" << g_lastError << ")" << endl;
			return false;
		}
		const vector<float>* rgb[3] = { nullptr, nullptr, nullptr };
		for(size_t c = 0; c < channels.size(); ++c)
			for(int k = 0; k < 3; ++k)
				if(channels[c].name == string(1, "RGB"[k])) rgb[k] = &planes[c];
		const size_t n = size_t(w) * h;
		int depth = 1; // grey image stored as three identical channels -> depth 1 (src/io/ImageIO.cpp:41-49 of the reference)
		for(size_t i = 0; i < n && depth == 1; ++i)
		{
			const float r = rgb[0] ? (*rgb[0])[i] : 0.f, g = rgb[1] ? (*rgb[1])[i] : 0.f, b = rgb[2] ? (*rgb[2])[i] : 0.f;
			if(r != g || r != b) depth = 3;
		}
		o_rImage.resize(w, h, depth);
		float* dst = o_rImage.getDataPtr();
		for(size_t i = 0; i < n; ++i)
			for(int z = 0; z < depth; ++z)
				dst[i * depth + z] = rgb[z] ? (*rgb[z])[i] : 0.f;
		return true;
	}

	bool ImageIO::loadMultiChannelsEXR(DeepImage<float>& o_rImage, const char* i_pFilePath)
	{
		cout << "Loading " << i_pFilePath << endl;
		int w = 0, h = 0;
		vector<Channel> channels;
		vector< vector<float> > planes;
		if(!readExr(i_pFilePath, w, h, channels, planes))
		{
			cerr << "error :: '" << i_pFilePath << "' not found  or not a correct exr image (" << g_lastError << ")" << endl;
			return false;
		}
		const int depth = int(channels.size());
		o_rImage.resize(w, h, depth);
		float* dst = o_rImage.getDataPtr();
		for(size_t i = 0, n = size_t(w) * h; i < n; ++i)
			for(int z = 0; z < depth; ++z)
				dst[i * depth + z] = planes[z][i];
		return true;
	}

	bool ImageIO::writeEXR(const DeepImage<float>& i_rImage, const char* i_pFilePath)
	{
		const int depth = i_rImage.getDepth();
		if(depth != 1 && depth != 3) return fail("writeEXR expects a 1- or 3-channel image");
		const vector<string> names = { "R", "G", "B", "A" };
		const vector<int> offsets = { 0, depth == 1 ? 0 : 1, depth == 1 ? 0 : 2, -1 };
		return writeExr(i_pFilePath, i_rImage.getWidth(), i_rImage.getHeight(), names, e_half, i_rImage.getDataPtr(), depth, offsets);
	}

	bool ImageIO::writeMultiChannelsEXR(const DeepImage<float>& i_rImage, const char* i_pFilePath)
	{
		const int depth = i_rImage.getDepth();
		vector<string> names(depth);
		vector<int> offsets(depth);
		for(int z = 0; z < depth; ++z)
		{
			char name[16];
			snprintf(name, sizeof(name), "Bin_%04d", z);
			names[z] = name;
			offsets[z] = z;
		}
		return writeExr(i_pFilePath, i_rImage.getWidth(), i_rImage.getHeight(), names, e_float, i_rImage.getDataPtr(), depth, offsets);
	}

} // namespace bcd
