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
#include <new>
#include <stdexcept>

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
		enum Compression { e_none = 0, e_rle = 1, e_zips = 2, e_zip = 3, e_piz = 4 };

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

		// ---- PIZ (read only): per chunk of 32 scanlines  [u16 minNonZero][u16 maxNonZero][bitmap bytes min..max][i32 length][Huffman data];
		// the 16-bit words of every channel are mapped through a lookup table of the values in use, Haar-wavelet transformed in
		// place and Huffman coded with a run-length pseudo symbol (the published OpenEXR PIZ scheme).  The reference writes its
		// colour images through Imf::RgbaOutputFile (src/io/exr/io_exr.cpp:147-164), whose default compression is PIZ.
		namespace piz
		{
			const int c_encBits = 16, c_encSize = (1 << c_encBits) + 1;
			const int c_shortZeroRun = 59, c_longZeroRun = 63, c_shortestLongRun = 2 + c_longZeroRun - c_shortZeroRun;
			const int c_maxCodeLength = 58;

			struct BitReader
			{
				const unsigned char* p;
				const unsigned char* end;
				uint64_t acc;
				int nb;
				bool ok;
				BitReader(const unsigned char* b, const unsigned char* e) : p(b), end(e), acc(0), nb(0), ok(true) {}
				uint32_t get(int n) // MSB first, n <= 32
				{
					while(nb < n)
					{
						if(p >= end) { ok = false; return 0; }
						acc = (acc << 8) | *p++;
						nb += 8;
					}
					nb -= n;
					return uint32_t((acc >> nb) & ((uint64_t(1) << n) - 1));
				}
			};

			/// canonical prefix code defined by the code lengths: the longest codes start at 0, codes of equal length follow the symbol order
			struct Decoder
			{
				uint64_t first[c_maxCodeLength + 2]; // first code of each length
				uint32_t count[c_maxCodeLength + 2];
				uint32_t offset[c_maxCodeLength + 2]; // into `symbols`
				vector<uint32_t> symbols;             // sorted by (length, symbol)
				vector<uint32_t> fast;                // 12-bit prefix -> (symbol << 6 | length), 0 = longer code
				bool build(const vector<unsigned char>& lengths)
				{
					memset(count, 0, sizeof(count));
					for(unsigned char l : lengths) { if(l > c_maxCodeLength) return false; ++count[l]; }
					uint64_t c = 0;
					for(int l = c_maxCodeLength; l >= 1; --l)
					{
						const uint64_t nc = (c + count[l]) >> 1;
						first[l] = c;
						c = nc;
					}
					uint32_t o = 0;
					for(int l = 1; l <= c_maxCodeLength; ++l) { offset[l] = o; o += count[l]; }
					symbols.assign(o, 0);
					vector<uint32_t> fill(offset, offset + c_maxCodeLength + 2);
					for(size_t sym = 0; sym < lengths.size(); ++sym)
						if(lengths[sym]) symbols[fill[lengths[sym]]++] = uint32_t(sym);
					fast.assign(1 << 12, 0);
					for(int l = 1; l <= 12; ++l)
						for(uint32_t k = 0; k < count[l]; ++k)
						{
							const uint64_t code = first[l] + k;
							if(code >> l) return false; // lengths that are no prefix code
							const uint32_t entry = (symbols[offset[l] + k] << 6) | uint32_t(l);
							for(uint32_t pad = 0; pad < (1u << (12 - l)); ++pad) fast[(size_t(code) << (12 - l)) | pad] = entry;
						}
					return true;
				}
			};

			bool huffmanDecode(const unsigned char* in, size_t inSize, vector<uint16_t>& out, size_t nbOfValues)
			{
				out.assign(nbOfValues, 0);
				if(inSize == 0) return nbOfValues == 0;
				if(inSize < 20) return false;
				uint32_t im, iM, nBits;
				memcpy(&im, in, 4); memcpy(&iM, in + 4, 4); memcpy(&nBits, in + 12, 4);
				if(im >= uint32_t(c_encSize) || iM >= uint32_t(c_encSize) || im > iM) return false;
				// packed table of code lengths (6 bits each, with zero-run escapes)
				vector<unsigned char> lengths(c_encSize, 0);
				BitReader tr(in + 20, in + inSize);
				for(uint32_t i = im; i <= iM; ++i)
				{
					const uint32_t l = tr.get(6);
					if(!tr.ok) return false;
					if(l == uint32_t(c_longZeroRun) || l >= uint32_t(c_shortZeroRun))
					{
						uint32_t run = l == uint32_t(c_longZeroRun) ? tr.get(8) + c_shortestLongRun : l - c_shortZeroRun + 2;
						if(!tr.ok || i + run > iM + 1) return false;
						i += run - 1; // (those lengths stay 0)
					}
					else
						lengths[i] = (unsigned char)l;
				}
				const unsigned char* data = tr.p; // the table is padded to a whole byte
				if(uint64_t(nBits) > uint64_t(in + inSize - data) * 8) return false;
				Decoder dec;
				if(!dec.build(lengths)) return false;
				BitReader br(data, in + inSize);
				uint64_t remaining = nBits;
				size_t o = 0;
				while(remaining > 0)
				{
					uint32_t sym = 0;
					int len = 0;
					// fast path: the next 12 bits (zero-extended at the end of the stream) select short codes directly
					const int peek = int(min<uint64_t>(12, remaining));
					while(br.nb < peek)
					{
						if(br.p >= br.end) return false;
						br.acc = (br.acc << 8) | *br.p++;
						br.nb += 8;
					}
					const uint32_t window = uint32_t((br.acc >> (br.nb - peek)) & ((1u << peek) - 1)) << (12 - peek);
					const uint32_t e = dec.fast[window];
					if(e && int(e & 63) <= peek)
					{
						len = int(e & 63);
						sym = e >> 6;
						br.nb -= len;
					}
					else
					{
						uint64_t code = 0;
						bool found = false;
						while(len < c_maxCodeLength && uint64_t(len) < remaining)
						{
							code = (code << 1) | br.get(1);
							if(!br.ok) return false;
							++len;
							if(dec.count[len] && code >= dec.first[len] && code - dec.first[len] < dec.count[len])
							{
								sym = dec.symbols[dec.offset[len] + uint32_t(code - dec.first[len])];
								found = true;
								break;
							}
						}
						if(!found) return false;
					}
					remaining -= uint64_t(len);
					if(sym == iM)
					{	// run-length pseudo symbol: repeat the previous value
						if(remaining < 8 || o == 0) return false;
						uint32_t run = br.get(8);
						if(!br.ok) return false;
						remaining -= 8;
						if(o + run > nbOfValues) return false;
						const uint16_t v = out[o - 1];
						while(run--) out[o++] = v;
					}
					else
					{
						if(o >= nbOfValues || sym > 0xffffu) return false;
						out[o++] = uint16_t(sym);
					}
				}
				return o == nbOfValues;
			}

			inline void wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b)
			{
				const int ls = int16_t(l), hs = int16_t(h);
				const int ai = ls + (hs & 1) + (hs >> 1);
				a = uint16_t(int16_t(ai));
				b = uint16_t(int16_t(ai - hs));
			}

			inline void wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b)
			{
				const int m = l, d = h;
				const int bb = (m - (d >> 1)) & 0xffff;
				const int aa = (d + bb - 0x8000) & 0xffff;
				b = uint16_t(bb);
				a = uint16_t(aa);
			}

			/// inverse 2-D Haar-like wavelet on nx x ny values with strides ox / oy (in uint16 units)
			void waveletDecode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx)
			{
				const bool w14 = mx < (1 << 14);
				const int n = nx > ny ? ny : nx;
				int p = 1, p2;
				while(p <= n) p <<= 1;
				p >>= 1;
				p2 = p;
				p >>= 1;
				while(p >= 1)
				{
					uint16_t* py = in;
					uint16_t* ey = in + ptrdiff_t(oy) * (ny - p2);
					const ptrdiff_t oy1 = ptrdiff_t(oy) * p, oy2 = ptrdiff_t(oy) * p2, ox1 = ptrdiff_t(ox) * p, ox2 = ptrdiff_t(ox) * p2;
					uint16_t i00, i01, i10, i11;
					for(; py <= ey; py += oy2)
					{
						uint16_t* px = py;
						uint16_t* ex = py + ptrdiff_t(ox) * (nx - p2);
						for(; px <= ex; px += ox2)
						{
							uint16_t* p01 = px + ox1;
							uint16_t* p10 = px + oy1;
							uint16_t* p11 = p10 + ox1;
							if(w14)
							{
								wdec14(*px, *p10, i00, i10); wdec14(*p01, *p11, i01, i11);
								wdec14(i00, i01, *px, *p01); wdec14(i10, i11, *p10, *p11);
							}
							else
							{
								wdec16(*px, *p10, i00, i10); wdec16(*p01, *p11, i01, i11);
								wdec16(i00, i01, *px, *p01); wdec16(i10, i11, *p10, *p11);
							}
						}
						if(nx & p)
						{
							uint16_t* p10 = px + oy1;
							if(w14) wdec14(*px, *p10, i00, *p10); else wdec16(*px, *p10, i00, *p10);
							*px = i00;
						}
					}
					if(ny & p)
					{
						uint16_t* px = py;
						uint16_t* ex = py + ptrdiff_t(ox) * (nx - p2);
						for(; px <= ex; px += ox2)
						{
							uint16_t* p01 = px + ox1;
							if(w14) wdec14(*px, *p01, i00, *p01); else wdec16(*px, *p01, i00, *p01);
							*px = i00;
						}
					}
					p2 = p;
					p >>= 1;
				}
			}

			/// one chunk: `lines` scanlines of W pixels; channelWords[c] = 16-bit words per pixel (1 half, 2 float / uint).  out = the chunk
			/// in the uncompressed scanline layout.
			bool decodeChunk(const unsigned char* src, size_t size, int W, int lines, const vector<int>& channelWords, vector<unsigned char>& out)
			{
				if(size < 4) return false;
				uint16_t minNonZero, maxNonZero;
				memcpy(&minNonZero, src, 2); memcpy(&maxNonZero, src + 2, 2);
				const int bitmapSize = 8192;
				if(maxNonZero >= bitmapSize) return false;
				vector<unsigned char> bitmap(bitmapSize, 0);
				size_t pos = 4;
				if(minNonZero <= maxNonZero)
				{
					const size_t n = size_t(maxNonZero) - minNonZero + 1;
					if(pos + n > size) return false;
					memcpy(&bitmap[minNonZero], src + pos, n);
					pos += n;
				}
				vector<uint16_t> lut(1 << 16, 0);
				int k = 0;
				for(int i = 0; i < (1 << 16); ++i)
					if(i == 0 || (bitmap[i >> 3] & (1 << (i & 7)))) lut[k++] = uint16_t(i);
				const uint16_t maxValue = uint16_t(k - 1);
				if(pos + 4 > size) return false;
				int32_t length;
				memcpy(&length, src + pos, 4);
				pos += 4;
				if(length < 0 || pos + size_t(length) > size) return false;
				size_t total = 0;
				for(int words : channelWords) total += size_t(W) * lines * words;
				vector<uint16_t> buffer;
				if(!huffmanDecode(src + pos, size_t(length), buffer, total)) return false;
				size_t start = 0;
				for(int words : channelWords)
				{
					for(int j = 0; j < words; ++j) waveletDecode(&buffer[start + j], W, words, lines, W * words, maxValue);
					start += size_t(W) * lines * words;
				}
				for(uint16_t& v : buffer) v = lut[v];
				out.resize(total * 2);
				unsigned char* o = out.data();
				vector<size_t> cursor(channelWords.size());
				start = 0;
				for(size_t c = 0; c < channelWords.size(); ++c) { cursor[c] = start; start += size_t(W) * lines * channelWords[c]; }
				for(int l = 0; l < lines; ++l)
					for(size_t c = 0; c < channelWords.size(); ++c)
					{
						const size_t n = size_t(W) * channelWords[c];
						memcpy(o, &buffer[cursor[c]], n * 2);
						cursor[c] += n;
						o += n * 2;
					}
				return true;
			}
		} // namespace piz

		struct Reader
		{
			const vector<unsigned char>& d;
			size_t pos;
			bool ok;
			explicit Reader(const vector<unsigned char>& data) : d(data), pos(0), ok(true) {}
			template<typename T> T get()
			{
				T v = T();
				if(pos > d.size() || d.size() - pos < sizeof(T)) { ok = false; return v; }
				memcpy(&v, &d[pos], sizeof(T));
				pos += sizeof(T);
				return v;
			}
			string str()
			{
				string s;
				while(pos < d.size() && d[pos]) s.push_back(char(d[pos++]));
				if(pos >= d.size()) { ok = false; return s; }
				++pos;
				return s;
			}
		};

		/// planes[c] = W*H floats of channel c (channels in file order)
		bool readExrUnchecked(const char* path, int& W, int& H, vector<Channel>& channels, vector< vector<float> >& planes)
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
				if(!r.ok || attrSize < 0 || size_t(attrSize) > data.size() - r.pos) return fail("corrupt EXR header");
				const size_t next = r.pos + size_t(attrSize);
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
			if(compression < e_none || compression > e_piz)
				return fail("unsupported EXR compression (NONE, RLE, ZIPS, ZIP and PIZ are implemented; re-save PXR24/B44/DWA files)");
			// window extents in 64 bits, and a size the file could possibly hold: every chunk costs at least 8 bytes of header + 8 of offset
			const int64_t W64 = int64_t(maxX) - minX + 1, H64 = int64_t(maxY) - minY + 1;
			if(W64 <= 0 || H64 <= 0 || W64 > (1 << 24) || H64 > (1 << 24)) return fail("EXR data window out of range");
			W = int(W64);
			H = int(H64);
			const int linesPerBlock = compression == e_zip ? 16 : (compression == e_piz ? 32 : 1);
			const int nbOfBlocks = (H + linesPerBlock - 1) / linesPerBlock;
			if(size_t(nbOfBlocks) > (data.size() - r.pos) / 16) return fail("EXR data window larger than the file can hold");
			size_t bytesPerLine = 0;
			vector<int> channelWords;
			for(const Channel& c : channels) { bytesPerLine += size_t(c.type == e_half ? 2 : 4) * W; channelWords.push_back(c.type == e_half ? 1 : 2); }
			if(uint64_t(W64) * uint64_t(H64) * channels.size() > (uint64_t(1) << 33)) return fail("EXR image too large");
			vector<uint64_t> offsets(nbOfBlocks);
			for(int b = 0; b < nbOfBlocks; ++b) offsets[b] = r.get<uint64_t>();
			if(!r.ok) return fail("truncated EXR offset table");

			planes.assign(channels.size(), vector<float>(size_t(W) * H, 0.f));
			// the chunks are independent (own offset, own lines): decoded on all host cores -- a 60-channel 1080p histogram file is
			// half a gigabyte of zlib streams, seconds on one core next to a 7 ms denoise
			const char* firstError = nullptr;
			bool outOfMemory = false;
#pragma omp parallel
			{
				vector<unsigned char> raw, tmp;
#pragma omp for schedule(dynamic, 1)
				for(int b = 0; b < nbOfBlocks; ++b)
				{
					const char* err = nullptr;
					try
					{
						err = [&]() -> const char*
						{
							if(offsets[b] > data.size() - 8) return "EXR chunk offset out of range";
							Reader rb = r;
							rb.pos = size_t(offsets[b]);
							const int y = rb.get<int32_t>() - minY;
							const int32_t chunkSize = rb.get<int32_t>();
							if(!rb.ok || chunkSize < 0 || size_t(chunkSize) > data.size() - rb.pos || y < 0 || y >= H) return "corrupt EXR chunk";
							const int lines = min(linesPerBlock, H - y);
							const size_t expected = bytesPerLine * lines;
							const unsigned char* src = &data[rb.pos];
							if(size_t(chunkSize) == expected || compression == e_none)
							{
								if(size_t(chunkSize) != expected) return "EXR chunk has an unexpected size";
								raw.assign(src, src + expected);
							}
							else if(compression == e_rle)
							{
								if(!rleDecode(src, size_t(chunkSize), raw, expected)) return "corrupt RLE data in EXR chunk";
								undoPredictorAndReorder(raw, tmp);
							}
							else if(compression == e_piz)
							{
								if(!piz::decodeChunk(src, size_t(chunkSize), W, lines, channelWords, raw) || raw.size() != expected) return "corrupt PIZ data in EXR chunk";
							}
							else
							{
								raw.resize(expected);
								uLongf outSize = uLongf(expected);
								if(uncompress(raw.data(), &outSize, src, uLong(chunkSize)) != Z_OK || outSize != expected) return "corrupt ZIP data in EXR chunk";
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
							return nullptr;
						}();
					}
					catch(const std::bad_alloc&) { err = "out of memory"; }   // (an exception must not leave the parallel region)
					catch(const std::length_error&) { err = "out of memory"; }
					if(err)
					{
#pragma omp critical(bcd_exr_read_error)
						{
							if(!firstError) firstError = err;
							if(string(err) == "out of memory") outOfMemory = true;
						}
					}
				}
			}
			if(outOfMemory) throw std::bad_alloc();
			if(firstError) return fail(firstError);
			return true;
		}

		/// a crafted or truncated file must end in `false`, never in an exception escaping the library
		bool readExr(const char* path, int& W, int& H, vector<Channel>& channels, vector< vector<float> >& planes)
		{
			try
			{
				return readExrUnchecked(path, W, H, channels, planes);
			}
			catch(const std::bad_alloc&)
			{
				return fail("out of memory while reading the EXR file (header announces more data than can be allocated)");
			}
			catch(const std::length_error&)
			{
				return fail("EXR header announces an impossible size");
			}
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
			// blocks are compressed on all host cores, then laid out in order
			vector< vector<unsigned char> > blocks(nbOfBlocks);
			bool zlibFailed = false;
#pragma omp parallel
			{
				vector<unsigned char> raw, shuffled, packed;
#pragma omp for schedule(dynamic, 1)
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
					if(compress2(packed.data(), &packedSize, shuffled.data(), uLong(shuffled.size()), Z_DEFAULT_COMPRESSION) != Z_OK)
					{
#pragma omp critical(bcd_exr_write_error)
						zlibFailed = true;
						continue;
					}
					vector<unsigned char>& blk = blocks[b];
					put<int32_t>(blk, y);
					if(packedSize < raw.size())
					{
						put<int32_t>(blk, int32_t(packedSize));
						blk.insert(blk.end(), packed.begin(), packed.begin() + packedSize);
					}
					else
					{	// incompressible block: stored raw
						put<int32_t>(blk, int32_t(raw.size()));
						blk.insert(blk.end(), raw.begin(), raw.end());
					}
				}
			}
			if(zlibFailed) return fail("zlib compression failed");
			for(int b = 0; b < nbOfBlocks; ++b)
			{
				const uint64_t offset = out.size();
				memcpy(&out[tablePos + size_t(b) * 8], &offset, 8);
				out.insert(out.end(), blocks[b].begin(), blocks[b].end());
				vector<unsigned char>().swap(blocks[b]);
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
		const long long n = (long long)w * h;
#pragma omp parallel for schedule(static)
		for(long long i = 0; i < n; ++i)
			for(int z = 0; z < depth; ++z)
				dst[size_t(i) * depth + z] = planes[z][size_t(i)];
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
