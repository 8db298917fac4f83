// DeepImage.h -- image container and window iterators of the bcd API (MI355X build).
//
// API-compatible with the reference's include/bcd/core/DeepImage.h + DeepImage.hpp (same class, typedef
// and member-function names, same semantics) so that code written against libbcdcore compiles unchanged;
// written from scratch for this build.  Contract that the GPU engine relies on (and that the reference
// defines at DeepImage.hpp:385-396): a W x H x D image is ONE contiguous buffer, pixel-major,
//     index(line, column, d) = line * W * D + column * D + d        (int indices)
// Windows are "centre +/- radius, CLIPPED (not shifted) to [border, dim-1-border]" (DeepImage.hpp:181-196),
// iterated row-major.
#ifndef DEEP_IMAGE_H
#define DEEP_IMAGE_H

#include <algorithm>
#include <cassert>
#include <iostream>
#include <type_traits>
#include <utility>
#include <vector>

namespace bcd
{

	class PixelPosition
	{
	public:
		PixelPosition() : m_line(0), m_column(0) {}
		PixelPosition(int i_line, int i_column) : m_line(i_line), m_column(i_column) {}
		PixelPosition(const PixelPosition&) = default;
		PixelPosition& operator=(const PixelPosition&) = default;

	public:
		int m_line;
		int m_column;

	public:
		void get(int& o_rLine, int& o_rColumn) const { o_rLine = m_line; o_rColumn = m_column; }
		PixelPosition operator+(const PixelPosition& p) const { return PixelPosition(m_line + p.m_line, m_column + p.m_column); }
		PixelPosition operator-(const PixelPosition& p) const { return PixelPosition(m_line - p.m_line, m_column - p.m_column); }
		bool operator==(const PixelPosition& p) const { return m_line == p.m_line && m_column == p.m_column; }
		bool operator!=(const PixelPosition& p) const { return !(*this == p); }
	};

	typedef PixelPosition PixelVector;
	typedef PixelPosition PixelWindowSize;
	typedef PixelPosition PatchSize;
	typedef PixelPosition ImageWindowSize;

	namespace detail
	{
		/// inclusive rectangle [min, max] = centre +/- radius clipped to [border, dim - 1 - border]
		struct ClippedRect
		{
			PixelPosition lo, hi;
			ClippedRect() {}
			ClippedRect(int w, int h, PixelPosition c, int radius, int border) :
				lo(std::max(border, c.m_line - radius), std::max(border, c.m_column - radius)),
				hi(std::min(h - 1 - border, c.m_line + radius), std::min(w - 1 - border, c.m_column + radius)) {}
			PixelWindowSize size() const { return PixelWindowSize(1, 1) + hi - lo; }
		};
	}

	/// Positions of a clipped square window, row-major
	class PixelWindow
	{
	public:
		class iterator
		{
		public:
			iterator() {}
			iterator(PixelPosition i_centralPixel, int i_radius) { reset(i_centralPixel, i_radius); }
			iterator(int i_bufferWidth, int i_bufferHeight, PixelPosition i_centralPixel, int i_radius, int i_border = 0)
			{ reset(i_bufferWidth, i_bufferHeight, i_centralPixel, i_radius, i_border); }
			iterator(PixelPosition i_minCorner, PixelPosition i_maxCorner, PixelPosition i_currentPixel) :
				m_minCorner(i_minCorner), m_maxCorner(i_maxCorner), m_currentPixel(i_currentPixel) {}
			iterator(const iterator&) = default;

		public:
			void reset(PixelPosition c, int r)
			{
				m_minCorner = PixelPosition(c.m_line - r, c.m_column - r);
				m_maxCorner = PixelPosition(c.m_line + r, c.m_column + r);
				m_currentPixel = m_minCorner;
			}
			void reset(int w, int h, PixelPosition c, int r, int border = 0)
			{
				detail::ClippedRect rc(w, h, c, r, border);
				m_minCorner = rc.lo; m_maxCorner = rc.hi; m_currentPixel = rc.lo;
			}
			PixelWindowSize getSize() const { return PixelWindowSize(1, 1) + m_maxCorner - m_minCorner; }
			const PixelPosition& operator*() const { return m_currentPixel; }
			iterator& operator++()
			{
				if(m_currentPixel.m_column == m_maxCorner.m_column)
				{
					++m_currentPixel.m_line;
					m_currentPixel.m_column = m_minCorner.m_column;
				}
				else
					++m_currentPixel.m_column;
				return *this;
			}
			bool hasEnded() const { return m_currentPixel.m_line > m_maxCorner.m_line; }
			bool operator!=(const iterator& it) const { return m_currentPixel != it.m_currentPixel; }

		private:
			PixelPosition m_minCorner, m_maxCorner, m_currentPixel;
		};

	public:
		PixelWindow() : m_width(0), m_height(0) {}
		PixelWindow(int i_bufferWidth, int i_bufferHeight, PixelPosition i_centralPixel, int i_radius, int i_border = 0)
		{ reset(i_bufferWidth, i_bufferHeight, i_centralPixel, i_radius, i_border); }

	public:
		void reset(int i_bufferWidth, int i_bufferHeight, PixelPosition i_centralPixel, int i_radius, int i_border = 0)
		{
			m_width = i_bufferWidth; m_height = i_bufferHeight;
			detail::ClippedRect rc(i_bufferWidth, i_bufferHeight, i_centralPixel, i_radius, i_border);
			m_minCorner = rc.lo; m_maxCorner = rc.hi;
		}
		PixelWindowSize getSize() const { return PixelWindowSize(1, 1) + m_maxCorner - m_minCorner; }
		iterator begin() const { return iterator(m_minCorner, m_maxCorner, m_minCorner); }
		iterator end() const { return iterator(m_minCorner, m_maxCorner, PixelPosition(m_maxCorner.m_line + 1, m_minCorner.m_column)); }

	private:
		int m_width, m_height;
		PixelPosition m_minCorner, m_maxCorner;
	};

	typedef PixelWindow PixelPatch;
	typedef PixelWindow::iterator PixWinIt;
	typedef PixelPatch::iterator PixPatchIt;

	/// 2D buffer of D-vectors seen as a 3D buffer of scalars (interleaved, see the header comment)
	template <typename scalar = float>
	class DeepImage
	{
		template <typename S>
		class PixelIterator
		{
		public:
			PixelIterator() : m_p(nullptr), m_stride(0) {}
			PixelIterator(S* p, int stride) : m_p(p), m_stride(stride) {}
			S* operator*() const { return m_p; }
			PixelIterator& operator++() { m_p += m_stride; return *this; }
			S& operator[](int d) const { return m_p[d]; }
			bool operator!=(const PixelIterator& it) const { return m_p != it.m_p; }
		private:
			S* m_p;
			int m_stride;
		};

	public:
		typedef PixelIterator<scalar> iterator;
		typedef PixelIterator<const scalar> const_iterator;

	public:
		DeepImage() : m_width(0), m_height(0), m_depth(0), m_widthTimesDepth(0) {}
		DeepImage(int i_width, int i_height, int i_depth) :
			m_width(i_width), m_height(i_height), m_depth(i_depth), m_widthTimesDepth(i_width * i_depth),
			m_data(static_cast<std::size_t>(i_width) * i_height * i_depth) {}
		DeepImage(const DeepImage&) = default;
		DeepImage(DeepImage&& o) : DeepImage() { *this = std::move(o); }
		DeepImage& operator=(const DeepImage&) = default;
		DeepImage& operator=(DeepImage&& o)
		{
			if(this != &o)
			{
				m_width = o.m_width; m_height = o.m_height; m_depth = o.m_depth; m_widthTimesDepth = o.m_widthTimesDepth;
				m_data = std::move(o.m_data);
				o.m_width = o.m_height = o.m_depth = o.m_widthTimesDepth = 0;
				o.m_data.clear();
			}
			return *this;
		}
		~DeepImage() = default;

	public:
		void resize(int i_width, int i_height, int i_depth)
		{
			m_width = i_width; m_height = i_height; m_depth = i_depth; m_widthTimesDepth = i_width * i_depth;
			m_data.resize(static_cast<std::size_t>(i_width) * i_height * i_depth);
		}
		void copyDataFrom(const scalar* i_pData) { std::copy(i_pData, i_pData + m_data.size(), m_data.begin()); }
		void copyDataTo(scalar* i_pData) const { std::copy(m_data.begin(), m_data.end(), i_pData); }

		int getWidth() const { return m_width; }
		int getHeight() const { return m_height; }
		int getDepth() const { return m_depth; }
		int getSize() const { return static_cast<int>(m_data.size()); }
		scalar* getDataPtr() { return m_data.data(); }
		const scalar* getDataPtr() const { return m_data.data(); }

		PixelPosition clamp(const PixelPosition& pos) const
		{
			return PixelPosition(std::max(0, std::min(pos.m_line, m_height - 1)), std::max(0, std::min(pos.m_column, m_width - 1)));
		}

		int glueIndices(int i_line, int i_column, int i_dimensionIndex) const
		{
			assert(i_line >= 0 && i_line < m_height && i_column >= 0 && i_column < m_width);
			assert(i_dimensionIndex >= 0 && i_dimensionIndex < m_depth);
			return i_line * m_widthTimesDepth + i_column * m_depth + i_dimensionIndex;
		}
		static int glueIndices(int i_width, int, int i_depth, int i_line, int i_column, int i_dimensionIndex)
		{
			return (i_line * i_width + i_column) * i_depth + i_dimensionIndex;
		}
		void splitIndex(int& o_rLine, int& o_rColumn, int& o_rDimensionIndex, int i_buffer1DIndex) const
		{
			splitIndex(o_rLine, o_rColumn, o_rDimensionIndex, i_buffer1DIndex, m_width, m_height, m_depth);
		}
		static void splitIndex(int& o_rLine, int& o_rColumn, int& o_rDimensionIndex, int i_buffer1DIndex, int i_width, int, int i_depth)
		{
			o_rDimensionIndex = i_buffer1DIndex % i_depth;
			int pixel = i_buffer1DIndex / i_depth;
			o_rColumn = pixel % i_width;
			o_rLine = pixel / i_width;
		}

		const scalar& get(int l, int c, int d) const { return m_data[glueIndices(l, c, d)]; }
		scalar& get(int l, int c, int d) { return m_data[glueIndices(l, c, d)]; }
		const scalar& get(PixelPosition p, int d) const { return m_data[glueIndices(p.m_line, p.m_column, d)]; }
		scalar& get(PixelPosition p, int d) { return m_data[glueIndices(p.m_line, p.m_column, d)]; }
		const scalar& get(int i) const { return m_data[i]; }
		scalar& get(int i) { return m_data[i]; }
		const scalar getValue(PixelPosition p, int d) const { return m_data[glueIndices(p.m_line, p.m_column, d)]; }
		scalar getValue(PixelPosition p, int d) { return m_data[glueIndices(p.m_line, p.m_column, d)]; }

		void set(int l, int c, int d, scalar v) { m_data[glueIndices(l, c, d)] = v; }
		void set(PixelPosition p, int d, scalar v) { m_data[glueIndices(p.m_line, p.m_column, d)] = v; }
		void set(int i, scalar v) { m_data[i] = v; }
		void set(int l, int c, const scalar* i_pVectorValue)
		{
			std::copy(i_pVectorValue, i_pVectorValue + m_depth, m_data.begin() + glueIndices(l, c, 0));
		}
		void set(PixelPosition p, const scalar* i_pVectorValue) { set(p.m_line, p.m_column, i_pVectorValue); }

		void isotropicalScale(scalar f) { for(auto& v : m_data) v *= f; }
		void anisotropicalScale(const scalar* i_scaleFactors)
		{
			for(std::size_t i = 0; i < m_data.size(); ++i) m_data[i] *= i_scaleFactors[i % m_depth];
		}
		void fill(scalar f) { std::fill(m_data.begin(), m_data.end(), f); }
		bool isEmpty() const { return m_width == 0 || m_height == 0 || m_depth == 0; }
		void clearAndFreeMemory()
		{
			m_width = m_height = m_depth = m_widthTimesDepth = 0;
			std::vector<scalar>().swap(m_data);
		}

		iterator begin() { return iterator(m_data.data(), m_depth); }
		iterator end() { return iterator(m_data.data() + m_data.size(), m_depth); }
		const_iterator begin() const { return const_iterator(m_data.data(), m_depth); }
		const_iterator end() const { return const_iterator(m_data.data() + m_data.size(), m_depth); }

		DeepImage& operator+=(const DeepImage& o)
		{
			assert(o.m_data.size() == m_data.size());
			for(std::size_t i = 0; i < m_data.size(); ++i) m_data[i] += o.m_data[i];
			return *this;
		}
		DeepImage& operator-=(const DeepImage& o)
		{
			assert(o.m_data.size() == m_data.size());
			for(std::size_t i = 0; i < m_data.size(); ++i) m_data[i] -= o.m_data[i];
			return *this;
		}

	private:
		int m_width, m_height, m_depth;
		int m_widthTimesDepth;
		std::vector<scalar> m_data;
	};

	typedef DeepImage<float> Deepimf;
	typedef Deepimf::iterator ImfIt;
	typedef Deepimf::const_iterator ImfConstIt;

	namespace detail
	{
		/// shared implementation of ImageWindow / ConstImageWindow; S is `scalar` or `const scalar`
		template <typename S>
		class WindowT
		{
			typedef typename std::remove_const<S>::type value_type;
			typedef typename std::conditional<std::is_const<S>::value, const DeepImage<value_type>, DeepImage<value_type> >::type image_type;

		public:
			class iterator
			{
			public:
				iterator() : m_width(0), m_height(0), m_depth(0), m_pCurrentDataPointer(nullptr) {}
				iterator(image_type& i_rImage, PixelPosition i_centralPixel, int i_radius, int i_border = 0)
				{ reset(i_rImage, i_centralPixel, i_radius, i_border); }
				iterator(int w, int h, int d, PixelPosition lo, PixelPosition hi, PixelPosition cur, S* p) :
					m_width(w), m_height(h), m_depth(d), m_minCorner(lo), m_maxCorner(hi), m_currentPixel(cur), m_pCurrentDataPointer(p) {}
				iterator(const iterator&) = default;

			public:
				void reset(image_type& i_rImage, PixelPosition c, int r, int border = 0)
				{
					m_width = i_rImage.getWidth(); m_height = i_rImage.getHeight(); m_depth = i_rImage.getDepth();
					ClippedRect rc(m_width, m_height, c, r, border);
					m_minCorner = rc.lo; m_maxCorner = rc.hi; m_currentPixel = rc.lo;
					m_pCurrentDataPointer = i_rImage.getDataPtr() + (rc.lo.m_line * m_width + rc.lo.m_column) * m_depth;
				}
				ImageWindowSize getSize() const { return ImageWindowSize(1, 1) + m_maxCorner - m_minCorner; }
				S* operator*() const { return m_pCurrentDataPointer; }
				iterator& operator++()
				{
					if(m_currentPixel.m_column == m_maxCorner.m_column)
					{	// jump to the first column of the next line
						m_pCurrentDataPointer += (m_width - (m_maxCorner.m_column - m_minCorner.m_column)) * m_depth;
						++m_currentPixel.m_line;
						m_currentPixel.m_column = m_minCorner.m_column;
					}
					else
					{
						m_pCurrentDataPointer += m_depth;
						++m_currentPixel.m_column;
					}
					return *this;
				}
				bool hasEnded() const { return m_currentPixel.m_line > m_maxCorner.m_line; }
				S& operator[](int d) const { return m_pCurrentDataPointer[d]; }
				bool operator!=(const iterator& it) const { return m_pCurrentDataPointer != it.m_pCurrentDataPointer; }

			private:
				int m_width, m_height, m_depth;
				PixelPosition m_minCorner, m_maxCorner, m_currentPixel;
				S* m_pCurrentDataPointer;
			};

		public:
			WindowT() : m_width(0), m_height(0), m_depth(0), m_pMinCornerDataPointer(nullptr) {}
			WindowT(image_type& i_rImage, PixelPosition i_centralPixel, int i_radius, int i_border = 0)
			{ reset(i_rImage, i_centralPixel, i_radius, i_border); }

		public:
			void reset(image_type& i_rImage, PixelPosition c, int r, int border = 0)
			{
				m_width = i_rImage.getWidth(); m_height = i_rImage.getHeight(); m_depth = i_rImage.getDepth();
				ClippedRect rc(m_width, m_height, c, r, border);
				m_minCorner = rc.lo; m_maxCorner = rc.hi;
				m_pMinCornerDataPointer = i_rImage.getDataPtr() + (rc.lo.m_line * m_width + rc.lo.m_column) * m_depth;
			}
			ImageWindowSize getSize() const { return ImageWindowSize(1, 1) + m_maxCorner - m_minCorner; }
			iterator begin() const
			{ return iterator(m_width, m_height, m_depth, m_minCorner, m_maxCorner, m_minCorner, m_pMinCornerDataPointer); }
			iterator end() const
			{
				return iterator(m_width, m_height, m_depth, m_minCorner, m_maxCorner,
						PixelPosition(m_maxCorner.m_line + 1, m_minCorner.m_column),
						m_pMinCornerDataPointer + (m_maxCorner.m_line + 1 - m_minCorner.m_line) * m_width * m_depth);
			}

		private:
			int m_width, m_height, m_depth;
			PixelPosition m_minCorner, m_maxCorner;
			S* m_pMinCornerDataPointer;
		};
	}

	/// Pixel-data pointers of a clipped window of an image, row-major (mutable / const flavours)
	template <typename scalar = float> using ImageWindow = detail::WindowT<scalar>;
	template <typename scalar = float> using ConstImageWindow = detail::WindowT<const scalar>;

	typedef ImageWindow<float> Win;
	typedef Win Patch;
	typedef Win::iterator WinIt;
	typedef WinIt PatchIt;
	typedef ConstImageWindow<float> ConstWin;
	typedef ConstWin ConstPatch;
	typedef ConstWin::iterator ConstWinIt;
	typedef ConstWinIt ConstPatchIt;

} // namespace bcd

#endif // DEEP_IMAGE_H
