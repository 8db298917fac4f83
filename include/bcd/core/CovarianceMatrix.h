// CovarianceMatrix.h -- 3x3 symmetric blocks; storage order is part of the file/buffer contract
// (reference include/bcd/core/CovarianceMatrix.h:18-56): xx, yy, zz, yz, xz, xy.
#ifndef COVARIANCE_MATRIX_H
#define COVARIANCE_MATRIX_H

#include <array>
#include <cstddef>
#include <vector>

namespace bcd
{

	enum class ESymmetricMatrix3x3Data { e_xx, e_yy, e_zz, e_yz, e_xz, e_xy, e_nb };
	typedef ESymmetricMatrix3x3Data ESymMatData;

	class SymmetricMatrix3x3
	{
	public:
		SymmetricMatrix3x3() {}
		SymmetricMatrix3x3& operator+=(const SymmetricMatrix3x3& o)
		{
			for(std::size_t i = 0; i < m_data.size(); ++i) m_data[i] += o.m_data[i];
			return *this;
		}
		SymmetricMatrix3x3& operator*=(float f)
		{
			for(float& v : m_data) v *= f;
			return *this;
		}
		void copyFrom(const float* p) { for(std::size_t i = 0; i < m_data.size(); ++i) m_data[i] = p[i]; }

	public:
		std::array<float, static_cast<std::size_t>(ESymMatData::e_nb)> m_data;
	};
	typedef SymmetricMatrix3x3 CovMat3x3;

	class Block3x3DiagonalSymmetricMatrix
	{
	public:
		Block3x3DiagonalSymmetricMatrix() {}
		Block3x3DiagonalSymmetricMatrix(std::size_t n) : m_blocks(n) {}
		Block3x3DiagonalSymmetricMatrix& operator+=(const Block3x3DiagonalSymmetricMatrix& o)
		{
			for(std::size_t i = 0; i < m_blocks.size(); ++i) m_blocks[i] += o.m_blocks[i];
			return *this;
		}
		Block3x3DiagonalSymmetricMatrix& operator*=(float f)
		{
			for(SymmetricMatrix3x3& b : m_blocks) b *= f;
			return *this;
		}

	public:
		std::vector<SymmetricMatrix3x3> m_blocks;
	};
	typedef Block3x3DiagonalSymmetricMatrix CovMatPatch;

} // namespace bcd

#endif // COVARIANCE_MATRIX_H
