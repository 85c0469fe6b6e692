// seed.h -- character tables of the q cyclic-polynomial hash functions.
//
// The reference draws, per hash function, a 256-entry table from two Mersenne Twister
// generators seeded from /dev/urandom (reference src/common/ngramhashing/characterhash.h:41-54,
// mersennetwister.h:242-263); only the entries of 'A','C','G','T','N' are ever read.  This module
// produces exactly those q x 5 values from a stream of 2q x 624 entropy words, taken either from
// /dev/urandom (default, like the reference) or from the deterministic stream
// tpc_urandom_word(seed, n, j) -- the same stream oracle/urandom_shim.c feeds the reference
// binary, so `--seed S` reproduces a reference run pinned with TPC_URANDOM_SEED=S.
#ifndef _TPC_SEED_H_
#define _TPC_SEED_H_

#include <cstddef>
#include <cstdint>
#include <vector>

namespace TwoPaCo
{
	uint64_t tpc_urandom_word(uint64_t seed, uint64_t nopen, uint64_t j);

	// table[i * 5 + c], c in A,C,G,T,N.  pinned = false reads /dev/urandom.
	std::vector<uint64_t> MakeSeedTable(size_t hashFunctions, size_t bits, bool pinned, uint64_t seed);

	// What VertexEnumerator::GetHashSeed() hands out (reference vertexrollinghash.h:13-52).
	class VertexRollingHashSeed
	{
	public:
		VertexRollingHashSeed() : vertexLength_(0), bits_(0) {}
		VertexRollingHashSeed(size_t numberOfFunctions, size_t vertexLength, size_t bits, const std::vector<uint64_t> & table)
			: vertexLength_(vertexLength), bits_(bits), functions_(numberOfFunctions), table_(table) {}
		size_t VertexLength() const { return vertexLength_; }
		size_t BitsNumber() const { return bits_; }
		size_t HashFunctionsNumber() const { return functions_; }
		const std::vector<uint64_t> & Table() const { return table_; }
	private:
		size_t vertexLength_;
		size_t bits_;
		size_t functions_;
		std::vector<uint64_t> table_;
	};
}

#endif
