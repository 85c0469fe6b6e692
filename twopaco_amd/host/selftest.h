// selftest.h -- `twopaco --test`: the randomized differential test of the reference
// (reference src/graphconstructor/test.{h,cpp}): junction positions of the GPU path against a
// naive set-based junction finder, plus GetId != INVALID_VERTEX for every junction.
#ifndef _TPC_SELFTEST_H_
#define _TPC_SELFTEST_H_

#include <string>
#include <utility>

namespace TwoPaCo
{
	typedef std::pair<size_t, size_t> Range;
	bool RunTests(size_t tests, size_t filterBits, size_t length, size_t chrNumber, Range vertexSize, Range hashFunctions,
		Range rounds, Range threads, double changeRate, double indelRate, const std::string & temporaryDir);
}

#endif
