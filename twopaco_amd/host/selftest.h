// selftest.h -- `twopaco --test`: the randomized differential test of the reference
// (reference src/graphconstructor/test.{h,cpp}): junction positions of the GPU path against a
// naive set-based junction finder, plus GetId != INVALID_VERTEX for every junction.
#ifndef _TPC_SELFTEST_H_
#define _TPC_SELFTEST_H_

#include <cstdint>
#include <string>
#include <utility>

namespace TwoPaCo
{
	typedef std::pair<size_t, size_t> Range;
	bool RunTests(size_t tests, size_t filterBits, size_t length, size_t chrNumber, Range vertexSize, Range hashFunctions,
		Range rounds, Range threads, double changeRate, double indelRate, const std::string & temporaryDir);

	// The same with a reproducible random stream (`twopaco --test --seed S`).  The reference draws every trial
	// from std::random_device (test.cpp:169) and a failure cannot be replayed; here trial t uses mt19937_64(seed + t) for its
	// sequences AND pins the hash tables of its CreateEnumerator calls to the same number, and a failing trial prints
	// "Test # t FAILED (replay: --test --seed <seed + t>)".  The unseeded form above draws `seed` from std::random_device.
	bool RunTestsSeeded(uint64_t seed, size_t tests, size_t filterBits, size_t length, size_t chrNumber, Range vertexSize, Range hashFunctions,
		Range rounds, Range threads, double changeRate, double indelRate, const std::string & temporaryDir);
}

#endif
