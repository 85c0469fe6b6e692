// vertexenumerator.h -- the operator boundary of the junction-enumeration hot path, source
// compatible with the reference (reference src/graphconstructor/vertexenumerator.h:23-46):
// same abstract class, same factory signature, same side effects (junction stream written to
// outFileName through JunctionPositionWriter, progress text on logStream, runtime_error on
// failure).  The work the reference's constructor does on CPU threads
// (vertexenumerator.h:122-466) runs on one MI355X through the C-ABI of include/twopaco_hip.h.
#ifndef _VERTEX_ENUMERATOR_H_
#define _VERTEX_ENUMERATOR_H_

// What the reference's own translation units rely on their vertexenumerator.h to pull in (constructor.cpp uses log2,
// test.cpp uses CHAR_MAX, UINT32_MAX, DnaChar, std::thread-era headers): kept here so that both compile UNCHANGED
// against this header (tests/test_host_cpu.py::test_reference_main_and_selftest_compile_against_host_headers).
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <memory>
#include <numeric>
#include <ostream>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "dnachar.h"
#include "junctionapi.h"
#include "seed.h"
#include "streamfastaparser.h"

namespace TwoPaCo
{
	extern const int64_t INVALID_VERTEX;  // reference graphconstructor/common.cpp:5

	class VertexEnumerator
	{
	public:
		virtual size_t GetVerticesCount() const = 0;
		virtual int64_t GetId(const std::string & vertex) const = 0;
		virtual const VertexRollingHashSeed & GetHashSeed() const = 0;
		virtual ~VertexEnumerator() {}
	};

	// Knobs that do not exist in the reference; defaults reproduce its behaviour.
	struct EnumeratorOptions
	{
		bool pinnedSeed;      // false: hash tables from /dev/urandom like the reference
		uint64_t seed;        // with pinnedSeed: the TPC_URANDOM_SEED of oracle/urandom_shim.c
		int device;           // HIP device ordinal
		bool insertTestFirst; // test-then-set insert (reference vertexenumerator.h:1088) instead of plain atomicOr
		int gpus;             // > 1: Bloom filter sharded by bit address over devices device .. device+gpus-1 (a power of two; multigpu.h)
		bool rccl;            // transport between the GPUs: RCCL (default) or device-to-device copies
		bool emulateRanks;    // testing: all `gpus` ranks on ONE device (copies instead of RCCL, which refuses duplicate devices)
		bool forceSharded;    // testing: take the sharded path (and its transport) even with gpus == 1
		// Checkpoint of the most expensive state of a run, the Bloom filter after a round's first-pass insert (the reference
		// kept this as the commented-out ReloadBloomFilter, reference vertexenumerator.h:29,113-121).  saveFilter: every round
		// writes its filter to this file (round r > 0: "<file>.<r>") with the parameters and hash tables it was built with.
		// loadFilter: every round reads its filter from there instead of running the insert; the hash tables come from the
		// file (pinnedSeed / seed are ignored) and k, filter size, hash functions and the round's range must match.  Single GPU.
		std::string saveFilter;
		std::string loadFilter;
		EnumeratorOptions() : pinnedSeed(false), seed(0), device(0), insertTestFirst(false), gpus(1), rccl(true), emulateRanks(false), forceSharded(false) {}
	};

	std::unique_ptr<VertexEnumerator> CreateEnumerator(const std::vector<std::string> & fileName,
		size_t vertexLength,
		size_t filterSize,
		size_t hashFunctions,
		size_t rounds,
		size_t threads,
		size_t abundance,
		const std::string & tmpFileName,
		const std::string & outFileName,
		std::ostream & logStream);

	std::unique_ptr<VertexEnumerator> CreateEnumerator(const std::vector<std::string> & fileName,
		size_t vertexLength,
		size_t filterSize,
		size_t hashFunctions,
		size_t rounds,
		size_t threads,
		size_t abundance,
		const std::string & tmpFileName,
		const std::string & outFileName,
		std::ostream & logStream,
		const EnumeratorOptions & options);
}

#endif
