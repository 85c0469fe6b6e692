// textpack.h -- parse-once replacement of the reference's reader thread.
//
// The reference re-parses every FASTA file 3*rounds+1 times and streams each record as
// 'N' + bases + 'N' in overlapping Tasks (reference vertexenumerator.h:1108-1226).  Here every
// file is parsed once into the global text  T = N rec0 N rec1 N ... rec(S-1) N  in the device
// layout of include/twopaco_hip.h: 2-bit codes (32 per uint64) plus an N bit mask.  Every record
// consumes a sequence id, also the ones shorter than k that the reference never dispatches
// (vertexenumerator.h:1135,1177).
#ifndef _TPC_TEXTPACK_H_
#define _TPC_TEXTPACK_H_

#include <cstdint>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

namespace TwoPaCo
{
	// std::allocator whose value-less construct() leaves trivially constructible elements uninitialised, so that
	// resize(n) does not run a serial fill over hundreds of MB (the packer's threads zero their own ranges)
	template<class T>
	struct DefaultInitAllocator : std::allocator<T>
	{
		template<class U> struct rebind { typedef DefaultInitAllocator<U> other; };
		DefaultInitAllocator() {}
		template<class U> DefaultInitAllocator(const DefaultInitAllocator<U> &) {}
		template<class U> void construct(U * p) { ::new(static_cast<void*>(p)) U; }
		template<class U, class... Args> void construct(U * p, Args &&... args) { ::new(static_cast<void*>(p)) U(std::forward<Args>(args)...); }
	};

	struct PackedText
	{
		std::vector<uint64_t, DefaultInitAllocator<uint64_t> > bases;  // base g at bits 2*(g%32) of bases[g/32]
		std::vector<uint32_t, DefaultInitAllocator<uint32_t> > nmask;  // bit g%32 of nmask[g/32]: T[g] is 'N'
		uint64_t length;                 // characters in T
		std::vector<uint64_t> recStart;  // global position of the first base of record r
		std::vector<uint64_t> recLength; // bases in record r

		PackedText() : length(0) {}
		void AppendCodes(const uint8_t * codes, uint64_t n);  // 0..3 bases, 4 = N
		void AppendPacked(const uint64_t * b, const uint32_t * m, uint64_t n);  // n characters already packed from bit 0
		void BeginText();                                      // leading separator
		void EndRecord(uint64_t recordBases);                  // bookkeeping + trailing separator
	};

	// Parses all files (threads > 1: files in parallel) and packs them.  Throws StreamFastaParser::Exception.
	void PackFastaFiles(const std::vector<std::string> & fileName, size_t threads, PackedText & out);
}

#endif
