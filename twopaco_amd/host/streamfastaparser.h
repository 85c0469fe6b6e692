// streamfastaparser.h -- FASTA reader with the record/character rules of the reference's
// StreamFastaParser (reference src/common/streamfastaparser.{h,cpp}): a record starts with '>'
// and its header runs to the first newline (streamfastaparser.cpp:29-59); sequence characters
// are upper-cased, whitespace is skipped, a '>' ends the record, anything outside
// "ACGTURYKMSWBDHWNXV" is an error (:61-93).  Unlike the reference's char-at-a-time reader the
// whole file is slurped and scanned in bulk, and the stale-byte hazard at buffer refills
// (reference streamfastaparser.cpp:95-133; see SURVEY 8a' item 5) does not exist here.
#ifndef _STREAM_FASTA_PARSER_H_
#define _STREAM_FASTA_PARSER_H_

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace TwoPaCo
{
	class StreamFastaParser
	{
	public:
		class Exception : public std::runtime_error
		{
		public:
			Exception(const std::string & msg) : std::runtime_error(msg) {}
		};

		StreamFastaParser(const std::string & fileName);
		bool ReadRecord();
		bool GetChar(char & ch);
		std::string GetCurrentHeader() const { return currentHeader_; }
		std::string GetErrorMessage() const { return std::string(); }

		// Bulk form of GetChar: appends the rest of the current record as codes A0 C1 G2 T3, 4 = any
		// other valid character ('N' after VertexEnumerator's mapping, reference vertexenumerator.h:1174).
		void ReadSequenceCodes(std::vector<uint8_t> & out);

		// Same scan, packed on the fly in the device layout: base i at bits 2*(i%32) of bases[i/32] (an
		// N keeps code 0), bit i%32 of nmask[i/32] set for N.  Returns the number of characters.
		uint64_t ReadSequencePacked(std::vector<uint64_t> & bases, std::vector<uint32_t> & nmask);

	private:
		std::string data_;
		size_t pos_;
		std::string currentHeader_;
	};
}

#endif
