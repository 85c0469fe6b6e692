// dnachar.h -- DNA alphabet helpers with the semantics of the reference's DnaChar
// (reference src/common/dnachar.{h,cpp}): codes A0 C1 G2 T3 (dnachar.cpp:18-33), complement
// with every other character -> 'N' (:52-58), validity over "ACGTURYKMSWBDHWNXV" (:11).
#ifndef _DNA_CHAR_
#define _DNA_CHAR_

#include <cstddef>
#include <cstdint>
#include <string>

namespace TwoPaCo
{
	class DnaChar
	{
	public:
		static const std::string & Literal() { static const std::string s("ACGT"); return s; }
		static const std::string & ExtLiteral() { static const std::string s("ACGTN"); return s; }
		static const std::string & ValidChars() { static const std::string s("ACGTURYKMSWBDHWNXV"); return s; }

		static bool IsDefinite(char ch) { return ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'; }
		static bool IsValid(char ch) { return ch != '\0' && ValidChars().find(ch) != std::string::npos; }

		static char ReverseChar(char ch)
		{
			switch (ch)
			{
			case 'A': return 'T';
			case 'T': return 'A';
			case 'C': return 'G';
			case 'G': return 'C';
			}
			return 'N';
		}

		static std::string ReverseCompliment(const std::string & str)
		{
			std::string ret(str.rbegin(), str.rend());
			for (char & ch : ret) ch = ReverseChar(ch);
			return ret;
		}

		static size_t MakeUpChar(char ch)
		{
			switch (ch)
			{
			case 'A': return 0;
			case 'C': return 1;
			case 'G': return 2;
			case 'T': return 3;
			}
			return static_cast<size_t>(-1);
		}

		static char UnMakeUpChar(size_t code) { return code < 4 ? "ACGT"[code] : 'N'; }

		// true iff the k-mer at `it` is lexicographically smaller than its reverse complement (dnachar.cpp:98-114)
		static bool LessSelfReverseComplement(std::string::const_iterator it, size_t size)
		{
			for (size_t i = 0; i < size; i++)
			{
				char fwd = *(it + i);
				char rev = ReverseChar(*(it + (size - 1 - i)));
				if (fwd != rev) return fwd < rev;
			}
			return false;
		}
	};
}

#endif
