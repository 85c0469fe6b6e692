#include "streamfastaparser.h"

#include <cctype>
#include <cstdio>
#include <sstream>

#include "dnachar.h"

namespace TwoPaCo
{
	namespace
	{
		// 0..3 = ACGT, 4 = other valid letter, 5 = whitespace, 6 = '>', 7 = invalid
		struct CharClass
		{
			uint8_t t[256];
			CharClass()
			{
				for (int c = 0; c < 256; c++)
				{
					int up = std::toupper(c);
					if (std::isspace(c)) t[c] = 5;
					else if (c == '>') t[c] = 6;
					else if (DnaChar::IsDefinite(static_cast<char>(up))) t[c] = static_cast<uint8_t>(DnaChar::MakeUpChar(static_cast<char>(up)));
					else if (DnaChar::IsValid(static_cast<char>(up))) t[c] = 4;
					else t[c] = 7;
				}
			}
		};
		const CharClass CLASS;
	}

	StreamFastaParser::StreamFastaParser(const std::string & fileName) : pos_(0)
	{
		FILE * f = std::fopen(fileName.c_str(), "rb");
		if (!f)
		{
			throw Exception("Can't open file " + fileName);
		}

		std::fseek(f, 0, SEEK_END);
		long size = std::ftell(f);
		std::fseek(f, 0, SEEK_SET);
		if (size > 0)
		{
			data_.resize(static_cast<size_t>(size));
			size_t got = std::fread(&data_[0], 1, data_.size(), f);
			data_.resize(got);
		}
		else
		{
			// not seekable (pipe): read in blocks
			char buf[1 << 16];
			size_t got;
			while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) data_.append(buf, got);
		}

		std::fclose(f);
	}

	bool StreamFastaParser::ReadRecord()
	{
		if (pos_ >= data_.size())
		{
			return false;
		}

		char ch = data_[pos_++];
		if (ch != '>')
		{
			throw Exception("The FASTA header should start with a '>', started with '" + std::string(1, ch) + "'");
		}

		size_t end = data_.find('\n', pos_);
		if (end == std::string::npos)
		{
			// header without a newline: the reference keeps the previous header and reaches end of input
			pos_ = data_.size();
			return true;
		}

		std::stringstream ss(data_.substr(pos_, end - pos_));
		currentHeader_.clear();
		ss >> currentHeader_;
		pos_ = end + 1;
		return true;
	}

	bool StreamFastaParser::GetChar(char & ch)
	{
		while (pos_ < data_.size())
		{
			unsigned char c = static_cast<unsigned char>(data_[pos_]);
			uint8_t cls = CLASS.t[c];
			if (cls == 5)
			{
				++pos_;
				continue;
			}

			if (cls == 6)
			{
				return false;
			}

			if (cls == 7)
			{
				throw Exception("Found an invalid character '" + std::string(1, static_cast<char>(c)) + "' in sequence " + currentHeader_);
			}

			++pos_;
			ch = static_cast<char>(std::toupper(c));
			return true;
		}

		return false;
	}

	void StreamFastaParser::ReadSequenceCodes(std::vector<uint8_t> & out)
	{
		const size_t n = data_.size();
		const unsigned char * d = reinterpret_cast<const unsigned char*>(data_.data());
		size_t p = pos_;
		for (; p < n; ++p)
		{
			uint8_t cls = CLASS.t[d[p]];
			if (cls <= 4)
			{
				out.push_back(cls);
			}
			else if (cls == 6)
			{
				break;
			}
			else if (cls == 7)
			{
				pos_ = p;
				throw Exception("Found an invalid character '" + std::string(1, static_cast<char>(d[p])) + "' in sequence " + currentHeader_);
			}
		}

		pos_ = p;
	}

	uint64_t StreamFastaParser::ReadSequencePacked(std::vector<uint64_t> & bases, std::vector<uint32_t> & nmask)
	{
		const size_t n = data_.size();
		const unsigned char * d = reinterpret_cast<const unsigned char*>(data_.data());
		size_t p = pos_;
		uint64_t count = 0, word = 0;
		uint32_t mask = 0;
		unsigned fill = 0;
		bases.clear();
		nmask.clear();
		for (; p < n; ++p)
		{
			const uint8_t cls = CLASS.t[d[p]];
			if (cls <= 4)
			{
				word |= static_cast<uint64_t>(cls & 3) << (2 * fill);
				mask |= static_cast<uint32_t>(cls >> 2) << fill;
				if (++fill == 32)
				{
					bases.push_back(word);
					nmask.push_back(mask);
					word = 0; mask = 0; fill = 0;
				}

				++count;
			}
			else if (cls == 6)
			{
				break;
			}
			else if (cls == 7)
			{
				pos_ = p;
				throw Exception("Found an invalid character '" + std::string(1, static_cast<char>(d[p])) + "' in sequence " + currentHeader_);
			}
		}

		if (fill)
		{
			bases.push_back(word);
			nmask.push_back(mask);
		}

		pos_ = p;
		return count;
	}
}
