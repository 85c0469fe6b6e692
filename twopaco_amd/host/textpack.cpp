#include "textpack.h"

#include <atomic>
#include <memory>
#include <thread>

#include "streamfastaparser.h"

namespace TwoPaCo
{
	void PackedText::BeginText()
	{
		bases.clear();
		nmask.clear();
		recStart.clear();
		recLength.clear();
		length = 0;
		const uint8_t n = 4;
		AppendCodes(&n, 1);
	}

	void PackedText::AppendCodes(const uint8_t * codes, uint64_t n)
	{
		const uint64_t end = length + n;
		const uint64_t words = (end + 31) / 32;
		if (bases.size() < words)
		{
			bases.resize(words, 0);
			nmask.resize(words, 0);
		}

		uint64_t g = length;
		uint64_t i = 0;
		while (i < n)
		{
			const uint64_t w = g >> 5;
			const unsigned o = static_cast<unsigned>(g & 31);
			const unsigned take = static_cast<unsigned>(n - i < 32 - o ? n - i : 32 - o);
			uint64_t b = 0;
			uint32_t m = 0;
			for (unsigned t = 0; t < take; t++)
			{
				const uint8_t c = codes[i + t];
				b |= static_cast<uint64_t>(c & 3) << (2 * t);
				m |= static_cast<uint32_t>(c >> 2) << t;
			}

			bases[w] |= b << (2 * o);
			nmask[w] |= m << o;
			// an N keeps code 0 in the base word
			if (m)
			{
				uint64_t clear = 0;
				for (unsigned t = 0; t < take; t++) if ((m >> t) & 1u) clear |= 3ull << (2 * (t + o));
				bases[w] &= ~clear;
			}

			i += take;
			g += take;
		}

		length = end;
	}

	void PackedText::AppendPacked(const uint64_t * b, const uint32_t * m, uint64_t n)
	{
		if (n == 0) return;
		const uint64_t end = length + n;
		const uint64_t words = (end + 31) / 32;
		if (bases.size() < words)
		{
			bases.resize(words, 0);
			nmask.resize(words, 0);
		}

		const uint64_t w0 = length >> 5;
		const unsigned o = static_cast<unsigned>(length & 31);
		const uint64_t nw = (n + 31) / 32;
		if (o == 0)
		{
			for (uint64_t i = 0; i < nw; i++) { bases[w0 + i] |= b[i]; nmask[w0 + i] |= m[i]; }
		}
		else
		{
			for (uint64_t i = 0; i < nw; i++)
			{
				bases[w0 + i] |= b[i] << (2 * o);
				nmask[w0 + i] |= m[i] << o;
				if (w0 + i + 1 < words)
				{
					bases[w0 + i + 1] |= b[i] >> (64 - 2 * o);
					nmask[w0 + i + 1] |= m[i] >> (32 - o);
				}
			}
		}

		length = end;
	}

	void PackedText::EndRecord(uint64_t recordBases)
	{
		recStart.push_back(length - recordBases);
		recLength.push_back(recordBases);
		const uint8_t n = 4;
		AppendCodes(&n, 1);
	}

	void PackFastaFiles(const std::vector<std::string> & fileName, size_t threads, PackedText & out)
	{
		// per file: the records' code strings
		struct Record { std::vector<uint64_t> bases; std::vector<uint32_t> nmask; uint64_t n; };
		struct Parsed { std::vector<Record> records; std::unique_ptr<StreamFastaParser::Exception> error; };
		std::vector<Parsed> parsed(fileName.size());
		std::atomic<size_t> next(0);
		auto work = [&]()
		{
			for (size_t f = next++; f < fileName.size(); f = next++)
			{
				try
				{
					StreamFastaParser parser(fileName[f]);
					while (parser.ReadRecord())
					{
						parsed[f].records.emplace_back();
						Record & rec = parsed[f].records.back();
						rec.n = parser.ReadSequencePacked(rec.bases, rec.nmask);
					}
				}
				catch (const StreamFastaParser::Exception & e)
				{
					parsed[f].error.reset(new StreamFastaParser::Exception(e.what()));
				}
			}
		};

		size_t workers = threads < 1 ? 1 : (threads < fileName.size() ? threads : fileName.size());
		if (workers <= 1)
		{
			work();
		}
		else
		{
			std::vector<std::thread> pool;
			for (size_t i = 0; i < workers; i++) pool.emplace_back(work);
			for (std::thread & t : pool) t.join();
		}

		out.BeginText();
		for (size_t f = 0; f < fileName.size(); f++)
		{
			if (parsed[f].error)
			{
				throw *parsed[f].error;
			}

			for (Record & rec : parsed[f].records)
			{
				out.AppendPacked(rec.bases.data(), rec.nmask.data(), rec.n);
				out.EndRecord(rec.n);
				std::vector<uint64_t>().swap(rec.bases);
				std::vector<uint32_t>().swap(rec.nmask);
			}
		}
	}
}
