#include "textpack.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <functional>
#include <chrono>
#include <cstdio>
#include <cstdlib>
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

		for (size_t f = 0; f < fileName.size(); f++)
		{
			if (parsed[f].error)
			{
				throw *parsed[f].error;
			}
		}

		// Layout of T = N rec0 N rec1 N ...: every record's start follows from the lengths alone, so the records are
		// placed by all threads at once.  A destination word inside one record is written by that record's thread
		// only; the first and last word of its span may be shared with the neighbours and are OR-ed atomically.
		struct Placed { const Record * rec; uint64_t start; };
		std::vector<Placed> placed;
		uint64_t length = 1;
		for (Parsed & p : parsed)
		{
			for (Record & rec : p.records)
			{
				placed.push_back(Placed{&rec, length});
				length += rec.n + 1;
			}
		}

		const uint64_t words = (length + 31) / 32;
		out.bases.clear();
		out.nmask.clear();
		out.bases.resize(words);  // uninitialised (DefaultInitAllocator): zeroed below, in parallel
		out.nmask.resize(words);
		out.recStart.resize(placed.size());
		out.recLength.resize(placed.size());
		out.length = length;
		uint64_t * const B = out.bases.data();
		uint32_t * const M = out.nmask.data();
		auto setN = [M](uint64_t g) { __atomic_fetch_or(&M[g >> 5], uint32_t(1) << (g & 31), __ATOMIC_RELAXED); };
		auto place = [&](const Placed & p)
		{
			const uint64_t n = p.rec->n;
			setN(p.start + n);  // trailing separator
			if (n == 0) return;
			const uint64_t * b = p.rec->bases.data();
			const uint32_t * m = p.rec->nmask.data();
			const uint64_t w0 = p.start >> 5;
			const unsigned o = unsigned(p.start & 31);
			const uint64_t nw = (n + 31) / 32;
			const uint64_t last = (p.start + n - 1) >> 5;  // last destination word holding a character of the record
			for (uint64_t j = w0; j <= last; j++)
			{
				const uint64_t i = j - w0;
				uint64_t vb = 0;
				uint32_t vm = 0;
				if (i < nw)
				{
					vb = b[i] << (2 * o);
					vm = m[i] << o;
				}

				if (o != 0 && i >= 1)
				{
					vb |= b[i - 1] >> (64 - 2 * o);
					vm |= m[i - 1] >> (32 - o);
				}

				if (j == w0 || j == last)
				{
					__atomic_fetch_or(&B[j], vb, __ATOMIC_RELAXED);
					__atomic_fetch_or(&M[j], vm, __ATOMIC_RELAXED);
				}
				else
				{
					B[j] = vb;
					M[j] = vm;
				}
			}
		};

		// phase 1: zero fill in parallel chunks (span ends are OR-ed into); phase 2: place the records
		const size_t team = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
		auto parallel = [team](size_t items, const std::function<void(size_t)> & fn)
		{
			if (team <= 1 || items <= 1)
			{
				for (size_t i = 0; i < items; i++) fn(i);
				return;
			}

			std::atomic<size_t> cursor(0);
			std::vector<std::thread> pool;
			for (size_t t = 0; t < std::min(team, items); t++)
			{
				pool.emplace_back([&]() { for (size_t i = cursor++; i < items; i = cursor++) fn(i); });
			}

			for (std::thread & th : pool) th.join();
		};

		const uint64_t CHUNK = uint64_t(1) << 18;  // words
		parallel(size_t((words + CHUNK - 1) / CHUNK), [&](size_t c)
		{
			const uint64_t a = c * CHUNK, e = std::min(words, a + CHUNK);
			std::memset(B + a, 0, (e - a) * sizeof(uint64_t));
			std::memset(M + a, 0, (e - a) * sizeof(uint32_t));
		});
		setN(0);  // leading separator
		parallel(placed.size(), [&](size_t r)
		{
			place(placed[r]);
			out.recStart[r] = placed[r].start;
			out.recLength[r] = placed[r].rec->n;
		});
	}
}
