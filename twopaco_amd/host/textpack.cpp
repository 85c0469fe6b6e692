#include "textpack.h"

#include <algorithm>
#include <cctype>
#include <sstream>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
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

	namespace
	{
		// character classes of the reference's parser (streamfastaparser.cpp:61-93): 0..3 = ACGT, 4 = another valid
		// letter (N after VertexEnumerator's mapping, vertexenumerator.h:1174), 5 = whitespace, 6 = '>', 7 = invalid
		struct PackClass
		{
			uint8_t t[256];
			PackClass()
			{
				for (int c = 0; c < 256; c++)
				{
					const int up = std::toupper(c);
					if (std::isspace(c)) t[c] = 5;
					else if (c == '>') t[c] = 6;
					else if (up == 'A') t[c] = 0;
					else if (up == 'C') t[c] = 1;
					else if (up == 'G') t[c] = 2;
					else if (up == 'T') t[c] = 3;
					else if (up != 0 && std::strchr("URYKMSWBDHWNXV", up)) t[c] = 4;
					else t[c] = 7;
				}
			}
		};
		const PackClass PACK;

		typedef std::vector<char, DefaultInitAllocator<char> > Bytes;

		struct InputFile
		{
			const char * data;      // the file's bytes: a private mapping of the page cache, or `owned` for pipes
			size_t size;
			bool mapped;
			Bytes owned;
			std::string error;      // first error of the file, in byte order
			size_t errorAt;
			InputFile() : data(0), size(0), mapped(false), errorAt(size_t(-1)) {}
			void Release()
			{
				if (mapped && data) ::munmap(const_cast<char*>(data), size);
				data = 0;
				mapped = false;
				Bytes().swap(owned);
			}

			void Fail(size_t at, const std::string & what)
			{
				if (at < errorAt)
				{
					errorAt = at;
					error = what;
				}
			}
		};

		// a run of sequence bytes of one record, packed on its own and placed later
		struct Piece
		{
			size_t file, record;      // record: index over all records of all files
			size_t begin, end;        // byte range in the file
			std::string header;
			uint64_t * bases;         // room for (end - begin) / 32 + 2 words each in the arenas of PackFastaFiles: a piece
			uint32_t * nmask;         // allocates nothing (contig-level inputs have 10^5 pieces)
			uint64_t n;
			Piece() : bases(0), nmask(0), n(0) {}
		};

		void PackPiece(const InputFile & in, Piece & p, size_t & errorAt, std::string & error)
		{
			const unsigned char * d = reinterpret_cast<const unsigned char*>(in.data);
			uint64_t word = 0;
			uint32_t mask = 0;
			unsigned fill = 0;
			uint64_t * const B = p.bases;
			uint32_t * const M = p.nmask;
			size_t w = 0;
			uint64_t n = 0;
			for (size_t i = p.begin; i < p.end; ++i)
			{
				const uint8_t cls = PACK.t[d[i]];
				if (cls <= 4)
				{
					word |= uint64_t(cls & 3) << (2 * fill);
					mask |= uint32_t(cls >> 2) << fill;
					if (++fill == 32)
					{
						B[w] = word;
						M[w] = mask;
						++w;
						word = 0; mask = 0; fill = 0;
					}

					++n;
				}
				else if (cls == 7)
				{
					errorAt = i;
					error = "Found an invalid character '" + std::string(1, char(d[i])) + "' in sequence " + p.header;
					p.n = n;
					return;
				}
			}

			if (fill)
			{
				B[w] = word;
				M[w] = mask;
			}

			p.n = n;
		}
	}

	// All files are read, indexed (records end at the next '>', streamfastaparser.cpp:61-76), cut into pieces of a few MiB
	// and packed by `threads` workers, whatever the number of files and the size of their records; the layout of
	// T = N rec0 N rec1 N ... follows from the piece lengths, so the pieces are then placed by all threads at once.
	void PackFastaFiles(const std::vector<std::string> & fileName, size_t threads, PackedText & out)
	{
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

		const bool timing = std::getenv("TWOPACO_TIMING") != 0;
		std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
		auto lap = [&](const char * what)
		{
			if (!timing) return;
			const std::chrono::steady_clock::time_point t1 = std::chrono::steady_clock::now();
			std::fprintf(stderr, "[timing]   parse: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
			t0 = t1;
		};

		size_t pieceBytes = size_t(4) << 20;
		if (const char * e = std::getenv("TWOPACO_PARSE_PIECE")) pieceBytes = std::max<size_t>(1, size_t(std::atoll(e)));

		// 1. map: the files' page-cache pages are mapped (no copy); pipes and devices are read to the end
		std::vector<InputFile> input(fileName.size());
		parallel(fileName.size(), [&](size_t f)
		{
			InputFile & in = input[f];
			const int fd = ::open(fileName[f].c_str(), O_RDONLY);
			struct stat st;
			if (fd < 0 || ::fstat(fd, &st) != 0)
			{
				if (fd >= 0) ::close(fd);
				in.Fail(0, "Can't open file " + fileName[f]);
				return;
			}

			if (S_ISREG(st.st_mode) && st.st_size > 0)
			{
				void * m = ::mmap(0, size_t(st.st_size), PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
				if (m != MAP_FAILED)
				{
					in.data = static_cast<const char*>(m);
					in.size = size_t(st.st_size);
					in.mapped = true;
					::close(fd);
					return;
				}
			}

			char buf[1 << 16];
			for (ssize_t got; (got = ::read(fd, buf, sizeof(buf))) > 0;) in.owned.insert(in.owned.end(), buf, buf + got);
			::close(fd);
			in.data = in.owned.data();
			in.size = in.owned.size();
		});
		lap("read");

		// 2. index the records of every file (memchr speed) and cut their sequence bytes into pieces
		std::vector<std::vector<Piece> > filePieces(fileName.size());
		std::vector<size_t> fileRecords(fileName.size(), 0);
		parallel(fileName.size(), [&](size_t f)
		{
			InputFile & in = input[f];
			if (!in.error.empty()) return;
			const char * d = in.data;
			const size_t size = in.size;
			size_t pos = 0;
			while (pos < size)
			{
				if (d[pos] != '>')
				{
					in.Fail(pos, "The FASTA header should start with a '>', started with '" + std::string(1, d[pos]) + "'");
					return;
				}

				const char * nl = static_cast<const char*>(std::memchr(d + pos + 1, '\n', size - pos - 1));
				Piece first;
				first.file = f;
				first.record = fileRecords[f]++;
				if (!nl)
				{
					// header without a newline: a record without sequence (the reference reaches the end of input)
					first.begin = first.end = size;
					filePieces[f].push_back(first);
					return;
				}

				std::stringstream ss(std::string(d + pos + 1, nl));
				ss >> first.header;
				const size_t seqBegin = size_t(nl - d) + 1;
				const char * gt = static_cast<const char*>(std::memchr(d + seqBegin, '>', size - seqBegin));
				const size_t seqEnd = gt ? size_t(gt - d) : size;
				size_t b = seqBegin;
				do
				{
					Piece p;
					p.file = f;
					p.record = first.record;
					p.header = first.header;
					p.begin = b;
					p.end = std::min(seqEnd, b + pieceBytes);
					filePieces[f].push_back(p);
					b = p.end;
				}
				while (b < seqEnd);
				pos = seqEnd;
			}
		});

		lap("index");
		std::vector<Piece> piece;
		{
			size_t recordBase = 0;
			for (size_t f = 0; f < fileName.size(); f++)
			{
				for (Piece & p : filePieces[f])
				{
					p.record += recordBase;
					piece.push_back(std::move(p));
				}

				recordBase += fileRecords[f];
				std::vector<Piece>().swap(filePieces[f]);
			}
		}

		// 3. pack the pieces, each into its own span of two arenas (no allocation per piece; the pages are touched first by the
		// packing threads)
		std::vector<size_t> arenaAt(piece.size() + 1, 0);
		for (size_t i = 0; i < piece.size(); i++) arenaAt[i + 1] = arenaAt[i] + (piece[i].end - piece[i].begin) / 32 + 2;
		std::unique_ptr<uint64_t[]> arenaBases(new uint64_t[arenaAt[piece.size()] + 1]);
		std::unique_ptr<uint32_t[]> arenaMask(new uint32_t[arenaAt[piece.size()] + 1]);
		for (size_t i = 0; i < piece.size(); i++)
		{
			piece[i].bases = arenaBases.get() + arenaAt[i];
			piece[i].nmask = arenaMask.get() + arenaAt[i];
		}

		std::vector<size_t> pieceErrorAt(piece.size(), size_t(-1));
		std::vector<std::string> pieceError(piece.size());
		// (contig-level assemblies: thousands of small pieces -- a task is a block of consecutive pieces, not one piece)
		const size_t block = std::max<size_t>(1, piece.size() / (team * 8));
		auto blocks = [&](const std::function<void(size_t)> & fn)
		{
			parallel((piece.size() + block - 1) / block, [&](size_t b) { for (size_t i = b * block; i < std::min(piece.size(), (b + 1) * block); i++) fn(i); });
		};

		blocks([&](size_t i) { PackPiece(input[piece[i].file], piece[i], pieceErrorAt[i], pieceError[i]); });
		lap("pack");
		for (size_t i = 0; i < piece.size(); i++)
		{
			if (pieceErrorAt[i] != size_t(-1)) input[piece[i].file].Fail(pieceErrorAt[i], pieceError[i]);
		}

		parallel(fileName.size(), [&](size_t f) { input[f].Release(); });
		for (size_t f = 0; f < fileName.size(); f++)
		{
			if (!input[f].error.empty())
			{
				throw StreamFastaParser::Exception(input[f].error);
			}
		}

		lap("free input");
		// 4. layout: a record's pieces follow each other, records are separated by one N
		size_t records = 0;
		for (const Piece & p : piece) records = std::max(records, p.record + 1);
		out.recStart.assign(records, 0);
		out.recLength.assign(records, 0);
		std::vector<uint64_t> start(piece.size());
		uint64_t length = 1;
		for (size_t i = 0; i < piece.size(); i++)
		{
			if (i > 0 && piece[i].record != piece[i - 1].record) ++length;  // separator after the previous record
			if (i == 0 || piece[i].record != piece[i - 1].record) out.recStart[piece[i].record] = length;
			start[i] = length;
			length += piece[i].n;
			out.recLength[piece[i].record] += piece[i].n;
		}

		if (!piece.empty()) ++length;  // trailing separator
		const uint64_t words = (length + 31) / 32;
		out.bases.clear();
		out.nmask.clear();
		out.bases.resize(words);  // uninitialised (DefaultInitAllocator): zeroed below, in parallel
		out.nmask.resize(words);
		out.length = length;
		uint64_t * const B = out.bases.data();
		uint32_t * const M = out.nmask.data();
		auto setN = [M](uint64_t g) { __atomic_fetch_or(&M[g >> 5], uint32_t(1) << (g & 31), __ATOMIC_RELAXED); };
		// A destination word inside one piece is written by that piece's thread only; the first and last word of its span
		// may be shared with the neighbours and are OR-ed atomically.
		auto place = [&](size_t idx)
		{
			const Piece & p = piece[idx];
			const uint64_t n = p.n;
			if (idx + 1 == piece.size() || piece[idx + 1].record != p.record) setN(start[idx] + n);  // separator after the record
			if (n == 0) return;
			const uint64_t * b = p.bases;
			const uint32_t * m = p.nmask;
			const uint64_t w0 = start[idx] >> 5;
			const unsigned o = unsigned(start[idx] & 31);
			const uint64_t nw = (n + 31) / 32;
			const uint64_t last = (start[idx] + n - 1) >> 5;
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

		const uint64_t CHUNK = uint64_t(1) << 18;  // words
		parallel(size_t((words + CHUNK - 1) / CHUNK), [&](size_t c)
		{
			const uint64_t a = c * CHUNK, e = std::min(words, a + CHUNK);
			std::memset(B + a, 0, (e - a) * sizeof(uint64_t));
			std::memset(M + a, 0, (e - a) * sizeof(uint32_t));
		});
		lap("zero");
		setN(0);  // leading separator
		blocks(place);
		lap("place");
	}
}
