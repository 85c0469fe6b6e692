#include "vertexenumerator.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstring>
#include <thread>
#include <iostream>
#include <cstdlib>
#include <ctime>
#include <numeric>
#include <stdexcept>

#include "../../include/twopaco_hip.h"
#include "streamfastaparser.h"
#include "textpack.h"

namespace TwoPaCo
{
	const int64_t INVALID_VERTEX = INT64_MAX;

	namespace
	{
		const uint64_t BINS_COUNT = uint64_t(1) << 24;  // reference vertexenumerator.h:471

		// TWOPACO_TIMING=1: millisecond phase timings on stderr (the log keeps the reference's whole seconds)
		struct PhaseTimer
		{
			bool on;
			std::chrono::steady_clock::time_point t;
			PhaseTimer() : on(std::getenv("TWOPACO_TIMING") != 0), t(std::chrono::steady_clock::now()) {}
			void Lap(const char * what)
			{
				std::chrono::steady_clock::time_point now = std::chrono::steady_clock::now();
				if (on) std::cerr << "[timing] " << what << ": " << std::chrono::duration<double, std::milli>(now - t).count() << " ms" << std::endl;
				t = now;
			}
		};

		class HipVertexEnumerator : public VertexEnumerator
		{
		public:
			HipVertexEnumerator() : ctx_(0), vertices_(0) {}
			~HipVertexEnumerator()
			{
				if (ctx_) tpc_ctx_destroy(ctx_);
			}

			size_t GetVerticesCount() const { return vertices_; }
			int64_t GetId(const std::string & vertex) const
			{
				if (vertex.size() < seed_.VertexLength()) return INVALID_VERTEX;
				return tpc_get_id(ctx_, vertex.c_str());
			}

			const VertexRollingHashSeed & GetHashSeed() const { return seed_; }

			void Check(int rc, const char * what)
			{
				if (rc != 0)
				{
					std::string msg = ctx_ ? tpc_last_error(ctx_) : "";
					throw std::runtime_error(std::string(what) + (msg.empty() ? "" : ": " + msg));
				}
			}

			// The body of the reference constructor, vertexenumerator.h:122-466.
			void Run(const std::vector<std::string> & fileName,
				size_t vertexLength,
				size_t filterSize,
				size_t hashFunctions,
				size_t rounds,
				size_t threads,
				size_t abundance,
				const std::string & outFileName,
				std::ostream & logStream,
				const EnumeratorOptions & options)
			{
				if (filterSize < 2 || filterSize > 62)
				{
					throw std::runtime_error("Unsupported filter size");
				}

				if (rounds < 1)
				{
					throw std::runtime_error("The number of rounds must be positive");
				}

				const size_t capacity = (vertexLength + 4 + 31) / 32;  // CalculateNeededCapacity
				if (capacity >= 20)
				{
					throw std::runtime_error("The value of K is too big. Please refer to documentaion how to increase the max supported value of K.");
				}

				const uint64_t realSize = uint64_t(1) << filterSize;
				logStream << "Threads = " << threads << std::endl;
				logStream << "Vertex length = " << vertexLength << std::endl;
				logStream << "Hash functions = " << hashFunctions << std::endl;
				logStream << "Filter size = " << realSize << std::endl;
				logStream << "Capacity = " << capacity << std::endl;
				logStream << "Files: " << std::endl;
				for (const std::string & fn : fileName)
				{
					logStream << fn << std::endl;
				}

				PhaseTimer timer;
				std::vector<uint64_t> table = MakeSeedTable(hashFunctions, filterSize, options.pinnedSeed, options.seed);
				seed_ = VertexRollingHashSeed(hashFunctions, vertexLength, filterSize, table);

				PackedText text;
				PackFastaFiles(fileName, threads, text);
				timer.Lap("parse + pack FASTA");

				int rc = tpc_ctx_create(options.device, &ctx_);
				if (rc != 0)
				{
					throw std::runtime_error("Can't create a GPU context (no MI355X visible?)");
				}

				Check(tpc_set_option(ctx_, "insert_test_first", options.insertTestFirst ? 1 : 0), "set_option");
				// a one-shot run never amortises device allocations, and on MI355X hipMalloc gets slow
				// (~25 ms per GiB) beyond the first ~48 GiB: keep the partition buffers small and batch
				Check(tpc_set_option(ctx_, "part_budget_bytes", int64_t(20) << 30), "set_option");
				Check(tpc_set_params(ctx_, int(vertexLength), int(filterSize), int(hashFunctions), table.data()), "set_params");
				Check(tpc_seq_upload(ctx_, text.bases.data(), text.nmask.data(), text.length), "seq_upload");
				timer.Lap("context + upload");

				Check(tpc_run_begin(ctx_), "run_begin");

				// records the reference dispatches: at least k bases (vertexenumerator.h:1177)
				std::vector<uint64_t> dispStart, dispLength;
				for (size_t r = 0; r < text.recStart.size(); r++)
				{
					if (text.recLength[r] >= vertexLength)
					{
						dispStart.push_back(text.recStart[r]);
						dispLength.push_back(text.recLength[r]);
					}
				}

				const uint64_t BIN_SIZE = std::max(uint64_t(1), realSize / BINS_COUNT);
				std::vector<uint32_t> binCounter;
				double roundSize = 0;
				if (rounds > 1)
				{
					logStream << "Splitting the input kmers set..." << std::endl;
					binCounter.resize(BINS_COUNT);
					Check(tpc_pass1_split_hist(ctx_, dispStart.data(), dispLength.data(), uint32_t(dispStart.size()), binCounter.data()), "split_hist");
					roundSize = double(std::accumulate(binCounter.begin(), binCounter.end(), size_t(0))) / rounds;
				}

				logStream << std::string(80, '-') << std::endl;
				uint64_t low = 0;
				uint64_t high = realSize;
				uint64_t lowBoundary = 0;
				uint64_t verticesCount = 0;
				time_t mark;
				for (size_t round = 0; round < rounds; round++)
				{
					mark = time(0);
					if (rounds > 1)
					{
						// reference vertexenumerator.h:234-250
						uint64_t accumulated = binCounter[lowBoundary];
						for (++lowBoundary; lowBoundary < BINS_COUNT; ++lowBoundary)
						{
							if (accumulated <= roundSize || round + 1 == rounds)
							{
								accumulated += binCounter[lowBoundary];
							}
							else
							{
								break;
							}
						}

						high = lowBoundary * BIN_SIZE;
					}
					else
					{
						high = realSize;
					}

					logStream << "Round " << round << ", " << low << ":" << high << std::endl;
					logStream << "Pass\tFilling\tFiltering" << std::endl << "1\t";
					Check(tpc_filter_reset(ctx_), "filter_reset");
					uint64_t kmers = 0;
					Check(tpc_pass1_insert(ctx_, low, high, &kmers), "pass1_insert");
					logStream << time(0) - mark << "\t";
					mark = time(0);
					uint64_t marks = 0;
					Check(tpc_pass1_query(ctx_, low, high, &marks), "pass1_query");
					logStream << time(0) - mark << "\t" << std::endl;

					mark = time(0);
					logStream << "2\t";
					uint64_t truePositives = 0, falsePositives = 0, hashTableSize = 0;
					Check(tpc_pass2_filter(ctx_, abundance, &truePositives, &falsePositives, &hashTableSize), "pass2_filter");
					logStream << time(0) - mark << "\t";
					mark = time(0);
					logStream << time(0) - mark << std::endl;
					logStream << "True junctions count = " << truePositives << std::endl;
					logStream << "False junctions count = " << falsePositives << std::endl;
					logStream << "Hash table size = " << hashTableSize << std::endl;
					logStream << "Candidate marks count = " << marks << std::endl;
					logStream << std::string(80, '-') << std::endl;
					verticesCount += truePositives;
					low = high + 1;
				}

				timer.Lap("rounds (insert, query, exact filter)");
				mark = time(0);
				uint64_t junctions = 0;
				Check(tpc_junctions_finalize(ctx_, &junctions), "junctions_finalize");
				vertices_ = junctions;
				logStream << "Reallocating bifurcations time: " << time(0) - mark << std::endl;

				mark = time(0);
				uint64_t marked = 0, valid = 0;
				Check(tpc_emit(ctx_, &marked, &valid), "emit");
				std::vector<uint64_t> g(marked);
				std::vector<int64_t> id(marked);
				Check(tpc_emit_fetch(ctx_, g.data(), id.data()), "emit_fetch");
				timer.Lap("sort + id lookup + fetch");

				// EdgeConstructionWorker, reference vertexenumerator.h:927-958, in (sequence, position)
				// order -- the order the reference's -t 1 run assigns stub ids in.  The byte stream is what
				// JunctionPositionWriter::WriteJunction would produce (junctionapi.h): 12-byte records, one
				// separator per sequence-id step.  Plan: a sequential walk over the records fixes every
				// output offset (records, stubs, separators); the records are then formatted by `threads`
				// workers and written with one call.
				uint64_t occurence = 0;
				uint64_t currentStubVertexId = verticesCount + 42;  // vertexenumerator.h:419
				struct Piece { size_t begin, end; uint64_t first; uint64_t offset; };  // marks [begin,end) of one record -> byte offset
				std::vector<Piece> pieces;
				std::vector<char> out;
				const size_t RECORD = sizeof(uint32_t) + sizeof(int64_t);
				auto put = [&out](uint64_t offset, uint32_t p, int64_t v)
				{
					std::memcpy(&out[offset], &p, sizeof(p));
					std::memcpy(&out[offset + sizeof(p)], &v, sizeof(v));
				};

				// pass 1 (parallel): number of valid ids in each block of marks
				const size_t BLOCK = size_t(1) << 16;
				const size_t blocks = (marked + BLOCK - 1) / BLOCK;
				std::vector<uint64_t> validBefore(blocks + 1, 0);
				const size_t workers = std::max<size_t>(1, std::min<size_t>(threads, 64));
				{
					std::vector<std::thread> pool;
					for (size_t t = 0; t < workers; t++)
					{
						pool.emplace_back([&, t]()
						{
							for (size_t bl = t; bl < blocks; bl += workers)
							{
								uint64_t n = 0;
								const size_t e = std::min(marked, (bl + 1) * BLOCK);
								for (size_t i = bl * BLOCK; i < e; i++) n += id[i] != INVALID_VERTEX;
								validBefore[bl + 1] = n;
							}
						});
					}
					for (std::thread & th : pool) th.join();
				}
				for (size_t bl = 0; bl < blocks; bl++) validBefore[bl + 1] += validBefore[bl];
				auto validUpTo = [&](size_t i)  // valid ids among marks [0, i)
				{
					const size_t bl = i / BLOCK;
					uint64_t n = validBefore[bl];
					for (size_t j = bl * BLOCK; j < i; j++) n += id[j] != INVALID_VERTEX;
					return n;
				};

				// pass 2 (sequential over the records): offsets, stubs, separators
				struct Fixed { uint64_t offset; uint32_t pos; int64_t id; };
				std::vector<Fixed> fixed;
				uint64_t offset = 0;
				uint32_t nowChr = 0;
				size_t cur = 0;
				for (size_t r = 0; r < text.recStart.size(); r++)
				{
					const uint64_t len = text.recLength[r];
					if (len < vertexLength)
					{
						continue;
					}

					const uint64_t first = text.recStart[r];
					const uint64_t last = first + len - vertexLength;
					cur = size_t(std::lower_bound(g.begin() + cur, g.end(), first) - g.begin());
					const size_t end = size_t(std::upper_bound(g.begin() + cur, g.end(), last) - g.begin());
					const bool firstValid = cur < end && g[cur] == first && id[cur] != INVALID_VERTEX;
					const bool lastValid = cur < end && g[end - 1] == last && id[end - 1] != INVALID_VERTEX;
					for (; nowChr < r; ++nowChr)  // JunctionPositionWriter: one separator per sequence-id step
					{
						fixed.push_back(Fixed{offset, UINT32_MAX, INT64_MAX});
						offset += RECORD;
					}

					// first / last k-mer of the sequence without a junction id get a stub id (vertexenumerator.h:942-948)
					if (!firstValid)
					{
						fixed.push_back(Fixed{offset, 0, int64_t(currentStubVertexId++)});
						offset += RECORD;
						++occurence;
					}

					const uint64_t nValid = validUpTo(end) - validUpTo(cur);
					pieces.push_back(Piece{cur, end, first, offset});
					offset += nValid * RECORD;
					occurence += nValid;
					if (last != first && !lastValid)
					{
						fixed.push_back(Fixed{offset, uint32_t(last - first), int64_t(currentStubVertexId++)});
						offset += RECORD;
						++occurence;
					}

					cur = end;
				}

				out.resize(offset);
				for (const Fixed & f : fixed) put(f.offset, f.pos, f.id);

				// pass 3 (parallel): format the junction records of every piece, in blocks of marks
				struct Task { size_t begin, end; uint64_t first; uint64_t offset; };
				std::vector<Task> tasks;
				for (const Piece & p : pieces)
				{
					uint64_t off = p.offset;
					for (size_t b0 = p.begin; b0 < p.end;)
					{
						const size_t e0 = std::min(p.end, (b0 / BLOCK + 1) * BLOCK);
						tasks.push_back(Task{b0, e0, p.first, off});
						off += (validUpTo(e0) - validUpTo(b0)) * RECORD;
						b0 = e0;
					}
				}
				{
					std::vector<std::thread> pool;
					for (size_t t = 0; t < workers; t++)
					{
						pool.emplace_back([&, t]()
						{
							for (size_t k = t; k < tasks.size(); k += workers)
							{
								uint64_t off = tasks[k].offset;
								for (size_t i = tasks[k].begin; i < tasks[k].end; i++)
								{
									if (id[i] != INVALID_VERTEX)
									{
										put(off, uint32_t(g[i] - tasks[k].first), id[i]);
										off += RECORD;
									}
								}
							}
						});
					}
					for (std::thread & th : pool) th.join();
				}

				{
					std::FILE * f = std::fopen(outFileName.c_str(), "wb");
					if (!f)
					{
						throw std::runtime_error("Can't create the output file");
					}

					const bool ok = out.empty() || std::fwrite(out.data(), 1, out.size(), f) == out.size();
					if (std::fclose(f) != 0 || !ok)
					{
						throw std::runtime_error("Can't write to the output file");
					}
				}

				timer.Lap("merge + write junction stream");
				logStream << "True marks count: " << occurence << std::endl;
				logStream << "Edges construction time: " << time(0) - mark << std::endl;
				logStream << std::string(80, '-') << std::endl;
			}

		private:
			tpc_ctx * ctx_;
			size_t vertices_;
			VertexRollingHashSeed seed_;
		};
	}

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
		const EnumeratorOptions & options)
	{
		(void)tmpFileName;  // candidate masks and junction keys stay in HBM: no scratch files
		std::unique_ptr<HipVertexEnumerator> ret(new HipVertexEnumerator());
		ret->Run(fileName, vertexLength, filterSize, hashFunctions, rounds, threads, abundance, outFileName, logStream, options);
		return std::unique_ptr<VertexEnumerator>(ret.release());
	}

	std::unique_ptr<VertexEnumerator> CreateEnumerator(const std::vector<std::string> & fileName,
		size_t vertexLength,
		size_t filterSize,
		size_t hashFunctions,
		size_t rounds,
		size_t threads,
		size_t abundance,
		const std::string & tmpFileName,
		const std::string & outFileName,
		std::ostream & logStream)
	{
		EnumeratorOptions options;
		if (const char * s = std::getenv("TWOPACO_SEED"))
		{
			options.pinnedSeed = true;
			options.seed = std::strtoull(s, 0, 0);
		}

		if (const char * d = std::getenv("TWOPACO_DEVICE"))
		{
			options.device = std::atoi(d);
		}

		return CreateEnumerator(fileName, vertexLength, filterSize, hashFunctions, rounds, threads, abundance, tmpFileName, outFileName, logStream, options);
	}
}
