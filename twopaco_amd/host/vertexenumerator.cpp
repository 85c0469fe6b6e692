#include "vertexenumerator.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstring>
#include <thread>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <iostream>
#include <cstdlib>
#include <ctime>
#include <numeric>
#include <stdexcept>

#include "../../include/twopaco_hip.h"
#include "multigpu.h"
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

		// ---- Bloom filter checkpoint (EnumeratorOptions::saveFilter / loadFilter; the reference's commented-out
		// ReloadBloomFilter, reference vertexenumerator.h:29,113-121 -- there the dump was ConcurrentBitVector::WriteToFile of
		// the whole vector, concurrentbitvector.cpp:59-67).  One file per round: header, q x 5 hash table, filter words.
		// Bloom filter checkpoint (--save-filter / --load-filter; the reference's commented-out ReloadBloomFilter, vertexenumerator.h:29,113-121).
		// The header is written field by field, little-endian (no raw struct: the format does not depend on the host ABI), and ties
		// the filter to the INPUT as well as to the parameters: a filter saved from other FASTA files would load cleanly and silently
		// drop junctions (Bloom false negatives), so the text's length, record count and a checksum of the packed text travel with it.
		struct FilterFileHeader
		{
			uint32_t version, k, bits, q, round, rounds;
			uint64_t low, high, words;
			uint64_t textLength, textRecords, textChecksum;  // version 2
			uint32_t shard, shards;                          // version 2: shard `shard` of `shards` (address-sharded filter: one file per rank)
		};

		const uint32_t FILTER_FILE_VERSION = 2;

		void PutLe(std::vector<unsigned char> & out, uint64_t v, int bytes)
		{
			for (int i = 0; i < bytes; i++) out.push_back((unsigned char)(v >> (8 * i)));
		}

		uint64_t GetLe(const unsigned char * p, int bytes)
		{
			uint64_t v = 0;
			for (int i = 0; i < bytes; i++) v |= uint64_t(p[i]) << (8 * i);
			return v;
		}

		const size_t FILTER_HEADER_BYTES = 8 + 6 * 4 + 6 * 8 + 2 * 4;

		std::vector<unsigned char> EncodeFilterHeader(const FilterFileHeader & h)
		{
			std::vector<unsigned char> out;
			out.insert(out.end(), "TPCBLOOM", "TPCBLOOM" + 8);
			PutLe(out, h.version, 4); PutLe(out, h.k, 4); PutLe(out, h.bits, 4); PutLe(out, h.q, 4); PutLe(out, h.round, 4); PutLe(out, h.rounds, 4);
			PutLe(out, h.low, 8); PutLe(out, h.high, 8); PutLe(out, h.words, 8);
			PutLe(out, h.textLength, 8); PutLe(out, h.textRecords, 8); PutLe(out, h.textChecksum, 8);
			PutLe(out, h.shard, 4); PutLe(out, h.shards, 4);
			return out;
		}

		// Cheap fingerprint of the packed text: its length, every record's length, and up to 2^20 evenly spaced words of the bases and
		// of the N mask (reading all of a human-scale text would cost more than the filter upload it guards).
		uint64_t TextChecksum(const PackedText & text)
		{
			auto mix = [](uint64_t h, uint64_t v)
			{
				h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
				h *= 0xBF58476D1CE4E5B9ull;
				return h ^ (h >> 31);
			};

			uint64_t h = mix(0x5450434B50540001ull, text.length);
			for (uint64_t len : text.recLength) h = mix(h, len);
			const uint64_t words = (text.length + 31) / 32;
			const uint64_t stride = std::max<uint64_t>(1, words >> 20);
			for (uint64_t w = 0; w < words; w += stride)
			{
				h = mix(h, text.bases[w]);
				h = mix(h, text.nmask[w]);
			}

			return h;
		}

		// one file per round; a filter sharded by bit address over several GPUs: one file per round and rank
		std::string FilterFileName(const std::string & base, size_t round, size_t shard = 0, size_t shards = 1)
		{
			std::string name = round == 0 ? base : base + "." + std::to_string(round);
			if (shards > 1) name += ".shard" + std::to_string(shard) + "of" + std::to_string(shards);
			return name;
		}

		void ReadFilterHeader(std::FILE * f, const std::string & name, FilterFileHeader & h, std::vector<uint64_t> & table)
		{
			unsigned char raw[FILTER_HEADER_BYTES];
			if (std::fread(raw, 1, sizeof(raw), f) != sizeof(raw) || std::memcmp(raw, "TPCBLOOM", 8) != 0)
			{
				throw std::runtime_error("Not a Bloom filter checkpoint: " + name);
			}

			const unsigned char * p = raw + 8;
			h.version = uint32_t(GetLe(p, 4)); h.k = uint32_t(GetLe(p + 4, 4)); h.bits = uint32_t(GetLe(p + 8, 4)); h.q = uint32_t(GetLe(p + 12, 4));
			h.round = uint32_t(GetLe(p + 16, 4)); h.rounds = uint32_t(GetLe(p + 20, 4));
			h.low = GetLe(p + 24, 8); h.high = GetLe(p + 32, 8); h.words = GetLe(p + 40, 8);
			h.textLength = GetLe(p + 48, 8); h.textRecords = GetLe(p + 56, 8); h.textChecksum = GetLe(p + 64, 8);
			h.shard = uint32_t(GetLe(p + 72, 4)); h.shards = uint32_t(GetLe(p + 76, 4));
			if (h.version != FILTER_FILE_VERSION || h.q == 0 || h.q > 64 || h.shards == 0 || h.shard >= h.shards)
			{
				throw std::runtime_error("Not a Bloom filter checkpoint of this version: " + name);
			}

			table.resize(size_t(h.q) * 5);
			unsigned char cell[8];
			for (uint64_t & t : table)
			{
				if (std::fread(cell, 1, 8, f) != 8) throw std::runtime_error("Truncated Bloom filter checkpoint: " + name);
				t = GetLe(cell, 8);
			}
		}

		class HipVertexEnumerator : public VertexEnumerator
		{
		public:
			HipVertexEnumerator() : ctx_(0), vertices_(0) {}
			~HipVertexEnumerator()
			{
				for (ShardedRank & r : peers_)
				{
					r.Release();
					if (r.ctx && r.ctx != ctx_) tpc_ctx_destroy(r.ctx);
				}

				if (ctx_) tpc_ctx_destroy(ctx_);
			}

			size_t GetVerticesCount() const { return vertices_; }
			int64_t GetId(const std::string & vertex) const
			{
				if (vertex.size() < seed_.VertexLength() || vertices_ == 0) return INVALID_VERTEX;
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

				// partition buffers per tile batch: a cold process pays for every GiB it allocates (hipMalloc gets slow,
				// ~25 ms per GiB, beyond the first ~48 GiB), an extra batch costs a pass over the filter and, for the 62-genome
				// workload, the deferred apply (the query must fit one batch): 40 GiB measured 0.28 s end to end against 0.35 s at 20.
				// Every rank of a sharded run gets the SAME budget: the batch geometry must agree on all of them.
				const char * budgetGb = std::getenv("TWOPACO_PART_BUDGET_GB");
				// so: whatever keeps filter + buffers inside those first 48 GiB, between 20 and 40 GiB (measured on the 1.12 Gbp / f = 38
				// workload: 16 / 20 / 28 GiB -> 0.82 / 0.76 / 0.83 s)
				const double filterGb = std::ldexp(1.0, int(filterSize) - 33);
				double autoGb = std::max(20.0, std::min(40.0, 48.0 - filterGb));
				// A large filter turns that around (round 6, profiles/r06_e2e_big.txt): every tile batch of the insert reads and rewrites the
				// whole filter (2 x 2^L / 8 bytes at ~5 TB/s: 55 ms at f = 40) in every round, and what looked like a price per allocated GiB
				// was the driver clearing the memory of the process that ran just before (tools/malloc_bench.hip: 0.04 ms per GiB on clean
				// memory, 20-90 after another process's exit) -- at configs[4]'s shape 20 GiB meant 17 batches and 1.0 s per round's insert,
				// against 0.36 s in 4.  From 16 GiB of filter on: room for the insert's entries in about four batches (q x 4 bytes x 2.8 per
				// position, one position per input byte), within 55 % of what the device has left beside the filter and the text.
				if (filterGb >= 16.0)
				{
					double inputGb = 0;
					for (const std::string & fn : fileName)
					{
						struct stat st;
						if (::stat(fn.c_str(), &st) == 0) inputGb += double(st.st_size) / double(1ull << 30);
					}

					const double entriesGb = inputGb * double(hashFunctions) * 4.0 * 2.8;
					const double roomGb = 0.55 * std::max(0.0, 280.0 - filterGb - inputGb * 0.75);
					autoGb = std::max(autoGb, std::min(roomGb, entriesGb / 4.0));
				}

				const int64_t partBudget = int64_t((budgetGb ? std::atof(budgetGb) : autoGb) * double(1ull << 30));
				// More than 16 hash functions run on the closed-form first-pass kernels (csrc/tpc_pass1_anyq.hip), which exist for the
				// whole filter only: such a run takes one GPU whatever --gpus says (and says so).
				// (... unless the filter is replicated -- the combined exchange, decided below from the same inputs: every rank then runs the
				//  closed-form kernels over its chunk of the text and the ranks' dense filters are OR-reduced)
				const char * multiGpuEnvEarly = std::getenv("TWOPACO_MULTIGPU");
				const bool combinedPossible = filterSize <= 38 && !(multiGpuEnvEarly && std::string(multiGpuEnvEarly) == "entries");
				const bool tooManyFunctionsToShard = hashFunctions > 16 && (options.gpus > 1 || options.forceSharded) && !combinedPossible;
				if (tooManyFunctionsToShard) logStream << "Hash functions = " << hashFunctions << " > 16: the Bloom filter is not sharded, running on one GPU" << std::endl;
				const int gpus = tooManyFunctionsToShard ? 1 : std::max(1, options.gpus);
				const bool sharded = !tooManyFunctionsToShard && (gpus > 1 || options.forceSharded);
				// several GPUs: the second pass's exact-filter table is sharded by key hash and the text stays sharded (multigpu.h:
				// ShardedSecondPass / ShardedFinish); TWOPACO_REPLICATED_PASS2=1: union of the candidate masks, then the single-GPU
				// second pass on rank 0, which then keeps the whole text
				const bool shardedPass2 = sharded && std::getenv("TWOPACO_REPLICATED_PASS2") == 0;
				// How the ranks share the filter.  Combined (default while 2^L / 8 bytes fit a GPU with room to spare and the passes have two
				// levels: L <= 38): every rank keeps the whole filter, inserts its chunk of the text locally and only the SET BITS of every
				// slice travel, once (multigpu.cpp:CombinedFirstPass); the query is local.  Entries (larger filters; TWOPACO_MULTIGPU=entries):
				// the filter is cut over the ranks and every hash hit of both passes is routed to the owner of its slice.
				const char * multiGpuEnv = std::getenv("TWOPACO_MULTIGPU");
				const bool combined = sharded && filterSize <= 38 && !(multiGpuEnv && std::string(multiGpuEnv) == "entries");
				// filter slices of 2^20 bits (128 KiB of LDS) unless the filter is too small to give every rank its level-1 buckets:
				// the fan-out 2^(L - slice_bits) is split over two levels and the first must have at least `gpus` buckets
				int logGpus = 0;
				while ((1 << logGpus) < gpus) ++logGpus;
				const int shardSliceBits = int(std::min<int64_t>(20, std::max<int64_t>(6, int64_t(filterSize) - std::max(2, 2 * logGpus))));
				const char * shardPeriodicEnv = std::getenv("TWOPACO_SHARD_PERIODIC");
				const bool shardPeriodic = !(shardPeriodicEnv && shardPeriodicEnv[0] == '0');
				if (gpus > 64 || (gpus & (gpus - 1)))
				{
					throw std::runtime_error("The number of GPUs must be a power of two (the Bloom filter is cut by bit address)");
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
				std::vector<uint64_t> table;
				// (a sharded run checkpoints every rank's shard in its own file, <name>[.<round>].shard<r>of<W>: the shard layout depends on
				//  the number of ranks, so such a checkpoint reloads into a run with the same --gpus only)
				const size_t ckptShards = sharded && !combined ? size_t(gpus) : 1;  // (combined: every rank holds the whole filter -- rank 0 writes it, every rank reads it)
				if (!options.loadFilter.empty())
				{
					// the filter's bits mean something only under the hash tables they were set with: those come from the file
					const std::string first = FilterFileName(options.loadFilter, 0, 0, ckptShards);
					std::FILE * f = std::fopen(first.c_str(), "rb");
					if (!f) throw std::runtime_error("Can't open the Bloom filter checkpoint " + first);
					FilterFileHeader h;
					try { ReadFilterHeader(f, first, h, table); } catch (...) { std::fclose(f); throw; }
					std::fclose(f);
					if (h.k != vertexLength || h.bits != filterSize || h.q != hashFunctions || h.rounds != rounds)
					{
						throw std::runtime_error("The Bloom filter checkpoint was made with other parameters (k = " + std::to_string(h.k) + ", f = " + std::to_string(h.bits) +
							", q = " + std::to_string(h.q) + ", r = " + std::to_string(h.rounds) + ")");
					}
				}
				else
				{
					table = MakeSeedTable(hashFunctions, filterSize, options.pinnedSeed, options.seed);
				}

				seed_ = VertexRollingHashSeed(hashFunctions, vertexLength, filterSize, table);

				// the device context and the filter allocation (HIP start-up, 2^L/8 bytes of hipMalloc) do not depend
				// on the input: they are set up by a second thread while this one parses and packs the FASTA files
				std::string setupError;
				std::thread warm;
				// joined after the upload, or when this scope is left by an exception
				struct ThreadJoiner
				{
					std::thread & t;
					~ThreadJoiner() { if (t.joinable()) t.join(); }
				} warmJoiner{warm};
				std::thread setup([&]()
				{
					try
					{
						PhaseTimer setupTimer;
						if (tpc_ctx_create(options.device, &ctx_) != 0)
						{
							throw std::runtime_error("Can't create a GPU context (no MI355X visible?)");
						}

						setupTimer.Lap("  setup thread: HIP start-up + context");
						if (std::getenv("TWOPACO_TIMING"))
						{
							std::cerr << "[timing]   device memory free at context creation: " << double(tpc_get_stat(ctx_, "device_free_bytes")) / double(1ull << 30) << " of "
								<< double(tpc_get_stat(ctx_, "device_total_bytes")) / double(1ull << 30) << " GiB" << std::endl;
						}

						// the kernels' code objects load on a third thread while the filter is allocated and the text goes up (started
						// beside tpc_ctx_create instead it gains nothing: the runtime's start-up is serial, and stalls of 0.3 s were seen)
						warm = std::thread([this]() { PhaseTimer warmTimer; tpc_warmup(ctx_); warmTimer.Lap("  warm-up thread: code objects"); });

						Check(tpc_set_option(ctx_, "insert_test_first", options.insertTestFirst ? 1 : 0), "set_option");
						Check(tpc_set_option(ctx_, "part_budget_bytes", partBudget), "set_option");
						if (sharded)
						{
							Check(tpc_set_option(ctx_, "slice_bits", shardSliceBits), "set_option");
							// with the second pass sharded by key hash no rank needs more than its chunk of the text, rank 0 included
							if (shardedPass2) Check(tpc_set_option(ctx_, "text_window", 1), "set_option");
							// tracts send nothing (multigpu.cpp:ShardedFirstPass copies their verdicts); TWOPACO_SHARD_PERIODIC=0: every position probes
							Check(tpc_set_option(ctx_, "shard_periodic_skip", shardPeriodic ? 1 : 0), "set_option");
							if (combined) Check(tpc_set_option(ctx_, "replicate_filter", 1), "set_option");
							Check(tpc_shard_config(ctx_, 0, uint32_t(gpus)), "shard_config");
						}

						Check(tpc_set_params(ctx_, int(vertexLength), int(filterSize), int(hashFunctions), table.data()), "set_params");
						setupTimer.Lap("  setup thread: parameters + filter allocation");
						// the partition buffers of the first pass, sized for what the files can hold at most (a base per byte):
						// allocated here, beside the parser, instead of inside the first round
						if (!sharded && options.loadFilter.empty())
						{
							uint64_t bytes = 0;
							for (const std::string & fn : fileName)
							{
								struct stat st;
								if (::stat(fn.c_str(), &st) == 0) bytes += uint64_t(st.st_size) + 2;
							}

							if (bytes > 0) tpc_reserve(ctx_, bytes + 2);
							setupTimer.Lap("  setup thread: partition buffers");
						}
					}
					catch (std::exception & e)
					{
						setupError = e.what();
					}
				});

				PackedText text;
				try
				{
					PackFastaFiles(fileName, threads, text);
				}
				catch (...)
				{
					setup.join();
					if (warm.joinable()) warm.join();
					throw;
				}

				timer.Lap("parse + pack FASTA");
				const uint64_t textChecksum = (options.loadFilter.empty() && options.saveFilter.empty()) ? 0 : TextChecksum(text);
				textFingerprint_[0] = text.length; textFingerprint_[1] = text.recLength.size(); textFingerprint_[2] = textChecksum;
				setup.join();
				if (!setupError.empty())
				{
					if (warm.joinable()) warm.join();
					throw std::runtime_error(setupError);
				}

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

				// Nothing to enumerate (only headers, or every record shorter than k): the reference runs its passes over an
				// empty task stream, prints zero counters and leaves an empty output file.  Same here, without device work.
				const bool nothing = dispStart.empty();
				{
					const int rcUpload = nothing ? 0 : tpc_seq_upload(ctx_, text.bases.data(), text.nmask.data(), text.length);
					if (warm.joinable()) warm.join();  // (letting the rounds start while the last objects load measured the same)
					Check(rcUpload, "seq_upload");
				}
				timer.Lap("context + upload");

				if (!nothing) Check(tpc_run_begin(ctx_), "run_begin");

				// ---- several GPUs: one rank (thread + context) per device, the filter cut by bit address (multigpu.h)
				std::unique_ptr<Transport> net;
				if (sharded && !nothing)
				{
					std::vector<int> devices(gpus);
					for (int r = 0; r < gpus; r++) devices[r] = options.emulateRanks ? options.device : options.device + r;
					net = MakeTransport(devices, options.rccl && !options.emulateRanks);
					logStream << "GPUs = " << gpus << (combined ? " (Bloom filter replicated through set-bit lists, combined by bit-address owner; transport: " : " (Bloom filter sharded by bit address; transport: ")
						<< net->Name() << ")" << std::endl;
					peers_.resize(gpus);
					peers_[0].rank = 0; peers_[0].device = devices[0]; peers_[0].ctx = ctx_;
					// The level-1 regions travel as equal blocks (sized tightly: tpc_shard_plan; the own block is read in place) up to four
					// ranks and packed to their exact sizes from eight on, where the wire is what a pass waits for and a region is small
					// enough for its 6-sigma slack to be ~12 %.  TWOPACO_EXCHANGE=packed / equal overrides (TWOPACO_EQUAL_EXCHANGE=1: equal).
					const char * exch = std::getenv("TWOPACO_EXCHANGE");
					bool compact = gpus >= 8;
					if (exch && !std::strcmp(exch, "packed")) compact = true;
					if ((exch && !std::strcmp(exch, "equal")) || std::getenv("TWOPACO_EQUAL_EXCHANGE") != 0) compact = false;
					for (int r = 0; r < gpus; r++)
					{
						peers_[r].compactExchange = compact;
						peers_[r].shardedSecondPass = shardedPass2;
						peers_[r].combined = combined;
						peers_[r].filterBits = int(filterSize);
					}
					std::vector<std::string> errors(gpus);
					std::vector<std::thread> pool;
					for (int r = 1; r < gpus; r++)
					{
						pool.emplace_back([&, r]()
						{
							try
							{
								ShardedRank & me = peers_[r];
								me.rank = r; me.device = devices[r];
								if (tpc_ctx_create(devices[r], &me.ctx) != 0) throw std::runtime_error("Can't create a GPU context on device " + std::to_string(devices[r]));
								auto check = [&](int rc, const char * what) { if (rc != 0) throw std::runtime_error(std::string(what) + ": " + tpc_last_error(me.ctx)); };
								check(tpc_set_option(me.ctx, "insert_test_first", options.insertTestFirst ? 1 : 0), "set_option");
								check(tpc_set_option(me.ctx, "slice_bits", shardSliceBits), "set_option");
								check(tpc_set_option(me.ctx, "part_budget_bytes", partBudget), "set_option");
								check(tpc_set_option(me.ctx, "text_window", 1), "set_option");  // ranks other than 0 keep only their chunk of the text (rank 0 runs the second pass)
								check(tpc_set_option(me.ctx, "shard_periodic_skip", shardPeriodic ? 1 : 0), "set_option");
								if (combined) check(tpc_set_option(me.ctx, "replicate_filter", 1), "set_option");
								check(tpc_shard_config(me.ctx, uint32_t(r), uint32_t(gpus)), "shard_config");
								check(tpc_set_params(me.ctx, int(vertexLength), int(filterSize), int(hashFunctions), table.data()), "set_params");
								check(tpc_seq_upload(me.ctx, text.bases.data(), text.nmask.data(), text.length), "seq_upload");
							}
							catch (std::exception & e)
							{
								errors[r] = e.what();
							}
						});
					}

					for (std::thread & th : pool) th.join();
					for (const std::string & e : errors) if (!e.empty()) throw std::runtime_error(e);
					timer.Lap("peer contexts + text upload");
				}

				const uint64_t BIN_SIZE = std::max(uint64_t(1), realSize / BINS_COUNT);
				std::vector<uint32_t> binCounter;
				double roundSize = 0;
				const bool rangesFromCheckpoint = !options.loadFilter.empty();  // a checkpoint names the range of the round it holds
				if (rounds > 1 && !rangesFromCheckpoint)
				{
					logStream << "Splitting the input kmers set..." << std::endl;
					binCounter.resize(BINS_COUNT + 1);
					if (sharded)
					{
						// The split pass (InitialFilterFillerWorker, vertexenumerator.h:503-583) needs ONE whole filter as scratch, which no
						// rank of a sharded run holds.  While a whole filter and the whole text still fit one device beside rank 0's shard
						// (f = 40 and 100 human genomes: 128 + 75 GB of 288) the pass runs there, in a context of its own that is gone again
						// before the rounds start: the histogram, hence the "Round n, lo:hi" lines, are then those of a one-GPU run.
						bool measured = false;
						if (!nothing && std::getenv("TWOPACO_ANALYTIC_SPLIT") == 0)
						{
							tpc_ctx * scratch = 0;
							if (tpc_ctx_create(options.device, &scratch) == 0)
							{
								measured = tpc_set_params(scratch, int(vertexLength), int(filterSize), int(hashFunctions), table.data()) == 0 &&
									tpc_seq_upload(scratch, text.bases.data(), text.nmask.data(), text.length) == 0 &&
									tpc_pass1_split_hist(scratch, dispStart.data(), dispLength.data(), uint32_t(dispStart.size()), binCounter.data()) == 0;
								tpc_ctx_destroy(scratch);
							}
						}

						if (!measured && !nothing)
						{
							// Otherwise the histogram is taken in expectation: the vertex hash is the minimum of two well mixed L-bit hashes
							// (density 2(1-x) over [0, 2^L)), so bin b gets the mass of [b, b+1) * BIN_SIZE; the planner below cuts equal shares.
							logStream << "(the split pass does not fit one GPU beside its filter shard: rounds cut at the analytic quantiles of the vertex hash)" << std::endl;
							const double bins = double(BINS_COUNT);
							for (uint64_t b = 0; b < BINS_COUNT; b++)
							{
								const double x0 = double(b) / bins, x1 = double(b + 1) / bins;
								binCounter[b] = uint32_t(((1.0 - (1.0 - x1) * (1.0 - x1)) - (1.0 - (1.0 - x0) * (1.0 - x0))) * 4e9 / 2.0);
							}
						}
					}
					else if (!nothing) Check(tpc_pass1_split_hist(ctx_, dispStart.data(), dispLength.data(), uint32_t(dispStart.size()), binCounter.data()), "split_hist");
					roundSize = double(std::accumulate(binCounter.begin(), binCounter.begin() + BINS_COUNT, size_t(0))) / rounds;
					timer.Lap("split pass (vertex-hash histogram)");
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
					if (rangesFromCheckpoint)
					{
						// the round's range is the one its filter was filled for (the split pass that chose it counts first-seen edges in
						// arrival order, vertexenumerator.h:559-570: a rerun may cut a saturated filter's rounds a few bins away)
						const std::string name = FilterFileName(options.loadFilter, round, 0, ckptShards);
						std::FILE * f = std::fopen(name.c_str(), "rb");
						if (!f) throw std::runtime_error("Can't open the Bloom filter checkpoint " + name);
						FilterFileHeader h;
						std::vector<uint64_t> fileTable;
						try { ReadFilterHeader(f, name, h, fileTable); } catch (...) { std::fclose(f); throw; }
						std::fclose(f);
						// the ranges come from the checkpoint, so what can be checked is that they chain (VE.h:234-254: round 0 starts at 0, every
						// round starts right after the one before, the last one reaches the end of the hash range) ...
						if (h.round != round || h.low != low || h.high < h.low || (round + 1 == rounds && h.high < realSize))
						{
							throw std::runtime_error("The Bloom filter checkpoint " + name + " does not continue the rounds before it (round " + std::to_string(h.round) +
								", range " + std::to_string(h.low) + ":" + std::to_string(h.high) + ")");
						}

						// ... and that the filter was filled from THIS input
						if (h.textLength != text.length || h.textRecords != text.recLength.size() || h.textChecksum != textChecksum)
						{
							throw std::runtime_error("The Bloom filter checkpoint " + name + " was made from other input files (text length, record count or checksum differ)");
						}

						low = h.low;
						high = h.high;
					}
					else if (rounds > 1)
					{
						// reference vertexenumerator.h:234-250
						uint64_t accumulated = binCounter[std::min<uint64_t>(lowBoundary, BINS_COUNT)];
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
					PhaseTimer sub;
					uint64_t kmers = 0;
					uint64_t marks = 0;
					std::vector<uint64_t> roundCounters;  // sharded second pass: {true, false, table, marks} per rank
					if (net)
					{
						// every rank runs the same round; rank 0 is this thread's context
						std::vector<std::string> errors(gpus);
						std::vector<std::thread> pool;
						roundCounters.assign(size_t(gpus) * 4, 0);
						for (int r = 0; r < gpus; r++)
						{
							pool.emplace_back([&, r]()
							{
								try
								{
									// checkpoint: this rank's shard of the round's filter instead of the sharded insert (the query half of the
									// pass is the same), or written out after it (the query does not touch the filter)
									peers_[r].filterLoaded = !options.loadFilter.empty();
									if (peers_[r].filterLoaded)
									{
										LoadFilter(FilterFileName(options.loadFilter, round, combined ? 0 : size_t(r), ckptShards), vertexLength, filterSize, hashFunctions, round, rounds, low, high, table,
											peers_[r].ctx, combined ? 0u : uint32_t(r), uint32_t(ckptShards));
									}

									ShardedFirstPass(peers_[r], *net, int(hashFunctions), low, high);
									if (!options.saveFilter.empty() && (!combined || r == 0))
									{
										SaveFilter(FilterFileName(options.saveFilter, round, combined ? 0 : size_t(r), ckptShards), vertexLength, filterSize, hashFunctions, round, rounds, low, high, table,
											peers_[r].ctx, combined ? 0u : uint32_t(r), uint32_t(ckptShards));
									}

									if (shardedPass2) ShardedSecondPass(peers_[r], *net, abundance, &roundCounters[size_t(r) * 4]);
								}
								catch (std::exception & e)
								{
									errors[r] = e.what();
									net->Barrier().Fail(e.what());
								}
							});
						}

						for (std::thread & th : pool) th.join();
						for (const std::string & e : errors) if (!e.empty()) throw std::runtime_error(e);
						sub.Lap("  round: sharded insert + query");
						logStream << time(0) - mark << "\t";
						mark = time(0);
						logStream << time(0) - mark << "\t" << std::endl;
					}
					else
					{
						if (!options.loadFilter.empty())
						{
							if (!nothing) LoadFilter(FilterFileName(options.loadFilter, round), vertexLength, filterSize, hashFunctions, round, rounds, low, high, table);
						}
						else
						{
							if (!nothing) Check(tpc_filter_reset(ctx_), "filter_reset");
							if (!nothing) Check(tpc_pass1_insert(ctx_, low, high, &kmers), "pass1_insert");
							if (!nothing && !options.saveFilter.empty()) SaveFilter(FilterFileName(options.saveFilter, round), vertexLength, filterSize, hashFunctions, round, rounds, low, high, table);
						}

						sub.Lap("  round: insert");
						logStream << time(0) - mark << "\t";
						mark = time(0);
						if (!nothing) Check(tpc_pass1_query(ctx_, low, high, &marks), "pass1_query");
						sub.Lap("  round: query");
						logStream << time(0) - mark << "\t" << std::endl;
					}

					mark = time(0);
					logStream << "2\t";
					uint64_t truePositives = 0, falsePositives = 0, hashTableSize = 0;
					if (net && shardedPass2)
					{
						// the ranks ran it (ShardedSecondPass above): keys and marks are disjoint over the ranks, the round's figures are sums
						for (int r = 0; r < gpus; r++)
						{
							truePositives += roundCounters[size_t(r) * 4];
							falsePositives += roundCounters[size_t(r) * 4 + 1];
							hashTableSize += roundCounters[size_t(r) * 4 + 2];
							marks += roundCounters[size_t(r) * 4 + 3];
						}
					}
					else
					{
						if (!nothing) Check(tpc_pass2_filter(ctx_, abundance, &truePositives, &falsePositives, &hashTableSize), "pass2_filter");
						if (net) marks = uint64_t(std::max<int64_t>(0, tpc_get_stat(ctx_, "round_marks")));
					}
					sub.Lap("  round: exact filter");
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
				// several GPUs with the second pass sharded: every rank formats and writes its own byte range of the output
				const bool perRankOutput = net && shardedPass2 && std::getenv("TWOPACO_GATHER_OUTPUT") == 0;
				if (net && shardedPass2)
				{
					// key union, key sort, id lookup of every rank's own positions; the (position, id) lists stay on their ranks
					// (TWOPACO_GATHER_OUTPUT=1: gathered on rank 0, which then formats the whole stream)
					std::vector<std::string> errors(gpus);
					std::vector<uint64_t> perRank(gpus, 0);
					std::vector<std::thread> pool;
					for (int r = 0; r < gpus; r++)
					{
						pool.emplace_back([&, r]()
						{
							try
							{
								ShardedFinish(peers_[r], *net, &perRank[r], !perRankOutput);
							}
							catch (std::exception & e)
							{
								errors[r] = e.what();
								net->Barrier().Fail(e.what());
							}
						});
					}

					for (std::thread & th : pool) th.join();
					for (const std::string & e : errors) if (!e.empty()) throw std::runtime_error(e);
					junctions = perRank[0];
				}
				else if (!nothing) Check(tpc_junctions_finalize(ctx_, &junctions), "junctions_finalize");
				vertices_ = junctions;
				logStream << "Reallocating bifurcations time: " << time(0) - mark << std::endl;

				mark = time(0);
				uint64_t marked = 0, valid = 0;
				if (!nothing && !(net && shardedPass2)) Check(tpc_emit(ctx_, &marked, &valid), "emit");
				// EdgeConstructionWorker + FlushEdgeResults + JunctionPositionWriter (reference vertexenumerator.h:837-854,
				// 927-958, junctionapi.h:118-132): the device formats the whole junction stream -- records in
				// (sequence, position) order, stub ids for sequence ends, one separator per sequence-id step
				// (csrc/tpc_stream.hip) -- and `threads` writers move it to the file in chunks through pinned buffers.
				uint64_t streamBytes = 0, occurence = 0;
				if (text.recStart.size() > UINT32_MAX)
				{
					throw std::runtime_error("Too many sequences");
				}

				struct Piece { tpc_ctx * ctx; uint64_t fileOffset, bytes; };
				std::vector<Piece> pieces;
				if (perRankOutput && !nothing)
				{
					// FlushEdgeResults' piece order (vertexenumerator.h:841-849) as an exclusive scan over the ranks: rank r learns
					// the slots in front of its chunk and formats its own bytes (multigpu.h: ShardedStream)
					std::vector<std::string> errors(gpus);
					std::vector<uint64_t> first(gpus, 0), nb(gpus, 0), rec(gpus, 0);
					std::vector<std::thread> pool;
					for (int r = 0; r < gpus; r++)
					{
						pool.emplace_back([&, r]()
						{
							try
							{
								ShardedStream(peers_[r], *net, text.recStart, text.recLength, vertexLength, &first[r], &nb[r], &rec[r]);
							}
							catch (std::exception & e)
							{
								errors[r] = e.what();
								net->Barrier().Fail(e.what());
							}
						});
					}

					for (std::thread & th : pool) th.join();
					for (const std::string & e : errors) if (!e.empty()) throw std::runtime_error(e);
					occurence = rec[0];
					for (int r = 0; r < gpus; r++)
					{
						if (nb[r]) pieces.push_back(Piece{ peers_[r].ctx, first[r], nb[r] });
						streamBytes += nb[r];
					}
				}
				else if (!nothing)
				{
					Check(tpc_emit_stream(ctx_, text.recStart.data(), text.recLength.data(), uint32_t(text.recStart.size()), &streamBytes, &occurence), "emit_stream");
					if (streamBytes) pieces.push_back(Piece{ ctx_, 0, streamBytes });
				}

				timer.Lap("sort + id lookup + junction stream");
				if (std::getenv("TWOPACO_TIMING") && ctx_)
				{
					// what the run holds on the device at its peak (filter, text, partition buffers, second pass, stream): total - free, for a process alone on the device
					std::cerr << "[timing] device memory in use after the rounds: " << double(tpc_get_stat(ctx_, "device_total_bytes") - tpc_get_stat(ctx_, "device_free_bytes")) / 1e9
						<< " GB" << std::endl;
				}
				{
					const int fd = ::open(outFileName.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
					if (fd < 0)
					{
						throw std::runtime_error("Can't create the output file");
					}

					const uint64_t CHUNK = uint64_t(4) << 20;
					struct Task { size_t piece; uint64_t offset, bytes; };
					std::vector<Task> tasks;
					for (size_t p = 0; p < pieces.size(); p++)
					{
						for (uint64_t off = 0; off < pieces[p].bytes; off += CHUNK) tasks.push_back(Task{ p, off, std::min(CHUNK, pieces[p].bytes - off) });
					}

					const size_t workers = size_t(std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(threads, 16), tasks.size())));
					if (streamBytes > 0) (void)::posix_fallocate(fd, 0, off_t(streamBytes));  // one extent up front instead of growing the file chunk by chunk
					// (Round 4 measured stores into a shared mapping of the preallocated file instead of pwrite(): 90-155 ms against 56-65 for the
					//  528 MB of the 62-genome output -- sixteen threads faulting pages of one mapping contend harder than the buffered writes do.)
					std::vector<int> failed(workers, 0);
					std::vector<std::thread> pool;
					for (size_t t = 0; t < workers; t++)
					{
						pool.emplace_back([&, t]()
						{
							void * pinned = 0;
							std::vector<char> pageable;
							char * buf = 0;
							if (tpc_host_alloc(&pinned, CHUNK) == 0)
							{
								buf = static_cast<char*>(pinned);
							}
							else
							{
								pageable.resize(CHUNK);
								buf = pageable.data();
							}

							for (size_t c = t; c < tasks.size() && !failed[t]; c += workers)
							{
								const Piece & piece = pieces[tasks[c].piece];
								const uint64_t n = tasks[c].bytes;
								if (tpc_emit_stream_fetch(piece.ctx, tasks[c].offset, n, buf) != 0)
								{
									failed[t] = 1;
									break;
								}

								const uint64_t off = piece.fileOffset + tasks[c].offset;
								for (uint64_t done = 0; done < n;)
								{
									const ssize_t w = ::pwrite(fd, buf + done, size_t(n - done), off_t(off + done));
									if (w <= 0)
									{
										failed[t] = 2;
										break;
									}

									done += uint64_t(w);
								}
							}

							tpc_host_free(pinned);
						});
					}

					for (std::thread & th : pool) th.join();
					const bool closed = ::close(fd) == 0;
					for (size_t t = 0; t < workers; t++)
					{
						if (failed[t] == 1)
						{
							throw std::runtime_error("Can't fetch the junction stream from the device");
						}

						if (failed[t] == 2 || !closed)
						{
							throw std::runtime_error("Can't write to the output file");
						}
					}
				}

				timer.Lap("write junction stream");
				logStream << "True marks count: " << occurence << std::endl;
				logStream << "Edges construction time: " << time(0) - mark << std::endl;
				logStream << std::string(80, '-') << std::endl;
			}

		private:
			void SaveFilter(const std::string & name, size_t k, size_t bits, size_t q, size_t round, size_t rounds, uint64_t low, uint64_t high, const std::vector<uint64_t> & table,
				tpc_ctx * ctx = 0, uint32_t shard = 0, uint32_t shards = 1)
			{
				if (!ctx) ctx = ctx_;
				FilterFileHeader h;
				std::memset(&h, 0, sizeof(h));
				h.version = FILTER_FILE_VERSION; h.k = uint32_t(k); h.bits = uint32_t(bits); h.q = uint32_t(q); h.round = uint32_t(round); h.rounds = uint32_t(rounds);
				h.low = low; h.high = high; h.words = tpc_filter_words(ctx);
				h.textLength = textFingerprint_[0]; h.textRecords = textFingerprint_[1]; h.textChecksum = textFingerprint_[2];
				h.shard = shard; h.shards = shards;
				std::vector<uint32_t> words(h.words);
				if (tpc_filter_download(ctx, words.data()) != 0) throw std::runtime_error(std::string("filter_download: ") + tpc_last_error(ctx));
				std::FILE * f = std::fopen(name.c_str(), "wb");
				if (!f) throw std::runtime_error("Can't create the Bloom filter checkpoint " + name);
				const std::vector<unsigned char> head = EncodeFilterHeader(h);
				std::vector<unsigned char> tab;
				for (uint64_t t : table) PutLe(tab, t, 8);
				const bool ok = std::fwrite(head.data(), 1, head.size(), f) == head.size() && std::fwrite(tab.data(), 1, tab.size(), f) == tab.size() &&
					std::fwrite(words.data(), sizeof(uint32_t), words.size(), f) == words.size();  // (filter words: little-endian uint32, as every supported host stores them)
				if (std::fclose(f) != 0 || !ok) throw std::runtime_error("Can't write the Bloom filter checkpoint " + name);
			}

			void LoadFilter(const std::string & name, size_t k, size_t bits, size_t q, size_t round, size_t rounds, uint64_t low, uint64_t high, const std::vector<uint64_t> & table,
				tpc_ctx * ctx = 0, uint32_t shard = 0, uint32_t shards = 1)
			{
				if (!ctx) ctx = ctx_;
				std::FILE * f = std::fopen(name.c_str(), "rb");
				if (!f) throw std::runtime_error("Can't open the Bloom filter checkpoint " + name);
				try
				{
					FilterFileHeader h;
					std::vector<uint64_t> fileTable;
					ReadFilterHeader(f, name, h, fileTable);
					if (h.k != k || h.bits != bits || h.q != q || h.round != round || h.rounds != rounds || h.low != low || h.high != high || fileTable != table ||
						h.words != tpc_filter_words(ctx) || h.shard != shard || h.shards != shards)
					{
						throw std::runtime_error("The Bloom filter checkpoint " + name + " does not belong to this round (parameters, hash tables, the round's range or the shard layout differ)");
					}

					if (h.textLength != textFingerprint_[0] || h.textRecords != textFingerprint_[1] || h.textChecksum != textFingerprint_[2])
					{
						throw std::runtime_error("The Bloom filter checkpoint " + name + " was made from other input files (text length, record count or checksum differ)");
					}

					std::vector<uint32_t> words(h.words);
					if (std::fread(words.data(), sizeof(uint32_t), words.size(), f) != words.size()) throw std::runtime_error("Truncated Bloom filter checkpoint: " + name);
					if (tpc_filter_upload(ctx, words.data()) != 0) throw std::runtime_error(std::string("filter_upload: ") + tpc_last_error(ctx));
				}
				catch (...)
				{
					std::fclose(f);
					throw;
				}

				std::fclose(f);
			}

			uint64_t textFingerprint_[3] = {0, 0, 0};  // length, records, checksum of the packed text (Bloom filter checkpoints)
			tpc_ctx * ctx_;
			size_t vertices_;
			VertexRollingHashSeed seed_;
			std::vector<ShardedRank> peers_;  // gpus > 1: one rank per device; peers_[0].ctx == ctx_
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
		try
		{
			ret->Run(fileName, vertexLength, filterSize, hashFunctions, rounds, threads, abundance, outFileName, logStream, options);
		}
		catch (std::runtime_error & e)
		{
			// A filter cut over several GPUs has no scattered-kernel fallback: address skew beyond its overflow lists (a saturated
			// or tiny filter, a poly-A genome) ends the sharded pass.  The single-GPU path handles any input (its partitioned
			// passes fall back to the direct kernels), so the run is repeated there instead of failing.
			if ((options.gpus <= 1 && !options.forceSharded) || std::string(e.what()).find("the sharded path handles") == std::string::npos) throw;
			logStream << "Address skew beyond what the sharded filter handles (" << e.what() << "): repeating the run on one GPU" << std::endl;
			ret.reset(new HipVertexEnumerator());
			EnumeratorOptions single(options);
			single.gpus = 1;
			single.forceSharded = false;
			ret->Run(fileName, vertexLength, filterSize, hashFunctions, rounds, threads, abundance, outFileName, logStream, single);
		}

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
