// host_capi.cpp -- plain-C entry points over the C++ host layer, for language bindings (the
// Python tests and bench.py reach CreateEnumerator, the text packer and the seed tables through
// these with ctypes).  Not part of the device ABI (include/twopaco_hip.h).
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "seed.h"
#include "streamfastaparser.h"
#include "textpack.h"
#include "vertexenumerator.h"

namespace
{
	thread_local std::string g_error;
	char * Dup(const std::string & s)
	{
		char * p = new char[s.size() + 1];
		std::memcpy(p, s.c_str(), s.size() + 1);
		return p;
	}
}

extern "C"
{
	const char * tpch_last_error() { return g_error.c_str(); }
	void tpch_free(void * p) { delete[] static_cast<char*>(p); }

	// q x 5 seed table for `--seed seed` (pinned != 0) or from /dev/urandom
	int tpch_seed_table(uint64_t seed, int pinned, int q, int bits, uint64_t * table)
	{
		try
		{
			std::vector<uint64_t> t = TwoPaCo::MakeSeedTable(size_t(q), size_t(bits), pinned != 0, seed);
			std::memcpy(table, t.data(), t.size() * sizeof(uint64_t));
			return 0;
		}
		catch (std::exception & e) { g_error = e.what(); return -1; }
	}

	// ---- packed text handle -------------------------------------------------------------
	void * tpch_text_new() { TwoPaCo::PackedText * t = new TwoPaCo::PackedText(); t->BeginText(); return t; }
	void tpch_text_free(void * h) { delete static_cast<TwoPaCo::PackedText*>(h); }
	int tpch_text_add_fasta(void * h, const char ** files, int nfiles, int threads)
	{
		try
		{
			TwoPaCo::PackedText * t = static_cast<TwoPaCo::PackedText*>(h);
			std::vector<std::string> names(files, files + nfiles);
			TwoPaCo::PackedText fresh;
			TwoPaCo::PackFastaFiles(names, size_t(threads), fresh);
			*t = fresh;
			return 0;
		}
		catch (std::exception & e) { g_error = e.what(); return -1; }
	}

	// one record given as codes 0..3, 4 = N
	void tpch_text_add_codes(void * h, const uint8_t * codes, uint64_t n)
	{
		TwoPaCo::PackedText * t = static_cast<TwoPaCo::PackedText*>(h);
		t->AppendCodes(codes, n);
		t->EndRecord(n);
	}

	uint64_t tpch_text_length(void * h) { return static_cast<TwoPaCo::PackedText*>(h)->length; }
	uint64_t tpch_text_words(void * h) { return static_cast<TwoPaCo::PackedText*>(h)->bases.size(); }
	const uint64_t * tpch_text_bases(void * h) { return static_cast<TwoPaCo::PackedText*>(h)->bases.data(); }
	const uint32_t * tpch_text_nmask(void * h) { return static_cast<TwoPaCo::PackedText*>(h)->nmask.data(); }
	uint32_t tpch_text_records(void * h) { return uint32_t(static_cast<TwoPaCo::PackedText*>(h)->recStart.size()); }
	const uint64_t * tpch_text_rec_start(void * h) { return static_cast<TwoPaCo::PackedText*>(h)->recStart.data(); }
	const uint64_t * tpch_text_rec_length(void * h) { return static_cast<TwoPaCo::PackedText*>(h)->recLength.data(); }

	// ---- CreateEnumerator ------------------------------------------------------------------
	// Returns an enumerator handle (or NULL; message via tpch_last_error); *log receives the
	// logStream text (free with tpch_free).
	void * tpch_create_enumerator(const char ** files, int nfiles, uint64_t k, uint64_t filterBits, uint64_t q, uint64_t rounds,
		uint64_t threads, uint64_t abundance, const char * tmpDir, const char * outFile, int pinned, uint64_t seed, int device,
		int testFirst, char ** log)
	{
		std::stringstream ss;
		try
		{
			std::vector<std::string> names(files, files + nfiles);
			TwoPaCo::EnumeratorOptions opt;
			opt.pinnedSeed = pinned != 0;
			opt.seed = seed;
			opt.device = device;
			opt.insertTestFirst = testFirst != 0;
			std::unique_ptr<TwoPaCo::VertexEnumerator> e = TwoPaCo::CreateEnumerator(names, k, filterBits, q, rounds, threads, abundance, tmpDir, outFile, ss, opt);
			if (log) *log = Dup(ss.str());
			return e.release();
		}
		catch (std::exception & e)
		{
			g_error = e.what();
			if (log) *log = Dup(ss.str());
			return 0;
		}
	}

	// the same with the multi-GPU knobs: gpus ranks, transport (rccl != 0: RCCL), emulate != 0: all ranks on `device`,
	// forceSharded != 0: the sharded path even for one GPU
	void * tpch_create_enumerator_mgpu(const char ** files, int nfiles, uint64_t k, uint64_t filterBits, uint64_t q, uint64_t rounds,
		uint64_t threads, uint64_t abundance, const char * tmpDir, const char * outFile, int pinned, uint64_t seed, int device,
		int gpus, int rccl, int emulate, int forceSharded, char ** log)
	{
		std::stringstream ss;
		try
		{
			std::vector<std::string> names(files, files + nfiles);
			TwoPaCo::EnumeratorOptions opt;
			opt.pinnedSeed = pinned != 0;
			opt.seed = seed;
			opt.device = device;
			opt.gpus = gpus;
			opt.rccl = rccl != 0;
			opt.emulateRanks = emulate != 0;
			opt.forceSharded = forceSharded != 0;
			std::unique_ptr<TwoPaCo::VertexEnumerator> e = TwoPaCo::CreateEnumerator(names, k, filterBits, q, rounds, threads, abundance, tmpDir, outFile, ss, opt);
			if (log) *log = Dup(ss.str());
			return e.release();
		}
		catch (std::exception & e)
		{
			g_error = e.what();
			if (log) *log = Dup(ss.str());
			return 0;
		}
	}

	void tpch_enumerator_free(void * h) { delete static_cast<TwoPaCo::VertexEnumerator*>(h); }
	uint64_t tpch_vertices_count(void * h) { return static_cast<TwoPaCo::VertexEnumerator*>(h)->GetVerticesCount(); }
	int64_t tpch_get_id(void * h, const char * kmer) { return static_cast<TwoPaCo::VertexEnumerator*>(h)->GetId(kmer); }
	int tpch_hash_seed(void * h, uint64_t * table)
	{
		const TwoPaCo::VertexRollingHashSeed & s = static_cast<TwoPaCo::VertexEnumerator*>(h)->GetHashSeed();
		std::memcpy(table, s.Table().data(), s.Table().size() * sizeof(uint64_t));
		return int(s.HashFunctionsNumber());
	}
}
