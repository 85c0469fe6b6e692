#include "selftest.h"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <random>
#include <set>
#include <sstream>
#include <vector>

#include "dnachar.h"
#include "vertexenumerator.h"

namespace TwoPaCo
{
	namespace
	{
		// Symbols: 0..3 bases; every non-ACGT character and both sequence ends are fresh unique
		// symbols >= 4, so two of them never compare equal (reference test.cpp:73-109).
		typedef std::vector<int> Symbols;

		void NaiveJunctions(const std::vector<std::string> & chr, size_t k, std::set<std::string> & junction, std::vector<std::vector<bool> > & marks)
		{
			int fresh = 4;
			std::vector<Symbols> genome;
			for (const std::string & s : chr)
			{
				Symbols fwd;
				fwd.push_back(fresh++);
				for (char ch : s) fwd.push_back(DnaChar::IsDefinite(ch) ? int(DnaChar::MakeUpChar(ch)) : fresh++);
				fwd.push_back(fresh++);
				Symbols rev;
				for (size_t i = fwd.size(); i-- > 0;) rev.push_back(fwd[i] < 4 ? 3 - fwd[i] : fresh++);
				genome.push_back(fwd);
				genome.push_back(rev);
			}

			std::map<std::string, std::set<int> > in, out;
			for (const Symbols & g : genome)
			{
				size_t bad = 0;
				for (size_t i = 0; i < g.size(); i++)
				{
					bad += g[i] >= 4;
					if (i >= k) bad -= g[i - k] >= 4;
					if (i + 1 >= k && bad == 0)
					{
						size_t start = i + 1 - k;
						std::string v;
						for (size_t t = start; t <= i; t++) v.push_back(DnaChar::UnMakeUpChar(size_t(g[t])));
						if (i + 1 < g.size()) out[v].insert(g[i + 1]);
						if (start > 0) in[v].insert(g[start - 1]);
					}
				}
			}

			for (auto * e : { &in, &out })
			{
				for (auto & kv : *e)
				{
					if (kv.second.size() > 1)
					{
						junction.insert(kv.first);
						junction.insert(DnaChar::ReverseCompliment(kv.first));
					}
				}
			}

			for (size_t i = 0; i < chr.size(); i++)
			{
				marks[i].assign(chr[i].size(), false);
				for (size_t pos = 0; pos < chr[i].size(); pos++)
				{
					if (pos == 0 || pos + k == chr[i].size() || (pos + k <= chr[i].size() && junction.count(chr[i].substr(pos, k)) > 0))
					{
						marks[i][pos] = true;
					}
				}
			}
		}
	}

	bool RunTests(size_t tests, size_t filterBits, size_t length, size_t chrNumber, Range vertexSize, Range hashFunctions,
		Range rounds, Range threads, double changeRate, double indelRate, const std::string & temporaryDir)
	{
		std::random_device rd;  // reference test.cpp:169: an unseeded run; the seed drawn here is printed if a trial fails
		const uint64_t seed = (uint64_t(rd()) << 32) ^ uint64_t(rd());
		return RunTestsSeeded(seed, tests, filterBits, length, chrNumber, vertexSize, hashFunctions, rounds, threads, changeRate, indelRate, temporaryDir);
	}

	bool RunTestsSeeded(uint64_t seed, size_t tests, size_t filterBits, size_t length, size_t chrNumber, Range vertexSize, Range hashFunctions,
		Range rounds, Range threads, double changeRate, double indelRate, const std::string & temporaryDir)
	{
		const std::string temporaryFasta = temporaryDir + "/test.fa";
		const std::string temporaryEdge = temporaryDir + "/out.bin";
		std::vector<std::string> fileName(1, temporaryFasta);
		std::uniform_real_distribution<> unit(0, 1);
		const std::string alphabet("ACGT");
		for (size_t t = 0; t < tests; t++)
		{
			const uint64_t trialSeed = seed + t;
			std::mt19937_64 rng(trialSeed);
			EnumeratorOptions options;
			options.pinnedSeed = true;  // the hash tables of this trial's runs (seed.h): part of what a replay must reproduce
			options.seed = trialSeed;
			if (const char * d = std::getenv("TWOPACO_DEVICE")) options.device = std::atoi(d);
			// chr0 random with N at rate 1/500, the others = chr0 with substitutions/indels (reference test.cpp:20-67)
			std::vector<std::string> chr(chrNumber);
			for (size_t i = 0; i < length; i++) chr[0].push_back(rng() % 500 == 0 ? 'N' : alphabet[rng() % 4]);
			for (size_t c = 1; c < chrNumber; c++)
			{
				for (char ch : chr[0])
				{
					if (unit(rng) <= changeRate)
					{
						if (unit(rng) <= indelRate) chr[c].push_back(alphabet[rng() % 4]);
						else if (unit(rng) <= 0.5) { chr[c].push_back(ch); chr[c].push_back(alphabet[rng() % 4]); }
					}
					else chr[c].push_back(ch);
				}
			}

			{
				std::ofstream test(temporaryFasta.c_str());
				if (!test) throw std::runtime_error("Can't create a temporary file for testing");
				for (size_t j = 0; j < chrNumber; ++j) test << ">" << j << std::endl << chr[j] << std::endl;
			}

			for (size_t k = vertexSize.first; k < vertexSize.second; k += 2)
			{
				std::set<std::string> junctions;
				std::vector<std::vector<bool> > naiveMarks(chrNumber), fastMarks(chrNumber);
				NaiveJunctions(chr, k, junctions, naiveMarks);
				for (size_t hf = hashFunctions.first; hf < hashFunctions.second; ++hf)
				for (size_t r = rounds.first; r < rounds.second; ++r)
				for (size_t thr = threads.first; thr < threads.second; ++thr)
				{
					std::stringstream null;
					std::unique_ptr<VertexEnumerator> vid = CreateEnumerator(fileName, k, filterBits, hf, r, thr, UINT32_MAX, temporaryDir, temporaryEdge, null, options);
					for (size_t i = 0; i < chrNumber; i++) fastMarks[i].assign(chr[i].size(), false);
					JunctionPositionReader reader(temporaryEdge);
					reader.RestoreAllVectors(fastMarks);
					bool ok = naiveMarks == fastMarks;
					if (!ok)
					{
						for (size_t i = 0; i < chrNumber; i++)
							for (size_t pos = 0; pos < chr[i].size(); pos++)
								if (fastMarks[i][pos] != naiveMarks[i][pos])
									std::cerr << "ERROR at chr " << i << " pos " << pos << ", " << fastMarks[i][pos] << " != " << naiveMarks[i][pos] << std::endl;
					}

					for (const std::string & vertex : junctions) ok = ok && vid->GetId(vertex) != INVALID_VERTEX;
					if (!ok)
					{
						std::cerr << "Test # " << t << " FAILED (k = " << k << ", q = " << hf << ", rounds = " << r << "; replay: --test --seed " << trialSeed << ")" << std::endl;
						return false;
					}
				}
			}

			std::remove(temporaryFasta.c_str());
			std::remove(temporaryEdge.c_str());
			std::cerr << "Test # " << t << " PASSED" << std::endl;
		}

		return true;
	}
}
