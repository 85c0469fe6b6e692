// constructor.cpp -- the `twopaco` command line, flag-compatible with the reference CLI
// (reference src/graphconstructor/constructor.cpp:53-218): -k/--kvalue (odd, default 25),
// -f/--filtersize xor --filtermemory (GB; bits = log2(GB*8e9) truncated, :158), -q/--hashfnumber
// (5), -r/--rounds (1), -t/--threads (1), -a/--abundance (UINT64_MAX), --tmpdir ("."),
// -o/--outfile ("de_bruijn.bin"), --test, and the FASTA file names.  Extra flags that do not
// exist in the reference: --seed S (pin the hash tables, see seed.h), --device N,
// --test-first (test-then-set insert), --gpus N (Bloom filter sharded by bit address over N GPUs of the node,
// RCCL between them; --no-rccl: device-to-device copies; --emulate-ranks: N ranks on one device, for testing),
// --save-filter F / --load-filter F (checkpoint of the Bloom filter after each round's first-pass insert, the reference's
// commented-out ReloadBloomFilter, vertexenumerator.h:29,113-121; a run that loads skips the insert), and with --test,
// --seed makes the trials reproducible (the reference's are not, test.cpp:169).  Errors go to stderr as "\nError: <what>\n", exit code 1
// (reference constructor.cpp:179-188).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <unistd.h>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "selftest.h"
#include "vertexenumerator.h"

namespace
{
	struct ArgError : public std::runtime_error
	{
		std::string arg;
		ArgError(const std::string & msg, const std::string & argId) : std::runtime_error(msg), arg(argId) {}
	};

	bool Match(const std::string & a, const char * shortName, const char * longName)
	{
		return (shortName && a == std::string("-") + shortName) || (longName && a == std::string("--") + longName);
	}

	template<class T> T Parse(const std::string & text, const std::string & argId)
	{
		try
		{
			size_t used = 0;
			if (text.empty() || text[0] == '-') throw std::invalid_argument(text);
			unsigned long long v = std::stoull(text, &used, 10);
			if (used != text.size()) throw std::invalid_argument(text);
			return static_cast<T>(v);
		}
		catch (std::exception &)
		{
			throw ArgError("Couldn't read argument value from string '" + text + "'", argId);
		}
	}

	void Usage()
	{
		std::cout << "USAGE: twopaco {-f <integer>|--filtermemory <float>} [-k <oddc>] [-q <integer>] [-r <integer>]" << std::endl
			<< "               [-t <integer>] [-a <integer>] [--tmpdir <directory name>] [-o <file name>] [--test]" << std::endl
			<< "               [--seed <integer>] [--device <integer>] [--test-first] [--gpus <power of two>] [--no-rccl]" << std::endl
			<< "               [--save-filter <file>] [--load-filter <file>]" << std::endl
			<< "               <fasta files with genomes> ..." << std::endl
			<< "       -q: 1..64 hash functions (the reference takes any number; more than 16 run on slower closed-form kernels)" << std::endl;
	}
}

namespace
{
	// TWOPACO_TIMING=1: milliseconds since the process was started (exec), from /proc/self/stat
	double SinceProcessStart()
	{
		std::FILE * f = std::fopen("/proc/self/stat", "r");
		if (!f) return -1;
		char buf[2048];
		size_t n = std::fread(buf, 1, sizeof(buf) - 1, f);
		std::fclose(f);
		buf[n] = 0;
		const char * p = std::strrchr(buf, ')');
		if (!p) return -1;
		unsigned long long start = 0;
		int field = 2;
		for (p++; *p && field < 22; p++) if (*p == ' ') { field++; if (field == 22) { start = std::strtoull(p + 1, 0, 10); break; } }
		struct timespec ts;
		clock_gettime(CLOCK_BOOTTIME, &ts);
		const double now = ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
		return now - double(start) * 1e3 / double(sysconf(_SC_CLK_TCK));
	}
}

int main(int argc, char * argv[])
{
	const bool timing = std::getenv("TWOPACO_TIMING") != 0;
	if (timing) std::cerr << "[timing] exec -> main: " << SinceProcessStart() << " ms" << std::endl;
	try
	{
		unsigned int kvalue = 25, hashFunctions = 5, rounds = 1, threads = 1;
		size_t abundance = UINT64_MAX;
		bool filterSizeSet = false, filterMemorySet = false, runTests = false;
		unsigned int filterSize = 32;
		double filterMemory = 4;
		std::string tmpDirName = ".", outFileName = "de_bruijn.bin";
		std::vector<std::string> fileName;
		TwoPaCo::EnumeratorOptions options;
		bool optionsSet = false;
		for (int i = 1; i < argc; i++)
		{
			std::string a = argv[i];
			auto value = [&](const std::string & argId) -> std::string
			{
				if (i + 1 >= argc) throw ArgError("Missing a value for this argument!", argId);
				return argv[++i];
			};

			if (Match(a, "k", "kvalue"))
			{
				kvalue = Parse<unsigned int>(value("(--kvalue)"), "(--kvalue)");
				if (kvalue % 2 != 1) throw ArgError("Value '" + std::to_string(kvalue) + "' does not meet constraint: value of K must be odd", "(--kvalue)");
			}
			else if (Match(a, "f", "filtersize")) { filterSize = Parse<unsigned int>(value("(--filtersize)"), "(--filtersize)"); filterSizeSet = true; }
			else if (Match(a, 0, "filtermemory")) { filterMemory = std::atof(value("(--filtermemory)").c_str()); filterMemorySet = true; }
			else if (Match(a, "q", "hashfnumber")) hashFunctions = Parse<unsigned int>(value("(--hashfnumber)"), "(--hashfnumber)");
			else if (Match(a, "r", "rounds")) rounds = Parse<unsigned int>(value("(--rounds)"), "(--rounds)");
			else if (Match(a, "t", "threads")) threads = Parse<unsigned int>(value("(--threads)"), "(--threads)");
			else if (Match(a, "a", "abundance")) abundance = Parse<size_t>(value("(--abundance)"), "(--abundance)");
			else if (Match(a, 0, "tmpdir")) tmpDirName = value("(--tmpdir)");
			else if (Match(a, "o", "outfile")) outFileName = value("(--outfile)");
			else if (Match(a, 0, "test")) runTests = true;
			else if (Match(a, 0, "seed")) { options.pinnedSeed = true; options.seed = Parse<uint64_t>(value("(--seed)"), "(--seed)"); optionsSet = true; }
			else if (Match(a, 0, "device")) { options.device = int(Parse<unsigned int>(value("(--device)"), "(--device)")); optionsSet = true; }
			else if (Match(a, 0, "test-first")) { options.insertTestFirst = true; optionsSet = true; }
			else if (Match(a, 0, "gpus")) { options.gpus = int(Parse<unsigned int>(value("(--gpus)"), "(--gpus)")); optionsSet = true; }
			else if (Match(a, 0, "no-rccl")) { options.rccl = false; optionsSet = true; }
			else if (Match(a, 0, "emulate-ranks")) { options.emulateRanks = true; optionsSet = true; }
			else if (Match(a, 0, "save-filter")) { options.saveFilter = value("(--save-filter)"); optionsSet = true; }
			else if (Match(a, 0, "load-filter")) { options.loadFilter = value("(--load-filter)"); optionsSet = true; }
			else if (Match(a, "h", "help")) { Usage(); return 0; }
			else if (a == "--version") { std::cout << argv[0] << "  version: 1.1.0" << std::endl; return 0; }
			else if (a.size() > 1 && a[0] == '-') throw ArgError("Couldn't find match for argument", "(" + a + ")");
			else fileName.push_back(a);
		}

		if (filterSizeSet == filterMemorySet)
		{
			throw ArgError(filterSizeSet ? "Mutually exclusive argument already set!" : "One of the required arguments is missing!", "(--filtersize|--filtermemory)");
		}

		if (fileName.empty())
		{
			throw ArgError("Required argument missing: filenames", "(filenames)");
		}

		if (runTests)
		{
			size_t trials = 10;
			if (const char * t = std::getenv("TWOPACO_SELFTEST_TRIALS")) trials = size_t(std::atoi(t));
			// reference constructor.cpp:164 runs the trials off std::random_device; --seed makes them (and a failure) reproducible
			if (options.pinnedSeed)
			{
				return TwoPaCo::RunTestsSeeded(options.seed, trials, 20, 9000, 6, TwoPaCo::Range(3, 11), TwoPaCo::Range(1, 2), TwoPaCo::Range(1, 5), TwoPaCo::Range(4, 5), 0.05, 0.1, tmpDirName) ? 0 : 1;
			}

			return TwoPaCo::RunTests(trials, 20, 9000, 6, TwoPaCo::Range(3, 11), TwoPaCo::Range(1, 2), TwoPaCo::Range(1, 5), TwoPaCo::Range(4, 5), 0.05, 0.1, tmpDirName) ? 0 : 1;
		}

		int64_t filterBits = filterSizeSet ? int64_t(filterSize) : int64_t(std::log2(filterMemory * 8e+9));
		std::unique_ptr<TwoPaCo::VertexEnumerator> vid = optionsSet
			? TwoPaCo::CreateEnumerator(fileName, kvalue, size_t(filterBits), hashFunctions, rounds, threads, abundance, tmpDirName, outFileName, std::cout, options)
			: TwoPaCo::CreateEnumerator(fileName, kvalue, size_t(filterBits), hashFunctions, rounds, threads, abundance, tmpDirName, outFileName, std::cout);
		if (vid)
		{
			std::cout << "Distinct junctions = " << vid->GetVerticesCount() << std::endl;
			std::cout << std::endl;
		}

		if (timing) std::cerr << "[timing] exec -> output complete: " << SinceProcessStart() << " ms" << std::endl;
		// Device memory is released here, explicitly (a few ms).  What is left of a normal exit is the HIP runtime's own
		// teardown (code objects, queues: ~0.1 s for nothing the process still needs), so the process ends right after
		// flushing its streams unless TWOPACO_CLEAN_EXIT is set.
		vid.reset();
		if (timing) std::cerr << "[timing] exec -> context destroyed: " << SinceProcessStart() << " ms" << std::endl;
		if (!std::getenv("TWOPACO_CLEAN_EXIT"))
		{
			std::cout.flush();
			std::cerr.flush();
			std::fflush(0);
			std::_Exit(0);
		}
	}
	catch (ArgError & e)
	{
		std::cerr << std::endl << "Error: " << e.what() << " for arg " << e.arg << std::endl;
		return 1;
	}
	catch (std::runtime_error & e)
	{
		std::cerr << std::endl << "Error: " << e.what() << std::endl;
		return 1;
	}

	return 0;
}
