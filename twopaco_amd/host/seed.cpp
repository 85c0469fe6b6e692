#include "seed.h"

#include <cstdio>
#include <ctime>
#include <stdexcept>

namespace TwoPaCo
{
	namespace
	{
		// MT19937 (Matsumoto & Nishimura 1998, init_by_array 2002) -- the generator behind the
		// reference's MTRand (mersennetwister.h); written from the published recurrences.
		class Mt19937
		{
		public:
			explicit Mt19937(const uint32_t * key, size_t len)
			{
				Init(19650218u);
				size_t i = 1, j = 0;
				for (size_t n = (N > len ? N : len); n; --n)
				{
					s_[i] = (s_[i] ^ ((s_[i - 1] ^ (s_[i - 1] >> 30)) * 1664525u)) + key[j] + static_cast<uint32_t>(j);
					if (++i >= N) { s_[0] = s_[N - 1]; i = 1; }
					if (++j >= len) j = 0;
				}

				for (size_t n = N - 1; n; --n)
				{
					s_[i] = (s_[i] ^ ((s_[i - 1] ^ (s_[i - 1] >> 30)) * 1566083941u)) - static_cast<uint32_t>(i);
					if (++i >= N) { s_[0] = s_[N - 1]; i = 1; }
				}

				s_[0] = 0x80000000u;
				idx_ = N;
			}

			uint32_t Next()
			{
				if (idx_ >= N) Twist();
				uint32_t y = s_[idx_++];
				y ^= y >> 11;
				y ^= (y << 7) & 0x9d2c5680u;
				y ^= (y << 15) & 0xefc60000u;
				return y ^ (y >> 18);
			}

		private:
			static const size_t N = 624, M = 397;
			void Init(uint32_t seed)
			{
				s_[0] = seed;
				for (size_t i = 1; i < N; i++) s_[i] = 1812433253u * (s_[i - 1] ^ (s_[i - 1] >> 30)) + static_cast<uint32_t>(i);
			}

			void Twist()
			{
				for (size_t i = 0; i < N; i++)
				{
					uint32_t y = (s_[i] & 0x80000000u) | (s_[(i + 1) % N] & 0x7fffffffu);
					s_[i] = s_[(i + M) % N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
				}
				idx_ = 0;
			}

			uint32_t s_[N];
			size_t idx_;
		};
	}

	uint64_t tpc_urandom_word(uint64_t seed, uint64_t nopen, uint64_t j)
	{
		uint64_t z = seed * 0x9E3779B97F4A7C15ull + nopen * 0xD1B54A32D192ED03ull + (j + 1) * 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		return z ^ (z >> 31);
	}

	std::vector<uint64_t> MakeSeedTable(size_t hashFunctions, size_t bits, bool pinned, uint64_t seed)
	{
		static const int CHARS[5] = { 'A', 'C', 'G', 'T', 'N' };
		if (bits < 2 || bits > 62)
		{
			throw std::runtime_error("Unsupported number of filter bits");
		}

		// maskfnc(wordsize) split into the two 32-bit generator ranges (characterhash.h:30-36,46-47)
		const uint32_t lomask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
		const uint32_t himask = bits > 32 ? static_cast<uint32_t>((1ull << (bits - 32)) - 1ull) : 0u;
		std::vector<uint64_t> table(hashFunctions * 5);
		FILE * urandom = pinned ? 0 : std::fopen("/dev/urandom", "rb");
		uint64_t fallback = static_cast<uint64_t>(std::time(0)) * 0x9E3779B97F4A7C15ull;
		for (size_t i = 0; i < hashFunctions; i++)
		{
			uint32_t key[2][624];
			for (size_t gen = 0; gen < 2; gen++)  // high-word generator first, then low-word (characterhash.h:46-47)
			{
				for (size_t j = 0; j < 624; j++)
				{
					uint64_t word;
					if (pinned) word = tpc_urandom_word(seed, 2 * i + gen, j);
					else if (!urandom || std::fread(&word, sizeof(word), 1, urandom) != 1) word = tpc_urandom_word(fallback, 2 * i + gen, j);
					key[gen][j] = static_cast<uint32_t>(word);  // low 32 bits of each unsigned long (mersennetwister.h:222)
				}
			}

			Mt19937 hi(key[0], 624), lo(key[1], 624);
			uint64_t all[256];
			for (int c = 0; c < 256; c++)
			{
				uint64_t l = lo.Next() & lomask;
				uint64_t h = hi.Next() & himask;
				all[c] = l | (h << 32);
			}

			for (int c = 0; c < 5; c++) table[i * 5 + c] = all[CHARS[c]];
		}

		if (urandom) std::fclose(urandom);
		return table;
	}
}
