/*
 * urandom_shim.c -- LD_PRELOAD determinism shim for the REFERENCE binary (oracle/_ref).
 *
 * TEST INFRASTRUCTURE ONLY.  The reference seeds each of its hash-function tables from
 * /dev/urandom (common/ngramhashing/mersennetwister.h:242-263), so its junction ids differ
 * run to run.  With TPC_URANDOM_SEED=<u64> in the environment this shim makes the n-th
 * fopen("/dev/urandom") return 624 eight-byte words, word j = the splitmix64-style
 * function below of (seed, n, j) -- the same function as orc_urandom_word() in
 * twopaco_oracle.c and tpc_urandom_word() in twopaco_amd/host/seed.cpp, which is how
 * `twopaco --seed S` reproduces the reference's tables.  Nothing of the reference is
 * replaced: only the bytes it reads as entropy are pinned.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t shim_word(uint64_t seed, uint64_t nopen, uint64_t j)
{
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + nopen * 0xD1B54A32D192ED03ull + (j + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static uint64_t n_open = 0;

FILE *fopen(const char *path, const char *mode)
{
    static FILE *(*real_fopen)(const char *, const char *) = NULL;
    if (!real_fopen) real_fopen = (FILE * (*)(const char *, const char *)) dlsym(RTLD_NEXT, "fopen");
    const char *s = getenv("TPC_URANDOM_SEED");
    if (s && path && strcmp(path, "/dev/urandom") == 0) {
        uint64_t seed = strtoull(s, NULL, 0);
        uint64_t *buf = (uint64_t *)malloc(624 * 8); /* leaked on purpose: tiny, fmemopen keeps using it */
        uint64_t n = __atomic_fetch_add(&n_open, 1, __ATOMIC_SEQ_CST);
        for (int j = 0; j < 624; j++) buf[j] = shim_word(seed, n, (uint64_t)j);
        return fmemopen(buf, 624 * 8, "rb");
    }
    return real_fopen(path, mode);
}
