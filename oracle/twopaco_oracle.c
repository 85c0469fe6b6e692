/*
 * twopaco_oracle.c -- CPU restatement of TwoPaCo's two-pass junction enumeration.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path in
 * twopaco_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (twopaco_amd/) never links or calls it.
 *
 * Parity status: PINNED.  tests/golden/ holds outputs of the real reference binary
 * (built from /root/reference by oracle/Makefile `ref`, /dev/urandom pinned by
 * oracle/urandom_shim.c) and tests/test_oracle_golden.py checks this restatement
 * against them byte for byte (de_bruijn.bin, log counters, round ranges).
 *
 * Every function cites the reference lines it restates.  Paths are relative to
 * /root/reference/src ; VE.h = graphconstructor/vertexenumerator.h .
 *
 * Formulation.  The reference streams each FASTA record as 'N' + bases + 'N', cut
 * into overlapping Tasks (VE.h:1108-1226).  Because consecutive Tasks overlap by
 * k+1 characters, every vertex (k-mer window) with both neighbours visible is
 * processed exactly once by the query/filter/output workers and every (k+1)-mer at
 * least once by the fill worker; the restatement therefore works on one global
 * text  T = N rec0 N rec1 N ... recS-1 N  (codes A0 C1 G2 T3 N4) indexed by a
 * global position g; seq coordinate = g - rec_start[r].
 */
#define _GNU_SOURCE /* MAP_ANONYMOUS / MAP_NORESERVE under -std=c11 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <sys/mman.h>

#define ORC_N 4
#define ORC_MAXQ 64 /* the reference takes any -q (constructor.cpp:83-90); 64 is what the product accepts */
#define ORC_INVALID_VERTEX INT64_MAX /* graphconstructor/common.cpp:5 */

typedef struct {
    int k, L, q;
    uint64_t h[ORC_MAXQ][5];  /* h[i][c]: character table of hash fn i, c in A,C,G,T,N */
    uint64_t hk[ORC_MAXQ][5]; /* rotl_L(h[i][c], k mod L) */
    /* text */
    uint8_t *txt;
    uint64_t ntxt, captxt;
    uint64_t *rec_start, *rec_len;
    uint32_t nrec, caprec;
    /* results */
    uint32_t *filter;   /* last round's Bloom filter, 2^L/32+1 words */
    uint32_t *mask;     /* OR of all rounds' candidate masks, bit g */
    uint32_t *rmask;    /* current round's candidate mask */
    uint64_t *keys;     /* J x C sorted junction keys */
    uint64_t nkeys;
    int C;
    /* per-round counters (VE.h:384-387) */
    uint64_t r_true[64], r_false[64], r_table[64], r_marks[64], r_low[64], r_high[64];
    int rounds;
    uint64_t true_marks; /* VE.h:463 */
    /* output records */
    uint32_t *out_seq, *out_pos;
    int64_t *out_id;
    uint64_t nout, capout;
    char err[256];
} orc_run;

/* ---------------------------------------------------------------- a1: seeds */

/* MT19937 as published (Matsumoto & Nishimura); the reference's copy is
 * common/ngramhashing/mersennetwister.h:175-324 (randInt, seed(array), reload). */
typedef struct { uint32_t s[624]; int left; int idx; } orc_mt;

static void mt_initialize(orc_mt *m, uint32_t seed)
{   /* mersennetwister.h:289-303 */
    m->s[0] = seed;
    for (int i = 1; i < 624; i++)
        m->s[i] = 1812433253u * (m->s[i - 1] ^ (m->s[i - 1] >> 30)) + (uint32_t)i;
}

static uint32_t mt_twist(uint32_t mm, uint32_t s0, uint32_t s1)
{   /* mersennetwister.h:326-327 */
    return mm ^ (((s0 & 0x80000000u) | (s1 & 0x7fffffffu)) >> 1) ^ ((s1 & 1u) ? 0x9908b0dfu : 0u);
}

static void mt_reload(orc_mt *m)
{   /* mersennetwister.h:305-318 */
    uint32_t *p = m->s;
    int i;
    for (i = 624 - 397; i--; ++p) *p = mt_twist(p[397], p[0], p[1]);
    for (i = 397; --i; ++p) *p = mt_twist(p[397 - 624], p[0], p[1]);
    *p = mt_twist(p[397 - 624], p[0], m->s[0]);
    m->left = 624;
    m->idx = 0;
}

static void mt_seed_array(orc_mt *m, const uint32_t *big, int n)
{   /* mersennetwister.h:206-237 (init_by_array) */
    mt_initialize(m, 19650218u);
    int i = 1, j = 0;
    int k = 624 > n ? 624 : n;
    for (; k; --k) {
        m->s[i] = (m->s[i] ^ ((m->s[i - 1] ^ (m->s[i - 1] >> 30)) * 1664525u)) + big[j] + (uint32_t)j;
        ++i; ++j;
        if (i >= 624) { m->s[0] = m->s[623]; i = 1; }
        if (j >= n) j = 0;
    }
    for (k = 623; k; --k) {
        m->s[i] = (m->s[i] ^ ((m->s[i - 1] ^ (m->s[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        ++i;
        if (i >= 624) { m->s[0] = m->s[623]; i = 1; }
    }
    m->s[0] = 0x80000000u;
    mt_reload(m);
}

static uint32_t mt_next(orc_mt *m)
{   /* mersennetwister.h:161-174 */
    if (m->left == 0) mt_reload(m);
    --m->left;
    uint32_t s1 = m->s[m->idx++];
    s1 ^= (s1 >> 11);
    s1 ^= (s1 << 7) & 0x9d2c5680u;
    s1 ^= (s1 << 15) & 0xefc60000u;
    return s1 ^ (s1 >> 18);
}

/* The deterministic stand-in for /dev/urandom shared with oracle/urandom_shim.c and
 * twopaco_amd/host (tpc_urandom_word): the n-th fopen("/dev/urandom") yields 624
 * 8-byte words; word j = splitmix64 output j of a stream keyed by (seed, n). */
static uint64_t orc_urandom_word(uint64_t seed, uint64_t nopen, uint64_t j)
{
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + nopen * 0xD1B54A32D192ED03ull + (j + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* Character tables of the q CyclicHash functions, restricted to A,C,G,T,N.
 * characterhash.h:41-54: for 64-bit hash values two generators are built per
 * function, first the "high word" one (range maxval>>32) then the "low word" one;
 * each MTRand() reads 624 unsigned longs from /dev/urandom and keeps the low 32 bits
 * (mersennetwister.h:242-258); entry c = c-th draw of each (randInt(n) never rejects
 * because n is 2^b-1, mersennetwister.h:177-193).  maskfnc: characterhash.h:30-36. */
void orc_seed_table(uint64_t seed, int q, int L, uint64_t *table /* q*5 */)
{
    static const int chars[5] = { 'A', 'C', 'G', 'T', 'N' };
    uint32_t big[624];
    uint32_t lomask = L >= 32 ? 0xFFFFFFFFu : ((1u << L) - 1u);
    uint32_t himask = L > 32 ? (uint32_t)((1ull << (L - 32)) - 1ull) : 0u;
    for (int i = 0; i < q; i++) {
        orc_mt hi, lo;
        for (int j = 0; j < 624; j++) big[j] = (uint32_t)orc_urandom_word(seed, 2 * i, j);
        mt_seed_array(&hi, big, 624);
        for (int j = 0; j < 624; j++) big[j] = (uint32_t)orc_urandom_word(seed, 2 * i + 1, j);
        mt_seed_array(&lo, big, 624);
        uint64_t all[256];
        for (int c = 0; c < 256; c++) {
            uint64_t l = mt_next(&lo) & lomask;
            uint64_t h = mt_next(&hi) & himask;
            all[c] = l | (h << 32);
        }
        for (int c = 0; c < 5; c++) table[i * 5 + c] = all[chars[c]];
    }
}

/* ------------------------------------------------------------ a2: cyclic hash */

static inline uint64_t rotl1(uint64_t x, int L)
{   /* cyclichash.h:46-48 fastleftshift1 */
    uint64_t mask1 = (1ull << (L - 1)) - 1ull;
    return ((x & mask1) << 1) | (x >> (L - 1));
}
static inline uint64_t rotr1(uint64_t x, int L)
{   /* cyclichash.h:50-52 fastrightshift1 */
    return (x >> 1) | ((x & 1ull) << (L - 1));
}
static inline uint64_t rotln(uint64_t x, int L, int r)
{   /* cyclichash.h:42-44 fastleftshiftn, r = n % wordsize */
    if (r == 0) return x;
    uint64_t maskn = (1ull << (L - r)) - 1ull;
    return ((x & maskn) << r) | (x >> (L - r));
}
static inline int rc(int c) { return c == ORC_N ? ORC_N : 3 - c; } /* dnachar.cpp:52-58 */

typedef struct { uint64_t pos[ORC_MAXQ], neg[ORC_MAXQ]; } orc_vhash;

/* VertexRollingHash ctor, vertexrollinghash.h:79-102: pos = eat left to right,
 * neg = eat reverse complement (right to left). eat: cyclichash.h:106-109 */
static void vh_init(const orc_run *R, orc_vhash *v, const uint8_t *w, int nf)
{
    for (int i = 0; i < nf; i++) {
        uint64_t p = 0, n = 0;
        for (int t = 0; t < R->k; t++) p = rotl1(p, R->L) ^ R->h[i][w[t]];
        for (int t = R->k - 1; t >= 0; t--) n = rotl1(n, R->L) ^ R->h[i][rc(w[t])];
        v->pos[i] = p;
        v->neg[i] = n;
    }
}

/* VertexRollingHash::Update, vertexrollinghash.h:104-113; update cyclichash.h:86-93,
 * reverse_update cyclichash.h:97-102 */
static void vh_update(const orc_run *R, orc_vhash *v, int prev, int next, int nf)
{
    for (int i = 0; i < nf; i++) {
        v->pos[i] = rotl1(v->pos[i], R->L) ^ R->hk[i][prev] ^ R->h[i][next];
        uint64_t x = v->neg[i] ^ R->hk[i][rc(next)] ^ R->h[i][rc(prev)];
        v->neg[i] = rotr1(x, R->L);
    }
}

static inline uint64_t vh_vertex(const orc_vhash *v)
{   /* GetVertexHash vertexrollinghash.h:137-142 */
    return v->pos[0] < v->neg[0] ? v->pos[0] : v->neg[0];
}

/* Outgoing edge v+c: DetermineStrandExtend (vertexrollinghash.h:170-184) then
 * GetOutgoingEdgeHash (:157-168); hash_extend cyclichash.h:112-114, hash_prepend :117-121 */
static void edge_out(const orc_run *R, const orc_vhash *v, int c, uint64_t *addr)
{
    uint64_t p[ORC_MAXQ], n[ORC_MAXQ];
    int neg = 0;
    for (int i = 0; i < R->q; i++) {
        p[i] = rotl1(v->pos[i], R->L) ^ R->h[i][c];
        n[i] = R->hk[i][rc(c)] ^ v->neg[i];
    }
    for (int i = 0; i < R->q; i++)
        if (p[i] != n[i]) { neg = n[i] < p[i]; break; }
    for (int i = 0; i < R->q; i++) addr[i] = neg ? n[i] : p[i];
}

/* Ingoing edge c+v: DetermineStrandPrepend (:186-200), GetIngoingEdgeHash (:144-155) */
static void edge_in(const orc_run *R, const orc_vhash *v, int c, uint64_t *addr)
{
    uint64_t p[ORC_MAXQ], n[ORC_MAXQ];
    int neg = 0;
    for (int i = 0; i < R->q; i++) {
        p[i] = R->hk[i][c] ^ v->pos[i];
        n[i] = rotl1(v->neg[i], R->L) ^ R->h[i][rc(c)];
    }
    for (int i = 0; i < R->q; i++)
        if (p[i] != n[i]) { neg = n[i] < p[i]; break; }
    for (int i = 0; i < R->q; i++) addr[i] = neg ? n[i] : p[i];
}

/* ------------------------------------------------------------ a4: bit vector */
static inline int bit_get(const uint32_t *f, uint64_t i) { return (f[i >> 5] >> (i & 31)) & 1u; } /* concurrentbitvector.cpp:39-52 */
static inline void bit_set(uint32_t *f, uint64_t i) { f[i >> 5] |= 1u << (i & 31); }           /* :31-37 */

/* ------------------------------------------------------------------ lifecycle */
static uint64_t filter_words(const orc_run *R);
/* A zeroed filter (ConcurrentBitVector ctor, concurrentbitvector.cpp:11-24).  calloc, not malloc + memset: the
 * pages of a 2^38..2^40-bit filter that a small test text never touches then cost nothing. */
static void filter_free(orc_run *R)
{
    if (R->filter) munmap(R->filter, filter_words(R) * sizeof(uint32_t));
    R->filter = NULL;
}
static int filter_zero(orc_run *R)
{   /* anonymous zero pages, not reserved: a 128 GiB filter that a test text touches in a few thousand places is fine */
    filter_free(R);
    void *p = mmap(NULL, filter_words(R) * sizeof(uint32_t), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { snprintf(R->err, sizeof R->err, "cannot map a %d-bit filter", R->L); return -1; }
    R->filter = (uint32_t *)p;
    return 0;
}

orc_run *orc_create(int k, int L, int q, const uint64_t *table)
{
    if (q < 1 || q > ORC_MAXQ || L < 2 || L > 62 || k < 1) return NULL;
    orc_run *R = (orc_run *)calloc(1, sizeof(orc_run));
    R->k = k; R->L = L; R->q = q;
    for (int i = 0; i < q; i++)
        for (int c = 0; c < 5; c++) {
            R->h[i][c] = table[i * 5 + c];
            R->hk[i][c] = rotln(table[i * 5 + c], L, k % L);
        }
    /* CalculateNeededCapacity candidateoccurence.h:129-133 */
    R->C = (k + 4 + 31) / 32;
    R->captxt = 1 << 16;
    R->txt = (uint8_t *)malloc(R->captxt);
    R->txt[0] = ORC_N;
    R->ntxt = 1;
    return R;
}

void orc_destroy(orc_run *R)
{
    if (!R) return;
    free(R->txt); free(R->rec_start); free(R->rec_len); filter_free(R); free(R->mask); free(R->rmask);
    free(R->keys); free(R->out_seq); free(R->out_pos); free(R->out_id);
    free(R);
}

const char *orc_error(const orc_run *R) { return R->err; }

static void txt_push(orc_run *R, uint8_t c)
{
    if (R->ntxt == R->captxt) { R->captxt *= 2; R->txt = (uint8_t *)realloc(R->txt, R->captxt); }
    R->txt[R->ntxt++] = c;
}

static void rec_begin(orc_run *R)
{
    if (R->nrec == R->caprec) {
        R->caprec = R->caprec ? R->caprec * 2 : 64;
        R->rec_start = (uint64_t *)realloc(R->rec_start, R->caprec * sizeof(uint64_t));
        R->rec_len = (uint64_t *)realloc(R->rec_len, R->caprec * sizeof(uint64_t));
    }
    R->rec_start[R->nrec] = R->ntxt;
    R->rec_len[R->nrec] = 0;
}
static void rec_end(orc_run *R)
{
    R->rec_len[R->nrec] = R->ntxt - R->rec_start[R->nrec];
    R->nrec++;
    txt_push(R, ORC_N);
}

static int code_of(int ch)
{   /* dnachar.cpp:18-33 MakeUpChar; VE.h:1174 maps every non-definite char to 'N' */
    switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; }
    return ORC_N;
}

static int is_valid(int ch)
{   /* dnachar.cpp:11 VALID_CHARS */
    return ch > 0 && strchr("ACGTURYKMSWBDHWNXV", ch) != NULL;
}

/* One record given as raw sequence characters (whitespace skipped, upper-cased,
 * validated) -- the per-character rules of StreamFastaParser::GetChar,
 * common/streamfastaparser.cpp:61-93. Returns 0, or -1 on an invalid character. */
int orc_add_record(orc_run *R, const char *s, uint64_t n)
{
    rec_begin(R);
    for (uint64_t i = 0; i < n; i++) {
        int ch = (unsigned char)s[i];
        if (isspace(ch)) continue;
        int up = toupper(ch);
        if (!is_valid(up)) {
            snprintf(R->err, sizeof R->err, "Found an invalid character '%c' in sequence", ch);
            return -1;
        }
        txt_push(R, (uint8_t)code_of(up));
    }
    rec_end(R);
    return 0;
}

/* A FASTA file: ReadRecord (streamfastaparser.cpp:29-59: '>' required, header runs to
 * the first '\n'), then GetChar until the next '>' (:61-93).  Every record consumes a
 * sequence id, also the ones too short to be dispatched (VE.h:1135,1177). */
int orc_add_fasta(orc_run *R, const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) { snprintf(R->err, sizeof R->err, "Can't open file %s", path); return -1; }
    int ch = fgetc(f);
    while (ch != EOF) {
        if (ch != '>') {
            snprintf(R->err, sizeof R->err, "The FASTA header should start with a '>', started with '%c'", ch);
            fclose(f);
            return -1;
        }
        while ((ch = fgetc(f)) != EOF && ch != '\n') {}
        rec_begin(R);
        while ((ch = fgetc(f)) != EOF && ch != '>') {
            if (isspace(ch)) continue;
            int up = toupper(ch);
            if (!is_valid(up)) {
                snprintf(R->err, sizeof R->err, "Found an invalid character '%c' in sequence", ch);
                fclose(f);
                return -1;
            }
            txt_push(R, (uint8_t)code_of(up));
        }
        rec_end(R);
    }
    fclose(f);
    return 0;
}

/* ---------------------------------------------------------------- key packing */

/* CompressedString layout compressedstring.h:188-195,252-264: base i at bits 2(i%32)
 * of word i/32.  CandidateOccurence::Set candidateoccurence.h:25-50: forward strand iff
 * posHash0 < negHash0, tie -> LessSelfReverseComplement (dnachar.cpp:98-114). */
static int less_self_rc(const uint8_t *w, int k)
{
    for (int i = 0; i < k; i++) {
        int r = rc(w[k - 1 - i]);
        if (w[i] != r) return w[i] < r;
    }
    return 0;
}
static void pack_fwd(const uint8_t *w, int k, int C, uint64_t *key)
{
    for (int i = 0; i < C; i++) key[i] = 0;
    for (int i = 0; i < k; i++) key[i >> 5] |= (uint64_t)w[i] << (2 * (i & 31));
}
static void pack_rc(const uint8_t *w, int k, int C, uint64_t *key)
{
    for (int i = 0; i < C; i++) key[i] = 0;
    for (int i = 0; i < k; i++) key[i >> 5] |= (uint64_t)(3 - w[k - 1 - i]) << (2 * (i & 31));
}

static int key_cmp_C;
static int key_cmp(const void *a, const void *b)
{   /* CompressedString::Less compressedstring.h:93-104 */
    const uint64_t *x = (const uint64_t *)a, *y = (const uint64_t *)b;
    for (int i = 0; i < key_cmp_C; i++)
        if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
    return 0;
}

static int64_t key_find(const orc_run *R, const uint64_t *key)
{
    uint64_t lo = 0, hi = R->nkeys;
    key_cmp_C = R->C;
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        if (key_cmp(R->keys + mid * R->C, key) < 0) lo = mid + 1; else hi = mid;
    }
    if (lo < R->nkeys && key_cmp(R->keys + lo * R->C, key) == 0) return (int64_t)lo;
    return -1;
}

/* BifurcationStorage::GetId bifurcationstorage.h:100-127 */
static int64_t get_id(const orc_run *R, const uint8_t *w)
{
    uint64_t key[32];
    pack_fwd(w, R->k, R->C, key);
    int64_t i = key_find(R, key);
    if (i >= 0) return i + 1;
    pack_rc(w, R->k, R->C, key);
    i = key_find(R, key);
    if (i >= 0) return -(i + 1);
    return ORC_INVALID_VERTEX;
}

int64_t orc_get_id(const orc_run *R, const char *kmer)
{
    uint8_t w[1024];
    for (int i = 0; i < R->k; i++) { w[i] = (uint8_t)code_of(kmer[i]); if (w[i] == ORC_N) return ORC_INVALID_VERTEX; }
    return get_id(R, w);
}

/* -------------------------------------------------- exact filter hash table (a9) */
typedef struct { uint64_t *keys; uint8_t *prevm, *nextm; uint64_t *count; uint64_t cap, n; int C; } orc_table;

static uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static void table_init(orc_table *t, uint64_t expect, int C)
{
    t->cap = 64;
    while (t->cap < expect * 2 + 2) t->cap *= 2;
    t->C = C; t->n = 0;
    t->keys = (uint64_t *)malloc(t->cap * C * sizeof(uint64_t));
    t->prevm = (uint8_t *)calloc(t->cap, 1);
    t->nextm = (uint8_t *)calloc(t->cap, 1);
    t->count = (uint64_t *)calloc(t->cap, sizeof(uint64_t));
}
static void table_free(orc_table *t) { free(t->keys); free(t->prevm); free(t->nextm); free(t->count); }

/* The reference keeps the first occurrence's (prev,next) and flags a bifurcation when a
 * later occurrence differs from it or both have an N on the same side (VE.h:778-796).
 * That predicate is order independent: with P / X the sets of prev / next letters
 * (N a letter) over all occurrences, isBif <=> count >= 2 && (|P| > 1 || |X| > 1 ||
 * N in P || N in X).  The oracle accumulates the sets. */
static void table_add(orc_table *t, const uint64_t *key, int prev, int next)
{
    uint64_t hsh = 0;
    for (int i = 0; i < t->C; i++) hsh = mix64(hsh ^ key[i]);
    uint64_t s = hsh & (t->cap - 1);
    for (;;) {
        if (t->count[s] == 0) {
            memcpy(t->keys + s * t->C, key, t->C * sizeof(uint64_t));
            t->n++;
            break;
        }
        if (memcmp(t->keys + s * t->C, key, t->C * sizeof(uint64_t)) == 0) break;
        s = (s + 1) & (t->cap - 1);
    }
    t->count[s]++;
    t->prevm[s] |= (uint8_t)(1u << prev);
    t->nextm[s] |= (uint8_t)(1u << next);
}

static int popc8(unsigned x) { int c = 0; while (x) { c += x & 1; x >>= 1; } return c; }

/* ------------------------------------------------------------------ the passes */

static int within(uint64_t v, uint64_t lo, uint64_t hi) { return v >= lo && v <= hi; } /* VE.h:473-476 */

static uint64_t filter_words(const orc_run *R) { return ((1ull << R->L) >> 5) + 1; } /* concurrentbitvector.cpp:12 */

/* window N-free helper: nfree[g] = number of consecutive non-N codes starting at g, capped */
static uint32_t *build_run_lengths(const orc_run *R)
{
    uint32_t *run = (uint32_t *)malloc((R->ntxt + 1) * sizeof(uint32_t));
    run[R->ntxt] = 0;
    for (uint64_t g = R->ntxt; g-- > 0;) {
        if (R->txt[g] == ORC_N) run[g] = 0;
        else { uint32_t nx = run[g + 1]; run[g] = nx == UINT32_MAX ? nx : nx + 1; }
    }
    return run;
}

/* Split pass: InitialFilterFillerWorker VE.h:503-583 (no N gate; every (k+1)-mer of
 * every dispatched record incl. the sentinels; sequential first-seen semantics = the
 * reference at -t 1).  bins: 2^24 counters, saturating at MAX_COUNTER (common.cpp:6). */
static void split_pass(orc_run *R, uint32_t *bins, uint64_t bin_size)
{
    uint64_t nw = filter_words(R);
    uint32_t *f = (uint32_t *)mmap(NULL, nw * sizeof(uint32_t), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (f == MAP_FAILED) abort();
    uint64_t addr[ORC_MAXQ];
    orc_vhash v;
    for (uint32_t r = 0; r < R->nrec; r++) {
        uint64_t len = R->rec_len[r];
        if (len < (uint64_t)R->k) continue; /* VE.h:1177: never dispatched */
        uint64_t g0 = R->rec_start[r] - 1;  /* sentinel N */
        uint64_t nchars = len + 2;
        vh_init(R, &v, R->txt + g0, R->q);
        for (uint64_t p = 0; p + R->k < nchars; p++) {
            const uint8_t *w = R->txt + g0 + p;
            int prev = w[0], next = w[R->k];
            uint64_t sv = vh_vertex(&v);
            edge_out(R, &v, next, addr);
            int was_set = 1;
            for (int i = 0; i < R->q; i++)
                if (!bit_get(f, addr[i])) { was_set = 0; bit_set(f, addr[i]); }
            vh_update(R, &v, prev, next, R->q);
            uint64_t ev = vh_vertex(&v);
            if (!was_set) {
                uint64_t vals[2] = { sv, ev };
                for (int j = 0; j < 2; j++) {
                    uint64_t b = vals[j] / bin_size;
                    if (bins[b] < (UINT32_MAX >> 1)) bins[b]++;
                }
            }
        }
    }
    munmap(f, nw * sizeof(uint32_t));
}

/* FilterFillerWorker VE.h:995-1105 over the global text. */
static void fill_pass(orc_run *R, const uint32_t *run, uint64_t low, uint64_t high)
{
    uint64_t addr[ORC_MAXQ];
    orc_vhash v;
    int have = 0;
    for (uint64_t g = 1; g + R->k < R->ntxt; g++) {
        if (run[g] < (uint32_t)R->k) { have = 0; continue; }
        const uint8_t *w = R->txt + g;
        if (!have) { vh_init(R, &v, w, R->q); have = 1; }
        int prev = w[-1], next = w[R->k];
        uint64_t first = vh_vertex(&v);
        uint64_t set[64];
        int ns = 0;
        if (next != ORC_N) { edge_out(R, &v, next, addr); for (int i = 0; i < R->q; i++) set[ns++] = addr[i]; }
        else {
            edge_out(R, &v, 0, addr); for (int i = 0; i < R->q; i++) set[ns++] = addr[i]; /* DUMMY_CHAR 'A' VE.h:1012 */
            edge_out(R, &v, 3, addr); for (int i = 0; i < R->q; i++) set[ns++] = addr[i]; /* REV_DUMMY_CHAR 'T' */
        }
        if (prev == ORC_N) { /* VE.h:1054-1058 (pos>0 always holds here: g>=1) */
            edge_in(R, &v, 0, addr); for (int i = 0; i < R->q; i++) set[ns++] = addr[i];
            edge_in(R, &v, 3, addr); for (int i = 0; i < R->q; i++) set[ns++] = addr[i];
        }
        vh_update(R, &v, w[0], next, R->q);
        uint64_t second = vh_vertex(&v);
        if (within(first, low, high) || within(second, low, high))
            for (int i = 0; i < ns; i++)
                if (!bit_get(R->filter, set[i])) bit_set(R->filter, set[i]); /* VE.h:1086-1092 */
    }
}

/* vertexrollinghash.h:208-234 */
static int in_bloom(const orc_run *R, const uint64_t *addr)
{
    for (int i = 0; i < R->q; i++) if (!bit_get(R->filter, addr[i])) return 0;
    return 1;
}

/* CandidateCheckingWorker VE.h:586-704 */
static uint64_t check_pass(orc_run *R, const uint32_t *run, uint64_t low, uint64_t high)
{
    uint64_t addr[ORC_MAXQ], marks = 0;
    orc_vhash v;
    int have = 0;
    for (uint64_t g = 1; g + R->k < R->ntxt; g++) {
        if (run[g] < (uint32_t)R->k) { have = 0; continue; }
        const uint8_t *w = R->txt + g;
        if (!have) { vh_init(R, &v, w, R->q); have = 1; }
        int prev = w[-1], next = w[R->k];
        if (within(vh_vertex(&v), low, high)) {
            int in = prev == ORC_N ? 2 : 0, out = next == ORC_N ? 2 : 0;
            for (int c = 0; c < 4 && in < 2 && out < 2; c++) {
                if (c == prev) in++; else { edge_in(R, &v, c, addr); if (in_bloom(R, addr)) in++; }
                if (c == next) out++; else { edge_out(R, &v, c, addr); if (in_bloom(R, addr)) out++; }
            }
            if (in > 1 || out > 1) { marks++; bit_set(R->rmask, g); }
        }
        vh_update(R, &v, w[0], next, R->q);
    }
    return marks;
}

/* CandidateFinalFilteringWorker VE.h:708-829 + TrueBifurcations VE.h:1228-1256.
 * Appends this round's junction keys to R->keys (unsorted). */
static void filter_pass(orc_run *R, uint64_t marks, uint64_t abundance, int round)
{
    orc_table t;
    table_init(&t, marks, R->C);
    orc_vhash v;
    uint64_t key[32];
    for (uint64_t g = 1; g + R->k < R->ntxt; g++) {
        if (!bit_get(R->rmask, g)) continue;
        const uint8_t *w = R->txt + g;
        int prev = w[-1], next = w[R->k];
        vh_init(R, &v, w, 1);
        if (v.pos[0] < v.neg[0] || (v.pos[0] == v.neg[0] && less_self_rc(w, R->k))) {
            pack_fwd(w, R->k, R->C, key);
            table_add(&t, key, prev, next);
        } else {
            pack_rc(w, R->k, R->C, key);
            table_add(&t, key, rc(next), rc(prev)); /* candidateoccurence.h:43-45 */
        }
    }
    uint64_t tp = 0;
    for (uint64_t s = 0; s < t.cap; s++) {
        if (!t.count[s]) continue;
        int bif = t.count[s] >= 2 && (popc8(t.prevm[s]) > 1 || popc8(t.nextm[s]) > 1 ||
                                      (t.prevm[s] & 16) || (t.nextm[s] & 16));
        if (bif && t.count[s] <= abundance) {
            R->keys = (uint64_t *)realloc(R->keys, (R->nkeys + 1) * R->C * sizeof(uint64_t));
            memcpy(R->keys + R->nkeys * R->C, t.keys + s * R->C, R->C * sizeof(uint64_t));
            R->nkeys++;
            tp++;
        }
    }
    R->r_true[round] = tp;
    R->r_table[round] = t.n;
    R->r_false[round] = t.n - tp;
    table_free(&t);
}

static void out_push(orc_run *R, uint32_t seq, uint32_t pos, int64_t id)
{
    if (R->nout == R->capout) {
        R->capout = R->capout ? R->capout * 2 : 1024;
        R->out_seq = (uint32_t *)realloc(R->out_seq, R->capout * sizeof(uint32_t));
        R->out_pos = (uint32_t *)realloc(R->out_pos, R->capout * sizeof(uint32_t));
        R->out_id = (int64_t *)realloc(R->out_id, R->capout * sizeof(int64_t));
    }
    R->out_seq[R->nout] = seq; R->out_pos[R->nout] = pos; R->out_id[R->nout] = id; R->nout++;
}

/* EdgeConstructionWorker VE.h:856-993 at -t 1 (stub ids in position order). */
static void emit_pass(orc_run *R, const uint32_t *run)
{
    int64_t stub = (int64_t)R->nkeys + 42; /* VE.h:419 */
    for (uint32_t r = 0; r < R->nrec; r++) {
        uint64_t len = R->rec_len[r];
        if (len < (uint64_t)R->k) continue;
        for (uint64_t s = 0; s + R->k <= len; s++) {
            uint64_t g = R->rec_start[r] + s;
            int64_t id = ORC_INVALID_VERTEX;
            if (run[g] >= (uint32_t)R->k && bit_get(R->mask, g)) {
                id = get_id(R, R->txt + g);
                if (id != ORC_INVALID_VERTEX) out_push(R, r, (uint32_t)s, id);
            }
            if ((s == 0 || s + R->k == len) && id == ORC_INVALID_VERTEX) out_push(R, r, (uint32_t)s, stub++);
        }
    }
    R->true_marks = R->nout;
}

/* The constructor body VE.h:122-466: split pass + round planner (VE.h:206-254, 391),
 * per round fill/check/filter, sort (bifurcationstorage.h:65), output pass. */
int orc_enumerate(orc_run *R, int rounds, uint64_t abundance)
{
    if (rounds < 1 || rounds > 64) { snprintf(R->err, sizeof R->err, "bad rounds"); return -1; }
    if (R->C >= 20) { /* vertexenumerator.cpp:56-70, MAX_CAPACITY VE.h:4 */
        snprintf(R->err, sizeof R->err, "The value of K is too big. Please refer to documentaion how to increase the max supported value of K.");
        return -1;
    }
    const uint64_t BINS = 1ull << 24; /* VE.h:471 */
    uint64_t real_size = 1ull << R->L;
    uint64_t bin_size = real_size / BINS > 1 ? real_size / BINS : 1; /* VE.h:169 */
    uint32_t *bins = NULL;
    double round_size = 0;
    if (rounds > 1) {
        bins = (uint32_t *)calloc(BINS, sizeof(uint32_t));
        split_pass(R, bins, bin_size);
        uint64_t tot = 0;
        for (uint64_t i = 0; i < BINS; i++) tot += bins[i];
        round_size = (double)tot / rounds; /* VE.h:209 */
    }
    uint32_t *run = build_run_lengths(R);
    uint64_t mw = (R->ntxt >> 5) + 1;
    free(R->mask); free(R->rmask); filter_free(R); free(R->keys);
    R->mask = (uint32_t *)calloc(mw, sizeof(uint32_t));
    R->rmask = (uint32_t *)calloc(mw, sizeof(uint32_t));
    R->filter = NULL;
    R->keys = NULL; R->nkeys = 0; R->nout = 0; R->rounds = rounds;
    uint64_t low = 0, high = real_size, low_boundary = 0;
    for (int round = 0; round < rounds; round++) {
        if (rounds > 1) { /* VE.h:234-250 */
            uint64_t acc = bins[low_boundary];
            for (++low_boundary; low_boundary < BINS; ++low_boundary) {
                if ((double)acc <= round_size || round + 1 == rounds) acc += bins[low_boundary];
                else break;
            }
            high = low_boundary * bin_size;
        } else high = real_size;
        R->r_low[round] = low; R->r_high[round] = high;
        if (filter_zero(R)) { free(run); free(bins); return -1; }
        memset(R->rmask, 0, mw * sizeof(uint32_t));
        fill_pass(R, run, low, high);
        R->r_marks[round] = check_pass(R, run, low, high);
        filter_pass(R, R->r_marks[round], abundance, round);
        for (uint64_t i = 0; i < mw; i++) R->mask[i] |= R->rmask[i]; /* MergeOr VE.h:909-913 */
        low = high + 1; /* VE.h:391 */
    }
    free(bins);
    key_cmp_C = R->C;
    qsort(R->keys, R->nkeys, R->C * sizeof(uint64_t), key_cmp);
    emit_pass(R, run);
    free(run);
    return 0;
}

/* JunctionPositionWriter common/junctionapi.h:107-137: 12-byte LE records, one
 * separator (0xFFFFFFFF, INT64_MAX) per sequence-id step before a later sequence. */
int orc_write_bin(const orc_run *R, const char *path)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    uint32_t now = 0;
    for (uint64_t i = 0; i < R->nout; i++) {
        for (; R->out_seq[i] > now; ++now) {
            uint32_t sp = 0xFFFFFFFFu; int64_t sb = INT64_MAX;
            fwrite(&sp, 4, 1, f); fwrite(&sb, 8, 1, f);
        }
        fwrite(&R->out_pos[i], 4, 1, f);
        fwrite(&R->out_id[i], 8, 1, f);
    }
    fclose(f);
    return 0;
}

/* ------------------------------------------------------------------ accessors */
uint64_t orc_text_len(const orc_run *R) { return R->ntxt; }
const uint8_t *orc_text(const orc_run *R) { return R->txt; }
uint32_t orc_num_records(const orc_run *R) { return R->nrec; }
const uint64_t *orc_rec_start(const orc_run *R) { return R->rec_start; }
const uint64_t *orc_rec_len(const orc_run *R) { return R->rec_len; }
uint64_t orc_filter_nwords(const orc_run *R) { return filter_words(R); }
const uint32_t *orc_filter(const orc_run *R) { return R->filter; }
uint64_t orc_mask_nwords(const orc_run *R) { return (R->ntxt >> 5) + 1; }
const uint32_t *orc_mask(const orc_run *R) { return R->mask; }
const uint32_t *orc_round_mask(const orc_run *R) { return R->rmask; }
int orc_capacity(const orc_run *R) { return R->C; }
uint64_t orc_num_keys(const orc_run *R) { return R->nkeys; }
const uint64_t *orc_keys(const orc_run *R) { return R->keys; }
uint64_t orc_num_out(const orc_run *R) { return R->nout; }
const uint32_t *orc_out_seq(const orc_run *R) { return R->out_seq; }
const uint32_t *orc_out_pos(const orc_run *R) { return R->out_pos; }
const int64_t *orc_out_id(const orc_run *R) { return R->out_id; }
uint64_t orc_round_stat(const orc_run *R, int round, int what)
{
    switch (what) {
    case 0: return R->r_true[round];
    case 1: return R->r_false[round];
    case 2: return R->r_table[round];
    case 3: return R->r_marks[round];
    case 4: return R->r_low[round];
    case 5: return R->r_high[round];
    }
    return 0;
}
uint64_t orc_true_marks(const orc_run *R) { return R->true_marks; }

/* Debug tap: pos/neg vertex hashes of the window at g for fn 0..q-1 (2q values), and the
 * q canonical out-edge addresses for next char c (or 0 values when c > 4). */
void orc_hash_dump(const orc_run *R, uint64_t g, uint64_t *posneg, int c, uint64_t *addr)
{
    orc_vhash v;
    vh_init(R, &v, R->txt + g, R->q);
    for (int i = 0; i < R->q; i++) { posneg[2 * i] = v.pos[i]; posneg[2 * i + 1] = v.neg[i]; }
    if (c >= 0 && c <= 4) edge_out(R, &v, c, addr);
}

/* ---- per-round primitives (used by the multi-process tests: one vertex-hash range per rank) ---- */
int orc_dist_begin(orc_run *R)
{
    uint64_t mw = (R->ntxt >> 5) + 1;
    free(R->mask); free(R->rmask); filter_free(R); free(R->keys);
    R->mask = (uint32_t *)calloc(mw, sizeof(uint32_t));
    R->rmask = (uint32_t *)calloc(mw, sizeof(uint32_t));
    R->filter = NULL;
    R->keys = NULL; R->nkeys = 0; R->nout = 0;
    return 0;
}

/* one round of VE.h:228-392 for the range [low,high]; stats = true,false,table,marks */
int orc_dist_round(orc_run *R, uint64_t low, uint64_t high, uint64_t abundance, uint64_t *stats)
{
    uint32_t *run = build_run_lengths(R);
    uint64_t mw = (R->ntxt >> 5) + 1;
    if (filter_zero(R)) { free(run); return -1; }
    memset(R->rmask, 0, mw * sizeof(uint32_t));
    fill_pass(R, run, low, high);
    uint64_t marks = check_pass(R, run, low, high);
    filter_pass(R, marks, abundance, 0);
    for (uint64_t i = 0; i < mw; i++) R->mask[i] |= R->rmask[i];
    stats[0] = R->r_true[0]; stats[1] = R->r_false[0]; stats[2] = R->r_table[0]; stats[3] = marks;
    free(run);
    return 0;
}

int orc_set_keys(orc_run *R, const uint64_t *keys, uint64_t n)
{
    free(R->keys);
    R->keys = (uint64_t *)malloc((n + 1) * R->C * sizeof(uint64_t));
    memcpy(R->keys, keys, n * R->C * sizeof(uint64_t));
    R->nkeys = n;
    key_cmp_C = R->C;
    qsort(R->keys, R->nkeys, R->C * sizeof(uint64_t), key_cmp);
    return 0;
}

/* ids of the marked positions of the run-wide mask, increasing g; returns their number */
uint64_t orc_lookup_marks(orc_run *R, uint64_t *g_out, int64_t *id_out, uint64_t cap)
{
    uint64_t n = 0;
    for (uint64_t g = 1; g + R->k < R->ntxt; g++) {
        if (!bit_get(R->mask, g)) continue;
        if (n < cap) { g_out[n] = g; id_out[n] = get_id(R, R->txt + g); }
        n++;
    }
    return n;
}

/* Split-pass bins only (2^24 counters, caller-provided). */
int orc_split_bins(orc_run *R, uint32_t *bins)
{
    const uint64_t BINS = 1ull << 24;
    uint64_t real_size = 1ull << R->L;
    uint64_t bin_size = real_size / BINS > 1 ? real_size / BINS : 1;
    memset(bins, 0, BINS * sizeof(uint32_t));
    split_pass(R, bins, bin_size);
    return 0;
}

/* Fill only (for filter-bitmap parity and the cpu_baseline insert timing). */
int orc_fill_only(orc_run *R, uint64_t low, uint64_t high)
{
    uint32_t *run = build_run_lengths(R);
    if (filter_zero(R)) { free(run); return -1; }
    fill_pass(R, run, low, high);
    free(run);
    return 0;
}

/* Query only against the current filter; returns marks and leaves the mask in rmask. */
uint64_t orc_check_only(orc_run *R, uint64_t low, uint64_t high)
{
    uint32_t *run = build_run_lengths(R);
    uint64_t mw = (R->ntxt >> 5) + 1;
    free(R->rmask);
    R->rmask = (uint32_t *)calloc(mw, sizeof(uint32_t));
    uint64_t m = check_pass(R, run, low, high);
    free(run);
    return m;
}
