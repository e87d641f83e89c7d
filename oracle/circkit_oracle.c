/*
 * circkit_oracle.c -- CPU restatement of the circkit `canonicalize` / `uniq` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under circkit_amd/ (the product) may include,
 * link or call this file.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg use it, and only as the checker / the reported CPU baseline.
 *
 * Parity status: PINNED by the reference's own known-answer tests and fixtures
 * (lib/src/canonicalize.rs:69-132, tests/canon_uniq.rs:24-29, tests/examples/{simple,
 * multiple_sequences,multiple_sequences_split_lines,rna_input,repeated,compressed_*});
 * see tests/test_oracle_golden.py.  The reference is Rust and cannot be built in this
 * image (no cargo/rustc), so there is no oracle/_ref build.  Third-party arithmetic
 * that is not under /root/reference is restated from the published algorithm of the
 * pinned crate version:
 *   bio 1.3.1        alphabets::dna::revcomp     (call site lib/src/canonicalize.rs:56)
 *   needletail 0.5.1 sequence::normalize(_,false) (call sites src/canonicalize.rs:24, src/uniq.rs:35)
 *   xxhash-rust 0.8.6 xxh3::xxh3_64 == XXH3-64, seed 0, default secret (call site src/uniq.rs:45);
 *                    pinned here against the Python `xxhash` package (tests/golden/xxh3_vectors.json).
 *
 * All functions work on unsigned bytes.  The reference panics on non-UTF-8 input
 * (lib/src/canonicalize.rs:6); for ASCII input char order == byte order, which is the
 * only domain on which parity is claimed.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

/* ------------------------------------------------------------------------------------------
 * lmsr_index -- literal transcription of lib/src/canonicalize.rs:5-36 (byte indexed).
 * Signed arithmetic as in the reference (isize, :7-14): r may become l-1 before the r+=1.
 * ---------------------------------------------------------------------------------------- */
size_t ck_oracle_lmsr_index(const uint8_t *s, size_t len)
{
    int64_t n = (int64_t)len;
    int64_t res = 0, l = 0;
    while (l < n) {                                   /* :11 */
        res = l;                                      /* :12 */
        int64_t r = l, p = l + 1;                     /* :13-14 */
        while (r < n) {                               /* :16 */
            uint8_t c = (p < n) ? s[p] : s[p - n];    /* :17-21 */
            if (s[r] > c) break;                      /* :22-24 */
            if (s[r] < c) r = l - 1;                  /* :25-27 */
            r += 1;                                   /* :28 */
            p += 1;                                   /* :29 */
        }
        int64_t a = r, b = l + p - r;                 /* :32 */
        l = a > b ? a : b;
    }
    return (size_t)res;                               /* :35 */
}

/* ------------------------------------------------------------------------------------------
 * COST MODEL of the reference as written (bench.py: cpu_baseline.reference_faithful_quadratic).
 * lib/src/canonicalize.rs:17-27 index the &str with `s.chars().nth(i)`: Chars::nth walks the UTF-8 text from its
 * start, one char boundary at a time -- O(i) per access, three accesses per step of the inner loop (:18/:20, :22, :25),
 * hence O(n^2) per record where the byte-indexed transcription above is linear.  Same answers (ASCII: every byte is a
 * char); only the time differs.  How fast the real binary's walk is depends on the rustc that built it (newer std
 * count boundaries a word at a time), so this is a model of the SHAPE of the cost, not a stand-in for the binary.
 * ---------------------------------------------------------------------------------------- */
static uint8_t chars_nth(const uint8_t *s, size_t len, size_t k)
{
    size_t seen = 0;
    for (size_t i = 0; i < len; ++i) {
        if ((s[i] & 0xC0) != 0x80) {                  /* not a continuation byte: a char starts here */
            if (seen == k) return s[i];
            ++seen;
        }
    }
    return 0;                                         /* (the reference would panic on the unwrap) */
}

size_t ck_oracle_lmsr_index_nth(const uint8_t *s, size_t len)
{
    int64_t n = (int64_t)len;
    int64_t res = 0, l = 0;
    while (l < n) {
        res = l;
        int64_t r = l, p = l + 1;
        while (r < n) {
            uint8_t c = (p < n) ? chars_nth(s, len, (size_t)p) : chars_nth(s, len, (size_t)(p - n));   /* :17-21 */
            if (chars_nth(s, len, (size_t)r) > c) break;                                               /* :22 */
            if (chars_nth(s, len, (size_t)r) < c) r = l - 1;                                           /* :25 */
            r += 1;
            p += 1;
        }
        int64_t a = r, b = l + p - r;
        l = a > b ? a : b;
    }
    return (size_t)res;
}

/* The naive oracle of the reference's own proptest (lib/src/canonicalize.rs:154-164):
 * smallest i whose rotation is lexicographically minimal (strict `<`, :159). */
size_t ck_oracle_lmsr_index_simple(const uint8_t *s, size_t n)
{
    size_t best = 0;
    for (size_t i = 1; i < n; ++i) {
        /* compare rotation i with rotation best */
        size_t k = 0;
        while (k < n) {
            uint8_t a = s[(i + k) % n], b = s[(best + k) % n];
            if (a != b) { if (a < b) best = i; break; }
            ++k;
        }
    }
    return best;
}

/* lmsr -- lib/src/canonicalize.rs:41-47: s[i..] ++ s[..i] */
void ck_oracle_lmsr(const uint8_t *s, size_t n, uint8_t *out)
{
    size_t i = ck_oracle_lmsr_index(s, n);
    memcpy(out, s + i, n - i);
    memcpy(out + (n - i), s, i);
}

/* bio 1.3.1 alphabets::dna complement table: identity, then
 * "AGCTYRWSKMDVHBN" -> "TCGARYWSMKHBDVN" and the same +32 (lower case). */
static uint8_t g_comp[256];
static int g_comp_ready = 0;
static void comp_init(void)
{
    if (g_comp_ready) return;
    for (int v = 0; v < 256; ++v) g_comp[v] = (uint8_t)v;
    const char *a = "AGCTYRWSKMDVHBN", *b = "TCGARYWSMKHBDVN";
    for (int i = 0; a[i]; ++i) {
        g_comp[(uint8_t)a[i]] = (uint8_t)b[i];
        g_comp[(uint8_t)a[i] + 32] = (uint8_t)(b[i] + 32);
    }
    g_comp_ready = 1;
}

uint8_t ck_oracle_complement(uint8_t c) { comp_init(); return g_comp[c]; }

/* revcomp -- bio: text.iter().rev().map(complement) */
void ck_oracle_revcomp(const uint8_t *s, size_t n, uint8_t *out)
{
    comp_init();
    for (size_t i = 0; i < n; ++i) out[i] = g_comp[s[n - 1 - i]];
}

/* canonicalize -- lib/src/canonicalize.rs:54-63.
 * a = lmsr(s); b = lmsr(revcomp(a)); return a if a < b else b (slice order: unsigned bytes). */
void ck_oracle_canonicalize(const uint8_t *s, size_t n, uint8_t *out)
{
    if (n == 0) return;
    uint8_t *a = (uint8_t *)malloc(n), *rc = (uint8_t *)malloc(n), *b = (uint8_t *)malloc(n);
    ck_oracle_lmsr(s, n, a);                 /* :55 */
    ck_oracle_revcomp(a, n, rc);             /* :56 */
    ck_oracle_lmsr(rc, n, b);                /* :56 */
    if (memcmp(a, b, n) < 0) memcpy(out, a, n);   /* :58-59 */
    else memcpy(out, b, n);                       /* :60-61 */
    free(a); free(rc); free(b);
}

/* canonicalize with the quadratic lmsr_index above (cost model) */
static void canonicalize_nth(const uint8_t *s, size_t n, uint8_t *out)
{
    if (n == 0) return;
    uint8_t *a = (uint8_t *)malloc(n), *rc = (uint8_t *)malloc(n), *b = (uint8_t *)malloc(n);
    size_t i = ck_oracle_lmsr_index_nth(s, n);
    memcpy(a, s + i, n - i); memcpy(a + (n - i), s, i);
    ck_oracle_revcomp(a, n, rc);
    i = ck_oracle_lmsr_index_nth(rc, n);
    memcpy(b, rc + i, n - i); memcpy(b + (n - i), rc, i);
    memcpy(out, memcmp(a, b, n) < 0 ? a : b, n);
    free(a); free(rc); free(b);
}

/* needletail 0.5.1 sequence::normalize(seq, iupac=false).
 * Returns the output length; *changed is set when the reference would return Some(..)
 * (any byte altered or dropped); when it returns None the caller uses the raw bytes
 * (src/canonicalize.rs:24-27) -- identical content either way. */
size_t ck_oracle_normalize(const uint8_t *s, size_t n, uint8_t *out, int *changed)
{
    size_t m = 0; int ch = 0;
    for (size_t i = 0; i < n; ++i) {
        uint8_t c = s[i], o;
        switch (c) {
        case 'A': case 'C': case 'G': case 'T': case 'N': case '-': o = c; break;
        case 'a': o = 'A'; ch = 1; break;
        case 'c': o = 'C'; ch = 1; break;
        case 'g': o = 'G'; ch = 1; break;
        case 't': case 'u': case 'U': o = 'T'; ch = 1; break;
        case '.': case '~': o = '-'; ch = 1; break;
        case ' ': case '\t': case '\r': case '\n': o = ' '; ch = 1; break;
        default: o = 'N'; ch = 1; break;
        }
        if (o != ' ') out[m++] = o;
    }
    if (changed) *changed = ch;
    return m;
}

/* ------------------------------------------------------------------------------------------
 * XXH3-64, seed 0, default secret (what xxhash_rust::xxh3::xxh3_64 computes, src/uniq.rs:45).
 * Written from the published XXH3 specification.
 * ---------------------------------------------------------------------------------------- */
static const uint8_t kSecret[192] = {
    0xb8, 0xfe, 0x6c, 0x39, 0x23, 0xa4, 0x4b, 0xbe, 0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad, 0x1c,
    0xde, 0xd4, 0x6d, 0xe9, 0x83, 0x90, 0x97, 0xdb, 0x72, 0x40, 0xa4, 0xa4, 0xb7, 0xb3, 0x67, 0x1f,
    0xcb, 0x79, 0xe6, 0x4e, 0xcc, 0xc0, 0xe5, 0x78, 0x82, 0x5a, 0xd0, 0x7d, 0xcc, 0xff, 0x72, 0x21,
    0xb8, 0x08, 0x46, 0x74, 0xf7, 0x43, 0x24, 0x8e, 0xe0, 0x35, 0x90, 0xe6, 0x81, 0x3a, 0x26, 0x4c,
    0x3c, 0x28, 0x52, 0xbb, 0x91, 0xc3, 0x00, 0xcb, 0x88, 0xd0, 0x65, 0x8b, 0x1b, 0x53, 0x2e, 0xa3,
    0x71, 0x64, 0x48, 0x97, 0xa2, 0x0d, 0xf9, 0x4e, 0x38, 0x19, 0xef, 0x46, 0xa9, 0xde, 0xac, 0xd8,
    0xa8, 0xfa, 0x76, 0x3f, 0xe3, 0x9c, 0x34, 0x3f, 0xf9, 0xdc, 0xbb, 0xc7, 0xc7, 0x0b, 0x4f, 0x1d,
    0x8a, 0x51, 0xe0, 0x4b, 0xcd, 0xb4, 0x59, 0x31, 0xc8, 0x9f, 0x7e, 0xc9, 0xd9, 0x78, 0x73, 0x64,
    0xea, 0xc5, 0xac, 0x83, 0x34, 0xd3, 0xeb, 0xc3, 0xc5, 0x81, 0xa0, 0xff, 0xfa, 0x13, 0x63, 0xeb,
    0x17, 0x0d, 0xdd, 0x51, 0xb7, 0xf0, 0xda, 0x49, 0xd3, 0x16, 0x55, 0x26, 0x29, 0xd4, 0x68, 0x9e,
    0x2b, 0x16, 0xbe, 0x58, 0x7d, 0x47, 0xa1, 0xfc, 0x8f, 0xf8, 0xb8, 0xd1, 0x7a, 0xd0, 0x31, 0xce,
    0x45, 0xcb, 0x3a, 0x8f, 0x95, 0x16, 0x04, 0x28, 0xaf, 0xd7, 0xfb, 0xca, 0xbb, 0x4b, 0x40, 0x7e,
};
#define P32_1 0x9E3779B1U
#define P32_2 0x85EBCA77U
#define P32_3 0xC2B2AE3DU
#define P64_1 0x9E3779B185EBCA87ULL
#define P64_2 0xC2B2AE3D27D4EB4FULL
#define P64_3 0x165667B19E3779F9ULL
#define P64_4 0x85EBCA77C2B2AE63ULL
#define P64_5 0x27D4EB2F165667C5ULL
#define PMX1  0x165667919E3779F9ULL
#define PMX2  0x9FB21C651E98DF25ULL

static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }   /* little-endian host */
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t bswap64(uint64_t x) { return __builtin_bswap64(x); }
static inline uint64_t mul128_fold64(uint64_t a, uint64_t b)
{
    __uint128_t p = (__uint128_t)a * b;
    return (uint64_t)p ^ (uint64_t)(p >> 64);
}
static inline uint64_t xxh3_avalanche(uint64_t h) { h ^= h >> 37; h *= PMX1; h ^= h >> 32; return h; }
static inline uint64_t xxh64_avalanche(uint64_t h)
{
    h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; h ^= h >> 32; return h;
}
static inline uint64_t mix16(const uint8_t *in, const uint8_t *sec)
{
    return mul128_fold64(rd64(in) ^ rd64(sec), rd64(in + 8) ^ rd64(sec + 8));
}
static inline void acc512(uint64_t acc[8], const uint8_t *in, const uint8_t *sec)
{
    for (int i = 0; i < 8; ++i) {
        uint64_t dv = rd64(in + 8 * i), dk = dv ^ rd64(sec + 8 * i);
        acc[i ^ 1] += dv;
        acc[i] += (uint64_t)(uint32_t)dk * (dk >> 32);
    }
}

uint64_t ck_oracle_xxh3_64(const uint8_t *in, size_t len)
{
    if (len == 0) return xxh64_avalanche(rd64(kSecret + 56) ^ rd64(kSecret + 64));
    if (len <= 3) {
        uint32_t c1 = in[0], c2 = in[len >> 1], c3 = in[len - 1];
        uint32_t comb = (c1 << 16) | (c2 << 24) | c3 | ((uint32_t)len << 8);
        uint64_t flip = (uint64_t)(rd32(kSecret) ^ rd32(kSecret + 4));
        return xxh64_avalanche((uint64_t)comb ^ flip);
    }
    if (len <= 8) {
        uint32_t i1 = rd32(in), i2 = rd32(in + len - 4);
        uint64_t flip = rd64(kSecret + 8) ^ rd64(kSecret + 16);
        uint64_t h = ((uint64_t)i2 + ((uint64_t)i1 << 32)) ^ flip;
        h ^= rotl64(h, 49) ^ rotl64(h, 24);
        h *= PMX2; h ^= (h >> 35) + len; h *= PMX2;
        return h ^ (h >> 28);
    }
    if (len <= 16) {
        uint64_t f1 = rd64(kSecret + 24) ^ rd64(kSecret + 32), f2 = rd64(kSecret + 40) ^ rd64(kSecret + 48);
        uint64_t lo = rd64(in) ^ f1, hi = rd64(in + len - 8) ^ f2;
        return xxh3_avalanche(len + bswap64(lo) + hi + mul128_fold64(lo, hi));
    }
    if (len <= 128) {
        uint64_t acc = len * P64_1;
        if (len > 32) {
            if (len > 64) {
                if (len > 96) { acc += mix16(in + 48, kSecret + 96); acc += mix16(in + len - 64, kSecret + 112); }
                acc += mix16(in + 32, kSecret + 64); acc += mix16(in + len - 48, kSecret + 80);
            }
            acc += mix16(in + 16, kSecret + 32); acc += mix16(in + len - 32, kSecret + 48);
        }
        acc += mix16(in, kSecret); acc += mix16(in + len - 16, kSecret + 16);
        return xxh3_avalanche(acc);
    }
    if (len <= 240) {
        uint64_t acc = len * P64_1;
        size_t rounds = len / 16;
        for (size_t i = 0; i < 8; ++i) acc += mix16(in + 16 * i, kSecret + 16 * i);
        acc = xxh3_avalanche(acc);
        for (size_t i = 8; i < rounds; ++i) acc += mix16(in + 16 * i, kSecret + 16 * (i - 8) + 3);
        acc += mix16(in + len - 16, kSecret + 136 - 17);
        return xxh3_avalanche(acc);
    }
    uint64_t acc[8] = { P32_3, P64_1, P64_2, P64_3, P64_4, P32_2, P64_5, P32_1 };
    const size_t spb = (192 - 64) / 8, blk = 64 * spb;
    size_t nb = (len - 1) / blk;
    for (size_t b = 0; b < nb; ++b) {
        for (size_t s = 0; s < spb; ++s) acc512(acc, in + b * blk + 64 * s, kSecret + 8 * s);
        for (int i = 0; i < 8; ++i) {
            uint64_t a = acc[i]; a ^= a >> 47; a ^= rd64(kSecret + 192 - 64 + 8 * i); a *= P32_1; acc[i] = a;
        }
    }
    size_t ns = ((len - 1) - blk * nb) / 64;
    for (size_t s = 0; s < ns; ++s) acc512(acc, in + nb * blk + 64 * s, kSecret + 8 * s);
    acc512(acc, in + len - 64, kSecret + 192 - 64 - 7);
    uint64_t r = len * P64_1;
    for (int i = 0; i < 4; ++i)
        r += mul128_fold64(acc[2 * i] ^ rd64(kSecret + 11 + 16 * i), acc[2 * i + 1] ^ rd64(kSecret + 11 + 16 * i + 8));
    return xxh3_avalanche(r);
}

/* ------------------------------------------------------------------------------------------
 * CSR batch drivers: bytes[offsets[i]..offsets[i+1]) is record i (already normalized), the
 * same layout the product's C ABI takes.  Mirrors the worker closure body after normalize
 * (src/canonicalize.rs:29, src/uniq.rs:40) applied to every record; `threads` plays the role
 * of the reference's --threads worker pool (src/canonicalize.rs:19).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *bytes; const uint64_t *off; uint8_t *out; uint64_t *hash;
    uint64_t lo, hi;
    int nth;                                          /* the quadratic cost model instead of the linear transcription */
} job_t;

static void *batch_worker(void *arg)
{
    job_t *j = (job_t *)arg;
    for (uint64_t i = j->lo; i < j->hi; ++i) {
        uint64_t o = j->off[i], n = j->off[i + 1] - o;
        if (j->nth) {
            canonicalize_nth(j->bytes + o, n, j->out + o);
        } else if (j->out) {
            ck_oracle_canonicalize(j->bytes + o, n, j->out + o);
            if (j->hash) j->hash[i] = ck_oracle_xxh3_64(j->out + o, n);
        } else if (j->hash) {
            uint8_t *tmp = (uint8_t *)malloc(n ? n : 1);
            ck_oracle_canonicalize(j->bytes + o, n, tmp);
            j->hash[i] = ck_oracle_xxh3_64(tmp, n);
            free(tmp);
        }
    }
    return NULL;
}

static void canonicalize_batch(const uint8_t *bytes, const uint64_t *offsets, uint64_t n_records,
                               uint8_t *out_bytes, uint64_t *out_xxh3, int threads, int nth);
void ck_oracle_canonicalize_batch(const uint8_t *bytes, const uint64_t *offsets, uint64_t n_records,
                                  uint8_t *out_bytes, uint64_t *out_xxh3, int threads)
{
    canonicalize_batch(bytes, offsets, n_records, out_bytes, out_xxh3, threads, 0);
}
/* the same batch through the quadratic cost model (out_bytes required) */
void ck_oracle_canonicalize_batch_nth(const uint8_t *bytes, const uint64_t *offsets, uint64_t n_records,
                                      uint8_t *out_bytes, int threads)
{
    canonicalize_batch(bytes, offsets, n_records, out_bytes, NULL, threads, 1);
}
static void canonicalize_batch(const uint8_t *bytes, const uint64_t *offsets, uint64_t n_records,
                               uint8_t *out_bytes, uint64_t *out_xxh3, int threads, int nth)
{
    comp_init();
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n_records) threads = n_records ? (int)n_records : 1;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    job_t *jobs = (job_t *)malloc(sizeof(job_t) * threads);
    for (int t = 0; t < threads; ++t) {
        jobs[t].bytes = bytes; jobs[t].off = offsets; jobs[t].out = out_bytes; jobs[t].hash = out_xxh3; jobs[t].nth = nth;
        jobs[t].lo = n_records * t / threads; jobs[t].hi = n_records * (t + 1) / threads;
        if (threads == 1) batch_worker(&jobs[t]);
        else pthread_create(&tid[t], NULL, batch_worker, &jobs[t]);
    }
    if (threads > 1) for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
    free(tid); free(jobs);
}

/* uniq first-seen resolution -- src/uniq.rs:42-78: records visited in input order; a record
 * is kept iff its hash has not been seen; first_seen[i] = index of the record that owns the
 * hash (== i for kept records).  Equality is hash-only (src/uniq.rs:27,47). */
void ck_oracle_uniq_first_seen(const uint64_t *hash, uint64_t n, uint64_t *first_seen)
{
    uint64_t cap = 16; while (cap < 2 * n + 2) cap <<= 1;
    uint64_t *keys = (uint64_t *)malloc(cap * 8), *vals = (uint64_t *)malloc(cap * 8);
    uint8_t *used = (uint8_t *)calloc(cap, 1);
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t h = hash[i], s = (h * 0x9E3779B97F4A7C15ULL) & (cap - 1);
        while (used[s] && keys[s] != h) s = (s + 1) & (cap - 1);
        if (!used[s]) { used[s] = 1; keys[s] = h; vals[s] = i; }
        first_seen[i] = vals[s];
    }
    free(keys); free(vals); free(used);
}

/* ------------------------------------------------------------------------------------------
 * Synthetic input generator: base b of record r is drawn from a counter-based
 * PRNG so host and device produce identical bytes.  One splitmix64 output covers 32 bases of
 * the flat base stream: word index w = (global_base_index >> 5), bits 2k..2k+1 select base k.
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

void ck_oracle_synth_fill(uint64_t seed, uint64_t first_base, uint64_t n_bases, uint8_t *out)
{
    static const char L[4] = { 'A', 'C', 'G', 'T' };
    for (uint64_t i = 0; i < n_bases; ++i) {
        uint64_t g = first_base + i;
        uint64_t w = splitmix64(seed * 0xD1342543DE82EF95ULL + (g >> 5));
        out[i] = (uint8_t)L[(w >> (2 * (g & 31))) & 3];
    }
}
