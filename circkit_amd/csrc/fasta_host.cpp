// fasta_host.cpp -- see fasta_host.h.  Host logic only.
#include "fasta_host.h"

#include <stdlib.h>
#include <string.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "../../include/circkit.h"

namespace ckhost {

ByteBuf::~ByteBuf()
{
    if (p) { if (release) release(p); else free(p); }
}

void ByteBuf::reserve(size_t n)
{
    if (n <= cap) return;
    size_t c = cap ? cap : 4096;
    while (c < n) c += c / 2 + 4096;
    uint8_t* q = (uint8_t*)(alloc ? alloc(c) : malloc(c));
    if (!q) abort();
    if (len) memcpy(q, p, len);
    if (p) { if (release) release(p); else free(p); }
    p = q;
    cap = c;
}

const uint8_t* normalize_lut()
{
    // needletail 0.5.1 sequence::normalize(_, iupac = false): ACGTN- kept, acg -> upper, t/u/U -> T,
    // . ~ -> -, whitespace dropped, everything else -> N.  A function-local static with an initializer is built once,
    // thread-safely (parser threads call this concurrently on their first chunk).
    struct Table {
        uint8_t v[256];
        Table()
        {
            for (int i = 0; i < 256; ++i) v[i] = 'N';
            const char* keep = "ACGTN-";
            for (int i = 0; keep[i]; ++i) v[(uint8_t)keep[i]] = (uint8_t)keep[i];
            v['a'] = 'A'; v['c'] = 'C'; v['g'] = 'G'; v['t'] = 'T'; v['u'] = 'T'; v['U'] = 'T';
            v['.'] = '-'; v['~'] = '-';
            v[' '] = 0; v['\t'] = 0; v['\r'] = 0; v['\n'] = 0;
        }
    };
    static const Table table;
    return table.v;
}

bool parse_chunk(const uint8_t* text, size_t n, bool first_chunk, bool final_chunk, Batch& out, size_t* consumed,
                 std::string& err)
{
    out.clear();
    out.text = text;
    size_t pos = 0;
    if (first_chunk) {
        while (pos < n && (text[pos] == '\n' || text[pos] == '\r')) ++pos;      // leading blank lines
        if (pos < n && text[pos] != '>') {
            err = "FASTA parse error: expected '>' at the start of the first record";
            return false;
        }
    }
    // where may we stop?  A record is complete once the start of the next one ("\n>") has been seen.
    size_t limit = n;
    if (!final_chunk) {
        size_t k = n;
        limit = pos;
        while (k > pos + 1) {
            const void* p = memrchr(text + pos, '>', k - pos);
            if (!p) break;
            const size_t g = (const uint8_t*)p - text;
            if (g > pos && text[g - 1] == '\n') { limit = g; break; }
            k = g;
        }
    }
    const uint8_t* lut = normalize_lut();
    out.bytes.reserve(limit - pos + 64);        // normalized never exceeds raw: one allocation, no growth below
    uint8_t* const payload = out.bytes.data();
    size_t plen = 0;
    out.offsets.push_back(0);
    while (pos < limit) {
        // header line
        const uint8_t* eol = (const uint8_t*)memchr(text + pos, '\n', limit - pos);
        size_t hend = eol ? (size_t)(eol - text) : limit;
        size_t hlen = hend - (pos + 1);
        if (hlen && text[pos + 1 + hlen - 1] == '\r') --hlen;
        out.head.push_back(Span{ pos + 1, hlen });
        size_t s0 = eol ? hend + 1 : limit;
        // sequence: up to the next "\n>" (or the chunk limit)
        size_t s1 = limit, next = limit;
        for (size_t k = s0; k < limit;) {
            const uint8_t* g = (const uint8_t*)memchr(text + k, '>', limit - k);
            if (!g) break;
            const size_t gi = g - text;
            if (gi == s0 || text[gi - 1] == '\n') { s1 = gi; next = gi; break; }
            k = gi + 1;
        }
        size_t rlen = s1 - s0;
        if (rlen && text[s0 + rlen - 1] == '\n') --rlen;       // final line terminator is not part of seq()
        if (rlen && text[s0 + rlen - 1] == '\r') --rlen;
        out.raw.push_back(Span{ s0, rlen });
        // normalize while packing.  Sixteen bytes at a time where all sixteen are A, C, G, T or N as they stand (the bulk of
        // any nucleotide file: they map to themselves and none is dropped) -- five compares and a store; a block with anything
        // else in it (a line break, lower case, U, IUPAC codes) goes byte by byte through the table.
        uint8_t* dst = payload + plen;
        size_t m = 0, k = 0;
#if defined(__SSE2__)
        {
            const __m128i cA = _mm_set1_epi8('A'), cC = _mm_set1_epi8('C'), cG = _mm_set1_epi8('G'), cT = _mm_set1_epi8('T'), cN = _mm_set1_epi8('N');
            const uint8_t* src = text + s0;
            while (k + 16 <= rlen) {
                const __m128i x = _mm_loadu_si128((const __m128i*)(src + k));
                const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(x, cA), _mm_cmpeq_epi8(x, cC)),
                                                             _mm_or_si128(_mm_cmpeq_epi8(x, cG), _mm_cmpeq_epi8(x, cT))), _mm_cmpeq_epi8(x, cN));
                if (_mm_movemask_epi8(ok) == 0xFFFF) {
                    _mm_storeu_si128((__m128i*)(dst + m), x);
                    m += 16; k += 16;
                    continue;
                }
                for (const size_t e = k + 16; k < e; ++k) {
                    const uint8_t o = lut[src[k]];
                    dst[m] = o;
                    m += (o != 0);
                }
            }
        }
#endif
        for (; k < rlen; ++k) {
            const uint8_t o = lut[text[s0 + k]];
            dst[m] = o;
            m += (o != 0);
        }
        plen += m;
        out.offsets.push_back(plen);
        pos = next;
    }
    memset(payload + plen, 0, 64);
    out.bytes.len = plen + 64;
    if (consumed) *consumed = limit;
    return true;
}

size_t record_start_from(const uint8_t* text, size_t len, size_t target)
{
    size_t k = target;
    while (k < len) {
        const void* g = memchr(text + k, '>', len - k);
        if (!g) return len;
        const size_t gi = (const uint8_t*)g - text;
        if (gi == 0 || text[gi - 1] == '\n') return gi;
        k = gi + 1;
    }
    return len;
}

void sub_range_bounds(const uint8_t* text, size_t len, int n_sub, int k, size_t* b0, size_t* b1)
{
    const size_t t0 = len / (size_t)n_sub * (size_t)k, t1 = len / (size_t)n_sub * (size_t)(k + 1);
    *b0 = k == 0 ? 0 : record_start_from(text, len, t0);
    *b1 = k + 1 == n_sub ? len : record_start_from(text, len, t1);
}

bool parse_sub_range(const uint8_t* text, size_t len, bool first_chunk, int n_sub, int k, Batch& sub, size_t* start, std::string& err)
{
    size_t b0, b1;
    sub_range_bounds(text, len, n_sub, k, &b0, &b1);
    *start = b0;
    if (b1 > b0 || k == 0) {
        size_t consumed = 0;
        return parse_chunk(text + b0, b1 - b0, first_chunk && k == 0, true, sub, &consumed, err);
    }
    sub.clear();                    // an empty sub-range (a record longer than the sub-range swallowed it)
    sub.text = text + b0;
    sub.offsets.push_back(0);
    return true;
}

void place_sub_batch(const Batch& sub, size_t start, uint64_t rec0, uint64_t byte0, Batch& whole)
{
    const uint64_t n = sub.n();
    if (n) memcpy(whole.bytes.data() + byte0, sub.bytes.data(), (size_t)sub.offsets[n]);
    for (uint64_t i = 0; i < n; ++i) {
        whole.offsets[rec0 + i] = byte0 + sub.offsets[i];
        whole.head[rec0 + i] = Span{ sub.head[i].off + start, sub.head[i].len };
        whole.raw[rec0 + i] = Span{ sub.raw[i].off + start, sub.raw[i].len };
    }
}

void csv_field(std::string& out, const uint8_t* p, size_t n, char delim)
{
    bool quote = n == 0;
    for (size_t i = 0; i < n && !quote; ++i) quote = p[i] == (uint8_t)delim || p[i] == '"' || p[i] == '\n' || p[i] == '\r';
    if (!quote) { out.append((const char*)p, n); return; }
    out.push_back('"');
    for (size_t i = 0; i < n; ++i) {
        if (p[i] == '"') out.push_back('"');
        out.push_back((char)p[i]);
    }
    out.push_back('"');
}

}  // namespace ckhost

// ------------------------------------------------------------------------------------------------
// C ABI of the packer (include/circkit.h): lets a host in another language -- and the CPU-only tests --
// drive the FASTA -> CSR step without touching C++ types.
// ------------------------------------------------------------------------------------------------
struct circkit_fasta_batch {
    ckhost::Batch b;
    std::string err;
};

extern "C" {

int circkit_fasta_parse(const uint8_t* text, size_t n, int first_chunk, int final_chunk, circkit_fasta_batch** out,
                        size_t* consumed)
{
    if (!out || (n && !text)) return CIRCKIT_ERR_INVALID_ARG;
    circkit_fasta_batch* fb = new circkit_fasta_batch();
    *out = fb;
    if (!ckhost::parse_chunk(text, n, first_chunk != 0, final_chunk != 0, fb->b, consumed, fb->err)) return CIRCKIT_ERR_INVALID_ARG;
    return CIRCKIT_OK;
}

const char* circkit_fasta_error(const circkit_fasta_batch* fb) { return fb ? fb->err.c_str() : "null batch"; }
uint64_t circkit_fasta_n_records(const circkit_fasta_batch* fb) { return fb ? fb->b.n() : 0; }
const uint8_t* circkit_fasta_bytes(const circkit_fasta_batch* fb) { return fb ? fb->b.bytes.data() : nullptr; }
const uint64_t* circkit_fasta_offsets(const circkit_fasta_batch* fb) { return fb ? fb->b.offsets.data() : nullptr; }

int circkit_fasta_record(const circkit_fasta_batch* fb, uint64_t i, size_t* head_off, size_t* head_len, size_t* raw_off,
                         size_t* raw_len)
{
    if (!fb || i >= fb->b.n()) return CIRCKIT_ERR_INVALID_ARG;
    if (head_off) *head_off = fb->b.head[i].off;
    if (head_len) *head_len = fb->b.head[i].len;
    if (raw_off) *raw_off = fb->b.raw[i].off;
    if (raw_len) *raw_len = fb->b.raw[i].len;
    return CIRCKIT_OK;
}

void circkit_fasta_free(circkit_fasta_batch* fb) { delete fb; }

}  // extern "C"
