// fasta_host.h -- host side of the canonicalize / uniq path: FASTA record parsing with seq_io semantics and
// the CSR batch packer that feeds the GPU.  Replaces, on the host:
//   seq_io 0.3.2 fasta::Reader + parallel_fasta record hand-off   (call sites src/canonicalize.rs:14-20, src/uniq.rs:24-32)
//   needletail 0.5.1 sequence::normalize(seq, false) in the worker closure (src/canonicalize.rs:24-27, src/uniq.rs:35-38)
// Everything here is plain C++ host logic (no GPU); the compute it feeds is the C ABI of include/circkit.h.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string>
#include <vector>

namespace ckhost {

struct Span { size_t off, len; };

// Growable byte buffer with a pluggable allocator, so the packer can write the CSR payload straight into
// pinned host memory (circkit_host_alloc) and the H2D copy is a plain DMA.
struct ByteBuf {
    uint8_t* p = nullptr;
    size_t len = 0, cap = 0;
    void* (*alloc)(size_t) = nullptr;       // nullptr = malloc / free
    void (*release)(void*) = nullptr;
    ByteBuf() = default;
    ByteBuf(const ByteBuf&) = delete;
    ByteBuf& operator=(const ByteBuf&) = delete;
    ~ByteBuf();
    void reserve(size_t n);                 // keeps the first `len` bytes
    uint8_t* data() { return p; }
    const uint8_t* data() const { return p; }
    size_t size() const { return len; }
};

// One parsed chunk of FASTA text: per record the header span, the raw sequence span (interior line breaks kept,
// final line break dropped -- what seq_io's RefRecord::seq() returns) and the normalized bytes in CSR layout.
struct Batch {
    const uint8_t* text = nullptr;          // the chunk the spans point into (owned by the caller)
    std::vector<Span> head, raw;
    ByteBuf bytes;                          // normalized payload, + 64 bytes of zero padding
    std::vector<uint64_t> offsets;          // n_records + 1
    size_t n() const { return head.size(); }
    void clear() { head.clear(); raw.clear(); bytes.len = 0; offsets.clear(); }
};

// needletail normalize LUT: 0 = drop (space, \t, \r, \n), otherwise the output byte.
const uint8_t* normalize_lut();

// Parses every complete record in text[0, n).  If `final_chunk` is false the last record may be cut by the chunk
// end, so parsing stops before the last record start and *consumed tells the caller where to resume; with
// `final_chunk` everything is consumed.  `first_chunk` enables the leading-blank-line skip and the
// "must start with '>'" check.  Returns false and sets err on a format error.
bool parse_chunk(const uint8_t* text, size_t n, bool first_chunk, bool final_chunk, Batch& out, size_t* consumed,
                 std::string& err);

// ---- a chunk parsed by several threads (the CLI's parser pool, round 4) ---------------------------------------------------
// The chunk text[0, len) -- cut at a record start by the reader, so it holds whole records only -- is split into n_sub
// sub-ranges at record starts; every sub-range is parsed on its own into a private Batch (phase A, any thread), the chunk's
// CSR is laid out by prefix sums over the sub-ranges' record and byte counts, and every sub-range is copied to its place in
// ONE contiguous CSR (phase B, any thread): the GPU call wants one payload and one offsets array.
// First record start (a '>' at the start of a line) at or behind `target`; len if there is none.
size_t record_start_from(const uint8_t* text, size_t len, size_t target);
// [*b0, *b1) of sub-range k of n_sub: record starts next to the k-th and (k+1)-th n_sub-th of the text
void sub_range_bounds(const uint8_t* text, size_t len, int n_sub, int k, size_t* b0, size_t* b1);
// phase A: parses sub-range k into `sub` (spans relative to text + *start)
bool parse_sub_range(const uint8_t* text, size_t len, bool first_chunk, int n_sub, int k, Batch& sub, size_t* start, std::string& err);
// phase B: `sub` (parsed from text + start) becomes records [rec0, rec0 + sub.n()) and payload bytes [byte0, ...) of `whole`,
// whose head / raw / offsets are sized for all records and whose payload buffer holds all bytes already
void place_sub_batch(const Batch& sub, size_t start, uint64_t rec0, uint64_t byte0, Batch& whole);

// seq_io Record::id(): header up to the first space.
inline Span record_id(const uint8_t* text, Span head)
{
    size_t k = 0;
    while (k < head.len && text[head.off + k] != ' ') ++k;
    return Span{ head.off, k };
}

// csv crate field quoting (QuoteStyle::Necessary): quote when the field holds the delimiter, a quote, CR or LF
// (or is empty); quotes are doubled.
void csv_field(std::string& out, const uint8_t* p, size_t n, char delim);

}  // namespace ckhost
