// Sanitizer driver of the C++ host packer (test infrastructure; built by tests/test_sanitizers.py with
// g++ -fsanitize=address,undefined together with circkit_amd/csrc/fasta_host.cpp -- CPU build only).
// Reads cases from argv[1]: u32 count, then per case {u8 first_chunk, u8 final_chunk, u32 len, bytes}.  Every text is copied
// into a heap block of exactly its length, so a read past either end is an ASan report.  Writes per case to stdout:
// i32 rc, u64 consumed, u64 n_records, n x {head_off, head_len, raw_off, raw_len} (u64), (n + 1) x u64 offsets, payload.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "../../circkit_amd/csrc/fasta_host.h"
#include "../../include/circkit.h"

static void put(const void* p, size_t n) { if (n && fwrite(p, 1, n, stdout) != n) abort(); }

// The CLI's parser pool in small (circkit_cli.cpp, stage 2): the chunk in n_sub sub-ranges, one THREAD each for phase A and
// again for phase B, the chunk's CSR laid out in between.  Returns false on a parse error.
static bool parse_in_sub_ranges(const uint8_t* text, size_t len, bool first, int n_sub, ckhost::Batch& whole)
{
    std::vector<ckhost::Batch> sub(n_sub);
    std::vector<size_t> start(n_sub, 0);
    std::vector<std::string> err(n_sub);
    std::vector<char> ok(n_sub, 1);
    std::vector<std::thread> th;
    for (int k = 0; k < n_sub; ++k)
        th.emplace_back([&, k] { ok[k] = ckhost::parse_sub_range(text, len, first, n_sub, k, sub[k], &start[k], err[k]) ? 1 : 0; });
    for (auto& t : th) t.join();
    for (int k = 0; k < n_sub; ++k) if (!ok[k]) return false;
    std::vector<uint64_t> rec0(n_sub), byte0(n_sub);
    uint64_t rec = 0, bytes = 0;
    for (int k = 0; k < n_sub; ++k) { rec0[k] = rec; byte0[k] = bytes; rec += sub[k].n(); bytes += sub[k].offsets[sub[k].n()]; }
    whole.clear();
    whole.text = text;
    whole.head.resize(rec); whole.raw.resize(rec); whole.offsets.resize(rec + 1);
    whole.bytes.reserve(bytes + 64);
    whole.bytes.len = bytes + 64;
    whole.offsets[rec] = bytes;
    memset(whole.bytes.data() + bytes, 0, 64);
    th.clear();
    for (int k = 0; k < n_sub; ++k) th.emplace_back([&, k] { ckhost::place_sub_batch(sub[k], start[k], rec0[k], byte0[k], whole); });
    for (auto& t : th) t.join();
    return true;
}
static bool same_batch(const ckhost::Batch& a, const ckhost::Batch& b)
{
    if (a.n() != b.n() || a.offsets != b.offsets) return false;
    for (size_t i = 0; i < a.n(); ++i)
        if (a.head[i].off != b.head[i].off || a.head[i].len != b.head[i].len || a.raw[i].off != b.raw[i].off || a.raw[i].len != b.raw[i].len) return false;
    return memcmp(a.bytes.data(), b.bytes.data(), (size_t)a.offsets[a.n()] + 64) == 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint32_t count = 0;
    if (fread(&count, 4, 1, f) != 1) return 2;
    for (uint32_t c = 0; c < count; ++c) {
        uint8_t flags[2];
        uint32_t len = 0;
        if (fread(flags, 1, 2, f) != 2 || fread(&len, 4, 1, f) != 1) return 2;
        uint8_t* text = (uint8_t*)malloc(len ? len : 1);
        if (len && fread(text, 1, len, f) != len) return 2;
        circkit_fasta_batch* fb = nullptr;
        size_t consumed = 0;
        const int32_t rc = circkit_fasta_parse(len ? text : nullptr, len, flags[0], flags[1], &fb, &consumed);
        const uint64_t n = rc == 0 ? circkit_fasta_n_records(fb) : 0, cons = consumed;
        put(&rc, 4); put(&cons, 8); put(&n, 8);
        for (uint64_t i = 0; i < n; ++i) {
            size_t v[4];
            if (circkit_fasta_record(fb, i, &v[0], &v[1], &v[2], &v[3]) != 0) abort();
            const uint64_t w[4] = { v[0], v[1], v[2], v[3] };
            put(w, 32);
        }
        if (rc == 0) {
            const uint64_t* offs = circkit_fasta_offsets(fb);
            put(offs, (n + 1) * 8);
            put(circkit_fasta_bytes(fb), offs[n]);
            // csv quoting of every header (uniq --table's id fields go through it)
        }
        circkit_fasta_free(fb);
        // whole texts only (final_chunk): the same CSR from every split into sub-ranges, threads and all
        if (flags[1]) {
            ckhost::Batch one;
            std::string err;
            size_t cons2 = 0;
            const bool ok1 = ckhost::parse_chunk(text, len, flags[0] != 0, true, one, &cons2, err);
            for (int n_sub : { 1, 2, 3, 5, 16 }) {
                ckhost::Batch many;
                const bool okn = parse_in_sub_ranges(text, len, flags[0] != 0, n_sub, many);
                if (okn != ok1 || (ok1 && !same_batch(one, many))) { fprintf(stderr, "case %u: %d sub-ranges differ from one\n", c, n_sub); return 3; }
            }
        }
        free(text);
    }
    fclose(f);
    return 0;
}
