// Sanitizer driver of the C++ host packer (test infrastructure; built by tests/test_sanitizers.py with
// g++ -fsanitize=address,undefined together with circkit_amd/csrc/fasta_host.cpp -- CPU build only).
// Reads cases from argv[1]: u32 count, then per case {u8 first_chunk, u8 final_chunk, u32 len, bytes}.  Every text is copied
// into a heap block of exactly its length, so a read past either end is an ASan report.  Writes per case to stdout:
// i32 rc, u64 consumed, u64 n_records, n x {head_off, head_len, raw_off, raw_len} (u64), (n + 1) x u64 offsets, payload.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/circkit.h"

static void put(const void* p, size_t n) { if (n && fwrite(p, 1, n, stdout) != n) abort(); }

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
        free(text);
    }
    fclose(f);
    return 0;
}
