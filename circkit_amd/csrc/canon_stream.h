// canon_stream.h -- the streaming kernel with workgroup-staged input.
//
// Same per-record routine as canon_fast.h (registers, DPP, fused XXH3); what changes is how record bytes reach
// the lanes.  Measured on MI355X (tools/microbench/*copy_bench.hip, 10M x 1000 B pure copies): one 16 B/lane
// request per record per wave at the record's own alignment tops out at 4.4 ms, while flat 16-byte-aligned loads of
// 8 KiB+ contiguous spans by the whole workgroup reach 3.7-3.9 ms -- the record-per-wave LOADS are what costs
// bandwidth, the stores are not.  So: a workgroup takes GROUP consecutive records, its four waves DMA the group's
// byte span into LDS as aligned 16-byte chunks (global_load_lds_dwordx4, two instructions per wave, a ring of
// three buffers = two groups in flight), and after one barrier every wave pulls its two records out of the LDS
// image (two ds_read_b128 + a byte funnel by the record's offset mod 16).  Output goes straight from registers to
// global memory as before.
#pragma once
#include "canon_fast.h"

namespace ck {

constexpr uint32_t STREAM_GROUP = 8;                     // records per group (two per wave)
constexpr uint32_t STREAM_SPAN = 8192;                   // bytes of one LDS image: 8 x 1008 + alignment slack fits
constexpr uint32_t STREAM_BUF_DW = (STREAM_SPAN + 64) / 4;   // + one chunk of read slack for the funnel
constexpr int STREAM_NBUF = 3;

struct StreamGroup {
    uint64_t base;      // byte offset (into a.bytes) of the image's first byte: span start rounded down to 16
    bool ok;            // staged (else: every record of the group goes to the general kernel)
};

// Every wave issues exactly two DMA instructions (chunks wave*128 + {0,64} + lane of the image), whatever the group
// looks like -- the vmcnt bookkeeping of the loop depends on it.  Lanes past the span re-fetch its last chunk;
// unstaged groups fetch the offsets array (always readable).
CK_DEV StreamGroup stream_issue(const CanonArgs& a, uint32_t g, uint32_t n_groups, uint32_t* buf)
{
    const uint32_t gg = g < n_groups ? g : n_groups - 1;
    const uint64_t N = a.n_records, r0 = (uint64_t)gg * STREAM_GROUP, r1 = r0 + STREAM_GROUP < N ? r0 + STREAM_GROUP : N;
    uint64_t s, e;
    sload_2u64(a.offsets + r0, a.offsets + r1, s, e);
    const uint64_t addr = (uint64_t)(uintptr_t)a.bytes + s;
    StreamGroup grp;
    grp.base = s - (addr & 15);
    const uint64_t nbytes = e - grp.base;
    // not staged: the batch's last group (its final chunk would read past the payload), a group whose first chunk
    // would start before the payload (unaligned d_bytes), empty or oversized spans
    grp.ok = g + 1 < n_groups && (addr & 15) <= s && nbytes != 0 && nbytes <= STREAM_SPAN;
    const uint32_t last = grp.ok ? (uint32_t)((nbytes - 1) >> 4) : 0u;
    const uint32_t w = wave_in_block(), t = lane_id();
#pragma unroll
    for (uint32_t i = 0; i < 2; ++i) {
        const uint32_t c = w * 128 + i * 64 + t;
        const uint8_t* src = grp.ok ? a.bytes + grp.base + 16ull * (c < last ? c : last) : (const uint8_t*)a.offsets;
        glds16_async(buf + (w * 128 + i * 64) * 4, src);
    }
    return grp;
}

// bytes [rel + 16t, rel + 16t + 16) of the image, rel = c0*16 + a16: two aligned chunks, funnelled by a16 bytes
CK_DEV u32x4 stream_fetch(const uint32_t* buf, uint32_t c0, uint32_t a16)
{
    const uint32_t* p = buf + 4 * (c0 + lane_id());
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3], w4 = p[4], w5 = p[5], w6 = p[6], w7 = p[7];
    const uint32_t sh = 8 * (a16 & 3);
    switch (a16 >> 2) {          // wave-uniform
    case 0: return u32x4{ funnel(w1, w0, 32 - sh) , funnel(w2, w1, 32 - sh), funnel(w3, w2, 32 - sh), funnel(w4, w3, 32 - sh) };
    case 1: return u32x4{ funnel(w2, w1, 32 - sh) , funnel(w3, w2, 32 - sh), funnel(w4, w3, 32 - sh), funnel(w5, w4, 32 - sh) };
    case 2: return u32x4{ funnel(w3, w2, 32 - sh) , funnel(w4, w3, 32 - sh), funnel(w5, w4, 32 - sh), funnel(w6, w5, 32 - sh) };
    default: return u32x4{ funnel(w4, w3, 32 - sh) , funnel(w5, w4, 32 - sh), funnel(w6, w5, 32 - sh), funnel(w7, w6, 32 - sh) };
    }
}

// loop of one wave of a workgroup; every wave of the workgroup runs the same number of iterations (barriers inside)
CK_DEV void canon_stream_wave_loop(const CanonArgs& a, const uint32_t* lut, uint32_t* ring, uint32_t* blk_count, uint32_t block,
                                   uint32_t nblocks)
{
    const uint32_t N = (uint32_t)a.n_records, n_groups = (N + STREAM_GROUP - 1) / STREAM_GROUP;
    if (block >= n_groups) return;
    const uint32_t w = wave_in_block();
    FastHashConst hc{};
    if (a.out_hash) hc = fast_hash_const();
    const bool stores = a.out_bytes != nullptr || a.out_hash != nullptr;
    // ring state in scalars: cur = the group being processed, nxt = the one in flight behind it
    StreamGroup cur = stream_issue(a, block, n_groups, ring);
    StreamGroup nxt = stream_issue(a, block + nblocks, n_groups, ring + STREAM_BUF_DW);
    vmem_wait<2>();                                  // the first group's two DMAs; the second's may still fly
    block_barrier();
    uint32_t bi = 0;                                 // buffer index of cur
    for (uint32_t g = block; g < n_groups; g += nblocks) {
        const uint32_t bf = bi ? bi - 1 : STREAM_NBUF - 1;                  // (bi + 2) % 3: the buffer freed last iteration
        const StreamGroup fut = stream_issue(a, g + 2 * nblocks, n_groups, ring + bf * STREAM_BUF_DW);
        const uint32_t* img = ring + bi * STREAM_BUF_DW;
        // this wave's two records of group g
        const uint64_t ra = (uint64_t)g * STREAM_GROUP + 2 * w;
        const uint64_t ia = ra < N ? ra : N, ib = ra + 1 < N ? ra + 1 : N, ic = ra + 2 < N ? ra + 2 : N;
        uint64_t o0, o1, o2, o3;
        sload_2u64(a.offsets + ia, a.offsets + ib, o0, o1);
        sload_2u64(a.offsets + ib, a.offsets + ic, o2, o3);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint64_t off = k ? o2 : o0;
            const uint32_t n = (uint32_t)((k ? o3 : o1) - off);
            const uint32_t rec = (uint32_t)ra + k;
            if (rec < N) {
                bool done = false;
                if (cur.ok && fast_eligible(n)) {
                    const uint32_t rel = (uint32_t)(off - cur.base);
                    done = fast_process<false>(a, lut, hc, rec, off, n, stream_fetch(img, rel >> 4, rel & 15));
                }
                if (!done) defer_record(a, blk_count, block, rec);
            }
        }
        // the next group's DMAs (issued one iteration ago) must have landed.  Younger vector-memory instructions:
        // the stores of the previous and of this iteration (>= 2 each: every record stores its bytes or hash, or
        // its deferral; only the batch's last group can hold fewer records, and nothing is waited for after it) and
        // the two DMAs issued above.
        if (stores) vmem_wait<6>(); else vmem_wait<2>();
        block_barrier();
        cur = nxt; nxt = fut;
        bi = bi + 1 == STREAM_NBUF ? 0 : bi + 1;
    }
}

}  // namespace ck
